/* capi_harness.c — a COMPILED caller of the C ABI (include/masp_hip.h), C99/C11, no Python in the call path.
 *
 * Plays the part SURVEY.md §7 step 5 / §8(b) give "a C++ harness playing the Rust caller": what a fork of masp_proofs would do where
 * /root/reference/masp_proofs/src/prover.rs:156-261 calls bellperson — fill masp_hip_r1cs / masp_hip_job by hand from flat arrays,
 * masp_hip_ctx_create_ex -> masp_hip_circuit_load -> masp_hip_prove / masp_hip_prove_batch -> 192-byte proofs — linked against
 * libmasp_hip.so like a Rust `#[link(name = "masp_hip")]` block would be.  The pytest wrapper (tests/test_capi_harness.py) writes
 * the circuit, the Parameters bytes and the jobs to a file, runs this program and compares the proofs it writes with the oracle's.
 *
 *   capi_harness --abi                      layout + the entry points that need no GPU (runs in the CPU suite)
 *   capi_harness <case.bin> <proofs.bin>    the proving flow on device 0 (-m gpu)
 *
 * case.bin (little-endian):  "MHH1" | u32 n_inputs n_aux n_constraints | 3 x { u32 nnz | rowptr[n_constraints + 1] | col[nnz] |
 *   coef[nnz][32] } | u64 params_len | params | u32 n_jobs | n_jobs x { u32 aux_form | u32 has_abc | inputs[n_inputs][32] |
 *   aux[n_aux][32] | has_abc ? a, b, c [n_constraints + n_inputs][32] each | r[32] | s[32] }
 * proofs.bin: 192 bytes from masp_hip_prove(job 0) | n_jobs x 192 bytes from ONE masp_hip_prove_batch over all jobs */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "masp_hip.h"

/* the layout once more, by the C compiler that builds THIS caller (the header asserts it too, under C11 and C++11) */
typedef char job_is_120[sizeof(masp_hip_job) == 120 ? 1 : -1];
typedef char job_r_at_48[offsetof(masp_hip_job, r) == 48 ? 1 : -1];
typedef char job_aux_form_at_112[offsetof(masp_hip_job, aux_form) == 112 ? 1 : -1];
typedef char r1cs_is_88[sizeof(masp_hip_r1cs) == 88 ? 1 : -1];
typedef char options_is_76[sizeof(masp_hip_options) == 76 ? 1 : -1];

#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            fprintf(stderr, "capi_harness: " __VA_ARGS__); \
            fprintf(stderr, "\n");                        \
            return 1;                                     \
        }                                                 \
    } while (0)

static int abi(void) {
    masp_hip_options o;
    masp_hip_ctx* ctx = (masp_hip_ctx*)0;
    int rc, n;
    printf("sizeof_job %u\n", (unsigned)sizeof(masp_hip_job));
    printf("sizeof_r1cs %u\n", (unsigned)sizeof(masp_hip_r1cs));
    printf("sizeof_options %u\n", (unsigned)sizeof(masp_hip_options));
    printf("offsetof_job_inputs %u\n", (unsigned)offsetof(masp_hip_job, inputs));
    printf("offsetof_job_r %u\n", (unsigned)offsetof(masp_hip_job, r));
    printf("offsetof_job_s %u\n", (unsigned)offsetof(masp_hip_job, s));
    printf("offsetof_job_aux_form %u\n", (unsigned)offsetof(masp_hip_job, aux_form));
    printf("offsetof_options_window_bits_b2 %u\n", (unsigned)offsetof(masp_hip_options, window_bits_b2));
    memset(&o, 0xff, sizeof o);
    masp_hip_options_default(&o);
    CHECK(o.struct_size == sizeof o && o.slots == 0 && o.window_bits_b2 == 0, "masp_hip_options_default did not clear the struct");
    CHECK(strcmp(masp_hip_strerror(MASP_HIP_OK), "ok") == 0 && masp_hip_strerror(MASP_HIP_E_NO_DEVICE)[0] != 0, "masp_hip_strerror");
    CHECK(masp_hip_last_error((const masp_hip_ctx*)0)[0] == 0, "masp_hip_last_error(NULL)");
    n = masp_hip_device_count();
    printf("devices %d\n", n);
    {
        char pci[64];
        rc = masp_hip_device_pci_bus_id(0, pci, sizeof pci);
        CHECK(n > 0 ? rc == MASP_HIP_OK && strlen(pci) >= 12 : rc == MASP_HIP_E_NO_DEVICE, "masp_hip_device_pci_bus_id: %d", rc);
        CHECK(masp_hip_device_pci_bus_id(0, pci, 4) == MASP_HIP_E_INVALID_ARG, "a buffer too short for a PCI address must be refused");
        if (n > 0) printf("pci %s\n", pci);
    }
    rc = masp_hip_ctx_create(0, &ctx);
    printf("ctx_create %d\n", rc);
    if (n == 0) CHECK(rc == MASP_HIP_E_NO_DEVICE && ctx == (masp_hip_ctx*)0, "without a device masp_hip_ctx_create must fail with MASP_HIP_E_NO_DEVICE");
    CHECK(masp_hip_ctx_get_options((const masp_hip_ctx*)0, &o) == MASP_HIP_E_INVALID_ARG, "get_options(NULL)");
    CHECK(masp_hip_prove_batch((masp_hip_ctx*)0, 0, (const masp_hip_job*)0, (uint8_t*)0) == MASP_HIP_E_INVALID_ARG, "prove_batch(NULL)");
    if (ctx) masp_hip_ctx_destroy(ctx);
    printf("abi ok\n");
    return 0;
}

typedef struct {
    uint8_t* p;
    size_t len, pos;
} reader;
static const uint8_t* take(reader* r, size_t n) {
    const uint8_t* q;
    if (r->len - r->pos < n) {
        fprintf(stderr, "capi_harness: case file truncated\n");
        exit(2);
    }
    q = r->p + r->pos;
    r->pos += n;
    return q;
}
static uint32_t take_u32(reader* r) {
    uint32_t v;
    memcpy(&v, take(r, 4), 4);
    return v;
}
static uint64_t take_u64(reader* r) {
    uint64_t v;
    memcpy(&v, take(r, 8), 8);
    return v;
}

int main(int argc, char** argv) {
    FILE* f;
    reader rd;
    masp_hip_r1cs cs;
    masp_hip_options opt, got;
    masp_hip_ctx* ctx = (masp_hip_ctx*)0;
    masp_hip_job* jobs;
    const uint8_t* params;
    uint64_t params_len;
    uint32_t n_jobs, j, flags = 0, nrows;
    uint8_t *proofs, lone[192];
    const uint32_t** ptrs[3][2];
    const uint8_t** coefs[3];
    int rc, dev = 0, i, devices;
    int32_t status[4];
    uint64_t requeued = 0, counts[4];
    long sz;
    struct {  /* a caller built against the round-4 header: its options struct ends before window_bits_h_lone */
        masp_hip_options head;
        uint32_t canary[4];
    } old_caller;

    if (argc == 2 && strcmp(argv[1], "--abi") == 0) return abi();
    CHECK(argc == 3, "usage: capi_harness --abi | capi_harness <case.bin> <proofs.bin>");
    f = fopen(argv[1], "rb");
    CHECK(f != NULL, "cannot open %s", argv[1]);
    fseek(f, 0, SEEK_END);
    sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    rd.p = (uint8_t*)malloc((size_t)sz);
    rd.len = (size_t)sz;
    rd.pos = 0;
    CHECK(rd.p != NULL && fread(rd.p, 1, rd.len, f) == rd.len, "cannot read %s", argv[1]);
    fclose(f);
    CHECK(memcmp(take(&rd, 4), "MHH1", 4) == 0, "bad magic");

    /* ---- masp_hip_r1cs, filled field by field as a binding would ---- */
    memset(&cs, 0, sizeof cs);
    cs.n_inputs = take_u32(&rd);
    cs.n_aux = take_u32(&rd);
    cs.n_constraints = take_u32(&rd);
    nrows = cs.n_constraints + cs.n_inputs;
    ptrs[0][0] = &cs.a_rowptr; ptrs[0][1] = &cs.a_col; coefs[0] = &cs.a_coef;
    ptrs[1][0] = &cs.b_rowptr; ptrs[1][1] = &cs.b_col; coefs[1] = &cs.b_coef;
    ptrs[2][0] = &cs.c_rowptr; ptrs[2][1] = &cs.c_col; coefs[2] = &cs.c_coef;
    for (i = 0; i < 3; ++i) {
        const uint32_t nnz = take_u32(&rd);
        /* (the file keeps every array 4-byte aligned: all sections are multiples of four bytes) */
        *ptrs[i][0] = (const uint32_t*)(const void*)take(&rd, 4 * ((size_t)cs.n_constraints + 1));
        *ptrs[i][1] = (const uint32_t*)(const void*)take(&rd, 4 * (size_t)nnz);
        *coefs[i] = take(&rd, 32 * (size_t)nnz);
        CHECK((*ptrs[i][0])[cs.n_constraints] == nnz, "matrix %d: rowptr does not end at nnz", i);
    }
    params_len = take_u64(&rd);
    params = take(&rd, (size_t)params_len);
    n_jobs = take_u32(&rd);
    CHECK(n_jobs >= 1 && n_jobs <= 4096, "n_jobs");
    jobs = (masp_hip_job*)calloc(n_jobs, sizeof *jobs);
    proofs = (uint8_t*)calloc(n_jobs, 192);
    CHECK(jobs != NULL && proofs != NULL, "out of memory");
    for (j = 0; j < n_jobs; ++j) {
        masp_hip_job* J = &jobs[j];
        const uint32_t aux_form = take_u32(&rd), has_abc = take_u32(&rd);
        J->circuit = 3;   /* any slot below MASP_HIP_MAX_CIRCUITS: the three conventional ones are not special */
        J->aux_form = aux_form;
        J->reserved = 0;
        J->inputs = take(&rd, 32 * (size_t)cs.n_inputs);
        J->aux = take(&rd, 32 * (size_t)cs.n_aux);
        if (has_abc) {
            J->a = take(&rd, 32 * (size_t)nrows);
            J->b = take(&rd, 32 * (size_t)nrows);
            J->c = take(&rd, 32 * (size_t)nrows);
        }
        memcpy(J->r, take(&rd, 32), 32);
        memcpy(J->s, take(&rd, 32), 32);
    }
    CHECK(rd.pos == rd.len, "trailing bytes in the case file");

    /* ---- the context: explicit options, one device ---- */
    devices = masp_hip_device_count();
    CHECK(devices >= 1, "no HIP device: %s", masp_hip_strerror(MASP_HIP_E_NO_DEVICE));
    masp_hip_options_default(&opt);
    opt.slots = 2;
    opt.batch_cap = 8;   /* the job list below is cut into launch sequences of at most 8 proofs */
    rc = masp_hip_ctx_create_ex(&dev, 1, &opt, &ctx);
    CHECK(rc == MASP_HIP_OK && ctx != NULL, "masp_hip_ctx_create_ex: %s", masp_hip_strerror(rc));
    memset(&got, 0, sizeof got);
    got.struct_size = (uint32_t)sizeof got;
    rc = masp_hip_ctx_get_options(ctx, &got);
    CHECK(rc == MASP_HIP_OK && got.struct_size == sizeof got && got.slots == 2 && got.batch_cap == 8 && got.hw_queues >= 1,
          "masp_hip_ctx_get_options: rc %d slots %d batch_cap %d hw_queues %d", rc, (int)got.slots, (int)got.batch_cap, (int)got.hw_queues);
    /* ... and as a caller whose struct is SHORTER than the library's: nothing behind its end may be written */
    memset(&old_caller, 0xa5, sizeof old_caller);
    old_caller.head.struct_size = (uint32_t)offsetof(masp_hip_options, window_bits_h_lone);
    rc = masp_hip_ctx_get_options(ctx, &old_caller.head);
    CHECK(rc == MASP_HIP_OK && old_caller.head.struct_size == offsetof(masp_hip_options, window_bits_h_lone) && old_caller.head.slots == 2,
          "truncated get_options: rc %d struct_size %u", rc, (unsigned)old_caller.head.struct_size);
    CHECK((uint32_t)old_caller.head.window_bits_h_lone == 0xa5a5a5a5u && (uint32_t)old_caller.head.window_bits_b2 == 0xa5a5a5a5u &&
              old_caller.canary[0] == 0xa5a5a5a5u,
          "masp_hip_ctx_get_options wrote past the caller's struct_size");
    printf("options slots %d batch_cap %d hw_queues %d\n", (int)got.slots, (int)got.batch_cap, (int)got.hw_queues);

    /* ---- load, prove ---- */
    CHECK(masp_hip_prove_batch(ctx, n_jobs, jobs, proofs) == MASP_HIP_E_NOT_LOADED, "proving on an empty slot must say MASP_HIP_E_NOT_LOADED");
    rc = masp_hip_circuit_load(ctx, 3, params, (size_t)params_len, &cs);
    CHECK(rc == MASP_HIP_OK, "masp_hip_circuit_load: %s (%s)", masp_hip_strerror(rc), masp_hip_last_error(ctx));
    CHECK(masp_hip_circuit_flags(ctx, 3, &flags) == MASP_HIP_OK, "masp_hip_circuit_flags");
    printf("circuit_flags %u\n", (unsigned)flags);
    /* masp_hip_prove takes canonical aux only: job 0 of the case file is canonical by convention */
    CHECK(jobs[0].aux_form == MASP_HIP_AUX_CANONICAL, "job 0 must be canonical");
    rc = masp_hip_prove(ctx, 3, jobs[0].inputs, jobs[0].aux, jobs[0].a, jobs[0].b, jobs[0].c, jobs[0].r, jobs[0].s, lone);
    CHECK(rc == MASP_HIP_OK, "masp_hip_prove: %s (%s)", masp_hip_strerror(rc), masp_hip_last_error(ctx));
    rc = masp_hip_prove_batch(ctx, n_jobs, jobs, proofs);
    CHECK(rc == MASP_HIP_OK, "masp_hip_prove_batch: %s (%s)", masp_hip_strerror(rc), masp_hip_last_error(ctx));
    /* the error contract: a reserved word that is not zero, a scalar >= r — and outputs untouched on failure */
    {
        uint8_t keep[192], big[32];
        masp_hip_job bad = jobs[0];
        memcpy(keep, proofs, 192);
        bad.reserved = 1;
        CHECK(masp_hip_prove_batch(ctx, 1, &bad, proofs) == MASP_HIP_E_INVALID_ARG, "reserved != 0 must be refused");
        bad.reserved = 0;
        memset(big, 0xff, 32);
        memcpy(bad.r, big, 32);
        CHECK(masp_hip_prove_batch(ctx, 1, &bad, proofs) == MASP_HIP_E_SCALAR_RANGE, "r >= modulus must be refused");
        CHECK(memcmp(keep, proofs, 192) == 0, "outputs must only be written on success");
    }
    CHECK(masp_hip_ctx_device_count(ctx) == 1, "device_count");
    CHECK(masp_hip_ctx_device_proofs(ctx, counts, 4) == MASP_HIP_OK && counts[0] == (uint64_t)n_jobs + 1, "device_proofs: %lu", (unsigned long)counts[0]);
    CHECK(masp_hip_ctx_device_status(ctx, status, 4, &requeued) == MASP_HIP_OK && status[0] == MASP_HIP_OK && requeued == 0, "device_status");
    CHECK(masp_hip_sync(ctx) == MASP_HIP_OK, "masp_hip_sync");
    masp_hip_ctx_destroy(ctx);

    f = fopen(argv[2], "wb");
    CHECK(f != NULL, "cannot write %s", argv[2]);
    CHECK(fwrite(lone, 1, 192, f) == 192 && fwrite(proofs, 192, n_jobs, f) == n_jobs, "short write");
    fclose(f);
    printf("proved %u jobs\n", (unsigned)n_jobs);
    free(jobs);
    free(proofs);
    free(rd.p);
    return 0;
}

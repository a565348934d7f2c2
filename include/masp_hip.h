/* masp_hip.h — C ABI of the MI355X-native Groth16 prover for the MASP Spend / Output / Convert circuits.
 *
 * The reference has no C ABI of its own for this path; these entry points sit exactly
 * where masp_proofs calls into bellperson, i.e. what a Rust FFI layer in a fork of masp_proofs binds:
 *
 *   masp_hip_circuit_load   <- bellman::groth16::Parameters::<Bls12>::read(_, false)
 *                              (/root/reference/masp_proofs/src/lib.rs:336-341) + the circuit's static
 *                              shape that bellperson re-derives on every proof (DensityTracker, A.3 step 2)
 *   masp_hip_prove          <- bellman::groth16::create_random_proof / create_proof
 *                              (/root/reference/masp_proofs/src/sapling/prover.rs:117,202,252) followed by
 *                              Proof::write (/root/reference/masp_proofs/src/prover.rs:190-193,218-221,245-248)
 *   masp_hip_prove_batch    <- the serial per-description loops of SaplingBuilder::build
 *                              (/root/reference/masp_primitives/src/transaction/components/sapling/builder.rs:935-1140)
 *   masp_hip_msm_g1/_g2,
 *   masp_hip_quotient_h     <- bellperson multiexp / EvaluationDomain (un-vendored);
 *                              exposed so each kernel family can be checked and timed on its own.
 *
 * Conventions: C linkage, no exceptions across the boundary, integer return codes (0 = ok), the caller
 * owns every buffer, outputs are written only on success.  A context may be shared between threads:
 * masp_hip_prove / masp_hip_prove_batch are re-entrant (concurrent callers each get batches on their own
 * stream + scratch from the context's pool, so their proofs overlap on the device); circuit loads and the
 * building-block / measurement entry points take the context exclusively.
 * Field elements cross the boundary as 32-byte little-endian
 * canonical integers (`Scalar::to_repr()`), points in the zcash encodings of the bellman wire format.
 */
#ifndef MASP_HIP_H
#define MASP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MASP_HIP_OK 0
#define MASP_HIP_E_INVALID_ARG 1          /* NULL / out-of-range argument */
#define MASP_HIP_E_PARAMS_FORMAT 2        /* undecodable Parameters bytes (the reference panics: lib.rs:337) */
#define MASP_HIP_E_PARAMS_SHAPE 3         /* query lengths disagree with the circuit (SURVEY.md App. C invariants) */
#define MASP_HIP_E_NO_DEVICE 4            /* no gfx950 device / HIP extension unusable — there is NO CPU fallback */
#define MASP_HIP_E_HIP 5                  /* HIP runtime error (see masp_hip_last_error) */
#define MASP_HIP_E_UNEXPECTED_IDENTITY 6  /* delta_g1 / delta_g2 is the identity (bellperson SynthesisError::UnexpectedIdentity) */
#define MASP_HIP_E_NOT_LOADED 7           /* circuit slot empty */
#define MASP_HIP_E_SCALAR_RANGE 8         /* a scalar >= r was supplied */

#define MASP_HIP_MAX_CIRCUITS 8
/* conventional slots for the three MASP circuits */
#define MASP_HIP_SPEND 0
#define MASP_HIP_OUTPUT 1
#define MASP_HIP_CONVERT 2

typedef struct masp_hip_ctx masp_hip_ctx;

/* Static R1CS of one circuit as three CSR matrices over n_constraints rows.  Column v < n_inputs is
 * Input(v) (Input(0) = ONE), otherwise Aux(v - n_inputs).  Terms are merged per variable and non-zero.
 * coef: 32 bytes little-endian canonical per term.  (What `Circuit::synthesize` emits through
 * `ConstraintSystem::enforce`, /root/reference/masp_proofs/src/circuit/sapling.rs:139-596, convert.rs:29-128.) */
typedef struct {
    uint32_t n_inputs, n_aux, n_constraints;
    const uint32_t* a_rowptr; const uint32_t* a_col; const uint8_t* a_coef;
    const uint32_t* b_rowptr; const uint32_t* b_col; const uint8_t* b_coef;
    const uint32_t* c_rowptr; const uint32_t* c_col; const uint8_t* c_coef;
} masp_hip_r1cs;

/* One proving job (masp_hip_prove_batch). a/b/c may be NULL: they are then computed on the GPU from the
 * circuit's static R1CS, so only the assignment crosses PCIe. */
typedef struct {
    uint32_t circuit;          /* slot */
    const uint8_t* inputs;     /* n_inputs x 32 (inputs[0] = ONE) */
    const uint8_t* aux;        /* n_aux x 32 */
    const uint8_t* a;          /* (n_constraints + n_inputs) x 32, or NULL */
    const uint8_t* b;
    const uint8_t* c;
    uint8_t r[32], s[32];      /* Groth16 blinding scalars (the reference draws them from OsRng) */
    uint32_t aux_form;         /* MASP_HIP_AUX_CANONICAL (0): aux[i] = Scalar::to_repr(), 32 bytes little-endian canonical;
                                  MASP_HIP_AUX_MONTGOMERY (1): aux[i] = the Montgomery residue a * 2^256 mod r as four little-endian
                                  u64 limbs — the IN-MEMORY form of blst_fr / bls12_381::Scalar: a binding can hand
                                  over the Vec<Scalar> of its recording constraint system without converting 100 000 elements per
                                  Spend (libmasp_host does: 16 % of its synthesis time).  inputs, a, b, c, r, s stay canonical. */
    uint32_t reserved;         /* must be 0 (MASP_HIP_E_INVALID_ARG otherwise).  ABI note: aux_form and reserved were appended in round 3
                                  (sizeof 112 -> 120 on LP64: 4 + 4 padding + five pointers + 64 bytes of r | s + these two words); the
                                  struct carries no size field, so a binding built against the older header must be rebuilt —
                                  INTEGRATION.md "ABI revisions".  The layout is asserted at the end of this header */
} masp_hip_job;
#define MASP_HIP_AUX_CANONICAL 0
#define MASP_HIP_AUX_MONTGOMERY 1

/* Number of HIP devices this process can see (0 if the runtime is unusable). */
int masp_hip_device_count(void);
/* PCI address of device `device` as "0000:c1:00.0" (cap >= 13): which card of the box a context proves on — its clock and power are
 * the files under /sys/bus/pci/devices/<address>/hwmon (bench.py's clock watch reads them during the timed regions). */
int masp_hip_device_pci_bus_id(int device, char* out, size_t cap);

/* Tuning of a prover, fixed when the context is created (masp_hip_ctx_create_ex).  Every field: 0 = the default.  The
 * library reads NO environment variables; bench.py and the tools translate their MASP_HIP_* variables into this. */
typedef struct {
    uint32_t struct_size;            /* sizeof(masp_hip_options) of the caller's header (versioning) */
    int32_t slots;                   /* batches in flight per device, each on its own HIP streams + scratch: 1..64 (default 4 since round 6: +1.7 % host to
                                        host over 3 at 16 hardware queues — a batch uploads while three compute; 42 GB of scratch per slot
                                        for Spend batches; profiles/r06_slots_3_4_5_at_16_hardware_queues.txt) */
    int32_t batch_cap;               /* proofs per launch sequence: 1..256 (default 256 = BASELINE.json configs[3]) */
    int32_t ntt_sub_batch;           /* proofs per sub-batch of the quotient's transforms (default 8: 160 MiB of work buffers
                                        stay in the Infinity Cache); -1 = the whole batch at once */
    int32_t window_bits_h;           /* window width of the h query (default: 16 from 49 152 points, 15 from 16 384) */
    int32_t window_bits_la;          /* ... of the l and a queries (default: from the expected non-trivial scalars) */
    int32_t window_bits_b;           /* ... of b_g1 / b_g2 */
    int32_t window_bits_b2_lone;     /* ... of the second b_g2 table set used by lone proofs (default 8); -1 = not built */
    int32_t witness_nontrivial_percent; /* share of a witness that is neither 0 nor 1, for window selection (default 30) */
    int32_t bucket_tree_levels;      /* levels of shared-inversion affine additions in front of the bucket accumulation of a batch
                                        (default 4, fewer for very short bucket runs); -1 = none (XYZZ accumulation only) */
    int32_t bucket_tree_sub_batch;   /* proofs that go through the tree at a time (default 86: a 256-proof batch in three; its scratch is ~0.4 GB per Spend proof) */
    int32_t bucket_tree_levels_g2;   /* the same for the G2 MSM if it should differ (default: bucket_tree_levels) */
    int32_t bucket_tree_scratch_mb;  /* upper bound of the tree's scratch per slot in MiB (default 0: whatever the device gives).  If the
                                        scratch of a sub-batch does not fit — this bound, or the device is out of memory — the sub-batch
                                        is halved (down to 8 proofs), then the rest of the batch goes through the XYZZ accumulation: same
                                        bytes, fewer proofs per second, never an error */
    int32_t bucket_tree_fallback_proofs; /* OUTPUT of masp_hip_ctx_get_options (ignored on input): proofs whose bucket runs went through
                                        the XYZZ accumulation for lack of tree scratch; bucket_tree_sub_batch there = the sub-batch in use now */
    int32_t lone_proof_graph;        /* 1: a batch of fewer than 8 proofs replays a captured HIP graph of its ~250 launches from its third
                                        call on.  Default off: with ROCm 7.2 the replay of this five-stream graph takes 11.5 ms where the
                                        launches enqueued one by one take 5.7 (profiles/r04_lone_proof_graph_ab.txt); the bytes are the same */
    int32_t hw_queues;               /* OUTPUT of masp_hip_ctx_get_options (ignored on input; was reserved[0]): the hardware queues the HIP
                                        runtime of this process spreads its streams over, MEASURED when the context was created (as many
                                        single-wave kernels as the context's slots have streams, at most 21, launched at once on streams of
                                        their own, once per process and device: how many ran concurrently).  A slot owns five streams; with
                                        fewer queues than streams, independent kernels wait for each other (round 4, 8 queues: -3 ... -10 %
                                        proofs/s) — and with MORE than MASP_HIP_MAX_USEFUL_HW_QUEUES in the process every dispatch is slower
                                        (see there).  The runtime's default is FOUR (profiles/r05_hw_queues_probe.txt); it reads
                                        GPU_MAX_HW_QUEUES at its first call — see masp_hip_runtime_prepare */
    int32_t window_bits_h_lone;      /* (round 5, appended: struct_size tells) window width of the h query's OWN table, which only lone
                                        proofs use — a batch runs h and l as one MSM over the merged table on window_bits_h.  Narrower
                                        windows mean more additions in the chip-filling accumulation and far fewer buckets in the
                                        latency-bound tails a lone proof waits for.  Default: see resolve_options (prover.hip) */
    int32_t digit_recoding;          /* RESERVED, must be 0.  (Round 5: 1 = width-(c + 1) non-adjacent-form digits over a table of 2^t P for every
                                        bit position t — 8 - 15 % fewer bucket additions for 16.6 GB of tables per Spend circuit, which is beyond
                                        the ~3.5 GiB of randomly gathered table an XCD's L2 TLB reaches: Spend -8 %, Output +2 % proofs/s.  Built,
                                        bit-exact, measured, removed in round 6: EXPERIMENTS.md, profiles/r05_naf_digits_*.txt.)  A context is
                                        created with 0 here whatever the caller passes; masp_hip_ctx_get_options returns 0 */
    int32_t window_bits_b2;          /* (round 6, appended) window width of the b_g2 query's batch tables when it should differ from
                                        window_bits_b: a G2 addition costs three G1 additions, so b_g2's optimum is wider than b_g1's —
                                        at the price of a sort of its own (with equal widths b_g2 is reduced from b_g1's sorted digit
                                        list).  0 = the default (resolve_options, prover.hip); -1 = as window_bits_b */
} masp_hip_options;
void masp_hip_options_default(masp_hip_options* opt);
/* More hardware queues than this make a process SLOWER (round 6, profiles/r06_second_context_root_cause.txt): the runtime creates a
 * hardware queue for every new stream until GPU_MAX_HW_QUEUES exist and never gives one back, and once a process holds 24 / 32 of them
 * every kernel dispatch — also of a lone launch sequence on an idle chip — is 7 / 21 % slower (up to 20: nothing).  A context owns
 * 5 x slots + 1 streams (a slot's four side streams only work during lone proofs). */
#define MASP_HIP_MAX_USEFUL_HW_QUEUES 20
/* The HIP runtime gives a process four hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and reads the variable ONCE, at the
 * process's first HIP call.  Loading this library sets GPU_MAX_HW_QUEUES=16 if the variable is not set (a constructor: the only write to
 * the environment the library ever does, and it reads nothing else from it) — enough for a process whose first HIP call comes after the
 * library is loaded, i.e. a Rust binary linking it.  A process that initialises HIP earlier (another HIP library's static initialisers)
 * sets the variable itself, or calls this function before that point: hw_queues <= 0 means 16, more than
 * MASP_HIP_MAX_USEFUL_HW_QUEUES is cut to it; an existing value is kept unless `overwrite`.  Returns the value now in the environment.  What the runtime really uses is reported per context:
 * masp_hip_options::hw_queues. */
int masp_hip_runtime_prepare(int hw_queues, int overwrite);

int masp_hip_ctx_create(int device, masp_hip_ctx** out);
/* The general constructor: n_devices == 1 gives a single-device context, more give the multi-device front described
 * below; opt == NULL means defaults. */
int masp_hip_ctx_create_ex(const int* devices, int n_devices, const masp_hip_options* opt, masp_hip_ctx** out);
/* the options a context runs with, defaults resolved (ntt_sub_batch / window_bits_b2_lone: 0 here means "whole batch" / "not built") */
int masp_hip_ctx_get_options(const masp_hip_ctx* ctx, masp_hip_options* out);
/* One prover over several GPUs of a node (SURVEY.md §8b "devices, n_dev"): the serial per-description loops of
 * SaplingBuilder::build (/root/reference/masp_primitives/src/transaction/components/sapling/builder.rs:935-1140) become one
 * masp_hip_prove_batch call whose jobs are dealt to the devices in full batches, one host thread per device inside
 * the library — a Rust caller reaches all GPUs without any Python / torch.distributed.  masp_hip_circuit_load
 * replicates the CRS on every device (side by side).  A device may be listed more than once (two contexts on one
 * GPU).  The building-block and measurement entry points of such a context run on devices[0]. */
int masp_hip_ctx_create_multi(const int* devices, int n_devices, masp_hip_ctx** out);
/* number of device contexts behind `ctx` (1 for masp_hip_ctx_create) */
int masp_hip_ctx_device_count(const masp_hip_ctx* ctx);
/* counts[d] = proofs written so far by device context d (d < min(cap, masp_hip_ctx_device_count)): lets a caller (and the
 * configs[4] test) see that a multi-device prover really deals its batches to all of its devices */
int masp_hip_ctx_device_proofs(const masp_hip_ctx* ctx, uint64_t* counts, int cap);
/* A multi-device prover survives a device that fails: masp_hip_prove_batch deals its blocks (<= batch_cap jobs of one circuit, the most
 * expensive first) from ONE queue that every device's host thread takes from when it is free — no static shares —; when a device's call
 * returns MASP_HIP_E_HIP the device is taken out, its unfinished block goes back on the queue and the other devices finish the list
 * (the reference's per-description loop fails per description, not per transaction batch:
 * /root/reference/masp_primitives/src/transaction/components/sapling/builder.rs:955-969).  The call fails only when NO device is left
 * (or for an error of the input, which no other device would cure).  status[d] (d < min(cap, device count)) = MASP_HIP_OK or the
 * error code that took device context d out; a device that is out stays out for the life of the context, and masp_hip_last_error
 * keeps its text.  requeued (may be NULL): proofs that were put back on the queue so far.  A single-device context reports status[0]
 * = MASP_HIP_OK. */
int masp_hip_ctx_device_status(const masp_hip_ctx* ctx, int32_t* status, int cap, uint64_t* requeued);
/* TEST HOOK (tests/test_gpu_device_failure.py): the `nth` masp_hip_prove_batch call (1 = the next) that device context `device` of a
 * multi-device prover receives fails with MASP_HIP_E_HIP before it touches the device, as a lost GPU would.  nth = 0 disarms. */
int masp_hip_ctx_inject_fault(masp_hip_ctx* ctx, int device, uint32_t nth);
/* How many of THIS context's own streams (mains_only: only the ones that work side by side — the context's, the slots' main streams, the
 * verifier's two) run a kernel at the same time, measured now (the context must be idle).  *concurrent < *n_streams means two of them
 * share a hardware queue and run their work one after the other.  Which queue the runtime gives a stream depends on everything the PROCESS
 * created before, so a context creates all of its streams when it is created — 15 with the default four slots (slots 2 and up use slot 1's
 * side streams), one queue each of the default 16 —, measures the ones that work side by side and replaces any that share a queue:
 * a first, second and third context of a process prove at the same rate (profiles/r06_slot_streams_creation_order.txt,
 * profiles/r06_second_context_root_cause.txt). */
int masp_hip_ctx_stream_concurrency(masp_hip_ctx* ctx, int mains_only, int* n_streams, int* concurrent);
/* *out = calls of masp_hip_prove_batch groups so far that were replayed from a captured launch graph
 * (masp_hip_options::lone_proof_graph); a caller that proves one description at a time sees it grow from its third proof on */
int masp_hip_ctx_lone_graph_launches(const masp_hip_ctx* ctx, uint64_t* out);
void masp_hip_ctx_destroy(masp_hip_ctx* ctx);
const char* masp_hip_strerror(int code);
/* last HIP runtime error text seen by this context ("" if none); the pointer belongs to the calling thread and stays
 * valid until the same thread calls this function again */
const char* masp_hip_last_error(const masp_hip_ctx* ctx);

/* Parse `params` (bellman Parameters wire format; trailing bytes such as the MPC transcript are ignored),
 * check the length invariants against `cs`, upload the CRS and build the window tables. */
int masp_hip_circuit_load(masp_hip_ctx* ctx, uint32_t slot, const uint8_t* params, size_t params_len, const masp_hip_r1cs* cs);
/* What the loader found out about the circuit in `slot` (MASP_HIP_E_NOT_LOADED if empty).  *flags:
 *   bit 0  alpha_g1, beta_g1, delta_g1 and every a / b_g1 query point lie in the prime-order subgroup: the multiplications
 *          s*A and r*B1 of every proof go through the curve endomorphism (half the doublings).  Clear for a CRS with a curve
 *          point outside the subgroup — Parameters::read(_, false) does not look, lib.rs:343-347 — whose proofs then come from
 *          the plain double-and-add, with the bytes the reference computes. */
#define MASP_HIP_CIRCUIT_G1_ENDOMORPHISM 1u
int masp_hip_circuit_flags(const masp_hip_ctx* ctx, uint32_t slot, uint32_t* flags);

/* One proof, blocking.  proof_out: 192 bytes = A (48, G1 compressed) | B (96, G2 compressed) | C (48). */
int masp_hip_prove(masp_hip_ctx* ctx, uint32_t slot, const uint8_t* inputs, const uint8_t* aux, const uint8_t* a,
                   const uint8_t* b, const uint8_t* c, const uint8_t r[32], const uint8_t s[32], uint8_t proof_out[192]);
/* n independent jobs, results in job order; proofs_out: n x 192 bytes. */
int masp_hip_prove_batch(masp_hip_ctx* ctx, size_t n, const masp_hip_job* jobs, uint8_t* proofs_out);

/* Groth16 parameters for `cs` from explicit toxic waste (tau | alpha | beta | gamma | delta, 5 x 32 bytes LE),
 * written in the bellman Parameters wire format.  Mirrors bellperson `generate_random_parameters`, which the
 * reference's benches call (/root/reference/masp_proofs/benches/sapling.rs:24-36); needed because the real
 * MPC parameters cannot be downloaded here.  *out_len receives the size; if cap is too small nothing is
 * written and MASP_HIP_E_INVALID_ARG is returned (upper bound: masp_hip_parameters_max_size). */
int masp_hip_generate_parameters(masp_hip_ctx* ctx, const masp_hip_r1cs* cs, const uint8_t toxic[160], uint8_t* out, size_t cap,
                                 size_t* out_len);
size_t masp_hip_parameters_max_size(const masp_hip_r1cs* cs);

/* ---- building blocks (same kernels the prover uses) ---- */
/* sum_i scalars[i] * bases[i]; bases uncompressed (96 / 192 B each), result uncompressed */
int masp_hip_msm_g1(masp_hip_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[96]);
int masp_hip_msm_g2(masp_hip_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[192]);
/* np MSMs over ONE set of n G1 bases, launched the way a batch of proofs launches them (one kernel sequence,
 * gridDim.y = np): scalars np x n x 32, out np x 96.  window_bits: 0 = chosen from n, else 2..16 (the prover's buckets: 16 for
 * the h query and 12 for the witness queries).  Exists so that the batched code path can be checked on its own. */
int masp_hip_msm_g1_multi(masp_hip_ctx* ctx, const uint8_t* bases, size_t n, const uint8_t* scalars, size_t np, int window_bits,
                          uint8_t* out);
/* the same over G2 (bases n x 192, out np x 192) */
int masp_hip_msm_g2_multi(masp_hip_ctx* ctx, const uint8_t* bases, size_t n, const uint8_t* scalars, size_t np, int window_bits,
                          uint8_t* out);
/* h = ((A*B - C)/Z) coefficients from evaluation vectors a,b,c (nrows x 32 each, zero-padded to 2^logm);
 * h_out: (2^logm - 1) x 32 */
int masp_hip_quotient_h(masp_hip_ctx* ctx, const uint8_t* a, const uint8_t* b, const uint8_t* c, size_t nrows,
                        uint32_t logm, uint8_t* h_out);
/* in-place radix-2 NTT over Fr of 2^logm elements, natural order in and out */
int masp_hip_ntt(masp_hip_ctx* ctx, uint8_t* data, uint32_t logm, int inverse);

/* ---- Groth16 batch verification on the GPU ----
 * <- bellman `verify_proofs_batch` (/root/reference/masp_proofs/src/sapling/verifier/batch.rs:24-31,201-239) and, batched, the
 *    prover's own `verify_proof` self-checks (/root/reference/masp_proofs/src/sapling/prover.rs:148,266) with the
 *    `PreparedVerifyingKey` of /root/reference/masp_proofs/src/lib.rs:391-393.
 * masp_hip_vk_prepare: `params` = Parameters bytes (only the verifying-key prefix is read).
 * masp_hip_verify_batch: n proofs (192 B each) of ONE circuit, their public inputs (n x n_public x 32 B, excluding ONE) and
 * n x 16 B of caller-supplied randomness z (one random linear combination, as bellman does).  Proof decompression, the
 * z_i-multiples and the n Miller loops (one wavefront per pairing) run on the device; the public-input combination, two
 * pairings and the single final exponentiation on the host.  *all_valid = 1 iff every proof verifies (up to 2^-127);
 * 0 says at least one does not, not which.  Re-entrant: runs next to proving calls on one of the context's two verifier streams (one
 * verification at a time per key; keys beyond two share a stream).  A key must be freed before its context is destroyed. */
typedef struct masp_hip_vk masp_hip_vk;
int masp_hip_vk_prepare(masp_hip_ctx* ctx, const uint8_t* params, size_t params_len, masp_hip_vk** out);
void masp_hip_vk_free(masp_hip_vk* vk);
int masp_hip_verify_batch(masp_hip_ctx* ctx, masp_hip_vk* vk, size_t n, const uint8_t* proofs, const uint8_t* public_inputs,
                          uint32_t n_public, const uint8_t* z, int* all_valid);

/* ---- measurement hooks (bench.py): device-resident workloads, HIP-event timing on the ctx stream ---- */
/* Keeps `n` jobs' assignments resident in HBM; returns a handle (>= 0) or a negative error code. */
int masp_hip_batch_upload(masp_hip_ctx* ctx, size_t n, const masp_hip_job* jobs);
/* Proves every resident job of `handle` once; proofs_out n x 192.  elapsed_ms (may be NULL) = HIP-event time
 * from first kernel to last proof written, inputs already in HBM. */
int masp_hip_batch_prove_resident(masp_hip_ctx* ctx, int handle, uint8_t* proofs_out, float* elapsed_ms);
/* The same, `steps` times over: step k proves every resident job with the blinding scalars rs[(k * n + job) * 64 ...]
 * (r | s, 32 + 32 bytes; NULL = the uploaded ones), all steps enqueued back to back without host synchronisation — one
 * step = one pass of the hot path over one batch of n jobs.  proofs_out: steps x n x 192 bytes. */
int masp_hip_batch_prove_resident_steps(masp_hip_ctx* ctx, int handle, size_t steps, const uint8_t* rs, uint8_t* proofs_out,
                                        float* elapsed_ms);
int masp_hip_batch_free(masp_hip_ctx* ctx, int handle);
/* Runs the G1 MSM of query `which` (0 h, 1 l, 2 a, 3 b_g1) of circuit `slot` `iters` times on resident job
 * `job` of `handle`; returns average kernel-sequence time per MSM in ms and the number of bases. */
int masp_hip_bench_msm(masp_hip_ctx* ctx, int handle, size_t job, int which, int iters, float* avg_ms, uint32_t* n_bases);

/* Live timing of the dominant kernel, k_msm_accumulate over G1 (bucket accumulation): when enabled, every launch
 * is bracketed by HIP events on its own stream.  read: summed duration, launch count and the algorithmic bytes
 * (n x (96 + 32) per G1 MSM of n points) of all launches since enable/reset. */
int masp_hip_profile_enable(masp_hip_ctx* ctx, int on);
int masp_hip_profile_read(masp_hip_ctx* ctx, double* total_ms, uint64_t* launches, uint64_t* alg_bytes);
/* The same launches by kernel group, summed milliseconds since enable / reset: ms[0] plan / records / copies of the bucket tree,
 * ms[1] its denominators pass (k_tree_pass1), ms[2] the shared inversions (k_binv_*), ms[3] its additions pass (k_tree_pass2),
 * ms[4] the XYZZ accumulation of what is left (k_msm_accumulate_pts, or k_msm_accumulate without a tree); ms[5..7] = 0. */
int masp_hip_profile_read_split(masp_hip_ctx* ctx, double ms[8]);
/* Where the chains of the last LONE proof (a batch of fewer than 8 proofs) proved with profiling on ended, in milliseconds of GPU time
 * after its first kernel could start — HIP events behind the stages, no profiler in the way: ms[1] MSM a, [2] s*A, [3] MSM b_g1, [4] r*B1,
 * [5] MSM b_g2, [6] g_b written, [7] quotient, [8] MSM h, [9] MSM l, [10] g_a / g_c written, [11] the proof complete; -1 = not recorded. */
int masp_hip_profile_read_lone(masp_hip_ctx* ctx, double ms[12]);
/* Page-locked host memory for assignments.  masp_hip_prove_batch recognises `aux` pointers that lie in page-locked
 * memory (from here or from the caller's own hipHostMalloc / hipHostRegister) and copies them to the device directly;
 * anything else goes through the library's own pinned staging buffer first (one extra host copy of ~3 MB per Spend).
 * The reference keeps the assignment in ordinary Vec<Scalar>s inside bellperson's ProvingAssignment; a binding that
 * synthesizes into a buffer obtained here saves that copy.  NULL on failure. */
void* masp_hip_host_alloc(masp_hip_ctx* ctx, size_t bytes);
void masp_hip_host_free(masp_hip_ctx* ctx, void* ptr);
/* hipDeviceSynchronize on the context's device */
int masp_hip_sync(masp_hip_ctx* ctx);

#ifdef __cplusplus
}
#endif

/* ---- the layout a binding relies on, checked wherever this header is compiled (LP64: the only ABI the library is built for).  A Rust
 * `#[repr(C)]` mirror must give the same numbers; tests/native/capi_harness.c repeats them with offsetof ---- */
#if defined(__cplusplus) && __cplusplus >= 201103L
#define MASP_HIP_STATIC_ASSERT(c, m) static_assert(c, m)
#elif defined(__STDC_VERSION__) && __STDC_VERSION__ >= 201112L
#define MASP_HIP_STATIC_ASSERT(c, m) _Static_assert(c, m)
#else
#define MASP_HIP_SA_CAT2(a, b) a##b
#define MASP_HIP_SA_CAT(a, b) MASP_HIP_SA_CAT2(a, b)
#define MASP_HIP_STATIC_ASSERT(c, m) typedef char MASP_HIP_SA_CAT(masp_hip_static_assert_, __LINE__)[(c) ? 1 : -1]
#endif
#if defined(__LP64__) || defined(_LP64)
MASP_HIP_STATIC_ASSERT(sizeof(masp_hip_job) == 120, "masp_hip_job: 4 + 4 pad + 5 pointers + r[32] + s[32] + aux_form + reserved");
MASP_HIP_STATIC_ASSERT(offsetof(masp_hip_job, inputs) == 8 && offsetof(masp_hip_job, c) == 40, "masp_hip_job pointers");
MASP_HIP_STATIC_ASSERT(offsetof(masp_hip_job, r) == 48 && offsetof(masp_hip_job, s) == 80, "masp_hip_job r | s");
MASP_HIP_STATIC_ASSERT(offsetof(masp_hip_job, aux_form) == 112 && offsetof(masp_hip_job, reserved) == 116, "masp_hip_job tail");
MASP_HIP_STATIC_ASSERT(sizeof(masp_hip_r1cs) == 88 && offsetof(masp_hip_r1cs, a_rowptr) == 16 && offsetof(masp_hip_r1cs, c_coef) == 80,
                       "masp_hip_r1cs: three counts + 4 pad + nine pointers");
MASP_HIP_STATIC_ASSERT(sizeof(masp_hip_options) == 76 && offsetof(masp_hip_options, window_bits_b2) == 72,
                       "masp_hip_options: struct_size + 18 int32 fields (appended-only: struct_size versions it)");
MASP_HIP_STATIC_ASSERT(offsetof(masp_hip_options, hw_queues) == 60 && offsetof(masp_hip_options, digit_recoding) == 68, "masp_hip_options tail");
#endif
#endif

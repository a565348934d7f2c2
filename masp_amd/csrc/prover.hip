// libmasp_hip: context, CRS loading, the per-proof kernel pipeline and the C ABI (include/masp_hip.h).
//
// Mirrors, on the GPU, what `create_random_proof(circuit, &Parameters, rng)` does after synthesis
// (/root/reference/masp_proofs/src/sapling/prover.rs:117,202,252 -> nam-bellperson):
//   witness -> a,b,c (static R1CS SpMV) -> quotient h (7 NTTs) -> MSMs H, L, A, B1 (G1), B2 (G2)
//   -> assembly with r, s -> 192-byte proof.
// There is deliberately NO CPU fallback: without a usable HIP device every entry point fails with
// MASP_HIP_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"

using namespace masp;

// host-to-device copies of concurrent calls one batch after the other (masp_hip_ctx::upload_tail); 0: as they come (A/B builds)
#ifndef MASP_STREAM_ORDER
#define MASP_STREAM_ORDER 2   // slot by slot (see create_single); separate_main_streams then makes sure the main streams do not share queues
#endif
#ifndef MASP_UPLOAD_CHAIN
#define MASP_UPLOAD_CHAIN 1
#endif

namespace {

// options of a context (include/masp_hip.h: masp_hip_options) with every "0 = default" resolved; no environment is read
static masp_hip_options resolve_options(const masp_hip_options* in) {
    masp_hip_options o;
    memset(&o, 0, sizeof o);
    if (in) memcpy(&o, in, std::min<size_t>(in->struct_size ? in->struct_size : sizeof o, sizeof o));
    o.struct_size = sizeof o;
    // (round 4: with the passes of the bucket tree no longer waiting on memory, three batches in flight keep the chip as busy as
    // four did — 1 227 vs 1 220 proofs/s — and leave their scratch to larger tree sub-batches: 3 x 86 proofs, 1 251 proofs/s, in the
    // ~100 GB that 4 x 64 took; profiles/r04_same_box_sweep_slots_tree_sub.txt)
    // (round 6: FOUR — +1.5 ... 1.7 % host to host over three at the default 16 hardware queues: a batch uploads while three compute;
    // five are not faster, and the measurements of round 5 that made four look no better were taken at 24 queues, in the runtime's
    // oversubscribed regime: profiles/r06_slots_3_4_5_at_16_hardware_queues.txt)
    o.slots = o.slots > 0 ? std::min<int>(o.slots, (int)masp_hip_ctx::MAX_SLOTS) : 4;
    // proofs per batched launch sequence (upper bound: a list of n same-circuit jobs is cut into ceil(n / cap) equal groups);
    // scratch memory follows the batches actually formed
    o.batch_cap = o.batch_cap > 0 ? std::min<int>(o.batch_cap, 256) : 256;
    o.ntt_sub_batch = o.ntt_sub_batch > 0 ? o.ntt_sub_batch : o.ntt_sub_batch < 0 ? 0 : 8;   // resolved: 0 = whole batch
    o.window_bits_b2_lone = o.window_bits_b2_lone > 0 ? o.window_bits_b2_lone : o.window_bits_b2_lone < 0 ? 0 : 8;  // resolved: 0 = not built
    // ~30 % of a MASP witness is neither 0 nor 1; measured +3..5 % throughput vs 100
    o.witness_nontrivial_percent = o.witness_nontrivial_percent > 0 ? std::min<int>(o.witness_nontrivial_percent, 100) : 30;
    o.bucket_tree_levels = o.bucket_tree_levels > 0 ? std::min<int>(o.bucket_tree_levels, 12) : o.bucket_tree_levels < 0 ? -1 : 0;  // resolved: 0 = automatic
    o.bucket_tree_sub_batch = o.bucket_tree_sub_batch > 0 ? std::min<int>(o.bucket_tree_sub_batch, 256) : 86;   // a 256-proof batch in three sub-batches
    o.bucket_tree_levels_g2 = o.bucket_tree_levels_g2 > 0 ? std::min<int>(o.bucket_tree_levels_g2, 12) : o.bucket_tree_levels_g2 < 0 ? -1 : o.bucket_tree_levels;
    o.bucket_tree_scratch_mb = std::max(o.bucket_tree_scratch_mb, 0);
    o.bucket_tree_fallback_proofs = 0;   // output only
    o.hw_queues = 0;                     // output only (measured when a device context is created)
    o.window_bits_h_lone = o.window_bits_h_lone > 0 ? std::min<int>(o.window_bits_h_lone, 16) : 0;   // resolved: 0 = as window_bits_h
    o.lone_proof_graph = o.lone_proof_graph > 0 ? 1 : 0;   // off unless asked for: measured slower with ROCm 7.2's graph launch
    o.digit_recoding = 0;   // (round 5's NAF digits over per-bit tables were removed in round 6: the field is reserved, must be 0 on input)
    // b_g2's batch tables on a window width of their own (2..16); 0 / -1 resolved where the circuit is loaded (masp_hip_circuit_load)
    o.window_bits_b2 = o.window_bits_b2 > 0 ? std::max(2, std::min<int>(o.window_bits_b2, 16)) : o.window_bits_b2 < 0 ? -1 : 0;
    return o;
}
// [0, n) cut into ceil(n / cap) groups whose sizes differ by at most one: (first, count) pairs
static std::vector<std::pair<size_t, size_t>> even_groups(size_t n, size_t cap) {
    std::vector<std::pair<size_t, size_t>> out;
    if (!n) return out;
    const size_t g = (n + cap - 1) / cap, base = n / g, extra = n % g;
    size_t lo = 0;
    for (size_t i = 0; i < g; ++i) {
        size_t cnt = base + (i < extra ? 1 : 0);
        out.push_back({lo, cnt});
        lo += cnt;
    }
    return out;
}

// slots 0 and 1 — and every slot when lone proofs are replayed from captured graphs — have side streams of their own (masp_hip_ctx::slot_streams)
static bool slot_owns_side_streams(const masp_hip_ctx* ctx, size_t si) { return si < 2 || ctx->opt.lone_proof_graph > 0; }
// the five streams the context created for its next slot (slot_mu held), or nullptr if it has none left
static const hipStream_t* take_slot_streams(masp_hip_ctx* ctx, bool* side_streams_are_its_own) {
    const size_t i = ctx->slots.size();
    *side_streams_are_its_own = true;
    if (i >= ctx->slot_streams.size() || ctx->slot_streams_taken[i]) return nullptr;
    ctx->slot_streams_taken[i] = 1;
    *side_streams_are_its_own = slot_owns_side_streams(ctx, i);
    return ctx->slot_streams[i].data();
}
static int ensure_slots(masp_hip_ctx* ctx, size_t want) {
    std::lock_guard<std::mutex> g(ctx->slot_mu);
    want = std::min(want, masp_hip_ctx::MAX_SLOTS);
    while (ctx->slots.size() < want) {
        std::unique_ptr<Slot> s(new Slot);
        bool own = true;
        const hipStream_t* st = take_slot_streams(ctx, &own);
        int rc = s->init(st, own);
        if (rc) return rc;
        s->configure(ctx->opt);
        ctx->slots.push_back(std::move(s));
        ctx->slot_busy.push_back(0);
    }
    // (a slot is configured once, when it is created: its options never change, and a busy slot's workspaces are read by the
    // thread that owns it — the tree sub-batch may also have been halved there for lack of scratch)
    for (auto& sl : ctx->slots) sl->profiling = ctx->profiling;
    return MASP_HIP_OK;
}
// Slot pool for concurrent provers.  try: a free slot, or a new one while fewer than ctx->n_slots exist.
// Returns MASP_HIP_OK and *si, or -1 if every slot is busy, or an error code.
static int slot_try_acquire(masp_hip_ctx* ctx, size_t* si) {
    std::lock_guard<std::mutex> g(ctx->slot_mu);
    for (size_t i = 0; i < ctx->slots.size(); ++i)
        if (!ctx->slot_busy[i]) {
            ctx->slot_busy[i] = 1;
            *si = i;
            return MASP_HIP_OK;
        }
    if (ctx->slots.size() >= (size_t)ctx->n_slots) return -1;
    std::unique_ptr<Slot> s(new Slot);
    bool own = true;
    const hipStream_t* st = take_slot_streams(ctx, &own);
    int rc = s->init(st, own);
    if (rc) return rc;
    s->profiling = ctx->profiling;
    s->configure(ctx->opt);
    ctx->slots.push_back(std::move(s));
    ctx->slot_busy.push_back(1);
    *si = ctx->slots.size() - 1;
    return MASP_HIP_OK;
}
static size_t slot_acquire_blocking(masp_hip_ctx* ctx) {
    std::unique_lock<std::mutex> g(ctx->slot_mu);
    size_t found = 0;
    ctx->slot_cv.wait(g, [&] {
        for (size_t i = 0; i < ctx->slots.size(); ++i)
            if (!ctx->slot_busy[i]) {
                found = i;
                return true;
            }
        return false;
    });
    ctx->slot_busy[found] = 1;
    return found;
}
static void slot_release(masp_hip_ctx* ctx, size_t si) {
    bool idle = true;
    {
        std::lock_guard<std::mutex> g(ctx->slot_mu);
        ctx->slot_busy[si] = 0;
        for (char b : ctx->slot_busy) idle = idle && !b;
    }
    ctx->slot_cv.notify_one();
    // nothing of this context is in flight: the moment to give back what its workspaces outgrew (util.h, dev_free)
    if (idle) dev_free_drain();
}
static int fail_shared(masp_hip_ctx* ctx, int rc) {
    if (rc == MASP_HIP_E_HIP) {
        std::lock_guard<std::mutex> g(ctx->slot_mu);
        ctx->err = last_hip_error();
    }
    return rc;
}

// Quotient for np proofs on slot buffers: in[i] + p * in_stride are Montgomery (mont_in) or canonical evaluation
// vectors of `nrows` entries on the device; result: sl.h + p * m = canonical coefficients h[0..m-1).
// h_out (stride h_stride elements per proof; NULL: sl.h, stride m).
static int enqueue_quotient(Slot& sl, const NttDomain& D, const Fr* const in[3], size_t in_stride, uint32_t nrows, bool mont_in, uint32_t np,
                            Fr* h_out = nullptr, size_t h_stride = 0) {
    hipStream_t s = sl.stream;
    const uint32_t m = (uint32_t)D.m, logm = D.logm;
    // A batch goes through the six transforms in sub-batches whose six work buffers (sub x 6 x 32 m bytes: 192 MiB for
    // eight Spend proofs) stay in the 256 MiB Infinity Cache from pass to pass, instead of every pass streaming the whole
    // batch (np x 4 MiB per buffer) through HBM.  The work buffers are only sub proofs long; h is the per-batch result.
    const uint32_t sub_max = sl.ntt_sub;  // masp_hip_options::ntt_sub_batch (0 = the whole batch at once)
    const uint32_t sub = sub_max ? std::min(sub_max, np) : np;
    int rc;
    // a, b and c go through every pass TOGETHER: the three transforms of a sub-batch lie back to back in one buffer
    // ([a | b | c] x q proofs) and a pass is one launch over 3 q transforms — three times the workgroups per launch
    // (a lone proof: 384 instead of 128 on 256 CUs) and 12 instead of 22 launches per sub-batch
    if ((rc = sl.x0.reserve((size_t)3 * m * sub)) || (rc = sl.x1.reserve((size_t)2 * m * sub))) return rc;
    if (!h_out) {
        if ((rc = sl.h.reserve((size_t)m * np))) return rc;
        h_out = sl.h.p;
        h_stride = m;
    }
    Fr *x0 = sl.x0.p, *x1 = sl.x1.p;
    for (uint32_t p0 = 0; p0 < np; p0 += sub) {
        const uint32_t q = std::min(sub, np - p0);
        const size_t part = (size_t)q * m;  // one of a / b / c for the whole sub-batch
        for (int i = 0; i < 3; ++i) {
            const Fr* src = in[i] + (size_t)p0 * in_stride;
            if (mont_in)
                launch_ntt_copy_bitrev(s, src, in_stride, nrows, x0 + i * part, logm, q);
            else
                launch_ntt_load_bitrev(s, src, in_stride, nrows, x0 + i * part, logm, q);
        }
        D.passes(s, x0, D.tw_inv.p, 3 * q);                                     // m A, m B, m C (coefficients)
        launch_ntt_scale_bitrev(s, x0, D.coset_scale.p, x1, logm, 2 * q);       // A, B: * g^k / m
        D.passes(s, x1, D.tw_fwd.p, 2 * q);                                     // A, B on the coset g H
        launch_ntt_ab_bitrev(s, x1, x1 + part, x0, logm, q);
        D.passes(s, x0, D.tw_inv.p, q);                                         // m g^k ((g^m - 1) h + C)_k
        // h = (that * g^-k / m - C) / (g^m - 1); leaves Montgomery form
        launch_fr_scale_sub(s, x0, D.h_scale.p, x0 + 2 * part, D.c_scale, h_out + (size_t)p0 * h_stride, m, q, h_stride);
    }
    return MASP_HIP_OK;
}

// Enqueue np proofs of circuit C as one batch.  d_w + p * w_stride: n_vars canonical scalars on the device (inputs then
// aux).  d_abc[i] + p * nrows: canonical evaluation vectors on the device, or all NULL.  d_rs + p * 16: r | s limbs.
// d_proof + p * 192: output.
// aux_montgomery: the aux part of every assignment (d_w + p * w_stride + n_inputs ...) holds Montgomery residues
// (masp_hip_job::aux_form); it is converted to canonical form in place first (d_w must then be writable).
static int enqueue_proofs(Slot& sl, Circuit& C, uint32_t np, const Fr* d_w, size_t w_stride, const Fr* const d_abc[3], const uint32_t* d_rs,
                          uint8_t* d_proof, bool aux_montgomery = false) {
    hipStream_t s = sl.stream;
    const uint32_t nv = C.n_inputs + C.n_aux;
    int rc;
    if ((rc = sl.reserve_batch(np))) return rc;
    // (sl.flags accumulates: the caller clears it before the first batch it will read the flag for)
    if ((rc = sl.wm.reserve((size_t)nv * np))) return rc;
    const bool lone = np < 8;
    // range check (+ Montgomery copy used by the SpMV); an aux part that arrived as Montgomery residues becomes canonical here,
    // BEFORE anything reads it as scalars
    if (aux_montgomery) launch_fr_split_forms(s, const_cast<Fr*>(d_w), w_stride, sl.wm.p, nv, C.n_inputs, np, sl.flags.p);
    if (lone) {
        // lone-proof mode: L, A, B1 and B2 only read the assignment (already on the device when this stream gets here), so
        // their streams fork NOW, before the range check and the SpMV are queued on the main stream
        sl.lone_mark(s, 0);
        HIP_TRY(hipEventRecord(sl.ev_fork, s));
        for (int i = 0; i < Slot::N_AUX; ++i) HIP_TRY(hipStreamWaitEvent(sl.aux[i], sl.ev_fork, 0));
    }
    if (!aux_montgomery) launch_fr_to_mont(s, d_w, w_stride, sl.wm.p, nv, np, sl.flags.p);
    const Fr* in[3];
    bool mont_in;
    if (d_abc[0]) {
        in[0] = d_abc[0];
        in[1] = d_abc[1];
        in[2] = d_abc[2];
        mont_in = false;
    } else {
        R1csMatrices M;
        for (int i = 0; i < 3; ++i) {
            if ((rc = sl.ev[i].reserve((size_t)C.nrows * np))) return rc;
            M.rowptr[i] = C.rowptr[i].p;
            M.order[i] = C.row_order[i].p;
            M.n_long[i] = C.n_long_rows[i];
            M.col[i] = C.col[i].p;
            M.coef[i] = C.coef[i].p;
            M.out[i] = sl.ev[i].p;
            in[i] = sl.ev[i].p;
        }
        launch_r1cs_eval(s, M, sl.wm.p, nv, C.n_constraints, C.n_inputs, np);  // a, b, c = A w, B w, C w in one launch
        mont_in = true;
    }
    if ((rc = sl.sa.reserve((size_t)C.na * np)) || (rc = sl.sb.reserve((size_t)C.nbq * np))) return rc;
    MsmProfile* prof = sl.profiling ? &sl.prof : nullptr;
    const size_t m8 = C.m * 8;
    const bool share_b = C.nbq && C.b2.n == C.b1.n && C.b2.g.c == C.b1.g.c;  // B2 runs over the same scalars as B1
    if (lone) {
        // the four witness MSMs run on their own streams (forked above) while the main stream runs SpMV -> quotient -> H.
        // The pieces of the assembly start as soon as what they read exists: the fixed-base multiplications (r and s only)
        // right away, s*A and r*B1 behind their MSMs, g_b behind B2 — after the join only g_a / g_c are left.
        launch_groth16_fixed_g1(sl.aux[0], C.fb1.p, d_rs, 16, sl.asm1.p, np);
        launch_groth16_fixed_g2(sl.aux[0], C.fb2.p, d_rs, 16, sl.asm2.p, np);
        HIP_TRY(hipEventRecord(sl.ev_fixed, sl.aux[0]));  // s*delta2 is read by g_b on aux[3]
        // Enqueueing a lone proof takes the host ~3 ms (~250 launches) — as long as its longest chain runs, so whatever is
        // enqueued last starts 3 ms late: the chains go out longest first — A and B1 + B2 (an MSM followed by 2.3 ms of
        // variable-base multiplication; B2's G2 tails), then the quotient and H (the SpMV in front of them is 25 us), then L,
        // which only reads the witness and is the shortest.  (Helper threads enqueueing the chains side by side, and high-
        // priority streams for A / B1, were measured: every chain then starts within 0.5 ms, they slow each other down and the
        // proof takes the same 6.1 - 6.4 ms: profiles/r03x_same_box_ab_lone_threads.txt.)
        const bool own_b2 = C.b2_lone.n != 0;  // B2 on its own narrow windows: sorts for itself
        auto chain_a = [&]() -> int {
            int r;
            if (C.na) launch_gather_scalars(sl.aux[1], d_w, w_stride, C.a_var.p, C.na, sl.sa.p, np);
            if ((r = msm_enqueue(sl.aux[1], C.a, sl.ws_a, (const uint32_t*)sl.sa.p, (size_t)C.na * 8, sl.res1.p + 2, 4, np))) return r;
            sl.lone_mark(sl.aux[1], 1);
            launch_groth16_var_mul(sl.aux[1], 0, sl.res1.p, d_rs, 16, sl.asm1.p, np, C.g1_endo);
            sl.lone_mark(sl.aux[1], 2);
            return MASP_HIP_OK;
        };
        auto chain_b = [&]() -> int {
            int r;
            if (C.nbq) launch_gather_scalars(sl.aux[2], d_w, w_stride, C.b_var.p, C.nbq, sl.sb.p, np);
            HIP_TRY(hipEventRecord(sl.ev_sort_b, sl.aux[2]));  // B2 on aux[3] reads sb (and, without tables of its own, B1's sort) behind this
            if (share_b) {
                if ((r = msm_sort_enqueue(sl.aux[2], C.b1.n, C.b1.g, sl.ws_b.sort, (const uint32_t*)sl.sb.p, (size_t)C.nbq * 8, np))) return r;
                if (!own_b2) HIP_TRY(hipEventRecord(sl.ev_sort_b, sl.aux[2]));
                if ((r = msm_reduce_enqueue(sl.aux[2], C.b1, sl.ws_b.sort, sl.ws_b, sl.res1.p + 3, 4))) return r;
            } else if ((r = msm_enqueue(sl.aux[2], C.b1, sl.ws_b, (const uint32_t*)sl.sb.p, (size_t)C.nbq * 8, sl.res1.p + 3, 4, np))) {
                return r;
            }
            sl.lone_mark(sl.aux[2], 3);
            launch_groth16_var_mul(sl.aux[2], 1, sl.res1.p, d_rs, 16, sl.asm1.p, np, C.g1_endo);
            sl.lone_mark(sl.aux[2], 4);
            HIP_TRY(hipStreamWaitEvent(sl.aux[3], sl.ev_sort_b, 0));
            if (own_b2) {
                if ((r = msm_enqueue(sl.aux[3], C.b2_lone, sl.ws2, (const uint32_t*)sl.sb.p, (size_t)C.nbq * 8, sl.res2.p, 1, np))) return r;
            } else if (share_b) {
                if ((r = msm_reduce_enqueue(sl.aux[3], C.b2, sl.ws_b.sort, sl.ws2, sl.res2.p, 1))) return r;
            } else if ((r = msm_enqueue(sl.aux[3], C.b2, sl.ws2, (const uint32_t*)sl.sb.p, (size_t)C.nbq * 8, sl.res2.p, 1, np))) {
                return r;
            }
            sl.lone_mark(sl.aux[3], 5);
            HIP_TRY(hipStreamWaitEvent(sl.aux[3], sl.ev_fixed, 0));
            launch_groth16_finish_b(sl.aux[3], C.vk.p, sl.asm2.p, sl.res2.p, d_proof, np);
            sl.lone_mark(sl.aux[3], 6);
            return MASP_HIP_OK;
        };
        auto chain_l = [&]() -> int {
            return msm_enqueue(sl.aux[0], C.l, sl.ws_l, (const uint32_t*)(d_w + C.n_inputs), w_stride * 8, sl.res1.p + 1, 4, np);
        };
        // (measured and dropped in round 5: the quotient first with the side chains' chip-filling kernels gated behind it — the quotient
        // then ends at 0.8 instead of 1.3 ms and MSM h, which now shares the chip with four accumulations, 0.3 ms LATER: a lone proof is
        // ~2.5 ms of chip-filling work, its chains can only trade places: profiles/r05_lone_proof_chains.txt)
        if ((rc = chain_a()) || (rc = chain_b())) return rc;
        if ((rc = enqueue_quotient(sl, *C.dom, in, C.nrows, C.nrows, mont_in, np))) return rc;
        sl.lone_mark(s, 7);
        if ((rc = msm_enqueue(s, C.h, sl.ws1, (const uint32_t*)sl.h.p, m8, sl.res1.p + 0, 4, np, prof))) return rc;
        sl.lone_mark(s, 8);
        if ((rc = chain_l())) return rc;
        sl.lone_mark(sl.aux[0], 9);
        // g_a and all of g_c but H need L, s*A and r*B1 (aux[0..2]): summed on aux[0] behind those chains, while H is still being
        // computed (k_groth16_finish_ac_early); the main stream joins aux[0] and adds H.  g_b finishes on aux[3] by itself and is only
        // joined before the proof leaves
        for (int i = 1; i < Slot::N_AUX - 1; ++i) {
            HIP_TRY(hipEventRecord(sl.ev_join[i], sl.aux[i]));
            HIP_TRY(hipStreamWaitEvent(sl.aux[0], sl.ev_join[i], 0));
        }
        launch_groth16_finish_ac_early(sl.aux[0], C.vk.p, sl.asm1.p, sl.res1.p, d_proof, np);
        HIP_TRY(hipEventRecord(sl.ev_join[0], sl.aux[0]));
        HIP_TRY(hipStreamWaitEvent(s, sl.ev_join[0], 0));
    } else {
        // H and L as one MSM over the merged base set (Circuit::hl): the scalars of a proof are its m - 1 quotient coefficients
        // followed by its aux assignment (the quotient's m-th, unused, coefficient lands on the first aux slot and is then
        // overwritten by the copy)
        const size_t hl_stride = C.hl.n;
        if ((rc = sl.hl.reserve(hl_stride * np + 1))) return rc;
        if ((rc = enqueue_quotient(sl, *C.dom, in, C.nrows, C.nrows, mont_in, np, sl.hl.p, hl_stride))) return rc;
        HIP_TRY(hipMemcpy2DAsync(sl.hl.p + (C.m - 1), hl_stride * sizeof(Fr), d_w + C.n_inputs, w_stride * sizeof(Fr), (size_t)C.n_aux * sizeof(Fr), np,
                                 hipMemcpyDeviceToDevice, s));
        // query scalars selected by density
        if (C.na) launch_gather_scalars(s, d_w, w_stride, C.a_var.p, C.na, sl.sa.p, np);
        if (C.nbq) launch_gather_scalars(s, d_w, w_stride, C.b_var.p, C.nbq, sl.sb.p, np);
        if ((rc = msm_enqueue(s, C.hl, sl.ws1, (const uint32_t*)sl.hl.p, hl_stride * 8, sl.res1.p + 0, 4, np, prof))) return rc;
        // L is inside H + L: its slot of every proof is the point at infinity (one strided fill, not np of them)
        HIP_TRY(hipMemset2DAsync(sl.res1.p + 1, 4 * sizeof(G1Xyzz), 0, sizeof(G1Xyzz), np, s));
        // (b_g2 on b_g1's window width is reduced from b_g1's sorted digit list; on a width of its own — masp_hip_options::window_bits_b2 — it sorts for itself)
        const BasesG1 &Aq = C.a, &B1q = C.b1;
        const BasesG2& B2q = C.b2;
        const bool share_bq = share_b;
        if ((rc = msm_enqueue(s, Aq, sl.ws1, (const uint32_t*)sl.sa.p, (size_t)C.na * 8, sl.res1.p + 2, 4, np, prof))) return rc;
        if ((rc = msm_enqueue(s, B1q, sl.ws1, (const uint32_t*)sl.sb.p, (size_t)C.nbq * 8, sl.res1.p + 3, 4, np, prof))) return rc;
        if (share_bq) {
            if ((rc = msm_reduce_enqueue(s, B2q, sl.ws1.sort, sl.ws2, sl.res2.p, 1))) return rc;
        } else if ((rc = msm_enqueue(s, B2q, sl.ws2, (const uint32_t*)sl.sb.p, (size_t)C.nbq * 8, sl.res2.p, 1, np))) {
            return rc;
        }
    }
    if (!lone) {
        launch_groth16_fixed_g1(s, C.fb1.p, d_rs, 16, sl.asm1.p, np);
        launch_groth16_fixed_g2(s, C.fb2.p, d_rs, 16, sl.asm2.p, np);
                launch_groth16_var_mul(s, 2, sl.res1.p, d_rs, 16, sl.asm1.p, np, C.g1_endo);
        launch_groth16_finish_b(s, C.vk.p, sl.asm2.p, sl.res2.p, d_proof, np);
    }
    if (lone)
        launch_groth16_finish_c_late(s, sl.asm1.p, sl.res1.p, d_proof, np);
    else
        launch_groth16_finish_ac(s, C.vk.p, sl.asm1.p, sl.res1.p, d_proof, np);
    if (lone) {
        sl.lone_mark(s, 10);   // g_a / g_c written
        HIP_TRY(hipEventRecord(sl.ev_join[Slot::N_AUX - 1], sl.aux[Slot::N_AUX - 1]));
        HIP_TRY(hipStreamWaitEvent(s, sl.ev_join[Slot::N_AUX - 1], 0));
        sl.lone_mark(s, 11);   // ... and g_b: the proof is complete
    }
    return MASP_HIP_OK;
}

// enqueue_proofs for a batch of fewer than 8 proofs, replayed from a captured HIP graph once the same call has been seen
// twice.  A lone Spend proof is ~250 launches on five streams: the host needs ~3 ms to enqueue what the GPU runs in 5 - 6 ms,
// and the chain enqueued last starts that much late (profiles/r04z_lone_proof_timeline.txt); one hipGraphLaunch of the same
// DAG does not.  The key is everything the launches' arguments are derived from: the circuit, the proof count, the slot's
// buffers the caller passes and the form of the aux part; the workspaces inside (grown on demand) are covered by
// device_alloc_epoch().  First call with a key: plain enqueue (sizes every buffer).  Second: captured (nothing runs), then
// launched.  Any failure on the way — an allocation inside the capture, a runtime that refuses a node — marks the key dead
// and the call is enqueued the plain way: never an error of its own.
static int enqueue_proofs_graphed(Slot& sl, Circuit& C, uint32_t np, const Fr* d_w, size_t w_stride, const Fr* const d_abc[3], const uint32_t* d_rs,
                                  uint8_t* d_proof, bool aux_montgomery) {
    if (np >= 8 || !sl.lone_graph || sl.profiling) return enqueue_proofs(sl, C, np, d_w, w_stride, d_abc, d_rs, d_proof, aux_montgomery);
    const uint64_t epoch = device_alloc_epoch().load(std::memory_order_relaxed);
    if (epoch != sl.graph_epoch) {
        sl.drop_graphs();
        sl.graph_epoch = epoch;
    }
    Slot::LoneGraph* e = nullptr;
    for (auto& g : sl.graphs)
        if (g.circuit == &C && g.np == np && g.w == d_w && g.w_stride == w_stride && g.abc == d_abc[0] && g.rs == d_rs && g.proof == d_proof &&
            g.mont == aux_montgomery)
            e = &g;
    if (!e) {
        if (sl.graphs.size() >= 16) sl.drop_graphs();
        sl.graphs.push_back(Slot::LoneGraph{&C, d_w, d_abc[0], d_rs, d_proof, w_stride, np, aux_montgomery, 0, false, nullptr});
        e = &sl.graphs.back();
    }
    if (e->exec) {
        HIP_TRY(hipGraphLaunch(e->exec, sl.stream));
        ++sl.graph_launches;
        return MASP_HIP_OK;
    }
    if (e->dead || e->runs == 0) {
        ++e->runs;
        return enqueue_proofs(sl, C, np, d_w, w_stride, d_abc, d_rs, d_proof, aux_montgomery);
    }
    bool ok = hipStreamBeginCapture(sl.stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        const int rc = enqueue_proofs(sl, C, np, d_w, w_stride, d_abc, d_rs, d_proof, aux_montgomery);
        hipGraph_t g = nullptr;
        const hipError_t ce = hipStreamEndCapture(sl.stream, &g);
        ok = rc == MASP_HIP_OK && ce == hipSuccess && g != nullptr && device_alloc_epoch().load(std::memory_order_relaxed) == epoch;
        if (ok) ok = hipGraphInstantiate(&e->exec, g, nullptr, nullptr, 0) == hipSuccess && e->exec != nullptr;
        if (g) hipGraphDestroy(g);
    }
    if (!ok) {
        (void)hipGetLastError();
        launch_error().clear();
        e->exec = nullptr;
        e->dead = true;
        sl.graph_epoch = device_alloc_epoch().load(std::memory_order_relaxed);
        return enqueue_proofs(sl, C, np, d_w, w_stride, d_abc, d_rs, d_proof, aux_montgomery);
    }
    HIP_TRY(hipGraphLaunch(e->exec, sl.stream));
    ++sl.graph_launches;
    return MASP_HIP_OK;
}

static bool rs_in_range(const uint8_t* x) {
    Fr v = fe_load_le<FrCfg>(x);
    return !fe_canonical_ge_mod(v);
}

struct ParamsLayout {
    const uint8_t *alpha_g1, *beta_g1, *beta_g2, *gamma_g2, *delta_g1, *delta_g2;
    const uint8_t *ic, *h, *l, *a, *b_g1, *b_g2;
    uint32_t n_ic, n_h, n_l, n_a, n_b1, n_b2;
};
static bool read_u32be(const uint8_t*& p, const uint8_t* end, uint32_t& v) {
    if (end - p < 4) return false;
    v = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    p += 4;
    return true;
}
// bellman Parameters wire format; trailing bytes are ignored like lib.rs:343-347 does
static bool parse_params(const uint8_t* buf, size_t len, ParamsLayout& L) {
    const uint8_t *p = buf, *end = buf + len;
    if (len < 864) return false;
    L.alpha_g1 = p; p += 96;
    L.beta_g1 = p;  p += 96;
    L.beta_g2 = p;  p += 192;
    L.gamma_g2 = p; p += 192;
    L.delta_g1 = p; p += 96;
    L.delta_g2 = p; p += 192;
    struct { const uint8_t** ptr; uint32_t* n; size_t sz; } secs[6] = {
        {&L.ic, &L.n_ic, 96}, {&L.h, &L.n_h, 96}, {&L.l, &L.n_l, 96}, {&L.a, &L.n_a, 96}, {&L.b_g1, &L.n_b1, 96}, {&L.b_g2, &L.n_b2, 192}};
    for (auto& sec : secs) {
        if (!read_u32be(p, end, *sec.n)) return false;
        if ((size_t)(end - p) < (size_t)*sec.n * sec.sz) return false;
        *sec.ptr = p;
        p += (size_t)*sec.n * sec.sz;
    }
    return true;
}

}  // namespace

// =================================================================================================
extern "C" {

const char* masp_hip_strerror(int code) {
    switch (code) {
        case MASP_HIP_OK: return "ok";
        case MASP_HIP_E_INVALID_ARG: return "invalid argument";
        case MASP_HIP_E_PARAMS_FORMAT: return "Parameters bytes are malformed";
        case MASP_HIP_E_PARAMS_SHAPE: return "Parameters do not match the circuit shape";
        case MASP_HIP_E_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
        case MASP_HIP_E_HIP: return "HIP runtime error";
        case MASP_HIP_E_UNEXPECTED_IDENTITY: return "delta is the identity (UnexpectedIdentity)";
        case MASP_HIP_E_NOT_LOADED: return "circuit slot is empty";
        case MASP_HIP_E_SCALAR_RANGE: return "scalar is not a canonical field element";
        default: return "unknown error";
    }
}

// The text is copied under the lock into a buffer of the calling thread: concurrent provers that fail at the same time
// each read a consistent message, and the pointer stays valid until this thread asks again.
const char* masp_hip_last_error(const masp_hip_ctx* ctx) {
    static thread_local std::string copy;
    if (!ctx) return "";
    if (!ctx->children.empty()) {
        for (const masp_hip_ctx* ch : ctx->children) {
            const char* e = masp_hip_last_error(ch);
            if (e[0]) return e;
        }
        return "";
    }
    std::lock_guard<std::mutex> g(ctx->slot_mu);
    copy = ctx->err;
    return copy.c_str();
}

int masp_hip_device_count(void) {
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess && count > 0 ? count : 0;
}

int masp_hip_device_pci_bus_id(int device, char* out, size_t cap) {
    if (!out || cap < 13) return MASP_HIP_E_INVALID_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return MASP_HIP_E_NO_DEVICE;
    return hipDeviceGetPCIBusId(out, (int)std::min<size_t>(cap, 64), device) == hipSuccess ? MASP_HIP_OK : MASP_HIP_E_HIP;
}

int masp_hip_runtime_prepare(int hw_queues, int overwrite) {
    char v[16];
    // (never more than 20: the runtime creates a hardware queue per new stream up to this number and never gives one back, and a process
    // that holds 24 / 32 of them dispatches every kernel 7 / 21 % slower — profiles/r06_second_context_root_cause.txt)
    snprintf(v, sizeof v, "%d", hw_queues > 0 ? std::min(hw_queues, (int)MASP_HIP_MAX_USEFUL_HW_QUEUES) : 16);
    setenv("GPU_MAX_HW_QUEUES", v, overwrite ? 1 : 0);
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    return e ? atoi(e) : 0;
}
// (see masp_hip_runtime_prepare in include/masp_hip.h: the one thing the library does to the environment)
__attribute__((constructor)) static void masp_hip_on_load() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

void masp_hip_options_default(masp_hip_options* opt) {
    if (!opt) return;
    memset(opt, 0, sizeof *opt);   // every field: 0 = the default (resolved when a context is created)
    opt->struct_size = sizeof *opt;
}

// ---- how many hardware queues do this process's streams really get?  (masp_hip_options::hw_queues) -------------------------------
// One single-wave kernel per stream, all launched at once, each spinning for 2 ms of the constant 100 MHz clock and leaving its
// [start, end): the largest number of them that overlapped = the hardware queues the streams were spread over (streams that share a
// queue run one after the other).  Temporary streams, created and destroyed BEFORE the context creates its own.
static __global__ void k_hwq_spin(unsigned long long* out, int idx, long long ticks) {
    const long long t0 = wall_clock64();
    // (bounded: a clock that does not advance must not hang the context's creation — 2^22 polls are ~100 ms)
    for (int it = 0; it < (1 << 22) && wall_clock64() - t0 < ticks; ++it) {
    }
    out[2 * idx] = (unsigned long long)t0;
    out[2 * idx + 1] = (unsigned long long)wall_clock64();
}
// Measured ONCE per (process, device) and remembered: the runtime fixes its queue count at the process's first HIP call, so a second
// context has nothing new to learn — and its probe would wait behind, and share the chip with, whatever the first context is running
// (ADVICE r05).  The probe synchronises its own streams only, and its buffer goes through dev_malloc / dev_free like every other.
static int measure_hw_queues(int device, int streams) {
    static std::mutex mu;
    static std::map<int, int> known;
    std::lock_guard<std::mutex> lock(mu);
    auto it = known.find(device);
    if (it != known.end()) return it->second;
    // (at most one more than MASP_HIP_MAX_USEFUL_HW_QUEUES: every stream created here makes the runtime create a hardware queue that it
    // keeps for the life of the process — the probe must not itself walk the pool into the slow regime; 21 says "more than 20")
    const int n = std::max(1, std::min(streams, (int)MASP_HIP_MAX_USEFUL_HW_QUEUES + 1));
    std::vector<hipStream_t> ss;
    unsigned long long* d = nullptr;
    int best = 0;
    if (dev_malloc(&d, 16 * n) != hipSuccess) return 0;
    for (int i = 0; i < n; ++i) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
        ss.push_back(s);
    }
    if ((int)ss.size() == n) {
        bool ok = true;
        for (int r = 0; r < 2; ++r) {  // (the first round loads the code object and creates the queues)
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_hwq_spin, dim3(1), dim3(64), 0, ss[i], d, i, r ? 200000LL : 1000LL);
            for (int i = 0; i < n; ++i) ok = hipStreamSynchronize(ss[i]) == hipSuccess && ok;
        }
        std::vector<unsigned long long> h(2 * n);
        if (ok && hipGetLastError() == hipSuccess && hipMemcpyAsync(h.data(), d, 16 * n, hipMemcpyDeviceToHost, ss[0]) == hipSuccess &&
            hipStreamSynchronize(ss[0]) == hipSuccess)
            for (int i = 0; i < n; ++i) {
                int c = 0;
                for (int j = 0; j < n; ++j) c += h[2 * j] <= h[2 * i] && h[2 * i] < h[2 * j + 1];
                best = std::max(best, c);
            }
    }
    for (hipStream_t s : ss) (void)hipStreamDestroy(s);
    (void)dev_free(d);
    (void)hipGetLastError();
    if (best > 0) known[device] = best;
    return best;
}

// The same measurement on a context's OWN streams (every slot's five + the main stream): how many of them run a kernel at the same time.
// Fewer than there are streams = two of them share a hardware queue, and kernels of one batch that could overlap wait for each other.
static int measure_streams(const std::vector<hipStream_t>& ss, int* n_streams, int* concurrent, std::vector<unsigned long long>* intervals = nullptr);
// mains_only: the context's own stream, the slots' main streams (the ones that carry batches) and the verifier's two
static int measure_own_streams(masp_hip_ctx* ctx, bool mains_only, int* n_streams, int* concurrent) {
    std::vector<hipStream_t> ss;
    ss.push_back(ctx->main_stream);
    for (size_t i = 0; i < ctx->slot_streams.size(); ++i) {
        ss.push_back(ctx->slot_streams[i][0]);
        if (!mains_only && slot_owns_side_streams(ctx, i))
            for (int j = 1; j < 5; ++j) ss.push_back(ctx->slot_streams[i][j]);
    }
    for (hipStream_t v : ctx->vk_streams) ss.push_back(v);
    return measure_streams(ss, n_streams, concurrent);
}
static int measure_streams(const std::vector<hipStream_t>& ss, int* n_streams, int* concurrent, std::vector<unsigned long long>* intervals) {
    const int n = (int)ss.size();
    unsigned long long* d = nullptr;
    HIP_TRY(dev_malloc(&d, 16 * n));
    int best = 0;
    bool ok = true;
    for (int r = 0; r < 2; ++r) {
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_hwq_spin, dim3(1), dim3(64), 0, ss[i], d, i, r ? 200000LL : 1000LL);
        for (int i = 0; i < n; ++i) ok = hipStreamSynchronize(ss[i]) == hipSuccess && ok;
    }
    std::vector<unsigned long long> h(2 * n);
    ok = ok && hipMemcpy(h.data(), d, 16 * n, hipMemcpyDeviceToHost) == hipSuccess;
    (void)dev_free(d);
    if (!ok) {
        last_hip_error() = "stream concurrency probe failed";
        return MASP_HIP_E_HIP;
    }
    for (int i = 0; i < n; ++i) {
        int c = 0;
        for (int j = 0; j < n; ++j) c += h[2 * j] <= h[2 * i] && h[2 * i] < h[2 * j + 1];
        best = std::max(best, c);
    }
    *n_streams = n;
    *concurrent = best;
    if (intervals) *intervals = h;
    return MASP_HIP_OK;
}

// The streams that carry batches — the context's own and every slot's main stream — must each sit on a hardware queue of its own: two
// of them on one queue run their batches one after the other (a later context of a process at 4 slots: -6 %, profiles/
// r06_second_context_root_cause.txt).  Which queue the runtime gives a new stream depends on everything the process has created before and
// cannot be asked, so it is MEASURED: a 2 ms single-wave kernel on each of them at once; a stream that starts a millisecond late shares its
// queue with the one that ended when it started.  Such a stream is replaced — the new one is created BEFORE the old one is destroyed, so
// that it cannot get the same queue back — and the measurement repeated, a few times at most.  Nothing has been launched on these streams
// yet.  Slot 0's main stream is never the one replaced (a lone proof runs 0.4 ms longer when its main stream was not created right in
// front of its side streams: profiles/r06_slot_streams_creation_order.txt).
static int separate_main_streams(masp_hip_ctx* ctx) {
    for (int attempt = 0; attempt < 8; ++attempt) {
        std::vector<hipStream_t*> at;
        at.push_back(&ctx->main_stream);
        for (auto& a : ctx->slot_streams) at.push_back(&a[0]);
        for (hipStream_t& v : ctx->vk_streams) at.push_back(&v);   // (the verifier's streams work next to the batches: the same rule)
        std::vector<hipStream_t> ss;
        for (hipStream_t* p : at) ss.push_back(*p);
        int n = 0, conc = 0;
        std::vector<unsigned long long> h;
        if (int rc = measure_streams(ss, &n, &conc, &h)) return rc;
        ctx->main_streams_concurrent = conc;
        if (conc >= n) return MASP_HIP_OK;
        unsigned long long t0 = ~0ull;
        for (int i = 0; i < n; ++i) t0 = std::min(t0, h[2 * i]);
        bool replaced = false;
        for (int i = 0; i < n; ++i) {
            if (h[2 * i] - t0 < 100000ull) continue;          // started with the others (100 MHz clock: 1 ms)
            int partner = -1;                                  // the stream whose kernel ended when this one's began
            unsigned long long bestd = ~0ull;
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                const unsigned long long d = h[2 * j + 1] > h[2 * i] ? h[2 * j + 1] - h[2 * i] : h[2 * i] - h[2 * j + 1];
                if (d < bestd) bestd = d, partner = j;
            }
            const int victim = i == 1 && partner >= 0 ? partner : i;   // (index 1 = slot 0's main stream)
            if (victim == 1) continue;
            hipStream_t fresh = nullptr;
            if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) return MASP_HIP_E_HIP;
            (void)hipStreamDestroy(*at[victim]);
            *at[victim] = fresh;
            replaced = true;
        }
        if (!replaced) return MASP_HIP_OK;
    }
    return MASP_HIP_OK;   // (still shared after eight rounds: a process with fewer queues than main streams — reported, not an error)
}

static int create_single(int device, const masp_hip_options& opt, masp_hip_ctx** out) {
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return MASP_HIP_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MASP_HIP_E_NO_DEVICE;
    std::unique_ptr<masp_hip_ctx> ctx(new masp_hip_ctx);
    ctx->device = device;
    ctx->opt = opt;
    ctx->n_slots = opt.slots;
    ctx->batch_cap = (size_t)opt.batch_cap;
    ctx->slots.reserve(masp_hip_ctx::MAX_SLOTS);      // never reallocates: see the locking note on masp_hip_ctx
    ctx->slot_busy.reserve(masp_hip_ctx::MAX_SLOTS);
    ctx->opt.hw_queues = measure_hw_queues(device, 5 * opt.slots);   // (a slot owns up to five streams)
    if (hipStreamCreateWithFlags(&ctx->main_stream, hipStreamNonBlocking) != hipSuccess) return MASP_HIP_E_NO_DEVICE;
    // every slot's streams now, main streams first (masp_hip_ctx::slot_streams)
    ctx->slot_streams.assign((size_t)opt.slots, std::array<hipStream_t, 5>{});
    ctx->slot_streams_taken.assign((size_t)opt.slots, 0);
    bool streams_ok = true;
    // order: slot by slot, a slot's main stream right in front of its side streams (a lone proof takes the first free slot, i.e. slot 0, and
    // runs 0.3 - 0.4 ms longer when its side streams were not created right behind its main stream — measured three ways, not understood:
    // profiles/r06_slot_streams_creation_order.txt); that the main streams then do not share hardware queues is measured and repaired:
    // separate_main_streams
    auto make = [&](int si, int j) {
        if (j > 0 && !slot_owns_side_streams(ctx.get(), (size_t)si))
            ctx->slot_streams[si][j] = ctx->slot_streams[1][j];   // (slot 1's: created before, in every order below)
        else
            streams_ok = streams_ok && hipStreamCreateWithFlags(&ctx->slot_streams[si][j], hipStreamNonBlocking) == hipSuccess;
    };
#if MASP_STREAM_ORDER == 1      // (A/B builds) every main stream first
    for (int si = 0; si < opt.slots; ++si) make(si, 0);
    for (int si = 0; si < opt.slots; ++si)
        for (int j = 1; j < 5; ++j) make(si, j);
#elif MASP_STREAM_ORDER == 2    // (A/B builds) slot by slot, as the slots created them before round 6
    for (int si = 0; si < opt.slots; ++si)
        for (int j = 0; j < 5; ++j) make(si, j);
#elif MASP_STREAM_ORDER == 3    // slot 0's side streams, then every main stream (slot 0's first), then the other side streams
    for (int j = 1; j < 5; ++j) make(0, j);
    for (int si = 0; si < opt.slots; ++si) make(si, 0);
    for (int si = 1; si < opt.slots; ++si)
        for (int j = 1; j < 5; ++j) make(si, j);
#else                           // slot 0's five, the other slots' main streams, their side streams
    for (int j = 0; j < 5; ++j) make(0, j);
    for (int si = 1; si < opt.slots; ++si) make(si, 0);
    for (int si = 1; si < opt.slots; ++si)
        for (int j = 1; j < 5; ++j) make(si, j);
#endif
    for (hipStream_t& v : ctx->vk_streams) streams_ok = streams_ok && hipStreamCreateWithFlags(&v, hipStreamNonBlocking) == hipSuccess;
    if (streams_ok && separate_main_streams(ctx.get()) != MASP_HIP_OK) streams_ok = false;
    if (!streams_ok) {
        for (size_t si = 0; si < ctx->slot_streams.size(); ++si)
            for (int j = 0; j < 5; ++j)
                if (ctx->slot_streams[si][j] && (j == 0 || slot_owns_side_streams(ctx.get(), si))) hipStreamDestroy(ctx->slot_streams[si][j]);
        for (hipStream_t v : ctx->vk_streams)
            if (v) hipStreamDestroy(v);
        hipStreamDestroy(ctx->main_stream);
        return MASP_HIP_E_NO_DEVICE;
    }
    *out = ctx.release();
    return MASP_HIP_OK;
}

int masp_hip_ctx_create_ex(const int* devices, int n_devices, const masp_hip_options* opt_in, masp_hip_ctx** out) {
    if (!out || !devices || n_devices <= 0 || n_devices > 64) return MASP_HIP_E_INVALID_ARG;
    if (opt_in && opt_in->struct_size != 0 && opt_in->struct_size < 4 * sizeof(int32_t)) return MASP_HIP_E_INVALID_ARG;
    *out = nullptr;
    const masp_hip_options opt = resolve_options(opt_in);
    if (n_devices == 1) return create_single(devices[0], opt, out);
    std::unique_ptr<masp_hip_ctx> front(new masp_hip_ctx);
    front->device = devices[0];
    front->opt = opt;
    front->n_slots = opt.slots;
    front->batch_cap = (size_t)opt.batch_cap;
    for (int i = 0; i < n_devices; ++i) {
        masp_hip_ctx* ch = nullptr;
        int rc = create_single(devices[i], opt, &ch);
        if (rc) {
            for (masp_hip_ctx* c : front->children) masp_hip_ctx_destroy(c);
            return rc;
        }
        front->children.push_back(ch);
    }
    front->dev_status.reset(new std::atomic<int>[n_devices]);
    for (int i = 0; i < n_devices; ++i) front->dev_status[i] = MASP_HIP_OK;
    *out = front.release();
    return MASP_HIP_OK;
}

int masp_hip_ctx_get_options(const masp_hip_ctx* ctx, masp_hip_options* out) {
    if (!ctx || !out) return MASP_HIP_E_INVALID_ARG;
    const masp_hip_ctx* c = ctx->children.empty() ? ctx : ctx->children[0];
    // `out->struct_size` on entry = the size of the CALLER's struct (0: this header's): a binding built against an older, shorter
    // masp_hip_options gets only the bytes its struct has room for (the plain assignment wrote sizeof(new struct) into it)
    const size_t room = out->struct_size ? std::min<size_t>(out->struct_size, sizeof(masp_hip_options)) : sizeof(masp_hip_options);
    if (room < 4 * sizeof(int32_t)) return MASP_HIP_E_INVALID_ARG;
    masp_hip_options full = c->opt, *const caller_out = out;
    out = &full;
    // what lack of tree scratch has changed since the context was created: the smallest sub-batch any slot works with now, and
    // the proofs whose bucket runs went through the XYZZ accumulation instead (all device contexts)
    uint64_t fb = 0;
    int sub = out->bucket_tree_sub_batch;
    auto scan = [&](const masp_hip_ctx* d) {
        std::lock_guard<std::mutex> g(d->slot_mu);
        for (const auto& sl : d->slots) {
            fb += sl->ws1.tree_fallbacks + sl->ws2.tree_fallbacks;
            sub = std::min<int>(sub, (int)std::min(sl->ws1.tree_sub, sl->ws2.tree_sub));
        }
    };
    if (ctx->children.empty())
        scan(ctx);
    else
        for (const masp_hip_ctx* d : ctx->children) scan(d);
    fb += c->block_tree_fallbacks.load();
    if (c->block_tree_sub.load() != 0xffffffffu) sub = std::min<int>(sub, (int)c->block_tree_sub.load());
    out->bucket_tree_sub_batch = sub;
    out->bucket_tree_fallback_proofs = (int32_t)std::min<uint64_t>(fb, 0x7fffffff);
    full.struct_size = (uint32_t)room;   // the bytes written
    memcpy(caller_out, &full, room);
    return MASP_HIP_OK;
}

int masp_hip_ctx_create(int device, masp_hip_ctx** out) {
    if (!out) return MASP_HIP_E_INVALID_ARG;
    return masp_hip_ctx_create_ex(&device, 1, nullptr, out);
}

// (a list of ONE device still gives the multi-device front: its callers test the sharding path with the same GPU listed twice
// or once)
int masp_hip_ctx_create_multi(const int* devices, int n_devices, masp_hip_ctx** out) {
    if (!out || !devices || n_devices <= 0 || n_devices > 64) return MASP_HIP_E_INVALID_ARG;
    if (n_devices > 1) return masp_hip_ctx_create_ex(devices, n_devices, nullptr, out);
    *out = nullptr;
    std::unique_ptr<masp_hip_ctx> front(new masp_hip_ctx);
    front->device = devices[0];
    front->opt = resolve_options(nullptr);
    front->batch_cap = (size_t)front->opt.batch_cap;
    masp_hip_ctx* ch = nullptr;
    int rc = create_single(devices[0], front->opt, &ch);
    if (rc) return rc;
    front->children.push_back(ch);
    front->dev_status.reset(new std::atomic<int>[1]);
    front->dev_status[0] = MASP_HIP_OK;
    *out = front.release();
    return MASP_HIP_OK;
}

int masp_hip_ctx_device_proofs(const masp_hip_ctx* ctx, uint64_t* counts, int cap) {
    if (!ctx || !counts || cap < 0) return MASP_HIP_E_INVALID_ARG;
    if (ctx->children.empty()) {
        if (cap >= 1) counts[0] = ctx->proofs_done.load();
        return MASP_HIP_OK;
    }
    for (size_t d = 0; d < ctx->children.size() && (int)d < cap; ++d) counts[d] = ctx->children[d]->proofs_done.load();
    return MASP_HIP_OK;
}
int masp_hip_ctx_device_status(const masp_hip_ctx* ctx, int32_t* status, int cap, uint64_t* requeued) {
    if (!ctx || !status || cap < 0) return MASP_HIP_E_INVALID_ARG;
    if (requeued) *requeued = ctx->requeued.load();
    if (ctx->children.empty()) {
        if (cap >= 1) status[0] = MASP_HIP_OK;
        return MASP_HIP_OK;
    }
    for (size_t d = 0; d < ctx->children.size() && (int)d < cap; ++d) status[d] = ctx->dev_status[d].load();
    return MASP_HIP_OK;
}
int masp_hip_ctx_inject_fault(masp_hip_ctx* ctx, int device, uint32_t nth) {
    if (!ctx || device < 0) return MASP_HIP_E_INVALID_ARG;
    if (ctx->children.empty()) {
        if (device != 0) return MASP_HIP_E_INVALID_ARG;
        ctx->fault_countdown = nth;
        return MASP_HIP_OK;
    }
    if ((size_t)device >= ctx->children.size()) return MASP_HIP_E_INVALID_ARG;
    ctx->children[device]->fault_countdown = nth;
    return MASP_HIP_OK;
}
int masp_hip_ctx_stream_concurrency(masp_hip_ctx* ctx, int mains_only, int* n_streams, int* concurrent) {
    if (!ctx || !n_streams || !concurrent) return MASP_HIP_E_INVALID_ARG;
    if (!ctx->children.empty()) ctx = ctx->children[0];
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    if (hipDeviceSynchronize() != hipSuccess) return fail(ctx, MASP_HIP_E_HIP);
    return fail(ctx, measure_own_streams(ctx, mains_only != 0, n_streams, concurrent));
}
int masp_hip_ctx_lone_graph_launches(const masp_hip_ctx* ctx, uint64_t* out) {
    if (!ctx || !out) return MASP_HIP_E_INVALID_ARG;
    *out = 0;
    if (!ctx->children.empty()) {
        for (const masp_hip_ctx* ch : ctx->children) {
            uint64_t v = 0;
            masp_hip_ctx_lone_graph_launches(ch, &v);
            *out += v;
        }
        return MASP_HIP_OK;
    }
    std::lock_guard<std::mutex> g(ctx->slot_mu);
    for (const auto& sl : ctx->slots) *out += sl->graph_launches.load();
    return MASP_HIP_OK;
}
int masp_hip_ctx_device_count(const masp_hip_ctx* ctx) { return !ctx ? 0 : ctx->children.empty() ? 1 : (int)ctx->children.size(); }

void masp_hip_ctx_destroy(masp_hip_ctx* ctx) {
    if (!ctx) return;
    if (!ctx->children.empty()) {
        for (masp_hip_ctx* c : ctx->children) masp_hip_ctx_destroy(c);
        delete ctx;
        return;
    }
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    ctx->batches.clear();
    ctx->slots.clear();
    for (auto& c : ctx->circ) c.reset();
    ctx->domains.clear();
    for (size_t i = 0; i < ctx->slot_streams.size(); ++i)    // (streams of slots that were never created)
        if (!ctx->slot_streams_taken[i])
            for (int j = 0; j < 5; ++j)
                if (ctx->slot_streams[i][j] && (j == 0 || slot_owns_side_streams(ctx, i))) hipStreamDestroy(ctx->slot_streams[i][j]);
    for (hipStream_t v : ctx->vk_streams)
        if (v) hipStreamDestroy(v);
    if (ctx->main_stream) hipStreamDestroy(ctx->main_stream);
    delete ctx;
    dev_free_drain();  // the context's buffers
}

int masp_hip_circuit_load(masp_hip_ctx* ctx, uint32_t slot, const uint8_t* params, size_t params_len, const masp_hip_r1cs* cs) {
    const ApiLaunchScope api_scope;
    if (!ctx || !params || !cs || slot >= MASP_HIP_MAX_CIRCUITS || cs->n_inputs == 0) return MASP_HIP_E_INVALID_ARG;
    if (!ctx->children.empty()) {  // the CRS is replicated: every device builds its own window tables, side by side
        std::vector<int> rcs(ctx->children.size(), MASP_HIP_OK);
        std::vector<std::thread> th;
        for (size_t d = 0; d < ctx->children.size(); ++d)
            th.emplace_back([&, d] { rcs[d] = masp_hip_circuit_load(ctx->children[d], slot, params, params_len, cs); });
        for (auto& t : th) t.join();
        for (int rc : rcs)
            if (rc) return rc;
        return MASP_HIP_OK;
    }
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    hipStream_t s = ctx->main_stream;
    ParamsLayout L;
    if (!parse_params(params, params_len, L)) return MASP_HIP_E_PARAMS_FORMAT;
    std::unique_ptr<Circuit> C(new Circuit);
    C->n_inputs = cs->n_inputs;
    C->n_aux = cs->n_aux;
    C->n_constraints = cs->n_constraints;
    C->nrows = cs->n_constraints + cs->n_inputs;
    C->logm = log2_ceil(C->nrows);
    C->m = (size_t)1 << C->logm;
    // density lists from the static structure (bellperson DensityTracker)
    const uint32_t nv = cs->n_inputs + cs->n_aux;
    std::vector<uint8_t> a_dense(nv, 0), b_dense(nv, 0);
    const uint32_t* rp[3] = {cs->a_rowptr, cs->b_rowptr, cs->c_rowptr};
    const uint32_t* cl[3] = {cs->a_col, cs->b_col, cs->c_col};
    const uint8_t* cf[3] = {cs->a_coef, cs->b_coef, cs->c_coef};
    for (int mi = 0; mi < 3; ++mi) {
        uint32_t nnz = rp[mi][cs->n_constraints];
        for (uint32_t t = 0; t < nnz; ++t) {
            if (cl[mi][t] >= nv) return MASP_HIP_E_INVALID_ARG;
            if (mi == 0) a_dense[cl[mi][t]] = 1;
            if (mi == 1) b_dense[cl[mi][t]] = 1;
        }
    }
    std::vector<uint32_t> a_var, b_var;
    for (uint32_t i = 0; i < cs->n_inputs; ++i) a_var.push_back(i);  // inputs are always dense for A
    for (uint32_t v = cs->n_inputs; v < nv; ++v)
        if (a_dense[v]) a_var.push_back(v);
    for (uint32_t v = 0; v < nv; ++v)
        if (b_dense[v]) b_var.push_back(v);
    C->na = a_var.size();
    C->nbq = b_var.size();
    // length invariants
    if (L.n_ic != cs->n_inputs || L.n_h < C->m - 1 || L.n_l != cs->n_aux || L.n_a != C->na || L.n_b1 != C->nbq || L.n_b2 != C->nbq)
        return MASP_HIP_E_PARAMS_SHAPE;
    int rc;
    if ((rc = C->a_var.upload(a_var.data(), a_var.size(), s)) || (rc = C->b_var.upload(b_var.data(), b_var.size(), s))) return fail(ctx, rc);
    // static R1CS -> device (coefficients to Montgomery form)
    DevBuf<int> d_flag;
    if ((rc = d_flag.reserve(1))) return fail(ctx, rc);
    hipMemsetAsync(d_flag.p, 0, sizeof(int), s);
    for (int mi = 0; mi < 3; ++mi) {
        uint32_t nnz = rp[mi][cs->n_constraints];
        DevBuf<Fr> raw;
        std::vector<uint32_t> order(cs->n_constraints);
        for (uint32_t r = 0; r < cs->n_constraints; ++r) order[r] = r;
        std::stable_sort(order.begin(), order.end(),
                         [&](uint32_t x, uint32_t y) { return rp[mi][x + 1] - rp[mi][x] > rp[mi][y + 1] - rp[mi][y]; });
        if ((rc = C->row_order[mi].upload(order.data(), order.size(), s))) return fail(ctx, rc);
        C->n_long_rows[mi] = 0;
        for (uint32_t r : order) {
            if (rp[mi][r + 1] - rp[mi][r] < R1CS_LONG_ROW) break;
            ++C->n_long_rows[mi];
        }
        if ((rc = C->rowptr[mi].upload(rp[mi], cs->n_constraints + 1, s)) || (rc = C->col[mi].upload(cl[mi], nnz, s)) ||
            (rc = raw.upload((const Fr*)cf[mi], nnz, s)) || (rc = C->coef[mi].reserve(nnz)))
            return fail(ctx, rc);
        if (nnz) launch_fr_to_mont(s, raw.p, (size_t)0, C->coef[mi].p, nnz, 1, d_flag.p);
        if (hipStreamSynchronize(s) != hipSuccess) return fail(ctx, MASP_HIP_E_HIP);
    }
    // verifying-key points used by the prover
    if ((rc = C->vk.reserve(1))) return fail(ctx, rc);
    {
        DevBuf<uint8_t> raw;
        if ((rc = raw.upload(params, 864, s))) return fail(ctx, rc);
        DevBuf<int> st;
        if ((rc = st.reserve(1))) return fail(ctx, rc);
        hipMemsetAsync(st.p, 0, sizeof(int), s);
        VkDevice* v = C->vk.p;
        launch_g1_import_one(s, raw.p + 0, &v->alpha_g1, st.p);
        launch_g1_import_one(s, raw.p + 96, &v->beta_g1, st.p);
        launch_g2_import_one(s, raw.p + 192, &v->beta_g2, st.p);
        launch_g1_import_one(s, raw.p + 576, &v->delta_g1, st.p);
        launch_g2_import_one(s, raw.p + 672, &v->delta_g2, st.p);
        int hst = 0, hflag = 0;
        if (hipMemcpyAsync(&hst, st.p, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(&hflag, d_flag.p, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            last_hip_error() = "vk import failed";
            return fail(ctx, MASP_HIP_E_HIP);
        }
        if (hflag) return MASP_HIP_E_SCALAR_RANGE;
        if (hst & (PT_BAD_FLAGS | PT_NOT_CANONICAL)) return MASP_HIP_E_PARAMS_FORMAT;
        // fixed-base tables for the points every proof multiplies by r, s, rs
        DevBuf<G1Affine> p1;
        DevBuf<G2Affine> p2;
        if ((rc = p1.reserve(3)) || (rc = p2.reserve(1)) || (rc = C->fb1.reserve(3 * 32 * 255)) || (rc = C->fb2.reserve(32 * 255))) return fail(ctx, rc);
        hipMemcpyAsync(p1.p + 0, &v->delta_g1, sizeof(G1Affine), hipMemcpyDeviceToDevice, s);
        hipMemcpyAsync(p1.p + 1, &v->alpha_g1, sizeof(G1Affine), hipMemcpyDeviceToDevice, s);
        hipMemcpyAsync(p1.p + 2, &v->beta_g1, sizeof(G1Affine), hipMemcpyDeviceToDevice, s);
        hipMemcpyAsync(p2.p, &v->delta_g2, sizeof(G2Affine), hipMemcpyDeviceToDevice, s);
        launch_fixed_table_g1(s, p1.p, C->fb1.p, 3);
        launch_fixed_table_g2(s, p2.p, C->fb2.p, 1);
        if (hipStreamSynchronize(s) != hipSuccess) {
            last_hip_error() = std::string("fixed-base tables failed: ") + hipGetErrorString(hipGetLastError());
            return fail(ctx, MASP_HIP_E_HIP);
        }
        // delta at infinity -> bellperson's UnexpectedIdentity; alpha/beta at infinity are merely degenerate
        if ((params[576] & 0x40) || (params[672] & 0x40)) return MASP_HIP_E_UNEXPECTED_IDENTITY;
    }
    // query vectors + window tables
    // h scalars are uniform in Fr; the witness queries are mostly 0 / 1 (70 % of a Spend witness is
    // boolean-constrained), so their effective size for window selection is a fraction of their length
    const uint32_t pct = (uint32_t)ctx->opt.witness_nontrivial_percent;
    auto eff = [&](uint32_t n) { return (uint32_t)((uint64_t)n * pct / 100); };
    const int c_la = ctx->opt.window_bits_la, c_b = ctx->opt.window_bits_b;   // 0: chosen from the expected non-trivial scalars
    // b_g2 of a batch: its own width if asked for (masp_hip_options::window_bits_b2), else b_g1's — it is then reduced from b_g1's sorted
    // digit list instead of sorting for itself (enqueue_proofs: share_bq)
    const int c_b2 = ctx->opt.window_bits_b2 > 0 ? ctx->opt.window_bits_b2 : c_b;
    // h has uniform scalars: 16-bit windows from 48 k points (Spend 131 071, Convert 65 535), 15 bits below (Output 32 767)
    const uint32_t n_h = (uint32_t)(C->m - 1);
    const int c_h = ctx->opt.window_bits_h ? ctx->opt.window_bits_h : n_h >= 49152 ? 16 : n_h >= 16384 ? 15 : 0;
    // (h's own table serves lone proofs only: its window width is theirs to choose — masp_hip_options::window_bits_h_lone)
    const int c_h_lone = ctx->opt.window_bits_h_lone ? ctx->opt.window_bits_h_lone : c_h;
    if ((rc = C->h.load_host(L.h, (uint32_t)(C->m - 1), s, 0xffffffffu, c_h_lone)) || (rc = C->l.load_host(L.l, L.n_l, s, eff(L.n_l), c_la)) ||
        (rc = C->a.load_host(L.a, L.n_a, s, eff(L.n_a), c_la)) || (rc = C->b1.load_host(L.b_g1, L.n_b1, s, eff(L.n_b1), c_b)) ||
        (rc = C->b2.load_host(L.b_g2, L.n_b2, s, eff(L.n_b2), c_b2)))
        return fail(ctx, rc);
    {
        const int c_lone = ctx->opt.window_bits_b2_lone;  // 0 = lone proofs share the batch tables (and B1's sort)
        if (c_lone > 0 && L.n_b2 && (rc = C->b2_lone.load_host(L.b_g2, L.n_b2, s, eff(L.n_b2), c_lone))) return fail(ctx, rc);
    }
    {
        const size_t nh = C->m - 1;
        std::vector<uint8_t> cat(96 * (nh + L.n_l));
        memcpy(cat.data(), L.h, 96 * nh);
        memcpy(cat.data() + 96 * nh, L.l, 96 * (size_t)L.n_l);
        const int c_hl = c_h ? c_h : ctx->opt.window_bits_h_lone ? 0 : C->h.g.c;
        if ((rc = C->hl.load_host(cat.data(), (uint32_t)(nh + L.n_l), s, 0xffffffffu, c_hl))) return fail(ctx, rc);
    }
    int st = C->h.import_status | C->l.import_status | C->a.import_status | C->b1.import_status | C->b2.import_status;
    if (st) return MASP_HIP_E_PARAMS_FORMAT;  // includes infinity inside a query vector, which bellman rejects
    {
        // may s*A and r*B1 use the endomorphism?  Only if everything A and B1 are sums of is in the subgroup (Circuit::g1_endo)
        DevBuf<int> out;
        if ((rc = out.reserve(1))) return fail(ctx, rc);
        hipMemsetAsync(out.p, 0, sizeof(int), s);
        launch_g1_subgroup_flag(s, C->a.tab, sizeof(*C->a.tab), C->a.n, out.p);
        launch_g1_subgroup_flag(s, C->b1.tab, sizeof(*C->b1.tab), C->b1.n, out.p);
        launch_g1_subgroup_flag(s, &C->vk.p->alpha_g1, 0, 1, out.p);
        launch_g1_subgroup_flag(s, &C->vk.p->beta_g1, 0, 1, out.p);
        launch_g1_subgroup_flag(s, &C->vk.p->delta_g1, 0, 1, out.p);
        int outside = 1;
        if (hipMemcpyAsync(&outside, out.p, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess ||
            launch_status() != MASP_HIP_OK)
            return fail(ctx, MASP_HIP_E_HIP);
        C->g1_endo = outside == 0;
    }
    if ((rc = get_domain(ctx, C->logm, &C->dom))) return fail(ctx, rc);
    ctx->circ[slot] = std::move(C);
    return MASP_HIP_OK;
}

int masp_hip_circuit_flags(const masp_hip_ctx* ctx, uint32_t slot, uint32_t* flags) {
    if (!ctx || !flags) return MASP_HIP_E_INVALID_ARG;
    if (!ctx->children.empty()) return masp_hip_circuit_flags(ctx->children[0], slot, flags);
    std::shared_lock<std::shared_mutex> lock(const_cast<masp_hip_ctx*>(ctx)->mu);
    if (slot >= MASP_HIP_MAX_CIRCUITS || !ctx->circ[slot]) return MASP_HIP_E_NOT_LOADED;
    *flags = ctx->circ[slot]->g1_endo ? MASP_HIP_CIRCUIT_G1_ENDOMORPHISM : 0u;
    return MASP_HIP_OK;
}

// What masp_hip_prove_batch refuses before it touches a device — the same tests for a single-device context and for the front of a
// multi-device one (there they must not be mistaken for a device's failure)
static int validate_jobs(const masp_hip_ctx* dev_ctx, size_t n, const masp_hip_job* jobs) {
    for (size_t j = 0; j < n; ++j) {
        const masp_hip_job& J = jobs[j];
        if (J.circuit >= MASP_HIP_MAX_CIRCUITS || !J.inputs || !J.aux) return MASP_HIP_E_INVALID_ARG;
        if (!dev_ctx->circ[J.circuit]) return MASP_HIP_E_NOT_LOADED;
        if ((J.a || J.b || J.c) && !(J.a && J.b && J.c)) return MASP_HIP_E_INVALID_ARG;
        if (J.reserved != 0) return MASP_HIP_E_INVALID_ARG;  // must be zero: a later revision of the struct can then give it a meaning
        if (J.aux_form > MASP_HIP_AUX_MONTGOMERY || (J.aux_form && J.a)) return MASP_HIP_E_INVALID_ARG;  // (a, b, c given: nothing reads aux as Montgomery)
        if (!rs_in_range(J.r) || !rs_in_range(J.s)) return MASP_HIP_E_SCALAR_RANGE;
    }
    return MASP_HIP_OK;
}

// Multi-device front (round 6: a queue instead of static shares).  The jobs of every kind (circuit, aux form, a/b/c mode) are cut into
// blocks of up to batch_cap proofs; the blocks wait on ONE queue, the most expensive first (constraints x proofs: a Spend block costs
// three Output blocks), and every device has `slots` host threads that each take the next block when their last one is done — a device
// that clocks lower, or shares its GPU, simply takes fewer.  A device whose call fails with a HIP error is taken OUT (for the life of
// the context: masp_hip_ctx_device_status), its block goes back on the queue and the other devices finish the list; the call fails only
// when no device is left, or for an error of the input (MASP_HIP_E_SCALAR_RANGE from the device's range check of an assignment), which
// no other device would cure.  Proofs are independent: there is no data-path exchange between devices; results are copied to the jobs'
// own positions.  (The reference's loop fails per description: /root/reference/masp_primitives/src/transaction/components/sapling/builder.rs:955-969.)
static int prove_batch_multi(masp_hip_ctx* ctx, size_t n, const masp_hip_job* jobs, uint8_t* proofs_out) {
    const size_t nd = ctx->children.size();
    auto alive = [&](size_t d) { return ctx->dev_status[d].load() == MASP_HIP_OK; };
    size_t n_alive = 0, first_alive = nd;
    for (size_t d = 0; d < nd; ++d)
        if (alive(d)) {
            ++n_alive;
            if (first_alive == nd) first_alive = d;
        }
    if (!n_alive) return MASP_HIP_E_HIP;   // (masp_hip_last_error: the text of the device that failed first)
    {
        std::shared_lock<std::shared_mutex> lock(ctx->children[first_alive]->mu);
        if (int rc = validate_jobs(ctx->children[first_alive], n, jobs)) return rc;
    }
    struct Block {
        std::vector<size_t> idx;
        uint64_t cost;
    };
    std::deque<Block> queue;
    {
        std::map<std::pair<uint32_t, bool>, std::vector<size_t>> by_kind;
        for (size_t j = 0; j < n; ++j) by_kind[{jobs[j].circuit | (jobs[j].aux_form ? 0x100u : 0u), jobs[j].a != nullptr}].push_back(j);
        std::vector<Block> blocks;
        for (auto& kv : by_kind) {
            // at least one block per device when the list is long enough to give every device a useful batch
            const size_t cap = std::min(ctx->batch_cap, std::max<size_t>((kv.second.size() + n_alive - 1) / n_alive, 8));
            const Circuit& C = *ctx->children[first_alive]->circ[kv.first.first & 0xffu];
            for (auto& g : even_groups(kv.second.size(), cap)) {
                Block b;
                b.idx.assign(kv.second.begin() + g.first, kv.second.begin() + g.first + g.second);
                b.cost = (uint64_t)C.nrows * g.second;
                blocks.push_back(std::move(b));
            }
        }
        std::stable_sort(blocks.begin(), blocks.end(), [](const Block& x, const Block& y) { return x.cost > y.cost; });
        for (auto& b : blocks) queue.push_back(std::move(b));
    }
    std::mutex mu;                 // queue, pending, fatal
    std::condition_variable cv;
    size_t pending = queue.size();  // blocks not yet proved (queued or in some worker's hands)
    int fatal = MASP_HIP_OK;
    auto worker = [&](size_t d) {
        masp_hip_ctx* ch = ctx->children[d];
        std::vector<masp_hip_job> mine;
        std::vector<uint8_t> out;
        for (;;) {
            Block b;
            {
                std::unique_lock<std::mutex> g(mu);
                // (an empty queue is not the end: a block in another worker's hands may come back when its device fails)
                cv.wait(g, [&] { return !queue.empty() || pending == 0 || fatal || !alive(d); });
                if (pending == 0 || fatal || !alive(d)) return;
                b = std::move(queue.front());
                queue.pop_front();
            }
            mine.resize(b.idx.size());
            for (size_t k = 0; k < mine.size(); ++k) mine[k] = jobs[b.idx[k]];
            out.resize(192 * mine.size());
            const int rc = masp_hip_prove_batch(ch, mine.size(), mine.data(), out.data());
            std::unique_lock<std::mutex> g(mu);
            if (rc == MASP_HIP_OK) {
                for (size_t k = 0; k < b.idx.size(); ++k) memcpy(proofs_out + 192 * b.idx[k], out.data() + 192 * k, 192);
                --pending;
            } else if (rc == MASP_HIP_E_HIP || rc == MASP_HIP_E_NO_DEVICE) {
                int ok = MASP_HIP_OK;
                ctx->dev_status[d].compare_exchange_strong(ok, rc);   // the device is out; its other workers leave at their next turn
                ctx->requeued += b.idx.size();
                queue.push_front(std::move(b));                      // (an expensive block that has already waited: first again)
                bool any = false;
                for (size_t e = 0; e < nd; ++e) any = any || alive(e);
                if (!any) fatal = rc;
            } else {
                fatal = rc;   // an error of the input: no other device would decide differently
            }
            cv.notify_all();
            if (rc != MASP_HIP_OK) return;
        }
    };
    std::vector<std::thread> th;
    for (size_t d = 0; d < nd; ++d) {
        if (!alive(d)) continue;
        // as many calls in flight per device as it has slots (masp_hip_prove_batch of a device context is re-entrant: each call's batches
        // take slots of its pool), but not more workers than there are blocks
        const size_t w = std::min<size_t>(std::max(ctx->children[d]->n_slots, 1), std::max<size_t>((queue.size() + n_alive - 1) / n_alive, 1));
        for (size_t k = 0; k < w; ++k) th.emplace_back(worker, d);
    }
    for (auto& t : th) t.join();
    if (fatal) return fatal;
    return pending == 0 ? MASP_HIP_OK : MASP_HIP_E_HIP;
}

int masp_hip_prove_batch(masp_hip_ctx* ctx, size_t n, const masp_hip_job* jobs, uint8_t* proofs_out) {
    const ApiLaunchScope api_scope;
    if (!ctx || (n && (!jobs || !proofs_out))) return MASP_HIP_E_INVALID_ARG;
    if (!ctx->children.empty()) return prove_batch_multi(ctx, n, jobs, proofs_out);
    std::shared_lock<std::shared_mutex> lock(ctx->mu);  // concurrent with other provers, exclusive with circuit loads
    if (int rc = validate_jobs(ctx, n, jobs)) return rc;
    // test hook (masp_hip_ctx_inject_fault): this call fails as a lost device's would, before anything is enqueued
    for (uint32_t left = ctx->fault_countdown.load(); left != 0;) {
        if (!ctx->fault_countdown.compare_exchange_weak(left, left - 1)) continue;
        if (left == 1) {
            last_hip_error() = "injected fault (masp_hip_ctx_inject_fault): this device context's call fails as a lost GPU's would";
            return fail_shared(ctx, MASP_HIP_E_HIP);
        }
        break;
    }
    hipSetDevice(ctx->device);
    // jobs are bucketed by (circuit, a/b/c mode) — whatever their order in the list — and every bucket is cut into
    // equal batches of at most batch_cap proofs (256 Spends with a cap of 96: 86 + 85 + 85, not 96 + 96 + 64); results go
    // back to the jobs' own positions
    struct Group {
        std::vector<size_t> idx;
    };
    std::vector<Group> groups;
    {
        std::map<std::pair<uint32_t, bool>, std::vector<size_t>> by_kind;
        for (size_t j = 0; j < n; ++j) by_kind[{jobs[j].circuit | (jobs[j].aux_form ? 0x100u : 0u), jobs[j].a != nullptr}].push_back(j);
        for (auto& kv : by_kind)
            for (auto& eg : even_groups(kv.second.size(), ctx->batch_cap)) {
                Group g;
                g.idx.assign(kv.second.begin() + eg.first, kv.second.begin() + eg.first + eg.second);
                groups.push_back(std::move(g));
            }
    }
    // Each batch runs on a slot taken from the context's pool; this call keeps up to all of them in flight, and calls
    // from other host threads interleave with it slot by slot.
    struct Owned {
        size_t si, gi;
    };
    std::deque<Owned> owned;
    int result = MASP_HIP_OK, rc;
    auto retire_oldest = [&]() -> size_t {  // waits for the oldest batch in flight; the slot stays ours
        Owned o = owned.front();
        owned.pop_front();
        Slot& sl = *ctx->slots[o.si];
        const Group& G = groups[o.gi];
        if (hipEventSynchronize(sl.done) != hipSuccess) {
            last_hip_error() = "event sync failed";
            result = fail_shared(ctx, MASP_HIP_E_HIP);
        } else if (*sl.h_flags) {
            result = MASP_HIP_E_SCALAR_RANGE;
        } else if (result == MASP_HIP_OK) {
            for (size_t p = 0; p < G.idx.size(); ++p) memcpy(proofs_out + 192 * G.idx[p], sl.h_proof + 192 * p, 192);
        }
        return o.si;
    };
#ifdef MASP_ENQ_TIMING
    const auto tcall = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
#endif
    for (size_t gi = 0; gi < groups.size() && result == MASP_HIP_OK; ++gi) {
        size_t si = 0;
#ifdef MASP_ENQ_TIMING
        const auto tg0 = std::chrono::steady_clock::now();
#endif
        rc = slot_try_acquire(ctx, &si);
        if (rc > 0) {
            result = fail_shared(ctx, rc);
            break;
        }
        if (rc < 0) si = owned.empty() ? slot_acquire_blocking(ctx) : retire_oldest();
        if (result) {
            slot_release(ctx, si);
            break;
        }
        const Group& G = groups[gi];
        Slot& sl = *ctx->slots[si];
        Circuit& C = *ctx->circ[jobs[G.idx[0]].circuit];
        const size_t nv = (size_t)C.n_inputs + C.n_aux, np = G.idx.size();
        const bool has_abc = jobs[G.idx[0]].a != nullptr;
#ifdef MASP_ENQ_TIMING
        const double t_wait = ms_since(tg0);
        const auto tg1 = std::chrono::steady_clock::now();
#endif
        // staging layout: [np][nv] witness | (a | b | c each [np][nrows]) | [np][16] r,s limbs
        const size_t w_bytes = 32 * nv * np, abc_bytes = has_abc ? 3 * 32 * (size_t)C.nrows * np : 0, rs_bytes = 64 * np;
        const size_t in_bytes = 32 * (size_t)C.n_inputs * np;  // the public inputs once more, packed (see below)
        bool ok = true;
        if ((rc = sl.stage_reserve(w_bytes + abc_bytes + rs_bytes + in_bytes)) || (rc = sl.w.reserve(nv * np)) || (rc = sl.inp.reserve((size_t)C.n_inputs * np)) ||
            (rc = sl.abc.reserve(has_abc ? 3 * (size_t)C.nrows * np : 1)) || (rc = sl.reserve_batch(np))) {
            result = fail_shared(ctx, rc);
            ok = false;
        }
        if (ok) {
            uint8_t* hs = sl.h_stage;
            // aux vectors that already sit in page-locked memory go to the device straight from there
            std::vector<char> direct(np, 0);
            bool any_staged = false;
            for (size_t p = 0; p < np; ++p) {
                hipPointerAttribute_t attr;
                direct[p] = hipPointerGetAttributes(&attr, jobs[G.idx[p]].aux) == hipSuccess && attr.type == hipMemoryTypeHost;
                any_staged = any_staged || !direct[p];
            }
            (void)hipGetLastError();  // an unregistered pointer makes hipPointerGetAttributes report an error: expected
            for (size_t p = 0; p < np; ++p) {
                const masp_hip_job& J = jobs[G.idx[p]];
                memcpy(hs + 32 * nv * p, J.inputs, 32 * (size_t)C.n_inputs);
                memcpy(hs + w_bytes + abc_bytes + rs_bytes + 32 * (size_t)C.n_inputs * p, J.inputs, 32 * (size_t)C.n_inputs);
                if (!direct[p]) memcpy(hs + 32 * nv * p + 32 * (size_t)C.n_inputs, J.aux, 32 * (size_t)C.n_aux);
                if (has_abc) {
                    const uint8_t* src[3] = {J.a, J.b, J.c};
                    for (int i = 0; i < 3; ++i) memcpy(hs + w_bytes + 32 * (size_t)C.nrows * (np * i + p), src[i], 32 * (size_t)C.nrows);
                }
                memcpy(hs + w_bytes + abc_bytes + 64 * p, J.r, 32);
                memcpy(hs + w_bytes + abc_bytes + 64 * p + 32, J.s, 32);
            }
            hipStream_t s = sl.stream;
#if MASP_UPLOAD_CHAIN
            // (see masp_hip_ctx::upload_tail; the lock covers wait + enqueue + record so that the chain has one order.)  Not with
            // lone_proof_graph: the tail is an event on ANOTHER slot's stream, and while that stream is being captured into a launch graph the
            // runtime refuses the wait ("dependency created on uncaptured work in another stream") although the event was recorded before the
            // capture began — 38 of 960 calls of three threads failed that way (profiles/r06_lone_graph_upload_chain_capture_isolation.txt)
            const bool chain = ctx->opt.lone_proof_graph == 0;
            std::unique_lock<std::mutex> upload_turn(ctx->upload_mu, std::defer_lock);
            if (chain) upload_turn.lock();
            if (chain && ctx->upload_tail && ctx->upload_tail != sl.ev_uploaded) ok = hipStreamWaitEvent(s, ctx->upload_tail, 0) == hipSuccess;
#endif
            if (!ok) {
            } else if (any_staged) {
                ok = hipMemcpyAsync(sl.w.p, hs, w_bytes, hipMemcpyHostToDevice, s) == hipSuccess;
            } else {
                // only the public inputs of each proof come from staging: ONE packed copy to the device and a strided copy on the
                // device instead of np small host-to-device copies (a strided host-to-device copy is the runtime's slow path)
                ok = hipMemcpyAsync(sl.inp.p, hs + w_bytes + abc_bytes + rs_bytes, in_bytes, hipMemcpyHostToDevice, s) == hipSuccess &&
                     hipMemcpy2DAsync(sl.w.p, 32 * nv, sl.inp.p, 32 * (size_t)C.n_inputs, 32 * (size_t)C.n_inputs, np, hipMemcpyDeviceToDevice, s) ==
                         hipSuccess;
            }
            for (size_t p = 0; p < np && ok; ++p)
                if (direct[p])
                    ok = hipMemcpyAsync(sl.w.p + nv * p + C.n_inputs, jobs[G.idx[p]].aux, 32 * (size_t)C.n_aux, hipMemcpyHostToDevice, s) ==
                         hipSuccess;
            ok = ok && (!has_abc || hipMemcpyAsync(sl.abc.p, hs + w_bytes, abc_bytes, hipMemcpyHostToDevice, s) == hipSuccess) &&
                 hipMemcpyAsync(sl.rs.p, hs + w_bytes + abc_bytes, rs_bytes, hipMemcpyHostToDevice, s) == hipSuccess;
#if MASP_UPLOAD_CHAIN
            if (chain && ok && hipEventRecord(sl.ev_uploaded, s) == hipSuccess) ctx->upload_tail = sl.ev_uploaded;
#endif
            if (!ok) {
                last_hip_error() = std::string("H2D copy failed: ") + hipGetErrorString(hipGetLastError());
                result = fail_shared(ctx, MASP_HIP_E_HIP);
            }
        }
        if (ok) {
            const Fr* abc[3] = {nullptr, nullptr, nullptr};
            if (has_abc)
                for (int i = 0; i < 3; ++i) abc[i] = sl.abc.p + (size_t)C.nrows * np * i;
            if (hipMemsetAsync(sl.flags.p, 0, sizeof(int), sl.stream) != hipSuccess) {
                last_hip_error() = "memset failed";
                result = fail_shared(ctx, MASP_HIP_E_HIP);
                ok = false;
            } else {
#ifdef MASP_ENQ_TIMING   // measurement builds only (tools/jobs): how long the host takes to enqueue one group
                const auto tq0 = std::chrono::steady_clock::now();
#endif
                if ((rc = enqueue_proofs_graphed(sl, C, (uint32_t)np, sl.w.p, nv, abc, sl.rs.p, sl.proof.p, jobs[G.idx[0]].aux_form == MASP_HIP_AUX_MONTGOMERY))) {
                    result = fail_shared(ctx, rc);
                    ok = false;
                }
#ifdef MASP_ENQ_TIMING
                fprintf(stderr, "[enq] %zu proofs: %.3f ms\n", np, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq0).count());
#endif
            }
        }
        if (ok) {
            hipStream_t s = sl.stream;
            ok = hipMemcpyAsync(sl.h_proof, sl.proof.p, 192 * np, hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipMemcpyAsync(sl.h_flags, sl.flags.p, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipEventRecord(sl.done, s) == hipSuccess;
            if (!ok) {
                last_hip_error() = "D2H copy failed";
                result = fail_shared(ctx, MASP_HIP_E_HIP);
            }
        }
        if (!ok) {
            for (int i = 0; i < Slot::N_AUX; ++i) hipStreamSynchronize(sl.aux[i]);
            hipStreamSynchronize(sl.stream);
            slot_release(ctx, si);
            break;
        }
        owned.push_back({si, gi});
#ifdef MASP_ENQ_TIMING
        fprintf(stderr, "[grp] call %p t=%8.2f ms group %zu circuit %u np %zu slot %zu: waited %.2f ms for the slot, staged + enqueued in %.2f ms\n", (void*)jobs,
                ms_since(tcall), gi, jobs[G.idx[0]].circuit, np, si, t_wait, ms_since(tg1));
#endif
    }
    while (!owned.empty()) slot_release(ctx, retire_oldest());
#ifdef MASP_ENQ_TIMING
    fprintf(stderr, "[call] %p done after %.2f ms (%zu jobs)\n", (void*)jobs, ms_since(tcall), n);
#endif
    if (result == MASP_HIP_OK) ctx->proofs_done += n;
    // a launch the runtime refused on this thread (MASP_LAUNCH keeps the first: kernel, file:line, HIP's text)
    const int launch_rc = launch_status();
    if (launch_rc && result == MASP_HIP_OK) result = fail_shared(ctx, launch_rc);
    if (hipGetLastError() != hipSuccess && result == MASP_HIP_OK) {
        last_hip_error() = "kernel launch failed";
        result = fail_shared(ctx, MASP_HIP_E_HIP);
    }
    return result;
}

int masp_hip_prove(masp_hip_ctx* ctx, uint32_t slot, const uint8_t* inputs, const uint8_t* aux, const uint8_t* a, const uint8_t* b,
                   const uint8_t* c, const uint8_t r[32], const uint8_t s[32], uint8_t proof_out[192]) {
    if (!r || !s || !proof_out) return MASP_HIP_E_INVALID_ARG;
    masp_hip_job J;
    J.circuit = slot;
    J.inputs = inputs;
    J.aux = aux;
    J.a = a;
    J.b = b;
    J.c = c;
    J.aux_form = MASP_HIP_AUX_CANONICAL;
    J.reserved = 0;
    memcpy(J.r, r, 32);
    memcpy(J.s, s, 32);
    uint8_t tmp[192];
    int rc = masp_hip_prove_batch(ctx, 1, &J, tmp);
    if (rc == MASP_HIP_OK) memcpy(proof_out, tmp, 192);
    return rc;
}

}  // extern "C"

// ---- building blocks ----------------------------------------------------------------------------
// np MSMs over ONE base set (how a batch of proofs uses the engine: gridDim.y = np, one launch per stage):
// scalars np x n x 32 B, out np x BYTES uncompressed.  window_bits 0 = the engine's own choice for n.
template <class O, int BYTES>
static int msm_multi(masp_hip_ctx* ctx, const uint8_t* bases, size_t n, const uint8_t* scalars, size_t np, int window_bits, uint8_t* out) {
    if (!ctx || !out || !np || np > 256 || !n || !bases || !scalars || n > (1u << 22) || window_bits < 0 || window_bits == 1 || window_bits > 16)
        return MASP_HIP_E_INVALID_ARG;
    if (!ctx->children.empty()) ctx = ctx->children[0];
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    hipStream_t s = ctx->main_stream;
    for (size_t i = 0; i < n * np; ++i)
        if (!rs_in_range(scalars + 32 * i)) return MASP_HIP_E_SCALAR_RANGE;
    MsmBases<O, BYTES> B;
    MsmWorkspace<O> ws;
    ws.tree_levels = ctx->opt.bucket_tree_levels;
    ws.tree_sub = (uint32_t)ctx->opt.bucket_tree_sub_batch;
    ws.tree.own.limit = (size_t)ctx->opt.bucket_tree_scratch_mb << 20;
    // (what lack of tree scratch does to this call's own workspace is reported like the slots': masp_hip_ctx_get_options)
    struct Report {
        masp_hip_ctx* c;
        MsmWorkspace<O>& w;
        ~Report() {
            c->block_tree_fallbacks += w.tree_fallbacks;
            c->block_tree_sub = std::min<uint32_t>(c->block_tree_sub, w.tree_sub);
        }
    } report{ctx, ws};
    DevBuf<Xyzz<O>> res;
    DevBuf<uint8_t> d_out;
    int rc;
    if ((rc = B.load_host(bases, (uint32_t)n, s, 0xffffffffu, window_bits))) return fail(ctx, rc);
    if (B.import_status & (PT_BAD_FLAGS | PT_NOT_CANONICAL)) return MASP_HIP_E_PARAMS_FORMAT;
    if ((rc = ctx->tmp_scalars.upload((const Fr*)scalars, n * np, s)) || (rc = res.reserve(np)) || (rc = d_out.reserve(BYTES * np))) return fail(ctx, rc);
    if ((rc = msm_enqueue(s, B, ws, (const uint32_t*)ctx->tmp_scalars.p, n * 8, res.p, 1, (uint32_t)np))) return fail(ctx, rc);
    for (size_t p = 0; p < np; ++p) {
        if constexpr (BYTES == 96)
            launch_g1_export(s, res.p + p, d_out.p + BYTES * p);
        else
            launch_g2_export(s, res.p + p, d_out.p + BYTES * p);
    }
    std::vector<uint8_t> tmp(BYTES * np);
    if (hipMemcpyAsync(tmp.data(), d_out.p, tmp.size(), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        last_hip_error() = std::string("msm failed: ") + hipGetErrorString(hipGetLastError());
        return fail(ctx, MASP_HIP_E_HIP);
    }
    memcpy(out, tmp.data(), tmp.size());
    return MASP_HIP_OK;
}
template <class O, int BYTES, class X>
static int msm_block(masp_hip_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t* out, DevBuf<X>& res) {
    if (!ctx || !out || (n && (!bases || !scalars)) || n > (1u << 26)) return MASP_HIP_E_INVALID_ARG;
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    hipStream_t s = ctx->main_stream;
    for (size_t i = 0; i < n; ++i)
        if (!rs_in_range(scalars + 32 * i)) return MASP_HIP_E_SCALAR_RANGE;
    MsmBases<O, BYTES> B;
    MsmWorkspace<O> ws;
    int rc;
    if ((rc = B.load_host(bases, (uint32_t)n, s))) return fail(ctx, rc);
    if (B.import_status & (PT_BAD_FLAGS | PT_NOT_CANONICAL)) return MASP_HIP_E_PARAMS_FORMAT;
    if ((rc = ctx->tmp_scalars.upload((const Fr*)scalars, n, s)) || (rc = res.reserve(1)) || (rc = ctx->tmp_out.reserve(BYTES))) return fail(ctx, rc);
    if ((rc = msm_enqueue(s, B, ws, (const uint32_t*)ctx->tmp_scalars.p, 0, res.p, 1, 1))) return fail(ctx, rc);
    if constexpr (BYTES == 96)
        launch_g1_export(s, res.p, ctx->tmp_out.p);
    else
        launch_g2_export(s, res.p, ctx->tmp_out.p);
    uint8_t tmp[BYTES];
    if (hipMemcpyAsync(tmp, ctx->tmp_out.p, BYTES, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        last_hip_error() = std::string("msm failed: ") + hipGetErrorString(hipGetLastError());
        return fail(ctx, MASP_HIP_E_HIP);
    }
    memcpy(out, tmp, BYTES);
    return MASP_HIP_OK;
}
extern "C" {

// (a multi-device context runs the building blocks and the measurement hooks on its first device)
#define FIRST_DEVICE(ctx) ((ctx) && !(ctx)->children.empty() ? (ctx)->children[0] : (ctx))

int masp_hip_msm_g1(masp_hip_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[96]) {
    const ApiLaunchScope api_scope;
    if (!ctx) return MASP_HIP_E_INVALID_ARG;
    ctx = FIRST_DEVICE(ctx);
    return msm_block<FpOps, 96>(ctx, bases, scalars, n, out, ctx->tmp_g1);
}
int masp_hip_msm_g2(masp_hip_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, size_t n, uint8_t out[192]) {
    const ApiLaunchScope api_scope;
    if (!ctx) return MASP_HIP_E_INVALID_ARG;
    ctx = FIRST_DEVICE(ctx);
    return msm_block<Fp2Ops, 192>(ctx, bases, scalars, n, out, ctx->tmp_g2);
}

int masp_hip_msm_g1_multi(masp_hip_ctx* ctx, const uint8_t* bases, size_t n, const uint8_t* scalars, size_t np, int window_bits, uint8_t* out) {
    const ApiLaunchScope api_scope;
    return msm_multi<FpOps, 96>(ctx, bases, n, scalars, np, window_bits, out);
}
int masp_hip_msm_g2_multi(masp_hip_ctx* ctx, const uint8_t* bases, size_t n, const uint8_t* scalars, size_t np, int window_bits, uint8_t* out) {
    const ApiLaunchScope api_scope;
    return msm_multi<Fp2Ops, 192>(ctx, bases, n, scalars, np, window_bits, out);
}

int masp_hip_quotient_h(masp_hip_ctx* ctx, const uint8_t* a, const uint8_t* b, const uint8_t* c, size_t nrows, uint32_t logm, uint8_t* h_out) {
    const ApiLaunchScope api_scope;
    if (!ctx || !a || !b || !c || !h_out || logm == 0 || logm > 20 || nrows > ((size_t)1 << logm)) return MASP_HIP_E_INVALID_ARG;
    ctx = FIRST_DEVICE(ctx);
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    int rc;
    if ((rc = ensure_slots(ctx, 1))) return fail(ctx, rc);
    Slot& sl = *ctx->slots[0];
    NttDomain* D;
    if ((rc = get_domain(ctx, logm, &D))) return fail(ctx, rc);
    if ((rc = sl.w.reserve(3 * nrows))) return fail(ctx, rc);
    hipStream_t s = sl.stream;
    const uint8_t* src[3] = {a, b, c};
    const Fr* in[3];
    for (int i = 0; i < 3; ++i) {
        if (hipMemcpyAsync(sl.w.p + i * nrows, src[i], 32 * nrows, hipMemcpyHostToDevice, s) != hipSuccess) return fail(ctx, MASP_HIP_E_HIP);
        in[i] = sl.w.p + i * nrows;
    }
    if ((rc = enqueue_quotient(sl, *D, in, 0, (uint32_t)nrows, false, 1))) return fail(ctx, rc);
    size_t m = (size_t)1 << logm;
    if (hipMemcpyAsync(h_out, sl.h.p, 32 * (m - 1), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        last_hip_error() = std::string("quotient failed: ") + hipGetErrorString(hipGetLastError());
        return fail(ctx, MASP_HIP_E_HIP);
    }
    return MASP_HIP_OK;
}

int masp_hip_ntt(masp_hip_ctx* ctx, uint8_t* data, uint32_t logm, int inverse) {
    const ApiLaunchScope api_scope;
    if (!ctx || !data || logm == 0 || logm > 20) return MASP_HIP_E_INVALID_ARG;
    ctx = FIRST_DEVICE(ctx);
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    int rc;
    if ((rc = ensure_slots(ctx, 1))) return fail(ctx, rc);
    Slot& sl = *ctx->slots[0];
    NttDomain* D;
    if ((rc = get_domain(ctx, logm, &D))) return fail(ctx, rc);
    uint32_t m = 1u << logm;
    if ((rc = sl.w.reserve(m)) || (rc = sl.x0.reserve(m)) || (rc = sl.x1.reserve(m))) return fail(ctx, rc);
    hipStream_t s = sl.stream;
    if (hipMemcpyAsync(sl.w.p, data, 32 * (size_t)m, hipMemcpyHostToDevice, s) != hipSuccess) return fail(ctx, MASP_HIP_E_HIP);
    launch_ntt_load_bitrev(s, sl.w.p, (size_t)0, m, sl.x0.p, logm, 1);
    D->passes(s, sl.x0.p, inverse ? D->tw_inv.p : D->tw_fwd.p);
    if (inverse) {
        // 1/m scaling: coset_scale[0] = g^0 / m
        Fr* minv_tab = sl.x1.p;
        Fr minv = fe_inv(fr_from_u64_mont(m));
        launch_fr_powers(s, minv_tab, m, fe_one<FrCfg>(), minv, 0);
        launch_fr_scale(s, sl.x0.p, minv_tab, sl.x0.p, m, 1);
    }
    launch_fr_from_mont(s, sl.x0.p, sl.w.p, m);
    if (hipMemcpyAsync(data, sl.w.p, 32 * (size_t)m, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        last_hip_error() = std::string("ntt failed: ") + hipGetErrorString(hipGetLastError());
        return fail(ctx, MASP_HIP_E_HIP);
    }
    return MASP_HIP_OK;
}

// ---- measurement hooks ----------------------------------------------------------------------------
int masp_hip_batch_upload(masp_hip_ctx* ctx, size_t n, const masp_hip_job* jobs) {
    const ApiLaunchScope api_scope;
    if (!ctx || !n || !jobs) return -MASP_HIP_E_INVALID_ARG;
    ctx = FIRST_DEVICE(ctx);
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    std::unique_ptr<ResidentBatch> B(new ResidentBatch);
    B->n = n;
    size_t total = 0;
    for (size_t j = 0; j < n; ++j) {
        if (jobs[j].circuit >= MASP_HIP_MAX_CIRCUITS || !ctx->circ[jobs[j].circuit]) return -MASP_HIP_E_NOT_LOADED;
        if (!rs_in_range(jobs[j].r) || !rs_in_range(jobs[j].s)) return -MASP_HIP_E_SCALAR_RANGE;
        if (jobs[j].aux_form > MASP_HIP_AUX_MONTGOMERY || jobs[j].reserved != 0) return -MASP_HIP_E_INVALID_ARG;
    }
    for (uint32_t c = 0; c < MASP_HIP_MAX_CIRCUITS; ++c)
        for (size_t j = 0; j < n; ++j)
            if (jobs[j].circuit == c) {
                Circuit& C = *ctx->circ[c];
                B->circuit.push_back(c);
                B->order.push_back(j);
                B->w_off.push_back(total);
                total += (size_t)C.n_inputs + C.n_aux;
            }
    if (B->w.reserve(total) || B->rs.reserve(16 * n)) return -fail(ctx, MASP_HIP_E_HIP);
    hipStream_t s = ctx->main_stream;
    for (size_t k = 0; k < n; ++k) {
        const masp_hip_job& J = jobs[B->order[k]];
        Circuit& C = *ctx->circ[J.circuit];
        bool ok = hipMemcpyAsync(B->w.p + B->w_off[k], J.inputs, 32 * (size_t)C.n_inputs, hipMemcpyHostToDevice, s) == hipSuccess &&
                  hipMemcpyAsync(B->w.p + B->w_off[k] + C.n_inputs, J.aux, 32 * (size_t)C.n_aux, hipMemcpyHostToDevice, s) == hipSuccess &&
                  hipMemcpyAsync(B->rs.p + 16 * k, J.r, 32, hipMemcpyHostToDevice, s) == hipSuccess &&
                  hipMemcpyAsync(B->rs.p + 16 * k + 8, J.s, 32, hipMemcpyHostToDevice, s) == hipSuccess;
        // resident assignments are proved repeatedly and proving converts a Montgomery aux part in place: the resident copy is made
        // canonical once, here
        if (ok && J.aux_form == MASP_HIP_AUX_MONTGOMERY && C.n_aux)
            launch_fr_from_mont(s, B->w.p + B->w_off[k] + C.n_inputs, B->w.p + B->w_off[k] + C.n_inputs, C.n_aux);
        if (!ok || hipStreamSynchronize(s) != hipSuccess || launch_status() != MASP_HIP_OK) return -fail(ctx, MASP_HIP_E_HIP);
    }
    for (size_t k = 0; k < ctx->batches.size(); ++k)
        if (!ctx->batches[k]) {
            ctx->batches[k] = std::move(B);
            return (int)k;
        }
    ctx->batches.push_back(std::move(B));
    return (int)ctx->batches.size() - 1;
}

int masp_hip_batch_free(masp_hip_ctx* ctx, int handle) {
    ctx = FIRST_DEVICE(ctx);
    if (!ctx || handle < 0 || (size_t)handle >= ctx->batches.size()) return MASP_HIP_E_INVALID_ARG;
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    ctx->batches[handle].reset();
    return MASP_HIP_OK;
}

// Proves every resident job of `handle` `steps` times (step k with the blinding scalars rs[k][job] = r | s, 64 bytes per job,
// or with the uploaded ones when rs is NULL): the whole launch sequence of all steps is enqueued on the slots' streams
// without any host synchronisation in between.  proofs_out: steps x n x 192 bytes, job order inside every step.
int masp_hip_batch_prove_resident_steps(masp_hip_ctx* ctx, int handle, size_t steps, const uint8_t* rs, uint8_t* proofs_out, float* elapsed_ms) {
    const ApiLaunchScope api_scope;
    ctx = FIRST_DEVICE(ctx);
    if (!ctx || handle < 0 || (size_t)handle >= ctx->batches.size() || !ctx->batches[handle] || !proofs_out || !steps) return MASP_HIP_E_INVALID_ARG;
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    ResidentBatch& B = *ctx->batches[handle];
    if (rs)
        for (size_t i = 0; i < 2 * steps * B.n; ++i)
            if (!rs_in_range(rs + 32 * i)) return MASP_HIP_E_SCALAR_RANGE;
    // groups of consecutive same-circuit jobs (their assignments are contiguous: stride = n_vars)
    struct Group {
        size_t first, count;
    };
    std::vector<Group> groups;
    for (size_t j = 0; j < B.n;) {
        size_t k = j + 1;
        while (k < B.n && B.circuit[k] == B.circuit[j]) ++k;
        for (auto& eg : even_groups(k - j, ctx->batch_cap)) groups.push_back({j + eg.first, eg.second});
        j = k;
    }
    size_t ns = std::min<size_t>(groups.size() * steps, (size_t)ctx->n_slots);
    int rc;
    if ((rc = ensure_slots(ctx, ns))) return fail(ctx, rc);
    DevBuf<uint8_t> d_proofs;
    DevBuf<uint32_t> d_rs;
    if ((rc = d_proofs.reserve(192 * B.n * steps))) return fail(ctx, rc);
    hipStream_t ms = ctx->main_stream;
    if (rs) {  // step-major, STORAGE order of the jobs inside a step
        std::vector<uint8_t> tmp(64 * B.n * steps);
        for (size_t st = 0; st < steps; ++st)
            for (size_t k = 0; k < B.n; ++k) memcpy(&tmp[64 * (st * B.n + k)], rs + 64 * (st * B.n + B.order[k]), 64);
        if ((rc = d_rs.reserve(16 * B.n * steps))) return fail(ctx, rc);
        HIP_TRY(hipMemcpyAsync(d_rs.p, tmp.data(), tmp.size(), hipMemcpyHostToDevice, ms));
        HIP_TRY(hipStreamSynchronize(ms));
    }
    hipEvent_t ev_start, ev_stop;
    HIP_TRY(hipEventCreate(&ev_start));
    HIP_TRY(hipEventCreate(&ev_stop));
    for (size_t si = 0; si < ns; ++si) HIP_TRY(hipMemsetAsync(ctx->slots[si]->flags.p, 0, sizeof(int), ms));
    HIP_TRY(hipEventRecord(ev_start, ms));
    for (size_t si = 0; si < ns; ++si) HIP_TRY(hipStreamWaitEvent(ctx->slots[si]->stream, ev_start, 0));
    const Fr* none[3] = {nullptr, nullptr, nullptr};
    size_t turn = 0;
    for (size_t st = 0; st < steps; ++st)
        for (size_t gi = 0; gi < groups.size(); ++gi, ++turn) {
            const Group& G = groups[gi];
            Slot& sl = *ctx->slots[turn % ns];
            Circuit& C = *ctx->circ[B.circuit[G.first]];
            const uint32_t* grs = rs ? d_rs.p + 16 * (st * B.n + G.first) : B.rs.p + 16 * G.first;
            if ((rc = enqueue_proofs(sl, C, (uint32_t)G.count, B.w.p + B.w_off[G.first], (size_t)C.n_inputs + C.n_aux, none, grs,
                                     d_proofs.p + 192 * (st * B.n + G.first))))
                return fail(ctx, rc);
        }
    for (size_t si = 0; si < ns; ++si) {
        HIP_TRY(hipEventRecord(ctx->slots[si]->done, ctx->slots[si]->stream));
        HIP_TRY(hipStreamWaitEvent(ms, ctx->slots[si]->done, 0));
    }
    HIP_TRY(hipEventRecord(ev_stop, ms));
    std::vector<uint8_t> stored(192 * B.n * steps);
    HIP_TRY(hipMemcpyAsync(stored.data(), d_proofs.p, stored.size(), hipMemcpyDeviceToHost, ms));
    HIP_TRY(hipStreamSynchronize(ms));
    for (size_t st = 0; st < steps; ++st)
        for (size_t k = 0; k < B.n; ++k) memcpy(proofs_out + 192 * (st * B.n + B.order[k]), stored.data() + 192 * (st * B.n + k), 192);
    if (elapsed_ms) HIP_TRY(hipEventElapsedTime(elapsed_ms, ev_start, ev_stop));
    hipEventDestroy(ev_start);
    hipEventDestroy(ev_stop);
    int flags = 0;
    for (size_t si = 0; si < ns; ++si) {
        int f = 0;
        HIP_TRY(hipMemcpy(&f, ctx->slots[si]->flags.p, sizeof(int), hipMemcpyDeviceToHost));
        flags |= f;
    }
    return flags ? MASP_HIP_E_SCALAR_RANGE : MASP_HIP_OK;
}

int masp_hip_batch_prove_resident(masp_hip_ctx* ctx, int handle, uint8_t* proofs_out, float* elapsed_ms) {
    return masp_hip_batch_prove_resident_steps(ctx, handle, 1, nullptr, proofs_out, elapsed_ms);
}

int masp_hip_profile_enable(masp_hip_ctx* ctx, int on) {
    if (!ctx) return MASP_HIP_E_INVALID_ARG;
    ctx = FIRST_DEVICE(ctx);
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    ctx->profiling = on != 0;
    for (auto& sl : ctx->slots) {
        sl->profiling = ctx->profiling;
        sl->prof.reset();
    }
    return MASP_HIP_OK;
}

int masp_hip_profile_read(masp_hip_ctx* ctx, double* total_ms, uint64_t* launches, uint64_t* alg_bytes) {
    if (!ctx) return MASP_HIP_E_INVALID_ARG;
    ctx = FIRST_DEVICE(ctx);
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    double t = 0;
    uint64_t l = 0, b = 0;
    for (auto& sl : ctx->slots) {
        sl->prof.collect();
        t += sl->prof.total_ms;
        l += sl->prof.launches;
        b += sl->prof.alg_bytes;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = l;
    if (alg_bytes) *alg_bytes = b;
    return MASP_HIP_OK;
}

int masp_hip_profile_read_split(masp_hip_ctx* ctx, double ms[8]) {
    if (!ctx || !ms) return MASP_HIP_E_INVALID_ARG;
    ctx = FIRST_DEVICE(ctx);
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    for (int i = 0; i < 8; ++i) ms[i] = 0;
    for (auto& sl : ctx->slots) {
        sl->prof.collect();
        for (int i = 0; i < MsmProfile::PH_N; ++i) ms[i] += sl->prof.split_ms[i];
    }
    return MASP_HIP_OK;
}

int masp_hip_profile_read_lone(masp_hip_ctx* ctx, double ms[12]) {
    if (!ctx || !ms) return MASP_HIP_E_INVALID_ARG;
    ctx = FIRST_DEVICE(ctx);
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    for (int i = 0; i < Slot::N_LONE_MARKS; ++i) ms[i] = -1.0;
    for (auto& sl : ctx->slots) {
        if (!sl->lone_marked || !sl->ev_lone[0]) continue;
        for (int i = 0; i < Slot::N_LONE_MARKS; ++i) {
            float t = 0;
            if (sl->ev_lone[i] && hipEventElapsedTime(&t, sl->ev_lone[0], sl->ev_lone[i]) == hipSuccess) ms[i] = t;
        }
        (void)hipGetLastError();
        return MASP_HIP_OK;
    }
    return MASP_HIP_E_INVALID_ARG;   // no lone proof has run with profiling on
}

void* masp_hip_host_alloc(masp_hip_ctx* ctx, size_t bytes) {
    if (!ctx || !bytes) return nullptr;
    hipSetDevice(ctx->device);
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {  // portable: every device of a multi-device context may read it
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void masp_hip_host_free(masp_hip_ctx* ctx, void* ptr) {
    if (!ctx || !ptr) return;
    hipSetDevice(ctx->device);
    hipHostFree(ptr);
}

int masp_hip_sync(masp_hip_ctx* ctx) {
    if (!ctx) return MASP_HIP_E_INVALID_ARG;
    if (!ctx->children.empty()) {
        for (masp_hip_ctx* c : ctx->children)
            if (int rc = masp_hip_sync(c)) return rc;
        return MASP_HIP_OK;
    }
    hipSetDevice(ctx->device);
    return hipDeviceSynchronize() == hipSuccess ? MASP_HIP_OK : MASP_HIP_E_HIP;
}

int masp_hip_bench_msm(masp_hip_ctx* ctx, int handle, size_t job, int which, int iters, float* avg_ms, uint32_t* n_bases) {
    const ApiLaunchScope api_scope;
    ctx = FIRST_DEVICE(ctx);
    if (!ctx || handle < 0 || (size_t)handle >= ctx->batches.size() || !ctx->batches[handle] || which < 0 || which > 3 || iters <= 0 || !avg_ms)
        return MASP_HIP_E_INVALID_ARG;
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    ResidentBatch& B = *ctx->batches[handle];
    if (job >= B.n) return MASP_HIP_E_INVALID_ARG;
    for (size_t k = 0; k < B.n; ++k)
        if (B.order[k] == job) {
            job = k;
            break;
        }
    int rc;
    if ((rc = ensure_slots(ctx, 1))) return fail(ctx, rc);
    Slot& sl = *ctx->slots[0];
    Circuit& C = *ctx->circ[B.circuit[job]];
    const Fr* none[3] = {nullptr, nullptr, nullptr};
    // one full proof first so that sl.h / sl.sa / sl.sb hold this job's real scalars
    if ((rc = enqueue_proofs(sl, C, 1, B.w.p + B.w_off[job], 0, none, B.rs.p + 16 * job, sl.proof.p))) return fail(ctx, rc);
    HIP_TRY(hipStreamSynchronize(sl.stream));
    const BasesG1* bases[4] = {&C.h, &C.l, &C.a, &C.b1};
    const uint32_t* scal[4] = {(const uint32_t*)sl.h.p, (const uint32_t*)(B.w.p + B.w_off[job] + C.n_inputs), (const uint32_t*)sl.sa.p,
                               (const uint32_t*)sl.sb.p};
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, sl.stream));
    for (int it = 0; it < iters; ++it)
        if ((rc = msm_enqueue(sl.stream, *bases[which], sl.ws1, scal[which], 0, sl.res1.p + which, 4, 1))) return fail(ctx, rc);
    HIP_TRY(hipEventRecord(e1, sl.stream));
    HIP_TRY(hipStreamSynchronize(sl.stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    if (n_bases) *n_bases = bases[which]->n;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return MASP_HIP_OK;
}

}  // extern "C"

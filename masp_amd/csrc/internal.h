// Host-side types shared by the translation units of libmasp_hip (prover.hip: context, CRS loading, pipeline, C ABI;
// k_setup.hip: parameter generation).  No kernels here.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include "device/io.hpp"
#include "device/ntt_geom.h"
#include "launch.h"
#include "msm_host.h"
#include <atomic>
#include "util.h"

namespace masp {

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) dev_free(p);
        p = nullptr;
        cap = 0;
    }
    int reserve(size_t n) {
        if (n <= cap) return MASP_HIP_OK;
        release();
        HIP_TRY(dev_malloc(&p, sizeof(T) * std::max<size_t>(n, 1)));
        cap = n;
        return MASP_HIP_OK;
    }
    int upload(const T* h, size_t n, hipStream_t s) {
        int rc = reserve(n);
        if (rc) return rc;
        if (n) HIP_TRY(hipMemcpyAsync(p, h, sizeof(T) * n, hipMemcpyHostToDevice, s));
        return MASP_HIP_OK;
    }
};

// host-side Fr helpers (the device field code is __host__ __device__)
static inline Fr fr_from_u64_mont(uint64_t x) {
    Fr v = fe_zero<FrCfg>();
    v.v[0] = (uint32_t)x;
    v.v[1] = (uint32_t)(x >> 32);
    return fe_to_mont(v);
}
static inline Fr fr_const(const uint32_t* limbs) {
    Fr v;
    for (int i = 0; i < 8; ++i) v.v[i] = limbs[i];
    return v;
}

struct NttDomain {
    uint32_t logm = 0;
    size_t m = 0;
    DevBuf<Fr> tw_fwd, tw_inv, coset_scale, h_scale;
    Fr c_scale;  // 1 / (m (g^m - 1)), plain form
    int init(uint32_t logm_, hipStream_t s) {
        logm = logm_;
        m = (size_t)1 << logm;
        Fr omega = fr_const(FrCfg::ROOT_OF_UNITY);
        for (uint32_t i = logm; i < 32; ++i) omega = fe_sqr(omega);
        Fr omega_inv = fe_inv(omega);
        Fr minv = fe_inv(fr_from_u64_mont(m));
        Fr g = fr_const(FrCfg::GEN), ginv = fr_const(FrCfg::GEN_INV);
        uint32_t e[2] = {(uint32_t)m, (uint32_t)((uint64_t)m >> 32)};
        const Fr zinv = fe_inv(fe_sub(fe_pow(g, e, 2), fe_one<FrCfg>()));  // 1 / Z on the coset g H
        const Fr mzinv = fe_mul(minv, zinv);
        c_scale = fe_from_mont(mzinv);
        Fr one = fe_one<FrCfg>();
        size_t half = std::max<size_t>(m / 2, 1);
        int rc;
        if ((rc = tw_fwd.reserve(half)) || (rc = tw_inv.reserve(half)) || (rc = coset_scale.reserve(m)) || (rc = h_scale.reserve(m))) return rc;
        launch_fr_powers(s, tw_fwd.p, (uint32_t)half, omega, one, 0);
        launch_fr_powers(s, tw_inv.p, (uint32_t)half, omega_inv, one, 0);
        launch_fr_powers(s, coset_scale.p, (uint32_t)m, g, minv, 0);
        launch_fr_powers(s, h_scale.p, (uint32_t)m, ginv, mzinv, 1);  // g^-k / (m (g^m - 1)), plain form
        HIP_TRY(hipStreamSynchronize(s));
        return MASP_HIP_OK;
    }
    // np transforms at data + p * m
    void passes(hipStream_t s, Fr* data, const Fr* tw, uint32_t np = 1) const {
        for (uint32_t s0 = 0; s0 < logm;) {
            uint32_t nst = std::min<uint32_t>(NTT_LT, logm - s0);
            launch_ntt_pass(s, data, tw, logm, s0, nst, np);
            s0 += nst;
        }
    }
};

struct Circuit {
    uint32_t n_inputs = 0, n_aux = 0, n_constraints = 0, nrows = 0, logm = 0;
    size_t m = 0;
    DevBuf<uint32_t> rowptr[3], col[3];
    DevBuf<uint32_t> row_order[3];  // constraint rows by decreasing length: lanes of a wave get rows of similar length
    uint32_t n_long_rows[3] = {0, 0, 0};  // rows of >= R1CS_LONG_ROW terms (the head of row_order): a wave each
    DevBuf<Fr> coef[3];
    DevBuf<uint32_t> a_var, b_var;
    uint32_t na = 0, nbq = 0;
    DevBuf<VkDevice> vk;
    DevBuf<G1Xyzz> fb1;  // fixed-base tables of delta1, alpha1, beta1
    DevBuf<G2Xyzz> fb2;  // fixed-base table of delta2
    BasesG1 h, l, a, b1;
    // h and l as ONE base set (h's points, then l's) on h's windows: C only needs H + L, so a batch runs them as one MSM over
    // one bucket set — l's scalars then cost 16 window digits instead of 22 and its sort / bucket tails disappear
    BasesG1 hl;
    BasesG2 b2;
    // the same G2 points on narrow windows (128 buckets), for lone proofs only: B2's bucket tails are the longest chain of
    // a lone proof (every G2 addition is 40 dependent 384-bit products on one lane), and with 128 instead of 2 048 buckets
    // the gather / weighted-sum levels nearly vanish for 45 % more (chip-filling) accumulation work; n = 0: not built
    BasesG2 b2_lone;
    // alpha_g1, beta_g1, delta_g1 and every point of the a / b_g1 queries lie in the prime-order subgroup (tested when the
    // circuit is loaded): s*A and r*B1 may then go through the endomorphism (k_groth16_var_mul).  A CRS with a curve point
    // outside the subgroup — the reference reads it unchecked — keeps the plain double-and-add, whose bytes are the reference's
    bool g1_endo = false;
    NttDomain* dom = nullptr;
};

// scratch for one batch of up to `ctx->batch_cap` proofs of the same circuit + a stream.  Every stage is ONE launch for the
// whole batch (gridDim.y = proofs); a few slots let the stages of different batches overlap on the device.
struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    MsmWorkspace<FpOps> ws1;
    MsmWorkspace<Fp2Ops> ws2;
    // lone-proof mode (a batch too small to fill the chip): the five MSMs run side by side on their own streams, each
    // with its own workspace, next to the quotient pipeline on the main stream
    static constexpr int N_AUX = 4;
    hipStream_t aux[N_AUX] = {nullptr, nullptr, nullptr, nullptr};
    bool owns_aux = true;   // false: the side streams are slot 1's (masp_hip_ctx::slot_streams)
    hipEvent_t ev_fork = nullptr, ev_sort_b = nullptr, ev_fixed = nullptr, ev_join[N_AUX] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_uploaded = nullptr;   // behind the host-to-device copies of this slot's batch (masp_hip_ctx::upload_tail)
    MsmWorkspace<FpOps> ws_l, ws_a, ws_b;
    DevBuf<Fr> w, inp, abc, wm, ev[3], x0, x1, h, hl, sa, sb;
    DevBuf<G1Xyzz> res1, asm1;  // asm1: the six G1 pieces of the assembly per proof, asm2: s*delta2
    DevBuf<G2Xyzz> res2, asm2;
    DevBuf<uint32_t> rs;
    DevBuf<uint8_t> proof;
    DevBuf<int> flags;
    MsmProfile prof;             // live HIP-event timing of k_msm_accumulate<G1> (bench roofline leg)
    bool profiling = false;
    // where a LONE proof's chains end, without a profiler in the way (rocprofv3 adds ~10 us per launch, which reorders five chains of
    // ~60 launches each): timing events recorded behind the stages of the last lone proof while profiling is on
    // (masp_hip_profile_read_lone).  Created on first use.
    static constexpr int N_LONE_MARKS = 12;
    hipEvent_t ev_lone[N_LONE_MARKS] = {};
    bool lone_marked = false;
    void lone_mark(hipStream_t st, int id) {
        if (!profiling) return;
        if (!ev_lone[id] && hipEventCreate(&ev_lone[id]) != hipSuccess) return;
        (void)hipEventRecord(ev_lone[id], st);
        if (id == N_LONE_MARKS - 1) lone_marked = true;
    }
    uint32_t ntt_sub = 8;        // masp_hip_options::ntt_sub_batch of the owning context (0 = whole batch)
    uint8_t* h_stage = nullptr;  // pinned staging for the assignment
    size_t h_stage_cap = 0;
    uint8_t* h_proof = nullptr;  // pinned, batch_cap x 192
    size_t h_proof_cap = 0;
    int* h_flags = nullptr;      // pinned
    // Launch graphs of lone proofs (prover.hip: enqueue_proofs_graphed; opt-in).  The ~250 launches of one proof take the
    // host about as long to enqueue as the GPU takes to run them; captured once per (circuit, buffers, form) they replay as
    // one hipGraphLaunch.  A graph holds raw pointers into workspaces that grow on demand: `graph_epoch` is the value of
    // device_alloc_epoch() it was captured under, and any later allocation or release drops every graph of the slot.
    struct LoneGraph {
        const void *circuit, *w, *abc, *rs, *proof;
        size_t w_stride;
        uint32_t np;
        bool mont;
        int runs;                // ungraphed runs so far (the first one sizes every buffer)
        bool dead;               // capture failed once: not tried again
        hipGraphExec_t exec;
    };
    std::vector<LoneGraph> graphs;
    uint64_t graph_epoch = 0;
    bool lone_graph = false;     // masp_hip_options::lone_proof_graph
    std::atomic<uint64_t> graph_launches{0};
    void drop_graphs() {
        for (auto& g : graphs)
            if (g.exec) hipGraphExecDestroy(g.exec);
        graphs.clear();
    }
    ~Slot() {
        drop_graphs();
        if (stream) hipStreamDestroy(stream);
        if (done) hipEventDestroy(done);
        for (int i = 0; i < N_AUX; ++i) {
            if (aux[i] && owns_aux) hipStreamDestroy(aux[i]);
            if (ev_join[i]) hipEventDestroy(ev_join[i]);
        }
        for (hipEvent_t e : ev_lone)
            if (e) hipEventDestroy(e);
        if (ev_fork) hipEventDestroy(ev_fork);
        if (ev_uploaded) hipEventDestroy(ev_uploaded);
        if (ev_sort_b) hipEventDestroy(ev_sort_b);
        if (ev_fixed) hipEventDestroy(ev_fixed);
        if (h_stage) hipHostFree(h_stage);
        if (h_proof) hipHostFree(h_proof);
        if (h_flags) hipHostFree(h_flags);
    }
    // the options of the owning context that the slot's launch sequences depend on
    void configure(const masp_hip_options& o) {
        ntt_sub = (uint32_t)o.ntt_sub_batch;
        ws1.tree_levels = o.bucket_tree_levels;
        ws2.tree_levels = o.bucket_tree_levels_g2;
        ws1.tree_levels_shared = o.bucket_tree_levels_g2;  // B2 is reduced from B1's sort (ws1.sort): padded if either wants a tree
        ws1.tree_sub = ws2.tree_sub = (uint32_t)o.bucket_tree_sub_batch;
        ws2.tree.arena = &ws1.tree.own;   // the G1 and G2 MSMs of a batch follow each other on the slot's stream: one tree arena
        ws1.tree.own.limit = (size_t)o.bucket_tree_scratch_mb << 20;
        lone_graph = o.lone_proof_graph > 0;
    }
    // `streams`: this slot's five streams (main, then the four side streams), created by the context when IT was created (see
    // masp_hip_ctx::slot_streams); nullptr: the slot creates its own
    int init(const hipStream_t* streams = nullptr, bool side_streams_are_mine = true) {
        owns_aux = side_streams_are_mine;
        if (streams) {
            stream = streams[0];
            for (int i = 0; i < N_AUX; ++i) aux[i] = streams[1 + i];
        } else {
            HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        }
        HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_uploaded, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_sort_b, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_fixed, hipEventDisableTiming));
        // (the four side streams are created WITH the slot.  Creating them at the slot's first lone proof instead — so that a batch prover
        // holds one stream per slot — was measured in round 6: batch throughput equal, lone proofs 4.1 - 4.2 instead of 3.6 - 3.7 ms inside
        // bench.py, equal in the standalone tool; not understood, not kept: profiles/r06_second_context_root_cause.txt.  With the default
        // 3 slots a context's 16 streams cover the default 16 hardware queues exactly once.)
        for (int i = 0; i < N_AUX; ++i) {
            if (!aux[i]) HIP_TRY(hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
        }
        HIP_TRY(hipHostMalloc(&h_flags, sizeof(int)));
        int rc;
        if ((rc = flags.reserve(1))) return rc;
        return reserve_batch(1);
    }
    int reserve_batch(size_t np) {
        int rc;
        if ((rc = res1.reserve(4 * np)) || (rc = res2.reserve(np)) || (rc = asm1.reserve(6 * np)) || (rc = asm2.reserve(np)) || (rc = rs.reserve(16 * np)) || (rc = proof.reserve(192 * np))) return rc;
        if (np > h_proof_cap) {
            if (h_proof) hipHostFree(h_proof);
            h_proof = nullptr;
            HIP_TRY(hipHostMalloc(&h_proof, 192 * np));
            h_proof_cap = np;
        }
        return MASP_HIP_OK;
    }
    int stage_reserve(size_t bytes) {
        if (bytes <= h_stage_cap) return MASP_HIP_OK;
        if (h_stage) hipHostFree(h_stage);
        h_stage = nullptr;
        HIP_TRY(hipHostMalloc(&h_stage, bytes));
        h_stage_cap = bytes;
        return MASP_HIP_OK;
    }
};

struct ResidentBatch {
    size_t n = 0;
    std::vector<uint32_t> circuit;  // in STORAGE order: jobs are stored grouped by circuit so that batches are strided
    std::vector<size_t> order;      // storage position -> caller's job index
    std::vector<size_t> w_off;      // element offsets into w
    DevBuf<Fr> w;
    DevBuf<uint32_t> rs;        // n x 16
};

}  // namespace masp

using masp::Circuit;
using masp::DevBuf;
using masp::Fr;
using masp::G1Affine;
using masp::G1Xyzz;
using masp::G2Affine;
using masp::G2Xyzz;
using masp::NttDomain;
using masp::ResidentBatch;
using masp::Slot;

// Locking: `mu` is held SHARED by masp_hip_prove_batch (any number of host threads prove concurrently, each batch on
// its own slot = stream + scratch) and EXCLUSIVE by everything that changes circuits or uses the shared scratch.  The
// slot pool has its own small lock (`slot_mu`); `err` is read and written under it.  `slots` / `slot_busy` are reserved
// to MAX_SLOTS at creation and never reallocate, so a prover thread may keep indexing them while another one adds a slot.
struct masp_hip_ctx {
    static constexpr size_t MAX_SLOTS = 64;
    int device = 0;
    // multi-device front (masp_hip_ctx_create_multi): one full context per device; this object only shards and forwards
    std::vector<masp_hip_ctx*> children;
    masp_hip_options opt{};   // resolved at creation (prover.hip: resolve_options); the library reads no environment
    int n_slots = 4;          // = opt.slots
    size_t batch_cap = 256;   // = opt.batch_cap
    std::shared_mutex mu;
    mutable std::mutex slot_mu;
    std::condition_variable slot_cv;
    std::vector<char> slot_busy;
    std::string err;
    hipStream_t main_stream = nullptr;
    std::unique_ptr<Circuit> circ[MASP_HIP_MAX_CIRCUITS];
    std::map<uint32_t, std::unique_ptr<NttDomain>> domains;
    std::vector<std::unique_ptr<Slot>> slots;
    std::vector<std::unique_ptr<ResidentBatch>> batches;
    bool profiling = false;
    std::atomic<uint64_t> proofs_done{0};   // proofs this device context has written (masp_hip_ctx_device_proofs)
    // multi-device front: MASP_HIP_OK or the error that took device context d out (prove_batch_multi), and the proofs it put back on the queue
    std::unique_ptr<std::atomic<int>[]> dev_status;
    std::atomic<uint64_t> requeued{0};
    // device context: the n-th masp_hip_prove_batch call from now fails before it touches the device (masp_hip_ctx_inject_fault; 0 = disarmed)
    std::atomic<uint32_t> fault_countdown{0};
    // Host-to-device copies of concurrent masp_hip_prove_batch calls go ONE BATCH AFTER THE OTHER: a batch's copies wait for the event
    // behind the previous batch's copies (whatever slot that was).  Three calls that start together (the beginning of a job list, of a
    // timed region) otherwise share the link, and none of them can start computing before all 3 x 820 MB have crossed it.
    // The streams of ALL slots are created with the context, slot by slot (create_single), and the main streams are then MEASURED and, where
    // two share a hardware queue, replaced (separate_main_streams): the runtime hands hardware queues to streams in creation order, so the main streams — the ones that carry batches — get
    // queues of their own whatever the process created before (round 6: with 21 streams created slot by slot over 16 queues, a later
    // context of a process could find two of its main streams on one queue: -3.5 % at 4 slots).  [slot][0] main, [slot][1..4] side streams;
    // a slot takes its five when it is created and owns them from then on (taken[slot]).
    // Slots 0 and 1 have side streams of their own; from slot 2 on a slot uses slot 1's (a lone proof takes the first free slot, so three lone
    // proofs must be in flight at once before two of them share side streams — they then interleave on them, each behind its own events).  A
    // default context so has 1 + 4 + 8 + 2 = 15 streams: one hardware queue each of the default 16, nothing shares.  (With
    // masp_hip_options::lone_proof_graph every slot keeps its own: a stream that is being captured cannot take another thread's launches.)
    std::vector<std::array<hipStream_t, 5>> slot_streams;
    // ... and two streams for the batch verifier's keys (masp_hip_vk_prepare hands them out in turn; a key does not own its stream): a
    // verification that shared a hardware queue with a slot's main stream waited behind a 185 ms batch (end to end -10 %)
    hipStream_t vk_streams[2] = {nullptr, nullptr};
    std::atomic<unsigned> vk_next{0};
    int main_streams_concurrent = 0;   // of the 1 + slots streams that carry batches, how many ran at once when the context was created
    std::vector<char> slot_streams_taken;
    std::mutex upload_mu;
    hipEvent_t upload_tail = nullptr;   // an event of some slot of this context (slots live as long as the context)
    // the building-block MSM entry points (masp_hip_msm_g1_multi ...) run on a workspace of their own: what lack of tree scratch did there
    std::atomic<uint64_t> block_tree_fallbacks{0};
    std::atomic<uint32_t> block_tree_sub{0xffffffffu};
    // scratch for the building-block entry points
    DevBuf<Fr> tmp_scalars;
    DevBuf<uint8_t> tmp_out;
    DevBuf<G1Xyzz> tmp_g1;
    DevBuf<G2Xyzz> tmp_g2;
    // fixed-base tables of the standard generators (parameter generation)
    DevBuf<G1Affine> fb_g1;
    DevBuf<G2Affine> fb_g2;
};

namespace masp {

static inline int fail(masp_hip_ctx* ctx, int rc) {
    if (rc == MASP_HIP_E_HIP) {
        std::lock_guard<std::mutex> g(ctx->slot_mu);
        ctx->err = last_hip_error();
    }
    return rc;
}

static inline int get_domain(masp_hip_ctx* ctx, uint32_t logm, NttDomain** out) {
    auto it = ctx->domains.find(logm);
    if (it == ctx->domains.end()) {
        std::unique_ptr<NttDomain> d(new NttDomain);
        int rc = d->init(logm, ctx->main_stream);
        if (rc) return rc;
        it = ctx->domains.emplace(logm, std::move(d)).first;
    }
    *out = it->second.get();
    return MASP_HIP_OK;
}

static inline uint32_t log2_ceil(uint32_t n) {
    uint32_t k = 0;
    while ((1ull << k) < n) ++k;
    return k;
}

}  // namespace masp

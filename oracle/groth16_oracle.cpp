// ORACLE — test infrastructure only.  Nothing under masp_amd/ may include, link or call this
// file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
// as the checker / the reported CPU baseline.
//
// PARITY STATUS: *parity unpinned at the proof-bytes boundary*.  The reference holds no golden
// Groth16 proof (no test calls create_random_proof) and its prover lives in an
// un-vendored dependency that cannot be built here (no Rust toolchain):
//   nam-bellperson 0.26.6-nam.1  (/root/reference/Cargo.lock:1355-1383), called at
//   /root/reference/masp_proofs/src/sapling/prover.rs:117,202,252.
// This file restates that dependency's published algorithm:
//   * generate_parameters  — bellperson groth16::generator (used at masp_proofs/benches/sapling.rs:24-36)
//   * create_proof         — bellperson groth16::prover::create_proof  (ProvingAssignment eval,
//                            EvaluationDomain {ifft, coset_fft, mul/sub_assign, divide_by_z_on_coset,
//                            icoset_fft}, multiexp with window ceil(ln n), assembly)
//   * Proof::write         — masp_proofs/src/prover.rs:190-193 (48|96|48 compressed)
//   * verify_proof         — masp_proofs/src/sapling/prover.rs:148,266
// What pins it instead (tests/test_oracle_*.py): python big-integer cross-checks of every field
// and group operation, the standard generator encodings, Fr ROOT_OF_UNITY, the toxic-waste closed
// form (oracle 1, no NTT/MSM involved) equal byte-for-byte to create_proof, and the pairing
// verification equation (oracle 2).
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "curve.hpp"

namespace oracle {

#include "consts.inc"

const PairingConsts& pairing_consts() {
    static PairingConsts k = [] {
        PairingConsts c;
        Fp12 w = Fp12::zero();
        w.c1.c0 = Fp2::one();
        Fp12 w2 = w * w, w3 = w2 * w;
        c.w2_inv = w2.inv();
        c.w3_inv = w3.inv();
        c.final_exp.assign(FINAL_EXP_LIMBS, FINAL_EXP_LIMBS + sizeof(FINAL_EXP_LIMBS) / 8);
        return c;
    }();
    return k;
}

// ------------------------------------------------------------------------------------------------
static int g_threads = 0;
static int nthreads() {
    if (g_threads > 0) return g_threads;
    unsigned n = std::thread::hardware_concurrency();
    return n ? (int)n : 1;
}
// Persistent worker pool (bellperson's CPU path runs on a global rayon pool; spawning threads per NTT
// stage would dominate on many-core hosts).  grain = minimum items per task.
class Pool {
  public:
    static Pool& get() {
        static Pool* p = new Pool;  // leaked on purpose: workers block on its condition variable at exit
        return *p;
    }
    void run(size_t n, size_t grain, const std::function<void(size_t, size_t, int)>& fn) {
        int T = nthreads();
        size_t maxT = grain ? (n + grain - 1) / grain : n;
        if ((size_t)T > maxT) T = (int)std::max<size_t>(maxT, 1);
        if (T <= 1 || n == 0) {
            fn(0, n, 0);
            return;
        }
        std::unique_lock<std::mutex> api(api_mu_);  // one parallel region at a time
        ensure_workers(T - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            parts_ = T;
            next_ = 1;  // part 0 runs on the caller
            pending_ = T - 1;
            ++gen_;
        }
        cv_.notify_all();
        size_t chunk = (n + T - 1) / T;
        fn(0, std::min(n, chunk), 0);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

  private:
    void ensure_workers(int want) {
        while ((int)workers_.size() < want) {
            int id = (int)workers_.size();
            workers_.emplace_back([this, id] { loop(id); });
            workers_.back().detach();
        }
    }
    void loop(int) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return gen_ != seen && next_ < parts_; });
            if (gen_ == seen) continue;
            while (next_ < parts_) {
                int part = next_++;
                const std::function<void(size_t, size_t, int)>* fn = fn_;
                size_t n = n_;
                int T = parts_;
                lk.unlock();
                size_t chunk = (n + T - 1) / T;
                size_t lo = std::min(n, chunk * part), hi = std::min(n, lo + chunk);
                (*fn)(lo, hi, part);
                lk.lock();
                if (--pending_ == 0) done_cv_.notify_all();
            }
            seen = gen_;
        }
    }
    std::mutex api_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    const std::function<void(size_t, size_t, int)>* fn_ = nullptr;
    size_t n_ = 0;
    int parts_ = 0, next_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
};
static void parallel_for(size_t n, const std::function<void(size_t, size_t, int)>& fn, size_t grain = 1) {
    Pool::get().run(n, grain, fn);
}
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------
// R1CS as three CSR matrices over `n_constraints` rows; column v < n_inputs is Input(v), otherwise
// Aux(v - n_inputs).  Terms are merged per variable and carry non-zero coefficients only, so the
// structural pattern *is* bellperson's DensityTracker content.
struct CsrView {
    const uint32_t* rowptr;
    const uint32_t* col;
    const uint8_t* coef;  // 32 B little-endian canonical each
};
struct R1csView {
    uint32_t n_inputs, n_aux, n_constraints;
    CsrView m[3];
};

static Fr fr_from_le(const uint8_t* b) {
    Fr x;
    if (!Fr::from_bytes_le(x, b)) {
        fprintf(stderr, "oracle: non-canonical Fr\n");
        abort();
    }
    return x;
}

static uint32_t log2_ceil(uint32_t n) {
    uint32_t k = 0;
    while ((1ull << k) < n) ++k;
    return k;
}
// ROOT_OF_UNITY = 7^((r-1)/2^32)  (order 2^32; value KAT-checked in tests against SURVEY.md §8c)
static Fr fr_root_of_unity() {
    uint64_t e[4];
    uint64_t one_[4] = {1, 0, 0, 0};
    sub_limbs<4>(e, Fr::ctx().p, one_);
    // >> 32
    for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 32) | (i < 3 ? e[i + 1] << 32 : 0);
    return Fr::from_u64(7).pow(e, 4);
}
static Fr fr_omega(uint32_t logm) {
    Fr w = fr_root_of_unity();
    for (uint32_t i = logm; i < 32; ++i) w = w.sqr();
    return w;
}

// ---- EvaluationDomain (bellperson domain.rs restated) -------------------------------------------
static void bitrev_permute(std::vector<Fr>& a, uint32_t logn) {
    const size_t n = a.size();
    if (logn == 0) return;
    parallel_for(n, [&](size_t lo, size_t hi, int) {
        for (size_t k = lo; k < hi; ++k) {
            uint32_t v = (uint32_t)k;   // reverse the 32 bits, keep the top logn
            v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
            v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
            v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
            v = __builtin_bswap32(v);
            const size_t rk = (size_t)(v >> (32 - logn));
            if (k < rk) std::swap(a[k], a[rk]);   // (every pair is swapped by exactly one of its two indices: no two threads touch the same pair)
        }
    }, 16384);
}
// omega^k for k < n / 2, computed once per (domain size, root) and kept: a proof runs seven transforms over the same two roots
// (bellperson recomputes nothing per butterfly either: its serial / parallel FFTs walk precomputed or incrementally updated
// twiddles).  Round 3 recomputed a twiddle vector per stage and divided twice per butterfly: 310 ms of NTT per proof on 16 cores
// against 204 ms of G1 MSM — not the MSM-dominated profile of a real CPU prover.
static const std::vector<Fr>& fft_twiddles(const Fr& omega, uint32_t logn) {
    struct Entry {
        uint32_t logn;
        Fr omega;
        std::vector<Fr> tw;
    };
    static std::mutex mu;
    static std::vector<std::unique_ptr<Entry>> cache;
    std::lock_guard<std::mutex> g(mu);
    for (auto& e : cache)
        if (e->logn == logn && e->omega == omega) return e->tw;
    std::unique_ptr<Entry> e(new Entry);
    e->logn = logn;
    e->omega = omega;
    const size_t half = logn ? (size_t)1 << (logn - 1) : 1;
    e->tw.resize(half);
    parallel_for(half, [&](size_t lo, size_t hi, int) {
        uint64_t k = lo;
        Fr u = omega.pow(&k, 1);
        for (size_t j = lo; j < hi; ++j) {
            e->tw[j] = u;
            u = u * omega;
        }
    }, 4096);
    cache.push_back(std::move(e));
    return cache.back()->tw;
}
static void fft(std::vector<Fr>& a, const Fr& omega, uint32_t logn) {
    const size_t n = a.size();
    bitrev_permute(a, logn);
    if (logn == 0) return;
    const std::vector<Fr>& tw = fft_twiddles(omega, logn);
    // the first stages stay inside blocks of 2^LB elements (256 KiB: a core's L2): a thread takes whole blocks through all of
    // them, no barrier and one pass over memory instead of LB
    const uint32_t LB = std::min<uint32_t>(logn, 13);
    parallel_for(n >> LB, [&](size_t lo, size_t hi, int) {
        for (size_t blk = lo; blk < hi; ++blk) {
            Fr* x = a.data() + (blk << LB);
            for (uint32_t s = 0; s < LB; ++s) {
                const size_t m = (size_t)1 << s;
                const uint32_t shift = logn - 1 - s;
                for (size_t k0 = 0; k0 < ((size_t)1 << LB); k0 += 2 * m)
                    for (size_t j = 0; j < m; ++j) {
                        const Fr y = x[k0 + j + m] * tw[j << shift];
                        x[k0 + j + m] = x[k0 + j] - y;
                        x[k0 + j] = x[k0 + j] + y;
                    }
            }
        }
    }, 1);
    for (uint32_t s = LB; s < logn; ++s) {
        const size_t m = (size_t)1 << s;
        const uint32_t shift = logn - 1 - s;   // stage s uses omega^(n / (2m) * j) = tw[j << shift]
        // parallel over the n / 2 butterflies of the stage (early stages: many small blocks, late stages: few large ones)
        parallel_for(n / 2, [&](size_t lo, size_t hi, int) {
            for (size_t t = lo; t < hi; ++t) {
                const size_t j = t & (m - 1), k = ((t >> s) << (s + 1)) + j;
                const Fr x = a[k + m] * tw[j << shift];
                a[k + m] = a[k] - x;
                a[k] = a[k] + x;
            }
        }, 8192);
    }
}
static void distribute_powers(std::vector<Fr>& a, const Fr& g) {
    parallel_for(a.size(), [&](size_t lo, size_t hi, int) {
        uint64_t e = lo;
        Fr u = g.pow(&e, 1);
        for (size_t i = lo; i < hi; ++i) {
            a[i] = a[i] * u;
            u = u * g;
        }
    }, 2048);
}
struct Domain {
    uint32_t logm;
    size_t m;
    Fr omega, omega_inv, minv, geninv;
    explicit Domain(uint32_t logm_) : logm(logm_), m((size_t)1 << logm_) {
        omega = fr_omega(logm);
        omega_inv = omega.inv();
        minv = Fr::from_u64(m).inv();
        geninv = Fr::from_u64(7).inv();
    }
    void ifft(std::vector<Fr>& a) const {
        fft(a, omega_inv, logm);
        parallel_for(a.size(), [&](size_t lo, size_t hi, int) {
            for (size_t i = lo; i < hi; ++i) a[i] = a[i] * minv;
        }, 2048);
    }
    void coset_fft(std::vector<Fr>& a) const {
        distribute_powers(a, Fr::from_u64(7));
        fft(a, omega, logm);
    }
    void icoset_fft(std::vector<Fr>& a) const {
        ifft(a);
        distribute_powers(a, geninv);
    }
    Fr z_on_coset_inv() const {
        uint64_t e = m;
        return (Fr::from_u64(7).pow(&e, 1) - Fr::one()).inv();
    }
};

// h = ((a*b - c) / Z) coefficients, bellperson's exact sequence.
static void quotient_h(std::vector<Fr>& a, std::vector<Fr>& b, std::vector<Fr>& c, uint32_t logm) {
    Domain d(logm);
    d.ifft(a);
    d.coset_fft(a);
    d.ifft(b);
    d.coset_fft(b);
    d.ifft(c);
    d.coset_fft(c);
    Fr zi = d.z_on_coset_inv();
    parallel_for(a.size(), [&](size_t lo, size_t hi, int) {
        for (size_t i = lo; i < hi; ++i) a[i] = (a[i] * b[i] - c[i]) * zi;
    }, 2048);
    d.icoset_fft(a);
    a.pop_back();  // truncate to m-1
}

// ---- multiexp (bellperson multiexp.rs CPU path restated) ----------------------------------------
// bases[i] pairs with scalars[i] (canonical 4x64 LE limbs).  Result independent of all choices.
// One multiexp split into per-window tasks so that several multiexps can share the worker pool, as
// bellperson runs its eight multiexp calls concurrently on the rayon pool.
struct WindowTask {
    std::function<void()> run;
};
template <class F>
struct MultiexpJob {
    const Affine<F>* bases;
    const uint64_t (*scalars)[4];
    size_t n;
    uint32_t c = 0, nwin = 0;
    std::vector<Jac<F>> wsum;
    MultiexpJob(const Affine<F>* b, const uint64_t (*s)[4], size_t n_) : bases(b), scalars(s), n(n_) {
        if (n == 0) return;
        c = n < 32 ? 3 : (uint32_t)std::ceil(std::log((double)n));
        nwin = (255 + c - 1) / c;
        wsum.assign(nwin, Jac<F>::infinity());
    }
    void window(uint32_t w) {
        uint32_t skip = w * c;
        std::vector<Jac<F>> buckets((size_t(1) << c) - 1, Jac<F>::infinity());
        Jac<F> acc = Jac<F>::infinity();
        for (size_t i = 0; i < n; ++i) {
            const uint64_t* e = scalars[i];
            if ((e[0] | e[1] | e[2] | e[3]) == 0) continue;
            if (e[0] == 1 && (e[1] | e[2] | e[3]) == 0) {
                if (w == 0) acc = acc.add_affine(bases[i]);
                continue;
            }
            uint32_t limb = skip / 64, off = skip % 64;
            uint64_t d = e[limb] >> off;
            if (off + c > 64 && limb + 1 < 4) d |= e[limb + 1] << (64 - off);
            d &= (1ull << c) - 1;
            if (d) buckets[d - 1] = buckets[d - 1].add_affine(bases[i]);
        }
        Jac<F> run = Jac<F>::infinity();
        for (size_t k = buckets.size(); k-- > 0;) {
            run = run.add(buckets[k]);
            acc = acc.add(run);
        }
        wsum[w] = acc;
    }
    void tasks(std::vector<WindowTask>& out) {
        for (uint32_t w = 0; w < nwin; ++w) out.push_back({[this, w] { window(w); }});
    }
    Jac<F> finish() const {
        Jac<F> total = Jac<F>::infinity();
        for (uint32_t w = nwin; w-- > 0;) {
            for (uint32_t k = 0; k < c; ++k) total = total.dbl();
            total = total.add(wsum[w]);
        }
        return total;
    }
};
static void run_tasks(std::vector<WindowTask>& tasks) {
    std::atomic<size_t> next(0);
    parallel_for((size_t)nthreads(), [&](size_t, size_t, int) {
        for (;;) {
            size_t t = next.fetch_add(1);
            if (t >= tasks.size()) break;
            tasks[t].run();
        }
    });
}
template <class F>
static Jac<F> multiexp(const Affine<F>* bases, const uint64_t (*scalars)[4], size_t n) {
    MultiexpJob<F> job(bases, scalars, n);
    std::vector<WindowTask> tasks;
    job.tasks(tasks);
    run_tasks(tasks);
    return job.finish();
}

// ---- QAP evaluation at tau (bellperson generator.rs semantics) --------------------
struct QapAtTau {
    uint32_t logm;
    size_t m;
    std::vector<Fr> lag;         // l_k(tau), k < n rows
    std::vector<Fr> at, bt, ct;  // per variable (inputs first, then aux)
    Fr z_tau;
};
static void qap_eval(const R1csView& cs, const Fr& tau, QapAtTau& q) {
    size_t nrows = (size_t)cs.n_constraints + cs.n_inputs;
    q.logm = log2_ceil((uint32_t)nrows);
    q.m = (size_t)1 << q.logm;
    Fr omega = fr_omega(q.logm);
    uint64_t e = q.m;
    q.z_tau = tau.pow(&e, 1) - Fr::one();
    Fr zm = q.z_tau * Fr::from_u64(q.m).inv();
    // l_k(tau) = (Z(tau)/m) * omega^k / (tau - omega^k)
    std::vector<Fr> wk(nrows), den(nrows), pre(nrows);
    Fr w = Fr::one();
    for (size_t k = 0; k < nrows; ++k) {
        wk[k] = w;
        den[k] = tau - w;
        w = w * omega;
    }
    Fr acc = Fr::one();
    for (size_t k = 0; k < nrows; ++k) {
        pre[k] = acc;
        acc = acc * den[k];
    }
    Fr inv = acc.inv();
    q.lag.resize(nrows);
    for (size_t k = nrows; k-- > 0;) {
        Fr di = inv * pre[k];
        inv = inv * den[k];
        q.lag[k] = zm * wk[k] * di;
    }
    size_t nv = (size_t)cs.n_inputs + cs.n_aux;
    q.at.assign(nv, Fr::zero());
    q.bt.assign(nv, Fr::zero());
    q.ct.assign(nv, Fr::zero());
    std::vector<Fr>* dst[3] = {&q.at, &q.bt, &q.ct};
    for (int mi = 0; mi < 3; ++mi) {
        const CsrView& M = cs.m[mi];
        for (uint32_t row = 0; row < cs.n_constraints; ++row)
            for (uint32_t t = M.rowptr[row]; t < M.rowptr[row + 1]; ++t) {
                Fr coef = fr_from_le(M.coef + 32 * (size_t)t);
                (*dst[mi])[M.col[t]] = (*dst[mi])[M.col[t]] + coef * q.lag[row];
            }
    }
    // the n_inputs extra rows  Input(i) * 0 = 0
    for (uint32_t i = 0; i < cs.n_inputs; ++i) q.at[i] = q.at[i] + q.lag[cs.n_constraints + i];
}

// ---- fixed-base scalar multiplication (window tables) -------------------------------------------
template <class F>
struct FixedBase {
    static constexpr int W = 8;
    std::vector<Affine<F>> table;  // [32][255]
    explicit FixedBase(const Affine<F>& g) {
        std::vector<Jac<F>> t(32 * 255);
        Jac<F> base = Jac<F>::from_affine(g);
        for (int w = 0; w < 32; ++w) {
            Jac<F> cur = base;
            for (int j = 0; j < 255; ++j) {
                t[w * 255 + j] = cur;
                cur = cur.add(base);
            }
            base = cur;  // 256 * previous base
        }
        batch_to_affine(t, table);
    }
    Jac<F> mul(const Fr& k) const {
        uint8_t b[32];
        k.to_bytes_le(b);
        Jac<F> r = Jac<F>::infinity();
        for (int w = 0; w < 32; ++w)
            if (b[w]) r = r.add_affine(table[w * 255 + b[w] - 1]);
        return r;
    }
    void mul_many(const std::vector<Fr>& ks, std::vector<Affine<F>>& out) const {
        std::vector<Jac<F>> tmp(ks.size());
        parallel_for(ks.size(), [&](size_t lo, size_t hi, int) {
            for (size_t i = lo; i < hi; ++i) tmp[i] = mul(ks[i]);
        });
        out.resize(ks.size());
        // batch-normalise in per-thread chunks
        parallel_for(ks.size(), [&](size_t lo, size_t hi, int) {
            std::vector<Jac<F>> part(tmp.begin() + lo, tmp.begin() + hi);
            std::vector<Affine<F>> aff;
            batch_to_affine(part, aff);
            for (size_t i = lo; i < hi; ++i) out[i] = aff[i - lo];
        });
    }
};
static const FixedBase<Fp>& fb_g1() {
    static FixedBase<Fp> t(g1_generator());
    return t;
}
static const FixedBase<Fp2>& fb_g2() {
    static FixedBase<Fp2> t(g2_generator());
    return t;
}

// ---- Parameters (bellman wire format) --------------------------------------------
struct Params {
    G1Affine alpha_g1, beta_g1, delta_g1;
    G2Affine beta_g2, gamma_g2, delta_g2;
    std::vector<G1Affine> ic, h, l, a, b_g1;
    std::vector<G2Affine> b_g2;
};
static void put_u32be(std::vector<uint8_t>& o, uint32_t v) {
    o.push_back(v >> 24);
    o.push_back(v >> 16);
    o.push_back(v >> 8);
    o.push_back(v);
}
static void put_g1(std::vector<uint8_t>& o, const G1Affine& p) {
    size_t k = o.size();
    o.resize(k + 96);
    g1_write_uncompressed(p, &o[k]);
}
static void put_g2(std::vector<uint8_t>& o, const G2Affine& p) {
    size_t k = o.size();
    o.resize(k + 192);
    g2_write_uncompressed(p, &o[k]);
}
static void params_write(const Params& P, std::vector<uint8_t>& o) {
    put_g1(o, P.alpha_g1);
    put_g1(o, P.beta_g1);
    put_g2(o, P.beta_g2);
    put_g2(o, P.gamma_g2);
    put_g1(o, P.delta_g1);
    put_g2(o, P.delta_g2);
    put_u32be(o, P.ic.size());
    for (auto& p : P.ic) put_g1(o, p);
    put_u32be(o, P.h.size());
    for (auto& p : P.h) put_g1(o, p);
    put_u32be(o, P.l.size());
    for (auto& p : P.l) put_g1(o, p);
    put_u32be(o, P.a.size());
    for (auto& p : P.a) put_g1(o, p);
    put_u32be(o, P.b_g1.size());
    for (auto& p : P.b_g1) put_g1(o, p);
    put_u32be(o, P.b_g2.size());
    for (auto& p : P.b_g2) put_g2(o, p);
}
struct Reader {
    const uint8_t* p;
    size_t left;
    bool ok = true;
    const uint8_t* take(size_t n) {
        if (left < n) {
            ok = false;
            return nullptr;
        }
        const uint8_t* r = p;
        p += n;
        left -= n;
        return r;
    }
    uint32_t u32be() {
        const uint8_t* b = take(4);
        if (!b) return 0;
        return ((uint32_t)b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3];
    }
    G1Affine g1() {
        G1Affine a = G1Affine::infinity();
        const uint8_t* b = take(96);
        if (b && !g1_read_uncompressed(a, b)) ok = false;
        return a;
    }
    G2Affine g2() {
        G2Affine a = G2Affine::infinity();
        const uint8_t* b = take(192);
        if (b && !g2_read_uncompressed(a, b)) ok = false;
        return a;
    }
};
// vk_only: stop after ic (what verify needs)
static bool params_read(Params& P, const uint8_t* buf, size_t len, bool vk_only) {
    Reader r{buf, len};
    P.alpha_g1 = r.g1();
    P.beta_g1 = r.g1();
    P.beta_g2 = r.g2();
    P.gamma_g2 = r.g2();
    P.delta_g1 = r.g1();
    P.delta_g2 = r.g2();
    uint32_t n = r.u32be();
    if (!r.ok || (size_t)n * 96 > r.left) return false;
    P.ic.resize(n);
    for (auto& p : P.ic) p = r.g1();
    if (vk_only) return r.ok;
    std::vector<G1Affine>* g1v[4] = {&P.h, &P.l, &P.a, &P.b_g1};
    for (auto* v : g1v) {
        n = r.u32be();
        if (!r.ok || (size_t)n * 96 > r.left) return false;
        v->resize(n);
        for (auto& p : *v) {
            p = r.g1();
            if (p.inf) return false;  // bellman rejects infinity inside query vectors
        }
    }
    n = r.u32be();
    if (!r.ok || (size_t)n * 192 > r.left) return false;
    P.b_g2.resize(n);
    for (auto& p : P.b_g2) {
        p = r.g2();
        if (p.inf) return false;
    }
    return r.ok;
}

struct Toxic {
    Fr tau, alpha, beta, gamma, delta;
};

static void generate_parameters(const R1csView& cs, const Toxic& tw, Params& P) {
    QapAtTau q;
    qap_eval(cs, tw.tau, q);
    Fr dinv = tw.delta.inv(), ginv = tw.gamma.inv();
    const FixedBase<Fp>& g1 = fb_g1();
    const FixedBase<Fp2>& g2 = fb_g2();
    P.alpha_g1 = g1.mul(tw.alpha).to_affine();
    P.beta_g1 = g1.mul(tw.beta).to_affine();
    P.delta_g1 = g1.mul(tw.delta).to_affine();
    P.beta_g2 = g2.mul(tw.beta).to_affine();
    P.gamma_g2 = g2.mul(tw.gamma).to_affine();
    P.delta_g2 = g2.mul(tw.delta).to_affine();
    // h[i] = tau^i * Z(tau) / delta, i in [0, m-2]
    std::vector<Fr> ks(q.m - 1);
    Fr cur = q.z_tau * dinv;
    for (size_t i = 0; i + 1 < q.m; ++i) {
        ks[i] = cur;
        cur = cur * tw.tau;
    }
    g1.mul_many(ks, P.h);
    // ic / l
    auto lc = [&](size_t v) { return tw.beta * q.at[v] + tw.alpha * q.bt[v] + q.ct[v]; };
    ks.resize(cs.n_inputs);
    for (uint32_t i = 0; i < cs.n_inputs; ++i) ks[i] = lc(i) * ginv;
    g1.mul_many(ks, P.ic);
    ks.resize(cs.n_aux);
    for (uint32_t j = 0; j < cs.n_aux; ++j) ks[j] = lc(cs.n_inputs + j) * dinv;
    g1.mul_many(ks, P.l);
    // a, b_g1, b_g2: inputs first then aux, identities filtered out
    size_t nv = (size_t)cs.n_inputs + cs.n_aux;
    ks.clear();
    for (size_t v = 0; v < nv; ++v)
        if (!q.at[v].is_zero()) ks.push_back(q.at[v]);
    g1.mul_many(ks, P.a);
    ks.clear();
    for (size_t v = 0; v < nv; ++v)
        if (!q.bt[v].is_zero()) ks.push_back(q.bt[v]);
    g1.mul_many(ks, P.b_g1);
    g2.mul_many(ks, P.b_g2);
}

// ---- ProvingAssignment (bellperson prover.rs restated) ------------------------------------------
struct Assignment {
    std::vector<Fr> in, aux;          // input_assignment (incl. ONE), aux_assignment
    std::vector<Fr> a, b, c;          // n rows
    std::vector<uint8_t> a_aux_density, b_input_density, b_aux_density;
};
static void synthesize_from_r1cs(const R1csView& cs, const uint8_t* inputs, const uint8_t* aux, Assignment& A) {
    A.in.resize(cs.n_inputs);
    A.aux.resize(cs.n_aux);
    for (uint32_t i = 0; i < cs.n_inputs; ++i) A.in[i] = fr_from_le(inputs + 32 * (size_t)i);
    for (uint32_t j = 0; j < cs.n_aux; ++j) A.aux[j] = fr_from_le(aux + 32 * (size_t)j);
    size_t nrows = (size_t)cs.n_constraints + cs.n_inputs;
    A.a.assign(nrows, Fr::zero());
    A.b.assign(nrows, Fr::zero());
    A.c.assign(nrows, Fr::zero());
    A.a_aux_density.assign(cs.n_aux, 0);
    A.b_input_density.assign(cs.n_inputs, 0);
    A.b_aux_density.assign(cs.n_aux, 0);
    std::vector<Fr>* dst[3] = {&A.a, &A.b, &A.c};
    for (int mi = 0; mi < 3; ++mi) {
        const CsrView& M = cs.m[mi];
        parallel_for(cs.n_constraints, [&](size_t lo, size_t hi, int) {
            for (size_t row = lo; row < hi; ++row) {
                Fr acc = Fr::zero();
                for (uint32_t t = M.rowptr[row]; t < M.rowptr[row + 1]; ++t) {
                    uint32_t v = M.col[t];
                    Fr coef = fr_from_le(M.coef + 32 * (size_t)t);
                    acc = acc + coef * (v < cs.n_inputs ? A.in[v] : A.aux[v - cs.n_inputs]);
                }
                (*dst[mi])[row] = acc;
            }
        });
        for (uint32_t t = 0; t < M.rowptr[cs.n_constraints]; ++t) {
            uint32_t v = M.col[t];
            if (mi == 0 && v >= cs.n_inputs) A.a_aux_density[v - cs.n_inputs] = 1;
            if (mi == 1) {
                if (v < cs.n_inputs)
                    A.b_input_density[v] = 1;
                else
                    A.b_aux_density[v - cs.n_inputs] = 1;
            }
        }
    }
    for (uint32_t i = 0; i < cs.n_inputs; ++i) A.a[cs.n_constraints + i] = A.in[i];
}

struct ProofPoints {
    G1Affine a, c;
    G2Affine b;
};
static void proof_write(const ProofPoints& p, uint8_t* out) {
    g1_write_compressed(p.a, out);
    g2_write_compressed(p.b, out + 48);
    g1_write_compressed(p.c, out + 144);
}

// timings_ms: [synthesis(eval), ntt, msm_g1, msm_g2, assembly]
static int create_proof(const Params& P, const R1csView& cs, const uint8_t* inputs, const uint8_t* aux, const Fr& r,
                        const Fr& s, ProofPoints& out, double* timings_ms) {
    double t0 = now_ms();
    Assignment A;
    synthesize_from_r1cs(cs, inputs, aux, A);
    size_t nrows = A.a.size();
    uint32_t logm = log2_ceil((uint32_t)nrows);
    size_t m = (size_t)1 << logm;
    double t1 = now_ms();
    std::vector<Fr> a = A.a, b = A.b, c = A.c;
    a.resize(m, Fr::zero());
    b.resize(m, Fr::zero());
    c.resize(m, Fr::zero());
    quotient_h(a, b, c, logm);  // a now holds h[0..m-2]
    double t2 = now_ms();

    // length invariants (SURVEY.md App. C "load-time invariants")
    size_t na_aux = 0, nb_in = 0, nb_aux = 0;
    for (auto d : A.a_aux_density) na_aux += d;
    for (auto d : A.b_input_density) nb_in += d;
    for (auto d : A.b_aux_density) nb_aux += d;
    if (P.h.size() < m - 1 || P.l.size() != cs.n_aux || P.a.size() != cs.n_inputs + na_aux ||
        P.b_g1.size() != nb_in + nb_aux || P.b_g2.size() != nb_in + nb_aux)
        return -2;

    auto canon = [](const std::vector<Fr>& v, std::vector<std::array<uint64_t, 4>>& o) {
        o.resize(v.size());
        parallel_for(v.size(), [&](size_t lo, size_t hi, int) {
            for (size_t i = lo; i < hi; ++i) v[i].to_canonical(o[i].data());
        });
    };
    std::vector<std::array<uint64_t, 4>> eh, eaux, ein;
    canon(a, eh);
    canon(A.aux, eaux);
    canon(A.in, ein);
    typedef const uint64_t(*SP)[4];
    std::vector<std::array<uint64_t, 4>> ea, eb;
    for (uint32_t j = 0; j < cs.n_aux; ++j)
        if (A.a_aux_density[j]) ea.push_back(eaux[j]);
    for (uint32_t i = 0; i < cs.n_inputs; ++i)
        if (A.b_input_density[i]) eb.push_back(ein[i]);
    for (uint32_t j = 0; j < cs.n_aux; ++j)
        if (A.b_aux_density[j]) eb.push_back(eaux[j]);
    MultiexpJob<Fp> jH(P.h.data(), (SP)eh.data(), m - 1), jL(P.l.data(), (SP)eaux.data(), cs.n_aux),
        jAin(P.a.data(), (SP)ein.data(), cs.n_inputs), jAaux(P.a.data() + cs.n_inputs, (SP)ea.data(), ea.size()),
        jB1(P.b_g1.data(), (SP)eb.data(), eb.size());
    // bellperson runs B_in and B_aux as two calls; a single call over the concatenation is the same sum
    MultiexpJob<Fp2> jB2(P.b_g2.data(), (SP)eb.data(), eb.size());
    // all multiexps share the pool (bellperson: concurrent on rayon).  The G1 and G2 legs are still timed
    // separately by running them back to back.
    std::vector<WindowTask> wt1, wt2;
    jH.tasks(wt1);
    jL.tasks(wt1);
    jAin.tasks(wt1);
    jAaux.tasks(wt1);
    jB1.tasks(wt1);
    jB2.tasks(wt2);
    run_tasks(wt1);
    G1 H = jH.finish(), L = jL.finish(), A_in = jAin.finish(), A_aux = jAaux.finish(), B1 = jB1.finish();
    double t3 = now_ms();
    run_tasks(wt2);
    G2 B2 = jB2.finish();
    double t4 = now_ms();

    if (P.delta_g1.inf || P.delta_g2.inf) return -3;  // UnexpectedIdentity
    G1 d1 = G1::from_affine(P.delta_g1);
    G2 d2 = G2::from_affine(P.delta_g2);
    G1 a_msm = A_in.add(A_aux);
    G1 g_a = d1.mul_fr(r).add_affine(P.alpha_g1).add(a_msm);
    G2 g_b = d2.mul_fr(s).add_affine(P.beta_g2).add(B2);
    G1 g_c = d1.mul_fr(r * s)
                 .add(G1::from_affine(P.alpha_g1).mul_fr(s))
                 .add(G1::from_affine(P.beta_g1).mul_fr(r))
                 .add(a_msm.mul_fr(s))
                 .add(B1.mul_fr(r))
                 .add(H)
                 .add(L);
    out.a = g_a.to_affine();
    out.b = g_b.to_affine();
    out.c = g_c.to_affine();
    double t5 = now_ms();
    if (timings_ms) {
        timings_ms[0] = t1 - t0;
        timings_ms[1] = t2 - t1;
        timings_ms[2] = t3 - t2;
        timings_ms[3] = t4 - t3;
        timings_ms[4] = t5 - t4;
    }
    return 0;
}

// Oracle 1: toxic-waste closed form.  Requires a satisfying assignment.
static int closed_form_proof(const R1csView& cs, const Toxic& tw, const uint8_t* inputs, const uint8_t* aux,
                             const Fr& r, const Fr& s, ProofPoints& out) {
    Assignment A;
    synthesize_from_r1cs(cs, inputs, aux, A);
    for (size_t k = 0; k < A.a.size(); ++k)
        if (A.a[k] * A.b[k] != A.c[k]) return -4;  // unsatisfied
    QapAtTau q;
    qap_eval(cs, tw.tau, q);
    Fr a_tau = Fr::zero(), b_tau = Fr::zero(), c_tau = Fr::zero(), l_aux = Fr::zero();
    for (uint32_t i = 0; i < cs.n_inputs; ++i) {
        a_tau = a_tau + A.in[i] * q.at[i];
        b_tau = b_tau + A.in[i] * q.bt[i];
        c_tau = c_tau + A.in[i] * q.ct[i];
    }
    for (uint32_t j = 0; j < cs.n_aux; ++j) {
        size_t v = cs.n_inputs + j;
        a_tau = a_tau + A.aux[j] * q.at[v];
        b_tau = b_tau + A.aux[j] * q.bt[v];
        c_tau = c_tau + A.aux[j] * q.ct[v];
        l_aux = l_aux + A.aux[j] * (tw.beta * q.at[v] + tw.alpha * q.bt[v] + q.ct[v]);
    }
    Fr ea = tw.alpha + a_tau + r * tw.delta;
    Fr eb = tw.beta + b_tau + s * tw.delta;
    Fr hz = a_tau * b_tau - c_tau;
    Fr ec = (l_aux + hz) * tw.delta.inv() + s * ea + r * eb - r * s * tw.delta;
    out.a = fb_g1().mul(ea).to_affine();
    out.b = fb_g2().mul(eb).to_affine();
    out.c = fb_g1().mul(ec).to_affine();
    return 0;
}

// Oracle 2: Groth16 verification equation.  public_inputs excludes ONE.
static int verify_proof(const Params& vk, const ProofPoints& pr, const uint8_t* public_inputs, uint32_t n_public) {
    if ((size_t)n_public + 1 != vk.ic.size()) return -1;
    G1 acc = G1::from_affine(vk.ic[0]);
    for (uint32_t i = 0; i < n_public; ++i) acc = acc.add(G1::from_affine(vk.ic[i + 1]).mul_fr(fr_from_le(public_inputs + 32 * i)));
    Fp12 lhs = miller_loop(pr.a, pr.b);
    Fp12 rhs = miller_loop(vk.alpha_g1, vk.beta_g2) * miller_loop(acc.to_affine(), vk.gamma_g2) *
               miller_loop(pr.c, vk.delta_g2);
    return final_exponentiation(lhs * rhs.inv()) == Fp12::one() ? 1 : 0;
}

}  // namespace oracle

// =================================================================================================
// C interface for ctypes (tests / bench cpu_baseline only)
// =================================================================================================
using namespace oracle;

extern "C" {

struct oracle_r1cs {
    uint32_t n_inputs, n_aux, n_constraints;
    const uint32_t* a_rowptr;
    const uint32_t* a_col;
    const uint8_t* a_coef;
    const uint32_t* b_rowptr;
    const uint32_t* b_col;
    const uint8_t* b_coef;
    const uint32_t* c_rowptr;
    const uint32_t* c_col;
    const uint8_t* c_coef;
};
static R1csView view(const oracle_r1cs* r) {
    R1csView v;
    v.n_inputs = r->n_inputs;
    v.n_aux = r->n_aux;
    v.n_constraints = r->n_constraints;
    v.m[0] = {r->a_rowptr, r->a_col, r->a_coef};
    v.m[1] = {r->b_rowptr, r->b_col, r->b_coef};
    v.m[2] = {r->c_rowptr, r->c_col, r->c_coef};
    return v;
}
static Toxic toxic(const uint8_t* t) {
    return Toxic{fr_from_le(t), fr_from_le(t + 32), fr_from_le(t + 64), fr_from_le(t + 96), fr_from_le(t + 128)};
}

void oracle_set_threads(int n) { g_threads = n; }
int oracle_get_threads() { return nthreads(); }

// ---- field / group primitives exposed for python big-integer cross-checks ----
// op: 0 add, 1 sub, 2 mul, 3 inv(a), 4 neg(a);  operands/result canonical little-endian
int oracle_fr_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    Fr x, y;
    if (!Fr::from_bytes_le(x, a) || !Fr::from_bytes_le(y, b)) return -1;
    Fr r = op == 0 ? x + y : op == 1 ? x - y : op == 2 ? x * y : op == 3 ? x.inv() : x.neg();
    r.to_bytes_le(out);
    return 0;
}
int oracle_fp_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    Fp x, y;
    if (!Fp::from_bytes_le(x, a) || !Fp::from_bytes_le(y, b)) return -1;
    Fp r = op == 0 ? x + y : op == 1 ? x - y : op == 2 ? x * y : op == 3 ? x.inv() : x.neg();
    r.to_bytes_le(out);
    return 0;
}
void oracle_fr_root_of_unity(uint8_t* out_le) { fr_root_of_unity().to_bytes_le(out_le); }
void oracle_fr_omega(uint32_t logm, uint8_t* out_le) { fr_omega(logm).to_bytes_le(out_le); }

// [k]G in uncompressed (96 / 192 B) and compressed (48 / 96 B) form; k canonical LE
void oracle_g1_mul_gen(const uint8_t* k, uint8_t* unc96, uint8_t* comp48) {
    G1Affine p = fb_g1().mul(fr_from_le(k)).to_affine();
    if (unc96) g1_write_uncompressed(p, unc96);
    if (comp48) g1_write_compressed(p, comp48);
}
void oracle_g2_mul_gen(const uint8_t* k, uint8_t* unc192, uint8_t* comp96) {
    G2Affine p = fb_g2().mul(fr_from_le(k)).to_affine();
    if (unc192) g2_write_uncompressed(p, unc192);
    if (comp96) g2_write_compressed(p, comp96);
}
// generic (double-and-add) [k]P for an arbitrary uncompressed point
int oracle_g1_mul(const uint8_t* p96, const uint8_t* k, uint8_t* out96) {
    G1Affine p;
    if (!g1_read_uncompressed(p, p96)) return -1;
    g1_write_uncompressed(G1::from_affine(p).mul_fr(fr_from_le(k)).to_affine(), out96);
    return 0;
}
int oracle_g2_mul(const uint8_t* p192, const uint8_t* k, uint8_t* out192) {
    G2Affine p;
    if (!g2_read_uncompressed(p, p192)) return -1;
    g2_write_uncompressed(G2::from_affine(p).mul_fr(fr_from_le(k)).to_affine(), out192);
    return 0;
}
int oracle_g1_add(const uint8_t* p96, const uint8_t* q96, uint8_t* out96) {
    G1Affine p, q;
    if (!g1_read_uncompressed(p, p96) || !g1_read_uncompressed(q, q96)) return -1;
    g1_write_uncompressed(G1::from_affine(p).add_affine(q).to_affine(), out96);
    return 0;
}
int oracle_g2_add(const uint8_t* p192, const uint8_t* q192, uint8_t* out192) {
    G2Affine p, q;
    if (!g2_read_uncompressed(p, p192) || !g2_read_uncompressed(q, q192)) return -1;
    g2_write_uncompressed(G2::from_affine(p).add_affine(q).to_affine(), out192);
    return 0;
}
int oracle_g1_decompress(const uint8_t* c48, uint8_t* out96) {
    G1Affine p;
    if (!g1_read_compressed(p, c48)) return -1;
    g1_write_uncompressed(p, out96);
    return 0;
}
int oracle_g2_decompress(const uint8_t* c96, uint8_t* out192) {
    G2Affine p;
    if (!g2_read_compressed(p, c96)) return -1;
    g2_write_uncompressed(p, out192);
    return 0;
}
int oracle_g1_compress(const uint8_t* p96, uint8_t* out48) {
    G1Affine p;
    if (!g1_read_uncompressed(p, p96)) return -1;
    g1_write_compressed(p, out48);
    return 0;
}
int oracle_g2_compress(const uint8_t* p192, uint8_t* out96) {
    G2Affine p;
    if (!g2_read_uncompressed(p, p192)) return -1;
    g2_write_compressed(p, out96);
    return 0;
}
// e(aG1, bG2) == e(G1,G2)^(ab) and != 1 : returns 1 if bilinear & non-degenerate
int oracle_pairing_selftest(const uint8_t* a_le, const uint8_t* b_le) {
    Fr a = fr_from_le(a_le), b = fr_from_le(b_le);
    G1Affine P = fb_g1().mul(a).to_affine();
    G2Affine Q = fb_g2().mul(b).to_affine();
    Fp12 e1 = pairing(P, Q);
    Fp12 e0 = pairing(g1_generator(), g2_generator());
    uint64_t ab[4];
    (a * b).to_canonical(ab);
    Fp12 e2 = e0.pow(std::vector<uint64_t>(ab, ab + 4));
    if (e0 == Fp12::one()) return 0;
    return e1 == e2 ? 1 : 0;
}

// ---- MSM / NTT building blocks (checkers for the individual HIP kernels) ----
// bases: n uncompressed points; scalars: n x 32 B canonical LE; out: uncompressed sum
int oracle_msm_g1(const uint8_t* bases96, const uint8_t* scalars, size_t n, uint8_t* out96) {
    std::vector<G1Affine> b(n);
    std::vector<std::array<uint64_t, 4>> e(n);
    for (size_t i = 0; i < n; ++i) {
        if (!g1_read_uncompressed(b[i], bases96 + 96 * i)) return -1;
        fr_from_le(scalars + 32 * i).to_canonical(e[i].data());
    }
    g1_write_uncompressed(multiexp<Fp>(b.data(), (const uint64_t(*)[4])e.data(), n).to_affine(), out96);
    return 0;
}
int oracle_msm_g2(const uint8_t* bases192, const uint8_t* scalars, size_t n, uint8_t* out192) {
    std::vector<G2Affine> b(n);
    std::vector<std::array<uint64_t, 4>> e(n);
    for (size_t i = 0; i < n; ++i) {
        if (!g2_read_uncompressed(b[i], bases192 + 192 * i)) return -1;
        fr_from_le(scalars + 32 * i).to_canonical(e[i].data());
    }
    g2_write_uncompressed(multiexp<Fp2>(b.data(), (const uint64_t(*)[4])e.data(), n).to_affine(), out192);
    return 0;
}
// n points k_i*G1 -> uncompressed (fixed-base tables); used to build test bases quickly
void oracle_g1_mul_gen_many(const uint8_t* ks, size_t n, uint8_t* out96) {
    std::vector<Fr> k(n);
    for (size_t i = 0; i < n; ++i) k[i] = fr_from_le(ks + 32 * i);
    std::vector<G1Affine> o;
    fb_g1().mul_many(k, o);
    for (size_t i = 0; i < n; ++i) g1_write_uncompressed(o[i], out96 + 96 * i);
}
void oracle_g2_mul_gen_many(const uint8_t* ks, size_t n, uint8_t* out192) {
    std::vector<Fr> k(n);
    for (size_t i = 0; i < n; ++i) k[i] = fr_from_le(ks + 32 * i);
    std::vector<G2Affine> o;
    fb_g2().mul_many(k, o);
    for (size_t i = 0; i < n; ++i) g2_write_uncompressed(o[i], out192 + 192 * i);
}
// forward NTT of 2^logm elements (natural order in and out), canonical LE
void oracle_ntt(uint8_t* data, uint32_t logm, int inverse) {
    size_t m = (size_t)1 << logm;
    std::vector<Fr> a(m);
    for (size_t i = 0; i < m; ++i) a[i] = fr_from_le(data + 32 * i);
    Domain d(logm);
    if (inverse)
        d.ifft(a);
    else
        fft(a, d.omega, logm);
    for (size_t i = 0; i < m; ++i) a[i].to_bytes_le(data + 32 * i);
}
// a,b,c: nrows x 32 B (canonical LE evaluation vectors); h_out: (m-1) x 32 B
void oracle_quotient_h(const uint8_t* a, const uint8_t* b, const uint8_t* c, size_t nrows, uint32_t logm, uint8_t* h_out) {
    size_t m = (size_t)1 << logm;
    std::vector<Fr> va(m, Fr::zero()), vb(m, Fr::zero()), vc(m, Fr::zero());
    for (size_t i = 0; i < nrows; ++i) {
        va[i] = fr_from_le(a + 32 * i);
        vb[i] = fr_from_le(b + 32 * i);
        vc[i] = fr_from_le(c + 32 * i);
    }
    quotient_h(va, vb, vc, logm);
    for (size_t i = 0; i + 1 < m; ++i) va[i].to_bytes_le(h_out + 32 * i);
}

// ---- Groth16 ----
// toxic: tau|alpha|beta|gamma|delta, 5 x 32 B LE.  Returns bytes written, or required size if cap too small.
size_t oracle_generate_parameters(const oracle_r1cs* cs, const uint8_t* toxic_le, uint8_t* out, size_t cap) {
    Params P;
    generate_parameters(view(cs), toxic(toxic_le), P);
    std::vector<uint8_t> o;
    params_write(P, o);
    if (o.size() <= cap) memcpy(out, o.data(), o.size());
    return o.size();
}
// evaluation vectors a,b,c (nrows = n_constraints + n_inputs, each 32 B LE) and density bitmaps (1 byte per var)
void oracle_r1cs_eval(const oracle_r1cs* cs, const uint8_t* inputs, const uint8_t* aux, uint8_t* a, uint8_t* b,
                      uint8_t* c, uint8_t* a_aux_density, uint8_t* b_input_density, uint8_t* b_aux_density) {
    R1csView v = view(cs);
    Assignment A;
    synthesize_from_r1cs(v, inputs, aux, A);
    for (size_t k = 0; k < A.a.size(); ++k) {
        if (a) A.a[k].to_bytes_le(a + 32 * k);
        if (b) A.b[k].to_bytes_le(b + 32 * k);
        if (c) A.c[k].to_bytes_le(c + 32 * k);
    }
    if (a_aux_density) memcpy(a_aux_density, A.a_aux_density.data(), v.n_aux);
    if (b_input_density) memcpy(b_input_density, A.b_input_density.data(), v.n_inputs);
    if (b_aux_density) memcpy(b_aux_density, A.b_aux_density.data(), v.n_aux);
}
// returns number of unsatisfied rows
size_t oracle_r1cs_unsatisfied(const oracle_r1cs* cs, const uint8_t* inputs, const uint8_t* aux) {
    Assignment A;
    synthesize_from_r1cs(view(cs), inputs, aux, A);
    size_t bad = 0;
    for (size_t k = 0; k < A.a.size(); ++k)
        if (A.a[k] * A.b[k] != A.c[k]) ++bad;
    return bad;
}

struct oracle_params;  // opaque parsed Parameters
oracle_params* oracle_params_parse(const uint8_t* buf, size_t len) {
    Params* P = new Params;
    if (!params_read(*P, buf, len, false)) {
        delete P;
        return nullptr;
    }
    return (oracle_params*)P;
}
void oracle_params_free(oracle_params* p) { delete (Params*)p; }
// lens: ic, h, l, a, b_g1, b_g2
void oracle_params_lens(const oracle_params* p, uint32_t* lens) {
    const Params* P = (const Params*)p;
    lens[0] = P->ic.size();
    lens[1] = P->h.size();
    lens[2] = P->l.size();
    lens[3] = P->a.size();
    lens[4] = P->b_g1.size();
    lens[5] = P->b_g2.size();
}

// timings_ms may be NULL; 5 doubles: eval, ntt, msm_g1, msm_g2, assembly
int oracle_create_proof(const oracle_params* p, const oracle_r1cs* cs, const uint8_t* inputs, const uint8_t* aux,
                        const uint8_t* r_le, const uint8_t* s_le, uint8_t* proof192, double* timings_ms) {
    ProofPoints pp;
    int rc = create_proof(*(const Params*)p, view(cs), inputs, aux, fr_from_le(r_le), fr_from_le(s_le), pp, timings_ms);
    if (rc) return rc;
    proof_write(pp, proof192);
    return 0;
}
int oracle_closed_form_proof(const oracle_r1cs* cs, const uint8_t* toxic_le, const uint8_t* inputs, const uint8_t* aux,
                             const uint8_t* r_le, const uint8_t* s_le, uint8_t* proof192) {
    ProofPoints pp;
    int rc = closed_form_proof(view(cs), toxic(toxic_le), inputs, aux, fr_from_le(r_le), fr_from_le(s_le), pp);
    if (rc) return rc;
    proof_write(pp, proof192);
    return 0;
}
// params may be the full file or only its vk prefix; public_inputs exclude ONE.  1 = valid, 0 = invalid, <0 error
int oracle_verify_proof(const uint8_t* params, size_t len, const uint8_t* proof192, const uint8_t* public_inputs,
                        uint32_t n_public) {
    Params vk;
    if (!params_read(vk, params, len, true)) return -1;
    ProofPoints pp;
    if (!g1_read_compressed(pp.a, proof192) || !g2_read_compressed(pp.b, proof192 + 48) ||
        !g1_read_compressed(pp.c, proof192 + 144))
        return -2;
    if (!on_curve(pp.a) || !on_curve(pp.b) || !on_curve(pp.c)) return -3;
    return verify_proof(vk, pp, public_inputs, n_public);
}

}  // extern "C"

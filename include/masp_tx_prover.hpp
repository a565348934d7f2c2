// masp_tx_prover.hpp — the host side of the drop-in in C++17, above the two C ABIs (masp_hip.h: the Groth16 prover on the GPU;
// masp_host.h: witness synthesis, native primitives, host verification).  Header-only; link -lmasp_hip -lmasp_host.
//
// Mirrors, name for name, what a user of the reference holds (citations relative to /root/reference):
//   masp::LocalTxProver            = masp_proofs::prover::LocalTxProver               masp_proofs/src/prover.rs:27-33,55-95,156-261
//     ::from_bytes / ::from_paths  = LocalTxProver::from_bytes / ::new                prover.rs:81-95 / :55-64
//     ::with_default_location      = LocalTxProver::with_default_location             prover.rs:120-136
//     ::new_sapling_proving_context, ::spend_proof, ::output_proof, ::convert_proof, ::binding_sig
//                                  = trait TxProver                                   masp_primitives/src/sapling/prover.rs:17-83
//   masp::SaplingProvingContext    = masp_proofs::sapling::SaplingProvingContext (bsk, cv_sum)   masp_proofs/src/sapling/prover.rs:26-47
// with the argument order and the error behaviour of the reference:
//   Result<_, ()>  ->  std::optional<_>: empty iff the diversifier is invalid (sapling/prover.rs:84) or the proof fails its
//                      self-verification (:148, :266); the context then holds the new bsk and the old cv_sum, as in the reference (:69-75 vs :154)
//   panic          ->  masp::Panic (a std::runtime_error): parameters that do not load (lib.rs:290-293,337,359-362), proving that fails
//                      ("proving should not fail", sapling/prover.rs:117,202,252), output_proof on a statement that cannot be synthesised
// and one extension the reference does not have: spend_proofs / output_proofs / convert_proofs take the
// descriptions of a whole transaction (the serial loops of SaplingBuilder::build, masp_primitives/src/transaction/components/sapling/
// builder.rs:955-969,1007-1016) and prove them in batches of masp_hip_options::batch_cap — witnesses synthesised on host threads while
// earlier batches prove, the Spend / Convert self-checks as one GPU batch verification per batch — with the context accumulated in
// description order, so that the proofs, cv, rk, bsk and cv_sum are those of the serial loop.
// binding_sig (sapling/prover.rs:279-326) — RedJubjub over the context's bsk / cv_sum; it does not touch the GPU prover (SURVEY.md §8b
// "Not touched by the build") and is here so that the trait is whole: BLAKE2b-512 and jubjub::Fr arithmetic in this header, the curve
// through libmasp_host.
// The C ABI does no hashing; this layer does what `load_parameters` / `parse_parameters` do (lib.rs:278-388): file sizes before
// any large read, the BLAKE2b-512 digest of every file against LocalTxProverConfig::expected (default: the pinned MPC files).
//
// Thread-safe like the reference's `&self` methods: one LocalTxProver may be shared by several threads, each with its own context.
#ifndef MASP_TX_PROVER_HPP
#define MASP_TX_PROVER_HPP
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#ifdef __linux__
#include <sched.h>
#include <sys/random.h>
#include <sys/types.h>
#endif

#include "masp_hip.h"
#include "masp_host.h"

namespace masp {

constexpr size_t GROTH_PROOF_SIZE = 192;  // masp_primitives/src/transaction/components.rs:15
using Bytes32 = std::array<uint8_t, 32>;
using GrothProofBytes = std::array<uint8_t, GROTH_PROOF_SIZE>;

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---- the argument types of the trait, as the bytes the reference's types serialise to ----
struct ProofGenerationKey {  // masp_primitives/src/sapling.rs: ak (a Jubjub point), nsk (jubjub::Fr)
    Bytes32 ak, nsk;
};
struct Diversifier {
    std::array<uint8_t, 11> bytes;
};
struct PaymentAddress {
    Diversifier diversifier;
    Bytes32 pk_d;
};
struct AssetType {
    Bytes32 identifier;
    static std::optional<AssetType> from_name(const std::string& name) {  // AssetType::new (asset_type.rs)
        AssetType a;
        if (masp_host_asset_identifier(reinterpret_cast<const uint8_t*>(name.data()), name.size(), a.identifier.data()) != MASP_HOST_OK) return std::nullopt;
        return a;
    }
};
struct I128 {                              // a signed 128-bit amount: 16 little-endian two's-complement bytes (the values of an I128Sum)
    std::array<uint8_t, 16> le;
    static I128 from_i64(int64_t v) {
        I128 x;
        for (int i = 0; i < 16; ++i) x.le[i] = i < 8 ? (uint8_t)((uint64_t)v >> (8 * i)) : (v < 0 ? 0xff : 0);
        return x;
    }
    bool negative() const { return (le[15] & 0x80) != 0; }
    bool is_min() const {                  // i128::MIN has no absolute value (checked_abs, sapling/mod.rs:14-20)
        for (int i = 0; i < 15; ++i)
            if (le[i]) return false;
        return le[15] == 0x80;
    }
    Bytes32 magnitude() const {            // |v| as a jubjub::Fr
        Bytes32 m{};
        unsigned carry = negative() ? 1 : 0;
        for (int i = 0; i < 16; ++i) {
            const unsigned b = negative() ? (uint8_t)~le[i] + carry : le[i];
            m[i] = (uint8_t)b;
            carry = negative() ? b >> 8 : 0;
        }
        return m;
    }
};
struct MerklePath {                        // MerklePath<Node>: the authentication path, leaf level first, and the leaf's position
    std::array<Bytes32, 32> auth_path;
    uint64_t position;
};
struct AllowedConversion {                 // masp_primitives/src/convert.rs:22-29: the circuit sees its asset generator only
    Bytes32 generator;
    // AllowedConversion::from(I128Sum) (convert.rs:86-118): (asset, signed 128-bit amount as 16 little-endian two's-complement bytes)
    static std::optional<AllowedConversion> from(const std::vector<std::pair<AssetType, I128>>& assets) {
        std::vector<uint8_t> ids(32 * assets.size()), vals(16 * assets.size());
        for (size_t i = 0; i < assets.size(); ++i) {
            std::memcpy(&ids[32 * i], assets[i].first.identifier.data(), 32);
            std::memcpy(&vals[16 * i], assets[i].second.le.data(), 16);
        }
        AllowedConversion c;
        if (masp_host_allowed_conversion(assets.size(), ids.data(), vals.data(), c.generator.data()) != MASP_HOST_OK) return std::nullopt;
        return c;
    }
};

namespace detail {
// little-endian 256-bit helpers for the two scalar fields that cross this interface
struct U256 {
    uint64_t w[4];
};
inline U256 load(const uint8_t* b) {
    U256 x;
    for (int i = 0; i < 4; ++i) {
        x.w[i] = 0;
        for (int j = 7; j >= 0; --j) x.w[i] = (x.w[i] << 8) | b[8 * i + j];
    }
    return x;
}
inline void store(const U256& x, uint8_t* b) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(x.w[i] >> (8 * j));
}
inline bool geq(const U256& a, const U256& b) {
    for (int i = 3; i >= 0; --i)
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    return true;
}
inline uint64_t add_in_place(U256& a, const U256& b) {
    uint64_t carry = 0;
    for (int i = 0; i < 4; ++i) {
        const uint64_t t = a.w[i] + carry;
        carry = t < carry;
        a.w[i] = t + b.w[i];
        carry += a.w[i] < t;
    }
    return carry;
}
inline void sub_in_place(U256& a, const U256& b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        const uint64_t t = a.w[i] - borrow;
        borrow = a.w[i] < borrow;
        a.w[i] = t - b.w[i];
        borrow += t < b.w[i];
    }
}
// the order of the Jubjub prime-order subgroup (jubjub::Fr) and the BLS12-381 scalar field (bls12_381::Scalar)
constexpr U256 JUBJUB_ORDER = {{0xd0970e5ed6f72cb7ULL, 0xa6682093ccc81082ULL, 0x06673b0101343b00ULL, 0x0e7db4ea6533afa9ULL}};
constexpr U256 FR_MODULUS = {{0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}};
// (a +/- b) mod the Jubjub order, both canonical
inline Bytes32 fs_add(const Bytes32& a, const Bytes32& b, bool subtract) {
    U256 x = load(a.data()), y = load(b.data());
    if (subtract) {
        if (!geq(x, y)) add_in_place(x, JUBJUB_ORDER);  // (< 2^253: no carry out)
        sub_in_place(x, y);
    } else {
        add_in_place(x, y);
        if (geq(x, JUBJUB_ORDER)) sub_in_place(x, JUBJUB_ORDER);
    }
    Bytes32 r;
    store(x, r.data());
    return r;
}
inline bool fs_canonical(const Bytes32& a) { return !geq(load(a.data()), JUBJUB_ORDER); }
// a 512-bit little-endian integer (8 x u64) mod the Jubjub order: jubjub::Fr::from_bytes_wide.  One bit at a time — this runs twice per
// binding signature, not in any loop over proofs
inline U256 fs_reduce_wide(const uint64_t x[8]) {
    U256 r = {{0, 0, 0, 0}};
    for (int bit = 511; bit >= 0; --bit) {
        for (int i = 3; i > 0; --i) r.w[i] = (r.w[i] << 1) | (r.w[i - 1] >> 63);  // r < 2^252: nothing falls off the top
        r.w[0] = (r.w[0] << 1) | ((x[bit >> 6] >> (bit & 63)) & 1);
        if (geq(r, JUBJUB_ORDER)) sub_in_place(r, JUBJUB_ORDER);
    }
    return r;
}
// a * b mod the Jubjub order
inline U256 fs_mul(const U256& a, const U256& b) {
    uint64_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; ++j) {  // 64 x 64 -> 128 by halves (no __int128: the header stays ISO C++)
            const uint64_t al = a.w[i] & 0xffffffffu, ah = a.w[i] >> 32, bl = b.w[j] & 0xffffffffu, bh = b.w[j] >> 32;
            const uint64_t ll = al * bl, lh = al * bh, hl = ah * bl, hh = ah * bh;
            const uint64_t mid = (ll >> 32) + (lh & 0xffffffffu) + (hl & 0xffffffffu);
            uint64_t lo = (ll & 0xffffffffu) | (mid << 32), hi = hh + (lh >> 32) + (hl >> 32) + (mid >> 32);
            lo += carry;
            hi += lo < carry;
            const uint64_t t = p[i + j] + lo;
            hi += t < lo;
            p[i + j] = t;
            carry = hi;
        }
        p[i + 4] = carry;
    }
    return fs_reduce_wide(p);
}
// BLAKE2b-512 (RFC 7693), unkeyed, with a 16-byte personalization: what RedJubjub's H* needs (masp_primitives/src/sapling/util.rs:9-15)
class Blake2b512 {
  public:
    explicit Blake2b512(const char personal[16]) {
        static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        for (int i = 0; i < 8; ++i) h_[i] = iv[i];
        h_[0] ^= 0x01010000ULL ^ 64;  // digest length 64, no key, fanout 1, depth 1
        uint64_t p[2] = {0, 0};
        for (int i = 0; i < 16; ++i) p[i >> 3] |= (uint64_t)(uint8_t)personal[i] << (8 * (i & 7));
        h_[6] ^= p[0];
        h_[7] ^= p[1];
    }
    void update(const uint8_t* in, size_t n) {
        while (n) {
            if (fill_ == 128) {  // (the last block is only compressed by finish: it carries the final flag)
                t_ += 128;
                compress(false);
                fill_ = 0;
            }
            const size_t k = std::min<size_t>(n, 128 - fill_);
            std::memcpy(buf_ + fill_, in, k);
            fill_ += k;
            in += k;
            n -= k;
        }
    }
    void finish(uint8_t out[64]) {
        t_ += fill_;
        std::memset(buf_ + fill_, 0, 128 - fill_);
        compress(true);
        for (int i = 0; i < 64; ++i) out[i] = (uint8_t)(h_[i >> 3] >> (8 * (i & 7)));
    }

  private:
    static uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void compress(bool last) {
        static const uint8_t sigma[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; ++i) {
            m[i] = 0;
            for (int j = 7; j >= 0; --j) m[i] = (m[i] << 8) | buf_[8 * i + j];
        }
        for (int i = 0; i < 8; ++i) v[i] = h_[i], v[i + 8] = iv[i];
        v[12] ^= t_;  // (messages here are far below 2^64 bytes: the high counter word stays 0)
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; ++r) {
            const uint8_t* s = sigma[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; ++i) h_[i] ^= v[i] ^ v[i + 8];
    }
    uint64_t h_[8], t_ = 0;
    uint8_t buf_[128];
    size_t fill_ = 0;
};
// RedJubjub's H*(a || b): BLAKE2b-512 personalised "MASP__RedJubjubH", reduced into jubjub::Fr (redjubjub.rs:36-38, util.rs:9-15)
inline U256 h_star(const uint8_t* a, size_t na, const uint8_t* b, size_t nb) {
    Blake2b512 h("MASP__RedJubjubH");
    h.update(a, na);
    h.update(b, nb);
    uint8_t d[64];
    h.finish(d);
    uint64_t x[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = 0;
        for (int j = 7; j >= 0; --j) x[i] = (x[i] << 8) | d[8 * i + j];
    }
    return fs_reduce_wide(x);
}
inline bool fr_canonical(const uint8_t* a) { return !geq(load(a), FR_MODULUS); }
// bellman multipack::compute_multipacking(bytes_to_bits_le(data)) for 32 bytes: two scalars of 254 and 2 bits (sapling/prover.rs:138-139)
inline void multipack32(const uint8_t data[32], uint8_t out[64]) {
    std::memcpy(out, data, 32);
    out[31] &= 0x3f;                    // bits 0..253
    std::memset(out + 32, 0, 32);
    out[32] = (uint8_t)(data[31] >> 6);  // bits 254, 255
}
// host threads worth starting: what the scheduler lets this process run on, capped by a cgroup CPU quota when one is set (a container
// that sees 256 logical CPUs may be entitled to 16 of them; oversubscribing it only adds contention)
inline unsigned effective_cpus() {
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = (unsigned)CPU_COUNT(&set);
#endif
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string quota;
    unsigned long long period = 0;
    if (f >> quota >> period && quota != "max" && period > 0) {
        const unsigned long long q = std::strtoull(quota.c_str(), nullptr, 10) / period;
        n = (unsigned)std::min<unsigned long long>(n, std::max<unsigned long long>(1, q));
    } else {
        std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
        long long q = 0, p = 0;
        if (fq >> q && fp >> p && q > 0 && p > 0) n = (unsigned)std::min<long long>(n, std::max<long long>(1, q / p));
    }
    return n;
}
// `n` permits handed out in ticket order: the batches of a transaction enter the context's slots in the order they were synthesised
// (their results are committed in that order: a later batch that overtook an earlier one for a slot would finish first and then wait,
// holding its buffers, while the pipeline behind it stands still)
class FifoPermits {
  public:
    explicit FifoPermits(size_t n = 1) : free_(n) {}
    void resize(size_t n) { free_ = n; }
    size_t ticket() {
        std::lock_guard<std::mutex> g(mu_);
        return issued_++;
    }
    void acquire(size_t ticket) {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return serving_ == ticket && free_ > 0; });
        --free_;
        ++serving_;
        cv_.notify_all();
    }
    void release() {
        std::lock_guard<std::mutex> g(mu_);
        ++free_;
        cv_.notify_all();
    }

  private:
    std::mutex mu_;
    std::condition_variable cv_;
    size_t free_, issued_ = 0, serving_ = 0;
};
// bytes from the operating system's generator: getrandom(2) where there is one (one system call per request), std::random_device otherwise
inline void os_random(uint8_t* out, size_t n) {
#ifdef __linux__
    size_t got = 0;
    while (got < n) {
        const ssize_t k = getrandom(out + got, n - got, 0);
        if (k <= 0) break;
        got += (size_t)k;
    }
    if (got == n) return;
#endif
    static thread_local std::random_device rd;
    for (size_t i = 0; i < n; i += 4) {
        const uint32_t w = rd();
        std::memcpy(out + i, &w, std::min<size_t>(4, n - i));
    }
}
inline std::string hip_error(masp_hip_ctx* ctx, int rc, const char* what) {
    std::string s = std::string(what) + ": " + masp_hip_strerror(rc);
    const char* more = ctx ? masp_hip_last_error(ctx) : nullptr;
    if (more && *more) s += std::string(" (") + more + ")";
    return s;
}
}  // namespace detail

// Rseed (masp_primitives/src/sapling.rs:643-647) and Note::rcm (:856-864): the note commitment randomness itself (BeforeZip212), or 32
// seed bytes from which it is derived as jubjub::Fr::from_bytes_wide(PRF^expand(rseed, [0x04])), PRF^expand = BLAKE2b-512 personalised
// "MASP__ExpandSeed" (masp_primitives/src/keys.rs:5-20)
struct Rseed {
    enum Kind { BeforeZip212, AfterZip212 } kind;
    Bytes32 bytes;
    static Rseed before_zip212(const Bytes32& rcm) { return Rseed{BeforeZip212, rcm}; }
    static Rseed after_zip212(const Bytes32& rseed) { return Rseed{AfterZip212, rseed}; }
    Bytes32 rcm() const {
        if (kind == BeforeZip212) return bytes;
        detail::Blake2b512 h("MASP__ExpandSeed");
        h.update(bytes.data(), 32);
        const uint8_t t = 0x04;
        h.update(&t, 1);
        uint8_t d[64];
        h.finish(d);
        uint64_t x[8];
        for (int i = 0; i < 8; ++i) {
            x[i] = 0;
            for (int j = 7; j >= 0; --j) x[i] = (x[i] << 8) | d[8 * i + j];
        }
        Bytes32 out;
        detail::store(detail::fs_reduce_wide(x), out.data());
        return out;
    }
};

// bsk / cv_sum of one transaction (sapling/prover.rs:26-47).  Independent of the proof bytes: the proofs of a transaction may be
// produced in any order or in one batch as long as the accumulations happen.
class SaplingProvingContext {
  public:
    SaplingProvingContext() {
        bsk_.fill(0);        // jubjub::Fr::zero()
        cv_sum_.fill(0);     // jubjub::ExtendedPoint::identity(): (u, v) = (0, 1)
        cv_sum_[0] = 1;
    }
    const Bytes32& bsk() const { return bsk_; }
    const Bytes32& cv_sum() const { return cv_sum_; }
    // a context with the accumulations of another one (a transaction built in several processes; tests)
    static SaplingProvingContext from_parts(const Bytes32& bsk, const Bytes32& cv_sum) {
        SaplingProvingContext c;
        c.bsk_ = bsk;
        c.cv_sum_ = cv_sum;
        return c;
    }

    // = SaplingProvingContext::binding_sig (sapling/prover.rs:279-326): the RedJubjub signature (Rbar || Sbar, 64 bytes) over
    // bvk || sighash under bsk, after checking — as the verifier will — that bvk = [bsk] G_rcv equals cv_sum minus the value balances.
    // amount: the components of the reference's I128Sum.  Empty (`Err(())`) when the balances do not match the accumulated commitments
    // or a balance is i128::MIN.  Host only; `nonce`: the signature's 80 random bytes T (nullptr: from the operating system, as OsRng).
    std::optional<std::array<uint8_t, 64>> binding_sig(const std::vector<std::pair<AssetType, I128>>& amount, const uint8_t sighash[32],
                                                       const uint8_t* nonce = nullptr) const;

  private:
    friend class LocalTxProver;
    void bsk_add(const Bytes32& rcv, bool subtract) { bsk_ = detail::fs_add(bsk_, rcv, subtract); }  // "Outputs subtract from the total."
    void cv_add(const Bytes32& cv, bool subtract) {
        Bytes32 out;
        if (masp_host_jubjub_add(cv_sum_.data(), cv.data(), subtract ? 1 : 0, out.data()) != MASP_HOST_OK) throw Panic("cv_sum: not a Jubjub point");
        cv_sum_ = out;
    }
    Bytes32 bsk_, cv_sum_;
};

inline std::optional<std::array<uint8_t, 64>> SaplingProvingContext::binding_sig(const std::vector<std::pair<AssetType, I128>>& amount, const uint8_t sighash[32],
                                                                                 const uint8_t* nonce) const {
    // value_commitment_randomness_generator() as point bytes: v with the sign of u in the top bit
    uint8_t uv[64];
    masp_host_generator(3, uv);
    Bytes32 g;
    std::memcpy(g.data(), uv + 32, 32);
    g[31] |= (uint8_t)((uv[0] & 1) << 7);
    Bytes32 bvk;  // PublicKey::from_private(&bsk, G_rcv)
    if (masp_host_jubjub_mul(g.data(), bsk_.data(), bvk.data()) != MASP_HOST_OK) return std::nullopt;
    Bytes32 final_bvk = cv_sum_;
    for (const auto& [asset, value] : amount) {
        if (value.is_min()) return std::nullopt;
        // masp_compute_value_balance: the asset's value commitment generator (its generator with the cofactor cleared) times |value|
        Bytes32 gen, eight{}, vcg, vb, next;
        eight[0] = 8;
        const Bytes32 mag = value.magnitude();
        if (masp_host_asset_generator(asset.identifier.data(), gen.data()) != MASP_HOST_OK || masp_host_jubjub_mul(gen.data(), eight.data(), vcg.data()) != MASP_HOST_OK ||
            masp_host_jubjub_mul(vcg.data(), mag.data(), vb.data()) != MASP_HOST_OK ||
            masp_host_jubjub_add(final_bvk.data(), vb.data(), value.negative() ? 0 : 1, next.data()) != MASP_HOST_OK)
            return std::nullopt;
        final_bvk = next;  // cv_sum minus the value balance (a negative balance: minus its negation)
    }
    if (bvk != final_bvk) return std::nullopt;  // "unless the provided valueBalance is wrong" (:313-316)
    uint8_t msg[64];
    std::memcpy(msg, bvk.data(), 32);
    std::memcpy(msg + 32, sighash, 32);
    // PrivateKey::sign (redjubjub.rs:138-160): T = 80 random bytes, r = H*(T || M), R = [r] G, S = r + H*(Rbar || M) bsk
    uint8_t t[80];
    if (nonce)
        std::memcpy(t, nonce, 80);
    else
        detail::os_random(t, 80);
    const detail::U256 r = detail::h_star(t, 80, msg, 64);
    Bytes32 rb, rbar;
    detail::store(r, rb.data());
    if (masp_host_jubjub_mul(g.data(), rb.data(), rbar.data()) != MASP_HOST_OK) return std::nullopt;
    detail::U256 sv = detail::fs_mul(detail::h_star(rbar.data(), 32, msg, 64), detail::load(bsk_.data()));
    detail::add_in_place(sv, r);
    if (detail::geq(sv, detail::JUBJUB_ORDER)) detail::sub_in_place(sv, detail::JUBJUB_ORDER);
    std::array<uint8_t, 64> sig;
    std::memcpy(sig.data(), rbar.data(), 32);
    detail::store(sv, sig.data() + 32);
    return sig;
}

// what one description of a transaction hands the prover: the arguments of the trait's methods after `ctx`
struct SpendInfo {
    ProofGenerationKey proof_generation_key;
    Diversifier diversifier;
    Bytes32 rcm;  // note.rcm() = Rseed::rcm()
    Bytes32 ar;
    AssetType asset_type;
    uint64_t value;
    Bytes32 anchor;
    MerklePath merkle_path;
    Bytes32 rcv;
};
struct OutputInfo {
    Bytes32 esk;
    PaymentAddress payment_address;
    Bytes32 rcm;
    AssetType asset_type;
    uint64_t value;
    Bytes32 rcv;
};
struct ConvertInfo {
    AllowedConversion allowed_conversion;
    uint64_t value;
    Bytes32 anchor;
    MerklePath merkle_path;
    Bytes32 rcv;
};
struct SpendProof {
    GrothProofBytes zkproof;
    Bytes32 cv;  // jubjub::ExtendedPoint
    Bytes32 rk;  // redjubjub::PublicKey
};
struct ValueProof {  // what output_proof / convert_proof return
    GrothProofBytes zkproof;
    Bytes32 cv;
};
// explicit Groth16 blinding scalars (deterministic replay, tests); the reference draws them from OsRng (sapling/prover.rs:66,174,225)
struct BlindingScalars {
    Bytes32 r, s;
};

// What `load_parameters` / `parse_parameters` hold a parameter file against (masp_proofs/src/lib.rs:278-328, :333-388): its size, checked
// before anything is read, and the BLAKE2b-512 digest of the whole file (body + MPC transcript).
struct ExpectedParameters {
    size_t bytes;
    const char* blake2b_hex;  // 128 hex digits
};
struct ExpectedParameterSet {
    ExpectedParameters spend, output, convert;
};
// the MPC parameters the reference pins: MASP_{SPEND,OUTPUT,CONVERT}_{HASH,BYTES} (lib.rs:60-76)
inline const ExpectedParameterSet& masp_mpc_parameters() {
    static const ExpectedParameterSet set = {
        {49848572, "196e7c717f25e16653431559ce2c8816e750a4490f98696e3c031efca37e25e0647182b7b013660806db11eb2b1e365fb2d6a0f24dbbd9a4a8314fef10a7cba2"},
        {16398620, "eafc3b1746cccc8b9eed2b69395692c5892f6aca83552a07dceb2dcbaa64dcd0e22434260b3aa3b049b633a08b008988cbe0d31effc77e2bc09bfab690a23724"},
        {22570940, "dc4aaf3c3ce056ab448b6c4a7f43c1d68502c2902ea89ab8769b1524a2e8ace9a5369621a73ee1daa52aec826907a19974a37874391cf8f11bbe0b0420de1ab7"}};
    return set;
}

// the builder's `Progress` notifications (builder.rs:946-952): descriptions of the call done so far, of how many
using Progress = std::function<void(size_t done, size_t total)>;

struct LocalTxProverConfig {
    int device = 0;
    std::vector<int> devices;                   // more than one GPU: the HIP devices of one multi-device context (empty: `device` alone)
    bool self_verify = true;                    // sapling/prover.rs:148,266 (tests of the failure paths switch it off)
    const masp_hip_options* options = nullptr;  // slots, batch_cap, ... (nullptr: the library's defaults)
    unsigned threads = 0;                       // synthesis threads of the *_proofs batch methods (0: the CPUs this process may use)
    const ExpectedParameterSet* expected = &masp_mpc_parameters();  // nullptr: parameters that are not the MPC files (benches, tests)
    unsigned calls_in_flight = 0;               // masp_hip_prove_batch calls of the *_proofs methods at a time (0: slots + 1 — one waits inside the
                                                // library for the slot that frees next)
    bool trace = false;                         // one line per batch on stderr: when it was synthesised, got a slot, was proved, verified, committed
};

class LocalTxProver {
  public:
    using Config = LocalTxProverConfig;

    // = LocalTxProver::from_bytes (prover.rs:81-95): the three parameter files' bytes in the bellman wire format
    static std::unique_ptr<LocalTxProver> from_bytes(const uint8_t* spend, size_t spend_len, const uint8_t* output, size_t output_len,
                                                     const uint8_t* convert, size_t convert_len, const Config& cfg = Config()) {
        return std::unique_ptr<LocalTxProver>(new LocalTxProver(spend, spend_len, output, output_len, convert, convert_len, cfg));
    }
    // = LocalTxProver::new (prover.rs:55-64): parameter files on disk
    static std::unique_ptr<LocalTxProver> from_paths(const std::string& spend_path, const std::string& output_path, const std::string& convert_path,
                                                     const Config& cfg = Config()) {
        std::vector<uint8_t> b[3];
        const std::string* p[3] = {&spend_path, &output_path, &convert_path};
        static const char* names[3] = {"masp spend", "masp output", "masp convert"};
        for (int i = 0; i < 3; ++i) {
            std::ifstream f(*p[i], std::ios::binary | std::ios::ate);
            if (!f) throw Panic("cannot open " + *p[i]);  // (the reference: File::open(..).expect(..), lib.rs:284-288)
            const std::streamoff size = f.tellg();
            if (cfg.expected) {  // verify_file_size (lib.rs:409-430): the file system's word, before any large read
                const ExpectedParameters& e = i == 0 ? cfg.expected->spend : i == 1 ? cfg.expected->output : cfg.expected->convert;
                if ((size_t)size != e.bytes)
                    throw Panic(std::string(names[i]) + " parameters " + *p[i] + ": " + std::to_string((long long)size) + " bytes on disk, expected " + std::to_string(e.bytes));
            }
            f.seekg(0);
            b[i].resize((size_t)size);
            f.read(reinterpret_cast<char*>(b[i].data()), size);
            if (!f) throw Panic("cannot read " + *p[i]);
        }
        return from_bytes(b[0].data(), b[0].size(), b[1].data(), b[1].size(), b[2].data(), b[2].size(), cfg);
    }
    // = LocalTxProver::with_default_location (prover.rs:120-136): the three files in default_params_folder() (lib.rs:100-108; this image's
    // platform: ~/.masp-params), or nullptr when the folder or one of them is missing
    static std::unique_ptr<LocalTxProver> with_default_location(const Config& cfg = Config()) {
        const char* home = std::getenv("HOME");
        if (!home || !*home) return nullptr;
        const std::string dir = std::string(home) + "/.masp-params/";
        const std::string p[3] = {dir + "masp-spend.params", dir + "masp-output.params", dir + "masp-convert.params"};
        for (const std::string& x : p)
            if (!std::ifstream(x, std::ios::binary)) return nullptr;
        return from_paths(p[0], p[1], p[2], cfg);
    }
    ~LocalTxProver() { release(); }
    LocalTxProver(const LocalTxProver&) = delete;
    LocalTxProver& operator=(const LocalTxProver&) = delete;

    SaplingProvingContext new_sapling_proving_context() const { return SaplingProvingContext(); }
    masp_hip_ctx* context() const { return ctx_; }
    size_t batch_cap() const { return batch_cap_; }

    // ---- trait TxProver ----
    std::optional<SpendProof> spend_proof(SaplingProvingContext& ctx, const ProofGenerationKey& proof_generation_key, const Diversifier& diversifier,
                                          const Rseed& rseed, const Bytes32& ar, const AssetType& asset_type, uint64_t value, const Bytes32& anchor,
                                          const MerklePath& merkle_path, const Bytes32& rcv, const BlindingScalars* rs = nullptr) {
        const SpendInfo d{proof_generation_key, diversifier, rseed.rcm(), ar, asset_type, value, anchor, merkle_path, rcv};
        auto out = spend_proofs(ctx, &d, 1, rs);
        return out[0];
    }
    ValueProof output_proof(SaplingProvingContext& ctx, const Bytes32& esk, const PaymentAddress& payment_address, const Bytes32& rcm,
                            const AssetType& asset_type, uint64_t value, const Bytes32& rcv, const BlindingScalars* rs = nullptr) {
        const OutputInfo d{esk, payment_address, rcm, asset_type, value, rcv};
        auto out = output_proofs(ctx, &d, 1, rs);
        if (!out[0]) throw Panic("output_proof: the statement cannot be synthesised (invalid diversifier or pk_d)");  // the reference unwraps here (:189-196)
        return *out[0];
    }
    std::optional<ValueProof> convert_proof(SaplingProvingContext& ctx, const AllowedConversion& allowed_conversion, uint64_t value, const Bytes32& anchor,
                                            const MerklePath& merkle_path, const Bytes32& rcv, const BlindingScalars* rs = nullptr) {
        const ConvertInfo d{allowed_conversion, value, anchor, merkle_path, rcv};
        auto out = convert_proofs(ctx, &d, 1, rs);
        return out[0];
    }

    // = TxProver::binding_sig: all of it is the context's (sapling/prover.rs:279-326); no GPU work
    std::optional<std::array<uint8_t, 64>> binding_sig(const SaplingProvingContext& ctx, const std::vector<std::pair<AssetType, I128>>& amount,
                                                       const uint8_t sighash[32]) const {
        return ctx.binding_sig(amount, sighash);
    }

    // ---- the descriptions of a whole transaction at once (see the head of this file).  Element i of the result is what the trait's
    // method returns for description i; `rs`: n explicit (r, s) pairs or nullptr ----
    std::vector<std::optional<SpendProof>> spend_proofs(SaplingProvingContext& ctx, const SpendInfo* d, size_t n, const BlindingScalars* rs = nullptr,
                                                        const Progress& progress = nullptr) {
        std::vector<std::optional<SpendProof>> out(n);
        for (size_t i = 0; i < n; ++i) ctx.bsk_add(d[i].rcv, false);  // sapling/prover.rs:69-75, before anything can fail
        std::vector<Bytes32> rk(n), nf(n);
        run<SpendInfo>(MASP_HIP_SPEND, ctx, false, d, n, rs, progress,
                       [&](size_t lo, size_t cnt, uint8_t* inputs, uint8_t* aux, Bytes32* cv, int* rc) {
                           std::vector<masp_host_spend_job> jobs(cnt);
                           for (size_t k = 0; k < cnt; ++k) {
                               const SpendInfo& x = d[lo + k];
                               jobs[k] = masp_host_spend_job{x.proof_generation_key.ak.data(), x.proof_generation_key.nsk.data(), x.diversifier.bytes.data(),
                                                             x.rcm.data(), x.ar.data(), x.asset_type.identifier.data(), x.value, x.anchor.data(),
                                                             x.merkle_path.auth_path[0].data(), x.merkle_path.position, x.rcv.data(),
                                                             inputs + k * 32 * n_inputs_[MASP_HIP_SPEND], aux + k * 32 * (size_t)n_aux_[MASP_HIP_SPEND],
                                                             cv[k].data(), rk[lo + k].data(), nf[lo + k].data(), 0};
                           }
                           masp_host_spend_assignments(cnt, jobs.data(), 2);
                           for (size_t k = 0; k < cnt; ++k) rc[k] = jobs[k].rc;
                       },
                       // public input of the self-check (sapling/prover.rs:121-145): rk, cv, anchor, the nullifier packed into two scalars
                       [&](size_t i, const Bytes32& cv, uint8_t* pub) {
                           if (masp_host_point_uv(rk[i].data(), pub) != MASP_HOST_OK || masp_host_point_uv(cv.data(), pub + 64) != MASP_HOST_OK) throw Panic("rk / cv: not a Jubjub point");
                           std::memcpy(pub + 128, d[i].anchor.data(), 32);
                           detail::multipack32(nf[i].data(), pub + 160);
                       },
                       7,
                       [&](size_t i, const GrothProofBytes& zk, const Bytes32& cv) { out[i] = SpendProof{zk, cv, rk[i]}; });  // (cv_sum += cv, :154: by `run`)
        return out;
    }
    std::vector<std::optional<ValueProof>> output_proofs(SaplingProvingContext& ctx, const OutputInfo* d, size_t n, const BlindingScalars* rs = nullptr,
                                                         const Progress& progress = nullptr) {
        std::vector<std::optional<ValueProof>> out(n);
        for (size_t i = 0; i < n; ++i) ctx.bsk_add(d[i].rcv, true);  // :177-183
        run<OutputInfo>(MASP_HIP_OUTPUT, ctx, true, d, n, rs, progress,
                        [&](size_t lo, size_t cnt, uint8_t* inputs, uint8_t* aux, Bytes32* cv, int* rc) {
                            for (size_t k = 0; k < cnt; ++k) {
                                const OutputInfo& x = d[lo + k];
                                rc[k] = masp_host_output_assignment(x.esk.data(), x.payment_address.diversifier.bytes.data(), x.payment_address.pk_d.data(),
                                                                    x.rcm.data(), x.asset_type.identifier.data(), x.value, x.rcv.data(), 2,
                                                                    inputs + k * 32 * n_inputs_[MASP_HIP_OUTPUT], aux + k * 32 * (size_t)n_aux_[MASP_HIP_OUTPUT],
                                                                    cv[k].data());
                            }
                        },
                        nullptr, 0,  // "Output proofs are not self-checked" (the reference verifies Spend and Convert only)
                        [&](size_t i, const GrothProofBytes& zk, const Bytes32& cv) { out[i] = ValueProof{zk, cv}; });  // (cv_sum -= cv, :205: by `run`)
        return out;
    }
    std::vector<std::optional<ValueProof>> convert_proofs(SaplingProvingContext& ctx, const ConvertInfo* d, size_t n, const BlindingScalars* rs = nullptr,
                                                          const Progress& progress = nullptr) {
        std::vector<std::optional<ValueProof>> out(n);
        for (size_t i = 0; i < n; ++i) ctx.bsk_add(d[i].rcv, false);  // :228-234
        run<ConvertInfo>(MASP_HIP_CONVERT, ctx, false, d, n, rs, progress,
                         [&](size_t lo, size_t cnt, uint8_t* inputs, uint8_t* aux, Bytes32* cv, int* rc) {
                             std::vector<masp_host_convert_job> jobs(cnt);
                             for (size_t k = 0; k < cnt; ++k) {
                                 const ConvertInfo& x = d[lo + k];
                                 jobs[k] = masp_host_convert_job{x.allowed_conversion.generator.data(), x.value, x.anchor.data(), x.merkle_path.auth_path[0].data(),
                                                                 x.merkle_path.position, x.rcv.data(), inputs + k * 32 * n_inputs_[MASP_HIP_CONVERT],
                                                                 aux + k * 32 * (size_t)n_aux_[MASP_HIP_CONVERT], cv[k].data(), 0};
                             }
                             masp_host_convert_assignments(cnt, jobs.data(), 2);
                             for (size_t k = 0; k < cnt; ++k) rc[k] = jobs[k].rc;
                         },
                         [&](size_t i, const Bytes32& cv, uint8_t* pub) {  // sapling/prover.rs:256-263: cv, anchor
                             if (masp_host_point_uv(cv.data(), pub) != MASP_HOST_OK) throw Panic("cv: not a Jubjub point");
                             std::memcpy(pub + 64, d[i].anchor.data(), 32);
                         },
                         3,
                         [&](size_t i, const GrothProofBytes& zk, const Bytes32& cv) { out[i] = ValueProof{zk, cv}; });  // (cv_sum += cv, :272: by `run`)
        return out;
    }

    // Pay at load time what the first batches otherwise pay in their own latency: every slot's device scratch at its final size (one
    // launch sequence of batch_cap proofs per slot over a witness of zeros: the scratch does not depend on the witness) and the
    // page-locked slabs a call over that many descriptions keeps in flight.  No counterpart in the reference (bellperson allocates per
    // proof); = masp_amd/prover.py LocalTxProver.warm_up.
    void warm_up(size_t spends, size_t outputs = 0, size_t converts = 0) {
        const size_t want[3] = {spends, outputs, converts};
        for (int kind = 0; kind < 3; ++kind) {
            if (!want[kind]) continue;
            const size_t batches = (want[kind] + batch_cap_ - 1) / batch_cap_, slabs = std::min(batches, slots_ + 3);
            std::vector<uint8_t*> held;
            for (size_t i = 0; i < slabs; ++i) held.push_back(slab_take(kind));  // (page-locks what the pool does not hold yet)
            const size_t np = std::min(batch_cap_, want[kind]);
            std::memset(held[0], 0, np * 32 * (size_t)n_aux_[kind]);  // Montgomery zero = canonical zero
            std::vector<uint8_t> inputs(32 * (size_t)n_inputs_[kind], 0);
            inputs[0] = 1;  // ONE
            std::vector<masp_hip_job> jobs(np);
            for (size_t k = 0; k < np; ++k) {
                std::memset(&jobs[k], 0, sizeof jobs[k]);
                jobs[k].circuit = (uint32_t)kind;
                jobs[k].inputs = inputs.data();
                jobs[k].aux = held[0] + k * 32 * (size_t)n_aux_[kind];
                jobs[k].aux_form = MASP_HIP_AUX_MONTGOMERY;
                jobs[k].r[0] = 1;
                jobs[k].s[0] = 2;
            }
            std::vector<std::future<int>> calls;  // `slots` calls side by side: each lands on a slot of its own
            for (size_t c = 0; c < std::min(slots_, batches); ++c)
                calls.push_back(std::async(std::launch::async, [this, &jobs]() {
                    std::vector<uint8_t> proofs(GROTH_PROOF_SIZE * jobs.size());
                    return masp_hip_prove_batch(ctx_, jobs.size(), jobs.data(), proofs.data());
                }));
            int rc = MASP_HIP_OK;
            for (auto& c : calls) {
                const int r = c.get();
                if (r != MASP_HIP_OK) rc = r;
            }
            for (uint8_t* p : held) slab_give(kind, p);
            if (rc != MASP_HIP_OK) throw Panic(detail::hip_error(ctx_, rc, "warm_up"));
        }
    }

    // a uniform bls12_381::Scalar as 32 canonical bytes from the operating system's generator (the reference: OsRng, sapling/prover.rs:66)
    static Bytes32 random_scalar() {
        for (;;) {
            Bytes32 b;
            detail::os_random(b.data(), b.size());
            b[31] &= 0x7f;
            if (detail::fr_canonical(b.data())) return b;
        }
    }

  private:
    LocalTxProver(const uint8_t* spend, size_t spend_len, const uint8_t* output, size_t output_len, const uint8_t* convert, size_t convert_len, const Config& cfg)
        : cfg_(cfg) {
        if (cfg.expected) {  // verify_hash (lib.rs:439-487): size and BLAKE2b-512 of the whole stream; the reference panics on a mismatch
            const uint8_t* blob[3] = {spend, output, convert};
            const size_t len[3] = {spend_len, output_len, convert_len};
            const ExpectedParameters* e[3] = {&cfg.expected->spend, &cfg.expected->output, &cfg.expected->convert};
            static const char* names[3] = {"masp spend", "masp output", "masp convert"};
            for (int k = 0; k < 3; ++k) {
                if (len[k] != e[k]->bytes) throw Panic(std::string(names[k]) + " parameters: " + std::to_string(len[k]) + " bytes, expected " + std::to_string(e[k]->bytes));
                const char zero[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                detail::Blake2b512 h(zero);
                h.update(blob[k], len[k]);
                uint8_t d[64];
                h.finish(d);
                static const char* hexd = "0123456789abcdef";
                std::string got;
                for (uint8_t x : d) {
                    got += hexd[x >> 4];
                    got += hexd[x & 15];
                }
                if (got != e[k]->blake2b_hex) throw Panic(std::string(names[k]) + " parameters: BLAKE2b-512 digest " + got + ", expected " + e[k]->blake2b_hex);
            }
        }
        masp_hip_options opt;
        if (cfg.options) {
            opt = *cfg.options;
        } else {
            std::memset(&opt, 0, sizeof opt);  // every field: 0 = the default
        }
        opt.struct_size = sizeof(masp_hip_options);
        // one context over one GPU, or over every GPU listed: the library then deals the batches of a call to the devices from one queue and
        // finishes on the others when one fails (masp_hip.h: masp_hip_ctx_create_ex)
        std::vector<int> devs = cfg.devices.empty() ? std::vector<int>{cfg.device} : cfg.devices;
        int rc = masp_hip_ctx_create_ex(devs.data(), (int)devs.size(), &opt, &ctx_);
        if (rc != MASP_HIP_OK) throw Panic(detail::hip_error(nullptr, rc, "masp_hip_ctx_create_ex"));
        try {
            masp_hip_options got;
            std::memset(&got, 0, sizeof got);
            got.struct_size = sizeof got;
            if ((rc = masp_hip_ctx_get_options(ctx_, &got)) != MASP_HIP_OK) throw Panic(detail::hip_error(ctx_, rc, "masp_hip_ctx_get_options"));
            batch_cap_ = (size_t)got.batch_cap;
            slots_ = (size_t)got.slots * devs.size();   // batches that prove side by side: `slots` per device
            permits_.resize(cfg_.calls_in_flight ? cfg_.calls_in_flight : slots_ + 1);
            const uint8_t* params[3] = {spend, output, convert};
            const size_t lens[3] = {spend_len, output_len, convert_len};
            for (int k = 0; k < 3; ++k) load_circuit(k, params[k], lens[k]);
            // spend_vk / convert_vk: PreparedVerifyingKey (prover.rs:27-33, lib.rs:391-393); on the host for single proofs and for finding the
            // culprit of a failing batch, on the GPU for the batches
            for (int k : {MASP_HIP_SPEND, MASP_HIP_CONVERT}) {
                host_vk_[k] = masp_host_vk_prepare(params[k], lens[k]);
                if (!host_vk_[k]) throw Panic("verifying key does not parse");
                if ((rc = masp_hip_vk_prepare(ctx_, params[k], lens[k], &gpu_vk_[k])) != MASP_HIP_OK) throw Panic(detail::hip_error(ctx_, rc, "masp_hip_vk_prepare"));
            }
        } catch (...) {
            release();
            throw;
        }
    }
    void release() {  // (a verifying key goes before its context: masp_hip.h)
        for (int k = 0; k < 3; ++k) {
            for (uint8_t* p : pool_[k]) masp_hip_host_free(ctx_, p);
            for (uint8_t* p : pool_one_[k]) masp_hip_host_free(ctx_, p);
            pool_[k].clear();
            pool_one_[k].clear();
            if (gpu_vk_[k]) masp_hip_vk_free(gpu_vk_[k]);
            if (host_vk_[k]) masp_host_vk_free(host_vk_[k]);
            gpu_vk_[k] = nullptr;
            host_vk_[k] = nullptr;
        }
        if (ctx_) masp_hip_ctx_destroy(ctx_);
        ctx_ = nullptr;
    }

    // the circuit's static R1CS (what bellperson's KeypairAssembly collects) from libmasp_host, the CRS from the caller's bytes
    void load_circuit(int kind, const uint8_t* params, size_t len) {
        void* h = masp_host_circuit_setup(kind);
        if (!h) throw Panic("masp_host_circuit_setup failed");
        uint32_t counts[6];
        masp_host_circuit_counts(h, counts);
        n_inputs_[kind] = counts[0];
        n_aux_[kind] = counts[1];
        std::vector<uint32_t> rowptr[3], col[3];
        std::vector<uint8_t> coef[3];
        for (int m = 0; m < 3; ++m) {
            rowptr[m].resize((size_t)counts[2] + 1);
            col[m].resize(counts[3 + m]);
            coef[m].resize((size_t)32 * counts[3 + m]);
            masp_host_circuit_matrix(h, m, rowptr[m].data(), col[m].data(), coef[m].data());
        }
        masp_host_circuit_free(h);
        masp_hip_r1cs cs;
        cs.n_inputs = counts[0];
        cs.n_aux = counts[1];
        cs.n_constraints = counts[2];
        cs.a_rowptr = rowptr[0].data(); cs.a_col = col[0].data(); cs.a_coef = coef[0].data();
        cs.b_rowptr = rowptr[1].data(); cs.b_col = col[1].data(); cs.b_coef = coef[1].data();
        cs.c_rowptr = rowptr[2].data(); cs.c_col = col[2].data(); cs.c_coef = coef[2].data();
        const int rc = masp_hip_circuit_load(ctx_, (uint32_t)kind, params, len, &cs);
        if (rc != MASP_HIP_OK) throw Panic(detail::hip_error(ctx_, rc, "masp_hip_circuit_load"));  // the reference panics on undecodable parameters (lib.rs:337)
    }

    // page-locked slabs of aux assignments per circuit — room for batch_cap of them, or for one (the trait's single-description methods
    // must not page-lock 0.8 GB for one Spend): the synthesizer writes where the DMA engine reads
    uint8_t* slab_take(int kind, bool single = false) {
        std::vector<uint8_t*>& pool = single ? pool_one_[kind] : pool_[kind];
        {
            std::lock_guard<std::mutex> g(pool_mu_);
            if (!pool.empty()) {
                uint8_t* p = pool.back();
                pool.pop_back();
                return p;
            }
        }
        void* p = masp_hip_host_alloc(ctx_, (single ? 1 : batch_cap_) * 32 * (size_t)n_aux_[kind]);
        if (!p) throw Panic("masp_hip_host_alloc failed");
        return static_cast<uint8_t*>(p);
    }
    void slab_give(int kind, uint8_t* p, bool single = false) {
        std::lock_guard<std::mutex> g(pool_mu_);
        (single ? pool_one_[kind] : pool_[kind]).push_back(p);
    }

    // One circuit's descriptions in batches of batch_cap: the host threads synthesise batch k + 1 while batches <= k prove — up to
    // slots + 3 batches handed over at a time: `slots` of them on a slot of the context each, one inside the library waiting for the next
    // free slot, two behind it, so that a slot
    // that finishes never waits for a synthesis (each waiting batch holds its page-locked slab: 0.8 GB for 256 Spends); results
    // committed in description order.
    //   synth(lo, cnt, inputs, aux, cv, rc): witnesses of descriptions [lo, lo + cnt) — called from several threads on disjoint ranges
    //   public_input(i, cv, pub) / n_public: the statement of description i (nullptr: no self-check)
    //   commit(i, zkproof, cv): description i proved (and verified); called in order of i, from the calling thread; the context's
    //   cv_sum takes the batch's commitments right after (+ or -: subtract_cv)
    template <class Info, class Synth, class PublicInput, class Commit>
    void run(int kind, SaplingProvingContext& ctx, bool subtract_cv, const Info*, size_t n, const BlindingScalars* rs, const Progress& progress, Synth synth,
             PublicInput public_input,
             uint32_t n_public, Commit commit) {
        struct Batch {
            size_t lo = 0, cnt = 0;
            uint8_t* aux = nullptr;
            std::vector<uint8_t> inputs, proofs;
            std::vector<Bytes32> cv;
            std::vector<int> rc;        // synthesis: MASP_HOST_OK or why the description cannot be proved
            std::vector<char> valid;    // after proving: the proof exists and passed its self-check
            Bytes32 cv_partial{};       // the sum of the value commitments of its valid proofs (one point addition per batch on the caller's thread)
            bool any_valid = false;
            size_t ticket = 0;          // its place in the queue for a slot of the context
            bool ticketed = false;
            double t[6] = {0, 0, 0, 0, 0, 0};  // (trace) seconds since the call began: synthesis from / to, slot taken, proved, verified, committed
            std::future<void> proving;
        };
        const size_t cap = batch_cap_, nin = n_inputs_[kind], naux = n_aux_[kind];
        const bool single = n == 1;
        const auto t_begin = std::chrono::steady_clock::now();
        auto now = [t_begin]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
        const unsigned threads = cfg_.threads ? cfg_.threads : detail::effective_cpus();
        std::deque<std::unique_ptr<Batch>> flying;
        auto land = [&]() {  // the oldest batch: wait, commit in order, recycle its slab
            std::unique_ptr<Batch> b = std::move(flying.front());
            flying.pop_front();
            struct Give {
                LocalTxProver* p; int kind; uint8_t* s; bool single;
                ~Give() { p->slab_give(kind, s, single); }
            } give{this, kind, b->aux, single};
            b->proving.get();  // (rethrows a Panic of the proving thread)
            for (size_t k = 0; k < b->cnt; ++k)
                if (b->valid[k]) {
                    GrothProofBytes zk;
                    std::memcpy(zk.data(), &b->proofs[GROTH_PROOF_SIZE * k], GROTH_PROOF_SIZE);
                    commit(b->lo + k, zk, b->cv[k]);
                }
            if (b->any_valid) ctx.cv_add(b->cv_partial, subtract_cv);  // cv_sum after the self-check (sapling/prover.rs:154,205,272)
            if (progress) progress(b->lo + b->cnt, n);
            b->t[5] = now();
            if (cfg_.trace)
                std::fprintf(stderr, "batch at %zu (%zu): synthesis %.3f - %.3f, slot %.3f, proved %.3f, verified %.3f, committed %.3f\n", b->lo, b->cnt, b->t[0], b->t[1], b->t[2],
                             b->t[3], b->t[4], b->t[5]);
        };
        try {
            for (size_t lo = 0; lo < n; lo += cap) {
                std::unique_ptr<Batch> b(new Batch);
                b->lo = lo;
                b->cnt = std::min(cap, n - lo);
                b->aux = slab_take(kind, single);
                b->inputs.resize(b->cnt * 32 * nin);
                b->proofs.resize(b->cnt * GROTH_PROOF_SIZE);
                b->cv.resize(b->cnt);
                b->rc.assign(b->cnt, MASP_HOST_OK);
                b->valid.assign(b->cnt, 0);
                b->ticket = permits_.ticket();
                b->ticketed = true;
                Batch* B = b.get();
                flying.push_back(std::move(b));
                // ---- synthesis: groups of 16 witnesses per native call (their Merkle blocks side by side), dealt to the threads
                B->t[0] = now();
                {
                    constexpr size_t GROUP = 16;
                    const size_t groups = (B->cnt + GROUP - 1) / GROUP;
                    std::atomic<size_t> next{0};
                    auto worker = [&]() {
                        for (size_t g; (g = next.fetch_add(1)) < groups;) {
                            const size_t k0 = g * GROUP, c = std::min(GROUP, B->cnt - k0);
                            synth(B->lo + k0, c, B->inputs.data() + k0 * 32 * nin, B->aux + k0 * 32 * naux, B->cv.data() + k0, B->rc.data() + k0);
                        }
                    };
                    std::vector<std::thread> ts;
                    for (unsigned t = 1; t < std::min<size_t>(threads, groups); ++t) ts.emplace_back(worker);
                    worker();
                    for (auto& t : ts) t.join();
                }
                B->t[1] = now();
                // ---- proving + self-check of this batch on a thread of its own
                B->proving = std::async(std::launch::async, [this, B, kind, nin, naux, rs, public_input, n_public, now]() {
                    std::vector<masp_hip_job> jobs;
                    std::vector<size_t> idx;
                    for (size_t k = 0; k < B->cnt; ++k) {
                        if (B->rc[k] != MASP_HOST_OK) continue;  // Err(()): invalid diversifier (sapling/prover.rs:84) — nothing to prove
                        masp_hip_job j;
                        std::memset(&j, 0, sizeof j);
                        j.circuit = (uint32_t)kind;
                        j.inputs = B->inputs.data() + k * 32 * nin;
                        j.aux = B->aux + k * 32 * naux;
                        j.aux_form = MASP_HIP_AUX_MONTGOMERY;
                        const Bytes32 r = rs ? rs[B->lo + k].r : random_scalar(), s = rs ? rs[B->lo + k].s : random_scalar();
                        std::memcpy(j.r, r.data(), 32);
                        std::memcpy(j.s, s.data(), 32);
                        jobs.push_back(j);
                        idx.push_back(k);
                    }
                    std::vector<uint8_t> proofs(GROTH_PROOF_SIZE * jobs.size());
                    int rc = MASP_HIP_OK;
                    permits_.acquire(B->ticket);   // (a batch with nothing to prove takes its turn too: the tickets are consecutive)
                    B->t[2] = now();
                    if (!jobs.empty()) rc = masp_hip_prove_batch(ctx_, jobs.size(), jobs.data(), proofs.data());
                    permits_.release();
                    B->t[3] = now();
                    if (jobs.empty()) return;
                    if (rc != MASP_HIP_OK) throw Panic(detail::hip_error(ctx_, rc, "proving should not fail"));  // .expect(..) at sapling/prover.rs:117,202,252
                    std::vector<char> ok(jobs.size(), 1);
                    if constexpr (!std::is_same<PublicInput, std::nullptr_t>::value) {
                        if (cfg_.self_verify) {
                            std::vector<uint8_t> pub(jobs.size() * 32 * n_public);
                            for (size_t q = 0; q < jobs.size(); ++q) public_input(B->lo + idx[q], B->cv[idx[q]], &pub[q * 32 * n_public]);
                            bool all = false;
                            if (jobs.size() > 1) {  // one random linear combination on the GPU (bellman's verify_proofs_batch, sapling/verifier/batch.rs:201-239)
                                std::vector<uint8_t> z(16 * jobs.size());
                                detail::os_random(z.data(), z.size());
                                int valid = 0;
                                rc = masp_hip_verify_batch(ctx_, gpu_vk_[kind], jobs.size(), proofs.data(), pub.data(), n_public, z.data(), &valid);
                                if (rc != MASP_HIP_OK) throw Panic(detail::hip_error(ctx_, rc, "masp_hip_verify_batch"));
                                all = valid == 1;
                            }
                            if (!all)  // a lone proof, or the batch holds a proof that does not verify: each one on the host (verify_proof, :148 / :266)
                                for (size_t q = 0; q < jobs.size(); ++q)
                                    ok[q] = masp_host_vk_verify(host_vk_[kind], &proofs[GROTH_PROOF_SIZE * q], &pub[q * 32 * n_public], n_public) == 1;
                        }
                    }
                    B->t[4] = now();
                    std::vector<uint8_t> cvs;
                    for (size_t q = 0; q < jobs.size(); ++q) {
                        std::memcpy(&B->proofs[GROTH_PROOF_SIZE * idx[q]], &proofs[GROTH_PROOF_SIZE * q], GROTH_PROOF_SIZE);
                        B->valid[idx[q]] = ok[q];
                        if (ok[q]) cvs.insert(cvs.end(), B->cv[idx[q]].begin(), B->cv[idx[q]].end());
                    }
                    if (!cvs.empty()) {  // one native call for the batch's commitments (a point addition per proof on the caller's thread: 30 ms per 256)
                        Bytes32 identity{};
                        identity[0] = 1;
                        if (masp_host_jubjub_sum(identity.data(), cvs.data(), cvs.size() / 32, nullptr, B->cv_partial.data()) != MASP_HOST_OK) throw Panic("cv: not a Jubjub point");
                        B->any_valid = true;
                    }
                });
                while (flying.size() > slots_ + 3) land();
            }
            while (!flying.empty()) land();
        } catch (...) {
            for (auto& b : flying) {  // let the proving threads finish with their buffers before those go away
                if (b->proving.valid()) {
                    b->proving.wait();
                } else if (b->ticketed) {  // never handed over: its turn must still pass, or every later batch of this prover waits for it
                    permits_.acquire(b->ticket);
                    permits_.release();
                }
                slab_give(kind, b->aux, single);
            }
            throw;
        }
    }

    Config cfg_;
    masp_hip_ctx* ctx_ = nullptr;
    size_t batch_cap_ = 256, slots_ = 4;
    uint32_t n_inputs_[3] = {0, 0, 0}, n_aux_[3] = {0, 0, 0};
    void* host_vk_[3] = {nullptr, nullptr, nullptr};
    masp_hip_vk* gpu_vk_[3] = {nullptr, nullptr, nullptr};
    std::mutex pool_mu_;
    std::vector<uint8_t*> pool_[3], pool_one_[3];
    detail::FifoPermits permits_;
};

}  // namespace masp
#endif

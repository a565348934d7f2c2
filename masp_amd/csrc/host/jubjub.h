// Jubjub (twisted Edwards  -u^2 + v^2 = 1 + d u^2 v^2  over the BLS12-381 scalar field), BLAKE2s-256 with
// personalisation, and the native (out-of-circuit) Sapling/MASP primitives the prover needs to compute
// witness values.  Restates what the reference takes from `nam-jubjub 1.10.1-nam.1`, `blake2s_simd` and
//   masp_primitives/src/sapling/group_hash.rs:15-43, sapling/pedersen_hash.rs:31-117,
//   masp_primitives/src/constants.rs:50-251 (generators, derivations pinned by its tests :323-374),
//   masp_primitives/src/asset_type.rs:30-102, sapling.rs:54-85,198-223,334-355,453-477,796-854,
//   masp_primitives/src/convert.rs:39-64          (all paths under /root/reference).
#pragma once
#include <array>
#include <stdexcept>
#include <vector>

#include "fr.h"

namespace masp_host {

// ---------------------------------------------------------------------------------------- BLAKE2s
class Blake2s {
  public:
    explicit Blake2s(const uint8_t* personal8 = nullptr, uint32_t outlen = 32) {
        static const uint32_t IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
        for (int i = 0; i < 8; ++i) h_[i] = IV[i];
        h_[0] ^= 0x01010000u ^ outlen;
        if (personal8) {
            h_[6] ^= le32(personal8);
            h_[7] ^= le32(personal8 + 4);
        }
        outlen_ = outlen;
    }
    void update(const uint8_t* data, size_t n) {
        while (n) {
            if (buflen_ == 64) {
                t_ += 64;
                compress(false);
                buflen_ = 0;
            }
            size_t take = std::min<size_t>(n, 64 - buflen_);
            memcpy(buf_ + buflen_, data, take);
            buflen_ += take;
            data += take;
            n -= take;
        }
    }
    void finalize(uint8_t* out) {
        t_ += buflen_;
        memset(buf_ + buflen_, 0, 64 - buflen_);
        compress(true);
        for (uint32_t i = 0; i < outlen_; ++i) out[i] = (uint8_t)(h_[i / 4] >> (8 * (i % 4)));
    }

  private:
    static uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void compress(bool last) {
        static const uint32_t IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
        static const uint8_t S[10][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
                                          {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
                                          {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
                                          {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
                                          {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint32_t m[16], v[16];
        for (int i = 0; i < 16; ++i) m[i] = le32(buf_ + 4 * i);
        for (int i = 0; i < 8; ++i) {
            v[i] = h_[i];
            v[i + 8] = IV[i];
        }
        v[12] ^= (uint32_t)t_;
        v[13] ^= (uint32_t)(t_ >> 32);
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
            v[a] = v[a] + v[b] + x;
            v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];
            v[b] = rotr(v[b] ^ v[c], 12);
            v[a] = v[a] + v[b] + y;
            v[d] = rotr(v[d] ^ v[a], 8);
            v[c] = v[c] + v[d];
            v[b] = rotr(v[b] ^ v[c], 7);
        };
        for (int r = 0; r < 10; ++r) {
            const uint8_t* s = S[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);
            G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);
            G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);
            G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]);
            G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; ++i) h_[i] ^= v[i] ^ v[i + 8];
    }
    uint32_t h_[8];
    uint8_t buf_[64];
    size_t buflen_ = 0;
    uint64_t t_ = 0;
    uint32_t outlen_;
};

// ---------------------------------------------------------------------------------------- Jubjub
inline const Fr& edwards_d() {  // /root/reference/masp_proofs/src/constants.rs:10-18
    static Fr d = Fr::from_limbs(0x01065fd6d6343eb1ull, 0x292d7f6d37579d26ull, 0xf5fd9207e6bd7fd4ull, 0x2a9318e74bfa2b48ull);
    return d;
}
inline const Fr& montgomery_a() {  // constants.rs:21-29
    static Fr a = Fr::from_limbs(0xa002, 0, 0, 0);
    return a;
}
inline const Fr& montgomery_scale() {  // constants.rs:32-40
    static Fr s = Fr::from_limbs(0x8f4535f7cf82b8d9ull, 0xce4069703da88abdull, 0x31de341e77d764e5ull, 0x2762de61e862645eull);
    return s;
}

struct JAffine {
    Fr u, v;
};
// extended coordinates (U : V : Z : T), T = U V / Z; unified a = -1 addition (complete: d is a non-square)
struct JPoint {
    Fr U, V, Z, T;
    static JPoint identity() { return {Fr::zero(), Fr::one(), Fr::one(), Fr::zero()}; }
    static JPoint from_affine(const JAffine& a) { return {a.u, a.v, Fr::one(), a.u * a.v}; }
    JPoint add(const JPoint& o) const {
        Fr d2 = edwards_d().dbl();
        Fr A = (V - U) * (o.V - o.U);
        Fr B = (V + U) * (o.V + o.U);
        Fr C = T * d2 * o.T;
        Fr D = (Z * o.Z).dbl();
        Fr E = B - A, F = D - C, G = D + C, H = B + A;
        return {E * F, G * H, F * G, E * H};
    }
    // dedicated doubling for a = -1 (4 squarings + 4 products; equals add(*this))
    JPoint dbl() const {
        const Fr A = U.square(), B = V.square(), C = Z.square().dbl();
        const Fr E = (U + V).square() - A - B, G = B - A, F = G - C, H = (A + B).neg();
        return {E * F, G * H, F * G, E * H};
    }
    JPoint neg() const { return {U.neg(), V, Z, T.neg()}; }
    // this + (u, v) given as (v - u, v + u, 2 d u v): 7 products
    struct Niels {
        Fr u, v, vmu, vpu, t2d;
        static Niels from(const JAffine& a) { return {a.u, a.v, a.v - a.u, a.v + a.u, (a.u * a.v * edwards_d()).dbl()}; }
    };
    JPoint add_niels(const Niels& n, bool negate = false) const {
        const Fr A = (V - U) * (negate ? n.vpu : n.vmu), B = (V + U) * (negate ? n.vmu : n.vpu);
        Fr C = T * n.t2d;
        if (negate) C = C.neg();
        const Fr D = Z.dbl();
        const Fr E = B - A, F = D - C, G = D + C, H = B + A;
        return {E * F, G * H, F * G, E * H};
    }
    // scalar: 32 bytes little-endian (any 256-bit integer); 4-bit windows from the top non-zero nibble down
    JPoint mul(const uint8_t* scalar_le) const {
        int top = 63;
        while (top >= 0 && ((scalar_le[top / 2] >> (4 * (top & 1))) & 15) == 0) --top;
        if (top < 0) return identity();
        JPoint tab[16];
        tab[1] = *this;
        for (int k = 2; k < 16; ++k) tab[k] = (k & 1) ? tab[k - 1].add(*this) : tab[k / 2].dbl();
        JPoint r = tab[(scalar_le[top / 2] >> (4 * (top & 1))) & 15];
        for (int i = top - 1; i >= 0; --i) {
            r = r.dbl().dbl().dbl().dbl();
            const unsigned nib = (scalar_le[i / 2] >> (4 * (i & 1))) & 15;
            if (nib) r = r.add(tab[nib]);
        }
        return r;
    }
    JPoint mul_u64(uint64_t k) const {
        uint8_t s[32] = {0};
        for (int i = 0; i < 8; ++i) s[i] = (uint8_t)(k >> (8 * i));
        return mul(s);
    }
    JPoint mul_by_cofactor() const { return dbl().dbl().dbl(); }
    bool is_identity() const { return U.is_zero() && V == Z; }
    JAffine to_affine() const {
        Fr zi;
        Z.invert(zi);
        return {U * zi, V * zi};
    }
    bool eq(const JPoint& o) const { return U * o.Z == o.U * Z && V * o.Z == o.V * Z; }
    // 32 bytes: v little-endian with the parity of u in bit 255 (`to_bytes`)
    void to_bytes(uint8_t* out) const {
        JAffine a = to_affine();
        a.v.to_bytes(out);
        if (a.u.is_odd()) out[31] |= 0x80;
    }
    static bool from_bytes(JPoint& out, const uint8_t* in) {
        uint8_t tmp[32];
        memcpy(tmp, in, 32);
        bool sign = tmp[31] >> 7;
        tmp[31] &= 0x7f;
        Fr v;
        if (!Fr::from_bytes(v, tmp)) return false;
        // u^2 = (v^2 - 1) / (1 + d v^2)
        Fr v2 = v.square();
        Fr den;
        if (!(Fr::one() + edwards_d() * v2).invert(den)) return false;
        Fr u;
        if (!((v2 - Fr::one()) * den).sqrt(u)) return false;
        if (u.is_odd() != sign) u = u.neg();
        // u = 0 with the sign bit set has no valid encoding (jubjub rejects "negative zero")
        if (u.is_zero() && sign) return false;
        out = from_affine({u, v});
        return true;
    }
};

// ---------------------------------------------------------------------------------------- group hash, generators
static const char GH_FIRST_BLOCK[65] = "096b36a5804bfacef1691e173c366a47ff5ba84a44f26ddd7e8d9f79d5b42df0";

inline bool group_hash(JPoint& out, const uint8_t* tag, size_t taglen, const char* personal8) {
    Blake2s h((const uint8_t*)personal8);
    h.update((const uint8_t*)GH_FIRST_BLOCK, 64);
    h.update(tag, taglen);
    uint8_t d[32];
    h.finalize(d);
    JPoint p;
    if (!JPoint::from_bytes(p, d)) return false;
    p = p.mul_by_cofactor();
    if (p.is_identity()) return false;
    out = p;
    return true;
}
inline JPoint find_group_hash(const uint8_t* m, size_t mlen, const char* personal8) {
    std::vector<uint8_t> tag(m, m + mlen);
    tag.push_back(0);
    for (;;) {
        JPoint p;
        if (group_hash(p, tag.data(), tag.size(), personal8)) return p;
        ++tag[mlen];
    }
}
struct Generators {
    JPoint proof_generation_key, note_commitment_randomness, nullifier_position, value_commitment_randomness, spending_key;
    JPoint pedersen[6];
};
inline const Generators& generators() {
    static Generators g = [] {
        Generators x;
        x.proof_generation_key = find_group_hash(nullptr, 0, "MASP__H_");
        x.note_commitment_randomness = find_group_hash((const uint8_t*)"r", 1, "MASP__PH");
        x.nullifier_position = find_group_hash(nullptr, 0, "MASP__J_");
        x.value_commitment_randomness = find_group_hash((const uint8_t*)"r", 1, "MASP__r_");
        x.spending_key = find_group_hash(nullptr, 0, "MASP__G_");
        for (uint32_t m = 0; m < 6; ++m) {
            uint8_t le[4] = {(uint8_t)m, 0, 0, 0};
            x.pedersen[m] = find_group_hash(le, 4, "MASP__PH");
        }
        return x;
    }();
    return g;
}

// ---------------------------------------------------------------------------------------- Pedersen hash
struct Personalization {
    bool note_commitment;
    unsigned depth;  // MerkleTree(depth) when !note_commitment
    std::array<bool, 6> bits() const {
        std::array<bool, 6> b;
        for (int i = 0; i < 6; ++i) b[i] = note_commitment ? true : ((depth >> i) & 1);
        return b;
    }
};
// (k + 1) * 16^w * G_s for k < 4, w < 63, s < 6 in affine form: the summands of the Pedersen hash.  Shared by the native hash
// below and by the witness generator's projective pre-pass of the in-circuit hash (circuits.h).
struct PedersenWindows {
    JPoint::Niels e[6][63][4];
};
inline const PedersenWindows& pedersen_windows() {
    static const PedersenWindows* t = [] {
        PedersenWindows* x = new PedersenWindows;
        for (int s = 0; s < 6; ++s) {
            JPoint gen = generators().pedersen[s];
            for (int w = 0; w < 63; ++w) {
                JPoint p = gen;
                for (int k = 0; k < 4; ++k) {
                    x->e[s][w][k] = JPoint::Niels::from(p.to_affine());
                    p = p.add(gen);
                }
                gen = gen.dbl().dbl().dbl().dbl();
            }
        }
        return x;
    }();
    return *t;
}
// chunk (a,b,c) of window j contributes (1 + a + 2b) * (-1)^c * 2^(4j); 63 chunks per generator
inline JPoint pedersen_hash(const Personalization& pers, const std::vector<bool>& msg) {
    std::vector<bool> bits;
    for (bool b : pers.bits()) bits.push_back(b);
    bits.insert(bits.end(), msg.begin(), msg.end());
    const PedersenWindows& T = pedersen_windows();
    JPoint result = JPoint::identity();
    size_t pos = 0;
    int seg = 0;
    while (pos < bits.size()) {
        if (seg >= 6) throw std::runtime_error("pedersen_hash: message too long");
        for (int j = 0; j < 63 && pos < bits.size(); ++j) {
            bool a = bits[pos++];
            bool b = pos < bits.size() ? bits[pos++] : false;
            bool c = pos < bits.size() ? bits[pos++] : false;
            result = result.add_niels(T.e[seg][j][(a ? 1 : 0) + (b ? 2 : 0)], c);
        }
        ++seg;
    }
    return result;
}
inline std::vector<bool> bytes_to_bits_le(const uint8_t* data, size_t n) {
    std::vector<bool> out;
    for (size_t i = 0; i < n; ++i)
        for (int b = 0; b < 8; ++b) out.push_back((data[i] >> b) & 1);
    return out;
}

// ---------------------------------------------------------------------------------------- MASP primitives
// asset generator (cofactor NOT cleared) from a 32-byte identifier; false if the identifier is invalid
inline bool asset_generator(JPoint& out, const uint8_t* identifier32) {
    Blake2s h((const uint8_t*)"MASP__v_");
    h.update(identifier32, 32);
    uint8_t d[32];
    h.finalize(d);
    JPoint p;
    if (!JPoint::from_bytes(p, d)) return false;
    if (p.mul_by_cofactor().is_identity()) return false;
    out = p;
    return true;
}
// AssetType::new(name): smallest nonce giving a valid identifier
inline bool asset_identifier(uint8_t* out32, const uint8_t* name, size_t len) {
    for (unsigned nonce = 0; nonce < 256; ++nonce) {
        Blake2s h((const uint8_t*)"MASP__t_");
        h.update((const uint8_t*)GH_FIRST_BLOCK, 64);
        h.update(name, len);
        uint8_t n8 = (uint8_t)nonce;
        h.update(&n8, 1);
        uint8_t id[32];
        h.finalize(id);
        JPoint p;
        if (asset_generator(p, id)) {
            memcpy(out32, id, 32);
            return true;
        }
    }
    return false;
}
// cv = [value]([8] asset_generator) + [rcv] G_vcr
inline JPoint value_commitment(const JPoint& asset_gen, uint64_t value, const uint8_t* rcv_le) {
    return asset_gen.mul_by_cofactor().mul_u64(value).add(generators().value_commitment_randomness.mul(rcv_le));
}
inline void crh_ivk(uint8_t* out32, const JPoint& ak, const JPoint& nk) {
    uint8_t a[32], n[32];
    ak.to_bytes(a);
    nk.to_bytes(n);
    Blake2s h((const uint8_t*)"MASP_ivk");
    h.update(a, 32);
    h.update(n, 32);
    h.finalize(out32);
    out32[31] &= 0x07;
}
// cm = PedersenHash(NoteCommitment, asset_gen | value | g_d | pk_d) + [rcm] G_ncr
inline JPoint note_commitment(const JPoint& asset_gen, uint64_t value, const JPoint& g_d, const JPoint& pk_d, const uint8_t* rcm_le) {
    uint8_t buf[104];
    asset_gen.to_bytes(buf);
    for (int i = 0; i < 8; ++i) buf[32 + i] = (uint8_t)(value >> (8 * i));
    g_d.to_bytes(buf + 40);
    pk_d.to_bytes(buf + 72);
    JPoint h = pedersen_hash({true, 0}, bytes_to_bits_le(buf, 104));
    return generators().note_commitment_randomness.mul(rcm_le).add(h);
}
inline void nullifier(uint8_t* out32, const JPoint& cm, uint64_t position, const JPoint& nk) {
    JPoint rho = cm.add(generators().nullifier_position.mul_u64(position));
    uint8_t a[32], b[32];
    nk.to_bytes(a);
    rho.to_bytes(b);
    Blake2s h((const uint8_t*)"MASP__nf");
    h.update(a, 32);
    h.update(b, 32);
    h.finalize(out32);
}
// parent = u-coordinate of PedersenHash(MerkleTree(depth), lhs[0..255) | rhs[0..255))
inline Fr merkle_hash(unsigned depth, const Fr& lhs, const Fr& rhs) {
    uint8_t l[32], r[32];
    lhs.to_bytes(l);
    rhs.to_bytes(r);
    std::vector<bool> bits;
    for (int i = 0; i < 255; ++i) bits.push_back((l[i / 8] >> (i % 8)) & 1);
    for (int i = 0; i < 255; ++i) bits.push_back((r[i / 8] >> (i % 8)) & 1);
    return pedersen_hash({false, depth}, bits).to_affine().u;
}

}  // namespace masp_host

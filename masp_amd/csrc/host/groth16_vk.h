// PreparedVerifyingKey of the host-side Groth16 verifier (shared by libmasp_host's masp_host_vk_* entry points and by
// libmasp_hip's GPU batch verifier, which keeps the public-input combination and the final exponentiation on the host).
#pragma once
#include <vector>

#include "fr.h"
#include "pairing.h"

namespace masp_host {

// PreparedVerifyingKey (lib.rs:391-393): the Miller value of (alpha, beta) is computed once; each IC point carries a
// table of d * 16^w * IC (d = 1..15, w = 0..63) so that the public-input combination costs 64 additions per input.
struct PreparedVk {
    bls::Fp12 alpha_beta;
    bls::G2A gamma, delta;
    std::vector<bls::G1A> ic;
    std::vector<std::vector<bls::G1J>> ic_tab;  // [input][w * 15 + d - 1]
    void build_tables() {
        ic_tab.resize(ic.size());
        for (size_t i = 1; i < ic.size(); ++i) {
            auto& t = ic_tab[i];
            t.resize(64 * 15);
            bls::G1J base = bls::G1J::from(ic[i]);
            for (int w = 0; w < 64; ++w) {
                t[w * 15] = base;
                for (int d = 2; d <= 15; ++d) t[w * 15 + d - 1] = t[w * 15 + d - 2].add(base);
                base = t[w * 15 + 7].dbl();  // 16 * base
            }
        }
    }
    // IC_i * k for a 32-byte little-endian scalar
    bls::G1J ic_mul(size_t i, const uint8_t* k32) const {
        bls::G1J r = bls::G1J::inf();
        const auto& t = ic_tab[i];
        for (int w = 0; w < 64; ++w) {
            int d = (k32[w >> 1] >> ((w & 1) * 4)) & 15;
            if (d) r = r.add(t[w * 15 + d - 1]);
        }
        return r;
    }
};

// vk bytes (the verifying-key prefix of a Parameters file) -> PreparedVk; false if malformed
inline bool prepare_vk(PreparedVk& vk, const uint8_t* params, size_t len) {
    if (len < 868) return false;
    bls::G1A alpha, delta1;
    bls::G2A beta;
    if (!bls::g1_uncompressed(alpha, params) || !bls::g2_uncompressed(beta, params + 192) || !bls::g2_uncompressed(vk.gamma, params + 384) ||
        !bls::g1_uncompressed(delta1, params + 576) || !bls::g2_uncompressed(vk.delta, params + 672))
        return false;
    uint32_t n = ((uint32_t)params[864] << 24) | (params[865] << 16) | (params[866] << 8) | params[867];
    if ((size_t)n * 96 + 868 > len || n == 0) return false;
    vk.ic.resize(n);
    for (uint32_t i = 0; i < n; ++i)
        if (!bls::g1_uncompressed(vk.ic[i], params + 868 + 96 * (size_t)i)) return false;
    vk.alpha_beta = bls::miller(alpha, beta);
    vk.build_tables();
    return true;
}

// The part of `verify_proofs_batch` that does not grow with the number of proofs' pairings.  With random z_i:
//   prod_i e(z_i A_i, B_i) == e(alpha, beta)^(sum z_i) e(sum_i z_i acc_i, gamma) e(sum_i z_i C_i, delta)
// f_pairs = prod_i ML(z_i A_i, B_i) (conjugated Miller values, pairing.h's convention) and csum = sum_i z_i C_i come from the
// caller (host loop in libmasp_host, GPU kernels in libmasp_hip); here: the public-input combination
// sum_i z_i acc_i = (sum z_i) IC_0 + sum_j (sum_i z_i x_ij) IC_j, the two remaining pairs, alpha_beta^(sum z), the final
// exponentiation.  z: n x 16 B (bit 0 of every z_i forced to 1).  1 = all valid, 0 = at least one invalid, -3 = an input >= r.
inline int batch_verify_finish(const PreparedVk& vk, size_t n, const uint8_t* public_inputs, uint32_t n_public, const uint8_t* z,
                               const bls::Fp12& f_pairs, const bls::G1A& csum) {
    std::vector<Fr> coef(n_public + 1, Fr::zero());  // sum_i z_i * input_ij  (j = 0: the constant ONE)
    for (size_t i = 0; i < n; ++i) {
        uint8_t zi[32] = {0};
        memcpy(zi, z + 16 * i, 16);
        zi[0] |= 1;
        Fr zf;
        Fr::from_bytes(zf, zi);
        coef[0] = coef[0] + zf;
        for (uint32_t j = 0; j < n_public; ++j) {
            Fr in;
            if (!Fr::from_bytes(in, public_inputs + 32 * ((size_t)i * n_public + j))) return -3;
            coef[j + 1] = coef[j + 1] + zf * in;
        }
    }
    uint8_t k32[32];
    coef[0].to_bytes(k32);
    bls::G1J acc = bls::G1J::from(vk.ic[0]).mul_le(k32, 256);
    for (uint32_t j = 0; j < n_public; ++j) {
        uint8_t kj[32];
        coef[j + 1].to_bytes(kj);
        acc = acc.add(vk.ic_mul(j + 1, kj));
    }
    bls::G1A nacc = acc.affine(), nc = csum;
    nacc.y = nacc.y.neg();
    nc.y = nc.y.neg();
    std::vector<bls::MillerPair> pairs;
    if (!nacc.inf) pairs.emplace_back(nacc, vk.gamma);
    if (!nc.inf) pairs.emplace_back(nc, vk.delta);
    // alpha_beta^(sum z)
    bls::Fp12 ab = bls::Fp12::one();
    for (int bit = 255; bit >= 0; --bit) {
        ab = ab.sq();
        if ((k32[bit >> 3] >> (bit & 7)) & 1) ab = ab * vk.alpha_beta;
    }
    bls::Fp12 f = f_pairs * bls::multi_miller(pairs) * ab.conj();
    return bls::final_exp(f) == bls::Fp12::one() ? 1 : 0;
}

}  // namespace masp_host

// Shared host-side modular arithmetic for the two prime fields of the witness generator and the self-verifier:
// Montgomery multiplication (single-pass CIOS; valid for moduli whose top limb has its most significant bit clear, which
// holds for both the 255-bit BLS12-381 scalar field and the 381-bit base field) and modular inversion by batched
// divsteps (Bernstein-Yang "safegcd" in its variable-time form: 62 division steps are decided on the low words alone and
// applied to the full-width values as one 2x2 integer matrix).
// Witness synthesis of one Spend performs ~7 300 field inversions (affine Jubjub additions in the Pedersen-hash and
// scalar-multiplication gadgets: /root/reference/masp_proofs/src/circuit/ecc.rs allocates the quotient of every addition),
// so inversion speed, not multiplication speed, sets the synthesis time.
#pragma once
#include <cstdint>
#include <cstring>

namespace masp_host {

typedef unsigned __int128 u128;
typedef __int128 i128;

template <int N>
static inline void mont_mul_n(uint64_t* out, const uint64_t* a, const uint64_t* b, const uint64_t* p, uint64_t inv) {
    uint64_t t[N];
#pragma GCC unroll 8
    for (int j = 0; j < N; ++j) t[j] = 0;
#pragma GCC unroll 8
    for (int i = 0; i < N; ++i) {
        u128 x = (u128)a[0] * b[i] + t[0];
        uint64_t A = (uint64_t)(x >> 64);
        uint64_t m = (uint64_t)x * inv;
        uint64_t C = (uint64_t)(((u128)m * p[0] + (uint64_t)x) >> 64);
#pragma GCC unroll 8
        for (int j = 1; j < N; ++j) {
            x = (u128)a[j] * b[i] + t[j] + A;
            A = (uint64_t)(x >> 64);
            u128 y = (u128)m * p[j] + (uint64_t)x + C;
            t[j - 1] = (uint64_t)y;
            C = (uint64_t)(y >> 64);
        }
        t[N - 1] = C + A;
    }
    // t < 2p: one conditional subtraction
    uint64_t d[N], borrow = 0;
#pragma GCC unroll 8
    for (int j = 0; j < N; ++j) {
        u128 s = (u128)t[j] - p[j] - borrow;
        d[j] = (uint64_t)s;
        borrow = (uint64_t)(s >> 64) & 1;
    }
#pragma GCC unroll 8
    for (int j = 0; j < N; ++j) out[j] = borrow ? t[j] : d[j];
}

// ---- inversion ------------------------------------------------------------------------------------------
// Values are held as L = N + 1 signed limbs of 62 bits.  out = x^-1 mod p for 0 < x < p (plain residues, p odd prime);
// returns false for x = 0.
template <int N>
struct ModInv {
    static constexpr int L = N + 1;
    static constexpr uint64_t M62 = ~0ull >> 2;
    int64_t p62[L];
    uint64_t p_inv62;  // p^-1 mod 2^62

    explicit ModInv(const uint64_t* p) {
        to62(p62, p);
        uint64_t v = 1;
        for (int i = 0; i < 6; ++i) v *= 2 - p[0] * v;
        p_inv62 = v & M62;
    }
    static void to62(int64_t* o, const uint64_t* a) {
        for (int i = 0; i < L; ++i) {
            int bit = 62 * i, w = bit >> 6, s = bit & 63;
            uint64_t lo = w < N ? a[w] >> s : 0;
            uint64_t hi = (s > 2 && w + 1 < N) ? a[w + 1] << (64 - s) : 0;
            o[i] = (int64_t)((lo | hi) & M62);
        }
    }
    static void from62(uint64_t* o, const int64_t* a) {  // a non-negative, normalised limbs
        for (int w = 0; w < N; ++w) o[w] = 0;
        for (int i = 0; i < L; ++i) {
            int bit = 62 * i, w = bit >> 6, s = bit & 63;
            uint64_t v = (uint64_t)a[i];
            if (w < N) o[w] |= v << s;
            if (s > 2 && w + 1 < N) o[w + 1] |= v >> (64 - s);
        }
    }
    // up to 62 division steps on the low words; eta = -delta.  2^62 (f', g') = [[u v] [q r]] (f, g)
    static int64_t divsteps(int64_t eta, uint64_t f, uint64_t g, int64_t* t) {
        uint64_t u = 1, v = 0, q = 0, r = 1;
        int i = 62;
        for (;;) {
            int zeros = __builtin_ctzll(g | (~0ull << i));
            g >>= zeros;
            u <<= zeros;
            v <<= zeros;
            eta -= zeros;
            i -= zeros;
            if (i == 0) break;
            if (eta < 0) {
                uint64_t tmp;
                eta = -eta;
                tmp = f; f = g; g = (uint64_t)0 - tmp;
                tmp = u; u = q; q = (uint64_t)0 - tmp;
                tmp = v; v = r; r = (uint64_t)0 - tmp;
                // cancel up to 6 low bits of g at once with a multiple of f:  w = -g / f mod 2^k
                int limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
                uint64_t m = (~0ull >> (64 - limit)) & 63u;
                uint64_t w = (f * g * (f * f - 2)) & m;
                g += f * w;
                q += u * w;
                r += v * w;
            } else {
                int limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
                uint64_t m = (~0ull >> (64 - limit)) & 15u;
                uint64_t w = f + (((f + 1) & 4) << 1);
                w = ((uint64_t)0 - w * g) & m;
                g += f * w;
                q += u * w;
                r += v * w;
            }
        }
        t[0] = (int64_t)u;
        t[1] = (int64_t)v;
        t[2] = (int64_t)q;
        t[3] = (int64_t)r;
        return eta;
    }
    static void update_fg(int64_t* f, int64_t* g, const int64_t* t, int len) {
        const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
        i128 cf = (i128)u * f[0] + (i128)v * g[0];
        i128 cg = (i128)q * f[0] + (i128)r * g[0];
        cf >>= 62;
        cg >>= 62;
        for (int i = 1; i < len; ++i) {
            cf += (i128)u * f[i] + (i128)v * g[i];
            cg += (i128)q * f[i] + (i128)r * g[i];
            f[i - 1] = (int64_t)((uint64_t)cf & M62);
            g[i - 1] = (int64_t)((uint64_t)cg & M62);
            cf >>= 62;
            cg >>= 62;
        }
        f[len - 1] = (int64_t)cf;
        g[len - 1] = (int64_t)cg;
    }
    // (d, e) <- t (d, e) / 2^62 mod p, both kept in (-2p, p)
    void update_de(int64_t* d, int64_t* e, const int64_t* t) const {
        const int64_t u = t[0], v = t[1], q = t[2], r = t[3];
        const int64_t sd = d[L - 1] >> 63, se = e[L - 1] >> 63;
        int64_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
        i128 cd = (i128)u * d[0] + (i128)v * e[0];
        i128 ce = (i128)q * d[0] + (i128)r * e[0];
        md -= (int64_t)((p_inv62 * (uint64_t)cd + (uint64_t)md) & M62);
        me -= (int64_t)((p_inv62 * (uint64_t)ce + (uint64_t)me) & M62);
        cd += (i128)p62[0] * md;
        ce += (i128)p62[0] * me;
        cd >>= 62;
        ce >>= 62;
        for (int i = 1; i < L; ++i) {
            cd += (i128)u * d[i] + (i128)v * e[i] + (i128)p62[i] * md;
            ce += (i128)q * d[i] + (i128)r * e[i] + (i128)p62[i] * me;
            d[i - 1] = (int64_t)((uint64_t)cd & M62);
            e[i - 1] = (int64_t)((uint64_t)ce & M62);
            cd >>= 62;
            ce >>= 62;
        }
        d[L - 1] = (int64_t)cd;
        e[L - 1] = (int64_t)ce;
    }
    // r in (-2p, p), optionally negated, -> [0, p) with limbs in [0, 2^62)
    void normalize(int64_t* r, bool negate) const {
        auto carry = [&]() {
            for (int i = 0; i < L - 1; ++i) {
                r[i + 1] += r[i] >> 62;
                r[i] &= (int64_t)M62;
            }
        };
        carry();
        if (r[L - 1] < 0) {
            for (int i = 0; i < L; ++i) r[i] += p62[i];
            carry();
        }
        if (negate) {
            for (int i = 0; i < L; ++i) r[i] = -r[i];
            carry();
        }
        while (r[L - 1] < 0) {
            for (int i = 0; i < L; ++i) r[i] += p62[i];
            carry();
        }
        // r >= 0 now; subtract p while r >= p
        for (;;) {
            int64_t t[L];
            for (int i = 0; i < L; ++i) t[i] = r[i] - p62[i];
            for (int i = 0; i < L - 1; ++i) {
                t[i + 1] += t[i] >> 62;
                t[i] &= (int64_t)M62;
            }
            if (t[L - 1] < 0) break;
            memcpy(r, t, sizeof(t));
        }
    }
    bool invert(uint64_t* out, const uint64_t* x) const {
        int64_t f[L], g[L], d[L], e[L];
        memcpy(f, p62, sizeof(f));
        to62(g, x);
        for (int i = 0; i < L; ++i) d[i] = e[i] = 0;
        e[0] = 1;
        uint64_t nz = 0;
        for (int i = 0; i < L; ++i) nz |= (uint64_t)g[i];
        if (!nz) return false;
        int64_t eta = -1;
        int len = L;
        for (;;) {
            int64_t t[4];
            eta = divsteps(eta, (uint64_t)f[0], (uint64_t)g[0], t);
            update_de(d, e, t);
            update_fg(f, g, t, len);
            if (g[0] == 0) {
                int64_t c = 0;
                for (int i = 1; i < len; ++i) c |= g[i];
                if (c == 0) break;
            }
            // drop a top limb once both values fit below it
            int64_t fn = f[len - 1], gn = g[len - 1];
            int64_t c = ((int64_t)len - 2) >> 63;
            c |= fn ^ (fn >> 63);
            c |= gn ^ (gn >> 63);
            if (c == 0) {
                f[len - 2] |= (int64_t)((uint64_t)fn << 62);
                g[len - 2] |= (int64_t)((uint64_t)gn << 62);
                --len;
            }
        }
        // f = +-1 ; x^-1 = d * f
        normalize(d, f[len - 1] < 0);
        from62(out, d);
        return true;
    }
};

}  // namespace masp_host

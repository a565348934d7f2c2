#!/usr/bin/env python3
"""Emits masp_amd/csrc/device/consts.hpp: 32-bit-limb Montgomery constants for BLS12-381 Fp / Fr.

Montgomery radix is R = 2^(32*N) — N = 12 for Fp (R = 2^384), N = 8 for Fr (R = 2^256) — i.e. the
same in-memory form as blst's 64-bit-limb `blst_fp` / `blst_fr` on a little-endian host
(SURVEY.md A.5), so Rust-side arrays could be handed over without conversion.
"""
import os

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def limbs(x, n):
    return [(x >> (32 * i)) & 0xffffffff for i in range(n)]


def arr(name, x, n):
    return "static constexpr uint32_t %s[%d] = {%s};" % (name, n, ", ".join("0x%08xu" % l for l in limbs(x, n)))


def block(tag, mod, n, extra):
    Rm = (1 << (32 * n)) % mod
    out = ["struct %s {" % tag, "    static constexpr int N = %d;" % n,
           "    " + arr("MOD", mod, n),
           "    " + arr("R1", Rm, n) + "   // R mod p  (Montgomery one)",
           "    " + arr("R2", Rm * Rm % mod, n) + "   // R^2 mod p",
           "    " + arr("R3", Rm * Rm * Rm % mod, n) + "   // R^3 mod p",
           "    static constexpr uint32_t INV = 0x%08xu;   // -p^-1 mod 2^32" % ((-pow(mod, -1, 1 << 32)) % (1 << 32)),
           "    " + arr("PM2", mod - 2, n) + "   // p - 2 (Fermat inverse exponent)",
           "    " + arr("HALF", (mod - 1) // 2, n) + "   // (p-1)/2 (lexicographic sign)"]
    out += extra
    out.append("};")
    return "\n".join(out)


G1X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
G1Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
G2X0 = 0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8
G2X1 = 0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e
G2Y0 = 0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801
G2Y1 = 0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be
RP = (1 << 384) % P
fp_extra = ["    // standard generators (affine, Montgomery form); encodings KAT-checked in tests"] + [
    "    " + arr(n, v * RP % P, 12) for n, v in (("G1_X", G1X), ("G1_Y", G1Y), ("G2_X0", G2X0), ("G2_X1", G2X1), ("G2_Y0", G2Y0), ("G2_Y1", G2Y1))]
# ---- constants of the subgroup membership tests (M. Scott, "A note on group membership tests for G1, G2 and GT on BLS
# pairing-friendly curves"): computed here, checked on the generators below before anything is written.
U = -0xd201000000010000                  # the curve parameter ("x" of BLS12-381)


def f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2mul(r, a)
        a = f2mul(a, a)
        e >>= 1
    return r


def f2inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, (-a[1]) * n % P)


def ec_add(mul, inv, p, q):
    if p is None:
        return q
    if q is None:
        return p
    sub = (lambda a, b: (a - b) % P) if isinstance(p[0], int) else (lambda a, b: ((a[0] - b[0]) % P, (a[1] - b[1]) % P))
    add = (lambda a, b: (a + b) % P) if isinstance(p[0], int) else (lambda a, b: ((a[0] + b[0]) % P, (a[1] + b[1]) % P))
    if p[0] == q[0]:
        if p[1] != q[1]:
            return None
        x2 = mul(p[0], p[0])
        lam = mul(add(add(x2, x2), x2), inv(add(p[1], p[1])))
    else:
        lam = mul(sub(q[1], p[1]), inv(sub(q[0], p[0])))
    x3 = sub(sub(mul(lam, lam), p[0]), q[0])
    return (x3, sub(mul(lam, sub(p[0], x3)), p[1]))


def ec_mul(mul, inv, p, k):
    r = None
    while k:
        if k & 1:
            r = ec_add(mul, inv, r, p)
        p = ec_add(mul, inv, p, p)
        k >>= 1
    return r


m1, i1 = (lambda a, b: a * b % P), (lambda a: pow(a, -1, P))
# G1: phi(x, y) = (beta x, y) acts on G1 as multiplication by -u^2 for ONE of the two primitive cube roots of unity
g = 2
while pow(g, (P - 1) // 3, P) == 1:
    g += 1
BETA = None
for cand in (pow(g, (P - 1) // 3, P), pow(g, 2 * (P - 1) // 3, P)):
    if (cand * G1X % P, G1Y) == ec_mul(m1, i1, (G1X, G1Y), (-(U * U)) % R):
        BETA = cand
assert BETA is not None
# G2: psi(x, y) = (cx conj(x), cy conj(y)) (untwist, Frobenius, twist) acts on G2 as multiplication by p = u (mod r)
xi3, xi2 = f2pow((1, 1), (P - 1) // 3), f2pow((1, 1), (P - 1) // 2)
PSI = None
G2P = ((G2X0, G2X1), (G2Y0, G2Y1))
want = ec_mul(f2mul, f2inv, G2P, U % R)
for cx in (xi3, f2inv(xi3)):
    for cy in (xi2, f2inv(xi2)):
        conj = lambda a: (a[0], (-a[1]) % P)   # noqa: E731
        if (f2mul(cx, conj(G2P[0])), f2mul(cy, conj(G2P[1]))) == want:
            PSI = (cx, cy)
assert PSI is not None
fp_extra += ["    // subgroup membership tests (Scott): on G1 (beta x, y) = -[u^2] P; on G2 (cx conj x, cy conj y) = [u] Q; Montgomery form",
             "    " + arr("ENDO_BETA", BETA * RP % P, 12),
             "    " + arr("PSI_CX0", PSI[0][0] * RP % P, 12), "    " + arr("PSI_CX1", PSI[0][1] * RP % P, 12),
             "    " + arr("PSI_CY0", PSI[1][0] * RP % P, 12), "    " + arr("PSI_CY1", PSI[1][1] * RP % P, 12),
             "    " + arr("U_ABS", -U, 2) + "   // |u|, u = -0xd201000000010000",
             "    " + arr("U_SQR", U * U, 4) + "   // u^2"]
rou = pow(7, (R - 1) >> 32, R)
fr_extra = ["    " + arr("ROOT_OF_UNITY", rou * (1 << 256) % R, 8) + "   // 7^((r-1)/2^32), Montgomery form",
            "    " + arr("GEN", 7 * (1 << 256) % R, 8) + "   // multiplicative generator 7, Montgomery form",
            "    " + arr("GEN_INV", pow(7, -1, R) * (1 << 256) % R, 8)]
text = """// GENERATED by tools/gen_device_consts.py — do not edit.
#pragma once
#include <cstdint>
namespace masp {
%s

%s
}  // namespace masp
""" % (block("FpCfg", P, 12, fp_extra), block("FrCfg", R, 8, fr_extra))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "masp_amd", "csrc", "device", "consts.hpp")
open(path, "w").write(text)
print("wrote", os.path.normpath(path))

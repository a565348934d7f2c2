"""Pins the oracle's arithmetic (oracle/field.hpp, curve.hpp) against an independent pure-Python
big-integer implementation and the published BLS12-381 KATs listed in SURVEY.md §8(c)."""
import random

import numpy as np

import oracle_lib as O
import pyref
from pyref import P, R, F1, F2


def test_fr_fp_ops_match_python_bigint():
    rng = random.Random(1)
    for mod, op in ((R, O.fr_op), (P, O.fp_op)):
        edge = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, 1 << 64, (1 << 255) % mod]
        vals = edge + [rng.randrange(mod) for _ in range(40)]
        for a in vals:
            b = rng.choice(vals)
            assert op(0, a, b) == (a + b) % mod
            assert op(1, a, b) == (a - b) % mod
            assert op(2, a, b) == a * b % mod
            assert op(4, a) == (-a) % mod
            if a:
                assert op(3, a) == pow(a, -1, mod)
        assert op(3, 0) == 0


def test_fr_root_of_unity_kat():
    # SURVEY.md §8(c) / A.4
    import ctypes as C
    out = C.create_string_buffer(32)
    O.lib().oracle_fr_root_of_unity(out)
    rou = int.from_bytes(out.raw, "little")
    assert rou == 0x16a2a19edfe81f20d09b681922c813b4b63683508c2280b93829971f439f0d2b
    assert rou == pow(7, (R - 1) >> 32, R)
    assert pow(rou, 1 << 32, R) == 1 and pow(rou, 1 << 31, R) != 1
    kat = {15: 0x3291357ee558b50d483405417a0cbe39c8d5f51db3f32699fbd047e11279bb6e,
           16: 0x2155379d12180caa88f39a78f1aeb57867a665ae1fcadc91d7118f85cd96b8ad,
           17: 0x224262332d8acbf4473a2eef772c33d6cd7f2bd6d0711b7d08692405f3b70f10}
    for k, v in kat.items():
        O.lib().oracle_fr_omega(k, out)
        assert int.from_bytes(out.raw, "little") == v


def test_generator_encodings_kat():
    # compressed generators
    u, c = O.g1_mul_gen(1)
    assert c.hex() == "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    assert u == pyref.g1_unc(pyref.G1)
    u2, c2 = O.g2_mul_gen(1)
    assert c2.hex() == ("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
                        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
    assert u2 == pyref.g2_unc(pyref.G2)
    # identity
    u0, c0 = O.g1_mul_gen(0)
    assert c0 == b"\xc0" + bytes(47) and u0 == b"\x40" + bytes(95)
    # group order
    assert O.g1_mul_gen(R - 1)[0] == pyref.g1_unc((pyref.G1[0], P - pyref.G1[1]))


def test_scalar_mul_and_encodings_match_python():
    rng = random.Random(2)
    import ctypes as C
    for k in [2, 3, 5, 0xdeadbeef, R - 2] + [rng.randrange(R) for _ in range(4)]:
        p1 = pyref.ec_mul(F1, pyref.G1, k)
        u, c = O.g1_mul_gen(k)
        assert u == pyref.g1_unc(p1) and c == pyref.g1_comp(p1)
        out = C.create_string_buffer(96)
        assert O.lib().oracle_g1_decompress(c, out) == 0 and out.raw == u
        # generic double-and-add path
        assert O.lib().oracle_g1_mul(pyref.g1_unc(pyref.G1), k.to_bytes(32, "little"), out) == 0 and out.raw == u
    for k in [2, 7, R - 3] + [rng.randrange(R) for _ in range(2)]:
        p2 = pyref.ec_mul(F2, pyref.G2, k)
        u, c = O.g2_mul_gen(k)
        assert u == pyref.g2_unc(p2) and c == pyref.g2_comp(p2)
        out = C.create_string_buffer(192)
        assert O.lib().oracle_g2_decompress(c, out) == 0 and out.raw == u
        assert O.lib().oracle_g2_mul(pyref.g2_unc(pyref.G2), k.to_bytes(32, "little"), out) == 0 and out.raw == u


def test_add_exceptional_cases():
    import ctypes as C
    g = pyref.g1_unc(pyref.G1)
    neg = pyref.g1_unc((pyref.G1[0], P - pyref.G1[1]))
    inf = b"\x40" + bytes(95)
    out = C.create_string_buffer(96)
    O.lib().oracle_g1_add(g, g, out)
    assert out.raw == pyref.g1_unc(pyref.ec_mul(F1, pyref.G1, 2))
    O.lib().oracle_g1_add(g, neg, out)
    assert out.raw == inf
    O.lib().oracle_g1_add(inf, g, out)
    assert out.raw == g
    O.lib().oracle_g1_add(g, inf, out)
    assert out.raw == g


def test_msm_matches_python():
    rng = random.Random(3)
    n = 40
    ks = [rng.randrange(R) for _ in range(n)]
    # boolean-heavy scalars like a MASP witness, plus edge values
    sc = [rng.choice([0, 1, 1, 0, rng.randrange(R), R - 1, 2]) for _ in range(n)]
    kb = np.stack([np.frombuffer(k.to_bytes(32, "little"), np.uint8) for k in ks])
    sb = np.stack([np.frombuffer(s.to_bytes(32, "little"), np.uint8) for s in sc])
    expect = sum(k * s for k, s in zip(ks, sc)) % R
    assert O.msm_g1(O.g1_mul_gen_many(kb), sb) == O.g1_mul_gen(expect)[0]
    assert O.msm_g2(O.g2_mul_gen_many(kb), sb) == O.g2_mul_gen(expect)[0]
    # against python point arithmetic directly (no discrete-log shortcut)
    acc = None
    for k, s in zip(ks[:6], sc[:6]):
        acc = pyref.ec_add(F1, acc, pyref.ec_mul(F1, pyref.ec_mul(F1, pyref.G1, k), s))
    assert O.msm_g1(O.g1_mul_gen_many(kb[:6]), sb[:6]) == pyref.g1_unc(acc)
    # empty
    assert O.msm_g1(np.zeros((0, 96), np.uint8), np.zeros((0, 32), np.uint8)) == b"\x40" + bytes(95)


def test_pairing_bilinear_nondegenerate():
    assert O.lib().oracle_pairing_selftest((5).to_bytes(32, "little"), (7).to_bytes(32, "little")) == 1
    rng = random.Random(4)
    a, b = rng.randrange(R), rng.randrange(R)
    assert O.lib().oracle_pairing_selftest(a.to_bytes(32, "little"), b.to_bytes(32, "little")) == 1

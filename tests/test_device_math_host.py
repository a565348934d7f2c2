"""The HIP kernels' field/curve/encoding source (masp_amd/csrc/device/*.hpp, 32-bit limbs) compiled for
the host and checked against python big integers and the oracle.  Runs without a GPU; the same
functions run on the device in the -m gpu tests."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import pyref
from pyref import P, R, F1, F2

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "device_math_host.hip")
SO = os.path.join(HERE, "native", "_device_math_host.so")


@pytest.fixture(scope="module")
def mh():
    hdrs = [os.path.join(HERE, "..", "masp_amd", "csrc", "device", f) for f in ("field.hpp", "curve.hpp", "io.hpp", "consts.hpp")] + [os.path.join(HERE, "..", "tools", "fp28.hpp")]
    newest = max(os.path.getmtime(p) for p in hdrs + [SRC])
    if not os.path.exists(SO) or os.path.getmtime(SO) < newest:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", SO])
    return C.CDLL(SO)


def test_field_ops(mh):
    rng = random.Random(10)
    for which, mod, nb in ((0, P, 48), (1, R, 32)):
        edge = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (mod + 1) // 2, 0xffffffff, 1 << 32, (1 << (8 * nb - 8)) % mod]
        vals = edge + [rng.randrange(mod) for _ in range(60)]
        out = C.create_string_buffer(nb)
        def op(o, a, b=0):
            mh.mh_field_op(which, o, a.to_bytes(nb, "little"), b.to_bytes(nb, "little"), out)
            return int.from_bytes(out.raw, "little")
        for a in vals:
            b = rng.choice(vals)
            assert op(0, a, b) == (a + b) % mod
            assert op(1, a, b) == (a - b) % mod
            assert op(2, a, b) == a * b % mod
            assert op(4, a) == (-a) % mod
            assert op(5, a) == a * a % mod
        for a in vals[:12]:
            assert op(3, a) == (pow(a, -1, mod) if a else 0)
        # the binary-gcd inverse (30 rounds at a time on 64-bit approximations) of the bucket trees' shared inversions: random values
        # and the shapes that stress the approximations (powers of two and their neighbours, long runs of zero / one bits)
        bits = mod.bit_length()
        special = [1 << k for k in range(1, bits - 1)] + [(1 << k) - 1 for k in range(2, bits - 1)] + [mod - (1 << k) for k in range(0, bits - 2)] + \
                  [((1 << 62) + 1) << k for k in range(0, bits - 64, 7)] + [(mod >> k) | 1 for k in range(1, 200, 3)]
        for a in vals + [x % mod for x in special] + [rng.randrange(mod) for _ in range(1500)] + [rng.randrange(1 << rng.randrange(1, bits)) for _ in range(500)]:
            assert op(6, a) == (pow(a, -1, mod) if a else 0), hex(a)
        for a in vals[:4]:
            assert op(7, a) == (pow(a, -1, mod) if a else 0)


@pytest.mark.gpu
def test_field_ops_on_the_device(mh):
    """The DEVICE overloads of add / sub / neg / dbl / the conditional subtraction behind every product are hand-written carry
    chains (field.hpp) that the host build never compiles: the same operations on the GPU, against python integers — the
    edge values against each other (sums and differences that land exactly on 0, p - 1, p, 2p - 2) and random pairs."""
    rng = random.Random(11)
    for which, mod, nb in ((0, P, 48), (1, R, 32)):
        edge = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (mod + 1) // 2, 0xffffffff, 1 << 32, (1 << 32) - 1, 1 << 64, (1 << 96) - 1,
                (1 << (8 * nb - 8)) % mod, mod - (1 << 32), mod - (1 << 64) + 1, (1 << (mod.bit_length() - 1)), (1 << (mod.bit_length() - 1)) - 1]
        pairs = [(a, b) for a in edge for b in edge] + [(a, mod - a) for a in edge if a] + \
                [(rng.randrange(mod), rng.randrange(mod)) for _ in range(4000)]
        n = len(pairs)
        A = b"".join(a.to_bytes(nb, "little") for a, _ in pairs)
        B = b"".join(b.to_bytes(nb, "little") for _, b in pairs)
        out = C.create_string_buffer(nb * n)
        want = {0: lambda a, b: (a + b) % mod, 1: lambda a, b: (a - b) % mod, 2: lambda a, b: a * b % mod, 4: lambda a, b: (-a) % mod,
                5: lambda a, b: a * a % mod, 8: lambda a, b: 2 * a % mod}
        for op, f in want.items():
            assert mh.mh_field_ops_gpu(which, op, A, B, out, n) == 0
            got = [int.from_bytes(out.raw[nb * i:nb * (i + 1)], "little") for i in range(n)]
            bad = [i for i in range(n) if got[i] != f(*pairs[i])]
            assert not bad, (which, op, hex(pairs[bad[0]][0]), hex(pairs[bad[0]][1]), hex(got[bad[0]]))


@pytest.mark.gpu
def test_raw_products_below_2p_on_the_device(mh):
    """The Montgomery products on RAW operands anywhere in [0, 2p): the bucket tree keeps its chains of products in that range
    (fe_mul_lazy: no conditional subtraction between the links), and the products skip the carry word after the terms that hold
    a top limb (below 2^30 for any operand below 2p).  Worst shapes: 2p - 1, every limb below the top one all ones, p, p + 1,
    2^381 + ..., against python integers."""
    rng = random.Random(12)
    nb, RM = 48, 1 << 384
    rinv = pow(RM, -1, P)
    top = (2 * P) >> 352
    edge = [0, 1, P - 1, P, P + 1, 2 * P - 1, 2 * P - 2, (top << 352) | ((2 * P) & ((1 << 352) - 1)) - 1, ((top - 1) << 352) | ((1 << 352) - 1),
            (1 << 381) - 1, 1 << 381, (1 << 381) | ((1 << 352) - 1), (1 << 352) - 1, ((1 << 32) - 1) << 320, (top << 352), (top << 352) | 0xffffffff,
            sum(0xffffffff << (64 * k) for k in range(6)) % (2 * P), sum(0xffffffff << (64 * k + 32) for k in range(5)) | ((top - 1) << 352)]
    assert all(0 <= e < 2 * P for e in edge)
    pairs = [(a, b) for a in edge for b in edge] + [(rng.randrange(2 * P), rng.randrange(2 * P)) for _ in range(6000)] + \
            [(rng.choice(edge), rng.randrange(2 * P)) for _ in range(500)] + [(rng.randrange(2 * P), rng.choice(edge)) for _ in range(500)]
    n = len(pairs)
    A = b"".join(a.to_bytes(nb, "little") for a, _ in pairs)
    B = b"".join(b.to_bytes(nb, "little") for _, b in pairs)
    out = C.create_string_buffer(nb * n)
    want = {16: (lambda a, b: a * b * rinv % P, True), 17: (lambda a, b: a * b * rinv % P, False), 18: (lambda a, b: a * a * rinv % P, True),
            19: (lambda a, b: 2 * a * b * rinv % P, True), 20: (lambda a, b: a * b * rinv * b * rinv % P, False)}
    for op, (f, canonical) in want.items():
        ps = pairs
        if op == 19:   # a b + z w must stay below p R: 2 (2p)^2 < 9.8 p^2 — every pair qualifies
            pass
        assert mh.mh_field_ops_gpu(0, op, A, B, out, n) == 0
        got = [int.from_bytes(out.raw[nb * i:nb * (i + 1)], "little") for i in range(n)]
        for i in range(n):
            g, w = got[i], f(*ps[i])
            assert g % P == w and (g < P if canonical else g < 2 * P), (op, hex(ps[i][0]), hex(ps[i][1]), hex(g))


def _le(x):
    return np.frombuffer(x.to_bytes(32, "little"), np.uint8)


def test_g1_group_law_and_encodings(mh):
    rng = random.Random(11)
    n = 12
    ks = [rng.randrange(R) for _ in range(n)]
    sc = [rng.choice([0, 1, 2, R - 1, rng.randrange(R)]) for _ in range(n)]
    pts = O.g1_mul_gen_many(np.stack([_le(k) for k in ks]))
    # include an infinity base
    pts[3] = 0
    pts[3, 0] = 0x40
    ks[3] = 0
    sb = np.stack([_le(s) for s in sc])
    o96, o48 = C.create_string_buffer(96), C.create_string_buffer(48)
    assert mh.mh_g1_lincomb(pts.ctypes.data_as(C.c_void_p), sb.ctypes.data_as(C.c_void_p), n, 0, o96, o48) == 0
    expect = sum(k * s for k, s in zip(ks, sc)) % R
    u, c = O.g1_mul_gen(expect)
    assert o96.raw == u and o48.raw == c
    # madd chain incl. doubling (same point twice), cancellation (P then -P) and infinity
    cnt = np.zeros((4, 32), np.uint8)
    p4 = np.stack([pts[0], pts[0], pts[1], pts[1]])
    cnt[0, 0] = 2            # P0 + P0  -> exercises the doubling branch
    cnt[1, 0] = 3
    cnt[2, 0] = 1
    cnt[3, 0], cnt[3, 1] = 1, 1   # + P1 - P1
    assert mh.mh_g1_lincomb(p4.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), 4, 1, o96, o48) == 0
    assert o96.raw == O.g1_mul_gen(5 * ks[0] % R)[0]
    cnt[:] = 0
    cnt[0, 0] = 1
    cnt[1, 0], cnt[1, 1] = 1, 1   # P0 - P0 = infinity
    assert mh.mh_g1_lincomb(p4.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), 2, 1, o96, o48) == 0
    assert o96.raw == b"\x40" + bytes(95) and o48.raw == b"\xc0" + bytes(47)


def test_g2_group_law_and_encodings(mh):
    rng = random.Random(12)
    n = 6
    ks = [rng.randrange(R) for _ in range(n)]
    sc = [rng.choice([1, 2, R - 1, rng.randrange(R)]) for _ in range(n)]
    pts = O.g2_mul_gen_many(np.stack([_le(k) for k in ks]))
    sb = np.stack([_le(s) for s in sc])
    o192, o96 = C.create_string_buffer(192), C.create_string_buffer(96)
    assert mh.mh_g2_lincomb(pts.ctypes.data_as(C.c_void_p), sb.ctypes.data_as(C.c_void_p), n, 0, o192, o96) == 0
    u, c = O.g2_mul_gen(sum(k * s for k, s in zip(ks, sc)) % R)
    assert o192.raw == u and o96.raw == c
    cnt = np.zeros((3, 32), np.uint8)
    p3 = np.stack([pts[0], pts[0], pts[0]])
    cnt[0, 0] = 2
    cnt[1, 0] = 2
    cnt[2, 0], cnt[2, 1] = 1, 1
    assert mh.mh_g2_lincomb(p3.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p), 3, 1, o192, o96) == 0
    assert o192.raw == O.g2_mul_gen(3 * ks[0] % R)[0]


# ---- the 28-bit-limb form of Fp (tools/fp28.hpp: an experiment, measured and rejected — DESIGN.md section 6) ----------------------------------------------------------------------
M28, R28 = (1 << 28) - 1, 1 << 392


def _l28(limbs):
    return b"".join(int(v).to_bytes(4, "little") for v in limbs)


def _n28(x):  # normalised limbs of an integer < 2^392
    return [(x >> (28 * i)) & M28 for i in range(14)]


def _v28(raw):
    return sum(int.from_bytes(raw[4 * i:4 * i + 4], "little") << (28 * i) for i in range(14))


def _fp28_cases():
    rng = random.Random(28)
    K2 = [0x1fff5556, 0x1fdffffe, 0x17ffff72, 0x1fffd629, 0x1c483d56, 0x141ed61d, 0x1ece61a4, 0x1e70a256, 0x18ee9708, 0x19759aeb, 0x174f6c85,
          0x1cd34962, 0x13d472fe, 0x34021]
    assert sum(k << (28 * i) for i, k in enumerate(K2)) == 2 * P
    canon_edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, M28, 1 << 28, (1 << 364) - 1, 1 << 364, (0x1a011 << 364), (0x1a011 << 364) - 1, P - (1 << 28),
                  P - (1 << 364), (1 << 380) | 5, sum(M28 << (28 * i) for i in range(13))]
    canon_edge = [c % P for c in canon_edge]
    canon = canon_edge + [rng.randrange(P) for _ in range(3000)]
    return rng, canon_edge, canon


def _run28(fn, op, A, B):
    n = len(A)
    out = C.create_string_buffer(56 * n)
    assert fn(op, b"".join(A), b"".join(B), out, n) == 0
    return [out.raw[56 * i:56 * (i + 1)] for i in range(n)]


def _check_fp28(fn):
    rng, canon_edge, canon = _fp28_cases()
    rinv = pow(R28, -1, P)
    z = _l28([0] * 14)
    # canon: lazy limbs < 2^31 with value < 8p — built from k p + c spread over lifted limbs, and plain random lazy limbs
    lazy = []
    for _ in range(3000):
        v = rng.randrange(8 * P)
        limbs = _n28(v)
        for i in range(13):  # move whole 2^28s down from the limb above where there are some: same value, fatter limbs
            t = min(limbs[i + 1], rng.randrange(8))
            limbs[i + 1] -= t
            limbs[i] += t << 28
        lazy.append(limbs)
    for k in range(8):
        for c in canon_edge:
            if k * P + c < 8 * P:
                lazy.append(_n28(k * P + c))
    lazy += [_n28(k * P) for k in range(8)] + [_n28(k * P - 1) for k in range(1, 9)]
    got = _run28(fn, 2, [_l28(l) for l in lazy], [z] * len(lazy))
    for l, g in zip(lazy, got):
        v = sum(x << (28 * i) for i, x in enumerate(l))
        assert _v28(g) == v % P and all(int.from_bytes(g[4 * i:4 * i + 4], "little") <= M28 for i in range(14)), (hex(v), g.hex())
    # products of lazy operands (limbs up to 2^30 on both sides), result N < 2p
    ops = [(rng.choice(lazy), rng.choice(lazy)) for _ in range(3000)] + [(_n28(a), _n28(b)) for a in canon_edge for b in canon_edge]
    fat = [[min(x * 3, (1 << 30) - 1) for x in _n28(rng.randrange(2 * P))] for _ in range(200)]   # limbs up to 2^30, any value
    ops += [(rng.choice(fat), rng.choice(fat)) for _ in range(500)] + [(_n28(rng.randrange(2 * P)), rng.choice(lazy)) for _ in range(2000)]
    val = lambda l: sum(x << (28 * i) for i, x in enumerate(l))
    got = _run28(fn, 0, [_l28(a) for a, _ in ops], [_l28(b) for _, b in ops])
    for (a, b), g in zip(ops, got):
        if val(a) * val(b) >= P * R28 or 14 * max(a) * max(b) + 14 * (1 << 56) + (1 << 36) >= 1 << 64:   # the product's contract
            continue
        assert _v28(g) < 2 * P and _v28(g) % P == val(a) * val(b) * rinv % P, (a, b)
        assert all(int.from_bytes(g[4 * i:4 * i + 4], "little") <= M28 for i in range(13))
    sq = [l for l in lazy if max(l) < (1 << 29)] + [_n28(c) for c in canon_edge]
    got = _run28(fn, 1, [_l28(a) for a in sq], [z] * len(sq))
    for a, g in zip(sq, got):
        assert _v28(g) < 2 * P and _v28(g) % P == val(a) ** 2 * rinv % P, a
    # sub_lazy / neg / add / sub / dbl / the x3 formula on canonical operands
    pairs = [(a, b) for a in canon_edge for b in canon_edge] + [(rng.choice(canon), rng.choice(canon)) for _ in range(3000)] + [(a, a) for a in canon_edge]
    A, B = [_l28(_n28(a)) for a, _ in pairs], [_l28(_n28(b)) for _, b in pairs]
    for op, f in ((3, lambda a, b: (a - b) % P), (8, lambda a, b: (a + b) % P), (9, lambda a, b: (a - b) % P), (10, lambda a, b: 2 * a % P),
                  (4, lambda a, b: (-a) % P), (11, lambda a, b: (a * a * rinv - 2 * b) % P)):
        got = _run28(fn, op, A, B)
        for (a, b), g in zip(pairs, got):
            if op == 3:
                assert _v28(g) % P == f(a, b) and _v28(g) < 3 * P
            else:
                assert _v28(g) == f(a, b), (op, hex(a), hex(b), hex(_v28(g)))
    got = _run28(fn, 12, A, B)
    for (a, b), g in zip(pairs, got):
        assert g[0] == (1 if a == 0 else 0) | (2 if a == b else 0)
    # to and from the 12 x 32-bit residues: s = x 2^384 mod p  <->  x 2^392 mod p
    fps = canon_edge + [rng.randrange(P) for _ in range(2000)]
    F = [s.to_bytes(48, "little") + b"\0" * 8 for s in fps]
    got = _run28(fn, 5, F, [z] * len(F))
    for s, g in zip(fps, got):
        assert _v28(g) == s * 256 % P
    fps2 = fps + [rng.randrange(2 * P) for _ in range(500)]
    F2_ = [s.to_bytes(48, "little") + b"\0" * 8 for s in fps2]
    got = _run28(fn, 6, F2_, [z] * len(F2_))
    for s, g in zip(fps2, got):
        assert _v28(g) == s * 256 and all(int.from_bytes(g[4 * i:4 * i + 4], "little") <= M28 for i in range(14))
    back = [rng.randrange(2 * P) for _ in range(2000)] + canon_edge + [P, P + 1, 2 * P - 1]
    got = _run28(fn, 7, [_l28(_n28(v)) for v in back], [z] * len(back))
    inv256 = pow(256, -1, P)
    for v, g in zip(back, got):
        assert int.from_bytes(g[:48], "little") == v * inv256 % P


def test_fp28_ops_host(mh):
    """The 28-bit-limb field written for the G1 bucket tree (tools/fp28.hpp; not in the product), portable build: products of lazy operands, the carry-free
    differences, canonicalisation from every k p + c, the conversions to and from the 12 x 32-bit Montgomery residues."""
    _check_fp28(mh.mh_fp28_ops)


@pytest.mark.gpu
def test_fp28_ops_on_the_device(mh):
    """The same on the GPU: there the products are v_mad_u64_u32 chains without a carry word, the column shift a v_alignbit."""
    _check_fp28(mh.mh_fp28_ops_gpu)

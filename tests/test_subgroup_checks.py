"""`groth16::Proof::read` — what the reference parses proofs with before verifying them
(/root/reference/masp_proofs/src/sapling/verifier/batch.rs:85,125,154, single.rs) — refuses points that are not in the
prime-order subgroups and encodings of the point at infinity with stray bits.  The pairing cannot see the cofactor part of a
point: a proof whose A (or C, or B) is shifted by a small-order point satisfies the Groth16 equation just like the original, so
a verifier without the membership test accepts malleated proofs the reference rejects.  Both product verifiers (host:
masp_amd/csrc/host/pairing.h; GPU: masp_hip_verify_batch) must refuse them."""
import random

import pytest

import oracle_lib as O
import toy_r1cs
from pyref import F1, F2, G1, G2, P, R, ec_add, ec_mul, g1_comp, g2_comp


def _sqrt_fp(a):
    r = pow(a, (P + 1) // 4, P)
    return r if r * r % P == a else None


def _sqrt_fp2(a):
    if a == (0, 0):
        return a
    a0, a1 = a
    n = _sqrt_fp((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    for sg in (n, (-n) % P):
        d = (a0 + sg) * pow(2, -1, P) % P
        x0 = _sqrt_fp(d)
        if x0:
            r = (x0, a1 * pow(2 * x0, -1, P) % P)
            if F2.mul(r, r) == a:
                return r
    return None


def _g1_decompress(b):
    x = int.from_bytes(bytes([b[0] & 0x1f]) + b[1:48], "big")
    y = _sqrt_fp((x ** 3 + 4) % P)
    if (y > (P - 1) // 2) != bool(b[0] & 0x20):
        y = P - y
    return (x, y)


def _g2_decompress(b):
    x = (int.from_bytes(b[48:96], "big"), int.from_bytes(bytes([b[0] & 0x1f]) + b[1:48], "big"))
    y = _sqrt_fp2(F2.add(F2.mul(F2.mul(x, x), x), (4, 4)))
    big = (y[1] > (P - 1) // 2) if y[1] else (y[0] > (P - 1) // 2)
    if big != bool(b[0] & 0x20):
        y = F2.neg(y)
    return (x, y)


def small_order_points():
    """T1 = (0, 2): order 3 on y^2 = x^3 + 4.  T2: a point of the twist outside G2, times r: its order divides the cofactor."""
    t1 = (0, 2)
    assert ec_mul(F1, t1, 3) is None
    x = (2, 0)
    while True:
        y = _sqrt_fp2(F2.add(F2.mul(F2.mul(x, x), x), (4, 4)))
        if y is not None and ec_mul(F2, (x, y), R) is not None:
            break
        x = (x[0] + 1, 0)
    t2 = ec_mul(F2, (x, y), R)
    return t1, t2


def malleated(proof):
    """-> {name: proof bytes} each satisfying the pairing equation like `proof` but with one point outside its subgroup"""
    t1, t2 = small_order_points()
    a, b, c = _g1_decompress(proof[:48]), _g2_decompress(proof[48:144]), _g1_decompress(proof[144:])
    assert g1_comp(a) == proof[:48] and g2_comp(b) == proof[48:144] and g1_comp(c) == proof[144:]
    return {"A + T": g1_comp(ec_add(F1, a, t1)) + proof[48:], "C + T": proof[:144] + g1_comp(ec_add(F1, c, t1)),
            "B + T2": proof[:48] + g2_comp(ec_add(F2, b, t2)) + proof[144:]}


def _toy():
    cs, inputs, aux, vals = toy_r1cs.make(5, 8, 40, 300)
    pbuf = O.generate_parameters(cs, toy_r1cs.toxic(5))
    proof = O.create_proof(O.Params(pbuf), cs, inputs, aux, 5, 6)
    return pbuf, proof, vals[1:8]


def test_host_verifier_refuses_points_outside_the_subgroups():
    from masp_amd import host as H
    pbuf, proof, pub = _toy()
    vk = H.PreparedVerifyingKey(pbuf)
    assert vk.verify(proof, pub) and vk.verify_batch([proof] * 3, [pub] * 3)
    for name, bad in malleated(proof).items():
        assert bad != proof and len(bad) == 192
        if name != "B + T2":
            # the Groth16 equation itself holds for the shifted proof (the oracle's pairing check has no membership test): only
            # the parse-time test tells it from the original
            assert O.verify_proof(pbuf, bad, pub) == 1, name
        assert not vk.verify(bad, pub), name
        assert not vk.verify_batch([proof, bad, proof], [pub] * 3), name
    # members of the subgroups keep passing: every multiple of the generators (membership test in isolation would also do)
    rng = random.Random(6)
    for _ in range(3):
        k = rng.randrange(1, R)
        other = g1_comp(ec_mul(F1, G1, k)) + g2_comp(ec_mul(F2, G2, k)) + g1_comp(ec_mul(F1, G1, k + 1))
        assert vk.verify(other, pub) is False          # decodes (all three points are in their subgroups), fails the equation
    # infinity with stray bits: sort flag set, or a non-zero byte further down
    for enc in (bytes([0xe0]) + bytes(47), bytes([0xc0]) + bytes(46) + b"\x01"):
        assert not vk.verify(enc + proof[48:], pub)
        assert not vk.verify(proof[:144] + enc, pub)
    assert not vk.verify(proof[:48] + bytes([0xe0]) + bytes(95) + proof[144:], pub)
    # a CLEAN encoding of the point at infinity (0xc0 || zeros) in any of the three positions: bellman's Proof::read refuses the
    # identity after decompression ("point at infinity"), so such bytes are never a proof — single and batched
    for bad in _with_infinity(proof):
        assert not vk.verify(bad, pub)
        assert not vk.verify_batch([proof, bad], [pub] * 2)


def _with_infinity(proof):
    inf1, inf2 = bytes([0xc0]) + bytes(47), bytes([0xc0]) + bytes(95)
    return [inf1 + proof[48:], proof[:48] + inf2 + proof[144:], proof[:144] + inf1, inf1 + inf2 + inf1]


@pytest.mark.gpu
def test_gpu_batch_verifier_refuses_points_outside_the_subgroups():
    import masp_amd
    pbuf, proof, pub = _toy()
    ctx = masp_amd.Context(0)
    try:
        gvk = ctx.prepare_verifying_key(pbuf)
        assert gvk.verify_batch([proof] * 5, [pub] * 5)
        for name, bad in malleated(proof).items():
            assert not gvk.verify_batch([bad], [pub]), name
            assert not gvk.verify_batch([proof, proof, bad, proof], [pub] * 4), name
        for enc in (bytes([0xe0]) + bytes(47), bytes([0xc0]) + bytes(46) + b"\x01"):
            assert not gvk.verify_batch([enc + proof[48:]], [pub])
            assert not gvk.verify_batch([proof[:144] + enc], [pub])
        assert not gvk.verify_batch([proof[:48] + bytes([0xe0]) + bytes(95) + proof[144:]], [pub])
        for bad in _with_infinity(proof):                 # clean infinity encodings are refused like by Proof::read
            assert not gvk.verify_batch([bad], [pub])
            assert not gvk.verify_batch([proof, bad, proof], [pub] * 3)
        assert gvk.verify_batch([proof] * 2, [pub] * 2)
        gvk.close()
    finally:
        ctx.close()

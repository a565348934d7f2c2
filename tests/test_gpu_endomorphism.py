"""s*A and r*B1 through the curve endomorphism (device/groth16.hpp: k_groth16_var_mul, endo): [k] P = [rem] P + [q] (beta x, -y)
with k = q u^2 + rem holds on the prime-order subgroup only.  A CRS whose points behind A and B1 are all in the subgroup takes
that path (every other proof test of the suite runs it); the reference reads its parameters unchecked
(/root/reference/masp_proofs/src/lib.rs:343-347, Parameters::read(_, false)), so a CRS with a curve point OUTSIDE the subgroup
in the a query, the b_g1 query or the verifying key must fall back to the plain double-and-add and still give, byte for byte,
what the CPU restatement's complete group law gives."""
import random
import struct

import numpy as np
import pytest

import oracle_lib as O
import toy_r1cs
from pyref import F1, G1, P, R, ec_add, ec_mul, g1_unc

pytestmark = pytest.mark.gpu

T3 = (0, 2)          # a point of order 3 on y^2 = x^3 + 4: outside the subgroup of order r


def _sections(pbuf):
    """offsets of the point sections of a bellman Parameters buffer: name -> (offset of the first point, count, point size)"""
    out, off = {}, 864
    for name, size in (("ic", 96), ("h", 96), ("l", 96), ("a", 96), ("b_g1", 96), ("b_g2", 192)):
        n = struct.unpack(">I", pbuf[off:off + 4])[0]
        out[name] = (off + 4, n, size)
        off += 4 + n * size
    return out


def _g1_at(pbuf, off):
    return (int.from_bytes(pbuf[off:off + 48], "big"), int.from_bytes(pbuf[off + 48:off + 96], "big"))


def _outside(pt):
    """the same point moved out of the subgroup by the order-3 point: still on the curve"""
    q = ec_add(F1, pt, T3)
    assert q is not None and (q[1] * q[1] - q[0] ** 3 - 4) % P == 0
    assert ec_mul(F1, q, R) is not None          # not of order r
    return q


@pytest.fixture(scope="module")
def toy():
    cs, inputs, aux, _ = toy_r1cs.make(71, 4, 60, 500, bool_share=0.5)
    return cs, inputs, aux, O.generate_parameters(cs, toy_r1cs.toxic(71))


@pytest.fixture(scope="module")
def ctx():
    import masp_amd
    c = masp_amd.Context(0, slots=1)
    yield c
    c.close()


def _same_as_oracle(ctx, slot, pbuf, cs, inputs, aux, seed):
    ctx.load_circuit(slot, pbuf, cs)
    Pm = O.Params(pbuf)
    rng = random.Random(seed)
    # scalars around the split k = q u^2 + rem: 0, 1, u^2 - 1, u^2, u^2 + 1, multiples, the largest, random ones
    u2 = 0xd201000000010000 ** 2
    ks = [0, 1, u2 - 1, u2, u2 + 1, 2 * u2, (R // u2) * u2, R - 1, (1 << 128) - 1, 1 << 128] + [rng.randrange(R) for _ in range(6)]
    for n in (1, 9):                              # lone-proof mode and batch mode launch the kernel differently
        rs = [(ks[i % len(ks)], ks[(i + 3) % len(ks)]) for i in range(0, 2 * len(ks), 2)][:max(n, 8) if n > 1 else 8]
        if n == 1:
            for r, s in rs:
                assert ctx.prove_batch([(slot, inputs, aux, r, s)]) == [O.create_proof(Pm, cs, inputs, aux, r, s)], (r, s)
        else:
            rs = rs + [(rng.randrange(R), rng.randrange(R))]
            got = ctx.prove_batch([(slot, inputs, aux, r, s) for r, s in rs])
            assert got == [O.create_proof(Pm, cs, inputs, aux, r, s) for r, s in rs]


def test_subgroup_crs_takes_the_endomorphism_and_matches(ctx, toy):
    cs, inputs, aux, pbuf = toy
    _same_as_oracle(ctx, 2, pbuf, cs, inputs, aux, 1)
    assert ctx.circuit_uses_endomorphism(2) is True


@pytest.mark.parametrize("where", ["a", "b_g1", "alpha_g1", "delta_g1"])
def test_crs_point_outside_the_subgroup_falls_back_and_matches(ctx, toy, where):
    cs, inputs, aux, pbuf = toy
    pbuf = bytes(bytearray(pbuf))
    buf = bytearray(pbuf)
    sec = _sections(pbuf)
    if where in ("a", "b_g1"):
        off0, n, size = sec[where]
        assert n > 3
        off = off0 + 2 * size
    else:
        off = {"alpha_g1": 0, "delta_g1": 576}[where]
    buf[off:off + 96] = g1_unc(_outside(_g1_at(pbuf, off)))
    _same_as_oracle(ctx, 1, np.frombuffer(bytes(buf), np.uint8), cs, inputs, aux, 2)
    assert ctx.circuit_uses_endomorphism(1) is False

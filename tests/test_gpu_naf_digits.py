"""Width-w non-adjacent-form digits over a table per bit position (masp_amd/csrc/device/msm_geom.h `msm_geom_naf`, the digit scan in
device/msm_sort.cuh `MsmDigitIter`, the table build `k_msm_precompute_bits`, the odd bucket weights in `k_msm_combine`): an option of the
prover since round 5 (masp_hip_options::digit_recoding = 1; off by default — the tables outgrow the TLBs).  Through the C ABI (`masp_hip_msm_g{1,2}_multi` with
MASP_HIP_MSM_NAF | w) against the CPU restatement's multiexp (bellperson `multiexp` [EXT]): every exceptional case
of the bucket tree again under the new digits, scalars whose recoding carries through long runs of ones / ends at bit 255 / has the most
digits a width allows, lone-proof mode (np < 8: XYZZ accumulation, quad / oct tails) and batch mode, G1 and G2 — and a prover with
NAF digits (digit_recoding = 1) writing the same proof bytes as the default one, on a toy circuit and on the real Output circuit.  Run with `-m gpu` on an MI355X."""
import random

import numpy as np
import pytest

import oracle_lib as O
from pyref import R
from test_gpu_bucket_tree import _cases, _le, _rand

gpu = pytest.mark.gpu
NAF = 0x100   # MASP_HIP_MSM_NAF


@pytest.fixture(scope="module")
def ctxs():
    import masp_amd
    made = {"auto": masp_amd.Context(0), "off": masp_amd.Context(0, bucket_tree_levels=-1),
            "deep": masp_amd.Context(0, bucket_tree_levels=11, bucket_tree_sub_batch=5)}
    assert made["auto"].options["digit_recoding"] == 0
    yield made
    for c in made.values():
        c.close()


def naf_digits(k, w):
    """reference recoding (plain python): [(position, digit)], digit odd, |digit| < 2^(w-1)"""
    out, pos = [], 0
    while k:
        if k & 1:
            d = k & ((1 << w) - 1)
            if d >= 1 << (w - 1):
                d -= 1 << w
            out.append((pos, d))
            k -= d
        k >>= 1
        pos += 1
    return out


def _edge_scalars(w):
    e = [0, 1, 2, 3, R - 1, R - 2, (R - 1) // 2, 1 << 254, (1 << 254) + 1, (1 << 254) - 1, (1 << 200) - 1, 0xffffffff, 0xffffffff << 32,
         (1 << 64) - 1, (1 << 96) - (1 << 31), ((1 << 255) - 1) % R, 0x5555555555555555555555555555555555555555555555555555555555555555 % R,
         0x2aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa, (1 << (w - 1)) - 1, 1 << (w - 1), (1 << (w - 1)) + 1, (1 << w) - 1]
    # the most digits the width allows: one at every w-th position from 0 up
    e.append(sum(1 << p for p in range(0, 255, w)) % R)
    e.append(sum(((1 << (w - 1)) + 1) << p for p in range(0, 255 - w, w + 1)) % R)   # every digit negative, each one carrying into the next
    return e


def test_reference_recoding_is_what_the_header_says():
    rng = random.Random(5)
    for w in (4, 5, 8, 13, 17):
        ks = [rng.randrange(R) for _ in range(300)] + _edge_scalars(w)
        most = 0
        for k in ks:
            d = naf_digits(k, w)
            assert sum(v << p for p, v in d) == k and all(v & 1 and abs(v) < 1 << (w - 1) for _, v in d)
            assert all(b[0] - a[0] >= w for a, b in zip(d, d[1:])) and all(p <= 255 for p, _ in d)
            most = max(most, len(d))
        assert most <= 255 // w + 1       # MsmGeom::W of msm_geom_naf


@gpu
@pytest.mark.parametrize("w", [5, 8, 13])
@pytest.mark.parametrize("which", ["auto", "off", "deep"])
def test_g1_exceptional_pairs_under_naf_digits(ctxs, which, w):
    rng = random.Random(4000 + w)
    n = 640
    pm, sc = _cases(rng, n, O.g1_mul_gen_many)
    want = [O.msm_g1(pm, sc[p]) for p in range(sc.shape[0])]
    assert ctxs[which].msm_g1_multi(pm, sc, window_bits=NAF | w) == want
    one = O.g1_mul_gen_many(_rand(rng, 1))
    rep = np.tile(one, (n, 1))
    sc2 = np.stack([np.tile(_le(rng.randrange(R)), (n, 1)) for _ in range(8)] + [_rand(rng, n) for _ in range(3)])
    assert ctxs[which].msm_g1_multi(rep, sc2, window_bits=NAF | w) == [O.msm_g1(rep, sc2[p]) for p in range(sc2.shape[0])]
    mixed = pm.copy()
    mixed[3::5, :] = 0
    mixed[3::5, 0] = 0x40
    assert ctxs[which].msm_g1_multi(mixed, sc[3:13], window_bits=NAF | w) == [O.msm_g1(mixed, sc[p]) for p in range(3, 13)]


@gpu
@pytest.mark.parametrize("which", ["auto", "off", "deep"])
def test_g2_exceptional_pairs_under_naf_digits(ctxs, which):
    rng = random.Random(5000)
    n = 320
    pm, sc = _cases(rng, n, O.g2_mul_gen_many)
    sc = sc[:10]
    assert ctxs[which].msm_g2_multi(pm, sc, window_bits=NAF | 8) == [O.msm_g2(pm, sc[p]) for p in range(sc.shape[0])]


@gpu
@pytest.mark.parametrize("w", [4, 9, 13, 17])
def test_edge_scalars_lone_and_batch(ctxs, w):
    """carries through runs of ones, the last digit at bit 255, the most digits a width allows; np = 1, 3 (lone-proof mode) and 12 (batch)"""
    rng = random.Random(6000 + w)
    edge = _edge_scalars(w)
    n = 4 * len(edge)
    bases = O.g1_mul_gen_many(_rand(rng, n))
    vecs = []
    for p in range(12):
        v = [edge[(i + p) % len(edge)] if (i + p) % 3 else rng.randrange(R) for i in range(n)]
        vecs.append(np.stack([_le(x) for x in v]))
    sc = np.stack(vecs)
    want = [O.msm_g1(bases, sc[p]) for p in range(12)]
    ctx = ctxs["auto"]
    assert ctx.msm_g1_multi(bases, sc, window_bits=NAF | w) == want
    assert ctx.msm_g1_multi(bases, sc[:3], window_bits=NAF | w) == want[:3]
    assert ctx.msm_g1_multi(bases, sc[5:6], window_bits=NAF | w) == want[5:6]
    if w in (9, 13):
        b2 = O.g2_mul_gen_many(_rand(rng, 96))
        s2 = sc[:9, :96]
        w2 = [O.msm_g2(b2, s2[p]) for p in range(9)]
        assert ctx.msm_g2_multi(b2, s2, window_bits=NAF | w) == w2
        assert ctx.msm_g2_multi(b2, s2[:2], window_bits=NAF | w) == w2[:2]


@gpu
def test_widths_outside_the_range_are_refused(ctxs):
    import masp_amd
    rng = random.Random(1)
    bases = O.g1_mul_gen_many(_rand(rng, 8))
    sc = _rand(rng, 8)[None]
    for bad in (NAF | 3, NAF | 18, NAF | 0, 17, 0x209):
        with pytest.raises(masp_amd.MaspHipError):
            ctxs["auto"].msm_g1_multi(bases, sc, window_bits=bad)


@gpu
def test_fixed_windows_and_naf_digits_write_the_same_proofs():
    """a prover on NAF digits (digit_recoding = 1: per-bit tables for the sets a batch runs over) and the default one (fixed windows):
    same CRS, same jobs, same (r, s) -> the same 192 bytes, lone and as a batch"""
    import masp_amd
    import toy_r1cs
    cs, inputs, aux, vals = toy_r1cs.make(77, n_inputs=4, n_free=300, n_constraints=5000, bool_share=0.6)
    tw = toy_r1cs.toxic(77)
    a = masp_amd.Context(0, slots=1, digit_recoding=1)
    b = masp_amd.Context(0, slots=1)
    try:
        assert a.options["digit_recoding"] == 1 and b.options["digit_recoding"] == 0
        params = a.generate_parameters(cs, tw)
        a.load_circuit(0, params, cs)
        b.load_circuit(0, params, cs)
        jobs = [(0, inputs, aux, 100 + k, 200 + k) for k in range(12)]
        pa, pb = a.prove_batch(jobs), b.prove_batch(jobs)
        assert pa == pb
        assert pa[0] == O.create_proof(O.Params(params), cs, inputs, aux, 100, 200)
        assert a.prove(0, inputs, aux, 7, 9) == b.prove(0, inputs, aux, 7, 9) == O.create_proof(O.Params(params), cs, inputs, aux, 7, 9)
    finally:
        a.close()
        b.close()


@gpu
def test_output_circuit_under_naf_digits():
    """the real Output circuit (reference: masp_proofs/src/circuit/sapling.rs Output::synthesize, benches/sapling.rs) with NAF digits: 40
    distinct instances as one batch (h + l merged on width-16 NAF digits, a / b_g1 / b_g2 on width-13) and one lone proof — every proof
    == the toxic-waste closed form, two == the CPU restatement's create_proof, all through the batched pairing check"""
    import masp_amd
    from masp_amd import host as H
    from masp_amd import workload as W
    from masp_amd.synthetic import toxic_waste
    ctx = masp_amd.Context(0, slots=2, digit_recoding=1)
    try:
        cs = H.circuit("output")[0]
        tw = toxic_waste(41)
        params = ctx.generate_parameters(cs, tw)
        ctx.load_circuit(1, params, cs)
        insts = W.instances("output", 40, first_seed=8800)
        rng = random.Random(88)
        rs = [(rng.randrange(R), rng.randrange(R)) for _ in insts]
        proofs = ctx.prove_batch([(1, i, a, r, s) for (i, a), (r, s) in zip(insts, rs)])
        want = [O.closed_form_proof(cs, tw, i, a, r, s) for (i, a), (r, s) in zip(insts, rs)]
        assert proofs == want
        P = O.Params(params)
        for j in (0, 39):
            assert proofs[j] == O.create_proof(P, cs, insts[j][0], insts[j][1], *rs[j])
        assert H.PreparedVerifyingKey(params).verify_batch(proofs, [W.public_inputs(i) for i, _ in insts])
        assert ctx.prove(1, insts[3][0], insts[3][1], *rs[3]) == want[3]
    finally:
        ctx.close()

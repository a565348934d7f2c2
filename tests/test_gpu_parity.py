"""GPU parity tests proper: every call goes through the C ABI (include/masp_hip.h) of libmasp_hip.so and is
compared bit-for-bit with the oracle on the same seeded inputs.  Run with `-m gpu` on an MI355X."""
import random

import numpy as np
import pytest

import oracle_lib as O
import toy_r1cs
from pyref import R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import masp_amd
    c = masp_amd.Context(0)
    yield c
    c.close()


def _le(x):
    return np.frombuffer((x % R).to_bytes(32, "little"), np.uint8)


def _rand_scalars(rng, n, bool_share=0.0):
    out = np.zeros((n, 32), np.uint8)
    for i in range(n):
        u = rng.random()
        if u < bool_share:
            v = rng.randint(0, 1)
        else:
            v = rng.randrange(R)
        out[i] = _le(v)
    return out


@pytest.mark.parametrize("logm", [1, 4, 9, 10, 11, 13, 15, 16, 17])       # 15 / 16 / 17: the Output / Convert / Spend domains
def test_ntt_matches_oracle(ctx, logm):
    rng = np.random.default_rng(logm)
    m = 1 << logm
    data = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
    data[:, 31] &= 0x3f  # < 2^254 < r
    assert (ctx.ntt(data, logm) == O.ntt(data, logm)).all()
    assert (ctx.ntt(data, logm, inverse=True) == O.ntt(data, logm, inverse=True)).all()
    assert (ctx.ntt(ctx.ntt(data, logm), logm, inverse=True) == data).all()


@pytest.mark.parametrize("seed,n_inputs,n_free,n_constraints", [(21, 2, 3, 11), (22, 4, 30, 500), (23, 8, 200, 5000)])
def test_quotient_matches_oracle(ctx, seed, n_inputs, n_free, n_constraints):
    cs, inputs, aux, _ = toy_r1cs.make(seed, n_inputs, n_free, n_constraints)
    a, b, c, *_ = O.r1cs_eval(cs, inputs, aux)
    assert (ctx.quotient_h(a, b, c, cs.logm) == O.quotient_h(a, b, c, cs.logm)).all()


def test_quotient_full_size_domain(ctx):
    # 2^17 like Spend: random (unsatisfied) evaluation vectors — bellperson's sequence is defined for those too
    rng = np.random.default_rng(5)
    n = 100645
    vecs = []
    for _ in range(3):
        v = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        v[:, 31] &= 0x3f
        vecs.append(v)
    assert (ctx.quotient_h(*vecs, 17) == O.quotient_h(*vecs, 17)).all()


def _edge_scalars(n, rng):
    sc = _rand_scalars(rng, n, 0.4)
    edge = [0, 1, 2, R - 1, R - 2, (1 << 255) % R, 0xffff, 0x10000, 0x8000, 0x8001, (1 << 128) - 1, 1 << 240]
    for i, e in enumerate(edge[:n]):
        sc[i] = _le(e)
    return sc


@pytest.mark.parametrize("n", [0, 1, 2, 40, 300, 5000, 40000])
def test_msm_g1_matches_oracle(ctx, n):
    rng = random.Random(n + 1)
    ks = _rand_scalars(rng, n)
    bases = O.g1_mul_gen_many(ks) if n else np.zeros((0, 96), np.uint8)
    sc = _edge_scalars(n, rng)
    if n >= 40:
        bases[5] = bases[4]           # repeated base (P + P inside a bucket)
        sc[5] = sc[4]
        bases[7, :] = 0
        bases[7, 0] = 0x40            # base at infinity
    assert ctx.msm_g1(bases, sc) == O.msm_g1(bases, sc)


def test_msm_g1_all_equal_and_cancelling(ctx):
    # every scalar identical -> one bucket takes everything; and P, -P pairs cancel to infinity
    rng = random.Random(77)
    n = 600
    bases = O.g1_mul_gen_many(_rand_scalars(rng, n))
    sc = np.tile(_le(0x1234567), (n, 1))
    assert ctx.msm_g1(bases, sc) == O.msm_g1(bases, sc)
    half = n // 2
    ks = _rand_scalars(rng, half)
    kneg = np.stack([_le(R - int.from_bytes(k.tobytes(), "little")) for k in ks])
    b2 = O.g1_mul_gen_many(np.concatenate([ks, kneg]))
    s2 = np.tile(_le(3), (n, 1))
    assert ctx.msm_g1(b2, s2) == b"\x40" + bytes(95) == O.msm_g1(b2, s2)


def test_msm_g2_repeated_and_cancelling_points(ctx):
    """The exceptional cases of the G2 group law (one out-of-line call each since round 2: equal points -> doubling, opposite
    points -> infinity) in the accumulation (mixed addition of a repeated table row) and in the bucket tails (full addition
    of equal partial sums): the same point 600 times with one scalar, with different scalars, and P / -P pairs."""
    rng = random.Random(78)
    n = 600
    one = O.g2_mul_gen_many(_rand_scalars(rng, 1))
    same = np.tile(one, (n, 1))
    for sc in (np.tile(_le(0x1234567), (n, 1)), _rand_scalars(rng, n), _edge_scalars(n, rng)):
        assert ctx.msm_g2(same, sc) == O.msm_g2(same, sc)
    half = n // 2
    ks = _rand_scalars(rng, half)
    kneg = np.stack([_le(R - int.from_bytes(k.tobytes(), "little")) for k in ks])
    pm = O.g2_mul_gen_many(np.concatenate([ks, kneg]))
    s3 = np.tile(_le(3), (n, 1))
    assert ctx.msm_g2(pm, s3) == b"\x40" + bytes(191) == O.msm_g2(pm, s3)
    # two copies of each point with the same random scalar: every bucket entry meets its twin
    twice = np.concatenate([pm[:half], pm[:half]])
    sc2 = _rand_scalars(rng, half)
    sc2 = np.concatenate([sc2, sc2])
    assert ctx.msm_g2(twice, sc2) == O.msm_g2(twice, sc2)


@pytest.mark.parametrize("n", [1, 33, 700, 9000, 62170])             # 62 170: the Spend circuit's b_g2 query, against the CPU multiexp itself
def test_msm_g2_matches_oracle(ctx, n):
    rng = random.Random(n + 100)
    bases = O.g2_mul_gen_many(_rand_scalars(rng, n))
    sc = _edge_scalars(n, rng)
    assert ctx.msm_g2(bases, sc) == O.msm_g2(bases, sc)


@pytest.mark.parametrize("group,n,bool_share", [("g1", 131071, 0.0), ("g1", 100497, 0.7), ("g2", 62170, 0.8)])
def test_msm_full_size_by_discrete_log_checksum(ctx, group, n, bool_share):
    """BASELINE sizes (Spend: H 131 071 uniform scalars on 16-bit windows; L / B2 with the witness' 0/1 share on 12-bit
    windows) checked through a size-independent identity instead of a CPU MSM: with bases P_i = k_i G,
    sum_i s_i P_i = (sum_i s_i k_i mod r) G."""
    rng = random.Random(n)
    ks = _rand_scalars(rng, n)
    sc = _edge_scalars(n, rng) if bool_share == 0.0 else _rand_scalars(rng, n, bool_share)
    k_int = [int.from_bytes(k.tobytes(), "little") for k in ks]
    s_int = [int.from_bytes(x.tobytes(), "little") for x in sc]
    total = sum(a * b for a, b in zip(k_int, s_int)) % R
    expect_scalar = np.frombuffer(_le(total), np.uint8).reshape(1, 32)
    if group == "g1":
        assert ctx.msm_g1(O.g1_mul_gen_many(ks), sc) == O.g1_mul_gen_many(expect_scalar)[0].tobytes()
    else:
        assert ctx.msm_g2(O.g2_mul_gen_many(ks), sc) == O.g2_mul_gen_many(expect_scalar)[0].tobytes()


def test_msm_g1_beyond_the_two_pass_sort_range(ctx):
    """1 126 400 points on 16-bit windows: n x W = 18 M digit entries do not fit the 24-bit row index of the two-pass
    placement, so the counting sort takes its single-pass scatter (the MASP circuits never get there: at most 3.7 M entries;
    the toy circuits reach the same kernel through their narrow windows).  4 096 distinct bases repeated 275 times, checked
    by the discrete-log identity."""
    rng = random.Random(77)
    distinct, reps = 4096, 275
    ks = _rand_scalars(rng, distinct)
    base = O.g1_mul_gen_many(ks)
    n = distinct * reps
    bases = np.tile(base, (reps, 1))
    raw = np.random.default_rng(77).integers(0, 256, size=(n, 32), dtype=np.uint8)
    raw[:, 31] &= 0x3f                                     # < 2^254 < r
    raw[::1000] = 0                                        # some zeros and ones among them
    raw[1::1000] = 0
    raw[1::1000, 0] = 1
    k_int = [int.from_bytes(k.tobytes(), "little") for k in ks]
    s_int = [int.from_bytes(raw[i].tobytes(), "little") for i in range(n)]
    total = sum(k_int[i % distinct] * s for i, s in enumerate(s_int)) % R
    expect_scalar = np.frombuffer(_le(total), np.uint8).reshape(1, 32)
    assert ctx.msm_g1(bases, raw) == O.g1_mul_gen_many(expect_scalar)[0].tobytes()


@pytest.mark.parametrize("pattern", ["one_value", "all_ones", "all_r_minus_1", "two_values_and_zeros"])
def test_msm_skewed_digit_distributions(ctx, pattern):
    """Digit distributions that put everything into a handful of buckets (the balanced chunking, the heavy-bucket path
    and the signed-digit carries at 70 000 points on 16-bit windows), checked by the discrete-log identity."""
    rng = random.Random({"one_value": 1, "all_ones": 2, "all_r_minus_1": 3, "two_values_and_zeros": 4}[pattern])
    n = 70000
    ks = _rand_scalars(rng, n)
    v = rng.randrange(R)
    if pattern == "one_value":
        s_int = [v] * n
    elif pattern == "all_ones":
        s_int = [1] * n
    elif pattern == "all_r_minus_1":
        s_int = [R - 1] * n
    else:
        w = (1 << 255) % R
        s_int = [(v, w, 0, 0)[rng.randrange(4)] for _ in range(n)]
    sc = np.stack([_le(x) for x in s_int])
    k_int = [int.from_bytes(k.tobytes(), "little") for k in ks]
    total = sum(a * b for a, b in zip(k_int, s_int)) % R
    expect = O.g1_mul_gen_many(np.frombuffer(_le(total), np.uint8).reshape(1, 32))[0].tobytes()
    assert ctx.msm_g1(O.g1_mul_gen_many(ks), sc) == expect


@pytest.mark.parametrize("seed,n_inputs,n_free,n_constraints", [(31, 2, 4, 9), (32, 4, 20, 200), (33, 8, 300, 3000)])
def test_proof_bytes_match_oracle_and_closed_form(ctx, seed, n_inputs, n_free, n_constraints):
    cs, inputs, aux, vals = toy_r1cs.make(seed, n_inputs, n_free, n_constraints, bool_share=0.7)
    tw = toy_r1cs.toxic(seed)
    pbuf = O.generate_parameters(cs, tw)
    ctx.load_circuit(3, pbuf, cs)
    rng = random.Random(seed)
    r, s = rng.randrange(R), rng.randrange(R)
    expect = O.create_proof(O.Params(pbuf), cs, inputs, aux, r, s)
    assert expect == O.closed_form_proof(cs, tw, inputs, aux, r, s)
    got = ctx.prove(3, inputs, aux, r, s)               # a,b,c computed on the GPU from the static R1CS
    assert got == expect
    a, b, c, *_ = O.r1cs_eval(cs, inputs, aux)
    assert ctx.prove(3, inputs, aux, r, s, (a, b, c)) == expect   # caller-supplied evaluation vectors
    assert O.verify_proof(pbuf, got, vals[1:n_inputs]) == 1
    # r = s = 0 (no blinding) and extreme blinding
    for rr, ss in ((0, 0), (R - 1, 1)):
        assert ctx.prove(3, inputs, aux, rr, ss) == O.create_proof(O.Params(pbuf), cs, inputs, aux, rr, ss)


def test_caller_supplied_abc_evaluations(ctx):
    """masp_hip_prove with a, b, c given (a caller whose constraint system evaluated the rows itself, SURVEY.md §8b
    "a,b,c or NULL") == the same proof with the rows evaluated on the GPU from the static R1CS; mixed in one batch too."""
    cs, inputs, aux, vals = toy_r1cs.make(35, 5, 40, 700)
    pbuf = O.generate_parameters(cs, toy_r1cs.toxic(35))
    ctx.load_circuit(3, pbuf, cs)
    a, b, c, *_ = O.r1cs_eval(cs, inputs, aux)
    want = O.create_proof(O.Params(pbuf), cs, inputs, aux, 21, 22)
    assert ctx.prove(3, inputs, aux, 21, 22, abc=(a, b, c)) == want == ctx.prove(3, inputs, aux, 21, 22)
    jobs = [(3, inputs, aux, 21, 22, (a, b, c)), (3, inputs, aux, 21, 22), (3, inputs, aux, 23, 24, (a, b, c))]
    got = ctx.prove_batch(jobs)
    assert got[0] == got[1] == want and got[2] == O.create_proof(O.Params(pbuf), cs, inputs, aux, 23, 24)


def test_unsatisfied_assignment_matches_oracle(ctx):
    # the reference bench proves an unsatisfiable witness (benches/sapling.rs:41,69): bytes must still agree
    cs, inputs, aux, vals = toy_r1cs.make(41, 3, 10, 120)
    aux = aux.copy()
    aux[-1, 0] ^= 1
    pbuf = O.generate_parameters(cs, toy_r1cs.toxic(41))
    ctx.load_circuit(4, pbuf, cs)
    got = ctx.prove(4, inputs, aux, 11, 12)
    assert got == O.create_proof(O.Params(pbuf), cs, inputs, aux, 11, 12)
    assert O.verify_proof(pbuf, got, vals[1:3]) == 0


def test_batch_mixed_circuits_in_job_order(ctx):
    specs = [(51, 2, 5, 40), (52, 5, 9, 90)]
    loaded = []
    for slot, (seed, ni, nf, nc) in enumerate(specs):
        cs, inputs, aux, vals = toy_r1cs.make(seed, ni, nf, nc)
        pbuf = O.generate_parameters(cs, toy_r1cs.toxic(seed))
        ctx.load_circuit(5 + slot, pbuf, cs)
        loaded.append((cs, inputs, aux, O.Params(pbuf)))
    jobs, expect = [], []
    for j in range(9):
        which = j % 2
        cs, inputs, aux, params = loaded[which]
        r, s = 1000 + j, 2000 + 7 * j
        jobs.append((5 + which, inputs, aux, r, s))
        expect.append(O.create_proof(params, cs, inputs, aux, r, s))
    assert ctx.prove_batch(jobs) == expect
    h, n = ctx.batch_upload(jobs)
    got, ms = ctx.batch_prove_resident(h, n)
    assert got == expect and ms > 0
    ctx.batch_free(h)


def test_error_paths(ctx):
    import masp_amd
    cs, inputs, aux, _ = toy_r1cs.make(61, 2, 4, 12)
    pbuf = O.generate_parameters(cs, toy_r1cs.toxic(61))
    with pytest.raises(masp_amd.MaspHipError) as e:
        ctx.prove(7, inputs, aux, 1, 2)
    assert e.value.code == 7                      # slot empty
    with pytest.raises(masp_amd.MaspHipError) as e:
        ctx.load_circuit(7, pbuf[:500], cs)
    assert e.value.code == 2                      # truncated params
    other, *_ = toy_r1cs.make(62, 2, 4, 30)
    with pytest.raises(masp_amd.MaspHipError) as e:
        ctx.load_circuit(7, pbuf, other)
    assert e.value.code == 3                      # shape mismatch
    ctx.load_circuit(7, pbuf, cs)
    bad = aux.copy()
    bad[0] = 0xff                                 # >= r
    with pytest.raises(masp_amd.MaspHipError) as e:
        ctx.prove(7, inputs, bad, 1, 2)
    assert e.value.code == 8
    with pytest.raises(masp_amd.MaspHipError) as e:
        ctx.prove(7, inputs, aux, R, 2)
    assert e.value.code == 8


@pytest.mark.parametrize("seed,n_inputs,n_free,n_constraints", [(71, 2, 4, 9), (72, 6, 50, 600)])
def test_generate_parameters_matches_oracle(ctx, seed, n_inputs, n_free, n_constraints):
    cs, inputs, aux, vals = toy_r1cs.make(seed, n_inputs, n_free, n_constraints)
    tw = toy_r1cs.toxic(seed)
    assert ctx.generate_parameters(cs, tw).tobytes() == O.generate_parameters(cs, tw).tobytes()


def test_spend_shaped_full_size_proof(ctx):
    """BASELINE.json configs[1] sizes: NTT 2^17, G1 MSMs 131071/100497/86931/62170, G2 MSM 62170."""
    from masp_amd import synthetic
    cs, inputs, aux = synthetic.shaped("spend", seed=3)
    tw = synthetic.toxic_waste(9)
    params = ctx.generate_parameters(cs, tw)
    assert params.size == 48482520          # params body bytes of the real Spend file
    ctx.load_circuit(0, params, cs)
    r, s = 0x1234567890abcdef, 0xfedcba0987654321
    proof = ctx.prove(0, inputs, aux, r, s)
    assert proof == O.closed_form_proof(cs, tw, inputs, aux, r, s)      # oracle 1: no NTT / MSM involved
    pub = [int.from_bytes(inputs[i].tobytes(), "little") for i in range(1, cs.n_inputs)]
    assert O.verify_proof(params[:868 + 96 * cs.n_inputs], proof, pub) == 1   # oracle 2: pairing check
    assert proof == O.create_proof(O.Params(params), cs, inputs, aux, r, s)  # oracle 3: CPU restatement


def test_local_tx_prover_real_circuits_match_oracle_and_verify(ctx):
    """LocalTxProver::{spend_proof, output_proof, convert_proof} on the REAL MASP circuits (structure hashes pinned by
    tests/test_circuits.py): proof bytes equal the CPU restatement and the toxic-waste closed form, and pass the
    pairing check with the public inputs the reference's verifier would build (sapling/prover.rs:121-145,256-263)."""
    import random
    from masp_amd import host as H
    from masp_amd import prover as P
    from masp_amd.synthetic import toxic_waste
    from test_circuits import spend_instance
    rng = random.Random(77)
    lp = P.LocalTxProver.with_synthetic_parameters(seed=5)
    pc = lp.new_sapling_proving_context()
    # ---- Spend (config 2 of BASELINE.json)
    inst, cmu, pk_d = spend_instance(300, value=42)
    r, s = rng.randrange(H.FR_MODULUS), rng.randrange(H.FR_MODULUS)
    zk, cv, rk = lp.spend_proof(pc, (inst["ak"], inst["nsk"]), inst["diversifier"], inst["rcm"], inst["ar"], inst["asset_identifier"],
                                inst["value"], inst["anchor"], (inst["path_siblings"], inst["position"]), inst["rcv"], rs=(r, s))
    assert len(zk) == P.GROTH_PROOF_SIZE
    cs, _ = H.circuit("spend")
    inputs, aux, cv2, rk2, nf = H.spend_assignment(check=False, **inst)
    assert (cv, rk) == (cv2, rk2)
    params = lp.parameters["spend"]
    assert params.size == 48482520
    assert zk == O.create_proof(O.Params(params), cs, inputs, aux, r, s)
    assert zk == O.closed_form_proof(cs, toxic_waste(15), inputs, aux, r, s)
    pub = [int.from_bytes(inputs[i].tobytes(), "little") for i in range(1, 8)]
    assert O.verify_proof(params[:868 + 96 * 8], zk, pub) == 1
    assert pc.bsk == inst["rcv"] % H.JUBJUB_ORDER and pc.cv_sum == cv
    # ---- Output
    ident = H.asset_identifier(b"benchmark")
    while True:
        d = bytes(rng.getrandbits(8) for _ in range(11))
        try:
            pk = H.jubjub_mul(H.point_bytes(*H.generator_uv(0)), rng.randrange(1, H.JUBJUB_ORDER))
            args = dict(esk=rng.randrange(1, H.JUBJUB_ORDER), payment_address=(d, pk), rcm=rng.randrange(1, H.JUBJUB_ORDER),
                        asset_type=ident, value=7, rcv=rng.randrange(1, H.JUBJUB_ORDER))
            zk_o, cv_o = lp.output_proof(pc, rs=(r, s), **args)
            break
        except P.ProvingError:
            continue
    cs_o, _ = H.circuit("output")
    i_o, a_o, _ = H.output_assignment(args["esk"], d, pk, args["rcm"], ident, 7, args["rcv"])
    po = lp.parameters["output"]
    assert po.size == 15032568 and zk_o == O.create_proof(O.Params(po), cs_o, i_o, a_o, r, s)
    assert O.verify_proof(po[:868 + 96 * 6], zk_o, [int.from_bytes(i_o[i].tobytes(), "little") for i in range(1, 6)]) == 1
    assert pc.bsk == (inst["rcv"] - args["rcv"]) % H.JUBJUB_ORDER and pc.cv_sum == H.jubjub_add(cv, cv_o, subtract=True)
    # ---- Convert
    gen = H.asset_generator(H.asset_identifier(b"asset 1"))
    sib = [rng.randrange(H.FR_MODULUS) for _ in range(32)]
    pos = rng.getrandbits(32)
    anchor = H.merkle_root(H.convert_cmu(gen), sib, pos)
    rcv = rng.randrange(1, H.JUBJUB_ORDER)
    zk_c, cv_c = lp.convert_proof(pc, gen, 99, anchor, (sib, pos), rcv, rs=(r, s))
    cs_c, _ = H.circuit("convert")
    i_c, a_c, _ = H.convert_assignment(gen, 99, anchor, sib, pos, rcv)
    pcv = lp.parameters["convert"]
    assert pcv.size == 21204888 and zk_c == O.create_proof(O.Params(pcv), cs_c, i_c, a_c, r, s)
    assert O.verify_proof(pcv[:868 + 96 * 4], zk_c, [int.from_bytes(i_c[i].tobytes(), "little") for i in range(1, 4)]) == 1
    # an unsatisfiable statement (wrong anchor, value != 0) still "proves" but fails the self-check -> Err(())
    with pytest.raises(P.ProvingError):
        lp.spend_proof(pc, (inst["ak"], inst["nsk"]), inst["diversifier"], inst["rcm"], inst["ar"], inst["asset_identifier"], inst["value"],
                       (int.from_bytes(inst["anchor"], "little") + 1) % H.FR_MODULUS, (inst["path_siblings"], inst["position"]), inst["rcv"])
    # invalid diversifier -> Err(())
    bad = next(bytes([k]) * 11 for k in range(256) if _invalid_diversifier(bytes([k]) * 11, inst))
    with pytest.raises(P.ProvingError):
        lp.spend_proof(pc, (inst["ak"], inst["nsk"]), bad, inst["rcm"], inst["ar"], inst["asset_identifier"], 1, inst["anchor"],
                       (inst["path_siblings"], inst["position"]), inst["rcv"])
    lp.close()


def _invalid_diversifier(d, inst):
    from masp_amd import host as H
    try:
        H.spend_leaf(inst["ak"], inst["nsk"], d, inst["rcm"], inst["asset_identifier"], 1)
        return False
    except H.HostError:
        return True


def test_local_tx_prover_from_parameter_files(ctx, tmp_path):
    """LocalTxProver::new = load_parameters (lib.rs:278-328): files with an MPC-transcript tail after the Parameters body,
    sizes checked first, BLAKE2b-512 over body + transcript; the loaded prover yields the same proof bytes."""
    import hashlib
    from masp_amd import host as H
    from masp_amd import params as PP
    from masp_amd import prover as P
    from test_circuits import spend_instance
    lp0 = P.LocalTxProver.with_synthetic_parameters(seed=9)
    blobs, paths, exp = {}, [], {}
    for i, kind in enumerate(PP.KINDS):
        blobs[kind] = lp0.parameters[kind].tobytes() + b"transcript of contribution %d" % i * 1000
        path = tmp_path / PP.EXPECTED[kind].name
        path.write_bytes(blobs[kind])
        paths.append(str(path))
        exp[kind] = PP.Expected(PP.EXPECTED[kind].name, hashlib.blake2b(blobs[kind], digest_size=64).hexdigest(), len(blobs[kind]))
    inst, _, _ = spend_instance(500, value=3)
    args = ((inst["ak"], inst["nsk"]), inst["diversifier"], inst["rcm"], inst["ar"], inst["asset_identifier"], inst["value"], inst["anchor"],
            (inst["path_siblings"], inst["position"]), inst["rcv"])
    want = lp0.spend_proof(lp0.new_sapling_proving_context(), *args, rs=(11, 12))
    lp0.close()
    lp = P.LocalTxProver.new(*paths, expected=exp)
    assert lp.spend_proof(lp.new_sapling_proving_context(), *args, rs=(11, 12)) == want
    lp.close()
    with pytest.raises(PP.ParameterError):                 # the real MPC digests cannot match synthetic files
        P.LocalTxProver.new(*paths)
    bad = dict(exp, output=PP.Expected(exp["output"].name, "00" * 64, exp["output"].bytes))
    with pytest.raises(PP.ParameterError, match="failed validation"):
        P.LocalTxProver.new(*paths, expected=bad)
    (tmp_path / exp["convert"].name).write_bytes(blobs["convert"][:-1])
    with pytest.raises(PP.ParameterError, match="bytes"):
        P.LocalTxProver.new(*paths, expected=exp)
    assert P.LocalTxProver.from_bytes(blobs["spend"], blobs["output"], blobs["convert"], expected=None) is not None


def test_concurrent_provers_share_one_context(ctx):
    """SURVEY.md §8(b) threading: one prover shared by several host threads — the native context is re-entrant, every
    caller gets its own proofs back, bit-identical to the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    cs, inputs, aux, vals = toy_r1cs.make(61, 3, 12, 150)
    pbuf = O.generate_parameters(cs, toy_r1cs.toxic(61))
    ctx.load_circuit(7, pbuf, cs)
    params = O.Params(pbuf)

    def caller(t):
        jobs = [(7, inputs, aux, 100 * t + j + 1, 7000 + 100 * t + j) for j in range(1 + (t * 5) % 13)]
        return ctx.prove_batch(jobs), [O.create_proof(params, cs, inputs, aux, r, s) for (_, _, _, r, s) in jobs]
    with ThreadPoolExecutor(8) as ex:
        for got, expect in ex.map(caller, range(24)):
            assert got == expect


def test_local_tx_prover_batch_equals_serial(ctx):
    """prove_batch (threaded synthesis + one GPU batch) == the serial TxProver calls, including the context state."""
    import random
    from masp_amd import host as H
    from masp_amd import prover as P
    from test_circuits import spend_instance
    rng = random.Random(5)
    lp = P.LocalTxProver.with_synthetic_parameters(seed=6)
    descs, rs = [], []
    for k in range(3):
        inst, _, _ = spend_instance(400 + k, value=10 + k)
        descs.append(("spend", dict(proof_generation_key=(inst["ak"], inst["nsk"]), diversifier=inst["diversifier"], rcm=inst["rcm"],
                                    ar=inst["ar"], asset_type=inst["asset_identifier"], value=inst["value"], anchor=inst["anchor"],
                                    merkle_path=(inst["path_siblings"], inst["position"]), rcv=inst["rcv"])))
    gen = H.asset_generator(H.asset_identifier(b"asset 2"))
    sib = [rng.randrange(H.FR_MODULUS) for _ in range(32)]
    anchor = H.merkle_root(H.convert_cmu(gen), sib, 77)
    descs.insert(1, ("convert", dict(allowed_conversion=gen, value=5, anchor=anchor, merkle_path=(sib, 77), rcv=rng.randrange(1, H.JUBJUB_ORDER))))
    rs = [(rng.randrange(H.FR_MODULUS), rng.randrange(H.FR_MODULUS)) for _ in descs]
    c1 = lp.new_sapling_proving_context()
    seen = []
    batch = lp.prove_batch(c1, descs, rs=rs, chunk=3, progress=lambda done, total: seen.append((done, total)))
    assert sorted(seen) == [(1, 4), (4, 4)] or sorted(seen) == [(3, 4), (4, 4)]
    c2 = lp.new_sapling_proving_context()
    serial = []
    for (kind, kw), r in zip(descs, rs):
        if kind == "spend":
            serial.append(lp.spend_proof(c2, kw["proof_generation_key"], kw["diversifier"], kw["rcm"], kw["ar"], kw["asset_type"], kw["value"],
                                         kw["anchor"], kw["merkle_path"], kw["rcv"], rs=r))
        else:
            serial.append(lp.convert_proof(c2, kw["allowed_conversion"], kw["value"], kw["anchor"], kw["merkle_path"], kw["rcv"], rs=r))
    assert batch == serial and (c1.bsk, c1.cv_sum) == (c2.bsk, c2.cv_sum)
    # one bad description (invalid diversifier) fails the whole call with Err(()) — and does not leave the pipeline hanging
    inst = spend_instance(400, value=10)[0]
    bad_d = next(bytes([k]) * 11 for k in range(256) if _invalid_diversifier(bytes([k]) * 11, inst))
    bad = ("spend", dict(descs[0][1], diversifier=bad_d))
    with pytest.raises(P.ProvingError):
        lp.prove_batch(lp.new_sapling_proving_context(), descs * 8 + [bad] + descs * 8, chunk=4, threads=4)
    # every page-locked aux buffer is back in the pool (reserved in slabs before the GPU got busy, handed out, returned):
    # a second failing call neither allocates more nor loses any
    pooled = {k: len(v) for k, v in lp._pool.items()}
    slabs = len(lp._ctx._pinned)
    with pytest.raises(P.ProvingError):
        lp.prove_batch(lp.new_sapling_proving_context(), descs * 8 + [bad] + descs * 8, chunk=4, threads=4)
    assert {k: len(v) for k, v in lp._pool.items()} == pooled and len(lp._ctx._pinned) == slabs
    assert lp.prove_batch(lp.new_sapling_proving_context(), descs, rs=rs) == serial        # and the prover is still usable
    lp.close()


def test_aux_as_montgomery_residues_gives_the_same_proofs(ctx):
    """masp_hip_job::aux_form = MASP_HIP_AUX_MONTGOMERY: the aux assignment arrives as Montgomery residues (blst_fr memory) and is
    made canonical on the device before anything reads it as scalars.  Same bytes as the canonical hand-over, lone and in a
    batch, mixed with canonical jobs in one call; a residue >= r is refused like a non-canonical scalar."""
    import masp_amd
    from masp_amd import host as H
    from masp_amd import workload as W
    from masp_amd.synthetic import toxic_waste
    cs = H.circuit("output")[0]
    params = ctx.generate_parameters(cs, toxic_waste(91))
    ctx.load_circuit(1, params, cs)
    rng = random.Random(12)
    descs = [W.description("output", 500 + i)[1] for i in range(12)]

    def asg(kw, mont):
        d, pk = kw["payment_address"]
        return H.output_assignment(kw["esk"], d, pk, kw["rcm"], kw["asset_type"], kw["value"], kw["rcv"], montgomery=mont)[:2]
    canon = [asg(kw, False) for kw in descs]
    mont = [asg(kw, True) for kw in descs]
    rs = [(rng.randrange(R), rng.randrange(R)) for _ in descs]
    want = ctx.prove_batch([(1, i, a, r, s) for (i, a), (r, s) in zip(canon, rs)])
    got = ctx.prove_batch([(1, i, a, r, s, None, 1) for (i, a), (r, s) in zip(mont, rs)])
    assert got == want
    assert ctx.prove_batch([(1, mont[0][0], mont[0][1], rs[0][0], rs[0][1], None, 1)]) == want[:1]          # lone-proof mode
    mixed = [(1, *(mont[j] if j % 2 else canon[j]), rs[j][0], rs[j][1], None, j % 2) for j in range(12)]
    assert ctx.prove_batch(mixed) == want
    bad = mont[1][1].copy()
    bad[5] = np.frombuffer((R + 1).to_bytes(32, "little"), np.uint8)
    with pytest.raises(masp_amd.MaspHipError) as e:
        ctx.prove_batch([(1, mont[1][0], bad, 1, 2, None, 1)])
    assert e.value.code == 8


def test_empty_and_ragged_job_lists(ctx):
    """Edge shapes of masp_hip_prove_batch: no jobs at all, one job, the last lone-proof size (7), the first batch-mode size (8: the
    merged h + l MSM, the bucket tree), 9 — every proof equal to the CPU restatement's for the same (r, s); an all-zero and an
    all-one auxiliary assignment (every witness MSM degenerates: no digits at all / one bucket); the empty verification batch."""
    cs, inputs, aux, vals = toy_r1cs.make(61, 4, 50, 600, bool_share=0.6)
    tw = toy_r1cs.toxic(61)
    pbuf = O.generate_parameters(cs, tw)
    ctx.load_circuit(3, pbuf, cs)
    P = O.Params(pbuf)
    assert ctx.prove_batch([]) == []
    rng = random.Random(61)
    for n in (1, 7, 8, 9):
        rs = [(rng.randrange(R), rng.randrange(R)) for _ in range(n)]
        got = ctx.prove_batch([(3, inputs, aux, r, s) for r, s in rs])
        assert got == [O.create_proof(P, cs, inputs, aux, r, s) for r, s in rs], n
    # degenerate assignments (not satisfying: Groth16's prover is defined for them all the same; the bytes must still agree)
    for fill in (0, 1):
        flat = np.zeros_like(aux)
        flat[:, 0] = fill
        for n in (1, 9):
            got = ctx.prove_batch([(3, inputs, flat, 5 + k, 6 + k) for k in range(n)])
            assert got == [O.create_proof(P, cs, inputs, flat, 5 + k, 6 + k) for k in range(n)], (fill, n)
    gvk = ctx.prepare_verifying_key(pbuf)
    try:
        assert gvk.verify_batch([], []) is True
    finally:
        gvk.close()

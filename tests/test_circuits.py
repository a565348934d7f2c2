"""The host-side circuit synthesizer and native primitives (masp_amd/csrc/host) against the reference's own KATs
(tests/golden/*.json, extracted from /root/reference by tests/golden/make_fixtures.py).  CPU only."""
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
from masp_amd import host as H

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RJ = H.JUBJUB_ORDER
R = H.FR_MODULUS


def load(name):
    return json.load(open(os.path.join(G, name)))


@pytest.mark.parametrize("kind", ["spend", "output", "convert"])
def test_constraint_system_hash_and_counts(kind):
    # circuit/sapling.rs:730-741,923-931,1024-1045 ; circuit/convert.rs:218-224
    exp = load("circuits.json")[kind]
    cs, h = H.circuit(kind)
    assert h == exp["hash"]
    assert cs.n_constraints == exp["constraints"] and cs.n_inputs == exp["inputs"]


@pytest.mark.parametrize("kind", ["spend", "output", "convert"])
def test_prover_sizes_close_the_parameter_file_size_equations(kind):
    # body = vk + h + l + a + b_g1 + b_g2 ; file = body + MPC transcript (lib.rs:74-76)
    from masp_amd.synthetic import SHAPES
    exp = load("circuits.json")
    cs, _ = H.circuit(kind)
    a_dense = np.zeros(cs.n_inputs + cs.n_aux, bool)
    b_dense = np.zeros(cs.n_inputs + cs.n_aux, bool)
    a_dense[cs.mats[0][1]] = True
    b_dense[cs.mats[1][1]] = True
    a_aux, b_in, b_aux = int(a_dense[cs.n_inputs:].sum()), int(b_dense[:cs.n_inputs].sum()), int(b_dense[cs.n_inputs:].sum())
    assert (cs.n_inputs, cs.n_aux, cs.n_constraints, a_aux, b_aux) == SHAPES[kind][:5] and b_in == 1
    m = 1 << cs.logm
    body = 868 + 96 * cs.n_inputs + 5 * 4 + 96 * ((m - 1) + cs.n_aux + (cs.n_inputs + a_aux) + (b_in + b_aux)) + 192 * (b_in + b_aux)
    assert body + exp["mpc_transcript_bytes"] == exp[kind]["params_file_bytes"]


def test_generators_match_reference_constants():
    # masp_primitives/src/constants.rs:50-251, derivations :323-374
    g = load("generators.json")
    for i, name in enumerate(H.GENERATOR_NAMES):
        assert H.generator_uv(i) == (int(g[name]["u"], 16), int(g[name]["v"], 16)), name
    for k, p in enumerate(g["pedersen_hash_generators"]):
        assert H.generator_uv(5 + k) == (int(p["u"], 16), int(p["v"], 16))


def test_pedersen_hash_vectors():
    # masp_primitives/src/test_vectors/pedersen_hash_vectors.rs via sapling/pedersen_hash.rs:133-154
    vs = load("pedersen_hash_vectors.json")
    assert len(vs) == 37
    for v in vs:
        pers = v["personalization"]
        expect_prefix = [1] * 6 if pers < 0 else [(pers >> i) & 1 for i in range(6)]
        assert v["input_bits"][:6] == expect_prefix
        assert H.pedersen_hash(pers, v["input_bits"][6:]) == (int(v["u"], 16), int(v["v"], 16))


def test_value_commitment_kats():
    # masp_proofs/src/circuit/sapling.rs:783-817
    k = load("value_commitments.json")
    ident = bytes.fromhex(k["asset_identifier"])
    for i in range(10):
        _, u, v = H.value_commitment(ident, i, 1000 * (i + 1))
        assert (u, v) == (int(k["u"][i]), int(k["v"][i]))


def test_note_commitment_and_value_commitment_vectors():
    # masp_primitives/src/test_vectors/note_encryption.rs, checked at sapling/note_encryption.rs:1357-1360
    k = load("note_vectors.json")
    ident = bytes.fromhex(k["asset_identifier"])
    for tv in k["vectors"]:
        d, pk_d, rcm = bytes.fromhex(tv["default_d"]), bytes.fromhex(tv["default_pk_d"]), bytes.fromhex(tv["rcm"])
        assert H.note_cmu(ident, tv["v"], d, pk_d, rcm).hex() == tv["cmu"]
        # pk_d = [ivk] g_d  and  epk = [esk] g_d
        # (g_d itself is exercised through the note commitment; recover it from pk_d's derivation is not possible,
        #  so check the DH relation epk = [esk] g_d through the Output witness below)


def _rand_scalar(rng):
    return rng.randrange(1, RJ)


def spend_instance(seed, value=1):
    """A random valid Spend instance shaped like the reference bench's (benches/sapling.rs:39-69), with a real anchor."""
    rng = random.Random(seed)
    ident = H.asset_identifier(b"benchmark")          # AssetType::new(b"benchmark")
    ak = H.jubjub_mul(H.point_bytes(*H.generator_uv(4)), _rand_scalar(rng))      # ak = [ask] G_spend: prime order
    nsk, ar, rcm, rcv = (_rand_scalar(rng) for _ in range(4))
    siblings = [rng.randrange(R) for _ in range(32)]
    pos = rng.getrandbits(32)
    while True:
        d = bytes(rng.getrandbits(8) for _ in range(11))
        try:
            cmu, pk_d = H.spend_leaf(ak, nsk, d, rcm, ident, value)
            break
        except H.HostError as e:
            assert e.code == 2      # invalid diversifier: draw another, like the reference's tests do
    anchor = H.merkle_root(cmu, siblings, pos)
    return dict(ak=ak, nsk=nsk, diversifier=d, rcm=rcm, ar=ar, asset_identifier=ident, value=value, anchor=anchor,
                path_siblings=siblings, position=pos, rcv=rcv), cmu, pk_d


def test_spend_witness_satisfies_circuit_and_public_inputs():
    inst, cmu, pk_d = spend_instance(100, value=123456789)
    inputs, aux, cv, rk, nf = H.spend_assignment(check=True, **inst)
    cs, _ = H.circuit("spend")
    assert O.r1cs_unsatisfied(cs, inputs, aux) == 0          # independent evaluation by the oracle
    # public inputs: rk.u, rk.v, cv.u, cv.v, anchor, nf packed into two 254-bit chunks (sapling/prover.rs:121-145)
    pub = [int.from_bytes(inputs[i].tobytes(), "little") for i in range(8)]
    assert pub[0] == 1 and pub[5] == int.from_bytes(inst["anchor"], "little")
    nf_bits = [(nf[i // 8] >> (i % 8)) & 1 for i in range(256)]
    assert pub[6] == sum(b << i for i, b in enumerate(nf_bits[:254])) and pub[7] == sum(b << i for i, b in enumerate(nf_bits[254:]))
    assert H.point_bytes(pub[1], pub[2]) == rk and H.point_bytes(pub[3], pub[4]) == cv
    assert H.value_commitment(inst["asset_identifier"], inst["value"], inst["rcv"])[0] == cv
    assert H.note_cmu(inst["asset_identifier"], inst["value"], inst["diversifier"], pk_d, inst["rcm"]) == cmu
    # a wrong anchor with a non-zero value is unsatisfiable ...
    bad = dict(inst, anchor=(int.from_bytes(inst["anchor"], "little") + 1) % R)
    with pytest.raises(H.HostError) as e:
        H.spend_assignment(check=True, **bad)
    assert e.value.code == 4
    # ... but accepted for value 0 (dummy spends: "(cur - rt) * value = 0", circuit/sapling.rs:366-374)
    inst0, _, _ = spend_instance(101, value=0)
    H.spend_assignment(check=True, **dict(inst0, anchor=12345))
    # invalid diversifier -> the reference returns Err(()) (sapling/prover.rs:84)
    n_invalid = 0
    for k in range(40):
        try:
            H.spend_assignment(check=False, **dict(inst, diversifier=bytes([k]) * 11))
        except H.HostError as e:
            assert e.code == 2
            n_invalid += 1
    assert 0 < n_invalid < 40


def test_output_and_convert_witnesses_satisfy_their_circuits():
    rng = random.Random(9)
    ident = H.asset_identifier(b"asset 0")
    # Output
    while True:
        d = bytes(rng.getrandbits(8) for _ in range(11))
        pk_d = H.jubjub_mul(H.point_bytes(*H.generator_uv(0)), _rand_scalar(rng))     # any prime-order point
        try:
            inputs, aux, cv = H.output_assignment(esk=_rand_scalar(rng), diversifier=d, pk_d=pk_d, rcm=_rand_scalar(rng),
                                                  asset_identifier=ident, value=rng.randrange(1 << 50), rcv=_rand_scalar(rng), check=True)
            break
        except H.HostError as e:
            assert e.code == 2
    cs, _ = H.circuit("output")
    assert O.r1cs_unsatisfied(cs, inputs, aux) == 0
    pub = [int.from_bytes(inputs[i].tobytes(), "little") for i in range(6)]
    assert H.point_bytes(pub[1], pub[2]) == cv                # inputs: cv.u, cv.v, epk.u, epk.v, cmu (verifier.rs:151-164)
    # Convert: leaf = PedersenHash(asset generator bits), path to the anchor
    gen = H.asset_generator(ident)
    siblings = [rng.randrange(R) for _ in range(32)]
    pos = rng.getrandbits(32)
    anchor = H.merkle_root(H.convert_cmu(gen), siblings, pos)
    inputs, aux, cv = H.convert_assignment(gen, rng.randrange(1, 1 << 40), anchor, siblings, pos, _rand_scalar(rng), check=True)
    cs, _ = H.circuit("convert")
    assert O.r1cs_unsatisfied(cs, inputs, aux) == 0
    pub = [int.from_bytes(inputs[i].tobytes(), "little") for i in range(4)]
    assert H.point_bytes(pub[1], pub[2]) == cv and pub[3] == int.from_bytes(anchor, "little")
    with pytest.raises(H.HostError):
        H.convert_assignment(gen, 7, (int.from_bytes(anchor, "little") + 1) % R, siblings, pos, 5, check=True)


def test_proving_mode_assignment_equals_recording_mode():
    """The synthesizer takes shortcuts when nothing records constraints (table coefficients skipped, one inversion per
    Pedersen segment, 0 / 1 written without a Montgomery conversion): the assignment must be the same bytes."""
    inst, _, _ = spend_instance(110, value=77)
    fast, slow = H.spend_assignment(check=False, **inst), H.spend_assignment(check=True, **inst)
    assert (fast[0] == slow[0]).all() and (fast[1] == slow[1]).all() and fast[2:] == slow[2:]
    rng = random.Random(9)
    ident = H.asset_identifier(b"benchmark")
    pk = H.jubjub_mul(H.point_bytes(*H.generator_uv(0)), _rand_scalar(rng))
    while True:
        d = bytes(rng.getrandbits(8) for _ in range(11))
        try:
            args = (_rand_scalar(rng), d, pk, _rand_scalar(rng), ident, 5, _rand_scalar(rng))
            fo = H.output_assignment(*args, check=False)
            break
        except H.HostError:
            continue
    so = H.output_assignment(*args, check=True)
    assert (fo[0] == so[0]).all() and (fo[1] == so[1]).all() and fo[2] == so[2]
    gen = H.asset_generator(H.asset_identifier(b"asset 3"))
    sib = [rng.randrange(R) for _ in range(32)]
    anchor = H.merkle_root(H.convert_cmu(gen), sib, 5)
    fc = H.convert_assignment(gen, 9, anchor, sib, 5, 11, check=False)
    sc = H.convert_assignment(gen, 9, anchor, sib, 5, 11, check=True)
    assert (fc[0] == sc[0]).all() and (fc[1] == sc[1]).all() and fc[2] == sc[2]


def test_host_verifier_agrees_with_oracle_pairing_check():
    """product-side verify_proof (masp_amd/csrc/host/pairing.h) vs the oracle's independent pairing on oracle-made proofs"""
    import toy_r1cs
    cs, inputs, aux, vals = toy_r1cs.make(3, 8, 40, 300)
    pbuf = O.generate_parameters(cs, toy_r1cs.toxic(3))
    P = O.Params(pbuf)
    vk = H.PreparedVerifyingKey(pbuf)
    proof = O.create_proof(P, cs, inputs, aux, 5, 6)
    other = O.create_proof(P, cs, inputs, aux, 5, 7)
    pub = vals[1:8]
    cases = [(proof, pub), (other, pub), (proof[:144] + other[144:], pub), (proof, [pub[0] + 1] + pub[1:]),
             (bytes([proof[0] ^ 0x20]) + proof[1:], pub)]
    for pr, pi in cases:
        assert vk.verify(pr, pi) == (O.verify_proof(pbuf, pr, pi) == 1)
    assert vk.verify(proof, pub) and not vk.verify(proof, pub[:-1])
    assert not vk.verify(bytes(192), pub)          # not even valid encodings


def test_host_batch_verifier():
    """`verify_proofs_batch` semantics (sapling/verifier/batch.rs:24-31): a batch passes iff every member verifies"""
    import toy_r1cs
    cs, inputs, aux, vals = toy_r1cs.make(4, 8, 40, 300)
    pbuf = O.generate_parameters(cs, toy_r1cs.toxic(4))
    P = O.Params(pbuf)
    vk = H.PreparedVerifyingKey(pbuf)
    pub = vals[1:8]
    proofs = [O.create_proof(P, cs, inputs, aux, 10 + i, 20 + i) for i in range(5)]
    assert all(vk.verify(p, pub) for p in proofs)
    assert vk.verify_batch(proofs, [pub] * 5)
    assert vk.verify_batch(proofs[:1], [pub]) and vk.verify_batch([], [])
    assert vk.verify_batch(proofs, [pub] * 5, randomness=bytes(80))            # coefficients are forced non-zero
    bad_c = proofs[2][:144] + proofs[3][144:]                                  # C of another proof
    assert not vk.verify_batch(proofs[:2] + [bad_c] + proofs[3:], [pub] * 5)
    assert not vk.verify_batch(proofs, [pub] * 4 + [[pub[0] + 1] + pub[1:]])   # one wrong public input
    assert not vk.verify_batch(proofs, [pub] * 4 + [pub[:-1]])                 # ragged
    assert not vk.verify_batch([bytes(192)], [pub])


def test_recording_survives_handles_freed_on_another_thread():
    """the 'build linear combinations' flag used to be a thread-local counter tied to the LIFETIME of a recording
    constraint system.  A circuit handle freed on another thread than the one that made it left that thread at -1, and the next
    recording synthesis there built empty LCs: empty matrices, a different structure hash, a satisfiability check of 0 * 0 = 0.
    Now the flag is scoped to the synthesis call: set up on one thread, free on another (several times), set up again there."""
    import ctypes as C
    import threading
    L = H.load_library()
    exp = load("circuits.json")["output"]
    handles, out = [], {}

    def make():
        handles.extend(L.masp_host_circuit_setup(H.KINDS["output"]) for _ in range(3))

    def free_then_setup():
        for h in handles:
            L.masp_host_circuit_free(h)
        h = L.masp_host_circuit_setup(H.KINDS["output"])
        buf = C.create_string_buffer(65)
        L.masp_host_circuit_hash(h, buf)
        counts = (C.c_uint32 * 6)()
        L.masp_host_circuit_counts(h, counts)
        L.masp_host_circuit_free(h)
        out["hash"], out["nnz"] = buf.value.decode(), list(counts)[3:]
    for fn in (make, free_then_setup):
        t = threading.Thread(target=fn)
        t.start()
        t.join()
    assert out["hash"] == exp["hash"] and min(out["nnz"]) > 0

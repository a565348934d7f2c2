"""Golden proof fixtures (tests/golden/proofs.json, made by tests/golden/make_proof_fixtures.py): fixed toxic waste, witness
seed and (r, s) -> 192 bytes, for a toy circuit and the three real MASP circuits.  The reference holds no golden proof
(SURVEY.md §0.3), so the bytes are pinned by three independent computations instead — the oracle's create_proof, the oracle's
closed form and the pure-Python closed form — and the GPU prover must reproduce them in lone-proof and in batch mode."""
import hashlib
import json
import os

import pytest

import oracle_lib as O
import pyclosed
import toy_r1cs

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "proofs.json")))
CASES = ["toy", "spend", "output", "convert"]


def _instance(name):
    from masp_amd import host as H
    from masp_amd import workload as W
    f = FIX[name]
    if name == "toy":
        cs, inputs, aux, _ = toy_r1cs.make(f["witness_seed"], n_inputs=4, n_free=40, n_constraints=300, bool_share=0.7)
    else:
        kind, kw = W.description(name, f["witness_seed"])
        inputs, aux = W.assignment(kind, kw)
        cs = H.circuit(name)[0]
    assert (cs.n_inputs, cs.n_aux, cs.n_constraints) == (f["n_inputs"], f["n_aux"], f["n_constraints"])
    # the witness the fixture was made from (a change of the synthesizer or of the instance generator shows up HERE, not as
    # a proof mismatch)
    assert hashlib.sha256(inputs.tobytes() + aux.tobytes()).hexdigest() == f["assignment_sha256"]
    toxic = [int(t, 16) for t in f["toxic"]]
    return cs, inputs, aux, toxic, int(f["r"], 16), int(f["s"], 16), bytes.fromhex(f["proof"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_golden_proofs(name):
    cs, inputs, aux, toxic, r, s, want = _instance(name)
    assert O.closed_form_proof(cs, toxic, inputs, aux, r, s) == want
    params = O.generate_parameters(cs, toxic)
    assert hashlib.sha256(params.tobytes()).hexdigest() == FIX[name]["params_sha256"] and params.size == FIX[name]["params_bytes"]
    if name in ("toy", "output"):          # the full CPU prover (NTT + multiexp) on the two cheap cases
        assert O.create_proof(O.Params(params), cs, inputs, aux, r, s) == want
    pub = [int.from_bytes(inputs[i].tobytes(), "little") for i in range(1, cs.n_inputs)]
    assert O.verify_proof(params[:868 + 96 * cs.n_inputs], want, pub) == 1


@pytest.mark.parametrize("name", ["toy", "output"])
def test_pure_python_closed_form_reproduces_the_golden_proofs(name):
    """independent of the C++ oracle: Python big integers + affine curve arithmetic (the generator script checks all four)"""
    cs, inputs, aux, toxic, r, s, want = _instance(name)
    assert pyclosed.closed_form_proof(cs, toxic, inputs, aux, r, s) == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_the_golden_proofs(name):
    import masp_amd
    cs, inputs, aux, toxic, r, s, want = _instance(name)
    ctx = masp_amd.Context(0)
    try:
        params = ctx.generate_parameters(cs, toxic)
        assert hashlib.sha256(params.tobytes()).hexdigest() == FIX[name]["params_sha256"]
        ctx.load_circuit(0, params, cs)
        assert ctx.prove(0, inputs, aux, r, s) == want                                   # lone-proof mode
        got = ctx.prove_batch([(0, inputs, aux, r, s)] * 9 + [(0, inputs, aux, r + 1, s)])   # batch mode (np = 10)
        assert got[:9] == [want] * 9 and got[9] != want
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("c_lone", [-1, 6, 11])
def test_gpu_lone_proof_whatever_windows_its_g2_tables_use(c_lone):
    """A lone proof runs B2 on its own narrow-window tables (masp_hip_options::window_bits_b2_lone, default 8: covered by the test above).
    Without them (-1: the batch tables and B1's digit sort, as in round 1) and on other widths (6: fewer than 128 buckets, the
    single-pass sort; 11) the bytes are the same."""
    import masp_amd
    cs, inputs, aux, toxic, r, s, want = _instance("spend")
    ctx = masp_amd.Context(0, window_bits_b2_lone=c_lone)
    assert ctx.options["window_bits_b2_lone"] == max(c_lone, 0)
    try:
        ctx.load_circuit(0, ctx.generate_parameters(cs, toxic), cs)
        assert ctx.prove(0, inputs, aux, r, s) == want
        got = ctx.prove_batch([(0, inputs, aux, r, s)] * 3)      # np = 3: lone-proof mode with gridDim.y = 3
        assert got == [want] * 3
    finally:
        ctx.close()

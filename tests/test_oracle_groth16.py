"""Oracle self-consistency: create_proof (NTT + multiexp) == toxic-waste closed form
byte-for-byte, and the pairing verification equation holds.  CPU only."""
import random

import numpy as np
import pytest

import oracle_lib as O
import toy_r1cs
from pyref import R


def _poly_h_python(a, b, c, logm):
    """h = (A*B - C)/Z via python big ints: interpolate on {w^k}, multiply, divide by X^m - 1."""
    m = 1 << logm
    w = pow(pow(7, (R - 1) >> 32, R), 1 << (32 - logm), R)
    def interp(ev):
        ev = list(ev) + [0] * (m - len(ev))
        winv = pow(w, -1, R)
        minv = pow(m, -1, R)
        return [sum(ev[k] * pow(winv, j * k, R) for k in range(m)) * minv % R for j in range(m)]
    A, B, C = interp(a), interp(b), interp(c)
    prod = [0] * (2 * m)
    for i, x in enumerate(A):
        for j, y in enumerate(B):
            prod[i + j] = (prod[i + j] + x * y) % R
    for i, x in enumerate(C):
        prod[i] = (prod[i] - x) % R
    # divide by X^m - 1
    q = [0] * m
    for i in range(2 * m - 1, m - 1, -1):
        q[i - m] = prod[i]
        prod[i - m] = (prod[i - m] + prod[i]) % R
        prod[i] = 0
    assert all(x == 0 for x in prod), "not divisible: unsatisfied witness"
    return q


def test_ntt_roundtrip_and_definition():
    rng = random.Random(5)
    logm = 4
    m = 1 << logm
    vals = [rng.randrange(R) for _ in range(m)]
    data = np.stack([np.frombuffer(v.to_bytes(32, "little"), np.uint8) for v in vals])
    f = O.ntt(data, logm)
    w = pow(pow(7, (R - 1) >> 32, R), 1 << (32 - logm), R)
    for k in range(m):
        assert int.from_bytes(f[k].tobytes(), "little") == sum(vals[j] * pow(w, j * k, R) for j in range(m)) % R
    assert (O.ntt(f, logm, inverse=True) == data).all()


def test_quotient_matches_python_polynomial_division():
    cs, inputs, aux, _ = toy_r1cs.make(11, n_inputs=2, n_free=3, n_constraints=11)
    a, b, c, *_ = O.r1cs_eval(cs, inputs, aux)
    logm = 4
    assert cs.nrows <= 16
    h = O.quotient_h(a, b, c, logm)
    toint = lambda arr: [int.from_bytes(x.tobytes(), "little") for x in arr]
    expect = _poly_h_python(toint(a), toint(b), toint(c), logm)
    assert expect[-1] == 0
    assert toint(h) == expect[:-1]


@pytest.mark.parametrize("seed,n_inputs,n_free,n_constraints", [(1, 2, 4, 9), (2, 4, 10, 61), (3, 8, 40, 300)])
def test_create_proof_equals_closed_form_and_verifies(seed, n_inputs, n_free, n_constraints):
    cs, inputs, aux, vals = toy_r1cs.make(seed, n_inputs, n_free, n_constraints)
    assert O.r1cs_unsatisfied(cs, inputs, aux) == 0
    tw = toy_r1cs.toxic(seed)
    pbuf = O.generate_parameters(cs, tw)
    params = O.Params(pbuf)
    lens = params.lens()
    a, b, c, da, dbi, dba = O.r1cs_eval(cs, inputs, aux)
    m = 1 << (cs.nrows - 1).bit_length()
    # length invariants, SURVEY.md App. C
    assert lens == {"ic": n_inputs, "h": m - 1, "l": cs.n_aux, "a": n_inputs + int(da.sum()),
                    "b_g1": int(dbi.sum() + dba.sum()), "b_g2": int(dbi.sum() + dba.sum())}
    assert pbuf.size == 868 + 96 * n_inputs + 5 * 4 + 96 * (lens["h"] + lens["l"] + lens["a"] + lens["b_g1"]) + 192 * lens["b_g2"]
    rng = random.Random(seed)
    r, s = rng.randrange(R), rng.randrange(R)
    proof = O.create_proof(params, cs, inputs, aux, r, s)
    assert proof == O.closed_form_proof(cs, tw, inputs, aux, r, s)
    pub = vals[1:n_inputs]
    assert O.verify_proof(pbuf, proof, pub) == 1
    # vk prefix alone is enough to verify
    assert O.verify_proof(pbuf[:868 + 96 * n_inputs], proof, pub) == 1
    # negative cases
    if pub:
        bad = list(pub)
        bad[0] = (bad[0] + 1) % R
        assert O.verify_proof(pbuf, proof, bad) == 0
    other = O.create_proof(params, cs, inputs, aux, r, (s + 1) % R)
    assert other != proof and O.verify_proof(pbuf, other, pub) == 1
    mixed = proof[:144] + other[144:]
    assert O.verify_proof(pbuf, mixed, pub) == 0


def test_unsatisfied_witness_is_rejected_by_verifier_but_prover_still_runs():
    # bellperson does not check satisfiability (the reference bench's witness is unsatisfiable,
    # masp_proofs/benches/sapling.rs:41,69); the proof must simply fail verification.
    cs, inputs, aux, vals = toy_r1cs.make(7, 3, 6, 40)
    aux = aux.copy()
    aux[-1, 0] ^= 1
    assert O.r1cs_unsatisfied(cs, inputs, aux) > 0
    tw = toy_r1cs.toxic(7)
    pbuf = O.generate_parameters(cs, tw)
    proof = O.create_proof(O.Params(pbuf), cs, inputs, aux, 5, 6)
    assert O.verify_proof(pbuf, proof, vals[1:3]) == 0
    with pytest.raises(RuntimeError):
        O.closed_form_proof(cs, tw, inputs, aux, 5, 6)

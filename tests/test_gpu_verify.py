"""GPU Groth16 batch verification (masp_hip_verify_batch: decompression + z-multiples + one wavefront per Miller loop on the
device; public-input combination, two pairings and the final exponentiation on the host) against the host batch verifier
and the oracle's independent pairing.  Run with `-m gpu` on an MI355X."""
import random

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


def test_toy_circuit_proofs_valid_and_invalid(ctx):
    cs, inputs, aux, vals = toy_r1cs.make(81, n_inputs=4, n_free=30, n_constraints=200, bool_share=0.6)
    tw = toy_r1cs.toxic(81)
    params = ctx.generate_parameters(cs, tw)
    ctx.load_circuit(3, params, cs)
    rng = random.Random(8)
    jobs = [(3, inputs, aux, rng.randrange(R), rng.randrange(R)) for _ in range(11)]
    proofs = ctx.prove_batch(jobs)
    pub = [vals[1:4]] * 11
    gvk = ctx.prepare_verifying_key(params)
    assert all(O.verify_proof(params, p, vals[1:4]) == 1 for p in proofs[:3])          # independent pairing
    for n in (1, 2, 3, 11):
        assert gvk.verify_batch(proofs[:n], pub[:n])
    # a wrong public input, a proof from another statement, swapped halves, an undecodable point
    assert not gvk.verify_batch(proofs[:4], [vals[1:4]] * 3 + [[vals[1], vals[2], (vals[3] + 1) % R]])
    mixed = proofs[0][:48] + proofs[1][48:144] + proofs[0][144:]
    assert not gvk.verify_batch([proofs[2], mixed, proofs[3]], pub[:3])
    bad = bytearray(proofs[0])
    bad[0] &= 0x7f                                                                       # compression flag cleared
    assert not gvk.verify_batch([bytes(bad)], pub[:1])
    bad = bytearray(proofs[0])
    bad[47] ^= 1                                                                        # x not on the curve (or another point)
    assert not gvk.verify_batch([proofs[1], bytes(bad)], pub[:2])
    assert gvk.verify_batch([], [])
    gvk.close()


@pytest.mark.parametrize("kind,n", [("spend", 70), ("output", 33), ("convert", 5)])
def test_real_circuit_batches_agree_with_the_host_verifier(ctx, kind, n):
    from masp_amd import host as H
    from masp_amd import workload as W
    from masp_amd.synthetic import toxic_waste
    cs = H.circuit(kind)[0]
    params = ctx.generate_parameters(cs, toxic_waste(60))
    ctx.load_circuit(0, params, cs)
    insts = W.instances(kind, n, first_seed=6000)
    rng = random.Random(9)
    proofs = ctx.prove_batch([(0, i, a, rng.randrange(R), rng.randrange(R)) for i, a in insts])
    pub = [W.public_inputs(i) for i, _ in insts]
    hvk = H.PreparedVerifyingKey(params)
    gvk = ctx.prepare_verifying_key(params)
    z = bytes(rng.getrandbits(8) for _ in range(16 * n))
    assert hvk.verify_batch(proofs, pub, randomness=z) and gvk.verify_batch(proofs, pub, randomness=z)
    assert gvk.verify_batch(proofs, pub)                                               # fresh randomness
    # one proof of the batch replaced by a valid proof of ANOTHER statement: both verifiers say no
    wrong = list(proofs)
    wrong[n // 2] = proofs[0]
    assert not hvk.verify_batch(wrong, pub, randomness=z) and not gvk.verify_batch(wrong, pub, randomness=z)
    gvk.close()

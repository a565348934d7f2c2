"""A multi-device prover that loses a device mid-call (SURVEY.md §5 "a failed GPU => re-queue its shard"):
masp_hip_prove_batch deals its blocks from one queue, the most expensive first, to whichever device context is free; a context whose
call fails with a HIP error is taken out, its block goes back on the queue and the other contexts finish the list — the reference's
per-description loop fails per description, not per transaction batch
(/root/reference/masp_primitives/src/transaction/components/sapling/builder.rs:955-969).  One GPU listed three times, the fault
injected through the test hook masp_hip_ctx_inject_fault.  In a module of its own (three device contexts on one GPU: memory).
Run with `-m gpu` on an MI355X."""
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_lib as O
from pyref import R

pytestmark = pytest.mark.gpu

KINDS = ("spend", "output", "convert")


def test_configs4_survives_a_device_that_fails_and_the_queue_redistributes_its_block():
    import masp_amd
    from masp_amd import host as H
    from masp_amd import workload as W
    from masp_amd.synthetic import toxic_waste
    multi = masp_amd.Context([0, 0, 0], batch_cap=256, slots=1, bucket_tree_sub_batch=32)
    try:
        assert multi.device_count == 3
        assert multi.device_status() == ([0, 0, 0], 0)
        cs = {k: H.circuit(k)[0] for k in KINDS}
        toxic = {k: toxic_waste(170 + i) for i, k in enumerate(KINDS)}
        params = {k: multi.generate_parameters(cs[k], toxic[k]) for k in KINDS}
        for slot, k in enumerate(KINDS):
            multi.load_circuit(slot, params[k], cs[k])
        N = 4096                                             # BASELINE.json configs[4]: job j of circuit j mod 3
        kinds = [KINDS[j % 3] for j in range(N)]
        per = {k: W.instances(k, kinds.count(k), first_seed=19000) for k in KINDS}
        it = {k: iter(per[k]) for k in KINDS}
        insts = [next(it[k]) for k in kinds]
        rng = random.Random(144)
        rs = [(rng.randrange(R), rng.randrange(R)) for _ in range(N)]
        jobs = [(KINDS.index(k), i, a, r, s) for k, (i, a), (r, s) in zip(kinds, insts, rs)]
        # device context 1 proves its first block and "dies" on its second call
        multi.inject_fault(1, 2)
        proofs = multi.prove_batch(jobs)                     # ... and the call still succeeds
        assert len(proofs) == N and len(set(proofs)) == N
        status, requeued = multi.device_status()
        assert status == [0, 5, 0], status                   # MASP_HIP_E_HIP took context 1 out, the others are fine
        done = multi.device_proofs()
        # 1 366 / 1 365 / 1 365 jobs per circuit over three live contexts: blocks of <= 256 -> 6 per circuit of ~228.  Context 1 finished
        # exactly one block (a Spend block: the most expensive go first); the block it died on went back and somebody else proved it
        assert sum(done) == N and 200 <= done[1] <= 256, done
        assert 200 <= requeued <= 256, requeued
        assert min(done[0], done[2]) >= 1000, done           # the two survivors shared the rest (one GPU: they alternate)
        # every proof verifies under ITS circuit's key at ITS job's statement, and a sample is byte-equal to the closed form
        for k in KINDS:
            vk = multi.prepare_verifying_key(params[k])
            sel = [j for j in range(N) if kinds[j] == k]
            assert vk.verify_batch([proofs[j] for j in sel], [W.public_inputs(insts[j][0]) for j in sel])
            vk.close()
        sample = [j for k in KINDS for j in [j for j in range(N) if kinds[j] == k][::137]]
        with ThreadPoolExecutor(H.effective_cpus()) as ex:
            want = list(ex.map(lambda j: O.closed_form_proof(cs[kinds[j]], toxic[kinds[j]], insts[j][0], insts[j][1], *rs[j]), sample))
        assert [proofs[j] for j in sample] == want
        # the device stays out: the next call runs on the two that are left, the same bytes
        again = multi.prove_batch(jobs[:600])
        assert again == proofs[:600]
        done2 = multi.device_proofs()
        assert done2[1] == done[1] and sum(done2) == N + 600 and multi.device_status()[0] == [0, 5, 0]
        # an error of the INPUT is not a device's failure: an assignment that is not a canonical scalar is refused by the device's range
        # check (MASP_HIP_E_SCALAR_RANGE) and nobody is taken out for it
        bad_aux = np.array(insts[1][1], copy=True)
        bad_aux[5] = np.frombuffer(R.to_bytes(32, "little"), np.uint8)
        with pytest.raises(masp_amd.MaspHipError) as e:
            multi.prove_batch([jobs[0], (jobs[1][0], insts[1][0], bad_aux, 5, 6)])
        assert e.value.code == 8 and multi.device_status()[0] == [0, 5, 0]
        # ... and when NO device is left the call fails, with the device's text
        multi.inject_fault(0, 1)
        multi.inject_fault(2, 1)
        with pytest.raises(masp_amd.MaspHipError) as e:
            multi.prove_batch(jobs[:40])
        assert e.value.code == 5 and "injected fault" in str(e.value)
        assert multi.device_status()[0] == [5, 5, 5]
        with pytest.raises(masp_amd.MaspHipError) as e:
            multi.prove_batch(jobs[:3])
        assert e.value.code == 5
    finally:
        multi.close()

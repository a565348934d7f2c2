"""BASELINE.json configs[4] under `pytest -m gpu`: 4096 mixed Spend / Output / Convert jobs through ONE masp_hip_prove_batch of a
multi-device prover.  In a module of its own: it needs most of the GPU's memory (two device contexts on a one-GPU box), so no
other module's contexts may be alive next to it.  Run with `-m gpu` on an MI355X."""
import random
from concurrent.futures import ThreadPoolExecutor

import pytest

import oracle_lib as O
from pyref import R

pytestmark = pytest.mark.gpu

KINDS = ("spend", "output", "convert")


def _rs(rng, n):
    return [(rng.randrange(R), rng.randrange(R)) for _ in range(n)]


def test_configs4_4096_mixed_jobs_sharded_over_the_devices_of_the_node():
    """BASELINE.json configs[4]: a batch of 4096 mixed Spend / Output / Convert proofs (job j of circuit j mod 3, SURVEY.md §8d
    config 5) through ONE masp_hip_prove_batch of a multi-device prover (masp_hip_ctx_create_ex over every GPU this process
    sees; on a one-GPU box the same GPU twice, so that the dealing of batches to devices, the per-device host threads and the
    reassembly in job order run either way).  Every proof through the GPU batch verifier, 32 per circuit byte-equal to the
    toxic-waste closed form (+ one per circuit to the CPU restatement), and the same jobs in a shuffled order give the
    permuted bytes.  The call sites this serves: the serial per-description loops of
    /root/reference/masp_primitives/src/transaction/components/sapling/builder.rs:935-1140."""
    import masp_amd
    from masp_amd import host as H
    from masp_amd import hip
    from masp_amd import workload as W
    from masp_amd.synthetic import toxic_waste
    n_dev = hip.device_count()
    assert n_dev >= 1
    devices = list(range(n_dev)) if n_dev > 1 else [0, 0]
    # the same GPU twice: both contexts' scratch has to fit one HBM (two batches in flight each, smaller tree sub-batches)
    opts = dict(batch_cap=256) if n_dev > 1 else dict(batch_cap=256, slots=2, bucket_tree_sub_batch=32)
    multi = masp_amd.Context(devices, **opts)
    try:
        assert multi.device_count == len(devices)
        cs = {k: H.circuit(k)[0] for k in KINDS}
        toxic = {k: toxic_waste(70 + i) for i, k in enumerate(KINDS)}
        params = {k: multi.generate_parameters(cs[k], toxic[k]) for k in KINDS}
        for slot, k in enumerate(KINDS):
            multi.load_circuit(slot, params[k], cs[k])
        N = 4096
        kinds = [KINDS[j % 3] for j in range(N)]
        per = {k: W.instances(k, kinds.count(k), first_seed=9000) for k in KINDS}
        it = {k: iter(per[k]) for k in KINDS}
        insts = [next(it[k]) for k in kinds]
        rs = _rs(random.Random(44), N)
        jobs = [(KINDS.index(k), i, a, r, s) for k, (i, a), (r, s) in zip(kinds, insts, rs)]
        assert multi.prove_batch([]) == []                  # an empty list through the multi-device front: no block, no worker waits for one
        before = multi.device_proofs()
        proofs = multi.prove_batch(jobs)
        assert len(proofs) == N and all(len(p) == 192 for p in proofs) and len(set(proofs)) == N
        # every device context took its share of the queue (round 6: blocks are TAKEN by whichever context is free, the most expensive
        # first, not dealt round-robin) — a prover that silently ran everything on its first device would show here
        done = [a - b for a, b in zip(multi.device_proofs(), before)]
        assert len(done) == len(devices) and sum(done) == N, done
        assert min(done) >= N // len(devices) // 2, done
        assert multi.device_status() == ([0] * len(devices), 0)
        # every proof verifies under ITS circuit's key at ITS job's statement (a proof at the wrong position would not)
        for k in KINDS:
            vk = multi.prepare_verifying_key(params[k])
            sel = [j for j in range(N) if kinds[j] == k]
            assert vk.verify_batch([proofs[j] for j in sel], [W.public_inputs(insts[j][0]) for j in sel])
            swapped = [proofs[sel[1]], proofs[sel[0]]] + [proofs[j] for j in sel[2:64]]
            assert not vk.verify_batch(swapped, [W.public_inputs(insts[j][0]) for j in sel[:64]])
            vk.close()
        # 32 per circuit (spread over the list: first, middle and last batches of every device) == closed form
        sample = []
        for k in KINDS:
            sel = [j for j in range(N) if kinds[j] == k]
            sample += sel[:11] + sel[len(sel) // 2:len(sel) // 2 + 10] + sel[-11:]
        with ThreadPoolExecutor(H.effective_cpus()) as ex:
            want = list(ex.map(lambda j: O.closed_form_proof(cs[kinds[j]], toxic[kinds[j]], insts[j][0], insts[j][1], *rs[j]), sample))
        bad = [j for j, w in zip(sample, want) if proofs[j] != w]
        assert not bad, "proofs %s differ from the closed form" % bad[:10]
        for k in KINDS:
            j = kinds.index(k)
            assert proofs[j] == O.create_proof(O.Params(params[k]), cs[k], insts[j][0], insts[j][1], *rs[j])
        # the same jobs in another order: the same bytes, permuted
        perm = list(range(N))
        random.Random(45).shuffle(perm)
        again = multi.prove_batch([jobs[j] for j in perm])
        assert again == [proofs[j] for j in perm]
    finally:
        multi.close()

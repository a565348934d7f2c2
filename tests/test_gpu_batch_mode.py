"""Parity of the code path `bench.py` times: batch mode (np >= 8 proofs per launch sequence: 16-/12-bit windows, 2^18/np
chunks, the 33 000-entry bucket-0 heavy path, G = 16 .. 64 weighted sums, h + l as one MSM, six-transform quotient) on the REAL
circuits at real size, through the C ABI, at 85 / 86 and at 256 proofs per launch sequence.

Oracles: the toxic-waste closed form for every proof (no NTT / MSM involved: one QAP evaluation + fixed-base
multiplications), the CPU restatement `create_proof` for a sample, and the pairing equation (host batch verifier,
cross-checked against the oracle's independent pairing in tests/test_circuits.py).  Instances follow the reference's
benches (masp_amd/workload.py: benches/sapling.rs:38-86, benches/convert.rs:31-66) — all distinct.
Run with `-m gpu` on an MI355X."""
import os
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_lib as O
from pyref import R

pytestmark = pytest.mark.gpu

KINDS = ("spend", "output", "convert")


class Rig:
    """One context with real circuits on a synthetic CRS of known toxic waste.  `cap` = masp_hip_options::batch_cap, the proofs
    per launch sequence: 96 cuts 256 Spends into 86 + 85 + 85, 256 (the default, what bench.py runs) runs them as one."""

    def __init__(self, device=0, kinds=KINDS, cap=96):
        import masp_amd
        from masp_amd import host as H
        from masp_amd.synthetic import toxic_waste
        self.ctx = masp_amd.Context(device, batch_cap=cap)
        assert self.ctx.options["batch_cap"] == cap and self.ctx.options["slots"] == 4
        self.cs = {k: H.circuit(k)[0] for k in kinds}
        self.toxic = {k: toxic_waste(40 + KINDS.index(k)) for k in kinds}
        self.params = {}
        self.vk = {}
        for k in kinds:
            self.params[k] = self.ctx.generate_parameters(self.cs[k], self.toxic[k])
            self.ctx.load_circuit(KINDS.index(k), self.params[k], self.cs[k])
            self.vk[k] = H.PreparedVerifyingKey(self.params[k])
        self.threads = H.effective_cpus()

    def close(self):
        self.ctx.close()


@pytest.fixture(scope="module")
def rig():
    r = Rig()
    yield r
    r.close()


def _rs(rng, n):
    return [(rng.randrange(R), rng.randrange(R)) for _ in range(n)]


def _check(rig, kinds, insts, rs, proofs, n_cpu):
    """every proof == closed form; the first n_cpu of each circuit == CPU restatement; all pass the batched pairing check"""
    from masp_amd import workload as W
    assert len(proofs) == len(insts) and all(len(p) == 192 for p in proofs) and len(set(proofs)) == len(proofs)

    def closed(j):
        k = kinds[j]
        return O.closed_form_proof(rig.cs[k], rig.toxic[k], insts[j][0], insts[j][1], *rs[j])
    with ThreadPoolExecutor(rig.threads) as ex:
        expect = list(ex.map(closed, range(len(insts))))
    bad = [j for j in range(len(insts)) if proofs[j] != expect[j]]
    assert not bad, "proofs %s differ from the closed form" % bad[:10]
    seen = {}
    for j, k in enumerate(kinds):
        if seen.get(k, 0) < n_cpu:
            seen[k] = seen.get(k, 0) + 1
            assert proofs[j] == O.create_proof(O.Params(rig.params[k]), rig.cs[k], insts[j][0], insts[j][1], *rs[j])
    for k in set(kinds):
        sel = [j for j in range(len(insts)) if kinds[j] == k]
        assert rig.vk[k].verify_batch([proofs[j] for j in sel], [W.public_inputs(insts[j][0]) for j in sel])
    # and the batch verifier does notice a wrong proof
    k0 = kinds[0]
    sel = [j for j in range(len(insts)) if kinds[j] == k0][:8]
    if len(sel) >= 2:
        swapped = [proofs[sel[1]], proofs[sel[0]]] + [proofs[j] for j in sel[2:]]
        assert not rig.vk[k0].verify_batch(swapped, [W.public_inputs(insts[j][0]) for j in sel])


def test_256_distinct_spends_through_one_call(rig):
    """BASELINE.json configs[3]: a batch of 256 distinct Spend proofs on one GPU through ONE masp_hip_prove_batch
    (-> groups of 86 / 85 / 85 proofs, batch mode)."""
    from masp_amd import workload as W
    insts = W.instances("spend", 256, first_seed=1000)
    assert len(set(a.tobytes() for _, a in insts)) == 256
    rs = _rs(random.Random(1), 256)
    proofs = rig.ctx.prove_batch([(0, i, a, r, s) for (i, a), (r, s) in zip(insts, rs)])
    _check(rig, ["spend"] * 256, insts, rs, proofs, n_cpu=8)
    # the device-resident path bench.py times runs the same batches: same bytes
    h, n = rig.ctx.batch_upload([(0, i, a, r, s) for (i, a), (r, s) in zip(insts, rs)])
    again, ms = rig.ctx.batch_prove_resident(h, n)
    rig.ctx.batch_free(h)
    assert again == proofs and ms > 0


def test_256_distinct_spends_as_one_launch_sequence(rig):
    """The configuration bench.py times since round 2: batch_cap = 256 (the default), all 256 Spends in ONE launch sequence
    (gridDim.y = 256: its own chunk geometry, 64 buckets per lane in the weighted sum of h + l).  Same toxic waste as the
    96-proof rig, so besides the closed form the bytes must equal what the 86 / 85 / 85 split produces."""
    from masp_amd import workload as W
    insts = W.instances("spend", 256, first_seed=1000)
    rs = _rs(random.Random(1), 256)
    jobs = [(0, i, a, r, s) for (i, a), (r, s) in zip(insts, rs)]
    big = Rig(kinds=("spend",), cap=256)
    try:
        proofs = big.ctx.prove_batch(jobs)
        _check(big, ["spend"] * 256, insts, rs, proofs, n_cpu=2)
        h, n = big.ctx.batch_upload(jobs)
        again, ms = big.ctx.batch_prove_resident(h, n)
        big.ctx.batch_free(h)
        assert again == proofs and ms > 0
    finally:
        big.close()
    assert proofs == rig.ctx.prove_batch(jobs)


def test_lone_and_batch_mode_meet_at_eight_proofs(rig):
    """np < 8 runs in lone-proof mode (MSMs side by side on their own streams, h and l apart, B2 on its narrow windows, the
    assembly's pieces spread over the streams), np >= 8 as a batch: 9 distinct Spends as one call of 9, as 8 + 1 and as 7 + 2
    give the same bytes, all equal to the closed form."""
    from masp_amd import workload as W
    insts = W.instances("spend", 9, first_seed=7000)
    rs = _rs(random.Random(7), 9)
    jobs = [(0, i, a, r, s) for (i, a), (r, s) in zip(insts, rs)]
    nine = rig.ctx.prove_batch(jobs)
    _check(rig, ["spend"] * 9, insts, rs, nine, n_cpu=1)
    assert rig.ctx.prove_batch(jobs[:8]) + rig.ctx.prove_batch(jobs[8:]) == nine
    assert rig.ctx.prove_batch(jobs[:7]) + rig.ctx.prove_batch(jobs[7:]) == nine


@pytest.mark.parametrize("kind", ["output", "convert"])
def test_64_distinct_proofs_of_the_smaller_circuits(rig, kind):
    from masp_amd import workload as W
    insts = W.instances(kind, 64, first_seed=2000)
    rs = _rs(random.Random(2), 64)
    slot = KINDS.index(kind)
    proofs = rig.ctx.prove_batch([(slot, i, a, r, s) for (i, a), (r, s) in zip(insts, rs)])
    _check(rig, [kind] * 64, insts, rs, proofs, n_cpu=4)


def test_mixed_three_circuit_list_in_job_order(rig):
    """BASELINE.json configs[4] job mix (job j of circuit j mod 3), 32 of each, interleaved in one call: every circuit's
    jobs form their own batch (np = 32) and every proof comes back at its job's position."""
    from masp_amd import workload as W
    per = {k: W.instances(k, 32, first_seed=3000) for k in KINDS}
    kinds = [KINDS[j % 3] for j in range(96)]
    insts = [per[k][j // 3] for j, k in enumerate(kinds)]
    rs = _rs(random.Random(3), 96)
    proofs = rig.ctx.prove_batch([(KINDS.index(k), i, a, r, s) for k, (i, a), (r, s) in zip(kinds, insts, rs)])
    _check(rig, kinds, insts, rs, proofs, n_cpu=2)


def _le(x):
    return np.frombuffer((x % R).to_bytes(32, "little"), np.uint8)


@pytest.mark.parametrize("n,window_bits,shape", [(131071, 16, "uniform"), (100497, 12, "witness")])
def test_msm_engine_batched_np16_full_size(rig, n, window_bits, shape):
    """The MSM engine alone in batch mode: 16 scalar vectors over one base set in one launch sequence (gridDim.y = 16), at
    the Spend sizes and window widths the prover uses (h: 131 071 uniform scalars, 16-bit windows; l: 100 497
    witness-shaped scalars — 38 % zeros, 33 % ones, the rest full width — 12-bit windows),
    checked by the discrete-log identity: with bases P_i = k_i G,  sum_i s_i P_i = (sum_i s_i k_i mod r) G."""
    rng = random.Random(n)
    np_ = 16
    k_int = [rng.randrange(R) for _ in range(n)]
    ks = np.frombuffer(b"".join(k.to_bytes(32, "little") for k in k_int), np.uint8).reshape(n, 32)
    bases = O.g1_mul_gen_many(ks)
    sc = np.zeros((np_, n, 32), np.uint8)
    totals = []
    for p in range(np_):
        if shape == "uniform":
            s_int = [rng.randrange(R) for _ in range(n)]
            for i, e in enumerate([0, 1, R - 1, (1 << 255) % R, 0x8000, 0x8001, 0xffff, 0x10000][:8]):
                s_int[(p * 8 + i) % n] = e
        else:
            s_int = [0 if u < 0.38 else 1 if u < 0.71 else rng.randrange(R) for u in (rng.random() for _ in range(n))]
        sc[p] = np.frombuffer(b"".join(s.to_bytes(32, "little") for s in s_int), np.uint8).reshape(n, 32)
        totals.append(sum(a * b for a, b in zip(k_int, s_int)) % R)
    got = rig.ctx.msm_g1_multi(bases, sc, window_bits=window_bits)
    want = O.g1_mul_gen_many(np.stack([_le(t) for t in totals]))
    assert got == [want[p].tobytes() for p in range(np_)]
    # np = 1 of the same data goes through lone-proof mode, and the generic entry point picks its own window: same point
    assert rig.ctx.msm_g1_multi(bases, sc[:1], window_bits=window_bits)[0] == got[0]
    assert rig.ctx.msm_g1(bases, sc[0]) == got[0]


def test_job_struct_reserved_field_must_be_zero(rig):
    """masp_hip_job carries no size field: `reserved` has to be 0 so that a later revision of the struct can give it a meaning
    (ADVICE r03).  A job with a non-zero value is refused before anything is enqueued."""
    import masp_amd
    from masp_amd import workload as W
    (i, a), = W.instances("output", 1, first_seed=77)
    arr, n, keep = rig.ctx.marshal_jobs([(KINDS.index("output"), i, a, 5, 6)])
    assert len(rig.ctx.prove_marshalled(arr, n)) == 1
    arr[0].reserved = 1
    with pytest.raises(masp_amd.MaspHipError) as e:
        rig.ctx.prove_marshalled(arr, n)
    assert e.value.code == 1        # MASP_HIP_E_INVALID_ARG


def test_multi_device_context_deals_batches_to_its_devices(rig):
    """masp_hip_ctx_create_multi: one prover over several device contexts (here the same GPU twice — the sharding, the
    per-device host threads and the reassembly are what is tested; with 8 GPUs the list is 0..7).  Same bytes as the
    single-device context, job order preserved, mixed circuits."""
    import masp_amd
    from masp_amd import workload as W
    multi = masp_amd.Context([0, 0], batch_cap=96)
    assert multi.device_count == 2 and rig.ctx.device_count == 1
    for slot, k in enumerate(KINDS[:2]):
        multi.load_circuit(slot, rig.params[k], rig.cs[k])
    sp, ou = W.instances("spend", 40, first_seed=4000), W.instances("output", 24, first_seed=4000)
    kinds = ["spend" if j % 8 < 5 else "output" for j in range(64)]
    it = {"spend": iter(sp), "output": iter(ou)}
    insts = [next(it[k]) for k in kinds]
    rs = _rs(random.Random(4), 64)
    jobs = [(KINDS.index(k), i, a, r, s) for k, (i, a), (r, s) in zip(kinds, insts, rs)]
    got = multi.prove_batch(jobs)
    assert got == rig.ctx.prove_batch(jobs)
    _check(rig, kinds, insts, rs, got, n_cpu=1)
    with pytest.raises(masp_amd.MaspHipError) as e:
        multi.prove_batch([(2, insts[0][0], insts[0][1], 1, 2)])        # Convert was not loaded on this prover
    assert e.value.code == 7
    multi.close()



def test_workspaces_growing_under_load_and_memory_returned(rig):
    """A fresh context whose slots first see small Output batches and then, from three host threads at once, Spend batches: every
    slot's workspaces (34 GB of tree scratch per slot at the default sub-batch; less here) grow while the other slots are busy.
    The buffers they outgrow are not freed on the spot — `hipFree` waits for every stream of the device, which stalled a growing
    slot's host thread for seconds behind the other slots' work (profiles/r04e_mixed_calls_timing.txt) — but put on a list that is
    emptied when the context goes idle or is destroyed (csrc/util.h, dev_free).  Checked: the proofs (against the shared rig's
    context, closed form and pairing) and that destroying the context returns the device's memory."""
    import ctypes as C
    import masp_amd
    from masp_amd import workload as W
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipSetDevice(0) == 0 and hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value
    ou, sp = W.instances("output", 24, first_seed=5100), W.instances("spend", 72, first_seed=5100, threads=rig.threads)
    rs_o, rs_s = _rs(random.Random(51), 24), _rs(random.Random(52), 72)
    jobs_o = [(1, i, a, r, s) for (i, a), (r, s) in zip(ou, rs_o)]
    jobs_s = [(0, i, a, r, s) for (i, a), (r, s) in zip(sp, rs_s)]
    want_o, want_s = rig.ctx.prove_batch(jobs_o), rig.ctx.prove_batch(jobs_s)   # (the shared context's own workspaces grow here: before free0)
    free0 = free_bytes()
    ctx = masp_amd.Context(0, batch_cap=96)
    for slot, k in enumerate(KINDS[:2]):
        ctx.load_circuit(slot, rig.params[k], rig.cs[k])
    with ThreadPoolExecutor(3) as ex:                       # every slot meets Outputs first ...
        small = list(ex.map(lambda _: ctx.prove_batch(jobs_o), range(3)))
    with ThreadPoolExecutor(3) as ex:                       # ... then Spends, three calls in flight: the slots grow side by side
        big = list(ex.map(lambda k: ctx.prove_batch(jobs_s[24 * k:24 * k + 24] + jobs_o[8 * k:8 * k + 8]), range(3)))
    assert small[0] == small[1] == small[2] == want_o
    got_s = [p for k in range(3) for p in big[k][:24]]
    assert got_s == want_s
    for k in range(3):
        assert big[k][24:] == small[0][8 * k:8 * k + 8]
    _check(rig, ["spend"] * 72, sp, rs_s, got_s, n_cpu=1)
    ctx.close()
    free1 = free_bytes()
    # (a slot's scratch is tens of GB; the runtime's own caches — code objects of kernels first used in between — are a few hundred MB)
    assert free0 - free1 < (1 << 30), "a destroyed context kept %.1f GB of the device" % ((free0 - free1) / 2 ** 30)


def test_local_tx_prover_on_an_existing_context(rig):
    """LocalTxProver(context=...): the prover loads its circuits into the process's context, proves through it (prove_batch with
    self-verification) and leaves it open when closed — what bench.py's end_to_end region does."""
    from masp_amd import workload as W
    from masp_amd.prover import LocalTxProver
    prover = LocalTxProver(rig.params["spend"], rig.params["output"], rig.params["convert"], expected=None, context=rig.ctx)
    assert prover._ctx is rig.ctx
    descs = [W.description("spend", 9000 + k) for k in range(20)] + [W.description("output", 9000 + k) for k in range(12)]
    res = prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=rig.threads, chunk=16)
    assert len(res) == 32 and all(len(r[0]) == 192 for r in res) and len(set(r[0] for r in res)) == 32
    for (kind, kw), r in zip(descs[20:], res[20:]):                      # Outputs are not self-checked by the prover: check them here
        assert kind == "output" and len(r) == 2
    prover.close()
    (i, a), = W.instances("output", 1, first_seed=78)                    # the context is still there
    assert len(rig.ctx.prove_batch([(1, i, a, 5, 6)])) == 1

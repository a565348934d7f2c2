"""Round 5's two host-visible additions around the lone proof and the cold start (run with `-m gpu`):
  * masp_hip_profile_read_lone — where the chains of a lone proof end, by HIP events (no profiler): marks are recorded only with profiling on,
    ordered the way the streams are (an MSM before the multiplication behind it, everything before `complete`), and a lone proof's bytes are
    the same with and without them;
  * LocalTxProver.warm_up — the page-locked pool and every slot's scratch sized at load time with a witness of zeros: the first real call
    afterwards allocates nothing more and proves the same bytes a cold prover does."""
import random

import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def test_lone_chain_marks_and_bytes():
    import masp_amd
    from masp_amd import host as H, synthetic, workload as W
    ctx = masp_amd.Context(0)
    cs = H.circuit("output")[0]
    tw = synthetic.toxic_waste(2)
    params = ctx.generate_parameters(cs, tw)
    ctx.load_circuit(1, params, cs)
    (inputs, aux), = W.instances("output", 1, first_seed=11)
    with pytest.raises(masp_amd.MaspHipError):
        ctx.profile_read_lone()                               # nothing recorded yet
    plain = ctx.prove(1, inputs, aux, 1234, 5678)
    assert plain == O.closed_form_proof(cs, tw, inputs, aux, 1234, 5678)
    ctx.profile_enable(True)
    assert ctx.prove(1, inputs, aux, 1234, 5678) == plain
    m = ctx.profile_read_lone()
    ctx.profile_enable(False)
    assert set(m) == set(masp_amd.Context.LONE_MARKS) and m["start"] == 0.0
    assert all(v > 0 for k, v in m.items() if k != "start"), m
    assert m["msm_a"] < m["s_A"] and m["msm_b1"] < m["r_B1"] and m["msm_b2"] < m["g_b"] and m["quotient"] < m["msm_h"] < m["g_a_g_c"]
    assert max(m.values()) == m["complete"] and m["complete"] < 50.0    # milliseconds of GPU time
    ctx.close()


def test_warm_up_sizes_pool_and_slots_then_the_first_call_is_like_any_other():
    from masp_amd import prover as P
    from masp_amd import workload as W
    rng = random.Random(8)
    descs = [W.description("spend", 900 + k) for k in range(24)] + [W.description("output", 900 + k) for k in range(8)]
    rs = [(rng.randrange(R), rng.randrange(R)) for _ in descs]
    cold = P.LocalTxProver.with_synthetic_parameters(seed=3)
    want = cold.prove_batch(cold.new_sapling_proving_context(), descs, rs=rs, threads=4)
    cold.close()
    warm = P.LocalTxProver.with_synthetic_parameters(seed=3)
    warm.warm_up(spends=24, outputs=8, threads=4, background=True)          # on a thread: the call below waits for it
    got = warm.prove_batch(warm.new_sapling_proving_context(), descs, rs=rs, threads=4)
    assert got == want
    pooled = {k: len(v) for k, v in warm._pool.items()}
    assert pooled[P.SPEND] >= 24 and pooled[P.OUTPUT] >= 8 and pooled[P.CONVERT] == 0     # reserved by warm_up, all returned
    warm.prove_batch(warm.new_sapling_proving_context(), descs, rs=rs, threads=4)
    assert {k: len(v) for k, v in warm._pool.items()} == pooled                            # the call needed no buffer beyond them
    opt = warm._ctx.current_options()
    assert opt["bucket_tree_fallback_proofs"] == 0 and opt["hw_queues"] >= 15
    warm.close()


def test_six_lone_proofs_at_once_share_side_streams_and_keep_their_bytes():
    """Round 6: slots 0 and 1 have side streams of their own, slots 2 and 3 use slot 1's (a default context so has 15 streams, one hardware
    queue each).  Six host threads prove lone Output proofs of the real circuit at the same time — four in flight, two of them interleaving
    with slot 1 on one set of side streams, each behind its own events —: every proof is the closed form's."""
    from concurrent.futures import ThreadPoolExecutor
    import masp_amd
    from masp_amd import host as H, synthetic, workload as W
    ctx = masp_amd.Context(0)
    try:
        assert ctx.options["slots"] == 4 and ctx.stream_concurrency()[0] == 15
        cs = H.circuit("output")[0]
        tw = synthetic.toxic_waste(3)
        ctx.load_circuit(1, ctx.generate_parameters(cs, tw), cs)
        insts = W.instances("output", 6, first_seed=21)

        def caller(t):
            inputs, aux = insts[t]
            return [ctx.prove(1, inputs, aux, 1000 * t + k + 1, 7 + 1000 * t + k) for k in range(12)]
        with ThreadPoolExecutor(6) as ex:
            got = list(ex.map(caller, range(6)))
        for t in range(6):
            inputs, aux = insts[t]
            for k in (0, 5, 11):
                assert got[t][k] == O.closed_form_proof(cs, tw, inputs, aux, 1000 * t + k + 1, 7 + 1000 * t + k), (t, k)
            assert len(set(got[t])) == 12
        vk = ctx.prepare_verifying_key(ctx.generate_parameters(cs, tw))
        assert vk.verify_batch([p for t in range(6) for p in got[t]], [W.public_inputs(insts[t][0]) for t in range(6) for _ in range(12)])
        vk.close()
    finally:
        ctx.close()

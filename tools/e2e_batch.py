"""End-to-end `LocalTxProver.prove_batch` over N real Spend descriptions (BASELINE.json configs[3]: batch of 256 Spends on
one GPU): host synthesis (C++ threads) -> GPU batch -> host self-verification, each stage timed.  Unlike bench.py (inputs
resident in HBM), this includes witness generation, the H2D copies and the pairing checks.

    python tools/e2e_batch.py [N=256] [threads=os.cpu_count()] [chunk=MASP_HIP_BATCH]
"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from bench import options_from_env              # noqa: E402  (MASP_HIP_* variables -> masp_hip_options; the library reads none)

from masp_amd import host as H                     # noqa: E402
from masp_amd import workload as W                 # noqa: E402
from masp_amd.prover import LocalTxProver, _int    # noqa: E402


def spend_description(seed):
    return W.description("spend", seed)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    threads = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else H.effective_cpus()
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else None
    t = time.time()
    prover = LocalTxProver.with_synthetic_parameters(seed=7, options=options_from_env())
    print("parameters generated + loaded: %.1f s" % (time.time() - t))
    with ThreadPoolExecutor(threads) as ex:
        descs = list(ex.map(spend_description, range(n)))
        # warm-up (workspace allocation, first-launch costs)
        # (five full batches: every slot's workspace gets its final size before anything is timed)
        prover.prove_batch(prover.new_sapling_proving_context(), descs[:min(n, 5 * prover._ctx.options["batch_cap"])], threads=threads)
        stage_n = min(n, 256)                     # the staged measurement holds every aux buffer at once: bound it
        t0 = time.time()
        jobs = list(ex.map(lambda d: prover.prepare_spend(**d[1]), descs[:stage_n]))
        t1 = time.time()
        proofs = prover.prove_prepared(jobs)
        t2 = time.time()

        def check(args):
            (kind, kw), job, zk = args
            pi = list(H.point_uv(job["rk"])) + list(H.point_uv(job["cv"])) + [_int(kw["anchor"])] + H.multipack(job["nf"])
            return prover.spend_vk.verify(zk, pi)
        ok = all(ex.map(check, zip(descs, jobs, proofs)))
        t3 = time.time()
        prover._aux_give(jobs)
    assert ok
    ctx = prover.new_sapling_proving_context()
    t4 = time.time()
    out = prover.prove_batch(ctx, descs, threads=threads, chunk=chunk)
    t5 = time.time()
    assert len(out) == n
    print("N = %d Spend descriptions, %d host threads; stage by stage on the first %d:" % (n, threads, stage_n))
    print("  synthesis  %8.1f ms  (%.2f ms/proof wall, %.1f proofs/s)" % ((t1 - t0) * 1e3, (t1 - t0) * 1e3 / stage_n, stage_n / (t1 - t0)))
    print("  GPU batch  %8.1f ms  (%.2f ms/proof, %.1f proofs/s; includes H2D of witnesses, D2H of proofs)" % ((t2 - t1) * 1e3, (t2 - t1) * 1e3 / stage_n, stage_n / (t2 - t1)))
    print("  verify     %8.1f ms  (%.2f ms/proof wall)" % ((t3 - t2) * 1e3, (t3 - t2) * 1e3 / stage_n))
    print("  prove_batch end to end (chunks of %s descriptions) %8.1f ms = %.1f proofs/s" % (chunk or "n/8 within 64..256 =", (t5 - t4) * 1e3, n / (t5 - t4)))


if __name__ == "__main__":
    main()

"""LocalTxProver.prove_batch end to end (descriptions -> witnesses -> proofs -> GPU self-verification) over chunk sizes and host
thread counts, one warm prover.  usage: python tools/e2e_sweep.py [n=1024]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from bench import options_from_env              # noqa: E402
from masp_amd import host as H                     # noqa: E402
from masp_amd import workload as W                 # noqa: E402
from masp_amd.prover import LocalTxProver          # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    cpus = H.effective_cpus()
    prover = LocalTxProver.with_synthetic_parameters(seed=7, options=options_from_env())
    with ThreadPoolExecutor(cpus) as ex:
        descs = list(ex.map(lambda k: W.description("spend", k), range(2 * n)))
    prover.prove_batch(prover.new_sapling_proving_context(), descs[:1280], threads=cpus)      # sizes every slot's scratch and the aux pool
    print("cpus %d" % cpus)
    for m in (n, 2 * n):
        for threads in (cpus, 8, 4):
            for chunk in (64, 128, 256):
                best = 0.0
                for _ in range(2):
                    t0 = time.perf_counter()
                    out = prover.prove_batch(prover.new_sapling_proving_context(), descs[:m], threads=threads, chunk=chunk)
                    best = max(best, m / (time.perf_counter() - t0))
                    assert len(out) == m
                print("n %5d  threads %2d  chunk %3d: %7.1f proofs/s" % (m, threads, chunk, best), flush=True)
    prover.close()


if __name__ == "__main__":
    main()

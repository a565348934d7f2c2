"""Where does LocalTxProver.prove_batch lose time against the bare GPU rate?  Same 2048 Spend descriptions: end to end, without the
self-verification, and the pre-synthesised jobs alone (prove_prepared in chunks, 4 in flight); then a cProfile of one end-to-end call."""
import cProfile
import os
import pstats
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from bench import options_from_env              # noqa: E402
from masp_amd import host as H                     # noqa: E402
from masp_amd import workload as W                 # noqa: E402
from masp_amd.prover import LocalTxProver          # noqa: E402

n = 2048
cpus = H.effective_cpus()
prover = LocalTxProver.with_synthetic_parameters(seed=7, options=options_from_env())
with ThreadPoolExecutor(cpus) as ex:
    descs = list(ex.map(lambda k: W.description("spend", k), range(n)))
prover.prove_batch(prover.new_sapling_proving_context(), descs[:1280], threads=cpus)


def e2e(label, **kw):
    best = 0
    for _ in range(2):
        t0 = time.perf_counter()
        prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=cpus, **kw)
        best = max(best, n / (time.perf_counter() - t0))
    print("%-60s %7.1f proofs/s" % (label, best), flush=True)


e2e("end to end, chunk 256", chunk=256)
prover._self_verify = False
e2e("end to end without self-verification, chunk 256", chunk=256)
prover._self_verify = True
# the jobs alone
jobs = []
for lo in range(0, n, 16):
    jobs += prover.prepare_group("spend", [kw for _, kw in descs[lo:lo + 16]])
for trial in range(2):
    t0 = time.perf_counter()
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(lambda lo: prover.prove_prepared(jobs[lo:lo + 256]), range(0, n, 256)))
    print("%-60s %7.1f proofs/s" % ("pre-synthesised jobs, prove_prepared x 8, 4 in flight", n / (time.perf_counter() - t0)), flush=True)
prover._aux_give(jobs)
pr = cProfile.Profile()
pr.enable()
prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=cpus, chunk=256)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
prover.close()

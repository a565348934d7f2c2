import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from bench import options_from_env
from masp_amd import host as H
from masp_amd import workload as W
from masp_amd.prover import LocalTxProver
n = 4096
cpus = H.effective_cpus()
prover = LocalTxProver.with_synthetic_parameters(seed=7, options=options_from_env())
with ThreadPoolExecutor(cpus) as ex:
    descs = list(ex.map(lambda k: W.description("spend", k), range(n)))
prover.prove_batch(prover.new_sapling_proving_context(), descs[:1536], threads=cpus, in_flight=5)
for rep in range(2):
    for fl in (3, 4, 5):
        for th in (16, 6):
            t0 = time.perf_counter()
            prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=th, in_flight=fl)
            print("in_flight %d threads %2d: %7.1f proofs/s" % (fl, th, n / (time.perf_counter() - t0)), flush=True)
prover.close()

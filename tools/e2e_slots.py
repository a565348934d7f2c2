"""LocalTxProver.prove_batch, 5 120 Spend descriptions (what bench.py's end_to_end region proves at --steps 20), warm prover: GPU calls in
flight x synthesis threads for the context's slot count (MASP_HIP_SLOTS / GPU_MAX_HW_QUEUES from the environment)."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from bench import options_from_env
from masp_amd import host as H
from masp_amd import workload as W
from masp_amd.prover import LocalTxProver
n = 5120
cpus = H.effective_cpus()
prover = LocalTxProver.with_synthetic_parameters(seed=7, options=options_from_env())
slots = prover._ctx.options["slots"]
with ThreadPoolExecutor(cpus) as ex:
    descs = list(ex.map(lambda k: W.description("spend", k), range(n)))
prover.prove_batch(prover.new_sapling_proving_context(), descs[:2048], threads=cpus, in_flight=slots + 2)
for rep in range(2):
    for fl in (slots, slots + 1, slots + 2):
        for th in (16, 8):
            t0 = time.perf_counter()
            prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=th, in_flight=fl)
            print("slots %d in_flight %d threads %2d: %7.1f proofs/s" % (slots, fl, th, n / (time.perf_counter() - t0)), flush=True)
prover.close()

"""The first prove_batch of a FRESH LocalTxProver with and without warm_up(): 2 048 Spend descriptions, each case a prover (and context) of
its own in a process of its own (argv[1] = cold | warm | background).  Prints load / warm-up / first / second call seconds."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.getcwd())
from bench import options_from_env
from masp_amd import host as H
from masp_amd import workload as W
from masp_amd.prover import LocalTxProver
mode, n = sys.argv[1], int(os.environ.get("E2E_N", "2048"))
cpus = H.effective_cpus()
with ThreadPoolExecutor(cpus) as ex:
    descs = list(ex.map(lambda k: W.description("spend", k), range(n)))
W.instances("spend", 2, first_seed=10 ** 6, threads=2)          # the synthesizer's one-time tables
t0 = time.perf_counter()
prover = LocalTxProver.with_synthetic_parameters(seed=7, options=options_from_env())
t_load = time.perf_counter() - t0
t0 = time.perf_counter()
if mode == "warm":
    prover.warm_up(spends=n, threads=cpus)
elif mode == "background":
    prover.warm_up(spends=n, threads=cpus, background=True)
t_warm = time.perf_counter() - t0
out = []
for rep in range(3):
    t0 = time.perf_counter()
    res = prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=cpus)
    out.append(time.perf_counter() - t0)
    assert len(res) == n
print("%-10s load %.2f s, warm_up %.2f s; prove_batch of %d Spends: first %.3f s (%.0f proofs/s), then %.3f, %.3f s (%.0f proofs/s): first = %.2f x warm"
      % (mode, t_load, t_warm, n, out[0], n / out[0], out[1], out[2], n / min(out[1:]), min(out[1:]) / out[0]), flush=True)
prover.close()

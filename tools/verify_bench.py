"""Groth16 batch verification: host (libmasp_host, one thread and all threads) against the GPU (masp_hip_verify_batch) on n
real Spend proofs.    python tools/verify_bench.py [n ...]"""
import os
import random
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import masp_amd                                   # noqa: E402
from masp_amd import host as H, workload as W     # noqa: E402
from masp_amd.synthetic import toxic_waste        # noqa: E402

R = H.FR_MODULUS
sizes = [int(x) for x in sys.argv[1:]] or [64, 256, 1024, 4096]
ctx = masp_amd.Context(0, batch_cap=128)
cs = H.circuit("spend")[0]
params = ctx.generate_parameters(cs, toxic_waste(3))
ctx.load_circuit(0, params, cs)
insts = W.instances("spend", 256)
rng = random.Random(1)
nmax = max(sizes)
proofs, pub = [], []
for lo in range(0, nmax, 256):
    proofs += ctx.prove_batch([(0, i, a, rng.randrange(R), rng.randrange(R)) for i, a in insts])
    pub += [W.public_inputs(i) for i, _ in insts]
hvk, gvk = H.PreparedVerifyingKey(params), ctx.prepare_verifying_key(params)
threads = H.effective_cpus()
gvk.verify_batch(proofs[:64], pub[:64])
for n in sizes:
    p, x = proofs[:n], pub[:n]
    t0 = time.perf_counter(); ok_h = hvk.verify_batch(p, x); th = time.perf_counter() - t0
    chunk = max(1, (n + threads - 1) // threads)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        ok_t = all(ex.map(lambda lo: hvk.verify_batch(p[lo:lo + chunk], x[lo:lo + chunk]), range(0, n, chunk)))
    tt = time.perf_counter() - t0
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ok_g = gvk.verify_batch(p, x); best = min(best, time.perf_counter() - t0)
    assert ok_h and ok_t and ok_g
    print("n = %5d   host 1 thread %8.1f ms (%.3f ms/proof)   host %d threads %7.1f ms (%.3f)   GPU %7.1f ms (%.4f ms/proof)"
          % (n, th * 1e3, th * 1e3 / n, threads, tt * 1e3, tt * 1e3 / n, best * 1e3, best * 1e3 / n))

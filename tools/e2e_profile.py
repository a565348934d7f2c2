"""Where the time of `LocalTxProver.prove_batch` goes: N Spend descriptions in chunks of `chunk`, four runs (the first pays the
page-locked buffers), per-stage totals summed over threads.  usage: python tools/e2e_profile.py N chunk"""
import os, sys, time, threading, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import e2e_batch as E
from masp_amd import host as H
from masp_amd.prover import LocalTxProver
from concurrent.futures import ThreadPoolExecutor
n = int(sys.argv[1]); chunk = int(sys.argv[2])
prover = LocalTxProver.with_synthetic_parameters(seed=7)
acc = collections.defaultdict(float); cnt = collections.defaultdict(int); lock = threading.Lock()
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.time()
        try:
            return f(*a, **k)
        finally:
            with lock:
                acc[label] += time.time() - t; cnt[label] += 1
    setattr(obj, name, g)
wrap(prover, "prove_prepared", "prove_prepared")
wrap(prover, "prepare_spend", "prepare_spend")
for k in prover._gpu_vk: wrap(prover._gpu_vk[k], "verify_batch", "gpu_verify")
wrap(prover._ctx, "prove_marshalled", "prove_marshalled")
wrap(prover._ctx, "marshal_jobs", "marshal_jobs")
wrap(prover._ctx, "host_alloc", "host_alloc")
with ThreadPoolExecutor(16) as ex:
    descs = list(ex.map(E.spend_description, range(n)))
for rep in range(4):
    acc.clear(); cnt.clear()
    t = time.time()
    out = prover.prove_batch(prover.new_sapling_proving_context(), descs, chunk=chunk)
    dt = time.time() - t
    print("run %d n=%d chunk=%d: %.1f ms = %.1f proofs/s" % (rep, n, chunk, dt * 1e3, n / dt))
    for k in sorted(acc): print("  %-18s calls %5d  total %8.1f ms  avg %7.2f ms" % (k, cnt[k], acc[k] * 1e3, acc[k] * 1e3 / cnt[k]))

"""Where does host synthesis time go on the GPU box?  Spend witnesses per second into ordinary memory and into page-locked
memory (masp_hip_host_alloc), one by one and in lockstep groups, on 1 / 4 / 16 threads."""
import statistics
import sys
import time

sys.path.insert(0, ".")
import masp_amd
from masp_amd import host as H
from masp_amd import workload as W

ctx = masp_amd.Context(0)
cs = H.circuit("spend")[0]
W.instances("spend", 2, first_seed=10 ** 6, threads=2)
kws = [W.description("spend", 3000 + k)[1] for k in range(256)]
print("cpus", H.effective_cpus())
from concurrent.futures import ThreadPoolExecutor


def run(label, threads, fn, n=256):
    groups = [kws[i:i + H.GROUP] for i in range(0, n, H.GROUP)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(fn, groups))
    dt = time.perf_counter() - t0
    print("%-64s %2d threads: %7.1f witnesses/s  (%.2f ms per witness and thread)" % (label, threads, n / dt, dt * threads / n * 1e3))


pinned = [ctx.host_alloc(cs.n_aux, 32) for _ in range(256)]
import numpy as np
plain = [np.zeros((cs.n_aux, 32), np.uint8) for _ in range(256)]
idx = {id(g[0]): i for i, g in enumerate([kws[i:i + H.GROUP] for i in range(0, 256, H.GROUP)])}
for threads in (1, 4, 16):
    n = 64 if threads == 1 else 256
    for name, bufs in (("ordinary memory", plain), ("page-locked memory", pinned)):
        def one_by_one(g, mont=False):
            b = idx[id(g[0])] * H.GROUP
            for j, kw in enumerate(g):
                ak, nsk = kw["proof_generation_key"]; sib, pos = kw["merkle_path"]
                H.spend_assignment(ak, nsk, kw["diversifier"], kw["rcm"], kw["ar"], kw["asset_type"], kw["value"], kw["anchor"], sib, pos, kw["rcv"], aux_out=bufs[b + j], montgomery=mont)
        def lockstep(g, mont=True):
            b = idx[id(g[0])] * H.GROUP
            W.assignments("spend", g, aux_outs=bufs[b:b + len(g)], montgomery=mont)
        run("one by one, canonical -> " + name, threads, one_by_one, n)
        run("one by one, Montgomery (in place) -> " + name, threads, lambda g: one_by_one(g, True), n)
        run("lockstep groups of 16, Montgomery (in place) -> " + name, threads, lockstep, n)
        run("lockstep groups of 16, canonical -> " + name, threads, lambda g: lockstep(g, False), n)
ctx.close()

"""What does a closed context leave behind on the HOST?  Witness synthesis rate (16 threads, no GPU work) before a context exists, while an idle
one exists, and after it was used and closed; plus this process's thread count and CPU time burnt while sleeping one second."""
import os, sys, time, random, threading, resource
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from concurrent.futures import ThreadPoolExecutor
import masp_amd
from bench import options_from_env
from masp_amd import host as H, synthetic, workload as W
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
cpus = H.effective_cpus()


def synth_rate(tag):
    t0 = time.perf_counter()
    W.instances("spend", 2048, first_seed=7000, threads=cpus, montgomery=True)
    r = 2048 / (time.perf_counter() - t0)
    nthreads = len(os.listdir("/proc/self/task"))
    c0 = resource.getrusage(resource.RUSAGE_SELF)
    time.sleep(1.0)
    c1 = resource.getrusage(resource.RUSAGE_SELF)
    print("%-44s synthesis %7.1f witnesses/s   OS threads %3d   CPU burnt while idle for 1 s: %.2f s" % (tag, r, nthreads, (c1.ru_utime + c1.ru_stime) - (c0.ru_utime + c0.ru_stime)), flush=True)


W.instances("spend", 64, first_seed=1, threads=cpus, montgomery=True)
synth_rate("no context yet")
cs = H.circuit("spend")[0]
c = masp_amd.Context(0, **options_from_env())
params = c.generate_parameters(cs, synthetic.toxic_waste(1))
c.load_circuit(0, params, cs)
synth_rate("context created, circuit loaded, idle")
insts = W.instances("spend", 256, first_seed=0, montgomery=True)
rng = random.Random(1)
rs = np.frombuffer(b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(2 * 4 * 256)), np.uint8).reshape(4, 256, 64)
jobs = [(0, i, a, bytes(rs[0, j, :32]), bytes(rs[0, j, 32:]), None, 1) for j, (i, a) in enumerate(insts)]
h, _ = c.batch_upload(jobs)
for _ in range(3):
    c.batch_prove_resident_steps(h, 256, 4, rs)
synth_rate("context used (12 batches), idle")
c.close()
synth_rate("context closed")
c2 = masp_amd.Context(0, **options_from_env())
c2.load_circuit(0, params, cs)
synth_rate("second context created, idle")
c2.close()

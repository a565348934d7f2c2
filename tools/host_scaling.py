"""Host-side scaling of the C++ witness synthesizer over threads (no GPU work): jobs/s for several pool sizes."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import e2e_batch as E          # noqa: E402
from masp_amd import host as H  # noqa: E402

for f in ("/sys/fs/cgroup/cpu.max", "/proc/loadavg"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, e)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
d = E.spend_description(1)[1]
ak, nsk = d["proof_generation_key"]
sib, pos = d["merkle_path"]


def run(_):
    return H.spend_assignment(ak, nsk, d["diversifier"], d["rcm"], d["ar"], d["asset_type"], d["value"], d["anchor"], sib, pos, d["rcv"])[3]


run(0)
t = time.time()
run(0)
print("1 job, 1 thread: %.1f ms" % ((time.time() - t) * 1e3))
for th in [int(x) for x in (sys.argv[1:] or [8, 32, 64, 128, 256])]:
    n = th * 4
    with ThreadPoolExecutor(th) as ex:
        list(ex.map(run, range(th)))
        t = time.time()
        list(ex.map(run, range(n)))
        dt = time.time() - t
    print("%4d threads: %6.1f jobs/s (%.1f ms per job per thread)" % (th, n / dt, dt * 1e3 * th / n))

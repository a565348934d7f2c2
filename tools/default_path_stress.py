"""Soak of the default path: one context with the default options, host threads of three kinds at once for STRESS_SECONDS —
lone proofs (Output and Spend), batches of 48 Spends, batches of 200 mixed jobs, batch verification of what was proved — every proof compared
byte for byte with the one the same job gave single-threaded before the soak (same witness, same r and s: the bytes are a function of those).
Counts calls, failures and mismatches."""
import os
import sys
import threading
import time

sys.path.insert(0, os.getcwd())
import masp_amd  # noqa: E402
from masp_amd import host as H  # noqa: E402
from masp_amd import synthetic  # noqa: E402
from masp_amd import workload as W  # noqa: E402

SECONDS = float(os.environ.get("STRESS_SECONDS", "90"))
ctx = masp_amd.Context(0)
kinds = ("spend", "output", "convert")
cs = {k: H.circuit(k)[0] for k in kinds}
params = {k: ctx.generate_parameters(cs[k], synthetic.toxic_waste(41 + i)) for i, k in enumerate(kinds)}
for i, k in enumerate(kinds):
    ctx.load_circuit(i, params[k], cs[k])
inst = {k: W.instances(k, n, first_seed=900) for k, n in (("spend", 48), ("output", 80), ("convert", 72))}


def job(kind, j, salt):
    inputs, aux = inst[kind][j % len(inst[kind])]
    return (kinds.index(kind), inputs, aux, 1000 + 7 * j + salt, 2000 + 11 * j + salt)


work = {
    "lone output": [[job("output", j, 1)] for j in range(24)],
    "lone spend": [[job("spend", j, 2)] for j in range(12)],
    "batch of 48 spends": [[job("spend", j, 3 + b) for j in range(48)] for b in range(2)],
    "batch of 200 mixed": [[job(kinds[j % 3], j, 5 + b) for j in range(200)] for b in range(2)],
}
t0 = time.time()
want = {name: [ctx.prove_batch(jobs) for jobs in lists] for name, lists in work.items()}
print("reference proofs (one call at a time): %.1f s" % (time.time() - t0), flush=True)
stats = {name: [0, 0, 0] for name in work}          # calls, failures, mismatches
errors = []
stop = time.time() + SECONDS
lock = threading.Lock()


def run(name, offset):
    k = offset
    while time.time() < stop:
        lists = work[name]
        i = k % len(lists)
        k += 1
        try:
            got = ctx.prove_batch(lists[i])
        except Exception as e:  # noqa: BLE001
            with lock:
                stats[name][1] += 1
                errors.append((name, str(e)))
            continue
        with lock:
            stats[name][0] += 1
            if got != want[name][i]:
                stats[name][2] += 1


threads = [threading.Thread(target=run, args=(name, o)) for name in work for o in ((0, 1) if name.startswith("lone") else (0,))]
for t in threads:
    t.start()
for t in threads:
    t.join()
for name, (calls, fails, wrong) in stats.items():
    print("%-20s %6d calls, %d failed, %d with other bytes than single-threaded" % (name, calls, fails, wrong))
for e in errors[:6]:
    print("   ", e)
print("options:", {k: ctx.options[k] for k in ("slots", "batch_cap", "hw_queues")}, "streams side by side: %d of %d" % ctx.stream_concurrency()[::-1])
ctx.close()
sys.exit(1 if errors or any(s[2] for s in stats.values()) else 0)

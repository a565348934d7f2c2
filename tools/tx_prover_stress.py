"""Soak of the Python mirror: ONE LocalTxProver shared by three host threads for STRESS_SECONDS — two calling prove_batch over 384 mixed descriptions
(synthesis threads, GPU batches, GPU batch self-verification inside), one calling the trait's single-description methods — each with a context of its
own and explicit blinding scalars, so that every result (proofs, cv, rk, bsk, cv_sum) must equal the one the same call gave before the soak."""
import os
import random
import sys
import threading
import time

sys.path.insert(0, os.getcwd())
from masp_amd import host as H  # noqa: E402
from masp_amd import prover as P  # noqa: E402
from masp_amd import workload as W  # noqa: E402

SECONDS = float(os.environ.get("STRESS_SECONDS", "60"))
lp = P.LocalTxProver.with_synthetic_parameters(seed=12)
rng = random.Random(8)
kinds = ("spend", "output", "convert")
lists = [[W.description(kinds[(j + s) % 3], 3000 + 1000 * s + j) for j in range(384)] for s in range(2)]
rss = [[(rng.randrange(H.FR_MODULUS), rng.randrange(H.FR_MODULUS)) for _ in l] for l in lists]
singles = [W.description(kinds[j % 3], 7000 + j) for j in range(9)]
srs = [(rng.randrange(H.FR_MODULUS), rng.randrange(H.FR_MODULUS)) for _ in singles]


def batch(i):
    ctx = lp.new_sapling_proving_context()
    out = lp.prove_batch(ctx, lists[i], rs=rss[i], threads=6)
    return [tuple(bytes(x) for x in o) for o in out], ctx.bsk, bytes(ctx.cv_sum)


def single():
    ctx = lp.new_sapling_proving_context()
    out = []
    for (kind, kw), rs in zip(singles, srs):
        if kind == "spend":
            out.append(lp.spend_proof(ctx, kw["proof_generation_key"], kw["diversifier"], kw["rcm"], kw["ar"], kw["asset_type"], kw["value"], kw["anchor"],
                                      kw["merkle_path"], kw["rcv"], rs=rs))
        elif kind == "output":
            out.append(lp.output_proof(ctx, kw["esk"], kw["payment_address"], kw["rcm"], kw["asset_type"], kw["value"], kw["rcv"], rs=rs))
        else:
            out.append(lp.convert_proof(ctx, kw["allowed_conversion"], kw["value"], kw["anchor"], kw["merkle_path"], kw["rcv"], rs=rs))
    return [tuple(bytes(x) for x in o) for o in out], ctx.bsk, bytes(ctx.cv_sum)


t0 = time.time()
want = {"batch 0": batch(0), "batch 1": batch(1), "singles": single()}
print("reference results: %.1f s" % (time.time() - t0), flush=True)
stats = {k: [0, 0, 0] for k in want}
errors = []
stop = time.time() + SECONDS
lock = threading.Lock()


def run(name, fn):
    while time.time() < stop:
        try:
            got = fn()
        except Exception as e:  # noqa: BLE001
            with lock:
                stats[name][1] += 1
                errors.append((name, repr(e)))
            continue
        with lock:
            stats[name][0] += 1
            stats[name][2] += got != want[name]


th = [threading.Thread(target=run, args=("batch 0", lambda: batch(0))), threading.Thread(target=run, args=("batch 1", lambda: batch(1))),
      threading.Thread(target=run, args=("singles", single))]
for t in th:
    t.start()
for t in th:
    t.join()
for name, (calls, fails, wrong) in stats.items():
    print("%-8s %5d calls (%s), %d failed, %d with other results than before the soak" % (name, calls, "384 mixed descriptions each" if name != "singles" else "9 descriptions each", fails, wrong))
for e in errors[:6]:
    print("   ", e)
lp.close()
sys.exit(1 if errors or any(s[2] for s in stats.values()) else 0)

"""The process-per-GPU path of bench.py with a REAL RCCL process group on one GPU: `MASP_BENCH_FORCE_DIST=1` makes a single rank
initialise torch.distributed (backend "nccl" = RCCL), broadcast the CRS, gather the proofs and reduce the timings through the
same collectives an 8-rank run uses.  If that path is broken, this fails on a one-GPU box; the figure must equal the plain
single-process run's within noise.  Run with `-m gpu` on an MI355X."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env):
    env = dict(os.environ, MASP_BENCH_E2E="0", **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), out.stderr


def test_bench_with_a_one_rank_rccl_process_group_matches_the_plain_run():
    plain, _ = _bench({})
    dist, err = _bench({"MASP_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29541"})
    assert "RCCL process group up: 1 rank(s)" in err
    assert plain["rccl_ranks"] == 1 and dist["rccl_ranks"] == 1 and dist["n_gpus"] == 1
    assert dist["collectives"] == "rccl" and plain["collectives"] == "none"
    for d in (plain, dist):
        assert d["verified"] == 6 * 256 and d["steps"] == 6 and d["unit"] == plain["unit"]
    # Same figure: with the wrappers' default of 16 hardware queues a live torch + RCCL runtime costs nothing next to the prover's own
    # streams (profiles/r04e_bench_plain_vs_one_rank_rccl_hw_queues.txt: 1 337 - 1 356 vs 1 337 - 1 351; with 8 queues it took 7 - 10 %).
    # Six-step regions on a shared box are noisy (one plain run in a dozen came out 12 % low): 12 % allowed, and a pair that
    # misses it is measured once more — the better run of each side counts
    def close_enough(a, b):
        return abs(a["value"] - b["value"]) <= 0.12 * b["value"] and abs(a["resident"]["value"] - b["resident"]["value"]) <= 0.12 * b["resident"]["value"]
    if not close_enough(dist, plain):
        plain2, _ = _bench({})
        dist2, _ = _bench({"MASP_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29542"})
        best = lambda x, y: x if x["value"] >= y["value"] else y
        plain, dist = best(plain, plain2), best(dist, dist2)
    assert close_enough(dist, plain), (dist["value"], plain["value"], dist["resident"]["value"], plain["resident"]["value"])

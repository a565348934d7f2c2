"""The process-per-GPU path of bench.py with a REAL RCCL process group of ONE rank on one GPU (`MASP_BENCH_FORCE_DIST=1`).

What runs: torch.distributed initialises backend "nccl" (= RCCL), and since round 5 the helpers of masp_amd/distributed.py no longer
return early for a group of one rank, so the calls an 8-rank run makes are really issued on `cuda` tensors — the CRS through
`broadcast_bytes` (an int64 length + a u8 payload of ~48 MB), the u8 `gather` of K x 256 x 192 bytes for both timed regions, the
float64 `all_reduce`s of the timings and counts.  The call sites count what they sent (`collective_calls` in the line) and this test
asserts the counts and the payload sizes; rank 0 re-verifies what the gather delivered (`gathered_checked`).
What does NOT run here: any transfer between two GPUs — a one-rank collective is a device-local copy inside RCCL.  The two-rank code
path with real proofs is tests/test_gpu_bench_two_ranks.py (gloo: RCCL refuses two ranks on one device).
The figure must equal the plain single-process run's: a live torch + RCCL runtime next to the prover's own streams costs nothing at
16 hardware queues (profiles/r04e_bench_plain_vs_one_rank_rccl_hw_queues.txt; at 8 queues it took 7 - 10 %).  Run with `-m gpu`."""
import json
import os
import statistics
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 6


def _bench(extra_env):
    env = dict(os.environ, MASP_BENCH_E2E="0", MASP_BENCH_LONE="0", **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(STEPS), "--warmup", "2", "--no-cpu-baseline"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), out.stderr


def test_bench_with_a_one_rank_rccl_process_group_runs_its_collectives_and_matches_the_plain_run():
    plain, dist = [], []
    for k in range(3):                       # interleaved, so that a box that gets busier hits both sides
        plain.append(_bench({})[0])
        d, err = _bench({"MASP_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29541 + k)})
        assert "RCCL process group up: 1 rank(s)" in err
        dist.append(d)
    for d in plain:
        assert d["rccl_ranks"] == 1 and d["collectives"] == "none" and d["collective_calls"] is None and d["gathered_checked"] is None
    for d in dist:
        assert d["rccl_ranks"] == 1 and d["n_gpus"] == 1 and d["collectives"] == "rccl" and d["collective_tensors"] == "cuda:0"
        calls = d["collective_calls"]
        # both timed regions gather K x 256 proofs of 192 bytes as ONE u8 device tensor each
        assert calls["gather"] == {"calls": 2, "bytes": 2 * STEPS * 256 * 192}
        # the Spend CRS: one int64 length + the payload (48 482 520-byte body + the verifying key and lengths in front)
        assert calls["broadcast"]["calls"] == 2 and calls["broadcast"]["bytes"] > 48_000_000
        # max over ranks of the two regions' times + the verified count (+ whatever else the line reduces)
        assert calls["all_reduce"]["calls"] >= 3 and calls["all_reduce"]["bytes"] == 8 * calls["all_reduce"]["calls"]
        g = d["gathered_checked"]
        assert g["ranks"] == 1 and g["proofs_verified"] == 2 * STEPS * 256 and g["closed_form_equal"] == 2
    for d in plain + dist:
        assert d["verified"] == STEPS * 256 and d["steps"] == STEPS and d["unit"] == plain[0]["unit"]
    # Same figure, medians of three: 6 % on the wall-clock rates, 4 % on the GPU events of the resident region (which see no host noise)
    med = lambda runs, f: statistics.median(f(r) for r in runs)
    for name, f, tol in (("value", lambda r: r["value"], 0.06), ("resident", lambda r: r["resident"]["value"], 0.06),
                         ("gpu_event_ms_per_step", lambda r: r["resident"]["gpu_event_ms_per_step"], 0.04)):
        a, b = med(dist, f), med(plain, f)
        assert abs(a - b) <= tol * b, (name, [f(r) for r in dist], [f(r) for r in plain])

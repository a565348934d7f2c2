"""The N = 2 code path of bench.py with REAL proofs on a one-GPU box: `python bench.py --gpus 2` re-executes itself under
torch.distributed.run as two ranks — backend gloo, because RCCL refuses two ranks on one device — which share device 0 (one slot
each, small tree sub-batches so that two provers fit side by side).  Everything an N-rank run does apart from the transport is
exercised with real data: the jobs sharded by rank (instances and blinding scalars seeded by rank), the host threads split between the
ranks, the CRS generated on rank 0 and broadcast, both timed regions with barrier + max over ranks, the gather of every rank's
K x 256 proofs to rank 0 — where EVERY gathered proof of BOTH ranks is verified against the statements rank 0 re-derives for that rank
and two per rank are byte-compared with the oracle's closed form at their job positions (bench.py, `gathered_checked`) — and the
end-to-end region per rank.  The figure itself means nothing (two ranks on one GPU).  Run with `-m gpu`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu_gathers_real_proofs():
    sys.path.insert(0, ROOT)
    from masp_amd.host import effective_cpus
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MASP_BENCH_BACKEND="gloo", MASP_HIP_SLOTS="1", MASP_HIP_TREE_SUB="32", MASP_BENCH_E2E="128", MASP_BENCH_LONE="0", OMP_NUM_THREADS="1")
    K = 2
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(K), "--warmup", "1", "--no-cpu-baseline"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert "torch.distributed.run" in out.stderr and "gloo process group up: 2 rank(s)" in out.stderr
    assert "rank 1 shares device 0" in out.stderr and "rank 1: CRS received" in out.stderr and "rank 0: CRS generated" in out.stderr
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collectives"] == "gloo" and d["collective_tensors"] == "cpu"
    assert d["steps"] == K and d["config"]["proofs_per_gpu"] == K * 256
    assert d["verified"] == 2 * K * 256                 # summed over ranks: each rank verified its own timed proofs
    g = d["gathered_checked"]                           # ... and rank 0 everything the gather delivered, both regions
    assert g == {"ranks": 2, "proofs_verified": 2 * 2 * K * 256, "closed_form_equal": 4, "seconds": g["seconds"]}
    calls = d["collective_calls"]
    assert calls["gather"] == {"calls": 2, "bytes": 2 * K * 256 * 192}          # (this rank's payload; rank 0 receives twice that)
    assert calls["broadcast"]["calls"] == 2 and calls["broadcast"]["bytes"] > 48_000_000
    assert d["value"] > 0 and d["resident"]["value"] > 0
    assert d["host_synthesis"]["threads"] == max(1, effective_cpus() // 2)      # the host cores split between the ranks
    e = d["end_to_end"]
    assert e["descriptions_per_gpu"] == 128 and e["threads_per_gpu"] == max(1, effective_cpus() // 2) and e["value"] > 0


def test_bench_eight_ranks_rehearsal_on_one_gpu():
    """The driver's SCALE run should not be the first time eight ranks meet: `bench.py --gpus 8` as eight gloo ranks
    that all share device 0 — one slot each, tree sub-batches of 16, the Output circuit so that eight provers fit one HBM side by side,
    K = 1 — with real proofs: eight CRS receives (rank 0 generates, seven receive the same bytes), eight shards of 256 proofs gathered
    and re-verified on rank 0 against the statements it re-derives per rank, the host threads split eight ways.  The line carries what the
    host limits: `host_synthesis.witnesses_per_s_per_rank` next to the GPU's rate and the `host_bound` flag."""
    sys.path.insert(0, ROOT)
    from masp_amd.host import effective_cpus
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MASP_BENCH_BACKEND="gloo", MASP_BENCH_CIRCUIT="output", MASP_HIP_SLOTS="1", MASP_HIP_TREE_SUB="16", MASP_BENCH_LONE="0", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=2400)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    assert len(lines[0]) <= 4608, len(lines[0])          # the line stays within what the driver's record keeps
    d = json.loads(lines[0])
    assert "gloo process group up: 8 rank(s)" in out.stderr
    for r in range(1, 8):
        assert "rank %d: CRS received" % r in out.stderr and "rank %d shares device 0" % r in out.stderr
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["collectives"] == "gloo" and d["metric"] == "output proofs/sec"
    assert d["verified"] == 8 * 256
    g = d["gathered_checked"]
    assert g["ranks"] == 8 and g["proofs_verified"] == 2 * 8 * 256 and g["closed_form_equal"] == 16
    assert d["collective_calls"]["gather"] == {"calls": 2, "bytes": 2 * 256 * 192}
    assert d["collective_calls"]["broadcast"]["calls"] == 2 and d["collective_calls"]["broadcast"]["bytes"] > 15_000_000      # the Output CRS
    hs = d["host_synthesis"]
    assert hs["threads"] == max(1, effective_cpus() // 8)
    assert hs["witnesses_per_s_per_rank"] > 0 and hs["proofs_per_s_per_gpu"] > 0
    assert hs["host_bound"] == (hs["witnesses_per_s_per_rank"] < hs["proofs_per_s_per_gpu"])
    assert d["notes"] == "BENCH_NOTES.md" and d["value"] > 0 and d["resident"]["value"] > 0

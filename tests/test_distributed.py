"""The N > 1 path of bench.py / prove sharding, exercised with world_size 2 on CPU (gloo): job sharding, the final
gather of 192-byte proofs to rank 0 in job order, and the max-over-ranks timing reduction."""
import hashlib
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_proof(job):
    return (hashlib.sha256(b"job%d" % job).digest() * 6)[:192]


def _worker(rank, world, port, n_jobs, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from masp_amd import distributed as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = [_fake_proof(j) for j in D.shard(n_jobs, rank, world)]
    allp = D.gather_proofs(mine, n_jobs, dist)
    t = D.max_over_ranks(1.0 + rank, dist)
    dist.barrier()
    q.put((rank, allp, t))
    dist.destroy_process_group()


def test_shard_partitions_jobs():
    from masp_amd import distributed as D
    for n in (0, 1, 7, 256, 4096):
        for world in (1, 2, 3, 8):
            seen = [j for r in range(world) for j in D.shard(n, r, world)]
            assert seen == list(range(n))
            sizes = [len(D.shard(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("n_jobs", [5, 8])
def test_gather_world_size_2_gloo(n_jobs):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_jobs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, allp, t = q.get(timeout=120)
        res[rank] = (allp, t)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1][0] is None
    assert res[0][0] == [_fake_proof(j) for j in range(n_jobs)]
    assert res[0][1] == res[1][1] == 2.0


def _run_bench(argv, env_extra, timeout=300):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_launches_its_own_ranks_world_size_2_gloo():
    """`python bench.py --gpus 2` without a launcher must itself become 2 ranks (it re-executes under
    torch.distributed.run), bring the process group up, gather every rank's proofs on rank 0 in job order and print ONE
    line that says n_gpus = 2.  --dry-run + gloo: the GPU work is skipped, the launcher / sharding / gather path is real."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run"], {"MASP_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["max_rank_seen"] == 1
    assert out["gathered"] == 2 * 3 * 256 and out["gather_ok"] is True and out["value"] is None
    assert "exec" in r.stderr and "torch.distributed.run" in r.stderr and "2 rank(s)" in r.stderr


def test_bench_refuses_a_gpu_count_it_was_not_launched_with():
    """A launcher that started 1 rank while --gpus says 4 (or the reverse) gets an error, not a line with another n_gpus."""
    r = _run_bench(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr and "{" not in r.stdout


def test_bench_dress_rehearsal_eight_ranks_as_the_driver_launches_them():
    """The 8-GPU line of SCALE_rNN.json, without the GPUs: the driver's own command (python -m torch.distributed.run --nnodes=1
    --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...), gloo instead of RCCL, --dry-run instead of
    proving.  Eight ranks come up, the CRS bytes of rank 0 reach all of them, each rank's K x 256 proofs arrive on rank 0 at
    their job positions, one line comes out and it says n_gpus = 8."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MASP_BENCH_BACKEND"] = "gloo"
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["max_rank_seen"] == 7
    assert out["gathered"] == 8 * 2 * 256 and out["gather_ok"] is True and out["crs_broadcast_ok_ranks"] == 8


def _one_rank_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch.distributed as dist
    from masp_amd import distributed as D
    dist.init_process_group("gloo", rank=0, world_size=1)
    before = D.collective_counts()
    got = D.gather_proofs([_fake_proof(j) for j in range(5)], 5, dist)
    crs = D.broadcast_bytes(np.arange(1000, dtype=np.uint8), dist)
    t = D.max_over_ranks(3.5, dist) + D.sum_over_ranks(1.0, dist)
    after = D.collective_counts()
    dist.destroy_process_group()
    # without a process group the helpers hand their argument back and count nothing
    D.gather_proofs([_fake_proof(0)], 1, None)
    D.max_over_ranks(1.0, None)
    q.put((before, after, D.collective_counts(), got == [_fake_proof(j) for j in range(5)], crs.tobytes() == bytes(range(256)) * 3 + bytes(range(232)), t))


def test_a_process_group_of_one_rank_still_runs_the_collectives():
    """Round 4's helpers returned early when the group had one rank, so the one-GPU box never issued a gather / broadcast / all_reduce
    (missing 1): with a process group the collectives run whatever its size; the call sites count them."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(port, q))
    p.start()
    before, after, final, gather_ok, crs_ok, t = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0 and gather_ok and crs_ok and t == 4.5
    assert before == {k: {"calls": 0, "bytes": 0} for k in ("gather", "broadcast", "all_reduce")}
    assert after == {"gather": {"calls": 1, "bytes": 5 * 192}, "broadcast": {"calls": 2, "bytes": 8 + 1000}, "all_reduce": {"calls": 2, "bytes": 16}}
    assert final == after

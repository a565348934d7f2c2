"""Multi-GPU sharding of independent proving jobs: one process per GPU, no data-path collective, one final gather
of the 192-byte proofs to rank 0.  Works with any initialised torch.distributed backend
("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import numpy as np

PROOF_BYTES = 192
# what actually went through the process group, per call kind: [calls, payload bytes of this rank] — bench.py prints it, the GPU
# tests assert on it (a helper that returns early without touching the backend leaves its counter where it was)
COLLECTIVES = {"gather": [0, 0], "broadcast": [0, 0], "all_reduce": [0, 0]}


def _count(kind, nbytes):
    COLLECTIVES[kind][0] += 1
    COLLECTIVES[kind][1] += int(nbytes)


def collective_counts():
    return {k: {"calls": v[0], "bytes": v[1]} for k, v in COLLECTIVES.items()}


def shard(n_jobs, rank, world):
    """Contiguous shard of job indices for `rank`; sizes differ by at most one."""
    base, extra = divmod(n_jobs, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def gather_proofs(local_proofs, n_jobs, dist=None, device=None):
    """Every rank passes the proofs of its shard in job order (a list of 192-byte strings or a u8 array [m, 192]); rank 0
    gets all n_jobs proofs in job order as a list of bytes, the others None."""
    import torch
    if not isinstance(local_proofs, np.ndarray):
        local_proofs = np.frombuffer(b"".join(local_proofs), dtype=np.uint8).reshape(-1, PROOF_BYTES)
    if dist is None:
        return [local_proofs[i].tobytes() for i in range(local_proofs.shape[0])]
    # (a process group of ONE rank runs the collective too: the one-GPU box then exercises the same RCCL calls on device tensors
    # that an 8-rank run makes)
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = (n_jobs + world - 1) // world               # gather needs equal sizes: pad to the largest shard
    buf = np.zeros((cap, PROOF_BYTES), dtype=np.uint8)
    buf[:local_proofs.shape[0]] = local_proofs
    mine = torch.from_numpy(buf)
    if device is not None:
        mine = mine.to(device)
    out = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, out, dst=0)
    _count("gather", mine.numel())
    if rank != 0:
        return None
    proofs = []
    for r in range(world):
        arr = out[r].cpu().numpy()
        for i in range(len(shard(n_jobs, r, world))):
            proofs.append(arr[i].tobytes())
    return proofs


def max_over_ranks(value, dist=None, device=None):
    import torch
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    _count("all_reduce", 8)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    import torch
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    _count("all_reduce", 8)
    return float(t.item())


def broadcast_bytes(data, dist=None, device=None, src=0):
    """`data` (bytes / u8 array) on rank `src`, None elsewhere -> the same u8 array on every rank (two broadcasts: length, payload).
    Used for the CRS: generated once, loaded by every rank from the same bytes."""
    import torch
    if dist is None:
        return np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
    rank = dist.get_rank()
    if rank == src:
        arr = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        n = torch.tensor([arr.size], dtype=torch.int64, device=device)
    else:
        n = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    _count("broadcast", 8)
    size = int(n.item())
    if rank == src:
        t = torch.from_numpy(arr.copy())
        if device is not None:
            t = t.to(device)
    else:
        t = torch.empty(size, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    _count("broadcast", size)
    return t.cpu().numpy()

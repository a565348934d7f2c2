#!/usr/bin/env python3
"""Spend proofs/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one Groth16 Spend proof (witness assignment already resident in HBM; static-R1CS evaluation, 7 NTTs,
4 G1 MSMs + 1 G2 MSM, assembly, 192-byte proof) — BASELINE.json configs[1].  Steps are independent proofs, so
ranks shard them with no data-path collective (weak scaling: K proofs per GPU); the only collective is the final
RCCL gather of the N*K*192 proof bytes to rank 0, inside the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
WORKLOAD = os.environ.get("MASP_BENCH_CIRCUIT", "spend")


def make_jobs(n_distinct, total, instances):
    """`total` jobs, job j of circuit slot j mod len(instances), cycling over `n_distinct` independent witnesses per
    circuit, each job with its own (r, s)."""
    import random
    rng = random.Random(0x5962be3d)  # the reference bench's XorShift seed bytes, benches/sapling.rs:19-22
    R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    jobs = []
    for j in range(total):
        slot = j % len(instances)
        _, inputs, aux = instances[slot][(j // len(instances)) % n_distinct]
        jobs.append((slot, inputs, aux, rng.randrange(R), rng.randrange(R)))
    return jobs


SYNTHESIS_MS = {}    # per circuit: wall time of building one instance (witness preparation + C++ synthesis), one host thread


def real_instances(kind, n, rank):
    """n independent, valid instances of the real MASP circuit `kind`, shaped like the reference's benches
    (masp_proofs/benches/sapling.rs:39-69, benches/convert.rs:32-53) but with the anchor set to the computed root."""
    import random
    from masp_amd import host as H
    cs, _ = H.circuit(kind)
    out = []
    for k in range(n):
        t_syn = time.perf_counter()
        rng = random.Random("masp-bench-%s-%d-%d" % (kind, rank, k))
        sc = lambda: rng.randrange(1, H.JUBJUB_ORDER)
        siblings = [rng.randrange(H.FR_MODULUS) for _ in range(32)]
        pos = rng.getrandbits(32)
        if kind == "spend":
            ident = H.asset_identifier(b"benchmark")
            ak = H.jubjub_mul(H.point_bytes(*H.generator_uv(4)), sc())
            nsk, ar, rcm, rcv = sc(), sc(), sc(), sc()
            while True:
                d = bytes(rng.getrandbits(8) for _ in range(11))
                try:
                    cmu, _ = H.spend_leaf(ak, nsk, d, rcm, ident, 1)
                    break
                except H.HostError:
                    continue
            inputs, aux, *_ = H.spend_assignment(ak, nsk, d, rcm, ar, ident, 1, H.merkle_root(cmu, siblings, pos), siblings, pos, rcv)
        elif kind == "output":
            ident = H.asset_identifier(b"benchmark")
            pk = H.jubjub_mul(H.point_bytes(*H.generator_uv(0)), sc())
            while True:
                d = bytes(rng.getrandbits(8) for _ in range(11))
                try:
                    inputs, aux, _ = H.output_assignment(sc(), d, pk, sc(), ident, 1, sc())
                    break
                except H.HostError:
                    continue
        else:
            gen = H.asset_generator(H.asset_identifier(b"asset %d" % k))
            inputs, aux, _ = H.convert_assignment(gen, 1 + rng.getrandbits(40), H.merkle_root(H.convert_cmu(gen), siblings, pos), siblings, pos, sc())
        out.append((cs, inputs, aux))
        SYNTHESIS_MS.setdefault(kind, []).append((time.perf_counter() - t_syn) * 1e3)
    return out


def cpu_baseline(cs, params, inputs, aux, budget_s=20.0):
    """Oracle (C++ restatement of bellperson's CPU prover, oracle/) timed on the host cores: reported baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from masp_amd.host import effective_cpus
    O.lib().oracle_set_threads(effective_cpus())     # the cores this process may actually use (cgroup quota), not nproc
    P = O.Params(params)
    n, t0 = 0, time.perf_counter()
    phases = {}
    while True:
        tm = {}
        O.create_proof(P, cs, inputs, aux, 7 + n, 11 + n, timings=tm)
        for k, v in tm.items():
            phases[k] = phases.get(k, 0.0) + v
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 8:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "proofs/s", "cores": O.lib().oracle_get_threads(), "kind": "port",
            "sample": "%d %s proofs, same circuit, CRS and witness as the GPU run (oracle/groth16_oracle.cpp, all host cores)" % (n, WORKLOAD),
            "phase_ms_per_proof": {k: round(v / n, 2) for k, v in phases.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed steps; one step = one GPU batch of MASP_HIP_BATCH (96) proofs")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # let independent proofs' kernels overlap (ROCm default: 4)
    os.environ.setdefault("MASP_HIP_SLOTS", "4")
    os.environ.setdefault("MASP_HIP_BATCH", "96")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or os.environ.get("MASP_BENCH_FORCE_DIST"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # torch first: it brings its own HIP runtime; libmasp_hip then binds to the already loaded one
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        local_rank %= max(1, torch.cuda.device_count())      # a launcher may expose one device per rank (HIP_VISIBLE_DEVICES)
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    import masp_amd
    from masp_amd import synthetic

    ctx = masp_amd.Context(local_rank)
    n_distinct = 4
    kinds = ["spend", "output", "convert"] if WORKLOAD == "mixed" else [WORKLOAD]   # mixed = BASELINE.json configs[4] job mix
    if os.environ.get("MASP_BENCH_SYNTHETIC_SHAPE"):
        instances = [[synthetic.shaped(k, seed=rank * 1000 + i) for i in range(n_distinct)] for k in kinds]
        circuit_desc = "%s-SHAPED synthetic R1CS (masp_amd/synthetic.py)" % WORKLOAD
    else:
        instances = [real_instances(k, n_distinct, rank) for k in kinds]
        circuit_desc = "the real MASP %s circuit(s) (structure hashes pinned to the reference's KATs), witnesses from the C++ synthesizer" % "/".join(kinds)
    shaped = instances[0]
    cs = shaped[0][0]
    params = None
    for slot, inst in enumerate(instances):
        p_ = ctx.generate_parameters(inst[0][0], synthetic.toxic_waste(1 + slot))   # same CRS on every rank
        ctx.load_circuit(slot, p_, inst[0][0])
        params = params if params is not None else p_
    # one step = one pass of the hot path over one batch: B proofs enqueued as a single launch sequence on one stream
    K, W, B = args.steps, args.warmup, int(os.environ["MASP_HIP_BATCH"])
    warm = ctx.batch_upload(make_jobs(n_distinct, max(W, 1) * B, instances))
    timed = ctx.batch_upload(make_jobs(n_distinct, K * B, instances))
    if W > 0:
        ctx.batch_prove_resident(*warm)
    # single-proof latency (not the headline value)
    one = ctx.batch_upload(make_jobs(n_distinct, 1, instances))
    ctx.batch_prove_resident(*one)              # sizes the lone-proof workspace
    lat = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.batch_prove_resident(*one)
        lat.append((time.perf_counter() - t0) * 1e3)
    latency_ms = sorted(lat)[len(lat) // 2]

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    ctx.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    proofs, gpu_ms = ctx.batch_prove_resident(*timed)       # exactly K steps
    from masp_amd import distributed as D
    if dist is not None:
        import torch
        dev = torch.device("cuda", local_rank)
        all_proofs = D.gather_proofs(proofs, K * B * world, dist, dev)   # RCCL over xGMI: N*K*B*192 bytes to rank 0
    else:
        all_proofs = proofs
    barrier()
    elapsed = time.perf_counter() - t0
    acc_ms, launches, alg_bytes = ctx.profile_read()
    ctx.profile_enable(False)
    if dist is not None:
        elapsed = D.max_over_ranks(elapsed, dist, dev)
        if rank == 0:
            assert len(all_proofs) == K * B * world
    if rank == 0:
        assert len(set(proofs)) == len(proofs) and all(len(p) == 192 for p in proofs)
        total = K * B * world
        achieved = alg_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Spend proofs/sec" if WORKLOAD == "spend" else "%s proofs/sec" % WORKLOAD, "value": total / elapsed, "unit": "proofs/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s proofs (BASELINE.json configs[1] instances), one step = one batch of %d independent proofs; %s + synthetic CRS from known toxic waste "
                                   "(NTT 2^%d, G1 MSMs %d/%d/%d/%d, G2 MSM %d), witness resident in HBM, "
                                   "batches of %s proofs per launch sequence on %s HIP streams"
                                   % (WORKLOAD, B, circuit_desc, cs.logm, (1 << cs.logm) - 1, cs.n_aux,
                                      synthetic.SHAPES[kinds[0]][3] + cs.n_inputs, synthetic.SHAPES[kinds[0]][4] + 1,
                                      synthetic.SHAPES[kinds[0]][4] + 1, os.environ.get("MASP_HIP_BATCH"), os.environ.get("MASP_HIP_SLOTS")),
                       "proofs_per_step": B, "proofs_per_gpu": K * B, "arithmetic": "384-bit Fp / 255-bit Fr modular integers in 32-bit limbs (v_mad_u64_u32)", "parallelism": "proofs sharded over %d GPU(s), RCCL gather of proofs" % world},
            "single_proof_latency_ms": latency_ms,
            # not part of `value` (assignments are resident when the timed region starts): one host thread, libmasp_host
            "host_synthesis_ms_per_proof": {k: round(min(v), 2) for k, v in SYNTHESIS_MS.items()},
            "ms_per_proof": elapsed * 1e3 / (K * B),
            "gpu_event_ms_per_step": gpu_ms / K,
            "roofline": {"bound": "hbm", "kernel": "k_msm_accumulate<G1> (bucket accumulation of the 4 G1 MSMs)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "launches": launches, "avg_launch_ms": acc_ms / launches if launches else None,
                         "alg_bytes_per_launch": alg_bytes / launches if launches else None,
                         "note": "algorithmic bytes = n x (96 B base + 32 B scalar) per G1 MSM; this path is bound by 32-bit integer "
                                 "multiply throughput, not HBM (DESIGN.md)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cs, params, shaped[0][1], shaped[0][2])
    else:
        out = None
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes its version banner (NCCL_DEBUG=VERSION) through C stdio, which is flushed at exit when stdout is a file:
    # push it out now so that the JSON line really is the last thing on stdout
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if out is not None:
        sys.stderr.flush()
        if world > 1:
            time.sleep(1.0)                     # the other ranks share this stdout: let their flushes land first
        print(json.dumps(out), flush=True)      # the ONE JSON line, last thing on stdout


if __name__ == "__main__":
    main()

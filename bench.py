#!/usr/bin/env python3
"""Spend proofs/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload = BASELINE.json configs[3]: one step = one batch of 256 DISTINCT Spend proofs on one GPU (256 independent
instances shaped like /root/reference/masp_proofs/benches/sapling.rs:38-86: own keys, diversifier, Merkle path, randomness),
each proof = static-R1CS evaluation, 7 NTTs of 2^17, 4 G1 MSMs + 1 G2 MSM, assembly, 192 bytes.  Steps reuse the 256
witnesses with fresh blinding scalars (r, s), so every timed proof is different.  MASP_BENCH_CIRCUIT=output|convert|mixed
selects the other workloads (mixed = configs[4]'s job mix: 512 jobs per GPU and step, job j of circuit j mod 3).

Two timed regions, K steps each, bracketed by barrier + device synchronisation, max over ranks:
  `value`     BASELINE.md §4's region: witnesses in page-locked HOST memory -> proofs on the host of rank 0, through
              masp_hip_prove_batch (H2D of 3.2 MB per Spend, D2H of the proofs and the RCCL gather of N*K*256*192 bytes inside)
  `resident`  witnesses already resident in HBM -> proofs on the host (what rounds 1-2 reported as `value`; 2 % faster)
plus `end_to_end`: LocalTxProver.prove_batch over 1 024 Spend descriptions per GPU — witness synthesis on the host cores, H2D,
proving and the GPU batch self-verification (not part of `value`: the metric starts from witnesses).
The library reads no environment; this script translates MASP_HIP_SLOTS / MASP_HIP_BATCH / MASP_HIP_NTT_SUB /
MASP_HIP_MSM_C_{H,LA,B,B2_LONE} into masp_hip_options (A/B runs of the tools).
After timing, EVERY timed proof is checked with the product's Groth16 batch verifier (pairing equation; no oracle involved)
and a sample is compared byte for byte with the oracle's toxic-waste closed form; the line carries `verified`.
Ranks shard the proofs with no data-path collective (weak scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import random
import socket
import statistics
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
WORKLOAD = os.environ.get("MASP_BENCH_CIRCUIT", "spend")
KINDS = ("spend", "output", "convert")
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
PROOFS_PER_STEP = {"spend": 256, "output": 256, "convert": 256, "mixed": 512}   # configs[3]; configs[4] = 4096 / 8 GPUs


ENV_OPTIONS = {"MASP_HIP_SLOTS": "slots", "MASP_HIP_BATCH": "batch_cap", "MASP_HIP_NTT_SUB": "ntt_sub_batch", "MASP_HIP_MSM_C_H": "window_bits_h",
               "MASP_HIP_MSM_C_LA": "window_bits_la", "MASP_HIP_MSM_C_B": "window_bits_b", "MASP_HIP_MSM_C_B2": "window_bits_b2", "MASP_HIP_MSM_C_B2_LONE": "window_bits_b2_lone",
               "MASP_HIP_WITNESS_NONTRIVIAL_PERCENT": "witness_nontrivial_percent", "MASP_HIP_TREE_LEVELS": "bucket_tree_levels",
               "MASP_HIP_TREE_SUB": "bucket_tree_sub_batch", "MASP_HIP_TREE_LEVELS_G2": "bucket_tree_levels_g2",
               "MASP_HIP_LONE_GRAPH": "lone_proof_graph", "MASP_HIP_MSM_C_H_LONE": "window_bits_h_lone"}


def options_from_env(env=os.environ):
    """masp_hip_options fields from the MASP_HIP_* variables of bench.py / tools (the library itself reads none)."""
    return {field: int(env[name]) for name, field in ENV_OPTIONS.items() if env.get(name, "") != ""}


class ClockWatch:
    """Shader clock and socket power while a timed region runs: the amdgpu hwmon files of every card (freq1_input = sclk in Hz, power1_input =
    PPT in microwatts), read every 50 ms by a thread.  A box shows all of its cards whatever this process may use; the cards whose power
    rises with the region (>= 60 % of the busiest one's mean) are the ones it ran on.  No rocm-smi process, no HIP call."""

    def __init__(self, device=None):
        import glob
        self.cards = []
        # the card THIS process proves on, by PCI address (masp_hip_device_pci_bus_id against the sysfs device links):
        # a box shows all eight cards and another tenant's may be busy at the same time (round 6: a line averaged two cards' clocks)
        mine = None
        if device is not None:
            try:
                from masp_amd.hip import device_pci_bus_id
                mine = device_pci_bus_id(device)
            except Exception:                        # (a watch that cannot find its card watches them all; it never stops the bench)
                mine = None
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
            pw = os.path.join(os.path.dirname(f), "power1_input")
            if os.path.exists(pw):
                pci = os.path.basename(os.path.realpath(f.split("/hwmon/")[0])).lower()
                if mine is None or pci == mine:
                    self.cards.append((f, pw))
        self.by_pci = mine is not None and len(self.cards) == 1
        if mine is not None and not self.cards:      # (no such link: fall back to every card and the power heuristic)
            self.__init__(None)
        self.regions, self._stop, self._thread, self._cur = {}, None, None, None

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self, stop, samples):
        while not stop.is_set():
            samples.append([(self._read(f), self._read(p)) for f, p in self.cards])
            stop.wait(0.05)

    def start(self, region):
        if not self.cards:
            return
        self._stop, self._cur = threading.Event(), self.regions.setdefault(region, [])
        self._thread = threading.Thread(target=self._run, args=(self._stop, self._cur), daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None

    def summary(self, region):
        samples = self.regions.get(region) or []
        if not samples:
            return None
        per_card = []
        for c in range(len(self.cards)):
            fs = [s[c][0] for s in samples if s[c][0] is not None and s[c][1] is not None]
            ps = [s[c][1] for s in samples if s[c][0] is not None and s[c][1] is not None]
            if fs:
                per_card.append((sum(ps) / len(ps) * 1e-6, sum(fs) / len(fs) * 1e-6, min(fs) * 1e-6, max(fs) * 1e-6, len(fs)))
        if not per_card:
            return None
        top = max(w for w, *_ in per_card)
        busy = [r for r in per_card if r[0] >= 0.6 * top]
        return {"sclk_mhz_mean": round(sum(r[1] for r in busy) / len(busy), 1), "sclk_mhz_min": round(min(r[2] for r in busy), 1),
                "sclk_mhz_max": round(max(r[3] for r in busy), 1), "socket_power_w_mean": round(sum(r[0] for r in busy) / len(busy), 1),
                "samples_per_card": busy[0][4], "cards_busy": len(busy), "cards_seen": len(per_card), "card_by_pci_address": bool(getattr(self, "by_pci", False))}


def library_sha16():
    """sha256[:16] of the libmasp_hip.so this process loaded: tracked measurements that cannot run inside the bench (PMC passes) carry the
    hash of the build they were taken on, and the line says whether that is this build."""
    import hashlib
    from masp_amd.hip import library_path
    try:
        return hashlib.sha256(open(library_path(), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: become N ranks (one per GPU) under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a launcher: exec %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def cpu_baseline(cs, params, inputs, aux):
    """Oracle (C++ restatement of bellperson's CPU prover, oracle/) timed on the host cores: reported baseline only.
    BASELINE.md §3: median of 10 runs after 2 warm-ups (criterion's sample_size(10))."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from masp_amd.host import effective_cpus
    O.lib().oracle_set_threads(effective_cpus())     # the cores this process may actually use (cgroup quota), not nproc
    P = O.Params(params)
    times, phases = [], {}
    for n in range(12):
        tm = {}
        t0 = time.perf_counter()
        O.create_proof(P, cs, inputs, aux, 7 + n, 11 + n, timings=tm)
        dt = time.perf_counter() - t0
        if n >= 2:
            times.append(dt)
            for k, v in tm.items():
                phases.setdefault(k, []).append(v)
    med = statistics.median(times)
    return {"value": round(1.0 / med, 4), "unit": "proofs/s", "cores": O.lib().oracle_get_threads(), "kind": "port",
            "sample": "median of 10 %s proofs after 2 warm-ups (instance 0 of the GPU batch, same CRS), oracle on all usable host cores"
                      % (WORKLOAD if WORKLOAD != "mixed" else "spend"),
            "median_ms_per_proof": round(med * 1e3, 2), "phase_ms_per_proof": {k: round(statistics.median(v), 2) for k, v in phases.items()}}


def main_in_library(args):
    """The other way to all GPUs of a node: ONE process, no torch / RCCL — masp_hip_ctx_create_ex over devices
    0 .. N-1, every step one masp_hip_prove_batch of N x 256 jobs that the library deals to its devices (one host thread per
    device inside the library), `slots` such calls in flight.  Same witnesses-in-page-locked-host-memory region as `value` of the
    process-per-GPU mode; prints one line in the same format (launcher: "in-library")."""
    import masp_amd
    from masp_amd import host as H
    from masp_amd import synthetic
    from masp_amd import workload as W
    N, K, Wm = args.gpus, args.steps, args.warmup
    kind, n = "spend", PROOFS_PER_STEP["spend"]
    ctx = masp_amd.Context(list(range(N)), **options_from_env())
    assert ctx.device_count == N
    SLOTS = ctx.options["slots"]
    cs = H.circuit(kind)[0]
    t0 = time.perf_counter()
    params = ctx.generate_parameters(cs, synthetic.toxic_waste(1))
    ctx.load_circuit(0, params, cs)                      # replicated on every device
    setup_s = time.perf_counter() - t0
    vk = ctx.prepare_verifying_key(params)
    threads = H.effective_cpus()
    W.instances(kind, 2, first_seed=10 ** 6, threads=2)
    insts = W.instances(kind, N * n, first_seed=0, threads=threads, montgomery=True, alloc=lambda kk: ctx.host_alloc(cs.n_aux, 32))
    rng = random.Random(0x5962be3d)

    def jobs_for_step():
        return [(0, i, a, rng.randrange(R).to_bytes(32, "little"), rng.randrange(R).to_bytes(32, "little"), None, 1) for i, a in insts]

    # (set-up + warm-up: two rounds of `slots` concurrent calls, so that every slot of every device has its scratch at its final size)
    warm = [ctx.marshal_jobs(jobs_for_step()) for _ in range(max(Wm, 2 * SLOTS))]
    timed = [ctx.marshal_jobs(jobs_for_step()) for _ in range(K)]
    with ThreadPoolExecutor(SLOTS) as ex:
        list(ex.map(lambda k: ctx.prove_marshalled(warm[k][0], N * n), range(len(warm))))          # also sizes every device's scratch
    out = np.zeros((K, N * n, 192), np.uint8)
    ctx.sync()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(SLOTS) as ex:
        list(ex.map(lambda k: ctx.prove_marshalled(timed[k][0], N * n, out[k]), range(K)))           # exactly K steps
    ctx.sync()
    elapsed = time.perf_counter() - t0
    pub = [W.public_inputs(i) for i, _ in insts]
    for k in range(K):
        if not vk.verify_batch([out[k, j].tobytes() for j in range(N * n)], pub):
            sys.exit("bench.py: a timed proof FAILED the pairing check — no figure reported")
    assert len(set(p.tobytes() for p in out.reshape(-1, 192))) == K * N * n
    vk.close()
    ctx.close()
    print(json.dumps({"metric": "Spend proofs/sec", "value": K * N * n / elapsed, "unit": "proofs/s", "n_gpus": N, "steps": K, "warmup": Wm,
                      "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                      "launcher": "in-library: one process, masp_hip_ctx_create_ex over %d device(s), no torch / RCCL" % N,
                      "config": {"workload": "BASELINE.json configs[3] per GPU: %d distinct Spend proofs per GPU and step, one masp_hip_prove_batch of %d jobs per "
                                             "step dealt to the devices inside the library, %d calls in flight; witnesses in page-locked host memory -> proofs "
                                             "in host memory (BASELINE.md §4)" % (n, N * n, SLOTS),
                                 "proofs_per_step": N * n, "parallelism": "proofs dealt to %d device context(s) by the library, no collective" % N},
                      "verified": K * N * n, "setup_seconds": round(setup_s, 2)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8, help="timed steps; one step = one batch of 256 distinct proofs per GPU")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-library", action="store_true", help="one process over all --gpus devices through masp_hip_ctx_create_ex (no torch, no RCCL)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / process-group / gather path only, no GPU work, no figure (CPU test hook)")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        sys.exit("bench.py: --gpus and --steps must be >= 1, --warmup >= 0")
    if args.in_library:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        return main_in_library(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s): refusing to report a figure for a different GPU count" % (args.gpus, world))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # let independent batches' kernels overlap (ROCm default: 4; 16: profiles/r04e_hw_queues_and_the_two_modes.txt)
    dist = dev = None
    backend = os.environ.get("MASP_BENCH_BACKEND", "nccl")      # "nccl" = RCCL on ROCm; "gloo" only for the CPU dry run
    if world > 1 or os.environ.get("MASP_BENCH_FORCE_DIST"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # torch first: it brings its own HIP runtime; libmasp_hip then binds to the already loaded one
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            n_dev = torch.cuda.device_count()
            if n_dev < 1:
                sys.exit("bench.py: no GPU visible")
            if os.environ.get("LOCAL_WORLD_SIZE") and n_dev < int(os.environ["LOCAL_WORLD_SIZE"]) and n_dev != 1:
                sys.exit("bench.py: %s ranks on this node but only %d GPUs visible" % (os.environ["LOCAL_WORLD_SIZE"], n_dev))
            local_rank %= n_dev                      # a launcher may expose one device per rank (HIP_VISIBLE_DEVICES)
            torch.cuda.set_device(local_rank)
            dev = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            if not args.dry_run:
                # gloo with real GPU work (tests/test_gpu_bench_two_ranks.py: the N = 2 code path on a one-GPU box): ranks that outnumber the
                # devices share them
                from masp_amd.hip import device_count
                n_dev = device_count()
                if n_dev < 1:
                    sys.exit("bench.py: no GPU visible")
                if local_rank >= n_dev:
                    sys.stderr.write("bench.py: rank %d shares device %d (%d device(s) visible)\n" % (rank, local_rank % n_dev, n_dev))
                local_rank %= n_dev
        if rank == 0:
            sys.stderr.write("bench.py: %s process group up: %d rank(s), one per GPU\n" % ("RCCL" if backend == "nccl" else backend, dist.get_world_size()))
    if args.dry_run:
        # what the multi-GPU path adds to the single-GPU one, without a GPU: rank sharding + the final gather + one line from rank 0
        from masp_amd import distributed as D
        n_local = args.steps * PROOFS_PER_STEP[WORKLOAD]
        fake = np.zeros((n_local, 192), np.uint8)
        fake[:, 0] = rank
        fake[:, 1:5] = np.arange(n_local, dtype=np.uint32).view(np.uint8).reshape(n_local, 4)
        got = D.gather_proofs(fake, n_local * world, dist, dev)
        t = D.max_over_ranks(float(rank), dist, dev)
        # the CRS path of N ranks: rank 0's bytes reach every rank unchanged (here: a 5 MB stand-in)
        import hashlib
        blob = np.frombuffer(hashlib.sha256(b"crs").digest() * (5 * 2 ** 20 // 32), np.uint8) if rank == 0 else None
        crs = D.broadcast_bytes(blob, dist, dev)
        crs_ok = int(D.sum_over_ranks(float(hashlib.sha256(crs.tobytes()).hexdigest() == hashlib.sha256(hashlib.sha256(b"crs").digest() * (5 * 2 ** 20 // 32)).hexdigest()),
                                      dist, dev))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            ok = len(got) == n_local * world and all(got[r * n_local + i][0] == r and int.from_bytes(got[r * n_local + i][1:5], "little") == i
                                                     for r in range(world) for i in range(n_local))
            print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": world, "backend": backend, "gathered": len(got), "gather_ok": bool(ok),
                              "max_rank_seen": int(t), "crs_broadcast_ok_ranks": crs_ok, "value": None}), flush=True)
        return
    import masp_amd
    from masp_amd import distributed as D
    from masp_amd import host as H
    from masp_amd import synthetic
    from masp_amd import workload as W

    ctx = masp_amd.Context(local_rank, **options_from_env())
    SLOTS, BATCH = ctx.options["slots"], ctx.options["batch_cap"]
    ctx_tree_levels = {0: "4 (default)", -1: "no"}.get(ctx.options["bucket_tree_levels"], str(ctx.options["bucket_tree_levels"]))
    ctx_tree_sub = ctx.options["bucket_tree_sub_batch"]
    threads = max(1, H.effective_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    kinds = list(KINDS) if WORKLOAD == "mixed" else [WORKLOAD]
    n = PROOFS_PER_STEP[WORKLOAD]
    K, Wm = args.steps, args.warmup
    # ---- circuits: the real MASP R1CS (structure hashes pinned to the reference's KATs) + CRS from known toxic waste, same on every rank
    # (N ranks: rank 0 generates each CRS, the others receive the same bytes over the process group instead of running the
    # set-up kernels eight times side by side)
    cs, params, vk = {}, {}, {}
    t_setup = time.perf_counter()
    for kind in kinds:
        cs[kind] = H.circuit(kind)[0]
        mine = ctx.generate_parameters(cs[kind], synthetic.toxic_waste(1 + KINDS.index(kind))) if rank == 0 else None
        params[kind] = D.broadcast_bytes(mine, dist, dev)
        ctx.load_circuit(KINDS.index(kind), params[kind], cs[kind])
        vk[kind] = ctx.prepare_verifying_key(params[kind])      # Groth16 batch verifier, Miller loops on the GPU (product code)
    setup_s = time.perf_counter() - t_setup
    sys.stderr.write("bench.py: rank %d: CRS %s + window tables in %.1f s\n" % (rank, "generated" if rank == 0 else "received", setup_s))
    # ---- n distinct instances per rank, synthesised on the host cores before anything is timed, aux written to page-locked memory
    W.instances(kinds[0], 2, first_seed=10 ** 6, threads=2)          # one-time table construction of the synthesizer
    syn = {}
    job_kind = [kinds[j % len(kinds)] for j in range(n)]
    t_syn = time.perf_counter()
    # (aux as Montgomery residues — the in-memory form of blst_fr, masp_hip_job::aux_form = 1: the synthesizer writes straight into
    # the page-locked buffer; host.GROUP witnesses per native call, their Merkle blocks side by side)
    # the page-locked aux buffers of a circuit are one slab, locked before the synthesis threads start (page-locking from 16 threads
    # at once serialises on the runtime: it used to be timed as "synthesis")
    slabs = {k: ctx.host_alloc(cs[k].n_aux * job_kind.count(k), 32) for k in kinds}
    handed = {k: iter(range(job_kind.count(k))) for k in kinds}
    hand_lock = threading.Lock()

    def take(kk):
        with hand_lock:
            j = next(handed[kk])
        return slabs[kk][j * cs[kk].n_aux:(j + 1) * cs[kk].n_aux]
    per = {k: W.instances(k, job_kind.count(k), first_seed=100000 * rank, threads=threads, timing=syn, montgomery=True, alloc=take) for k in kinds}
    synth_wall = time.perf_counter() - t_syn
    it = {k: iter(per[k]) for k in kinds}
    insts = [next(it[k]) for k in job_kind]
    assert len(set(a.tobytes() for _, a in insts)) == n, "instances are not distinct"
    rng = random.Random(0x5962be3d + rank)                            # (the reference bench's XorShift seed bytes, benches/sapling.rs:19-22)

    def fresh_rs(steps, rng=rng):
        b = bytearray()
        for _ in range(2 * steps * n):
            b += rng.randrange(R).to_bytes(32, "little")
        return np.frombuffer(bytes(b), np.uint8).reshape(steps, n, 64)

    def jobs_with(rs_step):
        return [(KINDS.index(k), i, a, bytes(rs_step[j, :32]), bytes(rs_step[j, 32:]), None, 1) for j, (k, (i, a)) in enumerate(zip(job_kind, insts))]

    rs_warm, rs_a, rs_b = fresh_rs(max(Wm, 1)), fresh_rs(K), fresh_rs(K)
    handle, _ = ctx.batch_upload(jobs_with(rs_a[0]))
    # set-up, not warm-up: every slot's workspace (hipMalloc on first use) gets its final size.  The library deals a call's groups
    # (<= batch_cap proofs of one circuit) to the slots round robin, starting at slot 0 in every call: group g of step s goes to slot
    # (s x groups_per_step + g) mod SLOTS, a pattern that repeats after SLOTS / gcd steps — so that many steps show every slot every
    # circuit it will ever get (three circuits on four slots: 4 steps, not 2 — with 2, two slots met their first Spend batch, and the
    # 40 GB hipMalloc that goes with it, inside the timed region: `resident` of the mixed workload read 1 522 instead of ~ 2 000)
    groups_per_step = sum(-(-job_kind.count(k) // BATCH) for k in kinds)
    sizing_steps = max(2, SLOTS // math.gcd(groups_per_step, SLOTS))
    ctx.batch_prove_resident_steps(handle, n, sizing_steps, fresh_rs(sizing_steps))
    if Wm > 0:
        ctx.batch_prove_resident_steps(handle, n, Wm, rs_warm)
    marshalled = [ctx.marshal_jobs(jobs_with(rs_b[k])) for k in range(K)]
    h2h_calls = int(os.environ.get("MASP_BENCH_H2H_CALLS", SLOTS))
    # the W warm-up steps of the region `value` is measured on: the same call pattern (h2h_calls masp_hip_prove_batch calls in
    # flight), which also gives the host path's staging buffers their size
    warm_b = [ctx.marshal_jobs(jobs_with(rs_warm[k])) for k in range(max(Wm, 1))]
    with ThreadPoolExecutor(h2h_calls) as ex:
        list(ex.map(lambda k: ctx.prove_marshalled(warm_b[k][0], n), range(len(warm_b))))
    # single-proof latency (not the headline value; MASP_BENCH_LONE=0 skips it: the tests that compare bench lines)
    LONE = os.environ.get("MASP_BENCH_LONE", "1") != "0"
    one = None
    latency_ms = latency_host_ms = None
    lone_graphs = 0

    def lone_latency(job):
        """One job as a caller of masp_hip_prove_batch sees it: witness in page-locked host memory -> proof bytes in host memory; median of 8
        after 4 calls (MASP_HIP_LONE_GRAPH=1: replayed from a captured launch graph from the third call on — masp_hip_options::lone_proof_graph)"""
        one_m = ctx.marshal_jobs([job])
        lat_h = []
        for _ in range(12):
            t0 = time.perf_counter()
            ctx.prove_marshalled(one_m[0], 1)
            lat_h.append((time.perf_counter() - t0) * 1e3)
        return sorted(lat_h[4:])[len(lat_h[4:]) // 2]
    if LONE:
        one, _ = ctx.batch_upload(jobs_with(rs_warm[0])[:1])
        ctx.batch_prove_resident(one, 1)              # sizes the lone-proof workspace
        lat = []
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.batch_prove_resident(one, 1)
            lat.append((time.perf_counter() - t0) * 1e3)
        latency_ms = sorted(lat)[len(lat) // 2]
        latency_host_ms = lone_latency(jobs_with(rs_warm[0])[0])
        lone_graphs = ctx.lone_graph_launches()

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # ---- region A (`resident`): witnesses resident in HBM -> K steps -> proofs gathered on rank 0
    ctx.profile_enable(True)
    clocks = ClockWatch(local_rank)
    barrier()
    clocks.start("resident")
    t0 = time.perf_counter()
    proofs_a, gpu_ms = ctx.batch_prove_resident_steps(handle, n, K, rs_a)       # exactly K steps
    gathered = D.gather_proofs(proofs_a.reshape(K * n, 192), K * n * world, dist, dev) if dist is not None else proofs_a.reshape(K * n, 192)
    barrier()
    elapsed = time.perf_counter() - t0
    clocks.stop()
    acc_ms, launches, alg_bytes = ctx.profile_read()
    split_a = ctx.profile_read_split()
    # The same launches with ONE batch in flight (one launch sequence of the first circuit's jobs, nothing else on the chip):
    # with four batches in flight the HIP events around a launch also span the time it waits for the chip behind other
    # streams' kernels, which is not kernel time (rocprofv3's begin / end of the same launches agree with THIS figure).
    first = [j for j in range(n) if job_kind[j] == kinds[0]][:BATCH]
    excl, n_excl = ctx.batch_upload([jobs_with(rs_warm[0])[j] for j in first])
    ctx.sync()
    ctx.batch_prove_resident(excl, n_excl)
    x_ms, x_launches, x_bytes = ctx.profile_read()          # cumulative since profile_enable
    x_ms, x_launches, x_bytes = x_ms - acc_ms, x_launches - launches, x_bytes - alg_bytes
    x_split = {k: v - split_a[k] for k, v in ctx.profile_read_split().items()}
    ctx.batch_free(excl)
    ctx.profile_enable(False)
    # ---- region B (`value`): witnesses in page-locked host memory -> K masp_hip_prove_batch calls (one per slot
    # in flight) -> proofs on the host of rank 0 (H2D, D2H and the RCCL gather inside)
    out_b = np.zeros((K, n, 192), np.uint8)
    barrier()
    clocks.start("value")
    t0 = time.perf_counter()
    with ThreadPoolExecutor(h2h_calls) as ex:
        list(ex.map(lambda k: ctx.prove_marshalled(marshalled[k][0], n, out_b[k]), range(K)))          # exactly K steps
    gathered_b = D.gather_proofs(out_b.reshape(K * n, 192), K * n * world, dist, dev) if dist is not None else out_b.reshape(K * n, 192)
    barrier()
    elapsed_b = time.perf_counter() - t0
    clocks.stop()
    if dist is not None:
        elapsed = D.max_over_ranks(elapsed, dist, dev)
        elapsed_b = D.max_over_ranks(elapsed_b, dist, dev)
        if rank == 0:
            assert len(gathered) == K * n * world and len(gathered_b) == K * n * world
    # ---- verification of what was timed (product code: GPU + host Groth16 batch verifiers; plus oracle closed form on a sample)
    pub = [W.public_inputs(i) for i, _ in insts]

    def verify_chunk(args_):
        proofs, step, kind = args_
        sel = [j for j in range(n) if job_kind[j] == kind]
        return len(sel) if vk[kind].verify_batch([proofs[step, j].tobytes() for j in sel], [pub[j] for j in sel]) else -10 ** 9
    t_ver = time.perf_counter()
    chunks = [(p, st, kind) for p in (proofs_a, out_b) for st in range(K) for kind in kinds]
    counts = [verify_chunk(c) for c in chunks]
    # ... and the host verifier (independent code path: libmasp_host's own Miller loop) on the first step of region A
    hvk = {kind: H.PreparedVerifyingKey(params[kind]) for kind in kinds}
    for kind in kinds:
        sel = [j for j in range(n) if job_kind[j] == kind][:64]
        if not hvk[kind].verify_batch([proofs_a[0, j].tobytes() for j in sel], [pub[j] for j in sel]):
            counts.append(-10 ** 9)
    if min(counts) < 0:
        sys.exit("bench.py: a timed proof FAILED the pairing check — no figure reported")
    verified_a = sum(c for c, ch in zip(counts, chunks) if ch[0] is proofs_a)
    verified_b = sum(c for c, ch in zip(counts, chunks) if ch[0] is out_b)
    assert verified_a == K * n and verified_b == K * n
    assert len(set(p.tobytes() for p in proofs_a.reshape(-1, 192))) == K * n
    verify_s = time.perf_counter() - t_ver
    closed_ok = 0
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        for got_, rs_, st, j in ((out_b, rs_b, 0, 0), (out_b, rs_b, K - 1, n - 1), (proofs_a, rs_a, K // 2, n // 2), (proofs_a, rs_a, 0, 1)):
            kind = job_kind[j]
            want = O.closed_form_proof(cs[kind], synthetic.toxic_waste(1 + KINDS.index(kind)), insts[j][0], H.aux_from_montgomery(insts[j][1]),
                                       int.from_bytes(rs_[st, j, :32].tobytes(), "little"), int.from_bytes(rs_[st, j, 32:].tobytes(), "little"))
            if got_[st, j].tobytes() != want:
                sys.exit("bench.py: timed proof (step %d, job %d) differs from the oracle's closed form — no figure reported" % (st, j))
            closed_ok += 1
    verified_total = int(D.sum_over_ranks(float(verified_b), dist, dev))      # every rank verified all of its own proofs (both regions)
    # ---- what the gather delivered: rank 0 checks EVERY gathered proof of EVERY rank (both regions) with the batch verifier against that
    # rank's statements — instances and blinding scalars are seeded by rank, so rank 0 re-derives them — and compares two proofs per rank
    # byte for byte with the oracle's closed form at their job positions (a gather that permuted, truncated or zero-filled a shard fails here)
    gathered_checked = None
    if dist is not None and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        t_g = time.perf_counter()
        gathered_checked = {"ranks": world, "proofs_verified": 0, "closed_form_equal": 0}
        for r in range(world):
            if r == 0:
                insts_r, rs_a_r, rs_b_r = [(i, H.aux_from_montgomery(a)) for i, a in (insts[0], insts[n - 1])], rs_a, rs_b
                pub_r = pub
            else:
                per_r = {k: W.instances(k, job_kind.count(k), first_seed=100000 * r, threads=threads) for k in kinds}
                it_r = {k: iter(per_r[k]) for k in kinds}
                all_r = [next(it_r[k]) for k in job_kind]
                pub_r = [W.public_inputs(i) for i, _ in all_r]
                insts_r = [all_r[0], all_r[n - 1]]
                rng_r = random.Random(0x5962be3d + r)
                _, rs_a_r, rs_b_r = fresh_rs(max(Wm, 1), rng_r), fresh_rs(K, rng_r), fresh_rs(K, rng_r)
            for gl in (gathered, gathered_b):
                block = gl[r * K * n:(r + 1) * K * n]
                for st in range(K):
                    for kind in kinds:
                        sel = [j for j in range(n) if job_kind[j] == kind]
                        if not vk[kind].verify_batch([bytes(block[st * n + j]) for j in sel], [pub_r[j] for j in sel]):
                            sys.exit("bench.py: a gathered proof of rank %d (step %d) FAILED the pairing check — no figure reported" % (r, st))
                        gathered_checked["proofs_verified"] += len(sel)
            for gl, rs_, st, j, (inp, aux_c) in ((gathered_b, rs_b_r, 0, 0, insts_r[0]), (gathered, rs_a_r, K - 1, n - 1, insts_r[1])):
                kind = job_kind[j]
                want = O.closed_form_proof(cs[kind], synthetic.toxic_waste(1 + KINDS.index(kind)), inp, aux_c,
                                           int.from_bytes(rs_[st, j, :32].tobytes(), "little"), int.from_bytes(rs_[st, j, 32:].tobytes(), "little"))
                if bytes(gl[r * K * n + st * n + j]) != want:
                    sys.exit("bench.py: gathered proof (rank %d, step %d, job %d) differs from the oracle's closed form — no figure reported" % (r, st, j))
                gathered_checked["closed_form_equal"] += 1
        gathered_checked["seconds"] = round(time.perf_counter() - t_g, 2)
    # ---- end to end (not part of `value`): LocalTxProver.prove_batch over E2E_N Spend descriptions per GPU — synthesis on this
    # rank's share of the host cores, page-locked buffers, H2D, GPU batches, GPU batch self-verification (sapling/prover.rs:148)
    e2e = None
    other = None            # CRS of Output and Convert (the end-to-end prover and the other circuits' short regions need them)
    circuits_loaded = False
    # (the aux buffers are page-locked memory of `ctx`: gone once it closes; the checker wants canonical values)
    base_instance = (per[kinds[0]][0][0].copy(), H.aux_from_montgomery(per[kinds[0]][0][1]))
    e2e_n = int(os.environ.get("MASP_BENCH_E2E", str(K * n)))          # as many proofs as the timed regions of `value` / `resident`
    if WORKLOAD == "spend" and e2e_n > 0:
        for k_ in vk.values():
            k_.close()
        vk = {}
        other = [ctx.generate_parameters(H.circuit(k)[0], synthetic.toxic_waste(1 + KINDS.index(k))) for k in ("output", "convert")]
        # the end-to-end prover works on THIS process's context (one per device: its slots' scratch is already sized).  A second
        # context created after this one had been used and closed ran the same call 4 - 9 % slower — the resident and the host-to-host
        # paths on it did not: profiles/r04e_second_context_in_a_process.txt —, which is not what a prover process looks like
        for h_ in (handle, one):
            if h_ is not None:
                ctx.batch_free(h_)
        from masp_amd.prover import LocalTxProver
        if os.environ.get("MASP_BENCH_GC_FREEZE", "1") != "0":
            # what the earlier regions left on the Python heap (job tuples, marshalled arrays, 10 000 proofs) is garbage-collected
            # here and the survivors are frozen: a generation-2 collection in the middle of prove_batch holds the GIL of its chunk threads
            import gc
            del marshalled, warm_b
            gc.collect()
            gc.freeze()
        t_e = time.perf_counter()
        prover = LocalTxProver(params["spend"], other[0], other[1], device=local_rank, expected=None, context=ctx)
        e2e_setup = time.perf_counter() - t_e
        with ThreadPoolExecutor(threads) as ex:
            descs = list(ex.map(lambda i: W.description("spend", 5 * 10 ** 6 + 10 ** 5 * rank + i), range(e2e_n)))
        # set-up, not measurement: LocalTxProver.warm_up — the page-locked aux pool a call over e2e_n descriptions keeps in flight (page-locking
        # is slow while the GPU is busy: 17 ms per 3 MB buffer) and one zero-witness launch sequence per slot (the scratch of this context is
        # sized already; on a fresh prover that is where the first hipMallocs go).  Then the call twice: `first_call` is the first prove_batch
        # of this prover, `value` the second
        t_e = time.perf_counter()
        prover.warm_up(spends=e2e_n, threads=threads)
        warm_up_s = time.perf_counter() - t_e
        if dist is not None:
            dist.barrier()
        t_e = time.perf_counter()
        res0 = prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=threads)
        first_s = D.max_over_ranks(time.perf_counter() - t_e, dist, dev)
        assert len(res0) == e2e_n
        del res0
        if dist is not None:
            dist.barrier()
        t_e = time.perf_counter()
        res = prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=threads)
        e2e_s = time.perf_counter() - t_e
        assert len(res) == e2e_n and len(set(r[0] for r in res)) == e2e_n
        e2e_opt = prover._ctx.current_options()   # (what lack of tree scratch changed, if anything: sub-batch in use, proofs through the XYZZ fallback)
        prover.close()
        circuits_loaded = True            # (the context is ours: the three circuits stay loaded in it)
        e2e_s = D.max_over_ranks(e2e_s, dist, dev)
        e2e = {"value": round(e2e_n * world / e2e_s, 2), "unit": "proofs/s", "descriptions_per_gpu": e2e_n, "seconds": round(e2e_s, 3), "threads_per_gpu": threads,
               "load_seconds": round(e2e_setup, 2), "warm_up_seconds": round(warm_up_s, 2),
               "first_call": {"value": round(e2e_n * world / first_s, 2), "seconds": round(first_s, 3), "of_warm": round(e2e_s / first_s, 3) if first_s else None},
               "bucket_tree_sub_batch_in_use": e2e_opt["bucket_tree_sub_batch"],
               "bucket_tree_fallback_proofs": e2e_opt["bucket_tree_fallback_proofs"]}
    # ---- the other two circuits (BASELINE.json configs[0] / configs[2]: masp_proofs/benches/convert.rs:31-66; the reference has no Output
    # bench): a short region shaped like `value` (OTHER_STEPS steps of 256 distinct proofs per GPU, witnesses in page-locked host memory ->
    # proofs in host memory, every proof verified) and the lone-proof latency of each.  Not the metric; driver-timed all the same.
    others = None
    if WORKLOAD == "spend" and os.environ.get("MASP_BENCH_OTHER", "1") != "0":
        others = {}
        OTHER_STEPS = 4
        if other is None:
            other = [ctx.generate_parameters(H.circuit(k)[0], synthetic.toxic_waste(1 + KINDS.index(k))) for k in ("output", "convert")]
        for oi, kind in enumerate(("output", "convert")):
            slot_k, cs_k = KINDS.index(kind), H.circuit(kind)[0]
            if not circuits_loaded:
                ctx.load_circuit(slot_k, other[oi], cs_k)
            vk_k = ctx.prepare_verifying_key(other[oi])
            slab_k, next_k = ctx.host_alloc(cs_k.n_aux * 256, 32), iter(range(256))

            def take_k(_kind, slab_k=slab_k, next_k=next_k, n_aux=cs_k.n_aux):
                with hand_lock:
                    j = next(next_k)
                return slab_k[j * n_aux:(j + 1) * n_aux]
            inst_k = W.instances(kind, 256, first_seed=100000 * rank, threads=threads, montgomery=True, alloc=take_k)
            rs_k = fresh_rs(3 + OTHER_STEPS)[:, :256]
            m_k = [ctx.marshal_jobs([(slot_k, i, a, bytes(rs_k[st, j, :32]), bytes(rs_k[st, j, 32:]), None, 1) for j, (i, a) in enumerate(inst_k)])
                   for st in range(3 + OTHER_STEPS)]
            with ThreadPoolExecutor(h2h_calls) as ex:                      # warm-up: the same call pattern
                list(ex.map(lambda k: ctx.prove_marshalled(m_k[k][0], 256), range(3)))
            out_k = np.zeros((OTHER_STEPS, 256, 192), np.uint8)
            barrier()
            t0 = time.perf_counter()
            with ThreadPoolExecutor(h2h_calls) as ex:
                list(ex.map(lambda k: ctx.prove_marshalled(m_k[3 + k][0], 256, out_k[k]), range(OTHER_STEPS)))
            barrier()
            el_k = D.max_over_ranks(time.perf_counter() - t0, dist, dev)
            pub_k = [W.public_inputs(i) for i, _ in inst_k]
            for st in range(OTHER_STEPS):
                if not vk_k.verify_batch([out_k[st, j].tobytes() for j in range(256)], pub_k):
                    sys.exit("bench.py: a timed %s proof FAILED the pairing check — no figure reported" % kind)
            lat_k = None
            if LONE:
                i0, a0 = inst_k[0]
                lat_k = lone_latency((slot_k, i0, a0, bytes(rs_k[0, 0, :32]), bytes(rs_k[0, 0, 32:]), None, 1))
            vk_k.close()
            others[kind] = {"value": round(OTHER_STEPS * 256 * world / el_k, 2), "unit": "proofs/s", "steps": OTHER_STEPS, "proofs_per_step": 256,
                            "ms_per_step": round(el_k * 1e3 / OTHER_STEPS, 3), "verified": OTHER_STEPS * 256,
                            "single_proof_latency_ms": None if lat_k is None else round(lat_k, 3), "constraints": cs_k.n_constraints}
    if rank == 0:
        total = K * n * world
        achieved = x_bytes / (x_ms * 1e-3) / 1e9 if x_ms > 0 else 0.0
        lib_sha = library_sha16()
        # HBM bytes per launch of the same stage: NOT measured by this run (PMC counters need rocprofv3 passes of their own) — read from the
        # tracked rocprofv3 result (tools/pmc_traffic.sh), with the build it was measured on next to it
        traffic = traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                doc = json.load(open(pmc))
                traffic = doc.get("hbm_bytes_per_launch")
                traffic_source = {"file": "profiles/pmc_traffic.json", "date": doc.get("date"), "library_sha16": doc.get("library_sha16"),
                                  "same_build": doc.get("library_sha16") == lib_sha}
            except Exception:
                traffic = traffic_source = None
        # ---- the binding roofline next to the contractual one: VALU issue (BENCH_NOTES.md#roofline_valu).  The instruction count is a tracked
        # rocprofv3 --pmc SQ_INSTS_VALU measurement (tools/valu_model.sh) stamped with the library it was taken on; the clock is this run's
        roofline_valu = None
        ck_val, ck_res = clocks.summary("value"), clocks.summary("resident")
        vm_path = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r06_valu_model.json", "r05_valu_model.json")) if os.path.exists(q)), None)
        if WORKLOAD == "spend" and vm_path and ck_val is not None:
            try:
                vm = json.load(open(vm_path))
                insts, cpi = vm["valu_insts_per_batch"] * n / 256.0, vm["cycles_per_inst_weighted"]

                def issue_frac(ms_step, ck):
                    return insts * cpi / (1024 * ck["sclk_mhz_mean"] * 1e6 * ms_step * 1e-3)
                peak = 1024 * ck_val["sclk_mhz_mean"] * 1e6 / cpi
                roofline_valu = {"bound": "valu_issue", "achieved": round(insts / (elapsed_b / K) / 1e9, 2), "peak": round(peak / 1e9, 2),
                                 "unit": "G wave-instructions/s per GPU", "frac": round(issue_frac(elapsed_b * 1e3 / K, ck_val), 4),
                                 "frac_resident_region": round(issue_frac(elapsed * 1e3 / K, ck_res), 4) if ck_res else None,
                                 "valu_insts_per_step_per_gpu": round(insts), "cycles_per_inst": round(cpi, 4), "simds": 1024,
                                 "sclk_mhz": ck_val["sclk_mhz_mean"], "sclk_boost_mhz": 2400.0, "clock_frac": round(ck_val["sclk_mhz_mean"] / 2400.0, 4),
                                 "socket_power_w": ck_val["socket_power_w_mean"], "model": os.path.relpath(vm_path, ROOT),
                                 #  the count was measured on the build whose hash the model file carries: a kernel change since then
                                 # makes it stale, and the line says so instead of presenting an old count at a new clock
                                 "model_library_sha16": vm.get("library_sha16"), "model_stale": vm.get("library_sha16") != lib_sha}
            except Exception as e:
                roofline_valu = {"error": repr(e)}
        c0 = cs[kinds[0]]
        sh = synthetic.SHAPES[kinds[0]]
        config_name = {"spend": "BASELINE.json configs[3]: batch of 256 distinct Spend proofs per step on one MI355X (throughput mode)",
                       "mixed": "BASELINE.json configs[4]'s job mix: 512 jobs per GPU and step, job j of circuit j mod 3"}.get(
                           WORKLOAD, "batch of 256 distinct %s proofs per step" % WORKLOAD)

        def r3(x):
            return None if x is None else round(x, 3)
        # witnesses per second THIS rank's share of the host cores synthesises, against the proofs per second its GPU proves: below 1 the
        # end-to-end region of an N-rank run is bound by the host, not by the GPU (`value` is not: witnesses exist before timing starts)
        syn_rate = sum(v["instances"] for v in syn.values()) / max(sum(v["synthesize_wall_s"] for v in syn.values()), 1e-9)
        per_gpu = total / elapsed_b / world
        host_bound = bool(syn_rate < per_gpu)
        if e2e is not None:
            e2e["host_bound"] = host_bound
        # The ONE line: numbers only, in <= 4.5 KB — what each key means, how it was measured and what was verified is in BENCH_NOTES.md
        # (the driver's record keeps the head and the tail of the line; two thirds of the 7.7 KB one were prose and its
        # middle — lone latencies, end_to_end, other_circuits — was lost).  Contract keys first; end_to_end / other_circuits / lone last.
        out = {
            "metric": "Spend proofs/sec" if WORKLOAD == "spend" else "%s proofs/sec" % WORKLOAD, "value": round(total / elapsed_b, 2), "unit": "proofs/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": r3(elapsed_b * 1e3 / K), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": config_name, "proofs_per_step": n, "proofs_per_gpu": K * n, "distinct_witnesses_per_gpu": n,
                       "ntt_log2": c0.logm, "g1_msm_points": [(1 << c0.logm) - 1, c0.n_aux, sh[3] + c0.n_inputs, sh[4] + 1], "g2_msm_points": sh[4] + 1,
                       "batch_cap": BATCH, "slots": SLOTS, "calls_in_flight": h2h_calls, "hw_queues": ctx.options.get("hw_queues"),
                       "parallelism": "proofs sharded over %d GPU(s), no data-path collective, gather of the proofs" % world},
            "notes": "BENCH_NOTES.md", "library_sha16": lib_sha,
            "roofline": {"bound": "hbm", "kernel": "G1 bucket-accumulation stage of one G1 MSM of a batch (bucket tree %s levels, sub-batches of %d, then "
                                                   "k_msm_accumulate_pts)" % (ctx_tree_levels, ctx_tree_sub),
                         "achieved": r3(achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_source, "launches": x_launches,
                         "avg_launch_ms": r3(x_ms / x_launches) if x_launches else None,
                         "alg_bytes_per_launch": x_bytes / x_launches if x_launches else None,
                         "kernel_ms_per_launch": {k: round(v / x_launches, 3) for k, v in x_split.items()} if x_launches else None,
                         "timed_region": {"launches": launches, "avg_span_ms": r3(acc_ms / launches) if launches else None}},
            "roofline_valu": roofline_valu,
        }
        if not args.no_cpu_baseline and world == 1:
            k0 = kinds[0]
            out["cpu_baseline"] = cpu_baseline(cs[k0], params[k0], base_instance[0], base_instance[1])
        out.update({
            "verified": verified_total, "closed_form_equal": closed_ok, "verify_seconds": round(verify_s, 2),
            "rccl_ranks": dist.get_world_size() if dist is not None else 1,
            "collectives": "rccl" if dist is not None and backend == "nccl" else backend if dist is not None else "none",
            "collective_calls": D.collective_counts() if dist is not None else None,
            "collective_tensors": ("cuda:%d" % local_rank if dev is not None else "cpu") if dist is not None else None,
            "gathered_checked": gathered_checked,
            "ms_per_proof": round(elapsed_b * 1e3 / (K * n), 4),
            "setup_seconds_rank0": round(setup_s, 2),
            "clocks": {"value_region": ck_val, "resident_region": ck_res},
            "host_synthesis": {"ms_per_proof_one_thread": {k: round(statistics.median(v["synthesize_ms"]), 2) for k, v in syn.items()},
                               "threads": threads, "witnesses_per_native_call": H.GROUP, "witnesses_per_s_per_rank": round(syn_rate, 1),
                               "proofs_per_s_per_gpu": round(per_gpu, 1), "host_bound": host_bound,
                               "descriptions_per_s_all_threads": round(sum(v["instances"] for v in syn.values()) / max(sum(v["describe_wall_s"] for v in syn.values()), 1e-9), 1)},
            "resident": {"value": round(total / elapsed, 2), "unit": "proofs/s", "ms_per_step": r3(elapsed * 1e3 / K), "gpu_event_ms_per_step": r3(gpu_ms / K)},
            "other_circuits": others,
            "single_proof_latency": {"host_to_host_ms": r3(latency_host_ms), "resident_witness_ms": r3(latency_ms), "graph_replays": lone_graphs},
            "single_proof_latency_ms": r3(latency_host_ms),
            "end_to_end": e2e,
        })
    else:
        out = None
    for k_ in vk.values():
        k_.close()
    if ctx is not None:
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

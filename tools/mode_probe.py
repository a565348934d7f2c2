#!/usr/bin/env python3
"""A bench process lands in one of two modes ~2.5 % apart (profiles/r04_same_box_ab_quad_kernels_in_their_own_unit.txt).  Is the mode a
property of the process, of the context (its allocations, its streams), or of time?  One process: the CRS once, then REPS times
{new context, load the Spend circuit, upload 256 witnesses, size the workspaces, three timed runs of 4 resident steps}.
usage: mode_probe.py [reps=5]"""
import random
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import masp_amd
from masp_amd import host as H
from masp_amd import synthetic
from masp_amd import workload as W
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = 256
cs = H.circuit("spend")[0]
ctx = masp_amd.Context(0)
params = ctx.generate_parameters(cs, synthetic.toxic_waste(1))
ctx.close()
insts = W.instances("spend", n, first_seed=0, montgomery=True)
rng = random.Random(1)


def fresh_rs(steps):
    b = bytearray()
    for _ in range(2 * steps * n):
        b += rng.randrange(R).to_bytes(32, "little")
    return np.frombuffer(bytes(b), np.uint8).reshape(steps, n, 64)


for rep in range(reps):
    ctx = masp_amd.Context(0)
    ctx.load_circuit(0, params, cs)
    rs0 = fresh_rs(1)
    jobs = [(0, i, a, bytes(rs0[0, j, :32]), bytes(rs0[0, j, 32:]), None, 1) for j, (i, a) in enumerate(insts)]
    handle, _ = ctx.batch_upload(jobs)
    ctx.batch_prove_resident_steps(handle, n, 3, fresh_rs(3))
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        _, ms = ctx.batch_prove_resident_steps(handle, n, 4, fresh_rs(4))
        out.append("%.2f" % (ms / 4))
    print("context %d: gpu ms per step %s" % (rep, " ".join(out)), flush=True)
    ctx.close()

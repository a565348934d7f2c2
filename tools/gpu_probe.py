"""Times individual C-ABI calls on the GPU box; every line is flushed so a timeout still leaves a trail."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import masp_amd  # noqa: E402
import oracle_lib as O  # noqa: E402
import toy_r1cs  # noqa: E402

T0 = time.time()


def log(*a):
    print("[%7.2fs]" % (time.time() - T0), *a, flush=True)


def timed(label, fn):
    t = time.time()
    r = fn()
    log("%-40s %8.3f s" % (label, time.time() - t))
    return r


ctx = timed("ctx create", lambda: masp_amd.Context(0))
rng = np.random.default_rng(1)


def rand_scalars(n):
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    s[:, 31] &= 0x3f
    return s


for logm in (4, 10, 17):
    d = rand_scalars(1 << logm)
    timed("gpu ntt 2^%d" % logm, lambda: ctx.ntt(d, logm))
    timed("gpu ntt 2^%d again" % logm, lambda: ctx.ntt(d, logm))
    timed("oracle ntt 2^%d" % logm, lambda: O.ntt(d, logm))
for n in (40, 5000, 40000):
    ks = rand_scalars(n)
    bases = timed("oracle g1_mul_gen_many %d" % n, lambda: O.g1_mul_gen_many(ks))
    sc = rand_scalars(n)
    g = timed("gpu msm_g1 %d" % n, lambda: ctx.msm_g1(bases, sc))
    g2 = timed("gpu msm_g1 %d again" % n, lambda: ctx.msm_g1(bases, sc))
    o = timed("oracle msm_g1 %d" % n, lambda: O.msm_g1(bases, sc))
    log("match:", g == o, g2 == o)
for n in (33, 2000):
    ks = rand_scalars(n)
    bases = timed("oracle g2_mul_gen_many %d" % n, lambda: O.g2_mul_gen_many(ks))
    sc = rand_scalars(n)
    g = timed("gpu msm_g2 %d" % n, lambda: ctx.msm_g2(bases, sc))
    o = timed("oracle msm_g2 %d" % n, lambda: O.msm_g2(bases, sc))
    log("match:", g == o)
cs, inputs, aux, vals = toy_r1cs.make(33, 8, 300, 3000, bool_share=0.7)
pbuf = timed("oracle generate_parameters", lambda: O.generate_parameters(cs, toy_r1cs.toxic(33)))
timed("gpu load_circuit", lambda: ctx.load_circuit(3, pbuf, cs))
p = timed("gpu prove", lambda: ctx.prove(3, inputs, aux, 5, 6))
p2 = timed("gpu prove again", lambda: ctx.prove(3, inputs, aux, 5, 6))
o = timed("oracle prove", lambda: O.create_proof(O.Params(pbuf), cs, inputs, aux, 5, 6))
log("match:", p == o, p2 == o)

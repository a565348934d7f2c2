"""The lone b_g2 MSM on its own (nothing else on the chip): masp_hip_msm_g2_multi with ONE scalar vector over the Spend CRS's b_g2 points on 8-bit
windows — the launch sequence a lone proof's B2 chain runs.  Under `rocprofv3 --kernel-trace --stats` the kernels' durations in isolation."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import masp_amd
from masp_amd import host as H, synthetic
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
ctx = masp_amd.Context(0)
cs = H.circuit("spend")[0]
params = np.asarray(ctx.generate_parameters(cs, synthetic.toxic_waste(1)))
off = 96 + 96 + 192 + 192 + 96 + 192
for size in (96, 96, 96, 96, 96):
    off += 4 + int.from_bytes(params[off:off + 4].tobytes(), "big") * size
n = int.from_bytes(params[off:off + 4].tobytes(), "big")
bases = np.ascontiguousarray(params[off + 4:off + 4 + 192 * n].reshape(n, 192))
rng = random.Random(5)
sc = np.zeros((1, n, 32), np.uint8)
for i in range(n):
    u = rng.random()
    if u < 0.33:
        sc[0, i, 0] = 1
    elif u > 0.70:
        sc[0, i] = np.frombuffer(rng.randrange(R).to_bytes(32, "little"), np.uint8)
for w in (int(a) for a in (sys.argv[1:] or ["8"])):
    ctx.msm_g2_multi(bases, sc, window_bits=w)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.msm_g2_multi(bases, sc, window_bits=w)
        t.append((time.perf_counter() - t0) * 1e3)
    print("b_g2 MSM, %d points, %d-bit windows, one scalar vector: %.2f ms per call (includes the table build: see the kernel trace)" % (n, w, sorted(t)[2]), flush=True)
ctx.close()

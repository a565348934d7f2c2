"""The bucket stage of one 256-proof Spend batch by kernel group (masp_hip_profile_read_split), nothing else on the chip, no verification:
a diagnostic for A/B builds whose proofs may be wrong on purpose.  Options from the MASP_HIP_* environment (bench.py's names)."""
import os, sys, random
sys.path.insert(0, os.getcwd())
import numpy as np
import masp_amd
from bench import options_from_env
from masp_amd import host as H, synthetic, workload as W
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
kind = os.environ.get("PROBE_CIRCUIT", "spend")
cs = H.circuit(kind)[0]
insts = W.instances(kind, 256, first_seed=0, montgomery=True)
rng = random.Random(1)
c = masp_amd.Context(0, slots=1, **options_from_env())
params = c.generate_parameters(cs, synthetic.toxic_waste(1))
c.load_circuit(0, params, cs)
rs = np.frombuffer(b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(2 * 256)), np.uint8).reshape(256, 64)
jobs = [(0, i, a, bytes(rs[j, :32]), bytes(rs[j, 32:]), None, 1) for j, (i, a) in enumerate(insts)]
h, n = c.batch_upload(jobs)
c.batch_prove_resident(h, n)
c.sync()
c.profile_enable(True)
a = c.profile_read_split()
ms0, l0, _ = c.profile_read()
for _ in range(2):
    c.batch_prove_resident(h, n)
c.sync()
b = c.profile_read_split()
ms1, l1, _ = c.profile_read()
print(" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("MASP_HIP_")) or "defaults",
      "| stage %.2f ms per MSM:" % ((ms1 - ms0) / max(l1 - l0, 1)), "  ".join("%s %.2f" % (k, (b[k] - a[k]) / max(l1 - l0, 1)) for k in b), flush=True)
c.close()

"""Lone-proof latency (one masp_hip_prove_batch call of ONE job, host to host) of the three circuits; options from the MASP_HIP_* environment
(bench.py's names, e.g. MASP_HIP_MSM_C_H_LONE, MASP_HIP_MSM_C_B2_LONE).  Median of 12 after 4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import masp_amd
from bench import options_from_env
from masp_amd import host as H, synthetic, workload as W
ctx = masp_amd.Context(0, **options_from_env())
out = []
for slot, kind in enumerate(("spend", "output", "convert")):
    if os.environ.get("LONE_ONLY") and kind != os.environ["LONE_ONLY"]:
        continue
    cs = H.circuit(kind)[0]
    params = ctx.generate_parameters(cs, synthetic.toxic_waste(1 + slot))
    ctx.load_circuit(slot, params, cs)
    (inputs, aux), = W.instances(kind, 1, first_seed=3, montgomery=True, alloc=lambda k: ctx.host_alloc(cs.n_aux, 32))
    arr, n, keep = ctx.marshal_jobs([(slot, inputs, aux, 1234567, 7654321, None, 1)])
    lat = []
    for _ in range(16):
        t0 = time.perf_counter()
        ctx.prove_marshalled(arr, 1)
        lat.append((time.perf_counter() - t0) * 1e3)
    out.append("%s %.2f" % (kind, sorted(lat[4:])[6]))
    if os.environ.get("LONE_CHAINS"):
        ctx.profile_enable(True)
        for _ in range(3):
            ctx.prove_marshalled(arr, 1)
            print("   ", kind, ctx.profile_read_lone(), flush=True)
        ctx.profile_enable(False)
print(" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("MASP_HIP_")) or "defaults", "| lone ms:", "  ".join(out), flush=True)
ctx.close()

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
    arr, n, keep = ctx.marshal_jobs([(slot, inputs, aux, 0x5a3c2b1d0e0f1a2b3c4d5e6f708192a3b4c5d6e7f8091a2b3c4d5e6f70819203 if os.environ.get('LONE_SMALL_RS') is None else 1234567,
                                     0x1f2e3d4c5b6a79880796a5b4c3d2e1f00f1e2d3c4b5a69788796a5b4c3d2e1f0 if os.environ.get('LONE_SMALL_RS') is None else 7654321, None, 1)])
    if os.environ.get("LONE_VK"):
        vk = ctx.prepare_verifying_key(params)
    if os.environ.get("LONE_RESIDENT"):
        import numpy as np
        insts = W.instances(kind, 256, first_seed=50, montgomery=True, alloc=lambda k: ctx.host_alloc(cs.n_aux, 32))
        jobs = [(slot, i, a, 5 + k, 6 + k, None, 1) for k, (i, a) in enumerate(insts)]
        handle, _ = ctx.batch_upload(jobs)
        rs = np.zeros((4, 256, 64), np.uint8); rs[:, :, 0] = 3; rs[:, :, 32] = 5
        ctx.batch_prove_resident_steps(handle, 256, 4, rs)
    if os.environ.get("LONE_AFTER_BATCHES"):
        # what bench.py has done by the time it measures a lone proof: batches of 256 on every slot
        from concurrent.futures import ThreadPoolExecutor
        big, nb_, keep2 = ctx.marshal_jobs([(slot, inputs, aux, 1000 + k, 2000 + k, None, 1) for k in range(256)])
        with ThreadPoolExecutor(3) as ex:
            list(ex.map(lambda _: ctx.prove_marshalled(big, 256), range(int(os.environ["LONE_AFTER_BATCHES"]))))
        time.sleep(float(os.environ.get("LONE_SLEEP", "0")))
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

"""Does a context created after another one of the same process was closed run slower?  Resident Spend steps (256 proofs each) on context 1,
close it, the same on context 2, 3.  MASP_HIP_SLOTS / GPU_MAX_HW_QUEUES from the environment."""
import os, sys, time, random
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import masp_amd
from bench import options_from_env
from masp_amd import host as H, synthetic, workload as W
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
cs = H.circuit("spend")[0]
insts = W.instances("spend", 256, first_seed=0, montgomery=True)
rng = random.Random(1)
params = None
for k in range(3):
    c = masp_amd.Context(0, **options_from_env())
    if params is None:
        params = c.generate_parameters(cs, synthetic.toxic_waste(1))
    c.load_circuit(0, params, cs)
    rs = np.frombuffer(b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(2 * 8 * 256)), np.uint8).reshape(8, 256, 64)
    jobs = [(0, i, a, bytes(rs[0, j, :32]), bytes(rs[0, j, 32:]), None, 1) for j, (i, a) in enumerate(insts)]
    h, _ = c.batch_upload(jobs)
    c.batch_prove_resident_steps(h, 256, 4, rs[:4])
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        c.batch_prove_resident_steps(h, 256, 8, rs)
        out.append(8 * 256 / (time.perf_counter() - t0))
    # host to host: the same witnesses from page-locked memory of THIS context, one call per slot in flight
    from concurrent.futures import ThreadPoolExecutor
    n_aux = cs.n_aux
    slab = c.host_alloc(n_aux * 256, 32)
    for j, (_, a) in enumerate(insts):
        slab[j * n_aux:(j + 1) * n_aux] = a
    pj = [(0, i, slab[j * n_aux:(j + 1) * n_aux], bytes(rs[0, j, :32]), bytes(rs[0, j, 32:]), None, 1) for j, (i, _) in enumerate(insts)]
    arr, n, keep = c.marshal_jobs(pj)
    S = c.options["slots"]
    with ThreadPoolExecutor(S) as ex:
        list(ex.map(lambda _: c.prove_marshalled(arr, n), range(S)))
    h2h = []
    for rep in range(2):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(S) as ex:
            list(ex.map(lambda _: c.prove_marshalled(arr, n), range(12)))
        h2h.append(12 * 256 / (time.perf_counter() - t0))
    print("context %d of this process (slots %d): resident %s   host to host %s proofs/s" % (k + 1, S, " ".join("%.1f" % v for v in out), " ".join("%.1f" % v for v in h2h)), flush=True)
    c.close()

"""bench.py's end_to_end region on its own: a FRESH LocalTxProver, one warm call over the same 5 120 Spend descriptions, then the timed call;
in_flight from argv (default: the prover's own choice, slots + 1).  MASP_HIP_SLOTS / GPU_MAX_HW_QUEUES from the environment."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from bench import options_from_env
from masp_amd import host as H
from masp_amd import workload as W
from masp_amd.prover import LocalTxProver
n = 5120
cpus = H.effective_cpus()
with ThreadPoolExecutor(cpus) as ex:
    descs = list(ex.map(lambda k: W.description("spend", k), range(n)))
if os.environ.get("E2E_PRELUDE"):
    # what bench.py has done by the time its end_to_end region starts: another context in this process, proved with, then closed
    import masp_amd, random
    import numpy as np
    from masp_amd import synthetic
    R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    cs = H.circuit("spend")[0]
    c0 = masp_amd.Context(0, **options_from_env())
    params = c0.generate_parameters(cs, synthetic.toxic_waste(1))
    c0.load_circuit(0, params, cs)
    insts = W.instances("spend", 256, first_seed=0, montgomery=True)
    rng = random.Random(1)
    rs = np.frombuffer(b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(2 * 4 * 256)), np.uint8).reshape(4, 256, 64)
    jobs = [(0, i, a, bytes(rs[0, j, :32]), bytes(rs[0, j, 32:]), None, 1) for j, (i, a) in enumerate(insts)]
    h, _ = c0.batch_upload(jobs)
    for _ in range(int(os.environ["E2E_PRELUDE"])):
        c0.batch_prove_resident_steps(h, 256, 4, rs)
    c0.close()
    print("prelude done", flush=True)
if os.environ.get("E2E_HOST_CHURN"):
    # host memory only: allocate, touch and free a few GB in 3 MB pieces, keeping every other one (what a process that has built
    # witnesses, job arrays and proof lists looks like to the page allocator) - no GPU context involved
    import numpy as np
    keep = []
    for _ in range(int(os.environ["E2E_HOST_CHURN"])):
        blocks = [np.ones(3 << 20, np.uint8) for _ in range(1024)]
        keep += blocks[::2]
        del blocks
    print("host churn done: %.1f GB kept" % (len(keep) * 3 / 1024), flush=True)
for fl in [int(a) for a in sys.argv[1:]] or [None]:
    prover = LocalTxProver.with_synthetic_parameters(seed=7, options=options_from_env())
    slots = prover._ctx.options["slots"]
    if os.environ.get("E2E_NO_VERIFY"):
        prover._self_verify = False
    cpus = int(os.environ.get("E2E_THREADS", cpus))
    prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=cpus, in_flight=fl)
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=cpus, in_flight=fl)
        out.append(n / (time.perf_counter() - t0))
    print("[verify %s threads %d prelude %s] slots %d in_flight %s: first timed call %.1f, then %.1f, %.1f proofs/s" % ("off" if os.environ.get("E2E_NO_VERIFY") else "on", cpus, os.environ.get("E2E_PRELUDE", "no"), slots, fl, out[0], out[1], out[2]), flush=True)
    prover.close()

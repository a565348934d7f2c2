"""Where does a SECOND context of a process lose its 4 - 9 % in LocalTxProver.prove_batch?  (profiles/r04e_second_context_in_a_process.txt:
the resident and host-to-host paths on such a context are not slower, the end-to-end call is.)

The same end-to-end call — 2 048 Spend descriptions, the default pipeline — on a prover whose context is the process's first heavily used one, then,
after that prover is closed (SCP_KEEP_FIRST=1: left open), on a second prover with a context of its own; every stage of prove_batch is timed from
the outside (the prover's methods wrapped): synthesis thread-seconds, GPU call wall-seconds, verification wall-seconds, plus what the PROCESS burnt
(utime + stime of /proc/self/stat: the HIP runtime's helper threads count) and how many threads it has.  Parameters are generated once, up front,
by a context that does nothing else."""
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import masp_amd  # noqa: E402
from bench import options_from_env  # noqa: E402
from masp_amd import host as H  # noqa: E402
from masp_amd import synthetic  # noqa: E402
from masp_amd import workload as W  # noqa: E402
from masp_amd.prover import LocalTxProver  # noqa: E402

N = int(os.environ.get("SCP_N", "2048"))
cpus = H.effective_cpus()
with ThreadPoolExecutor(cpus) as ex:
    descs = list(ex.map(lambda k: W.description("spend", k), range(N)))
c = masp_amd.Context(0)
params = [c.generate_parameters(H.circuit(k)[0], synthetic.toxic_waste(21 + i)) for i, k in enumerate(("spend", "output", "convert"))]
c.close()


def cpu_seconds():
    f = open("/proc/self/stat").read().rsplit(")", 1)[1].split()
    return (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK")


class Timers:
    def __init__(self):
        self.lock = threading.Lock()
        self.t = {}

    def wrap(self, obj, name, key):
        f = getattr(obj, name)

        def g(*a, **k):
            t0 = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                dt = time.perf_counter() - t0
                with self.lock:
                    s = self.t.setdefault(key, [0.0, 0])
                    s[0] += dt
                    s[1] += 1
        setattr(obj, name, g)


SLOTS_SEQ = [int(x) for x in os.environ.get("SCP_SLOTS", "").split(",") if x]


def pre_streams(n):
    """create, use and destroy n HIP streams before any context exists: the process's stream history without any prover in it"""
    import ctypes as C
    L = C.CDLL("libamdhip64.so")
    ss = [C.c_void_p() for _ in range(n)]
    buf = C.c_void_p()
    assert L.hipMalloc(C.byref(buf), 4096) == 0
    for h in ss:
        assert L.hipStreamCreateWithFlags(C.byref(h), 1) == 0
        assert L.hipMemsetAsync(buf, 0, 4096, h) == 0
    for h in ss:
        assert L.hipStreamSynchronize(h) == 0
    if not os.environ.get("SCP_PRE_STREAMS_KEEP"):
        for h in ss:
            assert L.hipStreamDestroy(h) == 0
    print("%d streams created, used%s before the first context" % (n, "" if os.environ.get("SCP_PRE_STREAMS_KEEP") else ", destroyed"), flush=True)


if os.environ.get("SCP_PRE_STREAMS"):
    pre_streams(int(os.environ["SCP_PRE_STREAMS"]))


def run(tag):
    opts = options_from_env()
    if SLOTS_SEQ:
        opts["slots"] = SLOTS_SEQ.pop(0)
    prover = LocalTxProver(*params, expected=None, options=opts)
    if os.environ.get("SCP_NO_VERIFY"):
        prover._self_verify = False
    prover.warm_up(spends=N, threads=cpus)
    prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=cpus)
    tm = Timers()
    tm.wrap(prover, "prepare_group", "synthesis (thread-s)")
    tm.wrap(prover, "prove_prepared", "gpu calls (wall-s, summed)")
    for k, vk in prover._gpu_vk.items():
        tm.wrap(vk, "verify_batch", "verify_batch %s (wall-s, summed)" % k)
    rates, cpu = [], []
    for rep in range(4):
        c0, t0 = cpu_seconds(), time.perf_counter()
        prover.prove_batch(prover.new_sapling_proving_context(), descs, threads=cpus)
        dt = time.perf_counter() - t0
        rates.append(N / dt)
        cpu.append((cpu_seconds() - c0) / dt)
    print("%-22s proofs/s %s | cores busy %s | threads %d" % (tag, " ".join("%.1f" % r for r in rates), " ".join("%.2f" % x for x in cpu),
                                                               len(os.listdir("/proc/self/task"))), flush=True)
    for k, (s, n) in sorted(tm.t.items()):
        print("    %-42s %8.3f s over %4d calls = %8.3f ms each" % (k, s / 4, n // 4, 1e3 * s / max(n, 1)), flush=True)
    # the host-to-host path on the same context right after (one call per slot in flight): is the GPU side itself slower here?
    jobs = prover.prepare_group("spend", [kw for _, kw in descs[:256]])
    arr, n, keep = prover._ctx.marshal_jobs([(j["slot"], j["inputs"], j["aux"], 5 + i, 6 + i, None, 1) for i, j in enumerate(jobs)])
    S = prover._ctx.options["slots"]
    with ThreadPoolExecutor(S) as ex:
        list(ex.map(lambda _: prover._ctx.prove_marshalled(arr, n), range(S)))
        t0 = time.perf_counter()
        list(ex.map(lambda _: prover._ctx.prove_marshalled(arr, n), range(12)))
    print("    host to host on this context: %.1f proofs/s" % (12 * 256 / (time.perf_counter() - t0)), flush=True)
    # ... and the kernels themselves: the G1 bucket stage of one 256-proof launch sequence alone on the chip (HIP events), resident steps
    import numpy as np
    ctx = prover._ctx
    h, nn = ctx.batch_upload([(j["slot"], j["inputs"], j["aux"], 5 + i, 6 + i, None, 1) for i, j in enumerate(jobs)])
    ctx.batch_prove_resident(h, nn)
    ctx.profile_enable(True)
    ctx.batch_prove_resident(h, nn)
    ms, launches, _ = ctx.profile_read()
    ctx.profile_enable(False)
    rs = np.zeros((6, nn, 64), np.uint8)
    rs[:, :, 0] = 3
    rs[:, :, 32] = 5
    ctx.batch_prove_resident_steps(h, nn, 2, rs[:2])
    t0 = time.perf_counter()
    ctx.batch_prove_resident_steps(h, nn, 6, rs)
    print("    slots %d: isolated G1 stage %.2f ms per MSM; resident %.1f proofs/s" % (S, ms / max(launches, 1), 6 * nn / (time.perf_counter() - t0)), flush=True)
    ctx.batch_free(h)
    if hasattr(prover._ctx._L, "masp_hip_ctx_stream_concurrency"):
        print("    own streams running at the same time: %d of %d; main streams: %d of %d   (hardware queues of the process: %s)" % (prover._ctx.stream_concurrency()[::-1] + prover._ctx.stream_concurrency(True)[::-1] + (os.environ.get("GPU_MAX_HW_QUEUES"),)), flush=True)
    return prover


first = run("context A (first used)")
if os.environ.get("SCP_CONTEXTS") == "1":
    first.close()
    sys.exit(0)
if not os.environ.get("SCP_KEEP_FIRST"):
    first.close()
second = run("context B (second)")
second.close()
if os.environ.get("SCP_KEEP_FIRST"):
    first.close()
third = run("context C (third)")
third.close()

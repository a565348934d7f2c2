"""masp_hip_options::hw_queues: the hardware queues the HIP runtime really gives this process, measured when a context is created
(the two-mode behaviour was avoided by an environment variable the Python wrappers set, which a caller linking
libmasp_hip.so directly never got — "nothing in the C ABI warns when the process has 8 queues for 15 streams").  Now the library sets
GPU_MAX_HW_QUEUES=16 on load if it is unset, offers masp_hip_runtime_prepare, and reports the measured figure.  Each case is a process of
its own: the runtime reads the variable once.  Run with `-m gpu`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import ctypes as C, os, sys, warnings
sys.path.insert(0, %r)
from masp_amd import hip
mode = sys.argv[1]
if mode == "bare":          # what a Rust binary linking the library gets: no wrapper touches the environment, the constructor does
    os.environ.pop("GPU_MAX_HW_QUEUES", None)
    L = C.CDLL(hip.library_path())
    L.masp_hip_runtime_prepare.argtypes = [C.c_int, C.c_int]
    print("prepare", L.masp_hip_runtime_prepare(0, 0))
    h = C.c_void_p()
    assert L.masp_hip_ctx_create(0, C.byref(h)) == 0
    got = hip.OptionsStruct()
    L.masp_hip_ctx_get_options.argtypes = [C.c_void_p, C.POINTER(hip.OptionsStruct)]
    assert L.masp_hip_ctx_get_options(h, C.byref(got)) == 0
    print("hw_queues", got.hw_queues, "slots", got.slots)
    L.masp_hip_ctx_destroy.argtypes = [C.c_void_p]
    L.masp_hip_ctx_destroy(h)
else:                        # the wrapper with the variable forced by the caller
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ctx = hip.Context(0)
        print("hw_queues", ctx.options["hw_queues"], "slots", ctx.options["slots"])
        print("warned", int(any("hardware queue" in str(x.message) for x in w)))
        ctx.close()
""" % ROOT


def _run(mode, env_extra):
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", PROBE, mode], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return dict(l.split()[:2] for l in out.stdout.splitlines() if l.strip())


def test_a_process_that_only_links_the_library_gets_sixteen_queues():
    got = _run("bare", {})
    assert got["prepare"] == "16"                      # the constructor had already set it; prepare(0, 0) keeps it
    assert int(got["hw_queues"]) >= 15                 # 3 slots x 5 streams probed, all concurrent


def test_too_few_queues_are_measured_and_the_wrapper_warns():
    got = _run("wrapper", {"GPU_MAX_HW_QUEUES": "4"})
    assert int(got["hw_queues"]) == 4 and got["warned"] == "1"
    got = _run("wrapper", {"GPU_MAX_HW_QUEUES": "16"})
    assert int(got["hw_queues"]) >= 15 and got["warned"] == "0"


def test_too_many_queues_warn_and_runtime_prepare_never_writes_more_than_twenty():
    """Round 6 (profiles/r06_second_context_root_cause.txt): a process that holds 24 / 32 hardware queues dispatches every kernel 7 / 21 % slower,
    and the runtime's pool only grows — the cap is what counts.  The wrapper warns when the environment allows more than 20;
    masp_hip_runtime_prepare cuts what it writes to MASP_HIP_MAX_USEFUL_HW_QUEUES."""
    got = _run("wrapper", {"GPU_MAX_HW_QUEUES": "32"})
    assert got["warned"] == "1" and int(got["hw_queues"]) <= 21          # (the probe itself stops at 21 streams: it must not grow the pool further)
    code = ("import ctypes as C, os, sys; sys.path.insert(0, %r); from masp_amd import hip; os.environ.pop('GPU_MAX_HW_QUEUES', None); "
            "L = C.CDLL(hip.library_path()); L.masp_hip_runtime_prepare.argtypes = [C.c_int, C.c_int]; "
            "print(L.masp_hip_runtime_prepare(64, 1), L.masp_hip_runtime_prepare(0, 0))" % ROOT)      # (setenv from C: os.environ does not see it)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.split() == ["20", "20"], out.stdout + out.stderr


def test_a_contexts_own_streams_each_get_a_hardware_queue():
    """masp_hip_ctx_stream_concurrency: with the default 16 hardware queues the 5 x slots + 1 streams of a default context run kernels at the
    same time (the probe behind round 6's root cause of the "slower second context": profiles/r06_second_context_root_cause.txt)."""
    import masp_amd
    ctx = masp_amd.Context(0, slots=2)
    try:
        n, c = ctx.stream_concurrency()
        assert n == 13 and c >= 11, (n, c)
        # the streams that work next to each other (context, slots, verifier): a queue each.  Which queue a stream gets depends on every
        # stream this pytest process has created before; the context repairs what it measures (separate_main_streams) and gives up after
        # eight rounds without failing — so one shared queue is reported here, two are a defect
        nm, cm = ctx.stream_concurrency(mains_only=True)
        assert nm == 5 and cm >= 4, (nm, cm)
        if cm < 5:
            import warnings
            warnings.warn("two of the context's five main streams share a hardware queue in this process (%d of %d side by side)" % (cm, nm))
    finally:
        ctx.close()

"""A compiled C caller of the C ABI: tests/native/capi_harness.c includes include/masp_hip.h, fills masp_hip_r1cs / masp_hip_job by
hand and links libmasp_hip.so — the stand-in this image allows for the Rust FFI a fork of masp_proofs would write where
/root/reference/masp_proofs/src/prover.rs:156-261 calls bellperson (SURVEY.md §7 step 5, §8b "a C++ harness plays that role";
VERDICT r05 missing 4).  No Python is in the harness's call path: this file only writes its input, runs it and compares the proofs
it wrote with the oracle's.  The CPU part compiles the harness as strict C99 and C11 (the header's static assertions on
sizeof / offsetof are then live) and runs the entry points that need no device."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import toy_r1cs
from pyref import R

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "native", "capi_harness.c")
EXE = os.path.join(HERE, "native", "_capi_harness")
LIBDIR = os.path.join(ROOT, "masp_amd")


def _build(std="c11", out=EXE):
    deps = [SRC, os.path.join(ROOT, "include", "masp_hip.h"), os.path.join(LIBDIR, "libmasp_hip.so")]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(p) for p in deps):
        return out
    # linked like a Rust `#[link(name = "masp_hip")]` block: against the shared library, found at run time next to the package
    subprocess.check_call(["gcc", "-std=" + std, "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", LIBDIR, "-lmasp_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", out])
    return out


def test_the_header_compiles_as_c99_and_c11_and_its_layout_is_what_the_bindings_say(tmp_path):
    for std in ("c99", "c11"):
        exe = _build(std, str(tmp_path / ("harness_" + std)))
        out = subprocess.run([exe, "--abi"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        got = dict(l.split() for l in out.stdout.splitlines() if len(l.split()) == 2 and l.split()[1].lstrip("-").isdigit())
        # the numbers include/masp_hip.h documents and asserts, INTEGRATION.md's #[repr(C)] structs mirror, and masp_amd/hip.py uses
        assert got["sizeof_job"] == "120" and got["sizeof_r1cs"] == "88" and got["sizeof_options"] == "76"
        assert got["offsetof_job_inputs"] == "8" and got["offsetof_job_r"] == "48" and got["offsetof_job_s"] == "80" and got["offsetof_job_aux_form"] == "112"
        assert got["offsetof_options_window_bits_b2"] == "72"
        assert "abi ok" in out.stdout
    from masp_amd import hip
    import ctypes as C
    assert C.sizeof(hip.JobStruct) == 120 and C.sizeof(hip.OptionsStruct) == 76
    from masp_amd.r1cs import R1csStruct
    assert C.sizeof(R1csStruct) == 88
    # ... and the prose agrees with the compiler (header and INTEGRATION.md said "88 -> 96")
    for path in ("include/masp_hip.h", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, path)).read()
        assert "88 -> 96" not in text and "88 → 96" not in text and "112 -> 120" in text.replace("→", "->"), path


def _mont(aux):
    out = np.zeros_like(aux)
    for j in range(aux.shape[0]):
        v = int.from_bytes(aux[j].tobytes(), "little") * (1 << 256) % R
        out[j] = np.frombuffer(v.to_bytes(32, "little"), np.uint8)
    return out


@pytest.mark.gpu
def test_a_c_program_proves_through_the_abi_and_the_bytes_are_the_oracles(tmp_path):
    exe = _build()
    cs, inputs, aux, vals = toy_r1cs.make(4242, n_inputs=5, n_free=80, n_constraints=900, bool_share=0.7)
    tw = toy_r1cs.toxic(4242)
    params = O.generate_parameters(cs, tw)          # the oracle's CRS bytes: the harness only ever sees the file
    a, b, c = O.r1cs_eval(cs, inputs, aux)[:3]
    aux_m = _mont(aux)
    jobs = []                                       # (aux_form, abc?, r, s): job 0 canonical (masp_hip_prove takes it), then the three forms mixed
    for k in range(19):
        form = (0, 1, 0)[k % 3] if k else 0
        jobs.append((form, k % 3 == 2, 1000 + 7 * k, R - 1 - k if k % 5 == 0 else 2000 + 11 * k))
    blob = bytearray(b"MHH1")
    blob += struct.pack("<III", cs.n_inputs, cs.n_aux, cs.n_constraints)
    for rp, col, coef in cs.mats:
        blob += struct.pack("<I", int(rp[-1])) + rp.tobytes() + col.tobytes() + coef.tobytes()
    blob += struct.pack("<Q", params.size) + params.tobytes()
    blob += struct.pack("<I", len(jobs))
    for form, abc, r, s in jobs:
        blob += struct.pack("<II", form, int(abc)) + inputs.tobytes() + (aux_m if form else aux).tobytes()
        if abc:
            blob += a.tobytes() + b.tobytes() + c.tobytes()
        blob += r.to_bytes(32, "little") + s.to_bytes(32, "little")
    case, out = tmp_path / "case.bin", tmp_path / "proofs.bin"
    case.write_bytes(bytes(blob))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}     # a process that only links the library: its constructor says 16
    run = subprocess.run([exe, str(case), str(out)], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "proved %d jobs" % len(jobs) in run.stdout
    got = out.read_bytes()
    assert len(got) == 192 * (1 + len(jobs))
    P = O.Params(params)
    want = [O.create_proof(P, cs, inputs, aux, r, s) for _, _, r, s in jobs]
    assert got[:192] == want[0], "masp_hip_prove (C caller) differs from the oracle"
    for k in range(len(jobs)):
        assert got[192 * (1 + k):192 * (2 + k)] == want[k], "job %d (aux_form %d, abc %s) differs from the oracle" % (k, jobs[k][0], jobs[k][1])
    assert want[0] == O.closed_form_proof(cs, tw, inputs, aux, jobs[0][2], jobs[0][3])
    assert all(O.verify_proof(params, w, vals[1:5]) == 1 for w in want[:4])
    hwq = [l for l in run.stdout.splitlines() if l.startswith("options ")][0].split()
    assert int(hwq[hwq.index("hw_queues") + 1]) >= 10        # 2 slots x 5 streams all concurrent: the library's constructor set the queue count

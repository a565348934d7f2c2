"""The C-ABI libraries load without a GPU and export every symbol their headers declare (no compute calls)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_libmasp_hip_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "masp_hip.h")).read()
    names = sorted(set(re.findall(r"\b(masp_hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    lib = C.CDLL(os.path.join(ROOT, "masp_amd", "libmasp_hip.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_device_fails_loudly_not_silently():
    import masp_amd
    import torch
    if torch.cuda.is_available():
        return
    try:
        masp_amd.Context(0)
    except masp_amd.MaspHipError as e:
        assert e.code == 4        # MASP_HIP_E_NO_DEVICE: there is no CPU fallback
    else:
        raise AssertionError("Context() must fail without a GPU")


def test_product_never_references_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "masp_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="replace").read()
                for line in text.splitlines():
                    code = line.split("//")[0].split("#")[0] if not f.endswith(".py") else line.split("#")[0]
                    if re.search(r"(import|include|from)\s+.*oracle", code) or "liboracle" in code or "oracle_lib" in code:
                        bad.append((f, line.strip()))
    assert not bad, bad


def test_libmasp_host_exports():
    lib = C.CDLL(os.path.join(ROOT, "masp_amd", "libmasp_host.so"))
    for n in ("masp_host_circuit_setup", "masp_host_spend_assignment", "masp_host_output_assignment", "masp_host_convert_assignment",
              "masp_host_vk_prepare", "masp_host_vk_verify", "masp_host_pedersen_hash", "masp_host_generator"):
        assert hasattr(lib, n), n


def _dynamic_symbols(lib):
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "masp_amd", lib)], text=True)
    return [line.split()[-1] for line in out.splitlines() if line.strip()]


def test_libraries_export_nothing_but_their_c_abi():
    """The ABI surface is the header, not whatever the compiler happened to emit: linked with version scripts
    (masp_amd/csrc/exports_*.map), `nm -D --defined-only` lists the entry points only."""
    hip = _dynamic_symbols("libmasp_hip.so")
    assert hip and all(s.startswith("masp_hip_") for s in hip), [s for s in hip if not s.startswith("masp_hip_")][:10]
    hdr = open(os.path.join(ROOT, "include", "masp_hip.h")).read()
    declared = set(re.findall(r"\b(masp_hip_[a-z0-9_]+)\s*\(", hdr))
    assert set(hip) == declared, (sorted(set(hip) - declared), sorted(declared - set(hip)))
    host = _dynamic_symbols("libmasp_host.so")
    assert host and all(s.startswith("masp_host_") for s in host), [s for s in host if not s.startswith("masp_host_")][:10]
    # ... and libmasp_host.so's header (include/masp_host.h, round 6) declares exactly what the library exports; host_api.cpp includes it,
    # so every signature is the compiler's to check
    hhdr = open(os.path.join(ROOT, "include", "masp_host.h")).read()
    hdecl = set(re.findall(r"\b(masp_host_[a-z0-9_]+)\s*\(", hhdr))
    assert set(host) == hdecl, (sorted(set(host) - hdecl), sorted(hdecl - set(host)))
    assert '#include "../../../include/masp_host.h"' in open(os.path.join(ROOT, "masp_amd", "csrc", "host", "host_api.cpp")).read()


def test_masp_host_header_is_plain_c():
    import subprocess
    for std in ("c99", "c11"):
        subprocess.check_call(["gcc", "-std=" + std, "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "masp_host.h")])
        subprocess.check_call(["gcc", "-std=" + std, "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "masp_hip.h")])


def test_the_cxx_host_mirror_reaches_the_libraries_through_their_headers_only():
    """include/masp_tx_prover.hpp (masp::LocalTxProver above the two C ABIs) calls nothing the C headers do not declare, includes nothing
    of the product's internals and nothing of the oracle: what it needs IS the boundary."""
    text = open(os.path.join(ROOT, "include", "masp_tx_prover.hpp")).read()
    code = "\n".join(line.split("//")[0] for line in text.splitlines())
    hdr = open(os.path.join(ROOT, "include", "masp_hip.h")).read() + open(os.path.join(ROOT, "include", "masp_host.h")).read()
    declared = set(re.findall(r"\b(masp_h(?:ip|ost)_[a-z0-9_]+)\s*\(", hdr)) | set(re.findall(r"\b(masp_h(?:ip|ost)_[a-z0-9_]+)\b(?=;|\s*\{)", hdr))
    types = {"masp_hip_ctx", "masp_hip_vk", "masp_hip_job", "masp_hip_r1cs", "masp_hip_options", "masp_host_spend_job", "masp_host_convert_job"}
    used = set(re.findall(r"\b(masp_h(?:ip|ost)_[a-z0-9_]+)\b", code))
    assert used - declared - types == set(), sorted(used - declared - types)
    assert {"masp_hip_prove_batch", "masp_hip_verify_batch", "masp_hip_circuit_load", "masp_host_spend_assignments", "masp_host_vk_verify"} <= used
    own = re.findall(r'#include\s+"([^"]+)"', text)                 # (the <...> ones are the standard library's and the system's)
    assert sorted(own) == ["masp_hip.h", "masp_host.h"], own

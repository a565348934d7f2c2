"""The host side of the drop-in in C++: include/masp_tx_prover.hpp (masp::LocalTxProver / masp::SaplingProvingContext above the two C
ABIs) used by a compiled program, tests/native/tx_prover_harness.cpp — the stand-in this image allows for a Rust crate that implemented
`trait TxProver` (/root/reference/masp_primitives/src/sapling/prover.rs:17-83) the way masp_proofs::prover::LocalTxProver does
(/root/reference/masp_proofs/src/prover.rs:156-261).  No Python is in the harness's call path: this file writes the case (parameter
bytes, descriptions, explicit blinding scalars), runs the program and compares what it wrote — proofs, cv, rk, and the context's bsk and
cv_sum — with the Python mirror (masp_amd/prover.py) on the same descriptions and with the oracle's proof bytes.

CPU part: the header compiles as strict C++17 (-pedantic -Wall -Wextra -Werror) with every template instantiated, and its scalar
helpers (jubjub::Fr addition of bsk, the nullifier's multipacking, the blinding-scalar sampler) agree with Python integers."""
import os
import random
import struct
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "native", "tx_prover_harness.cpp")
EXE = os.path.join(HERE, "native", "_tx_prover_harness")
LIBDIR = os.path.join(ROOT, "masp_amd")


def _build(out=EXE):
    deps = [SRC] + [os.path.join(ROOT, "include", h) for h in ("masp_tx_prover.hpp", "masp_hip.h", "masp_host.h")] + \
           [os.path.join(LIBDIR, l) for l in ("libmasp_hip.so", "libmasp_host.so")]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(p) for p in deps):
        return out
    subprocess.check_call(["g++", "-std=c++17", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", LIBDIR, "-lmasp_hip", "-lmasp_host", "-pthread", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib",
                           "-Wl,--allow-shlib-undefined", "-o", out])
    return out


def _b32(x):
    return x.to_bytes(32, "little") if isinstance(x, int) else bytes(x)


def test_the_header_compiles_as_strict_cxx17_and_its_scalar_helpers_agree_with_python(tmp_path):
    from masp_amd import host as H
    exe = _build(str(tmp_path / "harness"))
    RJ = H.JUBJUB_ORDER
    rng = random.Random(5)
    pairs = [(RJ - 1, RJ - 1), (0, 0), (0, RJ - 1), (5, 7), (RJ - 1, 1)] + [(rng.randrange(RJ), rng.randrange(RJ)) for _ in range(40)]
    packs = [bytes(range(224, 256)), b"\xff" * 32, b"\x00" * 32] + [bytes(rng.getrandbits(8) for _ in range(32)) for _ in range(8)]
    lines = []
    for a, b in pairs:
        lines += ["add %s %s" % (_b32(a).hex(), _b32(b).hex()), "sub %s %s" % (_b32(a).hex(), _b32(b).hex())]
    lines += ["pack " + p.hex() for p in packs]
    seeds = [bytes(32), b"\xff" * 32] + [bytes(rng.getrandbits(8) for _ in range(32)) for _ in range(6)]
    lines += ["rcm " + x.hex() for x in seeds] + ["random 2000"]
    vec = tmp_path / "vectors.txt"
    vec.write_text("\n".join(lines) + "\n")
    out = subprocess.run([exe, "--selftest", str(vec)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "selftest ok" in out.stdout, out.stdout + out.stderr
    got = out.stdout.splitlines()
    k = 0
    for a, b in pairs:                                        # SaplingProvingContext::bsk: "Outputs subtract from the total."
        assert got[k] == "add " + _b32((a + b) % RJ).hex() and got[k + 1] == "sub " + _b32((a - b) % RJ).hex(), (a, b)
        k += 2
    for p in packs:                                           # multipack::compute_multipacking(bytes_to_bits_le(nf)) (sapling/prover.rs:138-139)
        want = H.multipack(p)
        assert got[k] == "pack %s %s" % (_b32(want[0]).hex(), _b32(want[1]).hex()), p.hex()
        k += 1
    from masp_amd.prover import Rseed
    for x in seeds:                                           # Note::rcm of Rseed::AfterZip212 (sapling.rs:856-864): the two mirrors agree (the reference holds no vector for it)
        assert got[k] == "rcm " + _b32(Rseed.after_zip212(x).rcm()).hex(), x.hex()
        k += 1
    assert Rseed.before_zip212(7).rcm() == 7
    assert got[k] == "random 2000 canonical 2000 distinct 2000"
    assert got[k + 1] == "default location none"              # with_default_location (prover.rs:120-136): None without the folder
    assert got[k + 2] == "context %s %s" % (_b32(0).hex(), H.JUBJUB_IDENTITY.hex())


def _record(kind, kw, r, s):
    """one description in the harness's case format (tests/native/tx_prover_harness.cpp)"""
    from masp_amd import host as H
    if kind == "spend":
        ak, nsk = kw["proof_generation_key"]
        sib, pos = kw["merkle_path"]
        body = _b32(ak) + _b32(nsk) + bytes(kw["diversifier"]) + _b32(kw["rcm"]) + _b32(kw["ar"]) + _b32(kw["asset_type"]) + \
            struct.pack("<Q", kw["value"]) + _b32(kw["anchor"]) + b"".join(_b32(x) for x in sib) + struct.pack("<Q", pos) + _b32(kw["rcv"])
        k = 0
    elif kind == "output":
        d, pk = kw["payment_address"]
        body = _b32(kw["esk"]) + bytes(d) + _b32(pk) + _b32(kw["rcm"]) + _b32(kw["asset_type"]) + struct.pack("<Q", kw["value"]) + _b32(kw["rcv"])
        k = 1
    else:
        sib, pos = kw["merkle_path"]
        ac = kw["allowed_conversion"]
        gen = ac.generator if isinstance(ac, H.AllowedConversion) else ac
        body = _b32(gen) + struct.pack("<Q", kw["value"]) + _b32(kw["anchor"]) + b"".join(_b32(x) for x in sib) + struct.pack("<Q", pos) + _b32(kw["rcv"])
        k = 2
    return struct.pack("<I", k) + body + _b32(r) + _b32(s)


def _python_results(lp, descs, rs):
    """the Python mirror, one description at a time -> [(status, zkproof, cv, rk)], bsk, cv_sum"""
    from masp_amd import prover as P
    pc = lp.new_sapling_proving_context()
    out = []
    for (kind, kw), (r, s) in zip(descs, rs):
        try:
            if kind == "spend":
                zk, cv, rk = lp.spend_proof(pc, kw["proof_generation_key"], kw["diversifier"], P.Rseed.before_zip212(kw["rcm"]), kw["ar"], kw["asset_type"], kw["value"],
                                            kw["anchor"], kw["merkle_path"], kw["rcv"], rs=(r, s))
            elif kind == "output":
                zk, cv = lp.output_proof(pc, kw["esk"], kw["payment_address"], kw["rcm"], kw["asset_type"], kw["value"], kw["rcv"], rs=(r, s))
                rk = bytes(32)
            else:
                zk, cv = lp.convert_proof(pc, kw["allowed_conversion"], kw["value"], kw["anchor"], kw["merkle_path"], kw["rcv"], rs=(r, s))
                rk = bytes(32)
            out.append((1, bytes(zk), bytes(cv), bytes(rk)))
        except P.ProvingError:
            out.append((0, bytes(192), bytes(32), bytes(32)))
    return out, _b32(pc.bsk), bytes(pc.cv_sum)


def _descriptions(n_spend, n_output, n_convert, seed):
    """valid descriptions of the three kinds, interleaved, plus the reference's two Err(()) cases for a Spend: a diversifier without a
    group hash (sapling/prover.rs:84) and a statement that does not hold (wrong anchor: the proof fails its self-check, :148) — and
    a Convert whose anchor is wrong (:266)"""
    from masp_amd import host as H
    from masp_amd import workload as W
    descs = []
    for k in range(max(n_spend, n_output, n_convert)):
        if k < n_spend:
            descs.append(W.description("spend", seed + k))
        if k < n_output:
            descs.append(W.description("output", seed + k))
        if k < n_convert:
            descs.append(W.description("convert", seed + k))
    kind, kw = W.description("spend", seed + 1000)
    bad = None
    for b in range(256):
        try:
            H.spend_leaf(*kw["proof_generation_key"], bytes([b]) * 11, kw["rcm"], kw["asset_type"], 1)
        except H.HostError:
            bad = bytes([b]) * 11
            break
    assert bad is not None
    descs.insert(2, ("spend", dict(kw, diversifier=bad)))
    wrong = (int.from_bytes(_b32(kw["anchor"]), "little") + 1) % H.FR_MODULUS
    descs.insert(5, ("spend", dict(kw, anchor=wrong)))
    kind, kw = W.description("convert", seed + 1001)
    descs.append(("convert", dict(kw, anchor=(int.from_bytes(_b32(kw["anchor"]), "little") + 1) % H.FR_MODULUS)))
    return descs


def _run_case(tmp_path, lp, descs, rs, mode, batch_cap, threads, devices=None):
    blob = bytearray(b"MTP1")
    for k in ("spend", "output", "convert"):
        p = lp.parameters[k]
        blob += struct.pack("<Q", p.size) + p.tobytes()
    blob += struct.pack("<IIIII", 1, threads, batch_cap, mode, len(descs))
    for (kind, kw), (r, s) in zip(descs, rs):
        blob += _record(kind, kw, r, s)
    case, out = tmp_path / ("case%d.bin" % mode), tmp_path / ("out%d.bin" % mode)
    case.write_bytes(bytes(blob))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    if devices:
        env["MASP_TXP_DEVICES"] = devices
    run = subprocess.run([_build(), str(case), str(out)], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    got = out.read_bytes()
    assert len(got) == 260 * len(descs) + 64, run.stdout
    recs = [(got[260 * i], got[260 * i + 4:260 * i + 196], got[260 * i + 196:260 * i + 228], got[260 * i + 228:260 * i + 260]) for i in range(len(descs))]
    return recs, got[-64:-32], got[-32:], run.stdout


def test_binding_sig_in_cxx_is_the_python_mirrors_signature_and_verifies(tmp_path):
    """masp::SaplingProvingContext::binding_sig (= sapling/prover.rs:279-326; RedJubjub of masp_primitives/src/sapling/redjubjub.rs:138-160
    with BLAKE2b-512 and jubjub::Fr arithmetic inside the header): with the same 80-byte nonce it is byte for byte the Python mirror's
    signature, it verifies under the key the verifier reconstructs, and the reference's three Err(()) cases come back empty.  H* and the
    multiplication mod the Jubjub order are also compared on their own (message lengths around BLAKE2b's 128-byte block)."""
    from masp_amd import host as H
    from masp_amd import redjubjub as RJS
    from masp_amd.prover import SaplingProvingContext
    exe = _build(str(tmp_path / "harness"))
    RJ = H.JUBJUB_ORDER
    rng = random.Random(4)
    lines, want = [], []
    for la, lb in ((0, 0), (80, 64), (32, 64), (127, 1), (128, 0), (129, 300), (256, 5), (1, 127), (64, 64)):
        a, b = bytes(rng.getrandbits(8) for _ in range(la)), bytes(rng.getrandbits(8) for _ in range(lb))
        lines.append("hstar %s %s" % (a.hex() or "-", b.hex() or "-"))
        want.append("hstar " + _b32(RJS.h_star(a, b)).hex())
    for a, b in [(RJ - 1, RJ - 1), (0, 5), (1, RJ - 1)] + [(rng.randrange(RJ), rng.randrange(RJ)) for _ in range(20)]:
        lines.append("mul %s %s" % (_b32(a).hex(), _b32(b).hex()))
        want.append("mul " + _b32(a * b % RJ).hex())
    A, B = H.asset_identifier(b"asset A"), H.asset_identifier(b"asset B")
    v1, v2, v3 = 1000, 250, 7
    rcv = [rng.randrange(1, RJ) for _ in range(3)]
    ctx = SaplingProvingContext()                            # spend(v1, A) + spend(v3, B) - output(v2, A), as tests/test_binding_sig.py
    ctx._spend_like(rcv[0], H.value_commitment(A, v1, rcv[0])[0])
    ctx._output(rcv[1], H.value_commitment(A, v2, rcv[1])[0])
    ctx._spend_like(rcv[2], H.value_commitment(B, v3, rcv[2])[0])
    ctx2 = SaplingProvingContext()                           # outputs only: a negative balance
    ctx2._output(rcv[1], H.value_commitment(A, v2, rcv[1])[0])
    sighash, T = bytes(rng.getrandbits(8) for _ in range(32)), bytes(rng.getrandbits(8) for _ in range(80))

    def line(c, amount):
        return "sig %s %s %s %s %d %s" % (_b32(c.bsk).hex(), bytes(c.cv_sum).hex(), sighash.hex(), T.hex(), len(amount),
                                          " ".join("%s %s" % (bytes(a).hex(), (v % (1 << 128)).to_bytes(16, "little").hex()) for a, v in amount))
    sigs = []
    for c, amount in ((ctx, [(A, v1 - v2), (B, v3)]), (ctx2, [(A, -v2)])):
        lines.append(line(c, amount))
        sigs.append((c, c.binding_sig(amount, sighash, rng=lambda n: T)))
        want.append("sig " + sigs[-1][1].hex())
    for amount in ([(A, v1 - v2 + 1), (B, v3)], [(A, v1 - v2)], [(A, -(1 << 127))]):      # wrong balance, a missing asset, i128::MIN
        lines.append(line(ctx, amount))
        want.append("sig None")
    vec = tmp_path / "vectors.txt"
    vec.write_text("\n".join(lines) + "\n")
    out = subprocess.run([exe, "--selftest", str(vec)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "selftest ok" in out.stdout, out.stdout + out.stderr
    got = out.stdout.splitlines()
    for g, w, l in zip(got, want, lines):
        assert g == w, l[:80]
    g_rcv = H.point_bytes(*H.generator_uv(3))
    for (c, sig), g in zip(sigs, [x for x in got if x.startswith("sig ") and x != "sig None"]):
        bvk = RJS.public_key(c.bsk, g_rcv)
        assert RJS.verify(bvk, bvk + sighash, bytes.fromhex(g[4:]), g_rcv)


def _load(exe, paths, expected):
    args = [exe, "--load"] + [str(p) for p in paths]
    for n, hx in expected:
        args += [str(n), hx]
    return subprocess.run(args, capture_output=True, text=True, timeout=600).stdout.strip()


def test_from_paths_checks_sizes_then_digests_before_it_touches_a_device(tmp_path):
    """masp::LocalTxProver::from_paths = LocalTxProver::new = load_parameters (lib.rs:278-328): the file sizes first, from the file system,
    then the BLAKE2b-512 digest of every whole file (parse_parameters, lib.rs:351-388); the reference panics on a mismatch, this throws
    masp::Panic — before any device is touched (here, without a GPU, correct files get as far as "no usable HIP device")."""
    import hashlib
    exe = _build(str(tmp_path / "harness"))
    rng = random.Random(6)
    blobs = [bytes(rng.getrandbits(8) for _ in range(5000 + 300 * i)) for i in range(3)]
    paths = []
    for i, b in enumerate(blobs):
        paths.append(tmp_path / ("p%d.params" % i))
        paths[-1].write_bytes(b)
    good = [(len(b), hashlib.blake2b(b, digest_size=64).hexdigest()) for b in blobs]
    out = _load(exe, paths, good)
    assert out == "loaded" or "no usable HIP device" in out or "masp_hip_circuit_load" in out, out     # (random bytes are no Parameters on a GPU box)
    out = _load(exe, paths, [good[0], (good[1][0] + 1, good[1][1]), good[2]])
    assert out.startswith("panic at load: masp output parameters") and "5300 bytes on disk, expected 5301" in out, out
    out = _load(exe, paths, [good[0], good[1], (good[2][0], good[0][1])])
    assert out.startswith("panic at load: masp convert parameters: BLAKE2b-512 digest " + good[2][1] + ", expected " + good[0][1]), out
    out = _load(exe, paths, [("mpc", "-")] + good[1:])           # the default: the pinned MPC files (MASP_SPEND_BYTES, lib.rs:60-76)
    assert "5000 bytes on disk, expected 49848572" in out, out
    from masp_amd import params as P
    from masp_tx_prover_constants import constants                # the header's pinned sizes and digests are the Python mirror's
    assert constants() == [(P.EXPECTED[k].bytes, P.EXPECTED[k].hash) for k in P.KINDS]


@pytest.mark.gpu
def test_a_cxx_program_holds_a_local_tx_prover_and_gets_the_python_mirrors_bytes(tmp_path):
    """trait TxProver, one description at a time (mode 0), then the same descriptions through the batch methods in batches of 8 on four
    synthesis threads (mode 1): proofs, cv, rk, Err(()) cases, bsk and cv_sum are those of the Python mirror; a proof of each kind is the
    oracle's."""
    import oracle_lib as O
    from masp_amd import host as H
    from masp_amd import prover as P
    lp = P.LocalTxProver.with_synthetic_parameters(seed=8)
    descs = _descriptions(19, 5, 4, seed=700)
    rng = random.Random(99)
    rs = [(rng.randrange(H.FR_MODULUS), rng.randrange(H.FR_MODULUS)) for _ in descs]
    want, bsk, cv_sum = _python_results(lp, descs, rs)
    assert [w[0] for w in want].count(0) == 3 and want[2][0] == 0 and want[5][0] == 0 and want[-1][0] == 0
    # the oracle's bytes for the first valid description of every kind (the Python mirror is itself tested against it: test_gpu_parity.py)
    seen = set()
    for (kind, kw), (r, s), w in zip(descs, rs, want):
        if kind in seen or not w[0]:
            continue
        seen.add(kind)
        cs, _ = H.circuit(kind)
        if kind == "spend":
            inputs, aux = H.spend_assignment(*kw["proof_generation_key"], kw["diversifier"], kw["rcm"], kw["ar"], kw["asset_type"], kw["value"], kw["anchor"],
                                             *kw["merkle_path"], kw["rcv"])[:2]
        elif kind == "output":
            inputs, aux = H.output_assignment(kw["esk"], *kw["payment_address"], kw["rcm"], kw["asset_type"], kw["value"], kw["rcv"])[:2]
        else:
            inputs, aux = H.convert_assignment(kw["allowed_conversion"].generator, kw["value"], kw["anchor"], *kw["merkle_path"], kw["rcv"])[:2]
        assert w[1] == O.create_proof(O.Params(lp.parameters[kind]), cs, inputs, aux, r, s), kind
    assert seen == {"spend", "output", "convert"}
    lp.close()                                               # (its parameter bytes stay: the harness loads them into a context of its own)
    import hashlib
    paths = []
    for k in ("spend", "output", "convert"):                  # LocalTxProver::new on files with the right sizes and digests, then with a wrong one
        paths.append(tmp_path / (k + ".params"))
        paths[-1].write_bytes(lp.parameters[k].tobytes())
    good = [(lp.parameters[k].size, hashlib.blake2b(lp.parameters[k].tobytes(), digest_size=64).hexdigest()) for k in ("spend", "output", "convert")]
    assert _load(_build(), paths, good) == "loaded"
    assert "BLAKE2b-512 digest" in _load(_build(), paths, [good[0], (good[1][0], good[0][1]), good[2]])
    # ... one description at a time; the batch methods; the batch methods on a prover over two device contexts (this GPU listed twice: the
    # library deals the batches to them from one queue)
    for mode, cap, threads, devices in ((0, 0, 1, None), (1, 8, 4, None), (1, 8, 4, "0,0")):
        recs, got_bsk, got_cv_sum, log = _run_case(tmp_path, lp, descs, rs, mode, cap, threads, devices)
        assert ("devices 2" if devices else "devices 1") in log, log
        for i, (g, w) in enumerate(zip(recs, want)):
            assert g[0] == w[0], "description %d (%s): status %d, the Python mirror says %d\n%s" % (i, descs[i][0], g[0], w[0], log)
            assert g[1:] == w[1:], "description %d (%s), mode %d: bytes differ from the Python mirror" % (i, descs[i][0], mode)
        assert got_bsk == bsk and got_cv_sum == cv_sum, "mode %d: the context differs" % mode
        assert "Some %d, None 3, Panic 0" % (len(descs) - 3) in log
        if mode == 1:                                            # 21 Spend descriptions in batches of 8: the builder's Progress after each
            assert "batch_cap 8" in log and "progress: 3 calls, last 21 of 21" in log, log
            assert "shared by 3 threads: 36 calls, 0 differences" in log, log       # one prover, three threads, a context each

"""Synthetic proving workloads shaped like the reference's own benches — the instances `bench.py`, the tools and the
GPU parity tests prove.

    Spend    /root/reference/masp_proofs/benches/sapling.rs:38-86   (asset "benchmark", value 1, random keys / path; here the
             anchor is the real root of the path so that the statement is true and the proof verifies)
    Convert  /root/reference/masp_proofs/benches/convert.rs:31-66   (three-asset AllowedConversion: -(i+1) "asset i",
             +(i+1) "asset i+1", +(i+1) "reward")
    Output   no bench exists in the reference; same construction as its Spend bench.

Every instance is independent (own keys, diversifier, path, randomness): SURVEY.md §8d "256 independent instances".
Witnesses come from the C++ synthesizer (libmasp_host), which releases the GIL: `instances()` builds them on all host
cores.
"""
import random
import time
from concurrent.futures import ThreadPoolExecutor

from . import host as H

KIND_SLOT = {"spend": H.SPEND, "output": H.OUTPUT, "convert": H.CONVERT}


def description(kind, seed):
    """-> (kind, kwargs of LocalTxProver.prepare_<kind>) for instance `seed` (deterministic)."""
    rng = random.Random("masp-workload-%s-%d" % (kind, seed))
    sc = lambda: rng.randrange(1, H.JUBJUB_ORDER)     # noqa: E731
    siblings = [rng.randrange(H.FR_MODULUS) for _ in range(32)]
    pos = rng.getrandbits(32)
    if kind == "spend":
        ident = H.asset_identifier(b"benchmark")
        ak = H.jubjub_mul(H.point_bytes(*H.generator_uv(4)), sc())
        nsk, ar, rcm, rcv = sc(), sc(), sc(), sc()
        while True:
            d = bytes(rng.getrandbits(8) for _ in range(11))
            try:
                cmu, _ = H.spend_leaf(ak, nsk, d, rcm, ident, 1)
                break
            except H.HostError:
                continue
        return kind, dict(proof_generation_key=(ak, nsk), diversifier=d, rcm=rcm, ar=ar, asset_type=ident, value=1,
                          anchor=H.merkle_root(cmu, siblings, pos), merkle_path=(siblings, pos), rcv=rcv)
    if kind == "output":
        ident = H.asset_identifier(b"benchmark")
        pk = H.jubjub_mul(H.point_bytes(*H.generator_uv(0)), sc())
        while True:
            d = bytes(rng.getrandbits(8) for _ in range(11))
            try:
                H.note_cmu(ident, 1, d, pk, 1)               # only to test the diversifier: g_d must exist
                break
            except ValueError:
                continue
        return kind, dict(esk=sc(), payment_address=(d, pk), rcm=sc(), asset_type=ident, value=1 + rng.getrandbits(20), rcv=sc())
    if kind == "convert":
        i = rng.getrandbits(31)
        ac = H.AllowedConversion([(H.asset_identifier(b"asset %d" % i), -(i + 1)), (H.asset_identifier(b"asset %d" % (i + 1)), i + 1),
                                  (H.asset_identifier(b"reward"), i + 1)])
        return kind, dict(allowed_conversion=ac, value=1 + rng.getrandbits(40), anchor=H.merkle_root(ac.cmu(), siblings, pos),
                          merkle_path=(siblings, pos), rcv=sc())
    raise ValueError(kind)


def assignment(kind, kw, aux_out=None):
    """-> (inputs u8[n_in,32], aux u8[n_aux,32]) of a description (the synthesizer's output; `aux_out`: where to write aux)."""
    if kind == "spend":
        ak, nsk = kw["proof_generation_key"]
        sib, pos = kw["merkle_path"]
        inputs, aux, *_ = H.spend_assignment(ak, nsk, kw["diversifier"], kw["rcm"], kw["ar"], kw["asset_type"], kw["value"], kw["anchor"], sib, pos,
                                             kw["rcv"], aux_out=aux_out)
    elif kind == "output":
        d, pk = kw["payment_address"]
        inputs, aux, _ = H.output_assignment(kw["esk"], d, pk, kw["rcm"], kw["asset_type"], kw["value"], kw["rcv"], aux_out=aux_out)
    else:
        sib, pos = kw["merkle_path"]
        inputs, aux, _ = H.convert_assignment(kw["allowed_conversion"].generator, kw["value"], kw["anchor"], sib, pos, kw["rcv"], aux_out=aux_out)
    return inputs, aux


def public_inputs(inputs):
    """The statement of a proof = the input assignment without ONE (what `verify_proof` takes), as ints."""
    return [int.from_bytes(inputs[i].tobytes(), "little") for i in range(1, inputs.shape[0])]


def _spend_item(kw):
    ak, nsk = kw["proof_generation_key"]
    sib, pos = kw["merkle_path"]
    return (ak, nsk, kw["diversifier"], kw["rcm"], kw["ar"], kw["asset_type"], kw["value"], kw["anchor"], sib, pos, kw["rcv"])


def _convert_item(kw):
    sib, pos = kw["merkle_path"]
    return (kw["allowed_conversion"].generator, kw["value"], kw["anchor"], sib, pos, kw["rcv"])


def assignments(kind, kws, aux_outs=None, montgomery=False):
    """The assignments of several descriptions of one circuit in ONE native call (host.GROUP at a time is what the callers use):
    Spend and Convert witnesses run their Merkle blocks side by side.  -> list of (inputs, aux)."""
    if kind == "spend":
        res = H.spend_assignments([_spend_item(kw) for kw in kws], aux_outs=aux_outs, montgomery=montgomery)
    elif kind == "convert":
        res = H.convert_assignments([_convert_item(kw) for kw in kws], aux_outs=aux_outs, montgomery=montgomery)
    else:
        return [assignment(kind, kw, aux_out=aux_outs[j] if aux_outs else None) for j, kw in enumerate(kws)]
    for r in res:
        if isinstance(r, Exception):
            raise r
    return [(r[0], r[1]) for r in res]


def instances(kind, n, first_seed=0, threads=None, alloc=None, timing=None, montgomery=False):
    """n independent instances of circuit `kind` -> list of (inputs, aux).  alloc(kind) -> aux buffer (e.g. page-locked memory);
    timing: dict receiving per-instance synthesis milliseconds (description + assignment, one thread each; the assignments of
    host.GROUP instances are one native call, its time shared out evenly).  montgomery: aux as Montgomery residues (the in-memory form of
    blst_fr: masp_hip_job.aux_form = 1) — the synthesizer then writes straight into the buffer."""
    threads = threads or H.effective_cpus()
    # the descriptions first (python + small native calls: the instance's keys, its note, the root of its path), then the witnesses:
    # `timing` gets the wall time of each phase, so that "witnesses per second on all threads" means the synthesizer
    t_d = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        all_kws = list(ex.map(lambda k: description(kind, first_seed + k)[1], range(n)))
    describe_wall = time.perf_counter() - t_d

    def group(lo):
        ks = range(lo, min(n, lo + H.GROUP))
        t0 = time.perf_counter()
        kws = [all_kws[k] for k in ks]
        t1 = time.perf_counter()
        if kind == "output":
            out = [assignment(kind, kw, aux_out=alloc(kind) if alloc else None) if not montgomery else
                   H.output_assignment(kw["esk"], *kw["payment_address"], kw["rcm"], kw["asset_type"], kw["value"], kw["rcv"],
                                       aux_out=alloc(kind) if alloc else None, montgomery=True)[:2] for kw in kws]
        else:
            out = assignments(kind, kws, aux_outs=[alloc(kind) for _ in kws] if alloc else None, montgomery=montgomery)
        t2 = time.perf_counter()
        return out, (t1 - t0) * 1e3 / len(ks), (t2 - t1) * 1e3 / len(ks)

    t_s = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(group, range(0, n, H.GROUP)))
    synth_wall = time.perf_counter() - t_s
    if timing is not None:
        timing.setdefault(kind, {"describe_ms": [], "synthesize_ms": [], "describe_wall_s": 0.0, "synthesize_wall_s": 0.0, "instances": 0})
        for out, d_ms, s_ms in res:
            timing[kind]["describe_ms"] += [describe_wall * 1e3 * threads / max(n, 1)] * len(out)
            timing[kind]["synthesize_ms"] += [s_ms] * len(out)
        timing[kind]["describe_wall_s"] += describe_wall
        timing[kind]["synthesize_wall_s"] += synth_wall
        timing[kind]["instances"] += n
    return [x for out, _, _ in res for x in out]

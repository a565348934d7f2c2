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


def instances(kind, n, first_seed=0, threads=None, alloc=None, timing=None):
    """n independent instances of circuit `kind` -> list of (inputs, aux).  alloc(kind) -> aux buffer (e.g. page-locked memory);
    timing: dict receiving per-instance synthesis milliseconds (description + assignment, one thread each)."""
    threads = threads or H.effective_cpus()

    def one(k):
        t0 = time.perf_counter()
        _, kw = description(kind, first_seed + k)
        t1 = time.perf_counter()
        out = assignment(kind, kw, aux_out=alloc(kind) if alloc else None)
        t2 = time.perf_counter()
        return out, (t1 - t0) * 1e3, (t2 - t1) * 1e3

    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(one, range(n)))
    if timing is not None:
        timing.setdefault(kind, {"describe_ms": [], "synthesize_ms": []})
        timing[kind]["describe_ms"] += [r[1] for r in res]
        timing[kind]["synthesize_ms"] += [r[2] for r in res]
    return [r[0] for r in res]

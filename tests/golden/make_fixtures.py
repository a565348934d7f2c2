#!/usr/bin/env python3
"""Extracts the reference's own known-answer DATA (inputs and expected outputs, no code) into tests/golden/*.json.
Run in the build container, where /root/reference exists; the JSON files are committed and are what the tests read.

Sources (all under /root/reference):
  masp_primitives/src/test_vectors/pedersen_hash_vectors.rs          37 Pedersen-hash vectors (checked at sapling/pedersen_hash.rs:133-154)
  masp_primitives/src/constants.rs:50-251                            the 5 fixed + 6 Pedersen generators (derivations pinned at :323-374)
  masp_proofs/src/circuit/sapling.rs:783-817                         10 value-commitment (u, v) KATs
  masp_primitives/src/test_vectors/note_encryption.rs                note-commitment vectors (cmu checked at sapling/note_encryption.rs:1357-1360)
  masp_proofs/src/circuit/{sapling,convert}.rs, masp_proofs/src/lib.rs  pinned circuit hashes / counts / parameter-file sizes
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def read(p):
    return open(os.path.join(REF, p)).read()


def pedersen():
    s = read("masp_primitives/src/test_vectors/pedersen_hash_vectors.rs")
    out = []
    for m in re.finditer(r"TestVector \{\s*personalization: Personalization::(\w+)(?:\((\d+)\))?,\s*input_bits: vec!\[(.*?)\],\s*hash_u: \"Scalar\(0x([0-9a-f]+)\)\",\s*hash_v: \"Scalar\(0x([0-9a-f]+)\)\"", s, re.S):
        kind, depth, bits, u, v = m.groups()
        out.append({"personalization": -1 if kind == "NoteCommitment" else int(depth),
                    "input_bits": [int(x) for x in re.findall(r"[01]", bits)], "u": u, "v": v})
    assert len(out) == 37
    return out


def generators():
    s = read("masp_primitives/src/constants.rs")
    names = ["proof_generation_key_generator", "note_commitment_randomness_generator", "nullifier_position_generator",
             "value_commitment_randomness_generator", "spending_key_generator"]
    out = {}
    limb = r"0x([0-9a-f_]+),\s*0x([0-9a-f_]+),\s*0x([0-9a-f_]+),\s*0x([0-9a-f_]+)"
    pt = r"from_raw_unchecked\(\s*bls12_381::Scalar::from_u64s_le\(&\[\s*" + limb + r",?\s*\]\)\s*\.unwrap\(\),\s*bls12_381::Scalar::from_u64s_le\(&\[\s*" + limb + r",?\s*\]\)"

    def val(g):
        return sum(int(x.replace("_", ""), 16) << (64 * i) for i, x in enumerate(g))

    for n in names:
        i = s.index("pub fn %s()" % n)
        m = re.search(pt, s[i:], re.S)
        out[n] = {"u": "%064x" % val(m.groups()[:4]), "v": "%064x" % val(m.groups()[4:])}
    i = s.index("pub fn pedersen_hash_generators()")
    j = s.index("pub const PEDERSEN_HASH_CHUNKS_PER_GENERATOR")
    ped = [{"u": "%064x" % val(m.groups()[:4]), "v": "%064x" % val(m.groups()[4:])} for m in re.finditer(pt, s[i:j], re.S)]
    assert len(ped) == 6
    out["pedersen_hash_generators"] = ped
    return out


def value_commitments():
    s = read("masp_proofs/src/circuit/sapling.rs")
    i = s.index("fn test_input_circuit_with_bls12_381_external_test_vectors")
    us = re.search(r"expected_commitment_us = \[(.*?)\];", s[i:], re.S).group(1)
    vs = re.search(r"expected_commitment_vs = \[(.*?)\];", s[i:], re.S).group(1)
    us, vs = re.findall(r"\"(\d+)\"", us), re.findall(r"\"(\d+)\"", vs)
    assert len(us) == len(vs) == 10
    return {"asset_identifier": "734f0ec56f731e02cc737e6b693db52b821f6f6e4cd7fe3c764353f263669fbe",
            "comment": "value i, rcv = 1000 * (i + 1)", "u": us, "v": vs}


def note_vectors():
    s = read("masp_primitives/src/test_vectors/note_encryption.rs")
    out = []
    for blk in s.split("TestVector {")[2:]:
        def arr(name):
            m = re.search(name + r": \[(.*?)\],\s*\n", blk, re.S)
            return bytes(int(x, 16) for x in re.findall(r"0x([0-9a-f]{2})", m.group(1))).hex()
        v = int(re.search(r"\bv: (\d+),", blk).group(1))
        out.append({"ivk": arr("ivk"), "default_d": arr("default_d"), "default_pk_d": arr("default_pk_d"), "v": v, "rcm": arr("rcm"),
                    "cv": arr(r"\bcv"), "cmu": arr("cmu"), "esk": arr("esk"), "epk": arr("epk")})
    assert len(out) == 10
    return {"asset_identifier": b"testtesttesttesttesttesttesttest".hex(), "vectors": out}


def circuits():
    sap = read("masp_proofs/src/circuit/sapling.rs")
    conv = read("masp_proofs/src/circuit/convert.rs")
    lib = read("masp_proofs/src/lib.rs")
    hashes = re.findall(r"cs\.hash\(\),\s*\"([0-9a-f]{64})\"", sap)
    ch = re.findall(r"cs\.hash\(\),\s*\"([0-9a-f]{64})\"", conv)
    sizes = {k: int(v) for k, v in re.findall(r"const MASP_(SPEND|OUTPUT|CONVERT)_BYTES: u64 = (\d+);", lib)}
    assert hashes[0] == hashes[1]
    return {"spend": {"hash": hashes[0], "constraints": 100637, "inputs": 8, "params_file_bytes": sizes["SPEND"]},
            "output": {"hash": hashes[2], "constraints": 31205, "inputs": 6, "params_file_bytes": sizes["OUTPUT"]},
            "convert": {"hash": ch[0], "constraints": 47358, "inputs": 4, "params_file_bytes": sizes["CONVERT"]},
            "mpc_transcript_bytes": 1366052}


if __name__ == "__main__":
    data = {"pedersen_hash_vectors.json": pedersen(), "generators.json": generators(), "value_commitments.json": value_commitments(),
            "note_vectors.json": note_vectors(), "circuits.json": circuits()}
    for name, d in data.items():
        json.dump(d, open(os.path.join(OUT, name), "w"), indent=0 if name.startswith("pedersen") else 1)
        print(name, os.path.getsize(os.path.join(OUT, name)), "bytes")

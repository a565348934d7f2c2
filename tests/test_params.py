"""Parameter-file checks of the reference (masp_proofs/src/lib.rs:60-76,278-487): sizes, BLAKE2b-512 digest over body +
transcript, truncation.  No GPU: the decode to device tables is covered by tests/test_gpu_parity.py."""
import hashlib
import os

import pytest

from masp_amd import params as P


def fake_body(counts=(3, 7, 5, 4, 2, 2), fill=b"\x11"):
    out = fill * 864
    for c, size in zip(counts, (96, 96, 96, 96, 96, 192)):
        out += c.to_bytes(4, "big") + fill * (c * size)
    return out


def test_constants_are_the_reference_values():
    # lib.rs:70-76
    assert (P.MASP_SPEND_BYTES, P.MASP_OUTPUT_BYTES, P.MASP_CONVERT_BYTES) == (49848572, 16398620, 22570940)
    assert P.MASP_SPEND_HASH.startswith("196e7c717f25e166") and len(P.MASP_SPEND_HASH) == 128
    assert P.MASP_OUTPUT_HASH.startswith("eafc3b1746cccc8b") and P.MASP_CONVERT_HASH.startswith("dc4aaf3c3ce056ab")
    # SURVEY.md App. C size equations: body + 1 366 052-byte transcript
    from masp_amd.synthetic import SHAPES
    for kind, total in (("spend", P.MASP_SPEND_BYTES), ("output", P.MASP_OUTPUT_BYTES), ("convert", P.MASP_CONVERT_BYTES)):
        n_inputs, n_aux, n_constraints, na_aux, nb = SHAPES[kind][:5]
        m = 1 << (n_constraints + n_inputs - 1).bit_length()
        body = 864 + 6 * 4 + 96 * (n_inputs + (m - 1) + n_aux + (na_aux + n_inputs) + (nb + 1)) + 192 * (nb + 1)
        assert total - body == 1366052, kind


def test_body_length_and_transcript():
    body = fake_body()
    assert P.body_length(body) == len(body)
    assert P.body_length(body + b"transcript bytes") == len(body)
    with pytest.raises(P.ParameterError):
        P.body_length(body[:-1])
    with pytest.raises(P.ParameterError):
        P.body_length(body[:500])
    with pytest.raises(P.ParameterError):
        P.parse_parameters(b"\0", b"\0", b"\0")      # prover.rs:71-79 `from_bytes(&[0u8], ..)` panics


def test_digest_covers_the_whole_stream(tmp_path):
    blobs = {k: fake_body(fill=bytes([i + 1])) + b"mpc transcript %d" % i for i, k in enumerate(P.KINDS)}
    exp = {k: P.Expected("masp-%s.params" % k, hashlib.blake2b(v, digest_size=64).hexdigest(), len(v)) for k, v in blobs.items()}
    got = P.parse_parameters(blobs["spend"], blobs["output"], blobs["convert"], expected=exp)
    assert got.spend == blobs["spend"] and got.convert == blobs["convert"]
    # one flipped transcript byte -> digest mismatch
    bad = blobs["output"][:-1] + b"X"
    with pytest.raises(P.ParameterError, match="failed validation"):
        P.parse_parameters(blobs["spend"], bad, blobs["convert"], expected=exp)
    # against the real MPC digests nothing synthetic can pass
    with pytest.raises(P.ParameterError):
        P.parse_parameters(blobs["spend"], blobs["output"], blobs["convert"])
    # files: size is checked first
    paths = []
    for k in P.KINDS:
        p = tmp_path / exp[k].name
        p.write_bytes(blobs[k])
        paths.append(str(p))
    assert P.load_parameters(*paths, expected=exp).output == blobs["output"]
    (tmp_path / exp["convert"].name).write_bytes(blobs["convert"] + b"!")
    with pytest.raises(P.ParameterError, match="bytes"):
        P.load_parameters(*paths, expected=exp)
    with pytest.raises(P.ParameterError, match="expected: 49848572 bytes"):
        P.load_parameters(*paths)


def test_default_folder():
    d = P.default_params_folder()
    assert d is None or os.path.basename(d) in (".masp-params", "MASPParams")

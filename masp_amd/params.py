"""Parameter-file handling of the reference, restated: names, sizes, BLAKE2b-512 digests and the load / parse entry points.

    masp_proofs/src/lib.rs:60-76      file names, expected digests and sizes of the MPC parameter files
    masp_proofs/src/lib.rs:100-108    default_params_folder
    masp_proofs/src/lib.rs:278-328    load_parameters: sizes first (cheap), then parse + hash
    masp_proofs/src/lib.rs:333-403    parse_parameters: `Parameters::read(_, false)` on a hashing reader, then the rest of
                                      the stream (the MPC transcript) is drained so that the digest equals `b2sum file`
    masp_proofs/src/hashreader.rs     BLAKE2b-512 over every byte read

The reference panics on any mismatch; here that is `ParameterError`.  The MPC files cannot be downloaded in the build
image (no network; `download-params` is a control-plane feature outside SURVEY.md §8), so the benches and tests run on
parameters generated from known toxic waste and pass `expected=None` (no digest to compare against) — the decode and
length-invariant checks in `masp_hip_circuit_load` still run.
"""
import hashlib
import os
import sys
from collections import namedtuple

MASP_SPEND_NAME = "masp-spend.params"
MASP_OUTPUT_NAME = "masp-output.params"
MASP_CONVERT_NAME = "masp-convert.params"

MASP_SPEND_HASH = "196e7c717f25e16653431559ce2c8816e750a4490f98696e3c031efca37e25e0647182b7b013660806db11eb2b1e365fb2d6a0f24dbbd9a4a8314fef10a7cba2"
MASP_OUTPUT_HASH = "eafc3b1746cccc8b9eed2b69395692c5892f6aca83552a07dceb2dcbaa64dcd0e22434260b3aa3b049b633a08b008988cbe0d31effc77e2bc09bfab690a23724"
MASP_CONVERT_HASH = "dc4aaf3c3ce056ab448b6c4a7f43c1d68502c2902ea89ab8769b1524a2e8ace9a5369621a73ee1daa52aec826907a19974a37874391cf8f11bbe0b0420de1ab7"
MASP_SPEND_BYTES = 49848572
MASP_CONVERT_BYTES = 22570940
MASP_OUTPUT_BYTES = 16398620

Expected = namedtuple("Expected", "name hash bytes")
EXPECTED = {
    "spend": Expected(MASP_SPEND_NAME, MASP_SPEND_HASH, MASP_SPEND_BYTES),
    "output": Expected(MASP_OUTPUT_NAME, MASP_OUTPUT_HASH, MASP_OUTPUT_BYTES),
    "convert": Expected(MASP_CONVERT_NAME, MASP_CONVERT_HASH, MASP_CONVERT_BYTES),
}
KINDS = ("spend", "output", "convert")


class ParameterError(Exception):
    """Where the reference panics: wrong size, wrong digest, undecodable bytes."""


def default_params_folder():
    """lib.rs:100-108."""
    home = os.path.expanduser("~")
    if not home or home == "~":
        return None
    if sys.platform == "darwin":
        return os.path.join(home, "Library", "Application Support", "MASPParams")
    if sys.platform.startswith("win"):
        return os.path.join(os.environ.get("APPDATA", home), "MASPParams")
    return os.path.join(home, ".masp-params")


def verify_file_size(path, expected_bytes, name, source=None):
    """lib.rs:409-430: filesystem metadata only."""
    size = os.stat(path).st_size
    if size != expected_bytes:
        raise ParameterError("%s failed validation:\nexpected: %d bytes,\nactual:   %d bytes from %r" % (name, expected_bytes, size, source or str(path)))


def body_length(data):
    """Length of the `Parameters` body at the head of `data` (bellman wire format):
    vk (7 fixed points = 864 B, then u32-BE count + ic), then five u32-BE counted vectors h, l, a, b_g1 (96 B points)
    and b_g2 (192 B).  Whatever follows is the MPC transcript.  Raises on truncation."""
    n, off = len(data), 96 + 96 + 192 + 192 + 96 + 192
    for size in (96, 96, 96, 96, 96, 192):
        if off + 4 > n:
            raise ParameterError("couldn't deserialize parameters: truncated at offset %d" % off)
        count = int.from_bytes(data[off:off + 4], "big")
        off += 4 + count * size
        if off > n:
            raise ParameterError("couldn't deserialize parameters: vector of %d points runs past the end" % count)
    return off


def verify_hash(data, expected_hash, expected_bytes, name, source="a file"):
    """lib.rs:439-487: the digest is over the WHOLE stream (body + transcript)."""
    digest = hashlib.blake2b(data, digest_size=64).hexdigest()
    if digest != expected_hash:
        raise ParameterError("%s failed validation:\nexpected: %s hashing %d bytes,\nactual:   %s hashing %d bytes from %r"
                             % (name, expected_hash, expected_bytes, digest, len(data), source))


MaspParameterBytes = namedtuple("MaspParameterBytes", "spend output convert")


def parse_parameters(spend, output, convert, expected=EXPECTED):
    """lib.rs:333-403 on byte strings.  Returns the three byte strings (decoding to device tables happens in
    `masp_hip_circuit_load`, verifying-key preparation in `host.PreparedVerifyingKey`).  `expected=None` skips the digest
    comparison (synthetic parameters)."""
    blobs = {k: v if isinstance(v, bytes) else memoryview(v).cast("B").tobytes() for k, v in zip(KINDS, (spend, output, convert))}
    for kind in KINDS:
        try:
            body_length(blobs[kind])
        except ParameterError as e:
            raise ParameterError("couldn't deserialize MASP %s parameters file: %s" % (kind, e)) from None
    if expected is not None:
        for kind in KINDS:
            e = expected[kind]
            verify_hash(blobs[kind], e.hash, e.bytes, e.name)
    return MaspParameterBytes(blobs["spend"], blobs["output"], blobs["convert"])


def load_parameters(spend_path, output_path, convert_path, expected=EXPECTED):
    """lib.rs:278-328: file sizes are checked before any large read."""
    paths = dict(zip(KINDS, (spend_path, output_path, convert_path)))
    if expected is not None:
        for kind in KINDS:
            verify_file_size(paths[kind], expected[kind].bytes, "masp " + kind)
    blobs = []
    for kind in KINDS:
        with open(paths[kind], "rb") as f:
            blobs.append(f.read())
    return parse_parameters(*blobs, expected=expected)

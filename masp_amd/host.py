"""ctypes binding of libmasp_host.so: the host-side witness generator and native primitives
(masp_amd/csrc/host).  CPU-only code; the Groth16 hot path itself lives in libmasp_hip.so."""
import ctypes as C
import os

import numpy as np

from .r1cs import R1cs

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

SPEND, OUTPUT, CONVERT = 0, 1, 2
KINDS = {"spend": SPEND, "output": OUTPUT, "convert": CONVERT}
JUBJUB_ORDER = 6554484396890773809930967563523245729705921265872317281365359162392183254199
FR_MODULUS = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
TREE_DEPTH = 32
ERRORS = {1: "invalid encoding", 2: "invalid diversifier", 3: "synthesis error", 4: "assignment does not satisfy the circuit"}


class HostError(RuntimeError):
    def __init__(self, code):
        self.code = code
        super().__init__("masp_host error %d: %s" % (code, ERRORS.get(code, "?")))


def load_library():
    global _lib
    if _lib is None:
        # MASP_HOST_LIBRARY: another build of the same library (tools/sanitize_host.sh runs the CPU tests over an ASan / UBSan build)
        path = os.environ.get("MASP_HOST_LIBRARY") or os.path.join(_HERE, "libmasp_host.so")
        if not os.path.exists(path):
            raise ImportError("libmasp_host.so is not built: run `make -C masp_amd/csrc`")
        L = C.CDLL(path)
        L.masp_host_circuit_setup.restype = C.c_void_p
        L.masp_host_circuit_setup.argtypes = [C.c_int]
        L.masp_host_circuit_free.argtypes = [C.c_void_p]
        L.masp_host_circuit_counts.argtypes = [C.c_void_p, C.c_void_p]
        L.masp_host_circuit_matrix.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.masp_host_circuit_hash.argtypes = [C.c_void_p, C.c_char_p]
        cp, vp, u64 = C.c_char_p, C.c_void_p, C.c_uint64
        L.masp_host_spend_assignment.argtypes = [cp, cp, cp, cp, cp, cp, u64, cp, vp, u64, cp, C.c_int, vp, vp, cp, cp, cp]
        L.masp_host_output_assignment.argtypes = [cp, cp, cp, cp, cp, u64, cp, C.c_int, vp, vp, cp]
        L.masp_host_convert_assignment.argtypes = [cp, u64, cp, vp, u64, cp, C.c_int, vp, vp, cp]
        L.masp_host_spend_assignments.argtypes = [C.c_size_t, vp, C.c_int]
        L.masp_host_fr_from_montgomery.argtypes = [vp, vp, C.c_size_t]
        L.masp_host_fr_from_montgomery.restype = None
        L.masp_host_convert_assignments.argtypes = [C.c_size_t, vp, C.c_int]
        L.masp_host_generator.argtypes = [C.c_int, cp]
        L.masp_host_pedersen_hash.argtypes = [C.c_int, vp, C.c_size_t, cp]
        L.masp_host_asset_identifier.argtypes = [cp, C.c_size_t, cp]
        L.masp_host_asset_generator.argtypes = [cp, cp]
        L.masp_host_value_commitment.argtypes = [cp, u64, cp, cp, cp]
        L.masp_host_note_cmu.argtypes = [cp, u64, cp, cp, cp, cp]
        L.masp_host_merkle_hash.argtypes = [C.c_uint, cp, cp, cp]
        L.masp_host_jubjub_mul.argtypes = [cp, cp, cp]
        L.masp_host_convert_cmu.argtypes = [cp, cp]
        L.masp_host_vk_prepare.restype = vp
        L.masp_host_vk_prepare.argtypes = [vp, C.c_size_t]
        L.masp_host_vk_free.argtypes = [vp]
        L.masp_host_vk_verify.argtypes = [vp, cp, cp, C.c_uint32]
        L.masp_host_vk_verify_batch.argtypes = [vp, C.c_size_t, cp, cp, C.c_uint32, cp]
        L.masp_host_point_uv.argtypes = [cp, cp]
        L.masp_host_jubjub_add.argtypes = [cp, cp, C.c_int, cp]
        L.masp_host_jubjub_sum.argtypes = [cp, cp, C.c_size_t, cp, cp]
        L.masp_host_spend_leaf.argtypes = [cp, cp, cp, cp, cp, u64, cp, cp]
        L.masp_host_allowed_conversion.argtypes = [C.c_size_t, cp, cp, cp]
        _lib = L
    return _lib


def effective_cpus():
    """Host threads worth starting: the scheduler affinity, capped by a cgroup CPU quota when one is set (a container
    that sees 256 logical CPUs may be entitled to 16 of them; oversubscribing it only adds contention)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


_circuits = {}


def circuit(kind):
    """Static R1CS of a MASP circuit ("spend" | "output" | "convert") and its TestConstraintSystem hash."""
    if kind not in _circuits:
        L = load_library()
        h = L.masp_host_circuit_setup(KINDS[kind])
        if not h:
            raise RuntimeError("circuit setup failed")
        cnt = (C.c_uint32 * 6)()
        L.masp_host_circuit_counts(h, cnt)
        n_in, n_aux, n_con = cnt[0], cnt[1], cnt[2]
        mats = []
        for mi in range(3):
            nnz = cnt[3 + mi]
            rp = np.zeros(n_con + 1, np.uint32)
            col = np.zeros(nnz, np.uint32)
            coef = np.zeros((nnz, 32), np.uint8)
            L.masp_host_circuit_matrix(h, mi, rp.ctypes.data, col.ctypes.data, coef.ctypes.data)
            mats.append((rp, col, coef))
        buf = C.create_string_buffer(65)
        L.masp_host_circuit_hash(h, buf)
        L.masp_host_circuit_free(h)
        _circuits[kind] = (R1cs(n_in, n_aux, n_con, mats), buf.value.decode())
    return _circuits[kind]


def _b(x, n=32):
    if isinstance(x, int):
        return x.to_bytes(n, "little")
    x = bytes(x)
    assert len(x) == n, (len(x), n)
    return x


def _path(siblings):
    assert len(siblings) == TREE_DEPTH
    return np.frombuffer(b"".join(_b(s) for s in siblings), dtype=np.uint8).copy()


def _check(rc):
    if rc:
        raise HostError(rc)


def _aux_buffer(cs, aux_out):
    """The aux assignment is written into `aux_out` when given (e.g. page-locked memory from hip.Context.host_alloc)."""
    if aux_out is None:
        return np.zeros((cs.n_aux, 32), np.uint8)
    assert aux_out.dtype == np.uint8 and aux_out.shape == (cs.n_aux, 32) and aux_out.flags["C_CONTIGUOUS"]
    return aux_out


def spend_assignment(ak, nsk, diversifier, rcm, ar, asset_identifier, value, anchor, path_siblings, position, rcv, check=False,
                     aux_out=None, montgomery=False):
    """-> (inputs u8[8,32], aux u8[100497,32], cv, rk, nf)   — SaplingProvingContext::spend_proof up to the prover call.
    montgomery: aux leaves as Montgomery residues (blst_fr memory; masp_hip_job.aux_form = 1) instead of canonical bytes."""
    L = load_library()
    cs, _ = circuit("spend")
    inputs = np.zeros((cs.n_inputs, 32), np.uint8)
    aux = _aux_buffer(cs, aux_out)
    cv, rk, nf = (C.create_string_buffer(32) for _ in range(3))
    p = _path(path_siblings)
    _check(L.masp_host_spend_assignment(_b(ak), _b(nsk), _b(diversifier, 11), _b(rcm), _b(ar), _b(asset_identifier), value, _b(anchor),
                                        p.ctypes.data, position, _b(rcv), (1 if check else 0) | (2 if montgomery else 0), inputs.ctypes.data,
                                        aux.ctypes.data, cv, rk, nf))
    return inputs, aux, cv.raw, rk.raw, nf.raw


def output_assignment(esk, diversifier, pk_d, rcm, asset_identifier, value, rcv, check=False, aux_out=None, montgomery=False):
    L = load_library()
    cs, _ = circuit("output")
    inputs = np.zeros((cs.n_inputs, 32), np.uint8)
    aux = _aux_buffer(cs, aux_out)
    cv = C.create_string_buffer(32)
    _check(L.masp_host_output_assignment(_b(esk), _b(diversifier, 11), _b(pk_d), _b(rcm), _b(asset_identifier), value, _b(rcv),
                                         (1 if check else 0) | (2 if montgomery else 0), inputs.ctypes.data, aux.ctypes.data, cv))
    return inputs, aux, cv.raw


def convert_assignment(generator, value, anchor, path_siblings, position, rcv, check=False, aux_out=None, montgomery=False):
    L = load_library()
    cs, _ = circuit("convert")
    inputs = np.zeros((cs.n_inputs, 32), np.uint8)
    aux = _aux_buffer(cs, aux_out)
    cv = C.create_string_buffer(32)
    p = _path(path_siblings)
    _check(L.masp_host_convert_assignment(_b(generator), value, _b(anchor), p.ctypes.data, position, _b(rcv),
                                          (1 if check else 0) | (2 if montgomery else 0), inputs.ctypes.data, aux.ctypes.data, cv))
    return inputs, aux, cv.raw


def aux_from_montgomery(aux):
    """u8[n,32] of Montgomery residues (what the synthesizers write with montgomery=True) -> u8[n,32] canonical little-endian values."""
    L = load_library()
    aux = np.ascontiguousarray(aux, dtype=np.uint8)
    out = np.zeros_like(aux)
    L.masp_host_fr_from_montgomery(aux.ctypes.data, out.ctypes.data, aux.shape[0])
    return out


# ---- several witnesses per call: their Merkle blocks run in lockstep (csrc/host/circuits.h merkle_block_batch) ----
GROUP = 16      # witnesses per call: 3 x 16 chains of affine additions share every field inversion; 256 witnesses = 16 calls


class _SpendJob(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ak", "nsk", "diversifier", "rcm", "ar", "asset_identifier")] + [("value", C.c_uint64)] + \
               [(n, C.c_void_p) for n in ("anchor", "path_siblings")] + [("position", C.c_uint64), ("rcv", C.c_void_p)] + \
               [(n, C.c_void_p) for n in ("inputs", "aux", "cv_out", "rk_out", "nf_out")] + [("rc", C.c_int)]


class _ConvertJob(C.Structure):
    _fields_ = [("generator", C.c_void_p), ("value", C.c_uint64), ("anchor", C.c_void_p), ("path_siblings", C.c_void_p), ("position", C.c_uint64),
                ("rcv", C.c_void_p), ("inputs", C.c_void_p), ("aux", C.c_void_p), ("cv_out", C.c_void_p), ("rc", C.c_int)]


def _pin(keep, data, n=32):
    buf = C.create_string_buffer(_b(data, n), n)
    keep.append(buf)
    return C.addressof(buf)


def spend_assignments(items, check=False, aux_outs=None, montgomery=False):
    """items: list of (ak, nsk, diversifier, rcm, ar, asset_identifier, value, anchor, path_siblings, position, rcv) — the arguments of
    spend_assignment — synthesised in ONE native call (masp_host_spend_assignments: the Merkle blocks of the witnesses run side
    by side, ~1.8x the witnesses per second of the one-by-one call).  -> list of (inputs, aux, cv, rk, nf) or HostError instances."""
    L = load_library()
    cs, _ = circuit("spend")
    n = len(items)
    jobs = (_SpendJob * n)()
    keep, outs = [], []
    for j, (ak, nsk, d, rcm, ar, ident, value, anchor, sib, pos, rcv) in enumerate(items):
        inputs = np.zeros((cs.n_inputs, 32), np.uint8)
        aux = _aux_buffer(cs, aux_outs[j] if aux_outs else None)
        path = _path(sib)
        keep.append(path)
        o = C.create_string_buffer(96)
        J = jobs[j]
        J.ak, J.nsk, J.diversifier, J.rcm, J.ar, J.asset_identifier = (_pin(keep, ak), _pin(keep, nsk), _pin(keep, d, 11), _pin(keep, rcm),
                                                                       _pin(keep, ar), _pin(keep, ident))
        J.value, J.anchor, J.path_siblings, J.position, J.rcv = value, _pin(keep, anchor), path.ctypes.data, pos, _pin(keep, rcv)
        J.inputs, J.aux = inputs.ctypes.data, aux.ctypes.data
        J.cv_out, J.rk_out, J.nf_out = C.addressof(o), C.addressof(o) + 32, C.addressof(o) + 64
        outs.append((inputs, aux, o))
    L.masp_host_spend_assignments(n, jobs, (1 if check else 0) | (2 if montgomery else 0))
    return [HostError(jobs[j].rc) if jobs[j].rc else (i, a, o.raw[:32], o.raw[32:64], o.raw[64:]) for j, (i, a, o) in enumerate(outs)]


def convert_assignments(items, check=False, aux_outs=None, montgomery=False):
    """items: list of (generator, value, anchor, path_siblings, position, rcv) -> list of (inputs, aux, cv) or HostError instances."""
    L = load_library()
    cs, _ = circuit("convert")
    n = len(items)
    jobs = (_ConvertJob * n)()
    keep, outs = [], []
    for j, (gen, value, anchor, sib, pos, rcv) in enumerate(items):
        inputs = np.zeros((cs.n_inputs, 32), np.uint8)
        aux = _aux_buffer(cs, aux_outs[j] if aux_outs else None)
        path = _path(sib)
        keep.append(path)
        o = C.create_string_buffer(32)
        J = jobs[j]
        J.generator, J.value, J.anchor, J.path_siblings, J.position, J.rcv = _pin(keep, gen), value, _pin(keep, anchor), path.ctypes.data, pos, _pin(keep, rcv)
        J.inputs, J.aux, J.cv_out = inputs.ctypes.data, aux.ctypes.data, C.addressof(o)
        outs.append((inputs, aux, o))
    L.masp_host_convert_assignments(n, jobs, (1 if check else 0) | (2 if montgomery else 0))
    return [HostError(jobs[j].rc) if jobs[j].rc else (i, a, o.raw) for j, (i, a, o) in enumerate(outs)]


class PreparedVerifyingKey:
    """= bellman `prepare_verifying_key(&params.vk)` (lib.rs:391-393) + `verify_proof` (sapling/prover.rs:148,266)."""

    def __init__(self, params):
        buf = np.frombuffer(params, dtype=np.uint8) if isinstance(params, (bytes, bytearray, memoryview)) else np.ascontiguousarray(params, dtype=np.uint8)
        n_ic = int.from_bytes(buf[864:868].tobytes(), "big")
        self._buf = buf[:868 + 96 * n_ic].copy()
        self._L = load_library()
        self._h = self._L.masp_host_vk_prepare(self._buf.ctypes.data, self._buf.size)
        if not self._h:
            raise ValueError("malformed verifying key")

    def verify(self, proof, public_inputs):
        """public_inputs: ints or 32-byte LE values, excluding ONE -> True / False"""
        pi = b"".join(_b(x) for x in public_inputs)
        rc = self._L.masp_host_vk_verify(self._h, bytes(proof), pi, len(public_inputs))
        if rc < 0:
            return False
        return rc == 1

    def verify_batch(self, proofs, public_inputs, randomness=None):
        """= bellman `verify_proofs_batch` (sapling/verifier/batch.rs:24-31): one random linear combination, n + 2 Miller
        loops and a single final exponentiation for n proofs.  public_inputs: one list per proof.  True iff all verify
        (up to 2^-128); False says at least one is invalid, not which."""
        import secrets
        n = len(proofs)
        if n == 0:
            return True
        k = len(public_inputs[0])
        if any(len(pi) != k for pi in public_inputs) or len(public_inputs) != n:
            return False
        pi = b"".join(_b(x) for row in public_inputs for x in row)
        z = randomness if randomness is not None else secrets.token_bytes(16 * n)
        assert len(z) == 16 * n
        return self._L.masp_host_vk_verify_batch(self._h, n, b"".join(bytes(p) for p in proofs), pi, k, z) == 1

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.masp_host_vk_free(self._h)
            self._h = None


# ---- native primitives ----
GENERATOR_NAMES = ["proof_generation_key_generator", "note_commitment_randomness_generator", "nullifier_position_generator",
                   "value_commitment_randomness_generator", "spending_key_generator"]


def generator_uv(which):
    out = C.create_string_buffer(64)
    load_library().masp_host_generator(which, out)
    return int.from_bytes(out.raw[:32], "little"), int.from_bytes(out.raw[32:], "little")


def point_bytes(u, v):
    return (v | ((u & 1) << 255)).to_bytes(32, "little")


def pedersen_hash(personalization, bits):
    """personalization: -1 for NoteCommitment, else MerkleTree depth; bits: iterable of 0/1 -> (u, v)"""
    b = np.array(list(bits), dtype=np.uint8)
    out = C.create_string_buffer(64)
    load_library().masp_host_pedersen_hash(personalization, b.ctypes.data, b.size, out)
    return int.from_bytes(out.raw[:32], "little"), int.from_bytes(out.raw[32:], "little")


def asset_identifier(name):
    out = C.create_string_buffer(32)
    if load_library().masp_host_asset_identifier(bytes(name), len(name), out):
        raise ValueError("no valid asset identifier")
    return out.raw


def asset_generator(identifier):
    out = C.create_string_buffer(32)
    if load_library().masp_host_asset_generator(_b(identifier), out):
        raise ValueError("invalid asset identifier")
    return out.raw


def value_commitment(identifier, value, rcv):
    out, uv = C.create_string_buffer(32), C.create_string_buffer(64)
    if load_library().masp_host_value_commitment(_b(identifier), value, _b(rcv), out, uv):
        raise ValueError("invalid asset identifier")
    return out.raw, int.from_bytes(uv.raw[:32], "little"), int.from_bytes(uv.raw[32:], "little")


def note_cmu(identifier, value, diversifier, pk_d, rcm):
    out = C.create_string_buffer(32)
    if load_library().masp_host_note_cmu(_b(identifier), value, _b(diversifier, 11), _b(pk_d), _b(rcm), out):
        raise ValueError("invalid note")
    return out.raw


def merkle_hash(depth, lhs, rhs):
    out = C.create_string_buffer(32)
    if load_library().masp_host_merkle_hash(depth, _b(lhs), _b(rhs), out):
        raise ValueError("non-canonical node")
    return out.raw


def jubjub_mul(point, scalar):
    out = C.create_string_buffer(32)
    if load_library().masp_host_jubjub_mul(_b(point), _b(scalar), out):
        raise ValueError("invalid point")
    return out.raw


def point_uv(point):
    out = C.create_string_buffer(64)
    if load_library().masp_host_point_uv(_b(point), out):
        raise ValueError("invalid point")
    return int.from_bytes(out.raw[:32], "little"), int.from_bytes(out.raw[32:], "little")


def multipack(data):
    """bellman multipack::compute_multipacking(bytes_to_bits_le(data)): 254-bit little-endian chunks (sapling/prover.rs:138-139)"""
    # (bit i of the little-endian integer is bit i of bytes_to_bits_le: the chunks are 254-bit digits of that integer — 49 us per
    # nullifier as a list of bits, under the GIL of the thread that is about to start a chunk's self-verification)
    v, nbits = int.from_bytes(bytes(data), "little"), 8 * len(data)
    return [(v >> o) & ((1 << 254) - 1) for o in range(0, nbits, 254)]


def jubjub_add(p, q, subtract=False):
    out = C.create_string_buffer(32)
    if load_library().masp_host_jubjub_add(_b(p), _b(q), 1 if subtract else 0, out):
        raise ValueError("invalid point")
    return out.raw


JUBJUB_IDENTITY = (1).to_bytes(32, "little")   # (u, v) = (0, 1)


def jubjub_sum(points, subtract=None, acc=JUBJUB_IDENTITY):
    """acc + sum of the compressed points (those with a true `subtract` flag subtracted) in one native call."""
    pts = b"".join(_b(p) for p in points)
    flags = bytes(1 if f else 0 for f in subtract) if subtract is not None else None
    out = C.create_string_buffer(32)
    if load_library().masp_host_jubjub_sum(_b(acc), pts, len(points), flags, out):
        raise ValueError("invalid point")
    return out.raw


def convert_cmu(generator):
    out = C.create_string_buffer(32)
    if load_library().masp_host_convert_cmu(_b(generator), out):
        raise ValueError("invalid point")
    return out.raw


class AllowedConversion:
    """= masp_primitives::convert::AllowedConversion (/root/reference/masp_primitives/src/convert.rs:22-84,86-118).

    `AllowedConversion(assets)` mirrors `From<I128Sum>`: assets = iterable of (asset identifier[32], signed 128-bit
    value) — the components of the reference's I128Sum; generator = sum_i sign(v_i) [|v_i| as u64] asset_generator_i,
    cofactor not cleared.  i128::MIN raises ValueError (the reference panics "invalid conversion")."""

    def __init__(self, assets):
        merged = {}
        for ident, value in (assets.items() if isinstance(assets, dict) else assets):
            ident = _b(ident)
            merged[ident] = merged.get(ident, 0) + int(value)          # ValueSum addition merges equal asset types
        self.assets = {k: v for k, v in sorted(merged.items()) if v != 0}
        for v in self.assets.values():
            if not -(1 << 127) <= v < (1 << 127):
                raise OverflowError("value does not fit an i128")
        ids = b"".join(self.assets.keys())
        vals = b"".join(v.to_bytes(16, "little", signed=True) for v in self.assets.values())
        out = C.create_string_buffer(32)
        if load_library().masp_host_allowed_conversion(len(self.assets), ids, vals, out):
            raise ValueError("invalid conversion")
        self.generator = out.raw

    def cmu(self):
        """u-coordinate of PedersenHash(NoteCommitment, repr(generator)) (convert.rs:39-64), 32 bytes LE."""
        return convert_cmu(self.generator)

    commitment = cmu      # Node::from_scalar(self.cmu()) (convert.rs:80-84): the leaf of the conversion tree

    def value_commitment(self, value, randomness):
        """-> the commitment point cv = [value]([8]generator) + [randomness]G_vcr, 32 bytes (convert.rs:70-77, sapling.rs:204-209)."""
        h = jubjub_mul(jubjub_mul(self.generator, 8), value)
        g_rcv = point_bytes(*generator_uv(3))
        return jubjub_add(h, jubjub_mul(g_rcv, randomness))

    @staticmethod
    def uncommitted():
        return 1          # bls12_381::Scalar::ONE (convert.rs:32-36)


def spend_leaf(ak, nsk, diversifier, rcm, asset_identifier, value):
    """(cmu, pk_d) of the note a Spend proves knowledge of: nk = [nsk]G, ivk = CRH(ak, nk), pk_d = [ivk] g_d."""
    cmu, pk_d = C.create_string_buffer(32), C.create_string_buffer(32)
    _check(load_library().masp_host_spend_leaf(_b(ak), _b(nsk), _b(diversifier, 11), _b(rcm), _b(asset_identifier), value, cmu, pk_d))
    return cmu.raw, pk_d.raw


def merkle_root(leaf, path_siblings, position):
    """Root of the depth-32 tree from a leaf, its siblings (leaf level first) and its position."""
    cur = _b(leaf)
    for i, sib in enumerate(path_siblings):
        cur = merkle_hash(i, _b(sib), cur) if (position >> i) & 1 else merkle_hash(i, cur, _b(sib))
    return cur

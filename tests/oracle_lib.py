"""ctypes binding of oracle/_build/liboracle.so — test-side only (see oracle/groth16_oracle.cpp header)."""
import ctypes as C
import os
import subprocess

import numpy as np

import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
_SO = os.environ.get("MASP_ORACLE_LIBRARY") or os.path.join(ROOT, "oracle", "_build", "liboracle.so")   # (override: tools/sanitize_host.sh)


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


from masp_amd.r1cs import R1cs  # noqa: E402,F401  (plain container; identical layout to oracle_r1cs)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_generate_parameters.restype = C.c_size_t
        L.oracle_generate_parameters.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
        L.oracle_r1cs_unsatisfied.restype = C.c_size_t
        L.oracle_params_parse.restype = C.c_void_p
        L.oracle_params_parse.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_params_free.argtypes = [C.c_void_p]
        L.oracle_params_lens.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_create_proof.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p,
                                          C.c_void_p, C.c_void_p]
        L.oracle_closed_form_proof.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p,
                                               C.c_void_p]
        L.oracle_verify_proof.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_void_p, C.c_uint32]
        L.oracle_msm_g1.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_msm_g2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_g1_mul_gen_many.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_g2_mul_gen_many.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_quotient_h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.oracle_ntt.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.oracle_r1cs_eval.argtypes = [C.c_void_p] * 9
        L.oracle_r1cs_unsatisfied.argtypes = [C.c_void_p] * 3
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def fr_op(op, a, b=0):
    out = C.create_string_buffer(32)
    assert lib().oracle_fr_op(op, a.to_bytes(32, "little"), b.to_bytes(32, "little"), out) == 0
    return int.from_bytes(out.raw, "little")


def fp_op(op, a, b=0):
    out = C.create_string_buffer(48)
    assert lib().oracle_fp_op(op, a.to_bytes(48, "little"), b.to_bytes(48, "little"), out) == 0
    return int.from_bytes(out.raw, "little")


def g1_mul_gen(k):
    u, c = C.create_string_buffer(96), C.create_string_buffer(48)
    lib().oracle_g1_mul_gen(k.to_bytes(32, "little"), u, c)
    return u.raw, c.raw


def g2_mul_gen(k):
    u, c = C.create_string_buffer(192), C.create_string_buffer(96)
    lib().oracle_g2_mul_gen(k.to_bytes(32, "little"), u, c)
    return u.raw, c.raw


def g1_mul_gen_many(ks):
    """ks: u8[n,32] -> u8[n,96]"""
    ks = np.ascontiguousarray(ks, dtype=np.uint8)
    out = np.zeros((ks.shape[0], 96), dtype=np.uint8)
    lib().oracle_g1_mul_gen_many(_p(ks), ks.shape[0], _p(out))
    return out


def g2_mul_gen_many(ks):
    ks = np.ascontiguousarray(ks, dtype=np.uint8)
    out = np.zeros((ks.shape[0], 192), dtype=np.uint8)
    lib().oracle_g2_mul_gen_many(_p(ks), ks.shape[0], _p(out))
    return out


def msm_g1(bases, scalars):
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
    out = np.zeros(96, dtype=np.uint8)
    assert lib().oracle_msm_g1(_p(bases), _p(scalars), scalars.shape[0], _p(out)) == 0
    return out.tobytes()


def msm_g2(bases, scalars):
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
    out = np.zeros(192, dtype=np.uint8)
    assert lib().oracle_msm_g2(_p(bases), _p(scalars), scalars.shape[0], _p(out)) == 0
    return out.tobytes()


def quotient_h(a, b, c, logm):
    """a,b,c: u8[nrows,32] -> h u8[m-1,32]"""
    a, b, c = (np.ascontiguousarray(x, dtype=np.uint8) for x in (a, b, c))
    out = np.zeros(((1 << logm) - 1, 32), dtype=np.uint8)
    lib().oracle_quotient_h(_p(a), _p(b), _p(c), a.shape[0], logm, _p(out))
    return out


def ntt(data, logm, inverse=False):
    d = np.ascontiguousarray(data, dtype=np.uint8).copy()
    lib().oracle_ntt(_p(d), logm, 1 if inverse else 0)
    return d


def generate_parameters(cs, toxic):
    """toxic: 5 ints (tau, alpha, beta, gamma, delta) -> params bytes (bellman wire format)"""
    t = b"".join(x.to_bytes(32, "little") for x in toxic)
    need = lib().oracle_generate_parameters(cs.ref, t, None, 0)
    out = np.zeros(need, dtype=np.uint8)
    got = lib().oracle_generate_parameters(cs.ref, t, _p(out), need)
    assert got == need
    return out


def r1cs_eval(cs, inputs, aux):
    n = cs.nrows
    a, b, c = (np.zeros((n, 32), dtype=np.uint8) for _ in range(3))
    da, dbi, dba = np.zeros(cs.n_aux, np.uint8), np.zeros(cs.n_inputs, np.uint8), np.zeros(cs.n_aux, np.uint8)
    lib().oracle_r1cs_eval(cs.ref, _p(inputs), _p(aux), _p(a), _p(b), _p(c), _p(da), _p(dbi), _p(dba))
    return a, b, c, da, dbi, dba


def r1cs_unsatisfied(cs, inputs, aux):
    return lib().oracle_r1cs_unsatisfied(cs.ref, _p(inputs), _p(aux))


class Params:
    def __init__(self, buf):
        self.buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self.h = lib().oracle_params_parse(_p(self.buf), self.buf.size)
        if not self.h:
            raise ValueError("params parse failed")

    def lens(self):
        l = np.zeros(6, dtype=np.uint32)
        lib().oracle_params_lens(self.h, _p(l))
        return dict(zip(["ic", "h", "l", "a", "b_g1", "b_g2"], (int(x) for x in l)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_params_free(self.h)
            self.h = None


def create_proof(params, cs, inputs, aux, r, s, timings=None):
    out = np.zeros(192, dtype=np.uint8)
    t = np.zeros(5, dtype=np.float64)
    rc = lib().oracle_create_proof(params.h, cs.ref, _p(inputs), _p(aux), r.to_bytes(32, "little"),
                                   s.to_bytes(32, "little"), _p(out), _p(t))
    if rc:
        raise RuntimeError("oracle_create_proof rc=%d" % rc)
    if timings is not None:
        timings.update(dict(zip(["eval", "ntt", "msm_g1", "msm_g2", "assembly"], t.tolist())))
    return out.tobytes()


def closed_form_proof(cs, toxic, inputs, aux, r, s):
    t = b"".join(x.to_bytes(32, "little") for x in toxic)
    out = np.zeros(192, dtype=np.uint8)
    rc = lib().oracle_closed_form_proof(cs.ref, t, _p(inputs), _p(aux), r.to_bytes(32, "little"),
                                        s.to_bytes(32, "little"), _p(out))
    if rc:
        raise RuntimeError("oracle_closed_form_proof rc=%d" % rc)
    return out.tobytes()


def verify_proof(params_buf, proof, public_inputs):
    """public_inputs: list of ints (excluding ONE)"""
    pb = np.ascontiguousarray(params_buf, dtype=np.uint8)
    pi = b"".join(x.to_bytes(32, "little") for x in public_inputs)
    return lib().oracle_verify_proof(_p(pb), pb.size, bytes(proof), pi, len(public_inputs))

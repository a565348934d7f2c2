"""The Miller loop of the GPU batch verifier is a set of levelled straight-line programs over Fp built on the host
(masp_amd/csrc/host/pairing_prog.h) and interpreted by one wavefront per pairing.  The programs run on a host interpreter
too: here they are compared with the library's own Miller loop (host/pairing.h, itself cross-checked against the oracle's
independent pairing in tests/test_circuits.py) — no GPU needed."""
import ctypes as C

import pytest

import oracle_lib as O
from pyref import R


@pytest.mark.parametrize("k", [1, 2, 123456789, (1 << 200) + 7, 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000000])
def test_program_miller_loop_equals_the_host_miller_loop(k):
    from masp_amd import host as H
    L = H.load_library()
    L.masp_host_pairing_program_selftest.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint32)]
    st = (C.c_uint32 * 15)()
    p, _ = O.g1_mul_gen(k)
    q, _ = O.g2_mul_gen((3 * k + 1) % R)
    assert L.masp_host_pairing_program_selftest(p, q, st) == 0
    dbl, add, mul12 = list(st[0:5]), list(st[5:10]), list(st[10:15])
    # [ops, steps, steps containing products, products, slots]: a doubling iteration is 3 product steps, not ~135 products in a row
    assert dbl[2] == 3 and dbl[3] == 104 and add[2] == 4 and mul12[2] == 1 and mul12[3] == 54
    assert max(dbl[4], add[4], mul12[4]) * 48 <= 64 * 1024          # the slots of a pair fit the LDS of its wavefront

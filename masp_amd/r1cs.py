"""Static R1CS container (three CSR matrices) in the layout of `masp_hip_r1cs` (include/masp_hip.h)."""
import ctypes as C

import numpy as np


class R1csStruct(C.Structure):
    _fields_ = [("n_inputs", C.c_uint32), ("n_aux", C.c_uint32), ("n_constraints", C.c_uint32),
                ("a_rowptr", C.c_void_p), ("a_col", C.c_void_p), ("a_coef", C.c_void_p),
                ("b_rowptr", C.c_void_p), ("b_col", C.c_void_p), ("b_coef", C.c_void_p),
                ("c_rowptr", C.c_void_p), ("c_col", C.c_void_p), ("c_coef", C.c_void_p)]


class R1cs:
    """mats: [(rowptr u32[n_constraints+1], col u32[nnz], coef u8[nnz,32] little-endian canonical)] x 3 (A, B, C).

    Column v < n_inputs is Input(v) (Input(0) = ONE), otherwise Aux(v - n_inputs); terms are merged per
    variable and non-zero, so the pattern is exactly bellperson's density information."""

    def __init__(self, n_inputs, n_aux, n_constraints, mats):
        self.n_inputs, self.n_aux, self.n_constraints = int(n_inputs), int(n_aux), int(n_constraints)
        self.mats = [(np.ascontiguousarray(rp, dtype=np.uint32), np.ascontiguousarray(col, dtype=np.uint32),
                      np.ascontiguousarray(coef, dtype=np.uint8).reshape(-1, 32)) for rp, col, coef in mats]
        for rp, col, coef in self.mats:
            assert rp.shape == (self.n_constraints + 1,) and col.shape[0] == coef.shape[0] == rp[-1]
        s = R1csStruct()
        s.n_inputs, s.n_aux, s.n_constraints = self.n_inputs, self.n_aux, self.n_constraints
        for name, (rp, col, coef) in zip("abc", self.mats):
            setattr(s, name + "_rowptr", rp.ctypes.data)
            setattr(s, name + "_col", col.ctypes.data)
            setattr(s, name + "_coef", coef.ctypes.data)
        self.struct = s

    @property
    def nrows(self):
        return self.n_constraints + self.n_inputs

    @property
    def logm(self):
        return max(1, (self.nrows - 1).bit_length())

    @property
    def ref(self):
        return C.byref(self.struct)

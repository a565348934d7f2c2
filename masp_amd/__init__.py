"""masp_amd — MI355X-native Groth16 prover for the MASP Spend / Output / Convert circuits.

The compute path is the HIP library ``libmasp_hip.so`` (masp_amd/csrc, C ABI in include/masp_hip.h).
This package is the thin host side above that ABI; it contains no arithmetic fallback: if the
extension or a GPU is missing, calls raise.
"""
from .hip import Context, MaspHipError, library_path, load_library  # noqa: F401
from .r1cs import R1cs  # noqa: F401

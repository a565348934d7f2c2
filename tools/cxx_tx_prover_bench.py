"""Spend proofs per second through the C++ mirror (include/masp_tx_prover.hpp, masp::LocalTxProver::spend_proofs) — host to host:
descriptions -> witness synthesis on the host threads -> batches of 256 on the GPU -> GPU batch self-verification -> (zkproof, cv, rk),
the figure bench.py reports as `end_to_end` for the Python mirror.  Writes a case of N distinct Spend descriptions (synthetic CRS
generated on the GPU), runs tests/native/tx_prover_harness.cpp in its timed mode and prints its log.
  python tools/cxx_tx_prover_bench.py [N=2048]"""
import os
import random
import struct
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import masp_amd  # noqa: E402
import test_tx_prover_cpp as T  # noqa: E402
from masp_amd import host as H  # noqa: E402
from masp_amd import synthetic  # noqa: E402
from masp_amd import workload as W  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
with ThreadPoolExecutor(H.effective_cpus()) as ex:
    descs = list(ex.map(lambda k: W.description("spend", 5000 + k), range(N)))
c = masp_amd.Context(0)
params = [c.generate_parameters(H.circuit(k)[0], synthetic.toxic_waste(31 + i)) for i, k in enumerate(("spend", "output", "convert"))]
c.close()
rng = random.Random(3)
blob = bytearray(b"MTP1")
for p in params:
    blob += struct.pack("<Q", p.size) + p.tobytes()
blob += struct.pack("<IIIII", int(os.environ.get("MASP_TXP_SELF_VERIFY", "1")), int(os.environ.get("MASP_TXP_THREADS", "0")), 0, 2, len(descs))
for kind, kw in descs:
    blob += T._record(kind, kw, rng.randrange(H.FR_MODULUS), rng.randrange(H.FR_MODULUS))
with tempfile.TemporaryDirectory() as d:
    case, out = os.path.join(d, "case.bin"), os.path.join(d, "out.bin")
    open(case, "wb").write(bytes(blob))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    run = subprocess.run([T._build(), case, out], capture_output=True, text=True, timeout=1800, env=env)
    print(run.stdout + run.stderr)
    sys.exit(run.returncode)

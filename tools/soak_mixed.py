"""BASELINE.json configs[4] on however many GPUs this process sees (one here): N mixed Spend / Output / Convert
descriptions (job i has circuit i mod 3) through `LocalTxProver.prove_batch` — synthesis, GPU batches grouped by circuit,
batch self-verification of Spend / Convert — then every Output proof is verified too and the context state is compared
with a serial accumulation.  A stability run as much as a measurement.

    python tools/soak_mixed.py [N=4096]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench import options_from_env                     # noqa: E402

from masp_amd import workload as W                     # noqa: E402
from masp_amd import host as H                         # noqa: E402
from masp_amd.prover import LocalTxProver, _int         # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    prover = LocalTxProver.with_synthetic_parameters(seed=11, options=options_from_env())
    out_vk = H.PreparedVerifyingKey(prover.parameters["output"])
    from concurrent.futures import ThreadPoolExecutor
    t = time.time()
    with ThreadPoolExecutor(H.effective_cpus()) as ex:       # instances shaped like the reference's benches (masp_amd/workload.py)
        descs = list(ex.map(lambda i: W.description(("spend", "output", "convert")[i % 3], i), range(n)))
    print("%d descriptions built in %.1f s" % (n, time.time() - t))
    t = time.time()
    prover.prove_batch(prover.new_sapling_proving_context(), descs)      # first call: page-locked buffers, every slot's scratch at its final size
    print("first call (buffers and scratch allocated): %.2f s = %.1f proofs/s" % (time.time() - t, n / (time.time() - t)))
    ctx = prover.new_sapling_proving_context()
    seen = []
    t0 = time.time()
    out = prover.prove_batch(ctx, descs, progress=lambda done, total: seen.append(done))
    dt = time.time() - t0
    assert len(out) == n and seen[-1] == n and len({o[0] for o in out}) == n
    # Output proofs are not self-checked by the prover (like the reference): check them here
    t1 = time.time()
    proofs, pis = [], []
    for (kind, kw), o in zip(descs, out):
        if kind == "output":
            inputs, _, cv = H.output_assignment(kw["esk"], kw["payment_address"][0], kw["payment_address"][1], kw["rcm"], kw["asset_type"],
                                                kw["value"], kw["rcv"])
            assert cv == o[1]
            proofs.append(o[0])
            pis.append([int.from_bytes(inputs[i].tobytes(), "little") for i in range(1, 6)])
    assert out_vk.verify_batch(proofs, pis)
    # context state = serial accumulation
    bsk, cv_sum = 0, H.JUBJUB_IDENTITY
    for (kind, kw), o in zip(descs, out):
        sign = -1 if kind == "output" else 1
        bsk = (bsk + sign * _int(kw["rcv"])) % H.JUBJUB_ORDER
        cv_sum = H.jubjub_add(cv_sum, o[1], subtract=(kind == "output"))
    assert (ctx.bsk, ctx.cv_sum) == (bsk, cv_sum)
    print("prove_batch: %d mixed proofs in %.2f s = %.1f proofs/s (self-verified); %d Output proofs batch-verified in %.2f s; context state ok"
          % (n, dt, n / dt, len(proofs), time.time() - t1))


if __name__ == "__main__":
    main()

"""Three host threads prove single descriptions on one context with lone_proof_graph=1 (tests/test_gpu_lone_graph.py's last test), many
times over: how often does a call fail, and with what?  MASP_HIP_LIBRARY selects the build."""
import os
import sys
import threading

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import masp_amd  # noqa: E402
import oracle_lib as O  # noqa: E402
from test_gpu_lone_graph import _make  # noqa: E402

ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cs, inputs, aux, pbuf, P = _make()
want = {k: O.create_proof(P, cs, inputs, aux, 100 + k, 200 + k) for k in range(24)}
errors, bad, launches = [], [], 0
for rnd in range(ROUNDS):
    ctx = masp_amd.Context(0, slots=3, lone_proof_graph=int(os.environ.get("STRESS_GRAPH", "1")))
    try:
        ctx.load_circuit(2, pbuf, cs)

        def work(t):
            for k in range(t, 24, 3):
                try:
                    got = ctx.prove_batch([(2, inputs, aux, 100 + k, 200 + k)])
                    if got != [want[k]]:
                        bad.append((rnd, k))
                except Exception as e:  # noqa: BLE001
                    errors.append((rnd, k, str(e)))
        th = [threading.Thread(target=work, args=(t,)) for t in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        launches += ctx.lone_graph_launches()
    finally:
        ctx.close()
print("library %s: %d rounds x 24 proofs, %d graph launches, %d wrong proofs, %d failed calls" % (os.environ.get("MASP_HIP_LIBRARY", "default"), ROUNDS, launches, len(bad), len(errors)))
for e in errors[:8]:
    print("   ", e)

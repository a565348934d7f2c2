"""Host-side mirror of the pieces of masp_primitives the proving path touches: AllowedConversion
(/root/reference/masp_primitives/src/convert.rs:22-118), the canonical-scalar contract of jubjub::Fr, and the bench
instances (masp_amd/workload.py, shaped like /root/reference/masp_proofs/benches/{sapling,convert}.rs)."""
import pytest

import oracle_lib as O


def test_allowed_conversion_is_the_signed_sum_of_asset_generators():
    from masp_amd import host as H
    a, b, c = (H.asset_identifier(n) for n in (b"asset 7", b"asset 8", b"reward"))
    ac = H.AllowedConversion([(a, -8), (b, 8), (c, 8)])                      # the reference bench's shape (benches/convert.rs:32-44)
    g = H.JUBJUB_IDENTITY
    for ident, v in ((a, -8), (b, 8), (c, 8)):
        g = H.jubjub_add(g, H.jubjub_mul(H.asset_generator(ident), abs(v)), subtract=v < 0)
    assert ac.generator == g and ac.cmu() == H.convert_cmu(g)
    # ValueSum addition merges equal asset types and drops zeros; order does not matter
    assert H.AllowedConversion([(c, 8), (a, -3), (b, 8), (a, -5), (b"\x01" * 32, 0)]).generator == g
    assert H.AllowedConversion({}).generator == H.JUBJUB_IDENTITY
    # `abs as u64` keeps the low 64 bits of |value| (convert.rs:96-99), and i128::MIN has no absolute value ("invalid conversion")
    big = (1 << 64) + 5
    assert H.AllowedConversion([(a, big)]).generator == H.AllowedConversion([(a, 5)]).generator
    assert H.AllowedConversion([(a, -big)]).generator == H.AllowedConversion([(a, -5)]).generator
    with pytest.raises(ValueError):
        H.AllowedConversion([(a, -(1 << 127))])
    # value commitment = [value]([8] generator) + [rcv] G_vcr, as the Convert circuit's native side computes it
    from masp_amd import workload as W
    kind, kw = W.description("convert", 3)
    _, _, cv = H.convert_assignment(kw["allowed_conversion"].generator, kw["value"], kw["anchor"], kw["merkle_path"][0], kw["merkle_path"][1], kw["rcv"])
    assert kw["allowed_conversion"].value_commitment(kw["value"], kw["rcv"]) == cv


def test_unreduced_jubjub_scalars_are_rejected():
    """jubjub::Fr is canonical by construction in the reference; raw bytes >= the subgroup order would make the circuit's
    252-bit witness disagree with the natively computed cv / rk / cm — a silently invalid proof.  They are an error here."""
    from masp_amd import host as H
    from masp_amd import workload as W
    kind, kw = W.description("spend", 1)
    ak, nsk = kw["proof_generation_key"]
    sib, pos = kw["merkle_path"]
    args = [ak, nsk, kw["diversifier"], kw["rcm"], kw["ar"], kw["asset_type"], kw["value"], kw["anchor"], sib, pos, kw["rcv"]]
    H.spend_assignment(*args)
    for idx in (1, 3, 4, 10):                                # nsk, rcm, ar, rcv
        bad = list(args)
        bad[idx] = H.JUBJUB_ORDER + 5 if idx != 10 else H.JUBJUB_ORDER
        with pytest.raises(H.HostError) as e:
            H.spend_assignment(*bad)
        assert e.value.code == 1
    kind, kw = W.description("output", 1)
    d, pk = kw["payment_address"]
    with pytest.raises(H.HostError):
        H.output_assignment((1 << 256) - 1, d, pk, kw["rcm"], kw["asset_type"], kw["value"], kw["rcv"])
    kind, kw = W.description("convert", 1)
    with pytest.raises(H.HostError):
        H.convert_assignment(kw["allowed_conversion"].generator, 1, kw["anchor"], kw["merkle_path"][0], kw["merkle_path"][1], H.JUBJUB_ORDER)


@pytest.mark.parametrize("kind", ["spend", "output", "convert"])
def test_bench_instances_are_distinct_and_satisfy_their_circuits(kind):
    from masp_amd import host as H
    from masp_amd import workload as W
    insts = W.instances(kind, 6, first_seed=50, threads=2)
    cs = H.circuit(kind)[0]
    assert len({a.tobytes() for _, a in insts}) == 6 and len({i.tobytes() for i, _ in insts}) == 6
    for inputs, aux in insts:
        assert O.r1cs_unsatisfied(cs, inputs, aux) == 0      # independent evaluation by the oracle
        assert len(W.public_inputs(inputs)) == cs.n_inputs - 1
    # deterministic in (kind, seed)
    again = W.instances(kind, 2, first_seed=50, threads=1)
    assert all((x[0] == y[0]).all() and (x[1] == y[1]).all() for x, y in zip(insts, again))


def test_aux_assignment_as_montgomery_residues():
    """masp_hip_job::aux_form = 1: libmasp_host can hand over the aux assignment as Montgomery residues (four little-endian u64
    limbs of a * 2^256 mod r: the in-memory form of blst_fr) instead of canonical bytes; same witness, other encoding."""
    import numpy as np
    from masp_amd import host as H
    from masp_amd import workload as W
    R = H.FR_MODULUS
    for kind in ("spend", "output", "convert"):
        _, kw = W.description(kind, 3)
        inputs, aux = W.assignment(kind, kw)
        if kind == "spend":
            ak, nsk = kw["proof_generation_key"]
            sib, pos = kw["merkle_path"]
            in2, aux2, *_ = H.spend_assignment(ak, nsk, kw["diversifier"], kw["rcm"], kw["ar"], kw["asset_type"], kw["value"], kw["anchor"], sib, pos,
                                               kw["rcv"], montgomery=True)
        elif kind == "output":
            d, pk = kw["payment_address"]
            in2, aux2, _ = H.output_assignment(kw["esk"], d, pk, kw["rcm"], kw["asset_type"], kw["value"], kw["rcv"], montgomery=True)
        else:
            sib, pos = kw["merkle_path"]
            in2, aux2, _ = H.convert_assignment(kw["allowed_conversion"].generator, kw["value"], kw["anchor"], sib, pos, kw["rcv"], montgomery=True)
        assert (in2 == inputs).all() and aux2.shape == aux.shape
        rng = np.random.default_rng(1)
        for j in list(rng.integers(0, aux.shape[0], 300)) + [0, aux.shape[0] - 1]:
            a = int.from_bytes(aux[j].tobytes(), "little")
            assert int.from_bytes(aux2[j].tobytes(), "little") == a * (1 << 256) % R


def test_permits_are_served_in_ticket_order():
    """prove_batch's synthesis window (masp_amd/prover.py `_Permits`): a large request at the head of the queue must not be overtaken by
    later small ones (with all permits then held by later chunks' groups no chunk could finish and release)."""
    import threading
    import time
    from masp_amd.prover import _Permits
    p = _Permits(4)
    order, lock = [], threading.Lock()

    def want(k, name):
        p.acquire(k)
        with lock:
            order.append(name)
    big = threading.Thread(target=want, args=(16, "big"))
    big.start()
    time.sleep(0.2)                                   # `big` holds ticket 0 and waits for 16 permits
    small = [threading.Thread(target=want, args=(1, "small%d" % i)) for i in range(8)]
    for t in small:
        t.start()
    time.sleep(0.3)
    assert order == []                                # 4 permits are free, but the head of the queue wants 16
    p.release(12)
    big.join(timeout=10)
    assert order == ["big"]
    p.release(8)
    for t in small:
        t.join(timeout=10)
    assert order[0] == "big" and sorted(order[1:]) == sorted("small%d" % i for i in range(8))


def test_an_interrupted_waiter_gives_its_ticket_up():
    """a waiter that leaves `_Permits.acquire` by an exception (an interrupt inside wait()) must not leave a ticket nobody
    serves — every later acquirer would wait for ever."""
    import ctypes
    import threading
    import time
    from masp_amd.prover import _Permits
    p = _Permits(1)
    got, errors = [], []

    def head():
        try:
            p.acquire(5)                              # ticket 0: can never be met before the release below
            got.append("head")
        except KeyboardInterrupt:
            errors.append("interrupted")
    t0 = threading.Thread(target=head)
    t0.start()
    time.sleep(0.2)
    t1 = threading.Thread(target=lambda: (p.acquire(1), got.append("later")))   # ticket 1 queues behind ticket 0
    t1.start()
    time.sleep(0.2)
    assert got == []
    # raise KeyboardInterrupt inside the head waiter (what Ctrl-C does to the main thread), then wake it
    assert ctypes.pythonapi.PyThreadState_SetAsyncExc(ctypes.c_ulong(t0.ident), ctypes.py_object(KeyboardInterrupt)) == 1
    with p._cv:
        p._cv.notify_all()
    t0.join(timeout=10)
    t1.join(timeout=10)
    assert errors == ["interrupted"] and got == ["later"] and not t1.is_alive()


def test_a_failed_background_warm_up_is_reported_by_the_next_proving_call():
    """warm_up(background=True) runs on a daemon thread; its exception is kept and raised by `_warm_wait`, which every
    proving call passes through first."""
    from masp_amd.prover import LocalTxProver
    p = LocalTxProver.__new__(LocalTxProver)            # no GPU: only the two methods under test
    real = LocalTxProver.warm_up

    def warm_up(self, spends=None, outputs=0, converts=0, threads=None, background=False):
        if background:
            return real(self, spends, outputs, converts, threads, True)
        raise MemoryError("no scratch")
    p.warm_up = warm_up.__get__(p)
    p.warm_up(background=True)
    with pytest.raises(RuntimeError, match="warm_up.*failed.*no scratch"):
        p._warm_wait()
    p._warm_wait()                                      # reported once; the prover is the caller's to discard

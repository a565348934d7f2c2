"""libmasp_host's several-witnesses-per-call entry points (masp_host_spend_assignments / masp_host_convert_assignments): the
Merkle blocks of the witnesses run in lockstep — affine Montgomery additions with ONE shared field inversion per window index
across all witnesses (csrc/host/circuits.h merkle_block_batch) — and write the variables straight into the assignment in the
order the gadgets of /root/reference/masp_proofs/src/circuit/{sapling,pedersen_hash,ecc}.rs allocate them.  The generic
gadget path (which the structure-hash KATs pin) is the reference here: same inputs, same aux bytes, canonical and Montgomery."""
import numpy as np
import pytest

from masp_amd import host as H
from masp_amd import workload as W


def _kws(kind, n, first=700):
    return [W.description(kind, first + k)[1] for k in range(n)]


@pytest.mark.parametrize("kind", ["spend", "convert"])
def test_lockstep_witnesses_equal_the_gadget_path(kind):
    kws = _kws(kind, 7)                                   # an odd group size; positions, siblings and keys all differ
    one = [W.assignment(kind, kw) for kw in kws]
    many = W.assignments(kind, kws)
    assert len(many) == len(one)
    for (i1, a1), (i2, a2) in zip(one, many):
        assert i1.shape == i2.shape and a1.shape == a2.shape
        assert (i1 == i2).all() and (a1 == a2).all()
    # a group of one goes the same way
    solo = W.assignments(kind, kws[:1])
    assert (solo[0][1] == one[0][1]).all()
    # Montgomery form: written in place into the caller's buffer; back in canonical form it is the same assignment
    bufs = [np.full(a.shape, 0xA5, np.uint8) for _, a in one]
    mont = W.assignments(kind, kws, aux_outs=bufs, montgomery=True)
    for (i1, a1), (i2, a2), buf in zip(one, mont, bufs):
        assert a2 is buf and (i1 == i2).all()
        assert (H.aux_from_montgomery(a2) == a1).all()


def test_lockstep_witnesses_satisfy_the_recorded_constraints():
    """check=True records the constraints next to the witness (one by one through the gadgets) and evaluates them: the batch entry
    point gives the same verdicts, and a wrong anchor is found unsatisfied either way."""
    kws = _kws("spend", 3, first=900)
    items = [W._spend_item(kw) for kw in kws]
    ok = H.spend_assignments(items, check=True)
    assert all(not isinstance(r, Exception) for r in ok)
    bad = list(items[1])
    bad[7] = (int.from_bytes(H._b(bad[7]), "little") ^ 2).to_bytes(32, "little")      # another anchor
    res = H.spend_assignments([items[0], tuple(bad), items[2]], check=True)
    assert not isinstance(res[0], Exception) and not isinstance(res[2], Exception)
    assert isinstance(res[1], H.HostError) and res[1].code == 4                      # MASP_HOST_E_UNSATISFIED


def test_one_bad_description_does_not_take_its_group_down():
    kws = _kws("spend", 4, first=950)
    items = [list(W._spend_item(kw)) for kw in kws]
    items[2][2] = bytes(11)                     # a diversifier without a group hash (the all-zero one has none) ...
    try:
        H.spend_leaf(items[2][0], items[2][1], items[2][2], items[2][3], items[2][5], 1)
        pytest.skip("the all-zero diversifier happens to be valid")
    except H.HostError:
        pass
    res = H.spend_assignments([tuple(i) for i in items])
    assert isinstance(res[2], H.HostError) and res[2].code == 2                      # MASP_HOST_E_DIVERSIFIER
    good = [W.assignment("spend", kws[j]) for j in (0, 1, 3)]
    for j, (i1, a1) in zip((0, 1, 3), good):
        assert (res[j][0] == i1).all() and (res[j][1] == a1).all()

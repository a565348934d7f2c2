"""`SaplingProvingContext::binding_sig` (masp_proofs/src/sapling/prover.rs:279-326) and the RedJubjub it signs with
(masp_primitives/src/sapling/redjubjub.rs:138-160,191-239): host-only, no GPU."""
import random

import pytest

from masp_amd import host as H
from masp_amd import redjubjub as RJS
from masp_amd.prover import ProvingError, SaplingProvingContext

G_RCV = H.point_bytes(*H.generator_uv(3))


def test_redjubjub_sign_verify_roundtrip():
    rng = random.Random(1)
    for _ in range(4):
        sk = rng.randrange(1, H.JUBJUB_ORDER)
        vk = RJS.public_key(sk, G_RCV)
        msg = bytes(rng.getrandbits(8) for _ in range(64))
        sig = RJS.sign(sk, msg, G_RCV)
        assert len(sig) == 64 and RJS.verify(vk, msg, sig, G_RCV)
        assert not RJS.verify(vk, msg[:-1] + bytes([msg[-1] ^ 1]), sig, G_RCV)
        assert not RJS.verify(vk, msg, sig[:40] + bytes([sig[40] ^ 1]) + sig[41:], G_RCV)
        assert not RJS.verify(RJS.public_key(sk + 1, G_RCV), msg, sig, G_RCV)
        assert not RJS.verify(vk, msg, sig[:32] + (H.JUBJUB_ORDER).to_bytes(32, "little"), G_RCV)   # non-canonical S
    # deterministic nonce source -> deterministic signature (T is the only randomness)
    fixed = lambda n: bytes(range(n))          # noqa: E731
    assert RJS.sign(5, b"m" * 64, G_RCV, rng=fixed) == RJS.sign(5, b"m" * 64, G_RCV, rng=fixed)


def test_binding_sig_matches_what_the_verifier_reconstructs():
    """spend(v1, asset A) + spend(v3, asset B) - output(v2, asset A): the signature verifies under
    bvk = cv_sum - sum_assets [balance] vcg(asset), the key `final_check` derives (sapling/verifier.rs)."""
    rng = random.Random(2)
    a, b = H.asset_identifier(b"asset A"), H.asset_identifier(b"asset B")
    v1, v2, v3 = 1000, 250, 7
    rcv1, rcv2, rcv3 = (rng.randrange(1, H.JUBJUB_ORDER) for _ in range(3))
    ctx = SaplingProvingContext()
    ctx._spend_like(rcv1, H.value_commitment(a, v1, rcv1)[0])
    ctx._output(rcv2, H.value_commitment(a, v2, rcv2)[0])
    ctx._spend_like(rcv3, H.value_commitment(b, v3, rcv3)[0])
    sighash = bytes(rng.getrandbits(8) for _ in range(32))
    sig = ctx.binding_sig([(a, v1 - v2), (b, v3)], sighash)
    bvk = ctx.cv_sum
    for asset, bal in ((a, v1 - v2), (b, v3)):
        bvk = H.jubjub_add(bvk, H.jubjub_mul(H.jubjub_mul(H.asset_generator(asset), 8), bal), subtract=True)
    assert bvk == RJS.public_key(ctx.bsk, G_RCV)
    assert RJS.verify(bvk, bvk + sighash, sig, G_RCV)
    # negative balance (more outputs than spends of an asset)
    ctx2 = SaplingProvingContext()
    ctx2._output(rcv2, H.value_commitment(a, v2, rcv2)[0])
    sig2 = ctx2.binding_sig([(a, -v2)], sighash)
    assert RJS.verify(RJS.public_key(ctx2.bsk, G_RCV), RJS.public_key(ctx2.bsk, G_RCV) + sighash, sig2, G_RCV)
    # wrong balances -> Err(())
    with pytest.raises(ProvingError):
        ctx.binding_sig([(a, v1 - v2 + 1), (b, v3)], sighash)
    with pytest.raises(ProvingError):
        ctx.binding_sig([(a, v1 - v2)], sighash)
    with pytest.raises(ProvingError):
        ctx.binding_sig([(a, -(1 << 127))], sighash)          # i128::MIN: checked_abs fails

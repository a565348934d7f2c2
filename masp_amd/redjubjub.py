"""RedJubjub (RedDSA on Jubjub) for the binding signature of `SaplingProvingContext::binding_sig`
(/root/reference/masp_proofs/src/sapling/prover.rs:279-326), restated from
/root/reference/masp_primitives/src/sapling/redjubjub.rs:36-38,138-160,166-169,191-239 and
masp_primitives/src/sapling/util.rs:9-15.  Host-side and tiny: it closes the `TxProver` surface around the GPU prover;
group arithmetic goes through libmasp_host's Jubjub (`host.jubjub_mul / jubjub_add`)."""
import hashlib
import secrets

from . import host as H

RJ = H.JUBJUB_ORDER


def h_star(a, b):
    """H*(a || b) = BLAKE2b-512(personal "MASP__RedJubjubH") reduced into the Jubjub scalar field (`from_bytes_wide`)."""
    h = hashlib.blake2b(digest_size=64, person=b"MASP__RedJubjubH")
    h.update(a)
    h.update(b)
    return int.from_bytes(h.digest(), "little") % RJ


def public_key(sk, generator):
    """PublicKey::from_private: [sk] P_G."""
    return H.jubjub_mul(generator, sk % RJ)


def sign(sk, msg, generator, rng=secrets.token_bytes):
    """PrivateKey::sign -> 64 bytes Rbar || Sbar.  T = 80 random bytes, r = H*(T || M), R = [r] P_G, S = r + H*(Rbar || M) sk."""
    t = rng(80)
    r = h_star(t, msg)
    rbar = H.jubjub_mul(generator, r)
    s = (r + h_star(rbar, msg) * (sk % RJ)) % RJ
    return rbar + s.to_bytes(32, "little")


def verify(vk, msg, sig, generator):
    """PublicKey::verify (ZIP 216 rules: canonical R): [8]( -[S] P_G + R + [c] vk ) == identity."""
    if len(sig) != 64:
        return False
    rbar, sbar = sig[:32], sig[32:]
    s = int.from_bytes(sbar, "little")
    if s >= RJ:
        return False
    try:
        c = h_star(rbar, msg)
        acc = H.jubjub_add(H.jubjub_add(rbar, H.jubjub_mul(vk, c)), H.jubjub_mul(generator, s), subtract=True)
        return H.jubjub_mul(acc, 8) == H.JUBJUB_IDENTITY
    except (H.HostError, ValueError):
        return False            # R or vk is not a canonical point encoding

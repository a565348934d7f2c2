"""Synthetic R1CS instances with the exact prover-side shape of the MASP circuits.

`shaped("spend")` builds a satisfiable constraint system whose sizes equal those of the reference's
Spend circuit as derived in SURVEY.md App. C.3 (pinned by /root/reference/masp_proofs/src/circuit/sapling.rs:730-741
and the parameter-file size equations): number of inputs / aux / constraints, the A- and B-query density counts
(hence every MSM length and the NTT domain) and the share of boolean witness values inside each query.  It is NOT
the Spend circuit — the constraints are boolean checks, bit-packings and products laid out to hit those counts —
and exists so that kernels can be measured at the real problem sizes with the real value distribution
("a uniformly random witness would overstate MSM work by ~2x").
"""
import random

import numpy as np

from .r1cs import R1cs

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001

# SURVEY.md App. C.3
SHAPES = {
    #            n_in  n_aux  n_constraints a_aux_dens b_aux_dens bool_aux bool_in_A bool_in_B
    "spend":   (8, 100497, 100637, 86923, 62169, 70051, 69733, 51801),
    "output":  (6, 30896, 31205, 28753, 21383, 24285, 23995, 17704),
    "convert": (4, 47322, 47358, 35743, 24088, 23051, 23051, 17264),
}


def _le(x):
    return np.frombuffer((x % R).to_bytes(32, "little"), dtype=np.uint8)


def shaped(kind, seed=0):
    """-> (R1cs, inputs u8[n_in,32], aux u8[n_aux,32]) ; satisfiable, deterministic in (kind, seed)."""
    n_in, n_aux, n_con, a_dens, b_dens, n_bool, bool_a, bool_b = SHAPES[kind]
    rng = random.Random(("masp-shaped", kind, seed).__repr__())
    n_full = n_aux - n_bool
    full_a, full_b = a_dens - bool_a, b_dens - bool_b       # full-width variables in the A / B queries
    assert bool_b <= bool_a <= n_bool and full_b <= full_a <= n_full
    # variable numbering (column index = n_in + aux index): booleans first, then full-width
    B0 = n_in
    F0 = n_in + n_bool
    val = [0] * (n_in + n_aux)
    val[0] = 1
    for j in range(n_bool):
        val[B0 + j] = rng.getrandbits(1)
    n_pack_bools = bool_a - bool_b
    n_pack = (n_pack_bools + 7) // 8
    n_prod = n_full - n_pack - full_a if n_full - n_pack - full_a > 0 else 0
    # full-width vars: [0, full_a) A-pool (free), then n_pack packed words, then products, rest free
    for k in range(full_a):
        val[F0 + k] = rng.randrange(R)
    rows_a, rows_b, rows_c = [], [], []
    ONE, NEG = 1, R - 1
    # (1) boolean constraints (1 - b) * b = 0 : b in A and B
    for j in range(bool_b):
        v = B0 + j
        rows_a.append(((0, ONE), (v, NEG)))
        rows_b.append(((v, ONE),))
        rows_c.append(())
    # (2) packings  (sum 2^i b_i) * 1 = z : b in A only
    for k in range(n_pack):
        lo = bool_b + 8 * k
        hi = min(lo + 8, bool_a)
        z = F0 + full_a + k
        val[z] = sum(val[B0 + j] << (j - lo) for j in range(lo, hi)) % R
        rows_a.append(tuple((B0 + j, 1 << (j - lo)) for j in range(lo, hi)))
        rows_b.append(((0, ONE),))
        rows_c.append(((z, ONE),))
    # (3) products z = x * y with x from the A-pool, y from the B-pool (a prefix of the A-pool)
    base = F0 + full_a + n_pack
    n_prod = min(n_full - full_a - n_pack, n_con - len(rows_a) - (n_in - 1))
    for k in range(n_prod):
        x = F0 + (k % full_a)
        y = F0 + (k * 7 % full_b)
        z = base + k
        val[z] = val[x] * val[y] % R
        rows_a.append(((x, ONE),))
        rows_b.append(((y, ONE),))
        rows_c.append(((z, ONE),))
    for k in range(base + n_prod, n_in + n_aux):       # leftover full-width variables stay free
        val[k] = rng.randrange(R)
    # (4) public inputs: x * 1 = input_i
    for i in range(1, n_in):
        x = F0 + (i % full_a)
        val[i] = val[x]
        rows_a.append(((x, ONE),))
        rows_b.append(((0, ONE),))
        rows_c.append(((i, ONE),))
    # (4b) make sure every B-pool variable really occurs in B:  1 * y = y
    seen_b = set(lc[0][0] for lc in rows_b if lc)
    for k in range(full_b):
        y = F0 + k
        if y not in seen_b and len(rows_a) < n_con:
            rows_a.append(((0, ONE),))
            rows_b.append(((y, ONE),))
            rows_c.append(((y, ONE),))
    # (4c) every aux variable must occur somewhere, otherwise its L-query point is the identity, which the
    # bellman reader rejects.  Uncovered variables are paired with equal values and hung on the C side:
    # x * 1 = x + u - u'
    covered = set()
    for rows in (rows_a, rows_b, rows_c):
        for lc in rows:
            for v, _ in lc:
                covered.add(v)
    loose = [v for v in range(n_in, n_in + n_aux) if v not in covered]
    for k in range(0, len(loose), 2):
        x = F0 + (k % full_a)
        if k + 1 < len(loose):
            u, u2 = loose[k], loose[k + 1]
            val[u2] = val[u]
            lc = [(x, ONE), (u, ONE), (u2, NEG)]
        else:
            u = loose[k]
            val[u] = 0
            lc = [(x, ONE), (u, ONE)]
        rows_a.append(((x, ONE),))
        rows_b.append(((0, ONE),))
        rows_c.append(tuple(sorted(lc)))
    assert len(rows_a) <= n_con
    # (5) identities x * 1 = x to reach the constraint count without touching densities
    k = 0
    while len(rows_a) < n_con:
        x = F0 + (k % full_a)
        rows_a.append(((x, ONE),))
        rows_b.append(((0, ONE),))
        rows_c.append(((x, ONE),))
        k += 1
    assert len(rows_a) == n_con
    coef_cache = {}

    def csr(rows):
        rp = np.zeros(len(rows) + 1, dtype=np.uint32)
        cols, codes = [], []
        for r, lc in enumerate(rows):
            for v, c in lc:
                cols.append(v)
                if c not in coef_cache:
                    coef_cache[c] = len(coef_cache)
                codes.append(coef_cache[c])
            rp[r + 1] = len(cols)
        return rp, np.array(cols, dtype=np.uint32), np.array(codes, dtype=np.int64)

    mats = [csr(rows_a), csr(rows_b), csr(rows_c)]
    table = np.zeros((len(coef_cache), 32), dtype=np.uint8)
    for c, k in coef_cache.items():
        table[k] = _le(c)
    cs = R1cs(n_in, n_aux, n_con, [(rp, col, table[codes]) for rp, col, codes in mats])
    allv = np.frombuffer(b"".join((v % R).to_bytes(32, "little") for v in val), dtype=np.uint8).reshape(-1, 32)
    return cs, allv[:n_in].copy(), allv[n_in:].copy()


def toxic_waste(seed=0):
    rng = random.Random(("masp-toxic", seed).__repr__())
    return [rng.randrange(2, R) for _ in range(5)]

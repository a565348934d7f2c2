"""Seeded toy R1CS instances with satisfying witnesses (shared by oracle and GPU parity tests).

Shape mimics the MASP circuits' value distribution: a configurable share of the aux
variables are booleans, the rest are full-width field elements."""
import random

import numpy as np

from pyref import R
from oracle_lib import R1cs


def _le(x):
    return np.frombuffer((x % R).to_bytes(32, "little"), dtype=np.uint8)


def make(seed, n_inputs=3, n_free=6, n_constraints=20, bool_share=0.5, max_terms=3):
    """returns (R1cs, inputs u8[n_inputs,32], aux u8[n_aux,32], values list[int])"""
    rng = random.Random(seed)
    vals = [1] + [rng.randrange(R) for _ in range(n_inputs - 1)]          # inputs (ONE first)
    n_in = n_inputs
    aux = []
    rows = {"a": [], "b": [], "c": []}

    def var_value(v):
        return vals[v] if v < n_in else aux[v - n_in]

    def rand_lc(nvars):
        k = rng.randint(1, max_terms)
        vs = rng.sample(range(nvars), min(k, nvars))
        out = []
        for v in sorted(vs):
            c = rng.choice([1, 1, R - 1, 2, rng.randrange(1, R)])
            out.append((v, c))
        return out

    def lc_val(lc):
        return sum(c * var_value(v) for v, c in lc) % R

    for _ in range(n_free):
        if rng.random() < bool_share:
            b = rng.randint(0, 1)
            aux.append(b)
            v = n_in + len(aux) - 1
            # (1 - b) * b = 0
            rows["a"].append([(0, 1), (v, R - 1)])
            rows["b"].append([(v, 1)])
            rows["c"].append([])
        else:
            aux.append(rng.randrange(R))
    while len(rows["a"]) < n_constraints:
        nvars = n_in + len(aux)
        la, lb = rand_lc(nvars), rand_lc(nvars)
        z = lc_val(la) * lc_val(lb) % R
        kind = rng.random()
        if kind < 0.7:
            aux.append(z)
            rows["c"].append([(nvars, 1)])
        else:
            # k * z' + w = product  with fresh z'
            k = rng.randrange(1, R)
            w = rng.randrange(nvars)
            zz = (z - var_value(w)) * pow(k, -1, R) % R
            aux.append(zz)
            rows["c"].append(sorted([(w, 1), (nvars, k)]))
        rows["a"].append(la)
        rows["b"].append(lb)
    n_aux = len(aux)
    mats = []
    for name in "abc":
        rp, col, coef = [0], [], []
        for lc in rows[name]:
            for v, c in lc:
                col.append(v)
                coef.append(_le(c))
            rp.append(len(col))
        coef = np.stack(coef) if coef else np.zeros((0, 32), np.uint8)
        mats.append((np.array(rp, np.uint32), np.array(col, np.uint32), coef))
    cs = R1cs(n_in, n_aux, len(rows["a"]), mats)
    inputs = np.stack([_le(v) for v in vals])
    auxb = np.stack([_le(v) for v in aux])
    return cs, inputs, auxb, vals + aux


def toxic(seed):
    rng = random.Random(seed * 7919 + 13)
    return [rng.randrange(2, R) for _ in range(5)]

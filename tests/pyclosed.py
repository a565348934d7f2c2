"""Pure-Python toxic-waste closed form of a Groth16 proof (big integers + tests/pyref.py's affine curve arithmetic):
independent of the C++ oracle and of every HIP kernel.  With the trapdoor (tau, alpha, beta, gamma, delta) known, the
proof of a SATISFIED assignment is three fixed-base multiples of the generators:

    A = [alpha + a(tau) + r delta] G1
    B = [beta  + b(tau) + s delta] G2
    C = [ (sum_aux w_v (beta A_v + alpha B_v + C_v)(tau) + a(tau) b(tau) - c(tau)) / delta + s A' + r B' - r s delta ] G1

where a(X), b(X), c(X) interpolate the evaluation vectors of bellperson's ProvingAssignment over the 2^k domain (constraint
rows, then the extra "Input(i) * 0 = 0" rows that put the inputs into a) — evaluated here row by row through the
Lagrange basis L_k(tau) = (tau^m - 1)/m * w^k / (tau - w^k), not through per-variable QAP polynomials as the oracle does.
"""
import numpy as np

from pyref import F1, F2, G1, G2, R, ec_mul, g1_comp, g2_comp

ROOT_OF_UNITY_2_32 = pow(7, (R - 1) >> 32, R)     # 7 generates Fr^*; 2-adicity 32


def _coefs(coef_u8):
    return [int.from_bytes(coef_u8[t].tobytes(), "little") for t in range(coef_u8.shape[0])]


def closed_form_proof(cs, toxic, inputs, aux, r, s):
    """cs: masp_amd.r1cs.R1cs; inputs/aux: u8[n,32] little-endian; toxic = (tau, alpha, beta, gamma, delta) -> 192 bytes"""
    tau, alpha, beta, _gamma, delta = [t % R for t in toxic]
    w = [int.from_bytes(inputs[i].tobytes(), "little") for i in range(cs.n_inputs)] + \
        [int.from_bytes(aux[j].tobytes(), "little") for j in range(cs.n_aux)]
    nc, n_in = cs.n_constraints, cs.n_inputs
    nrows, logm = cs.nrows, cs.logm
    m = 1 << logm
    omega = pow(ROOT_OF_UNITY_2_32, 1 << (32 - logm), R)
    # Lagrange basis at tau for the rows in use, with one batched inversion
    z_over_m = (pow(tau, m, R) - 1) * pow(m, -1, R) % R
    wk, den = [], []
    cur = 1
    for _ in range(nrows):
        wk.append(cur)
        den.append((tau - cur) % R)
        cur = cur * omega % R
    pref = [1]
    for d in den:
        pref.append(pref[-1] * d % R)
    inv = pow(pref[-1], -1, R)
    lag = [0] * nrows
    for k in range(nrows - 1, -1, -1):
        lag[k] = z_over_m * wk[k] % R * (inv * pref[k] % R) % R
        inv = inv * den[k] % R
    # evaluation vectors row by row, and per-variable (beta A_v + alpha B_v + C_v)(tau) for the aux part of C
    evals = []
    lvar = [0] * len(w)
    for mi, (rp, col, coef) in enumerate(cs.mats):
        cf = _coefs(coef)
        weight = (beta, alpha, 1)[mi]
        ev = [0] * nrows
        rp = rp.tolist()
        col = col.tolist()
        for row in range(nc):
            acc = 0
            lk = lag[row]
            for t in range(rp[row], rp[row + 1]):
                acc += cf[t] * w[col[t]]
                lvar[col[t]] = (lvar[col[t]] + weight * cf[t] % R * lk) % R
            ev[row] = acc % R
        evals.append(ev)
    for i in range(n_in):                       # extra rows: a = input value, b = c = 0
        evals[0][nc + i] = w[i]
        lvar[i] = (lvar[i] + beta * lag[nc + i]) % R
    for row in range(nc):
        assert evals[0][row] * evals[1][row] % R == evals[2][row], "assignment does not satisfy row %d" % row
    a_tau, b_tau, c_tau = (sum(e * l for e, l in zip(ev, lag)) % R for ev in evals)
    l_aux = sum(w[v] * lvar[v] for v in range(n_in, len(w))) % R
    ea = (alpha + a_tau + r * delta) % R
    eb = (beta + b_tau + s * delta) % R
    ec = ((l_aux + a_tau * b_tau - c_tau) * pow(delta, -1, R) + s * ea + r * eb - r * s * delta) % R
    return g1_comp(ec_mul(F1, G1, ea)) + g2_comp(ec_mul(F2, G2, eb)) + g1_comp(ec_mul(F1, G1, ec))

// The MASP Spend / Output / Convert circuits and their Jubjub gadgets, statement for statement in the order of
//   /root/reference/masp_proofs/src/circuit/ecc.rs        (fixed_base_multiplication :27-73, EdwardsPoint :75-476, MontgomeryPoint :478-606)
//   /root/reference/masp_proofs/src/circuit/pedersen_hash.rs :19-103
//   /root/reference/masp_proofs/src/circuit/sapling.rs    (expose_value_commitment :71-137, Spend :139-417, Output :419-596)
//   /root/reference/masp_proofs/src/circuit/convert.rs    :29-128
//   /root/reference/masp_proofs/src/constants.rs          (window tables :44-94, Montgomery tables :99-173)
// The allocation / constraint order is the Rust statement order: the structure hashes pinned by the reference's
// tests (sapling.rs:733,1026, convert.rs:221) are reproduced bit for bit (tests/test_circuits.py).
#pragma once
#include "cs.h"

namespace masp_host {

// ------------------------------------------------------------------------------------------- window tables
typedef std::vector<std::vector<Coord>> FixedGenerator;
// 3-bit windows [0, 1, ..., 7] * 8^k * G, 84 windows
inline FixedGenerator generate_circuit_generator(JPoint gen) {
    FixedGenerator windows;
    for (int w = 0; w < 84; ++w) {
        std::vector<Coord> coeffs = {{Fr::zero(), Fr::one()}};
        JPoint g = gen;
        for (int k = 0; k < 7; ++k) {
            JAffine a = g.to_affine();
            coeffs.push_back({a.u, a.v});
            g = g.add(gen);
        }
        windows.push_back(coeffs);
        gen = g;  // 8 * gen
    }
    return windows;
}
inline Coord to_montgomery_coords(const JPoint& p) {
    JAffine g = p.to_affine();
    if (g.v == Fr::one()) throw SynthesisError("point at infinity in a Pedersen table");
    if (g.u.is_zero()) return {Fr::zero(), Fr::zero()};
    Fr inv1, invx;
    (Fr::one() - g.v).invert(inv1);
    g.u.invert(invx);
    Fr u = (Fr::one() + g.v) * inv1;
    Fr v = u * invx;
    return {u, v * montgomery_scale()};
}
struct CircuitTables {
    FixedGenerator proof_generation_key, note_commitment_randomness, nullifier_position, value_commitment_randomness, spending_key;
    std::vector<std::vector<std::vector<Coord>>> pedersen;  // [segment][window][4]
};
inline const CircuitTables& tables() {
    static CircuitTables t = [] {
        CircuitTables x;
        const Generators& g = generators();
        x.proof_generation_key = generate_circuit_generator(g.proof_generation_key);
        x.note_commitment_randomness = generate_circuit_generator(g.note_commitment_randomness);
        x.nullifier_position = generate_circuit_generator(g.nullifier_position);
        x.value_commitment_randomness = generate_circuit_generator(g.value_commitment_randomness);
        x.spending_key = generate_circuit_generator(g.spending_key);
        for (int s = 0; s < 6; ++s) {
            JPoint gen = g.pedersen[s];
            std::vector<std::vector<Coord>> windows;
            for (int w = 0; w < 63; ++w) {
                std::vector<Coord> coeffs;
                JPoint p = gen;
                for (int k = 0; k < 4; ++k) {
                    coeffs.push_back(to_montgomery_coords(p));
                    p = p.add(gen);
                }
                windows.push_back(coeffs);
                for (int k = 0; k < 4; ++k) gen = gen.dbl();
            }
            x.pedersen.push_back(windows);
        }
        return x;
    }();
    return t;
}

// 1/(1 + c) and 1/(1 - c) from a single inversion
inline void inv_pair(const Fr& c, Fr& inv_plus, Fr& inv_minus) {
    Fr p = Fr::one() + c, m = Fr::one() - c, i;
    if (!(p * m).invert(i)) throw SynthesisError("DivisionByZero");
    inv_plus = i * m;
    inv_minus = i * p;
}

// Affine coordinates of a whole chain of points from ONE field inversion (Montgomery's trick on the Z coordinates).  The
// in-circuit Edwards additions / doublings allocate the affine result of every step (circuit/ecc.rs `EdwardsPoint::{add,
// double}`), i.e. one inversion per step when computed step by step: ~1 050 of the ~1 500 inversions of a Spend.  The chains
// of a fixed-base multiplication (table lookups) and of a variable-base multiplication are known in advance from the bits,
// so they are first run natively in extended coordinates and converted here; the gadgets then take the results as hints.
inline void batch_to_affine(const std::vector<JPoint>& pts, std::vector<JAffine>& out) {
    const size_t n = pts.size();
    out.resize(n);
    if (!n) return;
    std::vector<Fr> pre(n);
    pre[0] = pts[0].Z;
    for (size_t k = 1; k < n; ++k) pre[k] = pre[k - 1] * pts[k].Z;
    Fr inv;
    if (!pre[n - 1].invert(inv)) throw SynthesisError("DivisionByZero");  // Z = 0 does not occur on the complete curve
    for (size_t k = n; k-- > 0;) {
        const Fr zi = k ? inv * pre[k - 1] : inv;
        inv = inv * pts[k].Z;
        out[k] = {pts[k].U * zi, pts[k].V * zi};
    }
}

// ------------------------------------------------------------------------------------------- Edwards gadget
struct EdwardsPoint {
    AllocatedNum u, v;

    static EdwardsPoint interpret(CS& cs, const AllocatedNum& u, const AllocatedNum& v) {
        AllocatedNum u2 = u.square(cs);
        AllocatedNum v2 = v.square(cs);
        AllocatedNum u2v2 = u2.mul(cs, v2);
        // -u^2 + v^2 = 1 + d u^2 v^2
        MASP_ENFORCE(cs, LC().sub(u2.var).add(v2.var), LC(ONE), LC(ONE).add(u2v2.var, edwards_d()));
        return {u, v};
    }
    static EdwardsPoint witness(CS& cs, const JPoint& p) {
        JAffine a = cs.has_witness() ? p.to_affine() : JAffine{Fr::zero(), Fr::one()};
        AllocatedNum u = AllocatedNum::alloc(cs, a.u);
        AllocatedNum v = AllocatedNum::alloc(cs, a.v);
        return interpret(cs, u, v);
    }
    // `hint`: the affine result if the caller already knows it (batch_to_affine); otherwise one inversion here
    EdwardsPoint dbl(CS& cs, const JAffine* hint = nullptr) const {
        // T = (u + v)^2
        Fr tv = (u.value + v.value).square();
        AllocatedNum t = AllocatedNum::alloc(cs, tv);
        MASP_ENFORCE(cs, LC(u.var).add(v.var), LC(u.var).add(v.var), LC(t.var));
        AllocatedNum a = u.mul(cs, v);
        // C = d A^2
        AllocatedNum c = AllocatedNum::alloc(cs, a.value.square() * edwards_d());
        MASP_ENFORCE(cs, LC().add(a.var, edwards_d()), LC(a.var), LC(c.var));
        // u3 = 2A / (1 + C),  v3 = (T - 2A) / (1 - C): both inverses from one inversion of (1 + C)(1 - C)
        Fr ip = Fr::zero(), im = Fr::zero();
        if (cs.has_witness() && !hint) inv_pair(c.value, ip, im);
        AllocatedNum u3 = AllocatedNum::alloc(cs, hint ? hint->u : a.value.dbl() * ip);
        MASP_ENFORCE(cs, LC(ONE).add(c.var), LC(u3.var), LC(a.var).add(a.var));
        AllocatedNum v3 = AllocatedNum::alloc(cs, hint ? hint->v : (t.value - a.value.dbl()) * im);
        MASP_ENFORCE(cs, LC(ONE).sub(c.var), LC(v3.var), LC(t.var).sub(a.var).sub(a.var));
        return {u3, v3};
    }
    EdwardsPoint add(CS& cs, const EdwardsPoint& o, const JAffine* hint = nullptr) const {
        // U = (u1 + v1)(u2 + v2)
        AllocatedNum uu = AllocatedNum::alloc(cs, (u.value + v.value) * (o.u.value + o.v.value));
        MASP_ENFORCE(cs, LC(u.var).add(v.var), LC(o.u.var).add(o.v.var), LC(uu.var));
        AllocatedNum a = o.v.mul(cs, u);  // A = v2 u1
        AllocatedNum b = o.u.mul(cs, v);  // B = u2 v1
        AllocatedNum c = AllocatedNum::alloc(cs, a.value * b.value * edwards_d());
        MASP_ENFORCE(cs, LC().add(a.var, edwards_d()), LC(b.var), LC(c.var));
        Fr ip = Fr::zero(), im = Fr::zero();
        if (cs.has_witness() && !hint) inv_pair(c.value, ip, im);
        AllocatedNum u3 = AllocatedNum::alloc(cs, hint ? hint->u : (a.value + b.value) * ip);
        MASP_ENFORCE(cs, LC(ONE).add(c.var), LC(u3.var), LC(a.var).add(b.var));
        AllocatedNum v3 = AllocatedNum::alloc(cs, hint ? hint->v : (uu.value - a.value - b.value) * im);
        MASP_ENFORCE(cs, LC(ONE).sub(c.var), LC(v3.var), LC(uu.var).sub(a.var).sub(b.var));
        return {u3, v3};
    }
    void assert_not_small_order(CS& cs) const {
        EdwardsPoint t = dbl(cs).dbl(cs).dbl(cs);
        t.u.assert_nonzero(cs);
    }
    void inputize(CS& cs) const {
        u.inputize(cs);
        v.inputize(cs);
    }
    std::vector<Boolean> repr(CS& cs) const {
        std::vector<Boolean> ub = u.to_bits_le_strict(cs);
        std::vector<Boolean> vb = v.to_bits_le_strict(cs);
        vb.push_back(ub[0]);
        return vb;
    }
    // self if condition else the neutral element (0, 1)
    EdwardsPoint conditionally_select(CS& cs, const Boolean& cond) const {
        bool c = cond.value();
        AllocatedNum up = AllocatedNum::alloc(cs, c ? u.value : Fr::zero());
        MASP_ENFORCE(cs, LC(u.var), cond.lc(Fr::one()), LC(up.var));
        AllocatedNum vp = AllocatedNum::alloc(cs, c ? v.value : Fr::one());
        MASP_ENFORCE(cs, LC(v.var), cond.lc(Fr::one()), LC(vp.var).sub(cond.not_().lc(Fr::one())));
        return {up, vp};
    }
    EdwardsPoint mul(CS& cs, const std::vector<Boolean>& by) const {
        // the two chains (2^i P and the running sum) natively first: one inversion for all their affine coordinates
        std::vector<JAffine> dbl_aff, sum_aff;
        if (cs.has_witness() && by.size() > 1) {
            std::vector<JPoint> dbls, sums;
            JPoint cur = JPoint::from_affine({u.value, v.value}), acc = JPoint::identity();
            for (size_t i = 0; i < by.size(); ++i) {
                if (i > 0) {
                    cur = cur.dbl();
                    dbls.push_back(cur);
                }
                const JPoint sel = by[i].value() ? cur : JPoint::identity();
                acc = i == 0 ? sel : acc.add(sel);
                if (i > 0) sums.push_back(acc);
            }
            batch_to_affine(dbls, dbl_aff);
            batch_to_affine(sums, sum_aff);
        }
        EdwardsPoint curbase = *this, result = *this;
        for (size_t i = 0; i < by.size(); ++i) {
            if (i > 0) curbase = curbase.dbl(cs, dbl_aff.empty() ? nullptr : &dbl_aff[i - 1]);
            EdwardsPoint thisbase = curbase.conditionally_select(cs, by[i]);
            result = i == 0 ? thisbase : result.add(cs, thisbase, sum_aff.empty() ? nullptr : &sum_aff[i - 1]);
        }
        return result;
    }
};

inline EdwardsPoint fixed_base_multiplication(CS& cs, const FixedGenerator& base, const std::vector<Boolean>& by) {
    EdwardsPoint result{};
    size_t nchunks = (by.size() + 2) / 3;
    // the running sums of the selected window entries natively first: one inversion for the whole chain
    std::vector<JAffine> sum_aff;
    if (cs.has_witness() && nchunks > 1) {
        std::vector<JPoint> sums;
        JPoint acc = JPoint::identity();
        for (size_t i = 0; i < nchunks && i < base.size(); ++i) {
            int idx = 0;
            for (int k = 0; k < 3; ++k)
                if (3 * i + k < by.size() && by[3 * i + k].value()) idx |= 1 << k;
            const Coord& c = base[i][idx];
            const JPoint sel = JPoint::from_affine({c.first, c.second});
            acc = i == 0 ? sel : acc.add(sel);
            if (i > 0) sums.push_back(acc);
        }
        batch_to_affine(sums, sum_aff);
    }
    for (size_t i = 0; i < nchunks && i < base.size(); ++i) {
        Boolean chunk[3];
        for (int k = 0; k < 3; ++k) chunk[k] = 3 * i + k < by.size() ? by[3 * i + k] : Boolean::constant(false);
        auto uv = lookup3_xy(cs, chunk, base[i]);
        EdwardsPoint p{uv.first, uv.second};
        result = i == 0 ? p : result.add(cs, p, sum_aff.empty() ? nullptr : &sum_aff[i - 1]);
    }
    return result;
}

// ------------------------------------------------------------------------------------------- Montgomery gadget
struct MontgomeryPoint {
    Num x, y;

    EdwardsPoint into_edwards(CS& cs) const {
        // u = scale * x / y ; v = (x - 1) / (x + 1): both inverses from one inversion of y (x + 1)
        Fr inv_y = Fr::zero(), inv_x1 = Fr::zero();
        if (cs.has_witness()) {
            const Fr x1 = x.value + Fr::one();
            Fr t;
            if (!(y.value * x1).invert(t)) throw SynthesisError("DivisionByZero");
            inv_y = t * x1;
            inv_x1 = t * y.value;
        }
        AllocatedNum u = AllocatedNum::alloc(cs, x.value * montgomery_scale() * inv_y);
        MASP_ENFORCE(cs, y.lc(Fr::one()), LC(u.var), x.lc(montgomery_scale()));
        AllocatedNum v = AllocatedNum::alloc(cs, (x.value - Fr::one()) * inv_x1);
        MASP_ENFORCE(cs, x.lc(Fr::one()).add(ONE), LC(v.var), x.lc(Fr::one()).sub(ONE));
        return {u, v};
    }
    // affine addition, undefined for equal x.  `inv_hint`: 1 / (o.x - x) if the caller already knows it (the Pedersen
    // gadget batches the inversions of a whole segment, see pedersen_segment_hints); checked with one product.
    MontgomeryPoint add(CS& cs, const MontgomeryPoint& o, const Fr* inv_hint = nullptr) const {
        Fr inv = Fr::zero();
        if (cs.has_witness()) {
            const Fr d = o.x.value - x.value;
            if (inv_hint && *inv_hint * d == Fr::one())
                inv = *inv_hint;
            else if (!d.invert(inv))
                throw SynthesisError("DivisionByZero");
        }
        AllocatedNum lambda = AllocatedNum::alloc(cs, (o.y.value - y.value) * inv);
        MASP_ENFORCE(cs, o.x.lc(Fr::one()).sub(x.lc(Fr::one())), LC(lambda.var), o.y.lc(Fr::one()).sub(y.lc(Fr::one())));
        // x'' = lambda^2 - A - x - x'
        AllocatedNum xp = AllocatedNum::alloc(cs, lambda.value.square() - montgomery_a() - x.value - o.x.value);
        MASP_ENFORCE(cs, LC(lambda.var), LC(lambda.var), LC().add(ONE, montgomery_a()).add(x.lc(Fr::one())).add(o.x.lc(Fr::one())).add(xp.var));
        // y'' = -(y + lambda (x'' - x))
        AllocatedNum yp = AllocatedNum::alloc(cs, ((xp.value - x.value) * lambda.value + y.value).neg());
        MASP_ENFORCE(cs, x.lc(Fr::one()).sub(xp.var), LC(lambda.var), LC(yp.var).add(y.lc(Fr::one())));
        return {Num::from(xp), Num::from(yp)};
    }
};

// The ~5 700 affine Montgomery additions of a Spend (one field inversion each for the slope, which the circuit
// allocates: circuit/ecc.rs `MontgomeryPoint::add`) dominate witness synthesis.  For one segment of the Pedersen hash the
// summands are table entries selected by the message bits, so the whole chain of partial sums is first run natively on the
// Edwards form of the same points in extended coordinates (7 products per step, no inversion); the Montgomery abscissa of a
// partial sum (U : V : Z : T) is (Z + V) / (Z - V), so the denominator of the next in-circuit addition is
//   acc.x - x = ((Z + V) - x (Z - V)) / (Z - V)
// and ONE inversion serves the segment.  ed[k] / neg[k]: the k-th summand (jubjub.h pedersen_windows) and its sign;
// x[k]: its Montgomery abscissa as the circuit sees it; hints[k] (k >= 1) = 1 / ((sum_{j<k} summand_j).x - x[k]).
// On any degenerate input (a zero denominator) no hints are produced and the gadget inverts one by one as before.
inline void pedersen_segment_hints(const std::vector<const JPoint::Niels*>& ed, const std::vector<bool>& neg, const std::vector<Fr>& x,
                                   std::vector<Fr>& hints) {
    hints.clear();
    const size_t n = ed.size();
    if (n < 2) return;
    std::vector<Fr> e(n), zmv(n), pre(n);
    JPoint acc = JPoint::from_affine({neg[0] ? ed[0]->u.neg() : ed[0]->u, ed[0]->v});
    for (size_t k = 1; k < n; ++k) {
        zmv[k] = acc.Z - acc.V;
        e[k] = acc.Z + acc.V - x[k] * zmv[k];
        if (e[k].is_zero()) return;
        if (k + 1 < n) acc = acc.add_niels(*ed[k], neg[k]);
    }
    // batch inversion of e[1..n)
    pre[1] = e[1];
    for (size_t k = 2; k < n; ++k) pre[k] = pre[k - 1] * e[k];
    Fr inv;
    if (!pre[n - 1].invert(inv)) return;
    hints.assign(n, Fr::zero());
    for (size_t k = n - 1; k >= 1; --k) {
        const Fr ek_inv = k > 1 ? inv * pre[k - 1] : inv;
        inv = inv * e[k];
        hints[k] = zmv[k] * ek_inv;
    }
}

inline EdwardsPoint pedersen_hash_gadget(CS& cs, const Personalization& pers, const std::vector<Boolean>& msg) {
    std::vector<Boolean> bits;
    for (bool b : pers.bits()) bits.push_back(Boolean::constant(b));
    bits.insert(bits.end(), msg.begin(), msg.end());
    const auto& gens = tables().pedersen;
    bool have_result = false;
    EdwardsPoint edwards_result{};
    size_t pos = 0, seg = 0;
    while (pos < bits.size()) {
        bool have_seg = false;
        MontgomeryPoint seg_result{};
        const auto& windows = gens.at(seg);
        std::vector<Fr> hints;
        if (cs.has_witness()) {  // the segment's summands, straight from the tables
            std::vector<const JPoint::Niels*> ed;
            std::vector<bool> neg;
            std::vector<Fr> xs;
            const PedersenWindows& T = pedersen_windows();
            for (size_t w = 0, q = pos; w < windows.size() && q < bits.size(); ++w) {
                const bool b0 = bits[q++].value(), b1 = q < bits.size() ? bits[q++].value() : false,
                           b2 = q < bits.size() ? bits[q++].value() : false;
                const int idx = (b0 ? 1 : 0) + (b1 ? 2 : 0);
                ed.push_back(&T.e[seg][w][idx]);
                neg.push_back(b2);
                xs.push_back(windows[w][idx].first);
            }
            pedersen_segment_hints(ed, neg, xs, hints);
        }
        for (size_t w = 0; w < windows.size() && pos < bits.size(); ++w) {
            Boolean chunk[3];
            chunk[0] = bits[pos++];
            chunk[1] = pos < bits.size() ? bits[pos++] : Boolean::constant(false);
            chunk[2] = pos < bits.size() ? bits[pos++] : Boolean::constant(false);
            auto xy = lookup3_xy_with_conditional_negation(cs, chunk, windows[w]);
            MontgomeryPoint tmp{xy.first, xy.second};
            if (!have_seg) {
                seg_result = tmp;
                have_seg = true;
            } else {
                seg_result = tmp.add(cs, seg_result, w < hints.size() ? &hints[w] : nullptr);
            }
        }
        EdwardsPoint e = seg_result.into_edwards(cs);
        if (have_result) {
            edwards_result = e.add(cs, edwards_result);
        } else {
            edwards_result = e;
            have_result = true;
        }
        ++seg;
    }
    return edwards_result;
}

// ------------------------------------------------------------------------------------------- witnesses
struct ValueCommitmentW {
    JPoint asset_generator;  // cofactor not cleared
    uint64_t value;
    uint8_t randomness[32];  // jubjub::Fr, little-endian canonical
};
struct MerklePathW {
    std::vector<std::pair<Fr, bool>> auth_path;  // (sibling, current node is the right child), leaf level first
};
struct SpendW {
    ValueCommitmentW vc;
    JPoint ak;
    uint8_t nsk[32];
    JPoint g_d, pk_d;
    uint8_t rcm[32];
    uint8_t ar[32];
    MerklePathW path;
    Fr anchor;
};
struct OutputW {
    ValueCommitmentW vc;
    uint8_t asset_identifier[32];
    JPoint g_d, pk_d;
    uint8_t rcm[32];
    uint8_t esk[32];
};
struct ConvertW {
    ValueCommitmentW vc;
    MerklePathW path;
    Fr anchor;
};

// circuit/sapling.rs:71-137
inline void expose_value_commitment(CS& cs, const ValueCommitmentW& vc, std::vector<Boolean>& asset_generator_bits,
                                    std::vector<Boolean>& value_bits) {
    EdwardsPoint asset_generator = EdwardsPoint::witness(cs, vc.asset_generator);
    asset_generator_bits = asset_generator.repr(cs);
    asset_generator = asset_generator.dbl(cs);
    asset_generator = asset_generator.dbl(cs);
    asset_generator = asset_generator.dbl(cs);
    asset_generator.u.assert_nonzero(cs);
    value_bits = u64_into_boolean_vec_le(cs, vc.value);
    EdwardsPoint value = asset_generator.mul(cs, value_bits);
    std::vector<Boolean> rcv = bits_into_boolean_vec_le(cs, vc.randomness, 252);
    EdwardsPoint rcvp = fixed_base_multiplication(cs, tables().value_commitment_randomness, rcv);
    EdwardsPoint cv = value.add(cs, rcvp);
    cv.inputize(cs);
}

static const int TREE_DEPTH = 32;

// shared by Spend and Convert: ascend the Merkle path from `cur`
inline AllocatedNum merkle_ascend(CS& cs, AllocatedNum cur, const MerklePathW& path, std::vector<Boolean>* position_bits) {
    for (int i = 0; i < TREE_DEPTH; ++i) {
        bool right = cs.has_witness() ? path.auth_path.at(i).second : false;
        Fr sibling = cs.has_witness() ? path.auth_path.at(i).first : Fr::zero();
        Boolean cur_is_right = Boolean::from(AllocatedBit::alloc(cs, right));
        if (position_bits) position_bits->push_back(cur_is_right);
        AllocatedNum path_element = AllocatedNum::alloc(cs, sibling);
        auto lr = AllocatedNum::conditionally_reverse(cs, cur, path_element, cur_is_right);
        std::vector<Boolean> preimage = lr.first.to_bits_le(cs);
        std::vector<Boolean> rb = lr.second.to_bits_le(cs);
        preimage.insert(preimage.end(), rb.begin(), rb.end());
        cur = pedersen_hash_gadget(cs, {false, (unsigned)i}, preimage).u;
    }
    return cur;
}
// ---- the Merkle block of B witnesses in lockstep (proving mode only) ---------------------------------------------------
// merkle_ascend is 60 % of a Spend's (90 % of a Convert's) synthesis time: 32 Pedersen hashes of 172 windows, every window an
// affine Montgomery addition whose slope needs a field inversion that the circuit allocates.  One witness alone has only three
// independent chains of additions (the hash's segments), so its inversions were amortised by running each chain natively in
// extended Edwards coordinates first (7 products per step), normalising, and deriving the slopes from that: ~17 products and a
// dozen small heap allocations per window.  B witnesses side by side have 3 B independent chains: the additions are done
// directly in the affine Montgomery form the circuit wants, ONE inversion per window index for all of them (Montgomery's trick:
// 3 products per element), 3 more for slope, x', y' — ~6.5 products per window — and every value goes straight into the
// assignment in the order the gadgets would have allocated it (tests/test_host_fast_merkle.py compares the two paths byte for
// byte: same variables, same values, same constraint count).
// cs[b]: witness b's constraint system (witness mode, not recording); cur[b]: in the leaf (cm.u), out the root, both as allocated
// numbers; pos_bits[b] (may be null): receives the 32 position bits.  Returns false if a denominator vanished somewhere (never
// for honest inputs: the caller then runs the generic gadgets, which raise the reference's DivisionByZero).
inline bool fr_batch_invert(Fr* v, size_t n, Fr* scratch) {  // v[i] <- 1 / v[i]; false (v untouched) if any is zero
    if (n == 0) return true;
    scratch[0] = v[0];
    for (size_t i = 1; i < n; ++i) scratch[i] = scratch[i - 1] * v[i];
    Fr inv;
    if (!scratch[n - 1].invert(inv)) return false;
    for (size_t i = n; i-- > 1;) {
        const Fr t = inv * scratch[i - 1];
        inv = inv * v[i];
        v[i] = t;
    }
    v[0] = inv;
    return true;
}
inline bool merkle_block_batch(size_t B, CS* const* cs, AllocatedNum* cur, const MerklePathW* const* paths, std::vector<Boolean>* const* pos_bits) {
    constexpr int NBITS = 6 + 255 + 255, NWIN = NBITS / 3, NSEG = 3;
    static_assert(NBITS % 3 == 0, "whole windows");
    const auto& gens = tables().pedersen;
    int seg_first[NSEG + 1];
    {
        int w0 = 0;
        for (int s = 0; s < NSEG; ++s) {
            seg_first[s] = w0;
            w0 += (int)std::min<size_t>(gens.at(s).size(), (size_t)(NWIN - w0));
        }
        seg_first[NSEG] = w0;
        if (w0 != NWIN) return false;
    }
    const int max_len = std::max(std::max(seg_first[1] - seg_first[0], seg_first[2] - seg_first[1]), seg_first[3] - seg_first[2]);
    struct Level {  // what one witness allocates at one level, in computation order
        Fr left, right;
        uint8_t bits[NBITS];
        Fr y[NWIN];                          // the looked-up ordinate of every window (sign applied)
        Fr lam[NWIN], xp[NWIN], yp[NWIN];    // the addition that folds window w (w > first of its segment) into the segment's sum
        Fr u[NSEG], v[NSEG];                 // the segments' sums on the Edwards form
        Fr ed[NSEG][6];                      // uu, a, b, c, u3, v3 of the addition that folds segment s >= 1 into the result
    };
    std::vector<Level> lv(B);
    std::vector<Fr> accx(B * NSEG), accy(B * NSEG), den(B * NSEG), scratch(B * NSEG), wx(B * NSEG);
    std::vector<Fr> value(B);
    for (size_t b = 0; b < B; ++b) value[b] = cur[b].value;
    const Fr one = Fr::one(), A = montgomery_a(), scale = montgomery_scale(), D = edwards_d();
    for (int depth = 0; depth < TREE_DEPTH; ++depth) {
        // ---- bits and table entries
        for (size_t b = 0; b < B; ++b) {
            Level& L = lv[b];
            const auto& node = paths[b]->auth_path.at(depth);
            L.left = node.second ? node.first : value[b];
            L.right = node.second ? value[b] : node.first;
            for (int i = 0; i < 6; ++i) L.bits[i] = (depth >> i) & 1;
            uint8_t le[32];
            L.left.to_bytes(le);
            for (int i = 0; i < 255; ++i) L.bits[6 + i] = (le[i / 8] >> (i % 8)) & 1;
            L.right.to_bytes(le);
            for (int i = 0; i < 255; ++i) L.bits[261 + i] = (le[i / 8] >> (i % 8)) & 1;
        }
        // ---- the three chains of every witness, one window index at a time
        for (int k = 0; k < max_len; ++k) {
            size_t m = 0;
            for (size_t b = 0; b < B; ++b) {
                Level& L = lv[b];
                for (int s = 0; s < NSEG; ++s) {
                    const int w = seg_first[s] + k;
                    if (w >= seg_first[s + 1]) continue;
                    const uint8_t* c = L.bits + 3 * w;
                    const Coord& e = gens[s][k][c[0] + 2 * c[1]];
                    L.y[w] = c[2] ? e.second.neg() : e.second;
                    if (k == 0) {
                        accx[b * NSEG + s] = e.first;
                        accy[b * NSEG + s] = L.y[w];
                    } else {
                        wx[m] = e.first;
                        den[m++] = accx[b * NSEG + s] - e.first;   // (o.x - x) of MontgomeryPoint::add: o = the sum so far
                    }
                }
            }
            if (k == 0) continue;
            if (!fr_batch_invert(den.data(), m, scratch.data())) return false;
            m = 0;
            for (size_t b = 0; b < B; ++b) {
                Level& L = lv[b];
                for (int s = 0; s < NSEG; ++s) {
                    const int w = seg_first[s] + k;
                    if (w >= seg_first[s + 1]) continue;
                    Fr &ax = accx[b * NSEG + s], &ay = accy[b * NSEG + s];
                    const Fr& x = wx[m];
                    const Fr lam = (ay - L.y[w]) * den[m];
                    const Fr xp = lam.square() - A - x - ax;
                    const Fr yp = ((xp - x) * lam + L.y[w]).neg();
                    L.lam[w] = lam;
                    L.xp[w] = xp;
                    L.yp[w] = yp;
                    ax = xp;
                    ay = yp;
                    ++m;
                }
            }
        }
        // ---- into_edwards: u = scale x / y, v = (x - 1) / (x + 1), both from 1 / (y (x + 1))
        for (size_t i = 0; i < B * NSEG; ++i) den[i] = accy[i] * (accx[i] + one);
        if (!fr_batch_invert(den.data(), B * NSEG, scratch.data())) return false;
        for (size_t b = 0; b < B; ++b)
            for (int s = 0; s < NSEG; ++s) {
                const size_t i = b * NSEG + s;
                const Fr x1 = accx[i] + one;
                lv[b].u[s] = accx[i] * scale * (den[i] * x1);
                lv[b].v[s] = (accx[i] - one) * (den[i] * accy[i]);
            }
        // ---- the two Edwards additions: this = segment s, o = the result so far
        std::vector<Fr> ru(B), rv(B);
        for (size_t b = 0; b < B; ++b) {
            ru[b] = lv[b].u[0];
            rv[b] = lv[b].v[0];
        }
        for (int s = 1; s < NSEG; ++s) {
            for (size_t b = 0; b < B; ++b) {
                Fr* e = lv[b].ed[s];
                const Fr &u = lv[b].u[s], &v = lv[b].v[s];
                e[0] = (u + v) * (ru[b] + rv[b]);
                e[1] = rv[b] * u;
                e[2] = ru[b] * v;
                e[3] = e[1] * e[2] * D;
                den[2 * b] = one + e[3];
                den[2 * b + 1] = one - e[3];
            }
            if (!fr_batch_invert(den.data(), 2 * B, scratch.data())) return false;
            for (size_t b = 0; b < B; ++b) {
                Fr* e = lv[b].ed[s];
                e[4] = (e[1] + e[2]) * den[2 * b];
                e[5] = (e[0] - e[1] - e[2]) * den[2 * b + 1];
                ru[b] = e[4];
                rv[b] = e[5];
            }
        }
        // ---- the level's variables, in the gadgets' order (merkle_ascend, pedersen_hash_gadget)
        for (size_t b = 0; b < B; ++b) {
            CS& c = *cs[b];
            const Level& L = lv[b];
            const auto& node = paths[b]->auth_path.at(depth);
            const AllocatedBit right{c.alloc_bit(node.second), node.second};
            if (pos_bits && pos_bits[b]) pos_bits[b]->push_back(Boolean::from(right));
            c.alloc(node.first);
            c.alloc(L.left);
            c.alloc(L.right);
            for (int i = 6; i < NBITS; ++i) c.alloc_bit(L.bits[i]);
            Var last = 0;
            for (int s = 0; s < NSEG; ++s) {
                for (int w = seg_first[s]; w < seg_first[s + 1]; ++w) {
                    c.alloc(L.y[w]);
                    if (3 * w >= 6) c.alloc_bit(L.bits[3 * w] && L.bits[3 * w + 1]);  // (the first two windows are the constant personalisation bits)
                    if (w > seg_first[s]) {
                        c.alloc(L.lam[w]);
                        c.alloc(L.xp[w]);
                        c.alloc(L.yp[w]);
                    }
                }
                last = c.alloc(L.u[s]);
                c.alloc(L.v[s]);
                if (s >= 1) {
                    for (int q = 0; q < 4; ++q) c.alloc(L.ed[s][q]);
                    last = c.alloc(L.ed[s][4]);
                    c.alloc(L.ed[s][5]);
                }
            }
            // constraints: bit 1, reverse 2, two unpackings 2 x 256, per window 1 lookup (+ 1 for its AND bit), 3 per addition,
            // 2 per into_edwards, 6 per Edwards addition
            c.count_constraints(1 + 2 + 512 + NWIN + (NWIN - 2) + 3 * (NWIN - NSEG) + 2 * NSEG + 6 * (NSEG - 1));
            value[b] = ru[b];
            cur[b] = AllocatedNum{last, ru[b]};
        }
    }
    return true;
}

inline void conditional_anchor(CS& cs, const AllocatedNum& cur, const Num& value_num, const Fr& anchor) {
    AllocatedNum rt = AllocatedNum::alloc(cs, anchor);
    // (cur - rt) * value = 0
    MASP_ENFORCE(cs, LC(cur.var).sub(rt.var), value_num.lc(Fr::one()), LC());
    rt.inputize(cs);
}

// circuit/sapling.rs:139-417, in three parts so that the Merkle block of several witnesses can run in lockstep
struct SpendState {
    EdwardsPoint cm;
    Num value_num = Num::zero();
    std::vector<Boolean> nf_preimage, position_bits;
    AllocatedNum cur;
};
inline void synthesize_spend_pre(CS& cs, const SpendW& w, SpendState& st) {
    LcRecordingScope lc_scope(cs.recording());
    EdwardsPoint ak = EdwardsPoint::witness(cs, w.ak);
    ak.assert_not_small_order(cs);
    {
        std::vector<Boolean> ar = bits_into_boolean_vec_le(cs, w.ar, 252);
        EdwardsPoint arp = fixed_base_multiplication(cs, tables().spending_key, ar);
        EdwardsPoint rk = ak.add(cs, arp);
        rk.inputize(cs);
    }
    EdwardsPoint nk;
    {
        std::vector<Boolean> nsk = bits_into_boolean_vec_le(cs, w.nsk, 252);
        nk = fixed_base_multiplication(cs, tables().proof_generation_key, nsk);
    }
    std::vector<Boolean> ivk_preimage = ak.repr(cs);
    std::vector<Boolean> nf_preimage;
    {
        std::vector<Boolean> repr_nk = nk.repr(cs);
        ivk_preimage.insert(ivk_preimage.end(), repr_nk.begin(), repr_nk.end());
        nf_preimage = repr_nk;
    }
    std::vector<Boolean> ivk = blake2s_gadget(cs, ivk_preimage, "MASP_ivk");
    ivk.resize(251);  // jubjub::Fr::CAPACITY
    EdwardsPoint g_d = EdwardsPoint::witness(cs, w.g_d);
    g_d.assert_not_small_order(cs);
    EdwardsPoint pk_d = g_d.mul(cs, ivk);
    std::vector<Boolean> note_contents;
    Num value_num = Num::zero();
    {
        std::vector<Boolean> asset_generator_bits, value_bits;
        expose_value_commitment(cs, w.vc, asset_generator_bits, value_bits);
        Fr coeff = Fr::one();
        for (auto& bit : value_bits) {
            value_num = value_num.add_bool_with_coeff(bit, coeff);
            coeff = coeff.dbl();
        }
        note_contents.insert(note_contents.end(), asset_generator_bits.begin(), asset_generator_bits.end());
        note_contents.insert(note_contents.end(), value_bits.begin(), value_bits.end());
    }
    {
        std::vector<Boolean> r = g_d.repr(cs);
        note_contents.insert(note_contents.end(), r.begin(), r.end());
        r = pk_d.repr(cs);
        note_contents.insert(note_contents.end(), r.begin(), r.end());
    }
    EdwardsPoint cm = pedersen_hash_gadget(cs, {true, 0}, note_contents);
    {
        std::vector<Boolean> rcm = bits_into_boolean_vec_le(cs, w.rcm, 252);
        EdwardsPoint rcmp = fixed_base_multiplication(cs, tables().note_commitment_randomness, rcm);
        cm = cm.add(cs, rcmp);
    }
    st.cm = cm;
    st.value_num = value_num;
    st.nf_preimage = nf_preimage;
    st.cur = cm.u;
}
// ... then the Merkle path from st.cur (merkle_ascend, or merkle_block_batch for several witnesses at once) ...
inline void synthesize_spend_post(CS& cs, const SpendW& w, SpendState& st) {
    LcRecordingScope lc_scope(cs.recording());
    const EdwardsPoint& cm = st.cm;
    std::vector<Boolean>& nf_preimage = st.nf_preimage;
    const std::vector<Boolean>& position_bits = st.position_bits;
    conditional_anchor(cs, st.cur, st.value_num, w.anchor);
    EdwardsPoint rho = cm;
    {
        EdwardsPoint position = fixed_base_multiplication(cs, tables().nullifier_position, position_bits);
        rho = rho.add(cs, position);
    }
    {
        std::vector<Boolean> r = rho.repr(cs);
        nf_preimage.insert(nf_preimage.end(), r.begin(), r.end());
    }
    std::vector<Boolean> nf = blake2s_gadget(cs, nf_preimage, "MASP__nf");
    pack_into_inputs(cs, nf);
}
inline void synthesize_spend(CS& cs, const SpendW& w) {
    LcRecordingScope lc_scope(cs.recording());
    SpendState st;
    synthesize_spend_pre(cs, w, st);
    st.cur = merkle_ascend(cs, st.cur, w.path, &st.position_bits);
    synthesize_spend_post(cs, w, st);
}

// circuit/sapling.rs:419-596
inline void synthesize_output(CS& cs, const OutputW& w) {
    LcRecordingScope lc_scope(cs.recording());
    std::vector<Boolean> note_contents;
    std::vector<Boolean> asset_generator_preimage;
    for (int i = 0; i < 256; ++i)
        asset_generator_preimage.push_back(Boolean::from(AllocatedBit::alloc(cs, (w.asset_identifier[i / 8] >> (i % 8)) & 1)));
    std::vector<Boolean> asset_generator_image = blake2s_gadget(cs, asset_generator_preimage, "MASP__v_");
    std::vector<Boolean> asset_generator_bits, value_bits;
    expose_value_commitment(cs, w.vc, asset_generator_bits, value_bits);
    for (int i = 0; i < 256; ++i) Boolean::enforce_equal(cs, asset_generator_bits[i], asset_generator_image[i]);
    note_contents.insert(note_contents.end(), asset_generator_bits.begin(), asset_generator_bits.end());
    note_contents.insert(note_contents.end(), value_bits.begin(), value_bits.end());
    {
        EdwardsPoint g_d = EdwardsPoint::witness(cs, w.g_d);
        g_d.assert_not_small_order(cs);
        std::vector<Boolean> r = g_d.repr(cs);
        note_contents.insert(note_contents.end(), r.begin(), r.end());
        std::vector<Boolean> esk = bits_into_boolean_vec_le(cs, w.esk, 252);
        EdwardsPoint epk = g_d.mul(cs, esk);
        epk.inputize(cs);
    }
    {
        JAffine pk = cs.has_witness() ? w.pk_d.to_affine() : JAffine{Fr::zero(), Fr::one()};
        uint8_t vle[32];
        pk.v.to_bytes(vle);
        std::vector<Boolean> v_contents = bits_into_boolean_vec_le(cs, vle, 255);
        Boolean sign_bit = Boolean::from(AllocatedBit::alloc(cs, pk.u.is_odd()));
        note_contents.insert(note_contents.end(), v_contents.begin(), v_contents.end());
        note_contents.push_back(sign_bit);
    }
    EdwardsPoint cm = pedersen_hash_gadget(cs, {true, 0}, note_contents);
    {
        std::vector<Boolean> rcm = bits_into_boolean_vec_le(cs, w.rcm, 252);
        EdwardsPoint rcmp = fixed_base_multiplication(cs, tables().note_commitment_randomness, rcm);
        cm = cm.add(cs, rcmp);
    }
    cm.u.inputize(cs);
}

// circuit/convert.rs:29-128 (in three parts like the Spend circuit)
struct ConvertState {
    Num value_num = Num::zero();
    AllocatedNum cur;
};
inline void synthesize_convert_pre(CS& cs, const ConvertW& w, ConvertState& st) {
    LcRecordingScope lc_scope(cs.recording());
    Num value_num = Num::zero();
    std::vector<Boolean> asset_generator_bits, value_bits;
    expose_value_commitment(cs, w.vc, asset_generator_bits, value_bits);
    {
        Fr coeff = Fr::one();
        for (auto& bit : value_bits) {
            value_num = value_num.add_bool_with_coeff(bit, coeff);
            coeff = coeff.dbl();
        }
    }
    EdwardsPoint cm = pedersen_hash_gadget(cs, {true, 0}, asset_generator_bits);
    st.value_num = value_num;
    st.cur = cm.u;
}
inline void synthesize_convert_post(CS& cs, const ConvertW& w, ConvertState& st) {
    LcRecordingScope lc_scope(cs.recording());
    conditional_anchor(cs, st.cur, st.value_num, w.anchor);
}
inline void synthesize_convert(CS& cs, const ConvertW& w) {
    LcRecordingScope lc_scope(cs.recording());
    ConvertState st;
    synthesize_convert_pre(cs, w, st);
    st.cur = merkle_ascend(cs, st.cur, w.path, nullptr);
    synthesize_convert_post(cs, w, st);
}

}  // namespace masp_host

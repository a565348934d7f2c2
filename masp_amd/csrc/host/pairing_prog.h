// The Miller loop of the GPU batch verifier as straight-line PROGRAMS over Fp.
//
// One pair's Miller loop is a chain of ~10 000 dependent 384-bit products when a single lane walks the Fp12 tower; on a
// lone wave each costs ~2.4 us.  But inside one iteration almost everything is independent: the 36 base-field products of
// the Fp12 squaring, the 39 of the sparse line multiplication, the dozen of the point doubling.  So the iteration is not
// written as device code at all: the formulas (the same ones as host/pairing.h: projective doubling / mixed addition on
// the twist, sparse line (c0, c1 v, c2 v w), Karatsuba tower) are written ONCE below as templates over the base field,
// instantiated with a symbolic type that records every Fp operation into a DAG, and the DAG is levelled (ASAP) into
// steps of mutually independent operations with values living in numbered 48-byte slots.  The device kernel
// (device/pairing.hpp) is then a tiny interpreter: one WAVE per pair, slots in LDS, lane k of the wave executes the k-th
// operation of the current step — a doubling iteration is ~6 steps that contain products instead of ~135 products in a row.
// The same program runs on the host interpreter below, which is how it is checked against host/pairing.h without a GPU
// (masp_host_pairing_program_selftest, tests/test_pairing_program.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <tuple>
#include <vector>

#include "pairing.h"

namespace masp_host {
namespace prog {

enum Op : uint32_t { OP_MUL = 0, OP_ADD = 1, OP_SUB = 2, OP_MOV = 3 };

// ---- fixed slots (the loop-carried state of one pair) -------------------------------------------------------
enum : int {
    SLOT_ZERO = 0,
    SLOT_F = 1,    // 12: a.a.a a.a.b a.b.a a.b.b a.c.a a.c.b b.a.a b.a.b b.b.a b.b.b b.c.a b.c.b
    SLOT_T = 13,   // 6: X.a X.b Y.a Y.b Z.a Z.b
    SLOT_P = 19,   // 2: xp yp
    SLOT_Q = 21,   // 4: xq.a xq.b yq.a yq.b
    SLOT_Y = 13,   // Fp12 product program: second operand in 13..24
    N_FIXED = 25,
};

// ---- tracer ---------------------------------------------------------------------------------------------------
struct Tracer {
    struct Node {
        int op;  // -1: input (a = fixed slot)
        int a, b;
    };
    std::vector<Node> nodes;
    std::map<std::tuple<int, int, int>, int> cse;
    int input(int slot) {
        auto key = std::make_tuple(-1, slot, 0);
        auto it = cse.find(key);
        if (it != cse.end()) return it->second;
        nodes.push_back({-1, slot, 0});
        return cse[key] = (int)nodes.size() - 1;
    }
    int emit(Op op, int a, int b) {
        if ((op == OP_MUL || op == OP_ADD) && a > b) std::swap(a, b);
        auto key = std::make_tuple((int)op, a, b);
        auto it = cse.find(key);
        if (it != cse.end()) return it->second;
        nodes.push_back({(int)op, a, b});
        return cse[key] = (int)nodes.size() - 1;
    }
    static Tracer*& cur() {
        static thread_local Tracer* t = nullptr;
        return t;
    }
};
// symbolic base-field element
struct S {
    int id;
    S operator+(const S& o) const { return {Tracer::cur()->emit(OP_ADD, id, o.id)}; }
    S operator-(const S& o) const { return {Tracer::cur()->emit(OP_SUB, id, o.id)}; }
    S operator*(const S& o) const { return {Tracer::cur()->emit(OP_MUL, id, o.id)}; }
    S neg() const { return S{Tracer::cur()->input(SLOT_ZERO)} - *this; }
    S dbl() const { return *this + *this; }
    S sq() const { return *this * *this; }
};

// ---- the tower and the Miller steps, over any base field F (host/pairing.h's formulas) -----------------------------
template <class F>
struct Fp2T {
    F a, b;
    Fp2T operator+(const Fp2T& o) const { return {a + o.a, b + o.b}; }
    Fp2T operator-(const Fp2T& o) const { return {a - o.a, b - o.b}; }
    Fp2T neg() const { return {a.neg(), b.neg()}; }
    Fp2T operator*(const Fp2T& o) const {
        F t0 = a * o.a, t1 = b * o.b;
        return {t0 - t1, (a + b) * (o.a + o.b) - t0 - t1};
    }
    Fp2T sq() const { return {(a + b) * (a - b), (a * b).dbl()}; }
    Fp2T xi() const { return {a - b, a + b}; }
    Fp2T scale(const F& k) const { return {a * k, b * k}; }
};
template <class F>
struct Fp6T {
    typedef Fp2T<F> E;
    E a, b, c;
    Fp6T operator+(const Fp6T& o) const { return {a + o.a, b + o.b, c + o.c}; }
    Fp6T operator-(const Fp6T& o) const { return {a - o.a, b - o.b, c - o.c}; }
    Fp6T operator*(const Fp6T& o) const {
        E v0 = a * o.a, v1 = b * o.b, v2 = c * o.c;
        return {v0 + ((b + c) * (o.b + o.c) - v1 - v2).xi(), (a + b) * (o.a + o.b) - v0 - v1 + v2.xi(), (a + c) * (o.a + o.c) - v0 - v2 + v1};
    }
    Fp6T mul_01(const E& x0, const E& x1) const {
        E v0 = a * x0, v1 = b * x1;
        return {v0 + (c * x1).xi(), (a + b) * (x0 + x1) - v0 - v1, v1 + c * x0};
    }
    Fp6T mul_1(const E& x1) const { return {(c * x1).xi(), a * x1, b * x1}; }
    Fp6T mulv() const { return {c.xi(), a, b}; }
};
template <class F>
struct Fp12T {
    typedef Fp2T<F> E;
    Fp6T<F> a, b;
    Fp12T operator*(const Fp12T& o) const {
        Fp6T<F> t0 = a * o.a, t1 = b * o.b;
        return {t0 + t1.mulv(), (a + b) * (o.a + o.b) - t0 - t1};
    }
    Fp12T sq() const {
        Fp6T<F> ab = a * b;
        return {(a + b) * (a + b.mulv()) - ab - ab.mulv(), ab + ab};
    }
    Fp12T mul_line(const E& c0, const E& c1, const E& c2) const {
        Fp6T<F> t0 = a.mul_01(c0, c1), t1 = b.mul_1(c2);
        return {t0 + t1.mulv(), (a + b).mul_01(c0, c1 + c2) - t0 - t1};
    }
};
template <class F>
struct MillerT {
    typedef Fp2T<F> E;
    F xp, yp;
    E xq, yq;
    E X, Y, Z;
    // T <- 2T ; f <- f * l_{T,T}(P)
    void dbl_step(Fp12T<F>& f) {
        E Y2 = Y.sq(), Z2 = Z.sq(), X2 = X.sq();
        E bz = Z2.xi();
        bz = bz + bz;
        bz = bz + bz;
        E Ee = bz + bz + bz;
        E YZ2 = Y * Z;
        YZ2 = YZ2 + YZ2;
        E X23 = X2 + X2 + X2;
        f = f.mul_line(Y2 - Ee, X23.scale(xp).neg(), YZ2.scale(yp));
        E E3 = Ee + Ee + Ee, XY = X * Y;
        E EY = Ee * Y2, E2 = Ee.sq();
        E EY2 = EY + EY, EY6 = EY2 + EY2 + EY2;
        X = (XY + XY) * (Y2 - E3);
        Y = Y2.sq() + EY6 - (E2 + E2 + E2);
        E Y2Z = Y2 * YZ2;
        Y2Z = Y2Z + Y2Z;
        Z = Y2Z + Y2Z;
    }
    // T <- T + Q ; f <- f * l_{T,Q}(P)
    void add_step(Fp12T<F>& f) {
        E N = Y - yq * Z, D = X - xq * Z;
        f = f.mul_line(N * xq - D * yq, N.scale(xp).neg(), D.scale(yp));
        E D2 = D.sq(), D3 = D2 * D, xqZ = xq * Z;
        E A = N.sq() * Z - D2 * (X + xqZ);
        E Y3 = N * (xqZ * D2 - A) - yq * Z * D3;
        X = A * D;
        Y = Y3;
        Z = Z * D3;
    }
};

// ---- program = levelled DAG --------------------------------------------------------------------------------------
struct Program {
    std::vector<uint32_t> ops;        // op | dst << 2 | a << 12 | b << 22   (slots < 1024)
    std::vector<uint32_t> step_start;  // ops of step s: [step_start[s], step_start[s + 1])
    uint32_t n_slots = 0;
    uint32_t n_mul = 0, n_mul_steps = 0;
    static uint32_t enc(uint32_t op, uint32_t dst, uint32_t a, uint32_t b) { return op | dst << 2 | a << 12 | b << 22; }
};
// outputs: (node id, fixed slot it must end up in)
inline Program levelize(const Tracer& t, const std::vector<std::pair<int, int>>& outputs) {
    const int n = (int)t.nodes.size();
    std::vector<char> live(n, 0);
    std::vector<int> stack;
    for (auto& o : outputs) stack.push_back(o.first);
    while (!stack.empty()) {
        int v = stack.back();
        stack.pop_back();
        if (live[v]) continue;
        live[v] = 1;
        if (t.nodes[v].op >= 0) {
            stack.push_back(t.nodes[v].a);
            stack.push_back(t.nodes[v].b);
        }
    }
    // Steps.  A product costs ~20x an addition and a step costs its slowest operation, so all products of the same
    // product-depth go into ONE step (the early ones wait for the late ones), with the additions that feed them in cheap
    // steps of their own in between:  [adds of level 0] [products of level 1] [adds of level 1] [products of level 2] ...
    std::vector<int> step(n, 0), last_use(n, 0), md(n, 0), ad(n, 0);
    int max_md = 0;
    for (int v = 0; v < n; ++v) {
        if (!live[v] || t.nodes[v].op < 0) continue;
        const int a = t.nodes[v].a, b = t.nodes[v].b;
        if (t.nodes[v].op == OP_MUL) {
            md[v] = std::max(md[a], md[b]) + 1;
        } else {
            md[v] = std::max(md[a], md[b]);
            // depth among the additions of this level (an operand of a lower level, or a product of this level, counts 0)
            auto dep = [&](int x) { return (t.nodes[x].op >= 0 && t.nodes[x].op != OP_MUL && md[x] == md[v]) ? ad[x] : 0; };
            ad[v] = std::max(dep(a), dep(b)) + 1;
        }
        max_md = std::max(max_md, md[v]);
    }
    std::vector<int> add_depth(max_md + 1, 0);
    for (int v = 0; v < n; ++v)
        if (live[v] && t.nodes[v].op >= 0 && t.nodes[v].op != OP_MUL) add_depth[md[v]] = std::max(add_depth[md[v]], ad[v]);
    std::vector<int> mul_step(max_md + 2, 0);  // step of the products of level L (L >= 1); additions of level L follow it
    int max_step = 0;
    {
        int cur = add_depth[0];                // additions of level 0: steps 1 .. add_depth[0]
        for (int L = 1; L <= max_md; ++L) {
            mul_step[L] = ++cur;
            cur += add_depth[L];
        }
        max_step = cur;
    }
    for (int v = 0; v < n; ++v) {
        if (!live[v] || t.nodes[v].op < 0) continue;
        step[v] = t.nodes[v].op == OP_MUL ? mul_step[md[v]] : (md[v] == 0 ? 0 : mul_step[md[v]]) + ad[v];
    }
    const int mov_step = max_step + 1;  // the outputs move into their fixed slots after every read of the old state
    for (int v = 0; v < n; ++v)
        if (live[v] && t.nodes[v].op >= 0) {
            last_use[t.nodes[v].a] = std::max(last_use[t.nodes[v].a], step[v]);
            last_use[t.nodes[v].b] = std::max(last_use[t.nodes[v].b], step[v]);
        }
    for (auto& o : outputs) last_use[o.first] = mov_step;
    // slots: inputs sit in their fixed slots; temporaries come from a free list; a slot is released only AFTER the step of
    // its value's last reader, so no step reads and writes the same slot (lanes run the ops of a step in no order)
    std::vector<int> slot(n, -1);
    std::vector<int> free_list;
    int next_slot = N_FIXED;
    std::vector<std::vector<int>> by_step(mov_step + 1), expiring(mov_step + 2);
    for (int v = 0; v < n; ++v) {
        if (!live[v]) continue;
        if (t.nodes[v].op < 0)
            slot[v] = t.nodes[v].a;
        else
            by_step[step[v]].push_back(v);
    }
    Program P;
    P.step_start.push_back(0);
    for (int s = 1; s <= max_step; ++s) {
        bool has_mul = false;
        for (int v : by_step[s]) {
            if (free_list.empty())
                slot[v] = next_slot++;
            else {
                slot[v] = free_list.back();
                free_list.pop_back();
            }
            expiring[last_use[v]].push_back(v);
            P.ops.push_back(Program::enc((uint32_t)t.nodes[v].op, (uint32_t)slot[v], (uint32_t)slot[t.nodes[v].a], (uint32_t)slot[t.nodes[v].b]));
            if (t.nodes[v].op == OP_MUL) {
                ++P.n_mul;
                has_mul = true;
            }
        }
        if (has_mul) ++P.n_mul_steps;
        P.step_start.push_back((uint32_t)P.ops.size());
        for (int v : expiring[s]) free_list.push_back(slot[v]);
    }
    for (auto& o : outputs) P.ops.push_back(Program::enc(OP_MOV, (uint32_t)o.second, (uint32_t)slot[o.first], 0));
    P.step_start.push_back((uint32_t)P.ops.size());
    P.n_slots = (uint32_t)next_slot;
    return P;
}

// ---- the three programs ---------------------------------------------------------------------------------------------
struct PairingPrograms {
    Program dbl;   // f <- f^2 * l_{T,T}(P),  T <- 2T
    Program add;   // f <- f   * l_{T,Q}(P),  T <- T + Q
    Program mul12; // f <- f * y   (y in slots SLOT_Y .. SLOT_Y + 11): the product tree over the pairs
    uint32_t n_slots = 0;
};
namespace detail {
inline Fp12T<S> sym_f(Tracer& t, int base) {
    S v[12];
    for (int i = 0; i < 12; ++i) v[i] = S{t.input(base + i)};
    return {{{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}}, {{v[6], v[7]}, {v[8], v[9]}, {v[10], v[11]}}};
}
inline void out_f(std::vector<std::pair<int, int>>& o, const Fp12T<S>& f) {
    const S v[12] = {f.a.a.a, f.a.a.b, f.a.b.a, f.a.b.b, f.a.c.a, f.a.c.b, f.b.a.a, f.b.a.b, f.b.b.a, f.b.b.b, f.b.c.a, f.b.c.b};
    for (int i = 0; i < 12; ++i) o.push_back({v[i].id, SLOT_F + i});
}
inline MillerT<S> sym_pair(Tracer& t) {
    auto in = [&](int s) { return S{t.input(s)}; };
    MillerT<S> m;
    m.xp = in(SLOT_P);
    m.yp = in(SLOT_P + 1);
    m.xq = {in(SLOT_Q), in(SLOT_Q + 1)};
    m.yq = {in(SLOT_Q + 2), in(SLOT_Q + 3)};
    m.X = {in(SLOT_T), in(SLOT_T + 1)};
    m.Y = {in(SLOT_T + 2), in(SLOT_T + 3)};
    m.Z = {in(SLOT_T + 4), in(SLOT_T + 5)};
    return m;
}
inline void out_t(std::vector<std::pair<int, int>>& o, const MillerT<S>& m) {
    const S v[6] = {m.X.a, m.X.b, m.Y.a, m.Y.b, m.Z.a, m.Z.b};
    for (int i = 0; i < 6; ++i) o.push_back({v[i].id, SLOT_T + i});
}
}  // namespace detail
inline const PairingPrograms& pairing_programs() {
    static const PairingPrograms P = [] {
        PairingPrograms pp;
        {
            Tracer t;
            Tracer::cur() = &t;
            Fp12T<S> f = detail::sym_f(t, SLOT_F).sq();
            MillerT<S> m = detail::sym_pair(t);
            m.dbl_step(f);
            std::vector<std::pair<int, int>> outs;
            detail::out_f(outs, f);
            detail::out_t(outs, m);
            pp.dbl = levelize(t, outs);
        }
        {
            Tracer t;
            Tracer::cur() = &t;
            Fp12T<S> f = detail::sym_f(t, SLOT_F);
            MillerT<S> m = detail::sym_pair(t);
            m.add_step(f);
            std::vector<std::pair<int, int>> outs;
            detail::out_f(outs, f);
            detail::out_t(outs, m);
            pp.add = levelize(t, outs);
        }
        {
            Tracer t;
            Tracer::cur() = &t;
            Fp12T<S> f = detail::sym_f(t, SLOT_F) * detail::sym_f(t, SLOT_Y);
            std::vector<std::pair<int, int>> outs;
            detail::out_f(outs, f);
            pp.mul12 = levelize(t, outs);
        }
        Tracer::cur() = nullptr;
        pp.n_slots = std::max(pp.dbl.n_slots, std::max(pp.add.n_slots, pp.mul12.n_slots));
        return pp;
    }();
    return P;
}

// ---- host interpreter (reference semantics of the device kernel) ------------------------------------------------------
inline void run_program(const Program& P, bls::Fp* slots) {
    for (size_t s = 0; s + 1 < P.step_start.size(); ++s)
        for (uint32_t k = P.step_start[s]; k < P.step_start[s + 1]; ++k) {
            const uint32_t w = P.ops[k], op = w & 3, dst = (w >> 2) & 1023, a = (w >> 12) & 1023, b = (w >> 22) & 1023;
            slots[dst] = op == OP_MUL ? slots[a] * slots[b] : op == OP_ADD ? slots[a] + slots[b] : op == OP_SUB ? slots[a] - slots[b] : slots[a];
        }
}
// Miller loop of one pair through the programs, as the device runs it: f_{|x|,Q}(P) conjugated (host/pairing.h's convention)
inline bls::Fp12 miller_by_program(const bls::G1A& Pt, const bls::G2A& Q) {
    const PairingPrograms& pp = pairing_programs();
    std::vector<bls::Fp> sl(pp.n_slots, bls::Fp::zero());
    sl[SLOT_F] = bls::Fp::one();
    sl[SLOT_T] = Q.x.a; sl[SLOT_T + 1] = Q.x.b; sl[SLOT_T + 2] = Q.y.a; sl[SLOT_T + 3] = Q.y.b; sl[SLOT_T + 4] = bls::Fp::one();
    sl[SLOT_P] = Pt.x; sl[SLOT_P + 1] = Pt.y;
    sl[SLOT_Q] = Q.x.a; sl[SLOT_Q + 1] = Q.x.b; sl[SLOT_Q + 2] = Q.y.a; sl[SLOT_Q + 3] = Q.y.b;
    const uint64_t xabs = 0xd201000000010000ull;
    for (int b = 62; b >= 0; --b) {
        run_program(pp.dbl, sl.data());      // (squaring f = 1 in the first iteration is harmless)
        if ((xabs >> b) & 1) run_program(pp.add, sl.data());
    }
    const bls::Fp* f = &sl[SLOT_F];
    bls::Fp12 r = {{{f[0], f[1]}, {f[2], f[3]}, {f[4], f[5]}}, {{f[6], f[7]}, {f[8], f[9]}, {f[10], f[11]}}};
    return r.conj();
}

}  // namespace prog
}  // namespace masp_host

// R1CS recording constraint system + the gadget library the MASP circuits are written against.
//
// Restates `bellman::{ConstraintSystem, LinearCombination}` and `bellman::gadgets::{boolean, num, lookup, uint32,
// multieq, blake2s, multipack}` (= bellpepper-core 0.2.1, un-vendored, /root/reference/Cargo.lock:154; used at
// /root/reference/masp_proofs/src/circuit/sapling.rs:5,19, ecc.rs:5-13, pedersen_hash.rs:4-6, gadgets.rs:1-2) exactly
// as specified in SURVEY.md Appendix B: the order of allocations and constraints is the reference's, which is what
// the pinned `TestConstraintSystem::hash()` values check (circuit/sapling.rs:733,1026; circuit/convert.rs:221).
//
// One code path, three uses (like bellman's KeypairAssembly / ProvingAssignment / TestConstraintSystem):
//   record = true,  witness = false : set-up   -> static R1CS (CSR) and the structure hash
//   record = false, witness = true  : proving  -> input_assignment, aux_assignment only (a, b, c are evaluated
//                                                 on the GPU from the static R1CS)
//   record = true,  witness = true  : testing  -> is_satisfied()
#pragma once
#include <array>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "jubjub.h"

namespace masp_host {

struct SynthesisError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

typedef uint32_t Var;  // input i -> i ; aux j -> AUX | j
static const Var AUX = 0x80000000u;
static const Var ONE = 0;

// In proving mode nothing records constraints, so the three linear combinations are never even built.
#define MASP_ENFORCE(CS_, A_, B_, C_)              \
    do {                                           \
        if ((CS_).recording())                     \
            (CS_).enforce((A_), (B_), (C_));       \
        else                                       \
            (CS_).count_constraint();              \
    } while (0)

// Linear combinations exist to be recorded: while no synthesis of this thread records (proving mode), building one is
// a no-op — the gadgets' `Num`s carry their LC along, which cost ~17 000 small heap allocations per Spend for nothing.
// The flag is scoped to the SYNTHESIS CALL (LcRecordingScope inside synthesize_*), not to the lifetime of a constraint system: a
// recording CS outlives its synthesis (masp_host_circuit_setup keeps it in the handle) and may be freed on another thread than the one
// that made it — tied to the object, that left the creating thread at +1 and took the freeing thread to -1, after which a recording
// synthesis on the latter built empty LCs and "checked" 0 * 0 = 0.
inline int& lc_recording_depth() {
    static thread_local int d = 0;
    return d;
}
struct LcRecordingScope {
    const bool on_;
    explicit LcRecordingScope(bool on) : on_(on) {
        if (on_) ++lc_recording_depth();
    }
    ~LcRecordingScope() {
        if (on_ && --lc_recording_depth() < 0) abort();  // cannot happen: every decrement is paired with this object's increment
    }
    LcRecordingScope(const LcRecordingScope&) = delete;
    LcRecordingScope& operator=(const LcRecordingScope&) = delete;
};
struct LC {
    std::vector<std::pair<Var, Fr>> t;
    static bool on() { return lc_recording_depth() > 0; }
    LC() {}
    LC(Var v) {
        if (on()) t.push_back({v, Fr::one()});
    }
    LC& add(Var v, const Fr& c) {
        if (on()) t.push_back({v, c});
        return *this;
    }
    LC& add(Var v) { return add(v, Fr::one()); }
    LC& sub(Var v) { return add(v, Fr::one().neg()); }
    LC& add(const LC& o) {
        if (on()) t.insert(t.end(), o.t.begin(), o.t.end());
        return *this;
    }
    LC& sub(const LC& o) {
        if (on())
            for (auto& x : o.t) t.push_back({x.first, x.second.neg()});
        return *this;
    }
    LC scaled(const Fr& k) const {
        LC r;
        if (on())
            for (auto& x : t) r.t.push_back({x.first, x.second * k});
        return r;
    }
};

class CS {
  public:
    // ext / ext_cap: where the auxiliary assignment goes, if the caller wants it in place (Fr is the Montgomery residue as four
    // little-endian u64 limbs: masp_hip_job::aux_form = MASP_HIP_AUX_MONTGOMERY reads exactly this memory — no 4 MB vector per
    // witness, no copy at the end); otherwise the object owns its storage
    CS(bool record, bool witness, Fr* ext = nullptr, size_t ext_cap = 0) : record_(record), witness_(witness) {
        inputs_.push_back(Fr::one());  // ONE
        if (ext) {
            aux_p_ = ext;
            aux_cap_ = ext_cap;
        } else if (witness) {
            grow(1u << 17);  // the largest circuit (Spend) has 100 497 auxiliary variables
        }
        ext_ = ext != nullptr;
    }
    CS(const CS&) = delete;
    CS& operator=(const CS&) = delete;
    bool has_witness() const { return witness_; }
    bool recording() const { return record_; }

    Var alloc(const Fr& value) {
        if (aux_n_ == aux_cap_) grow(aux_cap_ ? 2 * aux_cap_ : 1024);
        aux_p_[aux_n_] = value;
        return AUX | (Var)(aux_n_++);
    }
    Var alloc_bit(bool value) {
        static const Fr bit[2] = {Fr::zero(), Fr::one()};
        if (aux_n_ == aux_cap_) grow(aux_cap_ ? 2 * aux_cap_ : 1024);
        aux_p_[aux_n_] = bit[value];
        return AUX | (Var)(aux_n_++);
    }
    Var alloc_input(const Fr& value) {
        inputs_.push_back(value);
        return (Var)(inputs_.size() - 1);
    }
    void count_constraint() { ++n_constraints_; }
    void count_constraints(size_t k) { n_constraints_ += k; }
    // proving mode: forget the auxiliary variables from index `n` on (a block that is to be redone another way; the constraint
    // count is left alone: nothing reads it in proving mode)
    void truncate_aux(size_t n) {
        if (n < aux_n_) aux_n_ = n;
    }  // proving mode: a block whose constraints nobody records (merkle_block_batch)
    void enforce(const LC& a, const LC& b, const LC& c) {
        ++n_constraints_;
        if (!record_) return;
        push_row(0, a);
        push_row(1, b);
        push_row(2, c);
    }
    Fr value(Var v) const { return (v & AUX) ? aux_p_[v & ~AUX] : inputs_[v]; }
    Fr eval(const LC& lc) const {
        Fr acc = Fr::zero();
        for (auto& x : lc.t) acc = acc + x.second * value(x.first);
        return acc;
    }
    size_t num_inputs() const { return inputs_.size(); }
    size_t num_aux() const { return aux_n_; }
    bool aux_in_place() const { return ext_; }
    size_t num_constraints() const { return n_constraints_; }
    const std::vector<Fr>& inputs() const { return inputs_; }
    const Fr* aux() const { return aux_p_; }

    // merged, zero-free rows (inputs first, then aux, ascending) — the form hashed by TestConstraintSystem and
    // the form whose pattern equals bellperson's density trackers
    struct Matrix {
        std::vector<uint32_t> rowptr{0};
        std::vector<Var> col;
        std::vector<Fr> coef;
    };
    const Matrix& matrix(int i) const { return m_[i]; }

    // index of the first unsatisfied constraint, or -1
    long first_unsatisfied() const {
        for (size_t r = 0; r + 1 < m_[0].rowptr.size(); ++r) {
            Fr v[3];
            for (int i = 0; i < 3; ++i) {
                Fr acc = Fr::zero();
                for (uint32_t t = m_[i].rowptr[r]; t < m_[i].rowptr[r + 1]; ++t) acc = acc + m_[i].coef[t] * value(m_[i].col[t]);
                v[i] = acc;
            }
            if (v[0] * v[1] != v[2]) return (long)r;
        }
        return -1;
    }

    // hex BLAKE2s-256 over the structure, byte-for-byte `TestConstraintSystem::hash()`
    std::string hash() const {
        Blake2s h;
        auto put64 = [&](uint64_t x) {
            uint8_t b[8];
            for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(x >> (56 - 8 * i));
            h.update(b, 8);
        };
        put64(inputs_.size());
        put64(aux_n_);
        put64(n_constraints_);
        for (size_t r = 0; r + 1 < m_[0].rowptr.size(); ++r)
            for (int i = 0; i < 3; ++i) {
                uint32_t lo = m_[i].rowptr[r], hi = m_[i].rowptr[r + 1];
                put64(hi - lo);
                for (uint32_t t = lo; t < hi; ++t) {
                    Var v = m_[i].col[t];
                    uint8_t tag = (v & AUX) ? 'A' : 'I';
                    h.update(&tag, 1);
                    put64(v & ~AUX);
                    uint8_t le[32], be[32];
                    m_[i].coef[t].to_bytes(le);
                    for (int k = 0; k < 32; ++k) be[k] = le[31 - k];
                    h.update(be, 32);
                }
            }
        uint8_t d[32];
        h.finalize(d);
        static const char* hex = "0123456789abcdef";
        std::string s;
        for (int i = 0; i < 32; ++i) {
            s.push_back(hex[d[i] >> 4]);
            s.push_back(hex[d[i] & 15]);
        }
        return s;
    }

  private:
    void push_row(int mi, const LC& lc) {
        std::map<uint64_t, Fr> acc;  // key orders inputs before aux, then by index
        for (auto& x : lc.t) {
            uint64_t key = ((uint64_t)((x.first & AUX) ? 1 : 0) << 32) | (x.first & ~AUX);
            auto it = acc.find(key);
            if (it == acc.end())
                acc.emplace(key, x.second);
            else
                it->second = it->second + x.second;
        }
        Matrix& M = m_[mi];
        for (auto& kv : acc) {
            if (kv.second.is_zero()) continue;
            M.col.push_back(((kv.first >> 32) ? AUX : 0) | (Var)(kv.first & 0xffffffffu));
            M.coef.push_back(kv.second);
        }
        M.rowptr.push_back((uint32_t)M.col.size());
    }
    bool record_, witness_;
    std::vector<Fr> inputs_;
    std::unique_ptr<Fr[]> own_;   // (not a vector: its resize would clear 4 MB per witness for nothing)
    Fr* aux_p_ = nullptr;
    size_t aux_n_ = 0, aux_cap_ = 0;
    bool ext_ = false;
    void grow(size_t cap) {
        if (ext_) throw SynthesisError("auxiliary assignment larger than the caller's buffer");
        std::unique_ptr<Fr[]> bigger(new Fr[cap]);
        if (aux_n_) memcpy(bigger.get(), aux_p_, sizeof(Fr) * aux_n_);
        own_ = std::move(bigger);
        aux_p_ = own_.get();
        aux_cap_ = cap;
    }
    size_t n_constraints_ = 0;
    Matrix m_[3];
};

// =================================================================================================== boolean
struct AllocatedBit {
    Var var;
    bool value;  // meaningful only with a witness

    static AllocatedBit alloc(CS& cs, bool value) {
        Var v = cs.alloc_bit(value);
        // (1 - a) * a = 0
        MASP_ENFORCE(cs, LC(ONE).sub(v), LC(v), LC());
        return {v, value};
    }
    // a may be true only if must_be_false is false:  (1 - must_be_false - a) * a = 0
    static AllocatedBit alloc_conditionally(CS& cs, bool value, const AllocatedBit& must_be_false) {
        Var v = cs.alloc_bit(value);
        MASP_ENFORCE(cs, LC(ONE).sub(must_be_false.var).sub(v), LC(v), LC());
        return {v, value};
    }
    static AllocatedBit xor_(CS& cs, const AllocatedBit& a, const AllocatedBit& b) {
        bool r = a.value ^ b.value;
        Var v = cs.alloc_bit(r);
        // (a + a) * b = a + b - c
        MASP_ENFORCE(cs, LC(a.var).add(a.var), LC(b.var), LC(a.var).add(b.var).sub(v));
        return {v, r};
    }
    static AllocatedBit and_(CS& cs, const AllocatedBit& a, const AllocatedBit& b) {
        bool r = a.value && b.value;
        Var v = cs.alloc_bit(r);
        MASP_ENFORCE(cs, LC(a.var), LC(b.var), LC(v));
        return {v, r};
    }
    static AllocatedBit and_not(CS& cs, const AllocatedBit& a, const AllocatedBit& b) {
        bool r = a.value && !b.value;
        Var v = cs.alloc_bit(r);
        MASP_ENFORCE(cs, LC(a.var), LC(ONE).sub(b.var), LC(v));
        return {v, r};
    }
    static AllocatedBit nor(CS& cs, const AllocatedBit& a, const AllocatedBit& b) {
        bool r = !a.value && !b.value;
        Var v = cs.alloc_bit(r);
        MASP_ENFORCE(cs, LC(ONE).sub(a.var), LC(ONE).sub(b.var), LC(v));
        return {v, r};
    }
};

struct Boolean {
    enum Kind { IS, NOT, CONST } kind;
    AllocatedBit bit;  // for IS / NOT
    bool c;            // for CONST

    static Boolean constant(bool b) { return {CONST, {0, false}, b}; }
    static Boolean from(const AllocatedBit& a) { return {IS, a, false}; }
    bool is_constant() const { return kind == CONST; }
    bool value() const { return kind == CONST ? c : kind == IS ? bit.value : !bit.value; }
    Boolean not_() const {
        if (kind == CONST) return constant(!c);
        return {kind == IS ? NOT : IS, bit, false};
    }
    LC lc(const Fr& coeff) const {
        LC r;
        if (kind == CONST) {
            if (c) r.add(ONE, coeff);
        } else if (kind == IS) {
            r.add(bit.var, coeff);
        } else {
            r.add(ONE, coeff).add(bit.var, coeff.neg());
        }
        return r;
    }
    static Boolean xor_(CS& cs, const Boolean& a, const Boolean& b) {
        if (a.kind == CONST) return a.c ? b.not_() : b;
        if (b.kind == CONST) return b.c ? a.not_() : a;
        if (a.kind != b.kind) {  // (Is, Not) in either order -> not(xor(is, not))
            const Boolean& is = a.kind == IS ? a : b;
            const Boolean& nt = a.kind == IS ? b : a;
            return from(AllocatedBit::xor_(cs, is.bit, nt.bit)).not_();
        }
        return from(AllocatedBit::xor_(cs, a.bit, b.bit));
    }
    static Boolean and_(CS& cs, const Boolean& a, const Boolean& b) {
        if (a.kind == CONST) return a.c ? b : constant(false);
        if (b.kind == CONST) return b.c ? a : constant(false);
        if (a.kind != b.kind) {
            const Boolean& is = a.kind == IS ? a : b;
            const Boolean& nt = a.kind == IS ? b : a;
            return from(AllocatedBit::and_not(cs, is.bit, nt.bit));
        }
        if (a.kind == NOT) return from(AllocatedBit::nor(cs, a.bit, b.bit));
        return from(AllocatedBit::and_(cs, a.bit, b.bit));
    }
    static void enforce_equal(CS& cs, const Boolean& a, const Boolean& b) {
        if (a.kind == CONST && b.kind == CONST) {
            if (a.c != b.c) throw SynthesisError("Unsatisfiable");
            return;
        }
        if (a.kind == CONST || b.kind == CONST) {
            const Boolean& k = a.kind == CONST ? a : b;
            const Boolean& o = a.kind == CONST ? b : a;
            if (k.c)
                MASP_ENFORCE(cs, LC(), LC(), LC(ONE).sub(o.lc(Fr::one())));
            else
                MASP_ENFORCE(cs, LC(), LC(), o.lc(Fr::one()));
            return;
        }
        MASP_ENFORCE(cs, LC(), LC(), a.lc(Fr::one()).sub(b.lc(Fr::one())));
    }
};

inline std::vector<Boolean> u64_into_boolean_vec_le(CS& cs, uint64_t value) {
    std::vector<Boolean> out;
    for (int i = 0; i < 64; ++i) out.push_back(Boolean::from(AllocatedBit::alloc(cs, (value >> i) & 1)));
    return out;
}
// `nbits` low bits of a 256-bit little-endian value, LSB first, each a plain allocated bit (no range check):
// circuit/gadgets.rs:6-50 (252 bits for jubjub::Fr, 255 for bls12_381::Scalar)
inline std::vector<Boolean> bits_into_boolean_vec_le(CS& cs, const uint8_t* le32, int nbits) {
    std::vector<Boolean> out;
    out.reserve(nbits);
    for (int i = 0; i < nbits; ++i) out.push_back(Boolean::from(AllocatedBit::alloc(cs, (le32[i / 8] >> (i % 8)) & 1)));
    return out;
}

// =================================================================================================== num
struct AllocatedNum {
    Var var;
    Fr value;

    static AllocatedNum alloc(CS& cs, const Fr& value) { return {cs.alloc(value), value}; }
    void inputize(CS& cs) const {
        Var in = cs.alloc_input(value);
        MASP_ENFORCE(cs, LC(in), LC(ONE), LC(var));
    }
    AllocatedNum mul(CS& cs, const AllocatedNum& o) const {
        AllocatedNum r = alloc(cs, value * o.value);
        MASP_ENFORCE(cs, LC(var), LC(o.var), LC(r.var));
        return r;
    }
    AllocatedNum square(CS& cs) const {
        AllocatedNum r = alloc(cs, value.square());
        MASP_ENFORCE(cs, LC(var), LC(var), LC(r.var));
        return r;
    }
    void assert_nonzero(CS& cs) const {
        Fr inv = Fr::zero();
        if (cs.has_witness() && !value.invert(inv)) throw SynthesisError("DivisionByZero");
        Var v = cs.alloc(inv);
        MASP_ENFORCE(cs, LC(var), LC(v), LC(ONE));
    }
    // (c, d) = condition ? (b, a) : (a, b)
    static std::pair<AllocatedNum, AllocatedNum> conditionally_reverse(CS& cs, const AllocatedNum& a, const AllocatedNum& b,
                                                                       const Boolean& cond) {
        AllocatedNum c = alloc(cs, cond.value() ? b.value : a.value);
        MASP_ENFORCE(cs, LC(a.var).sub(b.var), cond.lc(Fr::one()), LC(a.var).sub(c.var));
        AllocatedNum d = alloc(cs, cond.value() ? a.value : b.value);
        MASP_ENFORCE(cs, LC(b.var).sub(a.var), cond.lc(Fr::one()), LC(b.var).sub(d.var));
        return {c, d};
    }
    // 255 plain bits + one unpacking constraint (no range check)
    std::vector<Boolean> to_bits_le(CS& cs) const {
        uint8_t le[32];
        value.to_bytes(le);
        std::vector<Boolean> bits = bits_into_boolean_vec_le(cs, le, 255);
        if (cs.recording()) {
            LC lc;
            Fr coeff = Fr::one();
            for (auto& b : bits) {
                lc.add(b.bit.var, coeff);
                coeff = coeff.dbl();
            }
            lc.sub(var);
            cs.enforce(LC(), LC(), lc);
        } else {
            cs.count_constraint();
        }
        return bits;
    }
    // bits proven to be the canonical representation (<= r - 1): SURVEY.md Appendix B `to_bits_le_strict`
    std::vector<Boolean> to_bits_le_strict(CS& cs) const {
        uint8_t a_le[32], b_le[32];
        value.to_bytes(a_le);
        Fr::one().neg().to_bytes(b_le);  // r - 1
        std::vector<AllocatedBit> result;  // big-endian order
        bool have_last = false;
        AllocatedBit last_run{0, false};
        std::vector<AllocatedBit> current_run;
        bool found_one = false;
        for (int i = 255; i >= 0; --i) {
            bool b = (b_le[i / 8] >> (i % 8)) & 1;
            bool a_bit = (a_le[i / 8] >> (i % 8)) & 1;
            found_one |= b;
            if (!found_one) continue;
            if (b) {
                AllocatedBit ab = AllocatedBit::alloc(cs, a_bit);
                current_run.push_back(ab);
                result.push_back(ab);
            } else {
                if (!current_run.empty()) {
                    if (have_last) current_run.push_back(last_run);
                    AllocatedBit cur = current_run[0];
                    for (size_t k = 1; k < current_run.size(); ++k) cur = AllocatedBit::and_(cs, cur, current_run[k]);
                    last_run = cur;
                    have_last = true;
                    current_run.clear();
                }
                result.push_back(AllocatedBit::alloc_conditionally(cs, a_bit, last_run));
            }
        }
        if (cs.recording()) {
            LC lc;
            Fr coeff = Fr::one();
            for (size_t k = result.size(); k-- > 0;) {
                lc.add(result[k].var, coeff);
                coeff = coeff.dbl();
            }
            lc.sub(var);
            cs.enforce(LC(), LC(), lc);
        } else {
            cs.count_constraint();
        }
        std::vector<Boolean> out;
        for (size_t k = result.size(); k-- > 0;) out.push_back(Boolean::from(result[k]));
        return out;
    }
};

struct Num {
    Fr value;
    LC lc_;
    static Num zero() { return {Fr::zero(), LC()}; }
    static Num from(const AllocatedNum& a) { return {a.value, LC(a.var)}; }
    Num add_bool_with_coeff(const Boolean& bit, const Fr& coeff) const {
        Num r = *this;
        if (bit.value()) r.value = r.value + coeff;
        r.lc_.add(bit.lc(coeff));
        return r;
    }
    LC lc(const Fr& coeff) const { return lc_.scaled(coeff); }
};

// =================================================================================================== lookup
inline void synth(int window, const std::vector<Fr>& constants, std::vector<Fr>& assignment) {
    assignment.assign((size_t)1 << window, Fr::zero());
    for (size_t i = 0; i < constants.size(); ++i) {
        Fr cur = constants[i] - assignment[i];
        assignment[i] = cur;
        for (size_t j = i + 1; j < assignment.size(); ++j)
            if ((j & i) == i) assignment[j] = assignment[j] + cur;
    }
}
typedef std::pair<Fr, Fr> Coord;
// 3-bit window table lookup of an (x, y) pair
inline std::pair<AllocatedNum, AllocatedNum> lookup3_xy(CS& cs, const Boolean bits[3], const std::vector<Coord>& coords) {
    int i = (bits[0].value() ? 1 : 0) + (bits[1].value() ? 2 : 0) + (bits[2].value() ? 4 : 0);
    AllocatedNum rx = AllocatedNum::alloc(cs, coords[i].first);
    AllocatedNum ry = AllocatedNum::alloc(cs, coords[i].second);
    Boolean precomp = Boolean::and_(cs, bits[1], bits[2]);
    if (!cs.recording()) {  // proving mode: the table coefficients and the two rows are only needed for the structure
        cs.count_constraint();
        cs.count_constraint();
        return {rx, ry};
    }
    std::vector<Fr> xs, ys, xc, yc;
    for (auto& c : coords) {
        xs.push_back(c.first);
        ys.push_back(c.second);
    }
    synth(3, xs, xc);
    synth(3, ys, yc);
    const std::vector<Fr>* co[2] = {&xc, &yc};
    const AllocatedNum* res[2] = {&rx, &ry};
    for (int k = 0; k < 2; ++k) {
        const std::vector<Fr>& c = *co[k];
        LC a;
        a.add(ONE, c[1]).add(bits[1].lc(c[3])).add(bits[2].lc(c[5])).add(precomp.lc(c[7]));
        LC cc(res[k]->var);
        cc.add(ONE, c[0].neg()).sub(bits[1].lc(c[2])).sub(bits[2].lc(c[4])).sub(precomp.lc(c[6]));
        MASP_ENFORCE(cs, a, bits[0].lc(Fr::one()), cc);
    }
    return {rx, ry};
}
// 2-bit lookup of (x, y) with y conditionally negated by the third bit
inline std::pair<Num, Num> lookup3_xy_with_conditional_negation(CS& cs, const Boolean bits[3], const std::vector<Coord>& coords) {
    int i = (bits[0].value() ? 1 : 0) + (bits[1].value() ? 2 : 0);
    Fr yv = coords[i].second;
    if (bits[2].value()) yv = yv.neg();
    AllocatedNum y = AllocatedNum::alloc(cs, yv);
    Boolean precomp = Boolean::and_(cs, bits[0], bits[1]);
    if (!cs.recording()) {  // proving mode: x is the table entry itself (the linear combination below evaluates to it)
        cs.count_constraint();
        return {Num{coords[i].first, LC()}, Num::from(y)};
    }
    std::vector<Fr> xs, ys, xc, yc;
    for (auto& c : coords) {
        xs.push_back(c.first);
        ys.push_back(c.second);
    }
    synth(2, xs, xc);
    synth(2, ys, yc);
    Num x = Num::zero()
                .add_bool_with_coeff(Boolean::constant(true), xc[0])
                .add_bool_with_coeff(bits[0], xc[1])
                .add_bool_with_coeff(bits[1], xc[2])
                .add_bool_with_coeff(precomp, xc[3]);
    LC ylc = precomp.lc(yc[3]);
    ylc.add(bits[1].lc(yc[2])).add(bits[0].lc(yc[1])).add(ONE, yc[0]);
    LC a = ylc;
    a.add(ylc);
    LC c = ylc;
    c.sub(y.var);
    MASP_ENFORCE(cs, a, bits[2].lc(Fr::one()), c);
    return {x, Num::from(y)};
}

// =================================================================================================== uint32 / multieq
struct MultiEq {
    CS& cs;
    int bits_used = 0;
    LC lhs, rhs;
    explicit MultiEq(CS& c) : cs(c) {}
    void accumulate() {
        MASP_ENFORCE(cs, lhs, LC(ONE), rhs);
        lhs = LC();
        rhs = LC();
        bits_used = 0;
    }
    // proving mode: only the packing of the equalities into constraints is replayed (their count), no combination is built
    void count_equal(int num_bits) {
        if (254 <= bits_used + num_bits) accumulate();
        bits_used += num_bits;
    }
    void enforce_equal(int num_bits, const LC& l, const LC& r) {
        if (254 <= bits_used + num_bits) accumulate();  // Scalar::CAPACITY
        Fr coeff = Fr::one();
        for (int i = 0; i < bits_used; ++i) coeff = coeff.dbl();
        lhs.add(l.scaled(coeff));
        rhs.add(r.scaled(coeff));
        bits_used += num_bits;
    }
    void finish() {
        if (bits_used > 0) accumulate();
    }
};

struct UInt32 {
    std::array<Boolean, 32> bits;  // LSB first (a fixed array: the BLAKE2s gadget builds ~2 000 of these per compression)
    uint32_t value;
    static UInt32 constant(uint32_t v) {
        UInt32 r;
        r.value = v;
        for (int i = 0; i < 32; ++i) r.bits[i] = Boolean::constant((v >> i) & 1);
        return r;
    }
    static UInt32 from_bits(const std::vector<Boolean>& b) {
        UInt32 r;
        r.value = 0;
        for (int i = 0; i < 32; ++i) {
            r.bits[i] = b[i];
            if (b[i].value()) r.value |= 1u << i;
        }
        return r;
    }
    UInt32 rotr(int by) const {
        UInt32 r;
        for (int i = 0; i < 32; ++i) r.bits[i] = bits[(i + by) % 32];
        r.value = (value >> by) | (value << (32 - by));
        return r;
    }
    UInt32 xor_(CS& cs, const UInt32& o) const {
        UInt32 r;
        r.value = value ^ o.value;
        for (int i = 0; i < 32; ++i) r.bits[i] = Boolean::xor_(cs, bits[i], o.bits[i]);
        return r;
    }
    static UInt32 addmany(MultiEq& me, std::initializer_list<const UInt32*> ops) {
        uint64_t max_value = (uint64_t)ops.size() * 0xffffffffull;
        uint64_t result_value = 0;
        const bool record = me.cs.recording();
        LC lc;
        bool all_constants = true;
        for (const UInt32* op : ops) {
            result_value += op->value;
            if (record) {
                Fr coeff = Fr::one();
                for (auto& bit : op->bits) {
                    lc.add(bit.lc(coeff));
                    coeff = coeff.dbl();
                }
            }
            for (auto& bit : op->bits) all_constants &= bit.is_constant();
        }
        uint32_t modular = (uint32_t)result_value;
        if (all_constants) return constant(modular);
        UInt32 r;
        r.value = modular;
        LC result_lc;
        Fr coeff = Fr::one();
        int i = 0;
        while (max_value != 0) {   // (the bits above 31 are allocated too: they carry the overflow of the sum)
            AllocatedBit b = AllocatedBit::alloc(me.cs, (result_value >> i) & 1);
            if (record) {
                result_lc.add(b.var, coeff);
                coeff = coeff.dbl();
            }
            if (i < 32) r.bits[i] = Boolean::from(b);
            max_value >>= 1;
            ++i;
        }
        if (record)
            me.enforce_equal(i, lc, result_lc);
        else
            me.count_equal(i);
        return r;
    }
};

// =================================================================================================== blake2s gadget
inline void blake2s_mixing_g(MultiEq& me, std::vector<UInt32>& v, int a, int b, int c, int d, const UInt32& x, const UInt32& y) {
    CS& cs = me.cs;
    v[a] = UInt32::addmany(me, {&v[a], &v[b], &x});
    v[d] = v[d].xor_(cs, v[a]).rotr(16);
    v[c] = UInt32::addmany(me, {&v[c], &v[d]});
    v[b] = v[b].xor_(cs, v[c]).rotr(12);
    v[a] = UInt32::addmany(me, {&v[a], &v[b], &y});
    v[d] = v[d].xor_(cs, v[a]).rotr(8);
    v[c] = UInt32::addmany(me, {&v[c], &v[d]});
    v[b] = v[b].xor_(cs, v[c]).rotr(7);
}
inline void blake2s_compression(CS& cs, std::vector<UInt32>& h, const std::vector<UInt32>& m, uint64_t t, bool f) {
    static const uint32_t IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
    static const uint8_t S[10][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
                                      {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
                                      {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
                                      {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
                                      {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    std::vector<UInt32> v(h);
    for (int i = 0; i < 8; ++i) v.push_back(UInt32::constant(IV[i]));
    v[12] = v[12].xor_(cs, UInt32::constant((uint32_t)t));
    v[13] = v[13].xor_(cs, UInt32::constant((uint32_t)(t >> 32)));
    if (f) v[14] = v[14].xor_(cs, UInt32::constant(0xffffffffu));
    {
        MultiEq me(cs);
        for (int i = 0; i < 10; ++i) {
            const uint8_t* s = S[i];
            blake2s_mixing_g(me, v, 0, 4, 8, 12, m[s[0]], m[s[1]]);
            blake2s_mixing_g(me, v, 1, 5, 9, 13, m[s[2]], m[s[3]]);
            blake2s_mixing_g(me, v, 2, 6, 10, 14, m[s[4]], m[s[5]]);
            blake2s_mixing_g(me, v, 3, 7, 11, 15, m[s[6]], m[s[7]]);
            blake2s_mixing_g(me, v, 0, 5, 10, 15, m[s[8]], m[s[9]]);
            blake2s_mixing_g(me, v, 1, 6, 11, 12, m[s[10]], m[s[11]]);
            blake2s_mixing_g(me, v, 2, 7, 8, 13, m[s[12]], m[s[13]]);
            blake2s_mixing_g(me, v, 3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        me.finish();
    }
    for (int i = 0; i < 8; ++i) {
        h[i] = h[i].xor_(cs, v[i]);
        h[i] = h[i].xor_(cs, v[i + 8]);
    }
}
inline std::vector<Boolean> blake2s_gadget(CS& cs, const std::vector<Boolean>& input, const char* personal8) {
    auto le32 = [](const char* p) { return (uint32_t)(uint8_t)p[0] | ((uint32_t)(uint8_t)p[1] << 8) | ((uint32_t)(uint8_t)p[2] << 16) | ((uint32_t)(uint8_t)p[3] << 24); };
    std::vector<UInt32> h = {UInt32::constant(0x6A09E667 ^ 0x01010000 ^ 32), UInt32::constant(0xBB67AE85), UInt32::constant(0x3C6EF372),
                             UInt32::constant(0xA54FF53A), UInt32::constant(0x510E527F), UInt32::constant(0x9B05688C),
                             UInt32::constant(0x1F83D9AB ^ le32(personal8)), UInt32::constant(0x5BE0CD19 ^ le32(personal8 + 4))};
    std::vector<std::vector<UInt32>> blocks;
    for (size_t off = 0; off < input.size(); off += 512) {
        std::vector<UInt32> blk;
        size_t end = std::min(input.size(), off + 512);
        for (size_t w = off; w < end; w += 32) {
            std::vector<Boolean> word(input.begin() + w, input.begin() + std::min(end, w + 32));
            while (word.size() < 32) word.push_back(Boolean::constant(false));
            blk.push_back(UInt32::from_bits(word));
        }
        while (blk.size() < 16) blk.push_back(UInt32::constant(0));
        blocks.push_back(blk);
    }
    if (blocks.empty()) blocks.push_back(std::vector<UInt32>(16, UInt32::constant(0)));
    for (size_t i = 0; i + 1 < blocks.size(); ++i) blake2s_compression(cs, h, blocks[i], (uint64_t)(i + 1) * 64, false);
    blake2s_compression(cs, h, blocks.back(), input.size() / 8, true);
    std::vector<Boolean> out;
    for (auto& w : h) out.insert(out.end(), w.bits.begin(), w.bits.end());
    return out;
}

// =================================================================================================== multipack
inline void pack_into_inputs(CS& cs, const std::vector<Boolean>& bits) {
    for (size_t off = 0; off < bits.size(); off += 254) {
        Num num = Num::zero();
        Fr coeff = Fr::one();
        for (size_t k = off; k < std::min(bits.size(), off + 254); ++k) {
            num = num.add_bool_with_coeff(bits[k], coeff);
            coeff = coeff.dbl();
        }
        Var in = cs.alloc_input(num.value);
        MASP_ENFORCE(cs, num.lc(Fr::one()), LC(ONE), LC(in));
    }
}

}  // namespace masp_host

// libmasp_host: C ABI over the host-side witness generator (circuits.h).  This is the part of
// `SaplingProvingContext::{spend_proof, output_proof, convert_proof}` that precedes `create_random_proof`
// (/root/reference/masp_proofs/src/sapling/prover.rs:51-113, :163-198, :214-248): native key / commitment /
// nullifier derivation, then synthesis of the circuit into (input_assignment, aux_assignment).
// It also exports the static R1CS of each circuit (what bellperson's KeypairAssembly collects) and the
// native primitives, so that tests can pin them against the reference's vectors.
#include <memory>

#include "../../../include/masp_host.h"   // the declarations of everything defined here: a mismatch does not compile
#include "circuits.h"
#include "groth16_vk.h"
#include "pairing_prog.h"
#include "pairing.h"

using namespace masp_host;



namespace {
struct CircuitHandle {
    std::unique_ptr<CS> cs;
};
Var remap(Var v, uint32_t n_inputs) { return (v & AUX) ? n_inputs + (v & ~AUX) : v; }

// aux_montgomery: the aux assignment leaves as Montgomery residues (four little-endian u64 limbs: blst_fr's memory, what
// masp_hip_job::aux_form = MASP_HIP_AUX_MONTGOMERY announces) — no conversion of ~30 000 non-boolean elements per Spend
void write_assignment(const CS& cs, uint8_t* inputs, uint8_t* aux, bool aux_montgomery = false) {
    for (size_t i = 0; i < cs.num_inputs(); ++i) cs.inputs()[i].to_bytes(inputs + 32 * i);
    if (aux_montgomery) {
        static_assert(sizeof(Fr) == 32, "Fr is four 64-bit Montgomery limbs");
        if (!cs.aux_in_place()) memcpy(aux, cs.aux(), 32 * cs.num_aux());   // (in place: the constraint system wrote into `aux` itself)
        return;
    }
    // 70 % of a MASP witness is 0 or 1 (booleans): no Montgomery conversion needed for those
    const Fr one = Fr::one();
    for (size_t j = 0; j < cs.num_aux(); ++j) {
        const Fr& v = cs.aux()[j];
        uint8_t* o = aux + 32 * j;
        if (v.is_zero()) {
            memset(o, 0, 32);
        } else if (v == one) {
            memset(o, 0, 32);
            o[0] = 1;
        } else {
            v.to_bytes(o);
        }
    }
}
// jubjub::Fr is a canonical value by construction in the reference; here scalars arrive as raw bytes.  The circuits witness
// only their low 252 bits while the native cv / rk / cm use all 256, so an unreduced scalar would give a silently
// invalid proof: reject anything >= the Jubjub subgroup order (0x0e7db4ea6533afa906673b0101343b00a6682093ccc81082d0970e5ed6f72cb7).
bool jubjub_scalar_canonical(const uint8_t s[32]) {
    static const uint8_t ORDER_LE[32] = {0xb7, 0x2c, 0xf7, 0xd6, 0x5e, 0x0e, 0x97, 0xd0, 0x82, 0x10, 0xc8, 0xcc, 0x93, 0x20, 0x68, 0xa6,
                                         0x00, 0x3b, 0x34, 0x01, 0x01, 0x3b, 0x67, 0x06, 0xa9, 0xaf, 0x33, 0x65, 0xea, 0xb4, 0x7d, 0x0e};
    for (int i = 31; i >= 0; --i) {
        if (s[i] < ORDER_LE[i]) return true;
        if (s[i] > ORDER_LE[i]) return false;
    }
    return false;  // equal to the order
}
// auxiliary variables of circuit `kind` (0 spend, 1 output, 2 convert): the size of the caller's aux buffer
size_t aux_count(int kind) {
    static const size_t n[3] = {[] {
                                    CS cs(false, false);
                                    SpendW w{};
                                    w.vc.asset_generator = w.ak = w.g_d = w.pk_d = JPoint::identity();
                                    synthesize_spend(cs, w);
                                    return cs.num_aux();
                                }(),
                                [] {
                                    CS cs(false, false);
                                    OutputW w{};
                                    w.vc.asset_generator = w.g_d = w.pk_d = JPoint::identity();
                                    synthesize_output(cs, w);
                                    return cs.num_aux();
                                }(),
                                [] {
                                    CS cs(false, false);
                                    ConvertW w{};
                                    w.vc.asset_generator = JPoint::identity();
                                    synthesize_convert(cs, w);
                                    return cs.num_aux();
                                }()};
    return n[kind];
}
// the constraint system of one witness: straight into the caller's aux buffer when that is to hold Montgomery residues (check & 2)
// and nothing is recorded
CS* new_witness_cs(int kind, int check, uint8_t* aux) {
    if ((check & 2) && !(check & 1) && ((uintptr_t)aux % alignof(Fr)) == 0) return new CS(false, true, reinterpret_cast<Fr*>(aux), aux_count(kind));
    return new CS((check & 1) != 0, true);
}
bool load_path(MerklePathW& p, const uint8_t* siblings, uint64_t position) {
    for (int i = 0; i < TREE_DEPTH; ++i) {
        Fr s;
        if (!Fr::from_bytes(s, siblings + 32 * i)) return false;
        p.auth_path.push_back({s, (bool)((position >> i) & 1)});
    }
    return true;
}
}  // namespace

#ifdef MASP_HOST_TIMING
#include <chrono>
#define TIMING_T0 auto _t = std::chrono::steady_clock::now(); static double _acc[8]; static const char* _nm[8]; static int _calls; int _k = 0;
#define TIMING_MARK(NAME) { auto _n = std::chrono::steady_clock::now(); int _i = 0; for (; _i < 8 && _nm[_i] && strcmp(_nm[_i], NAME); ++_i); _nm[_i] = NAME; _acc[_i] += std::chrono::duration<double, std::milli>(_n - _t).count(); _t = _n; }
#else
#define TIMING_T0
#define TIMING_MARK(NAME)
#endif

extern "C" {


// ---- static circuits: kind 0 spend, 1 output, 2 convert --------------------------------------------
void* masp_host_circuit_setup(int kind) {
    if (!Fr::one_constant_ok()) return nullptr;
    try {
        std::unique_ptr<CircuitHandle> h(new CircuitHandle);
        h->cs.reset(new CS(true, false));
        if (kind == 0) {
            SpendW w{};
            w.vc.asset_generator = JPoint::identity();
            w.ak = w.g_d = w.pk_d = JPoint::identity();
            synthesize_spend(*h->cs, w);
        } else if (kind == 1) {
            OutputW w{};
            w.vc.asset_generator = JPoint::identity();
            w.g_d = w.pk_d = JPoint::identity();
            synthesize_output(*h->cs, w);
        } else if (kind == 2) {
            ConvertW w{};
            w.vc.asset_generator = JPoint::identity();
            synthesize_convert(*h->cs, w);
        } else {
            return nullptr;
        }
        return h.release();
    } catch (...) {
        return nullptr;
    }
}
void masp_host_circuit_free(void* h) { delete (CircuitHandle*)h; }
// out: n_inputs, n_aux, n_constraints, nnz(A), nnz(B), nnz(C)
void masp_host_circuit_counts(void* hh, uint32_t* out) {
    CS& cs = *((CircuitHandle*)hh)->cs;
    out[0] = cs.num_inputs();
    out[1] = cs.num_aux();
    out[2] = cs.num_constraints();
    for (int i = 0; i < 3; ++i) out[3 + i] = cs.matrix(i).col.size();
}
// columns: input i -> i, aux j -> n_inputs + j (the masp_hip_r1cs convention); coef 32 B LE canonical
void masp_host_circuit_matrix(void* hh, int mi, uint32_t* rowptr, uint32_t* col, uint8_t* coef) {
    CS& cs = *((CircuitHandle*)hh)->cs;
    const CS::Matrix& M = cs.matrix(mi);
    memcpy(rowptr, M.rowptr.data(), 4 * M.rowptr.size());
    for (size_t t = 0; t < M.col.size(); ++t) {
        col[t] = remap(M.col[t], (uint32_t)cs.num_inputs());
        M.coef[t].to_bytes(coef + 32 * t);
    }
}
void masp_host_circuit_hash(void* hh, char* out65) {
    std::string s = ((CircuitHandle*)hh)->cs->hash();
    memcpy(out65, s.c_str(), 65);
}

// ---- witness generation -----------------------------------------------------------------------------
// check & 1: additionally record the constraints and fail with MASP_HOST_E_UNSATISFIED if any is violated.
// check & 2: write the aux assignment as Montgomery residues (see write_assignment) instead of canonical bytes.
// rcm is the note commitment randomness `note.rcm()` (sapling.rs:856-863) as 32 bytes LE.
int masp_host_spend_assignment(const uint8_t ak[32], const uint8_t nsk[32], const uint8_t diversifier[11], const uint8_t rcm[32],
                               const uint8_t ar[32], const uint8_t asset_identifier[32], uint64_t value, const uint8_t anchor[32],
                               const uint8_t* path_siblings /*32 x 32*/, uint64_t position, const uint8_t rcv[32], int check,
                               uint8_t* inputs /*8 x 32*/, uint8_t* aux /*100497 x 32*/, uint8_t cv_out[32], uint8_t rk_out[32],
                               uint8_t nf_out[32]) {
    try {
        SpendW w;
        if (!jubjub_scalar_canonical(nsk) || !jubjub_scalar_canonical(rcm) || !jubjub_scalar_canonical(ar) || !jubjub_scalar_canonical(rcv))
            return MASP_HOST_E_INVALID;
        if (!asset_generator(w.vc.asset_generator, asset_identifier)) return MASP_HOST_E_INVALID;
        w.vc.value = value;
        memcpy(w.vc.randomness, rcv, 32);
        if (!JPoint::from_bytes(w.ak, ak)) return MASP_HOST_E_INVALID;
        memcpy(w.nsk, nsk, 32);
        memcpy(w.rcm, rcm, 32);
        memcpy(w.ar, ar, 32);
        if (!Fr::from_bytes(w.anchor, anchor) || !load_path(w.path, path_siblings, position)) return MASP_HOST_E_INVALID;
        // viewing key, payment address (prover.rs:78-84)
        JPoint nk = generators().proof_generation_key.mul(nsk);
        uint8_t ivk[32];
        crh_ivk(ivk, w.ak, nk);
        if (!group_hash(w.g_d, diversifier, 11, "MASP__gd")) return MASP_HOST_E_DIVERSIFIER;
        w.pk_d = w.g_d.mul(ivk);
        // outputs the caller needs next to the proof (prover.rs:87-98,151-156)
        JPoint cv = value_commitment(w.vc.asset_generator, value, rcv);
        JPoint rk = w.ak.add(generators().spending_key.mul(ar));
        JPoint cm = note_commitment(w.vc.asset_generator, value, w.g_d, w.pk_d, rcm);
        cv.to_bytes(cv_out);
        rk.to_bytes(rk_out);
        nullifier(nf_out, cm, position, nk);
        std::unique_ptr<CS> cs_owner(new_witness_cs(0, check, aux));
        CS& cs = *cs_owner;
        synthesize_spend(cs, w);
        if ((check & 1) && cs.first_unsatisfied() >= 0) return MASP_HOST_E_UNSATISFIED;
        write_assignment(cs, inputs, aux, (check & 2) != 0);
        return MASP_HOST_OK;
    } catch (const SynthesisError&) {
        return MASP_HOST_E_SYNTHESIS;
    }
}
int masp_host_output_assignment(const uint8_t esk[32], const uint8_t diversifier[11], const uint8_t pk_d[32], const uint8_t rcm[32],
                                const uint8_t asset_identifier[32], uint64_t value, const uint8_t rcv[32], int check,
                                uint8_t* inputs /*6 x 32*/, uint8_t* aux /*30896 x 32*/, uint8_t cv_out[32]) {
    try {
        OutputW w;
        if (!jubjub_scalar_canonical(esk) || !jubjub_scalar_canonical(rcm) || !jubjub_scalar_canonical(rcv)) return MASP_HOST_E_INVALID;
        if (!asset_generator(w.vc.asset_generator, asset_identifier)) return MASP_HOST_E_INVALID;
        w.vc.value = value;
        memcpy(w.vc.randomness, rcv, 32);
        memcpy(w.asset_identifier, asset_identifier, 32);
        if (!group_hash(w.g_d, diversifier, 11, "MASP__gd")) return MASP_HOST_E_DIVERSIFIER;
        if (!JPoint::from_bytes(w.pk_d, pk_d)) return MASP_HOST_E_INVALID;
        memcpy(w.rcm, rcm, 32);
        memcpy(w.esk, esk, 32);
        value_commitment(w.vc.asset_generator, value, rcv).to_bytes(cv_out);
        std::unique_ptr<CS> cs_owner(new_witness_cs(1, check, aux));
        CS& cs = *cs_owner;
        synthesize_output(cs, w);
        if ((check & 1) && cs.first_unsatisfied() >= 0) return MASP_HOST_E_UNSATISFIED;
        write_assignment(cs, inputs, aux, (check & 2) != 0);
        return MASP_HOST_OK;
    } catch (const SynthesisError&) {
        return MASP_HOST_E_SYNTHESIS;
    }
}
// generator: the AllowedConversion's asset generator point (masp_primitives/src/convert.rs:23-29), 32 bytes
int masp_host_convert_assignment(const uint8_t generator[32], uint64_t value, const uint8_t anchor[32], const uint8_t* path_siblings,
                                 uint64_t position, const uint8_t rcv[32], int check, uint8_t* inputs /*4 x 32*/,
                                 uint8_t* aux /*47322 x 32*/, uint8_t cv_out[32]) {
    try {
        ConvertW w;
        if (!jubjub_scalar_canonical(rcv)) return MASP_HOST_E_INVALID;
        if (!JPoint::from_bytes(w.vc.asset_generator, generator)) return MASP_HOST_E_INVALID;
        w.vc.value = value;
        memcpy(w.vc.randomness, rcv, 32);
        if (!Fr::from_bytes(w.anchor, anchor) || !load_path(w.path, path_siblings, position)) return MASP_HOST_E_INVALID;
        value_commitment(w.vc.asset_generator, value, rcv).to_bytes(cv_out);
        std::unique_ptr<CS> cs_owner(new_witness_cs(2, check, aux));
        CS& cs = *cs_owner;
        synthesize_convert(cs, w);
        if ((check & 1) && cs.first_unsatisfied() >= 0) return MASP_HOST_E_UNSATISFIED;
        write_assignment(cs, inputs, aux, (check & 2) != 0);
        return MASP_HOST_OK;
    } catch (const SynthesisError&) {
        return MASP_HOST_E_SYNTHESIS;
    }
}

// n field elements as Montgomery residues (four little-endian u64 limbs: what `check & 2` makes the synthesizers write) ->
// 32-byte little-endian canonical values.  For callers that need both forms of an assignment (a checker next to the prover).
void masp_host_fr_from_montgomery(const uint8_t* in, uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        Fr v;
        memcpy(v.l, in + 32 * i, 32);
        v.to_bytes(out + 32 * i);
    }
}

// ---- several witnesses per call: the Merkle blocks in lockstep (circuits.h merkle_block_batch) -------------------------
// One job = the arguments of masp_host_spend_assignment / masp_host_convert_assignment plus its return code.  `check` as there
// (check & 1 records and verifies the constraints: that goes witness by witness through the generic gadgets).  Returns the number
// of jobs whose rc is not MASP_HOST_OK.  A caller gives each of its threads a group of ~16 jobs.
// (masp_host_spend_job / masp_host_convert_job: include/masp_host.h)
int masp_host_spend_assignments(size_t n, masp_host_spend_job* jobs, int check) {
    if (check & 1) {
        int bad = 0;
        for (size_t j = 0; j < n; ++j) {
            masp_host_spend_job& J = jobs[j];
            J.rc = masp_host_spend_assignment(J.ak, J.nsk, J.diversifier, J.rcm, J.ar, J.asset_identifier, J.value, J.anchor, J.path_siblings, J.position,
                                              J.rcv, check, J.inputs, J.aux, J.cv_out, J.rk_out, J.nf_out);
            bad += J.rc != MASP_HOST_OK;
        }
        return bad;
    }
    std::vector<SpendW> w(n);
    std::vector<std::unique_ptr<CS>> cs(n);
    std::vector<SpendState> st(n);
    std::vector<size_t> live;
    TIMING_T0
    for (size_t j = 0; j < n; ++j) {
        masp_host_spend_job& J = jobs[j];
        J.rc = MASP_HOST_OK;
        try {
            SpendW& W = w[j];
            if (!jubjub_scalar_canonical(J.nsk) || !jubjub_scalar_canonical(J.rcm) || !jubjub_scalar_canonical(J.ar) || !jubjub_scalar_canonical(J.rcv) ||
                !asset_generator(W.vc.asset_generator, J.asset_identifier) || !JPoint::from_bytes(W.ak, J.ak)) {
                J.rc = MASP_HOST_E_INVALID;
                continue;
            }
            W.vc.value = J.value;
            memcpy(W.vc.randomness, J.rcv, 32);
            memcpy(W.nsk, J.nsk, 32);
            memcpy(W.rcm, J.rcm, 32);
            memcpy(W.ar, J.ar, 32);
            if (!Fr::from_bytes(W.anchor, J.anchor) || !load_path(W.path, J.path_siblings, J.position)) {
                J.rc = MASP_HOST_E_INVALID;
                continue;
            }
            JPoint nk = generators().proof_generation_key.mul(J.nsk);
            uint8_t ivk[32];
            crh_ivk(ivk, W.ak, nk);
            if (!group_hash(W.g_d, J.diversifier, 11, "MASP__gd")) {
                J.rc = MASP_HOST_E_DIVERSIFIER;
                continue;
            }
            W.pk_d = W.g_d.mul(ivk);
            JPoint cv = value_commitment(W.vc.asset_generator, J.value, J.rcv);
            JPoint rk = W.ak.add(generators().spending_key.mul(J.ar));
            JPoint cm = note_commitment(W.vc.asset_generator, J.value, W.g_d, W.pk_d, J.rcm);
            cv.to_bytes(J.cv_out);
            rk.to_bytes(J.rk_out);
            nullifier(J.nf_out, cm, J.position, nk);
            cs[j].reset(new_witness_cs(0, check, J.aux));
            synthesize_spend_pre(*cs[j], W, st[j]);
            live.push_back(j);
        } catch (const SynthesisError&) {
            J.rc = MASP_HOST_E_SYNTHESIS;
        }
    }
    TIMING_MARK("pre")
    // the Merkle blocks of all live witnesses side by side
    {
        std::vector<CS*> c;
        std::vector<AllocatedNum> cur;
        std::vector<const MerklePathW*> paths;
        std::vector<std::vector<Boolean>*> pos;
        std::vector<size_t> mark;
        for (size_t j : live) {
            c.push_back(cs[j].get());
            cur.push_back(st[j].cur);
            paths.push_back(&w[j].path);
            pos.push_back(&st[j].position_bits);
            mark.push_back(cs[j]->num_aux());
        }
        bool fast = false;
        try {
            fast = !live.empty() && merkle_block_batch(live.size(), c.data(), cur.data(), paths.data(), pos.data());
        } catch (const std::exception&) {
            fast = false;
        }
        for (size_t i = 0; i < live.size(); ++i) {
            const size_t j = live[i];
            if (fast) {
                st[j].cur = cur[i];
                continue;
            }
            try {  // a vanished denominator somewhere in the batch: every witness through the gadgets, which report it as the reference does
                cs[j]->truncate_aux(mark[i]);
                st[j].position_bits.clear();
                st[j].cur = merkle_ascend(*cs[j], st[j].cur, w[j].path, &st[j].position_bits);
            } catch (const SynthesisError&) {
                jobs[j].rc = MASP_HOST_E_SYNTHESIS;
            }
        }
    }
    TIMING_MARK("merkle")
    int bad = 0;
    for (size_t j = 0; j < n; ++j) {
        masp_host_spend_job& J = jobs[j];
        if (J.rc == MASP_HOST_OK) {
            try {
                synthesize_spend_post(*cs[j], w[j], st[j]);
                TIMING_MARK("post")
                write_assignment(*cs[j], J.inputs, J.aux, (check & 2) != 0);
                TIMING_MARK("write")
            } catch (const SynthesisError&) {
                J.rc = MASP_HOST_E_SYNTHESIS;
            }
        }
        bad += J.rc != MASP_HOST_OK;
    }
#ifdef MASP_HOST_TIMING
    if (++_calls % 4 == 0) { for (int i = 0; i < 8 && _nm[i]; ++i) fprintf(stderr, "%s %.3f ms/inst  ", _nm[i], _acc[i] / (_calls * n)); fprintf(stderr, "\n"); }
#endif
    return bad;
}
int masp_host_convert_assignments(size_t n, masp_host_convert_job* jobs, int check) {
    if (check & 1) {
        int bad = 0;
        for (size_t j = 0; j < n; ++j) {
            masp_host_convert_job& J = jobs[j];
            J.rc = masp_host_convert_assignment(J.generator, J.value, J.anchor, J.path_siblings, J.position, J.rcv, check, J.inputs, J.aux, J.cv_out);
            bad += J.rc != MASP_HOST_OK;
        }
        return bad;
    }
    std::vector<ConvertW> w(n);
    std::vector<std::unique_ptr<CS>> cs(n);
    std::vector<ConvertState> st(n);
    std::vector<size_t> live;
    for (size_t j = 0; j < n; ++j) {
        masp_host_convert_job& J = jobs[j];
        J.rc = MASP_HOST_OK;
        try {
            ConvertW& W = w[j];
            if (!jubjub_scalar_canonical(J.rcv) || !JPoint::from_bytes(W.vc.asset_generator, J.generator)) {
                J.rc = MASP_HOST_E_INVALID;
                continue;
            }
            W.vc.value = J.value;
            memcpy(W.vc.randomness, J.rcv, 32);
            if (!Fr::from_bytes(W.anchor, J.anchor) || !load_path(W.path, J.path_siblings, J.position)) {
                J.rc = MASP_HOST_E_INVALID;
                continue;
            }
            value_commitment(W.vc.asset_generator, J.value, J.rcv).to_bytes(J.cv_out);
            cs[j].reset(new_witness_cs(2, check, J.aux));
            synthesize_convert_pre(*cs[j], W, st[j]);
            live.push_back(j);
        } catch (const SynthesisError&) {
            J.rc = MASP_HOST_E_SYNTHESIS;
        }
    }
    {
        std::vector<CS*> c;
        std::vector<AllocatedNum> cur;
        std::vector<const MerklePathW*> paths;
        std::vector<std::vector<Boolean>*> pos;
        std::vector<size_t> mark;
        for (size_t j : live) {
            c.push_back(cs[j].get());
            cur.push_back(st[j].cur);
            paths.push_back(&w[j].path);
            pos.push_back(nullptr);
            mark.push_back(cs[j]->num_aux());
        }
        bool fast = false;
        try {
            fast = !live.empty() && merkle_block_batch(live.size(), c.data(), cur.data(), paths.data(), pos.data());
        } catch (const std::exception&) {
            fast = false;
        }
        for (size_t i = 0; i < live.size(); ++i) {
            const size_t j = live[i];
            if (fast) {
                st[j].cur = cur[i];
                continue;
            }
            try {
                cs[j]->truncate_aux(mark[i]);
                st[j].cur = merkle_ascend(*cs[j], st[j].cur, w[j].path, nullptr);
            } catch (const SynthesisError&) {
                jobs[j].rc = MASP_HOST_E_SYNTHESIS;
            }
        }
    }
    int bad = 0;
    for (size_t j = 0; j < n; ++j) {
        masp_host_convert_job& J = jobs[j];
        if (J.rc == MASP_HOST_OK) {
            try {
                synthesize_convert_post(*cs[j], w[j], st[j]);
                write_assignment(*cs[j], J.inputs, J.aux, (check & 2) != 0);
            } catch (const SynthesisError&) {
                J.rc = MASP_HOST_E_SYNTHESIS;
            }
        }
        bad += J.rc != MASP_HOST_OK;
    }
    return bad;
}

// ---- Groth16 self-verification (sapling/prover.rs:148,266) ---------------------------------------------
// params: the Parameters bytes (only the verifying-key prefix is read).  Returns a handle or NULL.
void* masp_host_vk_prepare(const uint8_t* params, size_t len) {
    std::unique_ptr<PreparedVk> vk(new PreparedVk);
    if (!prepare_vk(*vk, params, len)) return nullptr;
    return vk.release();
}
void masp_host_vk_free(void* h) { delete (PreparedVk*)h; }
// public_inputs: n_public x 32 B LE, excluding ONE.  1 = valid, 0 = invalid, < 0 = malformed
int masp_host_vk_verify(const void* h, const uint8_t proof[192], const uint8_t* public_inputs, uint32_t n_public) {
    const PreparedVk& vk = *(const PreparedVk*)h;
    if ((size_t)n_public + 1 != vk.ic.size()) return -1;
    bls::G1A a, c;
    bls::G2A b;
    if (!bls::g1_compressed(a, proof) || !bls::g2_compressed(b, proof + 48) || !bls::g1_compressed(c, proof + 144)) return -2;
    if (a.inf || b.inf || c.inf) return -2;  // bellman's Proof::read refuses the identity in any of the three ("point at infinity")
    bls::G1J acc = bls::G1J::from(vk.ic[0]);
    for (uint32_t i = 0; i < n_public; ++i) {
        Fr chk;
        if (!Fr::from_bytes(chk, public_inputs + 32 * i)) return -3;
        acc = acc.add(vk.ic_mul(i + 1, public_inputs + 32 * i));
    }
    // e(A, B) == e(alpha, beta) e(acc, gamma) e(C, delta)   <=>   ML(A,B) ML(-acc,gamma) ML(-C,delta) / ML(alpha,beta) -> 1
    // (conj = inverse once final-exponentiated)
    bls::G1A nacc = acc.affine(), nc = c;
    nacc.y = nacc.y.neg();
    nc.y = nc.y.neg();
    std::vector<bls::MillerPair> pairs;
    if (!a.inf && !b.inf) pairs.emplace_back(a, b);
    if (!nacc.inf) pairs.emplace_back(nacc, vk.gamma);
    if (!nc.inf) pairs.emplace_back(nc, vk.delta);
    bls::Fp12 f = bls::multi_miller(pairs) * vk.alpha_beta.conj();
    return bls::final_exp(f) == bls::Fp12::one() ? 1 : 0;
}
// Batched form (bellman `verify_proofs_batch`, reached from /root/reference/masp_proofs/src/sapling/verifier/batch.rs:24-31):
// with random z_i,  prod_i e(z_i A_i, B_i) == e(alpha,beta)^(sum z_i) e(sum_i z_i acc_i, gamma) e(sum_i z_i C_i, delta).
// proofs: n x 192 B; public_inputs: n x n_public x 32 B; z: n x 16 B of caller-supplied randomness (128-bit coefficients).
// 1 = all valid, 0 = at least one invalid (the caller re-checks one by one to find it), < 0 = malformed.
int masp_host_vk_verify_batch(const void* h, size_t n, const uint8_t* proofs, const uint8_t* public_inputs, uint32_t n_public,
                              const uint8_t* z) {
    const PreparedVk& vk = *(const PreparedVk*)h;
    if ((size_t)n_public + 1 != vk.ic.size()) return -1;
    if (n == 0) return 1;
    std::vector<bls::MillerPair> pairs;
    pairs.reserve(n);
    bls::G1J csum = bls::G1J::inf();
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* pr = proofs + 192 * i;
        bls::G1A a, c;
        bls::G2A b;
        if (!bls::g1_compressed(a, pr) || !bls::g2_compressed(b, pr + 48) || !bls::g1_compressed(c, pr + 144)) return -2;
        if (a.inf || b.inf || c.inf) return -2;  // (as above: Proof::read refuses the identity)
        uint8_t zi[32] = {0};
        memcpy(zi, z + 16 * i, 16);
        zi[0] |= 1;  // never zero
        bls::G1A za = bls::G1J::from(a).mul_le(zi, 128).affine();
        csum = csum.add(bls::G1J::from(c).mul_le(zi, 128));
        if (!za.inf && !b.inf) pairs.emplace_back(za, b);
    }
    return batch_verify_finish(vk, n, public_inputs, n_public, z, bls::multi_miller(pairs), csum.affine());
}

// The Miller loop of the GPU batch verifier exists as levelled straight-line programs (host/pairing_prog.h).  This runs
// them on the host interpreter for one pair (P: 96 B uncompressed G1, Q: 192 B uncompressed G2) and compares the result,
// coefficient for coefficient, with this library's own Miller loop (pairing.h) and with the templated tower instantiated
// over the concrete field.  0 = equal.  stats (may be NULL): ops / steps / steps with products / products / slots of
// the doubling program, then the same five of the addition program, then of the Fp12 product.
int masp_host_pairing_program_selftest(const uint8_t* p96, const uint8_t* q192, uint32_t* stats) {
    bls::G1A P;
    bls::G2A Q;
    if (!bls::g1_uncompressed(P, p96) || !bls::g2_uncompressed(Q, q192) || P.inf || Q.inf) return -1;
    const prog::PairingPrograms& pp = prog::pairing_programs();
    if (stats) {
        const prog::Program* ps[3] = {&pp.dbl, &pp.add, &pp.mul12};
        for (int i = 0; i < 3; ++i) {
            stats[5 * i] = (uint32_t)ps[i]->ops.size();
            stats[5 * i + 1] = (uint32_t)ps[i]->step_start.size() - 1;
            stats[5 * i + 2] = ps[i]->n_mul_steps;
            stats[5 * i + 3] = ps[i]->n_mul;
            stats[5 * i + 4] = ps[i]->n_slots;
        }
    }
    const bls::Fp12 want = bls::miller(P, Q);
    if (!(prog::miller_by_program(P, Q) == want)) return 1;
    // the templated formulas over the concrete field, step by step
    prog::Fp12T<bls::Fp> f = {{{bls::Fp::one(), bls::Fp::zero()}, {bls::Fp::zero(), bls::Fp::zero()}, {bls::Fp::zero(), bls::Fp::zero()}},
                              {{bls::Fp::zero(), bls::Fp::zero()}, {bls::Fp::zero(), bls::Fp::zero()}, {bls::Fp::zero(), bls::Fp::zero()}}};
    prog::MillerT<bls::Fp> m;
    m.xp = P.x; m.yp = P.y;
    m.xq = {Q.x.a, Q.x.b}; m.yq = {Q.y.a, Q.y.b};
    m.X = m.xq; m.Y = m.yq; m.Z = {bls::Fp::one(), bls::Fp::zero()};
    const uint64_t xabs = 0xd201000000010000ull;
    for (int b = 62; b >= 0; --b) {
        f = f.sq();
        m.dbl_step(f);
        if ((xabs >> b) & 1) m.add_step(f);
    }
    bls::Fp12 g = {{{f.a.a.a, f.a.a.b}, {f.a.b.a, f.a.b.b}, {f.a.c.a, f.a.c.b}}, {{f.b.a.a, f.b.a.b}, {f.b.b.a, f.b.b.b}, {f.b.c.a, f.b.c.b}}};
    if (!(g.conj() == want)) return 2;
    // Fp12 product program
    std::vector<bls::Fp> sl(pp.n_slots, bls::Fp::zero());
    const bls::Fp12 y = want.sq();
    auto flat = [](const bls::Fp12& x, bls::Fp* o) {
        const bls::Fp v[12] = {x.a.a.a, x.a.a.b, x.a.b.a, x.a.b.b, x.a.c.a, x.a.c.b, x.b.a.a, x.b.a.b, x.b.b.a, x.b.b.b, x.b.c.a, x.b.c.b};
        for (int i = 0; i < 12; ++i) o[i] = v[i];
    };
    bls::Fp pf[12];
    flat(want, &sl[prog::SLOT_F]);
    flat(y, &sl[prog::SLOT_Y]);
    prog::run_program(pp.mul12, sl.data());
    flat(want * y, pf);
    for (int i = 0; i < 12; ++i)
        if (!(sl[prog::SLOT_F + i] == pf[i])) return 3;
    return 0;
}

// ---- native primitives (pinned by the reference's vectors in tests/) ---------------------------------
// which: 0 proof_generation_key 1 note_commitment_randomness 2 nullifier_position 3 value_commitment_randomness
//        4 spending_key 5..10 pedersen[0..5]; out: u | v as 2 x 32 B LE
void masp_host_generator(int which, uint8_t out64[64]) {
    const Generators& g = generators();
    const JPoint* p[11] = {&g.proof_generation_key, &g.note_commitment_randomness, &g.nullifier_position, &g.value_commitment_randomness,
                           &g.spending_key, &g.pedersen[0], &g.pedersen[1], &g.pedersen[2], &g.pedersen[3], &g.pedersen[4], &g.pedersen[5]};
    JAffine a = p[which]->to_affine();
    a.u.to_bytes(out64);
    a.v.to_bytes(out64 + 32);
}
// personalization: -1 NoteCommitment, otherwise MerkleTree(depth); bits: one byte per bit; out: u | v
void masp_host_pedersen_hash(int personalization, const uint8_t* bits, size_t nbits, uint8_t out64[64]) {
    std::vector<bool> b(nbits);
    for (size_t i = 0; i < nbits; ++i) b[i] = bits[i] != 0;
    Personalization p{personalization < 0, personalization < 0 ? 0u : (unsigned)personalization};
    JAffine a = pedersen_hash(p, b).to_affine();
    a.u.to_bytes(out64);
    a.v.to_bytes(out64 + 32);
}
int masp_host_asset_identifier(const uint8_t* name, size_t len, uint8_t out32[32]) { return asset_identifier(out32, name, len) ? 0 : 1; }
int masp_host_asset_generator(const uint8_t id[32], uint8_t out32[32]) {
    JPoint p;
    if (!asset_generator(p, id)) return 1;
    p.to_bytes(out32);
    return 0;
}
int masp_host_value_commitment(const uint8_t id[32], uint64_t value, const uint8_t rcv[32], uint8_t out32[32], uint8_t uv64[64]) {
    JPoint g;
    if (!asset_generator(g, id)) return 1;
    JPoint cv = value_commitment(g, value, rcv);
    cv.to_bytes(out32);
    if (uv64) {
        JAffine a = cv.to_affine();
        a.u.to_bytes(uv64);
        a.v.to_bytes(uv64 + 32);
    }
    return 0;
}
// note commitment u-coordinate (cmu) from (asset id, value, diversifier, pk_d, rcm)
int masp_host_note_cmu(const uint8_t id[32], uint64_t value, const uint8_t diversifier[11], const uint8_t pk_d[32], const uint8_t rcm[32],
                       uint8_t cmu32[32]) {
    JPoint g, gd, pk;
    if (!asset_generator(g, id) || !group_hash(gd, diversifier, 11, "MASP__gd") || !JPoint::from_bytes(pk, pk_d)) return 1;
    note_commitment(g, value, gd, pk, rcm).to_affine().u.to_bytes(cmu32);
    return 0;
}
int masp_host_merkle_hash(unsigned depth, const uint8_t lhs[32], const uint8_t rhs[32], uint8_t out32[32]) {
    Fr l, r;
    if (!Fr::from_bytes(l, lhs) || !Fr::from_bytes(r, rhs)) return 1;
    merkle_hash(depth, l, r).to_bytes(out32);
    return 0;
}
// [k]P for P given as 32 bytes; (point decode / scalar multiplication / encode round trip)
int masp_host_jubjub_mul(const uint8_t p32[32], const uint8_t k32[32], uint8_t out32[32]) {
    JPoint p;
    if (!JPoint::from_bytes(p, p32)) return 1;
    p.mul(k32).to_bytes(out32);
    return 0;
}
// affine coordinates (u | v, 2 x 32 B LE) of an encoded point
int masp_host_point_uv(const uint8_t p32[32], uint8_t out64[64]) {
    JPoint p;
    if (!JPoint::from_bytes(p, p32)) return 1;
    JAffine a = p.to_affine();
    a.u.to_bytes(out64);
    a.v.to_bytes(out64 + 32);
    return 0;
}
// p + q, or p - q when subtract != 0 (value-commitment bookkeeping of the proving context)
int masp_host_jubjub_add(const uint8_t p32[32], const uint8_t q32[32], int subtract, uint8_t out32[32]) {
    JPoint p, q;
    if (!JPoint::from_bytes(p, p32) || !JPoint::from_bytes(q, q32)) return 1;
    p.add(subtract ? q.neg() : q).to_bytes(out32);
    return 0;
}
// acc + sum_i (+/-) points[i]  (n compressed points; subtract[i] != 0: that point is subtracted; subtract may be NULL): the value
// commitments of a chunk of descriptions in one call (SaplingProvingContext::cv_sum, sapling/prover.rs:154,205,272)
int masp_host_jubjub_sum(const uint8_t acc32[32], const uint8_t* points, size_t n, const uint8_t* subtract, uint8_t out32[32]) {
    JPoint acc;
    if (!JPoint::from_bytes(acc, acc32)) return 1;
    for (size_t i = 0; i < n; ++i) {
        JPoint q;
        if (!JPoint::from_bytes(q, points + 32 * i)) return 1;
        acc = acc.add(subtract && subtract[i] ? q.neg() : q);
    }
    acc.to_bytes(out32);
    return 0;
}
// leaf of the commitment tree for a spendable note: derives nk, ivk, g_d, pk_d exactly as spend_proof does and
// returns cmu (what the wallet already knows as the note commitment); also pk_d for callers that want the address
int masp_host_spend_leaf(const uint8_t ak[32], const uint8_t nsk[32], const uint8_t diversifier[11], const uint8_t rcm[32],
                         const uint8_t id[32], uint64_t value, uint8_t cmu32[32], uint8_t pk_d32[32]) {
    JPoint akp, g, gd;
    if (!JPoint::from_bytes(akp, ak) || !asset_generator(g, id)) return MASP_HOST_E_INVALID;
    if (!group_hash(gd, diversifier, 11, "MASP__gd")) return MASP_HOST_E_DIVERSIFIER;
    JPoint nk = generators().proof_generation_key.mul(nsk);
    uint8_t ivk[32];
    crh_ivk(ivk, akp, nk);
    JPoint pk = gd.mul(ivk);
    note_commitment(g, value, gd, pk, rcm).to_affine().u.to_bytes(cmu32);
    if (pk_d32) pk.to_bytes(pk_d32);
    return MASP_HOST_OK;
}
// AllowedConversion::from(I128Sum) (masp_primitives/src/convert.rs:86-118): generator = sum_i sign(v_i) * [|v_i| as u64] G_i with
// G_i the asset generator of identifier i, cofactor NOT cleared.  values: n x 16 bytes, little-endian two's-complement
// i128.  `abs as u64` keeps the low 64 bits, as the reference's cast does; i128::MIN has no absolute value (the reference
// panics "invalid conversion"): MASP_HOST_E_INVALID.
int masp_host_allowed_conversion(size_t n, const uint8_t* identifiers /* n x 32 */, const uint8_t* values /* n x 16 */, uint8_t generator_out[32]) {
    JPoint acc = JPoint::identity();
    for (size_t i = 0; i < n; ++i) {
        JPoint g;
        if (!asset_generator(g, identifiers + 32 * i)) return MASP_HOST_E_INVALID;
        const uint8_t* v = values + 16 * i;
        uint64_t lo = 0, hi = 0;
        for (int b = 0; b < 8; ++b) {
            lo |= (uint64_t)v[b] << (8 * b);
            hi |= (uint64_t)v[8 + b] << (8 * b);
        }
        const bool negative = (hi >> 63) != 0;
        if (negative) {
            if (hi == (1ull << 63) && lo == 0) return MASP_HOST_E_INVALID;  // i128::MIN
            lo = ~lo + 1;                                                     // low 64 bits of -value
        }
        uint8_t k[32] = {0};
        for (int b = 0; b < 8; ++b) k[b] = (uint8_t)(lo >> (8 * b));
        JPoint t = g.mul(k);
        acc = acc.add(negative ? t.neg() : t);
    }
    acc.to_bytes(generator_out);
    return MASP_HOST_OK;
}
// leaf of the convert tree: u of PedersenHash(NoteCommitment, repr(generator))  (convert.rs:39-64)
int masp_host_convert_cmu(const uint8_t generator[32], uint8_t out32[32]) {
    JPoint g;
    if (!JPoint::from_bytes(g, generator)) return 1;
    uint8_t b[32];
    g.to_bytes(b);
    pedersen_hash({true, 0}, bytes_to_bits_le(b, 32)).to_affine().u.to_bytes(out32);
    return 0;
}
}  // extern "C"

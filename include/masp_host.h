/* masp_host.h — C ABI of libmasp_host.so: the host side of the proving path that PRECEDES the Groth16 prover.
 *
 * What `SaplingProvingContext::{spend_proof, output_proof, convert_proof}` do before they call `create_random_proof`
 * (/root/reference/masp_proofs/src/sapling/prover.rs:51-113, :163-198, :214-248): native key / commitment / nullifier derivation
 * and `Circuit::synthesize` (/root/reference/masp_proofs/src/circuit/sapling.rs:139-596, circuit/convert.rs:29-128) into
 * (input_assignment, aux_assignment) — plus the static R1CS of each circuit (what bellperson's KeypairAssembly collects), the
 * self-verification of sapling/prover.rs:148,266 (`verify_proof` with the PreparedVerifyingKey of lib.rs:391-393) and the native
 * primitives the reference's vectors pin (tests/golden/).  A Rust caller keeps its own synthesis and needs none
 * of this; a C / C++ caller feeds masp_hip_prove_batch (include/masp_hip.h) from here.  No GPU code: plain C++ behind C linkage.
 *
 * Conventions as in masp_hip.h: integer return codes, caller-owned buffers, no exceptions across the boundary.  Field elements are
 * 32-byte little-endian canonical values unless a parameter says Montgomery; Jubjub points are their 32-byte `to_bytes()` encoding.
 * Thread-safe: every function may be called from any number of threads (handles are immutable once built).
 * The definitions in masp_amd/csrc/host/host_api.cpp include this header, so the compiler checks every signature;
 * tests/test_capi_symbols.py checks the header against `nm -D`. */
#ifndef MASP_HOST_H
#define MASP_HOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MASP_HOST_OK 0
#define MASP_HOST_E_INVALID 1       /* an argument is not a canonical scalar / a point on the curve / a field element */
#define MASP_HOST_E_DIVERSIFIER 2   /* the diversifier has no group hash: the reference's Err(()) at sapling/prover.rs:84 */
#define MASP_HOST_E_SYNTHESIS 3     /* bellperson's SynthesisError */
#define MASP_HOST_E_UNSATISFIED 4   /* check & 1: a constraint does not hold */

/* `kind` everywhere: 0 Spend, 1 Output, 2 Convert (= MASP_HIP_SPEND / _OUTPUT / _CONVERT) */

/* ---- the static R1CS of a circuit (-> masp_hip_r1cs) ---- */
void* masp_host_circuit_setup(int kind);                 /* handle, or NULL */
void masp_host_circuit_free(void* h);
/* out[6]: n_inputs, n_aux, n_constraints, nnz(A), nnz(B), nnz(C) */
void masp_host_circuit_counts(void* h, uint32_t* out);
/* matrix mi (0 A, 1 B, 2 C) as CSR: rowptr[n_constraints + 1], col[nnz] (input i -> i, aux j -> n_inputs + j), coef[nnz][32] */
void masp_host_circuit_matrix(void* h, int mi, uint32_t* rowptr, uint32_t* col, uint8_t* coef);
/* bellperson's TestConstraintSystem::hash() of the circuit as 64 hex digits + NUL (pinned: circuit/sapling.rs:733,1026, convert.rs:221) */
void masp_host_circuit_hash(void* h, char* out65);

/* ---- witness generation.  check & 1: also record the constraints and fail with MASP_HOST_E_UNSATISFIED if one is violated;
 * check & 2: write the aux assignment as Montgomery residues (four little-endian u64 limbs: masp_hip_job::aux_form =
 * MASP_HIP_AUX_MONTGOMERY), straight into `aux` when it is 8-byte aligned.  rcm = note.rcm() as 32 bytes. ---- */
int masp_host_spend_assignment(const uint8_t ak[32], const uint8_t nsk[32], const uint8_t diversifier[11], const uint8_t rcm[32],
                               const uint8_t ar[32], const uint8_t asset_identifier[32], uint64_t value, const uint8_t anchor[32],
                               const uint8_t* path_siblings /* 32 x 32, leaf level first */, uint64_t position, const uint8_t rcv[32], int check,
                               uint8_t* inputs /* 8 x 32 */, uint8_t* aux /* 100497 x 32 */, uint8_t cv_out[32], uint8_t rk_out[32],
                               uint8_t nf_out[32]);
int masp_host_output_assignment(const uint8_t esk[32], const uint8_t diversifier[11], const uint8_t pk_d[32], const uint8_t rcm[32],
                                const uint8_t asset_identifier[32], uint64_t value, const uint8_t rcv[32], int check,
                                uint8_t* inputs /* 6 x 32 */, uint8_t* aux /* 30896 x 32 */, uint8_t cv_out[32]);
/* generator: the AllowedConversion's asset generator (masp_primitives/src/convert.rs:23-29) */
int masp_host_convert_assignment(const uint8_t generator[32], uint64_t value, const uint8_t anchor[32], const uint8_t* path_siblings,
                                 uint64_t position, const uint8_t rcv[32], int check, uint8_t* inputs /* 4 x 32 */,
                                 uint8_t* aux /* 47322 x 32 */, uint8_t cv_out[32]);
/* n Montgomery residues (as `check & 2` writes them) -> canonical bytes */
void masp_host_fr_from_montgomery(const uint8_t* in, uint8_t* out, size_t n);

/* Several witnesses per call: their Merkle blocks are synthesised in lockstep (one shared inversion per window across the group:
 * ~2x the witnesses per second and thread).  A job = the arguments of the single-witness call + its return code; the call returns
 * the number of jobs whose rc is not MASP_HOST_OK.  A caller gives each of its threads a group of ~16 jobs. */
typedef struct masp_host_spend_job {
    const uint8_t *ak, *nsk, *diversifier, *rcm, *ar, *asset_identifier;
    uint64_t value;
    const uint8_t *anchor, *path_siblings;
    uint64_t position;
    const uint8_t* rcv;
    uint8_t *inputs, *aux, *cv_out, *rk_out, *nf_out;
    int rc;
} masp_host_spend_job;
typedef struct masp_host_convert_job {
    const uint8_t* generator;
    uint64_t value;
    const uint8_t *anchor, *path_siblings;
    uint64_t position;
    const uint8_t* rcv;
    uint8_t *inputs, *aux, *cv_out;
    int rc;
} masp_host_convert_job;
int masp_host_spend_assignments(size_t n, masp_host_spend_job* jobs, int check);
int masp_host_convert_assignments(size_t n, masp_host_convert_job* jobs, int check);

/* ---- Groth16 verification on the host (sapling/prover.rs:148,266; batched: sapling/verifier/batch.rs:24-31) ---- */
void* masp_host_vk_prepare(const uint8_t* params, size_t len);   /* Parameters bytes: only the verifying-key prefix is read */
void masp_host_vk_free(void* h);
/* public_inputs: n_public x 32, excluding ONE.  1 valid, 0 invalid, < 0 malformed */
int masp_host_vk_verify(const void* h, const uint8_t proof[192], const uint8_t* public_inputs, uint32_t n_public);
/* n proofs, one random linear combination with the caller's z (n x 16 bytes).  1 all valid, 0 at least one is not, < 0 malformed */
int masp_host_vk_verify_batch(const void* h, size_t n, const uint8_t* proofs, const uint8_t* public_inputs, uint32_t n_public,
                              const uint8_t* z);
/* the GPU verifier's Miller-loop programs on a host interpreter against this library's own Miller loop, for one pair (P 96 B, Q 192 B
 * uncompressed); 0 = equal.  stats (may be NULL): 15 words, see host_api.cpp */
int masp_host_pairing_program_selftest(const uint8_t* p96, const uint8_t* q192, uint32_t* stats);

/* ---- native primitives (pinned by the reference's vectors: tests/golden/) ---- */
/* which: 0 proof_generation_key, 1 note_commitment_randomness, 2 nullifier_position, 3 value_commitment_randomness, 4 spending_key,
 * 5..10 pedersen[0..5]; out: u | v */
void masp_host_generator(int which, uint8_t out64[64]);
/* personalization: -1 NoteCommitment, else MerkleTree(depth); bits: one byte per bit; out: u | v */
void masp_host_pedersen_hash(int personalization, const uint8_t* bits, size_t nbits, uint8_t out64[64]);
int masp_host_asset_identifier(const uint8_t* name, size_t len, uint8_t out32[32]);
int masp_host_asset_generator(const uint8_t id[32], uint8_t out32[32]);
int masp_host_value_commitment(const uint8_t id[32], uint64_t value, const uint8_t rcv[32], uint8_t out32[32], uint8_t uv64[64]);
int masp_host_note_cmu(const uint8_t id[32], uint64_t value, const uint8_t diversifier[11], const uint8_t pk_d[32], const uint8_t rcm[32],
                       uint8_t cmu32[32]);
int masp_host_merkle_hash(unsigned depth, const uint8_t lhs[32], const uint8_t rhs[32], uint8_t out32[32]);
int masp_host_jubjub_mul(const uint8_t p32[32], const uint8_t k32[32], uint8_t out32[32]);
int masp_host_point_uv(const uint8_t p32[32], uint8_t out64[64]);
int masp_host_jubjub_add(const uint8_t p32[32], const uint8_t q32[32], int subtract, uint8_t out32[32]);
/* acc + sum_i (+/-) points[i]: the value commitments of a chunk of descriptions (SaplingProvingContext::cv_sum); subtract may be NULL */
int masp_host_jubjub_sum(const uint8_t acc32[32], const uint8_t* points, size_t n, const uint8_t* subtract, uint8_t out32[32]);
/* leaf of the commitment tree for a spendable note (cmu), and pk_d if wanted (may be NULL) */
int masp_host_spend_leaf(const uint8_t ak[32], const uint8_t nsk[32], const uint8_t diversifier[11], const uint8_t rcm[32],
                         const uint8_t id[32], uint64_t value, uint8_t cmu32[32], uint8_t pk_d32[32]);
/* AllowedConversion::from(I128Sum) (masp_primitives/src/convert.rs:86-118): identifiers n x 32, values n x 16 (LE two's-complement i128) */
int masp_host_allowed_conversion(size_t n, const uint8_t* identifiers, const uint8_t* values, uint8_t generator_out[32]);
/* leaf of the convert tree (convert.rs:39-64) */
int masp_host_convert_cmu(const uint8_t generator[32], uint8_t out32[32]);

#ifdef __cplusplus
}
#endif
#endif

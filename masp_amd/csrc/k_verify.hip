// GPU Groth16 batch verification: the C ABI entry points masp_hip_vk_prepare / masp_hip_verify_batch (include/masp_hip.h),
// the kernels of device/pairing.hpp, and the host-side remainder (host/groth16_vk.h: public-input combination, two pairs,
// final exponentiation — work that does not grow with the batch).
// Replaces bellman `verify_proofs_batch` as reached from /root/reference/masp_proofs/src/sapling/verifier/batch.rs:24-31,201-239
// and the per-proof `verify_proof` self-checks of the prover (sapling/prover.rs:148,266) when they are batched.
#include <mutex>

#include "device/pairing.hpp"
#include "host/groth16_vk.h"
#include "host/pairing_prog.h"
#include "internal.h"

using namespace masp;

struct masp_hip_vk {
    masp_host::PreparedVk vk;
    int device = 0;
    DevBuf<uint32_t> ops, steps;  // the three programs, concatenated
    PairingProgramDev dbl{}, add{}, mul12{};
    uint32_t n_slots = 0;
    bool lds_ok = false;  // the interpreter kernels' LDS limit was raised on THIS key's device (the attribute is per device)
    // work buffers, grown on demand and kept (hipFree would synchronise the device under the provers)
    DevBuf<uint8_t> d_proofs, d_z, d_sum;
    DevBuf<G1Affine> d_za;
    DevBuf<G2Affine> d_b;
    DevBuf<G1Xyzz> d_zc;
    DevBuf<int> d_status;
    DevBuf<Fp> d_f;
    hipStream_t stream = nullptr;  // verification runs next to the provers' batches, on one of the context's two verifier streams (not owned:
                                   // masp_hip_ctx::vk_streams — created and kept off the batch streams' hardware queues with the context)
    std::mutex mu;                 // one verification at a time per key
};

// (fail() takes slot_mu for the error text: fine under the shared context lock too)
static int fail_shared_v(masp_hip_ctx* ctx, int rc) { return fail(ctx, rc); }

#define FIRST_DEVICE(ctx) ((ctx) && !(ctx)->children.empty() ? (ctx)->children[0] : (ctx))

extern "C" {

int masp_hip_vk_prepare(masp_hip_ctx* ctx, const uint8_t* params, size_t params_len, masp_hip_vk** out) {
    if (!ctx || !params || !out) return MASP_HIP_E_INVALID_ARG;
    *out = nullptr;
    ctx = FIRST_DEVICE(ctx);
    std::unique_ptr<masp_hip_vk> v(new masp_hip_vk);
    if (!masp_host::prepare_vk(v->vk, params, params_len)) return MASP_HIP_E_PARAMS_FORMAT;
    std::unique_lock<std::shared_mutex> lock(ctx->mu);
    hipSetDevice(ctx->device);
    v->device = ctx->device;
    const masp_host::prog::PairingPrograms& pp = masp_host::prog::pairing_programs();
    const masp_host::prog::Program* ps[3] = {&pp.dbl, &pp.add, &pp.mul12};
    std::vector<uint32_t> ops, steps;
    size_t op_off[3], st_off[3];
    for (int i = 0; i < 3; ++i) {
        op_off[i] = ops.size();
        st_off[i] = steps.size();
        for (uint32_t s : ps[i]->step_start) steps.push_back(s + (uint32_t)op_off[i]);  // absolute indices into the concatenation
        ops.insert(ops.end(), ps[i]->ops.begin(), ps[i]->ops.end());
    }
    int rc;
    hipStream_t s = ctx->main_stream;
    if ((rc = v->ops.upload(ops.data(), ops.size(), s)) || (rc = v->steps.upload(steps.data(), steps.size(), s))) return fail(ctx, rc);
    if (hipStreamSynchronize(s) != hipSuccess) return fail(ctx, MASP_HIP_E_HIP);
    PairingProgramDev* dst[3] = {&v->dbl, &v->add, &v->mul12};
    for (int i = 0; i < 3; ++i) {
        dst[i]->ops = v->ops.p;
        dst[i]->steps = v->steps.p + st_off[i];
        dst[i]->n_steps = (uint32_t)ps[i]->step_start.size() - 1;
    }
    v->n_slots = pp.n_slots;
    v->lds_ok = hipFuncSetAttribute((const void*)k_miller_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess &&
                hipFuncSetAttribute((const void*)k_fp12_product, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess;
    v->stream = ctx->vk_streams[ctx->vk_next.fetch_add(1) % 2];
    *out = v.release();
    return MASP_HIP_OK;
}

void masp_hip_vk_free(masp_hip_vk* vk) {
    if (!vk) return;
    hipSetDevice(vk->device);
    delete vk;
}

int masp_hip_verify_batch(masp_hip_ctx* ctx, masp_hip_vk* vk, size_t n, const uint8_t* proofs, const uint8_t* public_inputs,
                          uint32_t n_public, const uint8_t* z, int* all_valid) {
    if (!ctx || !vk || !all_valid || (n && (!proofs || !z || (n_public && !public_inputs))) || n > (1u << 20)) return MASP_HIP_E_INVALID_ARG;
    if ((size_t)n_public + 1 != vk->vk.ic.size()) return MASP_HIP_E_PARAMS_SHAPE;
    *all_valid = 0;
    if (n == 0) {
        *all_valid = 1;
        return MASP_HIP_OK;
    }
    const ApiLaunchScope api_scope;
    ctx = FIRST_DEVICE(ctx);
    if (ctx->device != vk->device) return MASP_HIP_E_INVALID_ARG;
    std::shared_lock<std::shared_mutex> lock(ctx->mu);  // concurrent with provers (and with verifications under other keys)
    std::lock_guard<std::mutex> vlock(vk->mu);
    hipSetDevice(ctx->device);
    hipStream_t s = vk->stream;
    const uint32_t nn = (uint32_t)n;
    DevBuf<uint8_t>&d_proofs = vk->d_proofs, &d_z = vk->d_z, &d_sum = vk->d_sum;
    DevBuf<G1Affine>& d_za = vk->d_za;
    DevBuf<G2Affine>& d_b = vk->d_b;
    DevBuf<G1Xyzz>& d_zc = vk->d_zc;
    DevBuf<int>& d_status = vk->d_status;
    DevBuf<Fp>& d_f = vk->d_f;
    int rc;
    if ((rc = d_proofs.upload(proofs, 192 * n, s)) || (rc = d_z.upload(z, 16 * n, s)) || (rc = d_za.reserve(n)) || (rc = d_b.reserve(n)) ||
        (rc = d_zc.reserve(n)) || (rc = d_status.reserve(n)) || (rc = d_f.reserve(12 * n)) || (rc = d_sum.reserve(96)))
        return fail_shared_v(ctx, rc);
    HIP_TRY(hipMemsetAsync(d_status.p, 0, sizeof(int) * n, s));
    // the interpreter keeps its slots in LDS: n_slots x 48 bytes per wave
    const uint32_t lds = vk->n_slots * 48;
    if (!vk->lds_ok || lds > 64 * 1024) {
        last_hip_error() = "pairing interpreter: LDS configuration failed";
        return fail_shared_v(ctx, MASP_HIP_E_HIP);
    }
    MASP_LAUNCH(k_verify_prepare, dim3((nn + 63) / 64, 5), dim3(64), 0, s, d_proofs.p, d_z.p, nn, d_za.p, d_b.p, d_zc.p, d_status.p);
    MASP_LAUNCH(k_g1_sum_export, dim3(1), dim3(256), 0, s, d_zc.p, nn, d_sum.p);
    MASP_LAUNCH(k_miller_pairs, dim3(nn), dim3(64), lds, s, vk->dbl, vk->add, vk->n_slots, d_za.p, d_b.p, d_f.p);
    const uint32_t g = std::min<uint32_t>(nn, 64);
    MASP_LAUNCH(k_fp12_product, dim3(g), dim3(64), lds, s, vk->mul12, vk->n_slots, d_f.p, nn, g);
    if (g > 1) MASP_LAUNCH(k_fp12_product, dim3(1), dim3(64), lds, s, vk->mul12, vk->n_slots, d_f.p, g, 1u);
    std::vector<int> status(n);
    masp_host::bls::Fp12 f;
    uint8_t sum96[96];
    static_assert(sizeof(masp_host::bls::Fp12) == 12 * 48, "Fp12 is 12 packed Montgomery residues");
    if (hipMemcpyAsync(status.data(), d_status.p, sizeof(int) * n, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&f, d_f.p, 12 * 48, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(sum96, d_sum.p, 96, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        last_hip_error() = std::string("batch verification failed: ") + hipGetErrorString(hipGetLastError());
        return fail_shared_v(ctx, MASP_HIP_E_HIP);
    }
    if (launch_status() != MASP_HIP_OK) return fail_shared_v(ctx, MASP_HIP_E_HIP);  // a refused launch: the buffers read back mean nothing
    for (int st : status)
        if (st & (PT_BAD_FLAGS | PT_NOT_CANONICAL | PT_NOT_IN_SUBGROUP | PT_INFINITY)) return MASP_HIP_OK;  // what Proof::read refuses (the identity included: "point at infinity"): not valid (*all_valid stays 0)
    masp_host::bls::G1A csum;
    if (!masp_host::bls::g1_uncompressed(csum, sum96)) {
        last_hip_error() = "batch verification: device returned a malformed point";
        return fail_shared_v(ctx, MASP_HIP_E_HIP);
    }
    int v = masp_host::batch_verify_finish(vk->vk, n, public_inputs, n_public, z, f, csum);
    if (v < 0) return MASP_HIP_E_SCALAR_RANGE;
    *all_valid = v;
    return MASP_HIP_OK;
}

}  // extern "C"

// TEST-ONLY shim: compiles masp_amd/csrc/device/{field,curve,io}.hpp for the *host* so the exact
// source the HIP kernels use can be checked on a machine without a GPU (tests/test_device_math_host.py).
// It is never part of the product library.
#include "../../masp_amd/csrc/device/io.hpp"
#include "../../tools/fp28.hpp"   // (an experiment kept with its checks: see the header)
using namespace masp;

// ---- the same field functions ON THE DEVICE (their device overloads are hand-written carry chains / inline asm that the host
// build above never compiles): n lanes, lane i computes op(a_i, b_i).  op: 0 add 1 sub 2 mul 4 neg 5 sqr 8 dbl.  Returns 0, or a
// HIP error code.
template <class C>
__global__ void k_field_ops(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int B = 4 * C::N;
    if (op >= 16) {
        // RAW limbs in, raw limbs out (Fp only): the Montgomery products on operands anywhere in [0, 2p) — what the lazily
        // reduced chains of the bucket tree feed them, and the shapes that stress the carry-free top-limb terms (field.hpp,
        // MASP_MACNC).  16 mul, 17 mul_lazy (left in [0, 2p)), 18 sqr, 19 mul2(a, b, b, a), 20 mul_lazy(mul_lazy(a, b), b)
        if constexpr (C::N == 12) {
            const Fe<C> x = fe_load_le<C>(a + (size_t)B * i), y = fe_load_le<C>(b + (size_t)B * i);
            Fe<C> r;
            switch (op) {
                case 16: r = fe_mul(x, y); break;
                case 17: r = fe_mul_lazy(x, y); break;
                case 18: r = fe_sqr(x); break;
                case 19: r = fe_mul2(x, y, y, x); break;
                default: r = fe_mul_lazy(fe_mul_lazy(x, y), y);
            }
            fe_store_le(r, out + (size_t)B * i);
        }
        return;
    }
    Fe<C> x = fe_to_mont(fe_load_le<C>(a + (size_t)B * i)), y = fe_to_mont(fe_load_le<C>(b + (size_t)B * i)), r;
    switch (op) {
        case 0: r = fe_add(x, y); break;
        case 1: r = fe_sub(x, y); break;
        case 2: r = fe_mul(x, y); break;
        case 4: r = fe_neg(x); break;
        case 8: r = fe_dbl(x); break;
        default: r = fe_sqr(x);
    }
    fe_store_le(fe_from_mont(r), out + (size_t)B * i);
}

// ---- the 28-bit-limb form (device/fp28.hpp): raw limbs in and out, 56 bytes per element (an Fp operand / result uses the first
// 48).  op: 0 mul 1 sqr 2 canon 3 sub_lazy 4 neg 5 from_fp 6 from_fp_lazy 7 to_fp 8 Ops::add 9 Ops::sub 10 Ops::dbl
// 11 canon(sub_lazy(sub_lazy(sqr(a), b), b)) 12 is_zero(a) | eq(a, b) << 1
MASP_HD void fp28_test_op(int op, const uint8_t* pa, const uint8_t* pb, uint8_t* po) {
    F28 a, b, r = fp28_zero();
    Fp fa, fr;
    for (int i = 0; i < 14; ++i) {
        a.v[i] = (uint32_t)pa[4 * i] | (uint32_t)pa[4 * i + 1] << 8 | (uint32_t)pa[4 * i + 2] << 16 | (uint32_t)pa[4 * i + 3] << 24;
        b.v[i] = (uint32_t)pb[4 * i] | (uint32_t)pb[4 * i + 1] << 8 | (uint32_t)pb[4 * i + 2] << 16 | (uint32_t)pb[4 * i + 3] << 24;
    }
    for (int i = 0; i < 12; ++i) fa.v[i] = a.v[i];
    bool is_fp = false;
    switch (op) {
        case 0: r = fp28_mul(a, b); break;
        case 1: r = fp28_sqr(a); break;
        case 2: r = fp28_canon(a); break;
        case 3: r = fp28_sub_lazy(a, b); break;
        case 4: r = fp28_neg(a); break;
        case 5: r = fp28_from_fp(fa); break;
        case 6: r = fp28_from_fp_lazy(fa); break;
        case 7: fr = fp28_to_fp(a); is_fp = true; break;
        case 8: r = Fp28Ops::add(a, b); break;
        case 9: r = Fp28Ops::sub(a, b); break;
        case 10: r = Fp28Ops::dbl(a); break;
        case 11: r = fp28_canon(fp28_sub_lazy(fp28_sub_lazy(fp28_sqr(a), b), b)); break;
        default: r.v[0] = (fp28_is_zero(a) ? 1u : 0u) | (fp28_eq(a, b) ? 2u : 0u); break;
    }
    if (is_fp) {
        for (int i = 0; i < 12; ++i) r.v[i] = fr.v[i];
        r.v[12] = r.v[13] = 0;
    }
    for (int i = 0; i < 14; ++i)
        for (int k = 0; k < 4; ++k) po[4 * i + k] = (uint8_t)(r.v[i] >> (8 * k));
}
__global__ void k_fp28_ops(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fp28_test_op(op, a + 56 * (size_t)i, b + 56 * (size_t)i, out + 56 * (size_t)i);
}

extern "C" {
// op: 0 add 1 sub 2 mul 3 inv 4 neg 5 sqr 6 inv (binary gcd) 7 inv (Fermat); canonical little-endian in/out; which: 0 Fp (48 B), 1 Fr (32 B)
int mh_field_op(int which, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    if (which == 0) {
        Fp x = fe_to_mont(fe_load_le<FpCfg>(a)), y = fe_to_mont(fe_load_le<FpCfg>(b)), r;
        switch (op) {
            case 0: r = fe_add(x, y); break;
            case 1: r = fe_sub(x, y); break;
            case 2: r = fe_mul(x, y); break;
            case 3: r = fe_inv(x); break;
            case 4: r = fe_neg(x); break;
            case 6: r = fe_inv_bingcd(x); break;
            case 7: r = fe_inv_fermat(x); break;
            default: r = fe_sqr(x);
        }
        fe_store_le(fe_from_mont(r), out);
    } else {
        Fr x = fe_to_mont(fe_load_le<FrCfg>(a)), y = fe_to_mont(fe_load_le<FrCfg>(b)), r;
        switch (op) {
            case 0: r = fe_add(x, y); break;
            case 1: r = fe_sub(x, y); break;
            case 2: r = fe_mul(x, y); break;
            case 3: r = fe_inv(x); break;
            case 4: r = fe_neg(x); break;
            case 6: r = fe_inv_bingcd(x); break;
            case 7: r = fe_inv_fermat(x); break;
            default: r = fe_sqr(x);
        }
        fe_store_le(fe_from_mont(r), out);
    }
    return 0;
}
// sum_i [k_i] P_i with the XYZZ formulas; mode 0: scalar-mul each then xyzz_add; mode 1: signed madd chain
// (k_i interpreted as small counts: adds P_i k_i[0] times, negated if k_i[1] != 0)
int mh_g1_lincomb(const uint8_t* pts96, const uint8_t* scalars32, int n, int mode, uint8_t* out96, uint8_t* out48) {
    G1Xyzz acc = xyzz_inf<FpOps>();
    for (int i = 0; i < n; ++i) {
        G1Affine p;
        int st = g1_read_uncompressed(pts96 + 96 * i, p);
        if (st & ~PT_INFINITY) return -1;
        if (mode == 0) {
            Fr k = fe_load_le<FrCfg>(scalars32 + 32 * i);
            G1Xyzz t = xyzz_mul_scalar(xyzz_from_affine(p), k.v);
            xyzz_add(acc, t);
        } else {
            int cnt = scalars32[32 * i];
            bool neg = scalars32[32 * i + 1] != 0;
            for (int c = 0; c < cnt; ++c) xyzz_madd(acc, p, neg);
        }
    }
    G1Affine r = xyzz_to_affine(acc);
    g1_write_uncompressed(r, out96);
    g1_write_compressed(r, out48);
    return 0;
}
int mh_g2_lincomb(const uint8_t* pts192, const uint8_t* scalars32, int n, int mode, uint8_t* out192, uint8_t* out96) {
    G2Xyzz acc = xyzz_inf<Fp2Ops>();
    for (int i = 0; i < n; ++i) {
        G2Affine p;
        int st = g2_read_uncompressed(pts192 + 192 * i, p);
        if (st & ~PT_INFINITY) return -1;
        if (mode == 0) {
            Fr k = fe_load_le<FrCfg>(scalars32 + 32 * i);
            G2Xyzz t = xyzz_mul_scalar(xyzz_from_affine(p), k.v);
            xyzz_add(acc, t);
        } else {
            int cnt = scalars32[32 * i];
            bool neg = scalars32[32 * i + 1] != 0;
            for (int c = 0; c < cnt; ++c) xyzz_madd(acc, p, neg);
        }
    }
    G2Affine r = xyzz_to_affine(acc);
    g2_write_uncompressed(r, out192);
    g2_write_compressed(r, out96);
    return 0;
}

int mh_field_ops_gpu(int which, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, int n) {
    const size_t bytes = (size_t)(which == 0 ? 48 : 32) * n;
    uint8_t *da, *db, *dout;
    hipError_t e;
    if ((e = hipMalloc(&da, bytes)) || (e = hipMalloc(&db, bytes)) || (e = hipMalloc(&dout, bytes))) return (int)e;
    hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
    hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
    if (which == 0)
        hipLaunchKernelGGL((k_field_ops<FpCfg>), dim3((n + 63) / 64), dim3(64), 0, 0, op, da, db, dout, n);
    else
        hipLaunchKernelGGL((k_field_ops<FrCfg>), dim3((n + 63) / 64), dim3(64), 0, 0, op, da, db, dout, n);
    e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    hipFree(da);
    hipFree(db);
    hipFree(dout);
    return (int)e;
}
int mh_fp28_ops(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, int n) {
    for (int i = 0; i < n; ++i) fp28_test_op(op, a + 56 * (size_t)i, b + 56 * (size_t)i, out + 56 * (size_t)i);
    return 0;
}
int mh_fp28_ops_gpu(int op, const uint8_t* a, const uint8_t* b, uint8_t* out, int n) {
    const size_t bytes = (size_t)56 * n;
    uint8_t *da, *db, *dout;
    hipError_t e;
    if ((e = hipMalloc(&da, bytes)) || (e = hipMalloc(&db, bytes)) || (e = hipMalloc(&dout, bytes))) return (int)e;
    hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
    hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fp28_ops, dim3((n + 63) / 64), dim3(64), 0, 0, op, da, db, dout, n);
    e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    hipFree(da);
    hipFree(db);
    hipFree(dout);
    return (int)e;
}
}

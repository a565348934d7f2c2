// TEST-ONLY shim: compiles masp_amd/csrc/device/{field,curve,io}.cuh for the *host* so the exact
// source the HIP kernels use can be checked on a machine without a GPU (tests/test_device_math_host.py).
// It is never part of the product library.
#include "../../masp_amd/csrc/device/io.cuh"
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
        // reduced chains of the bucket tree feed them, and the shapes that stress the carry-free top-limb terms (field.cuh,
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
}

// exp2 for the pair kernels (gram.hip, grad.hip): 2^t for t <= n, evaluated from the CLAMPED quantity
//     w = clamp01(u'^2 + woff),   u' = (x - z) * scale_d / 32,   t = n - 1024 w,
// where log2(base variance) = n - 1024 woff with n = max(ceil(log2 bv), 0) riding in `magic` and woff >= 0
// (= 1.5*2^32 + n/1024).  The clamp is the free VOP3 output modifier of the v_fma_f64 that forms w, so no v_max is
// spent on keeping the exponent in range (t >= n - 1024).
//
//   a  = magic - w              rounds w to a multiple of 2^-20, i.e. t to a multiple of 1/1024; the low mantissa word of
//                               a is the integer k = 1024 n + 1024 * rounded(t) = 1024 e + j (two's complement)
//   Tb[j & 1023]                LDS table, Tb[j] = 4 * 2^(j/1024) with (j << 10) subtracted from its high word, so that
//   hi + (k << 10)              = hi(4 * 2^(j/1024)) + (e << 20): the exponent is patched by ONE v_lshl_add_u32 (no mask,
//                               no arithmetic shift).  The factor 4 keeps the exponent field positive down to e = -1024
//                               and is folded into the polynomial's constant term (0.25).
//   rw = w + (a - magic)        = w - rounded(w), exact, |rw| <= 2^-21; r_t = -1024 rw, |r_t| <= 2^-11
//   p  = 0.25 * 2^(r_t)         degree-3 polynomial in rw, constant term exactly 0.25, the other three coefficients a
//                               Chebyshev-weighted least-squares fit on the interval (max relative error 9.5e-17 = 0.43 ulp in
//                               exact arithmetic; the Taylor coefficients would leave 5.5e-16)
//
// 14 VALU instructions per value including forming u' and w.  r01 used a 512-entry table with a degree-4 Taylor polynomial
// (one FMA more per value, the kernels are issue-bound); the 8 KiB table still leaves two workgroups of the Gram kernel per CU
// at D = 16.  Measured max error <= 2 ulp + the conditioning of the argument (tests/test_gpu_gram.py::test_exp2_accuracy).
#pragma once
#include <hip/hip_runtime.h>

namespace oak {

static __device__ const double g_exp2_table1024[1024] = {
#include "exp2_table1024.inc"
};
constexpr int EW_BITS = 10;
constexpr int EW_N = 1 << EW_BITS;                      // LDS table entries
constexpr double EW_MAGIC = 6442450944.0;               // 1.5 * 2^32: ulp 2^-20 = the rounding step of w
// The 512-entry variant (TB = 9: every other table entry, degree-4 Taylor polynomial, magic 1.5 * 2^33) is kept for the
// shapes where the extra 4 KiB of LDS would cost the Gram kernel a workgroup per CU (D = 32 at 128 columns per workgroup).
template <int TB> constexpr double ew_magic() { return TB == 10 ? 6442450944.0 : 12884901888.0; }

template <int TB = EW_BITS>
__device__ __forceinline__ double biased_table_entry(int j) {
    const double t4 = 4.0 * g_exp2_table1024[j << (EW_BITS - TB)];
    return __hiloint2double(__double2hiint(t4) - (j << (20 - TB)), __double2loint(t4));
}

__device__ __forceinline__ double fma_clamp01(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// NV independent values at once, interleaved so the DP pipe always has independent work
template <int NV, int TB = EW_BITS>
__device__ __forceinline__ void exp2_w_vec(const double (&w)[NV], const double (&magic)[NV], double (&out)[NV],
                                           const double* __restrict__ tab) {
    // TB = 10: 0.25 * 2^(-1024 rw) on |rw| <= 2^-21 (see the header): hexadecimal literals, exactly the fitted doubles.
    // TB = 9: the degree-4 Taylor polynomial on |rw| <= 2^-20.
    constexpr double T1 = 6.931471805599453094e-01, T2 = 2.402265069591007123e-01, T3 = 5.550410866482157995e-02,
                     T4 = 9.618129107628477162e-03, S = -1024.0;
    constexpr double C1 = TB == 10 ? -0x1.62e42fefa39f1p+7 : 0.25 * T1 * S;
    constexpr double C2 = TB == 10 ? 0x1.ebfbe039d53f0p+15 : 0.25 * T2 * S * S;
    constexpr double C3 = TB == 10 ? -0x1.c6b08cf0db226p+23 : 0.25 * T3 * S * S * S;
    constexpr double C4 = 0.25 * T4 * S * S * S * S;
    double a[NV], r[NV], p[NV], tv[NV];
    int ki[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) a[v] = magic[v] - w[v];
#pragma unroll
    for (int v = 0; v < NV; ++v) ki[v] = __double2loint(a[v]);
#pragma unroll
    for (int v = 0; v < NV; ++v) tv[v] = tab[ki[v] & ((1 << TB) - 1)];
#pragma unroll
    for (int v = 0; v < NV; ++v) r[v] = w[v] + (a[v] - magic[v]);
    if constexpr (TB == 10) {
#pragma unroll
        for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(C3, r[v], C2);
    } else {
#pragma unroll
        for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(C4, r[v], C3);
#pragma unroll
        for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], C2);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], C1);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], 0.25);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int hi = __double2hiint(tv[v]) + (ki[v] << (20 - TB));
        out[v] = __hiloint2double(hi, __double2loint(tv[v])) * p[v];
    }
}

}  // namespace oak

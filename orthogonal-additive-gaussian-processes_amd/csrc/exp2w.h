// exp2 for the pair kernels (gram.hip, grad.hip): 2^t for t <= n, evaluated from the CLAMPED quantity
//     w = clamp01(u'^2 + woff),   u' = (x - z) * scale_d / 32,   t = n - 1024 w,
// where log2(base variance) = n - 1024 woff with n = max(ceil(log2 bv), 0) riding in `magic` and woff >= 0
// (= 1.5*2^33 + n/1024).  The clamp is the free VOP3 output modifier of the v_fma_f64 that forms w, so no v_max is
// spent on keeping the exponent in range (t >= n - 1024).
//
//   a  = magic - w              rounds w to a multiple of 2^-19, i.e. t to a multiple of 1/512; the low mantissa word of
//                               a is the integer k = 512 n + 512 * rounded(t) = 512 e + j (two's complement)
//   Tb[j & 511]                 LDS table, Tb[j] = 4 * 2^(j/512) with (j << 11) subtracted from its high word, so that
//   hi + (k << 11)              = hi(4 * 2^(j/512)) + (e << 20): the exponent is patched by ONE v_lshl_add_u32 (no mask,
//                               no arithmetic shift).  The factor 4 keeps the exponent field positive down to e = -1024
//                               and is folded into the polynomial's constant term (0.25).
//   rw = w + (a - magic)        = w - rounded(w), exact, |rw| <= 2^-20; r_t = -1024 rw, |r_t| <= 2^-10
//   p  = 0.25 * 2^(r_t)         degree-4 Taylor polynomial in rw (truncation 1.2e-18 relative)
//
// 15 VALU instructions per value including forming u' and w (was 20 with the v_max / mask / shift form); measured
// max error 1.6 ulp + the conditioning of the argument (tests/test_gpu_gram.py::test_exp2_accuracy).
#pragma once
#include <hip/hip_runtime.h>

namespace oak {

static __device__ const double g_exp2_table512[512] = {
#include "exp2_table512.inc"
};
constexpr int EW_BITS = 9;
constexpr int EW_N = 1 << EW_BITS;                      // LDS table entries
constexpr double EW_MAGIC = 12884901888.0;              // 1.5 * 2^33

__device__ __forceinline__ double biased_table_entry(int j) {
    const double t4 = 4.0 * g_exp2_table512[j];
    return __hiloint2double(__double2hiint(t4) - (j << (20 - EW_BITS)), __double2loint(t4));
}

__device__ __forceinline__ double fma_clamp01(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// NV independent values at once, interleaved so the DP pipe always has independent work
template <int NV>
__device__ __forceinline__ void exp2_w_vec(const double (&w)[NV], const double (&magic)[NV], double (&out)[NV],
                                           const double* __restrict__ tab) {
    constexpr double c1 = 6.931471805599453094e-01, c2 = 2.402265069591007123e-01, c3 = 5.550410866482157995e-02,
                     c4 = 9.618129107628477162e-03;
    constexpr double S = -1024.0;
    constexpr double C1 = 0.25 * c1 * S, C2 = 0.25 * c2 * S * S, C3 = 0.25 * c3 * S * S * S, C4 = 0.25 * c4 * S * S * S * S;
    double a[NV], r[NV], p[NV], tv[NV];
    int ki[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) a[v] = magic[v] - w[v];
#pragma unroll
    for (int v = 0; v < NV; ++v) ki[v] = __double2loint(a[v]);
#pragma unroll
    for (int v = 0; v < NV; ++v) tv[v] = tab[ki[v] & (EW_N - 1)];
#pragma unroll
    for (int v = 0; v < NV; ++v) r[v] = w[v] + (a[v] - magic[v]);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(C4, r[v], C3);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], C2);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], C1);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], 0.25);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int hi = __double2hiint(tv[v]) + (ki[v] << (20 - EW_BITS));
        out[v] = __hiloint2double(hi, __double2loint(tv[v])) * p[v];
    }
}

}  // namespace oak

// What the fp64 MFMA pipe sustains, and at which shader clock: s_memtime (shader cycles) against s_memrealtime (100 MHz),
// for 1 / 2 / 4 / 8 waves per SIMD and 2 / 4 / 8 independent accumulators per wave, plus the fp64 FMA and an idle loop.
//   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ void __launch_bounds__(256) probe(long long* out, double* sink, int iters) {
    const long long c0 = __builtin_readcyclecounter();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    double4_t acc[8];
    double f[8];
    for (int i = 0; i < 8; ++i) { acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0}; f[i] = 1e-3 * (threadIdx.x + i); }
    double a = 1.0 + 1e-9 * threadIdx.x, b = 0.5;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                        // fp64 MFMA, NACC independent accumulators, 16 MFMAs per iteration
#pragma unroll
            for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 1) {                 // fp64 FMA, 8 independent chains
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = __builtin_fma(f[i], a, b);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("s_nop 7");
        }
    }
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
    const long long c1 = __builtin_readcyclecounter();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (s == 12345.678) sink[0] = s;
}

template <int MODE, int NACC>
void run(const char* name, int wg, int iters, double flop_per_iter_per_wave) {
    long long* d; double* sink;
    (void)hipMalloc(&d, sizeof(long long) * 2 * wg); (void)hipMalloc(&sink, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE, NACC><<<wg, 256>>>(d, sink, iters / 10);                   // warm
    (void)hipEventRecord(e0);
    probe<MODE, NACC><<<wg, 256>>>(d, sink, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * wg);
    (void)hipMemcpy(h.data(), d, sizeof(long long) * 2 * wg, hipMemcpyDeviceToHost);
    double rsum = 0, csum = 0;
    for (int i = 0; i < wg; ++i) { rsum += (double)h[2 * i] / (double)h[2 * i + 1]; csum += (double)h[2 * i]; }
    const double tf = flop_per_iter_per_wave * iters * 4.0 * wg / (ms * 1e-3) / 1e12;
    const double waves_per_simd = wg * 4.0 / 1024.0;
    const double cyc_per_op = MODE == 0 ? (csum / wg) / ((double)iters * 16.0 * (waves_per_simd < 1 ? 1 : waves_per_simd)) : 0.0;
    printf("%-10s acc %d  %5d WG (%.0f waves/SIMD): %8.3f ms  %6.1f TFLOP/s  clock %.0f MHz  %.1f shader cycles per MFMA per SIMD\n",
           name, NACC, wg, waves_per_simd, ms, tf, rsum / wg * 100.0, cyc_per_op);
    (void)hipFree(d); (void)hipFree(sink);
}

int main() {
    run<2, 8>("idle", 2048, 20000, 0.0);
    const double mf = 16.0 * 2048.0;
    for (int wg : {256, 512, 1024, 2048}) {
        run<0, 2>("mfma_f64", wg, 20000, mf);
        run<0, 4>("mfma_f64", wg, 20000, mf);
        run<0, 8>("mfma_f64", wg, 20000, mf);
    }
    run<1, 8>("fma_f64", 2048, 40000, 32.0 * 2.0 * 64.0);
    return 0;
}

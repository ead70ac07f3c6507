// fp64 MFMA issue rate against the ORDER in which a wave walks its accumulators: NACC accumulators, each updated CHAIN times in
// a row before the next one (CHAIN = 1: plain rotation, the order a k-step-major GEMM inner loop produces).  Distinct A / B
// registers per MFMA (8 + 8 fragments in rotation), 2 waves per SIMD (512 workgroups of 256) and 1 wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC, int CHAIN>
__global__ void __launch_bounds__(256) probe(double* sink, long long* clk, int iters) {
    const long long c0 = __builtin_readcyclecounter();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    double4_t acc[NACC];
    double a[8], b[8];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + 1e-9 * (threadIdx.x + i); b[i] = 0.5 + 1e-9 * i; }
    for (int it = 0; it < iters; ++it) {
        // one "stage": every accumulator gets 8 updates, CHAIN of them back to back
#pragma unroll
        for (int k0 = 0; k0 < 8; k0 += CHAIN)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
#pragma unroll
                for (int k = k0; k < k0 + CHAIN; ++k)
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(k + i) & 7], b[(k + 3 * i) & 7], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int NACC, int CHAIN>
void run(int wg, int iters) {
    double* sink; (void)hipMalloc(&sink, 8);
    long long* clk; (void)hipMalloc(&clk, 16); long long hc[2];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<NACC, CHAIN><<<wg, 256>>>(sink, clk, iters / 10);
    (void)hipEventRecord(e0);
    probe<NACC, CHAIN><<<wg, 256>>>(sink, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double tf = 2048.0 * 8.0 * NACC * iters * 4.0 * wg / (ms * 1e-3) / 1e12;
    (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("acc %2d  chain %d  %4d WG: %8.3f ms  %6.1f TFLOP/s  clock %.0f MHz\n", NACC, CHAIN, wg, ms, tf, 100.0 * hc[0] / hc[1]);
    (void)hipFree(sink);
}

int main() {
    for (int wg : {512, 256, 1024}) {
        run<16, 1>(wg, 4000); run<16, 8>(wg, 4000); run<14, 1>(wg, 4000); run<12, 1>(wg, 4000); run<10, 1>(wg, 4000);
        run<8, 1>(wg, 8000); run<8, 8>(wg, 8000); run<6, 1>(wg, 8000);
        run<4, 1>(wg, 16000); run<2, 1>(wg, 32000);
    }
    return 0;
}

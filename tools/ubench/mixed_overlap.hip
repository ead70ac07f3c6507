// Microbenchmark 3: do LOW-PRECISION MFMA waves (int8, bf16) overlap with fp64 VALU waves on the same SIMDs?
// (fp64 MFMA does not: fp64_overlap.hip.)  Planning data for an Ozaki-style split of the fp64 SYRK onto int8 MFMA that
// would run under the DP-VALU Gram kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef int int4_t __attribute__((ext_vector_type(4)));
typedef int int16_t_ __attribute__((ext_vector_type(16)));
typedef float float16_t_ __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// kind 0: i8 32x32x32 ; 1: bf16 32x32x16.   mode 0: all 8 waves MFMA ; 1: all 8 waves fp64 FMA ; 2: waves 0-3 MFMA, 4-7 FMA ;
// 3: waves 0-3 MFMA only ; 4: waves 4-7 FMA only
template <int KIND>
__global__ void __launch_bounds__(512) k_roles(double* out, int iters, int mode, double seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_mfma = (mode == 0) || ((mode == 2 || mode == 3) && wave < 4);
    const bool do_fma = (mode == 1) || ((mode == 2 || mode == 4) && wave >= 4);
    double s = 0;
    if (do_mfma) {
        if constexpr (KIND == 0) {
            int16_t_ acc[2];
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
            int4_t a = {(int)threadIdx.x, 3, 5, 7}, b = {1, (int)threadIdx.x, 2, 9};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
        } else {
            float16_t_ acc[2];
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
            bf16x8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * threadIdx.x + j); b[j] = (__bf16)(1.0f - 0.01f * j); }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
        }
    } else if (do_fma) {
        double a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = seed + i * 0.001 + threadIdx.x * 1e-6;
        const double b = 0.999999, c = 1e-7;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], b, c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> float timeit(F f, int reps = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 2, iters = 2000;
    double* out; CK(hipMalloc(&out, sizeof(double) * blocks * 512));
    const char* names[] = {"all 8 waves MFMA", "all 8 waves fp64 FMA", "4 MFMA + 4 fp64-FMA waves", "4 MFMA waves only", "4 fp64-FMA waves only"};
    for (int kind = 0; kind < 2; ++kind) {
        printf("---- %s ----\n", kind == 0 ? "int8 v_mfma_i32_32x32x32_i8" : "bf16 v_mfma_f32_32x32x16_bf16");
        for (int mode = 0; mode < 5; ++mode) {
            float ms = kind == 0 ? timeit([&] { k_roles<0><<<blocks, 512>>>(out, iters, mode, 1.0); })
                                 : timeit([&] { k_roles<1><<<blocks, 512>>>(out, iters, mode, 1.0); });
            double extra = 0;
            if (mode == 0 || mode == 3) {
                const double waves = (mode == 0 ? 8.0 : 4.0) * blocks;
                const double ops = waves * iters * 16.0 * 2.0 * 32 * 32 * (kind == 0 ? 32 : 16);
                extra = ops / ms * 1e-9;
            }
            printf("%-28s %8.3f ms", names[mode], ms);
            if (extra > 0) printf("   %8.1f T%s/s", extra, kind == 0 ? "OP" : "FLOP");
            printf("\n");
        }
    }
    return 0;
}

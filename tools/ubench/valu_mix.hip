// Microbenchmark (r06): what does the fp32 / integer work of the residue conversion (csrc/gram.hip, CRT epilogue) cost next to the
// fp64 VALU stream of the Gram kernel?  (1) rates of v_fma_f32, v_cvt_f32_i32, v_perm_b32, v_bfe_i32 against v_fma_f64;
// (2) do fp32 waves overlap with fp64 waves on the same SIMD; (3) fp32 and fp64 instructions interleaved inside ONE wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// role 0: fp64 FMA chain x8; 1: fp32 FMA chain x8; 2: cvt_f32_i32 + fma ; 3: v_perm_b32 ; 4: bfe_i32; 5: interleaved 8 fp64 + 16 fp32 per unit
__device__ __forceinline__ double run_role(int role, int iters, double seed) {
    double s = 0;
    if (role == 0) {
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
    } else if (role == 1) {
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (float)seed + i * 0.001f + threadIdx.x * 1e-6f;
        const float b = 0.999999f, c = 1e-7f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], b, c);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
    } else if (role == 2) {
        int a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (int)seed + i + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) { float f = (float)a[i]; a[i] = __float_as_int(f) ; }      // cvt (+ a bit move that is free)
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
    } else if (role == 3) {
        unsigned a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (unsigned)seed + i + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_perm(a[i], a[(i + 1) & 7], 0x06020400u);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
    } else if (role == 4) {
        int a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (int)seed + i + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_sbfe(a[i] + 77, 1, 19);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
    } else if (role == 5) {
        double a[8]; float f[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = seed + i * 0.001 + threadIdx.x * 1e-6;
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = (float)seed + i * 0.001f + threadIdx.x * 1e-6f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { a[i] = __builtin_fma(a[i], 0.999999, 1e-7); f[2 * i] = __builtin_fmaf(f[2 * i], 0.999999f, 1e-7f); f[2 * i + 1] = __builtin_fmaf(f[2 * i + 1], 0.999999f, 1e-7f); }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) s += f[i];
    }
    return s;
}

// 512 threads = 8 waves = 2 per SIMD.  roleA for waves 0-3, roleB for waves 4-7 (-1: idle)
__global__ void __launch_bounds__(512) k_mix(double* out, int iters, int roleA, int roleB, double seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave < 4 ? roleA : roleB;
    double s = 0;
    if (role >= 0) s = run_role(role, iters, seed);
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
    const int blocks = p.multiProcessorCount, iters = 2000;
    double* out; CK(hipMalloc(&out, sizeof(double) * blocks * 512));
    struct { const char* name; int a, b; double instr; } cases[] = {
        {"fp64 FMA, 8 waves", 0, 0, 8 * 256.0}, {"fp64 FMA, waves 0-3 only", 0, -1, 4 * 256.0},
        {"fp32 FMA, 8 waves", 1, 1, 8 * 256.0}, {"fp32 FMA, waves 0-3 only", 1, -1, 4 * 256.0},
        {"4 fp64 waves + 4 fp32 waves", 0, 1, 0}, {"cvt_f32_i32, 8 waves", 2, 2, 8 * 128.0}, {"v_perm_b32, 8 waves", 3, 3, 8 * 256.0},
        {"add+bfe_i32, 8 waves", 4, 4, 8 * 512.0}, {"4 fp64 waves + 4 perm waves", 0, 3, 0}, {"4 fp64 waves + 4 bfe waves", 0, 4, 0},
        {"one wave: 8 fp64 + 16 fp32 interleaved, 8 waves", 5, 5, 8 * 768.0}, {"... waves 0-3 only", 5, -1, 4 * 768.0},
    };
    for (auto& c : cases) {
        const float ms = timeit([&] { k_mix<<<blocks, 512>>>(out, iters, c.a, c.b, 1.0); });
        printf("%-52s %8.3f ms", c.name, ms);
        if (c.instr > 0) printf("   %6.2f cycles per wave-instruction per SIMD at 2.4 GHz", ms * 1e-3 * 2.4e9 / (c.instr * iters / 4.0 * 1.0) );
        printf("\n");
    }
    return 0;
}

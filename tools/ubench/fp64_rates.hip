// Microbenchmark: fp64 VALU / MFMA / conversion instruction rates on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 fp64_rates.hip -o fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

typedef double double4_t __attribute__((ext_vector_type(4)));

template<int OP>
__global__ void __launch_bounds__(256) k_valu(double* out, int iters, double seed) {
    double a[8];
    #pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i * 0.001 + threadIdx.x * 1e-6;
    double b = 0.999999, c = 1e-7;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            #pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = __builtin_fma(a[i], b, c);
                else if (OP == 1) a[i] = a[i] * b;
                else if (OP == 2) a[i] = a[i] + c;
                else if (OP == 3) a[i] = __builtin_rint(a[i] * 1.0000001) ;         // mul + rndne
                else if (OP == 4) a[i] = __builtin_ldexp(a[i], (int)u - 1);        // ldexp
                else if (OP == 5) a[i] = (double)(int)(a[i]) + 0.5;               // cvt i32<-f64, f64<-i32, add
                else if (OP == 6) a[i] = __builtin_fmax(a[i], c) ;                 // max
                else if (OP == 7) a[i] = exp(a[i] * -1e-3);                        // ocml exp
                else if (OP == 8) { float f = (float)a[i]; f = __builtin_amdgcn_exp2f(f); a[i] = (double)f; } // cvt+exp2f+cvt
            }
        }
    }
    double s = 0;
    #pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
    double4_t acc[4];
    #pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            #pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
    }
    double s = 0;
    #pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// mixed: MFMA + FMA in same wave (are pipes concurrent?)
__global__ void __launch_bounds__(256) k_mix(double* out, int iters, double seed) {
    double4_t acc[4];
    #pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a[8];
    #pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i * 0.001;
    double fa = threadIdx.x * 1e-3, fb = 1.0 - threadIdx.x * 1e-4;
    double b = 0.999999, c = 1e-7;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            #pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc[i], 0, 0, 0);
                // 16 FMAs per MFMA = 64 cycles of VALU per 64-cycle MFMA (if both at 78.6TF)
                #pragma unroll
                for (int r = 0; r < 2; ++r)
                #pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], b, c);
            }
        }
    }
    double s = 0;
    #pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    #pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_copy(const double4_t* __restrict__ in, double4_t* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
__global__ void k_write(double4_t* __restrict__ out, size_t n, double v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double4_t x = {v, v, v, v};
    for (; i < n; i += stride) out[i] = x;
}

template<typename F> float timeit(F f, int reps = 5) {
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
    printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    const int blocks = p.multiProcessorCount * 8, threads = 256, iters = 2000;
    double* out; CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    const char* names[] = {"fma_f64", "mul_f64", "add_f64", "mul+rndne", "ldexp_f64", "cvt_i32+cvt_f64+add", "max_f64", "ocml exp(f64)", "cvt+exp2f+cvt"};
    double per_it_ops = 32.0; // ops per thread per iter
    for (int op = 0; op < 9; ++op) {
        float ms = 0;
        switch (op) {
            case 0: ms = timeit([&]{ k_valu<0><<<blocks, threads>>>(out, iters, 1.0); }); break;
            case 1: ms = timeit([&]{ k_valu<1><<<blocks, threads>>>(out, iters, 1.0); }); break;
            case 2: ms = timeit([&]{ k_valu<2><<<blocks, threads>>>(out, iters, 1.0); }); break;
            case 3: ms = timeit([&]{ k_valu<3><<<blocks, threads>>>(out, iters, 1.0); }); break;
            case 4: ms = timeit([&]{ k_valu<4><<<blocks, threads>>>(out, iters, 1.0); }); break;
            case 5: ms = timeit([&]{ k_valu<5><<<blocks, threads>>>(out, iters, 1.0); }); break;
            case 6: ms = timeit([&]{ k_valu<6><<<blocks, threads>>>(out, iters, 1.0); }); break;
            case 7: ms = timeit([&]{ k_valu<7><<<blocks, threads>>>(out, iters / 10, 1.0); }); break;
            case 8: ms = timeit([&]{ k_valu<8><<<blocks, threads>>>(out, iters, 1.0); }); break;
        }
        double n_it = (op == 7) ? iters / 10 : iters;
        double ops = (double)blocks * threads * n_it * per_it_ops;
        printf("%-24s %8.3f ms  %8.2f Tops/s (lane-ops)\n", names[op], ms, ops / ms * 1e-9);
    }
    {
        float ms = timeit([&]{ k_mfma<<<blocks, threads>>>(out, iters); });
        double flops = (double)blocks * (threads / 64) * iters * 16.0 * 2.0 * 16 * 16 * 4;
        printf("mfma_f64_16x16x4         %8.3f ms  %8.2f TFLOP/s\n", ms, flops / ms * 1e-9);
    }
    {
        float ms = timeit([&]{ k_mix<<<blocks, threads>>>(out, iters, 1.0); });
        double mf = (double)blocks * (threads / 64) * iters * 16.0 * 2.0 * 16 * 16 * 4;
        double vf = (double)blocks * threads * iters * 16.0 * 16 * 2.0;
        printf("mix mfma+fma             %8.3f ms  mfma %8.2f TFLOP/s + valu %8.2f TFLOP/s\n", ms, mf / ms * 1e-9, vf / ms * 1e-9);
    }
    {
        size_t bytes = (size_t)4 << 30; size_t n = bytes / 32;
        double4_t *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 0, bytes));
        float ms = timeit([&]{ k_copy<<<p.multiProcessorCount * 8, 256>>>(a, b, n); });
        printf("copy 4GiB->4GiB          %8.3f ms  %8.2f TB/s (rd+wr)\n", ms, 2.0 * bytes / ms * 1e-9);
        ms = timeit([&]{ k_write<<<p.multiProcessorCount * 8, 256>>>(b, n, 1.0); });
        printf("write 4GiB               %8.3f ms  %8.2f TB/s\n", ms, 1.0 * bytes / ms * 1e-9);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}

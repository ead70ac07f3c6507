// Microbenchmark 2: do fp64 MFMA waves and fp64 VALU waves on the SAME SIMD overlap?
// and throughput of candidate hand-written exp2 kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef double double4_t __attribute__((ext_vector_type(4)));

// mode 0: all 8 waves MFMA ; 1: all 8 waves FMA ; 2: waves 0-3 MFMA, 4-7 FMA ; 3: waves 0-3 MFMA only (4-7 exit) ; 4: waves 4-7 FMA only
__global__ void __launch_bounds__(512) k_roles(double* out, int iters, int mode, double seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bool do_mfma = (mode == 0) || ((mode == 2 || mode == 3) && wave < 4);
    bool do_fma  = (mode == 1) || ((mode == 2 || mode == 4) && wave >= 4);
    double s = 0;
    if (do_mfma) {
        double4_t acc[4];
        #pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (double4_t){0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-4;
        for (int it = 0; it < iters; ++it) {
            #pragma unroll
            for (int u = 0; u < 4; ++u)
                #pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        #pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (do_fma) {
        double a[8];
        #pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = seed + i * 0.001 + threadIdx.x * 1e-6;
        double b = 0.999999, c = 1e-7;
        for (int it = 0; it < iters; ++it) {
            #pragma unroll
            for (int u = 0; u < 32; ++u)   // 256 FMAs/iter = 16 per MFMA-slot -> same 64-cycle budget per MFMA
                #pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], b, c);
        }
        #pragma unroll
        for (int i = 0; i < 8; ++i) s += a[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- candidate exp2 implementations for t <= 0 ------------------------------------
__constant__ double c_poly[16];
__device__ __forceinline__ double exp2_poly13(double t) {
    t = __builtin_fmax(t, -1020.0);
    double kd = __builtin_rint(t);
    double r = t - kd;
    double p = c_poly[13];
    #pragma unroll
    for (int i = 12; i >= 0; --i) p = __builtin_fma(p, r, c_poly[i]);
    return __builtin_ldexp(p, (int)kd);
}
// magic-number rounding + integer exponent insertion
__device__ __forceinline__ double exp2_poly13_magic(double t) {
    t = __builtin_fmax(t, -1020.0);
    const double MAGIC = 6755399441055744.0; // 1.5*2^52
    double a = t + MAGIC;
    double kd = a - MAGIC;
    double r = t - kd;
    double p = c_poly[13];
    #pragma unroll
    for (int i = 12; i >= 0; --i) p = __builtin_fma(p, r, c_poly[i]);
    long long bits = __double_as_longlong(p);
    int k = (int)__double_as_longlong(a);   // low 32 bits hold k (two's complement)
    bits += ((long long)k) << 52;
    return __longlong_as_double(bits);
}
// table (64 entries in LDS) + degree-6 polynomial
__device__ __forceinline__ double exp2_tab64(double t, const double* __restrict__ tab) {
    t = __builtin_fmax(t, -1020.0);
    const double MAGIC = 6755399441055744.0 / 64.0; // rounds to multiples of 1/64
    double a = t + MAGIC;
    double kd = a - MAGIC;           // multiple of 1/64
    double r = t - kd;               // |r| <= 1/128
    int ki = (int)__double_as_longlong(a);  // = 64*kd as integer (low bits)
    double tv = tab[ki & 63];
    double p = c_poly[6 + 16 - 16];
    p = c_poly[6];
    #pragma unroll
    for (int i = 5; i >= 0; --i) p = __builtin_fma(p, r, c_poly[i]);
    long long bits = __double_as_longlong(p * tv);
    bits += ((long long)(ki >> 6)) << 52;
    return __longlong_as_double(bits);
}

template<int V>
__global__ void __launch_bounds__(256) k_exp(double* out, int iters, double seed) {
    __shared__ double tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = exp2((double)threadIdx.x / 64.0);
    __syncthreads();
    double x[8], acc[8];
    #pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = -(seed + i * 0.37 + threadIdx.x * 0.01); acc[i] = 0; }
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 8; ++i) {
            double e;
            if (V == 0) e = exp2_poly13(x[i]);
            else if (V == 1) e = exp2_poly13_magic(x[i]);
            else if (V == 2) e = exp2_tab64(x[i], tab);
            else e = exp2(x[i]);
            acc[i] += e;
            x[i] = x[i] * 0.9999 - 1e-3;
        }
    }
    double s = 0;
    #pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<int V>
__global__ void k_exp_check(const double* __restrict__ in, double* __restrict__ out, int n) {
    __shared__ double tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = exp2((double)threadIdx.x / 64.0);
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double t = in[i];
        out[i] = (V == 0) ? exp2_poly13(t) : (V == 1) ? exp2_poly13_magic(t) : (V == 2) ? exp2_tab64(t, tab) : exp2(t);
    }
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
    const int blocks = p.multiProcessorCount * 2, iters = 2000;
    double* out; CK(hipMalloc(&out, sizeof(double) * blocks * 512 * 4));
    const char* names[] = {"all 8 waves MFMA", "all 8 waves FMA", "4 MFMA + 4 FMA waves", "4 MFMA waves only", "4 FMA waves only"};
    for (int mode = 0; mode < 5; ++mode) {
        float ms = timeit([&]{ k_roles<<<blocks, 512>>>(out, iters, mode, 1.0); });
        printf("%-24s %8.3f ms\n", names[mode], ms);
    }
    // polynomial coefficients for 2^r : Taylor ln2^i / i!
    {
        double c[16]; double ln2 = 0.6931471805599453094; double f = 1.0;
        for (int i = 0; i < 16; ++i) { c[i] = f; f *= ln2 / (i + 1); }
        CK(hipMemcpyToSymbol(HIP_SYMBOL(c_poly), c, sizeof(c)));
    }
    const int eb = p.multiProcessorCount * 8;
    const char* en[] = {"exp2 poly13 rint+ldexp", "exp2 poly13 magic+int", "exp2 tab64+poly6", "ocml exp2"};
    for (int v = 0; v < 4; ++v) {
        float ms = 0;
        if (v == 0) ms = timeit([&]{ k_exp<0><<<eb, 256>>>(out, 500, 1.0); });
        if (v == 1) ms = timeit([&]{ k_exp<1><<<eb, 256>>>(out, 500, 1.0); });
        if (v == 2) ms = timeit([&]{ k_exp<2><<<eb, 256>>>(out, 500, 1.0); });
        if (v == 3) ms = timeit([&]{ k_exp<3><<<eb, 256>>>(out, 500, 1.0); });
        double n = (double)eb * 256 * 500 * 8;
        printf("%-24s %8.3f ms  %8.3f Texp/s (incl. 1 fma + 1 add overhead per exp)\n", en[v], ms, n / ms * 1e-9);
    }
    // accuracy check
    {
        const int n = 1 << 20;
        std::vector<double> h(n), o(n);
        srand(1);
        for (int i = 0; i < n; ++i) { double u = rand() / (double)RAND_MAX; h[i] = -u * u * 200.0; }
        h[0] = 0.0; h[1] = -0.5; h[2] = -1.0; h[3] = -1e-300; h[4] = -1019.5; h[5] = -5000.0;
        double *din, *dout; CK(hipMalloc(&din, n * 8)); CK(hipMalloc(&dout, n * 8));
        CK(hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice));
        for (int v = 0; v < 4; ++v) {
            if (v == 0) k_exp_check<0><<<n / 256, 256>>>(din, dout, n);
            if (v == 1) k_exp_check<1><<<n / 256, 256>>>(din, dout, n);
            if (v == 2) k_exp_check<2><<<n / 256, 256>>>(din, dout, n);
            if (v == 3) k_exp_check<3><<<n / 256, 256>>>(din, dout, n);
            CK(hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost));
            double maxrel = 0; int worst = 0;
            for (int i = 0; i < n; ++i) {
                long double ref = exp2l((long double)h[i]);
                if (h[i] < -1000) continue;
                double rel = (double)fabsl(((long double)o[i] - ref) / ref);
                if (rel > maxrel) { maxrel = rel; worst = i; }
            }
            printf("%-24s max rel err %.3e (%.2f ulp) at t=%g ; f(0)=%.17g f(-5000)=%g\n", en[v], maxrel, maxrel / 1.11e-16, h[worst], o[0], o[5]);
        }
    }
    return 0;
}

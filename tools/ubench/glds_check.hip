// Does global_load_lds_dwordx4 land lane-linear (16 B per lane from the M0 base)?  hipcc --offload-arch=gfx950 -O3 glds_check.hip -o glds_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(const double* __restrict__ src, double* out, int ntile) {
    __shared__ __attribute__((aligned(16))) double Ls[4 * 2048];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int t = 0; t < ntile; ++t) {
        const double* s = src + (size_t)t * 2048 + wave * 512 + lane * 2;
        double* d = Ls + (t & 3) * 2048 + wave * 512;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + q * 128),
                                             (__attribute__((address_space(3))) void*)(d + q * 128), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int e = tid; e < 2048; e += 256) out[(size_t)t * 2048 + e] = Ls[(t & 3) * 2048 + e];
        __builtin_amdgcn_s_barrier();
    }
}
int main() {
    const int ntile = 8;
    std::vector<double> h(ntile * 2048), o(ntile * 2048);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)i;
    double *d, *e;
    hipMalloc(&d, h.size() * 8); hipMalloc(&e, h.size() * 8);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    k<<<1, 256>>>(d, e, ntile);
    hipMemcpy(o.data(), e, h.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < h.size(); ++i) if (o[i] != h[i]) { if (bad < 8) printf("mismatch at %zu: %g\n", i, o[i]); ++bad; }
    printf("glds check: %d mismatches of %zu\n", bad, h.size());
    return bad != 0;
}

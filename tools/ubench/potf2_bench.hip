// One-wave 32 x 32 Cholesky (potf2_wave of csrc/factor.hip) timed inside the kernel.  -DOLD=1 -DOLD_SRC='"file"' builds
// against a saved copy of an earlier factor.hip (potf2_wave without the Lt scratch argument) for an A/B.
// Measured on MI355X (r02): 5.47 us left-looking (one LDS round trip + a j-long dot per column) -> 4.97 us right-looking.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I include -I orthogonal-additive-gaussian-processes_amd/csrc tools/ubench/potf2_bench.hip -o tools/ubench/potf2_bench
#ifdef OLD
#include OLD_SRC
#else
#include "factor.hip"
#endif
#include <cstdio>
#include <cmath>
#include <vector>
namespace oak {
void set_error(const char*, ...) {}
int get_buf(oak_ctx*, const char*, size_t, void**) { return 0; }
void* peek_buf(oak_ctx*, const char*) { return nullptr; }
int set_identity(oak_ctx*, double*, int64_t) { return 0; }
int transpose(oak_ctx*, const double*, int64_t, int64_t, int64_t, double*, int64_t) { return 0; }
__global__ void __launch_bounds__(64) t_potf2(const double* A, double* out, long long* cyc, int reps) {
    __shared__ __attribute__((aligned(16))) double Dg[PO_NB * PO_P];
    __shared__ __attribute__((aligned(16))) double Lt[PO_NB * PO_P];
    __shared__ double invd[PO_NB];
    long long total = 0;
    for (int r = 0; r < reps; ++r) {
        for (int k = threadIdx.x; k < PO_NB * PO_NB; k += 64) Dg[(k / PO_NB) * PO_P + k % PO_NB] = A[k];
        __syncthreads();
        const long long t0 = wall_clock64();
#ifdef OLD
        potf2_wave(Dg, invd, threadIdx.x, 0, nullptr);
#else
        potf2_wave(Dg, Lt, invd, threadIdx.x, 0, nullptr);
#endif
        __syncthreads();
        total += wall_clock64() - t0;
    }
    (void)Lt;
    for (int k = threadIdx.x; k < PO_NB * PO_NB; k += 64) out[k] = Dg[(k / PO_NB) * PO_P + k % PO_NB];
    if (threadIdx.x == 0) *cyc = total;
}
}
int main() {
    const int n = 32;
    std::vector<double> A(n * n), L(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = std::exp(-0.05 * std::abs(i - j)) + (i == j ? 0.01 : 0.0);
    double *dA, *dO; long long* dC;
    hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dO, sizeof(double) * n * n); hipMalloc(&dC, 8);
    hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    const int reps = 200;
    for (int it = 0; it < 2; ++it) oak::t_potf2<<<1, 64>>>(dA, dO, dC, reps);
    long long c = 0;
    hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost); hipMemcpy(L.data(), dO, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    double err = 0;   // ||L L^T - A||
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += L[i * n + k] * L[j * n + k]; err = std::fmax(err, std::fabs(s - A[i * n + j])); }
    printf("potf2 32x32: %.3f us per factorisation (100 MHz counter, %d reps), max |L L^T - A| = %.2e\n", c / 100.0 / reps, reps, err);
    return 0;
}

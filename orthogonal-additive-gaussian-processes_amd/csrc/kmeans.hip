// Lloyd k-means on the device: inducing-point initialisation for the sparse model.
//
// Replaces the scikit-learn KMeans.fit calls of the reference (oak/model_utils.py:31-41 get_kmeans_centers, called from
// oak_model.fit :377-391; oak/utils.py:533-574 for the continuous block of the mixed-type initialisers).  scikit-learn's
// single-run Lloyd loop (sklearn/cluster/_kmeans.py::_kmeans_single_lloyd) is followed step for step:
//     repeat:  E-step  labels_i = argmin_k |x_i - c_k|^2 (first minimum wins)
//              M-step  c_k = mean of the points labelled k; empty clusters take the points farthest from their centre
//              stop when the labels did not change (strict convergence) or sum_k |c_k_new - c_k|^2 <= tol
//     if not strictly converged: one more E-step against the final centres;  inertia = sum_i min_k |x_i - c_k|^2
// Seeding (k-means++) stays on the host: it is sequential in K and touches a subsample only.
//
// Kernels (all deterministic: no floating-point atomics, fixed reduction trees):
//   kmeans_assign_kernel   lane = P points held in registers, centres streamed through LDS (broadcast reads), distance in
//                          the direct form sum_d (x_d - c_d)^2 (2 DP instructions per point-centre-dimension).
//                          fp64-VALU bound: N*K*D*2 instructions; HBM traffic is N*D*8 B (negligible).
//   kmeans_chunk_sums_kernel + kmeans_combine_kernel (r04)  the M-step as a two-level sum over point chunks, cluster sums in LDS, fixed order
//                          (0.33 ms at N = 2^20, K = 1024, D = 16; the kernel below took 0.98)
//   kmeans_sums_kernel     (OAK_KMEANS_SCAN_SUMS=1) one WORKGROUP per cluster scans the label array (L2 / MALL resident, N*4 B; a quarter
//                          per wave) and gathers its member rows in ascending point order; per-lane partial sums, fixed butterfly,
//                          then the four waves in order.
//   kmeans_finalize_kernel centres = sums / counts, per-cluster squared shift.
#include "oak_internal.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace oak {

#ifndef OAK_KM_TWO_CENTRES
#define OAK_KM_TWO_CENTRES 1
#endif
static constexpr int KM_LDS_DOUBLES = 4096;   // 32 KiB of centres per LDS chunk

template <int DMAX, int P>
__global__ void __launch_bounds__(256)
kmeans_assign_kernel(const double* __restrict__ X, int64_t N, int D, int64_t ldx, const double* __restrict__ C /* K x DMAX, zero padded */,
                     int K, const int32_t* __restrict__ labels_old, int32_t* __restrict__ labels, double* __restrict__ mind,
                     int* __restrict__ changed) {
    constexpr int KCH = KM_LDS_DOUBLES / DMAX;
    __shared__ __attribute__((aligned(16))) double sC[KM_LDS_DOUBLES];
    const int tid = threadIdx.x;
    double x[P][DMAX];
    int64_t pt[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        pt[p] = ((int64_t)blockIdx.x * P + p) * 256 + tid;
        const int64_t i = pt[p] < N ? pt[p] : N - 1;          // clamped address: no divergent loads
#pragma unroll
        for (int d = 0; d < DMAX; ++d) x[p][d] = (d < D) ? X[i * ldx + d] : 0.0;
    }
    double best[P];
    int bi[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { best[p] = __builtin_inf(); bi[p] = 0; }
    for (int k0 = 0; k0 < K; k0 += KCH) {
        const int kc = (K - k0 < KCH) ? K - k0 : KCH;
        __syncthreads();
        for (int idx = tid; idx < kc * DMAX; idx += 256) sC[idx] = C[(int64_t)k0 * DMAX + idx];
        __syncthreads();
        int kk = 0;
#if OAK_KM_TWO_CENTRES
        // two centres per trip: 2 P independent accumulation chains per lane instead of P (each chain is DMAX dependent FMAs)
        for (; kk + 1 < kc; kk += 2) {
            double da[P], db[P];
#pragma unroll
            for (int p = 0; p < P; ++p) { da[p] = 0.0; db[p] = 0.0; }
#pragma unroll
            for (int d = 0; d < DMAX; d += 2) {
                const double2 ca = *reinterpret_cast<const double2*>(&sC[kk * DMAX + d]);          // uniform addresses: LDS broadcast
                const double2 cb = *reinterpret_cast<const double2*>(&sC[(kk + 1) * DMAX + d]);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const double a0 = x[p][d] - ca.x, b0 = x[p][d] - cb.x;
                    da[p] = __builtin_fma(a0, a0, da[p]);
                    db[p] = __builtin_fma(b0, b0, db[p]);
                    const double a1 = x[p][d + 1] - ca.y, b1 = x[p][d + 1] - cb.y;
                    da[p] = __builtin_fma(a1, a1, da[p]);
                    db[p] = __builtin_fma(b1, b1, db[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {                       // in centre order, strict <: the first minimum wins
                if (da[p] < best[p]) { best[p] = da[p]; bi[p] = k0 + kk; }
                if (db[p] < best[p]) { best[p] = db[p]; bi[p] = k0 + kk + 1; }
            }
        }
#endif
        for (; kk < kc; ++kk) {
            double dist[P];
#pragma unroll
            for (int p = 0; p < P; ++p) dist[p] = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; d += 2) {
                const double2 c2 = *reinterpret_cast<const double2*>(&sC[kk * DMAX + d]);   // uniform address: LDS broadcast
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const double t0 = x[p][d] - c2.x;
                    dist[p] = __builtin_fma(t0, t0, dist[p]);
                    const double t1 = x[p][d + 1] - c2.y;
                    dist[p] = __builtin_fma(t1, t1, dist[p]);
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p)
                if (dist[p] < best[p]) { best[p] = dist[p]; bi[p] = k0 + kk; }      // strict <: the first minimum wins
        }
    }
    int nchanged = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (pt[p] < N) {
            if (labels_old == nullptr || labels_old[pt[p]] != bi[p]) ++nchanged;
            labels[pt[p]] = bi[p];
            mind[pt[p]] = best[p];
        }
    }
    if (nchanged) atomicAdd(changed, nchanged);    // integer count: order-independent
}

__device__ __forceinline__ double km_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One workgroup per cluster.  Wave w scans the w-th quarter of the label array (lane l visits points q0 + l, q0 + l + 64,
// ... in ascending order, eight label loads in flight per trip to cover the L2 / MALL latency) and gathers the member
// rows into per-lane partial sums; fixed butterfly within the wave, then the four waves are added in order 0..3.
template <int DMAX>
__global__ void __launch_bounds__(256)
kmeans_sums_kernel(const double* __restrict__ X, int64_t N, int D, int64_t ldx, const int32_t* __restrict__ labels, int K,
                   double* __restrict__ sums /* K x DMAX */, int32_t* __restrict__ counts) {
    __shared__ double part[4][DMAX];
    __shared__ int pcnt[4];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = blockIdx.x;
    double s[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) s[d] = 0.0;
    int cnt = 0;
    const int64_t q = ((N + 3) / 4 + 63) / 64 * 64;          // quarter length, multiple of 64
    const int64_t q0 = (int64_t)w * q;
    const int64_t q1 = (q0 + q < N) ? q0 + q : N;
    auto take = [&](int64_t i) {
        const double* r = X + i * ldx;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) if (d < D) s[d] += r[d];
        ++cnt;
    };
    int64_t i = q0 + lane;
    for (; i + 7 * 64 < q1; i += 512) {
        int32_t l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) l[u] = labels[i + 64 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (l[u] == c) take(i + 64 * u);
    }
    for (; i < q1; i += 64) if (labels[i] == c) take(i);
#pragma unroll
    for (int d = 0; d < DMAX; ++d) s[d] = km_wave_sum(s[d]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) part[w][d] = s[d];
        pcnt[w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < DMAX) {
        const int d = threadIdx.x;
        sums[(int64_t)c * DMAX + d] = ((part[0][d] + part[1][d]) + part[2][d]) + part[3][d];
    }
    if (threadIdx.x == 0) counts[c] = ((pcnt[0] + pcnt[1]) + pcnt[2]) + pcnt[3];
}

// r04: the M-step as a two-level sum over POINT chunks instead of one label scan per cluster (K workgroups x N labels = 4 GB of L2 reads at
// N = 2^20, K = 1024: 0.98 ms).  A workgroup takes a chunk of points and keeps the running sums of a range of clusters in LDS
// ([KC][DMAX] doubles: 1024 x 16 fit); the chunk goes through LDS 64 rows at a time (coalesced loads, the next group prefetched into
// registers), wave w adds the points whose cluster index is = w mod 4 -- one point at a time in index order, DMAX lanes wide -- so no two
// waves ever touch the same accumulator and every sum is formed in a fixed order.  kmeans_combine_kernel adds the chunks' partial sums in
// chunk order.  Deterministic like the kernel it replaces (which stays for K beyond what the ranges cover cheaply: never, in practice).
template <int DMAX>
__global__ void __launch_bounds__(256)
kmeans_chunk_sums_kernel(const double* __restrict__ X, int64_t N, int D, int64_t ldx, const int32_t* __restrict__ labels, int K, int KC,
                         int64_t pch, double* __restrict__ part /* [chunk][range][KC][DMAX] */, int32_t* __restrict__ pcnt /* [chunk][range][KC] */) {
    extern __shared__ __attribute__((aligned(16))) double km_smem[];
    double* acc = km_smem;                                   // [KC][DMAX]
    double* rows = acc + (size_t)KC * DMAX;                  // [64][DMAX]
    int* cnt = reinterpret_cast<int*>(rows + 64 * DMAX);     // [KC]
    int* lab = cnt + KC;                                     // [64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nrange = gridDim.y, range = blockIdx.y;
    const int k0 = range * KC, kc = (K - k0 < KC) ? K - k0 : KC;
    const int64_t i0 = (int64_t)blockIdx.x * pch, i1 = (i0 + pch < N) ? i0 + pch : N;
    for (int idx = tid; idx < KC * DMAX; idx += 256) acc[idx] = 0.0;
    for (int idx = tid; idx < KC; idx += 256) cnt[idx] = 0;
    constexpr int NV = DMAX / 4;                             // values of a 64-row group per thread
    double v[NV];
    int lv = -1;
    auto fetch = [&](int64_t g0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int e = tid + 256 * q, r = e / DMAX, d = e - r * DMAX;
            const int64_t i = g0 + r;
            v[q] = (i < i1 && d < D) ? X[i * ldx + d] : 0.0;
        }
        lv = (tid < 64 && g0 + tid < i1) ? labels[g0 + tid] : -1;
    };
    if (i0 < i1) fetch(i0);
    for (int64_t g0 = i0; g0 < i1; g0 += 64) {
        __syncthreads();                                     // readers of the previous group done (and the accumulators zeroed)
#pragma unroll
        for (int q = 0; q < NV; ++q) rows[tid + 256 * q] = v[q];
        if (tid < 64) lab[tid] = lv;
        __syncthreads();
        if (g0 + 64 < i1) fetch(g0 + 64);
        const int l = lab[lane] - k0;
        unsigned long long mask = __ballot(l >= 0 && l < kc && (l & 3) == w);
        while (mask) {
            const int j = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int lj = lab[j] - k0;                      // uniform address: broadcast
            if (lane < DMAX) acc[lj * DMAX + lane] += rows[j * DMAX + lane];
            if (lane == 0) cnt[lj] += 1;
        }
    }
    __syncthreads();
    double* out = part + ((int64_t)blockIdx.x * nrange + range) * KC * DMAX;
    for (int idx = tid; idx < kc * DMAX; idx += 256) out[idx] = acc[idx];
    int32_t* oc = pcnt + ((int64_t)blockIdx.x * nrange + range) * KC;
    for (int idx = tid; idx < kc; idx += 256) oc[idx] = cnt[idx];
}

// sums[k][d] = sum over the chunks, in chunk order; counts likewise
__global__ void __launch_bounds__(256)
kmeans_combine_kernel(const double* __restrict__ part, const int32_t* __restrict__ pcnt, int nchunk, int nrange, int KC, int K, int DMAX,
                      double* __restrict__ sums, int32_t* __restrict__ counts) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)K * DMAX) return;
    const int k = (int)(e / DMAX), d = (int)(e - (int64_t)k * DMAX);
    const int range = k / KC, kl = k - range * KC;
    double s = 0.0;
    int c = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
        s += part[(((int64_t)ch * nrange + range) * KC + kl) * DMAX + d];
        if (d == 0) c += pcnt[((int64_t)ch * nrange + range) * KC + kl];
    }
    sums[e] = s;
    if (d == 0) counts[k] = c;
}

// centres_new = sums / counts (empty cluster: keep the old centre; the host relocates those before calling this);
// shift2[k] = |c_new - c_old|^2
__global__ void kmeans_finalize_kernel(const double* __restrict__ sums, const int32_t* __restrict__ counts, const double* __restrict__ Cold,
                                       double* __restrict__ Cnew, double* __restrict__ shift2, int K, int DMAX) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int n = counts[k];
    double sh = 0.0;
    for (int d = 0; d < DMAX; ++d) {
        const double co = Cold[(int64_t)k * DMAX + d];
        const double cn = n > 0 ? sums[(int64_t)k * DMAX + d] / (double)n : co;
        Cnew[(int64_t)k * DMAX + d] = cn;
        const double t = cn - co;
        sh = __builtin_fma(t, t, sh);
    }
    shift2[k] = sh;
}

template <int DMAX>
static int km_launch_assign(oak_ctx* ctx, const double* dX, int64_t N, int D, int64_t ldx, const double* dC, int K,
                            const int32_t* lab_old, int32_t* lab, double* mind, int* changed) {
    constexpr int P = DMAX <= 16 ? 4 : (DMAX <= 32 ? 2 : 1);
    const int64_t per_wg = 256 * P;
    const unsigned grid = (unsigned)((N + per_wg - 1) / per_wg);
    kmeans_assign_kernel<DMAX, P><<<grid, 256, 0, ctx->stream>>>(dX, N, D, ldx, dC, K, lab_old, lab, mind, changed);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

template <int DMAX>
static int km_launch_sums(oak_ctx* ctx, const double* dX, int64_t N, int D, int64_t ldx, const int32_t* lab, int K, double* sums,
                          int32_t* counts) {
    if (getenv("OAK_KMEANS_SCAN_SUMS") != nullptr) {          // the one-workgroup-per-cluster label scan this replaced (A/B)
        kmeans_sums_kernel<DMAX><<<(unsigned)K, 256, 0, ctx->stream>>>(dX, N, D, ldx, lab, K, sums, counts);
        OAK_HIP_CHECK(hipGetLastError());
        return OAK_OK;
    }
    // clusters per range: what 150 KiB of LDS hold next to the 64-row staging tile; point chunks: 2048 points, at most 1024 chunks
    int KC = (int)((150 * 1024 - 64 * DMAX * 8 - 256) / (DMAX * 8 + 4)) & ~3;
    if (KC > K) KC = (K + 3) & ~3;
    const int nrange = (K + KC - 1) / KC;
    int64_t pch = 2048;
    if ((N + pch - 1) / pch > 1024) pch = (((N + 1023) / 1024 + 63) / 64) * 64;
    const int nchunk = (int)((N + pch - 1) / pch);
    double* d_part = nullptr;
    int32_t* d_pcnt = nullptr;
    OAK_CHECK(get_buf_t(ctx, "km_part", (size_t)nchunk * nrange * KC * DMAX, &d_part));
    OAK_CHECK(get_buf_t(ctx, "km_pcnt", (size_t)nchunk * nrange * KC, &d_pcnt));
    const size_t lds = sizeof(double) * ((size_t)KC * DMAX + 64 * DMAX) + sizeof(int) * ((size_t)KC + 64);
    auto kern = kmeans_chunk_sums_kernel<DMAX>;
    if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)kern));
    kern<<<dim3((unsigned)nchunk, (unsigned)nrange), 256, lds, ctx->stream>>>(dX, N, D, ldx, lab, K, KC, pch, d_part, d_pcnt);
    OAK_HIP_CHECK(hipGetLastError());
    kmeans_combine_kernel<<<(unsigned)(((int64_t)K * DMAX + 255) / 256), 256, 0, ctx->stream>>>(d_part, d_pcnt, nchunk, nrange, KC, K, DMAX, sums, counts);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

#define KM_DISPATCH(FN, ...)                                  \
    (dmax == 8 ? FN<8>(__VA_ARGS__) : dmax == 16 ? FN<16>(__VA_ARGS__) : dmax == 32 ? FN<32>(__VA_ARGS__) : FN<64>(__VA_ARGS__))

}  // namespace oak

using namespace oak;

extern "C" int oak_kmeans(oak_ctx* ctx, const double* X, int64_t N, int32_t D, int32_t ldx, int32_t K, const double* init_centres,
                          int32_t max_iter, double tol, double* centres_out, int32_t* labels_out, double* inertia_out,
                          int32_t* n_iter_out) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(X && init_centres && centres_out, "oak_kmeans: NULL argument");
    OAK_REQUIRE(N >= 1 && D >= 1 && D <= 64 && ldx >= D && K >= 1 && K <= N && max_iter >= 1 && tol >= 0.0,
                "oak_kmeans: bad sizes (N=%lld D=%d ldx=%d K=%d max_iter=%d)", (long long)N, D, ldx, K, max_iter);
    const int dmax = D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : 64;
    PhaseTimer ttot(ctx, "kmeans");
    double *dX, *dC[2], *dSums, *dShift, *dMind, *dScal;
    int32_t *dLab[2], *dCnt;
    int* dChanged;
    OAK_CHECK(get_buf_t(ctx, "km_X", (size_t)N * ldx, &dX));
    OAK_CHECK(get_buf_t(ctx, "km_C0", (size_t)K * dmax, &dC[0]));
    OAK_CHECK(get_buf_t(ctx, "km_C1", (size_t)K * dmax, &dC[1]));
    OAK_CHECK(get_buf_t(ctx, "km_sums", (size_t)K * dmax, &dSums));
    OAK_CHECK(get_buf_t(ctx, "km_shift", (size_t)K, &dShift));
    OAK_CHECK(get_buf_t(ctx, "km_mind", (size_t)N, &dMind));
    OAK_CHECK(get_buf_t(ctx, "km_scal", 4, &dScal));
    OAK_CHECK(get_buf_t(ctx, "km_lab0", (size_t)N, &dLab[0]));
    OAK_CHECK(get_buf_t(ctx, "km_lab1", (size_t)N, &dLab[1]));
    OAK_CHECK(get_buf_t(ctx, "km_cnt", (size_t)K, &dCnt));
    OAK_CHECK(get_buf_t(ctx, "km_changed", 2, &dChanged));
    OAK_HIP_CHECK(hipMemcpyAsync(dX, X, sizeof(double) * (size_t)N * ldx, hipMemcpyHostToDevice, ctx->stream));
    std::vector<double> hC((size_t)K * dmax, 0.0), hSums;
    for (int k = 0; k < K; ++k)
        for (int d = 0; d < D; ++d) hC[(size_t)k * dmax + d] = init_centres[(size_t)k * D + d];
    OAK_HIP_CHECK(hipMemcpyAsync(dC[0], hC.data(), sizeof(double) * hC.size(), hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));

    std::vector<int32_t> hCnt((size_t)K);
    int cur = 0, lab = 0;            // dC[cur] = current centres; dLab[lab] receives this iteration's labels
    bool strict = false, have_old = false;
    int it = 0;
    for (; it < max_iter; ++it) {
        OAK_HIP_CHECK(hipMemsetAsync(dChanged, 0, sizeof(int), ctx->stream));
        OAK_CHECK(KM_DISPATCH(km_launch_assign, ctx, dX, N, D, ldx, dC[cur], K, have_old ? dLab[lab ^ 1] : nullptr, dLab[lab], dMind, dChanged));
        OAK_CHECK(KM_DISPATCH(km_launch_sums, ctx, dX, N, D, ldx, dLab[lab], K, dSums, dCnt));
        OAK_HIP_CHECK(hipMemcpyAsync(hCnt.data(), dCnt, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, ctx->stream));
        int h_changed = 0;
        OAK_HIP_CHECK(hipMemcpyAsync(&h_changed, dChanged, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        // empty clusters: give each the point farthest from its own centre, taken out of its donor cluster
        // (sklearn _relocate_empty_clusters_dense; largest distances first, ties by lower index)
        std::vector<int> empty;
        for (int k = 0; k < K; ++k) if (hCnt[(size_t)k] == 0) empty.push_back(k);
        if (!empty.empty()) {
            std::vector<double> hMind((size_t)N);
            std::vector<int32_t> hLab((size_t)N);
            hSums.resize((size_t)K * dmax);
            OAK_CHECK(copy_sync(ctx, hMind.data(), dMind, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost));
            OAK_CHECK(copy_sync(ctx, hLab.data(), dLab[lab], sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost));
            OAK_CHECK(copy_sync(ctx, hSums.data(), dSums, sizeof(double) * hSums.size(), hipMemcpyDeviceToHost));
            std::vector<int64_t> order((size_t)N);
            for (int64_t i = 0; i < N; ++i) order[(size_t)i] = i;
            const size_t ne = empty.size();
            std::partial_sort(order.begin(), order.begin() + (std::ptrdiff_t)ne, order.end(), [&](int64_t a, int64_t b) {
                return hMind[(size_t)a] > hMind[(size_t)b] || (hMind[(size_t)a] == hMind[(size_t)b] && a < b);
            });
            for (size_t e = 0; e < ne; ++e) {
                const int64_t far = order[e];
                const int donor = hLab[(size_t)far], target = empty[e];
                for (int d = 0; d < D; ++d) {
                    hSums[(size_t)donor * dmax + d] -= X[(size_t)far * ldx + d];
                    hSums[(size_t)target * dmax + d] = X[(size_t)far * ldx + d];
                }
                hCnt[(size_t)target] = 1;
                hCnt[(size_t)donor] -= 1;
            }
            OAK_CHECK(copy_sync(ctx, dSums, hSums.data(), sizeof(double) * hSums.size(), hipMemcpyHostToDevice));
            OAK_CHECK(copy_sync(ctx, dCnt, hCnt.data(), sizeof(int32_t) * (size_t)K, hipMemcpyHostToDevice));
        }
        kmeans_finalize_kernel<<<(unsigned)((K + 255) / 256), 256, 0, ctx->stream>>>(dSums, dCnt, dC[cur], dC[cur ^ 1], dShift, K, dmax);
        OAK_HIP_CHECK(hipGetLastError());
        OAK_CHECK(reduce_sum(ctx, dShift, K, dScal, 0, 1));
        double shift_tot = 0.0;
        OAK_HIP_CHECK(hipMemcpyAsync(&shift_tot, dScal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        cur ^= 1;                                        // centres <- centres_new
        if (have_old && h_changed == 0) { strict = true; ++it; break; }
        if (shift_tot <= tol) { lab ^= 1; have_old = true; ++it; break; }
        lab ^= 1;                                        // this iteration's labels become labels_old
        have_old = true;
    }
    // after the loop dLab[lab ^ 1] holds the most recent labels unless we broke out on strict convergence
    int32_t* dFinalLab = strict ? dLab[lab] : dLab[lab ^ 1];
    if (!strict) {
        // E-step against the final centres so labels and inertia match them
        OAK_HIP_CHECK(hipMemsetAsync(dChanged, 0, sizeof(int), ctx->stream));
        OAK_CHECK(KM_DISPATCH(km_launch_assign, ctx, dX, N, D, ldx, dC[cur], K, nullptr, dLab[lab], dMind, dChanged));
        dFinalLab = dLab[lab];
    }
    OAK_CHECK(reduce_sum(ctx, dMind, N, dScal, 0, 1));
    double inertia = 0.0;
    OAK_HIP_CHECK(hipMemcpyAsync(&inertia, dScal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(hC.data(), dC[cur], sizeof(double) * hC.size(), hipMemcpyDeviceToHost, ctx->stream));
    if (labels_out) OAK_HIP_CHECK(hipMemcpyAsync(labels_out, dFinalLab, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ttot.stop();
    for (int k = 0; k < K; ++k)
        for (int d = 0; d < D; ++d) centres_out[(size_t)k * D + d] = hC[(size_t)k * dmax + d];
    if (inertia_out) *inertia_out = inertia;
    if (n_iter_out) *n_iter_out = it;
    return OAK_OK;
}

// =====================================================================================================================
// k-means++ seeding on the device (greedy variant with local trials), following scikit-learn's _kmeans_plusplus
// (sklearn/cluster/_kmeans.py) that KMeans.fit runs before Lloyd:
//     first centre = given index; closest_i = |x_i - c_0|^2; pot = sum closest
//     for c = 1..K-1:  r_t = u[c-1][t] * pot (t < n_trials; the uniforms are drawn by the HOST from the caller's RandomState,
//                      in scikit-learn's order);  candidate_t = searchsorted(cumsum(closest), r_t), clipped to N-1;
//                      pot_t = sum_i min(closest_i, |x_i - x_cand_t|^2);  best = argmin_t pot_t (first minimum);
//                      closest <- min(closest, dist_best); pot <- pot_best; centre_c = candidate_best
// The whole loop is enqueued without a host synchronisation (five small kernels per centre).  Distances use the direct
// form; cumulative sums are two-level (4096-element blocks, fixed order): against scikit-learn the chosen indices can
// differ only when a random threshold falls within rounding distance of a cumulative-sum boundary.
// =====================================================================================================================
namespace oak {

static constexpr int KPP_BLOCK = 4096;      // elements per cumulative-sum block (16 per thread)
static constexpr int KPP_MAXT = 16;         // max local trials (2 + log K <= 16 for K <= 1.2e6)

__device__ __forceinline__ double kpp_block_reduce(double v, double* red /*[256]*/) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// closest_i = |x_i - x_first|^2 and the per-block sums of closest
template <int DMAX>
__global__ void __launch_bounds__(256)
kpp_init_kernel(const double* __restrict__ X, int64_t N, int D, int64_t ldx, int64_t first, double* __restrict__ closest,
                double* __restrict__ bsum, int64_t* __restrict__ indices) {
    __shared__ double red[256];
    if (blockIdx.x == 0 && threadIdx.x == 0) indices[0] = first;
    const int64_t b0 = (int64_t)blockIdx.x * KPP_BLOCK;
    double c[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) c[d] = d < D ? X[first * ldx + d] : 0.0;
    double s = 0.0;
    for (int q = 0; q < KPP_BLOCK / 256; ++q) {
        const int64_t i = b0 + (int64_t)threadIdx.x * (KPP_BLOCK / 256) + q;      // 16 consecutive elements per thread
        if (i < N) {
            double dist = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) { const double t = (d < D ? X[i * ldx + d] : 0.0) - c[d]; dist = __builtin_fma(t, t, dist); }
            closest[i] = dist;
            s += dist;
        }
    }
    s = kpp_block_reduce(s, red);
    if (threadIdx.x == 0) bsum[blockIdx.x] = s;
}

// One workgroup: prefix over the block sums, then for each trial the candidate index = first i with cumsum_i >= r_t.
// state[0] = current potential (in: from the previous selection; the cumulative total is NOT used, as in scikit-learn).
__global__ void __launch_bounds__(256)
kpp_pick_kernel(const double* __restrict__ closest, const double* __restrict__ bsum, int nb, int64_t N,
                const double* __restrict__ state, const double* __restrict__ uniforms, int n_trials, int64_t* __restrict__ cand) {
    extern __shared__ double sm[];          // [nb + 1] exclusive prefix of the block sums, then [256] scratch
    double* pref = sm;
    double* scr = sm + nb + 1;
    if (threadIdx.x == 0) {                 // nb <= 4096 (N <= 2^24): a short sequential pass, same order as a running sum
        double a = 0.0;
        for (int b = 0; b < nb; ++b) { pref[b] = a; a += bsum[b]; }
        pref[nb] = a;
    }
    __syncthreads();
    const double pot = state[0];
    for (int t = 0; t < n_trials; ++t) {
        const double r = uniforms[t] * pot;
        // block: last b with pref[b] < r  (so that the answer lies inside block b); r <= 0 -> element 0
        int lo = 0, hi = nb;                // invariant: pref[lo] < r or lo == 0
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pref[mid] < r) lo = mid; else hi = mid; }
        const int b = lo;
        const int64_t e0 = (int64_t)b * KPP_BLOCK + (int64_t)threadIdx.x * (KPP_BLOCK / 256);
        double v[KPP_BLOCK / 256], ls = 0.0;
#pragma unroll
        for (int q = 0; q < KPP_BLOCK / 256; ++q) { v[q] = (e0 + q < N) ? closest[e0 + q] : 0.0; ls += v[q]; }
        scr[threadIdx.x] = ls;
        __syncthreads();
        if (threadIdx.x == 0) {             // exclusive prefix of the 256 thread sums, sequential
            double a = pref[b];
            for (int j = 0; j < 256; ++j) { const double x = scr[j]; scr[j] = a; a += x; }
        }
        __syncthreads();
        // each thread checks its 16 elements; the smallest qualifying index wins
        double a = scr[threadIdx.x];
        long long mine = (long long)N;      // "not found"
#pragma unroll
        for (int q = 0; q < KPP_BLOCK / 256; ++q) {
            a += v[q];
            if (a >= r && e0 + q < N && mine == (long long)N) mine = (long long)(e0 + q);
        }
        __syncthreads();
        long long* imin = reinterpret_cast<long long*>(scr);
        imin[threadIdx.x] = mine;
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if ((int)threadIdx.x < off && imin[threadIdx.x + off] < imin[threadIdx.x]) imin[threadIdx.x] = imin[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            long long id = imin[0];
            if (id >= (long long)N) {       // rounding pushed r past this block's end: first element of the next block, clipped
                id = (long long)(b + 1) * KPP_BLOCK;
                if (id > (long long)N - 1) id = (long long)N - 1;
            }
            cand[t] = (int64_t)id;
        }
        __syncthreads();
    }
}

// newmin[t][i] = min(closest_i, |x_i - x_cand_t|^2) and per-block sums part[t][block].  Each point is read once and
// tested against all candidates (their coordinates sit in LDS, broadcast reads).
// KPP_DT threads per workgroup: with one workgroup per 4096 points the row-strided loads need every wave a CU can hold to cover their latency
// (256 threads: 178 us per selection step at N = 2^20; 1024: see tools/dev_kmeans.py)
static constexpr int KPP_DT = 256;
template <int DMAX>
__global__ void __launch_bounds__(KPP_DT)
kpp_dist_kernel(const double* __restrict__ X, int64_t N, int D, int64_t ldx, const int64_t* __restrict__ cand, int n_trials,
                const double* __restrict__ closest, double* __restrict__ newmin, double* __restrict__ part) {
    __shared__ double red[KPP_DT];
    __shared__ double cs[KPP_MAXT][DMAX];
    for (int idx = threadIdx.x; idx < n_trials * DMAX; idx += KPP_DT) {
        const int t = idx / DMAX, d = idx - t * DMAX;
        cs[t][d] = d < D ? X[cand[t] * ldx + d] : 0.0;
    }
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.x * KPP_BLOCK;
    double s[KPP_MAXT];
#pragma unroll
    for (int t = 0; t < KPP_MAXT; ++t) s[t] = 0.0;
    // The 256 rows of a trip come in as ONE contiguous, coalesced block and go through an LDS tile (odd pitch: conflict-free row reads);
    // read row by row per thread, every load instruction touched 64 cache lines for 8 bytes each.  Arithmetic and order unchanged.
    extern __shared__ double kpp_tile[];                     // [256][DMAX + 1]
    for (int q = 0; q < KPP_BLOCK / KPP_DT; ++q) {
        const int64_t r0 = b0 + (int64_t)q * KPP_DT;
        __syncthreads();
        const int64_t nel = (r0 + KPP_DT <= N ? KPP_DT : (r0 < N ? N - r0 : 0)) * ldx;
        for (int64_t e = threadIdx.x; e < nel; e += KPP_DT) {
            const int r = (int)(e / ldx), c = (int)(e - (int64_t)r * ldx);
            if (c < D) kpp_tile[r * (DMAX + 1) + c] = X[r0 * ldx + e];
        }
        __syncthreads();
        const int64_t i = r0 + threadIdx.x;
        if (i < N) {
            double x[DMAX];
#pragma unroll
            for (int d = 0; d < DMAX; ++d) x[d] = d < D ? kpp_tile[threadIdx.x * (DMAX + 1) + d] : 0.0;
            const double cl = closest[i];
#pragma unroll
            for (int t = 0; t < KPP_MAXT; ++t) {
                if (t < n_trials) {
                    double dist = 0.0;
#pragma unroll
                    for (int d = 0; d < DMAX; ++d) { const double u = x[d] - cs[t][d]; dist = __builtin_fma(u, u, dist); }
                    const double m = dist < cl ? dist : cl;
                    newmin[(int64_t)t * N + i] = m;
                    s[t] += m;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < KPP_MAXT; ++t) {
        if (t < n_trials) {
            const double ws = km_wave_sum(s[t]);        // fixed butterfly within the wave, then the waves in order: two barriers
            if ((threadIdx.x & 63) == 0) red[t * (KPP_DT / 64) + (threadIdx.x >> 6)] = ws;      // for ALL trials instead of eight trees of eight
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < n_trials) {
        double r = red[threadIdx.x * (KPP_DT / 64)];
        for (int w = 1; w < KPP_DT / 64; ++w) r += red[threadIdx.x * (KPP_DT / 64) + w];
        part[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = r;
    }
}

// One workgroup: potentials of the trials (fixed order), first minimum wins; records the chosen index and potential.
__global__ void __launch_bounds__(256)
kpp_select_kernel(const double* __restrict__ part, int nb, int n_trials, const int64_t* __restrict__ cand, double* __restrict__ state,
                  int64_t* __restrict__ indices, int c) {
    __shared__ double red[256];
    __shared__ double pots[KPP_MAXT];
    for (int t = 0; t < n_trials; ++t) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nb; b += 256) s += part[(int64_t)t * nb + b];
        s = kpp_block_reduce(s, red);
        if (threadIdx.x == 0) pots[t] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int t = 1; t < n_trials; ++t) if (pots[t] < pots[best]) best = t;
        state[0] = pots[best];
        state[1] = (double)best;
        indices[c] = cand[best];
    }
}

// closest <- newmin[best]; block sums of the new closest
__global__ void __launch_bounds__(256)
kpp_commit_kernel(const double* __restrict__ newmin, const double* __restrict__ part, int64_t N, const double* __restrict__ state,
                  double* __restrict__ closest, double* __restrict__ bsum) {
    const int best = (int)state[1];
    const int64_t b0 = (int64_t)blockIdx.x * KPP_BLOCK;
    for (int q = 0; q < KPP_BLOCK / 256; ++q) {
        const int64_t i = b0 + (int64_t)q * 256 + threadIdx.x;                     // coalesced copy
        if (i < N) closest[i] = newmin[(int64_t)best * N + i];
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = part[(int64_t)best * gridDim.x + blockIdx.x];
}

template <int DMAX>
static int kpp_run(oak_ctx* ctx, const double* dX, int64_t N, int D, int64_t ldx, int K, int64_t first, const double* dU, int n_trials,
                   int64_t* dIdx) {
    const int nb = (int)((N + KPP_BLOCK - 1) / KPP_BLOCK);
    double *dClosest, *dBsum, *dNewmin, *dPart, *dState;
    int64_t* dCand;
    OAK_CHECK(get_buf_t(ctx, "kpp_closest", (size_t)N, &dClosest));
    OAK_CHECK(get_buf_t(ctx, "kpp_bsum", (size_t)nb, &dBsum));
    OAK_CHECK(get_buf_t(ctx, "kpp_newmin", (size_t)N * n_trials, &dNewmin));
    OAK_CHECK(get_buf_t(ctx, "kpp_part", (size_t)nb * n_trials, &dPart));
    OAK_CHECK(get_buf_t(ctx, "kpp_state", 2, &dState));
    OAK_CHECK(get_buf_t(ctx, "kpp_cand", (size_t)KPP_MAXT, &dCand));
    kpp_init_kernel<DMAX><<<nb, 256, 0, ctx->stream>>>(dX, N, D, ldx, first, dClosest, dBsum, dIdx);
    OAK_HIP_CHECK(hipGetLastError());
    OAK_CHECK(reduce_sum(ctx, dBsum, nb, dState, 0, 1));                           // initial potential
    const size_t lds_pick = sizeof(double) * ((size_t)nb + 1 + 256);
    const size_t lds_dist = sizeof(double) * (size_t)KPP_DT * (DMAX + 1);
    if (lds_dist > 48 * 1024) OAK_CHECK(ensure_dynamic_lds((const void*)kpp_dist_kernel<DMAX>, lds_dist));      // next to ~10 KiB of static LDS
    for (int c = 1; c < K; ++c) {
        kpp_pick_kernel<<<1, 256, lds_pick, ctx->stream>>>(dClosest, dBsum, nb, N, dState, dU + (size_t)(c - 1) * n_trials, n_trials, dCand);
        kpp_dist_kernel<DMAX><<<nb, KPP_DT, lds_dist, ctx->stream>>>(dX, N, D, ldx, dCand, n_trials, dClosest, dNewmin, dPart);
        kpp_select_kernel<<<1, 256, 0, ctx->stream>>>(dPart, nb, n_trials, dCand, dState, dIdx, c);
        kpp_commit_kernel<<<nb, 256, 0, ctx->stream>>>(dNewmin, dPart, N, dState, dClosest, dBsum);
    }
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

}  // namespace oak

extern "C" int oak_kmeans_plusplus(oak_ctx* ctx, const double* X, int64_t N, int32_t D, int32_t ldx, int32_t K, int64_t first_index,
                                   const double* uniforms, int32_t n_trials, double* centres_out, int64_t* indices_out) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(X && centres_out && (uniforms || K == 1), "oak_kmeans_plusplus: NULL argument");
    OAK_REQUIRE(N >= 1 && N <= ((int64_t)1 << 24) && D >= 1 && D <= 64 && ldx >= D && K >= 1 && K <= N && first_index >= 0 &&
                    first_index < N && n_trials >= 1 && n_trials <= KPP_MAXT,
                "oak_kmeans_plusplus: bad sizes (N=%lld D=%d K=%d first=%lld trials=%d)", (long long)N, D, K, (long long)first_index,
                n_trials);
    PhaseTimer t(ctx, "kmeans_pp");
    const int dmax = D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : 64;
    double *dX, *dU;
    int64_t* dIdx;
    OAK_CHECK(get_buf_t(ctx, "km_X", (size_t)N * ldx, &dX));
    OAK_CHECK(get_buf_t(ctx, "kpp_u", (size_t)(K > 1 ? (K - 1) : 1) * n_trials, &dU));
    OAK_CHECK(get_buf_t(ctx, "kpp_idx", (size_t)K, &dIdx));
    OAK_HIP_CHECK(hipMemcpyAsync(dX, X, sizeof(double) * (size_t)N * ldx, hipMemcpyHostToDevice, ctx->stream));
    if (K > 1) OAK_HIP_CHECK(hipMemcpyAsync(dU, uniforms, sizeof(double) * (size_t)(K - 1) * n_trials, hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const int64_t first = first_index;
    OAK_CHECK(KM_DISPATCH(kpp_run, ctx, dX, N, D, ldx, K, first, dU, n_trials, dIdx));
    std::vector<int64_t> hIdx((size_t)K);
    OAK_HIP_CHECK(hipMemcpyAsync(hIdx.data(), dIdx, sizeof(int64_t) * (size_t)K, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    t.stop();
    for (int k = 0; k < K; ++k) {
        const int64_t i = hIdx[(size_t)k];
        if (i < 0 || i >= N) { set_error("oak_kmeans_plusplus: internal error, index %lld out of range", (long long)i); return OAK_E_STATE; }
        for (int d = 0; d < D; ++d) centres_out[(size_t)k * D + d] = X[(size_t)i * ldx + d];
        if (indices_out) indices_out[k] = i;
    }
    return OAK_OK;
}

// Cholesky, triangular solves and a general fp64 MFMA GEMM: the O(M^3) "tail" of the SGPR objective
// (replaces tf.linalg.cholesky / triangular_solve / matmul at oak/utils.py:187-198).
//
// The tail is latency-bound at M <= 2048, so the design goal is few, wide launches:
//   * gemm_mfma  : C = beta*C + alpha*A*op(B), 64x64 tiles, v_mfma_f64_16x16x4, LDS pitches chosen conflict-free.
//   * potrf_lower: right-looking, NB = 32, ONE fused launch per panel (potrf_step_kernel): the diagonal block is factored
//                  by one wave (lane = row, rows in registers, finished entries mirrored to LDS for broadcast reads, no
//                  barriers) while the other waves apply the previous panel's pending update to their rows, which are then
//                  solved one per lane; the remaining workgroups of the launch update the trailing matrix with MFMA
//                  tiles.  Every launch costs >= 5 us on this system, so the tail is launch-count bound.
//   * trsm_rows  : NB = 128 blocked: in-LDS leaf solves + MFMA GEMM updates; with >= 8192 right-hand sides left-looking,
//                  the diagonal blocks inverted once and applied as GEMMs.
#include "oak_internal.h"

namespace oak {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// general GEMM (row-major):  BT = 1: C[m x n] = beta*C + alpha * A[m x k] * B[n x k]^T
//                            BT = 0: C[m x n] = beta*C + alpha * A[m x k] * B[k x n]
// lower_only: skip 64x64 tiles strictly above the diagonal (symmetric rank-k updates).
// ---------------------------------------------------------------------------------------------
constexpr int GM_T = 64, GM_K = 32, GM_PA = GM_K + 2, GM_PB = GM_T + 16;

template <int BT>
__global__ void __launch_bounds__(256) gemm_mfma_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                        double* __restrict__ C, int64_t m, int64_t n, int64_t k, int64_t lda,
                                                        int64_t ldb, int64_t ldc, double alpha, double beta, int lower_only) {
    __shared__ __attribute__((aligned(16))) double As[GM_T * GM_PA];
    __shared__ __attribute__((aligned(16))) double Bs[(BT ? GM_T * GM_PA : GM_K * GM_PB)];
    if (lower_only && blockIdx.x > blockIdx.y) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t r0 = (int64_t)blockIdx.y * GM_T, c0 = (int64_t)blockIdx.x * GM_T;
    const int fi = lane & 15, fk = lane >> 4;
    double4_t acc[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[g][h] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // staging coordinates: 64 rows x 32 k per tile, 8 doubles per thread
    const int ar = tid >> 2, ak = (tid & 3) * 8;             // A (and B when BT): row 0..63, k offset 0,8,16,24
    const int bk = tid >> 3, bn = (tid & 7) * 8;             // B when !BT: k row 0..31, n offset 0..56
    const int64_t arow = (r0 + ar < m) ? r0 + ar : m - 1;
    const int64_t brow = BT ? ((c0 + ar < n) ? c0 + ar : n - 1) : 0;
    double va[8], vb[8];
    // branch-free (clamped) loads so that all 16 stay in flight; values outside the matrix are zeroed by selects
    auto load_stage = [&](int64_t k0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t kk = k0 + ak + q;
            const int64_t kc = kk < k ? kk : k - 1;
            const double x = A[arow * lda + kc];
            va[q] = (kk < k && r0 + ar < m) ? x : 0.0;
            if (BT) {
                const double y = B[brow * ldb + kc];
                vb[q] = (kk < k && c0 + ar < n) ? y : 0.0;
            } else {
                const int64_t kr = k0 + bk, nc = c0 + bn + q;
                const double y = B[(kr < k ? kr : k - 1) * ldb + (nc < n ? nc : n - 1)];
                vb[q] = (kr < k && nc < n) ? y : 0.0;
            }
        }
    };
    load_stage(0);
    for (int64_t k0 = 0; k0 < k; k0 += GM_K) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            As[ar * GM_PA + ak + q] = va[q];
            if (BT) Bs[ar * GM_PA + ak + q] = vb[q];
            else Bs[bk * GM_PB + bn + q] = vb[q];
        }
        __syncthreads();
        if (k0 + GM_K < k) load_stage(k0 + GM_K);            // register prefetch of the next K chunk under the MFMAs
#pragma unroll
        for (int ks = 0; ks < GM_K / 4; ++ks) {
            double a[2], b[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) a[g] = As[(wr * 32 + 16 * g + fi) * GM_PA + 4 * ks + fk];
#pragma unroll
            for (int h = 0; h < 2; ++h)
                b[h] = BT ? Bs[(wc * 32 + 16 * h + fi) * GM_PA + 4 * ks + fk] : Bs[(4 * ks + fk) * GM_PB + wc * 32 + 16 * h + fi];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[g][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g], b[h], acc[g][h], 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t row = r0 + wr * 32 + 16 * g + 4 * reg + fk, col = c0 + wc * 32 + 16 * h + fi;
                if (row < m && col < n) {
                    double* q = C + row * ldc + col;
                    const double v = alpha * acc[g][h][reg];
                    *q = (beta == 0.0) ? v : __builtin_fma(beta, *q, v);
                }
            }
}

// ---------------------------------------------------------------------------------------------
// Large NT GEMM for the adjoint panel  Gfu = Kfu H  (N x M x M, the dominant kernel of the backward pass):
// 128x128 tiles, 4 waves of 64x64, K-chunks of 16 staged through LDS with a register prefetch one stage ahead
// (branch-free, clamped loads -- see syrk_kernel), LDS pitch 18 doubles -> conflict-free ds_read_b64 fragments.
// Workgroups that share a row block of A are mapped to the same XCD (b % 8) so the A tile is fetched into one L2.
// Requirements: lda, ldb even and A, B 16-byte aligned (the caller falls back to the 64x64 kernel otherwise).
// ---------------------------------------------------------------------------------------------
constexpr int G2_T = 128, G2_K = 16, G2_P = G2_K + 2;
constexpr int G2_LPR = G2_K / 2;            // lanes per row chunk (16 B each)
constexpr int G2_RPL = 64 / G2_LPR;          // rows per wave-load
constexpr int G2_NQ = G2_T / (4 * G2_RPL);   // loads per thread per matrix

__global__ void __launch_bounds__(256, 2)
gemm128_nt_kernel(const double* A, const double* __restrict__ B, double* C /* may alias A: trsm_rows runs the diagonal-block product in place */, int64_t m, int64_t n,
                  int64_t k, int64_t lda, int64_t ldb, int64_t ldc, double alpha, double beta, int ntn) {
    __shared__ __attribute__((aligned(16))) double As[G2_T * G2_P];
    __shared__ __attribute__((aligned(16))) double Bs[G2_T * G2_P];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile mapping: the ntn column tiles of one row block run on the same XCD
    const int64_t grp_size = 8LL * ntn;
    const int64_t grp = blockIdx.x / grp_size, within = blockIdx.x - grp * grp_size;
    const int64_t rb = grp * 8 + (within & 7);
    const int64_t cb = within >> 3;
    const int64_t r0 = rb * G2_T, c0 = cb * G2_T;
    if (r0 >= m) return;
    const int fi = lane & 15, fk = lane >> 4;
    double4_t acc[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[g][h] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int lr = lane / G2_LPR, lc = (lane % G2_LPR) * 2;
    double2 ra[G2_NQ], rbv[G2_NQ];
    auto load_stage = [&](int64_t k0) {
#pragma unroll
        for (int q = 0; q < G2_NQ; ++q) {
            const int row = (wave * G2_NQ + q) * G2_RPL + lr;
            const int64_t ar = (r0 + row < m) ? r0 + row : m - 1;
            const int64_t br = (c0 + row < n) ? c0 + row : n - 1;
            const int64_t kc = (k0 + lc + 1 < k) ? k0 + lc : ((k >= 2) ? k - 2 : 0);
            const double2 va = *reinterpret_cast<const double2*>(A + ar * lda + kc);
            const double2 vb = *reinterpret_cast<const double2*>(B + br * ldb + kc);
            const bool okk = k0 + lc + 1 < k;
            const bool oka = okk && (r0 + row < m), okb = okk && (c0 + row < n);
            ra[q] = make_double2(oka ? va.x : 0.0, oka ? va.y : 0.0);
            rbv[q] = make_double2(okb ? vb.x : 0.0, okb ? vb.y : 0.0);
        }
    };
    load_stage(0);
    for (int64_t k0 = 0; k0 < k; k0 += G2_K) {
#pragma unroll
        for (int q = 0; q < G2_NQ; ++q) {
            const int row = (wave * G2_NQ + q) * G2_RPL + lr;
            *reinterpret_cast<double2*>(&As[row * G2_P + lc]) = ra[q];
            *reinterpret_cast<double2*>(&Bs[row * G2_P + lc]) = rbv[q];
        }
        __syncthreads();
        load_stage((k0 + G2_K < k) ? k0 + G2_K : 0);
#pragma unroll
        for (int ks = 0; ks < G2_K / 4; ++ks) {
            double a[4], b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g] = As[(64 * wr + 16 * g + fi) * G2_P + 4 * ks + fk];
#pragma unroll
            for (int h = 0; h < 4; ++h) b[h] = Bs[(64 * wc + 16 * h + fi) * G2_P + 4 * ks + fk];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int h = 0; h < 4; ++h) acc[g][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g], b[h], acc[g][h], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t row = r0 + 64 * wr + 16 * g + 4 * reg + fk, col = c0 + 64 * wc + 16 * h + fi;
                if (row < m && col < n) {
                    double* q = C + row * ldc + col;
                    const double v = alpha * acc[g][h][reg];
                    *q = (beta == 0.0) ? v : __builtin_fma(beta, *q, v);
                }
            }
}

static int gemm_launch(oak_ctx* ctx, int bt, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k,
                       int64_t lda, int64_t ldb, int64_t ldc, double alpha, double beta, int lower_only) {
    if (m <= 0 || n <= 0) return OAK_OK;
    dim3 grid((unsigned)((n + GM_T - 1) / GM_T), (unsigned)((m + GM_T - 1) / GM_T));
    if (bt) gemm_mfma_kernel<1><<<grid, 256, 0, ctx->stream>>>(dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, lower_only);
    else    gemm_mfma_kernel<0><<<grid, 256, 0, ctx->stream>>>(dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, lower_only);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}
int gemm_nn(oak_ctx* ctx, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k, int64_t lda,
            int64_t ldb, int64_t ldc, double alpha, double beta) {
    return gemm_launch(ctx, 0, dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, 0);
}
// true when gemm_nt takes the 128 x 128 kernel (one workgroup per output tile, every operand read before the tile is stored)
static bool gemm128_eligible(const double* dA, const double* dB, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int lower_only) {
    const bool aligned = ((lda | ldb) & 1) == 0 && (((uintptr_t)dA | (uintptr_t)dB) & 15) == 0 && k >= 2 && (k & 1) == 0;
    return !lower_only && aligned && m >= 2048 && n >= 128 && m > 0 && n > 0;
}
int gemm_nt(oak_ctx* ctx, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k, int64_t lda,
            int64_t ldb, int64_t ldc, double alpha, double beta, int lower_only) {
    if (gemm128_eligible(dA, dB, m, n, k, lda, ldb, lower_only)) {
        const int ntn = (int)((n + G2_T - 1) / G2_T);
        const int64_t nrb = (m + G2_T - 1) / G2_T;
        const int64_t ngrp = (nrb + 7) / 8;
        gemm128_nt_kernel<<<(unsigned)(ngrp * 8 * ntn), 256, 0, ctx->stream>>>(dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, ntn);
        OAK_HIP_CHECK(hipGetLastError());
        return OAK_OK;
    }
    return gemm_launch(ctx, 1, dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, lower_only);
}

// ---------------------------------------------------------------------------------------------
// Cholesky (lower), right-looking
// ---------------------------------------------------------------------------------------------
constexpr int PO_NB = 32;          // panel width (32 and 64 measure the same end-to-end: fewer launches vs a slower one-wave factorisation)

// 1/sqrt(d) to full fp64 accuracy: hardware estimate + two Newton steps (cheaper than sqrt followed by a divide)
__device__ __forceinline__ double rsqrt_newton(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
    y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
    return y;
}

constexpr int PO_P = PO_NB + 2;    // LDS pitch of the diagonal block (even: 16-byte aligned row starts)

// One wave factors the NB x NB diagonal block (Cholesky-Crout, column by column).  Lane i owns row i in registers and
// mirrors each finished entry into LDS, so row j is available to every lane as broadcast ds_read_b128s: no barriers,
// no cross-lane register traffic except one v_readlane of the pivot per column.  invd[j] = 1 / L_jj is kept for the
// panel solve.  Two partial sums halve the dependent FMA chain of each column.
__device__ __forceinline__ void potf2_wave(double* Dg /*[NB][PO_P]*/, double* invd /*[NB]*/, int lane, int64_t j0, int* info) {
    const int i = lane % PO_NB;                 // with NB = 32 lanes 32..63 shadow lanes 0..31 and write nothing
    const bool writer = lane < PO_NB;
    double a[PO_NB];
#pragma unroll
    for (int c = 0; c < PO_NB; ++c) a[c] = Dg[i * PO_P + c];
    int bad_at = -1;
#pragma unroll
    for (int j = 0; j < PO_NB; ++j) {
        double s0 = a[j], s1 = 0.0;
#pragma unroll
        for (int k = 0; k + 1 < j; k += 2) {
            const double2 l2 = *reinterpret_cast<const double2*>(&Dg[j * PO_P + k]);
            s0 = __builtin_fma(-a[k], l2.x, s0);
            s1 = __builtin_fma(-a[k + 1], l2.y, s1);
        }
        if (j & 1) s0 = __builtin_fma(-a[j - 1], Dg[j * PO_P + j - 1], s0);
        const double s = s0 + s1;
        const double d = readlane_f64(s, j);
        if (!(d > 0.0) && bad_at < 0) bad_at = j;
        const double r = rsqrt_newton(d);
        a[j] = (i == j) ? d * r : s * r;
        if (writer && i >= j) Dg[i * PO_P + j] = a[j];
        if (lane == j) invd[j] = r;
    }
#pragma unroll
    for (int c = 0; c < PO_NB; ++c)
        if (writer && c > i) Dg[i * PO_P + c] = 0.0;
    if (bad_at >= 0 && lane == 0 && info != nullptr) atomicMin(info, (int)(j0 + bad_at + 1));
}

// Every role-A workgroup re-factors the (tiny) diagonal block itself instead of waiting for one producer.  The factor must
// not be written back over A_jj while a sibling workgroup may still be loading the unfactored block, so the LAST workgroup
// to finish loading (arrival counter, one per panel) does the write-back.
// ---------------------------------------------------------------------------------------------------------------------
// Fused right-looking step: ONE launch per 32-column panel instead of two (panel kernel + trailing GEMM).
//   role A (first nA workgroups, 256 rows each): panel j.  Its columns still lack the rank-32 update of panel j-1 (all
//           earlier panels were applied by role B of earlier launches); the workgroup applies it to its own rows and --
//           redundantly, 32 x 32 x 32 -- to the diagonal block, then factors the block and solves its rows as before.
//   role B (remaining workgroups, one 64 x 64 lower tile each): the rank-32 update of panel j-1 on the trailing matrix
//           BEHIND panel j (columns >= j0 + 32), MFMA.
// The two roles touch disjoint columns and only read panel j-1, so there is no dependency inside a launch; the dependent
// chain of the factorisation is 32 launches of ~max(role A, role B) instead of 64 launches.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(320) potrf_step_kernel(double* __restrict__ A, int64_t n, int64_t nrows, int64_t lda, int64_t j0,
                                                         int* __restrict__ info, int* __restrict__ arrivals, int nA, int ntc, int pending) {
    __shared__ __attribute__((aligned(16))) double Dg[PO_NB * PO_P];
    __shared__ __attribute__((aligned(16))) double Pj[64 * PO_P];       // role A: rows j0.. of panel j-1 (32 used); role B: A-side tile
    __shared__ __attribute__((aligned(16))) double Pc[64 * PO_P];       // role B: B-side tile
    __shared__ double invd[PO_NB];
    __shared__ int last_loader;
    const int tid = threadIdx.x;
    const int64_t p0 = j0 - PO_NB;                                      // previous panel's first column (valid when j0 > 0)
    if ((int)blockIdx.x >= nA) {
        // ---------------- role B: C[r0:+64, c0:+64] -= P[r0:+64, :] P[c0:+64, :]^T,  P = A[:, p0:p0+32] ----------------
        const int t = blockIdx.x - nA;
        const int by = t / ntc, bx = t - by * ntc;
        if (bx > by || tid >= 256) return;                              // the fifth wave only exists for role A
        const int64_t s0 = j0 + PO_NB;
        const int64_t r0 = s0 + 64 * (int64_t)by, c0 = s0 + 64 * (int64_t)bx;
        {   // stage both 64 x 32 operand tiles (clamped rows; rows past the matrix contribute to outputs that are not stored)
            const int r = tid >> 2, k8 = (tid & 3) * 8;
            const int64_t ra = (r0 + r < nrows) ? r0 + r : nrows - 1, rb = (c0 + r < n) ? c0 + r : n - 1;
            double va[8], vb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { va[q] = A[ra * lda + p0 + k8 + q]; vb[q] = A[rb * lda + p0 + k8 + q]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) { Pj[r * PO_P + k8 + q] = va[q]; Pc[r * PO_P + k8 + q] = vb[q]; }
        }
        __syncthreads();
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wr = wave >> 1, wc = wave & 1, fi = lane & 15, fk = lane >> 4;
        double4_t acc[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[g][h] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < PO_NB / 4; ++ks) {
            double a[2], b[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) a[g] = Pj[(wr * 32 + 16 * g + fi) * PO_P + 4 * ks + fk];
#pragma unroll
            for (int h = 0; h < 2; ++h) b[h] = Pc[(wc * 32 + 16 * h + fi) * PO_P + 4 * ks + fk];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[g][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g], b[h], acc[g][h], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int64_t row = r0 + wr * 32 + 16 * g + 4 * reg + fk, col = c0 + wc * 32 + 16 * h + fi;
                    if (row < nrows && col < n) A[row * lda + col] -= acc[g][h][reg];
                }
        return;
    }
    // ---------------- role A: panel j ----------------
    // 320 threads: wave 0 factors the diagonal block while the 256 workers (waves 1..4, one row each) apply the pending
    // update to their rows -- the two halves of the step's critical path overlap.
    const int wid = tid - 64;                                           // worker id, < 0 for the factor wave
    const bool worker = wid >= 0;
    const int nb = (n - j0 < PO_NB) ? (int)(n - j0) : PO_NB;
    const bool pre = j0 > 0 && pending;                                 // pending = 0: the caller already applied panel j-1
    if (pre && worker) {   // rows j0 .. j0+31 of panel j-1
        const int r = wid >> 3, k4 = (wid & 7) * 4;
        const int64_t rr = (j0 + r < n) ? j0 + r : n - 1;
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = A[rr * lda + p0 + k4 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) Pj[r * PO_P + k4 + q] = (j0 + r < n) ? v[q] : 0.0;
    }
    __syncthreads();
    if (worker) {
        // NB*NB entries of the diagonal block, clamped addresses; identity padding beyond nb; pending update applied
        constexpr int PER = PO_NB * PO_NB / 256;
        double v[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = wid + 256 * q;
            const int i = idx / PO_NB, j = idx % PO_NB;
            const int ic = i < nb ? i : nb - 1, jc = j < nb ? j : nb - 1;
            double x = A[(j0 + ic) * lda + j0 + jc];
            if (pre) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < PO_NB; k += 2) {
                    const double2 a2 = *reinterpret_cast<const double2*>(&Pj[ic * PO_P + k]);
                    const double2 b2 = *reinterpret_cast<const double2*>(&Pj[jc * PO_P + k]);
                    s = __builtin_fma(a2.x, b2.x, s);
                    s = __builtin_fma(a2.y, b2.y, s);
                }
                x -= s;
            }
            v[q] = (i < nb && j < nb) ? ((j <= i) ? x : 0.0) : ((i == j) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) { const int idx = wid + 256 * q; Dg[(idx / PO_NB) * PO_P + (idx % PO_NB)] = v[q]; }
    }
    __syncthreads();
    if (tid == 319) last_loader = (atomicAdd(arrivals, 1) == nA - 1);
    const int64_t row = j0 + nb + (int64_t)blockIdx.x * 256 + wid;
    const bool has_row = worker && row < nrows && nb == PO_NB;
    double x[PO_NB];
    if (!worker) {
        potf2_wave(Dg, invd, tid, j0, blockIdx.x == 0 ? info : nullptr);
    } else if (has_row) {
        // this row of panel j with the pending update of panel j-1 applied (reads its own row and LDS only)
        const double* ap = A + row * lda + j0;
#pragma unroll
        for (int c = 0; c < PO_NB; ++c) x[c] = ap[c];
        if (pre) {
            double pr[PO_NB];
            const double* pp = A + row * lda + p0;
#pragma unroll
            for (int k = 0; k < PO_NB; ++k) pr[k] = pp[k];
#pragma unroll
            for (int c = 0; c < PO_NB; ++c) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int k = 0; k < PO_NB; k += 2) {
                    const double2 l2 = *reinterpret_cast<const double2*>(&Pj[c * PO_P + k]);
                    s0 = __builtin_fma(pr[k], l2.x, s0);
                    s1 = __builtin_fma(pr[k + 1], l2.y, s1);
                }
                x[c] -= s0 + s1;
            }
        }
    }
    __syncthreads();
    if (has_row) {   // X * L_jj^T = A_panel, one row per lane
#pragma unroll
        for (int c = 0; c < PO_NB; ++c) {
            double s0 = x[c], s1 = 0.0;
#pragma unroll
            for (int p = 0; p + 1 < c; p += 2) {
                const double2 l2 = *reinterpret_cast<const double2*>(&Dg[c * PO_P + p]);
                s0 = __builtin_fma(-x[p], l2.x, s0);
                s1 = __builtin_fma(-x[p + 1], l2.y, s1);
            }
            if (c & 1) s0 = __builtin_fma(-x[c - 1], Dg[c * PO_P + c - 1], s0);
            x[c] = (s0 + s1) * invd[c];
        }
        double* ap = A + row * lda + j0;
#pragma unroll
        for (int c = 0; c < PO_NB; ++c) ap[c] = x[c];
    }
    if (last_loader && worker) {
        for (int idx = wid; idx < nb * nb; idx += 256) {
            const int i = idx / nb, j = idx - i * nb;
            A[(j0 + i) * lda + j0 + j] = Dg[i * PO_P + j];
        }
    }
}

__global__ void zero_upper_kernel(double* __restrict__ A, int64_t n, int64_t lda) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j < n && j > i) A[i * lda + j] = 0.0;
}

static const int PO_BIG = 0x7fffffff;
__global__ void set_int_kernel(int* p, int v) { *p = v; }

int potrf_check(oak_ctx* ctx, int slot, int64_t n) {
    int* d_info = (int*)peek_buf(ctx, "potrf_info");
    int info = 0;
    OAK_HIP_CHECK(hipMemcpyAsync(&info, d_info + slot, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (info != PO_BIG) {
        set_error("Cholesky decomposition was not successful: leading minor of order %d is not positive definite (n=%lld)", info, (long long)n);
        return OAK_E_NOTPD;
    }
    return OAK_OK;
}

int potrf_lower(oak_ctx* ctx, double* dA, int64_t n, int64_t lda, bool check, int64_t nrows) {
    if (nrows < n) nrows = n;
    if (nrows > n && (n % PO_NB) != 0) { set_error("potrf_lower: extra rows need n to be a multiple of %d", PO_NB); return OAK_E_ARG; }
    int* d_info = nullptr;
    OAK_CHECK(get_buf_t(ctx, "potrf_info", 2, &d_info));
    const int slot = (ctx->side != nullptr && ctx->stream == ctx->side) ? 1 : 0;
    d_info += slot;
    set_int_kernel<<<1, 1, 0, ctx->stream>>>(d_info, PO_BIG);
    int* d_arr = nullptr;
    const size_t npanel = (size_t)((n + PO_NB - 1) / PO_NB);
    OAK_CHECK(get_buf_t(ctx, slot ? "potrf_arrivals_side" : "potrf_arrivals", npanel, &d_arr));
    OAK_HIP_CHECK(hipMemsetAsync(d_arr, 0, sizeof(int) * npanel, ctx->stream));
    // Fused mode (one launch per panel) is launch-latency optimal and wins up to n ~ 6000; beyond that the trailing updates
    // are HBM-bound and the pipelined 64 x 64 GEMM kernel moves them faster than the single-stage role-B tiles
    // (n = 16384: 253 ms fused vs 182 ms), so large factorisations fall back to panel + GEMM per step.
    const bool fused = n <= 6144;
    for (int64_t j0 = 0; j0 < n; j0 += PO_NB) {
        const int64_t below_rows = nrows - j0 - PO_NB;     // rows under the diagonal block (extra rows included)
        const int nA = below_rows > 0 ? (int)((below_rows + 255) / 256) : 1;
        const int64_t tc = n - j0 - PO_NB, tr = nrows - j0 - PO_NB;
        if (fused) {
            // role B applies panel j-1 to the trailing matrix behind panel j: columns >= j0 + 32 (none on the first step)
            const int ntc = (j0 > 0 && tc > 0) ? (int)((tc + 63) / 64) : 0;
            const int ntr = (j0 > 0 && tc > 0) ? (int)((tr + 63) / 64) : 0;
            potrf_step_kernel<<<(unsigned)(nA + ntr * ntc), 320, 0, ctx->stream>>>(dA, n, nrows, lda, j0, d_info, d_arr + j0 / PO_NB, nA,
                                                                                     ntc > 0 ? ntc : 1, 1);
        } else {
            potrf_step_kernel<<<(unsigned)nA, 320, 0, ctx->stream>>>(dA, n, nrows, lda, j0, d_info, d_arr + j0 / PO_NB, nA, 1, 0);
            if (tc > 0) {   // trailing update A22 -= L21 L21^T (lower tiles only), K = 32
                const double* L21 = dA + (j0 + PO_NB) * lda + j0;
                double* A22 = dA + (j0 + PO_NB) * lda + (j0 + PO_NB);
                OAK_CHECK(gemm_nt(ctx, L21, L21, A22, tr, tc, PO_NB, lda, lda, lda, -1.0, 1.0, 1));
            }
        }
    }
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
    zero_upper_kernel<<<grid, 256, 0, ctx->stream>>>(dA, n, lda);
    OAK_HIP_CHECK(hipGetLastError());
    return check ? potrf_check(ctx, slot, n) : OAK_OK;
}

// ---------------------------------------------------------------------------------------------
// rows-TRSM leaf: n <= 256.  Each right-hand side is a contiguous row of BT; a group of 32 lanes owns one rhs.
// ---------------------------------------------------------------------------------------------
template <int TRANS>
__global__ void __launch_bounds__(256) trsm_leaf_kernel(const double* __restrict__ L, int64_t n, int64_t ldl,
                                                        double* __restrict__ BT, int64_t nrhs, int64_t ldb, int nblk, int rb) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Lt = sm;                       // [32][33]
    double* xs = sm + 32 * 33;             // [rb][nblk*32]
    const int tid = threadIdx.x;           // 256 threads always; groups >= rb only help staging
    const int rr = tid >> 5, i = tid & 31;
    const bool worker = rr < rb;
    const int64_t npad = (int64_t)nblk * 32;
    const int64_t rhs = (int64_t)blockIdx.x * rb + rr;
    const bool valid = worker && rhs < nrhs;
    double* xr = xs + (int64_t)(worker ? rr : 0) * npad;
    if (worker)
        for (int64_t c = i; c < npad; c += 32) xr[c] = (valid && c < n) ? BT[rhs * ldb + c] : 0.0;
    auto load_block = [&](int rbk, int cbk) {   // Lt[a][b] = L[rbk*32+a][cbk*32+b]; identity padding past n
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q;
            const int a = idx >> 5, b = idx & 31;
            const int64_t gr = (int64_t)rbk * 32 + a, gc = (int64_t)cbk * 32 + b;
            const double x = L[(gr < n ? gr : n - 1) * ldl + (gc < n ? gc : n - 1)];
            v[q] = (gr < n && gc < n) ? ((gc <= gr) ? x : 0.0) : ((gr == gc) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int idx = tid + 256 * q; Lt[(idx >> 5) * 33 + (idx & 31)] = v[q]; }
    };
    if (TRANS == 0) {
        for (int jb = 0; jb < nblk; ++jb) {
            __syncthreads();
            double s = xr[jb * 32 + i];
            for (int kb = 0; kb < jb; ++kb) {
                __syncthreads();
                load_block(jb, kb);
                __syncthreads();
                const double* xk = xr + kb * 32;
#pragma unroll
                for (int k2 = 0; k2 < 32; ++k2) s = __builtin_fma(-Lt[i * 33 + k2], xk[k2], s);
            }
            __syncthreads();
            load_block(jb, jb);
            __syncthreads();
            double x = 0.0;
            const double inv = 1.0 / Lt[i * 33 + i];        // one division per lane per block; pivots are scaled by it
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                const double xp = __shfl(s * inv, p, 32);
                if (i == p) x = xp;
                if (i > p) s = __builtin_fma(-Lt[i * 33 + p], xp, s);
            }
            if (worker) xr[jb * 32 + i] = x;
        }
    } else {
        for (int jb = nblk - 1; jb >= 0; --jb) {
            __syncthreads();
            double s = xr[jb * 32 + i];
            for (int kb = nblk - 1; kb > jb; --kb) {
                __syncthreads();
                load_block(kb, jb);
                __syncthreads();
                const double* xk = xr + kb * 32;
#pragma unroll
                for (int k2 = 0; k2 < 32; ++k2) s = __builtin_fma(-Lt[k2 * 33 + i], xk[k2], s);
            }
            __syncthreads();
            load_block(jb, jb);
            __syncthreads();
            double x = 0.0;
            const double inv = 1.0 / Lt[i * 33 + i];
#pragma unroll
            for (int p = 31; p >= 0; --p) {
                const double xp = __shfl(s * inv, p, 32);
                if (i == p) x = xp;
                if (i < p) s = __builtin_fma(-Lt[p * 33 + i], xp, s);
            }
            if (worker) xr[jb * 32 + i] = x;
        }
    }
    __syncthreads();
    if (valid)
        for (int64_t c = i; c < n; c += 32) BT[rhs * ldb + c] = xr[c];
}

// Same forward substitution for MANY right-hand sides (the N-sized whitening of the Kuf panel): a group of 32 lanes owns
// RPG right-hand sides at once, so every L element fetched from LDS feeds RPG FMAs and the RPG dependent pivot chains
// interleave (the one-rhs kernel above is latency-bound: 2.8 TFLOP/s at nrhs = 2^20).
template <int RPG>
__global__ void __launch_bounds__(256) trsm_leaf_many_kernel(const double* __restrict__ L, int64_t n, int64_t ldl,
                                                             double* __restrict__ BT, int64_t nrhs, int64_t ldb, int nblk) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Lt = sm;                       // [32][33]
    double* xs = sm + 32 * 33;             // [8 * RPG][nblk*32]
    const int tid = threadIdx.x;
    const int rr = tid >> 5, i = tid & 31;
    const int64_t npad = (int64_t)nblk * 32;
    const int64_t rhs0 = ((int64_t)blockIdx.x * 8 + rr) * RPG;
    double* xr = xs + (int64_t)rr * RPG * npad;
#pragma unroll
    for (int q = 0; q < RPG; ++q)
        for (int64_t c = i; c < npad; c += 32) xr[q * npad + c] = (rhs0 + q < nrhs && c < n) ? BT[(rhs0 + q) * ldb + c] : 0.0;
    auto load_block = [&](int rbk, int cbk) {   // Lt[a][b] = L[rbk*32+a][cbk*32+b]; identity padding past n
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q;
            const int a = idx >> 5, b = idx & 31;
            const int64_t gr = (int64_t)rbk * 32 + a, gc = (int64_t)cbk * 32 + b;
            const double x = L[(gr < n ? gr : n - 1) * ldl + (gc < n ? gc : n - 1)];
            v[q] = (gr < n && gc < n) ? ((gc <= gr) ? x : 0.0) : ((gr == gc) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int idx = tid + 256 * q; Lt[(idx >> 5) * 33 + (idx & 31)] = v[q]; }
    };
    for (int jb = 0; jb < nblk; ++jb) {
        __syncthreads();
        double s[RPG];
#pragma unroll
        for (int q = 0; q < RPG; ++q) s[q] = xr[q * npad + jb * 32 + i];
        for (int kb = 0; kb < jb; ++kb) {
            __syncthreads();
            load_block(jb, kb);
            __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < 32; ++k2) {
                const double l = Lt[i * 33 + k2];
#pragma unroll
                for (int q = 0; q < RPG; ++q) s[q] = __builtin_fma(-l, xr[q * npad + kb * 32 + k2], s[q]);
            }
        }
        __syncthreads();
        load_block(jb, jb);
        __syncthreads();
        double x[RPG];
#pragma unroll
        for (int q = 0; q < RPG; ++q) x[q] = 0.0;
        const double inv = 1.0 / Lt[i * 33 + i];
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const double lp = Lt[i * 33 + p];
#pragma unroll
            for (int q = 0; q < RPG; ++q) {
                const double xp = __shfl(s[q] * inv, p, 32);
                if (i == p) x[q] = xp;
                if (i > p) s[q] = __builtin_fma(-lp, xp, s[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < RPG; ++q) xr[q * npad + jb * 32 + i] = x[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPG; ++q)
        if (rhs0 + q < nrhs)
            for (int64_t c = i; c < n; c += 32) BT[(rhs0 + q) * ldb + c] = xr[q * npad + c];
}

static int trsm_leaf(oak_ctx* ctx, const double* dL, int64_t n, int64_t ldl, double* dBT, int64_t nrhs, int64_t ldb, int trans) {
    const int nblk = (int)((n + 31) / 32);
    if (!trans && nrhs >= 4096) {
        constexpr int RPG = 4;
        const size_t lds = sizeof(double) * ((size_t)8 * RPG * nblk * 32 + 32 * 33);
        const unsigned grid = (unsigned)((nrhs + 8 * RPG - 1) / (8 * RPG));
        auto kern = trsm_leaf_many_kernel<RPG>;
        if (lds > 64 * 1024) OAK_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<grid, 256, lds, ctx->stream>>>(dL, n, ldl, dBT, nrhs, ldb, nblk);
        OAK_HIP_CHECK(hipGetLastError());
        return OAK_OK;
    }
    int rb = 8;
    while (rb > 1 && rb / 2 >= nrhs) rb >>= 1;
    const size_t lds = sizeof(double) * ((size_t)rb * nblk * 32 + 32 * 33);
    const unsigned grid = (unsigned)((nrhs + rb - 1) / rb);
    if (trans) trsm_leaf_kernel<1><<<grid, 256, lds, ctx->stream>>>(dL, n, ldl, dBT, nrhs, ldb, nblk, rb);
    else       trsm_leaf_kernel<0><<<grid, 256, lds, ctx->stream>>>(dL, n, ldl, dBT, nrhs, ldb, nblk, rb);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// Blocked rows-TRSM: every row b of BT is replaced by the solution of L x = b (trans = 0) or L^T x = b (trans = 1).
int trsm_rows(oak_ctx* ctx, const double* dL, int64_t n, int64_t ldl, double* dBT, int64_t nrhs, int64_t ldb, int trans) {
    if (n <= 0 || nrhs <= 0) return OAK_OK;
    constexpr int64_t NB = 128;
    if (n <= 2 * NB) return trsm_leaf(ctx, dL, n, ldl, dBT, nrhs, ldb, trans);
    if (!trans) {
        // Many right-hand sides (the N-sized whitening of the panel): the 128 x 128 diagonal solves dominate when done by
        // substitution on the vector pipe.  Invert each diagonal block once (substitution against I: exact to the block's
        // own conditioning) and apply it as an MFMA GEMM -- the usual blocked-TRSM formulation of GPU BLAS libraries; the
        // coupling between blocks is still eliminated block by block.
        const bool by_inverse = nrhs >= 8192;   // below this the substitution leaf wins (measured at nrhs = 2048..4096)
        double *dInvT = nullptr, *dInv = nullptr, *dTmp = nullptr;
        const int64_t nfull = n / NB;
        if (by_inverse && nfull > 0) {
            OAK_CHECK(get_buf_t(ctx, "trsm_invT", (size_t)nfull * NB * NB, &dInvT));
            OAK_CHECK(get_buf_t(ctx, "trsm_inv", (size_t)nfull * NB * NB, &dInv));
            OAK_CHECK(get_buf_t(ctx, "trsm_tmp", (size_t)nrhs * NB, &dTmp));
            for (int64_t b = 0; b < nfull; ++b) {
                double* it = dInvT + b * NB * NB;
                OAK_CHECK(set_identity(ctx, it, NB));
                OAK_CHECK(trsm_leaf(ctx, dL + (b * NB) * ldl + b * NB, NB, ldl, it, NB, NB, 0));   // rows = columns of L_bb^-1
                OAK_CHECK(transpose(ctx, it, NB, NB, NB, dInv + b * NB * NB, NB));
            }
        }
        if (by_inverse) {
            // LEFT-looking: block j first receives all earlier blocks' contributions in ONE GEMM with K = j0 (reads the
            // finished columns once, writes 128 columns), then its diagonal solve.  The right-looking order re-reads and
            // re-writes every trailing column at each step with K = 128: twice the HBM traffic for the same flops.
            for (int64_t j0 = 0; j0 < n; j0 += NB) {
                const int64_t nbj = (n - j0 < NB) ? n - j0 : NB;
                if (j0 > 0)     // B[:, j] -= X[:, 0:j0] L[j, 0:j0]^T
                    OAK_CHECK(gemm_nt(ctx, dBT, dL + j0 * ldl, dBT + j0, nrhs, nbj, j0, ldb, ldl, ldb, -1.0, 1.0, 0));
                if (nbj == NB) {
                    const double* inv = dInv + (j0 / NB) * NB * NB;
                    if (gemm128_eligible(dBT + j0, inv, nrhs, NB, NB, ldb, NB, 0)) {
                        // IN PLACE: with one 128-column tile per row block, the workgroup that stores rows r..r+127 of block j is
                        // the only one that reads them, and it has read all of them (its whole K loop) before its epilogue
                        OAK_CHECK(gemm_nt(ctx, dBT + j0, inv, dBT + j0, nrhs, NB, NB, ldb, NB, ldb, 1.0, 0.0, 0));
                    } else {
                        OAK_HIP_CHECK(hipMemcpy2DAsync(dTmp, sizeof(double) * NB, dBT + j0, sizeof(double) * (size_t)ldb, sizeof(double) * NB,
                                                       (size_t)nrhs, hipMemcpyDeviceToDevice, ctx->stream));
                        OAK_CHECK(gemm_nt(ctx, dTmp, inv, dBT + j0, nrhs, NB, NB, NB, NB, ldb, 1.0, 0.0, 0));
                    }
                } else {
                    OAK_CHECK(trsm_leaf(ctx, dL + j0 * ldl + j0, nbj, ldl, dBT + j0, nrhs, ldb, 0));
                }
            }
            return OAK_OK;
        }
        for (int64_t j0 = 0; j0 < n; j0 += NB) {
            const int64_t nbj = (n - j0 < NB) ? n - j0 : NB;
            OAK_CHECK(trsm_leaf(ctx, dL + j0 * ldl + j0, nbj, ldl, dBT + j0, nrhs, ldb, 0));
            const int64_t rest = n - j0 - nbj;
            if (rest > 0)   // B[:, rest] -= X_j L[rest, j]^T
                OAK_CHECK(gemm_nt(ctx, dBT + j0, dL + (j0 + nbj) * ldl + j0, dBT + j0 + nbj, nrhs, rest, nbj, ldb, ldl, ldb, -1.0, 1.0, 0));
        }
    } else {
        const int64_t last = ((n - 1) / NB) * NB;
        if (nrhs >= 8192) {
            // Same formulation for the transposed solve (rows x with x L = b, i.e. L^T x^T = b^T; the N-sized adjoint solve of
            // the SVGP reverse pass): blocks from the last to the first, left-looking.  Block j receives the finished
            // later blocks in one GEMM  B[:, j] -= X[:, j+1:] L[j+1:, j]  -- against a transposed copy of L so that it is
            // the NT form the 128 x 128 MFMA kernel takes -- then  X[:, j] = B[:, j] L_jj^-1.
            double *dInvT = nullptr, *dTmp = nullptr, *dLT = nullptr;
            const int64_t nfull = n / NB;
            const int64_t ldt = (n + 1) & ~(int64_t)1;
            OAK_CHECK(get_buf_t(ctx, "trsm_invT", (size_t)(nfull > 0 ? nfull : 1) * NB * NB, &dInvT));
            OAK_CHECK(get_buf_t(ctx, "trsm_tmp", (size_t)nrhs * NB, &dTmp));
            OAK_CHECK(get_buf_t(ctx, "trsm_LT", (size_t)n * ldt, &dLT));
            OAK_CHECK(transpose(ctx, dL, n, n, ldl, dLT, ldt));
            for (int64_t b = 0; b < nfull; ++b) {
                double* it = dInvT + b * NB * NB;
                OAK_CHECK(set_identity(ctx, it, NB));
                OAK_CHECK(trsm_leaf(ctx, dL + (b * NB) * ldl + b * NB, NB, ldl, it, NB, NB, 0));   // the matrix L_bb^-T
            }
            for (int64_t j0 = last; j0 >= 0; j0 -= NB) {
                const int64_t nbj = (n - j0 < NB) ? n - j0 : NB;
                const int64_t rest = n - j0 - nbj;
                if (rest > 0)
                    OAK_CHECK(gemm_nt(ctx, dBT + j0 + nbj, dLT + j0 * ldt + j0 + nbj, dBT + j0, nrhs, nbj, rest, ldb, ldt, ldb, -1.0, 1.0, 0));
                if (nbj == NB) {
                    const double* invt = dInvT + (j0 / NB) * NB * NB;
                    if (gemm128_eligible(dBT + j0, invt, nrhs, NB, NB, ldb, NB, 0)) {
                        OAK_CHECK(gemm_nt(ctx, dBT + j0, invt, dBT + j0, nrhs, NB, NB, ldb, NB, ldb, 1.0, 0.0, 0));   // in place, as above
                    } else {
                        OAK_HIP_CHECK(hipMemcpy2DAsync(dTmp, sizeof(double) * NB, dBT + j0, sizeof(double) * (size_t)ldb, sizeof(double) * NB,
                                                       (size_t)nrhs, hipMemcpyDeviceToDevice, ctx->stream));
                        OAK_CHECK(gemm_nt(ctx, dTmp, invt, dBT + j0, nrhs, NB, NB, NB, NB, ldb, 1.0, 0.0, 0));
                    }
                } else {
                    OAK_CHECK(trsm_leaf(ctx, dL + j0 * ldl + j0, nbj, ldl, dBT + j0, nrhs, ldb, 1));
                }
            }
            return OAK_OK;
        }
        for (int64_t j0 = last; j0 >= 0; j0 -= NB) {
            const int64_t nbj = (n - j0 < NB) ? n - j0 : NB;
            OAK_CHECK(trsm_leaf(ctx, dL + j0 * ldl + j0, nbj, ldl, dBT + j0, nrhs, ldb, 1));
            if (j0 > 0)     // B[:, 0:j0] -= X_j L[j, 0:j0]
                OAK_CHECK(gemm_nn(ctx, dBT + j0, dL + j0 * ldl, dBT, nrhs, j0, nbj, ldb, ldl, ldb, -1.0, 1.0));
        }
    }
    return OAK_OK;
}

}  // namespace oak

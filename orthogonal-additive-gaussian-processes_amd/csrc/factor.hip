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
#include <cstdlib>

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

// tri (bit mask) declares triangular operands so that a tile only walks the k range where both are non-zero:
//   1: A lower triangular (k < r0 + 64)   2: A upper triangular (k >= r0)
//   4: B lower triangular (k < c0 + 64)   8: B upper triangular (k >= c0)      (B bits for BT = 1, B indexed [n][k])
// nsplitk > 1: gridDim.z slices of each tile's k range write raw partial tiles to part[z][m][n]; gemm_splitk_reduce sums
// them in a fixed order (the O(M^3) tail multiplies M x M matrices: 64 to 256 tiles cannot fill 256 CUs on their own).
constexpr int GM_TRI_A_LOWER = 1, GM_TRI_A_UPPER = 2, GM_TRI_B_LOWER = 4, GM_TRI_B_UPPER = 8;

template <int BT>
__global__ void __launch_bounds__(256) gemm_mfma_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                        double* __restrict__ C, int64_t m, int64_t n, int64_t k, int64_t lda,
                                                        int64_t ldb, int64_t ldc, double alpha, double beta, int lower_only,
                                                        int tri, int nsplitk, double* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) double As[GM_T * GM_PA];
    __shared__ __attribute__((aligned(16))) double Bs[(BT ? GM_T * GM_PA : GM_K * GM_PB)];
    if (lower_only && blockIdx.x > blockIdx.y) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t r0 = (int64_t)blockIdx.y * GM_T, c0 = (int64_t)blockIdx.x * GM_T;
    // k range of this tile, then of this split
    int64_t kb = 0, ke = k;
    if (tri & GM_TRI_A_LOWER) ke = (r0 + GM_T < ke) ? r0 + GM_T : ke;
    if (tri & GM_TRI_A_UPPER) kb = (r0 > kb) ? r0 : kb;
    if (tri & GM_TRI_B_LOWER) ke = (c0 + GM_T < ke) ? c0 + GM_T : ke;
    if (tri & GM_TRI_B_UPPER) kb = (c0 > kb) ? c0 : kb;
    if (nsplitk > 1) {
        int64_t chunk = (ke - kb + nsplitk - 1) / nsplitk;
        chunk = ((chunk + GM_K - 1) / GM_K) * GM_K;
        kb += (int64_t)blockIdx.z * chunk;
        ke = (kb + chunk < ke) ? kb + chunk : ke;
    }
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
    // branch-free (clamped) loads so that all 16 stay in flight; values outside the matrix / k range are zeroed by selects
    auto load_stage = [&](int64_t k0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t kk = k0 + ak + q;
            const int64_t kc = kk < k ? kk : k - 1;
            const double x = A[arow * lda + kc];
            va[q] = (kk < ke && r0 + ar < m) ? x : 0.0;
            if (BT) {
                const double y = B[brow * ldb + kc];
                vb[q] = (kk < ke && c0 + ar < n) ? y : 0.0;
            } else {
                const int64_t kr = k0 + bk, nc = c0 + bn + q;
                const double y = B[(kr < k ? kr : k - 1) * ldb + (nc < n ? nc : n - 1)];
                vb[q] = (kr < ke && nc < n) ? y : 0.0;
            }
        }
    };
    if (kb < ke) load_stage(kb);
    for (int64_t k0 = kb; k0 < ke; k0 += GM_K) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            As[ar * GM_PA + ak + q] = va[q];
            if (BT) Bs[ar * GM_PA + ak + q] = vb[q];
            else Bs[bk * GM_PB + bn + q] = vb[q];
        }
        __syncthreads();
        if (k0 + GM_K < ke) load_stage(k0 + GM_K);           // register prefetch of the next K chunk under the MFMAs
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
    double* Cz = (nsplitk > 1) ? part + (int64_t)blockIdx.z * m * n : C;
    const int64_t ldz = (nsplitk > 1) ? n : ldc;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t row = r0 + wr * 32 + 16 * g + 4 * reg + fk, col = c0 + wc * 32 + 16 * h + fi;
                if (row < m && col < n) {
                    double* q = Cz + row * ldz + col;
                    if (nsplitk > 1) { *q = acc[g][h][reg]; continue; }
                    const double v = alpha * acc[g][h][reg];
                    *q = (beta == 0.0) ? v : __builtin_fma(beta, *q, v);
                }
            }
}

// C = beta*C + alpha * sum_z part[z]  (fixed order); lower_only: only the 64 x 64 tiles on or below the diagonal were written
__global__ void __launch_bounds__(256) gemm_splitk_reduce_kernel(const double* __restrict__ part, int nsplitk, int64_t m, int64_t n,
                                                                 double* __restrict__ C, int64_t ldc, double alpha, double beta, int lower_only) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= n) return;
    if (lower_only && (j / GM_T) > (i / GM_T)) return;
    double s = 0.0;
    for (int z = 0; z < nsplitk; ++z) s += part[(int64_t)z * m * n + i * n + j];
    double* q = C + i * ldc + j;
    const double v = alpha * s;
    *q = (beta == 0.0) ? v : __builtin_fma(beta, *q, v);
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

// RANKP (only with TAIL = false): the epilogue adds  sum_p Yx[p][row] * ax[p][col]  to the tile before it is stored -- the rank-one
// adjoints y_p a_p^T of the extra output columns of a multi-output model (grad.hip), 64 * nx FMAs per thread against the tile's
// 2^21 MFMA flops per wave, instead of a read-modify-write pass over the finished 8.6 GB panel.  The other instantiations are
// the kernels as they were.
template <bool TAIL, bool RANKP = false>     // TAIL: triangular k ranges + split-k partials (M x M products); false: the N-sized products, unchanged
__global__ void __launch_bounds__(256, 2)
gemm128_nt_kernel(const double* A, const double* __restrict__ B, double* C /* may alias A: trsm_rows runs the diagonal-block product in place */, int64_t m, int64_t n,
                  int64_t k, int64_t lda, int64_t ldb, int64_t ldc, double alpha, double beta, int ntn,
                  int tri /* GM_TRI_* */, int nsplitk /* gridDim.y slices of each tile's k range -> part[z][m][n] */, double* __restrict__ part,
                  const double* __restrict__ Yx = nullptr, int64_t ldy = 0, const double* __restrict__ ax = nullptr, int nx = 0, int xcd_cols = 0) {
    __shared__ __attribute__((aligned(16))) double As[G2_T * G2_P];
    __shared__ __attribute__((aligned(16))) double Bs[G2_T * G2_P];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile mapping: the ntn column tiles of one row block run on the same XCD ...
    const int64_t grp_size = 8LL * ntn;
    const int64_t grp = blockIdx.x / grp_size, within = blockIdx.x - grp * grp_size;
    int64_t rb = grp * 8 + (within & 7);
    int64_t cb = within >> 3;
    if (xcd_cols == 2) {
        // ... or (r05, ntn = 8): an XCD owns TWO column tiles (2 MB of B, resident in its 4 MB L2 -- with all eight, the 8 MB of B the
        // staggered workgroups keep in flight do not fit and B is re-fetched from the fabric by nearly every workgroup) and every second
        // row block; a row block of A is then fetched by four XCDs instead of one.
        const int xcd = (int)(blockIdx.x & 7);
        const int64_t j = blockIdx.x >> 3;
        rb = 2 * (j >> 1) + (xcd >> 2);
        cb = 2 * (xcd & 3) + (j & 1);
    }
    const int64_t r0 = rb * G2_T, c0 = cb * G2_T;
    if (r0 >= m) return;
    // k range of this tile (triangular operands), then of this split
    int64_t kb = 0, ke = k;
    if constexpr (TAIL) {
        if (tri & GM_TRI_A_LOWER) ke = (r0 + G2_T < ke) ? r0 + G2_T : ke;
        if (tri & GM_TRI_A_UPPER) kb = (r0 > kb) ? r0 : kb;
        if (tri & GM_TRI_B_LOWER) ke = (c0 + G2_T < ke) ? c0 + G2_T : ke;
        if (tri & GM_TRI_B_UPPER) kb = (c0 > kb) ? c0 : kb;
        if (nsplitk > 1) {
            int64_t chunk = (ke - kb + nsplitk - 1) / nsplitk;
            chunk = ((chunk + G2_K - 1) / G2_K) * G2_K;
            kb += (int64_t)blockIdx.y * chunk;
            ke = (kb + chunk < ke) ? kb + chunk : ke;
        }
    }
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
            const bool okk = k0 + lc + 1 < ke;
            const bool oka = okk && (r0 + row < m), okb = okk && (c0 + row < n);
            ra[q] = make_double2(oka ? va.x : 0.0, oka ? va.y : 0.0);
            rbv[q] = make_double2(okb ? vb.x : 0.0, okb ? vb.y : 0.0);
        }
    };
    load_stage(kb < ke ? kb : 0);
    for (int64_t k0 = kb; k0 < ke; k0 += G2_K) {
#pragma unroll
        for (int q = 0; q < G2_NQ; ++q) {
            const int row = (wave * G2_NQ + q) * G2_RPL + lr;
            *reinterpret_cast<double2*>(&As[row * G2_P + lc]) = ra[q];
            *reinterpret_cast<double2*>(&Bs[row * G2_P + lc]) = rbv[q];
        }
        __syncthreads();
        load_stage((k0 + G2_K < ke) ? k0 + G2_K : 0);
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
    if constexpr (RANKP) {
        // per output p: this thread's 16 row values and 4 column values (20 loads), 64 FMAs into the accumulators (alpha = 1 here)
        for (int p = 0; p < nx; ++p) {
            double av[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int64_t col = c0 + 64 * wc + 16 * h + fi;
                av[h] = ax[(int64_t)p * n + (col < n ? col : n - 1)];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int64_t row = r0 + 64 * wr + 16 * g + 4 * reg + fk;
                    const double yv = Yx[(int64_t)p * ldy + (row < m ? row : m - 1)];
#pragma unroll
                    for (int h = 0; h < 4; ++h) acc[g][h][reg] = __builtin_fma(yv, av[h], acc[g][h][reg]);
                }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t row = r0 + 64 * wr + 16 * g + 4 * reg + fk, col = c0 + 64 * wc + 16 * h + fi;
                if (row < m && col < n) {
                    if constexpr (TAIL) {
                        if (nsplitk > 1) { part[(int64_t)blockIdx.y * m * n + row * n + col] = acc[g][h][reg]; continue; }
                    }
                    double* q = C + row * ldc + col;
                    const double v = alpha * acc[g][h][reg];
                    *q = (beta == 0.0) ? v : __builtin_fma(beta, *q, v);
                }
            }
}

static int gemm_launch(oak_ctx* ctx, int bt, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k,
                       int64_t lda, int64_t ldb, int64_t ldc, double alpha, double beta, int lower_only, int tri = 0, bool allow_split = false) {
    if (m <= 0 || n <= 0) return OAK_OK;
    const int64_t tiles = ((n + GM_T - 1) / GM_T) * ((m + GM_T - 1) / GM_T);
    int nsplitk = 1;
    if (allow_split) {                     // few tiles, long k: slice k until ~4 workgroups per CU, at least two stages per slice
        nsplitk = (int)((4 * (int64_t)ctx->num_cu) / (tiles > 0 ? tiles : 1));
        if (nsplitk > 8) nsplitk = 8;
        while (nsplitk > 1 && k / nsplitk < 2 * GM_K) --nsplitk;
        if (nsplitk < 1) nsplitk = 1;
    }
    double* d_part = nullptr;
    if (nsplitk > 1) OAK_CHECK(get_buf_t(ctx, (ctx->side != nullptr && ctx->stream == ctx->side) ? "gemm_part_side" : "gemm_part",
                                         (size_t)nsplitk * m * n, &d_part));
    dim3 grid((unsigned)((n + GM_T - 1) / GM_T), (unsigned)((m + GM_T - 1) / GM_T), (unsigned)nsplitk);
    if (bt) gemm_mfma_kernel<1><<<grid, 256, 0, ctx->stream>>>(dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, lower_only, tri, nsplitk, d_part);
    else    gemm_mfma_kernel<0><<<grid, 256, 0, ctx->stream>>>(dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, lower_only, tri, nsplitk, d_part);
    OAK_HIP_CHECK(hipGetLastError());
    if (nsplitk > 1) {
        dim3 rg((unsigned)((n + 255) / 256), (unsigned)m);
        gemm_splitk_reduce_kernel<<<rg, 256, 0, ctx->stream>>>(d_part, nsplitk, m, n, dC, ldc, alpha, beta, lower_only);
        OAK_HIP_CHECK(hipGetLastError());
    }
    return OAK_OK;
}
int gemm_nn(oak_ctx* ctx, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k, int64_t lda,
            int64_t ldb, int64_t ldc, double alpha, double beta) {
    return gemm_launch(ctx, 0, dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, 0);
}
// The M x M products of the O(M^3) tail: triangular operands declared (tri: GM_TRI_* bits as OAK_TRI_* in oak_internal.h),
// k sliced over gridDim.z.  bt = 1: C = alpha A B^T + beta C; bt = 0: C = alpha A B + beta C (B bits then unsupported).
int gemm_tail(oak_ctx* ctx, int bt, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k, int64_t lda,
              int64_t ldb, int64_t ldc, double alpha, double beta, int tri) {
    if (!bt && (tri & (GM_TRI_B_LOWER | GM_TRI_B_UPPER))) { set_error("gemm_tail: B triangle bits need the NT form"); return OAK_E_ARG; }
    const bool aligned = ((lda | ldb) & 1) == 0 && (((uintptr_t)dA | (uintptr_t)dB) & 15) == 0 && k >= 2 && (k & 1) == 0;
    if (bt && aligned && m >= G2_T && n >= G2_T && getenv("OAK_TAIL_GEMM64") == nullptr) {
        // 128 x 128 tiles (16-byte operand loads, the kernel of the N-sized products) with k sliced over gridDim.y until
        // every CU has a workgroup: 64 tiles at M = 1024, 16 at M = 512
        const int ntn = (int)((n + G2_T - 1) / G2_T);
        const int64_t nrb = (m + G2_T - 1) / G2_T;
        const int64_t ngrp = (nrb + 7) / 8;
        int nsplitk = (int)((2 * (int64_t)ctx->num_cu) / (nrb * ntn));
        if (nsplitk > 16) nsplitk = 16;
        while (nsplitk > 1 && k / nsplitk < 2 * G2_K) --nsplitk;
        if (nsplitk < 1) nsplitk = 1;
        double* d_part = nullptr;
        if (nsplitk > 1) OAK_CHECK(get_buf_t(ctx, (ctx->side != nullptr && ctx->stream == ctx->side) ? "gemm_part_side" : "gemm_part",
                                             (size_t)nsplitk * m * n, &d_part));
        dim3 grid((unsigned)(ngrp * 8 * ntn), (unsigned)nsplitk);
        gemm128_nt_kernel<true><<<grid, 256, 0, ctx->stream>>>(dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, ntn, tri, nsplitk, d_part);
        OAK_HIP_CHECK(hipGetLastError());
        if (nsplitk > 1) {
            dim3 rg((unsigned)((n + 255) / 256), (unsigned)m);
            gemm_splitk_reduce_kernel<<<rg, 256, 0, ctx->stream>>>(d_part, nsplitk, m, n, dC, ldc, alpha, beta, 0);
            OAK_HIP_CHECK(hipGetLastError());
        }
        return OAK_OK;
    }
    return gemm_launch(ctx, bt, dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, 0, tri, true);
}
// true when gemm_nt takes the 128 x 128 kernel (one workgroup per output tile, every operand read before the tile is stored)
static bool gemm128_eligible(const double* dA, const double* dB, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int lower_only) {
    const bool aligned = ((lda | ldb) & 1) == 0 && (((uintptr_t)dA | (uintptr_t)dB) & 15) == 0 && k >= 2 && (k & 1) == 0;
    return !lower_only && aligned && m >= 2048 && n >= 128 && m > 0 && n > 0;
}
// C = A B^T + sum_p Yx[p][:]^T ax[p][:]   (Yx [nx x ldy] indexed by C's rows, ax [nx x n]); false when the shape does not take
// the 128 x 128 kernel (the caller then adds the rank-nx term in a pass of its own)
bool gemm_nt_rankp(oak_ctx* ctx, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb,
                   int64_t ldc, const double* dYx, int64_t ldy, const double* d_ax, int nx, int* status) {
    *status = OAK_OK;
    if (!gemm128_eligible(dA, dB, m, n, k, lda, ldb, 0)) return false;
    const int ntn = (int)((n + G2_T - 1) / G2_T);
    const int64_t nrb = (m + G2_T - 1) / G2_T;
    const int64_t ngrp = (nrb + 7) / 8;
    gemm128_nt_kernel<false, true><<<(unsigned)(ngrp * 8 * ntn), 256, 0, ctx->stream>>>(dA, dB, dC, m, n, k, lda, ldb, ldc, 1.0, 0.0, ntn, 0, 1, nullptr,
                                                                                       dYx, ldy, d_ax, nx);
    if (hipGetLastError() != hipSuccess) { set_error("gemm128_nt_kernel<false, true> launch failed"); *status = OAK_E_HIP; }
    return true;
}
int gemm_nt(oak_ctx* ctx, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k, int64_t lda,
            int64_t ldb, int64_t ldc, double alpha, double beta, int lower_only) {
    if (gemm128_eligible(dA, dB, m, n, k, lda, ldb, lower_only)) {
        const int ntn = (int)((n + G2_T - 1) / G2_T);
        const int64_t nrb = (m + G2_T - 1) / G2_T;
        const int64_t ngrp = (nrb + 7) / 8;
        // two column tiles per XCD when the matrix is exactly eight tiles wide and the row blocks pair up (M = 1024: the adjoint GEMM)
        // (fabric fetch of the adjoint GEMM 62.1 -> 52.4 GB per launch, 33.0 -> 32.85 ms; OAK_GEMM_XCD_COLS=0 restores the old mapping)
        static const int colmap = [] { const char* e = getenv("OAK_GEMM_XCD_COLS"); return e ? atoi(e) : 2; }();
        const int xcd_cols = (colmap == 2 && ntn == 8 && (nrb % 2) == 0 && dC != dA) ? 2 : 0;
        const unsigned nblk = xcd_cols == 2 ? (unsigned)(nrb * ntn) : (unsigned)(ngrp * 8 * ntn);
        gemm128_nt_kernel<false><<<nblk, 256, 0, ctx->stream>>>(dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta, ntn, 0, 1, nullptr, nullptr, 0, nullptr, 0,
                                                               xcd_cols);
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
constexpr int PO_RPW = 64;         // rows per role-A workgroup of the fused step (4 lanes per row, 256 worker lanes)

// value of quad lane `o` (0..3) to the four lanes of each quad (DPP quad_perm broadcast: no LDS, no SGPR round trip)
template <int O> __device__ __forceinline__ double quad_bcast_imm(double v) {
    constexpr int ctrl = O | (O << 2) | (O << 4) | (O << 6);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_bcast_f64(double v, int o) {
    switch (o & 3) {
        case 0: return quad_bcast_imm<0>(v);
        case 1: return quad_bcast_imm<1>(v);
        case 2: return quad_bcast_imm<2>(v);
        default: return quad_bcast_imm<3>(v);
    }
}

// One wave factors the NB x NB diagonal block.  Lane i owns row i in registers, right-looking: a[c] always carries the
// rank-1 updates of the finished columns.  The dependent chain of a column is
//     pivot (v_readlane) -> 1/sqrt (estimate + 2 Newton steps) -> scale -> the ONE entry the next pivot needs
//     (l_{j+1,j} by v_readlane) -> one FMA into a[j+1];
// the rank-1 update of the other entries runs one column behind: column j's multipliers are read back (broadcast reads of a
// column-major LDS copy) right after they are written and consumed only during column j+1, so neither their FMAs nor the
// LDS round trip sit on the chain.  All stores are unconditional (lanes i < j write don't-care values above the diagonal,
// the shadow lanes the same values to the same addresses) so that the 32 columns are ONE basic block: with a branch per
// column the compiler sinks every deferred FMA into the block that consumes it and spills the multipliers.
// Measured alone (tools/ubench/potf2_bench): see DESIGN.md.  invd[j] = 1 / L_jj is kept for the panel solve; Dg receives L
// row-major, Lt is scratch.
// COLUMN_FENCE: a scheduling barrier closes every column, so the multiplier reads issued in column j stay there (they are
// consumed one column later, off the chain) instead of being sunk next to their uses with a wait in front of each.
template <bool COLUMN_FENCE = false>
__device__ __forceinline__ void potf2_wave(double* Dg /*[NB][PO_P]*/, double* Lt /*[NB][PO_P]*/, double* invd /*[NB]*/, int lane,
                                           int64_t j0, int* info) {
    const int i = lane % PO_NB;                 // with NB = 32 lanes 32..63 shadow lanes 0..31
    double a[PO_NB], mp[PO_NB], mc[PO_NB];
#pragma unroll
    for (int c = 0; c < PO_NB; ++c) { a[c] = Dg[i * PO_P + c]; mp[c] = 0.0; mc[c] = 0.0; }
    double lprev = 0.0;
#pragma unroll
    for (int j = 0; j < PO_NB; ++j) {
        const double d = readlane_f64(a[j], j);                 // a[j] of lane j: all earlier columns applied
        const double r = rsqrt_newton(d);
        const double l = a[j] * r;                              // column j of L (lanes i >= j; lane j: d / sqrt(d))
        a[j] = l;
        if (j + 1 < PO_NB) a[j + 1] = __builtin_fma(-l, readlane_f64(l, j + 1), a[j + 1]);   // what the next pivot waits for
        Dg[i * PO_P + j] = l;
        Lt[j * PO_P + i] = l;
        invd[j] = r;
#pragma unroll
        for (int c = j + 2; c < PO_NB; ++c) mc[c] = Lt[j * PO_P + c];                        // multipliers of column j, for the next column
        if (j >= 1) {                                           // column j-1's update of the entries right of column j
#pragma unroll
            for (int c = j + 1; c < PO_NB; ++c) a[c] = __builtin_fma(-lprev, mp[c], a[c]);
        }
#pragma unroll
        for (int c = j + 2; c < PO_NB; ++c) mp[c] = mc[c];
        lprev = l;
        if constexpr (COLUMN_FENCE) __builtin_amdgcn_sched_barrier(0);
    }
    // A pivot d <= 0 (or NaN) makes 1/sqrt(d) and with it L_jj = d / sqrt(d) non-finite: the first such diagonal entry is the
    // failing leading minor.  Checked once here rather than per column: d is wave-uniform, so a per-column test compiles to a
    // scalar branch and splits the block (see above).
    const double ldiag = Dg[i * PO_P + i];
    const unsigned long long badmask = __ballot(!(ldiag > 0.0 && ldiag < __builtin_inf()));
    if (lane < PO_NB) {
#pragma unroll
        for (int c = 0; c < PO_NB; ++c)
            if (c > i) Dg[i * PO_P + c] = 0.0;
    }
    if (badmask != 0ull && lane == 0 && info != nullptr) atomicMin(info, (int)(j0 + (__ffsll((long long)badmask) - 1) + 1));
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
                                                         int* __restrict__ info, int* __restrict__ arrivals, int nA, int ntc, int pending,
                                                         long long* __restrict__ trace /* dev aid, normally NULL: 8 stamps per role */) {
#define PO_STAMP(slot) do { if (trace != nullptr && do_stamp) trace[slot] = wall_clock64(); } while (0)
    __shared__ __attribute__((aligned(16))) double Dg[PO_NB * PO_P];
    __shared__ __attribute__((aligned(16))) double Lt[PO_NB * PO_P];    // role A: column-major scratch of the one-wave factorisation
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
        const bool do_stamp = tid == 0 && (t == 0 || blockIdx.x == gridDim.x - 1);
        PO_STAMP(t == 0 ? 16 : 20);
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
        PO_STAMP(t == 0 ? 17 : 21);
        return;
    }
    // ---------------- role A: panel j ----------------
    // 320 threads: wave 0 factors the diagonal block; the 256 workers (waves 1..4) own 64 rows below it, FOUR LANES PER ROW
    // (lane q of a quad holds columns q, q+4, .., q+28 of its row).  One row per lane kept a single lane busy for 1024 + 512
    // dependent-ish FMAs on 64 scattered 8-byte loads (measured: 8.7 us for the pending update, 4.7 us for solve + store,
    // against 5-7 us for the one-wave factorisation they were meant to hide under); spread over a quad the update is 256
    // FMAs per lane and the triangular solve runs right-looking with the finished entry passed round the quad by DPP.
    const int wid = tid - 64;                                           // worker id, < 0 for the factor wave
    const bool worker = wid >= 0;
    const bool do_stamp = blockIdx.x == 0 && (tid == 0 || tid == 64);
    const int sb = tid == 0 ? 0 : 8;
    PO_STAMP(sb + 0);
    const int nb = (n - j0 < PO_NB) ? (int)(n - j0) : PO_NB;
    const bool pre = j0 > 0 && pending;                                 // pending = 0: the caller already applied panel j-1
    const int q = wid & 3;
    const int64_t row = j0 + nb + (int64_t)blockIdx.x * PO_RPW + (wid >> 2);
    const bool has_row = worker && row < nrows && nb == PO_NB;
    // every global load of the step is issued before the first barrier, in the order of use
    constexpr int PER = PO_NB * PO_NB / 256;
    double pjv[4], dv[PER];
    const int pr_r = wid >> 3, pr_k4 = (wid & 7) * 4;
    if (pre && worker) {   // rows j0 .. j0+31 of panel j-1
        const int64_t rr = (j0 + pr_r < n) ? j0 + pr_r : n - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) pjv[k] = A[rr * lda + p0 + pr_k4 + k];
    }
    if (worker) {
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int idx = wid + 256 * e;
            const int i = idx / PO_NB, j = idx % PO_NB;
            const int ic = i < nb ? i : nb - 1, jc = j < nb ? j : nb - 1;
            dv[e] = A[(j0 + ic) * lda + j0 + jc];
        }
    }
    double x[PO_NB / 4], pr[PO_NB];
    if (has_row) {
        const double* ap = A + row * lda + j0 + q;
#pragma unroll
        for (int t = 0; t < PO_NB / 4; ++t) x[t] = ap[4 * t];
        if (pre) {
            const double* pp = A + row * lda + p0;
#pragma unroll
            for (int k = 0; k < PO_NB; ++k) pr[k] = pp[k];
        }
    }
    if (pre && worker) {
#pragma unroll
        for (int k = 0; k < 4; ++k) Pj[pr_r * PO_P + pr_k4 + k] = (j0 + pr_r < n) ? pjv[k] : 0.0;
    }
    __syncthreads();
    PO_STAMP(sb + 1);
    if (worker) {
        // NB*NB entries of the diagonal block, clamped addresses; identity padding beyond nb; pending update applied
        double v[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int idx = wid + 256 * e;
            const int i = idx / PO_NB, j = idx % PO_NB;
            const int ic = i < nb ? i : nb - 1, jc = j < nb ? j : nb - 1;
            double xx = dv[e];
            if (pre) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < PO_NB; k += 2) {
                    const double2 a2 = *reinterpret_cast<const double2*>(&Pj[ic * PO_P + k]);
                    const double2 b2 = *reinterpret_cast<const double2*>(&Pj[jc * PO_P + k]);
                    s = __builtin_fma(a2.x, b2.x, s);
                    s = __builtin_fma(a2.y, b2.y, s);
                }
                xx -= s;
            }
            v[e] = (i < nb && j < nb) ? ((j <= i) ? xx : 0.0) : ((i == j) ? 1.0 : 0.0);
        }
#pragma unroll
        for (int e = 0; e < PER; ++e) { const int idx = wid + 256 * e; Dg[(idx / PO_NB) * PO_P + (idx % PO_NB)] = v[e]; }
    }
    __syncthreads();
    PO_STAMP(sb + 2);
    if (tid == 319) last_loader = (atomicAdd(arrivals, 1) == nA - 1);
    // wave-uniform split (scalar branch), each side with its own barrier: the workers' registers are not live across the
    // factorisation's code and vice versa
    if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
        potf2_wave(Dg, Lt, invd, tid, j0, blockIdx.x == 0 ? info : nullptr);
        PO_STAMP(sb + 3);
        __syncthreads();
        PO_STAMP(sb + 4);
    } else {
        if (has_row && pre) {
            // this lane's eight entries of panel j with the pending update of panel j-1 applied
#pragma unroll
            for (int t = 0; t < PO_NB / 4; ++t) {
                const double* pj = &Pj[(4 * t + q) * PO_P];
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int k = 0; k < PO_NB; k += 2) {
                    const double2 l2 = *reinterpret_cast<const double2*>(&pj[k]);
                    s0 = __builtin_fma(pr[k], l2.x, s0);
                    s1 = __builtin_fma(pr[k + 1], l2.y, s1);
                }
                x[t] -= s0 + s1;
            }
        }
        PO_STAMP(sb + 3);
        __syncthreads();
        PO_STAMP(sb + 4);
        if (has_row) {
            // X L_jj^T = A_panel, right-looking over the columns: the owner of column c finishes its entry, the quad receives it
            // by DPP and every lane eliminates it from its own later columns
#pragma unroll
            for (int c = 0; c < PO_NB; ++c) {
                const int to = c >> 2, owner = c & 3;
                const double xc = quad_bcast_f64(x[to] * invd[c], owner);
                x[to] = (q == owner) ? xc : x[to];
#pragma unroll
                for (int t = to; t < PO_NB / 4; ++t) {
                    const double l = Dg[(4 * t + q) * PO_P + c];
                    const double m = (t > to || q > owner) ? l : 0.0;
                    x[t] = __builtin_fma(-xc, m, x[t]);
                }
            }
            double* ap = A + row * lda + j0 + q;
#pragma unroll
            for (int t = 0; t < PO_NB / 4; ++t) ap[4 * t] = x[t];
        }
    }
    if (last_loader && worker) {
        for (int idx = wid; idx < nb * nb; idx += 256) {
            const int i = idx / nb, j = idx - i * nb;
            A[(j0 + i) * lda + j0 + j] = Dg[i * PO_P + j];
        }
    }
    PO_STAMP(sb + 5);
#undef PO_STAMP
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-level step (the default): panels of 32 columns inside outer blocks of PO_BLK = 128.  The trailing matrix behind an
// outer block is read and written ONCE per block (one rank-128 MFMA update, role B of the block's successor's first step)
// instead of once per panel; inside a block the factorisation is left-looking:
//   role A of panel j applies the KP pending columns itself -- the earlier panels of its own block (KP = 32, 64, 96) or, for
//          the first panel of a block, the whole previous block (KP = 128) -- to the diagonal block and to its 64 rows, both
//          as fp64 MFMA products against the staged rows j0 .. j0+31 of the pending columns (the 32-long dot products per
//          entry from LDS that the one-level step used cost 1.6 us on the critical path in front of the one-wave factorisation;
//          as 16 x 16 x 4 MFMAs they cost ~0.3 us per 32 pending columns);
//   role B (only in the launch of a block's first panel, KP = 128): C[r, c] -= P[r, :] P[c, :]^T over the 128 columns of the
//          previous block, for the lower 64 x 64 tiles right of panel j.
// Everything else (one-wave factorisation of the diagonal block, four lanes per row in the panel solve, the last loader
// writing the factor back) is the one-level step's.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PO_BLK = 128;

// amdgpu_waves_per_eu(1, 2): with the workers' registers gone the kernel would fit four waves per SIMD, and the scheduler, aiming
// for that occupancy, sinks every LDS read of the one-wave factorisation next to its use (216 waits instead of 90 in its 32
// columns, 4.9 -> 7.4 us).  Five waves per workgroup never occupy more than two slots of a SIMD, so the register budget of two is
// what the factorisation's schedule should be built for.
template <int KP>
__global__ void __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(1, 2))) potrf_blk_kernel(double* __restrict__ A, int64_t n, int64_t nrows, int64_t lda, int64_t j0,
                                                        int* __restrict__ info, int* __restrict__ arrivals, int nA, int ntc,
                                                        long long* __restrict__ trace /* dev aid, normally NULL */) {
#define PO_STAMP(slot) do { if (trace != nullptr && do_stamp) trace[slot] = wall_clock64(); } while (0)
    constexpr int PJP = (KP > 0 ? KP : 32) + 2;                      // LDS pitch of the staged pending rows
    // separate static arrays for the one-wave factorisation's scratch (distinct objects: the compiler may reorder their LDS
    // traffic freely; carved out of one dynamic block the same accesses may alias and every load waited for the store before it,
    // 4.9 -> 6.3 us per diagonal block); only the staged pending rows, whose size depends on KP, are dynamic
    __shared__ __attribute__((aligned(16))) double Dg[PO_NB * PO_P];
    __shared__ __attribute__((aligned(16))) double Lt[PO_NB * PO_P];
    __shared__ __attribute__((aligned(16))) double Pc[64 * PO_P];       // role A: C-layout -> quad-layout exchange; role B: operand tile
    __shared__ double invd[PO_NB];
    __shared__ int last_loader;
    extern __shared__ __attribute__((aligned(16))) double Pj[];         // role A: [32][PJP]; role B: [64][PO_P]
    const int tid = threadIdx.x;
    const int64_t p0 = j0 - KP;
    if ((int)blockIdx.x >= nA) {
        // ---------------- role B: C[r0:+64, c0:+64] -= P[r0:+64, :] P[c0:+64, :]^T,  P = A[:, p0 : p0 + KP] ----------------
        if constexpr (KP == PO_BLK) {
            const int t = blockIdx.x - nA;
            const int by = t / ntc, bx = t - by * ntc;
            if (bx > by || tid >= 256) return;
            const bool do_stamp = tid == 0 && (t == 0 || blockIdx.x == gridDim.x - 1);
            PO_STAMP(t == 0 ? 16 : 20);
            const int64_t s0 = j0 + PO_NB;
            const int64_t r0 = s0 + 64 * (int64_t)by, c0 = s0 + 64 * (int64_t)bx;
            const int r = tid >> 2, k8 = (tid & 3) * 8;
            const int64_t ra = (r0 + r < nrows) ? r0 + r : nrows - 1, rb = (c0 + r < n) ? c0 + r : n - 1;
            const double* pa = A + ra * lda + p0 + k8;
            const double* pb = A + rb * lda + p0 + k8;
            double va[8], vb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { va[q] = pa[q]; vb[q] = pb[q]; }
            const int lane = tid & 63;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int wr = wave >> 1, wc = wave & 1, fi = lane & 15, fk = lane >> 4;
            double4_t acc[2][2];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[g][h] = (double4_t){0.0, 0.0, 0.0, 0.0};
            // the C tile's own entries: loaded early, consumed by the epilogue
            double cv[2][2][4];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int64_t row = r0 + wr * 32 + 16 * g + 4 * reg + fk, col = c0 + wc * 32 + 16 * h + fi;
                        const int64_t rc = row < nrows ? row : nrows - 1, cc = col < n ? col : n - 1;
                        cv[g][h][reg] = A[rc * lda + cc];
                    }
#pragma unroll
            for (int kc = 0; kc < KP / 32; ++kc) {
                if (kc > 0) __syncthreads();
#pragma unroll
                for (int q = 0; q < 8; ++q) { Pj[r * PO_P + k8 + q] = va[q]; Pc[r * PO_P + k8 + q] = vb[q]; }
                __syncthreads();
                if (kc + 1 < KP / 32) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { va[q] = pa[32 * (kc + 1) + q]; vb[q] = pb[32 * (kc + 1) + q]; }
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
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
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int64_t row = r0 + wr * 32 + 16 * g + 4 * reg + fk, col = c0 + wc * 32 + 16 * h + fi;
                        if (row < nrows && col < n) A[row * lda + col] = cv[g][h][reg] - acc[g][h][reg];
                    }
            PO_STAMP(t == 0 ? 17 : 21);
        }
        return;
    }
    // ---------------- role A: panel j ----------------
    // Six waves: wave 0 factors the diagonal block; waves 1, 2, 3 and 5 are the 256 workers; wave 4 -- the one that shares wave 0's
    // SIMD (waves of a workgroup go to the SIMDs round-robin) -- only takes part in the barriers.  A worker wave on the factor
    // wave's SIMD puts its 64-cycle MFMAs into the one-wave factorisation's dependent chain (+1.6 us at KP = 128); two worker
    // waves sharing SIMD 1 merely take twice as long for an update that hides under the factorisation anyway.
    const int wv = tid >> 6;
    const int wid = (wv >= 1 && wv <= 3) ? tid - 64 : (wv == 5 ? tid - 128 : -1);     // worker id 0..255, < 0: factor wave / idle wave
    const bool worker = wid >= 0;
    const bool do_stamp = blockIdx.x == 0 && (tid == 0 || tid == 64);
    const int sb = tid == 0 ? 0 : 8;
    PO_STAMP(sb + 0);
    const int nb = (n - j0 < PO_NB) ? (int)(n - j0) : PO_NB;
    const int q = wid & 3;
    const int64_t row0 = j0 + nb + (int64_t)blockIdx.x * PO_RPW;        // first of this workgroup's 64 rows
    const int64_t row = row0 + (wid >> 2);
    const bool has_row = worker && row < nrows && nb == PO_NB;
    const int ww = wid >> 6;                                            // worker wave 0..3 (valid for workers)
    const int fi = tid & 15, fk = (tid & 63) >> 4;
    // the diagonal block's pending update runs BEFORE the factorisation starts, so it may use wave 4: waves 1..4 sit on four
    // different SIMDs and take one 16 x 16 tile each (with waves 1, 2, 3, 5 two tiles would queue on SIMD 1, on the critical path)
    const bool dworker = wv >= 1 && wv <= 4;
    const int tr = (wv - 1) >> 1, tc = (wv - 1) & 1;                    // this wave's 16 x 16 tile of the diagonal block
    // in front of the first barrier only what the critical path needs: the staged rows and the diagonal block
    constexpr int PJ_PER = KP > 0 ? KP / 16 : 1;
    double2 pjv[PJ_PER];
    double dacc[4], cacc[2][4], aop[KP > 0 ? KP / 4 : 1];
    const int pj_r = wid >> 3, pj_k = 2 * (wid & 7);                    // staging: eight workers per row, 16 bytes each per 128-byte piece
    if (worker) {
        if constexpr (KP > 0) {                                         // rows j0 .. j0+31 of the pending columns
            // ONE address per thread and immediate offsets (an index / KP, a clamp and a 64-bit row product per element made the
            // integer pipe, not the memory, the limit of this phase: +0.5 us per 32 pending columns)
            const int64_t rr = (j0 + pj_r < n) ? j0 + pj_r : n - 1;
            const double2* src = reinterpret_cast<const double2*>(A + rr * lda + p0 + pj_k);
#pragma unroll
            for (int e = 0; e < PJ_PER; ++e) pjv[e] = src[8 * e];
        }
    }
    // the staged rows are what the first barrier waits for: their loads go out first (left alone the compiler issues them last,
    // behind the operand loads, and the LDS writes in front of the barrier then wait for everything)
    __builtin_amdgcn_sched_barrier(0);
    if (dworker) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {                             // the diagonal block, MFMA C layout
            const int i = 16 * tr + 4 * reg + fk, j = 16 * tc + fi;
            const int ic = i < nb ? i : nb - 1, jc = j < nb ? j : nb - 1;
            dacc[reg] = A[(j0 + ic) * lda + j0 + jc];
        }
    }
    if (worker) {
        if constexpr (KP > 0) {
            const bool live = j0 + pj_r < n;
#pragma unroll
            for (int e = 0; e < PJ_PER; ++e)
                *reinterpret_cast<double2*>(&Pj[pj_r * PJP + pj_k + 16 * e]) = live ? pjv[e] : make_double2(0.0, 0.0);
        }
    }
    __syncthreads();
    PO_STAMP(sb + 1);
    if (dworker) {
        // diagonal block with the pending update applied; identity padding beyond nb; strict upper part zeroed
        double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
        if constexpr (KP > 0) {
#pragma unroll
            for (int ks = 0; ks < KP / 4; ++ks) {
                const double a = Pj[(16 * tr + fi) * PJP + 4 * ks + fk];
                const double b = Pj[(16 * tc + fi) * PJP + 4 * ks + fk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = 16 * tr + 4 * reg + fk, j = 16 * tc + fi;
            const double xx = dacc[reg] - acc[reg];
            Dg[i * PO_P + j] = (i < nb && j < nb) ? ((j <= i) ? xx : 0.0) : ((i == j) ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    PO_STAMP(sb + 2);
    if (tid == 383) last_loader = (atomicAdd(arrivals, 1) == nA - 1);
    double x[PO_NB / 4];
    if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
        potf2_wave<true>(Dg, Lt, invd, tid, j0, blockIdx.x == 0 ? info : nullptr);
        PO_STAMP(sb + 3);
        __syncthreads();
        PO_STAMP(sb + 4);
    } else {
        if (worker && nb == PO_NB) {
            // The workers' own operands are issued only now, behind the second barrier: they are consumed under the one-wave
            // factorisation (~5 us), which hides their latency.  Issued earlier, their ~80 KiB of requests queued in the CU's one
            // address path ahead of the staged rows (first barrier, +1.5 us at KP = 128) or ahead of the diagonal update.
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {                     // this wave's 16 rows of panel j, C layout
                    const int64_t rr = row0 + 16 * ww + 4 * reg + fk;
                    const int64_t rc = rr < nrows ? rr : nrows - 1;
                    cacc[h][reg] = A[rc * lda + j0 + 16 * h + fi];
                }
            if constexpr (KP > 0) {
                // A operand of the rows' pending update, straight into MFMA layout.  The k index of a 16 x 16 x 4 step is the lane
                // group fk; which pending column a (step, group) pair stands for is free as long as both operands agree.  Steps
                // 2 k2 and 2 k2 + 1 of group fk take columns 8 k2 + 2 fk + {0, 1}: one 16-byte load per lane, the four groups of a
                // row reading 64 contiguous bytes per instruction (8-byte loads 32 bytes apart fetched every line four times over
                // four instructions; a contiguous run per group made every instruction touch 64 different lines and thrashed L1).
                const int64_t ra = row0 + 16 * ww + fi;
                const double2* ap = reinterpret_cast<const double2*>(A + (ra < nrows ? ra : nrows - 1) * lda + p0 + 2 * fk);
#pragma unroll
                for (int k2 = 0; k2 < KP / 8; ++k2) { const double2 v = ap[4 * k2]; aop[2 * k2] = v.x; aop[2 * k2 + 1] = v.y; }
            }
            // this wave's 16 rows x 32 columns of panel j with the pending update applied, then from the MFMA C layout to the
            // solve's four-lanes-per-row layout through LDS (rows 16 ww .. 16 ww + 15 are written and read by this wave only)
            double4_t c0 = (double4_t){0.0, 0.0, 0.0, 0.0}, c1 = (double4_t){0.0, 0.0, 0.0, 0.0};
            if constexpr (KP > 0) {
#pragma unroll
                for (int ks = 0; ks < KP / 4; ++ks) {
                    const double b0 = Pj[fi * PJP + 8 * (ks >> 1) + 2 * fk + (ks & 1)];
                    const double b1 = Pj[(16 + fi) * PJP + 8 * (ks >> 1) + 2 * fk + (ks & 1)];
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ks], b0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ks], b1, c1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                Pc[(16 * ww + 4 * reg + fk) * PO_P + fi] = cacc[0][reg] - c0[reg];
                Pc[(16 * ww + 4 * reg + fk) * PO_P + 16 + fi] = cacc[1][reg] - c1[reg];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int t = 0; t < PO_NB / 4; ++t) x[t] = Pc[(wid >> 2) * PO_P + 4 * t + q];
        }
        PO_STAMP(sb + 3);
        __syncthreads();
        PO_STAMP(sb + 4);
        if (has_row) {
            // X L_jj^T = A_panel, right-looking over the columns (potrf_step_kernel)
#pragma unroll
            for (int c = 0; c < PO_NB; ++c) {
                const int to = c >> 2, owner = c & 3;
                const double xc = quad_bcast_f64(x[to] * invd[c], owner);
                x[to] = (q == owner) ? xc : x[to];
#pragma unroll
                for (int t = to; t < PO_NB / 4; ++t) {
                    const double l = Dg[(4 * t + q) * PO_P + c];
                    const double m = (t > to || q > owner) ? l : 0.0;
                    x[t] = __builtin_fma(-xc, m, x[t]);
                }
            }
            double* ap = A + row * lda + j0 + q;
#pragma unroll
            for (int t = 0; t < PO_NB / 4; ++t) ap[4 * t] = x[t];
        }
    }
    if (last_loader && worker) {
        for (int idx = wid; idx < nb * nb; idx += 256) {
            const int i = idx / nb, j = idx - i * nb;
            A[(j0 + i) * lda + j0 + j] = Dg[i * PO_P + j];
        }
    }
    PO_STAMP(sb + 5);
#undef PO_STAMP
}

template <int KP>
static int potrf_blk_launch(oak_ctx* ctx, unsigned grid, double* dA, int64_t n, int64_t nr, int64_t lda, int64_t j0, int* d_info,
                            int* d_arr, int nA, int ntc, long long* trc) {
    constexpr int PJP = (KP > 0 ? KP : 32) + 2;
    constexpr size_t pj_elems = (size_t)(32 * PJP > 64 * PO_P ? 32 * PJP : 64 * PO_P);
    constexpr size_t lds = sizeof(double) * pj_elems;                   // dynamic part; ~35 KiB more are static
    if (lds + 36 * 1024 > 64 * 1024) OAK_CHECK(ensure_dynamic_lds((const void*)potrf_blk_kernel<KP>, lds));
    potrf_blk_kernel<KP><<<grid, 384, lds, ctx->stream>>>(dA, n, nr, lda, j0, d_info, d_arr, nA, ntc, trc);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__global__ void zero_upper_kernel(double* __restrict__ A, int64_t n, int64_t lda) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j < n && j > i) A[i * lda + j] = 0.0;
}

static const int PO_BIG = 0x7fffffff;

// status word of a factorisation as a kernel left it in "potrf_info" (read by the caller together with its other results)
int potrf_check_value(int info, int64_t n) {
    if (info != PO_BIG) {
        set_error("Cholesky decomposition was not successful: leading minor of order %d is not positive definite (n=%lld)", info, (long long)n);
        return OAK_E_NOTPD;
    }
    return OAK_OK;
}
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

__global__ void potrf_reset_kernel(int* info, int info_value, int* arrivals, int npanel) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *info = info_value;
    if (i < npanel) arrivals[i] = 0;
}

int potrf_lower(oak_ctx* ctx, double* dA, int64_t n, int64_t lda, bool check, int64_t nrows, bool identity_below) {
    if (nrows < n) nrows = n;
    if (nrows > n && (n % PO_NB) != 0) { set_error("potrf_lower: extra rows need n to be a multiple of %d", PO_NB); return OAK_E_ARG; }
    int* d_info = nullptr;
    OAK_CHECK(get_buf_t(ctx, "potrf_info", 2, &d_info));
    const int slot = (ctx->side != nullptr && ctx->stream == ctx->side) ? 1 : 0;
    d_info += slot;
    int* d_arr = nullptr;
    const size_t npanel = (size_t)((n + PO_NB - 1) / PO_NB);
    OAK_CHECK(get_buf_t(ctx, slot ? "potrf_arrivals_side" : "potrf_arrivals", npanel, &d_arr));
    potrf_reset_kernel<<<(unsigned)((npanel + 255) / 256), 256, 0, ctx->stream>>>(d_info, PO_BIG, d_arr, (int)npanel);   // status + arrival counters in one launch
    // Fused mode (one launch per panel) is launch-latency optimal and wins up to n ~ 6000; beyond that the trailing updates
    // are HBM-bound and the pipelined 64 x 64 GEMM kernel moves them faster than the single-stage role-B tiles
    // (n = 16384: 253 ms fused vs 182 ms), so large factorisations fall back to panel + GEMM per step.
    const bool fused = n <= 6144;
    // dev aid: in-kernel time stamps, 24 per step.  Only while oak_bench_potrf has armed it (ctx->potrf_trace, with its capacity),
    // only for the main-stream factorisation it times, and never past the end of the buffer.
    long long* d_trace = (slot == 0) ? ctx->potrf_trace : nullptr;
    // one-level right-looking step (potrf_step_kernel), kept for A/B runs and for the in-kernel time stamps of oak_bench_potrf
    bool two_level = true;
    if (const char* e = getenv("OAK_POTRF_LEVELS")) two_level = atoi(e) != 1;
    for (int64_t j0 = 0; j0 < n; j0 += PO_NB) {
        long long* trc = (d_trace && j0 / PO_NB < ctx->potrf_trace_steps) ? d_trace + 24 * (j0 / PO_NB) : nullptr;
        // identity_below: the extra rows are the identity that the panel solves turn into L^-T (chol_with_inverse).  Row i of that
        // block is still e_i -- zero in every column < i, and so is every update it would receive -- until panel i / 32 reaches it:
        // a step only has to touch the extra rows [0, j0 + 32).  (M = 1024: the early steps fit the chip in one round of
        // workgroups instead of two, 18.5 -> 12 us each.)
        const int64_t nr = (identity_below && nrows > n && n + j0 + PO_NB < nrows) ? n + j0 + PO_NB : nrows;
        const int64_t below_rows = nr - j0 - PO_NB;        // rows under the diagonal block (extra rows included)
        const int nA = below_rows > 0 ? (int)((below_rows + PO_RPW - 1) / PO_RPW) : 1;
        const int64_t tc = n - j0 - PO_NB, tr = nr - j0 - PO_NB;
        if (two_level) {
            const int jj = (int)((j0 % PO_BLK) / PO_NB);                // panel index inside its outer block
            int* arr = d_arr + j0 / PO_NB;
            if (jj == 0 && j0 > 0) {
                if (fused) {
                    // pending = the whole previous block; role B applies it to the trailing matrix right of panel j
                    const int ntc = tc > 0 ? (int)((tc + 63) / 64) : 0, ntr = tc > 0 ? (int)((tr + 63) / 64) : 0;
                    OAK_CHECK(potrf_blk_launch<PO_BLK>(ctx, (unsigned)(nA + ntr * ntc), dA, n, nr, lda, j0, d_info, arr, nA, ntc > 0 ? ntc : 1, trc));
                } else {
                    // large n: the previous block was applied to everything behind it by the pipelined GEMM below
                    OAK_CHECK(potrf_blk_launch<0>(ctx, (unsigned)nA, dA, n, nr, lda, j0, d_info, arr, nA, 1, trc));
                }
            } else if (jj == 0) {
                OAK_CHECK(potrf_blk_launch<0>(ctx, (unsigned)nA, dA, n, nr, lda, j0, d_info, arr, nA, 1, trc));
            } else if (jj == 1) {
                OAK_CHECK(potrf_blk_launch<32>(ctx, (unsigned)nA, dA, n, nr, lda, j0, d_info, arr, nA, 1, trc));
            } else if (jj == 2) {
                OAK_CHECK(potrf_blk_launch<64>(ctx, (unsigned)nA, dA, n, nr, lda, j0, d_info, arr, nA, 1, trc));
            } else {
                OAK_CHECK(potrf_blk_launch<96>(ctx, (unsigned)nA, dA, n, nr, lda, j0, d_info, arr, nA, 1, trc));
            }
            if (!fused && jj == PO_BLK / PO_NB - 1 && tc > 0) {        // A22 -= L21 L21^T (lower tiles only), K = 128, behind the finished block
                const int64_t b0 = j0 + PO_NB - PO_BLK;
                const double* L21 = dA + (j0 + PO_NB) * lda + b0;
                double* A22 = dA + (j0 + PO_NB) * lda + (j0 + PO_NB);
                OAK_CHECK(gemm_nt(ctx, L21, L21, A22, tr, tc, PO_BLK, lda, lda, lda, -1.0, 1.0, 1));
            }
            continue;
        }
        if (fused) {
            // role B applies panel j-1 to the trailing matrix behind panel j: columns >= j0 + 32 (none on the first step)
            const int ntc = (j0 > 0 && tc > 0) ? (int)((tc + 63) / 64) : 0;
            const int ntr = (j0 > 0 && tc > 0) ? (int)((tr + 63) / 64) : 0;
            potrf_step_kernel<<<(unsigned)(nA + ntr * ntc), 320, 0, ctx->stream>>>(dA, n, nr, lda, j0, d_info, d_arr + j0 / PO_NB, nA,
                                                                                     ntc > 0 ? ntc : 1, 1, trc);
        } else {
            potrf_step_kernel<<<(unsigned)nA, 320, 0, ctx->stream>>>(dA, n, nr, lda, j0, d_info, d_arr + j0 / PO_NB, nA, 1, 0, trc);
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
// One right-hand side, L x = b, with the inverses of L's 128 x 128 diagonal blocks at hand (the diagonal blocks of the L^-1 that
// rode through the factorisation): x_j = inv(L_jj) (b_j - L_{j,<j} x_{<j}) -- the arithmetic of the fused many-row solve
// (trsm_fused.hip), i.e. the inverse is only ever applied to an already-cancelled residual.  One workgroup, x kept in LDS; a
// wave owns eight rows of the block and its lanes stride over k (1 KiB coalesced row segments), wave-reduced in a fixed
// order.  Replaces 15 dependent launches (8 leaf solves + 7 GEMMs, 0.3 ms at n = 1024) by one of ~40 us.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) trsv_blockinv_kernel(const double* __restrict__ L, int64_t n, int64_t ldl, const double* __restrict__ Linv,
                                                             int64_t ldinv, double* __restrict__ b) {
    extern __shared__ __attribute__((aligned(16))) double xs[];      // [n] solution so far (zero beyond it), then [128] residual of the current block
    double* rb = xs + n;
    double* bs = rb + 128;                                            // [n] the right-hand side
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 16 waves, eight rows of the current block each
    auto wave_sum = [](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    for (int64_t i = tid; i < n; i += 1024) { xs[i] = 0.0; bs[i] = b[i]; }
    __syncthreads();
    for (int64_t j0 = 0; j0 < n; j0 += 128) {
        // The chain is latency-bound (one workgroup, L comes from HBM): 16 independent 16-byte loads per lane are issued before
        // the first is used -- eight rows x two 128-column chunks (four would spill at the 128 VGPRs of a 1024-thread workgroup).  Chunks at or beyond j0 meet x = 0 (L is finite there), so the
        // chunk loop needs no tail case; it stops at the matrix edge.
        const double* Lr = L + (j0 + 8 * wave) * ldl + 2 * lane;
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int64_t k0 = 0; k0 < j0; k0 += 256) {
            double2 lv[8][2], xv[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int64_t k = (k0 + 128 * c < n) ? k0 + 128 * c : k0;       // a repeated chunk is multiplied by zero below
#pragma unroll
                for (int q = 0; q < 8; ++q) lv[q][c] = *reinterpret_cast<const double2*>(Lr + q * ldl + k);
            }
            __builtin_amdgcn_sched_barrier(0);          // all 16 loads leave before anything waits on the first
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bool in = k0 + 128 * c < j0;
                const double2 x2 = *reinterpret_cast<const double2*>(&xs[(k0 + 128 * c < n ? k0 + 128 * c : k0) + 2 * lane]);
                xv[c] = make_double2(in ? x2.x : 0.0, in ? x2.y : 0.0);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    s[q] = __builtin_fma(lv[q][c].x, xv[c].x, s[q]);
                    s[q] = __builtin_fma(lv[q][c].y, xv[c].y, s[q]);
                }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s[q] = wave_sum(s[q]);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) rb[8 * wave + q] = bs[j0 + 8 * wave + q] - s[q];
        }
        __syncthreads();
        {
            const double2 rv = *reinterpret_cast<const double2*>(&rb[2 * lane]);
            const double* Ir = Linv + (j0 + 8 * wave) * ldinv + j0 + 2 * lane;
            double2 iv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) iv[q] = *reinterpret_cast<const double2*>(Ir + q * ldinv);
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = wave_sum(__builtin_fma(iv[q].y, rv.y, iv[q].x * rv.x));
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) xs[j0 + 8 * wave + q] = t[q];
            }
        }
        __syncthreads();
    }
    for (int64_t i = tid; i < n; i += 1024) b[i] = xs[i];
}

// n a multiple of 128 and <= 8192 (LDS); dLinv = full row-major inverse of L (only its diagonal blocks are read)
int trsv_lower_blockinv(oak_ctx* ctx, const double* dL, int64_t n, int64_t ldl, const double* dLinv, int64_t ldinv, double* d_b) {
    OAK_REQUIRE(n > 0 && n % 128 == 0 && n <= 8192 && (ldl % 2) == 0 && (ldinv % 2) == 0, "trsv_lower_blockinv: n=%lld not supported", (long long)n);
    const size_t lds = sizeof(double) * (size_t)(2 * n + 128);
    if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)trsv_blockinv_kernel));
    trsv_blockinv_kernel<<<1, 1024, lds, ctx->stream>>>(dL, n, ldl, dLinv, ldinv, d_b);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
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
        if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)kern));
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
        // whole column blocks: one launch over all of them (trsm_fused.hip), in place.  (Callers whose panels carry zero-padded
        // columns up to the next multiple of 128 call trsm_rows_fused themselves.)
        if (nrhs >= 4096 && n <= 4096 && (n % NB) == 0 && (ldb % 2) == 0 && ((uintptr_t)dBT & 15) == 0 &&
            getenv("OAK_TRSM_UNFUSED") == nullptr)
            return trsm_rows_fused(ctx, dL, n, ldl, nullptr, 0, nullptr, dBT, ldb, dBT, ldb, nrhs);
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
        // whole column blocks: the fused kernel on column-reversed panels against J L^T J (trsm_fused.hip), in place
        if (nrhs >= 4096 && n <= 4096 && (n % NB) == 0 && (ldb % 2) == 0 && ((uintptr_t)dBT & 15) == 0 && getenv("OAK_TRSM_UNFUSED") == nullptr)
            return trsm_rows_fused(ctx, dL, n, ldl, nullptr, 0, nullptr, dBT, ldb, dBT, ldb, nrhs, true);
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

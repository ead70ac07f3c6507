// Dense fp64 linear algebra for the SGPR/GPR solve path (replaces the tf.linalg calls mirrored at
// oak/utils.py:187-198: cholesky, triangular_solve, matmul).
//
//  * syrk_panel : Phi += P^T P for a row panel P [nrows x M] of Kfu, fp64 MFMA (v_mfma_f64_16x16x4),
//                 128x128 tiles of the upper triangle x split-N, LDS-staged operands, partials reduced in fixed order.
//                 This is the dominant kernel of an ELBO evaluation: M(M+1)N flops.
//  * potrf_lower: blocked right-looking Cholesky (NB = 32), panel solve + trailing update.
//  * trsm_rows  : blocked forward/backward substitution, each right-hand side a contiguous row.
#include "oak_internal.h"
#include <cstdlib>

namespace oak {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// SYRK
// ---------------------------------------------------------------------------------------------
constexpr int SY_T = 128;          // Phi tile edge
constexpr int SY_KB = 16;          // panel rows per LDS stage
constexpr int SY_LD = SY_T + 16;   // LDS row stride (doubles): k-group rows land 32 banks apart -> conflict-free ds_read_b64

__global__ void __launch_bounds__(256, 2)
syrk_kernel(const double* __restrict__ P, int64_t ldp, int64_t nrows, int ntile, int nsplit, int64_t rows_per_split,
            double* __restrict__ part, int64_t Mp, int accumulate) {
    __shared__ __attribute__((aligned(16))) double As[SY_KB * SY_LD];
    __shared__ __attribute__((aligned(16))) double Bs[SY_KB * SY_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int pair = blockIdx.x / nsplit;
    const int split = blockIdx.x - pair * nsplit;
    int bi = 0, rem = pair;
    while (rem >= ntile - bi) { rem -= ntile - bi; ++bi; }
    const int bj = bi + rem;
    const bool diag = (bi == bj);
    const int64_t r0 = (int64_t)split * rows_per_split;
    int64_t r1 = r0 + rows_per_split;
    if (r1 > nrows) r1 = nrows;

    double4_t acc[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[g][h] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // global -> register staging: element e = 2*tid + 512*q -> row e/128, col e%128 (one wave-load = one 1 KiB row)
    const int lrow = (2 * tid) >> 7;          // 0..3
    const int lcol = (2 * tid) & 127;
    const double* pa = P + (int64_t)bi * SY_T + lcol;
    const double* pb = P + (int64_t)bj * SY_T + lcol;
    double2 ra[4], rb[4];
    auto load_stage = [&](int64_t n0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = n0 + lrow + 4 * q;
            if (row < r1) {
                ra[q] = *reinterpret_cast<const double2*>(pa + row * ldp);
                if (!diag) rb[q] = *reinterpret_cast<const double2*>(pb + row * ldp);
            } else {
                ra[q] = make_double2(0.0, 0.0);
                rb[q] = make_double2(0.0, 0.0);
            }
        }
    };
    if (r0 < r1) load_stage(r0);
    const double* Bsrc = diag ? As : Bs;
    const int fr = lane & 15, fk = lane >> 4;
    for (int64_t n0 = r0; n0 < r1; n0 += SY_KB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<double2*>(&As[(lrow + 4 * q) * SY_LD + lcol]) = ra[q];
            if (!diag) *reinterpret_cast<double2*>(&Bs[(lrow + 4 * q) * SY_LD + lcol]) = rb[q];
        }
        __syncthreads();
        if (n0 + SY_KB < r1) load_stage(n0 + SY_KB);
#pragma unroll
        for (int kk = 0; kk < SY_KB / 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g] = As[(4 * kk + fk) * SY_LD + 64 * wr + 16 * g + fr];
#pragma unroll
            for (int h = 0; h < 4; ++h) b[h] = Bsrc[(4 * kk + fk) * SY_LD + 64 * wc + 16 * h + fr];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int h = 0; h < 4; ++h)
                    acc[g][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g], b[h], acc[g][h], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: f64 16x16x4 C/D layout: col = lane&15, row = (lane>>4) + 4*reg
    double* dst = part + (int64_t)split * Mp * Mp;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t row = (int64_t)bi * SY_T + 64 * wr + 16 * g + 4 * reg + fk;
                const int64_t col = (int64_t)bj * SY_T + 64 * wc + 16 * h + fr;
                double* q = dst + row * Mp + col;
                const double v = acc[g][h][reg];
                *q = accumulate ? (*q + v) : v;
            }
}

__global__ void __launch_bounds__(256) syrk_reduce_kernel(const double* __restrict__ part, int nsplit, int64_t M, int64_t Mp,
                                                          double* __restrict__ phi, int accumulate) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= M) return;
    const bool upper = (i / SY_T) <= (j / SY_T);
    const int64_t src = upper ? (i * Mp + j) : (j * Mp + i);
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += part[(int64_t)sp * Mp * Mp + src];
    if (accumulate) phi[i * M + j] += s; else phi[i * M + j] = s;
}

int syrk_plan_splits(oak_ctx* ctx, int64_t M) {
    const int ntile = (int)((M + SY_T - 1) / SY_T);
    const int npairs = ntile * (ntile + 1) / 2;
    int per_cu = 2;
    if (const char* e = getenv("OAK_SYRK_WG_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 4) per_cu = v; }
    int nsplit = (ctx->num_cu * per_cu) / npairs;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 256) nsplit = 256;
    return nsplit;
}

int syrk_panel(oak_ctx* ctx, const double* d_panel, int64_t ldp, int64_t nrows, int64_t M, double* d_part, int nsplit, bool accumulate) {
    const int ntile = (int)((M + SY_T - 1) / SY_T);
    const int64_t Mp = (int64_t)ntile * SY_T;
    OAK_REQUIRE(ldp == Mp, "syrk: panel stride %lld must equal padded M %lld", (long long)ldp, (long long)Mp);
    const int npairs = ntile * (ntile + 1) / 2;
    int64_t rps = (nrows + nsplit - 1) / nsplit;
    rps = ((rps + SY_KB - 1) / SY_KB) * SY_KB;
    if (rps < SY_KB) rps = SY_KB;
    syrk_kernel<<<(unsigned)(npairs * nsplit), 256, 0, ctx->stream>>>(d_panel, ldp, nrows, ntile, nsplit, rps, d_part, Mp, accumulate ? 1 : 0);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

int syrk_reduce(oak_ctx* ctx, const double* d_part, int nsplit, int64_t M, double* d_phi, bool accumulate) {
    const int ntile = (int)((M + SY_T - 1) / SY_T);
    const int64_t Mp = (int64_t)ntile * SY_T;
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)M);
    syrk_reduce_kernel<<<grid, 256, 0, ctx->stream>>>(d_part, nsplit, M, Mp, d_phi, accumulate ? 1 : 0);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// ---------------------------------------------------------------------------------------------
// Cholesky (lower), blocked right-looking, NB = 32
// ---------------------------------------------------------------------------------------------
constexpr int PO_NB = 32;

__global__ void __launch_bounds__(256) potrf_panel_kernel(double* __restrict__ A, int64_t n, int64_t lda, int64_t j0, int* __restrict__ info) {
    __shared__ double Dg[PO_NB][PO_NB + 1];
    const int tid = threadIdx.x;
    const int nb = (n - j0 < PO_NB) ? (int)(n - j0) : PO_NB;
    for (int idx = tid; idx < PO_NB * PO_NB; idx += 256) {
        const int i = idx / PO_NB, j = idx - i * PO_NB;
        double v = (i == j) ? 1.0 : 0.0;                       // identity padding beyond nb
        if (i < nb && j < nb) v = (j <= i) ? A[(j0 + i) * lda + j0 + j] : 0.0;
        Dg[i][j] = v;
    }
    __syncthreads();
    for (int k = 0; k < PO_NB; ++k) {
        if (tid == 0) {
            const double d = Dg[k][k];
            if (!(d > 0.0) && blockIdx.x == 0) atomicMin(info, (int)(j0 + k + 1));
            Dg[k][k] = sqrt(d);
        }
        __syncthreads();
        if (tid > k && tid < PO_NB) Dg[tid][k] /= Dg[k][k];
        __syncthreads();
        for (int idx = tid; idx < PO_NB * PO_NB; idx += 256) {
            const int i = idx / PO_NB, j = idx - i * PO_NB;
            if (j > k && i >= j) Dg[i][j] -= Dg[i][k] * Dg[j][k];
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int idx = tid; idx < nb * nb; idx += 256) {
            const int i = idx / nb, j = idx - i * nb;
            A[(j0 + i) * lda + j0 + j] = (j <= i) ? Dg[i][j] : 0.0;
        }
    }
    // rows below the diagonal block: X * L_jj^T = A_panel, one row per thread
    const int64_t row = j0 + nb + (int64_t)blockIdx.x * 256 + tid;
    if (row < n && nb == PO_NB) {
        double x[PO_NB];
        double* ap = A + row * lda + j0;
#pragma unroll
        for (int k = 0; k < PO_NB; ++k) x[k] = ap[k];
#pragma unroll
        for (int k = 0; k < PO_NB; ++k) {
            double s = x[k];
#pragma unroll
            for (int p = 0; p < k; ++p) s = __builtin_fma(-x[p], Dg[k][p], s);
            x[k] = s / Dg[k][k];
        }
#pragma unroll
        for (int k = 0; k < PO_NB; ++k) ap[k] = x[k];
    }
}

// trailing update A[i][j] -= sum_p X[i][p] X[j][p], lower-triangle 64x64 tiles
__global__ void __launch_bounds__(256) potrf_update_kernel(double* __restrict__ A, int64_t n, int64_t lda, int64_t j0) {
    __shared__ double Xi[64][PO_NB + 1];
    __shared__ double Xj[64][PO_NB + 1];
    const int tid = threadIdx.x;
    int ti = 0, rem = blockIdx.x;
    while (rem > ti) { rem -= ti + 1; ++ti; }
    const int tj = rem;                                  // tj <= ti
    const int64_t base = j0 + PO_NB;
    const int64_t i0 = base + (int64_t)ti * 64, c0 = base + (int64_t)tj * 64;
    for (int idx = tid; idx < 64 * PO_NB; idx += 256) {
        const int r = idx / PO_NB, p = idx - r * PO_NB;
        Xi[r][p] = (i0 + r < n) ? A[(i0 + r) * lda + j0 + p] : 0.0;
        Xj[r][p] = (c0 + r < n) ? A[(c0 + r) * lda + j0 + p] : 0.0;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
#pragma unroll 8
    for (int p = 0; p < PO_NB; ++p) {
        double a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = Xi[ty * 4 + r][p];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = Xj[tx * 4 + c][p];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_fma(a[r], b[c], acc[r][c]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t row = i0 + ty * 4 + r, col = c0 + tx * 4 + c;
            if (row < n && col <= row) A[row * lda + col] -= acc[r][c];
        }
}

__global__ void zero_upper_kernel(double* __restrict__ A, int64_t n, int64_t lda) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j < n && j > i) A[i * lda + j] = 0.0;
}

int potrf_lower(oak_ctx* ctx, double* dA, int64_t n, int64_t lda) {
    int* d_info = nullptr;
    OAK_CHECK(get_buf_t(ctx, "potrf_info", 1, &d_info));
    const int big = 0x7fffffff;
    OAK_HIP_CHECK(hipMemcpyAsync(d_info, &big, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    for (int64_t j0 = 0; j0 < n; j0 += PO_NB) {
        const int64_t below = n - j0 - PO_NB;
        const unsigned gp = below > 0 ? (unsigned)((below + 255) / 256) : 1u;
        potrf_panel_kernel<<<gp, 256, 0, ctx->stream>>>(dA, n, lda, j0, d_info);
        if (below > 0) {
            const int64_t nt = (below + 63) / 64;
            potrf_update_kernel<<<(unsigned)(nt * (nt + 1) / 2), 256, 0, ctx->stream>>>(dA, n, lda, j0);
        }
    }
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
    zero_upper_kernel<<<grid, 256, 0, ctx->stream>>>(dA, n, lda);
    OAK_HIP_CHECK(hipGetLastError());
    int info = 0;
    OAK_HIP_CHECK(hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (info != big) {
        set_error("Cholesky decomposition was not successful: leading minor of order %d is not positive definite (n=%lld)", info, (long long)n);
        return OAK_E_NOTPD;
    }
    return OAK_OK;
}

// ---------------------------------------------------------------------------------------------
// rows-TRSM: nrhs right-hand sides, each a contiguous row of BT; L x = b (TRANS=0) or L^T x = b (TRANS=1)
// ---------------------------------------------------------------------------------------------
template <int TRANS>
__global__ void __launch_bounds__(256) trsm_rows_kernel(const double* __restrict__ L, int64_t n, int64_t ldl,
                                                        double* __restrict__ BT, int64_t nrhs, int64_t ldb, int nblk) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Lt = sm;                       // [32][33]
    double* xs = sm + 32 * 33;             // [RB][nblk*32]
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const int rr = tid >> 5, i = tid & 31;
    const int64_t npad = (int64_t)nblk * 32;
    const int64_t rhs = (int64_t)blockIdx.x * (nthr >> 5) + rr;
    const bool valid = rhs < nrhs;
    double* xr = xs + (int64_t)rr * npad;
    for (int64_t k = i; k < npad; k += 32) xr[k] = (valid && k < n) ? BT[rhs * ldb + k] : 0.0;
    auto load_block = [&](int rb, int cb) {   // Lt[a][b] = L[rb*32+a][cb*32+b], identity padding
        for (int idx = tid; idx < 32 * 32; idx += nthr) {
            const int a = idx >> 5, b = idx & 31;
            const int64_t gr = (int64_t)rb * 32 + a, gc = (int64_t)cb * 32 + b;
            double v = (gr == gc) ? 1.0 : 0.0;
            if (gr < n && gc < n) v = (gc <= gr) ? L[gr * ldl + gc] : 0.0;
            Lt[a * 33 + b] = v;
        }
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
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                const double xp = __shfl(s, p, 32) / Lt[p * 33 + p];
                if (i == p) x = xp;
                if (i > p) s = __builtin_fma(-Lt[i * 33 + p], xp, s);
            }
            xr[jb * 32 + i] = x;
        }
    } else {
        for (int jb = nblk - 1; jb >= 0; --jb) {
            __syncthreads();
            double s = xr[jb * 32 + i];
            for (int kb = nblk - 1; kb > jb; --kb) {
                __syncthreads();
                load_block(kb, jb);          // Lt[k2][i2] = L[kb*32+k2][jb*32+i2]
                __syncthreads();
                const double* xk = xr + kb * 32;
#pragma unroll
                for (int k2 = 0; k2 < 32; ++k2) s = __builtin_fma(-Lt[k2 * 33 + i], xk[k2], s);
            }
            __syncthreads();
            load_block(jb, jb);
            __syncthreads();
            double x = 0.0;
#pragma unroll
            for (int p = 31; p >= 0; --p) {
                const double xp = __shfl(s, p, 32) / Lt[p * 33 + p];
                if (i == p) x = xp;
                if (i < p) s = __builtin_fma(-Lt[p * 33 + i], xp, s);
            }
            xr[jb * 32 + i] = x;
        }
    }
    __syncthreads();
    for (int64_t k = i; k < n; k += 32) if (valid) BT[rhs * ldb + k] = xr[k];
}

int trsm_rows(oak_ctx* ctx, const double* dL, int64_t n, int64_t ldl, double* dBT, int64_t nrhs, int64_t ldb, int trans) {
    if (n <= 0 || nrhs <= 0) return OAK_OK;
    const int nblk = (int)((n + 31) / 32);
    int rb = 8;
    while (rb > 1 && sizeof(double) * ((size_t)rb * nblk * 32 + 32 * 33) > 150 * 1024) rb >>= 1;
    const size_t lds = sizeof(double) * ((size_t)rb * nblk * 32 + 32 * 33);
    OAK_REQUIRE(lds <= 160 * 1024, "trsm: n=%lld too large for the LDS-resident solver", (long long)n);
    if (nrhs < rb) { rb = 1; while (rb * 2 <= nrhs) rb *= 2; }
    const size_t lds2 = sizeof(double) * ((size_t)rb * nblk * 32 + 32 * 33);
    const unsigned grid = (unsigned)((nrhs + rb - 1) / rb);
    if (trans) {
        if (lds2 > 64 * 1024) OAK_HIP_CHECK(hipFuncSetAttribute((const void*)trsm_rows_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        trsm_rows_kernel<1><<<grid, rb * 32, lds2, ctx->stream>>>(dL, n, ldl, dBT, nrhs, ldb, nblk);
    } else {
        if (lds2 > 64 * 1024) OAK_HIP_CHECK(hipFuncSetAttribute((const void*)trsm_rows_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        trsm_rows_kernel<0><<<grid, rb * 32, lds2, ctx->stream>>>(dL, n, ldl, dBT, nrhs, ldb, nblk);
    }
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) transpose_kernel(const double* __restrict__ A, int64_t rows, int64_t cols, int64_t lda,
                                                        double* __restrict__ B, int64_t ldb) {
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + k, c = c0 + tx;
        t[k][tx] = (r < rows && c < cols) ? A[r * lda + c] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = c0 + k, c = r0 + tx;     // B is cols x rows
        if (r < cols && c < rows) B[r * ldb + c] = t[tx][k];
    }
}
int transpose(oak_ctx* ctx, const double* dA, int64_t rows, int64_t cols, int64_t lda, double* dB, int64_t ldb) {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    transpose_kernel<<<grid, 256, 0, ctx->stream>>>(dA, rows, cols, lda, dB, ldb);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__global__ void add_diag_kernel(double* A, int64_t n, int64_t lda, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[i * lda + i] += v;
}
int add_diag(oak_ctx* ctx, double* dA, int64_t n, int64_t lda, double v) {
    add_diag_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(dA, n, lda, v);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__global__ void scale_add_eye_kernel(const double* __restrict__ W, int64_t n, double s, double* __restrict__ B) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j < n) B[i * n + j] = W[i * n + j] * s + (i == j ? 1.0 : 0.0);
}
int scale_add_eye(oak_ctx* ctx, const double* dW, int64_t n, double s, double* dB) {
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
    scale_add_eye_kernel<<<grid, 256, 0, ctx->stream>>>(dW, n, s, dB);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__device__ __forceinline__ double block_reduce_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) { for (int k = 0; k < (int)(blockDim.x >> 6); ++k) r += sh[k]; }
    __syncthreads();
    return r;   // valid on thread 0
}

// mode 0: sum x ; 1: sum x^2 ; 2: sum log x ; x_k = x[k*stride]
__global__ void __launch_bounds__(256) reduce_stage1(const double* __restrict__ x, int64_t n, int64_t stride, int mode,
                                                     double* __restrict__ partial) {
    __shared__ double sh[4];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * per;
    int64_t e = b + per; if (e > n) e = n;
    double acc = 0.0;
    for (int64_t k = b + threadIdx.x; k < e; k += blockDim.x) {
        const double v = x[k * stride];
        acc += (mode == 0) ? v : (mode == 1 ? v * v : log(v));
    }
    const double r = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ void __launch_bounds__(256) reduce_stage2(const double* __restrict__ partial, int np, double* __restrict__ out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int k = threadIdx.x; k < np; k += blockDim.x) acc += partial[k];
    const double r = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = r;
}
int reduce_sum(oak_ctx* ctx, const double* d_x, int64_t n, double* d_out, int mode, int64_t stride) {
    double* d_part = nullptr;
    int nb = (int)((n + 4095) / 4096);
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    OAK_CHECK(get_buf_t(ctx, "reduce_part", 512, &d_part));
    reduce_stage1<<<nb, 256, 0, ctx->stream>>>(d_x, n, stride, mode, d_part);
    reduce_stage2<<<1, 256, 0, ctx->stream>>>(d_part, nb, d_out);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__global__ void __launch_bounds__(256) dot_stage1(const double* __restrict__ x, const double* __restrict__ y, int64_t n,
                                                  double* __restrict__ partial) {
    __shared__ double sh[4];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * per;
    int64_t e = b + per; if (e > n) e = n;
    double acc = 0.0;
    for (int64_t k = b + threadIdx.x; k < e; k += blockDim.x) acc = __builtin_fma(x[k], y[k], acc);
    const double r = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
int dot(oak_ctx* ctx, const double* d_x, const double* d_y, int64_t n, double* d_out) {
    double* d_part = nullptr;
    int nb = (int)((n + 4095) / 4096);
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    OAK_CHECK(get_buf_t(ctx, "reduce_part", 512, &d_part));
    dot_stage1<<<nb, 256, 0, ctx->stream>>>(d_x, d_y, n, d_part);
    reduce_stage2<<<1, 256, 0, ctx->stream>>>(d_part, nb, d_out);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// y_i = sum_j A_ij x_j : one wave per row
__global__ void __launch_bounds__(256) gemv_rows_kernel(const double* __restrict__ A, int64_t rows, int64_t cols, int64_t lda,
                                                        const double* __restrict__ x, double* __restrict__ y, int sq) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    double acc = 0.0;
    const double* a = A + row * lda;
    if (sq) { for (int64_t j = lane; j < cols; j += 64) acc = __builtin_fma(a[j], a[j], acc); }
    else    { for (int64_t j = lane; j < cols; j += 64) acc = __builtin_fma(a[j], x[j], acc); }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) y[row] = acc;
}
int gemv_rows(oak_ctx* ctx, const double* dA, int64_t rows, int64_t cols, int64_t lda, const double* d_x, double* d_y) {
    if (rows <= 0) return OAK_OK;
    gemv_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, ctx->stream>>>(dA, rows, cols, lda, d_x, d_y, 0);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}
int row_sumsq(oak_ctx* ctx, const double* dA, int64_t rows, int64_t cols, int64_t lda, double* d_out) {
    if (rows <= 0) return OAK_OK;
    gemv_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, ctx->stream>>>(dA, rows, cols, lda, nullptr, d_out, 1);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__global__ void axpy_kernel(double a, const double* __restrict__ x, double* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __builtin_fma(a, x[i], y[i]);
}
int axpy(oak_ctx* ctx, double a, const double* x, double* y, int64_t n) {
    axpy_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(a, x, y, n);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}
__global__ void scale_kernel(double a, double* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= a;
}
int scale_vec(oak_ctx* ctx, double a, double* x, int64_t n) {
    scale_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(a, x, n);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// C = alpha*A*B + beta*C, row-major, generic sizes; 64x64 tile, 4x4 per thread, K-step 16 (VALU; used for the
// small M x M products of the gradient tail -- fp64 MFMA and fp64 FMA share one DP pipe on gfx950).
__global__ void __launch_bounds__(256) gemm_nn_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C,
                                                      int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int64_t ldc,
                                                      double alpha, double beta) {
    __shared__ double As[16][64 + 1];
    __shared__ double Bs[16][64 + 1];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
    for (int64_t k0 = 0; k0 < k; k0 += 16) {
        for (int idx = tid; idx < 64 * 16; idx += 256) {
            const int r = idx >> 4, p = idx & 15;        // A tile: 64 rows x 16 k
            As[p][r] = (r0 + r < m && k0 + p < k) ? A[(r0 + r) * lda + k0 + p] : 0.0;
            const int p2 = idx >> 6, c = idx & 63;       // B tile: 16 k x 64 cols
            Bs[p2][c] = (k0 + p2 < k && c0 + c < n) ? B[(k0 + p2) * ldb + c0 + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            double a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = As[p][ty * 4 + r];
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = Bs[p][tx * 4 + c];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_fma(a[r], b[c], acc[r][c]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t row = r0 + ty * 4 + r, col = c0 + tx * 4 + c;
            if (row < m && col < n) {
                double* q = C + row * ldc + col;
                *q = (beta == 0.0) ? alpha * acc[r][c] : __builtin_fma(alpha, acc[r][c], beta * (*q));
            }
        }
}
int gemm_nn(oak_ctx* ctx, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k, int64_t lda,
            int64_t ldb, int64_t ldc, double alpha, double beta) {
    if (m <= 0 || n <= 0) return OAK_OK;
    dim3 grid((unsigned)((n + 63) / 64), (unsigned)((m + 63) / 64));
    gemm_nn_kernel<<<grid, 256, 0, ctx->stream>>>(dA, dB, dC, m, n, k, lda, ldb, ldc, alpha, beta);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

}  // namespace oak

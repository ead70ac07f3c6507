// fp32 STATISTICS variant of the two N-sized forward kernels (BASELINE.json config 5 asks for fp32; the reference itself is
// fp64-only, oak/oak_kernel.py:31-32, so this is an opt-in mode -- oak_sgpr_set_precision -- reported separately and never
// the headline).  Only what scales with N drops to fp32:
//   * gram32_kernel : the Kfu panel in fp32 -- same fused pair walk as gram_kernel (constrained base kernels + register ESP
//                     recurrence), the exponential by the hardware v_exp_f32 instead of the 11-instruction fp64 exp2;
//   * syrk32_kernel : Phi partials by v_mfma_f32_16x16x4_f32 (2x the fp64 MFMA rate), fp32 accumulation inside one row split
//                     only; partials leave the kernel as fp64 and are summed by the fp64 fixed-order reduction.
// Featurisation, kappa = sum K_diag, psi's final sum, the O(M^3) tail, prediction and every gradient stay fp64.
#include "oak_internal.h"
#include <cstdlib>

namespace oak {

typedef float float4_t __attribute__((ext_vector_type(4)));

template <int R>
__device__ __forceinline__ void esp_update32(float (&e)[R > 0 ? R : 1], float k) {
    if constexpr (R > 0) {
#pragma unroll
        for (int q = R - 1; q >= 1; --q) e[q] = __builtin_fmaf(k, e[q - 1], e[q]);
        e[0] += k;
    }
}

// 256 threads = 4 waves; lane tx owns 4 adjacent columns (one 16-byte store per row), wave ty owns RT rows per row-step.
template <int R, int RT>
__global__ void __launch_bounds__(256)
gram32_kernel(const DevDesc dd, const double* __restrict__ tables, const double* __restrict__ Axs, const double* __restrict__ Acn,
              int64_t a_ld, int64_t a0, int64_t na, const double* __restrict__ Bxs, const double* __restrict__ Bcn, int64_t b_ld,
              int64_t nb, float* __restrict__ out, int64_t ldo, int rows_per_wg, const double* __restrict__ yA,
              double* __restrict__ psi_part, int64_t zero_pad_to) {
    constexpr int CPT = 4, TJ = 64 * CPT, RS = 4 * RT, RR = R > 0 ? R : 1;
    extern __shared__ __attribute__((aligned(16))) float smem32[];
    const int D = dd.D;
    float* Bx = smem32;                 // [D][TJ]
    float* Bc = Bx + D * TJ;            // [D][TJ]
    float* Ax = Bc + D * TJ;            // [D][RS]
    float* Ac = Ax + D * RS;            // [D][RS]
    float* Ay = Ac + D * RS;            // [RS]
    float* Wt = Ay + RS;                // [R + 1] order weights, [D] log2 base variance
    float* Lb = Wt + (OAK_MAX_DEPTH + 1);
    const int tid = threadIdx.x, tx = tid & 63;
    const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t jb = (int64_t)blockIdx.x * TJ;
    const int64_t ib = (int64_t)blockIdx.y * rows_per_wg;
    const int64_t iend = (ib + rows_per_wg < na) ? ib + rows_per_wg : na;
    for (int idx = tid; idx < D * TJ; idx += 256) {
        const int d = idx / TJ, j = idx - d * TJ;
        const int64_t gj = jb + j;
        const bool ok = gj < nb;
        Bx[idx] = ok ? (float)Bxs[(int64_t)d * b_ld + gj] : 0.0f;
        Bc[idx] = ok ? (float)Bcn[(int64_t)d * b_ld + gj] : 0.0f;
    }
    if (tid <= OAK_MAX_DEPTH) Wt[tid] = (float)dd.w[tid];
    if (tid < D) Lb[tid] = (float)dd.log2bv[tid];
    float psi[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) psi[c] = 0.0f;
    for (int64_t i0 = ib; i0 < iend; i0 += RS) {
        __syncthreads();
        for (int idx = tid; idx < D * RS; idx += 256) {
            const int d = idx / RS, r = idx - d * RS;
            const int64_t gi = i0 + r;
            const bool ok = gi < iend;
            Ax[idx] = ok ? (float)Axs[(int64_t)d * a_ld + a0 + gi] : 0.0f;
            Ac[idx] = ok ? (float)Acn[(int64_t)d * a_ld + a0 + gi] : 0.0f;
        }
        if (yA != nullptr && tid < RS) Ay[tid] = (i0 + tid < iend) ? (float)yA[a0 + i0 + tid] : 0.0f;
        __syncthreads();
        float e[RT][CPT][RR];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < CPT; ++c)
#pragma unroll
                for (int q = 0; q < RR; ++q) e[r][c][q] = 0.0f;
        if constexpr (R > 0) {
            for (int d = 0; d < D; ++d) {
                const float4_t vx = *reinterpret_cast<const float4_t*>(&Bx[d * TJ + 4 * tx]);
                const float4_t vc = *reinterpret_cast<const float4_t*>(&Bc[d * TJ + 4 * tx]);
                float kk[RT][CPT];
                if (dd.type[d] == OAK_DIM_RBF) {
                    const float lb = Lb[d];
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        const float xa = Ax[d * RS + ty * RT + r], ca = Ac[d * RS + ty * RT + r];
#pragma unroll
                        for (int c = 0; c < CPT; ++c) {
                            const float u = xa - vx[c];
                            kk[r][c] = __builtin_fmaf(-ca, vc[c], __builtin_amdgcn_exp2f(__builtin_fmaf(-u, u, lb)));
                        }
                    }
                } else {
                    const int C = dd.ncat[d];
                    const double* tab = tables + dd.tab_off[d];
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int c = 0; c < CPT; ++c) kk[r][c] = (float)tab[(int)Ax[d * RS + ty * RT + r] * C + (int)vx[c]];
                }
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CPT; ++c) esp_update32<R>(e[r][c], kk[r][c]);
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int64_t gi = i0 + ty * RT + r;
            float kv[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                float K = Wt[0];
                if constexpr (R > 0) {
#pragma unroll
                    for (int q = 0; q < R; ++q) K = __builtin_fmaf(Wt[q + 1], e[r][c][q], K);
                }
                kv[c] = K;
            }
            if (yA != nullptr) {
                const float yv = Ay[ty * RT + r];
#pragma unroll
                for (int c = 0; c < CPT; ++c) psi[c] = __builtin_fmaf(kv[c], yv, psi[c]);
            }
            if (gi < iend && out != nullptr) {
                float* orow = out + gi * ldo;
                const int64_t gj = jb + 4 * tx;
                if (gj + 3 < nb && (ldo & 3) == 0) {
                    *reinterpret_cast<float4_t*>(orow + gj) = (float4_t){kv[0], kv[1], kv[2], kv[3]};
                } else {
#pragma unroll
                    for (int c = 0; c < CPT; ++c) {
                        if (gj + c < nb) orow[gj + c] = kv[c];
                        else if (gj + c < zero_pad_to) orow[gj + c] = 0.0f;
                    }
                }
            }
        }
    }
    if (yA != nullptr) {
        __syncthreads();
        float* red = smem32;            // [4][TJ]
#pragma unroll
        for (int c = 0; c < CPT; ++c) red[ty * TJ + 4 * tx + c] = psi[c];
        __syncthreads();
        for (int j = tid; j < TJ; j += 256) {
            const int64_t gj = jb + j;
            if (gj < nb) psi_part[(int64_t)blockIdx.y * nb + gj] = ((double)red[j] + (double)red[TJ + j]) + ((double)red[2 * TJ + j] + (double)red[3 * TJ + j]);
        }
    }
}

__global__ void __launch_bounds__(256) colsum_accum32_kernel(const double* __restrict__ part, int64_t rows, int64_t cols,
                                                             double* __restrict__ out) {
    __shared__ double red[8][33];
    const int cx = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t j = (int64_t)blockIdx.x * 32 + cx;
    double s = 0.0;
    if (j < cols) {
        int64_t r = g;
        for (; r + 56 < rows; r += 64) {          // eight loads in flight (the loop is latency-bound), summed in the same order
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = part[(r + 8 * q) * cols + j];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; r < rows; r += 8) s += part[r * cols + j];
    }
    red[g][cx] = s;
    __syncthreads();
    if (g == 0 && j < cols) {
        double t = red[0][cx];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += red[q][cx];
        out[j] += t;
    }
}

template <int R, int RT>
static int launch_gram32(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, float* d_out,
                         int64_t ldo, const double* d_yA, double* d_psi, int64_t zero_pad_to) {
    constexpr int TJ = 256, RS = 4 * RT;
    const int D = pk.dd.D;
    size_t lds = sizeof(float) * ((size_t)D * TJ * 2 + (size_t)D * RS * 2 + RS + (OAK_MAX_DEPTH + 1) + OAK_MAX_DIMS);
    if (lds < sizeof(float) * 4 * TJ) lds = sizeof(float) * 4 * TJ;
    OAK_REQUIRE(lds <= 160 * 1024, "gram32: LDS request %zu exceeds 160 KiB", lds);
    const int64_t nb = B.n, ncb = (nb + TJ - 1) / TJ;
    int64_t nrb = ((int64_t)ctx->num_cu * 16 + ncb - 1) / ncb;
    int64_t rows = (na + nrb - 1) / nrb;
    rows = ((rows + RS - 1) / RS) * RS;
    if (rows < RS) rows = RS;
    if (rows > 4096) rows = 4096;
    nrb = (na + rows - 1) / rows;
    if (nrb > 65535) { rows = ((na + 65534) / 65535 + RS - 1) / RS * RS; nrb = (na + rows - 1) / rows; }
    double* d_part = nullptr;
    if (d_yA != nullptr) OAK_CHECK(get_buf_t(ctx, "psi_part", (size_t)(nrb * nb), &d_part));
    auto kern = gram32_kernel<R, RT>;
    if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)kern));
    dim3 grid((unsigned)ncb, (unsigned)nrb);
    kern<<<grid, 256, lds, ctx->stream>>>(pk.dd, pk.d_tables, A.xs, A.cn, A.ld, a0, na, B.xs, B.cn, B.ld, nb, d_out, ldo, (int)rows,
                                          d_yA, d_part, zero_pad_to);
    OAK_HIP_CHECK(hipGetLastError());
    if (d_yA != nullptr) {
        colsum_accum32_kernel<<<(unsigned)((nb + 31) / 32), 256, 0, ctx->stream>>>(d_part, nrb, nb, d_psi);
        OAK_HIP_CHECK(hipGetLastError());
    }
    return OAK_OK;
}

// out[i * ldo + j] = (float) K(A_i, B_j); psi accumulation and zero padding as gram()
int gram_f32(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, float* d_out, int64_t ldo,
             const double* d_yA, double* d_psi, int64_t zero_pad_to) {
    if (na <= 0 || B.n <= 0) return OAK_OK;
#define OAK_G32(RR, RTT) return launch_gram32<RR, RTT>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
    switch (pk.dd.R <= 16 ? template_depth(pk.dd.R) : -1) {      // (depth > 16 never gets here: sgpr_local_stats keeps it in fp64)
        case 0: OAK_G32(0, 4) case 1: OAK_G32(1, 4) case 2: OAK_G32(2, 4) case 3: OAK_G32(3, 4) case 4: OAK_G32(4, 4)
        case 5: OAK_G32(5, 2) case 6: OAK_G32(6, 2) case 7: OAK_G32(7, 2) case 8: OAK_G32(8, 2)
        case 12: OAK_G32(12, 1) case 16: OAK_G32(16, 1)
    }
#undef OAK_G32
    set_error("gram32: unsupported depth %d", pk.dd.R);
    return OAK_E_ARG;
}

// ---------------------------------------------------------------------------------------------
// SYRK, fp32 operands: same decomposition as syrk_kernel (descriptor table of 64 x 64 blocks, four per workgroup, XCD-aware
// split mapping); each wave's 64 x 64 block as 4 x 4 tiles of v_mfma_f32_16x16x4_f32.  Partials are written as fp64.
// ---------------------------------------------------------------------------------------------
constexpr int S32_T = 128, S32_KB = 32, S32_LD = S32_T + 16, S32_DESC = 16, S32_FLUSH = 32;
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2)
syrk32_kernel(const float* __restrict__ P, int64_t ldp, int64_t nrows, const int* __restrict__ desc, int nwg, int nsplit,
              int64_t rows_per_split, double* __restrict__ part, int64_t Mp, int accumulate, int xcd_map) {
    __shared__ __attribute__((aligned(16))) float S[2][S32_KB * S32_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int unit, split;
    if (xcd_map) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int sl = j / nwg;
        unit = j - sl * nwg;
        split = xcd * (nsplit >> 3) + sl;
    } else {
        unit = blockIdx.x / nsplit;
        split = blockIdx.x - unit * nsplit;
    }
    const int* dsc = desc + unit * S32_DESC;
    const int wcode = dsc[4 + 3 * wave], wrb = dsc[5 + 3 * wave], wcb = dsc[6 + 3 * wave];
    const int ia = wcode & 3, ib = (wcode >> 2) & 3, wstore = (wcode >> 4) & 1;
    const float* Sa = S[ia >> 1] + 64 * (ia & 1);
    const float* Sb = S[ib >> 1] + 64 * (ib & 1);
    const int64_t r0 = (int64_t)split * rows_per_split;
    int64_t r1 = r0 + rows_per_split;
    if (r1 > nrows) r1 = nrows;
    // fp32 accumulation only over S32_FLUSH stages (1024 rows); then the tile is added into fp64 accumulators.  With fp32
    // accumulation over a whole split (8192 rows) the random-walk rounding of the sums (~sqrt(rows) * 6e-8) reached tr(AA^T)
    // at 3.5e-5 relative on the headline problem (ELBO 1.5e-4); with 256-row chunks ELBO 9e-7 but the flush (cvt + fp64 add
    // on the pipe the MFMAs use) cost 12 % of the kernel, with 1024-row chunks it is a quarter of that.
    float4_t acc[4][4];
    double4_t acc64[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h) { acc[g][h] = (float4_t){0.0f, 0.0f, 0.0f, 0.0f}; acc64[g][h] = (double4_t){0.0, 0.0, 0.0, 0.0}; }
    int since_flush = 0;
    // staging: 16 bytes per thread = 4 floats; 256 threads cover 8 rows of 128 columns (two 64-column panels)
    const int lrow = (4 * tid) >> 7;          // 0..7
    const int lcol = (4 * tid) & 127;
    const float* pa = P + (int64_t)dsc[lcol >> 6] * 64 + (lcol & 63);
    const float* pb = P + (int64_t)dsc[2 + (lcol >> 6)] * 64 + (lcol & 63);
    constexpr int NQ = S32_KB / 8;
    float4_t ra[NQ], rb[NQ];
    auto load_stage = [&](int64_t n0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int64_t row = n0 + lrow + 8 * q;
            const int64_t rc = row < r1 ? row : r1 - 1;
            const float4_t va = *reinterpret_cast<const float4_t*>(pa + rc * ldp);
            const float4_t vb = *reinterpret_cast<const float4_t*>(pb + rc * ldp);
            const bool ok = row < r1;
            ra[q] = ok ? va : (float4_t){0.0f, 0.0f, 0.0f, 0.0f};
            rb[q] = ok ? vb : (float4_t){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    if (r0 < r1) load_stage(r0);
    const int fr = lane & 15, fk = lane >> 4;
    for (int64_t n0 = r0; n0 < r1; n0 += S32_KB) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            *reinterpret_cast<float4_t*>(&S[0][(lrow + 8 * q) * S32_LD + lcol]) = ra[q];
            *reinterpret_cast<float4_t*>(&S[1][(lrow + 8 * q) * S32_LD + lcol]) = rb[q];
        }
        __syncthreads();
        load_stage((n0 + S32_KB < r1) ? n0 + S32_KB : r0);
#pragma unroll
        for (int kk = 0; kk < S32_KB / 4; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g] = Sa[(4 * kk + fk) * S32_LD + 16 * g + fr];
#pragma unroll
            for (int h = 0; h < 4; ++h) b[h] = Sb[(4 * kk + fk) * S32_LD + 16 * h + fr];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int h = 0; h < 4; ++h) acc[g][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g], b[h], acc[g][h], 0, 0, 0);
        }
        if (++since_flush == S32_FLUSH) {
            since_flush = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc64[g][h][reg] += (double)acc[g][h][reg];
                    acc[g][h] = (float4_t){0.0f, 0.0f, 0.0f, 0.0f};
                }
        }
        __syncthreads();
    }
    if (!wstore) return;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc64[g][h][reg] += (double)acc[g][h][reg];
    // f32 16x16x4 C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    double* dst = part + (int64_t)split * Mp * Mp;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t row = (int64_t)wrb * 64 + 16 * g + 4 * fk + reg;
                const int64_t col = (int64_t)wcb * 64 + 16 * h + fr;
                double* q = dst + row * Mp + col;
                const double v = acc64[g][h][reg];
                *q = accumulate ? (*q + v) : v;
            }
}

int syrk_panel_f32(oak_ctx* ctx, const float* d_panel, int64_t ldp, int64_t nrows, int64_t M, double* d_part, int nsplit, bool accumulate) {
    const int ntile = (int)((M + S32_T - 1) / S32_T);
    const int64_t Mp = (int64_t)ntile * S32_T;
    OAK_REQUIRE(ldp == Mp, "syrk32: panel stride %lld must equal padded M %lld", (long long)ldp, (long long)Mp);
    int* d_desc = nullptr;
    int npairs = 0;
    OAK_CHECK(syrk_descriptor_table(ctx, ntile, &d_desc, &npairs, false));   // full diagonal 64-blocks (this kernel has no diagonal-pair body)
    int64_t rps = (nrows + nsplit - 1) / nsplit;
    rps = ((rps + S32_KB - 1) / S32_KB) * S32_KB;
    if (rps < S32_KB) rps = S32_KB;
    const int xm = (nsplit % 8) == 0 ? 1 : 0;
    syrk32_kernel<<<(unsigned)(npairs * nsplit), 256, 0, ctx->stream>>>(d_panel, ldp, nrows, d_desc, npairs, nsplit, rps, d_part, Mp,
                                                                         accumulate ? 1 : 0, xm);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

}  // namespace oak

// Sobol indices and per-term predictions on the device.
// Replaces compute_sobol_oak (oak/utils.py:338-435) and its per-dimension integrals compute_L (:221-240, closed forms
// f1..f4 :116-165), compute_L_binary_kernel (:243-272), compute_L_categorical_kernel (:275-309),
// compute_L_empirical_measure (:312-335); and get_prediction_component (:491-530).
// The reference rebuilds every L_d inside the Python loop over terms (np.repeat/np.tile, no reuse); here each L_d is
// generated once on the device.  Two evaluations of the terms  c(S) * sum_ij alpha_i alpha_j prod_{d in S} L_d[i,j]:
//  * "Gram of products" (the default for many terms): with rows r = the unordered index pairs {i, j}, weights
//    w_r = alpha_i alpha_j (doubled off the diagonal) and columns col_T[r] = prod_{d in T} L_d[i,j] for the HALVES T of the
//    requested subsets (|T| <= 3: the empty product, single dims, pairs, triples), every term is ONE entry of the weighted
//    Gram matrix  G = P^T diag(w) P:  S = T1 u T2  ->  G[T1, T2].  G is two fp64-MFMA SYRKs (dense.hip) over panels scaled
//    by sqrt|w|: one over the rows with w >= 0, one over the rows with w < 0 (the index pairs of opposite-sign alphas; the
//    points are sorted by the sign of alpha, so the two row sets are two triangles and one rectangle).  All 41 448 terms of
//    a 32-dimensional depth-4 kernel at n = 2048 are a 529-column Gram over 2.1e6 rows: ~0.9e12 flop on the matrix pipe
//    instead of 5.4 TB of operand re-reads.  An order-4 term appears three times in G (ab|cd, ac|bd, ad|bc): their
//    agreement is recorded (oak_sobol_last_info) as a built-in check.
//  * one workgroup per term, a fused product-reduction over the stacked L_d (sobol_terms_kernel): subsets of more than six
//    dims, few terms, small n.
#include "oak_internal.h"
#include <cmath>
#include <algorithm>
#include <map>

namespace oak {

// ---- L_d generators (unit variance factor; the reference's variance handling is applied by the caller) --------------
// Gaussian measure N(mu, delta^2): f1 - f2 - f3 + f4 with sigma = 1 (oak/utils.py:116-165, 232-237)
__device__ __forceinline__ double sobol_f2(double x, double y, double l2, double d2, double mu, double l) {
    const double Mt = 1.0 / l2 + 1.0 / (l2 + d2);
    const double m = (mu / (l2 + d2) + x / l2) / Mt;
    const double C = x * x / l2 + mu * mu / (l2 + d2) - m * m * Mt;
    return l * sqrt((l2 + 2.0 * d2) / (d2 * Mt + 1.0)) * exp(-0.5 * C) / (l2 + d2) *
           exp(-((y - mu) * (y - mu)) / (2.0 * (l2 + d2))) * exp(-((m - mu) * (m - mu)) / (2.0 * (1.0 / Mt + d2)));
}

__global__ void __launch_bounds__(256) sobol_L_gaussian_kernel(const double* __restrict__ x, int64_t n, double l, double delta,
                                                               double mu, double* __restrict__ L) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (k >= n) return;
    const double xi = x[i], yk = x[k];
    const double l2 = l * l, d2 = delta * delta;
    const double f1 = l / sqrt(l2 + 2.0 * d2) * exp(-((xi - yk) * (xi - yk)) / (4.0 * l2)) *
                      exp(-((mu - 0.5 * (xi + yk)) * (mu - 0.5 * (xi + yk))) / (2.0 * d2 + l2));
    const double f2 = sobol_f2(xi, yk, l2, d2, mu, l);
    const double f3 = sobol_f2(yk, xi, l2, d2, mu, l);
    const double f4 = l2 * (l2 + 2.0 * d2) * sqrt((l2 + d2) / (l2 + 3.0 * d2)) / ((l2 + d2) * (l2 + d2)) *
                      exp(-((xi - mu) * (xi - mu) + (yk - mu) * (yk - mu)) / (2.0 * (l2 + d2)));
    L[i * n + k] = f1 - f2 - f3 + f4;
}

// binary: p0 g0(x) g0(y) + p1 g1(x) g1(y)   (oak/utils.py:264-269, without the leading variance)
__global__ void __launch_bounds__(256) sobol_L_binary_kernel(const double* __restrict__ x, int64_t n, double p0, double* __restrict__ L) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (k >= n) return;
    const double p1 = 1.0 - p0;
    const double xi = x[i], yk = x[k];
    L[i * n + k] = p0 * (p1 * p1 * (1.0 - xi) - p0 * p1 * xi) * (p1 * p1 * (1.0 - yk) - p0 * p1 * yk) +
                   p1 * (-p0 * p1 * (1.0 - xi) + p0 * p0 * xi) * (-p0 * p1 * (1.0 - yk) + p0 * p0 * yk);
}

// categorical: L[i,k] = sum_c B[c, x_i] B[c, x_k] p_c   (oak/utils.py:303-307); B already carries its variance factor
__global__ void __launch_bounds__(256) sobol_L_categorical_kernel(const double* __restrict__ x, int64_t n, const double* __restrict__ B,
                                                                  const double* __restrict__ p, int C, double bscale,
                                                                  double* __restrict__ L) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (k >= n) return;
    // category indices clamped to [0, C-1] exactly as the Gram / featurize path does (an inducing point moved by k-means or by
    // zfixed=False can carry an unseen code)
    int xi = (int)x[i], xk = (int)x[k];
    xi = xi < 0 ? 0 : (xi > C - 1 ? C - 1 : xi);
    xk = xk < 0 ? 0 : (xk > C - 1 ? C - 1 : xk);
    double acc = 0.0;
    for (int c = 0; c < C; ++c) acc += (B[c * C + xi] * bscale) * ((B[c * C + xk] * bscale) * p[c]);
    L[i * n + k] = acc;
}

__global__ void __launch_bounds__(256) scale_cols_kernel(const double* __restrict__ A, int64_t rows, int64_t cols,
                                                         const double* __restrict__ w, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j < cols) out[i * cols + j] = A[i * cols + j] * w[j];
}

__global__ void extract_col_kernel(const double* __restrict__ X, int64_t n, int ldx, int col, int trunc_flag,
                                   const int* __restrict__ perm, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t src = perm ? perm[i] : i;
    out[i] = trunc_flag ? trunc(X[src * ldx + col]) : X[src * ldx + col];
}

// L_d^{base} for dim d of desc into dL [n x n]; returns in *vexp how the reference's variance argument v enters:
// 2 -> v^2 (RBF, categorical), 1 -> v (binary, utils.py:266).
// d_perm != NULL: row / column i of L is point perm[i].
static int sobol_L_dim(oak_ctx* ctx, const oak_kernel_desc* desc, const PreparedKernel& pk, int d, const double* dXc, int64_t n,
                       int32_t ldx, double delta, double mu, double* dL, int* vexp, const int* d_perm = nullptr) {
    double* dx = nullptr;
    OAK_CHECK(get_buf_t(ctx, "sobol_x", (size_t)n, &dx));
    const int type = desc->dim_type[d];
    extract_col_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(dXc, n, ldx, desc->active_col[d], type != OAK_DIM_RBF, d_perm, dx);
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
    if (type == OAK_DIM_RBF) {
        *vexp = 2;
        const int meas = desc->measure[d];
        if (meas == OAK_MEAS_GAUSSIAN || meas == OAK_MEAS_NONE || meas == OAK_MEAS_UNIFORM) {
            // the reference applies the Gaussian closed form to every RBF sub-kernel that is neither empirical nor MOG (:388-400)
            sobol_L_gaussian_kernel<<<grid, 256, 0, ctx->stream>>>(dx, n, desc->lengthscale[d], delta, mu, dL);
        } else if (meas == OAK_MEAS_EMPIRICAL) {
            // L = Kxu^T diag(w) Kxu, Kxu = k_d(loc, z)   (:402-412, 330-333)
            const int K = desc->meas_k[d];
            const int32_t sub[1] = {d};
            PreparedKernel pc;
            OAK_CHECK(prepare_component(ctx, desc, sub, 1, 0, &pc));
            pc.dd.col[0] = 0;
            Feat Fz, Floc;
            OAK_CHECK(featurize(ctx, pc, dx, n, 1, "sobol_Fz", &Fz));
            OAK_CHECK(featurize(ctx, pc, pc.d_meas + desc->meas_off[d], K, 1, "sobol_Floc", &Floc));
            double *dKzl, *dKlz, *dKzlw;
            OAK_CHECK(get_buf_t(ctx, "sobol_Kzl", (size_t)n * K, &dKzl));
            OAK_CHECK(get_buf_t(ctx, "sobol_Klz", (size_t)n * K, &dKlz));
            OAK_CHECK(get_buf_t(ctx, "sobol_Kzlw", (size_t)n * K, &dKzlw));
            OAK_CHECK(gram(ctx, pc, Fz, 0, n, Floc, dKzl, K, nullptr, nullptr, 0));      // [n x K]
            OAK_CHECK(gram(ctx, pc, Floc, 0, K, Fz, dKlz, n, nullptr, nullptr, 0));      // [K x n]
            dim3 g2((unsigned)((K + 255) / 256), (unsigned)n);
            scale_cols_kernel<<<g2, 256, 0, ctx->stream>>>(dKzl, n, K, pc.d_meas + desc->meas_off[d] + K, dKzlw);
            OAK_CHECK(gemm_nn(ctx, dKzlw, dKlz, dL, n, n, K, K, n, n, 1.0, 0.0));
        } else {
            set_error("Sobol indices are not implemented for the MOG measure (oak/utils.py:413-414)");
            return OAK_E_ARG;
        }
    } else if (type == OAK_DIM_BINARY) {
        *vexp = 1;
        sobol_L_binary_kernel<<<grid, 256, 0, ctx->stream>>>(dx, n, desc->meas_p0[d], dL);
    } else {
        *vexp = 2;
        const int C = desc->meas_k[d];
        const double* dB = pk.d_meas + desc->meas_off[d];
        sobol_L_categorical_kernel<<<grid, 256, 0, ctx->stream>>>(dx, n, dB, dB + C * C, C, 1.0, dL);
    }
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// ---- all L_d of a Sobol pass in one launch -----------------------------------------------------------------------------------
// (the per-dimension kernels above serve oak_sobol_L, one matrix as the reference's helpers return it, in the reference's own
// arithmetic).  For the Gaussian closed form only f1 depends on the pair:
//     f2(x, y) = g2(x) h(y),  f3(x, y) = g2(y) h(x),  f4(x, y) = c4 h(x) h(y),   h(t) = exp(-(t - mu)^2 / (2 (l^2 + delta^2))),
// g2 = the x-dependent factors of eq. (45); with g2 and h tabulated per point an entry costs ONE exponential (f1's two merged)
// instead of nine -- 80 -> ~15 us per 2048 x 2048 matrix, and the 32 launches of a 32-input model become one.
struct SobolSlotDev {
    int type;                    // OAK_DIM_RBF (Gaussian closed form) / OAK_DIM_BINARY / OAK_DIM_CATEGORICAL; -1: filled elsewhere (empirical measure)
    int col, trunc, C, tab_off;
    double l, p0;
};

__global__ void __launch_bounds__(256) sobol_slots_prep_kernel(const double* __restrict__ X, int64_t n, int ldx, const int* __restrict__ perm,
                                                               const SobolSlotDev* __restrict__ slots, double delta, double mu,
                                                               double* __restrict__ xs, double* __restrict__ g2, double* __restrict__ hh) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= n) return;
    const SobolSlotDev sl = slots[s];
    if (sl.type < 0) return;
    const int64_t src = perm ? perm[i] : i;
    double x = X[src * ldx + sl.col];
    if (sl.trunc) x = trunc(x);
    xs[(int64_t)s * n + i] = x;
    if (sl.type == OAK_DIM_RBF) {
        const double l = sl.l, l2 = l * l, d2 = delta * delta;
        const double Mt = 1.0 / l2 + 1.0 / (l2 + d2);
        const double m = (mu / (l2 + d2) + x / l2) / Mt;
        const double C = x * x / l2 + mu * mu / (l2 + d2) - m * m * Mt;
        g2[(int64_t)s * n + i] = l * sqrt((l2 + 2.0 * d2) / (d2 * Mt + 1.0)) * exp(-0.5 * C) / (l2 + d2) *
                                 exp(-((m - mu) * (m - mu)) / (2.0 * (1.0 / Mt + d2)));
        hh[(int64_t)s * n + i] = exp(-((x - mu) * (x - mu)) / (2.0 * (l2 + d2)));
    }
}

// grid (ceil(n / 256), n, nslot); upper_only: entries left of the diagonal are not written (the Gram of products reads pairs p <= q only)
__global__ void __launch_bounds__(256) sobol_L_all_kernel(const SobolSlotDev* __restrict__ slots, const double* __restrict__ xs,
                                                          const double* __restrict__ g2, const double* __restrict__ hh, int64_t n, double delta,
                                                          double mu, const double* __restrict__ meas, int upper_only, double* __restrict__ Ls) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    const int s = blockIdx.z;
    if (upper_only && (int64_t)blockIdx.x * 256 + 255 < i) return;
    if (k >= n) return;
    const SobolSlotDev sl = slots[s];
    if (sl.type < 0) return;
    const double* x = xs + (int64_t)s * n;
    const double xi = x[i], yk = x[k];
    double v;
    if (sl.type == OAK_DIM_RBF) {
        const double l = sl.l, l2 = l * l, d2 = delta * delta;
        const double* g = g2 + (int64_t)s * n;
        const double* h = hh + (int64_t)s * n;
        const double dm = mu - 0.5 * (xi + yk);
        const double f1 = l / sqrt(l2 + 2.0 * d2) * exp(-((xi - yk) * (xi - yk)) / (4.0 * l2) - (dm * dm) / (2.0 * d2 + l2));
        const double c4 = l2 * (l2 + 2.0 * d2) * sqrt((l2 + d2) / (l2 + 3.0 * d2)) / ((l2 + d2) * (l2 + d2));
        v = f1 - g[i] * h[k] - g[k] * h[i] + c4 * (h[i] * h[k]);
    } else if (sl.type == OAK_DIM_BINARY) {
        const double p0 = sl.p0, p1 = 1.0 - p0;
        v = p0 * (p1 * p1 * (1.0 - xi) - p0 * p1 * xi) * (p1 * p1 * (1.0 - yk) - p0 * p1 * yk) +
            p1 * (-p0 * p1 * (1.0 - xi) + p0 * p0 * xi) * (-p0 * p1 * (1.0 - yk) + p0 * p0 * yk);
    } else {
        const int C = sl.C;
        const double* B = meas + sl.tab_off;
        const double* p = B + C * C;
        int ci = (int)xi, ck = (int)yk;
        ci = ci < 0 ? 0 : (ci > C - 1 ? C - 1 : ci);
        ck = ck < 0 ? 0 : (ck > C - 1 ? C - 1 : ck);
        double acc = 0.0;
        for (int c = 0; c < C; ++c) acc += B[c * C + ci] * (B[c * C + ck] * p[c]);
        v = acc;
    }
    Ls[(int64_t)s * n * n + i * n + k] = v;
}

// every used dim's L into dLs[slot]; vexp[d] as sobol_L_dim reports it
static int sobol_L_slots(oak_ctx* ctx, const oak_kernel_desc* desc, const PreparedKernel& pk, const std::vector<int>& slot, int nslot,
                         const double* dXc, int64_t n, int32_t ldx, double delta, double mu, double* dLs, std::vector<int>& vexp,
                         const int* d_perm, bool upper_only) {
    const int D = desc->num_dims;
    std::vector<SobolSlotDev> h((size_t)nslot);
    bool any_batched = false;
    for (int d = 0; d < D; ++d) {
        if (slot[d] < 0) continue;
        SobolSlotDev& sl = h[(size_t)slot[d]];
        sl = SobolSlotDev{-1, desc->active_col[d], 0, 0, 0, 1.0, 0.5};
        const int type = desc->dim_type[d];
        if (type == OAK_DIM_RBF) {
            vexp[d] = 2;
            const int meas = desc->measure[d];
            if (meas == OAK_MEAS_GAUSSIAN || meas == OAK_MEAS_NONE || meas == OAK_MEAS_UNIFORM) { sl.type = OAK_DIM_RBF; sl.l = desc->lengthscale[d]; }
            else if (meas != OAK_MEAS_EMPIRICAL) { set_error("Sobol indices are not implemented for the MOG measure (oak/utils.py:413-414)"); return OAK_E_ARG; }
        } else if (type == OAK_DIM_BINARY) {
            vexp[d] = 1; sl.type = OAK_DIM_BINARY; sl.trunc = 1; sl.p0 = desc->meas_p0[d];
        } else {
            vexp[d] = 2; sl.type = OAK_DIM_CATEGORICAL; sl.trunc = 1; sl.C = desc->meas_k[d]; sl.tab_off = desc->meas_off[d];
        }
        any_batched = any_batched || sl.type >= 0;
    }
    if (any_batched) {
        OAK_REQUIRE(n <= 65535, "oak_sobol: at most 65535 points (one grid row per point)");
        SobolSlotDev* d_slots = nullptr;
        double *d_xs, *d_g2, *d_hh;
        OAK_CHECK(get_buf_t(ctx, "sobol_slots", (size_t)nslot, &d_slots));
        OAK_CHECK(get_buf_t(ctx, "sobol_xs", (size_t)nslot * n, &d_xs));
        OAK_CHECK(get_buf_t(ctx, "sobol_g2", (size_t)nslot * n, &d_g2));
        OAK_CHECK(get_buf_t(ctx, "sobol_hh", (size_t)nslot * n, &d_hh));
        OAK_HIP_CHECK(hipMemcpyAsync(d_slots, h.data(), sizeof(SobolSlotDev) * (size_t)nslot, hipMemcpyHostToDevice, ctx->stream));
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));                  // h goes out of scope
        sobol_slots_prep_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)nslot), 256, 0, ctx->stream>>>(dXc, n, ldx, d_perm, d_slots, delta, mu,
                                                                                                           d_xs, d_g2, d_hh);
        OAK_HIP_CHECK(hipGetLastError());
        sobol_L_all_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)n, (unsigned)nslot), 256, 0, ctx->stream>>>(
            d_slots, d_xs, d_g2, d_hh, n, delta, mu, pk.d_meas, upper_only ? 1 : 0, dLs);
        OAK_HIP_CHECK(hipGetLastError());
    }
    for (int d = 0; d < D; ++d)            // empirical-measure dims: Kxu^T diag(w) Kxu through the Gram kernel, one at a time
        if (slot[d] >= 0 && h[(size_t)slot[d]].type < 0)
            OAK_CHECK(sobol_L_dim(ctx, desc, pk, d, dXc, n, ldx, delta, mu, dLs + (int64_t)slot[d] * n * n, &vexp[d], d_perm));
    return OAK_OK;
}

// one workgroup per subset: out[s] = mult[s] * sum_{i,k} alpha_i alpha_k prod_{d in S} L_d[i,k]
__global__ void __launch_bounds__(256) sobol_terms_kernel(const double* __restrict__ Ls, int64_t n, const double* __restrict__ alpha,
                                                          const int* __restrict__ subsets, const int* __restrict__ off,
                                                          const int* __restrict__ slot, const double* __restrict__ mult,
                                                          double* __restrict__ out) {
    __shared__ double sh[4];
    __shared__ int sdim[OAK_MAX_DIMS];
    const int s = blockIdx.x;
    const int len = off[s + 1] - off[s];
    if ((int)threadIdx.x < len) sdim[threadIdx.x] = slot[subsets[off[s] + threadIdx.x]];
    __syncthreads();
    const int64_t nn = n * n;
    double acc = 0.0;
    for (int64_t e = threadIdx.x; e < nn; e += blockDim.x) {
        const int64_t i = e / n, k = e - i * n;
        double p = alpha[i] * alpha[k];
        for (int q = 0; q < len; ++q) p *= Ls[(int64_t)sdim[q] * nn + e];
        acc += p;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[s] = mult[s] * (((sh[0] + sh[1]) + sh[2]) + sh[3]);
}

// ---- Gram of products ------------------------------------------------------------------------------------------------
// Row r of a sign set -> the index pair (p, q) in SORTED order (points with alpha >= 0 first: [0, npos), then the others).
//   mode 0 (weights >= 0): the upper triangle (diagonal included) of the first group, rows [0, T1), then that of the second;
//   mode 1 (weights  < 0): the npos x nneg rectangle, p in the first group, q in the second.
constexpr int SB_ROWS = 32;            // panel rows per workgroup of the builder

__device__ __forceinline__ void tri_decode(int64_t r, int64_t m, int64_t& p, int64_t& q) {
    // row-major upper triangle of an m x m block: row p starts at p*m - p(p-1)/2
    const double b = 2.0 * (double)m + 1.0;
    int64_t pp = (int64_t)floor((b - sqrt(b * b - 8.0 * (double)r)) * 0.5);
    if (pp < 0) pp = 0;
    if (pp > m - 1) pp = m - 1;
    while (pp > 0 && pp * m - pp * (pp - 1) / 2 > r) --pp;
    while (pp + 1 < m && (pp + 1) * m - (pp + 1) * pp / 2 <= r) ++pp;
    p = pp;
    q = pp + (r - (pp * m - pp * (pp - 1) / 2));
}

// panel[(r - row0) * Mp + c] = sqrt(w_r) * prod_{k < 3} Lt[r][cols[c].k]   for rows [row0, row0 + nrows) of the sign set;
// cols[c] packs three slot indices (8 bits each; slot nslot = the constant 1), columns [nc, Mp) are written as zero.
__global__ void __launch_bounds__(256) sobol_panel_kernel(const double* __restrict__ Ls, int64_t n, int nslot,
                                                          const double* __restrict__ aabs, int mode, int64_t npos, int64_t nneg,
                                                          int64_t row0, int64_t nrows, const int* __restrict__ cols, int nc, int Mp,
                                                          double* __restrict__ panel) {
    __shared__ double Lt[SB_ROWS][OAK_MAX_DIMS + 2];
    __shared__ double sw[SB_ROWS];
    __shared__ int64_t sidx[SB_ROWS];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * SB_ROWS;
    if (tid < SB_ROWS) {
        const int64_t rl = base + tid;
        double w = 0.0;
        int64_t e = 0;
        if (rl < nrows) {
            const int64_t r = row0 + rl;
            int64_t p, q;
            if (mode == 0) {
                const int64_t T1 = npos * (npos + 1) / 2;
                if (r < T1) tri_decode(r, npos, p, q);
                else { tri_decode(r - T1, nneg, p, q); p += npos; q += npos; }
                w = aabs[p] * aabs[q] * (p == q ? 1.0 : 2.0);
            } else {
                p = r / nneg;
                q = npos + (r - p * nneg);
                w = 2.0 * aabs[p] * aabs[q];
            }
            e = p * n + q;
        }
        sw[tid] = sqrt(w);
        sidx[tid] = e;
        Lt[tid][nslot] = 1.0;
    }
    __syncthreads();
    {
        const int rr = tid & (SB_ROWS - 1);
        const int64_t e = sidx[rr];
        const int64_t nn = n * n;
        for (int d = tid / SB_ROWS; d < nslot; d += 256 / SB_ROWS) Lt[rr][d] = Ls[(int64_t)d * nn + e];
    }
    __syncthreads();
    const int64_t live = (nrows - base < SB_ROWS) ? nrows - base : SB_ROWS;
    for (int c = tid; c < Mp; c += 256) {
        const bool real = c < nc;
        const int code = real ? cols[c] : 0;
        const int a = code & 255, b = (code >> 8) & 255, g = (code >> 16) & 255;
        double* dst = panel + base * Mp + c;
        for (int rr = 0; rr < (int)live; ++rr) {
            const double v = ((sw[rr] * Lt[rr][a]) * Lt[rr][b]) * Lt[rr][g];
            dst[(int64_t)rr * Mp] = real ? v : 0.0;
        }
    }
}

// out[s] = mult[s] * (Gpos - Gneg)[c1[s], c2[s]];  an order-4 term also reads its two other pairings (alt >= 0) and records the
// largest disagreement relative to |Gpos| + |Gneg| of the term (bits of a non-negative double order like integers)
__global__ void __launch_bounds__(256) sobol_gather_kernel(const double* __restrict__ G, int nc, const int* __restrict__ c12,
                                                           const int* __restrict__ alt, const double* __restrict__ mult,
                                                           const double* __restrict__ direct, int n_terms,
                                                           double* __restrict__ out, unsigned long long* __restrict__ dev_bits) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_terms) return;
    const double* Gp = G;
    const double* Gn = G + (int64_t)nc * nc;
    if (c12[2 * s] < 0) {                  // an order-1 term evaluated directly as alpha^T L_d alpha (no constant column)
        out[s] = mult[s] * direct[-1 - c12[2 * s]];
        return;
    }
    const int64_t e = (int64_t)c12[2 * s] * nc + c12[2 * s + 1];
    const double v = Gp[e] - Gn[e];
    out[s] = mult[s] * v;
    double dev = 0.0;
    const double scale = fabs(Gp[e]) + fabs(Gn[e]);
    for (int k = 0; k < 2; ++k) {
        const int a1 = alt[4 * s + 2 * k], a2 = alt[4 * s + 2 * k + 1];
        if (a1 < 0) continue;
        const int64_t ea = (int64_t)a1 * nc + a2;
        const double va = Gp[ea] - Gn[ea];
        const double dd = scale > 0.0 ? fabs(va - v) / scale : 0.0;
        dev = dd > dev ? dd : dev;
    }
    if (dev > 0.0) atomicMax(dev_bits, (unsigned long long)__double_as_longlong(dev));
}

// Direct order-1 terms alpha^T L_d alpha: one wave per row (coalesced), rows summed in a fixed order by a second pass.
__global__ void __launch_bounds__(256) sobol_quadform_rows_kernel(const double* __restrict__ Ls, int64_t n, const int* __restrict__ slots,
                                                                  const double* __restrict__ alpha, double* __restrict__ rowsum) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const double* row = Ls + (int64_t)slots[blockIdx.y] * n * n + i * n;
    // only the entries on or right of the diagonal are read (the Gram of products generates no others): L is symmetric, so
    // alpha^T L alpha = sum_i alpha_i (L_ii alpha_i + 2 sum_{k > i} L_ik alpha_k)
    double acc = 0.0;
    for (int64_t k = i + lane; k < n; k += 64) acc = __builtin_fma(k == i ? row[k] : 2.0 * row[k], alpha[k], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) rowsum[(int64_t)blockIdx.y * n + i] = alpha[i] * acc;
}
__global__ void __launch_bounds__(256) sobol_quadform_sum_kernel(const double* __restrict__ rowsum, int64_t n, double* __restrict__ out) {
    __shared__ double sh[4];
    const double* r = rowsum + (int64_t)blockIdx.x * n;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc += r[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

struct SobolPlan {
    std::vector<int> cols;                 // packed slot triples per column
    std::vector<int> c12, alt;             // per term: the two columns (or -1 - k: direct term k); the alternative pairings of an order-4 term or -1
    std::vector<int> direct;               // slots of the order-1 terms evaluated directly (no constant column)
    int maxlen = 0;
};

// Column dictionary over the halves of the requested subsets (slot indices, ascending); false when a subset has more than six
// dims or repeats one.  `forbidden` (nslot x nslot flags, may be empty): pair columns that must not be used -- a subset of three
// or four dims then takes another of its three pairings; drop_const: no constant column, order-1 terms become direct terms.
static bool sobol_make_plan(const int32_t* subsets, const int32_t* off, int n_subsets, const std::vector<int>& slot, int nslot,
                            const std::vector<char>& forbidden, bool drop_const, SobolPlan* plan) {
    const int B = nslot + 1;                         // digit nslot = "no factor" (the constant 1)
    std::vector<int> dict((size_t)B * B * B, -1);    // direct-indexed: <= 65^3 entries
    auto column = [&](const int* v, int len, bool create) {      // v ascending slot indices, len <= 3
        int idx = 0;
        uint32_t k = 0;
        for (int i = 0; i < 3; ++i) {
            const int digit = i < len ? v[i] : nslot;
            idx = idx * B + digit;
            k |= (uint32_t)digit << (8 * i);
        }
        if (dict[(size_t)idx] >= 0 || !create) return dict[(size_t)idx];
        const int id = (int)plan->cols.size();
        dict[(size_t)idx] = id;
        plan->cols.push_back((int)k);
        return id;
    };
    auto banned = [&](int a, int b) { return !forbidden.empty() && forbidden[(size_t)a * nslot + b] != 0; };
    *plan = SobolPlan();
    if (!drop_const) column(nullptr, 0, true);       // column 0: the constant (empty product)
    plan->c12.resize(2 * (size_t)n_subsets);
    plan->alt.assign(4 * (size_t)n_subsets, -1);
    for (int s = 0; s < n_subsets; ++s) {
        const int len = off[s + 1] - off[s];
        if (len < 1 || len > 6) return false;
        int v[6];
        for (int j = 0; j < len; ++j) v[j] = slot[subsets[off[s] + j]];
        std::sort(v, v + len);
        for (int j = 1; j < len; ++j) if (v[j] == v[j - 1]) return false;    // a repeated dim: the general kernel handles it
        if (len > plan->maxlen) plan->maxlen = len;
        if (len == 1 && drop_const) {
            plan->c12[2 * s] = -1 - (int)plan->direct.size();
            plan->c12[2 * s + 1] = 0;
            plan->direct.push_back(v[0]);
            continue;
        }
        if (len == 3 || len == 4) {                  // first pairing (in the order canonical, ac|bd, ad|bc / ab|c, ac|b, bc|a) free of banned pairs
            const int opt[3][4] = {{0, 1, 2, 3}, {0, 2, 1, 3}, {0, 3, 1, 2}};
            const int opt3[3][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}};
            bool placed = false;
            for (int k = 0; k < 3 && !placed; ++k) {
                int h1[2], h2[2];
                if (len == 4) { h1[0] = v[opt[k][0]]; h1[1] = v[opt[k][1]]; h2[0] = v[opt[k][2]]; h2[1] = v[opt[k][3]]; }
                else { h1[0] = v[opt3[k][0]]; h1[1] = v[opt3[k][1]]; h2[0] = v[opt3[k][2]]; h2[1] = 0; }
                if (banned(h1[0], h1[1]) || (len == 4 && banned(h2[0], h2[1]))) continue;
                plan->c12[2 * s] = column(h1, 2, true);
                plan->c12[2 * s + 1] = column(h2, len - 2, true);
                placed = true;
            }
            if (!placed) return false;
            continue;
        }
        const int h = (len + 1) / 2;
        plan->c12[2 * s] = column(v, h, true);
        plan->c12[2 * s + 1] = column(v + h, len - h, true);
    }
    for (int s = 0; s < n_subsets; ++s) {            // the other pairings of the order-4 terms, where both columns exist
        if (off[s + 1] - off[s] != 4) continue;
        int v[4];
        for (int j = 0; j < 4; ++j) v[j] = slot[subsets[off[s] + j]];
        std::sort(v, v + 4);
        const int p1[3][2] = {{v[0], v[1]}, {v[0], v[2]}, {v[0], v[3]}}, p2[3][2] = {{v[2], v[3]}, {v[1], v[3]}, {v[1], v[2]}};
        int k2 = 0;
        for (int k = 0; k < 3; ++k) {
            const int a = column(p1[k], 2, false), b = column(p2[k], 2, false);
            if (a < 0 || b < 0 || (a == plan->c12[2 * s] && b == plan->c12[2 * s + 1])) continue;
            if (k2 < 2) { plan->alt[4 * s + 2 * k2] = a; plan->alt[4 * s + 2 * k2 + 1] = b; ++k2; }
        }
    }
    return true;
}

// The SYRK works on 128-column tiles.  When the canonical plan ends a few columns above a multiple of 128 (D = 32, depth 4:
// 1 + 32 + 494 = 527 -> 640 executed), the same terms fit the lower multiple: the constant column goes (the D order-1 terms
// are evaluated directly as alpha^T L_d alpha) and `need` pair columns go -- any set of pairwise DISJOINT pairs can be
// spared, because the three pairings of four dims use six different pairs of which a matching contains at most two, never one
// from each pairing (likewise at most one of the three pairs of three dims).  The spared pairs avoid slots 0, 1 and the last
// one, whose pairs the canonical plan does not create in the first place.
static bool sobol_make_plan_budgeted(const int32_t* subsets, const int32_t* off, int n_subsets, const std::vector<int>& slot, int nslot,
                                     SobolPlan* plan) {
    const std::vector<char> none;
    if (!sobol_make_plan(subsets, off, n_subsets, slot, nslot, none, false, plan)) return false;
    const int nc = (int)plan->cols.size();
    const int rem = nc % 128, need = rem - 1;
    if (rem == 0 || nc < 128 || plan->maxlen > 4 || need > (nslot - 3) / 2) return true;
    std::vector<char> forbidden((size_t)nslot * nslot, 0);
    for (int k = 1; k <= need; ++k) forbidden[(size_t)(2 * k) * nslot + 2 * k + 1] = 1;
    SobolPlan lean;
    if (sobol_make_plan(subsets, off, n_subsets, slot, nslot, forbidden, true, &lean) && (int)lean.cols.size() <= nc - rem) *plan = lean;
    return true;
}

// modelled cost [s] of the two evaluations (fitted to MI355X runs: the terms kernel streams the stacked L_d at ~7.5 TB/s out of
// L2 / Infinity Cache, the SYRK sustains ~55 TFLOP/s on panels this narrow and the builder ~3 TB/s of panel writes)
// (a lone workgroup of the terms kernel streams its matrices at only ~4.5 GB/s -- an index division per element -- so a few
// hundred terms, which leave most CUs idle, are bound by one workgroup's time, not by the aggregate rate: 136 terms at n = 1024
// take 3.5 ms there against 0.8 ms through the Gram of products)
static double sobol_cost_terms(int64_t total_len, int64_t n, int maxlen) {
    const double nn = (double)n * (double)n;
    const double aggregate = 8.0 * (double)total_len * nn / 7.5e12, one_wg = 8.0 * (double)maxlen * nn / 4.5e9;
    return (aggregate > one_wg ? aggregate : one_wg) + 20e-6;
}
static double sobol_cost_gram(int nc, int64_t n) {
    const double Mp = (double)(((nc + 127) / 128) * 128), R = 0.5 * (double)n * (double)(n + 1);
    return R * Mp * (Mp + 128.0) / 55e12 + 8.0 * R * Mp / 3e12 + 150e-6;
}

// additive terms e_0..e_R of D stacked arrays (oak/oak_kernel.py:223-249 semantics, ESP recurrence)
__global__ void __launch_bounds__(256) additive_terms_kernel(const double* __restrict__ mats, int D, int64_t n, int R, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double e[OAK_MAX_DIMS + 1];
    for (int q = 0; q <= R; ++q) e[q] = (q == 0) ? 1.0 : 0.0;
    for (int d = 0; d < D; ++d) {
        const double k = mats[(int64_t)d * n + i];
        for (int q = R; q >= 1; --q) e[q] = __builtin_fma(k, e[q - 1], e[q]);
    }
    for (int q = 0; q <= R; ++q) out[(int64_t)q * n + i] = e[q];
}

}  // namespace oak

using namespace oak;

extern "C" {

int oak_additive_terms(oak_ctx* ctx, const double* mats, int32_t D, int64_t n, int32_t R, double* out) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(mats && out && D >= 1 && D <= OAK_MAX_DIMS && R >= 0 && R <= OAK_MAX_DIMS && n >= 0, "oak_additive_terms: bad arguments");
    if (n == 0) return OAK_OK;
    double *dm, *dout;
    OAK_CHECK(get_buf_t(ctx, "at_in", (size_t)D * n, &dm));
    OAK_CHECK(get_buf_t(ctx, "at_out", (size_t)(R + 1) * n, &dout));
    OAK_HIP_CHECK(hipMemcpyAsync(dm, mats, sizeof(double) * (size_t)D * n, hipMemcpyHostToDevice, ctx->stream));
    additive_terms_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(dm, D, n, R, dout);
    OAK_HIP_CHECK(hipGetLastError());
    OAK_HIP_CHECK(hipMemcpyAsync(out, dout, sizeof(double) * (size_t)(R + 1) * n, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OAK_OK;
}

int oak_cov_x_s(oak_ctx* ctx, const oak_kernel_desc* desc, int32_t dim, const double* X, int64_t n, int32_t ldx, double* c_out, double* var_s_out) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(desc && X && n >= 1 && dim >= 0 && dim < desc->num_dims && desc->dim_type[dim] == OAK_DIM_RBF, "oak_cov_x_s: bad arguments");
    PreparedKernel pk;
    OAK_CHECK(prepare_kernel(ctx, desc, &pk));
    double* dX = nullptr;
    OAK_CHECK(get_buf_t(ctx, "gX1", (size_t)n * ldx, &dX));
    OAK_HIP_CHECK(hipMemcpyAsync(dX, X, sizeof(double) * (size_t)n * ldx, hipMemcpyHostToDevice, ctx->stream));
    Feat F;
    OAK_CHECK(featurize(ctx, pk, dX, n, ldx, "gF1", &F));
    const double isv = pk.dm.inv_sqrt_v[dim];
    const double sv = isv > 0.0 ? 1.0 / isv : 0.0;
    if (c_out) {
        OAK_HIP_CHECK(hipMemcpyAsync(c_out, F.cn + (int64_t)dim * F.ld, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int64_t i = 0; i < n; ++i) c_out[i] *= sv;     // featurize stores c/sqrt(var_s); undo the normalisation
    }
    if (var_s_out) *var_s_out = sv * sv;
    return OAK_OK;
}

int oak_sobol_L(oak_ctx* ctx, const oak_kernel_desc* desc, int32_t dim, double v, double delta, double mu, const double* Xc,
                int64_t n, int32_t ldx, double* out) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(desc && Xc && out && n >= 1 && dim >= 0 && dim < desc->num_dims, "oak_sobol_L: bad arguments");
    PreparedKernel pk;
    OAK_CHECK(prepare_kernel(ctx, desc, &pk));
    double *dX, *dL;
    OAK_CHECK(get_buf_t(ctx, "sobol_X", (size_t)n * ldx, &dX));
    OAK_CHECK(get_buf_t(ctx, "sobol_L", (size_t)n * n, &dL));
    OAK_HIP_CHECK(hipMemcpyAsync(dX, Xc, sizeof(double) * (size_t)n * ldx, hipMemcpyHostToDevice, ctx->stream));
    int vexp = 2;
    OAK_CHECK(sobol_L_dim(ctx, desc, pk, dim, dX, n, ldx, delta, mu, dL, &vexp));
    OAK_CHECK(scale_vec(ctx, vexp == 2 ? v * v : v, dL, n * n));
    OAK_HIP_CHECK(hipMemcpyAsync(out, dL, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OAK_OK;
}

// collective: every rank of the context's communicator makes the same call; the work is sharded (pair rows of the Gram of
// products / blocks of terms) and the results summed over the ranks
static int sobol_run(oak_ctx* ctx, const oak_kernel_desc* desc, const double* Xc, int64_t n, int32_t ldx, const double* alpha,
                     const int32_t* subsets, const int32_t* subset_off, int32_t n_subsets, int32_t use_order_var, double delta,
                     double mu, double* out, bool collective) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(desc && Xc && alpha && subsets && subset_off && out && n >= 1 && n_subsets >= 0, "oak_sobol: bad arguments");
    if (n_subsets == 0) return OAK_OK;
    PreparedKernel pk;
    OAK_CHECK(prepare_kernel(ctx, desc, &pk));
    const int D = desc->num_dims;
    // which dims are used, and their slot in the stacked L buffer
    std::vector<int> slot(D, -1);
    int nslot = 0;
    const int total = subset_off[n_subsets];
    for (int t = 0; t < total; ++t) {
        const int d = subsets[t];
        OAK_REQUIRE(d >= 0 && d < D, "subset entry %d out of range", d);
        if (slot[d] < 0) slot[d] = nslot++;
    }
    for (int s = 0; s < n_subsets; ++s) {
        const int len = subset_off[s + 1] - subset_off[s];
        OAK_REQUIRE(len >= 1 && len <= OAK_MAX_DIMS, "subset %d has invalid length %d", s, len);
    }
    int nranks = 1, rank = 0;
    if (collective && ctx->comm != nullptr && ctx->nranks > 1 && !comm_is_loopback(ctx)) { nranks = ctx->nranks; rank = ctx->rank; }

    // ---- which evaluation --------------------------------------------------------------------------------------------
    SobolPlan plan;
    bool use_gram = false;
    if (ctx->sobol_path != 1 && sobol_make_plan_budgeted(subsets, subset_off, n_subsets, slot, nslot, &plan)) {
        const int nc = (int)plan.cols.size();
        // Device memory the Gram-of-products evaluation can ask for, bounded from the INPUTS alone (never from the free memory of
        // this GPU: under a communicator every rank must take the same path, or one rank's allocation failure leaves the others
        // waiting in the all-reduce): split partials (<= 256 splits of Mp x Mp), one panel launch (<= 16 GiB), the L_d stack.
        const double mp = (double)(((nc + 127) / 128) * 128), pair_rows = 0.5 * (double)n * ((double)n + 1.0);
        const double gram_bytes = 256.0 * mp * mp * 8.0 + std::min(16.0 * 1073741824.0, pair_rows * mp * 8.0) + (double)nslot * (double)n * (double)n * 8.0;
        const bool gram_fits = gram_bytes <= 96.0 * 1073741824.0;            // a third of an MI355X's 288 GB
        if (ctx->sobol_path == 2 && !gram_fits) {
            set_error("oak_sobol: the Gram-of-products evaluation of %d product columns over %lld points would need up to %.0f GiB of device memory", nc,
                      (long long)n, gram_bytes / 1073741824.0);
            return OAK_E_ARG;
        }
        use_gram = ctx->sobol_path == 2 || (gram_fits && nc <= 16384 && sobol_cost_gram(nc, n) < sobol_cost_terms(total, n, plan.maxlen));
    }
    OAK_REQUIRE(ctx->sobol_path != 2 || use_gram, "oak_sobol: the Gram-of-products evaluation needs subsets of 1..6 distinct dims");

    PhaseTimer t_all(ctx, "sobol");
    double *dX, *dLs, *dalpha, *dmult, *dout;
    OAK_CHECK(get_buf_t(ctx, "sobol_X", (size_t)n * ldx, &dX));
    OAK_CHECK(get_buf_t(ctx, "sobol_Ls", (size_t)nslot * n * n, &dLs));
    OAK_CHECK(get_buf_t(ctx, "sobol_alpha", (size_t)n, &dalpha));
    OAK_CHECK(get_buf_t(ctx, "sobol_mult", (size_t)n_subsets, &dmult));
    OAK_CHECK(get_buf_t(ctx, "sobol_out", (size_t)n_subsets, &dout));
    OAK_HIP_CHECK(hipMemcpyAsync(dX, Xc, sizeof(double) * (size_t)n * ldx, hipMemcpyHostToDevice, ctx->stream));

    // Gram of products: points sorted by the sign of alpha, |alpha| in that order
    std::vector<int> perm;
    std::vector<double> aabs;
    int64_t npos = 0;
    int* dperm = nullptr;
    double* dalpha_signed = nullptr;
    if (use_gram) {
        perm.resize((size_t)n); aabs.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) if (!(alpha[i] < 0.0)) perm[(size_t)npos++] = (int)i;
        int64_t k = npos;
        for (int64_t i = 0; i < n; ++i) if (alpha[i] < 0.0) perm[(size_t)k++] = (int)i;
        for (int64_t i = 0; i < n; ++i) aabs[(size_t)i] = fabs(alpha[perm[(size_t)i]]);
        OAK_CHECK(get_buf_t(ctx, "sobol_perm", (size_t)n, &dperm));
        OAK_HIP_CHECK(hipMemcpyAsync(dperm, perm.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        OAK_HIP_CHECK(hipMemcpyAsync(dalpha, aabs.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        if (!plan.direct.empty()) {                  // the direct order-1 terms want the signed values, in the sorted order
            for (int64_t i = 0; i < n; ++i) aabs[(size_t)i] = alpha[perm[(size_t)i]];
            OAK_CHECK(get_buf_t(ctx, "sobol_alpha_signed", (size_t)n, &dalpha_signed));
            OAK_HIP_CHECK(hipMemcpyAsync(dalpha_signed, aabs.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        }
    } else {
        OAK_HIP_CHECK(hipMemcpyAsync(dalpha, alpha, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    }
    std::vector<int> vexp(D, 2);
    {
        PhaseTimer tl(ctx, "sobol_L");
        OAK_CHECK(sobol_L_slots(ctx, desc, pk, slot, nslot, dX, n, ldx, delta, mu, dLs, vexp, dperm, use_gram));
        tl.stop();
    }
    // per-term scalar: the reference gives the first factor v = sigma2_{|S|} and the others v = 1 when variances are shared
    // (utils.py:376-380), else v = base variance of each factor (:382); v enters squared except for binary factors (:266)
    std::vector<double> mult(n_subsets, 1.0);
    for (int s = 0; s < n_subsets; ++s) {
        const int len = subset_off[s + 1] - subset_off[s];
        double m = 1.0;
        for (int j = 0; j < len; ++j) {
            const int d = subsets[subset_off[s] + j];
            double v = 1.0;
            if (use_order_var) {
                if (j == 0) {
                    OAK_REQUIRE(desc->share_var && len <= desc->max_depth, "subset order %d has no order variance", len);
                    v = desc->order_var[len];
                }
            } else {
                v = desc->base_var[d];
            }
            m *= (vexp[d] == 2) ? v * v : v;
        }
        mult[s] = m;
    }
    OAK_HIP_CHECK(hipMemcpyAsync(dmult, mult.data(), sizeof(double) * (size_t)n_subsets, hipMemcpyHostToDevice, ctx->stream));
    ctx->sobol_info[0] = use_gram ? 2.0 : 1.0; ctx->sobol_info[1] = 0.0; ctx->sobol_info[2] = 0.0; ctx->sobol_info[3] = 0.0;

    if (use_gram) {
        const int nc = (int)plan.cols.size();
        const int ntile = (nc + 127) / 128;
        const int64_t Mp = (int64_t)ntile * 128;
        const int64_t nneg = n - npos;
        const int64_t rows_set[2] = {npos * (npos + 1) / 2 + nneg * (nneg + 1) / 2, npos * nneg};
        int *dcols, *dc12, *dalt;
        double* dG;
        unsigned long long* ddev;
        OAK_CHECK(get_buf_t(ctx, "sobol_cols", (size_t)nc, &dcols));
        OAK_CHECK(get_buf_t(ctx, "sobol_c12", 2 * (size_t)n_subsets, &dc12));
        OAK_CHECK(get_buf_t(ctx, "sobol_alt", 4 * (size_t)n_subsets, &dalt));
        OAK_CHECK(get_buf_t(ctx, "sobol_G", 2 * (size_t)nc * nc + 1, &dG));
        ddev = reinterpret_cast<unsigned long long*>(dG + 2 * (size_t)nc * nc);
        OAK_HIP_CHECK(hipMemcpyAsync(dcols, plan.cols.data(), sizeof(int) * (size_t)nc, hipMemcpyHostToDevice, ctx->stream));
        OAK_HIP_CHECK(hipMemcpyAsync(dc12, plan.c12.data(), sizeof(int) * plan.c12.size(), hipMemcpyHostToDevice, ctx->stream));
        OAK_HIP_CHECK(hipMemcpyAsync(dalt, plan.alt.data(), sizeof(int) * plan.alt.size(), hipMemcpyHostToDevice, ctx->stream));
        OAK_CHECK(fill_zero(ctx, dG, sizeof(double) * (2 * (size_t)nc * nc + 1)));
        // panel chunks of <= ~2 GiB (a multiple of SB_ROWS rows); one split plan for every chunk so that the partials line up
        int64_t chunk_cap = ((int64_t)2 << 30) / (8 * Mp);
        if (const char* e = getenv("OAK_SOBOL_CHUNK_ROWS")) { const long long v = atoll(e); if (v >= SB_ROWS) chunk_cap = v; }
        chunk_cap = (chunk_cap / SB_ROWS) * SB_ROWS;
        if (chunk_cap < SB_ROWS) chunk_cap = SB_ROWS;
        int64_t rows_mine_max = 0;
        for (int m = 0; m < 2; ++m) {
            const int64_t per = (rows_set[m] + nranks - 1) / nranks;
            if (per > rows_mine_max) rows_mine_max = per;
        }
        // Both sign sets in ONE launch when their panels fit one buffer (<= 16 GiB; C5: 8.6 GB): the matrix is only a few tiles
        // wide, so a launch per set pays its ramp-up and ragged end twice (M = 512: 53.4 TFLOP/s for 1M rows, 55.7 for 2M).
        {
            int64_t lo2[2], hi2[2];
            for (int m = 0; m < 2; ++m) {
                const int64_t per = (rows_set[m] + nranks - 1) / nranks;
                lo2[m] = std::min<int64_t>(rows_set[m], per * rank); hi2[m] = std::min<int64_t>(rows_set[m], lo2[m] + per);
            }
            const int64_t rA = hi2[0] - lo2[0], rB = hi2[1] - lo2[1];
            const bool forced_chunks = getenv("OAK_SOBOL_CHUNK_ROWS") != nullptr;
            if (!forced_chunks && rA + rB > 0 && (double)(rA + rB) * (double)Mp * 8.0 <= 16.0 * 1024 * 1024 * 1024) {
                const int nsplit_t = std::max(16, 2 * syrk_plan_splits(ctx, nc, (rA + rB + 1) / 2));
                int64_t rps = ((rA + rB) + nsplit_t - 1) / nsplit_t;
                rps = ((rps + 31) / 32) * 32;
                const int nsA = rA > 0 ? (int)(((rA + rps - 1) / rps + 7) / 8) * 8 : 0;
                const int nsB = rB > 0 ? (int)(((rB + rps - 1) / rps + 7) / 8) * 8 : 0;
                double *dpanel2, *dpart2;
                OAK_CHECK(get_buf_t(ctx, "sobol_panel", (size_t)(rA + rB) * Mp, &dpanel2));
                OAK_CHECK(get_buf_t(ctx, "sobol_part", (size_t)(nsA + nsB) * Mp * Mp, &dpart2));
                {
                    PhaseTimer tp(ctx, "sobol_panel");
                    for (int m = 0; m < 2; ++m) {
                        const int64_t nr = hi2[m] - lo2[m];
                        if (nr <= 0) continue;
                        sobol_panel_kernel<<<(unsigned)((nr + SB_ROWS - 1) / SB_ROWS), 256, 0, ctx->stream>>>(
                            dLs, n, nslot, dalpha, m, npos, nneg, lo2[m], nr, dcols, nc, (int)Mp, dpanel2 + (m == 0 ? 0 : rA) * Mp);
                        OAK_HIP_CHECK(hipGetLastError());
                    }
                    tp.stop();
                }
                {
                    PhaseTimer ts(ctx, "sobol_syrk");
                    OAK_CHECK(syrk_panel_two(ctx, dpanel2, Mp, rA, rB, nc, dpart2, nsA, nsB, rps));
                    ts.stop();
                }
                if (nsA > 0) OAK_CHECK(syrk_reduce(ctx, dpart2, nsA, nc, dG, false));
                if (nsB > 0) OAK_CHECK(syrk_reduce(ctx, dpart2 + (size_t)nsA * Mp * Mp, nsB, nc, dG + (size_t)nc * nc, false));
                rows_mine_max = -1;                  // done: skip the chunked loop below
            }
        }
        const int64_t chunk_rows = rows_mine_max < chunk_cap ? (rows_mine_max > 0 ? rows_mine_max : 1) : chunk_cap;
        const int nsplit = syrk_plan_splits(ctx, nc, chunk_rows);
        double *dpanel, *dpart;
        OAK_CHECK(get_buf_t(ctx, "sobol_panel", (size_t)chunk_rows * Mp, &dpanel));
        OAK_CHECK(get_buf_t(ctx, "sobol_part", (size_t)nsplit * Mp * Mp, &dpart));
        for (int m = 0; m < 2 && rows_mine_max >= 0; ++m) {
            // this rank's contiguous share of the set's rows
            const int64_t per = (rows_set[m] + nranks - 1) / nranks;
            const int64_t lo = std::min<int64_t>(rows_set[m], per * rank), hi = std::min<int64_t>(rows_set[m], lo + per);
            if (hi <= lo) continue;
            bool first = true;
            for (int64_t r0 = lo; r0 < hi; r0 += chunk_rows) {
                const int64_t nr = std::min<int64_t>(chunk_rows, hi - r0);
                {
                    PhaseTimer tp(ctx, "sobol_panel");
                    sobol_panel_kernel<<<(unsigned)((nr + SB_ROWS - 1) / SB_ROWS), 256, 0, ctx->stream>>>(
                        dLs, n, nslot, dalpha, m, npos, nneg, r0, nr, dcols, nc, (int)Mp, dpanel);
                    OAK_HIP_CHECK(hipGetLastError());
                    tp.stop();
                }
                PhaseTimer ts(ctx, "sobol_syrk");
                OAK_CHECK(syrk_panel(ctx, dpanel, Mp, nr, nc, dpart, nsplit, !first));
                ts.stop();
                first = false;
            }
            OAK_CHECK(syrk_reduce(ctx, dpart, nsplit, nc, dG + (size_t)m * nc * nc, false));
        }
        if (nranks > 1) OAK_CHECK(comm_allreduce_dev(ctx, dG, 2 * (int64_t)nc * nc));
        double* ddirect = nullptr;
        const int ndirect = (int)plan.direct.size();
        if (ndirect > 0) {                           // alpha^T L_d alpha (replicated on every rank: D small terms)
            int* dd = nullptr;
            double* drows = nullptr;
            OAK_CHECK(get_buf_t(ctx, "sobol_direct_idx", (size_t)ndirect, &dd));
            OAK_CHECK(get_buf_t(ctx, "sobol_direct", (size_t)ndirect, &ddirect));
            OAK_CHECK(get_buf_t(ctx, "sobol_direct_rows", (size_t)ndirect * n, &drows));
            OAK_HIP_CHECK(hipMemcpyAsync(dd, plan.direct.data(), sizeof(int) * (size_t)ndirect, hipMemcpyHostToDevice, ctx->stream));
            sobol_quadform_rows_kernel<<<dim3((unsigned)((n + 3) / 4), (unsigned)ndirect), 256, 0, ctx->stream>>>(dLs, n, dd, dalpha_signed, drows);
            OAK_HIP_CHECK(hipGetLastError());
            sobol_quadform_sum_kernel<<<(unsigned)ndirect, 256, 0, ctx->stream>>>(drows, n, ddirect);
            OAK_HIP_CHECK(hipGetLastError());
        }
        sobol_gather_kernel<<<(unsigned)((n_subsets + 255) / 256), 256, 0, ctx->stream>>>(dG, nc, dc12, dalt, dmult, ddirect, n_subsets, dout, ddev);
        OAK_HIP_CHECK(hipGetLastError());
        OAK_HIP_CHECK(hipMemcpyAsync(out, dout, sizeof(double) * (size_t)n_subsets, hipMemcpyDeviceToHost, ctx->stream));
        unsigned long long bits = 0;
        OAK_HIP_CHECK(hipMemcpyAsync(&bits, ddev, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream));
        t_all.stop();
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        double dev;
        memcpy(&dev, &bits, sizeof(dev));
        ctx->sobol_info[1] = (double)nc;
        ctx->sobol_info[2] = dev;
        ctx->sobol_info[3] = (double)(rows_set[0] + rows_set[1]);
        return OAK_OK;
    }

    // ---- one workgroup per term; a collective call gives each rank a contiguous block of the terms -----------------------
    int *dsub, *doff, *dslot;
    OAK_CHECK(get_buf_t(ctx, "sobol_sub", (size_t)total + 1, &dsub));
    OAK_CHECK(get_buf_t(ctx, "sobol_off", (size_t)n_subsets + 1, &doff));
    OAK_CHECK(get_buf_t(ctx, "sobol_slot", (size_t)D, &dslot));
    OAK_HIP_CHECK(hipMemcpyAsync(dsub, subsets, sizeof(int) * (size_t)total, hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(doff, subset_off, sizeof(int) * (size_t)(n_subsets + 1), hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(dslot, slot.data(), sizeof(int) * (size_t)D, hipMemcpyHostToDevice, ctx->stream));
    const int per = (n_subsets + nranks - 1) / nranks;
    const int s_lo = std::min(n_subsets, per * rank), s_hi = std::min(n_subsets, s_lo + per);
    if (nranks > 1) OAK_CHECK(fill_zero(ctx, dout, sizeof(double) * (size_t)n_subsets));
    if (s_hi > s_lo) {
        sobol_terms_kernel<<<(unsigned)(s_hi - s_lo), 256, 0, ctx->stream>>>(dLs, n, dalpha, dsub, doff + s_lo, dslot, dmult + s_lo, dout + s_lo);
        OAK_HIP_CHECK(hipGetLastError());
    }
    if (nranks > 1) OAK_CHECK(comm_allreduce_dev(ctx, dout, n_subsets));
    OAK_HIP_CHECK(hipMemcpyAsync(out, dout, sizeof(double) * (size_t)n_subsets, hipMemcpyDeviceToHost, ctx->stream));
    t_all.stop();
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // also keeps the host vectors alive until the copies are done
    return OAK_OK;
}

int oak_sobol(oak_ctx* ctx, const oak_kernel_desc* desc, const double* Xc, int64_t n, int32_t ldx, const double* alpha,
              const int32_t* subsets, const int32_t* subset_off, int32_t n_subsets, int32_t use_order_var, double delta, double mu,
              double* out) {
    return sobol_run(ctx, desc, Xc, n, ldx, alpha, subsets, subset_off, n_subsets, use_order_var, delta, mu, out, false);
}

int oak_sobol_collective(oak_ctx* ctx, const oak_kernel_desc* desc, const double* Xc, int64_t n, int32_t ldx, const double* alpha,
                         const int32_t* subsets, const int32_t* subset_off, int32_t n_subsets, int32_t use_order_var, double delta,
                         double mu, double* out) {
    return sobol_run(ctx, desc, Xc, n, ldx, alpha, subsets, subset_off, n_subsets, use_order_var, delta, mu, out, true);
}

int oak_sobol_set_path(oak_ctx* ctx, int32_t path) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_REQUIRE(path >= 0 && path <= 2, "oak_sobol_set_path: path must be 0 (automatic), 1 (one workgroup per term) or 2 (Gram of products)");
    ctx->sobol_path = path;
    return OAK_OK;
}

int oak_sobol_last_info(oak_ctx* ctx, double* info4) {
    if (!ctx || !info4) { set_error("bad argument"); return OAK_E_ARG; }
    for (int i = 0; i < 4; ++i) info4[i] = ctx->sobol_info[i];
    return OAK_OK;
}

int oak_component_predict(oak_ctx* ctx, const oak_kernel_desc* desc, const double* Xs, int64_t ns, const double* Xc, int64_t n,
                          int32_t ldx, const double* alpha, const int32_t* subsets, const int32_t* subset_off, int32_t n_subsets,
                          int32_t use_order_var, double* out) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(desc && Xs && Xc && alpha && subsets && subset_off && out && ns >= 0 && n >= 1, "oak_component_predict: bad arguments");
    if (ns == 0 || n_subsets == 0) return OAK_OK;
    double *dXs, *dXc, *dalpha, *dK, *dy;
    OAK_CHECK(get_buf_t(ctx, "cp_Xs", (size_t)ns * ldx, &dXs));
    OAK_CHECK(get_buf_t(ctx, "cp_Xc", (size_t)n * ldx, &dXc));
    OAK_CHECK(get_buf_t(ctx, "cp_alpha", (size_t)n, &dalpha));
    OAK_CHECK(get_buf_t(ctx, "cp_K", (size_t)ns * n, &dK));
    OAK_CHECK(get_buf_t(ctx, "cp_y", (size_t)ns, &dy));
    OAK_HIP_CHECK(hipMemcpyAsync(dXs, Xs, sizeof(double) * (size_t)ns * ldx, hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(dXc, Xc, sizeof(double) * (size_t)n * ldx, hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(dalpha, alpha, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    for (int s = 0; s < n_subsets; ++s) {
        const int len = subset_off[s + 1] - subset_off[s];
        PreparedKernel pc;
        OAK_CHECK(prepare_component(ctx, desc, subsets + subset_off[s], len, use_order_var, &pc));
        Feat Fs, Fc;
        OAK_CHECK(featurize(ctx, pc, dXs, ns, ldx, "cp_Fs", &Fs));
        OAK_CHECK(featurize(ctx, pc, dXc, n, ldx, "cp_Fc", &Fc));
        OAK_CHECK(gram(ctx, pc, Fs, 0, ns, Fc, dK, n, nullptr, nullptr, 0));
        OAK_CHECK(gemv_rows(ctx, dK, ns, n, n, dalpha, dy));
        OAK_HIP_CHECK(hipMemcpyAsync(out + (int64_t)s * ns, dy, sizeof(double) * (size_t)ns, hipMemcpyDeviceToHost, ctx->stream));
    }
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OAK_OK;
}

}  // extern "C"

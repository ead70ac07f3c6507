// Sobol indices and per-term predictions on the device.
// Replaces compute_sobol_oak (oak/utils.py:338-435) and its per-dimension integrals compute_L (:221-240, closed forms
// f1..f4 :116-165), compute_L_binary_kernel (:243-272), compute_L_categorical_kernel (:275-309),
// compute_L_empirical_measure (:312-335); and get_prediction_component (:491-530).
// The reference rebuilds every L_d inside the Python loop over terms (np.repeat/np.tile, no reuse); here each L_d is
// generated once on the device and every term is a fused product-reduction  alpha^T (prod_d L_d) alpha.
#include "oak_internal.h"
#include <cmath>

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

__global__ void extract_col_kernel(const double* __restrict__ X, int64_t n, int ldx, int col, int trunc_flag, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = trunc_flag ? trunc(X[i * ldx + col]) : X[i * ldx + col];
}

// L_d^{base} for dim d of desc into dL [n x n]; returns in *vexp how the reference's variance argument v enters:
// 2 -> v^2 (RBF, categorical), 1 -> v (binary, utils.py:266).
static int sobol_L_dim(oak_ctx* ctx, const oak_kernel_desc* desc, const PreparedKernel& pk, int d, const double* dXc, int64_t n,
                       int32_t ldx, double delta, double mu, double* dL, int* vexp) {
    double* dx = nullptr;
    OAK_CHECK(get_buf_t(ctx, "sobol_x", (size_t)n, &dx));
    const int type = desc->dim_type[d];
    extract_col_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(dXc, n, ldx, desc->active_col[d], type != OAK_DIM_RBF, dx);
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

int oak_sobol(oak_ctx* ctx, const oak_kernel_desc* desc, const double* Xc, int64_t n, int32_t ldx, const double* alpha,
              const int32_t* subsets, const int32_t* subset_off, int32_t n_subsets, int32_t use_order_var, double delta, double mu,
              double* out) {
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
    double *dX, *dLs, *dalpha, *dmult, *dout;
    int *dsub, *doff, *dslot;
    OAK_CHECK(get_buf_t(ctx, "sobol_X", (size_t)n * ldx, &dX));
    OAK_CHECK(get_buf_t(ctx, "sobol_Ls", (size_t)nslot * n * n, &dLs));
    OAK_CHECK(get_buf_t(ctx, "sobol_alpha", (size_t)n, &dalpha));
    OAK_CHECK(get_buf_t(ctx, "sobol_mult", (size_t)n_subsets, &dmult));
    OAK_CHECK(get_buf_t(ctx, "sobol_out", (size_t)n_subsets, &dout));
    OAK_CHECK(get_buf_t(ctx, "sobol_sub", (size_t)total + 1, &dsub));
    OAK_CHECK(get_buf_t(ctx, "sobol_off", (size_t)n_subsets + 1, &doff));
    OAK_CHECK(get_buf_t(ctx, "sobol_slot", (size_t)D, &dslot));
    OAK_HIP_CHECK(hipMemcpyAsync(dX, Xc, sizeof(double) * (size_t)n * ldx, hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(dalpha, alpha, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    std::vector<int> vexp(D, 2);
    for (int d = 0; d < D; ++d)
        if (slot[d] >= 0) OAK_CHECK(sobol_L_dim(ctx, desc, pk, d, dX, n, ldx, delta, mu, dLs + (int64_t)slot[d] * n * n, &vexp[d]));
    // per-term scalar: the reference gives the first factor v = sigma2_{|S|} and the others v = 1 when variances are shared
    // (utils.py:376-380), else v = base variance of each factor (:382); v enters squared except for binary factors (:266)
    std::vector<double> mult(n_subsets, 1.0);
    for (int s = 0; s < n_subsets; ++s) {
        const int len = subset_off[s + 1] - subset_off[s];
        OAK_REQUIRE(len >= 1 && len <= OAK_MAX_DIMS, "subset %d has invalid length %d", s, len);
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
    OAK_HIP_CHECK(hipMemcpyAsync(dsub, subsets, sizeof(int) * (size_t)total, hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(doff, subset_off, sizeof(int) * (size_t)(n_subsets + 1), hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(dslot, slot.data(), sizeof(int) * (size_t)D, hipMemcpyHostToDevice, ctx->stream));
    sobol_terms_kernel<<<(unsigned)n_subsets, 256, 0, ctx->stream>>>(dLs, n, dalpha, dsub, doff, dslot, dmult, dout);
    OAK_HIP_CHECK(hipGetLastError());
    OAK_HIP_CHECK(hipMemcpyAsync(out, dout, sizeof(double) * (size_t)n_subsets, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // also keeps the host vectors alive until the copies are done
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

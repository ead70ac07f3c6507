// Whitened, diagonal-q SVGP with a Bernoulli likelihood: the model the reference's classification example builds on the OAK
// kernel (examples/uci/uci_classification_train.py:108-116: gpflow.models.SVGP(kernel, Bernoulli(invlink=inv_logit), Z,
// whiten=True, q_diag=True), full batch, BFGS over q_mu, q_sqrt and the kernel hyper-parameters) and whose posterior
// oak/utils.py:174-179 turns into (alpha, L).  GPflow 2.2.1 arithmetic (third-party, restated from its published
// definitions; see oracle/svgp_oracle.py for the op-by-op statement):
//   Lm = chol(K(Z) + jitter I);  A = Lm^-1 K(Z, X);  mu_n = sum_j A_jn q_mu_j;
//   var_n = Kdiag_n - sum_j A_jn^2 + sum_j (A_jn q_sqrt_j)^2
//   elbo = sum_n GH_n[ log Bernoulli(y_n | invlink(f)) ] - 1/2 sum_j (q_mu_j^2 + q_sqrt_j^2 - 1 - log q_sqrt_j^2)
// with GH_n the Gauss-Hermite rule f = mu_n + sqrt(2 var_n) x_i, weights w_i / sqrt(pi) (the caller passes x, w).
//
// Everything N-sized runs on the device: Gram rows, the row solves, one wave per row for the quadrature, and the
// reverse pass (adjoint of A through the triangular solve and the Cholesky factor, then the same pair-kernel
// contraction the SGPR backward uses).
#include "oak_internal.h"
#include <cmath>
#include <cstdlib>

namespace oak {

constexpr int SV_GH_MAX = 64;
struct SvQuad {
    int n, link;          // link 0: logistic, 1: probit (standard normal cdf)
    double eps;           // p = link(f) (1 - 2 eps) + eps
    double z[SV_GH_MAX];  // sqrt(2) x_i
    double w[SV_GH_MAX];  // w_i / sqrt(pi)
    double lw[SV_GH_MAX]; // log of the above
};

__device__ __forceinline__ double sv_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double sv_wave_max(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// log Bernoulli(y | p(f)) and its derivative in f
__device__ __forceinline__ double sv_logp(const SvQuad& q, double f, double y, double* dl) {
    double s, ds;
    if (q.link == 0) { s = 1.0 / (1.0 + exp(-f)); ds = s * (1.0 - s); }
    else { s = 0.5 * (1.0 + erf(f * 0.70710678118654752440)); ds = 0.39894228040143267794 * exp(-0.5 * f * f); }
    const double p = s * (1.0 - 2.0 * q.eps) + q.eps;
    const double dp = ds * (1.0 - 2.0 * q.eps);
    const bool one = (y == 1.0);
    const double pr = one ? p : 1.0 - p;
    *dl = (one ? dp : -dp) / pr;
    return log(pr);
}

// One wave per row n of AT (= column n of A = Lm^-1 Kuf): mean, variance, then the quadrature over the wave's lanes.
//   mode 0: ve[n] = variational expectation, gmu[n] = d ve / d mu, gv[n] = d ve / d var
//   mode 1: ve[n] = log predictive density  log sum_i w_i p(y_n | f_i)      (gmu, gv untouched)
//   mode 2: mean and variance only
__global__ void __launch_bounds__(256) svgp_rows_kernel(const double* __restrict__ AT, int64_t lda, int64_t N, int64_t M,
                                                        const double* __restrict__ qmu, const double* __restrict__ s2m1,
                                                        const double* __restrict__ kd, const double* __restrict__ y,
                                                        const SvQuad q, int mode, double* __restrict__ mu_out,
                                                        double* __restrict__ var_out, double* __restrict__ ve,
                                                        double* __restrict__ gmu, double* __restrict__ gv) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const double* a = AT + n * lda;
    double am = 0.0, av = 0.0;
    for (int64_t j = lane; j < M; j += 64) {
        const double x = a[j];
        am = __builtin_fma(x, qmu[j], am);
        av = __builtin_fma(x * x, s2m1[j], av);
    }
    const double mu = sv_wave_sum(am);
    const double var = kd[n] + sv_wave_sum(av);
    if (lane == 0) { mu_out[n] = mu; var_out[n] = var; }
    if (mode == 2) return;
    const double sd = sqrt(var);
    const bool on = lane < q.n;
    const int li = on ? lane : 0;
    double dl = 0.0;
    const double l = sv_logp(q, __builtin_fma(sd, q.z[li], mu), y[n], &dl);
    if (mode == 0) {
        const double wl = on ? q.w[li] : 0.0;
        const double e = sv_wave_sum(on ? wl * l : 0.0);
        const double g1 = sv_wave_sum(on ? wl * dl : 0.0);
        const double g2 = sv_wave_sum(on ? wl * dl * q.z[li] : 0.0);
        if (lane == 0) { ve[n] = e; gmu[n] = g1; gv[n] = g2 / (2.0 * sd); }
    } else {
        const double t = on ? l + q.lw[li] : -INFINITY;
        const double mx = sv_wave_max(t);
        const double sm = sv_wave_sum(on ? exp(t - mx) : 0.0);
        if (lane == 0) ve[n] = mx + log(sm);
    }
}

// Abar[n][j] = q_mu_j gmu_n + 2 (q_sqrt_j^2 - 1) A[n][j] gv_n     (adjoint of A, row layout)
// P[r(n)][j] = sqrt|gv_n| A[n][j]                                  (panel of the weighted SYRK, rows regrouped by the sign of gv_n)
// upart[by][j] = sum over this block's rows of A[n][j] gmu_n       (u = A gmu, finished by svgp_usum_kernel)
// W2 = A diag(gv) A^T is a SYRK with signed weights: W2 = P+^T P+ - P-^T P-, each part through the MFMA SYRK of the SGPR path on a
// panel scaled by sqrt|gv_n|.  (A plain M x M x N GEMM on a transposed copy ran at 16 TFLOP/s.)  gv_n < 0 for a log-concave
// likelihood; the jittered links leave some positive ones.  The rows are COMPACTED by sign -- the non-positive ones to the rows
// [0, n_neg) of P in their original order, the positive ones behind them -- so the two SYRKs together stream N rows, not 2 N
// (r03: they used to run over zero-filled full-height panels, 2 x 17.3 ms at N = 2^20).  Destinations come from per-block counts
// (svgp_sign_counts_kernel), scanned on the host, plus a rank inside the block that every column workgroup recomputes.
constexpr int SV_ROWS = 1024;     // rows per block: N / SV_ROWS partial rows of u
__global__ void __launch_bounds__(256) svgp_sign_counts_kernel(const double* __restrict__ gv, int64_t N, int* __restrict__ npos) {
    __shared__ int red[256];
    const int64_t n0 = (int64_t)blockIdx.x * SV_ROWS;
    int c = 0;
    for (int r = threadIdx.x; r < SV_ROWS; r += 256) c += (n0 + r < N && gv[n0 + r] > 0.0) ? 1 : 0;
    red[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) npos[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) svgp_adjoint_kernel(const double* __restrict__ AT, int64_t lda, int64_t N, int64_t M,
                                                           const double* __restrict__ qmu, const double* __restrict__ s2m1,
                                                           const double* __restrict__ gmu, const double* __restrict__ gv,
                                                           const int64_t* __restrict__ neg_off, const int64_t* __restrict__ pos_off,
                                                           int64_t n_neg, double* __restrict__ Abar, double* __restrict__ P,
                                                           double* __restrict__ upart) {
    __shared__ double red[8][33];
    __shared__ int rankp[SV_ROWS + 1];                           // exclusive count of positive rows before row r of this block
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 columns x 8 rows per step
    const int64_t jc = (int64_t)blockIdx.x * 32 + tx;            // < lda (grid covers the padded width)
    const bool live = jc < M;
    const double qm = live ? qmu[jc] : 0.0, sm = live ? s2m1[jc] : 0.0;
    const int64_t n_begin = (int64_t)blockIdx.y * SV_ROWS;
    const int64_t n_end = (n_begin + SV_ROWS < N) ? n_begin + SV_ROWS : N;
    {   // ranks: four consecutive rows per thread, then a scan of the 256 per-thread counts
        __shared__ int tsum[256];
        int f[4], c = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t n = n_begin + 4 * threadIdx.x + q;
            f[q] = (n < n_end && gv[n] > 0.0) ? 1 : 0;
            c += f[q];
        }
        tsum[threadIdx.x] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int t = 0; t < 256; ++t) { const int v = tsum[t]; tsum[t] = run; run += v; }
        }
        __syncthreads();
        int run = tsum[threadIdx.x];
#pragma unroll
        for (int q = 0; q < 4; ++q) { rankp[4 * threadIdx.x + q] = run; run += f[q]; }
        __syncthreads();
    }
    const int64_t nbase = neg_off[blockIdx.y], pbase = n_neg + pos_off[blockIdx.y];
    double cu = 0.0;
    for (int64_t n = n_begin + ty; n < n_end; n += 8) {
        const double g = gv[n], gm = gmu[n];
        const int r = (int)(n - n_begin), rp = rankp[r];
        const int64_t dest = g > 0.0 ? pbase + rp : nbase + (r - rp);
        double pv = 0.0;
        if (live) {
            const double x = AT[n * lda + jc];
            Abar[n * lda + jc] = __builtin_fma(qm, gm, 2.0 * sm * x * g);
            cu = __builtin_fma(x, gm, cu);
            pv = sqrt(fabs(g)) * x;
        }
        P[dest * lda + jc] = pv;
    }
    red[ty][tx] = cu;
    __syncthreads();
    if (ty == 0 && live) {
        double s = red[0][tx];
#pragma unroll
        for (int r = 1; r < 8; ++r) s += red[r][tx];
        upart[(int64_t)blockIdx.y * M + jc] = s;
    }
}
__global__ void svgp_usum_kernel(const double* __restrict__ upart, int64_t nby, int64_t M, double* __restrict__ u) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    double s = 0.0;
    for (int64_t b = 0; b < nby; ++b) s += upart[b * M + j];
    u[j] = s;
}
// W2 = Wpos - Wneg (Wpos may be NULL)
__global__ void svgp_w2_kernel(const double* __restrict__ Wpos, const double* __restrict__ Wneg, int64_t len, double* __restrict__ W2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) W2[i] = (Wpos ? Wpos[i] : 0.0) - Wneg[i];
}

// Q = q_mu u^T + 2 diag(q_sqrt^2 - 1) W2
__global__ void svgp_q_kernel(const double* __restrict__ W2, const double* __restrict__ u, const double* __restrict__ qmu,
                              const double* __restrict__ s2m1, int64_t M, double* __restrict__ Q) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < M) Q[i * M + j] = __builtin_fma(qmu[i], u[j], 2.0 * s2m1[i] * W2[i * M + j]);
}
// mode 0: out = -tril(in);  mode 1: out = tril(in) with the diagonal halved;  mode 2: out = (in + in^T) / 2
__global__ void svgp_tri_kernel(const double* __restrict__ in, int64_t M, int mode, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= M) return;
    const double x = in[i * M + j];
    double r;
    if (mode == 0) r = (j <= i) ? -x : 0.0;
    else if (mode == 1) r = (j < i) ? x : ((j == i) ? 0.5 * x : 0.0);
    else r = 0.5 * (x + in[j * M + i]);
    out[i * M + j] = r;
}
__global__ void svgp_diag_kernel(const double* __restrict__ W2, int64_t M, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < M) out[j] = W2[j * M + j];
}

static int sv_guard(oak_ctx* ctx) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    return OAK_OK;
}

static int sv_quad(const double* gh_x, const double* gh_w, int32_t n_gh, int32_t link, double eps, SvQuad* q) {
    OAK_REQUIRE(gh_x && gh_w && n_gh >= 1 && n_gh <= SV_GH_MAX, "SVGP: the Gauss-Hermite rule needs 1..%d nodes", SV_GH_MAX);
    OAK_REQUIRE(link == 0 || link == 1, "SVGP: link must be 0 (logistic) or 1 (probit)");
    OAK_REQUIRE(eps >= 0.0 && eps < 0.5, "SVGP: link jitter must lie in [0, 0.5)");
    q->n = n_gh; q->link = link; q->eps = eps;
    const double sqrt2 = std::sqrt(2.0), sqrtpi = std::sqrt(3.14159265358979323846);
    for (int i = 0; i < SV_GH_MAX; ++i) { q->z[i] = 0.0; q->w[i] = 0.0; q->lw[i] = 0.0; }
    for (int i = 0; i < n_gh; ++i) {
        OAK_REQUIRE(gh_w[i] > 0.0, "SVGP: Gauss-Hermite weights must be positive");
        q->z[i] = gh_x[i] * sqrt2;
        q->w[i] = gh_w[i] / sqrtpi;
        q->lw[i] = std::log(q->w[i]);
    }
    return OAK_OK;
}

// q_mu and q_sqrt^2 - 1 on the device; Lm = chol(K(Z) + jitter I) in "svL"
static int sv_prepare(oak_ctx* ctx, const PreparedKernel& pk, const double* q_mu, const double* q_sqrt, double jitter, bool with_grad,
                      Feat* FZ, double** dL, double** dqmu, double** ds2m1) {
    const int64_t M = ctx->M;
    std::vector<double> h((size_t)2 * M);
    for (int64_t j = 0; j < M; ++j) {
        OAK_REQUIRE(q_sqrt[j] > 0.0, "SVGP: q_sqrt must be positive");
        h[j] = q_mu[j];
        h[M + j] = q_sqrt[j] * q_sqrt[j] - 1.0;
    }
    double* dq = nullptr;
    OAK_CHECK(get_buf_t(ctx, "svq", (size_t)2 * M, &dq));
    OAK_HIP_CHECK(hipMemcpyAsync(dq, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));        // h leaves scope
    *dqmu = dq; *ds2m1 = dq + M;
    OAK_CHECK(featurize(ctx, pk, (double*)peek_buf(ctx, "Z"), M, ctx->ldx, with_grad ? "featZg" : "featZ", FZ, with_grad));
    OAK_CHECK(get_buf_t(ctx, "svL", (size_t)M * M, dL));
    OAK_CHECK(gram(ctx, pk, *FZ, 0, M, *FZ, *dL, M, nullptr, nullptr, 0));
    OAK_CHECK(add_diag(ctx, *dL, M, M, jitter));
    OAK_CHECK(potrf_lower(ctx, *dL, M, M));
    return OAK_OK;
}

}  // namespace oak

using namespace oak;

extern "C" {

int oak_svgp_elbo_grad(oak_ctx* ctx, const oak_kernel_desc* desc, const double* q_mu, const double* q_sqrt, double jitter,
                       const double* gh_x, const double* gh_w, int32_t n_gh, int32_t link, double link_eps, double* elbo_out,
                       double* grad_out, double* grad_qmu, double* grad_qsqrt) {
    OAK_CHECK(sv_guard(ctx));
    OAK_REQUIRE(desc && q_mu && q_sqrt && elbo_out, "oak_svgp_elbo_grad: bad arguments");
    OAK_REQUIRE(ctx->have_data && ctx->have_Z, "SVGP: oak_sgpr_set_data and oak_sgpr_set_inducing must be called first");
    const bool want_grad = grad_out != nullptr;
    OAK_REQUIRE(!want_grad || (grad_qmu && grad_qsqrt), "oak_svgp_elbo_grad: grad_qmu / grad_qsqrt are required with grad_out");
    SvQuad quad;
    OAK_CHECK(sv_quad(gh_x, gh_w, n_gh, link, link_eps, &quad));
    PreparedKernel pk;
    OAK_CHECK(prepare_kernel(ctx, desc, &pk, PK_GROUPED));
    const int64_t N = ctx->N, M = ctx->M, Mp = ((M + 127) / 128) * 128;
    OAK_REQUIRE((double)N * (double)Mp * 8.0 * 3.0 <= 160e9, "SVGP: N x M = %lld x %lld does not fit the unchunked path", (long long)N, (long long)M);
    PhaseTimer ttot(ctx, "total");
    Feat FZ, FX;
    double *dL, *dqmu, *ds2m1;
    OAK_CHECK(sv_prepare(ctx, pk, q_mu, q_sqrt, jitter, want_grad, &FZ, &dL, &dqmu, &ds2m1));
    OAK_CHECK(featurize(ctx, pk, (double*)peek_buf(ctx, "X"), N, ctx->ldx, want_grad ? "featXg" : "featX", &FX, want_grad));
    double *dAT, *dkd, *dmu, *dvar, *dve, *dgmu, *dgv, *dsc;
    OAK_CHECK(get_buf_t(ctx, "panel", (size_t)N * Mp, &dAT));
    OAK_CHECK(get_buf_t(ctx, "svrow", (size_t)6 * N, &dkd));
    dmu = dkd + N; dvar = dmu + N; dve = dvar + N; dgmu = dve + N; dgv = dgmu + N;
    OAK_CHECK(get_buf_t(ctx, "svsc", 8, &dsc));
    OAK_CHECK(gram(ctx, pk, FX, 0, N, FZ, dAT, Mp, nullptr, nullptr, Mp));      // rows = K(x_n, Z)
    OAK_CHECK(gram_diag(ctx, pk, FX, dkd, nullptr));
    OAK_CHECK(trsm_rows(ctx, dL, M, M, dAT, N, Mp, 0));                         // rows = columns of A = Lm^-1 Kuf
    svgp_rows_kernel<<<(unsigned)((N + 3) / 4), 256, 0, ctx->stream>>>(dAT, Mp, N, M, dqmu, ds2m1, dkd, (double*)peek_buf(ctx, "Y"),
                                                                      quad, 0, dmu, dvar, dve, dgmu, dgv);
    OAK_HIP_CHECK(hipGetLastError());
    // Row shards (one rank per GPU): the sum of the expectations, the sign flag, u, the SYRK results and the gradient
    // record are sums over rows and are all-reduced; K(Z), its factor and everything M-sized is replicated.
    OAK_CHECK(reduce_sum(ctx, dve, N, dsc, 0, 1));
    // per 1024-row block: how many rows have gv_n > 0 (the reverse pass regroups the rows by that sign)
    const int64_t nby = (N + SV_ROWS - 1) / SV_ROWS;
    int* dnpos = nullptr;
    std::vector<int> hnpos;
    if (want_grad) {
        OAK_CHECK(get_buf_t(ctx, "svnpos", (size_t)nby, &dnpos));
        svgp_sign_counts_kernel<<<(unsigned)nby, 256, 0, ctx->stream>>>(dgv, N, dnpos);
        OAK_HIP_CHECK(hipGetLastError());
        hnpos.resize((size_t)nby);
        OAK_HIP_CHECK(hipMemcpyAsync(hnpos.data(), dnpos, sizeof(int) * (size_t)nby, hipMemcpyDeviceToHost, ctx->stream));
    }
    OAK_CHECK(comm_allreduce_dev(ctx, dsc, 1));
    double hsc[2] = {0.0, 0.0};
    OAK_HIP_CHECK(hipMemcpyAsync(hsc, dsc, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    const double& sum_ve = hsc[0];
    double kl = 0.0;
    for (int64_t j = 0; j < M; ++j) {
        const double s2 = q_sqrt[j] * q_sqrt[j];
        kl += q_mu[j] * q_mu[j] + s2 - 1.0 - std::log(s2);
    }
    kl *= 0.5;
    if (!want_grad) {
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        ttot.stop();
        *elbo_out = sum_ve - kl;
        return OAK_OK;
    }
    // ---- reverse pass ----------------------------------------------------------------------------------------
    double *dAbar, *dP, *du, *dW2, *dQ, *dT, *dLinvT, *dLT, *dPm, *dGuu, *dw2d;
    OAK_CHECK(get_buf_t(ctx, "gpanel", (size_t)N * Mp, &dAbar));
    OAK_CHECK(get_buf_t(ctx, "svP", (size_t)N * Mp, &dP));
    OAK_CHECK(get_buf_t(ctx, "svu", (size_t)2 * M, &du));
    dw2d = du + M;
    OAK_CHECK(get_buf_t(ctx, "svW2", (size_t)M * M, &dW2));
    OAK_CHECK(get_buf_t(ctx, "svQ", (size_t)M * M, &dQ));
    OAK_CHECK(get_buf_t(ctx, "svT", (size_t)M * M, &dT));
    OAK_CHECK(get_buf_t(ctx, "svLinvT", (size_t)M * M, &dLinvT));
    OAK_CHECK(get_buf_t(ctx, "svLT", (size_t)M * M, &dLT));
    OAK_CHECK(get_buf_t(ctx, "svPm", (size_t)M * M, &dPm));
    OAK_CHECK(get_buf_t(ctx, "svGuu", (size_t)M * M, &dGuu));
    {
        double *dup = nullptr, *dPart = nullptr, *dWn = nullptr, *dWp = nullptr;
        OAK_CHECK(get_buf_t(ctx, "svupart", (size_t)nby * M, &dup));
        // destinations of the regrouped rows: exclusive scans of the per-block counts (host: nby values)
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));                        // the counts (and sum_ve) have landed
        std::vector<int64_t> hoff((size_t)2 * nby);
        int64_t n_pos = 0, n_negs = 0;
        for (int64_t b = 0; b < nby; ++b) {
            const int64_t rows_b = ((b + 1) * SV_ROWS <= N) ? SV_ROWS : N - b * SV_ROWS;
            hoff[(size_t)b] = n_negs; hoff[(size_t)(nby + b)] = n_pos;
            n_pos += hnpos[(size_t)b]; n_negs += rows_b - hnpos[(size_t)b];
        }
        int64_t* doff = nullptr;
        OAK_CHECK(get_buf_t(ctx, "svoff", (size_t)2 * nby, &doff));
        OAK_HIP_CHECK(hipMemcpyAsync(doff, hoff.data(), sizeof(int64_t) * hoff.size(), hipMemcpyHostToDevice, ctx->stream));
        dim3 grid((unsigned)(Mp / 32), (unsigned)nby);
        svgp_adjoint_kernel<<<grid, 256, 0, ctx->stream>>>(dAT, Mp, N, M, dqmu, ds2m1, dgmu, dgv, doff, doff + nby, n_negs, dAbar, dP, dup);
        OAK_HIP_CHECK(hipGetLastError());
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));                        // hoff leaves scope below; cheap next to the SYRKs
        svgp_usum_kernel<<<(unsigned)((M + 255) / 256), 256, 0, ctx->stream>>>(dup, nby, M, du);   // u = A gmu
        OAK_HIP_CHECK(hipGetLastError());
        OAK_CHECK(get_buf_t(ctx, "svWn", (size_t)M * M, &dWn));
        auto weighted_syrk = [&](const double* panel, int64_t rows, double* dW) -> int {
            if (rows <= 0) return fill_zero(ctx, dW, sizeof(double) * (size_t)M * M);
            const int nsplit = syrk_plan_splits(ctx, M, rows);
            OAK_CHECK(get_buf_t(ctx, "syrk_part", (size_t)nsplit * Mp * Mp, &dPart));
            OAK_CHECK(syrk_panel(ctx, panel, Mp, rows, M, dPart, nsplit, false));
            return syrk_reduce(ctx, dPart, nsplit, M, dW, false);
        };
        OAK_CHECK(weighted_syrk(dP, n_negs, dWn));
        OAK_CHECK(comm_allreduce_dev(ctx, du, M));
        OAK_CHECK(comm_allreduce_dev(ctx, dWn, M * M));
        // whether ANY rank has positive rows decides, for all ranks alike, whether the second product (and its collective) runs
        double any_pos = n_pos > 0 ? 1.0 : 0.0;
        if (ctx->comm != nullptr && ctx->nranks > 1) {
            double* dflag = nullptr;
            OAK_CHECK(get_buf_t(ctx, "svflag", 1, &dflag));
            OAK_HIP_CHECK(hipMemcpyAsync(dflag, &any_pos, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            OAK_CHECK(comm_allreduce_dev(ctx, dflag, 1));
            OAK_HIP_CHECK(hipMemcpyAsync(&any_pos, dflag, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
        if (any_pos > 0.0) {
            OAK_CHECK(get_buf_t(ctx, "svWp", (size_t)M * M, &dWp));
            OAK_CHECK(weighted_syrk(dP + n_negs * Mp, n_pos, dWp));
            OAK_CHECK(comm_allreduce_dev(ctx, dWp, M * M));
        }
        svgp_w2_kernel<<<(unsigned)((M * M + 255) / 256), 256, 0, ctx->stream>>>(dWp, dWn, M * M, dW2);   // W2 = A diag(gv) A^T
        OAK_HIP_CHECK(hipGetLastError());
    }
    const dim3 gm((unsigned)((M + 255) / 256), (unsigned)M);
    svgp_diag_kernel<<<(unsigned)((M + 255) / 256), 256, 0, ctx->stream>>>(dW2, M, dw2d);
    OAK_HIP_CHECK(hipGetLastError());
    svgp_q_kernel<<<gm, 256, 0, ctx->stream>>>(dW2, du, dqmu, ds2m1, M, dQ);    // Q = Abar A^T
    OAK_HIP_CHECK(hipGetLastError());
    OAK_CHECK(set_identity(ctx, dLinvT, M));
    OAK_CHECK(trsm_rows(ctx, dL, M, M, dLinvT, M, M, 0));                        // rows = columns of Lm^-1: the matrix Lm^-T
    OAK_CHECK(gemm_nn(ctx, dLinvT, dQ, dT, M, M, M, M, M, M, 1.0, 0.0));         // Lm^-T Q
    svgp_tri_kernel<<<gm, 256, 0, ctx->stream>>>(dT, M, 0, dPm);                  // Lbar = -tril(Lm^-T Q)
    OAK_HIP_CHECK(hipGetLastError());
    OAK_CHECK(transpose(ctx, dL, M, M, M, dLT, M));
    OAK_CHECK(gemm_nn(ctx, dLT, dPm, dT, M, M, M, M, M, M, 1.0, 0.0));            // Lm^T Lbar
    svgp_tri_kernel<<<gm, 256, 0, ctx->stream>>>(dT, M, 1, dPm);                  // Phi(.)
    OAK_HIP_CHECK(hipGetLastError());
    OAK_CHECK(gemm_nn(ctx, dLinvT, dPm, dT, M, M, M, M, M, M, 1.0, 0.0));         // Lm^-T Phi
    OAK_CHECK(gemm_nt(ctx, dT, dLinvT, dPm, M, M, M, M, M, M, 1.0, 0.0, 0));      // ... Lm^-1
    svgp_tri_kernel<<<gm, 256, 0, ctx->stream>>>(dPm, M, 2, dGuu);                // adjoint of Kuu, symmetrised
    OAK_HIP_CHECK(hipGetLastError());
    OAK_CHECK(trsm_rows(ctx, dL, M, M, dAbar, N, Mp, 1));                        // rows = columns of Lm^-T Abar: adjoint of Kuf
    const int64_t reclen = record_len(pk);
    double* d_rec = nullptr;
    OAK_CHECK(get_buf_t(ctx, "g_rec", (size_t)reclen, &d_rec));
    OAK_CHECK(fill_zero(ctx, d_rec, sizeof(double) * (size_t)reclen));
    const bool want_gk = desc->grad_base_var != 0;
    OAK_CHECK(gram_bwd(ctx, pk, FX, 0, N, FZ, dAbar, Mp, 1.0, nullptr, nullptr, d_rec, want_gk));
    // <adjoint of Kuu, dKuu> is replicated on every rank: each contributes 1/nranks, the all-reduce restores it once
    OAK_CHECK(gram_bwd(ctx, pk, FZ, 0, M, FZ, dGuu, M, 1.0 / (double)(ctx->comm ? ctx->nranks : 1), nullptr, nullptr, d_rec, want_gk));
    OAK_CHECK(diag_bwd(ctx, pk, FX, 1.0, d_rec, dgv));
    OAK_CHECK(comm_allreduce_dev(ctx, d_rec, reclen));
    std::vector<double> rec((size_t)reclen), hu((size_t)2 * M);
    OAK_HIP_CHECK(hipMemcpyAsync(rec.data(), d_rec, sizeof(double) * (size_t)reclen, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(hu.data(), du, sizeof(double) * (size_t)2 * M, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ttot.stop();
    scatter_record(desc, pk, rec, 0.0, grad_out);
    for (int64_t j = 0; j < M; ++j) {
        grad_qmu[j] = hu[j] - q_mu[j];
        grad_qsqrt[j] = 2.0 * q_sqrt[j] * hu[M + j] - q_sqrt[j] + 1.0 / q_sqrt[j];
    }
    *elbo_out = sum_ve - kl;
    return OAK_OK;
}

int oak_svgp_predict(oak_ctx* ctx, const oak_kernel_desc* desc, const double* q_mu, const double* q_sqrt, double jitter,
                     const double* Xs, int64_t Ns, int32_t ldx, double* mean, double* var, const double* Ys, double* logdens,
                     const double* gh_x, const double* gh_w, int32_t n_gh, int32_t link, double link_eps) {
    OAK_CHECK(sv_guard(ctx));
    OAK_REQUIRE(desc && q_mu && q_sqrt && Xs && mean && var && Ns >= 0, "oak_svgp_predict: bad arguments");
    OAK_REQUIRE(ctx->have_Z && ldx == ctx->ldx, "SVGP: oak_sgpr_set_inducing must be called first (same column count)");
    OAK_REQUIRE((Ys == nullptr) == (logdens == nullptr), "oak_svgp_predict: Ys and logdens go together");
    SvQuad quad;
    quad.n = 0; quad.link = 0; quad.eps = 0.0;
    if (Ys != nullptr) OAK_CHECK(sv_quad(gh_x, gh_w, n_gh, link, link_eps, &quad));
    if (Ns == 0) return OAK_OK;
    PreparedKernel pk;
    OAK_CHECK(prepare_kernel(ctx, desc, &pk, PK_GROUPED));
    const int64_t M = ctx->M;
    Feat FZ, FS;
    double *dL, *dqmu, *ds2m1;
    OAK_CHECK(sv_prepare(ctx, pk, q_mu, q_sqrt, jitter, false, &FZ, &dL, &dqmu, &ds2m1));
    int64_t chunk = (int64_t)(((size_t)1 << 29) / (size_t)M);
    if (chunk > Ns) chunk = Ns;
    if (chunk < 8) chunk = 8;
    double *dXs, *dK, *drow;
    OAK_CHECK(get_buf_t(ctx, "pXs", (size_t)chunk * ldx, &dXs));
    OAK_CHECK(get_buf_t(ctx, "pK", (size_t)chunk * M, &dK));
    OAK_CHECK(get_buf_t(ctx, "svprow", (size_t)5 * chunk, &drow));
    double *dkd = drow, *dmu = drow + chunk, *dvar = dmu + chunk, *dy = dvar + chunk, *dld = dy + chunk;
    PhaseTimer tp(ctx, "predict");
    for (int64_t a0 = 0; a0 < Ns; a0 += chunk) {
        const int64_t na = (a0 + chunk <= Ns) ? chunk : Ns - a0;
        OAK_HIP_CHECK(hipMemcpyAsync(dXs, Xs + a0 * ldx, sizeof(double) * (size_t)na * ldx, hipMemcpyHostToDevice, ctx->stream));
        if (Ys) OAK_HIP_CHECK(hipMemcpyAsync(dy, Ys + a0, sizeof(double) * (size_t)na, hipMemcpyHostToDevice, ctx->stream));
        OAK_CHECK(featurize(ctx, pk, dXs, na, ldx, "featS", &FS));
        OAK_CHECK(gram(ctx, pk, FS, 0, na, FZ, dK, M, nullptr, nullptr, 0));
        OAK_CHECK(gram_diag(ctx, pk, FS, dkd, nullptr));
        OAK_CHECK(trsm_rows(ctx, dL, M, M, dK, na, M, 0));
        svgp_rows_kernel<<<(unsigned)((na + 3) / 4), 256, 0, ctx->stream>>>(dK, M, na, M, dqmu, ds2m1, dkd, dy, quad, Ys ? 1 : 2, dmu, dvar,
                                                                           dld, nullptr, nullptr);
        OAK_HIP_CHECK(hipGetLastError());
        OAK_HIP_CHECK(hipMemcpyAsync(mean + a0, dmu, sizeof(double) * (size_t)na, hipMemcpyDeviceToHost, ctx->stream));
        OAK_HIP_CHECK(hipMemcpyAsync(var + a0, dvar, sizeof(double) * (size_t)na, hipMemcpyDeviceToHost, ctx->stream));
        if (Ys) OAK_HIP_CHECK(hipMemcpyAsync(logdens + a0, dld, sizeof(double) * (size_t)na, hipMemcpyDeviceToHost, ctx->stream));
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    tp.stop();
    return OAK_OK;
}

// posterior.alpha = Lm^-T q_mu and chol(inv(posterior.Qinv)), Qinv = Lm^-T (I - diag(q_sqrt^2)) Lm^-1 (oak/utils.py:174-179):
// inv(Qinv) = Lm (I - S)^-1 Lm^T, whose lower Cholesky factor is Lm diag(1 / sqrt(1 - q_sqrt^2)).
int oak_svgp_posterior(oak_ctx* ctx, const oak_kernel_desc* desc, const double* q_mu, const double* q_sqrt, double jitter,
                       double* alpha_out, double* L_out) {
    OAK_CHECK(sv_guard(ctx));
    OAK_REQUIRE(desc && q_mu && q_sqrt && alpha_out, "oak_svgp_posterior: bad arguments");
    OAK_REQUIRE(ctx->have_Z, "SVGP: oak_sgpr_set_inducing must be called first");
    PreparedKernel pk;
    OAK_CHECK(prepare_kernel(ctx, desc, &pk, PK_GROUPED));
    const int64_t M = ctx->M;
    if (L_out != nullptr) {
        for (int64_t j = 0; j < M; ++j)
            if (!(q_sqrt[j] * q_sqrt[j] < 1.0)) {
                set_error("SVGP: inv(Qinv) is not positive definite (q_sqrt[%lld] = %g >= 1)", (long long)j, q_sqrt[j]);
                return OAK_E_NOTPD;
            }
    }
    Feat FZ;
    double *dL, *dqmu, *ds2m1, *da;
    OAK_CHECK(sv_prepare(ctx, pk, q_mu, q_sqrt, jitter, false, &FZ, &dL, &dqmu, &ds2m1));
    OAK_CHECK(get_buf_t(ctx, "sva", (size_t)M, &da));
    OAK_CHECK(copy_d2d(ctx, da, dqmu, sizeof(double) * (size_t)M));
    OAK_CHECK(trsm_rows(ctx, dL, M, M, da, 1, M, 1));
    OAK_HIP_CHECK(hipMemcpyAsync(alpha_out, da, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
    if (L_out != nullptr) OAK_HIP_CHECK(hipMemcpyAsync(L_out, dL, sizeof(double) * (size_t)M * M, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (L_out != nullptr)
        for (int64_t i = 0; i < M; ++i)
            for (int64_t j = 0; j <= i; ++j) L_out[i * M + j] /= std::sqrt(1.0 - q_sqrt[j] * q_sqrt[j]);
    return OAK_OK;
}

}  // extern "C"

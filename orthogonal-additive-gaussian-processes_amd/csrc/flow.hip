// KL objective of the per-feature normalising flow and its gradient, one pass over the sample.
//
// Replaces the TensorFlow evaluation + autodiff of Normalizer.KL_objective (oak/normalising_flow.py:79-85) that
// oak_model.fit minimises with L-BFGS-B for every continuous feature (oak/model_utils.py:305-317):
//     y = sinh((asinh(z) + skewness) * tailweight),  z = scale * (g + shift),  g = log(x - offset) or x
//     KL = 1/2 mean(y^2) - mean(log |dy/dx|),
//     log |dy/dx| = log cosh(u) + log tailweight - 1/2 log(1 + z^2) + log scale - [g if log],   u = (asinh z + skewness) tailweight
// (TFP chain SinhArcsinh o Scale o Shift o Log o Shift(-offset), :16-55).  The gradient is with respect to the four
// CONSTRAINED parameters (scale, shift, skewness, tailweight); the host chains through the Exp transforms.
// O(N) per evaluation: the NumPy mirror with finite-difference gradients took 8.5 s per feature at N = 2^20 (80 passes);
// here one evaluation is one reduction kernel over HBM-resident data.  Deterministic (fixed two-level reduction tree).
#include "oak_internal.h"

namespace oak {

constexpr int FLOW_NWG = 512;

__global__ void __launch_bounds__(256)
flow_objective_kernel(const double* __restrict__ g, int64_t n, int use_log, double s, double b, double k, double t,
                      double* __restrict__ part /* [gridDim.x][5] */) {
    __shared__ double red[5][256];
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};     // f, df/ds, df/db, df/dk, df/dt
    const double log_t = log(t), log_s = log(s);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double gi = g[i];
        const double z = (gi + b) * s;
        const double r2 = __builtin_fma(z, z, 1.0);
        const double r = sqrt(r2);
        const double a = asinh(z);
        const double u = (a + k) * t;
        const double S = sinh(u), C = cosh(u), T = tanh(u);
        const double au = fabs(u);
        const double log_cosh = au + log1p(exp(-2.0 * au)) - 0.6931471805599453094;
        double ld = log_cosh + log_t - 0.5 * log(r2) + log_s;
        if (use_log) ld -= gi;
        const double fu = S * C - T;                 // d/du of 1/2 sinh^2 u - log cosh u
        const double fz = fu * t / r + z / r2;
        acc[0] += 0.5 * S * S - ld;
        acc[1] += fz * (gi + b);
        acc[2] += fz * s;
        acc[3] += fu * t;
        acc[4] += fu * (a + k);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) red[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int q = 0; q < 5; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x < 5) part[blockIdx.x * 5 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void flow_finish_kernel(const double* __restrict__ part, int nwg, double inv_n, double s, double t, double* __restrict__ out5) {
    const int q = threadIdx.x;
    if (q >= 5) return;
    double a = 0.0;
    for (int w = 0; w < nwg; ++w) a += part[w * 5 + q];
    a *= inv_n;
    if (q == 1) a -= 1.0 / s;                        // - d/ds log scale
    if (q == 4) a -= 1.0 / t;                        // - d/dt log tailweight
    out5[q] = a;
}

}  // namespace oak

using namespace oak;

extern "C" int oak_flow_objective(oak_ctx* ctx, const double* g_host, int64_t n, int32_t use_log, double scale, double shift,
                                  double skewness, double tailweight, double* objective_out, double* grad_out) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(n >= 1 && objective_out && scale > 0.0 && tailweight > 0.0, "oak_flow_objective: bad arguments");
    double *dG, *dPart, *dOut;
    if (g_host != nullptr) {                         // (re)load the sample; NULL re-uses the resident one
        OAK_CHECK(get_buf_t(ctx, "flow_g", (size_t)n, &dG));
        OAK_HIP_CHECK(hipMemcpyAsync(dG, g_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        ctx->flow_n = n;
    } else {
        OAK_REQUIRE(ctx->flow_n == n, "oak_flow_objective: no resident sample of %lld values (pass g once first)", (long long)n);
        dG = (double*)peek_buf(ctx, "flow_g");
    }
    OAK_CHECK(get_buf_t(ctx, "flow_part", (size_t)FLOW_NWG * 5, &dPart));
    OAK_CHECK(get_buf_t(ctx, "flow_out", 5, &dOut));
    int nwg = (int)((n + 255) / 256);
    if (nwg > FLOW_NWG) nwg = FLOW_NWG;
    flow_objective_kernel<<<nwg, 256, 0, ctx->stream>>>(dG, n, use_log ? 1 : 0, scale, shift, skewness, tailweight, dPart);
    flow_finish_kernel<<<1, 64, 0, ctx->stream>>>(dPart, nwg, 1.0 / (double)n, scale, tailweight, dOut);
    OAK_HIP_CHECK(hipGetLastError());
    double h[5];
    OAK_HIP_CHECK(hipMemcpyAsync(h, dOut, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *objective_out = h[0];
    if (grad_out) for (int q = 0; q < 4; ++q) grad_out[q] = h[1 + q];
    return OAK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Column-wise forward transform of a whole design matrix: what oak_model._transform_x / apply_normalise_flow do with
// one NumPy pass per column (oak/model_utils.py:179-191, :462-476).  kind[d]: 0 = copy, 1 = flow without log, 2 = flow
// with log(x - offset), 3 = affine (x - mean) / std (the StandardScaler columns).  params[d] = {offset, scale, shift,
// skewness, tailweight} for flows, {mean, std, -, -, -} for affine columns.  Elementwise, HBM/PCIe bound.
// ---------------------------------------------------------------------------------------------------------------------
namespace oak {

__global__ void __launch_bounds__(256)
flow_forward_kernel(const double* __restrict__ X, int64_t n, int ld, int D, const int* __restrict__ kind, const double* __restrict__ params,
                    double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;     // element of the N x ld array (row-major)
    if (e >= n * ld) return;
    const int d = (int)(e % ld);
    double x = X[e];
    if (d < D) {
        const int kd = kind[d];
        const double* p = params + 5 * d;
        if (kd == 1 || kd == 2) {
            const double g = kd == 2 ? log(x - p[0]) : x;
            const double z = (g + p[2]) * p[1];
            x = sinh((asinh(z) + p[3]) * p[4]);
        } else if (kd == 3) {
            x = (x - p[0]) / p[1];
        }
    }
    out[e] = x;
}

}  // namespace oak

extern "C" int oak_flow_forward(oak_ctx* ctx, const double* X, int64_t N, int32_t ldx, int32_t D, const int32_t* kind,
                                const double* params, double* out) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(X && out && kind && params && N >= 0 && D >= 1 && D <= 4096 && ldx >= D, "oak_flow_forward: bad arguments");
    for (int d = 0; d < D; ++d) OAK_REQUIRE(kind[d] >= 0 && kind[d] <= 3, "oak_flow_forward: kind[%d] = %d", d, kind[d]);
    if (N == 0) return OAK_OK;
    PhaseTimer t(ctx, "flow_forward");
    int* dKind;
    double *dPar, *dX, *dOut;
    const int64_t chunk_rows = (((int64_t)1 << 28) / ldx) > 0 ? (((int64_t)1 << 28) / ldx) : 1;      // <= 2 GiB per pass
    const int64_t cr = chunk_rows < N ? chunk_rows : N;
    OAK_CHECK(get_buf_t(ctx, "ff_kind", (size_t)D, &dKind));
    OAK_CHECK(get_buf_t(ctx, "ff_par", (size_t)5 * D, &dPar));
    OAK_CHECK(get_buf_t(ctx, "ff_x", (size_t)cr * ldx, &dX));
    OAK_CHECK(get_buf_t(ctx, "ff_out", (size_t)cr * ldx, &dOut));
    OAK_HIP_CHECK(hipMemcpyAsync(dKind, kind, sizeof(int) * (size_t)D, hipMemcpyHostToDevice, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(dPar, params, sizeof(double) * 5 * (size_t)D, hipMemcpyHostToDevice, ctx->stream));
    for (int64_t r0 = 0; r0 < N; r0 += cr) {
        const int64_t nr = (r0 + cr <= N) ? cr : N - r0;
        OAK_HIP_CHECK(hipMemcpyAsync(dX, X + r0 * ldx, sizeof(double) * (size_t)nr * ldx, hipMemcpyHostToDevice, ctx->stream));
        flow_forward_kernel<<<(unsigned)((nr * ldx + 255) / 256), 256, 0, ctx->stream>>>(dX, nr, ldx, D, dKind, dPar, dOut);
        OAK_HIP_CHECK(hipGetLastError());
        OAK_HIP_CHECK(hipMemcpyAsync(out + r0 * ldx, dOut, sizeof(double) * (size_t)nr * ldx, hipMemcpyDeviceToHost, ctx->stream));
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    t.stop();
    return OAK_OK;
}

// Runtime: context, scratch buffers, error reporting, kernel-description preparation, featurize.
#include "oak_internal.h"
#include <mutex>
#include <set>
#include <cstdarg>
#include <cmath>
#include <ctime>
#include <string>

namespace oak {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int get_buf(oak_ctx* ctx, const char* name, size_t bytes, void** out) {
    DevBuf& b = ctx->bufs[name];
    if (bytes == 0) bytes = 8;
    if (b.bytes < bytes) {
        if (b.p) {
            OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            OAK_HIP_CHECK(hipFree(b.p));
            b.p = nullptr; b.bytes = 0;
        }
        size_t want = (bytes + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) {
            b.p = nullptr;
            set_error("hipMalloc(%zu bytes) for '%s' failed: %s", want, name, hipGetErrorString(e));
            return OAK_E_HIP;
        }
        b.bytes = want;
    }
    *out = b.p;
    return OAK_OK;
}

void* peek_buf(oak_ctx* ctx, const char* name) {
    auto it = ctx->bufs.find(name);
    return it == ctx->bufs.end() ? nullptr : it->second.p;
}

static std::mutex g_ctx_mu;
static std::set<oak_ctx*> g_ctxs;                 // live contexts (oak_debug_state walks them)
static double wall_s() {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void debug_mark(oak_ctx* ctx, const char* literal) {
    const unsigned i = ctx->mark_n & 15u;
    ctx->marks[i] = literal; ctx->mark_t[i] = wall_s();
    ctx->mark_n = ctx->mark_n + 1;
}

PhaseTimer::PhaseTimer(oak_ctx* c, const char* n) : ctx(c), name(n), a(nullptr), b(nullptr), active(false) {
    debug_mark(c, n);
    if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) {
        (void)hipEventRecord(a, ctx->stream);
        active = true;
    }
}
// Finished phases are folded into ctx->timings.  `block` waits for every pending phase (oak_last_timing / reset / destroy);
// otherwise only phases whose end event has already completed are harvested, so a caller that never reads the timings does
// not accumulate events.
static void flush_timings(oak_ctx* ctx, bool block = true) {
    std::vector<PendingEvt> keep;
    for (auto& p : ctx->pending) {
        if (!block && hipEventQuery(p.b) != hipSuccess) { keep.push_back(p); continue; }
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            Timing& t = ctx->timings[p.name];
            t.ms += ms; t.count += 1;
        }
        (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
    }
    ctx->pending.swap(keep);
}
PhaseTimer::~PhaseTimer() {
    if (active) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); active = false; }
    else if (a != nullptr && b == nullptr) (void)hipEventDestroy(a);
}
void PhaseTimer::stop() {
    if (!active) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->pending.push_back({name, a, b});
    active = false;
    if (ctx->pending.size() > 64) flush_timings(ctx, false);
}
void reset_timings(oak_ctx* ctx) { flush_timings(ctx); ctx->timings.clear(); }

int copy_d2d(oak_ctx* ctx, void* dst, const void* src, size_t bytes) {
    OAK_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return OAK_OK;
}
int fill_zero(oak_ctx* ctx, void* dst, size_t bytes) {
    OAK_HIP_CHECK(hipMemsetAsync(dst, 0, bytes, ctx->stream));
    return OAK_OK;
}

// ---------------------------------------------------------------------------------------------
// featurize kernels
// ---------------------------------------------------------------------------------------------
// one point, one dimension: vx = scaled input (category index for discrete dims), vc = cov_X_s / sqrt(var_s),
// vd = its lengthscale derivative in the pair kernels' units
__device__ __forceinline__ void featurize_point(const DevDesc& dd, const DevMeasure& dm, const double* __restrict__ meas, int d,
                                                double x, double& vx, double& vc, double& vd) {
    vx = 0.0; vc = 0.0; vd = 0.0;
    if (dd.type[d] == OAK_DIM_RBF) {
        vx = x * dd.scale[d];
        const double l = dm.ls[d], bv = dd.bv[d];
        double c = 0.0, dc = 0.0;     // cov_X_s(x) and its lengthscale derivative
        switch (dm.kind[d]) {
            case OAK_MEAS_GAUSSIAN: {   // oak/ortho_rbf_kernel.py:82-92
                const double mu = dm.p0[d], var = dm.p1[d];
                const double s = l * l + var;
                const double u2 = (x - mu) * (x - mu);
                c = bv * l / sqrt(s) * exp(-0.5 * u2 / s);
                dc = c * (1.0 / l - l / s + u2 * l / (s * s));
            } break;
            case OAK_MEAS_UNIFORM: {    // oak/ortho_rbf_kernel.py:49-63
                const double a = dm.p0[d], b = dm.p1[d];
                const double r2l = 1.0 / (1.4142135623730951 * l);
                const double zb = (b - x) * r2l, za = (a - x) * r2l;
                const double pre = bv * l / (b - a) * 1.2533141373155001;
                c = pre * (erf(zb) - erf(za));
                dc = c / l + pre * 1.1283791670955126 * (-zb * exp(-zb * zb) + za * exp(-za * za)) / l;
            } break;
            case OAK_MEAS_EMPIRICAL: {  // oak/ortho_rbf_kernel.py:101-107
                const int K = dm.k[d];
                const double* loc = meas + dm.off[d];
                const double* w = loc + K;
                const double il2 = 0.5 / (l * l);
                double acc = 0.0, dacc = 0.0;
                for (int k = 0; k < K; ++k) {
                    const double u = x - loc[k];
                    const double e = w[k] * exp(-u * u * il2);
                    acc += e; dacc += e * u * u;
                }
                c = bv * acc;
                dc = bv * dacc / (l * l * l);
            } break;
            case OAK_MEAS_MOG: {        // oak/ortho_rbf_kernel.py:124-136
                const int K = dm.k[d];
                const double* mu = meas + dm.off[d];
                const double* var = mu + K;
                const double* w = var + K;
                double acc = 0.0, dacc = 0.0;
                for (int k = 0; k < K; ++k) {
                    const double s = l * l + var[k];
                    const double u = x - mu[k];
                    const double e = w[k] * exp(-0.5 * u * u / s) / sqrt(s);
                    acc += e; dacc += e * (-l / s + u * u * l / (s * s));
                }
                c = bv * l * acc;
                dc = c / l + bv * l * dacc;
            } break;
            default: c = 0.0; dc = 0.0;
        }
        vc = c * dm.inv_sqrt_v[d];
        // d/dl [ c / sqrt(v) ], pre-divided by 2 ln2 / l: the pair kernels form dk/dl in units of that factor (see grad.hip)
        vd = (dc - 0.5 * c * dm.dlogv[d]) * dm.inv_sqrt_v[d] * (l / 1.3862943611198906);
    } else {
        // tf.cast(float64 -> int32) truncates toward zero (ortho_binary_kernel.py:47); clamp keeps lookups in range
        double t = trunc(x);
        const double hi = (double)(dd.ncat[d] - 1);
        t = t < 0.0 ? 0.0 : (t > hi ? hi : t);
        vx = t; vc = 0.0;
        // the orthogonal binary kernel is rank one, bv * a(x) a(z) with a = (1 - p0, -p0) (ortho_binary_kernel.py:29-38): a(x) sqrt(bv) rides
        // in the cn slot and the forward pair kernel multiplies instead of gathering from the table (same products as the table's entries
        // when bv = 1)
        if (dd.type[d] == OAK_DIM_BINARY) vc = (t == 0.0 ? 1.0 - dm.p0[d] : -dm.p0[d]) * dm.p1[d];
    }
}

__device__ __forceinline__ void featurize_store(const DevDesc& dd, int d, int64_t idx, double vx, double vc, double vd,
                                                double* __restrict__ xs, double* __restrict__ cn, double* __restrict__ dcn,
                                                double* __restrict__ xs32, double* __restrict__ dcs) {
    xs[idx] = vx;
    cn[idx] = vc;
    if (dcn != nullptr) {
        dcn[idx] = vd;
        xs32[idx] = dd.type[d] == OAK_DIM_RBF ? vx * 0.03125 : vx;
        dcs[idx] = vd * 0.0009765625;
    }
}

// generic form: one thread per (point, dimension); the strided reads of X cost a 128-byte line per 8 bytes used
__global__ void __launch_bounds__(256) featurize_kernel(DevDesc dd, DevMeasure dm, const double* __restrict__ meas,
                                                        const double* __restrict__ X, int64_t n, int ldx, int64_t ld,
                                                        double* __restrict__ xs, double* __restrict__ cn, double* __restrict__ dcn,
                                                        double* __restrict__ xs32, double* __restrict__ dcs) {
    const int d = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ld) return;
    double vx = 0.0, vc = 0.0, vd = 0.0;
    if (i < n) featurize_point(dd, dm, meas, d, X[i * ldx + dd.col[d]], vx, vc, vd);
    featurize_store(dd, d, (int64_t)d * ld + i, vx, vc, vd, xs, cn, dcn, xs32, dcs);
}

// narrow inputs (ldx <= 31 columns): a workgroup reads its 256 rows of X ONCE, contiguously, into LDS (odd pitch: the
// column reads are conflict-free) and walks the dimensions; every output array is written dimension-major, coalesced.
// HBM traffic = the algorithmic bytes (the generic form fetched 16x the input at ldx = 16).
__global__ void __launch_bounds__(256) featurize_tile_kernel(DevDesc dd, DevMeasure dm, const double* __restrict__ meas,
                                                             const double* __restrict__ X, int64_t n, int ldx, int64_t ld,
                                                             double* __restrict__ xs, double* __restrict__ cn, double* __restrict__ dcn,
                                                             double* __restrict__ xs32, double* __restrict__ dcs,
                                                             const double* __restrict__ tables, double* __restrict__ kpart) {
    extern __shared__ __attribute__((aligned(16))) double tile[];      // [256][pitch]
    const int pitch = ldx | 1;
    const int64_t i0 = (int64_t)blockIdx.x * 256;
    const int64_t rows = (n - i0 < 256) ? (n - i0 > 0 ? n - i0 : 0) : 256;
    const double* src = X + i0 * ldx;
    for (int64_t e = threadIdx.x; e < rows * ldx; e += 256) {
        const int r = (int)(e / ldx), c = (int)(e - (int64_t)r * ldx);
        tile[r * pitch + c] = src[e];
    }
    __syncthreads();
    const int64_t i = i0 + threadIdx.x;
    // kpart != NULL: the workgroup also leaves the sum of K_diag over its rows (kappa of the SGPR bound) -- every k_d(x, x)
    // is at hand here, so the separate pass over the features (gram_diag + its reduction) drops off the critical path
    double e[OAK_MAX_DEPTH];
#pragma unroll
    for (int q = 0; q < OAK_MAX_DEPTH; ++q) e[q] = 0.0;
    const int R = dd.R;
    for (int d = 0; d < dd.D; ++d) {
        double vx = 0.0, vc = 0.0, vd = 0.0;
        if (i < n) featurize_point(dd, dm, meas, d, tile[threadIdx.x * pitch + dd.col[d]], vx, vc, vd);
        if (i < ld) featurize_store(dd, d, (int64_t)d * ld + i, vx, vc, vd, xs, cn, dcn, xs32, dcs);
        if (kpart != nullptr) {
            const double k = dd.type[d] == OAK_DIM_RBF ? __builtin_fma(-vc, vc, dd.bv[d])
                                                       : tables[dd.tab_off[d] + dd.ncat[d] * dd.ncat[d] + (int)vx];
#pragma unroll
            for (int q = OAK_MAX_DEPTH - 1; q >= 1; --q)
                if (q < R) e[q] = __builtin_fma(k, e[q - 1], e[q]);
            e[0] += k;
        }
    }
    if (kpart == nullptr) return;
    double kd = dd.w[0];
#pragma unroll
    for (int q = 0; q < OAK_MAX_DEPTH; ++q)
        if (q < R) kd = __builtin_fma(dd.w[q + 1], e[q], kd);
    __syncthreads();                                   // the X tile is no longer needed: reuse it for the reduction
    tile[threadIdx.x] = (i < n) ? kd : 0.0;
    tile[256 + threadIdx.x] = (i < n) ? kd : 0.0;     // ... and for the largest K_diag of the workgroup's rows (crt.hip: column scales)
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) {
            tile[threadIdx.x] += tile[threadIdx.x + off];
            tile[256 + threadIdx.x] = fmax(tile[256 + threadIdx.x], tile[256 + threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { kpart[blockIdx.x] = tile[0]; kpart[gridDim.x + blockIdx.x] = tile[256]; }
}

// tmp[k] = w_k * sum_l w_l * bv * exp(-(loc_k-loc_l)^2 / (2 l^2))   (var_s of the empirical measure, :109-120)
__global__ void empirical_var_kernel(const double* __restrict__ loc, const double* __restrict__ w, int K, double l,
                                     double bv, double* __restrict__ tmp, double* __restrict__ dtmp) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const double il2 = 0.5 / (l * l);
    double acc = 0.0, dacc = 0.0;
    for (int j = 0; j < K; ++j) {
        const double u = loc[k] - loc[j];
        const double e = w[j] * exp(-u * u * il2);
        acc += e; dacc += e * u * u;
    }
    tmp[k] = w[k] * bv * acc;
    dtmp[k] = w[k] * bv * dacc / (l * l * l);   // d/dl
}

// var_s of the measure and its lengthscale derivative (*dv)
static double host_var_s(int kind, double l, double bv, double p0, double p1, const double* data, int K, double* dv) {
    switch (kind) {
        case OAK_MEAS_GAUSSIAN: {                                                  // :94-97
            const double v = bv * l / std::sqrt(l * l + 2.0 * p1);
            *dv = v * (1.0 / l - l / (l * l + 2.0 * p1));
            return v;
        }
        case OAK_MEAS_UNIFORM: {                                                   // :65-78
            const double a = p0, b = p1;
            const double y = (b - a) / std::sqrt(2.0) / l;
            const double pre = 2.0 / ((b - a) * (b - a)) * bv * l * l;
            const double v = pre * (std::sqrt(M_PI) * y * std::erf(y) + std::exp(-y * y) - 1.0);
            *dv = 2.0 * v / l + pre * std::sqrt(M_PI) * std::erf(y) * (-y / l);
            return v;
        }
        case OAK_MEAS_MOG: {                                                       // :138-152
            const double *mu = data, *var = data + K, *w = data + 2 * K;
            double acc = 0.0, dacc = 0.0;
            for (int i = 0; i < K; ++i)
                for (int j = 0; j < K; ++j) {
                    const double s = l * l + var[i] + var[j];
                    const double dm = mu[i] - mu[j];
                    const double t = w[i] * w[j] * bv * l / std::sqrt(s) * std::exp(-0.5 * dm * dm / s);
                    acc += t;
                    dacc += t * (1.0 / l - l / s + dm * dm * l / (s * s));
                }
            *dv = dacc;
            return acc;
        }
        default: *dv = 0.0; return 0.0;
    }
}

int prepare_kernel(oak_ctx* ctx, const oak_kernel_desc* desc, PreparedKernel* pk, int allow) {
    OAK_REQUIRE(desc != nullptr, "kernel description is NULL");
    const int D = desc->num_dims, Rd = desc->max_depth;
    OAK_REQUIRE(D >= 1 && D <= OAK_MAX_DIMS, "num_dims=%d outside [1,%d]", D, OAK_MAX_DIMS);
    OAK_REQUIRE(Rd >= 0 && Rd <= OAK_MAX_DEPTH_DESC, "max_interaction_depth=%d outside [0,%d]", Rd, OAK_MAX_DEPTH_DESC);
    OAK_REQUIRE(desc->n_order_var == (desc->share_var ? Rd + 1 : 1), "n_order_var=%d inconsistent", desc->n_order_var);
    // the elementary symmetric polynomial e_r of D values is identically zero for r > D (the reference's Newton-Girard loop
    // computes those zeros, oak/oak_kernel.py:236-249): every kernel runs at depth min(R, D)
    const int R = Rd < D ? Rd : D;
    pk->R_desc = Rd;
    pk->deep = R > OAK_MAX_DEPTH;
    if (pk->deep && !(allow & PK_DEEP)) {
        set_error("effective interaction depth min(max_interaction_depth, num_dims) = %d exceeds the %d of the fused kernels "
                  "(only the explicit Gram entry points K / K_diag go deeper)", R, OAK_MAX_DEPTH);
        return OAK_E_ARG;
    }
    pk->grouped = false;
    pk->extra_off.assign((size_t)D + 1, 0);
    pk->extra_cols.clear();
    if (desc->extra_col_off != nullptr) {
        for (int d = 0; d < D; ++d) {
            const int n0 = desc->extra_col_off[d], n1 = desc->extra_col_off[d + 1];
            OAK_REQUIRE(n0 >= 0 && n1 >= n0 && n1 - n0 < 64, "extra_col_off[%d] invalid", d);
            if (n1 > n0) {
                OAK_REQUIRE(desc->extra_cols != nullptr && desc->dim_type[d] == OAK_DIM_RBF && desc->measure[d] == OAK_MEAS_NONE,
                            "sub-kernel %d: only an unconstrained RBF reads several columns (the constrained kernels are one-dimensional, "
                            "oak/ortho_rbf_kernel.py:50,83)", d);
                pk->grouped = true;
            }
            for (int q = n0; q < n1; ++q) {
                OAK_REQUIRE(desc->extra_cols[q] >= 0 && desc->extra_cols[q] < 32768, "extra column of sub-kernel %d invalid", d);
                pk->extra_cols.push_back(desc->extra_cols[q]);
            }
            pk->extra_off[d + 1] = (int)pk->extra_cols.size();
        }
    }
    if (pk->grouped && !(allow & PK_GROUPED)) {
        set_error("a sub-kernel over several columns (OAKKernel(active_dims=[[0, 1], ...])) is not evaluated by this entry point "
                  "(K / K_diag, the SGPR / GPR / SVGP objectives, their gradients and predictions are; the Sobol pass and "
                  "the fp32 Gram take one column per sub-kernel)");
        return OAK_E_ARG;
    }
    OAK_REQUIRE(pk->extra_cols.size() <= 64, "grouped sub-kernels: %zu extra columns (at most 64)", pk->extra_cols.size());
    DevDesc& dd = pk->dd;
    DevMeasure& dm = pk->dm;
    memset(&dd, 0, sizeof(dd));
    memset(&dm, 0, sizeof(dm));
    dd.D = D; dd.R = pk->deep ? OAK_MAX_DEPTH : R;
    pk->w_full.assign((size_t)R + 1, 0.0);
    for (int r = 0; r <= R; ++r) pk->w_full[r] = desc->share_var ? desc->order_var[r] : (r == 0 ? desc->order_var[0] : 1.0);
    for (int r = 0; r <= dd.R && r <= R; ++r) dd.w[r] = pk->w_full[r];
    for (int d = 0; d < D; ++d) { dd.xrow[d] = (short)pk->extra_off[d]; dd.nxc[d] = (unsigned char)(pk->extra_off[d + 1] - pk->extra_off[d]); }
    pk->extra_scale.assign(pk->extra_cols.size(), 0.0);
    pk->tables.clear();
    // upload measure data first (needed by the empirical variance kernel)
    double* d_meas = nullptr;
    const int mlen = desc->meas_data_len > 0 ? desc->meas_data_len : 0;
    OAK_CHECK(get_buf_t(ctx, "meas", (size_t)mlen + 1, &d_meas));
    if (mlen > 0)
        OAK_HIP_CHECK(hipMemcpyAsync(d_meas, desc->meas_data, sizeof(double) * mlen, hipMemcpyHostToDevice, ctx->stream));
    pk->d_meas = d_meas;
    for (int d = 0; d < D; ++d) {
        const int t = desc->dim_type[d];
        dd.type[d] = (unsigned char)t;
        OAK_REQUIRE(desc->active_col[d] >= 0 && desc->active_col[d] < 32768, "active_col[%d] invalid", d);
        dd.col[d] = (short)desc->active_col[d];
        const double bv = desc->base_var[d];
        dd.bv[d] = bv;
        if (t == OAK_DIM_RBF) {
            const double l = desc->lengthscale[d];
            OAK_REQUIRE(l > 0.0 && bv > 0.0, "dim %d: lengthscale and variance must be positive", d);
            dd.scale[d] = std::sqrt(0.5 * 1.4426950408889634074) / l;   // (x s - z s)^2 = (x-z)^2 log2(e) / (2 l^2)
            for (int q = pk->extra_off[d]; q < pk->extra_off[d + 1]; ++q) pk->extra_scale[q] = dd.scale[d];
            dd.log2bv[d] = std::log2(bv);
            {
                // n >= 0: the clamp w <= 1 bounds the exponent at n - 1024 and the biased table (x4) keeps the exponent field
                // positive only down to -1024; a base variance below 1 therefore goes entirely into woff
                OAK_REQUIRE(std::fabs(dd.log2bv[d]) <= 900.0, "dim %d: base variance out of range", d);
                const double n = std::fmax(std::ceil(dd.log2bv[d]), 0.0);
                dd.woff[d] = (n - dd.log2bv[d]) / 1024.0;
                dd.magic[d] = 6442450944.0 + n / 1024.0;        // EW_MAGIC (exp2w.h: 1.5 * 2^32) + n/1024
            }
            dd.ncat[d] = 0; dd.tab_off[d] = 0;
            const int kind = desc->measure[d];
            dm.kind[d] = (unsigned char)kind;
            dm.ls[d] = l;
            dm.p0[d] = desc->meas_p0 ? desc->meas_p0[d] : 0.0;
            dm.p1[d] = desc->meas_p1 ? desc->meas_p1[d] : 0.0;
            dm.k[d] = desc->meas_k ? desc->meas_k[d] : 0;
            dm.off[d] = desc->meas_off ? desc->meas_off[d] : 0;
            double v = 0.0, dv = 0.0;
            if (kind == OAK_MEAS_NONE) {
                dm.inv_sqrt_v[d] = 0.0;
                dm.dlogv[d] = 0.0;
            } else {
                if (kind == OAK_MEAS_EMPIRICAL) {
                    const int K = dm.k[d];
                    OAK_REQUIRE(K >= 1 && dm.off[d] + 2 * K <= mlen, "dim %d: empirical measure data out of range", d);
                    double* d_tmp = nullptr; double* d_s = nullptr;
                    OAK_CHECK(get_buf_t(ctx, "empvar_tmp", (size_t)2 * K, &d_tmp));
                    OAK_CHECK(get_buf_t(ctx, "empvar_s", 2, &d_s));
                    empirical_var_kernel<<<(K + 255) / 256, 256, 0, ctx->stream>>>(d_meas + dm.off[d], d_meas + dm.off[d] + K, K, l, bv, d_tmp, d_tmp + K);
                    OAK_CHECK(reduce_sum(ctx, d_tmp, K, d_s, 0, 1));
                    OAK_CHECK(reduce_sum(ctx, d_tmp + K, K, d_s + 1, 0, 1));
                    double hv[2] = {0, 0};
                    OAK_HIP_CHECK(hipMemcpyAsync(hv, d_s, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    v = hv[0]; dv = hv[1];
                } else {
                    if (kind == OAK_MEAS_MOG)
                        OAK_REQUIRE(dm.k[d] >= 1 && dm.off[d] + 3 * dm.k[d] <= mlen, "dim %d: MOG measure data out of range", d);
                    OAK_REQUIRE(kind == OAK_MEAS_GAUSSIAN || kind == OAK_MEAS_UNIFORM || kind == OAK_MEAS_MOG, "dim %d: unknown measure %d", d, kind);
                    v = host_var_s(kind, l, bv, dm.p0[d], dm.p1[d], desc->meas_data ? desc->meas_data + dm.off[d] : nullptr, dm.k[d], &dv);
                }
                OAK_REQUIRE(v > 0.0 && std::isfinite(v), "dim %d: measure variance var_s=%g is not positive", d, v);
                dm.inv_sqrt_v[d] = 1.0 / std::sqrt(v);
                dm.dlogv[d] = dv / v;
            }
        } else if (t == OAK_DIM_BINARY) {     // oak/ortho_binary_kernel.py:29-38
            const double p0 = desc->meas_p0[d], p1 = 1.0 - p0;
            dm.p0[d] = p0; dm.p1[d] = std::sqrt(bv);          // featurize: the rank-one factor a(x) sqrt(bv)
            dd.ncat[d] = 2;
            dd.tab_off[d] = (int)pk->tables.size();
            const double tab[6] = {p1 * p1 * bv, -p0 * p1 * bv, -p0 * p1 * bv, p0 * p0 * bv, p1 * p1 * bv, p0 * p0 * bv};
            pk->tables.insert(pk->tables.end(), tab, tab + 6);
        } else if (t == OAK_DIM_CATEGORICAL) {   // oak/ortho_categorical_kernel.py:34-53 (table built by the host mirror)
            const int C = desc->meas_k[d];
            const int off = desc->meas_off[d];
            OAK_REQUIRE(C >= 1 && C <= 4096 && off >= 0 && off + C * C + C <= mlen, "dim %d: categorical table out of range", d);
            dd.ncat[d] = C;
            dd.tab_off[d] = (int)pk->tables.size();
            for (int i = 0; i < C * C; ++i) pk->tables.push_back(desc->meas_data[off + i] * bv);
            for (int i = 0; i < C; ++i) pk->tables.push_back(desc->meas_data[off + i * C + i] * bv);
            dm.k[d] = C; dm.off[d] = off;
        } else {
            set_error("dim %d: unknown dim_type %d", d, t);
            return OAK_E_ARG;
        }
    }
    double* d_tab = nullptr;
    OAK_CHECK(get_buf_t(ctx, "tables", pk->tables.size() + 1, &d_tab));
    if (!pk->tables.empty())
        OAK_HIP_CHECK(hipMemcpyAsync(d_tab, pk->tables.data(), sizeof(double) * pk->tables.size(), hipMemcpyHostToDevice, ctx->stream));
    pk->d_tables = d_tab;
    // the host vector `tables` must outlive the async copy: synchronise (tiny copy, once per evaluation)
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OAK_OK;
}

int prepare_component(oak_ctx* ctx, const oak_kernel_desc* desc, const int32_t* subset, int32_t len,
                      int32_t apply_order_var, PreparedKernel* pk) {
    OAK_CHECK(prepare_kernel(ctx, desc, pk, PK_GROUPED));
    OAK_REQUIRE(len >= 0 && len <= desc->num_dims, "subset length %d invalid", len);
    // Build a description whose D sub-kernels are the subset and whose only non-zero ESP weight is e_len:
    //   K_S = sigma2_{|S|} * prod_{d in S} k_d = w_len * e_len(k_S)   (oak/oak_kernel.py:300-320)
    PreparedKernel full = *pk;
    DevDesc& dd = pk->dd; DevMeasure& dm = pk->dm;
    OAK_REQUIRE(len <= OAK_MAX_DEPTH, "component order %d exceeds OAK_MAX_DEPTH", len);
    double wv = 1.0;
    if (len == 0) wv = desc->order_var[0];
    else if (apply_order_var) {
        OAK_REQUIRE(desc->share_var && len <= desc->max_depth, "component order %d has no order variance", len);
        wv = desc->order_var[len];
    }
    for (int r = 0; r <= OAK_MAX_DEPTH; ++r) dd.w[r] = 0.0;
    dd.w[len] = wv;
    dd.R = len;
    pk->w_full.assign((size_t)len + 1, 0.0);        // the generic kernel (reference arithmetic, grouped sub-kernels) reads its weights here
    pk->w_full[len] = wv;
    if (len == 0) { dd.D = 1; return OAK_OK; }   // constant term: R = 0 -> K = w0 regardless of dims
    dd.D = len;
    for (int q = 0; q < len; ++q) {
        const int s = subset[q];
        OAK_REQUIRE(s >= 0 && s < desc->num_dims, "subset entry %d out of range", s);
        dd.type[q] = full.dd.type[s]; dd.col[q] = full.dd.col[s]; dd.ncat[q] = full.dd.ncat[s];
        dd.tab_off[q] = full.dd.tab_off[s]; dd.scale[q] = full.dd.scale[s]; dd.log2bv[q] = full.dd.log2bv[s];
        dd.bv[q] = full.dd.bv[s]; dd.woff[q] = full.dd.woff[s]; dd.magic[q] = full.dd.magic[s];
        dd.xrow[q] = full.dd.xrow[s]; dd.nxc[q] = full.dd.nxc[s];      // rows of Feat::xx, which always holds every extra column
        dm.kind[q] = full.dm.kind[s]; dm.k[q] = full.dm.k[s]; dm.off[q] = full.dm.off[s]; dm.p0[q] = full.dm.p0[s];
        dm.p1[q] = full.dm.p1[s]; dm.ls[q] = full.dm.ls[s]; dm.inv_sqrt_v[q] = full.dm.inv_sqrt_v[s]; dm.dlogv[q] = full.dm.dlogv[s];
    }
    return OAK_OK;
}

// further columns of grouped sub-kernels, scaled like the owning dim's first column (zero past n, like xs)
struct ExtraCols { short col[64]; double scale[64]; };
__global__ void __launch_bounds__(256) featurize_extra_kernel(ExtraCols ec, const double* __restrict__ X, int64_t n, int ldx, int64_t ld,
                                                              double* __restrict__ xx) {
    const int q = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ld) return;
    xx[(int64_t)q * ld + i] = i < n ? X[i * ldx + ec.col[q]] * ec.scale[q] : 0.0;
}

int featurize(oak_ctx* ctx, const PreparedKernel& pk, const double* dX, int64_t n, int32_t ldx, const char* bufname, Feat* out,
              bool with_grad, double* d_kdiag_sum, bool* kdiag_done) {
    if (kdiag_done) *kdiag_done = false;
    const int D = pk.dd.D;
    const int64_t ld = ((n + 63) / 64) * 64 + 64;   // padded so tile loads never run past the array
    double* base = nullptr;
    const int nx = (int)pk.extra_cols.size();
    OAK_CHECK(get_buf_t(ctx, bufname, (size_t)((with_grad ? 5 : 2) * D + nx) * ld, &base));
    out->xs = base; out->cn = base + (size_t)D * ld; out->n = n; out->ld = ld;
    out->nx = nx; out->xx = nx > 0 ? base + (size_t)(with_grad ? 5 : 2) * D * ld : nullptr;
    if (nx > 0) {
        ExtraCols ec;
        for (int q = 0; q < nx; ++q) { ec.col[q] = (short)pk.extra_cols[q]; ec.scale[q] = pk.extra_scale[q]; }
        featurize_extra_kernel<<<dim3((unsigned)((ld + 255) / 256), (unsigned)nx), 256, 0, ctx->stream>>>(ec, dX, n, ldx, ld, out->xx);
        OAK_HIP_CHECK(hipGetLastError());
    }
    out->dcn = with_grad ? base + (size_t)2 * D * ld : nullptr;
    out->xs32 = with_grad ? base + (size_t)3 * D * ld : nullptr;
    out->dcs = with_grad ? base + (size_t)4 * D * ld : nullptr;
    if (ldx <= 31 && n >= 4096) {
        size_t lds = sizeof(double) * 256 * (size_t)(ldx | 1);
        if (d_kdiag_sum != nullptr && lds < sizeof(double) * 512) lds = sizeof(double) * 512;      // the two reductions' scratch
        const unsigned nblk = (unsigned)((ld + 255) / 256);
        double* d_kpart = nullptr;
        if (d_kdiag_sum != nullptr) OAK_CHECK(get_buf_t(ctx, "feat_kpart", (size_t)2 * nblk, &d_kpart));      // [sums | maxima] per workgroup
        featurize_tile_kernel<<<nblk, 256, lds, ctx->stream>>>(pk.dd, pk.dm, pk.d_meas, dX, n, ldx, ld, out->xs, out->cn, out->dcn,
                                                             out->xs32, out->dcs, pk.d_tables, d_kpart);
        if (d_kdiag_sum != nullptr) {
            OAK_HIP_CHECK(hipGetLastError());
            OAK_CHECK(reduce_sum(ctx, d_kpart, nblk, d_kdiag_sum, 0, 1));      // fixed-order tree over the workgroup sums
            if (kdiag_done) *kdiag_done = true;
        }
    } else {
        dim3 grid((unsigned)((ld + 255) / 256), (unsigned)D);
        featurize_kernel<<<grid, 256, 0, ctx->stream>>>(pk.dd, pk.dm, pk.d_meas, dX, n, ldx, ld, out->xs, out->cn, out->dcn, out->xs32,
                                                         out->dcs);
    }
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

}  // namespace oak

// ---------------------------------------------------------------------------------------------
// C ABI: runtime
// ---------------------------------------------------------------------------------------------
namespace oak {
int copy_sync(oak_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    if (bytes == 0) return OAK_OK;
    OAK_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OAK_OK;
}

int ensure_max_dynamic_lds(const void* kernel) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;     // the attribute belongs to the current device's copy of the function
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    OAK_HIP_CHECK(hipGetDevice(&dev));
    if (done.count({dev, kernel})) return OAK_OK;
    OAK_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.insert({dev, kernel});
    return OAK_OK;
}
// the same for a kernel that also has static LDS: the attribute is the DYNAMIC size, and static + dynamic must fit 160 KiB
int ensure_dynamic_lds(const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> done;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    OAK_HIP_CHECK(hipGetDevice(&dev));
    auto it = done.find({dev, kernel});
    if (it != done.end() && it->second >= bytes) return OAK_OK;
    OAK_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done[{dev, kernel}] = bytes;
    return OAK_OK;
}
}  // namespace oak

extern "C" {

const char* oak_last_error(void) { return oak::g_err; }
const char* oak_version(void) { return "oak_hip 0.1.0 (gfx950)"; }

int oak_device_count(int* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; oak::set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return OAK_E_HIP; }
    *count = n;
    return OAK_OK;
}

namespace oak {
// The side stream carries the latency-bound chol(Kuu) / L^-1 chain next to the DP-saturating Gram and SYRK kernels: give it
// the highest queue priority so its small kernels are dispatched as soon as their predecessors retire.
static hipError_t create_side_stream(hipStream_t* s) {
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo &&
        hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi) == hipSuccess)
        return hipSuccess;
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
// The partition pair.  Mask bit i stands for compute unit i / 8 of XCD i % 8 (tools/ubench/cumask_probe.hip: bits 0..15 = two CUs in
// each of the eight XCDs; a mask that leaves an XCD empty is ignored by the driver), so both masks keep all XCDs with equal
// shares and every XCD-aware workgroup mapping stays valid.  OAK_PART_CUS = 0 disables, 8 / 16 / ... / 64 sets the side's share (default 32).
static int requested_part_cus() {
    int side_cus = 32;          // 4 CUs in each XCD: N/16 shards gain with 32 and lose with 16, N/8 and N/4 measure the same with either
    if (const char* e = getenv("OAK_PART_CUS")) side_cus = atoi(e);
    return (side_cus <= 0 || side_cus % 8 != 0 || side_cus > 64) ? 0 : side_cus;
}
static std::mutex g_orphan_mu;
static std::vector<hipStream_t> g_orphans;          // streams of a partition pair whose creation failed half-way (destroyed by shutdown_pool)
static void park_orphan_stream(hipStream_t s) { std::lock_guard<std::mutex> lock(g_orphan_mu); g_orphans.push_back(s); }
static void create_partition_streams(StreamSet* ss, int num_cu) {
    const int side_cus = requested_part_cus();
    ss->part_cus_req = side_cus;
    if (side_cus == 0 || num_cu != 256) return;
    uint32_t ms[8], mm[8];
    for (int w = 0; w < 8; ++w) { ms[w] = 0u; mm[w] = 0xffffffffu; }
    for (int b = 0; b < side_cus; ++b) { ms[b >> 5] |= 1u << (b & 31); mm[b >> 5] &= ~(1u << (b & 31)); }
    // NB hipExtStreamCreateWithCUMask takes no flags: the pair is created with hipStreamDefault semantics, i.e. it synchronises with the
    // legacy NULL stream (the unmasked pair is hipStreamNonBlocking).  The library itself never uses the NULL stream; a host process
    // that does serialises its NULL-stream work with partitioned passes.
    // A half-created pair is PARKED, not destroyed: this can run in the middle of an evaluation while other threads use the device,
    // and hipStreamDestroy is the call that deadlocks there (see below); oak_runtime_shutdown destroys the parked streams.
    hipStream_t a = nullptr, b = nullptr;
    if (hipExtStreamCreateWithCUMask(&a, 8, mm) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipExtStreamCreateWithCUMask(&b, 8, ms) != hipSuccess) { (void)hipGetLastError(); park_orphan_stream(a); return; }
    if (hipEventCreateWithFlags(&ss->ev3, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); park_orphan_stream(a); park_orphan_stream(b); return; }
    ss->main_part = a; ss->side_part = b; ss->part_cus = side_cus;
}

// Streams and events are POOLED per process and never destroyed.  hipStreamDestroy on this runtime (ROCm 7.2) can deadlock
// against the HSA runtime's asynchronous-event thread while other host threads use the device: the destroying thread holds a
// runtime lock while it tears the queue down, the event thread's completion callback waits for that lock, and the teardown does
// not finish (native stacks of the stall: profiles/r05_stream_destroy_deadlock.txt; it is the intermittent suite stall of
// rounds 2-4, reproduced in 10-80 iterations of tools/soak.py --only threads).  A context therefore hands its idle streams back
// on destruction and the next context of that device takes them over.
static std::mutex g_pool_mu;
static std::vector<StreamSet> g_pool;
// oak_runtime_shutdown: the idle (pooled) streams are synchronised and destroyed -- to be called by the host when no other
// thread uses the device any more (oak/_capi.py does at interpreter exit, after closing its contexts).  Not from a C atexit
// handler: under rocprofv3 the tool's per-thread state is gone by then and hipStreamDestroy aborts, while CU-masked queues
// left alive make its finalisation crash.
static int shutdown_pool() {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    for (const StreamSet& ss : g_pool) {
        if (hipSetDevice(ss.device) != hipSuccess) continue;
        for (hipStream_t st : {ss.main, ss.side, ss.main_part, ss.side_part})
            if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        for (hipEvent_t ev : {ss.ev0, ss.ev1, ss.ev2, ss.ev3})
            if (ev) (void)hipEventDestroy(ev);
    }
    g_pool.clear();
    {
        std::lock_guard<std::mutex> lock2(g_orphan_mu);
        for (hipStream_t st : g_orphans) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        g_orphans.clear();
    }
    return OAK_OK;
}
static int acquire_streams(int device, int num_cu, StreamSet* out) {
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        const int want = requested_part_cus();
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].device == device && g_pool[i].part_cus_req == want) {
                *out = g_pool[i];
                g_pool.erase(g_pool.begin() + (long)i);
                return OAK_OK;
            }
    }
    StreamSet ss;
    ss.device = device;
    // both streams are non-blocking: no implicit coupling to the legacy NULL stream, hence none between contexts
    if (hipStreamCreateWithFlags(&ss.main, hipStreamNonBlocking) != hipSuccess || create_side_stream(&ss.side) != hipSuccess ||
        hipEventCreateWithFlags(&ss.ev0, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ss.ev1, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ss.ev2, hipEventDisableTiming) != hipSuccess) {
        set_error("hipStreamCreate / hipEventCreate failed");
        return OAK_E_HIP;
    }
    ss.part_cus_req = requested_part_cus();       // the CU-masked pair itself is made on first use (ensure_partition_streams)
    (void)num_cu;
    *out = ss;
    return OAK_OK;
}
// The partition pair of a context, made when its first partitioned pass is about to run: a process that only ever evaluates large
// problems (the headline bench) never owns CU-masked queues.  false when the driver refuses them (then the pass runs unpartitioned).
extern "C++" bool ensure_partition_streams(oak_ctx* ctx) {
    if (ctx->main_part != nullptr) return true;
    if (ctx->part_tried) return false;
    ctx->part_tried = true;
    StreamSet ss;
    create_partition_streams(&ss, ctx->num_cu);
    if (ss.main_part == nullptr) return false;
    ctx->main_part = ss.main_part; ctx->side_part = ss.side_part; ctx->ev3 = ss.ev3; ctx->part_cus = ss.part_cus;
    return true;
}
static void release_streams(const StreamSet& ss) {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    g_pool.push_back(ss);
}
}  // namespace oak

int oak_ctx_create(int device, oak_ctx** out) {
    if (!out) { oak::set_error("out is NULL"); return OAK_E_ARG; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        oak::set_error("no HIP device available (%s): the OAK HIP path has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
        return OAK_E_HIP;
    }
    if (device < 0 || device >= n) { oak::set_error("device %d out of range [0,%d)", device, n); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    OAK_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    // The library holds gfx950 code objects only, and several kernels are sized for the 160 KiB of LDS of an MI355X CU (fused
    // triangular solve, blocked Cholesky, Gram tiles at 32+ sub-kernels): say so here instead of failing inside an evaluation.
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        oak::set_error("device %d is %s: this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return OAK_E_HIP;
    }
    oak_ctx* ctx = new oak_ctx();
    ctx->device = device;
    ctx->num_cu = prop.multiProcessorCount;
    oak::StreamSet ss;
    if (oak::acquire_streams(device, ctx->num_cu, &ss) != OAK_OK) { delete ctx; return OAK_E_HIP; }
    ctx->stream = ctx->main_full = ss.main; ctx->side = ctx->side_full = ss.side;
    ctx->ev0 = ss.ev0; ctx->ev1 = ss.ev1; ctx->ev2 = ss.ev2; ctx->ev3 = ss.ev3;
    ctx->main_part = ss.main_part; ctx->side_part = ss.side_part; ctx->part_cus = ss.part_cus;      // a pooled set may bring its pair along
    ctx->part_cus_req = ss.part_cus_req;
    // development / A-B knob: the statistics precision every new context starts with (oak_sgpr_set_precision overrides it)
    if (const char* e = getenv("OAK_PRECISION")) { const int v = atoi(e); if (v >= -1 && v <= 2) ctx->precision = v; }
    { std::lock_guard<std::mutex> lock(oak::g_ctx_mu); oak::g_ctxs.insert(ctx); }
    *out = ctx;
    return OAK_OK;
}

/* Post-mortem aid for a stalled process: what every live context last enqueued and whether its streams have drained.  Meant to
   be called from ANOTHER host thread than the stuck one (a test watchdog, tools/soak.py); takes no stream-level locks. */
int oak_debug_state(char* buf, int64_t cap) {
    if (!buf || cap <= 0) { oak::set_error("bad argument"); return OAK_E_ARG; }
    std::string out;
    char line[512];
    const double now = oak::wall_s();
    std::lock_guard<std::mutex> lock(oak::g_ctx_mu);
    snprintf(line, sizeof line, "%zu live context(s)\n", oak::g_ctxs.size()); out += line;
    for (oak_ctx* c : oak::g_ctxs) {
        const hipError_t qm = hipStreamQuery(c->stream), qs = hipStreamQuery(c->side);
        snprintf(line, sizeof line, "ctx %p dev %d: main stream %s, side stream %s, comm %s (rank %d of %d), N=%lld M=%lld route=%d\n", (void*)c,
                 c->device, qm == hipSuccess ? "idle" : (qm == hipErrorNotReady ? "BUSY" : hipGetErrorString(qm)),
                 qs == hipSuccess ? "idle" : (qs == hipErrorNotReady ? "BUSY" : hipGetErrorString(qs)),
                 c->comm == nullptr ? "none" : (c->host_allreduce ? "host" : "rccl/loopback"), c->rank, c->nranks, (long long)c->N,
                 (long long)c->M, c->route);
        out += line;
        const unsigned n = c->mark_n;
        for (unsigned k = (n > 16 ? n - 16 : 0); k < n; ++k) {
            const char* m = c->marks[k & 15u];
            snprintf(line, sizeof line, "    [%u] %-14s %.6f s ago\n", k, m ? m : "?", now - c->mark_t[k & 15u]);
            out += line;
        }
    }
    strncpy(buf, out.c_str(), (size_t)cap - 1);
    buf[cap - 1] = 0;
    return OAK_OK;
}

int oak_ctx_destroy(oak_ctx* ctx) {
    if (!ctx) return OAK_OK;
    { std::lock_guard<std::mutex> lock(oak::g_ctx_mu); oak::g_ctxs.erase(ctx); }
    (void)hipSetDevice(ctx->device);
    ctx->stream = ctx->main_full; ctx->side = ctx->side_full;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->side);
    if (ctx->main_part) { (void)hipStreamSynchronize(ctx->main_part); (void)hipStreamSynchronize(ctx->side_part); }
    oak::reset_timings(ctx);
    oak_comm_destroy(ctx);
    for (auto& kv : ctx->bufs) if (kv.second.p) (void)hipFree(kv.second.p);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    {   // the (idle) streams and events go back to the pool: see acquire_streams for why they are never destroyed
        oak::StreamSet ss;
        ss.device = ctx->device; ss.main = ctx->main_full; ss.side = ctx->side_full; ss.main_part = ctx->main_part; ss.side_part = ctx->side_part;
        ss.ev0 = ctx->ev0; ss.ev1 = ctx->ev1; ss.ev2 = ctx->ev2; ss.ev3 = ctx->ev3; ss.part_cus = ctx->part_cus; ss.part_cus_req = ctx->part_cus_req;
        oak::release_streams(ss);
    }
    delete ctx;
    return OAK_OK;
}

int oak_runtime_shutdown(void) { return oak::shutdown_pool(); }

int oak_sync(oak_ctx* ctx) {
    if (!ctx) { oak::set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->side));          // this context's work only: other contexts are not waited for
    if (ctx->main_part) { OAK_HIP_CHECK(hipStreamSynchronize(ctx->side_part)); OAK_HIP_CHECK(hipStreamSynchronize(ctx->main_part)); }
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OAK_OK;
}

int oak_last_timing(oak_ctx* ctx, const char* name, double* ms, int32_t* count) {
    if (!ctx || !name) { oak::set_error("bad argument"); return OAK_E_ARG; }
    oak::flush_timings(ctx);
    auto it = ctx->timings.find(name);
    if (it == ctx->timings.end()) { if (ms) *ms = 0; if (count) *count = 0; return OAK_OK; }
    if (ms) *ms = it->second.ms;
    if (count) *count = it->second.count;
    return OAK_OK;
}

int oak_reset_timings(oak_ctx* ctx) {
    if (!ctx) { oak::set_error("ctx is NULL"); return OAK_E_ARG; }
    oak::reset_timings(ctx);
    return OAK_OK;
}

int oak_device_mem_info(oak_ctx* ctx, double* free_bytes, double* total_bytes) {
    if (!ctx) { oak::set_error("ctx is NULL"); return OAK_E_ARG; }
    size_t f = 0, t = 0;
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (double)f;
    if (total_bytes) *total_bytes = (double)t;
    return OAK_OK;
}

}  // extern "C"

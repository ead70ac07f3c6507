// Fused OAK Gram kernel: constrained base kernels (RBF / binary / categorical) + elementary symmetric
// polynomial combination, one pass, nothing but the final K tile leaves registers.
//
// Replaces OAKKernel.K / K_diag (oak/oak_kernel.py:251-278): the reference materialises D per-dimension
// N x M matrices (:252-254), R+1 power sums and R Newton-Girard terms (:236-249).  Here each lane owns an
// RT x CPT block of (row, column) pairs and runs the e_r recurrence  e_r += k_d * e_{r-1}  (r = R..1) over d in
// registers -- algebraically the same elementary symmetric polynomials, D*R FMAs per pair and no pow().
//
// Roofline: fp64-VALU bound (one software exp2 per pair per dimension: ~17 DP ops), not HBM bound.
// fp64 MFMA shares the DP pipe with fp64 VALU on gfx950 (measured, tools/ubench), so there is nothing to
// overlap with; the kernel is written to issue the minimum number of DP instructions per pair.
#include "oak_internal.h"

namespace oak {

// 2^t for t <= 0, for CPT independent arguments at once (interleaved so the DP pipe always has independent work):
//   a = t + 1.5*2^46 rounds t to a multiple of 1/64 (round-to-nearest); its low mantissa bits hold k = 64*rint-part, so
//   j = k & 63 indexes a 64-entry table of 2^(j/64) (LDS), e = k >> 6 is added straight into the exponent field, and
//   r = t - k/64, |r| <= 1/128, needs only a degree-5 polynomial: 2^t = 2^e * T[j] * (1 + r*(c1 + ... + r*c5)).
// Max error 1.6 ulp against 50-digit arithmetic (tests/test_gpu_gram.py::test_exp2_accuracy); t is clamped at -1020 so the
// exponent arithmetic cannot wrap (values below 2^-1020 are far under the 1e-12 parity tolerance of any Gram entry).
__constant__ double c_exp2_table[64] = {
    1.0, 1.01088928605170046, 1.0218971486541166782, 1.0330248790212284225,
    1.0442737824274138403, 1.0556451783605571588, 1.0671404006768236182, 1.0787607977571197937,
    1.0905077326652576592, 1.1023825833078409436, 1.1143867425958925363, 1.1265216186082418998,
    1.1387886347566916537, 1.1511892299529827058, 1.1637248587775775138, 1.1763969916502812763,
    1.1892071150027210667, 1.2021567314527031421, 1.2152473599804688781, 1.2284805361068700057,
    1.2418578120734840486, 1.2553807570246910896, 1.2690509571917332226, 1.2828700160787782807,
    1.2968395546510096659, 1.3109612115247643419, 1.3252366431597412946, 1.3396675240533030054,
    1.3542555469368927283, 1.3690024229745906119, 1.3839098819638319549, 1.3989796725383111402,
    1.4142135623730950488, 1.4296133383919700112, 1.44518080697704662, 1.4609177941806469887,
    1.4768261459394993114, 1.4929077282912648492, 1.5091644275934227398, 1.5255981507445383069,
    1.5422108254079408236, 1.559004400237836967, 1.5759808451078864865, 1.5931421513422668979,
    1.6104903319492543082, 1.6280274218573477668, 1.6457554781539648445, 1.663676580326736435,
    1.6817928305074290861, 1.7001063537185234695, 1.7186192981224779156, 1.737333835273706249,
    1.7562521603732994831, 1.7753764925265212526, 1.7947090750031071864, 1.8142521755003987562,
    1.8340080864093424635, 1.8539791250833855684, 1.8741676341102999013, 1.8945759815869656413,
    1.9152065613971472939, 1.9360617934922944506, 1.957144124175400269, 1.9784560263879509683,
};

template <int NV>
__device__ __forceinline__ void exp2_neg_vec(const double (&t_in)[NV], double (&out)[NV], const double* __restrict__ tab) {
    constexpr double c1 = 6.931471805599453094e-01, c2 = 2.402265069591007123e-01, c3 = 5.550410866482157995e-02,
                     c4 = 9.618129107628477162e-03, c5 = 1.333355814642844342e-03;
    constexpr double MAGIC = 105553116266496.0;   // 1.5 * 2^46: ulp = 2^-6
    double t[NV], a[NV], r[NV], p[NV], tv[NV];
    int ki[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) t[v] = __builtin_fmax(t_in[v], -1020.0);
#pragma unroll
    for (int v = 0; v < NV; ++v) a[v] = t[v] + MAGIC;
#pragma unroll
    for (int v = 0; v < NV; ++v) ki[v] = __double2loint(a[v]);
#pragma unroll
    for (int v = 0; v < NV; ++v) tv[v] = tab[ki[v] & 63];
#pragma unroll
    for (int v = 0; v < NV; ++v) r[v] = t[v] - (a[v] - MAGIC);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(c5, r[v], c4);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], c3);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], c2);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], c1);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], 1.0);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        // scale the table value by 2^e through its exponent field (T[j] in [1,2), e >= -1020: always a normal number)
        const int hi = __double2hiint(tv[v]) + (ki[v] >> 6) * 1048576;
        out[v] = __hiloint2double(hi, __double2loint(tv[v])) * p[v];
    }
}


// Gram-kernel form of the same evaluation, two VALU instructions shorter per pair.  The caller passes
//   w = clamp01((xa' - xb')^2 + woff),   xa' = xs/32,   so that  t = n - 1024 w  is the base-2 exponent of the pair
// (n = ceil(log2 bv) rides in `magic` = 1.5*2^36 + n/1024, woff = (n - log2 bv)/1024).  The clamp is the free VOP3
// output modifier, so t >= n - 1024 needs no v_max.  a = magic - w rounds w to a multiple of 2^-16 (t to 1/64); the low
// mantissa word of a is k = 64 n + 64 rint-part(t) exactly as in exp2_neg_vec.  The table entries are biased:
//   Tb[j] = 4 * 2^(j/64) with (j << 14) subtracted from the high word, so that  hi + (k << 14) = hi(4 T[j]) + (e << 20)
// patches the exponent with ONE v_lshl_add_u32 (no mask, no arithmetic shift); the factor 4 keeps the exponent field
// positive down to e = -1024 and is folded into the polynomial (constant term 0.25, coefficients c_i * (-1024)^i / 4 in
// the variable rw = w - rounded(w), |rw| <= 2^-17).
__device__ __forceinline__ double biased_table_entry(int j) {
    const double t4 = 4.0 * c_exp2_table[j];
    return __hiloint2double(__double2hiint(t4) - (j << 14), __double2loint(t4));
}

__device__ __forceinline__ double fma_clamp01(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int NV>
__device__ __forceinline__ void exp2_w_vec(const double (&w)[NV], const double magic, double (&out)[NV], const double* __restrict__ tab) {
    constexpr double c1 = 6.931471805599453094e-01, c2 = 2.402265069591007123e-01, c3 = 5.550410866482157995e-02,
                     c4 = 9.618129107628477162e-03, c5 = 1.333355814642844342e-03;
    constexpr double S = -1024.0;
    constexpr double C1 = 0.25 * c1 * S, C2 = 0.25 * c2 * S * S, C3 = 0.25 * c3 * S * S * S, C4 = 0.25 * c4 * S * S * S * S,
                     C5 = 0.25 * c5 * S * S * S * S * S;
    double a[NV], r[NV], p[NV], tv[NV];
    int ki[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) a[v] = magic - w[v];
#pragma unroll
    for (int v = 0; v < NV; ++v) ki[v] = __double2loint(a[v]);
#pragma unroll
    for (int v = 0; v < NV; ++v) tv[v] = tab[ki[v] & 63];
#pragma unroll
    for (int v = 0; v < NV; ++v) r[v] = w[v] + (a[v] - magic);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(C5, r[v], C4);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], C3);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], C2);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], C1);
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v] = __builtin_fma(p[v], r[v], 0.25);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int hi = __double2hiint(tv[v]) + (ki[v] << 14);
        out[v] = __hiloint2double(hi, __double2loint(tv[v])) * p[v];
    }
}


template <int R>
__device__ __forceinline__ void esp_update(double (&e)[R > 0 ? R : 1], double k) {
    if constexpr (R > 0) {
#pragma unroll
        for (int q = R - 1; q >= 1; --q) e[q] = __builtin_fma(k, e[q - 1], e[q]);
        e[0] += k;
    }
}

template <int R>
__device__ __forceinline__ double esp_combine(const double (&e)[R > 0 ? R : 1], const DevDesc& dd) {
    double K = dd.w[0];
    if constexpr (R > 0) {
#pragma unroll
        for (int q = 0; q < R; ++q) K = __builtin_fma(dd.w[q + 1], e[q], K);
    }
    return K;
}

// Tile geometry: 256 threads = 4 waves.  Lane tx (0..63) owns CPT columns, wave ty owns RT rows per row-step.
//   CPT == 4: columns jb + 2*tx + {0,1} and jb + 128 + 2*tx + {0,1}   (two 16-byte stores per row)
//   CPT == 2: columns jb + 2*tx + {0,1}
// The B-side (column) features of all D dims stay in LDS for the whole workgroup; A-side (row) features are
// restaged per row-step.  Dynamic LDS = (64 + D*TJ*2 + D*4*RT*2 + 4*RT) doubles.
template <int R, int RT, int CPT, bool ALLRBF>
__global__ void __launch_bounds__(256)
gram_kernel(const DevDesc dd, const double* __restrict__ tables, const double* __restrict__ Axs,
            const double* __restrict__ Acn, int64_t a_ld, int64_t a0, int64_t na, const double* __restrict__ Bxs,
            const double* __restrict__ Bcn, int64_t b_ld, int64_t nb, double* __restrict__ out, int64_t ldo,
            int rows_per_wg, const double* __restrict__ yA, double* __restrict__ psi_part, int64_t zero_pad_to) {
    constexpr int TJ = 64 * CPT;
    constexpr int RS = 4 * RT;   // rows per row-step
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int D = dd.D;
    double* Tab = smem;                // [64]  biased 4 * 2^(j/64); first, so the lookups use a constant LDS base
    double* Bx = Tab + 64;             // [D][TJ]   (RBF dims: x * scale_d / 32)
    double* Bc = Bx + D * TJ;          // [D][TJ]
    double* Ax = Bc + D * TJ;          // [D][RS]
    double* Ac = Ax + D * RS;          // [D][RS]
    double* Ay = Ac + D * RS;          // [RS]
    const int tid = threadIdx.x;
    const int tx = tid & 63;
    const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t jb = (int64_t)blockIdx.x * TJ;
    const int64_t ib = (int64_t)blockIdx.y * rows_per_wg;      // relative to a0
    const int64_t iend = (ib + rows_per_wg < na) ? ib + rows_per_wg : na;

    // stage B-side features (coalesced over columns); Feat arrays are padded so reads past nb are safe zeros
    for (int idx = tid; idx < D * TJ; idx += 256) {
        const int d = idx / TJ, j = idx - d * TJ;
        const int64_t gj = jb + j;
        const bool ok = gj < nb;
        const double pre = (ALLRBF || dd.type[d] == OAK_DIM_RBF) ? 0.03125 : 1.0;
        Bx[idx] = ok ? Bxs[(int64_t)d * b_ld + gj] * pre : 0.0;
        Bc[idx] = ok ? Bcn[(int64_t)d * b_ld + gj] : 0.0;
    }
    if (tid < 64) Tab[tid] = biased_table_entry(tid);
    double psi[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) psi[c] = 0.0;

    for (int64_t i0 = ib; i0 < iend; i0 += RS) {
        __syncthreads();   // previous step's readers done (and B-side staged on first pass)
        for (int idx = tid; idx < D * RS; idx += 256) {
            const int d = idx / RS, r = idx - d * RS;
            const int64_t gi = i0 + r;
            const bool ok = gi < iend;
            const double pre = (ALLRBF || dd.type[d] == OAK_DIM_RBF) ? 0.03125 : 1.0;
            Ax[idx] = ok ? Axs[(int64_t)d * a_ld + a0 + gi] * pre : 0.0;
            Ac[idx] = ok ? Acn[(int64_t)d * a_ld + a0 + gi] : 0.0;
        }
        if (yA != nullptr && tid < RS) Ay[tid] = (i0 + tid < iend) ? yA[a0 + i0 + tid] : 0.0;
        __syncthreads();

        double e[RT][CPT][R > 0 ? R : 1];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < CPT; ++c)
#pragma unroll
                for (int q = 0; q < (R > 0 ? R : 1); ++q) e[r][c][q] = 0.0;

        if constexpr (R > 0) {
            for (int d = 0; d < D; ++d) {
                double xb[CPT], cb[CPT], xa[RT], ca[RT];
#pragma unroll
                for (int c2 = 0; c2 < CPT / 2; ++c2) {
                    const double2 vx = *reinterpret_cast<const double2*>(&Bx[d * TJ + c2 * 128 + 2 * tx]);
                    const double2 vc = *reinterpret_cast<const double2*>(&Bc[d * TJ + c2 * 128 + 2 * tx]);
                    xb[2 * c2] = vx.x; xb[2 * c2 + 1] = vx.y;
                    cb[2 * c2] = vc.x; cb[2 * c2 + 1] = vc.y;
                }
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    xa[r] = Ax[d * RS + ty * RT + r];
                    ca[r] = Ac[d * RS + ty * RT + r];
                }
                if (ALLRBF || dd.type[d] == OAK_DIM_RBF) {   // ALLRBF: no branch, so the e[] accumulators never change registers
                    const double woff = dd.woff[d], magic = dd.magic[d];
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        double w[CPT], E[CPT];
#pragma unroll
                        for (int c = 0; c < CPT; ++c) {
                            const double u = xa[r] - xb[c];
                            w[c] = fma_clamp01(u, u, woff);
                        }
                        exp2_w_vec<CPT>(w, magic, E, Tab);
#pragma unroll
                        for (int c = 0; c < CPT; ++c) esp_update<R>(e[r][c], __builtin_fma(-ca[r], cb[c], E[c]));
                    }
                } else {
                    const int C = dd.ncat[d];
                    const double* tab = tables + dd.tab_off[d];
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int c = 0; c < CPT; ++c) {
                            const double k = tab[(int)xa[r] * C + (int)xb[c]];
                            esp_update<R>(e[r][c], k);
                        }
                }
            }
        }
        // epilogue: combine orders, optional psi accumulation, store
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int64_t gi = i0 + ty * RT + r;
            double kv[CPT];
#pragma unroll
            for (int c = 0; c < CPT; ++c) kv[c] = esp_combine<R>(e[r][c], dd);
            if (yA != nullptr) {
                const double yv = Ay[ty * RT + r];
#pragma unroll
                for (int c = 0; c < CPT; ++c) psi[c] = __builtin_fma(kv[c], yv, psi[c]);
            }
            if (gi < iend && out != nullptr) {
                double* orow = out + gi * ldo;   // output rows are relative to the chunk start a0
#pragma unroll
                for (int c2 = 0; c2 < CPT / 2; ++c2) {
                    const int64_t gj = jb + c2 * 128 + 2 * tx;
                    const bool al = ((ldo & 1) == 0);
                    if (gj + 1 < nb && al) {
                        *reinterpret_cast<double2*>(orow + gj) = make_double2(kv[2 * c2], kv[2 * c2 + 1]);
                    } else {
                        if (gj < nb) orow[gj] = kv[2 * c2];
                        else if (gj < zero_pad_to) orow[gj] = 0.0;
                        if (gj + 1 < nb) orow[gj + 1] = kv[2 * c2 + 1];
                        else if (gj + 1 < zero_pad_to) orow[gj + 1] = 0.0;
                    }
                }
            }
        }
    }
    if (yA != nullptr) {
        // reduce the 4 waves' psi partials through LDS (fixed order -> deterministic), one row of partials per WG row-block
        __syncthreads();
        double* red = smem;   // reuse: [4][TJ]
#pragma unroll
        for (int c2 = 0; c2 < CPT / 2; ++c2) {
            red[ty * TJ + c2 * 128 + 2 * tx] = psi[2 * c2];
            red[ty * TJ + c2 * 128 + 2 * tx + 1] = psi[2 * c2 + 1];
        }
        __syncthreads();
        for (int j = tid; j < TJ; j += 256) {
            const int64_t gj = jb + j;
            if (gj < nb) psi_part[(int64_t)blockIdx.y * nb + gj] = ((red[j] + red[TJ + j]) + red[2 * TJ + j]) + red[3 * TJ + j];
        }
    }
}

// column sums of a [rows x cols] row-major matrix in fixed order, accumulated onto out
__global__ void __launch_bounds__(256) colsum_accum_kernel(const double* __restrict__ part, int64_t rows, int64_t cols,
                                                           double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cols) return;
    double s = 0.0;
    for (int64_t r = 0; r < rows; ++r) s += part[r * cols + j];
    out[j] += s;
}

template <int R>
__global__ void __launch_bounds__(256) gram_diag_kernel(const DevDesc dd, const double* __restrict__ tables,
                                                        const double* __restrict__ Axs, const double* __restrict__ Acn,
                                                        int64_t a_ld, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double e[R > 0 ? R : 1];
#pragma unroll
    for (int q = 0; q < (R > 0 ? R : 1); ++q) e[q] = 0.0;
    if constexpr (R > 0) {
        for (int d = 0; d < dd.D; ++d) {
            const double x = Axs[(int64_t)d * a_ld + i];
            double k;
            if (dd.type[d] == OAK_DIM_RBF) {
                const double c = Acn[(int64_t)d * a_ld + i];
                k = __builtin_fma(-c, c, dd.bv[d]);     // base K_diag = variance; minus c(x)^2/var_s (ortho_rbf_kernel.py:174-177)
            } else {
                const int C = dd.ncat[d];
                k = tables[dd.tab_off[d] + C * C + (int)x];
            }
            esp_update<R>(e, k);
        }
    }
    out[i] = esp_combine<R>(e, dd);
}

template <int R, int RT, int CPT>
static int launch_gram_t(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B,
                         double* d_out, int64_t ldo, const double* d_yA, double* d_psi, int64_t zero_pad_to) {
    constexpr int TJ = 64 * CPT;
    constexpr int RS = 4 * RT;
    const int D = pk.dd.D;
    size_t lds = sizeof(double) * ((size_t)D * TJ * 2 + (size_t)D * RS * 2 + RS + 64);
    const size_t lds_red = sizeof(double) * 4 * TJ;
    if (lds < lds_red) lds = lds_red;
    if (lds > 160 * 1024) { set_error("gram: LDS request %zu exceeds 160 KiB (D=%d)", lds, D); return OAK_E_ARG; }
    const int64_t nb = B.n;
    const int64_t ncb = (nb + TJ - 1) / TJ;
    // rows per workgroup: enough row-blocks to fill the chip (~8 WGs per CU), at least one row-step
    int64_t target_wg = (int64_t)ctx->num_cu * 8;
    int64_t nrb = (target_wg + ncb - 1) / ncb;
    if (nrb < 1) nrb = 1;
    int64_t rows = (na + nrb - 1) / nrb;
    rows = ((rows + RS - 1) / RS) * RS;
    if (rows < RS) rows = RS;
    if (rows > 4096) rows = 4096;
    nrb = (na + rows - 1) / rows;
    if (nrb > 65535) { rows = ((na + 65534) / 65535 + RS - 1) / RS * RS; nrb = (na + rows - 1) / rows; }
    double* d_part = nullptr;
    if (d_yA != nullptr) OAK_CHECK(get_buf_t(ctx, "psi_part", (size_t)(nrb * nb), &d_part));
    bool all_rbf = true;
    for (int d = 0; d < D; ++d) all_rbf = all_rbf && pk.dd.type[d] == OAK_DIM_RBF;
    auto kern = all_rbf ? gram_kernel<R, RT, CPT, true> : gram_kernel<R, RT, CPT, false>;
    if (lds > 64 * 1024) OAK_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((unsigned)ncb, (unsigned)nrb);
    kern<<<grid, 256, lds, ctx->stream>>>(pk.dd, pk.d_tables, A.xs, A.cn, A.ld, a0, na, B.xs, B.cn, B.ld, nb, d_out, ldo,
                                          (int)rows, d_yA, d_part, zero_pad_to);
    OAK_HIP_CHECK(hipGetLastError());
    if (d_yA != nullptr) {
        colsum_accum_kernel<<<(unsigned)((nb + 255) / 256), 256, 0, ctx->stream>>>(d_part, nrb, nb, d_psi);
        OAK_HIP_CHECK(hipGetLastError());
    }
    return OAK_OK;
}

template <int R>
static int launch_gram_r(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B,
                         double* d_out, int64_t ldo, const double* d_yA, double* d_psi, int64_t zero_pad_to) {
    const int D = pk.dd.D;
    // register blocking: 16 pairs/lane for R<=2, 8 for R<=4, 4 above; halve the column tile when D*TJ*16 B > 64 KiB
    if constexpr (R <= 2) {
        if (D <= 16) return launch_gram_t<R, 4, 4>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        return launch_gram_t<R, 4, 2>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
    } else if constexpr (R <= 4) {
        if (D <= 16) return launch_gram_t<R, 2, 4>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        return launch_gram_t<R, 2, 2>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
    } else {
        if (D <= 16) return launch_gram_t<R, 1, 4>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        return launch_gram_t<R, 1, 2>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
    }
}

int gram(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, double* d_out,
         int64_t ldo, const double* d_yA, double* d_psi, int64_t zero_pad_to) {
    if (na <= 0 || B.n <= 0) return OAK_OK;
    switch (pk.dd.R) {
        case 0: return launch_gram_r<0>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 1: return launch_gram_r<1>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 2: return launch_gram_r<2>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 3: return launch_gram_r<3>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 4: return launch_gram_r<4>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 5: return launch_gram_r<5>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 6: return launch_gram_r<6>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 7: return launch_gram_r<7>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 8: return launch_gram_r<8>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
    }
    set_error("gram: unsupported depth %d", pk.dd.R);
    return OAK_E_ARG;
}

int gram_diag(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, double* d_out, double* d_sum_accum) {
    const int64_t n = A.n;
    if (n <= 0) return OAK_OK;
    const unsigned g = (unsigned)((n + 255) / 256);
#define OAK_DIAG_CASE(RR) case RR: gram_diag_kernel<RR><<<g, 256, 0, ctx->stream>>>(pk.dd, pk.d_tables, A.xs, A.cn, A.ld, n, d_out); break;
    switch (pk.dd.R) {
        OAK_DIAG_CASE(0) OAK_DIAG_CASE(1) OAK_DIAG_CASE(2) OAK_DIAG_CASE(3) OAK_DIAG_CASE(4)
        OAK_DIAG_CASE(5) OAK_DIAG_CASE(6) OAK_DIAG_CASE(7) OAK_DIAG_CASE(8)
        default: set_error("gram_diag: unsupported depth %d", pk.dd.R); return OAK_E_ARG;
    }
#undef OAK_DIAG_CASE
    OAK_HIP_CHECK(hipGetLastError());
    if (d_sum_accum != nullptr) {
        double* d_s = nullptr;
        OAK_CHECK(get_buf_t(ctx, "diag_sum_tmp", 1, &d_s));
        OAK_CHECK(reduce_sum(ctx, d_out, n, d_s, 0, 1));
        OAK_CHECK(axpy(ctx, 1.0, d_s, d_sum_accum, 1));
    }
    return OAK_OK;
}

}  // namespace oak

// Analytic gradient of the SGPR bound / GPR log-marginal w.r.t. the (constrained) kernel and noise parameters.
// The reference obtains these from TensorFlow reverse-mode autodiff through SGPR.elbo (oak/model_utils.py:168-173);
// here they are closed-form adjoints:
//   F(Kuu, Phi, psi, kappa, s2) -> G_uu = dF/dKuu, H/(2 s2) = dF/dPhi, a/s2 = dF/dpsi, -1/(2 s2) = dF/dkappa, dF/ds2,
//   Gfu = dF/dKfu = Kfu H / s2 + y a^T / s2                                  (one N x M x M GEMM, fp64 MFMA)
//   dF/dtheta = <Gfu, dKfu/dtheta> + <G_uu, dKuu/dtheta> - 1/(2 s2) sum_n dKdiag_n/dtheta   (fused pair kernels below)
// Lengthscale derivatives are formed in units of 2 ln2 / l_d (featurize pre-divides dcn by it; scatter_record multiplies
// the finished sum back), so the pair kernels need no per-dimension derivative constant.
// The pair kernel recomputes each base-kernel value and contracts G with dK/dk_d * dk_d/dtheta, where
// dK/dk_d = sum_r w_r e_{r-1}^{(-d)} uses leave-one-out elementary symmetric polynomials (e^{(-d)}_q = e_q - k_d e^{(-d)}_{q-1}).
#include "oak_internal.h"
#include "exp2w.h"
#include <cmath>
#include <cstdlib>

namespace oak {

__constant__ double c_exp2_table_g[64] = {
    1.0, 1.0108892860517004600, 1.0218971486541166782, 1.0330248790212284225,
    1.0442737824274138403, 1.0556451783605571588, 1.0671404006768236182, 1.0787607977571197937,
    1.0905077326652576592, 1.1023825833078409436, 1.1143867425958925363, 1.1265216186082418997,
    1.1387886347566916537, 1.1511892299529827082, 1.1637248587775775138, 1.1763969916502812763,
    1.1892071150027210667, 1.2021567314527031420, 1.2152473599804688781, 1.2284805361068700056,
    1.2418578120734840486, 1.2553807570246910895, 1.2690509571917332225, 1.2828700160787782807,
    1.2968395546510096659, 1.3109612115247643419, 1.3252366431597412946, 1.3396675240533030053,
    1.3542555469368927283, 1.3690024229745906119, 1.3839098819638319549, 1.3989796725383111402,
    1.4142135623730950488, 1.4296133383919700113, 1.4451808069770466200, 1.4609177941806469887,
    1.4768261459394993114, 1.4929077282912648492, 1.5091644275934227398, 1.5255981507445383069,
    1.5422108254079408236, 1.5590044002378369670, 1.5759808451078864865, 1.5931421513422668980,
    1.6104903319492543082, 1.6280274218573478129, 1.6457554781539648445, 1.6636765803267364350,
    1.6817928305074290861, 1.7001063537185234695, 1.7186192981224779156, 1.7373338352737062489,
    1.7562521603732994832, 1.7753764925265212526, 1.7947090750031071864, 1.8142521755003987562,
    1.8340080864093424635, 1.8539791250833855684, 1.8741676341102999014, 1.8945759815869656413,
    1.9152065613971472939, 1.9360617934922944505, 1.9571441241754002690, 1.9784560263879509682
};

// scalar variant of the Gram kernel's exp2 (same table + degree-5 polynomial)
__device__ __forceinline__ double exp2_neg_tab(double t, const double* __restrict__ tab) {
    constexpr double c1 = 6.931471805599453094e-01, c2 = 2.402265069591007123e-01, c3 = 5.550410866482157995e-02,
                     c4 = 9.618129107628477162e-03, c5 = 1.333355814642844342e-03;
    constexpr double MAGIC = 105553116266496.0;
    t = __builtin_fmax(t, -1020.0);
    const double a = t + MAGIC;
    const int ki = __double2loint(a);
    const double tv = tab[ki & 63];
    const double r = t - (a - MAGIC);
    double p = __builtin_fma(c5, r, c4);
    p = __builtin_fma(p, r, c3);
    p = __builtin_fma(p, r, c2);
    p = __builtin_fma(p, r, c1);
    p = __builtin_fma(p, r, 1.0);
    const int hi = __double2hiint(tv) + (ki >> 6) * 1048576;
    return __hiloint2double(hi, __double2loint(tv)) * p;
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// layout of one partial / output record: [gl (D) | gk (D) | gw (R+1) | gtab (tablen)]
//   gl[d] = sum G dK/dk_d dk_d/dl_d ;  gk[d] = sum G dK/dk_d k_d  (-> d/d base_var_d after division by base_var_d)
//   gw[r] = sum G e_r ; gtab[off + ia*C + ib] = sum G dK/dk_d  over pairs with categories (ia, ib)  (x base_var later)
template <int R, int CPT>
__global__ void __launch_bounds__(256)
gram_bwd_kernel(const DevDesc dd, const double* __restrict__ tables, int tablen,
                const double* __restrict__ Axs, const double* __restrict__ Acn, const double* __restrict__ Adcn, int64_t a_ld,
                int64_t a0, int64_t na, const double* __restrict__ Bxs, const double* __restrict__ Bcn,
                const double* __restrict__ Bdcn, int64_t b_ld, int64_t nb, const double* __restrict__ G, int64_t ldg,
                const double* __restrict__ yA, const double* __restrict__ avec, double g_scale, int rows_per_wg,
                double* __restrict__ partial, const double* __restrict__ Axx, const double* __restrict__ Bxx, int nx) {
    constexpr int TJ = 64 * CPT;
    constexpr int RT = 2, RS = 4 * RT;
    constexpr int RR = R > 0 ? R : 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int D = dd.D;
    double* Bx = smem;                  // [D][TJ]
    double* Bc = Bx + D * TJ;
    double* Bd = Bc + D * TJ;
    double* Ax = Bd + D * TJ;           // [D][RS]
    double* Ac = Ax + D * RS;
    double* Ad = Ac + D * RS;
    double* Ay = Ad + D * RS;           // [RS]
    double* Av = Ay + RS;               // [TJ]  a_m / s2 slice (rank-1 term)
    double* Tab = Av + TJ;              // [64]
    double* accL = Tab + 64;            // [4][D]
    double* accK = accL + 4 * D;        // [4][D]
    double* accT = accK + 4 * D;        // [4][tablen]: one copy per wave (see accTw)
    double* Bq = accT + 4 * tablen;     // [nx][TJ]  further columns of grouped RBF dims (DevDesc::xrow / nxc, Feat::xx)
    double* Aq = Bq + nx * TJ;          // [nx][RS]
    const int tid = threadIdx.x, tx = tid & 63;
    const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t jb = (int64_t)blockIdx.x * TJ;
    const int64_t ib = (int64_t)blockIdx.y * rows_per_wg;
    const int64_t iend = (ib + rows_per_wg < na) ? ib + rows_per_wg : na;
    for (int idx = tid; idx < D * TJ; idx += 256) {
        const int d = idx / TJ, j = idx - d * TJ;
        const int64_t gj = jb + j;
        const bool ok = gj < nb;
        Bx[idx] = ok ? Bxs[(int64_t)d * b_ld + gj] : 0.0;
        Bc[idx] = ok ? Bcn[(int64_t)d * b_ld + gj] : 0.0;
        Bd[idx] = ok ? Bdcn[(int64_t)d * b_ld + gj] : 0.0;
    }
    for (int idx = tid; idx < nx * TJ; idx += 256) {
        const int q = idx / TJ, j = idx - q * TJ;
        Bq[idx] = (jb + j < nb) ? Bxx[(int64_t)q * b_ld + jb + j] : 0.0;
    }
    for (int j = tid; j < TJ; j += 256) Av[j] = (avec != nullptr && jb + j < nb) ? avec[jb + j] : 0.0;
    if (tid < 64) Tab[tid] = c_exp2_table_g[tid];
    for (int idx = tid; idx < 8 * D; idx += 256) accL[idx] = 0.0;     // accL and accK are contiguous
    for (int idx = tid; idx < 4 * tablen; idx += 256) accT[idx] = 0.0;
    // Categorical-table sums go through LDS atomics.  Each wave adds into ITS OWN copy: within one ds_add instruction the LDS
    // unit serialises colliding lanes in a fixed order, and a wave's instructions are ordered, so a copy's contents do not
    // depend on timing; what used to vary from run to run was the interleaving of the four waves on one shared copy.  The
    // copies are combined in a fixed order at the end (tests/test_gpu_grad.py::test_mixed_kernel_gradient_is_bitwise_repeatable).
    double* accTw = accT + ty * tablen;
    double gw[R + 1];
#pragma unroll
    for (int q = 0; q <= R; ++q) gw[q] = 0.0;
    const int lane0 = (tx == 0);

    for (int64_t i0 = ib; i0 < iend; i0 += RS) {
        __syncthreads();
        for (int idx = tid; idx < D * RS; idx += 256) {
            const int d = idx / RS, r = idx - d * RS;
            const int64_t gi = i0 + r;
            const bool ok = gi < iend;
            Ax[idx] = ok ? Axs[(int64_t)d * a_ld + a0 + gi] : 0.0;
            Ac[idx] = ok ? Acn[(int64_t)d * a_ld + a0 + gi] : 0.0;
            Ad[idx] = ok ? Adcn[(int64_t)d * a_ld + a0 + gi] : 0.0;
        }
        if (tid < RS) Ay[tid] = (yA != nullptr && i0 + tid < iend) ? yA[a0 + i0 + tid] : 0.0;
        for (int idx = tid; idx < nx * RS; idx += 256) {
            const int q = idx / RS, r = idx - q * RS;
            Aq[idx] = (i0 + r < iend) ? Axx[(int64_t)q * a_ld + a0 + i0 + r] : 0.0;
        }
        __syncthreads();
        // adjoint of K for this lane's RT x CPT pairs
        double g[RT][CPT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const int64_t gi = i0 + ty * RT + r, gj = jb + CPT * tx + c;
                double v = 0.0;
                if (gi < iend && gj < nb) v = g_scale * G[gi * ldg + gj] + Ay[ty * RT + r] * Av[CPT * tx + c];
                g[r][c] = v;
            }
        // pass 1: elementary symmetric polynomials
        double e[RT][CPT][RR];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < CPT; ++c)
#pragma unroll
                for (int q = 0; q < RR; ++q) e[r][c][q] = 0.0;
        if constexpr (R > 0) {
            for (int d = 0; d < D; ++d) {
                const bool rbf = dd.type[d] == OAK_DIM_RBF;
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CPT; ++c) {
                        const double xa = Ax[d * RS + ty * RT + r], xb = Bx[d * TJ + CPT * tx + c];
                        double k;
                        if (rbf) {
                            const double u = xa - xb;
                            double t = __builtin_fma(-u, u, dd.log2bv[d]);
                            for (int q = dd.xrow[d]; q < dd.xrow[d] + dd.nxc[d]; ++q) {      // grouped dim: the other columns' share of the exponent
                                const double uq = Aq[q * RS + ty * RT + r] - Bq[q * TJ + CPT * tx + c];
                                t = __builtin_fma(-uq, uq, t);
                            }
                            const double E = exp2_neg_tab(t, Tab);
                            k = __builtin_fma(-Ac[d * RS + ty * RT + r], Bc[d * TJ + CPT * tx + c], E);
                        } else {
                            k = tables[dd.tab_off[d] + (int)xa * dd.ncat[d] + (int)xb];
                        }
#pragma unroll
                        for (int q = R - 1; q >= 1; --q) e[r][c][q] = __builtin_fma(k, e[r][c][q - 1], e[r][c][q]);
                        e[r][c][0] += k;
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                gw[0] += g[r][c];
#pragma unroll
                for (int q = 1; q <= R; ++q) gw[q] = __builtin_fma(g[r][c], e[r][c][q - 1], gw[q]);
            }
        // pass 2: leave-one-out coefficients and parameter contractions
        if constexpr (R > 0) {
            for (int d = 0; d < D; ++d) {
                const bool rbf = dd.type[d] == OAK_DIM_RBF;
                double cl = 0.0, ck = 0.0;
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CPT; ++c) {
                        const double xa = Ax[d * RS + ty * RT + r], xb = Bx[d * TJ + CPT * tx + c];
                        double k, dkl = 0.0;
                        int tidx = 0;
                        if (rbf) {
                            const double u = xa - xb;
                            double u2 = u * u;
                            for (int q = dd.xrow[d]; q < dd.xrow[d] + dd.nxc[d]; ++q) {
                                const double uq = Aq[q * RS + ty * RT + r] - Bq[q * TJ + CPT * tx + c];
                                u2 = __builtin_fma(uq, uq, u2);
                            }
                            const double E = exp2_neg_tab(dd.log2bv[d] - u2, Tab);
                            const double ca = Ac[d * RS + ty * RT + r], cb = Bc[d * TJ + CPT * tx + c];
                            k = __builtin_fma(-ca, cb, E);
                            dkl = E * u2 - (Ad[d * RS + ty * RT + r] * cb + ca * Bd[d * TJ + CPT * tx + c]);   // in units of 2 ln2 / l
                        } else {
                            tidx = dd.tab_off[d] + (int)xa * dd.ncat[d] + (int)xb;
                            k = tables[tidx];
                        }
                        double f = 1.0, coef = dd.w[1];
#pragma unroll
                        for (int q = 1; q < R; ++q) {
                            f = __builtin_fma(-k, f, e[r][c][q - 1]);
                            coef = __builtin_fma(dd.w[q + 1], f, coef);
                        }
                        const double gc = g[r][c] * coef;
                        cl = __builtin_fma(gc, dkl, cl);
                        ck = __builtin_fma(gc, k, ck);
                        if (!rbf && dd.type[d] == OAK_DIM_CATEGORICAL && gc != 0.0) atomicAdd(&accTw[tidx], gc);
                    }
                cl = wave_sum(cl);
                ck = wave_sum(ck);
                if (lane0) { accL[ty * D + d] += cl; accK[ty * D + d] += ck; }
            }
        }
    }
    // workgroup record
#pragma unroll
    for (int q = 0; q <= R; ++q) gw[q] = wave_sum(gw[q]);
    __syncthreads();
    double* red = Ax;   // reuse [4][R+1]
    if (lane0) {
#pragma unroll
        for (int q = 0; q <= R; ++q) red[ty * (R + 1) + q] = gw[q];
    }
    __syncthreads();
    // the record is laid out for the kernel's ACTUAL depth dd.R <= R (depths 9..16 run the R = 12 / 16 instantiations with
    // zero weights above dd.R; the unused accumulators are simply not written)
    const int RA = dd.R;
    const int64_t reclen = 2 * D + (RA + 1) + tablen;
    double* rec = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * reclen;
    for (int idx = tid; idx < D; idx += 256) {
        rec[idx] = ((accL[idx] + accL[D + idx]) + accL[2 * D + idx]) + accL[3 * D + idx];
        rec[D + idx] = ((accK[idx] + accK[D + idx]) + accK[2 * D + idx]) + accK[3 * D + idx];
    }
    if (tid <= RA) rec[2 * D + tid] = ((red[tid] + red[(R + 1) + tid]) + red[2 * (R + 1) + tid]) + red[3 * (R + 1) + tid];
    for (int idx = tid; idx < tablen; idx += 256)
        rec[2 * D + (RA + 1) + idx] = ((accT[idx] + accT[tablen + idx]) + accT[2 * tablen + idx]) + accT[3 * tablen + idx];
}

// Row-major pack of the backward features of rows a0 .. a0+na-1:  out[i][q][d],  q = (xs32, cn, dcs),  d < DP;
// padding dimensions d >= D hold (-1, 0, 0) so that against the column-side padding (+1) the pair clamps to E = 2^-1024.
// split = 2 / 4 (the lane-pair / lane-quad form of the fast kernel, DP = 32 / 64): lane h of a group walks dimensions split*step + h; its features are laid out
// [h][chunk of four steps][q][step in chunk], so that a chunk's 12 values are contiguous (wide vector loads).
__global__ void __launch_bounds__(256) pack_rows_kernel(const double* __restrict__ xs32, const double* __restrict__ cn,
                                                        const double* __restrict__ dcs, int64_t ld, int64_t a0, int64_t na, int D,
                                                        int DP, double* __restrict__ out, int split, double xpad) {
    extern __shared__ double tile_mem[];                          // [3 * DP][65]: 3 * DP <= 192 (dynamic: 100 KB at DP = 64)
    auto tile = [&](int j, int r) -> double& { return tile_mem[j * 65 + r]; };
    const int64_t i0 = (int64_t)blockIdx.x * 64;
    const int W = 3 * DP;
    for (int idx = threadIdx.x; idx < W * 64; idx += 256) {       // coalesced over rows
        const int j = idx >> 6, r = idx & 63;
        const int q = j / DP, d = j - q * DP;
        const int64_t i = i0 + r;
        double v = (q == 0) ? xpad : 0.0;              // padding dims: -1 on the row side, +1 on the column side (w clamps to 1)
        if (d < D && i < na) v = (q == 0 ? xs32 : (q == 1 ? cn : dcs))[(int64_t)d * ld + a0 + i];
        tile(j, r) = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < W * 64; idx += 256) {       // coalesced over the packed row
        const int r = idx / W, j = idx - r * W;
        int src = j;
        if (split < 0) {                                          // j = (c * 3 + q) * 4 + v  <-  tile row q * DP + 4 c + v   (rows-in-lanes kernel's column side)
            const int v = j & 3, q = (j >> 2) % 3, c = (j >> 2) / 3;
            src = q * DP + 4 * c + v;
        } else if (split >= 2) {                                  // j = ((h * nch + c) * 3 + q) * 4 + v  <-  tile row q * DP + d
            const int v = j & 3, q = (j >> 2) % 3, hc = (j >> 2) / 3, nch = DP / (4 * split);
            const int h = hc / nch, c = hc - h * nch;
            src = q * DP + split * (4 * c + v) + h;
        }
        if (i0 + r < na) out[(i0 + r) * W + j] = tile(src, r);
    }
}
static int launch_pack_rows(oak_ctx* ctx, const Feat& A, int64_t a0, int64_t na, int D, int DP, double* d_pack, int split, double xpad = -1.0) {
    const size_t lds = sizeof(double) * (size_t)3 * DP * 65;
    if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)pack_rows_kernel));
    pack_rows_kernel<<<(unsigned)((na + 63) / 64), 256, lds, ctx->stream>>>(A.xs32, A.cn, A.dcs, A.ld, a0, na, D, DP, d_pack, split, xpad);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// Fast path (1 <= R <= 4, D <= DMAX <= 16): one exp2 per pair per dimension.  Each lane walks its pairs one at a time,
// keeps k_d and dk_d/dl_d of all dimensions in registers (vectorised 4 dimensions at a time for ILP), and accumulates the
// per-dimension contractions in registers across the whole row range; one workgroup reduction at the very end.
// exp2 in the clamped-w form of exp2w.h: features are staged as x*scale/32, the lengthscale-derivative features dcn as
// dcn/1024, so u^2 in the derivative is (w - woff) and the accumulated dF/dl comes out divided by 1024 (undone, exactly,
// when the record is written).  Padding dimensions d >= D are staged as xa = -1, xb = +1 (w clamps to 1, E = 2^-1024).
// UNITBV: every RBF dimension has base variance exactly 1 (OAK's share_var_across_orders default): woff = 0 and
// magic = EW_MAGIC are compile-time constants, no per-dimension constant is read at all.
// SPLIT = 2 (17..32 dimensions): a pair is shared by two ADJACENT LANES, each walking DMAX = 16 of the DT = 32 staged dimensions
// with the register footprint of the 16-dimension kernel (two waves per SIMD, no AGPR spill traffic; the 32-registers-per-array
// form ran one wave per SIMD at 38 instructions per pair-dimension).  The elementary symmetric polynomials of the two halves
// are exchanged once per pair (R values, adjacent-lane swap) and convolved, e_r = sum_{i+j=r} e_i(A) e_j(B); the leave-one-out
// identity behind the Horner coefficients holds for the TOTAL polynomials, so from there on each lane runs the unchanged
// per-dimension code on its own half.  Lane h of a pair takes the staged dimensions 2*step + h.  Row features are lane-dependent
// then: vector loads (L1-resident) from rows packed [half][chunk of four steps][feature][step] by pack_rows_kernel(split = 2).
template <int R, int DMAX, int CPT, bool ALLRBF, bool WANT_GK, bool UNITBV, int SPLIT = 1>
__global__ void __launch_bounds__(256, ((DMAX <= 16 && !(SPLIT >= 2 && R > 4) && R <= 8 && (SPLIT < 4 || (ALLRBF && UNITBV && !WANT_GK))) ? 2 : 1))      // <= 16 dims per lane: hold the register budget at two waves per SIMD
                                                                            // (lane pairs at depth 5..8: one wave, 370 registers -- at 256 it spilled 452 B: 65 vs 25 ms)
gram_bwd_fast_kernel(const DevDesc dd, const double* __restrict__ tables, int tablen,
                     const double* __restrict__ Apack, int64_t a0, int64_t na, const double* __restrict__ Bxs,
                     const double* __restrict__ Bcn, const double* __restrict__ Bdcn, int64_t b_ld, int64_t nb,
                     const double* __restrict__ G, int64_t ldg, const double* __restrict__ yA, const double* __restrict__ avec,
                     double g_scale, int rows_per_wg, double* __restrict__ partial) {
    // Apack: row-major [na][3][DT] = (xs32 | cn | dcs) of rows a0.. (pack_rows_kernel; padding dims hold -1, 0, 0); SPLIT = 2: the
    // same 3 * DT values per row in the [half][chunk][feature][step] order.
    // Bxs is the PRE-SCALED array Feat::xs32, Bdcn is Feat::dcs.
    constexpr int DT = DMAX * SPLIT;                        // staged dimensions
    constexpr int CW = 64 / SPLIT;                          // columns a wave covers per pass
    constexpr int TJ = CW * CPT, RT = 2, RS = 4 * RT;       // LDS: 3 * DT * TJ doubles
    constexpr int NGK = WANT_GK ? DMAX : 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int D = dd.D;
    double* Bx = smem;                  // [DT][TJ]
    double* Bc = Bx + DT * TJ;
    double* Bd = Bc + DT * TJ;
    double* Av = Bd + DT * TJ;          // [TJ]
    double* Tab = Av + TJ;              // [EW_N] biased exp2 table
    double* Cw = Tab + EW_N;            // [DT] woff per dim    (not allocated when UNITBV)
    double* Cm = Cw + DT;               // [DT] magic per dim
    double* accT = UNITBV ? Tab + EW_N : Cm + DT;     // [4][tablen]: one copy per wave (deterministic sums, see gram_bwd_kernel)
    double* Tbl = accT + 4 * tablen;    // [tablen] the discrete dimensions' kernel tables: a pair walks one lane at a time with one
                                        // wave per SIMD at 32 dims, so a look-up's latency is fully exposed (LDS ~100 cycles, global ~800)
    int* meta = reinterpret_cast<int*>(Tbl + tablen);        // [DT] tab_off | ncat << 16 of the discrete dims: read from LDS
                                        // inside the rare branch instead of living in ~100 SGPRs (they spilled to VGPR lanes)
    double* red = Bx;                   // [4][2*DT + R + 1], aliases the column features once the row loop is done
                                        // (54 272 B at DMAX = 16 without discrete tables: three workgroups per CU)
    const int tid = threadIdx.x, tx = tid & 63;
    const int cl = tx / SPLIT;                              // column of the wave's pass this lane works on
    const int half = tx & (SPLIT - 1);                      // which of the group's SPLIT lanes this is (SPLIT = 4, r04: 33..64 sub-kernels)
    // lane (half, step d) works on staged dimension SPLIT * d + half: with the usual ordering of a mixed kernel (continuous
    // columns first) both lanes of a pair then meet the same kind of dimension at most steps, and whole chunks of four steps
    // are all-RBF or all-discrete -- decided by wave-uniform tests on the dimension mask, not per lane
#define OAK_SD(d) (SPLIT * (d) + half)
    const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t jb = (int64_t)blockIdx.x * TJ;
    const int64_t ib = (int64_t)blockIdx.y * rows_per_wg;
    const int64_t iend = (ib + rows_per_wg < na) ? ib + rows_per_wg : na;
    for (int idx = tid; idx < DT * TJ; idx += 256) {
        const int d = idx / TJ, j = idx - d * TJ;
        const int64_t gj = jb + j;
        const bool ok = gj < nb && d < D;
        Bx[idx] = ok ? Bxs[(int64_t)d * b_ld + gj] : (d < D ? 0.0 : 1.0);
        Bc[idx] = ok ? Bcn[(int64_t)d * b_ld + gj] : 0.0;
        Bd[idx] = ok ? Bdcn[(int64_t)d * b_ld + gj] : 0.0;
    }
    for (int j = tid; j < TJ; j += 256) Av[j] = (avec != nullptr && jb + j < nb) ? avec[jb + j] : 0.0;
    for (int j = tid; j < EW_N; j += 256) Tab[j] = biased_table_entry(j);
    if constexpr (!UNITBV) {
        if (tid < DT) {
            const bool rbf = tid < D && dd.type[tid] == OAK_DIM_RBF;
            Cw[tid] = rbf ? dd.woff[tid] : 0.0;
            Cm[tid] = rbf ? dd.magic[tid] : EW_MAGIC;
        }
    }
    for (int idx = tid; idx < 4 * tablen; idx += 256) accT[idx] = 0.0;
    for (int idx = tid; idx < tablen; idx += 256) Tbl[idx] = tables[idx];
    double* accTw = accT + ty * tablen;
    unsigned long long rbf_mask = ~0ull, cat_mask = 0ull;
    if constexpr (!ALLRBF) {
        if (tid < DT) meta[tid] = tid < D ? (dd.tab_off[tid] | (dd.ncat[tid] << 16)) : 0;      // tablen <= 1024
        for (int d = 0; d < D; ++d) {
            if (dd.type[d] != OAK_DIM_RBF) rbf_mask &= ~(1ull << d);
            if (dd.type[d] == OAK_DIM_CATEGORICAL) cat_mask |= 1ull << d;
        }
    }
    unsigned my_rbf = 0u, my_cat = 0u;          // bit d: this lane's dimension at step d (lane-dependent when SPLIT = 2)
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
        my_rbf |= (unsigned)((rbf_mask >> OAK_SD(d)) & 1ull) << d;
        my_cat |= (unsigned)((cat_mask >> OAK_SD(d)) & 1ull) << d;
    }
    double gl[DMAX], gk[NGK], gw[R + 1];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) gl[d] = 0.0;
#pragma unroll
    for (int d = 0; d < NGK; ++d) gk[d] = 0.0;
#pragma unroll
    for (int q = 0; q <= R; ++q) gw[q] = 0.0;
    __syncthreads();

    // The row-side features of a pair are wave-uniform (a wave owns whole rows): they are read straight from the packed
    // row-major array with uniform addresses, i.e. as wide SCALAR loads into SGPRs, and never touch LDS.  The kernel is co-limited by
    // VALU issue and LDS bandwidth; this halves its LDS traffic (3 column-side reads per pair-dimension instead of 6)
    // and removes the per-row-step staging barriers.
    struct Chunk { double xa[4], xb[4], ca[4], cb[4], ad[4], bd[4], cw[4], cm[4]; };
    // The adjoint G of a pair is loaded one pair AHEAD (clamped address, no branch), so its HBM latency hides under the
    // ~400 DP instructions of the pair in flight instead of stalling every pair.
    auto g_addr = [&](int64_t i0n, int prn) -> const double* {
        const int64_t gin = i0n + ty * RT + prn / CPT, gjn = jb + cl + CW * (prn % CPT);
        return G + (gin < iend ? gin : iend - 1) * ldg + (gjn < nb ? gjn : nb - 1);
    };
    double graw_next = ib < iend ? *g_addr(ib, 0) : 0.0;
    auto fetch = [&](const double* __restrict__ prow, int col, int d0, Chunk& ch) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int d = d0 + v;
            if constexpr (SPLIT >= 2) {        // [half][chunk][q][v]: 12 contiguous values per chunk (prow already points at this half)
                ch.xa[v] = prow[3 * d0 + v]; ch.ca[v] = prow[3 * d0 + 4 + v]; ch.ad[v] = prow[3 * d0 + 8 + v];
            } else {
                ch.xa[v] = prow[d]; ch.ca[v] = prow[DT + d]; ch.ad[v] = prow[2 * DT + d];
            }
            ch.xb[v] = Bx[OAK_SD(d) * TJ + col]; ch.cb[v] = Bc[OAK_SD(d) * TJ + col]; ch.bd[v] = Bd[OAK_SD(d) * TJ + col];
            if constexpr (!UNITBV) { ch.cw[v] = Cw[OAK_SD(d)]; ch.cm[v] = Cm[OAK_SD(d)]; }
        }
    };
    auto row_ptr = [&](int64_t i0n, int prn) -> const double* {
        const int64_t gin = i0n + ty * RT + prn / CPT;
        return Apack + (gin < iend ? gin : iend - 1) * (3 * DT) + half * (3 * DMAX);
    };
    // The first four dimensions' features of a pair are fetched while the PREVIOUS pair is in its second phase (polynomial
    // coefficients, Horner, accumulation: no feature is live there), so a pair does not start on an exposed load.
    Chunk cur;
    if (ib < iend) fetch(row_ptr(ib, 0), cl, 0, cur);
    for (int64_t i0 = ib; i0 < iend; i0 += RS) {
#pragma unroll 1
        for (int pr = 0; pr < RT * CPT; ++pr) {      // one pair at a time: only one set of k[], dk[] is live
            const int r = pr / CPT, c = pr % CPT;
            const int col = cl + CW * c;                          // lanes own adjacent columns: conflict-free LDS reads
            const int64_t gi = i0 + ty * RT + r, gj = jb + col;
            const int64_t gr = gi < iend ? gi : iend - 1;         // uniform; rows past the end contribute g = 0
            const double* __restrict__ prow = Apack + gr * (3 * DT) + half * (3 * DMAX);
            const double yrow = yA != nullptr ? yA[a0 + gr] : 0.0;
            const double g = (gi < iend && gj < nb) ? __builtin_fma(g_scale, graw_next, yrow * Av[col]) : 0.0;
            graw_next = (pr + 1 < RT * CPT) ? *g_addr(i0, pr + 1) : *g_addr(i0 + RS, 0);
            double k[DMAX], dk[DMAX];
            Chunk nxt;
#pragma unroll
            for (int d0 = 0; d0 < DMAX; d0 += 4) {
                if (d0 + 4 < DMAX) fetch(prow, col, d0 + 4, nxt);      // software prefetch of the next 4 dimensions' features
                asm volatile("" ::: "memory");              // keep later chunks' loads below this point: bounds the live SGPRs
                // the staged dimensions SPLIT*d0 .. SPLIT*(d0+4)-1 of this chunk: wave-uniform view of their types
                constexpr unsigned CM = (1u << (4 * SPLIT)) - 1u;
                const unsigned cbits = ALLRBF ? CM : (unsigned)((rbf_mask >> (SPLIT * d0)) & CM);
                double kv4[4] = {0.0, 0.0, 0.0, 0.0}, dv4[4] = {0.0, 0.0, 0.0, 0.0};
                if (ALLRBF || cbits != 0u) {                // some lane has an RBF dimension here: the exp2 path (all lanes)
                    double w[4], u2[4], mg[4], E[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const double u = cur.xa[v] - cur.xb[v];
                        if constexpr (UNITBV) {
                            w[v] = fma_clamp01(u, u, 0.0); u2[v] = w[v]; mg[v] = EW_MAGIC;
                        } else {
                            w[v] = fma_clamp01(u, u, cur.cw[v]); u2[v] = w[v] - cur.cw[v]; mg[v] = cur.cm[v];
                        }
                    }
                    exp2_w_vec<4>(w, mg, E, Tab);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        kv4[v] = __builtin_fma(-cur.ca[v], cur.cb[v], E[v]);
                        dv4[v] = __builtin_fma(E[v], u2[v], -__builtin_fma(cur.ad[v], cur.cb[v], cur.ca[v] * cur.bd[v]));
                    }
                }
                if constexpr (!ALLRBF) {
                    if (cbits != CM) {                      // some lane has a discrete dimension here: table value, no lengthscale
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int d = d0 + v;
                            const bool disc = !((my_rbf >> d) & 1u);
                            const int mt = meta[OAK_SD(d)];
                            const int idx = disc ? (mt & 0xffff) + (int)cur.xa[v] * (mt >> 16) + (int)cur.xb[v] : 0;
                            const double tv = Tbl[idx];
                            kv4[v] = disc ? tv : kv4[v];
                            dv4[v] = disc ? 0.0 : dv4[v];
                        }
                    }
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) { k[d0 + v] = kv4[v]; dk[d0 + v] = dv4[v]; }
                if (d0 + 4 < DMAX) cur = nxt;
            }
            {   // next pair's first chunk (clamped addresses past the end: loaded, never used)
                const bool same = pr + 1 < RT * CPT;
                const int prn = same ? pr + 1 : 0;
                fetch(row_ptr(same ? i0 : i0 + RS, prn), cl + CW * (prn % CPT), 0, cur);
            }
            double e[R];
#pragma unroll
            for (int q = 0; q < R; ++q) e[q] = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
#pragma unroll
                for (int q = R - 1; q >= 1; --q) e[q] = __builtin_fma(k[d], e[q - 1], e[q]);
                e[0] += k[d];
            }
            if constexpr (SPLIT >= 2) {
                // e holds this lane's polynomials e_1..e_R; the partner lane holds another part's.  Total: e_r = sum_{i+j=r} e_i e'_j
                // (lane quads: pairs first, then the pairs of pairs -- the product of the parts' generating polynomials either way)
#pragma unroll
                for (int stage = 1; stage < SPLIT; stage <<= 1) {
                    double eo[R], et[R];
#pragma unroll
                    for (int q = 0; q < R; ++q) eo[q] = __shfl_xor(e[q], stage, 64);
#pragma unroll
                    for (int r = 1; r <= R; ++r) {
                        double t = e[r - 1] + eo[r - 1];
#pragma unroll
                        for (int i = 1; i < r; ++i) t = __builtin_fma(e[i - 1], eo[r - i - 1], t);
                        et[r - 1] = t;
                    }
#pragma unroll
                    for (int q = 0; q < R; ++q) e[q] = et[q];
                }
            }
            const double gq = (SPLIT >= 2 && half != 0) ? 0.0 : g;      // the order-variance sums count a pair once
            gw[0] += gq;
#pragma unroll
            for (int q = 1; q <= R; ++q) gw[q] = __builtin_fma(gq, e[q - 1], gw[q]);
            // dK/dk_d = sum_q w_{q+1} e_q^{(-d)} with the leave-one-out polynomials e_q^{(-d)} = sum_{i<=q} (-k_d)^i e_{q-i}: a
            // polynomial of degree R-1 in k_d whose coefficients belong to the PAIR.  They are formed once per pair (with g and
            // the signs folded in), leaving R-1 FMAs per dimension (Horner) instead of the 2(R-1)+1 of the recurrence.
            double cg[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double ci = dd.w[i + 1];                                      // q = i term: e_0 = 1
#pragma unroll
                for (int q = i + 1; q < R; ++q) ci = __builtin_fma(dd.w[q + 1], e[q - i - 1], ci);
                cg[i] = (i & 1) ? -(g * ci) : g * ci;
            }
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                double gc = cg[R - 1];
#pragma unroll
                for (int i = R - 2; i >= 0; --i) gc = __builtin_fma(gc, k[d], cg[i]);
                gl[d] = __builtin_fma(gc, dk[d], gl[d]);
                if constexpr (WANT_GK) gk[d] = __builtin_fma(gc, k[d], gk[d]);
                if constexpr (!ALLRBF) {
                    if ((cat_mask >> (SPLIT * d)) & ((1ull << SPLIT) - 1ull)) {       // wave-uniform: a categorical dimension at this step
                        if (((my_cat >> d) & 1u) && gc != 0.0) {
                            const int mt = meta[OAK_SD(d)];
                            atomicAdd(&accTw[(mt & 0xffff) + (int)prow[SPLIT >= 2 ? 3 * (d & ~3) + (d & 3) : d] * (mt >> 16) + (int)Bx[OAK_SD(d) * TJ + col]], gc);
                        }
                    }
                }
            }
        }
    }
    // workgroup reduction of the register accumulators
    constexpr int NACC = 2 * DT + R + 1;
    auto class_sum = [](double v) {          // over the lanes of this lane's half (all lanes when SPLIT = 1)
        for (int o = 32; o >= SPLIT; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
#pragma unroll
    for (int d = 0; d < DMAX; ++d) gl[d] = class_sum(gl[d]);
#pragma unroll
    for (int d = 0; d < NGK; ++d) gk[d] = class_sum(gk[d]);
#pragma unroll
    for (int q = 0; q <= R; ++q) gw[q] = class_sum(gw[q]);
    __syncthreads();
    if (tx < SPLIT) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) { red[ty * NACC + OAK_SD(d)] = gl[d]; red[ty * NACC + DT + OAK_SD(d)] = WANT_GK ? gk[WANT_GK ? d : 0] : 0.0; }
        if (tx == 0) {
#pragma unroll
            for (int q = 0; q <= R; ++q) red[ty * NACC + 2 * DT + q] = gw[q];
        }
    }
    __syncthreads();
    const int RA = dd.R;                                  // actual depth <= R (depth 9..16 runs the R = 12 / 16 instantiations, zero weights above)
    const int64_t reclen = 2 * D + (RA + 1) + tablen;
    double* rec = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * reclen;
    auto sum4 = [&](int j) { return ((red[j] + red[NACC + j]) + red[2 * NACC + j]) + red[3 * NACC + j]; };
    for (int d = tid; d < D; d += 256) { rec[d] = sum4(d) * 1024.0; rec[D + d] = sum4(DT + d); }
    if (tid <= RA) rec[2 * D + tid] = sum4(2 * DT + tid);
    for (int idx = tid; idx < tablen; idx += 256)
        rec[2 * D + (RA + 1) + idx] = ((accT[idx] + accT[tablen + idx]) + accT[2 * tablen + idx]) + accT[3 * tablen + idx];
}
#undef OAK_SD

// ---------------------------------------------------------------------------------------------------------------------
// Gradient with respect to the INDUCING INPUTS (create_model_oak(zfixed=False), oak/model_utils.py:156-157; the reference
// gets it from TensorFlow's autodiff through Kuf and Kuu).  Same pair walk as the fast kernel, but the accumulators belong
// to the COLUMN (inducing point) a lane owns instead of being summed over the workgroup:
//     dF/dz_{m,d} = sum_n g_nm * dK/dk_d(n,m) * dk_d/dz_m,      dk_d/dz_m = E * 64 ln2 s_d u' - cn_d(x_n) * dcn_d/dz (z_m)
// (u' = (xs_n - xs_m)/32, E the RBF factor incl. base variance, exp2w.h).  The constant 64 ln2 s_d is divided out of the
// staged column feature dzb = (dcn/dz) / (64 ln2 s_d) and multiplied back on the host; discrete dimensions contribute 0.
// A second pass over the pairs, paid only when the inducing inputs are trainable.
// ---------------------------------------------------------------------------------------------------------------------
// d cn / d x of the normalised constraint term cn = cov_X_s(x) / sqrt(var_s), per RBF dimension (featurize_kernel's cn)
__global__ void __launch_bounds__(256) featurize_dx_kernel(DevDesc dd, DevMeasure dm, const double* __restrict__ meas,
                                                           const double* __restrict__ X, int64_t n, int ldx, int64_t ld,
                                                           double* __restrict__ dzb) {
    const int d = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ld) return;
    double v = 0.0;
    if (i < n && dd.type[d] == OAK_DIM_RBF) {
        const double x = X[i * ldx + dd.col[d]];
        const double l = dm.ls[d], bv = dd.bv[d];
        double dc = 0.0;                       // d cov_X_s / dx
        switch (dm.kind[d]) {
            case OAK_MEAS_GAUSSIAN: {
                const double mu = dm.p0[d], s = l * l + dm.p1[d];
                dc = -bv * l / sqrt(s) * exp(-0.5 * (x - mu) * (x - mu) / s) * (x - mu) / s;
            } break;
            case OAK_MEAS_UNIFORM: {
                const double a = dm.p0[d], b = dm.p1[d], il2 = 0.5 / (l * l);
                dc = bv / (b - a) * (exp(-(a - x) * (a - x) * il2) - exp(-(b - x) * (b - x) * il2));
            } break;
            case OAK_MEAS_EMPIRICAL: {
                const int K = dm.k[d];
                const double* loc = meas + dm.off[d];
                const double* w = loc + K;
                const double il2 = 0.5 / (l * l);
                double acc = 0.0;
                for (int k = 0; k < K; ++k) { const double u = x - loc[k]; acc += w[k] * exp(-u * u * il2) * (-u); }
                dc = bv * acc / (l * l);
            } break;
            case OAK_MEAS_MOG: {
                const int K = dm.k[d];
                const double* mu = meas + dm.off[d];
                const double* var = mu + K;
                const double* w = var + K;
                double acc = 0.0;
                for (int k = 0; k < K; ++k) {
                    const double s = l * l + var[k], u = x - mu[k];
                    acc += w[k] * exp(-0.5 * u * u / s) / sqrt(s) * (-u / s);
                }
                dc = bv * l * acc;
            } break;
            default: dc = 0.0;
        }
        v = dc * dm.inv_sqrt_v[d] / (64.0 * 0.6931471805599453094 * dd.scale[d]);
    }
    dzb[(int64_t)d * ld + i] = v;
}

template <int R, int DMAX, int CPT, bool ALLRBF, bool UNITBV>
__global__ void __launch_bounds__(256, (DMAX <= 8 ? 2 : 1))      // 16 dims x 2 columns of accumulators: no spills at one wave per SIMD
gram_bwd_z_kernel(const DevDesc dd, const double* __restrict__ tables, const double* __restrict__ Apack, int64_t a0, int64_t na,
                  const double* __restrict__ Bxs, const double* __restrict__ Bcn, const double* __restrict__ Bdz, int64_t b_ld,
                  int64_t nb, const double* __restrict__ G, int64_t ldg, const double* __restrict__ yA,
                  const double* __restrict__ avec, double g_scale, int rows_per_wg, double* __restrict__ partial /* [nrb][nb][DMAX] */) {
    constexpr int TJ = 64 * CPT, RT = 2, RS = 4 * RT;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int D = dd.D;
    double* Bx = smem;                  // [DMAX][TJ]
    double* Bc = Bx + DMAX * TJ;
    double* Bd = Bc + DMAX * TJ;        // (dcn/dz) / (64 ln2 s_d)
    double* Av = Bd + DMAX * TJ;        // [TJ]
    double* Tab = Av + TJ;              // [EW_N]
    double* Cw = Tab + EW_N;            // [DMAX]  (not allocated when UNITBV)
    double* Cm = Cw + DMAX;
    int* meta = reinterpret_cast<int*>(UNITBV ? Tab + EW_N : Cm + DMAX);   // [2*DMAX] when !ALLRBF
    double* accum = Bx;                 // [TJ][DMAX] cross-wave accumulation, aliases the column features after the row loop
    const int tid = threadIdx.x, tx = tid & 63;
    const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t jb = (int64_t)blockIdx.x * TJ;
    const int64_t ib = (int64_t)blockIdx.y * rows_per_wg;
    const int64_t iend = (ib + rows_per_wg < na) ? ib + rows_per_wg : na;
    for (int idx = tid; idx < DMAX * TJ; idx += 256) {
        const int d = idx / TJ, j = idx - d * TJ;
        const int64_t gj = jb + j;
        const bool ok = gj < nb && d < D;
        Bx[idx] = ok ? Bxs[(int64_t)d * b_ld + gj] : (d < D ? 0.0 : 1.0);
        Bc[idx] = ok ? Bcn[(int64_t)d * b_ld + gj] : 0.0;
        Bd[idx] = ok ? Bdz[(int64_t)d * b_ld + gj] : 0.0;
    }
    for (int j = tid; j < TJ; j += 256) Av[j] = (avec != nullptr && jb + j < nb) ? avec[jb + j] : 0.0;
    for (int j = tid; j < EW_N; j += 256) Tab[j] = biased_table_entry(j);
    if constexpr (!UNITBV) {
        if (tid < DMAX) {
            const bool rbf = tid < D && dd.type[tid] == OAK_DIM_RBF;
            Cw[tid] = rbf ? dd.woff[tid] : 0.0;
            Cm[tid] = rbf ? dd.magic[tid] : EW_MAGIC;
        }
    }
    unsigned rbf_mask = 0xffffffffu;
    if constexpr (!ALLRBF) {
        if (tid < DMAX) { meta[2 * tid] = tid < D ? dd.tab_off[tid] : 0; meta[2 * tid + 1] = tid < D ? dd.ncat[tid] : 0; }
        for (int d = 0; d < D; ++d) if (dd.type[d] != OAK_DIM_RBF) rbf_mask &= ~(1u << d);
    }
    double gz[CPT][DMAX];
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int d = 0; d < DMAX; ++d) gz[c][d] = 0.0;
    __syncthreads();

    struct Chunk { double xa[4], xb[4], ca[4], cb[4], bd[4], cw[4], cm[4]; };
    auto g_addr = [&](int64_t i0n, int prn) -> const double* {
        const int64_t gin = i0n + ty * RT + prn / CPT, gjn = jb + tx + 64 * (prn % CPT);
        return G + (gin < iend ? gin : iend - 1) * ldg + (gjn < nb ? gjn : nb - 1);
    };
    double graw_next = ib < iend ? *g_addr(ib, 0) : 0.0;
    for (int64_t i0 = ib; i0 < iend; i0 += RS) {
#pragma unroll 1
        for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CPT; ++c) {                  // c is static: it selects the accumulator set
            const int pr = r * CPT + c;
            const int col = tx + 64 * c;
            const int64_t gi = i0 + ty * RT + r, gj = jb + col;
            const int64_t gr = gi < iend ? gi : iend - 1;
            const double* __restrict__ prow = Apack + gr * (3 * DMAX);
            const double yrow = yA != nullptr ? yA[a0 + gr] : 0.0;
            const double g = (gi < iend && gj < nb) ? __builtin_fma(g_scale, graw_next, yrow * Av[col]) : 0.0;
            graw_next = (pr + 1 < RT * CPT) ? *g_addr(i0, pr + 1) : *g_addr(i0 + RS, 0);
            double k[DMAX], zc[DMAX];
            auto fetch = [&](int d0, Chunk& ch) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int d = d0 + v;
                    ch.xa[v] = prow[d]; ch.ca[v] = prow[DMAX + d];
                    ch.xb[v] = Bx[d * TJ + col]; ch.cb[v] = Bc[d * TJ + col]; ch.bd[v] = Bd[d * TJ + col];
                    if constexpr (!UNITBV) { ch.cw[v] = Cw[d]; ch.cm[v] = Cm[d]; }
                }
            };
            Chunk cur, nxt;
            fetch(0, cur);
#pragma unroll
            for (int d0 = 0; d0 < DMAX; d0 += 4) {
                if (d0 + 4 < DMAX) fetch(d0 + 4, nxt);
                asm volatile("" ::: "memory");
                double w[4], up[4], mg[4], E[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    up[v] = cur.xa[v] - cur.xb[v];
                    if constexpr (UNITBV) { w[v] = fma_clamp01(up[v], up[v], 0.0); mg[v] = EW_MAGIC; }
                    else { w[v] = fma_clamp01(up[v], up[v], cur.cw[v]); mg[v] = cur.cm[v]; }
                }
                exp2_w_vec<4>(w, mg, E, Tab);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int d = d0 + v;
                    double kv = __builtin_fma(-cur.ca[v], cur.cb[v], E[v]);
                    double zv = __builtin_fma(-cur.ca[v], cur.bd[v], E[v] * up[v]);
                    if constexpr (!ALLRBF) {
                        if (!((rbf_mask >> d) & 1u)) { kv = tables[meta[2 * d] + (int)cur.xa[v] * meta[2 * d + 1] + (int)cur.xb[v]]; zv = 0.0; }
                    }
                    k[d] = kv; zc[d] = zv;
                }
                if (d0 + 4 < DMAX) cur = nxt;
            }
            double e[R];
#pragma unroll
            for (int q = 0; q < R; ++q) e[q] = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
#pragma unroll
                for (int q = R - 1; q >= 1; --q) e[q] = __builtin_fma(k[d], e[q - 1], e[q]);
                e[0] += k[d];
            }
            double cg[R];                                    // pair-level Horner coefficients of dK/dk_d (see gram_bwd_fast_kernel)
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double ci = dd.w[i + 1];
#pragma unroll
                for (int q = i + 1; q < R; ++q) ci = __builtin_fma(dd.w[q + 1], e[q - i - 1], ci);
                cg[i] = (i & 1) ? -(g * ci) : g * ci;
            }
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                double gc = cg[R - 1];
#pragma unroll
                for (int i = R - 2; i >= 0; --i) gc = __builtin_fma(gc, k[d], cg[i]);
                gz[c][d] = __builtin_fma(gc, zc[d], gz[c][d]);
            }
        }
    }
    // the four waves hold partial sums for the SAME columns: add them in wave order through LDS, then one store per column
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
        if (ty == wv) {
#pragma unroll
            for (int c = 0; c < CPT; ++c)
#pragma unroll
                for (int d = 0; d < DMAX; ++d) {
                    double* q = &accum[(tx + 64 * c) * DMAX + d];
                    *q = (wv == 0) ? gz[c][d] : *q + gz[c][d];
                }
        }
        __syncthreads();
    }
    double* out = partial + ((int64_t)blockIdx.y * nb + jb) * DMAX;
    for (int idx = tid; idx < TJ * DMAX; idx += 256)
        if (jb + idx / DMAX < nb) out[idx] = accum[idx];
}

// General form of the inducing-input gradient: any depth the backward pass is instantiated for, up to OAK_MAX_DIMS sub-kernels,
// grouped sub-kernels (the further columns of a group get their own output slots D .. D + nx - 1).  64 columns (inducing points)
// per workgroup, one per lane; the waves of a workgroup take the rows in turn and keep their own accumulators [zd][64] in LDS
// (as many waves as fit, merged in wave order at the end: deterministic).  Two passes per pair like gram_bwd_kernel -- e_1..e_R,
// then the leave-one-out coefficient of every sub-kernel -- with the row features read as wave-uniform scalars.  Same output
// scaling as the fast kernel: slot value = sum g dK/dk_d (E u / 32 - cn_d(x) dzb), times 64 ln2 s_d on the host.
template <int R>
__global__ void __launch_bounds__(256)
gram_bwd_z_general_kernel(const DevDesc dd, const double* __restrict__ tables, const double* __restrict__ Axs, const double* __restrict__ Acn,
                          const double* __restrict__ Axx, int64_t a_ld, int64_t a0, int64_t na, const double* __restrict__ Bxs,
                          const double* __restrict__ Bcn, const double* __restrict__ Bdz, const double* __restrict__ Bxx, int64_t b_ld, int64_t nb,
                          const double* __restrict__ G, int64_t ldg, const double* __restrict__ yA, const double* __restrict__ avec, double g_scale,
                          int rows_per_wg, int nx, double* __restrict__ partial) {
    constexpr int TJ = 64;
    constexpr int RR = R > 0 ? R : 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int D = dd.D, zd = D + nx;
    const int nw = blockDim.x >> 6;
    double* Bx = smem;                  // [D][64]
    double* Bc = Bx + D * TJ;
    double* Bd = Bc + D * TJ;
    double* Bq = Bd + D * TJ;           // [nx][64]
    double* Tab = Bq + nx * TJ;         // [64]
    double* acc = Tab + 64;             // [nw][zd][64]
    const int tid = threadIdx.x, tx = tid & 63;
    const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t jb = (int64_t)blockIdx.x * TJ;
    const int64_t ib = (int64_t)blockIdx.y * rows_per_wg;
    const int64_t iend = (ib + rows_per_wg < na) ? ib + rows_per_wg : na;
    for (int idx = tid; idx < D * TJ; idx += blockDim.x) {
        const int d = idx / TJ, j = idx - d * TJ;
        const bool ok = jb + j < nb;
        Bx[idx] = ok ? Bxs[(int64_t)d * b_ld + jb + j] : 0.0;
        Bc[idx] = ok ? Bcn[(int64_t)d * b_ld + jb + j] : 0.0;
        Bd[idx] = ok ? Bdz[(int64_t)d * b_ld + jb + j] : 0.0;
    }
    for (int idx = tid; idx < nx * TJ; idx += blockDim.x) {
        const int q = idx / TJ, j = idx - q * TJ;
        Bq[idx] = (jb + j < nb) ? Bxx[(int64_t)q * b_ld + jb + j] : 0.0;
    }
    if (tid < 64) Tab[tid] = c_exp2_table_g[tid];
    for (int idx = tid; idx < nw * zd * TJ; idx += blockDim.x) acc[idx] = 0.0;
    __syncthreads();
    double* my = acc + (size_t)ty * zd * TJ + tx;
    const int64_t col = jb + tx;
    const double av = (avec != nullptr && col < nb) ? avec[col] : 0.0;
    for (int64_t i = ib + ty; i < iend; i += nw) {
        const int64_t gi = a0 + i;
        double g = 0.0;
        if (col < nb) g = g_scale * G[i * ldg + col] + (yA != nullptr ? yA[gi] * av : 0.0);
        double e[RR];
#pragma unroll
        for (int q = 0; q < RR; ++q) e[q] = 0.0;
        auto pair_k = [&](int d, double& E, double& u) -> double {     // k_d of this pair; E and u of the dim's first column
            const double xa = Axs[(int64_t)d * a_ld + gi];
            if (dd.type[d] == OAK_DIM_RBF) {
                u = xa - Bx[d * TJ + tx];
                double t = __builtin_fma(-u, u, dd.log2bv[d]);
                for (int q = dd.xrow[d]; q < dd.xrow[d] + dd.nxc[d]; ++q) {
                    const double uq = Axx[(int64_t)q * a_ld + gi] - Bq[q * TJ + tx];
                    t = __builtin_fma(-uq, uq, t);
                }
                E = exp2_neg_tab(t, Tab);
                return __builtin_fma(-Acn[(int64_t)d * a_ld + gi], Bc[d * TJ + tx], E);
            }
            E = 0.0; u = 0.0;
            return tables[dd.tab_off[d] + (int)xa * dd.ncat[d] + (int)Bx[d * TJ + tx]];
        };
        if constexpr (R > 0) {
            for (int d = 0; d < D; ++d) {
                double E, u;
                const double k = pair_k(d, E, u);
#pragma unroll
                for (int q = R - 1; q >= 1; --q) e[q] = __builtin_fma(k, e[q - 1], e[q]);
                e[0] += k;
            }
            for (int d = 0; d < D; ++d) {
                if (dd.type[d] != OAK_DIM_RBF) continue;
                double E, u;
                const double k = pair_k(d, E, u);
                double f = 1.0, coef = dd.w[1];
#pragma unroll
                for (int q = 1; q < R; ++q) {
                    f = __builtin_fma(-k, f, e[q - 1]);
                    coef = __builtin_fma(dd.w[q + 1], f, coef);
                }
                const double gc = g * coef;
                const double zc = __builtin_fma(-Acn[(int64_t)d * a_ld + gi], Bd[d * TJ + tx], E * (u * 0.03125));
                my[d * TJ] = __builtin_fma(gc, zc, my[d * TJ]);
                for (int q = dd.xrow[d]; q < dd.xrow[d] + dd.nxc[d]; ++q) {
                    const double uq = Axx[(int64_t)q * a_ld + gi] - Bq[q * TJ + tx];
                    my[(D + q) * TJ] = __builtin_fma(gc, E * (uq * 0.03125), my[(D + q) * TJ]);
                }
            }
        }
    }
    __syncthreads();
    double* out = partial + ((int64_t)blockIdx.y * nb + jb) * zd;
    for (int idx = tid; idx < TJ * zd; idx += blockDim.x) {
        const int j = idx / zd, slot = idx - j * zd;
        if (jb + j >= nb) continue;
        double v = acc[(size_t)slot * TJ + j];
        for (int w = 1; w < nw; ++w) v += acc[((size_t)w * zd + slot) * TJ + j];
        out[idx] = v;
    }
}

// gz[m][d] += sum over row blocks (fixed order)
__global__ void __launch_bounds__(256) reduce_gz_kernel(const double* __restrict__ partial, int64_t nrb, int64_t len, double* __restrict__ gz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    double s = 0.0;
    for (int64_t b = 0; b < nrb; ++b) s += partial[b * len + i];
    gz[i] += s;
}

// Diagonal term: sum_n gconst * (gvec ? gvec[n] : 1) * dKdiag_n / dtheta, one point per lane.
template <int R>
__global__ void __launch_bounds__(256)
diag_bwd_kernel(const DevDesc dd, const double* __restrict__ tables, int tablen, const double* __restrict__ Axs,
                const double* __restrict__ Acn, const double* __restrict__ Adcn, int64_t a_ld, int64_t n, double gconst,
                const double* __restrict__ gvec, int64_t rows_per_wg, double* __restrict__ partial) {
    constexpr int RR = R > 0 ? R : 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int D = dd.D;
    double* accL = smem;             // [4][D]
    double* accK = accL + 4 * D;     // [4][D]
    double* accT = accK + 4 * D;     // [4][tablen]: one copy per wave
    double* red = accT + 4 * tablen; // [4][R+1]
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    for (int idx = tid; idx < 8 * D + 4 * tablen; idx += 256) smem[idx] = 0.0;
    __syncthreads();
    double gw[R + 1];
#pragma unroll
    for (int q = 0; q <= R; ++q) gw[q] = 0.0;
    const int64_t i_begin = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t i_end = (i_begin + rows_per_wg < n) ? i_begin + rows_per_wg : n;
    // DT points per lane per trip: pass 1 keeps their ESPs in registers, pass 2 runs dimension-outer so the two wave
    // reductions per dimension are paid once per DT * 64 points instead of once per 64
    constexpr int DT = 8;
    for (int64_t i0 = i_begin; i0 < i_end; i0 += 256 * DT) {
        double g[DT], e[DT][RR];
        int64_t ic[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int64_t i = i0 + t * 256 + tid;
            const bool ok = i < i_end;
            g[t] = ok ? (gvec != nullptr ? gconst * gvec[i] : gconst) : 0.0;
            ic[t] = ok ? i : i_begin;
#pragma unroll
            for (int q = 0; q < RR; ++q) e[t][q] = 0.0;
        }
        if constexpr (R > 0) {
            for (int d = 0; d < D; ++d) {
                const bool rbf = dd.type[d] == OAK_DIM_RBF;
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    double k;
                    if (rbf) { const double c = Acn[(int64_t)d * a_ld + ic[t]]; k = __builtin_fma(-c, c, dd.bv[d]); }
                    else k = tables[dd.tab_off[d] + dd.ncat[d] * dd.ncat[d] + (int)Axs[(int64_t)d * a_ld + ic[t]]];
#pragma unroll
                    for (int q = R - 1; q >= 1; --q) e[t][q] = __builtin_fma(k, e[t][q - 1], e[t][q]);
                    e[t][0] += k;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            gw[0] += g[t];
#pragma unroll
            for (int q = 1; q <= R; ++q) gw[q] = __builtin_fma(g[t], e[t][q - 1], gw[q]);
        }
        if constexpr (R > 0) {
            for (int d = 0; d < D; ++d) {
                const bool rbf = dd.type[d] == OAK_DIM_RBF;
                double cl = 0.0, ck = 0.0;
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    double k, dkl = 0.0;
                    int tidx = 0;
                    if (rbf) {
                        const double c = Acn[(int64_t)d * a_ld + ic[t]];
                        k = __builtin_fma(-c, c, dd.bv[d]);
                        dkl = -2.0 * c * Adcn[(int64_t)d * a_ld + ic[t]];
                    } else {
                        const int C = dd.ncat[d];
                        const int xi = (int)Axs[(int64_t)d * a_ld + ic[t]];
                        k = tables[dd.tab_off[d] + C * C + xi];
                        tidx = dd.tab_off[d] + xi * C + xi;
                    }
                    double f = 1.0, coef = dd.w[1];
#pragma unroll
                    for (int q = 1; q < R; ++q) { f = __builtin_fma(-k, f, e[t][q - 1]); coef = __builtin_fma(dd.w[q + 1], f, coef); }
                    const double gc = g[t] * coef;
                    cl = __builtin_fma(gc, dkl, cl);
                    ck = __builtin_fma(gc, k, ck);
                    if (!rbf && dd.type[d] == OAK_DIM_CATEGORICAL && gc != 0.0) atomicAdd(&accT[ty * tablen + tidx], gc);
                }
                cl = wave_sum(cl); ck = wave_sum(ck);
                if (tx == 0) { accL[ty * D + d] += cl; accK[ty * D + d] += ck; }
            }
        }
    }
#pragma unroll
    for (int q = 0; q <= R; ++q) gw[q] = wave_sum(gw[q]);
    if (tx == 0) {
#pragma unroll
        for (int q = 0; q <= R; ++q) red[ty * (R + 1) + q] = gw[q];
    }
    __syncthreads();
    const int RA = dd.R;                                  // actual depth <= R (see gram_bwd_kernel)
    const int64_t reclen = 2 * D + (RA + 1) + tablen;
    double* rec = partial + (int64_t)blockIdx.x * reclen;
    for (int idx = tid; idx < D; idx += 256) {
        rec[idx] = ((accL[idx] + accL[D + idx]) + accL[2 * D + idx]) + accL[3 * D + idx];
        rec[D + idx] = ((accK[idx] + accK[D + idx]) + accK[2 * D + idx]) + accK[3 * D + idx];
    }
    if (tid <= RA) rec[2 * D + tid] = ((red[tid] + red[(R + 1) + tid]) + red[2 * (R + 1) + tid]) + red[3 * (R + 1) + tid];
    for (int idx = tid; idx < tablen; idx += 256)
        rec[2 * D + (RA + 1) + idx] = ((accT[idx] + accT[tablen + idx]) + accT[2 * tablen + idx]) + accT[3 * tablen + idx];
}

// out[j] += sum_w partial[w][j]: one workgroup per entry j, thread t adds records t, t + 256, ... in order, then a fixed
// LDS tree -- deterministic, and 2048 records take microseconds instead of one serial chain per entry.
__global__ void __launch_bounds__(256) reduce_records_kernel(const double* __restrict__ partial, int64_t nrec, int64_t reclen,
                                                             double* __restrict__ out) {
    __shared__ double red[256];
    const int64_t j = blockIdx.x;
    double s = 0.0;
    for (int64_t w = threadIdx.x; w < nrec; w += 256) s += partial[w * reclen + j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[j] += red[0];
}

int64_t record_len(const PreparedKernel& pk) { return 2 * pk.dd.D + (pk.dd.R + 1) + (int64_t)pk.tables.size(); }

// d_rec[reclen] += contraction of G (na x nb block, + optional rank-1 yA avec^T) with dK/dtheta over pairs (A rows a0.., B)
int gram_bwd(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, const double* d_G,
             int64_t ldg, double g_scale, const double* d_yA, const double* d_avec, double* d_rec, bool want_gk) {
    if (na <= 0 || B.n <= 0) return OAK_OK;
    OAK_REQUIRE(A.dcn != nullptr && B.dcn != nullptr, "gram_bwd: features were not prepared for the backward pass");
    const int D = pk.dd.D, R = pk.dd.R;
    const int tablen = (int)pk.tables.size();
    OAK_REQUIRE(tablen <= 1024, "gradient: discrete tables too large (%d doubles; four per-wave copies are kept in LDS)", tablen);
    const int nx = pk.grouped ? A.nx : 0;       // grouped sub-kernels go through the general kernel
    if (pk.grouped) OAK_REQUIRE(A.xx != nullptr && B.xx != nullptr && A.nx == B.nx, "gram_bwd: features lack the grouped sub-kernels' further columns");
    // register-resident pair walk: depth <= 8 with <= 32 sub-kernels (r04: depth 5..8 too -- one instantiation per shape, the mixed / any-variance /
    // base-variance-gradient form; the general two-pass kernel took 3.5x (depth 8 of 16) to 6x (8 of 32) as long)
    // (depth 9..16, which needs >= 9 sub-kernels: the R = 12 / 16 instantiations at <= 16 sub-kernels, one wave per SIMD)
    // (33..64 sub-kernels: lane QUADS, depth <= 8)
    const bool fast = (R >= 1 && (D <= 32 ? R <= 16 : R <= 8) && nx == 0 && getenv("OAK_BWD_GENERIC") == nullptr);
    bool allrbf = true;
    for (int d = 0; d < D; ++d) allrbf = allrbf && pk.dd.type[d] == OAK_DIM_RBF;
    bool unitbv = true;
    for (int d = 0; d < D; ++d) unitbv = unitbv && (pk.dd.type[d] != OAK_DIM_RBF || pk.dd.bv[d] == 1.0);
    // depth 5..8: two forms per shape -- all-continuous / unit base variances / no base-variance sums (the reference's default model), and the
    // mixed / any-variance / base-variance-gradient form, which evaluates everything else (lane pairs, > 16 sub-kernels: only the latter)
    const bool plain58 = fast && allrbf && unitbv && !want_gk && ((R > 4 && D <= 16) || (R <= 4 && D > 32));     // (also the plain lane-quad form at depth <= 4)
    if (fast && (R > 4 || D > 32) && !plain58) { allrbf = false; unitbv = false; want_gk = true; }
    const int dmax = D <= 8 ? 8 : (D <= 16 ? 16 : (D <= 32 ? 32 : 64));
    // general kernel: two columns per lane while two workgroups still fit a CU's LDS (<= 24 sub-kernels) and the depth leaves registers (<= 16)
    int cpt = fast ? (dmax <= 16 ? 2 : 1) : ((D <= 24 && R <= 16) ? 2 : 1);
    if (const char* e = getenv("OAK_BWD_CPT")) { if (!fast && R <= 16 && (e[0] == '1' || e[0] == '2')) cpt = e[0] - '0'; }      // tuning knob (depth > 16 is instantiated for one column only)
    const int TJ = (fast && dmax == 64) ? 32 : 64 * cpt, RS = 8;       // lane quads: 16 columns per wave pass, two passes
    const size_t lds = fast ? sizeof(double) * ((size_t)3 * dmax * TJ + TJ + EW_N + (unitbv ? 0 : 2 * dmax) + 5 * tablen + (allrbf ? 0 : dmax))
                            : sizeof(double) * ((size_t)3 * D * TJ + (size_t)3 * D * RS + RS + TJ + 64 + 8 * D + 4 * tablen + 64 + (size_t)nx * (TJ + RS));
    OAK_REQUIRE(lds <= 160 * 1024, "gram_bwd: LDS request %zu exceeds 160 KiB", lds);
    const int64_t nb = B.n;
    const int64_t ncb = (nb + TJ - 1) / TJ;
    int wg_per_cu = 16;      // measured: 16.99 / 16.68 / 16.55 ms at 8 / 16 / 32 per CU (headline size)
    if (const char* e = getenv("OAK_BWD_WG_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 64) wg_per_cu = v; }   // tuning knob
    int64_t nrb = ((int64_t)ctx->num_cu * wg_per_cu + ncb - 1) / ncb;
    int64_t rows = (na + nrb - 1) / nrb;
    rows = ((rows + RS - 1) / RS) * RS;
    if (rows < RS) rows = RS;
    if (rows > 4096) rows = 4096;
    nrb = (na + rows - 1) / rows;
    if (nrb > 65535) { rows = (((na + 65534) / 65535 + RS - 1) / RS) * RS; nrb = (na + rows - 1) / rows; }
    const int64_t reclen = record_len(pk);
    double* d_part = nullptr;
    // r05: rows in lanes for the reference's default continuous model (grad_rows.hip) -- the column features are scalar loads, the
    // row's live in registers; three LDS reads per pair-dimension fewer.  OPT-IN (OAK_BWD_ROWS=1): built because the r04 verdict asked
    // for it, it fits 256 VGPRs without scratch (253) and halves the LDS instructions, but measures 15.1-15.5 ms against 14.8 at the
    // headline shape -- the wait cycles move from LDS to scalar / vector memory (profiles/r05_pmc_pair_kernels.json).
    bool rows_form = fast && allrbf && unitbv && !want_gk && tablen == 0 && (dmax == 8 || dmax == 16) && R >= 1 && R <= (dmax == 16 ? 3 : 4) && na >= 16384;
    { const char* e = getenv("OAK_BWD_ROWS"); rows_form = rows_form && e != nullptr && atoi(e) != 0; }
    if (rows_form) {
        OAK_REQUIRE(A.xs32 != nullptr && B.xs32 != nullptr, "gram_bwd: features were not prepared for the backward pass");
        int cols_per_wg = 256;
        if (const char* e = getenv("OAK_BWD_ROWS_COLS")) { int v = atoi(e); if (v >= 16 && v % 16 == 0) cols_per_wg = v; }
        const int64_t ncb_r = (nb + cols_per_wg - 1) / cols_per_wg, nrb_r = (na + 255) / 256;
        double *d_apack = nullptr, *d_bpack = nullptr;
        OAK_CHECK(get_buf_t(ctx, "bwd_part", (size_t)(nrb_r * ncb_r * reclen), &d_part));
        OAK_CHECK(get_buf_t(ctx, "bwd_pack", (size_t)na * 3 * dmax, &d_apack));
        OAK_CHECK(get_buf_t(ctx, "bwd_pack_cols", (size_t)nb * 3 * dmax, &d_bpack));
        OAK_CHECK(launch_pack_rows(ctx, A, a0, na, D, dmax, d_apack, 1, -1.0));
        OAK_CHECK(launch_pack_rows(ctx, B, 0, nb, D, dmax, d_bpack, -1, 1.0));      // [column][chunk of four dims][feature][dim in chunk]
        int64_t nrec = 0;
        OAK_CHECK(gram_bwd_rows_launch(ctx, pk, dmax, d_apack, a0, na, d_bpack, nb, d_G, ldg, d_yA, d_avec, g_scale, cols_per_wg, d_part, &nrec));
        reduce_records_kernel<<<(unsigned)reclen, 256, 0, ctx->stream>>>(d_part, nrec, reclen, d_rec);
        OAK_HIP_CHECK(hipGetLastError());
        return OAK_OK;
    }
    OAK_CHECK(get_buf_t(ctx, "bwd_part", (size_t)(nrb * ncb * reclen), &d_part));
    dim3 grid((unsigned)ncb, (unsigned)nrb);
    double* d_pack = nullptr;
    if (fast) {
        OAK_REQUIRE(A.xs32 != nullptr && B.xs32 != nullptr, "gram_bwd: features were not prepared for the backward pass");
        OAK_CHECK(get_buf_t(ctx, "bwd_pack", (size_t)na * 3 * dmax, &d_pack));
        OAK_CHECK(launch_pack_rows(ctx, A, a0, na, D, dmax, d_pack, dmax > 16 ? dmax / 16 : 1));
    }
#define OAK_BWD_LAUNCH(RR, CP)                                                                                                   \
    {                                                                                                                            \
        auto kern = gram_bwd_kernel<RR, CP>;                                                                                     \
        if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)kern)); \
        kern<<<grid, 256, lds, ctx->stream>>>(pk.dd, pk.d_tables, tablen, A.xs, A.cn, A.dcn, A.ld, a0, na, B.xs, B.cn, B.dcn,  \
                                              B.ld, nb, d_G, ldg, d_yA, d_avec, g_scale, (int)rows, d_part, A.xx, B.xx, nx);    \
    }
#define OAK_BWD_FAST_K(RR, DM, AR, GK, UB)                                                                                         \
    {                                                                                                                             \
        auto kern = gram_bwd_fast_kernel<RR, (DM <= 16 ? DM : 16), 2, AR, GK, UB, (DM <= 16 ? 1 : DM / 16)>;                       \
        if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)kern)); \
        kern<<<grid, 256, lds, ctx->stream>>>(pk.dd, pk.d_tables, tablen, d_pack, a0, na, B.xs32, B.cn, B.dcs, B.ld, nb, d_G, ldg, \
                                              d_yA, d_avec, g_scale, (int)rows, d_part);                                          \
    }
#define OAK_BWD_FAST_U(RR, DM, AR, GK) { if (unitbv) OAK_BWD_FAST_K(RR, DM, AR, GK, true) else OAK_BWD_FAST_K(RR, DM, AR, GK, false) }
#define OAK_BWD_FAST_G(RR, DM, AR) { if (want_gk) OAK_BWD_FAST_U(RR, DM, AR, true) else OAK_BWD_FAST_U(RR, DM, AR, false) }
#define OAK_BWD_FAST(RR, DM) { if (allrbf) OAK_BWD_FAST_G(RR, DM, true) else OAK_BWD_FAST_G(RR, DM, false) }
#define OAK_BWD_CASE(RR) case RR: if (cpt == 2) OAK_BWD_LAUNCH(RR, 2) else OAK_BWD_LAUNCH(RR, 1) break;
    if (fast) {
        switch ((R <= 8 ? R : template_depth(R)) * 100 + dmax) {
            case 1216: if (plain58) OAK_BWD_FAST_K(12, 16, true, false, true) else OAK_BWD_FAST_K(12, 16, false, true, false) break;
            case 1616: if (plain58) OAK_BWD_FAST_K(16, 16, true, false, true) else OAK_BWD_FAST_K(16, 16, false, true, false) break;
            case 164: if (plain58) OAK_BWD_FAST_K(1, 64, true, false, true) else OAK_BWD_FAST_K(1, 64, false, true, false) break;
            case 264: if (plain58) OAK_BWD_FAST_K(2, 64, true, false, true) else OAK_BWD_FAST_K(2, 64, false, true, false) break;
            case 364: if (plain58) OAK_BWD_FAST_K(3, 64, true, false, true) else OAK_BWD_FAST_K(3, 64, false, true, false) break;
            case 464: if (plain58) OAK_BWD_FAST_K(4, 64, true, false, true) else OAK_BWD_FAST_K(4, 64, false, true, false) break;
            case 564: OAK_BWD_FAST_K(5, 64, false, true, false) break;   case 664: OAK_BWD_FAST_K(6, 64, false, true, false) break;
            case 764: OAK_BWD_FAST_K(7, 64, false, true, false) break;   case 864: OAK_BWD_FAST_K(8, 64, false, true, false) break;
            case 1232: OAK_BWD_FAST_K(12, 32, false, true, false) break;
            case 1632: OAK_BWD_FAST_K(16, 32, false, true, false) break;
            case 108: OAK_BWD_FAST(1, 8) break;   case 116: OAK_BWD_FAST(1, 16) break;
            case 208: OAK_BWD_FAST(2, 8) break;   case 216: OAK_BWD_FAST(2, 16) break;
            case 308: OAK_BWD_FAST(3, 8) break;   case 316: OAK_BWD_FAST(3, 16) break;
            case 408: OAK_BWD_FAST(4, 8) break;   case 416: OAK_BWD_FAST(4, 16) break;
            case 132: OAK_BWD_FAST(1, 32) break;  case 232: OAK_BWD_FAST(2, 32) break;
            case 332: OAK_BWD_FAST(3, 32) break;  case 432: OAK_BWD_FAST(4, 32) break;
            case 508: if (plain58) OAK_BWD_FAST_K(5, 8, true, false, true) else OAK_BWD_FAST_K(5, 8, false, true, false) break;
            case 516: if (plain58) OAK_BWD_FAST_K(5, 16, true, false, true) else OAK_BWD_FAST_K(5, 16, false, true, false) break;
              case 532: OAK_BWD_FAST_K(5, 32, false, true, false) break;
            case 608: if (plain58) OAK_BWD_FAST_K(6, 8, true, false, true) else OAK_BWD_FAST_K(6, 8, false, true, false) break;
            case 616: if (plain58) OAK_BWD_FAST_K(6, 16, true, false, true) else OAK_BWD_FAST_K(6, 16, false, true, false) break;
              case 632: OAK_BWD_FAST_K(6, 32, false, true, false) break;
            case 708: if (plain58) OAK_BWD_FAST_K(7, 8, true, false, true) else OAK_BWD_FAST_K(7, 8, false, true, false) break;
            case 716: if (plain58) OAK_BWD_FAST_K(7, 16, true, false, true) else OAK_BWD_FAST_K(7, 16, false, true, false) break;
              case 732: OAK_BWD_FAST_K(7, 32, false, true, false) break;
            case 808: if (plain58) OAK_BWD_FAST_K(8, 8, true, false, true) else OAK_BWD_FAST_K(8, 8, false, true, false) break;
            case 816: if (plain58) OAK_BWD_FAST_K(8, 16, true, false, true) else OAK_BWD_FAST_K(8, 16, false, true, false) break;
              case 832: OAK_BWD_FAST_K(8, 32, false, true, false) break;
        }
    } else
    switch (template_depth(R)) {      // depths 9..32: the next larger instantiation, zero weights above R
        OAK_BWD_CASE(0) OAK_BWD_CASE(1) OAK_BWD_CASE(2) OAK_BWD_CASE(3) OAK_BWD_CASE(4)
        OAK_BWD_CASE(5) OAK_BWD_CASE(6) OAK_BWD_CASE(7) OAK_BWD_CASE(8) OAK_BWD_CASE(12) OAK_BWD_CASE(16)
        case 24: OAK_BWD_LAUNCH(24, 1) break;
        case 32: OAK_BWD_LAUNCH(32, 1) break;
        default: set_error("gram_bwd: unsupported depth %d", R); return OAK_E_ARG;
    }
#undef OAK_BWD_CASE
#undef OAK_BWD_LAUNCH
#undef OAK_BWD_FAST
#undef OAK_BWD_FAST_K
#undef OAK_BWD_FAST_U
#undef OAK_BWD_FAST_G
    OAK_HIP_CHECK(hipGetLastError());
    reduce_records_kernel<<<(unsigned)reclen, 256, 0, ctx->stream>>>(d_part, nrb * ncb, reclen, d_rec);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// stride of the inducing-input gradient buffer: the fast kernels' padded dim count, or one slot per column a sub-kernel reads
bool gram_bwd_z_fast(const PreparedKernel& pk) {
    const int D = pk.dd.D, R = pk.dd.R;
    return !pk.grouped && R >= 1 && D <= 32 && (R <= 4 || (R <= 8 && D <= 16)) && getenv("OAK_BWDZ_GENERAL") == nullptr;
}
int gram_bwd_z_stride(const PreparedKernel& pk) {
    if (gram_bwd_z_fast(pk)) return pk.dd.D <= 8 ? 8 : (pk.dd.D <= 16 ? 16 : 32);
    return pk.dd.D + (pk.grouped ? (int)pk.extra_cols.size() : 0);
}

static int gram_bwd_z_general(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, const double* d_dzb,
                              const double* d_G, int64_t ldg, double g_scale, const double* d_yA, const double* d_avec, double* d_gz) {
    const int D = pk.dd.D, R = pk.dd.R;
    if (R < 1) return OAK_OK;                                   // a constant kernel does not depend on the inducing inputs
    const int nx = pk.grouped ? A.nx : 0, zd = D + nx;
    if (pk.grouped) OAK_REQUIRE(A.xx != nullptr && B.xx != nullptr && A.nx == B.nx, "gram_bwd_z: features lack the grouped sub-kernels' further columns");
    int nw = 4;
    auto lds_for = [&](int w) { return sizeof(double) * ((size_t)(3 * D + nx) * 64 + 64 + (size_t)w * zd * 64); };
    while (nw > 1 && lds_for(nw) > 160 * 1024) nw >>= 1;
    const size_t lds = lds_for(nw);
    OAK_REQUIRE(lds <= 160 * 1024, "gradient w.r.t. inducing inputs: %d sub-kernels with %d further columns need %zu bytes of LDS", D, nx, lds);
    const int64_t nb = B.n, ncb = (nb + 63) / 64;
    int64_t nrb = ((int64_t)ctx->num_cu * 4 + ncb - 1) / ncb;
    int64_t rows = (na + nrb - 1) / nrb;
    if (rows < 4) rows = 4;
    nrb = (na + rows - 1) / rows;
    if (nrb > 65535) { rows = (na + 65534) / 65535; nrb = (na + rows - 1) / rows; }
    double* d_part = nullptr;
    OAK_CHECK(get_buf_t(ctx, "bwdz_part", (size_t)nrb * nb * zd, &d_part));
    dim3 grid((unsigned)ncb, (unsigned)nrb);
#define OAK_BZG(RR) case RR: {                                                                                                        \
        auto kern = gram_bwd_z_general_kernel<RR>;                                                                                        \
        if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)kern));                                                        \
        kern<<<grid, 64 * nw, lds, ctx->stream>>>(pk.dd, pk.d_tables, A.xs, A.cn, A.xx, A.ld, a0, na, B.xs, B.cn, d_dzb, B.xx, B.ld, nb,  \
                                                  d_G, ldg, d_yA, d_avec, g_scale, (int)rows, nx, d_part);                                \
    } break;
    switch (template_depth(R)) {
        OAK_BZG(1) OAK_BZG(2) OAK_BZG(3) OAK_BZG(4) OAK_BZG(5) OAK_BZG(6) OAK_BZG(7) OAK_BZG(8) OAK_BZG(12) OAK_BZG(16) OAK_BZG(24) OAK_BZG(32)
        default: set_error("gram_bwd_z: unsupported depth %d", R); return OAK_E_ARG;
    }
#undef OAK_BZG
    OAK_HIP_CHECK(hipGetLastError());
    const int64_t len = nb * zd;
    reduce_gz_kernel<<<(unsigned)((len + 255) / 256), 256, 0, ctx->stream>>>(d_part, nrb, len, d_gz);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// d_gz[nb][dmax] += column-side contraction of G (+ optional rank-1 yA avec^T) with dK/dz over the pairs (A rows a0.., B);
// d_dzb = featurize_dx of the B points, dmax = gram_bwd_z_stride(pk).  Depth <= 4 with <= 32 dims and depth <= 8 with <= 16 take the
// register-resident pair walk; everything else (deeper, wider, grouped sub-kernels) the general kernel above.
int gram_bwd_z(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, const double* d_dzb,
               const double* d_G, int64_t ldg, double g_scale, const double* d_yA, const double* d_avec, double* d_gz, int dmax) {
    if (na <= 0 || B.n <= 0) return OAK_OK;
    const int D = pk.dd.D, R = pk.dd.R;
    OAK_REQUIRE(dmax == gram_bwd_z_stride(pk), "gram_bwd_z: buffer stride %d does not match the kernel description", dmax);
    if (!gram_bwd_z_fast(pk)) return gram_bwd_z_general(ctx, pk, A, a0, na, B, d_dzb, d_G, ldg, g_scale, d_yA, d_avec, d_gz);
    OAK_REQUIRE(A.xs32 != nullptr && B.xs32 != nullptr, "gram_bwd_z: features were not prepared for the backward pass");
    bool allrbf = true, unitbv = true;
    for (int d = 0; d < D; ++d) {
        allrbf = allrbf && pk.dd.type[d] == OAK_DIM_RBF;
        unitbv = unitbv && (pk.dd.type[d] != OAK_DIM_RBF || pk.dd.bv[d] == 1.0);
    }
    const int cpt = dmax <= 16 ? 2 : 1;
    const int TJ = 64 * cpt, RS = 8;
    const size_t lds = sizeof(double) * ((size_t)3 * dmax * TJ + TJ + EW_N + (unitbv ? 0 : 2 * dmax) + (allrbf ? 0 : dmax));
    const int64_t nb = B.n;
    const int64_t ncb = (nb + TJ - 1) / TJ;
    int64_t nrb = ((int64_t)ctx->num_cu * 8 + ncb - 1) / ncb;
    int64_t rows = (na + nrb - 1) / nrb;
    rows = ((rows + RS - 1) / RS) * RS;
    if (rows < RS) rows = RS;
    if (rows > 4096) rows = 4096;
    nrb = (na + rows - 1) / rows;
    if (nrb > 65535) { rows = (((na + 65534) / 65535 + RS - 1) / RS) * RS; nrb = (na + rows - 1) / rows; }
    double *d_pack = nullptr, *d_part = nullptr;
    OAK_CHECK(get_buf_t(ctx, "bwd_pack", (size_t)na * 3 * dmax, &d_pack));
    OAK_CHECK(get_buf_t(ctx, "bwdz_part", (size_t)nrb * nb * dmax, &d_part));
    OAK_CHECK(launch_pack_rows(ctx, A, a0, na, D, dmax, d_pack, 1));
    dim3 grid((unsigned)ncb, (unsigned)nrb);
#define OAK_BZ_K(RR, DM, AR, UB)                                                                                                  \
    gram_bwd_z_kernel<RR, DM, (DM <= 16 ? 2 : 1), AR, UB><<<grid, 256, lds, ctx->stream>>>(pk.dd, pk.d_tables, d_pack, a0, na,   \
        B.xs32, B.cn, d_dzb, B.ld, nb, d_G, ldg, d_yA, d_avec, g_scale, (int)rows, d_part);
#define OAK_BZ_U(RR, DM, AR) { if (unitbv) OAK_BZ_K(RR, DM, AR, true) else OAK_BZ_K(RR, DM, AR, false) }
#define OAK_BZ(RR, DM) { if (allrbf) OAK_BZ_U(RR, DM, true) else OAK_BZ_U(RR, DM, false) }
    switch (R * 100 + dmax) {
        case 108: OAK_BZ(1, 8) break;   case 116: OAK_BZ(1, 16) break;   case 132: OAK_BZ(1, 32) break;
        case 208: OAK_BZ(2, 8) break;   case 216: OAK_BZ(2, 16) break;   case 232: OAK_BZ(2, 32) break;
        case 308: OAK_BZ(3, 8) break;   case 316: OAK_BZ(3, 16) break;   case 332: OAK_BZ(3, 32) break;
        case 408: OAK_BZ(4, 8) break;   case 416: OAK_BZ(4, 16) break;   case 432: OAK_BZ(4, 32) break;
        case 508: OAK_BZ(5, 8) break;   case 516: OAK_BZ(5, 16) break;
        case 608: OAK_BZ(6, 8) break;   case 616: OAK_BZ(6, 16) break;
        case 708: OAK_BZ(7, 8) break;   case 716: OAK_BZ(7, 16) break;
        case 808: OAK_BZ(8, 8) break;   case 816: OAK_BZ(8, 16) break;
        default: set_error("gram_bwd_z: unsupported configuration"); return OAK_E_ARG;
    }
#undef OAK_BZ
#undef OAK_BZ_U
#undef OAK_BZ_K
    OAK_HIP_CHECK(hipGetLastError());
    const int64_t len = nb * dmax;
    reduce_gz_kernel<<<(unsigned)((len + 255) / 256), 256, 0, ctx->stream>>>(d_part, nrb, len, d_gz);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

int diag_bwd(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, double gconst, double* d_rec, const double* d_gvec) {
    if (A.n <= 0) return OAK_OK;
    const int D = pk.dd.D, R = pk.dd.R;
    const int tablen = (int)pk.tables.size();
    const int64_t reclen = record_len(pk);
    int64_t nwg = (A.n + 2047) / 2048;              // one trip of 8 points per lane per workgroup
    if (nwg > 4096) nwg = 4096;
    int64_t rows = (A.n + nwg - 1) / nwg;
    rows = ((rows + 2047) / 2048) * 2048;
    nwg = (A.n + rows - 1) / rows;
    double* d_part = nullptr;
    OAK_CHECK(get_buf_t(ctx, "bwd_part_diag", (size_t)(nwg * reclen), &d_part));
    const int RTP = template_depth(R);                         // template depth the launch below instantiates
    const size_t lds = sizeof(double) * ((size_t)8 * D + 4 * tablen + 4 * (RTP + 1) + 8);
#define OAK_DB_CASE(RR) case RR: diag_bwd_kernel<RR><<<(unsigned)nwg, 256, lds, ctx->stream>>>(pk.dd, pk.d_tables, tablen, A.xs, A.cn, A.dcn, A.ld, A.n, gconst, d_gvec, rows, d_part); break;
    switch (RTP) {
        OAK_DB_CASE(0) OAK_DB_CASE(1) OAK_DB_CASE(2) OAK_DB_CASE(3) OAK_DB_CASE(4)
        OAK_DB_CASE(5) OAK_DB_CASE(6) OAK_DB_CASE(7) OAK_DB_CASE(8) OAK_DB_CASE(12) OAK_DB_CASE(16)
        OAK_DB_CASE(24) OAK_DB_CASE(32)
        default: set_error("diag_bwd: unsupported depth %d", R); return OAK_E_ARG;
    }
#undef OAK_DB_CASE
    OAK_HIP_CHECK(hipGetLastError());
    reduce_records_kernel<<<(unsigned)reclen, 256, 0, ctx->stream>>>(d_part, nwg, reclen, d_rec);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// ---- small M x M helpers ---------------------------------------------------------------------------------
__global__ void set_identity_kernel(double* __restrict__ A, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j < n) A[i * n + j] = (i == j) ? 1.0 : 0.0;
}
// H = Kinv - Sinv - a a^T ;  Guu = 0.5 H - 0.5 KWK / s2.  With nx extra outputs (rows of ax): the sum over the outputs,
// H = (1 + nx) (Kinv - Sinv) - sum_p a_p a_p^T ;  Guu = 0.5 H - 0.5 (1 + nx) KWK / s2
__global__ void combine_h_kernel(const double* __restrict__ Kinv, const double* __restrict__ Sinv, const double* __restrict__ a,
                                 const double* __restrict__ KWK, double s2, int64_t n, double* __restrict__ H, double* __restrict__ Guu,
                                 const double* __restrict__ ax, int nx) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= n) return;
    const double pt = (double)(1 + nx);
    double h = pt * (Kinv[i * n + j] - Sinv[i * n + j]) - a[i] * a[j];
    for (int p = 0; p < nx; ++p) h = __builtin_fma(-ax[(int64_t)p * n + i], ax[(int64_t)p * n + j], h);
    H[i * n + j] = h;
    Guu[i * n + j] = 0.5 * h - 0.5 * pt * KWK[i * n + j] / s2;
}
// G[r][m] += sum_p Yx[p][a0 + r] * ax[p][m]: the other outputs' rank-one adjoints y_p a_p^T folded into the adjoint panel
// (output 0's rides in the pair kernel)
__global__ void __launch_bounds__(256) rank_add_kernel(double* __restrict__ G, int64_t ldg, int64_t na, int64_t M, const double* __restrict__ Yx,
                                                       int64_t ldy, const double* __restrict__ ax, int nx) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * 16;
    if (m >= M) return;
    double acc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = 0.0;
    for (int p = 0; p < nx; ++p) {
        const double av = ax[(int64_t)p * M + m];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t r = r0 + u < na ? r0 + u : na - 1;
            acc[u] = __builtin_fma(Yx[(int64_t)p * ldy + r], av, acc[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (r0 + u < na) G[(r0 + u) * ldg + m] += acc[u];
}
// GPR: G = 0.5 (alpha alpha^T - Kinv)
__global__ void combine_gpr_kernel(const double* __restrict__ Kinv, const double* __restrict__ a, int64_t n, double* __restrict__ G) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j < n) G[i * n + j] = 0.5 * (a[i] * a[j] - Kinv[i * n + j]);
}

int set_identity(oak_ctx* ctx, double* dA, int64_t n) {
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
    set_identity_kernel<<<grid, 256, 0, ctx->stream>>>(dA, n);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// scatter a device record [gl | gk | gw | gtab] into the public gradient layout
// [lengthscale (D) | base_var (D) | order_var (n_order_var) | noise | dTable (meas_data_len)]
void scatter_record(const oak_kernel_desc* desc, const PreparedKernel& pk, const std::vector<double>& rec, double dnoise,
                           double* grad_out) {
    const int D = desc->num_dims, R = pk.dd.R;       // the record is laid out for the effective depth min(max_depth, D)
    const int64_t glen = 2 * D + desc->n_order_var + 1 + desc->meas_data_len;
    for (int64_t i = 0; i < glen; ++i) grad_out[i] = 0.0;
    for (int d = 0; d < D; ++d) {
        if (desc->dim_type[d] == OAK_DIM_RBF) grad_out[d] = rec[d] * (1.3862943611198906 / desc->lengthscale[d]);   // kernels work in units of 2 ln2 / l
        grad_out[D + d] = rec[D + d] / desc->base_var[d];        // k_d is linear in its base variance
    }
    if (desc->share_var) for (int r = 0; r <= R; ++r) grad_out[2 * D + r] = rec[2 * D + r];
    else grad_out[2 * D] = rec[2 * D];
    grad_out[2 * D + desc->n_order_var] = dnoise;
    double* gt = grad_out + 2 * D + desc->n_order_var + 1;
    for (int d = 0; d < D; ++d) {
        if (desc->dim_type[d] != OAK_DIM_CATEGORICAL) continue;
        const int C = desc->meas_k[d];
        const int toff = pk.dd.tab_off[d];
        for (int i = 0; i < C * C; ++i) gt[desc->meas_off[d] + i] = rec[2 * D + (R + 1) + toff + i] * desc->base_var[d];
    }
}

}  // namespace oak

using namespace oak;

extern "C" {

int64_t oak_grad_len(const oak_kernel_desc* desc) {
    return desc ? 2 * (int64_t)desc->num_dims + desc->n_order_var + 1 + desc->meas_data_len : 0;
}

int oak_sgpr_elbo_grad(oak_ctx* ctx, const oak_kernel_desc* desc, double noise_var, double jitter, double* elbo_out, double* grad_out) {
    return oak_sgpr_elbo_grad_z(ctx, desc, noise_var, jitter, elbo_out, grad_out, nullptr);
}

int oak_sgpr_elbo_grad_z(oak_ctx* ctx, const oak_kernel_desc* desc, double noise_var, double jitter, double* elbo_out, double* grad_out,
                         double* gradZ_out) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(grad_out != nullptr, "grad_out is NULL");
    OAK_REQUIRE(ctx->have_data && ctx->have_Z, "SGPR: set_data and set_inducing must be called first");
    PreparedKernel pk;
    OAK_CHECK(prepare_kernel(ctx, desc, &pk, PK_GROUPED));
    PhaseTimer ttot(ctx, "total");
    // ---- forward ------------------------------------------------------------------------------------------
    double elbo = 0.0, terms[8];
    ctx->keep_kfu = true;                                   // a whitening forward keeps the raw Kfu panel for the loop below
    ctx->crt_planes_valid = false;
    // Where the forward pass takes the int8 route (crt.hip) and converts the whole panel in one chunk, the adjoint panel Kfu H below
    // is formed from its residue planes on the int8 pipe as well (crt_gemm.hip), and the forward pass need not write the fp64 panel
    // at all.  OAK_CRT_GEMM=0: the fp64 GEMM throughout.
    ctx->grad_int8 = getenv("OAK_CRT_GEMM") == nullptr || atoi(getenv("OAK_CRT_GEMM")) != 0;
    const int frc = sgpr_forward(ctx, pk, noise_var, jitter, &elbo, terms);
    ctx->keep_kfu = false;
    bool planes_here = ctx->crt_planes_valid && ctx->grad_int8 && !ctx->stats_whitened;
    if (planes_here && !(getenv("OAK_CRT_GEMM") != nullptr && atoi(getenv("OAK_CRT_GEMM")) == 1)) {
        // the conditioning rule of the int8 product (sgpr.hip: the forward pass applied it already when the estimate arrived before its Gram
        // launch; in a partitioned pass it arrives later)
        planes_here = ctx->cond_seen && sgpr_cond_estimate_ok_for_int8_gemm(ctx);
    }
    const bool panel_here = ctx->crt_panel_written || !ctx->stats_crt;      // an int8-route forward writes the fp64 panel only on request
    ctx->crt_planes_valid = false;
    ctx->grad_int8 = false;
    ctx->crt_gemm_info[0] = ctx->crt_gemm_info[1] = 0;
    OAK_CHECK(frc);
    const int64_t N = ctx->N, M = ctx->M, Mp = ((M + 127) / 128) * 128;
    const double s2 = noise_var;
    double* dL = (double*)peek_buf(ctx, "L");
    double* dLB = (double*)peek_buf(ctx, "LB");
    double* dW = (double*)peek_buf(ctx, "T2");
    // a = Sigma^-1 psi / sigma^2 = L^-T LB^-T c.  The posterior entry points get it from two transposed triangular solves
    // (sgpr_ensure_alpha); for ONE right-hand side those are 2 x 15 dependent launches, 0.6 ms at M = 1024 -- here the matrix
    // (LB^-1 L^-1)^T is formed anyway (dPT below, it also gives Sigma^-1), so a is one matrix-vector product with it.
    double* da = nullptr;
    OAK_CHECK(get_buf_t(ctx, "g_a", (size_t)M, &da));
    double* dY = (double*)peek_buf(ctx, "Y");
    const int nx = ctx->n_extra;           // extra target columns: the objective is the sum over 1 + nx outputs
    double* d_ax = nullptr;
    // ---- M x M adjoints -----------------------------------------------------------------------------------
    double *dLinvT, *dPT, *dKinv, *dSinv, *dTmp, *dKWK, *dH, *dGuu, *dKuu, *dsc, *dvec;
    {
        PhaseTimer t(ctx, "bwd_tail");
        OAK_CHECK(get_buf_t(ctx, "g_LinvT", (size_t)M * M, &dLinvT));
        OAK_CHECK(get_buf_t(ctx, "g_PT", (size_t)M * M, &dPT));
        OAK_CHECK(get_buf_t(ctx, "g_Kinv", (size_t)M * M, &dKinv));
        OAK_CHECK(get_buf_t(ctx, "g_Sinv", (size_t)M * M, &dSinv));
        OAK_CHECK(get_buf_t(ctx, "g_Tmp", (size_t)M * M, &dTmp));
        OAK_CHECK(get_buf_t(ctx, "g_KWK", (size_t)M * M, &dKWK));
        OAK_CHECK(get_buf_t(ctx, "g_H", (size_t)M * M, &dH));
        OAK_CHECK(get_buf_t(ctx, "g_Guu", (size_t)M * M, &dGuu));
        OAK_CHECK(get_buf_t(ctx, "g_Kuu", (size_t)M * M, &dKuu));
        OAK_CHECK(get_buf_t(ctx, "g_sc", 8, &dsc));
        OAK_CHECK(get_buf_t(ctx, "g_vec", (size_t)M, &dvec));
        if (ctx->have_linv) {
            dLinvT = (double*)peek_buf(ctx, "LinvT");                                 // left by the forward tail
        } else {
            OAK_CHECK(set_identity(ctx, dLinvT, M));
            OAK_CHECK(trsm_rows(ctx, dL, M, M, dLinvT, M, M, 0));                   // rows = columns of L^-1
        }
        OAK_CHECK(copy_d2d(ctx, dPT, dLinvT, sizeof(double) * (size_t)M * M));
        OAK_CHECK(trsm_rows(ctx, dLB, M, M, dPT, M, M, 0));                         // rows = columns of LB^-1 L^-1
        // On the whitened route -- chosen because Kuu is ill-conditioned -- a comes from the two transposed triangular solves, the
        // arithmetic of oak_sgpr_alpha / predict: multiplying by the explicit (LB^-1 L^-1)^T loses ~cond(Kuu) ulps there (the forward
        // tail records 8e-9 on such a problem), and the gradient's a would no longer be the posterior's.  0.6 ms of a ~95 ms evaluation.
        const bool exact_a = ctx->stats_whitened;
        if (ctx->have_alpha) OAK_CHECK(copy_d2d(ctx, da, peek_buf(ctx, "alpha"), sizeof(double) * (size_t)M));
        else if (exact_a) { OAK_CHECK(sgpr_ensure_alpha(ctx)); OAK_CHECK(copy_d2d(ctx, da, peek_buf(ctx, "alpha"), sizeof(double) * (size_t)M)); }
        else OAK_CHECK(gemv_rows(ctx, dPT, M, M, M, (const double*)peek_buf(ctx, "c"), da));         // a = (LB^-1 L^-1)^T c
        // LinvT (rows = columns of L^-1) and PT are upper triangular, L lower: the products below skip the zero k ranges and
        // slice k over gridDim.z (gemm_tail)
        const int UU = OAK_TRI_A_UPPER | OAK_TRI_B_UPPER;
        OAK_CHECK(gemm_tail(ctx, 1, dLinvT, dLinvT, dKinv, M, M, M, M, M, M, 1.0, 0.0, UU));                 // Kuu^-1
        OAK_CHECK(gemm_tail(ctx, 1, dPT, dPT, dSinv, M, M, M, M, M, M, 1.0, 0.0, UU));                       // Sigma^-1
        // W is symmetric up to rounding (W_ij and W_ji are the same sum in a different order), so L^-T W is taken as the NT
        // product L^-T W^T, the form the 128 x 128 kernel computes
        OAK_CHECK(gemm_tail(ctx, 1, dLinvT, dW, dTmp, M, M, M, M, M, M, 1.0, 0.0, OAK_TRI_A_UPPER));         // L^-T W
        OAK_CHECK(gemm_tail(ctx, 1, dTmp, dLinvT, dKWK, M, M, M, M, M, M, 1.0, 0.0, OAK_TRI_B_UPPER));       // Kuu^-1 Phi Kuu^-1
        dim3 grid((unsigned)((M + 255) / 256), (unsigned)M);
        if (nx > 0) {                      // a_p = (LB^-1 L^-1)^T c_p for the other outputs: one product with the matrix at hand
            OAK_CHECK(get_buf_t(ctx, "g_ax", (size_t)nx * M, &d_ax));
            const double* d_cx = (const double*)peek_buf(ctx, "c_all") + M;
            if (exact_a) {
                OAK_CHECK(copy_d2d(ctx, d_ax, d_cx, sizeof(double) * (size_t)nx * M));
                OAK_CHECK(trsm_rows(ctx, dLB, M, M, d_ax, nx, M, 1));
                OAK_CHECK(trsm_rows(ctx, dL, M, M, d_ax, nx, M, 1));
            } else {
                OAK_CHECK(gemm_nt(ctx, d_cx, dPT, d_ax, nx, M, M, M, M, M, 1.0, 0.0, 0));
            }
        }
        combine_h_kernel<<<grid, 256, 0, ctx->stream>>>(dKinv, dSinv, da, dKWK, s2, M, dH, dGuu, d_ax, nx);
        OAK_HIP_CHECK(hipGetLastError());
        // Kuu (+ jitter) = L L^T, for tr(Sigma^-1 Kuu) and a^T Kuu a
        OAK_CHECK(gemm_tail(ctx, 1, dL, dL, dKuu, M, M, M, M, M, M, 1.0, 0.0, OAK_TRI_A_LOWER | OAK_TRI_B_LOWER));
        OAK_CHECK(dot(ctx, dSinv, dKuu, M * M, dsc + 0));                            // tr(Sigma^-1 Kuu)
        OAK_CHECK(gemv_rows(ctx, dKuu, M, M, M, da, dvec));
        OAK_CHECK(dot(ctx, dvec, da, M, dsc + 1));                                   // a^T Kuu a
        if (nx > 0) {                                                                // ... and sum_p a_p^T Kuu a_p
            double* d_kax = nullptr;
            OAK_CHECK(get_buf_t(ctx, "g_kax", (size_t)nx * M, &d_kax));
            OAK_CHECK(gemm_nt(ctx, d_ax, dKuu, d_kax, nx, M, M, M, M, M, 1.0, 0.0, 0));     // Kuu is symmetric
            OAK_CHECK(dot(ctx, d_kax, d_ax, (int64_t)nx * M, dsc + 3));
        }
        t.stop();
    }
    // psi^T a needs the raw psi: in the whitened route stats.psi is still raw (only Phi is replaced by W)
    double* d_stats = (double*)peek_buf(ctx, "stats");
    OAK_CHECK(dot(ctx, d_stats + M * M, da, M, dsc + 2));
    if (nx > 0) OAK_CHECK(dot(ctx, (const double*)peek_buf(ctx, "psix"), d_ax, (int64_t)nx * M, dsc + 4));
    double hs[5] = {0, 0, 0, 0, 0};
    OAK_HIP_CHECK(hipMemcpyAsync(hs, dsc, sizeof(double) * 5, hipMemcpyDeviceToHost, ctx->stream));
    // ---- N-sized contractions ---------------------------------------------------------------------------------
    const int64_t reclen = record_len(pk);
    double* d_rec = nullptr;
    OAK_CHECK(get_buf_t(ctx, "g_rec", (size_t)reclen, &d_rec));
    OAK_CHECK(fill_zero(ctx, d_rec, sizeof(double) * (size_t)reclen));
    Feat FX, FZ;
    if (ctx->feat_grad_valid) {                    // left by this call's forward pass
        FX = ctx->featXg; FZ = ctx->featZg;
        ctx->feat_grad_valid = false;
    } else {
        OAK_CHECK(featurize(ctx, pk, (double*)peek_buf(ctx, "Z"), M, ctx->ldx, "featZg", &FZ, true));
        OAK_CHECK(featurize(ctx, pk, (double*)peek_buf(ctx, "X"), N, ctx->ldx, "featXg", &FX, true));
    }
    // optional: gradient w.r.t. the inducing inputs (second pair pass, column-side accumulators)
    const int zdmax = gram_bwd_z_stride(pk);
    double *d_dzb = nullptr, *d_gz = nullptr;
    if (gradZ_out != nullptr) {
        OAK_CHECK(get_buf_t(ctx, "featZ_dx", (size_t)pk.dd.D * FZ.ld, &d_dzb));
        dim3 gdx((unsigned)((FZ.ld + 255) / 256), (unsigned)pk.dd.D);
        featurize_dx_kernel<<<gdx, 256, 0, ctx->stream>>>(pk.dd, pk.dm, pk.d_meas, (double*)peek_buf(ctx, "Z"), M, ctx->ldx, FZ.ld, d_dzb);
        OAK_HIP_CHECK(hipGetLastError());
        OAK_CHECK(get_buf_t(ctx, "g_gz", (size_t)M * zdmax, &d_gz));
        OAK_CHECK(fill_zero(ctx, d_gz, sizeof(double) * (size_t)M * zdmax));
    }
    // a / s2 as the rank-1 partner of y
    double* d_as2 = nullptr;
    OAK_CHECK(get_buf_t(ctx, "g_as2", (size_t)M, &d_as2));
    OAK_CHECK(copy_d2d(ctx, d_as2, da, sizeof(double) * (size_t)M));
    OAK_CHECK(scale_vec(ctx, 1.0 / s2, d_as2, M));
    int64_t rows = ctx->panel_rows > 0 ? ctx->panel_rows : (int64_t)(((size_t)16 << 30) / (sizeof(double) * (size_t)Mp));
    if (rows > N) rows = N;
    if (rows < 16) rows = 16;
    double *dPanel = nullptr, *dG = nullptr;
    OAK_CHECK(get_buf_t(ctx, "panel", (size_t)rows * Mp, &dPanel));
    OAK_CHECK(get_buf_t(ctx, "gpanel", (size_t)rows * Mp, &dG));
    const bool reuse_panel = (rows >= N) && panel_here && (!ctx->stats_whitened || ctx->kfu_kept);    // forward left the raw Kfu panel in place
    for (int64_t a0 = 0; a0 < N; a0 += rows) {
        const int64_t na = (a0 + rows <= N) ? rows : N - a0;
        // int8 route: the adjoint panel from the residue planes of the forward pass (crt_gemm.hip)
        bool done = false;
        if (planes_here && rows >= N) {
            PhaseTimer t(ctx, "bwd_gemm");
            done = crt_gemm_adjoint(ctx, ctx->crt_pl, M, na, dH, dG, Mp) == OAK_OK;
            if (!done) (void)hipGetLastError();          // (no room for H' next to the forward's moduli: the fp64 product below)
            else if (nx > 0) {                          // the other outputs' y_p a_p^T: a pass of its own over the adjoint panel
                rank_add_kernel<<<dim3((unsigned)((M + 255) / 256), (unsigned)((na + 15) / 16)), 256, 0, ctx->stream>>>(
                    dG, Mp, na, M, (const double*)peek_buf(ctx, "Yx") + a0, N, d_ax, nx);
                OAK_HIP_CHECK(hipGetLastError());
            }
            t.stop();
        }
        if (!reuse_panel && !done) {
            PhaseTimer t(ctx, "gram");
            OAK_CHECK(gram(ctx, pk, FX, a0, na, FZ, dPanel, Mp, nullptr, nullptr, Mp));
            t.stop();
        }
        if (!done) {
            PhaseTimer t(ctx, "bwd_gemm");      // Gfu = Kfu H   (scaled by 1/s2 inside the pair kernel)
            int rs = OAK_OK;
            if (nx > 0 && gemm_nt_rankp(ctx, dPanel, dH, dG, na, M, M, Mp, M, Mp, (const double*)peek_buf(ctx, "Yx") + a0, N, d_ax, nx, &rs)) {
                OAK_CHECK(rs);                  // the other outputs' y_p a_p^T went in with the GEMM's epilogue
            } else {
                OAK_CHECK(gemm_nt(ctx, dPanel, dH, dG, na, M, M, Mp, M, Mp, 1.0, 0.0, 0));
                if (nx > 0) {                   // small shapes (the 64 x 64 GEMM): a pass of its own over the adjoint panel
                    rank_add_kernel<<<dim3((unsigned)((M + 255) / 256), (unsigned)((na + 15) / 16)), 256, 0, ctx->stream>>>(
                        dG, Mp, na, M, (const double*)peek_buf(ctx, "Yx") + a0, N, d_ax, nx);
                    OAK_HIP_CHECK(hipGetLastError());
                }
            }
            t.stop();
        }
        {
            PhaseTimer t(ctx, "bwd_gram");
            OAK_CHECK(gram_bwd(ctx, pk, FX, a0, na, FZ, dG, Mp, 1.0 / s2, dY, d_as2, d_rec, desc->grad_base_var != 0));
            t.stop();
        }
        if (gradZ_out != nullptr) {
            PhaseTimer t(ctx, "bwd_z");
            OAK_CHECK(gram_bwd_z(ctx, pk, FX, a0, na, FZ, d_dzb, dG, Mp, 1.0 / s2, dY, d_as2, d_gz, zdmax));
            t.stop();
        }
    }
    {
        PhaseTimer t(ctx, "bwd_small");
        // <G_uu, dKuu> is replicated on every rank; each contributes 1/nranks so the all-reduce below restores it once
        OAK_CHECK(gram_bwd(ctx, pk, FZ, 0, M, FZ, dGuu, M, 1.0 / (double)(ctx->comm ? ctx->nranks : 1), nullptr, nullptr, d_rec, desc->grad_base_var != 0));
        OAK_CHECK(diag_bwd(ctx, pk, FX, -0.5 * (double)(1 + nx) / s2, d_rec));                   // -1/(2 s2) sum dKdiag, once per output
        t.stop();
    }
    if (gradZ_out != nullptr) {
        // <G_uu, dKuu/dz_m>: Kuu depends on z_m through its row AND its column m; G_uu and Kuu are symmetric, so the total is
        // twice the column-side sum (replicated on every rank, hence the 1/nranks before the all-reduce)
        OAK_CHECK(gram_bwd_z(ctx, pk, FZ, 0, M, FZ, d_dzb, dGuu, M, 2.0 / (double)(ctx->comm ? ctx->nranks : 1), nullptr, nullptr, d_gz, zdmax));
        if (ctx->comm != nullptr) OAK_CHECK(comm_allreduce_dev(ctx, d_gz, M * zdmax));
    }
    if (ctx->comm != nullptr) OAK_CHECK(comm_allreduce_dev(ctx, d_rec, reclen));
    std::vector<double> rec((size_t)reclen);
    OAK_HIP_CHECK(hipMemcpyAsync(rec.data(), d_rec, sizeof(double) * (size_t)reclen, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ttot.stop();
    const double n_tot = terms[5], kappa = terms[3], yy = terms[4], trW = terms[2] * s2;
    const double trSK = hs[0], aKa = hs[1] + (nx > 0 ? hs[3] : 0.0), psia = hs[2] + (nx > 0 ? hs[4] : 0.0), pt = (double)(1 + nx);
    double yy_all = yy;
    if (nx > 0) {                           // y_p^T y_p of the other outputs (summed over the ranks with their Kuf y)
        std::vector<double> yyx((size_t)nx);
        OAK_CHECK(copy_sync(ctx, yyx.data(), (const double*)peek_buf(ctx, "psix") + (int64_t)nx * M, sizeof(double) * (size_t)nx, hipMemcpyDeviceToHost));
        for (double v : yyx) yy_all += v;
    }
    const double dnoise = pt * (-0.5 * n_tot / s2 + 0.5 * kappa / (s2 * s2) - 0.5 * trW / (s2 * s2) + 0.5 * ((double)M - trSK) / s2) +
                          0.5 * yy_all / (s2 * s2) - psia / (s2 * s2) + 0.5 * (psia - s2 * aKa) / (s2 * s2);
    scatter_record(desc, pk, rec, dnoise, grad_out);
    if (gradZ_out != nullptr) {
        std::vector<double> gz((size_t)M * zdmax);
        OAK_CHECK(copy_sync(ctx, gz.data(), d_gz, sizeof(double) * gz.size(), hipMemcpyDeviceToHost));
        const int32_t ldz = ctx->ldx;
        for (int64_t i = 0; i < M * (int64_t)ldz; ++i) gradZ_out[i] = 0.0;
        for (int d = 0; d < pk.dd.D; ++d) {
            if (pk.dd.type[d] != OAK_DIM_RBF) continue;                       // discrete inputs have no derivative
            const double cd = 64.0 * 0.6931471805599453094 * pk.dd.scale[d];  // divided out in featurize_dx_kernel / the pair kernel
            for (int64_t m = 0; m < M; ++m) gradZ_out[m * ldz + pk.dd.col[d]] += cd * gz[(size_t)m * zdmax + d];
            for (int q = pk.dd.xrow[d]; q < pk.dd.xrow[d] + pk.dd.nxc[d]; ++q)          // a group's further columns (general kernel: slots D + q)
                for (int64_t m = 0; m < M; ++m) gradZ_out[m * ldz + pk.extra_cols[q]] += cd * gz[(size_t)m * zdmax + pk.dd.D + q];
        }
    }
    if (elbo_out) *elbo_out = elbo;
    return OAK_OK;
}

int oak_gpr_log_marginal_grad(oak_ctx* ctx, const oak_kernel_desc* desc, double noise_var, double* out, double* grad_out) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(grad_out != nullptr, "grad_out is NULL");
    double logml = 0.0;
    OAK_CHECK(oak_gpr_log_marginal(ctx, desc, noise_var, &logml));
    PreparedKernel pk;
    OAK_CHECK(prepare_kernel(ctx, desc, &pk, PK_GROUPED));
    const int64_t N = ctx->gN;
    double* dL = (double*)peek_buf(ctx, "gprL");
    double* dalpha = (double*)peek_buf(ctx, "gpralpha");
    double *dLinvT, *dKinv, *dG, *dsc;
    OAK_CHECK(get_buf_t(ctx, "g_LinvT", (size_t)N * N, &dLinvT));
    OAK_CHECK(get_buf_t(ctx, "g_Kinv", (size_t)N * N, &dKinv));
    OAK_CHECK(get_buf_t(ctx, "g_Guu", (size_t)N * N, &dG));
    OAK_CHECK(get_buf_t(ctx, "g_sc", 8, &dsc));
    OAK_CHECK(set_identity(ctx, dLinvT, N));
    OAK_CHECK(trsm_rows(ctx, dL, N, N, dLinvT, N, N, 0));
    OAK_CHECK(gemm_nt(ctx, dLinvT, dLinvT, dKinv, N, N, N, N, N, N, 1.0, 0.0, 0));        // (K + s2 I)^-1
    dim3 grid((unsigned)((N + 255) / 256), (unsigned)N);
    combine_gpr_kernel<<<grid, 256, 0, ctx->stream>>>(dKinv, dalpha, N, dG);              // dF/dK = 0.5 (alpha alpha^T - K^-1)
    OAK_HIP_CHECK(hipGetLastError());
    OAK_CHECK(reduce_sum(ctx, dG, N, dsc, 0, N + 1));                                     // dF/ds2 = tr(dF/dK)
    const int64_t reclen = record_len(pk);
    double* d_rec = nullptr;
    OAK_CHECK(get_buf_t(ctx, "g_rec", (size_t)reclen, &d_rec));
    OAK_CHECK(fill_zero(ctx, d_rec, sizeof(double) * (size_t)reclen));
    Feat FX;
    OAK_CHECK(featurize(ctx, pk, (double*)peek_buf(ctx, "gprX"), N, ctx->gldx, "featXg", &FX, true));
    OAK_CHECK(gram_bwd(ctx, pk, FX, 0, N, FX, dG, N, 1.0, nullptr, nullptr, d_rec));
    std::vector<double> rec((size_t)reclen);
    double dnoise = 0.0;
    OAK_HIP_CHECK(hipMemcpyAsync(rec.data(), d_rec, sizeof(double) * (size_t)reclen, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipMemcpyAsync(&dnoise, dsc, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    scatter_record(desc, pk, rec, dnoise, grad_out);
    if (out) *out = logml;
    return OAK_OK;
}

}  // extern "C"

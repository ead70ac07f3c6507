// Double-double M x M products for the whitening step W = L^-1 Phi L^-T of the phi route (oak/utils.py:187-190 forms A = L^-1 Kuf
// and A A^T instead; GPflow's op order).
//
// Why: in fp64 the two products with the explicit L^-1 lose cond(Kuu) eps -- |L^-1| |Phi| |L^-T| is cond(Kuu) times larger than W --
// and so does rounding Phi itself to one double (an error of eps |Phi| is not of the form Kuf^T (delta) Kuf and is amplified the same
// way).  That is why the auto route sent ill-conditioned evaluations through the N-sized triangular solve (M^2 N flops more, twice the
// forward time).  The int8 route (crt.hip) delivers Phi EXACTLY, as an integer, i.e. as a double-double with nothing lost; with the two
// M^3 products carried in double-double arithmetic as well (error-free products by FMA, two-sum accumulation: ~11 flops per
// multiply-add, 0.5 ms at M = 1024) the phi route's W has the whitened route's accuracy at any conditioning measured -- restated on
// the CPU before it was built: at cond(Kuu) = 2e8, tr W 5e-14 and log det B 1.1e-12 from the exact values with either route, against
// 2e-9 / 3e-7 for the fp64 phi route (what remains is the fp64 Cholesky factor both share).  L^-1 stays the fp64 inverse factor.
#include "oak_internal.h"

namespace oak {

struct dd_t { double h, l; };

// (c) += a * (bh + bl), a and the product error exact; the low word collects the small terms, renormalised by the caller
__device__ __forceinline__ void dd_fma(dd_t& c, double a, double bh, double bl) {
    const double p = a * bh;
    double e = __builtin_fma(a, bh, -p);
    e = __builtin_fma(a, bl, e);
    const double s = c.h + p, t = s - c.h;
    const double err = (c.h - (s - t)) + (p - t);
    c.l += err + e;
    c.h = s;
}
__device__ __forceinline__ void dd_renorm(dd_t& c) {
    const double s = c.h + c.l;
    c.l = c.l - (s - c.h);
    c.h = s;
}

// C (double-double, or its high word only when Cl == NULL) = A B over the k range where the triangular operand is non-zero:
//   TRI = 1:  A is lower triangular (k <= i);            B = (Bh + Bl)[k][j], row-major, double-double
//   TRI = 2:  B = Bm^T with Bm lower triangular, i.e. B[k][j] = Bm[j][k], non-zero for k <= j;   A = (Ah + Al)[i][k] double-double
// 32 x 32 outputs per workgroup, 2 x 2 per thread, k in chunks of 32 through LDS.  Workgroups are ordered longest k range first.
template <int TRI>
__global__ void __launch_bounds__(256) ddgemm_tri_kernel(const double* __restrict__ Ah, const double* __restrict__ Al, const double* __restrict__ Bh,
                                                         const double* __restrict__ Bl, int64_t M, double* __restrict__ Ch, double* __restrict__ Cl) {
    __shared__ double As[2][32][33], Bs[2][32][33];          // [hi / lo][row][k] and [hi / lo][k][col]
    const int nb = (int)((M + 31) / 32);
    // TRI 1: the k range grows with the row block; TRI 2: with the column block -- that block index runs from the last one down
    const int slow = nb - 1 - (int)blockIdx.y, fast = (int)blockIdx.x;
    const int bi = TRI == 1 ? slow : fast, bj = TRI == 1 ? fast : slow;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t kend = (int64_t)((TRI == 1 ? bi : bj) + 1) * 32 < M ? (int64_t)((TRI == 1 ? bi : bj) + 1) * 32 : M;
    dd_t c[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) c[u][v] = dd_t{0.0, 0.0};
    for (int64_t k0 = 0; k0 < kend; k0 += 32) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
            const int r = idx >> 5, q = idx & 31;
            const int64_t gi = (int64_t)bi * 32 + r, gk = k0 + q;
            const bool ok = gi < M && gk < M;
            As[0][r][q] = ok ? Ah[gi * M + gk] : 0.0;
            As[1][r][q] = (ok && Al != nullptr) ? Al[gi * M + gk] : 0.0;
            // B tile [k][col]
            const int64_t gk2 = k0 + r, gj = (int64_t)bj * 32 + q;
            const bool ok2 = gk2 < M && gj < M;
            if (TRI == 1) {
                Bs[0][r][q] = ok2 ? Bh[gk2 * M + gj] : 0.0;
                Bs[1][r][q] = (ok2 && Bl != nullptr) ? Bl[gk2 * M + gj] : 0.0;
            } else {
                // B[k][j] = Bm[j][k]: read Bm row-wise (coalesced over k), store transposed
                const int64_t gj2 = (int64_t)bj * 32 + r, gk3 = k0 + q;
                const bool ok3 = gj2 < M && gk3 < M;
                Bs[0][q][r] = ok3 ? Bh[gj2 * M + gk3] : 0.0;
                Bs[1][q][r] = 0.0;
            }
        }
        __syncthreads();
        const int kk_end = (kend - k0 < 32) ? (int)(kend - k0) : 32;
        for (int kk = 0; kk < kk_end; ++kk) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const double ah = As[0][ty * 2 + u][kk], al = As[1][ty * 2 + u][kk];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const double bh = Bs[0][kk][tx * 2 + v], bl = Bs[1][kk][tx * 2 + v];
                    if (TRI == 1) dd_fma(c[u][v], ah, bh, bl);          // a double, b double-double
                    else dd_fma(c[u][v], bh, ah, al);                   // b double, a double-double
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) dd_renorm(c[u][v]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int64_t gi = (int64_t)bi * 32 + ty * 2 + u, gj = (int64_t)bj * 32 + tx * 2 + v;
            if (gi < M && gj < M) {
                Ch[gi * M + gj] = c[u][v].h;
                if (Cl != nullptr) Cl[gi * M + gj] = c[u][v].l;
            }
        }
}

// out[i] = sum_{k <= i} Linv[i][k] x[k] in double-double, rounded to one double: one wave per row, fixed reduction tree
__global__ void __launch_bounds__(256) ddgemv_lower_kernel(const double* __restrict__ Linv, int64_t M, const double* __restrict__ x, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= M) return;
    dd_t c{0.0, 0.0};
    for (int64_t k = lane; k <= i; k += 64) dd_fma(c, Linv[i * M + k], x[k], 0.0);
    dd_renorm(c);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oh = __shfl_xor(c.h, off), ol = __shfl_xor(c.l, off);
        const double s = c.h + oh, t = s - c.h;
        const double err = (c.h - (s - t)) + (oh - t);
        c.l += err + ol;
        c.h = s;
        dd_renorm(c);
    }
    if (lane == 0) out[i] = c.h + c.l;
}

// [W ; (L^-1 psi)^T ; (L^-1 psi_p)^T ..] from the double-double Phi: d_out = (M + 1 + nx) x M array (rows 0 .. M-1 = W, row M = L^-1 psi,
// rows M + 1 .. the further output columns'), all one double
int dd_whiten(oak_ctx* ctx, const double* d_Linv, const double* d_phi_hi, const double* d_phi_lo, const double* d_psi, int64_t M, double* d_out,
              const double* d_psix, int nx) {
    double *d_th = nullptr, *d_tl = nullptr;
    OAK_CHECK(get_buf_t(ctx, "ddT_hi", (size_t)M * M, &d_th));
    OAK_CHECK(get_buf_t(ctx, "ddT_lo", (size_t)M * M, &d_tl));
    const unsigned nb = (unsigned)((M + 31) / 32);
    ddgemm_tri_kernel<1><<<dim3(nb, nb), 256, 0, ctx->stream>>>(d_Linv, nullptr, d_phi_hi, d_phi_lo, M, d_th, d_tl);       // T = L^-1 Phi
    OAK_HIP_CHECK(hipGetLastError());
    ddgemm_tri_kernel<2><<<dim3(nb, nb), 256, 0, ctx->stream>>>(d_th, d_tl, d_Linv, nullptr, M, d_out, nullptr);            // W = T L^-T
    OAK_HIP_CHECK(hipGetLastError());
    ddgemv_lower_kernel<<<(unsigned)((M + 3) / 4), 256, 0, ctx->stream>>>(d_Linv, M, d_psi, d_out + M * M);
    OAK_HIP_CHECK(hipGetLastError());
    for (int p = 0; p < nx; ++p) {          // further output columns: rows M + 1 .. = (L^-1 psi_p)^T
        ddgemv_lower_kernel<<<(unsigned)((M + 3) / 4), 256, 0, ctx->stream>>>(d_Linv, M, d_psix + (int64_t)p * M, d_out + (M + 1 + p) * M);
        OAK_HIP_CHECK(hipGetLastError());
    }
    return OAK_OK;
}

// ---- exact sum of the ranks' Phi through an fp64 all-reduce ------------------------------------------------------------------------------
// |Phi_total[a][b]| <= N_global bound_a bound_b < 2^E, E = e_a + e_b + en: every rank rounds its Phi[a][b] (a double-double when the int8
// route formed it, else one double) to the grid 2^(E - 51) -- the HIGH limb, at most 51 bits and a sign -- and the remainder to the grid
// 2^(E - 51 - lo_bits) -- the LOW limb.  Sums of the high limbs over the ranks stay below 2^52 grid units (the bound is on the total) and sums
// of the low limbs below 2^52 with lo_bits = 50 - ceil(log2 ranks): both all-reduces are exact, in any order.  Rounding by the
// add-and-subtract constant 1.5 * 2^(grid + 52).  What is dropped is below 2^(E - 51 - lo_bits): 98 bits under the bound at eight ranks.
__global__ void __launch_bounds__(256) dd_split_kernel(double* __restrict__ phi, const double* __restrict__ phi_lo, const int* __restrict__ eexp, int en,
                                                       int lo_bits, int64_t M, double* __restrict__ lo_out) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x, a = blockIdx.y;
    if (b >= M) return;
    const int E = eexp[a] + eexp[b] + en;
    const double ch = ldexp(1.5, E - 51 + 52), cl = ldexp(1.5, E - 51 - lo_bits + 52);
    const double xh = phi[a * M + b], xl = phi_lo != nullptr ? phi_lo[a * M + b] : 0.0;
    const double hi = (xh + ch) - ch;
    const double rem = (xh - hi) + xl;
    phi[a * M + b] = hi;
    lo_out[a * M + b] = (rem + cl) - cl;
}
__global__ void __launch_bounds__(256) dd_join_kernel(double* __restrict__ phi, const double* __restrict__ lo, int64_t n, double* __restrict__ phi_lo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double h = phi[i], l = lo[i];
    const double s = h + l, t = s - h;
    phi[i] = s;
    phi_lo[i] = (h - (s - t)) + (l - t);
}
int dd_exchange_split(oak_ctx* ctx, double* d_phi, const double* d_phi_lo, const int* d_eexp, int en, int lo_bits, int64_t M, double* d_lo) {
    dd_split_kernel<<<dim3((unsigned)((M + 255) / 256), (unsigned)M), 256, 0, ctx->stream>>>(d_phi, d_phi_lo, d_eexp, en, lo_bits, M, d_lo);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}
int dd_exchange_join(oak_ctx* ctx, double* d_phi, const double* d_lo, int64_t M, double* d_phi_lo) {
    dd_join_kernel<<<(unsigned)((M * M + 255) / 256), 256, 0, ctx->stream>>>(d_phi, d_lo, M * M, d_phi_lo);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

}  // namespace oak

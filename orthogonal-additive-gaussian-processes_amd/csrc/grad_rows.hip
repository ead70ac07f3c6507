// Backward pair kernel, ROWS IN LANES (r05): the reverse pass of oak/model_utils.py:168-173 through Kuf for the reference's default
// model -- all-continuous sub-kernels, unit base variances, depth <= 4, <= 16 sub-kernels (the headline shape).
//
// gram_bwd_fast_kernel (grad.hip) puts a COLUMN (inducing point) in every lane: the row's features are wave-uniform scalar loads,
// the column's come from LDS -- three 8-byte LDS reads per pair-dimension, ~64 % of a CU's LDS bandwidth by the instruction mix,
// and the kernel sits at 0.66 of the DP issue rate.  Here a lane owns a ROW for the whole column range of its workgroup: the row's
// 3 x DMAX features live in registers, the column's are wave-uniform SCALAR loads (packed [column][xs32 | cn | dcs][DMAX]), and
// LDS only carries the exp2 table and the adjoint tile, which a wave transposes for itself (coalesced 128-byte row segments in,
// one conflict-free read per pair out).  Arithmetic per pair-dimension is the fast kernel's, instruction for instruction.
#include "oak_internal.h"
#include "exp2w.h"

namespace oak {

constexpr int BR_GC = 16;                 // adjoint columns per transposed tile (one 128-byte line per row)
constexpr int BR_LD = 65;                 // tile row stride in LDS: [column][row], conflict-free both ways

template <int R, int DMAX>
__global__ void __launch_bounds__(256, 2)
gram_bwd_rows_kernel(const DevDesc dd, const double* __restrict__ Apack, int64_t a0, int64_t na, const double* __restrict__ Bpack, int64_t nb,
                     const double* __restrict__ G, int64_t ldg, const double* __restrict__ yA, const double* __restrict__ avec,
                     double g_scale, int cols_per_wg, double* __restrict__ partial) {
    // Apack: [na][3][DMAX] = (xs32 | cn | dcs) of rows a0.., padding dims (-1, 0, 0); Bpack: [nb][DMAX / 4][3][4] -- a column's
    // features chunk by chunk (12 contiguous doubles per chunk of four dimensions), padding dims (+1, 0, 0)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Tab = smem;                                   // [EW_N] biased exp2 table
    double* Gt = Tab + EW_N;                              // [4 waves][2 buffers][BR_GC][BR_LD]
    double* red = Gt;                                     // [4][2*DMAX + R + 1] once the column loop is done
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = dd.D;
    const int64_t ib = (int64_t)blockIdx.y * 256 + 64 * wave;      // first row of this wave
    const int64_t jb = (int64_t)blockIdx.x * cols_per_wg;
    const int64_t jend = (jb + cols_per_wg < nb) ? jb + cols_per_wg : nb;
    for (int j = tid; j < EW_N; j += 256) Tab[j] = biased_table_entry(j);
    double* Gw = Gt + wave * (2 * BR_GC * BR_LD);
    // this lane's row: features in registers for the whole column range
    const int64_t gi = ib + lane;
    const bool row_ok = gi < na;
    const int64_t gr = row_ok ? gi : na - 1;
    double xa[DMAX], ca[DMAX], ad[DMAX];
    {
        const double* pr = Apack + gr * (3 * DMAX);
#pragma unroll
        for (int d = 0; d < DMAX; d += 2) {
            const double2 v0 = *reinterpret_cast<const double2*>(pr + d);
            const double2 v1 = *reinterpret_cast<const double2*>(pr + DMAX + d);
            const double2 v2 = *reinterpret_cast<const double2*>(pr + 2 * DMAX + d);
            xa[d] = v0.x; xa[d + 1] = v0.y; ca[d] = v1.x; ca[d + 1] = v1.y; ad[d] = v2.x; ad[d + 1] = v2.y;
        }
    }
    const double yrow = (yA != nullptr && row_ok) ? yA[a0 + gr] : 0.0;
    double gl[DMAX], gw[R + 1];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) gl[d] = 0.0;
#pragma unroll
    for (int q = 0; q <= R; ++q) gw[q] = 0.0;
    // adjoint tile: load instruction q of a tile covers rows 4q .. 4q+3 of the wave (lane >> 4) and 16 columns (lane & 15)
    const int trow = lane >> 4, tcol = lane & 15;
    auto g_load = [&](int64_t j0, int q) -> double {
        const int64_t r = ib + 4 * q + trow, c = j0 + tcol;
        const bool ok = r < na && c < jend;
        const double v = G[(r < na ? r : na - 1) * ldg + (c < nb ? c : nb - 1)];
        return ok ? v : 0.0;
    };
    __syncthreads();                                      // exp2 table in place
    // first tile: all 16 loads at once (prologue only)
    if (jb < jend) {
#pragma unroll
        for (int q = 0; q < 16; ++q) Gw[tcol * BR_LD + 4 * q + trow] = g_load(jb, q);
    }
    struct Chunk { double xb[4], cb[4], bd[4]; };
    auto fetch = [&](const double* __restrict__ pc, int d0, Chunk& ch) {
#pragma unroll
        for (int v = 0; v < 4; ++v) { ch.xb[v] = pc[3 * d0 + v]; ch.cb[v] = pc[3 * d0 + 4 + v]; ch.bd[v] = pc[3 * d0 + 8 + v]; }
    };
    auto col_ptr = [&](int64_t gj) -> const double* { return Bpack + (gj < jend ? gj : jend - 1) * (3 * DMAX); };
    // The first chunk of a column is fetched while the PREVIOUS column is in its second phase (coefficients, Horner, accumulation:
    // no scalar value is live there); the later chunks one chunk ahead, as in the columns-in-lanes kernel.
    Chunk cur;
    if (jb < jend) fetch(col_ptr(jb), 0, cur);
    int buf = 0;
    double gnext = 0.0, gprev = 0.0;
    for (int64_t j0 = jb; j0 < jend; j0 += BR_GC) {
        const double* Gcur = Gw + buf * (BR_GC * BR_LD);
        double* Gnxt = Gw + (buf ^ 1) * (BR_GC * BR_LD);
        const bool more = j0 + BR_GC < jend;
#pragma unroll 1
        for (int jj = 0; jj < BR_GC; ++jj) {
            const int64_t gj = j0 + jj;
            // next tile's load jj goes out now and lands in LDS two columns later: its latency hides under two whole pairs
            if (jj > 1 && more) Gnxt[tcol * BR_LD + 4 * (jj - 2) + trow] = gprev;
            gprev = gnext;
            if (more) gnext = g_load(j0 + BR_GC, jj);
            const int64_t gc_ = gj < jend ? gj : jend - 1;                  // uniform; columns past the end contribute g = 0
            const double* __restrict__ pc = col_ptr(gj);
            const double av = avec != nullptr ? avec[gc_] : 0.0;
            const double graw = Gcur[jj * BR_LD + lane];
            const double g = (row_ok && gj < jend) ? __builtin_fma(g_scale, graw, yrow * av) : 0.0;
            double k[DMAX], dk[DMAX];
            Chunk nxt;
#pragma unroll
            for (int d0 = 0; d0 < DMAX; d0 += 4) {
                if (d0 + 4 < DMAX) fetch(pc, d0 + 4, nxt);
                asm volatile("" ::: "memory");              // keep later chunks' scalar loads below this point: bounds the live SGPRs
                double w[4], mg[4], E[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double u = xa[d0 + v] - cur.xb[v];
                    w[v] = fma_clamp01(u, u, 0.0); mg[v] = EW_MAGIC;
                }
                exp2_w_vec<4>(w, mg, E, Tab);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    k[d0 + v] = __builtin_fma(-ca[d0 + v], cur.cb[v], E[v]);
                    dk[d0 + v] = __builtin_fma(E[v], w[v], -__builtin_fma(ad[d0 + v], cur.cb[v], ca[d0 + v] * cur.bd[v]));
                }
                if (d0 + 4 < DMAX) cur = nxt;
            }
            fetch(col_ptr(gj + 1), 0, cur);      // next column's first chunk (clamped address past the end: loaded, never used)
            double e[R];
#pragma unroll
            for (int q = 0; q < R; ++q) e[q] = 0.0;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
#pragma unroll
                for (int q = R - 1; q >= 1; --q) e[q] = __builtin_fma(k[d], e[q - 1], e[q]);
                e[0] += k[d];
            }
            gw[0] += g;
#pragma unroll
            for (int q = 1; q <= R; ++q) gw[q] = __builtin_fma(g, e[q - 1], gw[q]);
            // pair-level Horner coefficients of dK/dk_d (see gram_bwd_fast_kernel)
            double cg[R];
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
                gl[d] = __builtin_fma(gc, dk[d], gl[d]);
            }
        }
        if (more) { Gnxt[tcol * BR_LD + 4 * (BR_GC - 2) + trow] = gprev; Gnxt[tcol * BR_LD + 4 * (BR_GC - 1) + trow] = gnext; }
        buf ^= 1;
    }
    // workgroup reduction of the register accumulators (fixed order: lanes by butterfly, then the four waves)
    constexpr int NACC = 2 * DMAX + R + 1;
    auto wave_sum = [](double v) {
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
#pragma unroll
    for (int d = 0; d < DMAX; ++d) gl[d] = wave_sum(gl[d]);
#pragma unroll
    for (int q = 0; q <= R; ++q) gw[q] = wave_sum(gw[q]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < DMAX; ++d) { red[wave * NACC + d] = gl[d]; red[wave * NACC + DMAX + d] = 0.0; }
#pragma unroll
        for (int q = 0; q <= R; ++q) red[wave * NACC + 2 * DMAX + q] = gw[q];
    }
    __syncthreads();
    const int RA = dd.R;
    const int64_t reclen = 2 * D + (RA + 1);
    double* rec = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * reclen;
    auto sum4 = [&](int j) { return ((red[j] + red[NACC + j]) + red[2 * NACC + j]) + red[3 * NACC + j]; };
    for (int d = tid; d < D; d += 256) { rec[d] = sum4(d) * 1024.0; rec[D + d] = 0.0; }
    if (tid <= RA) rec[2 * D + tid] = sum4(2 * DMAX + tid);
}

// host side: see gram_bwd (grad.hip).  cols_per_wg columns and 256 rows per workgroup.
size_t gram_bwd_rows_lds() { return sizeof(double) * ((size_t)EW_N + 4 * 2 * BR_GC * BR_LD); }

int gram_bwd_rows_launch(oak_ctx* ctx, const PreparedKernel& pk, int dmax, const double* d_apack, int64_t a0, int64_t na, const double* d_bpack,
                         int64_t nb, const double* d_G, int64_t ldg, const double* d_yA, const double* d_avec, double g_scale, int cols_per_wg,
                         double* d_part, int64_t* nrec_out) {
    const int R = pk.dd.R;
    const int64_t ncb = (nb + cols_per_wg - 1) / cols_per_wg, nrb = (na + 255) / 256;
    OAK_REQUIRE(nrb <= 65535, "gram_bwd_rows: too many row blocks");
    dim3 grid((unsigned)ncb, (unsigned)nrb);
    const size_t lds = gram_bwd_rows_lds();
#define OAK_BR(RR, DM)                                                                                                                      \
    {                                                                                                                                       \
        auto kern = gram_bwd_rows_kernel<RR, DM>;                                                                                           \
        if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)kern));                                                          \
        kern<<<grid, 256, lds, ctx->stream>>>(pk.dd, d_apack, a0, na, d_bpack, nb, d_G, ldg, d_yA, d_avec, g_scale, cols_per_wg, d_part);   \
    }
    switch (R * 100 + dmax) {
        case 108: OAK_BR(1, 8) break;   case 116: OAK_BR(1, 16) break;
        case 208: OAK_BR(2, 8) break;   case 216: OAK_BR(2, 16) break;
        case 308: OAK_BR(3, 8) break;   case 316: OAK_BR(3, 16) break;
        case 408: OAK_BR(4, 8) break;       // (depth 4 at 16 sub-kernels does not fit the register file: the columns kernel serves it)
        default: set_error("gram_bwd_rows: unsupported shape R=%d dmax=%d", R, dmax); return OAK_E_ARG;
    }
#undef OAK_BR
    OAK_HIP_CHECK(hipGetLastError());
    *nrec_out = nrb * ncb;
    return OAK_OK;
}

}  // namespace oak

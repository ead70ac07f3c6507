// Fused OAK Gram kernel: constrained base kernels (RBF / binary / categorical) + elementary symmetric
// polynomial combination, one pass, nothing but the final K tile leaves registers.
//
// Replaces OAKKernel.K / K_diag (oak/oak_kernel.py:251-278): the reference materialises D per-dimension
// N x M matrices (:252-254), R+1 power sums and R Newton-Girard terms (:236-249).  Here each lane owns an
// RT x CPT block of (row, column) pairs and runs the e_r recurrence  e_r += k_d * e_{r-1}  (r = R..1) over d in
// registers -- algebraically the same elementary symmetric polynomials, D*R FMAs per pair and no pow().
//
// Roofline: fp64-VALU bound (one software exp2 per pair per dimension, exp2w.h: 16.03 VALU wave-instructions per 64
// pair-dimensions at R = 2 including the ESP update -- PMC, profiles/pmc_counts.json), not HBM bound.
// fp64 MFMA shares the DP pipe with fp64 VALU on gfx950 (measured, tools/ubench), so there is nothing to
// overlap with; the kernel is written to issue the minimum number of DP instructions per pair.
#include "oak_internal.h"
#include "exp2w.h"
#include <cstdlib>

#ifndef OAK_GRAM_PREFETCH
#define OAK_GRAM_PREFETCH 1
#endif
#ifndef OAK_GRAM_WAVE_STAGE
#define OAK_GRAM_WAVE_STAGE 1
#endif

namespace oak {

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
// restaged per row-step.  Dynamic LDS = (EW_N + D*TJ*2 + D*4*RT*2 + 4*RT) doubles.
// GRP: some RBF dims read further columns (DevDesc::xrow / nxc, Feat::xx): their squared differences join the exponent before
// the one exponential of the dim.  A separate instantiation, so the kernels of ordinary descriptions compile as before.
template <int R, int RT, int CPT, bool ALLRBF, int TB, bool GRP = false>
__global__ void __launch_bounds__(256)
gram_kernel(const DevDesc dd, const double* __restrict__ tables, const double* __restrict__ Axs,
            const double* __restrict__ Acn, int64_t a_ld, int64_t a0, int64_t na, const double* __restrict__ Bxs,
            const double* __restrict__ Bcn, int64_t b_ld, int64_t nb, double* __restrict__ out, int64_t ldo,
            int rows_per_wg, const double* __restrict__ yA, double* __restrict__ psi_part, int64_t zero_pad_to,
            const double* __restrict__ Axx = nullptr, const double* __restrict__ Bxx = nullptr, int nx = 0, int tablen = 0) {
    constexpr int TJ = 64 * CPT;
    constexpr int RS = 4 * RT;   // rows per row-step
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int D = dd.D;
    constexpr int TABN = 1 << TB;      // exp2 table entries: 1024, or 512 where the larger table would cost a workgroup per CU
    double* Tab = smem;                // [TABN] biased exp2 table (exp2w.h); first, so the lookups use a constant LDS base
    double* Bx = Tab + TABN;             // [D][TJ]   (RBF dims: x * scale_d / 32)
    double* Bc = Bx + D * TJ;          // [D][TJ]
    double* Ax = Bc + D * TJ;          // [D][RS]
    double* Ac = Ax + D * RS;          // [D][RS]
    double* Ay = Ac + D * RS;          // [RS]
    double* Bq = Ay + RS;              // [nx][TJ]  further columns of grouped dims (GRP), pre-scaled like Bx
    double* Aq = Bq + (GRP ? nx : 0) * TJ;   // [nx][RS]
    double* Tl = Aq + (GRP ? nx : 0) * RS;   // [tablen] discrete tables (mixed kernels; tablen = 0: they stay in global memory)
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
    if constexpr (GRP) {
        for (int idx = tid; idx < nx * TJ; idx += 256) {
            const int q = idx / TJ, j = idx - q * TJ;
            Bq[idx] = (jb + j < nb) ? Bxx[(int64_t)q * b_ld + jb + j] * 0.03125 : 0.0;
        }
    }
    for (int j = tid; j < TABN; j += 256) Tab[j] = biased_table_entry<TB>(j);
    if constexpr (!ALLRBF) {
        for (int j = tid; j < tablen; j += 256) Tl[j] = tables[j];
    }
    double psi[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) psi[c] = 0.0;

#if OAK_GRAM_WAVE_STAGE
    // WAVE-PRIVATE row staging (r06): wave ty reads only the RT rows it owns (Ax[d * RS + ty * RT + r]), so it also stages only those
    // -- D * RT values per array, one (<= 16 sub-kernels) or RT (<= 64) per lane -- and the row loop needs NO workgroup barrier: a
    // wave's LDS operations execute in program order, its writes of step s + 1 follow its reads of step s.  (The cooperative form
    // below spent two __syncthreads per 16-row step: 17 % of the wave time parked, profiles/r05_pmc_pair_kernels.json.)  The next
    // step's values are fetched into registers under the current step's arithmetic, raw and from clamped addresses.
    constexpr int NSW = (CPT == 4) ? 1 : RT;
    const int lane = tid & 63;
    double rx[NSW], rc[NSW], ry = 0.0;
    auto fetch_rows = [&](int64_t i0n) {
#pragma unroll
        for (int s4 = 0; s4 < NSW; ++s4) {
            const int idx = lane + 64 * s4;
            const int d = idx / RT, r = idx - d * RT;
            int64_t gi = i0n + ty * RT + r;
            gi = gi < iend ? gi : iend - 1;
            const bool in = idx < D * RT;
            rx[s4] = in ? Axs[(int64_t)d * a_ld + a0 + gi] : 0.0;
            rc[s4] = in ? Acn[(int64_t)d * a_ld + a0 + gi] : 0.0;
        }
        if (yA != nullptr && lane < RT) { const int64_t gi = i0n + ty * RT + lane; ry = yA[a0 + (gi < iend ? gi : iend - 1)]; }
    };
    if (ib < iend) fetch_rows(ib);
    __syncthreads();   // B-side features, exp2 table, discrete tables staged
    for (int64_t i0 = ib; i0 < iend; i0 += RS) {
#pragma unroll
        for (int s4 = 0; s4 < NSW; ++s4) {
            const int idx = lane + 64 * s4;
            if (idx < D * RT) {
                const int d = idx / RT, r = idx - d * RT;
                const bool ok = i0 + ty * RT + r < iend;
                const double pre = (ALLRBF || dd.type[d] == OAK_DIM_RBF) ? 0.03125 : 1.0;
                Ax[d * RS + ty * RT + r] = ok ? rx[s4] * pre : 0.0;
                Ac[d * RS + ty * RT + r] = ok ? rc[s4] : 0.0;
            }
        }
        if (yA != nullptr && lane < RT) Ay[ty * RT + lane] = (i0 + ty * RT + lane < iend) ? ry : 0.0;
        if constexpr (GRP) {
            for (int idx = lane; idx < nx * RT; idx += 64) {
                const int q = idx / RT, r = idx - q * RT;
                Aq[q * RS + ty * RT + r] = (i0 + ty * RT + r < iend) ? Axx[(int64_t)q * a_ld + a0 + i0 + ty * RT + r] * 0.03125 : 0.0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (compiler ordering only: the wave's own LDS writes before its reads)
        if (i0 + RS < iend) fetch_rows(i0 + RS);
#else
#if OAK_GRAM_PREFETCH
    // (cooperative staging, kept for A/B: -DOAK_GRAM_WAVE_STAGE=0)  Row-side features of the NEXT row step are fetched into registers while the current one is computed (D * RS <= 1024 values per
    // array: at most four per thread), raw and from clamped addresses; mask and pre-scale are applied when they go to LDS.  Without
    // this the global loads sat between the two barriers of a step: every wave of the workgroup waited out an HBM latency per 16 rows.
    // (the launcher pairs CPT = 4 with <= 16 sub-kernels: one value per thread there; CPT = 2 goes to 64)
    constexpr int NS = (CPT == 4) ? 1 : RS / 4;       // values per array per thread: 16 sub-kernels x RS rows, or 64 x RS, over 256 threads
    double rx[NS], rc[NS], ry = 0.0;
    auto fetch_rows = [&](int64_t i0n) {
#pragma unroll
        for (int s4 = 0; s4 < NS; ++s4) {
            const int idx = tid + 256 * s4;
            const int d = idx / RS, r = idx - d * RS;
            int64_t gi = i0n + r;
            gi = gi < iend ? gi : iend - 1;
            const bool in = idx < D * RS;
            rx[s4] = in ? Axs[(int64_t)d * a_ld + a0 + gi] : 0.0;
            rc[s4] = in ? Acn[(int64_t)d * a_ld + a0 + gi] : 0.0;
        }
        if (yA != nullptr && tid < RS) { const int64_t gi = i0n + tid; ry = yA[a0 + (gi < iend ? gi : iend - 1)]; }
    };
    if (ib < iend) fetch_rows(ib);
#endif
    for (int64_t i0 = ib; i0 < iend; i0 += RS) {
        __syncthreads();   // previous step's readers done (and B-side staged on first pass)
#if OAK_GRAM_PREFETCH
#pragma unroll
        for (int s4 = 0; s4 < NS; ++s4) {
            const int idx = tid + 256 * s4;
            if (idx < D * RS) {
                const int d = idx / RS, r = idx - d * RS;
                const bool ok = i0 + r < iend;
                const double pre = (ALLRBF || dd.type[d] == OAK_DIM_RBF) ? 0.03125 : 1.0;
                Ax[idx] = ok ? rx[s4] * pre : 0.0;
                Ac[idx] = ok ? rc[s4] : 0.0;
            }
        }
        if (yA != nullptr && tid < RS) Ay[tid] = (i0 + tid < iend) ? ry : 0.0;
#else
        for (int idx = tid; idx < D * RS; idx += 256) {
            const int d = idx / RS, r = idx - d * RS;
            const int64_t gi = i0 + r;
            const bool ok = gi < iend;
            const double pre = (ALLRBF || dd.type[d] == OAK_DIM_RBF) ? 0.03125 : 1.0;
            Ax[idx] = ok ? Axs[(int64_t)d * a_ld + a0 + gi] * pre : 0.0;
            Ac[idx] = ok ? Acn[(int64_t)d * a_ld + a0 + gi] : 0.0;
        }
        if (yA != nullptr && tid < RS) Ay[tid] = (i0 + tid < iend) ? yA[a0 + i0 + tid] : 0.0;
#endif
        if constexpr (GRP) {
            for (int idx = tid; idx < nx * RS; idx += 256) {
                const int q = idx / RS, r = idx - q * RS;
                Aq[idx] = (i0 + r < iend) ? Axx[(int64_t)q * a_ld + a0 + i0 + r] * 0.03125 : 0.0;
            }
        }
        __syncthreads();
#if OAK_GRAM_PREFETCH
        if (i0 + RS < iend) fetch_rows(i0 + RS);
#endif

#endif   // OAK_GRAM_WAVE_STAGE
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
                // k_d of this lane's RT x CPT pairs first, ESP update after the branch re-joins: in the mixed-type kernel only
                // these RT*CPT values cross the merge (not the RT*CPT*R accumulators, which cost a v_mov each per dimension)
                double kk[RT][CPT];
                if (ALLRBF || dd.type[d] == OAK_DIM_RBF) {
                    const double woff = dd.woff[d], magic = (dd.magic[d] - EW_MAGIC) + ew_magic<TB>();   // exact: n/1024 + magic(TB)
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        double w[CPT], E[CPT];
#pragma unroll
                        for (int c = 0; c < CPT; ++c) {
                            const double u = xa[r] - xb[c];
                            w[c] = fma_clamp01(u, u, woff);
                        }
                        if constexpr (GRP) {
                            for (int q = dd.xrow[d]; q < dd.xrow[d] + dd.nxc[d]; ++q) {
                                const double qa = Aq[q * RS + ty * RT + r];
#pragma unroll
                                for (int c = 0; c < CPT; ++c) {
                                    const double u = qa - Bq[q * TJ + (c >> 1) * 128 + 2 * tx + (c & 1)];
                                    w[c] = fma_clamp01(u, u, w[c]);     // clamp01(clamp01(a) + b) = clamp01(a + b) for a, b >= 0
                                }
                            }
                        }
                        double mg[CPT];
#pragma unroll
                        for (int c = 0; c < CPT; ++c) mg[c] = magic;
                        exp2_w_vec<CPT, TB>(w, mg, E, Tab);
#pragma unroll
                        for (int c = 0; c < CPT; ++c) kk[r][c] = __builtin_fma(-ca[r], cb[c], E[c]);
                    }
                } else {
                    // a gather from the (tiny) table: out of LDS -- from global memory each dimension waited ~600 cycles on eight
                    // dependent loads per lane and a binary input cost as much as an exponential one (C5: 0.30 ms per dimension)
                    if (dd.type[d] == OAK_DIM_BINARY) {          // rank one: the factors ride in the cn slot (featurize_point)
#pragma unroll
                        for (int r = 0; r < RT; ++r)
#pragma unroll
                            for (int c = 0; c < CPT; ++c) kk[r][c] = ca[r] * cb[c];
                    } else {
                    const int C = dd.ncat[d];
                    int ia[RT], ib[CPT];
#pragma unroll
                    for (int r = 0; r < RT; ++r) ia[r] = dd.tab_off[d] + (int)xa[r] * C;
#pragma unroll
                    for (int c = 0; c < CPT; ++c) ib[c] = (int)xb[c];
                    if (tablen > 0) {
#pragma unroll
                        for (int r = 0; r < RT; ++r)
#pragma unroll
                            for (int c = 0; c < CPT; ++c) kk[r][c] = Tl[ia[r] + ib[c]];
                    } else {
#pragma unroll
                        for (int r = 0; r < RT; ++r)
#pragma unroll
                            for (int c = 0; c < CPT; ++c) kk[r][c] = tables[ia[r] + ib[c]];
                    }
                    }
                }
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CPT; ++c) esp_update<R>(e[r][c], kk[r][c]);
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

// column sums of a [rows x cols] row-major matrix in fixed order, accumulated onto out: 32 columns per workgroup, eight
// interleaved row groups per column, combined in order through LDS
__global__ void __launch_bounds__(256) colsum_accum_kernel(const double* __restrict__ part, int64_t rows, int64_t cols,
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
    OAK_REQUIRE(D * RS <= 256 * ((CPT == 4) ? 1 : RS / 4), "gram: %d sub-kernels do not fit this tile shape's row staging", D);
    // 1024-entry exp2 table unless its extra 4 KiB would lower the number of workgroups a CU holds (D = 32 at TJ = 128)
    const int nx = pk.grouped ? A.nx : 0;
    if (pk.grouped) OAK_REQUIRE(A.xx != nullptr && B.xx != nullptr && A.nx == B.nx, "gram: features lack the grouped sub-kernels' further columns");
    bool all_rbf = true;
    for (int d = 0; d < D; ++d) all_rbf = all_rbf && pk.dd.type[d] == OAK_DIM_RBF;
    const int tablen = (!all_rbf && pk.tables.size() <= 1024) ? (int)pk.tables.size() : 0;      // discrete tables ride in LDS up to 8 KiB
    const size_t lds_body = sizeof(double) * ((size_t)D * TJ * 2 + (size_t)D * RS * 2 + RS + (size_t)nx * (TJ + RS) + (size_t)tablen);
    const size_t cu_lds = 160 * 1024;
    const bool big_table = cu_lds / (lds_body + sizeof(double) * 1024) == cu_lds / (lds_body + sizeof(double) * 512);
    size_t lds = lds_body + sizeof(double) * (big_table ? 1024 : 512);
    const size_t lds_red = sizeof(double) * 4 * TJ;
    if (lds < lds_red) lds = lds_red;
    if (lds > 160 * 1024) { set_error("gram: LDS request %zu exceeds 160 KiB (D=%d)", lds, D); return OAK_E_ARG; }
    const int64_t nb = B.n;
    const int64_t ncb = (nb + TJ - 1) / TJ;
    // rows per workgroup: enough row-blocks to fill the chip several times over (r05, inside the step at 2^20 rows: 9.51-9.54 ms at 16
    // workgroups per CU, 9.25-9.36 at 24 -- twelve rounds instead of eight leave a shorter ragged end, PMC: 1.64 resident waves per SIMD
    // on average where two fit; resident back-to-back launches measure the same with either), at least one row-step
    // (row shards of 131 072 rows measure 1.32 ms at 16 and 1.33-1.345 at 24: the finer split only pays on long panels)
    int64_t target_wg = (int64_t)ctx->num_cu * (na >= (1 << 19) ? 24 : 16);
    if (const char* e = getenv("OAK_GRAM_WG_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 64) target_wg = (int64_t)ctx->num_cu * v; }   // tuning knob
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
    auto kern = big_table ? (all_rbf ? gram_kernel<R, RT, CPT, true, 10> : gram_kernel<R, RT, CPT, false, 10>)
                          : (all_rbf ? gram_kernel<R, RT, CPT, true, 9> : gram_kernel<R, RT, CPT, false, 9>);
    if (nx > 0) kern = gram_kernel<R, RT, CPT, false, 9, true>;      // grouped sub-kernels: one instantiation per shape (512-entry table)
    if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)kern));
    dim3 grid((unsigned)ncb, (unsigned)nrb);
    kern<<<grid, 256, lds, ctx->stream>>>(pk.dd, pk.d_tables, A.xs, A.cn, A.ld, a0, na, B.xs, B.cn, B.ld, nb, d_out, ldo,
                                          (int)rows, d_yA, d_part, zero_pad_to, A.xx, B.xx, nx, tablen);
    OAK_HIP_CHECK(hipGetLastError());
    if (d_yA != nullptr) {
        colsum_accum_kernel<<<(unsigned)((nb + 31) / 32), 256, 0, ctx->stream>>>(d_part, nrb, nb, d_psi);
        OAK_HIP_CHECK(hipGetLastError());
    }
    return OAK_OK;
}

template <int R>
static int launch_gram_r(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B,
                         double* d_out, int64_t ldo, const double* d_yA, double* d_psi, int64_t zero_pad_to) {
    const int D = pk.dd.D;
    // register blocking: 16 pairs/lane for R<=4 (8 beyond 16 sub-kernels), 4 above; halve the column tile when D*TJ*16 B > 64 KiB
    if constexpr (R <= 2) {
        if (D <= 16) return launch_gram_t<R, 4, 4>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        return launch_gram_t<R, 4, 2>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
    } else if constexpr (R <= 4) {
        // 16 pairs per lane here too (r04: depth 4 at 16 dims 11.2 -> 10.3 ms, depth 3 10.05 -> 9.7; 8 dims: 2 % / none)
        if (D <= 16) return launch_gram_t<R, 4, 4>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        return launch_gram_t<R, 4, 2>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
    } else if constexpr (R <= 16) {
        if constexpr (R <= 8) {      // depth 5..8: 8 pairs per lane (r04: 3-7 % over 4 at 10 and 16 sub-kernels)
            if (D <= 16) return launch_gram_t<R, 2, 4>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
            return launch_gram_t<R, 4, 2>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);     // 8 pairs here too: 13-34 % over 2 (20..64 sub-kernels)
        }
        // depth 9..16: 4 pairs per lane (2 x 4 at <= 16 sub-kernels measured WORSE than 1 x 4: 8.8 vs 6.7 ms at depth 12 of 13;
        // 2 x 2 beyond 16 sub-kernels against 1 x 2: 11.6 vs 13.1 ms at depth 12 of 24, 16.8 vs 21.7 at 16 of 32)
        if (D <= 16) return launch_gram_t<R, 1, 4>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        return launch_gram_t<R, 2, 2>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
    } else {
        // effective depth > 16 means more than 16 sub-kernels: two pairs per lane, R + 1 polynomials each
        // (2 x 2 here: 30.0 vs 24.7 ms at depth 20 of 32 -- the 4 x 25 polynomials no longer fit two waves per SIMD)
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
        // Depths 9..16 (the reference's regression example runs max_interaction_depth = D, up to 13 on UCI housing,
        // examples/uci/uci_regression_train.py:86): the next larger instantiation.  DevDesc::w is zero above R, so the extra
        // elementary symmetric polynomials are formed and weighted by 0.
        case 9: case 10: case 11: case 12: return launch_gram_r<12>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 13: case 14: case 15: case 16: return launch_gram_r<16>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 17: case 18: case 19: case 20: case 21: case 22: case 23: case 24:
            return launch_gram_r<24>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
        case 25: case 26: case 27: case 28: case 29: case 30: case 31: case 32:
            return launch_gram_r<32>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to);
    }
    set_error("gram: unsupported depth %d", pk.dd.R);
    return OAK_E_ARG;
}

// ---------------------------------------------------------------------------------------------
// Gram kernel with the int8 CRT epilogue (crt.hip, oak_sgpr_set_precision 2): the same pair arithmetic, and every finished entry
// K[n, m] leaves the kernel as its L residues  rint(K 2^s_m) mod p_i  (one signed byte each) in the plane layout
// [plane][n / 16][m][n % 16] the int8 SYRK reads -- the fp64 panel is written only when a gradient or a further output needs it
// (PANEL).  Geometry: a WAVE owns all 16 rows of a row step for 16 CPT columns (lane = (tr, tc): rows 4 tr .. 4 tr + 3, columns
// tc + 16 c), so that the four rows of a lane are one dword of a 16-byte plane unit and the four tr lanes of a column fill it.
// Conversion per entry and modulus, all on the DP pipe with exact results: xm = K 2^s + 1.5 2^52 (the integer x in the low mantissa
// bits), q = rint(x / p) by the same magic constant, xm - q p = r + 1.5 2^52: the low byte of the low word is r in two's complement.
// ---------------------------------------------------------------------------------------------
template <int R, int CPT, bool ALLRBF, int TB, bool PANEL>
__global__ void __launch_bounds__(256, 2)
gram_crt_kernel(const DevDesc dd, const double* __restrict__ tables, const double* __restrict__ Axs, const double* __restrict__ Acn,
                int64_t a_ld, int64_t a0, int64_t na, const double* __restrict__ Bxs, const double* __restrict__ Bcn, int64_t b_ld,
                int64_t nb, double* __restrict__ out, int64_t ldo, int rows_per_wg, const double* __restrict__ yA,
                double* __restrict__ psi_part, int64_t zero_pad_to, const CrtMod md, const int* __restrict__ sexp,
                int8_t* __restrict__ planes, int64_t plane_stride, int64_t Mp2, int tablen) {
    constexpr int RT = 4, RS = 16, WC = 16 * CPT, TJ = 4 * WC;
    constexpr double M52 = 6755399441055744.0;      // 1.5 * 2^52
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int D = dd.D;
    constexpr int TABN = 1 << TB;
    double* Tab = smem;
    double* Bx = Tab + TABN;           // [D][TJ], columns permuted so that a lane's CPT columns are 16-byte pairs (see pos below)
    double* Bc = Bx + D * TJ;
    double* Ax = Bc + D * TJ;          // [D][RS]
    double* Ac = Ax + D * RS;
    double* Ay = Ac + D * RS;          // [RS]
    double* Tl = Ay + RS;              // [tablen]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = lane >> 4, tc = lane & 15;
    const int64_t jb = (int64_t)blockIdx.x * TJ;
    const int64_t ib = (int64_t)blockIdx.y * rows_per_wg;
    const int64_t iend = (ib + rows_per_wg < na) ? ib + rows_per_wg : na;

    for (int idx = tid; idx < D * TJ; idx += 256) {
        const int d = idx / TJ, j = idx - d * TJ;
        const int ww = j / WC, jj = j - ww * WC, c = jj >> 4, t = jj & 15;
        const int pos = ww * WC + (c >> 1) * 32 + t * 2 + (c & 1);
        const int64_t gj = jb + j;
        const bool ok = gj < nb;
        const double pre = (ALLRBF || dd.type[d] == OAK_DIM_RBF) ? 0.03125 : 1.0;
        Bx[d * TJ + pos] = ok ? Bxs[(int64_t)d * b_ld + gj] * pre : 0.0;
        Bc[d * TJ + pos] = ok ? Bcn[(int64_t)d * b_ld + gj] : 0.0;
    }
    for (int j = tid; j < TABN; j += 256) Tab[j] = biased_table_entry<TB>(j);
    if constexpr (!ALLRBF) {
        for (int j = tid; j < tablen; j += 256) Tl[j] = tables[j];
    }
    double psi[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) psi[c] = 0.0;
    const int colbase = (int)jb + w * WC + tc;             // this lane's columns: colbase + 16 c  (the planes and sexp have Mp2 >= gridDim.x * TJ columns)
    const int nbi = (int)nb;

    constexpr int NS = (CPT == 4) ? 1 : RS / 4;
    double rx[NS], rc[NS], ry = 0.0;
    auto fetch_rows = [&](int64_t i0n) {
#pragma unroll
        for (int s4 = 0; s4 < NS; ++s4) {
            const int idx = tid + 256 * s4;
            const int d = idx / RS, r = idx - d * RS;
            int64_t gi = i0n + r;
            gi = gi < iend ? gi : iend - 1;
            const bool in = idx < D * RS;
            rx[s4] = in ? Axs[(int64_t)d * a_ld + a0 + gi] : 0.0;
            rc[s4] = in ? Acn[(int64_t)d * a_ld + a0 + gi] : 0.0;
        }
        if (yA != nullptr && tid < RS) { const int64_t gi = i0n + tid; ry = yA[a0 + (gi < iend ? gi : iend - 1)]; }
    };
    if (ib < iend) fetch_rows(ib);
    for (int64_t i0 = ib; i0 < iend; i0 += RS) {
        __syncthreads();
#pragma unroll
        for (int s4 = 0; s4 < NS; ++s4) {
            const int idx = tid + 256 * s4;
            if (idx < D * RS) {
                const int d = idx / RS, r = idx - d * RS;
                const bool ok = i0 + r < iend;
                const double pre = (ALLRBF || dd.type[d] == OAK_DIM_RBF) ? 0.03125 : 1.0;
                Ax[idx] = ok ? rx[s4] * pre : 0.0;
                Ac[idx] = ok ? rc[s4] : 0.0;
            }
        }
        if (yA != nullptr && tid < RS) Ay[tid] = (i0 + tid < iend) ? ry : 0.0;
        __syncthreads();
        if (i0 + RS < iend) fetch_rows(i0 + RS);

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
                    const double2 vx = *reinterpret_cast<const double2*>(&Bx[d * TJ + w * WC + c2 * 32 + 2 * tc]);
                    const double2 vc = *reinterpret_cast<const double2*>(&Bc[d * TJ + w * WC + c2 * 32 + 2 * tc]);
                    xb[2 * c2] = vx.x; xb[2 * c2 + 1] = vx.y;
                    cb[2 * c2] = vc.x; cb[2 * c2 + 1] = vc.y;
                }
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    xa[r] = Ax[d * RS + tr * RT + r];
                    ca[r] = Ac[d * RS + tr * RT + r];
                }
                double kk[RT][CPT];
                if (ALLRBF || dd.type[d] == OAK_DIM_RBF) {
                    const double woff = dd.woff[d], magic = (dd.magic[d] - EW_MAGIC) + ew_magic<TB>();
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        double wv[CPT], E[CPT], mg[CPT];
#pragma unroll
                        for (int c = 0; c < CPT; ++c) {
                            const double u = xa[r] - xb[c];
                            wv[c] = fma_clamp01(u, u, woff);
                            mg[c] = magic;
                        }
                        exp2_w_vec<CPT, TB>(wv, mg, E, Tab);
#pragma unroll
                        for (int c = 0; c < CPT; ++c) kk[r][c] = __builtin_fma(-ca[r], cb[c], E[c]);
                    }
                } else if (dd.type[d] == OAK_DIM_BINARY) {
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int c = 0; c < CPT; ++c) kk[r][c] = ca[r] * cb[c];
                } else {
                    const int C = dd.ncat[d];
                    int ia[RT], ibb[CPT];
#pragma unroll
                    for (int r = 0; r < RT; ++r) ia[r] = dd.tab_off[d] + (int)xa[r] * C;
#pragma unroll
                    for (int c = 0; c < CPT; ++c) ibb[c] = (int)xb[c];
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int c = 0; c < CPT; ++c) kk[r][c] = tablen > 0 ? Tl[ia[r] + ibb[c]] : tables[ia[r] + ibb[c]];
                }
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CPT; ++c) esp_update<R>(e[r][c], kk[r][c]);
            }
        }
        // epilogue: combine orders, psi, optional fp64 panel, scaled integers
        double xm[RT][CPT], xi[RT][CPT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int64_t gi = i0 + tr * RT + r;
            const double yv = (yA != nullptr) ? Ay[tr * RT + r] : 0.0;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const double kv = esp_combine<R>(e[r][c], dd);
                if (yA != nullptr) psi[c] = __builtin_fma(kv, yv, psi[c]);
                const int cc = colbase + 16 * c;
                const bool live = gi < iend && cc < nbi;
                if constexpr (PANEL) {
                    if (gi < iend) {
                        if (cc < nbi) out[gi * ldo + cc] = kv;
                        else if (cc < zero_pad_to) out[gi * ldo + cc] = 0.0;
                    }
                }
                const double sc = __builtin_ldexp(1.0, sexp[cc]);      // (an L1 hit per step: cheaper than four live doubles in the pair loop)
                xm[r][c] = live ? __builtin_fma(kv, sc, M52) : M52;
                xi[r][c] = xm[r][c] - M52;
            }
        }
        const int64_t unit = ((i0 >> 4) * Mp2) * 16 + tr * 4;          // byte offset of this lane's dword inside a plane, column 0
        for (int i = 0; i < md.L; ++i) {
            const double p = (double)md.p[i], ip = md.inv[i];
            int8_t* pl = planes + (int64_t)i * plane_stride + unit;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                unsigned b[RT];
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const double q = __builtin_rint(xi[r][c] * ip);        // (v_rndne_f64 issues in half the clocks of the add it replaces)
                    b[r] = (unsigned)__double2loint(__builtin_fma(q, -p, xm[r][c]));
                }
                // low bytes of b0 .. b3 -> one dword, three byte permutes (selector 0-3: bytes of the second source, 4-7: of the first)
                unsigned wv = __builtin_amdgcn_perm(b[1], b[0], 0x07060400u);             // b0.0, b1.0, (b1.2, b1.3)
                wv = __builtin_amdgcn_perm(b[2], wv, 0x07040100u);                        // b0.0, b1.0, b2.0, (b2.3)
                wv = __builtin_amdgcn_perm(b[3], wv, 0x04020100u);                        // b0.0, b1.0, b2.0, b3.0
                *reinterpret_cast<unsigned*>(pl + (colbase + 16 * c) * 16) = wv;
            }
        }
    }
    if (yA != nullptr) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            double v = psi[c];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (tr == 0 && colbase + 16 * c < nbi) psi_part[(int64_t)blockIdx.y * nb + colbase + 16 * c] = v;
        }
    }
}

// row groups [g_lo, g_hi) of every plane <- 0 (the rows between the end of the data and the end of the last row split)
__global__ void __launch_bounds__(256) crt_zero_groups_kernel(int8_t* __restrict__ planes, int64_t plane_stride, int64_t Mp2, int64_t g_lo, int64_t g_hi) {
    const int64_t n16 = (g_hi - g_lo) * Mp2;            // 16-byte units per plane
    uint4* base = reinterpret_cast<uint4*>(planes + (int64_t)blockIdx.y * plane_stride + g_lo * Mp2 * 16);
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < n16; u += (int64_t)gridDim.x * 256) base[u] = make_uint4(0, 0, 0, 0);
}

bool gram_crt_supported(const PreparedKernel& pk) { return !pk.deep && !pk.grouped && pk.dd.R <= 4 && pk.dd.D <= 64; }

template <int R, int CPT>
static int launch_gram_crt_t(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, double* d_out, int64_t ldo,
                             const double* d_yA, double* d_psi, int64_t zero_pad_to, const CrtMod& md, const int* d_sexp, int8_t* d_planes,
                             int64_t plane_stride, int64_t Mp2) {
    constexpr int TJ = 64 * CPT, RS = 16;
    const int D = pk.dd.D;
    OAK_REQUIRE(D * RS <= 256 * ((CPT == 4) ? 1 : RS / 4), "gram (CRT): %d sub-kernels do not fit this tile shape's row staging", D);
    OAK_REQUIRE(Mp2 % TJ == 0, "gram (CRT): the plane width %lld is not a multiple of the column tile", (long long)Mp2);
    bool all_rbf = true;
    for (int d = 0; d < D; ++d) all_rbf = all_rbf && pk.dd.type[d] == OAK_DIM_RBF;
    const int tablen = (!all_rbf && pk.tables.size() <= 1024) ? (int)pk.tables.size() : 0;
    const size_t lds_body = sizeof(double) * ((size_t)D * TJ * 2 + (size_t)D * RS * 2 + RS + (size_t)tablen);
    const size_t cu_lds = 160 * 1024;
    const bool big_table = cu_lds / (lds_body + sizeof(double) * 1024) == cu_lds / (lds_body + sizeof(double) * 512);
    const size_t lds = lds_body + sizeof(double) * (big_table ? 1024 : 512);
    if (lds > 160 * 1024) { set_error("gram (CRT): LDS request %zu exceeds 160 KiB (D=%d)", lds, D); return OAK_E_ARG; }
    const int64_t nb = B.n;
    const int64_t ncb = Mp2 / TJ;                       // every plane column is written (zeros beyond nb)
    int64_t target_wg = (int64_t)ctx->num_cu * (na >= (1 << 19) ? 24 : 16);
    if (const char* e = getenv("OAK_GRAM_WG_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 64) target_wg = (int64_t)ctx->num_cu * v; }
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
    const bool panel = d_out != nullptr;
#define OAK_GC(AR, TBV, PN) gram_crt_kernel<R, CPT, AR, TBV, PN>
    auto kern = big_table ? (all_rbf ? (panel ? OAK_GC(true, 10, true) : OAK_GC(true, 10, false)) : (panel ? OAK_GC(false, 10, true) : OAK_GC(false, 10, false)))
                          : (all_rbf ? (panel ? OAK_GC(true, 9, true) : OAK_GC(true, 9, false)) : (panel ? OAK_GC(false, 9, true) : OAK_GC(false, 9, false)));
#undef OAK_GC
    if (lds > 64 * 1024) OAK_CHECK(ensure_max_dynamic_lds((const void*)kern));
    dim3 grid((unsigned)ncb, (unsigned)nrb);
    kern<<<grid, 256, lds, ctx->stream>>>(pk.dd, pk.d_tables, A.xs, A.cn, A.ld, a0, na, B.xs, B.cn, B.ld, nb, d_out, ldo, (int)rows, d_yA, d_part,
                                          zero_pad_to, md, d_sexp, d_planes, plane_stride, Mp2, tablen);
    OAK_HIP_CHECK(hipGetLastError());
    if (d_yA != nullptr) {
        colsum_accum_kernel<<<(unsigned)((nb + 31) / 32), 256, 0, ctx->stream>>>(d_part, nrb, nb, d_psi);
        OAK_HIP_CHECK(hipGetLastError());
    }
    return OAK_OK;
}

// Gram panel chunk + its residue planes (rows [0, rows_pad) of the chunk: zeros beyond na).  d_out == NULL: no fp64 panel.
int gram_crt(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, double* d_out, int64_t ldo,
             const double* d_yA, double* d_psi, int64_t zero_pad_to, const CrtMod& md, const int* d_sexp, int8_t* d_planes, int64_t rows_pad,
             int64_t Mp2) {
    OAK_REQUIRE(gram_crt_supported(pk), "gram (CRT): kernel description outside the fused residue kernel's shapes");
    if (na <= 0 || B.n <= 0) return OAK_OK;
    const int64_t plane_stride = rows_pad * Mp2;
    const int64_t g_lo = (na + 15) / 16, g_hi = rows_pad / 16;
    if (g_hi > g_lo) {
        crt_zero_groups_kernel<<<dim3(64, (unsigned)md.L), 256, 0, ctx->stream>>>(d_planes, plane_stride, Mp2, g_lo, g_hi);
        OAK_HIP_CHECK(hipGetLastError());
    }
    const bool wide = pk.dd.D <= 16;
#define OAK_GCL(RR) case RR: return wide ? launch_gram_crt_t<RR, 4>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to, md, d_sexp, d_planes, plane_stride, Mp2) \
                                         : launch_gram_crt_t<RR, 2>(ctx, pk, A, a0, na, B, d_out, ldo, d_yA, d_psi, zero_pad_to, md, d_sexp, d_planes, plane_stride, Mp2);
    switch (pk.dd.R) { OAK_GCL(0) OAK_GCL(1) OAK_GCL(2) OAK_GCL(3) OAK_GCL(4) }
#undef OAK_GCL
    set_error("gram (CRT): unsupported depth %d", pk.dd.R);
    return OAK_E_ARG;
}

// ---------------------------------------------------------------------------------------------
// Generic Gram: one thread per entry, runtime depth (<= OAK_MAX_DIMS), two arithmetic forms (oak_internal.h: gram_generic).
// FORM 1 follows oracle/oak_oracle.py -- i.e. the reference -- operation by operation: u = x / l (IEEE division, as NumPy),
// r2 = ((-2 u_x) u_z + u_x^2) + u_z^2, k = variance * exp(-r2 / 2) - c(x) c(z) / var_s, s_p = sum_d k^p by repeated products,
// e_n = (1 / n) sum_k (-1)^(k-1) e_{n-k} s_k, K = sum_n sigma2_n e_n; no FMA contraction (the file is built with
// -ffp-contract=off), the library's exp.  FORM 0 is this library's arithmetic in the same slow shape: (u_x - u_z)^2 and the
// recurrence e_r += k_d e_{r-1}.
// ---------------------------------------------------------------------------------------------
struct GenericDims {
    int D, R;
    unsigned char type[OAK_MAX_DIMS];
    short col[OAK_MAX_DIMS];
    int ncat[OAK_MAX_DIMS], tab_off[OAK_MAX_DIMS];
    double ls[OAK_MAX_DIMS], bv[OAK_MAX_DIMS];
    short xrow[OAK_MAX_DIMS]; unsigned char nxc[OAK_MAX_DIMS];   // grouped sub-kernels: further columns of dim d are xcols[xrow[d] .. xrow[d] + nxc[d])
};

template <int FORM, bool DIAG>
__global__ void __launch_bounds__(256)
gram_generic_kernel(const GenericDims g, const double* __restrict__ w, const int* __restrict__ xcols, const double* __restrict__ tables, const double* __restrict__ Xa,
                    const double* __restrict__ Axs, const double* __restrict__ Acn, int64_t a_ld, int64_t na, const double* __restrict__ Xb,
                    const double* __restrict__ Bxs, const double* __restrict__ Bcn, int64_t b_ld, int64_t nb, int32_t ldx,
                    double* __restrict__ out, int64_t ldo) {
    const int64_t j = DIAG ? 0 : (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = DIAG ? (int64_t)blockIdx.x * 256 + threadIdx.x : (int64_t)blockIdx.y;
    if (i >= na || (!DIAG && j >= nb)) return;
    const int D = g.D, R = g.R;
    double acc[OAK_MAX_DIMS + 1];                 // FORM 1: power sums s_0..s_R; FORM 0: e_1..e_R (acc[r - 1])
    for (int r = 0; r <= R; ++r) acc[r] = 0.0;
    for (int d = 0; d < D; ++d) {
        double k;
        if (g.type[d] == OAK_DIM_RBF) {
            const double ca = Acn[(int64_t)d * a_ld + i];
            if (DIAG) {
                k = g.bv[d] - ca * ca;                                            // oak/ortho_rbf_kernel.py:174-177
            } else {
                const double ux = Xa[i * ldx + g.col[d]] / g.ls[d], uz = Xb[j * ldx + g.col[d]] / g.ls[d];
                double r2;
                if (g.nxc[d] > 0) {
                    // a sub-kernel over several columns: one RBF of the group's squared distance (oak_kernel.py:199-210)
                    if (FORM == 1) {                                               // gpflow: -2 X X2^T + |X|^2 + |X2|^2, sums over columns
                        double xz = ux * uz, xx = ux * ux, zz = uz * uz;
                        for (int q = g.xrow[d]; q < g.xrow[d] + g.nxc[d]; ++q) {
                            const double vx = Xa[i * ldx + xcols[q]] / g.ls[d], vz = Xb[j * ldx + xcols[q]] / g.ls[d];
                            xz += vx * vz; xx += vx * vx; zz += vz * vz;
                        }
                        r2 = (-2.0 * xz + xx) + zz;
                    } else {
                        const double dz0 = ux - uz;
                        r2 = dz0 * dz0;
                        for (int q = g.xrow[d]; q < g.xrow[d] + g.nxc[d]; ++q) {
                            const double dz = Xa[i * ldx + xcols[q]] / g.ls[d] - Xb[j * ldx + xcols[q]] / g.ls[d];
                            r2 += dz * dz;
                        }
                    }
                } else if (FORM == 1) r2 = ((-2.0 * ux) * uz + ux * ux) + uz * uz;  // gpflow square_distance
                else { const double dz = ux - uz; r2 = dz * dz; }
                k = g.bv[d] * exp(-0.5 * r2) - ca * Bcn[(int64_t)d * b_ld + j];
            }
        } else {
            const int xi = (int)Axs[(int64_t)d * a_ld + i];
            k = DIAG ? tables[g.tab_off[d] + g.ncat[d] * g.ncat[d] + xi]
                     : tables[g.tab_off[d] + xi * g.ncat[d] + (int)Bxs[(int64_t)d * b_ld + j]];
        }
        if (FORM == 1) {
            double kp = 1.0;
            for (int p = 0; p <= R; ++p) { acc[p] += kp; kp *= k; }
        } else {
            for (int q = R - 1; q >= 1; --q) acc[q] = acc[q] + k * acc[q - 1];
            if (R >= 1) acc[0] += k;
        }
    }
    double K;
    if (FORM == 1) {
        double e[OAK_MAX_DIMS + 1];
        e[0] = 1.0;
        for (int n = 1; n <= R; ++n) {
            double t = 0.0;
            for (int kk = 1; kk <= n; ++kk) t += (((kk - 1) & 1) ? -1.0 : 1.0) * e[n - kk] * acc[kk];
            e[n] = (1.0 / n) * t;
        }
        K = 0.0;
        for (int n = 0; n <= R; ++n) K += w[n] * e[n];
    } else {
        K = w[0];
        for (int r = 1; r <= R; ++r) K += w[r] * acc[r - 1];
    }
    out[DIAG ? i : i * ldo + j] = K;
}

int gram_generic(oak_ctx* ctx, const PreparedKernel& pk, int form, const double* dXa, const Feat& A, int64_t na, const double* dXb,
                 const Feat& B, int64_t nb, int32_t ldx, double* d_out, int64_t ldo, bool diag) {
    if (na <= 0 || (!diag && nb <= 0)) return OAK_OK;
    GenericDims g;
    memset(&g, 0, sizeof(g));
    g.D = pk.dd.D; g.R = (int)pk.w_full.size() - 1;
    for (int d = 0; d < g.D; ++d) {
        g.type[d] = pk.dd.type[d]; g.col[d] = pk.dd.col[d]; g.ncat[d] = pk.dd.ncat[d]; g.tab_off[d] = pk.dd.tab_off[d];
        g.ls[d] = pk.dm.ls[d]; g.bv[d] = pk.dd.bv[d];
    }
    for (int d = 0; d < g.D; ++d) { g.xrow[d] = pk.dd.xrow[d]; g.nxc[d] = pk.dd.nxc[d]; }      // (a component description renumbers its dims)
    double* d_w = nullptr;
    int* d_xc = nullptr;
    OAK_CHECK(get_buf_t(ctx, "gram_generic_w", (size_t)OAK_MAX_DIMS + 1, &d_w));
    OAK_CHECK(get_buf_t(ctx, "gram_generic_cols", pk.extra_cols.size() + 1, &d_xc));
    OAK_CHECK(copy_sync(ctx, d_w, pk.w_full.data(), sizeof(double) * pk.w_full.size(), hipMemcpyHostToDevice));
    if (!pk.extra_cols.empty()) OAK_CHECK(copy_sync(ctx, d_xc, pk.extra_cols.data(), sizeof(int) * pk.extra_cols.size(), hipMemcpyHostToDevice));
    if (dXb == nullptr) dXb = dXa;
#define OAK_GG(F, DG, GRID) gram_generic_kernel<F, DG><<<GRID, 256, 0, ctx->stream>>>(g, d_w, d_xc, pk.d_tables, dXa, A.xs, A.cn, A.ld, na, dXb, B.xs, B.cn, B.ld, nb, ldx, d_out, ldo)
    if (diag) {
        const dim3 grid((unsigned)((na + 255) / 256));
        if (form == 1) OAK_GG(1, true, grid); else OAK_GG(0, true, grid);
    } else {
        const dim3 grid((unsigned)((nb + 255) / 256), (unsigned)na);
        if (form == 1) OAK_GG(1, false, grid); else OAK_GG(0, false, grid);
    }
#undef OAK_GG
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

int gram_diag(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, double* d_out, double* d_sum_accum) {
    const int64_t n = A.n;
    if (n <= 0) return OAK_OK;
    const unsigned g = (unsigned)((n + 255) / 256);
#define OAK_DIAG_CASE(RR) case RR: gram_diag_kernel<RR><<<g, 256, 0, ctx->stream>>>(pk.dd, pk.d_tables, A.xs, A.cn, A.ld, n, d_out); break;
    switch (template_depth(pk.dd.R)) {
        OAK_DIAG_CASE(0) OAK_DIAG_CASE(1) OAK_DIAG_CASE(2) OAK_DIAG_CASE(3) OAK_DIAG_CASE(4)
        OAK_DIAG_CASE(5) OAK_DIAG_CASE(6) OAK_DIAG_CASE(7) OAK_DIAG_CASE(8) OAK_DIAG_CASE(12) OAK_DIAG_CASE(16)
        OAK_DIAG_CASE(24) OAK_DIAG_CASE(32)
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

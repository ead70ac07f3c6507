// Dense fp64 linear algebra for the SGPR/GPR solve path (replaces the tf.linalg calls mirrored at
// oak/utils.py:187-198: cholesky, triangular_solve, matmul).
//
//  * syrk_panel : Phi += P^T P for a row panel P [nrows x M] of Kfu, fp64 MFMA (v_mfma_f64_16x16x4),
//                 64x64 blocks of the upper triangle, four per workgroup (descriptor table) x split-N, LDS-staged operands,
//                 partials reduced in fixed order.
//                 This is the dominant kernel of an ELBO evaluation: M(M+1)N flops.
//  (Cholesky, triangular solves and the general MFMA GEMM live in factor.hip.)
#include "oak_internal.h"
#include <utility>
#include <vector>
#include <cstdlib>

namespace oak {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// SYRK
// ---------------------------------------------------------------------------------------------
constexpr int SY_T = 128;          // Phi tile edge
constexpr int SY_KB_DEFAULT = 32;  // panel rows per LDS stage
constexpr int SY_LD = SY_T + 16;   // LDS row stride (doubles): k-group rows land 32 banks apart -> conflict-free ds_read_b64

// Work decomposition.  The output is handled in 64 x 64 blocks, four per workgroup (one per wave), drawn from at most four
// staged 64-column panels Q0..Q3 of the Kuf panel:
//   * an off-diagonal 128 x 128 tile (bi < bj) is the usual 2 x 2 arrangement: Q = {2bi, 2bi+1, 2bj, 2bj+1};
//   * the diagonal tiles need only three blocks each -- (2I,2I), (2I,2I+1), (2I+1,2I+1) -- and of the two diagonal 64-blocks
//     only the upper 16 x 16 tiles (10 of 16).  Two diagonal tiles share a workgroup (panels p0..p3): wave w takes the 10 tiles
//     of the diagonal block of p_w plus HALF of its tile's off-diagonal block (32 rows, 8 tiles) -- 18 MFMAs per k-step from the
//     same eight LDS fragment reads an off-diagonal wave makes for 16 (syrk_diag_body).  M = 1024: 28 + 4 workgroups per row
//     split; the older packing (full diagonal blocks, three per tile: 28 + 6, still used by the fp32 variant) did 5.9 % more MFMAs.
// A descriptor table (built on the host per ntile, syrk_descriptors) gives each workgroup its panels and each wave its
// two LDS panels and its output block; the inner loop only sees two wave-uniform LDS offsets.
constexpr int SY_DESC = 16;        // ints per workgroup: c[4], then per wave {ia | ib << 2 | store << 4 | diagonal-pair type << 5, row block, col block}

// Workgroup of two diagonal 128-tiles (panels c[0..3]; see the decomposition above).  Same staging as the off-diagonal body.
// DK = panel rows per stage of this body: 18 accumulator tiles (144 VGPRs) leave room for 16-row staging registers, not 32.
template <int SY_KB, int DK>
__device__ __forceinline__ void syrk_diag_body(const double* __restrict__ P, int64_t ldp, const int* __restrict__ dsc, int64_t r0, int64_t r1,
                                               double* __restrict__ dst, int64_t Mp, int accumulate, double (&S)[2][SY_KB * SY_LD]) {
    static_assert(DK <= SY_KB && DK % 4 == 0, "stage depth");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* own = S[wave >> 1] + 64 * (wave & 1);
    const double* oth = S[wave >> 1] + 64 * ((wave & 1) ^ 1);
    double4_t accD[10], accO[8];
#pragma unroll
    for (int q = 0; q < 10; ++q) accD[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 8; ++q) accO[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int lrow = (2 * tid) >> 7;
    const int lcol = (2 * tid) & 127;
    const double* pa = P + (int64_t)dsc[lcol >> 6] * 64 + (lcol & 63);
    const double* pb = P + (int64_t)dsc[2 + (lcol >> 6)] * 64 + (lcol & 63);
    constexpr int NQ = DK / 4;
    double2 ra[NQ], rb[NQ];
    // raw loads (clamped row index) that stay in flight under the MFMAs of the current stage; rows past the end are zeroed when
    // the registers are written to LDS, so nothing forces a wait on them earlier
    int64_t nloaded = r0;
    auto load_stage = [&](int64_t n0) {
        nloaded = n0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int64_t row = n0 + lrow + 4 * q;
            const int64_t rc = row < r1 ? row : r1 - 1;
            ra[q] = *reinterpret_cast<const double2*>(pa + rc * ldp);
            rb[q] = *reinterpret_cast<const double2*>(pb + rc * ldp);
        }
    };
    if (r0 < r1) load_stage(r0);
    const int fr = lane & 15, fk = lane >> 4;
    // The off-diagonal half-block: rows 16*(g0+g), g < 2, of the tile's EVEN panel against all four column tiles of its ODD panel.
    // Even wave: rows from its own panel, columns from the other; odd wave: the reverse -- expressed through two wave-uniform
    // LDS pointers so that both run the same instruction stream (ten fragment reads, 18 MFMAs per k-step).
    const double* orow = (wave & 1) ? oth + 32 : own;
    const double* ocol = (wave & 1) ? own : oth;
    for (int64_t n0 = r0; n0 < r1; n0 += DK) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const bool ok = nloaded + lrow + 4 * q < r1;
            *reinterpret_cast<double2*>(&S[0][(lrow + 4 * q) * SY_LD + lcol]) = make_double2(ok ? ra[q].x : 0.0, ok ? ra[q].y : 0.0);
            *reinterpret_cast<double2*>(&S[1][(lrow + 4 * q) * SY_LD + lcol]) = make_double2(ok ? rb[q].x : 0.0, ok ? rb[q].y : 0.0);
        }
        __syncthreads();
        load_stage((n0 + DK < r1) ? n0 + DK : r0);
#pragma unroll
        for (int kk = 0; kk < DK / 4; ++kk) {
            double o[4], a[2], b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) o[g] = own[(4 * kk + fk) * SY_LD + 16 * g + fr];
#pragma unroll
            for (int g = 0; g < 2; ++g) a[g] = orow[(4 * kk + fk) * SY_LD + 16 * g + fr];
#pragma unroll
            for (int h = 0; h < 4; ++h) b[h] = ocol[(4 * kk + fk) * SY_LD + 16 * h + fr];
            int q = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int h = g; h < 4; ++h, ++q) accD[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(o[g], o[h], accD[q], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int h = 0; h < 4; ++h) accO[4 * g + h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g], b[h], accO[4 * g + h], 0, 0, 0);
        }
        __syncthreads();
    }
    if (!((dsc[4 + 3 * wave] >> 4) & 1)) return;          // second tile absent (odd tile count): padding waves
    const int64_t pown = dsc[wave], prow = dsc[wave & 2], pcol = dsc[(wave & 2) + 1];
    int q = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = g; h < 4; ++h, ++q)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                double* e = dst + (pown * 64 + 16 * g + 4 * reg + fk) * Mp + pown * 64 + 16 * h + fr;
                const double v = accD[q][reg];
                *e = accumulate ? (*e + v) : v;
            }
    const int g0 = 2 * (wave & 1);
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                double* e = dst + (prow * 64 + 16 * (g0 + g) + 4 * reg + fk) * Mp + pcol * 64 + 16 * h + fr;
                const double v = accO[4 * g + h][reg];
                *e = accumulate ? (*e + v) : v;
            }
}

// TWO: the panel holds two row sets back to back -- rows [0, rowsA) served by splits [0, nsplitA), rows [rowsA, nrows) by the
// others -- each split still writing its own partial: two independent Gram matrices from ONE launch (one ramp-up and one ragged
// end instead of two; the Sobol pass's positive- and negative-weight pair rows).  The <KB, false> instantiation is the kernel as
// it always was.
template <int SY_KB, bool TWO = false>
__global__ void __launch_bounds__(256, 2)
syrk_kernel(const double* __restrict__ P, int64_t ldp, int64_t nrows, const int* __restrict__ desc, int nwg, int nsplit,
            int64_t rows_per_split, double* __restrict__ part, int64_t Mp, int accumulate, int xcd_map, int nsplitA = 0, int64_t rowsA = 0) {
    __shared__ __attribute__((aligned(16))) double S[2][SY_KB * SY_LD];      // [Q0 | Q1] and [Q2 | Q3]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware decode: workgroup b runs on XCD b % 8 (observed dispatch rule; affects speed only).  XCD x owns the row
    // splits [x*s, (x+1)*s), s = nsplit/8, and walks them in order, all workgroups of one split before the next: the
    // ~64 resident workgroups of an XCD therefore stream the SAME panel rows at the same time and each row chunk is
    // fetched into that XCD's L2 once instead of once per tile pair (9x at M = 1024).
    int unit, split;
    if (xcd_map) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int sl = j / nwg;
        unit = j - sl * nwg;
        split = xcd * (nsplit >> 3) + sl;
    } else {
        unit = blockIdx.x / nsplit;
        split = blockIdx.x - unit * nsplit;
    }
    const int* dsc = desc + unit * SY_DESC;
    int64_t row_lo = (int64_t)split * rows_per_split, row_end = nrows;
    if constexpr (TWO) {
        if (split < nsplitA) row_end = rowsA;
        else row_lo = rowsA + (int64_t)(split - nsplitA) * rows_per_split;
    }
    if ((dsc[4] >> 5) & 1) {            // workgroup-uniform: a pair of diagonal tiles
        const int64_t q0 = row_lo;
        const int64_t q1 = (q0 + rows_per_split < row_end) ? q0 + rows_per_split : row_end;
        syrk_diag_body<SY_KB, SY_KB>(P, ldp, dsc, q0, q1, part + (int64_t)split * Mp * Mp, Mp, accumulate, S);
        return;
    }
    const int wcode = dsc[4 + 3 * wave], wrb = dsc[5 + 3 * wave], wcb = dsc[6 + 3 * wave];
    const int ia = wcode & 3, ib = (wcode >> 2) & 3, wstore = (wcode >> 4) & 1;
    const double* Sa = S[ia >> 1] + 64 * (ia & 1);
    const double* Sb = S[ib >> 1] + 64 * (ib & 1);
    const int64_t r0 = row_lo;
    int64_t r1 = r0 + rows_per_split;
    if (r1 > row_end) r1 = row_end;

    double4_t acc[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[g][h] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // global -> register staging: element e = 2*tid + 512*q -> row e/128, col e%128 (one wave-load = one 1 KiB row of
    // two 64-column panels)
    const int lrow = (2 * tid) >> 7;          // 0..3
    const int lcol = (2 * tid) & 127;
    const double* pa = P + (int64_t)dsc[lcol >> 6] * 64 + (lcol & 63);
    const double* pb = P + (int64_t)dsc[2 + (lcol >> 6)] * 64 + (lcol & 63);
    constexpr int NQ = SY_KB / 4;
    double2 ra[NQ], rb[NQ];
    // Branch-free staging: clamped row index + select, so all 2*NQ loads issue back-to-back and stay in flight under the
    // MFMAs of the current stage (conditional loads made hipcc wait vmcnt(0) inside the load block: -50% throughput).
    auto load_stage = [&](int64_t n0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int64_t row = n0 + lrow + 4 * q;
            const int64_t rc = row < r1 ? row : r1 - 1;
            const double2 va = *reinterpret_cast<const double2*>(pa + rc * ldp);
            const double2 vb = *reinterpret_cast<const double2*>(pb + rc * ldp);
            const bool ok = row < r1;
            ra[q] = make_double2(ok ? va.x : 0.0, ok ? va.y : 0.0);
            rb[q] = make_double2(ok ? vb.x : 0.0, ok ? vb.y : 0.0);
        }
    };
    if (r0 < r1) load_stage(r0);
    const int fr = lane & 15, fk = lane >> 4;
    for (int64_t n0 = r0; n0 < r1; n0 += SY_KB) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            *reinterpret_cast<double2*>(&S[0][(lrow + 4 * q) * SY_LD + lcol]) = ra[q];
            *reinterpret_cast<double2*>(&S[1][(lrow + 4 * q) * SY_LD + lcol]) = rb[q];
        }
        __syncthreads();
        {
            const int64_t nn = (n0 + SY_KB < r1) ? n0 + SY_KB : r0;   // last stage re-reads a valid row range; result unused
            load_stage(nn);
        }
#pragma unroll
        for (int kk = 0; kk < SY_KB / 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g] = Sa[(4 * kk + fk) * SY_LD + 16 * g + fr];
#pragma unroll
            for (int h = 0; h < 4; ++h) b[h] = Sb[(4 * kk + fk) * SY_LD + 16 * h + fr];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int h = 0; h < 4; ++h)
                    acc[g][h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g], b[h], acc[g][h], 0, 0, 0);
        }
        __syncthreads();
    }
    if (!wstore) return;       // padding wave of the last diagonal-block workgroup
    // epilogue: f64 16x16x4 C/D layout: col = lane&15, row = (lane>>4) + 4*reg
    double* dst = part + (int64_t)split * Mp * Mp;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t row = (int64_t)wrb * 64 + 16 * g + 4 * reg + fk;
                const int64_t col = (int64_t)wcb * 64 + 16 * h + fr;
                double* q = dst + row * Mp + col;
                const double v = acc[g][h][reg];
                *q = accumulate ? (*q + v) : v;
            }
}

// Fixed-order sum of the split partials.  Only tiles of the upper block triangle were written (of a diagonal 128-tile only
// the 16 x 16 tiles on or above the diagonal are relied on); each thread sums one element of such a tile over the splits
// (coalesced reads) and stores it to (i, j) and, off the diagonal tiles, to its mirror (j, i).
__global__ void __launch_bounds__(256) syrk_reduce_kernel(const double* __restrict__ part, int nsplit, int64_t M, int64_t Mp,
                                                          int ntile, double* __restrict__ phi, int accumulate) {
    int bi = 0, rem = blockIdx.y;
    while (rem >= ntile - bi) { rem -= ntile - bi; ++bi; }
    const int bj = bi + rem;
    const int e = blockIdx.x * 256 + threadIdx.x;          // element within the 128 x 128 tile
    const int64_t i = (int64_t)bi * SY_T + (e >> 7), j = (int64_t)bj * SY_T + (e & 127);
    if (i >= M || j >= M) return;
    // diagonal tile: only the 16 x 16 tiles on or above the diagonal were computed; an element below them is written as the
    // mirror of (j, i) by the thread that owns that one, so every read of the partials is coalesced
    const bool diag = bi == bj;
    if (diag && (i >> 4) > (j >> 4)) return;
    const bool mirror = !diag || (i >> 4) < (j >> 4);
    const int64_t src = i * Mp + j;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int sp = 0;
    for (; sp + 4 <= nsplit; sp += 4) {                      // four independent chains, combined in a fixed order
        s0 += part[(int64_t)(sp + 0) * Mp * Mp + src];
        s1 += part[(int64_t)(sp + 1) * Mp * Mp + src];
        s2 += part[(int64_t)(sp + 2) * Mp * Mp + src];
        s3 += part[(int64_t)(sp + 3) * Mp * Mp + src];
    }
    for (; sp < nsplit; ++sp) s0 += part[(int64_t)sp * Mp * Mp + src];
    const double s = (s0 + s1) + (s2 + s3);
    if (accumulate) {
        phi[i * M + j] += s;
        if (mirror) phi[j * M + i] += s;
    } else {
        phi[i * M + j] = s;
        if (mirror) phi[j * M + i] = s;
    }
}

static bool syrk_xcd_mapping() {
    const char* e = getenv("OAK_SYRK_XCD");
    return !(e && atoi(e) == 0);
}

// Number of row splits.  XCD mode: 8*s splits, s per XCD, chosen so that s*npairs workgroups fill the XCD's resident slots
// in (nearly) whole rounds while every workgroup still streams >= 2048 rows.
static int syrk_wg_per_split(int ntile, bool diag_pairs) {
    return ntile * (ntile - 1) / 2 + (diag_pairs ? (ntile + 1) / 2 : (3 * ntile + 3) / 4);
}

// Descriptor table of syrk_kernel for ntile 128-tiles (see the kernel's header comment).
static std::vector<int> syrk_descriptors(int ntile, bool diag_pairs) {
    std::vector<int> d;
    auto emit = [&](const int (&c)[4], const int (&w)[4][4]) {      // w[k] = {ia, ib, row block, col block} or ia < 0 for padding
        for (int k = 0; k < 4; ++k) d.push_back(c[k]);
        for (int k = 0; k < 4; ++k) {
            const bool pad = w[k][0] < 0;
            const int* src = pad ? w[0] : w[k];
            d.push_back(src[0] | (src[1] << 2) | ((pad ? 0 : 1) << 4));
            d.push_back(src[2]);
            d.push_back(src[3]);
        }
    };
    // the diagonal pairs first: they run ~12 % longer than the others, so they should not be the last to start
    if (diag_pairs) {                                               // two diagonal tiles per workgroup (syrk_diag_body)
        for (int I = 0; I < ntile; I += 2) {
            const bool two = I + 1 < ntile;
            d.push_back(2 * I); d.push_back(2 * I + 1); d.push_back(two ? 2 * I + 2 : 2 * I + 1); d.push_back(two ? 2 * I + 3 : 2 * I + 1);
            for (int k = 0; k < 4; ++k) {
                const int store = (k < 2 || two) ? 1 : 0;
                d.push_back((store << 4) | (1 << 5));
                d.push_back(0); d.push_back(0);
            }
        }
    }
    for (int bi = 0; bi < ntile; ++bi)
        for (int bj = bi + 1; bj < ntile; ++bj) {
            const int c[4] = {2 * bi, 2 * bi + 1, 2 * bj, 2 * bj + 1};
            const int w[4][4] = {{0, 2, 2 * bi, 2 * bj}, {0, 3, 2 * bi, 2 * bj + 1}, {1, 2, 2 * bi + 1, 2 * bj}, {1, 3, 2 * bi + 1, 2 * bj + 1}};
            emit(c, w);
        }
    if (diag_pairs) return d;
    std::vector<std::pair<int, int>> blocks;                        // upper 64-blocks of the diagonal tiles, in sequence
    for (int I = 0; I < ntile; ++I) { blocks.push_back({2 * I, 2 * I}); blocks.push_back({2 * I, 2 * I + 1}); blocks.push_back({2 * I + 1, 2 * I + 1}); }
    for (size_t g0 = 0; g0 < blocks.size(); g0 += 4) {
        int c[4] = {0, 0, 0, 0}, nc = 0;
        int w[4][4];
        auto panel = [&](int id) { for (int k = 0; k < nc; ++k) if (c[k] == id) return k; c[nc] = id; return nc++; };
        for (int k = 0; k < 4; ++k) {
            if (g0 + k < blocks.size()) {
                const int r = blocks[g0 + k].first, cc = blocks[g0 + k].second;
                w[k][0] = panel(r); w[k][1] = panel(cc); w[k][2] = r; w[k][3] = cc;
            } else {
                w[k][0] = -1; w[k][1] = 0; w[k][2] = 0; w[k][3] = 0;
            }
        }
        for (int k = nc; k < 4; ++k) c[k] = c[nc - 1];              // unused panel slots stage a valid block again
        emit(c, w);
    }
    return d;
}

int syrk_plan_splits(oak_ctx* ctx, int64_t M, int64_t nrows) {
    const int ntile = (int)((M + SY_T - 1) / SY_T);
    const int npairs = syrk_wg_per_split(ntile, true);
    int per_cu = 2;
    if (const char* e = getenv("OAK_SYRK_WG_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 4) per_cu = v; }
    if (!syrk_xcd_mapping()) {
        int nsplit = (ctx->num_cu * per_cu) / npairs;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > 256) nsplit = 256;
        return nsplit;
    }
    if (const char* e = getenv("OAK_SYRK_NSPLIT")) { int v = atoi(e); if (v >= 8 && v <= 256 && v % 8 == 0) return v; }   // tuning knob
    // Cost model fitted to sweeps on MI355X (tools/dev_syrk_sweep.py, profiles/r02_syrk_split_sweeps.txt, r02_syrk_diagpair_sweeps.txt).
    // A CU holds two workgroups, which share its one DP pipe: an XCD retires its W = s * npairs workgroups in batches of
    // 2 * CUs, each batch taking two workgroup lengths (W <= CUs: one length).  A length costs the rows a workgroup streams plus
    // ~120 row-equivalents of prologue / epilogue; ramp-up, the ragged end and workgroups that started together running in
    // phase cost another ~1.7 lengths (with 28 + 4 workgroups per split at M = 1024 this term is what separates 8 splits,
    // 34.6 ms, from 128-256, 17.4 ms); every 8 splits add M x M partials to the fixed-order reduction (~100 row-equivalents
    // at M <= 1024, growing with M^2).
    (void)per_cu;
    // a partitioned forward pass whose SYRK stays on main_part has the side stream's compute units taken out of every XCD
    const int cus_total = (ctx->part_active && !ctx->part_syrk_full && ctx->route != 2) ? ctx->num_cu - ctx->part_cus : ctx->num_cu;
    const int cus_per_xcd = cus_total / 8 > 0 ? cus_total / 8 : 1;
    const double red_m2 = 100.0 * ((double)ntile * SY_T / 1024.0) * ((double)ntile * SY_T / 1024.0);
    const double red = red_m2 > 100.0 ? red_m2 : 100.0;
    int best_s = 1;
    double best_cost = 0.0;
    const double part_bytes = 8.0 * 8.0 * (double)ntile * SY_T * (double)ntile * SY_T;     // partials of 8 splits
    for (int sp = 1; sp <= 16; ++sp) {     // beyond 128 splits the SYRK gains < 1 % (17.69 / 17.65 / 17.51 ms at 128 / 192 / 256) and the reduction pays for it
        const int64_t rps = (nrows + 8 * sp - 1) / (8 * sp);
        if (sp > 1 && (rps < 256 || part_bytes * sp > 4.0 * 1024 * 1024 * 1024)) break;   // <= 4 GiB of partials
        const int W = sp * npairs;
        const int rounds = W <= cus_per_xcd ? 1 : 2 * ((W + 2 * cus_per_xcd - 1) / (2 * cus_per_xcd));
        const double cost = ((double)rounds + 1.7) * (double)rps + 120.0 * rounds + red * sp;
        if (sp == 1 || cost < best_cost) { best_cost = cost; best_s = sp; }
    }
    // Fabric traffic (r05, profiles/r05_syrk_traffic_vs_splits.txt): at 128 splits of >= 8192 rows the workgroups of a split drift
    // ~1000 rows apart (the diagonal pairs run 12 % longer) and the rows between leader and laggard no longer fit the XCD's L2:
    // 32.5 GB fetched for an 8.6 GB panel.  Shorter splits shrink the drift in bytes -- 28.2 GB at 192, 24.8 GB at 256 -- while the
    // fixed-order reduction grows (0.125 / 0.19 / 0.24 ms) and the SYRK itself starts to lose (same box, whole step: 28.09-28.18 ms
    // at 128, 28.13-28.27 at 192, 28.33-28.45 at 208, 28.5 at 256).  192 where the time model had hit its cap: -4.3 GB of fabric
    // traffic for <= 0.1 ms of a 28 ms step; more is not free.
    if (best_s == 16 && (nrows + 127) / 128 >= 8192 && part_bytes * 24 <= 4.0 * 1024 * 1024 * 1024) return 192;
    return 8 * best_s;
}

// device descriptor table of the SYRK kernels for `ntile` 128-tiles (rebuilt when ntile changes); shared with the fp32 variant
int syrk_descriptor_table(oak_ctx* ctx, int ntile, int** d_desc_out, int* npairs_out, bool diag_pairs) {
    const int npairs = syrk_wg_per_split(ntile, diag_pairs);
    int* d_desc = nullptr;
    OAK_CHECK(get_buf_t(ctx, diag_pairs ? "syrk_desc" : "syrk_desc_blocks", (size_t)npairs * SY_DESC, &d_desc));
    int& cached = diag_pairs ? ctx->syrk_desc_ntile : ctx->syrk_desc_blocks_ntile;
    if (cached != ntile) {
        const std::vector<int> h = syrk_descriptors(ntile, diag_pairs);
        OAK_REQUIRE((int)h.size() == npairs * SY_DESC, "syrk: descriptor table size mismatch");
        OAK_HIP_CHECK(hipMemcpyAsync(d_desc, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice, ctx->stream));
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        cached = ntile;
    }
    *d_desc_out = d_desc; *npairs_out = npairs;
    return OAK_OK;
}

int syrk_panel(oak_ctx* ctx, const double* d_panel, int64_t ldp, int64_t nrows, int64_t M, double* d_part, int nsplit, bool accumulate) {
    const int ntile = (int)((M + SY_T - 1) / SY_T);
    const int64_t Mp = (int64_t)ntile * SY_T;
    OAK_REQUIRE(ldp == Mp, "syrk: panel stride %lld must equal padded M %lld", (long long)ldp, (long long)Mp);
    int* d_desc = nullptr;
    int npairs = 0;
    OAK_CHECK(syrk_descriptor_table(ctx, ntile, &d_desc, &npairs, true));
    int kb = SY_KB_DEFAULT;
    if (const char* e = getenv("OAK_SYRK_KB")) { int v = atoi(e); if (v == 8 || v == 16 || v == 32) kb = v; }
    int64_t rps = (nrows + nsplit - 1) / nsplit;
    rps = ((rps + kb - 1) / kb) * kb;
    if (rps < kb) rps = kb;
    const unsigned grid = (unsigned)(npairs * nsplit);
    const int xm = (syrk_xcd_mapping() && (nsplit % 8) == 0) ? 1 : 0;
    if (kb == 8) syrk_kernel<8><<<grid, 256, 0, ctx->stream>>>(d_panel, ldp, nrows, d_desc, npairs, nsplit, rps, d_part, Mp, accumulate ? 1 : 0, xm);
    else if (kb == 32) syrk_kernel<32><<<grid, 256, 0, ctx->stream>>>(d_panel, ldp, nrows, d_desc, npairs, nsplit, rps, d_part, Mp, accumulate ? 1 : 0, xm);
    else syrk_kernel<16><<<grid, 256, 0, ctx->stream>>>(d_panel, ldp, nrows, d_desc, npairs, nsplit, rps, d_part, Mp, accumulate ? 1 : 0, xm);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// Two Gram matrices from one launch: rows [0, rowsA) of the panel into the partials of splits [0, nsA), rows [rowsA, rowsA + rowsB)
// into those of splits [nsA, nsA + nsB) (nsA, nsB multiples of 8; reduce each range with syrk_reduce on its part of d_part).
int syrk_panel_two(oak_ctx* ctx, const double* d_panel, int64_t ldp, int64_t rowsA, int64_t rowsB, int64_t M, double* d_part, int nsA, int nsB,
                   int64_t rows_per_split) {
    const int ntile = (int)((M + SY_T - 1) / SY_T);
    const int64_t Mp = (int64_t)ntile * SY_T;
    OAK_REQUIRE(ldp == Mp && nsA % 8 == 0 && nsB % 8 == 0 && nsA + nsB >= 8 && rows_per_split % SY_KB_DEFAULT == 0 &&
                    (int64_t)nsA * rows_per_split >= rowsA && (int64_t)nsB * rows_per_split >= rowsB,
                "syrk_panel_two: bad split plan");
    int* d_desc = nullptr;
    int npairs = 0;
    OAK_CHECK(syrk_descriptor_table(ctx, ntile, &d_desc, &npairs, true));
    const int nsplit = nsA + nsB;
    syrk_kernel<32, true><<<(unsigned)(npairs * nsplit), 256, 0, ctx->stream>>>(d_panel, ldp, rowsA + rowsB, d_desc, npairs, nsplit, rows_per_split,
                                                                               d_part, Mp, 0, 1, nsA, rowsA);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

int syrk_reduce(oak_ctx* ctx, const double* d_part, int nsplit, int64_t M, double* d_phi, bool accumulate) {
    const int ntile = (int)((M + SY_T - 1) / SY_T);
    const int64_t Mp = (int64_t)ntile * SY_T;
    dim3 grid((unsigned)(SY_T * SY_T / 256), (unsigned)(ntile * (ntile + 1) / 2));
    syrk_reduce_kernel<<<grid, 256, 0, ctx->stream>>>(d_part, nsplit, M, Mp, ntile, d_phi, accumulate ? 1 : 0);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) transpose_kernel(const double* __restrict__ A, int64_t rows, int64_t cols, int64_t lda,
                                                        double* __restrict__ B, int64_t ldb) {
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = r0 + k, c = c0 + tx;
        t[k][tx] = (r < rows && c < cols) ? A[r * lda + c] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t r = c0 + k, c = r0 + tx;     // B is cols x rows
        if (r < cols && c < rows) B[r * ldb + c] = t[tx][k];
    }
}
int transpose(oak_ctx* ctx, const double* dA, int64_t rows, int64_t cols, int64_t lda, double* dB, int64_t ldb) {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    transpose_kernel<<<grid, 256, 0, ctx->stream>>>(dA, rows, cols, lda, dB, ldb);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__global__ void add_diag_kernel(double* A, int64_t n, int64_t lda, double v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[i * lda + i] += v;
}
int add_diag(oak_ctx* ctx, double* dA, int64_t n, int64_t lda, double v) {
    add_diag_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(dA, n, lda, v);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__global__ void scale_add_eye_kernel(const double* __restrict__ W, int64_t n, double s, double* __restrict__ B) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= n) return;
    if (i < n) B[i * n + j] = W[i * n + j] * s + (i == j ? 1.0 : 0.0);
    else B[i * n + j] = W[i * n + j];               // rows past the square ride along unchanged (right-hand sides of the solve)
}
int scale_add_eye(oak_ctx* ctx, const double* dW, int64_t n, double s, double* dB, int extra_rows) {
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)(n + extra_rows));
    scale_add_eye_kernel<<<grid, 256, 0, ctx->stream>>>(dW, n, s, dB);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}
__global__ void scaled_copy_kernel(double a, const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = a * src[i];
}
int scaled_copy(oak_ctx* ctx, double a, const double* d_src, double* d_dst, int64_t n) {      // dst = a * src
    if (n <= 0) return OAK_OK;
    scaled_copy_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(a, d_src, d_dst, n);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__device__ __forceinline__ double block_reduce_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) { for (int k = 0; k < (int)(blockDim.x >> 6); ++k) r += sh[k]; }
    __syncthreads();
    return r;   // valid on thread 0
}

// mode 0: sum x ; 1: sum x^2 ; 2: sum log x ; x_k = x[k*stride]
__global__ void __launch_bounds__(256) reduce_stage1(const double* __restrict__ x, int64_t n, int64_t stride, int mode,
                                                     double* __restrict__ partial) {
    __shared__ double sh[4];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * per;
    int64_t e = b + per; if (e > n) e = n;
    double acc = 0.0;
    for (int64_t k = b + threadIdx.x; k < e; k += blockDim.x) {
        const double v = x[k * stride];
        acc += (mode == 0) ? v : (mode == 1 ? v * v : log(v));
    }
    const double r = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ void __launch_bounds__(256) reduce_stage2(const double* __restrict__ partial, int np, double* __restrict__ out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int k = threadIdx.x; k < np; k += blockDim.x) acc += partial[k];
    const double r = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = r;
}
int reduce_sum(oak_ctx* ctx, const double* d_x, int64_t n, double* d_out, int mode, int64_t stride) {
    double* d_part = nullptr;
    int nb = (int)((n + 4095) / 4096);
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    OAK_CHECK(get_buf_t(ctx, "reduce_part", 512, &d_part));
    reduce_stage1<<<nb, 256, 0, ctx->stream>>>(d_x, n, stride, mode, d_part);
    reduce_stage2<<<1, 256, 0, ctx->stream>>>(d_part, nb, d_out);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__global__ void __launch_bounds__(256) dot_stage1(const double* __restrict__ x, const double* __restrict__ y, int64_t n,
                                                  double* __restrict__ partial) {
    __shared__ double sh[4];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b = (int64_t)blockIdx.x * per;
    int64_t e = b + per; if (e > n) e = n;
    double acc = 0.0;
    for (int64_t k = b + threadIdx.x; k < e; k += blockDim.x) acc = __builtin_fma(x[k], y[k], acc);
    const double r = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
int dot(oak_ctx* ctx, const double* d_x, const double* d_y, int64_t n, double* d_out) {
    double* d_part = nullptr;
    int nb = (int)((n + 4095) / 4096);
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    OAK_CHECK(get_buf_t(ctx, "reduce_part", 512, &d_part));
    dot_stage1<<<nb, 256, 0, ctx->stream>>>(d_x, d_y, n, d_part);
    reduce_stage2<<<1, 256, 0, ctx->stream>>>(d_part, nb, d_out);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// y_i = sum_j A_ij x_j : one wave per row
__global__ void __launch_bounds__(256) gemv_rows_kernel(const double* __restrict__ A, int64_t rows, int64_t cols, int64_t lda,
                                                        const double* __restrict__ x, double* __restrict__ y, int sq) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    double acc = 0.0;
    const double* a = A + row * lda;
    if (sq) { for (int64_t j = lane; j < cols; j += 64) acc = __builtin_fma(a[j], a[j], acc); }
    else    { for (int64_t j = lane; j < cols; j += 64) acc = __builtin_fma(a[j], x[j], acc); }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) y[row] = acc;
}
int gemv_rows(oak_ctx* ctx, const double* dA, int64_t rows, int64_t cols, int64_t lda, const double* d_x, double* d_y) {
    if (rows <= 0) return OAK_OK;
    gemv_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, ctx->stream>>>(dA, rows, cols, lda, d_x, d_y, 0);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}
int row_sumsq(oak_ctx* ctx, const double* dA, int64_t rows, int64_t cols, int64_t lda, double* d_out) {
    if (rows <= 0) return OAK_OK;
    gemv_rows_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, ctx->stream>>>(dA, rows, cols, lda, nullptr, d_out, 1);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

__global__ void axpy_kernel(double a, const double* __restrict__ x, double* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __builtin_fma(a, x[i], y[i]);
}
int axpy(oak_ctx* ctx, double a, const double* x, double* y, int64_t n) {
    axpy_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(a, x, y, n);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}
__global__ void scale_kernel(double a, double* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= a;
}
int scale_vec(oak_ctx* ctx, double a, double* x, int64_t n) {
    scale_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(a, x, n);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

}  // namespace oak

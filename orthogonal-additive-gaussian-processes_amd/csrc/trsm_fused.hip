// Many-row forward triangular solve  X L^T = B  (row n of X = L^-1 K(Z, x_n): GPflow's  A = triangular_solve(L, Kuf),
// oak/utils.py:189, and the two solves of predict_f) as ONE launch.
//
// A workgroup owns 64 rows of the panel from the first column block to the last (left-looking per row tile: no dependency
// between workgroups, so no launch boundary per block and no read-modify-write of the block by separate kernels).  For column
// block j (128 columns) it forms the residual  T = B_j - X_{<j} L_{j,<j}^T  with fp64 MFMAs and multiplies it by the inverse of
// the 128 x 128 diagonal block -- the arithmetic of the blocked solve this replaces (the residual is cancelled first, only the
// diagonal block is inverted, so the error stays that of a substitution with 128-wide pivots) -- but T never leaves the
// registers: the products are formed TRANSPOSED (D = L_block X^T), and the fp64 MFMA's D layout (lane (fi, fk), register v
// holds D[4v + fk][fi]) is exactly its B-operand layout for four consecutive k, so the residual feeds the diagonal product
// and the finished block feeds the next block's update without any data movement.
//
// The triangular factor reaches the kernel as a PACK: per stage one 128 x 16 tile (16 KiB, the LDS image itself, XOR
// swizzled for conflict-free ds_read_b64) in the order the kernel consumes them, negated / permuted / inverted by a small
// builder kernel.  All workgroups stream the same pack (4.7 MB at M = 1024), so it stays in L2.
// Stage order of block j:  8 tiles against the block finished last (X from registers), 8 (j-1) tiles against the older
// columns (X re-read from the output, 32 B per lane), 8 tiles of the inverse of the diagonal block (lower triangular:
// tile h only feeds the output tiles >= h).  MFMA count per 16 rows: M^2/128 + 9 M/8 = 8320 at M = 1024 against the
// 8192 of an exact triangular count.
#include "oak_internal.h"
#include <cstdlib>

namespace oak {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int TF_NB = 128;                  // column block
constexpr int TF_KC = 16;                   // k per stage
#ifndef OAK_TF_NW
#define OAK_TF_NW 4
#endif
constexpr int TF_NW = OAK_TF_NW;            // waves per workgroup (4: two workgroups per CU; 8: one, the pack tile shared by twice the rows)
constexpr int TF_ROWS = 16 * TF_NW;         // panel rows per workgroup (16 per wave)
constexpr int TF_TILE = TF_NB * TF_KC;      // doubles per stage tile

__host__ __device__ inline int64_t tf_stages_before(int64_t j) { return 4 * j * (j + 1); }   // block j has 8 j + 8 stages

// element (r, p) of a stage tile lives at r * 16 + (p ^ tf_swz(r)): the 16 rows of a fragment land on 16 different 8-byte
// columns, conflict-free both for ds_read_b64 (32 lanes = 16 rows x 2 k over 64 banks) and for the ds_read2_b64 pairs the
// compiler forms (16 lanes = 16 rows over 32 banks)
__host__ __device__ inline int tf_swz(int r) { return r & 15; }

// grid (8 * nb, nb): blockIdx.y = column block j, blockIdx.x = stage within the block (those beyond 8 j + 8 exit)
// REV: the factor seen by the solve is  Lt = J L^T J  (J = index reversal; n a multiple of 128): X L = B  is  (X J) Lt^T = B J, the
// same row solve on column-reversed panels -- how the transposed solve L^T x = b of many rows runs through this kernel.
template <bool REV>
__global__ void __launch_bounds__(256) trsm_pack_kernel(const double* __restrict__ L, int64_t n, int64_t ldl,
                                                        const double* __restrict__ Inv, int64_t inv_bs, int64_t inv_ld,
                                                        double* __restrict__ pack) {
    const int j = blockIdx.y, t = blockIdx.x;
    if (t >= 8 * j + 8) return;
    double* tile = pack + (tf_stages_before(j) + t) * TF_TILE;
    const int64_t j0 = (int64_t)j * TF_NB;
    const bool diag = t >= 8 * j;
    // update stages: first the block finished last (k0 = j0 - 128 ...), then the older columns from 0
    const int64_t k0 = diag ? j0 + 16 * (t - 8 * j) : (t < 8 ? j0 - TF_NB + 16 * t : 16 * (int64_t)(t - 8));
    for (int e = threadIdx.x; e < TF_TILE; e += 256) {
        const int r = e >> 4, pp = e & 15;
        const int p = pp ^ tf_swz(r);
        const int kk = p;
        const int64_t gi = j0 + r, gk = k0 + kk;
        double v;
        if (REV) {
            // Lt[gi][gk] = L[n-1-gk][n-1-gi];  inv(Lt_jj)[r][q] = inv(L_bb)[127-q][127-r] with b = nb - 1 - j
            if (diag) v = (gk <= gi) ? Inv[(int64_t)(gridDim.y - 1 - j) * inv_bs + (int64_t)(TF_NB - 1 - (gk - j0)) * inv_ld + (TF_NB - 1 - r)] : 0.0;
            else v = -L[(n - 1 - gk) * ldl + (n - 1 - gi)];
        } else if (diag) {
            if (gi < n && gk < n) v = (gk <= gi) ? Inv[(int64_t)j * inv_bs + (int64_t)r * inv_ld + (gk - j0)] : 0.0;
            else v = (gi == gk) ? 1.0 : 0.0;        // identity padding up to the next multiple of 128
        } else {
            v = (gi < n && gk < n) ? -L[gi * ldl + gk] : 0.0;
        }
        tile[e] = v;
    }
}

// Inverses of the 128 x 128 diagonal blocks of L (identity-padded past n) by forward substitution, one workgroup per block,
// thread c owning column c of the inverse: x_i = (e_c[i] - sum_{c <= k < i} L_ik x_k) / L_ii.  The column's history lives in
// LDS ([k][c]: conflict-free).  Row i of the block is fetched with two coalesced vector loads one row ahead and L_ik handed out
// by v_readlane (k is wave-uniform) -- as wave-uniform scalar loads from global memory the same loop was latency-bound at
// 0.6 ms per call.
__device__ __forceinline__ double tf_readlane(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(128) trsm_block_inverse_kernel(const double* __restrict__ L, int64_t n, int64_t ldl, double* __restrict__ inv) {
    extern __shared__ __attribute__((aligned(16))) double xs[];     // [128][128]
    const int c = threadIdx.x, lane = c & 63;
    const int64_t b0 = (int64_t)blockIdx.x * TF_NB;
    const int kmin = __builtin_amdgcn_readfirstlane(c) & ~63;        // first column of this wave: everything before it is zero
    double* out = inv + (int64_t)blockIdx.x * TF_NB * TF_NB;
    // lane l holds L[b0 + i][b0 + l] and L[b0 + i][b0 + 64 + l] of the row in hand (identity outside the matrix)
    auto load_row = [&](int i, double& ra, double& rb) {
        const int64_t gi = b0 + i;
        const bool rin = gi < n;
        ra = (rin && b0 + lane < n) ? L[gi * ldl + b0 + lane] : ((i == lane) ? 1.0 : 0.0);
        rb = (rin && b0 + 64 + lane < n) ? L[gi * ldl + b0 + 64 + lane] : ((i == 64 + lane) ? 1.0 : 0.0);
    };
    double ra, rb, na = 0.0, nb = 0.0;
    load_row(0, ra, rb);
    for (int i = 0; i < TF_NB; ++i) {
        if (i + 1 < TF_NB) load_row(i + 1, na, nb);
        double sum = (i == c) ? 1.0 : 0.0;
        const int kend = i < 64 ? i : 64;
        for (int k = kmin; k < kend; ++k) sum = __builtin_fma(-tf_readlane(ra, k), xs[k * TF_NB + c], sum);
        for (int k = (kmin > 64 ? kmin : 64); k < i; ++k) sum = __builtin_fma(-tf_readlane(rb, k - 64), xs[k * TF_NB + c], sum);
        const double lii = i < 64 ? tf_readlane(ra, i) : tf_readlane(rb, i - 64);
        const double x = (i >= c) ? sum / lii : 0.0;
        xs[i * TF_NB + c] = x;
        out[(int64_t)i * TF_NB + c] = x;           // row-major inverse: out[i][c]
        ra = na; rb = nb;
    }
}

// ---- memory pipeline ---------------------------------------------------------------------------------------------------
// Both streams of a stage go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write pass) one
// stage ahead, into two slots each: the pack tile (16 KiB, shared by the four waves) and the stage's 16 older X columns
// (8 KiB; each wave fetches and reads only its own 16 rows).  A stage ends with  s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier :
//   vmcnt(0)   everything this wave issued during the stage has landed -- the next tile (in flight for a whole stage of
//              >= 2048 MFMA cycles, so the wait is normally free), the right-hand sides of the next block, the stores;
//   lgkmcnt(0) this wave's LDS reads of the current tile are complete before any wave, past the barrier, lets the DMA
//              overwrite that slot (the barrier is the raw s_barrier: a __syncthreads() would add nothing but its own waits).
// Measured and NOT kept (r03): counted waits -- vmcnt(N) leaving the N youngest operations in flight across the barrier, two
// tiles ahead in three slots.  They were no faster (10.3 vs 10.1 ms on 2^19 rows: the latencies already fit inside a stage)
// and they were WRONG under load: the scheme needs loads to retire in issue order, and LDS-DMA loads, register loads and
// stores in one queue do not (LLVM's waitcnt pass also treats such a mix as unordered); 1e-4 errors in a few rows of a
// 2^20-row solve, two runs in three, nothing at 2^13 rows.  tests/test_gpu_trsm.py now compares every row at 2^18.
__device__ __forceinline__ void tf_stage_end() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// The MFMAs of a stage go accumulator group by accumulator group (TF_GRP tiles, all four k-steps of each), not k-step by k-step
// over all eight accumulators, which is the order hipcc's scheduler restores unless scheduling barriers pin the groups.  Same
// box, 2^19 rows, M = 1024: 10.25 ms for groups of 8 (the compiler's order), 9.70-9.91 for 4, 9.75-9.88 for 1 -- although each
// group now waits for its own fragment reads with nothing to cover them; combining the grouping with fragment reads issued
// two groups ahead (counted lgkmcnt waits) measured the same 9.71-9.79.  tools/ubench/mfma_chain.hip shows no such effect for
// MFMAs alone (71.3 TFLOP/s at two waves per SIMD for any rotation of <= 12 accumulators), so it is the interleaving with
// the LDS reads, not the MFMA pipe itself.
constexpr int TF_GRP = 4;
constexpr int TF_XT = TF_ROWS * TF_KC;       // doubles per X stage tile

template <bool REV>
__global__ void __launch_bounds__(64 * TF_NW, TF_NW <= 4 ? 2 : 1)
trsm_fused_kernel(const double* __restrict__ pack, const double* Bin, double* Xout, double* scratch, int64_t nrhs, int64_t ldin,
                  int64_t ldout, int nb) {
    // REV: logical column c of the panels is memory column 128 nb - 1 - c (see trsm_pack_kernel): column offsets are subtracted
    constexpr int64_t SG = REV ? -1 : 1;
    const int64_t ctop = REV ? (int64_t)nb * TF_NB - 1 : 0;
    __shared__ __attribute__((aligned(16))) double Ls[2 * (TF_TILE + TF_XT)];     // 48 KiB
    double* const Xs = Ls + 2 * TF_TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fk = lane >> 4;
    const int64_t row = (int64_t)blockIdx.x * TF_ROWS + 16 * wave + fi;
    // rows past the end are redirected to a scratch row instead of being masked: every lane issues every operation
    // B tile / X store: register v <-> column 16 ct + 4 v + fk
    const double* bin = (row < nrhs ? Bin + row * ldin : scratch) + ctop + SG * fk;
    double* xst = (row < nrhs ? Xout + row * ldout : scratch) + ctop + SG * fk;
    // X fetch: lane l of piece i brings 16 bytes of row 8 i + l / 8; the 16-byte chunk it brings is the one that belongs
    // at its (lane-linear) LDS position under the XOR swizzle  chunk' = chunk ^ ((row >> 1) & 7)
    const double* xsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 8 * i + (lane >> 3);
        const int64_t grow = (int64_t)blockIdx.x * TF_ROWS + 16 * wave + r;
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        // (REV: the 16 bytes at memory columns [m, m + 1], m = top - 1 - 2 chunk, are the logical columns 2 chunk + 1, 2 chunk)
        xsrc[i] = (grow < nrhs ? Xout + grow * ldout : scratch) + (REV ? ctop - 1 - 2 * chunk : 2 * chunk);
    }
    int off[4], xoff[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        off[v] = fi * 16 + ((4 * v + fk) ^ tf_swz(fi));
        const int k = 4 * v + fk;
        xoff[v] = wave * 256 + fi * 16 + 2 * ((k >> 1) ^ ((fi >> 1) & 7)) + ((k & 1) ^ (REV ? 1 : 0));
    }
    const int64_t nst = tf_stages_before(nb);
    int64_t s = 0;                                  // stage; its tile sits in slot s & 1
    auto glds_pack = [&](int64_t t) {              // tile t (clamped) -> slot t & 1; four 1 KiB pieces per wave
        const int64_t tc = t < nst ? t : nst - 1;
        constexpr int PER_WAVE = TF_TILE / TF_NW;          // doubles of the tile each wave brings (1 KiB pieces)
        const double* src = pack + tc * TF_TILE + wave * PER_WAVE + lane * 2;
        double* dst = Ls + (t & 1) * TF_TILE + wave * PER_WAVE;
#pragma unroll
        for (int q = 0; q < PER_WAVE / 128; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 128),
                                             (__attribute__((address_space(3))) void*)(dst + q * 128), 16, 0, 0);
    };
    auto glds_x = [&](int64_t k0, int64_t t) {      // this wave's 16 rows x 16 columns from k0 -> its part of X slot t & 1
        double* dst = Xs + (t & 1) * TF_XT + wave * 256;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[i] + SG * k0),
                                             (__attribute__((address_space(3))) void*)(dst + i * 128), 16, 0, 0);
    };

    double4_t bnext[8], xprev[8];
    glds_pack(0);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        xprev[h] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int v = 0; v < 4; ++v) bnext[h][v] = bin[SG * (16 * h + 4 * v)];      // right-hand sides of block 0
    }
    tf_stage_end();

    for (int j = 0; j < nb; ++j) {
        const int64_t j0 = (int64_t)j * TF_NB;
        // acc starts as B_j (loaded under the previous block's diagonal product); the pack holds -L, so the update stages add
        double4_t acc[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) acc[h] = bnext[h];
        if (j > 0) {
            const int64_t kold = j0 - TF_NB;               // columns [0, kold) are re-read from the output
            // the block finished last, straight from its registers -- and on its way to memory: tile h is stored under stage h
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                glds_pack(s + 1);
#pragma unroll
                for (int v = 0; v < 4; ++v) xst[SG * (kold + 16 * h + 4 * v)] = xprev[h][v];
                if (h == 7) glds_x(0, s + 1);              // X of the first older-column stage (a tile nobody reads when kold = 0)
                const double* Lt = Ls + (s & 1) * TF_TILE;
#pragma unroll
                for (int hp = 0; hp < 8; hp += TF_GRP) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int h2 = hp; h2 < hp + TF_GRP; ++h2)
                            acc[h2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[h2 * 256 + off[v]], xprev[h][v], acc[h2], 0, 0, 0);
                }
                tf_stage_end();
                ++s;
            }
            for (int64_t k0 = 0; k0 < kold; k0 += TF_KC) {
                glds_pack(s + 1);
                glds_x(k0 + TF_KC < kold ? k0 + TF_KC : k0, s + 1);      // (the last one fetches a tile nobody reads)
                const double* Lt = Ls + (s & 1) * TF_TILE;
                const double* Xt = Xs + (s & 1) * TF_XT;
                double xv[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) xv[v] = Xt[xoff[v]];
#pragma unroll
                for (int hp = 0; hp < 8; hp += TF_GRP) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int h2 = hp; h2 < hp + TF_GRP; ++h2)
                            acc[h2] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[h2 * 256 + off[v]], xv[v], acc[h2], 0, 0, 0);
                }
                tf_stage_end();
                ++s;
            }
        }
        // acc = residual T = B_j - X_{<j} L_{j,<j}^T;  X_j = T inv(L_jj)^T  (the inverse is lower triangular: stage h feeds the
        // tiles >= h).  The right-hand sides of the next block are fetched meanwhile (two tiles per stage under the first four
        // stages; the last block re-reads its own columns and drops them)
        double4_t out[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) out[h] = (double4_t){0.0, 0.0, 0.0, 0.0};
        const int64_t jn = (j + 1 < nb) ? j0 + TF_NB : j0;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            glds_pack(s + 1);
            if (h < 4) {
#pragma unroll
                for (int t = 2 * h; t < 2 * h + 2; ++t)
#pragma unroll
                    for (int v = 0; v < 4; ++v) bnext[t][v] = bin[SG * (jn + 16 * t + 4 * v)];
            }
            const double* Lt = Ls + (s & 1) * TF_TILE;
#pragma unroll
            for (int cp = h; cp < 8; cp += TF_GRP) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int ct = cp; ct < cp + TF_GRP && ct < 8; ++ct)
                        out[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[ct * 256 + off[v]], acc[h][v], out[ct], 0, 0, 0);
            }
            tf_stage_end();
            ++s;
        }
#pragma unroll
        for (int h = 0; h < 8; ++h) xprev[h] = out[h];
    }
    const int64_t jl = (int64_t)(nb - 1) * TF_NB;
#pragma unroll
    for (int h = 0; h < 8; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) xst[SG * (jl + 16 * h + 4 * v)] = xprev[h][v];
}

// dLinv != nullptr: a full row-major inverse of L (leading dimension ldinv; its diagonal blocks ARE the inverses of L's
// diagonal blocks); else dInvBlocks: compact [nb][128][128] inverses of the diagonal blocks; neither: computed here.
// transposed: rows x with L^T x = b instead of L x = b (n must then be a multiple of 128).
int trsm_rows_fused(oak_ctx* ctx, const double* dL, int64_t n, int64_t ldl, const double* dLinv, int64_t ldinv,
                    const double* dInvBlocks, const double* dBin, int64_t ldin, double* dXout, int64_t ldout, int64_t nrhs, bool transposed) {
    if (n <= 0 || nrhs <= 0) return OAK_OK;
    const int64_t npad = ((n + TF_NB - 1) / TF_NB) * TF_NB;
    OAK_REQUIRE(!transposed || npad == n, "trsm_rows_fused: the transposed solve needs n = %lld to be a multiple of 128", (long long)n);
    OAK_REQUIRE(ldin >= npad && ldout >= npad && (ldin % 2) == 0 && (ldout % 2) == 0 &&
                    (((uintptr_t)dBin | (uintptr_t)dXout) & 15) == 0,
                "trsm_rows_fused: the panels need %lld (padded) columns, even row strides and 16-byte alignment", (long long)npad);
    const int nb = (int)(npad / TF_NB);
    if (dLinv == nullptr && dInvBlocks == nullptr) {
        double* dInv = nullptr;
        OAK_CHECK(get_buf_t(ctx, "trsm_inv", (size_t)nb * TF_NB * TF_NB, &dInv));
        const size_t lds = sizeof(double) * TF_NB * TF_NB;
        OAK_CHECK(ensure_max_dynamic_lds((const void*)trsm_block_inverse_kernel));
        trsm_block_inverse_kernel<<<(unsigned)nb, 128, lds, ctx->stream>>>(dL, n, ldl, dInv);
        OAK_HIP_CHECK(hipGetLastError());
        dInvBlocks = dInv;
    }
    double* dPack = nullptr;
    OAK_CHECK(get_buf_t(ctx, "trsm_pack", (size_t)tf_stages_before(nb) * TF_TILE, &dPack));
    const double* inv = dLinv ? dLinv : dInvBlocks;
    const int64_t inv_bs = dLinv ? TF_NB * ldinv + TF_NB : (int64_t)TF_NB * TF_NB;
    const int64_t inv_ld = dLinv ? ldinv : TF_NB;
    if (transposed) trsm_pack_kernel<true><<<dim3((unsigned)(8 * nb), (unsigned)nb), 256, 0, ctx->stream>>>(dL, n, ldl, inv, inv_bs, inv_ld, dPack);
    else trsm_pack_kernel<false><<<dim3((unsigned)(8 * nb), (unsigned)nb), 256, 0, ctx->stream>>>(dL, n, ldl, inv, inv_bs, inv_ld, dPack);
    OAK_HIP_CHECK(hipGetLastError());
    const unsigned grid = (unsigned)((nrhs + TF_ROWS - 1) / TF_ROWS);
    double* dScratch = nullptr;                       // where the lanes of rows past the end read and write
    OAK_CHECK(get_buf_t(ctx, "trsm_scratch_row", (size_t)npad + 64, &dScratch));
    OAK_CHECK(fill_zero(ctx, dScratch, sizeof(double) * ((size_t)npad + 64)));
    if (transposed) trsm_fused_kernel<true><<<grid, 64 * TF_NW, 0, ctx->stream>>>(dPack, dBin, dXout, dScratch, nrhs, ldin, ldout, nb);
    else trsm_fused_kernel<false><<<grid, 64 * TF_NW, 0, ctx->stream>>>(dPack, dBin, dXout, dScratch, nrhs, ldin, ldout, nb);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

}  // namespace oak

// Phi = Kuf Kuf^T (oak/utils.py:189-190; GPflow's AAT before whitening) on the INT8 matrix pipe, exactly: the "Ozaki scheme II" /
// Chinese-remainder construction.  oak_sgpr_set_precision: the automatic default on phi-route problems with M >= 640 inducing points and N >= 32768 rows (M >= 512: N M^2 >= 2^38), forced
// by mode 2, off in mode 0; phi route only.
//
//   1. every column m of the Kfu panel gets a power-of-two scale 2^s_m from a bound that needs no pass over the panel:
//          |K(x_n, z_m)| <= sqrt(K(x_n, x_n) K(z_m, z_m)) <= sqrt(max_n K_diag(x_n) * K_diag(z_m))       (K is positive semi-definite)
//      -- max_n K_diag(x_n) falls out of the featurize pass's kappa reduction, K_diag(z_m) is one small launch; both are capped by
//      Kmax = sum_r |w_r| e_r(sup k_1, .., sup k_D) from the kernel description alone, which also serves a description that is not
//      positive semi-definite (negative order variance);
//      A[n, m] = rint(K[n, m] 2^s_m) is an integer of at most B bits, B = 48 .. 50 (what the moduli of step 2 leave room for);
//   2. L pairwise coprime moduli p_i <= 254 with prod p_i > 2 N 2^(2B-2); residue planes R_i = A mod p_i (int8, |r| <= 127),
//      laid out [plane][n / 16][m][n % 16] so that an MFMA operand fragment is one 16-byte unit;
//   3. C_i = R_i^T R_i by v_mfma_i32_32x32x32_i8, int32 accumulation over row splits short enough to be exact, summed (int64) and
//      reduced mod p_i;
//   4. Garner / mixed-radix reconstruction of the exact integer X = A^T A from (C_1 .. C_L), Phi[a, b] = X[a, b] 2^(-s_a - s_b).
// The only error is the rounding of step 1 (|delta| <= 1/2 in the last of B bits, independent from entry to entry): the sum over
// N rows is exact, where the fp64 MFMA SYRK rounds N times.  Measured against 80-bit accumulation (tests/test_gpu_crt.py): 2e-16 .. 9e-16
// of sqrt(Phi_aa Phi_bb), the fp64 accumulation 3e-16 .. 4e-16 on the same problems; r05 probe with true column maxima
// (tools/ubench/ozaki2_syrk.hip): 3.8e-16 against 1.7e-15.
#include "oak_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace oak {

typedef int crt_v4i __attribute__((ext_vector_type(4)));
typedef int crt_v16i __attribute__((ext_vector_type(16)));

// pairwise coprime, largest first; <= 254: every residue -- also one formed from a quotient that is off by one next to a tie -- fits a
// signed byte with |r| <= 127, so that 32768 rows of products stay below 2^31
static const int kCrtModuli[CRT_MAXL] = {254, 253, 251, 249, 247, 245, 241, 239, 233, 229, 227, 223, 211, 199, 197, 193, 191, 181, 179, 173};

static int modinv(int a, int p) {
    a %= p; if (a < 0) a += p;
    for (int x = 1; x < p; ++x) if ((a * x) % p == 1) return x;
    return 0;
}

// ---- 1. column scales -------------------------------------------------------------------------------------------------------------
// kmax_x[0] <- max of x[0 .. n) (one workgroup; n = the featurize pass's per-workgroup maxima of K_diag, or all K_diag values)
__global__ void __launch_bounds__(256) crt_max_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ out) {
    __shared__ double sh[256];
    double m = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) m = fmax(m, x[i]);
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// sexp[m] = B - 1 - e with bound_m < 2^e; columns beyond M (zero padding) get the scale of column 0 (their entries are exact zeros).
// psd = 0 (a negative order variance in the description: K need not be positive semi-definite): the entry-wise bound
// |K| <= sum_r |w_r| e_r(kmax_1 .. kmax_D) = kmax for every column instead of the Cauchy-Schwarz one
// psd = 1: |K(x, z_m)| <= sqrt(max_n K_diag(x_n) K_diag(z_m)) with the largest K_diag of THIS rank's rows from the device (kmax_x; never
// above the a-priori kmax, which caps it against a stray value)
__global__ void __launch_bounds__(256) crt_scales_kernel(const double* __restrict__ kdiagZ, int64_t M, int64_t Mp2, double kmax, const double* __restrict__ kmax_x,
                                                         int B, int psd, int* __restrict__ sexp) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= Mp2) return;
    const double kd = kdiagZ[m < M ? m : 0];
    double kx = kmax_x != nullptr ? kmax_x[0] : kmax;
    kx = (kx > 0.0 && kx < kmax) ? kx : kmax;
    const double bound = (psd ? sqrt(kx * (kd > 0.0 ? kd : 0.0)) : kmax) * (1.0 + 0x1p-20) + 0x1p-300;
    int e = 0;
    frexp(bound, &e);                                  // bound < 2^e
    sexp[m] = B - 1 - e;
}

// ---- 2. conversion (stand-alone form): fp64 panel -> L residue planes ----------------------------------------------------------------
// thread = (16-row group, column): reads 16 doubles of its column (coalesced across the columns of a wave), writes 16 bytes per plane.
// a mod p through fp64: q = rint(a / p), r = a - q p (exact: |a| < 2^52), folded into [-p/2, p/2).  Rows >= na and columns >= ncols
// of the panel are zeros.
__global__ void __launch_bounds__(256) crt_convert_kernel(const double* __restrict__ K, int64_t ldk, int64_t na, int64_t ncols, int64_t Mp2,
                                                          const int* __restrict__ sexp, CrtMod md, int8_t* __restrict__ planes, int64_t rows_pad) {
    const int64_t m = (int64_t)blockIdx.y * 256 + threadIdx.x;
    const int64_t g = blockIdx.x;
    const double sc = ldexp(1.0, sexp[m]);
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t n = g * 16 + j;
        a[j] = (n < na && m < ncols) ? rint(K[n * ldk + m] * sc) : 0.0;
    }
    const int64_t plane_bytes = rows_pad * Mp2;
    for (int i = 0; i < md.L; ++i) {
        const double p = (double)md.p[i], ip = md.inv[i], hp = 0.5 * p;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double r = __builtin_fma(-rint(a[j] * ip), p, a[j]);
            r = r >= hp ? r - p : (r < -hp ? r + p : r);
            w[j >> 2] |= ((uint32_t)(int)r & 0xffu) << (8 * (j & 3));
        }
        *reinterpret_cast<uint4*>(planes + i * plane_bytes + (g * Mp2 + m) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---- 3. int8 SYRK of all planes --------------------------------------------------------------------------------------------------------
// workgroup = one 256 x 256 tile (bi <= bj) of one row split of one plane, eight waves (2 x 4), 128 x 64 per wave = 4 x 2
// v_mfma_i32_32x32x32_i8 tiles (128 accumulator registers, two waves per SIMD: one wave's LDS / memory waits are covered by the
// other's MFMAs).  Operands staged through LDS in 128-row stages (32 KiB per side, double-buffered): a plane byte enters a CU once per
// workgroup.  Lane (h = l >> 5, c = l & 31) takes rows 16 h .. 16 h + 15 of a 32-row k-step of column c -- the same k order for both
// operands, so the contraction is right whatever order the instruction gives the bytes.  XCD-aware decode (workgroup b runs on XCD
// b % 8): all tile pairs of a row split on ONE XCD, so that the split's rows stream through that XCD's L2 once.
constexpr int CT2 = 256, CST = 128;
__global__ void __launch_bounds__(512, 1) crt_syrk_i8_kernel(const int8_t* __restrict__ planes, int64_t rows_pad, int Mp2, int nt2, int64_t rows_per_split,
                                                             int nsplit, int* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) crt_v4i crt_lds[];      // [2 buffers][2 sides][CST / 16 groups][CT2 cols]
    const int ntile = nt2 * (nt2 + 1) / 2;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int split = xcd + 8 * (jx / ntile);
    int bi = 0, rem = jx % ntile;
    while (rem >= nt2 - bi) { rem -= nt2 - bi; ++bi; }
    const int bj = bi + rem;
    const int8_t* plane = planes + (int64_t)blockIdx.y * rows_pad * Mp2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    const int h = lane >> 5, c = lane & 31;
    const int64_t g0 = (int64_t)split * rows_per_split / 16, g1 = g0 + rows_per_split / 16;
    const crt_v4i* P = reinterpret_cast<const crt_v4i*>(plane);
    constexpr int SG = CST / 16, SIDE = SG * CT2, HQ = SG / 2;
    // staging item (q, side): thread t moves column t & 255 of group 2 q + (t >> 8)
    const int tcol = tid & 255, thalf = tid >> 8;
    crt_v4i ra[HQ], rb[HQ];
    crt_v16i acc[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0;
    auto gaddr = [&](int64_t g, int q) { const int64_t gq = g + 2 * q + thalf; return (gq < g1 ? gq : g1 - 1) * Mp2; };
#pragma unroll
    for (int q = 0; q < HQ; ++q) { ra[q] = P[gaddr(g0, q) + bi * CT2 + tcol]; rb[q] = P[gaddr(g0, q) + bj * CT2 + tcol]; }
#pragma unroll
    for (int q = 0; q < HQ; ++q) { crt_lds[(0 * 2 + 0) * SIDE + (2 * q + thalf) * CT2 + tcol] = ra[q]; crt_lds[(0 * 2 + 1) * SIDE + (2 * q + thalf) * CT2 + tcol] = rb[q]; }
#pragma unroll
    for (int q = 0; q < HQ; ++q) { ra[q] = P[gaddr(g0 + SG, q) + bi * CT2 + tcol]; rb[q] = P[gaddr(g0 + SG, q) + bj * CT2 + tcol]; }
    __syncthreads();
    int buf = 0;
    for (int64_t g = g0; g < g1; g += SG) {
        const crt_v4i* A = crt_lds + (buf * 2 + 0) * SIDE + wr * 128 + c;
        const crt_v4i* Bf = crt_lds + (buf * 2 + 1) * SIDE + wc * 64 + c;
        crt_v4i fa[2][4], fb[2][2];
#pragma unroll
        for (int x = 0; x < 4; ++x) fa[0][x] = A[h * CT2 + 32 * x];
#pragma unroll
        for (int y = 0; y < 2; ++y) fb[0][y] = Bf[h * CT2 + 32 * y];
#pragma unroll
        for (int kk = 0; kk < SG / 2; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < SG / 2) {
#pragma unroll
                for (int x = 0; x < 4; ++x) fa[nxt][x] = A[(2 * (kk + 1) + h) * CT2 + 32 * x];
#pragma unroll
                for (int y = 0; y < 2; ++y) fb[nxt][y] = Bf[(2 * (kk + 1) + h) * CT2 + 32 * y];
            }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[cur][x], fb[cur][y], acc[x][y], 0, 0, 0);
            {   // this k-step's share of the staging (one group pair per k-step), in program order behind its MFMAs
                const int q = kk;
                crt_lds[((buf ^ 1) * 2 + 0) * SIDE + (2 * q + thalf) * CT2 + tcol] = ra[q];
                crt_lds[((buf ^ 1) * 2 + 1) * SIDE + (2 * q + thalf) * CT2 + tcol] = rb[q];
                ra[q] = P[gaddr(g + 2 * SG, q) + bi * CT2 + tcol];
                rb[q] = P[gaddr(g + 2 * SG, q) + bj * CT2 + tcol];
            }
        }
        __syncthreads();
        buf ^= 1;
    }
    int* dst = part + ((int64_t)blockIdx.y * nsplit + split) * Mp2 * Mp2;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bi * CT2 + wr * 128 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = bj * CT2 + wc * 64 + 32 * y + c;
                dst[(int64_t)row * Mp2 + col] = acc[x][y][r];
            }
}

// ---- 3 (default). LDS-DMA operands, deep pipeline, diagonal-aware --------------------------------------------------------------------
// Same tile and wave layout; the operands go global -> LDS by the DMA path (global_load_lds_dwordx4: no staging registers, no ds_write
// pass) in 64-row stages (32 KiB) through NBUF = 4 slots, requested three stages ahead; a stage ends with a COUNTED wait -- the loop's
// only memory operations are LDS-DMA loads, which retire in order among themselves (the unordered mix of trsm_fused.hip had register
// loads and stores in the same queue) -- vmcnt(8) = everything but the two youngest stages has landed -- and s_barrier.  Wave w
// brings row group w & 3 of side w >> 2.
// What was measured on the way (r06, profiles/r06_crt_syrk_{probes,clocks,loader_waves}.txt, 15 planes at N = 2^20, M = 1024): two
// 128-row slots with vmcnt(0) 9.6 ms, four or five 64-row slots 9.3-9.5, four extra LOADER waves (twelve per workgroup, the compute
// waves issue no memory instruction) 9.2 -- with 10 % fewer active cycles at a 10 % lower clock: the kernel runs against the POWER
// limit (1.55-1.75 GHz; the MFMAs alone take 9.8e6 of 1.27e7 cycles), so only work NOT done is time.  Hence the diagonal tiles
// (bi == bj): one side is fetched (both operands are the same columns) and only the 36 of 64 MFMA tiles per k-step that touch the
// upper triangle are issued -- by straight-line code per wave role (MASK, bit 2 x + y), not by predicates in one loop (those cost
// the schedule of EVERY tile: 9.8 ms).  The price: diagonal tiles now run ahead of the other tiles of their row split, the split's rows
// no longer stream through the L2 once (fabric fetch 19-27 GB -> 42-47 GB per launch, L2 hit rate 0.70 -> 0.32) -- and the kernel is
// still 0.3-0.5 ms faster (9.1-9.3 vs 9.5-9.7 ms with OAK_CRT_DIAG=0).  Spreading a diagonal tile's MFMA tiles evenly over the four SIMDs
// ({8+0, 8+0, 7+3, 7+3} instead of 3 / 7 / 11 / 15) changed nothing (its stage time is not the matrix pipe's), and neither did dispatching
// all off-diagonal tiles of an XCD ahead of its diagonal ones (fetch 37 GB, same 9.2-9.3 ms); PACING the diagonal tiles on the progress of
// an off-diagonal tile of their split (a counter per tile in L2, polled every 16 stages by one lane) cost 0.3 ms and left the fetch at
// 42 GB.  Integer results: any mistake in the hand-written waits shows as a Phi that differs from the
// register-staged kernel's bit for bit (tests/test_gpu_crt.py).
template <unsigned MASK> struct CrtMask { static constexpr unsigned value = MASK; };
__global__ void __launch_bounds__(512, 1) crt_syrk_i8_deep_kernel(const int8_t* __restrict__ planes, int64_t rows_pad, int Mp2, int nt2, int64_t rows_per_split,
                                                                  int nsplit, int* __restrict__ part, int diag_mode) {
    extern __shared__ __attribute__((aligned(16))) crt_v4i crt_lds[];      // [NBUF][2 sides][4 groups][CT2 cols]
    constexpr int NBUF = 4, SGR = 4, SIDE = SGR * CT2, SLOT = 2 * SIDE, D = NBUF - 1;
    const int ntile = nt2 * (nt2 + 1) / 2;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int split = xcd + 8 * (jx / ntile);
    int bi = 0, rem = jx % ntile;
    while (rem >= nt2 - bi) { rem -= nt2 - bi; ++bi; }
    const int bj = bi + rem;
    const int8_t* plane = planes + (int64_t)blockIdx.y * rows_pad * Mp2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool diag = bi == bj && diag_mode != 0;
    const int wr = wave >> 2, wc = wave & 3;
    const int h = lane >> 5, c = lane & 31;
    const int64_t g0 = (int64_t)split * rows_per_split / 16, g1 = g0 + rows_per_split / 16;
    const crt_v4i* P = reinterpret_cast<const crt_v4i*>(plane);
    crt_v16i acc[4][2];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0;
    const int fq = wave & 3, fside = wave >> 2;
    const bool filler = !(diag && fside);             // a diagonal tile has one side
    const int64_t fcol = (fside ? bj : bi) * CT2 + lane;
    auto fill = [&](int64_t g, int slot) {            // the stage starting at row group g (clamped: copies past the end are never read)
        const int64_t gq = (g + fq < g1) ? g + fq : g1 - 1;
        const crt_v4i* src = P + gq * Mp2 + fcol;
        crt_v4i* dst = crt_lds + slot * SLOT + fside * SIDE + fq * CT2;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * j), (__attribute__((address_space(3))) void*)(dst + 64 * j), 16, 0, 0);
    };
    if (filler) {
#pragma unroll
        for (int d = 0; d < D; ++d) fill(g0 + SGR * d, d);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int a_off = wr * 128 + c, b_off = (diag ? 0 : SIDE) + wc * 64 + c;
    auto run = [&](auto mask_c) {
        constexpr unsigned MASK = decltype(mask_c)::value;
        int slot = 0, fslot = D;
        for (int64_t g = g0; g < g1; g += SGR) {
            if (filler) fill(g + SGR * D, fslot);      // into the slot whose readers all passed the barrier that ended the previous stage
            const crt_v4i* A = crt_lds + slot * SLOT + a_off;
            const crt_v4i* Bf = crt_lds + slot * SLOT + b_off;
            crt_v4i fa[2][4], fb[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    if ((MASK >> (2 * x)) & 3u) fa[kk][x] = A[(2 * kk + h) * CT2 + 32 * x];
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    if (MASK & (0x55u << y)) fb[kk][y] = Bf[(2 * kk + h) * CT2 + 32 * y];
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 2; ++y)
                        if ((MASK >> (2 * x + y)) & 1u) acc[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[kk][x], fb[kk][y], acc[x][y], 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            slot = (slot + 1 == NBUF) ? 0 : slot + 1;
            fslot = (fslot + 1 == NBUF) ? 0 : fslot + 1;
        }
        int* dst = part + ((int64_t)blockIdx.y * nsplit + split) * Mp2 * Mp2;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
                if ((MASK >> (2 * x + y)) & 1u) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = bi * CT2 + wr * 128 + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int col = bj * CT2 + wc * 64 + 32 * y + c;
                        dst[(int64_t)row * Mp2 + col] = acc[x][y][r];
                    }
                }
    };
    // tile (x, y) of this wave covers 32-row block 4 wr + x and 32-column block 2 wc + y: on a diagonal tile it is needed when
    // 2 wc + y >= 4 wr + x, i.e. with dg = 2 wc - 4 wr: all eight for dg >= 4, seven (not (3, 0)) for dg = 2, three for dg = 0, none below
    const int dg = diag ? 2 * wc - 4 * wr : 4;
    if (dg >= 4) run(CrtMask<0xFFu>{});
    else if (dg == 2) run(CrtMask<0xBFu>{});
    else if (dg == 0) run(CrtMask<0x0Bu>{});
    else run(CrtMask<0u>{});
}

// ---- 3b + 4. split sums mod p_i, then (last chunk) Garner digits and Phi ---------------------------------------------------------------
// thread = four consecutive entries (a, b .. b + 3) of the upper triangle (16-byte loads of the int32 partials, eight splits in flight).
// res[i][a * Mp2 + b] carries the residues between the chunks of a panel that does not fit one pass.  The mixed-radix digits v_i give
// X = v_0 + v_1 p_0 + v_2 p_0 p_1 + ...: Horner in exact int64 inside groups of five digits (< 2^40), the two or three groups joined
// by fp64 FMAs (one rounding each).  All modular steps in floating point, exact: x - p rint(x / p) with |x| < 2^24 (fp32) or 2^53
// (fp64); a quotient off by one near a tie leaves a digit of magnitude <= p / 2 + 1, which the congruences and the head room of
// prod p_i (> 2 bits) do not mind.  LT = number of moduli (compile time: the digit arrays stay in registers).
struct CrtGarnerF {
    int L, ngroups;
    float p[CRT_MAXL], ip[CRT_MAXL];
    float inv[CRT_MAXL][CRT_MAXL];      // inv[j][i] = p_j^-1 mod p_i  (j < i), symmetric representative
    double pg[CRT_MAXL / 5];            // product of the moduli of digit group k (five digits per group: < 2^40, exact)
};
template <int LT>
__global__ void __launch_bounds__(256) crt_reduce_kernel(const int* __restrict__ part, int nsplit, int64_t M, int64_t Mp2, const CrtGarnerF gr,
                                                         int* __restrict__ res, int first, int last, const int* __restrict__ sexp,
                                                         double* __restrict__ phi, double* __restrict__ phi_lo) {
    const int64_t a = blockIdx.y;
    const int64_t b0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4 + (a & ~(int64_t)3);      // first column group that touches the diagonal
    if (b0 >= M) return;
    const int64_t e = a * Mp2 + b0, MM = Mp2 * Mp2;
    float v[LT][4];
#pragma unroll
    for (int i = 0; i < LT; ++i) {
        long long s[4] = {0, 0, 0, 0};
        if (!first) {
            const crt_v4i r0 = *reinterpret_cast<const crt_v4i*>(res + (int64_t)i * MM + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] = r0[k];
        }
        const int* pp = part + (int64_t)i * nsplit * MM + e;
        for (int sp = 0; sp < nsplit; sp += 8) {              // nsplit is a multiple of 8
            crt_v4i t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const crt_v4i*>(pp + (int64_t)(sp + u) * MM);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += t[u][k];
        }
        const double pd = (double)gr.p[i], ipd = 1.0 / pd;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double sd = (double)s[k];
            v[i][k] = (float)__builtin_fma(-__builtin_rint(sd * ipd), pd, sd);
        }
    }
    if (!last) {
#pragma unroll
        for (int i = 0; i < LT; ++i) {
            crt_v4i o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (int)v[i][k];
            *reinterpret_cast<crt_v4i*>(res + (int64_t)i * MM + e) = o;
        }
        return;
    }
#pragma unroll
    for (int i = 1; i < LT; ++i) {                            // Garner: v_i <- ((..((r_i - v_0) / p_0 - v_1) / p_1 ..) mod p_i
        const float p = gr.p[i], ip = gr.ip[i];
#pragma unroll
        for (int j = 0; j < i; ++j) {
            const float inv = gr.inv[j][i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float u = (v[i][k] - v[j][k]) * inv;
                v[i][k] = __builtin_fmaf(-__builtin_rintf(u * ip), p, u);
            }
        }
    }
    // The exact integer as a DOUBLE-DOUBLE (phi_lo != NULL: the whitening of an ill-conditioned Kuu needs Phi to more than one double,
    // ddgemm.hip): Horner over the digit groups with error-free products (FMA) and two-sums; the power-of-two scale is exact.
    const int sa = sexp[a];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t b = b0 + k;
        double xh = 0.0, xl = 0.0;
#pragma unroll
        for (int g = (LT + 4) / 5 - 1; g >= 0; --g) {
            const int lo = 5 * g, hi = (lo + 5 < LT) ? lo + 5 : LT;
            long long G = 0;
#pragma unroll
            for (int i = hi - 1; i >= lo; --i) G = G * (long long)gr.p[i] + (long long)v[i][k];
            const double Gd = (double)G, pgd = gr.pg[g];
            const double p = xh * pgd;
            double e = __builtin_fma(xh, pgd, -p) + xl * pgd;
            const double s = p + Gd, t = s - p;
            e += (p - (s - t)) + (Gd - t);
            xh = s + e;
            xl = e - (xh - s);
        }
        if (b >= a && b < M) {
            const int sh = -(sa + sexp[b]);
            const double oh = ldexp(xh, sh), ol = ldexp(xl, sh);
            phi[a * M + b] = oh;
            phi[b * M + a] = oh;
            if (phi_lo != nullptr) { phi_lo[a * M + b] = ol; phi_lo[b * M + a] = ol; }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------
static inline int64_t pad_to(int64_t v, int64_t q) { return ((v + q - 1) / q) * q; }

// Kmax = sum_r w_r e_r(kmax_1 .. kmax_D), kmax_d = sup_x k_d(x, x): the base variance of a continuous sub-kernel (k = bv - cn^2), the
// largest diagonal entry of a discrete one's table
static double crt_kmax(const PreparedKernel& pk) {
    const int D = pk.dd.D, R = (int)pk.w_full.size() - 1;
    std::vector<double> e((size_t)R + 1, 0.0);
    e[0] = 1.0;
    for (int d = 0; d < D; ++d) {
        double km = pk.dd.bv[d];
        if (pk.dd.type[d] != OAK_DIM_RBF) {
            const int C = pk.dd.ncat[d];
            km = 0.0;
            for (int c = 0; c < C; ++c) km = std::max(km, std::fabs(pk.tables[(size_t)pk.dd.tab_off[d] + (size_t)C * C + c]));
        }
        km = std::fabs(km);
        for (int q = R; q >= 1; --q) e[q] += km * e[q - 1];
    }
    double K = 0.0;
    for (int r = 0; r <= R; ++r) K += std::fabs(pk.w_full[r]) * e[r];
    return K;
}

bool crt_supported(const oak_ctx* ctx, int64_t M) {
    return pad_to(M, 256) <= 4096 && ctx->N >= 4096;      // (partials: L x nsplit x Mp2^2 x 4 bytes; 13-18 moduli cover 2^12 .. 2^45 rows)
}

// Plan of one panel chunk of `na` rows: moduli for n_total rows in all (residues are carried from chunk to chunk), row splits (a
// multiple of 8 -- one XCD each --, <= 32768 rows: int32 sums are exact to 2^31 / 128^2 = 131 072 rows, whole 128-row stages) and the
// buffers (grow-only, so a later, shorter chunk fits).
int crt_plan(oak_ctx* ctx, int64_t na, int64_t M, int64_t n_total, CrtPlan* pl) {
    // Integer width: the moduli are chosen for 48-bit integers (prod p_i > 2 N 2^(2B-2)); whatever head room their product then
    // leaves is spent on wider integers at no cost -- B = 49 at N = 2^20 (15 moduli, 117.9 bits >= 2 x 49 - 1 + 20.25), 50 up to
    // 2^18 rows (50 is the cap: the conversion's magic-constant arithmetic holds |x| < 2^51).
    pl->B = 48;
    pl->Mp2 = pad_to(M, 256);
    const double lgn = std::log2((double)n_total);
    const double need = 2.0 * pl->B - 1.0 + lgn + 0.25;
    pl->md.L = 0;
    double bits = 0.0;
    while (bits <= need && pl->md.L < CRT_MAXL) {
        pl->md.p[pl->md.L] = kCrtModuli[pl->md.L]; pl->md.inv[pl->md.L] = 1.0 / kCrtModuli[pl->md.L];
        bits += std::log2((double)kCrtModuli[pl->md.L]); ++pl->md.L;
    }
    OAK_REQUIRE(bits > need, "int8 CRT statistics: %lld rows need more than %d moduli", (long long)n_total, CRT_MAXL);
    while (pl->B < 50 && 2.0 * (pl->B + 1) - 1.0 + lgn + 0.25 < bits) ++pl->B;
    if (const char* e = getenv("OAK_CRT_BITS")) {      // experiment knob: a fixed width (the moduli follow)
        const int v = atoi(e);
        if (v >= 40 && v <= 50) {
            pl->B = v; pl->md.L = 0; bits = 0.0;
            while (bits <= 2.0 * v - 1.0 + lgn + 0.25 && pl->md.L < CRT_MAXL) {
                pl->md.p[pl->md.L] = kCrtModuli[pl->md.L]; pl->md.inv[pl->md.L] = 1.0 / kCrtModuli[pl->md.L];
                bits += std::log2((double)kCrtModuli[pl->md.L]); ++pl->md.L;
            }
        }
    }
    pl->nsplit = (int)pad_to((na + 32767) / 32768, 8);
    pl->rps = pad_to((na + pl->nsplit - 1) / pl->nsplit, CST);
    pl->rows_pad = pl->rps * pl->nsplit;
    OAK_CHECK(get_buf_t(ctx, "crt_sexp", (size_t)pl->Mp2, &pl->d_sexp));
    OAK_CHECK(get_buf_t(ctx, "crt_planes", (size_t)pl->md.L * pl->rows_pad * pl->Mp2, &pl->d_planes));
    OAK_CHECK(get_buf_t(ctx, "crt_part", (size_t)pl->md.L * pl->nsplit * pl->Mp2 * pl->Mp2, &pl->d_part));
    OAK_CHECK(get_buf_t(ctx, "crt_res", (size_t)pl->md.L * pl->Mp2 * pl->Mp2, &pl->d_res));
    return OAK_OK;
}

// column scales from K_diag(Z) and the a-priori bound on K_diag(x): once per evaluation, before the first chunk
int crt_scales(oak_ctx* ctx, const PreparedKernel& pk, const Feat& FX, const Feat& FZ, int64_t M, const CrtPlan& pl, bool kdiag_parts) {
    double *d_kdz = nullptr, *d_kmx = nullptr;
    OAK_CHECK(get_buf_t(ctx, "crt_kdiagZ", (size_t)M, &d_kdz));
    OAK_CHECK(get_buf_t(ctx, "crt_kmax_x", 1, &d_kmx));
    OAK_CHECK(gram_diag(ctx, pk, FZ, d_kdz, nullptr));
    if (kdiag_parts) {
        // the featurize pass of X left one maximum of K_diag per 256-row workgroup behind its per-workgroup sums ("feat_kpart")
        const int64_t nblk = (FX.ld + 255) / 256;
        crt_max_kernel<<<1, 256, 0, ctx->stream>>>((const double*)peek_buf(ctx, "feat_kpart") + nblk, nblk, d_kmx);
    } else {
        double* d_kd = nullptr;
        OAK_CHECK(get_buf_t(ctx, "kdiag", (size_t)FX.n, &d_kd));
        OAK_CHECK(gram_diag(ctx, pk, FX, d_kd, nullptr));
        crt_max_kernel<<<1, 256, 0, ctx->stream>>>(d_kd, FX.n, d_kmx);
    }
    OAK_HIP_CHECK(hipGetLastError());
    bool psd = true;
    for (double w : pk.w_full) psd = psd && w >= 0.0;
    for (int d = 0; d < pk.dd.D; ++d) psd = psd && pk.dd.bv[d] >= 0.0;
    crt_scales_kernel<<<(unsigned)((pl.Mp2 + 255) / 256), 256, 0, ctx->stream>>>(d_kdz, M, pl.Mp2, crt_kmax(pk), d_kmx, pl.B, psd ? 1 : 0, pl.d_sexp);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// e_m = exponent of the a-priori bound on |K(x, z_m)| (no data of this rank's rows in it: every rank of a communicator derives the same)
__global__ void __launch_bounds__(256) crt_eexp_kernel(const double* __restrict__ kdiagZ, int64_t M, double kmax, int psd, int* __restrict__ eexp) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const double kd = kdiagZ[m];
    const double bound = (psd ? sqrt(kmax * (kd > 0.0 ? kd : 0.0)) : kmax) * (1.0 + 0x1p-20) + 0x1p-300;
    int e = 0;
    frexp(bound, &e);
    eexp[m] = e;
}
int crt_bound_exponents(oak_ctx* ctx, const PreparedKernel& pk, const Feat& FZ, int64_t M, int* d_eexp) {
    double* d_kdz = nullptr;
    OAK_CHECK(get_buf_t(ctx, "crt_kdiagZ", (size_t)M, &d_kdz));
    OAK_CHECK(gram_diag(ctx, pk, FZ, d_kdz, nullptr));
    bool psd = true;
    for (double w : pk.w_full) psd = psd && w >= 0.0;
    for (int d = 0; d < pk.dd.D; ++d) psd = psd && pk.dd.bv[d] >= 0.0;
    crt_eexp_kernel<<<(unsigned)((M + 255) / 256), 256, 0, ctx->stream>>>(d_kdz, M, crt_kmax(pk), psd ? 1 : 0, d_eexp);
    OAK_HIP_CHECK(hipGetLastError());
    return OAK_OK;
}

// stand-alone conversion of an fp64 panel chunk (the fallback of the fused Gram epilogue, gram.hip::gram_crt_kernel)
int crt_convert_panel(oak_ctx* ctx, const CrtPlan& pl, const double* d_panel, int64_t ldp, int64_t na) {
    PhaseTimer t(ctx, "crt_convert");
    crt_convert_kernel<<<dim3((unsigned)(pl.rows_pad / 16), (unsigned)(pl.Mp2 / 256)), 256, 0, ctx->stream>>>(d_panel, ldp, na, std::min<int64_t>(ldp, pl.Mp2), pl.Mp2,
                                                                                                             pl.d_sexp, pl.md, pl.d_planes, pl.rows_pad);
    OAK_HIP_CHECK(hipGetLastError());
    t.stop();
    return OAK_OK;
}

// int8 SYRK of the chunk's planes, split sums joined to the residues carried so far; on the last chunk Phi (full, symmetric)
int crt_accumulate(oak_ctx* ctx, const CrtPlan& pl, int64_t M, bool first_chunk, bool last_chunk, double* d_phi, double* d_phi_lo) {
    CrtGarnerF gr;
    const CrtMod& md = pl.md;
    gr.L = md.L; gr.ngroups = (md.L + 4) / 5;
    for (int i = 0; i < md.L; ++i) {
        gr.p[i] = (float)md.p[i]; gr.ip[i] = 1.0f / (float)md.p[i];
        for (int j = 0; j < i; ++j) { int v = modinv(md.p[j], md.p[i]); if (2 * v >= md.p[i]) v -= md.p[i]; gr.inv[j][i] = (float)v; }
    }
    for (int k = 0; k < gr.ngroups; ++k) { double pgk = 1.0; for (int i = 5 * k; i < std::min(5 * k + 5, md.L); ++i) pgk *= md.p[i]; gr.pg[k] = pgk; }
    {
        PhaseTimer t(ctx, "crt_syrk");
        const int nt2 = (int)(pl.Mp2 / CT2), ntile2 = nt2 * (nt2 + 1) / 2;
        const size_t lds = sizeof(crt_v4i) * 2 * 2 * (CST / 16) * CT2;          // 128 KiB in both kernels (2 x 128 or 4 x 64 rows)
        // A/B knob (and the reference of tests/test_gpu_crt.py): OAK_CRT_SYRK=4 runs the register-staged kernel
        const char* ev = getenv("OAK_CRT_SYRK");
        const int variant = ev ? atoi(ev) : 64;
        const int threads = 512;
        const int diag_mode = getenv("OAK_CRT_DIAG") ? atoi(getenv("OAK_CRT_DIAG")) : 1;      // A/B knob: 0 = diagonal tiles run like any other tile
        const dim3 sgrid((unsigned)(ntile2 * pl.nsplit), (unsigned)md.L);
        if (variant == 4) {
            OAK_CHECK(ensure_max_dynamic_lds((const void*)crt_syrk_i8_kernel));
            crt_syrk_i8_kernel<<<sgrid, threads, lds, ctx->stream>>>(pl.d_planes, pl.rows_pad, (int)pl.Mp2, nt2, pl.rps, pl.nsplit, pl.d_part);
        } else {
            OAK_CHECK(ensure_max_dynamic_lds((const void*)crt_syrk_i8_deep_kernel));
            crt_syrk_i8_deep_kernel<<<sgrid, threads, lds, ctx->stream>>>(pl.d_planes, pl.rows_pad, (int)pl.Mp2, nt2, pl.rps, pl.nsplit, pl.d_part, diag_mode);
        }
        OAK_HIP_CHECK(hipGetLastError());
        t.stop();
    }
    {
        PhaseTimer t(ctx, "crt_reduce");
        const dim3 grid((unsigned)((M + 1023) / 1024), (unsigned)M);
#define OAK_CRT_RED(LL) case LL: crt_reduce_kernel<LL><<<grid, 256, 0, ctx->stream>>>(pl.d_part, pl.nsplit, M, pl.Mp2, gr, pl.d_res, first_chunk ? 1 : 0, \
                                                                                      last_chunk ? 1 : 0, pl.d_sexp, d_phi, d_phi_lo); break;
        switch (md.L) {
            OAK_CRT_RED(13) OAK_CRT_RED(14) OAK_CRT_RED(15) OAK_CRT_RED(16) OAK_CRT_RED(17) OAK_CRT_RED(18)
            default: set_error("int8 CRT statistics: %d moduli outside the instantiated range 13..18", md.L); return OAK_E_ARG;
        }
#undef OAK_CRT_RED
        OAK_HIP_CHECK(hipGetLastError());
        t.stop();
    }
    return OAK_OK;
}

}  // namespace oak

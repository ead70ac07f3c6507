// Adjoint panel of the gradient, G = Kfu H (grad.hip: dELBO/dKuf before the pair kernel contracts it with dK/dtheta; TF autodiff through
// SGPR.elbo in the reference, model_utils.py:168-173), on the INT8 matrix pipe from the residue planes the forward pass left behind.
//
// The fp64 GEMM is 2 M^2 N flops at 0.83 of the fp64 MFMA peak: 33.6 ms of a 73 ms forward+gradient step at the headline size, the largest
// single item.  The forward pass of the int8 route (crt.hip) has already written Kfu as integers K'[n, i] = rint(K[n, i] 2^a_i) of B <= 50
// bits in L residue planes, so the product only needs H in the same form:
//   H'[i, j] = rint(H[i, j] 2^-a_i c_j),  c_j from the 2-norm of column j of H 2^-a (Cauchy-Schwarz bound on the product, crt_hscale_kernel),
//   sum_i K'[n, i] H'[i, j] = c_j G[n, j]  up to the two roundings -- an integer below prod p / 2 for the first Lg moduli (13 at M = 1024:
//   46-49 bits for the entries of H'),
//   one v_mfma_i32_32x32x32_i8 GEMM per modulus (K = M: int32 sums of <= 1024 products of bytes are exact), reconstructed in the
//   epilogue WITHOUT storing residues: x / P = sum_p ((x v_p) mod p) / p  mod 1 with v_p = (P / p)^-1 mod p -- v_p is multiplied into the
//   residues of H' when they are made, so the accumulator of plane p IS x v_p mod p -- one fp64 fraction per output entry carried in
//   registers from plane to plane (each step adds a number below 1/2 in magnitude: one rounding of 2^-53 per plane).
// Accuracy: the fraction is good to ~2^-52, i.e. G to 2^-52 of the a-priori bound M max|K'| max|H'| (the integer must fit below P / 2
// whatever the signs); the fp64 GEMM is good to 2^-53 sqrt(M) of the largest product -- some 100x better, and G = Kfu H cancels like
// cond(Kuu).  Hence the conditioning rule in sgpr.hip (well-conditioned Kuu only).  It is NOT the exact arithmetic of the forward Phi.
//
// Layout problem and its answer: the planes are [plane][n / 16][m][n % 16] -- 16 consecutive ROWS n per 16-byte unit, what the SYRK's
// contraction over n wants -- but this product contracts over m.  gfx950's LDS transpose read does the turn for free: a 16-lane group
// that reads an [8 m][16 n] byte block with ds_read_b64_tr_b8 gets, in lane d, the eight m of column n = d (probed: tools/ubench/
// tr8_probe.hip).  Two of them are one MFMA operand (16 k values of one row n).  H' is converted straight into the other operand's
// natural layout [plane][k / 16][j][k % 16].
//
// Workgroup = 128 rows x 128 columns of G, four waves (2 x 2) of 64 x 64 (four 32 x 32 tiles: 64 accumulator + 128 fraction registers),
// TWO workgroups per CU: a wave folds its accumulators into the fractions (VALU, ~1400 cycles per plane) while the other workgroup's wave
// on the same SIMD keeps the matrix pipe busy.  Operands by LDS-DMA in 64-k stages (16 KiB) through four slots, counted waits as in
// crt_syrk_i8_deep_kernel.  Workgroups of one row tile (all column tiles) run on one XCD back to back: the planes are fetched once.
#include "oak_internal.h"
#include <cmath>
#include <cstdlib>

namespace oak {

typedef int cg_v2i __attribute__((ext_vector_type(2)));
typedef int cg_v4i __attribute__((ext_vector_type(4)));
typedef int cg_v16i __attribute__((ext_vector_type(16)));

struct CrtGemmC {
    int L;
    float p[CRT_MAXL], ip[CRT_MAXL];
    double ipd[CRT_MAXL];    // 1 / p
    int v[CRT_MAXL];         // (P / p)^-1 mod p, multiplied into the residues of H'
    double P;                // product of the L moduli
};

// Column scale of H' from the Cauchy-Schwarz bound  |sum_i K'[n, i] H'[i, j]| <= |K'[n, :]|_2 |H'[:, j]|_2 <= sqrt(M) 2^(B-1) |H'[:, j]|_2:
// cs[j] = target / |h_j|_2 with h_j[i] = H[i][j] 2^-sexp[i] and target = 0.99 (P / 2) / (sqrt(M) 2^(B-1)), capped so that the largest entry
// stays below 2^52.  The columns of H are peaked (|h_j|_2 is a few times max |h_j|, not sqrt(M) times): the integer fits below P / 2 with
// ~3 more bits for H' than the entry-wise bound M max|K'| max|H'| leaves -- and the reconstruction's rounding, which is relative to P,
// is that much smaller against G.  Not a power of two: G = x / cs[j] costs one more rounding.  (zero columns and the padding: 1)
__global__ void __launch_bounds__(256) crt_hscale_kernel(const double* __restrict__ H, int64_t M, int64_t Mp2, const int* __restrict__ sexp, double target,
                                                         double* __restrict__ cs) {
    __shared__ double shm[4][64], shn[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int64_t j = (int64_t)blockIdx.x * 64 + c;
    double m = 0.0, n2 = 0.0;
    if (j < M)
        for (int64_t i = rg; i < M; i += 4) { const double h = ldexp(fabs(H[i * M + j]), -sexp[i]); m = fmax(m, h); n2 += h * h; }
    shm[rg][c] = m; shn[rg][c] = n2;
    __syncthreads();
    if (rg == 0 && j < Mp2) {
        m = fmax(fmax(shm[0][c], shm[1][c]), fmax(shm[2][c], shm[3][c]));
        n2 = (shn[0][c] + shn[1][c]) + (shn[2][c] + shn[3][c]);
        double v = 1.0;
        if (m > 0.0 && n2 < 1e300) v = fmin(target / sqrt(n2), 0x1p52 * 0.99 / m);
        cs[j] = v;
    }
}

// residue planes of H' in the operand layout [plane][k / 16][j][k % 16] (k = row index i of H: the contraction index)
__global__ void __launch_bounds__(256) crt_hconvert_kernel(const double* __restrict__ H, int64_t M, int64_t Mp2, const int* __restrict__ sexp,
                                                           const double* __restrict__ cs, CrtMod md, CrtGemmC gc, int8_t* __restrict__ hplanes) {
    const int64_t j = (int64_t)blockIdx.y * 256 + threadIdx.x;
    const int64_t kb = blockIdx.x;
    const double cj = cs[j];
    double a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int64_t i = kb * 16 + t;
        a[t] = (i < M && j < M) ? rint(ldexp(H[i * M + j], -sexp[i]) * cj) : 0.0;
    }
    const int64_t plane_bytes = Mp2 * Mp2;
    for (int q = 0; q < gc.L; ++q) {
        const double p = (double)md.p[q], ip = md.inv[q], hp = 0.5 * p, v = (double)gc.v[q];
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            double r = __builtin_fma(-rint(a[t] * ip), p, a[t]);
            r *= v;                                                  // (H' mod p) (P / p)^-1 mod p: |r v| < 2^16
            r = __builtin_fma(-rint(r * ip), p, r);
            r = r >= hp ? r - p : (r < -hp ? r + p : r);
            w[t >> 2] |= ((uint32_t)(int)r & 0xffu) << (8 * (t & 3));
        }
        *reinterpret_cast<uint4*>(hplanes + q * plane_bytes + (kb * Mp2 + j) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

constexpr int GT = 128, GTJ = 256, GKS = 64;      // tile: rows n, columns j; k per stage
constexpr int G_NBUF = 6, G_D = G_NBUF - 1;
constexpr int G_AS = 72;                          // 16-byte units per row group of A in LDS: 64 k-rows + 8 of padding -- the two 16-lane groups of a
                                                  // half-wave read row groups nb and nb + 1: 1152 bytes apart puts them on different bank halves
constexpr int G_BO = 8 * G_AS;                    // B behind the eight row groups of A
constexpr int G_SLOT = G_BO + GTJ * GKS / 16;     // 16-byte units per slot: A [8 row groups][72][16 n], then B [4 k groups][256 j][16 k]

__device__ __forceinline__ void cg_glds(const cg_v4i* src, cg_v4i* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

__global__ void __launch_bounds__(512, 1) crt_gemm_i8_kernel(const int8_t* __restrict__ planes, int64_t rows_pad, const int8_t* __restrict__ hplanes, int Mp2,
                                                             int64_t na, int64_t M, const CrtGemmC gc, const double* __restrict__ cs,
                                                             double* __restrict__ G, int64_t ldg, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) cg_v4i cg_lds[];
    const int njt = Mp2 / GTJ;
    const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int jt = t % njt, nt = (t / njt) * 8 + xcd;
    if (nt >= ntiles) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int spp = Mp2 / GKS;                    // stages per plane
    // Operand requests: wave-uniform byte pointers bumped from stage to stage (scalar adds), the lane's 16 bytes as a 32-bit offset --
    // the loop's address arithmetic is a handful of SALU instructions (the first version recomputed 64-bit products per stage: 66 SALU
    // and 12 VALU instructions per wave and stage next to 8 MFMAs, and the waves' in-order issue, not a pipe, bounded the kernel).
    // A: row group `wave` of the tile, 64 k-rows = 1 KiB per stage, planes rows_pad * Mp2 bytes apart.
    // B: k group wave >> 1, column half wave & 1: the planes are contiguous, every stage is 64 * Mp2 bytes further.
    const char* pa_cur = reinterpret_cast<const char*>(planes) + (((int64_t)nt * (GT / 16) + wave) * Mp2) * 16;
    const char* pb_cur = reinterpret_cast<const char*>(hplanes) + ((int64_t)(wave >> 1) * Mp2 + (int64_t)jt * GTJ + (wave & 1) * 128) * 16;
    const int64_t a_wrap = rows_pad * (int64_t)Mp2 - (int64_t)Mp2 * 16;      // from the end of one plane's k range to the start of the next plane's
    const int64_t b_step = (int64_t)Mp2 * 64;
    const unsigned voff = (unsigned)lane * 16u;
    int fpl = 0, fks = 0;
    unsigned fdst = 0;                                                        // 16-byte units into the ring
    auto fill = [&]() {
        cg_v4i* d = cg_lds + fdst;
        cg_glds(reinterpret_cast<const cg_v4i*>(pa_cur + voff), d + wave * G_AS);
        cg_glds(reinterpret_cast<const cg_v4i*>(pb_cur + voff), d + G_BO + (wave >> 1) * GTJ + (wave & 1) * 128);
        cg_glds(reinterpret_cast<const cg_v4i*>(pb_cur + 1024 + voff), d + G_BO + (wave >> 1) * GTJ + (wave & 1) * 128 + 64);
        fdst = (fdst + G_SLOT == G_NBUF * G_SLOT) ? 0u : fdst + G_SLOT;
        pa_cur += 1024; pb_cur += b_step;
        if (++fks == spp) {
            fks = 0;
            if (++fpl < gc.L) pa_cur += a_wrap;
            else { fpl = gc.L - 1; pa_cur -= (int64_t)Mp2 * 16; pb_cur -= b_step * spp; }      // beyond the last stage: re-reads nobody consumes
        }
    };
#pragma unroll
    for (int d = 0; d < G_D; ++d) fill();
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // three requests per wave and stage: all but the four youngest stages
    __builtin_amdgcn_s_barrier();
    const int q = (lane >> 4) & 1, g = lane >> 5, i16 = lane & 15, c = lane & 31;
    const int a_lane = ((wr * 4 + q) * G_AS + g * 16 + (i16 >> 1)) * 16 + (i16 & 1) * 8;      // bytes
    const int b_lane = (G_BO + g * GTJ + wc * 64 + c) * 16;
    cg_v16i acc[2][2];
    double ys[2][2][16];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[x][y][r] = 0; ys[x][y][r] = 0.0; }
    const int S = gc.L * spp;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)cg_lds;      // LDS byte address of the ring
    int cpl = 0, cks = 0;
    unsigned rbase = lds0;                      // LDS byte address of the slot being read
    // LDS reads by inline asm: the compiler would put s_waitcnt vmcnt(0) in front of a transpose read it sees next to LDS-DMA traffic
    // (the stage being read landed before the barrier that precedes the read: the counted waits).  Their completion is waited for by
    // hand -- the "+v" operands tie the MFMAs behind the waits.  The loop is rotated by half a stage: the fragments of a stage's first
    // k-step are requested right behind the barrier that ends the previous stage, while that stage's last four MFMAs still run.
    cg_v2i al[2][2], ah[2][2];
    cg_v4i fb[2][2];
#define OAK_CG_READ(KK, PA, PB)                                                                                                              \
    _Pragma("unroll") for (int x = 0; x < 2; ++x) {                                                                                         \
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(al[KK][x]) : "v"(PA), "n"(x * 2 * G_AS * 16 + KK * 512));                  \
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(ah[KK][x]) : "v"(PA), "n"(x * 2 * G_AS * 16 + KK * 512 + 128));            \
    }                                                                                                                                        \
    _Pragma("unroll") for (int y = 0; y < 2; ++y)                                                                                           \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[KK][y]) : "v"(PB), "n"((KK * 2 * GTJ + y * 32) * 16));
    {
        const unsigned pa = lds0 + a_lane, pb = lds0 + b_lane;
        OAK_CG_READ(0, pa, pb)
    }
    for (int s = 0; s < S; ++s) {
        fill();
        {
            const unsigned pa = rbase + a_lane, pb = rbase + b_lane;
            OAK_CG_READ(1, pa, pb)
        }
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(al[0][0]), "+v"(ah[0][0]), "+v"(al[0][1]), "+v"(ah[0][1]), "+v"(fb[0][0]), "+v"(fb[0][1]));
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const cg_v4i fa = cg_v4i{al[0][x][0], al[0][x][1], ah[0][x][0], ah[0][x][1]};
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb[0][y], acc[x][y], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[1][0]), "+v"(ah[1][0]), "+v"(al[1][1]), "+v"(ah[1][1]), "+v"(fb[1][0]), "+v"(fb[1][1]));
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const cg_v4i fa = cg_v4i{al[1][x][0], al[1][x][1], ah[1][x][0], ah[1][x][1]};
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb[1][y], acc[x][y], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        rbase = (rbase + G_SLOT * 16 == lds0 + G_NBUF * G_SLOT * 16) ? lds0 : rbase + G_SLOT * 16;
        {   // first k-step of the next stage (beyond the last stage: a slot nobody filled anew -- read and dropped)
            const unsigned pa = rbase + a_lane, pb = rbase + b_lane;
            OAK_CG_READ(0, pa, pb)
        }
        ++cks;
        if ((cks & 15) == 0 || cks == spp) {
            // <= 1024 products of bytes per accumulator (< 2^24: exact in fp32): s = acc mod p = x v_p mod p, fraction += s / p (mod 1)
            const float pf = gc.p[cpl], ipf = gc.ip[cpl];
            const double w = gc.ipd[cpl];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float af = (float)acc[x][y][r];
                        const float rr = __builtin_fmaf(-__builtin_rintf(af * ipf), pf, af);
                        ys[x][y][r] = __builtin_amdgcn_fract(__builtin_fma((double)rr, w, ys[x][y][r]));
                        acc[x][y][r] = 0;
                    }
        }
        if (cks == spp) { cks = 0; ++cpl; }
    }
#undef OAK_CG_READ
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[0][0]), "+v"(ah[0][0]), "+v"(al[0][1]), "+v"(ah[0][1]), "+v"(fb[0][0]), "+v"(fb[0][1]));
    // x = P (fraction, centred), G = x / cs_j
    const int h = lane >> 5;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int64_t col = (int64_t)jt * GTJ + wc * 64 + y * 32 + c;
        const double sc = gc.P / cs[col];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = (int64_t)nt * GT + wr * 64 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const double f = ys[x][y][r];
                if (row < na && col < M) G[row * ldg + col] = (f >= 0.5 ? f - 1.0 : f) * sc;
            }
    }
}

static int cg_modinv(int a, int p) {
    a %= p; if (a < 0) a += p;
    for (int x = 1; x < p; ++x) if ((a * x) % p == 1) return x;
    return 0;
}

// G[0 .. na)[0 .. M) = Kfu H from the planes of `pl` (rows 0 .. na of the panel the forward pass converted) and the symmetric M x M
// matrix d_H.  Returns OAK_E_ARG without touching G when the moduli of the plan leave no room for a useful H' (the caller runs the fp64 GEMM).
int crt_gemm_adjoint(oak_ctx* ctx, const CrtPlan& pl, int64_t M, int64_t na, const double* d_H, double* d_G, int64_t ldg) {
    // moduli: the fewest whose product leaves a column with max |h| = |h|_2 / 4 (typical: the columns of H are peaked) `want` bits
    int want = 44;
    if (const char* e = getenv("OAK_CRT_GEMM_BITS")) { const int v = atoi(e); if (v >= 30 && v <= 52) want = v; }
    int Lg = 0, BH = 0;
    double bits = 0.0;
    while (Lg < pl.md.L) {
        bits += std::log2((double)pl.md.p[Lg]); ++Lg;
        BH = (int)std::floor(bits - 1.0 - 0.5 * std::log2((double)M) - (double)(pl.B - 1) - 2.0);
        if (BH >= want) break;
    }
    if (BH > 52) BH = 52;
    OAK_REQUIRE(BH >= 36, "int8 adjoint GEMM: the %d moduli of the forward pass leave %d bits for H", pl.md.L, BH);
    CrtGemmC gc;
    gc.L = Lg; gc.P = 1.0;
    for (int i = 0; i < Lg; ++i) gc.P *= (double)pl.md.p[i];
    for (int i = 0; i < Lg; ++i) {
        const int p = pl.md.p[i];
        int prod = 1;
        for (int k = 0; k < Lg; ++k) if (k != i) prod = (prod * (pl.md.p[k] % p)) % p;
        gc.p[i] = (float)p; gc.ip[i] = 1.0f / (float)p;
        gc.v[i] = cg_modinv(prod, p); gc.ipd[i] = 1.0 / (double)p;
    }
    double* d_cs = nullptr; int8_t* d_hpl = nullptr;
    OAK_CHECK(get_buf_t(ctx, "crt_hscale", (size_t)pl.Mp2, &d_cs));
    OAK_CHECK(get_buf_t(ctx, "crt_hplanes", (size_t)Lg * pl.Mp2 * pl.Mp2, &d_hpl));
    const double target = 0.99 * (0.5 * gc.P) / (std::sqrt((double)M) * std::ldexp(1.0, pl.B - 1));
    crt_hscale_kernel<<<(unsigned)(pl.Mp2 / 64), 256, 0, ctx->stream>>>(d_H, M, pl.Mp2, pl.d_sexp, target, d_cs);
    OAK_HIP_CHECK(hipGetLastError());
    crt_hconvert_kernel<<<dim3((unsigned)(pl.Mp2 / 16), (unsigned)(pl.Mp2 / 256)), 256, 0, ctx->stream>>>(d_H, M, pl.Mp2, pl.d_sexp, d_cs, pl.md, gc, d_hpl);
    OAK_HIP_CHECK(hipGetLastError());
    const int ntiles = (int)((na + GT - 1) / GT), njt = (int)(pl.Mp2 / GTJ);
    const size_t lds = sizeof(cg_v4i) * G_NBUF * G_SLOT;
    OAK_CHECK(ensure_max_dynamic_lds((const void*)crt_gemm_i8_kernel));
    crt_gemm_i8_kernel<<<(unsigned)(8 * ((ntiles + 7) / 8) * njt), 512, lds, ctx->stream>>>(pl.d_planes, pl.rows_pad, d_hpl, (int)pl.Mp2, na, M, gc, d_cs, d_G, ldg,
                                                                                            ntiles);
    OAK_HIP_CHECK(hipGetLastError());
    ctx->crt_gemm_info[0] = Lg; ctx->crt_gemm_info[1] = BH;
    return OAK_OK;
}

}  // namespace oak

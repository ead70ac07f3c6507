// Adjoint panel of the gradient, G = Kfu H (grad.hip: dELBO/dKuf before the pair kernel contracts it with dK/dtheta; TF autodiff through
// SGPR.elbo in the reference, model_utils.py:168-173), on the INT8 matrix pipe from the residue planes the forward pass left behind.
//
// The fp64 GEMM is 2 M^2 N flops at 0.83 of the fp64 MFMA peak: 33.6 ms of a 73 ms forward+gradient step at the headline size, the largest
// single item.  The forward pass of the int8 route (crt.hip) has already written Kfu as integers K'[n, i] = rint(K[n, i] 2^a_i) of B <= 50
// bits in L residue planes, so the product only needs H in the same form:
//   H'[i, j] = rint(H[i, j] 2^(b_j - a_i)),  b_j from the column maximum of |H[i, j]| 2^-a_i  (BH bits, 44 by default),
//   sum_i K'[n, i] H'[i, j] = 2^b_j G[n, j]  up to the two roundings -- an integer below prod p / 2 for the first Lg moduli (13 at M = 1024),
//   one v_mfma_i32_32x32x32_i8 GEMM per modulus (K = M: int32 sums of <= 1024 products of bytes are exact), reconstructed in the
//   epilogue WITHOUT storing residues: x / P = sum_p r_p w_p mod 1 with w_p = ((P / p)^-1 mod p) / p, one fp64 fraction per output entry
//   carried in registers from plane to plane.
// Accuracy: the fraction is good to ~2^-45 (14 roundings of 127 * 2^-54 and of the running sum), i.e. G to 2^-45 of the a-priori bound
// M max|K'| max|H'|; the fp64 GEMM is good to 2^-53 sqrt(M) of the largest product.  That is ~1e-11 of a typical entry instead of
// ~1e-14 -- errors independent from entry to entry, contracted over N M entries by the pair kernel; the gradient tests hold the result
// to the fp64 route's.  It is NOT the exact arithmetic of the forward Phi (there every bit of the bound matters; here a gradient).
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
    double w[CRT_MAXL];      // ((P / p)^-1 mod p) / p
    double P;                // product of the L moduli
};

// bexp[j] = BH - 1 - e with max_i |H[i][j]| 2^(-sexp[i]) < 2^e  (zero columns and the padding: 0)
__global__ void __launch_bounds__(256) crt_hscale_kernel(const double* __restrict__ H, int64_t M, int64_t Mp2, const int* __restrict__ sexp, int BH,
                                                         int* __restrict__ bexp) {
    __shared__ double sh[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int64_t j = (int64_t)blockIdx.x * 64 + c;
    double m = 0.0;
    if (j < M)
        for (int64_t i = rg; i < M; i += 4) m = fmax(m, ldexp(fabs(H[i * M + j]), -sexp[i]));
    sh[rg][c] = m;
    __syncthreads();
    if (rg == 0 && j < Mp2) {
        m = fmax(fmax(sh[0][c], sh[1][c]), fmax(sh[2][c], sh[3][c]));
        int e = 0;
        frexp(m * (1.0 + 0x1p-20), &e);
        bexp[j] = (m > 0.0 && m < 1e300) ? BH - 1 - e : 0;
    }
}

// residue planes of H' in the operand layout [plane][k / 16][j][k % 16] (k = row index i of H: the contraction index)
__global__ void __launch_bounds__(256) crt_hconvert_kernel(const double* __restrict__ H, int64_t M, int64_t Mp2, const int* __restrict__ sexp,
                                                           const int* __restrict__ bexp, CrtMod md, int L, int8_t* __restrict__ hplanes) {
    const int64_t j = (int64_t)blockIdx.y * 256 + threadIdx.x;
    const int64_t kb = blockIdx.x;
    const int bj = bexp[j];
    double a[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int64_t i = kb * 16 + t;
        a[t] = (i < M && j < M) ? rint(ldexp(H[i * M + j], bj - sexp[i])) : 0.0;
    }
    const int64_t plane_bytes = Mp2 * Mp2;
    for (int q = 0; q < L; ++q) {
        const double p = (double)md.p[q], ip = md.inv[q], hp = 0.5 * p;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            double r = __builtin_fma(-rint(a[t] * ip), p, a[t]);
            r = r >= hp ? r - p : (r < -hp ? r + p : r);
            w[t >> 2] |= ((uint32_t)(int)r & 0xffu) << (8 * (t & 3));
        }
        *reinterpret_cast<uint4*>(hplanes + q * plane_bytes + (kb * Mp2 + j) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

constexpr int GT = 128, GKS = 64;                 // tile edge (rows n and columns j), k per stage
constexpr int G_NBUF = 4, G_D = G_NBUF - 1;
constexpr int G_SLOT = 2 * GT * GKS / 16;         // 16-byte units per slot: A [8 row groups][64 k][16 n], then B [4 k groups][128 j][16 k]

__device__ __forceinline__ void cg_glds(const cg_v4i* src, cg_v4i* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

__global__ void __launch_bounds__(256, 2) crt_gemm_i8_kernel(const int8_t* __restrict__ planes, int64_t rows_pad, const int8_t* __restrict__ hplanes, int Mp2,
                                                             int64_t na, int64_t M, const CrtGemmC gc, const int* __restrict__ bexp,
                                                             double* __restrict__ G, int64_t ldg, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) cg_v4i cg_lds[];
    const int njt = Mp2 / GT;
    const int xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    const int jt = t % njt, nt = (t / njt) * 8 + xcd;
    if (nt >= ntiles) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int spp = Mp2 / GKS;                    // stages per plane
    const cg_v4i* PA = reinterpret_cast<const cg_v4i*>(planes);
    const cg_v4i* PB = reinterpret_cast<const cg_v4i*>(hplanes);
    const int64_t a_plane = (rows_pad / 16) * Mp2, b_plane = (int64_t)(Mp2 / 16) * Mp2;
    const int64_t a_base = ((int64_t)nt * (GT / 16) + 2 * wave) * Mp2 + lane;      // this wave's two row groups
    const int64_t b_base = (int64_t)wave * Mp2 + (int64_t)jt * GT + lane;          // this wave's k group
    int fpl = 0, fks = 0;
    auto fill = [&](int slot) {
        const cg_v4i* sa = PA + (int64_t)fpl * a_plane + a_base + (int64_t)fks * GKS;
        const cg_v4i* sb = PB + (int64_t)fpl * b_plane + b_base + (int64_t)fks * 4 * Mp2;
        cg_v4i* d = cg_lds + slot * G_SLOT;
        cg_glds(sa, d + (2 * wave) * 64);
        cg_glds(sa + Mp2, d + (2 * wave + 1) * 64);
        cg_glds(sb, d + 512 + wave * 128);
        cg_glds(sb + 64, d + 512 + wave * 128 + 64);
        if (++fks == spp) { fks = 0; if (fpl + 1 < gc.L) ++fpl; }      // beyond the last stage: re-reads nobody consumes
    };
#pragma unroll
    for (int d = 0; d < G_D; ++d) fill(d);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int q = (lane >> 4) & 1, g = lane >> 5, i16 = lane & 15, c = lane & 31;
    const int a_lane = ((wr * 4 + q) * 64 + g * 16 + (i16 >> 1)) * 16 + (i16 & 1) * 8;      // bytes
    const int b_lane = (512 + g * 128 + wc * 64 + c) * 16;
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
    int cpl = 0, cks = 0, slot = 0, fslot = G_D;
    for (int s = 0; s < S; ++s) {
        fill(fslot);
        // LDS reads by inline asm: the compiler would put s_waitcnt vmcnt(0) in front of a transpose read it sees next to LDS-DMA
        // traffic (the stage being read landed before the previous barrier: the counted wait below).  Their completion is waited
        // for by hand -- the "+v" operands tie the MFMAs behind the waits.
        const unsigned sbase = lds0 + (unsigned)(slot * G_SLOT * 16);
        const unsigned pa = sbase + a_lane, pb = sbase + b_lane;
        cg_v2i al[2][2], ah[2][2];
        cg_v4i fb[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(al[kk][x]) : "v"(pa), "n"(x * 2048 + kk * 512));
                asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(ah[kk][x]) : "v"(pa), "n"(x * 2048 + kk * 512 + 128));
            }
#pragma unroll
            for (int y = 0; y < 2; ++y) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[kk][y]) : "v"(pb), "n"((kk * 256 + y * 32) * 16));
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
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = (slot + 1 == G_NBUF) ? 0 : slot + 1;
        fslot = (fslot + 1 == G_NBUF) ? 0 : fslot + 1;
        ++cks;
        if ((cks & 15) == 0 || cks == spp) {
            // <= 1024 products of bytes per accumulator (< 2^24: exact in fp32): r = acc mod p, fraction += r w_p (mod 1)
            const float pf = gc.p[cpl], ipf = gc.ip[cpl];
            const double w = gc.w[cpl];
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
    // x = P (fraction, centred), G = x 2^-b_j
    const int h = lane >> 5;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const int64_t col = (int64_t)jt * GT + wc * 64 + y * 32 + c;
        const double sc = ldexp(gc.P, -bexp[col]);
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
    int want = 50;
    if (const char* e = getenv("OAK_CRT_GEMM_BITS")) { const int v = atoi(e); if (v >= 30 && v <= 52) want = v; }
    int Lg = 0, BH = 0;
    double bits = 0.0;
    while (Lg < pl.md.L) {
        bits += std::log2((double)pl.md.p[Lg]); ++Lg;
        BH = (int)std::floor(bits - (double)pl.B - std::log2((double)M) + 0.5);      // M 2^(B-1) 2^(BH-1) < P / 2, half a bit to spare
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
        gc.w[i] = (double)cg_modinv(prod, p) / (double)p;
    }
    int* d_bexp = nullptr; int8_t* d_hpl = nullptr;
    OAK_CHECK(get_buf_t(ctx, "crt_bexp", (size_t)pl.Mp2, &d_bexp));
    OAK_CHECK(get_buf_t(ctx, "crt_hplanes", (size_t)Lg * pl.Mp2 * pl.Mp2, &d_hpl));
    crt_hscale_kernel<<<(unsigned)(pl.Mp2 / 64), 256, 0, ctx->stream>>>(d_H, M, pl.Mp2, pl.d_sexp, BH, d_bexp);
    OAK_HIP_CHECK(hipGetLastError());
    crt_hconvert_kernel<<<dim3((unsigned)(pl.Mp2 / 16), (unsigned)(pl.Mp2 / 256)), 256, 0, ctx->stream>>>(d_H, M, pl.Mp2, pl.d_sexp, d_bexp, pl.md, Lg, d_hpl);
    OAK_HIP_CHECK(hipGetLastError());
    const int ntiles = (int)((na + GT - 1) / GT), njt = (int)(pl.Mp2 / GT);
    const size_t lds = sizeof(cg_v4i) * G_NBUF * G_SLOT;
    OAK_CHECK(ensure_max_dynamic_lds((const void*)crt_gemm_i8_kernel));
    crt_gemm_i8_kernel<<<(unsigned)(8 * ((ntiles + 7) / 8) * njt), 256, lds, ctx->stream>>>(pl.d_planes, pl.rows_pad, d_hpl, (int)pl.Mp2, na, M, gc, d_bexp, d_G, ldg,
                                                                                            ntiles);
    OAK_HIP_CHECK(hipGetLastError());
    ctx->crt_gemm_info[0] = Lg; ctx->crt_gemm_info[1] = BH;
    return OAK_OK;
}

}  // namespace oak

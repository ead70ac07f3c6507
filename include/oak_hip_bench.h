/* Measurement hooks of liboak_hip.so -- NOT part of the drop-in boundary (include/oak_hip.h): nothing in the reference binds to
   these.  bench.py and tools/ time single kernels of the path through them with the inputs already resident in HBM. */
#ifndef OAK_HIP_BENCH_H
#define OAK_HIP_BENCH_H
#include "oak_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device-resident benchmarking hooks (inputs already in HBM) ---------------------------- */
/* Explicit Kuf panel for the rows set with oak_sgpr_set_data, written to a device buffer and not
   copied back: the "Gram GB/s" workload.  bytes_out = algorithmic bytes 8*(N*M + N*D + M*D). */
int oak_bench_gram_resident(oak_ctx* ctx, const oak_kernel_desc* desc, double* bytes_out);
/* The Cholesky of the O(M^3) tail alone (tf.linalg.cholesky at oak/utils.py:188,193): factors an n x n SPD test matrix
   (exponential kernel + 1e-3 I, built on the device) `reps` times; *ms_out = mean GPU time per factorisation (HIP events),
   *logdet_out (may be NULL) = log det from the factor, for checking against a host Cholesky of the same matrix. */
int oak_bench_potrf(oak_ctx* ctx, int64_t n, int32_t reps, double* ms_out, double* logdet_out);
/* The many-right-hand-side triangular solve alone (tf.linalg.triangular_solve at oak/utils.py:189 and in predict_f): every row
   b of B (host, nrhs x n row-major, overwritten) becomes the solution of L x = b (trans = 0) or L^T x = b (trans = 1) for the
   lower-triangular host matrix L (n x n); *ms_out (may be NULL) = mean GPU time of `reps` solves.  For residual checks and
   timing of the blocked solve the whitened route, the SVGP and large prediction batches run. */
int oak_bench_trsm(oak_ctx* ctx, const double* L, int64_t n, double* B, int64_t nrhs, int32_t trans, int32_t reps, double* ms_out);

/* Shape of the most recent int8 CRT accumulation of Phi (oak_sgpr_set_precision 2 / automatic; csrc/crt.hip): info[0] = residue planes
   (moduli), info[1] = bits of the scaled integers, info[2] = row splits, info[3] = rows per split, info[4] bit 0 = the planes came
   out of the Gram kernel's epilogue (else: stand-alone conversion pass), bit 1 = the most recent tail whitened Phi in double-double arithmetic, bits 8-15 / 16-23 = residue planes / bits of H' of the most
   recent gradient call's int8 adjoint GEMM (csrc/crt_gemm.hip; 0: the fp64 GEMM ran), info[5] = plane columns (M rounded up to 256).  All zero when
   the last statistics were formed by the fp64 / fp32 kernels.  bench.py prices the int8 SYRK with it. */
int oak_bench_crt_info(oak_ctx* ctx, int64_t* info6);

#ifdef __cplusplus
}
#endif
#endif /* OAK_HIP_BENCH_H */

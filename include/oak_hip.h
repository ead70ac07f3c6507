/*
 * oak_hip.h -- C ABI of the MI355X-native OAK Gram / SGPR-ELBO / predict / Sobol path.
 *
 * This is the drop-in boundary.  The reference (amzn/orthogonal-additive-gaussian-processes)
 * is pure Python and has no FFI of its own; the path sits behind the GPflow `Kernel` protocol and
 * the GPflow SGPR/GPR model methods.  Each entry point below names the reference interface it
 * replaces (file:line relative to the upstream repo root).  The only caller is the ctypes layer
 * `orthogonal-additive-gaussian-processes_amd/oak/_capi.py`; INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns an int status (OAK_OK == 0, negative == error) and records a
 *     thread-local message retrievable with oak_last_error();
 *   - all host buffers are caller-owned, C-contiguous, row-major, float64 (int32 where stated);
 *   - the library owns all device memory behind an opaque oak_ctx (one per device; a main HIP
 *     stream plus a side stream for work that overlaps it, both non-blocking: nothing runs on
 *     the legacy NULL stream); a ctx is not thread-safe; distinct ctxs may be driven from
 *     different host threads concurrently -- they share no mutable library state (per-kernel
 *     attributes and the RCCL loader are initialised once under a lock; oak_sync waits for the
 *     caller's ctx only) -- tests/test_gpu_sgpr.py::test_independent_contexts_are_thread_safe;
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails with
 *     OAK_E_HIP.
 */
#ifndef OAK_HIP_H
#define OAK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OAK_OK        0
#define OAK_E_ARG    -1   /* bad argument / unsupported configuration            */
#define OAK_E_HIP    -2   /* HIP runtime error (no device, OOM, launch failure)   */
#define OAK_E_NOTPD  -3   /* Cholesky met a non-positive pivot (tf InvalidArgumentError analogue) */
#define OAK_E_NCCL   -4   /* RCCL error                                           */
#define OAK_E_STATE  -5   /* call order violated (e.g. predict before posterior)  */

#define OAK_MAX_DIMS   64  /* sub-kernels per OAK kernel                          */
#define OAK_MAX_DEPTH  32  /* EFFECTIVE depth min(max_interaction_depth, num_dims) of the fused kernels (SGPR / GPR / SVGP paths,
                               gradients, Sobol; the reference's examples use depth = num_dims, up to 32 on pumadyn32nm).  e_r
                               vanishes for r > num_dims, so any max_interaction_depth is accepted while num_dims <= 32 */
#define OAK_MAX_DEPTH_DESC 64 /* max_interaction_depth a description may carry; beyond an effective depth of 32 only the explicit
                               Gram entry points (oak_gram / oak_gram_diag: OAKKernel.K / K_diag) evaluate it, by a generic kernel */

/* dim_type: which constrained base kernel a sub-kernel is */
#define OAK_DIM_RBF          0  /* oak/ortho_rbf_kernel.py:20-177         */
#define OAK_DIM_BINARY       1  /* oak/ortho_binary_kernel.py:13-59       */
#define OAK_DIM_CATEGORICAL  2  /* oak/ortho_categorical_kernel.py:14-74  */

/* measure: input measure of an RBF sub-kernel (oak/input_measures.py:16-78) */
#define OAK_MEAS_NONE       0   /* unconstrained RBF (constrain_orthogonal=False, oak_kernel.py:199-210) */
#define OAK_MEAS_GAUSSIAN   1   /* p0 = mu, p1 = var                      */
#define OAK_MEAS_UNIFORM    2   /* p0 = a,  p1 = b                        */
#define OAK_MEAS_EMPIRICAL  3   /* data: K locations then K weights       */
#define OAK_MEAS_MOG        4   /* data: K means, K variances, K weights  */

/*
 * Plain-old-data description of an OAKKernel (oak/oak_kernel.py:59-221) at its current
 * (constrained) parameter values.  All pointers are host pointers, read during the call only.
 */
typedef struct oak_kernel_desc {
    int32_t num_dims;           /* D = len(kernel.kernels)                                   */
    int32_t max_depth;          /* R = max_interaction_depth                                 */
    int32_t share_var;          /* share_var_across_orders (oak_kernel.py:212-221)           */
    int32_t n_order_var;        /* R+1 if share_var else 1                                   */
    const double*  order_var;   /* [n_order_var] kernel.variances                            */
    const int32_t* dim_type;    /* [D] OAK_DIM_*                                             */
    const int32_t* active_col;  /* [D] column of X the sub-kernel reads (active_dims)        */
    const double*  lengthscale; /* [D] base_kernel.lengthscales (RBF dims)                   */
    const double*  base_var;    /* [D] base_kernel.variance / kernel.variance                */
    const int32_t* measure;     /* [D] OAK_MEAS_* (RBF dims)                                 */
    const double*  meas_p0;     /* [D] gaussian mu | uniform a | binary p0                   */
    const double*  meas_p1;     /* [D] gaussian var | uniform b                              */
    const int32_t* meas_k;      /* [D] #locations | #mixture comps | #categories             */
    const int32_t* meas_off;    /* [D] offset (in doubles) of this dim's block in meas_data  */
    const double*  meas_data;   /* empirical: loc[K],w[K]; mog: mu[K],var[K],w[K];
                                   categorical: row-major C*C table B WITHOUT the variance
                                   factor (A - Ap Ap^T / p^T A p, ortho_categorical_kernel.py:34-42)
                                   followed by the C-vector p                                */
    int32_t meas_data_len;
    int32_t grad_base_var;      /* gradient calls: also return d/d base_var (0 when the base variances are the
                                   constants of share_var_across_orders=True, oak_kernel.py:163-166,179,187)  */
    /* Grouped sub-kernels (OAKKernel(active_dims=[[0, 1], [2]]), oak_kernel.py:74-82,199-210: an unconstrained RBF over several
       columns with one lengthscale, i.e. the product of its one-column RBFs).  extra_col_off [D + 1] / extra_cols: the columns
       of sub-kernel d BEYOND active_col[d] are extra_cols[extra_col_off[d] .. extra_col_off[d + 1]).  Both NULL = every
       sub-kernel reads one column.  Taken by oak_gram*, the SGPR / GPR / SVGP objectives, their hyper-parameter gradients,
       predictions, components and the inducing-input gradient (every column of a group gets its entry of gradZ_out);
       oak_sobol* and oak_gram_f32 refuse a grouped description (OAK_E_ARG), the fp32 statistics mode falls back to fp64. */
    const int32_t* extra_col_off;
    const int32_t* extra_cols;
} oak_kernel_desc;

typedef struct oak_ctx oak_ctx;

/* ---- runtime ------------------------------------------------------------------------------ */
const char* oak_last_error(void);
const char* oak_version(void);
int oak_device_count(int* count);
int oak_ctx_create(int device, oak_ctx** out);
int oak_ctx_destroy(oak_ctx* ctx);
/* Streams and events of destroyed contexts are kept in a per-process pool and handed to the next context (hipStreamDestroy can
   deadlock against the runtime's event thread while other host threads use the device: DESIGN section 9).  This call destroys
   the pooled ones; call it once at the end of the process, after every context has been destroyed and while no other thread
   uses the device (the Python binding does so at interpreter exit).  Optional: an ordinary process may simply exit. */
int oak_runtime_shutdown(void);
int oak_sync(oak_ctx* ctx);
/* Post-mortem aid for a stalled process: one text record per live context -- whether its two streams have drained, its
   communicator, and the last 16 phases / collectives it enqueued with their age.  Call it from ANOTHER host thread than the
   stuck one (tests/conftest.py's watchdog, tools/soak.py). */
int oak_debug_state(char* buf, int64_t cap);
/* GPU time (ms, hipEvents on the ctx stream) accumulated per phase since oak_reset_timings; name in {"featurize","gram",
   "trsm","syrk","reduce","allreduce","tail","total","bwd_tail","bwd_gemm","bwd_gram","bwd_small","bwd_z","predict",
   "kmeans","kmeans_pp","flow_forward"}; count = number of
   times the phase ran ("gram","syrk","bwd_gemm","bwd_gram" are single kernel launches per panel pass). */
int oak_last_timing(oak_ctx* ctx, const char* name, double* ms, int32_t* count);
int oak_reset_timings(oak_ctx* ctx);
int oak_device_mem_info(oak_ctx* ctx, double* free_bytes, double* total_bytes);

/* ---- Gram (replaces OAKKernel.K / K_diag, oak/oak_kernel.py:251-278, and through it
 *      OrthogonalRBFKernel.K / K_diag ortho_rbf_kernel.py:157-177, OrthogonalBinary.K :40-59,
 *      OrthogonalCategorical.K :55-74, compute_additive_terms oak_kernel.py:223-249) ---------- */
/* out[n1 x n2] = K(X1, X2);  X2 == NULL means X2 = X1.  ldx = row stride of X1/X2 in doubles. */
int oak_gram(oak_ctx* ctx, const oak_kernel_desc* desc,
             const double* X1, int64_t n1, const double* X2, int64_t n2, int32_t ldx,
             double* out);
/* out[n] = K_diag(X) */
/* A/B switch of the explicit Gram entry points (oak_gram, oak_gram_diag): 0 (default) = this library's arithmetic --
   squared distance formed directly as (x/l - z/l)^2, elementary symmetric polynomials by the exact-sum recurrence;
   1 = the REFERENCE's arithmetic reproduced on the device -- GPflow's expanded |x/l|^2 + |z/l|^2 - 2 (x/l)(z/l), exp,
   power sums and the Newton-Girard alternating sum (oak/oak_kernel.py:236-249, ortho_rbf_kernel.py:157-172) -- so that
   "identical to the reference" can be shown entry by entry and the two deliberate deviations quantified
   (tests/test_gpu_gram.py).  A debug aid: one thread per entry, not a fast path. */
int oak_set_gram_form(oak_ctx* ctx, int32_t form);
/* out[n1 x n2] (float) = the fp32 Kuf panel that the opt-in fp32 statistics mode (oak_sgpr_set_precision) builds and feeds to
   its fp32-MFMA SYRK: exposed so that the mode's Gram can be checked against the reference arithmetic (<= 1e-5 of max|K|). */
int oak_gram_f32(oak_ctx* ctx, const oak_kernel_desc* desc, const double* X1, int64_t n1, const double* X2, int64_t n2,
                 int32_t ldx, float* out);
int oak_gram_diag(oak_ctx* ctx, const oak_kernel_desc* desc,
                  const double* X, int64_t n, int32_t ldx, double* out);
/* KernelComponenent.K (oak_kernel.py:300-320): sigma2_{|S|} * prod_{d in S} k_d; subset = indices
   into the D sub-kernels; apply_order_var mirrors share_var_across_orders of the component. */
int oak_gram_component(oak_ctx* ctx, const oak_kernel_desc* desc,
                       const int32_t* subset, int32_t subset_len, int32_t apply_order_var,
                       const double* X1, int64_t n1, const double* X2, int64_t n2, int32_t ldx,
                       double* out);
int oak_gram_component_diag(oak_ctx* ctx, const oak_kernel_desc* desc,
                            const int32_t* subset, int32_t subset_len, int32_t apply_order_var,
                            const double* X, int64_t n, int32_t ldx, double* out);

/* ---- SGPR (replaces gpflow.models.SGPR as constructed at oak/model_utils.py:149-157:
 *      elbo / predict_f / the alpha of oak/utils.py:180-198) --------------------------------- */
/* Upload training data (X [N x ldx], Y [N], single output column) and inducing inputs Z [M x ldx].
   Data stay resident in HBM until replaced. */
int oak_sgpr_set_data(oak_ctx* ctx, const double* X, const double* Y, int64_t N, int32_t ldx);
/* Replace the target column of the rows set by oak_sgpr_set_data (same N; X stays resident).  A model with P output columns
   (GPflow's N x P Y: independent outputs sharing kernel and noise, oak/utils.py:182-198 is written for it) is the sum of P
   single-output bounds; the host mirror evaluates them one after the other through this call. */
int oak_sgpr_set_targets(oak_ctx* ctx, const double* Y, int64_t N);
/* The other output columns of a P-column model (oak/utils.py:182-198: err, A err, c are N x P, M x P there): Yt holds
   columns 1 .. n_extra of Y, ONE COLUMN PER ROW ([n_extra x N], contiguous).  From then on oak_sgpr_elbo /
   oak_sgpr_elbo_grad(_z) return the bound and the gradient SUMMED over the 1 + n_extra outputs while everything that does
   not depend on y -- the Kuf panel, Phi, both Cholesky factors, the M x M adjoints -- is computed once: each extra output
   costs one more column of a pass over the resident Kuf panel (psi_p = Kuf y_p), two M-sized triangular solves and, in the
   gradient, one more rank-one term of the adjoint panel.  n_extra = 0 forgets them; oak_sgpr_set_data also does.
   (oak_sgpr_last_terms keeps reporting output 0's terms; the fp32 statistics mode is not combined with extra columns.) */
int oak_sgpr_set_extra_targets(oak_ctx* ctx, const double* Yt, int64_t N, int32_t n_extra);
/* Which output's posterior oak_sgpr_alpha / oak_sgpr_predict / oak_component_predict callers see (0 after every
   evaluation; 1 .. n_extra: the extra columns). */
int oak_sgpr_select_output(oak_ctx* ctx, int32_t p);
int oak_sgpr_set_inducing(oak_ctx* ctx, const double* Z, int64_t M, int32_t ldx);
/* Row budget of the N x M Kuf panel kept in HBM per pass (0 = library default). */
int oak_sgpr_set_panel_rows(oak_ctx* ctx, int64_t rows);
/* Sufficient statistics of the local rows, left on the device in the packed layout
   [Phi (M*M) | psi (M) | kappa | yy | n_rows | n_whitened | n_parts], Phi = Kuf Kuf^T, psi = Kuf y,
   kappa = sum K_diag(X), yy = y^T y; n_parts = 1 and n_whitened = 1 if this shard took the whitened route, so that
   a SUM of packed vectors carries how many of its shards whitened: the tail (and oak_sgpr_set_stats) fail with
   OAK_E_STATE unless n_whitened is 0 or n_parts. */
int oak_sgpr_local_stats(oak_ctx* ctx, const oak_kernel_desc* desc, double jitter);
/* Solve route.  1 = "phi": accumulate Phi = Kuf Kuf^T (M^2 N flops) and whiten the M x M result in the
   tail; deviation from GPflow's op order grows like cond(Kuu)*eps.  2 = "whitened": apply L^-1 to each
   Kuf column first, exactly GPflow's A = L^-1 Kuf (oak/utils.py:189), 2x the flops; the packed Phi slot
   then holds W = L^-1 Phi L^-T.  0 = auto: whitened while N*M <= 2^24; above that oak_sgpr_elbo and
   oak_sgpr_elbo_grad whiten only when chol(Kuu) looks ill-conditioned, (max diag L / min diag L)^2 > 1e3,
   which keeps the result within ~1e-10 of the literal route (the stand-alone oak_sgpr_local_stats uses the
   size rule alone).  Where the int8 route of oak_sgpr_set_precision runs -- on one rank, or under a communicator whose ranks all declared
   oak_sgpr_set_global_rows (the shards' Phi are then summed exactly: two fixed-point limbs per entry, one more all-reduce of M^2 doubles;
   RCCL and loopback communicators by default, the host exchange with OAK_COMM_DD=1) -- auto never whitens: Phi is then exact (kept as a
   double-double) and the tail whitens it with double-double M^3 products when chol(Kuu) looks ill-conditioned (estimate > 1e2;
   csrc/ddgemm.hip) -- the phi route at the whitened route's accuracy, without the N-sized triangular solve (tests/test_gpu_crt.py).
   Under a communicator N is the row count over ALL ranks: what oak_sgpr_set_global_rows declared
   (either every rank declares it or none does), else one scalar all-reduce on EVERY auto-route evaluation (never a
   per-rank cache: the sequence of collectives is then the same on all ranks whatever their history); the conditioning
   decision is rank 0's, shared with the other ranks; so all ranks take the same route even when their shards differ. */
int oak_sgpr_set_route(oak_ctx* ctx, int32_t route);
/* Rows over all shards when this ctx holds one shard and the statistics are exchanged outside the library
   (oak_sgpr_get_stats / oak_sgpr_set_stats): the auto route's size rule uses it.  0 = unknown (default). */
int oak_sgpr_set_global_rows(oak_ctx* ctx, int64_t n_total);
/* Arithmetic of the N-sized statistics.  -1 (default): mode 2 where it pays (phi route; M >= 640 and N >= 32768, or M >= 512 and N M^2 >= 2^38), mode 0 elsewhere.
   0: the fp64 kernels throughout, the reference's precision (oak/oak_kernel.py:31-32 sets float64 globally).  1: "fp32 statistics" -- the Kfu panel is generated in fp32 (hardware v_exp_f32) and Phi's row-split
   partials are formed by fp32 MFMA (fp32 accumulation within one split of a few thousand rows only), everything else --
   featurisation, kappa, the cross-split sums, the O(M^3) tail, prediction, Sobol, every gradient -- stays fp64.  Applies to
   forward evaluations through oak_sgpr_elbo on the phi route; the whitened route, oak_sgpr_elbo_grad and the stand-alone
   oak_sgpr_local_stats ignore it.  An fp32 error in Phi reaches W = L^-1 Phi L^-T divided by lambda_min(Kuu), so the mode is
   honoured only when chol(Kuu) looks well conditioned ((max diag L / min diag L)^2 <= 1e2 -- the auto route's estimate, which
   under-reads cond(Kuu) by 20-600x);
   otherwise that evaluation runs in fp64 -- oak_sgpr_stats_precision reports what the last statistics used.  Not the
   reference's arithmetic: ELBO within 1e-5 relative of the fp64 path on the benchmark problems (7e-6 at the headline size, terms <= 2e-6) (tests/test_gpu_fp32.py),
   never used for `value`.
   2: "int8 CRT" -- Phi = Kuf Kuf^T is accumulated EXACTLY on the int8 matrix pipe (csrc/crt.hip): every panel entry is scaled by
   an a-priori power of two per column and rounded to a 48-bit integer, split into residues modulo 15-16 coprime moduli <= 254,
   the residue planes are multiplied by v_mfma_i32_32x32x32_i8 with exact int32 / int64 sums, and the integer Gram matrix is
   reconstructed by the Chinese remainder theorem.  The one rounding per entry (2^-48 of the column bound) replaces the N
   roundings of an fp64 accumulation: this is not a lower precision (Phi is closer to the exact sum than the fp64 SYRK's).  phi
   route only (a whitened panel has no a-priori bound); M <= 4096, N >= 4096; otherwise the evaluation runs the fp64 kernels --
   oak_sgpr_stats_precision tells.
   Gradient calls (oak_sgpr_elbo_grad / _grad_z) whose forward pass took this route with the whole panel in one chunk also form the
   adjoint panel Kfu H of the backward pass from those residue planes on the int8 pipe (csrc/crt_gemm.hip) -- when chol(Kuu) looks
   well-conditioned ((max diag L / min diag L)^2 <= 1e2): its operands are 49- and 46-49-bit fixed-point numbers per column and the
   product cancels like cond(Kuu) (gradient within 1e-14 of the fp64 kernels' at the headline size, 3e-13 at an estimate of 200;
   tests/test_gpu_crt.py).  Otherwise, and with OAK_CRT_GEMM=0 in the environment, the fp64 MFMA GEMM runs (modes 0 and 1: always). */
int oak_sgpr_set_precision(oak_ctx* ctx, int32_t mode);
int oak_sgpr_stats_precision(oak_ctx* ctx, int32_t* mode);
int oak_sgpr_stats_whitened(oak_ctx* ctx, int32_t* flag);
int64_t oak_sgpr_stats_len(oak_ctx* ctx);                    /* M*M + M + 5 */
int oak_sgpr_get_stats(oak_ctx* ctx, double* packed_out);    /* D2H copy of the packed buffer */
/* H2D (externally reduced stats); `whitened` must agree with the counts the packed vector carries */
int oak_sgpr_set_stats(oak_ctx* ctx, const double* packed, int32_t whitened);
/* Replicated O(M^3) tail on the packed stats: L=chol(Kuu+jitter I), AAT, LB, c, alpha, ELBO
   (gpflow SGPR.elbo; op order of oak/utils.py:187-198).  terms_out (may be NULL) receives
   [sum log diag LB, c^T c, tr(AAT), kappa, yy, n_rows, log det Kuu, (max diag L / min diag L)^2 when the evaluation
   computed that conditioning estimate (auto route on a large problem, fp32 mode), else 0]. */
int oak_sgpr_tail(oak_ctx* ctx, const oak_kernel_desc* desc, double noise_var, double jitter,
                  double* elbo_out, double* terms_out);
/* Convenience: local_stats + (all-reduce when a communicator is attached) + tail. */
int oak_sgpr_elbo(oak_ctx* ctx, const oak_kernel_desc* desc, double noise_var, double jitter,
                  double* elbo_out);
/* The terms_out vector of the most recent successful tail, whichever entry point ran it (oak_sgpr_elbo and
   oak_sgpr_elbo_grad included): the kernel-dependent pieces of the bound, for term-by-term parity checks. */
int oak_sgpr_last_terms(oak_ctx* ctx, double* terms_out);
/* alpha [M] of oak/utils.py:197-198; valid after a successful tail/elbo call. */
int oak_sgpr_alpha(oak_ctx* ctx, double* alpha_out);
/* The "effective L" that get_model_sufficient_statistics(m, get_L=True) returns for a sparse model
   (oak/utils.py:199-204): inv(L^-1 - LB^-1 L^-1), M x M row-major.  After oak_sgpr_elbo / _tail. */
int oak_sgpr_effective_L(oak_ctx* ctx, double* L_out);
/* SGPR.predict_f(full_cov=False): mean[Ns], var[Ns]; valid after tail/elbo with the same desc. */
int oak_sgpr_predict(oak_ctx* ctx, const oak_kernel_desc* desc,
                     const double* Xs, int64_t Ns, int32_t ldx, double* mean, double* var);
/* d ELBO / d (constrained) parameters, packed as
   [lengthscale (D) | base_var (D) | order_var (n_order_var) | noise_var | dTable (meas_data_len,
   only categorical table entries are filled)].  Requires a preceding tail/elbo call. */
int64_t oak_grad_len(const oak_kernel_desc* desc);
int oak_sgpr_elbo_grad(oak_ctx* ctx, const oak_kernel_desc* desc, double noise_var, double jitter,
                       double* elbo_out, double* grad_out);
/* Same, plus the gradient with respect to the inducing inputs (create_model_oak(zfixed=False),
   oak/model_utils.py:156-157; TensorFlow autodiff through Kuf and Kuu in the reference).  gradZ_out is M x ldx
   row-major like Z; columns that no RBF sub-kernel reads, and binary / categorical columns, get 0.  Costs one more
   pass over the N x M pairs (register-resident kernels for an effective depth <= 4 with <= 32 sub-kernels or <= 8 with <= 16,
   a general kernel -- about 4-5x slower -- for every other shape, incl. grouped sub-kernels).  gradZ_out may be NULL. */
int oak_sgpr_elbo_grad_z(oak_ctx* ctx, const oak_kernel_desc* desc, double noise_var, double jitter,
                         double* elbo_out, double* grad_out, double* gradZ_out);

/* ---- GPR (replaces gpflow.models.GPR constructed at oak/model_utils.py:159;
 *      in-tree mirror oak/utils.py:206-211) -------------------------------------------------- */
int oak_gpr_set_data(oak_ctx* ctx, const double* X, const double* Y, int64_t N, int32_t ldx);
/* Likewise for the full GP. */
int oak_gpr_set_targets(oak_ctx* ctx, const double* Y, int64_t N);
int oak_gpr_log_marginal(oak_ctx* ctx, const oak_kernel_desc* desc, double noise_var, double* out);
int oak_gpr_alpha(oak_ctx* ctx, double* alpha_out);
/* L = chol(K + noise I) of the full GP, N x N row-major (get_model_sufficient_statistics(get_L=True),
   oak/utils.py:206-211).  After oak_gpr_log_marginal. */
int oak_gpr_chol(oak_ctx* ctx, double* L_out);         /* cholesky_solve(L, Y), utils.py:211 */
int oak_gpr_predict(oak_ctx* ctx, const oak_kernel_desc* desc,
                    const double* Xs, int64_t Ns, int32_t ldx, double* mean, double* var);
int oak_gpr_log_marginal_grad(oak_ctx* ctx, const oak_kernel_desc* desc, double noise_var,
                              double* out, double* grad_out);

/* ---- SVGP, whitened, diagonal q, Bernoulli likelihood: the model the classification example puts on the OAK
 *      kernel (examples/uci/uci_classification_train.py:108-116: gpflow.models.SVGP(kernel, Bernoulli(invlink),
 *      Z, whiten=True, q_diag=True); its elbo/predict_f/predict_log_density, and the posterior pieces
 *      oak/utils.py:174-179 reads).  Data and inducing inputs come from oak_sgpr_set_data (Y in {0,1}) and
 *      oak_sgpr_set_inducing.  q_mu, q_sqrt [M].  (gh_x, gh_w) [n_gh <= 64] is numpy's hermgauss rule
 *      (GPflow: 20 nodes); link 0: p = sigmoid(f)(1 - 2 eps) + eps (the example's inv_logit, eps = 1e-3),
 *      link 1: p = Phi(f)(1 - 2 eps) + eps (GPflow's inv_probit). ------------------------------------- */
/* elbo = sum_n E_q[log p(y_n | f_n)] - KL(q(v) || N(0, I)).  grad_out NULL: forward only; otherwise grad_out
   [oak_grad_len, noise slot 0], grad_qmu [M], grad_qsqrt [M] receive d elbo / d (constrained parameter).
   Under a communicator (oak_comm_init) the rows are one rank's shard: the row sums are all-reduced inside. */
int oak_svgp_elbo_grad(oak_ctx* ctx, const oak_kernel_desc* desc, const double* q_mu, const double* q_sqrt,
                       double jitter, const double* gh_x, const double* gh_w, int32_t n_gh, int32_t link,
                       double link_eps, double* elbo_out, double* grad_out, double* grad_qmu, double* grad_qsqrt);
/* predict_f: mean, var [Ns].  Ys/logdens non-NULL: also predict_log_density(Xs, Ys) [Ns] (needs the rule). */
int oak_svgp_predict(oak_ctx* ctx, const oak_kernel_desc* desc, const double* q_mu, const double* q_sqrt,
                     double jitter, const double* Xs, int64_t Ns, int32_t ldx, double* mean, double* var,
                     const double* Ys, double* logdens, const double* gh_x, const double* gh_w, int32_t n_gh,
                     int32_t link, double link_eps);
/* posterior.alpha [M] and (L_out non-NULL) chol(inv(posterior.Qinv)) [M x M], oak/utils.py:174-179;
   OAK_E_NOTPD when some q_sqrt >= 1 (the reference's Cholesky fails there). */
int oak_svgp_posterior(oak_ctx* ctx, const oak_kernel_desc* desc, const double* q_mu, const double* q_sqrt,
                       double jitter, double* alpha_out, double* L_out);

/* ---- Sobol (replaces compute_sobol_oak, oak/utils.py:338-435, with compute_L :221-240,
 *      compute_L_binary_kernel :243-272, compute_L_categorical_kernel :275-309,
 *      compute_L_empirical_measure :312-335) ------------------------------------------------- */
/* Xc [n x ldx] = inducing inputs (sparse) or training inputs (full GP); alpha [n].
   subsets: concatenated dim indices, subset_off [n_subsets+1].  use_order_var mirrors
   share_var_across_orders (utils.py:376-382).  out[n_subsets] = alpha^T (prod_d L_d) alpha. */
int oak_sobol(oak_ctx* ctx, const oak_kernel_desc* desc, const double* Xc, int64_t n, int32_t ldx,
              const double* alpha, const int32_t* subsets, const int32_t* subset_off,
              int32_t n_subsets, int32_t use_order_var, double delta, double mu, double* out);
/* The same evaluation as a collective over the context's communicator (oak_comm_init*): every rank makes the
   identical call and receives every term; the pair rows of the Gram of products (or blocks of terms) are sharded
   over the ranks and summed on the device.  Without a communicator it is oak_sobol. */
int oak_sobol_collective(oak_ctx* ctx, const oak_kernel_desc* desc, const double* Xc, int64_t n, int32_t ldx,
                         const double* alpha, const int32_t* subsets, const int32_t* subset_off,
                         int32_t n_subsets, int32_t use_order_var, double delta, double mu, double* out);
/* How oak_sobol evaluates the terms: 0 = automatic (cost model), 1 = one workgroup per term (a fused
   product-reduction over the stacked L_d, any subset size), 2 = Gram of products (every term one entry of the
   weighted fp64-MFMA Gram matrix of [1 | L_a | L_a L_b | L_a L_b L_c] over the index pairs; subsets of <= 6
   distinct dims, OAK_E_ARG otherwise). */
int oak_sobol_set_path(oak_ctx* ctx, int32_t path);
/* About the most recent oak_sobol call: info4 = {path taken (1 | 2), Gram columns, largest relative disagreement
   between the three pairings ab|cd, ac|bd, ad|bc of an order-4 term (0 when none was evaluated), pair rows}. */
int oak_sobol_last_info(oak_ctx* ctx, double* info4);
/* One per-dimension integral matrix L_d(v) [n x n] as the reference's compute_L* helpers return it
   (v = the helper's `variance` argument; delta is used as a standard deviation, utils.py:116-165). */
int oak_sobol_L(oak_ctx* ctx, const oak_kernel_desc* desc, int32_t dim, double v, double delta, double mu,
                const double* Xc, int64_t n, int32_t ldx, double* out);
/* cov_X_s(X) [n] and var_s of RBF sub-kernel `dim` (oak/ortho_rbf_kernel.py:47-152); either output may be NULL. */
int oak_cov_x_s(oak_ctx* ctx, const oak_kernel_desc* desc, int32_t dim, const double* X, int64_t n, int32_t ldx,
                double* c_out, double* var_s_out);
/* [e_0 .. e_R] of D stacked length-n arrays (OAKKernel.compute_additive_terms, oak/oak_kernel.py:223-249):
   mats [D x n] -> out [(R+1) x n]. */
int oak_additive_terms(oak_ctx* ctx, const double* mats, int32_t D, int64_t n, int32_t R, double* out);
/* Per-term predictive means (get_prediction_component, oak/utils.py:491-530):
   out[n_subsets x ns] = (sigma2_|S| prod_{d in S} k_d(Xs, Xc)) alpha */
int oak_component_predict(oak_ctx* ctx, const oak_kernel_desc* desc, const double* Xs, int64_t ns,
                          const double* Xc, int64_t n, int32_t ldx, const double* alpha,
                          const int32_t* subsets, const int32_t* subset_off, int32_t n_subsets,
                          int32_t use_order_var, double* out);

/* ---- multi-GPU (new; the reference has no distributed code).  One process per GPU; the packed
 *      statistics are summed with one RCCL all-reduce (reduce-scatter + all-gather) over xGMI. */
int oak_comm_unique_id(char* id_out_128);                      /* ncclGetUniqueId on rank 0   */
int oak_comm_init(oak_ctx* ctx, const char* id_128, int32_t nranks, int32_t rank);
/* Test communicator: pretends `nranks` ranks hold the SAME local rows, so every all-reduce multiplies by nranks.
   Runs the whole N > 1 code path on one GPU: the result must equal a single-rank run on the rows stacked nranks times
   (tests/test_gpu_distributed.py). */
int oak_comm_init_loopback(oak_ctx* ctx, int32_t nranks);
/* Host-exchange communicator: every sum over ranks the library needs (packed statistics, gradient records, route and
   precision decisions) is delegated to `fn`, called on a HOST copy of the buffer: it must replace buf[0..n) by the sum over
   all ranks and return 0, identically on every rank (a socket / MPI / gloo control plane of the host language).  The whole
   N > 1 path then runs where RCCL cannot: several ranks on one GPU, no xGMI fabric.  oak/distributed.py uses it when the
   exchange is set to "host". */
typedef int (*oak_host_allreduce_fn)(double* buf, int64_t n, void* user);
int oak_comm_init_host(oak_ctx* ctx, int32_t nranks, int32_t rank, oak_host_allreduce_fn fn, void* user);
/* Which librccl.so was loaded ($ROCM_PATH/lib is preferred: the library that belongs to the header this was compiled
   against), its ncclGetVersion and the header's NCCL_VERSION_CODE; a major-version mismatch is refused at load time. */
int oak_comm_info(char* path_out, int64_t cap, int32_t* version_out, int32_t* header_version_out);
int oak_comm_destroy(oak_ctx* ctx);
/* In-place sum over the ranks of the packed statistics AND, with extra target columns set (oak_sgpr_set_extra_targets), of their
   [Kuf y_p | y_p^T y_p]: local_stats -> allreduce_stats -> tail exchanges exactly what oak_sgpr_elbo does. */
int oak_comm_allreduce_stats(oak_ctx* ctx);
int oak_comm_allreduce_host(oak_ctx* ctx, double* buf, int64_t n); /* small host vector (gradients) */
/* All-gather of variable-sized blocks of a host vector: buf has sum(counts) doubles, rank r owns counts[r] of them at
   offset sum(counts[:r]); on return every rank holds all blocks (predictions are sharded with no other exchange).
   RCCL communicator: a device all-gather (one grouped ncclBroadcast per block); host communicator: the sum of
   zero-padded copies through the callback.  n_counts must equal the communicator's size (1 without one):
   OAK_E_STATE otherwise -- blocks announced for ranks the context cannot reach are never silently left empty. */
int oak_comm_allgatherv(oak_ctx* ctx, double* buf, const int64_t* counts, int32_t n_counts);

/* ---- input preprocessing ----------------------------------------------------------------------- */
/* KL objective of the per-feature normalising flow and its gradient (oak/normalising_flow.py:79-85
   Normalizer.KL_objective, minimised per continuous feature in oak_model.fit, oak/model_utils.py:305-317):
   y = sinh((asinh(z) + skewness) * tailweight), z = scale * (g + shift); KL = mean(y^2)/2 - mean(log|dy/dx|).
   g = log(x - offset) (use_log = 1; the host forms it once, offset = min(x) - 1) or x itself (use_log = 0).
   Pass g on the first call for a sample; pass NULL afterwards to evaluate on the device-resident copy.
   grad_out[4] = d KL / d (scale, shift, skewness, tailweight), constrained values; may be NULL. */
int oak_flow_objective(oak_ctx* ctx, const double* g, int64_t n, int32_t use_log, double scale, double shift,
                       double skewness, double tailweight, double* objective_out, double* grad_out);

/* Column-wise forward transform of a design matrix (oak_model._transform_x / apply_normalise_flow,
   oak/model_utils.py:179-191, :462-476).  X, out: N x ldx row-major (out may alias nothing); columns >= D are
   copied.  kind[d]: 0 copy, 1 flow on x, 2 flow on log(x - offset), 3 affine (x - mean) / std.
   params[5*d ..]: {offset, scale, shift, skewness, tailweight} for flows, {mean, std, 0, 0, 0} for affine. */
int oak_flow_forward(oak_ctx* ctx, const double* X, int64_t N, int32_t ldx, int32_t D, const int32_t* kind,
                     const double* params, double* out);

/* ---- inducing-point initialisation ----------------------------------------------------------- */
/* Lloyd k-means from given seeds, following scikit-learn's single-run loop (_kmeans_single_lloyd) that the
   reference reaches through sklearn.cluster.KMeans(n_clusters=K).fit(X).cluster_centers_
   (oak/model_utils.py:31-41 get_kmeans_centers, used by oak_model.fit :377-391; oak/utils.py:533-574 for the
   continuous block of the mixed-type initialisers).  X is N x D row-major (leading dimension ldx), D <= 64;
   init_centres and centres_out are K x D; labels_out (N, may be NULL) are the labels w.r.t. the returned centres;
   tol is ABSOLUTE (scikit-learn passes tol * mean(var(X, axis=0))).  Stops on unchanged labels, on
   sum_k |c_new - c_old|^2 <= tol, or after max_iter iterations; *n_iter_out = iterations run.  Empty clusters take
   the points farthest from their own centre (largest distance first, ties by lower index).  Deterministic. */
int oak_kmeans(oak_ctx* ctx, const double* X, int64_t N, int32_t D, int32_t ldx, int32_t K,
               const double* init_centres, int32_t max_iter, double tol, double* centres_out,
               int32_t* labels_out, double* inertia_out, int32_t* n_iter_out);

/* k-means++ seeding (greedy, with local trials) as scikit-learn's _kmeans_plusplus runs it inside KMeans.fit
   before the Lloyd loop above.  The caller draws the randomness from ITS generator in scikit-learn's order
   (first_index = rs.choice(N, p=uniform); uniforms = rs.uniform(size=(K-1, n_trials)), n_trials = 2 + int(log K),
   at most 16) so that a numpy RandomState reproduces scikit-learn's picks; everything else runs on the device
   without a host synchronisation per centre.  N <= 2^24, D <= 64.  centres_out K x D, indices_out (K, may be NULL).
   Indices can differ from scikit-learn's only when a random threshold lands within rounding of a cumulative-sum
   boundary (direct-form distances and blocked sums here vs GEMM-form distances and a sequential cumsum there). */
int oak_kmeans_plusplus(oak_ctx* ctx, const double* X, int64_t N, int32_t D, int32_t ldx, int32_t K,
                        int64_t first_index, const double* uniforms, int32_t n_trials, double* centres_out,
                        int64_t* indices_out);

/* Measurement hooks (oak_bench_*: resident Gram pass, the tail's Cholesky and the many-row triangular solve on their own) are not part
   of the drop-in surface: include/oak_hip_bench.h. */

#ifdef __cplusplus
}
#endif
#endif /* OAK_HIP_H */

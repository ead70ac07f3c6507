// Internal declarations shared by the HIP translation units of liboak_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include "oak_hip.h"
#include "oak_hip_bench.h"

namespace oak {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define OAK_HIP_CHECK(expr)                                                                    \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            oak::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                           __LINE__);                                                          \
            return OAK_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)
#define OAK_CHECK(expr)                                                                        \
    do {                                                                                       \
        int _s = (expr);                                                                       \
        if (_s != OAK_OK) return _s;                                                           \
    } while (0)
#define OAK_REQUIRE(cond, ...)                                                                 \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            oak::set_error(__VA_ARGS__);                                                       \
            return OAK_E_ARG;                                                                  \
        }                                                                                      \
    } while (0)

// ---- device-side kernel description (passed by value as a kernel argument) -------------------
struct DevDesc {
    int D;                         // number of sub-kernels
    int R;                         // max interaction depth
    double w[OAK_MAX_DEPTH + 1];   // weight of e_r in K: share_var ? sigma2_r : (r==0 ? sigma2_0 : 1)
    unsigned char type[OAK_MAX_DIMS];
    short col[OAK_MAX_DIMS];       // active column
    int ncat[OAK_MAX_DIMS];        // categories (2 for binary)
    int tab_off[OAK_MAX_DIMS];     // offset into the device table buffer: C*C table (variance applied) then C diag
    double scale[OAK_MAX_DIMS];    // sqrt(log2(e)/2)/lengthscale     (RBF)
    double log2bv[OAK_MAX_DIMS];   // log2(base variance)             (RBF)
    double bv[OAK_MAX_DIMS];       // base variance
    // pair-kernel exponent form (exp2w.h): log2(bv) = n - 1024 woff, n = max(ceil(log2 bv), 0)
    double woff[OAK_MAX_DIMS];     // (n - log2 bv) / 1024 >= 0
    double magic[OAK_MAX_DIMS];    // 1.5 * 2^32 + n / 1024 (EW_MAGIC + n/1024, exp2w.h)
    // Grouped sub-kernels (an unconstrained RBF over several columns, oak/oak_kernel.py:74-82,199-210): dim d reads nxc[d] further
    // columns; their scaled values are rows xrow[d] .. xrow[d] + nxc[d] - 1 of Feat::xx and their squared differences add to
    // the exponent of dim d's one exponential (a product of one-column RBFs with a shared lengthscale).
    short xrow[OAK_MAX_DIMS];
    unsigned char nxc[OAK_MAX_DIMS];
};

// measure parameters used only by the featurize kernels
struct DevMeasure {
    unsigned char kind[OAK_MAX_DIMS];
    int k[OAK_MAX_DIMS];
    int off[OAK_MAX_DIMS];         // offset into device meas buffer
    double p0[OAK_MAX_DIMS], p1[OAK_MAX_DIMS];   // continuous dims: measure parameters; binary dims: p0 and sqrt(base variance)
    double ls[OAK_MAX_DIMS];
    double inv_sqrt_v[OAK_MAX_DIMS];   // 1/sqrt(var_s)   (0 for unconstrained)
    double dlogv[OAK_MAX_DIMS];        // d log(var_s) / d lengthscale  (gradient path)
};

// Featurised point set, struct-of-arrays, dimension-major: xs[d*ld + i], cn[d*ld + i]
//   RBF dim:         xs = x * scale_d ;  cn = cov_X_s(x)/sqrt(var_s)
//   categorical dim: xs = category index (as double) ; cn = 0
//   binary dim:      xs = category index (as double) ; cn = a(x) * sqrt(bv), a = (1 - p0, -p0)[x]: the sub-kernel is the rank-one
//                    product cn_a * cn_b (ortho_binary_kernel.py:29-38) and the forward Gram kernel evaluates it as that, NOT as
//                    E - cn_a cn_b.  A kernel that walks every dim with the RBF form must branch on dd.type first (the backward,
//                    fp32, diag and generic kernels read the table for discrete dims and ignore cn there).
struct Feat {
    double* xs = nullptr;
    double* cn = nullptr;
    double* dcn = nullptr;     // d cn / d lengthscale_d (only when featurized for the backward pass)
    double* xs32 = nullptr;    // backward pass only: xs / 32 (RBF dims; category index otherwise) and dcn / 1024, the
    double* dcs = nullptr;     //   pre-scaled forms the fast backward kernel consumes (exp2w.h) without a multiply
    double* xx = nullptr;      // grouped sub-kernels: xx[q*ld + i] = x[i, extra column q] * scale of the owning dim (nx rows)
    int nx = 0;
    int64_t n = 0;
    int64_t ld = 0;
};

// the streams and events of one context (pooled per process, runtime.hip::acquire_streams)
struct StreamSet {
    int device = 0;
    hipStream_t main = nullptr, side = nullptr, main_part = nullptr, side_part = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    int part_cus = 0, part_cus_req = 0;
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct Timing { double ms = 0; int count = 0; };
struct PendingEvt { std::string name; hipEvent_t a, b; };   // a recorded phase whose elapsed time has not been read yet

struct PreparedKernel {
    DevDesc dd;
    DevMeasure dm;
    // Depth bookkeeping.  e_r of D sub-kernels vanishes identically for r > D, so the kernels run at depth min(R, D) = dd.R
    // (R_desc is what the caller described: the public gradient layout has R_desc + 1 order-variance slots).  Only the explicit
    // Gram entry points accept min(R, D) > OAK_MAX_DEPTH (deep = true: generic kernel, weights in w_full); dd.R is then unusable.
    int R_desc = 0;
    bool deep = false;
    std::vector<double> w_full;      // weights of e_0..e_{min(R, D)} (always filled)
    bool grouped = false;            // some sub-kernel reads more than one column (an unconstrained RBF over a group)
    std::vector<int> extra_off, extra_cols;   // columns beyond dd.col[d]: extra_cols[extra_off[d] .. extra_off[d + 1])
    std::vector<double> extra_scale;          // dd.scale of the dim that owns extra column q
    std::vector<double> tables;      // host copy of the discrete tables
    double* d_tables = nullptr;      // device (ctx scratch "tables")
    double* d_meas = nullptr;        // device (ctx scratch "meas")
    // host-side derivative helpers (per RBF dim): d inv_v / d lengthscale etc. are recomputed in grad code
};

// exact int8 / CRT accumulation (crt.hip): the moduli and the buffers of one panel chunk
constexpr int CRT_MAXL = 20;
struct CrtMod { int L; int p[CRT_MAXL]; double inv[CRT_MAXL]; };
struct CrtPlan {
    CrtMod md; int B = 48, nsplit = 0; int64_t Mp2 = 0, rows_pad = 0, rps = 0;
    int* d_sexp = nullptr; int8_t* d_planes = nullptr; int* d_part = nullptr; int* d_res = nullptr;
};

}  // namespace oak

struct oak_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;                  // side stream: Kuu factorisation overlapped with the N-sized stages
    hipEvent_t ev0 = nullptr, ev1 = nullptr;     // fork / join events for the side stream
    // Spatial partition for SMALL evaluations (sgpr_forward): `side_part` owns part_cus compute units (the same
    // ones in every XCD, hipExtStreamCreateWithCUMask), `main_part` the others.  The latency-bound chol(Kuu) chain then runs at
    // its stand-alone speed NEXT to the first Gram panel instead of in front of it.  `stream` / `side` point at the unmasked pair
    // (`main_full` / `side_full`) except inside a partitioned forward pass.
    hipStream_t main_full = nullptr, side_full = nullptr, main_part = nullptr, side_part = nullptr;
    hipEvent_t ev3 = nullptr;                    // main_part -> main_full hand-over
    int part_cus = 0;                            // compute units of side_part (0: no partition streams)
    int part_cus_req = 0;                        // what OAK_PART_CUS asked for when the streams were made (pool key)
    bool part_tried = false;                     // the CU-masked pair was asked for once (ensure_partition_streams)
    bool part_active = false;                    // inside a partitioned forward pass
    bool kuu_deferred = false;                   // ... whose side chain is enqueued by local_stats right behind its first Gram launch
    bool part_syrk_full = false;                 // ... whose SYRK runs on the whole chip once the side chain has finished
    double kuu_jitter = 0;                       // jitter of the deferred chain
    std::map<std::string, oak::DevBuf> bufs;     // named, grow-only device scratch
    std::map<std::string, oak::Timing> timings;
    std::vector<oak::PendingEvt> pending;       // harvested without blocking once it grows (PhaseTimer::stop), read by oak_last_timing
    // SGPR state
    int64_t N = 0, M = 0;
    int32_t ldx = 0;
    int64_t panel_rows = 0;
    bool have_data = false, have_Z = false, have_stats = false, have_post = false, stats_whitened = false, have_alpha = false;
    bool have_linv = false;          // buffers "Linv" / "LinvT" hold L^-1 and its transpose for the current L
    int route = 0;   // 0 auto, 1 phi, 2 whitened
    int gram_form = 0;               // explicit Gram entry points: 0 native arithmetic, 1 the reference's (oak_set_gram_form)
    int precision = -1;              // -1 (default): 2 where it pays (sgpr_local_stats), 0 elsewhere; 0: fp64 kernels throughout; 1: fp32 Kfu panel + fp32-MFMA Phi partials on the phi route (forward only);
                                     // 2: Phi accumulated exactly on the int8 matrix pipe (scaled 48-bit integers, CRT; phi route)
    int auto_whiten = -1;            // decision of the conditioning check for this evaluation (-1: none, use the size rule)
    bool auto_pending = false;       // the check's result (cond_mm) is still in flight on the side stream
    bool kuu_async = false;          // sgpr_forward started chol(Kuu + jitter I) and L^-1 on the side stream (join on ev1)
    bool cond_requested = false;     // this evaluation's side-stream factorisation also reports cond_mm (auto route / fp32 mode)
    bool cond_seen = false;          // ... and cond_mm holds it for the tail's report (oak_sgpr_last_terms slot 7)
    bool stats_fp32 = false;         // the statistics in "stats" came from the fp32 panel path
    bool stats_crt = false;          // ... Phi of the statistics in "stats" was accumulated exactly on the int8 pipe (crt.hip)
    bool last_tail_dd = false;       // the most recent tail whitened Phi in double-double arithmetic (oak_bench_crt_info slot 6)
    bool stats_phi_dd = false;       // ... and buffer "phi_lo" holds the low word of that Phi (cleared when the statistics are replaced / summed)
    // Under a communicator the shards' Phi are summed EXACTLY (comm.hip: two fixed-point limbs per entry on a grid every rank derives from the
    // same rank-independent bound, summed by the ordinary fp64 all-reduce), so that the double-double tail also serves multi-rank jobs.  Decided
    // per evaluation by a rule of rank-independent inputs only (sgpr.hip::comm_dd_rule): every rank then runs the same sequence of collectives.
    bool comm_dd = false;
    int64_t crt_info[6] = {0, 0, 0, 0, 0, 0};   // ... and how (oak_bench_crt_info)
    // The residue planes of the WHOLE Kfu panel are still in "crt_planes" (the forward pass of a gradient call converted all rows in one
    // chunk): the backward pass forms its adjoint panel from them on the int8 pipe (crt_gemm.hip).  Valid between that forward and the
    // backward of the same call only.
    bool crt_planes_valid = false; oak::CrtPlan crt_pl;
    bool grad_int8 = false;          // set by the gradient entry points around their forward pass: the backward will use those planes
    bool crt_panel_written = true;   // the int8-route forward pass also wrote the fp64 Kfu panel
    int64_t crt_gemm_info[2] = {0, 0};           // moduli and bits of H' of the most recent int8 adjoint GEMM (0: the fp64 GEMM ran)
    double cond_mm[2] = {1.0, 1.0};  // min / max of diag chol(Kuu), written by the side stream
    hipEvent_t ev2 = nullptr;        // side stream: conditioning estimate ready
    double noise_var = 0, jitter = 0;
    double last_terms[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // terms of the most recent tail (oak_sgpr_last_terms)
    // GPR state
    int64_t gN = 0; int32_t gldx = 0; bool g_have_data = false, g_have_post = false; double g_noise = 0;
    // communicator (RCCL, dlopen'ed)
    void* comm = nullptr; int nranks = 1, rank = 0;
    void* host_allreduce = nullptr; void* host_user = nullptr;   // host-exchange communicator (oak_comm_init_host)
    int64_t n_global_user = 0;       // rows over ALL shards as told by oak_sgpr_set_global_rows (0: not told)
    int64_t n_global_comm = 0;       // ... as summed over the communicator (0: not yet; reset by set_data / comm init / destroy)
    long long* potrf_trace = nullptr; int64_t potrf_trace_steps = 0;   // armed by oak_bench_potrf (OAK_POTRF_TRACE=1) around its own calls only
    double* h_pin = nullptr;         // pinned host scratch (64 doubles): where the tail's scalars land
    int num_cu = 256;
    int64_t flow_n = 0;              // length of the resident normalising-flow sample "flow_g"
    int syrk_desc_ntile = -1;        // ntile the device descriptor table "syrk_desc" was built for
    int syrk_desc_blocks_ntile = -1; // same for "syrk_desc_blocks" (the fp32 variant's table: full diagonal 64-blocks)
    // breadcrumbs for oak_debug_state (a watchdog on another thread reads them without locks: literals and integers only)
    const char* marks[16] = {nullptr}; double mark_t[16] = {0}; unsigned mark_n = 0;
    // a gradient entry point's forward pass featurizes X and Z WITH the backward pass's arrays (one pass over X instead of two);
    // valid only between that forward and the backward of the same call
    bool feat_grad_valid = false; oak::Feat featXg, featZg;
    bool keep_kfu = false;           // set by the gradient entry points: a whitening forward works on a COPY of the Kfu panel
    bool kfu_kept = false;           // ... and reports here that "panel" still holds the raw Kfu rows of the whole data set
    // Extra target columns (oak_sgpr_set_extra_targets): outputs 1 .. n_extra of a model with 1 + n_extra output columns that
    // share kernel, inducing points and noise.  Buffers: "Yx" [n_extra x N], "yyx" [n_extra] (y^T y of this rank's rows),
    // "psix" [n_extra x M | n_extra] (Kuf y per column, then the y^T y), "c_all" [(1 + n_extra) x M] (c of every output).
    int n_extra = 0;
    bool psix_valid = false;         // "psix" belongs to the statistics in "stats" (false once oak_sgpr_set_stats replaced those from outside)
    int psix_nwg = 0;                // row blocks whose partial sums "psix_part" holds (set by the first panel chunk)
    int out_sel = 0;                 // the output whose c sits in buffer "c" (alpha / predict): oak_sgpr_select_output
    int sobol_path = 0;              // 0 automatic (cost model), 1 one workgroup per term, 2 Gram of products (oak_sobol_set_path)
    double sobol_info[4] = {0, 0, 0, 0};   // last oak_sobol: path taken, Gram columns, largest order-4 pairing disagreement, pair rows
};

namespace oak {

// scratch management ---------------------------------------------------------------------------
int get_buf(oak_ctx* ctx, const char* name, size_t bytes, void** out);   // grow-only
template <typename T> inline int get_buf_t(oak_ctx* ctx, const char* name, size_t count, T** out) {
    void* p = nullptr; int s = get_buf(ctx, name, count * sizeof(T), &p); *out = (T*)p; return s;
}
void* peek_buf(oak_ctx* ctx, const char* name);

struct PhaseTimer {   // hipEvent timing of a phase on the ctx stream (accumulates into ctx->timings)
    oak_ctx* ctx; const char* name; hipEvent_t a, b; bool active;
    PhaseTimer(oak_ctx* c, const char* n);
    ~PhaseTimer();      // a phase abandoned on an error path releases its events
    void stop();
};
void reset_timings(oak_ctx* ctx);
void debug_mark(oak_ctx* ctx, const char* literal);      // breadcrumb: the last 16 are shown by oak_debug_state

// kernel description ----------------------------------------------------------------------------
// template depth a kernel of effective depth R is run with (zero weights above R): 0..8 exact, then 12, 16, 24, 32
inline int template_depth(int R) { return R <= 8 ? R : (R <= 12 ? 12 : (R <= 16 ? 16 : (R <= 24 ? 24 : 32))); }
// allow: what the caller's kernels can evaluate beyond the fused pair kernels' common ground
constexpr int PK_DEEP = 1;       // effective depth > OAK_MAX_DEPTH (the explicit Gram entry points' generic kernel)
constexpr int PK_GROUPED = 2;    // sub-kernels over several columns (gram / gram_diag / gram_bwd / gram_bwd_z / diag_bwd take them; the fp32
                                 // and Sobol kernels do not)
int prepare_kernel(oak_ctx* ctx, const oak_kernel_desc* desc, PreparedKernel* pk, int allow = 0);
bool ensure_partition_streams(oak_ctx* ctx);      // runtime.hip: the CU-masked stream pair, created on first use
// component (single subset) description derived from a full one
int prepare_component(oak_ctx* ctx, const oak_kernel_desc* desc, const int32_t* subset, int32_t len,
                      int32_t apply_order_var, PreparedKernel* pk);

// featurize -------------------------------------------------------------------------------------
// d_kdiag_sum != NULL: where the kernel form allows it, also write sum_i K_diag(x_i) there (*kdiag_done tells)
int featurize(oak_ctx* ctx, const PreparedKernel& pk, const double* dX, int64_t n, int32_t ldx,
              const char* bufname, Feat* out, bool with_grad = false, double* d_kdiag_sum = nullptr, bool* kdiag_done = nullptr);

// gram ------------------------------------------------------------------------------------------
// out[i*ldo + j] = K(A_i, B_j).  If yA != nullptr also accumulates psi[j] += sum_i K(i,j) y_i into d_psi
// (d_psi must be zero-initialised by the caller or accumulate onto existing contents).
// zero_pad_to: columns [B.n, zero_pad_to) of each row are written as 0 (panel padding for syrk).
int gram(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B,
         double* d_out, int64_t ldo, const double* d_yA, double* d_psi, int64_t zero_pad_to);
int gram_diag(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, double* d_out, double* d_sum_accum);
// Generic Gram (one thread per entry, any depth <= OAK_MAX_DIMS): the fallback of the explicit Gram entry points beyond depth
// 16 and the A/B path that reproduces the REFERENCE's arithmetic on the device (form 1: GPflow's expanded squared distance
// from x / l, exp, power sums + Newton-Girard, oak/oak_kernel.py:236-249; form 0: this library's own, direct distance and the
// exact-sum recurrence).  dXa / dXb: the raw inputs (row-major, ldx); B = A when dXb is NULL.  diag: K_diag of A into out[na].
int gram_generic(oak_ctx* ctx, const PreparedKernel& pk, int form, const double* dXa, const Feat& A, int64_t na, const double* dXb,
                 const Feat& B, int64_t nb, int32_t ldx, double* d_out, int64_t ldo, bool diag);

// dense linear algebra on the device (fp64) -------------------------------------------------------
int syrk_panel(oak_ctx* ctx, const double* d_panel, int64_t ldp, int64_t nrows, int64_t M, double* d_part,
               int nsplit, bool accumulate);
int syrk_plan_splits(oak_ctx* ctx, int64_t M, int64_t nrows);
int syrk_panel_two(oak_ctx* ctx, const double* d_panel, int64_t ldp, int64_t rowsA, int64_t rowsB, int64_t M, double* d_part, int nsA, int nsB,
                   int64_t rows_per_split);
int syrk_descriptor_table(oak_ctx* ctx, int ntile, int** d_desc_out, int* npairs_out, bool diag_pairs);
// fp32 statistics variant (gram32.hip): fp32 Kfu panel and fp32-MFMA partials, everything downstream fp64
int gram_f32(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, float* d_out, int64_t ldo,
             const double* d_yA, double* d_psi, int64_t zero_pad_to);
int syrk_panel_f32(oak_ctx* ctx, const float* d_panel, int64_t ldp, int64_t nrows, int64_t M, double* d_part, int nsplit, bool accumulate);
int syrk_reduce(oak_ctx* ctx, const double* d_part, int nsplit, int64_t M, double* d_phi /*[M*M]*/, bool accumulate);
// exact int8 / CRT accumulation of Phi (crt.hip; the fused Gram epilogue is gram.hip::gram_crt_kernel)
bool crt_supported(const oak_ctx* ctx, int64_t M);
int crt_plan(oak_ctx* ctx, int64_t na, int64_t M, int64_t n_total, CrtPlan* pl);
// kdiag_parts: the featurize pass of X ran its tiled form with the K_diag reduction (kappa_done): its per-workgroup maxima serve
int crt_scales(oak_ctx* ctx, const PreparedKernel& pk, const Feat& FX, const Feat& FZ, int64_t M, const CrtPlan& pl, bool kdiag_parts);
int crt_convert_panel(oak_ctx* ctx, const CrtPlan& pl, const double* d_panel, int64_t ldp, int64_t na);
// d_phi_lo (may be NULL): the low word of the double-double Phi (the integer Gram matrix holds ~118 bits)
int crt_accumulate(oak_ctx* ctx, const CrtPlan& pl, int64_t M, bool first_chunk, bool last_chunk, double* d_phi, double* d_phi_lo);
// exponents e_m with bound_m < 2^e_m for |K(x, z_m)| over ANY x (a-priori: rank-independent), into the int buffer d_eexp[M] (crt.hip)
bool comm_dd_rule(const oak_ctx* ctx, int64_t M);      // sgpr.hip
bool sgpr_cond_estimate_ok_for_int8_gemm(oak_ctx* ctx);  // sgpr.hip
int crt_bound_exponents(oak_ctx* ctx, const PreparedKernel& pk, const Feat& FZ, int64_t M, int* d_eexp);
// exact exchange of Phi between ranks (ddgemm.hip): split the (double-double) Phi in place into the high limb (the Phi slot itself) and the
// low limb d_lo on the grid 2^(e_a + e_b + en - 51); after the all-reduce of both, join them into Phi (one double) and its low word
int dd_exchange_split(oak_ctx* ctx, double* d_phi, const double* d_phi_lo, const int* d_eexp, int en, int lo_bits, int64_t M, double* d_lo);
int dd_exchange_join(oak_ctx* ctx, double* d_phi, const double* d_lo, int64_t M, double* d_phi_lo);
// adjoint panel G = Kfu H on the int8 pipe from the planes of `pl` (crt_gemm.hip)
int crt_gemm_adjoint(oak_ctx* ctx, const CrtPlan& pl, int64_t M, int64_t na, const double* d_H, double* d_G, int64_t ldg);
// [W ; (L^-1 psi)^T] = [L^-1 Phi L^-T ; (L^-1 psi)^T] in double-double arithmetic from the double-double Phi (ddgemm.hip), one double out
int dd_whiten(oak_ctx* ctx, const double* d_Linv, const double* d_phi_hi, const double* d_phi_lo, const double* d_psi, int64_t M, double* d_out,
              const double* d_psix, int nx);
bool gram_crt_supported(const PreparedKernel& pk);
int gram_crt(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, double* d_out, int64_t ldo,
             const double* d_yA, double* d_psi, int64_t zero_pad_to, const CrtMod& md, const int* d_sexp, int8_t* d_planes, int64_t rows_pad,
             int64_t Mp2);
// in place; strict upper zeroed.  nrows > n carries nrows - n extra rows through the panel solves and trailing updates
// (row r >= n ends up as  A[r, :n] L^-T,  i.e. the solution of L x = A[r, :n]^T: a right-hand side rides for free).
int potrf_lower(oak_ctx* ctx, double* dA, int64_t n, int64_t lda, bool check = true, int64_t nrows = -1, bool identity_below = false);
int set_identity(oak_ctx* ctx, double* dA, int64_t n);
int potrf_check(oak_ctx* ctx, int slot, int64_t n);
int potrf_check_value(int info, int64_t n);             // the same test on a status word the caller fetched itself   // deferred status of a check=false factorisation (slot 1 = side stream)
// rows-trsm: each of the nrhs rows of BT (row stride ldb) is a right-hand side; solves L x = b (trans=0)
// or L^T x = b (trans=1) in place.
int trsm_rows(oak_ctx* ctx, const double* dL, int64_t n, int64_t ldl, double* dBT, int64_t nrhs, int64_t ldb, int trans);
// The same forward solve for MANY rows as one launch (trsm_fused.hip): out of place allowed (dBin != dXout keeps the right-hand
// sides), n is padded to a multiple of 128 with the identity, so both panels must have pad128(n) columns (pad columns are
// copied).  The inverses of the 128 x 128 diagonal blocks come either from a full inverse dLinv (row-major, leading
// dimension ldinv: its diagonal blocks are those inverses) or from dInvBlocks = compact [n/128][128][128] block inverses.
int trsm_rows_fused(oak_ctx* ctx, const double* dL, int64_t n, int64_t ldl, const double* dLinv, int64_t ldinv,
                    const double* dInvBlocks, const double* dBin, int64_t ldin, double* dXout, int64_t ldout, int64_t nrhs,
                    bool transposed = false);
int transpose(oak_ctx* ctx, const double* dA, int64_t rows, int64_t cols, int64_t lda, double* dB, int64_t ldb);
int add_diag(oak_ctx* ctx, double* dA, int64_t n, int64_t lda, double v);
int scale_add_eye(oak_ctx* ctx, const double* dW, int64_t n, double s, double* dB, int extra_rows = 0);   // B = I + s*W (+ rows copied as they are)
int scaled_copy(oak_ctx* ctx, double a, const double* d_src, double* d_dst, int64_t n);               // dst = a * src
int reduce_sum(oak_ctx* ctx, const double* d_x, int64_t n, double* d_out /*1*/, int mode /*0 sum,1 sumsq,2 sumlog*/, int64_t stride);
int dot(oak_ctx* ctx, const double* d_x, const double* d_y, int64_t n, double* d_out);
int trsv_lower_blockinv(oak_ctx* ctx, const double* dL, int64_t n, int64_t ldl, const double* dLinv, int64_t ldinv, double* d_b);   // factor.hip
int gemv_rows(oak_ctx* ctx, const double* dA, int64_t rows, int64_t cols, int64_t lda, const double* d_x, double* d_y); // y = A x
int row_sumsq(oak_ctx* ctx, const double* dA, int64_t rows, int64_t cols, int64_t lda, double* d_out);                  // out_i = sum_j A_ij^2
int copy_d2d(oak_ctx* ctx, void* dst, const void* src, size_t bytes);
// host <-> device copy ON THE CONTEXT'S STREAM followed by a wait on that stream only.  The library never uses the legacy
// NULL stream: its implicit synchronisation with every blocking stream of the process would couple independent contexts
// driven from different host threads.
int copy_sync(oak_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
// raises the dynamic-LDS limit of a kernel to the hardware's 160 KiB, once per kernel and process (the attribute belongs to
// the function, not to the launch: setting it to each launch's own size would race between contexts on different threads)
int ensure_max_dynamic_lds(const void* kernel);
int ensure_dynamic_lds(const void* kernel, size_t bytes);      // for kernels that also hold static LDS: raises the DYNAMIC limit to `bytes`
int fill_zero(oak_ctx* ctx, void* dst, size_t bytes);
int axpy(oak_ctx* ctx, double a, const double* x, double* y, int64_t n);  // y += a x
int scale_vec(oak_ctx* ctx, double a, double* x, int64_t n);
int gemm_nn(oak_ctx* ctx, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k,
            int64_t lda, int64_t ldb, int64_t ldc, double alpha, double beta);   // C = alpha A B + beta C (row-major)
int gemm_nt(oak_ctx* ctx, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k,
            int64_t lda, int64_t ldb, int64_t ldc, double alpha, double beta, int lower_only);   // C = alpha A B^T + beta C
bool gemm_nt_rankp(oak_ctx* ctx, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb,
                   int64_t ldc, const double* dYx, int64_t ldy, const double* d_ax, int nx, int* status);   // factor.hip
// M x M products of the O(M^3) tail: split-k over gridDim.z + fixed-order reduce, and triangular operands declared so that a
// tile only walks the k range where both are non-zero.  bt = 1: C = alpha A B^T + beta C (B indexed [n][k]); bt = 0: A B.
constexpr int OAK_TRI_A_LOWER = 1, OAK_TRI_A_UPPER = 2, OAK_TRI_B_LOWER = 4, OAK_TRI_B_UPPER = 8;
int gemm_tail(oak_ctx* ctx, int bt, const double* dA, const double* dB, double* dC, int64_t m, int64_t n, int64_t k, int64_t lda,
              int64_t ldb, int64_t ldc, double alpha, double beta, int tri);

// SGPR pipeline pieces shared between translation units
int sgpr_local_stats(oak_ctx* ctx, const PreparedKernel& pk, double jitter);
// l_state: 0 = the tail builds L = chol(Kuu + jitter I) itself; 1 = buffer "L" already holds it on the main stream
// (whitened route); 2 = it is being factored on the side stream (sgpr_factor_kuu_async) and the tail joins on ev1.
int sgpr_tail(oak_ctx* ctx, const PreparedKernel& pk, double noise_var, double jitter, double* elbo_out, double* terms_out,
              int l_state = 0);
bool sgpr_route_whitened(const oak_ctx* ctx);
int sgpr_factor_kuu_async(oak_ctx* ctx, const PreparedKernel& pk, double jitter, double* cond_out = nullptr, bool fork = true, int phase = 0);
int sgpr_forward(oak_ctx* ctx, const PreparedKernel& pk, double noise_var, double jitter, double* elbo_out, double* terms_out);
int sgpr_ensure_alpha(oak_ctx* ctx);
// psix[p][m] (+)= sum_r panel[r][m] * Yx[p][a0 + r] for the extra target columns, one pass over a raw Kfu panel chunk
int sgpr_extra_psi(oak_ctx* ctx, const double* d_panel, int64_t ldp, int64_t a0, int64_t na, bool first_chunk);
int sgpr_extra_psi_finish(oak_ctx* ctx);

// backward pieces shared with the SVGP path (grad.hip) ------------------------------------------------
int64_t record_len(const PreparedKernel& pk);      // [d/d lengthscale' (D) | d/d log base_var (D) | d/d w (R+1) | d/d tables]
// d_rec += contraction of the adjoint block G (na x nb, + optional rank-1 yA avec^T), scaled by g_scale, with dK/dtheta
int gram_bwd(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, int64_t a0, int64_t na, const Feat& B, const double* d_G,
             int64_t ldg, double g_scale, const double* d_yA, const double* d_avec, double* d_rec, bool want_gk = true);
// d_rec += sum_n gconst * (d_gvec ? d_gvec[n] : 1) * dKdiag_n/dtheta
int diag_bwd(oak_ctx* ctx, const PreparedKernel& pk, const Feat& A, double gconst, double* d_rec, const double* d_gvec = nullptr);
// rows-in-lanes form of the backward pair kernel (grad_rows.hip): all-continuous, unit base variances, depth <= 4, <= 16 sub-kernels
int gram_bwd_rows_launch(oak_ctx* ctx, const PreparedKernel& pk, int dmax, const double* d_apack, int64_t a0, int64_t na, const double* d_bpack,
                         int64_t nb, const double* d_G, int64_t ldg, const double* d_yA, const double* d_avec, double g_scale, int cols_per_wg,
                         double* d_part, int64_t* nrec_out);
void scatter_record(const oak_kernel_desc* desc, const PreparedKernel& pk, const std::vector<double>& rec, double dnoise,
                    double* grad_out);

// collectives --------------------------------------------------------------------------------------
int comm_allreduce_dev(oak_ctx* ctx, double* d_buf, int64_t n, const char* stage = "comm_stage");
// sum over ranks of one host scalar, on the SIDE stream (so it can run underneath main-stream kernels); synchronises the side stream
int comm_allreduce_scalar_side(oak_ctx* ctx, double* value);
bool comm_is_loopback(const oak_ctx* ctx);
int64_t sgpr_route_rows(const oak_ctx* ctx);     // rows the auto route is decided on: global when known, else local

}  // namespace oak

// Multi-GPU exchange: one RCCL reduce-scatter + all-gather (= all-reduce) of the packed SGPR statistics over xGMI.
// The reference has no distributed code; every N-dependent term of the ELBO is a sum over rows, so
// each rank reduces its own row shard to [Phi | psi | kappa | yy | n | n_whitened | n_parts] and only that M^2+M+5 vector is exchanged.
// librccl.so is dlopen'ed on first use so single-GPU runs carry no RCCL dependency.
#include "oak_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <mutex>
#include <cstdlib>
#include <string>
#include <vector>

namespace oak {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static Rccl g_rccl;

static std::string g_rccl_path;      // where the loaded librccl.so lives (dladdr on one of its symbols)
static int g_rccl_version = 0;       // ncclGetVersion of the loaded library

static int load_rccl() {
    static std::mutex mu;                    // contexts on different host threads may race to the first use
    std::lock_guard<std::mutex> lock(mu);
    if (g_rccl.h) return OAK_OK;
    // The header this file compiles against is ROCm's (<rccl/rccl.h> under $ROCM_PATH/include): prefer the library that
    // belongs to it over whatever librccl.so the loader would find first (PyTorch wheels ship their own copy).
    std::vector<std::string> names;
    if (const char* rp = getenv("ROCM_PATH")) names.push_back(std::string(rp) + "/lib/librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so");
    names.push_back("librccl.so");
    names.push_back("librccl.so.1");
    void* h = nullptr;
    for (const auto& n : names) { h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { set_error("cannot dlopen librccl.so: %s", dlerror()); return OAK_E_NCCL; }
#define OAK_SYM(field, sym)                                                                  \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, sym);                                    \
    if (!g_rccl.field) { set_error("librccl.so lacks symbol %s", sym); return OAK_E_NCCL; }
    OAK_SYM(GetUniqueId, "ncclGetUniqueId")
    OAK_SYM(CommInitRank, "ncclCommInitRank")
    OAK_SYM(CommDestroy, "ncclCommDestroy")
    OAK_SYM(ReduceScatter, "ncclReduceScatter")
    OAK_SYM(AllGather, "ncclAllGather")
    OAK_SYM(AllReduce, "ncclAllReduce")
    OAK_SYM(Broadcast, "ncclBroadcast")
    OAK_SYM(GroupStart, "ncclGroupStart")
    OAK_SYM(GroupEnd, "ncclGroupEnd")
    OAK_SYM(GetErrorString, "ncclGetErrorString")
#undef OAK_SYM
    auto get_version = (ncclResult_t(*)(int*))dlsym(h, "ncclGetVersion");
    int v = 0;
    if (!get_version || get_version(&v) != ncclSuccess) { set_error("librccl.so: ncclGetVersion unavailable"); return OAK_E_NCCL; }
    // ncclGetVersion encodes major * 10000 + minor * 100 + patch (major * 1000 + ... before 2.9); the ABI of the calls used
    // here is stable within a major version
    const int major_loaded = v >= 10000 ? v / 10000 : v / 1000;
    if (major_loaded != NCCL_MAJOR) {
        set_error("librccl.so reports version %d (major %d) but liboak_hip was compiled against rccl.h %d.%d.%d", v, major_loaded,
                  NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH);
        return OAK_E_NCCL;
    }
    Dl_info info;
    if (dladdr((void*)g_rccl.CommInitRank, &info) && info.dli_fname) g_rccl_path = info.dli_fname;
    g_rccl_version = v;
    g_rccl.h = h;
    return OAK_OK;
}

#define OAK_NCCL_CHECK(expr)                                                                  \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != ncclSuccess) {                                                              \
            set_error("%s failed: %s", #expr, g_rccl.GetErrorString(_r));                     \
            return OAK_E_NCCL;                                                                \
        }                                                                                     \
    } while (0)

// In-place sum over ranks of d_buf[0..n): reduce-scatter of equal slices, then all-gather.  The scratch tail that
// pads n up to a multiple of nranks lives in a separate zeroed staging buffer so d_buf need not be over-allocated.
// Test communicator (oak_comm_init_loopback): every "rank" is assumed to hold the same local buffer, so the sum over ranks
// is nranks * local.  It runs the complete N > 1 code path (scaling of the replicated terms, placement of the reductions)
// on ONE GPU: a loopback run on data X must equal a single-rank run on X stacked nranks times.
static char g_loopback_tag, g_host_tag;
static inline bool is_loopback(const oak_ctx* ctx) { return ctx->comm == (void*)&g_loopback_tag; }
// Host-exchange communicator (oak_comm_init_host): the sum over ranks is delegated to a callback of the host language on a
// host copy of the buffer (a socket / MPI / gloo control plane).  Every collective of the library -- statistics, gradient
// records, route decisions -- goes through comm_allreduce_dev, so the whole N > 1 path runs unchanged with several ranks
// sharing one GPU or with no xGMI fabric at all; it is also what the model API falls back to when RCCL cannot be used.
static inline bool is_host(const oak_ctx* ctx) { return ctx->comm == (void*)&g_host_tag; }

bool comm_is_loopback(const oak_ctx* ctx) { return is_loopback(ctx); }

int comm_allreduce_dev(oak_ctx* ctx, double* d_buf, int64_t n, const char* stage) {
    if (ctx->comm == nullptr || ctx->nranks <= 1 || n <= 0) return OAK_OK;
    debug_mark(ctx, "allreduce>");
    struct Leave { oak_ctx* c; ~Leave() { debug_mark(c, "allreduce<"); } } leave{ctx};
    if (is_host(ctx)) {
        std::vector<double> h((size_t)n);
        OAK_HIP_CHECK(hipMemcpyAsync(h.data(), d_buf, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const int rc = ((oak_host_allreduce_fn)ctx->host_allreduce)(h.data(), n, ctx->host_user);
        if (rc != 0) { set_error("host all-reduce callback failed with status %d", rc); return OAK_E_NCCL; }
        OAK_HIP_CHECK(hipMemcpyAsync(d_buf, h.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));      // h goes out of scope
        return OAK_OK;
    }
    const int64_t P = ctx->nranks;
    const int64_t slice = (n + P - 1) / P;
    double* d_stage = nullptr;
    OAK_CHECK(get_buf_t(ctx, stage, (size_t)(slice * P), &d_stage));
    if (slice * P > n) OAK_CHECK(fill_zero(ctx, d_stage + n, sizeof(double) * (size_t)(slice * P - n)));
    OAK_CHECK(copy_d2d(ctx, d_stage, d_buf, sizeof(double) * (size_t)n));
    if (is_loopback(ctx)) {
        // the same slicing as the real exchange: slice r is what rank r's reduce-scatter would hold (every rank has the same
        // local vector, so the sum is nranks * local), and the all-gather is the identity on the staging buffer
        for (int64_t r = 0; r < P; ++r) OAK_CHECK(scale_vec(ctx, (double)P, d_stage + r * slice, slice));
    } else {
        OAK_CHECK(load_rccl());
        ncclComm_t comm = (ncclComm_t)ctx->comm;
        OAK_NCCL_CHECK(g_rccl.ReduceScatter(d_stage, d_stage + ctx->rank * slice, (size_t)slice, ncclFloat64, ncclSum, comm, ctx->stream));
        OAK_NCCL_CHECK(g_rccl.AllGather(d_stage + ctx->rank * slice, d_stage, (size_t)slice, ncclFloat64, comm, ctx->stream));
    }
    OAK_CHECK(copy_d2d(ctx, d_buf, d_stage, sizeof(double) * (size_t)n));
    return OAK_OK;
}

__global__ void set_scalar_kernel(double* p, double v) { *p = v; }

// Control-plane collective: every rank contributes one scalar, every rank gets the sum.  Runs on the side stream with
// its own staging buffer so that it neither waits for nor disturbs what the main stream has queued.
int comm_allreduce_scalar_side(oak_ctx* ctx, double* value) {
    if (ctx->comm == nullptr || ctx->nranks <= 1) return OAK_OK;
    double* d = nullptr;
    OAK_CHECK(get_buf_t(ctx, "comm_ctl", 1, &d));
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ctx->side;
    if (ctx->part_active && ctx->side == ctx->side_part && ctx->side_full != nullptr && !comm_is_loopback(ctx) && ctx->host_allreduce == nullptr) {
        // partitioned pass under a real RCCL communicator: the collective's kernels go to the UNMASKED side stream (a communicator
        // serves one stream pair; RCCL kernels on a CU-masked queue have never run on hardware), ordered behind what side_part holds
        OAK_HIP_CHECK(hipEventRecord(ctx->ev3, ctx->side_part));
        OAK_HIP_CHECK(hipStreamWaitEvent(ctx->side_full, ctx->ev3, 0));
        ctx->stream = ctx->side_full;
    }
    int rc = [&]() -> int {
        set_scalar_kernel<<<1, 1, 0, ctx->stream>>>(d, *value);
        OAK_HIP_CHECK(hipGetLastError());
        OAK_CHECK(comm_allreduce_dev(ctx, d, 1, "comm_ctl_stage"));
        OAK_HIP_CHECK(hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return OAK_OK;
    }();
    ctx->stream = main_stream;
    return rc;
}

}  // namespace oak

using namespace oak;

extern "C" {

int oak_comm_unique_id(char* id_out_128) {
    if (!id_out_128) { set_error("id_out is NULL"); return OAK_E_ARG; }
    OAK_CHECK(load_rccl());
    ncclUniqueId id;
    OAK_NCCL_CHECK(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id_out_128, &id, 128);
    return OAK_OK;
}

int oak_comm_init(oak_ctx* ctx, const char* id_128, int32_t nranks, int32_t rank) {
    if (!ctx || !id_128) { set_error("bad argument"); return OAK_E_ARG; }
    OAK_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d / nranks %d invalid", rank, nranks);
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_CHECK(load_rccl());
    if (ctx->comm) oak_comm_destroy(ctx);
    ncclUniqueId id;
    memcpy(&id, id_128, 128);
    ncclComm_t comm = nullptr;
    debug_mark(ctx, "commInit>");
    OAK_NCCL_CHECK(g_rccl.CommInitRank(&comm, nranks, id, rank));
    debug_mark(ctx, "commInit<");
    ctx->comm = (void*)comm; ctx->nranks = nranks; ctx->rank = rank; ctx->n_global_comm = 0;
    return OAK_OK;
}

int oak_comm_init_loopback(oak_ctx* ctx, int32_t nranks) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_REQUIRE(nranks >= 1, "nranks %d invalid", nranks);
    if (ctx->comm) oak_comm_destroy(ctx);
    ctx->comm = (void*)&g_loopback_tag; ctx->nranks = nranks; ctx->rank = 0; ctx->n_global_comm = 0;
    return OAK_OK;
}

int oak_comm_init_host(oak_ctx* ctx, int32_t nranks, int32_t rank, oak_host_allreduce_fn fn, void* user) {
    if (!ctx || !fn) { set_error("bad argument"); return OAK_E_ARG; }
    OAK_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d / nranks %d invalid", rank, nranks);
    if (ctx->comm) oak_comm_destroy(ctx);
    ctx->comm = (void*)&g_host_tag; ctx->nranks = nranks; ctx->rank = rank; ctx->n_global_comm = 0;
    ctx->host_allreduce = (void*)fn; ctx->host_user = user;
    return OAK_OK;
}

int oak_comm_info(char* path_out, int64_t cap, int32_t* version_out, int32_t* header_version_out) {
    OAK_CHECK(load_rccl());
    if (path_out && cap > 0) { strncpy(path_out, g_rccl_path.c_str(), (size_t)cap - 1); path_out[cap - 1] = 0; }
    if (version_out) *version_out = g_rccl_version;
    if (header_version_out) *header_version_out = NCCL_VERSION_CODE;
    return OAK_OK;
}

int oak_comm_destroy(oak_ctx* ctx) {
    if (!ctx || !ctx->comm) return OAK_OK;
    if (!is_loopback(ctx) && !is_host(ctx) && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->host_allreduce = nullptr; ctx->host_user = nullptr;
    ctx->comm = nullptr; ctx->nranks = 1; ctx->rank = 0; ctx->n_global_comm = 0;
    return OAK_OK;
}

int oak_comm_allreduce_stats(oak_ctx* ctx) {
    if (!ctx) { set_error("ctx is NULL"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    OAK_REQUIRE(ctx->have_stats, "no local statistics to reduce");
    if (ctx->comm == nullptr || ctx->nranks <= 1) return OAK_OK;
    double* d_stats = (double*)peek_buf(ctx, "stats");
    // rank-local validation BEFORE the first collective: a rank that fails it must join neither of the two (a rank that left after
    // the first would leave its peers blocked in the second)
    double* d_psix = ctx->n_extra > 0 ? (double*)peek_buf(ctx, "psix") : nullptr;
    if (ctx->n_extra > 0)
        OAK_REQUIRE(d_psix != nullptr && ctx->psix_valid, "oak_comm_allreduce_stats: the extra target columns' statistics are not those of the "
                    "packed statistics in place (form both with oak_sgpr_local_stats on every rank)");
    PhaseTimer t(ctx, "allreduce");
    // Exact sum of the shards' Phi (sgpr.hip::comm_dd_rule; ddgemm.hip): the Phi slot carries the high limbs through the ordinary exchange,
    // a second vector the low limbs; joined afterwards into Phi and its low word.  Without it: a sum of shards in fp64 rounds Phi, its low
    // word no longer belongs to it.
    const int64_t M = ctx->M;
    const bool dd = ctx->comm_dd && !ctx->stats_whitened && peek_buf(ctx, "dd_eexp") != nullptr;
    double *d_lo = nullptr, *d_phi_lo = nullptr;
    if (dd) {
        OAK_CHECK(get_buf_t(ctx, "dd_lo_limb", (size_t)M * M, &d_lo));
        const double* lo_in = ctx->stats_phi_dd ? (const double*)peek_buf(ctx, "phi_lo") : nullptr;
        int en = 0, lr = 0;
        while (((int64_t)1 << en) < ctx->n_global_user) ++en;
        while ((1 << lr) < ctx->nranks) ++lr;
        OAK_CHECK(dd_exchange_split(ctx, d_stats, lo_in, (const int*)peek_buf(ctx, "dd_eexp"), en + 1, 50 - lr, M, d_lo));
        OAK_CHECK(get_buf_t(ctx, "phi_lo", (size_t)M * M, &d_phi_lo));
    }
    ctx->stats_phi_dd = false;
    // the two trailing slots (shards that whitened, shards summed) ride along: the tail rejects a mixed sum
    OAK_CHECK(comm_allreduce_dev(ctx, d_stats, oak_sgpr_stats_len(ctx)));
    if (dd) {
        OAK_CHECK(comm_allreduce_dev(ctx, d_lo, M * M, "comm_stage_lo"));
        OAK_CHECK(dd_exchange_join(ctx, d_stats, d_lo, M, d_phi_lo));
        ctx->stats_phi_dd = true;
    }
    // extra target columns: [Kuf y_p | y_p^T y_p] is part of the same sum over the row shards -- reduced HERE, so that the documented
    // local_stats -> allreduce_stats -> tail sequence and the fused entry points exchange the same things
    if (ctx->n_extra > 0) OAK_CHECK(comm_allreduce_dev(ctx, d_psix, (int64_t)ctx->n_extra * ctx->M + ctx->n_extra, "comm_stage_x"));
    t.stop();
    return OAK_OK;
}

int oak_comm_allreduce_host(oak_ctx* ctx, double* buf, int64_t n) {
    if (!ctx || !buf || n < 0) { set_error("bad argument"); return OAK_E_ARG; }
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->comm == nullptr || ctx->nranks <= 1 || n == 0) return OAK_OK;
    double* d = nullptr;
    OAK_CHECK(get_buf_t(ctx, "comm_host", (size_t)n, &d));
    OAK_HIP_CHECK(hipMemcpyAsync(d, buf, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    OAK_CHECK(comm_allreduce_dev(ctx, d, n));
    OAK_HIP_CHECK(hipMemcpyAsync(buf, d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OAK_OK;
}

// All-gather of variable-sized host blocks: buf has sum(counts) doubles, rank r's block (counts[r] doubles) sits at offset
// sum(counts[:r]); this rank's block is read, every other block is overwritten with its owner's.
//   RCCL communicator:  a true all-gather on the device -- one grouped ncclBroadcast per non-empty block, each rank sending
//                       only its own block over xGMI (total bytes on the wire = the gathered vector, not nranks times it);
//   host communicator:  the sum of zero-padded copies through the callback (every block is owned by exactly one rank);
//   loopback:           every "rank" holds the same local block (the test communicator's premise), so the blocks must all have
//                       this rank's length and the result is that block repeated.
int oak_comm_allgatherv(oak_ctx* ctx, double* buf, const int64_t* counts, int32_t n_counts) {
    if (!ctx || !counts || n_counts < 1) { set_error("bad argument"); return OAK_E_ARG; }
    int64_t total = 0;
    for (int r = 0; r < n_counts; ++r) {
        OAK_REQUIRE(counts[r] >= 0, "oak_comm_allgatherv: negative block length");
        total += counts[r];
    }
    OAK_REQUIRE(buf || total == 0, "oak_comm_allgatherv: buf is NULL");
    const int nranks = (ctx->comm != nullptr) ? ctx->nranks : 1;
    if (n_counts != nranks) {
        // blocks were announced for ranks this context knows nothing about: returning would hand back a buffer that looks
        // gathered and holds zeros for them
        set_error("oak_comm_allgatherv: %d blocks announced but the context's communicator has %d rank(s)", n_counts, nranks);
        return OAK_E_STATE;
    }
    if (ctx->comm == nullptr || total == 0) return OAK_OK;
    if (nranks <= 1 && (is_loopback(ctx) || is_host(ctx))) return OAK_OK;
    // (a ONE-rank RCCL communicator still goes through the grouped broadcast below: the identity, but it is the only execution of
    // that call sequence a one-GPU box can give -- tests/test_gpu_distributed.py::test_single_rank_rccl_exchange_is_identity)
    OAK_HIP_CHECK(hipSetDevice(ctx->device));
    int64_t offset = 0;
    for (int r = 0; r < ctx->rank; ++r) offset += counts[r];
    const int64_t count = counts[ctx->rank];
    if (is_loopback(ctx)) {
        for (int r = 0; r < nranks; ++r) OAK_REQUIRE(counts[r] == count, "oak_comm_allgatherv (loopback): every block must have the local length");
        int64_t o = 0;
        for (int r = 0; r < nranks; ++r, o += count)
            if (o != offset) memcpy(buf + o, buf + offset, sizeof(double) * (size_t)count);
        return OAK_OK;
    }
    if (is_host(ctx)) {
        for (int64_t i = 0; i < offset; ++i) buf[i] = 0.0;
        for (int64_t i = offset + count; i < total; ++i) buf[i] = 0.0;
        return oak_comm_allreduce_host(ctx, buf, total);
    }
    OAK_CHECK(load_rccl());
    double* d = nullptr;
    OAK_CHECK(get_buf_t(ctx, "comm_gather", (size_t)total, &d));
    if (count > 0) OAK_HIP_CHECK(hipMemcpyAsync(d + offset, buf + offset, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    OAK_NCCL_CHECK(g_rccl.GroupStart());
    int64_t o = 0;
    for (int r = 0; r < nranks; ++r) {
        if (counts[r] > 0) {
            const ncclResult_t rc = g_rccl.Broadcast(d + o, d + o, (size_t)counts[r], ncclFloat64, r, comm, ctx->stream);
            if (rc != ncclSuccess) { (void)g_rccl.GroupEnd(); set_error("ncclBroadcast failed: %s", g_rccl.GetErrorString(rc)); return OAK_E_NCCL; }
        }
        o += counts[r];
    }
    OAK_NCCL_CHECK(g_rccl.GroupEnd());
    OAK_HIP_CHECK(hipMemcpyAsync(buf, d, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    OAK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return OAK_OK;
}

}  // extern "C"

// Temporary: entry points still to be implemented (kept so the library links and exports every symbol).
#include "oak_internal.h"
extern "C" {
int64_t oak_grad_len(const oak_kernel_desc* desc) { return desc ? 2 * desc->num_dims + desc->n_order_var + 1 + desc->meas_data_len : 0; }
int oak_sgpr_elbo_grad(oak_ctx*, const oak_kernel_desc*, double, double, double*, double*) { oak::set_error("oak_sgpr_elbo_grad: not implemented yet"); return OAK_E_ARG; }
int oak_gpr_log_marginal_grad(oak_ctx*, const oak_kernel_desc*, double, double*, double*) { oak::set_error("oak_gpr_log_marginal_grad: not implemented yet"); return OAK_E_ARG; }
int oak_sobol(oak_ctx*, const oak_kernel_desc*, const double*, int64_t, int32_t, const double*, const int32_t*, const int32_t*, int32_t, int32_t, double, double, double*) { oak::set_error("oak_sobol: not implemented yet"); return OAK_E_ARG; }
int oak_component_predict(oak_ctx*, const oak_kernel_desc*, const double*, int64_t, const double*, int64_t, int32_t, const double*, const int32_t*, const int32_t*, int32_t, int32_t, double*) { oak::set_error("oak_component_predict: not implemented yet"); return OAK_E_ARG; }
int oak_comm_unique_id(char*) { oak::set_error("comm: not implemented yet"); return OAK_E_NCCL; }
int oak_comm_init(oak_ctx*, const char*, int32_t, int32_t) { oak::set_error("comm: not implemented yet"); return OAK_E_NCCL; }
int oak_comm_destroy(oak_ctx*) { return OAK_OK; }
int oak_comm_allreduce_stats(oak_ctx*) { oak::set_error("comm: not implemented yet"); return OAK_E_NCCL; }
int oak_comm_allreduce_host(oak_ctx*, double*, int64_t) { oak::set_error("comm: not implemented yet"); return OAK_E_NCCL; }
}
namespace oak { int comm_allreduce_dev(oak_ctx*, double*, int64_t) { set_error("comm: not implemented yet"); return OAK_E_NCCL; } }

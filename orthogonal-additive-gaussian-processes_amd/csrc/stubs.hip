// Temporary: entry points still to be implemented (kept so the library links and exports every symbol).
#include "oak_internal.h"
extern "C" {
int64_t oak_grad_len(const oak_kernel_desc* desc) { return desc ? 2 * desc->num_dims + desc->n_order_var + 1 + desc->meas_data_len : 0; }
int oak_sgpr_elbo_grad(oak_ctx*, const oak_kernel_desc*, double, double, double*, double*) { oak::set_error("oak_sgpr_elbo_grad: not implemented yet"); return OAK_E_ARG; }
int oak_gpr_log_marginal_grad(oak_ctx*, const oak_kernel_desc*, double, double*, double*) { oak::set_error("oak_gpr_log_marginal_grad: not implemented yet"); return OAK_E_ARG; }
}

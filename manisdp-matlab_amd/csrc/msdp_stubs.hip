// Temporary: entry points of units not built yet.
#include "msdp_common.h"
#define UNSUP(name) { msdp_set_error(name ": not implemented in this build"); return MSDP_EUNSUPPORTED; }
int msdp_affine_costgrad(msdp_handle, int) UNSUP("affine costgrad")
int msdp_affine_hess(msdp_handle) UNSUP("affine hess")
int msdp_affine_setup(msdp_handle, const int64_t*, const int64_t*, const double*, const double*, const double*) UNSUP("affine setup")
int msdp_affine_set_multipliers(msdp_handle, const double*, double) UNSUP("affine multipliers")
int msdp_affine_linesearch_cost(msdp_handle, const double*, double*) UNSUP("affine linesearch")
int msdp_sphere_upd2(msdp_handle) UNSUP("sphere upd2")
int msdp_sphere_retract(msdp_handle) UNSUP("sphere retract")
int msdp_sphere_proj(msdp_handle, const double*, const double*, double*) UNSUP("sphere proj")
int msdp_sphere_retr(msdp_handle, const double*, const double*, double*, double) UNSUP("sphere retr")
int msdp_escape_impl(msdp_handle, int, double, int, double*, double*, double*, int*) UNSUP("escape eigs")

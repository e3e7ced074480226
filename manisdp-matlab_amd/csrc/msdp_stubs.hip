// Temporary: entry points of units not built yet.
#include "msdp_common.h"
#define UNSUP(name) { msdp_set_error(name ": not implemented in this build"); return MSDP_EUNSUPPORTED; }
int msdp_escape_impl(msdp_handle, int, double, int, double*, double*, double*, int*) UNSUP("escape eigs")

#include "common.h"
int eigh_batched(sc_ctx* ctx, double*, int64_t, int64_t, double*, double*) {
  return sc_set_error(ctx, SC_ERR_INVALID_ARG, "eigensolver not built yet");
}
size_t eigh_workspace_bytes(int64_t, int64_t, bool) { return 0; }

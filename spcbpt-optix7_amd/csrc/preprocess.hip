// Host-owned preprocessing (optixPathTracer.cpp:552-608) — filled in by the preprocessing milestone.
#include "context.h"

namespace spc {
int Context::launch_pretrace(uint32_t) { error = "\"pretrace\" is not built in this revision"; return SPCBPT_ERR_STATE; }
int Context::preprocess(int, int, bool) { error = "spcbpt_preprocess is not built in this revision"; return SPCBPT_ERR_STATE; }
}  // namespace spc

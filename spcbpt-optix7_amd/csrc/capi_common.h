// Shared prelude of the C ABI's translation units (round 6: capi.hip, 2 000 lines, became seven files):
//   ctx_state.hip     Context: timing spans, buffer sets, the subspace tuple, environment, light-pass geometry, capacities, lifetime
//   ctx_light.hip     Context: light passes (single / batched), the device sampler build (single / batched)
//   ctx_exchange.hip  Context: the shards and imports of a sharded job (what libspcbpt_mgpu drives)
//   ctx_render.hip    Context: eye / pt launches (single, deferred, batched), film merges
//   capi.hip          extern "C": create / destroy, state, launches by name
//   capi_exchange.hip extern "C": LVC export / import, film band packing
//   capi_debug.hip    extern "C": read-backs, counters, timing, debug and test hooks, preprocessing entry points
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/spcbpt.h"
#include "context.h"
#include "kernels.h"
namespace spc { void launch_repack_nodes_quad2(const float* nodes_q, float* out, int n_nodes, hipStream_t s); }   // quad_trace.hip (declared here: kernels.h is part of the megakernel's source hash)
#include "lbvh.h"
#include "env_host.h"
void spc_viewers_forget_context(spcbpt_ctx* ctx);   // viewer.cpp: called by spcbpt_destroy, so that a viewer outliving its context is safe

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e__);                   \
            return SPCBPT_ERR_HIP;                                                               \
        }                                                                                        \
    } while (0)

namespace spc {

template <class T>
static hipError_t dev_alloc(T** p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T));
}
template <class T>
static void dev_free(T*& p) {
    if (p) (void)hipFree((void*)p);
    p = nullptr;
}

}  // namespace spc

struct spcbpt_ctx : public spc::Context {};

#define CTX_CHECK(c)                                  \
    if (!(c)) return SPCBPT_ERR_INVALID_ARG;          \
    if (hipSetDevice((c)->device) != hipSuccess) { (c)->error = "hipSetDevice failed"; return SPCBPT_ERR_HIP; }

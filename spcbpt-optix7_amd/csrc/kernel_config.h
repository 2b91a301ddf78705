// Launch-shape constants shared by the kernel files (round 6: kernels.hip, 1 900 lines, became four translation units:
//   kernels.hip          the eye megakernel k_spcbpt, k_pt, the film merges and their launchers
//   kernels_light.hip    k_light_trace, the cache compaction, the shard / band packing of a sharded job
//   kernels_sampler.hip  the device sampler build (MyThrustOp::LVC_Process), single and batched, both forms
//   kernels_train.hip    the standalone traversal kernels and the pre-trace (TrainData) kernel
// each with the launch_* functions kernels.h declares for its kernels).
#pragma once
#include "device_lib.h"
#include "kernels.h"

namespace spc {

static constexpr int BLOCK = 256;
static constexpr int STACK_LDS = kStackLds;  // LDS stack entries per lane; deeper entries spill (TravStack)
#ifndef SPC_PRIO_LIGHT
#define SPC_PRIO_LIGHT 0   // issue priority of the light pass's waves (they share CUs with the eye megakernel when passes run ahead)
#endif
#ifndef SPC_WAVES
// minimum waves per SIMD requested from the register allocator for the kernels beside the eye megakernel.  4 like the eye kernel, and for
// its sake: with three 128-VGPR eye blocks resident on a CU, 128 registers per lane are what is left -- a light-pass block that
// wants 154 would only fit on CUs holding two eye blocks or fewer and starve next to a persistent eye kernel (measured: the step
// got SLOWER with the faster eye kernel until the light pass was compiled to fit)
#define SPC_WAVES 4
#endif

}  // namespace spc

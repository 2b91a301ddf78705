#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "layout.h"

namespace spc {

int render_thread_count(const KParams& p);
int spcbpt_block_threads();   // threads per block of the eye megakernel (kernels.hip: EYE_BLOCK)
void launch_spcbpt(const KParams& p, int variant, int max_blocks, hipStream_t s);   // 0 timed, 1 reference order + counters (generic), 2 timed + counters
int spcbpt_blocks_per_cu(int variant, bool batch, bool general);
int render_tile_count(const KParams& p);
void launch_pt(const KParams& p, bool count, hipStream_t s);
void launch_film_merge(const KParams& p, hipStream_t s);
void launch_light_trace(const KParams& p, int variant, int max_blocks, hipStream_t s);
struct CompactBatch { LightVertex* lvc[kMaxBatchFrames]; int* counts[kMaxBatchFrames]; };   // per frame of a batched light pass: compact LVC + (vertex_count, path_count) of its set
void launch_lvc_compact_batch(const LightVertex* scratch, const int* core_counts, const int* core_offsets, const int* path_counts, int core_count,
                              int core_padding, int n, const CompactBatch& dst, int capacity, uint32_t* overflow, hipStream_t s);
int light_trace_blocks(const KParams& p, int max_blocks);   // grid of the (batched) light pass: the spill area is sized for it
void launch_lvc_compact(const LightVertex* scratch, const int* core_counts, const int* core_offsets, int core_count, int core_padding,
                        LightVertex* lvc, uint32_t* keys, uint32_t* vals, float* weights, int* sampler_counts, int capacity, uint32_t* overflow,
                        hipStream_t s);
void launch_fill_keys(const LightVertex* lvc, int n, uint32_t* keys, uint32_t* vals, float* weights, int* sampler_counts, hipStream_t s);
void launch_fill_keys_devcount(const LightVertex* lvc, int bound, uint32_t* keys, uint32_t* vals, float* weights, const int* sampler_counts, hipStream_t s);
void launch_gather_compact(const LightVertex* gathered, const int* counts_all, int world, int cap, int lvc_capacity, const CompactBatch& dst, int nf,
                           int* overflow, hipStream_t s);
void launch_pack_shards(const CompactBatch& src, int nf, int cap, LightVertex* send, int* send_counts, hipStream_t s);
// the film merges of the frames of a batched eye launch, in frame order, as one pass over the pixels
struct MergeBatch { const float* result[kMaxBatchFrames]; uint32_t subframe[kMaxBatchFrames]; };
void launch_film_merge_batch(const KParams& p, const MergeBatch& m, int frames, hipStream_t s);
void launch_pack_bands(float* accum, int width, int height, int rank, int world, float* packed, bool unpack_all, hipStream_t s);
// the sampler build as one stable counting sort (four launches): kernels.hip "sampler build in four launches"
size_t sampler_build_hist_ints();
// ... of up to kMaxBatchFrames caches at once (blockIdx.y = frame): what differs per frame, and the scratch the frames share --
// frame f uses keys / weights / wsorted + f * item_stride and hist + f * sampler_build_hist_ints()
struct SamplerBuildBatch {
    const LightVertex* lvc[kMaxBatchFrames]; const int* n_dev[kMaxBatchFrames]; int* path_count[kMaxBatchFrames];
    DSubspace* sub[kMaxBatchFrames]; uint32_t* jump[kMaxBatchFrames]; float* cmfs[kMaxBatchFrames];
    LightVertex* lvc_sorted[kMaxBatchFrames];   // out: lvc_sorted[i] = lvc[jump[i]] (may be null)
    uint32_t* guide[kMaxBatchFrames];           // out: the second-stage guide table (layout.h KParams::guide; may be null)
    int n_host[kMaxBatchFrames];
    uint32_t* keys; float* weights; int* hist; double* wsorted; size_t item_stride;
};
void launch_sampler_build_batch(const SamplerBuildBatch& b, int frames, hipStream_t s);
void launch_sampler_build(const LightVertex* lvc, int n_host, const int* n_dev, uint32_t* keys, float* weights, int* hist, int* path_count, DSubspace* sub,
                          uint32_t* jump, double* wsorted, float* cmfs, LightVertex* lvc_sorted, uint32_t* guide, hipStream_t s);
void launch_sampler_guide(const DSubspace* sub, const float* cmfs, uint32_t* guide, hipStream_t s);   // (the radix-sort form of the build)
void launch_lvc_sorted_copy(const LightVertex* lvc, const uint32_t* jump, const int* sampler_counts, LightVertex* lvc_sorted, int capacity, hipStream_t s);   // (the radix-sort form of the build)
void launch_subspace_ranges(const uint32_t* sorted_keys, const int* sampler_counts, DSubspace* sub, int capacity, hipStream_t s);
void launch_gather_weights(const float* weights, const uint32_t* sorted_vals, const int* sampler_counts, double* out, int capacity, hipStream_t s);
void launch_cmf(const double* prefix, const uint32_t* sorted_keys, const int* sampler_counts, DSubspace* sub, float* cmfs, int capacity, hipStream_t s);
int spcbpt_batch_blocks(const KParams& p, int max_blocks);   // grid of the batched launch (the spill area is sized for it)
void launch_spcbpt_batch(const KParams& p, int max_blocks, hipStream_t s);
void launch_trace_closest(const KParams& p, const float* rays, int n, float* t, int* tri, float* uv, hipStream_t s);
void launch_pretrace(const KParams& p, uint32_t iteration, int num_core, int padding, spcbpt_pretrace_path* paths,
                     spcbpt_pretrace_node* nodes, hipStream_t s);
void launch_trace_any(const KParams& p, const float* rays, int n, int* vis, hipStream_t s);

// "SPCBPT_no_rmis": the full-path-MIS variant (full_mis.hip)
void launch_spcbpt_no_rmis(const KParams& p, hipStream_t s);

// traversal schedule A/B (quad_trace.hip): mode 0 lane per ray, 1 quad per ray; persistent, pool-fed
void launch_repack_nodes_quad(const float* nodes, float* out, int n_nodes, hipStream_t s);
int trace_bench_blocks_per_cu(int mode, bool any);
void launch_trace_bench(const KParams& p, int mode, bool any, bool stats, const float* nodes_q, const float* rays, int n, uint32_t* counter, float* t, int* tri,
                        float* uv, int* vis, unsigned long long* stat_out, int blocks, hipStream_t s);

// per-function harness (unit.hip)
void launch_unit(const KParams& p, int op, const uint32_t* in, int in_words, uint32_t* out, int out_words, int n, const float* aux, hipStream_t s);

#ifndef SPC_STACK_LDS
#define SPC_STACK_LDS 16
#endif
static const int kStackLds = SPC_STACK_LDS;  // LDS traversal-stack entries per lane (16 KB per block): with the ray pool (4 x 5 840 B) and the hot-node table (1 216 B) of k_spcbpt a block takes exactly 40 960 B and four blocks fit the 160 KB of a CU; deeper stacks spill to HBM (TravStack)

}  // namespace spc

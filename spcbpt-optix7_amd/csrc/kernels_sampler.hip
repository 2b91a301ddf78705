// The device sampler build <- MyThrustOp::LVC_Process (cuda_thrust/device_thrust.cu:241-332): counting-sort form (single and batched) and the radix-sort form's helper kernels
// (kernel_config.h maps the kernel files)
#include <hip/hip_runtime.h>

#include "device_lib.h"
#include "eye_walk.h"
#include "kernel_config.h"
#include "kernels.h"

namespace spc {

__global__ void k_subspace_ranges(const uint32_t* __restrict__ sorted_keys, const int* __restrict__ sampler_counts, DSubspace* __restrict__ sub) {
    const int n = sampler_counts[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = sorted_keys[i];
    if (i == 0 || sorted_keys[i - 1] != k) sub[k].jump_bias = i;
    if (i == n - 1 || sorted_keys[i + 1] != k) sub[k].size = i + 1;  // temporarily the END position; fixed in k_finish_ranges
}
__global__ void k_finish_ranges(DSubspace* __restrict__ sub) {
    // one block of 1024 threads: empty subspaces get jump_bias = end of the last non-empty one before them, like the
    // running offset of the reference's host loop (device_thrust.cu:301-309) -> inclusive max-scan of the END positions
    __shared__ int ends[1024];
    const int s = threadIdx.x;
    const int end = s < SPCBPT_NUM_SUBSPACE ? sub[s].size : 0;  // END position written by k_subspace_ranges, 0 if empty
    ends[s] = end;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = s >= off ? ends[s - off] : 0;
        __syncthreads();
        ends[s] = max(ends[s], v);
        __syncthreads();
    }
    if (s < SPCBPT_NUM_SUBSPACE) {
        if (end > 0) sub[s].size = end - sub[s].jump_bias;
        else { sub[s].jump_bias = s > 0 ? ends[s - 1] : 0; sub[s].size = 0; }
    }
}
__global__ void k_gather_weights(const float* __restrict__ weights, const uint32_t* __restrict__ sorted_vals, const int* __restrict__ sampler_counts,
                                 double* __restrict__ out) {
    const int n = sampler_counts[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)weights[sorted_vals[i]];
}
__global__ void k_cmf(const double* __restrict__ prefix, const uint32_t* __restrict__ sorted_keys, const int* __restrict__ sampler_counts,
                      DSubspace* __restrict__ sub, float* __restrict__ cmfs) {
    const int n = sampler_counts[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = sorted_keys[i];
    const int b = sub[k].jump_bias, e = b + sub[k].size;
    const double base = b > 0 ? prefix[b - 1] : 0.0;
    const double total = prefix[e - 1] - base;
    const bool last = i == e - 1;
    // sum_pmf == 0 gives NaN CMFs in the reference; guarded here: zero-weight subspaces sample uniformly
    float c = total > 0.0 ? (float)((prefix[i] - base) / total) : (float)(i - b + 1) / (float)(e - b);
    if (last) { c = 1.0f; sub[k].sum_pmf = (float)total; }
    cmfs[i] = c;
}

// ---- sampler build in four launches ------------------------------------------------------------------------------------------
// The build above is ~14 dependent launches (hipcub's radix sort and scan are five and two of them): 0.3 ms of launch latency
// however few vertices it sorts, and a batched eye launch waits for up to 16 of them.  Subspace ids are 10-bit keys, so one stable
// counting sort does: SB_BLOCKS single-wave blocks each own a contiguous chunk of the cache,
//   k_sb_hist     per-block histogram of the ids (LDS), keys + weights (+ the path count) on the way
//   k_sb_scan     one block: per-id running offsets over the blocks, exclusive scan over the ids -> jump_bias / size
//   k_sb_scatter  each block places its chunk in order (rank among equal ids inside a wave from ten ballots) -> jump buffer,
//                 weights in sorted order
//   k_sb_cmf      one block per subspace: double-precision scan of its weights -> CMF, sum_pmf
// Same tables as the sort: the order inside a subspace is the cache order (stable), empty subspaces carry the running offset.
// The CMF sums a subspace's weights by themselves (the scan above takes differences of a global prefix): equal to 1e-16 relative.
static constexpr int SB_BLOCKS = 512;
// (blockIdx.y = frame of a batched build: SamplerBuildBatch, kernels.h; a single build is a batch of one)
__global__ __launch_bounds__(64) void k_sb_hist(const SamplerBuildBatch B) {
    const int f = blockIdx.y;
    const LightVertex* __restrict__ lvc = B.lvc[f];
    const int n_host = B.n_host[f];
    const int* __restrict__ n_dev = B.n_dev[f];
    uint32_t* __restrict__ keys = B.keys + (size_t)f * B.item_stride;
    float* __restrict__ weights = B.weights + (size_t)f * B.item_stride;
    int* __restrict__ hist = B.hist + (size_t)f * (SB_BLOCKS + 1) * 1024;
    int* __restrict__ path_count = B.path_count[f];
    __shared__ uint32_t h[1024];
    const int lane = threadIdx.x, b = blockIdx.x;
#pragma unroll
    for (int t = 0; t < 16; t++) h[t * 64 + lane] = 0u;
    __syncthreads();
    const int n = n_dev ? n_dev[0] : n_host;
    const int chunk = (n + SB_BLOCKS - 1) / SB_BLOCKS, i0 = b * chunk, i1 = min(n, i0 + chunk);
    int starts = 0;
    for (int i = i0 + lane; i < i1; i += 64) {
        const LightVertex& v = lvc[i];
        float w = (v.flux[0] + v.flux[1] + v.flux[2]) / v.pdf;   // LVCSubspaceInfoCopy device_thrust.cu:191-212
        if (isinf(w) || isnan(w)) w = 0.0f;
        const uint32_t k = (uint32_t)v.subspace_id & 1023u;
        keys[i] = k;
        weights[i] = w;
        atomicAdd(&h[k], 1u);
        starts += v.depth == 0 ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; t++) hist[(size_t)b * 1024 + t * 64 + lane] = (int)h[t * 64 + lane];
    if (path_count) {
        for (int o = 32; o > 0; o >>= 1) starts += __shfl_down(starts, o, 64);
        if (lane == 0 && starts) atomicAdd(path_count, starts);
    }
}
__global__ __launch_bounds__(1024) void k_sb_scan(const SamplerBuildBatch B) {
    int* __restrict__ hist = B.hist + (size_t)blockIdx.y * (SB_BLOCKS + 1) * 1024;
    DSubspace* __restrict__ sub = B.sub[blockIdx.y];
    // thread = subspace id.  hist[b][id] becomes the number of items with that id in the blocks before b; row SB_BLOCKS receives the
    // position of the id's first item = items with smaller ids: for an empty subspace the end of the last non-empty one before it,
    // the running offset of the reference's host loop (device_thrust.cu:301-309)
    __shared__ int tot[1024];
    const int k = threadIdx.x;
    int run = 0;
#pragma unroll 8
    for (int b = 0; b < SB_BLOCKS; b++) {
        const int c = hist[(size_t)b * 1024 + k];
        hist[(size_t)b * 1024 + k] = run;
        run += c;
    }
    tot[k] = run;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = k >= off ? tot[k - off] : 0;
        __syncthreads();
        tot[k] += v;
        __syncthreads();
    }
    const int base = tot[k] - run;   // exclusive
    hist[(size_t)SB_BLOCKS * 1024 + k] = base;
    if (k < SPCBPT_NUM_SUBSPACE) { sub[k].jump_bias = base; sub[k].size = run; sub[k].sum_pmf = 0.0f; sub[k].pad = 0; }
}
__global__ __launch_bounds__(64) void k_sb_scatter(const SamplerBuildBatch B) {
    const int f = blockIdx.y;
    const uint32_t* __restrict__ keys = B.keys + (size_t)f * B.item_stride;
    const float* __restrict__ weights = B.weights + (size_t)f * B.item_stride;
    const int n_host = B.n_host[f];
    const int* __restrict__ n_dev = B.n_dev[f];
    const int* __restrict__ hist = B.hist + (size_t)f * (SB_BLOCKS + 1) * 1024;
    uint32_t* __restrict__ jump = B.jump[f];
    double* __restrict__ wsorted = B.wsorted + (size_t)f * B.item_stride;
    __shared__ uint32_t next[1024];   // where this block's next item of each id goes
    const int lane = threadIdx.x, b = blockIdx.x;
#pragma unroll
    for (int t = 0; t < 16; t++) next[t * 64 + lane] = (uint32_t)(hist[(size_t)SB_BLOCKS * 1024 + t * 64 + lane] + hist[(size_t)b * 1024 + t * 64 + lane]);
    __syncthreads();
    const int n = n_dev ? n_dev[0] : n_host;
    const int chunk = (n + SB_BLOCKS - 1) / SB_BLOCKS, i0 = b * chunk, i1 = min(n, i0 + chunk);
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = i0; base < i1; base += 64) {   // wave-uniform bounds: every lane takes part in the ballots
        const int i = base + lane;
        const bool valid = i < i1;
        const uint32_t k = valid ? keys[i] : 0u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 10; bit++) {
            const unsigned long long m = __ballot((k >> bit) & 1u);
            peers &= ((k >> bit) & 1u) ? m : ~m;
        }
        uint32_t pos = 0u;
        if (valid) pos = next[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt) == 0ull) next[k] = pos + (uint32_t)__popcll(peers);   // the first lane of each id moves the cursor on
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint32_t dst = pos + (uint32_t)__popcll(peers & lt);
            jump[dst] = (uint32_t)i;
            wsorted[dst] = (double)weights[i];
        }
    }
}
// the cache in the sampler's order: record i = lvc[jump[i]], one lane per QUAD (six consecutive lanes read one 96-B vertex and write
// its six quads next to each other: the stores of a wave are contiguous, the loads are whole records)
__global__ __launch_bounds__(256) void k_sb_copy(const SamplerBuildBatch B) {
    const int f = blockIdx.y;
    float4* __restrict__ dst = reinterpret_cast<float4*>(B.lvc_sorted[f]);
    if (!dst) return;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(B.lvc[f]);
    const uint32_t* __restrict__ jump = B.jump[f];
    const int* __restrict__ n_dev = B.n_dev[f];
    const long long n = n_dev ? n_dev[0] : B.n_host[f];
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n * 6; t += (long long)gridDim.x * 256) {
        const long long i = t / 6;
        const int q = (int)(t - i * 6);
        dst[t] = src[(size_t)jump[i] * 6 + q];
    }
}
// KParams::guide of one subspace: entry j = the first place k with cmf[k] > (j / n)(1 - 2^-20).  A random number u of bucket j --
// (int)(u * (float)n) == j, the product rounded to FP32 -- is at least (j / n)(1 - 2^-24), so no entry before that place is above u.
SPC_DEV void build_guide(const float* cmf, int n, uint32_t* guide, int t, int stride) {
    for (int j = t; j < n; j += stride) {
        const double tj = (double)j / (double)n * (1.0 - 1.0 / 1048576.0);
        int lo = 0, hi = n - 1;   // (the last entry is 1)
        while (lo < hi) {
            const int m = (lo + hi) >> 1;
            if ((double)cmf[m] > tj) hi = m; else lo = m + 1;
        }
        guide[j] = (uint32_t)lo;
    }
}
__global__ __launch_bounds__(256) void k_sb_cmf(const SamplerBuildBatch B) {
    DSubspace* __restrict__ sub = B.sub[blockIdx.y];
    const double* __restrict__ wsorted = B.wsorted + (size_t)blockIdx.y * B.item_stride;
    float* cmfs = B.cmfs[blockIdx.y];   // (read back for the guide table below: not __restrict__)
    __shared__ double sh[256];
    const int k = blockIdx.x, t = threadIdx.x;
    const int b = sub[k].jump_bias, sz = sub[k].size;
    if (sz <= 0) return;
    double acc = 0.0;
    for (int j = t; j < sz; j += 256) acc += wsorted[b + j];
    sh[t] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (t < off) sh[t] += sh[t + off]; __syncthreads(); }
    const double total = sh[0];
    __syncthreads();
    double carry = 0.0;
    for (int j0 = 0; j0 < sz; j0 += 256) {
        const int j = j0 + t;
        const double w = j < sz ? wsorted[b + j] : 0.0;
        sh[t] = w;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const double v = t >= off ? sh[t - off] : 0.0;
            __syncthreads();
            sh[t] += v;
            __syncthreads();
        }
        if (j < sz) {
            // sum_pmf == 0 gives NaN CMFs in the reference; guarded here: zero-weight subspaces sample uniformly
            float c = total > 0.0 ? (float)((carry + sh[t]) / total) : (float)(j + 1) / (float)sz;
            if (j == sz - 1) c = 1.0f;
            cmfs[b + j] = c;
        }
        carry += sh[255];
        __syncthreads();
    }
    if (t == 0) sub[k].sum_pmf = (float)total;
    if (B.guide[blockIdx.y]) build_guide(cmfs + b, sz, B.guide[blockIdx.y] + b, t, 256);   // (the loop above ends on a barrier: the block's CMF is written)
}
// the second-stage guide table next to a CMF that another path has written (the radix-sort build)
__global__ __launch_bounds__(256) void k_sb_guide(const DSubspace* __restrict__ sub, const float* cmfs, uint32_t* __restrict__ guide) {
    const int b = sub[blockIdx.x].jump_bias, sz = sub[blockIdx.x].size;
    if (sz > 0) build_guide(cmfs + b, sz, guide + b, threadIdx.x, 256);
}
void launch_sampler_guide(const DSubspace* sub, const float* cmfs, uint32_t* guide, hipStream_t s) {
    if (guide) hipLaunchKernelGGL(k_sb_guide, dim3(SPCBPT_NUM_SUBSPACE), dim3(256), 0, s, sub, cmfs, guide);
}
size_t sampler_build_hist_ints() { return (size_t)(SB_BLOCKS + 1) * 1024; }
void launch_sampler_build_batch(const SamplerBuildBatch& b, int frames, hipStream_t s) {
    if (frames <= 0) return;
    hipLaunchKernelGGL(k_sb_hist, dim3(SB_BLOCKS, frames), dim3(64), 0, s, b);
    hipLaunchKernelGGL(k_sb_scan, dim3(1, frames), dim3(1024), 0, s, b);
    hipLaunchKernelGGL(k_sb_scatter, dim3(SB_BLOCKS, frames), dim3(64), 0, s, b);
    hipLaunchKernelGGL(k_sb_cmf, dim3(SPCBPT_NUM_SUBSPACE, frames), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_sb_copy, dim3(256, frames), dim3(256), 0, s, b);
}
__global__ __launch_bounds__(256) void k_lvc_sorted_copy(const LightVertex* __restrict__ lvc, const uint32_t* __restrict__ jump, const int* __restrict__ counts,
                                                        LightVertex* __restrict__ out, int capacity) {
    const int n = min(counts[0], capacity);
    const float4* src = reinterpret_cast<const float4*>(lvc);
    float4* dst = reinterpret_cast<float4*>(out);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const size_t from = jump[i];
#pragma unroll
        for (int q = 0; q < 6; q++) dst[(size_t)i * 6 + q] = src[from * 6 + q];
    }
}
void launch_lvc_sorted_copy(const LightVertex* lvc, const uint32_t* jump, const int* sampler_counts, LightVertex* lvc_sorted, int capacity, hipStream_t s) {
    if (capacity <= 0 || !lvc_sorted) return;
    hipLaunchKernelGGL(k_lvc_sorted_copy, dim3(512), dim3(256), 0, s, lvc, jump, sampler_counts, lvc_sorted, capacity);
}
void launch_sampler_build(const LightVertex* lvc, int n_host, const int* n_dev, uint32_t* keys, float* weights, int* hist, int* path_count, DSubspace* sub,
                          uint32_t* jump, double* wsorted, float* cmfs, LightVertex* lvc_sorted, uint32_t* guide, hipStream_t s) {
    SamplerBuildBatch b = {};
    b.guide[0] = guide;
    b.lvc[0] = lvc; b.n_host[0] = n_host; b.n_dev[0] = n_dev; b.path_count[0] = path_count; b.sub[0] = sub; b.jump[0] = jump; b.cmfs[0] = cmfs;
    b.lvc_sorted[0] = lvc_sorted;
    b.keys = keys; b.weights = weights; b.hist = hist; b.wsorted = wsorted; b.item_stride = 0;
    launch_sampler_build_batch(b, 1, s);
}

void launch_subspace_ranges(const uint32_t* sorted_keys, const int* sampler_counts, DSubspace* sub, int capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_subspace_ranges, dim3((capacity + 255) / 256), dim3(256), 0, s, sorted_keys, sampler_counts, sub);
    hipLaunchKernelGGL(k_finish_ranges, dim3(1), dim3(1024), 0, s, sub);
}
void launch_gather_weights(const float* weights, const uint32_t* sorted_vals, const int* sampler_counts, double* out, int capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_weights, dim3((capacity + 255) / 256), dim3(256), 0, s, weights, sorted_vals, sampler_counts, out);
}
void launch_cmf(const double* prefix, const uint32_t* sorted_keys, const int* sampler_counts, DSubspace* sub, float* cmfs, int capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_cmf, dim3((capacity + 255) / 256), dim3(256), 0, s, prefix, sorted_keys, sampler_counts, sub, cmfs);
}

}  // namespace spc

// The light pass: k_light_trace <- __raygen__lightTrace + __closesthit__lightSubpath (raygen.cu:620-685, hit_program.cu:341-438), the compaction of the cache, and the shard / band packing of a sharded job
// (kernel_config.h maps the kernel files)
#include <hip/hip_runtime.h>

#include "device_lib.h"
#include "eye_walk.h"
#include "kernel_config.h"
#include "kernels.h"

namespace spc {

// ------------------------------------------------------------------------------------------------
// Light pass.  A core walks m_per_core light paths and fills its own padded slot range, exactly the launch geometry of
// LightTraceParams; the MI355X default is num_core = M, m_per_core = 1 (one path per core).  Persistent waves with per-lane
// regeneration, like the eye pass: a lane whose core is finished takes the next core of a global queue at once (one atomic
// per wave and refill), and every iteration of the wave advances all live paths by one segment.  Light paths end after 2.6
// vertices on average but may run for 50, so one core per lane for the whole launch kept a wave resident for its longest
// path with 1-2 live lanes; regenerating waves do the same work with a quarter of the resident blocks, which matters because
// the pass shares the GPU with persistent eye kernels that never yield a block slot.  What a core computes and where it
// stores it does not depend on the lane that runs it: seeds come from the global core index, slots from the core's range.
template <bool COUNT, bool CACHE>
__global__ __launch_bounds__(BLOCK, SPC_WAVES) void k_light_trace(const KParams p) {
    __shared__ uint32_t s_stack[BLOCK * STACK_LDS];
    if (SPC_PRIO_LIGHT) __builtin_amdgcn_s_setprio(SPC_PRIO_LIGHT);
    const DeviceScene& S = p.scene;
    const uint32_t lane = threadIdx.x & 63;
    Counts<COUNT> cn;
    cn.clear();
    TravStack<BLOCK, STACK_LDS> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)blockIdx.x * BLOCK + threadIdx.x, p.diag);
    int paths_started = 0;
    // per-core state
    bool has_core = false, exhausted = false;
    int local_core = 0, nverts = 0, npaths = 0, origins = 0;
    uint32_t seed = 0, pseed = 0;
    LightVertex* slots = nullptr;
    const bool batched = p.n_lframes > 0;   // several frames' passes in one queue (each lane then counts its cores' paths with one atomic per core)
    const uint32_t queue_len = batched ? (uint32_t)p.core_count * (uint32_t)p.n_lframes : (uint32_t)p.core_count;
    int fk = 0;   // frame of the lane's core within a batched pass
    // per-path state
    bool in_path = false;
    f3 origin = mk3(0.0f), dir = mk3(0.0f), next_flux = mk3(0.0f);
    float next_single_pdf = 0.0f;
    int depth = 0;
    uint32_t path_id = 0;
    LightVertex last;
    uint32_t pool_base = 0;
    int pool_left = 0;
    auto store = [&](const LightVertex& v) {
        float4* dst = reinterpret_cast<float4*>(slots + nverts);
        const float4* src = reinterpret_cast<const float4*>(&v);
#pragma unroll
        for (int q = 0; q < 6; q++) dst[q] = src[q];
        nverts++;
        cn.add(C_LVCW);
    };
    auto core_done = [&]() {   // the lane's core is complete: its vertex count, and the paths it started
        if (batched) {
            p.core_counts[(size_t)fk * (p.core_count + 1) + local_core] = nverts;
            atomicAdd(p.path_counter + fk, origins);
        } else { p.core_counts[local_core] = nverts; paths_started += origins; }
    };
    while (true) {
        // ---- regeneration: cores of the queue to lanes without one
        unsigned long long idle = __ballot(!has_core);
        while (idle != 0ull && !exhausted) {
            if (pool_left == 0) {
                uint32_t t = 0;
                if (lane == (uint32_t)__ffsll((long long)idle) - 1u) t = atomicAdd(p.work_counter, 64u);
                t = __shfl(t, __ffsll((long long)idle) - 1, 64);
                if (t >= queue_len) { exhausted = true; break; }
                pool_base = t;
                pool_left = min(64, (int)(queue_len - t));
            }
            const int n_idle = __popcll(idle);
            const int take = n_idle < pool_left ? n_idle : pool_left;
            const int my_rank = __popcll(idle & ((1ull << lane) - 1ull));
            if (!has_core && my_rank < take) {
                local_core = (int)pool_base + my_rank;
                has_core = true;
                uint32_t launch_frame = p.launch_frame;
                LightVertex* scratch = p.lvc_scratch;
                if (batched) {   // the queue spans the passes of n_lframes frames (layout.h: n_lframes)
                    fk = local_core / p.core_count;
                    local_core -= fk * p.core_count;
                    launch_frame += (uint32_t)fk;
                    scratch += (size_t)fk * p.core_count * p.core_padding;
                }
                const int core = p.core_begin + local_core;
                seed = tea4((uint32_t)core, launch_frame);  // light sampling stream
                // payload.seed: BSDF stream; the reference starts it equal to `seed`
                pseed = p.lt_decorrelate ? tea4((uint32_t)core ^ 0x80000000u, launch_frame) : seed;
                slots = scratch + (size_t)local_core * p.core_padding;
                nverts = 0; npaths = 0; origins = 0;
                in_path = false;
            }
            pool_base += (uint32_t)take;
            pool_left -= take;
            idle = __ballot(!has_core);
        }
        if (!__any(has_core)) break;   // queue exhausted and every core of the wave finished
        // ---- a core without a running path starts its next one: light sample + origin vertex (raygen.cu:620-668)
        if (has_core && !in_path) {
            const int lid = pick_light(S, seed);
            const DLight& L = S.lights[lid];
            LightSampleD ls;
            float dir_pdf;
            uint32_t origin_flags = 0u;
            if (L.type == 1) {   // the environment map: a sky direction, the sub-path starts on the sky disk and runs against it
                ls = env_light_sample(S, seed, dir_pdf);
                dir = ls.normal;
                origin_flags = SPCBPT_LV_DIRECTION;
            } else {
                const float r1 = rnd(seed), r2 = rnd(seed);
                ls = light_reverse_sample(S, L, r1, r2);
                const float d1 = rnd(seed), d2 = rnd(seed);  // traceMode
                const Onb onb(ls.normal);
                dir = onb.to_world(cosine_sample_hemisphere(d1, d2));
                dir_pdf = fabsf(dot(dir, ls.normal)) * kInvPi;
            }
            origin = ls.position;
            path_id = (uint32_t)(p.core_begin + local_core) * (uint32_t)p.m_per_core + (uint32_t)npaths;
            cn.add(C_LIGHT);
            // origin vertex (init_vertex_from_lightSample raygen.cu:172-195)
            LightVertex v;
            v.position[0] = ls.position.x; v.position[1] = ls.position.y; v.position[2] = ls.position.z; v.pdf = ls.pdf;
            v.normal[0] = ls.normal.x; v.normal[1] = ls.normal.y; v.normal[2] = ls.normal.z; v.single_pdf = ls.pdf;
            v.flux[0] = ls.emission.x; v.flux[1] = ls.emission.y; v.flux[2] = ls.emission.z; v.rmis_pointer = 1.0f;
            v.color[0] = v.color[1] = v.color[2] = 0.0f; v.last_lum = 0.0f;
            v.last_position[0] = v.last_position[1] = v.last_position[2] = 0.0f; v.last_normal_projection = 0.0f;
            v.material_id = (int16_t)L.id; v.subspace_id = (int16_t)ls.subspace; v.depth = 0; v.last_zone_id = 0;
            v.path_id = path_id; v.pad = origin_flags;
            store(v);
            origins++;
            last = v;
            next_flux = mk3(0.0f);
            next_single_pdf = dir_pdf;
            depth = 0;
            in_path = nverts < p.core_padding;   // a full slot range ends the core right after the origin vertex
            if (!in_path) { core_done(); has_core = false; }
        }
        // ---- one segment of every running path (hit_program.cu:341-438)
        const bool tracing = has_core && in_path;
        HitRec h;
        h.tri = -1;
        if (tracing) {
            cn.add(C_CLOSEST);
            traverse<false, COUNT>(S, st, origin, dir, kEps, 1e16f, h, cn);
        }
        if (tracing) {
            bool done = false, full = false;
            if (h.tri < 0) { done = true; }
            else {
                const Geom g = local_geometry(S, h);
                if (g.emitter) { done = true; }  // __closesthit__lightSource_subpath
                else {
                    Pbr pbr = load_pbr(S, g.mat);
                    color_tex_sample(S, g, pbr, cn);
                    f3 N = g.N;
                    if (dot(N, dir) > 0.f) N = -N;
                    const f3 inv_dir = -dir;
                    const f3 new_dir = bsdf_sample(pbr, N, inv_dir, pseed);
                    const float pdf = bsdf_pdf(pbr, N, inv_dir, new_dir);
                    if (!(pdf > 0.0f)) done = true;
                    const f3 last_n = ld3(last.normal), last_flux = ld3(last.flux);
                    const bool last_dir = (last.pad & SPCBPT_LV_DIRECTION) != 0u;   // LastVertex.is_DIRECTION(): parallel rays from the sky, no 1 / t^2 (hit_program.cu:372-375)
                    const float pdf_G = last_dir ? fabsf(dot(N, dir) * dot(last_n, dir)) : fabsf(dot(N, dir) * dot(last_n, dir)) / (h.t * h.t);
                    const f3 flux = last.depth == 0 ? last_flux * pdf_G : next_flux * last_flux * pdf_G;
                    LightVertex m;
                    m.position[0] = g.P.x; m.position[1] = g.P.y; m.position[2] = g.P.z;
                    m.normal[0] = N.x; m.normal[1] = N.y; m.normal[2] = N.z;
                    m.flux[0] = flux.x; m.flux[1] = flux.y; m.flux[2] = flux.z;
                    m.color[0] = pbr.base.x; m.color[1] = pbr.base.y; m.color[2] = pbr.base.z;
                    m.last_position[0] = last.position[0]; m.last_position[1] = last.position[1]; m.last_position[2] = last.position[2];
                    if (last_dir) { const f3 lp = g.P - dir; m.last_position[0] = lp.x; m.last_position[1] = lp.y; m.last_position[2] = lp.z; }   // hit_program.cu:386-389
                    m.last_normal_projection = fabsf(dot(last_n, dir));
                    m.material_id = (int16_t)g.mat;
                    // light-tree label of the new vertex and eye-tree relabel of the previous one (tracing_weight_light) in lock-step
                    int new_label, eye_label;
                    const f3 last_pos = ld3(last.position);
                    uint32_t own_eye_label = 0u;   // device_lib.h: label caching -- the new vertex's own eye-tree label + 1
                    if (CACHE) {
                        int own;
                        tree_label2<COUNT, true>(p.light_tree, g.P, N, inv_dir, true, p.eye_tree, g.P, N, inv_dir, true, new_label, own, cn);
                        own_eye_label = (uint32_t)own + 1u;
                        eye_label = (int)(last.pad & 0xffffu) - 1;   // the previous vertex's, cached when it was created (unused when it is the origin)
                    } else {
                        tree_label2(p.light_tree, g.P, N, inv_dir, true, p.eye_tree, last_pos, last_n, normalize(g.P - last_pos), last.depth != 0,
                                    new_label, eye_label, cn);
                    }
                    m.subspace_id = (int16_t)new_label;
                    m.last_zone_id = last.subspace_id;
                    m.depth = (int16_t)(last.depth + 1);
                    m.single_pdf = next_single_pdf * pdf_G / fabsf(dot(last_n, dir));
                    m.pdf = last.pdf * m.single_pdf;
                    m.last_lum = sum3(last_flux / last.pdf);
                    m.path_id = path_id; m.pad = own_eye_label | (last_dir ? SPCBPT_LV_LAST_DIRECTION : 0u);   // isLastVertex_direction (hit_program.cu:412: the predecessor is the origin)
                    if (last.depth == 0) {
                        m.rmis_pointer = last.rmis_pointer / last.single_pdf;  // tracing_init_light
                    } else {  // tracing_update_light (rmis.h:80-94)
                        const VCore lc = core_of(last);
                        const Pbr mat_last = load_pbr_colored(S, lc.mat, lc.color);
                        const f3 in_dir = normalize(g.P - lc.pos);
                        const float LL_pdf = rmis_last_pdf(mat_last, lc, in_dir);
                        const float wgt = rmis_weight_light_l(p, last.last_zone_id, last.last_lum, eye_label, cn);
                        m.rmis_pointer = (last.rmis_pointer * LL_pdf + wgt) / last.single_pdf;
                    }
                    cn.add(C_VERTEX);
                    next_flux = brdf_div(pbr, bsdf_eval(pbr, N, inv_dir, new_dir), N, new_dir);   // hit_program.cu:384
                    next_single_pdf = pdf;
                    origin = g.P;
                    dir = new_dir;
                    const float r = rnd(pseed);
                    const float rr = rr_of(pbr.base);
                    if (r > rr) done = true;
                    else next_single_pdf *= rr;
                    store(m);
                    last = m;
                    if (!(nverts < p.core_padding)) full = true;
                }
            }
            // the walk loop's exit tests (raygen.cu:646-676): slot range full -> the core ends; path done or too deep -> next path
            bool path_over = full;
            if (!full) {
                if (done || depth > 50) path_over = true;
                else depth += 1;
            }
            if (path_over) {
                in_path = false;
                bool core_over = full;
                if (!full) {
                    npaths++;
                    if (npaths >= p.m_per_core || !(nverts < p.core_padding)) core_over = true;
                }
                if (core_over) { core_done(); has_core = false; }
            }
        }
    }
    // path_count of the sampler (#depth-0 vertices, device_thrust.cu:324-326): one atomic per wave
    for (int o = 32; o > 0; o >>= 1) paths_started += __shfl_down(paths_started, o, 64);
    if (lane == 0 && paths_started) atomicAdd(p.path_counter, paths_started);
    cn.flush(p.counters);
}

// ------------------------------------------------------------------------------------------------
// Sampler build on device (LVC_Process).  Input: padded scratch + per-core counts.  Steps:
//   1. exclusive scan of core_counts (hipcub)                       -> core_offsets, vertex_count
//   2. k_lvc_compact: copy to the compact LVC in (core, slot) order, emit key = subspace id, weight, path starts
//   3. stable radix sort of (subspace id -> compact index) (hipcub)  -> jump_buffer
//   4. k_subspace_ranges: first/last position of each subspace in the sorted keys -> jump_bias, size
//   5. inclusive scan (double) of the weights in sorted order (hipcub), k_cmf: per-subspace normalised CMF
__global__ void k_lvc_compact(const LightVertex* __restrict__ scratch, const int* __restrict__ core_counts,
                              const int* __restrict__ core_offsets, int core_count, int core_padding, LightVertex* __restrict__ lvc,
                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, float* __restrict__ weights,
                              int* __restrict__ sampler_counts, int capacity, uint32_t* __restrict__ overflow) {
    // one thread per padded slot; only the filled slots (slot < count of its core) copy their 96-B record.  The compact cache holds
    // `capacity` vertices (sized from a measured pass with slack, not from the padded worst case: context.h); a pass that outgrows
    // it is cut off at the capacity and reported through *overflow (SPCBPT_ERR_CAPACITY at the next sync), never written past the end.
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && core_offsets[core_count] > capacity) { sampler_counts[0] = capacity; *overflow = 1u; }
    if (t < (long long)core_count * core_padding) {
        const int core = (int)(t / core_padding), slot = (int)(t % core_padding);
        if (slot < core_counts[core] && core_offsets[core] + slot < capacity) {
            const int dst_i = core_offsets[core] + slot;
            const float4* src = reinterpret_cast<const float4*>(scratch + t);
            float4* dst = reinterpret_cast<float4*>(lvc + dst_i);
            float4 q[6];
#pragma unroll
            for (int k = 0; k < 6; k++) q[k] = src[k];
#pragma unroll
            for (int k = 0; k < 6; k++) dst[k] = q[k];
            const LightVertex& v = *reinterpret_cast<const LightVertex*>(q);
            float w = (v.flux[0] + v.flux[1] + v.flux[2]) / v.pdf;  // LVCSubspaceInfoCopy device_thrust.cu:191-212
            if (isinf(w) || isnan(w)) w = 0.0f;
            keys[dst_i] = (uint32_t)v.subspace_id;
            vals[dst_i] = (uint32_t)dst_i;
            weights[dst_i] = w;
        }
    }
}

// Compaction of a batched light pass: grid.y = frame of the batch.  core_offsets is ONE exclusive scan over the n * (core_count + 1)
// counts (each frame's segment ends in a zero sentinel), so frame k's offsets are relative to its first entry and its total is the
// sentinel's offset minus that.  Keys are left to the sampler build (k_fill_keys_from_lvc), which also counts the paths again.
__global__ void k_lvc_compact_batch(const LightVertex* __restrict__ scratch, const int* __restrict__ core_counts, const int* __restrict__ core_offsets,
                                    const int* __restrict__ path_counts, int core_count, int core_padding, CompactBatch dst, int capacity,
                                    uint32_t* __restrict__ overflow) {
    const int k = blockIdx.y;
    const int* counts = core_counts + (size_t)k * (core_count + 1);
    const int* offs = core_offsets + (size_t)k * (core_count + 1);
    const int base = offs[0];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        const int total = offs[core_count] - base;
        if (total > capacity) *overflow = 1u;   // k_lvc_compact: cut off at the set's capacity and reported
        dst.counts[k][0] = min(total, capacity); dst.counts[k][1] = path_counts[k];
    }
    if (t < (long long)core_count * core_padding) {
        const int core = (int)(t / core_padding), slot = (int)(t % core_padding);
        if (slot < counts[core] && offs[core] - base + slot < capacity) {
            const float4* src = reinterpret_cast<const float4*>(scratch + (size_t)k * core_count * core_padding + t);
            float4* out = reinterpret_cast<float4*>(dst.lvc[k] + (offs[core] - base + slot));
            float4 q[6];
#pragma unroll
            for (int j = 0; j < 6; j++) q[j] = src[j];
#pragma unroll
            for (int j = 0; j < 6; j++) out[j] = q[j];
        }
    }
}

__global__ void k_fill_keys_from_lvc(const LightVertex* __restrict__ lvc, int n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                     float* __restrict__ weights, int* __restrict__ sampler_counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int start = 0;
    if (i < n) {
        const LightVertex& v = lvc[i];
        float w = (v.flux[0] + v.flux[1] + v.flux[2]) / v.pdf;
        if (isinf(w) || isnan(w)) w = 0.0f;
        keys[i] = (uint32_t)v.subspace_id;
        vals[i] = (uint32_t)i;
        weights[i] = w;
        start = v.depth == 0 ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) start += __shfl_down(start, o, 64);
    if ((threadIdx.x & 63) == 0 && start) atomicAdd(&sampler_counts[1], start);
}

// The same with the item count on the DEVICE (sampler_counts[0]) and a host-known upper bound `bound` as the grid: slots beyond the
// count get the pad key 1023 (no subspace id reaches it: ids are < 1000) and weight 0, so a 10-bit radix sort over `bound` items
// leaves the real items sorted in front.  sampler_counts[1] (path count) is left as the caller set it.
__global__ void k_fill_keys_devcount(const LightVertex* __restrict__ lvc, int bound, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                     float* __restrict__ weights, const int* __restrict__ sampler_counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bound) return;
    const int n = sampler_counts[0];
    if (i < n) {
        const LightVertex& v = lvc[i];
        float w = (v.flux[0] + v.flux[1] + v.flux[2]) / v.pdf;
        if (isinf(w) || isnan(w)) w = 0.0f;
        keys[i] = (uint32_t)v.subspace_id;
        weights[i] = w;
    } else {
        keys[i] = 1023u;
        weights[i] = 0.0f;
    }
    vals[i] = (uint32_t)i;
}

// Exchange 1 of a sharded job, receiving side: `gathered` holds `world` shards of `cap` slots each (the all-gather of every rank's
// compact shard, padded to the agreed capacity), counts_all[2 r] / [2 r + 1] the vertex / path count of rank r.  The shards are
// concatenated in rank order = global (path, depth) order into the set's LVC; the totals go to sampler_counts (device-resident:
// the sampler build sizes itself from them, no host round trip).  A shard that did not fit `cap` raises *overflow.
// Batched form (one exchange per light batch): grid.y = frame k of `nf`; rank q's block of the all-gather holds its nf shards one
// after the other, so frame k of rank q sits at (q nf + k) cap and its counts at 2 (q nf + k); every frame goes to its own set (dst).
__global__ void k_gather_compact(const LightVertex* __restrict__ gathered, const int* __restrict__ counts_all, int world, int cap, int lvc_capacity,
                                 CompactBatch dst, int nf, int* __restrict__ overflow) {
    const int chunks = (cap + 255) / 256;
    const int r = blockIdx.x / chunks, c = blockIdx.x % chunks, k = blockIdx.y;
    LightVertex* __restrict__ lvc = dst.lvc[k];
    int* __restrict__ sampler_counts = dst.counts[k];
    int base = 0, total = 0, paths = 0;
    bool over = false;
    for (int q = 0; q < world; q++) {
        const int n = counts_all[2 * (q * nf + k)];
        if (n > cap) over = true;
        if (q < r) base += min(n, cap);
        total += min(n, cap);
        paths += counts_all[2 * (q * nf + k) + 1];
    }
    if (total > lvc_capacity) over = true;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sampler_counts[0] = min(total, lvc_capacity);
        sampler_counts[1] = paths;
        if (over) *overflow = 1;
    }
    const int n_r = min(counts_all[2 * (r * nf + k)], cap);
    const int i = c * 256 + (int)threadIdx.x;
    if (i >= n_r || base + i >= lvc_capacity) return;
    const float4* src = reinterpret_cast<const float4*>(gathered + ((size_t)r * nf + k) * cap + i);
    float4* dst_q = reinterpret_cast<float4*>(lvc + base + i);
    float4 q[6];
#pragma unroll
    for (int j = 0; j < 6; j++) q[j] = src[j];
#pragma unroll
    for (int j = 0; j < 6; j++) dst_q[j] = q[j];
}
// Sending side of the batched exchange: the first min(count, cap) vertices of nf sets and their count pairs into one contiguous
// send buffer of nf x cap vertices (grid.y = frame).  The padding behind a shard is not copied (nobody reads it).
__global__ void k_pack_shards(CompactBatch src, int cap, LightVertex* __restrict__ send, int* __restrict__ send_counts) {
    const int k = blockIdx.y;
    const int n = src.counts[k][0];
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    if (i == 0) { send_counts[2 * k] = n; send_counts[2 * k + 1] = src.counts[k][1]; }
    if (i >= min(n, cap)) return;
    const float4* in = reinterpret_cast<const float4*>(src.lvc[k] + i);
    float4* out = reinterpret_cast<float4*>(send + (size_t)k * cap + i);
    float4 q[6];
#pragma unroll
    for (int j = 0; j < 6; j++) q[j] = in[j];
#pragma unroll
    for (int j = 0; j < 6; j++) out[j] = q[j];
}

// film exchange of a sharded job: the 8-row bands of rank `rank` (band b with b % world == rank) packed contiguously / unpacked
__global__ void k_pack_bands(const float4* __restrict__ accum, int width, int height, int rank, int world, float4* __restrict__ packed, int unpack_all) {
    // unpack_all == 0: accum -> packed (own bands, band-major); != 0: packed (world x bands_per_rank x 8 x width) -> accum (every band)
    const int bands = (height + 7) / 8, per_rank = (bands + world - 1) / world;
    const size_t band_px = (size_t)8 * width;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!unpack_all) {
        if (t >= (size_t)per_rank * band_px) return;
        const int k = (int)(t / band_px), b = rank + k * world;
        const size_t in_band = t % band_px;
        const int y = b * 8 + (int)(in_band / width), x = (int)(in_band % width);
        packed[t] = (b < bands && y < height) ? accum[(size_t)y * width + x] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        if (t >= (size_t)world * per_rank * band_px) return;
        const int r = (int)(t / ((size_t)per_rank * band_px));
        const size_t tr = t % ((size_t)per_rank * band_px);
        const int k = (int)(tr / band_px), b = r + k * world;
        const size_t in_band = tr % band_px;
        const int y = b * 8 + (int)(in_band / width), x = (int)(in_band % width);
        if (b < bands && y < height) reinterpret_cast<float4*>(const_cast<float4*>(accum))[(size_t)y * width + x] = packed[t];
    }
}

int light_trace_blocks(const KParams& p, int max_blocks) {
    const long long cores = (long long)p.core_count * (p.n_lframes > 0 ? p.n_lframes : 1);
    long long blocks = (cores + BLOCK - 1) / BLOCK;
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    return (int)blocks;
}
void launch_light_trace(const KParams& p, int variant, int max_blocks, hipStream_t s) {   // variants as launch_spcbpt
    const int blocks = light_trace_blocks(p, max_blocks);   // p.work_counter (the core queue head) must have been zeroed on `s`
    if (blocks <= 0) return;
    if (variant == 1) {   // (the reference's own evaluation of Gamma / Q, counted as such: launch_spcbpt)
        KParams q = p;
        q.gamma_q = nullptr;
        hipLaunchKernelGGL((k_light_trace<true, false>), dim3(blocks), dim3(BLOCK), 0, s, q);
    }
    else if (variant == 2) hipLaunchKernelGGL((k_light_trace<true, true>), dim3(blocks), dim3(BLOCK), 0, s, p);
    else hipLaunchKernelGGL((k_light_trace<false, true>), dim3(blocks), dim3(BLOCK), 0, s, p);
}
void launch_lvc_compact(const LightVertex* scratch, const int* core_counts, const int* core_offsets, int core_count, int core_padding,
                        LightVertex* lvc, uint32_t* keys, uint32_t* vals, float* weights, int* sampler_counts, int capacity, uint32_t* overflow,
                        hipStream_t s) {
    const long long total = (long long)core_count * core_padding;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(k_lvc_compact, dim3(blocks), dim3(256), 0, s, scratch, core_counts, core_offsets, core_count,
                       core_padding, lvc, keys, vals, weights, sampler_counts, capacity, overflow);
}
void launch_lvc_compact_batch(const LightVertex* scratch, const int* core_counts, const int* core_offsets, const int* path_counts, int core_count,
                              int core_padding, int n, const CompactBatch& dst, int capacity, uint32_t* overflow, hipStream_t s) {
    const long long total = (long long)core_count * core_padding;
    hipLaunchKernelGGL(k_lvc_compact_batch, dim3((unsigned)((total + 255) / 256), (unsigned)n), dim3(256), 0, s, scratch, core_counts, core_offsets,
                       path_counts, core_count, core_padding, dst, capacity, overflow);
}
void launch_fill_keys(const LightVertex* lvc, int n, uint32_t* keys, uint32_t* vals, float* weights, int* sampler_counts, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fill_keys_from_lvc, dim3((n + 255) / 256), dim3(256), 0, s, lvc, n, keys, vals, weights, sampler_counts);
}
void launch_fill_keys_devcount(const LightVertex* lvc, int bound, uint32_t* keys, uint32_t* vals, float* weights, const int* sampler_counts, hipStream_t s) {
    if (bound <= 0) return;
    hipLaunchKernelGGL(k_fill_keys_devcount, dim3((bound + 255) / 256), dim3(256), 0, s, lvc, bound, keys, vals, weights, sampler_counts);
}
void launch_gather_compact(const LightVertex* gathered, const int* counts_all, int world, int cap, int lvc_capacity, const CompactBatch& dst, int nf,
                           int* overflow, hipStream_t s) {
    const int chunks = (cap + 255) / 256;
    hipLaunchKernelGGL(k_gather_compact, dim3((unsigned)(world * chunks), (unsigned)nf), dim3(256), 0, s, gathered, counts_all, world, cap, lvc_capacity, dst,
                       nf, overflow);
}
void launch_pack_shards(const CompactBatch& src, int nf, int cap, LightVertex* send, int* send_counts, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_shards, dim3((unsigned)((cap + 255) / 256), (unsigned)nf), dim3(256), 0, s, src, cap, send, send_counts);
}
void launch_pack_bands(float* accum, int width, int height, int rank, int world, float* packed, bool unpack_all, hipStream_t s) {
    const int bands = (height + 7) / 8, per_rank = (bands + world - 1) / world;
    const size_t n = (size_t)(unpack_all ? world : 1) * per_rank * 8 * width;
    if (n == 0) return;
    hipLaunchKernelGGL(k_pack_bands, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(accum), width, height, rank, world,
                       reinterpret_cast<float4*>(packed), unpack_all ? 1 : 0);
}

}  // namespace spc

// Context: light passes (single and batched) and the device sampler build (MyThrustOp::LVC_Process, cuda_thrust/device_thrust.cu:241-332)
// (part of the C ABI library: see capi_common.h for the map of its translation units)
#include "capi_common.h"

using namespace spc;

namespace spc {

// "light trace": k_light_trace into the padded scratch, then compaction into the deterministic (core, slot) order
int Context::launch_light(uint32_t frame) {
    if (!have_subspace) { error = "light trace needs a subspace tuple (spcbpt_set_subspace)"; return SPCBPT_ERR_STATE; }
    if (!d_scratch) {
        spcbpt_light_trace_params d = {100000, 52, 1, 0, 0, 1};
        int rc = set_light_trace(d);
        if (rc) return rc;
    }
    if (lvc_probe_needed) { int rcp = probe_lvc_capacity(); if (rcp) return rcp; }
    // lane: passes running ahead alternate between the light stream and a second one (context.h); everything else uses lane 0
    int lane = 0;
    if (light_ahead && !counting && getenv("SPCBPT_LIGHT_LANES") == nullptr) { light_toggle ^= 1; lane = light_toggle; }
    if (lane) { int rcb = ensure_lane_b(); if (rcb) return rcb; }
    hipStream_t ls = lane ? lstream_b : stream;
    LightVertex* scratch = lane ? b_scratch : d_scratch;
    int* core_counts = lane ? b_core_counts : d_core_counts;
    int* core_offsets = lane ? b_core_offsets : d_core_offsets;
    uint32_t* keys = lane ? b_keys : d_keys;
    uint32_t* vals = lane ? b_vals : d_vals;
    float* weights = lane ? b_weights : d_weights;
    kp.num_core = lt.num_core; kp.core_padding = lt.core_padding; kp.m_per_core = lt.m_per_core;
    kp.core_begin = lt.core_begin; kp.core_count = lt.core_count; kp.launch_frame = frame;
    kp.n_lframes = 0;   // one pass (a batched launch that failed half-way must not leave its mode behind)
    kp.lt_decorrelate = lt.decorrelate_bsdf_stream;
    kp.lvc_scratch = scratch; kp.core_counts = core_counts;
    int rc = 0;
    {   // traversal-stack spill area of this lane's light kernel
        const int entries = spill_entries_needed();
        kp.spill_entries = entries;
        // TravStack indexes the area by blockIdx.x * 256 + threadIdx.x of the grid launched (launch_light_trace)
        const size_t need = (((size_t)lt.core_count + 255) / 256 * 256) * (size_t)entries;
        uint32_t*& buf = lane ? b_spill : d_spill;
        size_t& cap = lane ? b_spill_capacity : spill_capacity;
        if (entries == 0) kp.spill = nullptr;
        else {
            if (need > cap) { dev_free(buf); HIP_TRY(this, dev_alloc(&buf, need)); cap = need; }
            kp.spill = buf;
        }
    }
    kp.counters = counting ? d_counters : nullptr;
    // write the set the eye pass is NOT reading; it was last read by the render launch before the previous one
    lset = (lset + 1) % n_sets;
    select_set(lset);
    if (ev_render_set[lset]) HIP_TRY(this, hipStreamWaitEvent(ls, ev_render[render_event_of[lset]], 0));
    // ... and a sampler build or an import copy of the set's previous contents may still be queued on `stream` (a set that was
    // built or imported but never rendered carries no fresh ev_render): the second lane does not run in `stream`'s order
    if (ls != stream) {
        if (ev_set_touched[lset]) HIP_TRY(this, hipStreamWaitEvent(ls, ev_set_stream[lset], 0));
    }
    if (ev_exch_set[lset]) HIP_TRY(this, hipStreamWaitEvent(ls, ev_exch[lset], 0));   // a gathered import of the set's previous contents (exchange stream)
    set_bound[lset] = -1;
    HIP_TRY(this, hipMemsetAsync(core_counts, 0, ((size_t)lt.core_count + 1) * sizeof(int), ls));
    HIP_TRY(this, hipMemsetAsync(d_sampler_counts, 0, 2 * sizeof(int), ls));
    kp.path_counter = d_sampler_counts + 1;
    // persistent grid of the light pass: at most light_blocks blocks pull cores from a queue (kernels.hip)
    kp.work_counter = d_work_counter + kMaxRender + lane;
    HIP_TRY(this, hipMemsetAsync(kp.work_counter, 0, sizeof(uint32_t), ls));
    if (light_blocks < 0) { const char* lb = getenv("SPCBPT_LIGHT_BLOCKS"); light_blocks = lb ? std::max(1, atoi(lb)) : std::max(1, num_cus); }
    // Thin or wide (round 6).  One block per CU is right for a pass that runs BESIDE eye kernels (light-ahead mode: few long-lived blocks
    // take least from them).  In the reference's loop form -- light pass, build, eye launch, sync, strictly in turn
    // (optixPathTracer.cpp:791-822) -- nothing else is on the GPU while the pass runs, and one wave per SIMD leaves it a chain of
    // dependent fetches: there the pass gets a lane per core (100 000 paths: 391 blocks; four per CU at most), SPCBPT_LIGHT_BLOCKS_WIDE.
    if (light_blocks_wide < 0) { const char* lb = getenv("SPCBPT_LIGHT_BLOCKS_WIDE"); light_blocks_wide = lb ? std::max(1, atoi(lb)) : std::max(1, 4 * num_cus); }
    const int light_grid = (!light_ahead && lane == 0) ? std::max(light_blocks, light_blocks_wide) : light_blocks;
    time_begin("light_trace", ls);
    launch_light_trace(kp, kernel_variant(), light_grid, ls);   // direction trees: the generic instantiation (no label caching)
    time_end();
    HIP_TRY(this, hipGetLastError());
    // compaction: exclusive scan of per-core counts (+1 sentinel gives the total) -> offsets
    time_begin("lvc_compact", ls);
    size_t tb = 0;
    HIP_TRY(this, hipcub::DeviceScan::ExclusiveSum(nullptr, tb, core_counts, core_offsets, lt.core_count + 1, ls));
    unsigned char* temp = nullptr;
    if (lane) {
        if (tb > b_temp_capacity) { dev_free(b_temp); HIP_TRY(this, dev_alloc(&b_temp, tb)); b_temp_capacity = tb; }
        temp = b_temp;
    } else {
        rc = ensure_temp(tb);
        if (rc) return rc;
        temp = d_temp;
    }
    HIP_TRY(this, hipcub::DeviceScan::ExclusiveSum(temp, tb, core_counts, core_offsets, lt.core_count + 1, ls));
    HIP_TRY(this, hipMemcpyAsync(d_sampler_counts, core_offsets + lt.core_count, sizeof(int), hipMemcpyDeviceToDevice, ls));
    launch_lvc_compact(scratch, core_counts, core_offsets, lt.core_count, lt.core_padding, d_lvc, keys, vals, weights,
                       d_sampler_counts, (int)std::min<size_t>(lvc_capacity, 0x7fffffff), d_diag + 2, ls);
    time_end();
    HIP_TRY(this, hipGetLastError());
    if (lane == 0) { keys_ready = true; keys_set = lset; }
    else if (keys_set == lset) keys_ready = false;   // the set was rewritten by the other lane: lane 0's keys no longer describe it
    lvc_count = -1;  // known on the device only until the next host read
    set_count_host[lset] = -1;
    // A new light pass means "build before you render", as in the reference's loop -- also with passes running ahead: a host loop
    // that launches a pass and forgets the build gets SPCBPT_ERR_STATE from its next eye launch, not last frame's sampler.  The
    // tables of set `eset` are in fact still intact while later passes fill OTHER sets of the ring; a host that means to render
    // from them once more (csrc/viewer.cpp: a speculative frame dropped and traced again) says so with spcbpt_reuse_sampler.
    have_sampler = false;
    // (vertex_count, path_count) to pinned host memory, inside the event: the sampler build reads them after waiting for
    // THIS pass only, not for whatever else has been queued on the stream since
    HIP_TRY(this, hipMemcpyAsync(h_light_counts + 2 * lset, d_sampler_counts, 2 * sizeof(int), hipMemcpyDeviceToHost, ls));
    HIP_TRY(this, hipEventRecord(ev_light[lset], ls));
    light_counts_valid[lset] = true;
    light_lane_of_set[lset] = lane;
    for (auto it = built_sets.begin(); it != built_sets.end();) it = (*it == lset) ? built_sets.erase(it) : it + 1;   // its sampler is gone
    if (!light_ahead) pending.clear();   // default: a sampler build always takes the latest light pass
    for (auto it = pending.begin(); it != pending.end();) it = (*it == lset) ? pending.erase(it) : it + 1;  // a set that comes round again unbuilt
    pending.push_back(lset);
    return 0;
}

// Batched light pass: see context.h.  Sets lset+1 .. lset+n receive the passes of launch frames first_frame .. first_frame+n-1 and
// queue up in `pending` like n calls of launch_light; every one of them is bit-identical to the pass launch_light would have traced
// (same seeds per core, same (core, slot) order after compaction).
int Context::launch_light_batch(uint32_t first_frame, int n) {
    if (!have_subspace) { error = "light trace needs a subspace tuple (spcbpt_set_subspace)"; return SPCBPT_ERR_STATE; }
    if (n < 1 || n > kMaxBatchFrames || n > n_sets - 2) { error = "launch_light_batch: 1 .. min(32, sets - 2) frames per batch"; return SPCBPT_ERR_INVALID_ARG; }
    if (!light_ahead) { error = "launch_light_batch: the passes queue up for build_sampler -- enable spcbpt_set_light_ahead first"; return SPCBPT_ERR_STATE; }
    if (!d_scratch) {
        spcbpt_light_trace_params d = {100000, 52, 1, 0, 0, 1};
        int rc = set_light_trace(d);
        if (rc) return rc;
    }
    if (lvc_probe_needed) { int rcp = probe_lvc_capacity(); if (rcp) return rcp; }
    int rc = ensure_lane_b();
    if (rc) return rc;
    hipStream_t ls = lstream_b;
    const size_t slots = (size_t)lt.core_count * lt.core_padding, cstride = (size_t)lt.core_count + 1;
    if ((size_t)n * slots > lb_scratch_capacity) {
        if (sync_all()) return SPCBPT_ERR_HIP;
        dev_free(lb_scratch); HIP_TRY(this, dev_alloc(&lb_scratch, (size_t)n * slots)); lb_scratch_capacity = (size_t)n * slots;
    }
    if ((size_t)n * cstride > lb_counts_capacity) {
        if (sync_all()) return SPCBPT_ERR_HIP;
        dev_free(lb_core_counts); dev_free(lb_core_offsets); dev_free(lb_path_counts);
        HIP_TRY(this, dev_alloc(&lb_core_counts, (size_t)n * cstride)); HIP_TRY(this, dev_alloc(&lb_core_offsets, (size_t)n * cstride));
        HIP_TRY(this, dev_alloc(&lb_path_counts, (size_t)kMaxBatchFrames));
        lb_counts_capacity = (size_t)n * cstride;
    }
    kp.num_core = lt.num_core; kp.core_padding = lt.core_padding; kp.m_per_core = lt.m_per_core;
    kp.core_begin = lt.core_begin; kp.core_count = lt.core_count; kp.launch_frame = first_frame;
    kp.lt_decorrelate = lt.decorrelate_bsdf_stream;
    kp.lvc_scratch = lb_scratch; kp.core_counts = lb_core_counts; kp.path_counter = lb_path_counts;
    kp.n_lframes = n;
    // a THIN grid: the batch runs beside the eye kernels of the frames before it and only has to be done before they are; few
    // long-lived blocks take less from them than many (bench scene, one GPU, 20 / 64 steps, ms per step: 16 blocks 5.71 / 5.67,
    // 24: 5.73 / 5.66, 32: 5.74 / 5.68, 48: 5.85 / 5.71, 64: 5.94 / 5.76 -- the 20-frame batch then takes 76 ms beside an eye launch
    // of 110).  SPCBPT_LIGHT_BATCH_BLOCKS fixes the number; by default it goes ...
    if (light_batch_blocks < 0) { const char* lb = getenv("SPCBPT_LIGHT_BATCH_BLOCKS"); light_batch_blocks = lb ? std::max(1, atoi(lb)) : 0; }
    int grid_cap = light_batch_blocks;
    if (grid_cap == 0) {
        // ... in proportion to the light paths per pixel of this context's share of the frame (kp.row_step: the band step of the last
        // eye launch), so that the batch stays shorter than the eye launch it runs beside: 640 blocks per (path / pixel), i.e. 32 for
        // 100 000 paths against 1920 x 1080 pixels -- or against an eighth of both
        const double px = std::max(1.0, (double)kp.width * kp.height / std::max(1, (int)kp.row_step));
        const double ratio = (double)lt.core_count * std::max(1, lt.m_per_core) / px;
        // (a rank's share of a sharded frame wants more lanes for the same ratio: its eye launches are short, and the chain of a
        // batch -- passes, then one exchange and build per frame -- has to fit under them: N = 8 simulation 0.80-0.81 ms per
        // rank-frame with 48 blocks, 0.83-0.87 with 20)
        // (round 4: 400 -> 480 blocks per (path / pixel).  With the eye kernel 13 % faster the 20 blocks of 400 finished a 20-frame batch in
        // 75 ms beside an eye launch of 83 -- and 16 blocks, too few, cost 10 %: the batch became the critical path.  24 keep a fifth in hand
        // at no measurable cost: 20 / 24 / 28 / 32 blocks 4.176 / 4.204 / 4.181 / 4.231 ms per step)
        // (round 5: 480 -> 640, i.e. 32.  The eye kernel is another 8 % faster and runs its traversal pass at a raised issue priority, under
        // which the light pass that shares its CUs is slower: with 24 blocks the batch for the NEXT eye launch took 101-113 ms beside an eye
        // launch of 116 (tools/timeline_long.sh), so that its sampler build -- 1.3 ms -- ran in the gap between two eye kernels instead of
        // under the first.  Steady-state ms per step, 24 / 32 / 40 / 48 / 64 blocks: 3.706 / 3.661 / 3.70-3.77 / 3.75 / 3.87)
        // (round 6, advisor: 640 x 0.048225 = 30.9 -> ceil gave 31, a grid no sweep had covered; the product is now rounded up to a
        // multiple of 8 blocks -- one per XCD -- which IS the swept 32 on the bench scene)
        grid_cap = (int)std::max(kp.row_step > 1 ? 48.0 : 16.0, std::min(256.0, 8.0 * std::ceil(80.0 * ratio)));
    }
    // (cores that trace many paths one after the other -- the reference's geometry -- are long jobs: never two of them per lane)
    if (lt.m_per_core >= 8) grid_cap = std::max(grid_cap, (int)(((long long)n * lt.core_count + 255) / 256));
    const int blocks = light_trace_blocks(kp, grid_cap);
    {   // traversal-stack spill area, indexed by blockIdx.x * 256 + threadIdx.x of the grid launched
        const int entries = spill_entries_needed();
        kp.spill_entries = entries;
        const size_t need = (size_t)blocks * 256 * (size_t)entries;
        if (entries == 0) kp.spill = nullptr;
        else {
            if (need > lb_spill_capacity) { if (sync_all()) return SPCBPT_ERR_HIP; dev_free(lb_spill); HIP_TRY(this, dev_alloc(&lb_spill, need)); lb_spill_capacity = need; }
            kp.spill = lb_spill;
        }
    }
    kp.counters = counting ? d_counters : nullptr;
    CompactBatch dst = {};
    int sets[kMaxBatchFrames];
    int waited_for = -1;
    for (int k = 0; k < n; k++) {   // what launch_light waits for before it rewrites a set, for every set of the batch
        const int s = (lset + 1 + k) % n_sets;
        sets[k] = s;
        if (ev_render_set[s] && render_event_of[s] != waited_for) {   // (the sets of one batched eye launch share one event)
            HIP_TRY(this, hipStreamWaitEvent(ls, ev_render[render_event_of[s]], 0));
            waited_for = render_event_of[s];
        }
        if (ev_set_touched[s]) HIP_TRY(this, hipStreamWaitEvent(ls, ev_set_stream[s], 0));
        if (ev_exch_set[s]) HIP_TRY(this, hipStreamWaitEvent(ls, ev_exch[s], 0));
        set_bound[s] = -1;
        dst.lvc[k] = set_lvc[s]; dst.counts[k] = set_counts[s];
    }
    HIP_TRY(this, hipMemsetAsync(lb_core_counts, 0, (size_t)n * cstride * sizeof(int), ls));
    HIP_TRY(this, hipMemsetAsync(lb_path_counts, 0, kMaxBatchFrames * sizeof(int), ls));
    kp.work_counter = d_work_counter + kMaxRender + 1;   // the second lane's queue head
    HIP_TRY(this, hipMemsetAsync(kp.work_counter, 0, sizeof(uint32_t), ls));
    time_begin("light_trace", ls);
    launch_light_trace(kp, kernel_variant(), grid_cap, ls);
    time_end();
    kp.n_lframes = 0;
    HIP_TRY(this, hipGetLastError());
    time_begin("lvc_compact", ls);
    size_t tb = 0;
    const int items = (int)((size_t)n * cstride);
    HIP_TRY(this, hipcub::DeviceScan::ExclusiveSum(nullptr, tb, lb_core_counts, lb_core_offsets, items, ls));
    if (tb > b_temp_capacity) { if (sync_all()) return SPCBPT_ERR_HIP; dev_free(b_temp); HIP_TRY(this, dev_alloc(&b_temp, tb)); b_temp_capacity = tb; }
    HIP_TRY(this, hipcub::DeviceScan::ExclusiveSum(b_temp, tb, lb_core_counts, lb_core_offsets, items, ls));
    launch_lvc_compact_batch(lb_scratch, lb_core_counts, lb_core_offsets, lb_path_counts, lt.core_count, lt.core_padding, n, dst,
                             (int)std::min<size_t>(lvc_capacity, 0x7fffffff), d_diag + 2, ls);
    time_end();
    HIP_TRY(this, hipGetLastError());
    // (vertex_count, path_count) of the sets to pinned host memory: the sets are consecutive modulo n_sets -> at most two ranges
    {
        const int s0 = sets[0], first = std::min(n, n_sets - s0);
        HIP_TRY(this, hipMemcpyAsync(h_light_counts + 2 * s0, d_set_counts_all + 2 * s0, (size_t)first * 2 * sizeof(int), hipMemcpyDeviceToHost, ls));
        if (first < n) HIP_TRY(this, hipMemcpyAsync(h_light_counts, d_set_counts_all, (size_t)(n - first) * 2 * sizeof(int), hipMemcpyDeviceToHost, ls));
    }
    for (int k = 0; k < n; k++) {
        const int s = sets[k];
        HIP_TRY(this, hipEventRecord(ev_light[s], ls));
        if (keys_set == s) keys_ready = false;   // lane 0's keys no longer describe the set
        set_count_host[s] = -1;
        light_counts_valid[s] = true;
        light_lane_of_set[s] = 1;
        for (auto it = built_sets.begin(); it != built_sets.end();) it = (*it == s) ? built_sets.erase(it) : it + 1;
        for (auto it = pending.begin(); it != pending.end();) it = (*it == s) ? pending.erase(it) : it + 1;
        pending.push_back(s);
    }
    lset = sets[n - 1];
    select_set(lset);
    lvc_count = -1;
    have_sampler = false;
    return 0;
}

int Context::fetch_counts_of(int set) {
    int h[2] = {0, 0};
    if (light_counts_valid[set] && light_lane_of_set[set] != 0) HIP_TRY(this, hipEventSynchronize(ev_light[set]));   // traced on the second lane: `stream` does not order it
    HIP_TRY(this, hipMemcpyAsync(h, set_counts[set], sizeof(h), hipMemcpyDeviceToHost, stream));
    HIP_TRY(this, hipStreamSynchronize(stream));
    lvc_count = h[0];
    path_count = h[1];
    return 0;
}
int Context::fetch_counts() { return fetch_counts_of(lset); }

// LVC_Process on the device, for the oldest light pass (or imported cache) that has no sampler yet
int Context::build_sampler() {
    if (!d_lvc) { error = "build_sampler: no light-vertex cache (run \"light trace\" or spcbpt_lvc_import first)"; return SPCBPT_ERR_STATE; }
    const int bset = build_set();
    select_set(bset);
    int rc = 0;
    if (light_lane_of_set[bset] != 0 && light_counts_valid[bset]) HIP_TRY(this, hipStreamWaitEvent(stream, ev_light[bset], 0));   // traced on the second lane
    // the radix sort needs its item count on the host: an import told it, or the light pass left it in pinned memory (wait for
    // that pass's event), or -- a cache written some other way -- one 8-byte readback
    const bool dev_count = set_bound[bset] >= 0;   // gathered import: totals on the device, build over the upper bound
    if (dev_count && ev_exch_set[bset]) HIP_TRY(this, hipStreamWaitEvent(stream, ev_exch[bset], 0));
    bool count_known = set_count_host[bset] >= 0 || dev_count;
    // the counting build takes its item count on the device: a light pass's (vertex_count, path_count) need not reach the host first --
    // the reference-shaped loop (light pass -> build -> eye launch, one sync per frame) then runs without a host wait in the middle
    const bool lazy = counting_build && !count_known && light_counts_valid[bset];
    if (dev_count) { lvc_count = set_bound[bset]; path_count = -1; }
    else if (count_known) lvc_count = set_count_host[bset];
    else if (lazy) { lvc_count = (int)std::min<size_t>(lvc_capacity, 0x7fffffff); path_count = -1; count_known = true; }
    else if (light_counts_valid[bset]) {
        HIP_TRY(this, hipEventSynchronize(ev_light[bset]));
        lvc_count = h_light_counts[2 * bset]; path_count = h_light_counts[2 * bset + 1];
        count_known = true;
    } else rc = fetch_counts_of(bset);
    if (rc) { select_set(lset); return rc; }
    const int n = lvc_count;
    time_begin("sampler_build");
    if (counting_build) {
        // one stable counting sort over the 10-bit subspace ids: four launches (kernels.hip).  The path count is taken on the way
        // unless the light pass (or the gathered import) has left it in the set already.
        if (!d_hist) HIP_TRY(this, dev_alloc(&d_hist, sampler_build_hist_ints()));
        const bool count_paths = !dev_count && !lazy && !(keys_ready && keys_set == bset);   // (a light pass has left the path count in its set)
        if (count_paths) HIP_TRY(this, hipMemsetAsync(d_sampler_counts + 1, 0, sizeof(int), stream));
        launch_sampler_build(d_lvc, n, (dev_count || lazy) ? d_sampler_counts : nullptr, d_keys, d_weights, d_hist, count_paths ? d_sampler_counts + 1 : nullptr,
                             d_subspace, d_vals2, d_wsorted, d_cmfs, set_lvc_sorted[bset], set_guide[bset], stream);
        keys_ready = false;
    } else {
    if (dev_count) {
        launch_fill_keys_devcount(d_lvc, n, d_keys, d_vals, d_weights, d_sampler_counts, stream);
    } else if (!(keys_ready && keys_set == bset)) {
        HIP_TRY(this, hipMemsetAsync(d_sampler_counts + 1, 0, sizeof(int), stream));
        launch_fill_keys(d_lvc, n, d_keys, d_vals, d_weights, d_sampler_counts, stream);
    }
    keys_ready = false;   // the sort below consumes the keys
    HIP_TRY(this, hipMemsetAsync(d_subspace, 0, SPCBPT_NUM_SUBSPACE * sizeof(DSubspace), stream));
    if (n > 0) {
        size_t tb = 0, tb2 = 0;
        HIP_TRY(this, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0, 10, stream));
        HIP_TRY(this, hipcub::DeviceScan::InclusiveSum(nullptr, tb2, d_wsorted, d_prefix, n, stream));
        rc = ensure_temp(std::max(tb, tb2));
        if (rc) { select_set(lset); return rc; }
        HIP_TRY(this, hipcub::DeviceRadixSort::SortPairs(d_temp, tb, d_keys, d_keys2, d_vals, d_vals2, n, 0, 10, stream));
        launch_subspace_ranges(d_keys2, d_sampler_counts, d_subspace, n, stream);
        launch_gather_weights(d_weights, d_vals2, d_sampler_counts, d_wsorted, n, stream);
        HIP_TRY(this, hipcub::DeviceScan::InclusiveSum(d_temp, tb2, d_wsorted, d_prefix, n, stream));
        launch_cmf(d_prefix, d_keys2, d_sampler_counts, d_subspace, d_cmfs, n, stream);
        launch_lvc_sorted_copy(d_lvc, d_vals2, d_sampler_counts, set_lvc_sorted[bset], n, stream);
        launch_sampler_guide(d_subspace, d_cmfs, set_guide[bset], stream);
    }
    }
    time_end();
    HIP_TRY(this, hipGetLastError());
    // the host copy of (vertex_count, path_count): a second readback, skipped when the count came with an import (a sharded
    // job must not wait for the light stream here -- the next frame's light pass is already queued on it)
    if (!count_known && fetch_counts_of(bset)) { select_set(lset); return SPCBPT_ERR_HIP; }
    eset = bset;
    HIP_TRY(this, hipEventRecord(ev_sampler[eset], stream));
    ev_sampler_set[eset] = true;
    HIP_TRY(this, hipEventRecord(ev_set_stream[eset], stream));
    ev_set_touched[eset] = true;
    have_sampler = true;
    for (auto it = built_sets.begin(); it != built_sets.end();) it = (*it == bset) ? built_sets.erase(it) : it + 1;
    built_sets.push_back(bset);
    while ((int)built_sets.size() > kMaxBatchFrames) built_sets.pop_front();
    if (!pending.empty() && pending.front() == bset) pending.pop_front();
    if (dev_count || lazy) lvc_count = -1;   // the host does not know it (fetch_counts brings it when somebody asks)
    select_set(lset);   // the members name the latest light pass's set again
    return 0;
}

void Context::free_batch_build_scratch() {
    dev_free(sbb_keys); dev_free(sbb_weights); dev_free(sbb_wsorted); dev_free(sbb_hist);
    sbb_keys = nullptr; sbb_weights = nullptr; sbb_wsorted = nullptr; sbb_hist = nullptr;
    sbb_frames = 0; sbb_capacity = 0;
}
size_t Context::sbb_debug_limit() const {
    const char* e = getenv("SPCBPT_DEBUG_BATCH_SCRATCH_LIMIT");
    return e ? (size_t)strtoull(e, nullptr, 10) : ~(size_t)0;
}

// LVC_Process for the n OLDEST light passes that have no sampler yet, as ONE set of four launches (kernels.hip: SamplerBuildBatch).
// The tables are those of n build_sampler calls -- the same kernels with the frame in blockIdx.y -- and the sets end up in the same
// state; what goes is n - 1 times the four dependent launches (0.12 ms per build: 2.4 ms in front of a 20-frame eye launch that
// cannot start before the last of them).  Gathered imports (totals on the device) are built over their upper bound, as build_sampler
// does.  Falls back to n single builds for the radix-sort form and for a cache whose counts the host would have to read back.
int Context::build_sampler_batch(int n) {
    if (n < 1 || n > kMaxBatchFrames) { error = "build_sampler_batch: 1 .. 32 builds per call"; return SPCBPT_ERR_INVALID_ARG; }
    bool plain = counting_build && n > 1 && d_lvc && (int)pending.size() >= n;
    for (int k = 0; plain && k < n; k++) {
        const int b = pending[(size_t)k];
        if (set_bound[b] < 0 && !(set_count_host[b] >= 0 || light_counts_valid[b])) plain = false;   // counts the host would have to read back
    }
    if (!plain) {
        for (int k = 0; k < n; k++) { const int rc = build_sampler(); if (rc) return rc; }
        return 0;
    }
    // Scratch of the batch: per frame what d_keys / d_weights / d_wsorted / d_hist are to one build -- sized by the builds of THIS call
    // (n frames x the largest item bound among them: a host-known count, a gathered import's bound, or the set capacity for a pass
    // whose count only the device knows), not by the widest batch and the padded capacity the context could ever see: with an
    // uncalibrated cache (core_count x padding) 32 x capacity x 16 B would be gigabytes.  It only grows; spcbpt_lvc_set_capacity
    // and leaving light-ahead mode free it.  If the device cannot hold it the builds run one by one (build_sampler's own scratch).
    size_t stride = 1;
    for (int k = 0; k < n; k++) {
        const int b = pending[(size_t)k];
        const int count = set_bound[b] >= 0 ? set_bound[b] : set_count_host[b];
        stride = std::max(stride, count < 0 ? lvc_capacity : std::min((size_t)count, lvc_capacity));
    }
    stride = (stride + 4095) / 4096 * 4096;
    if (!sbb_keys || sbb_frames < n || sbb_capacity < stride) {
        const int frames = std::max(n, sbb_frames);
        const size_t cap = std::max(stride, sbb_capacity);
        const size_t limit = sbb_debug_limit();   // tests: pretend the device refuses more than this many bytes of batch scratch
        const size_t bytes = (size_t)frames * cap * (sizeof(uint32_t) + sizeof(float) + sizeof(double));
        // (round 6, advisor) A size the device has refused is not asked for again until the capacity or the mode changes -- every call used
        // to repeat the device-wide wait, four hipMallocs and the failure path -- and the scratch that exists is kept while the larger one
        // is tried: a later, smaller batch still fits it.  The first refusal is reported once on stderr; spcbpt_get_pipeline_state's
        // callers see the count in spcbpt_debug_get("sbb_fallbacks").
        bool ok = bytes <= limit && (sbb_refused_bytes == 0 || bytes < sbb_refused_bytes);
        uint32_t* nk = nullptr; float* nw = nullptr; double* ns = nullptr; int* nh = nullptr;
        if (ok) {
            if (sync_all()) return SPCBPT_ERR_HIP;
            ok = dev_alloc(&nk, (size_t)frames * cap) == hipSuccess;
            ok = ok && dev_alloc(&nw, (size_t)frames * cap) == hipSuccess;
            ok = ok && dev_alloc(&ns, (size_t)frames * cap) == hipSuccess;
            ok = ok && dev_alloc(&nh, (size_t)frames * sampler_build_hist_ints()) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); dev_free(nk); dev_free(nw); dev_free(ns); dev_free(nh); }   // (an allocation failure is sticky in hipGetLastError only)
        }
        if (!ok) {
            if (sbb_refused_bytes == 0 || bytes < sbb_refused_bytes) {
                if (sbb_refused_bytes == 0)
                    fprintf(stderr, "spcbpt: no room for %zu bytes of batched sampler-build scratch (%d frames x %zu vertices): building one by one\n", bytes, frames, cap);
                sbb_refused_bytes = bytes;
            }
            sbb_fallbacks++;
            for (int k = 0; k < n; k++) { const int rc = build_sampler(); if (rc) return rc; }
            return 0;
        }
        free_batch_build_scratch();
        sbb_keys = nk; sbb_weights = nw; sbb_wsorted = ns; sbb_hist = nh;
        sbb_frames = frames; sbb_capacity = cap;
    }
    SamplerBuildBatch B = {};
    B.keys = sbb_keys; B.weights = sbb_weights; B.wsorted = sbb_wsorted; B.hist = sbb_hist; B.item_stride = sbb_capacity;
    int sets[kMaxBatchFrames];
    for (int k = 0; k < n; k++) {
        const int b = pending[(size_t)k];
        sets[k] = b;
        if (light_lane_of_set[b] != 0 && light_counts_valid[b]) HIP_TRY(this, hipStreamWaitEvent(stream, ev_light[b], 0));   // traced on the second lane
        const bool dev_count = set_bound[b] >= 0;   // gathered import: totals (and the path count) on the device, build over the upper bound
        if (dev_count && ev_exch_set[b]) HIP_TRY(this, hipStreamWaitEvent(stream, ev_exch[b], 0));
        int count = dev_count ? set_bound[b] : set_count_host[b];
        const bool lazy = count < 0;   // a light pass's count: read on the device (no host wait for the pass)
        if (lazy) count = (int)std::min<size_t>(lvc_capacity, 0x7fffffff);
        // the path count: a light pass's compaction (and a gathered import) has left it in the set; only a cache that came some other way
        // (spcbpt_lvc_import) has it counted by the build, as build_sampler does
        const bool count_paths = !dev_count && !light_counts_valid[b];
        B.lvc[k] = set_lvc[b]; B.n_host[k] = count; B.n_dev[k] = (dev_count || lazy) ? set_counts[b] : nullptr; B.path_count[k] = count_paths ? set_counts[b] + 1 : nullptr;
        B.sub[k] = set_subspace[b]; B.jump[k] = set_vals2[b]; B.cmfs[k] = set_cmfs[b]; B.lvc_sorted[k] = set_lvc_sorted[b]; B.guide[k] = set_guide[b];
        if (count_paths) HIP_TRY(this, hipMemsetAsync(set_counts[b] + 1, 0, sizeof(int), stream));
    }
    time_begin("sampler_build");
    launch_sampler_build_batch(B, n, stream);
    time_end();
    HIP_TRY(this, hipGetLastError());
    keys_ready = false;
    for (int k = 0; k < n; k++) {
        const int b = sets[k];
        eset = b;
        HIP_TRY(this, hipEventRecord(ev_sampler[b], stream));
        ev_sampler_set[b] = true;
        HIP_TRY(this, hipEventRecord(ev_set_stream[b], stream));
        ev_set_touched[b] = true;
        for (auto it = built_sets.begin(); it != built_sets.end();) it = (*it == b) ? built_sets.erase(it) : it + 1;
        built_sets.push_back(b);
        while ((int)built_sets.size() > kMaxBatchFrames) built_sets.pop_front();
        if (!pending.empty() && pending.front() == b) pending.pop_front();
    }
    have_sampler = true;
    lvc_count = B.n_dev[n - 1] ? -1 : B.n_host[n - 1];   // the members describe the set built last, as after build_sampler (-1: only the device knows)
    select_set(lset);
    return 0;
}

}  // namespace spc

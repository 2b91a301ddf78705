// Context: eye / pt launches (single, deferred, batched) and the film merges
// (part of the C ABI library: see capi_common.h for the map of its translation units)
#include "capi_common.h"

using namespace spc;

namespace spc {

int Context::launch_render(const char* name, bool spcbpt_alg, uint32_t frame, int r0, int r1, int rs, bool full_mis, bool defer_merge) {
    if (deferred.active) { error = "a deferred frame is outstanding: spcbpt_merge_deferred(ctx, keep) first"; return SPCBPT_ERR_STATE; }
    if (defer_merge && (full_mis || counting)) { error = "launch_deferred: plain \"pt\" / \"SPCBPT_eye\" launches only"; return SPCBPT_ERR_INVALID_ARG; }
    if (!d_accum) { error = "render before spcbpt_resize"; return SPCBPT_ERR_STATE; }
    if (!have_camera) { error = "render before spcbpt_set_camera"; return SPCBPT_ERR_STATE; }
    if (spcbpt_alg && (!have_sampler || !have_subspace)) { error = "SPCBPT_eye needs a subspace tuple and a built sampler"; return SPCBPT_ERR_STATE; }
    if (rs < 1) rs = 1;
    if (r0 < 0 || (r0 % 8) != 0) { error = "row_begin must be a non-negative multiple of 8 (8-row bands)"; return SPCBPT_ERR_INVALID_ARG; }
    kp.subframe = frame; kp.row_begin = r0; kp.row_end = std::min(r1, (int)kp.height); kp.row_step = rs;
    kp.counters = counting ? d_counters : nullptr;
    rk = (rk + 1) % n_render;   // consecutive render launches rotate through the render streams (see context.h)
    rstream = rstreams[rk];
    kp.result = d_result[rk];
    if (spcbpt_alg) {
        // the sampler tables this launch reads (set `eset`) were built on `stream`
        kp.lvc = set_lvc[eset]; kp.lvc_sorted = set_lvc_sorted[eset]; kp.subspace = set_subspace[eset]; kp.cmfs = set_cmfs[eset]; kp.guide = set_guide[eset];
        kp.jump = reinterpret_cast<const int32_t*>(set_vals2[eset]); kp.sampler_counts = set_counts[eset];
        if (rstream != stream && ev_sampler_set[eset]) HIP_TRY(this, hipStreamWaitEvent(rstream, ev_sampler[eset], 0));
    }
    int rc = ensure_spill((size_t)render_thread_count(kp), true);
    if (rc) return rc;
    if (full_mis && kp.scene.env.valid) { error = "SPCBPT_no_rmis: not with an environment map (the full-path weights of cuProg.h:901-1105 know area lights only)"; return SPCBPT_ERR_STATE; }
    if (full_mis) {   // "SPCBPT_no_rmis": a plain one-lane-per-pixel launch over the same sampler tables
        time_begin(name, rstream);
        launch_spcbpt_no_rmis(kp, rstream);
        time_end();
        HIP_TRY(this, hipGetLastError());
        render_event_of[eset] = eset;
        HIP_TRY(this, hipEventRecord(ev_render[eset], rstream));
        ev_render_set[eset] = true;
        return finish_frame();
    }
    if (spcbpt_alg) {
        kp.n_tiles = (uint32_t)render_tile_count(kp);
        kp.work_counter = d_work_counter + rk;
        HIP_TRY(this, hipMemsetAsync(d_work_counter + rk, 0, sizeof(uint32_t), rstream));
        const int generic = kernel_variant();
        if (!blocks_per_cu[generic]) {
            blocks_per_cu[generic] = spcbpt_blocks_per_cu(generic, false, kp.scene.general != 0);
            // developer knob (occupancy experiments): fewer resident blocks per CU than the kernel's resources allow
            if (const char* e = getenv("SPCBPT_BLOCKS_PER_CU")) { const int v = atoi(e); if (v >= 1 && v < blocks_per_cu[generic]) blocks_per_cu[generic] = v; }
            if (const char* e = getenv("SPCBPT_TILES_PER_WAVE")) tiles_per_wave = std::max(1, atoi(e));
            if (const char* e = getenv("SPCBPT_GRID_PERCENT")) grid_percent = std::max(1, std::min(100, atoi(e)));   // else adaptive
        }
    }
    time_begin(name, rstream);
    if (spcbpt_alg) {
        // Persistent grid.  With one render stream the kernel takes every resident block slot.  With several it takes 94 % of
        // them: a persistent block never yields, so a full grid leaves the next frame's light pass (and through the host's
        // wait for its vertex count, the next eye launch) nothing to run on until whole blocks have drained; with a tenth of the
        // slots free the light pass runs at once and the two eye kernels share the machine from the start.  Measured on the
        // bench scene, two streams, before the light pass ran ahead: 248 -> 259.5 Mpaths/s at 90 %, 259 at 84 %, 257.5 at 75 %; with
        // the final host loop 259.4 at 100 %, 254 at 97 %, 263.5 at 94 %, 264 at 90 % -- and the kernel by itself takes 7.95 /
        // 8.15 / 8.19 / 8.43 ms at those shares, so 94 % it is.  A policy that looks whether the previous eye kernel is still running does
        // not work: by the time the host has the vertex count it waited for, that kernel has drained.
        // SPCBPT_GRID_PERCENT fixes the share; SPCBPT_TILES_PER_WAVE bounds the waves by the tile count (experiments).
        const int generic = kernel_variant();
        int max_blocks = num_cus * blocks_per_cu[generic];
        if (tiles_per_wave > 1) max_blocks = std::max(1, std::min(max_blocks, (int)(kp.n_tiles / (uint32_t)(4 * tiles_per_wave))));
        const int percent = grid_percent > 0 ? grid_percent : (n_render > 1 ? 94 : 100);
        if (percent < 100) max_blocks = std::max(1, max_blocks * percent / 100);
        launch_spcbpt(kp, generic, max_blocks, rstream);
    }
    else launch_pt(kp, counting, rstream);
    time_end();
    HIP_TRY(this, hipGetLastError());
    if (spcbpt_alg) {
        render_event_of[eset] = eset;
        HIP_TRY(this, hipEventRecord(ev_render[eset], rstream));
        ev_render_set[eset] = true;
    }
    if (defer_merge) {
        deferred.active = true; deferred.rk = rk; deferred.subframe = kp.subframe; deferred.result = kp.result;
        deferred.row_begin = kp.row_begin; deferred.row_end = kp.row_end; deferred.row_step = kp.row_step;
        return 0;
    }
    return finish_frame();
}

// The film merge of the deferred frame, now (keep) or never.  Dropping costs nothing but the kernel time already spent: the
// render kernel wrote its own `result` buffer only.
int Context::merge_deferred(bool keep) {
    if (!deferred.active) { error = "merge_deferred: no deferred frame"; return SPCBPT_ERR_STATE; }
    deferred.active = false;
    if (!keep) return 0;
    rk = deferred.rk;
    rstream = rstreams[rk];
    kp.subframe = deferred.subframe; kp.result = deferred.result;
    kp.row_begin = deferred.row_begin; kp.row_end = deferred.row_end; kp.row_step = deferred.row_step;
    return finish_frame();
}
// Host wait for the last film merge only (the frame to be displayed), not for work queued behind it (the next frame's light
// pass, sampler build and speculative eye launch).
int Context::sync_film() {
    if (last_merge_k >= 0 && ev_merge_set[last_merge_k]) HIP_TRY(this, hipEventSynchronize(ev_merge[last_merge_k]));
    return check_diag();
}

int Context::launch_eye_batch(int n, const uint32_t* subframes, int r0, int r1, int rs) {
    if (deferred.active) { error = "a deferred frame is outstanding: spcbpt_merge_deferred(ctx, keep) first"; return SPCBPT_ERR_STATE; }
    if (!d_accum) { error = "render before spcbpt_resize"; return SPCBPT_ERR_STATE; }
    if (!have_camera) { error = "render before spcbpt_set_camera"; return SPCBPT_ERR_STATE; }
    if (!have_subspace) { error = "SPCBPT_eye needs a subspace tuple and a built sampler"; return SPCBPT_ERR_STATE; }
    if (n < 1 || n > kMaxBatchFrames || !subframes) { error = "launch_eye_batch: 1..32 frames"; return SPCBPT_ERR_INVALID_ARG; }
    if (n > (int)built_sets.size()) { error = "launch_eye_batch: fewer samplers have been built (and are still intact) than frames were asked for"; return SPCBPT_ERR_STATE; }
    // (SPCBPT_EYE_BATCH at spcbpt_create only sizes the ring of buffer sets so that batches, light passes ahead and builds do not
    // wait for each other; correctness rests on the per-set events and on `built_sets` naming intact samplers)
    if (counting) { error = "launch_eye_batch: not with event counters enabled (count with spcbpt_launch per frame)"; return SPCBPT_ERR_STATE; }
    if (tree_has_direction) { error = "launch_eye_batch: the batched kernel caches vertex labels, which needs classifier trees without direction nodes (use spcbpt_launch per frame)"; return SPCBPT_ERR_STATE; }
    if (kp.width >= 65536u || kp.height >= 65536u) { error = "launch_eye_batch: image too large"; return SPCBPT_ERR_INVALID_ARG; }
    if (rs < 1) rs = 1;
    if (r0 < 0 || (r0 % 8) != 0) { error = "row_begin must be a non-negative multiple of 8 (8-row bands)"; return SPCBPT_ERR_INVALID_ARG; }
    kp.row_begin = r0; kp.row_end = std::min(r1, (int)kp.height); kp.row_step = rs;
    kp.counters = nullptr;
    rk = (rk + 1) % n_render;
    rstream = rstreams[rk];
    const size_t px = (size_t)kp.width * kp.height;
    if (!h_frames) HIP_TRY(this, hipHostMalloc(reinterpret_cast<void**>(&h_frames), sizeof(FrameDesc) * kMaxRender * kDescRing * kMaxBatchFrames));
    if (!d_frames[rk]) HIP_TRY(this, hipMalloc(reinterpret_cast<void**>(&d_frames[rk]), sizeof(FrameDesc) * kMaxBatchFrames));
    // the descriptors travel through a small ring of pinned slots: the host must not wait for the previous batch of this stream
    // (it would stop launching the light passes of the batches after it), only for the upload that used this slot 4 batches ago
    const int gen = desc_gen[rk]++ % kDescRing;
    FrameDesc* hf = h_frames + ((size_t)rk * kDescRing + gen) * kMaxBatchFrames;
    if (ev_desc[rk][gen]) HIP_TRY(this, hipEventSynchronize(ev_desc[rk][gen]));
    else HIP_TRY(this, hipEventCreateWithFlags(&ev_desc[rk][gen], hipEventDisableTiming));
    int sets[kMaxBatchFrames];
    for (int k = 0; k < n; k++) {
        const int e = built_sets[built_sets.size() - (size_t)n + (size_t)k];   // oldest of the last n first
        sets[k] = e;
        if (!d_result_b[rk][k]) HIP_TRY(this, dev_alloc(&d_result_b[rk][k], px * 4));
        hf[k].lvc = set_lvc[e]; hf[k].lvc_sorted = set_lvc_sorted[e]; hf[k].subspace = set_subspace[e]; hf[k].cmfs = set_cmfs[e]; hf[k].guide = set_guide[e];
        hf[k].sampler_counts = set_counts[e];
        hf[k].result = d_result_b[rk][k]; hf[k].subframe = subframes[k];
        if (rstream != stream && ev_sampler_set[e]) HIP_TRY(this, hipStreamWaitEvent(rstream, ev_sampler[e], 0));
    }
    HIP_TRY(this, hipMemcpyAsync(d_frames[rk], hf, sizeof(FrameDesc) * (size_t)n, hipMemcpyHostToDevice, rstream));
    HIP_TRY(this, hipEventRecord(ev_desc[rk][gen], rstream));
    kp.n_tiles = (uint32_t)render_tile_count(kp);
    kp.frames = d_frames[rk]; kp.n_frames = (uint32_t)n;
    kp.work_counter = d_work_counter + rk;
    kp.result = nullptr; kp.subframe = subframes[0];
    HIP_TRY(this, hipMemsetAsync(d_work_counter + rk, 0, sizeof(uint32_t), rstream));
    if (!blocks_per_cu_batch) blocks_per_cu_batch = spcbpt_blocks_per_cu(0, true, kp.scene.general != 0);
    int max_blocks = num_cus * blocks_per_cu_batch;
    // a batch kernel runs for tens of milliseconds: the light passes of the batches after it need block slots meanwhile -- few,
    // since they run as a thin grid (launch_light_batch): 97 % (64 steps on one GPU: 5.76 ms per step at 94 %, 5.69 at 97, 5.67 at 100;
    // a rank's share of a sharded frame is indifferent: 0.81-0.82 ms per rank-frame at N = 8 with all three)
    const int percent = grid_percent > 0 ? grid_percent : 97;
    if (percent < 100) max_blocks = std::max(1, max_blocks * percent / 100);
    // the spill area is indexed by the thread of the grid ACTUALLY launched: n frames' tiles, capped by the resident slots
    // (sizing it for one frame's tiles let the blocks beyond one frame's share write past its end whenever that share was below max_blocks)
    int rc = ensure_spill((size_t)spcbpt_batch_blocks(kp, max_blocks) * (size_t)spcbpt_block_threads(), true);
    if (rc) return rc;
    time_begin("spcbpt_render", rstream);
    launch_spcbpt_batch(kp, max_blocks, rstream);
    time_end();
    HIP_TRY(this, hipGetLastError());
    HIP_TRY(this, hipEventRecord(ev_render[sets[n - 1]], rstream));   // ONE event for the sets of the batch (context.h: render_event_of)
    for (int k = 0; k < n; k++) { render_event_of[sets[k]] = sets[n - 1]; ev_render_set[sets[k]] = true; }
    eset = sets[n - 1];
    // the frames' merges, in frame order, after the previous launch's merge
    if (last_merge_k >= 0 && last_merge_k != rk && rstreams[last_merge_k] != rstream) HIP_TRY(this, hipStreamWaitEvent(rstream, ev_merge[last_merge_k], 0));
    {   // ... as one pass over the pixels (kernels.hip: k_film_merge_batch -- the operations of n merges, per pixel in frame order)
        MergeBatch mb = {};
        for (int k = 0; k < n; k++) { mb.result[k] = d_result_b[rk][k]; mb.subframe[k] = subframes[k]; }
        kp.subframe = subframes[n - 1];
        kp.result = d_result_b[rk][0];
        launch_film_merge_batch(kp, mb, n, rstream);
        HIP_TRY(this, hipGetLastError());
    }
    kp.frames = nullptr; kp.n_frames = 0;
    HIP_TRY(this, hipEventRecord(ev_merge[rk], rstream));
    ev_merge_set[rk] = true;
    last_merge_k = rk;
    return 0;
}

// merge this launch's `result` into accum / frame, after the previous launch's merge (the only cross-frame ordering)
int Context::finish_frame() {
    if (last_merge_k >= 0 && last_merge_k != rk && rstreams[last_merge_k] != rstream) HIP_TRY(this, hipStreamWaitEvent(rstream, ev_merge[last_merge_k], 0));
    launch_film_merge(kp, rstream);
    HIP_TRY(this, hipGetLastError());
    HIP_TRY(this, hipEventRecord(ev_merge[rk], rstream));
    ev_merge_set[rk] = true;
    last_merge_k = rk;
    return 0;
}

}  // namespace spc

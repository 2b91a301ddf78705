// extern "C": read-backs, counters, kernel timing, debug and test hooks, preprocessing entry points (include/spcbpt.h)
// (part of the C ABI library: see capi_common.h for the map of its translation units)
#include "capi_common.h"

using namespace spc;

extern "C" {

int spcbpt_get_light_trace(spcbpt_ctx* c, spcbpt_light_trace_params* out) {
    CTX_CHECK(c);
    if (!out) return SPCBPT_ERR_INVALID_ARG;
    *out = c->lt;
    if (out->core_count == 0) out->core_count = out->num_core - out->core_begin;
    return SPCBPT_OK;
}
int spcbpt_image_size(spcbpt_ctx* c, int* w, int* h) { CTX_CHECK(c); if (w) *w = (int)c->kp.width; if (h) *h = (int)c->kp.height; return SPCBPT_OK; }

int spcbpt_lvc_read(spcbpt_ctx* c, spcbpt_light_vertex* out, int capacity, int* count) {
    CTX_CHECK(c);
    if (!count) return SPCBPT_ERR_INVALID_ARG;
    if (!c->d_lvc) { c->error = "no LVC"; return SPCBPT_ERR_STATE; }
    int rc = c->fetch_counts();
    if (rc) return rc;
    *count = c->lvc_count;
    if (!out) return SPCBPT_OK;
    if (capacity < c->lvc_count) { c->error = "lvc_read: buffer too small"; return SPCBPT_ERR_CAPACITY; }
    HIP_TRY(c, hipMemcpy(out, c->d_lvc, (size_t)c->lvc_count * sizeof(LightVertex), hipMemcpyDeviceToHost));
    return SPCBPT_OK;
}

int spcbpt_sampler_read(spcbpt_ctx* c, spcbpt_subspace* sub, float* cmfs, int32_t* jump, int capacity, int* vc, int* pc) {
    CTX_CHECK(c);
    if (!c->have_sampler) { c->error = "no sampler built"; return SPCBPT_ERR_STATE; }
    if (!sub || !vc || !pc) return SPCBPT_ERR_INVALID_ARG;
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    std::vector<DSubspace> h(SPCBPT_NUM_SUBSPACE);
    const int e = c->eset;   // the set of the last sampler build (not necessarily the latest light pass's)
    HIP_TRY(c, hipMemcpy(h.data(), c->set_subspace[e], h.size() * sizeof(DSubspace), hipMemcpyDeviceToHost));
    for (int i = 0; i < SPCBPT_NUM_SUBSPACE; i++) {
        sub[i].jump_bias = h[i].jump_bias; sub[i].id = i; sub[i].size = h[i].size; sub[i].sum_pmf = h[i].sum_pmf; sub[i].q = 0;
    }
    int hc[2] = {0, 0};
    HIP_TRY(c, hipMemcpy(hc, c->set_counts[e], sizeof(hc), hipMemcpyDeviceToHost));
    *vc = hc[0]; *pc = hc[1];
    if (cmfs && jump) {
        if (capacity < hc[0]) { c->error = "sampler_read: buffer too small"; return SPCBPT_ERR_CAPACITY; }
        HIP_TRY(c, hipMemcpy(cmfs, c->set_cmfs[e], (size_t)hc[0] * 4, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(jump, c->set_vals2[e], (size_t)hc[0] * 4, hipMemcpyDeviceToHost));
    }
    return SPCBPT_OK;
}

int spcbpt_read_accum(spcbpt_ctx* c, float* out) {
    CTX_CHECK(c);
    if (!out || !c->d_accum) { c->error = "no accum buffer"; return SPCBPT_ERR_STATE; }
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    if (int rc = c->check_diag()) return rc;
    HIP_TRY(c, hipMemcpy(out, c->d_accum, (size_t)c->kp.width * c->kp.height * 16, hipMemcpyDeviceToHost));
    return SPCBPT_OK;
}
int spcbpt_read_frame(spcbpt_ctx* c, uint8_t* out) {
    CTX_CHECK(c);
    if (!out || !c->d_frame) { c->error = "no frame buffer"; return SPCBPT_ERR_STATE; }
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    if (int rc = c->check_diag()) return rc;
    HIP_TRY(c, hipMemcpy(out, c->d_frame, (size_t)c->kp.width * c->kp.height * 4, hipMemcpyDeviceToHost));
    return SPCBPT_OK;
}
// The film as of the last queued merge (spcbpt_sync_film's wait), copied on a stream of its own: launches queued BEHIND that merge --
// the interactive loop's speculative next frame, light passes ahead -- are not waited for, which spcbpt_read_accum / _frame do.
int spcbpt_read_film(spcbpt_ctx* c, float* accum_out, uint8_t* frame_out) {
    CTX_CHECK(c);
    if (!c->d_accum || !c->d_frame) { c->error = "no film (spcbpt_resize first)"; return SPCBPT_ERR_STATE; }
    if (!c->cstream) HIP_TRY(c, hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
    if (c->last_merge_k >= 0 && c->ev_merge_set[c->last_merge_k]) HIP_TRY(c, hipStreamWaitEvent(c->cstream, c->ev_merge[c->last_merge_k], 0));
    const size_t px = (size_t)c->kp.width * c->kp.height;
    if (accum_out) HIP_TRY(c, hipMemcpyAsync(accum_out, c->d_accum, px * 16, hipMemcpyDeviceToHost, c->cstream));
    if (frame_out) HIP_TRY(c, hipMemcpyAsync(frame_out, c->d_frame, px * 4, hipMemcpyDeviceToHost, c->cstream));
    HIP_TRY(c, hipStreamSynchronize(c->cstream));
    return c->check_diag();
}
int spcbpt_debug_batch_scratch(spcbpt_ctx* c, int64_t* bytes, int* frames, int* fallbacks) {
    CTX_CHECK(c);
    if (bytes) *bytes = c->sbb_keys ? (int64_t)((size_t)c->sbb_frames * c->sbb_capacity * 16 + (size_t)c->sbb_frames * sampler_build_hist_ints() * sizeof(int)) : 0;
    if (frames) *frames = c->sbb_frames;
    if (fallbacks) *fallbacks = c->sbb_fallbacks;
    return SPCBPT_OK;
}
int spcbpt_debug_read_sampling_tables(spcbpt_ctx* c, uint32_t* guide2, int capacity2, uint16_t* guide1, float* gamma_q) {
    CTX_CHECK(c);
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    if (guide2) {
        if (!c->have_sampler) { c->error = "no sampler built"; return SPCBPT_ERR_STATE; }
        int hc[2] = {0, 0};
        HIP_TRY(c, hipMemcpy(hc, c->set_counts[c->eset], sizeof(hc), hipMemcpyDeviceToHost));
        if (capacity2 < hc[0]) { c->error = "debug_read_sampling_tables: buffer too small"; return SPCBPT_ERR_CAPACITY; }
        HIP_TRY(c, hipMemcpy(guide2, c->set_guide[c->eset], (size_t)hc[0] * 4, hipMemcpyDeviceToHost));
    }
    if ((guide1 || gamma_q) && !c->d_guide1) { c->error = "no subspace tuple installed"; return SPCBPT_ERR_STATE; }
    if (guide1) HIP_TRY(c, hipMemcpy(guide1, c->d_guide1, (size_t)SPCBPT_NUM_SUBSPACE * CMF_GUIDE1 * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if (gamma_q) HIP_TRY(c, hipMemcpy(gamma_q, c->d_gamma_q, (size_t)SPCBPT_NUM_SUBSPACE * SPCBPT_NUM_SUBSPACE * sizeof(float), hipMemcpyDeviceToHost));
    return SPCBPT_OK;
}
int spcbpt_accum_device_ptr(spcbpt_ctx* c, void** p) {
    CTX_CHECK(c);
    if (!p || !c->d_accum) return SPCBPT_ERR_STATE;
    *p = c->d_accum;
    return SPCBPT_OK;
}
int spcbpt_clear_accum(spcbpt_ctx* c) {
    CTX_CHECK(c);
    if (!c->d_accum) return SPCBPT_ERR_STATE;
    if (c->deferred.active) { c->error = "clear_accum: a deferred frame is outstanding (its merge would land in the cleared film): spcbpt_merge_deferred(ctx, keep) first"; return SPCBPT_ERR_STATE; }
    if (c->sync_all()) return SPCBPT_ERR_HIP;   // merges of both render streams may still be pending
    HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, (size_t)c->kp.width * c->kp.height * 16, c->rstreams[0]));
    HIP_TRY(c, hipStreamSynchronize(c->rstreams[0]));
    return SPCBPT_OK;
}

int spcbpt_get_counters(spcbpt_ctx* c, spcbpt_counters* o) {
    CTX_CHECK(c);
    if (!o) return SPCBPT_ERR_INVALID_ARG;
    unsigned long long h[C_COUNT];
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    HIP_TRY(c, hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    o->closest_rays = h[C_CLOSEST]; o->shadow_rays = h[C_SHADOW]; o->node_visits = h[C_NODE]; o->tri_tests = h[C_TRI];
    o->surface_vertices = h[C_VERTEX]; o->textured_hits = h[C_TEX]; o->tree_nodes = h[C_TREE]; o->cmf_probes = h[C_CMF];
    o->connections = h[C_CONN]; o->gamma_q_reads = h[C_GQ]; o->lvc_stores = h[C_LVCW]; o->pixel_samples = h[C_PIX];
    o->eye_paths = h[C_EYE]; o->light_paths = h[C_LIGHT];
    return SPCBPT_OK;
}
int spcbpt_debug_phase_clocks(spcbpt_ctx* c, uint64_t out[19]) {
    CTX_CHECK(c);
    if (!out) return SPCBPT_ERR_INVALID_ARG;
    unsigned long long h[C_COUNT];
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    HIP_TRY(c, hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 5; i++) out[i] = h[C_PUBLIC + i] << 4;
    for (int i = 5; i < 9; i++) out[i] = h[C_PUBLIC + i];
    out[9] = h[C_T_SAMPLE] << 4;
    out[10] = h[C_W_START_MIN]; out[11] = h[C_W_END_MAX]; out[12] = h[C_W_END_SUM]; out[13] = h[C_W_WAVES]; out[14] = h[C_U_TAIL_SLOTS]; out[15] = h[C_U_TAIL_CLOSEST]; out[16] = h[C_U_TAIL_SHADOW]; out[17] = h[C_U_JOB_SLOTS]; out[18] = h[C_U_JOB_LANES];  // summed over lanes (every lane that samples adds its own clock delta)
    return SPCBPT_OK;
}
int spcbpt_reset_counters(spcbpt_ctx* c) {
    CTX_CHECK(c);
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, C_COUNT * sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_counters + C_W_START_MIN, 0xff, sizeof(unsigned long long), c->stream));
    return SPCBPT_OK;
}
int spcbpt_enable_counters(spcbpt_ctx* c, int on) { CTX_CHECK(c); c->counting = on != 0; c->count_executed = on == 2; return SPCBPT_OK; }

int spcbpt_stream(spcbpt_ctx* c, void** s) { CTX_CHECK(c); if (!s) return SPCBPT_ERR_INVALID_ARG; *s = (void*)c->stream; return SPCBPT_OK; }
// Light passes may run ahead of the exchange / sampler build (a sharded job launches frame f + 1's light pass before it
// gathers and builds frame f's): with on != 0 every "light trace" launch queues its buffer set, and spcbpt_lvc_export,
// spcbpt_lvc_import, spcbpt_sync_light and spcbpt_build_sampler address the oldest queued set.  Off (default): they address the
// latest light pass, as the single-GPU loop expects.  Switching clears the queue.
int spcbpt_set_light_ahead(spcbpt_ctx* c, int on) {
    CTX_CHECK(c);
    if (c->deferred.active) { c->error = "set_light_ahead: a deferred frame is outstanding: spcbpt_merge_deferred(ctx, keep) first"; return SPCBPT_ERR_STATE; }
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    const bool was = c->light_ahead;
    c->light_ahead = on != 0;
    c->pending.clear();
    if (was && !c->light_ahead) { c->free_batch_build_scratch(); c->sbb_refused_bytes = 0; }   // only loops with passes ahead build in batches
    return SPCBPT_OK;
}

// What a host loop that shares the context with other code (csrc/viewer.cpp) re-validates its own flags against.
int spcbpt_get_pipeline_state(spcbpt_ctx* c, int* light_ahead, int* pending_passes, int* sampler_intact, int* deferred_outstanding) {
    CTX_CHECK(c);
    if (light_ahead) *light_ahead = c->light_ahead ? 1 : 0;
    if (pending_passes) *pending_passes = (int)c->pending.size();
    if (sampler_intact) *sampler_intact = c->sampler_intact() ? 1 : 0;
    if (deferred_outstanding) *deferred_outstanding = c->deferred.active ? 1 : 0;
    return SPCBPT_OK;
}

// The sampler built last serves eye launches again although a later light pass has been launched since -- if its tables are
// intact (the pass went to another set of the ring; nothing re-installed the tuple, the sky or the cache geometry meanwhile).
int spcbpt_reuse_sampler(spcbpt_ctx* c) {
    CTX_CHECK(c);
    if (!c->sampler_intact()) { c->error = "reuse_sampler: the tables of the last sampler build are gone (a light pass, an import or a new tuple took their set)"; return SPCBPT_ERR_STATE; }
    c->have_sampler = true;
    return SPCBPT_OK;
}

// A host that alternates two device staging buffers for spcbpt_lvc_import calls this before it overwrites one: it returns when
// the import copy that read that buffer (the import before the previous one) has run.  The copies are queued on the light
// stream behind whatever light passes were launched ahead, so no other wait of the exchange sequence implies this.
int spcbpt_lvc_import_wait(spcbpt_ctx* c) {
    CTX_CHECK(c);
    if (c->import_gen >= 2 && c->ev_import[c->import_gen & 1]) HIP_TRY(c, hipEventSynchronize(c->ev_import[c->import_gen & 1]));
    return SPCBPT_OK;
}

// Waits for the OLDEST pending light pass (what spcbpt_lvc_export hands out), not for everything queued on the light stream:
// a later light pass may already be running ahead.  With nothing pending it waits for the light stream.
int spcbpt_sync_light(spcbpt_ctx* c) {
    CTX_CHECK(c);
    if (!c->pending.empty()) HIP_TRY(c, hipEventSynchronize(c->ev_light[c->pending.front()]));
    else HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SPCBPT_OK;
}
int spcbpt_sync(spcbpt_ctx* c) { CTX_CHECK(c); if (c->sync_all()) return SPCBPT_ERR_HIP; return c->check_diag(); }

// Developer probe of the HBM part of the traversal stack (tests/): _arm fills every spill area allocated so far with a word no
// stack entry can hold; _count returns how many words kernels have overwritten since.  Zero kernel cost.
int spcbpt_debug_spill_arm(spcbpt_ctx* c) {
    CTX_CHECK(c);
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    if (c->d_spill) HIP_TRY(c, hipMemset(c->d_spill, 0xff, c->spill_capacity * 4));
    if (c->b_spill) HIP_TRY(c, hipMemset(c->b_spill, 0xff, c->b_spill_capacity * 4));
    for (int k = 0; k < Context::kMaxRender; k++) if (c->d_spill_rs[k]) HIP_TRY(c, hipMemset(c->d_spill_rs[k], 0xff, c->spill_rs_capacity[k] * 4));
    HIP_TRY(c, hipDeviceSynchronize());
    return SPCBPT_OK;
}
int spcbpt_debug_spill_count(spcbpt_ctx* c, uint64_t* written, int* entries_per_thread) {
    CTX_CHECK(c);
    if (!written) return SPCBPT_ERR_INVALID_ARG;
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    uint64_t n = 0;
    std::vector<uint32_t> h;
    auto scan = [&](const uint32_t* d, size_t words) -> int {
        if (!d || !words) return 0;
        h.resize(words);
        HIP_TRY(c, hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost));
        for (uint32_t w : h) n += w != 0xffffffffu;
        return 0;
    };
    if (scan(c->d_spill, c->spill_capacity) || scan(c->b_spill, c->b_spill_capacity)) return SPCBPT_ERR_HIP;
    for (int k = 0; k < Context::kMaxRender; k++) if (scan(c->d_spill_rs[k], c->spill_rs_capacity[k])) return SPCBPT_ERR_HIP;
    *written = n;
    if (entries_per_thread) *entries_per_thread = c->spill_entries_needed();
    return SPCBPT_OK;
}

int spcbpt_kernel_time(spcbpt_ctx* c, const char* name, double* avg_ms, int* launches) {
    CTX_CHECK(c);
    if (!name || !avg_ms || !launches) return SPCBPT_ERR_INVALID_ARG;
    c->resolve_spans();
    auto it = c->times.find(name);
    if (it == c->times.end() || it->second.second == 0) { *avg_ms = 0; *launches = 0; return SPCBPT_OK; }
    *avg_ms = it->second.first / it->second.second;
    *launches = it->second.second;
    return SPCBPT_OK;
}
int spcbpt_reset_kernel_time(spcbpt_ctx* c) { CTX_CHECK(c); c->resolve_spans(); c->times.clear(); return SPCBPT_OK; }
int spcbpt_enable_kernel_timing(spcbpt_ctx* c, int on) { CTX_CHECK(c); c->timing = on != 0; return SPCBPT_OK; }

static int trace_common(spcbpt_ctx* c, const float* rays, int n, float** d_rays) {
    if (!rays || n < 0) { c->error = "bad rays"; return SPCBPT_ERR_INVALID_ARG; }
    for (size_t i = 0; i < (size_t)n * 8; i++)
        if (!std::isfinite(rays[i]) && !(i % 8 == 7)) { c->error = "non-finite ray component"; return SPCBPT_ERR_INVALID_ARG; }
    HIP_TRY(c, dev_alloc(d_rays, (size_t)n * 8));
    HIP_TRY(c, hipMemcpyAsync(*d_rays, rays, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    return c->ensure_spill(((size_t)n + 255) / 256 * 256);
}
int spcbpt_trace_closest(spcbpt_ctx* c, const float* rays, int n, float* out_t, int32_t* out_tri, float* out_uv) {
    CTX_CHECK(c);
    if (!out_t || !out_tri || !out_uv) return SPCBPT_ERR_INVALID_ARG;
    float* d_rays = nullptr; float* d_t = nullptr; int* d_tri = nullptr; float* d_uv = nullptr;
    int rc = trace_common(c, rays, n, &d_rays);
    if (rc) { dev_free(d_rays); return rc; }
    HIP_TRY(c, dev_alloc(&d_t, (size_t)n)); HIP_TRY(c, dev_alloc(&d_tri, (size_t)n)); HIP_TRY(c, dev_alloc(&d_uv, (size_t)n * 2));
    launch_trace_closest(c->kp, d_rays, n, d_t, d_tri, d_uv, c->stream);
    hipError_t e = c->sync_all() ? hipErrorUnknown : hipSuccess;
    const int dg = e == hipSuccess ? c->check_diag() : 0;
    if (e == hipSuccess) e = hipMemcpy(out_t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_tri, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_uv, d_uv, (size_t)n * 8, hipMemcpyDeviceToHost);
    dev_free(d_rays); dev_free(d_t); dev_free(d_tri); dev_free(d_uv);
    if (e != hipSuccess) { c->error = hipGetErrorString(e); return SPCBPT_ERR_HIP; }
    return dg;
}
int spcbpt_trace_any(spcbpt_ctx* c, const float* rays, int n, int32_t* out_visible) {
    CTX_CHECK(c);
    if (!out_visible) return SPCBPT_ERR_INVALID_ARG;
    float* d_rays = nullptr; int* d_vis = nullptr;
    int rc = trace_common(c, rays, n, &d_rays);
    if (rc) { dev_free(d_rays); return rc; }
    HIP_TRY(c, dev_alloc(&d_vis, (size_t)n));
    launch_trace_any(c->kp, d_rays, n, d_vis, c->stream);
    hipError_t e = c->sync_all() ? hipErrorUnknown : hipSuccess;
    const int dg = e == hipSuccess ? c->check_diag() : 0;
    if (e == hipSuccess) e = hipMemcpy(out_visible, d_vis, (size_t)n * 4, hipMemcpyDeviceToHost);
    dev_free(d_rays); dev_free(d_vis);
    if (e != hipSuccess) { c->error = hipGetErrorString(e); return SPCBPT_ERR_HIP; }
    return dg;
}

int spcbpt_set_connection_sampler(spcbpt_ctx* c, int mode) {
    CTX_CHECK(c);
    if (mode != SPCBPT_SAMPLER_SUBSPACE && mode != SPCBPT_SAMPLER_UNIFORM) { c->error = "set_connection_sampler: unknown mode"; return SPCBPT_ERR_INVALID_ARG; }
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    c->kp.uniform_lvc = mode;
    return SPCBPT_OK;
}

int spcbpt_debug_trace_bench(spcbpt_ctx* c, const float* rays, int n, int mode, int any, int repeat, float* out_t, int32_t* out_tri, float* out_uv,
                             int32_t* out_visible, double* avg_ms, uint64_t stats[5]) {
    CTX_CHECK(c);
    if (!rays || n < 1 || mode < 0 || mode > 4 || repeat < 1 || (any && !out_visible) || (!any && (!out_t || !out_tri || !out_uv))) { c->error = "debug_trace_bench: bad arguments"; return SPCBPT_ERR_INVALID_ARG; }
    if (mode >= 1 && 3 * c->bvh_depth > (mode == 1 || mode == 4 ? 64 : 48)) { c->error = "debug_trace_bench: the quad kernel's per-ray LDS stack holds " + std::to_string(mode == 1 || mode == 4 ? 64 : 48) + " entries (3 x BVH depth " + std::to_string(c->bvh_depth) + " needed)"; return SPCBPT_ERR_CAPACITY; }
    float* d_rays = nullptr; float* d_t = nullptr; int* d_tri = nullptr; float* d_uv = nullptr; int* d_vis = nullptr;
    uint32_t* d_counter = nullptr; unsigned long long* d_stats = nullptr;
    int rc = trace_common(c, rays, n, &d_rays);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() {
        dev_free(d_rays); dev_free(d_t); dev_free(d_tri); dev_free(d_uv); dev_free(d_vis); dev_free(d_counter); dev_free(d_stats);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    if (rc) { cleanup(); return rc; }
    if (mode >= 1 && !c->d_nodes_q) {   // the quad layout of the same nodes, built once
        e = dev_alloc(&c->d_nodes_q, (size_t)c->n_nodes * 16);
        if (e == hipSuccess) { launch_repack_nodes_quad(c->d_nodes, c->d_nodes_q, c->n_nodes, c->stream); e = hipGetLastError(); }
    }
    if (e == hipSuccess && mode == 4 && !c->d_nodes_q2) {   // ... and the same with the scale exponents as signed bytes
        e = dev_alloc(&c->d_nodes_q2, (size_t)c->n_nodes * 16);
        if (e == hipSuccess) { launch_repack_nodes_quad2(c->d_nodes_q, c->d_nodes_q2, c->n_nodes, c->stream); e = hipGetLastError(); }
    }
    const float* nodes_q = mode == 4 ? c->d_nodes_q2 : c->d_nodes_q;
    const int per_cu = trace_bench_blocks_per_cu(mode, any != 0);
    const int rays_per_block = mode == 0 ? 256 : (mode == 4 ? 64 : 64 << (mode - 1));
    const int blocks = std::max(1, std::min(c->num_cus * per_cu, (n + rays_per_block - 1) / rays_per_block));
    if (e == hipSuccess && mode == 0) { rc = c->ensure_spill((size_t)blocks * 256); if (rc) { cleanup(); return rc; } }
    if (e == hipSuccess) e = dev_alloc(&d_counter, (size_t)1);
    if (e == hipSuccess) e = dev_alloc(&d_stats, (size_t)5);
    if (e == hipSuccess && !any) { e = dev_alloc(&d_t, (size_t)n); if (e == hipSuccess) e = dev_alloc(&d_tri, (size_t)n); if (e == hipSuccess) e = dev_alloc(&d_uv, (size_t)n * 2); }
    if (e == hipSuccess && any) e = dev_alloc(&d_vis, (size_t)n);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double total_ms = 0.0;
    for (int k = 0; k <= repeat && e == hipSuccess; k++) {   // launch 0 warms up
        e = hipMemsetAsync(d_counter, 0, sizeof(uint32_t), c->stream);
        if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
        if (e == hipSuccess) { launch_trace_bench(c->kp, mode, any != 0, false, nodes_q, d_rays, n, d_counter, d_t, d_tri, d_uv, d_vis, d_stats, blocks, c->stream); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (k > 0) total_ms += ms;
    }
    if (e == hipSuccess && stats) {
        e = hipMemsetAsync(d_counter, 0, sizeof(uint32_t), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_stats, 0, 5 * sizeof(unsigned long long), c->stream);
        if (e == hipSuccess) { launch_trace_bench(c->kp, mode, any != 0, true, nodes_q, d_rays, n, d_counter, d_t, d_tri, d_uv, d_vis, d_stats, blocks, c->stream); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(stats, d_stats, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    }
    const int dg = e == hipSuccess ? c->check_diag() : 0;
    if (e == hipSuccess && !any) { e = hipMemcpy(out_t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost); if (e == hipSuccess) e = hipMemcpy(out_tri, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost);
                                   if (e == hipSuccess) e = hipMemcpy(out_uv, d_uv, (size_t)n * 8, hipMemcpyDeviceToHost); }
    if (e == hipSuccess && any) e = hipMemcpy(out_visible, d_vis, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (avg_ms) *avg_ms = total_ms / repeat;
    cleanup();
    if (e != hipSuccess) { c->error = std::string("debug_trace_bench: ") + hipGetErrorString(e); return SPCBPT_ERR_HIP; }
    return dg;
}

int spcbpt_debug_unit(spcbpt_ctx* c, int op, const uint32_t* in, int in_words, uint32_t* out, int out_words, int n, const float* aux, int aux_floats) {
    CTX_CHECK(c);
    static const int need_in[8] = {24, 10, 2, 3, 2, 1, 52, 36}, need_out[8] = {12, 1, 6, 3, 5, 3, 4, 40};
    if (op < 0 || op > 7 || !in || !out || n < 0 || in_words < need_in[op] || out_words < need_out[op]) { c->error = "debug_unit: bad op or record size"; return SPCBPT_ERR_INVALID_ARG; }
    if (op != SPCBPT_UNIT_BSDF && op != SPCBPT_UNIT_BSEARCH && !c->have_subspace) { c->error = "debug_unit: needs a subspace tuple"; return SPCBPT_ERR_STATE; }
    if ((op == SPCBPT_UNIT_STAGE2 || op == SPCBPT_UNIT_UNIFORM) && !c->have_sampler) { c->error = "debug_unit: needs a built sampler"; return SPCBPT_ERR_STATE; }
    if (op == SPCBPT_UNIT_BSEARCH && (!aux || aux_floats < 1)) { c->error = "debug_unit: BSEARCH needs the CMF in aux"; return SPCBPT_ERR_INVALID_ARG; }
    if (n == 0) return SPCBPT_OK;
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    uint32_t *d_in = nullptr, *d_out = nullptr;
    float* d_aux = nullptr;
    int rc = SPCBPT_OK;
    hipError_t e = dev_alloc(&d_in, (size_t)n * in_words);
    if (e == hipSuccess) e = dev_alloc(&d_out, (size_t)n * out_words);
    if (e == hipSuccess && aux && aux_floats > 0) e = dev_alloc(&d_aux, (size_t)aux_floats);
    if (e == hipSuccess) e = hipMemcpy(d_in, in, (size_t)n * in_words * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, (size_t)n * out_words * 4, c->stream);   // on the kernel's (non-blocking) stream: a memset on the null stream is not ordered before it
    if (e == hipSuccess && d_aux) e = hipMemcpy(d_aux, aux, (size_t)aux_floats * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        KParams kp = c->kp;
        const int es = c->eset;   // the tables of the last sampler build
        kp.lvc = c->set_lvc[es]; kp.lvc_sorted = c->set_lvc_sorted[es]; kp.subspace = c->set_subspace[es]; kp.cmfs = c->set_cmfs[es]; kp.guide = c->set_guide[es];
        kp.jump = reinterpret_cast<const int32_t*>(c->set_vals2[es]); kp.sampler_counts = c->set_counts[es];
        kp.counters = nullptr;
        if (op == SPCBPT_UNIT_EYE_STEP) {
            rc = c->ensure_spill(((size_t)n + 255) / 256 * 256);
            kp.spill = c->kp.spill; kp.spill_entries = c->kp.spill_entries;
        }
        if (rc == SPCBPT_OK) {
            launch_unit(kp, op, d_in, in_words, d_out, out_words, n, d_aux, c->stream);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * out_words * 4, hipMemcpyDeviceToHost);
        }
    }
    dev_free(d_in); dev_free(d_out); dev_free(d_aux);
    if (e != hipSuccess) { c->error = std::string("debug_unit: ") + hipGetErrorString(e); return SPCBPT_ERR_HIP; }
    if (rc) return rc;
    return c->check_diag();
}

int spcbpt_preprocess(spcbpt_ctx* c, int target_paths, int target_q_paths, int train) {
    CTX_CHECK(c);
    return c->preprocess(target_paths, target_q_paths, train != 0);
}

int spcbpt_get_subspace(spcbpt_ctx* c, spcbpt_tree_node* et, int* ne, int cap_e, spcbpt_tree_node* lt, int* nl, int cap_l, float* q, float* g) {
    CTX_CHECK(c);
    if (!c->have_subspace) { c->error = "no subspace tuple installed"; return SPCBPT_ERR_STATE; }
    if (!ne || !nl) return SPCBPT_ERR_INVALID_ARG;
    *ne = (int)c->h_eye_tree.size(); *nl = (int)c->h_light_tree.size();
    if (et) { if (cap_e < *ne) return SPCBPT_ERR_CAPACITY; memcpy(et, c->h_eye_tree.data(), c->h_eye_tree.size() * sizeof(spcbpt_tree_node)); }
    if (lt) { if (cap_l < *nl) return SPCBPT_ERR_CAPACITY; memcpy(lt, c->h_light_tree.data(), c->h_light_tree.size() * sizeof(spcbpt_tree_node)); }
    if (q) memcpy(q, c->h_Q.data(), c->h_Q.size() * 4);
    if (g) memcpy(g, c->h_gamma.data(), c->h_gamma.size() * 4);
    return SPCBPT_OK;
}

int spcbpt_scene_info(spcbpt_ctx* c, int* nt, int* nn, int* depth) {
    CTX_CHECK(c);
    if (nt) *nt = c->n_triangles;
    if (nn) *nn = c->n_nodes;
    if (depth) *depth = c->bvh_depth;
    return SPCBPT_OK;
}

}  // extern "C"

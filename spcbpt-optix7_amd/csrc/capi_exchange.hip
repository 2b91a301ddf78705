// extern "C": light-vertex-cache export / import and film band packing of a sharded job (include/spcbpt.h; used by csrc/mgpu.cpp and dist.py)
// (part of the C ABI library: see capi_common.h for the map of its translation units)
#include "capi_common.h"

using namespace spc;

extern "C" {

int spcbpt_lvc_export(spcbpt_ctx* c, void** dv, void** dc, int* cap) {
    CTX_CHECK(c);
    if (!dv || !dc || !cap) return SPCBPT_ERR_INVALID_ARG;
    if (!c->d_lvc) { c->error = "no LVC allocated"; return SPCBPT_ERR_STATE; }
    const int b = c->build_set();   // the oldest light pass without a sampler: the shard that is exchanged next
    *dv = c->set_lvc[b]; *dc = c->set_counts[b]; *cap = (int)c->lvc_capacity;
    return SPCBPT_OK;
}

int spcbpt_lvc_import(spcbpt_ctx* c, const void* verts, int count, int is_device) {
    CTX_CHECK(c);
    if (!verts || count < 0) { c->error = "bad LVC import"; return SPCBPT_ERR_INVALID_ARG; }
    if ((size_t)std::max(count, 1) > c->lvc_capacity && c->pending.size() > 1) {
        c->error = "lvc_import: the cache does not fit and cannot grow while a later light pass is in flight (spcbpt_lvc_set_capacity before the first pass)";
        return SPCBPT_ERR_CAPACITY;
    }
    int rc = c->ensure_lvc_capacity((size_t)std::max(count, 1));
    if (rc) return rc;
    const int b = c->build_set();
    if (c->light_lane_of_set[b] != 0) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_light[b], 0));   // the pass that filled this set ran on the second lane
    if ((const void*)c->set_lvc[b] != verts)
        HIP_TRY(c, hipMemcpyAsync(c->set_lvc[b], verts, (size_t)count * sizeof(LightVertex), is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    int* h = c->h_import_counts + 2 * b;   // pinned: the upload may run after this call returns
    h[0] = count; h[1] = 0;
    HIP_TRY(c, hipMemcpyAsync(c->set_counts[b], h, 2 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    // Host memory: wait for the light stream, the caller may reuse `verts` at once.  Device memory: no wait at all -- the copy
    // is ordered on the light stream; the caller keeps `verts` untouched until a light pass launched AFTER this call has been
    // waited for with spcbpt_sync_light (dist.py alternates two staging buffers, which covers a light pass running one frame
    // ahead).  The render streams are never waited for: the set written here is not one an eye kernel in flight reads.
    if (!is_device) HIP_TRY(c, hipStreamSynchronize(c->stream));
    else {   // spcbpt_lvc_import_wait: when may the staging buffer of the import before the previous one be written again
        hipEvent_t& ev = c->ev_import[c->import_gen & 1];
        if (!ev) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(ev, c->stream));
        c->import_gen++;
    }
    HIP_TRY(c, hipEventRecord(c->ev_set_stream[b], c->stream));
    c->ev_set_touched[b] = true;
    c->set_count_host[b] = count;
    c->set_bound[b] = -1;
    c->light_counts_valid[b] = false;
    c->light_lane_of_set[b] = 0;   // from here on the set's contents are ordered on `stream`
    for (auto it = c->built_sets.begin(); it != c->built_sets.end();) it = (*it == b) ? c->built_sets.erase(it) : it + 1;   // a sampler built from the old contents is gone
    if (b == c->lset) c->lvc_count = count;
    if (c->keys_set == b) c->keys_ready = false;
    c->have_sampler = false;
    return SPCBPT_OK;
}

int spcbpt_lvc_export_on(spcbpt_ctx* c, void* hip_stream, void** dv, void** dc, int* cap) {
    CTX_CHECK(c);
    if (!dv || !dc || !cap) return SPCBPT_ERR_INVALID_ARG;
    return c->export_on(reinterpret_cast<hipStream_t>(hip_stream), dv, dc, cap);
}
int spcbpt_lvc_import_gathered(spcbpt_ctx* c, const void* shards, const void* counts_all, int world, int shard_capacity, void* hip_stream) {
    CTX_CHECK(c);
    return c->import_gathered(shards, reinterpret_cast<const int*>(counts_all), world, shard_capacity, reinterpret_cast<hipStream_t>(hip_stream), 1);
}
int spcbpt_lvc_export_batch_on(spcbpt_ctx* c, void* hip_stream, int n_frames, void* send, void* send_counts, int shard_capacity) {
    CTX_CHECK(c);
    return c->export_batch_on(reinterpret_cast<hipStream_t>(hip_stream), n_frames, send, reinterpret_cast<int*>(send_counts), shard_capacity);
}
int spcbpt_lvc_import_gathered_batch(spcbpt_ctx* c, const void* shards, const void* counts_all, int world, int n_frames, int shard_capacity, void* hip_stream) {
    CTX_CHECK(c);
    return c->import_gathered(shards, reinterpret_cast<const int*>(counts_all), world, shard_capacity, reinterpret_cast<hipStream_t>(hip_stream), n_frames);
}
// film exchange helpers of a sharded job (exchange 2, once per read-out): pack this rank's 8-row bands contiguously / scatter
// every rank's packed bands back into the full image.  Queued on `hip_stream` after the render streams' merges.
int spcbpt_film_pack_bands(spcbpt_ctx* c, int rank, int world, void* packed, void* hip_stream) {
    CTX_CHECK(c);
    if (!packed || !c->d_accum || world < 1 || rank < 0 || rank >= world) return SPCBPT_ERR_INVALID_ARG;
    if (c->sync_all()) return SPCBPT_ERR_HIP;   // a read-out: every frame's merge has to be in the film
    launch_pack_bands(c->d_accum, (int)c->kp.width, (int)c->kp.height, rank, world, reinterpret_cast<float*>(packed), false, reinterpret_cast<hipStream_t>(hip_stream));
    HIP_TRY(c, hipGetLastError());
    return SPCBPT_OK;
}
int spcbpt_film_unpack_bands(spcbpt_ctx* c, int world, const void* packed_all, void* out_image, void* hip_stream) {
    CTX_CHECK(c);
    if (!packed_all || !out_image || world < 1) return SPCBPT_ERR_INVALID_ARG;
    launch_pack_bands(reinterpret_cast<float*>(out_image), (int)c->kp.width, (int)c->kp.height, 0, world,
                      reinterpret_cast<float*>(const_cast<void*>(packed_all)), true, reinterpret_cast<hipStream_t>(hip_stream));
    HIP_TRY(c, hipGetLastError());
    return SPCBPT_OK;
}

}  // extern "C"

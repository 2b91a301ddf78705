// Context: shards and imports of a sharded job (driven by libspcbpt_mgpu: csrc/mgpu.cpp)
// (part of the C ABI library: see capi_common.h for the map of its translation units)
#include "capi_common.h"

using namespace spc;

namespace spc {

// The oldest pending light pass's shard for an exchange that runs on the caller's stream `xs`: instead of the host waiting for
// the pass (spcbpt_sync_light), `xs` waits for it on the device.
int Context::export_on(hipStream_t xs, void** dv, void** dc, int* cap) {
    if (!d_lvc) { error = "no LVC allocated"; return SPCBPT_ERR_STATE; }
    const int b = build_set();
    if (light_counts_valid[b]) HIP_TRY(this, hipStreamWaitEvent(xs, ev_light[b], 0));
    else {   // a cache written some other way (import): ordered on `stream`
        HIP_TRY(this, hipEventRecord(ev_set_stream[b], stream));
        ev_set_touched[b] = true;
        HIP_TRY(this, hipStreamWaitEvent(xs, ev_set_stream[b], 0));
    }
    *dv = set_lvc[b]; *dc = set_counts[b]; *cap = (int)lvc_capacity;
    return 0;
}

// Receiving side of exchange 1 (k_gather_compact): `shards` = world x shard_cap vertices as the all-gather left them, `counts_all`
// = world x (vertex_count, path_count), both device memory that `xs` has finished writing by the time this is queued.  Everything
// is queued on `xs`; nothing here waits on the host.
int Context::import_gathered(const void* shards, const int* counts_all, int world, int shard_cap, hipStream_t xs, int nf) {
    if (!shards || !counts_all || world < 1 || shard_cap < 1 || nf < 1 || nf > kMaxBatchFrames) { error = "lvc_import_gathered: bad arguments"; return SPCBPT_ERR_INVALID_ARG; }
    if (!d_lvc) { error = "no LVC allocated"; return SPCBPT_ERR_STATE; }
    if (nf > 1 && (int)pending.size() < nf) { error = "lvc_import_gathered_batch: fewer light passes are pending than frames were gathered"; return SPCBPT_ERR_STATE; }
    // the sets' previous readers: eye kernels (ev_render) were waited for by the light pass that refilled them; their own light
    // passes and the all-gather that read them as (or packed them into) the send buffer precede this call on `xs` (export_on)
    CompactBatch dst = {};
    int sets[kMaxBatchFrames];
    for (int k = 0; k < nf; k++) { sets[k] = nf == 1 ? build_set() : pending[(size_t)k]; dst.lvc[k] = set_lvc[sets[k]]; dst.counts[k] = set_counts[sets[k]]; }
    launch_gather_compact(reinterpret_cast<const LightVertex*>(shards), counts_all, world, shard_cap, (int)std::min<size_t>(lvc_capacity, 0x7fffffff),
                          dst, nf, reinterpret_cast<int*>(d_diag + 1), xs);
    HIP_TRY(this, hipGetLastError());
    for (int k = 0; k < nf; k++) {
        const int b = sets[k];
        HIP_TRY(this, hipEventRecord(ev_exch[b], xs));
        ev_exch_set[b] = true;
        set_bound[b] = (int)std::min<size_t>((size_t)world * (size_t)shard_cap, lvc_capacity);
        set_count_host[b] = -1;
        light_counts_valid[b] = false;
        light_lane_of_set[b] = 0;
        for (auto it = built_sets.begin(); it != built_sets.end();) it = (*it == b) ? built_sets.erase(it) : it + 1;
        if (b == lset) lvc_count = -1;
        if (keys_set == b) keys_ready = false;
    }
    have_sampler = false;
    return 0;
}

// Sending side of one exchange per light batch: the shards of the `nf` oldest pending passes packed into the caller's contiguous
// send buffer (nf x shard_cap vertices, nf count pairs) on `xs`, which waits on the device for the passes that fill them.
int Context::export_batch_on(hipStream_t xs, int nf, void* send, int* send_counts, int shard_cap) {
    if (!d_lvc) { error = "no LVC allocated"; return SPCBPT_ERR_STATE; }
    if (!send || !send_counts || nf < 1 || nf > kMaxBatchFrames || shard_cap < 1) { error = "lvc_export_batch_on: bad arguments"; return SPCBPT_ERR_INVALID_ARG; }
    if ((int)pending.size() < nf) { error = "lvc_export_batch_on: fewer light passes are pending than frames were asked for (launch the batch's passes first)"; return SPCBPT_ERR_STATE; }
    CompactBatch src = {};
    for (int k = 0; k < nf; k++) {
        const int b = pending[(size_t)k];
        if (light_counts_valid[b]) HIP_TRY(this, hipStreamWaitEvent(xs, ev_light[b], 0));
        else {
            HIP_TRY(this, hipEventRecord(ev_set_stream[b], stream));
            ev_set_touched[b] = true;
            HIP_TRY(this, hipStreamWaitEvent(xs, ev_set_stream[b], 0));
        }
        src.lvc[k] = set_lvc[b]; src.counts[k] = set_counts[b];
    }
    launch_pack_shards(src, nf, shard_cap, reinterpret_cast<LightVertex*>(send), send_counts, xs);
    HIP_TRY(this, hipGetLastError());
    return 0;
}

}  // namespace spc

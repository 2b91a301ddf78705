// libspcbpt_mgpu.so -- the N-GPU host of the hot path (include/spcbpt_mgpu.h): one rank per MI355X, RCCL over xGMI.  Everything
// device-side lives in libspcbpt_hip.so behind include/spcbpt.h (spcbpt_lvc_export_on, spcbpt_lvc_import_gathered,
// spcbpt_film_pack_bands / _unpack_bands); this file owns the communicator, its stream, the gather buffers and the call order.
// The reference has no counterpart (single GPU: optixPathTracer.cpp:791-822); the partitioning is BASELINE.json's north_star.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/spcbpt_mgpu.h"

namespace {

constexpr size_t kVertexBytes = sizeof(spcbpt_light_vertex);   // 96

struct LocalGroup;

}  // namespace

struct spcbpt_comm {
    spcbpt_ctx* ctx = nullptr;
    int rank = 0, world = 1, device = 0;
    ncclComm_t nccl = nullptr;      // RCCL transport, or
    LocalGroup* grp = nullptr;      // ranks that share one device (copies as the transport)
    hipStream_t xs = nullptr;       // exchange stream (high priority: small transfers next to persistent render kernels)
    int shard_cap = 0;              // vertices per shard in exchange 1 (agreed by all ranks)
    int scratch_cap = 0;            // upper limit of shard_cap: what a light pass of this rank can produce at most
    void* d_gather = nullptr; size_t gather_bytes = 0;      // world x frames x shard_cap vertices
    int* d_counts_all = nullptr;                            // world x frames x {vertex_count, path_count}
    void* d_send = nullptr; size_t send_bytes = 0;          // batched exchange: frames x shard_cap vertices, packed
    int* d_send_counts = nullptr;                           // frames x {vertex_count, path_count}
    int* h_counts_all = nullptr;                            // pinned mirror (calibration only)
    float* d_pack = nullptr; float* d_pack_all = nullptr; size_t pack_floats = 0;   // film bands: own block / every rank's
    void* d_stage = nullptr; size_t stage_bytes = 0;        // broadcast staging
    double* d_scalar = nullptr;                             // barrier / max
    std::string error;
    // local transport: what this rank contributed to the collective in flight
    void* l_send = nullptr; void* l_counts = nullptr; hipEvent_t l_ready = nullptr; bool l_posted = false;
    bool l_film_posted = false; void* l_film_out = nullptr;
    int l_frames = 1;
};

namespace {

struct LocalGroup {
    std::vector<spcbpt_comm*> ranks;
    std::mutex mu;
    int refs = 0;
    // broadcast_subspace
    bool have_tuple = false;
    std::vector<spcbpt_tree_node> et, lt;
    std::vector<float> q, g;
    std::vector<bool> wants_tuple;
    double running_max = 0.0; int max_calls = 0;
    int calib_max = 0, calib_min_lvc = 0;
    long long calib_calls = 0;   // every rank calibrates once per round (world calls)
};

int fail(spcbpt_comm* c, int code, const std::string& msg) {
    if (c) c->error = msg;
    return code;
}
void unpost_all(LocalGroup* g);
// Local transport: the exchange completes in the call of the LAST rank to post.  Whatever way that call ends -- success, a rank
// that posted another frame count, a failed allocation or HIP call -- every rank is un-posted and the events of the call are
// destroyed, so the group can exchange again (a call that merely posts keeps its mark: keep()).
struct PostGuard {
    LocalGroup* g;
    bool armed = true;
    std::vector<hipEvent_t> events;
    explicit PostGuard(LocalGroup* grp) : g(grp) {}
    void keep() { armed = false; }
    ~PostGuard() {
        for (hipEvent_t e : events) if (e) (void)hipEventDestroy(e);
        if (armed) unpost_all(g);
    }
};
#define HIPX(c, expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) return fail(c, SPCBPT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define NCCLX(c, expr)                                                                                    \
    do {                                                                                                  \
        ncclResult_t r__ = (expr);                                                                        \
        if (r__ != ncclSuccess) return fail(c, SPCBPT_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r__)); \
    } while (0)
#define CTXX(c, expr)                                                                                     \
    do {                                                                                                  \
        int r__ = (expr);                                                                                 \
        if (r__ != 0) return fail(c, r__, std::string(#expr) + ": " + spcbpt_last_error((c)->ctx));          \
    } while (0)

void unpost_all(LocalGroup* g) { for (spcbpt_comm* s : g->ranks) s->l_posted = false; }

int ensure_gather(spcbpt_comm* c, int frames = 1) {
    const size_t need = (size_t)c->world * (size_t)frames * (size_t)c->shard_cap * kVertexBytes;
    if (need > c->gather_bytes) {
        HIPX(c, hipStreamSynchronize(c->xs));
        if (c->d_gather) (void)hipFree(c->d_gather);
        c->d_gather = nullptr;
        HIPX(c, hipMalloc(&c->d_gather, need));
        c->gather_bytes = need;
    }
    return 0;
}

int ensure_send(spcbpt_comm* c, int frames) {
    const size_t need = (size_t)frames * (size_t)c->shard_cap * kVertexBytes;
    if (need > c->send_bytes) {
        HIPX(c, hipStreamSynchronize(c->xs));
        if (c->d_send) (void)hipFree(c->d_send);
        c->d_send = nullptr;
        HIPX(c, hipMalloc(&c->d_send, need));
        c->send_bytes = need;
    }
    return 0;
}

constexpr int kMaxFrames = 32;   // = kMaxBatchFrames of the core library (spcbpt_launch_light_batch)

int common_init(spcbpt_comm* c) {
    HIPX(c, hipGetDevice(&c->device));
    int least = 0, greatest = 0;
    HIPX(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPX(c, hipStreamCreateWithPriority(&c->xs, hipStreamNonBlocking, greatest));
    HIPX(c, hipMalloc(reinterpret_cast<void**>(&c->d_counts_all), (size_t)c->world * kMaxFrames * 2 * sizeof(int)));
    HIPX(c, hipMalloc(reinterpret_cast<void**>(&c->d_send_counts), (size_t)kMaxFrames * 2 * sizeof(int)));
    HIPX(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_counts_all), (size_t)c->world * 2 * sizeof(int)));
    HIPX(c, hipMalloc(reinterpret_cast<void**>(&c->d_scalar), 2 * sizeof(double)));
    HIPX(c, hipEventCreateWithFlags(&c->l_ready, hipEventDisableTiming));
    // default shard capacity: whatever a rank's light pass can produce (its padded scratch core_count x core_padding; the ranks
    // agree on the largest: the last rank of core_range takes the remainder).  Tightened by calibrate -- which a job should call:
    // the context sizes its caches from a measured pass (spcbpt_lvc_set_capacity), not from this worst case.
    spcbpt_light_trace_params lt;
    CTXX(c, spcbpt_get_light_trace(c->ctx, &lt));
    c->scratch_cap = (int)std::min<long long>((long long)lt.core_count * lt.core_padding, 0x7fffffff);
    c->shard_cap = c->scratch_cap;
    return 0;
}

// every rank's block of exchange 2: ceil(bands / world) bands of 8 rows
int film_block_floats(spcbpt_comm* c, size_t* floats) {
    int w = 0, h = 0;
    CTXX(c, spcbpt_image_size(c->ctx, &w, &h));
    if (w < 1 || h < 1) return fail(c, SPCBPT_ERR_STATE, "gather_film before spcbpt_resize");
    const int bands = (h + 7) / 8, per_rank = (bands + c->world - 1) / c->world;
    *floats = (size_t)per_rank * 8 * (size_t)w * 4;
    return 0;
}
int ensure_pack(spcbpt_comm* c, size_t floats) {
    if (floats > c->pack_floats) {
        HIPX(c, hipStreamSynchronize(c->xs));
        if (c->d_pack) (void)hipFree(c->d_pack);
        if (c->d_pack_all) (void)hipFree(c->d_pack_all);
        c->d_pack = c->d_pack_all = nullptr;
        HIPX(c, hipMalloc(reinterpret_cast<void**>(&c->d_pack), floats * sizeof(float)));
        HIPX(c, hipMalloc(reinterpret_cast<void**>(&c->d_pack_all), floats * sizeof(float) * (size_t)c->world));
        c->pack_floats = floats;
    }
    return 0;
}
int ensure_stage(spcbpt_comm* c, size_t bytes) {
    if (bytes > c->stage_bytes) {
        if (c->d_stage) (void)hipFree(c->d_stage);
        c->d_stage = nullptr;
        HIPX(c, hipMalloc(&c->d_stage, bytes));
        c->stage_bytes = bytes;
    }
    return 0;
}

// all-reduce (max) of the ranks' scratch capacities over RCCL; every rank stores the agreed value
int agree_scratch_cap(spcbpt_comm* c) {
    int* d = reinterpret_cast<int*>(c->d_scalar);
    int own = c->scratch_cap, agreed = 0;
    HIPX(c, hipMemcpyAsync(d, &own, sizeof(int), hipMemcpyHostToDevice, c->xs));
    NCCLX(c, ncclAllReduce(d, d + 1, 1, ncclInt32, ncclMax, c->nccl, c->xs));
    HIPX(c, hipMemcpyAsync(&agreed, d + 1, sizeof(int), hipMemcpyDeviceToHost, c->xs));
    HIPX(c, hipStreamSynchronize(c->xs));
    c->scratch_cap = c->shard_cap = agreed;
    return 0;
}

}  // namespace

extern "C" {

int spcbpt_comm_unique_id(char id[SPCBPT_UNIQUE_ID_BYTES]) {
    if (!id) return SPCBPT_ERR_INVALID_ARG;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return SPCBPT_ERR_HIP;
    static_assert(sizeof(u.internal) == SPCBPT_UNIQUE_ID_BYTES, "unique id size");
    memcpy(id, u.internal, SPCBPT_UNIQUE_ID_BYTES);
    return SPCBPT_OK;
}

int spcbpt_comm_create(spcbpt_ctx* ctx, int rank, int world, const char id[SPCBPT_UNIQUE_ID_BYTES], spcbpt_comm** out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return SPCBPT_ERR_INVALID_ARG;
    *out = nullptr;
    spcbpt_comm* c = new spcbpt_comm();
    c->ctx = ctx; c->rank = rank; c->world = world;
    int rc = common_init(c);
    if (rc) { fprintf(stderr, "spcbpt_comm_create: %s\n", c->error.c_str()); delete c; return rc; }
    ncclUniqueId u;
    memcpy(u.internal, id, SPCBPT_UNIQUE_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->nccl, world, u, rank);
    if (r != ncclSuccess) { fprintf(stderr, "spcbpt_comm_create: ncclCommInitRank: %s\n", ncclGetErrorString(r)); delete c; return SPCBPT_ERR_HIP; }
    // ONE capacity for every rank: ncclAllGather needs equal send counts, and core_range gives the last rank the remainder
    // (1000 cores on 3 ranks: 333 / 333 / 334), so the ranks' own scratch sizes differ whenever num_core % world != 0
    rc = agree_scratch_cap(c);
    if (rc) { fprintf(stderr, "spcbpt_comm_create: %s\n", c->error.c_str()); (void)ncclCommDestroy(c->nccl); delete c; return rc; }
    *out = c;
    return SPCBPT_OK;
}

int spcbpt_comm_create_local(spcbpt_ctx* const* ctxs, int world, spcbpt_comm** out) {
    if (!ctxs || !out || world < 1) return SPCBPT_ERR_INVALID_ARG;
    LocalGroup* g = new LocalGroup();
    g->wants_tuple.assign(world, false);
    for (int r = 0; r < world; r++) {
        spcbpt_comm* c = new spcbpt_comm();
        c->ctx = ctxs[r]; c->rank = r; c->world = world; c->grp = g;
        int rc = common_init(c);
        if (rc) { fprintf(stderr, "spcbpt_comm_create_local: %s\n", c->error.c_str()); delete c; return rc; }
        g->ranks.push_back(c);
        out[r] = c;
    }
    g->refs = world;
    // one agreed capacity: the largest scratch of any rank
    int cap = 0;
    for (spcbpt_comm* c : g->ranks) cap = std::max(cap, c->scratch_cap);
    for (spcbpt_comm* c : g->ranks) c->shard_cap = c->scratch_cap = cap;
    return SPCBPT_OK;
}

int spcbpt_comm_destroy(spcbpt_comm* c) {
    if (!c) return SPCBPT_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (c->xs) (void)hipStreamSynchronize(c->xs);
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    if (c->d_gather) (void)hipFree(c->d_gather);
    if (c->d_counts_all) (void)hipFree(c->d_counts_all);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_send_counts) (void)hipFree(c->d_send_counts);
    if (c->h_counts_all) (void)hipHostFree(c->h_counts_all);
    if (c->d_pack) (void)hipFree(c->d_pack);
    if (c->d_pack_all) (void)hipFree(c->d_pack_all);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_scalar) (void)hipFree(c->d_scalar);
    if (c->l_ready) (void)hipEventDestroy(c->l_ready);
    if (c->xs) (void)hipStreamDestroy(c->xs);
    if (c->grp && --c->grp->refs == 0) delete c->grp;
    delete c;
    return SPCBPT_OK;
}

const char* spcbpt_comm_last_error(const spcbpt_comm* c) { return c ? c->error.c_str() : "null communicator"; }

int spcbpt_comm_set_shard_capacity(spcbpt_comm* c, int vertices) {
    if (!c || vertices < 1) return SPCBPT_ERR_INVALID_ARG;
    c->shard_cap = std::min(vertices, c->scratch_cap);
    return SPCBPT_OK;
}
int spcbpt_comm_get_shard_capacity(const spcbpt_comm* c, int* vertices) {
    if (!c || !vertices) return SPCBPT_ERR_INVALID_ARG;
    *vertices = c->shard_cap;
    return SPCBPT_OK;
}

int spcbpt_comm_exchange_lvc(spcbpt_comm* c) {
    if (!c) return SPCBPT_ERR_INVALID_ARG;
    HIPX(c, hipSetDevice(c->device));
    int rc = ensure_gather(c);
    if (rc) return rc;
    void *dv = nullptr, *dc = nullptr;
    int cap = 0;
    CTXX(c, spcbpt_lvc_export_on(c->ctx, c->xs, &dv, &dc, &cap));   // xs waits for the light pass on the device
    if (cap < 1) return fail(c, SPCBPT_ERR_STATE, "exchange_lvc: no light pass has been traced");
    if (cap < c->shard_cap) {
        // The caches are sized per rank from each rank's own probe pass, so this test comes out differently on different ranks; a
        // rank that refused here would leave the others blocked in the all-gather.  A shard is sent as shard_cap slots: a smaller
        // cache is staged through the communicator's own send buffer instead (the slots behind the cache are never read by the
        // import: every rank's count is <= its cache, and a count above shard_cap is refused by EVERY rank, from the gathered
        // counts).  spcbpt_comm_calibrate makes this branch dead by agreeing min(cache) with the shard capacity.
        rc = ensure_send(c, 1);
        if (rc) return rc;
        HIPX(c, hipMemcpyAsync(c->d_send, dv, (size_t)cap * kVertexBytes, hipMemcpyDeviceToDevice, c->xs));
        dv = c->d_send;
    }
    if (c->nccl) {
        NCCLX(c, ncclGroupStart());
        NCCLX(c, ncclAllGather(dc, c->d_counts_all, 2, ncclInt32, c->nccl, c->xs));
        NCCLX(c, ncclAllGather(dv, c->d_gather, (size_t)c->shard_cap * kVertexBytes, ncclUint8, c->nccl, c->xs));
        NCCLX(c, ncclGroupEnd());
        CTXX(c, spcbpt_lvc_import_gathered(c->ctx, c->d_gather, c->d_counts_all, c->world, c->shard_cap, c->xs));
        return SPCBPT_OK;
    }
    LocalGroup* g = c->grp;
    std::lock_guard<std::mutex> lk(g->mu);
    if (c->l_posted) return fail(c, SPCBPT_ERR_STATE, "exchange_lvc (local): this rank already posted; every rank must call before any calls again");
    c->l_send = dv; c->l_counts = dc; c->l_posted = true;
    PostGuard guard(g);                                               // a failure below un-posts every rank and frees the events
    HIPX(c, hipEventRecord(c->l_ready, c->xs));                       // behind the wait for the light pass
    for (spcbpt_comm* s : g->ranks) if (!s->l_posted) { guard.keep(); return SPCBPT_OK; }   // completed by the last rank to post
    for (spcbpt_comm* s : g->ranks) { int r2 = ensure_gather(s); if (r2) return r2; }
    // copies first (on every destination's stream), then the imports -- which overwrite the send buffers -- behind all of them
    for (spcbpt_comm* d : g->ranks) {
        for (spcbpt_comm* s : g->ranks) {
            HIPX(d, hipStreamWaitEvent(d->xs, s->l_ready, 0));
            HIPX(d, hipMemcpyAsync(reinterpret_cast<char*>(d->d_gather) + (size_t)s->rank * d->shard_cap * kVertexBytes, s->l_send,
                                   (size_t)d->shard_cap * kVertexBytes, hipMemcpyDeviceToDevice, d->xs));
            HIPX(d, hipMemcpyAsync(d->d_counts_all + 2 * s->rank, s->l_counts, 2 * sizeof(int), hipMemcpyDeviceToDevice, d->xs));
        }
    }
    std::vector<hipEvent_t>& copied = guard.events;
    copied.assign(g->ranks.size(), nullptr);
    for (size_t k = 0; k < g->ranks.size(); k++) {
        HIPX(c, hipEventCreateWithFlags(&copied[k], hipEventDisableTiming));
        HIPX(c, hipEventRecord(copied[k], g->ranks[k]->xs));
    }
    int result = SPCBPT_OK;
    for (spcbpt_comm* d : g->ranks) {
        for (size_t k = 0; k < copied.size(); k++) (void)hipStreamWaitEvent(d->xs, copied[k], 0);
        int r2 = spcbpt_lvc_import_gathered(d->ctx, d->d_gather, d->d_counts_all, d->world, d->shard_cap, d->xs);
        if (r2 && !result) result = fail(c, r2, std::string("spcbpt_lvc_import_gathered: ") + spcbpt_last_error(d->ctx));
    }
    return result;   // the guard un-posts every rank and destroys the events
}

// One exchange per light batch: the shards of the n oldest pending passes travel as ONE all-gather (plus one of the count pairs),
// and ONE kernel concatenates every frame's shards into that frame's set.  Per-frame exchanges put 2 n small collectives in front of
// an n-frame eye launch; the batch's passes are one launch already (spcbpt_launch_light_batch), so their shards are ready together.
int spcbpt_comm_exchange_lvc_batch(spcbpt_comm* c, int n) {
    if (!c || n < 1 || n > kMaxFrames) return SPCBPT_ERR_INVALID_ARG;
    HIPX(c, hipSetDevice(c->device));
    int rc = ensure_gather(c, n);
    if (!rc) rc = ensure_send(c, n);
    if (rc) return rc;
    int lvc_cap = 0;
    CTXX(c, spcbpt_lvc_get_capacity(c->ctx, &lvc_cap, nullptr));
    if (lvc_cap < 1) return fail(c, SPCBPT_ERR_STATE, "exchange_lvc_batch: no light pass has been traced");
    CTXX(c, spcbpt_lvc_export_batch_on(c->ctx, c->xs, n, c->d_send, c->d_send_counts, c->shard_cap));   // xs waits for the passes on the device, then packs
    if (c->nccl) {
        NCCLX(c, ncclGroupStart());
        NCCLX(c, ncclAllGather(c->d_send_counts, c->d_counts_all, (size_t)2 * n, ncclInt32, c->nccl, c->xs));
        NCCLX(c, ncclAllGather(c->d_send, c->d_gather, (size_t)n * c->shard_cap * kVertexBytes, ncclUint8, c->nccl, c->xs));
        NCCLX(c, ncclGroupEnd());
        CTXX(c, spcbpt_lvc_import_gathered_batch(c->ctx, c->d_gather, c->d_counts_all, c->world, n, c->shard_cap, c->xs));
        return SPCBPT_OK;
    }
    LocalGroup* g = c->grp;
    std::lock_guard<std::mutex> lk(g->mu);
    if (c->l_posted) return fail(c, SPCBPT_ERR_STATE, "exchange_lvc_batch (local): this rank already posted; every rank must call before any calls again");
    c->l_send = c->d_send; c->l_counts = c->d_send_counts; c->l_posted = true; c->l_frames = n;
    PostGuard guard(g);                                               // a failure below un-posts every rank and frees the events
    HIPX(c, hipEventRecord(c->l_ready, c->xs));
    for (spcbpt_comm* s : g->ranks) if (!s->l_posted) { guard.keep(); return SPCBPT_OK; }   // completed by the last rank to post
    for (spcbpt_comm* s : g->ranks) if (s->l_frames != n) return fail(c, SPCBPT_ERR_STATE, "exchange_lvc_batch (local): the ranks posted different frame counts");
    for (spcbpt_comm* s : g->ranks) { int r2 = ensure_gather(s, n); if (r2) return r2; }
    const size_t block = (size_t)n * c->shard_cap * kVertexBytes;
    for (spcbpt_comm* d : g->ranks) {
        for (spcbpt_comm* s : g->ranks) {
            HIPX(d, hipStreamWaitEvent(d->xs, s->l_ready, 0));
            HIPX(d, hipMemcpyAsync(reinterpret_cast<char*>(d->d_gather) + (size_t)s->rank * block, s->l_send, block, hipMemcpyDeviceToDevice, d->xs));
            HIPX(d, hipMemcpyAsync(d->d_counts_all + (size_t)2 * n * s->rank, s->l_counts, (size_t)2 * n * sizeof(int), hipMemcpyDeviceToDevice, d->xs));
        }
    }
    // (the send buffers are the communicators' own: a rank's next pack waits on its own stream behind these copies only if it is
    // the destination too, so every destination records an event the senders' streams wait for)
    std::vector<hipEvent_t>& copied = guard.events;
    copied.assign(g->ranks.size(), nullptr);
    for (size_t k = 0; k < g->ranks.size(); k++) {
        HIPX(c, hipEventCreateWithFlags(&copied[k], hipEventDisableTiming));
        HIPX(c, hipEventRecord(copied[k], g->ranks[k]->xs));
    }
    int result = SPCBPT_OK;
    for (spcbpt_comm* d : g->ranks) {
        for (size_t k = 0; k < copied.size(); k++) (void)hipStreamWaitEvent(d->xs, copied[k], 0);
        int r2 = spcbpt_lvc_import_gathered_batch(d->ctx, d->d_gather, d->d_counts_all, d->world, n, d->shard_cap, d->xs);
        if (r2 && !result) result = fail(c, r2, std::string("spcbpt_lvc_import_gathered_batch: ") + spcbpt_last_error(d->ctx));
    }
    return result;   // the guard un-posts every rank and destroys the events
}

int spcbpt_comm_info(const spcbpt_comm* c, int* rank, int* world, int* transport) {
    if (!c) return SPCBPT_ERR_INVALID_ARG;
    int r = c->rank, w = c->world;
    if (c->nccl) {   // what RCCL itself says (bench.py reports it: a scaling number must come from the ranks RCCL really connected)
        if (ncclCommUserRank(c->nccl, &r) != ncclSuccess || ncclCommCount(c->nccl, &w) != ncclSuccess) return SPCBPT_ERR_HIP;
    }
    if (rank) *rank = r;
    if (world) *world = w;
    if (transport) *transport = c->nccl ? SPCBPT_COMM_RCCL : SPCBPT_COMM_LOCAL;
    return SPCBPT_OK;
}

int spcbpt_comm_calibrate(spcbpt_comm* c, int passes, uint32_t first_frame, float slack) {
    if (!c || passes < 1 || !(slack >= 1.0f)) return SPCBPT_ERR_INVALID_ARG;
    HIPX(c, hipSetDevice(c->device));
    int own_max = 0;
    for (int k = 0; k < passes; k++) {
        CTXX(c, spcbpt_launch(c->ctx, "light trace", first_frame + (uint32_t)k, 0, 0, 1));
        int n = 0;
        CTXX(c, spcbpt_lvc_read(c->ctx, nullptr, 0, &n));   // host wait: start-up only
        own_max = std::max(own_max, n);
    }
    // a shard is SENT out of the rank's cache, so it can be no larger than the smallest cache of any rank (the caches are sized
    // from each rank's own probe pass: spcbpt_lvc_set_capacity) -- agreed together with the largest shard
    int own_lvc = 0;
    CTXX(c, spcbpt_lvc_get_capacity(c->ctx, &own_lvc, nullptr));
    int global_max = own_max, min_lvc = own_lvc;
    if (c->nccl) {
        int* d = reinterpret_cast<int*>(c->d_scalar);   // 2 doubles = 4 ints: {max shard, -cache} -> max
        const int in[2] = {own_max, -own_lvc};
        int outv[2] = {0, 0};
        HIPX(c, hipMemcpyAsync(d, in, sizeof(in), hipMemcpyHostToDevice, c->xs));
        NCCLX(c, ncclAllReduce(d, d + 2, 2, ncclInt32, ncclMax, c->nccl, c->xs));
        HIPX(c, hipMemcpyAsync(outv, d + 2, sizeof(outv), hipMemcpyDeviceToHost, c->xs));
        HIPX(c, hipStreamSynchronize(c->xs));
        global_max = outv[0]; min_lvc = -outv[1];
    } else {   // local: the ranks calibrate one after the other; the agreed capacity follows the largest shard seen so far
        std::lock_guard<std::mutex> lk(c->grp->mu);
        if (c->grp->calib_calls % c->world == 0) c->grp->calib_max = c->grp->calib_min_lvc = 0;   // a new round: the caches may have grown since the last one
        c->grp->calib_calls++;
        c->grp->calib_max = std::max(c->grp->calib_max, own_max);
        c->grp->calib_min_lvc = c->grp->calib_min_lvc == 0 ? own_lvc : std::min(c->grp->calib_min_lvc, own_lvc);
        global_max = c->grp->calib_max; min_lvc = c->grp->calib_min_lvc;
    }
    int cap = (int)((double)global_max * slack) + 1;
    cap = (cap + 1023) / 1024 * 1024;
    cap = std::max(1, std::min(std::max(1024, std::min(cap, c->scratch_cap)), min_lvc));
    if (c->nccl) c->shard_cap = cap;
    else {
        std::lock_guard<std::mutex> lk(c->grp->mu);
        for (spcbpt_comm* s : c->grp->ranks) s->shard_cap = cap;
    }
    return SPCBPT_OK;
}

int spcbpt_comm_gather_film(spcbpt_comm* c, void* out_device) {
    if (!c) return SPCBPT_ERR_INVALID_ARG;
    HIPX(c, hipSetDevice(c->device));
    size_t floats = 0;
    int rc = film_block_floats(c, &floats);
    if (rc) return rc;
    rc = ensure_pack(c, floats);
    if (rc) return rc;
    void* accum = nullptr;
    CTXX(c, spcbpt_accum_device_ptr(c->ctx, &accum));
    void* out = out_device ? out_device : accum;
    CTXX(c, spcbpt_film_pack_bands(c->ctx, c->rank, c->world, c->d_pack, c->xs));   // waits for this rank's render streams
    if (c->nccl) {
        NCCLX(c, ncclAllGather(c->d_pack, c->d_pack_all, floats, ncclFloat, c->nccl, c->xs));
        CTXX(c, spcbpt_film_unpack_bands(c->ctx, c->world, c->d_pack_all, out, c->xs));
        HIPX(c, hipStreamSynchronize(c->xs));
        return SPCBPT_OK;
    }
    LocalGroup* g = c->grp;
    std::lock_guard<std::mutex> lk(g->mu);
    c->l_film_posted = true; c->l_film_out = out;
    HIPX(c, hipEventRecord(c->l_ready, c->xs));
    for (spcbpt_comm* s : g->ranks) if (!s->l_film_posted) return SPCBPT_OK;   // the image is complete when the last rank has called
    for (spcbpt_comm* d : g->ranks) {
        for (spcbpt_comm* s : g->ranks) {
            HIPX(d, hipStreamWaitEvent(d->xs, s->l_ready, 0));
            HIPX(d, hipMemcpyAsync(d->d_pack_all + (size_t)s->rank * floats, s->d_pack, floats * sizeof(float), hipMemcpyDeviceToDevice, d->xs));
        }
        int r2 = spcbpt_film_unpack_bands(d->ctx, d->world, d->d_pack_all, d->l_film_out, d->xs);
        if (r2) return fail(c, r2, std::string("spcbpt_film_unpack_bands: ") + spcbpt_last_error(d->ctx));
    }
    for (spcbpt_comm* d : g->ranks) { HIPX(d, hipStreamSynchronize(d->xs)); d->l_film_posted = false; }
    return SPCBPT_OK;
}

int spcbpt_comm_broadcast_subspace(spcbpt_comm* c, int root) {
    if (!c || root < 0 || root >= c->world) return SPCBPT_ERR_INVALID_ARG;
    HIPX(c, hipSetDevice(c->device));
    const size_t NS = SPCBPT_NUM_SUBSPACE;
    std::vector<spcbpt_tree_node> et, lt;
    std::vector<float> q(NS), g(NS * NS);
    int ne = 0, nl = 0;
    if (c->rank == root) {
        CTXX(c, spcbpt_get_subspace(c->ctx, nullptr, &ne, 0, nullptr, &nl, 0, nullptr, nullptr));
        et.resize(ne); lt.resize(nl);
        CTXX(c, spcbpt_get_subspace(c->ctx, et.data(), &ne, ne, lt.data(), &nl, nl, q.data(), g.data()));
    }
    if (c->nccl) {
        int sizes[2] = {ne, nl};
        int* d = reinterpret_cast<int*>(c->d_scalar);
        HIPX(c, hipMemcpyAsync(d, sizes, sizeof(sizes), hipMemcpyHostToDevice, c->xs));
        NCCLX(c, ncclBroadcast(d, d, 2, ncclInt32, root, c->nccl, c->xs));
        HIPX(c, hipMemcpyAsync(sizes, d, sizeof(sizes), hipMemcpyDeviceToHost, c->xs));
        HIPX(c, hipStreamSynchronize(c->xs));
        ne = sizes[0]; nl = sizes[1];
        if (ne < 1 || nl < 1) return fail(c, SPCBPT_ERR_STATE, "broadcast_subspace: the root has no tuple");
        const size_t b_et = (size_t)ne * sizeof(spcbpt_tree_node), b_lt = (size_t)nl * sizeof(spcbpt_tree_node), b_q = NS * 4, b_g = NS * NS * 4;
        const size_t total = b_et + b_lt + b_q + b_g;
        int rc = ensure_stage(c, total);
        if (rc) return rc;
        char* s = reinterpret_cast<char*>(c->d_stage);
        if (c->rank == root) {
            HIPX(c, hipMemcpyAsync(s, et.data(), b_et, hipMemcpyHostToDevice, c->xs));
            HIPX(c, hipMemcpyAsync(s + b_et, lt.data(), b_lt, hipMemcpyHostToDevice, c->xs));
            HIPX(c, hipMemcpyAsync(s + b_et + b_lt, q.data(), b_q, hipMemcpyHostToDevice, c->xs));
            HIPX(c, hipMemcpyAsync(s + b_et + b_lt + b_q, g.data(), b_g, hipMemcpyHostToDevice, c->xs));
        }
        NCCLX(c, ncclBroadcast(s, s, total, ncclUint8, root, c->nccl, c->xs));
        if (c->rank != root) {
            et.resize(ne); lt.resize(nl);
            HIPX(c, hipMemcpyAsync(et.data(), s, b_et, hipMemcpyDeviceToHost, c->xs));
            HIPX(c, hipMemcpyAsync(lt.data(), s + b_et, b_lt, hipMemcpyDeviceToHost, c->xs));
            HIPX(c, hipMemcpyAsync(q.data(), s + b_et + b_lt, b_q, hipMemcpyDeviceToHost, c->xs));
            HIPX(c, hipMemcpyAsync(g.data(), s + b_et + b_lt + b_q, b_g, hipMemcpyDeviceToHost, c->xs));
        }
        HIPX(c, hipStreamSynchronize(c->xs));
        if (c->rank != root) CTXX(c, spcbpt_set_subspace(c->ctx, et.data(), ne, lt.data(), nl, q.data(), g.data()));
        return SPCBPT_OK;
    }
    LocalGroup* grp = c->grp;
    std::lock_guard<std::mutex> lk(grp->mu);
    if (c->rank == root) {
        grp->et = et; grp->lt = lt; grp->q = q; grp->g = g; grp->have_tuple = true;
        for (spcbpt_comm* s : grp->ranks)
            if (grp->wants_tuple[s->rank]) {
                int r2 = spcbpt_set_subspace(s->ctx, grp->et.data(), (int)grp->et.size(), grp->lt.data(), (int)grp->lt.size(), grp->q.data(), grp->g.data());
                if (r2) return fail(c, r2, std::string("spcbpt_set_subspace: ") + spcbpt_last_error(s->ctx));
                grp->wants_tuple[s->rank] = false;
            }
    } else if (grp->have_tuple) {
        CTXX(c, spcbpt_set_subspace(c->ctx, grp->et.data(), (int)grp->et.size(), grp->lt.data(), (int)grp->lt.size(), grp->q.data(), grp->g.data()));
    } else {
        grp->wants_tuple[c->rank] = true;   // installed when the root calls
    }
    return SPCBPT_OK;
}

int spcbpt_comm_barrier(spcbpt_comm* c) {
    if (!c) return SPCBPT_ERR_INVALID_ARG;
    HIPX(c, hipSetDevice(c->device));
    if (c->nccl) {
        float* d = reinterpret_cast<float*>(c->d_scalar);
        NCCLX(c, ncclAllReduce(d, d + 1, 1, ncclFloat, ncclSum, c->nccl, c->xs));
    }
    HIPX(c, hipStreamSynchronize(c->xs));
    return SPCBPT_OK;
}

int spcbpt_comm_max_double(spcbpt_comm* c, double* value) {
    if (!c || !value) return SPCBPT_ERR_INVALID_ARG;
    HIPX(c, hipSetDevice(c->device));
    if (c->nccl) {
        HIPX(c, hipMemcpyAsync(c->d_scalar, value, sizeof(double), hipMemcpyHostToDevice, c->xs));
        NCCLX(c, ncclAllReduce(c->d_scalar, c->d_scalar + 1, 1, ncclDouble, ncclMax, c->nccl, c->xs));
        HIPX(c, hipMemcpyAsync(value, c->d_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, c->xs));
        HIPX(c, hipStreamSynchronize(c->xs));
        return SPCBPT_OK;
    }
    std::lock_guard<std::mutex> lk(c->grp->mu);   // local: the running maximum (exact once every rank has called)
    if (c->grp->max_calls % c->world == 0) c->grp->running_max = *value;
    c->grp->running_max = std::max(c->grp->running_max, *value);
    c->grp->max_calls++;
    *value = c->grp->running_max;
    return SPCBPT_OK;
}

}  // extern "C"

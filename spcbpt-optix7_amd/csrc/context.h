// Per-GPU context: owns every HBM buffer of the path (the reference leaves ownership to function-static
// thrust vectors and never frees, device_thrust.cu:287-293).
#pragma once
#include <hip/hip_runtime.h>

#include <deque>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "layout.h"

namespace spc {

struct Preprocessor;

struct TimedSpan {
    std::string name;
    hipEvent_t a, b;
    hipStream_t s;
};

struct Context {
    int device = 0;
    hipStream_t stream = nullptr;    // light pass, sampler build, preprocessing, uploads
    // Render launches ("pt", "SPCBPT_eye") run on their own stream.  The eye megakernel ends with a drain phase (paths of up
    // to 50 bounces finishing with ever fewer live lanes: the average wave has left after 83 % of the kernel span), and the
    // light pass + sampler build of the NEXT frame do not depend on it.  With the sampler tables double-buffered, the host
    // loop launch("light trace") -> build_sampler -> launch("SPCBPT_eye") keeps its meaning while frame f+1's light pass
    // fills the idle machine under frame f's drain.  SPCBPT_OVERLAP=0 puts everything back on one stream.
    hipStream_t rstream = nullptr;              // = rstreams[rk], the stream of the render launch being issued
    hipStream_t cstream = nullptr;              // spcbpt_read_film's copies (created on first use)
    // Consecutive render launches alternate between two streams, each with its own work counter, spill area and `result`
    // buffer: frame f+1's eye kernel starts filling the machine while frame f's drains.  Only the film merges (running mean +
    // tone map from `result` into accum/frame) are chained, by ev_merge, so the image is the same as with one stream.
    // n_render streams are in use (default 2; SPCBPT_RENDER_STREAMS=1..8).  More frames in flight pay when one frame does not
    // fill the GPU -- a rank of an 8-GPU job renders 1/8 of the image but its frame still ends with the same 50-bounce chain.
    static const int kMaxRender = 8, kMaxSets = 128;
    int n_render = 2, n_sets = 6;   // n_sets = n_render + 4: one set per eye kernel in flight + the light passes ahead of them
    hipStream_t rstreams[kMaxRender] = {};
    int rk = 0, last_merge_k = -1;
    float* d_result[kMaxRender] = {};
    hipEvent_t ev_merge[kMaxRender] = {};
    bool ev_merge_set[kMaxRender] = {};
    hipEvent_t ev_sampler[kMaxSets] = {}, ev_render[kMaxSets] = {};
    bool ev_sampler_set[kMaxSets] = {}, ev_render_set[kMaxSets] = {};
    // the event that marks the end of the last eye kernel that read the set: its own, or -- after a batched eye launch -- the ONE event
    // recorded for the whole batch (that of the batch's last set).  32 event records between a batch's eye kernel and its film merge
    // were 0.16 ms of idle GPU per launch (kernel trace of a long run).  Waiting on an event that has meanwhile been recorded again
    // waits for a LATER eye kernel -- more than needed, never a cycle: everything that kernel depends on was queued before it.
    int render_event_of[kMaxSets] = {};
    hipEvent_t ev_set_stream[kMaxSets] = {};   // last work queued on `stream` that reads or writes the set (sampler build, import copy):
    bool ev_set_touched[kMaxSets] = {};        // a light pass on the second lane waits for it before it rewrites the set
    int lset = 0, eset = 0;  // buffer set of the latest light pass, and of the sampler eye launches use
    // Light passes whose sampler has not been built yet, oldest first.  The host loop of a single GPU alternates
    // light pass -> sampler build, so the queue holds one set; a sharded job launches the NEXT frame's light pass before it
    // exchanges and builds the current one (the light pass is a 1 ms dependent chain however few paths it traces, and the
    // exchange makes the host wait for it), so export / import / build_sampler always address the OLDEST pending set.
    std::deque<int> pending;
    bool light_ahead = false;                // spcbpt_set_light_ahead: keep older unbuilt passes queued (default: only the latest)
    hipEvent_t ev_light[kMaxSets] = {};      // light pass + compaction of the set done (recorded on `stream`)
    int set_count_host[kMaxSets];            // vertex count of the set when the host knows it (after an import), else -1
    bool light_counts_valid[kMaxSets] = {};  // h_light_counts of the set describe its current contents
    int keys_set = -1;                       // the set whose compaction left d_keys / d_vals / d_weights (valid if keys_ready)
    // Sharded job without host round trips (spcbpt_lvc_export_on / spcbpt_lvc_import_gathered, libspcbpt_mgpu): the all-gathered
    // shards are compacted on the exchange stream with the totals left on the DEVICE; the sampler build then runs over a host-known
    // upper bound (`set_bound`, pad keys sort behind the real items) instead of reading the count back.
    int set_bound[kMaxSets];                 // >= 0: the set's vertex count lives on the device only, this is its upper bound
    hipEvent_t ev_exch[kMaxSets] = {};       // gathered compaction of the set done (recorded on the caller's exchange stream)
    bool ev_exch_set[kMaxSets] = {};
    int export_on(hipStream_t xs, void** dv, void** dc, int* cap);
    int import_gathered(const void* shards, const int* counts_all, int world, int shard_cap, hipStream_t xs, int nf);
    int export_batch_on(hipStream_t xs, int nf, void* send, int* send_counts, int shard_cap);   // one exchange per light batch
    hipEvent_t ev_import[2] = {};            // device-to-device import copies done (alternating: the caller alternates two staging buffers)
    long long import_gen = 0;
    int* h_import_counts = nullptr;          // pinned [kMaxSets][2]: source of the counts upload of an import (no host wait)
    int* h_light_counts = nullptr;           // pinned [kMaxSets][2]: (vertex_count, path_count) of a light pass, written on `stream`
                                             // before ev_light -- the host reads them after waiting for that event only
    int build_set() const { return pending.empty() ? lset : pending.front(); }
    // Second light lane.  On one in-order stream light pass, compaction, import copy and sampler build of consecutive frames
    // form a chain of ~1.3 ms per frame -- more than a rank's share of the eye pass costs when the frame is sharded eight ways.
    // With light passes running ahead, every other pass (kernel + compaction) therefore goes to a second stream with its own
    // scratch, counts, key and temp buffers; imports and sampler builds stay on `stream` and wait for the pass's event.
    hipStream_t lstream_b = nullptr;
    int light_toggle = 0;
    int light_lane_of_set[kMaxSets] = {};    // which lane's stream last wrote the set (its ev_light orders it)
    LightVertex* b_scratch = nullptr; size_t b_scratch_capacity = 0;
    int *b_core_counts = nullptr, *b_core_offsets = nullptr; size_t b_counts_capacity = 0;
    uint32_t *b_keys = nullptr, *b_vals = nullptr; float* b_weights = nullptr; size_t b_keys_capacity = 0;
    unsigned char* b_temp = nullptr; size_t b_temp_capacity = 0;
    uint32_t* b_spill = nullptr; size_t b_spill_capacity = 0;
    int ensure_lane_b();
    // Batched light pass (spcbpt_launch_light_batch): the passes of n consecutive launch frames as ONE persistent launch on the
    // second lane, each into its own set.  A rank of an 8-GPU job traces 1/8 of the cores per frame; its pass is then a ~1.2 ms
    // chain of 50 dependent bounces that the few block slots beside the eye grid run one after the other -- eight of them per eye
    // batch cost more than the eye batch itself.  In one queue they regenerate like one pass of eight times the cores.
    LightVertex* lb_scratch = nullptr; size_t lb_scratch_capacity = 0;     // n * core_count * core_padding
    int *lb_core_counts = nullptr, *lb_core_offsets = nullptr, *lb_path_counts = nullptr; size_t lb_counts_capacity = 0;   // n * (core_count + 1)
    uint32_t* lb_spill = nullptr; size_t lb_spill_capacity = 0;
    int launch_light_batch(uint32_t first_frame, int n);
    int* d_set_counts_all = nullptr;         // one allocation behind set_counts[]: set s at + 2 s (a batch copies its sets' counts to the host as ranges)
    LightVertex* set_lvc[kMaxSets] = {};
    LightVertex* set_lvc_sorted[kMaxSets] = {};   // the set's cache in its sampler's order (written by the sampler build, read by the eye megakernel)
    uint32_t* set_vals2[kMaxSets] = {};
    float* set_cmfs[kMaxSets] = {};
    uint32_t* set_guide[kMaxSets] = {};   // the set's second-stage guide table (layout.h KParams::guide)
    DSubspace* set_subspace[kMaxSets] = {};
    int* set_counts[kMaxSets] = {};
    uint32_t* d_spill_rs[kMaxRender] = {};   // traversal-stack spill areas of the render streams (d_spill serves `stream`)
    size_t spill_rs_capacity[kMaxRender] = {};
    int sync_all();
    void select_set(int s);
    std::string error;
    KParams kp;
    // scene
    float* d_nodes = nullptr;
    float* d_nodes_q = nullptr;            // the same nodes in the quad-lane layout (quad_trace.hip), built on first use
    float* d_nodes_q2 = nullptr;           // ... with the scale exponents as signed bytes (the lean quad kernel)
    float* d_tris = nullptr;
    int n_paired = 0;                      // triangles that are half of a fan pair (lbvh.h): what the pooled pass tests two at a time
    int32_t* d_tri_orig = nullptr;
    DMaterial* d_mats = nullptr;
    DLight* d_lights = nullptr;
    DTexture* d_tex = nullptr;
    std::vector<uint32_t*> d_tex_data;
    int n_triangles = 0, n_nodes = 0, bvh_depth = 0, n_lights = 0, n_mats = 0;
    // environment map (params.sky; spcbpt_set_environment): device copies of the flipped texture and the sampling CMF
    float* d_env_tex = nullptr;
    float* d_env_cmf = nullptr;
    std::vector<DLight> h_lights;          // the light list as uploaded (the ENV light is appended to it)
    float bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};   // of every vertex handed to spcbpt_create (+ the light quads)
    int set_environment(const float* rgba, int w, int h, const float* center, float radius);
    // film
    float* d_accum = nullptr;
    uint32_t* d_frame = nullptr;
    bool have_camera = false;
    // subspace tuple
    float* d_eye_tree = nullptr;
    float* d_light_tree = nullptr;
    float* d_Q = nullptr;
    float* d_gamma = nullptr;
    float* d_gamma2 = nullptr;   // three-level copy of d_gamma (layout.h: CMF2_ROW)
    float* d_gamma_q = nullptr;     // Gamma / Q table (layout.h: KParams::gamma_q)
    uint16_t* d_guide1 = nullptr;   // first-stage guide table (layout.h: KParams::cmf_guide1)
    bool gamma_monotone = false; // every row of the installed matrix is a proper CMF: first-stage sampling may count instead of bisect
    std::vector<spcbpt_tree_node> h_eye_tree, h_light_tree;
    std::vector<float> h_Q, h_gamma;
    bool have_subspace = false;
    bool tree_has_direction = false;   // a caller-supplied classifier with direction nodes (type 2): labels depend on the viewing
                                       // direction, so the label-caching kernels do not apply (device_lib.h) and the generic
                                       // (counting) instantiations run instead
    // light pass + LVC + sampler
    spcbpt_light_trace_params lt = {100000, 52, 1, 0, 100000, 1};
    LightVertex* d_scratch = nullptr;
    size_t scratch_capacity = 0;
    int* d_core_counts = nullptr;
    int* d_core_offsets = nullptr;
    size_t counts_capacity = 0;
    LightVertex* d_lvc = nullptr;
    // Vertices every buffer set (compact LVC, jump buffer, CMFs) holds.  NOT the padded worst case num_core x core_padding (5.2 M
    // vertices = 541 MB per set at the bench geometry, 53 GB for the 99 sets of a 32-frame pipeline): a light pass fills ~5 % of
    // its padded slots, so the sets are sized from a PROBE pass -- the first light pass after spcbpt_set_light_trace is traced
    // once more into the padded scratch and counted on the host (start-up) -- as 2 x that count (scaled to the whole job's cores
    // for a rank of a sharded job), at most the worst case.  The compaction kernels cut a pass off at the capacity and raise
    // diag[2] (SPCBPT_ERR_CAPACITY at the next sync): the vertex count of 100 000 paths varies by a fraction of a per cent, so
    // the flag means a changed scene or tuple, and spcbpt_lvc_set_capacity fixes the size by hand.
    size_t lvc_capacity = 0;
    size_t lvc_fixed = 0;            // spcbpt_lvc_set_capacity / SPCBPT_LVC_CAPACITY: explicit capacity (0 = from the probe pass)
    bool lvc_probe_needed = false;   // set by set_light_trace, consumed by the next light pass
    int probe_lvc_capacity();
    uint32_t *d_keys = nullptr, *d_keys2 = nullptr, *d_vals = nullptr, *d_vals2 = nullptr;
    float* d_weights = nullptr;
    double *d_wsorted = nullptr, *d_prefix = nullptr;
    float* d_cmfs = nullptr;
    DSubspace* d_subspace = nullptr;
    int* d_sampler_counts = nullptr;  // [0] vertex_count, [1] path_count
    int* d_hist = nullptr;            // per-block histograms / offsets of the four-launch sampler build (kernels.hip)
    bool counting_build = true;       // SPCBPT_SAMPLER_BUILD=hipcub selects the radix-sort form (same tables)
    int lvc_count = 0, path_count = 0;
    bool keys_ready = false, have_sampler = false;
    // scratch
    unsigned char* d_temp = nullptr;
    size_t temp_capacity = 0;
    uint32_t* d_spill = nullptr;
    size_t spill_capacity = 0;
    uint32_t* d_diag = nullptr;        // KParams::diag: [0] dropped traversal-stack entries, [1] shard / gathered-cache overflow of exchange 1, [2] a light pass outgrew the set capacity
    int spill_entries_debug = -1;      // SPCBPT_DEBUG_SPILL_ENTRIES: caps the spill entries per thread (tests of the overflow report)
    int spill_entries_needed() const;  // 3 * bvh_depth - kStackLds (a 4-wide node pushes up to 3 children per level)
    int check_diag();                  // after a sync: SPCBPT_ERR_STATE if a kernel dropped stack entries since the last check
    // instrumentation
    uint32_t* d_work_counter = nullptr;
    int num_cus = 0, blocks_per_cu[3] = {0, 0, 0};   // per kernel variant (single-frame launches), of the instantiation launched
    int blocks_per_cu_batch = 0;                       // the batched timed kernel's
    int grid_percent = 0;              // persistent grid as a share of the resident block slots; 0 = 94 with several render streams, else 100 (launch_render)
    int light_blocks_wide = -1;        // ... of a pass that has the GPU to itself (no passes ahead): a lane per core, at most four blocks per CU (SPCBPT_LIGHT_BLOCKS_WIDE)
    int light_blocks = -1;             // persistent grid of the light pass (SPCBPT_LIGHT_BLOCKS; default: one block per CU)
    int light_batch_blocks = -1;       // ... of a batched light pass (SPCBPT_LIGHT_BATCH_BLOCKS; 0 = in proportion to the light paths per pixel, >= 16: launch_light_batch)
    int tiles_per_wave = 1;            // lower bound of 8x8 tiles per persistent wave (SPCBPT_TILES_PER_WAVE)
    unsigned long long* d_counters = nullptr;
    bool counting = false, timing = false;
    bool count_executed = false;       // spcbpt_enable_counters(ctx, 2): count with the TIMED kernels' instantiations (label caching) instead of the reference's order
    int kernel_variant() const { return tree_has_direction ? 1 : (counting ? (count_executed ? 2 : 1) : 0); }   // kernels.h: launch_spcbpt
    std::vector<TimedSpan> spans;
    std::map<std::string, std::pair<double, int>> times;

    ~Context();
    void time_begin(const char* name, hipStream_t s = nullptr);
    void time_end();
    void resolve_spans();
    int ensure_spill(size_t threads, bool render = false);
    int ensure_temp(size_t bytes);
    int ensure_lvc_capacity(size_t n);
    int upload_tree(const spcbpt_tree_node* t, int n, float*& d_tree, std::vector<spcbpt_tree_node>& host_copy);
    int install_subspace(const spcbpt_tree_node* et, int ne, const spcbpt_tree_node* lt, int nl, const float* q, const float* g);
    int install_minimal_tuple();
    int set_light_trace(const spcbpt_light_trace_params& p);
    int launch_light(uint32_t frame);
    int fetch_counts();           // counts of the latest light pass's set (lset) -> lvc_count, path_count
    int fetch_counts_of(int set);
    int build_sampler();
    int build_sampler_batch(int n);   // the n oldest pending passes in one set of four launches (capi.hip)
    uint32_t* sbb_keys = nullptr; float* sbb_weights = nullptr; double* sbb_wsorted = nullptr; int* sbb_hist = nullptr;   // its scratch: per frame what d_keys .. d_hist are
    int sbb_frames = 0; size_t sbb_capacity = 0;   // frames x items per frame it holds
    int sbb_fallbacks = 0;                          // batches built one by one because the scratch could not be allocated
    size_t sbb_refused_bytes = 0;                   // the smallest scratch size the device has refused (0: none): not asked for again until the capacity or the mode changes
    void free_batch_build_scratch();
    size_t sbb_debug_limit() const;
    // the tables of the last sampler build are still what that build left (no later pass, import or re-allocation took the set)
    bool sampler_intact() const { return !built_sets.empty() && built_sets.back() == eset && ev_sampler_set[eset]; }
    int launch_render(const char* name, bool spcbpt_alg, uint32_t frame, int r0, int r1, int rs, bool full_mis = false, bool defer_merge = false);
    // spcbpt_launch_deferred: a render launch whose film merge (running mean + tone map from its `result` buffer) has not been queued:
    // the frame is either merged later (merge_deferred(true)) or never (false) -- the interactive loop's speculative next frame
    struct Deferred { bool active = false; int rk = 0; uint32_t subframe = 0; int row_begin = 0, row_end = 0, row_step = 1; float* result = nullptr; } deferred;
    int merge_deferred(bool keep);
    int sync_film();
    // Batched eye launch (spcbpt_launch_eye_batch): the last n built samplers, one per frame, rendered by ONE persistent kernel
    // whose tile queue spans the frames (kernels.hip: BATCH).  A rank's share of a sharded frame is a few thousand tiles --
    // about one per resident wave, i.e. all drain phase; four frames in one queue regenerate like one frame four times the size.
    std::deque<int> built_sets;                                  // sets of the most recent sampler builds, oldest first
    int eye_batch = 1;                                           // frames per batched launch the context is sized for (SPCBPT_EYE_BATCH)
    float* d_result_b[kMaxRender][kMaxBatchFrames] = {};         // per render stream and frame slot: radiance of that frame
    FrameDesc* d_frames[kMaxRender] = {};                        // device copies of the batch descriptors
    static const int kDescRing = 4;
    FrameDesc* h_frames = nullptr;                               // pinned [kMaxRender][kDescRing][kMaxBatchFrames]: descriptor uploads in flight
    hipEvent_t ev_desc[kMaxRender][kDescRing] = {};              // upload out of that pinned slot done
    int desc_gen[kMaxRender] = {};
    int launch_eye_batch(int n, const uint32_t* subframes, int r0, int r1, int rs);
    int finish_frame();
    // preprocess.hip
    Preprocessor* pre = nullptr;
    spcbpt_pretrace_path* d_pre_paths = nullptr;
    spcbpt_pretrace_node* d_pre_nodes = nullptr;
    size_t pre_capacity = 0;
    int pre_num_core = 100000, pre_padding = 10, pre_last_added = 0;
    int set_pretrace(int num_core, int padding);
    int launch_pretrace(uint32_t iteration);
    int preprocess_stage(int stage, int arg);
    int preprocess(int target_paths, int target_q_paths, bool train);
    void free_preprocess();
};

}  // namespace spc

// C ABI (include/spcbpt.h) over the HIP kernels: context, HBM-resident buffers, launches by name,
// device-side sampler build, LVC exchange, instrumentation.  The host side of the reference this
// replaces is cited per function in include/spcbpt.h.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/spcbpt.h"
#include "context.h"
#include "kernels.h"
namespace spc { void launch_repack_nodes_quad2(const float* nodes_q, float* out, int n_nodes, hipStream_t s); }   // quad_trace.hip (declared here: kernels.h is part of the megakernel's source hash)
#include "lbvh.h"
#include "env_host.h"
void spc_viewers_forget_context(spcbpt_ctx* ctx);   // viewer.cpp: called by spcbpt_destroy, so that a viewer outliving its context is safe

using namespace spc;

static thread_local std::string g_create_error;

#define HIP_TRY(ctx, expr)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e__);                   \
            return SPCBPT_ERR_HIP;                                                               \
        }                                                                                        \
    } while (0)

namespace spc {

template <class T>
static hipError_t dev_alloc(T** p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T));
}
template <class T>
static void dev_free(T*& p) {
    if (p) (void)hipFree((void*)p);
    p = nullptr;
}

void Context::time_begin(const char* name, hipStream_t s) {
    if (!timing) return;
    TimedSpan sp;
    sp.name = name;
    sp.s = s ? s : stream;
    (void)hipEventCreate(&sp.a);
    (void)hipEventCreate(&sp.b);
    (void)hipEventRecord(sp.a, sp.s);
    spans.push_back(sp);
}
void Context::time_end() {
    if (!timing || spans.empty()) return;
    (void)hipEventRecord(spans.back().b, spans.back().s);
}
int Context::sync_all() {
    HIP_TRY(this, hipStreamSynchronize(stream));
    if (lstream_b) HIP_TRY(this, hipStreamSynchronize(lstream_b));
    for (int k = 0; k < n_render; k++)
        if (rstreams[k] && rstreams[k] != stream) HIP_TRY(this, hipStreamSynchronize(rstreams[k]));
    return 0;
}
// the members d_lvc / d_vals2 / d_cmfs / d_subspace / d_sampler_counts always name the set of the light pass in progress
void Context::select_set(int s) {
    d_lvc = set_lvc[s]; d_vals2 = set_vals2[s]; d_cmfs = set_cmfs[s]; d_subspace = set_subspace[s]; d_sampler_counts = set_counts[s];
}
void Context::resolve_spans() {
    for (auto& sp : spans) {
        (void)hipEventSynchronize(sp.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            auto& acc = times[sp.name];
            acc.first += ms;
            acc.second += 1;
        }
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    spans.clear();
}

int Context::spill_entries_needed() const {
    const int entries = std::max(0, 3 * bvh_depth - kStackLds);  // a 4-wide node pushes up to 3 children
    return spill_entries_debug >= 0 ? std::min(entries, spill_entries_debug) : entries;
}
// A kernel that had to drop traversal-stack entries (deeper than LDS + spill area; cannot happen while the area is sized from
// the BVH depth) has lost subtrees: its results are wrong, and the caller is told so at the next synchronising call.
int Context::check_diag() {
    if (!d_diag) return 0;
    uint32_t h[4] = {0, 0, 0, 0};
    HIP_TRY(this, hipMemcpy(h, d_diag, sizeof(h), hipMemcpyDeviceToHost));
    if (h[0] == 0 && h[1] == 0 && h[2] == 0) return 0;
    HIP_TRY(this, hipMemsetAsync(d_diag, 0, sizeof(h), stream));   // (a null-stream memset is not ordered against the context's non-blocking streams)
    HIP_TRY(this, hipStreamSynchronize(stream));
    if (h[0] == 0 && h[1] == 0) {
        error = "light-vertex cache overflow: a light pass produced more vertices than a buffer set holds (" + std::to_string(lvc_capacity) +
                ", sized from a probe pass); frames since the last sync are invalid -- fix the capacity with spcbpt_lvc_set_capacity";
        return SPCBPT_ERR_CAPACITY;
    }
    if (h[0] == 0) {
        error = "LVC exchange: a rank's shard did not fit the agreed shard capacity (or the gathered cache did not fit the LVC); frames since the last sync are invalid -- raise the capacity (spcbpt_comm_set_shard_capacity)";
        return SPCBPT_ERR_CAPACITY;
    }
    error = "traversal stack overflow: " + std::to_string(h[0]) + " entries did not fit LDS + spill area (BVH depth " + std::to_string(bvh_depth) +
            ", spill entries per thread " + std::to_string(spill_entries_needed()) + "); results since the last sync are invalid";
    return SPCBPT_ERR_STATE;
}

int Context::ensure_spill(size_t threads, bool render) {
    const int entries = spill_entries_needed();
    kp.spill_entries = entries;
    if (entries == 0) { kp.spill = nullptr; return 0; }
    const size_t need = threads * (size_t)entries;
    uint32_t*& buf = render ? d_spill_rs[rk] : d_spill;   // one area per stream: kernels of all three may be in flight together
    size_t& cap = render ? spill_rs_capacity[rk] : spill_capacity;
    if (need > cap) {
        dev_free(buf);   // hipFree waits for the device
        HIP_TRY(this, dev_alloc(&buf, need));
        cap = need;
    }
    kp.spill = buf;
    return 0;
}

// Device layout of a classifier tree (layout.h): 16-B nodes, the eight children of a node in eight consecutive slots.  The
// caller's tree (classTree::tree_node: arbitrary child indices) is re-laid out breadth-first from the root; a child index
// that occurs twice is duplicated, so any input that classifies in finitely many steps keeps its labels.
int Context::upload_tree(const spcbpt_tree_node* t, int n, float*& d_tree, std::vector<spcbpt_tree_node>& host_copy) {
    for (int i = 0; i < n; i++) {
        if (t[i].leaf) {
            if (t[i].label < 0 || t[i].label >= SPCBPT_NUM_SUBSPACE) { error = "tree label out of range"; return SPCBPT_ERR_INVALID_ARG; }
        } else {
            if (t[i].type < 0 || t[i].type > 2) { error = "tree node type out of range"; return SPCBPT_ERR_INVALID_ARG; }
            for (int k = 0; k < 8; k++)
                if (t[i].child[k] < 0 || t[i].child[k] >= n) { error = "tree child index out of range"; return SPCBPT_ERR_INVALID_ARG; }
        }
    }
    host_copy.assign(t, t + n);
    const size_t budget = (size_t)16 * n + 64;      // a tree proper needs exactly n slots; sharing / cycles hit the budget
    std::vector<float> packed(4);
    std::vector<int> src(1, 0);                     // slot -> caller's node
    for (size_t slot = 0; slot < src.size(); slot++) {
        const spcbpt_tree_node& nd = t[src[slot]];
        uint32_t meta;
        if (nd.leaf) {
            meta = TREE_LEAF_BIT | (uint32_t)nd.label;
        } else {
            const size_t base = src.size();
            if (base + 8 > budget || base + 8 >= (1u << 29)) { error = "tree is not a finite tree (shared or cyclic children)"; return SPCBPT_ERR_INVALID_ARG; }
            meta = ((uint32_t)nd.type << 29) | (uint32_t)base;
            for (int k = 0; k < 8; k++) src.push_back(nd.child[k]);
            packed.resize(src.size() * 4);
        }
        float* q = &packed[slot * 4];
        q[0] = nd.mid[0]; q[1] = nd.mid[1]; q[2] = nd.mid[2];
        memcpy(q + 3, &meta, 4);
    }
    dev_free(d_tree);
    HIP_TRY(this, dev_alloc(&d_tree, packed.size()));
    HIP_TRY(this, hipMemcpyAsync(d_tree, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipStreamSynchronize(stream));
    return 0;
}

int Context::install_subspace(const spcbpt_tree_node* et, int ne, const spcbpt_tree_node* lt, int nl, const float* q, const float* g) {
    if (!et || !lt || !q || !g || ne < 1 || nl < 1) { error = "set_subspace: all four of eye_tree, light_tree, q, cmf_gamma are required"; return SPCBPT_ERR_INVALID_ARG; }
    if (sync_all()) return SPCBPT_ERR_HIP;  // a render launch may still be reading the tuple that is replaced in place
    int rc = upload_tree(et, ne, d_eye_tree, h_eye_tree);
    if (rc) return rc;
    rc = upload_tree(lt, nl, d_light_tree, h_light_tree);
    if (rc) return rc;
    tree_has_direction = false;
    for (int i = 0; i < ne; i++) if (!et[i].leaf && et[i].type == 2) tree_has_direction = true;
    for (int i = 0; i < nl; i++) if (!lt[i].leaf && lt[i].type == 2) tree_has_direction = true;
    h_Q.assign(q, q + SPCBPT_NUM_SUBSPACE);
    h_gamma.assign(g, g + (size_t)SPCBPT_NUM_SUBSPACE * SPCBPT_NUM_SUBSPACE);
    if (!d_Q) HIP_TRY(this, dev_alloc(&d_Q, SPCBPT_NUM_SUBSPACE));
    if (!d_gamma) HIP_TRY(this, dev_alloc(&d_gamma, (size_t)SPCBPT_NUM_SUBSPACE * SPCBPT_NUM_SUBSPACE));
    HIP_TRY(this, hipMemcpyAsync(d_Q, h_Q.data(), h_Q.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemcpyAsync(d_gamma, h_gamma.data(), h_gamma.size() * 4, hipMemcpyHostToDevice, stream));
    {   // three-level copy for first-stage sampling (device_lib.h: sample_first_stage3; layout.h: CMF2_*)
        std::vector<float> two((size_t)SPCBPT_NUM_SUBSPACE * CMF2_ROW, 2.0f);
        for (int e = 0; e < SPCBPT_NUM_SUBSPACE; e++) {
            float* row = &two[(size_t)e * CMF2_ROW];
            float* fine = row + CMF2_COARSE + CMF2_MID;
            memcpy(fine, &h_gamma[(size_t)e * SPCBPT_NUM_SUBSPACE], SPCBPT_NUM_SUBSPACE * sizeof(float));
            for (int m = 0; m < CMF2_MID; m++) row[CMF2_COARSE + m] = fine[8 * m + 7];
            for (int k = 0; k < CMF2_COARSE; k++) row[k] = fine[64 * k + 63];
        }
        if (!d_gamma2) HIP_TRY(this, dev_alloc(&d_gamma2, two.size()));
        HIP_TRY(this, hipMemcpy(d_gamma2, two.data(), two.size() * sizeof(float), hipMemcpyHostToDevice));
        {   // gamma_ss as a table (layout.h: KParams::gamma_q): the device's own FP32 subtraction and division, done once here
            std::vector<float> gq((size_t)SPCBPT_NUM_SUBSPACE * SPCBPT_NUM_SUBSPACE);
            for (int e = 0; e < SPCBPT_NUM_SUBSPACE; e++) {
                const float* row = &h_gamma[(size_t)e * SPCBPT_NUM_SUBSPACE];
                for (int l = 0; l < SPCBPT_NUM_SUBSPACE; l++) {
                    const float g = l == 0 ? row[0] : row[l] - row[l - 1];
                    gq[(size_t)e * SPCBPT_NUM_SUBSPACE + l] = g / h_Q[l];
                }
            }
            if (!d_gamma_q) HIP_TRY(this, dev_alloc(&d_gamma_q, gq.size()));
            HIP_TRY(this, hipMemcpy(d_gamma_q, gq.data(), gq.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        {   // first-stage guide table (layout.h: KParams::cmf_guide1): per row, the first entry above b / CMF_GUIDE1 for every bucket b
            std::vector<uint16_t> guide((size_t)SPCBPT_NUM_SUBSPACE * CMF_GUIDE1);
            for (int e = 0; e < SPCBPT_NUM_SUBSPACE; e++) {
                const float* row = &h_gamma[(size_t)e * SPCBPT_NUM_SUBSPACE];
                int k = 0;
                for (int b = 0; b < CMF_GUIDE1; b++) {
                    const float t = (float)b / (float)CMF_GUIDE1;   // exact; u * CMF_GUIDE1 is exact too, so every u of bucket b is >= t
                    while (k < SPCBPT_NUM_SUBSPACE && !(row[k] > t)) k++;
                    guide[(size_t)e * CMF_GUIDE1 + b] = (uint16_t)k;
                }
            }
            if (!d_guide1) HIP_TRY(this, dev_alloc(&d_guide1, guide.size()));
            HIP_TRY(this, hipMemcpy(d_guide1, guide.data(), guide.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        gamma_monotone = true;   // counting equals bisecting only on a non-decreasing row that ends above every random number
        for (int e = 0; e < SPCBPT_NUM_SUBSPACE && gamma_monotone; e++) {
            const float* row = &h_gamma[(size_t)e * SPCBPT_NUM_SUBSPACE];
            for (int l = 1; l < SPCBPT_NUM_SUBSPACE; l++) if (!(row[l] >= row[l - 1])) { gamma_monotone = false; break; }
            if (!(row[SPCBPT_NUM_SUBSPACE - 1] >= 1.0f)) gamma_monotone = false;
        }
    }
    HIP_TRY(this, hipStreamSynchronize(stream));
    kp.eye_tree = d_eye_tree; kp.light_tree = d_light_tree; kp.Q = d_Q; kp.cmf_gamma = d_gamma; kp.cmf_gamma2 = gamma_monotone ? d_gamma2 : nullptr; kp.cmf_guide1 = d_guide1; kp.gamma_q = d_gamma_q;
    have_subspace = true;
    return 0;
}

// The environment map as one more light: env_params_setup (optixPathTracer.cpp:431-461) + the ENV entry and the patch-subspace
// shift of LightSource_shift (scene_shift.cpp:108-153).
int Context::set_environment(const float* rgba, int w, int h, const float* center, float radius) {
    if (!rgba || w < 1 || h < 1 || (long long)w * h > (1ll << 26)) { error = "set_environment: bad image"; return SPCBPT_ERR_INVALID_ARG; }
    if (kp.scene.env.valid) { error = "set_environment: the context already has an environment map"; return SPCBPT_ERR_STATE; }
    int patches = 0;
    for (const DLight& L : h_lights) patches += L.div_level * L.div_level;
    if (patches > SPCBPT_NUM_SUBSPACE_LIGHTSOURCE / 2) { error = "set_environment: with an environment map the quad lights may use at most 100 patch subspaces (sum of div_level^2)"; return SPCBPT_ERR_INVALID_ARG; }
    for (size_t i = 0; i < (size_t)w * h * 4; i++) if (!std::isfinite(rgba[i])) { error = "set_environment: non-finite texel"; return SPCBPT_ERR_INVALID_ARG; }
    if (sync_all()) return SPCBPT_ERR_HIP;
    std::vector<float> tex, cmf;
    env_build(rgba, w, h, tex, cmf);
    if (!(cmf.back() > 0.0f) || !std::isfinite(cmf.back())) { error = "set_environment: the image holds no energy"; return SPCBPT_ERR_INVALID_ARG; }
    dev_free(d_env_tex); dev_free(d_env_cmf);
    HIP_TRY(this, dev_alloc(&d_env_tex, tex.size()));
    HIP_TRY(this, dev_alloc(&d_env_cmf, cmf.size()));
    HIP_TRY(this, hipMemcpy(d_env_tex, tex.data(), tex.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(this, hipMemcpy(d_env_cmf, cmf.data(), cmf.size() * 4, hipMemcpyHostToDevice));
    // scene_shift.cpp:110: the quad lights' patches start at 0.5 * NUM_SUBSPACE_LIGHTSOURCE, the sky's divLevel^2 directions at 0
    for (DLight& L : h_lights) L.ss_base += SPCBPT_NUM_SUBSPACE_LIGHTSOURCE / 2;
    DLight E;
    memset(&E, 0, sizeof(E));
    E.type = 1; E.id = (int)h_lights.size();   // (Light() leaves id / divLevel / ssBase indeterminate upstream)
    h_lights.push_back(E);
    dev_free(d_lights);
    HIP_TRY(this, dev_alloc(&d_lights, h_lights.size()));
    HIP_TRY(this, hipMemcpy(d_lights, h_lights.data(), h_lights.size() * sizeof(DLight), hipMemcpyHostToDevice));
    n_lights = (int)h_lights.size();
    DEnv& V = kp.scene.env;
    V.tex = d_env_tex; V.cmf = d_env_cmf;
    V.width = w; V.height = h; V.size = w * h;
    V.div_level = (int)sqrt(0.5 * SPCBPT_NUM_SUBSPACE_LIGHTSOURCE);
    if (center && radius > 0.0f) { memcpy(V.center, center, 12); V.r = radius; }
    else {   // the scene's bounding box: centre and diagonal (sky.center / sky.r of env_params_setup, over the TRUE box: SURVEY q7)
        double d2 = 0.0;
        for (int k = 0; k < 3; k++) { V.center[k] = 0.5f * (bbox_lo[k] + bbox_hi[k]); const double e = (double)bbox_lo[k] - (double)bbox_hi[k]; d2 += e * e; }
        V.r = (float)sqrt(d2);
    }
    V.project_pdf = (float)(1 / (3.14159265358979323846 * V.r * V.r));
    V.valid = 1;
    kp.scene.general = 1;
    blocks_per_cu[0] = blocks_per_cu_batch = 0;   // other instantiations from now on: ask again
    kp.scene.lights = d_lights; kp.scene.n_lights = n_lights;
    // every cache traced so far is without sky vertices
    have_sampler = false; pending.clear(); built_sets.clear(); lvc_count = 0;
    lvc_probe_needed = lvc_fixed == 0;
    return 0;
}

int Context::set_light_trace(const spcbpt_light_trace_params& p) {
    if (p.num_core < 1 || p.core_padding < 1 || p.m_per_core < 1) { error = "set_light_trace: sizes must be positive"; return SPCBPT_ERR_INVALID_ARG; }
    int begin = p.core_begin, count = p.core_count == 0 ? p.num_core - p.core_begin : p.core_count;
    if (begin < 0 || count < 1 || begin + count > p.num_core) { error = "set_light_trace: core range out of bounds"; return SPCBPT_ERR_INVALID_ARG; }
    lt = p;
    lt.core_count = count;
    const size_t slots = (size_t)count * p.core_padding;
    if (slots > scratch_capacity) {
        dev_free(d_scratch);
        HIP_TRY(this, dev_alloc(&d_scratch, slots));
        scratch_capacity = slots;
    }
    if ((size_t)count + 1 > counts_capacity) {
        dev_free(d_core_counts); dev_free(d_core_offsets);
        HIP_TRY(this, dev_alloc(&d_core_counts, (size_t)count + 1));
        HIP_TRY(this, dev_alloc(&d_core_offsets, (size_t)count + 1));
        counts_capacity = (size_t)count + 1;
    }
    // the compact LVC holds the whole job's cache (every rank's shard after an all-gather): sized by hand, or from a probe pass
    // at the next light pass (context.h: lvc_capacity)
    if (lvc_fixed) return ensure_lvc_capacity(lvc_fixed);
    lvc_probe_needed = true;
    return 0;
}

// Sizes the buffer sets from one pass of this context's cores, traced into the padded scratch and counted on the host.
int Context::probe_lvc_capacity() {
    lvc_probe_needed = false;
    const size_t worst = (size_t)lt.num_core * lt.core_padding;
    if (sync_all()) return SPCBPT_ERR_HIP;
    kp.num_core = lt.num_core; kp.core_padding = lt.core_padding; kp.m_per_core = lt.m_per_core;
    kp.core_begin = lt.core_begin; kp.core_count = lt.core_count; kp.launch_frame = 0x7f000001u;
    kp.n_lframes = 0;
    kp.lt_decorrelate = lt.decorrelate_bsdf_stream;
    kp.lvc_scratch = d_scratch; kp.core_counts = d_core_counts;
    {
        const int entries = spill_entries_needed();
        kp.spill_entries = entries;
        const size_t need = (((size_t)lt.core_count + 255) / 256 * 256) * (size_t)entries;
        if (entries == 0) kp.spill = nullptr;
        else {
            if (need > spill_capacity) { dev_free(d_spill); HIP_TRY(this, dev_alloc(&d_spill, need)); spill_capacity = need; }
            kp.spill = d_spill;
        }
    }
    kp.counters = nullptr;
    HIP_TRY(this, hipMemsetAsync(d_core_counts, 0, ((size_t)lt.core_count + 1) * sizeof(int), stream));
    kp.path_counter = d_set_counts_all + 2 * kMaxSets;   // a spare word behind the sets' counts
    kp.work_counter = d_work_counter + kMaxRender;
    HIP_TRY(this, hipMemsetAsync(kp.work_counter, 0, sizeof(uint32_t), stream));
    if (light_blocks < 0) { const char* lb = getenv("SPCBPT_LIGHT_BLOCKS"); light_blocks = lb ? std::max(1, atoi(lb)) : std::max(1, num_cus); }
    launch_light_trace(kp, tree_has_direction ? 1 : 0, light_blocks, stream);
    HIP_TRY(this, hipGetLastError());
    std::vector<int> h((size_t)lt.core_count);
    HIP_TRY(this, hipMemcpyAsync(h.data(), d_core_counts, h.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(this, hipStreamSynchronize(stream));
    double total = 0.0;
    for (int v : h) total += (double)v;
    total *= (double)lt.num_core / (double)std::max(1, lt.core_count);   // a rank's share of a sharded job -> the gathered cache
    size_t cap = (size_t)std::max(2.0 * total, total + 65536.0);
    cap = (cap + 4095) / 4096 * 4096;
    cap = std::max<size_t>(1, std::min(cap, worst));
    return ensure_lvc_capacity(cap);
}

int Context::ensure_lvc_capacity(size_t n) {
    if (n <= lvc_capacity) return 0;
    if (sync_all()) return SPCBPT_ERR_HIP;
    // Footprint (round 6, advisor): per buffer set TWO copies of the cache (own order + the sampler's order: 2 x 96 B per vertex),
    // jump 4 B, CMF 4 B, guide 4 B = 204 B per vertex and set; n_sets = eye_batch * (n_render + 2) + 3 (83 for 20-frame batches).  A
    // calibrated cache (spcbpt_lvc_calibrate: ~2 x a measured pass, 0.5 M vertices on the bench scene) is 100 MB per set; the
    // uncalibrated worst case core_count x padding (5.2 M) is 1.06 GB per set -- INTEGRATION.md section 4 tells hosts to calibrate.
    // The old buffers are gone from here on; if an allocation below fails the context is left EMPTY and consistent (capacity 0, every
    // pointer null, no sampler): the failed call returns SPCBPT_ERR_HIP, and a later call with a size the device can hold succeeds.
    dev_free(d_keys); dev_free(d_keys2); dev_free(d_vals); dev_free(d_weights);
    dev_free(d_wsorted); dev_free(d_prefix);
    for (int s = 0; s < n_sets; s++) { dev_free(set_lvc[s]); dev_free(set_lvc_sorted[s]); dev_free(set_vals2[s]); dev_free(set_cmfs[s]); dev_free(set_guide[s]); }
    for (int s2 = 0; s2 < kMaxSets; s2++) { set_count_host[s2] = -1; light_counts_valid[s2] = false; set_bound[s2] = -1; ev_exch_set[s2] = false; }   // the sets are empty again
    pending.clear();
    built_sets.clear();   // samplers built in the old allocations went with them
    free_batch_build_scratch();
    sbb_refused_bytes = 0;   // (a new capacity is a new question to the allocator)
    lvc_capacity = 0;
    lvc_count = 0;
    have_sampler = false;
    hipError_t e = hipSuccess;
    for (int s = 0; s < n_sets && e == hipSuccess; s++) {   // what the eye pass reads exists once per frame in flight (see context.h)
        e = dev_alloc(&set_lvc[s], n);
        if (e == hipSuccess) e = dev_alloc(&set_lvc_sorted[s], n);
        if (e == hipSuccess) e = dev_alloc(&set_vals2[s], n);
        if (e == hipSuccess) e = dev_alloc(&set_cmfs[s], n + 8);   // (the eye kernel reads a CMF in aligned windows of eight: kernels.hip guide_window)
        if (e == hipSuccess) e = dev_alloc(&set_guide[s], n);
    }
    if (e == hipSuccess) e = dev_alloc(&d_keys, n);
    if (e == hipSuccess) e = dev_alloc(&d_keys2, n);
    if (e == hipSuccess) e = dev_alloc(&d_vals, n);
    if (e == hipSuccess) e = dev_alloc(&d_weights, n);
    if (e == hipSuccess) e = dev_alloc(&d_wsorted, n);
    if (e == hipSuccess) e = dev_alloc(&d_prefix, n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dev_free(d_keys); dev_free(d_keys2); dev_free(d_vals); dev_free(d_weights); dev_free(d_wsorted); dev_free(d_prefix);
        for (int s = 0; s < n_sets; s++) { dev_free(set_lvc[s]); dev_free(set_lvc_sorted[s]); dev_free(set_vals2[s]); dev_free(set_cmfs[s]); dev_free(set_guide[s]); }
        select_set(lset);
        error = std::string("light-vertex cache of ") + std::to_string(n) + " vertices x " + std::to_string(n_sets) + " buffer sets: " + hipGetErrorString(e) +
                " (the context now holds NO cache: call spcbpt_lvc_set_capacity / spcbpt_lvc_calibrate with a size the device can hold)";
        return SPCBPT_ERR_HIP;
    }
    select_set(lset);
    lvc_capacity = n;
    return 0;
}

int Context::ensure_temp(size_t bytes) {
    if (bytes <= temp_capacity) return 0;
    dev_free(d_temp);
    HIP_TRY(this, dev_alloc(&d_temp, bytes));
    temp_capacity = bytes;
    return 0;
}

int Context::ensure_lane_b() {
    if (!lstream_b) {
        int least = 0, greatest = 0;
        HIP_TRY(this, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(this, hipStreamCreateWithPriority(&lstream_b, hipStreamNonBlocking, greatest));
    }
    const size_t slots = (size_t)lt.core_count * lt.core_padding;
    if (slots > b_scratch_capacity) { dev_free(b_scratch); HIP_TRY(this, dev_alloc(&b_scratch, slots)); b_scratch_capacity = slots; }
    if ((size_t)lt.core_count + 1 > b_counts_capacity) {
        dev_free(b_core_counts); dev_free(b_core_offsets);
        HIP_TRY(this, dev_alloc(&b_core_counts, (size_t)lt.core_count + 1));
        HIP_TRY(this, dev_alloc(&b_core_offsets, (size_t)lt.core_count + 1));
        b_counts_capacity = (size_t)lt.core_count + 1;
    }
    if (lvc_capacity > b_keys_capacity) {
        dev_free(b_keys); dev_free(b_vals); dev_free(b_weights);
        HIP_TRY(this, dev_alloc(&b_keys, lvc_capacity)); HIP_TRY(this, dev_alloc(&b_vals, lvc_capacity)); HIP_TRY(this, dev_alloc(&b_weights, lvc_capacity));
        b_keys_capacity = lvc_capacity;
    }
    return 0;
}

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

// The minimal VALID subspace tuple: single-leaf trees, Q from a few light passes, Gamma rows ~ Q.
int Context::install_minimal_tuple() {
    spcbpt_tree_node leaf;
    memset(&leaf, 0, sizeof(leaf));
    leaf.leaf = 1; leaf.label = 0;
    std::vector<float> q(SPCBPT_NUM_SUBSPACE, 1.0f), g((size_t)SPCBPT_NUM_SUBSPACE * SPCBPT_NUM_SUBSPACE);
    for (int e = 0; e < SPCBPT_NUM_SUBSPACE; e++)
        for (int l = 0; l < SPCBPT_NUM_SUBSPACE; l++) g[(size_t)e * SPCBPT_NUM_SUBSPACE + l] = (float)(l + 1) / SPCBPT_NUM_SUBSPACE;
    int rc = install_subspace(&leaf, 1, &leaf, 1, q.data(), g.data());
    if (rc) return rc;
    std::vector<double> acc(SPCBPT_NUM_SUBSPACE, 0.0);
    long long paths = 0;
    std::vector<LightVertex> host;
    for (int f = 0; f < 4; f++) {
        rc = launch_light(10000u + f);
        if (rc) return rc;
        rc = fetch_counts();
        if (rc) return rc;
        host.resize(lvc_count);
        HIP_TRY(this, hipMemcpy(host.data(), d_lvc, (size_t)lvc_count * sizeof(LightVertex), hipMemcpyDeviceToHost));
        for (const auto& v : host) {
            float w = (v.flux[0] + v.flux[1] + v.flux[2]) / v.pdf;
            if (std::isnan(w) || std::isinf(w)) w = 0;
            acc[v.subspace_id] += w;
            if (v.depth == 0) paths++;
        }
    }
    double total = 0;
    for (int s = 0; s < SPCBPT_NUM_SUBSPACE; s++) { acc[s] /= (double)std::max(1LL, paths); total += acc[s]; }
    if (!(total > 0)) { error = "minimal tuple: the light pass produced no weight (no emitters?)"; return SPCBPT_ERR_STATE; }
    double run = 0;
    std::vector<float> row(SPCBPT_NUM_SUBSPACE);
    for (int s = 0; s < SPCBPT_NUM_SUBSPACE; s++) { run += acc[s] / total; row[s] = (float)run; q[s] = acc[s] == 0 ? FLT_MAX : (float)acc[s]; }
    row[SPCBPT_NUM_SUBSPACE - 1] = 1.0f;
    for (int s = 1; s < SPCBPT_NUM_SUBSPACE; s++) row[s] = std::max(row[s], row[s - 1]);
    for (int e = 0; e < SPCBPT_NUM_SUBSPACE; e++) memcpy(&g[(size_t)e * SPCBPT_NUM_SUBSPACE], row.data(), SPCBPT_NUM_SUBSPACE * sizeof(float));
    return install_subspace(&leaf, 1, &leaf, 1, q.data(), g.data());
}

Context::~Context() {
    resolve_spans();
    free_preprocess();
    dev_free(d_nodes); dev_free(d_nodes_q); dev_free(d_nodes_q2); /* d_tris lives in d_nodes' allocation */ dev_free(d_tri_orig); dev_free(d_mats); dev_free(d_lights); dev_free(d_tex);
    for (auto p : d_tex_data) (void)hipFree(p);
    dev_free(d_env_tex); dev_free(d_env_cmf); dev_free(d_accum); dev_free(d_frame); dev_free(d_eye_tree); dev_free(d_light_tree); dev_free(d_Q); dev_free(d_gamma); dev_free(d_gamma2); dev_free(d_guide1); dev_free(d_gamma_q);
    dev_free(d_scratch); dev_free(d_core_counts); dev_free(d_core_offsets); dev_free(d_keys); dev_free(d_keys2);
    dev_free(d_vals); dev_free(d_weights); dev_free(d_wsorted); dev_free(d_prefix);
    for (int s = 0; s < kMaxSets; s++) { dev_free(set_lvc[s]); dev_free(set_lvc_sorted[s]); dev_free(set_vals2[s]); dev_free(set_cmfs[s]); dev_free(set_guide[s]); dev_free(set_subspace[s]); }
    dev_free(d_set_counts_all); dev_free(lb_scratch); dev_free(lb_core_counts); dev_free(lb_core_offsets); dev_free(lb_path_counts); dev_free(lb_spill);
    dev_free(d_counters); dev_free(d_diag); dev_free(d_work_counter); if (h_import_counts) (void)hipHostFree(h_import_counts); if (h_light_counts) (void)hipHostFree(h_light_counts);
    for (int s2 = 0; s2 < kMaxRender; s2++) { for (int k = 0; k < kMaxBatchFrames; k++) dev_free(d_result_b[s2][k]); if (d_frames[s2]) (void)hipFree(d_frames[s2]); }
    if (h_frames) (void)hipHostFree(h_frames);
    for (int g2 = 0; g2 < 2; g2++) if (ev_import[g2]) (void)hipEventDestroy(ev_import[g2]);
    for (int s2 = 0; s2 < kMaxRender; s2++) for (int g2 = 0; g2 < kDescRing; g2++) if (ev_desc[s2][g2]) (void)hipEventDestroy(ev_desc[s2][g2]);
    dev_free(b_scratch); dev_free(b_core_counts); dev_free(b_core_offsets); dev_free(b_keys); dev_free(b_vals); dev_free(b_weights); dev_free(b_temp); dev_free(b_spill);
    if (lstream_b) (void)hipStreamDestroy(lstream_b); dev_free(d_spill); dev_free(d_temp); dev_free(d_hist);
    dev_free(sbb_keys); dev_free(sbb_weights); dev_free(sbb_wsorted); dev_free(sbb_hist);
    for (int s = 0; s < kMaxRender; s++) {
        if (rstreams[s] && rstreams[s] != stream) (void)hipStreamDestroy(rstreams[s]);
        if (ev_merge[s]) (void)hipEventDestroy(ev_merge[s]);
        dev_free(d_result[s]); dev_free(d_spill_rs[s]);
    }
    if (cstream) (void)hipStreamDestroy(cstream);
    if (stream) (void)hipStreamDestroy(stream);
    for (int s = 0; s < kMaxSets; s++) {
        if (ev_sampler[s]) (void)hipEventDestroy(ev_sampler[s]);
        if (ev_render[s]) (void)hipEventDestroy(ev_render[s]);
        if (ev_light[s]) (void)hipEventDestroy(ev_light[s]);
        if (ev_set_stream[s]) (void)hipEventDestroy(ev_set_stream[s]);
        if (ev_exch[s]) (void)hipEventDestroy(ev_exch[s]);
    }
}

}  // namespace spc

struct spcbpt_ctx : public spc::Context {};

extern "C" {

int spcbpt_create(const spcbpt_scene_desc* sc, int device, spcbpt_ctx** out) {
    if (!sc || !out) { g_create_error = "null argument"; return SPCBPT_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!sc->vertices || !sc->indices || !sc->tri_material || sc->n_vertices < 3 || sc->n_triangles < 1 || sc->n_materials < 1 ||
        !sc->materials || sc->n_lights < 1 || !sc->lights) {
        g_create_error = "scene needs vertices, indices, tri_material, >=1 material and >=1 quad light";
        return SPCBPT_ERR_INVALID_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device available (the MI355X path has no CPU fallback)"; return SPCBPT_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { g_create_error = "device ordinal out of range"; return SPCBPT_ERR_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return SPCBPT_ERR_NO_DEVICE; }
    // validate indices before anything reaches a kernel
    for (int t = 0; t < sc->n_triangles; t++) {
        for (int k = 0; k < 3; k++)
            if (sc->indices[3 * (size_t)t + k] >= (uint32_t)sc->n_vertices) { g_create_error = "vertex index out of range"; return SPCBPT_ERR_INVALID_ARG; }
        if (sc->tri_material[t] < 0 || sc->tri_material[t] >= sc->n_materials) { g_create_error = "material index out of range"; return SPCBPT_ERR_INVALID_ARG; }
    }
    int patches = 0;
    for (int i = 0; i < sc->n_lights; i++) {
        if (sc->lights[i].div_level < 1) { g_create_error = "light div_level must be >= 1"; return SPCBPT_ERR_INVALID_ARG; }
        patches += sc->lights[i].div_level * sc->lights[i].div_level;
    }
    if (patches > SPCBPT_NUM_SUBSPACE_LIGHTSOURCE) { g_create_error = "sum of div_level^2 exceeds NUM_SUBSPACE_LIGHTSOURCE (200)"; return SPCBPT_ERR_INVALID_ARG; }
    if (sc->n_materials + sc->n_lights > 32767) { g_create_error = "too many materials (int16 material ids)"; return SPCBPT_ERR_INVALID_ARG; }

    spcbpt_ctx* c = new spcbpt_ctx();
    c->device = device;
#define CREATE_TRY(expr)                                                                                           \
    do {                                                                                                           \
        hipError_t e__ = (expr);                                                                                   \
        if (e__ != hipSuccess) { g_create_error = std::string(#expr) + ": " + hipGetErrorString(e__); delete c; return SPCBPT_ERR_HIP; } \
    } while (0)
    // non-blocking streams: a host that drives collectives on the legacy default stream (torch) must not be serialised with
    // the render stream; every hand-over in this file is an explicit event or synchronize
    // Priorities: the light pass, the sampler build and the exchange copies are small, latency-critical kernels that the host
    // (or the next eye launch) waits for, while several persistent eye kernels queue for every block slot that frees up; the
    // light stream therefore gets the highest priority and the render streams the lowest (SPCBPT_STREAM_PRIORITY=0: all equal).
    int prio_least = 0, prio_greatest = 0;
    CREATE_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const char* sp = getenv("SPCBPT_STREAM_PRIORITY");
    const bool use_prio = !(sp && std::string(sp) == "0") && prio_least != prio_greatest;
    if (use_prio) CREATE_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest));
    else CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        const char* ov = getenv("SPCBPT_OVERLAP");
        const char* nr = getenv("SPCBPT_RENDER_STREAMS");
        c->n_render = nr ? std::max(1, std::min((int)Context::kMaxRender, atoi(nr))) : 2;
        if (ov && std::string(ov) == "0") c->n_render = 1;
        if (const char* eb = getenv("SPCBPT_EYE_BATCH")) c->eye_batch = std::max(1, std::min((int)kMaxBatchFrames, atoi(eb)));
        // sets: one per frame of every eye launch in flight + the light passes ahead of them + the one being built
        c->n_sets = std::min((int)Context::kMaxSets, c->eye_batch > 1 ? c->eye_batch * (c->n_render + 2) + 3 : c->n_render + 4);   // batches in flight + one being built + a batch of light passes ahead
        if (const char* sb = getenv("SPCBPT_SAMPLER_BUILD")) c->counting_build = std::string(sb) != "hipcub";
        if (const char* lc = getenv("SPCBPT_LVC_CAPACITY")) c->lvc_fixed = (size_t)std::max(0ll, atoll(lc));   // vertices per buffer set (0 = probe)
        if (const char* ns = getenv("SPCBPT_SETS")) c->n_sets = std::max(3, std::min((int)Context::kMaxSets, atoi(ns)));   // developer knob
        for (int s = 0; s < c->n_render; s++) {
            if (ov && std::string(ov) == "0") c->rstreams[s] = c->stream;
            else if (use_prio) CREATE_TRY(hipStreamCreateWithPriority(&c->rstreams[s], hipStreamNonBlocking, prio_least));
            else CREATE_TRY(hipStreamCreateWithFlags(&c->rstreams[s], hipStreamNonBlocking));
            CREATE_TRY(hipEventCreateWithFlags(&c->ev_merge[s], hipEventDisableTiming));
        }
        c->rstream = c->rstreams[0];
        for (int s = 0; s < c->n_sets; s++) {
            CREATE_TRY(hipEventCreateWithFlags(&c->ev_sampler[s], hipEventDisableTiming));
            CREATE_TRY(hipEventCreateWithFlags(&c->ev_render[s], hipEventDisableTiming));
            CREATE_TRY(hipEventCreateWithFlags(&c->ev_light[s], hipEventDisableTiming));
            CREATE_TRY(hipEventCreateWithFlags(&c->ev_set_stream[s], hipEventDisableTiming));
            CREATE_TRY(hipEventCreateWithFlags(&c->ev_exch[s], hipEventDisableTiming));
            c->set_bound[s] = -1;
            c->set_count_host[s] = -1;
        }
        CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_import_counts), (size_t)Context::kMaxSets * 2 * sizeof(int)));
        CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_light_counts), (size_t)Context::kMaxSets * 2 * sizeof(int)));
    }

    // ---- scene assembly (scene_shift.cpp:64-154, 184-328)
    std::vector<float> V(sc->vertices, sc->vertices + 3 * (size_t)sc->n_vertices);
    std::vector<float> UV(2 * (size_t)sc->n_vertices, 0.0f);
    if (sc->texcoords) UV.assign(sc->texcoords, sc->texcoords + 2 * (size_t)sc->n_vertices);
    std::vector<uint32_t> I(sc->indices, sc->indices + 3 * (size_t)sc->n_triangles);
    std::vector<int32_t> TM(sc->tri_material, sc->tri_material + sc->n_triangles);
    std::vector<uint8_t> EM(sc->n_triangles, 0);
    std::vector<DMaterial> mats;
    for (int i = 0; i < sc->n_materials; i++) {
        const spcbpt_material& m = sc->materials[i];
        if (m.albedo_tex < 0 || m.albedo_tex > sc->n_textures) { g_create_error = "albedo_tex out of range"; delete c; return SPCBPT_ERR_INVALID_ARG; }
        DMaterial d;
        memset(&d, 0, sizeof(d));
        memcpy(d.base_color, m.base_color, 12);
        d.metallic = m.metallic; d.roughness = m.roughness; d.specular = m.specular; d.specular_tint = m.specular_tint;
        d.subsurface = m.subsurface; d.sheen = m.sheen; d.sheen_tint = m.sheen_tint; d.clearcoat = m.clearcoat;
        d.clearcoat_gloss = m.clearcoat_gloss; d.albedo_tex = m.albedo_tex; d.light_id = -1;
        d.brdf = m.brdf != 0;   // Pbr::brdf is a bool (MaterialData.h:99) assigned from the scene file's int (scene_shift.cpp:75)
        mats.push_back(d);
    }
    std::vector<DLight> lights;
    int ss_base = 0;
    for (int i = 0; i < sc->n_lights; i++) {
        const spcbpt_quad_light& s = sc->lights[i];
        DLight L;
        memset(&L, 0, sizeof(L));
        float cr[3] = {s.u[1] * s.v[2] - s.u[2] * s.v[1], s.u[2] * s.v[0] - s.u[0] * s.v[2], s.u[0] * s.v[1] - s.u[1] * s.v[0]};
        float len = sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
        if (!(len > 0)) { g_create_error = "degenerate quad light"; delete c; return SPCBPT_ERR_INVALID_ARG; }
        float inv = 1.0f / len;
        for (int k = 0; k < 3; k++) {
            L.corner[k] = s.position[k]; L.u[k] = s.position[k] + s.u[k]; L.v[k] = s.position[k] + s.v[k];
            L.emission[k] = s.emission[k]; L.normal[k] = cr[k] * inv;
        }
        L.area = len; L.div_level = s.div_level; L.ss_base = ss_base; L.id = (int)lights.size();
        ss_base += s.div_level * s.div_level;
        lights.push_back(L);
        DMaterial d;  // emissive pseudo-material with MaterialData() defaults
        memset(&d, 0, sizeof(d));
        d.base_color[0] = d.base_color[1] = d.base_color[2] = 1.0f; d.metallic = 1.0f; d.roughness = 1.0f; d.specular = 0.5f;
        d.sheen_tint = 0.5f; d.clearcoat_gloss = 1.0f; d.light_id = i;
        mats.push_back(d);
        uint32_t base = (uint32_t)(V.size() / 3);
        float p3[3] = {L.u[0] + L.v[0] - L.corner[0], L.u[1] + L.v[1] - L.corner[1], L.u[2] + L.v[2] - L.corner[2]};
        V.insert(V.end(), L.corner, L.corner + 3); V.insert(V.end(), L.u, L.u + 3); V.insert(V.end(), L.v, L.v + 3); V.insert(V.end(), p3, p3 + 3);
        const float quv[8] = {0, 0, 1, 0, 0, 1, 1, 1};
        UV.insert(UV.end(), quv, quv + 8);
        const uint32_t qi[6] = {base, base + 1, base + 3, base, base + 3, base + 2};
        I.insert(I.end(), qi, qi + 6);
        TM.push_back((int)mats.size() - 1); TM.push_back((int)mats.size() - 1);
        EM.push_back(1); EM.push_back(1);
    }
    HostMesh mesh;
    mesh.vertices = V.data(); mesh.texcoords = UV.data(); mesh.indices = I.data(); mesh.tri_material = TM.data(); mesh.tri_emitter = EM.data();
    mesh.n_vertices = (int)(V.size() / 3); mesh.n_triangles = (int)(I.size() / 3);
    Lbvh bvh;
    build_lbvh(mesh, bvh);
    c->n_triangles = mesh.n_triangles; c->n_nodes = (int)(bvh.nodes.size() / 16); c->bvh_depth = bvh.depth;
    c->n_lights = (int)lights.size(); c->n_mats = (int)mats.size();

    // nodes and triangles in ONE allocation, the triangle records right behind the node records (both 64 B): the pooled traversal
    // step fetches "the record of its next step" through one base pointer (device_lib.h: SPC_FETCH_STEP__)
    CREATE_TRY(dev_alloc(&c->d_nodes, bvh.nodes.size() + bvh.tris.size()));
    c->d_tris = c->d_nodes + bvh.nodes.size();
    CREATE_TRY(dev_alloc(&c->d_tri_orig, bvh.tri_orig.size()));
    CREATE_TRY(dev_alloc(&c->d_mats, mats.size()));
    CREATE_TRY(dev_alloc(&c->d_lights, lights.size()));
    CREATE_TRY(hipMemcpy(c->d_nodes, bvh.nodes.data(), bvh.nodes.size() * 4, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(c->d_tris, bvh.tris.data(), bvh.tris.size() * 4, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(c->d_tri_orig, bvh.tri_orig.data(), bvh.tri_orig.size() * 4, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(c->d_mats, mats.data(), mats.size() * sizeof(DMaterial), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(c->d_lights, lights.data(), lights.size() * sizeof(DLight), hipMemcpyHostToDevice));
    c->h_lights = lights;
    for (int k = 0; k < 3; k++) { c->bbox_lo[k] = 1e30f; c->bbox_hi[k] = -1e30f; }
    for (size_t i = 0; i < V.size(); i += 3)
        for (int k = 0; k < 3; k++) { c->bbox_lo[k] = std::min(c->bbox_lo[k], V[i + k]); c->bbox_hi[k] = std::max(c->bbox_hi[k], V[i + k]); }
    std::vector<DTexture> texs;
    for (int i = 0; i < sc->n_textures; i++) {
        const spcbpt_texture& t = sc->textures[i];
        if (!t.rgba || t.width < 1 || t.height < 1) { g_create_error = "bad texture"; delete c; return SPCBPT_ERR_INVALID_ARG; }
        uint32_t* d = nullptr;
        CREATE_TRY(dev_alloc(&d, (size_t)t.width * t.height));
        c->d_tex_data.push_back(d);
        CREATE_TRY(hipMemcpy(d, t.rgba, (size_t)t.width * t.height * 4, hipMemcpyHostToDevice));
        texs.push_back(DTexture{d, t.width, t.height});
    }
    CREATE_TRY(dev_alloc(&c->d_tex, texs.size()));
    if (!texs.empty()) CREATE_TRY(hipMemcpy(c->d_tex, texs.data(), texs.size() * sizeof(DTexture), hipMemcpyHostToDevice));
    CREATE_TRY(dev_alloc(&c->d_set_counts_all, (size_t)2 * Context::kMaxSets + 2));   // + a spare pair (probe_lvc_capacity)
    CREATE_TRY(hipMemset(c->d_set_counts_all, 0, (2 * Context::kMaxSets + 2) * sizeof(int)));
    for (int s = 0; s < c->n_sets; s++) {
        CREATE_TRY(dev_alloc(&c->set_subspace[s], (size_t)SPCBPT_NUM_SUBSPACE));
        c->set_counts[s] = c->d_set_counts_all + 2 * s;
    }
    c->select_set(0);
    {
        hipDeviceProp_t prop;
        CREATE_TRY(hipGetDeviceProperties(&prop, device));
        c->num_cus = prop.multiProcessorCount;
    }
    CREATE_TRY(dev_alloc(&c->d_work_counter, (size_t)Context::kMaxRender + 2));   // tile queues of the render streams + core queues of the two light lanes
    CREATE_TRY(dev_alloc(&c->d_diag, (size_t)4));
    CREATE_TRY(hipMemset(c->d_diag, 0, 4 * sizeof(uint32_t)));
    if (const char* e = getenv("SPCBPT_DEBUG_SPILL_ENTRIES")) c->spill_entries_debug = std::max(0, atoi(e));
    CREATE_TRY(dev_alloc(&c->d_counters, (size_t)C_COUNT));
    CREATE_TRY(hipMemset(c->d_counters, 0, C_COUNT * sizeof(unsigned long long)));
    memset(&c->kp, 0, sizeof(c->kp));
    c->kp.scene.nodes = c->d_nodes; c->kp.scene.tris = c->d_tris; c->kp.scene.tri_base = c->n_nodes; c->kp.scene.tri_orig = c->d_tri_orig; c->kp.scene.mats = c->d_mats;
    c->kp.scene.lights = c->d_lights; c->kp.scene.tex = c->d_tex; c->kp.scene.n_lights = c->n_lights; c->kp.scene.n_mats = c->n_mats;
    // the same nodes, one record per child: the quad tail of the pooled traversal pass (device_lib.h) and the traversal A/B harness
    CREATE_TRY(dev_alloc(&c->d_nodes_q, (size_t)c->n_nodes * 16));
    launch_repack_nodes_quad(c->d_nodes, c->d_nodes_q, c->n_nodes, c->stream);
    CREATE_TRY(hipGetLastError());
    CREATE_TRY(hipStreamSynchronize(c->stream));
    c->kp.scene.nodes_q = getenv("SPCBPT_NO_QUAD_TAIL") ? nullptr : c->d_nodes_q;
    c->kp.scene.fan_tail = getenv("SPCBPT_NO_FAN_TAIL") ? 0 : 1;
    c->kp.scene.general = 0;   // no environment map yet; a flagged material (Pbr::brdf) selects the general kernels as well
    for (const DMaterial& m : mats) if (m.brdf) c->kp.scene.general = 1;
    c->kp.sampler_counts = c->d_sampler_counts;
    c->kp.diag = c->d_diag;
    c->kp.row_step = 1;
    CREATE_TRY(hipDeviceSynchronize());   // the uploads above went through the default stream; the context's streams do not wait for it
#undef CREATE_TRY
    *out = c;
    return SPCBPT_OK;
}

int spcbpt_destroy(spcbpt_ctx* c) {
    if (!c) return SPCBPT_ERR_INVALID_ARG;
    spc_viewers_forget_context(c);   // viewers of this context live on as state machines without one (viewer.cpp)
    (void)hipSetDevice(c->device);
    (void)c->sync_all();
    delete c;
    return SPCBPT_OK;
}

const char* spcbpt_build_arithmetic(void) {
#if defined(SPCBPT_FAST_MATH_BUILD)
    return "approx";
#else
    return "ieee";
#endif
}

const char* spcbpt_last_error(const spcbpt_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

#define CTX_CHECK(c)                                  \
    if (!(c)) return SPCBPT_ERR_INVALID_ARG;          \
    if (hipSetDevice((c)->device) != hipSuccess) { (c)->error = "hipSetDevice failed"; return SPCBPT_ERR_HIP; }

int spcbpt_set_camera(spcbpt_ctx* c, const float eye[3], const float U[3], const float V[3], const float W[3]) {
    CTX_CHECK(c);
    if (!eye || !U || !V || !W) { c->error = "null camera vector"; return SPCBPT_ERR_INVALID_ARG; }
    memcpy(c->kp.eye, eye, 12); memcpy(c->kp.U, U, 12); memcpy(c->kp.V, V, 12); memcpy(c->kp.W, W, 12);
    c->have_camera = true;
    return SPCBPT_OK;
}

int spcbpt_set_camera_lookat(spcbpt_ctx* c, const float eye[3], const float lookat[3], const float up[3], float fov, float aspect) {
    CTX_CHECK(c);
    if (!eye || !lookat || !up) { c->error = "null camera vector"; return SPCBPT_ERR_INVALID_ARG; }
    // sutil::Camera::UVWFrame (sutil/Camera.cpp:34-45)
    auto cross = [](const float* a, const float* b, float* r) { r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0]; };
    auto norm = [](float* v) { float inv = 1.0f / sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] *= inv; v[1] *= inv; v[2] *= inv; };
    float W[3] = {lookat[0] - eye[0], lookat[1] - eye[1], lookat[2] - eye[2]}, U[3], V[3];
    float wlen = sqrtf(W[0] * W[0] + W[1] * W[1] + W[2] * W[2]);
    cross(W, up, U); norm(U);
    cross(U, W, V); norm(V);
    float vlen = wlen * tanf(0.5f * fov * 3.14159265358979323846f / 180.0f);
    for (int k = 0; k < 3; k++) V[k] *= vlen;
    float ulen = vlen * aspect;
    for (int k = 0; k < 3; k++) U[k] *= ulen;
    return spcbpt_set_camera(c, eye, U, V, W);
}

int spcbpt_resize(spcbpt_ctx* c, int w, int h) {
    CTX_CHECK(c);
    if (w < 1 || h < 1 || (long long)w * h > (1ll << 28)) { c->error = "bad image size"; return SPCBPT_ERR_INVALID_ARG; }
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    c->deferred.active = false;   // a deferred frame of the old size is dropped with its buffer
    dev_free(c->d_accum); dev_free(c->d_frame);
    HIP_TRY(c, dev_alloc(&c->d_accum, (size_t)w * h * 4));
    HIP_TRY(c, dev_alloc(&c->d_frame, (size_t)w * h));
    for (int s = 0; s < Context::kMaxRender; s++)
        for (int k = 0; k < kMaxBatchFrames; k++) { dev_free(c->d_result_b[s][k]); c->d_result_b[s][k] = nullptr; }   // re-allocated at the new size on demand
    for (int s = 0; s < c->n_render; s++) {
        dev_free(c->d_result[s]);
        HIP_TRY(c, dev_alloc(&c->d_result[s], (size_t)w * h * 4));
    }
    HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, (size_t)w * h * 16, c->rstream));
    HIP_TRY(c, hipMemsetAsync(c->d_frame, 0, (size_t)w * h * 4, c->rstream));
    HIP_TRY(c, hipStreamSynchronize(c->rstream));
    c->kp.width = w; c->kp.height = h; c->kp.accum = c->d_accum; c->kp.frame = c->d_frame;
    return SPCBPT_OK;
}

int spcbpt_set_subspace(spcbpt_ctx* c, const spcbpt_tree_node* et, int ne, const spcbpt_tree_node* lt, int nl, const float* q, const float* g) {
    CTX_CHECK(c);
    if (!et && !lt && !q && !g) return c->install_minimal_tuple();
    return c->install_subspace(et, ne, lt, nl, q, g);
}

int spcbpt_set_environment(spcbpt_ctx* c, const float* rgba, int width, int height, const float* center, float radius) {
    CTX_CHECK(c);
    return c->set_environment(rgba, width, height, center, radius);
}
int spcbpt_get_environment(spcbpt_ctx* c, int* width, int* height, float center[3], float* radius, int* n_lights) {
    CTX_CHECK(c);
    const DEnv& V = c->kp.scene.env;
    if (width) *width = V.valid ? V.width : 0;
    if (height) *height = V.valid ? V.height : 0;
    if (center) memcpy(center, V.center, 12);
    if (radius) *radius = V.valid ? V.r : 0.0f;
    if (n_lights) *n_lights = c->n_lights;
    return SPCBPT_OK;
}

int spcbpt_set_light_trace(spcbpt_ctx* c, const spcbpt_light_trace_params* p) {
    CTX_CHECK(c);
    if (!p) { c->error = "null params"; return SPCBPT_ERR_INVALID_ARG; }
    if (c->sync_all()) return SPCBPT_ERR_HIP;
    return c->set_light_trace(*p);
}

int spcbpt_lvc_set_capacity(spcbpt_ctx* c, int vertices) {
    CTX_CHECK(c);
    if (vertices < 0) { c->error = "lvc_set_capacity: negative capacity"; return SPCBPT_ERR_INVALID_ARG; }
    c->lvc_fixed = (size_t)vertices;
    if (c->sbb_keys) { if (c->sync_all()) return SPCBPT_ERR_HIP; c->free_batch_build_scratch(); }   // sized for the old capacity's builds
    c->sbb_refused_bytes = 0;
    if (vertices == 0) { c->lvc_probe_needed = true; return SPCBPT_OK; }   // back to the probe pass (the sets only ever grow)
    c->lvc_probe_needed = false;
    return c->ensure_lvc_capacity((size_t)vertices);
}
int spcbpt_lvc_get_capacity(spcbpt_ctx* c, int* vertices, int* sets) {
    CTX_CHECK(c);
    if (vertices) *vertices = (int)std::min<size_t>(c->lvc_capacity, 0x7fffffff);
    if (sets) *sets = c->n_sets;
    return SPCBPT_OK;
}

int spcbpt_launch(spcbpt_ctx* c, const char* name, uint32_t frame, int r0, int r1, int rs) {
    CTX_CHECK(c);
    if (!name) { c->error = "null algorithm name"; return SPCBPT_ERR_INVALID_ARG; }
    const std::string alg(name);
    if (alg == "light trace") return c->launch_light(frame);
    if (alg == "SPCBPT_eye") return c->launch_render("spcbpt_render", true, frame, r0, r1, rs);
    if (alg == "pt") return c->launch_render("pt", false, frame, r0, r1, rs);
    if (alg == "SPCBPT_no_rmis") return c->launch_render("spcbpt_no_rmis", true, frame, r0, r1, rs, true);   // raygen.cu:465: defined upstream, wired to no program group
    if (alg == "pretrace") return c->launch_pretrace(frame);
    c->error = "unknown algorithm '" + alg + "' (expected \"pt\", \"light trace\", \"SPCBPT_eye\", \"pretrace\" or \"SPCBPT_no_rmis\")";
    return SPCBPT_ERR_UNKNOWN_ALG;
}

int spcbpt_launch_deferred(spcbpt_ctx* c, const char* name, uint32_t frame, int r0, int r1, int rs) {
    CTX_CHECK(c);
    if (!name) { c->error = "null algorithm name"; return SPCBPT_ERR_INVALID_ARG; }
    const std::string alg(name);
    if (alg == "SPCBPT_eye") return c->launch_render("spcbpt_render", true, frame, r0, r1, rs, false, true);
    if (alg == "pt") return c->launch_render("pt", false, frame, r0, r1, rs, false, true);
    c->error = "launch_deferred: \"pt\" or \"SPCBPT_eye\"";
    return SPCBPT_ERR_UNKNOWN_ALG;
}
int spcbpt_merge_deferred(spcbpt_ctx* c, int keep) { CTX_CHECK(c); return c->merge_deferred(keep != 0); }
int spcbpt_sync_film(spcbpt_ctx* c) { CTX_CHECK(c); return c->sync_film(); }

int spcbpt_launch_eye_batch(spcbpt_ctx* c, int n_frames, const uint32_t* subframes, int r0, int r1, int rs) {
    CTX_CHECK(c);
    return c->launch_eye_batch(n_frames, subframes, r0, r1, rs);
}

int spcbpt_build_sampler_batch(spcbpt_ctx* c, int n_builds) {
    CTX_CHECK(c);
    return c->build_sampler_batch(n_builds);
}
int spcbpt_launch_light_batch(spcbpt_ctx* c, uint32_t first_frame, int n_frames) {
    CTX_CHECK(c);
    return c->launch_light_batch(first_frame, n_frames);
}

int spcbpt_build_sampler(spcbpt_ctx* c) {
    CTX_CHECK(c);
    return c->build_sampler();
}

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

// Context: timing spans, buffer sets, the subspace tuple, the environment, light-pass geometry, capacities, lifetime
// (part of the C ABI library: see capi_common.h for the map of its translation units)
#include "capi_common.h"

using namespace spc;

namespace spc {

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
    if (const char* lim = getenv("SPCBPT_DEBUG_LVC_LIMIT")) {   // tests: pretend the device refuses a cache of more than this many vertices per set
        if (n > (size_t)strtoull(lim, nullptr, 10)) e = hipErrorOutOfMemory;
    }
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
    dev_free(d_nodes); dev_free(d_nodes_q); dev_free(d_nodes_q2); dev_free(d_tris); dev_free(d_tri_orig); dev_free(d_mats); dev_free(d_lights); dev_free(d_tex);
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

// C ABI (include/spcbpt.h) over the HIP kernels: create / destroy, state, launches by name.  The host side of the reference this
// replaces is cited per function in include/spcbpt.h.  (Read-backs, counters and test hooks: capi_debug.hip; exchange: capi_exchange.hip.)
#include "capi_common.h"

using namespace spc;

static thread_local std::string g_create_error;

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

    // nodes and the PAIR records of the triangles (lbvh.h: Lbvh::pairs, one 64-B slot per triangle) in ONE allocation, the pair records
    // right behind the node records: the pooled traversal step fetches "the record of its next step" through one base pointer
    // (dev_traversal.h: SPC_FETCH_STEP__).  The per-triangle records (corners + UVs + material: what the tails, the one-ray-per-lane
    // loop and the shading read) live in an allocation of their own.
    CREATE_TRY(dev_alloc(&c->d_nodes, bvh.nodes.size() + bvh.pairs.size()));
    CREATE_TRY(dev_alloc(&c->d_tris, bvh.tris.size()));
    if (getenv("SPCBPT_NO_TRI_PAIRS")) {   // test switch: every slot a single triangle (the same device code, one test per step): films must not change
        for (size_t i = 0; i < bvh.pairs.size() / 16; i++) { uint32_t fl; memcpy(&fl, &bvh.pairs[i * 16 + 15], 4); fl &= 0x80000000u; memcpy(&bvh.pairs[i * 16 + 15], &fl, 4); }
        bvh.n_paired = 0;
    }
    CREATE_TRY(hipMemcpy(c->d_nodes + bvh.nodes.size(), bvh.pairs.data(), bvh.pairs.size() * 4, hipMemcpyHostToDevice));
    c->n_paired = bvh.n_paired;
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

}  // extern "C"

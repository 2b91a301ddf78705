// ORACLE — test infrastructure only (see vec.h).  C entry points so tests,
// smoke() and bench.py's cpu_baseline leg can drive the CPU restatement through
// ctypes.  Mirrors the frame sequence of optixPathTracer.cpp:491-635
// (launchLightTrace -> LVC_Process -> launchSubframe).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "oracle_ctx.h"
#include "spcbpt_ref.h"

using namespace orc;



static void copy_tree(const spcbpt_tree_node* in, int n, std::vector<tree_node>& out) {
    out.resize(n);
    for (int i = 0; i < n; i++) {
        out[i].mid = load3(in[i].mid);
        for (int k = 0; k < 8; k++) out[i].child[k] = in[i].child[k];
        out[i].label = in[i].label;
        out[i].type = in[i].type;
        out[i].leaf = in[i].leaf != 0;
    }
}

template <class F>
static void parallel_for(int n, int nthreads, F f) {
    if (nthreads <= 1) {
        for (int i = 0; i < n; i++) f(i, 0);
        return;
    }
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t]() {
            for (;;) {
                int i = next.fetch_add(1);
                if (i >= n) break;
                f(i, t);
            }
        });
    for (auto& x : th) x.join();
}

extern "C" {

orc_ctx* orc_create(const spcbpt_scene_desc* d) {
    orc_ctx* c = new orc_ctx();
    c->scene.build(*d);
    c->P.scene = &c->scene;
    return c;
}
void orc_destroy(orc_ctx* c) { orc_free_pre(c); delete c; }

int orc_set_camera(orc_ctx* c, const float* eye, const float* U, const float* V, const float* W) {
    c->P.eye = load3(eye); c->P.U = load3(U); c->P.V = load3(V); c->P.W = load3(W);
    return 0;
}
// sutil::Camera::UVWFrame (sutil/Camera.cpp:34-45)
int orc_camera_frame(const float* eye, const float* lookat, const float* up, float fovY, float aspect, float* U, float* V, float* W) {
    float3 w = load3(lookat) - load3(eye);
    float wlen = length(w);
    float3 u = normalize(cross(w, load3(up)));
    float3 v = normalize(cross(u, w));
    float vlen = wlen * tanf(0.5f * fovY * M_PIf_ / 180.0f);
    v *= vlen;
    float ulen = vlen * aspect;
    u *= ulen;
    U[0] = u.x; U[1] = u.y; U[2] = u.z; V[0] = v.x; V[1] = v.y; V[2] = v.z; W[0] = w.x; W[1] = w.y; W[2] = w.z;
    return 0;
}
int orc_resize(orc_ctx* c, int w, int h) {
    c->P.width = w; c->P.height = h;
    c->accum.assign((size_t)w * h, float4{0, 0, 0, 0});
    c->frame.assign((size_t)w * h, 0u);
    c->P.accum_buffer = c->accum.data();
    c->P.frame_buffer = c->frame.data();
    return 0;
}
int orc_set_subspace(orc_ctx* c, const spcbpt_tree_node* eye_tree, int n_eye, const spcbpt_tree_node* light_tree,
                     int n_light, const float* q, const float* cmf_gamma) {
    if (eye_tree) { copy_tree(eye_tree, n_eye, c->eye_tree); c->P.eye_tree = c->eye_tree.data(); } else c->P.eye_tree = nullptr;
    if (light_tree) { copy_tree(light_tree, n_light, c->light_tree); c->P.light_tree = c->light_tree.data(); } else c->P.light_tree = nullptr;
    if (q) { c->Q.assign(q, q + SPCBPT_NUM_SUBSPACE); c->P.Q = c->Q.data(); } else c->P.Q = nullptr;
    if (cmf_gamma) {
        c->CMFGamma.assign(cmf_gamma, cmf_gamma + (size_t)SPCBPT_NUM_SUBSPACE * SPCBPT_NUM_SUBSPACE);
        c->P.CMFGamma = c->CMFGamma.data();
    } else c->P.CMFGamma = nullptr;
    return 0;
}
int orc_set_light_decorrelate(orc_ctx* c, int on) { c->P.lt.decorrelate_bsdf_stream = on != 0; return 0; }
int orc_set_light_trace(orc_ctx* c, int num_core, int core_padding, int m_per_core) {
    c->P.lt.num_core = num_core; c->P.lt.core_padding = core_padding; c->P.lt.M_per_core = m_per_core;
    c->lvc.assign((size_t)num_core * core_padding, BDPTVertex());
    c->lvc_valid.assign((size_t)num_core * core_padding, 0);
    c->P.lt.ans = c->lvc.data();
    c->P.lt.validState = c->lvc_valid.data();
    return 0;
}
// env_params_setup + the ENV entry of LightSource_shift: `rgba` = width x height RGBA floats as the .hdr file stores them (row 0 = top)
int orc_set_environment(orc_ctx* c, const float* rgba, int w, int h, const float* center, float radius) {
    if (!rgba || w < 1 || h < 1 || !center || !(radius > 0)) return SPCBPT_ERR_INVALID_ARG;
    c->scene.set_environment(rgba, w, h, load3(center), radius);
    return 0;
}
int orc_set_cmf_double(orc_ctx* c, int on) { c->P.cmf_double = on != 0; return 0; }
int orc_set_skip_null_connections(orc_ctx* c, int on) { c->P.skip_null_connections = on != 0; return 0; }
int orc_set_pt_env_nee_fixed(orc_ctx* c, int on) { c->P.pt_env_nee_fixed = on != 0; return 0; }
int orc_set_env_miss_strategy(orc_ctx* c, int on) { c->P.env_miss_strategy = on != 0; return 0; }
int orc_set_count_as_executed(orc_ctx* c, int on) { c->P.count_as_executed = on != 0; return 0; }
int orc_set_uniform_lvc(orc_ctx* c, int on) { c->P.uniform_lvc = on != 0; return 0; }
int orc_enable_counters(orc_ctx* c, int on) { c->count_events = on != 0; return 0; }

// switchRaygen(name) + optixLaunch
int orc_launch(orc_ctx* c, const char* name, unsigned frame, int row_begin, int row_end, int row_step, int nthreads) {
    std::string alg(name);
    if (nthreads < 1) nthreads = 1;
    std::vector<Counters> tc(nthreads);
    std::vector<Params> tp(nthreads);  // per-thread views (Params holds only pointers and scalars)
    auto make_views = [&]() {
        for (int t = 0; t < nthreads; t++) {
            tp[t] = c->P;
            tp[t].counters = c->count_events ? &tc[t] : nullptr;
        }
    };
    if (alg == "light trace") {
        c->P.lt.launch_frame = (int)frame;
        if (c->lvc.empty()) orc_set_light_trace(c, c->P.lt.num_core, c->P.lt.core_padding, c->P.lt.M_per_core);
        make_views();
        parallel_for(c->P.lt.num_core, nthreads, [&](int i, int t) { raygen_lightTrace(tp[t], i); });
    } else if (alg == "pt" || alg == "SPCBPT_eye" || alg == "SPCBPT_no_rmis") {
        if (alg != "pt" && !c->P.sampler.subspace) return SPCBPT_ERR_STATE;
        if (!c->P.accum_buffer) return SPCBPT_ERR_STATE;
        c->P.subframe_index = frame;
        if (row_step < 1) row_step = 1;
        std::vector<int> rows;
        // 8-row bands, band k of every row_step (same rule as spcbpt_launch)
        for (int y = row_begin; y < row_end && y < (int)c->P.height; y++)
            if (((y / 8) - (row_begin / 8)) % row_step == 0) rows.push_back(y);
        bool spc = alg == "SPCBPT_eye", full_mis = alg == "SPCBPT_no_rmis";
        make_views();
        // work items of the dynamic queue: 8-pixel-wide pieces of a row (the rows of a band are consecutive in `rows`, so the items of
        // a band's 8 x 8 tiles are neighbours in the queue).  Whole rows -- 1 080 items for 256 threads -- left the last threads of a
        // frame idle for a fifth of it; pixels are independent, so the image does not depend on the split (test_thread_count_invariance).
        const int xt = ((int)c->P.width + 7) / 8;
        parallel_for((int)rows.size() * xt, nthreads, [&](int i, int t) {
            const unsigned y = (unsigned)rows[i / xt];
            const unsigned x0 = (unsigned)(i % xt) * 8u, x1 = std::min(x0 + 8u, c->P.width);
            for (unsigned x = x0; x < x1; x++) {
                if (spc) raygen_SPCBPT(tp[t], x, y);
                else if (full_mis) raygen_SPCBPT_no_rmis(tp[t], x, y);
                else raygen_pinhole(tp[t], x, y);
            }
        });
    } else {
        return SPCBPT_ERR_UNKNOWN_ALG;  // "pretrace" lives in oracle/preprocess
    }
    for (auto& t : tc) c->counters.add(t);
    return 0;
}

int orc_build_sampler(orc_ctx* c) { LVC_Process(c->P, c->sampler_storage); return 0; }

static void export_vertex(const BDPTVertex& v, spcbpt_light_vertex& o) {
    memset(&o, 0, sizeof(o));
    o.position[0] = v.position.x; o.position[1] = v.position.y; o.position[2] = v.position.z;
    o.pdf = v.pdf;
    o.normal[0] = v.normal.x; o.normal[1] = v.normal.y; o.normal[2] = v.normal.z;
    o.single_pdf = v.singlePdf;
    o.flux[0] = v.flux.x; o.flux[1] = v.flux.y; o.flux[2] = v.flux.z;
    o.rmis_pointer = v.RMIS_pointer;
    o.color[0] = v.color.x; o.color[1] = v.color.y; o.color[2] = v.color.z;
    o.last_lum = v.last_lum;
    o.last_position[0] = v.lastPosition.x; o.last_position[1] = v.lastPosition.y; o.last_position[2] = v.lastPosition.z;
    o.last_normal_projection = v.lastNormalProjection;
    o.material_id = v.materialId; o.subspace_id = v.subspaceId; o.depth = v.depth; o.last_zone_id = v.lastZoneId;
    o.path_id = v.path_id;
    o.pad = (v.type == ENV ? SPCBPT_LV_DIRECTION : 0u) | (v.isLastVertex_direction ? SPCBPT_LV_LAST_DIRECTION : 0u);   // (no cached label: bits 0-15 stay 0)
}
static void import_vertex(const spcbpt_light_vertex& o, BDPTVertex& v) {
    v = BDPTVertex();
    v.position = load3(o.position); v.pdf = o.pdf;
    v.normal = load3(o.normal); v.singlePdf = o.single_pdf;
    v.flux = load3(o.flux); v.RMIS_pointer = o.rmis_pointer;
    v.color = load3(o.color); v.last_lum = o.last_lum;
    v.lastPosition = load3(o.last_position); v.lastNormalProjection = o.last_normal_projection;
    v.materialId = o.material_id; v.subspaceId = o.subspace_id; v.depth = o.depth; v.lastZoneId = o.last_zone_id;
    v.path_id = o.path_id;
    v.isOrigin = o.depth == 0;
    v.type = o.depth == 0 ? ((o.pad & SPCBPT_LV_DIRECTION) ? ENV : QUAD) : NORMALHIT;
    v.isLastVertex_direction = (o.pad & SPCBPT_LV_LAST_DIRECTION) != 0;
}
// valid LVC slots in slot order (= (path_id, depth) order)
int orc_lvc_read(orc_ctx* c, spcbpt_light_vertex* out, int capacity, int* count) {
    int n = 0;
    const LightTraceParams& lt = c->P.lt;
    for (size_t i = 0; i < c->lvc.size(); i++) {
        if (!lt.validState[i]) continue;
        if (out && n < capacity) export_vertex(lt.ans[i], out[n]);
        n++;
    }
    *count = n;
    return (out && n > capacity) ? SPCBPT_ERR_CAPACITY : 0;
}
// replace the LVC by a compact list (slot i = vertex i)
int orc_lvc_import(orc_ctx* c, const spcbpt_light_vertex* in, int count) {
    LightTraceParams& lt = c->P.lt;
    lt.num_core = 1; lt.core_padding = count;
    c->lvc.resize(count);
    c->lvc_valid.assign(count, 1);
    lt.ans = c->lvc.data();
    lt.validState = c->lvc_valid.data();
    for (int i = 0; i < count; i++) import_vertex(in[i], lt.ans[i]);
    return 0;
}
// jump entries are returned as indices into the COMPACT valid list (orc_lvc_read order)
int orc_sampler_read(orc_ctx* c, spcbpt_subspace* subspace, float* cmfs, int32_t* jump, int capacity, int* vertex_count, int* path_count) {
    const SubspaceSampler& s = c->P.sampler;
    if (!s.subspace) return SPCBPT_ERR_STATE;
    *vertex_count = s.vertex_count; *path_count = s.path_count;
    for (int i = 0; i < SPCBPT_NUM_SUBSPACE; i++) {
        subspace[i].jump_bias = s.subspace[i].jump_bias; subspace[i].id = s.subspace[i].id;
        subspace[i].size = s.subspace[i].size; subspace[i].sum_pmf = s.subspace[i].sum_pmf; subspace[i].q = 0;
    }
    if (s.vertex_count > capacity) return SPCBPT_ERR_CAPACITY;
    std::vector<int> compact(c->lvc.size(), -1);
    int n = 0;
    for (size_t i = 0; i < c->lvc.size(); i++) if (c->lvc_valid[i]) compact[i] = n++;
    for (int i = 0; i < s.vertex_count; i++) { cmfs[i] = s.cmfs[i]; jump[i] = compact[s.jump_buffer[i]]; }
    return 0;
}
int orc_read_accum(orc_ctx* c, float* out) { memcpy(out, c->accum.data(), c->accum.size() * sizeof(float4)); return 0; }
int orc_read_frame(orc_ctx* c, uint8_t* out) { memcpy(out, c->frame.data(), c->frame.size() * 4); return 0; }
int orc_clear_accum(orc_ctx* c) { std::fill(c->accum.begin(), c->accum.end(), float4{0, 0, 0, 0}); return 0; }
int orc_get_counters(orc_ctx* c, spcbpt_counters* o) {
    const Counters& k = c->counters;
    o->closest_rays = k.closest_rays; o->shadow_rays = k.shadow_rays; o->node_visits = k.node_visits; o->tri_tests = k.tri_tests;
    o->surface_vertices = k.surface_vertices; o->textured_hits = k.textured_hits; o->tree_nodes = k.tree_nodes;
    o->cmf_probes = k.cmf_probes; o->connections = k.connections; o->gamma_q_reads = k.gamma_q_reads;
    o->lvc_stores = k.lvc_stores; o->pixel_samples = k.pixel_samples; o->eye_paths = k.eye_paths; o->light_paths = k.light_paths;
    return 0;
}
int orc_reset_counters(orc_ctx* c) { c->counters = Counters(); return 0; }
int orc_scene_info(orc_ctx* c, int* n_tri, int* n_nodes) { *n_tri = c->scene.n_triangles(); *n_nodes = (int)c->scene.nodes.size(); return 0; }

// rays: n*8 floats (o, tmin, d, tmax)
int orc_trace_closest(orc_ctx* c, const float* rays, int n, float* out_t, int32_t* out_tri, float* out_uv, int nthreads) {
    parallel_for((n + 1023) / 1024, nthreads, [&](int blk, int) {
        for (int i = blk * 1024; i < n && i < (blk + 1) * 1024; i++) {
            const float* r = rays + 8 * (size_t)i;
            Hit h = c->scene.closest_hit(load3(r), load3(r + 4), r[3], r[7], nullptr);
            out_t[i] = h.t; out_tri[i] = h.tri; out_uv[2 * i] = h.bu; out_uv[2 * i + 1] = h.bv;
        }
    });
    return 0;
}
int orc_trace_any(orc_ctx* c, const float* rays, int n, int32_t* out_visible, int nthreads) {
    parallel_for((n + 1023) / 1024, nthreads, [&](int blk, int) {
        for (int i = blk * 1024; i < n && i < (blk + 1) * 1024; i++) {
            const float* r = rays + 8 * (size_t)i;
            out_visible[i] = c->scene.any_hit(load3(r), load3(r + 4), r[3], r[7], nullptr) ? 0 : 1;
        }
    });
    return 0;
}

// ---- per-function hooks for unit parity tests ----
uint32_t orc_tea4(uint32_t a, uint32_t b) { return tea<4>(a, b); }
float orc_rnd(uint32_t* seed) { return rnd(*seed); }
static Pbr pbr_of(const spcbpt_material* m) {
    Pbr p;
    p.base_color = load3(m->base_color); p.metallic = m->metallic; p.roughness = m->roughness; p.specular = m->specular;
    p.specularTint = m->specular_tint; p.subsurface = m->subsurface; p.sheen = m->sheen; p.sheenTint = m->sheen_tint;
    p.clearcoat = m->clearcoat; p.clearcoatGloss = m->clearcoat_gloss; p.albedo_tex = m->albedo_tex; p.brdf = m->brdf != 0;
    return p;
}
// n records: N(3) V(3) L(3) -> f(3), pdf(1)
int orc_bsdf_eval_pdf(const spcbpt_material* m, const float* nvl, int n, float* out_f, float* out_pdf) {
    Pbr p = pbr_of(m);
    for (int i = 0; i < n; i++) {
        const float* r = nvl + 9 * (size_t)i;
        float3 f = Eval(p, load3(r), load3(r + 3), load3(r + 6));
        out_f[3 * i] = f.x; out_f[3 * i + 1] = f.y; out_f[3 * i + 2] = f.z;
        out_pdf[i] = Pdf(p, load3(r), load3(r + 3), load3(r + 6));
    }
    return 0;
}
// n records: N(3) V(3), seeds[n] (updated) -> L(3)
int orc_bsdf_sample(const spcbpt_material* m, const float* nv, uint32_t* seeds, int n, float* out_l) {
    Pbr p = pbr_of(m);
    for (int i = 0; i < n; i++) {
        const float* r = nv + 6 * (size_t)i;
        float3 l = Sample(p, load3(r), load3(r + 3), seeds[i]);
        out_l[3 * i] = l.x; out_l[3 * i + 1] = l.y; out_l[3 * i + 2] = l.z;
    }
    return 0;
}
int orc_binary_sample(const float* cmf, int size, uint32_t* seed, float* pmf) {
    Params P;
    return binary_sample(P, cmf, size, *seed, *pmf);
}
int orc_tree_index(const spcbpt_tree_node* tree, int n_nodes, const float* pnd, int n, int32_t* out) {
    std::vector<tree_node> t;
    copy_tree(tree, n_nodes, t);
    for (int i = 0; i < n; i++) {
        const float* r = pnd + 9 * (size_t)i;
        out[i] = tree_index(t.data(), load3(r), load3(r + 3), load3(r + 6), nullptr);
    }
    return 0;
}
// ToneMap(c, 1.5) + make_color : n rgb triples -> n RGBA8
int orc_tonemap(const float* rgb, int n, uint8_t* out) {
    for (int i = 0; i < n; i++) {
        float4 v = ToneMap(load3(rgb + 3 * (size_t)i), 1.5f);
        uint32_t c = make_color(make_float3(v.x, v.y, v.z));
        memcpy(out + 4 * (size_t)i, &c, 4);
    }
    return 0;
}
int orc_srgb(const float* rgb, int n, float* out_srgb, uint8_t* out_q) {
    for (int i = 0; i < n; i++) {
        float3 s = toSRGB(load3(rgb + 3 * (size_t)i));
        out_srgb[3 * i] = s.x; out_srgb[3 * i + 1] = s.y; out_srgb[3 * i + 2] = s.z;
        uint32_t c = make_color(load3(rgb + 3 * (size_t)i));
        memcpy(out_q + 4 * (size_t)i, &c, 4);
    }
    return 0;
}
// connection value for explicit vertices (a = eye vertex record, b = light vertex) — used by parity tests of
// connectVertex_SPCBPT + rmis on identical inputs.  eye record: spcbpt_light_vertex fields + rmis3 + single_pdf etc.
typedef struct orc_eye_vertex {
    float position[3], normal[3], flux[3], color[3], last_position[3], rmis3[3];
    float pdf, single_pdf, last_normal_projection;
    int32_t material_id, subspace_id, depth, last_zone_id;
} orc_eye_vertex;
int orc_connect(orc_ctx* c, const orc_eye_vertex* a, const spcbpt_light_vertex* b, int n, float* out_rgb, float* out_w) {
    for (int i = 0; i < n; i++) {
        BDPTVertex ev;
        ev.position = load3(a[i].position); ev.normal = load3(a[i].normal); ev.flux = load3(a[i].flux);
        ev.color = load3(a[i].color); ev.lastPosition = load3(a[i].last_position);
        ev.RMIS_pointer_3 = load3(a[i].rmis3); ev.pdf = a[i].pdf; ev.singlePdf = a[i].single_pdf;
        ev.lastNormalProjection = a[i].last_normal_projection; ev.materialId = (short)a[i].material_id;
        ev.subspaceId = (short)a[i].subspace_id; ev.depth = (short)a[i].depth; ev.lastZoneId = (short)a[i].last_zone_id;
        ev.type = NORMALHIT;
        BDPTVertex lv;
        import_vertex(b[i], lv);
        float3 r = connectVertex_SPCBPT(c->P, ev, lv);
        out_rgb[3 * i] = r.x; out_rgb[3 * i + 1] = r.y; out_rgb[3 * i + 2] = r.z;
        out_w[i] = lv.is_DIRECTION() ? rmis::connection_direction_lightSource(c->P, ev, lv)
                 : lv.depth == 0 ? rmis::connection_lightSource(c->P, ev, lv) : rmis::general_connection(c->P, ev, lv);
    }
    return 0;
}

static void import_eye_vertex(const orc_eye_vertex& a, BDPTVertex& ev) {
    ev.position = load3(a.position); ev.normal = load3(a.normal); ev.flux = load3(a.flux);
    ev.color = load3(a.color); ev.lastPosition = load3(a.last_position);
    ev.RMIS_pointer_3 = load3(a.rmis3); ev.pdf = a.pdf; ev.singlePdf = a.single_pdf;
    ev.lastNormalProjection = a.last_normal_projection; ev.materialId = (short)a.material_id;
    ev.subspaceId = (short)a.subspace_id; ev.depth = (short)a.depth; ev.lastZoneId = (short)a.last_zone_id;
    ev.type = NORMALHIT;
    ev.isOrigin = a.depth == 0;
}
static void export_eye_vertex(const BDPTVertex& v, orc_eye_vertex& a) {
    auto st3 = [](float* d, float3 s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; };
    st3(a.position, v.position); st3(a.normal, v.normal); st3(a.flux, v.flux); st3(a.color, v.color);
    st3(a.last_position, v.lastPosition); st3(a.rmis3, v.RMIS_pointer_3);
    a.pdf = v.pdf; a.single_pdf = v.singlePdf; a.last_normal_projection = v.lastNormalProjection;
    a.material_id = v.materialId; a.subspace_id = v.subspaceId; a.depth = v.depth; a.last_zone_id = v.lastZoneId;
}
// One step of the eye walk from an explicit state: traceEyeSubPath + the closest-hit program the SBT picks
// (__closesthit__eyeSubpath / __closesthit__eyeSubpath_LightSource with rmis::light_hit) + lightStraghtHit, on the record
// layout of spcbpt_debug_unit(SPCBPT_UNIT_EYE_STEP) (include/spcbpt.h): the per-function parity of rows a8 / a10 / a16.
typedef struct orc_eye_step_in { orc_eye_vertex last; float next_flux[3]; float next_single_pdf; uint32_t seed; float dir[3]; uint32_t flags; uint32_t pad[2]; } orc_eye_step_in;
typedef struct orc_eye_step_out {
    uint32_t kind; orc_eye_vertex mid; float dir[3]; float next_flux[3]; float next_single_pdf; uint32_t seed; uint32_t done; float emit[3]; float t_hit; uint32_t pad;
} orc_eye_step_out;
static_assert(sizeof(orc_eye_step_in) == 36 * 4 && sizeof(orc_eye_step_out) == 40 * 4, "unit record sizes (include/spcbpt.h)");
int orc_eye_step(orc_ctx* c, const orc_eye_step_in* in, int n, orc_eye_step_out* out) {
    const Scene& S = c->scene;
    for (int i = 0; i < n; i++) {
        Params P = c->P;
        P.counters = nullptr;
        P.skip_null_connections = (in[i].flags & 1u) != 0;
        PayloadBDPTVertex prd;
        prd.clear();
        prd.path.size = 1;
        import_eye_vertex(in[i].last, prd.path.currentVertex());
        prd.path.nextVertex().flux = load3(in[i].next_flux);
        prd.path.nextVertex().singlePdf = in[i].next_single_pdf;
        prd.seed = in[i].seed;
        prd.origin = prd.path.currentVertex().position;
        prd.ray_direction = load3(in[i].dir);
        orc_eye_step_out& o = out[i];
        memset(&o, 0, sizeof(o));
        const float3 d = prd.ray_direction;
        Hit h = S.closest_hit(prd.origin, d, SPCBPT_SCENE_EPSILON, 1e16f, nullptr);
        if (h.tri < 0) { o.kind = 0; continue; }
        o.t_hit = h.t;
        HitInfo hi{h.tri, h.t, d, h.bu, h.bv};
        const int size0 = prd.path.size;
        if (S.tri_is_emitter(h.tri)) {
            closesthit_eyeSubpath_LightSource(P, &prd, hi);
            o.kind = prd.path.size == size0 ? 3u : 2u;   // back side: no vertex
            if (prd.path.size != size0) {
                float3 e = lightStraghtHit(prd.path.currentVertex());
                o.emit[0] = e.x; o.emit[1] = e.y; o.emit[2] = e.z;
            }
        } else {
            closesthit_subpath(P, &prd, hi, false);
            o.kind = 1;
            export_eye_vertex(prd.path.currentVertex(), o.mid);
            o.dir[0] = prd.ray_direction.x; o.dir[1] = prd.ray_direction.y; o.dir[2] = prd.ray_direction.z;
            const BDPTVertex& nx = prd.path.nextVertex();
            o.next_flux[0] = nx.flux.x; o.next_flux[1] = nx.flux.y; o.next_flux[2] = nx.flux.z;
            o.next_single_pdf = nx.singlePdf; o.seed = prd.seed; o.done = prd.done ? 1u : 0u;
        }
    }
    return 0;
}

}  // extern "C"

// ---- test utility: do the recursive-MIS weights of the sky's strategies form a partition of unity?  For `n` camera paths that
// leave the scene after exactly `depth` surface vertices (c, x1 .. xD, sky direction w), every strategy that can produce the path
// is built by the generation code itself -- the eye sub-path as traced, the light sub-path y0 (the sky direction w), y1 = xD,
// y2 = xD-1 ... re-traced from the sky with the scattering directions forced -- and its weight is evaluated by the very functions
// the renderers call: connection_direction_lightSource(eD, y0), general_connection(e_d, y_k) for d + k = D, light_hit_env for the
// miss.  out[7 * i]: the weights' sum; [1]: miss; [2 .. 5]: k = 0 .. 3; [6]: sum of the weights RECOMPUTED from first principles
// (rate of a strategy = eye pdf x connectRate_SOL x light pdf; miss = eye pdf x solid-angle pdf) -- 1 by construction, a check of
// the check.  Returns the number of paths found.
extern "C" int orc_debug_env_partition(orc_ctx* c, int depth, int n, unsigned frame, float* out, float* truth, orc_eye_vertex* ev_out, spcbpt_light_vertex* lv_out) {
    Params P = c->P;
    P.counters = nullptr;
    const Scene& S = *P.scene;
    if (!S.sky.valid || depth < 1 || depth > 4) return -1;
    int found = 0;
    for (unsigned pix = 0; found < n && pix < 4000000u; pix++) {
        const unsigned x = pix % P.width, y = (pix / P.width) % P.height;
        uint32_t seed = tea<4>(y * P.width + x, frame + pix / (P.width * P.height));
        float jx = rnd(seed), jy = rnd(seed);
        const float dx = 2.0f * (((float)x + jx) / (float)P.width) - 1.0f, dy = 2.0f * (((float)y + jy) / (float)P.height) - 1.0f;
        float3 dir = normalize(dx * P.U + dy * P.V + P.W), org = P.eye;
        PayloadBDPTVertex prd;
        prd.clear(); prd.seed = seed; prd.ray_direction = dir; prd.origin = org;
        init_EyeSubpath(prd.path, org, dir);
        std::vector<BDPTVertex> ev;   // e1 .. eD
        bool ok = true;
        for (int d = 0; d < depth; d++) {
            const int before = prd.path.size;
            trace_subpath(P, prd.origin, prd.ray_direction, &prd, false);
            if (prd.path.size == before || prd.path.hit_lightSource() || prd.done) { ok = false; break; }
            ev.push_back(prd.path.currentVertex());
        }
        if (!ok) continue;
        // the next segment must leave the scene
        const float3 w = prd.ray_direction;
        if (S.closest_hit(prd.origin, w, SPCBPT_SCENE_EPSILON, 1e16f, nullptr).tri >= 0) continue;
        const float next_single_pdf = prd.path.nextVertex().singlePdf;   // solid-angle pdf x RR of the direction w at xD
        // ---- miss strategy
        Params Pm = P; Pm.env_miss_strategy = true;
        PayloadBDPTVertex pm = prd;
        trace_subpath(Pm, prd.origin, w, &pm, false);
        if (!(float3weight(pm.path.currentVertex().flux) > 0.0f)) continue;   // (as in orc_debug_quad_partition: a zero-flux path)
        const float w_miss = 1.0f / pm.path.currentVertex().RMIS_pointer;
        // ---- the light sub-path of the same path: y0 = direction w with the disk point above xD, y1 = xD, y2 = xD-1, ...
        const Light& sky_light = S.lights.back();
        lightSample ls;
        ls.bindLight = &sky_light; ls.direction = w; ls.emission = S.sky.color(w); ls.subspaceId = S.sky.getLabel(w); ls.uv = dir2uv(w);
        ls.pdf = S.sky.pdf(w) / (float)S.lights.size();
        const float3 xD = ev[depth - 1].position;
        ls.position = xD + dot(S.sky.center + 10 * S.sky.r * w - xD, w) * w;
        ls.dir_pdf = S.sky.projectPdf();
        PayloadBDPTVertex pl;
        pl.clear();
        init_lightSubPath_from_lightSample(ls, pl.path);
        std::vector<BDPTVertex> lv;   // y0 .. yD
        lv.push_back(pl.path.currentVertex());
        float3 lo = ls.position, ld = -w;
        for (int k = 1; k <= depth && ok; k++) {
            const int before = pl.path.size;
            pl.done = false;
            trace_subpath(P, lo, ld, &pl, true);
            if (pl.path.size == before) { ok = false; break; }
            BDPTVertex& Mid = pl.path.currentVertex();
            const BDPTVertex& want = ev[depth - k];
            if (length(Mid.position - want.position) > 1e-3f) { ok = false; break; }
            lv.push_back(Mid);
            if (k == depth) break;
            // force the scattered direction towards the next vertex of the path (x_{D-k}) and set what Sample / Pdf / Eval would have
            const float3 nd = normalize(ev[depth - k - 1].position - Mid.position);
            Pbr pbr = S.materials[Mid.materialId];
            pbr.base_color = Mid.color;
            BDPTVertex& Next = pl.path.nextVertex();
            Next.flux = Eval(pbr, Mid.normal, -ld, nd) / (pbr.brdf ? fabsf(dot(Mid.normal, nd)) : 1.0f);   // hit_program.cu:384
            Next.singlePdf = Pdf(pbr, Mid.normal, -ld, nd) * rr_rate_of(Mid.color);
            lo = Mid.position; ld = nd;
        }
        if (!ok) continue;
        float ws[4] = {0, 0, 0, 0}, rate[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < depth && k < 4; k++) {
            const BDPTVertex& e = ev[depth - 1 - k];
            const BDPTVertex& l = lv[k];
            ws[k] = k == 0 ? rmis::connection_direction_lightSource(P, e, l) : rmis::general_connection(P, e, l);
            rate[k] = e.pdf * float3weight(connectRate_SOL(P, e.subspaceId, l.subspaceId, l.flux / l.pdf)) * l.pdf;
            if (ev_out && lv_out) { export_eye_vertex(e, ev_out[4 * (size_t)found + k]); export_vertex(l, lv_out[4 * (size_t)found + k]); }   // the pair of strategy k, for the device harness
        }
        rate[4] = ev[depth - 1].pdf * next_single_pdf;
        double rs = rate[4];
        for (int k = 0; k < depth && k < 4; k++) rs += rate[k];
        float* o = out + 7 * (size_t)found;
        o[0] = w_miss + ws[0] + ws[1] + ws[2] + ws[3]; o[1] = w_miss; o[2] = ws[0]; o[3] = ws[1]; o[4] = ws[2]; o[5] = ws[3]; o[6] = 1.0f;
        float* t = truth + 5 * (size_t)found;
        t[0] = (float)(rate[4] / rs);
        for (int k = 0; k < 4; k++) t[1 + k] = (float)(rate[k] / rs);
        found++;
    }
    return found;
}

// The same for QUAD emitters: camera paths c, x1 .. xD whose next segment HITS an emitter at z (front side).  Strategies: the
// emitter hit (rmis::light_hit through __closesthit__eyeSubpath_LightSource), e_D <-> y0 = z (connection_lightSource),
// e_{D-k} <-> y_k for k >= 1 (general_connection), the light sub-path re-traced from z with its directions forced.  out / truth as above
// ([1] / [0] = the emitter hit).  Rates in area measure over (x1 .. xD, z).
extern "C" int orc_debug_quad_partition(orc_ctx* c, int depth, int n, unsigned frame, float* out, float* truth, orc_eye_vertex* ev_out, spcbpt_light_vertex* lv_out) {
    Params P = c->P;
    P.counters = nullptr;
    const Scene& S = *P.scene;
    if (depth < 1 || depth > 4) return -1;
    int found = 0;
    for (unsigned pix = 0; found < n && pix < 40000000u; pix++) {
        const unsigned x = pix % P.width, y = (pix / P.width) % P.height;
        uint32_t seed = tea<4>(y * P.width + x, frame + pix / (P.width * P.height));
        float jx = rnd(seed), jy = rnd(seed);
        const float dx = 2.0f * (((float)x + jx) / (float)P.width) - 1.0f, dy = 2.0f * (((float)y + jy) / (float)P.height) - 1.0f;
        float3 dir = normalize(dx * P.U + dy * P.V + P.W), org = P.eye;
        PayloadBDPTVertex prd;
        prd.clear(); prd.seed = seed; prd.ray_direction = dir; prd.origin = org;
        init_EyeSubpath(prd.path, org, dir);
        std::vector<BDPTVertex> ev;
        bool ok = true;
        for (int d = 0; d < depth; d++) {
            const int before = prd.path.size;
            trace_subpath(P, prd.origin, prd.ray_direction, &prd, false);
            if (prd.path.size == before || prd.path.hit_lightSource() || prd.done) { ok = false; break; }
            ev.push_back(prd.path.currentVertex());
        }
        if (!ok) continue;
        {   // the next segment must hit an emitter on its front side
            const int before = prd.path.size;
            trace_subpath(P, prd.origin, prd.ray_direction, &prd, false);
            if (prd.path.size == before || prd.path.currentVertex().type != HIT_LIGHT_SOURCE) continue;
        }
        const BDPTVertex hit = prd.path.currentVertex();
        if (!(float3weight(hit.flux) > 0.0f)) continue;   // a path that scattered INTO a surface on its way (upstream lets it live with a flux of zero, DESIGN d11): it carries nothing, and no light sub-path can retrace it
        const float w_hit = 1.0f / hit.RMIS_pointer;
        const Light& light = S.lights[hit.materialId];
        lightSample ls;
        ls.ReverseSample(P, light, hit.uv);
        const float3 xD = ev[depth - 1].position;
        ls.direction = normalize(xD - ls.position);
        ls.dir_pdf = fabsf(dot(ls.direction, light.normal)) / M_PIf_;   // traceMode's pdf of this direction
        PayloadBDPTVertex pl;
        pl.clear();
        init_lightSubPath_from_lightSample(ls, pl.path);
        std::vector<BDPTVertex> lv;
        lv.push_back(pl.path.currentVertex());
        float3 lo = ls.position, ld = ls.direction;
        for (int k = 1; k <= depth && ok; k++) {
            const int before = pl.path.size;
            pl.done = false;
            trace_subpath(P, lo, ld, &pl, true);
            if (pl.path.size == before) { ok = false; break; }
            BDPTVertex& Mid = pl.path.currentVertex();
            if (length(Mid.position - ev[depth - k].position) > 1e-3f) { ok = false; break; }
            lv.push_back(Mid);
            if (k == depth) break;
            const float3 nd = normalize(ev[depth - k - 1].position - Mid.position);
            Pbr pbr = S.materials[Mid.materialId];
            pbr.base_color = Mid.color;
            BDPTVertex& Next = pl.path.nextVertex();
            Next.flux = Eval(pbr, Mid.normal, -ld, nd) / (pbr.brdf ? fabsf(dot(Mid.normal, nd)) : 1.0f);   // hit_program.cu:384
            Next.singlePdf = Pdf(pbr, Mid.normal, -ld, nd) * rr_rate_of(Mid.color);
            lo = Mid.position; ld = nd;
        }
        if (!ok) continue;
        float ws[4] = {0, 0, 0, 0}, rate[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < depth && k < 4; k++) {
            const BDPTVertex& e = ev[depth - 1 - k];
            const BDPTVertex& l = lv[k];
            ws[k] = k == 0 ? rmis::connection_lightSource(P, e, l) : rmis::general_connection(P, e, l);
            rate[k] = e.pdf * float3weight(connectRate_SOL(P, e.subspaceId, l.subspaceId, l.flux / l.pdf)) * l.pdf;
            if (ev_out && lv_out) { export_eye_vertex(e, ev_out[4 * (size_t)found + k]); export_vertex(l, lv_out[4 * (size_t)found + k]); }
        }
        rate[4] = hit.pdf;
        double rs = rate[4];
        for (int k = 0; k < depth && k < 4; k++) rs += rate[k];
        float* o = out + 7 * (size_t)found;
        o[0] = w_hit + ws[0] + ws[1] + ws[2] + ws[3]; o[1] = w_hit; o[2] = ws[0]; o[3] = ws[1]; o[4] = ws[2]; o[5] = ws[3]; o[6] = 1.0f;
        float* t = truth + 5 * (size_t)found;
        t[0] = (float)(rate[4] / rs);
        for (int k = 0; k < 4; k++) t[1 + k] = (float)(rate[k] / rs);
        found++;
    }
    return found;
}

// "SPCBPT_no_rmis": __raygen__SPCBPT_no_rmis (raygen.cu:445-606) with contriCompute / pdfCompute / MISWeight_SPCBPT
// (cuProg.h:901-1105) -- the subspace sampler with classic FULL-PATH MIS weights instead of the recursive ones.  The reference
// defines the raygen program but binds it to no program group (sutil/Scene.cpp:1642-1789 knows four names), so it cannot be
// launched there; it is built here because it is an independent estimator of the image "SPCBPT_eye" renders: its weights use none
// of rmis.h, so agreement of the two (tests/test_gpu_configs.py) checks the recursive weights against a second derivation.
// A validation mode, not a timed path: one pixel-sample per lane, the path (<= 20 vertices) in per-lane scratch memory, every
// strategy's pdf recomputed per connection (O(n^2)), double precision where the reference's literals promote.
#include <hip/hip_runtime.h>

#include "device_lib.h"
#include "eye_walk.h"
#include "kernels.h"

namespace spc {

static constexpr int FBLOCK = 256;
static constexpr int MAX_PATH = 20;   // MAX_PATH_LENGTH_FOR_MIS

struct PVertex {   // the BDPTVertex fields the full-path functions read
    f3 pos, n, color, flux;
    float pdf;
    int mat, sub, depth;
};

SPC_DEV f3 path_contri(const DeviceScene& S, const PVertex* path, int n) {  // contriCompute, cuProg.h:901-934
    const PVertex& light = path[n - 1];
    const f3 lightDirection = normalize(path[n - 2].pos - light.pos);
    const float lAng = dot(light.n, lightDirection);
    if (lAng < 0.0f) return mk3(0.0f);
    f3 throughput = mk3(1.0f) * (light.flux * lAng);
    for (int i = 1; i < n; i++) {
        const f3 line = path[i].pos - path[i - 1].pos;
        throughput = throughput / dot(line, line);
    }
    for (int i = 1; i < n - 1; i++) {
        const PVertex& mid = path[i];
        const f3 lastDirection = normalize(path[i - 1].pos - mid.pos), nextDirection = normalize(path[i + 1].pos - mid.pos);
        const Pbr mat = load_pbr_colored(S, mid.mat, mid.color);
        throughput = throughput * ((fabsf(dot(mid.n, lastDirection)) * fabsf(dot(mid.n, nextDirection))) * bsdf_eval(mat, mid.n, lastDirection, nextDirection));
    }
    return throughput;
}
SPC_DEV float eye_side_pdf(const DeviceScene& S, const PVertex* path, int eyePathLength) {  // cuProg.h:972-994 = 1004-1027
    float pdf = 1.0f;
    for (int i = 1; i < eyePathLength; i++) {
        const f3 line = path[i].pos - path[i - 1].pos;
        pdf *= 1.0f / dot(line, line) * fabsf(dot(path[i].n, normalize(line)));
    }
    for (int i = 1; i < eyePathLength - 1; i++) {
        const PVertex& mid = path[i];
        const f3 lastDirection = normalize(path[i - 1].pos - mid.pos), nextDirection = normalize(path[i + 1].pos - mid.pos);
        const Pbr mat = load_pbr_colored(S, mid.mat, mid.color);
        pdf *= bsdf_pdf(mat, mid.n, lastDirection, nextDirection) * max3(mid.color);   // rr_rate unclamped here, as written (cuProg.h:991)
    }
    return pdf;
}
SPC_DEV float path_pdf(const DeviceScene& S, const PVertex* path, int n, int strategy_id) {  // pdfCompute, cuProg.h:935-996
    const int eyePathLength = strategy_id, lightPathLength = n - eyePathLength;
    float pdf = 1.0f;
    if (lightPathLength > 0) pdf *= path[n - 1].pdf;
    if (lightPathLength > 1) {
        const PVertex& light = path[n - 1];
        const f3 lightDirection = normalize(path[n - 2].pos - light.pos);
        pdf = (float)((double)pdf * ((double)fabsf(dot(lightDirection, light.n)) / 3.14159265358979323846));   // `/ M_PI`: double
        for (int i = 1; i < lightPathLength; i++) {
            const PVertex& mid = path[n - i - 1];
            const f3 line = mid.pos - path[n - i].pos;
            pdf = (float)((double)pdf * (1.0 / (double)dot(line, line) * (double)fabsf(dot(mid.n, normalize(line)))));   // `1.0 / ...`: double
        }
        for (int i = 1; i < lightPathLength - 1; i++) {
            const PVertex& mid = path[n - i - 1];
            const f3 lastDirection = normalize(path[n - i].pos - mid.pos), nextDirection = normalize(path[n - i - 2].pos - mid.pos);
            const Pbr mat = load_pbr_colored(S, mid.mat, mid.color);
            pdf *= bsdf_pdf(mat, mid.n, lastDirection, nextDirection) * max3(mid.color);
        }
    }
    return pdf * eye_side_pdf(S, path, eyePathLength);
}
SPC_DEV float mis_weight(const KParams& p, const PVertex* path, int n, int strategy_id) {  // MISWeight_SPCBPT, cuProg.h:998-1105
    const DeviceScene& S = p.scene;
    if (strategy_id <= 1 || strategy_id == n) return path_pdf(S, path, n, strategy_id);
    const int eyePathLength = strategy_id, lightPathLength = n - eyePathLength;
    const float pdf = eye_side_pdf(S, path, eyePathLength);
    f3 light_contri = mk3(1.0f);
    if (lightPathLength > 0) light_contri = light_contri * path[n - 1].flux;
    if (lightPathLength > 1) {
        const PVertex& lastMid = path[n - 2];
        for (int i = 1; i < lightPathLength; i++) {
            const PVertex& mid = path[n - i - 1];
            const f3 line = mid.pos - path[n - i].pos, dir = normalize(line);
            const double g = 1.0 / (double)dot(line, line) * (double)fabsf(dot(mid.n, dir)) * (double)fabsf(dot(lastMid.n, dir));   // lastMidPoint, as written
            light_contri = light_contri * (float)g;
        }
        for (int i = 1; i < lightPathLength - 1; i++) {
            const PVertex& mid = path[n - i - 1];
            const f3 lastDirection = normalize(path[n - i].pos - mid.pos), nextDirection = normalize(path[n - i - 2].pos - mid.pos);
            const Pbr mat = load_pbr_colored(S, mid.mat, mid.color);
            light_contri = light_contri * bsdf_eval(mat, mid.n, lastDirection, nextDirection);
        }
    }
    Counts<false> cn;
    const PVertex& e = path[strategy_id - 1];
    const int eye_sub = tree_label(p.eye_tree, e.pos, e.n, normalize(path[strategy_id - 2].pos - e.pos), cn);
    int light_sub;
    if (strategy_id == n - 1) light_sub = path[strategy_id].sub;
    else {
        const PVertex& l = path[strategy_id];
        light_sub = tree_label(p.light_tree, l.pos, l.n, normalize(path[strategy_id + 1].pos - l.pos), cn);
    }
    return pdf * sum3(gamma_ss(p, eye_sub, light_sub, cn) * light_contri * (float)SPCBPT_CONNECTION_N);   // float3weight(connectRate_SOL(..))
}
SPC_DEV f3 eval_path(const KParams& p, const PVertex* path, int n, int strategy_id) {  // raygen.cu:445-464
    const float pdf = path_pdf(p.scene, path, n, strategy_id);
    const f3 contri = path_contri(p.scene, path, n);
    const float w = mis_weight(p, path, n, strategy_id);
    float denom = 0.0f;
    for (int i = 2; i <= n; i++) denom += mis_weight(p, path, n, i);
    const f3 ans = contri / pdf * (w / denom);
    return is_invalid(ans) ? mk3(0.0f) : ans;
}
SPC_DEV PVertex pv_of(const EyeVertex& v) {
    PVertex q;
    q.pos = v.c.pos; q.n = v.c.n; q.color = v.c.color; q.flux = v.flux; q.pdf = v.pdf; q.mat = v.c.mat; q.sub = v.sub; q.depth = v.depth;
    return q;
}
SPC_DEV PVertex pv_of(const LightVertex& b) {
    PVertex q;
    q.pos = ld3(b.position); q.n = ld3(b.normal); q.color = ld3(b.color); q.flux = ld3(b.flux); q.pdf = b.pdf; q.mat = b.material_id;
    q.sub = b.subspace_id; q.depth = b.depth;
    return q;
}

__global__ __launch_bounds__(FBLOCK) void k_spcbpt_no_rmis(const KParams p) {
    __shared__ uint32_t s_stack[FBLOCK * kStackLds];
    uint32_t x, y;
    if (!lane_pixel(p, x, y)) return;
    const DeviceScene& S = p.scene;
    Counts<false> cn;
    TravStack<FBLOCK, kStackLds> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)blockIdx.x * FBLOCK + threadIdx.x, p.diag);
    WalkState w;
    w.dir = camera_ray(p, x, y, w.seed);
    w.origin = ld3(p.eye);
    w.done = false; w.next_flux = mk3(0.0f); w.next_single_pdf = 1.0f;
    EyeVertex cur;   // init_EyeSubpath
    cur.c.pos = w.origin; cur.c.n = w.dir; cur.c.color = mk3(0.0f); cur.c.lastPos = w.origin; cur.c.lnp = 0.0f; cur.c.mat = 0; cur.c.lld = false;
    cur.flux = mk3(1.0f); cur.R3 = mk3(0.0f); cur.pdf = 1.0f; cur.singlePdf = 1.0f; cur.sub = 0; cur.lastZone = 0; cur.depth = 0; cur.lsub = 0;
    PVertex path[MAX_PATH];
    int size = 0;
    path[size++] = pv_of(cur);
    f3 result = mk3(0.0f);
    int depth = 0;
    const int path_count = p.sampler_counts[1];
    while (true) {
        if (w.done || depth > 50) break;
        HitRec h;
        const f3 ray_dir = w.dir;
        if (!traverse<false, false>(S, st, w.origin, w.dir, kEps, 1e16f, h, cn)) break;   // __miss__BDPTVertex
        const Geom g = local_geometry(S, h);
        depth += 1;
        if (g.emitter) {   // __closesthit__eyeSubpath_LightSource: back side -> no vertex; else the path ends on the emitter
            const DLight& L = S.lights[load_pbr(S, g.mat).light_id];
            if (dot(ray_dir, ld3(L.normal)) > 0) break;
            const LightSampleD ls = light_reverse_sample(S, L, g.u, g.v);   // init_vertex_from_lightSample of ReverseSample(hit uv)
            PVertex lv;
            lv.pos = ls.position; lv.n = ls.normal; lv.color = mk3(0.0f); lv.flux = ls.emission; lv.pdf = ls.pdf; lv.mat = L.id; lv.sub = ls.subspace; lv.depth = 0;
            path[size++] = lv;
            result += eval_path(p, path, size, size);
            break;
        }
        EyeVertex mid;
        eye_surface_hit<false, false>(p, g, h.t, ray_dir, cur.depth == 0, cur, w, mid, cn);
        cur = mid;
        path[size++] = pv_of(cur);
        if (size >= MAX_PATH) break;
        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
            float pmf1 = 1.0f, pmf2;
            const int l = sample_first_stage(p, cur.sub, w.seed, pmf1, cn);
            const DSubspace ss = p.subspace[l];
            if (ss.size == 0) continue;
            const int k = binary_sample(p.cmfs + ss.jump_bias, ss.size, w.seed, pmf2, cn);
            const int lslot = p.jump[ss.jump_bias + k];
            const LightVertex b = p.lvc[lslot];
            if (size + b.depth + 1 > MAX_PATH) continue;
            const f3 bias = ld3(b.position) - cur.c.pos;
            const float len = sqrtf(dot(bias, bias));
            HitRec sh;
            if (traverse<true, false>(S, st, cur.c.pos, bias / len, kEps, len - kEps, sh, cn)) continue;   // occluded
            const float pmf = (float)path_count * pmf2 * pmf1;
            const int origin_size = size;
            for (int j = 0; j <= b.depth; j++) path[size++] = pv_of(p.lvc[lslot - j]);   // the light sub-path sits in consecutive slots, origin first
            f3 res = eval_path(p, path, size, origin_size) / pmf;
            size = origin_size;
            if (!is_invalid(res)) result += res / (float)SPCBPT_CONNECTION_N;
        }
    }
    film_write(p, x, y, result);
}

void launch_spcbpt_no_rmis(const KParams& p, hipStream_t s) {
    const int tiles_x = ((int)p.width + 7) / 8;
    const int band_begin = p.row_begin / 8, band_end = (std::min(p.row_end, (int)p.height) + 7) / 8;
    const int step = p.row_step < 1 ? 1 : p.row_step;
    const int nb = band_end > band_begin ? (band_end - band_begin + step - 1) / step : 0;
    const int blocks = (tiles_x * nb + 3) / 4;
    if (blocks <= 0) return;
    hipLaunchKernelGGL(k_spcbpt_no_rmis, dim3(blocks), dim3(FBLOCK), 0, s, p);
}

}  // namespace spc

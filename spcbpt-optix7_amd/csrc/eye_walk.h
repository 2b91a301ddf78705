// Eye-walk steps shared by the megakernel, the per-function harness and the pretrace kernel.
#pragma once
#include "device_lib.h"

namespace spc {

// ------------------------------------------------------------------------------------------------
// Eye-walk step shared by k_spcbpt and the pretrace kernel: build the vertex at a surface hit
// (hit_program.cu:246-340).  `last` is the previous vertex (camera when last.depth == 0).
struct WalkState {
    f3 origin, dir;        // next ray
    f3 next_flux;          // NextVertex.flux  = BSDF value of the sampled direction
    float next_single_pdf; // NextVertex.singlePdf = solid-angle pdf (x RR once survived)
    uint32_t seed;
    bool done;
};

template <bool COUNT, bool CACHE = false, bool ENV = true>   // ENV = false: DeviceScene::general == 0 (brdf_div compiled out)
SPC_DEV void eye_surface_hit(const KParams& p, const Geom& g, float t_hit, f3 ray_dir, bool last_is_origin, const EyeVertex& last,
                             WalkState& w, EyeVertex& mid, Counts<COUNT>& cn, bool stop_dead_paths = false) {
    const DeviceScene& S = p.scene;
    Pbr pbr = load_pbr(S, g.mat);
    color_tex_sample(S, g, pbr, cn);
    f3 N = g.N;
    if (dot(N, ray_dir) > 0.f) N = -N;
    const f3 inv_dir = -ray_dir;
    const f3 new_dir = bsdf_sample(pbr, N, inv_dir, w.seed);
    const float pdf = bsdf_pdf(pbr, N, inv_dir, new_dir);
    if (!(pdf > 0.0f)) w.done = true;

    mid.c.pos = g.P;
    mid.c.n = N;
    const float pdf_G = fabsf(dot(N, ray_dir) * dot(last.c.n, ray_dir)) / (t_hit * t_hit);
    mid.flux = last_is_origin ? last.flux * pdf_G : w.next_flux * last.flux * pdf_G;
    mid.c.lastPos = last.c.pos;
    mid.c.color = pbr.base;
    mid.c.lnp = fabsf(dot(last.c.n, ray_dir));
    mid.c.lld = false;
    mid.c.mat = g.mat;
    // eye-tree label of the new vertex and light-tree relabel of the previous one (tracing_weight_eye, depth >= 3) together
    int light_label;
    mid.lsub = 0;
    if (CACHE) {   // both labels of the NEW vertex (device_lib.h: label caching); the previous vertex brings its own
        tree_label2<COUNT, true>(p.eye_tree, g.P, N, inv_dir, true, p.light_tree, g.P, N, inv_dir, last.depth + 1 != 1, mid.sub, mid.lsub, cn);
        light_label = last.lsub;
    } else {
        tree_label2(p.eye_tree, g.P, N, inv_dir, true, p.light_tree, last.c.pos, last.c.n, normalize(g.P - last.c.pos),
                    last.depth + 1 != 1 && last.depth != 1, mid.sub, light_label, cn);
    }
    mid.lastZone = last.sub;
    mid.depth = last.depth + 1;
    mid.singlePdf = w.next_single_pdf * pdf_G / fabsf(dot(last.c.n, ray_dir));
    mid.pdf = last.pdf * mid.singlePdf;
    // recursive MIS (rmis.h:189-207)
    if (mid.depth == 1) {
        mid.R3 = mk3(0.0f);
    } else {
        const Pbr mat_last = load_pbr_colored(S, last.c.mat, last.c.color);
        const f3 in_dir = normalize(mid.c.pos - last.c.pos);
        const float LL_pdf = rmis_last_pdf(mat_last, last.c, in_dir);
        const float wgt = rmis_weight_eye_l(p, last.depth, last.lastZone, light_label, cn);
        const f3 fm = rmis_flux_multiplier<ENV>(mat_last, last.c, in_dir, normalize(last.c.lastPos - last.c.pos));
        mid.R3 = (last.R3 * LL_pdf * fm + mk3(wgt)) / last.singlePdf;
    }
    cn.add(C_VERTEX);
    // next segment + Russian roulette (the vertex itself is kept; hit_program.cu:324-337)
    w.next_flux = brdf_div<ENV>(pbr, bsdf_eval(pbr, N, inv_dir, new_dir), N, new_dir);   // hit_program.cu:286
    // a sampled direction whose BSDF value is exactly zero (it points below the surface) gives every later
    // vertex a flux of exactly zero -- emitter hits and connections of the rest of the path contribute 0 (or NaN -> rejected).
    // The render kernels end the path after this vertex's connections; the image is unchanged.
    if (stop_dead_paths && w.next_flux.x == 0.0f && w.next_flux.y == 0.0f && w.next_flux.z == 0.0f) w.done = true;
    w.next_single_pdf = pdf;
    w.origin = g.P;
    w.dir = new_dir;
    const float r = rnd(w.seed);
    const float rr = rr_of(mid.c.color);
    if (r > rr) w.done = true;
    else w.next_single_pdf *= rr;
}

// __closesthit__eyeSubpath_LightSource + rmis::light_hit + lightStraghtHit (hit_program.cu:62-147, rmis.h:359-389,
// raygen.cu:305-317): contribution of an eye path that runs into an emitter.
template <bool COUNT, bool CACHE = false, bool ENV = true>
SPC_DEV f3 eye_emitter_hit(const KParams& p, const Geom& g, float t_hit, f3 ray_dir, bool last_is_origin, const EyeVertex& last,
                           const WalkState& w, Counts<COUNT>& cn) {
    const DeviceScene& S = p.scene;
    const int light_id = load_pbr(S, g.mat).light_id;
    const DLight& L = S.lights[light_id];
    const f3 ln = ld3(L.normal);
    if (dot(ray_dir, ln) > 0) return mk3(0.0f);
    const LightSampleD ls = light_reverse_sample(S, L, g.u, g.v);
    const float pdf_G = fabsf(dot(ln, ray_dir) * dot(last.c.n, ray_dir)) / (t_hit * t_hit);
    const f3 flux = last_is_origin ? last.flux * pdf_G * ls.emission : w.next_flux * last.flux * pdf_G * ls.emission;
    const float singlePdf = w.next_single_pdf * pdf_G / fabsf(dot(last.c.n, ray_dir));
    const float pdf = last.pdf * singlePdf;
    float rmis_pointer = 1.0f;
    if (last.depth + 1 != 1) {
        // light_hit(eye = last, light = virtual vertex at the hit point)
        const f3 lpos = g.P;
        const f3 connect_dir = normalize(last.c.pos - lpos);
        const f3 lflux = ls.emission / ls.pdf;
        const Pbr mat_e = load_pbr_colored(S, last.c.mat, last.c.color);
        const f3 LB = normalize(last.c.lastPos - last.c.pos);
        const float LL_pdf_A = rmis_last_pdf(mat_e, last.c, -connect_dir);
        const f3 fm0 = rmis_flux_multiplier<ENV>(mat_e, last.c, -connect_dir, LB);
        const float wA = CACHE ? rmis_weight_eye_l(p, last.depth, last.lastZone, last.lsub, cn) : rmis_weight_eye(p, last.c, last.depth, last.lastZone, lpos, cn);
        const f3 D_A_0 = last.R3 * LL_pdf_A * fm0 + mk3(wA);
        const float pdf_A = rmis_pdf_from_light(lpos, ln, last.c.pos, last.c.n);
        const float D_A = sum3(D_A_0 * pdf_A * kPi * lflux / last.singlePdf);
        const float weight = sum3(gamma_ss(p, last.sub, ls.subspace, cn) * lflux * (float)SPCBPT_CONNECTION_N);
        const float D_B = 1.0f;
        const float pdf_B = rmis_get_pdf(mat_e, last.c, lpos, ln, LB);
        const float lh = D_B / ((weight + D_A) / pdf_B * ls.pdf + D_B);
        rmis_pointer = 1.0f / lh;
    }
    const f3 ans = flux / pdf / rmis_pointer;
    return is_invalid(ans) ? mk3(0.0f) : ans;
}

// ------------------------------------------------------------------------------------------------
// Pixel of work-slot `slot` (0..63) of tile `tile`: tiles are 8x8 pixels, enumerated x-major inside the selected bands.
SPC_DEV bool tile_pixel(const KParams& p, uint32_t tile, uint32_t slot, uint32_t& x, uint32_t& y) {
    const uint32_t tiles_x = (p.width + 7) / 8;
    const uint32_t tile_x = tile % tiles_x, band_k = tile / tiles_x;
    const uint32_t band = (uint32_t)(p.row_begin / 8) + band_k * (uint32_t)p.row_step;
    x = tile_x * 8 + (slot & 7);
    y = band * 8 + (slot >> 3);
    return x < p.width && y < p.height && (int)y < p.row_end;
}

// pixel of this lane in the plain (non-persistent) launches: 8x8 tile per wave, 4 tiles (in x) per 256-thread block, bands of 8
// rows selected by (row_begin, row_step)
SPC_DEV bool lane_pixel(const KParams& p, uint32_t& x, uint32_t& y) {
    const uint32_t tiles_x = (p.width + 7) / 8;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t tile_x = wave % tiles_x, band_k = wave / tiles_x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t band = (uint32_t)(p.row_begin / 8) + band_k * (uint32_t)p.row_step;
    x = tile_x * 8 + (lane & 7);
    y = band * 8 + (lane >> 3);
    return x < p.width && y < p.height && (int)y < p.row_end && (int)y >= p.row_begin;
}

}  // namespace spc

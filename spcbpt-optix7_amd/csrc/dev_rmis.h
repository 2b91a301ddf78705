// Part of device_lib.h (split in round 6 for readability; included by it, in this order, inside the one translation unit of each
// .hip file -- the device code generated is the same as from the single header: tests/test_codegen_guard.py):
// path vertices, connection evaluation and recursive MIS (rmis.h, raygen.cu:236-317).
#pragma once
#include "device_lib.h"

namespace spc {

// ---- recursive MIS (rmis.h) ------------------------------------------------------
// The fields of a path vertex the RMIS recursions read, shared by eye and light vertices.
struct VCore {
    f3 pos, n, color, lastPos;
    float lnp;  // lastNormalProjection
    int mat;
    bool lld;   // is_LL_DIRECTION (BDPTVertex.h:67): the vertex was hit straight from the environment map (light vertices only)
};
SPC_DEV float rr_of(f3 color) { return fmaxf(max3(color), SPCBPT_MIN_RR_RATE); }  // getRR rmis.h:28-40 (q10)

// getLast_pdf (rmis.h:41-51): pdf of stepping from v back to its predecessor given arrival from in_dir
SPC_DEV float rmis_last_pdf(const Pbr& mat, const VCore& v, f3 in_dir) {
    const f3 out_vec = v.lastPos - v.pos;
    const f3 out_dir = normalize(out_vec);
    // rmis.h:45-47: the step back from a vertex lit straight by the sky leads to a direction, not to a point: no area measure
    float pdf = v.lld ? bsdf_pdf(mat, v.n, in_dir, out_dir) : bsdf_pdf(mat, v.n, in_dir, out_dir) / dot(out_vec, out_vec) * v.lnp;
    return pdf * rr_of(v.color);
}
// getFluxMultiplier (rmis.h:102-118)
template <bool ENV = true>
SPC_DEV f3 rmis_flux_multiplier(const Pbr& mat, const VCore& v, f3 in_dir, f3 out_dir) {
    const f3 flux_ratio = brdf_div<ENV>(mat, bsdf_eval(mat, v.n, in_dir, out_dir), v.n, out_dir);   // rmis.h:105
    const float pdf_ratio = bsdf_pdf(mat, v.n, in_dir, out_dir);
    const float rr = rr_of(v.color);
    const float cos_theta = fabsf(dot(v.n, out_dir));
    return flux_ratio * cos_theta / pdf_ratio / rr;
}
// getPdf (rmis.h:153-172): pdf of generating `end` from `begin` given arrival from in_dir
SPC_DEV float rmis_get_pdf(const Pbr& mat, const VCore& begin, f3 end_pos, f3 end_n, f3 in_dir) {
    const f3 out_vec = end_pos - begin.pos;
    const f3 out_dir = normalize(out_vec);
    float pdf = bsdf_pdf(mat, begin.n, in_dir, out_dir) / dot(out_vec, out_vec) * fabsf(dot(out_dir, end_n));
    return pdf * rr_of(begin.color);
}
// getPdf_from_light_source (rmis.h:173-188)
SPC_DEV float rmis_pdf_from_light(f3 light_pos, f3 light_n, f3 end_pos, f3 end_n) {
    const f3 conn_vec = end_pos - light_pos;
    const f3 conn_dir = normalize(conn_vec);
    const float pdf_angle = fabsf(dot(light_n, conn_dir)) * kInvPi;
    const float angle2a = fabsf(dot(end_n, conn_dir)) / dot(conn_vec, conn_vec);
    return pdf_angle * angle2a;
}

// Eye-side vertex kept in registers while walking (the live BDPTVertex fields of the eye sub-path)
struct EyeVertex {
    VCore c;
    f3 flux, R3;       // flux, RMIS_pointer_3
    float pdf, singlePdf;
    int sub, lastZone, depth;
    int lsub;          // label caching (see label_cache below): the vertex's own light-tree label
};

// tracing_weight_eye (rmis.h:131-151) with Last = `last`, Mid at `mid_pos`
template <bool COUNT>
SPC_DEV float rmis_weight_eye(const KParams& p, const VCore& last, int last_depth, int last_lastZone, f3 mid_pos, Counts<COUNT>& cn) {
    if (last_depth == 1) return 0.0f;
    const f3 inver_dir = normalize(mid_pos - last.pos);
    const int light_label = tree_label(p.light_tree, last.pos, last.n, inver_dir, cn);
    return gamma_ss(p, last_lastZone, light_label, cn) * (float)SPCBPT_CONNECTION_N;
}
// the same two weights with the relabel already done (tree_label2)
template <bool COUNT>
SPC_DEV float rmis_weight_eye_l(const KParams& p, int last_depth, int last_lastZone, int light_label, Counts<COUNT>& cn) {
    if (last_depth == 1) return 0.0f;
    return gamma_ss(p, last_lastZone, light_label, cn) * (float)SPCBPT_CONNECTION_N;
}
template <bool COUNT>
SPC_DEV float rmis_weight_light_l(const KParams& p, int last_lastZone, float last_lum, int eye_label, Counts<COUNT>& cn) {
    return gamma_ss(p, eye_label, last_lastZone, cn) * last_lum * (float)SPCBPT_CONNECTION_N;
}
// tracing_weight_light (rmis.h:58-79) with Last = light vertex `last`
template <bool COUNT>
SPC_DEV float rmis_weight_light(const KParams& p, const VCore& last, int last_lastZone, float last_lum, f3 mid_pos, Counts<COUNT>& cn) {
    const f3 inver_dir = normalize(mid_pos - last.pos);
    const int eye_label = tree_label(p.eye_tree, last.pos, last.n, inver_dir, cn);
    return gamma_ss(p, eye_label, last_lastZone, cn) * last_lum * (float)SPCBPT_CONNECTION_N;
}

template <bool ENV = true>
SPC_DEV VCore core_of(const LightVertex& b) {
    VCore c;
    c.pos = ld3(b.position); c.n = ld3(b.normal); c.color = ld3(b.color); c.lastPos = ld3(b.last_position);
    c.lnp = b.last_normal_projection; c.mat = b.material_id;
    c.lld = ENV && (b.pad & SPCBPT_LV_LAST_DIRECTION) != 0u;
    return c;
}

// A connection whose value is exactly zero whatever the visibility: the eye vertex sees the light vertex from
// behind its own surface (bsdf_eval returns 0 for N.V <= 0), or the light vertex faces away (N.L <= 0 on a surface vertex,
// the one-sided term on an emitter vertex).  Same vectors and the same normalize() as connect_vertices.
// ... and for a direction of the environment map: direction_connect_ZGCBPT contributes only with the sky above the eye vertex's surface
SPC_DEV bool null_connection_direction(f3 an, f3 bn) { return !(dot(an, -bn) > 0.0f); }
SPC_DEV bool null_connection(f3 apos, f3 an, f3 bpos, f3 bn) {
    const f3 connectDir = normalize(apos - bpos);
    return dot(an, -connectDir) <= 0.0f || dot(bn, connectDir) < 0.0f;
}

// direction_connect_ZGCBPT (raygen.cu:234-252) with rmis::connection_direction_lightSource (rmis.h:249-280): the light vertex is a
// direction of the environment map (type ENV: normal = minus the sky direction, position = its point on the sky disk).
template <bool COUNT, bool CACHE>
SPC_DEV f3 connect_direction(const KParams& p, const EyeVertex& a, const LightVertex& b, Counts<COUNT>& cn, float* w_out) {
    const DeviceScene& S = p.scene;
    const f3 bn = ld3(b.normal), bflux = ld3(b.flux);
    const f3 connectDir = -bn;
    if (w_out) *w_out = 0.0f;
    if (!(dot(a.c.n, connectDir) > 0.0f)) return mk3(0.0f);
    const f3 LA_DIR = normalize(a.c.lastPos - a.c.pos);
    const Pbr mat_a = load_pbr_colored(S, a.c.mat, a.c.color);
    const f3 f = bsdf_eval(mat_a, a.c.n, LA_DIR, connectDir) * dot(a.c.n, connectDir);
    const f3 lflux = bflux / b.pdf;
    // getLL_pdf(light, eye): the incoming direction runs from the eye vertex to the light vertex's POSITION on the sky disk (as written)
    const float LL_pdf_A = rmis_last_pdf(mat_a, a.c, normalize(ld3(b.position) - a.c.pos));
    const f3 fm0 = rmis_flux_multiplier(mat_a, a.c, -bn, LA_DIR);                       // getFluxMultiplier(eye, -connect_dir), connect_dir = light.normal
    int light_label = a.lsub;
    if (!CACHE && a.depth != 1) light_label = tree_label(p.light_tree, a.c.pos, a.c.n, -bn, cn);   // tracing_weight_eye: inver_dir = -Mid.normal for a direction (rmis.h:141)
    const float wA = rmis_weight_eye_l(p, a.depth, a.lastZone, light_label, cn);
    const f3 D_A_0 = a.R3 * LL_pdf_A * fm0 + mk3(wA);
    const float pdf_A = S.env.project_pdf * fabsf(dot(bn, a.c.n));                     // getPdf_from_light_source, direction branch (183-187)
    const float fm1 = (float)(1.0 / S.env.project_pdf);
    const float D_A = sum3(D_A_0 * pdf_A * fm1 * lflux / a.singlePdf);
    const float weight = sum3(gamma_ss(p, a.sub, b.subspace_id, cn) * lflux * (float)SPCBPT_CONNECTION_N);
    const float pdf_B = bsdf_pdf(mat_a, a.c.n, LA_DIR, -bn) * rr_of(a.c.color);        // getPdf(eye, light, LB), end is a direction (158-162)
    const float D_B = b.rmis_pointer * pdf_B / b.single_pdf;
    const float w_rmis = weight / (weight + D_A + D_B);
    if (w_out) *w_out = w_rmis;
    return a.flux / a.pdf * f * bflux / b.pdf * w_rmis;
}

// connectVertex_SPCBPT (raygen.cu:253-303) with rmis::general_connection / connection_lightSource
// (rmis.h:212-247 / 281-313) fused: every BSDF lobe is fetched once.
// ENV = false: the scene has no environment map -- no vertex carries a direction flag, and the two tests fold away (the timed
// kernels of a scene without a sky are instantiated so: 3 % of the frame) -- and no `brdf`-flagged material (brdf_div)
template <bool COUNT, bool CACHE = false, bool ENV = true>
SPC_DEV f3 connect_vertices(const KParams& p, const EyeVertex& a, const LightVertex& b, Counts<COUNT>& cn, float* w_out = nullptr) {
    if (ENV && (b.pad & SPCBPT_LV_DIRECTION)) return connect_direction<COUNT, CACHE>(p, a, b, cn, w_out);   // raygen.cu:255-258
    const DeviceScene& S = p.scene;
    const f3 bpos = ld3(b.position), bn = ld3(b.normal), bflux = ld3(b.flux);
    const f3 connectVec = a.c.pos - bpos;
    const f3 connectDir = normalize(connectVec);
    const float r2 = dot(connectVec, connectVec);
    const float G = fabsf(dot(a.c.n, connectDir)) * fabsf(dot(bn, connectDir)) / r2;
    const f3 LA_DIR = normalize(a.c.lastPos - a.c.pos);
    const Pbr mat_a = load_pbr_colored(S, a.c.mat, a.c.color);
    const f3 fa = brdf_div<ENV>(mat_a, bsdf_eval(mat_a, a.c.n, -connectDir, LA_DIR), a.c.n, connectDir);   // raygen.cu:271
    const f3 lflux = bflux / b.pdf;  // `flux` of the rmis functions

    // ---- eye side terms shared by both connection kinds
    const float LL_pdf_A = rmis_last_pdf(mat_a, a.c, -connectDir);                 // getLL_pdf(light, eye)
    const f3 fm0 = rmis_flux_multiplier<ENV>(mat_a, a.c, -connectDir, LA_DIR);      // getFluxMultiplier(eye, -connect_dir)
    // the two relabels of the connection (light-tree label of the eye vertex seen from b, eye-tree label of the light vertex
    // seen from a) in one lock-step descent; the first is skipped at depth 1, the second for an emitter vertex, as in rmis.h
    int light_label, eye_label;
    if (CACHE) {
        light_label = a.lsub;                    // unused at depth 1, like the descent it replaces
        eye_label = (int)(b.pad & 0xffffu) - 1;  // unused for an emitter vertex
        // imported cache without labels (0), or a word that is not a label at all (spcbpt.h: never used as a row index unchecked)
        if ((b.pad & 0xffffu) - 1u >= (uint32_t)SPCBPT_NUM_SUBSPACE && b.depth != 0) eye_label = tree_label(p.eye_tree, bpos, bn, normalize(a.c.pos - bpos), cn);
    } else {
        tree_label2(p.light_tree, a.c.pos, a.c.n, normalize(bpos - a.c.pos), a.depth != 1,
                    p.eye_tree, bpos, bn, normalize(a.c.pos - bpos), b.depth != 0, light_label, eye_label, cn);
    }
    const float wA = rmis_weight_eye_l(p, a.depth, a.lastZone, light_label, cn);    // tracing_weight_eye(light, eye)
    const f3 D_A_0 = a.R3 * LL_pdf_A * fm0 + mk3(wA);
    const float weight = sum3(gamma_ss(p, a.sub, b.subspace_id, cn) * lflux * (float)SPCBPT_CONNECTION_N);
    const float pdf_B = rmis_get_pdf(mat_a, a.c, bpos, bn, LA_DIR);                 // getPdf(eye, light, LB)

    f3 fb;
    float D_A, D_B;
    if (b.depth == 0) {  // connection_lightSource
        fb = dot(bn, -connectDir) > 0.0f ? mk3(0.0f) : mk3(1.0f);
        const float pdf_A = rmis_pdf_from_light(bpos, bn, a.c.pos, a.c.n);
        D_A = sum3(D_A_0 * pdf_A * kPi * lflux / a.singlePdf);
        D_B = b.rmis_pointer * pdf_B / b.single_pdf;
    } else {  // general_connection
        const VCore bc = core_of<ENV>(b);
        const Pbr mat_b = load_pbr_colored(S, bc.mat, bc.color);
        const f3 LB_DIR = normalize(bc.lastPos - bc.pos);
        fb = brdf_div<ENV>(mat_b, bsdf_eval(mat_b, bn, connectDir, LB_DIR), bn, connectDir);   // raygen.cu:278
        const float pdf_A = rmis_get_pdf(mat_b, bc, a.c.pos, a.c.n, LB_DIR);        // getPdf(light, eye, LA)
        const f3 fm1 = rmis_flux_multiplier<ENV>(mat_b, bc, LB_DIR, connectDir);
        D_A = sum3(D_A_0 * pdf_A * fm1 * lflux / a.singlePdf);
        const float LL_pdf_B = rmis_last_pdf(mat_b, bc, connectDir);               // getLL_pdf(eye, light)
        const float wB = rmis_weight_light_l(p, b.last_zone_id, b.last_lum, eye_label, cn);
        D_B = (b.rmis_pointer * LL_pdf_B + wB) * pdf_B / b.single_pdf;
    }
    const float w_rmis = weight / (weight + D_A + D_B);
    if (w_out) *w_out = w_rmis;   // per-function harness only (unit.hip); the render kernels pass nothing
    const f3 contri = a.flux * bflux * fa * fb * G;
    const f3 ans = contri / (a.pdf * b.pdf) * w_rmis;
    return ans;
}

SPC_DEV bool is_invalid(f3 a) {  // ISINVALIDVALUE raygen.cu:43
    return a.x > 100000.0f || isnan(a.x) || a.y > 100000.0f || isnan(a.y) || a.z > 100000.0f || isnan(a.z);
}

}  // namespace spc

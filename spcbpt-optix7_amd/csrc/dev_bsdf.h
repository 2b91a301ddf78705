// Part of device_lib.h (split in round 6 for readability; included by it, in this order, inside the one translation unit of each
// .hip file -- the device code generated is the same as from the single header: tests/test_codegen_guard.py):
// materials, textures, hit geometry and the Disney BSDF (cuProg.h:686-899).
#pragma once
#include "device_lib.h"

namespace spc {

// ---- materials / textures / hit geometry -------------------------------------
struct Pbr {
    f3 base;
    float metallic, roughness, specular, specularTint, subsurface, sheen, sheenTint, clearcoat, clearcoatGloss;
    int albedo_tex, light_id;
    int brdf;   // MaterialData::Pbr::brdf: see brdf_div
};
SPC_DEV Pbr load_pbr(const DeviceScene& S, int id) {
    const float4* p = reinterpret_cast<const float4*>(S.mats + id);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    Pbr m;
    m.base = mk3(a.x, a.y, a.z); m.metallic = a.w;
    m.roughness = b.x; m.specular = b.y; m.specularTint = b.z; m.subsurface = b.w;
    m.sheen = c.x; m.sheenTint = c.y; m.clearcoat = c.z; m.clearcoatGloss = c.w;
    m.albedo_tex = __float_as_int(d.x); m.light_id = __float_as_int(d.y); m.brdf = __float_as_int(d.z);
    return m;
}
SPC_DEV Pbr load_pbr_colored(const DeviceScene& S, int id, f3 color) {  // rmis::getMat (rmis.h:16-21)
    Pbr m = load_pbr(S, id);
    m.base = color;
    return m;
}
// bilinear + wrap RGBA8 fetch (cudaReadModeNormalizedFloat semantics; exact-fraction weights)
SPC_DEV f3 tex_fetch_rgb(const DTexture& T, float u, float v) {
    const float x = u * (float)T.width - 0.5f, y = v * (float)T.height - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    int x0 = (int)fx % T.width, y0 = (int)fy % T.height;
    if (x0 < 0) x0 += T.width;
    if (y0 < 0) y0 += T.height;
    int x1 = x0 + 1 == T.width ? 0 : x0 + 1, y1 = y0 + 1 == T.height ? 0 : y0 + 1;
    const uint32_t t00 = T.rgba[(size_t)y0 * T.width + x0], t10 = T.rgba[(size_t)y0 * T.width + x1];
    const uint32_t t01 = T.rgba[(size_t)y1 * T.width + x0], t11 = T.rgba[(size_t)y1 * T.width + x1];
    const float w00 = (1 - ax) * (1 - ay), w10 = ax * (1 - ay), w01 = (1 - ax) * ay, w11 = ax * ay;
    const float s = 1.0f / 255.0f;
    f3 r;
    r.x = w00 * ((t00 & 255u) * s) + w10 * ((t10 & 255u) * s) + w01 * ((t01 & 255u) * s) + w11 * ((t11 & 255u) * s);
    r.y = w00 * (((t00 >> 8) & 255u) * s) + w10 * (((t10 >> 8) & 255u) * s) + w01 * (((t01 >> 8) & 255u) * s) + w11 * (((t11 >> 8) & 255u) * s);
    r.z = w00 * (((t00 >> 16) & 255u) * s) + w10 * (((t10 >> 16) & 255u) * s) + w01 * (((t01 >> 16) & 255u) * s) + w11 * (((t11 >> 16) & 255u) * s);
    return r;
}
struct Geom { f3 P, N; float u, v; int mat; bool emitter; };
SPC_DEV Geom local_geometry(const DeviceScene& S, const HitRec& h) {
    const size_t base = (size_t)h.tri * 4;
    const float4 a = ldq(S.tris, base), b = ldq(S.tris, base + 1), c = ldq(S.tris, base + 2), d = ldq(S.tris, base + 3);
    const f3 P0 = mk3(a.x, a.y, a.z), P1 = mk3(b.x, b.y, b.z), P2 = mk3(c.x, c.y, c.z);
    Geom g;
    const float w = 1.0f - h.u - h.v;
    g.P = w * P0 + h.u * P1 + h.v * P2;
    g.N = normalize(cross(P1 - P0, P2 - P0));
    g.u = w * a.w + h.u * c.w + h.v * d.y;
    g.v = w * b.w + h.u * d.x + h.v * d.z;
    const uint32_t meta = __float_as_uint(d.w);
    g.mat = (int)(meta & 0x7fffffffu);
    g.emitter = (meta & 0x80000000u) != 0;
    return g;
}
template <bool COUNT>
SPC_DEV void color_tex_sample(const DeviceScene& S, const Geom& g, Pbr& m, Counts<COUNT>& cn) {  // hit_program.cu:182-198
    if (m.albedo_tex > 0) {
        const f3 t = tex_fetch_rgb(S.tex[m.albedo_tex - 1], g.u, g.v);
        m.base = mk3(powf(t.x, 2.2f), powf(t.y, 2.2f), powf(t.z, 2.2f));  // linearize cuProg.h:361-368
        cn.add(C_TEX);
    }
}

// ---- Disney BSDF (cuProg.h:686-899) ---------------------------------------------
struct Onb {
    f3 t, b, n;
    SPC_DEV explicit Onb(f3 normal) {
        n = normal;
        if (fabsf(n.x) > fabsf(n.z)) b = mk3(-n.y, n.x, 0.0f);
        else b = mk3(0.0f, -n.z, n.y);
        b = normalize(b);
        t = cross(b, n);
    }
    SPC_DEV f3 to_world(f3 p) const { return p.x * t + p.y * b + p.z * n; }
};
SPC_DEV f3 cosine_sample_hemisphere(float u1, float u2) {
    const float r = sqrtf(u1);
    const float phi = 2.0f * kPi * u2;
    float s, c;
    sincosf(phi, &s, &c);
    f3 p;
    p.x = r * c; p.y = r * s;
    p.z = sqrtf(fmaxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
    return p;
}
SPC_DEV float schlick(float u) { float m = clampf(1.0f - u, 0.0f, 1.0f); float m2 = m * m; return m2 * m2 * m; }
SPC_DEV float gtr1(float NdH, float a) {
    if (a >= 1.0f) return kInvPi;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NdH * NdH;
    return (a2 - 1.0f) / (kPi * logf(a2) * t);
}
SPC_DEV float gtr2(float NdH, float a) {
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NdH * NdH;
    return a2 / (kPi * t * t);
}
SPC_DEV float smith_ggx(float NdV, float alphaG) {
    float a = alphaG * alphaG, b = NdV * NdV;
    return 1.0f / (NdV + sqrtf(a + b - a * b));
}
SPC_DEV f3 bsdf_eval(const Pbr& m, f3 N, f3 V, f3 L) {
    const float NdL = dot(N, L), NdV = dot(N, V);
    if (NdL <= 0.0f || NdV <= 0.0f) return mk3(0.0f);
    const f3 H = normalize(L + V);
    const float NdH = dot(N, H), LdH = dot(L, H);
    const f3 Cd = m.base;
    const float lum = 0.3f * Cd.x + 0.6f * Cd.y + 0.1f * Cd.z;
    const f3 Ctint = lum > 0.0f ? Cd / lum : mk3(1.0f);
    const f3 Cspec0 = lerp3(m.specular * 0.08f * lerp3(mk3(1.0f), Ctint, m.specularTint), Cd, m.metallic);
    const float FL = schlick(NdL), FV = schlick(NdV);
    const float Fd90 = 0.5f + 2.0f * LdH * LdH * m.roughness;
    const float Fd = lerpf(1.0f, Fd90, FL) * lerpf(1.0f, Fd90, FV);
    const float Fss90 = LdH * LdH * m.roughness;
    const float Fss = lerpf(1.0f, Fss90, FL) * lerpf(1.0f, Fss90, FV);
    const float ss = 1.25f * (Fss * (1.0f / (NdL + NdV) - 0.5f) + 0.5f);
    const float a = fmaxf(0.001f, m.roughness);
    const float Ds = gtr2(NdH, a);
    const float FH = schlick(LdH);
    const f3 Fs = lerp3(Cspec0, mk3(1.0f), FH);
    const float rg = (m.roughness * 0.5f + 0.5f) * (m.roughness * 0.5f + 0.5f);
    const float Gs = smith_ggx(NdL, rg) * smith_ggx(NdV, rg);
    f3 out = ((kInvPi * lerpf(Fd, ss, m.subsurface)) * Cd) * (1.0f - m.metallic) + Gs * Fs * Ds;
    if (m.sheen != 0.0f) {  // sheen term is exactly zero for sheen == 0
        const f3 Csheen = lerp3(mk3(1.0f), Ctint, m.sheenTint);
        out = ((kInvPi * lerpf(Fd, ss, m.subsurface)) * Cd + FH * m.sheen * Csheen) * (1.0f - m.metallic) + Gs * Fs * Ds;
    }
    if (m.clearcoat != 0.0f) {  // clearcoat term is exactly zero for clearcoat == 0
        const float Dr = gtr1(NdH, lerpf(0.1f, 0.001f, m.clearcoatGloss));
        const float Fr = lerpf(0.04f, 1.0f, FH);
        const float Gr = smith_ggx(NdL, 0.25f) * smith_ggx(NdV, 0.25f);
        out = out + mk3(0.25f * m.clearcoat * Gr * Fr * Dr);
    }
    return out;
}
// `Eval(...) / (mat.brdf ? abs(dot(n, dir)) : 1.0f)`: the un-guarded ternary of the bidirectional programs (hit_program.cu:286, 384;
// raygen.cu:271, 278; rmis.h:105) for a material with `brdf <nonzero>` in its .scene block.  operator/(float3, float) multiplies by
// the reciprocal (sutil/vec_math.h:483-487) and x * (1.0f / 1.0f) is x, so the division is only executed on the flagged branch.
// A grazing direction (|n.dir| == 0) gives inf / NaN as upstream: the vertex's later contributions fail ISINVALIDVALUE there and here.
// ENV = false (the timed kernels of a scene with neither an environment map nor a flagged material, DeviceScene::general == 0)
// compiles the test away: 0.6 % of the bedroom frame (A/B on one box, profiles/r04_experiments.md).
template <bool ENV = true>
SPC_DEV f3 brdf_div(const Pbr& m, f3 f, f3 n, f3 dir) {
    if (ENV && m.brdf) f = f / fabsf(dot(n, dir));
    return f;
}
SPC_DEV f3 bsdf_sample(const Pbr& m, f3 N, f3 V, uint32_t& seed) {
    const float probability = rnd(seed);
    const float diffuseRatio = 0.5f * (1.0f - m.metallic);
    const float r1 = rnd(seed), r2 = rnd(seed);
    const Onb onb(N);
    if (probability < diffuseRatio) return onb.to_world(cosine_sample_hemisphere(r1, r2));
    const float a = fmaxf(0.001f, m.roughness);
    const float phi = r1 * 2.0f * kPi;
    const float cosTheta = sqrtf((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float sinPhi, cosPhi;
    sincosf(phi, &sinPhi, &cosPhi);
    const f3 half = onb.to_world(mk3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
    return 2.0f * dot(V, half) * half - V;
}
SPC_DEV float bsdf_pdf(const Pbr& m, f3 n, f3 V, f3 L) {
    const float specularAlpha = fmaxf(0.001f, m.roughness);
    const float diffuseRatio = 0.5f * (1.0f - m.metallic);
    const float specularRatio = 1.0f - diffuseRatio;
    const f3 half = normalize(L + V);
    const float cosTheta = fabsf(dot(half, n));
    const float pdfGTR2 = gtr2(cosTheta, specularAlpha) * cosTheta;
    // kept as written in the reference even for clearcoat == 0: lerp(g1, g2, 1) = g1 + (g2 - g1) is NOT g2 in fp32
    const float pdfGTR1 = gtr1(cosTheta, lerpf(0.1f, 0.001f, m.clearcoatGloss)) * cosTheta;
    const float mix = lerpf(pdfGTR1, pdfGTR2, 1.0f / (1.0f + m.clearcoat));
    const float pdfSpec = mix / (4.0f * fabsf(dot(L, half)));
    const float pdfDiff = fabsf(dot(L, n)) * kInvPi;
    return diffuseRatio * pdfDiff + specularRatio * pdfSpec;
}

}  // namespace spc

// ORACLE — test infrastructure only (see vec.h).
// Restates the Disney-principled BSDF of OptiXPathTracer/cuProg.h:
//   Onb 81-112, cosine_sample_hemisphere 114-124, SchlickFresnel 686-691,
//   GTR1 693-699, GTR2 701-706, smithG_GGX 708-713, Eval 735-799,
//   Sample 826-866, Pdf 868-899.  The `#ifdef BRDF` branches are dead in the
//   reference (macro never defined) and are not restated; the LIVE uses of
//   Pbr::brdf are the five `/ (mat.brdf ? abs(dot(n, L)) : 1.0f)` ternaries at the
//   call sites of Eval (spcbpt_ref.h, oracle_capi.cpp).
// Known answer (recorded from the reference's own code):
//   rough .5, metal 0, base (.8,.5,.3), N=+z, V=norm(.3,.2,.9), seed tea<4>(1,2)
//   -> L=(0.9040936,-0.304971,0.2993451) f=(0.2700801,0.1705985,0.1042775) pdf=0.06876558
#pragma once
#include "rng.h"
#include "vec.h"

namespace orc {

// Field set of MaterialData::Pbr (cuda/MaterialData.h:82-100); defaults of
// MaterialData() (41-58) for the parameters the .scene hand-off drops (q17).
struct Pbr {
    float3 base_color = {1, 1, 1};
    float metallic = 1.0f;
    float roughness = 1.0f;
    float specular = 0.5f;
    float specularTint = 0.0f;
    float subsurface = 0.0f;
    float sheen = 0.0f;
    float sheenTint = 0.5f;
    float clearcoat = 0.0f;
    float clearcoatGloss = 1.0f;
    int albedo_tex = 0;  // 0 none, else texture index + 1
    bool brdf = false;   // MaterialData.h:99; read by the five un-guarded ternaries (hit_program.cu:286, 384; raygen.cu:271, 278; rmis.h:105)
};

struct Onb {  // cuProg.h:81-112
    float3 m_tangent, m_binormal, m_normal;
    explicit Onb(const float3& normal) {
        m_normal = normal;
        if (fabsf(m_normal.x) > fabsf(m_normal.z)) {
            m_binormal.x = -m_normal.y;
            m_binormal.y = m_normal.x;
            m_binormal.z = 0;
        } else {
            m_binormal.x = 0;
            m_binormal.y = -m_normal.z;
            m_binormal.z = m_normal.y;
        }
        m_binormal = normalize(m_binormal);
        m_tangent = cross(m_binormal, m_normal);
    }
    void inverse_transform(float3& p) const {
        p = p.x * m_tangent + p.y * m_binormal + p.z * m_normal;
    }
};

inline void cosine_sample_hemisphere(float u1, float u2, float3& p) {  // cuProg.h:114-124
    const float r = sqrtf(u1);
    const float phi = 2.0f * M_PIf_ * u2;
    p.x = r * cosf(phi);
    p.y = r * sinf(phi);
    p.z = sqrtf(fmaxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
}

inline float sqr(float x) { return x * x; }
inline float SchlickFresnel(float u) {
    float m = clampf(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
inline float GTR1(float NDotH, float a) {
    if (a >= 1.0f) return (1.0f / M_PIf_);
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return (a2 - 1.0f) / (M_PIf_ * logf(a2) * t);
}
inline float GTR2(float NDotH, float a) {
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return a2 / (M_PIf_ * t * t);
}
inline float smithG_GGX(float NDotv, float alphaG) {
    float a = alphaG * alphaG;
    float b = NDotv * NDotv;
    return 1.0f / (NDotv + sqrtf(a + b - a * b));
}

inline float3 Eval(const Pbr& mat, const float3& normal, const float3& V, const float3& L) {  // cuProg.h:735-799
    float3 N = normal;
    float NDotL = dot(N, L);
    float NDotV = dot(N, V);
    if (NDotL <= 0.0f || NDotV <= 0.0f) return make_float3(0.0f);

    float3 H = normalize(L + V);
    float NDotH = dot(N, H);
    float LDotH = dot(L, H);

    float3 Cdlin = mat.base_color;
    float Cdlum = 0.3f * Cdlin.x + 0.6f * Cdlin.y + 0.1f * Cdlin.z;

    float3 Ctint = Cdlum > 0.0f ? Cdlin / Cdlum : make_float3(1.0f);
    float3 Cspec0 = lerp(mat.specular * 0.08f * lerp(make_float3(1.0f), Ctint, mat.specularTint), Cdlin, mat.metallic);
    float3 Csheen = lerp(make_float3(1.0f), Ctint, mat.sheenTint);

    float FL = SchlickFresnel(NDotL), FV = SchlickFresnel(NDotV);
    float Fd90 = 0.5f + 2.0f * LDotH * LDotH * mat.roughness;
    float Fd = lerp(1.0f, Fd90, FL) * lerp(1.0f, Fd90, FV);

    float Fss90 = LDotH * LDotH * mat.roughness;
    float Fss = lerp(1.0f, Fss90, FL) * lerp(1.0f, Fss90, FV);
    float ss = 1.25f * (Fss * (1.0f / (NDotL + NDotV) - 0.5f) + 0.5f);

    float a = fmaxf(0.001f, mat.roughness);
    float Ds = GTR2(NDotH, a);
    float FH = SchlickFresnel(LDotH);
    float3 Fs = lerp(Cspec0, make_float3(1.0f), FH);
    float roughg = sqr(mat.roughness * 0.5f + 0.5f);
    float Gs = smithG_GGX(NDotL, roughg) * smithG_GGX(NDotV, roughg);

    float3 Fsheen = FH * mat.sheen * Csheen;

    float Dr = GTR1(NDotH, lerp(0.1f, 0.001f, mat.clearcoatGloss));
    float Fr = lerp(0.04f, 1.0f, FH);
    float Gr = smithG_GGX(NDotL, 0.25f) * smithG_GGX(NDotV, 0.25f);

    float3 out = ((1.0f / M_PIf_) * lerp(Fd, ss, mat.subsurface) * Cdlin + Fsheen) * (1.0f - mat.metallic) +
                 Gs * Fs * Ds + make_float3(0.25f * mat.clearcoat * Gr * Fr * Dr);
    return out;
}

inline float3 Sample(const Pbr& mat, const float3& N, const float3& V, uint32_t& seed) {  // cuProg.h:826-866
    float3 dir;
    float probability = rnd(seed);
    float diffuseRatio = 0.5f * (1.0f - mat.metallic);
    float r1 = rnd(seed);
    float r2 = rnd(seed);
    Onb onb(N);
    if (probability < diffuseRatio) {
        cosine_sample_hemisphere(r1, r2, dir);
        onb.inverse_transform(dir);
    } else {
        float a = fmaxf(0.001f, mat.roughness);
        float phi = r1 * 2.0f * M_PIf_;
        float cosTheta = sqrtf((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
        float sinTheta = sqrtf(1.0f - (cosTheta * cosTheta));
        float sinPhi = sinf(phi);
        float cosPhi = cosf(phi);
        float3 half = make_float3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
        onb.inverse_transform(half);
        dir = 2.0f * dot(V, half) * half - V;
    }
    return dir;
}

inline float Pdf(const Pbr& mat, float3 normal, float3 V, float3 L) {  // cuProg.h:868-899
    float3 n = normal;
    float specularAlpha = fmaxf(0.001f, mat.roughness);
    float clearcoatAlpha = lerp(0.1f, 0.001f, mat.clearcoatGloss);
    float diffuseRatio = 0.5f * (1.f - mat.metallic);
    float specularRatio = 1.f - diffuseRatio;
    float3 half = normalize(L + V);
    float cosTheta = fabsf(dot(half, n));
    float pdfGTR2 = GTR2(cosTheta, specularAlpha) * cosTheta;
    float pdfGTR1 = GTR1(cosTheta, clearcoatAlpha) * cosTheta;
    float ratio = 1.0f / (1.0f + mat.clearcoat);
    // `4.0` is a double literal in the reference (cuProg.h:892)
    float pdfSpec = (float)((double)lerp(pdfGTR1, pdfGTR2, ratio) / (4.0 * (double)fabsf(dot(L, half))));
    float pdfDiff = fabsf(dot(L, n)) * (1.0f / M_PIf_);
    float pdf = diffuseRatio * pdfDiff + specularRatio * pdfSpec;
    return pdf;
}

}  // namespace orc

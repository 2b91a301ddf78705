// Device library of the MI355X SPCBPT hot path (gfx950, wave64).  Scalar branchy FP32 — no MFMA.
// What each block replaces in the reference (paths relative to src/OptiXPathTracer):
//   traversal      -> optixTrace closest / terminate-on-first-hit (cuProg.h:384-487)
//   hit geometry   -> getLocalGeometry (../cuda/LocalGeometry.h:59-175), ColorTexSample (hit_program.cu:182-198)
//   bsdf_*         -> Tracer::Eval / Sample / Pdf (cuProg.h:735-899)
//   tree_label     -> classTree::tree_index (decisionTree/classTree_common.h:39-51)
//   rmis_*         -> rmis.h:16-389
//   binary_sample  -> cuProg.h:245-264
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace spc {

#define SPC_DEV __device__ __forceinline__
static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kInvPi = 1.0f / 3.14159265358979323846f;
static constexpr float kEps = SPCBPT_SCENE_EPSILON;

struct f3 { float x, y, z; };
SPC_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
SPC_DEV f3 mk3(float s) { return mk3(s, s, s); }
SPC_DEV f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
SPC_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
SPC_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
SPC_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
SPC_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
SPC_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
SPC_DEV f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
SPC_DEV f3 operator/(f3 a, float s) { float inv = 1.0f / s; return a * inv; }
SPC_DEV f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
SPC_DEV f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
SPC_DEV f3& operator*=(f3& a, f3 b) { a = a * b; return a; }
SPC_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// No FMA contraction here: fma(a.y, b.z, -(a.z*b.y)) turns the EXACT zero components of axis-aligned / vertical
// geometry normals into +-1e-9 rounding residue, and the subspace octrees split normals at 0 (`n.x > mid.x`), so a
// contracted cross product re-labels every vertex on such faces (measured: 1.5 % of the light vertices of the Cornell box).
SPC_DEV f3 cross(f3 a, f3 b) {
#pragma clang fp contract(off)
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
SPC_DEV f3 normalize(f3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
SPC_DEV float lerpf(float a, float b, float t) { return a + t * (b - a); }
SPC_DEV f3 lerp3(f3 a, f3 b, float t) { return a + t * (b - a); }
SPC_DEV float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
SPC_DEV float max3(f3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
SPC_DEV float sum3(f3 a) { return a.x + a.y + a.z; }
SPC_DEV float4 ldq(const float* base, size_t quad) { return reinterpret_cast<const float4*>(base)[quad]; }

// ---- RNG (../cuda/random.h:31-67) -------------------------------------------
SPC_DEV uint32_t tea4(uint32_t v0, uint32_t v1) {
    uint32_t s0 = 0;
#pragma unroll
    for (int n = 0; n < 4; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
SPC_DEV float rnd(uint32_t& s) {
    s = 1664525u * s + 1013904223u;
    return (float)(s & 0x00FFFFFFu) / (float)0x01000000;
}

// ---- event counters ----------------------------------------------------------
template <bool ON>
struct Counts {
    unsigned v[ON ? C_COUNT : 1];
    SPC_DEV void clear() { if (ON) { for (int i = 0; i < C_COUNT; i++) v[i] = 0; } }
    SPC_DEV void add(int slot, unsigned n = 1) { if (ON) v[slot] += n; }
    SPC_DEV void flush(unsigned long long* g) {
        if (ON && g) {
#pragma unroll
            for (int i = 0; i < C_COUNT; i++) {
                unsigned x = v[i];
                for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
                if ((threadIdx.x & 63) == 0 && x) atomicAdd(&g[i], (unsigned long long)x);
            }
        }
    }
};

}  // namespace spc

// The rest of the device library, in dependency order (round 6: one 1 700-line header became five; same translation unit, same code)
#include "dev_traversal.h"
#include "dev_bsdf.h"
#include "dev_sampling.h"
#include "dev_rmis.h"

namespace spc {

// ---- the environment map as a light (envInfo_device, cuProg.h:125-243; uv2dir / dir2uv, optixPathTracer.h:139-165) ------------
// `2 * v - 1.0` and `0.5 * M_1_PIf` promote to double upstream; kept (these run once per light path).
SPC_DEV f3 uv2dir(float u, float v) {
    const float phi = asinf((float)(2 * v - 1.0));
    const float theta = (float)(u / (0.5 * 0.318309886183790671538f) - kPi);
    return mk3(cosf(phi) * sinf(theta), cosf(kPi * 0.5f - phi), cosf(phi) * cosf(theta));
}
SPC_DEV void dir2uv(f3 dir, float& u, float& v) {
    const float theta = atan2f(dir.x, dir.z);
    const float phi = kPi * 0.5f - acosf(dir.y);
    u = (theta + kPi) * (0.5f * 0.318309886183790671538f);
    v = 0.5f * (1.0f + sinf(phi));
}
SPC_DEV f3 env_sample(const DEnv& E, uint32_t& seed) {   // 164-184: the bisection of binary_sample over the texels, then a jittered point of the texel
    const float index = rnd(seed);
    int mid = E.size / 2 - 1, l = 0, r = E.size;
    while (r - l > 1) {
        if (index < E.cmf[mid]) r = mid + 1;
        else l = mid + 1;
        mid = (l + r) / 2 - 1;
    }
    const int cx = l % E.width, cy = l / E.width;
    const float r1 = rnd(seed), r2 = rnd(seed);
    return uv2dir((float)(cx + r1) / (float)E.width, (float)(cy + r2) / (float)E.height);
}
SPC_DEV int env_label(const DEnv& E, f3 dir) {   // 201-216
    float u, v;
    dir2uv(dir, u, v);
    const int ux = min(max((int)floorf(u * E.div_level), 0), E.div_level - 1);
    const int uy = min(max((int)floorf(v * E.div_level), 0), E.div_level - 1);
    return SPCBPT_NUM_SUBSPACE - 1 - (ux * E.div_level + uy);
}
SPC_DEV f3 env_color(const DEnv& E, f3 dir) {   // 217-226: tex2D<float4>, normalised coordinates, wrap, linear (exact-fraction bilinear, like tex_fetch_rgb)
    float u, v;
    dir2uv(dir, u, v);
    const float x = u * (float)E.width - 0.5f, y = v * (float)E.height - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    int x0 = (int)fx % E.width, y0 = (int)fy % E.height;
    if (x0 < 0) x0 += E.width;
    if (y0 < 0) y0 += E.height;
    const int x1 = x0 + 1 == E.width ? 0 : x0 + 1, y1 = y0 + 1 == E.height ? 0 : y0 + 1;
    const float4 t00 = ldq(E.tex, (size_t)y0 * E.width + x0), t10 = ldq(E.tex, (size_t)y0 * E.width + x1),
                 t01 = ldq(E.tex, (size_t)y1 * E.width + x0), t11 = ldq(E.tex, (size_t)y1 * E.width + x1);
    const float w00 = (1 - ax) * (1 - ay), w10 = ax * (1 - ay), w01 = (1 - ax) * ay, w11 = ax * ay;
    return mk3(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x, w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
               w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z);
}
SPC_DEV float env_pdf(const DEnv& E, f3 dir) {   // 227-241 (M_PI: double)
    float u, v;
    dir2uv(dir, u, v);
    const int cx = min((int)(u * E.width), E.width - 1), cy = min((int)(v * E.height), E.height - 1);
    const int index = cx + cy * E.width;
    const float pdf1 = index == 0 ? E.cmf[index] : E.cmf[index] - E.cmf[index - 1];
    return (float)(pdf1 * E.size / (4 * 3.14159265358979323846));
}

// ---- light sampling (cuProg.h:554-666: QUAD, ENV) ------------------------------
struct LightSampleD { f3 position, emission, normal; float pdf; int subspace; };
// lightSample::operator() for the ENV light (611-619) + traceMode (660-664): the sky direction, its radiance, and the start of the
// sub-path: a point on the disk of radius r that faces the scene from 10 r away (sample_projectPos, 185-195), pdf projectPdf per area.
// Returns the sample with `normal` = minus the sky direction = the direction the sub-path is shot in (lightSample::normal, 639).
SPC_DEV LightSampleD env_light_sample(const DeviceScene& S, uint32_t& seed, float& dir_pos_pdf) {
    const DEnv& E = S.env;
    LightSampleD s;
    const f3 direction = env_sample(E, seed);
    s.emission = env_color(E, direction);
    s.subspace = env_label(E, direction);
    s.pdf = env_pdf(E, direction) / (float)S.n_lights;
    s.normal = -direction;
    const float r1 = rnd(seed), r2 = rnd(seed);
    const Onb onb(direction);
    const f3 pos = cosine_sample_hemisphere(r1, r2);
    s.position = 10 * E.r * direction + pos.x * E.r * onb.t + pos.y * E.r * onb.b + ld3(E.center);
    dir_pos_pdf = E.project_pdf;
    return s;
}
SPC_DEV LightSampleD light_reverse_sample(const DeviceScene& S, const DLight& L, float r1, float r2) {
    LightSampleD s;
    const float r3 = 1 - r1 - r2;
    s.position = ld3(L.u) * r1 + ld3(L.v) * r2 + ld3(L.corner) * r3;
    s.emission = ld3(L.emission);
    s.normal = ld3(L.normal);
    s.pdf = (1.0f / L.area) / (float)S.n_lights;
    const int xb = min(max((int)floorf(r1 * L.div_level), 0), L.div_level - 1);
    const int yb = min(max((int)floorf(r2 * L.div_level), 0), L.div_level - 1);
    s.subspace = SPCBPT_NUM_SUBSPACE - (L.ss_base + xb * L.div_level + yb) - 1;
    return s;
}
SPC_DEV int pick_light(const DeviceScene& S, uint32_t& seed) {
    return min(max((int)floorf(rnd(seed) * S.n_lights), 0), S.n_lights - 1);
}

// ---- film (raygen.cu:45-68, 430-442; ../cuda/helpers.h:35-67) ------------------------
SPC_DEV float to_srgb(float c) {
    return c < 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
}
SPC_DEV uint32_t quant8(float x) {
    x = clampf(x, 0.0f, 1.0f);
    return min((uint32_t)(x * 256.0f), 255u);
}
SPC_DEV void film_store(float* buf, uint32_t width, uint32_t x, uint32_t y, f3 result) {  // radiance of one frame's sample
    reinterpret_cast<float4*>(buf)[(size_t)y * width + x] = make_float4(result.x, result.y, result.z, 1.0f);
}
SPC_DEV void film_write(const KParams& p, uint32_t x, uint32_t y, f3 result) {
    const size_t idx = (size_t)y * p.width + x;
    if (p.result) {  // deferred: k_film_merge applies the running mean and the tone map in frame order
        reinterpret_cast<float4*>(p.result)[idx] = make_float4(result.x, result.y, result.z, 1.0f);
        return;
    }
    float4* acc = reinterpret_cast<float4*>(p.accum);
    f3 c = result;
    if (p.subframe > 0) {
        const float a = 1.0f / (float)(p.subframe + 1);
        const float4 prev = acc[idx];
        c = lerp3(mk3(prev.x, prev.y, prev.z), c, a);
    }
    acc[idx] = make_float4(c.x, c.y, c.z, 1.0f);
    if (p.frame) {
        const float lum = 0.3f * c.x + 0.6f * c.y + 0.1f * c.z;
        const float s = 1.0f / (1.0f + lum / 1.5f);
        const f3 t = c * s;
        p.frame[idx] = quant8(to_srgb(clampf(t.x, 0.f, 1.f))) | (quant8(to_srgb(clampf(t.y, 0.f, 1.f))) << 8) |
                       (quant8(to_srgb(clampf(t.z, 0.f, 1.f))) << 16) | (255u << 24);
    }
}
SPC_DEV f3 camera_ray(const KParams& p, uint32_t x, uint32_t y, uint32_t& seed, uint32_t subframe) {  // raygen.cu:332-343
    seed = tea4(y * p.width + x, subframe);
    float jx = 0.5f, jy = 0.5f;
    if (subframe != 0) { jx = rnd(seed); jy = rnd(seed); }
    const float dx = 2.0f * (((float)x + jx) / (float)p.width) - 1.0f;
    const float dy = 2.0f * (((float)y + jy) / (float)p.height) - 1.0f;
    return normalize(dx * ld3(p.U) + dy * ld3(p.V) + ld3(p.W));
}
SPC_DEV f3 camera_ray(const KParams& p, uint32_t x, uint32_t y, uint32_t& seed) { return camera_ray(p, x, y, seed, p.subframe); }

}  // namespace spc

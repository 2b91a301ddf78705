// ORACLE — test infrastructure only (see vec.h).
// Scene container + CPU BVH standing in for the closed OptiX traversal.
//   * scene assembly follows OptiXPathTracer/scene_shift.cpp: materials 64-91,
//     emissive pseudo-materials 92-103, Light[] with ssBase/divLevel 108-154,
//     two triangles per quad light with UVs (0,0)(1,0)(0,1)(1,1) 252-328.
//   * closest hit = optixTrace(..., OPTIX_RAY_FLAG_CULL_BACK_FACING_TRIANGLES, ...)
//     (cuProg.h:384-461): nearest triangle in (tmin, tmax); back faces are
//     culled only on single-sided geometry, i.e. the light quads
//     (sutil/Scene.cpp:1030 + scene_shift.cpp:68).
//   * any hit = visibilityTest (cuProg.h:463-487): no culling.
// Parity UNPINNED at this boundary: OptiX's ray/triangle arithmetic and CUDA's
// texture filter are closed; this file uses Moller-Trumbore and
// exact-fraction bilinear filtering.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "../include/spcbpt.h"
#include "bsdf.h"
#include "vec.h"

namespace orc {

struct Light {  // cuda/Light.h:65-84 (QUAD) ; u, v are ABSOLUTE corner points (scene_shift.cpp:130-131)
    float3 corner, u, v, emission, normal;
    float area;
    int id, divLevel, ssBase;
    bool env = false;   // Light::Type::ENV: default-constructed otherwise (scene_shift.cpp:146-151: id / divLevel / ssBase stay as Light() leaves them)
};

// uv2dir / dir2uv (optixPathTracer.h:139-165); `2 * v - 1.0` and `0.5 * M_1_PIf` promote to double as written
inline float3 uv2dir(float2 uv) {
    float3 dir;
    float u = uv.x, v = uv.y;
    float phi = asinf((float)(2 * v - 1.0));
    float theta = (float)(u / (0.5 * M_1_PIf_) - M_PIf_);
    dir.y = cosf(M_PIf_ * 0.5f - phi);
    dir.x = cosf(phi) * sinf(theta);
    dir.z = cosf(phi) * cosf(theta);
    return dir;
}
inline float2 dir2uv(float3 dir) {
    float theta = atan2f(dir.x, dir.z);
    float phi = M_PIf_ * 0.5f - acosf(dir.y);
    float u = (theta + M_PIf_) * (0.5f * M_1_PIf_);
    float v = 0.5f * (1.0f + sinf(phi));
    return float2{u, v};
}

// envInfo + envInfo_device (optixPathTracer.h:98-137, cuProg.h:125-243) and its set-up env_params_setup / envMapCMFBuild /
// surroundsIndex (optixPathTracer.cpp:381-461).  The texture is the raster with its rows flipped (HDRLoader::loadTexture,
// scene_shift.cpp:528-540) while the CMF is built over the raster as read -- the sampling density is the image upside down.  As
// written upstream (unbiased: pdf() uses the same CMF); kept.
struct EnvInfo {
    std::vector<float4> tex;   // width x height, row j = raster row height - 1 - j
    std::vector<float> cmf;
    float r = 0;
    float3 center{};
    int size = 0, width = 0, height = 0, divLevel = 0, ssBase = 0;
    bool valid = false;

    int coord2index(int cx, int cy) const { return cx + cy * width; }
    float3 sample(uint32_t& seed) const {   // cuProg.h:164-184: the bespoke bisection of binary_sample, then a jittered point of the texel
        float index = rnd(seed);
        int mid = size / 2 - 1, l = 0, rr = size;
        while (rr - l > 1) {
            if (index < cmf[mid]) rr = mid + 1;
            else l = mid + 1;
            mid = (l + rr) / 2 - 1;
        }
        int cx = l % width, cy = l / width;
        float r1 = rnd(seed), r2 = rnd(seed);
        float u = (float)(cx + r1) / (float)width, v = (float)(cy + r2) / (float)height;
        return uv2dir(float2{u, v});
    }
    float3 sample_projectPos(float3 dir, uint32_t& seed) const {   // 185-195
        const float r1 = rnd(seed);
        const float r2 = rnd(seed);
        float3 pos;
        Onb onb(dir);
        cosine_sample_hemisphere(r1, r2, pos);
        return 10 * r * (dir) + pos.x * r * onb.m_tangent + pos.y * r * onb.m_binormal + center;
    }
    float projectPdf() const { return (float)(1 / (3.14159265358979323846 * r * r)); }   // 196-199 (M_PI: double)
    void uv2coord(float2 uv, int& x, int& y) const {
        x = (int)(uv.x * width); y = (int)(uv.y * height);
        x = x < width - 1 ? x : width - 1; y = y < height - 1 ? y : height - 1;
    }
    int getLabel(float3 dir) const {   // 201-216
        float2 uv = dir2uv(dir);
        int ux = clampi((int)floorf(uv.x * divLevel), 0, divLevel - 1);
        int uy = clampi((int)floorf(uv.y * divLevel), 0, divLevel - 1);
        return SPCBPT_NUM_SUBSPACE - 1 - (ux * divLevel + uy);
    }
    float3 color(float3 dir) const {   // 217-226: tex2D<float4>, normalised coordinates, wrap, linear filter (exact-fraction bilinear here)
        float2 uv = dir2uv(dir);
        const float x = uv.x * (float)width - 0.5f, y = uv.y * (float)height - 0.5f;
        const float fx = floorf(x), fy = floorf(y);
        const float ax = x - fx, ay = y - fy;
        int x0 = (int)fx % width, y0 = (int)fy % height;
        if (x0 < 0) x0 += width;
        if (y0 < 0) y0 += height;
        const int x1 = x0 + 1 == width ? 0 : x0 + 1, y1 = y0 + 1 == height ? 0 : y0 + 1;
        const float4 t00 = tex[(size_t)y0 * width + x0], t10 = tex[(size_t)y0 * width + x1], t01 = tex[(size_t)y1 * width + x0], t11 = tex[(size_t)y1 * width + x1];
        const float w00 = (1 - ax) * (1 - ay), w10 = ax * (1 - ay), w01 = (1 - ax) * ay, w11 = ax * ay;
        return make_float3(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x, w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
                           w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z);
    }
    float pdf(float3 dir) const {   // 227-241 (M_PI: double)
        float2 uv = dir2uv(dir);
        int cx, cy;
        uv2coord(uv, cx, cy);
        int index = coord2index(cx, cy);
        float pdf1 = index == 0 ? cmf[index] : cmf[index] - cmf[index - 1];
        return (float)(pdf1 * size / (4 * 3.14159265358979323846));
    }
    // env_params_setup (optixPathTracer.cpp:431-461): `raster` = width x height RGBA floats as HDRLoader leaves them (row 0 = top)
    void setup(const float* raster, int w, int h, float3 c, float radius) {
        width = w; height = h; size = w * h;
        divLevel = (int)sqrt(0.5 * SPCBPT_NUM_SUBSPACE_LIGHTSOURCE);
        ssBase = 0;
        tex.resize((size_t)size);
        for (int i = 0; i < w; i++)
            for (int j = 0; j < h; j++) {
                const float* q = raster + ((size_t)(h - j - 1) * w + i) * 4;
                tex[(size_t)j * w + i] = float4{q[0], q[1], q[2], 1.0f};
            }
        // envMapCMFBuild (404-430): luminance + the mean of the up-to-12 neighbours within |dx| + |dy| <= 2, accumulated in float
        std::vector<float> p2((size_t)size);
        const float uniform_rate = 0.25f;
        const float uniform_pdf = (float)(1.0 / size);
        auto lum = [&](int i) { return raster[(size_t)i * 4] + raster[(size_t)i * 4 + 1] + raster[(size_t)i * 4 + 2]; };
        for (int i = 0; i < size; i++) {
            const int cx = i % w, cy = i / w;
            int n = 0, idxs[13];
            for (int dx = -2; dx <= 2; dx++)
                for (int dy = -2; dy <= 2; dy++)
                    if (abs(dx) + abs(dy) <= 2) {
                        const int sx = cx + dx, sy = cy + dy;
                        if (sx >= 0 && sy >= 0 && sx < w && sy < h) idxs[n++] = sx + sy * w;
                    }
            p2[i] = lum(i);
            for (int k = 0; k < n; k++) p2[i] += lum(idxs[k]) / n;
            if (i >= 1) p2[i] += p2[i - 1];
        }
        const float sum = p2[size - 1];
        for (int i = 0; i < size; i++) {
            p2[i] /= sum;
            p2[i] = p2[i] * (1 - uniform_rate) + (uniform_pdf * (i + 1) * uniform_rate);
        }
        cmf = p2;
        center = c; r = radius;
        valid = true;
    }
};

struct Texture {
    std::vector<uint8_t> rgba;
    int w = 0, h = 0;
};

struct Hit {
    float t;
    int tri;  // -1 = miss
    float bu, bv;
};

struct Counters {
    uint64_t closest_rays = 0, shadow_rays = 0, node_visits = 0, tri_tests = 0, surface_vertices = 0,
             textured_hits = 0, tree_nodes = 0, cmf_probes = 0, connections = 0, gamma_q_reads = 0,
             lvc_stores = 0, pixel_samples = 0, eye_paths = 0, light_paths = 0;
    void add(const Counters& o) {
        closest_rays += o.closest_rays; shadow_rays += o.shadow_rays; node_visits += o.node_visits;
        tri_tests += o.tri_tests; surface_vertices += o.surface_vertices; textured_hits += o.textured_hits;
        tree_nodes += o.tree_nodes; cmf_probes += o.cmf_probes; connections += o.connections;
        gamma_q_reads += o.gamma_q_reads; lvc_stores += o.lvc_stores; pixel_samples += o.pixel_samples;
        eye_paths += o.eye_paths; light_paths += o.light_paths;
    }
};

struct BVHNode {
    float3 lo, hi;
    int left, right;   // children (internal)
    int first, count;  // triangle range (leaf when count > 0)
};

struct Scene {
    std::vector<float3> P;
    std::vector<float2> UV;
    std::vector<uint32_t> idx;
    std::vector<int> tri_mat;
    std::vector<Pbr> materials;      // params.materials: scene materials, then one pseudo-material per light
    std::vector<int> mat_light_id;   // MaterialData::light_id for emissive pseudo-materials, else -1
    std::vector<Light> lights;       // params.lights
    EnvInfo sky;                     // params.sky
    std::vector<Texture> textures;
    std::vector<BVHNode> nodes;
    std::vector<int> tri_order;
    int n_scene_triangles = 0;

    int n_triangles() const { return (int)(idx.size() / 3); }
    bool tri_is_emitter(int t) const { return mat_light_id[tri_mat[t]] >= 0; }

    void build(const spcbpt_scene_desc& d) {
        P.resize(d.n_vertices);
        UV.assign(d.n_vertices, float2{0, 0});
        for (int i = 0; i < d.n_vertices; i++) {
            P[i] = load3(d.vertices + 3 * i);
            if (d.texcoords) UV[i] = float2{d.texcoords[2 * i], d.texcoords[2 * i + 1]};
        }
        idx.assign(d.indices, d.indices + 3 * (size_t)d.n_triangles);
        tri_mat.assign(d.tri_material, d.tri_material + d.n_triangles);
        n_scene_triangles = d.n_triangles;
        for (int i = 0; i < d.n_materials; i++) {
            const spcbpt_material& m = d.materials[i];
            Pbr p;
            p.base_color = load3(m.base_color);
            p.metallic = m.metallic; p.roughness = m.roughness; p.specular = m.specular;
            p.specularTint = m.specular_tint; p.subsurface = m.subsurface; p.sheen = m.sheen;
            p.sheenTint = m.sheen_tint; p.clearcoat = m.clearcoat; p.clearcoatGloss = m.clearcoat_gloss;
            p.albedo_tex = m.albedo_tex;
            p.brdf = m.brdf != 0;   // scene_shift.cpp:75 (int -> bool)
            materials.push_back(p);
            mat_light_id.push_back(-1);
        }
        for (int i = 0; i < d.n_textures; i++) {
            Texture t;
            t.w = d.textures[i].width; t.h = d.textures[i].height;
            t.rgba.assign(d.textures[i].rgba, d.textures[i].rgba + (size_t)4 * t.w * t.h);
            textures.push_back(std::move(t));
        }
        int ssBase = 0;  // no env map (scene_shift.cpp:110)
        for (int i = 0; i < d.n_lights; i++) {
            const spcbpt_quad_light& s = d.lights[i];
            Light l;
            float3 pos = load3(s.position), su = load3(s.u), sv = load3(s.v);
            l.emission = load3(s.emission);
            l.corner = pos;
            l.u = pos + su;
            l.v = pos + sv;
            l.normal = normalize(cross(su, sv));
            l.area = length(cross(su, sv));
            l.id = (int)lights.size();
            l.ssBase = ssBase;
            l.divLevel = s.div_level;
            ssBase += s.div_level * s.div_level;
            lights.push_back(l);
            // emissive pseudo-material (scene_shift.cpp:92-103); pbr fields keep MaterialData() defaults
            Pbr p;
            materials.push_back(p);
            mat_light_id.push_back(i);
            // quad geometry (scene_shift.cpp:276-293)
            uint32_t base = (uint32_t)P.size();
            P.push_back(l.corner); P.push_back(l.u); P.push_back(l.v); P.push_back(l.u + l.v - l.corner);
            UV.push_back(float2{0, 0}); UV.push_back(float2{1, 0}); UV.push_back(float2{0, 1}); UV.push_back(float2{1, 1});
            uint32_t q[6] = {base, base + 1, base + 3, base, base + 3, base + 2};
            idx.insert(idx.end(), q, q + 6);
            tri_mat.push_back((int)materials.size() - 1);
            tri_mat.push_back((int)materials.size() - 1);
        }
        build_bvh();
    }

    // The environment map as one more light (scene_shift.cpp:108-153, optixPathTracer.cpp:431-461): the quad lights' patch
    // subspaces start at 0.5 * NUM_SUBSPACE_LIGHTSOURCE, the sky takes 0 .. divLevel^2 - 1, and an ENV light is appended.
    void set_environment(const float* raster, int w, int h, float3 center, float radius) {
        if (sky.valid) return;
        for (Light& l : lights) l.ssBase += (int)(0.5 * SPCBPT_NUM_SUBSPACE_LIGHTSOURCE);
        Light e{};
        e.env = true; e.id = (int)lights.size(); e.divLevel = 0; e.ssBase = 0;   // (Light() leaves them indeterminate upstream: scene_shift.cpp:146-151)
        lights.push_back(e);
        sky.setup(raster, w, h, center, radius);
    }

    // ---- BVH (median split on the widest centroid axis, leaves <= 4) ----
    void tri_bounds(int t, float3& lo, float3& hi) const {
        float3 a = P[idx[3 * t]], b = P[idx[3 * t + 1]], c = P[idx[3 * t + 2]];
        lo = {fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z))};
        hi = {fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z))};
    }
    int build_rec(int first, int count, const std::vector<float3>& cen) {
        BVHNode n;
        n.lo = make_float3(1e30f); n.hi = make_float3(-1e30f);
        float3 clo = make_float3(1e30f), chi = make_float3(-1e30f);
        for (int i = first; i < first + count; i++) {
            float3 lo, hi;
            tri_bounds(tri_order[i], lo, hi);
            n.lo = {fminf(n.lo.x, lo.x), fminf(n.lo.y, lo.y), fminf(n.lo.z, lo.z)};
            n.hi = {fmaxf(n.hi.x, hi.x), fmaxf(n.hi.y, hi.y), fmaxf(n.hi.z, hi.z)};
            float3 c = cen[tri_order[i]];
            clo = {fminf(clo.x, c.x), fminf(clo.y, c.y), fminf(clo.z, c.z)};
            chi = {fmaxf(chi.x, c.x), fmaxf(chi.y, c.y), fmaxf(chi.z, c.z)};
        }
        n.left = n.right = -1; n.first = first; n.count = count;
        int id = (int)nodes.size();
        nodes.push_back(n);
        if (count <= 4) return id;
        float3 ext = chi - clo;
        int axis = ext.x > ext.y ? (ext.x > ext.z ? 0 : 2) : (ext.y > ext.z ? 1 : 2);
        auto key = [&](int t) { const float3& c = cen[t]; return axis == 0 ? c.x : (axis == 1 ? c.y : c.z); };
        int mid = first + count / 2;
        std::nth_element(tri_order.begin() + first, tri_order.begin() + mid, tri_order.begin() + first + count,
                         [&](int a, int b) { return key(a) < key(b); });
        int l = build_rec(first, mid - first, cen);
        int r = build_rec(mid, first + count - mid, cen);
        nodes[id].left = l; nodes[id].right = r; nodes[id].count = 0;
        return id;
    }
    void build_bvh() {
        int nt = n_triangles();
        tri_order.resize(nt);
        std::vector<float3> cen(nt);
        for (int t = 0; t < nt; t++) {
            tri_order[t] = t;
            float3 lo, hi;
            tri_bounds(t, lo, hi);
            cen[t] = (lo + hi) * 0.5f;
        }
        nodes.clear();
        nodes.reserve(2 * nt / 3 + 16);
        if (nt > 0) build_rec(0, nt, cen);
    }

    static bool slab(const BVHNode& n, const float3& o, const float3& inv, float tmin, float tmax) {
        float tx0 = (n.lo.x - o.x) * inv.x, tx1 = (n.hi.x - o.x) * inv.x;
        float ty0 = (n.lo.y - o.y) * inv.y, ty1 = (n.hi.y - o.y) * inv.y;
        float tz0 = (n.lo.z - o.z) * inv.z, tz1 = (n.hi.z - o.z) * inv.z;
        float t0 = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
        float t1 = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tmax));
        return t0 <= t1 * 1.0000004f;
    }
    // Moller-Trumbore; accepts tmin < t < tmax.
    bool tri_hit(int t, const float3& o, const float3& d, float tmin, float tmax, bool cull_backface,
                 float& ot, float& ou, float& ov) const {
        const float3 v0 = P[idx[3 * t]], v1 = P[idx[3 * t + 1]], v2 = P[idx[3 * t + 2]];
        const float3 e1 = v1 - v0, e2 = v2 - v0;
        if (cull_backface && dot(cross(e1, e2), d) > 0.0f) return false;
        const float3 p = cross(d, e2);
        const float det = dot(e1, p);
        if (det == 0.0f) return false;
        const float inv = 1.0f / det;
        const float3 tv = o - v0;
        const float u = dot(tv, p) * inv;
        if (u < 0.0f || u > 1.0f) return false;
        const float3 q = cross(tv, e1);
        const float v = dot(d, q) * inv;
        if (v < 0.0f || u + v > 1.0f) return false;
        const float tt = dot(e2, q) * inv;
        if (!(tt > tmin && tt < tmax)) return false;
        ot = tt; ou = u; ov = v;
        return true;
    }
    static float3 safe_inv(const float3& d) {
        auto f = [](float x) { return 1.0f / (fabsf(x) > 1e-20f ? x : (x < 0 ? -1e-20f : 1e-20f)); };
        return {f(d.x), f(d.y), f(d.z)};
    }
    Hit closest_hit(const float3& o, const float3& d, float tmin, float tmax, Counters* c) const {
        Hit h{tmax, -1, 0, 0};
        if (c) c->closest_rays++;
        if (nodes.empty()) return h;
        const float3 inv = safe_inv(d);
        int stack[128], sp = 0;
        stack[sp++] = 0;
        while (sp) {
            const BVHNode& n = nodes[stack[--sp]];
            if (c) c->node_visits++;
            if (!slab(n, o, inv, tmin, h.t)) continue;
            if (n.count > 0) {
                for (int i = n.first; i < n.first + n.count; i++) {
                    int t = tri_order[i];
                    if (c) c->tri_tests++;
                    float tt, u, v;
                    if (tri_hit(t, o, d, tmin, h.t, tri_is_emitter(t), tt, u, v)) h = Hit{tt, t, u, v};
                }
            } else {
                stack[sp++] = n.left;
                stack[sp++] = n.right;
            }
        }
        return h;
    }
    bool any_hit(const float3& o, const float3& d, float tmin, float tmax, Counters* c) const {
        if (c) c->shadow_rays++;
        if (nodes.empty()) return false;
        const float3 inv = safe_inv(d);
        int stack[128], sp = 0;
        stack[sp++] = 0;
        while (sp) {
            const BVHNode& n = nodes[stack[--sp]];
            if (c) c->node_visits++;
            if (!slab(n, o, inv, tmin, tmax)) continue;
            if (n.count > 0) {
                for (int i = n.first; i < n.first + n.count; i++) {
                    if (c) c->tri_tests++;
                    float tt, u, v;
                    if (tri_hit(tri_order[i], o, d, tmin, tmax, false, tt, u, v)) return true;
                }
            } else {
                stack[sp++] = n.left;
                stack[sp++] = n.right;
            }
        }
        return false;
    }
    // visibilityTest(pos_A, pos_B) cuProg.h:463-487
    bool visibilityTest(const float3& A, const float3& B, Counters* c) const {
        float3 bias_pos = B - A;
        float len = length(bias_pos);
        float3 dir = bias_pos / len;
        return !any_hit(A, dir, SPCBPT_SCENE_EPSILON, len - SPCBPT_SCENE_EPSILON, c);
    }

    // bilinear + wrap fetch of RGBA8 -> [0,1] floats (cudaReadModeNormalizedFloat, Scene.cpp:634-645)
    void tex_fetch(int tex_id, float u, float v, float out[4]) const {
        const Texture& T = textures[tex_id];
        float x = u * (float)T.w - 0.5f, y = v * (float)T.h - 0.5f;
        float fx = floorf(x), fy = floorf(y);
        float ax = x - fx, ay = y - fy;
        auto wrap = [](int i, int n) { int m = i % n; return m < 0 ? m + n : m; };
        int x0 = wrap((int)fx, T.w), x1 = wrap((int)fx + 1, T.w);
        int y0 = wrap((int)fy, T.h), y1 = wrap((int)fy + 1, T.h);
        for (int k = 0; k < 4; k++) {
            float t00 = T.rgba[4 * ((size_t)y0 * T.w + x0) + k] * (1.0f / 255.0f);
            float t10 = T.rgba[4 * ((size_t)y0 * T.w + x1) + k] * (1.0f / 255.0f);
            float t01 = T.rgba[4 * ((size_t)y1 * T.w + x0) + k] * (1.0f / 255.0f);
            float t11 = T.rgba[4 * ((size_t)y1 * T.w + x1) + k] * (1.0f / 255.0f);
            out[k] = (1 - ax) * (1 - ay) * t00 + ax * (1 - ay) * t10 + (1 - ax) * ay * t01 + ax * ay * t11;
        }
    }
};

// getLocalGeometry, triangle case (cuda/LocalGeometry.h:59-175): P by barycentric
// lerp, N = Ng = normalize(cross(P1-P0, P2-P0)) (normals are never uploaded,
// scene_shift.cpp:234), UV lerp.
struct LocalGeometry {
    float3 P, N;
    float2 UV;
};
inline LocalGeometry getLocalGeometry(const Scene& s, int tri, float bu, float bv) {
    uint32_t i0 = s.idx[3 * tri], i1 = s.idx[3 * tri + 1], i2 = s.idx[3 * tri + 2];
    const float3 P0 = s.P[i0], P1 = s.P[i1], P2 = s.P[i2];
    LocalGeometry g;
    g.P = (1.0f - bu - bv) * P0 + bu * P1 + bv * P2;
    g.N = normalize(cross(P1 - P0, P2 - P0));
    const float2 U0 = s.UV[i0], U1 = s.UV[i1], U2 = s.UV[i2];
    g.UV = (1.0f - bu - bv) * U0 + bu * U1 + bv * U2;
    return g;
}

// ColorTexSample + linearize (hit_program.cu:182-198, cuProg.h:361-368).
inline void ColorTexSample(const Scene& s, const LocalGeometry& g, Pbr& pbr, Counters* c) {
    if (pbr.albedo_tex > 0) {
        float t[4];
        s.tex_fetch(pbr.albedo_tex - 1, g.UV.x, g.UV.y, t);
        pbr.base_color = make_float3(powf(t[0], 2.2f), powf(t[1], 2.2f), powf(t[2], 2.2f));
        if (c) c->textured_hits++;
    }
}

}  // namespace orc

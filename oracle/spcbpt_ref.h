// ORACLE — test infrastructure only (see vec.h).
// CPU restatement of the SPCBPT hot path of ssufujia/SPCBPT-OptiX7.  Every
// function cites the reference file:line it follows (paths relative to
// src/OptiXPathTracer unless prefixed).  Scope: QUAD lights and the light side
// of the environment map (ENV: sub-paths that start on the sky, connections to
// them, NEE of "pt" -- row f4; the sky is never SEEN by an eye sub-path, SURVEY
// q1: light_hit_env has no caller); DIRECTION lights never reach params.lights
// (scene_shift.cpp:117-123); `isBrdf/inBrdf/lastBrdf` are never true (SURVEY
// q3) so their early-outs are omitted.
//
// PARITY STATUS: the reference cannot be built here (needs the OptiX 7.5 SDK
// headers + nvcc; no stand-ins are written).  Pinned pieces: rng.h against the
// reference's own cuda/random.h (oracle/_ref), sRGB/quantize and the camera
// frame against cuda/helpers.h and sutil/Camera.cpp (oracle/_ref), the BSDF
// against the known answer recorded in SURVEY.md a7.  Everything else in this
// file is PARITY UNPINNED: a careful restatement with no reference output to
// check against.
#pragma once
#include <cfloat>
#include <vector>

#include "bsdf.h"
#include "rng.h"
#include "scene.h"
#include "vec.h"

namespace orc {

enum LightType { SPHERE, QUAD, DIRECTION, ENV, HIT_LIGHT_SOURCE, ENV_MISS, NORMALHIT };  // light_parameters.h:8-11

// classTree::tree_node (decisionTree/classTree_common.h:11-38)
struct tree_node {
    float3 mid;
    int child[8];
    int label;
    int type;  // 0 position, 1 normal, 2 direction
    bool leaf;
    int getChild(float3 p) const {
        int ind = 0;
        ind += p.x > mid.x ? 1 : 0;
        ind += p.y > mid.y ? 2 : 0;
        ind += p.z > mid.z ? 4 : 0;
        return child[ind];
    }
    int operator()(float3 position, float3 normal, float3 direction) const {
        return type == 0 ? getChild(position) : (type == 1 ? getChild(normal) : getChild(direction));
    }
};
// classTree::tree_index (classTree_common.h:39-51)
inline int tree_index(const tree_node* root, float3 position, float3 normal, float3 direction, Counters* c) {
    int node_id = 0;
    if (c) c->tree_nodes++;
    while (root[node_id].leaf == false) {
        node_id = root[node_id](position, normal, direction);
        if (c) c->tree_nodes++;
    }
    return root[node_id].label;
}

// BDPTVertex (BDPTVertex.h:9-70), value-initialised
struct BDPTVertex {
    float3 position{}, normal{}, flux{}, color{}, lastPosition{}, RMIS_pointer_3{};
    float2 uv{};
    float RMIS_pointer = 0, last_lum = 0, lastNormalProjection = 0, pdf = 0, singlePdf = 0, lastSinglePdf = 0;
    short materialId = 0, subspaceId = 0, depth = 0, lastZoneId = 0, type = QUAD;
    bool isOrigin = false;
    bool isLastVertex_direction = false;   // this vertex comes straight from the environment map (BDPTVertex.h:62)
    uint32_t path_id = 0;  // oracle-only bookkeeping (global light path index)
    bool is_LL_DIRECTION() const { return isLastVertex_direction; }            // BDPTVertex.h:67
    bool is_DIRECTION() const { return type == DIRECTION || type == ENV; }     // BDPTVertex.h:68
};

struct BDPTPath {  // BDPTVertex.h:72-117, 3-slot ring
    BDPTVertex v[3];
    int size = 0;
    BDPTVertex& operator()(int i) { return v[(size - 1 - i) % 3]; }
    BDPTVertex& currentVertex() { return (*this)(0); }
    BDPTVertex& nextVertex() { return (*this)(-1); }
    BDPTVertex& lastVertex() { return (*this)(1); }
    void clear() { size = 0; }
    void push() { size++; }
    bool hit_lightSource() { return currentVertex().type == HIT_LIGHT_SOURCE || currentVertex().type == ENV_MISS; }
};

struct Subspace { int jump_bias, id, size; float sum_pmf, Q; };  // optixPathTracer.h:43-51

struct LightTraceParams {  // optixPathTracer.h:52-66
    int num_core = 1000, core_padding = 800, M_per_core = 100, launch_frame = 0;
    // test knob, NOT reference behaviour (see spcbpt_light_trace_params::decorrelate_bsdf_stream); false = reference
    bool decorrelate_bsdf_stream = false;
    BDPTVertex* ans = nullptr;
    uint8_t* validState = nullptr;
    int get_element_count() const { return num_core * core_padding; }
};

struct SubspaceSampler {  // optixPathTracer.h:89-97
    const BDPTVertex* LVC = nullptr;
    const Subspace* subspace = nullptr;
    const float* cmfs = nullptr;
    const int* jump_buffer = nullptr;
    int vertex_count = 0, path_count = 0;
};
struct SamplerStorage {  // the function-static thrust vectors of LVC_Process (device_thrust.cu:287-293)
    std::vector<Subspace> subspace;
    std::vector<float> cmfs;
    std::vector<int> jump_buffer;
};

struct Params {  // MyParams (optixPathTracer.h:191-199 + whitted.h:64-84)
    const Scene* scene = nullptr;
    unsigned width = 0, height = 0, subframe_index = 0;
    float4* accum_buffer = nullptr;
    uint32_t* frame_buffer = nullptr;  // uchar4 packed, may be null
    float3 eye{}, U{}, V{}, W{};
    LightTraceParams lt;
    SubspaceSampler sampler;
    // subspaceMacroInfo (optixPathTracer.h:166-190)
    const tree_node* eye_tree = nullptr;
    const tree_node* light_tree = nullptr;
    const float* Q = nullptr;
    const float* CMFGamma = nullptr;
    Counters* counters = nullptr;
    // test knob, NOT reference behaviour: accumulate the per-subspace CMFs in double (the product's device scan does);
    // false = the reference's serial float prefix sums (device_thrust.cu:273-286)
    bool cmf_double = false;
    // test knob, NOT reference behaviour: skip work whose contribution is exactly zero, as the product does -- the shadow ray
    // of a connection whose BSDF factor is zero and the rest of an eye path whose sampled direction has a
    // BSDF value of zero (d11).  The image is unchanged; only the event counts differ.
    bool skip_null_connections = false;
    // "plain BDPT", the comparator of BASELINE config 5: draw the light vertex with SubspaceSampler_device::uniformSample
    // (cuProg.h:283-289; defined in the reference, never called there) instead of the two-stage subspace sampler
    bool uniform_lvc = false;
    // test knob, COUNTERS ONLY (values are untouched): charge the classification and first-stage events the product's timed
    // kernels execute instead of the reference's -- every vertex is classified once under both trees when it is created and the
    // relabels of rmis.h:58-79 / 131-151 read those labels (DIR_JUDGE 0 makes a label a property of the vertex), and
    // the first sampling stage reads 2 x 32 CMF values + the bin's two instead of bisecting (device_lib.h: sample_first_stage)
    bool count_as_executed = false;
    // test knob, NOT reference behaviour: "pt" shoots the shadow ray of its environment-map next-event sample ALONG the sampled
    // direction.  Upstream (hit_program.cu:513) aims it at P + direction + 2 r with the scalar added to every component, i.e.
    // towards (1, 1, 1): its "pt" shadows the sky wrongly.  The product restates that as written; the fixed form is the
    // independent estimator the unbiasedness test of the sky's LIGHT side compares "SPCBPT_eye" with.
    bool pt_env_nee_fixed = false;
    // test knob, NOT reference behaviour: an eye sub-path that leaves the scene SEES the sky, weighted by rmis::light_hit_env
    // (rmis.h:325-358 -- defined upstream, called nowhere: __miss__BDPTVertex only sets `done`).  The connection weights of
    // the sky's light side (connection_direction_lightSource etc.) reserve a share for this strategy; without it the reference's
    // image is too dark by that share.  With the knob the estimator is complete, which is what the consistency test of the
    // restated light side needs: SPCBPT + this == PT.
    bool env_miss_strategy = false;

    float Gamma(int eye_id, int light_id) const {
        if (CMFGamma && Q) {
            if (counters && !count_as_executed) counters->gamma_q_reads += light_id == 0 ? 1 : 2;   // (executed order: one read of the product's Gamma / Q table, charged by gamma_ss)
            return light_id == 0 ? CMFGamma[eye_id * SPCBPT_NUM_SUBSPACE + light_id]
                                 : CMFGamma[eye_id * SPCBPT_NUM_SUBSPACE + light_id] -
                                       CMFGamma[eye_id * SPCBPT_NUM_SUBSPACE + light_id - 1];
        }
        return 1;
    }
    float gamma_ss(int eye_id, int light_id) const {
        if (CMFGamma && Q) {
            if (counters) counters->gamma_q_reads++;
            return Gamma(eye_id, light_id) / Q[light_id];
        }
        return 1;
    }
};

// labelUnit::getLabel (cuProg.h:1109-1123)
inline int getLabel(const Params& P, float3 position, float3 normal, float3 dir, bool light_side, bool count = true) {
    Counters* c = count ? P.counters : nullptr;
    if (light_side) {
        if (P.light_tree) return tree_index(P.light_tree, position, normal, dir, c);
    } else {
        if (P.eye_tree) return tree_index(P.eye_tree, position, normal, dir, c);
    }
    return 0;
}

// connectRate_SOL (cuProg.h:70-78)
inline float connectRate_SOL(const Params& P, int eye_label, int light_label, float lum_sum) {
    return P.gamma_ss(eye_label, light_label) * lum_sum * SPCBPT_CONNECTION_N;
}
inline float3 connectRate_SOL(const Params& P, int eye_label, int light_label, float3 lum_sum) {
    return P.gamma_ss(eye_label, light_label) * lum_sum * (float)SPCBPT_CONNECTION_N;
}

// lightSample (cuProg.h:554-666), QUAD branch
struct lightSample {
    float3 position{}, emission{}, direction{};
    float2 uv{};
    const Light* bindLight = nullptr;
    float pdf = 0, dir_pdf = 0;
    int subspaceId = 0;
    void ReverseSample(const Params& P, const Light& light, float2 uv_) {  // 571-591
        bindLight = &light;
        float r1 = uv_.x, r2 = uv_.y, r3 = 1 - r1 - r2;
        position = light.u * r1 + light.v * r2 + light.corner * r3;
        emission = light.emission;
        pdf = (float)(1.0 / (double)light.area);
        pdf /= (float)P.scene->lights.size();
        uv = make_float2(r1, r2);
        int x_block = clampi((int)floorf(uv.x * light.divLevel), 0, light.divLevel - 1);
        int y_block = clampi((int)floorf(uv.y * light.divLevel), 0, light.divLevel - 1);
        int lightSpaceId = light.ssBase + x_block * light.divLevel + y_block;
        subspaceId = SPCBPT_NUM_SUBSPACE - lightSpaceId - 1;
    }
    void sample(const Params& P, const Light& light, uint32_t& seed) {  // 602-621
        if (!light.env) {
            float r1 = rnd(seed);
            float r2 = rnd(seed);
            ReverseSample(P, light, make_float2(r1, r2));
        } else {   // Light::Type::ENV (611-619)
            const EnvInfo& SKY = P.scene->sky;
            direction = SKY.sample(seed);
            emission = SKY.color(direction);
            subspaceId = SKY.getLabel(direction);
            uv = dir2uv(direction);
            pdf = SKY.pdf(direction);
            pdf /= (float)P.scene->lights.size();
        }
        bindLight = &light;
    }
    void sample(const Params& P, uint32_t& seed) {  // 622-627
        int n = (int)P.scene->lights.size();
        int light_id = clampi((int)floorf(rnd(seed) * n), 0, n - 1);
        sample(P, P.scene->lights[light_id], seed);
    }
    // the light pick of the training pass: QUAD lights only.  Upstream picks among all lights there too (raygen.cu:820-823) and then
    // reads the sample's `position`, which the ENV branch never sets: undefined, so the sky is left out of the NEE candidates.
    void sample_quad(const Params& P, uint32_t& seed) {
        int n = (int)P.scene->lights.size() - (P.scene->sky.valid ? 1 : 0);
        int light_id = clampi((int)floorf(rnd(seed) * n), 0, n - 1);
        sample(P, P.scene->lights[light_id], seed);
    }
    float3 normal() const { return bindLight ? (bindLight->env ? -direction : bindLight->normal) : make_float3(0); }  // 628-643
    float3 trace_direction() const { return bindLight->env ? -direction : direction; }   // 644-647
    void traceMode(const Params& P, uint32_t& seed) {                                    // 648-665 (dir_pdf and dir_pos_pdf share a union)
        if (!bindLight->env) {
            float r1 = rnd(seed);
            float r2 = rnd(seed);
            Onb onb(bindLight->normal);
            cosine_sample_hemisphere(r1, r2, direction);
            onb.inverse_transform(direction);
            dir_pdf = fabsf(dot(direction, bindLight->normal)) / M_PIf_;
        } else {
            position = P.scene->sky.sample_projectPos(direction, seed);
            dir_pdf = P.scene->sky.projectPdf();
        }
    }
};

// PayloadBDPTVertex (cuProg.h:303-323)
struct PayloadBDPTVertex {
    BDPTPath path;
    float3 origin{}, ray_direction{}, throughput{}, result{};
    float pdf = 0;
    uint32_t seed = 0;
    int depth = 0;
    bool done = false;
    void clear() {
        path.clear();
        depth = 0;
        done = false;
        throughput = make_float3(1);
        result = make_float3(0.0f);
    }
};
// whitted::PayloadRadiance (whitted.h:86-108)
struct PayloadRadiance {
    float3 vis_pos_A{}, vis_pos_B{}, currentResult{};
    float3 result = make_float3(0), origin{}, ray_direction{}, throughput = make_float3(1.0f);
    float pdf = 0;
    int depth = 0;
    uint32_t seed = 0;
    bool done = false;
};

// ------------------------------------------------------------------ rmis.h
namespace rmis {
inline Pbr getMat(const Params& P, const BDPTVertex& v) {  // rmis.h:16-21
    Pbr mat = P.scene->materials[v.materialId];
    mat.base_color = v.color;
    return mat;
}
inline void tracing_init_light(BDPTVertex& Mid, BDPTVertex& Last) {  // 22-26
    Mid.RMIS_pointer = Last.RMIS_pointer / Last.singlePdf;
}
inline float getRR(const BDPTVertex& v) {  // 28-40 ; max(float, 0.3) with the double literal (q10)
    float rr_rate = fmaxf3(v.color);
    rr_rate = (float)((double)rr_rate > 0.3 ? (double)rr_rate : 0.3);
    return rr_rate;
}
inline float getLast_pdf(const Params& P, const BDPTVertex& Mid, float3 in_dir) {  // 41-51
    Pbr mat = getMat(P, Mid);
    float3 out_vec = Mid.lastPosition - Mid.position;
    float3 out_dir = normalize(out_vec);
    float pdf = Mid.is_LL_DIRECTION() ? Pdf(mat, Mid.normal, in_dir, out_dir)   // 45-47: the step back leads to the sky: a direction, no area measure
                                      : Pdf(mat, Mid.normal, in_dir, out_dir) / dot(out_vec, out_vec) * Mid.lastNormalProjection;
    pdf *= getRR(Mid);
    return pdf;
}
inline float getLL_pdf(const Params& P, const BDPTVertex& Mid, const BDPTVertex& Last) {  // 52-57
    float3 in_dir = normalize(Mid.position - Last.position);
    return getLast_pdf(P, Last, in_dir);
}
inline float tracing_weight_light(const Params& P, const BDPTVertex& Mid, const BDPTVertex& Last) {  // 58-79
    float3 inver_dir = normalize(Mid.position - Last.position);
    int eye_label = getLabel(P, Last.position, Last.normal, inver_dir, false, !P.count_as_executed);
    int light_label = Last.lastZoneId;
    float lum_sum = Last.last_lum;
    return connectRate_SOL(P, eye_label, light_label, lum_sum);
}
inline void tracing_update_light(const Params& P, BDPTVertex& Mid, BDPTVertex& Last) {  // 80-94
    float LL_pdf = getLL_pdf(P, Mid, Last);
    float weight = tracing_weight_light(P, Mid, Last);
    float last_single_pdf = Last.singlePdf;
    Mid.RMIS_pointer = ((Last.RMIS_pointer * LL_pdf) + weight) / last_single_pdf;
}
inline float3 getFluxMultiplier(const Params& P, const BDPTVertex& v, float3 in_dir, float3 out_dir) {  // 102-112
    Pbr mat = getMat(P, v);
    float3 flux_ratio = Eval(mat, v.normal, in_dir, out_dir) / (mat.brdf ? fabsf(dot(v.normal, out_dir)) : 1.0f);   // 105
    float pdf_ratio = Pdf(mat, v.normal, in_dir, out_dir);
    float rr = getRR(v);
    float cos_theta = fabsf(dot(v.normal, out_dir));
    return flux_ratio * cos_theta / pdf_ratio / rr;
}
inline float3 getFluxMultiplier(const Params& P, const BDPTVertex& v, float3 in_dir) {  // 113-118
    float3 out_vec = v.lastPosition - v.position;
    float3 out_dir = normalize(out_vec);
    return getFluxMultiplier(P, v, in_dir, out_dir);
}
inline float3 tracing_weight_eye(const Params& P, const BDPTVertex& Mid, const BDPTVertex& Last) {  // 131-151
    if (Last.depth == 1) return make_float3(0.0f);
    float3 inver_dir = Mid.is_DIRECTION() ? -Mid.normal : normalize(Mid.position - Last.position);   // 141
    int eye_label = Last.lastZoneId;
    int light_label = getLabel(P, Last.position, Last.normal, inver_dir, true, !P.count_as_executed);
    float3 lum = make_float3(1.0f);
    return connectRate_SOL(P, eye_label, light_label, lum);
}
inline float getPdf(const Params& P, const BDPTVertex& begin, const BDPTVertex& end, float3 in_dir) {  // 153-172
    Pbr mat = getMat(P, begin);
    float pdf;
    if (end.is_DIRECTION()) {   // 158-162
        float3 out_dir = -end.normal;
        pdf = Pdf(mat, begin.normal, in_dir, out_dir);
    } else {
        float3 out_vec = end.position - begin.position;
        float3 out_dir = normalize(out_vec);
        pdf = Pdf(mat, begin.normal, in_dir, out_dir) / dot(out_vec, out_vec) * fabsf(dot(out_dir, end.normal));
    }
    pdf *= getRR(begin);
    return pdf;
}
inline float getPdf_from_light_source(const Params& P, const BDPTVertex& light, const BDPTVertex& end) {  // 173-188 ; M_PI is double
    if (light.is_DIRECTION()) {   // 183-187
        float3 dir = light.normal;
        return P.scene->sky.projectPdf() * fabsf(dot(dir, end.normal));
    }
    float3 conn_vec = end.position - light.position;
    float3 conn_dir = normalize(conn_vec);
    float pdf_angle = (float)((double)fabsf(dot(light.normal, conn_dir)) / 3.14159265358979323846);
    float angle2a = fabsf(dot(end.normal, conn_dir)) / (dot(conn_vec, conn_vec));
    return pdf_angle * angle2a;
}
inline void tracing_update_eye(const Params& P, BDPTVertex& Mid, BDPTVertex& Last) {  // 189-203
    float LL_pdf = getLL_pdf(P, Mid, Last);
    float3 weight = tracing_weight_eye(P, Mid, Last);
    float last_single_pdf = Last.singlePdf;
    float3 flux_multiplier = getFluxMultiplier(P, Last, normalize(Mid.position - Last.position));
    Mid.RMIS_pointer_3 = ((Last.RMIS_pointer_3 * LL_pdf * flux_multiplier) + weight) / last_single_pdf;
}
inline void tracing_init_eye(BDPTVertex& Mid, BDPTVertex&) { Mid.RMIS_pointer_3 = make_float3(0.0f); }  // 204-207

inline float general_connection(const Params& P, const BDPTVertex& eye, const BDPTVertex& light) {  // 212-247
    float3 connect_vec = eye.position - light.position;
    float3 connect_dir = normalize(connect_vec);
    float3 flux = light.flux / light.pdf;

    float LL_pdf_A = getLL_pdf(P, light, eye);
    float3 flux_multiplier_0 = getFluxMultiplier(P, eye, -connect_dir);
    float3 weight_A = tracing_weight_eye(P, light, eye);
    float3 D_A_0 = ((eye.RMIS_pointer_3 * LL_pdf_A * flux_multiplier_0) + weight_A);

    float3 LA = normalize(light.lastPosition - light.position);
    float pdf_A = getPdf(P, light, eye, LA);
    float3 flux_multiplier_1 = getFluxMultiplier(P, light, LA, connect_dir);
    float D_A = float3weight(D_A_0 * pdf_A * flux_multiplier_1 * flux / eye.singlePdf);

    float weight = float3weight(connectRate_SOL(P, eye.subspaceId, light.subspaceId, flux));

    float LL_pdf_B = getLL_pdf(P, eye, light);
    float weight_B = tracing_weight_light(P, eye, light);
    float D_B_0 = (light.RMIS_pointer * LL_pdf_B) + weight_B;

    float3 LB = normalize(eye.lastPosition - eye.position);
    float pdf_B = getPdf(P, eye, light, LB);
    float D_B = D_B_0 * pdf_B / light.singlePdf;
    return weight / (weight + D_A + D_B);
}
inline float connection_direction_lightSource(const Params& P, const BDPTVertex& eye, const BDPTVertex& light) {  // 249-280
    float3 connect_dir = light.normal;
    float3 flux = light.flux / light.pdf;

    float LL_pdf_A = getLL_pdf(P, light, eye);   // (in_dir runs from the eye vertex to the light vertex's POSITION on the sky disk: as written)
    float3 flux_multiplier_0 = getFluxMultiplier(P, eye, -connect_dir);
    float3 weight_A = tracing_weight_eye(P, light, eye);
    float3 D_A_0 = ((eye.RMIS_pointer_3 * LL_pdf_A * flux_multiplier_0) + weight_A);

    float pdf_A = getPdf_from_light_source(P, light, eye);
    float flux_multiplier_1 = light.is_DIRECTION() ? (float)(1.0 / P.scene->sky.projectPdf()) : M_PIf_;
    float D_A = float3weight(D_A_0 * pdf_A * flux_multiplier_1 * flux / eye.singlePdf);

    float weight = float3weight(connectRate_SOL(P, eye.subspaceId, light.subspaceId, flux));

    float D_B_0 = light.RMIS_pointer;
    float3 LB = normalize(eye.lastPosition - eye.position);
    float pdf_B = getPdf(P, eye, light, LB);
    float D_B = D_B_0 * pdf_B / light.singlePdf;
    return weight / (weight + D_A + D_B);
}
inline float connection_lightSource(const Params& P, const BDPTVertex& eye, const BDPTVertex& light) {  // 281-313
    float3 connect_vec = eye.position - light.position;
    float3 connect_dir = normalize(connect_vec);
    float3 flux = light.flux / light.pdf;

    float LL_pdf_A = getLL_pdf(P, light, eye);
    float3 flux_multiplier_0 = getFluxMultiplier(P, eye, -connect_dir);
    float3 weight_A = tracing_weight_eye(P, light, eye);
    float3 D_A_0 = ((eye.RMIS_pointer_3 * LL_pdf_A * flux_multiplier_0) + weight_A);

    float pdf_A = getPdf_from_light_source(P, light, eye);
    float flux_multiplier_1 = M_PIf_;
    float D_A = float3weight(D_A_0 * pdf_A * flux_multiplier_1 * flux / eye.singlePdf);

    float weight = float3weight(connectRate_SOL(P, eye.subspaceId, light.subspaceId, flux));

    float D_B_0 = light.RMIS_pointer;
    float3 LB = normalize(eye.lastPosition - eye.position);
    float pdf_B = getPdf(P, eye, light, LB);
    float D_B = D_B_0 * pdf_B / light.singlePdf;
    return weight / (weight + D_A + D_B);
}
inline float light_hit(const Params& P, BDPTVertex& eye, BDPTVertex& light) {  // 359-389
    float3 connect_vec = eye.position - light.position;
    float3 connect_dir = normalize(connect_vec);
    float3 flux = light.flux / light.pdf;

    float LL_pdf_A = getLL_pdf(P, light, eye);
    float3 flux_multiplier_0 = getFluxMultiplier(P, eye, -connect_dir);
    float3 weight_A = tracing_weight_eye(P, light, eye);
    float3 D_A_0 = ((eye.RMIS_pointer_3 * LL_pdf_A * flux_multiplier_0) + weight_A);

    float pdf_A = getPdf_from_light_source(P, light, eye);
    float flux_multiplier_1 = M_PIf_;
    float D_A = float3weight(D_A_0 * pdf_A * flux_multiplier_1 * flux / eye.singlePdf);
    float weight = float3weight(connectRate_SOL(P, eye.subspaceId, light.subspaceId, flux));

    float D_B = light.RMIS_pointer;
    float3 LB = normalize(eye.lastPosition - eye.position);
    float pdf_B = getPdf(P, eye, light, LB);
    return D_B / ((weight + D_A) / pdf_B * light.singlePdf + D_B);
}
// (uncalled upstream; see Params::env_miss_strategy).  As written upstream the flux multiplier of the eye vertex is taken with
// in_dir = -connect_dir although connect_dir here already points TO the sky (= -light.normal; light_hit, which it was copied from, has
// connect_dir pointing from the light to the eye vertex): the multiplier of a direction below the surface, i.e. the strategies with
// a shorter eye sub-path drop out of the weight from eye depth 2 on (orc_debug_env_partition: miss weight 0.03 where first principles
// say 0.99).  Dead code upstream, so nothing there shows it.  `as_written` = false (what the knob uses) points it at the sky.
inline float light_hit_env(const Params& P, BDPTVertex& eye, BDPTVertex& light, bool as_written = false) {  // 325-358
    float3 connect_dir = -light.normal;
    float3 flux = light.flux / light.pdf;
    float LL_pdf_A = getLast_pdf(P, eye, connect_dir);
    float3 flux_multiplier_0 = getFluxMultiplier(P, eye, as_written ? -connect_dir : connect_dir);
    float3 weight_A = tracing_weight_eye(P, light, eye);
    float3 D_A_0 = ((eye.RMIS_pointer_3 * LL_pdf_A * flux_multiplier_0) + weight_A);
    float pdf_A = getPdf_from_light_source(P, light, eye);
    float flux_multiplier_1 = (float)(1.0 / P.scene->sky.projectPdf());
    float D_A = float3weight(D_A_0 * pdf_A * flux_multiplier_1 * flux / eye.singlePdf);
    float weight = float3weight(connectRate_SOL(P, eye.subspaceId, light.subspaceId, flux));
    float D_B = light.RMIS_pointer;
    float3 LB = normalize(eye.lastPosition - eye.position);
    float pdf_B = getPdf(P, eye, light, LB);
    return D_B / ((weight + D_A) / pdf_B * light.singlePdf + D_B);
}
}  // namespace rmis

// -------------------------------------------------------------- hit programs
struct HitInfo {  // what optixGet* hands a closest-hit program
    int tri;
    float t_hit;
    float3 ray_direction;
    float bu, bv;
};

inline float rr_rate_of(float3 color) {  // hit_program.cu:325-328 / 423-427
    float rr_rate = fmaxf3(color);
    rr_rate = rr_rate < SPCBPT_MIN_RR_RATE ? SPCBPT_MIN_RR_RATE : rr_rate;
    return rr_rate;
}

// __closesthit__eyeSubpath (hit_program.cu:246-340) and __closesthit__lightSubpath (341-438)
inline void closesthit_subpath(const Params& P, PayloadBDPTVertex* prd, const HitInfo& h, bool light_side) {
    const Scene& S = *P.scene;
    const LocalGeometry geom = getLocalGeometry(S, h.tri, h.bu, h.bv);
    float t_hit = h.t_hit;
    float3 ray_direction = h.ray_direction;
    float3 inver_ray_direction = -ray_direction;
    Pbr currentPbr = S.materials[S.tri_mat[h.tri]];
    ColorTexSample(S, geom, currentPbr, P.counters);
    float3 N = geom.N;
    if (dot(N, ray_direction) > 0.f) N = -N;
    prd->ray_direction = Sample(currentPbr, N, inver_ray_direction, prd->seed);
    prd->pdf = Pdf(currentPbr, N, inver_ray_direction, prd->ray_direction);
    prd->origin = geom.P;
    if (!(prd->pdf > 0.0f)) prd->done = true;

    prd->path.push();
    BDPTVertex& Mid = prd->path.currentVertex();
    BDPTVertex& Next = prd->path.nextVertex();
    BDPTVertex& Last = prd->path.lastVertex();
    Mid.position = geom.P;
    Mid.normal = N;
    Mid.type = NORMALHIT;
    float pdf_G = fabsf(dot(Mid.normal, ray_direction) * dot(Last.normal, ray_direction)) / (t_hit * t_hit);
    if (light_side && Last.is_DIRECTION()) pdf_G = fabsf(dot(Mid.normal, ray_direction) * dot(Last.normal, ray_direction));   // hit_program.cu:372-375 (parallel rays: no 1 / t^2)
    if (Last.isOrigin) Mid.flux = Last.flux * pdf_G;
    else Mid.flux = Mid.flux * Last.flux * pdf_G;
    Next.flux = Eval(currentPbr, N, -ray_direction, prd->ray_direction) / (currentPbr.brdf ? fabsf(dot(Mid.normal, prd->ray_direction)) : 1.0f);   // hit_program.cu:286 / 384
    if (P.skip_null_connections && !light_side && Next.flux.x == 0.0f && Next.flux.y == 0.0f && Next.flux.z == 0.0f) prd->done = true;  // d11
    Next.singlePdf = prd->pdf;

    Mid.lastPosition = Last.position;
    if (Last.is_DIRECTION()) Mid.lastPosition = Mid.position - ray_direction;   // hit_program.cu:290-293 / 386-389
    Mid.color = currentPbr.base_color;
    Mid.lastNormalProjection = fabsf(dot(Last.normal, ray_direction));
    Mid.materialId = (short)S.tri_mat[h.tri];
    Mid.subspaceId = (short)getLabel(P, Mid.position, Mid.normal, -ray_direction, light_side);
    // (counters only) the product classifies the new vertex under the other tree at once too; an eye vertex of depth 1 never needs it
    if (P.count_as_executed && P.counters && (light_side || Last.depth + 1 != 1)) (void)getLabel(P, Mid.position, Mid.normal, -ray_direction, !light_side);
    Mid.lastZoneId = Last.subspaceId;
    Mid.isOrigin = false;
    Mid.depth = Last.depth + 1;
    Mid.uv = geom.UV;

    Mid.singlePdf = Mid.singlePdf * pdf_G / fabsf(dot(Last.normal, ray_direction));
    Mid.pdf = Last.pdf * Mid.singlePdf;
    if (light_side) Mid.last_lum = float3weight(Last.flux / Last.pdf);  // 408

    Mid.lastSinglePdf = Last.singlePdf;
    if (light_side) Mid.isLastVertex_direction = Last.depth == 0 && Last.is_DIRECTION();   // hit_program.cu:412
    if (light_side) {
        if (Last.isOrigin) rmis::tracing_init_light(Mid, Last);
        else rmis::tracing_update_light(P, Mid, Last);
    } else {
        if (Mid.depth == 1) rmis::tracing_init_eye(Mid, Last);
        else rmis::tracing_update_eye(P, Mid, Last);
    }
    if (P.counters) P.counters->surface_vertices++;

    float r = rnd(prd->seed);
    float rr_rate = rr_rate_of(Mid.color);
    if (r > rr_rate) {
        prd->done = true;
    } else {
        Next.singlePdf *= rr_rate;
        if (!light_side) prd->throughput *= Next.flux / prd->pdf / rr_rate * dot(N, prd->ray_direction);
    }
}

// __closesthit__eyeSubpath_LightSource (hit_program.cu:62-147)
inline void closesthit_eyeSubpath_LightSource(const Params& P, PayloadBDPTVertex* prd, const HitInfo& h) {
    const Scene& S = *P.scene;
    prd->done = true;
    const int light_id = S.mat_light_id[S.tri_mat[h.tri]];
    const Light& light = S.lights[light_id];
    if (dot(prd->ray_direction, light.normal) > 0) return;

    const LocalGeometry geom = getLocalGeometry(S, h.tri, h.bu, h.bv);
    float t_hit = h.t_hit;
    float3 ray_direction = h.ray_direction;

    prd->path.push();
    BDPTVertex& Mid = prd->path.currentVertex();
    BDPTVertex& Last = prd->path.lastVertex();
    Mid.position = geom.P;
    Mid.normal = light.normal;
    Mid.type = HIT_LIGHT_SOURCE;
    Mid.uv = geom.UV;
    lightSample light_sample;
    light_sample.ReverseSample(P, light, Mid.uv);
    float lightPdf = light_sample.pdf;

    float pdf_G = fabsf(dot(Mid.normal, ray_direction) * dot(Last.normal, ray_direction)) / (t_hit * t_hit);
    if (Last.isOrigin) Mid.flux = Last.flux * pdf_G * light_sample.emission;
    else Mid.flux = Mid.flux * Last.flux * pdf_G * light_sample.emission;

    Mid.lastPosition = Last.position;
    Mid.lastNormalProjection = fabsf(dot(Last.normal, ray_direction));
    Mid.subspaceId = (short)light_sample.subspaceId;
    Mid.lastZoneId = Last.subspaceId;
    Mid.singlePdf = Mid.singlePdf * pdf_G / fabsf(dot(Last.normal, ray_direction));
    Mid.pdf = Last.pdf * Mid.singlePdf;
    Mid.materialId = (short)light_id;
    Mid.depth = Last.depth + 1;

    if (Mid.depth == 1) {
        Mid.RMIS_pointer = 1.0f;
        return;
    }
    BDPTVertex virtual_light;
    virtual_light.position = Mid.position;
    virtual_light.RMIS_pointer = 1;
    virtual_light.normal = Mid.normal;
    virtual_light.pdf = lightPdf;
    virtual_light.singlePdf = lightPdf;
    virtual_light.flux = light_sample.emission;
    virtual_light.subspaceId = Mid.subspaceId;
    Mid.RMIS_pointer = (float)(1.0 / (double)rmis::light_hit(P, Last, virtual_light));
}

// __closesthit__radiance (hit_program.cu:439-552), QUAD branch
inline void closesthit_radiance(const Params& P, PayloadRadiance* prd, const HitInfo& h) {
    const Scene& S = *P.scene;
    const LocalGeometry geom = getLocalGeometry(S, h.tri, h.bu, h.bv);
    Pbr currentPbr = S.materials[S.tri_mat[h.tri]];
    ColorTexSample(S, geom, currentPbr, P.counters);
    float3 N = geom.N;
    if (dot(N, h.ray_direction) > 0.f) N = -N;
    float3 in_dir = -prd->ray_direction;
    float3 result = make_float3(0.0f);
    // clamp(fmaxf(color), MIN_RR_RATE, 1.0): the only viable overload is vec_math's clamp(float,float,float)
    float rr_rate = clampf(fmaxf3(currentPbr.base_color), 0.3f, 1.0f);
    int n_lights = (int)S.lights.size();
    int light_id = clampi((int)floorf(rnd(prd->seed) * n_lights), 0, n_lights - 1);
    const Light& light = S.lights[light_id];
    if (light.env) {   // hit_program.cu:502-518
        lightSample light_sample;
        light_sample.sample(P, light, prd->seed);
        const float3 V = -normalize(h.ray_direction);
        const float3 L = light_sample.direction;
        float L_dot_N = dot(light_sample.direction, N);
        if (L_dot_N > 0.0f) {
            prd->vis_pos_A = geom.P;
            prd->vis_pos_B = geom.P + light_sample.direction + make_float3(S.sky.r * 2);   // float3 + float adds the scalar to every component: as written
            if (P.pt_env_nee_fixed) prd->vis_pos_B = geom.P + light_sample.direction * (S.sky.r * 2);
            float3 eval = Eval(currentPbr, N, V, L);
            result += prd->throughput * light_sample.emission / light_sample.pdf * eval * L_dot_N;
        }
    } else {
        lightSample light_sample;
        light_sample.sample(P, light, prd->seed);
        const float L_dist = length(light_sample.position - geom.P);
        const float3 L = (light_sample.position - geom.P) / L_dist;
        const float3 V = -normalize(h.ray_direction);
        const float3 LN = light.normal;
        const float L_dot_LN = dot(-L, LN);
        const float N_dot_L = dot(N, L);
        const float N_dot_V = dot(N, V);
        if (N_dot_L > 0.0f && N_dot_V > 0.0f && L_dot_LN > 0.0f) {
            prd->vis_pos_A = geom.P;
            prd->vis_pos_B = light_sample.position;
            float3 eval = Eval(currentPbr, N, V, L);
            float MIS_weight;
            {
                float pdf_area = light_sample.pdf;
                float pdf_hit = Pdf(currentPbr, N, V, L) * fabsf(L_dot_LN) / (L_dist * L_dist) * rr_rate;
                MIS_weight = pdf_area / (pdf_hit + pdf_area);
            }
            result += prd->throughput * light_sample.emission * 1.0f / light_sample.pdf * N_dot_L * L_dot_LN /
                      L_dist / L_dist * eval * MIS_weight;
        }
    }
    prd->currentResult += result;
    prd->origin = geom.P;
    if (P.counters) P.counters->surface_vertices++;
    if (rnd(prd->seed) > rr_rate) {
        prd->done = true;
    } else {
        prd->ray_direction = Sample(currentPbr, N, in_dir, prd->seed);
        float pdf = Pdf(currentPbr, N, in_dir, prd->ray_direction);
        if (pdf > 0.0f) {
            prd->throughput *= Eval(currentPbr, N, in_dir, prd->ray_direction) * fabsf(dot(prd->ray_direction, N)) / pdf / rr_rate;
            prd->pdf = pdf * rr_rate;
        } else {
            prd->done = true;
        }
    }
}

// __closesthit__lightsource (hit_program.cu:148-180)
inline void closesthit_lightsource(const Params& P, PayloadRadiance* prd, const HitInfo& h) {
    const Scene& S = *P.scene;
    const Light& light = S.lights[S.mat_light_id[S.tri_mat[h.tri]]];
    const LocalGeometry geom = getLocalGeometry(S, h.tri, h.bu, h.bv);
    lightSample light_sample;
    light_sample.ReverseSample(P, light, geom.UV);
    float t_hit = h.t_hit;
    float3 ray_direction = h.ray_direction;
    if (dot(prd->ray_direction, light_sample.normal()) <= 0) {
        float MIS_weight = 1;
        if (prd->depth != 0) {
            float pdf_hit = prd->pdf * fabsf(dot(ray_direction, light_sample.normal())) / (t_hit * t_hit);
            float pdf_area = light_sample.pdf;
            MIS_weight = pdf_hit / (pdf_area + pdf_hit);
        }
        prd->result += prd->throughput * light_sample.emission * MIS_weight;
    }
    prd->done = true;
}

// ---------------------------------------------------------------- raygen.cu
inline bool is_invalid(float3 a) {  // ISINVALIDVALUE raygen.cu:43
    return a.x > 100000.0f || std::isnan(a.x) || a.y > 100000.0f || std::isnan(a.y) || a.z > 100000.0f || std::isnan(a.z);
}
inline float4 ToneMap(float3 c, float limit) {  // raygen.cu:50-58
    float luminance = 0.3f * c.x + 0.6f * c.y + 0.1f * c.z;
    float s = 1.0f / (1.0f + 1 * luminance / limit);
    // `c * 1.0f / (...)`: (c*1.0f) / x  == c * (1/x) with vec_math's operator/(float4,float)
    return {c.x * s, c.y * s, c.z * s, 1.0f};
}
inline float3 toSRGB(float3 c) {  // cuda/helpers.h:35-43
    float invGamma = 1.0f / 2.4f;
    float3 p = make_float3(powf(c.x, invGamma), powf(c.y, invGamma), powf(c.z, invGamma));
    return make_float3(c.x < 0.0031308f ? 12.92f * c.x : 1.055f * p.x - 0.055f,
                       c.y < 0.0031308f ? 12.92f * c.y : 1.055f * p.y - 0.055f,
                       c.z < 0.0031308f ? 12.92f * c.z : 1.055f * p.z - 0.055f);
}
inline unsigned char quantizeUnsigned8Bits(float x) {  // cuda/helpers.h:50-55
    x = clampf(x, 0.0f, 1.0f);
    unsigned v = (unsigned)(x * 256.0f);
    return (unsigned char)(v < 255u ? v : 255u);
}
inline uint32_t make_color(float3 c) {  // cuda/helpers.h:57-62
    float3 srgb = toSRGB(clamp3(c, 0.0f, 1.0f));
    return (uint32_t)quantizeUnsigned8Bits(srgb.x) | ((uint32_t)quantizeUnsigned8Bits(srgb.y) << 8) |
           ((uint32_t)quantizeUnsigned8Bits(srgb.z) << 16) | (255u << 24);
}

// camera ray (raygen.cu:100-112 == 332-344)
inline float3 camera_ray(const Params& P, unsigned x, unsigned y, uint32_t& seed) {
    const int subframe_index = (int)P.subframe_index;
    seed = tea<4>(y * P.width + x, (uint32_t)subframe_index);
    float jx, jy;
    if (subframe_index == 0) { jx = 0.5f; jy = 0.5f; }
    else { jx = rnd(seed); jy = rnd(seed); }
    const float dx = 2.0f * (((float)x + jx) / (float)P.width) - 1.0f;
    const float dy = 2.0f * (((float)y + jy) / (float)P.height) - 1.0f;
    return normalize(dx * P.U + dy * P.V + P.W);
}
inline void accumulate(const Params& P, unsigned x, unsigned y, float3 result) {  // raygen.cu:157-169 / 430-442
    const unsigned image_index = y * P.width + x;
    float3 accum_color = result;
    if (P.subframe_index > 0) {
        const float a = 1.0f / (float)(P.subframe_index + 1);
        const float4 prev = P.accum_buffer[image_index];
        accum_color = lerp(make_float3(prev.x, prev.y, prev.z), accum_color, a);
    }
    P.accum_buffer[image_index] = {accum_color.x, accum_color.y, accum_color.z, 1.0f};
    float4 val = ToneMap(accum_color, 1.5f);
    if (P.frame_buffer) P.frame_buffer[image_index] = make_color(make_float3(val.x, val.y, val.z));
}

// __raygen__pinhole (raygen.cu:71-170)
inline void raygen_pinhole(const Params& P, unsigned x, unsigned y) {
    const Scene& S = *P.scene;
    uint32_t seed;
    float3 ray_direction = camera_ray(P, x, y, seed);
    float3 ray_origin = P.eye;
    PayloadRadiance payload;
    payload.seed = seed;
    payload.origin = P.eye;
    payload.ray_direction = ray_direction;
    payload.currentResult = make_float3(0);
    if (P.counters) { P.counters->pixel_samples++; P.counters->eye_paths++; }
    while (true) {
        ray_direction = payload.ray_direction;
        ray_origin = payload.origin;
        Hit h = S.closest_hit(ray_origin, ray_direction, SPCBPT_SCENE_EPSILON, 1e16f, P.counters);
        if (h.tri < 0) {  // __miss__constant_radiance (raygen.cu:687-697)
            payload.done = true;
            payload.currentResult = make_float3(0);
            if (payload.depth == 0 && S.sky.valid) payload.result = payload.throughput * S.sky.color(payload.ray_direction);
        } else {
            HitInfo hi{h.tri, h.t, ray_direction, h.bu, h.bv};
            if (S.tri_is_emitter(h.tri)) closesthit_lightsource(P, &payload, hi);
            else closesthit_radiance(P, &payload, hi);
        }
        if (float3weight(payload.currentResult) > 0.0f) {
            if (S.visibilityTest(payload.vis_pos_A, payload.vis_pos_B, P.counters)) payload.result += payload.currentResult;
            payload.currentResult = make_float3(0);
        }
        if (payload.done || payload.depth > 30) break;
        payload.depth += 1;
    }
    accumulate(P, x, y, payload.result);
}

// init_vertex_from_lightSample / init_lightSubPath_from_lightSample / init_EyeSubpath (raygen.cu:172-231)
inline void init_vertex_from_lightSample(lightSample& ls, BDPTVertex& v) {
    v.position = ls.position;
    v.normal = ls.normal();
    v.flux = ls.emission;
    v.pdf = ls.pdf;
    v.singlePdf = v.pdf;
    v.isOrigin = true;
    v.subspaceId = (short)ls.subspaceId;
    v.depth = 0;
    v.materialId = (short)ls.bindLight->id;
    v.RMIS_pointer = 1;
    v.uv = ls.uv;
    v.type = ls.bindLight->env ? ENV : QUAD;
}
inline void init_lightSubPath_from_lightSample(lightSample& ls, BDPTPath& p) {
    p.clear();
    p.push();
    BDPTVertex& v = p.currentVertex();
    p.nextVertex().singlePdf = ls.dir_pdf;
    init_vertex_from_lightSample(ls, v);
}
inline void init_EyeSubpath(BDPTPath& p, float3 origin, float3 direction) {
    p.push();
    p.currentVertex().position = origin;
    p.currentVertex().flux = make_float3(1.0f);
    p.currentVertex().pdf = 1.0f;
    p.currentVertex().RMIS_pointer = 0;
    p.currentVertex().normal = direction;
    p.currentVertex().isOrigin = true;
    p.currentVertex().depth = 0;
    p.currentVertex().singlePdf = 1.0f;
    p.nextVertex().singlePdf = 1.0f;
}

// traceEyeSubPath / traceLightSubPath + SBT dispatch (cuProg.h:409-461; sutil/Scene.cpp:1642-1691)
inline void trace_subpath(const Params& P, float3 o, float3 d, PayloadBDPTVertex* prd, bool light_side) {
    const Scene& S = *P.scene;
    Hit h = S.closest_hit(o, d, SPCBPT_SCENE_EPSILON, 1e16f, P.counters);
    if (h.tri < 0) {  // __miss__BDPTVertex raygen.cu:699-704
        prd->done = true;
        if (!light_side && P.env_miss_strategy && S.sky.valid) {   // test knob: the eye sub-path sees the sky (an ENV_MISS vertex, like the emitter hit)
            prd->path.push();
            BDPTVertex& Mid = prd->path.currentVertex();
            BDPTVertex& Last = prd->path.lastVertex();
            const float3 Le = S.sky.color(d);
            const float lightPdf = S.sky.pdf(d) / (float)S.lights.size();
            Mid.type = ENV_MISS;
            Mid.normal = -d;
            const float pdf_G = fabsf(dot(Last.normal, d));   // a direction: no 1 / t^2, no cosine at the sky
            if (Last.isOrigin) Mid.flux = Last.flux * pdf_G * Le;
            else Mid.flux = Mid.flux * Last.flux * pdf_G * Le;
            Mid.singlePdf = Mid.singlePdf * pdf_G / fabsf(dot(Last.normal, d));
            Mid.pdf = Last.pdf * Mid.singlePdf;
            Mid.depth = Last.depth + 1;
            Mid.subspaceId = (short)S.sky.getLabel(d);
            if (Mid.depth == 1) Mid.RMIS_pointer = 1.0f;
            else {
                BDPTVertex virtual_light;   // construct_virtual_env_light (rmis.h:314-324)
                virtual_light.type = DIRECTION; virtual_light.flux = Le; virtual_light.pdf = lightPdf; virtual_light.singlePdf = lightPdf;
                virtual_light.normal = -d; virtual_light.RMIS_pointer = 1; virtual_light.subspaceId = Mid.subspaceId;
                Mid.RMIS_pointer = (float)(1.0 / (double)rmis::light_hit_env(P, Last, virtual_light));
            }
        }
        return;
    }
    HitInfo hi{h.tri, h.t, d, h.bu, h.bv};
    if (S.tri_is_emitter(h.tri)) {
        if (light_side) prd->done = true;  // __closesthit__lightSource_subpath hit_program.cu:239-244
        else closesthit_eyeSubpath_LightSource(P, prd, hi);
    } else {
        closesthit_subpath(P, prd, hi, light_side);
    }
}

// binary_sample (cuProg.h:245-264) — bespoke bisection, restated exactly (q9)
inline int binary_sample(const Params& P, const float* cmf, int size, uint32_t& seed, float& pmf, bool count = true) {
    float index = rnd(seed) * 1.0f;
    int mid = size / 2 - 1, l = 0, r = size;
    while (r - l > 1) {
        if (P.counters && count) P.counters->cmf_probes++;
        if (index < cmf[mid]) r = mid + 1;
        else l = mid + 1;
        mid = (l + r) / 2 - 1;
    }
    pmf = l == 0 ? cmf[l] : cmf[l] - cmf[l - 1];
    return l;
}
// What the product's guided search reads on its way to the bisection's bin (executed-order counters only; csrc/device_lib.h
// guide_window): one guide entry, then aligned windows of eight CMF values from the entry in front of the guide's place up to the
// window that holds the first value above u.  `base` = the CMF's place in the array the windows are aligned in.
inline unsigned guided_search_reads(const float* cmf, int size, int base, int guide_place, int answer) {
    (void)cmf; (void)size;
    const int c0 = guide_place > 0 ? guide_place - 1 : 0;
    const int pos = (base + c0) & ~3;
    return 1u + 8u * (unsigned)((base + answer - pos) / 8 + 1);
}
inline const BDPTVertex& sampleSecondStage(const Params& P, int subspaceId, uint32_t& seed, float& sample_pmf) {  // cuProg.h:268-280
    const SubspaceSampler& s = P.sampler;
    int begin_index = s.subspace[subspaceId].jump_bias;
    const int size = s.subspace[subspaceId].size;
    uint32_t peek = seed;
    const float u = rnd(peek);
    const int k = binary_sample(P, s.cmfs + begin_index, size, seed, sample_pmf, !P.count_as_executed);
    if (P.count_as_executed && P.counters && size > 0) {
        // the second-stage guide table (csrc/kernels.hip build_guide): bucket (int)(u * n), place = the first k with cmf[k] > (j / n)(1 - 2^-20)
        const float* cmf = s.cmfs + begin_index;
        int j = (int)(u * (float)size);
        if (j > size - 1) j = size - 1;
        const double tj = (double)j / (double)size * (1.0 - 1.0 / 1048576.0);
        int g = 0, hi = size - 1;
        while (g < hi) { const int m = (g + hi) >> 1; if ((double)cmf[m] > tj) hi = m; else g = m + 1; }
        P.counters->cmf_probes += guided_search_reads(cmf, size, begin_index, g, k);
    }
    int index = k + begin_index;
    return s.LVC[s.jump_buffer[index]];
}
inline const BDPTVertex& uniformSample(const Params& P, uint32_t& seed, float& sample_pmf) {  // cuProg.h:283-289
    const SubspaceSampler& s = P.sampler;
    sample_pmf = (float)(1.0 / s.vertex_count);
    int index = (int)(rnd(seed) * s.vertex_count);
    if (index > s.vertex_count - 1) index = s.vertex_count - 1;  // rnd * n may round up to n in FP32: the reference would read one past jump_buffer
    return s.LVC[s.jump_buffer[index]];
}
inline int sampleFirstStage(const Params& P, int eye_subspace, uint32_t& seed, float& sample_pmf) {  // cuProg.h:290-301
    int begin_index = eye_subspace * SPCBPT_NUM_SUBSPACE;
    uint32_t peek = seed;
    const float u = rnd(peek);
    const int l = binary_sample(P, P.CMFGamma + begin_index, SPCBPT_NUM_SUBSPACE, seed, sample_pmf, !P.count_as_executed);
    if (P.count_as_executed && P.counters) {
        // the first-stage guide table (csrc/capi.hip): 1024 buckets per row, place = the first entry above bucket / 1024; each row is
        // aligned on its own (base 0)
        const float* cmf = P.CMFGamma + begin_index;
        const float t = (float)(int)(u * 1024.0f) / 1024.0f;
        int g = 0, hi = SPCBPT_NUM_SUBSPACE;   // (a non-decreasing row: the executed-order counters are those of a trained or uniform matrix)
        while (g < hi) { const int m = (g + hi) >> 1; if (cmf[m] > t) hi = m; else g = m + 1; }
        P.counters->cmf_probes += guided_search_reads(cmf, SPCBPT_NUM_SUBSPACE, 0, g, l);
    }
    return l;
}

// connectVertex_SPCBPT (raygen.cu:253-303)
// direction_connect_ZGCBPT (raygen.cu:234-252): the light vertex is a direction of the environment map
inline float3 direction_connect_ZGCBPT(const Params& P, const BDPTVertex& a, const BDPTVertex& b) {
    const Scene& S = *P.scene;
    float3 L = make_float3(0.0f);
    float3 connectDir = -b.normal;
    if (dot(a.normal, connectDir) > 0.0f) {
        Pbr mat_a = S.materials[a.materialId];
        mat_a.base_color = a.color;
        float3 f = Eval(mat_a, a.normal, normalize(a.lastPosition - a.position), connectDir) * dot(a.normal, connectDir);
        L = a.flux / a.pdf * f * b.flux / b.pdf * rmis::connection_direction_lightSource(P, a, b);
    }
    if (is_invalid(L)) return make_float3(0.0f);
    return L;
}
// visibilityTest(handle, eyeVertex, lightVertex) (cuProg.h:489-500)
inline bool visibilityTest_vertices(const Params& P, const BDPTVertex& eye, const BDPTVertex& light) {
    const Scene& S = *P.scene;
    if (light.is_DIRECTION()) {
        float3 n_pos = -10 * S.sky.r * light.normal + eye.position;
        return S.visibilityTest(eye.position, n_pos, P.counters);
    }
    return S.visibilityTest(eye.position, light.position, P.counters);
}
inline float3 connectVertex_SPCBPT(const Params& P, const BDPTVertex& a, const BDPTVertex& b) {
    const Scene& S = *P.scene;
    if (b.is_DIRECTION()) return direction_connect_ZGCBPT(P, a, b);   // raygen.cu:255-258
    float3 connectVec = a.position - b.position;
    float3 connectDir = normalize(connectVec);
    float G = fabsf(dot(a.normal, connectDir)) * fabsf(dot(b.normal, connectDir)) / dot(connectVec, connectVec);
    float3 LA_DIR = normalize(a.lastPosition - a.position);
    float3 LB_DIR = normalize(b.lastPosition - b.position);
    float3 fa, fb;
    Pbr mat_a = S.materials[a.materialId];
    mat_a.base_color = a.color;
    fa = Eval(mat_a, a.normal, -connectDir, LA_DIR) / (mat_a.brdf ? fabsf(dot(a.normal, connectDir)) : 1.0f);   // raygen.cu:271
    if (!b.isOrigin) {
        Pbr mat_b = S.materials[b.materialId];
        mat_b.base_color = b.color;
        fb = Eval(mat_b, b.normal, connectDir, LB_DIR) / (mat_b.brdf ? fabsf(dot(b.normal, connectDir)) : 1.0f);   // raygen.cu:278
    } else {
        if (dot(b.normal, -connectDir) > 0.0f) fb = make_float3(0.0f);
        else fb = make_float3(1.0f);
    }
    float3 contri = a.flux * b.flux * fa * fb * G;
    float pdf = a.pdf * b.pdf;
    float3 ans = contri / pdf * (b.depth == 0 ? rmis::connection_lightSource(P, a, b) : rmis::general_connection(P, a, b));
    if (is_invalid(ans)) return make_float3(0.0f);
    return ans;
}
inline float3 lightStraghtHit(BDPTVertex& a) {  // raygen.cu:305-317
    float3 ans = a.flux / a.pdf / a.RMIS_pointer;
    if (is_invalid(ans)) return make_float3(0.0f);
    return ans;
}

// __raygen__SPCBPT (raygen.cu:319-443); returns the pixel-sample radiance
inline float3 spcbpt_sample(const Params& P, unsigned x, unsigned y) {
    const Scene& S = *P.scene;
    uint32_t seed;
    float3 ray_direction = camera_ray(P, x, y, seed);
    float3 ray_origin = P.eye;
    float3 result = make_float3(0);
    PayloadBDPTVertex payload;
    payload.clear();
    payload.seed = seed;
    payload.ray_direction = ray_direction;
    payload.origin = ray_origin;
    init_EyeSubpath(payload.path, ray_origin, ray_direction);
    if (P.counters) { P.counters->pixel_samples++; P.counters->eye_paths++; }
    while (true) {
        ray_direction = payload.ray_direction;
        ray_origin = payload.origin;
        if (payload.done || payload.depth > 50) break;
        int begin_depth = payload.path.size;
        trace_subpath(P, ray_origin, ray_direction, &payload, false);
        if (payload.path.size == begin_depth) break;
        payload.depth += 1;
        if (payload.path.hit_lightSource()) {
            result += lightStraghtHit(payload.path.currentVertex());
            break;
        }
        BDPTVertex& eye_subpath = payload.path.currentVertex();
        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
            int light_id = 0;
            float pmf_firstStage = 1;
            float pmf_secondStage;
            if (P.uniform_lvc && P.sampler.vertex_count == 0) continue;
            if (!P.uniform_lvc) {
                if (P.light_tree) light_id = sampleFirstStage(P, eye_subpath.subspaceId, payload.seed, pmf_firstStage);
                if (P.sampler.subspace[light_id].size == 0) continue;
            }
            const BDPTVertex& light_subpath = P.uniform_lvc ? uniformSample(P, payload.seed, pmf_secondStage)
                                                            : sampleSecondStage(P, light_id, payload.seed, pmf_secondStage);
            if (P.counters) P.counters->connections++;
            if (P.skip_null_connections) {
                if (light_subpath.is_DIRECTION()) {   // direction_connect_ZGCBPT contributes only with the sky above the eye vertex's surface
                    if (!(dot(eye_subpath.normal, -light_subpath.normal) > 0.0f)) continue;
                } else {
                    const float3 cd = normalize(eye_subpath.position - light_subpath.position);
                    if (dot(eye_subpath.normal, -cd) <= 0.0f || dot(light_subpath.normal, cd) < 0.0f) continue;
                }
            }
            if (visibilityTest_vertices(P, eye_subpath, light_subpath)) {
                float pmf = P.sampler.path_count * pmf_secondStage * pmf_firstStage;
                float3 res = connectVertex_SPCBPT(P, eye_subpath, light_subpath) / pmf;
                if (!is_invalid(res)) result += res / (float)SPCBPT_CONNECTION_N;
            }
        }
    }
    result += payload.result;
    return result;
}
inline void raygen_SPCBPT(const Params& P, unsigned x, unsigned y) { accumulate(P, x, y, spcbpt_sample(P, x, y)); }

// ---- __raygen__SPCBPT_no_rmis (raygen.cu:445-606): the same sampler with classic full-path MIS weights instead of the recursive
// ones.  Defined in the reference but bound to no program group (Scene::switchRaygen knows four names); restated because it is
// an INDEPENDENT estimator of the same image: its weights come from contriCompute / pdfCompute / MISWeight_SPCBPT over the whole
// path (cuProg.h:901-1105), none of rmis.h.  MAX_PATH_LENGTH_FOR_MIS = 20 bounds the path (longer paths are dropped: a small,
// deliberate bias of the variant).
static constexpr int MAX_PATH_LENGTH_FOR_MIS = 20;
inline float3 contriCompute(const Params& P, const BDPTVertex* path, int path_size) {  // cuProg.h:901-934
    const Scene& S = *P.scene;
    float3 throughput = make_float3(1);
    const BDPTVertex& light = path[path_size - 1];
    const BDPTVertex& lastMidPoint = path[path_size - 2];
    float3 lightLine = lastMidPoint.position - light.position;
    float3 lightDirection = normalize(lightLine);
    float lAng = dot(light.normal, lightDirection);
    if (lAng < 0.0f) return make_float3(0.0f);
    float3 Le = light.flux * lAng;
    throughput = throughput * Le;
    for (int i = 1; i < path_size; i++) {
        float3 line = path[i].position - path[i - 1].position;
        throughput = throughput / dot(line, line);
    }
    for (int i = 1; i < path_size - 1; i++) {
        const BDPTVertex& midPoint = path[i];
        float3 lastDirection = normalize(path[i - 1].position - midPoint.position);
        float3 nextDirection = normalize(path[i + 1].position - midPoint.position);
        Pbr mat = S.materials[midPoint.materialId];
        mat.base_color = midPoint.color;
        throughput = throughput * (fabsf(dot(midPoint.normal, lastDirection)) * fabsf(dot(midPoint.normal, nextDirection)) *
                                   Eval(mat, midPoint.normal, lastDirection, nextDirection));
    }
    return throughput;
}
// the factors shared by pdfCompute and MISWeight_SPCBPT: eye sub-path of `eyePathLength` vertices traced from the camera
inline float eye_side_pdf(const Params& P, const BDPTVertex* path, int eyePathLength) {  // cuProg.h:972-994 = 1004-1027
    const Scene& S = *P.scene;
    float pdf = 1.0f;
    for (int i = 1; i < eyePathLength; i++) {
        float3 line = path[i].position - path[i - 1].position;
        float3 lineDirection = normalize(line);
        pdf *= 1.0f / dot(line, line) * fabsf(dot(path[i].normal, lineDirection));
    }
    for (int i = 1; i < eyePathLength - 1; i++) {
        const BDPTVertex& midPoint = path[i];
        float3 lastDirection = normalize(path[i - 1].position - midPoint.position);
        float3 nextDirection = normalize(path[i + 1].position - midPoint.position);
        Pbr mat = S.materials[midPoint.materialId];
        mat.base_color = midPoint.color;
        float rr_rate = fmaxf3(midPoint.color);   // NOT clamped to MIN_RR_RATE here, unlike the walk (hit_program.cu:325-328): kept as written
        pdf *= Pdf(mat, midPoint.normal, lastDirection, nextDirection) * rr_rate;
    }
    return pdf;
}
inline float pdfCompute(const Params& P, const BDPTVertex* path, int path_size, int strategy_id) {  // cuProg.h:935-996
    const Scene& S = *P.scene;
    int eyePathLength = strategy_id;
    int lightPathLength = path_size - eyePathLength;
    float pdf = 1.0f;
    if (lightPathLength > 0) pdf *= path[path_size - 1].pdf;
    if (lightPathLength > 1) {
        const BDPTVertex& light = path[path_size - 1];
        const BDPTVertex& lastMidPoint = path[path_size - 2];
        float3 lightDirection = normalize(lastMidPoint.position - light.position);
        pdf = (float)((double)pdf * ((double)fabsf(dot(lightDirection, light.normal)) / M_PI));   // `pdf *= abs(..) / M_PI` with the double literal
        for (int i = 1; i < lightPathLength; i++) {
            const BDPTVertex& midPoint = path[path_size - i - 1];
            float3 line = midPoint.position - path[path_size - i].position;
            float3 lineDirection = normalize(line);
            pdf = (float)((double)pdf * (1.0 / (double)dot(line, line) * (double)fabsf(dot(midPoint.normal, lineDirection))));   // `1.0 / dot(..)`: double
        }
        for (int i = 1; i < lightPathLength - 1; i++) {
            const BDPTVertex& midPoint = path[path_size - i - 1];
            float3 lastDirection = normalize(path[path_size - i].position - midPoint.position);
            float3 nextDirection = normalize(path[path_size - i - 2].position - midPoint.position);
            Pbr mat = S.materials[midPoint.materialId];
            mat.base_color = midPoint.color;
            float rr_rate = fmaxf3(midPoint.color);
            pdf *= Pdf(mat, midPoint.normal, lastDirection, nextDirection) * rr_rate;
        }
    }
    return pdf * eye_side_pdf(P, path, eyePathLength);
}
inline float MISWeight_SPCBPT(const Params& P, const BDPTVertex* path, int path_size, int strategy_id) {  // cuProg.h:998-1105
    const Scene& S = *P.scene;
    if (strategy_id <= 1 || strategy_id == path_size) return pdfCompute(P, path, path_size, strategy_id);
    int eyePathLength = strategy_id;
    int lightPathLength = path_size - eyePathLength;
    float pdf = eye_side_pdf(P, path, eyePathLength);
    float3 light_contri = make_float3(1.0f);
    if (lightPathLength > 0) light_contri = light_contri * path[path_size - 1].flux;
    if (lightPathLength > 1) {
        const BDPTVertex& lastMidPoint = path[path_size - 2];
        for (int i = 1; i < lightPathLength; i++) {
            const BDPTVertex& midPoint = path[path_size - i - 1];
            float3 line = midPoint.position - path[path_size - i].position;
            float3 lineDirection = normalize(line);
            // `1.0 / dot(line, line) * |n_mid . dir| * |n_lastMid . dir|` in double, times the float3 -- note lastMidPoint, not lastPoint, as written
            double g = 1.0 / (double)dot(line, line) * (double)fabsf(dot(midPoint.normal, lineDirection)) * (double)fabsf(dot(lastMidPoint.normal, lineDirection));
            light_contri = light_contri * (float)g;   // operator*=(float3&, float): the double narrows first
        }
        for (int i = 1; i < lightPathLength - 1; i++) {
            const BDPTVertex& midPoint = path[path_size - i - 1];
            float3 lastDirection = normalize(path[path_size - i].position - midPoint.position);
            float3 nextDirection = normalize(path[path_size - i - 2].position - midPoint.position);
            Pbr mat = S.materials[midPoint.materialId];
            mat.base_color = midPoint.color;
            light_contri = light_contri * Eval(mat, midPoint.normal, lastDirection, nextDirection);
        }
    }
    float3 position = path[strategy_id - 1].position, normal = path[strategy_id - 1].normal;
    float3 dir = normalize(path[strategy_id - 2].position - path[strategy_id - 1].position);
    int eye_subspace_id = getLabel(P, position, normal, dir, false);
    int light_subspace_id;
    if (strategy_id == path_size - 1) light_subspace_id = path[strategy_id].subspaceId;
    else {
        position = path[strategy_id].position; normal = path[strategy_id].normal;
        dir = normalize(path[strategy_id + 1].position - path[strategy_id].position);
        light_subspace_id = getLabel(P, position, normal, dir, true);
    }
    return pdf * float3weight(connectRate_SOL(P, eye_subspace_id, light_subspace_id, light_contri));
}
inline float3 eval_path(const Params& P, const BDPTVertex* path, int path_size, int strategy_id) {  // raygen.cu:445-464
    float pdf = pdfCompute(P, path, path_size, strategy_id);
    float3 contri = contriCompute(P, path, path_size);
    float MIS_weight_not_normalize = MISWeight_SPCBPT(P, path, path_size, strategy_id);
    float MIS_weight_dominator = 0.0f;
    for (int i = 2; i <= path_size; i++) MIS_weight_dominator += MISWeight_SPCBPT(P, path, path_size, i);
    float3 ans = contri / pdf * (MIS_weight_not_normalize / MIS_weight_dominator);
    if (is_invalid(ans)) return make_float3(0.0f);
    return ans;
}
inline float3 spcbpt_no_rmis_sample(const Params& P, unsigned x, unsigned y) {  // raygen.cu:465-590
    const Scene& S = *P.scene;
    uint32_t seed;
    float3 ray_direction = camera_ray(P, x, y, seed);
    float3 ray_origin = P.eye;
    float3 result = make_float3(0);
    PayloadBDPTVertex payload;
    payload.clear();
    payload.seed = seed;
    payload.ray_direction = ray_direction;
    payload.origin = ray_origin;
    init_EyeSubpath(payload.path, ray_origin, ray_direction);
    BDPTVertex pathBuffer[MAX_PATH_LENGTH_FOR_MIS];
    int buffer_size = 0;
    pathBuffer[buffer_size] = payload.path.currentVertex(); buffer_size++;
    if (P.counters) { P.counters->pixel_samples++; P.counters->eye_paths++; }
    while (true) {
        ray_direction = payload.ray_direction;
        ray_origin = payload.origin;
        if (payload.done || payload.depth > 50) break;
        int begin_depth = payload.path.size;
        trace_subpath(P, ray_origin, ray_direction, &payload, false);
        if (payload.path.size == begin_depth) break;
        payload.depth += 1;
        pathBuffer[buffer_size] = payload.path.currentVertex(); buffer_size++;
        if (payload.path.hit_lightSource()) {
            lightSample light_sample;
            light_sample.ReverseSample(P, S.lights[payload.path.currentVertex().materialId], payload.path.currentVertex().uv);
            BDPTVertex light_vertex;
            init_vertex_from_lightSample(light_sample, light_vertex);
            pathBuffer[buffer_size - 1] = light_vertex;
            result += eval_path(P, pathBuffer, buffer_size, buffer_size);
            break;
        }
        if (buffer_size >= MAX_PATH_LENGTH_FOR_MIS) break;
        BDPTVertex& eye_subpath = payload.path.currentVertex();
        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
            int light_id = 0;
            float pmf_firstStage = 1;
            if (P.light_tree) light_id = sampleFirstStage(P, eye_subpath.subspaceId, payload.seed, pmf_firstStage);
            if (P.sampler.subspace[light_id].size == 0) continue;
            float pmf_secondStage;
            const BDPTVertex& light_subpath = sampleSecondStage(P, light_id, payload.seed, pmf_secondStage);
            if (P.counters) P.counters->connections++;
            if ((buffer_size + light_subpath.depth + 1 <= MAX_PATH_LENGTH_FOR_MIS) &&
                S.visibilityTest(eye_subpath.position, light_subpath.position, P.counters)) {
                float pmf = P.sampler.path_count * pmf_secondStage * pmf_firstStage;
                int origin_buffer_size = buffer_size;
                const BDPTVertex* light_ptr = &light_subpath;
                while (true) {   // the light sub-path sits in consecutive LVC slots, origin first: walk it backwards
                    pathBuffer[buffer_size] = *light_ptr; buffer_size++;
                    if (light_ptr->depth == 0) break;
                    light_ptr--;
                }
                float3 res = eval_path(P, pathBuffer, buffer_size, origin_buffer_size) / pmf;
                buffer_size = origin_buffer_size;
                if (!is_invalid(res)) result += res / (float)SPCBPT_CONNECTION_N;
            }
        }
    }
    return result;
}
inline void raygen_SPCBPT_no_rmis(const Params& P, unsigned x, unsigned y) { accumulate(P, x, y, spcbpt_no_rmis_sample(P, x, y)); }

// __raygen__lightTrace (raygen.cu:620-685) for one core (= one OptiX thread)
inline void raygen_lightTrace(const Params& P, int launch_index) {
    const LightTraceParams& lt = P.lt;
    const Scene& S = *P.scene;
    uint32_t seed = tea<4>((uint32_t)launch_index, (uint32_t)lt.launch_frame);
    PayloadBDPTVertex payload;
    payload.seed = lt.decorrelate_bsdf_stream ? tea<4>((uint32_t)launch_index ^ 0x80000000u, (uint32_t)lt.launch_frame) : seed;
    unsigned bufferBias = (unsigned)lt.core_padding * (unsigned)launch_index;
    unsigned lightVertexCount = 0, lightPathCount = 0;
    auto push = [&](BDPTVertex& v) {  // pushVertexToLVC raygen.cu:613-619
        lt.ans[lightVertexCount + bufferBias] = v;
        lt.ans[lightVertexCount + bufferBias].path_id = (uint32_t)launch_index * (uint32_t)lt.M_per_core + lightPathCount;
        lt.validState[lightVertexCount + bufferBias] = 1;
        lightVertexCount++;
        if (P.counters) P.counters->lvc_stores++;
    };
#define ORC_CHECK_LIGHT_BUFFER if (!(lightVertexCount < (unsigned)lt.core_padding)) break;
    while (true) {
        payload.clear();
        int n_lights = (int)S.lights.size();
        int light_id = clampi((int)floorf(rnd(seed) * n_lights), 0, n_lights - 1);
        const Light& light = S.lights[light_id];
        lightSample light_sample;
        light_sample.sample(P, light, seed);
        light_sample.traceMode(P, seed);
        float3 ray_direction = light_sample.trace_direction();
        float3 ray_origin = light_sample.position;
        init_lightSubPath_from_lightSample(light_sample, payload.path);
        if (P.counters) P.counters->light_paths++;
        push(payload.path.currentVertex());
        ORC_CHECK_LIGHT_BUFFER;
        while (true) {
            int begin_depth = payload.path.size;
            trace_subpath(P, ray_origin, ray_direction, &payload, true);
            if (payload.path.size > begin_depth) {
                push(payload.path.currentVertex());
                ORC_CHECK_LIGHT_BUFFER;
            }
            ray_direction = payload.ray_direction;
            ray_origin = payload.origin;
            if (payload.done || payload.depth > 50) break;
            payload.depth += 1;
        }
        lightPathCount++;
        if (lightPathCount >= (unsigned)lt.M_per_core) break;
        ORC_CHECK_LIGHT_BUFFER;
    }
#undef ORC_CHECK_LIGHT_BUFFER
    for (int i = (int)lightVertexCount; i < lt.core_padding; i++) lt.validState[i + bufferBias] = 0;
}

// MyThrustOp::LVC_Process (cuda_thrust/device_thrust.cu:241-332)
inline void LVC_Process(Params& P, SamplerStorage& st) {
    const LightTraceParams& lt = P.lt;
    const int countRange = lt.get_element_count();
    SubspaceSampler& sampler = P.sampler;
    std::vector<int> num(SPCBPT_NUM_SUBSPACE, 0);
    std::vector<float> Qs(SPCBPT_NUM_SUBSPACE, 0.0f);
    std::vector<std::vector<int>> jump(SPCBPT_NUM_SUBSPACE);
    std::vector<std::vector<float>> pmf(SPCBPT_NUM_SUBSPACE);
    std::vector<std::vector<double>> pmfd(SPCBPT_NUM_SUBSPACE);
    int valid_count = 0, path_count = 0;
    for (int i = 0; i < countRange; i++) {
        if (!lt.validState[i]) continue;
        valid_count++;
        const BDPTVertex& v = lt.ans[i];
        if (v.depth == 0) path_count++;
        float res = float3weight(v.flux) / v.pdf;  // LVCSubspaceInfoCopy 191-212
        res = std::isinf(res) ? 0 : res;
        float w = std::isnan(res) ? 0 : res;
        int s = v.subspaceId;
        num[s] += 1;
        Qs[s] += w;
        jump[s].push_back(i);
        pmf[s].push_back(w);
        if (pmf[s].size() > 1) pmf[s][pmf[s].size() - 1] += pmf[s][pmf[s].size() - 2];
        if (P.cmf_double) pmfd[s].push_back((pmfd[s].empty() ? 0.0 : pmfd[s].back()) + (double)w);
    }
    st.cmfs.resize(valid_count);
    st.jump_buffer.resize(valid_count);
    st.subspace.resize(SPCBPT_NUM_SUBSPACE);
    int acc = 0, jump_bias = 0;
    for (int i = 0; i < SPCBPT_NUM_SUBSPACE; i++) {
        Subspace& ss = st.subspace[i];
        ss.id = i;
        ss.jump_bias = jump_bias;
        ss.size = (int)jump[i].size();
        ss.sum_pmf = Qs[i];
        ss.Q = 0;
        jump_bias += ss.size;
        for (int j = 0; j < ss.size; j++) {
            st.jump_buffer[acc] = jump[i][j];
            st.cmfs[acc] = pmf[i][j] / ss.sum_pmf;  // q11: NaN when sum_pmf == 0 — kept as in the reference
            if (P.cmf_double) {
                const double tot = pmfd[i].back();
                st.cmfs[acc] = tot > 0.0 ? (float)(pmfd[i][j] / tot) : (float)(j + 1) / (float)ss.size;
                if (j == ss.size - 1) { st.cmfs[acc] = 1.0f; ss.sum_pmf = (float)tot; }
            }
            acc++;
        }
    }
    sampler.vertex_count = valid_count;
    sampler.path_count = path_count;
    sampler.LVC = lt.ans;
    sampler.subspace = st.subspace.data();
    sampler.cmfs = st.cmfs.data();
    sampler.jump_buffer = st.jump_buffer.data();
}

}  // namespace orc

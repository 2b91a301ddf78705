// Device/host shared data layout of the MI355X SPCBPT hot path (HBM-resident).
// All records are 16-byte aligned so every lane fetch is a global_load_dwordx4.
#pragma once
#include <stdint.h>

#include "../../include/spcbpt.h"

namespace spc {

// ---- software LBVH ---------------------------------------------------------
// One node = 64 B = 4 x float4: a 4-wide node folded from the binary tree (lbvh.cpp) with its child boxes quantised to
// 8 bits per plane relative to the node's own box (Ylitie et al. 2017 style).  A node visit is a divergent gather and the
// L1/TA path delivers ~16 B per clock per CU for those (tools/micro/gather_bench.hip: 74 G visits/s with 128-B nodes,
// 126 G visits/s with 64-B nodes), so halving the node is worth far more than the ~30 extra VALU ops of decoding it.
//   q0 = origin.xyz (f32), w = biased scale exponents ex | ey << 8 | ez << 16   (plane = origin + q * 2^(e - 127))
//   q1 = qlo.x[4], qlo.y[4], qlo.z[4], qhi.x[4]      (one byte per child, child i in byte i)
//   q2 = qhi.y[4], qhi.z[4], ref[0], ref[1]
//   q3 = ref[2], ref[3], 0, 0
// ref is the traversal-stack word of the child: internal node index, or 0x80000000 | first triangle << 3 | count for a leaf
// (count 1..LEAF_MAX).  An empty slot has ref 0x80000000 (a leaf of zero triangles) and the inverted box qlo = 255, qhi = 0,
// which the sign-aware slab test of device_lib.h cannot hit.  Quantisation rounds lo down and hi up, so a child box only
// grows: the nearest hit found is unchanged.
#ifndef SPC_LEAF_MAX
#define SPC_LEAF_MAX 4   // (experiments: smaller leaves trade triangle steps for node steps -- profiles/r06_experiments.md)
#endif
static const int LEAF_MAX = SPC_LEAF_MAX;
static const int NODE_QUADS = 4;
static const int HOT_NODES = 64;   // lbvh.cpp numbers the nodes of largest surface area 0 .. HOT_NODES - 1 (largest first; node 0 is the root)
static const uint32_t NODE_EMPTY = 0x80000000u;  // decodes to a leaf of zero triangles: harmless even if a ray "hits" the slot

// One triangle = 64 B = 4 x float4 in BVH order; the intersection test reads the
// first three quads (48 B), hit shading reads all four:
//   t0 = (P0.xyz, uv0.x)  t1 = (P1.xyz, uv0.y)  t2 = (P2.xyz, uv1.x)
//   t3 = (uv1.y, uv2.x, uv2.y, as_float(material | emitter << 31))
static const int TRI_QUADS = 4;

struct DMaterial {  // 64 B; MaterialData::Pbr values (cuda/MaterialData.h:82-100)
    float base_color[3];
    float metallic;
    float roughness, specular, specular_tint, subsurface;
    float sheen, sheen_tint, clearcoat, clearcoat_gloss;
    int32_t albedo_tex;  // 0 none, else texture index + 1
    int32_t light_id;    // >= 0 for the emissive pseudo-materials (scene_shift.cpp:92-103)
    int32_t brdf;        // MaterialData::Pbr::brdf (MaterialData.h:99): the five |N.L| divisions of the bidirectional programs
    int32_t pad;
};
static_assert(sizeof(DMaterial) == 64, "DMaterial");

struct DLight {  // 80 B; Light::QUAD (cuda/Light.h:65-84); u, v are absolute corner points
    float corner[3]; float area;
    float u[3]; int32_t div_level;
    float v[3]; int32_t ss_base;
    float emission[3]; int32_t id;
    float normal[3]; int32_t type;   // 0 = QUAD, 1 = ENV (the environment map: every other field unused)
};
static_assert(sizeof(DLight) == 80, "DLight");

// params.sky (envInfo, optixPathTracer.h:98-137): the environment map as a light.  `tex` is the .hdr raster with its rows flipped
// (HDRLoader::loadTexture), `cmf` the sampling CMF over the raster as read (env_params_setup): device copies owned by the context.
struct DEnv {
    const float* tex;     // width x height float4
    const float* cmf;     // size
    float center[3]; float r;
    int32_t width, height, size, div_level;
    int32_t valid; float project_pdf;   // 1 / (pi r^2)
    int32_t pad[2];
};

struct DTexture {
    const uint32_t* rgba;  // device pointer, RGBA8 packed little-endian
    int32_t width, height;
};

// Subspace-tree node, 16 B = 1 x float4: (mid.xyz, meta).  A leaf has meta = TREE_LEAF_BIT | label; an internal node has
// meta = type << 29 | first child, its eight children sitting in eight consecutive slots (child = first + octant,
// octant = [p.x > mid.x] + 2 [p.y > mid.y] + 4 [p.z > mid.z] as in classTree::tree_index).  One quad per level instead of the
// 56-B reference node (three quads): a trained pair of trees is 2.8 MB instead of 8.4 MB, which is the difference between
// living in the 4 MB L2 of an XCD next to the BVH top and missing it (measured: half of the tree fetches missed L2).
static const int TREE_QUADS = 1;
static const uint32_t TREE_LEAF_BIT = 0x80000000u;

typedef spcbpt_light_vertex LightVertex;  // 96 B = 6 x float4, AoS because it is fetched by random gather
static_assert(sizeof(LightVertex) == 96, "LightVertex");

// First-stage sampling table (three counting levels, 16 x 8 x 8): per eye subspace 16 coarse entries (row[64 k + 63]), 128 middle
// entries (row[8 m + 7]) and the row itself padded to 1024 entries (padding 2.0 > any random number), all 16-B aligned: a level is
// 4 / 2 / 2 quads, and the CONNECTION_N samples of a vertex share the coarse ones (device_lib.h: sample_first_stage3).
static const int CMF2_COARSE = 16, CMF2_MID = 128, CMF2_FINE = 1024, CMF2_ROW = CMF2_COARSE + CMF2_MID + CMF2_FINE;
static const int CMF_GUIDE1 = 1024;   // buckets of the first-stage guide table per row (a power of two: u * CMF_GUIDE1 is exact)

struct DSubspace {  // 16 B
    int32_t jump_bias;
    int32_t size;
    float sum_pmf;
    int32_t pad;
};

enum CounterSlot {
    C_CLOSEST = 0, C_SHADOW, C_NODE, C_TRI, C_VERTEX, C_TEX, C_TREE, C_CMF, C_CONN, C_GQ, C_LVCW, C_PIX, C_EYE, C_LIGHT,
    C_PUBLIC,  // slots above are spcbpt_counters; the rest are wave-clock totals of the megakernel phases (>> 4), developer only
    C_T_REGEN = C_PUBLIC, C_T_CLOSEST, C_T_SHADE, C_T_POOL, C_T_CONNECT,
    C_U_NODE_SLOTS, C_U_NODE_LANES, C_U_TRI_SLOTS, C_U_TRI_LANES,  // lane utilisation of the traversal loops: slots = 64 x wave iterations
    C_T_SAMPLE,  // part of C_T_SHADE spent in the two-stage resampling
    C_W_START_MIN, C_W_END_MAX, C_W_END_SUM, C_W_WAVES,  // wall-clock (100 MHz) start/end of the megakernel's waves: tail analysis
    C_U_TAIL_SLOTS, C_U_TAIL_CLOSEST, C_U_TAIL_SHADOW,  // node steps of the pooled pass after the wave's pool ran dry: 64 x iterations, lanes by kind of ray
    C_U_JOB_SLOTS, C_U_JOB_LANES,  // connection evaluations (the job loop of the connect phase): 64 x rounds, jobs
    C_COUNT
};

struct DeviceScene {
    const float* nodes;      // float4 x NODE_QUADS per 4-wide node
    const float* nodes_q;    // the same nodes, one 16-B record per CHILD (device_lib.h: quad tail of trace_pool); null = no quad tail
    const float* tris;       // float4 x TRI_QUADS per triangle (BVH order)
    const int32_t* tri_orig; // BVH order -> caller's triangle index (quad-light triangles follow the scene's)
    const DMaterial* mats;
    const DLight* lights;
    const DTexture* tex;
    int32_t n_lights;        // QUAD lights, then the ENV light if the scene has an environment map
    int32_t n_mats;
    int32_t general;         // != 0: the scene has an environment map or a material with `brdf` set -> the timed kernels' ENV = true forms
    int32_t fan_tail;        // != 0: the shadow rays of the quad tail fan out over the idle quads (device_lib.h: fan_tail)
    int32_t tri_base;        // the PAIR records (lbvh.h: Lbvh::pairs) follow the node records in one allocation: record n_nodes + t of `nodes` is the slot of triangle t
    int32_t pad_tri_base;
    DEnv env;
};

// One frame of a batched eye launch (k_spcbpt<*, BATCH = true>): what differs between the frames that share a tile queue.
struct FrameDesc {  // 64 B
    const spcbpt_light_vertex* lvc;
    const struct DSubspace* subspace;
    const float* cmfs;
    const uint32_t* guide;   // KParams::guide of this frame's sampler
    const int32_t* sampler_counts;
    float* result;       // float4 per pixel: the radiance of this frame's samples (merged in frame order afterwards)
    const spcbpt_light_vertex* lvc_sorted;   // the cache in the sampler's order (KParams::lvc_sorted)
    uint32_t subframe;
    uint32_t pad[1];
};
static_assert(sizeof(FrameDesc) == 64, "FrameDesc");
static const int kMaxBatchFrames = 32;  // the frame id of a published eye vertex travels in 6 bits next to its subspace ids and depth
static_assert(kMaxBatchFrames <= 64, "the frame id of a batched launch is packed into 6 bits (kernels.hip: published eye vertex)");


struct KParams {  // passed by value as the kernel argument block (the MyParams analogue)
    DeviceScene scene;
    float eye[3], U[3], V[3], W[3];
    uint32_t width, height, subframe;
    int32_t row_begin, row_end, row_step;  // 8-row bands: see spcbpt_launch
    float* accum;        // float4 per pixel
    float* result;       // float4 per pixel or null: radiance of THIS subframe, merged into accum/frame by k_film_merge (render
                         // launches of consecutive frames overlap; only the merges are ordered)
    uint32_t* frame;     // RGBA8 per pixel
    // subspace tuple (subspaceMacroInfo)
    const float* eye_tree;
    const float* light_tree;
    const float* Q;
    const float* cmf_gamma;
    const float* cmf_gamma2;  // three-level copy of cmf_gamma for first-stage sampling (CMF2_ROW floats per row, see device_lib.h)
    const float* gamma_q;     // Gamma(e, l) / Q[l] = (cmf_gamma[e][l] - cmf_gamma[e][l - 1]) / Q[l], the quotient gamma_ss evaluates (null: evaluated from
                              // cmf_gamma and Q, the reference-order counting form)
    const uint16_t* cmf_guide1;   // first-stage guide table: row e, bucket b of CMF_GUIDE1 = the first entry of row e above b / CMF_GUIDE1
    // sampler (SubspaceSampler)
    const LightVertex* lvc;
    const LightVertex* lvc_sorted;  // the same vertices in the sampler's order (record i = lvc[jump[i]]): the vertices of a light subspace
                                    // are contiguous, and the eye megakernel fetches the vertex it has drawn by its position in the CMF
                                    // -- no read of `jump` in between (written by the sampler build, kernels.hip k_sb_scatter)
    const DSubspace* subspace;
    const float* cmfs;
    const uint32_t* guide;          // second-stage guide table, one entry per light vertex: entry jump_bias + j of a subspace of n vertices = the
                                    // first place k of its CMF with cmf[k] > (j / n)(1 - 2^-20) -- a lower bound of the bisection's answer for every
                                    // random number u with (int)(u * n) == j (written by the sampler build next to the CMF)
    const int32_t* jump;
    const int32_t* sampler_counts;  // [0] vertex_count, [1] path_count (device-resident: no host round trip)
    int32_t uniform_lvc;            // != 0: "SPCBPT_eye" draws its light vertices with uniformSample (cuProg.h:283-289) = plain LVC-BDPT
    // light pass (LightTraceParams)
    int32_t num_core, core_padding, m_per_core, core_begin, core_count, lt_decorrelate;
    uint32_t launch_frame;
    LightVertex* lvc_scratch;   // core_count * core_padding padded slots
    int32_t* core_counts;       // vertices stored per core
    int32_t* path_counter;      // number of light paths started (= depth-0 vertices) by this launch
    int32_t n_lframes;          // > 0: batched light pass -- the passes of launch frames launch_frame .. launch_frame + n_lframes - 1 share
                                // one core queue (core t of the queue = local core t % core_count of frame t / core_count); frame k stores
                                // to lvc_scratch + k * core_count * core_padding, counts to core_counts + k * (core_count + 1) and paths to path_counter[k]
    // instrumentation / traversal scratch
    uint32_t* work_counter;        // tile queue head of the persistent megakernel (zeroed before each launch)
    uint32_t n_tiles;              // 8x8 pixel tiles in the selected bands
    const FrameDesc* frames;       // batched launch: n_frames frames share the queue (tile t belongs to frame t / n_tiles)
    uint32_t n_frames;
    unsigned long long* counters;  // C_COUNT slots or null
    uint32_t* spill;               // per-thread traversal stack overflow area
    int32_t spill_entries;         // entries per thread in `spill`
    uint32_t* diag;                // [0] traversal-stack entries dropped (deeper than LDS + spill): reported by the host as an error
};

}  // namespace spc

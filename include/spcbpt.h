/*
 * spcbpt.h — C ABI of the MI355X-native SPCBPT hot path.
 *
 * This is the drop-in boundary for the path BASELINE.json's north_star names:
 * light sub-path tracing -> light-vertex-cache (LVC) sampler build -> eye
 * sub-path tracing with subspace classification, two-stage resampling through
 * the subspace sampling matrix, shadow ray, BSDF + recursive-MIS connection,
 * accumulation.  Every entry point cites the reference interface it replaces
 * (paths relative to the reference checkout's src/ directory).
 *
 * The reference dispatches this path by *name* through
 *   sutil::Scene::switchRaygen("pt" | "light trace" | "SPCBPT_eye" | "pretrace")
 *   (sutil/Scene.cpp:1642-1789) followed by optixLaunch(pipeline, 0, d_params,
 *   sizeof(MyParams), sbt, w, h, 1) (OptiXPathTracer/optixPathTracer.cpp:503-512,
 *   535-544, 623-632).  spcbpt_launch() keeps the same four names.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative spcbpt_status; nothing throws across the ABI; all
 * device memory is owned by the context; one context per GPU; calls on one
 * context are not re-entrant.  Launches are asynchronous on the context's HIP
 * stream; spcbpt_sync() waits.
 */
#ifndef SPCBPT_H
#define SPCBPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Compile-time constants of the reference (OptiXPathTracer/optixPathTracer.h:31-39). */
#define SPCBPT_NUM_SUBSPACE 1000             /* NUM_SUBSPACE */
#define SPCBPT_NUM_SUBSPACE_LIGHTSOURCE 200  /* NUM_SUBSPACE_LIGHTSOURCE = int(0.2*NUM_SUBSPACE) */
#define SPCBPT_CONNECTION_N 3                /* CONNECTION_N */
#define SPCBPT_MIN_RR_RATE 0.3f              /* MIN_RR_RATE */
#define SPCBPT_SCENE_EPSILON 1e-3f           /* cuProg.h:39 SCENE_EPSILON */

typedef enum spcbpt_status {
    SPCBPT_OK = 0,
    SPCBPT_ERR_INVALID_ARG = -1,
    SPCBPT_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at create */
    SPCBPT_ERR_HIP = -3,         /* a HIP call failed; see spcbpt_last_error */
    SPCBPT_ERR_UNKNOWN_ALG = -4, /* name is not one of the four launch names */
    SPCBPT_ERR_STATE = -5,       /* call order violated (e.g. SPCBPT_eye before a sampler exists) */
    SPCBPT_ERR_CAPACITY = -6,    /* a caller-provided buffer is too small */
    SPCBPT_ERR_IO = -7           /* a file could not be read / written or is malformed */
} spcbpt_status;

/* Disney-principled material; field set of MaterialData::Pbr
 * (cuda/MaterialData.h:82-100).  albedo_tex is 0 for "none", else texture
 * index + 1 (OptiXPathTracer/scene_shift.cpp:76-79 uses the same 1-based id).
 * brdf is MaterialData::Pbr::brdf (MaterialData.h:99; `brdf <int>` of a .scene
 * material block, sceneLoader.cpp:107 -> scene_shift.cpp:75): nonzero makes
 * the bidirectional programs divide the BSDF value by |N.L| at the five
 * un-guarded ternaries hit_program.cu:286, 384, raygen.cu:271, 278 and
 * rmis.h:105 (the `#ifdef BRDF` branches of Eval/Pdf are dead).  "pt"
 * (hit_program.cu:439-552) has no such division. */
typedef struct spcbpt_material {
    float base_color[3];
    float metallic;
    float roughness;
    float specular;
    float specular_tint;
    float subsurface;
    float sheen;
    float sheen_tint;
    float clearcoat;
    float clearcoat_gloss;
    int32_t albedo_tex;
    int32_t brdf;
} spcbpt_material;

/* 8-bit RGBA image, row 0 first, sampled bilinear + wrap, then pow(c, 2.2)
 * (scene_shift.cpp:40-61, hit_program.cu:182-198, cuProg.h:361-368). */
typedef struct spcbpt_texture {
    const uint8_t* rgba;
    int32_t width;
    int32_t height;
} spcbpt_texture;

/* Quad emitter as the `.scene` light{} block gives it
 * (sceneLoader.cpp:130-193): corner `position`, absolute corner points v1, v2
 * are stored as edges u = v1-position, v = v2-position.  div_level^2 emitter
 * patches become light subspaces allocated downward from id 999
 * (scene_shift.cpp:108-154, cuProg.h:586-589). */
typedef struct spcbpt_quad_light {
    float position[3];
    float u[3];
    float v[3];
    float emission[3];
    int32_t div_level;
} spcbpt_quad_light;

/* Flat triangle soup.  The library appends two triangles per quad light with
 * UVs (0,0)(1,0)(0,1)(1,1) and an emissive pseudo-material after the scene
 * materials, exactly as scene_shift.cpp:92-103, 252-328 does; those triangles
 * are single-sided for path rays and opaque to shadow rays. */
typedef struct spcbpt_scene_desc {
    const float* vertices;        /* 3 * n_vertices */
    const float* texcoords;       /* 2 * n_vertices, may be NULL (zeros, scene_shift.cpp:204-207) */
    int32_t n_vertices;
    const uint32_t* indices;      /* 3 * n_triangles */
    const int32_t* tri_material;  /* n_triangles, index into materials */
    int32_t n_triangles;
    const spcbpt_material* materials;
    int32_t n_materials;
    const spcbpt_texture* textures;
    int32_t n_textures;
    const spcbpt_quad_light* lights;
    int32_t n_lights;
} spcbpt_scene_desc;

/* Octree node of the subspace classifiers; values of classTree::tree_node
 * (decisionTree/classTree_common.h:11-38).  type: 0 position, 1 normal,
 * 2 direction.  leaf != 0 -> label is the subspace id. */
typedef struct spcbpt_tree_node {
    float mid[3];
    int32_t child[8];
    int32_t label;
    int32_t type;
    int32_t leaf;
} spcbpt_tree_node;

/* Launch geometry of the light pass = LightTraceParams
 * (optixPathTracer.h:52-66; reference values num_core 1000, core_padding 800,
 * M_per_core 100 from optixPathTracer.cpp:462-468).  Path index p of core c
 * draws its random numbers from the core's running seed tea<4>(c, frame).
 * The MI355X default is one light path per lane: num_core = M, M_per_core = 1. */
typedef struct spcbpt_light_trace_params {
    int32_t num_core;
    int32_t core_padding;
    int32_t m_per_core;
    int32_t core_begin; /* first core traced by this context (multi-GPU sharding) */
    int32_t core_count; /* number of cores traced by this context; 0 = all */
    /* 0 = reference behaviour: the BSDF stream (payload.seed) starts equal to the light-sampling stream
     * (raygen.cu:625-628), which correlates the first path of every core with its own light sample.
     * Harmless at 100 paths per core (1 % of paths), a measured +1.5 % image bias at one path per core.
     * 1 = the BSDF stream starts from tea<4>(core ^ 0x80000000, frame) instead. */
    int32_t decorrelate_bsdf_stream;
} spcbpt_light_trace_params;

/* One light vertex as exchanged between ranks / checked by tests.  Values of
 * the BDPTVertex fields the connection code reads (BDPTVertex.h:9-70); the
 * device layout is the library's own. 96 bytes. */
typedef struct spcbpt_light_vertex {
    float position[3];
    float pdf;
    float normal[3];
    float single_pdf;
    float flux[3];
    float rmis_pointer;
    float color[3];
    float last_lum;
    float last_position[3];
    float last_normal_projection;
    int16_t material_id;
    int16_t subspace_id;
    int16_t depth;
    int16_t last_zone_id;
    uint32_t path_id;  /* global light path index = core * m_per_core + k */
    /* Cached classification: the vertex's label under the EYE tree + 1, written by the light pass for surface
     * vertices; 0 = not computed (an emitter vertex, or a cache the caller assembled: write 0).  The connection code uses
     * (pad & 0xffff) - 1 when that is a label (1 .. SPCBPT_NUM_SUBSPACE) and re-derives the label by a tree descent for any other value, so a
     * cache imported with a stale or uninitialised word costs time, never a wrong row of Gamma.  The label belongs to the
     * trees the cache was traced under (like subspace_id): trace a new cache after spcbpt_set_subspace. */
    uint32_t pad;
} spcbpt_light_vertex;
/* spcbpt_light_vertex::pad: bits 0-15 the cached eye-tree label + 1 (above); bit 31: the vertex is a direction of the environment
 * map (BDPTVertex::type == ENV: position = its point on the sky disk, normal = minus the sky direction); bit 30: the vertex was hit
 * straight from the environment map (BDPTVertex::isLastVertex_direction, hit_program.cu:412). */
#define SPCBPT_LV_DIRECTION 0x80000000u
#define SPCBPT_LV_LAST_DIRECTION 0x40000000u

/* Per-subspace record of the sampler = struct Subspace (optixPathTracer.h:43-51). */
typedef struct spcbpt_subspace {
    int32_t jump_bias;
    int32_t id;
    int32_t size;
    float sum_pmf;
    float q;
} spcbpt_subspace;

/* Event counters behind the algorithmic-bytes roofline. */
typedef struct spcbpt_counters {
    uint64_t closest_rays;
    uint64_t shadow_rays;
    uint64_t node_visits;
    uint64_t tri_tests;
    uint64_t surface_vertices;   /* closest hits that became path vertices */
    uint64_t textured_hits;
    uint64_t tree_nodes;         /* octree nodes visited by all classifications */
    uint64_t cmf_probes;         /* stage-1 + stage-2 binary-search probes */
    uint64_t connections;        /* light vertices fetched for connection */
    uint64_t gamma_q_reads;
    uint64_t lvc_stores;
    uint64_t pixel_samples;
    uint64_t eye_paths;
    uint64_t light_paths;
} spcbpt_counters;

typedef struct spcbpt_ctx spcbpt_ctx;

/* Replaces LoadScene + LightSource_shift + Scene_shift + sutil::Scene::finalize
 * (optixPathTracer.cpp:729-741; sutil/Scene.cpp:731-739: context, GAS, IAS,
 * modules, program groups, pipeline, SBT).  Uploads the scene and builds the
 * software LBVH.  device = HIP device ordinal. */
int spcbpt_create(const spcbpt_scene_desc* scene, int device, spcbpt_ctx** out);

/* Frees everything the context owns (the reference never frees). */
int spcbpt_destroy(spcbpt_ctx* ctx);

/* Last error text for this context (or for a failed create when ctx is NULL). */
const char* spcbpt_last_error(const spcbpt_ctx* ctx);

/* Replaces handleCameraUpdate: params.eye + Camera::UVWFrame
 * (optixPathTracer.cpp:352-370; sutil/Camera.cpp:34-45). */
int spcbpt_set_camera(spcbpt_ctx* ctx, const float eye[3], const float U[3],
                      const float V[3], const float W[3]);

/* Convenience: derive U,V,W exactly as sutil::Camera::UVWFrame does. */
int spcbpt_set_camera_lookat(spcbpt_ctx* ctx, const float eye[3], const float lookat[3],
                             const float up[3], float fov_y_deg, float aspect);

/* Replaces the accum_buffer allocation + params.width/height
 * (optixPathTracer.cpp:260-265, 319-333).  Clears accumulation. */
int spcbpt_resize(spcbpt_ctx* ctx, int width, int height);

/* Replaces subspaceInfo.{eye_tree, light_tree, Q, CMFGamma} assignment
 * (optixPathTracer.cpp:569-572, 606-607).  q has 1000 entries, cmf_gamma
 * 1000*1000 (row = eye subspace, inclusive CMF over light subspaces).
 * Passing all-NULL installs the minimal valid tuple of SURVEY.md 7 step 8
 * (single-leaf trees, Gamma rows proportional to Q estimated from light
 * passes) instead of the reference's biased null-tree fallback. */
int spcbpt_set_subspace(spcbpt_ctx* ctx,
                        const spcbpt_tree_node* eye_tree, int n_eye,
                        const spcbpt_tree_node* light_tree, int n_light,
                        const float* q, const float* cmf_gamma);

/* Replaces lt_params_setup (optixPathTracer.cpp:462-477). */
int spcbpt_set_light_trace(spcbpt_ctx* ctx, const spcbpt_light_trace_params* p);

/* A fifth launch name, "SPCBPT_no_rmis": the raygen program __raygen__SPCBPT_no_rmis (raygen.cu:445-606) exists in the
 * reference but is bound to no program group, so switchRaygen cannot select it there.  Same call sequence as "SPCBPT_eye"
 * ("light trace" -> spcbpt_build_sampler -> launch); the connections are weighted with classic full-path MIS
 * (contriCompute / pdfCompute / MISWeight_SPCBPT, cuProg.h:901-1105) instead of the recursive weights of rmis.h, and paths
 * longer than MAX_PATH_LENGTH_FOR_MIS = 20 vertices are dropped, as written.  A validation mode (one lane per pixel-sample, the
 * path in scratch memory, O(n^2) per connection): an independent estimator of the image "SPCBPT_eye" renders. */
/* Replaces switchRaygen(name) + optixLaunch (see file header).  name is one of
 * "pt", "light trace", "SPCBPT_eye", "pretrace".  For "pt" and "SPCBPT_eye"
 * frame is params.subframe_index; the image is cut into bands of 8 rows and
 * the rows y in [row_begin, row_end) with ((y/8 - row_begin/8) % row_step) == 0
 * are rendered: (0, height, 1) = whole image; rank r of N passes
 * (8*r, height, N) and gets every N-th band.  row_begin must be a multiple of
 * 8.  For "light trace" frame is lt_params.launch_frame
 * and the row arguments are ignored.  For "pretrace" frame is
 * pr_params.iteration.  Asynchronous. */
int spcbpt_launch(spcbpt_ctx* ctx, const char* name, uint32_t frame,
                  int row_begin, int row_end, int row_step);

/* Replaces MyThrustOp::LVC_Process (cuda_thrust/device_thrust.cu:241-332):
 * builds cmfs / jump_buffer / Subspace[1000] / vertex_count / path_count from
 * the LVC currently held by the context, on the device. Asynchronous. */
int spcbpt_build_sampler(spcbpt_ctx* ctx);

/* Multi-GPU exchange of the compacted LVC shard (no reference counterpart: the
 * reference is single-GPU).  export: device pointer to the
 * shard (array of spcbpt_light_vertex) and a device pointer to its int32
 * count; capacity in vertices.  import: replace the context's LVC by `count`
 * vertices from a device (is_device != 0) or host buffer. */
int spcbpt_lvc_export(spcbpt_ctx* ctx, void** d_vertices, void** d_count, int* capacity);
int spcbpt_lvc_import(spcbpt_ctx* ctx, const void* vertices, int count, int is_device);
/* Capacity, in vertices, of the context's compact light-vertex caches (the reference allocates LVC_MAX_NUM padded slots once,
 * optixPathTracer.cpp:470-475; here the cache exists once per frame in flight -- `sets` buffer sets of capacity x 104 B each --
 * so it is sized from the cache a pass really produces).  Default (0): the first light pass after spcbpt_set_light_trace is
 * traced once more as a probe (host wait, start-up) and the sets hold 2 x its vertex count (scaled to num_core for a rank's
 * share), at most num_core x core_padding.  A later pass that does not fit is cut off and the next spcbpt_sync returns
 * SPCBPT_ERR_CAPACITY.  set_capacity(vertices > 0) fixes the size by hand (allocates at once; the sets never shrink);
 * environment: SPCBPT_LVC_CAPACITY.  INTEGRATION.md lists the resulting HBM footprint. */
int spcbpt_lvc_set_capacity(spcbpt_ctx* ctx, int vertices);
int spcbpt_lvc_get_capacity(spcbpt_ctx* ctx, int* vertices, int* sets);
/* The same exchange without host round trips (libspcbpt_mgpu's RCCL host, include/spcbpt_mgpu.h):
 *   spcbpt_lvc_export_on        hands out the oldest pending shard like spcbpt_lvc_export and makes `hip_stream` (the caller's
 *                               exchange stream) wait for the light pass that fills it -- the host does not.
 *   spcbpt_lvc_import_gathered  `shards` = world x shard_capacity vertices exactly as an all-gather of every rank's shard (padded
 *                               to shard_capacity) leaves them, `counts_all` = world x {vertex_count, path_count} (device int32).
 *                               A device kernel queued on `hip_stream` concatenates the shards in rank order into the set and
 *                               leaves the totals on the device; spcbpt_build_sampler then runs over the upper bound
 *                               min(world x shard_capacity, LVC capacity) with pad keys instead of reading a count back.  A
 *                               shard larger than shard_capacity raises a device flag: the next spcbpt_sync returns
 *                               SPCBPT_ERR_CAPACITY.
 *   spcbpt_film_pack_bands / spcbpt_film_unpack_bands   exchange 2: this rank's 8-row bands (band b with b % world == rank) as one
 *                               contiguous block of ceil(bands / world) x 8 x width float4, and back from the all-gathered
 *                               world x that block into a full width x height float4 image (`out_image`, device). */
int spcbpt_lvc_export_on(spcbpt_ctx* ctx, void* hip_stream, void** d_vertices, void** d_count, int* capacity);
int spcbpt_lvc_import_gathered(spcbpt_ctx* ctx, const void* shards, const void* counts_all, int world, int shard_capacity, void* hip_stream);
/* One exchange per light BATCH (spcbpt_launch_light_batch): export_batch_on packs the shards of the n_frames oldest pending passes
 * (only their filled part) and their count pairs into the caller's contiguous send buffer `send` (n_frames x shard_capacity
 * vertices) / `send_counts` (n_frames x 2 int32) on `hip_stream`, which waits on the device for those passes; after ONE all-gather
 * of each, import_gathered_batch concatenates every frame's shards into that frame's set with one kernel (rank q's block holds its
 * frames one after the other: frame k at (q n_frames + k) x shard_capacity, counts at 2 (q n_frames + k)) and leaves the sets as
 * n_frames single spcbpt_lvc_import_gathered calls would: n_frames spcbpt_build_sampler calls follow. */
int spcbpt_lvc_export_batch_on(spcbpt_ctx* ctx, void* hip_stream, int n_frames, void* send, void* send_counts, int shard_capacity);
int spcbpt_lvc_import_gathered_batch(spcbpt_ctx* ctx, const void* shards, const void* counts_all, int world, int n_frames, int shard_capacity, void* hip_stream);
int spcbpt_film_pack_bands(spcbpt_ctx* ctx, int rank, int world, void* packed, void* hip_stream);
int spcbpt_film_unpack_bands(spcbpt_ctx* ctx, int world, const void* packed_all, void* out_image, void* hip_stream);
int spcbpt_image_size(spcbpt_ctx* ctx, int* width, int* height);
/* The environment map as one more light (row f4; upstream "unfinished", readme.md:29 -- what its live code does is built):
 * env_params_setup (optixPathTracer.cpp:431-461) + the ENV entry of LightSource_shift (scene_shift.cpp:108-153).  `rgba` = width x
 * height RGBA floats as the .hdr file stores them (row 0 = top; spcbpt_hdr_load); the context keeps the row-flipped texture
 * HDRLoader::loadTexture makes and the sampling CMF envMapCMFBuild makes.  Light sub-paths then start on the sky with probability
 * 1 / n_lights (cuProg.h:611-666), "SPCBPT_eye" connects to sky vertices (raygen.cu:234-258, rmis.h:249-280) and "pt" samples the
 * sky by next-event estimation and shows it to primary rays (hit_program.cu:502-518, raygen.cu:687-697); an eye SUB-PATH that
 * leaves the scene never sees it.  center / radius = sky.center / sky.r (upstream: centre and diagonal of the scene
 * box it computes); radius <= 0 or center == NULL: centre and diagonal of the true bounding box.  The quad lights'
 * patch subspaces move up by 100 (their div_level^2 may sum to 100 at most).  Call once, before the first light pass. */
int spcbpt_set_environment(spcbpt_ctx* ctx, const float* rgba, int width, int height, const float* center, float radius);
int spcbpt_get_environment(spcbpt_ctx* ctx, int* width, int* height, float center[3], float* radius, int* n_lights);
/* Radiance .hdr reader = HDRLoader (scene_shift.cpp:334-500): RGBE, flat or new-style RLE scanlines, "-Y h +X w" only.  rgba == NULL:
 * size query.  The fourth float of a texel is 0 (upstream leaves it unset). */
int spcbpt_hdr_load(const char* path, int* width, int* height, float* rgba, size_t capacity_floats);
/* The light-pass geometry in force (spcbpt_set_light_trace with core_count resolved; the defaults before any call). */
int spcbpt_get_light_trace(spcbpt_ctx* ctx, spcbpt_light_trace_params* out);
/* Host copy of the LVC in deterministic order (path_id, depth). */
int spcbpt_lvc_read(spcbpt_ctx* ctx, spcbpt_light_vertex* out, int capacity, int* count);

/* Host copies of the sampler tables (tests, checkpointing). */
int spcbpt_sampler_read(spcbpt_ctx* ctx, spcbpt_subspace* subspace /*1000*/,
                        float* cmfs, int32_t* jump, int capacity,
                        int* vertex_count, int* path_count);

/* accum_buffer (float4 per pixel, row 0 = bottom of the view) and the
 * tone-mapped sRGB frame (raygen.cu:430-442).  Device pointer variant for
 * RCCL exchange. */
int spcbpt_read_accum(spcbpt_ctx* ctx, float* rgba_out);
int spcbpt_read_frame(spcbpt_ctx* ctx, uint8_t* rgba8_out);
int spcbpt_accum_device_ptr(spcbpt_ctx* ctx, void** d_accum);
/* SPCBPT_ERR_STATE while a deferred frame is outstanding (its merge would land in the cleared film). */
int spcbpt_clear_accum(spcbpt_ctx* ctx);
/* The film as of the LAST QUEUED merge (what spcbpt_sync_film waits for): accum (float4 per pixel) and / or the tone-mapped frame
 * (either may be NULL).  Unlike spcbpt_read_accum / spcbpt_read_frame, which wait for every stream of the context, this does not
 * wait for launches queued behind that merge -- a frame being traced ahead (spcbpt_launch_deferred), light passes ahead. */
int spcbpt_read_film(spcbpt_ctx* ctx, float* accum_rgba_out, uint8_t* frame_rgba8_out);

int spcbpt_get_counters(spcbpt_ctx* ctx, spcbpt_counters* out);
int spcbpt_reset_counters(spcbpt_ctx* ctx);
/* Developer aid (no reference counterpart): wave-clock totals the counting build of the "SPCBPT_eye" megakernel spent in
 * its phases since the last reset: [0] regeneration, [1] closest-hit traversal, [2] vertex + resampling, [3] pooled shadow
 * traversal, [4] connection + film (only ratios are meaningful); [5..8] lane utilisation of the traversal loops of all
 * kernels: node-loop slots (64 x wave iterations) and lanes, triangle-loop slots and lanes; [9] lane-clocks of the two-stage
 * resampling (part of [2], summed over the lanes that sample); [10..13] 100 MHz wall clock of the megakernel's waves: earliest
 * start, latest end, sum of ends, number of waves (how long the last waves run alone); [14..16] the tail of the pooled traversal
 * pass, i.e. its node steps after the wave's ray pool ran dry: slots (64 x iterations), lanes still on a closest-hit ray, lanes on
 * a shadow ray; [17..18] the connection evaluations of the connect phase: slots (64 x rounds of its job loop) and jobs. */
int spcbpt_debug_phase_clocks(spcbpt_ctx* ctx, uint64_t out[19]);
/* Hash of the sources this library was built from (csrc/source_hash.py); the Python mirror refuses a stale library. */
const char* spcbpt_build_source_hash(void);
/* The FP32 arithmetic of this library's device code: "ieee" for libspcbpt_hip.so (correctly rounded division and square root,
 * no contraction -- the oracle's operations; what every function-level parity test and the film hashes assume), "approx" for the
 * opt-in libspcbpt_hip_fast.so (hardware reciprocal / square root, as the reference's own `--use_fast_math` build,
 * src/CMakeLists.txt:214; image-level bars only: tests/test_gpu_fast_build.py). */
const char* spcbpt_build_arithmetic(void);
/* sizeof of the structs of this header as the library was COMPILED, in declaration order: material, texture, quad_light,
 * scene_desc, tree_node, light_trace_params, light_vertex, subspace, counters, unit_eye_vertex, pretrace_path,
 * pretrace_node, viewer_state.  A binding in another language (ctypes / cgo / JNI) checks its mirrors against these instead of
 * against literals (tests/test_abi.py).  Returns the number of structs (13) and writes min(capacity, 13) sizes. */
int spcbpt_abi_struct_sizes(int32_t* sizes, int capacity);
/* Developer probe of the HBM part of the traversal stack.  The per-lane stack holds SPC_STACK_LDS (16) entries in LDS; deeper
 * entries go to a per-thread spill area of 3 * bvh_depth - 16 words, which cannot overflow.  _arm fills every area allocated
 * so far with a word no stack entry can hold, _count returns how many words kernels have overwritten since (tests prove the
 * spill path ran) and the entries per thread.  Should a kernel ever drop an entry (SPCBPT_DEBUG_SPILL_ENTRIES=n in the
 * environment at spcbpt_create caps the area, for tests), spcbpt_sync / spcbpt_read_* / spcbpt_trace_* return
 * SPCBPT_ERR_STATE once and name the count: the reference's optixTrace has no such failure mode, a lost subtree is never silent. */
int spcbpt_debug_spill_arm(spcbpt_ctx* ctx);
int spcbpt_debug_spill_count(spcbpt_ctx* ctx, uint64_t* words_written, int* entries_per_thread);

/* "Plain BDPT", the comparator BASELINE config 5 names: with mode = SPCBPT_SAMPLER_UNIFORM the "SPCBPT_eye" launch draws each of
 * its CONNECTION_N light vertices with SubspaceSampler_device::uniformSample (cuProg.h:283-289: uniform over jump_buffer, pmf
 * 1 / vertex_count, one random number) instead of sampleFirstStage + sampleSecondStage; everything else -- shadow ray,
 * connectVertex_SPCBPT, recursive MIS weights (a partition of unity over strategies whatever the sampler) -- is unchanged, so
 * the estimator stays unbiased.  The reference defines uniformSample and never calls it.  Default SPCBPT_SAMPLER_SUBSPACE. */
enum { SPCBPT_SAMPLER_SUBSPACE = 0, SPCBPT_SAMPLER_UNIFORM = 1 };
int spcbpt_set_connection_sampler(spcbpt_ctx* ctx, int mode);

/* Per-function device harness (tests/test_gpu_units.py): evaluates ONE device function of the hot path on n caller-supplied
 * records (one lane each) with the context's scene, subspace tuple and -- for STAGE2 / UNIFORM -- the sampler of the last
 * spcbpt_build_sampler.  Records are arrays of 32-bit words (floats by bit pattern), `in_words` / `out_words` per record:
 *   SPCBPT_UNIT_BSDF     in 24: material(base3, metallic, roughness, specular, specular_tint, subsurface, sheen, sheen_tint, clearcoat,
 *                        clearcoat_gloss) N3 V3 L3 seed pad2        out 12: Sample(N,V;seed)3, seed', Eval(N,V,L)3, Pdf(N,V,L), Eval(N,V,Ls)3, Pdf(N,V,Ls)
 *   SPCBPT_UNIT_TREE     in 10: tree(0 eye, 1 light) position3 normal3 direction3                      out 1: label
 *   SPCBPT_UNIT_STAGE1   in 2: eye subspace, seed        out 6: l, pmf, seed' as the kernels sample (guide table + window) ; l, pmf, seed' by binary_sample
 *   SPCBPT_UNIT_BSEARCH  in 3: offset, size, seed (CMF = aux + offset)                                  out 3: bin, pmf, seed'
 *   SPCBPT_UNIT_STAGE2   in 2: light subspace, seed                                                     out 5: size, bin (-1: empty), LVC slot, pmf, seed'
 *   SPCBPT_UNIT_UNIFORM  in 1: seed                                                                     out 3: LVC slot, pmf, seed'
 *   SPCBPT_UNIT_CONNECT  in 52: eye vertex (spcbpt_unit_eye_vertex, 25) light vertex (spcbpt_light_vertex, 24) pad3   out 4: connectVertex_SPCBPT rgb, RMIS weight
 *   SPCBPT_UNIT_EYE_STEP in 36: last eye vertex (25) NextVertex.flux3 NextVertex.singlePdf seed ray direction3 flags(bit 0: d11) pad2
 *                        out 40: kind (0 miss, 1 surface vertex, 2 emitter front, 3 emitter back), new vertex (25), next direction3,
 *                        NextVertex.flux3, NextVertex.singlePdf, seed', done, emitter radiance3 (lightStraghtHit), t_hit, pad
 * The ray of EYE_STEP starts at the last vertex's position.  Host pointers; returns after the kernel has run. */
typedef struct spcbpt_unit_eye_vertex {   /* the BDPTVertex fields an eye sub-path vertex carries (BDPTVertex.h:9-70) */
    float position[3], normal[3], flux[3], color[3], last_position[3], rmis3[3];
    float pdf, single_pdf, last_normal_projection;
    int32_t material_id, subspace_id, depth, last_zone_id;
} spcbpt_unit_eye_vertex;
enum spcbpt_unit_op { SPCBPT_UNIT_BSDF = 0, SPCBPT_UNIT_TREE = 1, SPCBPT_UNIT_STAGE1 = 2, SPCBPT_UNIT_BSEARCH = 3, SPCBPT_UNIT_STAGE2 = 4,
                      SPCBPT_UNIT_UNIFORM = 5, SPCBPT_UNIT_CONNECT = 6, SPCBPT_UNIT_EYE_STEP = 7 };
int spcbpt_debug_unit(spcbpt_ctx* ctx, int op, const uint32_t* in, int in_words, uint32_t* out, int out_words, int n,
                      const float* aux, int aux_floats);

/* Developer A/B of two traversal schedules on the SAME rays (round 4, csrc/quad_trace.hip): mode 0 = one lane per ray (the loop
 * of the render kernels), mode 1 = four lanes per ray (one coalesced 64-B node fetch per visit, one child box / one leaf
 * triangle per lane), modes 2 / 3 = the same with 2 / 4 rays per quad in flight; all persistent and pool-fed.  rays: n x {origin3, tmin, direction3, tmax}; any != 0: terminate on first hit
 * (visibilityTest semantics, out_visible) else nearest hit with emitter culling (out_t / out_tri / out_uv; the unused outputs may be
 * NULL).  The launch is repeated `repeat` times on device-resident rays; *avg_ms is the HIP-event mean of one launch.  stats (may be
 * NULL): [0] node visits, [1] leaf visits, [2] triangle tests, [3] lane slots (64 x wave iterations), [4] lanes holding a ray in them,
 * from one more, counting launch.  Mode 1 needs 3 x BVH depth <= 64, modes 2 / 3 <= 48 (their per-ray LDS stacks). */
int spcbpt_debug_trace_bench(spcbpt_ctx* ctx, const float* rays, int n, int mode, int any, int repeat, float* out_t, int32_t* out_tri,
                             float* out_uv, int32_t* out_visible, double* avg_ms, uint64_t stats[5]);
/* Event counting in the kernels (off for timed runs).  1: the counting instantiations evaluate in the REFERENCE's order and charge
 * its events (two relabels per connection and one per RMIS update, a ten-probe bisection per first sampling stage): the contract's
 * byte table of SURVEY.md 8(d).  2: the instantiations the timed runs use, with counters -- the events that really execute (labels
 * cached per vertex; one guide entry + eight CMF values per window of either resampling stage, one Gamma / Q value per
 * evaluation): what the roofline fraction is computed from. */
int spcbpt_enable_counters(spcbpt_ctx* ctx, int enabled);

/* Streams.  A context owns two non-blocking HIP streams: the one returned here (hipStream_t as void*) carries "light trace",
 * the sampler build, LVC import/export copies and the preprocessing; render launches ("pt", "SPCBPT_eye") go to a second one,
 * ordered against the first with events, so that the next frame's light pass can run while an eye kernel drains.
 * spcbpt_sync waits for both (the CUDA_SYNC_CHECK() of the reference's loop); spcbpt_sync_light waits only for the light /
 * sampler stream -- what a multi-GPU host needs before it all-gathers the LVC shard (no reference counterpart). */
int spcbpt_stream(spcbpt_ctx* ctx, void** stream);
int spcbpt_sync(spcbpt_ctx* ctx);
int spcbpt_sync_light(spcbpt_ctx* ctx);
/* A frame ahead in the interactive loop (no reference counterpart; optixPathTracer.cpp:791-822 renders and displays strictly in
 * turn).  spcbpt_launch_deferred is spcbpt_launch("pt" | "SPCBPT_eye", ...) WITHOUT the film merge: the kernel renders into a buffer
 * of its own, accum / frame are untouched.  spcbpt_merge_deferred(ctx, 1) queues that merge -- from then on the frame is exactly
 * what spcbpt_launch would have produced -- and (ctx, 0) drops the frame (the camera moved: its samples belong to no image).  One
 * frame may be outstanding; until it is merged or dropped every other render launch (spcbpt_launch "pt" / "SPCBPT_eye" /
 * "SPCBPT_no_rmis", spcbpt_launch_deferred, spcbpt_launch_eye_batch), spcbpt_clear_accum and spcbpt_set_light_ahead return
 * SPCBPT_ERR_STATE -- the film would otherwise take frames out of order (spcbpt_resize drops it).
 * spcbpt_sync_film makes the host wait for the LAST QUEUED film merge only: the frame to display is complete, work queued behind it
 * (the next frame's light pass, sampler build, deferred eye launch) keeps running.  csrc/viewer.cpp builds its default loop on
 * these: frame f+1 is traced while frame f is shown, and every displayed frame is the reference loop's frame. */
int spcbpt_launch_deferred(spcbpt_ctx* ctx, const char* alg, uint32_t subframe_index, int row_begin, int row_end, int row_step);
int spcbpt_merge_deferred(spcbpt_ctx* ctx, int keep);
int spcbpt_sync_film(spcbpt_ctx* ctx);
/* Batched eye launch (no reference counterpart): renders the samplers of the last n_frames spcbpt_build_sampler calls -- one
 * frame each, oldest first, subframe index subframes[k] -- with ONE persistent kernel whose tile queue spans the frames, and
 * merges them into the film in that order.  The result is that of n_frames spcbpt_launch("SPCBPT_eye") calls; the point is the
 * drain phase of the megakernel, which is paid once per launch: a rank's eighth of a sharded frame is about one 8x8 tile per
 * resident wave, i.e. nothing but drain.  n_frames <= 32 and <= the number of samplers built since and still intact
 * (SPCBPT_ERR_STATE otherwise).  SPCBPT_EYE_BATCH = F in the environment at spcbpt_create sizes the ring of sampler buffer
 * sets for batches of F, so that batches in flight, light passes ahead and builds never wait for a set. */
int spcbpt_launch_eye_batch(spcbpt_ctx* ctx, int n_frames, const uint32_t* subframes, int row_begin, int row_end, int row_step);
/* Batched light pass (no reference counterpart; needs spcbpt_set_light_ahead on): the "light trace" launches of launch frames
 * first_frame .. first_frame + n_frames - 1 as ONE persistent kernel whose core queue spans the frames.  Every pass lands in its own
 * buffer set and queues up exactly as n_frames spcbpt_launch("light trace", first_frame + k) calls would have left them -- the caches
 * are bit-identical to those -- so the host goes on with n_frames x (export / import, spcbpt_build_sampler) and one
 * spcbpt_launch_eye_batch.  The point is again the dependent chain: a rank of an 8-GPU job traces 1/8 of the cores per frame, each
 * pass still takes the ~1.2 ms of its longest path, and beside the eye grid they run one after the other; in one queue they cost
 * about one full-size pass.  n_frames <= 32 (SPCBPT_ERR_INVALID_ARG), SPCBPT_ERR_STATE without light-ahead mode. */
int spcbpt_launch_light_batch(spcbpt_ctx* ctx, uint32_t first_frame, int n_frames);
/* Batched sampler build (no reference counterpart): n_builds spcbpt_build_sampler calls -- the n_builds OLDEST queued light
 * passes -- with the kernels of ONE build (the frame in the grid's second dimension).  Same tables, same state afterwards; what
 * goes is the chain of 4 x n_builds small dependent launches in front of a batched eye launch that cannot start before the last
 * of them (0.12 ms per build on the bench scene).  With SPCBPT_SAMPLER_BUILD=hipcub, or for a cache whose counts the host has to
 * read back first, the call is n_builds times spcbpt_build_sampler -- and also when the device cannot hold the batch's scratch
 * (n_builds x the largest item bound among the builds x 16 B; it only grows, spcbpt_lvc_set_capacity and leaving light-ahead mode
 * free it).  1 <= n_builds <= 32 (SPCBPT_ERR_INVALID_ARG). */
int spcbpt_build_sampler_batch(spcbpt_ctx* ctx, int n_builds);
/* Test hook: bytes and frames of that scratch, and how many batches fell back to single builds for want of it
 * (SPCBPT_DEBUG_BATCH_SCRATCH_LIMIT=<bytes> in the environment makes larger requests fail). */
int spcbpt_debug_batch_scratch(spcbpt_ctx* ctx, int64_t* bytes, int* frames, int* fallbacks);
/* Test hook: host copies of the tables the eye kernel samples through (no reference counterpart; csrc/layout.h KParams::guide,
 * cmf_guide1, gamma_q).  guide2: one entry per light vertex of the last sampler build, in the order of spcbpt_sampler_read's cmfs
 * (capacity2 entries at least the vertex count, else SPCBPT_ERR_CAPACITY); guide1: 1000 x 1024 entries; gamma_q: 1000 x 1000.
 * Any of the three may be NULL. */
int spcbpt_debug_read_sampling_tables(spcbpt_ctx* ctx, uint32_t* guide2, int capacity2, uint16_t* guide1, float* gamma_q);

/* Light passes running ahead (multi-GPU host loops; no reference counterpart).  The light pass is a ~1 ms dependent chain
 * however few paths a rank traces, and the LVC exchange makes the host wait for it; with on != 0 the host may launch frame
 * f + 1's "light trace" BEFORE it exchanges and builds frame f's: every light pass queues its buffer set, and
 * spcbpt_lvc_export / spcbpt_lvc_import / spcbpt_sync_light / spcbpt_build_sampler address the OLDEST queued pass (sync_light
 * then waits for that pass only, not for the stream).  Off (default): they address the latest light pass.  Either way a "light
 * trace" launch invalidates the sampler for eye launches (SPCBPT_ERR_STATE until the next spcbpt_build_sampler), as the reference's
 * loop implies.  Switching waits for the context's streams, clears the queue of unbuilt passes and, when switched off, frees the
 * batched build's scratch. */
int spcbpt_set_light_ahead(spcbpt_ctx* ctx, int on);
/* What the context holds of a loop that runs ahead: light-ahead mode, the number of light passes launched but not built yet, whether
 * the tables of the LAST sampler build are still intact (no later pass, import or re-allocation took their buffer set), whether a
 * deferred frame is outstanding.  Any pointer may be NULL. */
int spcbpt_get_pipeline_state(spcbpt_ctx* ctx, int* light_ahead, int* pending_passes, int* sampler_intact, int* deferred_outstanding);
/* Render once more from the sampler built last although a later "light trace" has been launched since (with passes ahead it went to
 * another buffer set): the interactive loop's re-render of a frame whose speculative launch was dropped.  SPCBPT_ERR_STATE if the
 * tables are gone. */
int spcbpt_reuse_sampler(spcbpt_ctx* ctx);
/* spcbpt_lvc_import(..., is_device = 1) reads its source asynchronously (on the light stream, possibly behind light passes launched
 * ahead).  A host that alternates TWO staging buffers calls this before it overwrites one of them: it returns when the import
 * before the previous one -- the last reader of that buffer -- has copied. */
int spcbpt_lvc_import_wait(spcbpt_ctx* ctx);

/* Kernel timing measured with HIP events on the context's stream: average
 * milliseconds per launch of `name` since the last reset, and launch count. */
int spcbpt_kernel_time(spcbpt_ctx* ctx, const char* name, double* avg_ms, int* launches);
int spcbpt_reset_kernel_time(spcbpt_ctx* ctx);
int spcbpt_enable_kernel_timing(spcbpt_ctx* ctx, int enabled);

/* Standalone traversal entry points (parity tests of the software LBVH
 * against the oracle's BVH; they are the building block optixTrace is replaced
 * by: cuProg.h:384-461 closest hit, 463-487 visibilityTest).
 * rays: n * 8 floats (origin xyz, tmin, direction xyz, tmax).
 * closest: out_t[n], out_tri[n] (-1 = miss), out_uv[2n]; cull_emitter_backfaces
 * as OPTIX_RAY_FLAG_CULL_BACK_FACING_TRIANGLES acts on the single-sided quads.
 * any: out_visible[n] = 1 when nothing is hit in (tmin, tmax). */
int spcbpt_trace_closest(spcbpt_ctx* ctx, const float* rays, int n,
                         float* out_t, int32_t* out_tri, float* out_uv);
int spcbpt_trace_any(spcbpt_ctx* ctx, const float* rays, int n, int32_t* out_visible);

/* Training records of the "pretrace" pass: values of TrainData::pathInfo_sample and TrainData::pathInfo_node
 * (optixPathTracer.h:325-383).  begin_ind/end_ind index the node array; label_A holds the eye depth until the trees
 * exist (node_label, device_thrust.cu:554-573); label_B is pre-filled for emitter vertices only. */
typedef struct spcbpt_pretrace_path {
    float contri[3];
    float sample_pdf;
    float fix_pdf;
    int32_t begin_ind;
    int32_t end_ind;
    int32_t choice_id;
    int32_t pixel_id[2];
    int32_t valid;
    int32_t pad;
} spcbpt_pretrace_path; /* 48 bytes */

typedef struct spcbpt_pretrace_node {
    float a_position[3];
    float b_position[3];
    float a_dir[3];
    float b_dir[3];
    float a_normal[3];
    float b_normal[3];
    float peak_pdf;
    int32_t path_id;
    int32_t label_a;
    int32_t label_b;
    int32_t valid;
    int32_t light_source;
} spcbpt_pretrace_node; /* 96 bytes */

/* Replaces preTracer_params_setup (optixPathTracer.cpp:479-490): threads per "pretrace" launch (reference 10000) and
 * node slots per thread (reference PRETRACE_CONN_PADDING = 10, the maximum). */
int spcbpt_set_pretrace(spcbpt_ctx* ctx, int num_core, int padding);

/* The training set accumulated by "pretrace" launches (= neat_paths / neat_conns after valid_sample_gather,
 * device_thrust.cu:457-493): counts, host copies, replacement (tests inject the oracle's records), reset. */
int spcbpt_train_records_count(spcbpt_ctx* ctx, int* n_paths, int* n_nodes);
int spcbpt_train_records_read(spcbpt_ctx* ctx, spcbpt_pretrace_path* paths, int cap_paths,
                              spcbpt_pretrace_node* nodes, int cap_nodes);
int spcbpt_train_records_import(spcbpt_ctx* ctx, const spcbpt_pretrace_path* paths, int n_paths,
                                const spcbpt_pretrace_node* nodes, int n_nodes);
int spcbpt_train_records_clear(spcbpt_ctx* ctx);

/* Host-owned preprocessing = preprocessing() (optixPathTracer.cpp:552-608):
 * pretrace -> reweight -> subspace trees -> Q -> labels -> Gamma_0 -> Adam
 * training -> CMF Gamma.  Installs the result with spcbpt_set_subspace.
 * target_paths / target_q_paths are the reference's 2,000,000 each; smaller
 * values give a coarser but still valid tuple. */
int spcbpt_preprocess(spcbpt_ctx* ctx, int target_paths, int target_q_paths, int train);
/* The stages of spcbpt_preprocess one by one, on the records currently held (tests compare each with the oracle):
 * stage 1 = sample_reweight + both subspace trees, 2 = Q from light passes (needs target_q_paths), 3 = node labels +
 * outlier clean + Gamma_0, 4 = Adam training, 5 = CMF Gamma + install.  image_width is the 10x10-pixel tile pitch of
 * sample_reweight (the reference hard-codes 1920). */
int spcbpt_preprocess_stage(spcbpt_ctx* ctx, int stage, int arg);
/* Intermediate results for tests / checkpoints: Gamma before the CMF transform (row-major 1000x1000). */
int spcbpt_get_gamma(spcbpt_ctx* ctx, float* gamma);

/* Checkpoint files of the subspace tuple in the reference's own text formats (row f4): tree_eye.txt, tree_light.txt
 * (classTree::tree_load, decisionTree/classTree_host.h:15-59), Q.txt (MyThrustOp::load_Q_file,
 * cuda_thrust/device_thrust.cu:3389-3404) and E.txt = Gamma before the CMF transform (load_Gamma_file, 3347-3380).  The
 * reference has only the readers, with their call sites commented out (optixPathTracer.cpp:573-581, 597, 603); the
 * writers emit what those readers parse.  The first three need neither a context nor a GPU.
 *   spcbpt_checkpoint_read: `gamma` (NUM_SUBSPACE^2) is in/out.  As in load_Gamma_file, the columns of the emitter
 *   subspaces (light id >= NUM_SUBSPACE - NUM_SUBSPACE_LIGHTSOURCE) keep the caller's current values when
 *   have_current_gamma != 0; with 0 every entry comes from the file.
 *   spcbpt_gamma_to_cmf = MyThrustOp::Gamma2CMFGamma (3406-3433).
 *   spcbpt_checkpoint_save writes the installed trees and Q plus the Gamma of the last preprocessing run (SPCBPT_ERR_STATE
 *   if the context never preprocessed: a CMF Gamma handed to spcbpt_set_subspace cannot be inverted exactly).
 *   spcbpt_checkpoint_load reads the four files, applies Gamma2CMFGamma and installs the tuple (spcbpt_set_subspace). */
int spcbpt_checkpoint_write(const char* dir, const spcbpt_tree_node* eye_tree, int n_eye,
                            const spcbpt_tree_node* light_tree, int n_light, const float* q, const float* gamma);
int spcbpt_checkpoint_read(const char* dir, spcbpt_tree_node* eye_tree, int* n_eye, int cap_eye,
                           spcbpt_tree_node* light_tree, int* n_light, int cap_light, float* q, float* gamma,
                           int have_current_gamma);
int spcbpt_gamma_to_cmf(const float* gamma, float* cmf_gamma);
int spcbpt_checkpoint_save(spcbpt_ctx* ctx, const char* dir);
int spcbpt_checkpoint_load(spcbpt_ctx* ctx, const char* dir);

/* Texture file -> RGBA8 as the reference's stbi_load(path, &w, &h, &c, STBI_rgb_alpha) delivers it
 * (OptiXPathTracer/scene_shift.cpp:35-40): baseline / progressive JPEG, PNG, binary PPM, chosen by content.  Two-call
 * protocol: rgba == NULL returns the size only.  The scene loaders use the same decoder.  Needs no context, no GPU. */
int spcbpt_image_load(const char* path, int* width, int* height, uint8_t* rgba, size_t capacity_bytes);

/* Row f3 -- the interactive loop without a window: the state machine of optixPathTracer.cpp (GLFW callbacks 121-241,
 * updateState / handleCameraUpdate / handleResize 333-379, initCameraState 661-670, one pass of the render loop 791-822)
 * over sutil::Trackball and sutil::Camera.  A front end forwards its window events one to one (the arguments are GLFW's:
 * button 0 left / 1 right, action 1 press / 2 repeat / 0 release, key codes ESCAPE 256, SPACE 32, C 67, P 80, W 87) and
 * calls spcbpt_viewer_frame once per displayed subframe; tools/spcbpt_viewer.cpp replays an event script.  `ctx` may be
 * null: the state machine then runs without launching (tests).  Left drag orbits the eye (LookAtFixed), right drag turns
 * the view (EyeFixed), the wheel zooms, SPACE switches pt <-> SPCBPT_eye and restarts the accumulation, P restarts it on
 * every frame, W walks forward by 0.5 / render_fps, ESCAPE sets should_close. */
typedef struct spcbpt_viewer spcbpt_viewer;
typedef struct spcbpt_viewer_state {
    float eye[3], lookat[3], up[3];
    float U[3], V[3], W[3];          /* Camera::UVWFrame at the current window aspect */
    float fov_y, aspect;
    int32_t width, height;
    uint32_t subframe_index;         /* index the NEXT frame will render (0 after a camera / size / algorithm change) */
    int32_t alg_id;                  /* 0 "pt", 1 "SPCBPT_eye" (render_alg, optixPathTracer.cpp:91) */
    int32_t should_close, one_frame_render_only, camera_changed;
    float render_fps;
} spcbpt_viewer_state;
int spcbpt_viewer_create(spcbpt_ctx* ctx, const float eye[3], const float lookat[3], const float up[3], float fov_y,
                         int width, int height, spcbpt_viewer** out);
/* Either order of tear-down is safe (round 6): spcbpt_destroy(ctx) tells the context's live viewers, which go on as state machines
 * without a context (events and frames still advance their state, nothing is launched); destroying the viewer first hands the
 * context back as spcbpt_viewer_create found it. */
void spcbpt_viewer_destroy(spcbpt_viewer* v);
int spcbpt_viewer_mouse_button(spcbpt_viewer* v, int button, int action, double x, double y);
int spcbpt_viewer_cursor_pos(spcbpt_viewer* v, double x, double y);
int spcbpt_viewer_scroll(spcbpt_viewer* v, double xscroll, double yscroll);
int spcbpt_viewer_window_size(spcbpt_viewer* v, int width, int height);
int spcbpt_viewer_iconify(spcbpt_viewer* v, int iconified);
int spcbpt_viewer_key(spcbpt_viewer* v, int key, int action);
int spcbpt_viewer_set_fps(spcbpt_viewer* v, float fps);
/* How far the loop runs ahead of what it shows (the reference has no such modes; every mode shows the SAME frames -- same launch
 * frames, same caches, same images, tests/test_viewer.py):
 *   0  the reference's order: light pass, sampler build, eye launch, device sync, strictly in turn (optixPathTracer.cpp:791-822);
 *   1  with "SPCBPT_eye" the NEXT frame's light pass is launched right after this frame's eye launch, before the sync, so that it
 *      runs beside the eye kernel (light sub-paths do not depend on the camera);
 *   2  (default) the next frame is traced while this one is shown: its sampler build and its eye launch WITHOUT the film merge
 *      (spcbpt_launch_deferred) are queued before spcbpt_viewer_frame returns, which waits for the shown frame's merge only
 *      (spcbpt_sync_film).  The next call merges that frame if nothing it depends on has changed since -- camera, size, algorithm,
 *      subframe restart -- and drops it otherwise; light pass and sampler are kept either way, so the k-th "SPCBPT_eye" frame
 *      always uses the k-th light pass.  Only a call that saw NO event speculates: while the camera is dragged (every call sees a
 *      change) the loop runs as mode 1 -- no frame is queued just to be dropped by the next event -- and the first steady call
 *      starts tracing ahead again.  A steady view costs the eye kernel per displayed frame.
 * Modes 1 and 2 set spcbpt_set_light_ahead on the context; spcbpt_viewer_destroy drops what the viewer queued ahead and restores the
 * mode it found, so a host can go on with the plain loop on the same context (destroy the viewer BEFORE its context: it calls into it).  A host that touches the context BETWEEN two
 * viewer frames (merges / drops the deferred frame, new tuple, sky, cache import) is tolerated: each frame re-validates the viewer's
 * flags against spcbpt_get_pipeline_state.  To DISPLAY a frame read it with spcbpt_read_film (or spcbpt_accum_device_ptr after
 * spcbpt_viewer_frame): spcbpt_read_frame / _accum wait for everything queued, i.e. also for the frame being traced ahead.
 * spcbpt_viewer_set_light_ahead(v, on) = set_pipeline(v, on ? 1 : 0). */
int spcbpt_viewer_set_pipeline(spcbpt_viewer* v, int mode);
int spcbpt_viewer_set_light_ahead(spcbpt_viewer* v, int on);
int spcbpt_viewer_frame(spcbpt_viewer* v);
int spcbpt_viewer_get_state(spcbpt_viewer* v, spcbpt_viewer_state* state);
const char* spcbpt_viewer_alg_name(int alg_id);

/* Read back the installed subspace tuple (checkpoint writer the reference lacks). */
int spcbpt_get_subspace(spcbpt_ctx* ctx,
                        spcbpt_tree_node* eye_tree, int* n_eye, int cap_eye,
                        spcbpt_tree_node* light_tree, int* n_light, int cap_light,
                        float* q, float* cmf_gamma);

/* `.scene` + OBJ ingestion: replaces LoadScene (OptiXPathTracer/sceneLoader.cpp:47-308) and the
 * flattening Scene_shift does (OptiXPathTracer/scene_shift.cpp:32-328).  data_root plays SAMPLES_DIR "/data": mesh and
 * texture paths of the file are relative to it (back-slashes accepted).  The desc points into the handle's storage
 * and stays valid until spcbpt_scene_file_free.  Needs no GPU. */
typedef struct spcbpt_scene_file spcbpt_scene_file;
int spcbpt_scene_file_load(const char* scene_path, const char* data_root, spcbpt_scene_file** out);
/* glTF 2.0 (.gltf with external / base64 buffers, or .glb) into the same handle, read the way the reference's own glTF route
 * reads it (sutil::loadScene + processGLTFNode, sutil/Scene.cpp:119-210, 266-550; never called by the reference app): root
 * nodes = nodes without a parent, transform = parent * matrix^T * T * R * S in fp32, TRIANGLES primitives with POSITION /
 * TEXCOORD_0 / indices, baseColor / metallic / roughness factors, baseColorTexture (binary PPM images only), first perspective
 * camera.  Quad lights come from the caller or from this build's root `extras.spcbpt_quad_lights`.  On failure the message
 * is written to `error` (may be NULL). */
int spcbpt_gltf_load(const char* path, spcbpt_scene_file** out, char* error, int error_capacity);
int spcbpt_scene_file_desc(spcbpt_scene_file* s, spcbpt_scene_desc* desc);
int spcbpt_scene_file_camera(spcbpt_scene_file* s, float eye[3], float lookat[3], float up[3], float* fov_y_deg,
                             int* width, int* height);
/* The scene's environment map: `env_file` of the cameraSetting block (sceneLoader.cpp:242), read like HDRLoader does (width = 0: the
 * scene names none, or it could not be read: see the warnings), and the sky.center / sky.r env_params_setup would derive from
 * the reference's scene box (optixPathTracer.cpp:458-459; SURVEY q7: only the first third of every OBJ shape's vertices enters it)
 * -- to be handed to spcbpt_set_environment.  The pointers stay valid until spcbpt_scene_file_free. */
int spcbpt_scene_file_environment(spcbpt_scene_file* scene, const float** rgba, int* width, int* height, float center[3], float* radius);
const char* spcbpt_scene_file_warnings(spcbpt_scene_file* s);
int spcbpt_scene_file_free(spcbpt_scene_file* s);

/* Scene statistics after create. */
int spcbpt_scene_info(spcbpt_ctx* ctx, int* n_triangles, int* n_bvh_nodes, int* bvh_depth);

#ifdef __cplusplus
}
#endif
#endif /* SPCBPT_H */

// Standalone traversal kernels (BVH parity, optixTrace stand-ins) and the pre-trace kernel <- __raygen__TrainData (raygen.cu:686-776)
// (kernel_config.h maps the kernel files)
#include <hip/hip_runtime.h>

#include "device_lib.h"
#include "eye_walk.h"
#include "kernel_config.h"
#include "kernels.h"

namespace spc {

// ------------------------------------------------------------------------------------------------
// Standalone traversal kernels (parity of the software LBVH against the oracle's BVH)
__global__ __launch_bounds__(BLOCK) void k_trace_closest(const KParams p, const float* __restrict__ rays, int n, float* __restrict__ out_t,
                                                        int* __restrict__ out_tri, float* __restrict__ out_uv) {
    __shared__ uint32_t s_stack[BLOCK * STACK_LDS];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    TravStack<BLOCK, STACK_LDS> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)i, p.diag);
    const float* r = rays + (size_t)i * 8;
    Counts<false> cn;
    HitRec h;
    traverse<false, false>(p.scene, st, mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7], h, cn);
    out_t[i] = h.t;
    out_tri[i] = h.tri >= 0 ? p.scene.tri_orig[h.tri] : -1;
    out_uv[2 * i] = h.u; out_uv[2 * i + 1] = h.v;
}
__global__ __launch_bounds__(BLOCK) void k_trace_any(const KParams p, const float* __restrict__ rays, int n, int* __restrict__ out_visible) {
    __shared__ uint32_t s_stack[BLOCK * STACK_LDS];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    TravStack<BLOCK, STACK_LDS> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)i, p.diag);
    const float* r = rays + (size_t)i * 8;
    Counts<false> cn;
    HitRec h;
    out_visible[i] = traverse<true, false>(p.scene, st, mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7], h, cn) ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// "pretrace": one PT+NEE eye path per lane producing the training records of the sampling-matrix optimisation
// (__raygen__TrainData raygen.cu:751-868, PreTrace_buildPathInfo 708-740, TrainData::nVertex_device cuProg.h:1128-1292).
struct NVertex {  // TrainData::nVertex, the live fields
    f3 position, dir, normal, weight, color;
    float pdf;
    int materialId, label_id, depth;  // materialId < 0: area light source
};
SPC_DEV NVertex nv_from_eye(const EyeVertex& a) {  // nVertex(const BDPTVertex&, eye_side = true)
    NVertex v;
    v.position = a.c.pos; v.normal = a.c.n; v.color = a.c.color; v.materialId = a.c.mat; v.pdf = a.pdf; v.label_id = a.sub;
    v.depth = a.depth;
    v.dir = a.depth == 0 ? mk3(0.0f) : normalize(a.c.lastPos - a.c.pos);
    v.weight = mk3(a.pdf);
    return v;
}
SPC_DEV NVertex nv_from_light(const LightSampleD& ls) {  // nVertex(light BDPTVertex, eye_side = false), depth 0, QUAD
    NVertex v;
    v.position = ls.position; v.normal = ls.normal; v.color = mk3(0.0f); v.materialId = -1; v.pdf = ls.pdf; v.label_id = ls.subspace;
    v.depth = 0; v.dir = mk3(0.0f); v.weight = ls.emission;
    return v;
}
SPC_DEV Pbr nv_mat(const DeviceScene& S, const NVertex& v) { return load_pbr_colored(S, v.materialId, v.color); }
SPC_DEV f3 nv_forward_eye(const DeviceScene& S, const NVertex& self, const NVertex& b) {  // cuProg.h:1220-1243
    const f3 vec = b.position - self.position;
    const f3 c_dir = normalize(vec);
    const float g = fabsf(dot(c_dir, b.normal)) / dot(vec, vec);
    const float d_pdf = bsdf_pdf(nv_mat(S, self), self.normal, self.dir, c_dir);
    return self.weight * d_pdf * rr_of(self.color) * g;
}
SPC_DEV f3 nv_forward_light(const DeviceScene& S, const NVertex& self, const NVertex& b) {  // cuProg.h:1245-1282
    const f3 vec = b.position - self.position;
    const f3 c_dir = normalize(vec);
    const float g = fabsf(dot(c_dir, b.normal)) * fabsf(dot(c_dir, self.normal)) / dot(vec, vec);
    if (self.materialId < 0) return self.weight * g;
    return self.weight * g * bsdf_eval(nv_mat(S, self), self.normal, self.dir, c_dir);
}
SPC_DEV float nv_forward_light_pdf(const DeviceScene& S, const NVertex& self, const NVertex& b) {  // cuProg.h:1193-1218
    const f3 vec = b.position - self.position;
    const f3 c_dir = normalize(vec);
    float g = fabsf(dot(c_dir, b.normal)) / dot(vec, vec);
    if (self.materialId < 0) {
        g *= fabsf(dot(self.normal, c_dir));
        return self.pdf * g * kInvPi;
    }
    const float d_pdf = bsdf_pdf(nv_mat(S, self), self.normal, self.dir, c_dir);
    return self.pdf * d_pdf * rr_of(self.color) * g;
}
// nVertex_device(a, b, eye_side): the vertex a seen as the next vertex after b
SPC_DEV NVertex nv_extend(const DeviceScene& S, const NVertex& a, const NVertex& b, bool eye_side) {
    NVertex v;
    v.position = a.position;
    v.dir = normalize(b.position - a.position);
    v.normal = a.normal;
    v.weight = eye_side ? nv_forward_eye(S, b, a) : nv_forward_light(S, b, a);
    v.pdf = eye_side ? v.weight.x : nv_forward_light_pdf(S, b, a);
    v.color = a.color; v.materialId = a.materialId; v.label_id = a.label_id; v.depth = b.depth + 1;
    return v;
}
static constexpr int PRETRACE_MAX = 10;  // PRETRACE_CONN_PADDING

SPC_DEV void pretrace_build_path(const DeviceScene& S, const EyeVertex* buffer, int buffer_size, NVertex light,
                                 spcbpt_pretrace_path& path, spcbpt_pretrace_node* conn) {
    path.valid = 1;
    path.begin_ind = 0;
    path.end_ind = buffer_size - 1;
    int e = buffer_size - 1;
    NVertex n_eye = nv_from_eye(buffer[e]);
    const NVertex n_next_eye = nv_extend(S, light, n_eye, true);
    const f3 vec = light.position - n_eye.position;
    const f3 seg_contri = bsdf_eval(nv_mat(S, n_eye), n_eye.normal, n_eye.dir, normalize(vec));  // local_contri
    path.sample_pdf = n_next_eye.pdf + n_eye.pdf * light.pdf;
    path.fix_pdf = n_next_eye.pdf;
    f3 contri = buffer[e].flux * nv_forward_light(S, light, n_eye) * seg_contri;
    for (int i = 0; i < path.end_ind; i++) {
        spcbpt_pretrace_node& nd = conn[path.end_ind - i - 1];  // pathInfo_node(n_eye, light)
        nd.a_position[0] = n_eye.position.x; nd.a_position[1] = n_eye.position.y; nd.a_position[2] = n_eye.position.z;
        nd.b_position[0] = light.position.x; nd.b_position[1] = light.position.y; nd.b_position[2] = light.position.z;
        nd.a_dir[0] = n_eye.dir.x; nd.a_dir[1] = n_eye.dir.y; nd.a_dir[2] = n_eye.dir.z;
        nd.b_dir[0] = light.dir.x; nd.b_dir[1] = light.dir.y; nd.b_dir[2] = light.dir.z;
        nd.a_normal[0] = n_eye.normal.x; nd.a_normal[1] = n_eye.normal.y; nd.a_normal[2] = n_eye.normal.z;
        nd.b_normal[0] = light.normal.x; nd.b_normal[1] = light.normal.y; nd.b_normal[2] = light.normal.z;
        nd.peak_pdf = n_eye.weight.x * sum3(light.weight);
        nd.path_id = 0;
        nd.label_a = n_eye.depth;  // set_eye_depth
        nd.label_b = light.label_id;
        nd.valid = 1;
        nd.light_source = light.materialId < 0 ? 1 : 0;
        e--;
        light = nv_extend(S, n_eye, light, false);
        n_eye = nv_from_eye(buffer[e]);
    }
    const float wgt = sum3(contri) / path.sample_pdf;
    if (isnan(wgt) || isinf(wgt)) contri = mk3(0.0f);
    path.contri[0] = contri.x; path.contri[1] = contri.y; path.contri[2] = contri.z;
}

__global__ __launch_bounds__(BLOCK) void k_pretrace(const KParams p, uint32_t iteration, int num_core, int padding,
                                                    spcbpt_pretrace_path* __restrict__ paths, spcbpt_pretrace_node* __restrict__ nodes) {
    __shared__ uint32_t s_stack[BLOCK * STACK_LDS];
    const int launch_index = blockIdx.x * BLOCK + threadIdx.x;
    if (launch_index >= num_core) return;
    const DeviceScene& S = p.scene;
    Counts<false> cn;
    TravStack<BLOCK, STACK_LDS> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)launch_index, p.diag);
    WalkState w;
    w.seed = tea4((uint32_t)launch_index, iteration);
    const float jx = rnd(w.seed), jy = rnd(w.seed);
    w.dir = normalize((2.0f * jx - 1.0f) * ld3(p.U) + (2.0f * jy - 1.0f) * ld3(p.V) + ld3(p.W));
    w.origin = ld3(p.eye);
    w.done = false; w.next_flux = mk3(0.0f); w.next_single_pdf = 1.0f;
    EyeVertex buffer[PRETRACE_MAX];
    EyeVertex& cam = buffer[0];
    cam.c.pos = w.origin; cam.c.n = w.dir; cam.c.color = mk3(0.0f); cam.c.lastPos = w.origin; cam.c.lnp = 0.0f; cam.c.mat = 0; cam.c.lld = false;
    cam.flux = mk3(1.0f); cam.R3 = mk3(0.0f); cam.pdf = 1.0f; cam.singlePdf = 1.0f; cam.sub = 0; cam.lastZone = 0; cam.depth = 0;
    int buffer_size = 1, resample_number = 0, depth = 0;
    spcbpt_pretrace_path path;
    path.valid = 0; path.begin_ind = path.end_ind = 0; path.choice_id = 0; path.sample_pdf = path.fix_pdf = 0.0f;
    path.contri[0] = path.contri[1] = path.contri[2] = 0.0f; path.pad = 0;
    spcbpt_pretrace_node* conn = nodes + (size_t)launch_index * padding;
    while (true) {
        HitRec h;
        if (!traverse<false, false>(S, st, w.origin, w.dir, kEps, 1e16f, h, cn)) break;
        const Geom g = local_geometry(S, h);
        const EyeVertex& last = buffer[buffer_size - 1];
        const f3 ray_dir = w.dir;
        if (g.emitter) {
            const Pbr lm = load_pbr(S, g.mat);
            const DLight& L = S.lights[lm.light_id];
            if (dot(ray_dir, ld3(L.normal)) > 0) break;                       // back of the emitter: no vertex
            if (buffer_size + 1 > 2) {                                        // payload.path.size > 2
                const float r = rnd(w.seed);                                  // rr_acc_accept
                if (1.0f / (resample_number + 1) > r) {
                    const LightSampleD ls = light_reverse_sample(S, L, g.u, g.v);
                    pretrace_build_path(S, buffer, buffer_size, nv_from_light(ls), path, conn);
                    resample_number++;
                }
            }
            break;
        }
        if (buffer_size >= PRETRACE_MAX) break;  // cannot happen: the padding check below stops the walk first
        EyeVertex mid;
        eye_surface_hit(p, g, h.t, ray_dir, last.depth == 0, last, w, mid, cn);
        buffer[buffer_size] = mid;
        buffer_size++;
        // next-event candidate
        // QUAD lights only: upstream picks among all lights here too (raygen.cu:820-823) and then reads the sample's position, which
        // the ENV branch never sets -- undefined, so the sky is left out of the training pass's next-event candidates
        const int n_quads = S.n_lights - (S.env.valid ? 1 : 0);
        const int lid = min(max((int)floorf(rnd(w.seed) * n_quads), 0), n_quads - 1);
        const float r1 = rnd(w.seed), r2 = rnd(w.seed);
        const LightSampleD ls = light_reverse_sample(S, S.lights[lid], r1, r2);
        const f3 vis_vec = ls.position - mid.c.pos;
        const float len = sqrtf(dot(vis_vec, vis_vec));
        HitRec sh;
        if (!traverse<true, false>(S, st, mid.c.pos, vis_vec / len, kEps, len - kEps, sh, cn)) {
            const float r = rnd(w.seed);
            if (1.0f / (resample_number + 1) > r) {
                if (dot(vis_vec, ls.normal) < 0) {
                    pretrace_build_path(S, buffer, buffer_size, nv_from_light(ls), path, conn);
                    resample_number++;
                }
            }
        }
        if (w.done || depth > 50) break;
        if (buffer_size >= padding) break;  // PRETRACER_PADDING_VERTICES_CHECK
        depth += 1;
    }
    int begin_index = 0;
    if (path.valid) begin_index += path.end_ind - path.begin_ind;
    for (int i = begin_index; i < padding; i++) { conn[i].valid = 0; }
    path.sample_pdf = path.sample_pdf / (float)resample_number;
    const int bias = launch_index * padding;
    path.begin_ind += bias;
    path.end_ind += bias;
    path.pixel_id[0] = (int)((float)p.width * jx);
    path.pixel_id[1] = (int)((float)p.height * jy);
    if (path.begin_ind == path.end_ind && path.valid) path.valid = 0;
    paths[launch_index] = path;
}
void launch_pretrace(const KParams& p, uint32_t iteration, int num_core, int padding, spcbpt_pretrace_path* paths,
                     spcbpt_pretrace_node* nodes, hipStream_t s) {
    if (num_core <= 0) return;
    hipLaunchKernelGGL(k_pretrace, dim3((num_core + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, p, iteration, num_core, padding, paths, nodes);
}

void launch_trace_closest(const KParams& p, const float* rays, int n, float* t, int* tri, float* uv, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_trace_closest, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, p, rays, n, t, tri, uv);
}
void launch_trace_any(const KParams& p, const float* rays, int n, int* vis, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_trace_any, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, p, rays, n, vis);
}

}  // namespace spc

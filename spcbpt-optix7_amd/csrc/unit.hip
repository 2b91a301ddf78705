// Per-function device harness (spcbpt_debug_unit): the device functions of the hot path evaluated on caller-supplied records,
// one record per lane, so that tests/ can hold each of them to the oracle function by function instead of through images.
// Nothing here is on the product path; the functions called are the very ones k_spcbpt / k_light_trace inline (device_lib.h,
// eye_walk.h).  Record layouts (32-bit words) are documented in include/spcbpt.h next to spcbpt_debug_unit.
//   BSDF      Tracer::Sample / Eval / Pdf                                   cuProg.h:735-899
//   TREE      classTree::tree_index / labelUnit::getLabel                   classTree_common.h:39-52, cuProg.h:1109-1123
//   STAGE1    SubspaceSampler_device::sampleFirstStage (both forms)         cuProg.h:290-301
//   BSEARCH   binary_sample on a caller-supplied CMF                        cuProg.h:245-264
//   STAGE2    sampleSecondStage on the current sampler tables                cuProg.h:268-280
//   UNIFORM   uniformSample on the current sampler tables                    cuProg.h:283-289
//   CONNECT   connectVertex_SPCBPT + rmis::general_connection / connection_lightSource   raygen.cu:253-303, rmis.h:212-313
//   EYE_STEP  traceEyeSubPath + __closesthit__eyeSubpath / _LightSource + rmis::light_hit + lightStraghtHit
//             (cuProg.h:434-461, hit_program.cu:62-147, 246-340, rmis.h:359-389, raygen.cu:305-317)
#include <hip/hip_runtime.h>

#include "device_lib.h"
#include "eye_walk.h"
#include "kernels.h"

namespace spc {

static constexpr int UBLOCK = 256;

SPC_DEV f3 ldw3(const uint32_t* w) { return mk3(__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2])); }
SPC_DEV void stw3(uint32_t* w, f3 v) { w[0] = __float_as_uint(v.x); w[1] = __float_as_uint(v.y); w[2] = __float_as_uint(v.z); }
SPC_DEV float ldf(const uint32_t* w) { return __uint_as_float(*w); }
SPC_DEV void stf(uint32_t* w, float v) { *w = __float_as_uint(v); }

// spcbpt_unit_eye_vertex (25 words): position, normal, flux, color, last_position, rmis3, pdf, single_pdf,
// last_normal_projection, material_id, subspace_id, depth, last_zone_id
SPC_DEV EyeVertex load_eye_vertex(const uint32_t* w) {
    EyeVertex a;
    a.c.pos = ldw3(w); a.c.n = ldw3(w + 3); a.flux = ldw3(w + 6); a.c.color = ldw3(w + 9); a.c.lastPos = ldw3(w + 12); a.R3 = ldw3(w + 15);
    a.pdf = ldf(w + 18); a.singlePdf = ldf(w + 19); a.c.lnp = ldf(w + 20); a.c.lld = false;
    a.c.mat = (int)w[21]; a.sub = (int)w[22]; a.depth = (int)w[23]; a.lastZone = (int)w[24];
    return a;
}
SPC_DEV void store_eye_vertex(uint32_t* w, const EyeVertex& a) {
    stw3(w, a.c.pos); stw3(w + 3, a.c.n); stw3(w + 6, a.flux); stw3(w + 9, a.c.color); stw3(w + 12, a.c.lastPos); stw3(w + 15, a.R3);
    stf(w + 18, a.pdf); stf(w + 19, a.singlePdf); stf(w + 20, a.c.lnp);
    w[21] = (uint32_t)a.c.mat; w[22] = (uint32_t)a.sub; w[23] = (uint32_t)a.depth; w[24] = (uint32_t)a.lastZone;
}

__global__ __launch_bounds__(UBLOCK) void k_unit(const KParams p, int op, const uint32_t* __restrict__ in, int in_words,
                                                uint32_t* __restrict__ out, int out_words, int n, const float* __restrict__ aux) {
    __shared__ uint32_t s_stack[UBLOCK * kStackLds];
    const int i = blockIdx.x * UBLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t* r = in + (size_t)i * in_words;
    uint32_t* o = out + (size_t)i * out_words;
    Counts<false> cn;
    switch (op) {
    case SPCBPT_UNIT_BSDF: {
        Pbr m;
        m.base = ldw3(r); m.metallic = ldf(r + 3); m.roughness = ldf(r + 4); m.specular = ldf(r + 5); m.specularTint = ldf(r + 6);
        m.subsurface = ldf(r + 7); m.sheen = ldf(r + 8); m.sheenTint = ldf(r + 9); m.clearcoat = ldf(r + 10); m.clearcoatGloss = ldf(r + 11);
        m.albedo_tex = 0; m.light_id = -1; m.brdf = 0;
        const f3 N = ldw3(r + 12), V = ldw3(r + 15), L = ldw3(r + 18);
        uint32_t seed = r[21];
        const f3 Ls = bsdf_sample(m, N, V, seed);
        stw3(o, Ls); o[3] = seed;
        stw3(o + 4, bsdf_eval(m, N, V, L)); stf(o + 7, bsdf_pdf(m, N, V, L));
        stw3(o + 8, bsdf_eval(m, N, V, Ls)); stf(o + 11, bsdf_pdf(m, N, V, Ls));
        break;
    }
    case SPCBPT_UNIT_TREE: {
        const float* tree = r[0] ? p.light_tree : p.eye_tree;
        o[0] = (uint32_t)tree_label(tree, ldw3(r + 1), ldw3(r + 4), ldw3(r + 7), cn);
        break;
    }
    case SPCBPT_UNIT_STAGE1: {
        uint32_t s1 = r[1], s2 = r[1];
        float pmf1 = 0.0f, pmf2 = 0.0f;
        const int l1 = sample_first_stage(p, (int)r[0], s1, pmf1, cn);   // what the kernels run: counting passes on a monotone matrix
        const int l2 = binary_sample(p.cmf_gamma + (size_t)r[0] * SPCBPT_NUM_SUBSPACE, SPCBPT_NUM_SUBSPACE, s2, pmf2, cn);   // the reference's bisection
        o[0] = (uint32_t)l1; stf(o + 1, pmf1); o[2] = s1; o[3] = (uint32_t)l2; stf(o + 4, pmf2); o[5] = s2;
        break;
    }
    case SPCBPT_UNIT_BSEARCH: {
        uint32_t seed = r[2];
        float pmf = 0.0f;
        const int k = binary_sample(aux + r[0], (int)r[1], seed, pmf, cn);
        o[0] = (uint32_t)k; stf(o + 1, pmf); o[2] = seed;
        break;
    }
    case SPCBPT_UNIT_STAGE2: {
        const DSubspace ss = p.subspace[r[0]];
        uint32_t seed = r[1];
        float pmf = 0.0f;
        int k = -1, slot = -1;
        if (ss.size != 0) {   // raygen.cu:400-403: an empty subspace is skipped before any random number is drawn
            k = binary_sample(p.cmfs + ss.jump_bias, ss.size, seed, pmf, cn);
            slot = p.jump[ss.jump_bias + k];
        }
        o[0] = (uint32_t)ss.size; o[1] = (uint32_t)k; o[2] = (uint32_t)slot; stf(o + 3, pmf); o[4] = seed;
        break;
    }
    case SPCBPT_UNIT_UNIFORM: {
        uint32_t seed = r[0];
        float pmf = 0.0f;
        const int slot = uniform_sample(p.jump, p.sampler_counts[0], seed, pmf);
        o[0] = (uint32_t)slot; stf(o + 1, pmf); o[2] = seed;
        break;
    }
    case SPCBPT_UNIT_CONNECT: {
        const EyeVertex a = load_eye_vertex(r);
        LightVertex b;
        uint32_t* bw = reinterpret_cast<uint32_t*>(&b);
        for (int k = 0; k < 24; k++) bw[k] = r[25 + k];
        float w = 0.0f;
        f3 res = connect_vertices(p, a, b, cn, &w);
        if (is_invalid(res)) res = mk3(0.0f);   // connectVertex_SPCBPT's own guard (raygen.cu:298)
        stw3(o, res); stf(o + 3, w);
        break;
    }
    case SPCBPT_UNIT_EYE_STEP: {
        TravStack<UBLOCK, kStackLds> st;
        st.init(s_stack, p.spill, p.spill_entries, (size_t)i, p.diag);
        EyeVertex last = load_eye_vertex(r);
        WalkState w;
        w.next_flux = ldw3(r + 25); w.next_single_pdf = ldf(r + 28); w.seed = r[29];
        w.origin = last.c.pos; w.dir = ldw3(r + 30); w.done = false;
        for (int k = 0; k < 40; k++) o[k] = 0u;
        HitRec h;
        const f3 ray_dir = w.dir;
        if (!traverse<false, false>(p.scene, st, w.origin, w.dir, kEps, 1e16f, h, cn)) { o[0] = 0u; break; }   // __miss__BDPTVertex
        const Geom g = local_geometry(p.scene, h);
        const bool last_is_origin = last.depth == 0;
        if (g.emitter) {
            const DLight& L = p.scene.lights[load_pbr(p.scene, g.mat).light_id];
            const bool back = dot(ray_dir, ld3(L.normal)) > 0;
            o[0] = back ? 3u : 2u;
            stw3(o + 35, eye_emitter_hit(p, g, h.t, ray_dir, last_is_origin, last, w, cn));
            stf(o + 38, h.t);
        } else {
            EyeVertex mid;
            eye_surface_hit(p, g, h.t, ray_dir, last_is_origin, last, w, mid, cn, (r[33] & 1u) != 0);
            o[0] = 1u;
            store_eye_vertex(o + 1, mid);
            stw3(o + 26, w.dir); stw3(o + 29, w.next_flux); stf(o + 32, w.next_single_pdf); o[33] = w.seed; o[34] = w.done ? 1u : 0u;
            stf(o + 38, h.t);
        }
        break;
    }
    default: break;
    }
}

void launch_unit(const KParams& p, int op, const uint32_t* in, int in_words, uint32_t* out, int out_words, int n, const float* aux, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unit, dim3((n + UBLOCK - 1) / UBLOCK), dim3(UBLOCK), 0, s, p, op, in, in_words, out, out_words, n, aux);
}

}  // namespace spc

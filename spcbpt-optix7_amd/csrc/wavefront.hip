// Wavefront (streaming) form of the SPCBPT eye pass for gfx950.  Same arithmetic, same RNG streams and the same
// per-pixel summation order as the megakernel k_spcbpt (kernels.hip), but each phase of __raygen__SPCBPT
// (raygen.cu:319-443) is its own kernel over a queue in HBM:
//   k_wf_gen      camera ray + init_EyeSubpath (raygen.cu:216-231, 332-343)
//   k_wf_extend   optixTrace closest             -> hit records
//   k_wf_shade    __closesthit__eyeSubpath(+_LightSource) / __miss__BDPTVertex (hit_program.cu:58-147, 246-340) and the
//                 two-stage resampling of CONNECTION_N light vertices (raygen.cu:390-407) -> shadow-ray records
//   k_wf_shadow   visibilityTest (cuProg.h:463-487) -> compacted list of unoccluded connections
//   k_wf_connect  connectVertex_SPCBPT + rmis (raygen.cu:253-303, rmis.h)  -> per-connection contributions
//   k_wf_film     pixel write (raygen.cu:421-442)
// Why: inside the megakernel a wave spends most issue slots with a handful of live lanes (measured 17 % VALU lane
// utilisation) because the lanes sit in different phases, and traversal needs half the registers of the connection
// code it shares a kernel with.  Per-phase kernels run at their own occupancy over dense queues; the price is ~0.4 KB of
// state traffic per path and bounce, which HBM3E absorbs (< 2 % of a frame).
// One thread per queue item; launches are sized from a host-side upper bound of the queue length and blocks past the
// device-side length exit at once, so no host round trip sits between the phases.  Queue appends are aggregated per
// block (one atomic per 256 items): same-address atomics cost ~11 ns each on MI355X and would otherwise dominate.
#include <hip/hip_runtime.h>

#include "device_lib.h"
#include "eye_walk.h"
#include "kernels.h"

namespace spc {

static constexpr int WBLOCK = 256;
static constexpr int WSTACK = kStackLds;

SPC_DEV float4* wfq(const WfState& wf, int a) { return reinterpret_cast<float4*>(wf.a[a]); }
SPC_DEV uint32_t* wfc(const WfState& wf, int bounce, int k) { return wf.counts + bounce * WFC_ROW + k; }
SPC_DEV float4 pack(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
SPC_DEV float4 packi(f3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }
SPC_DEV f3 xyz(float4 q) { return mk3(q.x, q.y, q.z); }

// Index of this thread's item in a list grown by every thread of the block for which `pred` holds: one atomic per block.
// Must be reached by all WBLOCK threads.  `s_cnt` is 4 dwords of LDS.
SPC_DEV uint32_t block_append(uint32_t* counter, bool pred, uint32_t* s_cnt) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(pred);
    if (lane == 0) s_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
        const uint32_t total = c0 + c1 + c2 + c3;
        const uint32_t base = total ? atomicAdd(counter, total) : 0u;
        s_cnt[0] = base; s_cnt[1] = base + c0; s_cnt[2] = base + c0 + c1; s_cnt[3] = base + c0 + c1 + c2;
    }
    __syncthreads();
    return s_cnt[wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

SPC_DEV void load_eye_vertex(const WfState& wf, uint32_t slot, EyeVertex& v) {
    const float4 q_pos = wfq(wf, WF_POS)[slot], q_nrm = wfq(wf, WF_NRM)[slot], q_col = wfq(wf, WF_COL)[slot];
    const float4 q_lp = wfq(wf, WF_LASTPOS)[slot], q_fl = wfq(wf, WF_FLUX)[slot], q_r3 = wfq(wf, WF_R3)[slot];
    v.c.pos = xyz(q_pos); v.c.lnp = q_pos.w;
    v.c.n = xyz(q_nrm); v.c.mat = (int)__float_as_uint(q_nrm.w);
    v.c.color = xyz(q_col); v.pdf = q_col.w;
    v.c.lastPos = xyz(q_lp); v.singlePdf = q_lp.w;
    v.flux = xyz(q_fl);
    const uint32_t z = __float_as_uint(q_fl.w);
    v.sub = (int)(z & 0xffffu); v.lastZone = (int)(z >> 16);
    v.R3 = xyz(q_r3); v.depth = (int)__float_as_uint(q_r3.w);
}
SPC_DEV void store_eye_vertex(const WfState& wf, uint32_t slot, const EyeVertex& v) {
    wfq(wf, WF_POS)[slot] = pack(v.c.pos, v.c.lnp);
    wfq(wf, WF_NRM)[slot] = packi(v.c.n, (uint32_t)v.c.mat);
    wfq(wf, WF_COL)[slot] = pack(v.c.color, v.pdf);
    wfq(wf, WF_LASTPOS)[slot] = pack(v.c.lastPos, v.singlePdf);
    wfq(wf, WF_FLUX)[slot] = packi(v.flux, (uint32_t)v.sub | ((uint32_t)v.lastZone << 16));
    wfq(wf, WF_R3)[slot] = packi(v.R3, (uint32_t)v.depth);
}

// ------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(WBLOCK) void k_wf_gen(const KParams p, const WfState wf) {
    __shared__ uint32_t s_cnt[4];
    Counts<COUNT> cn;
    cn.clear();
    const uint32_t i = blockIdx.x * WBLOCK + threadIdx.x;
    bool valid = false;
    if (i < wf.n_slots) {
        uint32_t x, y;
        valid = tile_pixel(p, i >> 6, i & 63u, x, y);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4* contrib = reinterpret_cast<float4*>(wf.contrib);
        contrib[3 * (size_t)i] = zero; contrib[3 * (size_t)i + 1] = zero; contrib[3 * (size_t)i + 2] = zero;
        if (valid) {
            uint32_t seed;
            const f3 dir = camera_ray(p, x, y, seed);
            const f3 origin = ld3(p.eye);
            wfq(wf, WF_DIR)[i] = packi(dir, seed);
            wfq(wf, WF_NEXT)[i] = pack(mk3(0.0f), 1.0f);
            EyeVertex cur;  // init_EyeSubpath
            cur.c.pos = origin; cur.c.n = dir; cur.c.color = mk3(0.0f); cur.c.lastPos = origin; cur.c.lnp = 0.0f; cur.c.mat = 0;
            cur.flux = mk3(1.0f); cur.R3 = mk3(0.0f); cur.pdf = 1.0f; cur.singlePdf = 1.0f; cur.sub = 0; cur.lastZone = 0; cur.depth = 0;
            store_eye_vertex(wf, i, cur);
            wfq(wf, WF_RESULT)[i] = packi(mk3(0.0f), x | (y << 16));
            cn.add(C_PIX); cn.add(C_EYE);
        } else {
            wfq(wf, WF_RESULT)[i] = packi(mk3(0.0f), 0xffffffffu);
        }
    }
    const uint32_t k = block_append(wfc(wf, 0, WFC_EXT_COUNT), valid, s_cnt);
    if (valid) wf.queue[0][k] = i;
    cn.flush(p.counters);
}

template <bool COUNT>
__global__ __launch_bounds__(WBLOCK) void k_wf_extend(const KParams p, const WfState wf, int bounce) {
    __shared__ uint32_t s_stack[WBLOCK * WSTACK];
    const uint32_t count = *wfc(wf, bounce, WFC_EXT_COUNT);
    if (blockIdx.x * WBLOCK >= count) return;
    Counts<COUNT> cn;
    cn.clear();
    const uint32_t item = blockIdx.x * WBLOCK + threadIdx.x;
    if (item < count) {
        TravStack<WBLOCK, WSTACK> st;
        st.init(s_stack, p.spill, p.spill_entries, (size_t)item, p.diag);
        const uint32_t slot = wf.queue[bounce & 1][item];
        const float4 o = wfq(wf, WF_POS)[slot], d = wfq(wf, WF_DIR)[slot];
        HitRec h;
        cn.add(C_CLOSEST);
        const bool hit = traverse<false, COUNT>(p.scene, st, xyz(o), xyz(d), kEps, 1e16f, h, cn);
        wfq(wf, WF_HIT)[slot] = make_float4(h.t, __int_as_float(hit ? h.tri : -1), h.u, h.v);
    }
    cn.flush(p.counters);
}

template <bool COUNT>
__global__ __launch_bounds__(WBLOCK) void k_wf_shade(const KParams p, const WfState wf, int bounce) {
    __shared__ uint32_t s_cnt[4];
    const DeviceScene& S = p.scene;
    const uint32_t count = *wfc(wf, bounce, WFC_EXT_COUNT);
    if (blockIdx.x * WBLOCK >= count) return;
    Counts<COUNT> cn;
    cn.clear();
    const int path_count = p.sampler_counts[1];
    float4* contrib = reinterpret_cast<float4*>(wf.contrib);
    float4* conn_ray = reinterpret_cast<float4*>(wf.conn_ray);
    uint4* conn_rec = reinterpret_cast<uint4*>(wf.conn_rec);
    const uint32_t item = blockIdx.x * WBLOCK + threadIdx.x;
    bool survives = false;
    uint32_t slot = 0;
    if (item < count) {
        slot = wf.queue[bounce & 1][item];
        // fold the connections of the previous vertex into the path's radiance, in connection order
        const float4 rq = wfq(wf, WF_RESULT)[slot];
        f3 result = xyz(rq);
        if (bounce > 0) {
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                result += xyz(contrib[3 * (size_t)slot + it]);
                contrib[3 * (size_t)slot + it] = zero;
            }
        }
        const float4 hq = wfq(wf, WF_HIT)[slot];
        HitRec h; h.t = hq.x; h.tri = __float_as_int(hq.y); h.u = hq.z; h.v = hq.w;
        bool surface = false;
        if (h.tri >= 0) {  // else __miss__BDPTVertex
            EyeVertex cur;
            load_eye_vertex(wf, slot, cur);
            const float4 dq = wfq(wf, WF_DIR)[slot], nq = wfq(wf, WF_NEXT)[slot];
            WalkState w;
            w.origin = cur.c.pos; w.dir = xyz(dq); w.seed = __float_as_uint(dq.w);
            w.next_flux = xyz(nq); w.next_single_pdf = nq.w; w.done = false;
            const Geom g = local_geometry(S, h);
            const bool last_is_origin = cur.depth == 0;
            const f3 ray_dir = w.dir;
            if (g.emitter) {
                result += eye_emitter_hit(p, g, h.t, ray_dir, last_is_origin, cur, w, cn);
            } else {
                surface = true;
                EyeVertex mid;
                eye_surface_hit(p, g, h.t, ray_dir, last_is_origin, cur, w, mid, cn, true);
                store_eye_vertex(wf, slot, mid);
                // shadow-ray records sit at fixed positions (connection-major: it * count + item), no compaction needed
                for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                    float pmf1, pmf2 = 0.0f;
                    const int l = sample_first_stage(p, mid.sub, w.seed, pmf1, cn);
                    const DSubspace ss = p.subspace[l];
                    const size_t k = (size_t)it * count + item;
                    if (ss.size != 0) {
                        const int kk = binary_sample(p.cmfs + ss.jump_bias, ss.size, w.seed, pmf2, cn);
                        const int lslot = p.jump[ss.jump_bias + kk];
                        const float4 bq0 = reinterpret_cast<const float4*>(p.lvc + lslot)[0];
                        const float4 bq1 = reinterpret_cast<const float4*>(p.lvc + lslot)[1];
                        const f3 bias = xyz(bq0) - mid.c.pos;
                        const float len = sqrtf(dot(bias, bias));
                        const f3 sdir = bias / len;
                        cn.add(C_CONN);
                        const float pmf = (float)path_count * pmf2 * pmf1;
                        conn_ray[k] = pack(sdir, len);
                        // zero-valued pairs are not traced (null_connection, device_lib.h)
                        const bool null_conn = null_connection(mid.c.pos, mid.c.n, xyz(bq0), xyz(bq1));
                        conn_rec[k] = make_uint4((uint32_t)lslot, __float_as_uint(pmf), null_conn ? 0xffffffffu : slot, (uint32_t)it);
                    } else {
                        conn_rec[k] = make_uint4(0u, 0u, 0xffffffffu, (uint32_t)it);
                    }
                }
                survives = !(w.done || mid.depth > 50);  // the loop-top test of raygen.cu:361
                if (survives) {
                    wfq(wf, WF_DIR)[slot] = packi(w.dir, w.seed);
                    wfq(wf, WF_NEXT)[slot] = pack(w.next_flux, w.next_single_pdf);
                }
            }
        }
        if (!surface) {
#pragma unroll
            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) conn_rec[(size_t)it * count + item] = make_uint4(0u, 0u, 0xffffffffu, (uint32_t)it);
        }
        wfq(wf, WF_RESULT)[slot] = make_float4(result.x, result.y, result.z, rq.w);
    }
    const uint32_t k = block_append(wfc(wf, bounce + 1, WFC_EXT_COUNT), survives, s_cnt);
    if (survives) wf.queue[(bounce + 1) & 1][k] = slot;
    cn.flush(p.counters);
}

template <bool COUNT>
__global__ __launch_bounds__(WBLOCK) void k_wf_shadow(const KParams p, const WfState wf, int bounce) {
    __shared__ uint32_t s_stack[WBLOCK * WSTACK];
    __shared__ uint32_t s_cnt[4];
    const uint32_t count = SPCBPT_CONNECTION_N * *wfc(wf, bounce, WFC_EXT_COUNT);
    if (blockIdx.x * WBLOCK >= count) return;
    Counts<COUNT> cn;
    cn.clear();
    const uint32_t item = blockIdx.x * WBLOCK + threadIdx.x;
    bool visible = false;
    if (item < count) {
        const uint32_t slot = reinterpret_cast<const uint4*>(wf.conn_rec)[item].z;
        if (slot != 0xffffffffu) {
            TravStack<WBLOCK, WSTACK> st;
            st.init(s_stack, p.spill, p.spill_entries, (size_t)item, p.diag);
            const float4 r = reinterpret_cast<const float4*>(wf.conn_ray)[item];
            const float4 o = wfq(wf, WF_POS)[slot];
            HitRec sh;
            cn.add(C_SHADOW);
            visible = !traverse<true, COUNT>(p.scene, st, xyz(o), xyz(r), kEps, r.w - kEps, sh, cn);
        }
    }
    const uint32_t k = block_append(wfc(wf, bounce, WFC_VIS_COUNT), visible, s_cnt);
    if (visible) wf.vis[k] = item;
    cn.flush(p.counters);
}

template <bool COUNT>
__global__ __launch_bounds__(WBLOCK) void k_wf_connect(const KParams p, const WfState wf, int bounce) {
    const uint32_t count = *wfc(wf, bounce, WFC_VIS_COUNT);
    if (blockIdx.x * WBLOCK >= count) return;
    Counts<COUNT> cn;
    cn.clear();
    const uint32_t item = blockIdx.x * WBLOCK + threadIdx.x;
    if (item < count) {
        const uint4 rec = reinterpret_cast<const uint4*>(wf.conn_rec)[wf.vis[item]];
        EyeVertex cur;
        load_eye_vertex(wf, rec.z, cur);
        LightVertex b;
        const float4* src = reinterpret_cast<const float4*>(p.lvc + rec.x);
        float4* dst = reinterpret_cast<float4*>(&b);
#pragma unroll
        for (int q = 0; q < 6; q++) dst[q] = src[q];
        f3 res = connect_vertices(p, cur, b, cn);
        if (is_invalid(res)) res = mk3(0.0f);
        res = res / __uint_as_float(rec.y);
        if (!is_invalid(res)) reinterpret_cast<float4*>(wf.contrib)[3 * (size_t)rec.z + rec.w] = pack(res / (float)SPCBPT_CONNECTION_N, 0.0f);
    }
    cn.flush(p.counters);
}

__global__ __launch_bounds__(WBLOCK) void k_wf_film(const KParams p, const WfState wf) {
    const uint32_t i = blockIdx.x * WBLOCK + threadIdx.x;
    if (i >= wf.n_slots) return;
    const float4 rq = wfq(wf, WF_RESULT)[i];
    const uint32_t pix = __float_as_uint(rq.w);
    if (pix == 0xffffffffu) return;
    const float4* contrib = reinterpret_cast<const float4*>(wf.contrib);
    f3 result = xyz(rq);
#pragma unroll
    for (int it = 0; it < SPCBPT_CONNECTION_N; it++) result += xyz(contrib[3 * (size_t)i + it]);
    film_write(p, pix & 0xffffu, pix >> 16, result);
}

// ------------------------------------------------------------------------------------------------
static inline dim3 blocks_for(size_t items) { return dim3((unsigned)std::max<size_t>(1, (items + WBLOCK - 1) / WBLOCK)); }

void launch_wf_gen(const KParams& p, const WfState& wf, bool count, hipStream_t s) {
    if (count) hipLaunchKernelGGL(k_wf_gen<true>, blocks_for(wf.n_slots), dim3(WBLOCK), 0, s, p, wf);
    else hipLaunchKernelGGL(k_wf_gen<false>, blocks_for(wf.n_slots), dim3(WBLOCK), 0, s, p, wf);
}
// `bound` = host-side upper bound of this bounce's queue length
void launch_wf_bounce(const KParams& p, const WfState& wf, int bounce, bool count, size_t bound, hipStream_t s) {
    const dim3 g1 = blocks_for(bound), g3 = blocks_for(bound * SPCBPT_CONNECTION_N), b(WBLOCK);
    if (count) {
        hipLaunchKernelGGL(k_wf_extend<true>, g1, b, 0, s, p, wf, bounce);
        hipLaunchKernelGGL(k_wf_shade<true>, g1, b, 0, s, p, wf, bounce);
        hipLaunchKernelGGL(k_wf_shadow<true>, g3, b, 0, s, p, wf, bounce);
        hipLaunchKernelGGL(k_wf_connect<true>, g3, b, 0, s, p, wf, bounce);
    } else {
        hipLaunchKernelGGL(k_wf_extend<false>, g1, b, 0, s, p, wf, bounce);
        hipLaunchKernelGGL(k_wf_shade<false>, g1, b, 0, s, p, wf, bounce);
        hipLaunchKernelGGL(k_wf_shadow<false>, g3, b, 0, s, p, wf, bounce);
        hipLaunchKernelGGL(k_wf_connect<false>, g3, b, 0, s, p, wf, bounce);
    }
}
void launch_wf_film(const KParams& p, const WfState& wf, hipStream_t s) {
    hipLaunchKernelGGL(k_wf_film, blocks_for(wf.n_slots), dim3(WBLOCK), 0, s, p, wf);
}

}  // namespace spc

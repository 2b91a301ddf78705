// Static quad-lane traversal (round 4 prototype): FOUR consecutive lanes trace ONE ray.
//
// The lane-per-ray loop of device_lib.h fetches a 64-B node with 4 x global_load_dwordx4 per lane from 64 different lines (256 tag
// lookups in the vector L1 per 64 node visits), slab-tests four children and sorts four keys per lane, and runs its triangle step
// at a tenth of the lanes.  Here a quad shares the ray:
//   * node visit: lane r of the quad loads 16 B -- record r of the node, ONE 64-B line per ray and ONE load instruction per visit
//     (16 tag lookups per wave-instruction); the record holds child r's six quantised planes, its stack word and a quarter of the
//     node's shared header (origin.xyz, scale exponents), which the quad exchanges with DPP quad_perm broadcasts; the lane tests
//     ITS child; the four entry distances are ranked across the quad with three quad_perm rotations (no sorting network), the
//     nearest hit continues, the others go to the quad's LDS stack at rank-computed slots (one ds_write for the whole quad);
//   * leaf: lane r tests triangle r of the (<= 4) triangles of the leaf, the nearest distance is a two-step quad reduction;
//     each lane keeps the best hit among ITS triangles and the quad picks the winner once, at the end of the ray;
//   * one LDS stack per RAY: 64 entries per ray in the 16 KB that hold 16 entries per lane in the lane-per-ray kernels: a BVH of
//     depth <= 21 never spills to HBM.
// Both forms here are persistent, pool-fed kernels (a lane / a quad pulls the next ray of the launch when its ray is finished, as
// trace_pool does inside the megakernel), so that the comparison is not decided by the tail of a one-ray-per-lane launch.
// Same node contents (re-laid out per node, see k_repack_nodes_quad), same slab arithmetic, same triangle test, same tie rules
// (strictly nearer replaces; within a leaf the lower triangle index wins): the hits are the lane-per-ray kernel's hits.
#include "device_lib.h"
#include "kernels.h"

namespace spc {

static constexpr int QBLOCK = 256;
static constexpr int QRAYS = QBLOCK / 4;    // rays (quads) in flight per block
static constexpr int QSTACK = 64;           // stack entries per ray
static constexpr int QSTRIDE = QSTACK + 1;  // odd stride: the four pushes of a quad and the quads of a wave spread over the banks

// DPP quad_perm controls
static constexpr int kBcast0 = 0x00, kBcast1 = 0x55, kBcast2 = 0xAA, kBcast3 = 0xFF;
static constexpr int kRot1 = 0x39, kRot2 = 0x4E, kRot3 = 0x93, kXor1 = 0xB1;   // [1,2,3,0] [2,3,0,1] [3,0,1,2] [1,0,3,2]
template <int CTRL>
SPC_DEV uint32_t qperm(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
SPC_DEV float qpermf(float v) { return __uint_as_float(qperm<CTRL>(__float_as_uint(v))); }

// node layout of the quad kernels: record r (float4 r of the 64-B node) =
//   x: qlo.x[r] | qlo.y[r] << 8 | qlo.z[r] << 16 | qhi.x[r] << 24     y: qhi.y[r] | qhi.z[r] << 8
//   z: stack word of child r (layout.h: internal node index, or leaf ref)   w: r = 0..2: origin.x/y/z, r = 3: scale exponents
__global__ void k_repack_nodes_quad(const float4* __restrict__ nodes, float4* __restrict__ out, int n_nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const float4 q0 = nodes[(size_t)i * 4], q1 = nodes[(size_t)i * 4 + 1], q2 = nodes[(size_t)i * 4 + 2], q3 = nodes[(size_t)i * 4 + 3];
    const uint32_t lx = __float_as_uint(q1.x), ly = __float_as_uint(q1.y), lz = __float_as_uint(q1.z), hx = __float_as_uint(q1.w),
                   hy = __float_as_uint(q2.x), hz = __float_as_uint(q2.y);
    const uint32_t ref[4] = {__float_as_uint(q2.z), __float_as_uint(q2.w), __float_as_uint(q3.x), __float_as_uint(q3.y)};
    const float shared[4] = {q0.x, q0.y, q0.z, q0.w};
    for (int r = 0; r < 4; r++) {
        const uint32_t s = 8u * (uint32_t)r;
        const uint32_t a = ((lx >> s) & 255u) | (((ly >> s) & 255u) << 8) | (((lz >> s) & 255u) << 16) | (((hx >> s) & 255u) << 24);
        const uint32_t b = ((hy >> s) & 255u) | (((hz >> s) & 255u) << 8);
        out[(size_t)i * 4 + r] = make_float4(__uint_as_float(a), __uint_as_float(b), __uint_as_float(ref[r]), shared[r]);
    }
}

struct QuadArgs {
    const float4* nodes_q;   // quad layout
    const float* rays;       // 8 floats per ray: origin, tmin, direction, tmax
    int n;
    uint32_t* counter;       // pool cursor (zeroed before the launch)
    float* out_t; int* out_tri; float* out_uv;   // closest
    int* out_visible;                            // any
    uint32_t chunk;                              // rays a wave claims per atomic (launch_trace_bench: a quarter of a wave's fair share, 64 .. 1024)
    unsigned long long* stats;                   // [0] node visits, [1] leaf visits, [2] triangle tests, [3] wave iterations (x 64 lane slots), [4] lanes busy in them
};

// Ray acquisition.  A wave claims the launch's rays in chunks of A.chunk (ONE global atomic per chunk; the chunk is a quarter of a
// wave's fair share of the launch, so that every resident wave gets work and the tail stays short) and hands them to its lanes /
// quads from a wave-uniform cursor with ballot arithmetic.  (First version: one atomicAdd on the launch's single counter per
// refill, i.e. in nearly every iteration of every wave -- 8 192 waves serialised on one L2 atomic and BOTH schedules measured that,
// not their traversal: lane 660, quad 195 Mrays/s on the primary rays.)  All lanes of the wave call this together; `want` = the
// lanes that take a ray (one per lane / per quad), `rank` = number of takers before this lane.  Returns the ray index or 0xffffffff
// (no ray this time: the chunk ran out -- the next call claims a new one -- or the launch has no more rays).
SPC_DEV uint32_t pool_take(const QuadArgs& A, uint32_t& chunk_next, uint32_t& chunk_end, bool& exhausted, unsigned long long want, uint32_t rank) {
    if (chunk_next >= chunk_end) {
        uint32_t base = 0;
        if ((threadIdx.x & 63u) == 0u) base = atomicAdd(A.counter, A.chunk);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= (uint32_t)A.n) { exhausted = true; return 0xffffffffu; }
        chunk_next = base;
        chunk_end = min(base + A.chunk, (uint32_t)A.n);
    }
    const uint32_t mine = chunk_next + rank;
    chunk_next = min(chunk_next + (uint32_t)__popcll(want), chunk_end);
    return mine < chunk_end ? mine : 0xffffffffu;
}

// ---- four lanes per ray ------------------------------------------------------------------------------------------------------
template <bool ANY, bool STATS>
__global__ __launch_bounds__(QBLOCK) void k_trace_quad(const DeviceScene S, const QuadArgs A) {
    __shared__ uint32_t s_stack[QRAYS * QSTRIDE];
    const uint32_t lane = threadIdx.x & 63u, r = lane & 3u;
    uint32_t* stack = s_stack + (threadIdx.x >> 2) * QSTRIDE;   // this ray's column
    int sp = 0;
    int node = kTravDone, leaf_count = 0;
    int ray = -1;
    f3 o = mk3(0.f), d = mk3(0.f), inv = mk3(1.f), ood = mk3(0.f);
    float tmin = 0.f, best_t = 0.f;                    // best_t: the quad's nearest hit so far (same in its four lanes)
    float my_t = 1e30f, my_u = 0.f, my_v = 0.f;        // the best hit among the triangles THIS lane tested
    int my_tri = -1;
    bool occluded = false;
    bool exhausted = false;                            // the pool is empty (wave-uniform once set)
    uint32_t chunk_next = 0, chunk_end = 0;            // this wave's claimed range of the launch's rays (wave-uniform)
    unsigned long long n_node = 0, n_leaf = 0, n_tri = 0, n_iter = 0, n_busy = 0;
    while (true) {
        // ---- quads without a ray draw the next rays of the wave's chunk (pool_take: no atomic unless the chunk is used up)
        const bool need = node == kTravDone;
        if (!exhausted) {
            const unsigned long long want = __ballot(need && r == 0u);
            if (want) {
                const uint32_t mine = pool_take(A, chunk_next, chunk_end, exhausted, want, (uint32_t)__popcll(want & ((1ull << (lane & ~3u)) - 1ull)));
                if (need && mine != 0xffffffffu) {
                    ray = (int)mine;
                    const float4 ra = ldq(A.rays, (size_t)ray * 2), rb = ldq(A.rays, (size_t)ray * 2 + 1);
                    o = mk3(ra.x, ra.y, ra.z); d = mk3(rb.x, rb.y, rb.z);
                    tmin = ra.w; best_t = rb.w;
                    inv = safe_inv(d); ood = o * inv;
                    node = 0; sp = 0; leaf_count = 0;
                    my_t = 1e30f; my_tri = -1; occluded = false;
                }
            }
        }
        if (!__any(node != kTravDone)) break;
        if (STATS) { n_iter += lane == 0 ? 1 : 0; n_busy += node != kTravDone ? 1 : 0; }
        if (node != kTravDone) {
            bool finished = false;
            if (node >= 0) {
                // ---- node visit: one 16-B record per lane, the child of this lane
                const float4 rec = ldq(reinterpret_cast<const float*>(A.nodes_q), (size_t)node * 4 + r);
                if (STATS && r == 0u) n_node++;
                const float ox = qpermf<kBcast0>(rec.w), oy = qpermf<kBcast1>(rec.w), oz = qpermf<kBcast2>(rec.w);
                const uint32_t e = qperm<kBcast3>(__float_as_uint(rec.w));
                const float ax = __uint_as_float((e & 0xffu) << 23) * inv.x, ay = __uint_as_float(((e >> 8) & 0xffu) << 23) * inv.y,
                            az = __uint_as_float(((e >> 16) & 0xffu) << 23) * inv.z;
                const float bx = fmaf(ox, inv.x, -ood.x), by = fmaf(oy, inv.y, -ood.y), bz = fmaf(oz, inv.z, -ood.z);
                const uint32_t pa = __float_as_uint(rec.x), pb = __float_as_uint(rec.y);
                const float lx = (float)(pa & 255u), ly = (float)((pa >> 8) & 255u), lz = (float)((pa >> 16) & 255u), hx = (float)(pa >> 24),
                            hy = (float)(pb & 255u), hz = (float)((pb >> 8) & 255u);
                const bool sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
                const float tnx = fmaf(sx ? hx : lx, ax, bx), tfx = fmaf(sx ? lx : hx, ax, bx);
                const float tny = fmaf(sy ? hy : ly, ay, by), tfy = fmaf(sy ? ly : hy, ay, by);
                const float tnz = fmaf(sz ? hz : lz, az, bz), tfz = fmaf(sz ? lz : hz, az, bz);
                const float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
                const float t1 = fminf(fminf(tfx, tfy), fminf(tfz, best_t));
                const uint32_t key = (t0 <= t1 * 1.0000004f) ? ((__float_as_uint(t0) & ~3u) | r) : 0xffffffffu;
                const uint32_t ref = __float_as_uint(rec.z);
                // ---- rank across the quad (keys of hits are distinct: the slot sits in the low bits)
                const uint32_t k1 = qperm<kRot1>(key), k2 = qperm<kRot2>(key), k3 = qperm<kRot3>(key);
                const bool hit = key != 0xffffffffu;
                const int rank = (k1 < key ? 1 : 0) + (k2 < key ? 1 : 0) + (k3 < key ? 1 : 0);
                const int nh = (hit ? 1 : 0) + (k1 != 0xffffffffu ? 1 : 0) + (k2 != 0xffffffffu ? 1 : 0) + (k3 != 0xffffffffu ? 1 : 0);
                uint32_t next;
                if (nh == 0) {
                    if (sp == 0) next = 0xffffffffu; else next = stack[--sp];
                } else {
                    if (hit && rank > 0) stack[sp + nh - 1 - rank] = ref;   // farthest deepest, second nearest on top
                    sp += nh - 1;
                    const uint32_t mine = (hit && rank == 0) ? ref : 0u;
                    next = mine | qperm<kRot1>(mine) | qperm<kRot2>(mine) | qperm<kRot3>(mine);
                }
                if (next == 0xffffffffu) { node = kTravDone; finished = true; }
                else if (next & 0x80000000u) { node = ~(int)((next & 0x7fffffffu) >> 3); leaf_count = (int)(next & 7u); }
                else node = (int)next;
            }
            if (!finished && node < 0) {
                // ---- leaf: triangle r of the leaf for lane r (an empty slot's leaf has no triangles)
                float t = 1e30f, u = 0.f, v = 0.f;
                bool h = false;
                const int tri = ~node + (int)r;
                if ((int)r < leaf_count) {
                    const size_t base = (size_t)tri * 4;
                    const float4 a = ldq(S.tris, base), b = ldq(S.tris, base + 1), c = ldq(S.tris, base + 2);
                    bool cull = false;
                    if (!ANY) cull = (__float_as_uint(ldq(S.tris, base + 3).w) & 0x80000000u) != 0;   // single-sided emitters (q16)
                    h = tri_test(a, b, c, o, d, tmin, best_t, cull, t, u, v);
                    if (!h) t = 1e30f;
                    if (STATS) n_tri++;
                }
                if (STATS && r == 0u) n_leaf++;
                if (h && t < my_t) { my_t = t; my_tri = tri; my_u = u; my_v = v; }   // (t < best_t <= my_t whenever h)
                float tq = fminf(t, qpermf<kXor1>(t));
                tq = fminf(tq, qpermf<kRot2>(tq));
                if (ANY) {
                    if (tq < 1e30f) { occluded = true; node = kTravDone; finished = true; }
                } else best_t = fminf(best_t, tq);
                if (!finished) {
                    if (sp == 0) { node = kTravDone; finished = true; }
                    else {
                        const uint32_t w = stack[--sp];
                        if (w & 0x80000000u) { node = ~(int)((w & 0x7fffffffu) >> 3); leaf_count = (int)(w & 7u); }
                        else node = (int)w;
                    }
                }
            }
            if (finished) {
                if (ANY) {
                    if (r == 0u) A.out_visible[ray] = occluded ? 0 : 1;
                } else {
                    // the winner: the lane that holds the quad's nearest hit; the lowest lane on a tie (= the lower triangle index)
                    const uint32_t cand = (my_tri >= 0 && my_t == best_t) ? r : 4u;
                    uint32_t w = min(cand, qperm<kXor1>(cand));
                    w = min(w, qperm<kRot2>(w));
                    if (w == 4u) { if (r == 0u) { A.out_t[ray] = best_t; A.out_tri[ray] = -1; A.out_uv[2 * ray] = 0.f; A.out_uv[2 * ray + 1] = 0.f; } }
                    else if (w == r) { A.out_t[ray] = my_t; A.out_tri[ray] = S.tri_orig[my_tri]; A.out_uv[2 * ray] = my_u; A.out_uv[2 * ray + 1] = my_v; }
                }
            }
        }
    }
    if (STATS) {
        const unsigned long long v[5] = {n_node, n_leaf, n_tri, n_iter * 64ull, n_busy};
        for (int k = 0; k < 5; k++) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) atomicAdd(&A.stats[k], x);
        }
    }
}

// ---- four lanes per ray, lean node step ("quad2") --------------------------------------------------------------------------------
// k_trace_quad is instruction-bound (VALU issue 0.70, 2.3 x the lane form's instructions): this form attacks the count.
//   * the near / far plane bytes of the lane's child are picked by two v_perm_b32 whose selectors are per-RAY constants (the signs of
//     the direction), then six byte->float converts: 8 instructions instead of 6 converts + 3 compares + 6 selects;
//   * the node's scale exponents are stored as signed bytes (k_repack_nodes_quad2) and applied with v_ldexp_f32: 2 instead of 3 per axis;
//   * which children were hit comes from ONE ballot (the compare's own result) and a shift, the nearest from two DPP min steps, and the
//     other hits are pushed in SLOT order (a popcount), not sorted: the traversal is correct in any order, only the nearest child
//     matters much for the visit count (measured below);
//   * 32-bit node offsets from a scalar base.
__global__ void k_repack_nodes_quad2(const float4* __restrict__ q1, float4* __restrict__ out, int n_nodes) {   // in: the quad layout; out: exponents as signed bytes
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    for (int r = 0; r < 4; r++) {
        float4 v = q1[(size_t)i * 4 + r];
        if (r == 3) {
            const uint32_t e = __float_as_uint(v.w);
            const uint32_t s = ((e & 255u) - 127u) & 255u, t = (((e >> 8) & 255u) - 127u) & 255u, u = (((e >> 16) & 255u) - 127u) & 255u;
            v.w = __uint_as_float(s | (t << 8) | (u << 16));
        }
        out[(size_t)i * 4 + r] = v;
    }
}
template <bool ANY, bool STATS>
__global__ __launch_bounds__(QBLOCK) void k_trace_quad2(const DeviceScene S, const QuadArgs A) {
    __shared__ uint32_t s_stack[QRAYS * QSTRIDE];
    const uint32_t lane = threadIdx.x & 63u, r = lane & 3u;
    uint32_t* stack = s_stack + (threadIdx.x >> 2) * QSTRIDE;
    const uint32_t below = (1u << r) - 1u, quad_shift = lane & 60u;
    const char* const node_base = reinterpret_cast<const char*>(A.nodes_q);
    int sp = 0;
    int node = kTravDone, leaf_count = 0;
    int ray = -1;
    f3 o = mk3(0.f), d = mk3(0.f), inv = mk3(1.f), ood = mk3(0.f);
    uint32_t sel_n = 0, sel_f = 0;                     // v_perm selectors of the near / far plane bytes (from the direction's signs)
    float tmin = 0.f, best_t = 0.f;
    float my_t = 1e30f, my_u = 0.f, my_v = 0.f;
    int my_tri = -1;
    bool occluded = false, exhausted = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    unsigned long long n_node = 0, n_leaf = 0, n_tri = 0, n_iter = 0, n_busy = 0;
    while (true) {
        const bool need = node == kTravDone;
        if (!exhausted) {
            const unsigned long long want = __ballot(need && r == 0u);
            if (want) {
                const uint32_t mine = pool_take(A, chunk_next, chunk_end, exhausted, want, (uint32_t)__popcll(want & ((1ull << quad_shift) - 1ull)));
                if (need && mine != 0xffffffffu) {
                    ray = (int)mine;
                    const float4 ra = ldq(A.rays, (size_t)ray * 2), rb = ldq(A.rays, (size_t)ray * 2 + 1);
                    o = mk3(ra.x, ra.y, ra.z); d = mk3(rb.x, rb.y, rb.z);
                    tmin = ra.w; best_t = rb.w;
                    inv = safe_inv(d); ood = o * inv;
                    // record bytes: x = [lo.x, lo.y, lo.z, hi.x] (perm indices 0..3), y = [hi.y, hi.z, -, -] (indices 4, 5)
                    const uint32_t sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
                    sel_n = (sx ? 3u : 0u) | ((sy ? 4u : 1u) << 8) | ((sz ? 5u : 2u) << 16) | ((sx ? 0u : 3u) << 24);   // near x, y, z, far x
                    sel_f = (sy ? 1u : 4u) | ((sz ? 2u : 5u) << 8) | (0x0cu << 16) | (0x0cu << 24);                     // far y, z, 0, 0
                    node = 0; sp = 0; leaf_count = 0;
                    my_t = 1e30f; my_tri = -1; occluded = false;
                }
            }
        }
        if (!__any(node != kTravDone)) break;
        if (STATS) { n_iter += lane == 0 ? 1 : 0; n_busy += node != kTravDone ? 1 : 0; }
        if (node != kTravDone) {
            bool finished = false;
            if (node >= 0) {
                const float4 rec = *reinterpret_cast<const float4*>(node_base + ((uint32_t)node * 64u + r * 16u));
                if (STATS && r == 0u) n_node++;
                const float ox = qpermf<kBcast0>(rec.w), oy = qpermf<kBcast1>(rec.w), oz = qpermf<kBcast2>(rec.w);
                const int e = (int)qperm<kBcast3>(__float_as_uint(rec.w));
                const float ax = ldexpf(inv.x, (int)(signed char)(e & 255)), ay = ldexpf(inv.y, (int)(signed char)((e >> 8) & 255)),
                            az = ldexpf(inv.z, (int)(signed char)((e >> 16) & 255));
                const float bx = fmaf(ox, inv.x, -ood.x), by = fmaf(oy, inv.y, -ood.y), bz = fmaf(oz, inv.z, -ood.z);
                const uint32_t pa = __float_as_uint(rec.x), pb = __float_as_uint(rec.y);
                const uint32_t nn = __builtin_amdgcn_perm(pb, pa, sel_n), ff = __builtin_amdgcn_perm(pb, pa, sel_f);
                const float tnx = fmaf((float)(nn & 255u), ax, bx), tny = fmaf((float)((nn >> 8) & 255u), ay, by), tnz = fmaf((float)((nn >> 16) & 255u), az, bz);
                const float tfx = fmaf((float)(nn >> 24), ax, bx), tfy = fmaf((float)(ff & 255u), ay, by), tfz = fmaf((float)((ff >> 8) & 255u), az, bz);
                const float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
                const float t1 = fminf(fminf(tfx, tfy), fminf(tfz, best_t));
                const bool hit = t0 <= t1 * 1.0000004f;
                const uint32_t key = hit ? ((__float_as_uint(t0) & ~3u) | r) : 0xffffffffu;
                const uint32_t ref = __float_as_uint(rec.z);
                const uint32_t nib = (uint32_t)(__ballot(hit) >> quad_shift) & 15u;       // the quad's four hit flags
                uint32_t next;
                if (nib == 0u) {
                    if (sp == 0) next = 0xffffffffu; else next = stack[--sp];
                } else {
                    uint32_t kmin = min(key, qperm<kXor1>(key));
                    kmin = min(kmin, qperm<kRot2>(kmin));
                    const bool nearest = key == kmin;
                    const uint32_t others = nib & ~(1u << (kmin & 3u));
                    if (hit && !nearest) stack[sp + (int)__popc(others & below)] = ref;
                    sp += (int)__popc(others);
                    uint32_t mine = nearest ? ref : 0u;
                    mine |= qperm<kXor1>(mine);
                    next = mine | qperm<kRot2>(mine);
                }
                if (next == 0xffffffffu) { node = kTravDone; finished = true; }
                else if (next & 0x80000000u) { node = ~(int)((next & 0x7fffffffu) >> 3); leaf_count = (int)(next & 7u); }
                else node = (int)next;
            }
            if (!finished && node < 0) {
                float t = 1e30f, u = 0.f, v = 0.f;
                bool h = false;
                const int tri = ~node + (int)r;
                if ((int)r < leaf_count) {
                    const size_t base = (size_t)tri * 4;
                    const float4 a = ldq(S.tris, base), b = ldq(S.tris, base + 1), c = ldq(S.tris, base + 2);
                    bool cull = false;
                    if (!ANY) cull = (__float_as_uint(ldq(S.tris, base + 3).w) & 0x80000000u) != 0;
                    h = tri_test(a, b, c, o, d, tmin, best_t, cull, t, u, v);
                    if (!h) t = 1e30f;
                    if (STATS) n_tri++;
                }
                if (STATS && r == 0u) n_leaf++;
                if (h && t < my_t) { my_t = t; my_tri = tri; my_u = u; my_v = v; }
                float tq = fminf(t, qpermf<kXor1>(t));
                tq = fminf(tq, qpermf<kRot2>(tq));
                if (ANY) {
                    if (tq < 1e30f) { occluded = true; node = kTravDone; finished = true; }
                } else best_t = fminf(best_t, tq);
                if (!finished) {
                    if (sp == 0) { node = kTravDone; finished = true; }
                    else {
                        const uint32_t w = stack[--sp];
                        if (w & 0x80000000u) { node = ~(int)((w & 0x7fffffffu) >> 3); leaf_count = (int)(w & 7u); }
                        else node = (int)w;
                    }
                }
            }
            if (finished) {
                if (ANY) {
                    if (r == 0u) A.out_visible[ray] = occluded ? 0 : 1;
                } else {
                    const uint32_t cand = (my_tri >= 0 && my_t == best_t) ? r : 4u;
                    uint32_t w = min(cand, qperm<kXor1>(cand));
                    w = min(w, qperm<kRot2>(w));
                    if (w == 4u) { if (r == 0u) { A.out_t[ray] = best_t; A.out_tri[ray] = -1; A.out_uv[2 * ray] = 0.f; A.out_uv[2 * ray + 1] = 0.f; } }
                    else if (w == r) { A.out_t[ray] = my_t; A.out_tri[ray] = S.tri_orig[my_tri]; A.out_uv[2 * ray] = my_u; A.out_uv[2 * ray + 1] = my_v; }
                }
            }
        }
    }
    if (STATS) {
        const unsigned long long v[5] = {n_node, n_leaf, n_tri, n_iter * 64ull, n_busy};
        for (int k = 0; k < 5; k++) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) atomicAdd(&A.stats[k], x);
        }
    }
}

// ---- four lanes per ray, K rays per quad -------------------------------------------------------------------------------------------
// k_trace_quad keeps 16 rays per wave in flight where the lane kernel keeps 64: with one dependent fetch per ray per iteration it is
// latency-bound at a quarter of the lane kernel's memory-level parallelism (measured: 0.29-0.36 x its speed, section 3 of
// profiles/r04_experiments.md).  Here every quad works on K rays at once: the K node fetches of an iteration are issued back to back
// (K x 16 lines per wave in flight), then the K visits are computed one after the other, each at (nearly) full lane utilisation.
// K = 4 restores the 64 rays per wave.  Per-ray state is small (reciprocal direction, origin x reciprocal, interval, node, stack
// pointer); origin and direction are re-read from the ray buffer for the (rare) leaf visits.
template <bool ANY, bool STATS, int K>
__global__ __launch_bounds__(QBLOCK) void k_trace_quadk(const DeviceScene S, const QuadArgs A) {
    constexpr int STACK = 48, STRIDE = STACK + 1;
    __shared__ uint32_t s_stack[QRAYS * K * STRIDE];
    const uint32_t lane = threadIdx.x & 63u, r = lane & 3u;
    uint32_t* const stack0 = s_stack + (threadIdx.x >> 2) * K * STRIDE;
    int node[K], sp[K], leafc[K], ray[K], my_tri[K];
    f3 inv[K], ood[K];
    float tmin[K], best_t[K], my_t[K], my_u[K], my_v[K];
#pragma unroll
    for (int k = 0; k < K; k++) { node[k] = kTravDone; sp[k] = 0; leafc[k] = 0; ray[k] = -1; my_tri[k] = -1; my_t[k] = 1e30f; my_u[k] = my_v[k] = 0.f;
                                  inv[k] = mk3(1.f); ood[k] = mk3(0.f); tmin[k] = 0.f; best_t[k] = 0.f; }
    bool exhausted = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    unsigned long long n_node = 0, n_leaf = 0, n_tri = 0, n_iter = 0, n_busy = 0;
    while (true) {
        if (!exhausted) {   // every empty ray slot of the wave draws from the wave's chunk
#pragma unroll
            for (int k = 0; k < K; k++) {
                const unsigned long long want = __ballot(node[k] == kTravDone && r == 0u);
                if (want && !exhausted) {
                    const uint32_t mine = pool_take(A, chunk_next, chunk_end, exhausted, want, (uint32_t)__popcll(want & ((1ull << (lane & ~3u)) - 1ull)));
                    if (node[k] == kTravDone && mine != 0xffffffffu) {
                        ray[k] = (int)mine;
                        const float4 ra = ldq(A.rays, (size_t)mine * 2), rb = ldq(A.rays, (size_t)mine * 2 + 1);
                        const f3 o = mk3(ra.x, ra.y, ra.z), d = mk3(rb.x, rb.y, rb.z);
                        tmin[k] = ra.w; best_t[k] = rb.w;
                        inv[k] = safe_inv(d); ood[k] = o * inv[k];
                        node[k] = 0; sp[k] = 0; leafc[k] = 0;
                        my_t[k] = 1e30f; my_tri[k] = -1;
                    }
                }
            }
        }
        bool live = false;
#pragma unroll
        for (int k = 0; k < K; k++) live = live || node[k] != kTravDone;
        if (!__any(live)) break;
        if (STATS) {
            n_iter += lane == 0 ? (unsigned long long)K : 0ull;
#pragma unroll
            for (int k = 0; k < K; k++) n_busy += node[k] != kTravDone ? 1 : 0;
        }
        // ---- the K node fetches of this iteration, back to back
        float4 rec[K];
#pragma unroll
        for (int k = 0; k < K; k++)
            if (node[k] != kTravDone && node[k] >= 0) rec[k] = ldq(reinterpret_cast<const float*>(A.nodes_q), (size_t)node[k] * 4 + r);
#pragma unroll
        for (int k = 0; k < K; k++) {
            uint32_t* const stack = stack0 + k * STRIDE;
            bool finished = false;
            if (node[k] != kTravDone && node[k] >= 0) {
                if (STATS && r == 0u) n_node++;
                const float ox = qpermf<kBcast0>(rec[k].w), oy = qpermf<kBcast1>(rec[k].w), oz = qpermf<kBcast2>(rec[k].w);
                const uint32_t e = qperm<kBcast3>(__float_as_uint(rec[k].w));
                const f3 iv = inv[k], od = ood[k];
                const float ax = __uint_as_float((e & 0xffu) << 23) * iv.x, ay = __uint_as_float(((e >> 8) & 0xffu) << 23) * iv.y,
                            az = __uint_as_float(((e >> 16) & 0xffu) << 23) * iv.z;
                const float bx = fmaf(ox, iv.x, -od.x), by = fmaf(oy, iv.y, -od.y), bz = fmaf(oz, iv.z, -od.z);
                const uint32_t pa = __float_as_uint(rec[k].x), pb = __float_as_uint(rec[k].y);
                const float lx = (float)(pa & 255u), ly = (float)((pa >> 8) & 255u), lz = (float)((pa >> 16) & 255u), hx = (float)(pa >> 24),
                            hy = (float)(pb & 255u), hz = (float)((pb >> 8) & 255u);
                const bool sx = iv.x < 0.0f, sy = iv.y < 0.0f, sz = iv.z < 0.0f;
                const float tnx = fmaf(sx ? hx : lx, ax, bx), tfx = fmaf(sx ? lx : hx, ax, bx);
                const float tny = fmaf(sy ? hy : ly, ay, by), tfy = fmaf(sy ? ly : hy, ay, by);
                const float tnz = fmaf(sz ? hz : lz, az, bz), tfz = fmaf(sz ? lz : hz, az, bz);
                const float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin[k]));
                const float t1 = fminf(fminf(tfx, tfy), fminf(tfz, best_t[k]));
                const uint32_t key = (t0 <= t1 * 1.0000004f) ? ((__float_as_uint(t0) & ~3u) | r) : 0xffffffffu;
                const uint32_t ref = __float_as_uint(rec[k].z);
                const uint32_t k1 = qperm<kRot1>(key), k2 = qperm<kRot2>(key), k3 = qperm<kRot3>(key);
                const bool hit = key != 0xffffffffu;
                const int rank = (k1 < key ? 1 : 0) + (k2 < key ? 1 : 0) + (k3 < key ? 1 : 0);
                const int nh = (hit ? 1 : 0) + (k1 != 0xffffffffu ? 1 : 0) + (k2 != 0xffffffffu ? 1 : 0) + (k3 != 0xffffffffu ? 1 : 0);
                uint32_t next;
                if (nh == 0) {
                    if (sp[k] == 0) next = 0xffffffffu; else next = stack[--sp[k]];
                } else {
                    if (hit && rank > 0) stack[sp[k] + nh - 1 - rank] = ref;
                    sp[k] += nh - 1;
                    const uint32_t mine = (hit && rank == 0) ? ref : 0u;
                    next = mine | qperm<kRot1>(mine) | qperm<kRot2>(mine) | qperm<kRot3>(mine);
                }
                if (next == 0xffffffffu) { node[k] = kTravDone; finished = true; }
                else if (next & 0x80000000u) { node[k] = ~(int)((next & 0x7fffffffu) >> 3); leafc[k] = (int)(next & 7u); }
                else node[k] = (int)next;
            }
            if (!finished && node[k] < 0) {
                float t = 1e30f, u = 0.f, v = 0.f;
                bool h = false;
                const int tri = ~node[k] + (int)r;
                const float4 ra = ldq(A.rays, (size_t)ray[k] * 2), rb = ldq(A.rays, (size_t)ray[k] * 2 + 1);   // origin, direction: leaf visits only
                if ((int)r < leafc[k]) {
                    const size_t base = (size_t)tri * 4;
                    const float4 a = ldq(S.tris, base), b = ldq(S.tris, base + 1), c = ldq(S.tris, base + 2);
                    bool cull = false;
                    if (!ANY) cull = (__float_as_uint(ldq(S.tris, base + 3).w) & 0x80000000u) != 0;
                    h = tri_test(a, b, c, mk3(ra.x, ra.y, ra.z), mk3(rb.x, rb.y, rb.z), tmin[k], best_t[k], cull, t, u, v);
                    if (!h) t = 1e30f;
                    if (STATS) n_tri++;
                }
                if (STATS && r == 0u) n_leaf++;
                if (h && t < my_t[k]) { my_t[k] = t; my_tri[k] = tri; my_u[k] = u; my_v[k] = v; }
                float tq = fminf(t, qpermf<kXor1>(t));
                tq = fminf(tq, qpermf<kRot2>(tq));
                bool occluded = false;
                if (ANY) {
                    if (tq < 1e30f) { occluded = true; node[k] = kTravDone; finished = true; }
                } else best_t[k] = fminf(best_t[k], tq);
                if (!finished) {
                    if (sp[k] == 0) { node[k] = kTravDone; finished = true; }
                    else {
                        const uint32_t w = stack[--sp[k]];
                        if (w & 0x80000000u) { node[k] = ~(int)((w & 0x7fffffffu) >> 3); leafc[k] = (int)(w & 7u); }
                        else node[k] = (int)w;
                    }
                }
                if (ANY && finished && r == 0u) A.out_visible[ray[k]] = occluded ? 0 : 1;
            } else if (ANY && finished && r == 0u) A.out_visible[ray[k]] = 1;   // the stack ran empty on a node visit: nothing hit
            if (!ANY && finished) {
                const uint32_t cand = (my_tri[k] >= 0 && my_t[k] == best_t[k]) ? r : 4u;
                uint32_t w = min(cand, qperm<kXor1>(cand));
                w = min(w, qperm<kRot2>(w));
                const int rk = ray[k];
                if (w == 4u) { if (r == 0u) { A.out_t[rk] = best_t[k]; A.out_tri[rk] = -1; A.out_uv[2 * rk] = 0.f; A.out_uv[2 * rk + 1] = 0.f; } }
                else if (w == r) { A.out_t[rk] = my_t[k]; A.out_tri[rk] = S.tri_orig[my_tri[k]]; A.out_uv[2 * rk] = my_u[k]; A.out_uv[2 * rk + 1] = my_v[k]; }
            }
        }
    }
    if (STATS) {
        const unsigned long long v[5] = {n_node, n_leaf, n_tri, n_iter * 64ull, n_busy};
        for (int k = 0; k < 5; k++) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) atomicAdd(&A.stats[k], x);
        }
    }
}

// ---- one lane per ray, pool-fed: the loop of device_lib.h (SPC_NODE_STEP, tri_test, TravStack) as the megakernel's trace_pool runs it
template <bool ANY, bool STATS>
__global__ __launch_bounds__(QBLOCK) void k_trace_lane(const KParams p, const QuadArgs A) {
    __shared__ uint32_t s_stack[QBLOCK * kStackLds];
    const DeviceScene& S = p.scene;
    constexpr bool COUNT = false;
    Counts<false> cn;
    TravStack<QBLOCK, kStackLds> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)blockIdx.x * QBLOCK + threadIdx.x, p.diag);
    const uint32_t lane = threadIdx.x & 63u;
    int node = kTravDone, leaf_count = 0, ray = -1, best_tri = -1;
    f3 o = mk3(0.f), d = mk3(0.f), inv = mk3(1.f), ood = mk3(0.f);
    float tmin = 0.f, best_t = 0.f, best_u = 0.f, best_v = 0.f;
    bool exhausted = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    unsigned long long n_node = 0, n_leaf = 0, n_tri = 0, n_iter = 0, n_busy = 0;
    while (true) {
        const bool need = node == kTravDone;
        if (!exhausted) {
            const unsigned long long want = __ballot(need);
            if (want) {
                const uint32_t mine = pool_take(A, chunk_next, chunk_end, exhausted, want, (uint32_t)__popcll(want & ((1ull << lane) - 1ull)));
                if (need && mine != 0xffffffffu) {
                    ray = (int)mine;
                    const float4 ra = ldq(A.rays, (size_t)ray * 2), rb = ldq(A.rays, (size_t)ray * 2 + 1);
                    o = mk3(ra.x, ra.y, ra.z); d = mk3(rb.x, rb.y, rb.z);
                    tmin = ra.w; best_t = rb.w;
                    inv = safe_inv(d); ood = o * inv;
                    node = 0; st.sp = 0; leaf_count = 0; best_tri = -1; best_u = best_v = 0.0f;
                }
            }
        }
        if (!__any(node != kTravDone)) break;
        if (STATS) { n_iter += lane == 0 ? 1 : 0; n_busy += node != kTravDone ? 1 : 0; }
        if (node != kTravDone) {
            bool finished = false, occluded = false;
            if (node >= 0) { if (STATS) n_node++; SPC_NODE_STEP(tmin, best_t); finished = node == kTravDone; }
            if (node < 0 && leaf_count <= 0) {
                SPC_TRAV_POP();
                finished = node == kTravDone;
            } else if (node < 0) {
                const int tri = ~node;
                const size_t base = (size_t)tri * 4;
                const float4 a = ldq(S.tris, base), b = ldq(S.tris, base + 1), c = ldq(S.tris, base + 2);
                if (STATS) n_tri++;
                bool cull = false;
                if (!ANY) cull = (__float_as_uint(ldq(S.tris, base + 3).w) & 0x80000000u) != 0;
                float t, u, v;
                const bool h = tri_test(a, b, c, o, d, tmin, best_t, cull, t, u, v);
                if (h && ANY) { occluded = true; finished = true; node = kTravDone; }
                else {
                    if (h) { best_t = t; best_tri = tri; best_u = u; best_v = v; }
                    node -= 1;
                    leaf_count -= 1;
                    if (leaf_count == 0) { if (STATS) n_leaf++; SPC_TRAV_POP(); finished = node == kTravDone; }
                }
            }
            if (finished) {
                if (ANY) A.out_visible[ray] = occluded ? 0 : 1;
                else { A.out_t[ray] = best_t; A.out_tri[ray] = best_tri >= 0 ? S.tri_orig[best_tri] : -1; A.out_uv[2 * ray] = best_u; A.out_uv[2 * ray + 1] = best_v; }
            }
        }
    }
    if (STATS) {
        const unsigned long long v[5] = {n_node, n_leaf, n_tri, n_iter * 64ull, n_busy};
        for (int k = 0; k < 5; k++) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) atomicAdd(&A.stats[k], x);
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
void launch_repack_nodes_quad2(const float* nodes_q, float* out, int n_nodes, hipStream_t s) {
    hipLaunchKernelGGL(k_repack_nodes_quad2, dim3((n_nodes + 255) / 256), dim3(256), 0, s, reinterpret_cast<const float4*>(nodes_q), reinterpret_cast<float4*>(out), n_nodes);
}
void launch_repack_nodes_quad(const float* nodes, float* out, int n_nodes, hipStream_t s) {
    hipLaunchKernelGGL(k_repack_nodes_quad, dim3((n_nodes + 255) / 256), dim3(256), 0, s, reinterpret_cast<const float4*>(nodes), reinterpret_cast<float4*>(out), n_nodes);
}
template <int K>
static int quadk_blocks(bool any) {
    int n = 0;
    hipError_t e = any ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_quadk<true, false, K>, QBLOCK, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_quadk<false, false, K>, QBLOCK, 0);
    return e == hipSuccess && n > 0 ? n : 1;
}
template <int K>
static void quadk_launch(bool any, bool stats, int blocks, hipStream_t s, const DeviceScene& S, const QuadArgs& A) {
    if (any) { if (stats) hipLaunchKernelGGL((k_trace_quadk<true, true, K>), dim3(blocks), dim3(QBLOCK), 0, s, S, A); else hipLaunchKernelGGL((k_trace_quadk<true, false, K>), dim3(blocks), dim3(QBLOCK), 0, s, S, A); }
    else { if (stats) hipLaunchKernelGGL((k_trace_quadk<false, true, K>), dim3(blocks), dim3(QBLOCK), 0, s, S, A); else hipLaunchKernelGGL((k_trace_quadk<false, false, K>), dim3(blocks), dim3(QBLOCK), 0, s, S, A); }
}
int trace_bench_blocks_per_cu(int mode, bool any) {
    int n = 0;
    hipError_t e;
    if (mode == 4) {
        hipError_t e4 = any ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_quad2<true, false>, QBLOCK, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_quad2<false, false>, QBLOCK, 0);
        return e4 == hipSuccess && n > 0 ? n : 1;
    }
    if (mode == 2) return quadk_blocks<2>(any);
    if (mode == 3) return quadk_blocks<4>(any);
    if (mode == 1) e = any ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_quad<true, false>, QBLOCK, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_quad<false, false>, QBLOCK, 0);
    else e = any ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_lane<true, false>, QBLOCK, 0) : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace_lane<false, false>, QBLOCK, 0);
    return e == hipSuccess && n > 0 ? n : 1;
}
void launch_trace_bench(const KParams& p, int mode, bool any, bool stats, const float* nodes_q, const float* rays, int n, uint32_t* counter, float* t, int* tri,
                        float* uv, int* vis, unsigned long long* stat_out, int blocks, hipStream_t s) {
    QuadArgs A;
    A.nodes_q = reinterpret_cast<const float4*>(nodes_q); A.rays = rays; A.n = n; A.counter = counter;
    A.out_t = t; A.out_tri = tri; A.out_uv = uv; A.out_visible = vis; A.stats = stat_out;
    const long long waves = (long long)blocks * (QBLOCK / 64);
    A.chunk = (uint32_t)std::max(64LL, std::min(1024LL, (long long)n / std::max(1LL, waves * 4)));
    if (mode == 4) {
        if (any) { if (stats) hipLaunchKernelGGL((k_trace_quad2<true, true>), dim3(blocks), dim3(QBLOCK), 0, s, p.scene, A); else hipLaunchKernelGGL((k_trace_quad2<true, false>), dim3(blocks), dim3(QBLOCK), 0, s, p.scene, A); }
        else { if (stats) hipLaunchKernelGGL((k_trace_quad2<false, true>), dim3(blocks), dim3(QBLOCK), 0, s, p.scene, A); else hipLaunchKernelGGL((k_trace_quad2<false, false>), dim3(blocks), dim3(QBLOCK), 0, s, p.scene, A); }
    }
    else if (mode == 2) quadk_launch<2>(any, stats, blocks, s, p.scene, A);
    else if (mode == 3) quadk_launch<4>(any, stats, blocks, s, p.scene, A);
    else if (mode == 1) {
        if (any) { if (stats) hipLaunchKernelGGL((k_trace_quad<true, true>), dim3(blocks), dim3(QBLOCK), 0, s, p.scene, A); else hipLaunchKernelGGL((k_trace_quad<true, false>), dim3(blocks), dim3(QBLOCK), 0, s, p.scene, A); }
        else { if (stats) hipLaunchKernelGGL((k_trace_quad<false, true>), dim3(blocks), dim3(QBLOCK), 0, s, p.scene, A); else hipLaunchKernelGGL((k_trace_quad<false, false>), dim3(blocks), dim3(QBLOCK), 0, s, p.scene, A); }
    } else {
        if (any) { if (stats) hipLaunchKernelGGL((k_trace_lane<true, true>), dim3(blocks), dim3(QBLOCK), 0, s, p, A); else hipLaunchKernelGGL((k_trace_lane<true, false>), dim3(blocks), dim3(QBLOCK), 0, s, p, A); }
        else { if (stats) hipLaunchKernelGGL((k_trace_lane<false, true>), dim3(blocks), dim3(QBLOCK), 0, s, p, A); else hipLaunchKernelGGL((k_trace_lane<false, false>), dim3(blocks), dim3(QBLOCK), 0, s, p, A); }
    }
}

}  // namespace spc

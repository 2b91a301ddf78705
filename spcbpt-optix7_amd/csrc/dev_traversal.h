// Part of device_lib.h (split in round 6 for readability; included by it, in this order, inside the one translation unit of each
// .hip file -- the device code generated is the same as from the single header: tests/test_codegen_guard.py):
// software BVH traversal: the per-lane stack (LDS + HBM part), slab and triangle tests, traverse<> (one ray per lane to its end), the quad and fan-out tails, trace_pool (the pooled pass of k_spcbpt).
#pragma once
#include "device_lib.h"

namespace spc {

// ---- software LBVH traversal: per-lane stack in LDS --------------------------
// The stack is laid out entry-major ([entry][thread]) so the 64 lanes of a wave hit 64 consecutive
// dwords = all LDS banks, conflict-free.  Entries beyond STACK_LDS spill to HBM (rare: LBVH depth).
// The LDS column of a lane is NOT kept in a register: under the megakernel's pressure the allocator spilled it, and every push
// and pop of the hot loop then began with a scratch reload and a vmcnt(0) wait that also drained the node fetches in
// flight (measured: 16 M extra VMEM instructions and +1 ms per frame).  It is re-derived where needed from the wave's base
// (wave-uniform, an SGPR) and the lane id (two v_mbcnt, volatile so that the result is never a long-lived value).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
SPC_DEV uint32_t lane_id_fresh() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// The HBM part of the traversal stack (entries past the LDS ones): rare, and kept OUT of line so that its address arithmetic, its
// bounds logic and the overflow report do not sit in the traversal loop three times over (one copy per push site).
__device__ __attribute__((noinline)) static void stack_push_slow(uint32_t* spill, int spill_entries, int idx, uint32_t v, uint32_t* diag) {
    if (spill && idx < spill_entries) spill[idx] = v;
    else if (diag) atomicAdd(diag, 1u);   // the subtree is lost: never silently (spcbpt_sync and the read-backs fail)
}
__device__ __attribute__((noinline)) static uint32_t stack_pop_slow(const uint32_t* spill, int spill_entries, int idx) {
    if (spill && idx < spill_entries) return spill[idx];
    return NODE_EMPTY;  // the entry push() had to drop (and counted in diag[0]): a leaf of zero triangles, nothing is read
}
template <int BLOCK, int STACK_LDS>
struct TravStack {
    lds_u32* wave_lds;  // column 0 of this wave inside the BLOCK * STACK_LDS dword array (wave-uniform)
    uint32_t* spill;    // per-thread spill area or null
    uint32_t* diag;     // KParams::diag: [0] counts entries that fit neither LDS nor the spill area (host reports an error)
    int spill_entries;
    int sp;
    SPC_DEV void init(uint32_t* l, uint32_t* s, int se, size_t gtid, uint32_t* dg) {
        wave_lds = (lds_u32*)l + __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
        spill = s ? s + gtid * (size_t)se : nullptr;
        spill_entries = se;
        diag = dg;
        sp = 0;
    }
    SPC_DEV lds_u32* column() const { return wave_lds + lane_id_fresh(); }
    SPC_DEV void push(uint32_t v) {
        if (sp < STACK_LDS) column()[sp * BLOCK] = v;
        else stack_push_slow(spill, spill_entries, sp - STACK_LDS, v, diag);
        sp++;
    }
    // pushes the (up to three) farther children of a node visit, farthest first; c1 >= c2 >= c3 (hits are sorted to the front)
    SPC_DEV void push_far(uint32_t r1, bool c1, uint32_t r2, bool c2, uint32_t r3, bool c3) {
        if (sp + 3 <= STACK_LDS) {  // common case: straight LDS stores at computed slots, no per-entry bounds logic
            lds_u32* lds = column();
            if (c3) lds[sp * BLOCK] = r3;
            const int p2 = sp + (c3 ? 1 : 0);
            if (c2) lds[p2 * BLOCK] = r2;
            const int p1 = p2 + (c2 ? 1 : 0);
            if (c1) lds[p1 * BLOCK] = r1;
            sp = p1 + (c1 ? 1 : 0);
        } else {
            if (c3) push(r3);
            if (c2) push(r2);
            if (c1) push(r1);
        }
    }
    SPC_DEV uint32_t pop() {
        sp--;
        if (sp < STACK_LDS) return column()[sp * BLOCK];
        return stack_pop_slow(spill, spill_entries, sp - STACK_LDS);
    }
    // ... of a step whose caller has established that no lane of the wave is within three entries of the LDS part's end (trace_pool
    // votes once per iteration): no bounds logic, no call sites of the HBM part -- ~30 instructions of every iteration
    SPC_DEV void push_far_lds(uint32_t r1, bool c1, uint32_t r2, bool c2, uint32_t r3, bool c3) {
        lds_u32* lds = column();
        if (c3) lds[sp * BLOCK] = r3;
        const int p2 = sp + (c3 ? 1 : 0);
        if (c2) lds[p2 * BLOCK] = r2;
        const int p1 = p2 + (c2 ? 1 : 0);
        if (c1) lds[p1 * BLOCK] = r1;
        sp = p1 + (c1 ? 1 : 0);
    }
    SPC_DEV uint32_t pop_lds() { sp--; return column()[sp * BLOCK]; }
};

struct HitRec { float t; int tri; float u, v; };

// Reciprocal direction of the slab tests.  v_rcp_f32 (1 ulp) instead of the IEEE division sequence (10 instructions per
// component): the reciprocal only decides which quantised child boxes are entered, and those are rounded outwards by up to 1/255
// of the node and tested with a relative slack (slab4q), so the last bit cannot lose a hit -- triangle tests use o and d.  The
// ray set-up runs once per ray with a handful of lanes active (lanes pull rays from the pool as they finish), i.e. its
// instructions are paid by the whole wave at ~5 % utilisation: 55 -> 25 instructions there was worth 3 % of the kernel.
SPC_DEV f3 safe_inv(f3 d) {
    const float tiny = 1e-20f;
    f3 r;
    r.x = __builtin_amdgcn_rcpf(fabsf(d.x) > tiny ? d.x : copysignf(tiny, d.x));
    r.y = __builtin_amdgcn_rcpf(fabsf(d.y) > tiny ? d.y : copysignf(tiny, d.y));
    r.z = __builtin_amdgcn_rcpf(fabsf(d.z) > tiny ? d.z : copysignf(tiny, d.z));
    return r;
}
SPC_DEV bool slab(float4 lo, float4 hi, f3 o, f3 inv, float tmin, float tmax, float& tnear) {
    float tx0 = (lo.x - o.x) * inv.x, tx1 = (hi.x - o.x) * inv.x;
    float ty0 = (lo.y - o.y) * inv.y, ty1 = (hi.y - o.y) * inv.y;
    float tz0 = (lo.z - o.z) * inv.z, tz1 = (hi.z - o.z) * inv.z;
    float t0 = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
    float t1 = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tmax));
    tnear = t0;
    return t0 <= t1 * 1.0000004f;
}
// Moller-Trumbore on (P0, P1, P2); accepts tmin < t < tmax; culls the back face when asked (emitter quads).
// The triangle step of the traversal loop runs at ~10 % lane utilisation (a lane sits on a leaf in one iteration out of ten, and
// nearly every iteration has SOME lane on one), so every instruction here is paid by the whole wave: the cross products may
// contract to FMAs (unlike cross(), whose exact zeros only matter for shading normals), 1 / det is v_rcp_f32 (1 ulp) instead of
// the IEEE division sequence, and the back-face test reuses the determinant: dot(cross(e1, e2), d) = -det.
SPC_DEV f3 cross_fma(f3 a, f3 b) {
    return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
// EDGES = true: q1, q2 hold the edges P1 - P0, P2 - P0 already (the pair slots of lbvh.h: the same FP32 differences, taken on the host)
template <bool EDGES = false>
SPC_DEV bool tri_test(float4 q0, float4 q1, float4 q2, f3 o, f3 d, float tmin, float tmax, bool cull, float& ot, float& ou, float& ov) {
    const f3 v0 = mk3(q0.x, q0.y, q0.z);
    const f3 e1 = EDGES ? mk3(q1.x, q1.y, q1.z) : mk3(q1.x, q1.y, q1.z) - v0, e2 = EDGES ? mk3(q2.x, q2.y, q2.z) : mk3(q2.x, q2.y, q2.z) - v0;
    const f3 p = cross_fma(d, e2);
    const float det = dot(e1, p);
    if (det == 0.0f || (cull && det < 0.0f)) return false;
    const float inv = __builtin_amdgcn_rcpf(det);
    const f3 tv = o - v0;
    const float u = dot(tv, p) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    const f3 q = cross_fma(tv, e1);
    const float v = dot(d, q) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = dot(e2, q) * inv;
    if (!(t > tmin && t < tmax)) return false;
    ot = t; ou = u; ov = v;
    return true;
}

// ANY = terminate on first hit, no culling (visibilityTest); else nearest hit with emitter back-face culling.
// "while-while" traversal: all lanes first descend through internal nodes (lanes that already sit on a leaf wait), then
// the wave processes leaves together, so the two code paths are not interleaved per iteration inside a divergent wave.
static constexpr int kTravDone = 0x7fffffff;
#ifndef SPC_QUAD_TAIL
#define SPC_ONE_FETCH 1
#ifndef SPC_PRIO_TAIL
#define SPC_PRIO_TAIL -1  // >= 0: s_setprio at the entry of the quad / fan tails (kernels.hip sets the pass's and resets after it)
#endif
#ifndef SPC_TRI_BATCH
#define SPC_TRI_BATCH 8   // trace_pool: N > 1 = lanes on a leaf wait until N of them are (or nobody is on an internal node) before the triangle step
#endif
#ifndef SPC_PROBE_DROP_TAIL
#define SPC_PROBE_DROP_TAIL 0
#endif
#ifndef SPC_TRI_PAIRS
#define SPC_TRI_PAIRS 1   // trace_pool: a lane's triangle step tests both halves of a fan pair (lbvh.h: Lbvh::pairs); 0 = one triangle per step (the slot's first three corners)
#endif
#ifndef SPC_TRI_PAIRS_LANE
#define SPC_TRI_PAIRS_LANE 1   // traverse<> (one ray per lane to its end: "pt", the light pass, the pre-trace): the same pair step
#endif
#ifndef SPC_PROBE_TRI_PAIRS
#define SPC_PROBE_TRI_PAIRS 0
#endif
#ifndef SPC_ROOT_AHEAD
#define SPC_ROOT_AHEAD 1  // trace_pool: a lane that is about to draw a ray requests the root with the other lanes' next records
#endif
#define SPC_QUAD_TAIL 1   // the last <= 16 rays of a pooled pass continue on four lanes each (trace_pool); 0 = the lane loop to the end
#define SPC_FAN_TAIL 1    // ... and its shadow rays on as many quads as the wave has idle (fan_tail); 0 = one quad per ray to the end
#endif
// pop the next stack entry into (node, leaf_count); leaf refs carry their count: 1<<31 | first<<3 | count (count <= 4).
// A macro, not a lambda: a by-reference capture keeps node / leaf_count in scratch memory inside the loop.
#define SPC_TRAV_POP() SPC_TRAV_POP_(pop)
#define SPC_TRAV_POP_(POP)                                                                          \
    do {                                                                                            \
        if (st.sp == 0) { node = kTravDone; }                                                       \
        else {                                                                                      \
            const uint32_t w__ = st.POP();                                                          \
            if (w__ & 0x80000000u) { node = ~(int)((w__ & 0x7fffffffu) >> 3); leaf_count = (int)(w__ & 7u); } \
            else node = (int)w__;                                                                   \
        }                                                                                           \
    } while (0)
// slab test of the 4 quantised child boxes of a node; misses get key 0xffffffff, hits the entry distance with the slot index
// in the two low mantissa bits (t >= 0, so unsigned order = float order).  plane distance = (org + q s - o) / d
// = q (s inv) + (org inv - o inv): two per-axis constants per node, then one byte->float convert and one FMA per plane.
// The near/far plane of each axis is picked once per node from the sign of the ray direction (swap of the lo/hi byte
// quads), so no per-child min/max is needed, and an empty slot (qlo = 255, qhi = 0: an inverted box) misses by itself.
SPC_DEV void slab4q(const float4 q0, const float4 q1, const float4 q2, f3 ood, f3 inv, float tmin, float tmax, uint32_t key[4]) {
    const uint32_t e = __float_as_uint(q0.w);
    const float ax = __uint_as_float((e & 0xffu) << 23) * inv.x, ay = __uint_as_float(((e >> 8) & 0xffu) << 23) * inv.y,
                az = __uint_as_float(((e >> 16) & 0xffu) << 23) * inv.z;
    const float bx = fmaf(q0.x, inv.x, -ood.x), by = fmaf(q0.y, inv.y, -ood.y), bz = fmaf(q0.z, inv.z, -ood.z);
    const uint32_t lxb = __float_as_uint(q1.x), lyb = __float_as_uint(q1.y), lzb = __float_as_uint(q1.z);
    const uint32_t hxb = __float_as_uint(q1.w), hyb = __float_as_uint(q2.x), hzb = __float_as_uint(q2.y);
    const bool sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
    const uint32_t nxb = sx ? hxb : lxb, fxb = sx ? lxb : hxb;
    const uint32_t nyb = sy ? hyb : lyb, fyb = sy ? lyb : hyb;
    const uint32_t nzb = sz ? hzb : lzb, fzb = sz ? lzb : hzb;
    // near and far plane of an axis share the per-node constants: one packed FMA (v_pk_fma_f32) yields both distances
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const v2f qx = {(float)((nxb >> (8 * i)) & 0xffu), (float)((fxb >> (8 * i)) & 0xffu)};
        const v2f qy = {(float)((nyb >> (8 * i)) & 0xffu), (float)((fyb >> (8 * i)) & 0xffu)};
        const v2f qz = {(float)((nzb >> (8 * i)) & 0xffu), (float)((fzb >> (8 * i)) & 0xffu)};
        const v2f tx = __builtin_elementwise_fma(qx, ax2, bx2), ty = __builtin_elementwise_fma(qy, ay2, by2),
                  tz = __builtin_elementwise_fma(qz, az2, bz2);
        const float t0 = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tmin));
        const float t1 = fminf(fminf(tx.y, ty.y), fminf(tz.y, tmax));
        key[i] = (t0 <= t1 * 1.0000004f) ? ((__float_as_uint(t0) & ~3u) | (uint32_t)i) : 0xffffffffu;
    }
}
SPC_DEV uint32_t sel4u(const uint32_t r[4], uint32_t i) {  // two-level select: three v_cndmask, no control flow
    const uint32_t a = (i & 1u) ? r[1] : r[0];
    const uint32_t b = (i & 1u) ? r[3] : r[2];
    return (i & 2u) ? b : a;
}
SPC_DEV int sel4i(const float4 q, uint32_t i) {
    const float v = i == 0 ? q.x : (i == 1 ? q.y : (i == 2 ? q.z : q.w));
    return __float_as_int(v);
}
SPC_DEV uint32_t stack_word(int ref, int count) {
    return ref >= 0 ? (uint32_t)ref : (0x80000000u | ((uint32_t)(~ref) << 3) | (uint32_t)count);
}

// One node visit of the current lane: slab-test the four children, continue with the nearest hit, push the others.
// Macros, not lambdas/functions taking references: see SPC_TRAV_POP.  Uses o/d-derived `inv`, `ood`, the ray interval
// (TMIN, TMAX) and the traversal state `node`, `leaf_count`, `st` of the enclosing scope.
#define SPC_UTIL_COUNT(lanes_slot, slots_slot)                                                                        \
    if (COUNT) { cn.add(lanes_slot); if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) cn.add(slots_slot, 64); }
#define SPC_NODE_STEP(TMIN, TMAX)                                                                                     \
    do {                                                                                                              \
        const size_t nb__ = (size_t)node * NODE_QUADS;                                                                \
        const float4 Q0 = ldq(S.nodes, nb__ + 0), Q1 = ldq(S.nodes, nb__ + 1), Q2 = ldq(S.nodes, nb__ + 2),            \
                     Q3 = ldq(S.nodes, nb__ + 3);                                                                     \
        SPC_NODE_STEP_Q(TMIN, TMAX, Q0, Q1, Q2, Q3, push_far, pop);                                                   \
    } while (0)
/* ... on a node record that is already in registers; PUSH / POP name the stack operations (the plain ones, or the LDS-only ones) */ \
#define SPC_NODE_STEP_Q(TMIN, TMAX, Q0, Q1, Q2, Q3, PUSH, POP)                                                        \
    do {                                                                                                              \
        cn.add(C_NODE); /* one 64-B visit */                                                                          \
        SPC_UTIL_COUNT(C_U_NODE_LANES, C_U_NODE_SLOTS)                                                                \
        const uint32_t ref__[4] = {__float_as_uint(Q2.z), __float_as_uint(Q2.w), __float_as_uint(Q3.x),               \
                                   __float_as_uint(Q3.y)};                                                            \
        uint32_t k__[4];                                                                                              \
        slab4q(Q0, Q1, Q2, ood, inv, TMIN, TMAX, k__);                                                                \
        /* sort the four keys ascending: nearest child first (5 compare-exchanges); misses (0xffffffff) end up last */ \
        SPC_CSWAP__(0, 1) SPC_CSWAP__(2, 3) SPC_CSWAP__(0, 2) SPC_CSWAP__(1, 3) SPC_CSWAP__(1, 2)                      \
        const uint32_t r0__ = sel4u(ref__, k__[0] & 3u), r1__ = sel4u(ref__, k__[1] & 3u),                            \
                       r2__ = sel4u(ref__, k__[2] & 3u), r3__ = sel4u(ref__, k__[3] & 3u);                            \
        if (k__[0] == 0xffffffffu) {                                                                                  \
            SPC_TRAV_POP_(POP);                                                                                       \
        } else {                                                                                                      \
            st.PUSH(r1__, k__[1] != 0xffffffffu, r2__, k__[2] != 0xffffffffu, r3__, k__[3] != 0xffffffffu);           \
            if (r0__ & 0x80000000u) { node = ~(int)((r0__ & 0x7fffffffu) >> 3); leaf_count = (int)(r0__ & 7u); }      \
            else node = (int)r0__;                                                                                    \
        }                                                                                                             \
    } while (0)
#define SPC_CSWAP__(a, b) { const uint32_t lo__ = min(k__[a], k__[b]), hi__ = max(k__[a], k__[b]); k__[a] = lo__; k__[b] = hi__; }

// ANY = terminate on first hit, no culling (visibilityTest); else nearest hit with emitter back-face culling.
// Schedule: "if-if" -- every iteration of the wave does one node visit for the lanes that sit on an internal node and then
// one triangle test for the lanes that sit on a leaf.  With 64-wide waves the classic "while-while" schedule (descend
// until every lane has a leaf) left 77 % of the lane slots of the node loop idle on this workload (measured with the
// C_U_* counters); if-if bounds a lane's wait to one step of the other kind.  Holding the triangle step back until N lanes
// wait on a leaf was measured too (bench scene, ms per frame): N = 1 (plain if-if) 11.10, 8 -> 11.73, 16 -> 11.94, 32 -> 12.92.
template <bool ANY, bool COUNT, int BLOCK, int STACK_LDS>
SPC_DEV bool traverse(const DeviceScene& S, TravStack<BLOCK, STACK_LDS>& st, f3 o, f3 d, float tmin, float tmax, HitRec& hit,
                      Counts<COUNT>& cn) {
    const f3 inv = safe_inv(d);
    const f3 ood = o * inv;
    float best_t = tmax, best_u = 0.0f, best_v = 0.0f;
    int best_tri = -1;
    st.sp = 0;
    int node = 0;        // >= 0 internal node, < 0 leaf (~next triangle to test), kTravDone = finished
    int leaf_count = 0;  // triangles left in the current leaf
    // (A single 64-B fetch per iteration serving node OR triangle lanes loses HERE, where every lane follows one ray to its end:
    // round 1 14.0 against 11.1 ms per frame; round 4, fetched a step ahead: pt frame 4.29 -> 4.43 ms, light pass 1.13 -> 1.16 ms --
    // a lane that reaches a leaf waits a whole iteration for its first triangle.  It WINS in trace_pool, whose triangle step runs
    // at a tenth of the lanes in 83 % of the iterations: see SPC_ONE_FETCH there.)
    // The iteration in two instantiations, as in trace_pool: LDS-only stack operations while no lane still in the loop is within three
    // entries of the end of its LDS part (a vote per iteration), the plain ones otherwise.
#define SPC_TRAVERSE_ITER__(PUSH, POP)                                                                                \
    do {                                                                                                              \
        if (node >= 0) {                                                                                              \
            const size_t nb__ = (size_t)node * NODE_QUADS;                                                            \
            const float4 Q0 = ldq(S.nodes, nb__ + 0), Q1 = ldq(S.nodes, nb__ + 1), Q2 = ldq(S.nodes, nb__ + 2),        \
                         Q3 = ldq(S.nodes, nb__ + 3);                                                                 \
            SPC_NODE_STEP_Q(tmin, best_t, Q0, Q1, Q2, Q3, PUSH, POP);                                                 \
        }                                                                                                             \
        if (node < 0 && leaf_count <= 0) {                                                                            \
            SPC_TRAV_POP_(POP);  /* an empty slot's zero-triangle leaf (only reachable through rounding): nothing to test */ \
        } else if (node < 0) {                                                                                        \
            const int tri = ~node;                                                                                    \
            cn.add(C_TRI);                                                                                            \
            SPC_UTIL_COUNT(C_U_TRI_LANES, C_U_TRI_SLOTS)                                                              \
            bool h;                                                                                                   \
            int adv__ = 1;                                                                                            \
            float t, u, v;                                                                                            \
            if (SPC_TRI_PAIRS_LANE) {                                                                                 \
                /* the triangle's PAIR slot (lbvh.h; behind the node records): both halves of a quad in one step, A first */ \
                const size_t base = ((size_t)(uint32_t)S.tri_base + (size_t)tri) * 4;                                 \
                const float4 a = ldq(S.nodes, base), b = ldq(S.nodes, base + 1), c = ldq(S.nodes, base + 2), e = ldq(S.nodes, base + 3); \
                const uint32_t fl__ = __float_as_uint(e.w);                                                           \
                h = tri_test<true>(a, b, c, o, d, tmin, best_t, !ANY && (fl__ & 0x80000000u) != 0, t, u, v);                \
                if (h) { best_t = t; best_tri = tri; best_u = u; best_v = v; }                                        \
                if ((fl__ & 1u) != 0) {                                                                               \
                    adv__ = 2;                                                                                        \
                    if (!(ANY && h)) {                                                                                \
                        cn.add(C_TRI);                                                                                \
                        const bool hb__ = tri_test<true>(a, c, e, o, d, tmin, best_t, !ANY && (fl__ & 0x40000000u) != 0, t, u, v); \
                        if (hb__) { best_t = t; best_tri = tri + 1; best_u = u; best_v = v; }                         \
                        h = h || hb__;                                                                                \
                    }                                                                                                 \
                }                                                                                                     \
            } else {                                                                                                  \
                const size_t base = (size_t)tri * 4;                                                                  \
                const float4 a = ldq(S.tris, base), b = ldq(S.tris, base + 1), c = ldq(S.tris, base + 2);             \
                bool cull = false;                                                                                    \
                if (!ANY) {                                                                                           \
                    /* emitter flag lives in quad 3; only fetched for closest-hit rays (single-sided emitters, q16) */ \
                    cull = (__float_as_uint(ldq(S.tris, base + 3).w) & 0x80000000u) != 0;                             \
                }                                                                                                     \
                h = tri_test(a, b, c, o, d, tmin, best_t, cull, t, u, v);                                             \
                if (h) { best_t = t; best_tri = tri; best_u = u; best_v = v; }                                        \
            }                                                                                                         \
            if (ANY && h) {                                                                                           \
                node = kTravDone;                                                                                     \
            } else {                                                                                                  \
                node -= adv__;  /* ~(tri + 1) */                                                                      \
                leaf_count -= adv__;                                                                                  \
                if (leaf_count == 0) SPC_TRAV_POP_(POP);                                                              \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
    while (node != kTravDone) {
        if (!__any(st.sp + 3 > STACK_LDS)) SPC_TRAVERSE_ITER__(push_far_lds, pop_lds);
        else SPC_TRAVERSE_ITER__(push_far, pop);
    }
#undef SPC_TRAVERSE_ITER__
    hit.t = best_t; hit.tri = best_tri; hit.u = best_u; hit.v = best_v;
    return best_tri >= 0;
}

// ---- DPP quad_perm helpers (four consecutive lanes) ---------------------------------------------------------------------
static constexpr int kQBcast0 = 0x00, kQBcast1 = 0x55, kQBcast2 = 0xAA, kQBcast3 = 0xFF;
static constexpr int kQRot1 = 0x39, kQRot2 = 0x4E, kQRot3 = 0x93, kQXor1 = 0xB1;   // [1,2,3,0] [2,3,0,1] [3,0,1,2] [1,0,3,2]
template <int CTRL>
SPC_DEV uint32_t quad_perm(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
SPC_DEV float quad_permf(float v) { return __uint_as_float(quad_perm<CTRL>(__float_as_uint(v))); }

// ---- fan-out tail: the last shadow rays of a wave's pass, each on as many quads as the wave has to spare -----------------
// A shadow ray asks one question (is ANY triangle in the way), so the order in which its nodes are visited is free: the
// traversal stack is a bag.  Once at most 16 rays are left (the quad tail of trace_pool) and none of them is a closest-hit ray,
// ray k's bag moves to ROW k of the wave's stack array (64 entries; what the owner lane had in HBM stays there as the bag's
// second part) and the ray is worked on by G = 4, 8, .. 64 lanes -- the wave's 64 divided by the rays still alive, regrouped each
// time their number halves.  Every quad of a group holds one node: it slab-tests the node's four children, continues with the
// nearest hit and throws the others into the bag; a quad without a node takes one out.  Measured before this existed: 47 % of
// the quad-tail iterations ran with one or two rays -- 56 lanes waiting for a dependent chain of fetches that they can now share.
// An unoccluded ray visits exactly the nodes it visited before (all that its segment touches); an occluded one may find its
// occluder earlier or later.  The pass records the same answers either way, so the film does not change.
template <bool COUNT, int BLOCK, int STACK_LDS>
SPC_DEV void fan_tail(const DeviceScene& S, const TravStack<BLOCK, STACK_LDS>& st, bool live, f3 o, f3 d, float tmax, uint32_t cur, int sp,
                      int owner, uint32_t vis_slot, float4* s_rayw, uint8_t* list, Counts<COUNT>& cn) {
    static_assert(STACK_LDS == 16, "one bag row per ray of the quad tail, four entries per lane in the move");
    constexpr uint32_t NONE = 0xffffffffu;
    const uint32_t lane = lane_id_fresh(), qr = lane & 3u;
    // ---- the owner's LDS entries (rows 0 .. 15 of ITS column) become row (lane >> 2), columns 0 .. 15 -----------------------------
    int row = (int)(lane >> 2);
    const int lcnt = sp < STACK_LDS ? sp : STACK_LDS;
    int hb = sp - lcnt;
    {
        const lds_u32* col = st.wave_lds + owner;
        uint32_t e[4];
#pragma unroll
        for (int j = 0; j < 4; j++) e[j] = (live && (int)qr + 4 * j < lcnt) ? col[((int)qr + 4 * j) * BLOCK] : 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        lds_u32* dst = st.wave_lds + row * BLOCK;
#pragma unroll
        for (int j = 0; j < 4; j++) if (live && (int)qr + 4 * j < lcnt) dst[(int)qr + 4 * j] = e[j];
    }
    sp = lcnt;
    f3 inv = safe_inv(d), ood = o * inv;
    uint32_t gsh = 2u;   // log2 of the lanes per ray (wave-uniform)
    bool first = true;
    while (true) {
        const uint32_t G = 1u << gsh;
        const unsigned long long leaders = __ballot(live && (lane & (G - 1u)) == 0u);
        if (leaders == 0ull) break;
        const uint32_t n = (uint32_t)__popcll(leaders);
        uint32_t gn = gsh;
        while ((n << (gn + 1u)) <= 64u) gn++;
        if (first || gn != gsh) {   // regroup: the j-th ray alive gets lanes j * Gn .. (j + 1) * Gn - 1; its quads keep their nodes
            if (live && (lane & (G - 1u)) == 0u) list[__builtin_amdgcn_mbcnt_hi((uint32_t)(leaders >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)leaders, 0u))] = (uint8_t)lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t grp = lane >> gn, in_g = lane & ((1u << gn) - 1u);
            const bool has = grp < n;
            const int lead = has ? (int)list[grp] : (int)lane;
            const uint32_t c2 = (uint32_t)__shfl((int)cur, lead + (int)(in_g < G ? in_g : 0u), 64);
            cur = (has && in_g < G) ? c2 : NONE;
            o = mk3(__shfl(o.x, lead, 64), __shfl(o.y, lead, 64), __shfl(o.z, lead, 64));
            d = mk3(__shfl(d.x, lead, 64), __shfl(d.y, lead, 64), __shfl(d.z, lead, 64));
            tmax = __shfl(tmax, lead, 64);
            sp = __shfl(sp, lead, 64); hb = __shfl(hb, lead, 64); row = __shfl(row, lead, 64); owner = __shfl(owner, lead, 64);
            vis_slot = (uint32_t)__shfl((int)vis_slot, lead, 64);
            live = has;
            inv = safe_inv(d); ood = o * inv;
            gsh = gn;
            first = false;
            __builtin_amdgcn_wave_barrier();   // the list is read before the next regroup writes it
        }
        if (live) {
            const uint32_t Gc = 1u << gsh, gbase = lane & ~(Gc - 1u);
            const unsigned long long gm = (gsh == 6u ? ~0ull : ((1ull << Gc) - 1ull)) << gbase;
            lds_u32* const bag = st.wave_lds + row * BLOCK;
            // (the owner's HBM area is addressed only inside the two rare paths that use it: no pointer is kept across the loop)
#define SPC_FAN_SPILL__ (st.spill ? st.spill + ((long long)owner - (long long)lane) * (long long)st.spill_entries : nullptr)
            if (sp == 0 && hb > 0) {   // the LDS part is empty: the HBM part (the owner's spilled entries, or this bag's overflow) refills it
                const int nf = hb < (int)Gc ? hb : (int)Gc;
                const int in_g = (int)(lane - gbase);
                if (in_g < nf) bag[in_g] = stack_pop_slow(SPC_FAN_SPILL__, st.spill_entries, hb - 1 - in_g);
                sp = nf; hb -= nf;
            }
            // How many quads may work this step.  A node visit adds up to three entries, and the bag's room is what the ORDERED traversal
            // was given (STACK_LDS + spill_entries >= 3 x depth: what one quad needs from any starting point) plus the 48 entries this
            // layout adds in LDS: k quads need 3 k of those 48; with less left only quad 0 works -- a depth-first descent again, which
            // fits by the host's sizing.  A quad that is held back keeps its node.
            const int a_max = (48 - sp - hb) >= 6 ? (48 - sp - hb) / 3 : 1;
            const bool en = (int)((lane - gbase) >> 2) < a_max;
            // enabled quads without a node take one from the bag
            {
                const unsigned long long nm = __ballot(en && cur == NONE && qr == 0u) & gm;
                const int want = (int)__popcll(nm), take = want < sp ? want : sp;
                const int rk = (int)__popcll(nm & ((1ull << (lane & ~3u)) - 1ull));
                if (en && cur == NONE && rk < take) cur = bag[sp - 1 - rk];
                sp -= take;
            }
            // node part
            bool hit = false;
            uint32_t key = NONE, cref = 0u;
            const bool internal = en && (cur & 0x80000000u) == 0u;   // NONE has the leaf bit
            if (internal) {
                const float4 rec = ldq(S.nodes_q, (size_t)cur * NODE_QUADS + qr);
                if (qr == 0u) cn.add(C_NODE);
                if (COUNT) { cn.add(C_U_NODE_LANES); cn.add(C_U_TAIL_SHADOW); }
                const float ox = quad_permf<kQBcast0>(rec.w), oy = quad_permf<kQBcast1>(rec.w), oz = quad_permf<kQBcast2>(rec.w);
                const uint32_t e = quad_perm<kQBcast3>(__float_as_uint(rec.w));
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
                const float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, kEps));
                const float t1 = fminf(fminf(tfx, tfy), fminf(tfz, tmax));
                hit = t0 <= t1 * 1.0000004f;
                key = hit ? ((__float_as_uint(t0) & ~3u) | qr) : NONE;
                cref = __float_as_uint(rec.z);
            }
            if (COUNT && (int)lane == __ffsll((long long)__ballot(1)) - 1) { cn.add(C_U_NODE_SLOTS, 64); cn.add(C_U_TAIL_SLOTS, 64); }
            {
                uint32_t km = min(key, quad_perm<kQXor1>(key));
                km = min(km, quad_perm<kQRot2>(km));
                const bool cont = hit && key == km;
                uint32_t nx = cont ? cref : 0u;
                nx |= quad_perm<kQXor1>(nx);
                nx |= quad_perm<kQRot2>(nx);
                if (internal) cur = km != NONE ? nx : NONE;   // the nearest hit child stays with the quad
                const bool push = hit && !cont;
                const unsigned long long pm = __ballot(push) & gm;
                const int pos = sp + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                if (push) {
                    if (pos < 64) bag[pos] = cref;
                    else stack_push_slow(SPC_FAN_SPILL__, st.spill_entries, hb + pos - 64, cref, st.diag);
                }
                sp += (int)__popcll(pm);
                if (sp > 64) { hb += sp - 64; sp = 64; }
            }
            // leaf part: lane r tests triangle r of the quad's leaf (one that was taken from the bag, or the nearest child just found)
            bool occl = false;
            if (en && cur != NONE && (cur & 0x80000000u) != 0u) {
                const int tri = (int)((cur & 0x7fffffffu) >> 3) + (int)qr;
                if ((int)qr < (int)(cur & 7u)) {
                    const size_t base = (size_t)tri * 4;
                    const float4 a = ldq(S.tris, base), b = ldq(S.tris, base + 1), c = ldq(S.tris, base + 2);
                    cn.add(C_TRI);
                    if (COUNT) cn.add(C_U_TRI_LANES);
                    float t, u, v;
                    occl = tri_test(a, b, c, o, d, kEps, tmax, false, t, u, v);
                }
                cur = NONE;
            }
            if (COUNT && (int)lane == __ffsll((long long)__ballot(1)) - 1) cn.add(C_U_TRI_SLOTS, 64);
            const bool blocked = (__ballot(occl) & gm) != 0ull;
            const bool more = sp != 0 || hb != 0 || (__ballot(cur != NONE) & gm) != 0ull;
            if (blocked || !more) {
                live = false;
                if (lane == gbase && blocked) s_rayw[vis_slot].w = -1.0f;   // an occluded pair is no connection (kernels.hip: the connect phase's test)
            }
#undef SPC_FAN_SPILL__
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---- wave-cooperative traversal: one closest-hit ray per lane + a pool of shadow rays -----------------------------------
// Per iteration of the megakernel a wave has up to 64 closest-hit rays (the next path segments) and up to 192 shadow rays
// (CONNECTION_N per eye vertex of the previous segment).  Neither depends on the other, so both are traced in ONE pass:
// every lane first traces its own closest-hit ray, then pulls shadow rays from an LDS pool until it is empty -- lanes
// without a path (ended, or waiting for the next tile) pull from the start.  Ray lengths differ by two orders of magnitude
// (45 % of the shadow rays are occluded after a few nodes, others cross the whole scene): with one ray per lane per phase
// the wave waits for its longest ray (measured: 21 % VALU lane utilisation in any-hit traversal); pulling keeps lanes busy.
//   s_org[64]   origin of the shadow rays owned by lane l (its eye vertex)
//   s_ray[192]  shadow ray it * 64 + l: direction.xyz, length (< 0: no ray in this slot); out: the length of an OCCLUDED ray is
//               set to -1 -- after the pass the slots that still hold a ray are the unoccluded pairs
//   s_next      pool cursor, must be 0 on entry
// Wave-scope fences around the call order the LDS traffic; all 64 lanes must call this together.
static constexpr int POOL_RAYS = 64 * SPCBPT_CONNECTION_N;
// Compacts the slots of the wave's ray pool that hold a ray (length >= 0) into s_list; returns their number (wave-uniform).  All 64
// lanes call it after the rays of the iteration have been written (wave-scope fence before and after).
// LONGEST FIRST (SPC_POOL_BUCKETS > 1): a pass ends when its last ray ends, and the lanes that find the pool empty idle until then --
// 37 % of the pass's iterations ran after the pool was dry, at 47 % of the lanes.  The rays are drawn in list order, so the list is
// written in order of decreasing length class (bounds 2, 1, 1/2 of the wave's mean length: the work of an unoccluded shadow ray
// grows with the nodes its segment crosses): the long rays start first and the short ones fill the end of the pass, as in
// longest-processing-time-first scheduling.  Which lane traces which ray, and when, changes; every ray and its answer do not.
#ifndef SPC_POOL_BUCKETS
#define SPC_POOL_BUCKETS 4
#endif
#ifndef SPC_PROBE_HALVE_LONG
#define SPC_PROBE_HALVE_LONG 0   // TIMING PROBE (images invalid): the rays of the longest class end at half their length -- what splitting them in two could save at most
#endif
SPC_DEV uint32_t pool_ray_list(const float4* s_ray, uint8_t* s_list, float* s_mean = nullptr) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n = 0u;
    float len[SPCBPT_CONNECTION_N];
#pragma unroll
    for (int it = 0; it < SPCBPT_CONNECTION_N; it++) len[it] = s_ray[it * 64 + lane].w;
    if (SPC_POOL_BUCKETS <= 1) {
#pragma unroll
        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
            const bool has = len[it] >= 0.0f;
            const unsigned long long m = __ballot(has);
            if (has) s_list[n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)(it * 64 + lane);
            n += (uint32_t)__popcll(m);
        }
        return n;
    }
    // the wave's mean ray length
    float sum = 0.0f;
    uint32_t cnt = 0u;
#pragma unroll
    for (int it = 0; it < SPCBPT_CONNECTION_N; it++) { sum += fmaxf(len[it], 0.0f); cnt += (uint32_t)__popcll(__ballot(len[it] >= 0.0f)); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (cnt == 0u) return 0u;
    const float mean = sum / (float)cnt;
    if (SPC_PROBE_HALVE_LONG && s_mean && lane == 0) *s_mean = mean;
    // class 0 = longest.  SPC_POOL_BUCKETS = 2: [mean, inf), [0, mean); 4: [2 mean, inf), [mean, 2 mean), [mean / 2, mean), [0, mean / 2)
    int cls[SPCBPT_CONNECTION_N];
#pragma unroll
    for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
        if (SPC_POOL_BUCKETS == 2) cls[it] = len[it] >= mean ? 0 : 1;
        else cls[it] = len[it] >= 2.0f * mean ? 0 : (len[it] >= mean ? 1 : (len[it] >= 0.5f * mean ? 2 : 3));
        if (!(len[it] >= 0.0f)) cls[it] = -1;
    }
#pragma unroll
    for (int c = 0; c < (SPC_POOL_BUCKETS == 2 ? 2 : 4); c++) {
#pragma unroll
        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
            const bool has = cls[it] == c;
            const unsigned long long m = __ballot(has);
            if (has) s_list[n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)(it * 64 + lane);
            n += (uint32_t)__popcll(m);
        }
    }
    return n;
}
//   s_list[n_rays]  the slots that hold a ray, compacted by the caller (pool_ray_list): a lane never draws an empty slot -- each
//                   empty draw cost the whole wave an LDS atomic round trip with one lane active, and half the 192 slots are empty
template <bool COUNT, int BLOCK, int STACK_LDS>
SPC_DEV void trace_pool(const DeviceScene& S, TravStack<BLOCK, STACK_LDS>& st, bool own, f3 own_o, f3 own_d, HitRec& own_hit,
                        const float4* s_org, float4* s_ray, uint32_t* s_next, const uint8_t* s_list, uint32_t n_rays,
                        Counts<COUNT>& cn, const float4* s_hot = nullptr, const int n_hot = 0) {
    uint32_t r = 0;
    bool closest = own, done = false;
    unsigned long long quad_live = 0ull;   // != 0: the lanes whose rays the quad tail takes over
    f3 o = own_o, d = own_d;
    f3 inv = safe_inv(d), ood = o * inv;
    float best_t = 1e16f, best_u = 0.0f, best_v = 0.0f;
    int best_tri = -1;
    int node = own ? 0 : kTravDone, leaf_count = 0;  // kTravDone = this lane holds no ray
    st.sp = 0;
    if (own) cn.add(C_CLOSEST);
    own_hit.t = 1e16f; own_hit.tri = -1; own_hit.u = own_hit.v = 0.0f;
#if SPC_ONE_FETCH
    // ONE gather per iteration, issued one step AHEAD: a lane on an internal node needs its 64-B node record, a lane on a leaf its
    // 64-B triangle record -- the same four loads with another base.  The record of the NEXT step is requested as soon as the step
    // that decides it is done, so the fetch is in flight while the finished lanes store their results and draw new rays (an LDS
    // atomic and three dependent LDS reads) and the wave votes; an iteration waits for (what is left of) one round trip where the
    // if-if schedule of rounds 1-3 waited for two in a row, node record then triangle.  A leaf that a node step reaches is tested in
    // the next iteration: a lane advances one step per iteration.  Same steps, same order per ray: the films do not change.
    static_assert(NODE_QUADS == 4, "node and triangle records are both four quads");
    float4 R0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), R1 = R0, R2 = R0, R3 = R0;
    // A lane WITHOUT a ray that may still draw one (node == kTravDone, !done) requests the ROOT here, with everybody else's next
    // record: the ray it draws at the top of the next iteration then finds its first record under way, and the loop has ONE request
    // site.  (A root fetch at the draw was a second site writing R0..R3 from LDS while the other lanes' records were in flight to the
    // same registers from memory: the compiler has to wait for those first -- vmcnt(0) in nearly every iteration.)
#define SPC_FETCH_STEP__()                                                                                            \
    do {                                                                                                              \
        const int fn__ = (SPC_ROOT_AHEAD && node == kTravDone) ? 0 : node;                                            \
        if ((uint32_t)fn__ < (uint32_t)n_hot) {                                                                \
            /* one of the hottest nodes (lbvh.cpp numbers them first): the block's LDS copy, no trip through the vector L1 */ \
            const float4* h__ = s_hot + fn__ * 4;                                                                     \
            R0 = h__[0]; R1 = h__[1]; R2 = h__[2]; R3 = h__[3];                                                       \
        } else {                                                                                                      \
        /* one base: the triangle records follow the node records (DeviceScene::tri_base = n_nodes) */                \
        const size_t rb__ = (size_t)(uint32_t)(fn__ < 0 ? S.tri_base + ~fn__ : fn__) * 4;                             \
        R0 = ldq(S.nodes, rb__); R1 = ldq(S.nodes, rb__ + 1); R2 = ldq(S.nodes, rb__ + 2); R3 = ldq(S.nodes, rb__ + 3); \
        }                                                                                                             \
    } while (0)
    if (SPC_ROOT_AHEAD || node != kTravDone) SPC_FETCH_STEP__();
#endif
    while (true) {
        if (node == kTravDone && !done) {  // acquire the next shadow ray of the pool
            const uint32_t k = atomicAdd(s_next, 1u);
            if (k >= n_rays) { done = true; }
            else {
            r = s_list[k];
            const float4 rq = s_ray[r];
            const float4 oq = s_org[r & 63u];
            o = mk3(oq.x, oq.y, oq.z); d = mk3(rq.x, rq.y, rq.z);
            inv = safe_inv(d); ood = o * inv;
            best_t = rq.w - kEps;
            if (SPC_PROBE_HALVE_LONG && rq.w >= 2.0f * __uint_as_float(s_next[1])) best_t = 0.5f * rq.w;
            node = 0; st.sp = 0;
            cn.add(C_SHADOW);
#if SPC_ONE_FETCH
            if (!SPC_ROOT_AHEAD) SPC_FETCH_STEP__();   // the root
#endif
            }
        }
        const unsigned long long live__ = __ballot(node != kTravDone);
        if (live__ == 0ull) break;
        // The pool is dry (some lane found it empty; the cursor only grows) and at most 16 rays are still in flight: the rest of the
        // pass is the tail that used to run these iterations at a fifth of the lanes.  Hand each ray to FOUR lanes (quad tail below).
#if SPC_PROBE_DROP_TAIL
        // TIMING PROBES (images invalid; profiles/r06_experiments.md): what the end of the pass costs -- the upper bound of anything that
        // would carry its unfinished shadow rays into the next pass.  1: the rays the quad tail would take over are dropped (left
        // unoccluded) when no closest-hit ray is among them; 2: every shadow ray still in flight once the pool is dry and the
        // closest-hit rays are done.
        if (__any(done) && !__any(node != kTravDone && closest) && (SPC_PROBE_DROP_TAIL == 2 || __popcll(live__) <= 16)) break;
#endif
        if (SPC_QUAD_TAIL && S.nodes_q && __popcll(live__) <= 16 && __any(done)) { quad_live = live__; break; }
        const bool tail = COUNT && __any(done);   // (counting build) some lane found the pool empty: what follows is the pass's tail
        bool finished = false, occluded = false;
        const bool shallow__ = !__any(node != kTravDone && st.sp + 3 > STACK_LDS);   // no lane near the end of its LDS entries (wave-uniform)
        bool hold__ = false;
        if (SPC_TRI_BATCH > 1) {
            const unsigned long long leaf_m = __ballot(node < 0), inner_m = __ballot(node >= 0 && node != kTravDone);
            hold__ = node < 0 && __popcll(leaf_m) < SPC_TRI_BATCH && inner_m != 0ull;
        }
        if (node != kTravDone) {
            if (COUNT && tail && node >= 0) {
                cn.add(closest ? C_U_TAIL_CLOSEST : C_U_TAIL_SHADOW);
                if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) cn.add(C_U_TAIL_SLOTS, 64);
            }
            // all sixteen dwords of the record fetched a step ago, here: without this the compiler narrows the four 16-B loads to what both
            // kinds of step read and fetches the rest inside the branches, after the wait -- a second round trip
            asm volatile("" : "+v"(R0.x), "+v"(R0.y), "+v"(R0.z), "+v"(R0.w), "+v"(R1.x), "+v"(R1.y), "+v"(R1.z), "+v"(R1.w));
            asm volatile("" : "+v"(R2.x), "+v"(R2.y), "+v"(R2.z), "+v"(R2.w), "+v"(R3.x), "+v"(R3.y), "+v"(R3.z), "+v"(R3.w));
            // The step, in two instantiations: with the plain stack operations (bounds logic and call sites of the HBM part at every push and
            // pop: ~30 instructions that a step pays whether or not any lane is near the end of its 16 LDS entries), and with the LDS-only
            // ones for the iterations in which no lane of the wave is (a vote per iteration; probe: 1.8 % of the kernel).
#define SPC_POOL_STEP__(PUSH, POP)                                                                                    \
            do {                                                                                                      \
                const bool at_leaf = node < 0;                                                                        \
                if (!at_leaf) { SPC_NODE_STEP_Q(kEps, best_t, R0, R1, R2, R3, PUSH, POP); finished = node == kTravDone; } \
                else if (hold__) { /* SPC_TRI_BATCH: the triangle step waits for company (its record stays in R0..R3) */ } \
                else if (leaf_count <= 0) {                                                                           \
                    SPC_TRAV_POP_(POP);  /* an empty slot's zero-triangle leaf (only reachable through rounding): nothing to test */ \
                    finished = node == kTravDone;                                                                     \
                } else {                                                                                              \
                    const int tri = ~node;                                                                            \
                    cn.add(C_TRI);                                                                                    \
                    SPC_UTIL_COUNT(C_U_TRI_LANES, C_U_TRI_SLOTS)                                                      \
                    /* the slot of a FAN PAIR (lbvh.h: R0 .. R2 = A's corners, R3.xyz = B's third corner, B = (A.P0, A.P2, R3)): this step */ \
                    /* tests both, A first and B against what A left of the interval -- the operations and the order of two steps */ \
                    const uint32_t fl__ = __float_as_uint(R3.w);                                                      \
                    const bool pair__ = SPC_TRI_PAIRS && (fl__ & 1u) != 0;                                            \
                    const bool cull = closest && (fl__ & 0x80000000u) != 0;  /* single-sided emitters */              \
                    float t, u, v;                                                                                    \
                    bool h = tri_test<true>(R0, R1, R2, o, d, kEps, best_t, cull, t, u, v);                                 \
                    if (h && closest) { best_t = t; best_tri = tri; best_u = u; best_v = v; }                         \
                    if (pair__ && !(h && !closest)) {                                                                 \
                        cn.add(C_TRI);                                                                                \
                        if (COUNT) cn.add(C_U_TRI_LANES);                                                             \
                        const bool hb__ = tri_test<true>(R0, R2, R3, o, d, kEps, best_t, closest && (fl__ & 0x40000000u) != 0, t, u, v); \
                        if (hb__ && closest) { best_t = t; best_tri = tri + 1; best_u = u; best_v = v; }              \
                        h = h || hb__;                                                                                \
                    }                                                                                                 \
                    if (h && !closest) {                                                                              \
                        occluded = true; finished = true; node = kTravDone;                                           \
                    } else {                                                                                          \
                        /* SPC_PROBE_TRI_PAIRS (timing probe, images invalid): a step answers for two triangles of the leaf */ \
                        const int adv__ = (pair__ || (SPC_PROBE_TRI_PAIRS && leaf_count >= 2)) ? 2 : 1;               \
                        node -= adv__;  /* ~(tri + 1) */                                                              \
                        leaf_count -= adv__;                                                                          \
                        if (leaf_count == 0) { SPC_TRAV_POP_(POP); finished = node == kTravDone; }                    \
                    }                                                                                                 \
                }                                                                                                     \
            } while (0)
            if (shallow__) SPC_POOL_STEP__(push_far_lds, pop_lds);
            else SPC_POOL_STEP__(push_far, pop);
#undef SPC_POOL_STEP__
        }
#if SPC_ONE_FETCH
        if ((node != kTravDone || (SPC_ROOT_AHEAD && !done)) && !hold__) SPC_FETCH_STEP__();   // the next step's record: ONE request site after the step, outside its branches
#endif
        if (finished) {
            if (closest) {
                own_hit.t = best_t; own_hit.tri = best_tri; own_hit.u = best_u; own_hit.v = best_v;
                closest = false;
                best_tri = -1;
            } else {
                if (occluded) s_ray[r].w = -1.0f;   // the answer of a shadow ray: an occluded pair's slot holds no ray any more (the connect phase's test)
            }
        }
    }
#undef SPC_FETCH_STEP__
    if (SPC_QUAD_TAIL && quad_live != 0ull) {
#if SPC_PRIO_TAIL >= 0
        __builtin_amdgcn_s_setprio(SPC_PRIO_TAIL);   // (experiment: the tails at another priority than the lane loop; the caller resets it after the pass)
#endif
        // ---- quad tail: the k-th ray still in flight continues on lanes 4 k .. 4 k + 3 -------------------------------------------
        // Lane r of a quad loads record r of the node (one coalesced 64-B line per ray), tests ITS child, and the four entry
        // distances are ranked across the quad: the same keys, the same order, the same pushes as SPC_NODE_STEP -- onto the SAME
        // stack, the owner lane's LDS column (and its HBM part) -- so the ray visits what it would have visited.  At a leaf lane r
        // tests triangle r.  An iteration is ~100 instructions instead of ~275 and serves up to 16 rays, which is all there are.
        const uint32_t lane = lane_id_fresh(), qr = lane & 3u;
        uint8_t* list = const_cast<uint8_t*>(s_list);   // the ray list of the pass is used up: the owners' lane ids go there
        if (node != kTravDone) list[__popcll(quad_live & ((1ull << lane) - 1ull))] = (uint8_t)lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n_live = (int)__popcll(quad_live);
        const bool has = (int)(lane >> 2) < n_live;
        const int owner = has ? (int)list[lane >> 2] : (int)lane;
        // the ray's state, from its owner
        const f3 qo = mk3(__shfl(o.x, owner, 64), __shfl(o.y, owner, 64), __shfl(o.z, owner, 64));
        const f3 qd = mk3(__shfl(d.x, owner, 64), __shfl(d.y, owner, 64), __shfl(d.z, owner, 64));
        float q_best = __shfl(best_t, owner, 64);
        int q_node = __shfl(node, owner, 64), q_leaf = __shfl(leaf_count, owner, 64), q_sp = __shfl(st.sp, owner, 64);
        const bool q_closest = __shfl((int)closest, owner, 64) != 0;
        const uint32_t q_r = (uint32_t)__shfl((int)r, owner, 64);
        // the owner's best hit so far continues as lane 0's (a closest-hit ray may have found one before the hand-over)
        float my_t = 1e30f, my_u = 0.0f, my_v = 0.0f;
        int my_tri = -1;
        {
            const int bt = __shfl(best_tri, owner, 64);
            const float bu = __shfl(best_u, owner, 64), bv = __shfl(best_v, owner, 64);
            if (qr == 0u && bt >= 0) { my_t = q_best; my_tri = bt; my_u = bu; my_v = bv; }
        }
        if (!has) q_node = kTravDone;
        const f3 qinv = safe_inv(qd), qood = qo * qinv;
        lds_u32* const col = st.wave_lds + owner;
        uint32_t* const q_spill = st.spill ? st.spill + ((long long)owner - (long long)lane) * (long long)st.spill_entries : nullptr;
        bool q_occluded = false, q_done = !has;
        while (__any(q_node != kTravDone)) {
            // no closest-hit ray left: the shadow rays that remain need no order any more (fan_tail below)
            if (SPC_FAN_TAIL && S.fan_tail && !__any(q_node != kTravDone && q_closest)) break;
            if (q_node != kTravDone) {
                bool finished = false;
                if (q_node >= 0) {
                    const float4 rec = ldq(S.nodes_q, (size_t)q_node * NODE_QUADS + qr);
                    if (qr == 0u) cn.add(C_NODE);
                    if (COUNT) { cn.add(C_U_NODE_LANES); cn.add(q_closest ? C_U_TAIL_CLOSEST : C_U_TAIL_SHADOW); if ((int)lane == __ffsll((long long)__ballot(1)) - 1) { cn.add(C_U_NODE_SLOTS, 64); cn.add(C_U_TAIL_SLOTS, 64); } }
                    const float ox = quad_permf<kQBcast0>(rec.w), oy = quad_permf<kQBcast1>(rec.w), oz = quad_permf<kQBcast2>(rec.w);
                    const uint32_t e = quad_perm<kQBcast3>(__float_as_uint(rec.w));
                    const float ax = __uint_as_float((e & 0xffu) << 23) * qinv.x, ay = __uint_as_float(((e >> 8) & 0xffu) << 23) * qinv.y,
                                az = __uint_as_float(((e >> 16) & 0xffu) << 23) * qinv.z;
                    const float bx = fmaf(ox, qinv.x, -qood.x), by = fmaf(oy, qinv.y, -qood.y), bz = fmaf(oz, qinv.z, -qood.z);
                    const uint32_t pa = __float_as_uint(rec.x), pb = __float_as_uint(rec.y);
                    const float lx = (float)(pa & 255u), ly = (float)((pa >> 8) & 255u), lz = (float)((pa >> 16) & 255u), hx = (float)(pa >> 24),
                                hy = (float)(pb & 255u), hz = (float)((pb >> 8) & 255u);
                    const bool sx = qinv.x < 0.0f, sy = qinv.y < 0.0f, sz = qinv.z < 0.0f;
                    const float tnx = fmaf(sx ? hx : lx, ax, bx), tfx = fmaf(sx ? lx : hx, ax, bx);
                    const float tny = fmaf(sy ? hy : ly, ay, by), tfy = fmaf(sy ? ly : hy, ay, by);
                    const float tnz = fmaf(sz ? hz : lz, az, bz), tfz = fmaf(sz ? lz : hz, az, bz);
                    const float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, kEps));
                    const float t1 = fminf(fminf(tfx, tfy), fminf(tfz, q_best));
                    const uint32_t key = (t0 <= t1 * 1.0000004f) ? ((__float_as_uint(t0) & ~3u) | qr) : 0xffffffffu;
                    const uint32_t ref = __float_as_uint(rec.z);
                    const uint32_t k1 = quad_perm<kQRot1>(key), k2 = quad_perm<kQRot2>(key), k3 = quad_perm<kQRot3>(key);
                    const bool hit = key != 0xffffffffu;
                    const int rank = (k1 < key ? 1 : 0) + (k2 < key ? 1 : 0) + (k3 < key ? 1 : 0);
                    const int nh = (hit ? 1 : 0) + (k1 != 0xffffffffu ? 1 : 0) + (k2 != 0xffffffffu ? 1 : 0) + (k3 != 0xffffffffu ? 1 : 0);
                    uint32_t next;
                    if (nh == 0) {
                        if (q_sp == 0) next = 0xffffffffu;
                        else { q_sp--; next = q_sp < STACK_LDS ? col[q_sp * BLOCK] : stack_pop_slow(q_spill, st.spill_entries, q_sp - STACK_LDS); }
                    } else {
                        if (hit && rank > 0) {   // farthest deepest, second nearest on top: push_far's order
                            const int e2 = q_sp + nh - 1 - rank;
                            if (e2 < STACK_LDS) col[e2 * BLOCK] = ref;
                            else stack_push_slow(q_spill, st.spill_entries, e2 - STACK_LDS, ref, st.diag);
                        }
                        q_sp += nh - 1;
                        const uint32_t mine = (hit && rank == 0) ? ref : 0u;
                        next = mine | quad_perm<kQRot1>(mine) | quad_perm<kQRot2>(mine) | quad_perm<kQRot3>(mine);
                    }
                    if (next == 0xffffffffu) { q_node = kTravDone; finished = true; }
                    else if (next & 0x80000000u) { q_node = ~(int)((next & 0x7fffffffu) >> 3); q_leaf = (int)(next & 7u); }
                    else q_node = (int)next;
                }
                if (!finished && q_node < 0) {
                    float t = 1e30f, u = 0.0f, v = 0.0f;
                    bool h = false;
                    const int tri = ~q_node + (int)qr;
                    if ((int)qr < q_leaf) {
                        const size_t base = (size_t)tri * 4;
                        const float4 a = ldq(S.tris, base), b = ldq(S.tris, base + 1), c = ldq(S.tris, base + 2);
                        cn.add(C_TRI);
                        bool cull = false;
                        if (q_closest) cull = (__float_as_uint(ldq(S.tris, base + 3).w) & 0x80000000u) != 0;
                        h = tri_test(a, b, c, qo, qd, kEps, q_best, cull, t, u, v);
                        if (!h) t = 1e30f;
                    }
                    if (COUNT) { if ((int)qr < q_leaf) cn.add(C_U_TRI_LANES); if ((int)lane == __ffsll((long long)__ballot(1)) - 1) cn.add(C_U_TRI_SLOTS, 64); }
                    if (h && t < my_t) { my_t = t; my_tri = tri; my_u = u; my_v = v; }
                    float tq = fminf(t, quad_permf<kQXor1>(t));
                    tq = fminf(tq, quad_permf<kQRot2>(tq));
                    if (!q_closest) {
                        if (tq < 1e30f) { q_occluded = true; q_node = kTravDone; finished = true; }
                    } else q_best = fminf(q_best, tq);
                    if (!finished) {
                        if (q_sp == 0) { q_node = kTravDone; finished = true; }
                        else {
                            q_sp--;
                            const uint32_t w = q_sp < STACK_LDS ? col[q_sp * BLOCK] : stack_pop_slow(q_spill, st.spill_entries, q_sp - STACK_LDS);
                            if (w & 0x80000000u) { q_node = ~(int)((w & 0x7fffffffu) >> 3); q_leaf = (int)(w & 7u); }
                            else q_node = (int)w;
                        }
                    }
                }
                if (finished) {
                    q_done = true;
                    if (!q_closest && qr == 0u && q_occluded) s_ray[q_r].w = -1.0f;
                }
            }
        }
        // ---- closest-hit rays of the tail: the winner's record goes back to the owner lane through the owner's stack column, which is
        // empty again (a finished ray has popped everything).  The lane that holds the quad's nearest hit is the lowest lane with
        // my_t == q_best (= the lower triangle index on a tie; lane 0 carries a hit found before the hand-over, which a later equal
        // distance does not replace: strict <, as in the lane loop).
        {
            const uint32_t cand = (has && q_closest && my_tri >= 0 && my_t == q_best) ? qr : 4u;
            uint32_t wq = min(cand, quad_perm<kQXor1>(cand));
            wq = min(wq, quad_perm<kQRot2>(wq));
            if (has && q_closest && (wq == 4u ? qr == 0u : wq == qr)) {
                col[0 * BLOCK] = __float_as_uint(wq == 4u ? q_best : my_t);
                col[1 * BLOCK] = (uint32_t)(wq == 4u ? -1 : my_tri);
                col[2 * BLOCK] = __float_as_uint(wq == 4u ? 0.0f : my_u);
                col[3 * BLOCK] = __float_as_uint(wq == 4u ? 0.0f : my_v);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (((quad_live >> lane) & 1ull) != 0ull && closest) {
                const lds_u32* mine = st.wave_lds + lane;
                own_hit.t = __uint_as_float(mine[0 * BLOCK]); own_hit.tri = (int)mine[1 * BLOCK];
                own_hit.u = __uint_as_float(mine[2 * BLOCK]); own_hit.v = __uint_as_float(mine[3 * BLOCK]);
            }
        }
        (void)q_done;
        if (SPC_FAN_TAIL && S.fan_tail && __any(q_node != kTravDone))
            fan_tail<COUNT, BLOCK, STACK_LDS>(S, st, q_node != kTravDone, qo, qd, q_best, q_node == kTravDone ? 0xffffffffu : stack_word(q_node, q_leaf > 0 ? q_leaf : 0), q_sp, owner,
                                              q_r, s_ray, list, cn);
    }
}

}  // namespace spc

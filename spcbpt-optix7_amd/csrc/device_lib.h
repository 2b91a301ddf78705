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
SPC_DEV bool tri_test(float4 q0, float4 q1, float4 q2, f3 o, f3 d, float tmin, float tmax, bool cull, float& ot, float& ou, float& ov) {
    const f3 v0 = mk3(q0.x, q0.y, q0.z);
    const f3 e1 = mk3(q1.x, q1.y, q1.z) - v0, e2 = mk3(q2.x, q2.y, q2.z) - v0;
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
            const size_t base = (size_t)tri * 4;                                                                      \
            const float4 a = ldq(S.tris, base), b = ldq(S.tris, base + 1), c = ldq(S.tris, base + 2);                 \
            cn.add(C_TRI);                                                                                            \
            SPC_UTIL_COUNT(C_U_TRI_LANES, C_U_TRI_SLOTS)                                                              \
            bool cull = false;                                                                                        \
            if (!ANY) {                                                                                               \
                /* emitter flag lives in quad 3; only fetched for closest-hit rays (single-sided emitters, q16) */     \
                cull = (__float_as_uint(ldq(S.tris, base + 3).w) & 0x80000000u) != 0;                                 \
            }                                                                                                         \
            float t, u, v;                                                                                            \
            const bool h = tri_test(a, b, c, o, d, tmin, best_t, cull, t, u, v);                                      \
            if (h) { best_t = t; best_tri = tri; best_u = u; best_v = v; }                                            \
            if (ANY && h) {                                                                                           \
                node = kTravDone;                                                                                     \
            } else {                                                                                                  \
                node -= 1;  /* ~(tri + 1) */                                                                          \
                leaf_count -= 1;                                                                                      \
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
SPC_DEV uint32_t pool_ray_list(const float4* s_ray, uint8_t* s_list) {
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
                    const bool cull = closest && (__float_as_uint(R3.w) & 0x80000000u) != 0;  /* single-sided emitters */ \
                    float t, u, v;                                                                                    \
                    const bool h = tri_test(R0, R1, R2, o, d, kEps, best_t, cull, t, u, v);                           \
                    if (h && !closest) {                                                                              \
                        occluded = true; finished = true; node = kTravDone;                                           \
                    } else {                                                                                          \
                        if (h) { best_t = t; best_tri = tri; best_u = u; best_v = v; }                                \
                        /* SPC_PROBE_TRI_PAIRS (timing probe, images invalid): a step answers for two triangles of the leaf */ \
                        const int adv__ = (SPC_PROBE_TRI_PAIRS && leaf_count >= 2) ? 2 : 1;                           \
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

// ---- materials / textures / hit geometry -------------------------------------
struct Pbr {
    f3 base;
    float metallic, roughness, specular, specularTint, subsurface, sheen, sheenTint, clearcoat, clearcoatGloss;
    int albedo_tex, light_id;
    int brdf;   // MaterialData::Pbr::brdf: see brdf_div
};
SPC_DEV Pbr load_pbr(const DeviceScene& S, int id) {
    const float4* p = reinterpret_cast<const float4*>(S.mats + id);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    Pbr m;
    m.base = mk3(a.x, a.y, a.z); m.metallic = a.w;
    m.roughness = b.x; m.specular = b.y; m.specularTint = b.z; m.subsurface = b.w;
    m.sheen = c.x; m.sheenTint = c.y; m.clearcoat = c.z; m.clearcoatGloss = c.w;
    m.albedo_tex = __float_as_int(d.x); m.light_id = __float_as_int(d.y); m.brdf = __float_as_int(d.z);
    return m;
}
SPC_DEV Pbr load_pbr_colored(const DeviceScene& S, int id, f3 color) {  // rmis::getMat (rmis.h:16-21)
    Pbr m = load_pbr(S, id);
    m.base = color;
    return m;
}
// bilinear + wrap RGBA8 fetch (cudaReadModeNormalizedFloat semantics; exact-fraction weights)
SPC_DEV f3 tex_fetch_rgb(const DTexture& T, float u, float v) {
    const float x = u * (float)T.width - 0.5f, y = v * (float)T.height - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    int x0 = (int)fx % T.width, y0 = (int)fy % T.height;
    if (x0 < 0) x0 += T.width;
    if (y0 < 0) y0 += T.height;
    int x1 = x0 + 1 == T.width ? 0 : x0 + 1, y1 = y0 + 1 == T.height ? 0 : y0 + 1;
    const uint32_t t00 = T.rgba[(size_t)y0 * T.width + x0], t10 = T.rgba[(size_t)y0 * T.width + x1];
    const uint32_t t01 = T.rgba[(size_t)y1 * T.width + x0], t11 = T.rgba[(size_t)y1 * T.width + x1];
    const float w00 = (1 - ax) * (1 - ay), w10 = ax * (1 - ay), w01 = (1 - ax) * ay, w11 = ax * ay;
    const float s = 1.0f / 255.0f;
    f3 r;
    r.x = w00 * ((t00 & 255u) * s) + w10 * ((t10 & 255u) * s) + w01 * ((t01 & 255u) * s) + w11 * ((t11 & 255u) * s);
    r.y = w00 * (((t00 >> 8) & 255u) * s) + w10 * (((t10 >> 8) & 255u) * s) + w01 * (((t01 >> 8) & 255u) * s) + w11 * (((t11 >> 8) & 255u) * s);
    r.z = w00 * (((t00 >> 16) & 255u) * s) + w10 * (((t10 >> 16) & 255u) * s) + w01 * (((t01 >> 16) & 255u) * s) + w11 * (((t11 >> 16) & 255u) * s);
    return r;
}
struct Geom { f3 P, N; float u, v; int mat; bool emitter; };
SPC_DEV Geom local_geometry(const DeviceScene& S, const HitRec& h) {
    const size_t base = (size_t)h.tri * 4;
    const float4 a = ldq(S.tris, base), b = ldq(S.tris, base + 1), c = ldq(S.tris, base + 2), d = ldq(S.tris, base + 3);
    const f3 P0 = mk3(a.x, a.y, a.z), P1 = mk3(b.x, b.y, b.z), P2 = mk3(c.x, c.y, c.z);
    Geom g;
    const float w = 1.0f - h.u - h.v;
    g.P = w * P0 + h.u * P1 + h.v * P2;
    g.N = normalize(cross(P1 - P0, P2 - P0));
    g.u = w * a.w + h.u * c.w + h.v * d.y;
    g.v = w * b.w + h.u * d.x + h.v * d.z;
    const uint32_t meta = __float_as_uint(d.w);
    g.mat = (int)(meta & 0x7fffffffu);
    g.emitter = (meta & 0x80000000u) != 0;
    return g;
}
template <bool COUNT>
SPC_DEV void color_tex_sample(const DeviceScene& S, const Geom& g, Pbr& m, Counts<COUNT>& cn) {  // hit_program.cu:182-198
    if (m.albedo_tex > 0) {
        const f3 t = tex_fetch_rgb(S.tex[m.albedo_tex - 1], g.u, g.v);
        m.base = mk3(powf(t.x, 2.2f), powf(t.y, 2.2f), powf(t.z, 2.2f));  // linearize cuProg.h:361-368
        cn.add(C_TEX);
    }
}

// ---- Disney BSDF (cuProg.h:686-899) ---------------------------------------------
struct Onb {
    f3 t, b, n;
    SPC_DEV explicit Onb(f3 normal) {
        n = normal;
        if (fabsf(n.x) > fabsf(n.z)) b = mk3(-n.y, n.x, 0.0f);
        else b = mk3(0.0f, -n.z, n.y);
        b = normalize(b);
        t = cross(b, n);
    }
    SPC_DEV f3 to_world(f3 p) const { return p.x * t + p.y * b + p.z * n; }
};
SPC_DEV f3 cosine_sample_hemisphere(float u1, float u2) {
    const float r = sqrtf(u1);
    const float phi = 2.0f * kPi * u2;
    float s, c;
    sincosf(phi, &s, &c);
    f3 p;
    p.x = r * c; p.y = r * s;
    p.z = sqrtf(fmaxf(0.0f, 1.0f - p.x * p.x - p.y * p.y));
    return p;
}
SPC_DEV float schlick(float u) { float m = clampf(1.0f - u, 0.0f, 1.0f); float m2 = m * m; return m2 * m2 * m; }
SPC_DEV float gtr1(float NdH, float a) {
    if (a >= 1.0f) return kInvPi;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NdH * NdH;
    return (a2 - 1.0f) / (kPi * logf(a2) * t);
}
SPC_DEV float gtr2(float NdH, float a) {
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NdH * NdH;
    return a2 / (kPi * t * t);
}
SPC_DEV float smith_ggx(float NdV, float alphaG) {
    float a = alphaG * alphaG, b = NdV * NdV;
    return 1.0f / (NdV + sqrtf(a + b - a * b));
}
SPC_DEV f3 bsdf_eval(const Pbr& m, f3 N, f3 V, f3 L) {
    const float NdL = dot(N, L), NdV = dot(N, V);
    if (NdL <= 0.0f || NdV <= 0.0f) return mk3(0.0f);
    const f3 H = normalize(L + V);
    const float NdH = dot(N, H), LdH = dot(L, H);
    const f3 Cd = m.base;
    const float lum = 0.3f * Cd.x + 0.6f * Cd.y + 0.1f * Cd.z;
    const f3 Ctint = lum > 0.0f ? Cd / lum : mk3(1.0f);
    const f3 Cspec0 = lerp3(m.specular * 0.08f * lerp3(mk3(1.0f), Ctint, m.specularTint), Cd, m.metallic);
    const float FL = schlick(NdL), FV = schlick(NdV);
    const float Fd90 = 0.5f + 2.0f * LdH * LdH * m.roughness;
    const float Fd = lerpf(1.0f, Fd90, FL) * lerpf(1.0f, Fd90, FV);
    const float Fss90 = LdH * LdH * m.roughness;
    const float Fss = lerpf(1.0f, Fss90, FL) * lerpf(1.0f, Fss90, FV);
    const float ss = 1.25f * (Fss * (1.0f / (NdL + NdV) - 0.5f) + 0.5f);
    const float a = fmaxf(0.001f, m.roughness);
    const float Ds = gtr2(NdH, a);
    const float FH = schlick(LdH);
    const f3 Fs = lerp3(Cspec0, mk3(1.0f), FH);
    const float rg = (m.roughness * 0.5f + 0.5f) * (m.roughness * 0.5f + 0.5f);
    const float Gs = smith_ggx(NdL, rg) * smith_ggx(NdV, rg);
    f3 out = ((kInvPi * lerpf(Fd, ss, m.subsurface)) * Cd) * (1.0f - m.metallic) + Gs * Fs * Ds;
    if (m.sheen != 0.0f) {  // sheen term is exactly zero for sheen == 0
        const f3 Csheen = lerp3(mk3(1.0f), Ctint, m.sheenTint);
        out = ((kInvPi * lerpf(Fd, ss, m.subsurface)) * Cd + FH * m.sheen * Csheen) * (1.0f - m.metallic) + Gs * Fs * Ds;
    }
    if (m.clearcoat != 0.0f) {  // clearcoat term is exactly zero for clearcoat == 0
        const float Dr = gtr1(NdH, lerpf(0.1f, 0.001f, m.clearcoatGloss));
        const float Fr = lerpf(0.04f, 1.0f, FH);
        const float Gr = smith_ggx(NdL, 0.25f) * smith_ggx(NdV, 0.25f);
        out = out + mk3(0.25f * m.clearcoat * Gr * Fr * Dr);
    }
    return out;
}
// `Eval(...) / (mat.brdf ? abs(dot(n, dir)) : 1.0f)`: the un-guarded ternary of the bidirectional programs (hit_program.cu:286, 384;
// raygen.cu:271, 278; rmis.h:105) for a material with `brdf <nonzero>` in its .scene block.  operator/(float3, float) multiplies by
// the reciprocal (sutil/vec_math.h:483-487) and x * (1.0f / 1.0f) is x, so the division is only executed on the flagged branch.
// A grazing direction (|n.dir| == 0) gives inf / NaN as upstream: the vertex's later contributions fail ISINVALIDVALUE there and here.
// ENV = false (the timed kernels of a scene with neither an environment map nor a flagged material, DeviceScene::general == 0)
// compiles the test away: 0.6 % of the bedroom frame (A/B on one box, profiles/r04_experiments.md).
template <bool ENV = true>
SPC_DEV f3 brdf_div(const Pbr& m, f3 f, f3 n, f3 dir) {
    if (ENV && m.brdf) f = f / fabsf(dot(n, dir));
    return f;
}
SPC_DEV f3 bsdf_sample(const Pbr& m, f3 N, f3 V, uint32_t& seed) {
    const float probability = rnd(seed);
    const float diffuseRatio = 0.5f * (1.0f - m.metallic);
    const float r1 = rnd(seed), r2 = rnd(seed);
    const Onb onb(N);
    if (probability < diffuseRatio) return onb.to_world(cosine_sample_hemisphere(r1, r2));
    const float a = fmaxf(0.001f, m.roughness);
    const float phi = r1 * 2.0f * kPi;
    const float cosTheta = sqrtf((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float sinPhi, cosPhi;
    sincosf(phi, &sinPhi, &cosPhi);
    const f3 half = onb.to_world(mk3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
    return 2.0f * dot(V, half) * half - V;
}
SPC_DEV float bsdf_pdf(const Pbr& m, f3 n, f3 V, f3 L) {
    const float specularAlpha = fmaxf(0.001f, m.roughness);
    const float diffuseRatio = 0.5f * (1.0f - m.metallic);
    const float specularRatio = 1.0f - diffuseRatio;
    const f3 half = normalize(L + V);
    const float cosTheta = fabsf(dot(half, n));
    const float pdfGTR2 = gtr2(cosTheta, specularAlpha) * cosTheta;
    // kept as written in the reference even for clearcoat == 0: lerp(g1, g2, 1) = g1 + (g2 - g1) is NOT g2 in fp32
    const float pdfGTR1 = gtr1(cosTheta, lerpf(0.1f, 0.001f, m.clearcoatGloss)) * cosTheta;
    const float mix = lerpf(pdfGTR1, pdfGTR2, 1.0f / (1.0f + m.clearcoat));
    const float pdfSpec = mix / (4.0f * fabsf(dot(L, half)));
    const float pdfDiff = fabsf(dot(L, n)) * kInvPi;
    return diffuseRatio * pdfDiff + specularRatio * pdfSpec;
}

// ---- subspace classification ---------------------------------------------------
template <bool COUNT>
SPC_DEV int tree_label(const float* tree, f3 position, f3 normal, f3 dir, Counts<COUNT>& cn) {
    if (!tree) return 0;
    int node = 0;
    while (true) {
        const float4 q = ldq(tree, (size_t)node);  // the whole 16-B node in one round trip (layout.h)
        cn.add(C_TREE);
        const uint32_t meta = __float_as_uint(q.w);
        if (meta & TREE_LEAF_BIT) return (int)(meta & ~TREE_LEAF_BIT);
        const uint32_t type = (meta >> 29) & 3u;
        const f3 p = type == 0 ? position : (type == 1 ? normal : dir);
        node = (int)(meta & 0x1fffffffu) + (p.x > q.x ? 1 : 0) + (p.y > q.y ? 2 : 0) + (p.z > q.z ? 4 : 0);
    }
}

// Two independent classifications descended in lock-step: a descent is a chain of dependent fetches (one per level, up to 15
// levels), and the callers below always need two of them (eye-tree label of a new vertex + light-tree label for its RMIS
// recursion; the two relabels of a connection).  Interleaving halves the exposed latency; the labels are the same.
// The step is written without branches (selects on `go`): as nested ifs it compiled to ~45 scalar instructions of EXEC bookkeeping per
// iteration next to its ~45 vector ones, and a vertex's pair of descents -- the wave goes round until its deepest lane is done, ~13
// times -- was 4.7 % of the megakernel (profiles/r05_experiments.md, sections 27-28).  NODIR: the caller's trees hold no direction
// nodes (the label-caching kernels: Context::tree_has_direction sends every other tree to the generic instantiations), so the
// split point is compared with the position or the normal only.
template <bool COUNT, bool NODIR = false>
SPC_DEV void tree_label2(const float* treeA, f3 posA, f3 nA, f3 dirA, bool needA, const float* treeB, f3 posB, f3 nB, f3 dirB, bool needB,
                         int& labelA, int& labelB, Counts<COUNT>& cn) {
    int nodeA = 0, nodeB = 0;
    bool goA = needA && treeA != nullptr, goB = needB && treeB != nullptr;
    labelA = 0; labelB = 0;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a;   // (a lane that is done keeps its last record: the selects below ignore it)
    while (goA || goB) {
        if (goA) { a = ldq(treeA, (size_t)nodeA); cn.add(C_TREE); }
        if (goB) { b = ldq(treeB, (size_t)nodeB); cn.add(C_TREE); }
#define SPC_TREE_STEP__(q, pos, nrm, dir, node, label, go)                                                              \
        {                                                                                                             \
            const uint32_t meta = __float_as_uint(q.w);                                                               \
            const bool leaf = (meta & TREE_LEAF_BIT) != 0u;                                                           \
            const uint32_t type = (meta >> 29) & 3u;                                                                  \
            const f3 p = NODIR ? mk3(type != 0u ? nrm.x : pos.x, type != 0u ? nrm.y : pos.y, type != 0u ? nrm.z : pos.z)  \
                               : (type == 0u ? pos : (type == 1u ? nrm : dir));                                       \
            const int next = (int)(meta & 0x1fffffffu) + (p.x > q.x ? 1 : 0) + (p.y > q.y ? 2 : 0) + (p.z > q.z ? 4 : 0);   \
            label = (go && leaf) ? (int)(meta & ~TREE_LEAF_BIT) : label;                                              \
            node = (go && !leaf) ? next : node;                                                                       \
            go = go && !leaf;                                                                                         \
        }
        SPC_TREE_STEP__(a, posA, nA, dirA, nodeA, labelA, goA)
        SPC_TREE_STEP__(b, posB, nB, dirB, nodeB, labelB, goB)
#undef SPC_TREE_STEP__
    }
}

// Label caching.  The reference classifies with labelUnit(position, normal, direction) and re-derives two labels for every
// connection and one for every RMIS update (rmis.h:58-79, 131-151): the light-tree label of the EYE-side vertex and the eye-tree
// label of the LIGHT-side vertex, each seen from the other.  With DIR_JUDGE 0 (optixPathTracer.h:39) the classifiers never split
// on the direction -- classTree_host.h builds position / normal nodes only -- so a vertex's label under either tree is a property
// of the vertex alone.  The timed kernels therefore classify every vertex ONCE under both trees when it is created (the same
// lock-step pair of descents the vertex step already pays) and carry the labels along: EyeVertex::lsub, and the light vertex's
// eye-tree label + 1 in spcbpt_light_vertex::pad (0 = not computed: an imported cache; the connection then descends as before).
// 13 of the 17 descents per eye path disappear, and with them the longest dependent fetch chain of the connect phase; every label
// is the label the reference computes.  CACHE = false (the counting instantiations, the per-function harness) evaluates in the reference's order and charges its events; a caller-supplied tree WITH direction nodes (type 2) runs on
// those instantiations (Context::tree_has_direction).

// Gamma(e,l)/Q[l] (optixPathTracer.h:173-189); the product always runs with a full tuple installed
template <bool COUNT>
SPC_DEV float gamma_ss(const KParams& p, int e, int l, Counts<COUNT>& cn) {
    // the timed kernels read the quotient from a table of the same FP32 operations done once per tuple (layout.h KParams::gamma_q):
    // one load instead of three and no division per evaluation, ~12 evaluations per eye path
    if (p.gamma_q) { cn.add(C_GQ, 1); return p.gamma_q[(size_t)e * SPCBPT_NUM_SUBSPACE + l]; }
    const float* row = p.cmf_gamma + (size_t)e * SPCBPT_NUM_SUBSPACE;
    const float g = l == 0 ? row[0] : row[l] - row[l - 1];
    cn.add(C_GQ, l == 0 ? 2 : 3);
    return g / p.Q[l];
}

// binary_sample (cuProg.h:245-264): bespoke bisection restated exactly
template <bool COUNT>
SPC_DEV int binary_sample(const float* cmf, int size, uint32_t& seed, float& pmf, Counts<COUNT>& cn) {
    const float index = rnd(seed);
    int mid = size / 2 - 1, l = 0, r = size;
    while (r - l > 1) {
        cn.add(C_CMF);
        if (index < cmf[mid]) r = mid + 1;
        else l = mid + 1;
        mid = (l + r) / 2 - 1;
    }
    pmf = l == 0 ? cmf[l] : cmf[l] - cmf[l - 1];
    return l;
}

// uniformSample (cuProg.h:283-289): "plain BDPT" draws the light vertex uniformly over the whole cache, pmf 1 / vertex_count
// (double division, as `1.0 / vertex_count` is written).  `rnd * vertex_count` can round up to vertex_count in FP32 for large
// caches -- the reference would then read one past jump_buffer; clamped here.
SPC_DEV int uniform_sample_index(int vertex_count, uint32_t& seed, float& pmf) {   // the place in the jump buffer
    pmf = (float)(1.0 / (double)vertex_count);
    return min((int)(rnd(seed) * (float)vertex_count), vertex_count - 1);
}
SPC_DEV int uniform_sample(const int32_t* jump, int vertex_count, uint32_t& seed, float& pmf) {
    return jump[uniform_sample_index(vertex_count, seed, pmf)];
}

// sampleFirstStage (cuProg.h:290-301) = binary_sample over the 1000-entry CMF row of the eye subspace: ten DEPENDENT probes.
// For a non-decreasing CMF the bisection returns the first bin with u < cmf[bin], i.e. the number of entries <= u, which THREE
// counting passes find in three round trips of 4 / 2 / 2 independent 16-B loads (layout.h CMF2_*: 16 coarse entries row[64 k + 63],
// the 8 middle entries row[8 m + 7] of coarse group k, the 8 entries of middle group m).  The two CMF values of the pmf need no
// fetch of their own: in a non-decreasing row cmf[l] is the smallest value > u of the last group and cmf[l - 1] the largest value
// <= u among everything the passes have read (the previous entry of the same group, or -- at a group's first entry -- the last
// entry of the group before, which IS the middle / coarse value in front of the one that was counted).  Same bin, same pmf, same
// random number; the probe counter (algorithmic bytes) is charged what the bisection would have probed.
// (Rounds 1-4 ran two levels of 32: 16 loads and two more for the pmf per sample, 54 per vertex; this form reads 8 per sample and,
// with the coarse level shared by the CONNECTION_N samples of a vertex, 16 per vertex.)
struct Cmf3 { int count; float lo, hi; };   // entries <= u so far; largest entry <= u (-inf: none); smallest entry > u of the LAST pass
SPC_DEV void cmf3_pass(float4 q, float u, Cmf3& c) {
    const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool le = v[i] <= u;
        c.count += le ? 1 : 0;
        c.lo = fmaxf(c.lo, le ? v[i] : -INFINITY);
        c.hi = fminf(c.hi, le ? INFINITY : v[i]);
    }
}
// the CONNECTION_N (or fewer) samples of ONE eye subspace with the random numbers u[0 .. N): the coarse quads are fetched once
template <int N, bool SERIAL = false>   // SERIAL: the middle and fine passes of one sample after the other (fewer registers in flight)
SPC_DEV void sample_first_stage_n(const float* cmf_gamma2, int eye_subspace, const float u[N], int l[N], float pmf[N]) {
    const float4* R = reinterpret_cast<const float4*>(cmf_gamma2 + (size_t)eye_subspace * CMF2_ROW);
    const float4 c0 = R[0], c1 = R[1], c2 = R[2], c3 = R[3];
    Cmf3 s[N];
    float4 a[N], b[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        s[i].count = 0; s[i].lo = -INFINITY; s[i].hi = INFINITY;
        cmf3_pass(c0, u[i], s[i]); cmf3_pass(c1, u[i], s[i]); cmf3_pass(c2, u[i], s[i]); cmf3_pass(c3, u[i], s[i]);
    }
    if (SERIAL) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const float4* M = R + CMF2_COARSE / 4 + (size_t)s[i].count * 2;
            const float4 m0 = M[0], m1 = M[1];
            s[i].count *= 8; cmf3_pass(m0, u[i], s[i]); cmf3_pass(m1, u[i], s[i]);
            const float4* F = R + (CMF2_COARSE + CMF2_MID) / 4 + (size_t)s[i].count * 2;
            const float4 f0 = F[0], f1 = F[1];
            s[i].count *= 8; s[i].hi = INFINITY;
            cmf3_pass(f0, u[i], s[i]); cmf3_pass(f1, u[i], s[i]);
            l[i] = s[i].count;
            pmf[i] = s[i].count == 0 ? s[i].hi : s[i].hi - s[i].lo;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < N; i++) { const float4* M = R + CMF2_COARSE / 4 + (size_t)s[i].count * 2; a[i] = M[0]; b[i] = M[1]; }
#pragma unroll
    for (int i = 0; i < N; i++) { s[i].count *= 8; cmf3_pass(a[i], u[i], s[i]); cmf3_pass(b[i], u[i], s[i]); }
#pragma unroll
    for (int i = 0; i < N; i++) { const float4* F = R + (CMF2_COARSE + CMF2_MID) / 4 + (size_t)s[i].count * 2; a[i] = F[0]; b[i] = F[1]; }
#pragma unroll
    for (int i = 0; i < N; i++) {
        s[i].count *= 8; s[i].hi = INFINITY;
        cmf3_pass(a[i], u[i], s[i]); cmf3_pass(b[i], u[i], s[i]);
        l[i] = s[i].count;
        pmf[i] = s[i].count == 0 ? s[i].hi : s[i].hi - s[i].lo;
    }
}
// Guided form (round 5, the cutpoint method): a guide table names, for the bucket (int)(u * buckets) of the random number, a place g
// that the answer cannot precede (layout.h: KParams::guide, cmf_guide1), and the entries from g - 1 on are read in aligned windows of
// eight (two 16-B loads) until one is above u: in a non-decreasing CMF the answer is the number of entries <= u, cmf[answer] the
// smallest entry > u of the last window and cmf[answer - 1] the largest entry <= u read (entry g - 1 is in the first window for that).
// One guide entry and -- nearly always -- one window per sample instead of 32 values in three round trips (first stage) or one
// probe per level and two for the pmf (second stage); same bin, same pmf, same random number.
#ifndef SPC_GUIDE
#define SPC_GUIDE 1
#endif
#ifndef SPC_GUIDE_WINDOW
#define SPC_GUIDE_WINDOW 8   // 4: windows of one 16-B load (fewer values read, more often a second round trip: measured, section 25)
#endif
struct GuideScan { int cnt; float lo, hi; };   // entries <= u so far; the largest of them; the smallest entry > u
// the entries at places [pos, pos + 8) of an array, of which [first, end) take part.  first - pos <= 3 (pos is first rounded down to a
// quad, or a later window), so only the first three entries can lie in front of it.  RANGE = false: every entry takes part (the
// first stage: a row of its own, padded with 2.0 -- the entries in front of the guide's place are <= u like the one it names, so the
// caller counts from the window's start instead of masking them).
template <bool RANGE = true>
SPC_DEV void guide_window(float4 q0, float4 q1, int pos, int first, int end, float u, GuideScan& s) {
    const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int i = 0; i < SPC_GUIDE_WINDOW; i++) {
        const bool in = !RANGE || ((i >= 3 || pos + i >= first) && pos + i < end);
        const bool le = in && v[i] <= u, gt = in && !(v[i] <= u);
        s.cnt += le ? 1 : 0;
        s.lo = fmaxf(s.lo, le ? v[i] : -INFINITY);
        s.hi = fminf(s.hi, gt ? v[i] : INFINITY);
    }
}
// sampleFirstStage through the guide table; returns the number of windows read (the executed-order probe count)
SPC_DEV int sample_first_stage_guided(const float* cmf_gamma2, const uint16_t* guide1, int eye_subspace, float u, int& l, float& pmf) {
    const float* fine = cmf_gamma2 + (size_t)eye_subspace * CMF2_ROW + CMF2_COARSE + CMF2_MID;   // 1000 entries, 2.0 up to CMF2_FINE
    const int g = guide1[(size_t)eye_subspace * CMF_GUIDE1 + (int)(u * (float)CMF_GUIDE1)];
    const int c0 = max(g - 1, 0);
    int pos = c0 & ~3, windows = 0;
    GuideScan s = {pos, -INFINITY, INFINITY};   // (the entries of the first quad in front of c0 are counted with it: all <= u)
    do {
        const float4 q0 = *reinterpret_cast<const float4*>(fine + pos), q1 = SPC_GUIDE_WINDOW == 8 ? *reinterpret_cast<const float4*>(fine + pos + 4) : q0;
        guide_window<false>(q0, q1, pos, c0, CMF2_FINE, u, s);
        pos += SPC_GUIDE_WINDOW; windows++;
    } while (!(s.hi < INFINITY) && pos < CMF2_FINE);
    l = s.cnt;
    pmf = l == 0 ? s.hi : s.hi - s.lo;
    return windows;
}
// ... of the CONNECTION_N samples of one eye subspace: the guide entries in flight together, the windows one after the other
template <int N>
SPC_DEV void sample_first_stage_guided_n(const float* cmf_gamma2, const uint16_t* guide1, int eye_subspace, const float u[N], int l[N], float pmf[N], int windows[N]) {
    const float* fine = cmf_gamma2 + (size_t)eye_subspace * CMF2_ROW + CMF2_COARSE + CMF2_MID;
    int g[N];
#pragma unroll
    for (int i = 0; i < N; i++) g[i] = guide1[(size_t)eye_subspace * CMF_GUIDE1 + (int)(u[i] * (float)CMF_GUIDE1)];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int c0 = max(g[i] - 1, 0);
        int pos = c0 & ~3;
        GuideScan s = {pos, -INFINITY, INFINITY};
        windows[i] = 0;
        do {
            const float4 q0 = *reinterpret_cast<const float4*>(fine + pos), q1 = SPC_GUIDE_WINDOW == 8 ? *reinterpret_cast<const float4*>(fine + pos + 4) : q0;
            guide_window<false>(q0, q1, pos, c0, CMF2_FINE, u[i], s);
            pos += SPC_GUIDE_WINDOW; windows[i]++;
        } while (!(s.hi < INFINITY) && pos < CMF2_FINE);
        l[i] = s.cnt;
        pmf[i] = s.cnt == 0 ? s.hi : s.hi - s.lo;
    }
}
SPC_DEV int bisection_probes(int l, int size) {   // the probes of the reference's bisection on its way to bin l
    int n = 0, mid = size / 2 - 1, a = 0, b = size;
    while (b - a > 1) {
        n++;
        if (l <= mid) b = mid + 1; else a = mid + 1;
        mid = (a + b) / 2 - 1;
    }
    return n;
}
template <bool COUNT, bool EXEC = false>   // EXEC: charge what the guided form really reads (one guide entry, eight values per window), not the bisection's probes
SPC_DEV int sample_first_stage(const KParams& p, int eye_subspace, uint32_t& seed, float& pmf, Counts<COUNT>& cn) {
    // a caller-supplied matrix with a decreasing row (not a CMF) keeps the bisection, whose answer is then its own definition
    if (!p.cmf_gamma2) return binary_sample(p.cmf_gamma + (size_t)eye_subspace * SPCBPT_NUM_SUBSPACE, SPCBPT_NUM_SUBSPACE, seed, pmf, cn);
    const float u[1] = {rnd(seed)};
    int l[1];
    float pm[1];
#if SPC_GUIDE
    const int windows = sample_first_stage_guided(p.cmf_gamma2, p.cmf_guide1, eye_subspace, u[0], l[0], pm[0]);
    pmf = pm[0];
    if (COUNT) cn.add(C_CMF, EXEC ? 1u + (unsigned)SPC_GUIDE_WINDOW * (unsigned)windows : (unsigned)bisection_probes(l[0], SPCBPT_NUM_SUBSPACE));
#else
    sample_first_stage_n<1>(p.cmf_gamma2, eye_subspace, u, l, pm);
    pmf = pm[0];
    if (COUNT) cn.add(C_CMF, EXEC ? 32u : (unsigned)bisection_probes(l[0], SPCBPT_NUM_SUBSPACE));
#endif
    return l[0];
}

// ---- recursive MIS (rmis.h) ------------------------------------------------------
// The fields of a path vertex the RMIS recursions read, shared by eye and light vertices.
struct VCore {
    f3 pos, n, color, lastPos;
    float lnp;  // lastNormalProjection
    int mat;
    bool lld;   // is_LL_DIRECTION (BDPTVertex.h:67): the vertex was hit straight from the environment map (light vertices only)
};
SPC_DEV float rr_of(f3 color) { return fmaxf(max3(color), SPCBPT_MIN_RR_RATE); }  // getRR rmis.h:28-40 (q10)

// getLast_pdf (rmis.h:41-51): pdf of stepping from v back to its predecessor given arrival from in_dir
SPC_DEV float rmis_last_pdf(const Pbr& mat, const VCore& v, f3 in_dir) {
    const f3 out_vec = v.lastPos - v.pos;
    const f3 out_dir = normalize(out_vec);
    // rmis.h:45-47: the step back from a vertex lit straight by the sky leads to a direction, not to a point: no area measure
    float pdf = v.lld ? bsdf_pdf(mat, v.n, in_dir, out_dir) : bsdf_pdf(mat, v.n, in_dir, out_dir) / dot(out_vec, out_vec) * v.lnp;
    return pdf * rr_of(v.color);
}
// getFluxMultiplier (rmis.h:102-118)
template <bool ENV = true>
SPC_DEV f3 rmis_flux_multiplier(const Pbr& mat, const VCore& v, f3 in_dir, f3 out_dir) {
    const f3 flux_ratio = brdf_div<ENV>(mat, bsdf_eval(mat, v.n, in_dir, out_dir), v.n, out_dir);   // rmis.h:105
    const float pdf_ratio = bsdf_pdf(mat, v.n, in_dir, out_dir);
    const float rr = rr_of(v.color);
    const float cos_theta = fabsf(dot(v.n, out_dir));
    return flux_ratio * cos_theta / pdf_ratio / rr;
}
// getPdf (rmis.h:153-172): pdf of generating `end` from `begin` given arrival from in_dir
SPC_DEV float rmis_get_pdf(const Pbr& mat, const VCore& begin, f3 end_pos, f3 end_n, f3 in_dir) {
    const f3 out_vec = end_pos - begin.pos;
    const f3 out_dir = normalize(out_vec);
    float pdf = bsdf_pdf(mat, begin.n, in_dir, out_dir) / dot(out_vec, out_vec) * fabsf(dot(out_dir, end_n));
    return pdf * rr_of(begin.color);
}
// getPdf_from_light_source (rmis.h:173-188)
SPC_DEV float rmis_pdf_from_light(f3 light_pos, f3 light_n, f3 end_pos, f3 end_n) {
    const f3 conn_vec = end_pos - light_pos;
    const f3 conn_dir = normalize(conn_vec);
    const float pdf_angle = fabsf(dot(light_n, conn_dir)) * kInvPi;
    const float angle2a = fabsf(dot(end_n, conn_dir)) / dot(conn_vec, conn_vec);
    return pdf_angle * angle2a;
}

// Eye-side vertex kept in registers while walking (the live BDPTVertex fields of the eye sub-path)
struct EyeVertex {
    VCore c;
    f3 flux, R3;       // flux, RMIS_pointer_3
    float pdf, singlePdf;
    int sub, lastZone, depth;
    int lsub;          // label caching (see label_cache below): the vertex's own light-tree label
};

// tracing_weight_eye (rmis.h:131-151) with Last = `last`, Mid at `mid_pos`
template <bool COUNT>
SPC_DEV float rmis_weight_eye(const KParams& p, const VCore& last, int last_depth, int last_lastZone, f3 mid_pos, Counts<COUNT>& cn) {
    if (last_depth == 1) return 0.0f;
    const f3 inver_dir = normalize(mid_pos - last.pos);
    const int light_label = tree_label(p.light_tree, last.pos, last.n, inver_dir, cn);
    return gamma_ss(p, last_lastZone, light_label, cn) * (float)SPCBPT_CONNECTION_N;
}
// the same two weights with the relabel already done (tree_label2)
template <bool COUNT>
SPC_DEV float rmis_weight_eye_l(const KParams& p, int last_depth, int last_lastZone, int light_label, Counts<COUNT>& cn) {
    if (last_depth == 1) return 0.0f;
    return gamma_ss(p, last_lastZone, light_label, cn) * (float)SPCBPT_CONNECTION_N;
}
template <bool COUNT>
SPC_DEV float rmis_weight_light_l(const KParams& p, int last_lastZone, float last_lum, int eye_label, Counts<COUNT>& cn) {
    return gamma_ss(p, eye_label, last_lastZone, cn) * last_lum * (float)SPCBPT_CONNECTION_N;
}
// tracing_weight_light (rmis.h:58-79) with Last = light vertex `last`
template <bool COUNT>
SPC_DEV float rmis_weight_light(const KParams& p, const VCore& last, int last_lastZone, float last_lum, f3 mid_pos, Counts<COUNT>& cn) {
    const f3 inver_dir = normalize(mid_pos - last.pos);
    const int eye_label = tree_label(p.eye_tree, last.pos, last.n, inver_dir, cn);
    return gamma_ss(p, eye_label, last_lastZone, cn) * last_lum * (float)SPCBPT_CONNECTION_N;
}

template <bool ENV = true>
SPC_DEV VCore core_of(const LightVertex& b) {
    VCore c;
    c.pos = ld3(b.position); c.n = ld3(b.normal); c.color = ld3(b.color); c.lastPos = ld3(b.last_position);
    c.lnp = b.last_normal_projection; c.mat = b.material_id;
    c.lld = ENV && (b.pad & SPCBPT_LV_LAST_DIRECTION) != 0u;
    return c;
}

// A connection whose value is exactly zero whatever the visibility: the eye vertex sees the light vertex from
// behind its own surface (bsdf_eval returns 0 for N.V <= 0), or the light vertex faces away (N.L <= 0 on a surface vertex,
// the one-sided term on an emitter vertex).  Same vectors and the same normalize() as connect_vertices.
// ... and for a direction of the environment map: direction_connect_ZGCBPT contributes only with the sky above the eye vertex's surface
SPC_DEV bool null_connection_direction(f3 an, f3 bn) { return !(dot(an, -bn) > 0.0f); }
SPC_DEV bool null_connection(f3 apos, f3 an, f3 bpos, f3 bn) {
    const f3 connectDir = normalize(apos - bpos);
    return dot(an, -connectDir) <= 0.0f || dot(bn, connectDir) < 0.0f;
}

// direction_connect_ZGCBPT (raygen.cu:234-252) with rmis::connection_direction_lightSource (rmis.h:249-280): the light vertex is a
// direction of the environment map (type ENV: normal = minus the sky direction, position = its point on the sky disk).
template <bool COUNT, bool CACHE>
SPC_DEV f3 connect_direction(const KParams& p, const EyeVertex& a, const LightVertex& b, Counts<COUNT>& cn, float* w_out) {
    const DeviceScene& S = p.scene;
    const f3 bn = ld3(b.normal), bflux = ld3(b.flux);
    const f3 connectDir = -bn;
    if (w_out) *w_out = 0.0f;
    if (!(dot(a.c.n, connectDir) > 0.0f)) return mk3(0.0f);
    const f3 LA_DIR = normalize(a.c.lastPos - a.c.pos);
    const Pbr mat_a = load_pbr_colored(S, a.c.mat, a.c.color);
    const f3 f = bsdf_eval(mat_a, a.c.n, LA_DIR, connectDir) * dot(a.c.n, connectDir);
    const f3 lflux = bflux / b.pdf;
    // getLL_pdf(light, eye): the incoming direction runs from the eye vertex to the light vertex's POSITION on the sky disk (as written)
    const float LL_pdf_A = rmis_last_pdf(mat_a, a.c, normalize(ld3(b.position) - a.c.pos));
    const f3 fm0 = rmis_flux_multiplier(mat_a, a.c, -bn, LA_DIR);                       // getFluxMultiplier(eye, -connect_dir), connect_dir = light.normal
    int light_label = a.lsub;
    if (!CACHE && a.depth != 1) light_label = tree_label(p.light_tree, a.c.pos, a.c.n, -bn, cn);   // tracing_weight_eye: inver_dir = -Mid.normal for a direction (rmis.h:141)
    const float wA = rmis_weight_eye_l(p, a.depth, a.lastZone, light_label, cn);
    const f3 D_A_0 = a.R3 * LL_pdf_A * fm0 + mk3(wA);
    const float pdf_A = S.env.project_pdf * fabsf(dot(bn, a.c.n));                     // getPdf_from_light_source, direction branch (183-187)
    const float fm1 = (float)(1.0 / S.env.project_pdf);
    const float D_A = sum3(D_A_0 * pdf_A * fm1 * lflux / a.singlePdf);
    const float weight = sum3(gamma_ss(p, a.sub, b.subspace_id, cn) * lflux * (float)SPCBPT_CONNECTION_N);
    const float pdf_B = bsdf_pdf(mat_a, a.c.n, LA_DIR, -bn) * rr_of(a.c.color);        // getPdf(eye, light, LB), end is a direction (158-162)
    const float D_B = b.rmis_pointer * pdf_B / b.single_pdf;
    const float w_rmis = weight / (weight + D_A + D_B);
    if (w_out) *w_out = w_rmis;
    return a.flux / a.pdf * f * bflux / b.pdf * w_rmis;
}

// connectVertex_SPCBPT (raygen.cu:253-303) with rmis::general_connection / connection_lightSource
// (rmis.h:212-247 / 281-313) fused: every BSDF lobe is fetched once.
// ENV = false: the scene has no environment map -- no vertex carries a direction flag, and the two tests fold away (the timed
// kernels of a scene without a sky are instantiated so: 3 % of the frame) -- and no `brdf`-flagged material (brdf_div)
template <bool COUNT, bool CACHE = false, bool ENV = true>
SPC_DEV f3 connect_vertices(const KParams& p, const EyeVertex& a, const LightVertex& b, Counts<COUNT>& cn, float* w_out = nullptr) {
    if (ENV && (b.pad & SPCBPT_LV_DIRECTION)) return connect_direction<COUNT, CACHE>(p, a, b, cn, w_out);   // raygen.cu:255-258
    const DeviceScene& S = p.scene;
    const f3 bpos = ld3(b.position), bn = ld3(b.normal), bflux = ld3(b.flux);
    const f3 connectVec = a.c.pos - bpos;
    const f3 connectDir = normalize(connectVec);
    const float r2 = dot(connectVec, connectVec);
    const float G = fabsf(dot(a.c.n, connectDir)) * fabsf(dot(bn, connectDir)) / r2;
    const f3 LA_DIR = normalize(a.c.lastPos - a.c.pos);
    const Pbr mat_a = load_pbr_colored(S, a.c.mat, a.c.color);
    const f3 fa = brdf_div<ENV>(mat_a, bsdf_eval(mat_a, a.c.n, -connectDir, LA_DIR), a.c.n, connectDir);   // raygen.cu:271
    const f3 lflux = bflux / b.pdf;  // `flux` of the rmis functions

    // ---- eye side terms shared by both connection kinds
    const float LL_pdf_A = rmis_last_pdf(mat_a, a.c, -connectDir);                 // getLL_pdf(light, eye)
    const f3 fm0 = rmis_flux_multiplier<ENV>(mat_a, a.c, -connectDir, LA_DIR);      // getFluxMultiplier(eye, -connect_dir)
    // the two relabels of the connection (light-tree label of the eye vertex seen from b, eye-tree label of the light vertex
    // seen from a) in one lock-step descent; the first is skipped at depth 1, the second for an emitter vertex, as in rmis.h
    int light_label, eye_label;
    if (CACHE) {
        light_label = a.lsub;                    // unused at depth 1, like the descent it replaces
        eye_label = (int)(b.pad & 0xffffu) - 1;  // unused for an emitter vertex
        // imported cache without labels (0), or a word that is not a label at all (spcbpt.h: never used as a row index unchecked)
        if ((b.pad & 0xffffu) - 1u >= (uint32_t)SPCBPT_NUM_SUBSPACE && b.depth != 0) eye_label = tree_label(p.eye_tree, bpos, bn, normalize(a.c.pos - bpos), cn);
    } else {
        tree_label2(p.light_tree, a.c.pos, a.c.n, normalize(bpos - a.c.pos), a.depth != 1,
                    p.eye_tree, bpos, bn, normalize(a.c.pos - bpos), b.depth != 0, light_label, eye_label, cn);
    }
    const float wA = rmis_weight_eye_l(p, a.depth, a.lastZone, light_label, cn);    // tracing_weight_eye(light, eye)
    const f3 D_A_0 = a.R3 * LL_pdf_A * fm0 + mk3(wA);
    const float weight = sum3(gamma_ss(p, a.sub, b.subspace_id, cn) * lflux * (float)SPCBPT_CONNECTION_N);
    const float pdf_B = rmis_get_pdf(mat_a, a.c, bpos, bn, LA_DIR);                 // getPdf(eye, light, LB)

    f3 fb;
    float D_A, D_B;
    if (b.depth == 0) {  // connection_lightSource
        fb = dot(bn, -connectDir) > 0.0f ? mk3(0.0f) : mk3(1.0f);
        const float pdf_A = rmis_pdf_from_light(bpos, bn, a.c.pos, a.c.n);
        D_A = sum3(D_A_0 * pdf_A * kPi * lflux / a.singlePdf);
        D_B = b.rmis_pointer * pdf_B / b.single_pdf;
    } else {  // general_connection
        const VCore bc = core_of<ENV>(b);
        const Pbr mat_b = load_pbr_colored(S, bc.mat, bc.color);
        const f3 LB_DIR = normalize(bc.lastPos - bc.pos);
        fb = brdf_div<ENV>(mat_b, bsdf_eval(mat_b, bn, connectDir, LB_DIR), bn, connectDir);   // raygen.cu:278
        const float pdf_A = rmis_get_pdf(mat_b, bc, a.c.pos, a.c.n, LB_DIR);        // getPdf(light, eye, LA)
        const f3 fm1 = rmis_flux_multiplier<ENV>(mat_b, bc, LB_DIR, connectDir);
        D_A = sum3(D_A_0 * pdf_A * fm1 * lflux / a.singlePdf);
        const float LL_pdf_B = rmis_last_pdf(mat_b, bc, connectDir);               // getLL_pdf(eye, light)
        const float wB = rmis_weight_light_l(p, b.last_zone_id, b.last_lum, eye_label, cn);
        D_B = (b.rmis_pointer * LL_pdf_B + wB) * pdf_B / b.single_pdf;
    }
    const float w_rmis = weight / (weight + D_A + D_B);
    if (w_out) *w_out = w_rmis;   // per-function harness only (unit.hip); the render kernels pass nothing
    const f3 contri = a.flux * bflux * fa * fb * G;
    const f3 ans = contri / (a.pdf * b.pdf) * w_rmis;
    return ans;
}

SPC_DEV bool is_invalid(f3 a) {  // ISINVALIDVALUE raygen.cu:43
    return a.x > 100000.0f || isnan(a.x) || a.y > 100000.0f || isnan(a.y) || a.z > 100000.0f || isnan(a.z);
}

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

// Part of device_lib.h (split in round 6 for readability; included by it, in this order, inside the one translation unit of each
// .hip file -- the device code generated is the same as from the single header: tests/test_codegen_guard.py):
// subspace classification (classTree_common.h) and the two resampling stages (cuProg.h:245-301) through guide tables.
#pragma once
#include "device_lib.h"

namespace spc {

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

}  // namespace spc

// developer: offline quality of the BVH builder (csrc/lbvh.cpp) on a dumped mesh -- no GPU.  Builds the tree exactly as the
// product does (SPCBPT_BVH*, SPCBPT_BVH_REINSERT ... are read by the builder), then traverses the 4-wide quantised nodes on the CPU
// with the device's rules (children sorted by entry distance, any-hit rays end at the first hit) for
//   closest-hit rays  from area-weighted surface points along cosine-weighted directions (what path segments look like), and
//   shadow rays       between pairs of area-weighted surface points (what connections look like),
// and prints node visits and triangle tests per ray.  mesh file: int32 nv, nt; float32 vertices[nv][3]; uint32 indices[nt][3]
//   g++ -O2 -std=c++17 -o /tmp/bvh_eval tools/bvh_eval.cpp && /tmp/bvh_eval mesh.bin [rays]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../spcbpt-optix7_amd/csrc/lbvh.cpp"

using namespace spc;

struct V3 { float x, y, z; };
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V3 norm(V3 a) { float l = std::sqrt(dot(a, a)); return a * (1.0f / l); }

static uint64_t g_depth_ray[64], g_depth_step[64];   // stack depth: per ray its maximum, per step (node visit) the entries held -- what an N-entry LDS stack would have to spill
static std::vector<uint64_t> g_visits;   // per node: how often it was visited (top-of-tree share, printed at the end)
struct Stats { double nodes = 0, tris = 0, leaves = 0, rays = 0, hits = 0, leaves_exact = 0, tris_exact = 0, tri_steps = 0; };   // tri_steps: triangle steps of a lane when a fan pair is one step (Lbvh::pairs)   // *_exact: leaf visits / tests left if the leaf's child box were the exact bounds of its triangles

static bool tri_hit(const float* q, V3 o, V3 d, float tmin, float tmax, float& t) {
    V3 v0{q[0], q[1], q[2]}, v1{q[4], q[5], q[6]}, v2{q[8], q[9], q[10]};
    V3 e1 = v1 - v0, e2 = v2 - v0, p = cross(d, e2);
    float det = dot(e1, p);
    if (det == 0.0f) return false;
    float inv = 1.0f / det;
    V3 tv = o - v0;
    float u = dot(tv, p) * inv;
    if (u < 0 || u > 1) return false;
    V3 qq = cross(tv, e1);
    float v = dot(d, qq) * inv;
    if (v < 0 || u + v > 1) return false;
    t = dot(e2, qq) * inv;
    return t > tmin && t < tmax;
}

static bool traverse(const Lbvh& B, V3 o, V3 d, float tmin, float tmax, bool any, Stats& st) {
    uint32_t stack[256];
    int sp = 0;
    uint32_t cur = 0;   // node index, or leaf ref with bit 31
    float best = tmax;
    bool hit = false;
    const float inv[3] = {1.0f / (std::fabs(d.x) > 1e-20f ? d.x : 1e-20f), 1.0f / (std::fabs(d.y) > 1e-20f ? d.y : 1e-20f), 1.0f / (std::fabs(d.z) > 1e-20f ? d.z : 1e-20f)};
    const float oo[3] = {o.x, o.y, o.z};
    st.rays++;
    int sp_max = 0;
    struct DepthNote { int& m; ~DepthNote() { g_depth_ray[std::min(m, 63)]++; } } note{sp_max};
    while (true) {
        sp_max = std::max(sp_max, sp);
        if (cur & 0x80000000u) {
            if (cur != 0x80000000u) {
                const int first = (int)((cur & 0x7fffffffu) >> 3), cnt = (int)(cur & 7u);
                st.leaves++;
                {   // would the EXACT box of the leaf's triangles have been entered?  (upper bound of what tighter leaf boxes can save)
                    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
                    for (int t = first; t < first + cnt; t++)
                        for (int v = 0; v < 3; v++)
                            for (int k = 0; k < 3; k++) { const float x = B.tris[(size_t)t * 16 + 4 * v + k]; lo[k] = std::min(lo[k], x); hi[k] = std::max(hi[k], x); }
                    float t0 = tmin, t1 = best;
                    for (int k = 0; k < 3; k++) { float a = (lo[k] - oo[k]) * inv[k], b = (hi[k] - oo[k]) * inv[k]; if (a > b) std::swap(a, b); t0 = std::max(t0, a); t1 = std::min(t1, b); }
                    if (t0 <= t1 * 1.0000004f) { st.leaves_exact++; st.tris_exact += cnt; }
                }
                for (int t = first; t < first + cnt; t++) {
                    st.tris++;
                    uint32_t fl = 0, flp = 0;
                    if (!B.pairs.empty()) { memcpy(&fl, &B.pairs[(size_t)t * 16 + 15], 4); if (t > first) memcpy(&flp, &B.pairs[(size_t)(t - 1) * 16 + 15], 4); }
                    if (!(flp & 1u)) st.tri_steps++;   // the second half of a pair rides on its first half's step
                    float th;
                    if (tri_hit(&B.tris[(size_t)t * 16], o, d, tmin, best, th)) { best = th; hit = true; if (any) { st.hits++; return true; } }
                }
            }
            if (sp == 0) break;
            cur = stack[--sp];
            continue;
        }
        st.nodes++;
        g_depth_step[std::min(sp, 63)]++;
        if (!g_visits.empty()) g_visits[cur]++;
        uint32_t w[16];
        memcpy(w, &B.nodes[(size_t)cur * 16], sizeof(w));
        float org[3]; memcpy(org, w, 12);
        float sc[3];
        for (int k = 0; k < 3; k++) { const uint32_t e = ((w[3] >> (8 * k)) & 0xffu) << 23; memcpy(&sc[k], &e, 4); }
        const uint32_t qlo[3] = {w[4], w[5], w[6]}, qhi[3] = {w[7], w[8], w[9]};
        const uint32_t refs[4] = {w[10], w[11], w[12], w[13]};
        float key[4]; uint32_t rf[4]; int n = 0;
        for (int i = 0; i < 4; i++) {
            if (refs[i] == 0x80000000u) continue;
            float t0 = tmin, t1 = best;
            for (int k = 0; k < 3; k++) {
                const float lo = org[k] + (float)((qlo[k] >> (8 * i)) & 0xffu) * sc[k], hi = org[k] + (float)((qhi[k] >> (8 * i)) & 0xffu) * sc[k];
                float a = (lo - oo[k]) * inv[k], b = (hi - oo[k]) * inv[k];
                if (a > b) std::swap(a, b);
                t0 = std::max(t0, a); t1 = std::min(t1, b);
            }
            if (t0 <= t1 * 1.0000004f) { key[n] = t0; rf[n] = refs[i]; n++; }
        }
        for (int i = 1; i < n; i++) for (int j = i; j > 0 && key[j] < key[j - 1]; j--) { std::swap(key[j], key[j - 1]); std::swap(rf[j], rf[j - 1]); }
        if (n == 0) { if (sp == 0) break; cur = stack[--sp]; continue; }
        for (int i = n - 1; i >= 1; i--) stack[sp++] = rf[i];
        cur = rf[0];
    }
    if (hit) st.hits++;
    return hit;
}

// ---- what a WIDER node would buy (offline only: BVH_EVAL_WIDE=8): the builder's binary tree folded into W-wide nodes with the same rule as
// the product's 4-wide collapse (largest child opened first), EXACT child boxes, the same traversal rules and the same rays ---------------
struct WNode { float lo[8][3], hi[8][3]; int ref[8], cnt[8], n; };   // ref >= 0: wide node index; ref < 0: leaf ~first, cnt triangles
struct WideTree {
    std::vector<WNode> nodes;
    const std::vector<float>* bin = nullptr;
    int W = 8;
    static int geti(const float* p) { int v; memcpy(&v, p, 4); return v; }
    struct Slot { float lo[3], hi[3]; int ref, count; };
    void children_of(int b, Slot& x, Slot& y) const {
        const float* q = &(*bin)[(size_t)b * 16];
        for (int k = 0; k < 3; k++) { x.lo[k] = q[k]; x.hi[k] = q[4 + k]; y.lo[k] = q[8 + k]; y.hi[k] = q[12 + k]; }
        x.ref = geti(q + 3); y.ref = geti(q + 7); x.count = geti(q + 11); y.count = geti(q + 15);
    }
    static float area(const Slot& s) { const float dx = s.hi[0] - s.lo[0], dy = s.hi[1] - s.lo[1], dz = s.hi[2] - s.lo[2]; return dx * dy + dy * dz + dz * dx; }
    int emit(int bnode) {
        Slot s[8]; int n = 2;
        children_of(bnode, s[0], s[1]);
        while (n < W) {
            int best = -1; float ba = -1;
            for (int i = 0; i < n; i++) if (s[i].ref >= 0 && area(s[i]) > ba) { ba = area(s[i]); best = i; }
            if (best < 0) break;
            Slot a, b; children_of(s[best].ref, a, b); s[best] = a; s[n++] = b;
        }
        const int id = (int)nodes.size();
        nodes.emplace_back();
        for (int i = 0; i < n; i++) { const int r = s[i].ref >= 0 ? emit(s[i].ref) : s[i].ref; WNode& w = nodes[id]; for (int k = 0; k < 3; k++) { w.lo[i][k] = s[i].lo[k]; w.hi[i][k] = s[i].hi[k]; } w.ref[i] = r; w.cnt[i] = s[i].ref >= 0 ? 0 : s[i].count; }
        nodes[id].n = n;
        return id;
    }
};
struct WStats { double nodes = 0, tri_steps = 0, rays = 0, depth_hist[64] = {0}; };
static bool wtraverse(const WideTree& T, const Lbvh& B, V3 o, V3 d, float tmin, float tmax, bool any, WStats& st) {
    struct E { int ref, cnt; };
    E stack[512]; int sp = 0, sp_max = 0;
    E cur{0, 0};
    float best = tmax; bool hit = false;
    const float inv[3] = {1.0f / (std::fabs(d.x) > 1e-20f ? d.x : 1e-20f), 1.0f / (std::fabs(d.y) > 1e-20f ? d.y : 1e-20f), 1.0f / (std::fabs(d.z) > 1e-20f ? d.z : 1e-20f)};
    const float oo[3] = {o.x, o.y, o.z};
    st.rays++;
    while (true) {
        sp_max = std::max(sp_max, sp);
        if (cur.ref < 0) {
            const int first = ~cur.ref;
            for (int t = first; t < first + cur.cnt; t++) {
                uint32_t flp = 0;
                if (t > first) memcpy(&flp, &B.pairs[(size_t)(t - 1) * 16 + 15], 4);
                if (!(flp & 1u)) st.tri_steps++;
                float th;
                if (tri_hit(&B.tris[(size_t)t * 16], o, d, tmin, best, th)) { best = th; hit = true; if (any) { st.depth_hist[std::min(sp_max, 63)]++; return true; } }
            }
            if (sp == 0) break;
            cur = stack[--sp];
            continue;
        }
        st.nodes++;
        const WNode& w = T.nodes[(size_t)cur.ref];
        float key[8]; E rf[8]; int n = 0;
        for (int i = 0; i < w.n; i++) {
            float t0 = tmin, t1 = best;
            for (int k = 0; k < 3; k++) { float a = (w.lo[i][k] - oo[k]) * inv[k], b = (w.hi[i][k] - oo[k]) * inv[k]; if (a > b) std::swap(a, b); t0 = std::max(t0, a); t1 = std::min(t1, b); }
            if (t0 <= t1 * 1.0000004f) { key[n] = t0; rf[n] = E{w.ref[i], w.cnt[i]}; n++; }
        }
        for (int i = 1; i < n; i++) for (int j = i; j > 0 && key[j] < key[j - 1]; j--) { std::swap(key[j], key[j - 1]); std::swap(rf[j], rf[j - 1]); }
        if (n == 0) { if (sp == 0) break; cur = stack[--sp]; continue; }
        for (int i = n - 1; i >= 1; i--) stack[sp++] = rf[i];
        cur = rf[0];
    }
    st.depth_hist[std::min(sp_max, 63)]++;
    return hit;
}

// ---- "open once": every 4-wide node with its largest internal child opened in place (up to 7 children whose refs are still 4-wide node
// ids / leaf refs, so that the tails could go on with the 4-wide records); the opened child's boxes re-quantised OUTWARD on the parent's
// 8-bit grid, as a device record would hold them (BVH_EVAL_OPEN1=1) ---------------------------------------------------------------------
struct ONode { float lo[7][3], hi[7][3]; uint32_t ref[7]; int n; };
static void decode4(const Lbvh& B, uint32_t node, float org[3], float sc[3], float lo[4][3], float hi[4][3], uint32_t ref[4]) {
    uint32_t w[16]; memcpy(w, &B.nodes[(size_t)node * 16], sizeof(w));
    memcpy(org, w, 12);
    for (int k = 0; k < 3; k++) { const uint32_t e = ((w[3] >> (8 * k)) & 0xffu) << 23; memcpy(&sc[k], &e, 4); }
    for (int i = 0; i < 4; i++) {
        ref[i] = w[10 + i];
        for (int k = 0; k < 3; k++) { lo[i][k] = org[k] + (float)((w[4 + k] >> (8 * i)) & 0xffu) * sc[k]; hi[i][k] = org[k] + (float)((w[7 + k] >> (8 * i)) & 0xffu) * sc[k]; }
    }
}
static std::vector<ONode> open_once(const Lbvh& B) {
    const size_t n = B.nodes.size() / 16;
    std::vector<ONode> out(n);
    for (size_t nd = 0; nd < n; nd++) {
        float org[3], sc[3], lo[4][3], hi[4][3]; uint32_t ref[4];
        decode4(B, (uint32_t)nd, org, sc, lo, hi, ref);
        int best = -1; float ba = -1;
        for (int i = 0; i < 4; i++) if (ref[i] != 0x80000000u && !(ref[i] & 0x80000000u)) { const float dx = hi[i][0] - lo[i][0], dy = hi[i][1] - lo[i][1], dz = hi[i][2] - lo[i][2]; const float a = dx * dy + dy * dz + dz * dx; if (a > ba) { ba = a; best = i; } }
        ONode& o = out[nd]; o.n = 0;
        for (int i = 0; i < 4; i++) {
            if (ref[i] == 0x80000000u || i == best) continue;
            for (int k = 0; k < 3; k++) { o.lo[o.n][k] = lo[i][k]; o.hi[o.n][k] = hi[i][k]; }
            o.ref[o.n++] = ref[i];
        }
        if (best >= 0) {
            float org2[3], sc2[3], lo2[4][3], hi2[4][3]; uint32_t ref2[4];
            decode4(B, ref[best], org2, sc2, lo2, hi2, ref2);
            for (int i = 0; i < 4; i++) {
                if (ref2[i] == 0x80000000u) continue;
                for (int k = 0; k < 3; k++) {   // outward on the PARENT's grid
                    const float a = std::floor((lo2[i][k] - org[k]) / sc[k]), b = std::ceil((hi2[i][k] - org[k]) / sc[k]);
                    o.lo[o.n][k] = org[k] + std::max(0.0f, std::min(255.0f, a)) * sc[k]; o.hi[o.n][k] = org[k] + std::max(0.0f, std::min(255.0f, b)) * sc[k];
                }
                o.ref[o.n++] = ref2[i];
            }
        }
    }
    return out;
}
static bool otraverse(const std::vector<ONode>& T, const Lbvh& B, V3 o, V3 d, float tmin, float tmax, bool any, bool sorted, WStats& st) {
    uint32_t stack[512]; int sp = 0, sp_max = 0;
    uint32_t cur = 0;
    float best = tmax; bool hit = false;
    const float inv[3] = {1.0f / (std::fabs(d.x) > 1e-20f ? d.x : 1e-20f), 1.0f / (std::fabs(d.y) > 1e-20f ? d.y : 1e-20f), 1.0f / (std::fabs(d.z) > 1e-20f ? d.z : 1e-20f)};
    const float oo[3] = {o.x, o.y, o.z};
    st.rays++;
    while (true) {
        sp_max = std::max(sp_max, sp);
        if (cur & 0x80000000u) {
            const int first = (int)((cur & 0x7fffffffu) >> 3), cnt = (int)(cur & 7u);
            for (int t = first; t < first + cnt; t++) {
                uint32_t flp = 0;
                if (t > first) memcpy(&flp, &B.pairs[(size_t)(t - 1) * 16 + 15], 4);
                if (!(flp & 1u)) st.tri_steps++;
                float th;
                if (tri_hit(&B.tris[(size_t)t * 16], o, d, tmin, best, th)) { best = th; hit = true; if (any) { st.depth_hist[std::min(sp_max, 63)]++; return true; } }
            }
            if (sp == 0) break;
            cur = stack[--sp];
            continue;
        }
        st.nodes++;
        const ONode& w = T[cur];
        float key[7]; uint32_t rf[7]; int n = 0;
        for (int i = 0; i < w.n; i++) {
            float t0 = tmin, t1 = best;
            for (int k = 0; k < 3; k++) { float a = (w.lo[i][k] - oo[k]) * inv[k], b = (w.hi[i][k] - oo[k]) * inv[k]; if (a > b) std::swap(a, b); t0 = std::max(t0, a); t1 = std::min(t1, b); }
            if (t0 <= t1 * 1.0000004f) { key[n] = t0; rf[n] = w.ref[i]; n++; }
        }
        if (n == 0) { if (sp == 0) break; cur = stack[--sp]; continue; }
        if (sorted) { for (int i = 1; i < n; i++) for (int j = i; j > 0 && key[j] < key[j - 1]; j--) { std::swap(key[j], key[j - 1]); std::swap(rf[j], rf[j - 1]); } }
        else { int m = 0; for (int i = 1; i < n; i++) if (key[i] < key[m]) m = i; std::swap(key[0], key[m]); std::swap(rf[0], rf[m]); }   // nearest first, the others as they come
        for (int i = n - 1; i >= 1; i--) stack[sp++] = rf[i];
        cur = rf[0];
    }
    st.depth_hist[std::min(sp_max, 63)]++;
    return hit;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: bvh_eval mesh.bin [rays]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[2];
    if (fread(hdr, 4, 2, f) != 2) return 2;
    const int nv = hdr[0], nt = hdr[1];
    std::vector<float> V((size_t)3 * nv);
    std::vector<uint32_t> I((size_t)3 * nt);
    if (fread(V.data(), 4, V.size(), f) != V.size() || fread(I.data(), 4, I.size(), f) != I.size()) return 2;
    fclose(f);
    const int n_rays = argc > 2 ? atoi(argv[2]) : 200000;
    std::vector<int32_t> mat(nt, 0);
    std::vector<uint8_t> emi(nt, 0);
    HostMesh m;
    m.vertices = V.data(); m.indices = I.data(); m.tri_material = mat.data(); m.tri_emitter = emi.data(); m.n_vertices = nv; m.n_triangles = nt;
    Lbvh B;
    auto t0 = std::chrono::steady_clock::now();
    build_lbvh(m, B);
    const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // area-weighted surface points
    std::vector<double> cdf(nt);
    double acc = 0;
    auto vert = [&](int t, int k) { const float* p = &V[3 * (size_t)I[3 * (size_t)t + k]]; return V3{p[0], p[1], p[2]}; };
    for (int t = 0; t < nt; t++) { V3 n = cross(vert(t, 1) - vert(t, 0), vert(t, 2) - vert(t, 0)); acc += 0.5 * std::sqrt((double)dot(n, n)); cdf[t] = acc; }
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    auto surface_point = [&](V3& P, V3& N) {
        const double r = U(rng) * acc;
        int t = (int)(std::lower_bound(cdf.begin(), cdf.end(), r) - cdf.begin());
        t = std::min(t, nt - 1);
        float a = (float)U(rng), b = (float)U(rng);
        if (a + b > 1) { a = 1 - a; b = 1 - b; }
        P = vert(t, 0) * (1 - a - b) + vert(t, 1) * a + vert(t, 2) * b;
        N = norm(cross(vert(t, 1) - vert(t, 0), vert(t, 2) - vert(t, 0)));
    };
    Stats sc, ss;
    g_visits.assign(B.nodes.size() / 16, 0);
    for (int i = 0; i < n_rays; i++) {
        V3 P, N;
        surface_point(P, N);
        if (U(rng) < 0.5) N = N * -1.0f;
        // cosine-weighted direction around N
        const float r1 = (float)U(rng), r2 = (float)U(rng), rr = std::sqrt(r1), ph = 6.2831853f * r2;
        V3 b1 = std::fabs(N.x) > std::fabs(N.z) ? norm(V3{-N.y, N.x, 0}) : norm(V3{0, -N.z, N.y});
        V3 b2 = cross(b1, N);
        V3 d = norm(b1 * (rr * std::cos(ph)) + b2 * (rr * std::sin(ph)) + N * std::sqrt(std::max(0.0f, 1 - r1)));
        traverse(B, P, d, 1e-3f, 1e16f, false, sc);
        V3 Q, M;
        surface_point(Q, M);
        V3 dv = Q - P;
        const float len = std::sqrt(dot(dv, dv));
        if (len > 1e-4f) traverse(B, P, dv * (1.0f / len), 1e-3f, len - 1e-3f, true, ss);
    }
    printf("nodes %zu  depth %d  build %.2f s\n", B.nodes.size() / 16, B.depth, build_s);
    printf("steps per ray (node visits + triangle steps, a fan pair = one step): closest %.2f  shadow %.2f\n", (sc.nodes + sc.tri_steps) / sc.rays, (ss.nodes + ss.tri_steps) / ss.rays);
    printf("closest: node visits %.2f  leaf visits %.2f  triangle tests %.2f  hit rate %.3f   with exact leaf boxes: leaf visits %.2f  tests <= %.2f\n", sc.nodes / sc.rays, sc.leaves / sc.rays, sc.tris / sc.rays, sc.hits / sc.rays, sc.leaves_exact / sc.rays, sc.tris_exact / sc.rays);
    printf("shadow : node visits %.2f  leaf visits %.2f  triangle tests %.2f  occluded %.3f   with exact leaf boxes: leaf visits %.2f  tests <= %.2f\n", ss.nodes / ss.rays, ss.leaves / ss.rays, ss.tris / ss.rays, ss.hits / ss.rays, ss.leaves_exact / ss.rays, ss.tris_exact / ss.rays);
    {   // how concentrated the visits are: share of all node visits that go to the K most visited nodes (what a K-node LDS copy would
        // serve), and to the first K nodes in memory order
        std::vector<uint64_t> v = g_visits;
        double total = 0;
        for (uint64_t x : v) total += (double)x;
        std::vector<uint64_t> sorted = v;
        std::sort(sorted.begin(), sorted.end(), [](uint64_t a, uint64_t b) { return a > b; });
        printf("visit share of the K hottest nodes / of nodes [0, K):");
        for (int K : {1, 5, 16, 32, 64, 128, 256, 1024, 4096}) {
            double hot = 0, first = 0;
            for (int i = 0; i < K && i < (int)v.size(); i++) { hot += (double)sorted[i]; first += (double)v[i]; }
            printf("  K=%d %.3f/%.3f", K, hot / total, first / total);
        }
        printf("\n");
    }
    if (getenv("BVH_EVAL_OPEN1")) {
        const std::vector<ONode> T = open_once(B);
        for (int sorted = 1; sorted >= 0; sorted--) {
            WStats wc, ws;
            std::mt19937_64 rng2(12345);
            std::uniform_real_distribution<double> U2(0.0, 1.0);
            auto sp2 = [&](V3& P, V3& N) {
                const double r = U2(rng2) * acc;
                int t = (int)(std::lower_bound(cdf.begin(), cdf.end(), r) - cdf.begin());
                t = std::min(t, nt - 1);
                float a = (float)U2(rng2), b = (float)U2(rng2);
                if (a + b > 1) { a = 1 - a; b = 1 - b; }
                P = vert(t, 0) * (1 - a - b) + vert(t, 1) * a + vert(t, 2) * b;
                N = norm(cross(vert(t, 1) - vert(t, 0), vert(t, 2) - vert(t, 0)));
            };
            for (int i = 0; i < n_rays; i++) {
                V3 P, N; sp2(P, N);
                if (U2(rng2) < 0.5) N = N * -1.0f;
                const float r1 = (float)U2(rng2), r2 = (float)U2(rng2), rr = std::sqrt(r1), ph = 6.2831853f * r2;
                V3 b1 = std::fabs(N.x) > std::fabs(N.z) ? norm(V3{-N.y, N.x, 0}) : norm(V3{0, -N.z, N.y});
                V3 b2 = cross(b1, N);
                V3 d = norm(b1 * (rr * std::cos(ph)) + b2 * (rr * std::sin(ph)) + N * std::sqrt(std::max(0.0f, 1 - r1)));
                otraverse(T, B, P, d, 1e-3f, 1e16f, false, sorted != 0, wc);
                V3 Q, M; sp2(Q, M);
                V3 dv = Q - P; const float len = std::sqrt(dot(dv, dv));
                if (len > 1e-4f) otraverse(T, B, P, dv * (1.0f / len), 1e-3f, len - 1e-3f, true, sorted != 0, ws);
            }
            double c9 = 0, c12 = 0, tot = 0;
            for (int i = 0; i < 64; i++) { tot += wc.depth_hist[i] + ws.depth_hist[i]; if (i <= 9) c9 += wc.depth_hist[i] + ws.depth_hist[i]; if (i <= 12) c12 += wc.depth_hist[i] + ws.depth_hist[i]; }
            printf("open once (<= 7 children, parent's grid), %s push: closest: node visits %.2f, steps %.2f; shadow: node visits %.2f, steps %.2f; rays whose stack stays <= 9 / <= 12 entries: %.4f / %.4f\n",
                   sorted ? "sorted" : "nearest-first + unsorted", wc.nodes / wc.rays, (wc.nodes + wc.tri_steps) / wc.rays, ws.nodes / ws.rays, (ws.nodes + ws.tri_steps) / ws.rays, c9 / tot, c12 / tot);
        }
    }
    if (const char* wenv = getenv("BVH_EVAL_WIDE")) {
        Lbvh bin;   // the binary tree the builder emits before the collapse (its triangle order is B's before the pairing: rebuild B's records for it)
        Builder bb(m, bin);
        bb.run();
        for (int W : {4, atoi(wenv)}) {
            WideTree T; T.bin = &bin.nodes; T.W = W;
            T.emit(0);
            // the pair flags of `bin`'s own triangle order
            Lbvh tmp; tmp.nodes.assign(16, 0.0f); tmp.tris = bin.tris; tmp.tri_orig = bin.tri_orig;
            // a flat list of leaves for make_fan_pairs: fake one 4-wide node per leaf range is overkill -- pair greedily inside the wide tree's leaves instead
            tmp.pairs.assign(tmp.tris.size(), 0.0f);
            for (const WNode& w : T.nodes) for (int i = 0; i < w.n; i++) if (w.ref[i] < 0) {
                const int first = ~w.ref[i];
                for (int t = first; t + 1 < first + w.cnt[i]; t++) {
                    const float* A = &tmp.tris[(size_t)t * 16]; const float* Bq = &tmp.tris[(size_t)(t + 1) * 16];
                    uint32_t prev = 0; if (t > first) memcpy(&prev, &tmp.pairs[(size_t)(t - 1) * 16 + 15], 4);
                    if (!(prev & 1u) && memcmp(A, Bq, 12) == 0 && memcmp(A + 8, Bq + 4, 12) == 0) { uint32_t one = 1u; memcpy(&tmp.pairs[(size_t)t * 16 + 15], &one, 4); }
                }
            }
            WStats wc, ws;
            std::mt19937_64 rng2(12345);
            std::uniform_real_distribution<double> U2(0.0, 1.0);
            auto sp2 = [&](V3& P, V3& N) {
                const double r = U2(rng2) * acc;
                int t = (int)(std::lower_bound(cdf.begin(), cdf.end(), r) - cdf.begin());
                t = std::min(t, nt - 1);
                float a = (float)U2(rng2), b = (float)U2(rng2);
                if (a + b > 1) { a = 1 - a; b = 1 - b; }
                P = vert(t, 0) * (1 - a - b) + vert(t, 1) * a + vert(t, 2) * b;
                N = norm(cross(vert(t, 1) - vert(t, 0), vert(t, 2) - vert(t, 0)));
            };
            for (int i = 0; i < n_rays; i++) {
                V3 P, N; sp2(P, N);
                if (U2(rng2) < 0.5) N = N * -1.0f;
                const float r1 = (float)U2(rng2), r2 = (float)U2(rng2), rr = std::sqrt(r1), ph = 6.2831853f * r2;
                V3 b1 = std::fabs(N.x) > std::fabs(N.z) ? norm(V3{-N.y, N.x, 0}) : norm(V3{0, -N.z, N.y});
                V3 b2 = cross(b1, N);
                V3 d = norm(b1 * (rr * std::cos(ph)) + b2 * (rr * std::sin(ph)) + N * std::sqrt(std::max(0.0f, 1 - r1)));
                wtraverse(T, tmp, P, d, 1e-3f, 1e16f, false, wc);
                V3 Q, M; sp2(Q, M);
                V3 dv = Q - P; const float len = std::sqrt(dot(dv, dv));
                if (len > 1e-4f) wtraverse(T, tmp, P, dv * (1.0f / len), 1e-3f, len - 1e-3f, true, ws);
            }
            double c9 = 0, c15 = 0, tot = 0;
            for (int i = 0; i < 64; i++) { tot += wc.depth_hist[i] + ws.depth_hist[i]; if (i <= 9) c9 += wc.depth_hist[i] + ws.depth_hist[i]; if (i <= 15) c15 += wc.depth_hist[i] + ws.depth_hist[i]; }
            printf("W = %d (exact child boxes): %zu nodes; closest: node visits %.2f, steps %.2f; shadow: node visits %.2f, steps %.2f; rays whose stack stays <= 9 / <= 15 entries: %.4f / %.4f\n",
                   W, T.nodes.size(), wc.nodes / wc.rays, (wc.nodes + wc.tri_steps) / wc.rays, ws.nodes / ws.rays, (ws.nodes + ws.tri_steps) / ws.rays, c9 / tot, c15 / tot);
        }
    }
    {   // traversal-stack depth (entries held): cumulative share of the rays whose maximum / of the node visits made at depth <= N
        double rays = 0, steps = 0;
        for (int i = 0; i < 64; i++) { rays += (double)g_depth_ray[i]; steps += (double)g_depth_step[i]; }
        printf("stack depth <= N: rays (their maximum) / node visits:");
        double cr = 0, cs = 0;
        for (int i = 0; i < 64; i++) {
            cr += (double)g_depth_ray[i]; cs += (double)g_depth_step[i];
            if (i == 3 || i == 5 || i == 7 || i == 9 || i == 11 || i == 13 || i == 15 || i == 19) printf("  N=%d %.4f/%.4f", i, cr / rays, cs / steps);
        }
        printf("\n");
    }
    return 0;
}

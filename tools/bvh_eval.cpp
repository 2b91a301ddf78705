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

// Host-side structural check of the BVH builders (csrc/lbvh.cpp): every triangle sits in exactly one leaf, every slot box
// contains what hangs below it, empty slots are point boxes at 1e30.  Usage: lbvh_check <n_triangles> <seed>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../spcbpt-optix7_amd/csrc/lbvh.cpp"

using namespace spc;

struct Check {
    const Lbvh& b;
    std::vector<int> seen;
    long long leaves = 0, nodes = 0;
    double sah = 0, slack = 0;
    int bad = 0;
    bool ok = true;
    explicit Check(const Lbvh& l) : b(l), seen(l.tri_orig.size(), 0) {}
    static int geti(const float* p) { int v; memcpy(&v, p, 4); return v; }
    // returns the box of everything below `node`; checks that every decoded (quantised) slot box contains its subtree
    void walk(int node, float lo[3], float hi[3], int depth) {
        nodes++;
        if (depth > 200) { ok = false; return; }
        uint32_t w[16];
        memcpy(w, &b.nodes[(size_t)node * 16], sizeof(w));
        float org[3]; memcpy(org, w, 12);
        float sc[3];
        for (int k = 0; k < 3; k++) { const uint32_t e = ((w[3] >> (8 * k)) & 0xffu) << 23; memcpy(&sc[k], &e, 4); }
        const uint32_t qlo[3] = {w[4], w[5], w[6]}, qhi[3] = {w[7], w[8], w[9]};
        const uint32_t refs[4] = {w[10], w[11], w[12], w[13]};
        for (int k = 0; k < 3; k++) { lo[k] = 1e30f; hi[k] = -1e30f; }
        bool seen_empty = false;
        for (int i = 0; i < 4; i++) {
            const uint32_t ref = refs[i];
            if (ref == 0x80000000u) {
                seen_empty = true;
                for (int k = 0; k < 3; k++) if (((qlo[k] >> (8 * i)) & 0xffu) != 255u || ((qhi[k] >> (8 * i)) & 0xffu) != 0u) ok = false;
                continue;
            }
            if (seen_empty) ok = false;  // used slots come first
            float slo[3], shi[3];
            for (int k = 0; k < 3; k++) {
                slo[k] = org[k] + (float)((qlo[k] >> (8 * i)) & 0xffu) * sc[k];
                shi[k] = org[k] + (float)((qhi[k] >> (8 * i)) & 0xffu) * sc[k];
            }
            float clo[3], chi[3];
            int cnt = 1;
            if (ref & 0x80000000u) {
                const int first = (int)((ref & 0x7fffffffu) >> 3);
                cnt = (int)(ref & 7u);
                if (cnt < 1 || cnt > 4) ok = false;
                for (int k = 0; k < 3; k++) { clo[k] = 1e30f; chi[k] = -1e30f; }
                for (int t = first; t < first + cnt; t++) {
                    if (t < 0 || t >= (int)seen.size()) { ok = false; continue; }
                    seen[t]++;
                    const float* tr = &b.tris[(size_t)t * 16];
                    for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], tr[4 * v + k]); chi[k] = std::max(chi[k], tr[4 * v + k]); }
                }
                leaves++;
            } else {
                if ((int)ref <= node || (size_t)ref * 16 >= b.nodes.size()) { ok = false; continue; }
                walk((int)ref, clo, chi, depth + 1);
            }
            for (int k = 0; k < 3; k++) {
                if (clo[k] < slo[k] || chi[k] > shi[k]) { ok = false; if (bad++ < 5) printf("node %d slot %d axis %d: [%g,%g] not in [%g,%g]\n", node, i, k, clo[k], chi[k], slo[k], shi[k]); }
                lo[k] = std::min(lo[k], clo[k]); hi[k] = std::max(hi[k], chi[k]);
                slack += (shi[k] - slo[k]) - (chi[k] - clo[k]);
            }
            const float dx = shi[0] - slo[0], dy = shi[1] - slo[1], dz = shi[2] - slo[2];
            sah += (double)(dx * dy + dy * dz + dz * dx) * cnt;
        }
    }
};

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 20000;
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1u;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::vector<float> V; std::vector<uint32_t> I; std::vector<int32_t> M; std::vector<uint8_t> E;
    int n_quads = 0;
    for (int t = 0; t < n; t++) {
        // clustered small triangles plus a few large ones and exact duplicates (coincident centroids)
        float c[3] = {U(rng) * 10, U(rng) * 3, U(rng) * 10};
        float s = (t % 97 == 0) ? 3.0f : 0.05f;
        if (t % 50 == 1 && t > 0) { for (int k = 0; k < 9; k++) V.push_back(V[V.size() - 9]); }
        else if (t % 5 == 2 && t > 0) {   // the second half of a quad: (P0, P2 of the triangle before, a new corner) -- a fan pair (lbvh.h)
            const size_t a = V.size() - 9;
            for (int k = 0; k < 3; k++) V.push_back(V[a + k]);
            for (int k = 0; k < 3; k++) V.push_back(V[a + 6 + k]);
            for (int k = 0; k < 3; k++) V.push_back(V[a + k] + V[a + 6 + k] - V[a + 3 + k]);
            n_quads++;
        }
        else for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) V.push_back(c[k] + s * (U(rng) - 0.5f));
        for (int v = 0; v < 3; v++) I.push_back(3 * t + v);
        M.push_back(0); E.push_back((uint8_t)(t % 5 == 2 || t % 5 == 1 ? (t / 5) % 2 : 0));
    }
    HostMesh m; m.vertices = V.data(); m.indices = I.data(); m.tri_material = M.data(); m.tri_emitter = E.data();
    m.n_vertices = 3 * n; m.n_triangles = n;
    Lbvh out;
    auto t0 = std::chrono::steady_clock::now();
    build_lbvh(m, out);
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    Check c(out);
    float lo[3], hi[3];
    c.walk(0, lo, hi, 1);
    for (int s : c.seen) if (s != 1) c.ok = false;
    std::vector<int> perm(out.tri_orig.begin(), out.tri_orig.end());
    std::sort(perm.begin(), perm.end());
    for (int i = 0; i < n; i++) if (perm[i] != i) c.ok = false;
    // the triangle records still belong to the triangles tri_orig names (make_fan_pairs reorders inside leaves) ...
    for (int i = 0; i < n; i++) {
        const float* tr = &out.tris[(size_t)i * 16];
        const float* src = &V[(size_t)9 * out.tri_orig[i]];
        for (int v = 0; v < 3; v++) if (memcmp(tr + 4 * v, src + 3 * v, 12) != 0) { c.ok = false; printf("record %d is not triangle %d\n", i, out.tri_orig[i]); break; }
    }
    // ... and the pair slots (lbvh.h: Lbvh::pairs): a slot flagged "pair" holds (A.P0, A.P1 - A.P0, A.P2 - A.P0, B.P2 - A.P0) of triangles i, i + 1 of ONE
    // leaf with B = (A.P0, A.P2, B.P2) bit for bit and both culling flags; every other slot holds its own triangle
    {
        std::vector<int> leaf_of((size_t)n, -1);
        int leaf_id = 0;
        for (size_t nd = 0; nd < out.nodes.size() / 16; nd++)
            for (int k = 0; k < 4; k++) {
                uint32_t ref; memcpy(&ref, &out.nodes[nd * 16 + 10 + k], 4);
                if (!(ref & 0x80000000u)) continue;
                for (uint32_t t = (ref & 0x7fffffffu) >> 3, e = t + (ref & 7u); t < e && t < (uint32_t)n; t++) leaf_of[t] = leaf_id;
                leaf_id++;
            }
        int pairs = 0;
        if ((int)out.pairs.size() != 16 * n) { c.ok = false; printf("pair slots: %zu floats for %d triangles\n", out.pairs.size(), n); }
        for (int i = 0; c.ok && i < n; i++) {
            const float* q = &out.pairs[(size_t)i * 16];
            const float* a = &out.tris[(size_t)i * 16];
            uint32_t fl, ma; memcpy(&fl, q + 15, 4); memcpy(&ma, a + 15, 4);
            // (the slot holds the corner P0 and the EDGES P1 - P0, P2 - P0 as FP32 differences)
            float e1[3], e2[3];
            for (int k = 0; k < 3; k++) { e1[k] = a[4 + k] - a[k]; e2[k] = a[8 + k] - a[k]; }
            bool good = memcmp(q, a, 12) == 0 && memcmp(q + 4, e1, 12) == 0 && memcmp(q + 8, e2, 12) == 0 && ((fl ^ ma) & 0x80000000u) == 0;
            if (fl & 1u) {
                pairs++;
                if (i + 1 >= n || leaf_of[i + 1] != leaf_of[i] || (i > 0 && (([&] { uint32_t p; memcpy(&p, &out.pairs[(size_t)(i - 1) * 16 + 15], 4); return p & 1u; })()))) good = false;
                else {
                    const float* b = &out.tris[(size_t)(i + 1) * 16];
                    uint32_t mb; memcpy(&mb, b + 15, 4);
                    float e3[3];
                    for (int k = 0; k < 3; k++) e3[k] = b[8 + k] - a[k];
                    good = good && memcmp(b, a, 12) == 0 && memcmp(b + 4, a + 8, 12) == 0 && memcmp(q + 12, e3, 12) == 0 && ((fl >> 30) & 1u) == (mb >> 31);
                }
            } else good = good && (fl & 0x7fffffffu) == 0;
            if (!good) { c.ok = false; printf("pair slot %d is wrong (flags %08x)\n", i, fl); }
        }
        if (out.n_paired != 2 * pairs) { c.ok = false; printf("n_paired %d, slots flagged %d\n", out.n_paired, pairs); }
        // the halves of a quad share a bounding box, so the builder rarely separates them: most of the quads made above end up paired
        if (n >= 700 && pairs * 2 < n_quads) { c.ok = false; printf("only %d of %d quads paired\n", pairs, n_quads); }
    }
    // every node is reachable exactly once (the walk counts them), and the first HOT_NODES records are the hottest-first crown of the
    // tree (lbvh.cpp: hot_nodes_first): node 0 is the root, the parent of a hot node is hot and comes before it, and no node outside
    // the crown has a larger box than the crown's last (the crown is the top of the area order, ties aside)
    const int n_nodes = (int)(out.nodes.size() / 16);
    if (c.nodes != n_nodes) { c.ok = false; printf("walk reached %lld of %d nodes\n", c.nodes, n_nodes); }
    {
        std::vector<int> parent((size_t)n_nodes, -1);
        std::vector<float> area((size_t)n_nodes, 0.0f);
        for (int node = 0; node < n_nodes; node++) {
            uint32_t w[16];
            memcpy(w, &out.nodes[(size_t)node * 16], sizeof(w));
            float org[3]; memcpy(org, w, 12);
            float sc[3];
            for (int k = 0; k < 3; k++) { const uint32_t e = ((w[3] >> (8 * k)) & 0xffu) << 23; memcpy(&sc[k], &e, 4); }
            for (int i = 0; i < 4; i++) {
                const uint32_t ref = w[10 + i];
                if (ref & 0x80000000u) continue;
                float e[3];
                for (int k = 0; k < 3; k++) e[k] = (float)((w[7 + k] >> (8 * i)) & 0xffu) * sc[k] - (float)((w[4 + k] >> (8 * i)) & 0xffu) * sc[k];
                parent[ref] = node; area[ref] = e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
            }
        }
        // (round 6, advisor) areas are read from the 8-bit boxes, which are rounded outward on the PARENT's grid: a child's stored area
        // may exceed by a grid cell per axis (~1 %) what its parent measures in the grandparent's grid, so the largest-first pop order
        // of the builder is monotone only up to that slack.  The structural property (a hot node's parent is hot and precedes it) is exact.
        const float slack = 1.02f;
        const int hot = std::min(n_nodes, (int)HOT_NODES);
        for (int i = 1; i < hot; i++) {
            if (parent[i] < 0 || parent[i] >= i) { c.ok = false; printf("hot node %d: parent %d\n", i, parent[i]); }
            if (i > 1 && area[i] > area[i - 1] * slack) { c.ok = false; printf("hot node %d: area %g after %g\n", i, area[i], area[i - 1]); }
        }
        for (int i = hot; i < n_nodes && hot > 1; i++)
            if (parent[i] >= 0 && parent[i] < hot && area[i] > area[hot - 1] * slack) { c.ok = false; printf("node %d (area %g) should be in the crown (last %g)\n", i, area[i], area[hot - 1]); break; }
    }
    printf("%s n=%d nodes=%lld leaves=%lld depth=%d sah=%.1f build=%.3fs\n", c.ok ? "OK" : "FAIL", n, c.nodes, c.leaves, out.depth, c.sah, sec);
    return c.ok ? 0 : 1;
}

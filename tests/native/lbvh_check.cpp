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
    for (int t = 0; t < n; t++) {
        // clustered small triangles plus a few large ones and exact duplicates (coincident centroids)
        float c[3] = {U(rng) * 10, U(rng) * 3, U(rng) * 10};
        float s = (t % 97 == 0) ? 3.0f : 0.05f;
        if (t % 50 == 1 && t > 0) { for (int k = 0; k < 9; k++) V.push_back(V[V.size() - 9]); }
        else for (int v = 0; v < 3; v++) for (int k = 0; k < 3; k++) V.push_back(c[k] + s * (U(rng) - 0.5f));
        for (int v = 0; v < 3; v++) I.push_back(3 * t + v);
        M.push_back(0); E.push_back(0);
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

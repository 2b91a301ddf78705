// Host-side LBVH construction (replaces OptiX's GAS/IAS build, sutil/Scene.cpp:943-1338).
// Morton-ordered primitives (63-bit codes) -> Karras 2012 radix-tree topology ->
// collapse of every subtree covering <= LEAF_MAX primitives into one leaf ->
// depth-first node emission with both child boxes stored in the parent (layout.h).
#include "lbvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <numeric>
#include <string>

namespace spc {

static inline uint64_t expand21(uint64_t v) {  // spread the low 21 bits to every third bit
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

struct Builder {
    const HostMesh& m;
    int n;
    std::vector<uint64_t> keys;
    std::vector<int> order;       // sorted position -> original triangle
    std::vector<int> left, right; // Karras internal nodes: child ids (>= 0 internal, < 0 -> leaf ~id = sorted position)
    std::vector<int> rlo, rhi;    // covered range of each internal node
    std::vector<float> tlo, thi;  // per sorted triangle bounds
    Lbvh& out;
    int max_depth = 0;

    Builder(const HostMesh& mesh, Lbvh& o) : m(mesh), n(mesh.n_triangles), out(o) {}

    inline int delta(int i, int j) const {
        if (j < 0 || j >= n) return -1;
        uint64_t a = keys[i], b = keys[j];
        if (a != b) return __builtin_clzll(a ^ b);
        return 64 + __builtin_clz((unsigned)i ^ (unsigned)j);
    }

    void topology() {
        left.assign(n - 1, 0); right.assign(n - 1, 0); rlo.assign(n - 1, 0); rhi.assign(n - 1, 0);
        for (int i = 0; i < n - 1; i++) {
            int d = (delta(i, i + 1) - delta(i, i - 1)) >= 0 ? 1 : -1;
            int dmin = delta(i, i - d);
            int lmax = 2;
            while (delta(i, i + lmax * d) > dmin) lmax *= 2;
            int l = 0;
            for (int t = lmax / 2; t >= 1; t /= 2)
                if (delta(i, i + (l + t) * d) > dmin) l += t;
            int j = i + l * d;
            int dnode = delta(i, j);
            int s = 0;
            for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
                if (delta(i, i + (s + t) * d) > dnode) s += t;
                if (t == 1) break;
            }
            int gamma = i + s * d + std::min(d, 0);
            int lo = std::min(i, j), hi = std::max(i, j);
            left[i] = (lo == gamma) ? ~gamma : gamma;
            right[i] = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
            rlo[i] = lo; rhi[i] = hi;
        }
    }

    struct Box { float lo[3], hi[3]; };
    Box range_box(int a, int b) const {
        Box r;
        for (int k = 0; k < 3; k++) { r.lo[k] = 1e30f; r.hi[k] = -1e30f; }
        for (int i = a; i <= b; i++)
            for (int k = 0; k < 3; k++) {
                r.lo[k] = std::min(r.lo[k], tlo[3 * i + k]);
                r.hi[k] = std::max(r.hi[k], thi[3 * i + k]);
            }
        return r;
    }
    static Box merge(const Box& a, const Box& b) {
        Box r;
        for (int k = 0; k < 3; k++) { r.lo[k] = std::min(a.lo[k], b.lo[k]); r.hi[k] = std::max(a.hi[k], b.hi[k]); }
        return r;
    }
    // child id (Karras) -> covered range
    void child_range(int c, int& a, int& b) const {
        if (c < 0) { a = b = ~c; } else { a = rlo[c]; b = rhi[c]; }
    }
    // Emits the subtree rooted at Karras node `k` (which covers > LEAF_MAX primitives) as output node; returns its box.
    Box emit(int k, int out_index, int depth) {
        max_depth = std::max(max_depth, depth);
        int cid[2] = {left[k], right[k]};
        Box cb[2];
        int cref[2], ccount[2];
        for (int s = 0; s < 2; s++) {
            int a, b;
            child_range(cid[s], a, b);
            if (b - a + 1 <= LEAF_MAX) {
                cb[s] = range_box(a, b);
                cref[s] = ~a;
                ccount[s] = b - a + 1;
            } else {
                int idx = (int)(out.nodes.size() / 16);
                out.nodes.resize(out.nodes.size() + 16);
                cb[s] = emit(cid[s], idx, depth + 1);
                cref[s] = idx;
                ccount[s] = 0;
            }
        }
        float* q = &out.nodes[(size_t)out_index * 16];
        auto put_i = [](float* p, int v) { memcpy(p, &v, 4); };
        q[0] = cb[0].lo[0]; q[1] = cb[0].lo[1]; q[2] = cb[0].lo[2]; put_i(q + 3, cref[0]);
        q[4] = cb[0].hi[0]; q[5] = cb[0].hi[1]; q[6] = cb[0].hi[2]; put_i(q + 7, cref[1]);
        q[8] = cb[1].lo[0]; q[9] = cb[1].lo[1]; q[10] = cb[1].lo[2]; put_i(q + 11, ccount[0]);
        q[12] = cb[1].hi[0]; q[13] = cb[1].hi[1]; q[14] = cb[1].hi[2]; put_i(q + 15, ccount[1]);
        return merge(cb[0], cb[1]);
    }

    // ---- split selection by SAH along the Morton order -------------------------------------------------------------
    // The primitives stay in Morton-curve order (that is what makes this an LBVH: one sort, contiguous ranges), but every
    // node cuts its range where surface-area cost is lowest instead of at the highest differing Morton bit.  Two linear
    // sweeps per node (suffix boxes, then prefix boxes).
    static float half_area(const Box& b) {
        float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    std::vector<Box> suffix;
    int sah_split(int a, int b) {  // returns s in (a, b]: left = [a, s), right = [s, b]
        const int n2 = b - a + 1;
        if ((int)suffix.size() < n2) suffix.resize(n2);
        Box acc = range_box(b, b);
        suffix[n2 - 1] = acc;
        for (int i = b - 1; i > a; i--) {
            Box t = range_box(i, i);
            acc = merge(acc, t);
            suffix[i - a] = acc;
        }
        Box left = range_box(a, a);
        float best = 3.4e38f;
        int best_s = a + n2 / 2;
        for (int s = a + 1; s <= b; s++) {
            // left = [a, s), right = [s, b]
            const float cost = half_area(left) * (float)(s - a) + half_area(suffix[s - a]) * (float)(b - s + 1);
            if (cost < best) { best = cost; best_s = s; }
            left = merge(left, range_box(s, s));
        }
        return best_s;
    }
    Box emit_sah(int a, int b, int out_index, int depth) {
        max_depth = std::max(max_depth, depth);
        const int s = sah_split(a, b);
        const int ra[2] = {a, s}, rb[2] = {s - 1, b};
        Box cb[2];
        int cref[2], ccount[2];
        for (int k = 0; k < 2; k++) {
            if (rb[k] - ra[k] + 1 <= LEAF_MAX) {
                cb[k] = range_box(ra[k], rb[k]);
                cref[k] = ~ra[k];
                ccount[k] = rb[k] - ra[k] + 1;
            } else {
                int idx = (int)(out.nodes.size() / 16);
                out.nodes.resize(out.nodes.size() + 16);
                cb[k] = emit_sah(ra[k], rb[k], idx, depth + 1);
                cref[k] = idx;
                ccount[k] = 0;
            }
        }
        float* q = &out.nodes[(size_t)out_index * 16];
        auto put_i = [](float* p, int v) { memcpy(p, &v, 4); };
        q[0] = cb[0].lo[0]; q[1] = cb[0].lo[1]; q[2] = cb[0].lo[2]; put_i(q + 3, cref[0]);
        q[4] = cb[0].hi[0]; q[5] = cb[0].hi[1]; q[6] = cb[0].hi[2]; put_i(q + 7, cref[1]);
        q[8] = cb[1].lo[0]; q[9] = cb[1].lo[1]; q[10] = cb[1].lo[2]; put_i(q + 11, ccount[0]);
        q[12] = cb[1].hi[0]; q[13] = cb[1].hi[1]; q[14] = cb[1].hi[2]; put_i(q + 15, ccount[1]);
        return merge(cb[0], cb[1]);
    }

    // ---- full binned-SAH top-down build ------------------------------------------------------------------------------
    // OptiX builds its acceleration structure with a surface-area heuristic; a Morton-order LBVH costs 1.5-2x the node
    // visits for the incoherent shadow rays of the connection stage.  The build runs once per scene on the host (outside
    // any timed region): 32 bins per axis, all three axes evaluated, median split as the fallback for degenerate ranges.
    struct SNode { int a, b, left, right; };
    std::vector<SNode> snodes;
    std::vector<float> blo, bhi;  // per original triangle
    std::vector<int> idx;         // permutation being partitioned
    static constexpr int NB = 32;
    float node_cost = 1.0f;  // SPCBPT_BVH_NODE_COST (0 = always make the largest leaf that fits); measured on the bench scene: 0.3 -> 13.57 ms, 1 -> 13.04, 2 -> 13.22, leaf-always -> 13.42
    int sah_build(const std::vector<float>& cen, int a, int b, int depth) {
        const int id = (int)snodes.size();
        snodes.push_back({a, b, -1, -1});
        const int cnt = b - a + 1;
        if (cnt <= 1 || (cnt <= LEAF_MAX && node_cost <= 0.0f)) return id;
        float clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
        Box whole; for (int d = 0; d < 3; d++) { whole.lo[d] = 1e30f; whole.hi[d] = -1e30f; }
        for (int i = a; i <= b; i++) {
            const int t = idx[i];
            const float* c = &cen[3 * (size_t)t];
            for (int k = 0; k < 3; k++) {
                clo[k] = std::min(clo[k], c[k]); chi[k] = std::max(chi[k], c[k]);
                whole.lo[k] = std::min(whole.lo[k], blo[3 * (size_t)t + k]); whole.hi[k] = std::max(whole.hi[k], bhi[3 * (size_t)t + k]);
            }
        }
        struct Bin { float lo[3], hi[3]; int n; };
        Bin bins[3][NB];
        float scale[3];
        for (int k = 0; k < 3; k++) {
            scale[k] = chi[k] > clo[k] ? (float)NB / (chi[k] - clo[k]) : 0.0f;
            for (int j = 0; j < NB; j++) {
                bins[k][j].n = 0;
                for (int d = 0; d < 3; d++) { bins[k][j].lo[d] = 1e30f; bins[k][j].hi[d] = -1e30f; }
            }
        }
        for (int i = a; i <= b; i++) {
            const int t = idx[i];
            const float* c = &cen[3 * (size_t)t];
            for (int k = 0; k < 3; k++) {
                if (scale[k] == 0.0f) continue;
                const int j = std::min(NB - 1, std::max(0, (int)((c[k] - clo[k]) * scale[k])));
                Bin& B = bins[k][j];
                B.n++;
                for (int d = 0; d < 3; d++) { B.lo[d] = std::min(B.lo[d], blo[3 * (size_t)t + d]); B.hi[d] = std::max(B.hi[d], bhi[3 * (size_t)t + d]); }
            }
        }
        float best = 3.4e38f;
        int best_axis = -1, best_bin = -1;
        for (int k = 0; k < 3; k++) {
            if (scale[k] == 0.0f) continue;
            float rarea[NB];
            int rn[NB];
            Box acc; for (int d = 0; d < 3; d++) { acc.lo[d] = 1e30f; acc.hi[d] = -1e30f; }
            int n_acc = 0;
            for (int j = NB - 1; j > 0; j--) {
                if (bins[k][j].n) { for (int d = 0; d < 3; d++) { acc.lo[d] = std::min(acc.lo[d], bins[k][j].lo[d]); acc.hi[d] = std::max(acc.hi[d], bins[k][j].hi[d]); } }
                n_acc += bins[k][j].n;
                rarea[j] = n_acc ? half_area(acc) : 0.0f;
                rn[j] = n_acc;
            }
            for (int d = 0; d < 3; d++) { acc.lo[d] = 1e30f; acc.hi[d] = -1e30f; }
            n_acc = 0;
            for (int j = 0; j < NB - 1; j++) {  // left = bins [0, j], right = bins [j + 1, NB)
                if (bins[k][j].n) { for (int d = 0; d < 3; d++) { acc.lo[d] = std::min(acc.lo[d], bins[k][j].lo[d]); acc.hi[d] = std::max(acc.hi[d], bins[k][j].hi[d]); } }
                n_acc += bins[k][j].n;
                if (n_acc == 0 || rn[j + 1] == 0) continue;
                const float cost = half_area(acc) * (float)n_acc + rarea[j + 1] * (float)rn[j + 1];
                if (cost < best) { best = cost; best_axis = k; best_bin = j; }
            }
        }
        // a range that fits a leaf is split further only where the surface-area heuristic says the extra node pays for itself
        // (cost of a triangle test = 1, cost of the extra node slot = node_cost)
        if (cnt <= LEAF_MAX && (best_axis < 0 || best + node_cost * half_area(whole) >= (float)cnt * half_area(whole))) return id;
        int mid;
        if (best_axis >= 0) {
            const int k = best_axis;
            const float lo = clo[k], sc = scale[k];
            int* first = idx.data() + a;
            int* last = idx.data() + b + 1;
            int* m2 = std::partition(first, last, [&](int t) {
                return std::min(NB - 1, std::max(0, (int)((cen[3 * (size_t)t + k] - lo) * sc))) <= best_bin;
            });
            mid = (int)(m2 - idx.data());
        } else {
            mid = a + cnt / 2;  // all centroids coincide: any balanced cut
        }
        if (mid <= a || mid > b) {  // cannot happen with consistent binning; keep the tree valid regardless
            int k = 0;
            for (int d = 1; d < 3; d++) if (chi[d] - clo[d] > chi[k] - clo[k]) k = d;
            mid = a + cnt / 2;
            std::nth_element(idx.begin() + a, idx.begin() + mid, idx.begin() + b + 1,
                             [&](int x, int y) { return cen[3 * (size_t)x + k] < cen[3 * (size_t)y + k]; });
        }
        const int l = sah_build(cen, a, mid - 1, depth + 1);
        const int r = sah_build(cen, mid, b, depth + 1);
        snodes[id].left = l; snodes[id].right = r;
        return id;
    }
    Box emit_tree(int sn, int out_index, int depth) {
        max_depth = std::max(max_depth, depth);
        const int cid[2] = {snodes[sn].left, snodes[sn].right};
        Box cb[2];
        int cref[2], ccount[2];
        for (int k = 0; k < 2; k++) {
            const SNode c = snodes[cid[k]];
            if (c.left < 0) {
                cb[k] = range_box(c.a, c.b);
                cref[k] = ~c.a;
                ccount[k] = c.b - c.a + 1;
            } else {
                int idx2 = (int)(out.nodes.size() / 16);
                out.nodes.resize(out.nodes.size() + 16);
                cb[k] = emit_tree(cid[k], idx2, depth + 1);
                cref[k] = idx2;
                ccount[k] = 0;
            }
        }
        float* q = &out.nodes[(size_t)out_index * 16];
        auto put_i = [](float* p, int v) { memcpy(p, &v, 4); };
        q[0] = cb[0].lo[0]; q[1] = cb[0].lo[1]; q[2] = cb[0].lo[2]; put_i(q + 3, cref[0]);
        q[4] = cb[0].hi[0]; q[5] = cb[0].hi[1]; q[6] = cb[0].hi[2]; put_i(q + 7, cref[1]);
        q[8] = cb[1].lo[0]; q[9] = cb[1].lo[1]; q[10] = cb[1].lo[2]; put_i(q + 11, ccount[0]);
        q[12] = cb[1].hi[0]; q[13] = cb[1].hi[1]; q[14] = cb[1].hi[2]; put_i(q + 15, ccount[1]);
        return merge(cb[0], cb[1]);
    }

    void run() {
        const float* P = m.vertices;
        // centroid bounds
        float clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
        std::vector<float> cen((size_t)3 * n);
        blo.resize((size_t)3 * n); bhi.resize((size_t)3 * n);
        for (int t = 0; t < n; t++) {
            const float* a = P + 3 * (size_t)m.indices[3 * t];
            const float* b = P + 3 * (size_t)m.indices[3 * t + 1];
            const float* c = P + 3 * (size_t)m.indices[3 * t + 2];
            for (int k = 0; k < 3; k++) {
                float lo = std::min(a[k], std::min(b[k], c[k])), hi = std::max(a[k], std::max(b[k], c[k]));
                float ce = 0.5f * (lo + hi);
                blo[3 * (size_t)t + k] = lo; bhi[3 * (size_t)t + k] = hi;
                cen[3 * (size_t)t + k] = ce;
                clo[k] = std::min(clo[k], ce); chi[k] = std::max(chi[k], ce);
            }
        }
        keys.resize(n);
        std::vector<std::pair<uint64_t, int>> kv(n);
        for (int t = 0; t < n; t++) {
            uint64_t code = 0;
            for (int k = 0; k < 3; k++) {
                double ext = (double)chi[k] - (double)clo[k];
                double f = ext > 0 ? ((double)cen[3 * (size_t)t + k] - clo[k]) / ext : 0.0;
                uint64_t q = (uint64_t)std::min(2097151.0, std::max(0.0, f * 2097152.0));
                code |= expand21(q) << (2 - k);
            }
            kv[t] = {code, t};
        }
        std::sort(kv.begin(), kv.end());
        order.resize(n);
        for (int i = 0; i < n; i++) { keys[i] = kv[i].first; order[i] = kv[i].second; }
        // SPCBPT_BVH selects the builder: "sah" = binned SAH top-down, "lbvh" = Karras radix tree over the Morton order,
        // "lbvh-sah" = SAH cuts along the Morton order (experiment)
        const char* mode_env = getenv("SPCBPT_BVH");
        const std::string mode = mode_env ? mode_env : "sah";
        if (mode == "sah" && n > LEAF_MAX) {
            idx = order;  // start from the Morton order: equal-cost ties keep spatial locality
            if (const char* nc = getenv("SPCBPT_BVH_NODE_COST")) node_cost = (float)atof(nc);
            snodes.reserve((size_t)n);
            sah_build(cen, 0, n - 1, 1);
            order = idx;
        }
        tlo.resize((size_t)3 * n); thi.resize((size_t)3 * n);
        out.tris.resize((size_t)16 * n);
        out.tri_orig.resize(n);
        for (int i = 0; i < n; i++) {
            int t = order[i];
            out.tri_orig[i] = t;
            const uint32_t* ix = m.indices + 3 * (size_t)t;
            float* q = &out.tris[(size_t)16 * i];
            float uv[3][2];
            for (int v = 0; v < 3; v++) {
                const float* p = P + 3 * (size_t)ix[v];
                q[4 * v + 0] = p[0]; q[4 * v + 1] = p[1]; q[4 * v + 2] = p[2];
                uv[v][0] = m.texcoords ? m.texcoords[2 * (size_t)ix[v]] : 0.0f;
                uv[v][1] = m.texcoords ? m.texcoords[2 * (size_t)ix[v] + 1] : 0.0f;
            }
            q[3] = uv[0][0]; q[7] = uv[0][1]; q[11] = uv[1][0];
            q[12] = uv[1][1]; q[13] = uv[2][0]; q[14] = uv[2][1];
            uint32_t meta = (uint32_t)m.tri_material[t] | (m.tri_emitter[t] ? 0x80000000u : 0u);
            memcpy(q + 15, &meta, 4);
            for (int k = 0; k < 3; k++) {
                tlo[3 * (size_t)i + k] = std::min(q[k], std::min(q[4 + k], q[8 + k]));
                thi[3 * (size_t)i + k] = std::max(q[k], std::max(q[4 + k], q[8 + k]));
            }
        }
        out.nodes.clear();
        out.nodes.resize(16);
        if (n <= LEAF_MAX) {
            // degenerate tiny scene: root with one real leaf and one empty child
            Box b = range_box(0, n - 1);
            float* q = &out.nodes[0];
            int c0 = ~0, cnt0 = n, c1 = ~0, cnt1 = 0;
            q[0] = b.lo[0]; q[1] = b.lo[1]; q[2] = b.lo[2]; memcpy(q + 3, &c0, 4);
            q[4] = b.hi[0]; q[5] = b.hi[1]; q[6] = b.hi[2]; memcpy(q + 7, &c1, 4);
            q[8] = 1e30f; q[9] = 1e30f; q[10] = 1e30f; memcpy(q + 11, &cnt0, 4);
            q[12] = -1e30f; q[13] = -1e30f; q[14] = -1e30f; memcpy(q + 15, &cnt1, 4);
            out.depth = 1;
            return;
        }
        // "lbvh-sah" cuts each Morton range at the lowest surface-area cost; measured on the bedroom scene it visits 4 % MORE
        // nodes than the radix tree (358 vs 344 per eye path), so it stays an experiment switch.
        if (mode == "sah") {
            emit_tree(0, 0, 1);
        } else if (mode == "lbvh-sah") {
            emit_sah(0, n - 1, 0, 1);
        } else {
            topology();
            emit(0, 0, 1);
        }
        out.depth = max_depth;
    }
};

// ---- 2-wide -> 4-wide collapse + quantisation -------------------------------------------------------------------------
// A traversal step is one dependent memory round trip whatever the node holds, so the binary tree is folded into 4-wide
// nodes: the child with the largest surface area is replaced by its own two children until four slots are used.  The node is
// then written in the 64-B quantised form described in layout.h.
namespace {
struct Slot { float lo[3], hi[3]; int ref, count; };
struct Collapser {
    const std::vector<float>& bin;
    std::vector<float>& out;
    int max_depth = 0;
    Collapser(const std::vector<float>& b, std::vector<float>& o) : bin(b), out(o) {}
    static int geti(const float* p) { int v; memcpy(&v, p, 4); return v; }
    void children_of(int bnode, Slot& a, Slot& b) const {
        const float* q = &bin[(size_t)bnode * 16];
        for (int k = 0; k < 3; k++) { a.lo[k] = q[k]; a.hi[k] = q[4 + k]; b.lo[k] = q[8 + k]; b.hi[k] = q[12 + k]; }
        a.ref = geti(q + 3); b.ref = geti(q + 7); a.count = geti(q + 11); b.count = geti(q + 15);
    }
    static float area(const Slot& s) {
        float dx = s.hi[0] - s.lo[0], dy = s.hi[1] - s.lo[1], dz = s.hi[2] - s.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    void emit(int bnode, int out_index, int depth) {
        max_depth = std::max(max_depth, depth);
        Slot s[4];
        int n = 2;
        children_of(bnode, s[0], s[1]);
        if (s[1].count == 0 && s[1].ref < 0) n = 1;  // degenerate root of a tiny scene: second child is an empty leaf
        while (n < 4) {
            int best = -1;
            float best_a = -1.0f;
            for (int i = 0; i < n; i++)
                if (s[i].ref >= 0 && area(s[i]) > best_a) { best_a = area(s[i]); best = i; }
            if (best < 0) break;
            Slot a, b;
            children_of(s[best].ref, a, b);
            s[best] = a;
            s[n++] = b;
        }
        for (int i = 0; i < n; i++)
            if (s[i].ref >= 0) {
                const int idx = (int)(out.size() / 16);
                out.resize(out.size() + 16);
                emit(s[i].ref, idx, depth + 1);
                s[i].ref = idx;
                s[i].count = 0;
            }
        // quantise the child boxes against the node's own box: plane = org + q * 2^e, lo rounded down, hi rounded up
        float org[3], ext[3];
        uint32_t eb[3];
        for (int k = 0; k < 3; k++) {
            float lo = 1e30f, hi = -1e30f;
            for (int i = 0; i < n; i++) { lo = std::min(lo, s[i].lo[k]); hi = std::max(hi, s[i].hi[k]); }
            org[k] = lo; ext[k] = hi - lo;
            int e = 0;  // flat box (all q = 0): scale 1 keeps the inverted box of an empty slot inverted on this axis too
            if (ext[k] > 0.0f) { int fe; std::frexp((double)ext[k] / 255.0, &fe); e = fe; }  // 2^fe > ext / 255
            e = std::max(-120, std::min(120, e));
            for (;; e++) {  // make sure the largest hi is reachable in float arithmetic
                const float sc = std::ldexp(1.0f, e);
                if (org[k] + 255.0f * sc >= hi || e >= 120) break;
            }
            eb[k] = (uint32_t)(e + 127);
        }
        uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                const double sc = std::ldexp(1.0, (int)eb[k] - 127);
                int a = (int)std::floor(((double)s[i].lo[k] - (double)org[k]) / sc);
                int b2 = (int)std::ceil(((double)s[i].hi[k] - (double)org[k]) / sc);
                a = std::max(0, std::min(255, a)); b2 = std::max(0, std::min(255, b2));
                // the device evaluates org + q * scale in fp32: step outwards if rounding landed inside the true box
                while (a > 0 && org[k] + (float)a * (float)sc > s[i].lo[k]) a--;
                while (b2 < 255 && org[k] + (float)b2 * (float)sc < s[i].hi[k]) b2++;
                qlo[k] |= (uint32_t)a << (8 * i);
                qhi[k] |= (uint32_t)b2 << (8 * i);
            }
        for (int i = n; i < 4; i++)
            for (int k = 0; k < 3; k++) qlo[k] |= 255u << (8 * i);  // empty slot: inverted box (qhi stays 0)
        uint32_t w[16];
        memcpy(&w[0], &org[0], 4); memcpy(&w[1], &org[1], 4); memcpy(&w[2], &org[2], 4);
        w[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16);
        w[4] = qlo[0]; w[5] = qlo[1]; w[6] = qlo[2]; w[7] = qhi[0];
        w[8] = qhi[1]; w[9] = qhi[2];
        uint32_t refs[4];
        for (int i = 0; i < 4; i++) {
            if (i >= n) refs[i] = NODE_EMPTY;
            else if (s[i].ref >= 0) refs[i] = (uint32_t)s[i].ref;
            else refs[i] = 0x80000000u | ((uint32_t)(~s[i].ref) << 3) | (uint32_t)s[i].count;
        }
        w[10] = refs[0]; w[11] = refs[1]; w[12] = refs[2]; w[13] = refs[3]; w[14] = 0; w[15] = 0;
        memcpy(&out[(size_t)out_index * 16], w, sizeof(w));
    }
};
// ---- the hottest nodes first ------------------------------------------------------------------------------------------
// A ray's chance of visiting a node goes with the node's surface area, and a child's box lies inside its parent's: the K nodes of
// largest area are an ancestor-closed crown of the tree that takes a third of all node visits at K = 32 and two fifths at K = 64
// (tools/bvh_eval.cpp on the bench scene: K = 1 0.07, 5 0.17, 16 0.27, 32 0.34, 64 0.41, 256 0.55).  They are moved to the indices
// [0, K) -- largest first, the root stays node 0 -- so that a kernel may keep records [0, K) somewhere closer than the L1
// (kernels.hip: the megakernel's LDS copy).  Everything else keeps its depth-first order.  Pure renumbering: the tree, every box and
// the order in which a ray visits its nodes are unchanged.
void hot_nodes_first(std::vector<float>& nodes, int K) {
    const int n = (int)(nodes.size() / 16);
    if (K > n) K = n;
    if (K <= 1) return;
    auto word = [&](int node, int w) { uint32_t v; memcpy(&v, &nodes[(size_t)node * 16 + w], 4); return v; };
    struct Cand { float area; int node; };
    auto worse = [](const Cand& a, const Cand& b) { return a.area < b.area || (a.area == b.area && a.node > b.node); };
    std::vector<Cand> heap;
    std::vector<int> hot;   // old indices, hottest first
    hot.push_back(0);
    auto push_children = [&](int node) {
        float org[3], sc[3];
        for (int k = 0; k < 3; k++) { memcpy(&org[k], &nodes[(size_t)node * 16 + k], 4); sc[k] = std::ldexp(1.0f, (int)((word(node, 3) >> (8 * k)) & 0xffu) - 127); }
        for (int i = 0; i < 4; i++) {
            const uint32_t ref = word(node, 10 + i);
            if (ref & 0x80000000u) continue;
            float e[3];
            for (int k = 0; k < 3; k++) {
                const float lo = org[k] + (float)((word(node, 4 + k) >> (8 * i)) & 0xffu) * sc[k], hi = org[k] + (float)((word(node, 7 + k) >> (8 * i)) & 0xffu) * sc[k];
                e[k] = hi - lo;
            }
            heap.push_back({e[0] * e[1] + e[1] * e[2] + e[2] * e[0], (int)ref});
            std::push_heap(heap.begin(), heap.end(), worse);
        }
    };
    push_children(0);
    while ((int)hot.size() < K && !heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), worse);
        const int node = heap.back().node;
        heap.pop_back();
        hot.push_back(node);
        push_children(node);
    }
    std::vector<int> new_index((size_t)n, -1);
    for (size_t i = 0; i < hot.size(); i++) new_index[(size_t)hot[i]] = (int)i;
    int next = (int)hot.size();
    for (int i = 0; i < n; i++) if (new_index[(size_t)i] < 0) new_index[(size_t)i] = next++;
    std::vector<float> moved(nodes.size());
    for (int i = 0; i < n; i++) {
        float* dst = &moved[(size_t)new_index[(size_t)i] * 16];
        memcpy(dst, &nodes[(size_t)i * 16], 64);
        for (int c = 0; c < 4; c++) {
            uint32_t ref; memcpy(&ref, dst + 10 + c, 4);
            if (!(ref & 0x80000000u)) { ref = (uint32_t)new_index[(size_t)ref]; memcpy(dst + 10 + c, &ref, 4); }
        }
    }
    nodes.swap(moved);
}
}  // namespace

// ---- fan pairs (round 6) -------------------------------------------------------------------------------------------------
// Inside every leaf the triangles are reordered so that the halves of a quad sit next to each other, A first (B.P0 == A.P0 and
// B.P1 == A.P2, compared as bits: B is then (v0, v2, v3) of the fan (v0, v1, v2, v3), and testing B from the pair record runs the
// very operations its own record would).  A leaf's box and its set of triangles do not change; the order in which its triangles are
// tested does (a tie in t between two triangles of one leaf goes to the one tested first, as before).
static void make_fan_pairs(Lbvh& out) {
    const size_t n = out.tri_orig.size();
    out.pairs.assign(16 * n, 0.0f);
    out.n_paired = 0;
    std::vector<uint8_t> is_a(n, 0);
    auto same3 = [](const float* a, const float* b) { return memcmp(a, b, 12) == 0; };
    auto fan = [&](const float* A, const float* B) { return same3(A, B) && same3(A + 8, B + 4); };   // records of 16 floats: P0 at 0, P1 at 4, P2 at 8
    const size_t n_nodes = out.nodes.size() / 16;
    for (size_t nd = 0; nd < n_nodes; nd++) {
        for (int c = 0; c < 4; c++) {
            uint32_t ref; memcpy(&ref, &out.nodes[nd * 16 + 10 + c], 4);
            if (!(ref & 0x80000000u)) continue;
            const size_t first = (ref & 0x7fffffffu) >> 3;
            const int cnt = (int)(ref & 7u);
            if (cnt < 2 || cnt > LEAF_MAX) continue;   // (a leaf never holds more than LEAF_MAX triangles: the arrays below are sized for that)
            // greedy matching in the leaf's own order (stable: unmatched triangles keep their relative order)
            float rec[LEAF_MAX > 0 ? LEAF_MAX : 1][16]; int32_t orig[LEAF_MAX > 0 ? LEAF_MAX : 1];
            for (int i = 0; i < cnt; i++) { memcpy(rec[i], &out.tris[(first + i) * 16], 64); orig[i] = out.tri_orig[first + i]; }
            int used[8] = {0, 0, 0, 0, 0, 0, 0, 0}, seq[8], a_flag[8] = {0, 0, 0, 0, 0, 0, 0, 0}, m = 0;
            for (int i = 0; i < cnt; i++) {
                if (used[i]) continue;
                int partner = -1, as_b = 0;
                for (int j = 0; j < cnt && partner < 0; j++) {
                    if (j == i || used[j]) continue;
                    if (fan(rec[i], rec[j])) { partner = j; as_b = 0; }
                    else if (fan(rec[j], rec[i])) { partner = j; as_b = 1; }
                }
                used[i] = 1;
                if (partner < 0) { seq[m++] = i; continue; }
                used[partner] = 1;
                a_flag[m] = 1;
                seq[m++] = as_b ? partner : i;
                seq[m++] = as_b ? i : partner;
            }
            for (int i = 0; i < cnt; i++) {
                memcpy(&out.tris[(first + i) * 16], rec[seq[i]], 64);
                out.tri_orig[first + i] = orig[seq[i]];
                if (a_flag[i]) { is_a[first + i] = 1; out.n_paired += 2; }
            }
        }
    }
    for (size_t i = 0; i < n; i++) {
        const float* t = &out.tris[i * 16];
        float* q = &out.pairs[i * 16];
        uint32_t meta; memcpy(&meta, t + 15, 4);
        uint32_t flags = meta & 0x80000000u;
        // P0 and the EDGES P1 - P0, P2 - P0 (, B.P2 - P0): the FP32 differences the triangle test starts with, taken here once (the
        // same IEEE subtractions the device would do per test: bit-identical operands)
        for (int k = 0; k < 3; k++) { q[k] = t[k]; q[4 + k] = t[4 + k] - t[k]; q[8 + k] = t[8 + k] - t[k]; }
        if (is_a[i]) {
            const float* b = &out.tris[(i + 1) * 16];
            uint32_t mb; memcpy(&mb, b + 15, 4);
            for (int k = 0; k < 3; k++) q[12 + k] = b[8 + k] - t[k];   // B.P2 - P0 (B.P0 = A.P0, B.P1 = A.P2: B's first edge is A's second)
            flags |= 1u | ((mb & 0x80000000u) >> 1);
        }
        memcpy(q + 15, &flags, 4);
    }
}

void build_lbvh(const HostMesh& mesh, Lbvh& out) {
    Builder b(mesh, out);
    b.run();
    std::vector<float> binary;
    binary.swap(out.nodes);
    out.nodes.assign(16, 0.0f);
    Collapser c(binary, out.nodes);
    c.emit(0, 0, 1);
    out.binary_depth = out.depth;
    out.depth = c.max_depth;
    int hot = HOT_NODES;
    if (const char* e = getenv("SPCBPT_BVH_HOT_NODES")) hot = std::max(0, atoi(e));   // developer knob (0 = depth-first order throughout)
    hot_nodes_first(out.nodes, hot);
    make_fan_pairs(out);
}

}  // namespace spc

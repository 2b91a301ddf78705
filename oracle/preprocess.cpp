// ORACLE — test infrastructure only (see vec.h).  C entry points over preprocess_ref.h with the same stage numbering as
// spcbpt_preprocess_stage (include/spcbpt.h) so the parity tests can compare stage by stage.
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "oracle_ctx.h"
#include "preprocess_ref.h"

using namespace orc;

static PreState& pre_of(orc_ctx* c) {
    if (!c->pre) c->pre = new PreState();
    return *static_cast<PreState*>(c->pre);
}
void orc_free_pre(orc_ctx* c) {
    delete static_cast<PreState*>(c->pre);
    c->pre = nullptr;
}

static void export_tree(const std::vector<tree_node>& t, spcbpt_tree_node* out) {
    for (size_t i = 0; i < t.size(); i++) {
        out[i].mid[0] = t[i].mid.x; out[i].mid[1] = t[i].mid.y; out[i].mid[2] = t[i].mid.z;
        for (int k = 0; k < 8; k++) out[i].child[k] = t[i].child[k];
        out[i].label = t[i].label; out[i].type = t[i].type; out[i].leaf = t[i].leaf ? 1 : 0;
    }
}

extern "C" {
int orc_set_subspace(orc_ctx* c, const spcbpt_tree_node*, int, const spcbpt_tree_node*, int, const float*, const float*);
int orc_launch(orc_ctx* c, const char* name, unsigned frame, int row_begin, int row_end, int row_step, int nthreads);
int orc_set_light_trace(orc_ctx* c, int num_core, int core_padding, int m_per_core);

// launchPretrace (optixPathTracer.cpp:523-551): one "pretrace" launch + valid_sample_gather
int orc_pretrace(orc_ctx* c, int iteration, int num_core, int padding, int nthreads) {
    PreState& st = pre_of(c);
    std::vector<preTracePath> paths(num_core);
    std::vector<preTraceConnection> conns((size_t)num_core * padding);
    std::atomic<int> next(0);
    auto work = [&]() {
        Params P = c->P;
        P.counters = nullptr;
        for (;;) {
            int i = next.fetch_add(64);
            if (i >= num_core) break;
            for (int k = i; k < std::min(i + 64, num_core); k++) raygen_TrainData(P, k, iteration, padding, paths.data(), conns.data());
        }
    };
    if (nthreads <= 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(work);
        for (auto& x : th) x.join();
    }
    return valid_sample_gather(st, paths, conns);
}
int orc_train_records_count(orc_ctx* c, int* n_paths, int* n_nodes) {
    PreState& st = pre_of(c);
    *n_paths = (int)st.neat_paths.size(); *n_nodes = (int)st.neat_conns.size();
    return 0;
}
int orc_train_records_read(orc_ctx* c, spcbpt_pretrace_path* paths, spcbpt_pretrace_node* nodes) {
    PreState& st = pre_of(c);
    memcpy(paths, st.neat_paths.data(), st.neat_paths.size() * sizeof(preTracePath));
    memcpy(nodes, st.neat_conns.data(), st.neat_conns.size() * sizeof(preTraceConnection));
    return 0;
}
int orc_train_records_import(orc_ctx* c, const spcbpt_pretrace_path* paths, int n_paths, const spcbpt_pretrace_node* nodes, int n_nodes) {
    PreState& st = pre_of(c);
    st.neat_paths.assign(paths, paths + n_paths);
    st.neat_conns.assign(nodes, nodes + n_nodes);
    return 0;
}
int orc_train_records_clear(orc_ctx* c) { PreState& st = pre_of(c); st.neat_paths.clear(); st.neat_conns.clear(); return 0; }

int orc_preprocess_stage(orc_ctx* c, int stage, int arg, int nthreads) {
    PreState& st = pre_of(c);
    const int NS = SPCBPT_NUM_SUBSPACE;
    switch (stage) {
        case 1: {
            sample_reweight(st, (int)c->P.width, (int)c->P.height);
            auto s = get_weighted_point_for_tree_building(st, true, 100000);
            st.eye_tree = buildTreeBaseOnExistSample()(s, NS, 0);
            s = get_weighted_point_for_tree_building(st, false, 100000);
            st.light_tree = buildTreeBaseOnExistSample()(s, NS - SPCBPT_NUM_SUBSPACE_LIGHTSOURCE, 0);
            return 0;
        }
        case 2: {
            // trees installed, Q / CMFGamma still null -> gamma_ss() == 1 exactly as in the reference at this point
            c->eye_tree = st.eye_tree; c->light_tree = st.light_tree;
            c->P.eye_tree = c->eye_tree.data(); c->P.light_tree = c->light_tree.data();
            c->P.Q = nullptr; c->P.CMFGamma = nullptr;
            st.h_Q_vec.clear();
            int target = arg > 0 ? arg : 2000000;
            unsigned frame = 0;
            while (st.acc_valid_path < target || st.h_Q_vec.empty()) {
                int rc = orc_launch(c, "light trace", ++frame, 0, 0, 1, nthreads);
                if (rc) return rc;
                preprocess_getQ(st, c->P.lt);
            }
            Q_zero_handle(st);
            return 0;
        }
        case 3:
            node_label(st);
            build_optimal_E_train_data(st, arg > 0 ? arg : (int)st.neat_paths.size());
            preprocess_getGamma(st);
            return 0;
        case 4:
            train_optimal_E(st, arg > 0 ? arg : 20000, 1, 0.01f);
            return 0;
        case 5: {
            Gamma2CMFGamma(st);
            c->eye_tree = st.eye_tree; c->light_tree = st.light_tree; c->Q = st.h_Q_vec; c->CMFGamma = st.CMFGamma;
            c->P.eye_tree = c->eye_tree.data(); c->P.light_tree = c->light_tree.data(); c->P.Q = c->Q.data(); c->P.CMFGamma = c->CMFGamma.data();
            return 0;
        }
    }
    return -1;
}
int orc_get_gamma(orc_ctx* c, float* g) { PreState& st = pre_of(c); memcpy(g, st.h_Gamma.data(), st.h_Gamma.size() * 4); return 0; }
int orc_get_q(orc_ctx* c, float* q) { PreState& st = pre_of(c); memcpy(q, st.h_Q_vec.data(), st.h_Q_vec.size() * 4); return 0; }
int orc_get_cmf_gamma(orc_ctx* c, float* g) { PreState& st = pre_of(c); memcpy(g, st.CMFGamma.data(), st.CMFGamma.size() * 4); return 0; }
int orc_get_tree(orc_ctx* c, int light, spcbpt_tree_node* out, int cap, int* n) {
    PreState& st = pre_of(c);
    const std::vector<tree_node>& t = light ? st.light_tree : st.eye_tree;
    *n = (int)t.size();
    if (!out) return 0;
    if (cap < *n) return SPCBPT_ERR_CAPACITY;
    export_tree(t, out);
    return 0;
}
// a single tree build on caller-provided samples (10 floats each: position, dir, normal, weight)
int orc_build_tree(const float* samples, int n, int subspace_size, int label_bias, spcbpt_tree_node* out, int cap, int* n_out) {
    std::vector<divide_weight> s(n);
    for (int i = 0; i < n; i++) {
        const float* r = samples + 10 * (size_t)i;
        s[i].position = load3(r); s[i].dir = load3(r + 3); s[i].normal = load3(r + 6); s[i].weight = r[9];
    }
    auto t = buildTreeBaseOnExistSample()(s, subspace_size, label_bias);
    *n_out = (int)t.size();
    if (cap < *n_out) return SPCBPT_ERR_CAPACITY;
    export_tree(t, out);
    return 0;
}
}

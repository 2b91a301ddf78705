// Device-driven half of the host-owned preprocessing (optixPathTracer.cpp:523-608): pretrace launches + record gather,
// light passes for Q, and the stage driver around Preprocessor (preprocess_host.cpp).
#include <hip/hip_runtime.h>

#include <cstring>

#include "context.h"
#include "kernels.h"
#include "preprocess_host.h"

namespace spc {

#define HIP_TRY_P(expr)                                                                  \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) { error = std::string(#expr) + ": " + hipGetErrorString(e__); return SPCBPT_ERR_HIP; } \
    } while (0)

int Context::set_pretrace(int num_core, int padding) {
    if (num_core < 1 || padding < 2 || padding > 10) { error = "set_pretrace: num_core >= 1 and 2 <= padding <= 10 (PRETRACE_CONN_PADDING)"; return SPCBPT_ERR_INVALID_ARG; }
    pre_num_core = num_core; pre_padding = padding;
    return 0;
}

int Context::launch_pretrace(uint32_t iteration) {
    if (!have_camera || !d_accum) { error = "pretrace needs a camera and an image size (pixel ids)"; return SPCBPT_ERR_STATE; }
    if (!have_subspace) { int rc = install_minimal_tuple(); if (rc) return rc; }
    if (!pre) pre = new Preprocessor();
    const size_t np = (size_t)pre_num_core, nn = np * (size_t)pre_padding;
    if (np > pre_capacity) {
        if (d_pre_paths) (void)hipFree(d_pre_paths);
        if (d_pre_nodes) (void)hipFree(d_pre_nodes);
        d_pre_paths = nullptr; d_pre_nodes = nullptr;
        HIP_TRY_P(hipMalloc((void**)&d_pre_paths, np * sizeof(spcbpt_pretrace_path)));
        HIP_TRY_P(hipMalloc((void**)&d_pre_nodes, np * 10 * sizeof(spcbpt_pretrace_node)));
        pre_capacity = np;
    }
    int rc = ensure_spill((np + 255) / 256 * 256);
    if (rc) return rc;
    kp.counters = nullptr;
    time_begin("pretrace");
    ::spc::launch_pretrace(kp, iteration, pre_num_core, pre_padding, d_pre_paths, d_pre_nodes, stream);
    time_end();
    HIP_TRY_P(hipGetLastError());
    std::vector<spcbpt_pretrace_path> hp(np);
    std::vector<spcbpt_pretrace_node> hn(nn);
    HIP_TRY_P(hipMemcpyAsync(hp.data(), d_pre_paths, np * sizeof(spcbpt_pretrace_path), hipMemcpyDeviceToHost, stream));
    HIP_TRY_P(hipMemcpyAsync(hn.data(), d_pre_nodes, nn * sizeof(spcbpt_pretrace_node), hipMemcpyDeviceToHost, stream));
    HIP_TRY_P(hipStreamSynchronize(stream));
    pre_last_added = pre->gather(hp.data(), (int)np, hn.data(), pre_padding);
    return 0;
}

int Context::preprocess_stage(int stage, int arg) {
    if (!pre) { error = "no training records (run \"pretrace\" or import records first)"; return SPCBPT_ERR_STATE; }
    Preprocessor& P = *pre;
    switch (stage) {
        case 1: {  // sample_reweight + trees (optixPathTracer.cpp:562-572)
            if (P.paths.empty()) { error = "stage 1: no records"; return SPCBPT_ERR_STATE; }
            P.sample_reweight((int)kp.width, (int)kp.height);
            P.eye_tree = Preprocessor::build_tree(P.tree_samples(true, 100000), SPCBPT_NUM_SUBSPACE, 0);
            P.light_tree = Preprocessor::build_tree(P.tree_samples(false, 100000), SPCBPT_NUM_SUBSPACE - SPCBPT_NUM_SUBSPACE_LIGHTSOURCE, 0);
            return 0;
        }
        case 2: {  // Q from light passes with the new trees (583-593)
            if (P.eye_tree.empty()) { error = "stage 2 before stage 1"; return SPCBPT_ERR_STATE; }
            std::vector<float> q1(SPCBPT_NUM_SUBSPACE, 1.0f), g((size_t)SPCBPT_NUM_SUBSPACE * SPCBPT_NUM_SUBSPACE);
            for (int e = 0; e < SPCBPT_NUM_SUBSPACE; e++)
                for (int l = 0; l < SPCBPT_NUM_SUBSPACE; l++) g[(size_t)e * SPCBPT_NUM_SUBSPACE + l] = (float)(l + 1) / SPCBPT_NUM_SUBSPACE;
            int rc = install_subspace(P.eye_tree.data(), (int)P.eye_tree.size(), P.light_tree.data(), (int)P.light_tree.size(), q1.data(), g.data());
            if (rc) return rc;
            P.Q.clear();
            P.q_acc_paths = 0;
            const int target = arg > 0 ? arg : 2000000;
            // the whole pass is traced by this context (no sharding while preprocessing)
            spcbpt_light_trace_params saved = lt, full = lt;
            full.core_begin = 0; full.core_count = 0;
            rc = set_light_trace(full);
            if (rc) return rc;
            std::vector<uint32_t> keys;
            std::vector<float> w;
            uint32_t frame = 0;
            while (P.q_acc_paths < target) {
                rc = launch_light(++frame);
                if (rc) return rc;
                rc = fetch_counts();
                if (rc) return rc;
                keys.resize(lvc_count); w.resize(lvc_count);
                HIP_TRY_P(hipMemcpy(keys.data(), d_keys, (size_t)lvc_count * 4, hipMemcpyDeviceToHost));
                HIP_TRY_P(hipMemcpy(w.data(), d_weights, (size_t)lvc_count * 4, hipMemcpyDeviceToHost));
                if (path_count <= 0) { error = "stage 2: a light pass produced no paths"; return SPCBPT_ERR_STATE; }
                P.accumulate_q(keys.data(), w.data(), lvc_count, path_count);
            }
            P.q_zero_handle();
            rc = set_light_trace(saved);
            return rc;
        }
        case 3: {  // node_label + build_optimal_E_train_data + preprocess_getGamma (594-599)
            if (P.Q.empty()) { error = "stage 3 before stage 2"; return SPCBPT_ERR_STATE; }
            P.label_nodes();
            P.build_train_data(arg > 0 ? arg : (int)P.paths.size());
            P.initial_gamma();
            return 0;
        }
        case 4: {  // train_optimal_E (600)
            if (P.gamma.empty()) { error = "stage 4 before stage 3"; return SPCBPT_ERR_STATE; }
            P.train(arg > 0 ? arg : 20000, 1, 0.01f);
            return 0;
        }
        case 5: {  // Gamma2CMFGamma + install (606-607)
            if (P.gamma.empty()) { error = "stage 5 before stage 3"; return SPCBPT_ERR_STATE; }
            P.make_cmf();
            return install_subspace(P.eye_tree.data(), (int)P.eye_tree.size(), P.light_tree.data(), (int)P.light_tree.size(), P.Q.data(), P.cmf_gamma.data());
        }
        default: error = "unknown preprocessing stage"; return SPCBPT_ERR_INVALID_ARG;
    }
}

int Context::preprocess(int target_paths, int target_q_paths, bool train) {
    if (target_paths < 1000) { error = "preprocess: target_paths must be >= 1000 (outlier probe size)"; return SPCBPT_ERR_INVALID_ARG; }
    if (pre) { pre->paths.clear(); pre->nodes.clear(); }
    uint32_t iteration = 0;
    int guard = 0;
    while (!pre || (int)pre->paths.size() < target_paths) {
        int rc = launch_pretrace(++iteration);
        if (rc) return rc;
        if (pre_last_added == 0 && ++guard > 8) { error = "pretrace produced no valid paths (no light reaches the camera?)"; return SPCBPT_ERR_STATE; }
    }
    const int batch = 20000;
    int n_train = train && target_paths >= batch ? target_paths / batch * batch : target_paths;
    int rc;
    if ((rc = preprocess_stage(1, 0))) return rc;
    if ((rc = preprocess_stage(2, target_q_paths))) return rc;
    if ((rc = preprocess_stage(3, n_train))) return rc;
    if (train && (rc = preprocess_stage(4, std::min(batch, n_train)))) return rc;
    return preprocess_stage(5, 0);
}

void Context::free_preprocess() {
    delete pre;
    pre = nullptr;
    if (d_pre_paths) (void)hipFree(d_pre_paths);
    if (d_pre_nodes) (void)hipFree(d_pre_nodes);
    d_pre_paths = nullptr; d_pre_nodes = nullptr;
}

}  // namespace spc

using namespace spc;
struct spcbpt_ctx : public spc::Context {};

extern "C" {
int spcbpt_set_pretrace(spcbpt_ctx* c, int num_core, int padding) {
    if (!c) return SPCBPT_ERR_INVALID_ARG;
    return c->set_pretrace(num_core, padding);
}
int spcbpt_train_records_count(spcbpt_ctx* c, int* n_paths, int* n_nodes) {
    if (!c || !n_paths || !n_nodes) return SPCBPT_ERR_INVALID_ARG;
    *n_paths = c->pre ? (int)c->pre->paths.size() : 0;
    *n_nodes = c->pre ? (int)c->pre->nodes.size() : 0;
    return SPCBPT_OK;
}
int spcbpt_train_records_read(spcbpt_ctx* c, spcbpt_pretrace_path* paths, int cap_paths, spcbpt_pretrace_node* nodes, int cap_nodes) {
    if (!c || !paths || !nodes) return SPCBPT_ERR_INVALID_ARG;
    if (!c->pre) { c->error = "no training records"; return SPCBPT_ERR_STATE; }
    if (cap_paths < (int)c->pre->paths.size() || cap_nodes < (int)c->pre->nodes.size()) { c->error = "records_read: buffers too small"; return SPCBPT_ERR_CAPACITY; }
    memcpy(paths, c->pre->paths.data(), c->pre->paths.size() * sizeof(spcbpt_pretrace_path));
    memcpy(nodes, c->pre->nodes.data(), c->pre->nodes.size() * sizeof(spcbpt_pretrace_node));
    return SPCBPT_OK;
}
int spcbpt_train_records_import(spcbpt_ctx* c, const spcbpt_pretrace_path* paths, int n_paths, const spcbpt_pretrace_node* nodes, int n_nodes) {
    if (!c || !paths || !nodes || n_paths < 0 || n_nodes < 0) return SPCBPT_ERR_INVALID_ARG;
    for (int i = 0; i < n_paths; i++)
        if (paths[i].begin_ind < 0 || paths[i].end_ind < paths[i].begin_ind || paths[i].end_ind > n_nodes) { c->error = "records_import: node range out of bounds"; return SPCBPT_ERR_INVALID_ARG; }
    if (!c->pre) c->pre = new Preprocessor();
    c->pre->paths.assign(paths, paths + n_paths);
    c->pre->nodes.assign(nodes, nodes + n_nodes);
    return SPCBPT_OK;
}
int spcbpt_train_records_clear(spcbpt_ctx* c) {
    if (!c) return SPCBPT_ERR_INVALID_ARG;
    if (c->pre) { c->pre->paths.clear(); c->pre->nodes.clear(); }
    return SPCBPT_OK;
}
int spcbpt_preprocess_stage(spcbpt_ctx* c, int stage, int arg) {
    if (!c) return SPCBPT_ERR_INVALID_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return SPCBPT_ERR_HIP;
    return c->preprocess_stage(stage, arg);
}
int spcbpt_get_gamma(spcbpt_ctx* c, float* gamma) {
    if (!c || !gamma) return SPCBPT_ERR_INVALID_ARG;
    if (!c->pre || c->pre->gamma.empty()) { c->error = "no Gamma yet (preprocessing stage 3)"; return SPCBPT_ERR_STATE; }
    memcpy(gamma, c->pre->gamma.data(), c->pre->gamma.size() * sizeof(float));
    return SPCBPT_OK;
}

// Checkpoint files (checkpoint.cpp) on the context's tuple
int spcbpt_checkpoint_save(spcbpt_ctx* c, const char* dir) {
    if (!c || !dir) return SPCBPT_ERR_INVALID_ARG;
    if (!c->have_subspace) { c->error = "checkpoint_save: no subspace tuple installed"; return SPCBPT_ERR_STATE; }
    if (!c->pre || c->pre->gamma.empty()) { c->error = "checkpoint_save: no Gamma (the context never ran preprocessing stage 3+)"; return SPCBPT_ERR_STATE; }
    const int rc = spcbpt_checkpoint_write(dir, c->h_eye_tree.data(), (int)c->h_eye_tree.size(), c->h_light_tree.data(),
                                           (int)c->h_light_tree.size(), c->h_Q.data(), c->pre->gamma.data());
    if (rc) c->error = std::string("checkpoint_save: cannot write the checkpoint files in ") + dir;
    return rc;
}
int spcbpt_checkpoint_load(spcbpt_ctx* c, const char* dir) {
    if (!c || !dir) return SPCBPT_ERR_INVALID_ARG;
    const int cap = 1 << 20;
    std::vector<spcbpt_tree_node> et(cap), lt(cap);
    std::vector<float> q(SPCBPT_NUM_SUBSPACE), gamma((size_t)SPCBPT_NUM_SUBSPACE * SPCBPT_NUM_SUBSPACE, 0.0f), cmf(gamma.size());
    const bool have = c->pre && !c->pre->gamma.empty();
    if (have) gamma = c->pre->gamma;
    int ne = 0, nl = 0;
    int rc = spcbpt_checkpoint_read(dir, et.data(), &ne, cap, lt.data(), &nl, cap, q.data(), gamma.data(), have ? 1 : 0);
    if (rc) { c->error = std::string("checkpoint_load: missing or malformed checkpoint files in ") + dir; return rc; }
    spcbpt_gamma_to_cmf(gamma.data(), cmf.data());
    rc = c->install_subspace(et.data(), ne, lt.data(), nl, q.data(), cmf.data());
    if (rc) return rc;
    if (!c->pre) c->pre = new Preprocessor();
    c->pre->gamma = gamma;  // what spcbpt_get_gamma / the next save return
    return SPCBPT_OK;
}
}

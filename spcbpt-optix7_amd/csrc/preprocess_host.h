// Host-owned preprocessing of SPCBPT (north_star: "C++ host code owns ... sampling-matrix training"): everything
// preprocessing() does between the pretrace launches and the installation of (eye tree, light tree, Q, CMF Gamma)
// — optixPathTracer.cpp:552-608.  Pure host C++; the device passes (pretrace, light trace) are driven by Context.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/spcbpt.h"

namespace spc {

struct TreeSample {  // classTree::divide_weight (decisionTree/classTree_common.h:76-91)
    float position[3], dir[3], normal[3];
    float weight;
};

struct Preprocessor {
    std::vector<spcbpt_pretrace_path> paths;  // neat_paths
    std::vector<spcbpt_pretrace_node> nodes;  // neat_conns
    std::vector<spcbpt_tree_node> eye_tree, light_tree;
    std::vector<float> Q;          // h_Q_vec (1000)
    std::vector<float> gamma;      // h_Gamma / E (1000 x 1000)
    std::vector<float> cmf_gamma;  // CMFGamma
    long long q_acc_paths = 0;     // acc_valid_path of preprocess_getQ
    float last_mean_loss = 0.0f;

    // valid_sample_gather (device_thrust.cu:457-493) on one launch's padded output; returns #valid paths appended
    int gather(const spcbpt_pretrace_path* raw_paths, int n_paths, const spcbpt_pretrace_node* raw_nodes, int padding);
    // sample_reweight (574-623); tile pitch = ceil(width / 10) (the reference hard-codes 192 = 1920 / 10)
    void sample_reweight(int width, int height);
    // get_weighted_point_for_tree_building (494-527); light-source nodes are SKIPPED on the light side (the reference
    // pushes an uninitialised sample there)
    std::vector<TreeSample> tree_samples(bool eye_side, int max_paths) const;
    // classTree::buildTreeBaseOnExistSample (decisionTree/classTree_host.h:61-431)
    static std::vector<spcbpt_tree_node> build_tree(std::vector<TreeSample> samples, int subspace_size, int label_bias,
                                                    float threshold = 0.99f, int max_depth = 15);
    // preprocess_getQ (347-409) for one light pass: per-vertex subspace ids and weights in LVC slot order + #paths
    void accumulate_q(const uint32_t* subspace, const float* weight, int n, int path_count);
    void q_zero_handle();  // 335-346
    // node_label (554-573)
    void label_nodes();
    // build_optimal_E_train_data (3261-3325) + preprocess_getGamma (627-667)
    void build_train_data(int n_samples);
    void initial_gamma();
    // train_optimal_E (3327-3344): matrix_parameter::fit (1615-1655) with matrix_optimal_operator (923-1228) + Adam (1438-1559)
    void train(int batch_size = 20000, int epochs = 1, float lr = 0.01f);
    // Gamma2CMFGamma (3406-3433)
    void make_cmf();

    // training set in the layout of matrix_parameter::train_data
    std::vector<float> f_square, pdf0, pdf_peak;
    std::vector<int> label_E, label_P, P2N;
    int n_train_paths = 0, m_train_nodes = 0;
};

int tree_index_host(const std::vector<spcbpt_tree_node>& t, const float* position, const float* normal, const float* dir);

}  // namespace spc

// See preprocess_host.h.  Reference line numbers refer to src/OptiXPathTracer/cuda_thrust/device_thrust.cu unless
// another file is named.
#include "preprocess_host.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace spc {

static const int NS = SPCBPT_NUM_SUBSPACE;

static inline float sum3(const float* c) { return c[0] + c[1] + c[2]; }

int tree_index_host(const std::vector<spcbpt_tree_node>& t, const float* position, const float* normal, const float* dir) {
    int node = 0;
    while (!t[node].leaf) {
        const spcbpt_tree_node& n = t[node];
        const float* p = n.type == 0 ? position : (n.type == 1 ? normal : dir);
        int ind = (p[0] > n.mid[0] ? 1 : 0) + (p[1] > n.mid[1] ? 2 : 0) + (p[2] > n.mid[2] ? 4 : 0);
        node = n.child[ind];
    }
    return t[node].label;
}

int Preprocessor::gather(const spcbpt_pretrace_path* raw_paths, int n_paths, const spcbpt_pretrace_node* raw_nodes, int padding) {
    int added = 0;
    for (int i = 0; i < n_paths; i++) {
        const spcbpt_pretrace_path& rp = raw_paths[i];
        if (!rp.valid) continue;
        spcbpt_pretrace_path p = rp;
        const int count = rp.end_ind - rp.begin_ind;
        p.begin_ind = (int)nodes.size();
        p.end_ind = p.begin_ind + count;
        const int id = (int)paths.size();
        for (int k = 0; k < count; k++) {
            spcbpt_pretrace_node nd = raw_nodes[(size_t)rp.begin_ind + k];  // begin_ind already carries launch_index * padding
            nd.path_id = id;
            nodes.push_back(nd);
        }
        paths.push_back(p);
        added++;
    }
    (void)padding;
    return added;
}

void Preprocessor::sample_reweight(int width, int height) {
    const int pitch = (width + 9) / 10, rows = (height + 9) / 10;
    std::vector<float> weight((size_t)pitch * rows + pitch, 0.0f);
    auto tile = [&](const spcbpt_pretrace_path& s) {
        int id_x = s.pixel_id[0] / 10, id_y = s.pixel_id[1] / 10;
        int n = id_x + id_y * pitch;
        return std::min(std::max(n, 0), (int)weight.size() - 1);
    };
    for (const auto& s : paths) {
        float ww = sum3(s.contri) / s.sample_pdf;
        if (std::isnan(ww) || std::isinf(ww)) continue;
        weight[tile(s)] += ww;
    }
    for (auto& s : paths) {
        float w = (float)((double)(weight[tile(s)] / 100) + 0.1);
        float inv = 1.0f / w;  // float3 / float of vec_math.h multiplies by the reciprocal
        s.contri[0] *= inv; s.contri[1] *= inv; s.contri[2] *= inv;
    }
}

std::vector<TreeSample> Preprocessor::tree_samples(bool eye_side, int max_paths) const {
    std::vector<TreeSample> ans;
    const int limit = max_paths == 0 ? (int)paths.size() : std::min((int)paths.size(), max_paths);
    for (int i = 0; i < limit; i++) {
        const float w = sum3(paths[i].contri) / paths[i].sample_pdf;
        for (int j = paths[i].begin_ind; j < paths[i].end_ind; j++) {
            const spcbpt_pretrace_node& n = nodes[j];
            TreeSample t;
            if (eye_side) {
                memcpy(t.dir, n.a_dir, 12); memcpy(t.normal, n.a_normal, 12); memcpy(t.position, n.a_position, 12);
            } else {
                if (n.light_source) continue;
                memcpy(t.dir, n.b_dir, 12); memcpy(t.normal, n.b_normal, 12); memcpy(t.position, n.b_position, 12);
            }
            t.weight = w;
            ans.push_back(t);
        }
    }
    return ans;
}

// ---------------------------------------------------------------- subspace trees (classTree_host.h)
namespace {
struct LSample { TreeSample s; int label; };
struct DivideNode {
    spcbpt_tree_node n;
    std::vector<LSample> v;
    int depth = 0, father = 0, position_depth = 0, normal_depth = 0, dir_depth = 0;
    float weight = 0.0f, correct_weight = 0.0f;
    DivideNode() {
        memset(&n, 0, sizeof(n));
        n.leaf = 1; n.label = 0; n.type = 0;
    }
    void add_sample(const LSample& w) { v.push_back(w); weight += w.s.weight; }
    bool need_split() const { return !v.empty() && correct_weight < weight; }
};
struct TreeBuilder {
    std::vector<DivideNode> v;
    std::vector<float> block_size, dir_block_size;  // 3 floats per level
    float bbox_min[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float bbox_max[3] = {FLT_MIN, FLT_MIN, FLT_MIN};  // FLT_MIN (not -FLT_MAX) exactly as classTree_host.h:100-101

    int child_of(const DivideNode& d, const TreeSample& s) const {
        const float* p = d.n.type == 0 ? s.position : (d.n.type == 1 ? s.normal : s.dir);
        int ind = (p[0] > d.n.mid[0] ? 1 : 0) + (p[1] > d.n.mid[1] ? 2 : 0) + (p[2] > d.n.mid[2] ? 4 : 0);
        return d.n.child[ind];
    }
    void color(int id) {  // classTree_host.h:243-284
        DivideNode& t = v[id];
        if (t.v.empty()) { t.correct_weight = 0.0f; return; }
        bool need = false;
        t.n.label = t.v[0].label;
        for (const auto& s : t.v) if (s.label != t.n.label) { need = true; break; }
        if (need) {
            static thread_local std::vector<float> weights;
            weights.assign(NS, 0.0f);
            float max_weight = 0.0f;
            int max_id = t.n.label;
            for (const auto& s : t.v) {
                weights[s.label] += s.s.weight;
                if (max_weight < weights[s.label]) { max_weight = weights[s.label]; max_id = s.label; }
            }
            t.n.label = max_id;
            t.correct_weight = max_weight;
        } else {
            t.correct_weight = t.weight;
        }
    }
    float split(int id) {  // classTree_host.h:103-211 (DIR_JUDGE == 0: never a direction split)
        const int type = (v[id].depth % 2 == 0 || v[id].normal_depth > 3) ? 0 : 1;
        const int back = (int)v.size();
        v[id].n.leaf = 0;
        const float* inch = type == 0 ? &block_size[3 * (v[id].position_depth + 1)] : &dir_block_size[3 * (v[id].normal_depth + 1)];
        float mid[3];
        if (v[id].normal_depth == 0 && type == 1) {
            mid[0] = mid[1] = mid[2] = 0.0f;
        } else if (v[id].position_depth == 0) {
            memcpy(mid, v[id].n.mid, 12);
        } else {
            int L_id = id, t_id = v[id].father;
            while (t_id != 0 && v[t_id].n.type != type) { L_id = t_id; t_id = v[t_id].father; }
            memcpy(mid, v[t_id].n.mid, 12);
            int c = 0;
            for (; c < 8; c++) if (v[t_id].n.child[c] == L_id) break;
            mid[0] += ((c >> 0) % 2 == 0) ? -inch[0] : inch[0];
            mid[1] += ((c >> 1) % 2 == 0) ? -inch[1] : inch[1];
            mid[2] += ((c >> 2) % 2 == 0) ? -inch[2] : inch[2];
        }
        memcpy(v[id].n.mid, mid, 12);
        v[id].n.type = type;
        for (int i = 0; i < 8; i++) {
            v[id].n.child[i] = back + i;
            DivideNode c;
            c.father = id;
            c.depth = v[id].depth + 1;
            c.n.label = v[id].n.label;
            c.position_depth = v[id].position_depth + (type == 0);
            c.normal_depth = v[id].normal_depth + (type == 1);
            c.dir_depth = v[id].dir_depth;
            v.push_back(std::move(c));
        }
        for (const auto& s : v[id].v) v[child_of(v[id], s.s)].add_sample(s);
        float n_correct = 0.0f;
        for (int i = 0; i < 8; i++) {
            color(v[id].n.child[i]);
            n_correct += v[v[id].n.child[i]].correct_weight;
        }
        v[id].weight = 0;
        v[id].v.clear();
        v[id].v.shrink_to_fit();
        return n_correct;
    }
};
}  // namespace

std::vector<spcbpt_tree_node> Preprocessor::build_tree(std::vector<TreeSample> samples, int subspace_size, int label_bias, float threshold,
                                                      int max_depth) {
    std::vector<spcbpt_tree_node> out;
    const int n = (int)samples.size();
    if (n < 2) {
        spcbpt_tree_node leaf;
        memset(&leaf, 0, sizeof(leaf));
        leaf.leaf = 1; leaf.label = label_bias;
        out.push_back(leaf);
        return out;
    }
    // get_position_variance (classTree_host.h:286-301)
    float mean[3] = {0, 0, 0}, var[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) mean[k] += samples[i].position[k] * (1.0f / (float)n);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            float d = mean[k] - samples[i].position[k];
            var[k] += d * d * (1.0f / (float)(n - 1));
        }
    const float diversity2 = std::max(var[0], std::max(var[1], var[2]));
    // centroids by weight stride (303-322)
    float weight_sum = 0;
    for (const auto& p : samples) weight_sum += p.weight;
    std::vector<TreeSample> centers;
    float acc = 0;
    const float stride = weight_sum / subspace_size;
    for (const auto& p : samples) {
        acc += p.weight;
        if (acc > stride) { acc -= stride; centers.push_back(p); }
    }
    // nearest-centroid labels under d = |dp|^2 + diversity2 * (1 - n.n') (323-340; DIR_JUDGE = 0)
    std::vector<LSample> labeled(n);
    for (int i = 0; i < n; i++) {
        const TreeSample& p = samples[i];
        float min_d = FLT_MAX;
        int id = 0;
        for (int c = 0; c < (int)centers.size(); c++) {
            const TreeSample& a = centers[c];
            float dx = a.position[0] - p.position[0], dy = a.position[1] - p.position[1], dz = a.position[2] - p.position[2];
            float d_a = dx * dx + dy * dy + dz * dz;
            float diff_direction = p.dir[0] * a.dir[0] + p.dir[1] * a.dir[1] + p.dir[2] * a.dir[2];
            float diff_normal = p.normal[0] * a.normal[0] + p.normal[1] * a.normal[1] + p.normal[2] * a.normal[2];
            float d = d_a + diversity2 * ((1 - diff_normal) + (1 - diff_direction) * 0.0f);
            if (d < min_d) { min_d = d; id = c + label_bias; }
        }
        labeled[i].s = p;
        labeled[i].label = id;
    }
    // para_initial (213-241)
    TreeBuilder b;
    float unnorm = 0.0f;
    for (auto& s : labeled) {
        unnorm += s.s.weight;
        for (int k = 0; k < 3; k++) {
            b.bbox_min[k] = fminf(b.bbox_min[k], s.s.position[k]);
            b.bbox_max[k] = fmaxf(b.bbox_max[k], s.s.position[k]);
        }
    }
    for (auto& s : labeled) s.s.weight /= unnorm;
    float blk[3] = {b.bbox_max[0] - b.bbox_min[0], b.bbox_max[1] - b.bbox_min[1], b.bbox_max[2] - b.bbox_min[2]};
    for (int i = 0; i < max_depth + 10; i++) {
        b.block_size.insert(b.block_size.end(), blk, blk + 3);
        for (int k = 0; k < 3; k++) blk[k] *= 0.5f;
    }
    float dblk[3] = {2.0f, 2.0f, 2.0f};
    for (int i = 0; i < 15; i++) {
        b.dir_block_size.insert(b.dir_block_size.end(), dblk, dblk + 3);
        for (int k = 0; k < 3; k++) dblk[k] *= 0.5f;
    }
    // breadth-first refinement (344-372)
    b.v.emplace_back();
    b.v[0].v = std::move(labeled);
    b.v[0].weight = 1;
    for (int k = 0; k < 3; k++) b.v[0].n.mid[k] = (b.bbox_max[k] + b.bbox_min[k]) * 0.5f;
    b.color(0);
    float c_w = b.v[0].correct_weight;
    for (size_t i = 0; i < b.v.size(); i++) {
        if (b.v[i].need_split() && b.v[i].depth < max_depth && threshold > c_w) {
            c_w -= b.v[i].correct_weight;
            c_w += b.split((int)i);
        }
    }
    out.resize(b.v.size());
    for (size_t i = 0; i < b.v.size(); i++) out[i] = b.v[i].n;
    return out;
}

// ---------------------------------------------------------------- Q
void Preprocessor::accumulate_q(const uint32_t* subspace, const float* weight, int n, int path_count) {
    if (Q.empty()) { Q.assign(NS, 0.0f); q_acc_paths = 0; }
    std::vector<float> tmp(NS, 0.0f);
    for (int i = 0; i < n; i++) tmp[subspace[i]] += weight[i];
    q_acc_paths += path_count;
    const float t = path_count / (float)q_acc_paths;
    for (int i = 0; i < NS; i++) {
        tmp[i] /= path_count;
        Q[i] = Q[i] * (1 - t) + tmp[i] * t;
    }
}
void Preprocessor::q_zero_handle() {
    for (int i = 0; i < NS; i++) if (Q[i] == 0) Q[i] = FLT_MAX;
}

void Preprocessor::label_nodes() {
    for (auto& s : nodes) {
        s.label_a = tree_index_host(eye_tree, s.a_position, s.a_normal, s.a_dir);
        if (!s.light_source) s.label_b = tree_index_host(light_tree, s.b_position, s.b_normal, s.b_dir);
    }
}

// ---------------------------------------------------------------- training data
static const float kLossThreshold = 1000000.0f;  // optimal_E_loss_threshold (3097)

void Preprocessor::build_train_data(int n_samples) {
    n_samples = std::min(n_samples, (int)paths.size());
    n_train_paths = n_samples;
    m_train_nodes = n_samples > 0 ? paths[n_samples - 1].end_ind : 0;
    auto outlier_value = [&](const spcbpt_pretrace_path& s) {  // get_outler_value (3172-3197)
        float ov = s.fix_pdf;
        float w = sum3(s.contri);
        float loss = w * w / s.sample_pdf;
        if (loss > kLossThreshold || std::isnan(loss)) loss = kLossThreshold;
        for (int i = s.begin_ind; i < s.end_ind; i++) ov = (float)((double)ov + (double)(nodes[i].peak_pdf / Q[nodes[i].label_b]) / 1000.0);
        return loss / ov;
    };
    float thr = 0.0f;
    const int probe = std::min(1000, (int)paths.size());
    {
        std::vector<float> t(probe);
        for (int i = 0; i < probe; i++) t[i] = outlier_value(paths[i]);
        std::sort(t.begin(), t.end());  // NaNs: the reference sorts too; take the last element like h_outler[999]
        thr = probe ? t[probe - 1] : 0.0f;
    }
    for (auto& s : paths)
        if (outlier_value(s) > thr) { s.contri[0] *= 0; s.contri[1] *= 0; s.contri[2] *= 0; }
    f_square.resize(n_samples); pdf0.resize(n_samples); P2N.resize(n_samples);
    for (int id = 0; id < n_samples; id++) {  // construct_optimal_E_data_sample (3124-3145)
        const auto& s = paths[id];
        float w = sum3(s.contri);
        float f = w * w / s.sample_pdf;
        if (f > kLossThreshold || std::isnan(f)) f = kLossThreshold;
        f_square[id] = f; pdf0[id] = s.fix_pdf; P2N[id] = s.begin_ind;
    }
    pdf_peak.resize(m_train_nodes); label_E.resize(m_train_nodes); label_P.resize(m_train_nodes);
    for (int id = 0; id < m_train_nodes; id++) {  // construct_optimal_E_data_node (3147-3171)
        const auto& s = nodes[id];
        label_E[id] = s.label_a * NS + s.label_b;
        label_P[id] = s.path_id;
        float pk = Q[s.label_b] > 0.0 ? s.peak_pdf / Q[s.label_b] : 0.0f;
        if (std::isnan(pk) || std::isinf(pk)) pk = 0;
        pdf_peak[id] = pk;
    }
}

void Preprocessor::initial_gamma() {  // preprocess_getGamma (627-667)
    gamma.assign((size_t)NS * NS, 0.0f);
    for (const auto& p : paths) {
        const float weight = sum3(p.contri) / p.sample_pdf;
        for (int j = p.begin_ind; j < p.end_ind; j++) {
            const int id = nodes[j].label_a * NS + nodes[j].label_b;
            const float w2 = (float)fmin((double)weight, 10.0);
            gamma[id] += w2;
        }
    }
    for (int i = 0; i < NS; i++) {
        float ws = 0;
        for (int j = 0; j < NS; j++) ws += gamma[(size_t)i * NS + j];
        for (int j = 0; j < NS; j++) {
            gamma[(size_t)i * NS + j] /= ws;
            if (ws <= 1e-10f) gamma[(size_t)i * NS + j] = (float)(1.0 / NS);
        }
    }
}

// ---------------------------------------------------------------- Adam on the row-normalised sigmoid matrix
static inline float sigmoidf_ref(float a) { return (float)(1.0 / (1.0 + (double)expf(-a))); }  // sigmoid<float> (704-712)

void Preprocessor::train(int batch_size, int epochs, float lr) {
    const size_t NP = (size_t)NS * NS;
    std::vector<float> theta(NP), m(NP, 0.0f), v(NP, 0.0f), E(NP), E_sum(NS), dE(NP), dloss(NP), dE_sum(NS);
    for (size_t i = 0; i < NP; i++) theta[i] = (float)(-log(1.0 / (double)gamma[i] - 1));  // inver_sigmoid (714-722)
    const float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
    int t_step = 0;
    const int num_batches = batch_size > 0 ? n_train_paths / batch_size : 0;
    std::vector<float> pdfs_p(batch_size), d_pdfs(batch_size);
    auto forward_E = [&]() {  // get_E (1149-1173)
        for (int r = 0; r < NS; r++) {
            float s = 0;
            for (int c = 0; c < NS; c++) { float sg = sigmoidf_ref(theta[(size_t)r * NS + c]); E[(size_t)r * NS + c] = sg; s += sg; }
            E_sum[r] = s;
            for (int c = 0; c < NS; c++) {
                float e = E[(size_t)r * NS + c] / s;
                e = e * (float)(1 - 0.2);
                e = e + (float)(0.2 / (float)NS);
                E[(size_t)r * NS + c] = e;
            }
        }
    };
    for (int epoch = 0; epoch < epochs; epoch++) {
        for (int batch = 0; batch < num_batches; batch++) {
            const int bias_sample = batch * batch_size;
            const int bias_node = P2N[bias_sample];
            // == the reference's (batch == num_batches-1 ? M - bias : P2N[next] - bias) whenever N is a multiple of the batch size
            // (it always is there: 2,000,000 / 20,000); otherwise the batch covers exactly its own paths' nodes
            const int seg_end = (bias_sample + batch_size < n_train_paths) ? P2N[bias_sample + batch_size] : m_train_nodes;
            const int seg_nodes = seg_end - bias_node;
            forward_E();
            // get_forward_pdfs + get_loss_gradient (981-1029): per-path sum of peak * E over its nodes, + pdf0
            std::fill(pdfs_p.begin(), pdfs_p.end(), 0.0f);
            for (int k = 0; k < seg_nodes; k++) {
                const int nd = bias_node + k;
                pdfs_p[label_P[nd] % batch_size] += pdf_peak[nd] * E[label_E[nd]];
            }
            double loss_acc = 0;
            for (int s = 0; s < batch_size; s++) {
                pdfs_p[s] += pdf0[bias_sample + s];
                d_pdfs[s] = -f_square[bias_sample + s] / pdfs_p[s] / pdfs_p[s];
                loss_acc += f_square[bias_sample + s] / pdfs_p[s];
            }
            last_mean_loss = (float)(loss_acc / batch_size);
            // get_dE (1045-1089)
            std::fill(dE.begin(), dE.end(), 0.0f);
            for (int k = 0; k < seg_nodes; k++) {
                const int nd = bias_node + k;
                dE[label_E[nd]] += pdf_peak[nd] * d_pdfs[label_P[nd] % batch_size];
            }
            // gradient_E2theta (1090-1148): E is the conservative-mixed matrix, as in the reference
            for (int r = 0; r < NS; r++) {
                const float S = E_sum[r];
                float acc = 0;
                for (int c = 0; c < NS; c++) {
                    const size_t i = (size_t)r * NS + c;
                    const float value = E[i] * S;                // inver_gradient_res
                    acc += (-value / S / S) * dE[i];
                }
                dE_sum[r] = acc;
                for (int c = 0; c < NS; c++) {
                    const size_t i = (size_t)r * NS + c;
                    const float sg = sigmoidf_ref(theta[i]);      // sigmoid_gradient_theta
                    float g = sg * (1 - sg) * acc;
                    const float sgm = E[i] * S;                   // theta_gradient
                    g += (sgm * (1 - sgm) / S) * dE[i];
                    dloss[i] = g;
                }
            }
            // Adam (1438-1477)
            t_step += 1;
            const float b1t = 1 - powf(beta1, (float)t_step), b2t = 1 - powf(beta2, (float)t_step);
            for (size_t i = 0; i < NP; i++) {
                const float g = dloss[i];
                m[i] = beta1 * m[i] + (1 - beta1) * g;
                v[i] = beta2 * v[i] + (1 - beta2) * (g * g);
                const float m_hat = m[i] / b1t, v_hat = v[i] / b2t;
                const float step = m_hat / (sqrtf(v_hat) + eps);
                if (!std::isnan(step)) theta[i] -= lr * step;
            }
        }
    }
    // toE (1579-1599): row-normalised sigmoid without the conservative mix
    for (int r = 0; r < NS; r++) {
        float s = 0;
        for (int c = 0; c < NS; c++) { float sg = sigmoidf_ref(theta[(size_t)r * NS + c]); gamma[(size_t)r * NS + c] = sg; s += sg; }
        for (int c = 0; c < NS; c++) gamma[(size_t)r * NS + c] /= s;
    }
}

void Preprocessor::make_cmf() {  // Gamma2CMFGamma (3406-3433)
    cmf_gamma = gamma;
    const float t = 0.2f;
    for (size_t i = 0; i < cmf_gamma.size(); i++) cmf_gamma[i] = (float)((double)(cmf_gamma[i] * (1 - t)) + (1.0 / NS) * (double)t);
    for (int i = 0; i < NS; i++) {
        for (int j = 1; j < NS; j++) cmf_gamma[(size_t)i * NS + j] += cmf_gamma[(size_t)i * NS + j - 1];
        cmf_gamma[(size_t)(i + 1) * NS - 1] = 1;
    }
}

}  // namespace spc

// ORACLE — test infrastructure only (see vec.h).
// CPU restatement of the reference's preprocessing (optixPathTracer.cpp:552-608), step by step in the reference's own
// structure: __raygen__TrainData (raygen.cu:708-868) with TrainData::nVertex_device (cuProg.h:1128-1292),
// valid_sample_gather / sample_reweight / get_weighted_point_for_tree_building / preprocess_getQ / node_label /
// build_optimal_E_train_data / preprocess_getGamma / train_optimal_E / Gamma2CMFGamma (cuda_thrust/device_thrust.cu) and
// classTree::buildTreeBaseOnExistSample (decisionTree/classTree_host.h).  PARITY UNPINNED (no reference output exists).
// Deliberate deviations, shared with the product and listed in DESIGN.md: q6 (light-source nodes are skipped when
// collecting light-tree samples instead of pushing an uninitialised record), q8 (reweight tile pitch = ceil(width/10)).
#pragma once
#include <algorithm>
#include <cfloat>
#include <vector>

#include "spcbpt_ref.h"

namespace orc {

// ---- TrainData::nVertex / nVertex_device (optixPathTracer.h:264-324, cuProg.h:1128-1292) ----
struct nVertex {
    float3 position{}, dir{}, normal{}, weight{}, color{};
    float pdf = 0;
    int materialId = 0, label_id = 0, depth = 0;
    bool valid = false;
    bool isLightSource() const { return materialId < 0; }
    bool isAreaLight() const { return materialId == -1; }
    nVertex() {}
    nVertex(const BDPTVertex& a, bool eye_side)  // optixPathTracer.h:311-322
        : position(a.position), normal(a.normal), color(a.color), pdf(a.pdf), materialId(a.materialId), label_id(a.subspaceId),
          depth(a.depth), valid(true) {
        dir = a.depth == 0 ? make_float3(0.0f) : normalize(a.lastPosition - a.position);
        weight = eye_side ? make_float3(pdf) : a.flux;
        if (eye_side == false && a.depth == 0 && a.type == QUAD) materialId = -1;
    }
};
struct nVertex_device : nVertex {
    nVertex_device() {}
    nVertex_device(const BDPTVertex& a, bool eye_side) : nVertex(a, eye_side) {}
    float forward_light_pdf(const Params& P, const nVertex& b) const {  // cuProg.h:1193-1218
        float3 vec = b.position - position;
        float3 c_dir = normalize(vec);
        float g = fabsf(dot(c_dir, b.normal)) / dot(vec, vec);
        if (isLightSource()) {
            g *= fabsf(dot(normal, c_dir));
            return (float)((double)(pdf * g) * 1.0 / 3.14159265358979323846);
        }
        Pbr mat = P.scene->materials[materialId];
        mat.base_color = color;
        float d_pdf = Pdf(mat, normal, dir, c_dir);
        float RR_rate = (float)std::max((double)fmaxf3(color), 0.3);
        return pdf * d_pdf * RR_rate * g;
    }
    float3 forward_eye(const Params& P, const nVertex& b) const {  // cuProg.h:1220-1243
        float3 vec = b.position - position;
        float3 c_dir = normalize(vec);
        float g = fabsf(dot(c_dir, b.normal)) / dot(vec, vec);
        Pbr mat = P.scene->materials[materialId];
        mat.base_color = color;
        float d_pdf = Pdf(mat, normal, dir, c_dir);
        float RR_rate = (float)std::max((double)fmaxf3(color), 0.3);
        return weight * d_pdf * RR_rate * g;
    }
    float3 forward_light(const Params& P, const nVertex& b) const {  // cuProg.h:1245-1282
        float3 vec = b.position - position;
        float3 c_dir = normalize(vec);
        float g = fabsf(dot(c_dir, b.normal)) * fabsf(dot(c_dir, normal)) / dot(vec, vec);
        if (isAreaLight()) return weight * g;
        Pbr mat = P.scene->materials[materialId];
        mat.base_color = color;
        float3 d_contri = Eval(mat, normal, dir, c_dir);
        return weight * g * d_contri;
    }
    float3 local_contri(const Params& P, const nVertex_device& b) const {  // cuProg.h:1283-1290
        float3 c_dir = normalize(b.position - position);
        Pbr mat = P.scene->materials[materialId];
        mat.base_color = color;
        return Eval(mat, normal, dir, c_dir);
    }
    nVertex_device(const Params& P, const nVertex& a, const nVertex_device& b, bool eye_side) {  // cuProg.h:1130-1148
        position = a.position;
        dir = normalize(b.position - a.position);
        normal = a.normal;
        weight = eye_side ? b.forward_eye(P, a) : b.forward_light(P, a);
        pdf = eye_side ? weight.x : b.forward_light_pdf(P, a);
        color = a.color; materialId = a.materialId; valid = true; label_id = a.label_id;
        depth = b.depth + 1;
    }
};

typedef spcbpt_pretrace_path preTracePath;        // TrainData::pathInfo_sample
typedef spcbpt_pretrace_node preTraceConnection;  // TrainData::pathInfo_node

inline void store3(float* d, float3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

inline preTraceConnection make_conn(nVertex& a, nVertex& b) {  // pathInfo_node(nVertex&, nVertex&) optixPathTracer.h:343-352
    preTraceConnection c{};
    store3(c.a_position, a.position); store3(c.b_position, b.position);
    store3(c.a_dir, a.dir); store3(c.b_dir, b.dir);
    store3(c.a_normal, a.normal); store3(c.b_normal, b.normal);
    c.valid = 1; c.light_source = b.isLightSource() ? 1 : 0; c.label_b = b.label_id;
    c.peak_pdf = a.weight.x * float3weight(b.weight);
    c.label_a = a.depth;  // set_eye_depth
    return c;
}

// PreTrace_buildPathInfo (raygen.cu:708-740)
inline void PreTrace_buildPathInfo(const Params& P, BDPTVertex* eye, nVertex_device light, preTracePath* path, preTraceConnection* conn,
                                   int pathSize) {
    path->valid = 1;
    path->begin_ind = 0;
    path->end_ind = pathSize - 1;
    path->sample_pdf = 0;
    nVertex_device n_eye(*eye, true);
    nVertex_device n_next_eye(P, light, n_eye, true);
    float3 seg_contri = n_eye.local_contri(P, light);
    path->sample_pdf = n_next_eye.pdf;
    path->sample_pdf += n_eye.pdf * light.pdf;
    path->fix_pdf = n_next_eye.pdf;
    float3 contri = eye->flux * light.forward_light(P, n_eye) * seg_contri;
    for (int i = 0; i < path->end_ind; i++) {
        conn[path->end_ind - i - 1] = make_conn(n_eye, light);
        eye--;
        light = nVertex_device(P, n_eye, light, false);
        n_eye = nVertex_device(*eye, true);
    }
    float weight = float3weight(contri) / path->sample_pdf;
    if (std::isnan(weight)) contri = make_float3(0);
    if (std::isinf(weight)) contri = make_float3(0);
    store3(path->contri, contri);
}

inline bool rr_acc_accept(int acc_num, uint32_t& seed) {  // raygen.cu:741-749
    float r = rnd(seed);
    return 1.0f / (acc_num + 1) > r;
}

// __raygen__TrainData (raygen.cu:751-868)
inline void raygen_TrainData(const Params& P, int launch_index, int iteration, int padding, preTracePath* paths, preTraceConnection* conns) {
    const Scene& S = *P.scene;
    uint32_t seed = tea<4>((uint32_t)launch_index, (uint32_t)iteration);
    const float jx = rnd(seed), jy = rnd(seed);
    const float dx = 2.0f * jx - 1.0f, dy = 2.0f * jy - 1.0f;
    float3 ray_direction = normalize(dx * P.U + dy * P.V + P.W);
    float3 ray_origin = P.eye;
    BDPTVertex buffer[10];  // PRETRACE_CONN_PADDING
    int buffer_size = 0, resample_number = 0;
    PayloadBDPTVertex payload;
    payload.clear();
    payload.seed = seed;
    init_EyeSubpath(payload.path, ray_origin, ray_direction);
    unsigned bufferBias = (unsigned)launch_index * (unsigned)padding;
    preTracePath* currentPath = paths + launch_index;
    preTraceConnection* currentConn = conns + bufferBias;
    *currentPath = preTracePath{};
    currentPath->valid = 0;
    buffer[buffer_size++] = payload.path.currentVertex();
    while (true) {
        int begin_depth = payload.path.size;
        trace_subpath(P, ray_origin, ray_direction, &payload, false);
        if (payload.path.size == begin_depth) break;
        if (payload.path.hit_lightSource()) {
            if (payload.path.size > 2 && rr_acc_accept(resample_number, payload.seed)) {
                lightSample light_sample;
                int light_id = payload.path.currentVertex().materialId;
                light_sample.ReverseSample(P, S.lights[light_id], payload.path.currentVertex().uv);
                BDPTVertex light_vertex;
                init_vertex_from_lightSample(light_sample, light_vertex);
                PreTrace_buildPathInfo(P, buffer + buffer_size - 1, nVertex_device(light_vertex, false), currentPath, currentConn, buffer_size);
                resample_number++;
            }
            break;
        }
        buffer[buffer_size++] = payload.path.currentVertex();
        BDPTVertex& eye_subpath = payload.path.currentVertex();
        lightSample light_sample;
        light_sample.sample_quad(P, payload.seed);
        float3 vis_vec = light_sample.position - eye_subpath.position;
        BDPTVertex light_vertex;
        init_vertex_from_lightSample(light_sample, light_vertex);
        if (S.visibilityTest(eye_subpath.position, light_vertex.position, P.counters) && rr_acc_accept(resample_number, payload.seed)) {
            if (dot(vis_vec, light_sample.normal()) < 0) {
                PreTrace_buildPathInfo(P, buffer + buffer_size - 1, nVertex_device(light_vertex, false), currentPath, currentConn, buffer_size);
                resample_number++;
            }
        }
        if (payload.done || payload.depth > 50) break;
        if (buffer_size >= padding) break;  // PRETRACER_PADDING_VERTICES_CHECK
        ray_direction = payload.ray_direction;
        ray_origin = payload.origin;
        payload.depth += 1;
    }
    int beginIndex = 0;
    if (currentPath->valid) beginIndex += currentPath->end_ind - currentPath->begin_ind;
    for (int i = beginIndex; i < padding; i++) currentConn[i].valid = 0;
    currentPath->sample_pdf /= resample_number;
    currentPath->begin_ind += bufferBias;
    currentPath->end_ind += bufferBias;
    currentPath->pixel_id[0] = (int)(P.width * jx);
    currentPath->pixel_id[1] = (int)(P.height * jy);
    if (currentPath->begin_ind == currentPath->end_ind && currentPath->valid) currentPath->valid = 0;
}

// ---- MyThrustOp state (device_thrust.cu) ----
struct divide_weight { float3 position, dir, normal; float weight; };  // classTree_common.h:76-91
struct divide_weight_with_label : divide_weight { int label; };

struct PreState {
    std::vector<preTracePath> neat_paths;
    std::vector<preTraceConnection> neat_conns;
    std::vector<float> h_Q_vec;
    long long acc_valid_path = 0;
    std::vector<float> h_Gamma, CMFGamma;
    std::vector<tree_node> eye_tree, light_tree;
    // E_td
    std::vector<float> b_f_square, b_pdf0, b_pdf_peak;
    std::vector<int> b_label_E, b_label_P, b_P2N_ind;
    int N_path = 0, M_node = 0;
};

// valid_sample_gather (457-493): stream compaction of valid paths / connections, begin/end re-based, path ids assigned
inline int valid_sample_gather(PreState& st, const std::vector<preTracePath>& raw_paths, const std::vector<preTraceConnection>& raw_conns) {
    const int acc_num_samples = (int)st.neat_paths.size(), acc_num_nodes = (int)st.neat_conns.size();
    std::vector<int> sample_bias_flag(raw_conns.size());  // exclusive scan of the valid flags
    int run = 0;
    for (size_t i = 0; i < raw_conns.size(); i++) { sample_bias_flag[i] = run; run += raw_conns[i].valid ? 1 : 0; }
    int sample_count = 0;
    for (const auto& p : raw_paths) if (p.valid) { st.neat_paths.push_back(p); sample_count++; }
    for (const auto& c : raw_conns) if (c.valid) st.neat_conns.push_back(c);
    for (int id = 0; id < sample_count; id++) {  // bias_arrange_op
        preTracePath& s = st.neat_paths[acc_num_samples + id];
        int bias = s.begin_ind - sample_bias_flag[s.begin_ind];
        s.begin_ind += acc_num_nodes - bias;
        s.end_ind += acc_num_nodes - bias;
        for (int i = s.begin_ind; i < s.end_ind; i++) st.neat_conns[i].path_id = id + acc_num_samples;
    }
    return sample_count;
}

inline void sample_reweight(PreState& st, int width, int height) {  // 574-623 (tile pitch generalised: q8)
    const int pitch = (width + 9) / 10, rows = (height + 9) / 10;
    std::vector<float> weight((size_t)pitch * rows + pitch, 0.0f);
    auto nid = [&](const preTracePath& s) {
        int n_id = s.pixel_id[0] / 10 + (s.pixel_id[1] / 10) * pitch;
        return std::min(std::max(n_id, 0), (int)weight.size() - 1);
    };
    for (auto& s : st.neat_paths) {
        float ww = (s.contri[0] + s.contri[1] + s.contri[2]) / s.sample_pdf;
        if (std::isnan(ww) || std::isinf(ww)) continue;
        weight[nid(s)] += ww;
    }
    for (auto& s : st.neat_paths) {
        float w = (float)((double)(weight[nid(s)] / 100) + 0.1);
        float3 c = make_float3(s.contri[0], s.contri[1], s.contri[2]) / w;
        store3(s.contri, c);
    }
}

inline std::vector<divide_weight> get_weighted_point_for_tree_building(const PreState& st, bool eye_side, int max_size) {  // 494-527
    std::vector<divide_weight> ans;
    int n = (int)st.neat_paths.size();
    int sizeLimit = max_size == 0 ? n : (n > max_size ? max_size : n);
    for (int i = 0; i < sizeLimit; i++)
        for (int j = st.neat_paths[i].begin_ind; j < st.neat_paths[i].end_ind; j++) {
            const preTraceConnection& c = st.neat_conns[j];
            divide_weight t;
            const preTracePath& p = st.neat_paths[i];
            if (eye_side) {
                t.dir = load3(c.a_dir); t.normal = load3(c.a_normal); t.position = load3(c.a_position);
            } else if (c.light_source == 0) {
                t.dir = load3(c.b_dir); t.normal = load3(c.b_normal); t.position = load3(c.b_position);
            } else {
                continue;  // q6
            }
            t.weight = (p.contri[0] + p.contri[1] + p.contri[2]) / p.sample_pdf;
            ans.push_back(t);
        }
    return ans;
}

// classTree::buildTreeBaseOnExistSample (classTree_host.h:61-431)
struct buildTreeBaseOnExistSample {
    struct devide_node : tree_node {
        std::vector<divide_weight_with_label> v;
        int depth = 0;
        float weight = 0, correct_weight = 0;
        int father = 0, position_depth = 0, normal_depth = 0, dir_depth = 0;
        devide_node() {
            mid = make_float3(0); for (int& c : child) c = 0;
            leaf = true; label = 0; type = 0;
        }
        void add_sample(const divide_weight_with_label& w) { v.push_back(w); weight += w.weight; }
        bool need_split() const { return v.size() != 0 && correct_weight < weight; }
    };
    std::vector<devide_node> v;
    std::vector<float3> block_size, direction_block_size;
    float3 bbox_min = make_float3(FLT_MAX), bbox_max = make_float3(FLT_MIN);

    void color(int id) {  // 243-284
        devide_node& t = v[id];
        if (t.v.size() == 0) { t.correct_weight = 0.0f; return; }
        bool need_split = false;
        t.label = t.v[0].label;
        for (size_t i = 0; i < t.v.size(); i++) if (t.v[i].label != t.label) { need_split = true; break; }
        if (need_split) {
            std::vector<float> weights(SPCBPT_NUM_SUBSPACE, 0.0f);
            float max_weight = 0.0f;
            int max_weight_id = t.label;
            for (size_t i = 0; i < t.v.size(); i++) {
                weights[t.v[i].label] += t.v[i].weight;
                if (max_weight < weights[t.v[i].label]) { max_weight = weights[t.v[i].label]; max_weight_id = t.v[i].label; }
            }
            t.label = max_weight_id;
            t.correct_weight = max_weight;
        } else {
            t.correct_weight = t.weight;
        }
    }
    float split(int id) {  // 103-211
        int split_type = (v[id].depth % 2 == 0 || v[id].normal_depth > 3) ? 0 : 1;
        int back = (int)v.size();
        v[id].leaf = false;
        float3 inch = split_type == 0 ? block_size[v[id].position_depth + 1] : direction_block_size[v[id].normal_depth + 1];
        float3 mid;
        if (v[id].normal_depth == 0 && split_type == 1) mid = make_float3(0.0f);
        else if (v[id].position_depth == 0) mid = v[id].mid;
        else {
            int L_id = id, t_id = v[id].father;
            while (t_id != 0 && v[t_id].type != split_type) { L_id = t_id; t_id = v[t_id].father; }
            mid = v[t_id].mid;
            int c_id_local = 0;
            for (; c_id_local < 8; c_id_local++) if (v[t_id].child[c_id_local] == L_id) break;
            float3 delta_mid = make_float3((c_id_local >> 0) % 2 == 0 ? -inch.x : inch.x, (c_id_local >> 1) % 2 == 0 ? -inch.y : inch.y,
                                           (c_id_local >> 2) % 2 == 0 ? -inch.z : inch.z);
            mid += delta_mid;
        }
        v[id].mid = mid;
        v[id].type = split_type;
        for (int i = 0; i < 8; i++) {
            v[id].child[i] = back + i;
            v.push_back(devide_node());
            v.back().father = id;
            v.back().depth = v[id].depth + 1;
            v.back().label = v[id].label;
            v.back().position_depth = v[id].position_depth + (split_type == 0);
            v.back().normal_depth = v[id].normal_depth + (split_type == 1);
            v.back().dir_depth = v[id].dir_depth;
        }
        for (size_t k = 0; k < v[id].v.size(); k++) {
            const divide_weight_with_label& p = v[id].v[k];
            v[v[id](p.position, p.normal, p.dir)].add_sample(p);
        }
        float n_correct_weight = 0.0f;
        for (int i = 0; i < 8; i++) {
            color(v[id].child[i]);
            n_correct_weight += v[v[id].child[i]].correct_weight;
        }
        v[id].weight = 0;
        v[id].v.clear();
        return n_correct_weight;
    }
    std::vector<tree_node> run(std::vector<divide_weight_with_label>& samples, float threshold, int max_depth = 15) {  // 344-429
        float unnorm = 0.0f;  // para_initial 213-241
        for (auto& p : samples) {
            unnorm += p.weight;
            bbox_min = make_float3(fminf(bbox_min.x, p.position.x), fminf(bbox_min.y, p.position.y), fminf(bbox_min.z, p.position.z));
            bbox_max = make_float3(fmaxf(bbox_max.x, p.position.x), fmaxf(bbox_max.y, p.position.y), fmaxf(bbox_max.z, p.position.z));
        }
        for (auto& p : samples) p.weight /= unnorm;
        float3 bbox_block = bbox_max - bbox_min;
        for (int i = 0; i < max_depth + 10; i++) { block_size.push_back(bbox_block); bbox_block /= 2; }
        float3 direction_block = make_float3(2.0f);
        for (int i = 0; i < 15; i++) { direction_block_size.push_back(direction_block); direction_block /= 2; }
        v.push_back(devide_node());
        v[0].v = samples;
        v[0].weight = 1;
        v[0].mid = (bbox_max + bbox_min) / 2;
        color(0);
        float c_w = v[0].correct_weight;
        for (size_t i = 0; i < v.size(); i++)
            if (v[i].need_split() && v[i].depth < max_depth && threshold > c_w) {
                c_w -= v[i].correct_weight;
                c_w += split((int)i);
            }
        std::vector<tree_node> out(v.size());
        for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
        return out;
    }
    std::vector<tree_node> operator()(std::vector<divide_weight>& samples, int subspaceSize, int labelBias = 0) {  // 302-342
        if (samples.size() < 2) {
            devide_node leaf;
            leaf.label = labelBias;
            return std::vector<tree_node>(1, leaf);
        }
        // get_position_variance 286-301
        int it = (int)samples.size();
        float3 mean = make_float3(0.0f);
        for (int i = 0; i < it; i++) mean += samples[i].position / (float)it;
        float3 var = make_float3(0.0f);
        for (int i = 0; i < it; i++) { float3 diff = mean - samples[i].position; var += diff * diff / (float)(it - 1); }
        float scene_diversity2 = std::max(var.x, std::max(var.y, var.z));
        float weight_sum = 0;
        for (auto& p : samples) weight_sum += p.weight;
        std::vector<divide_weight> centers;
        float acc = 0;
        for (auto& p : samples) {
            acc += p.weight;
            if (acc > weight_sum / subspaceSize) { acc -= weight_sum / subspaceSize; centers.push_back(p); }
        }
        std::vector<divide_weight_with_label> labeled;
        labeled.reserve(samples.size());
        for (auto& p : samples) {
            float min_distance = FLT_MAX;
            int subspaceId = 0;
            for (size_t i = 0; i < centers.size(); i++) {
                const divide_weight& a = centers[i];  // divide_weight::d(a, diag2) classTree_common.h:82-90, DIR_JUDGE = 0
                float3 diff = a.position - p.position;
                float d_a = dot(diff, diff);
                float diff_direction = dot(p.dir, a.dir);
                float diff_normal = dot(p.normal, a.normal);
                float d = d_a + scene_diversity2 * ((1 - diff_normal) + (1 - diff_direction) * 0.0f);
                if (d < min_distance) { min_distance = d; subspaceId = (int)i + labelBias; }
            }
            divide_weight_with_label t;
            t.dir = p.dir; t.normal = p.normal; t.position = p.position; t.weight = p.weight; t.label = subspaceId;
            labeled.push_back(t);
        }
        return run(labeled, 0.99f);
    }
};

// preprocess_getQ (347-409) on the oracle's LVC
inline void preprocess_getQ(PreState& st, const LightTraceParams& lt) {
    if (st.h_Q_vec.empty()) { st.acc_valid_path = 0; st.h_Q_vec.assign(SPCBPT_NUM_SUBSPACE, 0.0f); }
    std::vector<float> tmp_Q_vec(SPCBPT_NUM_SUBSPACE, 0.0f);
    int path_count = 0;
    for (int i = 0; i < lt.get_element_count(); i++) {
        if (!lt.validState[i]) continue;
        const BDPTVertex& v = lt.ans[i];
        if (v.depth == 0) path_count++;
        float res = float3weight(v.flux) / v.pdf;
        res = std::isinf(res) ? 0 : res;
        tmp_Q_vec[v.subspaceId] += std::isnan(res) ? 0 : res;
    }
    st.acc_valid_path += path_count;
    float t = path_count / (float)(st.acc_valid_path);
    for (int i = 0; i < SPCBPT_NUM_SUBSPACE; i++) {
        tmp_Q_vec[i] /= path_count;
        st.h_Q_vec[i] = st.h_Q_vec[i] * (1 - t) + tmp_Q_vec[i] * t;
    }
}
inline void Q_zero_handle(PreState& st) {  // 335-346
    for (int i = 0; i < SPCBPT_NUM_SUBSPACE; i++) if (st.h_Q_vec[i] == 0) st.h_Q_vec[i] = FLT_MAX;
}
inline void node_label(PreState& st) {  // 554-573
    for (auto& s : st.neat_conns) {
        s.label_a = tree_index(st.eye_tree.data(), load3(s.a_position), load3(s.a_normal), load3(s.a_dir), nullptr);
        if (!s.light_source) s.label_b = tree_index(st.light_tree.data(), load3(s.b_position), load3(s.b_normal), load3(s.b_dir), nullptr);
    }
}

static const float optimal_E_loss_threshold = 1000000.0f;  // 3097
inline float outler_value_of(const PreState& st, const preTracePath& s) {  // get_outler_value 3172-3197
    float outler_value = s.fix_pdf;
    float weight = s.contri[0] + s.contri[1] + s.contri[2];
    float loss = weight * weight / s.sample_pdf;
    if (loss > optimal_E_loss_threshold || std::isnan(loss)) loss = optimal_E_loss_threshold;
    for (int i = s.begin_ind; i < s.end_ind; i++)
        outler_value = (float)((double)outler_value + (double)(st.neat_conns[i].peak_pdf / st.h_Q_vec[st.neat_conns[i].label_b]) / 1000.0);
    return loss / outler_value;
}
inline void build_optimal_E_train_data(PreState& st, int N_samples) {  // 3261-3325
    N_samples = std::min(N_samples, (int)st.neat_paths.size());
    int M_nodes = N_samples > 0 ? st.neat_paths[N_samples - 1].end_ind : 0;
    int probe = std::min(1000, (int)st.neat_paths.size());
    std::vector<float> t_outler(probe);
    for (int i = 0; i < probe; i++) t_outler[i] = outler_value_of(st, st.neat_paths[i]);
    std::sort(t_outler.begin(), t_outler.end());
    float outler_value = probe ? t_outler[probe - 1] : 0.0f;
    for (auto& s : st.neat_paths)  // clean_outler_value
        if (outler_value_of(st, s) > outler_value) { s.contri[0] *= 0; s.contri[1] *= 0; s.contri[2] *= 0; }
    st.b_f_square.resize(N_samples); st.b_pdf0.resize(N_samples); st.b_P2N_ind.resize(N_samples);
    for (int id = 0; id < N_samples; id++) {  // construct_optimal_E_data_sample
        const preTracePath& s = st.neat_paths[id];
        float weight = s.contri[0] + s.contri[1] + s.contri[2];
        float f = weight * weight / s.sample_pdf;
        if (f > optimal_E_loss_threshold || std::isnan(f)) f = optimal_E_loss_threshold;
        st.b_f_square[id] = f; st.b_pdf0[id] = s.fix_pdf; st.b_P2N_ind[id] = s.begin_ind;
    }
    st.b_pdf_peak.resize(M_nodes); st.b_label_E.resize(M_nodes); st.b_label_P.resize(M_nodes);
    for (int id = 0; id < M_nodes; id++) {  // construct_optimal_E_data_node
        const preTraceConnection& s = st.neat_conns[id];
        st.b_label_E[id] = s.label_a * SPCBPT_NUM_SUBSPACE + s.label_b;
        st.b_label_P[id] = s.path_id;
        float q = st.h_Q_vec[s.label_b];
        float pk = q > 0.0 ? s.peak_pdf / q : 0.0f;
        if (std::isnan(pk) || std::isinf(pk)) pk = 0;
        st.b_pdf_peak[id] = pk;
    }
    st.N_path = N_samples; st.M_node = M_nodes;
}
inline void preprocess_getGamma(PreState& st) {  // 627-667
    const int NS = SPCBPT_NUM_SUBSPACE;
    st.h_Gamma.assign((size_t)NS * NS, 0.0f);
    for (const auto& p : st.neat_paths) {
        float weight = (p.contri[0] + p.contri[1] + p.contri[2]) / p.sample_pdf;
        for (int j = p.begin_ind; j < p.end_ind; j++) {
            int GammaId = st.neat_conns[j].label_a * NS + st.neat_conns[j].label_b;
            float weight2 = (float)fmin((double)weight, 10.0);
            st.h_Gamma[GammaId] += weight2;
        }
    }
    for (int i = 0; i < NS; i++) {
        float weightS = 0;
        for (int j = 0; j < NS; j++) weightS += st.h_Gamma[(size_t)i * NS + j];
        for (int j = 0; j < NS; j++) {
            st.h_Gamma[(size_t)i * NS + j] /= weightS;
            if (weightS <= 1e-10f) st.h_Gamma[(size_t)i * NS + j] = (float)(1.0 / NS);
        }
    }
}

// train_optimal_E (3327-3344) = matrix_parameter::fit (1615-1655), matrix_optimal_operator (923-1228), adam_step_func (1438-1477).
// Each thrust call is one loop; reductions run in index order (thrust leaves the order unspecified).
inline float sigmoid_f(float a) { return (float)(1.0 / (1.0 + (double)expf(-a))); }
inline void train_optimal_E(PreState& st, int batch_size, int epoches, float lr) {
    const int NS = SPCBPT_NUM_SUBSPACE;
    const size_t num_paras = (size_t)NS * NS;
    std::vector<float> data(num_paras), m(num_paras, 0.0f), v(num_paras, 0.0f);
    for (size_t i = 0; i < num_paras; i++) data[i] = (float)(-log(1.0 / (double)st.h_Gamma[i] - 1));  // initial_with_inver_sigmoid
    int t = 0;
    const float beta1 = 0.9f, beta2 = 0.999f, epsilon = 1e-8f;
    const int num_samples = st.N_path, num_nodes = st.M_node;
    const int num_batches = batch_size > 0 ? num_samples / batch_size : 0;
    std::vector<float> E(num_paras), E_sum(NS), pdfs_strategy, pdfs_p(batch_size), d_pdfs(batch_size), d_E_each, dE(num_paras), dEdSum(num_paras),
        dE_sum(NS), dloss_dtheta(num_paras), de_dtheta_0(num_paras);
    for (int epoch = 0; epoch < epoches; epoch++)
        for (int batch = 0; batch < num_batches; batch++) {
            int bias_sample = batch * batch_size;
            int bias_node = st.b_P2N_ind[bias_sample];
            int seg_end = (bias_sample + batch_size < num_samples) ? st.b_P2N_ind[bias_sample + batch_size] : num_nodes;
            int seg_nodes = seg_end - bias_node;
            const float* pdf0 = st.b_pdf0.data() + bias_sample;
            const float* loss_weight = st.b_f_square.data() + bias_sample;
            const float* peak_pdf = st.b_pdf_peak.data() + bias_node;
            const int* label_E = st.b_label_E.data() + bias_node;
            const int* label_P = st.b_label_P.data() + bias_node;
            // get_E
            for (size_t i = 0; i < num_paras; i++) E[i] = sigmoid_f(data[i]);
            for (int r = 0; r < NS; r++) { float s = 0; for (int c = 0; c < NS; c++) s += E[(size_t)r * NS + c]; E_sum[r] = s; }
            for (size_t i = 0; i < num_paras; i++) E[i] = E[i] / E_sum[i / NS];
            for (size_t i = 0; i < num_paras; i++) E[i] = E[i] * (float)(1 - 0.2);
            for (size_t i = 0; i < num_paras; i++) E[i] = E[i] + (float)(0.2 / (float)NS);
            // get_forward_pdfs
            pdfs_strategy.resize(seg_nodes);
            for (int k = 0; k < seg_nodes; k++) pdfs_strategy[k] = peak_pdf[k] * E[label_E[k]];
            // get_loss_gradient: reduce_by_key over runs of label_P, + pdf0, inver_gradient
            std::fill(pdfs_p.begin(), pdfs_p.end(), 0.0f);
            {
                int run = -1, prev = -1;
                for (int k = 0; k < seg_nodes; k++) {
                    if (label_P[k] != prev) { run++; prev = label_P[k]; }
                    if (run < batch_size) pdfs_p[run] += pdfs_strategy[k];
                }
            }
            for (int s = 0; s < batch_size; s++) pdfs_p[s] = pdfs_p[s] + pdf0[s];
            for (int s = 0; s < batch_size; s++) d_pdfs[s] = -loss_weight[s] / pdfs_p[s] / pdfs_p[s];
            // get_dE
            d_E_each.resize(seg_nodes);
            for (int k = 0; k < seg_nodes; k++) d_E_each[k] = peak_pdf[k] * d_pdfs[label_P[k] % batch_size];
            std::fill(dE.begin(), dE.end(), 0.0f);
            for (int k = 0; k < seg_nodes; k++) dE[label_E[k]] += d_E_each[k];
            // gradient_E2theta
            for (size_t i = 0; i < num_paras; i++) { float den = E_sum[i / NS]; float value = E[i] * den; dEdSum[i] = -value / den / den; }
            for (size_t i = 0; i < num_paras; i++) dEdSum[i] = dEdSum[i] * dE[i];
            for (int r = 0; r < NS; r++) { float s = 0; for (int c = 0; c < NS; c++) s += dEdSum[(size_t)r * NS + c]; dE_sum[r] = s; }
            for (size_t i = 0; i < num_paras; i++) { float sig = sigmoid_f(data[i]); dloss_dtheta[i] = sig * (1 - sig); }
            for (size_t i = 0; i < num_paras; i++) dloss_dtheta[i] = dloss_dtheta[i] * dE_sum[i / NS];
            for (size_t i = 0; i < num_paras; i++) { float ss = E_sum[i / NS]; float sg = E[i] * ss; de_dtheta_0[i] = sg * (1 - sg) / ss; }
            for (size_t i = 0; i < num_paras; i++) de_dtheta_0[i] = de_dtheta_0[i] * dE[i];
            for (size_t i = 0; i < num_paras; i++) dloss_dtheta[i] = dloss_dtheta[i] + de_dtheta_0[i];
            // minimize
            t += 1;
            for (size_t id = 0; id < num_paras; id++) {
                float theta = data[id];
                float g = dloss_dtheta[id];
                m[id] = beta1 * m[id] + (1 - beta1) * g;
                v[id] = beta2 * v[id] + (1 - beta2) * (g * g);
                float m_hat = m[id] / (1 - powf(beta1, (float)t));
                float v_hat = v[id] / (1 - powf(beta2, (float)t));
                if (!std::isnan(m_hat / (sqrtf(v_hat) + epsilon))) theta -= lr * m_hat / (sqrtf(v_hat) + epsilon);
                data[id] = theta;
            }
        }
    // toE
    for (size_t i = 0; i < num_paras; i++) st.h_Gamma[i] = sigmoid_f(data[i]);
    for (int r = 0; r < NS; r++) {
        float s = 0;
        for (int c = 0; c < NS; c++) s += st.h_Gamma[(size_t)r * NS + c];
        for (int c = 0; c < NS; c++) st.h_Gamma[(size_t)r * NS + c] /= s;
    }
}
inline void Gamma2CMFGamma(PreState& st) {  // 3406-3433
    const int NS = SPCBPT_NUM_SUBSPACE;
    st.CMFGamma = st.h_Gamma;
    for (int i = 0; i < NS; i++)
        for (int j = 0; j < NS; j++) {
            float t = 0.2f;
            st.CMFGamma[(size_t)i * NS + j] = (float)((double)(st.CMFGamma[(size_t)i * NS + j] * (1 - t)) + (1.0 / NS) * (double)t);
        }
    for (int i = 0; i < NS; i++) {
        for (int j = 0; j < NS; j++) {
            int index = i * NS + j;
            if (j != 0) st.CMFGamma[index] += st.CMFGamma[index - 1];
        }
        st.CMFGamma[(size_t)(i + 1) * NS - 1] = 1;
    }
}

}  // namespace orc

// Checkpoint files of the subspace tuple in the reference's own text formats.  The reference ships only
// the READERS -- classTree::tree_load (decisionTree/classTree_host.h:15-59: tree_eye.txt / tree_light.txt),
// MyThrustOp::load_Q_file (cuda_thrust/device_thrust.cu:3389-3404: Q.txt), MyThrustOp::load_Gamma_file (3347-3380: E.txt) --
// with their call sites commented out (optixPathTracer.cpp:573-581, 597, 603); the writers here emit exactly what those
// readers parse, so a tuple trained by this build can be dropped next to the reference's executable and vice versa.
//   tree_*.txt  one node per record: `leaf label`, and for an internal node `type mid.x mid.y mid.z child[0..7]`
//   Q.txt       NUM_SUBSPACE numbers
//   E.txt       Gamma (before the CMF transform), row-major eye x light, NUM_SUBSPACE^2 numbers
// Pure host code: no GPU, no context.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/spcbpt.h"

namespace {

const int NS = SPCBPT_NUM_SUBSPACE;

std::string join(const char* dir, const char* name) {
    std::string d = dir ? dir : ".";
    if (!d.empty() && d.back() != '/') d += '/';
    return d + name;
}

bool write_tree(const std::string& path, const spcbpt_tree_node* t, int n) {
    FILE* f = fopen(path.c_str(), "w");
    if (!f) return false;
    for (int i = 0; i < n; i++) {
        const spcbpt_tree_node& nd = t[i];
        if (nd.leaf) {
            fprintf(f, "1 %d\n", nd.label);
        } else {  // %.9g round-trips an IEEE single exactly
            fprintf(f, "0 %d %d %.9g %.9g %.9g", nd.label, nd.type, nd.mid[0], nd.mid[1], nd.mid[2]);
            for (int c = 0; c < 8; c++) fprintf(f, " %d", nd.child[c]);
            fputc('\n', f);
        }
    }
    return fclose(f) == 0;
}

// classTree::tree_load: `while (inFile >> leaf)` with formatted extraction; fields a leaf does not carry keep the
// default-constructed values of tree_node (classTree_common.h:18-26: mid 0, child 0, type 0 -- spelled out here)
int read_tree(const std::string& path, spcbpt_tree_node* out, int cap) {
    std::ifstream in(path);
    if (!in) return -1;
    int n = 0;
    bool leaf;
    while (in >> leaf) {
        spcbpt_tree_node nd;
        memset(&nd, 0, sizeof(nd));
        in >> nd.label;
        nd.leaf = leaf ? 1 : 0;
        if (!leaf) {
            in >> nd.type >> nd.mid[0] >> nd.mid[1] >> nd.mid[2];
            for (int c = 0; c < 8; c++) in >> nd.child[c];
        }
        if (n >= cap) return -2;
        out[n++] = nd;
    }
    return n;
}

bool write_floats(const std::string& path, const float* v, size_t n, int per_line) {
    FILE* f = fopen(path.c_str(), "w");
    if (!f) return false;
    for (size_t i = 0; i < n; i++) fprintf(f, "%.9g%c", v[i], ((int)((i + 1) % per_line) == 0) ? '\n' : ' ');
    return fclose(f) == 0;
}

}  // namespace

extern "C" {

int spcbpt_checkpoint_write(const char* dir, const spcbpt_tree_node* eye_tree, int n_eye, const spcbpt_tree_node* light_tree,
                            int n_light, const float* q, const float* gamma) {
    if (!eye_tree || !light_tree || !q || !gamma || n_eye < 1 || n_light < 1) return SPCBPT_ERR_INVALID_ARG;
    if (!write_tree(join(dir, "tree_eye.txt"), eye_tree, n_eye) || !write_tree(join(dir, "tree_light.txt"), light_tree, n_light) ||
        !write_floats(join(dir, "Q.txt"), q, NS, 1) || !write_floats(join(dir, "E.txt"), gamma, (size_t)NS * NS, NS))
        return SPCBPT_ERR_IO;
    return SPCBPT_OK;
}

int spcbpt_checkpoint_read(const char* dir, spcbpt_tree_node* eye_tree, int* n_eye, int cap_eye, spcbpt_tree_node* light_tree,
                           int* n_light, int cap_light, float* q, float* gamma, int have_current_gamma) {
    if (!eye_tree || !light_tree || !n_eye || !n_light || !q || !gamma) return SPCBPT_ERR_INVALID_ARG;
    const int ne = read_tree(join(dir, "tree_eye.txt"), eye_tree, cap_eye);
    const int nl = read_tree(join(dir, "tree_light.txt"), light_tree, cap_light);
    if (ne == -2 || nl == -2) return SPCBPT_ERR_CAPACITY;
    if (ne < 1 || nl < 1) return SPCBPT_ERR_IO;
    *n_eye = ne; *n_light = nl;
    {  // load_Q_file: every number of the file, in order
        std::ifstream in(join(dir, "Q.txt"));
        if (!in) return SPCBPT_ERR_IO;
        float v;
        int k = 0;
        while (in >> v) { if (k < NS) q[k] = v; k++; }
        if (k != NS) return SPCBPT_ERR_IO;
    }
    {  // load_Gamma_file: the columns of the emitter subspaces (light id >= NUM_SUBSPACE - NUM_SUBSPACE_LIGHTSOURCE) keep the
       // CURRENT Gamma -- the file's number is consumed and dropped (3364-3371).  Without a current Gamma (have_current_gamma
       // == 0: nothing was preprocessed in this process) there is nothing to keep and the file's number is used.
        std::ifstream in(join(dir, "E.txt"));
        if (!in) return SPCBPT_ERR_IO;
        float v;
        size_t k = 0;
        const size_t total = (size_t)NS * NS;
        while (in >> v) {
            if (k < total) {
                const int id_light = (int)(k % NS);
                if (id_light < NS - SPCBPT_NUM_SUBSPACE_LIGHTSOURCE || !have_current_gamma) gamma[k] = v;
            }
            k++;
        }
        if (k != total) return SPCBPT_ERR_IO;
    }
    return SPCBPT_OK;
}

// MyThrustOp::Gamma2CMFGamma (device_thrust.cu:3406-3433): conservative mixing with the uniform row (CONSERVATIVE_RATE 0.2),
// fp32 running sum per row, last entry forced to 1.  `(1.0 / NUM_SUBSPACE) * t` is double arithmetic in the reference.
int spcbpt_gamma_to_cmf(const float* gamma, float* cmf_gamma) {
    if (!gamma || !cmf_gamma) return SPCBPT_ERR_INVALID_ARG;
    const float t = 0.2f;
    for (size_t i = 0; i < (size_t)NS * NS; i++) cmf_gamma[i] = (float)((double)(gamma[i] * (1 - t)) + (1.0 / NS) * (double)t);
    for (int i = 0; i < NS; i++) {
        for (int j = 1; j < NS; j++) cmf_gamma[(size_t)i * NS + j] += cmf_gamma[(size_t)i * NS + j - 1];
        cmf_gamma[(size_t)(i + 1) * NS - 1] = 1;
    }
    return SPCBPT_OK;
}

}  // extern "C"

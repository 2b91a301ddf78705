// ORACLE — test infrastructure only (row f2).  The reference's OWN vendored glTF stack, included from where it lies under
// /root/reference/src (never copied): support/tinygltf/{tiny_gltf.h, json.hpp, stb_image*.h} and sutil/{Matrix.h, Quaternion.h}.
// sutil/Scene.cpp itself (loadScene / processGLTFNode) includes <optix.h> and cannot be compiled here; this file walks the
// tinygltf model in the order and with the arithmetic Scene.cpp:119-210, 266-550 uses (root nodes = nodes without a parent,
// node_xform = parent * matrix^T * T * R * S with sutil::Matrix4x4 / Quaternion, buffer views with byteStride, TRIANGLES
// only) and bakes each instance into world space with Matrix4x4 * float4 — the product's csrc/gltf_file.cpp is pinned
// against the result.  Built by `make -C oracle ref` into oracle/_ref/libref_gltf.so (its own .so: tinygltf brings its own
// stb_image implementation).
#define TINYGLTF_IMPLEMENTATION
#define STB_IMAGE_IMPLEMENTATION
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include <cuda_runtime.h>

#include <support/tinygltf/tiny_gltf.h>
#include <sutil/Matrix.h>
#include <sutil/Quaternion.h>

#include <cstring>
#include <string>
#include <vector>

using sutil::Matrix4x4;
using sutil::Quaternion;

namespace {
struct Out {
    std::vector<float> pos, uv;
    std::vector<unsigned> idx;
    std::vector<int> mat;
    float eye[3] = {0, 0, 0}, up[3] = {0, 0, 0}, fov = 0, aspect = 0;
    int cameras = 0;
};
struct Span { const unsigned char* base; size_t stride, count; int comp; };
Span span_of(const tinygltf::Model& m, int accessor) {  // bufferViewFromGLTF (Scene.cpp:82-117)
    const auto& a = m.accessors[accessor];
    const auto& bv = m.bufferViews[a.bufferView];
    const int esz = a.componentType == TINYGLTF_COMPONENT_TYPE_UNSIGNED_SHORT ? 2 : a.componentType == TINYGLTF_COMPONENT_TYPE_UNSIGNED_INT ? 4
                  : a.componentType == TINYGLTF_COMPONENT_TYPE_FLOAT ? 4 : 0;
    Span s;
    s.base = m.buffers[bv.buffer].data.data() + bv.byteOffset + a.byteOffset;
    s.stride = bv.byteStride;
    s.count = a.count;
    s.comp = a.componentType;
    if (s.stride == 0) {
        s.stride = esz;
        switch (a.type) {
            case TINYGLTF_TYPE_VEC2: s.stride *= 2; break;
            case TINYGLTF_TYPE_VEC3: s.stride *= 3; break;
            case TINYGLTF_TYPE_VEC4: s.stride *= 4; break;
            default: break;
        }
    }
    return s;
}
void node(const tinygltf::Model& model, const tinygltf::Node& n, const Matrix4x4& parent, Out& o) {
    const Matrix4x4 translation = n.translation.empty() ? Matrix4x4::identity()
        : Matrix4x4::translate(make_float3((float)n.translation[0], (float)n.translation[1], (float)n.translation[2]));
    const Matrix4x4 rotation = n.rotation.empty() ? Matrix4x4::identity()
        : Quaternion((float)n.rotation[3], (float)n.rotation[0], (float)n.rotation[1], (float)n.rotation[2]).rotationMatrix();
    const Matrix4x4 scale = n.scale.empty() ? Matrix4x4::identity()
        : Matrix4x4::scale(make_float3((float)n.scale[0], (float)n.scale[1], (float)n.scale[2]));
    std::vector<float> gm;
    for (double x : n.matrix) gm.push_back((float)x);
    const Matrix4x4 matrix = n.matrix.empty() ? Matrix4x4::identity() : Matrix4x4(gm.data()).transpose();
    const Matrix4x4 xf = parent * matrix * translation * rotation * scale;
    if (n.camera != -1) {
        const auto& c = model.cameras[n.camera];
        if (c.type != "perspective") return;
        if (o.cameras++ == 0) {
            const float4 e = xf * make_float4(0.0f, 0.0f, 0.0f, 1.0f), u = xf * make_float4(0.0f, 1.0f, 0.0f, 0.0f);
            o.eye[0] = e.x; o.eye[1] = e.y; o.eye[2] = e.z; o.up[0] = u.x; o.up[1] = u.y; o.up[2] = u.z;
            o.fov = (float)c.perspective.yfov * 180.0f / (float)M_PI;
            o.aspect = (float)c.perspective.aspectRatio;
        }
    } else if (n.mesh != -1) {
        for (const auto& pr : model.meshes[n.mesh].primitives) {
            if (pr.mode != TINYGLTF_MODE_TRIANGLES) continue;
            const Span p = span_of(model, pr.attributes.at("POSITION"));
            const unsigned base = (unsigned)(o.pos.size() / 3);
            auto tc = pr.attributes.find("TEXCOORD_0");
            Span t = {nullptr, 0, 0, 0};
            if (tc != pr.attributes.end()) t = span_of(model, tc->second);
            for (size_t i = 0; i < p.count; i++) {
                float v[3];
                memcpy(v, p.base + i * p.stride, 12);
                const float4 w = xf * make_float4(v[0], v[1], v[2], 1.0f);
                o.pos.push_back(w.x); o.pos.push_back(w.y); o.pos.push_back(w.z);
                float q[2] = {0.0f, 0.0f};
                if (t.base) memcpy(q, t.base + i * t.stride, 8);
                o.uv.push_back(q[0]); o.uv.push_back(q[1]);
            }
            const Span ix = span_of(model, pr.indices);
            for (size_t i = 0; i + 3 <= ix.count; i += 3) {
                for (int k = 0; k < 3; k++) {
                    unsigned v = 0;
                    if (ix.comp == TINYGLTF_COMPONENT_TYPE_UNSIGNED_INT) memcpy(&v, ix.base + (i + k) * ix.stride, 4);
                    else { unsigned short h; memcpy(&h, ix.base + (i + k) * ix.stride, 2); v = h; }
                    o.idx.push_back(base + v);
                }
                o.mat.push_back(pr.material);
            }
        }
    }
    for (int c : n.children) node(model, model.nodes[c], xf, o);
}
}  // namespace

extern "C" {
// Two-call protocol (sizes first with null outputs).  Returns 0, or -1 when tinygltf rejects the file.
// materials: per material base colour rgb, metallic, roughness, base-colour image index (-1 = none) -> 6 floats.
int ref_gltf_load(const char* path, int* n_vertices, int* n_triangles, int* n_materials, float* pos, float* uv, unsigned* idx, int* tri_mat,
                  float* materials, float* camera /* eye3 up3 fov aspect count */) {
    tinygltf::Model model;
    tinygltf::TinyGLTF loader;
    std::string err, warn;
    const std::string fn(path);
    const bool ok = fn.size() >= 4 && fn.compare(fn.size() - 4, 4, ".glb") == 0 ? loader.LoadBinaryFromFile(&model, &err, &warn, fn)
                                                                                   : loader.LoadASCIIFromFile(&model, &err, &warn, fn);
    if (!ok) return -1;
    Out o;
    std::vector<int> root(model.nodes.size(), 1);
    for (auto& n : model.nodes) for (int c : n.children) root[c] = 0;
    for (size_t i = 0; i < root.size(); i++) if (root[i]) node(model, model.nodes[i], Matrix4x4::identity(), o);
    *n_vertices = (int)(o.pos.size() / 3); *n_triangles = (int)(o.idx.size() / 3); *n_materials = (int)model.materials.size();
    if (pos) {
        memcpy(pos, o.pos.data(), o.pos.size() * 4); memcpy(uv, o.uv.data(), o.uv.size() * 4);
        memcpy(idx, o.idx.data(), o.idx.size() * 4); memcpy(tri_mat, o.mat.data(), o.mat.size() * 4);
        for (size_t k = 0; k < model.materials.size(); k++) {
            auto& gm = model.materials[k];
            float* d = materials + 6 * k;
            d[0] = d[1] = d[2] = 1.0f; d[3] = 1.0f; d[4] = 1.0f; d[5] = -1.0f;
            auto bc = gm.values.find("baseColorFactor");   // Scene.cpp:372-388
            if (bc != gm.values.end()) { auto c = bc->second.ColorFactor(); d[0] = (float)c[0]; d[1] = (float)c[1]; d[2] = (float)c[2]; }
            auto me = gm.values.find("metallicFactor");
            if (me != gm.values.end()) d[3] = (float)me->second.Factor();
            auto ro = gm.values.find("roughnessFactor");
            if (ro != gm.values.end()) d[4] = (float)ro->second.Factor();
            const int ti = gm.pbrMetallicRoughness.baseColorTexture.index;
            if (ti >= 0 && ti < (int)model.textures.size()) d[5] = (float)model.textures[ti].source;
        }
        memcpy(camera, o.eye, 12); memcpy(camera + 3, o.up, 12); camera[6] = o.fov; camera[7] = o.aspect; camera[8] = (float)o.cameras;
    }
    return 0;
}
}

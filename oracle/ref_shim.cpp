// ORACLE — test infrastructure only.  Thin extern "C" exports over the
// reference's OWN source files, included from where they lie under
// /root/reference/src (never copied): cuda/random.h, cuda/helpers.h,
// sutil/Camera.{h,cpp}, sutil/vec_math.h, and (row f1) the reference's vendored
// OptiXPathTracer/tiny_obj_loader.{h,cc} and stb_image.{h,cpp}, consumed exactly
// as scene_shift.cpp:35-63 and 187-250 consume them.  These are the only files on
// or next to the hot path that compile without the OptiX SDK; the CUDA vector-type headers they need
// ship with this image (triton's bundled CUDA include directory).  Built by
// `make -C oracle ref` into oracle/_ref/libref.so (git-ignored, not gpurun-ignored).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cuda_runtime.h>

#include <cuda/helpers.h>
#include <cuda/random.h>
#include <sutil/Camera.h>

#include <string>
#include <vector>
#include <OptiXPathTracer/tiny_obj_loader.h>
#include <OptiXPathTracer/stb_image.h>

extern "C" {
unsigned ref_tea4(unsigned a, unsigned b) { return tea<4>(a, b); }
unsigned ref_lcg(unsigned* s) { return lcg(*s); }
float ref_rnd(unsigned* s) { return rnd(*s); }
void ref_toSRGB(const float* rgb, int n, float* out) {
    for (int i = 0; i < n; i++) {
        float3 s = toSRGB(make_float3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
        out[3 * i] = s.x; out[3 * i + 1] = s.y; out[3 * i + 2] = s.z;
    }
}
void ref_make_color(const float* rgb, int n, unsigned char* out) {
    for (int i = 0; i < n; i++) {
        uchar4 c = make_color(make_float3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
        out[4 * i] = c.x; out[4 * i + 1] = c.y; out[4 * i + 2] = c.z; out[4 * i + 3] = c.w;
    }
}
void ref_uvw(const float* eye, const float* lookat, const float* up, float fovY, float aspect, float* U, float* V, float* W) {
    sutil::Camera cam(make_float3(eye[0], eye[1], eye[2]), make_float3(lookat[0], lookat[1], lookat[2]),
                      make_float3(up[0], up[1], up[2]), fovY, aspect);
    float3 u, v, w;
    cam.UVWFrame(u, v, w);
    U[0] = u.x; U[1] = u.y; U[2] = u.z; V[0] = v.x; V[1] = v.y; V[2] = v.z; W[0] = w.x; W[1] = w.y; W[2] = w.z;
}

// tinyobj::LoadObj as Scene::getMeshData calls it (sceneLoader.cpp:333-342), flattened as the loop of
// scene_shift.cpp:187-250 hands the shapes on: per shape positions, texcoords zero-padded to 2 per vertex, indices.
// Two-call protocol: sizes first (out pointers null), then fill.  Returns the number of shapes, -1 on failure.
int ref_obj_load(const char* path, int* n_vertices, int* n_indices, int* shape_vertex_count, int* shape_index_count, int shape_cap,
                 float* positions, float* texcoords, unsigned* indices) {
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> mats;
    std::string err;
    tinyobj::LoadObj(shapes, mats, err, path);
    int nv = 0, ni = 0;
    for (size_t j = 0; j < shapes.size(); j++) {
        tinyobj::mesh_t m = shapes[j].mesh;
        while (m.texcoords.size() < m.positions.size() / 3 * 2) m.texcoords.push_back(0);
        const int v = (int)(m.positions.size() / 3), k = (int)m.indices.size();
        if ((int)j < shape_cap && shape_vertex_count) { shape_vertex_count[j] = v; shape_index_count[j] = k; }
        if (positions) {
            memcpy(positions + 3 * (size_t)nv, m.positions.data(), sizeof(float) * 3 * v);
            memcpy(texcoords + 2 * (size_t)nv, m.texcoords.data(), sizeof(float) * 2 * v);
            memcpy(indices + ni, m.indices.data(), sizeof(unsigned) * k);
        }
        nv += v; ni += k;
    }
    *n_vertices = nv; *n_indices = ni;
    return (int)shapes.size();
}
// stbi_load(..., STBI_rgb_alpha) as Material_shift calls it (scene_shift.cpp:38-40).  Returns 0 on success.
int ref_image_load(const char* path, int* w, int* h, unsigned char* rgba, int capacity_bytes) {
    int c = 0;
    stbi_uc* px = stbi_load(path, w, h, &c, STBI_rgb_alpha);
    if (!px) return -1;
    const int bytes = *w * *h * 4;
    if (rgba && bytes <= capacity_bytes) memcpy(rgba, px, bytes);
    stbi_image_free(px);
    return 0;
}
}

// Private to csrc/: the loaded-scene handle behind spcbpt_scene_file_* (scene_file.cpp: `.scene` + OBJ; gltf_file.cpp: glTF 2.0).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/spcbpt.h"

struct spcbpt_scene_file {
    std::vector<float> V, UV;
    std::vector<uint32_t> I;
    std::vector<int32_t> M;
    std::vector<spcbpt_material> materials;
    std::vector<spcbpt_quad_light> lights;
    std::vector<std::vector<uint8_t>> tex_pixels;
    std::vector<spcbpt_texture> textures;
    float eye[3] = {0, 0, 0}, lookat[3] = {0, 0, -1}, up[3] = {0, 1, 0};
    float fov = 35.0f;
    int width = 1920, height = 1001;  // sceneLoader.cpp:201-203 defaults (parsed, ignored by the app)
    int n_mesh_blocks = 0;
    std::string warnings;
    // environment map: `env_file` of the cameraSetting block (sceneLoader.cpp:242), read with the HDRLoader restatement
    std::string env_file;
    std::vector<float> env_rgba;   // width x height RGBA floats, row 0 = top
    int env_w = 0, env_h = 0;
    // the box the reference's sky.center / sky.r come from (optixPathTracer.cpp:458-459): per OBJ shape get_aabb(std::vector<float>)
    // covers only the first third of its vertices (scene_shift.cpp:21-32); the light quads enter with all four corners
    float ref_lo[3] = {1e30f, 1e30f, 1e30f}, ref_hi[3] = {-1e30f, -1e30f, -1e30f};
};


namespace spc_loader {
// binary PPM (P6, maxval 255) -> RGBA8, pinned against the reference's stb_image (tests/test_scene_file.py)
bool load_ppm(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h);
// JPEG / PNG / binary PPM by content -> RGBA8, what stbi_load(..., STBI_rgb_alpha) returns (image_file.cpp; pinned bit-exactly
// against the reference's stb_image, tests/test_image_file.py)
bool load_image(const std::string& path, std::vector<uint8_t>& rgba, int& w, int& h);
}

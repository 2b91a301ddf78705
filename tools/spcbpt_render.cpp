// Headless driver over the C ABI: the frame sequence of the reference app (OptiXPathTracer/optixPathTracer.cpp main 680-837,
// preprocessing 552-608, render loop 791-822) without GLFW/GL — loads a `.scene`, builds the LBVH, runs the preprocessing,
// renders N subframes of "pt" or "SPCBPT_eye" and writes the linear accum buffer as PFM and the tone-mapped frame as PPM.
//   spcbpt_render <file.scene> <data_root> [--alg pt|SPCBPT_eye] [--dim=WxH] [--frames N] [--train-paths N] [--minimal] [--out prefix]
// Build: make -C tools   (links libspcbpt_hip.so)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/spcbpt.h"

static void die(spcbpt_ctx* c, const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, spcbpt_last_error(c));
    exit(1);  // optixPathTracer.cpp:830-834: print and return 1
}
#define CHECK(c, call) do { int rc__ = (call); if (rc__) die(c, #call, rc__); } while (0)

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <file.scene | file.gltf | file.glb> <data_root (ignored for glTF)> [--alg pt|SPCBPT_eye] [--dim=WxH] [--frames N] [--train-paths N] [--minimal] [--out prefix]\n", argv[0]);
        return 0;
    }
    std::string alg = "SPCBPT_eye", out = "render";
    int width = 1920, height = 1000, frames = 16, train_paths = 2000000;  // optixPathTracer.cpp:84-85 default size
    bool minimal = false;
    for (int i = 3; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--alg" && i + 1 < argc) alg = argv[++i];
        else if (a.rfind("--dim=", 0) == 0) { if (sscanf(a.c_str() + 6, "%dx%d", &width, &height) != 2) { fprintf(stderr, "bad --dim\n"); return 1; } }
        else if (a == "--frames" && i + 1 < argc) frames = atoi(argv[++i]);
        else if (a == "--train-paths" && i + 1 < argc) train_paths = atoi(argv[++i]);
        else if (a == "--minimal") minimal = true;
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else { fprintf(stderr, "Unknown option '%s'\n", argv[i]); return 1; }
    }
    spcbpt_scene_file* sf = nullptr;
    const std::string in(argv[1]);
    const bool gltf = in.size() > 5 && (in.compare(in.size() - 5, 5, ".gltf") == 0 || in.compare(in.size() - 4, 4, ".glb") == 0);
    char load_err[512] = {0};
    if (gltf ? spcbpt_gltf_load(argv[1], &sf, load_err, sizeof(load_err)) : spcbpt_scene_file_load(argv[1], argv[2], &sf)) {
        fprintf(stderr, "cannot read %s %s\n", argv[1], load_err);
        return 1;
    }
    if (*spcbpt_scene_file_warnings(sf)) fprintf(stderr, "scene warnings: %s\n", spcbpt_scene_file_warnings(sf));
    spcbpt_scene_desc desc;
    spcbpt_scene_file_desc(sf, &desc);
    float eye[3], lookat[3], up[3], fov;
    spcbpt_scene_file_camera(sf, eye, lookat, up, &fov, nullptr, nullptr);
    spcbpt_ctx* ctx = nullptr;
    int rc = spcbpt_create(&desc, 0, &ctx);
    if (rc) die(nullptr, "spcbpt_create", rc);
    int nt, nn, depth;
    CHECK(ctx, spcbpt_scene_info(ctx, &nt, &nn, &depth));
    printf("scene: %d triangles, BVH %d nodes depth %d\n", nt, nn, depth);
    CHECK(ctx, spcbpt_set_camera_lookat(ctx, eye, lookat, up, fov, (float)width / (float)height));
    CHECK(ctx, spcbpt_resize(ctx, width, height));
    spcbpt_light_trace_params lt = {100000, 52, 1, 0, 0, 1};
    CHECK(ctx, spcbpt_set_light_trace(ctx, &lt));
    auto t0 = std::chrono::steady_clock::now();
    if (alg == "SPCBPT_eye") {
        if (minimal) CHECK(ctx, spcbpt_set_subspace(ctx, nullptr, 0, nullptr, 0, nullptr, nullptr));
        else CHECK(ctx, spcbpt_preprocess(ctx, train_paths, train_paths, 1));
    }
    auto t1 = std::chrono::steady_clock::now();
    printf("preprocessing: %.2f s\n", std::chrono::duration<double>(t1 - t0).count());
    unsigned lt_frame = 1000000;  // continues after the Q passes of the preprocessing like lt_params.launch_frame
    for (int f = 0; f < frames; f++) {
        if (alg == "SPCBPT_eye") {  // launchLVCTrace (optixPathTracer.cpp:515-522)
            CHECK(ctx, spcbpt_launch(ctx, "light trace", ++lt_frame, 0, 0, 1));
            CHECK(ctx, spcbpt_build_sampler(ctx));
        }
        CHECK(ctx, spcbpt_launch(ctx, alg.c_str(), (uint32_t)f, 0, height, 1));  // launchSubframe (609-635)
    }
    CHECK(ctx, spcbpt_sync(ctx));
    auto t2 = std::chrono::steady_clock::now();
    const double sec = std::chrono::duration<double>(t2 - t1).count();
    printf("%d subframes of %s at %dx%d: %.3f s, %.2f Mpaths/s\n", frames, alg.c_str(), width, height, sec,
           ((double)width * height + (alg == "SPCBPT_eye" ? lt.num_core : 0)) * frames / sec / 1e6);
    std::vector<float> accum((size_t)width * height * 4);
    std::vector<uint8_t> frame((size_t)width * height * 4);
    CHECK(ctx, spcbpt_read_accum(ctx, accum.data()));
    CHECK(ctx, spcbpt_read_frame(ctx, frame.data()));
    {  // PFM: bottom row first, which is exactly the accum_buffer orientation
        FILE* f = fopen((out + ".pfm").c_str(), "wb");
        fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
        for (size_t i = 0; i < (size_t)width * height; i++) fwrite(&accum[4 * i], 4, 3, f);
        fclose(f);
    }
    {  // PPM: top row first
        FILE* f = fopen((out + ".ppm").c_str(), "wb");
        fprintf(f, "P6\n%d %d\n255\n", width, height);
        for (int y = height - 1; y >= 0; y--)
            for (int x = 0; x < width; x++) fwrite(&frame[4 * ((size_t)y * width + x)], 1, 3, f);
        fclose(f);
    }
    printf("wrote %s.pfm and %s.ppm\n", out.c_str(), out.c_str());
    spcbpt_destroy(ctx);
    spcbpt_scene_file_free(sf);
    return 0;
}

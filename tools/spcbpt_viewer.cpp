// The reference application's interactive session without a window (row f3): main() of OptiXPathTracer/optixPathTracer.cpp
// (680-837) with GLFW's event queue replaced by an event script and GLDisplay by PPM snapshots.  Pure C++ over the C ABI:
// load scene -> create -> initCameraState -> preprocessing (or a checkpoint) -> loop { events; spcbpt_viewer_frame }.
//   spcbpt_viewer <file.scene | file.gltf | file.glb> <data_root> [--dim=WxH] [--script file | -] [--train-paths N] [--minimal]
//                 [--load-checkpoint dir] [--save-checkpoint dir]
// Script (one event per line, `#` comments; what a GLFW front end would forward to the same spcbpt_viewer_* calls):
//   press left|right|middle X Y     release left|right|middle X Y     move X Y     scroll DY     resize W H     iconify 0|1
//   key ESCAPE|SPACE|C|P|W [press|repeat|release]     fps F (fixed playback rate; 0 = measured)
//   frames N (run the render loop N times)     save PREFIX (PREFIX.ppm, the displayed frame)     state (print the camera)
// Build: make -C tools
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../include/spcbpt.h"

static void die(spcbpt_ctx* c, const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, c ? spcbpt_last_error(c) : "");
    exit(1);
}
#define CHECK(c, call) do { int rc__ = (call); if (rc__) die(c, #call, rc__); } while (0)

static int button_code(const std::string& s) { return s == "left" ? 0 : s == "right" ? 1 : 2; }
static int key_code(const std::string& s) {
    if (s == "ESCAPE") return 256;
    if (s == "SPACE") return 32;
    return s.size() == 1 ? (int)s[0] : -1;  // GLFW letter keys are their ASCII capitals
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <file.scene | file.gltf | file.glb> <data_root> [--dim=WxH] [--script file|-] [--train-paths N] [--minimal] [--load-checkpoint dir] [--save-checkpoint dir]\n", argv[0]);
        return 0;
    }
    int width = 1920, height = 1000, train_paths = 2000000;  // optixPathTracer.cpp:686-687
    bool minimal = false;
    std::string script = "-", load_ckpt, save_ckpt;
    for (int i = 3; i < argc; i++) {
        std::string a = argv[i];
        if (a.rfind("--dim=", 0) == 0) { if (sscanf(a.c_str() + 6, "%dx%d", &width, &height) != 2) { fprintf(stderr, "bad --dim\n"); return 1; } }
        else if (a == "--script" && i + 1 < argc) script = argv[++i];
        else if (a == "--train-paths" && i + 1 < argc) train_paths = atoi(argv[++i]);
        else if (a == "--minimal") minimal = true;
        else if (a == "--load-checkpoint" && i + 1 < argc) load_ckpt = argv[++i];
        else if (a == "--save-checkpoint" && i + 1 < argc) save_ckpt = argv[++i];
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
    spcbpt_viewer* v = nullptr;
    CHECK(ctx, spcbpt_viewer_create(ctx, eye, lookat, up, fov, width, height, &v));  // initCameraState
    spcbpt_light_trace_params lt = {100000, 52, 1, 0, 0, 1};
    CHECK(ctx, spcbpt_set_light_trace(ctx, &lt));
    {  // "pre tracing" block of main (763-766): handleCameraUpdate, then preprocessing
        CHECK(ctx, spcbpt_set_camera_lookat(ctx, eye, lookat, up, fov, (float)width / (float)height));
        CHECK(ctx, spcbpt_resize(ctx, width, height));
        const auto t0 = std::chrono::steady_clock::now();
        if (!load_ckpt.empty()) CHECK(ctx, spcbpt_checkpoint_load(ctx, load_ckpt.c_str()));
        else if (minimal) CHECK(ctx, spcbpt_set_subspace(ctx, nullptr, 0, nullptr, 0, nullptr, nullptr));
        else CHECK(ctx, spcbpt_preprocess(ctx, train_paths, train_paths, 1));
        printf("preprocessing: %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        if (!save_ckpt.empty()) CHECK(ctx, spcbpt_checkpoint_save(ctx, save_ckpt.c_str()));
    }
    std::ifstream file;
    if (script != "-") { file.open(script); if (!file) { fprintf(stderr, "cannot open script %s\n", script.c_str()); return 1; } }
    std::istream& src = script == "-" ? std::cin : file;
    std::string line;
    long long total_frames = 0;
    const auto t_loop = std::chrono::steady_clock::now();
    spcbpt_viewer_state st;
    while (std::getline(src, line)) {
        std::istringstream ls(line.substr(0, line.find('#')));
        std::string cmd, a;
        if (!(ls >> cmd)) continue;
        double x = 0, y = 0;
        if (cmd == "press" || cmd == "release") { ls >> a >> x >> y; CHECK(ctx, spcbpt_viewer_mouse_button(v, button_code(a), cmd == "press" ? 1 : 0, x, y)); }
        else if (cmd == "move") { ls >> x >> y; CHECK(ctx, spcbpt_viewer_cursor_pos(v, x, y)); }
        else if (cmd == "scroll") { ls >> y; CHECK(ctx, spcbpt_viewer_scroll(v, 0.0, y)); }
        else if (cmd == "resize") { int w = 0, h = 0; ls >> w >> h; CHECK(ctx, spcbpt_viewer_window_size(v, w, h)); }
        else if (cmd == "iconify") { int on = 0; ls >> on; CHECK(ctx, spcbpt_viewer_iconify(v, on)); }
        else if (cmd == "key") {
            std::string act = "press";
            ls >> a >> act;
            const int code = key_code(a);
            if (code < 0) { fprintf(stderr, "unknown key %s\n", a.c_str()); return 1; }
            CHECK(ctx, spcbpt_viewer_key(v, code, act == "press" ? 1 : act == "repeat" ? 2 : 0));
        }
        else if (cmd == "fps") { float f = 0; ls >> f; CHECK(ctx, spcbpt_viewer_set_fps(v, f)); }
        else if (cmd == "frames") {
            int n = 1;
            ls >> n;
            for (int i = 0; i < n; i++) { CHECK(ctx, spcbpt_viewer_frame(v)); total_frames++; }
        }
        else if (cmd == "state") {
            spcbpt_viewer_get_state(v, &st);
            printf("%s subframe %u  %dx%d  eye %g %g %g  lookat %g %g %g  up %g %g %g  %.1f fps\n", spcbpt_viewer_alg_name(st.alg_id),
                   st.subframe_index, st.width, st.height, st.eye[0], st.eye[1], st.eye[2], st.lookat[0], st.lookat[1], st.lookat[2],
                   st.up[0], st.up[1], st.up[2], st.render_fps);
        }
        else if (cmd == "save") {  // what GLDisplay would show: the tone-mapped frame, top row first
            ls >> a;
            spcbpt_viewer_get_state(v, &st);
            std::vector<uint8_t> frame((size_t)st.width * st.height * 4);
            CHECK(ctx, spcbpt_read_frame(ctx, frame.data()));
            FILE* f = fopen((a + ".ppm").c_str(), "wb");
            if (!f) { fprintf(stderr, "cannot write %s.ppm\n", a.c_str()); return 1; }
            fprintf(f, "P6\n%d %d\n255\n", st.width, st.height);
            for (int yy = st.height - 1; yy >= 0; yy--)
                for (int xx = 0; xx < st.width; xx++) fwrite(&frame[4 * ((size_t)yy * st.width + xx)], 1, 3, f);
            fclose(f);
            printf("wrote %s.ppm (%s, %u subframes)\n", a.c_str(), spcbpt_viewer_alg_name(st.alg_id), st.subframe_index);
        }
        else { fprintf(stderr, "unknown script command '%s'\n", cmd.c_str()); return 1; }
        spcbpt_viewer_get_state(v, &st);
        if (st.should_close) break;  // glfwWindowShouldClose
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
    printf("%lld frames in %.3f s (%.1f fps)\n", total_frames, sec, total_frames / (sec > 0 ? sec : 1));
    spcbpt_viewer_destroy(v);
    spcbpt_destroy(ctx);
    spcbpt_scene_file_free(sf);
    return 0;
}

// Row f3: the reference's interactive loop without a window.  The application state machine of optixPathTracer.cpp -- GLFW
// callbacks (121-241), updateState / handleCameraUpdate / handleResize (333-379), initCameraState (661-670) and one pass of
// the render loop (791-822) -- over sutil::Trackball (sutil/Trackball.cpp:49-213) and sutil::Camera (sutil/Camera.cpp:34-45),
// restated as host C++ behind the C ABI.  Events come from the caller (tools/spcbpt_viewer.cpp replays an event script; a
// windowing front end would forward its callbacks one to one); display is the caller's business (spcbpt_read_frame).
// The camera / trackball arithmetic is pinned bit-exactly against the reference's own Trackball.cpp + Camera.cpp
// (oracle/_ref, tests/test_viewer.py).  Pure host code: with a null context the state machine runs without launching.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/spcbpt.h"

namespace {

struct V3 { float x, y, z; };
inline V3 mk(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { const float inv = 1.0f / s; return a * inv; }  // vec_math.h: operator/(float3, float)
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float length(V3 v) { return sqrtf(dot(v, v)); }
inline V3 normalize(V3 v) { const float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }

const float kPif = 3.14159265358979323846f, k1Pif = 0.318309886183790671538f;  // M_PIf, M_1_PIf (vec_math.h:43-51)

struct Camera {  // sutil::Camera
    V3 eye, lookat, up;
    float fovY, aspect;
    void uvw(V3& U, V3& V, V3& W) const {  // Camera.cpp:34-45
        W = lookat - eye;
        const float wlen = length(W);
        U = normalize(cross(W, up));
        V = normalize(cross(U, W));
        const float vlen = wlen * tanf(0.5f * fovY * kPif / 180.0f);
        V = V * vlen;
        const float ulen = vlen * aspect;
        U = U * ulen;
    }
};

// Trackball.cpp calls the C library's cos / sin / atan2 / asin / fmod unqualified after <cmath>: the double versions, narrowed
// on assignment.  Spelled out here so that the rounding is the same.
inline float radiansf(float degrees) { return degrees * kPif / 180.0f; }
inline float degreesf(float radians) { return radians * k1Pif * 180.0f; }

struct Trackball {  // sutil::Trackball
    enum { EyeFixed = 0, LookAtFixed = 1 };
    bool gimbalLock = false;
    int viewMode = LookAtFixed;
    Camera* cam = nullptr;
    float dist = 0.0f, zoomMultiplier = 1.1f, moveSpeed = 1.0f;
    float latitude = 0.0f, longitude = 0.0f;
    int prevX = 0, prevY = 0;
    bool tracking = false;
    V3 u = {0, 0, 0}, v = {0, 0, 0}, w = {0, 0, 0};

    void startTracking(int x, int y) { prevX = x; prevY = y; tracking = true; }
    void reinitOrientationFromCamera() {  // 145-155
        cam->uvw(u, v, w);
        u = normalize(u);
        v = normalize(v);
        w = normalize(-w);
        std::swap(v, w);
        latitude = 0.0f;
        longitude = 0.0f;
        dist = length(cam->lookat - cam->eye);
    }
    void setCamera(Camera* c) { cam = c; reinitOrientationFromCamera(); }
    void setReferenceFrame(V3 a, V3 b, V3 c) {  // 121-134
        u = a; v = b; w = c;
        const V3 dirWS = -normalize(cam->lookat - cam->eye);
        const V3 dl = mk(dot(dirWS, a), dot(dirWS, b), dot(dirWS, c));
        longitude = (float)::atan2((double)dl.x, (double)dl.y);
        latitude = (float)::asin((double)dl.z);
    }
    void updateCamera() {  // 98-119
        V3 ld;
        ld.x = (float)(::cos((double)latitude) * ::sin((double)longitude));
        ld.y = (float)(::cos((double)latitude) * ::cos((double)longitude));
        ld.z = (float)::sin((double)latitude);
        const V3 dirWS = u * ld.x + v * ld.y + w * ld.z;
        if (viewMode == EyeFixed) cam->lookat = cam->eye - dirWS * dist;
        else cam->eye = cam->lookat + dirWS * dist;
    }
    void updateTracking(int x, int y) {  // 74-96
        if (!tracking) { startTracking(x, y); return; }
        const int dx = x - prevX, dy = y - prevY;
        prevX = x; prevY = y;
        latitude = radiansf(std::min(89.0f, std::max(-89.0f, degreesf(latitude) + 0.5f * dy)));
        longitude = radiansf((float)::fmod((double)(degreesf(longitude) - 0.5f * dx), (double)360.0f));
        updateCamera();
        if (!gimbalLock) { reinitOrientationFromCamera(); cam->up = w; }
    }
    void zoom(int direction) {  // 136-143
        const float z = direction > 0 ? 1 / zoomMultiplier : zoomMultiplier;
        dist *= z;
        cam->eye = cam->lookat + (cam->eye - cam->lookat) * z;
    }
    bool wheelEvent(int dir) { zoom(dir); return true; }
};

const char* const kAlgs[2] = {"pt", "SPCBPT_eye"};  // render_alg, optixPathTracer.cpp:91

}  // namespace

struct spcbpt_viewer {
    spcbpt_ctx* ctx;
    Camera camera;
    Trackball trackball;
    int width, height;
    bool camera_changed = true, resize_dirty = false, minimized = false, one_frame_render_only = false, should_close = false;
    int mouse_button = -1;           // optixPathTracer.cpp:82
    int render_alg_id = 1;           // 92: starts on SPCBPT_eye
    float render_fps = 60.0f;        // 94
    uint32_t subframe_index = 0, lt_launch_frame = 0;
    double cursor_x = 0, cursor_y = 0;  // what glfwGetCursorPos returns inside mouseButtonCallback
    bool fixed_fps = false;
    // How far the loop runs ahead of what it shows (spcbpt_viewer_set_pipeline; every mode displays the same frames):
    //   0  the reference's order: light pass, sampler build, eye launch, device sync -- strictly in turn (optixPathTracer.cpp:791-822)
    //   1  the NEXT frame's light pass is launched beside this frame's eye kernel
    //   2  (default) ... and the next frame is traced speculatively while this one is shown: its sampler build and eye launch are
    //      queued before this call returns (spcbpt_launch_deferred); the next call merges it if nothing it depends on changed
    //      (camera, size, algorithm, subframe restart) and drops it otherwise -- the light pass and the sampler are kept either
    //      way, they do not depend on the camera, so the k-th "SPCBPT_eye" frame always uses the k-th light pass.
    int pipeline = 2;
    bool light_pending = false;       // a light pass has been launched whose sampler is not built yet
    bool sampler_ready = false;       // the sampler of the NEXT frame to show is built (its light pass consumed)
    bool spec_in_flight = false;      // a deferred eye / pt launch is outstanding ...
    int spec_alg = -1;                // ... of this algorithm
    uint32_t spec_subframe = 0;       // ... and subframe index
    long long frames = 0, spec_hits = 0, spec_drops = 0;
    int ctx_light_ahead_at_create = 0;   // the context's mode as the viewer found it: restored by spcbpt_viewer_destroy
    void revalidate();
};

// The viewer's flags describe work it queued on the context; a host that shares the context may have consumed or invalidated it
// between two frames (merged / dropped the deferred frame, switched light-ahead mode, installed a new tuple or sky, imported a
// cache).  Each frame starts from what the context says; the light-pass counter is wound back over passes that are gone.
void spcbpt_viewer::revalidate() {
    if (!ctx) return;
    int ahead = 0, pend = 0, intact = 0, deferred = 0;
    if (spcbpt_get_pipeline_state(ctx, &ahead, &pend, &intact, &deferred)) return;
    if (spec_in_flight && !deferred) spec_in_flight = false;
    if (light_pending && pend == 0) { light_pending = false; lt_launch_frame--; }
    if (sampler_ready && !intact) { sampler_ready = false; lt_launch_frame--; }
}

// Live viewers, so that a context that is destroyed FIRST can tell them (round 6, advisor: spcbpt_viewer_destroy calls into the
// context; with `Renderer.close()` before `Viewer.close()` that was a use-after-free).  spcbpt_destroy calls
// spc_viewers_forget_context (not part of the C ABI: exports.map keeps it local to the library); a forgotten viewer is the
// state-machine-only viewer of `ctx == NULL`, which every entry point already supports.
namespace {
std::mutex g_viewers_mutex;
std::vector<spcbpt_viewer*> g_viewers;
}  // namespace

void spc_viewers_forget_context(spcbpt_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_viewers_mutex);
    for (spcbpt_viewer* v : g_viewers) {
        if (v->ctx != ctx) continue;
        v->ctx = nullptr;
        v->light_pending = v->sampler_ready = v->spec_in_flight = false;
    }
}

extern "C" {

int spcbpt_viewer_create(spcbpt_ctx* ctx, const float eye[3], const float lookat[3], const float up[3], float fov_y, int width,
                         int height, spcbpt_viewer** out) {
    if (!eye || !lookat || !up || !out || width < 1 || height < 1) return SPCBPT_ERR_INVALID_ARG;
    spcbpt_viewer* v = new spcbpt_viewer();
    v->ctx = ctx;
    v->width = width; v->height = height;
    v->camera.eye = mk(eye[0], eye[1], eye[2]);
    v->camera.lookat = mk(lookat[0], lookat[1], lookat[2]);
    v->camera.up = mk(up[0], up[1], up[2]);
    v->camera.fovY = fov_y;
    v->camera.aspect = 1.0f;  // sutil::Camera's default until handleCameraUpdate sets width / height
    // initCameraState (661-670)
    v->camera_changed = true;
    v->trackball.setCamera(&v->camera);
    v->trackball.moveSpeed = 10.0f;
    v->trackball.setReferenceFrame(mk(1.0f, 0.0f, 0.0f), mk(0.0f, 0.0f, 1.0f), mk(0.0f, 1.0f, 0.0f));
    v->trackball.gimbalLock = true;
    if (ctx) {   // the default loop runs a frame ahead (pipeline 2): light passes may then be launched before their sampler is built
        int rc = spcbpt_get_pipeline_state(ctx, &v->ctx_light_ahead_at_create, nullptr, nullptr, nullptr);
        if (!rc) rc = spcbpt_set_light_ahead(ctx, 1);
        if (rc) { delete v; return rc; }
    }
    {
        std::lock_guard<std::mutex> lock(g_viewers_mutex);
        g_viewers.push_back(v);
    }
    *out = v;
    return SPCBPT_OK;
}

// Leaves the context as spcbpt_viewer_create found it: the frame traced ahead is dropped, the light passes launched ahead leave
// the queue and the light-ahead mode goes back to what it was -- a host that goes on with the plain loop (light pass, build, eye
// launch) on the same context gets the reference's frames, not SPCBPT_ERR_STATE or the sampler of a pass the viewer queued.
void spcbpt_viewer_destroy(spcbpt_viewer* v) {
    if (!v) return;
    if (v->ctx) {
        int deferred = 0;
        if (!spcbpt_get_pipeline_state(v->ctx, nullptr, nullptr, nullptr, &deferred) && deferred) (void)spcbpt_merge_deferred(v->ctx, 0);
        (void)spcbpt_set_light_ahead(v->ctx, v->ctx_light_ahead_at_create);   // (waits for what is queued; clears the pending passes)
    }
    {
        std::lock_guard<std::mutex> lock(g_viewers_mutex);
        g_viewers.erase(std::remove(g_viewers.begin(), g_viewers.end(), v), g_viewers.end());
    }
    delete v;
}

// mouseButtonCallback (121-136).  GLFW codes: button 0 left, 1 right, 2 middle; action 1 press, 0 release.  The position is
// the cursor's at the time of the click (glfwGetCursorPos).
int spcbpt_viewer_mouse_button(spcbpt_viewer* v, int button, int action, double x, double y) {
    if (!v) return SPCBPT_ERR_INVALID_ARG;
    v->cursor_x = x; v->cursor_y = y;
    if (action == 1) {
        v->mouse_button = button;
        v->trackball.startTracking((int)x, (int)y);
    } else {
        v->mouse_button = -1;
    }
    return SPCBPT_OK;
}

// cursorPosCallback (139-155): left drag orbits the eye around the look-at point, right drag turns the view around the eye
int spcbpt_viewer_cursor_pos(spcbpt_viewer* v, double x, double y) {
    if (!v) return SPCBPT_ERR_INVALID_ARG;
    v->cursor_x = x; v->cursor_y = y;
    if (v->mouse_button == 0) {
        v->trackball.viewMode = Trackball::LookAtFixed;
        v->trackball.updateTracking((int)x, (int)y);
        v->camera_changed = true;
    } else if (v->mouse_button == 1) {
        v->trackball.viewMode = Trackball::EyeFixed;
        v->trackball.updateTracking((int)x, (int)y);
        v->camera_changed = true;
    }
    return SPCBPT_OK;
}

int spcbpt_viewer_scroll(spcbpt_viewer* v, double /*xscroll*/, double yscroll) {  // scrollCallback (237-241)
    if (!v) return SPCBPT_ERR_INVALID_ARG;
    if (v->trackball.wheelEvent((int)yscroll)) v->camera_changed = true;
    return SPCBPT_OK;
}

int spcbpt_viewer_window_size(spcbpt_viewer* v, int res_x, int res_y) {  // windowSizeCallback (158-172)
    if (!v) return SPCBPT_ERR_INVALID_ARG;
    if (v->minimized) return SPCBPT_OK;
    if (res_x < 1) res_x = 1;  // sutil::ensureMinimumSize
    if (res_y < 1) res_y = 1;
    v->width = res_x; v->height = res_y;
    v->camera_changed = true;
    v->resize_dirty = true;
    return SPCBPT_OK;
}

int spcbpt_viewer_iconify(spcbpt_viewer* v, int iconified) {  // windowIconifyCallback (175-178)
    if (!v) return SPCBPT_ERR_INVALID_ARG;
    v->minimized = iconified > 0;
    return SPCBPT_OK;
}

// keyCallback (181-234).  GLFW codes: ESCAPE 256, SPACE 32, C 67, G 71, P 80, W 87; action 1 press, 2 repeat, 0 release.
// As in the reference the W branch is outside the `action == PRESS` test: it fires on press, repeat and release alike.
int spcbpt_viewer_key(spcbpt_viewer* v, int key, int action) {
    if (!v) return SPCBPT_ERR_INVALID_ARG;
    if (action == 1) {
        if (key == 256) {
            v->should_close = true;
        } else if (key == 67) {
            printf("Camera Info:\n");
            printf("up      %f %f %f\n", v->camera.up.x, v->camera.up.y, v->camera.up.z);
            printf("eye     %f %f %f\n", v->camera.eye.x, v->camera.eye.y, v->camera.eye.z);
            printf("lookat  %f %f %f\n", v->camera.lookat.x, v->camera.lookat.y, v->camera.lookat.z);
        } else if (key == 32) {
            v->render_alg_id++;
            if (v->render_alg_id >= 2) v->render_alg_id = 0;
            v->camera_changed = true;
            v->resize_dirty = true;
        } else if (key == 80) {
            v->one_frame_render_only = !v->one_frame_render_only;
        }
    }
    if (key == 87) {
        V3 eye = v->camera.eye, lookat = v->camera.lookat;
        const V3 dir = normalize(lookat - eye);
        const float speed = 0.5;
        eye = eye + dir / v->render_fps * speed;
        lookat = lookat + dir / v->render_fps * speed;
        v->camera.eye = eye;
        v->camera.lookat = lookat;
        v->camera_changed = true;
        v->resize_dirty = true;
    }
    return SPCBPT_OK;
}

// Playback at a fixed frame rate: render_fps stays at `fps` instead of following the measured loop time (the W key step
// divides by it).  fps <= 0 returns to the measured rate.
int spcbpt_viewer_set_fps(spcbpt_viewer* v, float fps) {
    if (!v) return SPCBPT_ERR_INVALID_ARG;
    v->fixed_fps = fps > 0.0f;
    if (v->fixed_fps) v->render_fps = fps;
    return SPCBPT_OK;
}

int spcbpt_viewer_set_pipeline(spcbpt_viewer* v, int mode) {
    if (!v || mode < 0 || mode > 2) return SPCBPT_ERR_INVALID_ARG;
    if (v->ctx) {
        if (v->spec_in_flight) { const int rc = spcbpt_merge_deferred(v->ctx, 0); if (rc) return rc; }
        const int rc = spcbpt_set_light_ahead(v->ctx, mode != 0);   // (drops a pass launched ahead)
        if (rc) return rc;
    }
    // what was launched ahead is gone; the light-pass counter is wound back over it, so that the k-th frame keeps the k-th pass
    if (v->light_pending) v->lt_launch_frame--;
    if (v->sampler_ready) v->lt_launch_frame--;
    v->pipeline = mode;
    v->light_pending = v->sampler_ready = v->spec_in_flight = false;
    return SPCBPT_OK;
}
int spcbpt_viewer_set_light_ahead(spcbpt_viewer* v, int on) { return spcbpt_viewer_set_pipeline(v, on ? 1 : 0); }   // (the round-3 name)

// One pass of the render loop (791-822): updateState -> [SPCBPT_eye: launchLVCTrace] -> launchSubframe -> ++subframe_index.
int spcbpt_viewer_frame(spcbpt_viewer* v) {
    if (!v) return SPCBPT_ERR_INVALID_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    v->revalidate();
    // updateState (372-379)
    const bool state_changed = v->camera_changed || v->resize_dirty || v->one_frame_render_only;   // what a frame traced ahead did not know
    if (state_changed) v->subframe_index = 0;
    int rc = SPCBPT_OK;
    // the frame traced ahead: still this frame, or overtaken by an event (dropped before the size or the camera change)
    bool have = false;
    if (v->spec_in_flight && v->ctx) {
        have = !state_changed && v->spec_alg == v->render_alg_id && v->spec_subframe == v->subframe_index;
        if (!have) { v->spec_in_flight = false; v->spec_drops++; rc = spcbpt_merge_deferred(v->ctx, 0); if (rc) return rc; }   // (whatever the call says, the context holds no deferred frame afterwards)
    }
    if (v->camera_changed) {  // handleCameraUpdate (352-370)
        v->camera_changed = false;
        v->camera.aspect = (float)v->width / (float)v->height;
        V3 U, V, W;
        v->camera.uvw(U, V, W);
        if (v->ctx) rc = spcbpt_set_camera(v->ctx, &v->camera.eye.x, &U.x, &V.x, &W.x);
        if (rc) return rc;
    }
    if (v->resize_dirty) {  // handleResize (333-350): also how the accumulation restarts after an algorithm switch
        v->resize_dirty = false;
        if (v->ctx) rc = spcbpt_resize(v->ctx, v->width, v->height);
        if (rc) return rc;
    }
    if (v->ctx) {
        const bool spcbpt = v->render_alg_id == 1;
        // ---- the frame to show: the speculative one if it is still this frame, else render it now
        if (have) {
            v->spec_in_flight = false;
            v->spec_hits++;
            rc = spcbpt_merge_deferred(v->ctx, 1);
            if (rc) return rc;
        }
        if (!have) {
            // a frame traced ahead was dropped: its sampler (built a call ago, a later pass launched since) serves again if the context still has it
            if (spcbpt && v->sampler_ready && spcbpt_reuse_sampler(v->ctx) != SPCBPT_OK) { v->sampler_ready = false; v->lt_launch_frame--; }
            if (spcbpt && !v->sampler_ready) {  // launchLVCTrace (515-522)
                if (!v->light_pending) rc = spcbpt_launch(v->ctx, "light trace", ++v->lt_launch_frame, 0, 0, 1);   // (else: launched ahead by the previous frame)
                if (rc) return rc;
                v->light_pending = false;
                rc = spcbpt_build_sampler(v->ctx);
                if (rc) return rc;
            }
            rc = spcbpt_launch(v->ctx, kAlgs[v->render_alg_id], v->subframe_index, 0, v->height, 1);  // launchSubframe (609-635)
            if (rc) return rc;
        }
        if (spcbpt) v->sampler_ready = false;   // consumed by the frame being shown
        // ---- ahead of the display
        // Mode 2 speculates only from a steady view: a call that itself saw an event (a drag in progress, a resize, a key) is likely to
        // be followed by another, the frame queued here would be dropped by it -- after running to completion beside the real frame,
        // i.e. two eye kernels per displayed frame for as long as the camera moves.  Such a call runs ahead as mode 1 does.
        const bool speculate = v->pipeline == 2 && !v->one_frame_render_only && !state_changed;
        if (v->pipeline >= 1 && !speculate && spcbpt && !v->light_pending) {   // the next frame's light pass, beside this frame's eye kernel
            rc = spcbpt_launch(v->ctx, "light trace", ++v->lt_launch_frame, 0, 0, 1);
            if (rc) return rc;
            v->light_pending = true;
        }
        if (speculate) {
            // frame f + 1, assuming that nothing changes: sampler build (its light pass was launched a call ago and ran beside the
            // eye kernel of f), eye launch without the film merge, and the light pass of f + 2 beside it
            if (spcbpt) {
                if (!v->light_pending) { rc = spcbpt_launch(v->ctx, "light trace", ++v->lt_launch_frame, 0, 0, 1); if (rc) return rc; }
                v->light_pending = false;
                rc = spcbpt_build_sampler(v->ctx);
                if (rc) return rc;
                v->sampler_ready = true;
            }
            rc = spcbpt_launch_deferred(v->ctx, kAlgs[v->render_alg_id], v->subframe_index + 1, 0, v->height, 1);
            if (rc) return rc;
            v->spec_in_flight = true; v->spec_alg = v->render_alg_id; v->spec_subframe = v->subframe_index + 1;
            if (spcbpt) {
                rc = spcbpt_launch(v->ctx, "light trace", ++v->lt_launch_frame, 0, 0, 1);
                if (rc) return rc;
                v->light_pending = true;
            }
            rc = spcbpt_sync_film(v->ctx);   // the frame to show is complete; what was queued behind it keeps running
        } else if (v->pipeline == 2) {
            rc = spcbpt_sync_film(v->ctx);   // (the light pass launched ahead keeps running)
        } else {
            rc = spcbpt_sync(v->ctx);  // CUDA_SYNC_CHECK: the interactive loop shows every subframe
        }
        if (rc) return rc;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!v->fixed_fps && dt > 0.0) v->render_fps = (float)(1.0 / dt);  // 813
    ++v->subframe_index;
    ++v->frames;
    return SPCBPT_OK;
}

int spcbpt_viewer_get_state(spcbpt_viewer* v, spcbpt_viewer_state* s) {
    if (!v || !s) return SPCBPT_ERR_INVALID_ARG;
    memset(s, 0, sizeof(*s));
    Camera c = v->camera;
    c.aspect = (float)v->width / (float)v->height;
    V3 U, V, W;
    c.uvw(U, V, W);
    memcpy(s->eye, &c.eye, 12); memcpy(s->lookat, &c.lookat, 12); memcpy(s->up, &c.up, 12);
    memcpy(s->U, &U, 12); memcpy(s->V, &V, 12); memcpy(s->W, &W, 12);
    s->fov_y = c.fovY; s->aspect = c.aspect;
    s->width = v->width; s->height = v->height;
    s->subframe_index = v->subframe_index;
    s->alg_id = v->render_alg_id;
    s->should_close = v->should_close ? 1 : 0;
    s->one_frame_render_only = v->one_frame_render_only ? 1 : 0;
    s->camera_changed = v->camera_changed ? 1 : 0;
    s->render_fps = v->render_fps;
    return SPCBPT_OK;
}

const char* spcbpt_viewer_alg_name(int alg_id) { return alg_id >= 0 && alg_id < 2 ? kAlgs[alg_id] : ""; }

}  // extern "C"

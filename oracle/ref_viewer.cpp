// ORACLE — test infrastructure only.  Drives the reference's OWN sutil::Trackball (sutil/Trackball.cpp) and sutil::Camera
// (sutil/Camera.cpp), compiled from where they lie under /root/reference/src (never copied), through the few lines of glue
// the application's GLFW callbacks consist of (optixPathTracer.cpp: initCameraState 661-670, mouseButtonCallback 121-136,
// cursorPosCallback 139-155, scrollCallback 237-241, the W key 216-229).  Built by `make -C oracle ref` into
// oracle/_ref/libref_viewer.so; pins csrc/viewer.cpp (row f3) bit-exactly.
#include <cuda_runtime.h>

#include <sutil/Camera.h>
#include <sutil/Trackball.h>
#include <sutil/vec_math.h>

extern "C" {
// events: n records of 4 doubles (type, a, b, c):
//   0 press(button a, x b, y c)   1 release(button a)   2 cursor(x b, y c)   3 scroll(yscroll a)   4 key W (render_fps a)
// out: per event eye, lookat, up, then U, V, W at `aspect` (18 floats)
int ref_viewer_replay(const float* eye, const float* lookat, const float* up, float fov_y, float aspect, const double* ev, int n,
                      float* out) {
    sutil::Camera camera(make_float3(eye[0], eye[1], eye[2]), make_float3(lookat[0], lookat[1], lookat[2]),
                         make_float3(up[0], up[1], up[2]), fov_y, 1.0f);
    sutil::Trackball trackball;
    trackball.setCamera(&camera);
    trackball.setMoveSpeed(10.0f);
    trackball.setReferenceFrame(make_float3(1.0f, 0.0f, 0.0f), make_float3(0.0f, 0.0f, 1.0f), make_float3(0.0f, 1.0f, 0.0f));
    trackball.setGimbalLock(true);
    int mouse_button = -1;
    for (int i = 0; i < n; i++) {
        const double* e = ev + 4 * i;
        const int type = (int)e[0];
        if (type == 0) {
            mouse_button = (int)e[1];
            trackball.startTracking(static_cast<int>(e[2]), static_cast<int>(e[3]));
        } else if (type == 1) {
            mouse_button = -1;
        } else if (type == 2) {
            if (mouse_button == 0) {
                trackball.setViewMode(sutil::Trackball::LookAtFixed);
                trackball.updateTracking(static_cast<int>(e[2]), static_cast<int>(e[3]), 0, 0);
            } else if (mouse_button == 1) {
                trackball.setViewMode(sutil::Trackball::EyeFixed);
                trackball.updateTracking(static_cast<int>(e[2]), static_cast<int>(e[3]), 0, 0);
            }
        } else if (type == 3) {
            trackball.wheelEvent((int)e[1]);
        } else if (type == 4) {
            float render_fps = (float)e[1];
            float3 eye = camera.eye();
            float3 lookat = camera.lookat();
            float3 dir = normalize(lookat - eye);
            float speed = 0.5;
            eye += dir / render_fps * speed;
            lookat += dir / render_fps * speed;
            camera.setEye(eye);
            camera.setLookat(lookat);
        }
        camera.setAspectRatio(aspect);
        float3 U, V, W;
        camera.UVWFrame(U, V, W);
        float* o = out + 18 * i;
        o[0] = camera.eye().x; o[1] = camera.eye().y; o[2] = camera.eye().z;
        o[3] = camera.lookat().x; o[4] = camera.lookat().y; o[5] = camera.lookat().z;
        o[6] = camera.up().x; o[7] = camera.up().y; o[8] = camera.up().z;
        o[9] = U.x; o[10] = U.y; o[11] = U.z; o[12] = V.x; o[13] = V.y; o[14] = V.z; o[15] = W.x; o[16] = W.y; o[17] = W.z;
    }
    return 0;
}
}

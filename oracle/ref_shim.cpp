// ORACLE — test infrastructure only.  Thin extern "C" exports over the
// reference's OWN source files, included from where they lie under
// /root/reference/src (never copied): cuda/random.h, cuda/helpers.h,
// sutil/Camera.{h,cpp}, sutil/vec_math.h.  These are the only hot-path files
// that compile without the OptiX SDK; the CUDA vector-type headers they need
// ship with this image (triton's bundled CUDA include directory).  Built by
// `make -C oracle ref` into oracle/_ref/libref.so (git-ignored, not gpurun-ignored).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cuda_runtime.h>

#include <cuda/helpers.h>
#include <cuda/random.h>
#include <sutil/Camera.h>

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
}

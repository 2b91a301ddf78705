// ORACLE — test infrastructure only.  Never linked into, imported by or called
// from the product path (spcbpt-optix7_amd/, include/); only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// Minimal float3/float2 algebra with the operation order of the reference's
// sutil/vec_math.h (normalize = v * (1/sqrtf(dot)), lerp = a + t*(b-a),
// clamp = fmaxf(a, fminf(f, b)), fmaxf(float3) = max component).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

inline float2 make_float2(float x, float y) { return {x, y}; }
inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
inline float3 make_float3(float s) { return {s, s, s}; }
inline float3 load3(const float* p) { return {p[0], p[1], p[2]}; }

inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator-(float3 a) { return {-a.x, -a.y, -a.z}; }
inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, float3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator/(float3 a, float3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
// sutil/vec_math.h: operator/(float3, float) multiplies by the reciprocal
inline float3 operator/(float3 a, float s) { float inv = 1.0f / s; return a * inv; }
inline float3 operator/(float s, float3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline float3& operator+=(float3& a, float3 b) { a = a + b; return a; }
inline float3& operator*=(float3& a, float3 b) { a = a * b; return a; }
inline float3& operator*=(float3& a, float s) { a = a * s; return a; }
inline float3& operator/=(float3& a, float s) { a = a / s; return a; }

inline float2 operator+(float2 a, float2 b) { return {a.x + b.x, a.y + b.y}; }
inline float2 operator*(float2 a, float s) { return {a.x * s, a.y * s}; }
inline float2 operator*(float s, float2 a) { return {a.x * s, a.y * s}; }

inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(float3 a, float3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(float3 v) { return sqrtf(dot(v, v)); }
inline float3 normalize(float3 v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }
inline float3 lerp(float3 a, float3 b, float t) { return a + t * (b - a); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
inline int clampi(int f, int a, int b) { return f < a ? a : (f > b ? b : f); }
inline float3 clamp3(float3 v, float a, float b) { return {clampf(v.x, a, b), clampf(v.y, a, b), clampf(v.z, a, b)}; }
inline float fmaxf3(float3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
inline float float3weight(float3 a) { return a.x + a.y + a.z; }  // BDPTVertex.h:124

static const float M_PIf_ = 3.14159265358979323846f;
static const float M_1_PIf_ = 0.318309886183790671538f;

}  // namespace orc

// device_math.h -- CDNA4 device-side value types for the wavefront path tracer.
//
// Arithmetic follows the reference's inline math (core/Geometry.h, core/GNXRayTracer.h) operation for
// operation: the library is compiled with -ffp-contract=off so hipcc does not fuse a*b+c (the x86-64
// reference has no FMA), division and sqrt are IEEE (hipcc default), and libm calls go through
// double-precision OCML and are rounded once to float (gx_sin etc.), which reproduces glibc's
// correctly-rounded float results in all but ~1e-9 of the cases.  Together this keeps GPU paths on
// the same discrete decisions (lobe choice, hit/miss, Russian roulette) as the CPU reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gnxr {

#define GX_DEV __device__ __forceinline__

static constexpr float GX_INF = __builtin_huge_valf();
static constexpr float GX_PI = 3.14159265358979323846f;
static constexpr float GX_INV_PI = 0.31830988618379067154f;
static constexpr float GX_INV_2PI = 0.15915494309189533577f;
static constexpr float GX_PI_OVER_2 = 1.57079632679489661923f;
static constexpr float GX_PI_OVER_4 = 0.78539816339744830961f;
static constexpr float GX_ONE_MINUS_EPS = 0x1.fffffep-1f;
static constexpr float GX_MACH_EPS = 0x1p-24f;       // std::numeric_limits<float>::epsilon() * 0.5
static constexpr float GX_SHADOW_EPS = 0.0001f;
// gamma(n) = (n * MachineEpsilon) / (1 - n * MachineEpsilon), GNXRayTracer.h:354-357 (constant-folded in fp32)
#define GX_GAMMA(n) (((n) * GX_MACH_EPS) / (1 - (n) * GX_MACH_EPS))

// ---- libm through double (see header comment) ----
GX_DEV float gx_sin(float x) { return (float)sin((double)x); }
GX_DEV float gx_cos(float x) { return (float)cos((double)x); }
// sin and cos of the same angle share OCML's argument reduction (same polynomials, same values as the two separate calls)
GX_DEV void gx_sincos(float x, float *s, float *c) { double ds, dc; sincos((double)x, &ds, &dc); *s = (float)ds; *c = (float)dc; }
GX_DEV float gx_tan(float x) { return (float)tan((double)x); }
GX_DEV float gx_acos(float x) { return (float)acos((double)x); }
GX_DEV float gx_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }
GX_DEV float gx_log(float x) { return (float)log((double)x); }
GX_DEV float gx_exp(float x) { return (float)exp((double)x); }
GX_DEV float gx_pow(float x, float y) { return (float)pow((double)x, (double)y); }
// __fsqrt_rn maps to the *native* (not correctly rounded) sqrt in this ROCm; the builtin is IEEE under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.
GX_DEV float gx_sqrt(float x) { return __builtin_sqrtf(x); }

struct V3 {
    float x, y, z;
    GX_DEV V3() : x(0), y(0), z(0) {}
    GX_DEV V3(float x, float y, float z) : x(x), y(y), z(z) {}
    GX_DEV float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
GX_DEV V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
GX_DEV V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
GX_DEV V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
GX_DEV V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
GX_DEV V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
GX_DEV V3 operator/(V3 a, float f) { float inv = 1.f / f; return V3(a.x * inv, a.y * inv, a.z * inv); }  // Geometry.h:206-210
GX_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GX_DEV float absdot(V3 a, V3 b) { return fabsf(dot(a, b)); }
GX_DEV float length_sq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
GX_DEV float length(V3 a) { return gx_sqrt(length_sq(a)); }
GX_DEV V3 normalize(V3 a) { return a / length(a); }
GX_DEV V3 vabs(V3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
GX_DEV V3 cross(V3 a, V3 b) {  // double products, Geometry.h:925-931
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3((float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx)));
}
GX_DEV float max_component(V3 v) { return fmaxf(v.x, fmaxf(v.y, v.z)); }
GX_DEV V3 faceforward(V3 n, V3 v) { return (dot(n, v) < 0.f) ? -n : n; }
GX_DEV bool is_zero(V3 v) { return v.x == 0 && v.y == 0 && v.z == 0; }
GX_DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
GX_DEV float lerpf(float t, float a, float b) { return (1 - t) * a + t * b; }
GX_DEV void coordinate_system(V3 v1, V3 *v2, V3 *v3) {  // Geometry.h:988-995
    if (fabsf(v1.x) > fabsf(v1.y)) *v2 = V3(-v1.z, 0, v1.x) / gx_sqrt(v1.x * v1.x + v1.z * v1.z);
    else *v2 = V3(0, v1.z, -v1.y) / gx_sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3 = cross(v1, *v2);
}

// RGBSpectrum (core/Spectrum.h)
struct Spec {
    float r, g, b;
    GX_DEV Spec() : r(0), g(0), b(0) {}
    GX_DEV explicit Spec(float v) : r(v), g(v), b(v) {}
    GX_DEV Spec(float r, float g, float b) : r(r), g(g), b(b) {}
    GX_DEV bool is_black() const { return r == 0.f && g == 0.f && b == 0.f; }
    GX_DEV float max_value() const { return fmaxf(r, fmaxf(g, b)); }
    GX_DEV float y() const { return 0.212671f * r + 0.715160f * g + 0.072169f * b; }  // Spectrum.h:429-432
};
GX_DEV Spec operator+(Spec a, Spec b) { return Spec(a.r + b.r, a.g + b.g, a.b + b.b); }
GX_DEV Spec operator-(Spec a, Spec b) { return Spec(a.r - b.r, a.g - b.g, a.b - b.b); }
GX_DEV Spec operator*(Spec a, Spec b) { return Spec(a.r * b.r, a.g * b.g, a.b * b.b); }
GX_DEV Spec operator/(Spec a, Spec b) { return Spec(a.r / b.r, a.g / b.g, a.b / b.b); }
GX_DEV Spec operator*(Spec a, float s) { return Spec(a.r * s, a.g * s, a.b * s); }
GX_DEV Spec operator*(float s, Spec a) { return Spec(a.r * s, a.g * s, a.b * s); }
GX_DEV Spec operator/(Spec a, float s) { return Spec(a.r / s, a.g / s, a.b / s); }  // true division, Spectrum.h:146-152
GX_DEV Spec ssqrt(Spec a) { return Spec(gx_sqrt(a.r), gx_sqrt(a.g), gx_sqrt(a.b)); }
GX_DEV Spec slerp(float t, Spec a, Spec b) { return (1 - t) * a + t * b; }
GX_DEV Spec spec3(const float *p) { return Spec(p[0], p[1], p[2]); }

// core/GNXRayTracer.h:179-205
GX_DEV float next_float_up(float v) {
    if (isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = __float_as_uint(v);
    if (v >= 0) ++ui; else --ui;
    return __uint_as_float(ui);
}
GX_DEV float next_float_down(float v) {
    if (isinf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = __float_as_uint(v);
    if (v > 0) --ui; else ++ui;
    return __uint_as_float(ui);
}
// core/Geometry.h:1408-1422
GX_DEV V3 offset_ray_origin(V3 p, V3 pError, V3 n, V3 w) {
    float d = dot(vabs(n), pError);
    V3 offset = d * n;
    if (dot(w, n) < 0) offset = -offset;
    V3 po = p + offset;
    if (offset.x > 0) po.x = next_float_up(po.x); else if (offset.x < 0) po.x = next_float_down(po.x);
    if (offset.y > 0) po.y = next_float_up(po.y); else if (offset.y < 0) po.y = next_float_down(po.y);
    if (offset.z > 0) po.z = next_float_up(po.z); else if (offset.z < 0) po.z = next_float_down(po.z);
    return po;
}

// Transform::operator()(Point3) / (Vector3), Transform.h:196-218, on a row-major float[16]
GX_DEV V3 xform_point(const float *m, V3 p) {
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1) return V3(xp, yp, zp);
    float inv = 1.f / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
GX_DEV V3 xform_vector(const float *m, V3 v) {
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

}  // namespace gnxr

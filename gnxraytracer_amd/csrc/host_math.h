// host_math.h -- host-side value types of the gnxr scene compiler.
//
// The scene compiler (scene_builder.cpp, bvh_build.cpp, tables.cpp) runs once per scene on the CPU
// and produces the flat device tables.  It has to apply the same float arithmetic as the reference's
// own setup code so that device inputs (world-space vertices, camera matrices, light cdfs) carry the
// same bits: Transform ops follow core/Transform.{h,cpp}, bounds follow core/Geometry.h.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace gnxr {

static constexpr float kPi = 3.14159265358979323846f;
static constexpr float kInvPi = 0.31830988618379067154f;
static constexpr float kInv2Pi = 0.15915494309189533577f;
static constexpr float kMachineEpsilon = std::numeric_limits<float>::epsilon() * 0.5f;
inline float gammaf_(int n) { return (n * kMachineEpsilon) / (1 - n * kMachineEpsilon); }  // GNXRayTracer.h:354-357

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() {}
    Vec3(float x, float y, float z) : x(x), y(y), z(z) {}
    float operator[](int i) const { return (&x)[i]; }
    float &operator[](int i) { return (&x)[i]; }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, float f) { float inv = 1.f / f; return {a.x * inv, a.y * inv, a.z * inv}; }  // Geometry.h:206-210
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
inline Vec3 normalize(Vec3 a) { return a / length(a); }
inline Vec3 cross(Vec3 a, Vec3 b) {  // double products, Geometry.h:925-931
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return {(float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx))};
}

struct Mat4 {
    float m[4][4];
    Mat4() { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i][j] = i == j ? 1.f : 0.f; }
};
inline Mat4 mul(const Mat4 &a, const Mat4 &b) {  // Transform.h:52-59
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
inline Mat4 transpose(const Mat4 &a) { Mat4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i]; return r; }
Mat4 inverse(const Mat4 &m);  // Gauss-Jordan, Transform.cpp:54-108

// A transform is (m, mInv) as in core/Transform.h
struct Xf { Mat4 m, inv; };
inline Xf xmul(const Xf &a, const Xf &b) { return {mul(a.m, b.m), mul(b.inv, a.inv)}; }
inline Xf xinverse(const Xf &a) { return {a.inv, a.m}; }
Xf translate(Vec3 d);
Xf scale(float x, float y, float z);
Xf rotate_x(float deg);
Xf rotate_axis(float deg, Vec3 axis);
Xf rotate_y(float deg);
Xf look_at(Vec3 pos, Vec3 look, Vec3 up);
Xf perspective(float fov, float n, float f);
Vec3 xform_point(const Mat4 &m, Vec3 p);   // Transform.h:196-209
Vec3 xform_vector(const Mat4 &m, Vec3 v);  // Transform.h:211-218

struct Box3 {
    Vec3 lo{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    Vec3 hi{std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()};
    void grow(Vec3 p) {
        lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
        hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
    }
    void grow(const Box3 &b) {  // Union(b1, b2), Geometry.h -- also correct for empty boxes
        lo = {std::min(lo.x, b.lo.x), std::min(lo.y, b.lo.y), std::min(lo.z, b.lo.z)};
        hi = {std::max(hi.x, b.hi.x), std::max(hi.y, b.hi.y), std::max(hi.z, b.hi.z)};
    }
    Vec3 diag() const { return hi - lo; }
    float area() const { Vec3 d = diag(); return 2 * (d.x * d.y + d.x * d.z + d.y * d.z); }
    int max_extent() const { Vec3 d = diag(); if (d.x > d.y && d.x > d.z) return 0; else if (d.y > d.z) return 1; else return 2; }
    Vec3 offset(Vec3 p) const {
        Vec3 o = p - lo;
        if (hi.x > lo.x) o.x /= hi.x - lo.x;
        if (hi.y > lo.y) o.y /= hi.y - lo.y;
        if (hi.z > lo.z) o.z /= hi.z - lo.z;
        return o;
    }
    static float lerp1(float t, float a, float b) { return (1 - t) * a + t * b; }
    Vec3 lerp(Vec3 t) const { return {lerp1(t.x, lo.x, hi.x), lerp1(t.y, lo.y, hi.y), lerp1(t.z, lo.z, hi.z)}; }
};

}  // namespace gnxr

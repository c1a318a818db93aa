// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product; nothing under
// gnxraytracer_amd/ may include, link or call this.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg use it, as the checker.
//
// o_math.h: CPU restatement of the reference's value types and helpers
// (core/Geometry.h, core/GNXRayTracer.h, core/Spectrum.h, core/Transform.{h,cpp}).
// Arithmetic order follows the reference so that results agree bit-for-bit with the
// compiled reference (oracle/_ref) on x86-64 SSE2 without FMA.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace gnxo {

typedef float Float;

// core/GNXRayTracer.h:138-150
static constexpr Float Infinity = std::numeric_limits<Float>::infinity();
static constexpr Float MachineEpsilon = std::numeric_limits<Float>::epsilon() * 0.5;
static constexpr Float ShadowEpsilon = 0.0001f;
static constexpr Float Pi = 3.14159265358979323846;
static constexpr Float InvPi = 0.31830988618379067154;
static constexpr Float Inv2Pi = 0.15915494309189533577;
static constexpr Float Inv4Pi = 0.07957747154594766788;
static constexpr Float PiOver2 = 1.57079632679489661923;
static constexpr Float PiOver4 = 0.78539816339744830961;
static constexpr Float OneMinusEpsilon = 0x1.fffffep-1;  // core/RNG.h:17-24

// core/GNXRayTracer.h:354-357
inline Float gamma(int n) { return (n * MachineEpsilon) / (1 - n * MachineEpsilon); }

inline uint32_t FloatToBits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float BitsToFloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// core/GNXRayTracer.h:179-205
inline float NextFloatUp(float v) {
    if (std::isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = FloatToBits(v);
    if (v >= 0) ++ui; else --ui;
    return BitsToFloat(ui);
}
inline float NextFloatDown(float v) {
    if (std::isinf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = FloatToBits(v);
    if (v > 0) --ui; else ++ui;
    return BitsToFloat(ui);
}

template <typename T, typename U, typename V>
inline T Clamp(T val, U low, V high) {
    if (val < low) return low;
    else if (val > high) return high;
    else return val;
}
inline Float Lerp(Float t, Float v1, Float v2) { return (1 - t) * v1 + t * v2; }
inline Float Radians(Float deg) { return (Pi / 180) * deg; }

// One 3-float type stands for Vector3f / Point3f / Normal3f (core/Geometry.h); the
// reference's per-type operator quirks that matter for bits are kept as named functions.
struct V3 {
    Float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(Float x, Float y, Float z) : x(x), y(y), z(z) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    V3 operator+(const V3 &v) const { return V3(x + v.x, y + v.y, z + v.z); }
    V3 operator-(const V3 &v) const { return V3(x - v.x, y - v.y, z - v.z); }
    V3 operator-() const { return V3(-x, -y, -z); }
    V3 operator*(Float s) const { return V3(x * s, y * s, z * s); }
    V3 &operator+=(const V3 &v) { x += v.x; y += v.y; z += v.z; return *this; }
    V3 &operator*=(Float s) { x *= s; y *= s; z *= s; return *this; }
    // Vector3::operator/ multiplies by the reciprocal, core/Geometry.h:206-210
    V3 operator/(Float f) const { Float inv = (Float)1 / f; return V3(x * inv, y * inv, z * inv); }
    bool operator==(const V3 &v) const { return x == v.x && y == v.y && z == v.z; }
    bool operator!=(const V3 &v) const { return x != v.x || y != v.y || z != v.z; }
    Float LengthSquared() const { return x * x + y * y + z * z; }
    Float Length() const { return std::sqrt(LengthSquared()); }
};
inline V3 operator*(Float s, const V3 &v) { return V3(v.x * s, v.y * s, v.z * s); }
inline Float Dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Float AbsDot(const V3 &a, const V3 &b) { return std::abs(Dot(a, b)); }
inline V3 Abs(const V3 &v) { return V3(std::abs(v.x), std::abs(v.y), std::abs(v.z)); }
inline V3 Normalize(const V3 &v) { return v / v.Length(); }
// Cross promotes to double, core/Geometry.h:925-931
inline V3 Cross(const V3 &v1, const V3 &v2) {
    double v1x = v1.x, v1y = v1.y, v1z = v1.z;
    double v2x = v2.x, v2y = v2.y, v2z = v2.z;
    return V3((v1y * v2z) - (v1z * v2y), (v1z * v2x) - (v1x * v2z), (v1x * v2y) - (v1y * v2x));
}
inline Float MaxComponent(const V3 &v) { return std::max(v.x, std::max(v.y, v.z)); }
inline int MaxDimension(const V3 &v) { return (v.x > v.y) ? ((v.x > v.z) ? 0 : 2) : ((v.y > v.z) ? 1 : 2); }
inline V3 Permute(const V3 &v, int x, int y, int z) { return V3(v[x], v[y], v[z]); }
inline V3 Faceforward(const V3 &n, const V3 &v) { return (Dot(n, v) < 0.f) ? -n : n; }
inline Float DistanceSquared(const V3 &a, const V3 &b) { return (a - b).LengthSquared(); }
// core/Geometry.h CoordinateSystem
inline void CoordinateSystem(const V3 &v1, V3 *v2, V3 *v3) {
    if (std::abs(v1.x) > std::abs(v1.y))
        *v2 = V3(-v1.z, 0, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else
        *v2 = V3(0, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3 = Cross(v1, *v2);
}
inline V3 SphericalDirection(Float sinTheta, Float cosTheta, Float phi) {
    return V3(sinTheta * std::cos(phi), sinTheta * std::sin(phi), cosTheta);
}
inline Float SphericalTheta(const V3 &v) { return std::acos(Clamp(v.z, -1, 1)); }
inline Float SphericalPhi(const V3 &v) {
    Float p = std::atan2(v.y, v.x);
    return (p < 0) ? (p + 2 * Pi) : p;
}

struct P2 {
    Float x, y;
    P2() : x(0), y(0) {}
    P2(Float x, Float y) : x(x), y(y) {}
    Float operator[](int i) const { return i == 0 ? x : y; }
};

// core/Geometry.h:1408-1422
inline V3 OffsetRayOrigin(const V3 &p, const V3 &pError, const V3 &n, const V3 &w) {
    Float d = Dot(Abs(n), pError);
    V3 offset = d * n;
    if (Dot(w, n) < 0) offset = -offset;
    V3 po = p + offset;
    for (int i = 0; i < 3; ++i) {
        if (offset[i] > 0) po[i] = NextFloatUp(po[i]);
        else if (offset[i] < 0) po[i] = NextFloatDown(po[i]);
    }
    return po;
}

// RGBSpectrum (Spectrum = RGBSpectrum, core/GNXRayTracer.h:82-86), core/Spectrum.h
struct Spec {
    Float c[3];
    Spec(Float v = 0.f) { c[0] = c[1] = c[2] = v; }
    Spec(Float r, Float g, Float b) { c[0] = r; c[1] = g; c[2] = b; }
    Float operator[](int i) const { return c[i]; }
    Float &operator[](int i) { return c[i]; }
    Spec operator+(const Spec &s) const { return Spec(c[0] + s.c[0], c[1] + s.c[1], c[2] + s.c[2]); }
    Spec operator-(const Spec &s) const { return Spec(c[0] - s.c[0], c[1] - s.c[1], c[2] - s.c[2]); }
    Spec operator*(const Spec &s) const { return Spec(c[0] * s.c[0], c[1] * s.c[1], c[2] * s.c[2]); }
    Spec operator/(const Spec &s) const { return Spec(c[0] / s.c[0], c[1] / s.c[1], c[2] / s.c[2]); }
    Spec operator*(Float a) const { return Spec(c[0] * a, c[1] * a, c[2] * a); }
    Spec operator/(Float a) const { return Spec(c[0] / a, c[1] / a, c[2] / a); }  // Spectrum.h:146-152 (true division)
    Spec &operator+=(const Spec &s) { c[0] += s.c[0]; c[1] += s.c[1]; c[2] += s.c[2]; return *this; }
    Spec &operator*=(const Spec &s) { c[0] *= s.c[0]; c[1] *= s.c[1]; c[2] *= s.c[2]; return *this; }
    Spec &operator*=(Float a) { c[0] *= a; c[1] *= a; c[2] *= a; return *this; }
    Spec &operator/=(Float a) { c[0] /= a; c[1] /= a; c[2] /= a; return *this; }
    Spec operator-() const { return Spec(-c[0], -c[1], -c[2]); }
    bool IsBlack() const { return c[0] == 0. && c[1] == 0. && c[2] == 0.; }
    Spec Clamp(Float low = 0, Float high = Infinity) const {
        return Spec(gnxo::Clamp(c[0], low, high), gnxo::Clamp(c[1], low, high), gnxo::Clamp(c[2], low, high));
    }
    Float MaxComponentValue() const { return std::max(c[0], std::max(c[1], c[2])); }
    // Spectrum.h:429-432
    Float y() const { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; }
};
inline Spec operator*(Float a, const Spec &s) { return s * a; }
inline Spec Sqrt(const Spec &s) { return Spec(std::sqrt(s.c[0]), std::sqrt(s.c[1]), std::sqrt(s.c[2])); }
inline Spec Lerp(Float t, const Spec &a, const Spec &b) { return (1 - t) * a + t * b; }

// ---- 4x4 matrices and the Transform constructors used on the path (core/Transform.cpp) ----
struct M44 {
    Float m[4][4];
    M44() { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i][j] = (i == j) ? 1.f : 0.f; }
    static M44 FromRowMajor(const float *p) { M44 r; memcpy(r.m, p, 64); return r; }
};
// Transform.h:52-59
inline M44 Mul(const M44 &a, const M44 &b) {
    M44 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
inline M44 Transpose(const M44 &a) { M44 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i]; return r; }
// Transform.cpp:54-108 (Gauss-Jordan with full pivoting)
inline M44 Inverse(const M44 &mm) {
    int indxc[4], indxr[4];
    int ipiv[4] = {0, 0, 0, 0};
    Float minv[4][4];
    memcpy(minv, mm.m, 64);
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        Float big = 0.f;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] != 1) {
                for (int k = 0; k < 4; k++) {
                    if (ipiv[k] == 0) {
                        if (std::abs(minv[j][k]) >= big) { big = Float(std::abs(minv[j][k])); irow = j; icol = k; }
                    }
                }
            }
        }
        ++ipiv[icol];
        if (irow != icol) for (int k = 0; k < 4; ++k) std::swap(minv[irow][k], minv[icol][k]);
        indxr[i] = irow;
        indxc[i] = icol;
        Float pivinv = 1. / minv[icol][icol];
        minv[icol][icol] = 1.;
        for (int j = 0; j < 4; j++) minv[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j != icol) {
                Float save = minv[j][icol];
                minv[j][icol] = 0;
                for (int k = 0; k < 4; k++) minv[j][k] -= minv[icol][k] * save;
            }
        }
    }
    for (int j = 3; j >= 0; j--) {
        if (indxr[j] != indxc[j]) for (int k = 0; k < 4; k++) std::swap(minv[k][indxr[j]], minv[k][indxc[j]]);
    }
    M44 r; memcpy(r.m, minv, 64); return r;
}
struct Xform { M44 m, mInv; };
inline Xform XMul(const Xform &a, const Xform &b) { Xform r; r.m = Mul(a.m, b.m); r.mInv = Mul(b.mInv, a.mInv); return r; }
inline Xform XInverse(const Xform &a) { Xform r; r.m = a.mInv; r.mInv = a.m; return r; }
inline Xform Translate(const V3 &d) {
    Xform t;
    t.m.m[0][3] = d.x; t.m.m[1][3] = d.y; t.m.m[2][3] = d.z;
    t.mInv.m[0][3] = -d.x; t.mInv.m[1][3] = -d.y; t.mInv.m[2][3] = -d.z;
    return t;
}
inline Xform Scale(Float x, Float y, Float z) {
    Xform t;
    t.m.m[0][0] = x; t.m.m[1][1] = y; t.m.m[2][2] = z;
    t.mInv.m[0][0] = 1 / x; t.mInv.m[1][1] = 1 / y; t.mInv.m[2][2] = 1 / z;
    return t;
}
// Transform.cpp:181-215
inline Xform LookAt(const V3 &pos, const V3 &look, const V3 &up) {
    M44 c2w;
    c2w.m[0][3] = pos.x; c2w.m[1][3] = pos.y; c2w.m[2][3] = pos.z; c2w.m[3][3] = 1;
    V3 dir = Normalize(look - pos);
    V3 right = Normalize(Cross(Normalize(up), dir));
    V3 newUp = Cross(dir, right);
    c2w.m[0][0] = right.x; c2w.m[1][0] = right.y; c2w.m[2][0] = right.z; c2w.m[3][0] = 0.;
    c2w.m[0][1] = newUp.x; c2w.m[1][1] = newUp.y; c2w.m[2][1] = newUp.z; c2w.m[3][1] = 0.;
    c2w.m[0][2] = dir.x; c2w.m[1][2] = dir.y; c2w.m[2][2] = dir.z; c2w.m[3][2] = 0.;
    Xform r; r.m = Inverse(c2w); r.mInv = c2w; return r;
}
// Transform.cpp:287-296
inline Xform Perspective(Float fov, Float n, Float f) {
    M44 persp;
    persp.m[2][2] = f / (f - n); persp.m[2][3] = -f * n / (f - n);
    persp.m[3][2] = 1; persp.m[3][3] = 0;
    Float invTanAng = 1 / std::tan(Radians(fov) / 2);
    Xform p; p.m = persp; p.mInv = Inverse(persp);
    return XMul(Scale(invTanAng, invTanAng, 1), p);
}
// Transform.h:196-209 (Point3)
inline V3 XPoint(const M44 &m, const V3 &p) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
    Float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
    Float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
    Float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
    if (wp == 1) return V3(xp, yp, zp);
    Float inv = (Float)1 / wp;  // Point3::operator/, Geometry.h:461-465
    return V3(inv * xp, inv * yp, inv * zp);
}
// Transform.h:259-283 (Point3 with error)
inline V3 XPointErr(const M44 &m, const V3 &p, V3 *pError) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = (m.m[0][0] * x + m.m[0][1] * y) + (m.m[0][2] * z + m.m[0][3]);
    Float yp = (m.m[1][0] * x + m.m[1][1] * y) + (m.m[1][2] * z + m.m[1][3]);
    Float zp = (m.m[2][0] * x + m.m[2][1] * y) + (m.m[2][2] * z + m.m[2][3]);
    Float wp = (m.m[3][0] * x + m.m[3][1] * y) + (m.m[3][2] * z + m.m[3][3]);
    Float xAbsSum = (std::abs(m.m[0][0] * x) + std::abs(m.m[0][1] * y) + std::abs(m.m[0][2] * z) + std::abs(m.m[0][3]));
    Float yAbsSum = (std::abs(m.m[1][0] * x) + std::abs(m.m[1][1] * y) + std::abs(m.m[1][2] * z) + std::abs(m.m[1][3]));
    Float zAbsSum = (std::abs(m.m[2][0] * x) + std::abs(m.m[2][1] * y) + std::abs(m.m[2][2] * z) + std::abs(m.m[2][3]));
    *pError = gamma(3) * V3(xAbsSum, yAbsSum, zAbsSum);
    if (wp == 1) return V3(xp, yp, zp);
    Float inv = (Float)1 / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
// Transform.h:211-218 (Vector3)
inline V3 XVector(const M44 &m, const V3 &v) {
    Float x = v.x, y = v.y, z = v.z;
    return V3(m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z, m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z,
              m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z);
}

struct Ray {
    V3 o, d;
    mutable Float tMax;
    int medium;  // index into scene media, -1 == none (Ray::medium, Geometry.h:851)
    Ray() : tMax(Infinity), medium(-1) {}
    Ray(const V3 &o, const V3 &d, Float tMax = Infinity, int medium = -1) : o(o), d(d), tMax(tMax), medium(medium) {}
    V3 operator()(Float t) const { return o + d * t; }
    // RayDifferential, core/Geometry.h:855-892: only camera rays and the specular children of Whitted / DirectLighting carry
    // differentials; every other ray is a plain Ray (hasDifferentials == false)
    bool hasDifferentials = false;
    V3 rxOrigin, ryOrigin, rxDirection, ryDirection;
    void ScaleDifferentials(Float s) {   // Geometry.h:874-880
        rxOrigin = o + (rxOrigin - o) * s;
        ryOrigin = o + (ryOrigin - o) * s;
        rxDirection = d + (rxDirection - d) * s;
        ryDirection = d + (ryDirection - d) * s;
    }
};

struct Bounds3 {
    V3 pMin, pMax;
    Bounds3() : pMin(std::numeric_limits<Float>::max(), std::numeric_limits<Float>::max(), std::numeric_limits<Float>::max()),
                pMax(std::numeric_limits<Float>::lowest(), std::numeric_limits<Float>::lowest(), std::numeric_limits<Float>::lowest()) {}
    Bounds3(const V3 &p) : pMin(p), pMax(p) {}
    Bounds3(const V3 &p1, const V3 &p2)
        : pMin(std::min(p1.x, p2.x), std::min(p1.y, p2.y), std::min(p1.z, p2.z)),
          pMax(std::max(p1.x, p2.x), std::max(p1.y, p2.y), std::max(p1.z, p2.z)) {}
    const V3 &operator[](int i) const { return i == 0 ? pMin : pMax; }
    V3 Diagonal() const { return pMax - pMin; }
    Float SurfaceArea() const { V3 d = Diagonal(); return 2 * (d.x * d.y + d.x * d.z + d.y * d.z); }
    int MaximumExtent() const {
        V3 d = Diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        else if (d.y > d.z) return 1;
        else return 2;
    }
    V3 Lerp(const V3 &t) const {
        return V3(gnxo::Lerp(t.x, pMin.x, pMax.x), gnxo::Lerp(t.y, pMin.y, pMax.y), gnxo::Lerp(t.z, pMin.z, pMax.z));
    }
    V3 Offset(const V3 &p) const {
        V3 o = p - pMin;
        if (pMax.x > pMin.x) o.x /= pMax.x - pMin.x;
        if (pMax.y > pMin.y) o.y /= pMax.y - pMin.y;
        if (pMax.z > pMin.z) o.z /= pMax.z - pMin.z;
        return o;
    }
};
inline Bounds3 Union(const Bounds3 &b, const V3 &p) {
    Bounds3 r;
    r.pMin = V3(std::min(b.pMin.x, p.x), std::min(b.pMin.y, p.y), std::min(b.pMin.z, p.z));
    r.pMax = V3(std::max(b.pMax.x, p.x), std::max(b.pMax.y, p.y), std::max(b.pMax.z, p.z));
    return r;
}
inline Bounds3 Union(const Bounds3 &a, const Bounds3 &b) {
    Bounds3 r;
    r.pMin = V3(std::min(a.pMin.x, b.pMin.x), std::min(a.pMin.y, b.pMin.y), std::min(a.pMin.z, b.pMin.z));
    r.pMax = V3(std::max(a.pMax.x, b.pMax.x), std::max(a.pMax.y, b.pMax.y), std::max(a.pMax.z, b.pMax.z));
    return r;
}
// core/Geometry.h:1380-1406
inline bool SlabTest(const Bounds3 &bounds, const Ray &ray, const V3 &invDir, const int dirIsNeg[3]) {
    Float tMin = (bounds[dirIsNeg[0]].x - ray.o.x) * invDir.x;
    Float tMax = (bounds[1 - dirIsNeg[0]].x - ray.o.x) * invDir.x;
    Float tyMin = (bounds[dirIsNeg[1]].y - ray.o.y) * invDir.y;
    Float tyMax = (bounds[1 - dirIsNeg[1]].y - ray.o.y) * invDir.y;
    tMax *= 1 + 2 * gamma(3);
    tyMax *= 1 + 2 * gamma(3);
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    Float tzMin = (bounds[dirIsNeg[2]].z - ray.o.z) * invDir.z;
    Float tzMax = (bounds[1 - dirIsNeg[2]].z - ray.o.z) * invDir.z;
    tzMax *= 1 + 2 * gamma(3);
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    return (tMin < ray.tMax) && (tMax > 0);
}
// core/Geometry.h:1356-1377  Bounds3::IntersectP(ray, t0, t1)
inline bool BoundsIntersectP(const Bounds3 &b, const Ray &ray, Float *hitt0, Float *hitt1) {
    Float t0 = 0, t1 = ray.tMax;
    for (int i = 0; i < 3; ++i) {
        Float invRayDir = 1 / ray.d[i];
        Float tNear = (b.pMin[i] - ray.o[i]) * invRayDir;
        Float tFar = (b.pMax[i] - ray.o[i]) * invRayDir;
        if (tNear > tFar) std::swap(tNear, tFar);
        tFar *= 1 + 2 * gamma(3);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return false;
    }
    if (hitt0) *hitt0 = t0;
    if (hitt1) *hitt1 = t1;
    return true;
}

}  // namespace gnxo

// device_lights.h -- light sampling on the device.
//   DiffuseAreaLight::L / Sample_Li / Pdf_Li        lights/DiffuseAreaLight.{h,cpp}:22-27,37-57
//   Shape::Sample(ref,u,pdf) / Pdf(ref,wi)          core/Shape.cpp:21-53
//   Triangle::Sample                                shape/Triangle.cpp:464-492
//   InfiniteAreaLight::Le / Sample_Li / Pdf_Li      lights/InfiniteAreaLight.cpp:91-132
//   MIPMap::Lookup -> triangle(0, st), Repeat wrap  core/MIPMap.h:225-256
//   Distribution1D / 2D sampling                    core/Sampling.h:37-70, 101-122
//   SkyBoxLight::Le / Sample_Li (no image)          lights/SkyBoxLight.cpp:43-85
//   LightDistribution::Lookup + SampleDiscrete      core/LightDistribution.cpp:109-204 (dense table)
#pragma once
#include "device_geom.h"
#include "device_sampler.h"

namespace gnxr {

struct DLightTables {
    const DLight *lights;
    int n_lights;
    const int32_t *infinite;   // indices of lights flagged Infinite
    int n_infinite;
    DLightGrid grid;
    const float *grid_table;
    // env
    int has_env;
    DEnv env;
    const float4 *env_texels;   // level-0 texels of the environment map, rgb_ (one dwordx4 each)
    const float *env_cond_func, *env_cond_cdf, *env_cond_int, *env_marg_func, *env_marg_cdf;
    const uint16_t *env_marg_guide, *env_cond_guide;   // FindInterval guide tables (scene_compile.cpp build_env)
};

struct LightSample {
    Spec Li;
    V3 wi;
    float pdf;
    V3 p1, p1Error, n1;   // VisibilityTester end point (Interaction p / pError / n)
};

// DiffuseAreaLight::L: `bool dotNW = Dot(intr.n, w)` truncates to bool (DiffuseAreaLight.h:24), so any
// non-zero (or NaN) dot product emits.
GX_DEV Spec area_L(const DLight &l, V3 n, V3 w) {
    float d = dot(n, w);
    bool dotNW = (d != 0.f);  // float -> bool conversion
    return (l.two_sided || dotNW) ? spec3(l.le) : Spec(0.f);
}

GX_DEV int find_interval(const float *cdf, int size, float u) {  // GNXRayTracer.h:336-349, pred = cdf[i] <= u
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    return min(max(first - 1, 0), size - 2);
}
// the same search started from a guide table: guide[b] = number of cdf entries <= b / G, so for u in bucket b the partition point lies in
// [guide[b], guide[b + 1]] and the bisection (same predicate) runs over those few entries only
GX_DEV int find_interval_guided(const float *cdf, int size, float u, const uint16_t *guide, int G) {
    const int b = min(max((int)(u * (float)G), 0), G - 1);
    int first = guide[b], len = (int)guide[b + 1] - first;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    return min(max(first - 1, 0), size - 2);
}
GX_DEV float dist1d_sample_continuous(const float *func, const float *cdf, int n, float funcInt, float u, float *pdf, int *off, const uint16_t *guide = nullptr, int G = 0) {
    int offset = guide ? find_interval_guided(cdf, n + 1, u, guide, G) : find_interval(cdf, n + 1, u);
    *off = offset;
    float du = u - cdf[offset];
    if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
    *pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
    return (offset + du) / n;
}

GX_DEV int modi(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }
GX_DEV Spec env_lookup(const DLightTables &t, float s_, float t_) {  // MIPMap::triangle(0, st), MIPMap.h:244-256
    int rx = t.env.w, ry = t.env.h;
    float s = s_ * rx - 0.5f, tt = t_ * ry - 0.5f;
    int s0 = (int)floorf(s), t0 = (int)floorf(tt);
    float ds = s - s0, dt = tt - t0;
    const float4 *tex = t.env_texels;
    int sa = modi(s0, rx), sb = modi(s0 + 1, rx), ta = modi(t0, ry), tb = modi(t0 + 1, ry);
    const float4 q00 = tex[(size_t)ta * rx + sa], q01 = tex[(size_t)tb * rx + sa], q10 = tex[(size_t)ta * rx + sb], q11 = tex[(size_t)tb * rx + sb];
    Spec c00(q00.x, q00.y, q00.z), c01(q01.x, q01.y, q01.z), c10(q10.x, q10.y, q10.z), c11(q11.x, q11.y, q11.z);
    return (1 - ds) * (1 - dt) * c00 + (1 - ds) * dt * c01 + ds * (1 - dt) * c10 + ds * dt * c11;
}
GX_DEV float spherical_theta(V3 v) { return gx_acos(clampf(v.z, -1, 1)); }
GX_DEV float spherical_phi(V3 v) { float p = gx_atan2(v.y, v.x); return (p < 0) ? (p + 2 * GX_PI) : p; }

GX_DEV Spec env_Le(const DLightTables &t, V3 rd) {  // InfiniteAreaLight.cpp:91-96
    V3 w = normalize(xform_vector(t.env.w2l, rd));
    return env_lookup(t, spherical_phi(w) * GX_INV_2PI, spherical_theta(w) * GX_INV_PI);
}
GX_DEV Spec skybox_Le(const DLight &l, V3 ro, V3 rd) {  // SkyBoxLight.cpp:55-85, data == nullptr
    V3 center(l.center[0], l.center[1], l.center[2]);
    float R = l.radius;
    V3 oc = ro - center;
    float a = dot(rd, rd);
    float b = (float)(2.0 * (double)dot(oc, rd));
    float c = dot(oc, oc) - R * R;
    float disc = b * b - 4 * a * c;
    if (disc < 0) return Spec(0.f);
    // float sqrt (SkyBoxLight.cpp sees <math.h> via stb_image.h), double division by `2.0 * a`
    float tt = (float)((double)(-b + gx_sqrt(disc)) / (2.0 * (double)a));
    V3 hp = ro + tt * rd;
    V3 q = hp - center;
    return Spec((q.x + R) / (2.f * R), (q.y + R) / (2.f * R), (q.z + R) / (2.f * R));
}
// LT: compile-time mask of the light types a scene holds (bit 0 area triangles, bit 1 InfiniteAreaLight,
// bit 2 SkyBoxLight, bit 3 the delta lights Point / Spot / Distant); unused types are compiled out of the shade kernels.
constexpr int LT_AREA = 1, LT_ENV = 2, LT_SKY = 4, LT_DELTA = 8, LT_ALL = 15;
// IsDeltaLight(light.flags), core/Light.h:28-32
template <int LT>
GX_DEV bool light_is_delta(const DLight &l) { return (LT & LT_DELTA) && l.type >= GNXR_LIGHT_POINT; }

// Light::Le(ray) for an escaped ray
template <int LT>
GX_DEV Spec light_Le(const DLightTables &t, int li, V3 ro, V3 rd) {
    const DLight &l = t.lights[li];
    if ((LT & LT_ENV) && l.type == GNXR_LIGHT_INFINITE) return env_Le(t, rd);
    if ((LT & LT_SKY) && l.type == GNXR_LIGHT_SKYBOX) return skybox_Le(l, ro, rd);
    return Spec(0.f);
}

// <Light>::Sample_Li(ref, u)
template <int LT>
GX_DEV LightSample light_sample(const DLightTables &t, int li, V3 refP, float u0, float u1) {
    const DLight &l = t.lights[li];
    LightSample s;
    s.pdf = 0; s.Li = Spec(0.f);
    if ((LT & LT_AREA) && (LT == LT_AREA || l.type == GNXR_LIGHT_AREA_TRI)) {
        // Triangle::Sample(u, pdf), Triangle.cpp:464-492
        float su0 = gx_sqrt(u0);
        float b0 = 1 - su0, b1 = u1 * su0;
        V3 p0(l.p0[0], l.p0[1], l.p0[2]), p1(l.p1[0], l.p1[1], l.p1[2]), p2(l.p2[0], l.p2[1], l.p2[2]);
        V3 p = b0 * p0 + b1 * p1 + (1 - b0 - b1) * p2;
        V3 n(l.n[0], l.n[1], l.n[2]);
        V3 pAbsSum = vabs(b0 * p0) + vabs(b1 * p1) + vabs((1 - b0 - b1) * p2);
        V3 pError = GX_GAMMA(6) * pAbsSum;
        float pdf = 1 / l.area;
        // Shape::Sample(ref, u, pdf), Shape.cpp:21-35
        V3 wi = p - refP;
        if (length_sq(wi) == 0) pdf = 0;
        else {
            wi = normalize(wi);
            pdf *= length_sq(refP - p) / absdot(n, -wi);
            if (isinf(pdf)) pdf = 0.f;
        }
        // DiffuseAreaLight::Sample_Li, DiffuseAreaLight.cpp:37-52
        if (pdf == 0 || length_sq(p - refP) == 0) return s;
        s.wi = normalize(p - refP);
        s.pdf = pdf;
        s.p1 = p; s.p1Error = pError; s.n1 = n;
        s.Li = area_L(l, n, -s.wi);
        return s;
    } else if ((LT & LT_ENV) && l.type == GNXR_LIGHT_INFINITE) {  // InfiniteAreaLight.cpp:98-121
        const DEnv &e = t.env;
        float pdfs0, pdfs1;
        int v, dummy;
        float d1 = dist1d_sample_continuous(t.env_marg_func, t.env_marg_cdf, e.dh, e.marg_func_int, u1, &pdfs1, &v, t.env_marg_guide, kEnvGuideMarg);
        float d0 = dist1d_sample_continuous(t.env_cond_func + (size_t)v * e.dw, t.env_cond_cdf + (size_t)v * (e.dw + 1), e.dw, t.env_cond_int[v], u0, &pdfs0, &dummy,
                                            t.env_cond_guide + (size_t)v * (kEnvGuideCond + 1), kEnvGuideCond);
        float mapPdf = pdfs0 * pdfs1;
        if (mapPdf == 0) return s;
        float theta = d1 * GX_PI, phi = d0 * 2 * GX_PI;
        float cosTheta, sinTheta, sinPhi, cosPhi;
        gx_sincos(theta, &sinTheta, &cosTheta);
        gx_sincos(phi, &sinPhi, &cosPhi);
        s.wi = xform_vector(e.l2w, V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
        s.pdf = mapPdf / (2 * GX_PI * GX_PI * sinTheta);
        if (sinTheta == 0) s.pdf = 0;
        s.p1 = refP + s.wi * (2 * e.world_radius);
        s.p1Error = V3(); s.n1 = V3();
        s.Li = env_lookup(t, d0, d1);
        return s;
    } else if ((LT & LT_DELTA) && l.type >= GNXR_LIGHT_POINT) {
        // PointLight.cpp:13-22, SpotLight.cpp:19-40, DistantLight.cpp:15-25.  DLight reuse: p0 = pLight, n = wLight (world),
        // radius = worldRadius, area / inv_area = cosTotalWidth / cosFalloffStart, p1 / p2 / center = rows of WorldToLight's 3x3
        const Spec I(l.le[0], l.le[1], l.le[2]);
        s.pdf = 1.f;
        s.p1Error = V3(); s.n1 = V3();
        if (l.type == GNXR_LIGHT_DISTANT) {
            V3 w(l.n[0], l.n[1], l.n[2]);
            s.wi = w;
            s.p1 = refP + w * (2 * l.radius);
            s.Li = I;
        } else {
            V3 pL(l.p0[0], l.p0[1], l.p0[2]);
            s.wi = normalize(pL - refP);
            s.p1 = pL;
            if (l.type == GNXR_LIGHT_SPOT) {
                V3 w = -s.wi;
                V3 wl = normalize(V3(l.p1[0] * w.x + l.p1[1] * w.y + l.p1[2] * w.z, l.p2[0] * w.x + l.p2[1] * w.y + l.p2[2] * w.z,
                                     l.center[0] * w.x + l.center[1] * w.y + l.center[2] * w.z));
                float cosTheta = wl.z, falloff;
                if (cosTheta < l.area) falloff = 0;
                else if (cosTheta >= l.inv_area) falloff = 1;
                else {
                    float delta = (cosTheta - l.area) / (l.inv_area - l.area);
                    falloff = (delta * delta) * (delta * delta);
                }
                s.Li = I * falloff / length_sq(pL - refP);
            } else s.Li = I / length_sq(pL - refP);
        }
        return s;
    } else if ((LT & LT_SKY) && l.type == GNXR_LIGHT_SKYBOX) {  // SkyBoxLight::Sample_Li, SkyBoxLight.cpp:43-53: black, pdf 1/4pi
        float theta = u1 * GX_PI, phi = u0 * 2 * GX_PI;
        float cosTheta, sinTheta, sinPhi, cosPhi;
        gx_sincos(theta, &sinTheta, &cosTheta);
        gx_sincos(phi, &sinPhi, &cosPhi);
        s.wi = V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
        s.pdf = 1.f / (4 * GX_PI);
        s.p1 = refP + s.wi * (2 * l.radius);
        s.p1Error = V3(); s.n1 = V3();
        s.Li = Spec(0.f);
        return s;
    }
    return s;
}

// <Light>::Pdf_Li(ref, wi); the reference point's (p, pError, n) are needed for Shape::Pdf's SpawnRay
template <int LT>
GX_DEV float light_pdf(const DLightTables &t, int li, V3 refP, V3 refPError, V3 refN, V3 wi) {
    const DLight &l = t.lights[li];
    if ((LT & LT_AREA) && (LT == LT_AREA || l.type == GNXR_LIGHT_AREA_TRI)) {  // Shape::Pdf(ref, wi), Shape.cpp:37-53
        V3 o = offset_ray_origin(refP, refPError, refN, wi);
        V3 p0(l.p0[0], l.p0[1], l.p0[2]), p1(l.p1[0], l.p1[1], l.p1[2]), p2(l.p2[0], l.p2[1], l.p2[2]);
        TriHit h;
        if (!tri_test(p0, p1, p2, o, wi, GX_INF, &h)) return 0.f;
        // the full Triangle::Intersect can still reject a degenerate triangle; light triangles have area > 0
        V3 pHit = h.b0 * p0 + h.b1 * p1 + h.b2 * p2;
        V3 nHit = normalize(cross(p0 - p2, p1 - p2));
        float pdf = length_sq(refP - pHit) / (absdot(nHit, -wi) * l.area);
        if (isinf(pdf)) pdf = 0.f;
        return pdf;
    } else if ((LT & LT_ENV) && l.type == GNXR_LIGHT_INFINITE) {  // InfiniteAreaLight.cpp:123-132
        const DEnv &e = t.env;
        V3 w = xform_vector(e.w2l, wi);
        float theta = spherical_theta(w), phi = spherical_phi(w);
        float sinTheta = gx_sin(theta);
        if (sinTheta == 0) return 0.f;
        float px = phi * GX_INV_2PI, py = theta * GX_INV_PI;
        int iu = min(max((int)(px * e.dw), 0), e.dw - 1);
        int iv = min(max((int)(py * e.dh), 0), e.dh - 1);
        float mapPdf = t.env_cond_func[(size_t)iv * e.dw + iu] / e.marg_func_int;
        return mapPdf / (2 * GX_PI * GX_PI * sinTheta);
    }
    return 0.f;
}

// lightDistribution->Lookup(p) + Distribution1D::SampleDiscrete(u, &pdf)
GX_DEV int light_select(const DLightTables &t, V3 p, float u, float *pdf) {
    const DLightGrid &g = t.grid;
    const float *rec = t.grid_table;
    int nl = g.n_lights;
    if (g.spatial) {
        // Bounds3::Offset + voxel clamp, LightDistribution.cpp:113-116
        V3 o(p.x - g.lo[0], p.y - g.lo[1], p.z - g.lo[2]);
        if (g.hi[0] > g.lo[0]) o.x /= g.hi[0] - g.lo[0];
        if (g.hi[1] > g.lo[1]) o.y /= g.hi[1] - g.lo[1];
        if (g.hi[2] > g.lo[2]) o.z /= g.hi[2] - g.lo[2];
        int px = min(max((int)(o.x * g.nvox[0]), 0), g.nvox[0] - 1);
        int py = min(max((int)(o.y * g.nvox[1]), 0), g.nvox[1] - 1);
        int pz = min(max((int)(o.z * g.nvox[2]), 0), g.nvox[2] - 1);
        rec += (((size_t)px * g.nvox[1] + py) * g.nvox[2] + pz) * g.stride;
    }
    if (nl <= 3 && (g.stride & 3) == 0) {
        // padded record (build_light_grid): the whole of it in one or two aligned loads, then the same search on registers
        const float4 a = *reinterpret_cast<const float4 *>(rec);
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nl >= 2) b = *reinterpret_cast<const float4 *>(rec + 4);
        float c0 = a.x, c1 = 0.f, c2 = 0.f, f0, f1 = 0.f, f2 = 0.f, fi;
        if (nl == 1) { f0 = a.y; fi = a.z; }
        else if (nl == 2) { c1 = a.y; f0 = a.z; f1 = a.w; fi = b.x; }
        else { c1 = a.y; c2 = a.z; f0 = a.w; f1 = b.x; f2 = b.y; fi = b.z; }
        int first = 0, len = nl + 1;
        while (len > 0) {
            int half = len >> 1, middle = first + half;
            float c = (middle == 0) ? 0.f : (middle == 1 ? c0 : (middle == 2 ? c1 : c2));
            if (c <= u) { first = middle + 1; len -= half + 1; }
            else len = half;
        }
        int offset = min(max(first - 1, 0), nl - 1);
        float func = offset == 0 ? f0 : (offset == 1 ? f1 : f2);
        *pdf = (fi > 0) ? func / (fi * nl) : 0;
        return offset;
    }
    // FindInterval over cdf[0..nl] where cdf[0] = 0 and rec[i] = cdf[i+1]
    int first = 0, len = nl + 1;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        float c = (middle == 0) ? 0.f : rec[middle - 1];
        if (c <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int offset = min(max(first - 1, 0), nl - 1);
    float func = rec[nl + offset], funcInt = rec[2 * nl];
    *pdf = (funcInt > 0) ? func / (funcInt * nl) : 0;
    return offset;
}

}  // namespace gnxr

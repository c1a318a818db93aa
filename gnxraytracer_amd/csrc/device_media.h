// device_media.h -- participating media on the device (BASELINE config 5).
//   HomogeneousMedium::Tr / Sample               media/HomogeneousMedium.cpp:11-43
//   GridDensityMedium::Density / Sample / Tr     media/GridDensityMedium.cpp:14-87   (delta / ratio tracking)
//   HenyeyGreenstein::p / Sample_p, PhaseHG      core/Medium.cpp:164-189, core/Medium.h:34-38
//   Transform::operator()(Ray)                   core/Transform.h:230-244
//   Bounds3::IntersectP(ray, t0, t1)             core/Geometry.h:1358-1378
// The tracking loops draw from the path's Halton stream, so the number of dimensions a segment consumes
// depends on the data; the stream position is carried per path (vol_kernel.hip.h).
#pragma once
#include "device_sampler.h"

namespace gnxr {

struct DMediaTables {
    const DMedium *media;
    const float *density;       // all grids, DMedium::density_offset
    const int2 *tri_media;      // per leaf-order triangle: (inside, outside); nullptr == no medium boundaries
};

static constexpr float GX_INV_4PI = 0.07957747154594766788f;
static constexpr float GX_MAX_FLOAT = 3.402823466e+38f;

GX_DEV Spec spec_exp(Spec s) { return Spec(gx_exp(s.r), gx_exp(s.g), gx_exp(s.b)); }

// core/Medium.h:34-38
GX_DEV float phase_hg(float cosTheta, float g) {
    float denom = 1 + g * g + 2 * g * cosTheta;
    return GX_INV_4PI * (1 - g * g) / (denom * gx_sqrt(denom));
}
// core/Medium.cpp:170-189; SphericalDirection(sinTheta, cosTheta, phi, x, y, z), Geometry.h:1429-1434
GX_DEV float hg_sample_p(float g, V3 wo, V3 *wi, float u0, float u1) {
    float cosTheta;
    if ((double)fabsf(g) < 1e-3) cosTheta = 1 - 2 * u0;
    else {
        float sqrTerm = (1 - g * g) / (1 + g - 2 * g * u0);
        cosTheta = -(1 + g * g - sqrTerm * sqrTerm) / (2 * g);
    }
    float sinTheta = gx_sqrt(fmaxf(0.f, 1 - cosTheta * cosTheta));
    float phi = 2 * GX_PI * u1;
    V3 v1, v2;
    coordinate_system(wo, &v1, &v2);
    float sinPhi, cosPhi;
    gx_sincos(phi, &sinPhi, &cosPhi);
    *wi = sinTheta * cosPhi * v1 + sinTheta * sinPhi * v2 + cosTheta * wo;
    return phase_hg(cosTheta, g);
}

// Transform::operator()(const Ray &), Transform.h:230-244 with (*this)(r.o, &oError), Transform.h:259-283
GX_DEV void xform_ray(const float *m, V3 ro, V3 rd, float tMaxIn, V3 *o2, V3 *d2, float *tMax2) {
    float x = ro.x, y = ro.y, z = ro.z;
    float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    float wp = (m[12] * x + m[13] * y) + (m[14] * z + m[15]);
    float xAbs = (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
    float yAbs = (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
    float zAbs = (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    V3 oError = GX_GAMMA(3) * V3(xAbs, yAbs, zAbs);
    V3 o = (wp == 1) ? V3(xp, yp, zp) : V3((1.f / wp) * xp, (1.f / wp) * yp, (1.f / wp) * zp);
    V3 d = xform_vector(m, rd);
    float lengthSquared = length_sq(d);
    float tMax = tMaxIn;
    if (lengthSquared > 0) {
        float dt = dot(vabs(d), oError) / lengthSquared;
        o = o + d * dt;
        tMax -= dt;
    }
    *o2 = o; *d2 = d; *tMax2 = tMax;
}

// Bounds3(0,0,0 .. 1,1,1).IntersectP(ray, &t0, &t1), Geometry.h:1358-1378
GX_DEV bool unit_box_intersect(V3 o, V3 d, float tMaxRay, float *hitt0, float *hitt1) {
    float t0 = 0, t1 = tMaxRay;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float invRayDir = 1 / d[i];
        float tNear = (0.f - o[i]) * invRayDir;
        float tFar = (1.f - o[i]) * invRayDir;
        if (tNear > tFar) { float tmp = tNear; tNear = tFar; tFar = tmp; }
        tFar *= 1 + 2 * GX_GAMMA(3);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return false;
    }
    *hitt0 = t0; *hitt1 = t1;
    return true;
}

// GridDensityMedium::D / Density, GridDensityMedium.h:34-38, GridDensityMedium.cpp:14-29
GX_DEV float grid_D(const DMedium &m, const float *__restrict__ d, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= m.nx || y >= m.ny || z >= m.nz) return 0;
    return d[(z * m.ny + y) * m.nx + x];
}
GX_DEV float grid_density(const DMedium &m, const float *__restrict__ d, V3 p) {
    V3 ps(p.x * (float)m.nx - .5f, p.y * (float)m.ny - .5f, p.z * (float)m.nz - .5f);
    int px = (int)floorf(ps.x), py = (int)floorf(ps.y), pz = (int)floorf(ps.z);
    V3 dd = ps - V3((float)px, (float)py, (float)pz);
    float d00 = lerpf(dd.x, grid_D(m, d, px, py, pz), grid_D(m, d, px + 1, py, pz));
    float d10 = lerpf(dd.x, grid_D(m, d, px, py + 1, pz), grid_D(m, d, px + 1, py + 1, pz));
    float d01 = lerpf(dd.x, grid_D(m, d, px, py, pz + 1), grid_D(m, d, px + 1, py, pz + 1));
    float d11 = lerpf(dd.x, grid_D(m, d, px, py + 1, pz + 1), grid_D(m, d, px + 1, py + 1, pz + 1));
    float d0 = lerpf(dd.y, d00, d10);
    float d1 = lerpf(dd.y, d01, d11);
    return lerpf(dd.z, d0, d1);
}

// Medium::Tr(ray, sampler): the ray is (ro, rd, tMax) in world space
GX_DEV Spec medium_tr(const DMediaTables &mt, int mi, V3 ro, V3 rd, float tMaxRay, SampleStream &ss) {
    const DMedium &m = mt.media[mi];
    if (m.type == GNXR_MEDIUM_HOMOGENEOUS) {   // HomogeneousMedium.cpp:11-15
        Spec sigma_t = spec3(m.sigma_s) + spec3(m.sigma_a);
        return spec_exp(Spec(-sigma_t.r, -sigma_t.g, -sigma_t.b) * fminf(tMaxRay * length(rd), GX_MAX_FLOAT));
    }
    // GridDensityMedium.cpp:57-87 (ratio tracking)
    V3 o, d;
    float rtMax;
    xform_ray(m.w2m, ro, normalize(rd), tMaxRay * length(rd), &o, &d, &rtMax);
    float tMin, tMax;
    if (!unit_box_intersect(o, d, rtMax, &tMin, &tMax)) return Spec(1.f);
    const float *__restrict__ dens = mt.density + m.density_offset;
    float Tr = 1, t = tMin;
    while (true) {
        t -= gx_log(1 - ss.get1d()) * m.inv_max_density / m.sigma_t;
        if (t >= tMax) break;
        float density = grid_density(m, dens, o + d * t);
        Tr *= 1 - fmaxf(0.f, density * m.inv_max_density);
        const float rrThreshold = .1f;
        if (Tr < rrThreshold) {
            float q = fmaxf(.05f, 1 - Tr);
            if (ss.get1d() < q) return Spec(0.f);
            Tr /= 1 - q;
        }
    }
    return Spec(Tr);
}

// Medium::Sample(ray, sampler, arena, &mi): returns the throughput weight; *valid / *tOut describe the sampled
// medium interaction (mi.p = ray(t) on the WORLD ray, mi.wo = -ray.d)
GX_DEV Spec medium_sample(const DMediaTables &mt, int mi, V3 ro, V3 rd, float tMaxRay, SampleStream &ss, bool *valid, float *tOut) {
    const DMedium &m = mt.media[mi];
    *valid = false;
    if (m.type == GNXR_MEDIUM_HOMOGENEOUS) {   // HomogeneousMedium.cpp:17-43
        Spec sigma_s = spec3(m.sigma_s);
        Spec sigma_t = sigma_s + spec3(m.sigma_a);
        int channel = min((int)(ss.get1d() * 3), 3 - 1);
        float st = channel == 0 ? sigma_t.r : (channel == 1 ? sigma_t.g : sigma_t.b);
        float dist = -gx_log(1 - ss.get1d()) / st;
        float len = length(rd);
        float t = fminf(dist / len, tMaxRay);
        bool sampledMedium = t < tMaxRay;
        if (sampledMedium) { *valid = true; *tOut = t; }
        Spec Tr = spec_exp(Spec(-sigma_t.r, -sigma_t.g, -sigma_t.b) * fminf(t, GX_MAX_FLOAT) * len);
        Spec density = sampledMedium ? (sigma_t * Tr) : Tr;
        float pdf = 0;
        pdf += density.r; pdf += density.g; pdf += density.b;
        pdf *= 1 / (float)3;
        if (pdf == 0) pdf = 1;
        return sampledMedium ? (Tr * sigma_s / pdf) : (Tr / pdf);
    }
    // GridDensityMedium.cpp:31-55 (delta tracking)
    V3 o, d;
    float rtMax;
    xform_ray(m.w2m, ro, normalize(rd), tMaxRay * length(rd), &o, &d, &rtMax);
    float tMin, tMax;
    if (!unit_box_intersect(o, d, rtMax, &tMin, &tMax)) return Spec(1.f);
    const float *__restrict__ dens = mt.density + m.density_offset;
    float t = tMin;
    while (true) {
        t -= gx_log(1 - ss.get1d()) * m.inv_max_density / m.sigma_t;
        if (t >= tMax) break;
        if (grid_density(m, dens, o + d * t) * m.inv_max_density > ss.get1d()) {
            *valid = true; *tOut = t;
            return spec3(m.sigma_s) / m.sigma_t;
        }
    }
    return Spec(1.f);
}

}  // namespace gnxr

// device_geom.h -- ray/triangle and BVH traversal on the device.
//   Triangle::Intersect / IntersectP      shape/Triangle.cpp:71-303, 305-453 (watertight test, fp64 edge fallback)
//   Bounds3::IntersectP(ray,invDir,neg)   core/Geometry.h:1380-1406
//   BVHAccel::Intersect / IntersectP      accelerator/BVHAccel.cpp:653-729
// One ray per lane.  The traversal stack lives in LDS as stack[depth][lane] (bank == lane, so no
// conflicts whatever depth each lane is at); nodes are 32 B = two dwordx4 loads, triangles 48 B = three.
#pragma once
#include "device_math.h"
#include "gnxr_device_types.h"

namespace gnxr {

struct TriHit {
    float t, b0, b1, b2;
};

// Ray-only part of the watertight test (Triangle.cpp:91-105): the permutation and the shear depend on the ray
// direction alone, so k_trace computes them once per ray instead of once per triangle (three IEEE divisions and the
// component permutation leave the per-triangle loop; the values are the ones the reference recomputes each time).
struct RayShear {
    int kx, ky, kz;
    float Sx, Sy, Sz;
};
GX_DEV RayShear ray_shear(V3 rd) {
    RayShear r;
    V3 ad = vabs(rd);
    r.kz = (ad.x > ad.y) ? ((ad.x > ad.z) ? 0 : 2) : ((ad.y > ad.z) ? 1 : 2);  // MaxDimension
    r.kx = r.kz + 1; if (r.kx == 3) r.kx = 0;
    r.ky = r.kx + 1; if (r.ky == 3) r.ky = 0;
    V3 d(rd[r.kx], rd[r.ky], rd[r.kz]);
    r.Sx = -d.x / d.z; r.Sy = -d.y / d.z; r.Sz = 1.f / d.z;
    return r;
}
GX_DEV V3 permute3(V3 v, int kz) {   // (v[kx], v[ky], v[kz]) with kx = kz+1, ky = kz+2 (mod 3)
    return kz == 0 ? V3(v.y, v.z, v.x) : (kz == 1 ? V3(v.z, v.x, v.y) : V3(v.x, v.y, v.z));
}

// The ray-space part of Triangle::Intersect (Triangle.cpp:82-168): returns true and fills `h` when the
// ray hits within (0, tMax].  Identical arithmetic for Intersect and IntersectP.
GX_DEV bool tri_test_sheared(V3 p0, V3 p1, V3 p2, V3 ro, const RayShear &rs, float tMax, TriHit *h) {
    V3 p0t = permute3(p0 - ro, rs.kz), p1t = permute3(p1 - ro, rs.kz), p2t = permute3(p2 - ro, rs.kz);
    const float Sx = rs.Sx, Sy = rs.Sy, Sz = rs.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // Triangle.cpp:117-128
        double p2txp1ty = (double)p2t.x * (double)p1t.y;
        double p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y;
        double p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y;
        double p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > tMax * det)) return false;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;
    float maxZt = max_component(vabs(V3(p0t.z, p1t.z, p2t.z)));
    float deltaZ = GX_GAMMA(3) * maxZt;
    float maxXt = max_component(vabs(V3(p0t.x, p1t.x, p2t.x)));
    float maxYt = max_component(vabs(V3(p0t.y, p1t.y, p2t.y)));
    float deltaX = GX_GAMMA(5) * (maxXt + maxZt);
    float deltaY = GX_GAMMA(5) * (maxYt + maxZt);
    float deltaE = 2 * (GX_GAMMA(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = max_component(vabs(V3(e0, e1, e2)));
    float deltaT = 3 * (GX_GAMMA(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * fabsf(invDet);
    if (t <= deltaT) return false;
    h->t = t; h->b0 = b0; h->b1 = b1; h->b2 = b2;
    return true;
}
GX_DEV bool tri_test(V3 p0, V3 p1, V3 p2, V3 ro, V3 rd, float tMax, TriHit *h) {
    RayShear rs = ray_shear(rd);
    return tri_test_sheared(p0, p1, p2, ro, rs, tMax, h);
}

// (t, b0, b1, b2) of a hit the traversal has already accepted: the value-producing operations of tri_test_sheared in the same order
// (Triangle.cpp:91-135, 149-155), without the rejection tests and the error bound (Triangle.cpp:129-148, 156-168), which can only say
// "hit" again for the ray, tMax and triangle the traversal accepted.
GX_DEV void tri_hit_recompute(V3 p0, V3 p1, V3 p2, V3 ro, V3 rd, TriHit *h) {
    const RayShear rs = ray_shear(rd);
    V3 p0t = permute3(p0 - ro, rs.kz), p1t = permute3(p1 - ro, rs.kz), p2t = permute3(p2 - ro, rs.kz);
    const float Sx = rs.Sx, Sy = rs.Sy, Sz = rs.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // Triangle.cpp:117-128
        double p2txp1ty = (double)p2t.x * (double)p1t.y;
        double p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y;
        double p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y;
        double p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    const float det = e0 + e1 + e2;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    const float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    const float invDet = 1 / det;
    h->b0 = e0 * invDet; h->b1 = e1 * invDet; h->b2 = e2 * invDet;
    h->t = tScaled * invDet;
}

GX_DEV void load_tri(const DTri *tris, int leaf, V3 *p0, V3 *p1, V3 *p2) {
    const float4 *q = reinterpret_cast<const float4 *>(tris + leaf);
    float4 a = q[0], b = q[1], c = q[2];
    *p0 = V3(a.x, a.y, a.z); *p1 = V3(b.x, b.y, b.z); *p2 = V3(c.x, c.y, c.z);
}

// Bounds3::IntersectP(ray, invDir, dirIsNeg), Geometry.h:1380-1406
GX_DEV bool slab_test(float4 n0, float4 n1, V3 ro, V3 invDir, const int neg[3], float tMaxRay) {
    // n0 = (lo.x lo.y lo.z hi.x), n1 = (hi.y hi.z offset meta)
    float lox = n0.x, loy = n0.y, loz = n0.z, hix = n0.w, hiy = n1.x, hiz = n1.y;
    float tMin = ((neg[0] ? hix : lox) - ro.x) * invDir.x;
    float tMax = ((neg[0] ? lox : hix) - ro.x) * invDir.x;
    float tyMin = ((neg[1] ? hiy : loy) - ro.y) * invDir.y;
    float tyMax = ((neg[1] ? loy : hiy) - ro.y) * invDir.y;
    const float k = 1 + 2 * GX_GAMMA(3);
    tMax *= k;
    tyMax *= k;
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = ((neg[2] ? hiz : loz) - ro.z) * invDir.z;
    float tzMax = ((neg[2] ? loz : hiz) - ro.z) * invDir.z;
    tzMax *= k;
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    return (tMin < tMaxRay) && (tMax > 0);
}

struct TraceCounters {
    uint32_t nodes, tris;
};

// BVHAccel::Intersect (ANY == false) / IntersectP (ANY == true).  `stack` points at this lane's column of
// the block's LDS stack: entry d is stack[d * STRIDE].  Returns the leaf-order triangle index of the
// closest hit (or of any hit), -1 when nothing is hit.
template <bool ANY, int STRIDE, bool COUNT>
GX_DEV int bvh_traverse(const float4 *__restrict__ nodes, const DTri *__restrict__ tris, V3 ro, V3 rd, float tMax, int *stack, TriHit *best,
                        TraceCounters *cnt) {
    V3 invDir(1.f / rd.x, 1.f / rd.y, 1.f / rd.z);
    int neg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
    int toVisit = 0, current = 0;
    int hitLeaf = -1;
    while (true) {
        float4 n0 = nodes[2 * current], n1 = nodes[2 * current + 1];
        if (COUNT) cnt->nodes++;
        if (slab_test(n0, n1, ro, invDir, neg, tMax)) {
            int offset = __float_as_int(n1.z);
            uint32_t meta = __float_as_uint(n1.w);
            int nPrims = (int)(meta & 0xffffu);
            if (nPrims > 0) {
                for (int i = 0; i < nPrims; ++i) {
                    V3 p0, p1, p2;
                    load_tri(tris, offset + i, &p0, &p1, &p2);
                    if (COUNT) cnt->tris++;
                    TriHit h;
                    if (tri_test(p0, p1, p2, ro, rd, tMax, &h)) {
                        if (ANY) return offset + i;
                        tMax = h.t;  // GeometricPrimitive::Intersect shrinks ray.tMax, Primitive.cpp:36
                        *best = h;
                        hitLeaf = offset + i;
                    }
                }
                if (toVisit == 0) break;
                current = stack[(--toVisit) * STRIDE];
            } else {
                int axis = (int)(meta >> 16);
                if (neg[axis]) {
                    stack[(toVisit++) * STRIDE] = current + 1;
                    current = offset;
                } else {
                    stack[(toVisit++) * STRIDE] = offset;
                    current = current + 1;
                }
            }
        } else {
            if (toVisit == 0) break;
            current = stack[(--toVisit) * STRIDE];
        }
    }
    return hitLeaf;
}

// ---- the world-space part of Triangle::Intersect (Triangle.cpp:170-226) + Material::Bump + BSDF frame ----
struct SurfacePoint {
    V3 p, pError, n;       // Interaction::p / pError / n (geometric normal)
    V3 ns, ss, ts;         // BSDF frame: shading.n, Normalize(shading.dpdu), Cross(ns, ss)   (Reflection.h:106-111)
    bool valid;
};

GX_DEV SurfacePoint surface_point(V3 p0, V3 p1, V3 p2, const TriHit &h, bool has_bump) {
    SurfacePoint s;
    s.valid = true;
    // default UVs (0,0),(1,0),(1,1) (Triangle.h:60-74): duv02 = (-1,-1), duv12 = (0,-1)
    const float duv02_0 = 0.f - 1.f, duv02_1 = 0.f - 1.f, duv12_0 = 1.f - 1.f, duv12_1 = 0.f - 1.f;
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = duv02_0 * duv12_1 - duv02_1 * duv12_0;
    V3 dpdu, dpdv;
    bool degenerateUV = fabsf(determinant) < 1e-8f;
    if (!degenerateUV) {
        float invdet = 1 / determinant;
        dpdu = (duv12_1 * dp02 - duv02_1 * dp12) * invdet;
        dpdv = (-duv12_0 * dp02 + duv02_0 * dp12) * invdet;
    }
    if (degenerateUV || length_sq(cross(dpdu, dpdv)) == 0) {
        V3 ng = cross(p2 - p0, p1 - p0);
        if (length_sq(ng) == 0) { s.valid = false; return s; }
        coordinate_system(normalize(ng), &dpdu, &dpdv);
    }
    float xAbsSum = (fabsf(h.b0 * p0.x) + fabsf(h.b1 * p1.x) + fabsf(h.b2 * p2.x));
    float yAbsSum = (fabsf(h.b0 * p0.y) + fabsf(h.b1 * p1.y) + fabsf(h.b2 * p2.y));
    float zAbsSum = (fabsf(h.b0 * p0.z) + fabsf(h.b1 * p1.z) + fabsf(h.b2 * p2.z));
    s.pError = GX_GAMMA(7) * V3(xAbsSum, yAbsSum, zAbsSum);
    s.p = h.b0 * p0 + h.b1 * p1 + h.b2 * p2;
    s.n = normalize(cross(dp02, dp12));  // Triangle.cpp:223
    V3 sn = s.n, sdpdu = dpdu, sdpdv = dpdv;
    if (has_bump) {
        // Material::Bump with ConstantTexture(0), core/Material.cpp:16-52: displace = uDisplace = vDisplace = 0,
        // du = dv = .0005 -> dpdu' = dpdu + 0/du * n + 0 * dndu ; then SetShadingGeometry(.., false)
        const float du = .0005f;
        V3 zero(0, 0, 0);
        sdpdu = dpdu + (0.f - 0.f) / du * sn + 0.f * zero;
        sdpdv = dpdv + (0.f - 0.f) / du * sn + 0.f * zero;
        sn = normalize(cross(sdpdu, sdpdv));
        sn = faceforward(sn, s.n);
    }
    s.ns = sn;
    s.ss = normalize(sdpdu);
    s.ts = cross(s.ns, s.ss);
    return s;
}

// ---- pbrt-v3 quadratic sphere (include/gnxr.h: the reference's Sphere is an unfinished stub; parity unpinned) ----
// Same operations as oracle/o_scene.h SphereIntersect: quadratic in double, nearest root in (0, tMax].
GX_DEV bool sphere_test(const DSphere &sp, V3 ro, V3 rd, float tMax, float *tHit) {
    V3 o = ro - V3(sp.c[0], sp.c[1], sp.c[2]);
    double ox = o.x, oy = o.y, oz = o.z, dx = rd.x, dy = rd.y, dz = rd.z;
    double a = dx * dx + dy * dy + dz * dz;
    double b = 2 * (dx * ox + dy * oy + dz * oz);
    double cc = ox * ox + oy * oy + oz * oz - (double)sp.r * (double)sp.r;
    double discrim = b * b - 4 * a * cc;
    if (discrim < 0) return false;
    double rootDiscrim = __builtin_sqrt(discrim);
    double q = (b < 0) ? -.5 * (b - rootDiscrim) : -.5 * (b + rootDiscrim);
    double t0 = q / a, t1 = cc / q;
    if (t0 > t1) { double tmp = t0; t0 = t1; t1 = tmp; }
    if (!(t0 <= (double)tMax) || !(t1 > 0)) return false;
    double tShapeHit = t0;
    if (tShapeHit <= 0) {
        tShapeHit = t1;
        if (tShapeHit > (double)tMax) return false;
    }
    *tHit = (float)tShapeHit;
    return true;
}
// hit point, error bound, normal and BSDF frame of a sphere hit at tHit (+ Material::Bump with the constant-0 map)
GX_DEV SurfacePoint sphere_surface_point(const DSphere &sp, V3 ro, V3 rd, float tHit, bool has_bump) {
    SurfacePoint s;
    s.valid = true;
    const float radius = sp.r;
    V3 c(sp.c[0], sp.c[1], sp.c[2]);
    V3 o = ro - c;
    V3 pHit = o + rd * tHit;
    pHit = pHit * (radius / length(pHit));
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
    const float phiMax = 2 * GX_PI, thetaMin = GX_PI, thetaMax = 0;
    float theta = gx_acos(clampf(pHit.z / radius, -1, 1));
    float zRadius = gx_sqrt(pHit.x * pHit.x + pHit.y * pHit.y);
    float invZRadius = 1 / zRadius;
    float cosPhi = pHit.x * invZRadius, sinPhi = pHit.y * invZRadius;
    V3 dpdu(-phiMax * pHit.y, phiMax * pHit.x, 0);
    V3 dpdv = (thetaMax - thetaMin) * V3(pHit.z * cosPhi, pHit.z * sinPhi, -radius * gx_sin(theta));
    V3 pError = GX_GAMMA(5) * vabs(pHit);
    s.p = pHit + c;
    s.pError = V3((GX_GAMMA(3) + 1) * pError.x + GX_GAMMA(3) * (fabsf(pHit.x) + fabsf(c.x)),
                  (GX_GAMMA(3) + 1) * pError.y + GX_GAMMA(3) * (fabsf(pHit.y) + fabsf(c.y)),
                  (GX_GAMMA(3) + 1) * pError.z + GX_GAMMA(3) * (fabsf(pHit.z) + fabsf(c.z)));
    s.n = normalize(normalize(cross(dpdu, dpdv)));
    V3 sn = s.n, sdpdu = dpdu, sdpdv = dpdv;
    if (has_bump) {   // Material::Bump, core/Material.cpp:16-52, displacement 0 (see surface_point)
        const float du = .0005f;
        V3 zero(0, 0, 0);
        sdpdu = dpdu + (0.f - 0.f) / du * sn + 0.f * zero;
        sdpdv = dpdv + (0.f - 0.f) / du * sn + 0.f * zero;
        sn = normalize(cross(sdpdu, sdpdv));
        sn = faceforward(sn, s.n);
    }
    s.ns = sn;
    s.ss = normalize(sdpdu);
    s.ts = cross(s.ns, s.ss);
    return s;
}

// Interaction::SpawnRay / SpawnRayTo, Interaction.h:33-53
GX_DEV void spawn_ray(V3 p, V3 pError, V3 n, V3 d, V3 *o) { *o = offset_ray_origin(p, pError, n, d); }
GX_DEV void spawn_ray_to(V3 p, V3 pError, V3 n, V3 p2, V3 p2Error, V3 n2, V3 *o, V3 *d) {
    V3 origin = offset_ray_origin(p, pError, n, p2 - p);
    V3 target = offset_ray_origin(p2, p2Error, n2, origin - p2);
    *o = origin;
    *d = target - origin;
}

}  // namespace gnxr

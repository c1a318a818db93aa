// device_texture.h -- image textures on the device.
//   ImageTexture::Evaluate / UVMapping2D::Map        textures/ImageTexture.h:56-63, core/Texture.cpp:168-175
//   MIPMap::Lookup (trilinear, EWA) / triangle / EWA / Texel   core/MIPMap.h:203-334
//   Triangle::GetUVs defaults, uvHit, dpdu / dpdv    shape/Triangle.h:60-74, shape/Triangle.cpp:170-205
//   SurfaceInteraction::ComputeDifferentials         core/Interaction.cpp:65-112, SolveLinearSystem2x2 core/Transform.cpp:12-20
//   camera offset rays, ScaleDifferentials           camera/Perspective.cpp:86-110, core/Geometry.h:874-880, core/Integrator.cpp:283
//   SpecularReflect / SpecularTransmit differentials core/Integrator.cpp:335-354, 376-436
// The pyramid (y flip, convertIn, Lanczos resample, box filter) and the EWA weight table are built on the host
// (scene_compile.cpp build_textures).  PathIntegrator slices the camera RayDifferential to a Ray (PathIntegrator.cpp:67), so
// under Path every lookup is unfiltered (zero differentials); VolPath filters at the camera ray's first surface vertex;
// Whitted / DirectLighting carry the differentials along their specular chains.
#pragma once
#include "device_geom.h"

namespace gnxr {

struct DTexTables {
    const DTexture *textures;
    const float4 *texels;
    const float *ewa_lut;                // MIPMap::weightLut[128]
    const float *tri_uv;                 // null, or per leaf-order triangle (u,v) x 3 corners + 2 pad floats (TriangleMesh::uv)
    const float *tri_n;                  // null, or per leaf-order triangle 3 shading normals (12 floats; zeros == none) (TriangleMesh::n)
    const float *tri_s;                  // the same for the shading tangents (TriangleMesh::s)
};

// The tables travel in the slot BEFORE the first material (the material array is uploaded with one leading record) instead of in
// DScene: only the TEX kernels read them, and the kernel-argument block of every other kernel stays what it was.
static_assert(sizeof(DTexTables) <= sizeof(DMaterial), "DTexTables must fit the leading material slot");
GX_DEV const DTexTables &tex_tables(const DMaterial *materials) { return *reinterpret_cast<const DTexTables *>(materials - 1); }

struct RayDiff {          // RayDifferential's offset rays
    bool has;
    V3 rxo, ryo, rxd, ryd;
};
struct UVDiff {           // what ComputeDifferentials leaves in the SurfaceInteraction
    float dudx, dvdx, dudy, dvdy;
    V3 dpdx, dpdy;
};

GX_DEV int tex_w(const DTexture &t, int level) { return max(1, t.w0 >> level); }
GX_DEV int tex_h(const DTexture &t, int level) { return max(1, t.h0 >> level); }
GX_DEV int tex_mod(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }
// MIPMap::Texel, MIPMap.h:203-223
GX_DEV Spec tex_texel(const DTexTables &tt, const DTexture &t, int level, int s, int u) {
    const int w = tex_w(t, level), h = tex_h(t, level);
    if (t.wrap == GNXR_WRAP_REPEAT) { s = tex_mod(s, w); u = tex_mod(u, h); }
    else if (t.wrap == GNXR_WRAP_CLAMP) { s = min(max(s, 0), w - 1); u = min(max(u, 0), h - 1); }
    else if (s < 0 || s >= w || u < 0 || u >= h) return Spec(0.f);
    float4 v = tt.texels[t.level_offset[level] + u * w + s];
    return Spec(v.x, v.y, v.z);
}
// MIPMap::triangle, MIPMap.h:244-256
GX_DEV Spec tex_triangle(const DTexTables &tt, const DTexture &t, int level, float s_, float t_) {
    level = min(max(level, 0), t.n_levels - 1);
    float s = s_ * (float)tex_w(t, level) - 0.5f;
    float u = t_ * (float)tex_h(t, level) - 0.5f;
    int s0 = (int)floorf(s), t0 = (int)floorf(u);
    float ds = s - (float)s0, dt = u - (float)t0;
    return (1 - ds) * (1 - dt) * tex_texel(tt, t, level, s0, t0) + (1 - ds) * dt * tex_texel(tt, t, level, s0, t0 + 1) +
           ds * (1 - dt) * tex_texel(tt, t, level, s0 + 1, t0) + ds * dt * tex_texel(tt, t, level, s0 + 1, t0 + 1);
}
GX_DEV float tex_log2(float x) { const float invLog2 = 1.442695040888963387004650940071f; return gx_log(x) * invLog2; }
GX_DEV Spec tex_lerp(float t, Spec a, Spec b) { return (1 - t) * a + t * b; }
// MIPMap::Lookup(st, width), MIPMap.h:225-242
GX_DEV Spec tex_lookup_width(const DTexTables &tt, const DTexture &t, float s, float u, float width) {
    float level = (float)(t.n_levels - 1) + tex_log2(fmaxf(width, 1e-8f));
    if (level < 0) return tex_triangle(tt, t, 0, s, u);
    else if (level >= (float)(t.n_levels - 1)) return tex_texel(tt, t, t.n_levels - 1, 0, 0);
    int iLevel = (int)floorf(level);
    float delta = level - (float)iLevel;
    return tex_lerp(delta, tex_triangle(tt, t, iLevel, s, u), tex_triangle(tt, t, iLevel + 1, s, u));
}
// MIPMap::EWA, MIPMap.h:288-334
GX_DEV Spec tex_ewa(const DTexTables &tt, const DTexture &t, int level, float s, float u, float d0x, float d0y, float d1x, float d1y) {
    if (level >= t.n_levels) return tex_texel(tt, t, t.n_levels - 1, 0, 0);
    const float w = (float)tex_w(t, level), h = (float)tex_h(t, level);
    s = s * w - 0.5f;
    u = u * h - 0.5f;
    d0x *= w; d0y *= h;
    d1x *= w; d1y *= h;
    float A = d0y * d0y + d1y * d1y + 1;
    float B = -2 * (d0x * d0y + d1x * d1y);
    float C = d0x * d0x + d1x * d1x + 1;
    float invF = 1 / (A * C - B * B * 0.25f);
    A *= invF; B *= invF; C *= invF;
    float det = -B * B + 4 * A * C;
    float invDet = 1 / det;
    float uSqrt = gx_sqrt(det * C), vSqrt = gx_sqrt(A * det);
    int s0 = (int)ceilf(s - 2 * invDet * uSqrt);
    int s1 = (int)floorf(s + 2 * invDet * uSqrt);
    int t0 = (int)ceilf(u - 2 * invDet * vSqrt);
    int t1 = (int)floorf(u + 2 * invDet * vSqrt);
    Spec sum(0.f);
    float sumWts = 0;
    for (int it = t0; it <= t1; ++it) {
        float tq = (float)it - u;
        for (int is = s0; is <= s1; ++is) {
            float sq = (float)is - s;
            float r2 = A * sq * sq + B * sq * tq + C * tq * tq;
            if (r2 < 1) {
                int index = min((int)(r2 * 128), 128 - 1);
                float weight = tt.ewa_lut[index];
                sum = sum + tex_texel(tt, t, level, is, it) * weight;
                sumWts += weight;
            }
        }
    }
    return sum / sumWts;
}
// ImageTexture::Evaluate = UVMapping2D::Map + MIPMap::Lookup(st, dst0, dst1), MIPMap.h:258-286 (convertOut is the identity)
GX_DEV Spec tex_evaluate(const DTexTables &tt, int texture, float u, float v, const UVDiff &d) {
    const DTexture &t = tt.textures[texture];
    float d0x = t.su * d.dudx, d0y = t.sv * d.dvdx, d1x = t.su * d.dudy, d1y = t.sv * d.dvdy;
    const float s = t.su * u + t.du, w = t.sv * v + t.dv;
    if (t.trilinear) {
        float width = fmaxf(fmaxf(fabsf(d0x), fabsf(d0y)), fmaxf(fabsf(d1x), fabsf(d1y)));
        return tex_lookup_width(tt, t, s, w, width);
    }
    if (d0x * d0x + d0y * d0y < d1x * d1x + d1y * d1y) { float a = d0x, b = d0y; d0x = d1x; d0y = d1y; d1x = a; d1y = b; }
    float majorLength = gx_sqrt(d0x * d0x + d0y * d0y);
    float minorLength = gx_sqrt(d1x * d1x + d1y * d1y);
    if (minorLength * t.max_aniso < majorLength && minorLength > 0) {
        float scale = majorLength / (minorLength * t.max_aniso);
        d1x *= scale; d1y *= scale;
        minorLength *= scale;
    }
    if (minorLength == 0) return tex_triangle(tt, t, 0, s, w);
    float lod = fmaxf(0.f, (float)t.n_levels - 1.f + tex_log2(minorLength));
    int ilod = (int)floorf(lod);
    return tex_lerp(lod - (float)ilod, tex_ewa(tt, t, ilod, s, w, d0x, d0y, d1x, d1y), tex_ewa(tt, t, ilod + 1, s, w, d0x, d0y, d1x, d1y));
}

// Spectrum::Clamp(0, Infinity), core/Spectrum.h + Clamp(), GNXRayTracer.h
GX_DEV float tex_clamp0(float v) { return v < 0.f ? 0.f : (v > GX_INF ? GX_INF : v); }
// <Matte|Plastic>Material::ComputeScatteringFunctions with image textures (materials/MatteMaterial.cpp:14-32,
// materials/PlasticMaterial.cpp:15-41): `src` is the host-built template whose lobe 0 is the Kd lobe and lobe 1 (Plastic) the Ks
// lobe, both always present; look Kd / Ks up at this hit and keep the lobes whose reflectance is not black, in order.
GX_DEV void textured_material(const DTexTables &tt, const DMaterial &src, float u, float v, const UVDiff &d, DMaterial *out) {
    out->has_bump = src.has_bump; out->eta = src.eta; out->shade_class = src.shade_class; out->kd_tex = src.kd_tex; out->ks_tex = src.ks_tex;
    int n = 0;
    for (int i = 0; i < src.n_lobes && i < 2; ++i) {
        const int tex = i == 0 ? src.kd_tex : src.ks_tex;
        Spec R(src.lobes[i].R[0], src.lobes[i].R[1], src.lobes[i].R[2]);
        if (tex > 0) { Spec e = tex_evaluate(tt, tex - 1, u, v, d); R = Spec(tex_clamp0(e.r), tex_clamp0(e.g), tex_clamp0(e.b)); }
        if (!R.is_black()) {
            out->lobes[n] = src.lobes[i];
            out->lobes[n].R[0] = R.r; out->lobes[n].R[1] = R.g; out->lobes[n].R[2] = R.b;
            ++n;
        }
    }
    out->n_lobes = n;
    out->n_nonspecular = n;   // Lambert / OrenNayar / microfacet reflection: none is specular
}

// Triangle::GetUVs, shape/Triangle.h:60-74
struct TriUV { float u0, v0, u1, v1, u2, v2; };
GX_DEV TriUV tri_uvs(const DTexTables &tt, int leaf) {
    TriUV t = {0.f, 0.f, 1.f, 0.f, 1.f, 1.f};
    if (tt.tri_uv && leaf >= 0) {   // the table holds the defaults for triangles that were never given uvs
        const float4 *q = reinterpret_cast<const float4 *>(tt.tri_uv + (size_t)leaf * 8);
        float4 a = q[0], b = q[1];
        t.u0 = a.x; t.v0 = a.y; t.u1 = a.z; t.v1 = a.w; t.u2 = b.x; t.v2 = b.y;
    }
    return t;
}
// dpdu / dpdv of Triangle::Intersect for arbitrary uvs (shape/Triangle.cpp:170-196): the arithmetic surface_point (device_geom.h)
// has folded for the default uvs; false when the triangle is degenerate (ng == 0: the reference reports no hit)
GX_DEV bool tri_dpduv(V3 p0, V3 p1, V3 p2, const TriUV &uv, V3 *dpdu, V3 *dpdv) {
    const float duv02_0 = uv.u0 - uv.u2, duv02_1 = uv.v0 - uv.v2, duv12_0 = uv.u1 - uv.u2, duv12_1 = uv.v1 - uv.v2;
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = duv02_0 * duv12_1 - duv02_1 * duv12_0;
    bool degenerateUV = (double)fabsf(determinant) < 1e-8;   // `std::abs(determinant) < 1e-8`: a double comparison
    if (!degenerateUV) {
        float invdet = 1 / determinant;
        *dpdu = (duv12_1 * dp02 - duv02_1 * dp12) * invdet;
        *dpdv = (-duv12_0 * dp02 + duv02_0 * dp12) * invdet;
    }
    if (degenerateUV || length_sq(cross(*dpdu, *dpdv)) == 0) {
        V3 ng = cross(p2 - p0, p1 - p0);
        if (length_sq(ng) == 0) return false;
        coordinate_system(normalize(ng), dpdu, dpdv);
    }
    return true;
}
// TriangleMesh::n of a triangle; false when it has none
struct TriN { V3 n0, n1, n2; };
GX_DEV bool tri_normals(const float *table, int leaf, TriN *out) {
    if (!table || leaf < 0) return false;
    const float4 *q = reinterpret_cast<const float4 *>(table + (size_t)leaf * 12);
    float4 a = q[0], b = q[1], c = q[2];
    out->n0 = V3(a.x, a.y, a.z); out->n1 = V3(a.w, b.x, b.y); out->n2 = V3(b.z, b.w, c.x);
    return a.x != 0 || a.y != 0 || a.z != 0 || a.w != 0 || b.x != 0 || b.y != 0 || b.z != 0 || b.w != 0 || c.x != 0;
}
// The shading-geometry block of Triangle::Intersect for a triangle with per-vertex normals and / or tangents (shape/Triangle.cpp:
// 228-297): interpolated shading normal, the (ss, ts) frame, dndu / dndv, and SetShadingGeometry(ss, ts, dndu, dndv, true), which
// flips the GEOMETRIC normal onto the shading normal's side.  In: the flat SurfacePoint's n and the unshaded dpdu.  Out: n (flipped),
// shading.n, shading.dpdu (= ss), shading.dpdv (= ts), dndu, dndv.
struct ShadingGeom { V3 n, sn, sdpdu, sdpdv, dndu, dndv; };
GX_DEV ShadingGeom tri_shading_geometry(const TriN *tnp, const TriN *tsp, const TriHit &h, const TriUV &uv, V3 n, V3 dpdu) {
    ShadingGeom g;
    V3 ns = n;
    if (tnp) {
        ns = (h.b0 * tnp->n0 + h.b1 * tnp->n1 + h.b2 * tnp->n2);
        if (length_sq(ns) > 0) ns = normalize(ns);
        else ns = n;
    }
    V3 ss = normalize(dpdu);
    if (tsp) {   // mesh->s, Triangle.cpp:242-250
        ss = (h.b0 * tsp->n0 + h.b1 * tsp->n1 + h.b2 * tsp->n2);
        if (length_sq(ss) > 0) ss = normalize(ss);
        else ss = normalize(dpdu);
    }
    V3 ts = cross(ss, ns);
    if (length_sq(ts) > 0.f) {
        ts = normalize(ts);
        ss = cross(ts, ns);
    } else coordinate_system(ns, &ss, &ts);
    g.dndu = g.dndv = V3(0, 0, 0);
    if (tnp) {
        const TriN &tn = *tnp;
        const float duv02_0 = uv.u0 - uv.u2, duv02_1 = uv.v0 - uv.v2, duv12_0 = uv.u1 - uv.u2, duv12_1 = uv.v1 - uv.v2;
        V3 dn1 = tn.n0 - tn.n2, dn2 = tn.n1 - tn.n2;
        float determinant = duv02_0 * duv12_1 - duv02_1 * duv12_0;
        if ((double)fabsf(determinant) < 1e-8) {
            V3 dn = cross(tn.n2 - tn.n0, tn.n1 - tn.n0);
            if (length_sq(dn) == 0) g.dndu = g.dndv = V3(0, 0, 0);
            else coordinate_system(dn, &g.dndu, &g.dndv);
        } else {
            float invDet = 1 / determinant;
            g.dndu = (duv12_1 * dn1 - duv02_1 * dn2) * invDet;
            g.dndv = (-duv12_0 * dn1 + duv02_0 * dn2) * invDet;
        }
    }
    // SetShadingGeometry(ss, ts, dndu, dndv, true), Interaction.cpp:36-54
    g.sn = normalize(cross(ss, ts));
    g.n = faceforward(n, g.sn);
    g.sdpdu = ss; g.sdpdv = ts;
    return g;
}
// SurfacePoint of a triangle hit with its attributes: per-corner uvs (`uv`, defaults when it has none) and, when `tn` is given,
// per-vertex normals.  Material::Bump then runs on the shading geometry (core/Material.cpp:16-52 with the constant-0 displacement).
// *dndu / *dndv receive shading.dndu / dndv (zero without normals).
GX_DEV SurfacePoint surface_point_attr(V3 p0, V3 p1, V3 p2, const TriHit &h, bool has_bump, const TriUV &uv, const TriN *tn, const TriN *ts, V3 *dndu, V3 *dndv) {
    SurfacePoint s;
    s.valid = true;
    *dndu = *dndv = V3(0, 0, 0);
    V3 dpdu, dpdv;
    if (!tri_dpduv(p0, p1, p2, uv, &dpdu, &dpdv)) { s.valid = false; return s; }
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float xAbsSum = (fabsf(h.b0 * p0.x) + fabsf(h.b1 * p1.x) + fabsf(h.b2 * p2.x));
    float yAbsSum = (fabsf(h.b0 * p0.y) + fabsf(h.b1 * p1.y) + fabsf(h.b2 * p2.y));
    float zAbsSum = (fabsf(h.b0 * p0.z) + fabsf(h.b1 * p1.z) + fabsf(h.b2 * p2.z));
    s.pError = GX_GAMMA(7) * V3(xAbsSum, yAbsSum, zAbsSum);
    s.p = h.b0 * p0 + h.b1 * p1 + h.b2 * p2;
    s.n = normalize(cross(dp02, dp12));
    V3 sn = s.n, sdpdu = dpdu, sdpdv = dpdv;
    if (tn || ts) {   // `if (mesh->n || mesh->s)`, Triangle.cpp:228
        ShadingGeom g = tri_shading_geometry(tn, ts, h, uv, s.n, dpdu);
        s.n = g.n; sn = g.sn; sdpdu = g.sdpdu; sdpdv = g.sdpdv;
        *dndu = g.dndu; *dndv = g.dndv;
    }
    if (has_bump) {
        const float du = .0005f;
        sdpdu = sdpdu + (0.f - 0.f) / du * sn + 0.f * *dndu;
        sdpdv = sdpdv + (0.f - 0.f) / du * sn + 0.f * *dndv;
        sn = normalize(cross(sdpdu, sdpdv));
        sn = faceforward(sn, s.n);
    }
    s.ns = sn;
    s.ss = normalize(sdpdu);
    s.ts = cross(s.ns, s.ss);
    return s;
}

// the general-queue kernels shade every triangle hit through this: attributes from the scene tables (defaults when absent)
GX_DEV SurfacePoint surface_point_tables(const DTexTables &tt, int leaf, V3 p0, V3 p1, V3 p2, const TriHit &h, bool has_bump, V3 *dndu, V3 *dndv) {
    TriN tn, ts;
    const bool hasN = tri_normals(tt.tri_n, leaf, &tn), hasS = tri_normals(tt.tri_s, leaf, &ts);
    return surface_point_attr(p0, p1, p2, h, has_bump, tri_uvs(tt, leaf), hasN ? &tn : nullptr, hasS ? &ts : nullptr, dndu, dndv);
}

// uv of the hit (`uvHit = b0 * uv[0] + b1 * uv[1] + b2 * uv[2]`, Triangle.cpp:205) and the UNSHADED dpdu / dpdv
GX_DEV void tri_uv_frame(V3 p0, V3 p1, V3 p2, const TriHit &h, const TriUV &uv, float *u, float *v, V3 *dpdu, V3 *dpdv) {
    (void)tri_dpduv(p0, p1, p2, uv, dpdu, dpdv);
    *u = h.b0 * uv.u0 + h.b1 * uv.u1 + h.b2 * uv.u2;
    *v = h.b0 * uv.v0 + h.b1 * uv.v1 + h.b2 * uv.v2;
}

GX_DEV bool solve_2x2(float a00, float a01, float a10, float a11, float b0, float b1, float *x0, float *x1) {
    float det = a00 * a11 - a01 * a10;
    if (fabsf(det) < 1e-10f) return false;
    *x0 = (a11 * b0 - a01 * b1) / det;
    *x1 = (a00 * b1 - a10 * b0) / det;
    if (isnan(*x0) || isnan(*x1)) return false;
    return true;
}
GX_DEV float v3c(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
// SurfaceInteraction::ComputeDifferentials; n = the interaction's geometric normal, p its position
GX_DEV UVDiff compute_differentials(const RayDiff &rd, V3 p, V3 n, V3 dpdu, V3 dpdv) {
    UVDiff o;
    o.dudx = o.dvdx = o.dudy = o.dvdy = 0;
    o.dpdx = o.dpdy = V3(0, 0, 0);
    if (!rd.has) return o;
    float d = dot(n, V3(p.x, p.y, p.z));
    float tx = -(dot(n, rd.rxo) - d) / dot(n, rd.rxd);
    if (isinf(tx) || isnan(tx)) return o;
    V3 px = rd.rxo + tx * rd.rxd;
    float ty = -(dot(n, rd.ryo) - d) / dot(n, rd.ryd);
    if (isinf(ty) || isnan(ty)) return o;
    V3 py = rd.ryo + ty * rd.ryd;
    o.dpdx = px - p;
    o.dpdy = py - p;
    int d0, d1;
    if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) { d0 = 1; d1 = 2; }
    else if (fabsf(n.y) > fabsf(n.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    float a00 = v3c(dpdu, d0), a01 = v3c(dpdv, d0), a10 = v3c(dpdu, d1), a11 = v3c(dpdv, d1);
    float bx0 = v3c(px, d0) - v3c(p, d0), bx1 = v3c(px, d1) - v3c(p, d1);
    float by0 = v3c(py, d0) - v3c(p, d0), by1 = v3c(py, d1) - v3c(p, d1);
    if (!solve_2x2(a00, a01, a10, a11, bx0, bx1, &o.dudx, &o.dvdx)) o.dudx = o.dvdx = 0;
    if (!solve_2x2(a00, a01, a10, a11, by0, by1, &o.dudy, &o.dvdy)) o.dudy = o.dvdy = 0;
    return o;
}

// Offset rays of the specular children, SamplerIntegrator::SpecularReflect / SpecularTransmit (core/Integrator.cpp:335-354,
// 376-436).  dndu / dndv = shading.dndu / dndv (zero without per-vertex normals; the products with the uv differentials are kept
// either way).  p / ns: the vertex; wo = isect.wo; wi the sampled direction.
GX_DEV RayDiff reflect_differentials(const RayDiff &ray, const UVDiff &ud, V3 p, V3 ns, V3 dndu, V3 dndv, V3 wo, V3 wi) {
    RayDiff rd;
    rd.has = ray.has;
    if (!ray.has) return rd;
    rd.rxo = p + ud.dpdx;
    rd.ryo = p + ud.dpdy;
    V3 dndx = dndu * ud.dudx + dndv * ud.dvdx;
    V3 dndy = dndu * ud.dudy + dndv * ud.dvdy;
    V3 dwodx = -ray.rxd - wo, dwody = -ray.ryd - wo;
    float dDNdx = dot(dwodx, ns) + dot(wo, dndx);
    float dDNdy = dot(dwody, ns) + dot(wo, dndy);
    rd.rxd = wi - dwodx + 2.f * V3(dot(wo, ns) * dndx + dDNdx * ns);
    rd.ryd = wi - dwody + 2.f * V3(dot(wo, ns) * dndy + dDNdy * ns);
    return rd;
}
GX_DEV RayDiff transmit_differentials(const RayDiff &ray, const UVDiff &ud, V3 p, V3 ns, V3 dndu, V3 dndv, float bsdfEta, V3 wo, V3 wi) {
    RayDiff rd;
    rd.has = ray.has;
    if (!ray.has) return rd;
    rd.rxo = p + ud.dpdx;
    rd.ryo = p + ud.dpdy;
    V3 dndx = dndu * ud.dudx + dndv * ud.dvdx;
    V3 dndy = dndu * ud.dudy + dndv * ud.dvdy;
    float eta = 1 / bsdfEta;
    if (dot(wo, ns) < 0) {
        eta = 1 / eta;
        ns = -ns;
        dndx = -dndx;
        dndy = -dndy;
    }
    V3 dwodx = -ray.rxd - wo, dwody = -ray.ryd - wo;
    float dDNdx = dot(dwodx, ns) + dot(wo, dndx);
    float dDNdy = dot(dwody, ns) + dot(wo, dndy);
    float mu = eta * dot(wo, ns) - absdot(wi, ns);
    float dmudx = (eta - (eta * eta * dot(wo, ns)) / absdot(wi, ns)) * dDNdx;
    float dmudy = (eta - (eta * eta * dot(wo, ns)) / absdot(wi, ns)) * dDNdy;
    rd.rxd = wi - eta * dwodx + V3(mu * dndx + dmudx * ns);
    rd.ryd = wi - eta * dwody + V3(mu * dndy + dmudy * ns);
    return rd;
}

}  // namespace gnxr

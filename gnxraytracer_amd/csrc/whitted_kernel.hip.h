// whitted_kernel.hip.h -- WhittedIntegrator::Li (integrators/WhittedIntegrator.cpp:14-68) with SamplerIntegrator::
// SpecularReflect / SpecularTransmit (core/Integrator.cpp:321-442) as a per-path depth-first state machine (BASELINE
// config 1; the reference runs it on the CPU only).
//
// The recursion consumes the sample stream in DFS order -- all lights of a vertex, the reflected subtree, then the
// transmitted subtree -- and combines radiance on the way back up as `f * Li(child) * |cos| / pdf`, so each path keeps an
// explicit stack of frames (one per recursion level: the vertex's ray + hit to re-establish its BSDF, the radiance
// accumulated so far, which child is pending and that child's weight) and has one closest-hit ray in flight.  A round is
//     k_trace (the path's ray + the shadow rays of the vertex established in the previous round) -> k_whitted_step.
// Every light is sampled at every vertex (WhittedIntegrator.cpp:44-55); the shadow rays use the NEE records of
// PathArrays at slot `light * cap + path` (shadow part only).  Ray differentials are not carried: they only feed texture
// filtering and every texture on this path is constant.
//
// DirectLightingIntegrator::Li (integrators/DirectLightingIntegrator.cpp:30-64) is the same recursion with a different
// direct-lighting term at each vertex, so it shares the state machine (MODE):
//   WM_DIRECT_ONE  UniformSampleOneLight without a distribution (core/Integrator.cpp:57-79): one NEE record per vertex
//   WM_DIRECT_ALL  UniformSampleAllLights (core/Integrator.cpp:25-55): Light::nSamples records per light, drawn from the
//                  sampler's 2D ARRAYS (core/Sampler.cpp:52-72, 116-146).  Preprocess requests maxDepth x lights x 2 arrays;
//                  they live in Halton dimensions [5, 5 + 4 * maxDepth * lights), which the regular stream skips, and element
//                  k of an array for pixel sample s comes from Halton index GetIndexForSample(s * n + k).  The DFS can visit
//                  more than maxDepth vertices; once the arrays are used up Get2DArray returns nullptr and the reference falls
//                  back to one Get2D pair per light (Integrator.cpp:38-43) -- counted per path in ws.w.
// Both use full EstimateDirect records (shadow ray + MIS closest-hit ray, estimate_direct_record in kernels.hip.h).
#pragma once
#include "kernels.hip.h"

namespace gnxr {

enum WhittedMode { WM_WHITTED = 0, WM_DIRECT_ONE = 1, WM_DIRECT_ALL = 2 };

struct WhittedArrays {
    int4 *ws;        // x: frames on the stack (= recursion depth of the ray in flight), y: Halton dimension,
                     // z: frame whose shadow rays are in flight (-1: none), w: bit0 = no ray in flight, waiting for those shadow
                     // rays; bit1 = those records were drawn from sample arrays; bits 2..: vertices that consumed arrays so far
    float4 *fr_o;    // [depth * cap + path] ray that reached the vertex: o.xyz, tMax
    float4 *fr_d;    // d.xyz, w: hit code (int bits)
    float4 *fr_L;    // radiance accumulated at the vertex, w: stage (int bits: 0 reflect next, 1 transmit next, 2 done)
    float4 *fr_w;    // weight of the pending child: f.rgb, w: |cos|
    float *fr_pdf;   // pdf of the pending child
    int cap;
    int n_lights;
    // TEX only: offset rays (RayDifferential) of the ray that reaches depth d, [d * cap + path]; rxo.w: hasDifferentials
    float4 *fr_rxo, *fr_rxd, *fr_ryo, *fr_ryd;
    int n_records;   // NEE records per vertex: lights (Whitted), 1 (DIRECT_ONE), sum of Light::nSamples (DIRECT_ALL)
    int start_dim;   // first regular dimension after the camera sample: 5, or arrayEndDim when arrays were requested
};

static __global__ void __launch_bounds__(kBlock) k_whitted_init(PathArrays pa, WhittedArrays wa, int n_paths) {
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_paths; slot += gridDim.x * blockDim.x) {
        uint2 m = pa.meta[(size_t)slot * kRSm];
        // GlobalSampler::Get1D / Get2D jump over [arrayStartDim, arrayEndDim) (core/Sampler.cpp:165-166, 173-174); the camera
        // sample ends exactly at arrayStartDim = 5
        wa.ws[slot] = make_int4(0, max((int)m.y, wa.start_dim), -1, 0);
    }
}

GX_DEV void store_ray_diff(const WhittedArrays &wa, size_t slot, const RayDiff &d) {
    wa.fr_rxo[slot] = make_float4(d.rxo.x, d.rxo.y, d.rxo.z, d.has ? 1.f : 0.f);
    if (!d.has) return;
    wa.fr_rxd[slot] = make_float4(d.rxd.x, d.rxd.y, d.rxd.z, 0.f);
    wa.fr_ryo[slot] = make_float4(d.ryo.x, d.ryo.y, d.ryo.z, 0.f);
    wa.fr_ryd[slot] = make_float4(d.ryd.x, d.ryd.y, d.ryd.z, 0.f);
}
GX_DEV RayDiff load_ray_diff(const WhittedArrays &wa, size_t slot) {
    RayDiff d;
    float4 a = wa.fr_rxo[slot];
    d.has = a.w != 0.f;
    if (d.has) {
        float4 b = wa.fr_rxd[slot], c = wa.fr_ryo[slot], e = wa.fr_ryd[slot];
        d.rxo = V3(a.x, a.y, a.z); d.rxd = V3(b.x, b.y, b.z); d.ryo = V3(c.x, c.y, c.z); d.ryd = V3(e.x, e.y, e.z);
    }
    return d;
}
// the camera ray's offset rays (scenes with image textures): differentials of the ray that reaches depth 0
static __global__ void __launch_bounds__(kBlock) k_whitted_init_diff(DScene sc, DRender r, PathArrays pa, WhittedArrays wa, int n_paths) {
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_paths; slot += gridDim.x * blockDim.x) {
        int px, py;
        local_pixel(r, slot % r.npix, &px, &py);
        store_ray_diff(wa, (size_t)slot, camera_ray_diff(r.cam, sc.st, px, py, pa.meta[(size_t)slot * kRSm].x, r.spp));
    }
}

// record ids of the shadow rays of the paths in `q` (n paths): light * cap + path, light-major inside a path
static __global__ void __launch_bounds__(kBlock) k_whitted_expand(const int *__restrict__ q, int n, int n_lights, int cap, int *__restrict__ out) {
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < (long long)n * n_lights; i += (long long)gridDim.x * blockDim.x) {
        int p = (int)(i / n_lights), l = (int)(i - (long long)p * n_lights);
        out[i] = l * cap + (q ? q[p] : p);
    }
}

// pflags: bit0 the path is still alive, bit1 it has shadow rays to trace this round, bit2 it has a closest-hit ray to trace.
// ray_counts[0] += shadow rays, ray_counts[1] += MIS closest-hit rays spawned here
// TEX: the scene has image-textured materials -- the frames carry ray differentials (WhittedIntegrator / DirectLightingIntegrator
// take the camera RayDifferential and SpecularReflect / SpecularTransmit derive the children's), textures are filtered with them.
// Minimum waves per SIMD.  Every instantiation wants 256 registers; held to 168 (three waves, 120 - 260 dwords spilled) the Whitted and the
// UniformSampleAll kernels are 5 % / 8 % faster (whole render -2.3 % / -4 %), the UniformSampleOne kernel 4 % slower
// (profiles/r03_ab_whitted_occupancy.log); the textured instantiations likewise (image-textured Cornell box: Whitted +-0, UniformSampleAll +7 %,
// profiles/r03_ab_textured_occupancy.log).
#ifndef GX_WHITTED_W
#define GX_WHITTED_W 3
#endif
template <int MODE, bool TEX> constexpr int whitted_min_waves() { return MODE != WM_DIRECT_ONE ? GX_WHITTED_W : 2; }
template <int MODE, int LT, bool SPH, bool TEX = false>
__global__ void __launch_bounds__(kBlock, (whitted_min_waves<MODE, TEX>())) k_whitted_step(DScene sc, DRender r, PathArrays pa, WhittedArrays wa, const int *__restrict__ queue, int n,
                                                         unsigned long long *ray_counts) {
    unsigned long long nShadow = 0, nMis = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int path = queue ? queue[i] : i;
        const size_t cap = (size_t)wa.cap;
        int4 ws = wa.ws[path];
        int top = ws.x, pend = ws.z;
        bool pendArrays = (ws.w & 2) != 0;
        int nArrayVertices = ws.w >> 2;
        const uint32_t index = pa.meta[(size_t)path * kRSm].x;
        SampleStream ss(sc.st, index, ws.y);
        // ---- (a) the shadow rays of the most recent vertex have been traced: lightL, then L += lightL
        if (pend >= 0) {
            Spec lightL(0.f);
            if (MODE == WM_WHITTED) {
                for (int l = 0; l < wa.n_lights; ++l) {
                    const size_t rec = (size_t)l * cap + path;
                    if ((__float_as_int(pa.sh_d[(size_t)rec * kRS].w) & 1) && pa.sh_o[(size_t)rec * kRS].w == 1.f) {
                        float4 X = pa.sh_X[(size_t)rec * kRS];
                        lightL = lightL + Spec(X.x, X.y, X.z);
                    }
                }
            } else if (MODE == WM_DIRECT_ONE) {   // EstimateDirect(...) / lightPdf, Integrator.cpp:78
                float lightPdf;
                Spec Ld = nee_record_Ld(pa, (size_t)path, &lightPdf);
                lightL = Ld / lightPdf;
            } else {                              // UniformSampleAllLights, Integrator.cpp:32-54
                int base = 0;
                float xw;
                for (int l = 0; l < wa.n_lights; ++l) {
                    const int nSamples = sc.lt.lights[l].n_samples;
                    if (pendArrays) {
                        Spec Ld(0.f);
                        for (int k = 0; k < nSamples; ++k) Ld = Ld + nee_record_Ld(pa, (size_t)(base + k) * cap + path, &xw);
                        lightL = lightL + Ld / (float)nSamples;
                    } else lightL = lightL + nee_record_Ld(pa, (size_t)base * cap + path, &xw);
                    base += nSamples;
                }
            }
            float4 *Lp = &wa.fr_L[(size_t)pend * cap + path];
            float4 L4 = *Lp;
            *Lp = make_float4(L4.x + lightL.r, L4.y + lightL.g, L4.z + lightL.b, L4.w);
            pend = -1;
        }
        bool haveResult = false, traceClosest = false, traceShadow = false, done = false;
        Spec result(0.f);
        if ((ws.w & 1) == 0) {
            // ---- (b) the ray in flight (recursion depth `top`) has been traced
            float4 o4 = pa.ray_o[(size_t)path * kRS], d4 = pa.ray_d[(size_t)path * kRS];
            V3 ro(o4.x, o4.y, o4.z), rd(d4.x, d4.y, d4.z);
            const int leaf = pa.hit[path];
            bool found = leaf != -1;
            V3 p0, p1, p2;
            int triMat = -1, triLight = -1;
            TriHit h;
            SurfacePoint sp;
            sp.valid = false;
            if (SPH && leaf < -1) {
                const DSphere &sph = sc.spheres[-2 - leaf];
                triMat = sph.material;
                found = sphere_test(sph, ro, rd, o4.w, &h.t);
                if (found) sp = sphere_surface_point(sph, ro, rd, h.t, triMat >= 0 ? sc.materials[triMat].has_bump != 0 : false);
            } else if (found) {
                const float4 *q = reinterpret_cast<const float4 *>(sc.tris + leaf);
                float4 a = q[0], b = q[1], c = q[2];
                p0 = V3(a.x, a.y, a.z); p1 = V3(b.x, b.y, b.z); p2 = V3(c.x, c.y, c.z);
                triMat = __float_as_int(b.w); triLight = __float_as_int(c.w);
                found = tri_test(p0, p1, p2, ro, rd, o4.w, &h);
                if (found) {
                    sp = surface_point(p0, p1, p2, h, triMat >= 0 ? sc.materials[triMat].has_bump != 0 : false);
                    if (TEX) { V3 dndu, dndv; sp = surface_point_tables(tex_tables(sc.materials), leaf, p0, p1, p2, h, triMat >= 0 ? sc.materials[triMat].has_bump != 0 : false, &dndu, &dndv); }
                    found = sp.valid;
                }
            }
            if (!found) {           // `for (light : scene.lights) L += light->Le(ray)`
                for (int l = 0; l < sc.lt.n_lights; ++l) result = result + light_Le<LT>(sc.lt, l, ro, rd);
                haveResult = true;
            } else if (triMat < 0) { // no BSDF: Li(isect.SpawnRay(ray.d), ..., depth)
                V3 o2 = offset_ray_origin(sp.p, sp.pError, sp.n, rd);
                pa.ray_o[(size_t)path * kRS] = make_float4(o2.x, o2.y, o2.z, GX_INF);
                if (TEX) wa.fr_rxo[(size_t)top * cap + path] = make_float4(0.f, 0.f, 0.f, 0.f);   // isect.SpawnRay(ray.d): a plain Ray
                traceClosest = true;
            } else {
                // establish the frame of this vertex: L = Le + (lightL, next round)
                const DMaterial *mat = sc.materials + triMat;
                DMaterial tm;
                if (TEX && leaf >= 0 && (mat->kd_tex | mat->ks_tex)) {
                    float tu, tv;
                    V3 dpdu, dpdv;
                    tri_uv_frame(p0, p1, p2, h, tri_uvs(tex_tables(sc.materials), leaf), &tu, &tv, &dpdu, &dpdv);
                    textured_material(tex_tables(sc.materials), *mat, tu, tv, compute_differentials(load_ray_diff(wa, (size_t)top * cap + path), sp.p, sp.n, dpdu, dpdv), &tm);
                    mat = &tm;
                }
                Bsdf<LM_ALL> bsdf;
                bsdf.mat = mat; bsdf.ns = sp.ns; bsdf.ng = sp.n; bsdf.ss = sp.ss; bsdf.ts = sp.ts;
                const V3 woN = normalize(-rd);
                Spec L(0.f);
                if (triLight >= 0) L = L + area_L(sc.lt.lights[triLight], sp.n, woN);
                if (MODE == WM_WHITTED) {
                    for (int l = 0; l < wa.n_lights; ++l) {
                        float u0, u1;
                        ss.get2d(&u0, &u1);
                        const size_t rec = (size_t)l * cap + path;
                        int flag = 0;
                        LightSample ls = light_sample<LT>(sc.lt, l, sp.p, u0, u1);
                        if (!(ls.Li.is_black() || ls.pdf == 0)) {
                            Spec f = bsdf.f(woN, ls.wi, BSDF_ALL);
                            if (!f.is_black()) {
                                V3 so, sd;
                                spawn_ray_to(sp.p, sp.pError, sp.n, ls.p1, ls.p1Error, ls.n1, &so, &sd);
                                Spec X = f * ls.Li * absdot(ls.wi, sp.ns) / ls.pdf;
                                pa.sh_o[(size_t)rec * kRS] = make_float4(so.x, so.y, so.z, 1 - GX_SHADOW_EPS);
                                pa.sh_X[(size_t)rec * kRS] = make_float4(X.r, X.g, X.b, 0.f);
                                pa.sh_d[(size_t)rec * kRS] = make_float4(sd.x, sd.y, sd.z, __int_as_float(1));
                                flag = 1;
                                ++nShadow;
                            }
                        }
                        if (!flag) pa.sh_d[(size_t)rec * kRS] = make_float4(0.f, 0.f, 0.f, __int_as_float(0));
                    }
                } else if (sc.lt.n_lights > 0) {
                    auto record = [&](int slot, int lightNum, float ul0, float ul1, float us0, float us1, float xw) {
                        const size_t rec = (size_t)slot * cap + path;
                        const int nflags = estimate_direct_record<LM_ALL, LT>(sc, bsdf, sp, woN, lightNum, ul0, ul1, us0, us1, pa, rec, xw);
                        if (!nflags) { pa.sh_d[(size_t)rec * kRS] = make_float4(0.f, 0.f, 0.f, __int_as_float(0)); pa.sh_X[(size_t)rec * kRS] = make_float4(0.f, 0.f, 0.f, xw); }
                        nShadow += nflags & 1;
                        nMis += (nflags >> 1) & 1;
                    };
                    if (MODE == WM_DIRECT_ONE) {
                        const int nLights = sc.lt.n_lights;
                        const int lightNum = min((int)(ss.get1d() * (float)nLights), nLights - 1);
                        const float lightPdf = 1.f / (float)nLights;
                        float ul0, ul1, us0, us1;
                        ss.get2d(&ul0, &ul1);
                        ss.get2d(&us0, &us1);
                        record(0, lightNum, ul0, ul1, us0, us1, lightPdf);
                    } else {
                        pendArrays = nArrayVertices < r.max_depth;   // Get2DArray still has arrays to hand out
                        const uint32_t stride = (uint32_t)max(1, sc.st.h.stride);
                        const uint32_t sampleNum = index / stride, pixelOffset = index - sampleNum * stride;
                        int base = 0;
                        for (int l = 0; l < wa.n_lights; ++l) {
                            const int nSamples = sc.lt.lights[l].n_samples;
                            if (pendArrays) {
                                const int arrayDim = 5 + 2 * ((nArrayVertices * wa.n_lights + l) * 2);   // uLightArray; uScatteringArray is the next one
                                for (int k = 0; k < nSamples; ++k) {
                                    const uint32_t idx = pixelOffset + (sampleNum * (uint32_t)nSamples + (uint32_t)k) * stride;
                                    float ul0, ul1, us0, us1;
                                    halton_sample_pair(sc.st, idx, arrayDim, &ul0, &ul1);
                                    halton_sample_pair(sc.st, idx, arrayDim + 2, &us0, &us1);
                                    record(base + k, l, ul0, ul1, us0, us1, 1.f);
                                }
                            } else {
                                float ul0, ul1, us0, us1;
                                ss.get2d(&ul0, &ul1);
                                ss.get2d(&us0, &us1);
                                record(base, l, ul0, ul1, us0, us1, 1.f);
                                for (int k = 1; k < nSamples; ++k) pa.sh_d[(size_t)((size_t)(base + k) * cap + path) * kRS] = make_float4(0.f, 0.f, 0.f, __int_as_float(0));
                            }
                            base += nSamples;
                        }
                        if (pendArrays) ++nArrayVertices;
                    }
                } else {
                    // `if (scene.lights.size() > 0)` (DirectLightingIntegrator.cpp:53): no direct term and no sample draws
                    for (int k = 0; k < wa.n_records; ++k) {
                        pa.sh_d[(size_t)((size_t)k * cap + path) * kRS] = make_float4(0.f, 0.f, 0.f, __int_as_float(0));
                        pa.sh_X[(size_t)((size_t)k * cap + path) * kRS] = make_float4(0.f, 0.f, 0.f, 1.f);
                    }
                }
                const size_t fi = (size_t)top * cap + path;
                wa.fr_o[fi] = o4;
                wa.fr_d[fi] = make_float4(rd.x, rd.y, rd.z, __int_as_float(leaf));
                wa.fr_L[fi] = make_float4(L.r, L.g, L.b, __int_as_float(0));
                pend = top;
                traceShadow = true;
                ++top;
            }
        } else {
            // the frame on top was complete and only waited for its shadow rays
            float4 L4 = wa.fr_L[(size_t)(top - 1) * cap + path];
            result = Spec(L4.x, L4.y, L4.z);
            --top;
            haveResult = true;
        }
        // ---- (c) deliver finished subtrees to their parents and advance the frame on top until a ray is needed
        while (!traceClosest && !done) {
            if (haveResult) {
                if (top == 0) { pa.L[path] = make_float4(result.r, result.g, result.b, 0.f); done = true; break; }
                const size_t fp = (size_t)(top - 1) * cap + path;
                float4 w4 = wa.fr_w[fp], L4 = wa.fr_L[fp];
                Spec add = Spec(w4.x, w4.y, w4.z) * result * w4.w / wa.fr_pdf[fp];   // f * Li(...) * AbsDot(wi, ns) / pdf
                wa.fr_L[fp] = make_float4(L4.x + add.r, L4.y + add.g, L4.z + add.b, L4.w);
                haveResult = false;
            }
            const int f = top - 1;
            if (f < 0) break;   // only reachable right after a null-material pass-through handled above
            const size_t fi = (size_t)f * cap + path;
            float4 L4 = wa.fr_L[fi];
            int stage = __float_as_int(L4.w);
            bool emitted = false;
            if (f + 1 < r.max_depth && stage < 2) {
                // re-establish the vertex's BSDF from its ray and hit (same arithmetic as when it was first reached)
                float4 fo = wa.fr_o[fi], fd = wa.fr_d[fi];
                V3 ro(fo.x, fo.y, fo.z), rd(fd.x, fd.y, fd.z);
                const int leaf = __float_as_int(fd.w);
                SurfacePoint sp;
                int triMat;
                TriHit h;
                V3 dpdu(0, 0, 0), dpdv(0, 0, 0);   // TEX: unshaded dpdu / dpdv for ComputeDifferentials (a sphere's only feed the uv
                                                   // differentials, which multiply dndu = dndv = 0 below)
                V3 dndu(0, 0, 0), dndv(0, 0, 0);   // shading.dndu / dndv: non-zero for triangles with per-vertex normals
                if (SPH && leaf < -1) {
                    const DSphere &sph = sc.spheres[-2 - leaf];
                    triMat = sph.material;
                    (void)sphere_test(sph, ro, rd, fo.w, &h.t);
                    sp = sphere_surface_point(sph, ro, rd, h.t, sc.materials[triMat].has_bump != 0);
                } else {
                    const float4 *q = reinterpret_cast<const float4 *>(sc.tris + leaf);
                    float4 a = q[0], b = q[1], c = q[2];
                    V3 p0(a.x, a.y, a.z), p1(b.x, b.y, b.z), p2(c.x, c.y, c.z);
                    triMat = __float_as_int(b.w);
                    (void)tri_test(p0, p1, p2, ro, rd, fo.w, &h);
                    sp = surface_point(p0, p1, p2, h, sc.materials[triMat].has_bump != 0);
                    if (TEX) {
                        sp = surface_point_tables(tex_tables(sc.materials), leaf, p0, p1, p2, h, sc.materials[triMat].has_bump != 0, &dndu, &dndv);
                        float tu, tv;
                        tri_uv_frame(p0, p1, p2, h, tri_uvs(tex_tables(sc.materials), leaf), &tu, &tv, &dpdu, &dpdv);
                    }
                }
                // (image-textured materials have no specular lobe, so the host template's lobe list gives the same answers here)
                Bsdf<LM_ALL> bsdf;
                bsdf.mat = sc.materials + triMat; bsdf.ns = sp.ns; bsdf.ng = sp.n; bsdf.ss = sp.ss; bsdf.ts = sp.ts;
                const V3 woN = normalize(-rd);
                while (stage < 2 && !emitted) {
                    const int type = stage == 0 ? (BSDF_REFLECTION | BSDF_SPECULAR) : (BSDF_TRANSMISSION | BSDF_SPECULAR);
                    ++stage;
                    float u0, u1, pdf = 0;
                    ss.get2d(&u0, &u1);
                    V3 wi;
                    int sampledType;
                    Spec fs = bsdf.sample_f(woN, &wi, u0, u1, &pdf, type, &sampledType);
                    if (pdf > 0.f && !fs.is_black() && absdot(wi, sp.ns) != 0.f) {
                        V3 o2 = offset_ray_origin(sp.p, sp.pError, sp.n, wi);
                        pa.ray_o[(size_t)path * kRS] = make_float4(o2.x, o2.y, o2.z, GX_INF);
                        pa.ray_d[(size_t)path * kRS] = make_float4(wi.x, wi.y, wi.z, __int_as_float(-1));
                        wa.fr_w[fi] = make_float4(fs.r, fs.g, fs.b, absdot(wi, sp.ns));
                        wa.fr_pdf[fi] = pdf;
                        if (TEX) {   // the child's offset rays, Integrator.cpp:335-354 / 376-436
                            const RayDiff mine = load_ray_diff(wa, fi);
                            const UVDiff ud = compute_differentials(mine, sp.p, sp.n, dpdu, dpdv);
                            const RayDiff child = stage == 1 ? reflect_differentials(mine, ud, sp.p, sp.ns, dndu, dndv, woN, wi)
                                                             : transmit_differentials(mine, ud, sp.p, sp.ns, dndu, dndv, bsdf.mat->eta, woN, wi);
                            store_ray_diff(wa, (size_t)(f + 1) * cap + path, child);
                        }
                        emitted = true;
                    }
                }
                wa.fr_L[fi] = make_float4(L4.x, L4.y, L4.z, __int_as_float(stage));
            }
            if (emitted) { traceClosest = true; break; }
            // the frame is complete
            if (pend == f) break;   // ... but its shadow rays are still in flight: wait one round
            result = Spec(L4.x, L4.y, L4.z);
            --top;
            haveResult = true;
        }
        const bool waiting = !traceClosest && !done;
        wa.ws[path] = make_int4(top, ss.dim, pend, (waiting ? 1 : 0) | (pendArrays ? 2 : 0) | (nArrayVertices << 2));
        pa.pflags[path] = (unsigned char)(done ? 0 : (1 | (traceShadow ? 2 : 0) | (traceClosest ? 4 : 0)));
    }
    if (nShadow) atomicAdd(ray_counts, nShadow);
    if (nMis) atomicAdd(ray_counts + 1, nMis);
}

}  // namespace gnxr

// kernels.hip.h -- the wavefront stages of the path tracer as hand-written HIP kernels for gfx950.
//
// Pipeline per pass (a pass = `k` consecutive Halton samples of every pixel this shard owns):
//
//   k_raygen        Sampler::GetCameraSample + PerspectiveCamera::GenerateRayDifferential      (A2, A4)
//   loop over path vertices:
//     k_trace       BVHAccel::Intersect / IntersectP for the continuation, shadow and MIS rays   (A5, A6)
//                   (trace_kernel.hip.h)
//     k_nee_combine L += beta * Ld once the two visibility results of the vertex are known       (A16)
//     k_shade       PathIntegrator::Li body at one vertex: Le, BSDF, UniformSampleOneLight set-up
//                   (shadow ray + MIS ray records), BSDF sampling, Russian roulette, compaction  (A7-A21)
//   k_resolve       colObj += Li in sample order, box average                                   (A1)
//
// State lives in HBM as 32-byte records grouped by consumer plus a few plain arrays (PathArrays below); queues hold path slots and
// are compacted order-preservingly with wave64 ballots + a tile scan (compact_kernel.hip.h: no atomics).
// No MFMA: the work is BVH pointer chasing and divergent shading, bound by VALU issue and memory latency.
#pragma once
#include "device_bsdf.h"
#include "device_lights.h"
#include "device_texture.h"

namespace gnxr {

constexpr int kBlock = 256;

// Per-path state lives in 32-byte RECORDS, not in one array per field: late in a pass the live paths are a sparse subset of the slots, every
// access is then one 16-byte element per lane, and a kernel that needs four fields of a path pays four memory transactions where two
// records cost two.  Five groups of two float4, by who reads them together, so that nobody drags bytes it does not want through the
// memory system (k_trace reads exactly the ray, k_nee_combine exactly the weights, k_shade two records where it read four arrays):
//   0 ray {ray_o, ray_d}   1 throughput {beta, meta}   2 shadow ray {sh_o, sh_d}   3 NEE weights {sh_X, nbeta}   4 MIS ray {mis_o, mis_d}
// (64-byte records -- groups 0+1 and 2+3 in one -- make k_shade another 4 % faster and k_trace 3 % slower, a net loss:
// profiles/r03_ab_records.log.)  The pointers below address the first element of their field and element i sits at [i * kRS]
// (meta: [i * kRSm]).  What is streamed densely (L by k_resolve, hit by the compaction), is a byte (pflags, pclass) or has no partner
// (mis_Y) stays an array of its own.
constexpr int kRS = 2;          // float4 stride of the record fields
constexpr int kRSm = 2 * kRS;   // the same stride in uint2 units (meta)
constexpr int kRecGroups = 5;
struct PathArrays {
    float4 *ray_o;    // [0] origin.xyz, tMax
    float4 *ray_d;    // [0] direction.xyz, medium (int bits)
    float4 *beta;     // [1] beta.rgb, etaScale
    uint2 *meta;      // [1] x: Halton sample index, y: dim | bounces << 16 | specularBounce << 31   (8 of 16 bytes)
    float4 *L;        // L.rgb
    int *hit;         // leaf-order triangle of the closest hit, -1 == miss
    unsigned char *pflags;  // written by k_shade: bit0 continues, bit1 NEE record, bit2 shadow ray, bit3 MIS ray
    unsigned char *pclass;  // written by k_trace: shade-kernel class of the hit material (DMaterial::shade_class)
    // next-event-estimation records written by k_shade, consumed by k_nee
    float4 *sh_o;     // [2] shadow ray origin, tMax
    float4 *sh_d;     // [2] shadow ray direction, flags (bit0 shadow ray valid, bit1 MIS ray valid)
    float4 *sh_X;     // [3] f*Li*w/lightPdf, w: light-selection pdf
    float4 *nbeta;    // [3] beta at the vertex
    float4 *mis_o;    // [4] MIS ray origin, w: expected leaf triangle (int bits; -1 == expects a miss)
    float4 *mis_d;    // [4] MIS ray direction
    float4 *mis_Y;    // f*Li*w/scatteringPdf if the expectation holds (plain array)
    unsigned int *nee_vis;   // PathIntegrator: per path [0] shadow ray unoccluded, [1] MIS ray found what it expects (bytes written by k_trace), [2] record flags (k_shade)
    // bind the record fields to the five groups
    __host__ __device__ void bind_records(float4 *const g[kRecGroups]) {
        ray_o = g[0]; ray_d = g[0] + 1; beta = g[1]; meta = reinterpret_cast<uint2 *>(g[1] + 1);
        sh_o = g[2]; sh_d = g[2] + 1; sh_X = g[3]; nbeta = g[3] + 1;
        mis_o = g[4]; mis_d = g[4] + 1;
    }
    // meta is stored with its 8 bytes of padding, so that a record is always written whole (no partial-sector write)
    __device__ void store_meta(size_t i, uint32_t x, uint32_t y) const { *reinterpret_cast<uint4 *>(&meta[i * kRSm]) = make_uint4(x, y, 0u, 0u); }
    // the same arrays seen from slot `base` on (a region of the state arrays)
    __host__ __device__ PathArrays at(size_t base) const {
        PathArrays q = *this;
        q.ray_o += base * kRS; q.ray_d += base * kRS; q.beta += base * kRS; q.meta += base * kRSm;
        q.sh_o += base * kRS; q.sh_d += base * kRS; q.sh_X += base * kRS; q.nbeta += base * kRS;
        q.mis_o += base * kRS; q.mis_d += base * kRS;
        q.mis_Y += base; q.L += base; q.hit += base; q.pflags += base; q.pclass += base; q.nee_vis += base;
        return q;
    }
};

constexpr int kMaxRegions = 8;   // sub-passes in flight at most (each in its own region of the state arrays)
struct Counters {
    unsigned long long nodes, tris;
    unsigned int q_next, q_nee, q_shadow, q_mis;   // k_compact_scan totals: paths that continue / have NEE / shadow rays / MIS rays
    unsigned int q_low;                             // ... / continue and live in the lower half of the state arrays (follows q_mis: the fifth total)
    unsigned int q_class[4];                        // fill counts of the per-material-class shade queues (3: image-textured)
    unsigned int cursor;                            // k_trace work cursor
    unsigned long long whitted_shadow;              // shadow rays queued by k_whitted_step
    unsigned long long whitted_mis;                 // MIS closest-hit rays queued by k_whitted_step (DirectLighting); follows whitted_shadow
    unsigned long long media_steps;                 // tracking-loop iterations of k_vol_media<COUNT>
    unsigned long long media_cont;                  // segments k_vol_media left at its step cap (each is traced and handed to it once more)
    unsigned long long retests;                     // k_trace<COUNT, WIDE>: leaf boxes re-tested against a shrunken tMax (32 B each)
    unsigned long long nodes_global;                // k_trace4<COUNT>: node visits served from global memory (the others come from the LDS copy of the top of the tree)
    // ---- the device-driven PathIntegrator loop (api.hip): the host never waits for these, it reads lagging copies
    unsigned int n_queue;                           // entries of the trace queue after k_queue_merge put a new sub-pass in
    unsigned int iter;                              // stamp: the loop iteration whose shade stage produced the counts above
    unsigned int region_lb[kMaxRegions + 1];        // position in the survivors' queue of the first slot of each state region (k_loop_tail)
    unsigned int region_alive[kMaxRegions];         // paths of each region that continue
    unsigned long long rays_continue, rays_shadow, rays_mis;   // summed over the iterations: continuation rays of surviving paths, shadow rays, MIS rays
};

struct DScene {
    const float4 *nodes;
    const float4 *nodes4;   // DNode4[] as 8 float4 each
    int root4;
    const DTri *tris;
    const unsigned char *tri_class;   // shade class of each leaf-order triangle's material (CompiledScene::tri_class)
    const float4 *leaf_box;   // 2 float4 per leaf-order triangle: bounds of the leaf that starts there (CompiledScene::leaf_boxes)
    int leaf1_from_verts;     // every one-triangle leaf's bounds equal the min / max of its vertices (checked by the scene compiler): no table read for them
    const DSphere *spheres;   // tested before the BVH; hit code -2 - index
    int n_spheres;
    int escape_class;         // shade queue of a continuation ray that hits nothing in a scene WITH infinite lights (it still has to collect their Le):
                              // 3 = a queue of its own (k_shade_escape) when the scene has no image-textured materials, else 0 (the diffuse queue, as before)
    const DMaterial *materials;   // materials[-1] holds the DTexTables of the scene (tex_tables(), device_texture.h)
    DLightTables lt;
    DSamplerTables st;
};

struct DRender {
    DCamera cam;
    int W, H, spp, max_depth;
    float rr_threshold;
    int shard_index, shard_count, shard_rows;
    int npix;       // pixels owned by this shard (local rows * W)
};

}  // namespace gnxr
#include "compact_kernel.hip.h"
#include "trace_kernel.hip.h"
namespace gnxr {

// local pixel -> raster coordinates for the row-interleaved sharding
GX_DEV void local_pixel(const DRender &r, int lp, int *x, int *y) {
    int row = lp / r.W;
    *x = lp - row * r.W;
    int blk = row / r.shard_rows;
    *y = (blk * r.shard_count + r.shard_index) * r.shard_rows + (row - blk * r.shard_rows);
}

// Sampler::GetCameraSample (core/Sampler.cpp:14-20) + PerspectiveCamera::GenerateRayDifferential main ray
// (camera/Perspective.cpp:62-112) + Transform::operator()(Ray) (Transform.h:230-244)
GX_DEV void camera_ray(const DCamera &cam, const DSamplerTables &st, int px, int py, uint32_t index, V3 *o, V3 *d, float *tMax, int *dimOut) {
    SampleStream s(st, index, 0);
    float fx, fy, lx, ly;
    s.get2d(&fx, &fy);
    float pfx = (float)px + fx, pfy = (float)py + fy;
    // CameraSample::time and ::pLens are drawn by every GetCameraSample; their VALUES are read only by a thin-lens camera (time: never --
    // start == end transform), so a pinhole camera just steps over the three dimensions (3 of the 5 radical inverses of k_raygen)
    s.dim += 1;  // time
    if (cam.lens_radius > 0) s.get2d(&lx, &ly); else { s.dim += 2; lx = ly = 0.f; }
    V3 pCamera = xform_point(cam.r2c, V3(pfx, pfy, 0));
    V3 dir = normalize(V3(pCamera.x, pCamera.y, pCamera.z));
    V3 oc(0, 0, 0), dc = dir;
    if (cam.ortho) { oc = pCamera; dc = V3(0, 0, 1); }   // OrthographicCamera::GenerateRayDifferential, camera/Orthographic.cpp:45-47
    if (cam.lens_radius > 0) {
        float dx, dy;
        concentric_sample_disk(lx, ly, &dx, &dy);
        float plx = cam.lens_radius * dx, ply = cam.lens_radius * dy;
        float ft = cam.focal_distance / dc.z;
        V3 pFocus = oc + dc * ft;
        oc = V3(plx, ply, 0);
        dc = normalize(pFocus - oc);
    }
    // (*this)(r.o, &oError), Transform.h:259-283
    const float *m = cam.c2w;
    float x = oc.x, y = oc.y, z = oc.z;
    float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    float wp = (m[12] * x + m[13] * y) + (m[14] * z + m[15]);
    float xAbs = (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
    float yAbs = (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
    float zAbs = (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    V3 oError = GX_GAMMA(3) * V3(xAbs, yAbs, zAbs);
    V3 ow = (wp == 1) ? V3(xp, yp, zp) : V3((1.f / wp) * xp, (1.f / wp) * yp, (1.f / wp) * zp);
    V3 dw = xform_vector(m, dc);
    float lengthSquared = length_sq(dw);
    float tm = GX_INF;
    if (lengthSquared > 0) {
        float dt = dot(vabs(dw), oError) / lengthSquared;
        ow = ow + dw * dt;
        tm -= dt;
    }
    *o = ow; *d = dw; *tMax = tm; *dimOut = s.dim;
}

// The offset rays of the same camera sample: PerspectiveCamera::GenerateRayDifferential (camera/Perspective.cpp:86-110),
// Transform::operator()(RayDifferential) (Transform.h:246-256: plain point / vector transforms) and
// ray.ScaleDifferentials(1 / sqrt(samplesPerPixel)) (core/Integrator.cpp:283, Geometry.h:874-880).  A pure function of
// (pixel, Halton index), so a kernel that needs them recomputes them instead of carrying 48 bytes per path.
GX_DEV RayDiff camera_ray_diff(const DCamera &cam, const DSamplerTables &st, int px, int py, uint32_t index, int spp) {
    V3 o, d;
    float tMax;
    int dim;
    camera_ray(cam, st, px, py, index, &o, &d, &tMax, &dim);
    SampleStream s(st, index, 0);
    float fx, fy, lx, ly;
    s.get2d(&fx, &fy);
    s.dim += 1;
    if (cam.lens_radius > 0) s.get2d(&lx, &ly); else { s.dim += 2; lx = ly = 0.f; }
    V3 pCamera = xform_point(cam.r2c, V3((float)px + fx, (float)py + fy, 0));
    V3 dxCamera = xform_point(cam.r2c, V3(1, 0, 0)) - xform_point(cam.r2c, V3(0, 0, 0));
    V3 dyCamera = xform_point(cam.r2c, V3(0, 1, 0)) - xform_point(cam.r2c, V3(0, 0, 0));
    V3 rxO, ryO, rxD, ryD;
    if (cam.ortho) {   // camera/Orthographic.cpp:62-78; dxCamera = RasterToCamera(Vector3f(1, 0, 0)) (Orthographic.h:22-23)
        V3 dxo = xform_vector(cam.r2c, V3(1, 0, 0)), dyo = xform_vector(cam.r2c, V3(0, 1, 0));
        if (cam.lens_radius > 0) {
            float ddx, ddy;
            concentric_sample_disk(lx, ly, &ddx, &ddy);
            float plx = cam.lens_radius * ddx, ply = cam.lens_radius * ddy;
            // ray->d after the lens update: Normalize(pFocus - pLens) with pFocus = pCamera + (focalDistance / 1) * (0, 0, 1)
            V3 pF0 = pCamera + V3(0, 0, 1) * (cam.focal_distance / 1.f);
            V3 dMain = normalize(pF0 - V3(plx, ply, 0));
            float ft = cam.focal_distance / dMain.z;
            V3 pFocus = pCamera + dxo + (ft * V3(0, 0, 1));
            rxO = V3(plx, ply, 0);
            rxD = normalize(pFocus - rxO);
            pFocus = pCamera + dyo + (ft * V3(0, 0, 1));
            ryO = V3(plx, ply, 0);
            ryD = normalize(pFocus - ryO);
        } else {
            rxO = pCamera + dxo;
            ryO = pCamera + dyo;
            rxD = ryD = V3(0, 0, 1);
        }
    } else if (cam.lens_radius > 0) {
        float ddx, ddy;
        concentric_sample_disk(lx, ly, &ddx, &ddy);
        float plx = cam.lens_radius * ddx, ply = cam.lens_radius * ddy;
        V3 dx = normalize(pCamera + dxCamera);
        float ft = cam.focal_distance / dx.z;
        V3 pFocus = V3(0, 0, 0) + (ft * dx);
        rxO = V3(plx, ply, 0);
        rxD = normalize(pFocus - rxO);
        V3 dy = normalize(pCamera + dyCamera);
        ft = cam.focal_distance / dy.z;
        pFocus = V3(0, 0, 0) + (ft * dy);
        ryO = V3(plx, ply, 0);
        ryD = normalize(pFocus - ryO);
    } else {
        rxO = ryO = V3(0, 0, 0);
        rxD = normalize(pCamera + dxCamera);
        ryD = normalize(pCamera + dyCamera);
    }
    RayDiff rdf;
    rdf.has = true;
    rdf.rxo = xform_point(cam.c2w, rxO);
    rdf.ryo = xform_point(cam.c2w, ryO);
    rdf.rxd = xform_vector(cam.c2w, rxD);
    rdf.ryd = xform_vector(cam.c2w, ryD);
    const float sc = 1 / gx_sqrt((float)(long long)spp);
    rdf.rxo = o + (rdf.rxo - o) * sc;
    rdf.ryo = o + (rdf.ryo - o) * sc;
    rdf.rxd = d + (rdf.rxd - d) * sc;
    rdf.ryd = d + (rdf.ryd - d) * sc;
    return rdf;
}

// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(kBlock) k_raygen(DScene sc, DRender r, PathArrays pa, int n_paths, int s0) {
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_paths; slot += gridDim.x * blockDim.x) {
        int j = slot / r.npix;
        int lp = slot - j * r.npix;
        int px, py;
        local_pixel(r, lp, &px, &py);
        uint32_t index = halton_pixel_offset(sc.st.h, px, py) + (uint32_t)(s0 + j) * (uint32_t)sc.st.h.stride;
        V3 o, d;
        float tMax;
        int dim;
        camera_ray(r.cam, sc.st, px, py, index, &o, &d, &tMax, &dim);
        pa.ray_o[(size_t)slot * kRS] = make_float4(o.x, o.y, o.z, tMax);
        pa.ray_d[(size_t)slot * kRS] = make_float4(d.x, d.y, d.z, __int_as_float(r.cam.medium));
        pa.beta[(size_t)slot * kRS] = make_float4(1.f, 1.f, 1.f, 1.f);
        pa.L[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        pa.store_meta(slot, index, (uint32_t)dim);
    }
}

// ------------------------------------------------------------------------------------------------
// EstimateDirect, core/Integrator.cpp:93-210 (handleMedia = false, specular = false), up to the two visibility rays: writes
// the NEE record `rec` (shadow ray + weighted light-sample term X, MIS ray + weighted BSDF-sample term Y and what that ray
// must find) and returns its flags (bit0 shadow ray, bit1 MIS ray; 0 = nothing written).  `xw` travels in sh_X.w.
template <uint32_t LM, int LT>
GX_DEV int estimate_direct_record(const DScene &sc, const Bsdf<LM> &bsdf, const SurfacePoint &sp, V3 woN, int lightNum, float ul0, float ul1, float us0,
                                  float us1, const PathArrays &pa, size_t rec, float xw) {
    const int bsdfFlags = BSDF_ALL & ~BSDF_SPECULAR;
    int nflags = 0;
    V3 so, sd, mo, wi2;
    Spec X(0.f), Y(0.f);
    int expect = -1;
    LightSample ls = light_sample<LT>(sc.lt, lightNum, sp.p, ul0, ul1);
    float scatteringPdf = 0;
    if (ls.pdf > 0 && !ls.Li.is_black()) {
        Spec f = bsdf.f(woN, ls.wi, bsdfFlags) * absdot(ls.wi, sp.ns);
        scatteringPdf = bsdf.pdf(woN, ls.wi, bsdfFlags);
        if (!f.is_black()) {
            // visibility.Unoccluded(scene): shadow ray p0.SpawnRayTo(p1), Light.cpp:28-31
            spawn_ray_to(sp.p, sp.pError, sp.n, ls.p1, ls.p1Error, ls.n1, &so, &sd);
            // IsDeltaLight: `Ld += f * Li / lightPdf` (:157-158) == the weighted form with weight 1, and no BSDF-sampling half (:168)
            float weight = light_is_delta<LT>(sc.lt.lights[lightNum]) ? 1.f : power_heuristic(ls.pdf, scatteringPdf);
            X = f * ls.Li * weight / ls.pdf;
            nflags |= 1;
        }
    }
    if (!light_is_delta<LT>(sc.lt.lights[lightNum])) {
        int sampledType;
        Spec f = bsdf.sample_f(woN, &wi2, us0, us1, &scatteringPdf, bsdfFlags, &sampledType);
        f = f * absdot(wi2, sp.ns);
        bool sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
        if (!f.is_black() && scatteringPdf > 0) {
            float weight = 1;
            bool skip = false;
            if (!sampledSpecular) {
                float lightPdf = light_pdf<LT>(sc.lt, lightNum, sp.p, sp.pError, sp.n, wi2);
                if (lightPdf == 0) skip = true;  // `return Ld`
                else weight = power_heuristic(scatteringPdf, lightPdf);
            }
            if (!skip) {
                // closest-hit ray isect.SpawnRay(wi) (Integrator.cpp:193-197); what it must find for
                // the light to contribute is known up front: this light's triangle, or nothing.
                const DLight &lt = sc.lt.lights[lightNum];
                mo = offset_ray_origin(sp.p, sp.pError, sp.n, wi2);
                Spec Li2;
                if (LT == LT_AREA || lt.type == GNXR_LIGHT_AREA_TRI) {
                    V3 lp0(lt.p0[0], lt.p0[1], lt.p0[2]), lp1(lt.p1[0], lt.p1[1], lt.p1[2]), lp2(lt.p2[0], lt.p2[1], lt.p2[2]);
                    V3 ln = normalize(cross(lp0 - lp2, lp1 - lp2));  // lightIsect.n
                    Li2 = area_L(lt, ln, -wi2);
                    expect = lt.tri_leaf;
                } else {
                    Li2 = light_Le<LT>(sc.lt, lightNum, mo, wi2);
                    expect = -1;
                }
                if (!Li2.is_black()) Y = f * Li2 * Spec(1.f) * weight / scatteringPdf;
                nflags |= 2;  // traced (and counted) even when Li2 is black, as in the reference
            }
        }
    }
    if (nflags) {
        pa.sh_o[(size_t)rec * kRS] = make_float4(so.x, so.y, so.z, 1 - GX_SHADOW_EPS);
        pa.sh_d[(size_t)rec * kRS] = make_float4(sd.x, sd.y, sd.z, __int_as_float(nflags));
        pa.sh_X[(size_t)rec * kRS] = make_float4(X.r, X.g, X.b, xw);
        if (nflags & 2) {
            pa.mis_o[(size_t)rec * kRS] = make_float4(mo.x, mo.y, mo.z, __int_as_float(expect));
            pa.mis_d[(size_t)rec * kRS] = make_float4(wi2.x, wi2.y, wi2.z, 0.f);
            pa.mis_Y[rec] = make_float4(Y.r, Y.g, Y.b, 0.f);
        }
    }
    return nflags;
}
// Ld of one NEE record once both visibility rays are traced: `Ld += f * Li * weight / lightPdf` (:160) if unoccluded, then
// `Ld += f * Li * Tr * weight / scatteringPdf` (:204) if the MIS ray found the light
GX_DEV Spec nee_record_Ld(const PathArrays &pa, size_t rec, float *xw) {
    float4 sd4 = pa.sh_d[(size_t)rec * kRS], X4 = pa.sh_X[(size_t)rec * kRS];
    int flags = __float_as_int(sd4.w);
    Spec Ld(0.f);
    if ((flags & 1) && pa.sh_o[(size_t)rec * kRS].w == 1.f) Ld = Ld + Spec(X4.x, X4.y, X4.z);
    if (flags & 2) {
        float4 Y4 = pa.mis_Y[rec];
        Spec Y(Y4.x, Y4.y, Y4.z);
        if (pa.mis_o[(size_t)rec * kRS].w == 1.f && !Y.is_black()) Ld = Ld + Y;
    }
    *xw = X4.w;
    return Ld;
}

// ------------------------------------------------------------------------------------------------
// One specialisation per (lobe set LM, light-type set LT): device_bsdf.h LM_*, device_lights.h LT_*.  `n_dev`
// points at the fill count of `queue` written by k_compact_scan (device-side, no host round trip).
#ifndef GX_SHADE_MINWAVES
#define GX_SHADE_MINWAVES 2
#endif
#ifdef GX_SHADE_WAVES
#define GX_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(GX_SHADE_WAVES, GX_SHADE_WAVES)))
#else
// at least two waves per SIMD: with launch bounds of 256 threads alone the register allocator may take up to 512 registers per lane, and the
// Disney-class instantiations settled at 257 (256 VGPRs + 1 AGPR) -- one register past the point where a SIMD holds two waves
// (tests/test_abi.py::test_kernel_register_budgets reads the code-object notes of every kernel)
#define GX_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(GX_SHADE_MINWAVES)))
#endif
// TEX: the queue holds hits on image-textured materials (shade class 3): Kd / Ks are looked up per hit, unfiltered -- PathIntegrator
// slices the camera RayDifferential (`Ray ray(r)`, PathIntegrator.cpp:67), so ComputeDifferentials always takes its zero branch.
#ifdef GX_SHADE_STATS
// development builds only: wave-time (s_memtime ticks) per section of k_shade, summed over all waves
static __device__ unsigned long long g_shade_stats[16];
#define GX_STICK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) sst_[i] += t_ - stick_; stick_ = t_; } while (0)
#else
#define GX_STICK(i) do {} while (0)
#endif
// minimum waves per SIMD of each material class (tuning knobs: tools/build_variant.sh -DGX_SHADE_W_DIFFUSE=4 ...).  Measured on cfg 3 /
// cfg 4 (profiles/README.md, round 3): three waves for the diffuse and the glossy class (168 registers: the glossy kernels spill ~100
// dwords to scratch and still gain) -- shade -7 % / -15 %; four waves for the diffuse class lose again; the Disney class stays at two.
#ifndef GX_SHADE_W_DIFFUSE
#define GX_SHADE_W_DIFFUSE 3
#endif
#ifndef GX_SHADE_W_GLOSSY
#define GX_SHADE_W_GLOSSY 3
#endif
#ifndef GX_SHADE_W_ALL
#define GX_SHADE_W_ALL GX_SHADE_MINWAVES
#endif
// (the image-textured class: three waves as well -- PathIntegrator on the image-textured Cornell box +6 %, profiles/r03_ab_textured_occupancy.log)
#ifndef GX_SHADE_W_TEX
#define GX_SHADE_W_TEX 3
#endif
template <uint32_t LM> constexpr int shade_min_waves() { return LM == LM_DIFFUSE ? GX_SHADE_W_DIFFUSE : (LM == LM_GLOSSY ? GX_SHADE_W_GLOSSY : GX_SHADE_W_ALL); }
template <uint32_t LM, int LT, bool SPH, bool TEX = false>
__global__ void __launch_bounds__(kBlock, (TEX ? GX_SHADE_W_TEX : shade_min_waves<LM>())) k_shade(DScene sc, DRender r, PathArrays pa, const int *__restrict__ queue, const unsigned int *n_dev, int lds_dims, int lds_nperm, int lds_mats, int lds_lights) {
    extern __shared__ int shade_smem[];   // the Halton tables of dimensions [0, lds_dims): device_sampler.h LdsSampler | the scene's DMaterial[] | DLight[]
    const int n = (int)*n_dev;
    if (blockIdx.x * blockDim.x >= (unsigned)n) return;   // this block has no item: skip the table fill
    const LdsSampler lsam = lds_sampler_fill(sc.st, lds_dims, lds_nperm, shade_smem, threadIdx.x, kBlock);
    // The lobe parameters of the hit material and the sampled light are read field by field along the BSDF code -- each a dependent
    // gather that two or three waves per SIMD cannot hide.  Small scenes' tables (a few KB) are copied into LDS; `mats` / `ltab.lights`
    // then point there (generic pointers: same code either way).
    const DMaterial *mats = sc.materials;
    DLightTables ltab = sc.lt;
    {
        int *dst = shade_smem + (lds_sampler_bytes(lds_nperm, lds_dims) >> 2);
        if (!TEX && lds_mats > 0) {
            const int *src = reinterpret_cast<const int *>(sc.materials);
            const int nd = lds_mats * (int)(sizeof(DMaterial) / 4);
            for (int k = threadIdx.x; k < nd; k += kBlock) dst[k] = src[k];
            mats = reinterpret_cast<const DMaterial *>(dst);
            dst += nd;
        }
        if (lds_lights > 0) {
            const int *src = reinterpret_cast<const int *>(sc.lt.lights);
            const int nd = lds_lights * (int)(sizeof(DLight) / 4);
            for (int k = threadIdx.x; k < nd; k += kBlock) dst[k] = src[k];
            ltab.lights = reinterpret_cast<const DLight *>(dst);
        }
    }
    __syncthreads();
#ifdef GX_SHADE_STATS
    unsigned long long sst_[16] = {0};
    unsigned long long stick_ = __builtin_amdgcn_s_memtime();
#endif
    // The item's loads form a dependent chain (queue -> path state, hit -> triangle) that two or three waves per SIMD cannot hide, so the
    // queue entry is fetched two iterations and the hit one iteration ahead (k_trace wrote `hit`; nothing in this kernel changes it): the
    // state and the triangle of an item are then requested together.
    const int stride_ = (int)(gridDim.x * blockDim.x);
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int pathCur_ = i < n ? queue[i] : -1;
    int leafCur_ = pathCur_ >= 0 ? pa.hit[pathCur_] : -1;
    int pathNext_ = (long long)i + stride_ < n ? queue[i + stride_] : -1;
    for (; i < n; i += stride_) {
        bool survive = false, wantNee = false, wantShadow = false, wantMis = false;
        int path = -1;
        GX_STICK(9);
        const int leafNext_ = pathNext_ >= 0 ? pa.hit[pathNext_] : -1;
        const int pathNext2_ = (long long)i + 2ll * stride_ < n ? queue[i + 2 * stride_] : -1;
        {
            path = pathCur_;
            uint2 m = pa.meta[(size_t)path * kRSm];
            uint32_t index = m.x;
            int dim = (int)(m.y & 0xffffu), bounces = (int)((m.y >> 16) & 0xffu);
            bool specularBounce = (m.y >> 31) != 0;
            float4 o4 = pa.ray_o[(size_t)path * kRS], d4 = pa.ray_d[(size_t)path * kRS], b4 = pa.beta[(size_t)path * kRS];
            V3 ro(o4.x, o4.y, o4.z), rd(d4.x, d4.y, d4.z);
            Spec beta(b4.x, b4.y, b4.z);
            float etaScale = b4.w;
            int leaf = leafCur_;
            bool found = leaf != -1;
            V3 p0, p1, p2;
            int triMat = -1, triLight = -1;
            TriHit h;
            SurfacePoint sp;
            sp.valid = false;
            if (SPH && leaf < -1) {   // sphere
                const DSphere &sph = sc.spheres[-2 - leaf];
                triMat = sph.material;
                float tH;
                found = sphere_test(sph, ro, rd, o4.w, &tH);
                if (found) sp = sphere_surface_point(sph, ro, rd, tH, triMat >= 0 ? mats[triMat].has_bump != 0 : false);
            } else if (found) {
                const float4 *q = reinterpret_cast<const float4 *>(sc.tris + leaf);
                float4 a = q[0], b = q[1], c = q[2];
                p0 = V3(a.x, a.y, a.z); p1 = V3(b.x, b.y, b.z); p2 = V3(c.x, c.y, c.z);
                triMat = __float_as_int(b.w); triLight = __float_as_int(c.w);
                tri_hit_recompute(p0, p1, p2, ro, rd, &h);   // the traversal accepted this triangle for this ray: same arithmetic, same (t, b0, b1, b2)
                if (found) {
                    sp = surface_point(p0, p1, p2, h, triMat >= 0 ? mats[triMat].has_bump != 0 : false);
                    if (TEX) {   // per-corner uvs / shading normals (defaults when the triangle has none: same arithmetic as above)
                        V3 dndu, dndv;
                        sp = surface_point_tables(tex_tables(sc.materials), leaf, p0, p1, p2, h, triMat >= 0 ? mats[triMat].has_bump != 0 : false, &dndu, &dndv);
                    }
                    found = sp.valid;
                }
            }
            GX_STICK(0);   // state + triangle loads, tri_test, surface_point
            // PathIntegrator.cpp:101-111: emitted light at the vertex / from the environment
            // (L is read and written only by the vertices that add to it here -- the camera vertex or a vertex after a specular bounce, on a light
            // or out in the infinite lights; k_nee_combine owns the other additions: 32 B of state traffic less for every other vertex)
            if ((bounces == 0 || specularBounce) && (found ? triLight >= 0 : ltab.n_infinite > 0)) {
                const float4 L4 = pa.L[path];
                Spec L(L4.x, L4.y, L4.z);
                if (found) L = L + beta * area_L(ltab.lights[triLight], sp.n, -rd);
                else for (int k = 0; k < ltab.n_infinite; ++k) L = L + beta * light_Le<LT>(ltab, ltab.infinite[k], ro, rd);
                pa.L[path] = make_float4(L.r, L.g, L.b, 0.f);
            }
            GX_STICK(1);   // Le
            if (found && bounces < r.max_depth) {
                if (triMat < 0) {
                    // null material: skip the boundary, PathIntegrator.cpp:121-126 (bounces-- ; continue)
                    V3 o2 = offset_ray_origin(sp.p, sp.pError, sp.n, rd);
                    pa.ray_o[(size_t)path * kRS] = make_float4(o2.x, o2.y, o2.z, GX_INF);
                    survive = true;
                } else {
                    const DMaterial *mat = mats + triMat;
                    DMaterial tm;
                    if (TEX && leaf >= 0 && (mat->kd_tex | mat->ks_tex)) {
                        float tu, tv;
                        V3 dpdu, dpdv;
                        tri_uv_frame(p0, p1, p2, h, tri_uvs(tex_tables(sc.materials), leaf), &tu, &tv, &dpdu, &dpdv);
                        RayDiff none;
                        none.has = false;
                        textured_material(tex_tables(sc.materials), *mat, tu, tv, compute_differentials(none, sp.p, sp.n, dpdu, dpdv), &tm);
                        mat = &tm;
                    }
                    Bsdf<LM> bsdf;
                    bsdf.mat = mat; bsdf.ns = sp.ns; bsdf.ng = sp.n; bsdf.ss = sp.ss; bsdf.ts = sp.ts;
                    SampleStream ss(sc.st, index, dim, lsam);
                    V3 woN = normalize(-rd);  // Interaction::wo
                    // ---- UniformSampleOneLight, Integrator.cpp:57-79
                    if (mat->n_nonspecular > 0 && ltab.n_lights > 0) {
                        float lightPdfSel;
                        // (fetching the voxel's whole record ahead of the Halton value and searching it in registers was tried: +2.4 % shade time)
                        int lightNum = light_select(ltab, sp.p, ss.get1d(), &lightPdfSel);
                        GX_STICK(2);   // light selection (1 Halton value + grid lookup)
                        if (lightPdfSel != 0) {
                            float ul0, ul1, us0, us1;
                            ss.get2d(&ul0, &ul1);
                            ss.get2d(&us0, &us1);
                            GX_STICK(3);   // 4 Halton values
                            // ---- EstimateDirect, Integrator.cpp:93-210 (handleMedia = false, specular = false): the arithmetic of
                            // estimate_direct_record above, kept in line here (the factored call cost k_shade 3.5 % on cfg 3)
                            const int bsdfFlags = BSDF_ALL & ~BSDF_SPECULAR;
                            int nflags = 0;
                            V3 so, sd, mo, wi2;
                            Spec X(0.f), Y(0.f);
                            int expect = -1;
                            LightSample ls = light_sample<LT>(ltab, lightNum, sp.p, ul0, ul1);
                            GX_STICK(4);   // light_sample
                            float scatteringPdf = 0;
                            if (ls.pdf > 0 && !ls.Li.is_black()) {
                                Spec f = bsdf.f(woN, ls.wi, bsdfFlags) * absdot(ls.wi, sp.ns);
                                scatteringPdf = bsdf.pdf(woN, ls.wi, bsdfFlags);
                                if (!f.is_black()) {
                                    // visibility.Unoccluded(scene): shadow ray p0.SpawnRayTo(p1), Light.cpp:28-31
                                    spawn_ray_to(sp.p, sp.pError, sp.n, ls.p1, ls.p1Error, ls.n1, &so, &sd);
                                    float weight = light_is_delta<LT>(ltab.lights[lightNum]) ? 1.f : power_heuristic(ls.pdf, scatteringPdf);
                                    X = f * ls.Li * weight / ls.pdf;
                                    nflags |= 1;
                                }
                            }
                            GX_STICK(5);   // BSDF f / pdf towards the light sample, shadow ray
                            if (!light_is_delta<LT>(ltab.lights[lightNum])) {
                                int sampledType;
                                Spec f = bsdf.sample_f(woN, &wi2, us0, us1, &scatteringPdf, bsdfFlags, &sampledType);
                                f = f * absdot(wi2, sp.ns);
                                bool sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
                                GX_STICK(6);   // BSDF sample_f of the MIS half
                                if (!f.is_black() && scatteringPdf > 0) {
                                    float weight = 1;
                                    bool skip = false;
                                    if (!sampledSpecular) {
                                        float lightPdf = light_pdf<LT>(ltab, lightNum, sp.p, sp.pError, sp.n, wi2);
                                        if (lightPdf == 0) skip = true;  // `return Ld`
                                        else weight = power_heuristic(scatteringPdf, lightPdf);
                                    }
                                    if (!skip) {
                                        // closest-hit ray isect.SpawnRay(wi) (Integrator.cpp:193-197); what it must find for
                                        // the light to contribute is known up front: this light's triangle, or nothing.
                                        const DLight &lt = ltab.lights[lightNum];
                                        mo = offset_ray_origin(sp.p, sp.pError, sp.n, wi2);
                                        Spec Li2;
                                        if (LT == LT_AREA || lt.type == GNXR_LIGHT_AREA_TRI) {
                                            V3 lp0(lt.p0[0], lt.p0[1], lt.p0[2]), lp1(lt.p1[0], lt.p1[1], lt.p1[2]), lp2(lt.p2[0], lt.p2[1], lt.p2[2]);
                                            V3 ln = normalize(cross(lp0 - lp2, lp1 - lp2));  // lightIsect.n
                                            Li2 = area_L(lt, ln, -wi2);
                                            expect = lt.tri_leaf;
                                        } else {
                                            Li2 = light_Le<LT>(ltab, lightNum, mo, wi2);
                                            expect = -1;
                                        }
                                        if (!Li2.is_black()) Y = f * Li2 * Spec(1.f) * weight / scatteringPdf;
                                        nflags |= 2;  // traced (and counted) even when Li2 is black, as in the reference
                                    }
                                }
                            }
                            GX_STICK(7);   // light_pdf + MIS record
                            if (nflags) {
                                pa.sh_o[(size_t)path * kRS] = make_float4(so.x, so.y, so.z, 1 - GX_SHADOW_EPS);
                                pa.sh_d[(size_t)path * kRS] = make_float4(sd.x, sd.y, sd.z, __int_as_float(nflags));
                                pa.sh_X[(size_t)path * kRS] = make_float4(X.r, X.g, X.b, lightPdfSel);
                                if (nflags & 2) {
                                    pa.mis_o[(size_t)path * kRS] = make_float4(mo.x, mo.y, mo.z, __int_as_float(expect));
                                    pa.mis_d[(size_t)path * kRS] = make_float4(wi2.x, wi2.y, wi2.z, 0.f);
                                    pa.mis_Y[path] = make_float4(Y.r, Y.g, Y.b, 0.f);
                                }
                                pa.nbeta[(size_t)path * kRS] = make_float4(beta.r, beta.g, beta.b, 0.f);
                                // byte 2: the record's flags; bit 7 of it: beta is not finite (k_nee_combine may then not skip a vertex whose rays both failed:
                                // the reference's L += beta * 0 would poison the pixel, and so must this)
                                const bool betaFinite = __builtin_isfinite(beta.r) && __builtin_isfinite(beta.g) && __builtin_isfinite(beta.b);
                                pa.nee_vis[path] = ((unsigned)nflags | (betaFinite ? 0u : 0x80u)) << 16;
                                wantNee = true;
                                wantShadow = (nflags & 1) != 0;
                                wantMis = (nflags & 2) != 0;
                            }
                        }
                    }
                    GX_STICK(8);   // NEE record stores
                    // ---- BSDF sampling for the next path vertex, PathIntegrator.cpp:144-163
                    V3 wo = -rd, wi;
                    float pdf, u0, u1;
                    int flags;
                    ss.get2d(&u0, &u1);
                    Spec f = bsdf.sample_f(wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
                    if (!(f.is_black() || pdf == 0.f)) {
                        beta = beta * (f * absdot(wi, sp.ns) / pdf);
                        specularBounce = (flags & BSDF_SPECULAR) != 0;
                        if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                            float eta = mat->eta;
                            etaScale *= (dot(wo, sp.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                        }
                        V3 o2 = offset_ray_origin(sp.p, sp.pError, sp.n, wi);
                        // Russian roulette, PathIntegrator.cpp:198-204
                        Spec rrBeta = beta * etaScale;
                        survive = true;
                        if (rrBeta.max_value() < r.rr_threshold && bounces > 3) {
                            float q = fmaxf(.05f, 1 - rrBeta.max_value());
                            if (ss.get1d() < q) survive = false;
                            else beta = beta / (1 - q);
                        }
                        if (survive) {
                            pa.ray_o[(size_t)path * kRS] = make_float4(o2.x, o2.y, o2.z, GX_INF);
                            pa.ray_d[(size_t)path * kRS] = make_float4(wi.x, wi.y, wi.z, d4.w);
                            pa.beta[(size_t)path * kRS] = make_float4(beta.r, beta.g, beta.b, etaScale);
                            pa.store_meta(path, index, (uint32_t)ss.dim | ((uint32_t)(bounces + 1) << 16) | (specularBounce ? 0x80000000u : 0u));
                        }
                    }
                }
            }
            GX_STICK(10);  // continuation: 2 Halton values, BSDF sample_f, Russian roulette, state stores
        }
        pa.pflags[path] = (unsigned char)((survive ? 1 : 0) | (wantNee ? 2 : 0) | (wantShadow ? 4 : 0) | (wantMis ? 8 : 0));
        pathCur_ = pathNext_; leafCur_ = leafNext_; pathNext_ = pathNext2_;
    }
#ifdef GX_SHADE_STATS
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 16; ++k) if (sst_[k]) atomicAdd(&g_shade_stats[k], sst_[k]);
#endif
}

// PathIntegrator.cpp:107-108 for the paths whose continuation ray escaped a scene with infinite lights: `L += beta * light->Le(ray)` for every
// infinite light when the vertex before was the camera or specular, and the path ends.  Binned as a class of their own (DScene::escape_class)
// these paths no longer sit idle in the lanes of the diffuse shade kernel (cfg 4: one continuation ray in six leaves through the open
// front of the box).  Same arithmetic, same order as the miss branch of k_shade.
template <int LT>
__global__ void __launch_bounds__(kBlock) k_shade_escape(DScene sc, PathArrays pa, const int *__restrict__ queue, const unsigned int *n_dev) {
    const int n = (int)*n_dev;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int path = queue[i];
        const uint2 m = pa.meta[(size_t)path * kRSm];
        const int bounces = (int)((m.y >> 16) & 0xffu);
        const bool specularBounce = (m.y >> 31) != 0;
        if (bounces == 0 || specularBounce) {
            const float4 o4 = pa.ray_o[(size_t)path * kRS], d4 = pa.ray_d[(size_t)path * kRS], b4 = pa.beta[(size_t)path * kRS], L4 = pa.L[path];
            const V3 ro(o4.x, o4.y, o4.z), rd(d4.x, d4.y, d4.z);
            const Spec beta(b4.x, b4.y, b4.z);
            Spec L(L4.x, L4.y, L4.z);
            for (int k = 0; k < sc.lt.n_infinite; ++k) L = L + beta * light_Le<LT>(sc.lt, sc.lt.infinite[k], ro, rd);
            pa.L[path] = make_float4(L.r, L.g, L.b, 0.f);
        }
        pa.pflags[path] = 0;
    }
}

// The queue of the next trace when a sub-pass starts in state region `region` = slots [base, base + n_new): the survivors of the
// sub-passes in flight (ascending slots, none of them inside the region -- it was empty) and every slot of the new one, merged in
// ascending order.  The survivors' count and the split position (how many of them lie below `base`: k_loop_tail) are read on the device;
// a launch before any shade stage has run passes first = 1 (no survivors).  The merged count goes to ctr->n_queue.
static __global__ void __launch_bounds__(kBlock) k_queue_merge(const int *__restrict__ q_old, Counters *ctr, int first, int region, int base, int n_new, int *__restrict__ q_out) {
    const unsigned n_old = first ? 0u : ctr->q_next, split = first ? 0u : ctr->region_lb[region];
    const long long total = (long long)n_old + n_new;
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int v;
        if (i < split) v = q_old[i];
        else if (i < (long long)split + n_new) v = base + (int)(i - split);
        else v = q_old[i - n_new];
        q_out[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->n_queue = (unsigned)total;
}

// After the flags compaction of a shade stage: where each state region starts in the survivors' queue (binary search: the queue is
// ascending), how many paths of each region are left, the ray totals of the next trace, and the iteration stamp the host's lagging copy
// is recognised by.
static __global__ void __launch_bounds__(64) k_loop_tail(const int *__restrict__ q_next, Counters *ctr, int n_regions, int region_size, unsigned iter) {
    __shared__ unsigned lb[kMaxRegions + 1];
    const unsigned n = ctr->q_next;
    const int r = threadIdx.x;
    if (r <= n_regions) {
        const long long key = (long long)r * region_size;
        unsigned lo = 0, hi = n;
        while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if ((long long)q_next[mid] < key) lo = mid + 1; else hi = mid; }
        lb[r] = lo;
        ctr->region_lb[r] = lo;
    }
    __syncthreads();
    if (r < n_regions) ctr->region_alive[r] = lb[r + 1] - lb[r];
    if (r == 0) {
        ctr->rays_continue += n; ctr->rays_shadow += ctr->q_shadow; ctr->rays_mis += ctr->q_mis;
        // the count of the queue the next trace and the next shade stage work on lives in n_queue (k_queue_merge adds a new sub-pass to
        // it): q_next is rewritten by the next stage's own compaction while its kernels still read the input count
        ctr->n_queue = n;
        ctr->iter = iter;
    }
}

// ------------------------------------------------------------------------------------------------
// colObj += Li(...) in sample order (core/Integrator.cpp:286), one lane per pixel, no atomics.
static __global__ void __launch_bounds__(kBlock) k_resolve(PathArrays pa, float4 *accum, int npix, int k) {
    for (int lp = blockIdx.x * blockDim.x + threadIdx.x; lp < npix; lp += gridDim.x * blockDim.x) {
        float4 a = accum[lp];
        for (int j = 0; j < k; ++j) {
            float4 l = pa.L[(size_t)j * npix + lp];
            a.x += l.x; a.y += l.y; a.z += l.z;
        }
        accum[lp] = a;
    }
}
// colObj / samplesPerPixel and the FrameBuffer layout (x + y*W)*4 + c (core/Integrator.cpp:293-310)
static __global__ void __launch_bounds__(kBlock) k_finish(DRender r, const float4 *accum, float4 *out) {
    for (int lp = blockIdx.x * blockDim.x + threadIdx.x; lp < r.npix; lp += gridDim.x * blockDim.x) {
        int x, y;
        local_pixel(r, lp, &x, &y);
        float4 a = accum[lp];
        float spp = (float)(long long)r.spp;
        out[(size_t)y * r.W + x] = make_float4(a.x / spp, a.y / spp, a.z / spp, 1.f);
    }
}

// ------------------------------------------------------------------------------------------------
// Aggregate seam kernels: Scene::Intersect / IntersectP for caller-supplied rays
template <int STACK>
__global__ void __launch_bounds__(kBlock) k_trace_closest_api(DScene sc, const gnxr_ray *rays, long long n, gnxr_hit *hits) {
    __shared__ int stack[STACK * kBlock];
    TraceCounters tc = {0, 0};
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        gnxr_ray r = rays[i];
        V3 ro(r.o[0], r.o[1], r.o[2]), rd(r.d[0], r.d[1], r.d[2]);
        TriHit h;
        float tMax = r.tmax;
        int sphereHit = -1;
        for (int si = 0; si < sc.n_spheres; ++si) { float tH; if (sphere_test(sc.spheres[si], ro, rd, tMax, &tH)) { tMax = tH; sphereHit = si; } }
        int leaf = bvh_traverse<false, kBlock, false>(sc.nodes, sc.tris, ro, rd, tMax, &stack[threadIdx.x], &h, &tc);
        gnxr_hit out;
        out.prim = -1; out.t = 0; out.b0 = out.b1 = out.b2 = 0; out.n[0] = out.n[1] = out.n[2] = 0;
        if (leaf < 0 && sphereHit >= 0) {
            SurfacePoint sp = sphere_surface_point(sc.spheres[sphereHit], ro, rd, tMax, false);
            out.prim = sc.spheres[sphereHit].prim; out.t = tMax;
            out.n[0] = sp.n.x; out.n[1] = sp.n.y; out.n[2] = sp.n.z;
        }
        if (leaf >= 0) {
            V3 p0, p1, p2;
            load_tri(sc.tris, leaf, &p0, &p1, &p2);
            V3 nn = normalize(cross(p0 - p2, p1 - p2));
            const DTexTables &tt = tex_tables(sc.materials);
            if (tt.tri_n || tt.tri_s) {   // per-vertex normals / tangents flip isect->n onto the shading side (SetShadingGeometry(..., true), Triangle.cpp:296)
                V3 dndu, dndv;
                SurfacePoint sp = surface_point_tables(tt, leaf, p0, p1, p2, h, false, &dndu, &dndv);
                if (sp.valid) nn = sp.n;
            }
            out.prim = sc.tris[leaf].prim; out.t = h.t; out.b0 = h.b0; out.b1 = h.b1; out.b2 = h.b2;
            out.n[0] = nn.x; out.n[1] = nn.y; out.n[2] = nn.z;
        }
        hits[i] = out;
    }
}
template <int STACK>
__global__ void __launch_bounds__(kBlock) k_trace_any_api(DScene sc, const gnxr_ray *rays, long long n, unsigned char *occluded) {
    __shared__ int stack[STACK * kBlock];
    TraceCounters tc = {0, 0};
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        gnxr_ray r = rays[i];
        TriHit h;
        V3 ro(r.o[0], r.o[1], r.o[2]), rd(r.d[0], r.d[1], r.d[2]);
        bool hitSphere = false;
        for (int si = 0; si < sc.n_spheres && !hitSphere; ++si) { float tH; hitSphere = sphere_test(sc.spheres[si], ro, rd, r.tmax, &tH); }
        int leaf = hitSphere ? 0 : bvh_traverse<true, kBlock, false>(sc.nodes, sc.tris, ro, rd, r.tmax, &stack[threadIdx.x], &h, &tc);
        occluded[i] = leaf >= 0 ? 1 : 0;
    }
}

// probes
static __global__ void k_halton_probe(DSamplerTables st, const int *px, const int *py, const long long *s, const int *dim, long long n, float *out) {
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint32_t index = halton_pixel_offset(st.h, px[i], py[i]) + (uint32_t)s[i] * (uint32_t)st.h.stride;
        out[i] = halton_sample(st, index, dim[i]);
    }
}
static __global__ void k_camera_probe(DSamplerTables st, DCamera cam, const int *px, const int *py, const long long *s, long long n, float *o_out, float *d_out) {
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint32_t index = halton_pixel_offset(st.h, px[i], py[i]) + (uint32_t)s[i] * (uint32_t)st.h.stride;
        V3 o, d;
        float tMax;
        int dim;
        camera_ray(cam, st, px[i], py[i], index, &o, &d, &tMax, &dim);
        o_out[3 * i] = o.x; o_out[3 * i + 1] = o.y; o_out[3 * i + 2] = o.z;
        d_out[3 * i] = d.x; d_out[3 * i + 1] = d.y; d_out[3 * i + 2] = d.z;
    }
}

// SpatialLightDistribution::ComputeDistribution (core/LightDistribution.cpp:206-274) for every voxel at scene set-up: the
// reference fills voxels lazily from the render threads; here one lane per voxel evaluates the 128 probe points x all lights
// with the same light_sample<> the integrator uses, and writes the voxel's Distribution1D (cdf[1..n], func[0..n-1], funcInt).
constexpr int kGridMaxLights = 16;
template <int LT>
__global__ void __launch_bounds__(kBlock) k_light_grid(DLightTables lt, DLightGrid g, const float *__restrict__ ri, float *__restrict__ table) {
    const int nl = g.n_lights;
    const long long nv = (long long)g.nvox[0] * g.nvox[1] * g.nvox[2];
    for (long long idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nv; idx += (long long)gridDim.x * blockDim.x) {
        const int z = (int)(idx % g.nvox[2]), y = (int)((idx / g.nvox[2]) % g.nvox[1]), x = (int)(idx / ((long long)g.nvox[2] * g.nvox[1]));
        V3 p0((float)x / (float)g.nvox[0], (float)y / (float)g.nvox[1], (float)z / (float)g.nvox[2]);
        V3 p1((float)(x + 1) / (float)g.nvox[0], (float)(y + 1) / (float)g.nvox[1], (float)(z + 1) / (float)g.nvox[2]);
        V3 lo(g.lo[0], g.lo[1], g.lo[2]), hi(g.hi[0], g.hi[1], g.hi[2]);
        V3 a(lerpf(p0.x, lo.x, hi.x), lerpf(p0.y, lo.y, hi.y), lerpf(p0.z, lo.z, hi.z));   // Bounds3::Lerp
        V3 b(lerpf(p1.x, lo.x, hi.x), lerpf(p1.y, lo.y, hi.y), lerpf(p1.z, lo.z, hi.z));
        V3 vlo(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)), vhi(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z));
        float contrib[kGridMaxLights];
        for (int j = 0; j < kGridMaxLights; ++j) contrib[j] = 0.f;
        for (int i = 0; i < 128; ++i) {
            V3 po(lerpf(ri[i], vlo.x, vhi.x), lerpf(ri[128 + i], vlo.y, vhi.y), lerpf(ri[256 + i], vlo.z, vhi.z));
            const float u0 = ri[384 + i], u1 = ri[512 + i];
#pragma unroll
            for (int j = 0; j < kGridMaxLights; ++j) {
                if (j < nl) {
                    LightSample ls = light_sample<LT>(lt, j, po, u0, u1);
                    if (ls.pdf > 0) contrib[j] += ls.Li.y() / ls.pdf;
                }
            }
        }
        float sum = 0;
#pragma unroll
        for (int j = 0; j < kGridMaxLights; ++j) if (j < nl) sum += contrib[j];
        const float avg = sum / (float)(128 * nl);
        const float minC = (avg > 0) ? (float)(.001 * (double)avg) : 1.f;
        float *dst = table + (size_t)idx * g.stride;
        // Distribution1D ctor, Sampling.h:22-35
        float cdf[kGridMaxLights + 1];
        cdf[0] = 0;
#pragma unroll
        for (int j = 0; j < kGridMaxLights; ++j) {
            if (j < nl) { contrib[j] = fmaxf(contrib[j], minC); cdf[j + 1] = cdf[j] + contrib[j] / (float)nl; }
            else cdf[j + 1] = cdf[j];
        }
        float funcInt = 0;
#pragma unroll
        for (int j = 0; j < kGridMaxLights; ++j) if (j + 1 == nl) funcInt = cdf[j + 1];
#pragma unroll
        for (int j = 0; j < kGridMaxLights; ++j) {
            if (j < nl) {
                dst[j] = (funcInt == 0) ? (float)(j + 1) / (float)nl : cdf[j + 1] / funcInt;
                dst[nl + j] = contrib[j];
            }
        }
        dst[2 * nl] = funcInt;
    }
}

// The same for any number of lights (mesh lights: one light per emissive triangle): the voxel's slice of the table is the scratch
// space.  Per light the 128 contributions are summed in the same order as above (i ascending), and the sum over the lights, the
// floor and the Distribution1D follow in light order, so the table is the same bit for bit.
template <int LT>
__global__ void __launch_bounds__(kBlock) k_light_grid_any(DLightTables lt, DLightGrid g, const float *__restrict__ ri, float *__restrict__ table) {
    const int nl = g.n_lights;
    const long long nv = (long long)g.nvox[0] * g.nvox[1] * g.nvox[2];
    for (long long idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nv; idx += (long long)gridDim.x * blockDim.x) {
        const int z = (int)(idx % g.nvox[2]), y = (int)((idx / g.nvox[2]) % g.nvox[1]), x = (int)(idx / ((long long)g.nvox[2] * g.nvox[1]));
        V3 p0((float)x / (float)g.nvox[0], (float)y / (float)g.nvox[1], (float)z / (float)g.nvox[2]);
        V3 p1((float)(x + 1) / (float)g.nvox[0], (float)(y + 1) / (float)g.nvox[1], (float)(z + 1) / (float)g.nvox[2]);
        V3 lo(g.lo[0], g.lo[1], g.lo[2]), hi(g.hi[0], g.hi[1], g.hi[2]);
        V3 a(lerpf(p0.x, lo.x, hi.x), lerpf(p0.y, lo.y, hi.y), lerpf(p0.z, lo.z, hi.z));
        V3 b(lerpf(p1.x, lo.x, hi.x), lerpf(p1.y, lo.y, hi.y), lerpf(p1.z, lo.z, hi.z));
        V3 vlo(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)), vhi(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z));
        float *dst = table + (size_t)idx * g.stride;
        float sum = 0;
        for (int j = 0; j < nl; ++j) {
            float c = 0.f;
            for (int i = 0; i < 128; ++i) {
                V3 po(lerpf(ri[i], vlo.x, vhi.x), lerpf(ri[128 + i], vlo.y, vhi.y), lerpf(ri[256 + i], vlo.z, vhi.z));
                LightSample ls = light_sample<LT>(lt, j, po, ri[384 + i], ri[512 + i]);
                if (ls.pdf > 0) c += ls.Li.y() / ls.pdf;
            }
            dst[nl + j] = c;
            sum += c;
        }
        const float avg = sum / (float)(128 * nl);
        const float minC = (avg > 0) ? (float)(.001 * (double)avg) : 1.f;
        float cdf = 0;   // Distribution1D ctor, Sampling.h:22-35: running cdf[j + 1], parked in dst[j] until funcInt is known
        for (int j = 0; j < nl; ++j) {
            const float f = fmaxf(dst[nl + j], minC);
            dst[nl + j] = f;
            cdf = cdf + f / (float)nl;
            dst[j] = cdf;
        }
        const float funcInt = cdf;
        for (int j = 0; j < nl; ++j) dst[j] = (funcInt == 0) ? (float)(j + 1) / (float)nl : dst[j] / funcInt;
        dst[2 * nl] = funcInt;
    }
}

// test hook: the device's float libm (device_math.h) on caller-supplied arguments
static __global__ void k_libm_probe(int fn, const float *x, const float *x2, long long n, float *out) {
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = x[i];
        if (fn == 8) out[i] = gx_pow(v, x2 ? x2[i] : 1.0f);
        else if (fn == 6) out[i] = gx_acos(v);
        else if (fn == 7) out[i] = gx_atan2(v, x2 ? x2[i] : 1.0f);
        else if (fn >= 4) { float sv, cv; gx_sincos(v, &sv, &cv); out[i] = fn == 4 ? sv : cv; }
        else out[i] = fn == 0 ? gx_log(v) : (fn == 1 ? gx_exp(v) : (fn == 2 ? gx_sin(v) : gx_cos(v)));
    }
}

// test hook: the double-precision libm calls of the path (device_bsdf.h: sincos / sqrt, device_math.h: tan)
static __global__ void k_libm_probe_f64(int fn, const float *x, long long n, double *out) {
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double v = (double)x[i];
        if (fn <= 1) out[i] = fn == 0 ? gx_sin_d(v) : gx_cos_d(v);
        else out[i] = fn == 2 ? sqrt(v) : tan(v);
    }
}

// FrameBuffer::update_f_u_c, ui/FrameBuffer.h:127-149
static __global__ void k_framebuffer_update(float *mean, const float *frame, long long nvals, int frame_count, unsigned char *rgba8) {
    for (long long i = blockIdx.x * blockDim.x + threadIdx.x; i < nvals; i += (long long)gridDim.x * blockDim.x) {
        if ((i & 3) == 3) { rgba8[i] = 255; continue; }  // set_uc(i, j, 3, 255)
        float weight = (1.0f / (float)frame_count);
        float fValue = weight * frame[i] + (1.0f - weight) * mean[i];
        mean[i] = fValue;
        float exposure = 0.75f;
        float temp_c = 1.0f - gx_exp(-fValue * 1.0f / (1 - exposure));
        rgba8[i] = (unsigned char)(temp_c * 255);
    }
}

}  // namespace gnxr

// ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's render hot path
// (SamplerIntegrator::Render -> PathIntegrator::Li -> BVHAccel::Intersect/IntersectP) over the
// same flat scene description the product consumes (include/gnxr.h is shared for the POD layout
// only).  Built by oracle/Makefile into oracle/libgnx_oracle.so; loaded with ctypes by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product.
//
// Pinning: every function below oracle/o_integrator.h is checked against the compiled reference
// (oracle/_ref, built from /root/reference sources without stand-in headers) by
// oracle/make_goldens.py; the goldens are committed under tests/golden/.  The integrator loop
// itself (core/Integrator.cpp, integrators/PathIntegrator.cpp, core/LightDistribution.cpp) cannot
// be compiled here (needs Qt via ui/FrameBuffer.h) and is pinned by the reference's recorded ray
// counts, see DESIGN.md.
#include <omp.h>

#include <chrono>
#include <cstdio>

#include "o_integrator.h"
#include "o_media.h"

using namespace gnxo;

struct gnxo_scene {
    Scene scene;
};

extern "C" {

int gnxo_scene_create(const gnxr_scene_desc *d, gnxo_scene **out) {
    if (!d || !out || d->abi_version != GNXR_ABI_VERSION) return -1;
    gnxo_scene *s = new gnxo_scene();
    s->scene.Load(d);
    *out = s;
    return 0;
}
void gnxo_scene_destroy(gnxo_scene *s) { delete s; }

int gnxo_scene_info(const gnxo_scene *s, int32_t *nNodes, int32_t *maxDepth) {
    *nNodes = (int)s->scene.nodes.size();
    *maxDepth = s->scene.bvhMaxDepth;
    return 0;
}
// flattened BVH for same-toolchain comparisons: nodes as 8 floats/ints each (bounds, offset, nPrims|axis<<16)
int gnxo_scene_bvh(const gnxo_scene *s, float *bounds6, int32_t *offset, int32_t *nprims, int32_t *axis, int32_t *ordered) {
    const Scene &sc = s->scene;
    for (size_t i = 0; i < sc.nodes.size(); ++i) {
        const LinearBVHNode &n = sc.nodes[i];
        bounds6[6 * i + 0] = n.bounds.pMin.x; bounds6[6 * i + 1] = n.bounds.pMin.y; bounds6[6 * i + 2] = n.bounds.pMin.z;
        bounds6[6 * i + 3] = n.bounds.pMax.x; bounds6[6 * i + 4] = n.bounds.pMax.y; bounds6[6 * i + 5] = n.bounds.pMax.z;
        offset[i] = n.offset; nprims[i] = n.nPrimitives; axis[i] = n.axis;
    }
    for (size_t i = 0; i < sc.orderedPrims.size(); ++i) ordered[i] = sc.orderedPrims[i];
    return 0;
}

// Replace the scene's BVH by one handed in (bounds6 / offset / nprims / axis per node, leaf order -> authoring index): lets the
// oracle traverse the tree the compiled reference built with another SplitMethod (HLBVH) without restating that builder.
int gnxo_scene_set_bvh(gnxo_scene *s, const float *bounds6, const int32_t *offset, const int32_t *nprims, const int32_t *axis, int64_t nNodes,
                       const int32_t *ordered, int64_t nOrdered) {
    Scene &sc = s->scene;
    sc.nodes.assign((size_t)nNodes, LinearBVHNode());
    for (int64_t i = 0; i < nNodes; ++i) {
        LinearBVHNode &n = sc.nodes[i];
        n.bounds.pMin = V3(bounds6[6 * i + 0], bounds6[6 * i + 1], bounds6[6 * i + 2]);
        n.bounds.pMax = V3(bounds6[6 * i + 3], bounds6[6 * i + 4], bounds6[6 * i + 5]);
        n.offset = offset[i]; n.nPrimitives = (uint16_t)nprims[i]; n.axis = (uint8_t)axis[i]; n.pad = 0;
    }
    sc.orderedPrims.assign(ordered, ordered + nOrdered);
    return 0;
}

// SamplerIntegrator::Render, core/Integrator.cpp:225-319.  The pixel loop ignores pixelBounds.pMin,
// box-averages without rayWeight and writes (x + y*W)*4 + c (ui/FrameBuffer.h:136).  The two libc
// rand() draws (:262-263) do not reach any output and are not reproduced.
int gnxo_render(gnxo_scene *s, const gnxr_render_params *p, float *rgba, gnxr_stats *stats, int nThreads) {
    const Scene &scene = s->scene;
    RenderContext rc;
    rc.Init(&scene, p->light_strategy);
    Camera camera(scene.camera, p->width, p->height);
    camera.medium = scene.cameraMedium;
    Halton halton(p->spp, p->width, p->height, false);
    PathParams pp; pp.maxDepth = p->max_depth; pp.rrThreshold = p->rr_threshold;
    DirectParams dp; dp.maxDepth = p->max_depth; dp.strategy = p->direct_strategy;
    if (p->integrator == GNXR_INTEGRATOR_DIRECT) dp.Preprocess(scene);   // SamplerIntegrator::Render, core/Integrator.cpp:227
    VolContext vc;
    vc.rc = &rc;
    vc.media.Init(&scene);
    rc.mediaSet = &vc;
    int sBegin = p->spp_begin, sEnd = p->spp_end > 0 ? p->spp_end : p->spp;
    int shardCount = std::max(1, p->shard_count), shardRows = std::max(1, p->shard_rows);
    scene.counters.nIntersect = 0; scene.counters.nIntersectP = 0; scene.counters.nNodes = 0; scene.counters.nTris = 0;
    if (nThreads > 0) omp_set_num_threads(nThreads);
    auto t0 = std::chrono::steady_clock::now();
    const int W = p->width, H = p->height;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < W; i++) {
        for (int j = 0; j < H; j++) {
            if ((j / shardRows) % shardCount != p->shard_index) continue;
            Spec colObj(.0f);
            for (int sidx = sBegin; sidx < sEnd; ++sidx) {
                SampleStream sampler(&halton, i, j, sidx);
                if (!dp.arraySizes.empty()) sampler.Request2DArrays(&dp.arraySizes);
                // Sampler::GetCameraSample, core/Sampler.cpp:14-20
                P2 f2 = sampler.Get2D();
                P2 pFilm((Float)i + f2.x, (Float)j + f2.y);
                (void)sampler.Get1D();  // time
                P2 pLens = sampler.Get2D();
                Ray ray = camera.GenerateRay(pFilm, pLens);
                ray.ScaleDifferentials(1 / std::sqrt((Float)(int64_t)p->spp));   // core/Integrator.cpp:283
                Spec Li;
                if (p->integrator == GNXR_INTEGRATOR_VOLPATH) Li = VolPathLi(rc, pp, ray, sampler);
                else if (p->integrator == GNXR_INTEGRATOR_WHITTED) Li = WhittedLi(rc, pp, ray, sampler, 0);
                else if (p->integrator == GNXR_INTEGRATOR_DIRECT) Li = DirectLi(rc, dp, ray, sampler, 0);
                else Li = PathLi(rc, pp, ray, sampler);
                colObj += Li;
            }
            colObj = colObj / (Float)(int64_t)p->spp;  // `colObj / pixel_sampler->samplesPerPixel` (int64 -> Float)
            size_t o = ((size_t)i + (size_t)j * W) * 4;
            rgba[o + 0] = colObj[0]; rgba[o + 1] = colObj[1]; rgba[o + 2] = colObj[2]; rgba[o + 3] = 1.f;
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->rays_closest = scene.counters.nIntersect;
        stats->rays_any = scene.counters.nIntersectP;
        stats->nodes_visited = scene.counters.nNodes;
        stats->tris_tested = scene.counters.nTris;
        stats->seconds_render = std::chrono::duration<double>(t1 - t0).count();
        stats->seconds_total = stats->seconds_render;
    }
    return 0;
}
void gnxo_set_count_traversal(gnxo_scene *s, int on) { s->scene.countTraversal = on != 0; }
int gnxo_max_dimension(int reset) { int v = SampleStream::MaxDimensionSeen().load(); if (reset) SampleStream::MaxDimensionSeen().store(0); return v; }

int gnxo_trace_closest(gnxo_scene *s, const gnxr_ray *rays, int64_t n, gnxr_hit *hits) {
    const Scene &scene = s->scene;
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) {
        Ray r(V3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), V3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].tmax);
        SurfaceInteraction isect;
        gnxr_hit h; memset(&h, 0, sizeof(h)); h.prim = -1;
        if (scene.Intersect(r, &isect)) {
            h.prim = isect.prim; h.t = isect.t; h.b0 = isect.b0; h.b1 = isect.b1; h.b2 = isect.b2;
            h.n[0] = isect.n.x; h.n[1] = isect.n.y; h.n[2] = isect.n.z;
        }
        hits[i] = h;
    }
    return 0;
}
int gnxo_trace_any(gnxo_scene *s, const gnxr_ray *rays, int64_t n, uint8_t *occluded) {
    const Scene &scene = s->scene;
#pragma omp parallel for
    for (int64_t i = 0; i < n; ++i) {
        Ray r(V3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), V3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].tmax);
        occluded[i] = scene.IntersectP(r) ? 1 : 0;
    }
    return 0;
}

int gnxo_sample_halton(int32_t width, int32_t height, const int32_t *px, const int32_t *py, const int64_t *sidx,
                       const int32_t *dim, int64_t n, float *out) {
    Halton h(1, width, height, false);
    for (int64_t i = 0; i < n; ++i) out[i] = h.SampleDimension(h.IndexForSample(px[i], py[i], sidx[i]), dim[i]);
    return 0;
}
int gnxo_camera_rays(const gnxr_camera *cam, int32_t width, int32_t height, const int32_t *px, const int32_t *py,
                     const int64_t *sidx, int64_t n, float *o_out, float *d_out) {
    Camera camera(*cam, width, height);
    Halton h(1, width, height, false);
    for (int64_t i = 0; i < n; ++i) {
        SampleStream sampler(&h, px[i], py[i], sidx[i]);
        P2 f2 = sampler.Get2D();
        P2 pFilm((Float)px[i] + f2.x, (Float)py[i] + f2.y);
        (void)sampler.Get1D();
        P2 pLens = sampler.Get2D();
        Ray r = camera.GenerateRay(pFilm, pLens);
        o_out[3 * i] = r.o.x; o_out[3 * i + 1] = r.o.y; o_out[3 * i + 2] = r.o.z;
        d_out[3 * i] = r.d.x; d_out[3 * i + 1] = r.d.y; d_out[3 * i + 2] = r.d.z;
    }
    return 0;
}

// ---- bit-exactness probes ----
void gnxo_rng_u32(int useSeq, uint64_t seq, int n, uint32_t *out) {
    RNG rng;
    if (useSeq) rng.SetSequence(seq);
    for (int i = 0; i < n; ++i) out[i] = rng.UniformUInt32();
}
int64_t gnxo_perm_table(uint16_t *out, int64_t cap) {
    const std::vector<uint16_t> &p = RadicalInversePermutations();
    int64_t n = (int64_t)p.size();
    if (out) for (int64_t i = 0; i < std::min(n, cap); ++i) out[i] = p[i];
    return n;
}
void gnxo_primes(int32_t *primes, int32_t *sums) {
    for (int i = 0; i < PrimeTableSize; ++i) { primes[i] = Primes().primes[i]; sums[i] = Primes().primeSums[i]; }
}

// ---- BSDF probe: build the BSDF at a hit of ray (o,d) and evaluate / sample it ----
// out layout per query (16 floats): f[3], pdf, sample_f[3], sample_pdf, wi[3], sampledType, nComponents, hit, 0, 0
// flags: BxDFType in the low byte; bits 8.. = 1000 * eps of synthetic ray differentials (0: none; texture-filter probe)
int gnxo_bsdf_probe(gnxo_scene *s, const gnxr_ray *rays, const float *wiW, const float *u2, int64_t n, int flagsIn, float *out) {
    const Scene &scene = s->scene;
    const int flags = flagsIn & 0xff;
    const float diffEps = (float)(flagsIn >> 8) / 1000.f;
    for (int64_t i = 0; i < n; ++i) {
        float *o = out + 16 * i;
        for (int k = 0; k < 16; ++k) o[k] = 0;
        Ray r(V3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), V3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].tmax);
        if (diffEps > 0) {
            r.hasDifferentials = true;
            r.rxOrigin = r.ryOrigin = r.o;
            r.rxDirection = r.d + V3(diffEps, 0, 0);
            r.ryDirection = r.d + V3(0, diffEps, 0);
        }
        SurfaceInteraction isect;
        if (!scene.Intersect(r, &isect)) continue;
        BSDF bsdf;
        if (!ComputeScatteringFunctions(scene, r, &isect, true, &bsdf)) continue;
        o[13] = 1; o[14] = isect.dudx; o[15] = isect.dvdy;
        V3 wi(wiW[3 * i], wiW[3 * i + 1], wiW[3 * i + 2]);
        Spec f = bsdf.f(isect.wo, wi, flags);
        o[0] = f[0]; o[1] = f[1]; o[2] = f[2];
        o[3] = bsdf.Pdf(isect.wo, wi, flags);
        V3 wis; Float pdf = 0; int st = 0;
        Spec sf = bsdf.Sample_f(isect.wo, &wis, P2(u2[2 * i], u2[2 * i + 1]), &pdf, flags, &st);
        if (pdf == 0) { sf = Spec(0.f); wis = V3(); }
        o[4] = sf[0]; o[5] = sf[1]; o[6] = sf[2]; o[7] = pdf;
        o[8] = wis.x; o[9] = wis.y; o[10] = wis.z; o[11] = (float)st; o[12] = (float)bsdf.NumComponents(flags);
    }
    return 0;
}

// ---- light probe: Sample_Li / Pdf_Li / distribution at reference points ----
// out per query (12 floats): Li[3], pdf, wi[3], pdf_li(wiQuery), lightPdfSelect(light), p1[3]
int gnxo_light_probe(gnxo_scene *s, int light, int strategy, const float *refP, const float *refN, const float *u2,
                     const float *wiQuery, int64_t n, float *out) {
    const Scene &scene = s->scene;
    RenderContext rc;
    rc.Init(&scene, strategy);
    for (int64_t i = 0; i < n; ++i) {
        float *o = out + 12 * i;
        Interaction ref;
        ref.p = V3(refP[3 * i], refP[3 * i + 1], refP[3 * i + 2]);
        ref.n = V3(refN[3 * i], refN[3 * i + 1], refN[3 * i + 2]);
        LightSample ls = rc.Sample_Li(light, ref, P2(u2[2 * i], u2[2 * i + 1]));
        o[0] = ls.Li[0]; o[1] = ls.Li[1]; o[2] = ls.Li[2]; o[3] = ls.pdf;
        o[4] = ls.wi.x; o[5] = ls.wi.y; o[6] = ls.wi.z;
        o[7] = rc.Pdf_Li(light, ref, V3(wiQuery[3 * i], wiQuery[3 * i + 1], wiQuery[3 * i + 2]));
        const Distribution1D *d = rc.Lookup(ref.p);
        o[8] = d->func[light] / (d->funcInt * d->Count());
        o[9] = ls.p1.p.x; o[10] = ls.p1.p.y; o[11] = ls.p1.p.z;
    }
    return 0;
}
// Le of escaped rays for light `light`
int gnxo_light_le(gnxo_scene *s, int light, const gnxr_ray *rays, int64_t n, float *out) {
    RenderContext rc;
    rc.Init(&s->scene, GNXR_LIGHTS_UNIFORM);
    for (int64_t i = 0; i < n; ++i) {
        Ray r(V3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), V3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].tmax);
        Spec L = rc.LightLe(light, r);
        out[3 * i] = L[0]; out[3 * i + 1] = L[1]; out[3 * i + 2] = L[2];
    }
    return 0;
}

// The host's double-precision libm on float arguments widened to double (what `cos(phi)` / `sin(phi)` / `sqrt(..)` at
// core/MicroFacet.cpp:220-223 evaluate): the checker for the device's restatement of glibc's __sin / __cos.  fn: 0 sin, 1 cos, 2 sqrt, 3 tan.
int gnxo_libm_f64(int32_t fn, const float *x, int64_t n, double *out) {
    if (!x || !out || n < 0 || fn < 0 || fn > 3) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const double v = (double)x[i];
        out[i] = fn == 0 ? sin(v) : fn == 1 ? cos(v) : fn == 2 ? sqrt(v) : tan(v);
    }
    return 0;
}

// FrameBuffer::update_f_u_c (ui/FrameBuffer.h:127-149) for every pixel and channel 0..2, as SamplerIntegrator::Render calls it
// (core/Integrator.cpp:307-310), plus set_uc(i, j, 3, 255): running mean over `frame_count` (= curRenderCount) Render() calls, tone map
// 1 - expf(-x / (1 - 0.75)), implicit float -> unsigned char conversion.  `mean` is the fbuffer plane (RGBA, updated in place).
int gnxo_framebuffer_update(float *mean, const float *frame, int32_t width, int32_t height, int32_t frame_count, uint8_t *rgba8) {
    if (!mean || !frame || !rgba8 || width <= 0 || height <= 0 || frame_count <= 0) return -1;
    const int curRenderCount = frame_count, channals = 4;
    for (int h = 0; h < height; ++h)
        for (int w = 0; w < width; ++w) {
            for (int shifting = 0; shifting < 3; ++shifting) {
                const float dat = frame[(w + h * width) * channals + shifting];
                int offset = (w + h * width) * channals + shifting;
                float weight = (1.0f / (float)curRenderCount);
                float fValue = weight * dat + (1.0f - weight) * mean[offset];
                mean[offset] = fValue;
                float exposure = 0.75;
                float temp_c = 1.0f - expf(-mean[offset] * 1.0f / (1 - exposure));
                rgba8[offset] = (unsigned char)(temp_c * 255);
            }
            rgba8[(w + h * width) * channals + 3] = 255;
        }
    return 0;
}

}  // extern "C"

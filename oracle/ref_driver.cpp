// ORACLE -- TEST INFRASTRUCTURE ONLY; built and run in the development container only.
//
// ref_driver.cpp: links the reference's OWN translation units (compiled by oracle/Makefile from the
// sources where they lie under /root/reference, no stand-in headers) and answers probe queries that
// pin the CPU restatement (oracle/libgnx_oracle.so): RNG, permutations, Halton values, camera rays,
// BVH layout, closest/any hits, BSDF f/Pdf/Sample_f per material, light Sample_Li/Pdf_Li/Le.
//
// NOT available from the reference here: core/Integrator.cpp, core/LightDistribution.cpp and
// integrators/*.cpp (they include ui/FrameBuffer.h -> <QObject>; Qt is absent and no stand-in is
// written).  The `render` command therefore runs a restated Render/Li/EstimateDirect loop ON TOP OF
// the real reference classes (pbr::Scene, pbr::BVHAccel, pbr::Triangle, pbr::BSDF, pbr::Light,
// pbr::HaltonSampler, pbr::PerspectiveCamera); its ray counts are compared with the counts the
// survey recorded from the complete reference.
//
// usage: gnx_ref <scene.bin|-> <cmd> <in.bin|-> <out.bin> [args...]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <vector>

#define private public
#define protected public
#include "accelerator/BVHAccel.h"
#undef private
#undef protected
#include "camera/Perspective.h"
#include "camera/Orthographic.h"
#include "core/Interaction.h"
#include "core/Light.h"
#include "core/Reflection.h"
#include "core/Sampling.h"
#include "core/Scene.h"
#include "lights/DiffuseAreaLight.h"
#include "lights/InfiniteAreaLight.h"
#include "lights/SkyBoxLight.h"
#include "materials/DisneyMaterial.h"
#include "materials/GlassMaterial.h"
#include "materials/MatteMaterial.h"
#include "materials/MetalMaterial.h"
#include "materials/MirrorMaterial.h"
#include "materials/PlasticMaterial.h"
#include "media/GridDensityMedium.h"
#include "media/HomogeneousMedium.h"
#include "samplers/HaltonSampler.h"
#include "samplers/LowDiscrepancy.h"
#include "shape/Triangle.h"
#include "textures/ConstantTexture.h"
#include "textures/ImageTexture.h"
#include "lights/PointLight.h"
#include "lights/SpotLight.h"
#include "lights/DistantLight.h"

#include "include/gnxr.h"

#include <omp.h>

using namespace pbr;

extern "C" float *stbi_loadf(char const *filename, int *x, int *y, int *comp, int req_comp);

namespace {

struct RefLinearBVHNode {  // accelerator/BVHAccel.cpp:54-65 (defined in the .cpp, layout restated)
    Bounds3f bounds;
    int offset;
    uint16_t nPrimitives;
    uint8_t axis;
    uint8_t pad[1];
};

// counting aggregate (what the survey used to count rays)
struct CountingAggregate : public Aggregate {
    std::shared_ptr<Primitive> inner;
    mutable std::atomic<uint64_t> nI{0}, nP{0};
    CountingAggregate(std::shared_ptr<Primitive> p) : inner(p) {}
    Bounds3f WorldBound() const { return inner->WorldBound(); }
    bool Intersect(const Ray &r, SurfaceInteraction *si) const { nI.fetch_add(1, std::memory_order_relaxed); return inner->Intersect(r, si); }
    bool IntersectP(const Ray &r) const { nP.fetch_add(1, std::memory_order_relaxed); return inner->IntersectP(r); }
};

struct SceneFile {
    int32_t nv, nt, nm, nl, nmed, envw, envh, camMedium;
    gnxr_camera cam;
    std::vector<float> verts;
    std::vector<int32_t> idx, triMat, triLight, medIn, medOut;
    std::vector<gnxr_material> mats;
    std::vector<gnxr_light> lights;
    std::vector<gnxr_medium> media;
    std::vector<float> density, env;
    std::string hdrPath;
    std::vector<gnxr_texture> textures;        // file version 2
    std::vector<std::string> texturePaths;     // the image file each ImageTexture loads (the texels themselves stay behind)
    std::vector<float> triUV;                  // file version 3: empty or 6 floats per triangle
    std::vector<float> triN;                   // file version 4: empty or 9 floats per triangle (zeros == no normals)
    std::vector<float> triS;                   // file version 5: the same for tangents
    int32_t splitMethod = 0;                   // file version 6: BVHAccel::SplitMethod
};

bool readScene(const char *path, SceneFile *s) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char magic[4];
    int32_t ver;
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "GNXS", 4) || fread(&ver, 4, 1, f) != 1) return false;
    int32_t hdr[8];
    if (fread(hdr, 4, 8, f) != 8) return false;
    s->nv = hdr[0]; s->nt = hdr[1]; s->nm = hdr[2]; s->nl = hdr[3]; s->nmed = hdr[4]; s->envw = hdr[5]; s->envh = hdr[6]; s->camMedium = hdr[7];
    if (fread(&s->cam, sizeof(gnxr_camera), 1, f) != 1) return false;
    auto rd = [&](auto &v, size_t n) { v.resize(n); return n == 0 || fread(v.data(), sizeof(v[0]), n, f) == n; };
    bool ok = rd(s->verts, 3 * (size_t)s->nv) && rd(s->idx, 3 * (size_t)s->nt) && rd(s->triMat, s->nt) && rd(s->triLight, s->nt) &&
              rd(s->medIn, s->nt) && rd(s->medOut, s->nt) && rd(s->mats, s->nm) && rd(s->lights, s->nl) && rd(s->media, s->nmed);
    int64_t nd = 0;
    ok = ok && fread(&nd, 8, 1, f) == 1 && rd(s->density, (size_t)nd) && rd(s->env, 3 * (size_t)s->envw * s->envh);
    int32_t plen = 0;
    if (ok && fread(&plen, 4, 1, f) == 1 && plen > 0) { s->hdrPath.resize(plen); ok = fread(&s->hdrPath[0], 1, plen, f) == (size_t)plen; }
    if (ok && ver >= 2) {
        int32_t ntex = 0;
        ok = fread(&ntex, 4, 1, f) == 1 && rd(s->textures, (size_t)ntex);
        for (int i = 0; ok && i < ntex; ++i) {
            std::string tp;
            ok = fread(&plen, 4, 1, f) == 1;
            if (ok && plen > 0) { tp.resize(plen); ok = fread(&tp[0], 1, plen, f) == (size_t)plen; }
            s->texturePaths.push_back(tp);
        }
    }
    if (ok && ver >= 3) {
        int32_t hasUV = 0;
        ok = fread(&hasUV, 4, 1, f) == 1 && (!hasUV || rd(s->triUV, 6 * (size_t)s->nt));
    }
    if (ok && ver >= 4) {
        int32_t hasN = 0;
        ok = fread(&hasN, 4, 1, f) == 1 && (!hasN || rd(s->triN, 9 * (size_t)s->nt));
    }
    if (ok && ver >= 5) {
        int32_t hasS = 0;
        ok = fread(&hasS, 4, 1, f) == 1 && (!hasS || rd(s->triS, 9 * (size_t)s->nt));
    }
    if (ok && ver >= 6) ok = fread(&s->splitMethod, 4, 1, f) == 1;
    fclose(f);
    return ok;
}

template <typename T> std::shared_ptr<Texture<T>> C(const T &v) { return std::make_shared<ConstantTexture<T>>(v); }
Spectrum S3(const float *p) { Spectrum s; s[0] = p[0]; s[1] = p[1]; s[2] = p[2]; return s; }

// the reference's own ImageTexture / UVMapping2D / MIPMap, loading the file through its stb_image path
std::shared_ptr<Texture<Spectrum>> imageOrConstant(const SceneFile &sf, int texture, const float *constant) {
    if (texture <= 0) return C(S3(constant));
    const gnxr_texture &t = sf.textures[texture - 1];
    std::unique_ptr<TextureMapping2D> map = std::make_unique<UVMapping2D>(t.su, t.sv, t.du, t.dv);
    ImageWrap wrap = t.wrap == GNXR_WRAP_REPEAT ? ImageWrap::Repeat : (t.wrap == GNXR_WRAP_BLACK ? ImageWrap::Black : ImageWrap::Clamp);
    return std::make_shared<ImageTexture<RGBSpectrum, Spectrum>>(std::move(map), sf.texturePaths[texture - 1], t.trilinear != 0, t.max_aniso, wrap, t.scale, t.gamma != 0);
}

std::shared_ptr<Material> makeMaterial(const SceneFile &sf, const gnxr_material &m) {
    std::shared_ptr<Texture<Float>> bump = m.has_bump ? C<Float>(0.0f) : nullptr;
    switch (m.type) {
    case GNXR_MAT_MATTE: return std::make_shared<MatteMaterial>(imageOrConstant(sf, m.kd_texture, m.kd), C<Float>(m.sigma), bump);
    case GNXR_MAT_MIRROR: return std::make_shared<MirrorMaterial>(C(S3(m.kr)), bump);
    case GNXR_MAT_GLASS:
        return std::make_shared<GlassMaterial>(C(S3(m.kr)), C(S3(m.kt)), C<Float>(m.urough), C<Float>(m.vrough), C<Float>(m.eta[0]), bump, m.remap_roughness != 0);
    case GNXR_MAT_METAL:
        return std::make_shared<MetalMaterial>(C(S3(m.eta)), C(S3(m.k)), C<Float>(m.urough), C<Float>(m.urough), C<Float>(m.vrough), bump, m.remap_roughness != 0);
    case GNXR_MAT_PLASTIC:
        return std::make_shared<PlasticMaterial>(imageOrConstant(sf, m.kd_texture, m.kd), imageOrConstant(sf, m.ks_texture, m.ks), C<Float>(m.urough), bump, m.remap_roughness != 0);
    case GNXR_MAT_DISNEY:
        return std::make_shared<DisneyMaterial>(C(S3(m.kd)), C<Float>(m.disney_metallic), C<Float>(m.eta[0]), C<Float>(m.disney_roughness),
                                                C<Float>(m.disney_spec_tint), C<Float>(m.disney_anisotropic), C<Float>(m.disney_sheen),
                                                C<Float>(m.disney_sheen_tint), C<Float>(m.disney_clearcoat), C<Float>(m.disney_clearcoat_gloss),
                                                C<Float>(m.disney_spec_trans), C(S3(m.disney_scatter_distance)), m.disney_thin != 0,
                                                C<Float>(m.disney_flatness), C<Float>(m.disney_diff_trans), bump);
    default: return nullptr;
    }
}

struct RefScene {
    SceneFile sf;
    Transform identity, identityInv;
    std::vector<std::shared_ptr<TriangleMesh>> meshes;
    std::vector<std::shared_ptr<Shape>> shapes;
    std::vector<std::shared_ptr<Material>> materials;
    std::vector<std::shared_ptr<Medium>> media;
    std::vector<std::shared_ptr<Light>> lights;
    std::vector<std::shared_ptr<Primitive>> prims;
    std::unordered_map<const Primitive *, int> primIndex;
    std::shared_ptr<BVHAccel> bvh;
    std::shared_ptr<CountingAggregate> counting;
    std::unique_ptr<Scene> scene;

    void build() {
        for (auto &m : sf.mats) materials.push_back(makeMaterial(sf, m));
        for (auto &m : sf.media) {
            if (m.type == GNXR_MEDIUM_HOMOGENEOUS) media.push_back(std::make_shared<HomogeneousMedium>(S3(m.sigma_a), S3(m.sigma_s), m.g));
            else {
                Matrix4x4 mm;
                memcpy(mm.m, m.medium_to_world, 64);
                media.push_back(std::make_shared<GridDensityMedium>(S3(m.sigma_a), S3(m.sigma_s), m.g, m.nx, m.ny, m.nz, Transform(mm), sf.density.data() + m.density_offset));
            }
        }
        // one single-triangle mesh per triangle, identity transform: world-space vertices pass through
        // TriangleMesh's ObjectToWorld(P[i]) unchanged (1*x + 0*y + 0*z + 0, wp == 1)
        lights.resize(sf.nl);
        for (int t = 0; t < sf.nt; ++t) {
            Point3f P[3];
            int vi[3] = {0, 1, 2};
            for (int k = 0; k < 3; ++k) {
                int v = sf.idx[3 * t + k];
                P[k] = Point3f(sf.verts[3 * v], sf.verts[3 * v + 1], sf.verts[3 * v + 2]);
            }
            Point2f UV[3];
            if (!sf.triUV.empty()) for (int k = 0; k < 3; ++k) UV[k] = Point2f(sf.triUV[6 * (size_t)t + 2 * k], sf.triUV[6 * (size_t)t + 2 * k + 1]);
            Normal3f N[3];
            bool hasN = false;
            if (!sf.triN.empty()) for (int k = 0; k < 3; ++k) {
                N[k] = Normal3f(sf.triN[9 * (size_t)t + 3 * k], sf.triN[9 * (size_t)t + 3 * k + 1], sf.triN[9 * (size_t)t + 3 * k + 2]);
                hasN = hasN || N[k].x != 0 || N[k].y != 0 || N[k].z != 0;
            }
            Vector3f S[3];
            bool hasS = false;
            if (!sf.triS.empty()) for (int k = 0; k < 3; ++k) {
                S[k] = Vector3f(sf.triS[9 * (size_t)t + 3 * k], sf.triS[9 * (size_t)t + 3 * k + 1], sf.triS[9 * (size_t)t + 3 * k + 2]);
                hasS = hasS || S[k].x != 0 || S[k].y != 0 || S[k].z != 0;
            }
            auto mesh = std::make_shared<TriangleMesh>(identity, 1, vi, 3, P, hasS ? S : nullptr, hasN ? N : nullptr, sf.triUV.empty() ? nullptr : UV, nullptr);
            meshes.push_back(mesh);
            auto tri = std::make_shared<Triangle>(&identity, &identityInv, false, mesh, 0);
            shapes.push_back(tri);
            std::shared_ptr<AreaLight> area;
            int li = sf.triLight[t];
            if (li >= 0) {
                const gnxr_light &l = sf.lights[li];
                area = std::make_shared<DiffuseAreaLight>(identity, MediumInterface(), S3(l.le), l.n_samples, tri, l.two_sided != 0);
                lights[li] = area;
            }
            std::shared_ptr<Material> mat = sf.triMat[t] >= 0 ? materials[sf.triMat[t]] : nullptr;
            MediumInterface mif(sf.medIn[t] >= 0 ? media[sf.medIn[t]].get() : nullptr, sf.medOut[t] >= 0 ? media[sf.medOut[t]].get() : nullptr);
            auto prim = std::make_shared<GeometricPrimitive>(tri, mat, area, mif);
            primIndex[prim.get()] = t;
            prims.push_back(prim);
        }
        for (int i = 0; i < sf.nl; ++i) {
            const gnxr_light &l = sf.lights[i];
            if (l.type == GNXR_LIGHT_INFINITE) {
                Matrix4x4 m;
                memcpy(m.m, l.light_to_world, 64);
                lights[i] = std::make_shared<InfiniteAreaLight>(Transform(m), S3(l.le), l.n_samples, sf.hdrPath);
            } else if (l.type == GNXR_LIGHT_SKYBOX) {
                lights[i] = std::make_shared<SkyBoxLight>(Transform(), Point3f(l.center[0], l.center[1], l.center[2]), l.radius, "1", l.n_samples);
            } else if (l.type == GNXR_LIGHT_POINT || l.type == GNXR_LIGHT_SPOT || l.type == GNXR_LIGHT_DISTANT) {
                Matrix4x4 m;
                memcpy(m.m, l.light_to_world, 64);
                Transform l2w(m);   // Transform(const Matrix4x4 &) computes the inverse itself
                if (l.type == GNXR_LIGHT_POINT) lights[i] = std::make_shared<PointLight>(l2w, MediumInterface(), S3(l.le));
                else if (l.type == GNXR_LIGHT_SPOT) lights[i] = std::make_shared<SpotLight>(l2w, MediumInterface(), S3(l.le), l.radius, l.falloff_start);
                else lights[i] = std::make_shared<DistantLight>(l2w, S3(l.le), Vector3f(l.center[0], l.center[1], l.center[2]));
            }
        }
        bvh = std::make_shared<BVHAccel>(prims, 1, sf.splitMethod == 1 ? BVHAccel::SplitMethod::HLBVH : sf.splitMethod == 2 ? BVHAccel::SplitMethod::Middle
                                                  : sf.splitMethod == 3 ? BVHAccel::SplitMethod::EqualCounts : BVHAccel::SplitMethod::SAH);
        counting = std::make_shared<CountingAggregate>(bvh);
        scene.reset(new Scene(counting, lights));
    }
};

std::vector<char> readAll(const char *path) {
    std::vector<char> v;
    if (!strcmp(path, "-")) return v;
    FILE *f = fopen(path, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n);
    if (n && fread(v.data(), 1, n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}
void writeAll(const char *path, const void *p, size_t n) {
    FILE *f = fopen(path, "wb");
    fwrite(p, 1, n, f);
    fclose(f);
}

// ---- restated SpatialLightDistribution (core/LightDistribution.cpp:70-274) over pbr::Light ----
struct RefLightDistribution {
    const Scene &scene;
    int strategy;
    int nVoxels[3];
    std::unique_ptr<Distribution1D> fixed;
    std::vector<std::unique_ptr<Distribution1D>> vox;
    std::vector<std::atomic<int>> state;
    RefLightDistribution(const Scene &s, int strat) : scene(s), strategy(strat) {
        size_t nl = s.lights.size();
        if (nl == 0) return;
        if (strat == GNXR_LIGHTS_UNIFORM || nl == 1) {
            strategy = GNXR_LIGHTS_UNIFORM;
            std::vector<Float> prob(nl, Float(1));
            fixed.reset(new Distribution1D(&prob[0], int(nl)));
        } else if (strat == GNXR_LIGHTS_POWER) {
            std::vector<Float> power;
            for (const auto &l : s.lights) power.push_back(l->Power().y());
            fixed.reset(new Distribution1D(&power[0], int(nl)));
        } else {
            Bounds3f b = s.WorldBound();
            Vector3f diag = b.Diagonal();
            Float bmax = diag[b.MaximumExtent()];
            for (int i = 0; i < 3; ++i) nVoxels[i] = std::max(1, int(std::round(diag[i] / bmax * 64)));
            size_t nv = (size_t)nVoxels[0] * nVoxels[1] * nVoxels[2];
            vox.resize(nv);
            state = std::vector<std::atomic<int>>(nv);
            for (auto &a : state) a.store(0);
        }
    }
    Distribution1D *compute(Point3i pi) const {
        Point3f p0(Float(pi[0]) / Float(nVoxels[0]), Float(pi[1]) / Float(nVoxels[1]), Float(pi[2]) / Float(nVoxels[2]));
        Point3f p1(Float(pi[0] + 1) / Float(nVoxels[0]), Float(pi[1] + 1) / Float(nVoxels[1]), Float(pi[2] + 1) / Float(nVoxels[2]));
        Bounds3f voxelBounds(scene.WorldBound().Lerp(p0), scene.WorldBound().Lerp(p1));
        int nSamples = 128;
        std::vector<Float> lightContrib(scene.lights.size(), Float(0));
        for (int i = 0; i < nSamples; ++i) {
            Point3f po = voxelBounds.Lerp(Point3f(RadicalInverse(0, i), RadicalInverse(1, i), RadicalInverse(2, i)));
            Interaction intr(po, Normal3f(), Vector3f(), Vector3f(1, 0, 0), 0, MediumInterface());
            Point2f u(RadicalInverse(3, i), RadicalInverse(4, i));
            for (size_t j = 0; j < scene.lights.size(); ++j) {
                Float pdf;
                Vector3f wi;
                VisibilityTester vis;
                Spectrum Li = scene.lights[j]->Sample_Li(intr, u, &wi, &pdf, &vis);
                if (pdf > 0) lightContrib[j] += Li.y() / pdf;
            }
        }
        Float sumContrib = std::accumulate(lightContrib.begin(), lightContrib.end(), Float(0));
        Float avgContrib = sumContrib / (nSamples * lightContrib.size());
        Float minContrib = (avgContrib > 0) ? .001 * avgContrib : 1;
        for (size_t i = 0; i < lightContrib.size(); ++i) lightContrib[i] = std::max(lightContrib[i], minContrib);
        return new Distribution1D(&lightContrib[0], int(lightContrib.size()));
    }
    const Distribution1D *Lookup(const Point3f &p) {
        if (strategy != GNXR_LIGHTS_SPATIAL) return fixed.get();
        Vector3f offset = scene.WorldBound().Offset(p);
        Point3i pi;
        for (int i = 0; i < 3; ++i) pi[i] = Clamp(int(offset[i] * nVoxels[i]), 0, nVoxels[i] - 1);
        size_t idx = ((size_t)pi[0] * nVoxels[1] + pi[1]) * nVoxels[2] + pi[2];
        if (state[idx].load(std::memory_order_acquire) == 2) return vox[idx].get();
        int expected = 0;
        if (state[idx].compare_exchange_strong(expected, 1)) {
            vox[idx].reset(compute(pi));
            state[idx].store(2, std::memory_order_release);
        } else
            while (state[idx].load(std::memory_order_acquire) != 2) {}
        return vox[idx].get();
    }
};

// ---- restated EstimateDirect / UniformSampleOneLight / PathIntegrator::Li over pbr classes ----
Spectrum refEstimateDirect(const SurfaceInteraction &isect, const Point2f &uScattering, const Light &light, const Point2f &uLight,
                           const Scene &scene) {
    BxDFType bsdfFlags = BxDFType(BSDF_ALL & ~BSDF_SPECULAR);
    Spectrum Ld(0.f);
    Vector3f wi;
    Float lightPdf = 0, scatteringPdf = 0;
    VisibilityTester visibility;
    Spectrum Li = light.Sample_Li(isect, uLight, &wi, &lightPdf, &visibility);
    if (lightPdf > 0 && !Li.IsBlack()) {
        Spectrum f = isect.bsdf->f(isect.wo, wi, bsdfFlags) * AbsDot(wi, isect.shading.n);
        scatteringPdf = isect.bsdf->Pdf(isect.wo, wi, bsdfFlags);
        if (!f.IsBlack()) {
            if (!visibility.Unoccluded(scene)) Li = Spectrum(0.f);
            if (!Li.IsBlack()) {
                if (IsDeltaLight(light.flags)) Ld += f * Li / lightPdf;
                else {
                    Float weight = PowerHeuristic(1, lightPdf, 1, scatteringPdf);
                    Ld += f * Li * weight / lightPdf;
                }
            }
        }
    }
    if (!IsDeltaLight(light.flags)) {
        BxDFType sampledType;
        Spectrum f = isect.bsdf->Sample_f(isect.wo, &wi, uScattering, &scatteringPdf, bsdfFlags, &sampledType);
        f *= AbsDot(wi, isect.shading.n);
        bool sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
        if (!f.IsBlack() && scatteringPdf > 0) {
            Float weight = 1;
            if (!sampledSpecular) {
                lightPdf = light.Pdf_Li(isect, wi);
                if (lightPdf == 0) return Ld;
                weight = PowerHeuristic(1, scatteringPdf, 1, lightPdf);
            }
            SurfaceInteraction lightIsect;
            Ray ray = isect.SpawnRay(wi);
            bool found = scene.Intersect(ray, &lightIsect);
            Spectrum Li2(0.f);
            if (found) {
                if (lightIsect.primitive->GetAreaLight() == &light) Li2 = lightIsect.Le(-wi);
            } else
                Li2 = light.Le(ray);
            if (!Li2.IsBlack()) Ld += f * Li2 * Spectrum(1.f) * weight / scatteringPdf;
        }
    }
    return Ld;
}

Spectrum refPathLi(const RayDifferential &r, const Scene &scene, Sampler &sampler, MemoryArena &arena, RefLightDistribution &ld, int maxDepth,
                   Float rrThreshold) {
    Spectrum L(0.f), beta(1.f);
    Ray ray(r);
    bool specularBounce = false;
    int bounces;
    Float etaScale = 1;
    for (bounces = 0;; ++bounces) {
        SurfaceInteraction isect;
        bool foundIntersection = scene.Intersect(ray, &isect);
        if (bounces == 0 || specularBounce) {
            if (foundIntersection) L += beta * isect.Le(-ray.d);
            else for (const auto &light : scene.infiniteLights) L += beta * light->Le(ray);
        }
        if (!foundIntersection || bounces >= maxDepth) break;
        isect.ComputeScatteringFunctions(ray, arena, true);
        if (!isect.bsdf) { ray = isect.SpawnRay(ray.d); bounces--; continue; }
        const Distribution1D *distrib = ld.Lookup(isect.p);
        if (isect.bsdf->NumComponents(BxDFType(BSDF_ALL & ~BSDF_SPECULAR)) > 0) {
            // UniformSampleOneLight, core/Integrator.cpp:57-79
            Spectrum Ld(0.f);
            int nLights = int(scene.lights.size());
            if (nLights > 0) {
                Float lightPdf;
                int lightNum = distrib->SampleDiscrete(sampler.Get1D(), &lightPdf);
                if (lightPdf != 0) {
                    const std::shared_ptr<Light> &light = scene.lights[lightNum];
                    Point2f uLight = sampler.Get2D();
                    Point2f uScattering = sampler.Get2D();
                    Ld = refEstimateDirect(isect, uScattering, *light, uLight, scene) / lightPdf;
                }
            }
            L += beta * Ld;
        }
        Vector3f wo = -ray.d, wi;
        Float pdf;
        BxDFType flags;
        Spectrum f = isect.bsdf->Sample_f(wo, &wi, sampler.Get2D(), &pdf, BSDF_ALL, &flags);
        if (f.IsBlack() || pdf == 0.f) break;
        beta *= f * AbsDot(wi, isect.shading.n) / pdf;
        specularBounce = (flags & BSDF_SPECULAR) != 0;
        if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
            Float eta = isect.bsdf->eta;
            etaScale *= (Dot(wo, isect.n) > 0) ? (eta * eta) : 1 / (eta * eta);
        }
        ray = isect.SpawnRay(wi);
        Spectrum rrBeta = beta * etaScale;
        if (rrBeta.MaxComponentValue() < rrThreshold && bounces > 3) {
            Float q = std::max((Float).05, 1 - rrBeta.MaxComponentValue());
            if (sampler.Get1D() < q) break;
            beta /= 1 - q;
        }
    }
    return L;
}

// ---- restated EstimateDirect (handleMedia = true) / VolPathIntegrator::Li over pbr classes ----
Spectrum refEstimateDirectMedia(const Interaction &it, const Point2f &uScattering, const Light &light, const Point2f &uLight, const Scene &scene,
                                Sampler &sampler) {
    BxDFType bsdfFlags = BxDFType(BSDF_ALL & ~BSDF_SPECULAR);
    Spectrum Ld(0.f);
    Vector3f wi;
    Float lightPdf = 0, scatteringPdf = 0;
    VisibilityTester visibility;
    Spectrum Li = light.Sample_Li(it, uLight, &wi, &lightPdf, &visibility);
    if (lightPdf > 0 && !Li.IsBlack()) {
        Spectrum f;
        if (it.IsSurfaceInteraction()) {
            const SurfaceInteraction &isect = (const SurfaceInteraction &)it;
            f = isect.bsdf->f(isect.wo, wi, bsdfFlags) * AbsDot(wi, isect.shading.n);
            scatteringPdf = isect.bsdf->Pdf(isect.wo, wi, bsdfFlags);
        } else {
            const MediumInteraction &mi = (const MediumInteraction &)it;
            Float p = mi.phase->p(mi.wo, wi);
            f = Spectrum(p);
            scatteringPdf = p;
        }
        if (!f.IsBlack()) {
            Li *= visibility.Tr(scene, sampler);
            if (!Li.IsBlack()) {
                if (IsDeltaLight(light.flags)) Ld += f * Li / lightPdf;
                else {
                    Float weight = PowerHeuristic(1, lightPdf, 1, scatteringPdf);
                    Ld += f * Li * weight / lightPdf;
                }
            }
        }
    }
    if (!IsDeltaLight(light.flags)) {
        Spectrum f;
        bool sampledSpecular = false;
        if (it.IsSurfaceInteraction()) {
            BxDFType sampledType;
            const SurfaceInteraction &isect = (const SurfaceInteraction &)it;
            f = isect.bsdf->Sample_f(isect.wo, &wi, uScattering, &scatteringPdf, bsdfFlags, &sampledType);
            f *= AbsDot(wi, isect.shading.n);
            sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
        } else {
            const MediumInteraction &mi = (const MediumInteraction &)it;
            Float p = mi.phase->Sample_p(mi.wo, &wi, uScattering);
            f = Spectrum(p);
            scatteringPdf = p;
        }
        if (!f.IsBlack() && scatteringPdf > 0) {
            Float weight = 1;
            if (!sampledSpecular) {
                lightPdf = light.Pdf_Li(it, wi);
                if (lightPdf == 0) return Ld;
                weight = PowerHeuristic(1, scatteringPdf, 1, lightPdf);
            }
            SurfaceInteraction lightIsect;
            Ray ray = it.SpawnRay(wi);
            Spectrum Tr(1.f);
            bool found = scene.IntersectTr(ray, sampler, &lightIsect, &Tr);
            Spectrum Li2(0.f);
            if (found) {
                if (lightIsect.primitive->GetAreaLight() == &light) Li2 = lightIsect.Le(-wi);
            } else
                Li2 = light.Le(ray);
            if (!Li2.IsBlack()) Ld += f * Li2 * Tr * weight / scatteringPdf;
        }
    }
    return Ld;
}

Spectrum refOneLightMedia(const Interaction &it, const Scene &scene, Sampler &sampler, const Distribution1D *distrib) {
    int nLights = int(scene.lights.size());
    if (nLights == 0) return Spectrum(0.f);
    Float lightPdf;
    int lightNum = distrib->SampleDiscrete(sampler.Get1D(), &lightPdf);
    if (lightPdf == 0) return Spectrum(0.f);
    const std::shared_ptr<Light> &light = scene.lights[lightNum];
    Point2f uLight = sampler.Get2D();
    Point2f uScattering = sampler.Get2D();
    return refEstimateDirectMedia(it, uScattering, *light, uLight, scene, sampler) / lightPdf;
}

Spectrum refVolPathLi(const RayDifferential &r, const Scene &scene, Sampler &sampler, MemoryArena &arena, RefLightDistribution &ld, int maxDepth, Float rrThreshold) {
    Spectrum L(0.f), beta(1.f);
    RayDifferential ray(r);
    bool specularBounce = false;
    int bounces;
    Float etaScale = 1;
    for (bounces = 0;; ++bounces) {
        SurfaceInteraction isect;
        bool foundIntersection = scene.Intersect(ray, &isect);
        MediumInteraction mi;
        if (ray.medium) beta *= ray.medium->Sample(ray, sampler, arena, &mi);
        if (beta.IsBlack()) break;
        if (mi.IsValid()) {
            if (bounces >= maxDepth) break;
            const Distribution1D *distrib = ld.Lookup(mi.p);
            L += beta * refOneLightMedia(mi, scene, sampler, distrib);
            Vector3f wo = -ray.d, wi;
            mi.phase->Sample_p(wo, &wi, sampler.Get2D());
            ray = mi.SpawnRay(wi);
            specularBounce = false;
        } else {
            if (bounces == 0 || specularBounce) {
                if (foundIntersection) L += beta * isect.Le(-ray.d);
                else for (const auto &light : scene.infiniteLights) L += beta * light->Le(ray);
            }
            if (!foundIntersection || bounces >= maxDepth) break;
            isect.ComputeScatteringFunctions(ray, arena, true);
            if (!isect.bsdf) { ray = isect.SpawnRay(ray.d); bounces--; continue; }
            const Distribution1D *distrib = ld.Lookup(isect.p);
            L += beta * refOneLightMedia(isect, scene, sampler, distrib);
            Vector3f wo = -ray.d, wi;
            Float pdf;
            BxDFType flags;
            Spectrum f = isect.bsdf->Sample_f(wo, &wi, sampler.Get2D(), &pdf, BSDF_ALL, &flags);
            if (f.IsBlack() || pdf == 0.f) break;
            beta *= f * AbsDot(wi, isect.shading.n) / pdf;
            specularBounce = (flags & BSDF_SPECULAR) != 0;
            if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                Float eta = isect.bsdf->eta;
                etaScale *= (Dot(wo, isect.n) > 0) ? (eta * eta) : 1 / (eta * eta);
            }
            ray = isect.SpawnRay(wi);
        }
        Spectrum rrBeta = beta * etaScale;
        if (rrBeta.MaxComponentValue() < rrThreshold && bounces > 3) {
            Float q = std::max((Float).05, 1 - rrBeta.MaxComponentValue());
            if (sampler.Get1D() < q) break;
            beta /= 1 - q;
        }
    }
    return L;
}

// Ray differentials of the specular children, SamplerIntegrator::SpecularReflect / SpecularTransmit (core/Integrator.cpp:335-354,
// 376-436), on the reference's classes.  They reach the radiance only through image-texture filtering.
void refReflectDifferentials(const RayDifferential &ray, const SurfaceInteraction &isect, const Vector3f &wo, const Vector3f &wi, RayDifferential *rd) {
    if (!ray.hasDifferentials) return;
    const Normal3f &ns = isect.shading.n;
    rd->hasDifferentials = true;
    rd->rxOrigin = isect.p + isect.dpdx;
    rd->ryOrigin = isect.p + isect.dpdy;
    Normal3f dndx = isect.shading.dndu * isect.dudx + isect.shading.dndv * isect.dvdx;
    Normal3f dndy = isect.shading.dndu * isect.dudy + isect.shading.dndv * isect.dvdy;
    Vector3f dwodx = -ray.rxDirection - wo, dwody = -ray.ryDirection - wo;
    Float dDNdx = Dot(dwodx, ns) + Dot(wo, dndx);
    Float dDNdy = Dot(dwody, ns) + Dot(wo, dndy);
    rd->rxDirection = wi - dwodx + 2.f * Vector3f(Dot(wo, ns) * dndx + dDNdx * ns);
    rd->ryDirection = wi - dwody + 2.f * Vector3f(Dot(wo, ns) * dndy + dDNdy * ns);
}
void refTransmitDifferentials(const RayDifferential &ray, const SurfaceInteraction &isect, const Vector3f &wo, const Vector3f &wi, RayDifferential *rd) {
    if (!ray.hasDifferentials) return;
    Normal3f ns = isect.shading.n;
    rd->hasDifferentials = true;
    rd->rxOrigin = isect.p + isect.dpdx;
    rd->ryOrigin = isect.p + isect.dpdy;
    Normal3f dndx = isect.shading.dndu * isect.dudx + isect.shading.dndv * isect.dvdx;
    Normal3f dndy = isect.shading.dndu * isect.dudy + isect.shading.dndv * isect.dvdy;
    Float eta = 1 / isect.bsdf->eta;
    if (Dot(wo, ns) < 0) {
        eta = 1 / eta;
        ns = -ns;
        dndx = -dndx;
        dndy = -dndy;
    }
    Vector3f dwodx = -ray.rxDirection - wo, dwody = -ray.ryDirection - wo;
    Float dDNdx = Dot(dwodx, ns) + Dot(wo, dndx);
    Float dDNdy = Dot(dwody, ns) + Dot(wo, dndy);
    Float mu = eta * Dot(wo, ns) - AbsDot(wi, ns);
    Float dmudx = (eta - (eta * eta * Dot(wo, ns)) / AbsDot(wi, ns)) * dDNdx;
    Float dmudy = (eta - (eta * eta * Dot(wo, ns)) / AbsDot(wi, ns)) * dDNdy;
    rd->rxDirection = wi - eta * dwodx + Vector3f(mu * dndx + dmudx * ns);
    rd->ryDirection = wi - eta * dwody + Vector3f(mu * dndy + dmudy * ns);
}

// WhittedIntegrator::Li + SamplerIntegrator::SpecularReflect / SpecularTransmit (integrators/WhittedIntegrator.cpp:14-68,
// core/Integrator.cpp:321-442) restated on the reference's classes.
Spectrum refWhittedLi(const RayDifferential &ray, const Scene &scene, Sampler &sampler, MemoryArena &arena, int maxDepth, int depth) {
    Spectrum L(0.);
    SurfaceInteraction isect;
    if (!scene.Intersect(ray, &isect)) {
        for (const auto &light : scene.lights) L += light->Le(ray);
        return L;
    }
    const Normal3f &n = isect.shading.n;
    Vector3f wo = isect.wo;
    isect.ComputeScatteringFunctions(ray, arena);
    if (!isect.bsdf) return refWhittedLi(isect.SpawnRay(ray.d), scene, sampler, arena, maxDepth, depth);
    L += isect.Le(wo);
    Spectrum lightL(0.0);
    for (const auto &light : scene.lights) {
        Vector3f wi;
        Float pdf;
        VisibilityTester visibility;
        Spectrum Li = light->Sample_Li(isect, sampler.Get2D(), &wi, &pdf, &visibility);
        if (Li.IsBlack() || pdf == 0) continue;
        Spectrum f = isect.bsdf->f(wo, wi);
        if (!f.IsBlack() && visibility.Unoccluded(scene)) lightL += f * Li * AbsDot(wi, n) / pdf;
    }
    L += lightL;
    if (depth + 1 < maxDepth) {
        {
            Vector3f wi;
            Float pdf;
            BxDFType type = BxDFType(BSDF_REFLECTION | BSDF_SPECULAR);
            Spectrum f = isect.bsdf->Sample_f(wo, &wi, sampler.Get2D(), &pdf, type);
            const Normal3f &ns = isect.shading.n;
            if (pdf > 0.f && !f.IsBlack() && AbsDot(wi, ns) != 0.f) {
                RayDifferential rd = isect.SpawnRay(wi);
                refReflectDifferentials(ray, isect, wo, wi, &rd);
                L += f * refWhittedLi(rd, scene, sampler, arena, maxDepth, depth + 1) * AbsDot(wi, ns) / pdf;
            } else L += Spectrum(0.f);
        }
        {
            Vector3f wi;
            Float pdf;
            Spectrum f = isect.bsdf->Sample_f(wo, &wi, sampler.Get2D(), &pdf, BxDFType(BSDF_TRANSMISSION | BSDF_SPECULAR));
            Spectrum Lt = Spectrum(0.f);
            Normal3f ns = isect.shading.n;
            if (pdf > 0.f && !f.IsBlack() && AbsDot(wi, ns) != 0.f) {
                RayDifferential rd = isect.SpawnRay(wi);
                refTransmitDifferentials(ray, isect, wo, wi, &rd);
                Lt = f * refWhittedLi(rd, scene, sampler, arena, maxDepth, depth + 1) * AbsDot(wi, ns) / pdf;
            }
            L += Lt;
        }
    }
    return L;
}

// DirectLightingIntegrator::Li (integrators/DirectLightingIntegrator.cpp:30-64) with UniformSampleAllLights /
// UniformSampleOneLight (core/Integrator.cpp:25-79) restated on the reference's classes; the sample arrays come from the
// reference's own Sampler::Request2DArray / Get2DArray / GlobalSampler::StartPixel.
Spectrum refDirectLi(const RayDifferential &ray, const Scene &scene, Sampler &sampler, MemoryArena &arena, int strategy,
                     const std::vector<int> &nLightSamples, int maxDepth, int depth) {
    Spectrum L(0.f);
    SurfaceInteraction isect;
    if (!scene.Intersect(ray, &isect)) {
        for (const auto &light : scene.lights) L += light->Le(ray);
        return L;
    }
    isect.ComputeScatteringFunctions(ray, arena);
    if (!isect.bsdf) return refDirectLi(isect.SpawnRay(ray.d), scene, sampler, arena, strategy, nLightSamples, maxDepth, depth);
    Vector3f wo = isect.wo;
    L += isect.Le(wo);
    if (scene.lights.size() > 0) {
        if (strategy == 0) {   // L += UniformSampleAllLights(...): the lights are summed in the callee's own L first (core/Integrator.cpp:31-54)
            Spectrum Lall(0.f);
            for (size_t j = 0; j < scene.lights.size(); ++j) {
                const std::shared_ptr<Light> &light = scene.lights[j];
                int nSamples = nLightSamples[j];
                const Point2f *uLightArray = sampler.Get2DArray(nSamples);
                const Point2f *uScatteringArray = sampler.Get2DArray(nSamples);
                if (!uLightArray || !uScatteringArray) {
                    Point2f uLight = sampler.Get2D();
                    Point2f uScattering = sampler.Get2D();
                    Lall += refEstimateDirect(isect, uScattering, *light, uLight, scene);
                } else {
                    Spectrum Ld(0.f);
                    for (int k = 0; k < nSamples; ++k) Ld += refEstimateDirect(isect, uScatteringArray[k], *light, uLightArray[k], scene);
                    Lall += Ld / nSamples;
                }
            }
            L += Lall;
        } else {               // UniformSampleOneLight, lightDistrib == nullptr
            int nLights = int(scene.lights.size());
            int lightNum = std::min((int)(sampler.Get1D() * nLights), nLights - 1);
            Float lightPdf = Float(1) / nLights;
            const std::shared_ptr<Light> &light = scene.lights[lightNum];
            Point2f uLight = sampler.Get2D();
            Point2f uScattering = sampler.Get2D();
            L += refEstimateDirect(isect, uScattering, *light, uLight, scene) / lightPdf;
        }
    }
    if (depth + 1 < maxDepth) {
        {
            Vector3f wi;
            Float pdf;
            Spectrum f = isect.bsdf->Sample_f(wo, &wi, sampler.Get2D(), &pdf, BxDFType(BSDF_REFLECTION | BSDF_SPECULAR));
            const Normal3f &ns = isect.shading.n;
            if (pdf > 0.f && !f.IsBlack() && AbsDot(wi, ns) != 0.f) {
                RayDifferential rd = isect.SpawnRay(wi);
                refReflectDifferentials(ray, isect, wo, wi, &rd);
                L += f * refDirectLi(rd, scene, sampler, arena, strategy, nLightSamples, maxDepth, depth + 1) * AbsDot(wi, ns) / pdf;
            } else L += Spectrum(0.f);
        }
        {
            Vector3f wi;
            Float pdf;
            Spectrum f = isect.bsdf->Sample_f(wo, &wi, sampler.Get2D(), &pdf, BxDFType(BSDF_TRANSMISSION | BSDF_SPECULAR));
            Spectrum Lt = Spectrum(0.f);
            Normal3f ns = isect.shading.n;
            if (pdf > 0.f && !f.IsBlack() && AbsDot(wi, ns) != 0.f) {
                RayDifferential rd = isect.SpawnRay(wi);
                refTransmitDifferentials(ray, isect, wo, wi, &rd);
                Lt = f * refDirectLi(rd, scene, sampler, arena, strategy, nLightSamples, maxDepth, depth + 1) * AbsDot(wi, ns) / pdf;
            }
            L += Lt;
        }
    }
    return L;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: gnx_ref <scene.bin|-> <cmd> <in.bin|-> <out.bin> [args]\n"); return 2; }
    const char *scenePath = argv[1], *cmd = argv[2], *inPath = argv[3], *outPath = argv[4];
    std::vector<char> in = readAll(inPath);

    if (!strcmp(cmd, "rng")) {  // out: 64 u32 default-seeded + 64 u32 SetSequence(7)
        std::vector<uint32_t> out;
        RNG a;
        for (int i = 0; i < 64; ++i) out.push_back(a.UniformUInt32());
        RNG b;
        b.SetSequence(7);
        for (int i = 0; i < 64; ++i) out.push_back(b.UniformUInt32());
        writeAll(outPath, out.data(), out.size() * 4);
        return 0;
    }
    if (!strcmp(cmd, "perm")) {
        RNG rng;
        std::vector<uint16_t> p = ComputeRadicalInversePermutations(rng);
        writeAll(outPath, p.data(), p.size() * 2);
        return 0;
    }
    if (!strcmp(cmd, "primes")) {
        std::vector<int32_t> out(Primes, Primes + PrimeTableSize);
        out.insert(out.end(), PrimeSums, PrimeSums + PrimeTableSize);
        writeAll(outPath, out.data(), out.size() * 4);
        return 0;
    }
    if (!strcmp(cmd, "halton")) {  // args: W H ; in: int64 quads (px,py,s,dim)
        int W = atoi(argv[5]), H = atoi(argv[6]);
        HaltonSampler sampler(1, Bounds2i(Point2i(0, 0), Point2i(W, H)), false);
        size_t n = in.size() / 32;
        const int64_t *q = (const int64_t *)in.data();
        std::vector<float> out(n);
        for (size_t i = 0; i < n; ++i) {
            sampler.StartPixel(Point2i((int)q[4 * i], (int)q[4 * i + 1]));
            int64_t idx = sampler.GetIndexForSample(q[4 * i + 2]);
            out[i] = sampler.SampleDimension(idx, (int)q[4 * i + 3]);
        }
        writeAll(outPath, out.data(), out.size() * 4);
        return 0;
    }
    if (!strcmp(cmd, "hdr")) {  // args: path ; out: int32 w,h then floats rgb
        int w, h, nc;
        float *d = stbi_loadf(argv[5], &w, &h, &nc, 0);
        if (!d) return 3;
        std::vector<float> out((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; ++i) for (int c = 0; c < 3; ++c) out[3 * i + c] = d[i * nc + c];
        FILE *f = fopen(outPath, "wb");
        int32_t wh[2] = {w, h};
        fwrite(wh, 4, 2, f);
        fwrite(out.data(), 4, out.size(), f);
        fclose(f);
        return 0;
    }

    RefScene rs;
    if (!readScene(scenePath, &rs.sf)) { fprintf(stderr, "cannot read scene %s\n", scenePath); return 2; }

    if (!strcmp(cmd, "camrays")) {  // args: W H ; in: int64 triples (px,py,s) ; out: o[3] d[3] per ray
        int W = atoi(argv[5]), H = atoi(argv[6]);
        const gnxr_camera &c = rs.sf.cam;
        Transform lookat = LookAt(Point3f(c.eye[0], c.eye[1], c.eye[2]), Point3f(c.look[0], c.look[1], c.look[2]), Vector3f(c.up[0], c.up[1], c.up[2]));
        Transform c2w = Inverse(lookat), c2wEnd = c2w;
        AnimatedTransform anim(&c2w, 0.0f, &c2wEnd, 1.0f);
        std::unique_ptr<Camera> cam(c.orthographic ? (Camera *)CreateOrthographicCamera(W, H, anim) : (Camera *)CreatePerspectiveCamera(W, H, anim));
        HaltonSampler sampler(1, Bounds2i(Point2i(0, 0), Point2i(W, H)), false);
        size_t n = in.size() / 24;
        const int64_t *q = (const int64_t *)in.data();
        std::vector<float> out(6 * n);
        for (size_t i = 0; i < n; ++i) {
            Point2i pixel((int)q[3 * i], (int)q[3 * i + 1]);
            sampler.StartPixel(pixel);
            sampler.SetSampleNumber(q[3 * i + 2]);
            CameraSample cs = sampler.GetCameraSample(pixel);
            RayDifferential ray;
            cam->GenerateRayDifferential(cs, &ray);
            out[6 * i] = ray.o.x; out[6 * i + 1] = ray.o.y; out[6 * i + 2] = ray.o.z;
            out[6 * i + 3] = ray.d.x; out[6 * i + 4] = ray.d.y; out[6 * i + 5] = ray.d.z;
        }
        writeAll(outPath, out.data(), out.size() * 4);
        return 0;
    }

    rs.build();
    const Scene &scene = *rs.scene;

    if (!strcmp(cmd, "bvh")) {  // out: int32 nNodes, then per node 6 floats + 3 int32 (offset,nPrims,axis); then nt int32 ordered prims
        // node count: DFS over the flattened array
        const RefLinearBVHNode *nodes = (const RefLinearBVHNode *)rs.bvh->nodes;
        int nNodes = 0;
        {  // nodes are contiguous in DFS order; count by walking
            std::vector<int> stack{0};
            while (!stack.empty()) {
                int i = stack.back(); stack.pop_back();
                nNodes = std::max(nNodes, i + 1);
                if (nodes[i].nPrimitives == 0) { stack.push_back(i + 1); stack.push_back(nodes[i].offset); }
            }
        }
        FILE *f = fopen(outPath, "wb");
        int32_t nn = nNodes;
        fwrite(&nn, 4, 1, f);
        for (int i = 0; i < nNodes; ++i) {
            float b[6] = {nodes[i].bounds.pMin.x, nodes[i].bounds.pMin.y, nodes[i].bounds.pMin.z, nodes[i].bounds.pMax.x, nodes[i].bounds.pMax.y, nodes[i].bounds.pMax.z};
            int32_t m[3] = {nodes[i].offset, nodes[i].nPrimitives, nodes[i].nPrimitives ? 0 : nodes[i].axis};
            fwrite(b, 4, 6, f);
            fwrite(m, 4, 3, f);
        }
        for (size_t i = 0; i < rs.bvh->primitives.size(); ++i) { int32_t p = rs.primIndex[rs.bvh->primitives[i].get()]; fwrite(&p, 4, 1, f); }
        fclose(f);
        return 0;
    }
    if (!strcmp(cmd, "closest")) {  // in: gnxr_ray[] ; out: gnxr_hit[]
        size_t n = in.size() / sizeof(gnxr_ray);
        const gnxr_ray *rays = (const gnxr_ray *)in.data();
        std::vector<gnxr_hit> out(n);
        for (size_t i = 0; i < n; ++i) {
            Ray r(Point3f(rays[i].o[0], rays[i].o[1], rays[i].o[2]), Vector3f(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].tmax);
            SurfaceInteraction isect;
            gnxr_hit h;
            memset(&h, 0, sizeof(h));
            h.prim = -1;
            if (scene.Intersect(r, &isect)) {
                h.prim = rs.primIndex[isect.primitive];
                h.t = r.tMax;
                // barycentrics are not kept by the reference; recover p-based check values instead
                h.b0 = isect.p.x; h.b1 = isect.p.y; h.b2 = isect.p.z;
                h.n[0] = isect.n.x; h.n[1] = isect.n.y; h.n[2] = isect.n.z;
            }
            out[i] = h;
        }
        writeAll(outPath, out.data(), out.size() * sizeof(gnxr_hit));
        return 0;
    }
    if (!strcmp(cmd, "any")) {
        size_t n = in.size() / sizeof(gnxr_ray);
        const gnxr_ray *rays = (const gnxr_ray *)in.data();
        std::vector<uint8_t> out(n);
        for (size_t i = 0; i < n; ++i) {
            Ray r(Point3f(rays[i].o[0], rays[i].o[1], rays[i].o[2]), Vector3f(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].tmax);
            out[i] = scene.IntersectP(r) ? 1 : 0;
        }
        writeAll(outPath, out.data(), out.size());
        return 0;
    }
    if (!strcmp(cmd, "bsdf")) {  // args: flags ; in: n * (gnxr_ray, wi[3], u[2]) packed as arrays: rays | wi | u
        int flags = atoi(argv[5]);
        const float diffEps = argc > 6 ? (float)atof(argv[6]) : 0.f;
        size_t n = in.size() / (sizeof(gnxr_ray) + 12 + 8);
        const gnxr_ray *rays = (const gnxr_ray *)in.data();
        const float *wiW = (const float *)(in.data() + n * sizeof(gnxr_ray));
        const float *u2 = wiW + 3 * n;
        std::vector<float> out(16 * n, 0.f);
        for (size_t i = 0; i < n; ++i) {
            float *o = &out[16 * i];
            RayDifferential r(Point3f(rays[i].o[0], rays[i].o[1], rays[i].o[2]), Vector3f(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].tmax);
            if (diffEps > 0) {   // synthetic offset rays (texture filtering probe): same origin, directions nudged along x / y
                r.hasDifferentials = true;
                r.rxOrigin = r.ryOrigin = r.o;
                r.rxDirection = r.d + Vector3f(diffEps, 0, 0);
                r.ryDirection = r.d + Vector3f(0, diffEps, 0);
            }
            SurfaceInteraction isect;
            if (!scene.Intersect(r, &isect)) continue;
            MemoryArena arena;
            isect.ComputeScatteringFunctions(r, arena, true);
            if (!isect.bsdf) continue;
            o[13] = 1; o[14] = isect.dudx; o[15] = isect.dvdy;
            Vector3f wi(wiW[3 * i], wiW[3 * i + 1], wiW[3 * i + 2]);
            Spectrum f = isect.bsdf->f(isect.wo, wi, BxDFType(flags));
            o[0] = f[0]; o[1] = f[1]; o[2] = f[2];
            o[3] = isect.bsdf->Pdf(isect.wo, wi, BxDFType(flags));
            Vector3f wis(0, 0, 0);
            Float pdf = 0;
            BxDFType st = BxDFType(0);
            Spectrum sf = isect.bsdf->Sample_f(isect.wo, &wis, Point2f(u2[2 * i], u2[2 * i + 1]), &pdf, BxDFType(flags), &st);
            if (pdf == 0) { sf = Spectrum(0.f); wis = Vector3f(0, 0, 0); }
            o[4] = sf[0]; o[5] = sf[1]; o[6] = sf[2]; o[7] = pdf;
            o[8] = wis.x; o[9] = wis.y; o[10] = wis.z; o[11] = (float)(int)st; o[12] = (float)isect.bsdf->NumComponents(BxDFType(flags));
        }
        writeAll(outPath, out.data(), out.size() * 4);
        return 0;
    }
    if (!strcmp(cmd, "light")) {  // args: light ; in: refP[3n] | refN[3n] | u[2n] | wiQuery[3n]
        int li = atoi(argv[5]);
        size_t n = in.size() / (4 * 11);
        const float *refP = (const float *)in.data(), *refN = refP + 3 * n, *u2 = refN + 3 * n, *wiQ = u2 + 2 * n;
        std::vector<float> out(12 * n, 0.f);
        const Light &light = *scene.lights[li];
        for (size_t i = 0; i < n; ++i) {
            float *o = &out[12 * i];
            Interaction ref;
            ref.p = Point3f(refP[3 * i], refP[3 * i + 1], refP[3 * i + 2]);
            ref.n = Normal3f(refN[3 * i], refN[3 * i + 1], refN[3 * i + 2]);
            Vector3f wi(0, 0, 0);
            Float pdf = 0;
            VisibilityTester vis;
            Spectrum Li = light.Sample_Li(ref, Point2f(u2[2 * i], u2[2 * i + 1]), &wi, &pdf, &vis);
            o[0] = Li[0]; o[1] = Li[1]; o[2] = Li[2]; o[3] = pdf;
            o[4] = wi.x; o[5] = wi.y; o[6] = wi.z;
            o[7] = light.Pdf_Li(ref, Vector3f(wiQ[3 * i], wiQ[3 * i + 1], wiQ[3 * i + 2]));
            o[8] = 0;
            if (pdf > 0) { o[9] = vis.P1().p.x; o[10] = vis.P1().p.y; o[11] = vis.P1().p.z; }
        }
        writeAll(outPath, out.data(), out.size() * 4);
        return 0;
    }
    if (!strcmp(cmd, "le")) {  // args: light ; in: gnxr_ray[]
        int li = atoi(argv[5]);
        size_t n = in.size() / sizeof(gnxr_ray);
        const gnxr_ray *rays = (const gnxr_ray *)in.data();
        std::vector<float> out(3 * n);
        for (size_t i = 0; i < n; ++i) {
            RayDifferential r(Point3f(rays[i].o[0], rays[i].o[1], rays[i].o[2]), Vector3f(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].tmax);
            Spectrum L = scene.lights[li]->Le(r);
            out[3 * i] = L[0]; out[3 * i + 1] = L[1]; out[3 * i + 2] = L[2];
        }
        writeAll(outPath, out.data(), out.size() * 4);
        return 0;
    }
    if (!strcmp(cmd, "lightdist")) {  // args: strategy ; in: p[3n] ; out: per point nl floats (discrete pdf per light)
        int strat = atoi(argv[5]);
        RefLightDistribution ld(scene, strat);
        size_t n = in.size() / 12, nl = scene.lights.size();
        const float *p = (const float *)in.data();
        std::vector<float> out(n * nl);
        for (size_t i = 0; i < n; ++i) {
            const Distribution1D *d = ld.Lookup(Point3f(p[3 * i], p[3 * i + 1], p[3 * i + 2]));
            for (size_t j = 0; j < nl; ++j) out[i * nl + j] = d->DiscretePDF((int)j);
        }
        writeAll(outPath, out.data(), out.size() * 4);
        return 0;
    }
    if (!strcmp(cmd, "render")) {  // args: W H spp maxDepth rrThreshold strategy [threads] ; out: float rgba[W*H*4] + 2 uint64 counts + double secs
        int W = atoi(argv[5]), H = atoi(argv[6]), spp = atoi(argv[7]), maxDepth = atoi(argv[8]);
        Float rr = (Float)atof(argv[9]);
        int strat = atoi(argv[10]);
        if (argc > 11 && atoi(argv[11]) > 0) omp_set_num_threads(atoi(argv[11]));
        const int integ = argc > 12 ? atoi(argv[12]) : 0;   // 0 Path, 1 VolPath, 2 Whitted, 3 DirectLighting
        const int directStrategy = argc > 13 ? atoi(argv[13]) : 0;   // LightStrategy: 0 UniformSampleAll, 1 UniformSampleOne
        const bool volpath = integ == 1;
        const gnxr_camera &c = rs.sf.cam;
        Transform lookat = LookAt(Point3f(c.eye[0], c.eye[1], c.eye[2]), Point3f(c.look[0], c.look[1], c.look[2]), Vector3f(c.up[0], c.up[1], c.up[2]));
        Transform c2w = Inverse(lookat), c2wEnd = c2w;
        AnimatedTransform anim(&c2w, 0.0f, &c2wEnd, 1.0f);
        std::unique_ptr<Camera> cam(c.orthographic ? (Camera *)CreateOrthographicCamera(W, H, anim) : (Camera *)CreatePerspectiveCamera(W, H, anim));
        HaltonSampler proto(spp, Bounds2i(Point2i(0, 0), Point2i(W, H)), false);
        RefLightDistribution ld(scene, strat);
        std::vector<int> nLightSamples;
        if (integ == 3 && directStrategy == 0) {   // DirectLightingIntegrator::Preprocess, DirectLightingIntegrator.cpp:11-28
            for (const auto &light : scene.lights) nLightSamples.push_back(proto.RoundCount(light->nSamples));
            for (int i = 0; i < maxDepth; ++i)
                for (size_t j = 0; j < scene.lights.size(); ++j) { proto.Request2DArray(nLightSamples[j]); proto.Request2DArray(nLightSamples[j]); }
        }
        std::vector<float> img((size_t)W * H * 4, 0.f);
        rs.counting->nI = 0; rs.counting->nP = 0;
        auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1)
        for (int i = 0; i < W; i++) {
            for (int j = 0; j < H; j++) {
                MemoryArena arena;
                std::unique_ptr<Sampler> ps = proto.Clone(i + W * j);
                Point2i pixel(i, j);
                ps->StartPixel(pixel);
                Spectrum col(.0f);
                do {
                    CameraSample cs = ps->GetCameraSample(pixel);
                    RayDifferential ray;
                    cam->GenerateRayDifferential(cs, &ray);
                    ray.ScaleDifferentials(1 / std::sqrt((Float)ps->samplesPerPixel));
                    col += integ == 3 ? refDirectLi(ray, scene, *ps, arena, directStrategy, nLightSamples, maxDepth, 0) : integ == 2 ? refWhittedLi(ray, scene, *ps, arena, maxDepth, 0)
                                      : (volpath ? refVolPathLi(ray, scene, *ps, arena, ld, maxDepth, rr) : refPathLi(ray, scene, *ps, arena, ld, maxDepth, rr));
                } while (ps->StartNextSample());
                col = col / ps->samplesPerPixel;
                size_t o = ((size_t)i + (size_t)j * W) * 4;
                img[o] = col[0]; img[o + 1] = col[1]; img[o + 2] = col[2]; img[o + 3] = 1.f;
            }
        }
        auto t1 = std::chrono::steady_clock::now();
        double secs = std::chrono::duration<double>(t1 - t0).count();
        uint64_t counts[2] = {rs.counting->nI.load(), rs.counting->nP.load()};
        FILE *f = fopen(outPath, "wb");
        fwrite(img.data(), 4, img.size(), f);
        fwrite(counts, 8, 2, f);
        fwrite(&secs, 8, 1, f);
        fclose(f);
        double sum = 0;
        for (size_t p = 0; p < (size_t)W * H; ++p) sum += (double)img[4 * p] + img[4 * p + 1] + img[4 * p + 2];
        fprintf(stderr, "render %dx%d spp=%d: %.3f s, rays closest=%llu any=%llu, checksum(sum rgb)=%.6f\n", W, H, spp, secs,
                (unsigned long long)counts[0], (unsigned long long)counts[1], sum);
        return 0;
    }
    fprintf(stderr, "unknown command %s\n", cmd);
    return 2;
}

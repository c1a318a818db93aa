// ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_integrator.h: CPU restatement of the light side and the estimator loop:
//   DiffuseAreaLight::L / Sample_Li / Pdf_Li          lights/DiffuseAreaLight.{h,cpp}:22-27,37-57
//   VisibilityTester::Unoccluded                       core/Light.cpp:28-31
//   Uniform/Power/SpatialLightDistribution             core/LightDistribution.cpp:15-274
//   UniformSampleOneLight / EstimateDirect             core/Integrator.cpp:57-79, 93-210
//   PathIntegrator::Li                                 integrators/PathIntegrator.cpp:62-208
//   SamplerIntegrator::Render                          core/Integrator.cpp:225-319
// core/Integrator.cpp, core/LightDistribution.cpp and integrators/*.cpp cannot be compiled in this
// image (they include ui/FrameBuffer.h -> <QObject>, Qt is absent), so for these functions the
// restatement is pinned by the reference's recorded ray counts and by
// function-level agreement of everything they call with oracle/_ref.
#pragma once
#include <memory>
#include <unordered_map>

#include "o_bsdf.h"
#include "o_envlight.h"

namespace gnxo {

struct RenderContext {
    const Scene *scene = nullptr;
    std::vector<int> lightTri;             // per light: authoring triangle (AREA_TRI)
    std::vector<Float> lightArea;
    std::vector<int> infiniteLights;       // indices of lights flagged Infinite (Scene ctor, Scene.h:20-27)
    std::vector<std::unique_ptr<InfiniteAreaLight>> envLights;  // per light index (null if not INFINITE)
    int lightStrategy = GNXR_LIGHTS_SPATIAL;
    const void *mediaSet = nullptr;        // MediaSet (o_media.h), set by the render entry point for VolPath
    // light distributions
    Distribution1D uniformOrPower;
    int nVoxels[3] = {1, 1, 1};
    mutable std::vector<std::unique_ptr<Distribution1D>> voxelDistrib;  // dense, lazily filled
    mutable std::vector<std::atomic<int>> voxelState;                    // 0 empty, 1 building, 2 ready
    Bounds3 worldBound;

    // delta lights: lights/PointLight.cpp, lights/SpotLight.cpp, lights/DistantLight.cpp
    struct DeltaLight {
        V3 pLight;                    // Point / Spot: LightToWorld(Point3f(0, 0, 0))
        M44 worldToLight;             // Spot::Falloff
        Float cosTotalWidth = 0, cosFalloffStart = 0;
        V3 wLight;                    // Distant: Normalize(LightToWorld(wLight))
        Float worldRadius = 0;        // Distant::Preprocess
    };
    std::vector<DeltaLight> deltaLights;   // per light index (unused entries for the other types)
    static bool IsDeltaType(int type) { return type == GNXR_LIGHT_POINT || type == GNXR_LIGHT_SPOT || type == GNXR_LIGHT_DISTANT; }
    // IsDeltaLight(flags), core/Light.h:28-32
    bool IsDeltaLight(int light) const { return IsDeltaType(scene->lights[light].type); }
    // SpotLight::Falloff, SpotLight.cpp:30-40
    Float SpotFalloff(const DeltaLight &d, const V3 &w) const {
        V3 wl = Normalize(XVector(d.worldToLight, w));
        Float cosTheta = wl.z;
        if (cosTheta < d.cosTotalWidth) return 0;
        if (cosTheta >= d.cosFalloffStart) return 1;
        Float delta = (cosTheta - d.cosTotalWidth) / (d.cosFalloffStart - d.cosTotalWidth);
        return (delta * delta) * (delta * delta);
    }

    // DiffuseAreaLight::L, DiffuseAreaLight.h:22-27.  `bool dotNW = Dot(intr.n, w)` truncates the dot
    // product to bool, so the light emits from both faces whenever the dot product is non-zero.
    Spec AreaL(int light, const V3 &n, const V3 &w) const {
        const gnxr_light &l = scene->lights[light];
        bool dotNW = Dot(n, w);
        return (l.two_sided || dotNW > 0) ? Spec(l.le[0], l.le[1], l.le[2]) : Spec(0.f);
    }
    // SurfaceInteraction::Le, Interaction.cpp:116-120
    Spec Le(const SurfaceInteraction &isect, const V3 &w) const {
        int light = scene->triLight[isect.prim];
        return light >= 0 ? AreaL(light, isect.n, w) : Spec(0.f);
    }
    // Light::Le(ray) for escaped rays
    Spec LightLe(int light, const Ray &ray) const {
        const gnxr_light &l = scene->lights[light];
        if (l.type == GNXR_LIGHT_INFINITE) return envLights[light]->Le(ray);
        if (l.type == GNXR_LIGHT_SKYBOX) return SkyBoxLe(l, ray);
        return Spec(0.f);
    }
    // <Light>::Sample_Li
    LightSample Sample_Li(int light, const Interaction &ref, const P2 &u) const {
        const gnxr_light &l = scene->lights[light];
        LightSample s;
        if (l.type == GNXR_LIGHT_AREA_TRI) {  // DiffuseAreaLight.cpp:37-52
            Interaction pShape = scene->ShapeSample(l.tri, ref, u, &s.pdf);
            if (s.pdf == 0 || (pShape.p - ref.p).LengthSquared() == 0) { s.pdf = 0; s.Li = Spec(0.f); return s; }
            s.wi = Normalize(pShape.p - ref.p);
            s.p1 = pShape;
            s.Li = AreaL(light, pShape.n, -s.wi);
            return s;
        } else if (l.type == GNXR_LIGHT_INFINITE) {
            return envLights[light]->Sample_Li(ref, u);
        } else if (IsDeltaType(l.type)) {
            const DeltaLight &d = deltaLights[light];
            const Spec I(l.le[0], l.le[1], l.le[2]);
            s.pdf = 1.f;
            s.p1 = Interaction();   // Interaction(p, time, mediumInterface): n = pError = 0
            if (l.type == GNXR_LIGHT_DISTANT) {          // DistantLight.cpp:15-25
                s.wi = d.wLight;
                s.p1.p = ref.p + d.wLight * (2 * d.worldRadius);
                s.Li = I;
            } else {                                     // PointLight.cpp:13-22, SpotLight.cpp:19-28
                s.wi = Normalize(d.pLight - ref.p);
                s.p1.p = d.pLight;
                if (l.type == GNXR_LIGHT_SPOT) s.Li = I * SpotFalloff(d, -s.wi) / DistanceSquared(d.pLight, ref.p);
                else s.Li = I / DistanceSquared(d.pLight, ref.p);
            }
            return s;
        } else {  // SkyBoxLight::Sample_Li, SkyBoxLight.cpp:43-53
            return SkyBoxSample_Li(l, ref, u);
        }
    }
    Float Pdf_Li(int light, const Interaction &ref, const V3 &wi) const {
        const gnxr_light &l = scene->lights[light];
        if (l.type == GNXR_LIGHT_AREA_TRI) return scene->ShapePdf(l.tri, ref, wi);  // DiffuseAreaLight.cpp:54-57
        if (l.type == GNXR_LIGHT_INFINITE) return envLights[light]->Pdf_Li(ref, wi);
        return 0;  // SkyBoxLight::Pdf_Li
    }
    // <Light>::Power().y() for the "power" strategy
    Float LightPowerY(int light) const {
        const gnxr_light &l = scene->lights[light];
        if (l.type == GNXR_LIGHT_AREA_TRI) {  // DiffuseAreaLight.cpp:32-35
            Spec P = (l.two_sided ? 2 : 1) * Spec(l.le[0], l.le[1], l.le[2]) * lightArea[light] * Pi;
            return P.y();
        }
        if (l.type == GNXR_LIGHT_INFINITE) return envLights[light]->Power().y();
        const Spec I(l.le[0], l.le[1], l.le[2]);
        if (l.type == GNXR_LIGHT_POINT) return (4 * Pi * I).y();                                       // PointLight.cpp:24
        if (l.type == GNXR_LIGHT_SPOT)                                                                  // SpotLight.cpp:42-45
            return (I * 2 * Pi * (1 - .5f * (deltaLights[light].cosFalloffStart + deltaLights[light].cosTotalWidth))).y();
        if (l.type == GNXR_LIGHT_DISTANT) return (I * Pi * deltaLights[light].worldRadius * deltaLights[light].worldRadius).y();   // DistantLight.cpp:27-30
        return 0;
    }

    void Init(const Scene *s, int strategy) {
        scene = s;
        lightStrategy = strategy;
        int nl = (int)s->lights.size();
        lightTri.assign(nl, -1);
        lightArea.assign(nl, 0);
        envLights.resize(nl);
        deltaLights.assign(nl, DeltaLight());
        worldBound = s->WorldBound();
        for (int i = 0; i < nl; ++i) {
            const gnxr_light &l = s->lights[i];
            if (l.type == GNXR_LIGHT_AREA_TRI) { lightTri[i] = l.tri; lightArea[i] = s->TriArea(l.tri); }
            else if (IsDeltaType(l.type)) {
                DeltaLight &d = deltaLights[i];
                M44 l2w = M44::FromRowMajor(l.light_to_world);
                d.worldToLight = Inverse(l2w);
                d.pLight = XPoint(l2w, V3(0, 0, 0));
                d.cosTotalWidth = std::cos(Radians(l.radius));
                d.cosFalloffStart = std::cos(Radians(l.falloff_start));
                d.wLight = Normalize(XVector(l2w, V3(l.center[0], l.center[1], l.center[2])));
                V3 c = (worldBound.pMin + worldBound.pMax) / 2;   // Preprocess: scene.WorldBound().BoundingSphere, Geometry.h:770-773
                bool inside = c.x >= worldBound.pMin.x && c.x <= worldBound.pMax.x && c.y >= worldBound.pMin.y && c.y <= worldBound.pMax.y &&
                              c.z >= worldBound.pMin.z && c.z <= worldBound.pMax.z;
                d.worldRadius = inside ? (c - worldBound.pMax).Length() : 0;
            } else {
                infiniteLights.push_back(i);
                if (l.type == GNXR_LIGHT_INFINITE) {
                    bool flipY = false;  // a SkyBoxLight built earlier switched stb_image to flipped loading
                    for (int k = 0; k < i; ++k) if (s->lights[k].type == GNXR_LIGHT_SKYBOX) flipY = true;
                    envLights[i].reset(new InfiniteAreaLight(l, s->envRgb.data(), s->envW, s->envH, flipY));
                    envLights[i]->Preprocess(worldBound);  // Scene ctor -> light->Preprocess(*this)
                }
            }
        }
        // CreateLightSampleDistribution, LightDistribution.cpp:15-33
        if (nl == 0) return;
        if (strategy == GNXR_LIGHTS_UNIFORM || nl == 1) {
            lightStrategy = GNXR_LIGHTS_UNIFORM;
            std::vector<Float> prob(nl, Float(1));
            uniformOrPower = Distribution1D(&prob[0], nl);
        } else if (strategy == GNXR_LIGHTS_POWER) {  // ComputeLightPowerDistribution, Integrator.cpp:212-220
            std::vector<Float> lightPower;
            for (int i = 0; i < nl; ++i) lightPower.push_back(LightPowerY(i));
            uniformOrPower = Distribution1D(&lightPower[0], nl);
        } else {  // SpatialLightDistribution ctor, LightDistribution.cpp:70-97 (maxVoxels = 64)
            V3 diag = worldBound.Diagonal();
            Float bmax = diag[worldBound.MaximumExtent()];
            for (int i = 0; i < 3; ++i) nVoxels[i] = std::max(1, int(std::round(diag[i] / bmax * 64)));
            size_t nv = (size_t)nVoxels[0] * nVoxels[1] * nVoxels[2];
            voxelDistrib.resize(nv);
            voxelState = std::vector<std::atomic<int>>(nv);
            for (auto &a : voxelState) a.store(0);
        }
    }

    // SpatialLightDistribution::ComputeDistribution, LightDistribution.cpp:206-274
    Distribution1D *ComputeDistribution(int pi[3]) const {
        V3 p0(Float(pi[0]) / Float(nVoxels[0]), Float(pi[1]) / Float(nVoxels[1]), Float(pi[2]) / Float(nVoxels[2]));
        V3 p1(Float(pi[0] + 1) / Float(nVoxels[0]), Float(pi[1] + 1) / Float(nVoxels[1]), Float(pi[2] + 1) / Float(nVoxels[2]));
        Bounds3 voxelBounds(worldBound.Lerp(p0), worldBound.Lerp(p1));
        int nSamples = 128;
        int nl = (int)scene->lights.size();
        std::vector<Float> lightContrib(nl, Float(0));
        for (int i = 0; i < nSamples; ++i) {
            V3 po = voxelBounds.Lerp(V3(RadicalInverse(0, i), RadicalInverse(1, i), RadicalInverse(2, i)));
            Interaction intr;
            intr.p = po; intr.n = V3(); intr.pError = V3(); intr.wo = Normalize(V3(1, 0, 0));
            P2 u(RadicalInverse(3, i), RadicalInverse(4, i));
            for (int j = 0; j < nl; ++j) {
                LightSample ls = Sample_Li(j, intr, u);
                if (ls.pdf > 0) lightContrib[j] += ls.Li.y() / ls.pdf;
            }
        }
        Float sumContrib = 0;
        for (Float c : lightContrib) sumContrib += c;  // std::accumulate(.., Float(0))
        Float avgContrib = sumContrib / (nSamples * lightContrib.size());
        Float minContrib = (avgContrib > 0) ? .001 * avgContrib : 1;
        for (size_t i = 0; i < lightContrib.size(); ++i) lightContrib[i] = std::max(lightContrib[i], minContrib);
        return new Distribution1D(&lightContrib[0], nl);
    }
    // LightDistribution::Lookup.  The reference's lock-free hash (LightDistribution.cpp:109-204) only
    // memoises ComputeDistribution per voxel; a dense lazily-filled table returns the same object.
    const Distribution1D *Lookup(const V3 &p) const {
        // no lights: the integrators return before they read the distribution (Integrator.cpp:63-64)
        if (scene->lights.empty() || lightStrategy != GNXR_LIGHTS_SPATIAL) return &uniformOrPower;
        V3 offset = worldBound.Offset(p);
        int pi[3];
        for (int i = 0; i < 3; ++i) pi[i] = Clamp(int(offset[i] * nVoxels[i]), 0, nVoxels[i] - 1);
        size_t idx = ((size_t)pi[0] * nVoxels[1] + pi[1]) * nVoxels[2] + pi[2];
        int st = voxelState[idx].load(std::memory_order_acquire);
        if (st == 2) return voxelDistrib[idx].get();
        int expected = 0;
        if (voxelState[idx].compare_exchange_strong(expected, 1)) {
            voxelDistrib[idx].reset(ComputeDistribution(pi));
            voxelState[idx].store(2, std::memory_order_release);
        } else {
            while (voxelState[idx].load(std::memory_order_acquire) != 2) {}
        }
        return voxelDistrib[idx].get();
    }
};

// EstimateDirect for a surface interaction with handleMedia = false, specular = false,
// core/Integrator.cpp:93-210 (the stray printf at :143 is not reproduced: it changes no value).
inline Spec EstimateDirect(const RenderContext &rc, const SurfaceInteraction &isect, const BSDF &bsdf, const P2 &uScattering,
                           int light, const P2 &uLight) {
    const Scene &scene = *rc.scene;
    int bsdfFlags = BSDF_ALL & ~BSDF_SPECULAR;
    Spec Ld(0.f);
    Float lightPdf = 0, scatteringPdf = 0;
    LightSample ls = rc.Sample_Li(light, isect, uLight);
    V3 wi = ls.wi;
    lightPdf = ls.pdf;
    Spec Li = ls.Li;
    if (lightPdf > 0 && !Li.IsBlack()) {
        Spec f = bsdf.f(isect.wo, wi, bsdfFlags) * AbsDot(wi, isect.sn);
        scatteringPdf = bsdf.Pdf(isect.wo, wi, bsdfFlags);
        if (!f.IsBlack()) {
            if (scene.IntersectP(isect.SpawnRayTo(ls.p1))) Li = Spec(0.f);  // !visibility.Unoccluded(scene)
            if (!Li.IsBlack()) {
                if (rc.IsDeltaLight(light)) Ld += f * Li / lightPdf;
                else {
                    Float weight = PowerHeuristic(1, lightPdf, 1, scatteringPdf);
                    Ld += f * Li * weight / lightPdf;
                }
            }
        }
    }
    if (!rc.IsDeltaLight(light)) {
        Spec f;
        bool sampledSpecular = false;
        int sampledType = 0;
        f = bsdf.Sample_f(isect.wo, &wi, uScattering, &scatteringPdf, bsdfFlags, &sampledType);
        f *= AbsDot(wi, isect.sn);
        sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
        if (!f.IsBlack() && scatteringPdf > 0) {
            Float weight = 1;
            if (!sampledSpecular) {
                lightPdf = rc.Pdf_Li(light, isect, wi);
                if (lightPdf == 0) return Ld;
                weight = PowerHeuristic(1, scatteringPdf, 1, lightPdf);
            }
            SurfaceInteraction lightIsect;
            Ray ray = isect.SpawnRay(wi);
            bool foundSurfaceInteraction = scene.Intersect(ray, &lightIsect);  // closest-hit, Integrator.cpp:196-197
            Spec Li2(0.f);
            if (foundSurfaceInteraction) {
                if (scene.triLight[lightIsect.prim] == light) Li2 = rc.Le(lightIsect, -wi);  // GetAreaLight() == &light
            } else
                Li2 = rc.LightLe(light, ray);
            if (!Li2.IsBlack()) Ld += f * Li2 * Spec(1.f) * weight / scatteringPdf;
        }
    }
    return Ld;
}

// core/Integrator.cpp:57-79
inline Spec UniformSampleOneLight(const RenderContext &rc, const SurfaceInteraction &isect, const BSDF &bsdf, SampleStream &sampler,
                                  const Distribution1D *lightDistrib) {
    int nLights = (int)rc.scene->lights.size();
    if (nLights == 0) return Spec(0.f);
    int lightNum;
    Float lightPdf;
    if (lightDistrib) {
        lightNum = lightDistrib->SampleDiscrete(sampler.Get1D(), &lightPdf);
        if (lightPdf == 0) return Spec(0.f);
    } else {
        lightNum = std::min((int)(sampler.Get1D() * nLights), nLights - 1);
        lightPdf = Float(1) / nLights;
    }
    P2 uLight = sampler.Get2D();
    P2 uScattering = sampler.Get2D();
    return EstimateDirect(rc, isect, bsdf, uScattering, lightNum, uLight) / lightPdf;
}

struct PathParams { int maxDepth = 5; Float rrThreshold = 1; };

// integrators/PathIntegrator.cpp:62-208
inline Spec PathLi(const RenderContext &rc, const PathParams &pp, const Ray &r, SampleStream &sampler) {
    const Scene &scene = *rc.scene;
    Spec L(0.f), beta(1.f);
    Ray ray(r);
    ray.hasDifferentials = false;   // `Ray ray(r);` (PathIntegrator.cpp:67) slices the RayDifferential: Path never filters textures
    bool specularBounce = false;
    int bounces;
    Float etaScale = 1;
    for (bounces = 0;; ++bounces) {
        SurfaceInteraction isect;
        bool foundIntersection = scene.Intersect(ray, &isect);
        if (bounces == 0 || specularBounce) {
            if (foundIntersection) L += beta * rc.Le(isect, -ray.d);
            else for (int light : rc.infiniteLights) L += beta * rc.LightLe(light, ray);
        }
        if (!foundIntersection || bounces >= pp.maxDepth) break;
        BSDF bsdf;
        if (!ComputeScatteringFunctions(scene, ray, &isect, true, &bsdf)) {
            ray = isect.SpawnRay(ray.d);
            bounces--;
            continue;
        }
        const Distribution1D *distrib = rc.Lookup(isect.p);
        if (bsdf.NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
            Spec Ld = beta * UniformSampleOneLight(rc, isect, bsdf, sampler, distrib);
            L += Ld;
        }
        V3 wo = -ray.d, wi;
        Float pdf;
        int flags = 0;
        Spec f = bsdf.Sample_f(wo, &wi, sampler.Get2D(), &pdf, BSDF_ALL, &flags);
        if (f.IsBlack() || pdf == 0.f) break;
        beta *= f * AbsDot(wi, isect.sn) / pdf;
        specularBounce = (flags & BSDF_SPECULAR) != 0;
        if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
            Float eta = bsdf.eta;
            etaScale *= (Dot(wo, isect.n) > 0) ? (eta * eta) : 1 / (eta * eta);
        }
        ray = isect.SpawnRay(wi);
        Spec rrBeta = beta * etaScale;
        if (rrBeta.MaxComponentValue() < pp.rrThreshold && bounces > 3) {
            Float q = std::max((Float).05, 1 - rrBeta.MaxComponentValue());
            if (sampler.Get1D() < q) break;
            beta /= 1 - q;
        }
    }
    return L;
}

}  // namespace gnxo

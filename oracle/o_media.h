// ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_media.h: CPU restatement of the participating-media path (BASELINE config 5):
//   HomogeneousMedium::Tr / Sample                  media/HomogeneousMedium.cpp:11-43
//   GridDensityMedium::Density / Sample / Tr        media/GridDensityMedium.cpp:14-87, GridDensityMedium.h:19-44
//   HenyeyGreenstein::p / Sample_p, PhaseHG         core/Medium.cpp:164-189, core/Medium.h:34-38
//   VisibilityTester::Tr                            core/Light.cpp:33-53
//   Scene::IntersectTr                              core/Scene.cpp:26-40
//   EstimateDirect (handleMedia = true)             core/Integrator.cpp:93-210
//   VolPathIntegrator::Li                           integrators/VolPathIntegrator.cpp:24-159
//   WhittedIntegrator::Li / SpecularReflect/Transmit integrators/WhittedIntegrator.cpp:14-68, core/Integrator.cpp:321-442
#pragma once
#include "o_integrator.h"

namespace gnxo {

static constexpr Float MaxFloat = std::numeric_limits<Float>::max();
inline Spec Exp(const Spec &s) { return Spec(std::exp(s.c[0]), std::exp(s.c[1]), std::exp(s.c[2])); }

// core/Medium.h:34-38
inline Float PhaseHG(Float cosTheta, Float g) {
    Float denom = 1 + g * g + 2 * g * cosTheta;
    return Inv4Pi * (1 - g * g) / (denom * std::sqrt(denom));
}
// core/Medium.cpp:164-189
inline Float HG_p(Float g, const V3 &wo, const V3 &wi) { return PhaseHG(Dot(wo, wi), g); }
inline Float HG_Sample_p(Float g, const V3 &wo, V3 *wi, const P2 &u) {
    Float cosTheta;
    if (std::abs(g) < 1e-3) cosTheta = 1 - 2 * u.x;
    else {
        Float sqrTerm = (1 - g * g) / (1 + g - 2 * g * u.x);
        cosTheta = -(1 + g * g - sqrTerm * sqrTerm) / (2 * g);
    }
    Float sinTheta = std::sqrt(std::max((Float)0, 1 - cosTheta * cosTheta));
    Float phi = 2 * Pi * u.y;
    V3 v1, v2;
    CoordinateSystem(wo, &v1, &v2);
    // SphericalDirection(sinTheta, cosTheta, phi, x, y, z), Geometry.h
    *wi = sinTheta * std::cos(phi) * v1 + sinTheta * std::sin(phi) * v2 + cosTheta * wo;
    return PhaseHG(cosTheta, g);
}

struct MediumInteraction {
    bool valid = false;
    V3 p, wo;
    int medium = -1;
    Float g = 0;
    Interaction AsInteraction() const {
        Interaction it;
        it.p = p; it.wo = wo; it.n = V3(); it.pError = V3();
        it.mediumInside = it.mediumOutside = medium;
        return it;
    }
};

// Transform::operator()(Ray) with the error-bounded origin shift, Transform.h:230-244
inline Ray XRay(const M44 &m, const Ray &r) {
    V3 oError;
    V3 o = XPointErr(m, r.o, &oError);
    V3 d = XVector(m, r.d);
    Float lengthSquared = d.LengthSquared();
    Float tMax = r.tMax;
    if (lengthSquared > 0) {
        Float dt = Dot(Abs(d), oError) / lengthSquared;
        o += d * dt;
        tMax -= dt;
    }
    return Ray(o, d, tMax, r.medium);
}

struct MediaSet {
    const Scene *scene = nullptr;
    struct Grid { M44 worldToMedium; Float sigma_t, invMaxDensity; const float *d; int nx, ny, nz; };
    std::vector<Grid> grids;  // per medium (valid for GRID)
    void Init(const Scene *s) {
        scene = s;
        grids.resize(s->media.size());
        for (size_t i = 0; i < s->media.size(); ++i) {
            const gnxr_medium &m = s->media[i];
            if (m.type != GNXR_MEDIUM_GRID) continue;
            Grid &g = grids[i];
            g.worldToMedium = Inverse(M44::FromRowMajor(m.medium_to_world));
            g.nx = m.nx; g.ny = m.ny; g.nz = m.nz;
            g.d = s->gridDensity.data() + m.density_offset;
            g.sigma_t = (Spec(m.sigma_a[0], m.sigma_a[1], m.sigma_a[2]) + Spec(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]))[0];
            Float maxDensity = 0;
            for (int k = 0; k < m.nx * m.ny * m.nz; ++k) maxDensity = std::max(maxDensity, g.d[k]);
            g.invMaxDensity = 1 / maxDensity;
        }
    }
    // GridDensityMedium::D / Density
    Float D(const Grid &g, int x, int y, int z) const {
        if (x < 0 || y < 0 || z < 0 || x >= g.nx || y >= g.ny || z >= g.nz) return 0;
        return g.d[(z * g.ny + y) * g.nx + x];
    }
    Float Density(const Grid &g, const V3 &p) const {
        V3 ps(p.x * g.nx - .5f, p.y * g.ny - .5f, p.z * g.nz - .5f);
        int px = (int)std::floor(ps.x), py = (int)std::floor(ps.y), pz = (int)std::floor(ps.z);
        V3 d = ps - V3((Float)px, (Float)py, (Float)pz);
        Float d00 = Lerp(d.x, D(g, px, py, pz), D(g, px + 1, py, pz));
        Float d10 = Lerp(d.x, D(g, px, py + 1, pz), D(g, px + 1, py + 1, pz));
        Float d01 = Lerp(d.x, D(g, px, py, pz + 1), D(g, px + 1, py, pz + 1));
        Float d11 = Lerp(d.x, D(g, px, py + 1, pz + 1), D(g, px + 1, py + 1, pz + 1));
        Float d0 = Lerp(d.y, d00, d10);
        Float d1 = Lerp(d.y, d01, d11);
        return Lerp(d.z, d0, d1);
    }
    Spec Tr(int mi, const Ray &ray, SampleStream &sampler) const {
        const gnxr_medium &m = scene->media[mi];
        if (m.type == GNXR_MEDIUM_HOMOGENEOUS) {  // HomogeneousMedium.cpp:11-15
            Spec sigma_t = Spec(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]) + Spec(m.sigma_a[0], m.sigma_a[1], m.sigma_a[2]);
            return Exp(-sigma_t * std::min(ray.tMax * ray.d.Length(), MaxFloat));
        }
        const Grid &g = grids[mi];  // GridDensityMedium.cpp:57-87 (ratio tracking)
        Ray r = XRay(g.worldToMedium, Ray(ray.o, Normalize(ray.d), ray.tMax * ray.d.Length()));
        Float tMin, tMax;
        if (!BoundsIntersectP(Bounds3(V3(0, 0, 0), V3(1, 1, 1)), r, &tMin, &tMax)) return Spec(1.f);
        Float Tr = 1, t = tMin;
        while (true) {
            t -= std::log(1 - sampler.Get1D()) * g.invMaxDensity / g.sigma_t;
            if (t >= tMax) break;
            Float density = Density(g, r(t));
            Tr *= 1 - std::max((Float)0, density * g.invMaxDensity);
            const Float rrThreshold = .1;
            if (Tr < rrThreshold) {
                Float q = std::max((Float).05, 1 - Tr);
                if (sampler.Get1D() < q) return Spec(0.f);
                Tr /= 1 - q;
            }
        }
        return Spec(Tr);
    }
    Spec Sample(int mi, const Ray &ray, SampleStream &sampler, MediumInteraction *out) const {
        const gnxr_medium &m = scene->media[mi];
        if (m.type == GNXR_MEDIUM_HOMOGENEOUS) {  // HomogeneousMedium.cpp:17-43
            Spec sigma_s(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]);
            Spec sigma_t = sigma_s + Spec(m.sigma_a[0], m.sigma_a[1], m.sigma_a[2]);
            int channel = std::min((int)(sampler.Get1D() * 3), 3 - 1);
            Float dist = -std::log(1 - sampler.Get1D()) / sigma_t[channel];
            Float t = std::min(dist / ray.d.Length(), ray.tMax);
            bool sampledMedium = t < ray.tMax;
            if (sampledMedium) { out->valid = true; out->p = ray(t); out->wo = -ray.d; out->medium = mi; out->g = m.g; }
            Spec Tr = Exp(-sigma_t * std::min(t, MaxFloat) * ray.d.Length());
            Spec density = sampledMedium ? (sigma_t * Tr) : Tr;
            Float pdf = 0;
            for (int i = 0; i < 3; ++i) pdf += density[i];
            pdf *= 1 / (Float)3;
            if (pdf == 0) pdf = 1;
            return sampledMedium ? (Tr * sigma_s / pdf) : (Tr / pdf);
        }
        const Grid &g = grids[mi];  // GridDensityMedium.cpp:31-55 (delta tracking)
        Ray r = XRay(g.worldToMedium, Ray(ray.o, Normalize(ray.d), ray.tMax * ray.d.Length()));
        Float tMin, tMax;
        if (!BoundsIntersectP(Bounds3(V3(0, 0, 0), V3(1, 1, 1)), r, &tMin, &tMax)) return Spec(1.f);
        Float t = tMin;
        while (true) {
            t -= std::log(1 - sampler.Get1D()) * g.invMaxDensity / g.sigma_t;
            if (t >= tMax) break;
            if (Density(g, r(t)) * g.invMaxDensity > sampler.Get1D()) {
                out->valid = true; out->p = ray(t); out->wo = -ray.d; out->medium = mi; out->g = m.g;
                return Spec(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]) / g.sigma_t;
            }
        }
        return Spec(1.f);
    }
};

struct VolContext {
    const RenderContext *rc;
    MediaSet media;
};

// VisibilityTester::Tr, core/Light.cpp:33-53
inline Spec VisibilityTr(const VolContext &vc, const Interaction &p0, const Interaction &p1, SampleStream &sampler) {
    const Scene &scene = *vc.rc->scene;
    Ray ray = p0.SpawnRayTo(p1);
    Spec Tr(1.f);
    while (true) {
        SurfaceInteraction isect;
        bool hitSurface = scene.Intersect(ray, &isect);
        if (hitSurface && scene.triMaterial[isect.prim] >= 0 && scene.materials[scene.triMaterial[isect.prim]].type != GNXR_MAT_NONE) return Spec(0.0f);
        if (ray.medium >= 0) Tr *= vc.media.Tr(ray.medium, ray, sampler);
        if (!hitSurface) break;
        ray = isect.SpawnRayTo(p1);
    }
    return Tr;
}
// Scene::IntersectTr, core/Scene.cpp:26-40
inline bool IntersectTr(const VolContext &vc, Ray ray, SampleStream &sampler, SurfaceInteraction *isect, Spec *Tr) {
    const Scene &scene = *vc.rc->scene;
    *Tr = Spec(1.f);
    while (true) {
        bool hitSurface = scene.Intersect(ray, isect);
        if (ray.medium >= 0) *Tr *= vc.media.Tr(ray.medium, ray, sampler);
        if (!hitSurface) return false;
        int mat = scene.triMaterial[isect->prim];
        if (mat >= 0 && scene.materials[mat].type != GNXR_MAT_NONE) return true;
        ray = isect->SpawnRay(ray.d);
    }
}

// EstimateDirect with handleMedia = true for a surface (bsdf != nullptr) or medium (mi != nullptr) interaction
inline Spec EstimateDirectMedia(const VolContext &vc, const Interaction &it, const SurfaceInteraction *isect, const BSDF *bsdf, const MediumInteraction *mi,
                                const P2 &uScattering, int light, const P2 &uLight, SampleStream &sampler) {
    const RenderContext &rc = *vc.rc;
    int bsdfFlags = BSDF_ALL & ~BSDF_SPECULAR;
    Spec Ld(0.f);
    Float lightPdf = 0, scatteringPdf = 0;
    LightSample ls = rc.Sample_Li(light, it, uLight);
    V3 wi = ls.wi;
    lightPdf = ls.pdf;
    Spec Li = ls.Li;
    if (lightPdf > 0 && !Li.IsBlack()) {
        Spec f;
        if (isect) {
            f = bsdf->f(isect->wo, wi, bsdfFlags) * AbsDot(wi, isect->sn);
            scatteringPdf = bsdf->Pdf(isect->wo, wi, bsdfFlags);
        } else {
            Float p = HG_p(mi->g, mi->wo, wi);
            f = Spec(p);
            scatteringPdf = p;
        }
        if (!f.IsBlack()) {
            Li *= VisibilityTr(vc, it, ls.p1, sampler);
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
        if (isect) {
            int sampledType = 0;
            f = bsdf->Sample_f(isect->wo, &wi, uScattering, &scatteringPdf, bsdfFlags, &sampledType);
            f *= AbsDot(wi, isect->sn);
            sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
        } else {
            Float p = HG_Sample_p(mi->g, mi->wo, &wi, uScattering);
            f = Spec(p);
            scatteringPdf = p;
        }
        if (!f.IsBlack() && scatteringPdf > 0) {
            Float weight = 1;
            if (!sampledSpecular) {
                lightPdf = rc.Pdf_Li(light, it, wi);
                if (lightPdf == 0) return Ld;
                weight = PowerHeuristic(1, scatteringPdf, 1, lightPdf);
            }
            SurfaceInteraction lightIsect;
            Ray ray = it.SpawnRay(wi);
            Spec Tr(1.f);
            bool found = IntersectTr(vc, ray, sampler, &lightIsect, &Tr);
            Spec Li2(0.f);
            if (found) {
                if (rc.scene->triLight[lightIsect.prim] == light) Li2 = rc.Le(lightIsect, -wi);
            } else
                Li2 = rc.LightLe(light, ray);
            if (!Li2.IsBlack()) Ld += f * Li2 * Tr * weight / scatteringPdf;
        }
    }
    return Ld;
}

inline Spec UniformSampleOneLightMedia(const VolContext &vc, const Interaction &it, const SurfaceInteraction *isect, const BSDF *bsdf, const MediumInteraction *mi,
                                       SampleStream &sampler, const Distribution1D *lightDistrib) {
    int nLights = (int)vc.rc->scene->lights.size();
    if (nLights == 0) return Spec(0.f);
    Float lightPdf;
    int lightNum = lightDistrib->SampleDiscrete(sampler.Get1D(), &lightPdf);
    if (lightPdf == 0) return Spec(0.f);
    P2 uLight = sampler.Get2D();
    P2 uScattering = sampler.Get2D();
    return EstimateDirectMedia(vc, it, isect, bsdf, mi, uScattering, lightNum, uLight, sampler) / lightPdf;
}

// integrators/VolPathIntegrator.cpp:24-159
inline Spec VolPathLi(const RenderContext &rc, const PathParams &pp, const Ray &r, SampleStream &sampler) {
    const Scene &scene = *rc.scene;
    const VolContext &vc = *static_cast<const VolContext *>(rc.mediaSet);
    Spec L(0.f), beta(1.f);
    Ray ray(r);
    bool specularBounce = false;
    int bounces;
    Float etaScale = 1;
    for (bounces = 0;; ++bounces) {
        SurfaceInteraction isect;
        bool foundIntersection = scene.Intersect(ray, &isect);
        MediumInteraction mi;
        if (ray.medium >= 0) beta *= vc.media.Sample(ray.medium, ray, sampler, &mi);
        if (beta.IsBlack()) break;
        if (mi.valid) {
            if (bounces >= pp.maxDepth) break;
            Interaction mit = mi.AsInteraction();
            const Distribution1D *lightDistrib = rc.Lookup(mi.p);
            L += beta * UniformSampleOneLightMedia(vc, mit, nullptr, nullptr, &mi, sampler, lightDistrib);
            V3 wo = -ray.d, wi;
            HG_Sample_p(mi.g, wo, &wi, sampler.Get2D());
            ray = mit.SpawnRay(wi);
            specularBounce = false;
        } else {
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
            const Distribution1D *lightDistrib = rc.Lookup(isect.p);
            L += beta * UniformSampleOneLightMedia(vc, isect, &isect, &bsdf, nullptr, sampler, lightDistrib);
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
        }
        Spec rrBeta = beta * etaScale;
        if (rrBeta.MaxComponentValue() < pp.rrThreshold && bounces > 3) {
            Float q = std::max((Float).05, 1 - rrBeta.MaxComponentValue());
            if (sampler.Get1D() < q) break;
            beta /= 1 - q;
        }
    }
    return L;
}

// Ray differentials of the specular children, SamplerIntegrator::SpecularReflect / SpecularTransmit (core/Integrator.cpp:335-354,
// 376-436).  shading.dndu / dndv are zero unless the triangle has per-vertex normals.
inline void ReflectDifferentials(const Ray &ray, const SurfaceInteraction &isect, const V3 &wo, const V3 &wi, Ray *rd) {
    if (!ray.hasDifferentials) return;
    const V3 ns = isect.sn, dndu = isect.dndu, dndv = isect.dndv;
    rd->hasDifferentials = true;
    rd->rxOrigin = isect.p + isect.dpdx;
    rd->ryOrigin = isect.p + isect.dpdy;
    V3 dndx = dndu * isect.dudx + dndv * isect.dvdx;
    V3 dndy = dndu * isect.dudy + dndv * isect.dvdy;
    V3 dwodx = -ray.rxDirection - wo, dwody = -ray.ryDirection - wo;
    Float dDNdx = Dot(dwodx, ns) + Dot(wo, dndx);
    Float dDNdy = Dot(dwody, ns) + Dot(wo, dndy);
    rd->rxDirection = wi - dwodx + 2.f * V3(Dot(wo, ns) * dndx + dDNdx * ns);
    rd->ryDirection = wi - dwody + 2.f * V3(Dot(wo, ns) * dndy + dDNdy * ns);
}
inline void TransmitDifferentials(const Ray &ray, const SurfaceInteraction &isect, Float bsdfEta, const V3 &wo, const V3 &wi, Ray *rd) {
    if (!ray.hasDifferentials) return;
    V3 ns = isect.sn;
    const V3 dndu = isect.dndu, dndv = isect.dndv;
    rd->hasDifferentials = true;
    rd->rxOrigin = isect.p + isect.dpdx;
    rd->ryOrigin = isect.p + isect.dpdy;
    V3 dndx = dndu * isect.dudx + dndv * isect.dvdx;
    V3 dndy = dndu * isect.dudy + dndv * isect.dvdy;
    Float eta = 1 / bsdfEta;
    if (Dot(wo, ns) < 0) {
        eta = 1 / eta;
        ns = -ns;
        dndx = -dndx;
        dndy = -dndy;
    }
    V3 dwodx = -ray.rxDirection - wo, dwody = -ray.ryDirection - wo;
    Float dDNdx = Dot(dwodx, ns) + Dot(wo, dndx);
    Float dDNdy = Dot(dwody, ns) + Dot(wo, dndy);
    Float mu = eta * Dot(wo, ns) - AbsDot(wi, ns);
    Float dmudx = (eta - (eta * eta * Dot(wo, ns)) / AbsDot(wi, ns)) * dDNdx;
    Float dmudy = (eta - (eta * eta * Dot(wo, ns)) / AbsDot(wi, ns)) * dDNdy;
    rd->rxDirection = wi - eta * dwodx + V3(mu * dndx + dmudx * ns);
    rd->ryDirection = wi - eta * dwody + V3(mu * dndy + dmudy * ns);
}

// integrators/WhittedIntegrator.cpp:14-68 + SamplerIntegrator::SpecularReflect / SpecularTransmit, core/Integrator.cpp:321-442
// (BASELINE config 1: the reference's CPU-only path).
// The depth-first recursion consumes the sample stream in DFS order: all lights of a vertex, then the reflected subtree,
// then the transmitted subtree.
inline Spec WhittedLi(const RenderContext &rc, const PathParams &pp, const Ray &ray, SampleStream &sampler, int depth) {
    const Scene &scene = *rc.scene;
    Spec L(0.f);
    SurfaceInteraction isect;
    if (!scene.Intersect(ray, &isect)) {
        for (int light = 0; light < (int)scene.lights.size(); ++light) L += rc.LightLe(light, ray);
        return L;
    }
    V3 wo = isect.wo;
    BSDF bsdf;
    if (!ComputeScatteringFunctions(scene, ray, &isect, false, &bsdf)) return WhittedLi(rc, pp, isect.SpawnRay(ray.d), sampler, depth);
    const V3 n = isect.sn;   // `const Normal3f &n = isect.shading.n` is read after Bump ran
    L += rc.Le(isect, wo);
    Spec lightL(0.f);
    for (int light = 0; light < (int)scene.lights.size(); ++light) {
        LightSample ls = rc.Sample_Li(light, isect, sampler.Get2D());
        if (ls.Li.IsBlack() || ls.pdf == 0) continue;
        Spec f = bsdf.f(wo, ls.wi, BSDF_ALL);
        if (!f.IsBlack() && !scene.IntersectP(isect.SpawnRayTo(ls.p1))) lightL += f * ls.Li * AbsDot(ls.wi, n) / ls.pdf;
    }
    L += lightL;
    if (depth + 1 < pp.maxDepth) {
        {   // SpecularReflect
            V3 wi;
            Float pdf;
            int sampledType = 0;
            Spec f = bsdf.Sample_f(wo, &wi, sampler.Get2D(), &pdf, BSDF_REFLECTION | BSDF_SPECULAR, &sampledType);
            const V3 ns = isect.sn;
            if (pdf > 0.f && !f.IsBlack() && AbsDot(wi, ns) != 0.f) {
                Ray rd = isect.SpawnRay(wi);
                ReflectDifferentials(ray, isect, wo, wi, &rd);
                L += f * WhittedLi(rc, pp, rd, sampler, depth + 1) * AbsDot(wi, ns) / pdf;
            } else L += Spec(0.f);
        }
        {   // SpecularTransmit
            V3 wi;
            Float pdf;
            int sampledType = 0;
            Spec f = bsdf.Sample_f(wo, &wi, sampler.Get2D(), &pdf, BSDF_TRANSMISSION | BSDF_SPECULAR, &sampledType);
            const V3 ns = isect.sn;
            Spec Lt(0.f);
            if (pdf > 0.f && !f.IsBlack() && AbsDot(wi, ns) != 0.f) {
                Ray rd = isect.SpawnRay(wi);
                TransmitDifferentials(ray, isect, bsdf.eta, wo, wi, &rd);
                Lt = f * WhittedLi(rc, pp, rd, sampler, depth + 1) * AbsDot(wi, ns) / pdf;
            }
            L += Lt;
        }
    }
    return L;
}

// integrators/DirectLightingIntegrator.cpp:11-67 with UniformSampleAllLights (core/Integrator.cpp:25-55) and the specular
// recursion of SamplerIntegrator (core/Integrator.cpp:321-442).  No scene of the reference instantiates this integrator; it
// is the second half of SURVEY 8(f).1.
struct DirectParams {
    int maxDepth = 5;
    int strategy = 0;                  // enum class LightStrategy { UniformSampleAll, UniformSampleOne }
    std::vector<int> nLightSamples;    // Preprocess: sampler.RoundCount(light->nSamples) (Halton: identity)
    std::vector<int> arraySizes;       // Preprocess: maxDepth x lights x {Request2DArray(n), Request2DArray(n)}
    void Preprocess(const Scene &scene) {
        nLightSamples.clear(); arraySizes.clear();
        if (strategy != 0) return;
        for (const gnxr_light &l : scene.lights) nLightSamples.push_back(std::max(1, l.n_samples));
        for (int i = 0; i < maxDepth; ++i)
            for (size_t j = 0; j < scene.lights.size(); ++j) { arraySizes.push_back(nLightSamples[j]); arraySizes.push_back(nLightSamples[j]); }
    }
};
// core/Integrator.cpp:25-55 (handleMedia = false)
inline Spec UniformSampleAllLights(const RenderContext &rc, const SurfaceInteraction &isect, const BSDF &bsdf, SampleStream &sampler,
                                   const std::vector<int> &nLightSamples) {
    Spec L(0.f);
    std::vector<P2> uLightArray, uScatteringArray;
    for (size_t j = 0; j < rc.scene->lights.size(); ++j) {
        int nSamples = nLightSamples[j];
        bool haveLight = sampler.Get2DArray(nSamples, &uLightArray);
        bool haveScattering = sampler.Get2DArray(nSamples, &uScatteringArray);
        if (!haveLight || !haveScattering) {
            P2 uLight = sampler.Get2D();
            P2 uScattering = sampler.Get2D();
            L += EstimateDirect(rc, isect, bsdf, uScattering, (int)j, uLight);
        } else {
            Spec Ld(0.f);
            for (int k = 0; k < nSamples; ++k) Ld += EstimateDirect(rc, isect, bsdf, uScatteringArray[k], (int)j, uLightArray[k]);
            L += Ld / (Float)nSamples;
        }
    }
    return L;
}
inline Spec DirectLi(const RenderContext &rc, const DirectParams &dp, const Ray &ray, SampleStream &sampler, int depth) {
    const Scene &scene = *rc.scene;
    Spec L(0.f);
    SurfaceInteraction isect;
    if (!scene.Intersect(ray, &isect)) {
        for (int light = 0; light < (int)scene.lights.size(); ++light) L += rc.LightLe(light, ray);
        return L;
    }
    BSDF bsdf;
    if (!ComputeScatteringFunctions(scene, ray, &isect, false, &bsdf)) return DirectLi(rc, dp, isect.SpawnRay(ray.d), sampler, depth);
    V3 wo = isect.wo;
    L += rc.Le(isect, wo);
    if (scene.lights.size() > 0) {
        if (dp.strategy == 0) L += UniformSampleAllLights(rc, isect, bsdf, sampler, dp.nLightSamples);
        else L += UniformSampleOneLight(rc, isect, bsdf, sampler, nullptr);
    }
    if (depth + 1 < dp.maxDepth) {
        const V3 ns = isect.sn;
        {   // SpecularReflect, core/Integrator.cpp:321-371
            V3 wi;
            Float pdf;
            int sampledType = 0;
            Spec f = bsdf.Sample_f(wo, &wi, sampler.Get2D(), &pdf, BSDF_REFLECTION | BSDF_SPECULAR, &sampledType);
            if (pdf > 0.f && !f.IsBlack() && AbsDot(wi, ns) != 0.f) {
                Ray rd = isect.SpawnRay(wi);
                ReflectDifferentials(ray, isect, wo, wi, &rd);
                L += f * DirectLi(rc, dp, rd, sampler, depth + 1) * AbsDot(wi, ns) / pdf;
            } else L += Spec(0.f);
        }
        {   // SpecularTransmit, core/Integrator.cpp:373-442
            V3 wi;
            Float pdf;
            int sampledType = 0;
            Spec f = bsdf.Sample_f(wo, &wi, sampler.Get2D(), &pdf, BSDF_TRANSMISSION | BSDF_SPECULAR, &sampledType);
            Spec Lt(0.f);
            if (pdf > 0.f && !f.IsBlack() && AbsDot(wi, ns) != 0.f) {
                Ray rd = isect.SpawnRay(wi);
                TransmitDifferentials(ray, isect, bsdf.eta, wo, wi, &rd);
                Lt = f * DirectLi(rc, dp, rd, sampler, depth + 1) * AbsDot(wi, ns) / pdf;
            }
            L += Lt;
        }
    }
    return L;
}

}  // namespace gnxo

// ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_envlight.h: CPU restatement of the infinite lights:
//   InfiniteAreaLight ctor / Le / Sample_Li / Pdf_Li / Power   lights/InfiniteAreaLight.cpp:12-132
//   MIPMap ctor (Lanczos resample to pow2, pyramid) / Lookup / triangle / Texel   core/MIPMap.h:41-256
//   Lanczos                                                    core/Texture.cpp:152-161
//   SkyBoxLight::Le / Sample_Li (image load failed)            lights/SkyBoxLight.cpp:43-85
#pragma once
#include "o_scene.h"   // o_texture.h (MIPMap) comes with it

namespace gnxo {

}  // namespace gnxo

#include "o_lightsample.h"

namespace gnxo {

struct InfiniteAreaLight {
    M44 lightToWorld, worldToLight;
    MIPMapRGB Lmap;
    Distribution2D distribution;
    V3 worldCenter;
    Float worldRadius = 0;
    // InfiniteAreaLight.cpp:12-82.  env_rgb is the decoded .hdr (what stbi_loadf returns); texels =
    // (L*rgb)^1.5 (`r * Sqrt(r)`, :41).
    // flipY: SkyBoxLight::loadImage calls stbi_set_flip_vertically_on_load(true) (SkyBoxLight.cpp:19), a
    // process-wide stb_image switch, so an InfiniteAreaLight constructed AFTER a SkyBoxLight (the order of
    // ui/RenderThread.cpp:145-151) decodes its map upside down.
    InfiniteAreaLight(const gnxr_light &l, const float *rgb, int w, int h, bool flipY = false) {
        lightToWorld = M44::FromRowMajor(l.light_to_world);
        worldToLight = Inverse(lightToWorld);
        std::vector<Spec> texels;
        int rx = w, ry = h;
        if (rgb && w > 0 && h > 0) {
            texels.resize((size_t)w * h);
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++) {
                    Spec r;
                    int js = flipY ? (h - 1 - j) : j;
                    r[0] = l.le[0] * rgb[(i + js * w) * 3 + 0];
                    r[1] = l.le[1] * rgb[(i + js * w) * 3 + 1];
                    r[2] = l.le[2] * rgb[(i + js * w) * 3 + 2];
                    texels[i + j * w] = r * Sqrt(r);
                }
        } else {
            rx = ry = 1;
            texels.assign(1, Spec(l.le[0], l.le[1], l.le[2]));
        }
        Lmap.Build(rx, ry, texels.data());
        int width = 2 * Lmap.resX, height = 2 * Lmap.resY;
        std::vector<Float> img((size_t)width * height);
        float fwidth = 0.5f / std::min(width, height);
        for (int v = 0; v < height; v++) {
            Float vp = (v + .5f) / (Float)height;
            Float sinTheta = std::sin(Pi * (v + .5f) / height);
            for (int u = 0; u < width; ++u) {
                Float up = (u + .5f) / (Float)width;
                img[u + v * width] = Lmap.Lookup(P2(up, vp), fwidth).y();
                img[u + v * width] *= sinTheta;
            }
        }
        distribution = Distribution2D(img.data(), width, height);
    }
    // InfiniteAreaLight.h:23-26, Geometry.h:770-773
    void Preprocess(const Bounds3 &wb) {
        worldCenter = (wb.pMin + wb.pMax) / 2;
        bool inside = worldCenter.x >= wb.pMin.x && worldCenter.x <= wb.pMax.x && worldCenter.y >= wb.pMin.y &&
                      worldCenter.y <= wb.pMax.y && worldCenter.z >= wb.pMin.z && worldCenter.z <= wb.pMax.z;
        worldRadius = inside ? (worldCenter - wb.pMax).Length() : 0;
    }
    Spec Power() const { return Pi * worldRadius * worldRadius * Lmap.Lookup(P2(.5f, .5f), .5f); }
    Spec Le(const Ray &ray) const {
        V3 w = Normalize(XVector(worldToLight, ray.d));
        P2 st(SphericalPhi(w) * Inv2Pi, SphericalTheta(w) * InvPi);
        return Lmap.Lookup(st);
    }
    LightSample Sample_Li(const Interaction &ref, const P2 &u) const {
        LightSample s;
        Float mapPdf;
        P2 uv = distribution.SampleContinuous(u, &mapPdf);
        if (mapPdf == 0) { s.Li = Spec(0.f); s.pdf = 0; return s; }  // *pdf left untouched (0) in the reference
        Float theta = uv.y * Pi, phi = uv.x * 2 * Pi;
        Float cosTheta = std::cos(theta), sinTheta = std::sin(theta);
        Float sinPhi = std::sin(phi), cosPhi = std::cos(phi);
        s.wi = XVector(lightToWorld, V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
        s.pdf = mapPdf / (2 * Pi * Pi * sinTheta);
        if (sinTheta == 0) s.pdf = 0;
        s.p1 = Interaction();
        s.p1.p = ref.p + s.wi * (2 * worldRadius);
        s.Li = Lmap.Lookup(uv);
        return s;
    }
    Float Pdf_Li(const Interaction &, const V3 &w) const {
        V3 wi = XVector(worldToLight, w);
        Float theta = SphericalTheta(wi), phi = SphericalPhi(wi);
        Float sinTheta = std::sin(theta);
        if (sinTheta == 0) return 0;
        return distribution.Pdf(P2(phi * Inv2Pi, theta * InvPi)) / (2 * Pi * Pi * sinTheta);
    }
};

// SkyBoxLight::Le with data == nullptr, SkyBoxLight.cpp:55-85.  The double/float mix follows the source.
inline Spec SkyBoxLe(const gnxr_light &l, const Ray &ray) {
    V3 worldCenter(l.center[0], l.center[1], l.center[2]);
    float worldRadius = l.radius;
    V3 oc = ray.o - worldCenter;
    float a = Dot(ray.d, ray.d);
    float b = 2.0 * Dot(oc, ray.d);
    float c = Dot(oc, oc) - worldRadius * worldRadius;
    float discriminant = b * b - 4 * a * c;
    float t;
    if (discriminant < 0) return Spec(0.f);
    // SkyBoxLight.cpp pulls in <math.h> through 3rd/stb_image.h, so the unqualified sqrt(float) binds to the float
    // overload there (unlike MicroFacet.cpp / DisneyMaterial.cpp); the division by `2.0 * a` is a double expression.
    t = (-b + std::sqrt(discriminant)) / (2.0 * a);
    V3 hitPos = ray.o + t * ray.d;
    V3 hitPos_temp = hitPos - worldCenter;
    Spec Col;
    Col[0] = (hitPos_temp.x + worldRadius) / (2.f * worldRadius);
    Col[1] = (hitPos_temp.y + worldRadius) / (2.f * worldRadius);
    Col[2] = (hitPos_temp.z + worldRadius) / (2.f * worldRadius);
    return Col;
}
// SkyBoxLight::Sample_Li, SkyBoxLight.cpp:43-53 (no image: value 0, pdf 1/4pi; LightToWorld = identity)
inline LightSample SkyBoxSample_Li(const gnxr_light &l, const Interaction &ref, const P2 &u) {
    LightSample s;
    float theta = u.y * Pi, phi = u.x * 2 * Pi;
    float cosTheta = std::cos(theta), sinTheta = std::sin(theta);
    float sinPhi = std::sin(phi), cosPhi = std::cos(phi);
    s.wi = V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
    s.pdf = 1.f / (4 * Pi);
    s.p1 = Interaction();
    s.p1.p = ref.p + s.wi * (2 * l.radius);
    s.Li = Spec(0.f);
    return s;
}

}  // namespace gnxo

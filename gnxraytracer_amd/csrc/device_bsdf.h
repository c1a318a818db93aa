// device_bsdf.h -- BxDF evaluation / sampling on the device.
//   BxDFs, Fresnel, BSDF::f / Pdf / Sample_f     core/Reflection.{h,cpp}
//   TrowbridgeReitzDistribution                  core/MicroFacet.cpp:129-136,150-159,215-324
//   Disney lobes                                 materials/DisneyMaterial.cpp:50-342
// Lobe parameter blocks (DLobe) are precomputed per material by the host scene compiler; a hit only
// adds the shading frame.  Lobes are visited in the order the reference's <Material>::
// ComputeScatteringFunctions adds them, so sums associate identically.
#pragma once
#include "device_sampler.h"

namespace gnxr {

GX_DEV float cos_theta(V3 w) { return w.z; }
GX_DEV float cos2_theta(V3 w) { return w.z * w.z; }
GX_DEV float abs_cos_theta(V3 w) { return fabsf(w.z); }
GX_DEV float sin2_theta(V3 w) { return fmaxf(0.f, 1.f - cos2_theta(w)); }
GX_DEV float sin_theta(V3 w) { return gx_sqrt(sin2_theta(w)); }
GX_DEV float tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
GX_DEV float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
GX_DEV float cos_phi(V3 w) { float s = sin_theta(w); return (s == 0) ? 1 : clampf(w.x / s, -1, 1); }
GX_DEV float sin_phi(V3 w) { float s = sin_theta(w); return (s == 0) ? 0 : clampf(w.y / s, -1, 1); }
GX_DEV float cos2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
GX_DEV float sin2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
GX_DEV bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0; }
GX_DEV V3 reflect(V3 wo, V3 n) { return -wo + 2 * dot(wo, n) * n; }
GX_DEV bool refract(V3 wi, V3 n, float eta, V3 *wt) {  // Reflection.h:68-80
    float cosThetaI = dot(n, wi);
    float sin2ThetaI = fmaxf(0.f, 1 - cosThetaI * cosThetaI);
    float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    float cosThetaT = gx_sqrt(1 - sin2ThetaT);
    *wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}

GX_DEV float fr_dielectric(float cosThetaI, float etaI, float etaT) {  // Reflection.cpp:16-38
    cosThetaI = clampf(cosThetaI, -1, 1);
    bool entering = cosThetaI > 0.f;
    if (!entering) { float t = etaI; etaI = etaT; etaT = t; cosThetaI = fabsf(cosThetaI); }
    float sinThetaI = gx_sqrt(fmaxf(0.f, 1 - cosThetaI * cosThetaI));
    float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    float cosThetaT = gx_sqrt(fmaxf(0.f, 1 - sinThetaT * sinThetaT));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
GX_DEV Spec fr_conductor(float cosThetaI, Spec etat, Spec k) {  // Reflection.cpp:41-64 with etai == 1
    cosThetaI = clampf(cosThetaI, -1, 1);
    Spec etai(1.f);
    Spec eta = etat / etai;
    Spec etak = k / etai;
    float cosThetaI2 = cosThetaI * cosThetaI;
    float sinThetaI2 = (float)(1. - (double)cosThetaI2);  // `1. - cosThetaI2` is a double expression
    Spec eta2 = eta * eta;
    Spec etak2 = etak * etak;
    Spec t0 = eta2 - etak2 - Spec(sinThetaI2);
    Spec a2plusb2 = ssqrt(t0 * t0 + 4 * eta2 * etak2);
    Spec t1 = a2plusb2 + Spec(cosThetaI2);
    Spec a = ssqrt(0.5f * (a2plusb2 + t0));
    Spec t2 = 2.f * cosThetaI * a;
    Spec Rs = (t1 - t2) / (t1 + t2);
    Spec t3 = cosThetaI2 * a2plusb2 + Spec(sinThetaI2 * sinThetaI2);
    Spec t4 = t2 * sinThetaI2;
    Spec Rp = Rs * (t3 - t4) / (t3 + t4);
    return 0.5f * (Rp + Rs);
}
GX_DEV float schlick_weight(float cosTheta) { float m = clampf(1 - cosTheta, 0, 1); return (m * m) * (m * m) * m; }
GX_DEV float fr_schlick(float R0, float cosTheta) { return lerpf(schlick_weight(cosTheta), R0, 1); }
GX_DEV Spec fr_schlick_spec(Spec R0, float cosTheta) { return slerp(schlick_weight(cosTheta), R0, Spec(1.f)); }
GX_DEV float gtr1(float cosTheta, float alpha) {  // DisneyMaterial.cpp:224-229
    float alpha2 = alpha * alpha;
    return (alpha2 - 1) / (GX_PI * gx_log(alpha2) * (1 + (alpha2 - 1) * cosTheta * cosTheta));
}
GX_DEV float smith_g_ggx(float cosTheta, float alpha) {  // DisneyMaterial.cpp:232-237, unqualified sqrt -> double
    float alpha2 = alpha * alpha;
    float cosTheta2 = cosTheta * cosTheta;
    return (float)(1 / ((double)cosTheta + sqrt((double)(alpha2 + cosTheta2 - alpha2 * cosTheta2))));
}

// LM: compile-time mask of the lobe kinds (bit LobeKind) and Fresnel kinds (bit 16 + FresnelKind) a shade-kernel
// specialisation can meet; everything else is compiled out, which is what keeps the Matte kernel small.
#define GX_HAS_LOBE(K) ((LM >> (K)) & 1u)
#define GX_HAS_FRESNEL(K) ((LM >> (16 + (K))) & 1u)
constexpr uint32_t lobe_bit(int k) { return 1u << k; }
constexpr uint32_t fresnel_bit(int k) { return 1u << (16 + k); }
constexpr uint32_t LM_DIFFUSE = lobe_bit(LOBE_LAMBERT) | lobe_bit(LOBE_OREN);
constexpr uint32_t LM_GLOSSY = LM_DIFFUSE | lobe_bit(LOBE_SPEC_REFL) | lobe_bit(LOBE_SPEC_TRANS) | lobe_bit(LOBE_FRESNEL_SPEC) | lobe_bit(LOBE_MICRO_REFL) |
                               lobe_bit(LOBE_MICRO_TRANS) | fresnel_bit(FRESNEL_NOOP) | fresnel_bit(FRESNEL_DIELECTRIC) | fresnel_bit(FRESNEL_CONDUCTOR);
constexpr uint32_t LM_ALL = 0xffffffffu;

template <uint32_t LM>
GX_DEV Spec fresnel_eval(const DLobe &l, float cosI) {
    if (GX_HAS_FRESNEL(FRESNEL_DIELECTRIC) && l.fresnel == FRESNEL_DIELECTRIC) return Spec(fr_dielectric(cosI, l.f_etaI, l.f_etaT));
    if (GX_HAS_FRESNEL(FRESNEL_CONDUCTOR) && l.fresnel == FRESNEL_CONDUCTOR) return fr_conductor(fabsf(cosI), spec3(l.f_cEtaT), spec3(l.f_cK));
    if (GX_HAS_FRESNEL(FRESNEL_DISNEY) && l.fresnel == FRESNEL_DISNEY)
        return slerp(l.f_metallic, Spec(fr_dielectric(cosI, 1, l.f_eta)), fr_schlick_spec(spec3(l.f_R0), cosI));
    return Spec(1.f);
}

// ---- Trowbridge-Reitz, MicroFacet.cpp ----
GX_DEV float tr_D(float ax, float ay, V3 wh) {
    float tan2Theta = tan2_theta(wh);
    if (isinf(tan2Theta)) return 0.f;
    const float cos4Theta = cos2_theta(wh) * cos2_theta(wh);
    float e = (cos2_phi(wh) / (ax * ax) + sin2_phi(wh) / (ay * ay)) * tan2Theta;
    return 1 / (GX_PI * ax * ay * cos4Theta * (1 + e) * (1 + e));
}
GX_DEV float tr_lambda(float ax, float ay, V3 w) {
    float absTanTheta = fabsf(tan_theta(w));
    if (isinf(absTanTheta)) return 0.f;
    float alpha = gx_sqrt(cos2_phi(w) * ax * ax + sin2_phi(w) * ay * ay);
    float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + gx_sqrt(1.f + alpha2Tan2Theta)) / 2;
}
GX_DEV float tr_G1(float ax, float ay, V3 w) { return 1 / (1 + tr_lambda(ax, ay, w)); }
GX_DEV float tr_G(const DLobe &l, V3 wo, V3 wi) {
    if (l.disney_g) return tr_G1(l.alphax, l.alphay, wo) * tr_G1(l.alphax, l.alphay, wi);  // DisneyMaterial.cpp:338-342
    return 1 / (1 + tr_lambda(l.alphax, l.alphay, wo) + tr_lambda(l.alphax, l.alphay, wi));
}
GX_DEV float tr_pdf(float ax, float ay, V3 wo, V3 wh) {  // sampleVisibleArea, MicroFacet.cpp:318-324
    return tr_D(ax, ay, wh) * tr_G1(ax, ay, wo) * absdot(wo, wh) / abs_cos_theta(wo);
}
GX_DEV void tr_sample11(float cosTheta, float U1, float U2, float *slope_x, float *slope_y) {  // MicroFacet.cpp:215-260
    if (cosTheta > .9999) {
        // unqualified sqrt/cos/sin bind to the double versions in the reference (MicroFacet.cpp:220-223)
        float r = (float)sqrt((double)(U1 / (1 - U1)));
        float phi = (float)(6.28318530718 * (double)U2);
        const double sphi = gx_sin_d((double)phi), cphi = gx_cos_d((double)phi);   // glibc's __sin / __cos, restated (device_math.h)
        *slope_x = (float)((double)r * cphi);
        *slope_y = (float)((double)r * sphi);
        return;
    }
    float sinTheta = gx_sqrt(fmaxf(0.f, 1.f - cosTheta * cosTheta));
    float tanTheta = sinTheta / cosTheta;
    float a = 1 / tanTheta;
    float G1 = 2 / (1 + gx_sqrt(1.f + 1.f / (a * a)));
    float A = 2 * U1 / G1 - 1;
    float tmp = 1.f / (A * A - 1.f);
    if (tmp > 1e10f) tmp = 1e10f;
    float B = tanTheta;
    float D = gx_sqrt(fmaxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.f));
    float slope_x_1 = B * tmp - D;
    float slope_x_2 = B * tmp + D;
    *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
    float S;
    if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
    else { S = -1.f; U2 = 2.f * (.5f - U2); }
    float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    *slope_y = S * z * gx_sqrt(1.f + *slope_x * *slope_x);
}
GX_DEV V3 tr_sample_wh(float ax, float ay, V3 wo, float u0, float u1) {  // MicroFacet.cpp:262-316
    bool flip = wo.z < 0;
    V3 wi = flip ? -wo : wo;
    V3 wiStretched = normalize(V3(ax * wi.x, ay * wi.y, wi.z));
    float slope_x, slope_y;
    tr_sample11(cos_theta(wiStretched), u0, u1, &slope_x, &slope_y);
    float tmp = cos_phi(wiStretched) * slope_x - sin_phi(wiStretched) * slope_y;
    slope_y = sin_phi(wiStretched) * slope_x + cos_phi(wiStretched) * slope_y;
    slope_x = tmp;
    slope_x = ax * slope_x;
    slope_y = ay * slope_y;
    V3 wh = normalize(V3(-slope_x, -slope_y, 1.f));
    if (flip) wh = -wh;
    return wh;
}

// ---- per-lobe f / Pdf / Sample_f (local shading space) ----
template <uint32_t LM>
GX_DEV Spec lobe_f(const DLobe &l, V3 wo, V3 wi) {
    switch (l.kind) {
    case LOBE_LAMBERT: if (GX_HAS_LOBE(LOBE_LAMBERT)) return spec3(l.R) * GX_INV_PI; break;
    case LOBE_LAMBERT_TRANS: if (GX_HAS_LOBE(LOBE_LAMBERT_TRANS)) return spec3(l.T) * GX_INV_PI; break;
    case LOBE_OREN: if (GX_HAS_LOBE(LOBE_OREN)) {  // Reflection.cpp:173-198
        float sinThetaI = sin_theta(wi), sinThetaO = sin_theta(wo);
        float maxCos = 0;
        if (sinThetaI > 1e-4 && sinThetaO > 1e-4) {
            float sinPhiI = sin_phi(wi), cosPhiI = cos_phi(wi);
            float sinPhiO = sin_phi(wo), cosPhiO = cos_phi(wo);
            float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
            maxCos = fmaxf(0.f, dCos);
        }
        // (one division for both branches of Reflection.cpp:189-195)
        const bool iBig = abs_cos_theta(wi) > abs_cos_theta(wo);
        const float sinAlpha = iBig ? sinThetaO : sinThetaI;
        const float tanBeta = (iBig ? sinThetaI : sinThetaO) / (iBig ? abs_cos_theta(wi) : abs_cos_theta(wo));
        return spec3(l.R) * GX_INV_PI * (l.A + l.B * maxCos * sinAlpha * tanBeta);
    }
    break;
    // MicrofacetReflection::f (Reflection.cpp:223-237) and MicrofacetTransmission::f (:278-302, TransportMode::Radiance): the
    // half vector and the Fresnel argument are lobe-specific, D, G and F are evaluated once for whichever lobe a lane holds
    // (rough glass puts both lobes into one wave).
    case LOBE_MICRO_REFL: case LOBE_MICRO_TRANS: if (GX_HAS_LOBE(LOBE_MICRO_REFL) || GX_HAS_LOBE(LOBE_MICRO_TRANS)) {
        const bool refl = l.kind == LOBE_MICRO_REFL;
        float cosThetaO, cosThetaI, eta = 1, cosArg;
        V3 wh;
        if (refl) {
            cosThetaO = abs_cos_theta(wo); cosThetaI = abs_cos_theta(wi);
            wh = wi + wo;
            if (cosThetaI == 0 || cosThetaO == 0) return Spec(0.f);
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.f);
            wh = normalize(wh);
            cosArg = dot(wi, faceforward(wh, V3(0, 0, 1)));
        } else {
            if (same_hemisphere(wo, wi)) return Spec(0.f);
            cosThetaO = cos_theta(wo); cosThetaI = cos_theta(wi);
            if (cosThetaI == 0 || cosThetaO == 0) return Spec(0.f);
            { const bool up = cos_theta(wo) > 0; eta = (up ? l.etaB : l.etaA) / (up ? l.etaA : l.etaB); }   // one division for both sides
            wh = normalize(wo + wi * eta);
            if (wh.z < 0) wh = -wh;
            if (dot(wo, wh) * dot(wi, wh) > 0) return Spec(0.f);
            cosArg = dot(wo, wh);
        }
        const Spec F = fresnel_eval<LM>(l, cosArg);
        const float D = tr_D(l.alphax, l.alphay, wh), G = tr_G(l, wo, wi);
        if (refl) return spec3(l.R) * D * G * F / (4 * cosThetaI * cosThetaO);
        float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
        float factor = (1 / eta);
        return (Spec(1.f) - F) * spec3(l.T) *
               fabsf(D * G * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor / (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
    }
    break;
    case LOBE_DISNEY_DIFFUSE: if (GX_HAS_LOBE(LOBE_DISNEY_DIFFUSE)) {  // DisneyMaterial.cpp:64-72
        float Fo = schlick_weight(abs_cos_theta(wo)), Fi = schlick_weight(abs_cos_theta(wi));
        return spec3(l.R) * GX_INV_PI * (1 - Fo / 2) * (1 - Fi / 2);
    }
    break;
    case LOBE_DISNEY_FAKESS: if (GX_HAS_LOBE(LOBE_DISNEY_FAKESS)) {  // DisneyMaterial.cpp:105-122
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.f);
        wh = normalize(wh);
        float cosThetaD = dot(wi, wh);
        float Fss90 = cosThetaD * cosThetaD * l.roughness;
        float Fo = schlick_weight(abs_cos_theta(wo)), Fi = schlick_weight(abs_cos_theta(wi));
        float Fss = lerpf(Fo, 1.0f, Fss90) * lerpf(Fi, 1.0f, Fss90);
        float ss = 1.25f * (Fss * (1 / (abs_cos_theta(wo) + abs_cos_theta(wi)) - .5f) + .5f);
        return spec3(l.R) * GX_INV_PI * ss;
    }
    break;
    case LOBE_DISNEY_RETRO: if (GX_HAS_LOBE(LOBE_DISNEY_RETRO)) {  // DisneyMaterial.cpp:151-164
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.f);
        wh = normalize(wh);
        float cosThetaD = dot(wi, wh);
        float Fo = schlick_weight(abs_cos_theta(wo)), Fi = schlick_weight(abs_cos_theta(wi));
        float Rr = 2 * l.roughness * cosThetaD * cosThetaD;
        return spec3(l.R) * GX_INV_PI * Rr * (Fo + Fi + Fo * Fi * (Rr - 1));
    }
    break;
    case LOBE_DISNEY_SHEEN: if (GX_HAS_LOBE(LOBE_DISNEY_SHEEN)) {  // DisneyMaterial.cpp:189-197
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.f);
        wh = normalize(wh);
        return spec3(l.R) * schlick_weight(dot(wi, wh));
    }
    break;
    case LOBE_DISNEY_CLEARCOAT: if (GX_HAS_LOBE(LOBE_DISNEY_CLEARCOAT)) {  // DisneyMaterial.cpp:239-253
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.f);
        wh = normalize(wh);
        float Dr = gtr1(abs_cos_theta(wh), l.gloss);
        float Fr = fr_schlick(.04f, dot(wo, wh));
        float Gr = smith_g_ggx(abs_cos_theta(wo), .25f) * smith_g_ggx(abs_cos_theta(wi), .25f);
        return Spec(l.weight * Gr * Fr * Dr / 4);
    }
    break;
    default: break;  // specular lobes
    }
    return Spec(0.f);
}

template <uint32_t LM>
GX_DEV float lobe_pdf(const DLobe &l, V3 wo, V3 wi) {
    switch (l.kind) {
    case LOBE_SPEC_REFL: case LOBE_SPEC_TRANS: case LOBE_FRESNEL_SPEC: return 0.f;
    case LOBE_LAMBERT_TRANS: if (GX_HAS_LOBE(LOBE_LAMBERT_TRANS)) return !same_hemisphere(wo, wi) ? abs_cos_theta(wi) * GX_INV_PI : 0.f; break;
    // MicrofacetReflection::Pdf (Reflection.cpp:216-221) / MicrofacetTransmission::Pdf (:262-276): one distribution->Pdf call
    case LOBE_MICRO_REFL: case LOBE_MICRO_TRANS: if (GX_HAS_LOBE(LOBE_MICRO_REFL) || GX_HAS_LOBE(LOBE_MICRO_TRANS)) {
        const bool refl = l.kind == LOBE_MICRO_REFL;
        V3 wh;
        float scale;
        if (refl) {
            if (!same_hemisphere(wo, wi)) return 0.f;
            wh = normalize(wo + wi);
            scale = 4 * dot(wo, wh);
        } else {
            if (same_hemisphere(wo, wi)) return 0.f;
            const bool up = cos_theta(wo) > 0;
            float eta = (up ? l.etaB : l.etaA) / (up ? l.etaA : l.etaB);
            wh = normalize(wo + wi * eta);
            if (dot(wo, wh) * dot(wi, wh) > 0) return 0.f;
            float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
            scale = fabsf((eta * eta * dot(wi, wh)) / (sqrtDenom * sqrtDenom));
        }
        const float p = tr_pdf(l.alphax, l.alphay, wo, wh);
        return refl ? p / scale : p * scale;
    }
    break;
    case LOBE_DISNEY_CLEARCOAT: if (GX_HAS_LOBE(LOBE_DISNEY_CLEARCOAT)) {
        if (!same_hemisphere(wo, wi)) return 0.f;
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return 0.f;
        wh = normalize(wh);
        float Dr = gtr1(abs_cos_theta(wh), l.gloss);
        return Dr * abs_cos_theta(wh) / (4 * dot(wo, wh));
    }
    break;
    default: break;
    }
    return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * GX_INV_PI : 0.f;  // BxDF::Pdf
}

// *pdf is left untouched on early-outs (the caller zeroes it), as in the reference.  need_f = false: the caller (BSDF::Sample_f for a
// non-specular lobe, Reflection.cpp:548-556) replaces the returned value by the sum over all matching lobes, so the lobe's own f is not
// evaluated a first time here.
template <uint32_t LM>
GX_DEV Spec lobe_sample(const DLobe &l, V3 wo, V3 *wi, float u0, float u1, float *pdf, int *sampledType, bool need_f = true) {
    switch (l.kind) {
    case LOBE_SPEC_REFL: if (GX_HAS_LOBE(LOBE_SPEC_REFL)) {  // Reflection.cpp:89-97
        *wi = V3(-wo.x, -wo.y, wo.z);
        *pdf = 1;
        return fresnel_eval<LM>(l, cos_theta(*wi)) * spec3(l.R) / abs_cos_theta(*wi);
    }
    break;
    case LOBE_SPEC_TRANS: if (GX_HAS_LOBE(LOBE_SPEC_TRANS)) {  // Reflection.cpp:105-122
        bool entering = cos_theta(wo) > 0;
        float etaI = entering ? l.etaA : l.etaB, etaT = entering ? l.etaB : l.etaA;
        if (!refract(wo, faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return Spec(0.f);
        *pdf = 1;
        Spec ft = spec3(l.T) * (Spec(1.f) - fresnel_eval<LM>(l, cos_theta(*wi)));
        ft = ft * ((etaI * etaI) / (etaT * etaT));
        return ft / abs_cos_theta(*wi);
    }
    break;
    case LOBE_FRESNEL_SPEC: if (GX_HAS_LOBE(LOBE_FRESNEL_SPEC)) {  // Reflection.cpp:346-380
        float F = fr_dielectric(cos_theta(wo), l.etaA, l.etaB);
        if (u0 < F) {
            *wi = V3(-wo.x, -wo.y, wo.z);
            *sampledType = BSDF_SPECULAR | BSDF_REFLECTION;
            *pdf = F;
            return F * spec3(l.R) / abs_cos_theta(*wi);
        } else {
            bool entering = cos_theta(wo) > 0;
            float etaI = entering ? l.etaA : l.etaB, etaT = entering ? l.etaB : l.etaA;
            if (!refract(wo, faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return Spec(0.f);
            Spec ft = spec3(l.T) * (1 - F);
            ft = ft * ((etaI * etaI) / (etaT * etaT));
            *sampledType = BSDF_SPECULAR | BSDF_TRANSMISSION;
            *pdf = 1 - F;
            return ft / abs_cos_theta(*wi);
        }
    }
    break;
    // MicrofacetReflection::Sample_f (Reflection.cpp:206-214) and MicrofacetTransmission::Sample_f (:249-260) start with the
    // same visible-normal sample; rough glass picks one of the two lobes per path, so a wave holds both kinds -- sharing the
    // (expensive) Sample_wh keeps that divergence out of it.
    case LOBE_MICRO_REFL: case LOBE_MICRO_TRANS: if (GX_HAS_LOBE(LOBE_MICRO_REFL) || GX_HAS_LOBE(LOBE_MICRO_TRANS)) {
        if (wo.z == 0) return Spec(0.f);
        V3 wh = tr_sample_wh(l.alphax, l.alphay, wo, u0, u1);
        if (dot(wo, wh) < 0) return Spec(0.f);
        if (l.kind == LOBE_MICRO_REFL) {
            *wi = reflect(wo, wh);
            if (!same_hemisphere(wo, *wi)) return Spec(0.f);
            *pdf = tr_pdf(l.alphax, l.alphay, wo, wh) / (4 * dot(wo, wh));
        } else {
            const bool up = cos_theta(wo) > 0;
            float eta = (up ? l.etaA : l.etaB) / (up ? l.etaB : l.etaA);
            if (!refract(wo, wh, eta, wi)) return Spec(0.f);
            *pdf = lobe_pdf<LM>(l, wo, *wi);
        }
        return need_f ? lobe_f<LM>(l, wo, *wi) : Spec(0.f);
    }
    break;
    case LOBE_LAMBERT_TRANS: if (GX_HAS_LOBE(LOBE_LAMBERT_TRANS)) {  // Reflection.cpp:146-155
        *wi = cosine_sample_hemisphere(u0, u1);
        if (wo.z > 0) wi->z *= -1;
        *pdf = lobe_pdf<LM>(l, wo, *wi);
        return need_f ? lobe_f<LM>(l, wo, *wi) : Spec(0.f);
    }
    break;
    case LOBE_DISNEY_CLEARCOAT: if (GX_HAS_LOBE(LOBE_DISNEY_CLEARCOAT)) {  // DisneyMaterial.cpp:255-276
        if (wo.z == 0) return Spec(0.f);
        float alpha2 = l.gloss * l.gloss;
        float cosTheta = gx_sqrt(fmaxf(0.f, (1 - gx_pow(alpha2, 1 - u0)) / (1 - alpha2)));
        float sinTheta = gx_sqrt(fmaxf(0.f, 1 - cosTheta * cosTheta));
        float phi = 2 * GX_PI * u1;
        float sinPhi, cosPhi;
        gx_sincos(phi, &sinPhi, &cosPhi);
        V3 wh(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
        if (!same_hemisphere(wo, wh)) wh = -wh;
        *wi = reflect(wo, wh);
        if (!same_hemisphere(wo, *wi)) return Spec(0.f);
        *pdf = lobe_pdf<LM>(l, wo, *wi);
        return need_f ? lobe_f<LM>(l, wo, *wi) : Spec(0.f);
    }
    break;
    default: break;
    }
    // BxDF::Sample_f, Reflection.cpp:394-401
    *wi = cosine_sample_hemisphere(u0, u1);
    if (wo.z < 0) wi->z *= -1;
    *pdf = lobe_pdf<LM>(l, wo, *wi);
    return need_f ? lobe_f<LM>(l, wo, *wi) : Spec(0.f);
}

// ---- BSDF container, Reflection.h:102-154 + Reflection.cpp:440-563 ----
template <uint32_t LM>
struct Bsdf {
    const DMaterial *mat;
    V3 ns, ng, ss, ts;
    GX_DEV V3 to_local(V3 v) const { return V3(dot(v, ss), dot(v, ts), dot(v, ns)); }
    GX_DEV V3 to_world(V3 v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    GX_DEV static bool matches(int type, int flags) { return (type & flags) == type; }
    GX_DEV int num_components(int flags) const {
        int n = 0;
        for (int i = 0; i < mat->n_lobes; ++i) if (matches(mat->lobes[i].type, flags)) ++n;
        return n;
    }
    // The diffuse class (LM_DIFFUSE) holds materials of at most ONE lobe, a Lambert or Oren-Nayar reflection (compile_material): lobe
    // counting, choice and the sums over "the other lobes" of Reflection.cpp:440-563 reduce to that lobe -- same arithmetic, without the
    // loops over mat->lobes[i].type (each a dependent table read).
    static constexpr bool kSingle = (LM == LM_DIFFUSE);
    static constexpr int kSingleType = BSDF_REFLECTION | BSDF_DIFFUSE;
    GX_DEV Spec f(V3 woW, V3 wiW, int flags) const {
        V3 wi = to_local(wiW), wo = to_local(woW);
        if (wo.z == 0) return Spec(0.f);
        bool refl = dot(wiW, ng) * dot(woW, ng) > 0;
        if (kSingle) return (mat->n_lobes > 0 && refl && matches(kSingleType, flags)) ? lobe_f<LM>(mat->lobes[0], wo, wi) : Spec(0.f);
        return sum_f(wo, wi, flags, refl);
    }
    // sum of lobe_f over the lobes that match `flags` and the reflect / transmit side, in lobe order (Reflection.cpp:458-463).
    // Each lane walks ITS OWN list of applicable lobes: lanes of a wave whose applicable lobe differs (rough glass: the
    // reflection lobe for some, the transmission lobe for others) evaluate them in the same iteration instead of in turn.
    GX_DEV Spec sum_f(V3 wo, V3 wi, int flags, bool refl) const {
        Spec f(0.f);
        const int n = mat->n_lobes;
        int i = 0;
        while (true) {
            while (i < n) {
                const int t = mat->lobes[i].type;
                if (matches(t, flags) && ((refl && (t & BSDF_REFLECTION)) || (!refl && (t & BSDF_TRANSMISSION)))) break;
                ++i;
            }
            if (i >= n) break;
            f = f + lobe_f<LM>(mat->lobes[i], wo, wi);
            ++i;
        }
        return f;
    }
    GX_DEV float pdf(V3 woW, V3 wiW, int flags) const {
        if (mat->n_lobes == 0) return 0.f;
        V3 wo = to_local(woW), wi = to_local(wiW);
        if (wo.z == 0) return 0.f;
        if (kSingle) return matches(kSingleType, flags) ? 0.f + lobe_pdf<LM>(mat->lobes[0], wo, wi) : 0.f;   // (p = 0; p += pdf; p / 1)
        float p = 0.f;
        int matching = 0;
        for (int i = 0; i < mat->n_lobes; ++i) {
            const DLobe &l = mat->lobes[i];
            if (matches(l.type, flags)) { ++matching; p += lobe_pdf<LM>(l, wo, wi); }
        }
        return matching > 0 ? p / matching : 0.f;
    }
    GX_DEV Spec sample_f(V3 woW, V3 *wiW, float u0, float u1, float *pdf, int flags, int *sampledType) const {
        if (kSingle) {   // matchingComps == 1: comp = 0, uRemapped = min(u[0] * 1 - 0, OneMinusEpsilon), no other lobes' pdfs, f = that lobe's f if it reflects
            *sampledType = 0;
            if (mat->n_lobes == 0 || !matches(kSingleType, flags)) { *pdf = 0; return Spec(0.f); }
            const DLobe &bx = mat->lobes[0];
            const float ur0 = fminf(u0 * 1 - 0, GX_ONE_MINUS_EPS);
            V3 wi, wo = to_local(woW);
            *pdf = 0;
            if (wo.z == 0) return Spec(0.f);
            *sampledType = kSingleType;
            (void)lobe_sample<LM>(bx, wo, &wi, ur0, u1, pdf, sampledType, false);
            if (*pdf == 0) { *sampledType = 0; return Spec(0.f); }
            *wiW = to_world(wi);
            const bool refl = dot(*wiW, ng) * dot(woW, ng) > 0;
            return refl ? lobe_f<LM>(bx, wo, wi) : Spec(0.f);
        }
        int matching = num_components(flags);
        *sampledType = 0;
        if (matching == 0) { *pdf = 0; return Spec(0.f); }
        int comp = min((int)floorf(u0 * matching), matching - 1);
        int which = -1, count = comp;
        for (int i = 0; i < mat->n_lobes; ++i)
            if (matches(mat->lobes[i].type, flags) && count-- == 0) { which = i; break; }
        const DLobe &bx = mat->lobes[which];
        float ur0 = fminf(u0 * matching - comp, GX_ONE_MINUS_EPS);
        V3 wi, wo = to_local(woW);
        *pdf = 0;
        if (wo.z == 0) return Spec(0.f);
        *sampledType = bx.type;
        Spec f = lobe_sample<LM>(bx, wo, &wi, ur0, u1, pdf, sampledType, (bx.type & BSDF_SPECULAR) != 0);
        if (*pdf == 0) { *sampledType = 0; return Spec(0.f); }
        *wiW = to_world(wi);
        if (!(bx.type & BSDF_SPECULAR) && matching > 1) {   // the other matching lobes' pdfs, in lobe order; per-lane walk as in sum_f
            const int n = mat->n_lobes;
            int i = 0;
            while (true) {
                while (i < n && !(i != which && matches(mat->lobes[i].type, flags))) ++i;
                if (i >= n) break;
                *pdf += lobe_pdf<LM>(mat->lobes[i], wo, wi);
                ++i;
            }
        }
        if (matching > 1) *pdf /= matching;
        if (!(bx.type & BSDF_SPECULAR)) {
            bool refl = dot(*wiW, ng) * dot(woW, ng) > 0;
            f = sum_f(wo, wi, flags, refl);
        }
        return f;
    }
};

}  // namespace gnxr

// ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_bsdf.h: CPU restatement of the reference's scattering functions:
//   BxDFs, Fresnel, BSDF container   core/Reflection.{h,cpp}
//   Trowbridge-Reitz distribution    core/MicroFacet.{h,cpp}
//   Material::Bump                   core/Material.cpp:16-52
//   Matte/Mirror/Glass/Metal/Plastic/Disney ::ComputeScatteringFunctions   materials/*.cpp
// The virtual BxDF hierarchy becomes one tagged POD (`Lobe`); arithmetic order is the reference's.
#pragma once
#include "o_scene.h"

namespace gnxo {

enum BxDFType {
    BSDF_REFLECTION = 1 << 0, BSDF_TRANSMISSION = 1 << 1, BSDF_DIFFUSE = 1 << 2, BSDF_GLOSSY = 1 << 3,
    BSDF_SPECULAR = 1 << 4,
    BSDF_ALL = BSDF_DIFFUSE | BSDF_GLOSSY | BSDF_SPECULAR | BSDF_REFLECTION | BSDF_TRANSMISSION,
};

// Reflection.h:19-88
inline Float CosTheta(const V3 &w) { return w.z; }
inline Float Cos2Theta(const V3 &w) { return w.z * w.z; }
inline Float AbsCosTheta(const V3 &w) { return std::abs(w.z); }
inline Float Sin2Theta(const V3 &w) { return std::max((Float)0, (Float)1 - Cos2Theta(w)); }
inline Float SinTheta(const V3 &w) { return std::sqrt(Sin2Theta(w)); }
inline Float TanTheta(const V3 &w) { return SinTheta(w) / CosTheta(w); }
inline Float Tan2Theta(const V3 &w) { return Sin2Theta(w) / Cos2Theta(w); }
inline Float CosPhi(const V3 &w) { Float sinTheta = SinTheta(w); return (sinTheta == 0) ? 1 : Clamp(w.x / sinTheta, -1, 1); }
inline Float SinPhi(const V3 &w) { Float sinTheta = SinTheta(w); return (sinTheta == 0) ? 0 : Clamp(w.y / sinTheta, -1, 1); }
inline Float Cos2Phi(const V3 &w) { return CosPhi(w) * CosPhi(w); }
inline Float Sin2Phi(const V3 &w) { return SinPhi(w) * SinPhi(w); }
inline V3 Reflect(const V3 &wo, const V3 &n) { return -wo + 2 * Dot(wo, n) * n; }
inline bool Refract(const V3 &wi, const V3 &n, Float eta, V3 *wt) {
    Float cosThetaI = Dot(n, wi);
    Float sin2ThetaI = std::max(Float(0), Float(1 - cosThetaI * cosThetaI));
    Float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    Float cosThetaT = std::sqrt(1 - sin2ThetaT);
    *wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}
inline bool SameHemisphere(const V3 &w, const V3 &wp) { return w.z * wp.z > 0; }

// Reflection.cpp:16-38
inline Float FrDielectric(Float cosThetaI, Float etaI, Float etaT) {
    cosThetaI = Clamp(cosThetaI, -1, 1);
    bool entering = cosThetaI > 0.f;
    if (!entering) { std::swap(etaI, etaT); cosThetaI = std::abs(cosThetaI); }
    Float sinThetaI = std::sqrt(std::max((Float)0, 1 - cosThetaI * cosThetaI));
    Float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    Float cosThetaT = std::sqrt(std::max((Float)0, 1 - sinThetaT * sinThetaT));
    Float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    Float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
// Reflection.cpp:41-64
inline Spec FrConductor(Float cosThetaI, const Spec &etai, const Spec &etat, const Spec &k) {
    cosThetaI = Clamp(cosThetaI, -1, 1);
    Spec eta = etat / etai;
    Spec etak = k / etai;
    Float cosThetaI2 = cosThetaI * cosThetaI;
    Float sinThetaI2 = 1. - cosThetaI2;
    Spec eta2 = eta * eta;
    Spec etak2 = etak * etak;
    Spec t0 = eta2 - etak2 - Spec(sinThetaI2);
    Spec a2plusb2 = Sqrt(t0 * t0 + 4 * eta2 * etak2);
    Spec t1 = a2plusb2 + Spec(cosThetaI2);
    Spec a = Sqrt(0.5f * (a2plusb2 + t0));
    Spec t2 = (Float)2 * cosThetaI * a;
    Spec Rs = (t1 - t2) / (t1 + t2);
    Spec t3 = cosThetaI2 * a2plusb2 + Spec(sinThetaI2 * sinThetaI2);
    Spec t4 = t2 * sinThetaI2;
    Spec Rp = Rs * (t3 - t4) / (t3 + t4);
    return 0.5 * (Rp + Rs);
}

// DisneyMaterial.cpp:50-60
inline Float sqr(Float x) { return x * x; }
inline Float SchlickWeight(Float cosTheta) { Float m = Clamp(1 - cosTheta, 0, 1); return (m * m) * (m * m) * m; }
inline Float FrSchlick(Float R0, Float cosTheta) { return Lerp(SchlickWeight(cosTheta), R0, 1); }
inline Spec FrSchlick(const Spec &R0, Float cosTheta) { return Lerp(SchlickWeight(cosTheta), R0, Spec(1.)); }
inline Float SchlickR0FromEta(Float eta) { return sqr(eta - 1) / sqr(eta + 1); }
// DisneyMaterial.cpp:224-237
inline Float GTR1(Float cosTheta, Float alpha) {
    Float alpha2 = alpha * alpha;
    return (alpha2 - 1) / (Pi * std::log(alpha2) * (1 + (alpha2 - 1) * cosTheta * cosTheta));
}
inline Float smithG_GGX(Float cosTheta, Float alpha) {
    Float alpha2 = alpha * alpha;
    Float cosTheta2 = cosTheta * cosTheta;
    // unqualified `sqrt` in DisneyMaterial.cpp:236 binds to ::sqrt(double): the sum and the reciprocal are
    // evaluated in double and rounded to Float once.
    return 1 / (cosTheta + ::sqrt((double)(alpha2 + cosTheta2 - alpha2 * cosTheta2)));
}

enum FresnelKind { F_NOOP = 0, F_DIELECTRIC, F_CONDUCTOR, F_DISNEY };
struct FresnelP {
    int kind = F_NOOP;
    Float etaI = 1, etaT = 1;        // dielectric
    Spec cEtaI, cEtaT, cK;           // conductor
    Spec R0; Float metallic = 0, eta = 1;  // disney
    Spec Evaluate(Float cosI) const {
        switch (kind) {
        case F_DIELECTRIC: return Spec(FrDielectric(cosI, etaI, etaT));                 // Reflection.cpp:79-82
        case F_CONDUCTOR: return FrConductor(std::abs(cosI), cEtaI, cEtaT, cK);        // Reflection.cpp:67-70
        case F_DISNEY: return Lerp(metallic, Spec(FrDielectric(cosI, 1, eta)), FrSchlick(R0, cosI));  // DisneyMaterial.cpp:312-316
        default: return Spec(1.);                                                       // FresnelNoOp
        }
    }
};

// TrowbridgeReitzDistribution, MicroFacet.cpp:129-136,150-159,215-316 (sampleVisibleArea = true everywhere)
struct TRDist {
    Float alphax = 0.001f, alphay = 0.001f;
    bool disneyG = false;  // DisneyMicrofacetDistribution::G, DisneyMaterial.cpp:338-342
    void Set(Float ax, Float ay) { alphax = std::max(Float(0.001), ax); alphay = std::max(Float(0.001), ay); }
    static Float RoughnessToAlpha(Float roughness) {  // MicroFacet.h:97-103
        roughness = std::max(roughness, (Float)1e-3);
        Float x = std::log(roughness);
        return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
    }
    Float D(const V3 &wh) const {
        Float tan2Theta = Tan2Theta(wh);
        if (std::isinf(tan2Theta)) return 0.;
        const Float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
        Float e = (Cos2Phi(wh) / (alphax * alphax) + Sin2Phi(wh) / (alphay * alphay)) * tan2Theta;
        return 1 / (Pi * alphax * alphay * cos4Theta * (1 + e) * (1 + e));
    }
    Float Lambda(const V3 &w) const {
        Float absTanTheta = std::abs(TanTheta(w));
        if (std::isinf(absTanTheta)) return 0.;
        Float alpha = std::sqrt(Cos2Phi(w) * alphax * alphax + Sin2Phi(w) * alphay * alphay);
        Float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
        return (-1 + std::sqrt(1.f + alpha2Tan2Theta)) / 2;
    }
    Float G1(const V3 &w) const { return 1 / (1 + Lambda(w)); }
    Float G(const V3 &wo, const V3 &wi) const {
        if (disneyG) return G1(wo) * G1(wi);
        return 1 / (1 + Lambda(wo) + Lambda(wi));
    }
    Float Pdf(const V3 &wo, const V3 &wh) const { return D(wh) * G1(wo) * AbsDot(wo, wh) / AbsCosTheta(wo); }
    // MicroFacet.cpp:215-260.  The normal-incidence branch mixes double and float exactly as the
    // reference source does (double literal 6.28318530718, unqualified sqrt/cos/sin).
    static void Sample11(Float cosTheta, Float U1, Float U2, Float *slope_x, Float *slope_y) {
        if (cosTheta > .9999) {
            // MicroFacet.cpp:220-223: unqualified sqrt/cos/sin bind to the double versions
            Float r = ::sqrt((double)(U1 / (1 - U1)));
            Float phi = 6.28318530718 * U2;
            *slope_x = r * ::cos((double)phi);
            *slope_y = r * ::sin((double)phi);
            return;
        }
        Float sinTheta = std::sqrt(std::max((Float)0, (Float)1 - cosTheta * cosTheta));
        Float tanTheta = sinTheta / cosTheta;
        Float a = 1 / tanTheta;
        Float G1 = 2 / (1 + std::sqrt(1.f + 1.f / (a * a)));
        Float A = 2 * U1 / G1 - 1;
        Float tmp = 1.f / (A * A - 1.f);
        if (tmp > 1e10) tmp = 1e10;
        Float B = tanTheta;
        Float D = std::sqrt(std::max(Float(B * B * tmp * tmp - (A * A - B * B) * tmp), Float(0)));
        Float slope_x_1 = B * tmp - D;
        Float slope_x_2 = B * tmp + D;
        *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
        Float S;
        if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
        else { S = -1.f; U2 = 2.f * (.5f - U2); }
        Float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        *slope_y = S * z * std::sqrt(1.f + *slope_x * *slope_x);
    }
    // MicroFacet.cpp:262-285
    static V3 SampleTR(const V3 &wi, Float alpha_x, Float alpha_y, Float U1, Float U2) {
        V3 wiStretched = Normalize(V3(alpha_x * wi.x, alpha_y * wi.y, wi.z));
        Float slope_x, slope_y;
        Sample11(CosTheta(wiStretched), U1, U2, &slope_x, &slope_y);
        Float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
        slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
        slope_x = tmp;
        slope_x = alpha_x * slope_x;
        slope_y = alpha_y * slope_y;
        return Normalize(V3(-slope_x, -slope_y, 1.));
    }
    V3 Sample_wh(const V3 &wo, const P2 &u) const {
        bool flip = wo.z < 0;
        V3 wh = SampleTR(flip ? -wo : wo, alphax, alphay, u.x, u.y);
        if (flip) wh = -wh;
        return wh;
    }
};

enum LobeKind {
    L_LAMBERT, L_OREN, L_SPEC_REFL, L_SPEC_TRANS, L_FRESNEL_SPEC, L_MICRO_REFL, L_MICRO_TRANS, L_LAMBERT_TRANS,
    L_DISNEY_DIFFUSE, L_DISNEY_FAKESS, L_DISNEY_RETRO, L_DISNEY_SHEEN, L_DISNEY_CLEARCOAT
};

struct Lobe {
    int kind = L_LAMBERT;
    int type = 0;
    Spec R, T;
    Float A = 0, B = 0;          // OrenNayar
    Float etaA = 1, etaB = 1;    // transmission lobes
    FresnelP fresnel;
    TRDist dist;
    Float roughness = 0, weight = 0, gloss = 0;  // disney
    bool MatchesFlags(int t) const { return (type & t) == type; }

    Spec f(const V3 &wo, const V3 &wi) const {
        switch (kind) {
        case L_LAMBERT: return R * InvPi;  // Reflection.cpp:135-138
        case L_LAMBERT_TRANS: return T * InvPi;
        case L_OREN: {  // Reflection.cpp:173-198
            Float sinThetaI = SinTheta(wi), sinThetaO = SinTheta(wo);
            Float maxCos = 0;
            if (sinThetaI > 1e-4 && sinThetaO > 1e-4) {
                Float sinPhiI = SinPhi(wi), cosPhiI = CosPhi(wi);
                Float sinPhiO = SinPhi(wo), cosPhiO = CosPhi(wo);
                Float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
                maxCos = std::max((Float)0, dCos);
            }
            Float sinAlpha, tanBeta;
            if (AbsCosTheta(wi) > AbsCosTheta(wo)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / AbsCosTheta(wi); }
            else { sinAlpha = sinThetaI; tanBeta = sinThetaO / AbsCosTheta(wo); }
            return R * InvPi * (A + B * maxCos * sinAlpha * tanBeta);
        }
        case L_SPEC_REFL: case L_SPEC_TRANS: case L_FRESNEL_SPEC: return Spec(0.f);
        case L_MICRO_REFL: {  // Reflection.cpp:223-237
            Float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
            V3 wh = wi + wo;
            if (cosThetaI == 0 || cosThetaO == 0) return Spec(0.);
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
            wh = Normalize(wh);
            Spec F = fresnel.Evaluate(Dot(wi, Faceforward(wh, V3(0, 0, 1))));
            return R * dist.D(wh) * dist.G(wo, wi) * F / (4 * cosThetaI * cosThetaO);
        }
        case L_MICRO_TRANS: {  // Reflection.cpp:278-302
            if (SameHemisphere(wo, wi)) return Spec(0.f);
            Float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
            if (cosThetaI == 0 || cosThetaO == 0) return Spec(0);
            Float eta = CosTheta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
            V3 wh = Normalize(wo + wi * eta);
            if (wh.z < 0) wh = -wh;
            if (Dot(wo, wh) * Dot(wi, wh) > 0) return Spec(0);
            Spec F = fresnel.Evaluate(Dot(wo, wh));
            Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
            Float factor = (1 / eta);  // TransportMode::Radiance
            return (Spec(1.f) - F) * T *
                   std::abs(dist.D(wh) * dist.G(wo, wi) * eta * eta * AbsDot(wi, wh) * AbsDot(wo, wh) * factor * factor /
                            (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
        }
        case L_DISNEY_DIFFUSE: {  // DisneyMaterial.cpp:64-72
            Float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
            return R * InvPi * (1 - Fo / 2) * (1 - Fi / 2);
        }
        case L_DISNEY_FAKESS: {  // DisneyMaterial.cpp:105-122
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
            wh = Normalize(wh);
            Float cosThetaD = Dot(wi, wh);
            Float Fss90 = cosThetaD * cosThetaD * roughness;
            Float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
            Float Fss = Lerp(Fo, 1.0, Fss90) * Lerp(Fi, 1.0, Fss90);
            Float ss = 1.25f * (Fss * (1 / (AbsCosTheta(wo) + AbsCosTheta(wi)) - .5f) + .5f);
            return R * InvPi * ss;
        }
        case L_DISNEY_RETRO: {  // DisneyMaterial.cpp:151-164
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
            wh = Normalize(wh);
            Float cosThetaD = Dot(wi, wh);
            Float Fo = SchlickWeight(AbsCosTheta(wo)), Fi = SchlickWeight(AbsCosTheta(wi));
            Float Rr = 2 * roughness * cosThetaD * cosThetaD;
            return R * InvPi * Rr * (Fo + Fi + Fo * Fi * (Rr - 1));
        }
        case L_DISNEY_SHEEN: {  // DisneyMaterial.cpp:189-197
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
            wh = Normalize(wh);
            Float cosThetaD = Dot(wi, wh);
            return R * SchlickWeight(cosThetaD);
        }
        case L_DISNEY_CLEARCOAT: {  // DisneyMaterial.cpp:239-253
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
            wh = Normalize(wh);
            Float Dr = GTR1(AbsCosTheta(wh), gloss);
            Float Fr = FrSchlick(.04, Dot(wo, wh));
            Float Gr = smithG_GGX(AbsCosTheta(wo), .25) * smithG_GGX(AbsCosTheta(wi), .25);
            return Spec(weight * Gr * Fr * Dr / 4);
        }
        }
        return Spec(0.f);
    }

    Float Pdf(const V3 &wo, const V3 &wi) const {
        switch (kind) {
        case L_SPEC_REFL: case L_SPEC_TRANS: case L_FRESNEL_SPEC: return 0;
        case L_LAMBERT_TRANS: return !SameHemisphere(wo, wi) ? AbsCosTheta(wi) * InvPi : 0;  // Reflection.cpp:157-160
        case L_MICRO_REFL: {  // Reflection.cpp:216-221
            if (!SameHemisphere(wo, wi)) return 0;
            V3 wh = Normalize(wo + wi);
            return dist.Pdf(wo, wh) / (4 * Dot(wo, wh));
        }
        case L_MICRO_TRANS: {  // Reflection.cpp:262-276
            if (SameHemisphere(wo, wi)) return 0;
            Float eta = CosTheta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
            V3 wh = Normalize(wo + wi * eta);
            if (Dot(wo, wh) * Dot(wi, wh) > 0) return 0;
            Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
            Float dwh_dwi = std::abs((eta * eta * Dot(wi, wh)) / (sqrtDenom * sqrtDenom));
            return dist.Pdf(wo, wh) * dwh_dwi;
        }
        case L_DISNEY_CLEARCOAT: {  // DisneyMaterial.cpp:278-295
            if (!SameHemisphere(wo, wi)) return 0;
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return 0;
            wh = Normalize(wh);
            Float Dr = GTR1(AbsCosTheta(wh), gloss);
            return Dr * AbsCosTheta(wh) / (4 * Dot(wo, wh));
        }
        default: return SameHemisphere(wo, wi) ? AbsCosTheta(wi) * InvPi : 0;  // BxDF::Pdf, Reflection.cpp:403-406
        }
    }

    // Returns f; *pdf left untouched on early-outs exactly as the reference does (BSDF::Sample_f
    // zeroes it beforehand).
    Spec Sample_f(const V3 &wo, V3 *wi, const P2 &u, Float *pdf, int *sampledType) const {
        switch (kind) {
        case L_SPEC_REFL: {  // Reflection.cpp:89-97
            *wi = V3(-wo.x, -wo.y, wo.z);
            *pdf = 1;
            return fresnel.Evaluate(CosTheta(*wi)) * R / AbsCosTheta(*wi);
        }
        case L_SPEC_TRANS: {  // Reflection.cpp:105-122
            bool entering = CosTheta(wo) > 0;
            Float etaI = entering ? etaA : etaB;
            Float etaT = entering ? etaB : etaA;
            if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return Spec(0.f);
            *pdf = 1;
            Spec ft = T * (Spec(1.) - fresnel.Evaluate(CosTheta(*wi)));
            ft *= (etaI * etaI) / (etaT * etaT);
            return ft / AbsCosTheta(*wi);
        }
        case L_FRESNEL_SPEC: {  // Reflection.cpp:346-380
            Float F = FrDielectric(CosTheta(wo), etaA, etaB);
            if (u.x < F) {
                *wi = V3(-wo.x, -wo.y, wo.z);
                if (sampledType) *sampledType = BSDF_SPECULAR | BSDF_REFLECTION;
                *pdf = F;
                return F * R / AbsCosTheta(*wi);
            } else {
                bool entering = CosTheta(wo) > 0;
                Float etaI = entering ? etaA : etaB;
                Float etaT = entering ? etaB : etaA;
                if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return Spec(0.f);
                Spec ft = T * (1 - F);
                ft *= (etaI * etaI) / (etaT * etaT);
                if (sampledType) *sampledType = BSDF_SPECULAR | BSDF_TRANSMISSION;
                *pdf = 1 - F;
                return ft / AbsCosTheta(*wi);
            }
        }
        case L_MICRO_REFL: {  // Reflection.cpp:206-214
            if (wo.z == 0) return Spec(0.);
            V3 wh = dist.Sample_wh(wo, u);
            if (Dot(wo, wh) < 0) return Spec(0.);
            *wi = Reflect(wo, wh);
            if (!SameHemisphere(wo, *wi)) return Spec(0.f);
            *pdf = dist.Pdf(wo, wh) / (4 * Dot(wo, wh));
            return f(wo, *wi);
        }
        case L_MICRO_TRANS: {  // Reflection.cpp:249-260
            if (wo.z == 0) return Spec(0.);
            V3 wh = dist.Sample_wh(wo, u);
            if (Dot(wo, wh) < 0) return Spec(0.);
            Float eta = CosTheta(wo) > 0 ? (etaA / etaB) : (etaB / etaA);
            if (!Refract(wo, wh, eta, wi)) return Spec(0.f);
            *pdf = Pdf(wo, *wi);
            return f(wo, *wi);
        }
        case L_LAMBERT_TRANS: {  // Reflection.cpp:146-155
            *wi = CosineSampleHemisphere(u);
            if (wo.z > 0) wi->z *= -1;
            *pdf = Pdf(wo, *wi);
            return f(wo, *wi);
        }
        case L_DISNEY_CLEARCOAT: {  // DisneyMaterial.cpp:255-276
            if (wo.z == 0) return Spec(0.);
            Float alpha2 = gloss * gloss;
            Float cosTheta = std::sqrt(std::max(Float(0), (1 - std::pow(alpha2, 1 - u.x)) / (1 - alpha2)));
            Float sinTheta = std::sqrt(std::max((Float)0, 1 - cosTheta * cosTheta));
            Float phi = 2 * Pi * u.y;
            V3 wh = SphericalDirection(sinTheta, cosTheta, phi);
            if (!SameHemisphere(wo, wh)) wh = -wh;
            *wi = Reflect(wo, wh);
            if (!SameHemisphere(wo, *wi)) return Spec(0.f);
            *pdf = Pdf(wo, *wi);
            return f(wo, *wi);
        }
        default: {  // BxDF::Sample_f, Reflection.cpp:394-401
            *wi = CosineSampleHemisphere(u);
            if (wo.z < 0) wi->z *= -1;
            *pdf = Pdf(wo, *wi);
            return f(wo, *wi);
        }
        }
    }
};

// BSDF, Reflection.h:102-154 + Reflection.cpp:440-563
struct BSDF {
    Float eta = 1;
    V3 ns, ng, ss, ts;
    int nBxDFs = 0;
    Lobe bxdfs[8];
    BSDF() {}
    BSDF(const SurfaceInteraction &si, Float eta = 1)
        : eta(eta), ns(si.sn), ng(si.n), ss(Normalize(si.sdpdu)), ts(Cross(ns, ss)) {}
    void Add(const Lobe &l) { bxdfs[nBxDFs++] = l; }
    int NumComponents(int flags = BSDF_ALL) const {
        int num = 0;
        for (int i = 0; i < nBxDFs; ++i) if (bxdfs[i].MatchesFlags(flags)) ++num;
        return num;
    }
    V3 WorldToLocal(const V3 &v) const { return V3(Dot(v, ss), Dot(v, ts), Dot(v, ns)); }
    V3 LocalToWorld(const V3 &v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    Spec f(const V3 &woW, const V3 &wiW, int flags = BSDF_ALL) const {
        V3 wi = WorldToLocal(wiW), wo = WorldToLocal(woW);
        if (wo.z == 0) return Spec(0.);
        bool reflect = Dot(wiW, ng) * Dot(woW, ng) > 0;
        Spec f(0.f);
        for (int i = 0; i < nBxDFs; ++i)
            if (bxdfs[i].MatchesFlags(flags) &&
                ((reflect && (bxdfs[i].type & BSDF_REFLECTION)) || (!reflect && (bxdfs[i].type & BSDF_TRANSMISSION))))
                f += bxdfs[i].f(wo, wi);
        return f;
    }
    Float Pdf(const V3 &woWorld, const V3 &wiWorld, int flags = BSDF_ALL) const {
        if (nBxDFs == 0.f) return 0.f;
        V3 wo = WorldToLocal(woWorld), wi = WorldToLocal(wiWorld);
        if (wo.z == 0) return 0.;
        Float pdf = 0.f;
        int matchingComps = 0;
        for (int i = 0; i < nBxDFs; ++i)
            if (bxdfs[i].MatchesFlags(flags)) { ++matchingComps; pdf += bxdfs[i].Pdf(wo, wi); }
        Float v = matchingComps > 0 ? pdf / matchingComps : 0.f;
        return v;
    }
    // Reflection.cpp:474-546.  NOTE: on the `wo.z == 0` early-out the reference returns before writing
    // *pdf; callers only test `f.IsBlack() || pdf == 0` so the value never matters.  We set *pdf = 0.
    Spec Sample_f(const V3 &woWorld, V3 *wiWorld, const P2 &u, Float *pdf, int type = BSDF_ALL, int *sampledType = nullptr) const {
        int matchingComps = NumComponents(type);
        if (matchingComps == 0) { *pdf = 0; if (sampledType) *sampledType = 0; return Spec(0); }
        int comp = std::min((int)std::floor(u.x * matchingComps), matchingComps - 1);
        const Lobe *bxdf = nullptr;
        int count = comp, which = -1;
        for (int i = 0; i < nBxDFs; ++i)
            if (bxdfs[i].MatchesFlags(type) && count-- == 0) { bxdf = &bxdfs[i]; which = i; break; }
        P2 uRemapped(std::min(u.x * matchingComps - comp, OneMinusEpsilon), u.y);
        V3 wi, wo = WorldToLocal(woWorld);
        if (wo.z == 0) { *pdf = 0; return Spec(0.); }
        *pdf = 0;
        if (sampledType) *sampledType = bxdf->type;
        Spec f = bxdf->Sample_f(wo, &wi, uRemapped, pdf, sampledType);
        if (*pdf == 0) { if (sampledType) *sampledType = 0; return Spec(0); }
        *wiWorld = LocalToWorld(wi);
        if (!(bxdf->type & BSDF_SPECULAR) && matchingComps > 1)
            for (int i = 0; i < nBxDFs; ++i)
                if (i != which && bxdfs[i].MatchesFlags(type)) *pdf += bxdfs[i].Pdf(wo, wi);
        if (matchingComps > 1) *pdf /= matchingComps;
        if (!(bxdf->type & BSDF_SPECULAR)) {
            bool reflect = Dot(*wiWorld, ng) * Dot(woWorld, ng) > 0;
            f = Spec(0.);
            for (int i = 0; i < nBxDFs; ++i)
                if (bxdfs[i].MatchesFlags(type) &&
                    ((reflect && (bxdfs[i].type & BSDF_REFLECTION)) || (!reflect && (bxdfs[i].type & BSDF_TRANSMISSION))))
                    f += bxdfs[i].f(wo, wi);
        }
        return f;
    }
};

inline Spec S3(const float *p) { return Spec(p[0], p[1], p[2]); }

// Material::Bump with a ConstantTexture<float>(0) displacement, core/Material.cpp:16-52.
// uDisplace = vDisplace = displace = 0, du = dv = .0005 (no differentials reach Path, Geometry.h:866),
// dndu = dndv = 0, so dpdu' = dpdu + (0-0)/du * n + 0 * dndu -- evaluated as written.
inline void Bump(SurfaceInteraction *si) {
    Float du = .0005f, dv = .0005f;
    Float uDisplace = 0, vDisplace = 0, displace = 0;
    V3 dpdu = si->sdpdu + (uDisplace - displace) / du * si->sn + displace * si->dndu;
    V3 dpdv = si->sdpdv + (vDisplace - displace) / dv * si->sn + displace * si->dndv;
    si->SetShadingGeometry(dpdu, dpdv, false);   // (dndu / dndv pass through unchanged)
}

// <Material>::ComputeScatteringFunctions(si, arena, TransportMode::Radiance, allowMultipleLobes)
// Returns false for a null material (no BSDF: PathIntegrator.cpp:121-126).
// SurfaceInteraction::ComputeScatteringFunctions(ray, ...) computes the uv differentials first (Interaction.cpp:56-63).
inline bool ComputeScatteringFunctions(const Scene &scene, const Ray &ray, SurfaceInteraction *si, bool allowMultipleLobes, BSDF *out) {
    si->ComputeDifferentials(ray);
    int mi = scene.triMaterial[si->prim];
    if (mi < 0) return false;
    const gnxr_material &m = scene.materials[mi];
    if (m.type == GNXR_MAT_NONE) return false;
    if (m.has_bump) Bump(si);
    switch (m.type) {
    case GNXR_MAT_MATTE: {  // MatteMaterial.cpp:14-32
        BSDF b(*si);
        Spec r = (m.kd_texture > 0 ? scene.textures[m.kd_texture - 1].Evaluate(si->uv, si->dudx, si->dvdx, si->dudy, si->dvdy) : S3(m.kd)).Clamp();
        Float sig = Clamp(m.sigma, 0, 90);
        if (!r.IsBlack()) {
            Lobe l;
            l.R = r;
            l.type = BSDF_REFLECTION | BSDF_DIFFUSE;
            if (sig == 0) l.kind = L_LAMBERT;
            else {  // OrenNayar ctor, Reflection.h:236-243
                l.kind = L_OREN;
                Float sigma = Radians(sig);
                Float sigma2 = sigma * sigma;
                l.A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                l.B = 0.45f * sigma2 / (sigma2 + 0.09f);
            }
            b.Add(l);
        }
        *out = b;
        return true;
    }
    case GNXR_MAT_MIRROR: {  // MirrorMaterial.cpp:13-24
        BSDF b(*si);
        Spec R = S3(m.kr).Clamp();
        if (!R.IsBlack()) {
            Lobe l; l.kind = L_SPEC_REFL; l.type = BSDF_REFLECTION | BSDF_SPECULAR; l.R = R; l.fresnel.kind = F_NOOP;
            b.Add(l);
        }
        *out = b;
        return true;
    }
    case GNXR_MAT_GLASS: {  // GlassMaterial.cpp:14-61
        Float eta = m.eta[0];
        Float urough = m.urough, vrough = m.vrough;
        Spec R = S3(m.kr).Clamp(), T = S3(m.kt).Clamp();
        BSDF b(*si, eta);
        if (R.IsBlack() && T.IsBlack()) { *out = b; return true; }
        bool isSpecular = urough == 0 && vrough == 0;
        if (isSpecular && allowMultipleLobes) {
            Lobe l; l.kind = L_FRESNEL_SPEC; l.type = BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR;
            l.R = R; l.T = T; l.etaA = 1.f; l.etaB = eta;
            b.Add(l);
        } else {
            if (m.remap_roughness) { urough = TRDist::RoughnessToAlpha(urough); vrough = TRDist::RoughnessToAlpha(vrough); }
            TRDist dist; if (!isSpecular) dist.Set(urough, vrough);
            if (!R.IsBlack()) {
                Lobe l; l.R = R; l.fresnel.kind = F_DIELECTRIC; l.fresnel.etaI = 1.f; l.fresnel.etaT = eta;
                if (isSpecular) { l.kind = L_SPEC_REFL; l.type = BSDF_REFLECTION | BSDF_SPECULAR; }
                else { l.kind = L_MICRO_REFL; l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.dist = dist; }
                b.Add(l);
            }
            if (!T.IsBlack()) {
                Lobe l; l.T = T; l.etaA = 1.f; l.etaB = eta;
                l.fresnel.kind = F_DIELECTRIC; l.fresnel.etaI = 1.f; l.fresnel.etaT = eta;  // fresnel(etaA, etaB), Reflection.h
                if (isSpecular) { l.kind = L_SPEC_TRANS; l.type = BSDF_TRANSMISSION | BSDF_SPECULAR; }
                else { l.kind = L_MICRO_TRANS; l.type = BSDF_TRANSMISSION | BSDF_GLOSSY; l.dist = dist; }
                b.Add(l);
            }
        }
        *out = b;
        return true;
    }
    case GNXR_MAT_METAL: {  // MetalMaterial.cpp:28-49
        BSDF b(*si);
        Float uRough = m.urough, vRough = m.vrough;
        if (m.remap_roughness) { uRough = TRDist::RoughnessToAlpha(uRough); vRough = TRDist::RoughnessToAlpha(vRough); }
        Lobe l; l.kind = L_MICRO_REFL; l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.R = Spec(1.);
        l.fresnel.kind = F_CONDUCTOR; l.fresnel.cEtaI = Spec(1.); l.fresnel.cEtaT = S3(m.eta); l.fresnel.cK = S3(m.k);
        l.dist.Set(uRough, vRough);
        b.Add(l);
        *out = b;
        return true;
    }
    case GNXR_MAT_PLASTIC: {  // PlasticMaterial.cpp:15-41
        BSDF b(*si);
        auto tex = [&](int t, const float *c) { return t > 0 ? scene.textures[t - 1].Evaluate(si->uv, si->dudx, si->dvdx, si->dudy, si->dvdy) : S3(c); };
        Spec kd = tex(m.kd_texture, m.kd).Clamp();
        if (!kd.IsBlack()) { Lobe l; l.kind = L_LAMBERT; l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.R = kd; b.Add(l); }
        Spec ks = tex(m.ks_texture, m.ks).Clamp();
        if (!ks.IsBlack()) {
            Lobe l; l.kind = L_MICRO_REFL; l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.R = ks;
            l.fresnel.kind = F_DIELECTRIC; l.fresnel.etaI = 1.5f; l.fresnel.etaT = 1.f;
            Float rough = m.urough;
            if (m.remap_roughness) rough = TRDist::RoughnessToAlpha(rough);
            l.dist.Set(rough, rough);
            b.Add(l);
        }
        *out = b;
        return true;
    }
    case GNXR_MAT_DISNEY: {  // DisneyMaterial.cpp:467-581 (scatterDistance must be black: BSSRDF branch out of scope)
        BSDF b(*si);
        Spec c = S3(m.kd).Clamp();
        Float metallicWeight = m.disney_metallic;
        Float e = m.eta[0];
        Float strans = m.disney_spec_trans;
        Float diffuseWeight = (1 - metallicWeight) * (1 - strans);
        Float dt = m.disney_diff_trans / 2;
        Float rough = m.disney_roughness;
        Float lum = c.y();
        Spec Ctint = lum > 0 ? (c / lum) : Spec(1.);
        Float sheenWeight = m.disney_sheen;
        Spec Csheen;
        if (sheenWeight > 0) { Float stint = m.disney_sheen_tint; Csheen = Lerp(stint, Spec(1.), Ctint); }
        bool thin = m.disney_thin != 0;
        if (diffuseWeight > 0) {
            if (thin) {
                Float flat = m.disney_flatness;
                Lobe l; l.kind = L_DISNEY_DIFFUSE; l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.R = diffuseWeight * (1 - flat) * (1 - dt) * c; b.Add(l);
                Lobe l2; l2.kind = L_DISNEY_FAKESS; l2.type = BSDF_REFLECTION | BSDF_DIFFUSE; l2.R = diffuseWeight * flat * (1 - dt) * c; l2.roughness = rough; b.Add(l2);
            } else {
                Lobe l; l.kind = L_DISNEY_DIFFUSE; l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.R = diffuseWeight * c; b.Add(l);
            }
            Lobe lr; lr.kind = L_DISNEY_RETRO; lr.type = BSDF_REFLECTION | BSDF_DIFFUSE; lr.R = diffuseWeight * c; lr.roughness = rough; b.Add(lr);
            if (sheenWeight > 0) { Lobe ls; ls.kind = L_DISNEY_SHEEN; ls.type = BSDF_REFLECTION | BSDF_DIFFUSE; ls.R = diffuseWeight * sheenWeight * Csheen; b.Add(ls); }
        }
        Float aspect = std::sqrt(1 - m.disney_anisotropic * .9);  // double expression, DisneyMaterial.cpp:527
        Float ax = std::max(Float(.001), sqr(rough) / aspect);
        Float ay = std::max(Float(.001), sqr(rough) * aspect);
        TRDist dist; dist.Set(ax, ay); dist.disneyG = true;
        Float specTint = m.disney_spec_tint;
        Spec Cspec0 = Lerp(metallicWeight, SchlickR0FromEta(e) * Lerp(specTint, Spec(1.), Ctint), c);
        {
            Lobe l; l.kind = L_MICRO_REFL; l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.R = Spec(1.);
            l.fresnel.kind = F_DISNEY; l.fresnel.R0 = Cspec0; l.fresnel.metallic = metallicWeight; l.fresnel.eta = e;
            l.dist = dist; b.Add(l);
        }
        Float cc = m.disney_clearcoat;
        if (cc > 0) {
            Lobe l; l.kind = L_DISNEY_CLEARCOAT; l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.weight = cc;
            l.gloss = Lerp(m.disney_clearcoat_gloss, .1, .001); b.Add(l);
        }
        if (strans > 0) {
            Spec T = strans * Sqrt(c);
            Lobe l; l.kind = L_MICRO_TRANS; l.type = BSDF_TRANSMISSION | BSDF_GLOSSY; l.T = T; l.etaA = 1.; l.etaB = e;
            l.fresnel.kind = F_DIELECTRIC; l.fresnel.etaI = 1.; l.fresnel.etaT = e;
            if (thin) {
                Float rscaled = (0.65f * e - 0.35f) * rough;
                Float ax2 = std::max(Float(.001), sqr(rscaled) / aspect);
                Float ay2 = std::max(Float(.001), sqr(rscaled) * aspect);
                l.dist.Set(ax2, ay2); l.dist.disneyG = false;
            } else l.dist = dist;
            b.Add(l);
        }
        if (thin) { Lobe l; l.kind = L_LAMBERT_TRANS; l.type = BSDF_TRANSMISSION | BSDF_DIFFUSE; l.T = dt * c; b.Add(l); }
        *out = b;
        return true;
    }
    }
    return false;
}

}  // namespace gnxo

// ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_texture.h: CPU restatement of the image-texture stack:
//   MIPMap ctor (Lanczos resample to pow2, pyramid, EWA weights) / Lookup / triangle / EWA / Texel   core/MIPMap.h:85-337
//   Lanczos                                                    core/Texture.cpp:152-161
//   ImageTexture ctor / GetTexture / convertIn / Evaluate      textures/ImageTexture.cpp:40-109, ImageTexture.h:56-86
//   UVMapping2D::Map                                           core/Texture.cpp:168-175
//   InverseGammaCorrect                                        core/GNXRayTracer.h:367-371
#pragma once
#include <vector>

#include "../include/gnxr.h"
#include "o_math.h"

namespace gnxo {

inline Float Lanczos(Float x, Float tau = 2) {
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x *= Pi;
    Float s = std::sin(x * tau) / (x * tau);
    Float lanczos = std::sin(x) / x;
    return s * lanczos;
}
inline bool IsPowerOf2(int v) { return v && !(v & (v - 1)); }
inline int RoundUpPow2(int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
inline int Log2Int(uint32_t v) { return 31 - __builtin_clz(v); }
inline Float Log2(Float x) { const Float invLog2 = 1.442695040888963387004650940071; return std::log(x) * invLog2; }
inline int ModI(int a, int b) { int r = a - (a / b) * b; return (r < 0) ? r + b : r; }

// MIPMap<RGBSpectrum>, core/MIPMap.h:41-337.  The InfiniteAreaLight uses (doTrilinear = false, Repeat) and only the
// width-0 Lookup; ImageTexture uses the (st, dst0, dst1) Lookup: trilinear or EWA.
struct MIPMapRGB {
    int resX = 0, resY = 0;
    bool doTrilinear = false;
    Float maxAnisotropy = 8.f;
    int wrapMode = GNXR_WRAP_REPEAT;
    static constexpr int WeightLUTSize = 128;
    static const Float *WeightLut() {   // MIPMap.h:191-198
        static Float lut[WeightLUTSize];
        static bool init = [] {
            for (int i = 0; i < WeightLUTSize; ++i) {
                Float alpha = 2;
                Float r2 = Float(i) / Float(WeightLUTSize - 1);
                lut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
            }
            return true;
        }();
        (void)init;
        return lut;
    }
    int WrapIndex(int v, int res) const {   // the wrap step of the resampling loops, MIPMap.h:118-121, 148-151
        if (wrapMode == GNXR_WRAP_REPEAT) return ModI(v, res);
        if (wrapMode == GNXR_WRAP_CLAMP) return Clamp(v, 0, res - 1);
        return v;
    }
    std::vector<std::vector<Spec>> pyramid;
    std::vector<int> lw, lh;
    struct ResampleWeight { int firstTexel; Float weight[4]; };
    static std::vector<ResampleWeight> resampleWeights(int oldRes, int newRes) {
        std::vector<ResampleWeight> wt(newRes);
        Float filterwidth = 2.f;
        for (int i = 0; i < newRes; ++i) {
            Float center = (i + .5f) * oldRes / newRes;
            wt[i].firstTexel = std::floor((center - filterwidth) + 0.5f);
            for (int j = 0; j < 4; ++j) {
                Float pos = wt[i].firstTexel + j + .5f;
                wt[i].weight[j] = Lanczos((pos - center) / filterwidth);
            }
            Float invSumWts = 1 / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
            for (int j = 0; j < 4; ++j) wt[i].weight[j] *= invSumWts;
        }
        return wt;
    }
    void Build(int rx, int ry, const Spec *img) {
        resX = rx; resY = ry;
        std::vector<Spec> resampled;
        if (!IsPowerOf2(resX) || !IsPowerOf2(resY)) {
            int px = RoundUpPow2(resX), py = RoundUpPow2(resY);
            std::vector<ResampleWeight> sWeights = resampleWeights(resX, px);
            resampled.assign((size_t)px * py, Spec(0.f));
            for (int64_t t = 0; t < resY; ++t)
                for (int s = 0; s < px; ++s) {
                    resampled[t * px + s] = Spec(0.f);
                    for (int j = 0; j < 4; ++j) {
                        int origS = sWeights[s].firstTexel + j;
                        origS = WrapIndex(origS, resX);
                        if (origS >= 0 && origS < resX) resampled[t * px + s] += sWeights[s].weight[j] * img[t * resX + origS];
                    }
                }
            std::vector<ResampleWeight> tWeights = resampleWeights(resY, py);
            std::vector<Spec> workData(py);
            for (int64_t s = 0; s < px; ++s) {
                for (int t = 0; t < py; ++t) {
                    workData[t] = Spec(0.f);
                    for (int j = 0; j < 4; ++j) {
                        int offset = tWeights[t].firstTexel + j;
                        offset = WrapIndex(offset, resY);
                        if (offset >= 0 && offset < resY) workData[t] += tWeights[t].weight[j] * resampled[offset * px + s];
                    }
                }
                for (int t = 0; t < py; ++t) resampled[t * px + s] = workData[t].Clamp(0.f, Infinity);
            }
            resX = px; resY = py;
        }
        int nLevels = 1 + Log2Int(std::max(resX, resY));
        pyramid.resize(nLevels); lw.resize(nLevels); lh.resize(nLevels);
        lw[0] = resX; lh[0] = resY;
        if (!resampled.empty()) pyramid[0] = resampled;
        else pyramid[0].assign(img, img + (size_t)resX * resY);
        for (int i = 1; i < nLevels; ++i) {
            int sRes = std::max(1, lw[i - 1] / 2), tRes = std::max(1, lh[i - 1] / 2);
            lw[i] = sRes; lh[i] = tRes;
            pyramid[i].resize((size_t)sRes * tRes);
            for (int t = 0; t < tRes; t++)
                for (int s = 0; s < sRes; ++s)
                    pyramid[i][t * sRes + s] = .25f * (Texel(i - 1, 2 * s, 2 * t) + Texel(i - 1, 2 * s + 1, 2 * t) +
                                                       Texel(i - 1, 2 * s, 2 * t + 1) + Texel(i - 1, 2 * s + 1, 2 * t + 1));
        }
    }
    int Levels() const { return (int)pyramid.size(); }
    const Spec &Texel(int level, int s, int t) const {   // MIPMap.h:203-223
        static const Spec black(0.f);
        switch (wrapMode) {
        case GNXR_WRAP_REPEAT: s = ModI(s, lw[level]); t = ModI(t, lh[level]); break;
        case GNXR_WRAP_CLAMP: s = Clamp(s, 0, lw[level] - 1); t = Clamp(t, 0, lh[level] - 1); break;
        default: if (s < 0 || s >= lw[level] || t < 0 || t >= lh[level]) return black; break;
        }
        return pyramid[level][(size_t)t * lw[level] + s];
    }
    Spec triangle(int level, const P2 &st) const {
        level = Clamp(level, 0, Levels() - 1);
        Float s = st.x * lw[level] - 0.5f;
        Float t = st.y * lh[level] - 0.5f;
        int s0 = std::floor(s), t0 = std::floor(t);
        Float ds = s - s0, dt = t - t0;
        return (1 - ds) * (1 - dt) * Texel(level, s0, t0) + (1 - ds) * dt * Texel(level, s0, t0 + 1) +
               ds * (1 - dt) * Texel(level, s0 + 1, t0) + ds * dt * Texel(level, s0 + 1, t0 + 1);
    }
    // MIPMap.h:258-286
    Spec Lookup(const P2 &st, P2 dst0, P2 dst1) const {
        if (doTrilinear) {
            Float width = std::max(std::max(std::abs(dst0.x), std::abs(dst0.y)), std::max(std::abs(dst1.x), std::abs(dst1.y)));
            return Lookup(st, width);
        }
        if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) std::swap(dst0, dst1);
        Float majorLength = std::sqrt(dst0.x * dst0.x + dst0.y * dst0.y);
        Float minorLength = std::sqrt(dst1.x * dst1.x + dst1.y * dst1.y);
        if (minorLength * maxAnisotropy < majorLength && minorLength > 0) {
            Float scale = majorLength / (minorLength * maxAnisotropy);
            dst1.x *= scale; dst1.y *= scale;
            minorLength *= scale;
        }
        if (minorLength == 0) return triangle(0, st);
        Float lod = std::max((Float)0, Levels() - (Float)1 + Log2(minorLength));
        int ilod = std::floor(lod);
        return Lerp(lod - ilod, EWA(ilod, st, dst0, dst1), EWA(ilod + 1, st, dst0, dst1));
    }
    // MIPMap.h:288-334
    Spec EWA(int level, P2 st, P2 dst0, P2 dst1) const {
        if (level >= Levels()) return Texel(Levels() - 1, 0, 0);
        st.x = st.x * lw[level] - 0.5f;
        st.y = st.y * lh[level] - 0.5f;
        dst0.x *= lw[level]; dst0.y *= lh[level];
        dst1.x *= lw[level]; dst1.y *= lh[level];
        Float A = dst0.y * dst0.y + dst1.y * dst1.y + 1;
        Float B = -2 * (dst0.x * dst0.y + dst1.x * dst1.y);
        Float C = dst0.x * dst0.x + dst1.x * dst1.x + 1;
        Float invF = 1 / (A * C - B * B * 0.25f);
        A *= invF; B *= invF; C *= invF;
        Float det = -B * B + 4 * A * C;
        Float invDet = 1 / det;
        Float uSqrt = std::sqrt(det * C), vSqrt = std::sqrt(A * det);
        int s0 = std::ceil(st.x - 2 * invDet * uSqrt);
        int s1 = std::floor(st.x + 2 * invDet * uSqrt);
        int t0 = std::ceil(st.y - 2 * invDet * vSqrt);
        int t1 = std::floor(st.y + 2 * invDet * vSqrt);
        Spec sum(0.f);
        Float sumWts = 0;
        const Float *weightLut = WeightLut();
        for (int it = t0; it <= t1; ++it) {
            Float tt = it - st.y;
            for (int is = s0; is <= s1; ++is) {
                Float ss = is - st.x;
                Float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
                if (r2 < 1) {
                    int index = std::min((int)(r2 * WeightLUTSize), WeightLUTSize - 1);
                    Float weight = weightLut[index];
                    sum += Texel(level, is, it) * weight;
                    sumWts += weight;
                }
            }
        }
        return sum / sumWts;
    }
    Spec Lookup(const P2 &st, Float width = 0.f) const {
        Float level = Levels() - 1 + Log2(std::max(width, (Float)1e-8));
        if (level < 0) return triangle(0, st);
        else if (level >= Levels() - 1) return Texel(Levels() - 1, 0, 0);
        else {
            int iLevel = std::floor(level);
            Float delta = level - iLevel;
            return Lerp(delta, triangle(iLevel, st), triangle(iLevel + 1, st));
        }
    }
};

// ImageTexture<RGBSpectrum, Spectrum> over UVMapping2D
struct ImageTexture {
    gnxr_texture t;
    MIPMapRGB mipmap;
    static Float InverseGammaCorrect(Float value) {
        if (value <= 0.04045f) return value * 1.f / 12.92f;
        return std::pow((value + 0.055f) * 1.f / 1.055f, (Float)2.4f);
    }
    // GetTexture, ImageTexture.cpp:50-106: y flip, convertIn, MIPMap
    void Build(const gnxr_texture &tex, const float *rgb) {
        t = tex;
        const int w = tex.width, h = tex.height;
        std::vector<Spec> texels((size_t)w * h);
        for (size_t i = 0; i < texels.size(); ++i) texels[i] = Spec(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
        for (int y = 0; y < h / 2; ++y)
            for (int x = 0; x < w; ++x) std::swap(texels[(size_t)y * w + x], texels[(size_t)(h - 1 - y) * w + x]);
        for (Spec &s : texels)
            for (int c = 0; c < 3; ++c) s[c] = tex.scale * (tex.gamma ? InverseGammaCorrect(s[c]) : s[c]);
        mipmap.doTrilinear = tex.trilinear != 0;
        mipmap.maxAnisotropy = tex.max_aniso;
        mipmap.wrapMode = tex.wrap;
        mipmap.Build(w, h, texels.data());
    }
    // Evaluate with UVMapping2D::Map inlined: uv and the four uv differentials of the SurfaceInteraction
    Spec Evaluate(const P2 &uv, Float dudx, Float dvdx, Float dudy, Float dvdy) const {
        P2 dstdx(t.su * dudx, t.sv * dvdx), dstdy(t.su * dudy, t.sv * dvdy);
        P2 st(t.su * uv.x + t.du, t.sv * uv.y + t.dv);
        return mipmap.Lookup(st, dstdx, dstdy);   // convertOut: ToRGB / FromRGB are the identity for RGBSpectrum
    }
};

}  // namespace gnxo

// ORACLE -- TEST INFRASTRUCTURE ONLY (see o_math.h header).
//
// o_sampler.h: CPU restatement of the reference's QMC stream:
//   RNG (PCG32)                         core/RNG.h:30-110
//   Shuffle                             core/Sampling.h:129-137
//   ComputeRadicalInversePermutations   samplers/LowDiscrepancy.cpp:2459-2473
//   RadicalInverse / ScrambledRadicalInverse   samplers/LowDiscrepancy.cpp:358-393, 396-2457, 2475-4532
//   HaltonSampler                       samplers/HaltonSampler.cpp:33-99
//   GlobalSampler / Sampler             core/Sampler.cpp:14-20, 116-179
// The 1000-entry Primes / PrimeSums tables (LowDiscrepancy.cpp:9,93) are regenerated with a
// sieve instead of being copied.
#pragma once
#include <atomic>
#include <vector>

#include "o_math.h"

namespace gnxo {

static constexpr int PrimeTableSize = 1000;  // samplers/LowDiscrepancy.h:17

struct PrimeTables {
    int primes[PrimeTableSize];
    int primeSums[PrimeTableSize];
    PrimeTables() {
        int n = 0;
        for (int c = 2; n < PrimeTableSize; ++c) {
            bool isPrime = true;
            for (int d = 2; d * d <= c; ++d)
                if (c % d == 0) { isPrime = false; break; }
            if (isPrime) primes[n++] = c;
        }
        int sum = 0;
        for (int i = 0; i < PrimeTableSize; ++i) { primeSums[i] = sum; sum += primes[i]; }
    }
};
inline const PrimeTables &Primes() { static PrimeTables t; return t; }

// core/RNG.h:26-110
struct RNG {
    uint64_t state, inc;
    RNG() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
    explicit RNG(uint64_t seq) { SetSequence(seq); }
    void SetSequence(uint64_t initseq) {
        state = 0u;
        inc = (initseq << 1u) | 1u;
        UniformUInt32();
        state += 0x853c49e6748fea9bULL;
        UniformUInt32();
    }
    uint32_t UniformUInt32() {
        uint64_t oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
        uint32_t rot = (uint32_t)(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    uint32_t UniformUInt32(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        while (true) {
            uint32_t r = UniformUInt32();
            if (r >= threshold) return r % b;
        }
    }
    Float UniformFloat() { return std::min(OneMinusEpsilon, Float(UniformUInt32() * 2.3283064365386963e-10f)); }
};

// samplers/LowDiscrepancy.cpp:2459-2473 with Shuffle (core/Sampling.h:129-137), default-seeded
// RNG as in HaltonSampler.cpp:36-39.
inline const std::vector<uint16_t> &RadicalInversePermutations() {
    static std::vector<uint16_t> perms = [] {
        const PrimeTables &pt = Primes();
        int permArraySize = 0;
        for (int i = 0; i < PrimeTableSize; ++i) permArraySize += pt.primes[i];
        std::vector<uint16_t> v(permArraySize);
        RNG rng;
        uint16_t *p = v.data();
        for (int i = 0; i < PrimeTableSize; ++i) {
            int count = pt.primes[i];
            for (int j = 0; j < count; ++j) p[j] = j;
            for (int j = 0; j < count; ++j) {
                int other = j + rng.UniformUInt32(count - j);
                std::swap(p[j], p[other]);
            }
            p += count;
        }
        return v;
    }();
    return perms;
}

// samplers/LowDiscrepancy.h:31-45
inline uint32_t ReverseBits32(uint32_t n) {
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
    n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
    n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
    n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
    return n;
}
inline uint64_t ReverseBits64(uint64_t n) {
    uint64_t n0 = ReverseBits32((uint32_t)n);
    uint64_t n1 = ReverseBits32((uint32_t)(n >> 32));
    return (n0 << 32) | n1;
}

// samplers/LowDiscrepancy.cpp:358-372 (template<int base> there; base is a run-time value here,
// integer division is exact either way).
inline Float RadicalInverseBase(int base, uint64_t a) {
    const Float invBase = (Float)1 / (Float)base;
    uint64_t reversedDigits = 0;
    Float invBaseN = 1;
    while (a) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reversedDigits = reversedDigits * base + digit;
        invBaseN *= invBase;
        a = next;
    }
    return std::min(reversedDigits * invBaseN, OneMinusEpsilon);
}
// samplers/LowDiscrepancy.cpp:374-393
inline Float ScrambledRadicalInverseBase(int base, const uint16_t *perm, uint64_t a) {
    const Float invBase = (Float)1 / (Float)base;
    uint64_t reversedDigits = 0;
    Float invBaseN = 1;
    while (a) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reversedDigits = reversedDigits * base + perm[digit];
        invBaseN *= invBase;
        a = next;
    }
    return std::min(invBaseN * (reversedDigits + invBase * perm[0] / (1 - invBase)), OneMinusEpsilon);
}
// samplers/LowDiscrepancy.cpp:396-403: base 2 goes through a double product.
inline Float RadicalInverse(int baseIndex, uint64_t a) {
    if (baseIndex == 0) return ReverseBits64(a) * 5.4210108624275222e-20;
    return RadicalInverseBase(Primes().primes[baseIndex], a);
}
inline Float ScrambledRadicalInverse(int baseIndex, uint64_t a, const uint16_t *perm) {
    return ScrambledRadicalInverseBase(Primes().primes[baseIndex], perm, a);
}

// samplers/LowDiscrepancy.h:47-56
inline uint64_t InverseRadicalInverse(int base, uint64_t inverse, int nDigits) {
    uint64_t index = 0;
    for (int i = 0; i < nDigits; ++i) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}

inline int64_t Mod64(int64_t a, int64_t b) { int64_t r = a - (a / b) * b; return (r < 0) ? r + b : r; }  // GNXRayTracer.h Mod
// samplers/HaltonSampler.cpp:12-31
inline void extendedGCD(uint64_t a, uint64_t b, int64_t *x, int64_t *y) {
    if (b == 0) { *x = 1; *y = 0; return; }
    int64_t d = a / b, xp, yp;
    extendedGCD(b, a % b, &xp, &yp);
    *x = yp;
    *y = xp - (d * yp);
}
inline uint64_t multiplicativeInverse(int64_t a, int64_t n) {
    int64_t x, y;
    extendedGCD(a, n, &x, &y);
    return Mod64(x, n);
}

// HaltonSampler + GlobalSampler state for ONE pixel sample: (index, dimension counter).
struct Halton {
    static constexpr int kMaxResolution = 128;  // HaltonSampler.cpp:10
    int baseScales[2], baseExponents[2];
    int sampleStride;
    int multInverse[2];
    int64_t samplesPerPixel;
    bool sampleAtPixelCenter;
    // HaltonSampler.cpp:33-60
    Halton(int64_t spp, int resX, int resY, bool center = false) : samplesPerPixel(spp), sampleAtPixelCenter(center) {
        int res[2] = {resX, resY};
        for (int i = 0; i < 2; ++i) {
            int base = (i == 0) ? 2 : 3;
            int scale = 1, exp = 0;
            while (scale < std::min(res[i], kMaxResolution)) { scale *= base; ++exp; }
            baseScales[i] = scale;
            baseExponents[i] = exp;
        }
        sampleStride = baseScales[0] * baseScales[1];
        multInverse[0] = (int)multiplicativeInverse(baseScales[1], baseScales[0]);
        multInverse[1] = (int)multiplicativeInverse(baseScales[0], baseScales[1]);
    }
    // HaltonSampler.cpp:63-83
    int64_t OffsetForPixel(int px, int py) const {
        int64_t offset = 0;
        if (sampleStride > 1) {
            int pm[2] = {(int)Mod64(px, kMaxResolution), (int)Mod64(py, kMaxResolution)};
            for (int i = 0; i < 2; ++i) {
                uint64_t dimOffset = (i == 0) ? InverseRadicalInverse(2, pm[i], baseExponents[i])
                                              : InverseRadicalInverse(3, pm[i], baseExponents[i]);
                offset += dimOffset * (sampleStride / baseScales[i]) * multInverse[i];
            }
            offset %= sampleStride;
        }
        return offset;
    }
    int64_t IndexForSample(int px, int py, int64_t sampleNum) const {
        return OffsetForPixel(px, py) + sampleNum * sampleStride;
    }
    // HaltonSampler.cpp:85-94.  Dimensions >= PrimeTableSize are undefined in the reference
    // (PrimeSums read out of bounds, HaltonSampler.h:37-42); this build defines them to wrap
    // into [2, PrimeTableSize).
    Float SampleDimension(int64_t index, int dim) const {
        if (sampleAtPixelCenter && (dim == 0 || dim == 1)) return 0.5f;
        if (dim >= PrimeTableSize) dim = 2 + (dim - 2) % (PrimeTableSize - 2);
        if (dim == 0) return RadicalInverse(dim, index >> baseExponents[0]);
        else if (dim == 1) return RadicalInverse(dim, index / baseScales[1]);
        else return ScrambledRadicalInverse(dim, index, &RadicalInversePermutations()[Primes().primeSums[dim]]);
    }
};

// The per-sample view the integrators consume (GlobalSampler::Get1D/Get2D, core/Sampler.cpp:162-179;
// no sample arrays are ever requested so arrayStartDim..arrayEndDim is empty).
struct SampleStream {
    const Halton *h;
    int64_t index;
    int dimension;
    // array samples, core/Sampler.cpp:52-72 + GlobalSampler::StartPixel :116-146: the 2D arrays requested by the integrator's
    // Preprocess occupy dimensions [arrayStartDim, arrayEndDim) and the regular stream skips that range (:161-179).  No
    // integrator on this path requests 1D arrays.  Element k of array i for pixel sample s is drawn from Halton index
    // GetIndexForSample(s * n + k), dimensions arrayStartDim + 2i and + 2i + 1.
    static constexpr int arrayStartDim = 5;   // core/Sampler.h:92
    int arrayEndDim = arrayStartDim;
    int px = 0, py = 0;
    int64_t sampleNum = 0;
    const std::vector<int> *array2DSizes = nullptr;
    size_t array2DOffset = 0;
    SampleStream(const Halton *h, int px, int py, int64_t s) : h(h), index(h->IndexForSample(px, py, s)), dimension(0), px(px), py(py), sampleNum(s) {}
    void Request2DArrays(const std::vector<int> *sizes) {
        array2DSizes = sizes;
        arrayEndDim = arrayStartDim + 2 * (int)sizes->size();
    }
    // Sampler::Get2DArray, core/Sampler.cpp:67-72; false stands for the nullptr return (all requested arrays consumed)
    bool Get2DArray(int n, std::vector<P2> *out) {
        if (!array2DSizes || array2DOffset == array2DSizes->size()) return false;
        int i = (int)array2DOffset++;
        out->resize(n);
        for (int k = 0; k < n; ++k) {
            int64_t idx = h->IndexForSample(px, py, sampleNum * n + k);
            (*out)[k] = P2(h->SampleDimension(idx, arrayStartDim + 2 * i), h->SampleDimension(idx, arrayStartDim + 2 * i + 1));
        }
        return true;
    }
    ~SampleStream() {  // highest dimension any sample used: fixtures must stay below PrimeTableSize (reference UB beyond)
        int cur = MaxDimensionSeen().load(std::memory_order_relaxed);
        while (dimension > cur && !MaxDimensionSeen().compare_exchange_weak(cur, dimension)) {}
    }
    static std::atomic<int> &MaxDimensionSeen() { static std::atomic<int> m{0}; return m; }
    Float Get1D() {
        if (dimension >= arrayStartDim && dimension < arrayEndDim) dimension = arrayEndDim;
        return h->SampleDimension(index, dimension++);
    }
    P2 Get2D() {
        if (dimension + 1 >= arrayStartDim && dimension < arrayEndDim) dimension = arrayEndDim;
        P2 p(h->SampleDimension(index, dimension), h->SampleDimension(index, dimension + 1));
        dimension += 2;
        return p;
    }
};

// ---- warps, core/Sampling.{h,cpp} ----
// Sampling.cpp:87-105
inline P2 ConcentricSampleDisk(const P2 &u) {
    P2 uOffset(2.f * u.x - 1, 2.f * u.y - 1);
    if (uOffset.x == 0 && uOffset.y == 0) return P2(0, 0);
    Float theta, r;
    if (std::abs(uOffset.x) > std::abs(uOffset.y)) {
        r = uOffset.x;
        theta = PiOver4 * (uOffset.y / uOffset.x);
    } else {
        r = uOffset.y;
        theta = PiOver2 - PiOver4 * (uOffset.x / uOffset.y);
    }
    return P2(r * std::cos(theta), r * std::sin(theta));
}
// Sampling.h:140-145
inline V3 CosineSampleHemisphere(const P2 &u) {
    P2 d = ConcentricSampleDisk(u);
    Float z = std::sqrt(std::max((Float)0, 1 - d.x * d.x - d.y * d.y));
    return V3(d.x, d.y, z);
}
// Sampling.cpp:131-135
inline P2 UniformSampleTriangle(const P2 &u) {
    Float su0 = std::sqrt(u.x);
    return P2(1 - su0, u.y * su0);
}
// Sampling.h:157-161
inline Float PowerHeuristic(int nf, Float fPdf, int ng, Float gPdf) {
    Float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
inline V3 UniformSampleSphere(const P2 &u) {
    Float z = 1 - 2 * u.x;
    Float r = std::sqrt(std::max((Float)0, (Float)1 - z * z));
    Float phi = 2 * Pi * u.y;
    return V3(r * std::cos(phi), r * std::sin(phi), z);
}

// GNXRayTracer.h:336-349
template <typename Predicate>
inline int FindInterval(int size, const Predicate &pred) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (pred(middle)) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    return Clamp(first - 1, 0, size - 2);
}

// core/Sampling.h:19-70
struct Distribution1D {
    std::vector<Float> func, cdf;
    Float funcInt;
    Distribution1D() : funcInt(0) {}
    Distribution1D(const Float *f, int n) : func(f, f + n), cdf(n + 1) {
        cdf[0] = 0;
        for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
        funcInt = cdf[n];
        if (funcInt == 0) { for (int i = 1; i < n + 1; ++i) cdf[i] = Float(i) / Float(n); }
        else { for (int i = 1; i < n + 1; ++i) cdf[i] /= funcInt; }
    }
    int Count() const { return (int)func.size(); }
    Float SampleContinuous(Float u, Float *pdf, int *off = nullptr) const {
        int offset = FindInterval((int)cdf.size(), [&](int index) { return cdf[index] <= u; });
        if (off) *off = offset;
        Float du = u - cdf[offset];
        if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
        if (pdf) *pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
        return (offset + du) / Count();
    }
    int SampleDiscrete(Float u, Float *pdf = nullptr) const {
        int offset = FindInterval((int)cdf.size(), [&](int index) { return cdf[index] <= u; });
        if (pdf) *pdf = (funcInt > 0) ? func[offset] / (funcInt * Count()) : 0;
        return offset;
    }
};
// core/Sampling.h:97-126, Sampling.cpp:137-150
struct Distribution2D {
    std::vector<Distribution1D> pConditionalV;
    Distribution1D pMarginal;
    Distribution2D() {}
    Distribution2D(const Float *func, int nu, int nv) {
        pConditionalV.reserve(nv);
        for (int v = 0; v < nv; ++v) pConditionalV.emplace_back(&func[v * nu], nu);
        std::vector<Float> marginalFunc;
        marginalFunc.reserve(nv);
        for (int v = 0; v < nv; ++v) marginalFunc.push_back(pConditionalV[v].funcInt);
        pMarginal = Distribution1D(&marginalFunc[0], nv);
    }
    P2 SampleContinuous(const P2 &u, Float *pdf) const {
        Float pdfs[2];
        int v;
        Float d1 = pMarginal.SampleContinuous(u.y, &pdfs[1], &v);
        Float d0 = pConditionalV[v].SampleContinuous(u.x, &pdfs[0]);
        *pdf = pdfs[0] * pdfs[1];
        return P2(d0, d1);
    }
    Float Pdf(const P2 &p) const {
        int iu = Clamp(int(p.x * pConditionalV[0].Count()), 0, pConditionalV[0].Count() - 1);
        int iv = Clamp(int(p.y * pMarginal.Count()), 0, pMarginal.Count() - 1);
        return pConditionalV[iv].func[iu] / pMarginal.funcInt;
    }
};

}  // namespace gnxo

// device_sampler.h -- the reference's Halton stream on the device, bit-exact.
//   HaltonSampler::GetIndexForSample / SampleDimension   samplers/HaltonSampler.cpp:63-94
//   RadicalInverse / ScrambledRadicalInverse             samplers/LowDiscrepancy.cpp:358-403, 2475-4532
//   warps                                                core/Sampling.cpp:87-135, core/Sampling.h:140-161
// A sample index never exceeds offset + spp * stride < 2^32 for the configs in scope (stride <= 31104,
// spp <= 131072), so digits are peeled with exact 32-bit magic-number division (2 VALU ops + shifts)
// instead of the reference's 64-bit `a / base`; the integer results are identical by construction and
// the float tail keeps the reference's operation order.
#pragma once
#include "device_math.h"
#include "gnxr_device_types.h"

namespace gnxr {

struct DSamplerTables {
    const uint16_t *perms;
    const int32_t *primes;
    const int32_t *prime_sums;
    const uint32_t *prime_magic;  // (M, s) per prime
    DHalton h;
};

// exact n / d for any n < 2^32 (Granlund-Montgomery round-up method)
GX_DEV uint32_t div_magic(uint32_t n, uint32_t M, uint32_t s) {
    uint32_t t = __umulhi(M, n);
    return (t + ((n - t) >> 1)) >> (s - 1);
}

// RadicalInverseSpecialized<base>, LowDiscrepancy.cpp:358-372
GX_DEV float radical_inverse_base(uint32_t a, uint32_t base, uint32_t M, uint32_t s) {
    const float invBase = 1.f / (float)base;
    uint64_t reversedDigits = 0;
    float invBaseN = 1;
    while (a) {
        uint32_t next = div_magic(a, M, s);
        uint32_t digit = a - next * base;
        reversedDigits = reversedDigits * base + digit;
        invBaseN *= invBase;
        a = next;
    }
    return fminf((float)reversedDigits * invBaseN, GX_ONE_MINUS_EPS);
}
// ScrambledRadicalInverseSpecialized<base>, LowDiscrepancy.cpp:374-393.  Digits are peeled four at a time so that the four
// permutation-table loads are independent and in flight together (the serial loop waited for one L1/L2 round trip per
// digit); the accumulation then runs in the reference's order, digit by digit, for as many digits as the index has.
GX_DEV float scrambled_radical_inverse_base(uint32_t a, uint32_t base, uint32_t M, uint32_t s, const uint16_t *__restrict__ perm) {
    const float invBase = 1.f / (float)base;
    uint64_t reversedDigits = 0;
    float invBaseN = 1;
    while (a) {
        const uint32_t n1 = div_magic(a, M, s), n2 = div_magic(n1, M, s), n3 = div_magic(n2, M, s), n4 = div_magic(n3, M, s);
        const uint32_t p0 = perm[a - n1 * base], p1 = perm[n1 - n2 * base], p2 = perm[n2 - n3 * base], p3 = perm[n3 - n4 * base];
        reversedDigits = reversedDigits * base + p0; invBaseN *= invBase;
        if (n1) { reversedDigits = reversedDigits * base + p1; invBaseN *= invBase; }
        if (n2) { reversedDigits = reversedDigits * base + p2; invBaseN *= invBase; }
        if (n3) { reversedDigits = reversedDigits * base + p3; invBaseN *= invBase; }
        a = n4;
    }
    return fminf(invBaseN * ((float)reversedDigits + invBase * (float)(int)perm[0] / (1 - invBase)), GX_ONE_MINUS_EPS);
}
// base 2: ReverseBits64(a) * 2^-64 evaluated in double, LowDiscrepancy.cpp:396-403.  For a < 2^32 the
// reversed word is brev(a) << 32, so the product is brev(a) * 2^-32 and one rounding to float remains.
GX_DEV float radical_inverse_2(uint32_t a) { return (float)((double)__brev(a) * 0x1p-32); }

// HaltonSampler::SampleDimension, HaltonSampler.cpp:85-94 (sampleAtPixelCenter == false)
GX_DEV float halton_sample(const DSamplerTables &t, uint32_t index, int dim) {
    if (dim >= 1000) dim = 2 + (dim - 2) % 998;  // reference reads PrimeSums out of bounds here; defined to wrap
    if (dim == 0) return radical_inverse_2(index >> t.h.base_exp[0]);
    uint32_t base = (uint32_t)t.primes[dim], M = t.prime_magic[2 * dim], s = t.prime_magic[2 * dim + 1];
    if (dim == 1) return radical_inverse_base(index / (uint32_t)t.h.base_scale[1], base, M, s);
    return scrambled_radical_inverse_base(index, base, M, s, t.perms + t.prime_sums[dim]);
}

// Two consecutive dimensions (both >= 2) of the same sample index at once: the two digit chains are independent, so their
// permutation-table loads overlap (one memory round trip instead of two).  Same operations per dimension as
// scrambled_radical_inverse_base, hence the same values.
GX_DEV void halton_sample_pair(const DSamplerTables &t, uint32_t index, int dim, float *u0, float *u1) {
    int d0 = dim, d1 = dim + 1;
    if (d0 >= 1000) d0 = 2 + (d0 - 2) % 998;
    if (d1 >= 1000) d1 = 2 + (d1 - 2) % 998;
    if (d0 < 2 || d1 < 2) { *u0 = halton_sample(t, index, dim); *u1 = halton_sample(t, index, dim + 1); return; }
    const uint32_t b0 = (uint32_t)t.primes[d0], M0 = t.prime_magic[2 * d0], s0 = t.prime_magic[2 * d0 + 1];
    const uint32_t b1 = (uint32_t)t.primes[d1], M1 = t.prime_magic[2 * d1], s1 = t.prime_magic[2 * d1 + 1];
    const uint16_t *__restrict__ perm0 = t.perms + t.prime_sums[d0], *__restrict__ perm1 = t.perms + t.prime_sums[d1];
    const float inv0 = 1.f / (float)b0, inv1 = 1.f / (float)b1;
    uint64_t rev0 = 0, rev1 = 0;
    float invN0 = 1, invN1 = 1;
    uint32_t a0 = index, a1 = index;
    const uint32_t z0 = perm0[0], z1 = perm1[0];
    while (a0 | a1) {
        const uint32_t n01 = div_magic(a0, M0, s0), n02 = div_magic(n01, M0, s0), n03 = div_magic(n02, M0, s0), n04 = div_magic(n03, M0, s0);
        const uint32_t n11 = div_magic(a1, M1, s1), n12 = div_magic(n11, M1, s1), n13 = div_magic(n12, M1, s1), n14 = div_magic(n13, M1, s1);
        const uint32_t p00 = perm0[a0 - n01 * b0], p01 = perm0[n01 - n02 * b0], p02 = perm0[n02 - n03 * b0], p03 = perm0[n03 - n04 * b0];
        const uint32_t p10 = perm1[a1 - n11 * b1], p11 = perm1[n11 - n12 * b1], p12 = perm1[n12 - n13 * b1], p13 = perm1[n13 - n14 * b1];
        if (a0) { rev0 = rev0 * b0 + p00; invN0 *= inv0; }
        if (n01) { rev0 = rev0 * b0 + p01; invN0 *= inv0; }
        if (n02) { rev0 = rev0 * b0 + p02; invN0 *= inv0; }
        if (n03) { rev0 = rev0 * b0 + p03; invN0 *= inv0; }
        if (a1) { rev1 = rev1 * b1 + p10; invN1 *= inv1; }
        if (n11) { rev1 = rev1 * b1 + p11; invN1 *= inv1; }
        if (n12) { rev1 = rev1 * b1 + p12; invN1 *= inv1; }
        if (n13) { rev1 = rev1 * b1 + p13; invN1 *= inv1; }
        a0 = n04; a1 = n14;
    }
    *u0 = fminf(invN0 * ((float)rev0 + inv0 * (float)(int)z0 / (1 - inv0)), GX_ONE_MINUS_EPS);
    *u1 = fminf(invN1 * ((float)rev1 + inv1 * (float)(int)z1 / (1 - inv1)), GX_ONE_MINUS_EPS);
}

// HaltonSampler::GetIndexForSample offset part, HaltonSampler.cpp:63-83 (kMaxResolution = 128)
GX_DEV uint32_t halton_pixel_offset(const DHalton &h, int px, int py) {
    if (h.stride <= 1) return 0;
    uint32_t pm0 = (uint32_t)(px & 127), pm1 = (uint32_t)(py & 127);  // Mod(p, 128) for p >= 0
    // InverseRadicalInverse<2>, <3>  (LowDiscrepancy.h:47-56)
    uint64_t i0 = 0, i1 = 0;
    for (int i = 0; i < h.base_exp[0]; ++i) { uint32_t digit = pm0 & 1; pm0 >>= 1; i0 = i0 * 2 + digit; }
    for (int i = 0; i < h.base_exp[1]; ++i) { uint32_t digit = pm1 % 3; pm1 /= 3; i1 = i1 * 3 + digit; }
    uint64_t offset = i0 * (uint64_t)(h.stride / h.base_scale[0]) * (uint64_t)h.mult_inv[0] +
                      i1 * (uint64_t)(h.stride / h.base_scale[1]) * (uint64_t)h.mult_inv[1];
    return (uint32_t)(offset % (uint64_t)h.stride);
}

// GlobalSampler::Get1D / Get2D view of one pixel sample, core/Sampler.cpp:162-179
struct SampleStream {
    const DSamplerTables &t;
    uint32_t index;
    int dim;
    GX_DEV SampleStream(const DSamplerTables &t, uint32_t index, int dim) : t(t), index(index), dim(dim) {}
    GX_DEV float get1d() { return halton_sample(t, index, dim++); }
    GX_DEV void get2d(float *u0, float *u1) { halton_sample_pair(t, index, dim, u0, u1); dim += 2; }
};

// core/Sampling.cpp:87-105
GX_DEV void concentric_sample_disk(float u0, float u1, float *dx, float *dy) {
    float ox = 2.f * u0 - 1, oy = 2.f * u1 - 1;
    if (ox == 0 && oy == 0) { *dx = 0; *dy = 0; return; }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) { r = ox; theta = GX_PI_OVER_4 * (oy / ox); }
    else { r = oy; theta = GX_PI_OVER_2 - GX_PI_OVER_4 * (ox / oy); }
    float st, ct;
    gx_sincos(theta, &st, &ct);
    *dx = r * ct;
    *dy = r * st;
}
// core/Sampling.h:140-145
GX_DEV V3 cosine_sample_hemisphere(float u0, float u1) {
    float dx, dy;
    concentric_sample_disk(u0, u1, &dx, &dy);
    float z = gx_sqrt(fmaxf(0.f, 1 - dx * dx - dy * dy));
    return V3(dx, dy, z);
}
// core/Sampling.h:157-161
GX_DEV float power_heuristic(float fPdf, float gPdf) {
    float f = 1 * fPdf, g = 1 * gPdf;
    return (f * f) / (f * f + g * g);
}

}  // namespace gnxr

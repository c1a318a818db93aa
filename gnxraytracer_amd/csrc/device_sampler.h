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
    const uint32_t *prime_magic;  // 8 words per prime: M, s (exact division), base, permutation offset, 1 / base, perm[0] tail, ceil(2^32 / base), 0
    DHalton h;
};

// exact n / d for any n < 2^32 (Granlund-Montgomery round-up method)
GX_DEV uint32_t div_magic(uint32_t n, uint32_t M, uint32_t s) {
    uint32_t t = __umulhi(M, n);
    return (t + ((n - t) >> 1)) >> (s - 1);
}

// What a scrambled radical inverse needs to know about its dimension.  invBase = 1 / (float)base and the tail of the reference's last line,
// invBase * perm[0] / (1 - invBase) (LowDiscrepancy.cpp:392: the digits beyond the index's own are all perm[0]), are constants of the
// dimension: the LDS copy holds them (computed once per block with the same IEEE operations), the global path computes them per call.
struct DimInfo { uint32_t base, M, s, off; float invBase, tail; uint32_t M32; };   // M32 = ceil(2^32 / base): n / base == umulhi(n, M32) while n * base < 2^32
GX_DEV float halton_tail(float invBase, uint32_t perm0) { return invBase * (float)(int)perm0 / (1 - invBase); }

// Where the tables are read from.  GlobalTab: the scene's arrays in global memory.  LdsTab: a block's LDS copy of the first `dims`
// dimensions (k_shade: the permutation digits of a path vertex are ~90 per-lane gathers, and the vector-memory path takes ~1 lane per
// clock per CU whatever they hit -- tools/probes/gather_probe.hip; from LDS they are ds_reads).  Same values, same arithmetic.
struct GlobalTab {
    const DSamplerTables &t;
    GX_DEV uint32_t perm(uint32_t i) const { return t.perms[i]; }
    GX_DEV DimInfo dim(int d) const {   // one 32-byte record per dimension, built by the scene compiler (build_sampler_tables)
        typedef unsigned int u4g __attribute__((ext_vector_type(4)));
        const u4g *rec = reinterpret_cast<const u4g *>(t.prime_magic) + 2 * (size_t)d;
        const u4g v = rec[0], w = rec[1];
        DimInfo q;
        q.M = v.x; q.s = v.y; q.base = v.z; q.off = v.w; q.invBase = __uint_as_float(w.x); q.tail = __uint_as_float(w.y); q.M32 = w.z;
        return q;
    }
};
typedef __attribute__((address_space(3))) const uint16_t lds_u16c;
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u4v lds_u4c;
struct LdsSampler {
    lds_u16c *perms;   // perms[0 .. prime_sums[dims])
    lds_u4c *tab;      // per dimension two entries: (prime, magic M, magic s, prime_sum), (invBase, tail, 0, 0)
    int dims;          // dimensions [0, dims) are in LDS; 0 == no LDS copy
};
struct LdsTab {
    const LdsSampler &l;
    GX_DEV uint32_t perm(uint32_t i) const { return l.perms[i]; }
    GX_DEV DimInfo dim(int d) const {
        const u4v v = l.tab[2 * d], w = l.tab[2 * d + 1];
        DimInfo q;
        q.base = v.x; q.M = v.y; q.s = v.z; q.off = v.w; q.invBase = __uint_as_float(w.x); q.tail = __uint_as_float(w.y); q.M32 = w.z;
        return q;
    }
};

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
// n / base for the digit loop.  With a 32-bit accumulator the index satisfies n * base < 2^32 (DHalton::base32_max), and then one
// multiply-high by ceil(2^32 / base) is exact: the error term n * e / 2^32 (e = M32 * base - 2^32 < base) stays below 1 / base.  (MI355X
// issues v_mul_hi_u32 and shifts at half the rate of adds: the general round-up form costs four times as much, tools/probes/valu_probe.hip.)
template <class ACC> GX_DEV uint32_t digit_div(uint32_t n, const DimInfo &q);
template <> GX_DEV uint32_t digit_div<uint32_t>(uint32_t n, const DimInfo &q) { return __umulhi(n, q.M32); }
template <> GX_DEV uint32_t digit_div<uint64_t>(uint32_t n, const DimInfo &q) { return div_magic(n, q.M, q.s); }
// ScrambledRadicalInverseSpecialized<base>, LowDiscrepancy.cpp:374-393.  Digits are peeled four at a time so that the four
// permutation-table loads are independent and in flight together (the serial loop waited for one L1/L2 round trip per
// digit); the accumulation then runs in the reference's order, digit by digit, for as many digits as the index has.
// ACC: the integer type reversedDigits is kept in.  The reference's is 64 bits wide; its value stays below base * index, so when that
// fits 32 bits (DHalton::base32_max, set per render from the largest index) a 32-bit accumulator holds the same integer, the digit
// step is one multiply-add instead of a two-part 64-bit one, and the final conversion to float rounds the same integer once.
template <class ACC, class TAB>
GX_DEV float scrambled_radical_inverse_acc(uint32_t a, const DimInfo &q, const TAB &tab) {
    const uint32_t base = q.base, off = q.off;
    ACC reversedDigits = 0;
    float invBaseN = 1;
    while (a) {
        const uint32_t n1 = digit_div<ACC>(a, q), n2 = digit_div<ACC>(n1, q), n3 = digit_div<ACC>(n2, q), n4 = digit_div<ACC>(n3, q);
        const uint32_t p0 = tab.perm(off + a - n1 * base), p1 = tab.perm(off + n1 - n2 * base), p2 = tab.perm(off + n2 - n3 * base), p3 = tab.perm(off + n3 - n4 * base);
        reversedDigits = reversedDigits * base + p0; invBaseN *= q.invBase;
        reversedDigits = n1 ? reversedDigits * base + p1 : reversedDigits; invBaseN = n1 ? invBaseN * q.invBase : invBaseN;
        reversedDigits = n2 ? reversedDigits * base + p2 : reversedDigits; invBaseN = n2 ? invBaseN * q.invBase : invBaseN;
        reversedDigits = n3 ? reversedDigits * base + p3 : reversedDigits; invBaseN = n3 ? invBaseN * q.invBase : invBaseN;
        a = n4;
    }
    return fminf(invBaseN * ((float)reversedDigits + q.tail), GX_ONE_MINUS_EPS);
}
// base 2: ReverseBits64(a) * 2^-64 evaluated in double, LowDiscrepancy.cpp:396-403.  For a < 2^32 the
// reversed word is brev(a) << 32, so the product is brev(a) * 2^-32 and one rounding to float remains.
GX_DEV float radical_inverse_2(uint32_t a) { return (float)((double)__brev(a) * 0x1p-32); }

// HaltonSampler::SampleDimension, HaltonSampler.cpp:85-94 (sampleAtPixelCenter == false); dim already wrapped below 1000
template <class TAB>
GX_DEV float halton_sample_t(const TAB &tab, const DHalton &h, uint32_t index, int dim) {
    if (dim == 0) return radical_inverse_2(index >> h.base_exp[0]);
    const DimInfo q = tab.dim(dim);
    if (dim == 1) return radical_inverse_base(index / (uint32_t)h.base_scale[1], q.base, q.M, q.s);
    if (q.base <= (uint32_t)h.base32_max) return scrambled_radical_inverse_acc<uint32_t>(index, q, tab);
    return scrambled_radical_inverse_acc<uint64_t>(index, q, tab);
}
GX_DEV float halton_sample(const DSamplerTables &t, uint32_t index, int dim) {
    if (dim >= 1000) dim = 2 + (dim - 2) % 998;  // reference reads PrimeSums out of bounds here; defined to wrap
    return halton_sample_t(GlobalTab{t}, t.h, index, dim);
}

// Two consecutive dimensions (both >= 2, already wrapped) of the same sample index at once: the two digit chains are independent, so
// their permutation-table loads overlap (one memory round trip instead of two).  Same operations per dimension as
// scrambled_radical_inverse_acc, hence the same values.
template <class ACC, class TAB>
GX_DEV void halton_sample_pair_acc(const TAB &tab, uint32_t index, const DimInfo &q0, const DimInfo &q1, float *u0, float *u1) {
    const uint32_t b0 = q0.base, o0 = q0.off, b1 = q1.base, o1 = q1.off;
    const float inv0 = q0.invBase, inv1 = q1.invBase;
    ACC rev0 = 0, rev1 = 0;
    float invN0 = 1, invN1 = 1;
    uint32_t a0 = index, a1 = index;
    while (a0 | a1) {
        const uint32_t n01 = digit_div<ACC>(a0, q0), n02 = digit_div<ACC>(n01, q0), n03 = digit_div<ACC>(n02, q0), n04 = digit_div<ACC>(n03, q0);
        const uint32_t n11 = digit_div<ACC>(a1, q1), n12 = digit_div<ACC>(n11, q1), n13 = digit_div<ACC>(n12, q1), n14 = digit_div<ACC>(n13, q1);
        const uint32_t p00 = tab.perm(o0 + a0 - n01 * b0), p01 = tab.perm(o0 + n01 - n02 * b0), p02 = tab.perm(o0 + n02 - n03 * b0), p03 = tab.perm(o0 + n03 - n04 * b0);
        const uint32_t p10 = tab.perm(o1 + a1 - n11 * b1), p11 = tab.perm(o1 + n11 - n12 * b1), p12 = tab.perm(o1 + n12 - n13 * b1), p13 = tab.perm(o1 + n13 - n14 * b1);
        rev0 = a0 ? rev0 * b0 + p00 : rev0; invN0 = a0 ? invN0 * inv0 : invN0;
        rev0 = n01 ? rev0 * b0 + p01 : rev0; invN0 = n01 ? invN0 * inv0 : invN0;
        rev0 = n02 ? rev0 * b0 + p02 : rev0; invN0 = n02 ? invN0 * inv0 : invN0;
        rev0 = n03 ? rev0 * b0 + p03 : rev0; invN0 = n03 ? invN0 * inv0 : invN0;
        rev1 = a1 ? rev1 * b1 + p10 : rev1; invN1 = a1 ? invN1 * inv1 : invN1;
        rev1 = n11 ? rev1 * b1 + p11 : rev1; invN1 = n11 ? invN1 * inv1 : invN1;
        rev1 = n12 ? rev1 * b1 + p12 : rev1; invN1 = n12 ? invN1 * inv1 : invN1;
        rev1 = n13 ? rev1 * b1 + p13 : rev1; invN1 = n13 ? invN1 * inv1 : invN1;
        a0 = n04; a1 = n14;
    }
    *u0 = fminf(invN0 * ((float)rev0 + q0.tail), GX_ONE_MINUS_EPS);
    *u1 = fminf(invN1 * ((float)rev1 + q1.tail), GX_ONE_MINUS_EPS);
}
template <class TAB>
GX_DEV void halton_sample_pair_t(const TAB &tab, const DHalton &h, uint32_t index, int d0, int d1, float *u0, float *u1) {
    const DimInfo q0 = tab.dim(d0), q1 = tab.dim(d1);
    if (q1.base <= (uint32_t)h.base32_max) halton_sample_pair_acc<uint32_t>(tab, index, q0, q1, u0, u1);   // primes ascend with the dimension
    else halton_sample_pair_acc<uint64_t>(tab, index, q0, q1, u0, u1);
}
GX_DEV void halton_sample_pair(const DSamplerTables &t, uint32_t index, int dim, float *u0, float *u1) {
    int d0 = dim, d1 = dim + 1;
    if (d0 >= 1000) d0 = 2 + (d0 - 2) % 998;
    if (d1 >= 1000) d1 = 2 + (d1 - 2) % 998;
    if (d0 < 2 || d1 < 2) { *u0 = halton_sample(t, index, dim); *u1 = halton_sample(t, index, dim + 1); return; }
    if (d1 < d0) { *u0 = halton_sample(t, index, d0); *u1 = halton_sample(t, index, d1); return; }   // the pair straddles the wrap
    halton_sample_pair_t(GlobalTab{t}, t.h, index, d0, d1, u0, u1);
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

// The radical inverses as real (non-inlined) functions: a path vertex draws seven to nine values from five call sites, and every inlined
// copy carries the LDS / global and 32- / 64-bit variants of the unrolled digit loop -- tens of KB of code per shade kernel, more than the
// instruction cache holds.  GX_HALTON_CALLS = 0 inlines them again.
#ifndef GX_HALTON_CALLS
#define GX_HALTON_CALLS 1
#endif
#if GX_HALTON_CALLS
#define GX_NOINLINE_DEV static __device__ __attribute__((noinline))
GX_NOINLINE_DEV float2 halton_pair_lds_call(lds_u16c *perms, lds_u4c *tab, int base32_max, uint32_t index, int d) {
    LdsSampler l; l.perms = perms; l.tab = tab; l.dims = 0;
    DHalton h; h.base32_max = base32_max;
    float u0, u1;
    halton_sample_pair_t(LdsTab{l}, h, index, d, d + 1, &u0, &u1);
    return make_float2(u0, u1);
}
GX_NOINLINE_DEV float halton_one_lds_call(lds_u16c *perms, lds_u4c *tab, int base_scale1, int base32_max, uint32_t index, int d) {
    LdsSampler l; l.perms = perms; l.tab = tab; l.dims = 0;
    DHalton h; h.base_scale[1] = base_scale1; h.base_exp[0] = 0; h.base32_max = base32_max;
    return halton_sample_t(LdsTab{l}, h, index, d);   // d >= 1
}
GX_NOINLINE_DEV float2 halton_pair_global_call(const uint16_t *perms, const int32_t *primes, const int32_t *prime_sums, const uint32_t *prime_magic, int base_exp0, int base_scale1,
                                               int base32_max, uint32_t index, int d) {
    DSamplerTables t; t.perms = perms; t.primes = primes; t.prime_sums = prime_sums; t.prime_magic = prime_magic;
    t.h.base_exp[0] = base_exp0; t.h.base_scale[1] = base_scale1; t.h.base32_max = base32_max;
    float u0, u1;
    halton_sample_pair(t, index, d, &u0, &u1);
    return make_float2(u0, u1);
}
GX_NOINLINE_DEV float halton_one_global_call(const uint16_t *perms, const int32_t *primes, const int32_t *prime_sums, const uint32_t *prime_magic, int base_exp0, int base_scale1,
                                             int base32_max, uint32_t index, int d) {
    DSamplerTables t; t.perms = perms; t.primes = primes; t.prime_sums = prime_sums; t.prime_magic = prime_magic;
    t.h.base_exp[0] = base_exp0; t.h.base_scale[1] = base_scale1; t.h.base32_max = base32_max;
    return halton_sample(t, index, d);
}
#endif

// GlobalSampler::Get1D / Get2D view of one pixel sample, core/Sampler.cpp:162-179
struct SampleStream {
    const DSamplerTables &t;
    uint32_t index;
    int dim;
    LdsSampler lds;   // optional LDS copy of the first dimensions' tables (dims == 0: none)
    GX_DEV SampleStream(const DSamplerTables &t, uint32_t index, int dim) : t(t), index(index), dim(dim) { lds.perms = nullptr; lds.tab = nullptr; lds.dims = 0; }
    GX_DEV SampleStream(const DSamplerTables &t, uint32_t index, int dim, const LdsSampler &l) : t(t), index(index), dim(dim), lds(l) {}
    GX_DEV float get1d() {
        const int d = dim++;
#if GX_HALTON_CALLS
        if (d >= 1 && d < lds.dims) return halton_one_lds_call(lds.perms, lds.tab, t.h.base_scale[1], t.h.base32_max, index, d);
        return halton_one_global_call(t.perms, t.primes, t.prime_sums, t.prime_magic, t.h.base_exp[0], t.h.base_scale[1], t.h.base32_max, index, d);
#else
        if (d >= 1 && d < lds.dims) return halton_sample_t(LdsTab{lds}, t.h, index, d);
        return halton_sample(t, index, d);
#endif
    }
    GX_DEV void get2d(float *u0, float *u1) {
        const int d = dim;
        dim += 2;
#if GX_HALTON_CALLS
        const float2 r = (d >= 2 && d + 1 < lds.dims) ? halton_pair_lds_call(lds.perms, lds.tab, t.h.base32_max, index, d)
                                                      : halton_pair_global_call(t.perms, t.primes, t.prime_sums, t.prime_magic, t.h.base_exp[0], t.h.base_scale[1], t.h.base32_max, index, d);
        *u0 = r.x; *u1 = r.y;
#else
        if (d >= 2 && d + 1 < lds.dims) { halton_sample_pair_t(LdsTab{lds}, t.h, index, d, d + 1, u0, u1); return; }
        halton_sample_pair(t, index, d, u0, u1);
#endif
    }
};
// a block's LDS copy of the first `dims` dimensions: perms[0 .. prime_sums[dims]) followed (16-byte aligned) by the per-dimension table
GX_DEV size_t lds_sampler_bytes(int n_perm, int dims) { return (((size_t)n_perm * 2 + 15) & ~(size_t)15) + (size_t)dims * 32; }
GX_DEV LdsSampler lds_sampler_fill(const DSamplerTables &t, int dims, int n_perm, int *smem, int tid, int nthreads) {
    LdsSampler l;
    l.perms = nullptr; l.tab = nullptr; l.dims = 0;
    if (dims <= 0) return l;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) u4v lds_u4;
    lds_u32 *pw = (lds_u32 *)smem;
    const uint32_t *gp = reinterpret_cast<const uint32_t *>(t.perms);   // perms is 4-byte aligned (hipMalloc)
    const int nw = (n_perm + 1) / 2;
    for (int i = tid; i < nw; i += nthreads) pw[i] = gp[i];
    lds_u4 *tb = (lds_u4 *)((__attribute__((address_space(3))) char *)smem + (((size_t)n_perm * 2 + 15) & ~(size_t)15));
    for (int d = tid; d < dims; d += nthreads) {
        const DimInfo q = GlobalTab{t}.dim(d);
        u4v v, w;
        v.x = q.base; v.y = q.M; v.z = q.s; v.w = q.off;
        w.x = __float_as_uint(q.invBase); w.y = __float_as_uint(q.tail); w.z = q.M32; w.w = 0;
        tb[2 * d] = v; tb[2 * d + 1] = w;
    }
    l.perms = (lds_u16c *)smem; l.tab = (lds_u4c *)tb; l.dims = dims;
    return l;
}

// core/Sampling.cpp:87-105
GX_DEV void concentric_sample_disk(float u0, float u1, float *dx, float *dy) {
    float ox = 2.f * u0 - 1, oy = 2.f * u1 - 1;
    if (ox == 0 && oy == 0) { *dx = 0; *dy = 0; return; }
    // (one division for both branches of Sampling.cpp:95-101: the lanes of a wave split evenly between them)
    const bool xBig = fabsf(ox) > fabsf(oy);
    const float q = (xBig ? oy : ox) / (xBig ? ox : oy);
    const float r = xBig ? ox : oy;
    const float theta = xBig ? GX_PI_OVER_4 * q : GX_PI_OVER_2 - GX_PI_OVER_4 * q;
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

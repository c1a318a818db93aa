// device_math.h -- CDNA4 device-side value types for the wavefront path tracer.
//
// Arithmetic follows the reference's inline math (core/Geometry.h, core/GNXRayTracer.h) operation for
// operation: the library is compiled with -ffp-contract=off so hipcc does not fuse a*b+c (the x86-64
// reference has no FMA), division and sqrt are IEEE (hipcc default), and the float libm calls the
// reference makes (logf, expf, sinf, cosf) are glibc's own algorithms restated (gx_log etc. below).
// Together this keeps GPU paths on the same discrete decisions (lobe choice, hit/miss, Russian
// roulette, delta-tracking collisions) as the CPU reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gnxr {

#define GX_DEV __device__ __forceinline__

static constexpr float GX_INF = __builtin_huge_valf();
static constexpr float GX_PI = 3.14159265358979323846f;
static constexpr float GX_INV_PI = 0.31830988618379067154f;
static constexpr float GX_INV_2PI = 0.15915494309189533577f;
static constexpr float GX_PI_OVER_2 = 1.57079632679489661923f;
static constexpr float GX_PI_OVER_4 = 0.78539816339744830961f;
static constexpr float GX_ONE_MINUS_EPS = 0x1.fffffep-1f;
static constexpr float GX_MACH_EPS = 0x1p-24f;       // std::numeric_limits<float>::epsilon() * 0.5
static constexpr float GX_SHADOW_EPS = 0.0001f;
// gamma(n) = (n * MachineEpsilon) / (1 - n * MachineEpsilon), GNXRayTracer.h:354-357 (constant-folded in fp32)
#define GX_GAMMA(n) (((n) * GX_MACH_EPS) / (1 - (n) * GX_MACH_EPS))

// ---- float libm: glibc 2.35's own algorithms (sysdeps/ieee754/flt-32/{e_logf,e_expf,s_sinf,s_cosf}.c, the ARM
// optimized-routines implementations: table + short polynomial evaluated in double, one rounding to float at the end).
// The reference calls std::log / std::exp / std::sin / std::cos on floats, i.e. exactly these functions, and they are
// NOT correctly rounded (0.5-0.9 ulp): rounding a double-precision result to float differs from them for 0.7 % (logf)
// to 1.3 % (sinf, cosf) of the arguments.  Restating the algorithms operation for operation (same tables, same
// evaluation order) reproduces glibc bit for bit -- checked against libm.so.6 on 4e5 random arguments per function --
// and costs a dozen fp64 operations instead of OCML's full double-precision routines.
// On x86-64 hosts with FMA (every host this runs next to) glibc dispatches to its -mfma builds of these functions, in
// which GCC contracts a * b + c wherever every use of the product is an addition or subtraction; the explicit fma()
// calls below are exactly those contractions (the plain products are the ones GCC leaves alone), e.g. in expf both
// kd = z + Shift and r = z - kd take the unrounded product z = InvLn2N * x.
// powf is restated further below; tan is used by the host-side camera set-up only.
__device__ static const double gx_logf_invc[16] = {
    0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0, 0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,
    0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0, 0x1p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
    0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
__device__ static const double gx_logf_logc[16] = {
    -0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3, -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,
    -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4, -0x1.252f438e10c1ep-5, 0x0p+0, 0x1.aa5aa5df25984p-5, 0x1.c5e53aa362eb4p-4,
    0x1.526e57720db08p-3, 0x1.bc2860d22477p-3, 0x1.1058bc8a07ee1p-2, 0x1.4043057b6ee09p-2};
// exp2f_data.tab: asuint64(2^(i/32)) - (i << 47)
__device__ static const unsigned long long gx_expf_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

GX_DEV float gx_log(float x) {   // e_logf.c
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -__builtin_huge_valf();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");
        ix = __float_as_uint(x * 0x1p23f);   // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = gx_logf_invc[i], logc = gx_logf_logc[i];
    const double z = (double)__uint_as_float(iz);
    const double r = fma(z, invc, -1.0);
    const double y0 = fma((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = fma(-0x1.00ea348b88334p-2, r2, y);
    y = fma(y, r2, y0 + r);
    return (float)y;
}
GX_DEV float gx_exp(float x) {   // e_expf.c (N = 32)
    const uint32_t abstop = (__float_as_uint(x) >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {   // |x| >= 88 or NaN
        if (__float_as_uint(x) == 0xff800000u) return 0.f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_huge_valf();
        if (x < -0x1.9fe368p6f) return 0.f;
    }
    const double xd = (double)x;
    double kd = fma(0x1.71547652b82fep+0 * 32, xd, 0x1.8p+52);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= 0x1.8p+52;
    const double r = fma(0x1.71547652b82fep+0 * 32, xd, -kd);
    const unsigned long long t = gx_expf_tab[ki & 31u] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    const double zz = fma(0x1.c6af84b912394p-5 / 32 / 32 / 32, r, 0x1.ebfce50fac4f3p-3 / 32 / 32);
    const double r2 = r * r;
    double y = fma(0x1.62e42ff0c52d6p-1 / 32, r, 1.0);
    y = fma(zz, r2, y);
    y = y * s;
    return (float)y;
}
// s_sincosf.h: sinf_poly with the coefficient set of __sincosf_table[neg] (neg: cosine coefficients negated)
GX_DEV double gx_sinf_poly(double x, double x2, bool neg, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
        double x7 = x3 * x2;
        double s = fma(x3, -0x1.555545995a603p-3, x);
        return fma(x7, s1, s);
    }
    const double sg = neg ? -1.0 : 1.0;   // exact sign flips of the table constants
    double x4 = x2 * x2;
    double c2 = fma(x2, sg * 0x1.99343027bf8c3p-16, sg * -0x1.6c087e89a359dp-10);
    double c1 = fma(x2, sg * -0x1.ffffffd0c621cp-2, sg * 0x1p0);
    double x6 = x4 * x2;
    double c = fma(x4, sg * 0x1.55553e1068f19p-5, c1);
    return fma(x6, c2, c);
}
GX_DEV double gx_reduce_fast(double x, int *np) {   // |x| < 120
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ((int)r + 0x800000) >> 24;
    *np = n;
    return fma(-(double)n, 0x1.921FB54442D18p0, x);
}
GX_DEV float gx_sin(float y) {   // s_sinf.c
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
    double x = (double)y;
    if (top < 0x3f4u) {            // |y| < pi/4 (abstop12 compare)
        if (top < 0x398u) return y;   // |y| < 2^-12
        return (float)gx_sinf_poly(x, x * x, false, 0);
    }
    if (top < 0x42fu) {            // |y| < 120
        int n;
        x = gx_reduce_fast(x, &n);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;   // sign[n & 3] = {1, -1, -1, 1}
        return (float)gx_sinf_poly(x * s, x * x, (n & 2) != 0, n);
    }
    return (float)sin((double)y);   // large arguments do not occur on this path
}
GX_DEV float gx_cos(float y) {   // s_cosf.c
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
    double x = (double)y;
    if (top < 0x3f4u) {
        if (top < 0x398u) return 1.0f;
        return (float)gx_sinf_poly(x, x * x, false, 1);
    }
    if (top < 0x42fu) {
        int n;
        x = gx_reduce_fast(x, &n);
        const int m = n + 1;
        const double s = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;
        return (float)gx_sinf_poly(x * s, x * x, (m & 2) != 0, n ^ 1);
    }
    return (float)cos((double)y);
}
// sinf(y) and cosf(y) of the same argument: the two calls share the argument reduction, and between them evaluate the
// sine polynomial once and the cosine polynomial once (which of the two results gets which depends on the quadrant).
GX_DEV void gx_sincos(float y, float *sOut, float *cOut) {
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
    double x = (double)y;
    if (top < 0x3f4u) {
        if (top < 0x398u) { *sOut = y; *cOut = 1.0f; return; }
        const double x2 = x * x;
        *sOut = (float)gx_sinf_poly(x, x2, false, 0);
        *cOut = (float)gx_sinf_poly(x, x2, false, 1);
        return;
    }
    if (top < 0x42fu) {
        int n;
        x = gx_reduce_fast(x, &n);
        const int m = n + 1;
        const double ss = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;   // sign[n & 3]
        const double cs = ((m & 3) == 1 || (m & 3) == 2) ? -1.0 : 1.0;   // sign[(n + 1) & 3]
        const double x2 = x * x;
        const bool odd = (n & 1) != 0;
        // the even-index branch of sinf_poly takes the signed argument, the odd-index branch only x2 and the table
        const double pe = gx_sinf_poly(x * (odd ? cs : ss), x2, false, 0);
        const double po = gx_sinf_poly(0.0, x2, odd ? (n & 2) != 0 : (m & 2) != 0, 1);
        *sOut = (float)(odd ? po : pe);
        *cOut = (float)(odd ? pe : po);
        return;
    }
    *sOut = (float)sin((double)y);
    *cOut = (float)cos((double)y);
}
GX_DEV float gx_tan(float x) { return (float)tan((double)x); }
// acosf / atanf / atan2f: glibc 2.35 still ships the fdlibm float versions (sysdeps/ieee754/flt-32/{e_acosf,s_atanf,e_atan2f}.c),
// plain float arithmetic, no FMA build.  Constants are the decimal literals of those files (for aT[0] the literal,
// 3.3333334327e-01 = 0x3eaaaaab, not the 0x3eaaaaaa of its comment).  Checked against libm.so.6 like the functions above.
GX_DEV float gx_acos(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f,
                pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const int32_t hx = __float_as_int(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {   // |x| < 0.5
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        float z = x * x;
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx < 0) {     // x < -0.5
        float z = (one + x) * 0.5f;
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float s = __builtin_sqrtf(z);
        float r = p / q;
        float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {                 // x > 0.5
        float z = (one - x) * 0.5f;
        float s = __builtin_sqrtf(z);
        float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
        float c = (z - df * df) / (s + df);
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float r = p / q;
        float w = r * s + c;
        return 2.0f * (df + w);
    }
}
GX_DEV float gx_atan(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f, aT4 = 9.0908870101e-02f,
                aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f, aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f,
                aT10 = 1.6285819933e-02f;
    const float one = 1.0f;
    const int32_t hx = __float_as_int(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {   // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {    // |x| < 0.4375
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float hi = id == 0 ? atanhi[0] : (id == 1 ? atanhi[1] : (id == 2 ? atanhi[2] : atanhi[3]));
    const float lo = id == 0 ? atanlo[0] : (id == 1 ? atanlo[1] : (id == 2 ? atanlo[2] : atanlo[3]));
    z = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -z : z;
}
GX_DEV float gx_atan2(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f, pi_o_4 = 7.8539818525e-01f;
    const int32_t hx = __float_as_int(x), ix = hx & 0x7fffffff, hy = __float_as_int(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return gx_atan(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = gx_atan(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return -z;
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}
// powf: glibc 2.35 e_powf.c (ARM optimized routines; POWF_SCALE_BITS = 0 because TOINT_INTRINSICS is off on x86-64):
// log2(x) from a 16-entry table + degree-5 polynomial, y * log2(x), then 2^z with expf's 32-entry table.  Arguments on this
// path are x = alpha^2 in (0, 1], y = 1 - u in (0, 1] (DisneyClearcoat::Sample_f, DisneyMaterial.cpp:262); everything
// else (x <= 0, subnormal, inf, nan) goes through double OCML.
__device__ static const double gx_powf_logc[16] = {
    -0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2, -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7afp-3,
    -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4, -0x1.a6f9db6475fcep-5, 0x0p+0, 0x1.338ca9f24f53dp-4, 0x1.476a9543891bap-3,
    0x1.e840b4ac4e4d2p-3, 0x1.40645f0c6651cp-2, 0x1.88e9c2c1b9ff8p-2, 0x1.ce0a44eb17bccp-2};
GX_DEV float gx_pow(float x, float y) {
    const uint32_t ix = __float_as_uint(x), iy = __float_as_uint(y);
    const bool specialX = ix - 0x00800000u >= 0x7f800000u - 0x00800000u;   // x < 0x1p-126, inf or nan
    const bool specialY = 2 * iy - 1 >= 2u * 0x7f800000u - 1;               // y is 0, inf or nan
    if (specialX || specialY) return (float)pow((double)x, (double)y);
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int)top >> 23;
    const double invc = gx_logf_invc[i], logc = gx_powf_logc[i];   // the same invc as logf's table
    const double z = (double)__uint_as_float(iz);
    const double r = fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r4 = r2 * r2;
    double q = fma(0x1.71547652ab82bp0, r, y0);
    q = fma(p, r2, q);
    yy = fma(yy, r4, q);
    const double ylogx = (double)y * yy;
    if (((unsigned long long)__double_as_longlong(ylogx) >> 47 & 0xffff) >= ((unsigned long long)__double_as_longlong(126.0) >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6) return __builtin_huge_valf();   // |y * log2(x)| >= 126
        if (ylogx <= -150.0) return 0.f;
    }
    // exp2_inline (sign_bias = 0: x > 0)
    double kd = ylogx + 0x1.8p+52 / 32;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= 0x1.8p+52 / 32;
    const double rr = ylogx - kd;
    const unsigned long long t = gx_expf_tab[ki & 31u] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    const double zz = fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double out = fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    out = fma(zz, rr2, out);
    out = out * s;
    return (float)out;
}
// __fsqrt_rn maps to the *native* (not correctly rounded) sqrt in this ROCm; the builtin is IEEE under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.
GX_DEV float gx_sqrt(float x) { return __builtin_sqrtf(x); }

struct V3 {
    float x, y, z;
    GX_DEV V3() : x(0), y(0), z(0) {}
    GX_DEV V3(float x, float y, float z) : x(x), y(y), z(z) {}
    GX_DEV float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
GX_DEV V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
GX_DEV V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
GX_DEV V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
GX_DEV V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
GX_DEV V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
GX_DEV V3 operator/(V3 a, float f) { float inv = 1.f / f; return V3(a.x * inv, a.y * inv, a.z * inv); }  // Geometry.h:206-210
GX_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GX_DEV float absdot(V3 a, V3 b) { return fabsf(dot(a, b)); }
GX_DEV float length_sq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
GX_DEV float length(V3 a) { return gx_sqrt(length_sq(a)); }
GX_DEV V3 normalize(V3 a) { return a / length(a); }
GX_DEV V3 vabs(V3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
GX_DEV V3 cross(V3 a, V3 b) {  // double products, Geometry.h:925-931
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3((float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx)));
}
GX_DEV float max_component(V3 v) { return fmaxf(v.x, fmaxf(v.y, v.z)); }
GX_DEV V3 faceforward(V3 n, V3 v) { return (dot(n, v) < 0.f) ? -n : n; }
GX_DEV bool is_zero(V3 v) { return v.x == 0 && v.y == 0 && v.z == 0; }
GX_DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
GX_DEV float lerpf(float t, float a, float b) { return (1 - t) * a + t * b; }
GX_DEV void coordinate_system(V3 v1, V3 *v2, V3 *v3) {  // Geometry.h:988-995
    if (fabsf(v1.x) > fabsf(v1.y)) *v2 = V3(-v1.z, 0, v1.x) / gx_sqrt(v1.x * v1.x + v1.z * v1.z);
    else *v2 = V3(0, v1.z, -v1.y) / gx_sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3 = cross(v1, *v2);
}

// RGBSpectrum (core/Spectrum.h)
struct Spec {
    float r, g, b;
    GX_DEV Spec() : r(0), g(0), b(0) {}
    GX_DEV explicit Spec(float v) : r(v), g(v), b(v) {}
    GX_DEV Spec(float r, float g, float b) : r(r), g(g), b(b) {}
    GX_DEV bool is_black() const { return r == 0.f && g == 0.f && b == 0.f; }
    GX_DEV float max_value() const { return fmaxf(r, fmaxf(g, b)); }
    GX_DEV float y() const { return 0.212671f * r + 0.715160f * g + 0.072169f * b; }  // Spectrum.h:429-432
};
GX_DEV Spec operator+(Spec a, Spec b) { return Spec(a.r + b.r, a.g + b.g, a.b + b.b); }
GX_DEV Spec operator-(Spec a, Spec b) { return Spec(a.r - b.r, a.g - b.g, a.b - b.b); }
GX_DEV Spec operator*(Spec a, Spec b) { return Spec(a.r * b.r, a.g * b.g, a.b * b.b); }
GX_DEV Spec operator/(Spec a, Spec b) { return Spec(a.r / b.r, a.g / b.g, a.b / b.b); }
GX_DEV Spec operator*(Spec a, float s) { return Spec(a.r * s, a.g * s, a.b * s); }
GX_DEV Spec operator*(float s, Spec a) { return Spec(a.r * s, a.g * s, a.b * s); }
GX_DEV Spec operator/(Spec a, float s) { return Spec(a.r / s, a.g / s, a.b / s); }  // true division, Spectrum.h:146-152
GX_DEV Spec ssqrt(Spec a) { return Spec(gx_sqrt(a.r), gx_sqrt(a.g), gx_sqrt(a.b)); }
GX_DEV Spec slerp(float t, Spec a, Spec b) { return (1 - t) * a + t * b; }
GX_DEV Spec spec3(const float *p) { return Spec(p[0], p[1], p[2]); }

// core/GNXRayTracer.h:179-205
GX_DEV float next_float_up(float v) {
    if (isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = __float_as_uint(v);
    if (v >= 0) ++ui; else --ui;
    return __uint_as_float(ui);
}
GX_DEV float next_float_down(float v) {
    if (isinf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = __float_as_uint(v);
    if (v > 0) --ui; else ++ui;
    return __uint_as_float(ui);
}
// core/Geometry.h:1408-1422
GX_DEV V3 offset_ray_origin(V3 p, V3 pError, V3 n, V3 w) {
    float d = dot(vabs(n), pError);
    V3 offset = d * n;
    if (dot(w, n) < 0) offset = -offset;
    V3 po = p + offset;
    if (offset.x > 0) po.x = next_float_up(po.x); else if (offset.x < 0) po.x = next_float_down(po.x);
    if (offset.y > 0) po.y = next_float_up(po.y); else if (offset.y < 0) po.y = next_float_down(po.y);
    if (offset.z > 0) po.z = next_float_up(po.z); else if (offset.z < 0) po.z = next_float_down(po.z);
    return po;
}

// Transform::operator()(Point3) / (Vector3), Transform.h:196-218, on a row-major float[16]
GX_DEV V3 xform_point(const float *m, V3 p) {
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1) return V3(xp, yp, zp);
    float inv = 1.f / wp;
    return V3(inv * xp, inv * yp, inv * zp);
}
GX_DEV V3 xform_vector(const float *m, V3 v) {
    return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

}  // namespace gnxr

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
// ---- double-precision sin / cos: glibc 2.35 sysdeps/ieee754/dbl-64/s_sin.c (__sin / __cos; IBM Accurate Mathematical Library with the
// 2018 clean-up: table of sin / cos at k / 128 + short polynomials, 0.55 ULP, NOT correctly rounded), restated for the argument range the
// path produces (|x| < 105414350) with the FMA contractions GCC makes in the `_fma` multiarch variant every x86-64 host here selects
// through ifunc (read off the disassembly of libm.a's s_sin-fma.o).  The reference reaches them at core/MicroFacet.cpp:220-223, where
// the unqualified `cos(phi)` / `sin(phi)` bind to the double functions; OCML's results differ from glibc's in ~3 % of the arguments.
// `gx_sincostab` is glibc's __sincostab (sincostab.c): sin(k/128), its low part, cos(k/128), its low part, k = 0 .. 109.
__device__ static const double gx_sincostab[440] = {
    0x0.0p+0, 0x0.0p+0, 0x1.0000000000000p+0, 0x0.0p+0,
    0x1.fffeaaaaeeeefp-8, -0x1.e45e2ec67b77cp-62, 0x1.fffc000155552p-1, 0x1.f4a01a0196daep-55,
    0x1.fffaaaaeeeed5p-7, -0x1.2ab639a9f0777p-63, 0x1.fff000155549fp-1, 0x1.28a28a03a5ef3p-55,
    0x1.7ff7001033255p-6, 0x1.efe2b51527336p-64, 0x1.ffdc006bff7e6p-1, 0x1.ae6dae86977bdp-55,
    0x1.ffeaaaeeee86fp-6, -0x1.cd406fb224ae2p-60, 0x1.ffc00155527d3p-1, -0x1.3b54492d89b5bp-55,
    0x1.3feb2b12d45d5p-5, 0x1.4ec54203d1c11p-60, 0x1.ff9c03414a7bap-1, 0x1.991f4be6c59bfp-57,
    0x1.7fdc01032fba9p-5, -0x1.599bdf46e997ap-59, 0x1.ff7006bfdf99fp-1, -0x1.8b3b560648d5fp-56,
    0x1.bfc6d78586dacp-5, 0x1.8e4fd03dbf236p-62, 0x1.ff3c0c8103a31p-1, 0x1.4856dbddc0e66p-56,
    0x1.ffaaaeeed4edbp-5, -0x1.2d16d32684b69p-59, 0x1.ff0015549f4d3p-1, 0x1.328387b99426fp-55,
    0x1.1fc343d808befp-4, -0x1.f3d32e6f3be4fp-58, 0x1.febc222a8ef9fp-1, 0x1.7934934f54c77p-58,
    0x1.3facb12d1755bp-4, -0x1.921915299468cp-58, 0x1.fe7034129ef6fp-1, -0x1.cbf4337c96f97p-57,
    0x1.5f911fd10b737p-4, -0x1.0184f02be9102p-58, 0x1.fe1c4c3c873ebp-1, -0x1.5a9c9057c4a02p-60,
    0x1.7f701032550e4p-4, 0x1.afc2d1800501ap-60, 0x1.fdc06bf7e6b9bp-1, 0x1.31902b535f8dbp-55,
    0x1.9f4902d55d1f9p-4, 0x1.2696d7eac1dc1p-58, 0x1.fd5c94b43e000p-1, -0x1.2e768cb4f92f9p-57,
    0x1.bf1b78568391dp-4, 0x1.e91841dea4cc8p-58, 0x1.fcf0c800e99b1p-1, 0x1.ea3d786d186acp-57,
    0x1.dee6f16c1cce6p-4, -0x1.50f8e2fb71673p-59, 0x1.fc7d078d1bc88p-1, 0x1.075d2447db685p-55,
    0x1.feaaeee86ee36p-4, -0x1.afcb2bcc6f03bp-59, 0x1.fc015527d5bd3p-1, 0x1.b68f35094efb8p-55,
    0x1.0f3378ddd71d1p-3, 0x1.d8468724f0f9ep-57, 0x1.fb7db2bfe0695p-1, 0x1.21dadf4f65ab1p-55,
    0x1.1f0d3d7afceafp-3, -0x1.6ef95099769a5p-57, 0x1.faf22263c4bd3p-1, -0x1.52ace133a2769p-58,
    0x1.2ee285e4ab88fp-3, -0x1.e4d0f05dee058p-57, 0x1.fa5ea641c36f2p-1, 0x1.04da6ed17cc7cp-59,
    0x1.3eb312c5d66cbp-3, 0x1.47d666b66cb91p-57, 0x1.f9c340a7cc428p-1, 0x1.c5b6b063b7462p-55,
    0x1.4e7ea4dc5f27bp-3, 0x1.949db2ac072fcp-58, 0x1.f91ff40374d01p-1, -0x1.7d03f4d3a9e4cp-57,
    0x1.5e44fcfa126f3p-3, -0x1.6f443063f89b6p-57, 0x1.f874c2e1eecf6p-1, -0x1.c6514e1332b16p-55,
    0x1.6e05dc05a4d4cp-3, -0x1.32c5c8b81c940p-66, 0x1.f7c1afeffde24p-1, -0x1.8f55bc47540b1p-56,
    0x1.7dc102fbaf2b5p-3, 0x1.5ab50e23c97c3p-59, 0x1.f706bdf9ece1cp-1, -0x1.698c80c36dcb4p-55,
    0x1.8d7632efaa944p-3, -0x1.20fa262cbb953p-57, 0x1.f643efeb82acdp-1, 0x1.6b00ac1fe28acp-56,
    0x1.9d252d0cec312p-3, 0x1.9c43d80b1137dp-58, 0x1.f57948cff6797p-1, 0x1.e3a0d3e03b1d5p-57,
    0x1.accdb297a0765p-3, -0x1.9883b57d6cdebp-58, 0x1.f4a6cbd1e3a79p-1, 0x1.13df0edaebb57p-55,
    0x1.bc6f84edc6199p-3, 0x1.9c1a56a7b0cabp-57, 0x1.f3cc7c3b3d16ep-1, -0x1.21a3ad28a3494p-57,
    0x1.cc0a6588289a3p-3, -0x1.868d09bc87c6bp-57, 0x1.f2ea5d753ffedp-1, 0x1.cc4215f56d583p-55,
    0x1.db9e15fb5a5d0p-3, -0x1.32e20d6cc6fc2p-57, 0x1.f20073086649fp-1, 0x1.b940416c1984bp-56,
    0x1.eb2a57f8ae5a3p-3, -0x1.0be06af572cebp-57, 0x1.f10ec09c5873bp-1, 0x1.d9072762c1283p-55,
    0x1.faaeed4f31577p-3, -0x1.15d88508e32b8p-57, 0x1.f01549f7deea1p-1, 0x1.d3c1e99e5cafdp-55,
    0x1.0515cbf65155cp-2, -0x1.9b8c29dfd8ec8p-56, 0x1.ef141300d2f26p-1, -0x1.2aa1b08ded372p-55,
    0x1.0cd00cef36436p-2, -0x1.9fb0a0c93e2b5p-56, 0x1.ee0b1fbc0f11cp-1, -0x1.bfd2380bbc3b1p-59,
    0x1.14861aa94ddebp-2, -0x1.be881b5b615a4p-57, 0x1.ecfa744d5efa1p-1, -0x1.56d0a4af541d0p-58,
    0x1.1c37d64c6b876p-2, 0x1.46076fe0dcff5p-56, 0x1.ebe214f76efa8p-1, -0x1.02f9f12ba543ep-55,
    0x1.23e52111aaf36p-2, -0x1.4f080334eff18p-56, 0x1.eac2061bbaf4fp-1, 0x1.2c1d53e94658dp-57,
    0x1.2b8ddc43eb49fp-2, 0x1.1553899f2d807p-57, 0x1.e99a4c3a7cd83p-1, -0x1.2264b1bc53ce8p-55,
    0x1.3331e94049f87p-2, 0x1.e0cb6b40c302cp-56, 0x1.e86aebf29a9edp-1, 0x1.9397afdbb58a7p-55,
    0x1.3ad129769d3d8p-2, 0x1.03d5504878398p-63, 0x1.e733ea0193d40p-1, -0x1.6428b3546ce13p-55,
    0x1.426b7e69ee697p-2, -0x1.f09c75705c59fp-56, 0x1.e5f54b436e9d0p-1, 0x1.7eb0fd02fc8bcp-55,
    0x1.4a00c9b0f3d20p-2, 0x1.823ba6bb08eadp-56, 0x1.e4af14b2a449cp-1, -0x1.68ca02e8a6833p-55,
    0x1.5190ecf68a77ap-2, 0x1.b357155eef0f3p-56, 0x1.e3614b680d6a5p-1, -0x1.27793aa015237p-56,
    0x1.591bc9fa2f597p-2, 0x1.7c74bac3fe0cbp-57, 0x1.e20bf49acd6c1p-1, -0x1.660aec7ef636cp-58,
    0x1.60a1429078775p-2, 0x1.b1fd80ba89133p-58, 0x1.e0af15a03dbcep-1, 0x1.fe8e702771ae6p-58,
    0x1.682138a38d7f7p-2, -0x1.d889202444aadp-56, 0x1.df4ab3ebd875ep-1, -0x1.e2d8a7e6736c4p-55,
    0x1.6f9b8e33a0255p-2, 0x1.42bc14ee9da0dp-56, 0x1.ddded50f228d6p-1, -0x1.e80c8d42ba2bfp-57,
    0x1.7710255764214p-2, -0x1.6ead7314bb6cep-57, 0x1.dc6b7eb995912p-1, 0x1.4b364776dcd35p-58,
    0x1.7e7ee03c86d4ep-2, -0x1.b63bcdabf5af2p-56, 0x1.daf0b6b888e83p-1, 0x1.a249e2b5e5ceap-55,
    0x1.85e7a12826949p-2, 0x1.8a40e9b5face0p-56, 0x1.d96e82f71a9dcp-1, 0x1.ff61bd5d2039dp-55,
    0x1.8d4a4a774992fp-2, 0x1.44a02ea766326p-56, 0x1.d7e4e97e17b4ap-1, -0x1.3b770352bed94p-57,
    0x1.94a6be9f546c5p-2, -0x1.69ce13e683f58p-56, 0x1.d653f073e4040p-1, -0x1.76236434bec37p-55,
    0x1.9bfce02e80510p-2, 0x1.09e39a320b0a4p-56, 0x1.d4bb9e1c619e0p-1, 0x1.f34bb77858f61p-55,
    0x1.a34c91cc50ccap-2, -0x1.a310e3b50cecdp-58, 0x1.d31bf8d8d7c06p-1, 0x1.e60dd3089cbddp-56,
    0x1.aa95b63a09277p-2, -0x1.6293eb13c0381p-57, 0x1.d1750727d94f0p-1, 0x1.0d52b1ec1a48ep-55,
    0x1.b1d8305321617p-2, -0x1.ae242cb99f519p-56, 0x1.cfc6cfa52ad9fp-1, 0x1.8b5b5508f2a0dp-55,
    0x1.b913e30dbac43p-2, -0x1.e38ad2f6c3ff1p-56, 0x1.ce115909a82e5p-1, 0x1.1f139bb31109ap-55,
    0x1.c048b17b140a3p-2, 0x1.19fe6757e9fa7p-57, 0x1.cc54aa2b2972ep-1, 0x1.4ee162ba83a98p-57,
    0x1.c7767ec7fd19ep-2, -0x1.eb14d1a3d5826p-58, 0x1.ca90c9fc67d0bp-1, -0x1.46a81485e3462p-57,
    0x1.ce9d2e3d4a51fp-2, -0x1.2fc8a12dae298p-57, 0x1.c8c5bf8ce1a84p-1, 0x1.ab3d1a1590123p-56,
    0x1.d5bca34047661p-2, 0x1.28a44a75fc29cp-56, 0x1.c6f39208be53bp-1, -0x1.741dbfbaadb42p-55,
    0x1.dcd4c15329c9ap-2, 0x1.0d4c6e171fd9ap-56, 0x1.c51a48b8b175ep-1, -0x1.1bbb43b9aa880p-57,
    0x1.e3e56c1582a69p-2, -0x1.0a4821099f88fp-58, 0x1.c339eb01ddd81p-1, -0x1.caaf5ee82c5c0p-55,
    0x1.eaee8744b05f0p-2, -0x1.789b43c9b027dp-58, 0x1.c1528065b7d50p-1, -0x1.892111312e828p-55,
    0x1.f1eff6bc4f97bp-2, 0x1.17212f8a7525cp-56, 0x1.bf641081e7536p-1, 0x1.b7bd71628a9a1p-55,
    0x1.f8e99e76abc97p-2, 0x1.9d950af2d00a3p-58, 0x1.bd6ea310294f5p-1, 0x1.31bbcc88c109dp-56,
    0x1.ffdb628d2f57ap-2, 0x1.f4a992e905b6ap-57, 0x1.bb723fe630f32p-1, 0x1.72bd2452d0a39p-56,
    0x1.0362939c69955p-1, -0x1.2d8cd78397b01p-55, 0x1.b96eeef58840ep-1, 0x1.45a3cc78fade0p-58,
    0x1.06d3686946e5bp-1, 0x1.3f5ae4538ff1bp-55, 0x1.b764b84b704c2p-1, -0x1.f5848c21b389bp-55,
    0x1.0a4021e9e1001p-1, -0x1.6f643a13914f6p-55, 0x1.b553a410c104ep-1, 0x1.8ff7947027a16p-58,
    0x1.0da8b26b5672ep-1, -0x1.a58def0bee909p-55, 0x1.b33bba89c8948p-1, 0x1.ea6a51d1f6ca9p-55,
    0x1.110d0c4b69c3bp-1, 0x1.d918998809981p-55, 0x1.b11d04162a4c6p-1, 0x1.1dd561efbc0c2p-56,
    0x1.146d21f8b7f82p-1, 0x1.bf9535e2739a8p-56, 0x1.aef78930bd275p-1, -0x1.f836279746f94p-56,
    0x1.17c8e5f2eedb0p-1, 0x1.35e57102e2488p-57, 0x1.accb526f69de5p-1, 0x1.8fb6a8dd6b6ccp-55,
    0x1.1b204acb02fddp-1, -0x1.f190c70cbb5ffp-58, 0x1.aa98688308913p-1, -0x1.b83d607cd5070p-63,
    0x1.1e7343236574cp-1, 0x1.22a3fa4f41d5ap-56, 0x1.a85ed4373e02dp-1, 0x1.9be06385ec792p-57,
    0x1.21c1c1b0394cfp-1, 0x1.e5b324b23aa31p-58, 0x1.a61e9e72586afp-1, 0x1.58330e2fd453fp-55,
    0x1.250bb93788bbbp-1, 0x1.ea3d02457bccep-56, 0x1.a3d7d0352bdcfp-1, -0x1.68dbaeca19669p-55,
    0x1.28511c917a067p-1, -0x1.01df1d9a16b70p-55, 0x1.a18a729aee445p-1, 0x1.95e25736c0358p-60,
    0x1.2b91dea88421ep-1, -0x1.fa371db216ab0p-55, 0x1.9f368ed912f85p-1, -0x1.1d200c5791606p-55,
    0x1.2ecdf279a3082p-1, 0x1.d3557e0e7e37ep-55, 0x1.9cdc2e3f25e5cp-1, 0x1.3f99112993f62p-55,
    0x1.32054b148bc4fp-1, 0x1.f6b42095a135bp-55, 0x1.9a7b5a36a6514p-1, 0x1.722cfcc9fa7a9p-55,
    0x1.3537db9be0367p-1, 0x1.b327e7af040f0p-57, 0x1.98141c42e1310p-1, 0x1.d1ff80488f08dp-55,
    0x1.386597456282bp-1, -0x1.10fada93b07a8p-56, 0x1.95a67e00cb1fdp-1, -0x1.0befda21f862dp-55,
    0x1.3b8e715a2840ap-1, -0x1.97653a7d2f07bp-56, 0x1.93328926d9e92p-1, -0x1.bb77003600cdap-55,
    0x1.3eb25d36cd53ap-1, -0x1.be570e1570fc0p-58, 0x1.90b84784ddaf7p-1, -0x1.0feb10ab93b87p-56,
    0x1.41d14e4ba6790p-1, 0x1.4608fd287ecf5p-55, 0x1.8e37c303d9ad1p-1, -0x1.463a4b53d4bf8p-57,
    0x1.44eb381cf386bp-1, -0x1.3ed6c1e6a5505p-55, 0x1.8bb105a5dc900p-1, 0x1.863e03e9474c1p-55,
    0x1.48000e431159fp-1, -0x1.b194a7463ed10p-55, 0x1.89241985d871fp-1, 0x1.c48d9c413ed84p-55,
    0x1.4b0fc46aab761p-1, 0x1.0da05738cc59ap-61, 0x1.869108d77a6c6p-1, 0x1.338ffe2bfe9ddp-56,
    0x1.4e1a4e54ed51bp-1, -0x1.a492f89b7c76ap-55, 0x1.83f7dde701ca0p-1, -0x1.152cf609bc6e8p-59,
    0x1.511f9fd7b351cp-1, -0x1.5c0e861c48831p-55, 0x1.8158a31916d5dp-1, -0x1.de8b90b8228dep-57,
    0x1.541facddbb724p-1, 0x1.232c28520d391p-56, 0x1.7eb362eaa1488p-1, 0x1.a1d65a4a5959fp-58,
    0x1.571a6966d59b3p-1, 0x1.c843b4d0fb198p-58, 0x1.7c0827f09e54fp-1, -0x1.c73d6d72aee68p-57,
    0x1.5a0fc98813a12p-1, -0x1.d82e2b7d4227bp-55, 0x1.7956fcd7f6543p-1, -0x1.ab276e9d45ae4p-55,
    0x1.5cffc16bf8f0dp-1, 0x1.96cb370eb578ap-55, 0x1.769fec655211fp-1, -0x1.827d5cf8c68c5p-57,
    0x1.5fea4552a9e57p-1, 0x1.0b6cef7ee20b7p-55, 0x1.73e30174efba1p-1, -0x1.5d3ae3d94ad5fp-57,
    0x1.62cf49921ac79p-1, -0x1.edd9855b6241ap-55, 0x1.712046fa77678p-1, 0x1.425b0a5029c81p-55,
    0x1.65aec2963e755p-1, 0x1.126f96b71053cp-55, 0x1.6e57c800cf55ep-1, 0x1.60286dedbd0a6p-55,
    0x1.6888a4e134b2fp-1, -0x1.6b7d37644d5e6p-55, 0x1.6b898fa9efb5dp-1, 0x1.15ac786ccf4b2p-56,
    0x1.6b5ce50b7821ap-1, -0x1.5d5158f702e0fp-57, 0x1.68b5a92eb6253p-1, -0x1.9a91ad985f89cp-55,
    0x1.6e2b77c40bde1p-1, -0x1.0e729857fad53p-56, 0x1.65dc1fdeb8cbap-1, -0x1.97c1b47337c77p-58,
    0x1.70f451d0a8c40p-1, 0x1.97ede3885770dp-57, 0x1.62fcff20191c7p-1, 0x1.d9143895756efp-57,
    0x1.73b7680dea578p-1, -0x1.2248306dc12a2p-56, 0x1.6018526f563dfp-1, 0x1.46ca5e0e432d0p-55,
    0x1.7674af6f7b524p-1, 0x1.e9d3f94ac84a8p-56, 0x1.5d2e255f1f17ap-1, 0x1.0314104c8892bp-55,
    0x1.792c1d0041d52p-1, -0x1.abf05eeb354ebp-55, 0x1.5a3e839824077p-1, 0x1.428aa2759be62p-55,
    0x1.7bdda5e28b3c2p-1, 0x1.ad1197ccd0393p-59, 0x1.574978d8e83f2p-1, 0x1.f4714af282d23p-55,
    0x1.7e893f5037959p-1, 0x1.0eefbaa650c4cp-55, 0x1.544f10f592ca5p-1, -0x1.e7ae8e6c7a62fp-55,
    0x1.812ede9ae4ba4p-1, -0x1.7830adf402ddap-55, 0x1.514f57d7bf3dap-1, 0x1.47a108073c259p-56,
};
GX_DEV double gx_sd_fnma(double a, double b, double c) { return fma(-a, b, c); }   // x86 vfnmadd: -(a * b) + c, one rounding
GX_DEV double gx_sd_taylor_sin(double xx, double x, double dx) {   // TAYLOR_SIN
    double p = fma(xx, -0x1.addffc2fcdf59p-26, 0x1.71de27b9a7ed9p-19);
    p = fma(xx, p, -0x1.a01a019db08b8p-13);
    p = fma(xx, p, 0x1.1111111110ecep-7);
    p = fma(xx, p, -0x1.5555555555555p-3);
    const double t = fma(xx, fma(p, x, -(0.5 * dx)), dx);
    return x + t;
}
GX_DEV double gx_sd_do_sin(double x, double dx) {
    if (fabs(x) < 0.126) return gx_sd_taylor_sin(x * x, x, dx);
    const double xold = x;
    if (x <= 0) dx = -dx;
    const double big = 0x1.8p45;
    const double u = big + fabs(x);
    x = fabs(x) - (u - big);
    const double xx = x * x;
    const double s = x + fma(x * xx, fma(xx, 0x1.11110e829872fp-7, -0x1.5555555555515p-3), dx);
    const double c = fma(x, dx, xx * fma(xx, fma(xx, 0x1.6c16bedd9e239p-10, -0x1.5555555555535p-5), 0.5));
    const int k = (int)((unsigned)__double2loint(u) << 2);
    const double sn = gx_sincostab[k], ssn = gx_sincostab[k + 1], cs = gx_sincostab[k + 2], ccs = gx_sincostab[k + 3];
    const double cor = fma(s, cs, gx_sd_fnma(c, sn, fma(s, ccs, ssn)));
    return copysign(sn + cor, xold);
}
GX_DEV double gx_sd_do_cos(double x, double dx) {
    if (x < 0) dx = -dx;
    const double big = 0x1.8p45;
    const double u = big + fabs(x);
    x = fabs(x) - (u - big) + dx;
    const double xx = x * x;
    const double s = fma(x * xx, fma(xx, 0x1.11110e829872fp-7, -0x1.5555555555515p-3), x);
    const double c = xx * fma(xx, fma(xx, 0x1.6c16bedd9e239p-10, -0x1.5555555555535p-5), 0.5);
    const int k = (int)((unsigned)__double2loint(u) << 2);
    const double sn = gx_sincostab[k], ssn = gx_sincostab[k + 1], cs = gx_sincostab[k + 2], ccs = gx_sincostab[k + 3];
    const double cor = gx_sd_fnma(s, sn, gx_sd_fnma(c, cs, gx_sd_fnma(s, ssn, ccs)));
    return cs + cor;
}
GX_DEV int gx_sd_reduce(double x, double *a, double *da) {   // reduce_sincos: |x| < 105414350
    const double toint = 0x1.8p52;
    const double t = fma(x, 0x1.45f306dc9c883p-1, toint);
    const double xn = t - toint;
    const double y = gx_sd_fnma(xn, -0x1.dde973c000000p-27, gx_sd_fnma(xn, 0x1.921fb58000000p+0, x));
    const int n = __double2loint(t) & 3;
    const double pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
    const double t2 = gx_sd_fnma(xn, pp3, y);
    double db = gx_sd_fnma(pp3, xn, y - t2);
    const double b = gx_sd_fnma(xn, pp4, t2);
    db += gx_sd_fnma(xn, pp4, t2 - b);
    *a = b; *da = db;
    return n;
}
GX_DEV double gx_sd_do_sincos(double a, double da, int n) {
    const double r = (n & 1) ? gx_sd_do_cos(a, da) : gx_sd_do_sin(a, da);
    return (n & 2) ? -r : r;
}
// valid for |x| < 105414350 (the path's arguments are float-valued angles in [0, 2 pi]); beyond it OCML answers
GX_DEV double gx_sin_d(double x) {
    const int k = __double2hiint(x) & 0x7fffffff;
    if (k < 0x3e500000) return x;
    if (k < 0x3feb6000) return gx_sd_do_sin(x, 0.0);
    if (k < 0x400368fd) return copysign(gx_sd_do_cos(0x1.921fb54442d18p+0 - fabs(x), 0x1.1a62633145c07p-54), x);
    if (k < 0x419921FB) { double a, da; const int n = gx_sd_reduce(x, &a, &da); return gx_sd_do_sincos(a, da, n); }
    return sin(x);
}
GX_DEV double gx_cos_d(double x) {
    const int k = __double2hiint(x) & 0x7fffffff;
    if (k < 0x3e400000) return 1.0;
    if (k < 0x3feb6000) return gx_sd_do_cos(x, 0.0);
    if (k < 0x400368fd) {
        const double hp1 = 0x1.1a62633145c07p-54;
        const double y = 0x1.921fb54442d18p+0 - fabs(x);
        const double a = y + hp1;
        const double da = (y - a) + hp1;
        return gx_sd_do_sin(a, da);
    }
    if (k < 0x419921FB) { double a, da; const int n = gx_sd_reduce(x, &a, &da); return gx_sd_do_sincos(a, da, n + 1); }
    return cos(x);
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
    // (one square root and one division for both branches: the operands are selected first)
    const bool xBig = fabsf(v1.x) > fabsf(v1.y);
    const float a = xBig ? v1.x : v1.y;
    const V3 num = xBig ? V3(-v1.z, 0, v1.x) : V3(0, v1.z, -v1.y);
    *v2 = num / gx_sqrt(a * a + v1.z * v1.z);
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

// valu_probe.hip -- issue rate of single VALU instruction kinds on gfx950 (dev tool; results in profiles/README.md).
// Every kernel runs 8 independent chains of one instruction per lane, 2048 blocks of 256 threads (8 waves per SIMD): the loop is bound
// by how fast a SIMD issues that instruction, nothing else.  build: hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define PROBE(NAME, DECL, BODY, SINK)                                                            \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters, float a, float b) {       \
        DECL                                                                                     \
        for (int i = 0; i < iters; ++i) {                                                        \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) { REP8(BODY) }                          \
        }                                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = SINK;                                       \
    }
#define FDECL float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
#define FSINK ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7))
#define IDECL unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
#define ISINK __uint_as_float(((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7)))
#define DDECL double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; double da = a, db = b;
#define DSINK (float)(((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7)))

#define B_FMA(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x##n) : "v"(a), "v"(b));
#define B_MUL(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x##n) : "v"(a));
#define B_ADD(n) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x##n) : "v"(a));
#define B_MAX3(n) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x##n) : "v"(a), "v"(b));
#define B_MIN(n) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x##n) : "v"(a));
#define B_CNDMASK(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x##n) : "v"(a) : );
#define B_CMP(n) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x##n), "v"(a) : "vcc");
#define B_CMPSEL(n) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x##n) : "v"(a), "v"(b) : "vcc");
#define B_CMPS(n) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(x##n), "v"(a) : "s20", "s21");
#define B_AND(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x##n) : "v"(ua));
#define B_ADDU(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x##n) : "v"(ua));
#define B_LSHL(n) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x##n));
#define B_LSHLADD(n) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x##n) : "v"(ua));
#define B_MULLO(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x##n) : "v"(ua));
#define B_MULHI(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x##n) : "v"(ua));
#define B_MAD24(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x##n) : "v"(ua), "v"(ub));
#define B_CVT(n) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x##n));
#define B_RCP(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(x##n));
#define B_SQRT(n) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x##n));
#define B_DIVFIX(n) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x##n) : "v"(a), "v"(b));
#define B_PKMUL(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##n) : "v"(pa));
#define B_FMA64(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x##n) : "v"(da), "v"(db));
#define B_MUL64(n) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x##n) : "v"(da));
#define B_ADD64(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x##n) : "v"(da));
#define B_MADU64(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##n) : "v"(ua), "v"(ub) : "vcc");
#define B_MOV(n) asm volatile("v_mov_b32 %0, %1" : "=v"(x##n) : "v"(a));
#define B_BFE(n) asm volatile("v_bfe_u32 %0, %0, 1, 8" : "+v"(x##n));
#define B_MED3(n) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x##n) : "v"(a), "v"(b));
#define B_SUBREV(n) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x##n) : "v"(a));

PROBE(k_fma, FDECL, B_FMA, FSINK) PROBE(k_mul, FDECL, B_MUL, FSINK) PROBE(k_add, FDECL, B_ADD, FSINK) PROBE(k_max3, FDECL, B_MAX3, FSINK) PROBE(k_min, FDECL, B_MIN, FSINK)
PROBE(k_cndmask, FDECL, B_CNDMASK, FSINK) PROBE(k_cmp, FDECL, B_CMP, FSINK) PROBE(k_cmpsel, FDECL, B_CMPSEL, FSINK) PROBE(k_cmps, FDECL, B_CMPS, FSINK)
PROBE(k_and, IDECL, B_AND, ISINK) PROBE(k_addu, IDECL, B_ADDU, ISINK) PROBE(k_lshl, IDECL, B_LSHL, ISINK) PROBE(k_lshladd, IDECL, B_LSHLADD, ISINK) PROBE(k_mullo, IDECL, B_MULLO, ISINK)
PROBE(k_mulhi, IDECL, B_MULHI, ISINK) PROBE(k_mad24, IDECL, B_MAD24, ISINK) PROBE(k_cvt, FDECL, B_CVT, FSINK) PROBE(k_rcp, FDECL, B_RCP, FSINK) PROBE(k_sqrt, FDECL, B_SQRT, FSINK)
PROBE(k_divfix, FDECL, B_DIVFIX, FSINK) PROBE(k_fma64, DDECL, B_FMA64, DSINK) PROBE(k_mul64, DDECL, B_MUL64, DSINK) PROBE(k_add64, DDECL, B_ADD64, DSINK)
PROBE(k_mov, FDECL, B_MOV, FSINK) PROBE(k_bfe, IDECL, B_BFE, ISINK) PROBE(k_med3, FDECL, B_MED3, FSINK) PROBE(k_subrev, FDECL, B_SUBREV, FSINK)
typedef float f2 __attribute__((ext_vector_type(2)));
#define PDECL f2 p0 = {(float)threadIdx.x, 1.f}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f; f2 pa = {a, b};
#define PSINK (((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7))).x
PROBE(k_pkmul, PDECL, B_PKMUL, PSINK)
#define QDECL unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3, q4 = q0 + 4, q5 = q0 + 5, q6 = q0 + 6, q7 = q0 + 7; unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
#define QSINK (float)(((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7)))
PROBE(k_madu64, QDECL, B_MADU64, QSINK)

int main() {
    hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount, blocks = cus * 8, iters = 2000;
    float *out; CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    struct P { const char *name; void (*k)(float *, int, float, float); };
    std::vector<P> ps = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_sub_f32", k_subrev}, {"v_max3_f32", k_max3}, {"v_med3_f32", k_med3}, {"v_min_f32", k_min}, {"v_mov_b32", k_mov},
                         {"v_cndmask_b32", k_cndmask}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr", k_cmps}, {"v_cmp+v_cndmask (2 instr)", k_cmpsel}, {"v_and_b32", k_and}, {"v_add_u32", k_addu}, {"v_lshlrev_b32", k_lshl},
                         {"v_lshl_add_u32", k_lshladd}, {"v_bfe_u32", k_bfe}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mad_u32_u24", k_mad24}, {"v_mad_u64_u32", k_madu64}, {"v_cvt_f32_u32", k_cvt},
                         {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_div_fixup_f32", k_divfix}, {"v_pk_mul_f32", k_pkmul}, {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}};
    printf("%d CUs, clock %d MHz; G wave-instr/s and cycles per wave-instruction per SIMD (at the reported clock)\n", cus, pr.clockRate / 1000);
    for (auto &p : ps) {
        hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
        CHK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double insts = (double)blocks * 4 /*waves*/ * iters * 64.0 * (strstr(p.name, "2 instr") ? 2 : 1);
        const double rate = insts / (best * 1e-3);
        printf("%-28s %8.1f G/s   %5.2f cyc\n", p.name, rate * 1e-9, (double)cus * 4 * pr.clockRate * 1e3 / rate);
    }
    return 0;
}

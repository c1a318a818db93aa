// Probe (dev tool): how many per-lane 16-byte gathers per second does an MI355X sustain from an L2 / Infinity-Cache-resident table?
//   mode 0: every lane reads the 8 dwordx4 of ITS OWN 128-byte record (the access pattern of a BVH node step: 8 instructions, 64 lines each)
//   mode 1: 8 lanes share a record, lane j reads chunk j (8 instructions cover 64 records: 8 lines per instruction)
//   mode 2: every lane reads ONE dwordx4 of its own record per step (1 instruction, 64 lines)
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/gather_probe.hip -o gather_probe ; run: ./gather_probe [table_MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void __launch_bounds__(256) k_gather(const float4 *__restrict__ tab, unsigned nrec, int iters, int mode, float *out) {
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        idx = idx * 1664525u + 1013904223u;
        unsigned rec = (idx >> 8) % nrec;
        if (mode == 0) {
            const float4 *p = tab + (size_t)rec * 8;
            float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
            acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        } else if (mode == 1) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                unsigned rr = __shfl(rec, (lane & ~7) + r);
                float4 a = tab[(size_t)rr * 8 + (lane & 7)];
                s += a.x + a.w;
            }
            acc += s;
        } else if (mode == 2) {
            float4 a = tab[(size_t)rec * 8 + (it & 7)];
            acc += a.x + a.w;
        } else if (mode == 3) {   // own 128-byte record inside a 16 KB window: L1 hits, 8 instructions x 64 lines
            const float4 *p = tab + (size_t)(rec & 127u) * 8;
            float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
            acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        } else if (mode == 9) {
            acc += 1.f;
        } else if (mode == 4) {   // own 64-byte record (4 dwordx4): half a line
            const float4 *p = tab + (size_t)rec * 8 + ((idx >> 4) & 1u) * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (mode == 6) {   // 8 random dword gathers inside a 32 KB window (L1 hits): the Halton permutation-table pattern
            const float *p = reinterpret_cast<const float *>(tab);
            unsigned j = idx >> 8;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) { s += p[(j >> r) & 8191u]; j = j * 747796405u + 2891336453u; }
            acc += s;
        } else if (mode == 7) {   // 8 random ushort gathers inside a 32 KB window
            const unsigned short *p = reinterpret_cast<const unsigned short *>(tab);
            unsigned j = idx >> 8;
            unsigned s = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) { s += p[(j >> r) & 16383u]; j = j * 747796405u + 2891336453u; }
            acc += (float)s;
        } else if (mode == 8) {   // 8 dword loads of the SAME address for all lanes (uniform table read through the vector path)
            const float *p = reinterpret_cast<const float *>(tab);
            unsigned j = __shfl(idx >> 8, 0);
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) { s += p[(j >> r) & 8191u]; j = j * 747796405u + 2891336453u; }
            acc += s;
        } else {                  // two 64-byte halves of two different records (8 dwordx4, 2 lines)
            const float4 *p = tab + (size_t)rec * 8, *q = tab + (size_t)((rec * 7919u) % nrec) * 8 + 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3], e = q[0], f = q[1], g = q[2], h = q[3];
            acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        }
        idx ^= __float_as_uint(acc) & 1u;   // dependent chain like a traversal
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main(int argc, char **argv) {
    size_t mb = argc > 1 ? atoi(argv[1]) : 8;
    unsigned nrec = (unsigned)(mb * 1024 * 1024 / 128);
    float4 *tab; float *out;
    hipMalloc(&tab, (size_t)nrec * 128); hipMemset(tab, 0, (size_t)nrec * 128);
    int per_cu = argc > 2 ? atoi(argv[2]) : 5;
    int blocks = 256 * per_cu; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 10; ++mode) {
        int iters = 2000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a); hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, nrec, iters, mode, out); hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        double loads = (double)blocks * 256 * iters * (mode == 2 ? 1 : (mode == 4 ? 4 : 8));
        printf("blocks/CU %d table %zu MB mode %d: %.3f ms, %.3e lane-loads(16B)/s = %.2f TB/s, %.1f B/clk/CU @2.4GHz\n", per_cu, mb, mode, ms, loads / (ms * 1e-3), loads * 16 / (ms * 1e-3) / 1e12, loads * 16 / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}

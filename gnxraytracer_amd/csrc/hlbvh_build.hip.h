// hlbvh_build.hip.h -- BVHAccel::HLBVHBuild (accelerator/BVHAccel.cpp:369-626) on the device, SURVEY 8(f).4.
//
//   Morton codes of the centroids             BVHAccel.cpp:377-394, EncodeMorton3 / LeftShift3 :68-100     k_morton_codes
//   stable LSD radix sort, 5 passes x 6 bits  RadixSort :102-141                                          k_rs_hist / scan / k_rs_scatter (own sort)
//   one LBVH per run of equal top 12 bits     :402-421, emitLBVH :462-524                                 k_hl_runs, k_hl_leaves, k_hl_internal, k_hl_fit
//   SAH over the treelet roots                buildUpperSAH :526-626                                      k_hl_upper_level (one wave per range, one launch per level)
//
// emitLBVH splits a sorted code range at the highest bit in which its first and last code differ and makes leaves only where the bits
// run out (equal codes): that is the binary radix tree of the distinct codes of a treelet, which has a closed form per node (Karras 2012:
// every internal node finds its own range and split from common-prefix lengths), so all nodes of all treelets are emitted in one pass
// instead of by recursion.  Bounds are unions (min / max: exact in any order), fitted bottom-up.  buildUpperSAH's partition decides only
// WHICH roots go left -- bucket counts, bucket boxes and costs do not depend on the order inside a range -- so its std::partition is
// replaced by a scan-based one.  The result is the reference's tree node for node (tests/test_gpu_parity.py::
// test_hlbvh_build_matches_reference compares the flattened LinearBVHNode[] and the primitive order with a dump of the compiled reference).
#pragma once
#include <hip/hip_runtime.h>

#include "host_scene.h"

namespace gnxr {
namespace hlbvh {

constexpr int kB = 256;                 // threads per block
constexpr int kTile = kB * 8;           // items per block in the scan / sort passes

__device__ __forceinline__ uint32_t left_shift3(uint32_t x) {
    if (x == (1u << 10)) --x;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}
static __global__ void __launch_bounds__(kB) k_morton_codes(const float *__restrict__ cen, int n, float lx, float ly, float lz, float hx, float hy, float hz,
                                                            uint32_t *__restrict__ codes, uint32_t *__restrict__ prims) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float ox = cen[3 * (size_t)i] - lx, oy = cen[3 * (size_t)i + 1] - ly, oz = cen[3 * (size_t)i + 2] - lz;   // Bounds3::Offset
        if (hx > lx) ox /= hx - lx;
        if (hy > ly) oy /= hy - ly;
        if (hz > lz) oz /= hz - lz;
        const float sc = (float)(1 << 10);   // mortonScale
        codes[i] = (left_shift3((uint32_t)(oz * sc)) << 2) | (left_shift3((uint32_t)(oy * sc)) << 1) | left_shift3((uint32_t)(ox * sc));
        prims[i] = (uint32_t)i;
    }
}

// ---- exclusive scan of n unsigned values (in place), total to *total: per-tile scan, one block over the tile sums, add back
static __global__ void __launch_bounds__(kB) k_scan_tiles(uint32_t *__restrict__ v, int n, uint32_t *__restrict__ tile_sums) {
    __shared__ uint32_t wsum[kB / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * kTile + (size_t)threadIdx.x * 8;
    uint32_t x[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { x[k] = base + k < (size_t)n ? v[base + k] : 0u; s += x[k]; }
    uint32_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(inc, off); if (lane >= off) inc += y; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    uint32_t run = woff + inc - s;
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (base + k < (size_t)n) v[base + k] = run; run += x[k]; }
    if (threadIdx.x == kB - 1) tile_sums[blockIdx.x] = woff + inc;
}
static __global__ void __launch_bounds__(1024) k_scan_sums(uint32_t *__restrict__ sums, int m, uint32_t *__restrict__ total) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < m ? sums[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if (lane >= off) x += y; }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wtot[w];
        const uint32_t carry = carry_s;
        if (i < m) sums[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry_s;
}
static __global__ void __launch_bounds__(kB) k_scan_add(uint32_t *__restrict__ v, int n, const uint32_t *__restrict__ tile_offsets) {
    const uint32_t o = tile_offsets[blockIdx.x];
    const size_t base = (size_t)blockIdx.x * kTile + (size_t)threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (base + k < (size_t)n) v[base + k] += o;
}

// ---- one pass of the LSD radix sort (6 bits): per-tile digit histograms, [digit][tile] layout so that one scan gives every tile its
// output offset for every digit; the scatter keeps equal digits in input order (stable, as RadixSort's counting pass is)
static __global__ void __launch_bounds__(kB) k_rs_hist(const uint32_t *__restrict__ keys, int n, int shift, int n_tiles, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kTile;
    for (int k = 0; k < 8; ++k) {
        const size_t i = base + (size_t)k * kB + threadIdx.x;
        if (i < (size_t)n) atomicAdd(&h[(keys[i] >> shift) & 63u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}
static __global__ void __launch_bounds__(kB) k_rs_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, int n, int shift, int n_tiles,
                                                          const uint32_t *__restrict__ offsets, uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t run[64];                 // next output position of each digit for this tile
    __shared__ uint32_t wc[kB / 64][64];         // per wave: items of each digit in the current chunk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) run[threadIdx.x] = offsets[(size_t)threadIdx.x * n_tiles + blockIdx.x];
    const size_t base = (size_t)blockIdx.x * kTile;
    for (int k = 0; k < 8; ++k) {
        for (int d = threadIdx.x; d < (kB / 64) * 64; d += kB) (&wc[0][0])[d] = 0;
        __syncthreads();
        const size_t i = base + (size_t)k * kB + threadIdx.x;
        const bool valid = i < (size_t)n;
        const uint32_t key = valid ? keys[i] : 0u, val = valid ? vals[i] : 0u;
        const uint32_t dg = (key >> shift) & 63u;
        unsigned long long peers = __ballot(valid);   // lanes of this wave with the same digit
#pragma unroll
        for (int b = 0; b < 6; ++b) { const unsigned long long m = __ballot((dg >> b) & 1u); peers &= ((dg >> b) & 1u) ? m : ~m; }
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wc[wave][dg] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t pos = run[dg] + rank;
            for (int w = 0; w < wave; ++w) pos += wc[w][dg];
            keys_out[pos] = key; vals_out[pos] = val;
        }
        __syncthreads();
        if (threadIdx.x < 64) { uint32_t t = 0; for (int w = 0; w < kB / 64; ++w) t += wc[w][threadIdx.x]; run[threadIdx.x] += t; }
        __syncthreads();
    }
}

// ---- runs of equal codes (the LBVH leaves) and runs of equal top 12 bits (the treelets)
static __global__ void __launch_bounds__(kB) k_hl_flags(const uint32_t *__restrict__ codes, int n, uint32_t *__restrict__ head) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) head[i] = (i == 0 || codes[i] != codes[i - 1]) ? 1u : 0u;
}
// head_scan: exclusive scan of the head flags = index of the run an element belongs to, minus one for non-heads ... so run(i) = scan[i] + head - 1
static __global__ void __launch_bounds__(kB) k_hl_runs(const uint32_t *__restrict__ codes, const uint32_t *__restrict__ head_scan, int n, uint32_t *__restrict__ ukey, uint32_t *__restrict__ ustart) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const bool is_head = i == 0 || codes[i] != codes[i - 1];
        if (is_head) { const uint32_t u = head_scan[i]; ukey[u] = codes[i]; ustart[u] = (uint32_t)i; }
    }
}
static __global__ void __launch_bounds__(kB) k_hl_tflags(const uint32_t *__restrict__ ukey, int U, uint32_t *__restrict__ thead) {
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) thead[u] = (u == 0 || (ukey[u] >> 18) != (ukey[u - 1] >> 18)) ? 1u : 0u;
}

// leaf u = the primitives of run u, in sorted order (emitLBVH's bitIndex == -1 case, BVHAccel.cpp:467-480)
static __global__ void __launch_bounds__(kB) k_hl_leaves(const uint32_t *__restrict__ ustart, int U, int n, const uint32_t *__restrict__ prims, const float *__restrict__ pb6,
                                                         HlbvhNode *__restrict__ nodes, int *__restrict__ failed) {
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) {
        const int first = (int)ustart[u], end = u + 1 < U ? (int)ustart[u + 1] : n;
        float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
        for (int i = first; i < end; ++i) {
            const float *b = pb6 + 6 * (size_t)prims[i];
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], b[a]); hi[a] = fmaxf(hi[a], b[3 + a]); }
        }
        HlbvhNode nd;
        for (int a = 0; a < 3; ++a) { nd.b[a] = lo[a]; nd.b[3 + a] = hi[a]; }
        nd.child[0] = nd.child[1] = -1; nd.axis = 0; nd.first = first; nd.n = end - first;
        nodes[u] = nd;
        if (end - first > 0xffff) *failed = 1;   // LinearBVHNode::nPrimitives is a uint16_t
    }
}

// internal node U + i covers a range of runs of ONE treelet and splits it at the highest differing bit (emitLBVH, BVHAccel.cpp:481-523);
// it exists when runs i and i + 1 belong to the same treelet.  delta(i, j) = common leading bits of the two distinct codes, -1 across treelets.
static __global__ void __launch_bounds__(kB) k_hl_internal(const uint32_t *__restrict__ ukey, int U, HlbvhNode *__restrict__ nodes, int *__restrict__ parent) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < U - 1; i += gridDim.x * blockDim.x) {
        const uint32_t ki = ukey[i];
        auto delta = [&](int j) -> int {
            if (j < 0 || j >= U) return -1;
            const uint32_t kj = ukey[j];
            if ((kj >> 18) != (ki >> 18)) return -1;
            return __clz((int)(ki ^ kj));
        };
        if (delta(i + 1) < 0) continue;   // run i is the last of its treelet: no internal node here
        const int d = (delta(i + 1) - delta(i - 1)) > 0 ? 1 : -1;
        const int dmin = delta(i - d);
        int lmax = 2;
        while (delta(i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2) if (delta(i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = delta(j);
        int s = 0;
        for (int t = (l + 1) / 2;; t = (t + 1) / 2) {   // ceil halving: t = l/2, l/4, ... , 1
            if (delta(i + (s + t) * d) > dnode) s += t;
            if (t == 1) break;
        }
        const int gamma = i + s * d + (d < 0 ? d : 0);
        const int lo = d > 0 ? i : j, hi = d > 0 ? j : i;
        const int c0 = (lo == gamma) ? gamma : U + gamma, c1 = (hi == gamma + 1) ? gamma + 1 : U + gamma + 1;
        HlbvhNode nd;
        for (int a = 0; a < 6; ++a) nd.b[a] = 0.f;
        nd.child[0] = c0; nd.child[1] = c1;
        nd.axis = (31 - dnode) % 3;   // bitIndex % 3: the bit the range splits at
        nd.first = 0; nd.n = 0;
        nodes[U + i] = nd;
        parent[c0] = U + i; parent[c1] = U + i;
    }
}
// bounds of the internal nodes, bottom-up: the second child to arrive at a node forms the union and moves on
static __global__ void __launch_bounds__(kB) k_hl_fit(int U, HlbvhNode *__restrict__ nodes, const int *__restrict__ parent, unsigned int *__restrict__ arrived) {
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) {
        int p = parent[u];
        while (p >= 0) {
            __threadfence();
            if (atomicAdd(&arrived[p], 1u) == 0u) break;   // the first child waits for nobody: the sibling's thread finishes the node
            __threadfence();
            // the sibling's box was written by another CU: read it past this CU's vector cache (agent-scope loads)
            const float *a = nodes[nodes[p].child[0]].b, *b = nodes[nodes[p].child[1]].b;
            for (int k = 0; k < 3; ++k) {
                nodes[p].b[k] = fminf(__hip_atomic_load(a + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(b + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                nodes[p].b[3 + k] = fmaxf(__hip_atomic_load(a + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(b + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            p = parent[p];
        }
    }
}
// root of each treelet, in code order (`treeletsToBuild`, BVHAccel.cpp:402-421): the internal node at the treelet's first run, or that run's leaf
static __global__ void __launch_bounds__(kB) k_hl_roots(const uint32_t *__restrict__ ukey, const uint32_t *__restrict__ thead_scan, int U, int *__restrict__ roots) {
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) {
        const bool is_head = u == 0 || (ukey[u] >> 18) != (ukey[u - 1] >> 18);
        if (!is_head) continue;
        const bool single = u + 1 >= U || (ukey[u + 1] >> 18) != (ukey[u] >> 18);
        roots[thead_scan[u]] = single ? u : U + u;
    }
}

// ---- buildUpperSAH (BVHAccel.cpp:526-626) over the T <= 4096 treelet roots, one LEVEL of the recursion per launch: every range of the
// level is split by one wave (bounds, the 12 buckets, their costs, the partition), the two halves become ranges of the next level.
// Ranges of a level are disjoint, so their waves share nothing but the node counter and the output list.
constexpr int kNB = 12;
struct UpRange { int start, end, slot; };   // roots[start, end); slot: (parent node << 1 | child) to link the new node into, -1: it is the root
__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int wave_sum(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
static __global__ void __launch_bounds__(kB) k_hl_upper_level(const UpRange *__restrict__ in, int n_in, UpRange *__restrict__ out, int *__restrict__ n_out, int *__restrict__ roots,
                                                              int *__restrict__ tmp, HlbvhNode *__restrict__ nodes, int upper_base, int *__restrict__ next_node,
                                                              int *__restrict__ root_out, int *__restrict__ failed) {
    const int lane = threadIdx.x & 63;
    const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const float FMAX = 3.402823466e+38f;
    for (int ri = wid; ri < n_in; ri += n_waves) {
        const int start = in[ri].start, end = in[ri].end, slot = in[ri].slot;
        // bounds of the range and of its centroids (an empty Bounds3 is (FLT_MAX, -FLT_MAX), as on the host)
        float lo[3] = {FMAX, FMAX, FMAX}, hi[3] = {-FMAX, -FMAX, -FMAX}, clo[3] = {FMAX, FMAX, FMAX}, chi[3] = {-FMAX, -FMAX, -FMAX};
        for (int i = start + lane; i < end; i += 64) {
            const HlbvhNode &nd = nodes[roots[i]];
            for (int a = 0; a < 3; ++a) {
                lo[a] = fminf(lo[a], nd.b[a]); hi[a] = fmaxf(hi[a], nd.b[3 + a]);
                const float c = (nd.b[a] + nd.b[3 + a]) * 0.5f;
                clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c);
            }
        }
        for (int a = 0; a < 3; ++a) { lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]); clo[a] = wave_min(clo[a]); chi[a] = wave_max(chi[a]); }
        const float dx = chi[0] - clo[0], dy = chi[1] - clo[1], dz = chi[2] - clo[2];
        const int dim = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);   // Bounds3::MaximumExtent
        const float cl = dim == 0 ? clo[0] : (dim == 1 ? clo[1] : clo[2]), ch = dim == 0 ? chi[0] : (dim == 1 ? chi[1] : chi[2]);
        if (!(ch > cl)) { if (lane == 0) *failed = 1; continue; }   // CHECK_NE(centroidBounds.pMax[dim], centroidBounds.pMin[dim])
        auto bucket = [&](int node) -> int {
            const HlbvhNode &nd = nodes[node];
            const float centroid = (nd.b[dim] + nd.b[3 + dim]) * 0.5f;
            int b = (int)(kNB * ((centroid - cl) / (ch - cl)));
            if (b == kNB) b = kNB - 1;
            return b;
        };
        // the 12 buckets: counts and boxes (every lane ends up with all of them)
        int cnt[kNB];
        float blo[kNB][3], bhi[kNB][3];
        bool bad = false;
        for (int b = 0; b < kNB; ++b) {
            int c = 0;
            float l3[3] = {FMAX, FMAX, FMAX}, h3[3] = {-FMAX, -FMAX, -FMAX};
            for (int i = start + lane; i < end; i += 64) {
                const int node = roots[i];
                const int bb = bucket(node);
                if (bb < 0 || bb >= kNB) bad = true;
                if (bb == b) {
                    ++c;
                    const HlbvhNode &nd = nodes[node];
                    for (int a = 0; a < 3; ++a) { l3[a] = fminf(l3[a], nd.b[a]); h3[a] = fmaxf(h3[a], nd.b[3 + a]); }
                }
            }
            cnt[b] = wave_sum(c);
            for (int a = 0; a < 3; ++a) { blo[b][a] = wave_min(l3[a]); bhi[b][a] = wave_max(h3[a]); }
        }
        if (__ballot(bad)) { if (lane == 0) *failed = 1; continue; }
        auto area = [](const float *l, const float *h) { const float x = h[0] - l[0], y = h[1] - l[1], z = h[2] - l[2]; return 2 * (x * y + x * z + y * z); };
        const float barea = area(lo, hi);
        float minCost = 0;
        int split = 0;
        for (int i = 0; i < kNB - 1; ++i) {
            float l0[3] = {FMAX, FMAX, FMAX}, h0[3] = {-FMAX, -FMAX, -FMAX}, l1[3] = {FMAX, FMAX, FMAX}, h1[3] = {-FMAX, -FMAX, -FMAX};
            int c0 = 0, c1 = 0;
            for (int j = 0; j <= i; ++j) { for (int a = 0; a < 3; ++a) { l0[a] = fminf(l0[a], blo[j][a]); h0[a] = fmaxf(h0[a], bhi[j][a]); } c0 += cnt[j]; }
            for (int j = i + 1; j < kNB; ++j) { for (int a = 0; a < 3; ++a) { l1[a] = fminf(l1[a], blo[j][a]); h1[a] = fmaxf(h1[a], bhi[j][a]); } c1 += cnt[j]; }
            const float cost = .125f + (c0 * area(l0, h0) + c1 * area(l1, h1)) / barea;
            if (i == 0 || cost < minCost) { minCost = cost; split = i; }
        }
        int nl = 0;
        for (int j = 0; j <= split; ++j) nl += cnt[j];
        const int mid = start + nl;
        if (nl <= 0 || nl >= end - start) { if (lane == 0) *failed = 1; continue; }   // CHECK_GT(mid, start) / CHECK_LT(mid, end)
        // partition: roots with bucket <= split first (which roots go left is all that matters, see the header)
        int lbase = start, rbase = mid;
        int firstL = -1, firstR = -1;   // the roots that land at positions `start` and `mid` (the children when a half is a single root)
        for (int c0 = start; c0 < end; c0 += 64) {
            const int i = c0 + lane;
            const bool valid = i < end;
            const int node = valid ? roots[i] : 0;
            const bool left = valid && bucket(node) <= split;
            const unsigned long long ml = __ballot(left), mv = __ballot(valid);
            const int lrank = __popcll(ml & ((1ull << lane) - 1ull)), vrank = __popcll(mv & ((1ull << lane) - 1ull));
            if (valid) {
                const int pos = left ? lbase + lrank : rbase + (vrank - lrank);
                tmp[pos] = node;
                if (pos == start) firstL = node;
                if (pos == mid) firstR = node;
            }
            lbase += __popcll(ml); rbase += __popcll(mv) - __popcll(ml);
        }
        for (int o = 32; o > 0; o >>= 1) { firstL = max(firstL, __shfl_xor(firstL, o)); firstR = max(firstR, __shfl_xor(firstR, o)); }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (int i = start + lane; i < end; i += 64) roots[i] = tmp[i];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (lane == 0) {
            const int me = upper_base + atomicAdd(next_node, 1);
            HlbvhNode nd;
            for (int a = 0; a < 3; ++a) { nd.b[a] = lo[a]; nd.b[3 + a] = hi[a]; }   // == Union(c0->bounds, c1->bounds): the union of the same roots' boxes
            nd.child[0] = mid - start == 1 ? firstL : -1;
            nd.child[1] = end - mid == 1 ? firstR : -1;
            nd.axis = dim; nd.first = 0; nd.n = 0;
            nodes[me] = nd;
            if (slot < 0) *root_out = me; else nodes[slot >> 1].child[slot & 1] = me;
            if (mid - start > 1) { const int o = atomicAdd(n_out, 1); out[o].start = start; out[o].end = mid; out[o].slot = me << 1; }
            if (end - mid > 1) { const int o = atomicAdd(n_out, 1); out[o].start = mid; out[o].end = end; out[o].slot = (me << 1) | 1; }
        }
    }
}

}  // namespace hlbvh
}  // namespace gnxr

// compact_kernel.hip.h -- stream compaction of the path queues without global atomics.
//
// Atomic appends cap out at ~88 increments/us per address on MI355X (MI355X_MICROARCH.md, row "dequeue"): with
// one append per wave the first version spent 1.3 ms per launch just binning 16 M paths (profiles/r01_c_*).
// Here each 256-item tile counts its survivors with wave64 ballots, one small block scans the tile counts, and the
// scatter pass recomputes the ballots to write every survivor at its final, ORDER-PRESERVING position: queues stay
// sorted by path slot, so the float4 state arrays keep being read in (mostly) ascending, coalesced order.
//
//   mode FLAGS : predicate o = bit o of pflags[path]   (bit0 path continues, bit1 has NEE record,
//                                                       bit2 shadow ray valid, bit3 MIS ray valid -- 2,3 counted only)
//   mode CLASS : predicate o = (pclass[path] == o)     (shade-kernel specialisation of the hit material)
#pragma once
#include "device_math.h"

namespace gnxr {

constexpr int kCompactBlock = 256;
enum CompactMode { COMPACT_FLAGS = 0, COMPACT_CLASS = 1, COMPACT_HITCLASS = 2 };

template <int MODE>
GX_DEV unsigned compact_key(const unsigned char *keys, int path) { return keys[path]; }
template <int MODE>
GX_DEV bool compact_pred(unsigned key, int o) { return MODE == COMPACT_FLAGS ? ((key >> o) & 1u) != 0 : key == (unsigned)o; }
//   mode HITCLASS (count pass only): a path whose ray hit a triangle gets its class from tri_class[hit[path]] here, and the pass leaves
//                 it in keys[path] for the scatter pass (mode CLASS); k_trace writes keys[path] itself only for misses and sphere hits, so
//                 that its retire step has no dependent gather

// A tile = kCompactTile items = kCompactChunks chunks of one block's width: 8 x fewer tile counts for the single-block scan of pass 2
// (1 M counts per predicate at 265 M paths took 0.28 ms per scan, 1.7 % of the GPU time of cfg 3; 130 k take 0.04 ms).
constexpr int kCompactChunks = 8;
constexpr int kCompactTile = kCompactBlock * kCompactChunks;

// pass 1: tile_counts[o * nTiles + tile] = number of items of the tile that satisfy predicate o
template <int MODE, int NOUT>
__global__ void __launch_bounds__(kCompactBlock) k_compact_count(const int *__restrict__ q_in, int n, const unsigned char *__restrict__ keys, unsigned int *tile_counts,
                                                                 int nTiles, const int *__restrict__ hit = nullptr, const unsigned char *__restrict__ tri_class = nullptr,
                                                                 unsigned char *keys_out = nullptr, int split = 0, const unsigned *n_dev = nullptr) {
    __shared__ unsigned int wsum[NOUT][kCompactBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // n_dev: the item count lives on the device (the device-driven path loop); `n` / `nTiles` then bound it, and nTiles stays the row stride
    // of tile_counts in all three passes
    if (n_dev) n = (int)*n_dev;
    const int tilesUsed = n_dev ? (n + kCompactTile - 1) / kCompactTile : nTiles;
    for (int tile = blockIdx.x; tile < tilesUsed; tile += gridDim.x) {
        unsigned acc[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) acc[o] = 0;
        for (int c = 0; c < kCompactChunks; ++c) {
            const long long i = (long long)tile * kCompactTile + c * kCompactBlock + threadIdx.x;
            unsigned key = 0xffu;
            const bool valid = i < n;
            int path = -1;
            if (valid) {
                path = q_in ? q_in[i] : (int)i;
                if (MODE == COMPACT_HITCLASS) {
                    const int h = hit[path];
                    if (h >= 0) { key = tri_class[h]; keys_out[path] = (unsigned char)key; }
                    else key = keys_out[path];   // (the same array as `keys`; read through the pointer that also writes it)
                } else key = compact_key<MODE>(keys, path);
            }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                // FLAGS, fifth count: continues and lives below `split` (the sub-pass in the lower half of the state arrays)
                const bool pr = (MODE == COMPACT_FLAGS && o == 4) ? ((key & 1u) != 0 && path < split) : compact_pred<MODE>(key, o);
                acc[o] += (unsigned)__popcll(__ballot(valid && pr));   // wave-uniform
            }
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) if (lane == 0) wsum[o][wave] = acc[o];
        __syncthreads();
        if (threadIdx.x < NOUT) {
            unsigned s = 0;
            for (int w = 0; w < kCompactBlock / 64; ++w) s += wsum[threadIdx.x][w];
            tile_counts[(size_t)threadIdx.x * nTiles + tile] = s;
        }
        __syncthreads();
    }
}

// pass 2: exclusive scan of each predicate's tile counts (one 1024-thread block per predicate), totals[o] = sum
static __global__ void __launch_bounds__(1024) k_compact_scan(unsigned int *tile_counts, int nTiles, unsigned int *totals, const unsigned *n_dev = nullptr) {
    __shared__ unsigned int wtot[16];
    __shared__ unsigned int carry_s;
    unsigned int *c = tile_counts + (size_t)blockIdx.x * nTiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    if (n_dev) nTiles = ((int)*n_dev + kCompactTile - 1) / kCompactTile;   // tiles in use; `c` keeps the row stride of the launch
    __syncthreads();
    for (int base = 0; base < nTiles; base += 1024) {
        int i = base + threadIdx.x;
        unsigned v = i < nTiles ? c[i] : 0u;
        unsigned x = v;  // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wave; ++w) woff += wtot[w];
        unsigned carry = carry_s;
        if (i < nTiles) c[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

// pass 3: write the survivors of the first NSCATTER predicates at tile_offset + rank-within-tile (chunk by chunk, in order)
template <int MODE, int NSCATTER>
__global__ void __launch_bounds__(kCompactBlock) k_compact_scatter(const int *__restrict__ q_in, int n, const unsigned char *__restrict__ keys,
                                                                   const unsigned int *__restrict__ tile_offsets, int nTiles, int *out0, int *out1, int *out2, int *out3 = nullptr,
                                                                   const unsigned *n_dev = nullptr) {
    __shared__ unsigned int wsum[NSCATTER][kCompactBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *outs[4] = {out0, out1, out2, out3};
    if (n_dev) n = (int)*n_dev;
    const int tilesUsed = n_dev ? (n + kCompactTile - 1) / kCompactTile : nTiles;
    for (int tile = blockIdx.x; tile < tilesUsed; tile += gridDim.x) {
        unsigned base[NSCATTER];
#pragma unroll
        for (int o = 0; o < NSCATTER; ++o) base[o] = tile_offsets[(size_t)o * nTiles + tile];
        for (int c = 0; c < kCompactChunks; ++c) {
            const long long i = (long long)tile * kCompactTile + c * kCompactBlock + threadIdx.x;
            unsigned key = 0xffu;
            const bool valid = i < n;
            int path = -1;
            if (valid) { path = q_in ? q_in[i] : (int)i; key = compact_key<MODE>(keys, path); }
            unsigned long long masks[NSCATTER];
#pragma unroll
            for (int o = 0; o < NSCATTER; ++o) {
                masks[o] = __ballot(valid && compact_pred<MODE>(key, o));
                if (lane == 0) wsum[o][wave] = (unsigned)__popcll(masks[o]);
            }
            __syncthreads();
#pragma unroll
            for (int o = 0; o < NSCATTER; ++o) {
                unsigned woff = 0, tot = 0;
                for (int w = 0; w < kCompactBlock / 64; ++w) { const unsigned v = wsum[o][w]; if (w < wave) woff += v; tot += v; }
                if (valid && compact_pred<MODE>(key, o)) outs[o][base[o] + woff + (unsigned)__popcll(masks[o] & ((1ull << lane) - 1ull))] = path;
                base[o] += tot;
            }
            __syncthreads();
        }
    }
}

}  // namespace gnxr

// trace_kernel.hip.h -- k_trace: the unified BVH traversal stage (BVHAccel::Intersect / IntersectP,
// accelerator/BVHAccel.cpp:653-729) for all three ray kinds a path vertex produces:
//   kind 0  continuation ray      closest hit  -> PathArrays::hit[path]
//   kind 1  shadow ray            any hit      -> sh_o[path].w  = 1 if unoccluded        (Light.cpp:28-31)
//   kind 2  MIS ray               closest hit  -> mis_o[path].w = 1 if it found what the light sample expects
//   (PathIntegrator launches pass TraceWork::vis and get both results as bytes of one word per path instead)
//                                                                                         (Integrator.cpp:193-203)
//
// CDNA4 structure (this is where the time goes, so it is shaped for wave64 rather than for one ray):
//   * persistent waves: a wave pulls CHUNK rays at a time from one global cursor (one atomic per CHUNK
//     rays -- a single word sustains only ~88 dequeues/us) and lanes refill from the wave's pool as soon as
//     their ray finishes, so short rays do not wait for the longest ray of the wave;
//   * "while-while" traversal: all lanes first walk interior nodes until each holds a leaf (or is done),
//     then all lanes run the triangle test together; the 250-instruction watertight test is no longer
//     executed for the whole wave every time one lane reaches a leaf (lane utilisation of the first
//     if-if version was 13 %, profiles/r01_b_*);
//   * leaves are still visited in the reference's order and tMax shrinks the same way, so hit records stay
//     bit-identical (ties resolve as in BVHAccel::Intersect);
//   * per-lane stack column in LDS: stack[depth * 256 + tid], bank == lane.  Only the first `lds_entries` levels live in
//     LDS (the host picks them so that LDS does not cap the occupancy below what the VGPR budget allows); the rare
//     deeper levels spill to a per-lane column in global memory.
#pragma once
#include "device_geom.h"

namespace gnxr {

struct TraceWork {
    const int *q_closest; int n_closest;   // path slots (nullptr == identity)
    const int *q_nee; int n_nee;           // paths with an NEE record: two work items each (shadow ray, MIS ray)
    const unsigned *order;                 // nullptr, or a permutation of the work items: position in the launch -> work item (a caller-supplied ordering; unused by the library itself)
    unsigned char *vis;                    // nullptr: visibility results go to sh_o[path].w / mis_o[path].w (float 1 / 0).  Otherwise 4 bytes per path:
                                           // [0] shadow ray unoccluded, [1] MIS ray found what the light sample expects (written here), [2] the record's
                                           // flags (written by k_shade) -- k_nee_combine then reads one word instead of three float4
    const unsigned *n_closest_dev, *n_nee_dev;   // nullptr, or where the two counts live on the device (the device-driven path loop: the host never reads
                                           // them; n_closest / n_nee then only bound the launch)
};
// the counts a traversal kernel works on: the device-side ones when the launch carries them
GX_DEV void trace_work_counts(TraceWork &w) {
    if (w.n_closest_dev) { w.n_closest = (int)*w.n_closest_dev; w.n_nee = w.n_nee_dev ? (int)*w.n_nee_dev : 0; }
}
// rays a wave takes per global atomic: `chunk_max` for big launches; for thin ones (late bounces) small enough that every wave gets a chunk --
// the number of atomics stays <= the number of waves, well under the ~88/us a single address sustains
GX_DEV int trace_chunk(unsigned total, int chunk_max) {
    const unsigned waves = gridDim.x * (blockDim.x / 64u);
    const unsigned per = ((total + waves - 1u) / waves + 63u) / 64u * 64u;
    return (int)min((unsigned)chunk_max, max(64u, per));
}

// Hand-out schedule of a launch's work items (k_trace4 / k_trace4d).  The global cursor counts CHUNKS, one atomic per fetch, and chunk v
// maps to a range of items by position: `c`-item chunks (512, or a wave's even share when the launch is thin) for most of the launch,
// then 128-item and finally 64-item chunks for the last ~1.5 chunks' worth of work per wave -- so the waves of a launch finish within a
// 64-ray batch of each other instead of within a 512-ray chunk (~0.15 ms per launch: nothing for a 265 M-ray launch, 3 - 4 % of the 25 M-ray
// launches of small sub-passes).  Sizing the request from a fresh read of the cursor was tried first and lost badly (trace +67 %: the extra
// load of the contended line); this form costs no memory operation beyond the one atomic.
struct ChunkPlan { unsigned c, big, k1, mid_end, k2; };
GX_DEV ChunkPlan chunk_plan(unsigned total, unsigned c) {
    const unsigned waves = gridDim.x * (blockDim.x / 64u);
    ChunkPlan p;
    p.c = c;
    const unsigned tail = waves * 384u;
    p.big = (c > 128u && total > tail) ? (total - tail) / c * c : 0u;
    p.k1 = p.big / c;
    const unsigned mid = min(total - p.big, waves * 256u) / 128u * 128u;
    p.mid_end = p.big + mid;
    p.k2 = mid / 128u;
    return p;
}
GX_DEV bool chunk_range(const ChunkPlan &p, unsigned v, unsigned total, unsigned *base, unsigned *count) {
    unsigned b, n;
    if (v < p.k1) { b = v * p.c; n = p.c; }
    else if (v < p.k1 + p.k2) { b = p.big + (v - p.k1) * 128u; n = 128u; }
    else {
        const unsigned long long bb = (unsigned long long)p.mid_end + (unsigned long long)(v - p.k1 - p.k2) * 64ull;
        if (bb >= total) return false;
        b = (unsigned)bb; n = 64u;
    }
    if (b >= total) return false;
    *base = b; *count = min(n, total - b);
    return true;
}

constexpr int kTraceChunk = 512;   // most rays a wave takes per global atomic (the host shrinks the chunk for thin launches so that every wave gets one)
#ifndef GX_TRACE_LEAVE_MUL
#define GX_TRACE_LEAVE_MUL 1
#endif
#ifndef GX_TRACE_LEAVE_DIV
#define GX_TRACE_LEAVE_DIV 2
#endif
#ifndef GX_REFILL_MIN
#define GX_REFILL_MIN 1
#endif
#ifndef GX_SPECULATE
#define GX_SPECULATE 1
#endif
constexpr bool kSpeculate = GX_SPECULATE != 0;
#ifndef GX_NEE_INTERLEAVED
#define GX_NEE_INTERLEAVED 0
#endif
constexpr int kRefillMin = GX_REFILL_MIN;   // refill only when at least this many lanes are idle (or none is live)
constexpr int kTraceLeaveMul = GX_TRACE_LEAVE_MUL, kTraceLeaveDiv = GX_TRACE_LEAVE_DIV;   // leave phase A when searching <= live * MUL / DIV

// Traversal stack of one lane: levels [0, K) in LDS, deeper levels in global memory (coalesced across the wave).
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(1))) int global_int;
struct LaneStack {
    lds_int *lds;        // explicit address spaces: keeps the two paths as ds_* / global_* instructions (no flat pointer select)
    global_int *spill;
    int K;
    int stride;   // lanes in the grid (< 2^21), so (n - K) * stride fits 32 bits
    GX_DEV void push(int &n, int v) const {
        if (n < K) lds[n * kBlock] = v;
        else spill[(n - K) * stride] = v;
        ++n;
    }
    GX_DEV int pop(int &n) const {
        --n;
        return n < K ? lds[n * kBlock] : spill[(n - K) * stride];
    }
};

// Per-ray constants of the 4-wide walk: which float4 of a DNode4 holds the near / far plane of each axis for this ray's
// direction signs (the reference selects them per box with dirIsNeg, Geometry.h:1384-1400 -- here the selection is an
// address), and the shift that picks this octant's byte of the node's visiting-order table.
struct RayOctant {
    int nx, fx, ny, fy, nz, fz;   // float4 indices into the node: lox 0 loy 1 loz 2 hix 3 hiy 4 hiz 5
    int shift;                    // 8 * (neg0 | neg1 << 1 | neg2 << 2)
};
GX_DEV RayOctant ray_octant(int neg0, int neg1, int neg2) {
    RayOctant r;
    r.nx = neg0 ? 3 : 0; r.fx = 3 - r.nx;
    r.ny = neg1 ? 4 : 1; r.fy = 5 - r.ny;
    r.nz = neg2 ? 5 : 2; r.fz = 7 - r.nz;
    r.shift = 8 * (neg0 | (neg1 << 1) | (neg2 << 2));
    return r;
}

// 4-wide step: test the four children of `node` (Bounds3::IntersectP, Geometry.h:1380-1406, same operations in the same
// order for each box), return the first one hit in the reference's visiting order and push the others, farthest first.
// Straight-line code: absent children carry inverted boxes that fail the test, the order comes from the node's table.
GX_DEV int bvh4_step(const float4 *__restrict__ n4, int node, V3 ro, V3 invDir, const RayOctant &oc, float tMaxRay, const LaneStack &stack, int &toVisit) {
    const float4 *q = n4 + 8 * (size_t)node;
    const float4 nX = q[oc.nx], fX = q[oc.fx], nY = q[oc.ny], fY = q[oc.fy], nZ = q[oc.nz], fZ = q[oc.fz];
    const float4 cf = q[6], mf = q[7];
    const float k = 1 + 2 * GX_GAMMA(3);
    unsigned hitMask = 0;
#define GX_SLAB(C, BIT)                                                                   \
    {                                                                                     \
        float tMin = (nX.C - ro.x) * invDir.x, tMax = (fX.C - ro.x) * invDir.x;           \
        float tyMin = (nY.C - ro.y) * invDir.y, tyMax = (fY.C - ro.y) * invDir.y;         \
        tMax *= k; tyMax *= k;                                                            \
        bool ok = !(tMin > tyMax || tyMin > tMax);                                        \
        if (tyMin > tMin) tMin = tyMin;                                                   \
        if (tyMax < tMax) tMax = tyMax;                                                   \
        float tzMin = (nZ.C - ro.z) * invDir.z, tzMax = (fZ.C - ro.z) * invDir.z;         \
        tzMax *= k;                                                                       \
        ok = ok && !(tMin > tzMax || tzMin > tMax);                                       \
        if (tzMin > tMin) tMin = tzMin;                                                   \
        if (tzMax < tMax) tMax = tzMax;                                                   \
        ok = ok && (tMin < tMaxRay) && (tMax > 0);                                        \
        hitMask |= ok ? (BIT) : 0u;                                                       \
    }
    GX_SLAB(x, 1u) GX_SLAB(y, 2u) GX_SLAB(z, 4u) GX_SLAB(w, 8u)
#undef GX_SLAB
    if (hitMask == 0) return (toVisit == 0) ? kRefDone : stack.pop(toVisit);
    const unsigned long long table = ((unsigned long long)__float_as_uint(mf.y) << 32) | __float_as_uint(mf.x);
    const unsigned ord = (unsigned)(table >> oc.shift);
    const int c0 = __float_as_int(cf.x), c1 = __float_as_int(cf.y), c2 = __float_as_int(cf.z), c3 = __float_as_int(cf.w);
    int next = kRefDone;
    bool have = false;
#pragma unroll
    for (int kk = 3; kk >= 0; --kk) {
        const unsigned slot = (ord >> (2 * kk)) & 3u;
        if ((hitMask >> slot) & 1u) {
            if (have) stack.push(toVisit, next);
            next = slot == 0 ? c0 : (slot == 1 ? c1 : (slot == 2 ? c2 : c3));
            have = true;
        }
    }
    return next;
}

#ifdef GX_TRACE_STATS
static __device__ unsigned long long g_trace_stats[24];
#define GX_STAT(i, v) do { if (lane == 0) st_[i] += (unsigned long long)(v); } while (0)
#else
#define GX_STAT(i, v) do {} while (0)
#endif

// COUNT: count nodes / triangles (profiling).  WIDE: traverse the collapsed 4-wide tree (sc.nodes4) instead of the
// reference's binary nodes; the counting runs use WIDE = false so that the counts are those of the reference traversal.
// SPH: the scene has spheres (a separate instantiation keeps their registers out of the triangle-only kernel).
template <bool COUNT, bool WIDE, bool SPH>
__global__ void __launch_bounds__(kBlock) k_trace(DScene sc, PathArrays pa, TraceWork w, unsigned int *cursor, Counters *ctr, int lds_entries, int *spill, int chunk) {
    extern __shared__ int stack_mem[];   // lds_entries * kBlock ints
    LaneStack stack;
    stack.lds = (lds_int *)&stack_mem[threadIdx.x];
    stack.K = lds_entries;
    stack.stride = (int)gridDim.x * kBlock;
    stack.spill = (global_int *)(spill + (size_t)blockIdx.x * kBlock + threadIdx.x);
    const int lane = __lane_id();
    trace_work_counts(w);
    const unsigned total = (unsigned)w.n_closest + 2u * (unsigned)w.n_nee;
    chunk = trace_chunk(total, chunk);
    const float4 *__restrict__ nodes = sc.nodes;
    const DTri *__restrict__ tris = sc.tris;

    unsigned poolBase = 0, poolCount = 0;   // wave-uniform
    bool exhausted = false;                 // wave-uniform: the global cursor ran past `total`

    // per-lane ray state
    int item = -1, kind = 0, path = -1;
    V3 ro, rd, invDir;
    RayShear shear;
    RayOctant oct = ray_octant(0, 0, 0);
    float tMax = 0;
    int neg0 = 0, neg1 = 0, neg2 = 0;
    int cur = -1, toVisit = 0, leafOff = 0, leafN = 0, hitLeaf = -1, expect = -1;
    uint32_t cntNodes = 0, cntTris = 0, cntRetests = 0;
#ifdef GX_TRACE_STATS
    unsigned long long st_[24] = {0};
#endif

    while (true) {
        GX_STAT(0, 1);
        // ---------------- refill idle lanes from the wave pool ----------------
        bool need = item < 0;
        unsigned long long needMask = __ballot(need);
        if (kRefillMin > 1 && __popcll(needMask) < kRefillMin && needMask != ~0ull) needMask = 0;   // batch the refills
        if (needMask) {
            GX_STAT(7, 1);
            GX_STAT(8, __popcll(needMask));
            if (poolCount == 0 && !exhausted) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(cursor, (unsigned)chunk);
                base = __shfl(base, 0);
                if (base >= total) exhausted = true;
                else { poolBase = base; poolCount = min((unsigned)chunk, total - base); }
            }
            if (poolCount > 0) {
                unsigned rank = (unsigned)__popcll(needMask & ((1ull << lane) - 1ull));
                unsigned take = min(poolCount, (unsigned)__popcll(needMask));
                if (need && rank < take) {
                    unsigned i = poolBase + rank;
                    if (w.order) i = w.order[i];
                    item = (int)i;
                    float4 o4, d4;
                    if (i < (unsigned)w.n_closest) {
                        kind = 0;
                        path = w.q_closest ? w.q_closest[i] : (int)i;
                        o4 = pa.ray_o[(size_t)path * kRS]; d4 = pa.ray_d[(size_t)path * kRS];
                        tMax = o4.w;
                    } else {
                        // NEE work items: first all shadow rays, then all MIS rays (a wave then holds rays of one kind that
                        // start from neighbouring vertices and -- the shadow rays -- all head for the same few lights)
                        unsigned e = i - (unsigned)w.n_closest;
                        const bool isShadow = GX_NEE_INTERLEAVED ? (e & 1u) == 0 : e < (unsigned)w.n_nee;
                        path = w.q_nee[GX_NEE_INTERLEAVED ? (e >> 1) : (isShadow ? e : e - (unsigned)w.n_nee)];
                        int nflags = __float_as_int(pa.sh_d[(size_t)path * kRS].w);
                        if (isShadow) {
                            kind = 1;
                            if (nflags & 1) { o4 = pa.sh_o[(size_t)path * kRS]; d4 = pa.sh_d[(size_t)path * kRS]; tMax = o4.w; }
                            else item = -1;     // this vertex spawned no shadow ray
                        } else {
                            kind = 2;
                            if (nflags & 2) { o4 = pa.mis_o[(size_t)path * kRS]; d4 = pa.mis_d[(size_t)path * kRS]; expect = __float_as_int(o4.w); tMax = GX_INF; }
                            else item = -1;
                        }
                    }
                    if (item >= 0) {
                        ro = V3(o4.x, o4.y, o4.z); rd = V3(d4.x, d4.y, d4.z);
                        invDir = V3(1.f / rd.x, 1.f / rd.y, 1.f / rd.z);
                        shear = ray_shear(rd);
                        neg0 = invDir.x < 0; neg1 = invDir.y < 0; neg2 = invDir.z < 0;
                        if (WIDE) oct = ray_octant(neg0, neg1, neg2);
                        cur = WIDE ? sc.root4 : 0; toVisit = 0; leafN = 0; hitLeaf = -1;
                        // spheres live outside the BVH and are tested first (Scene::Intersect of this build, oracle/o_scene.h)
                        for (int si = 0; SPH && si < sc.n_spheres; ++si) {
                            float tH;
                            if (sphere_test(sc.spheres[si], ro, rd, tMax, &tH)) {
                                hitLeaf = -2 - si;
                                if (kind == 1) { cur = -1; break; }
                                tMax = tH;
                            }
                        }
                    }
                }
                poolBase += take; poolCount -= take;
            }
        }
        if (__ballot(item >= 0) == 0) {
            if (exhausted) break;
            continue;   // pool was empty this round: fetch a chunk on the next iteration
        }

        // ---------------- phase A: interior traversal until every active lane holds a leaf ----------------
        // Leave the walk as soon as at most a quarter of the live lanes is still searching: waiting for the last
        // straggler left ~3/4 of the wave idle (the stragglers simply resume in the next round).
        const int nLive = __popcll(__ballot(item >= 0));
        GX_STAT(9, nLive);
        while (true) {
            if (WIDE && item >= 0 && leafN == 0 && cur < -1) {   // the next reference is a leaf: stage it, pre-pop its successor
                int lr = ~cur;
                leafOff = lr & 0xffffff; leafN = (lr >> 24) & 0x7f;
                cur = (toVisit == 0) ? -1 : stack.pop(toVisit);
            }
            // A lane that already holds a leaf keeps walking (speculatively, with the tMax it has) until it reaches a second
            // leaf: every node it visits is one the reference visits or a superset of them (tMax only shrinks), and each
            // leaf is re-tested against the current tMax before its triangles are (phase B), so results do not change.
            bool searching = item >= 0 && cur >= 0 && (kSpeculate && WIDE ? true : leafN == 0);
            bool hungry = searching && leafN == 0;
            int nSearching = __popcll(__ballot(hungry));
            if (nSearching * kTraceLeaveDiv <= nLive * kTraceLeaveMul && (nSearching == 0 || nSearching < nLive)) break;
            GX_STAT(1, 1);
            GX_STAT(2, __popcll(__ballot(searching)));
            if (searching) {
                if (COUNT) cntNodes++;
                if (WIDE) {
                    int next = bvh4_step(sc.nodes4, cur, ro, invDir, oct, tMax, stack, toVisit);
                    cur = (next == kRefDone) ? -1 : next;   // interior index, leaf reference (< -1) or done
                } else {
                    float4 n0 = nodes[2 * cur], n1 = nodes[2 * cur + 1];
                    int neg[3] = {neg0, neg1, neg2};
                    bool hitBox = slab_test(n0, n1, ro, invDir, neg, tMax);
                    int offset = __float_as_int(n1.z);
                    uint32_t meta = __float_as_uint(n1.w);
                    int nPrims = (int)(meta & 0xffffu);
                    int next;
                    if (hitBox && nPrims == 0) {
                        int axis = (int)(meta >> 16);
                        int ng = axis == 0 ? neg0 : (axis == 1 ? neg1 : neg2);
                        int nearC = ng ? offset : cur + 1, farC = ng ? cur + 1 : offset;
                        stack.push(toVisit, farC);
                        next = nearC;
                    } else {
                        if (hitBox) { leafOff = offset; leafN = nPrims; }
                        next = (toVisit == 0) ? -1 : stack.pop(toVisit);
                    }
                    cur = next;
                }
            }
        }
        // ---------------- phase B: triangle tests ----------------
#ifdef GX_TRACE_STATS
        {
            unsigned long long lm = __ballot(item >= 0 && leafN > 0);
            if (lm) {
                GX_STAT(3, 1);
                GX_STAT(4, __popcll(lm));
                int mx = leafN;
                for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
                int sm = (item >= 0) ? leafN : 0;
                for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
                GX_STAT(5, mx);
                GX_STAT(6, sm);
                GX_STAT(10, __popcll(__ballot(item >= 0 && leafN > 0 && hitLeaf >= 0)));
            }
        }
#endif
        if (item >= 0 && leafN > 0) {
            bool visit = true;
            if (WIDE && (SPH ? hitLeaf != -1 : hitLeaf >= 0)) {   // tMax only ever shrinks after a hit: before the first hit the earlier test stands
                // BVHAccel::Intersect tests a leaf's box when it pops it, i.e. against the CURRENT ray.tMax; the 4-wide
                // step tested it earlier with an older tMax.  Re-test here so that exact ties (t == tMax on flat,
                // axis-aligned boxes such as the Cornell walls) resolve as in the reference.  The leaf's bounds (the floats
                // of its LinearBVHNode) sit in a table addressed by its first triangle: two dwordx4 instead of re-reading
                // and re-bounding the triangles.
                const float4 b0 = sc.leaf_box[2 * (size_t)leafOff], b1 = sc.leaf_box[2 * (size_t)leafOff + 1];
                int neg[3] = {neg0, neg1, neg2};
                if (COUNT) cntRetests++;
                visit = slab_test(b0, b1, ro, invDir, neg, tMax);
            }
            for (int i = 0; visit && i < leafN; ++i) {
                V3 p0, p1, p2;
                load_tri(tris, leafOff + i, &p0, &p1, &p2);
                if (COUNT) cntTris++;
                TriHit h;
                if (tri_test_sheared(p0, p1, p2, ro, shear, tMax, &h)) {
                    hitLeaf = leafOff + i;
                    if (kind == 1) { cur = -1; break; }   // IntersectP returns at the first hit
                    tMax = h.t;                            // GeometricPrimitive::Intersect shrinks ray.tMax
                }
            }
            leafN = 0;
        }
        // ---------------- phase C: retire finished rays ----------------
        if (item >= 0 && cur == -1 && leafN == 0) {
            if (kind == 0) {
                pa.hit[path] = hitLeaf;
                if (hitLeaf < 0) {   // triangle hits: the binning pass looks the class up from `hit` (k_compact_count<COMPACT_HITCLASS>)
                    int cls = sc.escape_class;   // misses that still have to collect an infinite light: a queue of their own, or the code of class 0
                    if (SPH && hitLeaf != -1) { const int mat = sc.spheres[-2 - hitLeaf].material; if (mat >= 0) cls = sc.materials[mat].shade_class; }
                    else if (sc.lt.n_infinite == 0) {
                        // a ray that escapes a scene without infinite lights adds nothing and ends its path (PathIntegrator.cpp:101-113):
                        // no shading class (4 is binned nowhere), so it does not take a lane in a k_shade wave
                        cls = 4;
                        pa.pflags[path] = 0;
                    }
                    pa.pclass[path] = (unsigned char)cls;
                }
            }
            else if (kind == 1) {
                if (w.vis) w.vis[4 * (size_t)path] = hitLeaf == -1 ? 1 : 0;
                else reinterpret_cast<float *>(&pa.sh_o[(size_t)path * kRS])[3] = hitLeaf == -1 ? 1.f : 0.f;
            } else {
                bool ok = (expect >= 0) ? (hitLeaf == expect) : (hitLeaf == -1);
                if (w.vis) w.vis[4 * (size_t)path + 1] = ok ? 1 : 0;
                else reinterpret_cast<float *>(&pa.mis_o[(size_t)path * kRS])[3] = ok ? 1.f : 0.f;
            }
            item = -1;
        }
    }
#ifdef GX_TRACE_STATS
    if (lane == 0) for (int i = 0; i < 24; ++i) if (st_[i]) atomicAdd(&g_trace_stats[i], st_[i]);
#endif
    if (COUNT) {
        atomicAdd(&ctr->nodes, (unsigned long long)cntNodes);
        atomicAdd(&ctr->tris, (unsigned long long)cntTris);
        atomicAdd(&ctr->retests, (unsigned long long)cntRetests);
    }
}

// L += beta * (EstimateDirect(...) / lightPdf), core/Integrator.cpp:78 + PathIntegrator.cpp:135-141, once the
// two visibility results of the vertex are known.  Pure streaming: per NEE vertex one word of flags + results (TraceWork::vis), X, Y when
// the vertex has a MIS ray, beta and L (72 B read, 16 B written; the first version read the three float4 that carried flags and results
// in their w lanes: 128 B).
static __global__ void __launch_bounds__(kBlock) k_nee_combine(PathArrays pa, const int *__restrict__ queue, int n, const unsigned char *__restrict__ vis, const unsigned *n_dev = nullptr) {
    if (n_dev) n = (int)*n_dev;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int path = queue[i];
        const unsigned v = reinterpret_cast<const unsigned *>(vis)[path];
        const int flags = (int)((v >> 16) & 0xffu);
        // neither ray arrived (the light sample is occluded and the BSDF sample missed the light): Ld = 0, and L + beta * (0 / pdf) == L bit
        // for bit when beta is finite (k_shade says so in bit 7 of the flags; pdf > 0) -- a third to a half of the vertices; their three
        // records and L are not touched
        if (!(flags & 0x80) && !(((flags & 1) && (v & 0xffu)) || ((flags & 2) && ((v >> 8) & 0xffu)))) continue;
        float4 X4 = pa.sh_X[(size_t)path * kRS];
        Spec Ld(0.f);
        if ((flags & 1) && (v & 0xffu)) Ld = Ld + Spec(X4.x, X4.y, X4.z);
        if ((flags & 2) && ((v >> 8) & 0xffu)) {
            float4 Y4 = pa.mis_Y[path];
            Spec Y(Y4.x, Y4.y, Y4.z);
            if (!Y.is_black()) Ld = Ld + Y;
        }
        float4 nb = pa.nbeta[(size_t)path * kRS], L4 = pa.L[path];
        Spec add = Spec(nb.x, nb.y, nb.z) * (Ld / X4.w);
        pa.L[path] = make_float4(L4.x + add.r, L4.y + add.g, L4.z + add.b, 0.f);
    }
}

}  // namespace gnxr

// trace4d_kernel.hip.h -- k_trace4d: k_trace4 with TWO rays per lane.
//
// k_trace4 (trace4_kernel.hip.h) issues at the VALU ceiling of its instruction mix (SQ busy 0.96) with 46 % of the lanes on: in the node
// phase the lanes that already hold a leaf (or have finished) idle, in the triangle phase the lanes that are still walking do (69 % / 29 %
// of the lanes on, profiles/README.md).  Nothing about ONE ray fixes that -- whether its next step is a node or a triangle is the ray's own
// business.  Here every lane owns two rays, each with its own state registers and its own LDS stack column, and in every step of a phase a
// lane works on whichever of its two rays wants that kind of step (v_cndmask selects of the dozen operands the step reads: ~15 extra
// two-cycle instructions on a ~100-instruction step).  A lane idles in a phase only when NEITHER ray wants it.
//
// Per ray, the walk is k_trace4's, operation for operation: same visiting order (the per-octant order table), same slab arithmetic, same
// speculative second leaf, same re-test of a leaf against the current tMax, same triangle test -- so the result of every ray (hit triangle,
// occlusion, MIS expectation) is the same, and images and ray counts stay bit-identical.  What changes is only which lanes are on.
// The price: 2 x 19 state registers (the kernel is built for 4 waves per SIMD instead of 5), two stack columns per lane (LDS: the node
// cache shrinks to 32 nodes to make room).
#pragma once
#include "trace4_kernel.hip.h"

namespace gnxr {

#ifndef GX_T4D_WAVES
#define GX_T4D_WAVES 4
#endif
#ifndef GX_T4D_CACHE
#define GX_T4D_CACHE 32
#endif
// phase policy (tuning knobs): leave the node phase when the lanes with a ray that looks for its first leaf are <= MUL / DIV of the live
// lanes; run a second triangle trip right away when at least GX_T4D_B2 lanes still hold a staged leaf (0: never)
#ifndef GX_T4D_LEAVE_MUL
#define GX_T4D_LEAVE_MUL 1
#endif
#ifndef GX_T4D_LEAVE_DIV
#define GX_T4D_LEAVE_DIV 2
#endif
#ifndef GX_T4D_B2
#define GX_T4D_B2 0
#endif
constexpr int kTopCacheD = GX_T4D_CACHE;   // DNode4[0 .. kTopCacheD) are served from LDS

struct Ray4 {            // the registers of one ray of a lane
    float ox, oy, oz, ix, iy, iz, Sx, Sy, tMax;
    int pk, path;        // pk as in k_trace4; path < 0: the slot is empty
    int cur, toVisit, leaf, hitLeaf;   // leaf: staged leaf reference (offset | count << 24), 0 = none
    unsigned oNX, oNY, oNZ, ordShift;
};

template <bool SPH, bool SPILL>
__global__ void __launch_bounds__(kBlock, GX_T4D_WAVES) k_trace4d(DScene sc, PathArrays pa, TraceWork w, unsigned int *cursor, Counters *ctr, int lds_entries, int spill_levels,
                                                                 int *spill, int chunk, int n_top) {
    // LDS: [2 * lds_entries * kBlock] stack columns (ray 0 | ray 1) | [(11 | 12) * kRqStride] ray records | [8 * kTopCacheD float4] node cache | [128 B] order table
    extern __shared__ int smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    lds_int *const stk0 = (lds_int *)&smem[threadIdx.x];
    lds_int *const stk1 = (lds_int *)&smem[lds_entries * kBlock + threadIdx.x];
    global_int *const spl0 = (global_int *)(spill + (size_t)blockIdx.x * kBlock + threadIdx.x);
    const int spillStride = (int)gridDim.x * kBlock;
    global_int *const spl1 = spl0 + (size_t)spill_levels * spillStride;
    const int lane = __lane_id();
    lds_int *const rq = (lds_int *)&smem[2 * lds_entries * kBlock + (threadIdx.x >> 6) * kRayQueue];
    trace_work_counts(w);
    const unsigned total = (unsigned)w.n_closest + 2u * (unsigned)w.n_nee;
    chunk = trace_chunk(total, chunk);
    const ChunkPlan plan = chunk_plan(total, (unsigned)chunk);
    const unsigned nWaves = gridDim.x * (kBlock / 64u), waveId = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6);
    bool firstFetch = true;   // wave-uniform
    const DTri *__restrict__ tris = sc.tris;
    const char *__restrict__ nb = reinterpret_cast<const char *>(sc.nodes4);
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) f4v lds_f4;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    lds_f4 *const topN = (lds_f4 *)&smem[2 * lds_entries * kBlock + (kRayRecDwords + (SPH ? 1 : 0)) * kRqStride];
    lds_u8 *const lut = (lds_u8 *)(topN + 8 * kTopCacheD);
    {
        const f4v *gn = reinterpret_cast<const f4v *>(sc.nodes4);
        for (int i = threadIdx.x; i < n_top * 8; i += kBlock) topN[(i & 7) * kTopCacheD + (i >> 3)] = gn[i];
        if (threadIdx.x < 128) lut[threadIdx.x] = (unsigned char)order_entry(threadIdx.x >> 4, threadIdx.x & 15u);
        __syncthreads();
    }
    // stack of ray `one` (false: ray 0): entry n of the lane's column
    auto push = [&](bool one, int &n, int v) {
        lds_int *const st = one ? stk1 : stk0;
        if (!SPILL || n < lds_entries) st[n * kBlock] = v;
        else (one ? spl1 : spl0)[(n - lds_entries) * spillStride] = v;
        ++n;
    };
    auto pop = [&](bool one, int &n) -> int {
        --n;
        lds_int *const st = one ? stk1 : stk0;
        return (!SPILL || n < lds_entries) ? st[n * kBlock] : (one ? spl1 : spl0)[(n - lds_entries) * spillStride];
    };

    unsigned poolBase = 0, poolCount = 0;
    bool exhausted = false;
    unsigned rqHead = 0, rqCount = 0;
    Ray4 R0, R1;
    R0.path = R1.path = -1;
    R0.cur = R1.cur = -1; R0.toVisit = R1.toVisit = 0; R0.leaf = R1.leaf = 0; R0.hitLeaf = R1.hitLeaf = -1; R0.pk = R1.pk = 0;
    R0.ox = R0.oy = R0.oz = R0.ix = R0.iy = R0.iz = R0.Sx = R0.Sy = R0.tMax = 0.f; R1.ox = R1.oy = R1.oz = R1.ix = R1.iy = R1.iz = R1.Sx = R1.Sy = R1.tMax = 0.f;
    R0.oNX = R1.oNX = 0; R0.oNY = R1.oNY = 16; R0.oNZ = R1.oNZ = 32; R0.ordShift = R1.ordShift = 0;

    auto take_ray = [&](Ray4 &r, unsigned slot) {
        const lds_float *q = (const lds_float *)(rq + slot);
        r.ox = q[0 * kRqStride]; r.oy = q[1 * kRqStride]; r.oz = q[2 * kRqStride]; r.tMax = q[3 * kRqStride];
        r.ix = q[4 * kRqStride]; r.iy = q[5 * kRqStride]; r.iz = q[6 * kRqStride]; r.Sx = q[7 * kRqStride];
        r.Sy = q[8 * kRqStride];
        r.pk = __float_as_int(q[9 * kRqStride]);
        r.path = __float_as_int(q[10 * kRqStride]);
        r.hitLeaf = SPH ? rq[slot + 11 * kRqStride] : -1;
        const int neg0 = r.ix < 0, neg1 = r.iy < 0, neg2 = r.iz < 0;
        r.oNX = neg0 ? 48u : 0u; r.oNY = neg1 ? 64u : 16u; r.oNZ = neg2 ? 80u : 32u;
        r.ordShift = 8u * (unsigned)(neg0 | (neg1 << 1) | (neg2 << 2));
        r.cur = ((r.pk >> 4) & 1) ? -1 : sc.root4; r.toVisit = 0; r.leaf = 0;
    };
    auto retire_ray = [&](Ray4 &r) {
        const int kind = r.pk & 3, path = r.path, hitLeaf = r.hitLeaf;
        if (kind == 0) {
            pa.hit[path] = hitLeaf;
            if (hitLeaf < 0) {
                int cls = sc.escape_class;
                if (SPH && hitLeaf != -1) { const int mat = sc.spheres[-2 - hitLeaf].material; if (mat >= 0) cls = sc.materials[mat].shade_class; }
                else if (sc.lt.n_infinite == 0) { cls = 4; pa.pflags[path] = 0; }
                pa.pclass[path] = (unsigned char)cls;
            }
        } else if (kind == 1) {
            if (w.vis) w.vis[4 * (size_t)path] = hitLeaf == -1 ? 1 : 0;
            else reinterpret_cast<float *>(&pa.sh_o[(size_t)path * kRS])[3] = hitLeaf == -1 ? 1.f : 0.f;
        } else {
            const int expect = __float_as_int(pa.mis_o[(size_t)path * kRS].w);
            const bool ok = (expect >= 0) ? (hitLeaf == expect) : (hitLeaf == -1);
            if (w.vis) w.vis[4 * (size_t)path + 1] = ok ? 1 : 0;
            else reinterpret_cast<float *>(&pa.mis_o[(size_t)path * kRS])[3] = ok ? 1.f : 0.f;
        }
        r.path = -1;
    };

    while (true) {
        // ---------------- refill: both ray slots of every lane ----------------
        const unsigned long long need0 = __ballot(R0.path < 0), need1 = __ballot(R1.path < 0);
        if (need0 | need1) {
            if (rqCount == 0) {
                if (poolCount == 0 && !exhausted) {
                    // the cursor counts chunks (chunk_plan / chunk_range, trace_kernel.hip.h); a wave's FIRST chunk is its own number -- no
                    // atomic: 5120 waves asking at once at the start of a launch queue up behind one address for ~60 us
                    unsigned v = waveId;
                    if (!firstFetch) {
                        if (lane == 0) v = nWaves + atomicAdd(cursor, 1u);
                        v = __shfl(v, 0);
                    }
                    firstFetch = false;
                    if (!chunk_range(plan, v, total, &poolBase, &poolCount)) { exhausted = true; poolCount = 0; }
                }
                if (poolCount > 0) {
                    // ---- batch set-up, as in k_trace4
                    const unsigned take = min(poolCount, (unsigned)kRayQueue);
                    bool valid = false;
                    float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
                    int sphHit = -1;
                    if ((unsigned)lane < take) {
                        unsigned i = poolBase + (unsigned)lane;
                        if (w.order) i = w.order[i];
                        float4 o4, d4;
                        int kind_ = 0, path_, any_ = 0;
                        float tMax_;
                        valid = true;
                        if (i < (unsigned)w.n_closest) {
                            path_ = w.q_closest ? w.q_closest[i] : (int)i;
                            o4 = pa.ray_o[(size_t)path_ * kRS]; d4 = pa.ray_d[(size_t)path_ * kRS];
                            tMax_ = o4.w;
                        } else {
                            const unsigned e = i - (unsigned)w.n_closest;
                            const bool isShadow = e < (unsigned)w.n_nee;
                            path_ = w.q_nee[isShadow ? e : e - (unsigned)w.n_nee];
                            const bool together = w.vis != nullptr;
                            if (isShadow) {
                                kind_ = 1; any_ = 1;
                                o4 = pa.sh_o[(size_t)path_ * kRS]; d4 = pa.sh_d[(size_t)path_ * kRS]; tMax_ = o4.w;
                                valid = (__float_as_int(d4.w) & 1) != 0;
                            } else {
                                kind_ = 2;
                                const int nflags = __float_as_int(pa.sh_d[(size_t)path_ * kRS].w);
                                valid = (nflags & 2) != 0;
                                if (together || valid) { o4 = pa.mis_o[(size_t)path_ * kRS]; d4 = pa.mis_d[(size_t)path_ * kRS]; }
                                else { o4 = make_float4(0, 0, 0, 0); d4 = o4; }
                                tMax_ = GX_INF;
                                any_ = __float_as_int(o4.w) < 0 ? 1 : 0;
                            }
                        }
                        if (valid) {
                            const V3 o(o4.x, o4.y, o4.z), d(d4.x, d4.y, d4.z);
                            const V3 iv(1.f / d.x, 1.f / d.y, 1.f / d.z);
                            const RayShear sh = ray_shear(d);
                            int done = 0;
                            for (int si = 0; SPH && si < sc.n_spheres; ++si) {
                                float tH;
                                if (sphere_test(sc.spheres[si], o, d, tMax_, &tH)) {
                                    sphHit = -2 - si;
                                    if (any_) { done = 1; break; }
                                    tMax_ = tH;
                                }
                            }
                            const int ex = (__builtin_isinf(iv.x) || __builtin_isinf(iv.y) || __builtin_isinf(iv.z)) ? 1 : 0;
                            r0 = make_float4(o.x, o.y, o.z, tMax_);
                            r1 = make_float4(iv.x, iv.y, iv.z, sh.Sx);
                            r2 = make_float4(sh.Sy, __int_as_float(kind_ | (sh.kz << 2) | (done << 4) | (ex << 5) | (any_ << 6)), __int_as_float(path_), 0.f);
                        }
                    }
                    const unsigned long long vm = __ballot(valid);
                    if (valid) {
                        const int slot = __popcll(vm & ((1ull << lane) - 1ull));
                        lds_float *q = (lds_float *)(rq + slot);
                        q[0 * kRqStride] = r0.x; q[1 * kRqStride] = r0.y; q[2 * kRqStride] = r0.z; q[3 * kRqStride] = r0.w;
                        q[4 * kRqStride] = r1.x; q[5 * kRqStride] = r1.y; q[6 * kRqStride] = r1.z; q[7 * kRqStride] = r1.w;
                        q[8 * kRqStride] = r2.x; q[9 * kRqStride] = r2.y; q[10 * kRqStride] = r2.z;
                        if (SPH) rq[slot + 11 * kRqStride] = sphHit;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    rqCount = (unsigned)__popcll(vm); rqHead = 0;
                    poolBase += take; poolCount -= take;
                }
            }
            // hand the waiting records out: first to the empty slots 0, then to the empty slots 1
            if (rqCount > 0 && need0) {
                const unsigned rank = (unsigned)__popcll(need0 & ((1ull << lane) - 1ull));
                if (R0.path < 0 && rank < rqCount) take_ray(R0, rqHead + rank);
                const unsigned t = min(rqCount, (unsigned)__popcll(need0));
                rqHead += t; rqCount -= t;
            }
            if (rqCount > 0 && need1) {
                const unsigned rank = (unsigned)__popcll(need1 & ((1ull << lane) - 1ull));
                if (R1.path < 0 && rank < rqCount) take_ray(R1, rqHead + rank);
                const unsigned t = min(rqCount, (unsigned)__popcll(need1));
                rqHead += t; rqCount -= t;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");       // all reads of the queue precede the next batch's writes
        }
        const unsigned long long liveMask = __ballot(R0.path >= 0 || R1.path >= 0);
        if (liveMask == 0) {
            if (exhausted && rqCount == 0 && poolCount == 0) break;
            continue;
        }

        // ---------------- phase A: node steps on whichever ray of the lane wants one ----------------
        const int nLive = __popcll(liveMask);
        while (true) {
            // stage a leaf reference, pre-pop its successor (per ray)
            if (R0.path >= 0 && R0.leaf == 0 && R0.cur < -1) { R0.leaf = ~R0.cur; R0.cur = (R0.toVisit == 0) ? -1 : pop(false, R0.toVisit); }
            if (R1.path >= 0 && R1.leaf == 0 && R1.cur < -1) { R1.leaf = ~R1.cur; R1.cur = (R1.toVisit == 0) ? -1 : pop(true, R1.toVisit); }
            const bool s0 = R0.path >= 0 && R0.cur >= 0 && (kSpeculate || R0.leaf == 0), s1 = R1.path >= 0 && R1.cur >= 0 && (kSpeculate || R1.leaf == 0);
            const bool h0 = s0 && R0.leaf == 0, h1 = s1 && R1.leaf == 0;
            const int nHungry = __popcll(__ballot(h0 || h1));
            if (nHungry * GX_T4D_LEAVE_DIV <= nLive * GX_T4D_LEAVE_MUL && (nHungry == 0 || nHungry < nLive)) break;
            // a ray that looks for its first leaf goes before one that walks on speculatively
            const bool one = h0 ? false : (h1 ? true : !s0);
            if (s0 || s1) {
                const int cur = one ? R1.cur : R0.cur;
                const float rox = one ? R1.ox : R0.ox, roy = one ? R1.oy : R0.oy, roz = one ? R1.oz : R0.oz;
                const float ivx = one ? R1.ix : R0.ix, ivy = one ? R1.iy : R0.iy, ivz = one ? R1.iz : R0.iz;
                const float tMax = one ? R1.tMax : R0.tMax;
                const unsigned oNX = one ? R1.oNX : R0.oNX, oNY = one ? R1.oNY : R0.oNY, oNZ = one ? R1.oNZ : R0.oNZ, ordShift = one ? R1.ordShift : R0.ordShift;
                const int pk = one ? R1.pk : R0.pk;
                int toVisit = one ? R1.toVisit : R0.toVisit;
                f4v nX, fX, nY, fY, nZ, fZ, cf;
                uint2 tb;
                if (cur < n_top) {
                    const lds_f4 *L = topN + cur;
                    nX = L[(oNX >> 4) * kTopCacheD]; fX = L[((48u - oNX) >> 4) * kTopCacheD];
                    nY = L[(oNY >> 4) * kTopCacheD]; fY = L[((80u - oNY) >> 4) * kTopCacheD];
                    nZ = L[(oNZ >> 4) * kTopCacheD]; fZ = L[((112u - oNZ) >> 4) * kTopCacheD];
                    cf = L[6 * kTopCacheD];
                    const f4v t7 = L[7 * kTopCacheD];
                    tb = make_uint2(__float_as_uint(t7.x), __float_as_uint(t7.y));
                } else {
                    const unsigned off = (unsigned)cur << 7;
#define GX_LD4(o) (*reinterpret_cast<const f4v *>(nb + (unsigned)(off + (o))))
                    nX = GX_LD4(oNX); fX = GX_LD4(48u - oNX); nY = GX_LD4(oNY); fY = GX_LD4(80u - oNY); nZ = GX_LD4(oNZ); fZ = GX_LD4(112u - oNZ);
                    cf = GX_LD4(96u);
                    tb = *reinterpret_cast<const uint2 *>(nb + (unsigned)(off + 112u));
#undef GX_LD4
                }
                const float k = 1 + 2 * GX_GAMMA(3);
                unsigned hitMask = 0;
                if (!(pk & 32)) {
#define GX_SLAB3(C, BIT)                                                                                   \
    {                                                                                                      \
        const float e = fmaxf(fmaxf((nX.C - rox) * ivx, (nY.C - roy) * ivy), (nZ.C - roz) * ivz);           \
        const float x = fminf(fminf((fX.C - rox) * ivx, (fY.C - roy) * ivy), (fZ.C - roz) * ivz) * k;       \
        hitMask |= (e <= x && e < tMax && x > 0.f) ? (BIT) : 0u;                                            \
    }
                    GX_SLAB3(x, 1u) GX_SLAB3(y, 2u) GX_SLAB3(z, 4u) GX_SLAB3(w, 8u)
#undef GX_SLAB3
                } else {
#define GX_SLAB(C, BIT)                                                                   \
    {                                                                                     \
        float tMin = (nX.C - rox) * ivx, tMx = (fX.C - rox) * ivx;                        \
        float tyMin = (nY.C - roy) * ivy, tyMax = (fY.C - roy) * ivy;                     \
        tMx *= k; tyMax *= k;                                                             \
        bool ok = !(tMin > tyMax || tyMin > tMx);                                         \
        if (tyMin > tMin) tMin = tyMin;                                                   \
        if (tyMax < tMx) tMx = tyMax;                                                     \
        float tzMin = (nZ.C - roz) * ivz, tzMax = (fZ.C - roz) * ivz;                     \
        tzMax *= k;                                                                       \
        ok = ok && !(tMin > tzMax || tzMin > tMx);                                        \
        if (tzMin > tMin) tMin = tzMin;                                                   \
        if (tzMax < tMx) tMx = tzMax;                                                     \
        ok = ok && (tMin < tMax) && (tMx > 0);                                            \
        hitMask |= ok ? (BIT) : 0u;                                                       \
    }
                    GX_SLAB(x, 1u) GX_SLAB(y, 2u) GX_SLAB(z, 4u) GX_SLAB(w, 8u)
#undef GX_SLAB
                }
                int next;
                if (hitMask == 0) next = (toVisit == 0) ? -1 : pop(one, toVisit);
                else {
                    const unsigned word = (ordShift >= 32u ? tb.y : tb.x) >> (ordShift & 31u);
                    const unsigned e = lut[((word & 3u) << 5) | (word & 16u) | hitMask];
                    const int n = __popc(hitMask);
                    const int c0 = __float_as_int(cf.x), c1 = __float_as_int(cf.y), c2 = __float_as_int(cf.z), c3 = __float_as_int(cf.w);
                    auto child = [&](unsigned s) -> int { const int lo = (s & 1u) ? c1 : c0, hi = (s & 1u) ? c3 : c2; return (s & 2u) ? hi : lo; };
                    if (n >= 4) push(one, toVisit, child(e >> 6));
                    if (n >= 3) push(one, toVisit, child((e >> 4) & 3u));
                    if (n >= 2) push(one, toVisit, child((e >> 2) & 3u));
                    next = child(e & 3u);
                }
                if (one) { R1.cur = next; R1.toVisit = toVisit; } else { R0.cur = next; R0.toVisit = toVisit; }
            }
        }
        // ---------------- phase B: one staged leaf per lane, of whichever ray holds one ----------------
        for (int trip = 0; trip < (GX_T4D_B2 > 0 ? 2 : 1); ++trip) {
            if (trip == 1 && __popcll(__ballot((R0.path >= 0 && R0.leaf != 0) || (R1.path >= 0 && R1.leaf != 0))) < GX_T4D_B2) break;
            const bool l0 = R0.path >= 0 && R0.leaf != 0, l1 = R1.path >= 0 && R1.leaf != 0;
            if (l0 || l1) {
                const bool one = !l0;
                const float rox = one ? R1.ox : R0.ox, roy = one ? R1.oy : R0.oy, roz = one ? R1.oz : R0.oz;
                const float ivx = one ? R1.ix : R0.ix, ivy = one ? R1.iy : R0.iy, ivz = one ? R1.iz : R0.iz;
                const float Sx = one ? R1.Sx : R0.Sx, Sy = one ? R1.Sy : R0.Sy;
                float tMax = one ? R1.tMax : R0.tMax;
                const int pk = one ? R1.pk : R0.pk, lr = one ? R1.leaf : R0.leaf;
                int hitLeaf = one ? R1.hitLeaf : R0.hitLeaf;
                const int leafOff = lr & 0xffffff, leafN = (lr >> 24) & 0x7f;
                const V3 ro(rox, roy, roz), inv(ivx, ivy, ivz);
                bool visit = true, ended = false;
                const bool retest = SPH ? hitLeaf != -1 : hitLeaf >= 0;
                const bool fromVerts = retest && leafN == 1 && sc.leaf1_from_verts;
                int neg[3] = {ivx < 0, ivy < 0, ivz < 0};
                if (retest && !fromVerts) {
                    const float4 b0 = sc.leaf_box[2 * (size_t)leafOff], b1 = sc.leaf_box[2 * (size_t)leafOff + 1];
                    visit = slab_test(b0, b1, ro, inv, neg, tMax);
                }
                if (visit) {
                    RayShear shear;
                    const int kz = (pk >> 2) & 3;
                    shear.kz = kz; shear.kx = kz == 2 ? 0 : kz + 1; shear.ky = shear.kx == 2 ? 0 : shear.kx + 1;
                    shear.Sx = Sx; shear.Sy = Sy; shear.Sz = kz == 0 ? ivx : (kz == 1 ? ivy : ivz);
                    for (int i = 0; i < leafN; ++i) {
                        V3 p0, p1, p2;
                        load_tri(tris, leafOff + i, &p0, &p1, &p2);
                        if (fromVerts) {
                            const float4 b0 = make_float4(fminf(fminf(p0.x, p1.x), p2.x), fminf(fminf(p0.y, p1.y), p2.y), fminf(fminf(p0.z, p1.z), p2.z), fmaxf(fmaxf(p0.x, p1.x), p2.x));
                            const float4 b1 = make_float4(fmaxf(fmaxf(p0.y, p1.y), p2.y), fmaxf(fmaxf(p0.z, p1.z), p2.z), 0.f, 0.f);
                            if (!slab_test(b0, b1, ro, inv, neg, tMax)) break;
                        }
                        TriHit h;
                        if (tri_test_sheared(p0, p1, p2, ro, shear, tMax, &h)) {
                            hitLeaf = leafOff + i;
                            if (pk & 64) { ended = true; break; }
                            tMax = h.t;
                        }
                    }
                }
                if (one) { R1.tMax = tMax; R1.hitLeaf = hitLeaf; R1.leaf = 0; if (ended) R1.cur = -1; }
                else { R0.tMax = tMax; R0.hitLeaf = hitLeaf; R0.leaf = 0; if (ended) R0.cur = -1; }
            }
        }
        // ---------------- phase C: retire finished rays ----------------
        if (R0.path >= 0 && R0.cur == -1 && R0.leaf == 0) retire_ray(R0);
        if (R1.path >= 0 && R1.cur == -1 && R1.leaf == 0) retire_ray(R1);
    }
}

}  // namespace gnxr

// trace4_kernel.hip.h -- k_trace4: the 4-wide BVH traversal stage (BVHAccel::Intersect / IntersectP, accelerator/BVHAccel.cpp:653-729)
// for every ray a path vertex produces (continuation: closest hit; shadow: any hit; MIS: closest hit), second generation.
//
// Same contract and the same visiting order as k_trace<*, WIDE = true> (trace_kernel.hip.h), whose wave statistics
// (tests/dev_stats.py, -DGX_TRACE_STATS) showed where the VALU issue slots went: node steps 56 % (at 69 % of the lanes), triangle
// tests 24 % (29 %), per-ray set-up 17 % -- the set-up (six IEEE divisions: 1 / d, the shear of the watertight test) ran whenever ANY
// lane was idle, for ~11 of 64 lanes at a time.  What changed:
//   * set-up in batches of 64 through LDS: when a wave's ready queue is empty ALL its lanes set one ray up each (lanes that are in
//     the middle of a traversal included -- their state is untouched) and park the 11-dword records in LDS; an idle lane then just
//     pops a record.  The divisions run with every lane on, and the wave stalls on the three dependent loads of a ray once per 64
//     rays instead of once per refill;
//   * node step: for rays whose 1 / d is finite the three slab intervals of a child are merged with v_max3 / v_min3 and ONE
//     interval test (equivalent to Bounds3::IntersectP's pairwise tests, Geometry.h:1380-1406, whenever no product is NaN -- shown in
//     DESIGN.md section 4; a ray with a zero direction component can produce 0 * inf and takes the reference's exact sequence);
//   * the hit children are put in visiting order by one LDS table lookup (the three bits that fix a node's visiting order for the
//     ray's octant x hit mask -> compacted slot list) instead of a four-step select / push chain;
//   * the breadth-first top 64 nodes of the tree are read from an LDS copy: the kernel turned out to be bound by the rate at which a
//     CU's vector-memory path takes per-lane gathers (~1 lane per clock: 8 dwordx4 per node visit), not by VALU issue or by cache
//     misses, and 55 % of all node visits go to those 64 nodes;
//   * the traversal stack is LDS-only when the tree's worst case fits (template SPILL = false): no address-space branch per push / pop;
//   * node addresses are a uniform base + 32-bit offsets.
#pragma once
#include "kernels.hip.h"

namespace gnxr {

// Hit children in visiting order.  The order byte of a node for one ray octant (4 x 2-bit child slots, nearest first) is one of 8
// patterns, fixed by three bits: which half (children 0,1 or 2,3) comes first and whether each half is swapped (scene_compile.cpp:
// collapse) -- bit 1 of slot 0, bit 0 of slot 0, bit 0 of slot 2.  order_entry(code, hitMask) = the slots that are hit, nearest first,
// 2 bits each from bit 0 (their number is the mask's population count).  The 128 entries live in LDS (filled at kernel start).
GX_DEV unsigned order_entry(unsigned code, unsigned hm) {
    const unsigned base0 = (code & 4u) ? 2u : 0u, sw0 = (code >> 1) & 1u, sw1 = code & 1u, base1 = 2u - base0;
    const unsigned ord[4] = {base0 + sw0, base0 + 1u - sw0, base1 + sw1, base1 + 1u - sw1};
    unsigned e = 0, n = 0;
    for (int i = 0; i < 4; ++i)
        if ((hm >> ord[i]) & 1u) { e |= ord[i] << (2u * n); ++n; }
    return e;
}
#ifndef GX_T4_CACHE
#define GX_T4_CACHE 64
#endif
constexpr int kTopCache = GX_T4_CACHE;   // DNode4[0 .. kTopCache) -- the breadth-first top of the tree -- are served from LDS
constexpr int kRayRecDwords = 11;        // LDS record of a set-up ray: o.xyz tMax | 1/d.xyz Sx | Sy flags path  (+1 with spheres: hit code)
#ifndef GX_T4_RQ
#define GX_T4_RQ 64
#endif
constexpr int kRayQueue = GX_T4_RQ;      // set-up rays a wave parks in LDS per batch (32 frees LDS for 6 more stack levels but halves the set-up's lane use: slower)
constexpr int kRqStride = (kBlock / 64) * kRayQueue;   // dwords per record field per block
GX_DEV int trace4_lds_dwords_per_thread(bool sph) { return kRayRecDwords + (sph ? 1 : 0); }

// COUNT: count node steps / triangle tests / leaf re-tests.  SPH: the scene has spheres.  SPILL: the traversal stack may outgrow its LDS part.
// Tuning switches (tools/build_variant.sh + tests/dev_ab.py; the A/B table is in profiles/README.md).  Defaults = the fastest measured:
// 5 waves per SIMD (96 VGPRs), scalar slab arithmetic (the packed-fp32 forms measured in round 2 -- 124 VGPRs, 4 waves -- are gone), 64-ray set-up batches, 64 cached nodes.
#ifndef GX_T4_WAVES
#define GX_T4_WAVES 5
#endif
#if GX_T4_WAVES > 0
#define GX_T4_BOUNDS __launch_bounds__(kBlock, GX_T4_WAVES)
#else
#define GX_T4_BOUNDS __launch_bounds__(kBlock)
#endif
template <bool COUNT, bool SPH, bool SPILL>
__global__ void GX_T4_BOUNDS k_trace4(DScene sc, PathArrays pa, TraceWork w, unsigned int *cursor, Counters *ctr, int lds_entries, int *spill, int chunk, int n_top) {
    // LDS: [lds_entries * kBlock] stack columns | [(11 | 12) * kRqStride] ray records (SoA: field * kRqStride + wave * kRayQueue + slot) |
    //      [8 * kTopCache float4] top-of-tree nodes, SoA by plane (plane * kTopCache + node: conflict-free across nodes) | [128 B] order table
    extern __shared__ int smem[];
    typedef __attribute__((address_space(3))) float lds_float;
    lds_int *const stk = (lds_int *)&smem[threadIdx.x];
    global_int *const spl = (global_int *)(spill + (size_t)blockIdx.x * kBlock + threadIdx.x);
    const int spillStride = (int)gridDim.x * kBlock;
    const int lane = __lane_id();
    lds_int *const rq = (lds_int *)&smem[lds_entries * kBlock + (threadIdx.x >> 6) * kRayQueue];   // this wave's records: rq[field * kRqStride + slot]
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
    lds_f4 *const topN = (lds_f4 *)&smem[lds_entries * kBlock + (kRayRecDwords + (SPH ? 1 : 0)) * kRqStride];
    lds_u8 *const lut = (lds_u8 *)(topN + 8 * kTopCache);
    {   // fill the block's node cache and order table.  A vector-memory gather costs ~1 lane per clock per CU whatever it hits
        // (tools/probes/gather_probe.hip: 18 B/clk/CU for 16-byte gathers, L1-resident or not), and over half of all node visits go to
        // these few nodes (tests/dev_stats.py): from LDS they cost a ds_read instead.
        const f4v *gn = reinterpret_cast<const f4v *>(sc.nodes4);
        for (int i = threadIdx.x; i < n_top * 8; i += kBlock) topN[(i & 7) * kTopCache + (i >> 3)] = gn[i];
        if (threadIdx.x < 128) lut[threadIdx.x] = (unsigned char)order_entry(threadIdx.x >> 4, threadIdx.x & 15u);
        __syncthreads();
    }

    auto push = [&](int &n, int v) {
        if (!SPILL || n < lds_entries) stk[n * kBlock] = v;
        else spl[(n - lds_entries) * spillStride] = v;
        ++n;
    };
    auto pop = [&](int &n) -> int {
        --n;
        return (!SPILL || n < lds_entries) ? stk[n * kBlock] : spl[(n - lds_entries) * spillStride];
    };

#ifndef GX_T4_REFILL_IN_A
#define GX_T4_REFILL_IN_A 0   // measured: 48.2 vs 37.9 ms (profiles/README.md) -- the extra code in the node loop costs more than the lanes it keeps on
#endif
    unsigned poolBase = 0, poolCount = 0;   // wave-uniform: the chunk of work items this wave owns
    bool exhausted = false;                 // wave-uniform: the global cursor ran past `total`
    unsigned rqHead = 0, rqCount = 0;       // wave-uniform: set-up rays waiting in LDS

    // per-lane ray state
    bool live = false;
    int pk = 0, path = -1;   // pk: kind (bits 0-1: 0 continuation, 1 shadow, 2 MIS) | kz << 2 (Triangle.cpp:91) | done << 4 | exact << 5 | first hit ends the walk << 6 | the staged leaf is in progress (its box is decided) << 7
    V3 ro, inv;
    float Sx = 0, Sy = 0, tMax = 0;
    unsigned oNX = 0, oNY = 16, oNZ = 32;   // byte offset of the near plane of each axis inside a DNode4 (far = 48 | 80 | 112 - near ... see below)
    unsigned ordShift = 0;                  // bit offset of this octant's byte in the node's 64-bit order table
    int cur = -1, toVisit = 0, leafOff = 0, leafN = 0, hitLeaf = -1;
    uint32_t cntNodes = 0, cntTris = 0, cntRetests = 0, cntNodesGlobal = 0;
#ifdef GX_TRACE_STATS
    unsigned long long st_[24] = {0};
#endif

    // take record `slot` of this wave's ready queue into the lane
    auto take_ray = [&](unsigned slot) {
        const lds_float *q = (const lds_float *)(rq + slot);
        ro = V3(q[0 * kRqStride], q[1 * kRqStride], q[2 * kRqStride]); tMax = q[3 * kRqStride];
        inv = V3(q[4 * kRqStride], q[5 * kRqStride], q[6 * kRqStride]); Sx = q[7 * kRqStride];
        Sy = q[8 * kRqStride];
        pk = __float_as_int(q[9 * kRqStride]);
        path = __float_as_int(q[10 * kRqStride]);
        hitLeaf = SPH ? rq[slot + 11 * kRqStride] : -1;
        const int neg0 = inv.x < 0, neg1 = inv.y < 0, neg2 = inv.z < 0;
        oNX = neg0 ? 48u : 0u; oNY = neg1 ? 64u : 16u; oNZ = neg2 ? 80u : 32u;   // lox 0 loy 16 loz 32 hix 48 hiy 64 hiz 80
        ordShift = 8u * (unsigned)(neg0 | (neg1 << 1) | (neg2 << 2));
        cur = ((pk >> 4) & 1) ? -1 : sc.root4; toVisit = 0; leafN = 0;
        live = true;
    };
    // write the result of the lane's finished ray
    auto retire_ray = [&]() {
        const int kind = pk & 3;
        if (kind == 0) {
            pa.hit[path] = hitLeaf;
            // The shade class of a triangle hit is looked up from `hit` by the binning pass (k_compact_count<COMPACT_HITCLASS>: tri_class[hit]) --
            // a dependent gather here would stall the whole wave once per retire.  Only hits without a triangle get their class here.
            if (hitLeaf < 0) {
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
            const int expect = __float_as_int(pa.mis_o[(size_t)path * kRS].w);   // the leaf triangle the light sample expects (-1: nothing), written by k_shade
            bool ok = (expect >= 0) ? (hitLeaf == expect) : (hitLeaf == -1);
            if (w.vis) w.vis[4 * (size_t)path + 1] = ok ? 1 : 0;
            else reinterpret_cast<float *>(&pa.mis_o[(size_t)path * kRS])[3] = ok ? 1.f : 0.f;
        }
        live = false;
    };
#ifdef GX_TRACE_STATS
#define GX_TICK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); GX_STAT(i, t_ - tick_); tick_ = t_; } while (0)
    unsigned long long tick_ = __builtin_amdgcn_s_memtime();
#else
#define GX_TICK(i) do {} while (0)
#endif
    while (true) {
        GX_STAT(0, 1);
        GX_TICK(15);   // retire + loop overhead of the previous trip
        // ---------------- refill ----------------
        const bool need = !live;
        const unsigned long long needMask = __ballot(need);
        if (needMask) {
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
                    // ---- batch set-up: every lane prepares one work item (the ray-only part of Bounds3::IntersectP and Triangle::Intersect)
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
                            // NEE work items: first all shadow rays, then all MIS rays
                            const unsigned e = i - (unsigned)w.n_closest;
                            const bool isShadow = e < (unsigned)w.n_nee;
                            path_ = w.q_nee[isShadow ? e : e - (unsigned)w.n_nee];
                            // PathIntegrator launches (w.vis: one record per path, every record array has an entry for it): the record's flags and
                            // its ray are loaded together -- one memory round trip after the queue read, not two; a record without the ray has
                            // stale floats there, which are dropped.  Other integrators size the MIS arrays by what they use: flags first.
                            const bool together = w.vis != nullptr;
                            if (isShadow) {
                                kind_ = 1; any_ = 1;
                                o4 = pa.sh_o[(size_t)path_ * kRS]; d4 = pa.sh_d[(size_t)path_ * kRS]; tMax_ = o4.w;
                                valid = (__float_as_int(d4.w) & 1) != 0;     // else: this vertex spawned no shadow ray
                            } else {
                                kind_ = 2;
                                const int nflags = __float_as_int(pa.sh_d[(size_t)path_ * kRS].w);
                                valid = (nflags & 2) != 0;
                                if (together || valid) { o4 = pa.mis_o[(size_t)path_ * kRS]; d4 = pa.mis_d[(size_t)path_ * kRS]; }
                                else { o4 = make_float4(0, 0, 0, 0); d4 = o4; }
                                tMax_ = GX_INF;
                                // a MIS ray that expects to escape (an infinite light was sampled) asks "is there any hit at all": with tMax = inf
                                // the closest-hit walk and the any-hit walk visit the same nodes up to the first accepted triangle, and that
                                // triangle already decides the answer
                                any_ = __float_as_int(o4.w) < 0 ? 1 : 0;
                            }
                        }
                        if (valid) {
                            const V3 o(o4.x, o4.y, o4.z), d(d4.x, d4.y, d4.z);
                            const V3 iv(1.f / d.x, 1.f / d.y, 1.f / d.z);          // BVHAccel.cpp:657
                            const RayShear sh = ray_shear(d);                       // Triangle.cpp:91-105 (Sz == 1 / d[kz] == iv[kz])
                            int done = 0;
                            for (int si = 0; SPH && si < sc.n_spheres; ++si) {     // spheres live outside the BVH and are tested first
                                float tH;
                                if (sphere_test(sc.spheres[si], o, d, tMax_, &tH)) {
                                    sphHit = -2 - si;
                                    if (any_) { done = 1; break; }
                                    tMax_ = tH;
                                }
                            }
                            // a zero direction component makes 1 / d infinite and (plane - o) * (1 / d) possibly NaN: such rays walk with the
                            // reference's exact comparison sequence
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
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the records are read by other lanes of this wave
                    rqCount = (unsigned)__popcll(vm); rqHead = 0;
                    poolBase += take; poolCount -= take;
                    GX_STAT(7, 1);
                    GX_STAT(8, take);
                }
            }
            if (rqCount > 0) {
                const unsigned rank = (unsigned)__popcll(needMask & ((1ull << lane) - 1ull));
                if (need && rank < rqCount) take_ray(rqHead + rank);
                const unsigned t = min(rqCount, (unsigned)__popcll(needMask));
                rqHead += t; rqCount -= t;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");       // all reads of the queue precede the next batch's writes
            }
        }
        GX_TICK(11);
        const unsigned long long liveMask = __ballot(live);
        if (liveMask == 0) {
            if (exhausted && rqCount == 0 && poolCount == 0) break;
            continue;   // nothing was ready this round: fetch / set up on the next iteration
        }

        // ---------------- phase A: interior traversal until at most half of the live lanes still look for a leaf ----------------
        int nLive = __popcll(liveMask);
        GX_STAT(9, nLive);
        while (true) {
#if GX_T4_REFILL_IN_A
            // a lane whose ray has ended does not wait for phase C: it writes its result and takes the next set-up ray from the wave's
            // queue right here (a pop is a dozen ds_reads), so the node steps below run with more lanes on
            {
                const bool fin = live && cur == -1 && leafN == 0;
                const unsigned long long finMask = __ballot(fin);
                if (finMask != 0 && rqCount > 0) {
                    if (fin) retire_ray();
                    const unsigned rank = (unsigned)__popcll(finMask & ((1ull << lane) - 1ull));
                    if (fin && rank < rqCount) take_ray(rqHead + rank);
                    const unsigned t = min(rqCount, (unsigned)__popcll(finMask));
                    rqHead += t; rqCount -= t;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    nLive = __popcll(__ballot(live));
                }
            }
#endif
            if (live && leafN == 0 && cur < -1) {   // the next reference is a leaf: stage it, pre-pop its successor
                const int lr = ~cur;
                leafOff = lr & 0xffffff; leafN = (lr >> 24) & 0x7f;
                cur = (toVisit == 0) ? -1 : pop(toVisit);
            }
            // A lane that already holds a leaf keeps walking (speculatively, with the tMax it has) until it reaches a second
            // leaf: every node it visits is one the reference visits or a superset of them (tMax only shrinks), and each
            // leaf is re-tested against the current tMax before its triangles are (phase B), so results do not change.
            const bool searching = live && cur >= 0 && (kSpeculate || leafN == 0);
            const bool hungry = searching && leafN == 0;
            const int nSearching = __popcll(__ballot(hungry));
            if (nSearching * kTraceLeaveDiv <= nLive * kTraceLeaveMul && (nSearching == 0 || nSearching < nLive)) break;
#ifdef GX_TRACE_STATS
            { const int ns = __popcll(__ballot(searching)); GX_STAT(1, 1); GX_STAT(2, ns); const int nh_ = __popcll(__ballot(live && leafN > 0 && !searching)), nf_ = __popcll(__ballot(live && cur == -1 && leafN == 0)), nsp_ = __popcll(__ballot(searching && leafN > 0));
              GX_STAT(20, nLive); GX_STAT(21, nh_); GX_STAT(22, nf_); GX_STAT(23, nsp_);
              const int n16 = __popcll(__ballot(searching && cur < 16)), n64 = __popcll(__ballot(searching && cur < 64)), n256 = __popcll(__ballot(searching && cur < 256)), n1k = __popcll(__ballot(searching && cur < 1024));
              GX_STAT(16, n16); GX_STAT(17, n64); GX_STAT(18, n256); GX_STAT(19, n1k); }
#endif
            if (searching) {
                if (COUNT) cntNodes++;
                // ---- one 4-wide step: test the four children (Bounds3::IntersectP per box), continue with the first one hit in the
                // reference's visiting order, push the others farthest first
                f4v nX, fX, nY, fY, nZ, fZ, cf;
                uint2 tb;
                if (cur < n_top) {   // top of the tree: from the block's LDS copy
                    const lds_f4 *L = topN + cur;
                    nX = L[(oNX >> 4) * kTopCache]; fX = L[((48u - oNX) >> 4) * kTopCache];
                    nY = L[(oNY >> 4) * kTopCache]; fY = L[((80u - oNY) >> 4) * kTopCache];
                    nZ = L[(oNZ >> 4) * kTopCache]; fZ = L[((112u - oNZ) >> 4) * kTopCache];
                    cf = L[6 * kTopCache];
                    const f4v t7 = L[7 * kTopCache];
                    tb = make_uint2(__float_as_uint(t7.x), __float_as_uint(t7.y));
                } else {
                    if (COUNT) cntNodesGlobal++;
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
                    // finite 1 / d: no product is NaN, and the pairwise tests of Geometry.h:1389-1403 hold exactly when the merged
                    // interval [max of the three entries, min of the three (k-scaled) exits] is non-empty and overlaps (0, tMax)
// (rounding is monotone and k > 0, so the smallest of the three k-scaled exits is the k-scaled smallest exit: one multiply instead of three)
#define GX_SLAB3(C, BIT)                                                                                       \
    {                                                                                                          \
        const float e = fmaxf(fmaxf((nX.C - ro.x) * inv.x, (nY.C - ro.y) * inv.y), (nZ.C - ro.z) * inv.z);      \
        const float x = fminf(fminf((fX.C - ro.x) * inv.x, (fY.C - ro.y) * inv.y), (fZ.C - ro.z) * inv.z) * k;  \
        hitMask |= (e <= x && e < tMax && x > 0.f) ? (BIT) : 0u;                                                \
    }
                    GX_SLAB3(x, 1u) GX_SLAB3(y, 2u) GX_SLAB3(z, 4u) GX_SLAB3(w, 8u)
#undef GX_SLAB3
                } else {
#define GX_SLAB(C, BIT)                                                                   \
    {                                                                                     \
        float tMin = (nX.C - ro.x) * inv.x, tMx = (fX.C - ro.x) * inv.x;                  \
        float tyMin = (nY.C - ro.y) * inv.y, tyMax = (fY.C - ro.y) * inv.y;               \
        tMx *= k; tyMax *= k;                                                             \
        bool ok = !(tMin > tyMax || tyMin > tMx);                                         \
        if (tyMin > tMin) tMin = tyMin;                                                   \
        if (tyMax < tMx) tMx = tyMax;                                                     \
        float tzMin = (nZ.C - ro.z) * inv.z, tzMax = (fZ.C - ro.z) * inv.z;               \
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
                if (hitMask == 0) next = (toVisit == 0) ? -1 : pop(toVisit);
                else {
                    const unsigned word = (ordShift >= 32u ? tb.y : tb.x) >> (ordShift & 31u);   // this octant's order byte in the low bits
                    const unsigned e = lut[((word & 3u) << 5) | (word & 16u) | hitMask];               // (slot0 bit 1, slot0 bit 0, slot2 bit 0, hit mask)
                    const int n = __popc(hitMask);
                    const int c0 = __float_as_int(cf.x), c1 = __float_as_int(cf.y), c2 = __float_as_int(cf.z), c3 = __float_as_int(cf.w);
                    auto child = [&](unsigned s) -> int { const int lo = (s & 1u) ? c1 : c0, hi = (s & 1u) ? c3 : c2; return (s & 2u) ? hi : lo; };
                    if (n >= 4) push(toVisit, child(e >> 6));
                    if (n >= 3) push(toVisit, child((e >> 4) & 3u));
                    if (n >= 2) push(toVisit, child((e >> 2) & 3u));
                    next = child(e & 3u);
                }
                cur = next;   // interior index (>= 0), leaf reference (< -1) or -1: done
            }
        }
        GX_TICK(12);
        // ---------------- phase B: triangle tests ----------------
#ifdef GX_TRACE_STATS
        {
            unsigned long long lm = __ballot(live && leafN > 0);
            if (lm) {
                int mx = leafN;
                for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
                int sm = live ? leafN : 0;
                for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
                const int nre = __popcll(__ballot(live && leafN > 0 && hitLeaf >= 0));
                GX_STAT(3, 1); GX_STAT(4, __popcll(lm)); GX_STAT(5, mx); GX_STAT(6, sm); GX_STAT(10, nre);
            }
        }
#endif
        if (live && leafN > 0) {
            bool visit = true;
            // BVHAccel::Intersect tests a leaf's box when it pops it, i.e. against the CURRENT ray.tMax; the 4-wide step tested it earlier with
            // an older tMax.  tMax only ever shrinks after a hit (before the first hit the earlier test stands), so a ray that has one re-tests
            // the leaf's own bounds -- exact ties (t == tMax on flat, axis-aligned boxes such as the Cornell walls) then resolve as in the
            // reference.  The bounds of a one-triangle leaf are the componentwise min / max of its vertices (Triangle::WorldBound, exact),
            // which this phase loads anyway; only larger leaves read the floats of their LinearBVHNode (leaf_box, addressed by the first triangle).
            //
            // ONE triangle per lane and trip.  A leaf of several triangles (the reference keys its build on the centroid of a primitive's BOUNDS,
            // so the two triangles of an axis-aligned quad -- every Cornell wall -- share a leaf, and almost every ray ends on a wall) stays staged
            // with its next triangle: looping over it here kept the whole wave for a second round with a handful of lanes on in more than half
            // of all trips (tests/dev_stats.py: 1.55 rounds per trip).  Its box is tested once, at its first triangle (pk bit 7 remembers);
            // the triangles are still tested in order, each against the tMax the previous ones left.
            const bool first = (pk & 128) == 0;
            const bool retest = first && (SPH ? hitLeaf != -1 : hitLeaf >= 0);
            const bool fromVerts = retest && leafN == 1 && sc.leaf1_from_verts;
            int neg[3] = {inv.x < 0, inv.y < 0, inv.z < 0};
            if (retest && !fromVerts) {
                const float4 b0 = sc.leaf_box[2 * (size_t)leafOff], b1 = sc.leaf_box[2 * (size_t)leafOff + 1];
                if (COUNT) cntRetests++;
                visit = slab_test(b0, b1, ro, inv, neg, tMax);
            }
            bool more = false;   // the leaf has further triangles to test
            if (visit) {
                RayShear shear;
                const int kz = (pk >> 2) & 3;
                shear.kz = kz; shear.kx = kz == 2 ? 0 : kz + 1; shear.ky = shear.kx == 2 ? 0 : shear.kx + 1;
                shear.Sx = Sx; shear.Sy = Sy; shear.Sz = kz == 0 ? inv.x : (kz == 1 ? inv.y : inv.z);
                V3 p0, p1, p2;
                load_tri(tris, leafOff, &p0, &p1, &p2);
                if (COUNT) cntTris++;
                bool inBox = true;
                if (fromVerts) {
                    const float4 b0 = make_float4(fminf(fminf(p0.x, p1.x), p2.x), fminf(fminf(p0.y, p1.y), p2.y), fminf(fminf(p0.z, p1.z), p2.z), fmaxf(fmaxf(p0.x, p1.x), p2.x));
                    const float4 b1 = make_float4(fmaxf(fmaxf(p0.y, p1.y), p2.y), fmaxf(fmaxf(p0.z, p1.z), p2.z), 0.f, 0.f);
                    inBox = slab_test(b0, b1, ro, inv, neg, tMax);
                }
                more = leafN > 1;
                TriHit h;
                if (inBox && tri_test_sheared(p0, p1, p2, ro, shear, tMax, &h)) {
                    hitLeaf = leafOff;
                    if (pk & 64) { cur = -1; more = false; }   // IntersectP returns at the first hit (and so may a MIS ray that expects a miss)
                    else tMax = h.t;                            // GeometricPrimitive::Intersect shrinks ray.tMax
                }
            }
            if (more) { leafOff += 1; leafN -= 1; pk |= 128; }
            else { leafN = 0; pk &= ~128; }
        }
        GX_TICK(13);
        // ---------------- phase C: retire finished rays ----------------
        if (live && cur == -1 && leafN == 0) retire_ray();
    }
#ifdef GX_TRACE_STATS
    if (lane == 0) for (int i = 0; i < 24; ++i) if (st_[i]) atomicAdd(&g_trace_stats[i], st_[i]);
#endif
    if (COUNT) {
        atomicAdd(&ctr->nodes, (unsigned long long)cntNodes);
        atomicAdd(&ctr->tris, (unsigned long long)cntTris);
        atomicAdd(&ctr->retests, (unsigned long long)cntRetests);
        atomicAdd(&ctr->nodes_global, (unsigned long long)cntNodesGlobal);
    }
}

}  // namespace gnxr

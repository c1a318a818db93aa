// trace_kernel.hip.h -- k_trace: the unified BVH traversal stage (BVHAccel::Intersect / IntersectP,
// accelerator/BVHAccel.cpp:653-729) for all three ray kinds a path vertex produces:
//   kind 0  continuation ray      closest hit  -> PathArrays::hit[path]
//   kind 1  shadow ray            any hit      -> sh_o[path].w  = 1 if unoccluded        (Light.cpp:28-31)
//   kind 2  MIS ray               closest hit  -> mis_o[path].w = 1 if it found what the light sample expects
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
//   * per-lane stack column in LDS: stack[depth * 256 + tid], bank == lane.
#pragma once
#include "device_geom.h"

namespace gnxr {

struct TraceWork {
    const int *q_closest; int n_closest;   // path slots (nullptr == identity)
    const int *q_nee; int n_nee;           // paths with an NEE record: two work items each (shadow ray, MIS ray)
};

constexpr int kTraceChunk = 512;   // rays a wave takes per global atomic

template <int STACK, bool COUNT>
__global__ void __launch_bounds__(kBlock) k_trace(DScene sc, PathArrays pa, TraceWork w, unsigned int *cursor, Counters *ctr) {
    __shared__ int stack_mem[STACK * kBlock];
    int *stack = &stack_mem[threadIdx.x];
    const int lane = __lane_id();
    const unsigned total = (unsigned)w.n_closest + 2u * (unsigned)w.n_nee;
    const float4 *__restrict__ nodes = sc.nodes;
    const DTri *__restrict__ tris = sc.tris;

    unsigned poolBase = 0, poolCount = 0;   // wave-uniform
    bool exhausted = false;                 // wave-uniform: the global cursor ran past `total`

    // per-lane ray state
    int item = -1, kind = 0, path = -1;
    V3 ro, rd, invDir;
    float tMax = 0;
    int neg0 = 0, neg1 = 0, neg2 = 0;
    int cur = -1, toVisit = 0, leafOff = 0, leafN = 0, hitLeaf = -1, expect = -1;
    uint32_t cntNodes = 0, cntTris = 0;

    while (true) {
        // ---------------- refill idle lanes from the wave pool ----------------
        bool need = item < 0;
        unsigned long long needMask = __ballot(need);
        if (needMask) {
            if (poolCount == 0 && !exhausted) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(cursor, (unsigned)kTraceChunk);
                base = __shfl(base, 0);
                if (base >= total) exhausted = true;
                else { poolBase = base; poolCount = min((unsigned)kTraceChunk, total - base); }
            }
            if (poolCount > 0) {
                unsigned rank = (unsigned)__popcll(needMask & ((1ull << lane) - 1ull));
                unsigned take = min(poolCount, (unsigned)__popcll(needMask));
                if (need && rank < take) {
                    unsigned i = poolBase + rank;
                    item = (int)i;
                    float4 o4, d4;
                    if (i < (unsigned)w.n_closest) {
                        kind = 0;
                        path = w.q_closest ? w.q_closest[i] : (int)i;
                        o4 = pa.ray_o[path]; d4 = pa.ray_d[path];
                        tMax = o4.w;
                    } else {
                        unsigned e = i - (unsigned)w.n_closest;
                        path = w.q_nee[e >> 1];
                        int nflags = __float_as_int(pa.sh_d[path].w);
                        if ((e & 1u) == 0) {
                            kind = 1;
                            if (nflags & 1) { o4 = pa.sh_o[path]; d4 = pa.sh_d[path]; tMax = o4.w; }
                            else item = -1;     // this vertex spawned no shadow ray
                        } else {
                            kind = 2;
                            if (nflags & 2) { o4 = pa.mis_o[path]; d4 = pa.mis_d[path]; expect = __float_as_int(o4.w); tMax = GX_INF; }
                            else item = -1;
                        }
                    }
                    if (item >= 0) {
                        ro = V3(o4.x, o4.y, o4.z); rd = V3(d4.x, d4.y, d4.z);
                        invDir = V3(1.f / rd.x, 1.f / rd.y, 1.f / rd.z);
                        neg0 = invDir.x < 0; neg1 = invDir.y < 0; neg2 = invDir.z < 0;
                        cur = 0; toVisit = 0; leafN = 0; hitLeaf = -1;
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
        while (true) {
            bool searching = item >= 0 && cur >= 0 && leafN == 0;
            if (__ballot(searching) == 0) break;
            if (searching) {
                float4 n0 = nodes[2 * cur], n1 = nodes[2 * cur + 1];
                if (COUNT) cntNodes++;
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
                    stack[(toVisit++) * kBlock] = farC;
                    next = nearC;
                } else {
                    if (hitBox) { leafOff = offset; leafN = nPrims; }
                    next = (toVisit == 0) ? -1 : stack[(--toVisit) * kBlock];
                }
                cur = next;
            }
        }
        // ---------------- phase B: triangle tests ----------------
        if (item >= 0 && leafN > 0) {
            for (int i = 0; i < leafN; ++i) {
                V3 p0, p1, p2;
                load_tri(tris, leafOff + i, &p0, &p1, &p2);
                if (COUNT) cntTris++;
                TriHit h;
                if (tri_test(p0, p1, p2, ro, rd, tMax, &h)) {
                    hitLeaf = leafOff + i;
                    if (kind == 1) { cur = -1; break; }   // IntersectP returns at the first hit
                    tMax = h.t;                            // GeometricPrimitive::Intersect shrinks ray.tMax
                }
            }
            leafN = 0;
        }
        // ---------------- phase C: retire finished rays ----------------
        if (item >= 0 && cur < 0 && leafN == 0) {
            if (kind == 0) {
                pa.hit[path] = hitLeaf;
                int cls = 0;   // misses and null materials only need the emission / pass-through code of class 0
                if (hitLeaf >= 0) { int mat = tris[hitLeaf].material; if (mat >= 0) cls = sc.materials[mat].shade_class; }
                pa.pclass[path] = (unsigned char)cls;
            }
            else if (kind == 1) reinterpret_cast<float *>(&pa.sh_o[path])[3] = hitLeaf < 0 ? 1.f : 0.f;
            else {
                bool ok = (expect >= 0) ? (hitLeaf == expect) : (hitLeaf < 0);
                reinterpret_cast<float *>(&pa.mis_o[path])[3] = ok ? 1.f : 0.f;
            }
            item = -1;
        }
    }
    if (COUNT) {
        atomicAdd(&ctr->nodes, (unsigned long long)cntNodes);
        atomicAdd(&ctr->tris, (unsigned long long)cntTris);
    }
}

// L += beta * (EstimateDirect(...) / lightPdf), core/Integrator.cpp:78 + PathIntegrator.cpp:135-141, once the
// two visibility results of the vertex are known.  Pure streaming.
__global__ void __launch_bounds__(kBlock) k_nee_combine(PathArrays pa, const int *__restrict__ queue, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int path = queue[i];
        float4 sd4 = pa.sh_d[path], X4 = pa.sh_X[path];
        int flags = __float_as_int(sd4.w);
        Spec Ld(0.f);
        if ((flags & 1) && pa.sh_o[path].w == 1.f) Ld = Ld + Spec(X4.x, X4.y, X4.z);
        if (flags & 2) {
            float4 Y4 = pa.mis_Y[path];
            Spec Y(Y4.x, Y4.y, Y4.z);
            if (pa.mis_o[path].w == 1.f && !Y.is_black()) Ld = Ld + Y;
        }
        float4 nb = pa.nbeta[path], L4 = pa.L[path];
        Spec add = Spec(nb.x, nb.y, nb.z) * (Ld / X4.w);
        pa.L[path] = make_float4(L4.x + add.r, L4.y + add.g, L4.z + add.b, 0.f);
    }
}

}  // namespace gnxr

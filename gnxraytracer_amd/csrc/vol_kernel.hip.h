// vol_kernel.hip.h -- VolPathIntegrator::Li (integrators/VolPathIntegrator.cpp:24-159) as a wavefront state machine.
//
// In a scene with media every ray of a path vertex is a closest-hit ray (VisibilityTester::Tr and Scene::IntersectTr
// walk through null-material boundaries with Scene::Intersect, core/Light.cpp:33-53, core/Scene.cpp:26-40), and the
// delta / ratio tracking loops of each segment draw from the path's own Halton stream.  The number of dimensions a
// segment consumes depends on what the previous segment hit, so the rays of one path are strictly sequential.  Each
// path therefore has exactly ONE ray in flight; a round is
//
//     k_trace      closest hit for every live path
//     k_vol_media  Medium::Sample / Medium::Tr for the paths whose ray travels inside a medium (delta / ratio tracking)
//     k_vol_step   advance every live path to its next ray
//
// The tracking loops are the bulk of the work of config 5 (hundreds of null collisions per segment at sigma_t = 100)
// and their length varies from 0 to ~1000 iterations between neighbouring paths, so they run in their own lean kernel:
// persistent waves whose lanes pull the next path as soon as their loop ends (the same pool scheme as k_trace), a few
// dozen VGPRs instead of the 256 the BSDF code needs, 8 waves per SIMD to hide the latency of the permutation-table
// and density loads.
//
// and a path is in one of three states:
//   VS_MAIN    the ray is the path's main ray: medium sampling, Le, null-boundary pass-through, then the vertex
//              (surface or medium interaction) sets up UniformSampleOneLight (core/Integrator.cpp:57-79, 93-210)
//   VS_SHADOW  the ray is one segment of VisibilityTester::Tr towards the sampled light point
//   VS_MIS     the ray is one segment of Scene::IntersectTr along the BSDF- / phase-sampled direction
// After the light estimate is complete the vertex is re-established from its saved main ray (same arithmetic, same
// bits) and the continuation direction is sampled at the stream position the transmittance loops left behind.
#pragma once
#include "device_media.h"

namespace gnxr {

enum : int { VS_MAIN = 0, VS_SHADOW = 1, VS_MIS = 2, VS_PARKED = 3 };   // VS_PARKED: only in VolArrays::state (k_vol_media's step cap)

struct VolArrays {
    int4 *vs;        // x: state, y: Halton dimension, z: hit leaf of the saved vertex, w: bit 0: the main ray is still the camera ray (it has ray differentials),
                     //    bit 1 (kVsCont): k_vol_media left the segment in flight at its step cap -- the path sits this round out (its `state` is
                     //    VS_PARKED, which no k_vol_step bin takes) and the tracking loop goes on from mres in the next one
    float4 *sv_o;    // saved main ray of the vertex: o.xyz, tMax
    float4 *sv_d;    // d.xyz, medium (int bits)
    float4 *p1;      // light sample point p1.xyz, w: light-selection pdf
    float4 *p1e;     // p1Error.xyz, w: light pdf (Sample_Li)
    float4 *n1;      // n1.xyz, w: MIS weight of the light sample
    float4 *f;       // f.xyz of the light sample (|cos| included), w: flags (bit0 light-sample part, bit1 scattering part, bit2 Li2 != 0)
    float4 *Li;      // Li.xyz, w: leaf the scattering ray must reach (int bits; -1 == must escape)
    float4 *Tr;      // transmittance accumulated along the current NEE ray, w: t of the medium interaction (-1: surface vertex)
    float4 *Ld;      // direct lighting accumulated so far, w: MIS weight of the scattering sample
    float4 *mis_o;   // scattering ray origin, w: medium (int bits)
    float4 *mis_d;   // scattering ray direction, w: scattering pdf
    float4 *mis_Y;   // f * Li2
    unsigned char *state;   // copy of vs.x for the live paths: key of the per-state binning before k_vol_step
    float4 *mres;    // k_vol_media result for the ray in flight: Medium::Sample weight / Medium::Tr (rgb), w: t of the sampled interaction or -1;
                     // while vs.w has kVsCont: x: Tr so far, y: t reached (same ray, same stream position next round: same result)
    int *orig;       // the slot (sample j * npix + pixel) a path started in: packing (k_vol_pack) moves the live paths to the front of the state arrays
    float4 *Lout;    // final radiance of a path, written once when it ends, at its ORIGINAL slot: what k_resolve sums in sample order
    // sv_o sv_d | p1 p1e | mis_o mis_d are the fields of the slot's record groups 2 - 4 (PathArrays, kernels.hip.h): element i at [i * kRS]
    __host__ __device__ void bind_records(float4 *const g[kRecGroups]) { sv_o = g[2]; sv_d = g[2] + 1; p1 = g[3]; p1e = g[3] + 1; mis_o = g[4]; mis_d = g[4] + 1; }
};

static __global__ void __launch_bounds__(kBlock) k_vol_init(PathArrays pa, VolArrays va, int n_paths) {
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n_paths; slot += gridDim.x * blockDim.x) {
        uint2 m = pa.meta[(size_t)slot * kRSm];
        va.vs[slot] = make_int4(VS_MAIN, (int)m.y, -1, 1);
        va.state[slot] = (unsigned char)VS_MAIN;
        va.orig[slot] = slot;
        pa.store_meta(slot, m.x, 0u);   // y: bounces << 16 | specularBounce << 31
    }
}

// Interaction::GetMedium(w) for a surface hit: GeometricPrimitive::Intersect (core/Primitive.cpp:32-46) keeps the
// primitive's MediumInterface when it is a transition and the ray's medium otherwise.
GX_DEV void hit_interface(const DScene &sc, const DMediaTables &mt, int leaf, int rayMedium, int *mi, int *mo) {
    *mi = rayMedium; *mo = rayMedium;
    if (leaf < -1) {   // sphere
        const DSphere &sph = sc.spheres[-2 - leaf];
        if (sph.med_in != sph.med_out) { *mi = sph.med_in; *mo = sph.med_out; }
    } else if (mt.tri_media) {
        int2 tm = mt.tri_media[leaf];
        if (tm.x != tm.y) { *mi = tm.x; *mo = tm.y; }
    }
}
GX_DEV int hit_medium(const DScene &sc, const DMediaTables &mt, int leaf, int rayMedium, V3 n, V3 w) {
    int mi, mo;
    hit_interface(sc, mt, leaf, rayMedium, &mi, &mo);
    return dot(w, n) > 0 ? mo : mi;
}

// ------------------------------------------------------------------------------------------------
// Medium::Sample (state VS_MAIN) / Medium::Tr (VS_SHADOW, VS_MIS) for the rays that travel inside a medium.
// `queue` lists those paths (pflags bit1 of the previous step, compacted).  Persistent waves; a lane runs one tracking
// loop and refills from the wave's pool when it ends.  Results: va.mres[path], and the path's stream position va.vs[path].y.
constexpr int kMediaChunk = 256;   // most paths a wave takes per global atomic (smaller for thin launches, chosen by the host)

// COUNT: add the number of tracking-loop iterations to ctr->media_steps (profiling run; gnxr_set_profiling bit 2)
constexpr int kVmRecDwords = 13;                 // LDS record of a set-up medium segment (k_vol_media)
constexpr int kVsCont = 2;                       // VolArrays::vs[].w bit: the segment is to be continued
constexpr int kVmStride = (kBlock / 64) * 64;     // ints per record field per block
template <bool COUNT>
__global__ void __launch_bounds__(kBlock) k_vol_media(DScene sc, DMediaTables mt, PathArrays pa, VolArrays va, const int *__restrict__ queue, int n, unsigned int *cursor, int chunk, Counters *ctr, int step_cap) {
    // LDS: kVmRecDwords x (4 waves x 64) ints -- the set-up segments a wave has parked (SoA: field * kVmStride + wave * 64 + slot)
    extern __shared__ int vm_smem[];
    lds_int *const rq = (lds_int *)&vm_smem[(threadIdx.x >> 6) * 64];
    unsigned long long cntSteps = 0;
    unsigned cntCont = 0;   // segments this lane left unfinished at the step cap
    int nsteps = 0;         // tracking steps of the lane's current segment in THIS launch
    int camBit = 0, realState = 0, wasParked = 0;   // bit 0 of the path's vs.w, its state, and whether this launch continues a parked segment
    const int lane = __lane_id();
    const unsigned total = (unsigned)n;
    unsigned poolBase = 0, poolCount = 0;
    bool exhausted = false;
    unsigned rqHead = 0, rqCount = 0;   // wave-uniform: set-up segments waiting in LDS
    // per-lane tracking state (grid media; homogeneous media are closed-form and finish at set-up)
    int path = -1, dim = 0, nx = 0, ny = 0, nz = 0, medium = -1;
    bool sampleMode = false;
    uint32_t index = 0;
    V3 o, d;
    float t = 0, tMax = 0, Tr = 1, invMaxDensity = 0, stepScale = 0, cachedU = 0;
    int cachedDim = -1;   // dimension whose value is already in cachedU (the second half of the last pair drawn)
    const float *__restrict__ dens = nullptr;

    while (true) {
        bool need = path < 0;
        unsigned long long needMask = __ballot(need);
        if (needMask) {
            // A segment's set-up (the triangle re-test that gives its length, the ray in medium space, the clip against the grid's box:
            // ~350 instructions) used to run for whichever lanes had just finished -- one or two of 64 in nearly every iteration of a
            // loop whose segments take ~20 tracking steps.  As in k_trace4 it now runs in batches: when the wave's ready queue is empty
            // ALL lanes set one work item up each (lanes in the middle of a segment too: their tracking state is untouched) and park the
            // grid-medium segments in LDS; an idle lane then just pops a record.
            if (rqCount == 0) {
                if (poolCount == 0 && !exhausted) {
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(cursor, (unsigned)chunk);
                    base = __shfl(base, 0);
                    if (base >= total) exhausted = true;
                    else { poolBase = base; poolCount = min((unsigned)chunk, total - base); }
                }
                if (poolCount > 0) {
                    const unsigned take = min(poolCount, 64u);
                    bool valid = false;
                    int rPath = -1, rMed = 0, rDim = 0;
                    uint32_t rIndex = 0;
                    V3 rO, rD;
                    float rT = 0, rTMax = 0, rTr = 1.f;
                    if ((unsigned)lane < take) {
                        const int p = queue ? queue[poolBase + lane] : (int)(poolBase + lane);
                        int4 vs = va.vs[p];
                        float4 o4 = pa.ray_o[(size_t)p * kRS], d4 = pa.ray_d[(size_t)p * kRS];
                        V3 ro(o4.x, o4.y, o4.z), rd(d4.x, d4.y, d4.z);
                        const int med = __float_as_int(d4.w);
                        const int leaf = pa.hit[p];
                        bool found = leaf != -1;
                        int triMat = -1;
                        TriHit h;
                        if (leaf < -1) {
                            const DSphere &sph = sc.spheres[-2 - leaf];
                            triMat = sph.material;
                            found = sphere_test(sph, ro, rd, o4.w, &h.t);
                        } else if (found) {
                            const float4 *q = reinterpret_cast<const float4 *>(sc.tris + leaf);
                            float4 a = q[0], b = q[1], c = q[2];
                            triMat = __float_as_int(b.w);
                            found = tri_test(V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), ro, rd, o4.w, &h);
                        }
                        // VisibilityTester::Tr returns 0 at an opaque hit before it evaluates the segment's transmittance
                        const bool skip = med < 0 || (vs.x == VS_SHADOW && found && triMat >= 0);
                        if (!skip) {
                            const float tSeg = found ? h.t : o4.w;
                            const DMedium &m = mt.media[med];
                            SampleStream ss(sc.st, pa.meta[(size_t)p * kRSm].x, vs.y);
                            if (m.type == GNXR_MEDIUM_HOMOGENEOUS) {
                                float4 res;
                                if (vs.x == VS_MAIN) {
                                    bool ok; float tt = -1.f;
                                    Spec w = medium_sample(mt, med, ro, rd, tSeg, ss, &ok, &tt);
                                    res = make_float4(w.r, w.g, w.b, ok ? tt : -1.f);
                                } else {
                                    Spec w = medium_tr(mt, med, ro, rd, tSeg, ss);
                                    res = make_float4(w.r, w.g, w.b, -1.f);
                                }
                                va.mres[p] = res;
                                reinterpret_cast<int *>(&va.vs[p])[1] = ss.dim;
                            } else {
                                // GridDensityMedium::Sample / Tr set-up, GridDensityMedium.cpp:33-40, 59-66
                                float rtMax, tMin, tEnd;
                                V3 oo, dd;
                                xform_ray(m.w2m, ro, normalize(rd), tSeg * length(rd), &oo, &dd, &rtMax);
                                if (!unit_box_intersect(oo, dd, rtMax, &tMin, &tEnd)) {
                                    va.mres[p] = make_float4(1.f, 1.f, 1.f, -1.f);
                                } else {
                                    valid = true;
                                    rPath = p; rMed = med | (vs.x == VS_MAIN ? 0x10000 : 0);
                                    rIndex = ss.index; rDim = ss.dim;
                                    rO = oo; rD = dd; rT = tMin; rTMax = tEnd;
                                    // a segment the last round left at the step cap goes on where it stopped (vs.y already is its stream position)
                                    if (vs.w & kVsCont) { const float4 prev = va.mres[p]; rTr = prev.x; rT = prev.y; }
                                    // the camera-ray flag, the state and whether this is a continuation travel with the record: vs.w / state are rewritten when the segment parks, and again when it ends
                                    rMed |= ((vs.w & 1) << 17) | ((vs.x & 3) << 18) | ((vs.w & kVsCont) ? (1 << 20) : 0);
                                }
                            }
                        }
                    }
                    const unsigned long long vmask = __ballot(valid);
                    if (valid) {
                        lds_int *q = rq + __popcll(vmask & ((1ull << lane) - 1ull));
                        q[0 * kVmStride] = rPath; q[1 * kVmStride] = rMed; q[2 * kVmStride] = (int)rIndex; q[3 * kVmStride] = rDim;
                        q[4 * kVmStride] = __float_as_int(rO.x); q[5 * kVmStride] = __float_as_int(rO.y); q[6 * kVmStride] = __float_as_int(rO.z);
                        q[7 * kVmStride] = __float_as_int(rD.x); q[8 * kVmStride] = __float_as_int(rD.y); q[9 * kVmStride] = __float_as_int(rD.z);
                        q[10 * kVmStride] = __float_as_int(rT); q[11 * kVmStride] = __float_as_int(rTMax); q[12 * kVmStride] = __float_as_int(rTr);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the records are read by other lanes of this wave
                    rqCount = (unsigned)__popcll(vmask); rqHead = 0;
                    poolBase += take; poolCount -= take;
                }
            }
            if (rqCount > 0) {
                const unsigned rank = (unsigned)__popcll(needMask & ((1ull << lane) - 1ull));
                if (need && rank < rqCount) {
                    const lds_int *q = rq + rqHead + rank;
                    path = q[0 * kVmStride];
                    const int pm = q[1 * kVmStride];
                    medium = pm & 0xffff; sampleMode = (pm & 0x10000) != 0; camBit = (pm >> 17) & 1; realState = (pm >> 18) & 3; wasParked = (pm >> 20) & 1;
                    index = (uint32_t)q[2 * kVmStride]; dim = q[3 * kVmStride]; cachedDim = -1;
                    o = V3(__int_as_float(q[4 * kVmStride]), __int_as_float(q[5 * kVmStride]), __int_as_float(q[6 * kVmStride]));
                    d = V3(__int_as_float(q[7 * kVmStride]), __int_as_float(q[8 * kVmStride]), __int_as_float(q[9 * kVmStride]));
                    t = __int_as_float(q[10 * kVmStride]); tMax = __int_as_float(q[11 * kVmStride]); Tr = __int_as_float(q[12 * kVmStride]); nsteps = 0;
                    const DMedium &m = mt.media[medium];
                    nx = m.nx; ny = m.ny; nz = m.nz;
                    invMaxDensity = m.inv_max_density;
                    stepScale = m.sigma_t;
                    dens = mt.density + m.density_offset;
                }
                const unsigned tk = min(rqCount, (unsigned)__popcll(needMask));
                rqHead += tk; rqCount -= tk;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");       // all reads of the queue precede the next batch's writes
            }
        }
        if (__ballot(path >= 0) == 0) {
            if (exhausted && poolCount == 0 && rqCount == 0) break;
            continue;
        }
        // ---------------- one tracking iteration for every active lane ----------------
        if (path >= 0) {
            if (COUNT) ++cntSteps;
            bool done = false;
            float resW = 1.f, resT = -1.f;
            bool resSigma = false;
            // samples are drawn in pairs (dimension dim and dim + 1 together): the second one is the next draw of this loop
            auto draw = [&]() -> float {
                if (dim == cachedDim) { ++dim; return cachedU; }
                float ua, ub;
                halton_sample_pair(sc.st, index, dim, &ua, &ub);
                cachedU = ub; cachedDim = ++dim;
                return ua;
            };
            t -= gx_log(1 - draw()) * invMaxDensity / stepScale;
            if (t >= tMax) { done = true; resW = sampleMode ? 1.f : Tr; }
            else {
                // GridDensityMedium::Density, GridDensityMedium.cpp:14-29
                V3 pp = o + d * t;
                V3 ps(pp.x * (float)nx - .5f, pp.y * (float)ny - .5f, pp.z * (float)nz - .5f);
                int px = (int)floorf(ps.x), py = (int)floorf(ps.y), pz = (int)floorf(ps.z);
                V3 dl = ps - V3((float)px, (float)py, (float)pz);
                auto D = [&](int x, int y, int z) -> float {
                    if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return 0.f;
                    return dens[(z * ny + y) * nx + x];
                };
                float d00 = lerpf(dl.x, D(px, py, pz), D(px + 1, py, pz));
                float d10 = lerpf(dl.x, D(px, py + 1, pz), D(px + 1, py + 1, pz));
                float d01 = lerpf(dl.x, D(px, py, pz + 1), D(px + 1, py, pz + 1));
                float d11 = lerpf(dl.x, D(px, py + 1, pz + 1), D(px + 1, py + 1, pz + 1));
                float density = lerpf(dl.z, lerpf(dl.y, d00, d10), lerpf(dl.y, d01, d11));
                if (sampleMode) {
                    if (density * invMaxDensity > draw()) { done = true; resSigma = true; resT = t; }
                } else {
                    Tr *= 1 - fmaxf(0.f, density * invMaxDensity);
                    const float rrThreshold = .1f;
                    if (Tr < rrThreshold) {
                        float q = fmaxf(.05f, 1 - Tr);
                        if (draw() < q) { done = true; resW = 0.f; }
                        else Tr /= 1 - q;
                    }
                }
            }
            if (done) {
                float4 res = make_float4(resW, resW, resW, resT);
                if (resSigma) {
                    const DMedium &m = mt.media[medium];
                    Spec w = spec3(m.sigma_s) / m.sigma_t;
                    res = make_float4(w.r, w.g, w.b, resT);
                }
                va.mres[path] = res;
                reinterpret_cast<int *>(&va.vs[path])[1] = dim;
                if (wasParked) { reinterpret_cast<int *>(&va.vs[path])[3] = camBit; va.state[path] = (unsigned char)realState; }   // back among the living
                path = -1;
            } else if (step_cap > 0 && ++nsteps >= step_cap) {
                // Step cap: a launch ends with its longest segment (0 ... ~1000 steps of ~2.4 us each once the chip has drained: 0.6 ms per launch,
                // a quarter of the kernel's time over the ~40 launches of a pass), so a segment that is still running after step_cap steps parks
                // its state; k_vol_step leaves the path alone this round, the next round re-traces the same ray and comes back here.
                va.mres[path] = make_float4(Tr, t, 0.f, 0.f);
                reinterpret_cast<int *>(&va.vs[path])[1] = dim;   // (a pending second half of a pair is simply drawn again: same dimension, same value)
                reinterpret_cast<int *>(&va.vs[path])[3] = camBit | kVsCont;
                va.state[path] = (unsigned char)VS_PARKED;   // no k_vol_step bin takes it this round
                pa.pflags[path] = (unsigned char)(1 | 2);    // alive, ray inside a medium: k_vol_step, which writes the flags of the paths it handles, will not see this one
                                                             // (and k_trace4 clears the flags of a ray that leaves a scene without infinite lights)
                ++cntCont;
                path = -1;
            }
        }
    }
    if (COUNT && cntSteps) atomicAdd(&ctr->media_steps, cntSteps);
    for (int o = 32; o > 0; o >>= 1) cntCont += __shfl_xor(cntCont, o);
    if (cntCont && lane == 0) atomicAdd(&ctr->media_cont, (unsigned long long)cntCont);
}

// LM: the lobe set every material of the scene fits in (device_bsdf.h LM_*): a scene of Matte walls and media runs the
// diffuse-only instantiation, which needs far fewer registers than the Disney-capable one.
// ST: the state every path of `queue` is in (the host bins the live paths by state before each step, so a wave runs one
// of the three phases instead of all of them in turn); `n_dev` is the bin's fill count written by the binning.
// TEX: the scene has image-textured materials.  VolPathIntegrator keeps the camera RayDifferential (`RayDifferential ray(r)`,
// VolPathIntegrator.cpp:30) until the ray is replaced (pass-through, medium or surface scattering all assign a plain SpawnRay), so
// a surface vertex reached by the camera ray itself filters its textures with the camera differentials (vs.w).
// minimum waves per SIMD (tuning knobs, tools/build_variant.sh).  The kernel waits on dependent gathers for 62 % of its wave cycles, so a third wave
// is worth the 22 dwords the MAIN state of the diffuse class then spills (198 -> 168 registers): k_vol_step + compaction -5 ... -9 %, cfg 5 -2 ... -4 %;
// a fourth wave for the segment states (152 -> 128) changes nothing (profiles/r03_ab_vol_step_occupancy_cfg5.log).  The glossy / Disney instantiations
// (235 - 256 registers at two waves) take three as well: VolPath on the cfg 3 / cfg 4 scenes +4.4 % / +5.5 %; the textured ones too (+2.5 % on the
// image-textured Cornell box, profiles/r03_ab_textured_occupancy.log).
#ifndef GX_VOL_W_MAIN
#define GX_VOL_W_MAIN 3
#endif
#ifndef GX_VOL_W_SEG
#define GX_VOL_W_SEG 2
#endif
#ifndef GX_VOL_W_OTHER
#define GX_VOL_W_OTHER 3
#endif
template <uint32_t LM, int ST, bool TEX> constexpr int vol_min_waves() { return (LM == LM_DIFFUSE && !TEX) ? (ST == VS_MAIN ? GX_VOL_W_MAIN : GX_VOL_W_SEG) : GX_VOL_W_OTHER; }
template <uint32_t LM, int LT, int ST, bool TEX = false>
__global__ void __launch_bounds__(kBlock, (vol_min_waves<LM, ST, TEX>())) k_vol_step(DScene sc, DMediaTables mt, DRender r, PathArrays pa, VolArrays va, const int *__restrict__ queue, const unsigned int *n_dev, int lds_mats, int lds_lights) {
    extern __shared__ int vstep_smem[];   // the scene's DMaterial[] | DLight[] when they are small (as in k_shade: dependent gathers along the BSDF code become LDS reads)
    const int n = (int)*n_dev;
    if (blockIdx.x * blockDim.x >= (unsigned)n) return;
    const DMaterial *mats = sc.materials;
    DLightTables ltab = sc.lt;
    {
        int *dst = vstep_smem;
        if (!TEX && lds_mats > 0) {
            const int *src = reinterpret_cast<const int *>(sc.materials);
            const int nd = lds_mats * (int)(sizeof(DMaterial) / 4);
            for (int k = threadIdx.x; k < nd; k += kBlock) dst[k] = src[k];
            mats = reinterpret_cast<const DMaterial *>(dst);
            dst += nd;
        }
        if (lds_lights > 0) {
            const int *src = reinterpret_cast<const int *>(sc.lt.lights);
            const int nd = lds_lights * (int)(sizeof(DLight) / 4);
            for (int k = threadIdx.x; k < nd; k += kBlock) dst[k] = src[k];
            ltab.lights = reinterpret_cast<const DLight *>(dst);
        }
    }
    __syncthreads();
    // (fetching the queue entry two iterations and the hit one iteration ahead, as k_shade does, was measured here and lost: +7 % step time)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int path = queue[i];
        int4 vs = va.vs[path];
        vs.x = ST;   // == the stored state
        uint2 meta = pa.meta[(size_t)path * kRSm];
        const uint32_t index = meta.x;
        int bounces = (int)((meta.y >> 16) & 0xffu);
        bool specularBounce = (meta.y >> 31) != 0;
        SampleStream ss(sc.st, index, vs.y);
        float4 o4 = pa.ray_o[(size_t)path * kRS], d4 = pa.ray_d[(size_t)path * kRS];
        V3 ro(o4.x, o4.y, o4.z), rd(d4.x, d4.y, d4.z);
        int rayMedium = __float_as_int(d4.w);
        const int leaf = pa.hit[path];
        bool survive = false;      // the path has a ray in flight after this step
        bool vertexNew = false;    // phase 2 established a vertex at this step
        bool vertexDone = false;   // the light estimate of the saved vertex is complete
        float miT = -1.f;
        Spec beta, L;
        float etaScale = 1;

        // the traced ray's hit, shared by all three states
        bool found = leaf != -1;
        V3 p0, p1, p2;
        int triMat = -1, triLight = -1;
        TriHit h;
        if (leaf < -1) {   // sphere
            const DSphere &sph = sc.spheres[-2 - leaf];
            triMat = sph.material;
            found = sphere_test(sph, ro, rd, o4.w, &h.t);
        } else if (found) {
            const float4 *q = reinterpret_cast<const float4 *>(sc.tris + leaf);
            float4 a = q[0], b = q[1], c = q[2];
            p0 = V3(a.x, a.y, a.z); p1 = V3(b.x, b.y, b.z); p2 = V3(c.x, c.y, c.z);
            triMat = __float_as_int(b.w); triLight = __float_as_int(c.w);
            found = tri_test(p0, p1, p2, ro, rd, o4.w, &h);   // same arithmetic as the traversal
        }
        // hit point / error bound / normal of the traced ray's hit (no Bump: only p, pError, n are used)
        auto hit_geometry = [&]() {
            if (leaf < -1) return sphere_surface_point(sc.spheres[-2 - leaf], ro, rd, h.t, false);
            if (TEX) { V3 dndu, dndv; return surface_point_tables(tex_tables(sc.materials), leaf, p0, p1, p2, h, false, &dndu, &dndv); }   // n follows the shading normals
            return surface_point(p0, p1, p2, h, false);
        };

        if (ST != VS_MAIN) {
            // ---------------- phase 1: one segment of the light-sample ray or of the scattering ray ----------------
            float4 T4 = va.Tr[path], Ld4 = va.Ld[path], f4 = va.f[path];
            Spec Tr(T4.x, T4.y, T4.z), Ld(Ld4.x, Ld4.y, Ld4.z);
            const int nflags = __float_as_int(f4.w);
            SurfacePoint sp;
            sp.valid = false;
            if (found) { sp = hit_geometry(); found = sp.valid; }
            bool segDone = false;
            int segMedium = -1;        // medium of the next segment's ray
            if (ST == VS_SHADOW) {   // VisibilityTester::Tr, Light.cpp:33-53
                if (found && triMat >= 0) {
                    segDone = true;    // blocked: Tr = 0, Li becomes black, nothing is added
                } else {
                    if (rayMedium >= 0) { float4 w = va.mres[path]; Tr = Tr * Spec(w.x, w.y, w.z); }
                    if (!found) {
                        segDone = true;
                        float4 Li4 = va.Li[path];
                        Spec Li = Spec(Li4.x, Li4.y, Li4.z) * Tr;
                        if (!Li.is_black()) Ld = Ld + Spec(f4.x, f4.y, f4.z) * Li * va.n1[path].w / va.p1e[(size_t)path * kRS].w;
                    } else {           // ray = isect.SpawnRayTo(p1)
                        float4 q1 = va.p1[(size_t)path * kRS], q1e = va.p1e[(size_t)path * kRS], qn1 = va.n1[path];
                        V3 so, sd;
                        spawn_ray_to(sp.p, sp.pError, sp.n, V3(q1.x, q1.y, q1.z), V3(q1e.x, q1e.y, q1e.z), V3(qn1.x, qn1.y, qn1.z), &so, &sd);
                        segMedium = hit_medium(sc, mt, leaf, rayMedium, sp.n, sd);
                        pa.ray_o[(size_t)path * kRS] = make_float4(so.x, so.y, so.z, 1 - GX_SHADOW_EPS);
                        pa.ray_d[(size_t)path * kRS] = make_float4(sd.x, sd.y, sd.z, __int_as_float(segMedium));
                    }
                }
                if (segDone && (nflags & 2)) {   // go on with the scattering ray: it.SpawnRay(wi), Integrator.cpp:193
                    float4 mo = va.mis_o[(size_t)path * kRS], md = va.mis_d[(size_t)path * kRS];
                    pa.ray_o[(size_t)path * kRS] = make_float4(mo.x, mo.y, mo.z, GX_INF);
                    pa.ray_d[(size_t)path * kRS] = make_float4(md.x, md.y, md.z, mo.w);
                    segMedium = __float_as_int(mo.w);
                    Tr = Spec(1.f);
                    vs.x = VS_MIS;
                    segDone = false;
                }
            } else {                   // Scene::IntersectTr, Scene.cpp:26-40
                if (rayMedium >= 0) { float4 w = va.mres[path]; Tr = Tr * Spec(w.x, w.y, w.z); }
                if (found && triMat < 0) {   // ray = isect->SpawnRay(ray.d)
                    V3 o2 = offset_ray_origin(sp.p, sp.pError, sp.n, rd);
                    segMedium = hit_medium(sc, mt, leaf, rayMedium, sp.n, rd);
                    pa.ray_o[(size_t)path * kRS] = make_float4(o2.x, o2.y, o2.z, GX_INF);
                    pa.ray_d[(size_t)path * kRS] = make_float4(rd.x, rd.y, rd.z, __int_as_float(segMedium));
                } else {
                    segDone = true;
                    const int expect = __float_as_int(va.Li[path].w);
                    bool ok = found ? (leaf == expect) : (expect < 0);
                    if (ok && (nflags & 4)) {
                        float4 Y4 = va.mis_Y[path];
                        Ld = Ld + Spec(Y4.x, Y4.y, Y4.z) * Tr * Ld4.w / va.mis_d[(size_t)path * kRS].w;
                    }
                }
            }
            if (!segDone) {
                va.Tr[path] = make_float4(Tr.r, Tr.g, Tr.b, T4.w);
                va.Ld[path] = make_float4(Ld.r, Ld.g, Ld.b, Ld4.w);
                vs.y = ss.dim;
                va.vs[path] = vs; va.state[path] = (unsigned char)vs.x;
                pa.pflags[path] = (unsigned char)(1 | (segMedium >= 0 ? 2 : 0));
                continue;
            }
            // light estimate complete: L += beta * (Ld / lightPdf), Integrator.cpp:78 + VolPathIntegrator.cpp:55/99
            float4 b4 = pa.beta[(size_t)path * kRS], L4 = pa.L[path];
            beta = Spec(b4.x, b4.y, b4.z); etaScale = b4.w;
            L = Spec(L4.x, L4.y, L4.z) + beta * (Ld / va.p1[(size_t)path * kRS].w);
            vertexDone = true;
            // re-establish the vertex from its saved main ray
            o4 = va.sv_o[(size_t)path * kRS]; d4 = va.sv_d[(size_t)path * kRS];
            ro = V3(o4.x, o4.y, o4.z); rd = V3(d4.x, d4.y, d4.z);
            rayMedium = __float_as_int(d4.w);
            miT = T4.w;
        } else {
            // ---------------- phase 2: the main ray, VolPathIntegrator.cpp:36-80 ----------------
            float4 b4 = pa.beta[(size_t)path * kRS], L4 = pa.L[path];
            beta = Spec(b4.x, b4.y, b4.z); etaScale = b4.w;
            L = Spec(L4.x, L4.y, L4.z);
            SurfacePoint sp0;
            sp0.valid = false;
            if (found) { sp0 = hit_geometry(); found = sp0.valid; }
            bool miValid = false;
            if (rayMedium >= 0) {   // beta *= ray.medium->Sample(ray, sampler, arena, &mi), evaluated by k_vol_media
                float4 w = va.mres[path];
                beta = beta * Spec(w.x, w.y, w.z);
                miValid = w.w >= 0.f;
                miT = w.w;
            }
            bool alive = !beta.is_black();
            if (alive && miValid) {
                if (bounces >= r.max_depth) alive = false;
                else vertexNew = true;
            } else if (alive) {
                miT = -1.f;
                if (bounces == 0 || specularBounce) {
                    if (found) {
                        if (triLight >= 0) L = L + beta * area_L(ltab.lights[triLight], sp0.n, -rd);
                    } else {
                        for (int k = 0; k < ltab.n_infinite; ++k) L = L + beta * light_Le<LT>(ltab, ltab.infinite[k], ro, rd);
                    }
                }
                if (!found || bounces >= r.max_depth) alive = false;
                else if (triMat < 0) {
                    // no BSDF: ray = isect.SpawnRay(ray.d); bounces--; continue  (VolPathIntegrator.cpp:88-92)
                    V3 o2 = offset_ray_origin(sp0.p, sp0.pError, sp0.n, rd);
                    const int nm = hit_medium(sc, mt, leaf, rayMedium, sp0.n, rd);
                    pa.ray_o[(size_t)path * kRS] = make_float4(o2.x, o2.y, o2.z, GX_INF);
                    pa.ray_d[(size_t)path * kRS] = make_float4(rd.x, rd.y, rd.z, __int_as_float(nm));
                    pa.beta[(size_t)path * kRS] = make_float4(beta.r, beta.g, beta.b, etaScale);
                    pa.L[path] = make_float4(L.r, L.g, L.b, 0.f);
                    vs.y = ss.dim;
                    vs.w = 0;   // isect.SpawnRay(ray.d): a plain Ray
                    va.vs[path] = vs; va.state[path] = (unsigned char)vs.x;
                    pa.pflags[path] = (unsigned char)(1 | (nm >= 0 ? 2 : 0));
                    continue;
                } else vertexNew = true;
            }
            if (!alive) {
                va.Lout[va.orig[path]] = make_float4(L.r, L.g, L.b, 0.f);
                pa.pflags[path] = 0;
                continue;
            }
        }

        // ---------------- phase 3: the vertex (fresh, or re-established after its light estimate) ----------------
        const bool isMedium = miT >= 0.f;
        int vleaf = vertexDone ? vs.z : leaf;
        SurfacePoint sp;
        sp.valid = true;
        const DMaterial *mat = nullptr;
        DMaterial tm;
        Bsdf<LM> bsdf;
        V3 itP, itPError, itN;
        int medIn = rayMedium, medOut = rayMedium;
        float g = 0;
        if (isMedium) {
            itP = ro + rd * miT;    // mi.p = ray(t)
            g = mt.media[rayMedium].g;
        } else {
            if (vleaf < -1) {
                const DSphere &sph = sc.spheres[-2 - vleaf];
                triMat = sph.material;
                (void)sphere_test(sph, ro, rd, o4.w, &h.t);
                mat = mats + triMat;
                sp = sphere_surface_point(sph, ro, rd, h.t, mat->has_bump != 0);
            } else {
                const float4 *q = reinterpret_cast<const float4 *>(sc.tris + vleaf);
                float4 a = q[0], b = q[1], c = q[2];
                p0 = V3(a.x, a.y, a.z); p1 = V3(b.x, b.y, b.z); p2 = V3(c.x, c.y, c.z);
                triMat = __float_as_int(b.w);
                (void)tri_test(p0, p1, p2, ro, rd, o4.w, &h);
                mat = mats + triMat;
                sp = surface_point(p0, p1, p2, h, mat->has_bump != 0);
                if (TEX) { V3 dndu, dndv; sp = surface_point_tables(tex_tables(sc.materials), vleaf, p0, p1, p2, h, mat->has_bump != 0, &dndu, &dndv); }
                if (TEX && (mat->kd_tex | mat->ks_tex)) {
                    float tu, tv;
                    V3 dpdu, dpdv;
                    tri_uv_frame(p0, p1, p2, h, tri_uvs(tex_tables(sc.materials), vleaf), &tu, &tv, &dpdu, &dpdv);
                    RayDiff rdf;
                    rdf.has = false;
                    if (vs.w & 1) {
                        int px, py;
                        local_pixel(r, va.orig[path] % r.npix, &px, &py);
                        rdf = camera_ray_diff(r.cam, sc.st, px, py, pa.meta[(size_t)path * kRSm].x, r.spp);
                    }
                    textured_material(tex_tables(sc.materials), *mat, tu, tv, compute_differentials(rdf, sp.p, sp.n, dpdu, dpdv), &tm);
                    mat = &tm;
                }
            }
            bsdf.mat = mat; bsdf.ns = sp.ns; bsdf.ng = sp.n; bsdf.ss = sp.ss; bsdf.ts = sp.ts;
            itP = sp.p; itPError = sp.pError; itN = sp.n;
            hit_interface(sc, mt, vleaf, rayMedium, &medIn, &medOut);
        }
        const V3 woN = normalize(-rd);   // SurfaceInteraction::wo
        const V3 woM = -rd;              // MediumInteraction::wo

        if (vertexNew && ltab.n_lights > 0) {
            // ---- UniformSampleOneLight + EstimateDirect(handleMedia = true), Integrator.cpp:57-79, 93-210
            float lightPdfSel;
            int lightNum = light_select(ltab, itP, ss.get1d(), &lightPdfSel);
            if (lightPdfSel != 0) {
                float ul0, ul1, us0, us1;
                ss.get2d(&ul0, &ul1);
                ss.get2d(&us0, &us1);
                const int bsdfFlags = BSDF_ALL & ~BSDF_SPECULAR;
                int nflags = 0;
                V3 so, sd, mo, wi2;
                Spec fX(0.f), Y(0.f);
                float weightX = 0, weightY = 0, scatPdf2 = 0;
                int expect = -1, shMedium = -1, misMedium = -1;
                LightSample ls = light_sample<LT>(ltab, lightNum, itP, ul0, ul1);
                if (ls.pdf > 0 && !ls.Li.is_black()) {
                    float scatteringPdf;
                    if (isMedium) {
                        float p = phase_hg(dot(woM, ls.wi), g);
                        fX = Spec(p); scatteringPdf = p;
                    } else {
                        fX = bsdf.f(woN, ls.wi, bsdfFlags) * absdot(ls.wi, sp.ns);
                        scatteringPdf = bsdf.pdf(woN, ls.wi, bsdfFlags);
                    }
                    if (!fX.is_black()) {
                        spawn_ray_to(itP, itPError, itN, ls.p1, ls.p1Error, ls.n1, &so, &sd);
                        shMedium = dot(sd, itN) > 0 ? medOut : medIn;
                        weightX = light_is_delta<LT>(ltab.lights[lightNum]) ? 1.f : power_heuristic(ls.pdf, scatteringPdf);   // IsDeltaLight: no MIS weight
                        nflags |= 1;
                    }
                }
                if (!light_is_delta<LT>(ltab.lights[lightNum])) {   // ... and no scattering-sample half (Integrator.cpp:168)
                    Spec f2;
                    bool sampledSpecular = false;
                    if (isMedium) {
                        float p = hg_sample_p(g, woM, &wi2, us0, us1);
                        f2 = Spec(p); scatPdf2 = p;
                    } else {
                        int sampledType;
                        f2 = bsdf.sample_f(woN, &wi2, us0, us1, &scatPdf2, bsdfFlags, &sampledType);
                        f2 = f2 * absdot(wi2, sp.ns);
                        sampledSpecular = (sampledType & BSDF_SPECULAR) != 0;
                    }
                    if (!f2.is_black() && scatPdf2 > 0) {
                        weightY = 1;
                        bool skip = false;
                        if (!sampledSpecular) {
                            float lightPdf = light_pdf<LT>(ltab, lightNum, itP, itPError, itN, wi2);
                            if (lightPdf == 0) skip = true;
                            else weightY = power_heuristic(scatPdf2, lightPdf);
                        }
                        if (!skip) {
                            const DLight &lt = ltab.lights[lightNum];
                            mo = offset_ray_origin(itP, itPError, itN, wi2);
                            misMedium = dot(wi2, itN) > 0 ? medOut : medIn;
                            Spec Li2;
                            if (LT == LT_AREA || lt.type == GNXR_LIGHT_AREA_TRI) {
                                V3 lp0(lt.p0[0], lt.p0[1], lt.p0[2]), lp1(lt.p1[0], lt.p1[1], lt.p1[2]), lp2(lt.p2[0], lt.p2[1], lt.p2[2]);
                                V3 ln = normalize(cross(lp0 - lp2, lp1 - lp2));
                                Li2 = area_L(lt, ln, -wi2);
                                expect = lt.tri_leaf;
                            } else {
                                Li2 = light_Le<LT>(ltab, lightNum, mo, wi2);
                                expect = -1;
                            }
                            if (!Li2.is_black()) { Y = f2 * Li2; nflags |= 4; }
                            nflags |= 2;
                        }
                    }
                }
                if (nflags & 3) {
                    va.sv_o[(size_t)path * kRS] = o4;
                    va.sv_d[(size_t)path * kRS] = d4;
                    va.p1[(size_t)path * kRS] = make_float4(ls.p1.x, ls.p1.y, ls.p1.z, lightPdfSel);
                    va.p1e[(size_t)path * kRS] = make_float4(ls.p1Error.x, ls.p1Error.y, ls.p1Error.z, ls.pdf);
                    va.n1[path] = make_float4(ls.n1.x, ls.n1.y, ls.n1.z, weightX);
                    va.f[path] = make_float4(fX.r, fX.g, fX.b, __int_as_float(nflags));
                    va.Li[path] = make_float4(ls.Li.r, ls.Li.g, ls.Li.b, __int_as_float(expect));
                    va.Tr[path] = make_float4(1.f, 1.f, 1.f, miT);
                    va.Ld[path] = make_float4(0.f, 0.f, 0.f, weightY);
                    if (nflags & 2) {
                        va.mis_o[(size_t)path * kRS] = make_float4(mo.x, mo.y, mo.z, __int_as_float(misMedium));
                        va.mis_d[(size_t)path * kRS] = make_float4(wi2.x, wi2.y, wi2.z, scatPdf2);
                        va.mis_Y[path] = make_float4(Y.r, Y.g, Y.b, 0.f);
                    }
                    if (nflags & 1) {
                        pa.ray_o[(size_t)path * kRS] = make_float4(so.x, so.y, so.z, 1 - GX_SHADOW_EPS);
                        pa.ray_d[(size_t)path * kRS] = make_float4(sd.x, sd.y, sd.z, __int_as_float(shMedium));
                        vs.x = VS_SHADOW;
                    } else {
                        pa.ray_o[(size_t)path * kRS] = make_float4(mo.x, mo.y, mo.z, GX_INF);
                        pa.ray_d[(size_t)path * kRS] = make_float4(wi2.x, wi2.y, wi2.z, __int_as_float(misMedium));
                        vs.x = VS_MIS;
                    }
                    vs.y = ss.dim;
                    vs.z = leaf;
                    va.vs[path] = vs; va.state[path] = (unsigned char)vs.x;
                    pa.beta[(size_t)path * kRS] = make_float4(beta.r, beta.g, beta.b, etaScale);
                    pa.L[path] = make_float4(L.r, L.g, L.b, 0.f);
                    pa.pflags[path] = (unsigned char)(1 | (((nflags & 1) ? shMedium : misMedium) >= 0 ? 2 : 0));
                    continue;
                }
            }
        }

        // ---- continuation: phase function / BSDF sampling, VolPathIntegrator.cpp:58-64, 104-127; Russian roulette 131-140
        int nextMedium = -1;
        {
            V3 wo = -rd, wi, o2;
            float u0, u1;
            ss.get2d(&u0, &u1);
            bool ok = true;
            if (isMedium) {
                (void)hg_sample_p(g, wo, &wi, u0, u1);
                o2 = itP;                       // mi.SpawnRay(wi): n == 0, so the origin is not offset
                nextMedium = rayMedium;
                specularBounce = false;
            } else {
                float pdf;
                int flags;
                Spec f = bsdf.sample_f(wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
                if (f.is_black() || pdf == 0.f) ok = false;
                else {
                    beta = beta * (f * absdot(wi, sp.ns) / pdf);
                    specularBounce = (flags & BSDF_SPECULAR) != 0;
                    if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                        float eta = mat->eta;
                        etaScale *= (dot(wo, sp.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                    }
                    o2 = offset_ray_origin(sp.p, sp.pError, sp.n, wi);
                    nextMedium = dot(wi, sp.n) > 0 ? medOut : medIn;
                }
            }
            if (ok) {
                Spec rrBeta = beta * etaScale;
                survive = true;
                if (rrBeta.max_value() < r.rr_threshold && bounces > 3) {
                    float q = fmaxf(.05f, 1 - rrBeta.max_value());
                    if (ss.get1d() < q) survive = false;
                    else beta = beta / (1 - q);
                }
                if (survive) {
                    pa.ray_o[(size_t)path * kRS] = make_float4(o2.x, o2.y, o2.z, GX_INF);
                    pa.ray_d[(size_t)path * kRS] = make_float4(wi.x, wi.y, wi.z, __int_as_float(nextMedium));
                    pa.beta[(size_t)path * kRS] = make_float4(beta.r, beta.g, beta.b, etaScale);
                    pa.store_meta(path, index, ((uint32_t)(bounces + 1) << 16) | (specularBounce ? 0x80000000u : 0u));
                    vs.x = VS_MAIN;
                    vs.y = ss.dim;
                    vs.w = 0;   // mi.SpawnRay(wi) / isect.SpawnRay(wi): a plain Ray
                    va.vs[path] = vs; va.state[path] = (unsigned char)vs.x;
                }
            }
        }
        if (survive) pa.L[path] = make_float4(L.r, L.g, L.b, 0.f);
        else va.Lout[va.orig[path]] = make_float4(L.r, L.g, L.b, 0.f);
        pa.pflags[path] = (unsigned char)(survive ? (1 | (nextMedium >= 0 ? 2 : 0)) : 0);
    }
}

// ---- packing.  A VolPath pass runs ~60 rounds over a population that thins out all the time (cfg 5: on average 15 % of the slots hold a
// live path), and every per-path array is then read one 16-byte element per 64-byte line: 470 - 740 B of HBM traffic per path and round
// against ~210 B of state (profiles/traffic_latest_cfg5.json, round 2).  When the survivors have dropped to half the span they are spread
// over, their state is copied to the front of a second set of arrays (queue order == slot order is kept, so everything stays sorted and
// coalesced), the queues are renumbered, and the rounds go on densely.  A path's results go to its ORIGINAL slot (`orig`, `Lout`).
constexpr int kVolPackF4 = 9;    // float4-sized per-path arrays that carry state from round to round
struct VolPackSet {
    float4 *f4[kVolPackF4];      // L | vs n1 f Li Tr Ld mis_Y mres
    float4 *rec[kRecGroups];     // the record groups of a slot (PathArrays): {ray_o ray_d} {beta meta} {sv_o sv_d} {p1 p1e} {mis_o mis_d}
    unsigned char *state;
    int *orig;
};
static __global__ void __launch_bounds__(kBlock) k_vol_pack(const int *__restrict__ queue, int n, VolPackSet src, VolPackSet dst, int *__restrict__ newslot) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int p = queue ? queue[i] : i;
#pragma unroll
        for (int k = 0; k < kVolPackF4; ++k) dst.f4[k][i] = src.f4[k][p];
#pragma unroll
        for (int k = 0; k < kRecGroups; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e) dst.rec[k][(size_t)i * kRS + e] = src.rec[k][(size_t)p * kRS + e];
        dst.state[i] = src.state[p];
        dst.orig[i] = src.orig[p];
        newslot[p] = i;
    }
}
static __global__ void __launch_bounds__(kBlock) k_vol_remap(int *__restrict__ queue, int n, const int *__restrict__ newslot) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) queue[i] = newslot[queue[i]];
}

}  // namespace gnxr

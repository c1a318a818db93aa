// scene_compile.cpp -- the host scene compiler: gnxr_scene_desc -> flat device tables.
//
//   BVH            SAH build + DFS flatten, same split decisions as accelerator/BVHAccel.cpp:201-367,628-646
//   materials      BSDF lobe lists precomputed per material (materials/*.cpp ComputeScatteringFunctions):
//                  every texture on the path is a ConstantTexture, so the lobes do not depend on the hit
//   lights         per-light triangle data (lights/DiffuseAreaLight.cpp, shape/Triangle.cpp:455-492)
//   sampler        Primes / PrimeSums / permutations (samplers/LowDiscrepancy.cpp:9,93,2459-2473)
//   camera         camera/Perspective.cpp:114-135 + core/Camera.h:54-75
//   env light      MIPMap level 0 + Distribution2D (lights/InfiniteAreaLight.cpp:12-82, core/MIPMap.h:85-198)
// Pure host code.
#include <cstdio>
#include <thread>

#include "host_scene.h"

namespace gnxr {

// ------------------------------------------------------------------ BVH
namespace {
struct PrimInfo { int prim; Box3 b; Vec3 c; };
struct BuildNode { Box3 b; int child[2]; int axis, first, n; };

struct BvhBuilder {
    std::vector<PrimInfo> info;
    std::vector<BuildNode> nodes;
    std::vector<int> ordered;
    int method = GNXR_BVH_SAH;   // SAH, Middle or EqualCounts for build(); HLBVH has build_hlbvh()

    int build(int start, int end) {
        int me = (int)nodes.size();
        nodes.push_back(BuildNode());
        Box3 bounds;
        for (int i = start; i < end; ++i) bounds.grow(info[i].b);
        int n = end - start;
        auto leaf = [&]() {
            nodes[me].b = bounds; nodes[me].first = (int)ordered.size(); nodes[me].n = n;
            nodes[me].child[0] = nodes[me].child[1] = -1;
            for (int i = start; i < end; ++i) ordered.push_back(info[i].prim);
            return me;
        };
        if (n == 1) return leaf();
        Box3 cb;
        for (int i = start; i < end; ++i) cb.grow(info[i].c);
        int dim = cb.max_extent();
        int mid = (start + end) / 2;
        if (cb.hi[dim] == cb.lo[dim]) return leaf();
        bool partitioned = false;
        if (method == GNXR_BVH_MIDDLE) {   // BVHAccel.cpp:243-258; falls through to EqualCounts when everything lands on one side
            float pmid = (cb.lo[dim] + cb.hi[dim]) / 2;
            PrimInfo *midPtr = std::partition(&info[start], &info[end - 1] + 1, [dim, pmid](const PrimInfo &pi) { return pi.c[dim] < pmid; });
            mid = (int)(midPtr - &info[0]);
            partitioned = mid != start && mid != end;
            if (!partitioned) mid = (start + end) / 2;
        }
        if (partitioned) {
        } else if (n <= 2 || method == GNXR_BVH_MIDDLE || method == GNXR_BVH_EQUAL_COUNTS) {   // EqualCounts, BVHAccel.cpp:259-268 (and SAH with <= 2 primitives, :273-282)
            std::nth_element(&info[start], &info[mid], &info[end - 1] + 1,
                             [dim](const PrimInfo &a, const PrimInfo &b) { return a.c[dim] < b.c[dim]; });
        } else {
            constexpr int NB = 12;
            int count[NB] = {0};
            Box3 bb[NB];
            for (int i = start; i < end; ++i) {
                int b = NB * cb.offset(info[i].c)[dim];
                if (b == NB) b = NB - 1;
                count[b]++;
                bb[b].grow(info[i].b);
            }
            float cost[NB - 1];
            for (int i = 0; i < NB - 1; ++i) {
                Box3 b0, b1;
                int c0 = 0, c1 = 0;
                for (int j = 0; j <= i; ++j) { b0.grow(bb[j]); c0 += count[j]; }
                for (int j = i + 1; j < NB; ++j) { b1.grow(bb[j]); c1 += count[j]; }
                cost[i] = 1 + (c0 * b0.area() + c1 * b1.area()) / bounds.area();
            }
            float minCost = cost[0];
            int split = 0;
            for (int i = 1; i < NB - 1; ++i) if (cost[i] < minCost) { minCost = cost[i]; split = i; }
            // maxPrimsInNode = 1 (RenderThread.cpp:155): always split when n > 1
            PrimInfo *pmid = std::partition(&info[start], &info[end - 1] + 1, [=](const PrimInfo &pi) {
                int b = NB * cb.offset(pi.c)[dim];
                if (b == NB) b = NB - 1;
                return b <= split;
            });
            mid = (int)(pmid - &info[0]);
        }
        // `node->InitInterior(dim, recursiveBuild(.., start, mid, ..), recursiveBuild(.., mid, end, ..))` (BVHAccel.cpp:355-360): g++
        // evaluates the arguments right to left, so the SECOND half is built first and its primitives come first in orderedPrims
        int c1 = build(mid, end);
        int c0 = build(start, mid);
        nodes[me].child[0] = c0; nodes[me].child[1] = c1;
        Box3 u = nodes[c0].b; u.grow(nodes[c1].b);
        nodes[me].b = u; nodes[me].axis = dim; nodes[me].n = 0;
        return me;
    }

    // ---- HLBVHBuild, accelerator/BVHAccel.cpp:369-626 (maxPrimsInNode = 1): built on the device (hlbvh_build.hip.h); `info` stays in
    // primitive order here.  Where the reference's CHECKs would fire (coincident treelet centroids: its recursion would not terminate)
    // or a leaf exceeds LinearBVHNode's 16-bit primitive count the device stage reports failure.
    int build_hlbvh(HlbvhBuildFn device_build) {
        const int n = (int)info.size();
        Box3 cb;   // bounds of the centroids, BVHAccel.cpp:372-375
        std::vector<float> cen(3 * (size_t)n), pb(6 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            cb.grow(info[i].c);
            cen[3 * (size_t)i] = info[i].c.x; cen[3 * (size_t)i + 1] = info[i].c.y; cen[3 * (size_t)i + 2] = info[i].c.z;
            const Box3 &b = info[i].b;
            float *q = &pb[6 * (size_t)i];
            q[0] = b.lo.x; q[1] = b.lo.y; q[2] = b.lo.z; q[3] = b.hi.x; q[4] = b.hi.y; q[5] = b.hi.z;
        }
        std::vector<HlbvhNode> hn;
        std::vector<uint32_t> sorted_prims(n);
        int root = -1;
        if (!device_build || !device_build(pb.data(), cen.data(), n, &cb.lo.x, &cb.hi.x, &hn, &root, sorted_prims.data())) {
            if (!device_build) set_error("HLBVH: the device build stage is not available");
            return -1;
        }
        ordered.assign(sorted_prims.begin(), sorted_prims.end());
        nodes.resize(hn.size());
        for (size_t i = 0; i < hn.size(); ++i) {
            BuildNode &o = nodes[i];
            o.b.lo = {hn[i].b[0], hn[i].b[1], hn[i].b[2]}; o.b.hi = {hn[i].b[3], hn[i].b[4], hn[i].b[5]};
            o.child[0] = hn[i].child[0]; o.child[1] = hn[i].child[1]; o.axis = hn[i].axis; o.first = hn[i].first; o.n = hn[i].n;
        }
        return root;
    }
};
}  // namespace

static int flatten(const std::vector<BuildNode> &bn, int node, std::vector<DNode> &out, int depth, int *maxDepth) {
    *maxDepth = std::max(*maxDepth, depth);
    int me = (int)out.size();
    out.push_back(DNode());
    const BuildNode &b = bn[node];
    DNode d;
    d.lo[0] = b.b.lo.x; d.lo[1] = b.b.lo.y; d.lo[2] = b.b.lo.z;
    d.hi0 = b.b.hi.x; d.hi1 = b.b.hi.y; d.hi2 = b.b.hi.z;
    if (b.n > 0) {
        d.offset = b.first;
        d.meta = (uint32_t)b.n;
        out[me] = d;
    } else {
        flatten(bn, b.child[0], out, depth + 1, maxDepth);
        d.offset = flatten(bn, b.child[1], out, depth + 1, maxDepth);
        d.meta = ((uint32_t)b.axis) << 16;
        out[me] = d;
    }
    return me;
}

// ------------------------------------------------------------------ BVH4 collapse
namespace {
inline bool is_leaf(const DNode &n) { return (n.meta & 0xffffu) != 0; }
inline int32_t leaf_ref(const DNode &n) { return ~(int32_t)((uint32_t)n.offset | ((n.meta & 0x7fu) << 24)); }

// returns a child reference for binary node `bi`; interior nodes become DNode4s (pre-order)
int32_t collapse(const std::vector<DNode> &bn, int bi, std::vector<DNode4> &out, int depthStack, int *needStack) {
    const DNode &N = bn[bi];
    if (is_leaf(N)) return leaf_ref(N);
    int me = (int)out.size();
    out.push_back(DNode4());
    DNode4 d;
    memset(&d, 0, sizeof(d));
    for (int k = 0; k < 4; ++k) d.child[k] = kNode4Empty;
    int axis0 = (int)(N.meta >> 16), axisA = 0, axisB = 0;
    int A = bi + 1, B = N.offset;
    int grand[4] = {-1, -1, -1, -1};
    auto group = [&](int X, int base, int *axisOut) {
        const DNode &x = bn[X];
        if (is_leaf(x)) { grand[base] = X; *axisOut = 0; }
        else { grand[base] = X + 1; grand[base + 1] = x.offset; *axisOut = (int)(x.meta >> 16); }
    };
    group(A, 0, &axisA);
    group(B, 2, &axisB);
    // order in which BVHAccel::Intersect reaches the four grandchildren for each ray octant: the near child of N first
    // (dirIsNeg[axis0]), inside each half the near grandchild first (dirIsNeg[axisA] / dirIsNeg[axisB])
    uint64_t table = 0;
    for (int oct = 0; oct < 8; ++oct) {
        int neg[3] = {oct & 1, (oct >> 1) & 1, (oct >> 2) & 1};
        int n0 = neg[axis0], nA = neg[axisA], nB = neg[axisB];
        int base0 = n0 ? 2 : 0, base1 = 2 - base0;
        int sw0 = n0 ? nB : nA, sw1 = n0 ? nA : nB;
        int order[4] = {base0 + sw0, base0 + 1 - sw0, base1 + sw1, base1 + 1 - sw1};
        uint64_t byte = (uint64_t)(order[0] | (order[1] << 2) | (order[2] << 4) | (order[3] << 6));
        table |= byte << (8 * oct);
    }
    d.order_lo = (uint32_t)table; d.order_hi = (uint32_t)(table >> 32);
    d.axes = axis0 | (axisA << 2) | (axisB << 4);
    for (int k = 0; k < 4; ++k) {   // absent children: inverted boxes, which fail every slab test
        d.lox[k] = d.loy[k] = d.loz[k] = std::numeric_limits<float>::infinity();
        d.hix[k] = d.hiy[k] = d.hiz[k] = -std::numeric_limits<float>::infinity();
    }
    int nchild = 0;
    for (int k = 0; k < 4; ++k) if (grand[k] >= 0) ++nchild;
    // a node can leave nchild-1 references on the stack while its first child is being traversed
    int below = depthStack + nchild - 1;
    *needStack = std::max(*needStack, below + 1);
    for (int k = 0; k < 4; ++k) {
        if (grand[k] < 0) continue;
        const DNode &g = bn[grand[k]];
        d.lox[k] = g.lo[0]; d.loy[k] = g.lo[1]; d.loz[k] = g.lo[2];
        d.hix[k] = g.hi0; d.hiy[k] = g.hi1; d.hiz[k] = g.hi2;
        d.child[k] = collapse(bn, grand[k], out, below, needStack);
    }
    out[me] = d;
    return me;
}
}  // namespace

// ------------------------------------------------------------------ materials -> lobes
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float roughness_to_alpha(float roughness) {  // MicroFacet.h:97-103
    roughness = std::max(roughness, (float)1e-3);
    float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
static inline float lerpf(float t, float a, float b) { return (1 - t) * a + t * b; }
static inline float sqrf(float x) { return x * x; }
struct RGB { float c[3]; };
static inline RGB clamp0(const float *p) { return {{clampf(p[0], 0, INFINITY), clampf(p[1], 0, INFINITY), clampf(p[2], 0, INFINITY)}}; }
static inline bool black(const RGB &r) { return r.c[0] == 0 && r.c[1] == 0 && r.c[2] == 0; }
static inline RGB scale_rgb(const RGB &r, float s) { return {{r.c[0] * s, r.c[1] * s, r.c[2] * s}}; }

static DLobe blank_lobe(int kind, int type) {
    DLobe l;
    memset(&l, 0, sizeof(l));
    l.kind = kind; l.type = type; l.etaA = l.etaB = 1; l.f_etaI = l.f_etaT = 1; l.f_eta = 1;
    l.alphax = l.alphay = 0.001f;
    return l;
}
static void set_alpha(DLobe &l, float ax, float ay) { l.alphax = std::max(0.001f, ax); l.alphay = std::max(0.001f, ay); }  // MicroFacet.h:75-79
static void set_R(DLobe &l, const RGB &r) { memcpy(l.R, r.c, 12); }
static void set_T(DLobe &l, const RGB &r) { memcpy(l.T, r.c, 12); }

static bool compile_material(const gnxr_material &m, DMaterial *out, bool allowMultipleLobes = true) {
    memset(out, 0, sizeof(*out));
    out->has_bump = m.has_bump;
    out->eta = 1;
    auto add = [&](const DLobe &l) { if (out->n_lobes < 8) out->lobes[out->n_lobes++] = l; };
    const bool textured = m.kd_texture > 0 || m.ks_texture > 0;
    if (textured && m.type != GNXR_MAT_PLASTIC && m.type != GNXR_MAT_MATTE) { set_error("image textures: Kd of MATTE, Kd / Ks of PLASTIC"); return false; }
    out->kd_tex = m.kd_texture; out->ks_tex = m.type == GNXR_MAT_PLASTIC ? m.ks_texture : 0;
    switch (m.type) {
    case GNXR_MAT_NONE: break;
    case GNXR_MAT_MATTE: {  // MatteMaterial.cpp:14-32
        RGB r = clamp0(m.kd);
        float sig = clampf(m.sigma, 0, 90);
        if (!black(r) || textured) {   // textured: lobe 0 is the Kd lobe, its R is looked up per hit (and the lobe dropped when black)
            if (sig == 0) { DLobe l = blank_lobe(LOBE_LAMBERT, BSDF_REFLECTION | BSDF_DIFFUSE); set_R(l, r); add(l); }
            else {  // OrenNayar ctor, Reflection.h:236-243
                DLobe l = blank_lobe(LOBE_OREN, BSDF_REFLECTION | BSDF_DIFFUSE);
                set_R(l, r);
                float sigma = (kPi / 180) * sig;
                float sigma2 = sigma * sigma;
                l.A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                l.B = 0.45f * sigma2 / (sigma2 + 0.09f);
                add(l);
            }
        }
        break;
    }
    case GNXR_MAT_MIRROR: {  // MirrorMaterial.cpp:13-24
        RGB R = clamp0(m.kr);
        if (!black(R)) { DLobe l = blank_lobe(LOBE_SPEC_REFL, BSDF_REFLECTION | BSDF_SPECULAR); set_R(l, R); l.fresnel = FRESNEL_NOOP; add(l); }
        break;
    }
    case GNXR_MAT_GLASS: {  // GlassMaterial.cpp:14-61 with allowMultipleLobes = true (PathIntegrator.cpp:120)
        float eta = m.eta[0], urough = m.urough, vrough = m.vrough;
        RGB R = clamp0(m.kr), T = clamp0(m.kt);
        out->eta = eta;
        if (black(R) && black(T)) break;
        bool isSpecular = urough == 0 && vrough == 0;
        if (isSpecular && allowMultipleLobes) {
            DLobe l = blank_lobe(LOBE_FRESNEL_SPEC, BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR);
            set_R(l, R); set_T(l, T); l.etaA = 1.f; l.etaB = eta;
            add(l);
        } else if (isSpecular) {   // allowMultipleLobes == false (WhittedIntegrator.cpp:34): SpecularReflection + SpecularTransmission
            if (!black(R)) {
                DLobe l = blank_lobe(LOBE_SPEC_REFL, BSDF_REFLECTION | BSDF_SPECULAR);
                set_R(l, R); l.fresnel = FRESNEL_DIELECTRIC; l.f_etaI = 1.f; l.f_etaT = eta;
                add(l);
            }
            if (!black(T)) {
                DLobe l = blank_lobe(LOBE_SPEC_TRANS, BSDF_TRANSMISSION | BSDF_SPECULAR);
                set_T(l, T); l.etaA = 1.f; l.etaB = eta; l.fresnel = FRESNEL_DIELECTRIC; l.f_etaI = 1.f; l.f_etaT = eta;
                add(l);
            }
        } else {
            if (m.remap_roughness) { urough = roughness_to_alpha(urough); vrough = roughness_to_alpha(vrough); }
            if (!black(R)) {
                DLobe l = blank_lobe(LOBE_MICRO_REFL, BSDF_REFLECTION | BSDF_GLOSSY);
                set_R(l, R); l.fresnel = FRESNEL_DIELECTRIC; l.f_etaI = 1.f; l.f_etaT = eta; set_alpha(l, urough, vrough);
                add(l);
            }
            if (!black(T)) {
                DLobe l = blank_lobe(LOBE_MICRO_TRANS, BSDF_TRANSMISSION | BSDF_GLOSSY);
                set_T(l, T); l.etaA = 1.f; l.etaB = eta; l.fresnel = FRESNEL_DIELECTRIC; l.f_etaI = 1.f; l.f_etaT = eta;
                set_alpha(l, urough, vrough);
                add(l);
            }
        }
        break;
    }
    case GNXR_MAT_METAL: {  // MetalMaterial.cpp:28-49
        float ur = m.urough, vr = m.vrough;
        if (m.remap_roughness) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        DLobe l = blank_lobe(LOBE_MICRO_REFL, BSDF_REFLECTION | BSDF_GLOSSY);
        l.R[0] = l.R[1] = l.R[2] = 1.f;
        l.fresnel = FRESNEL_CONDUCTOR;
        memcpy(l.f_cEtaT, m.eta, 12); memcpy(l.f_cK, m.k, 12);
        set_alpha(l, ur, vr);
        add(l);
        break;
    }
    case GNXR_MAT_PLASTIC: {  // PlasticMaterial.cpp:15-41
        RGB kd = clamp0(m.kd), ks = clamp0(m.ks);
        // textured: lobe 0 = Kd, lobe 1 = Ks always; a constant or looked-up black one is dropped per hit (device: textured_material)
        if (!black(kd) || textured) { DLobe l = blank_lobe(LOBE_LAMBERT, BSDF_REFLECTION | BSDF_DIFFUSE); set_R(l, kd); add(l); }
        if (!black(ks) || textured) {
            DLobe l = blank_lobe(LOBE_MICRO_REFL, BSDF_REFLECTION | BSDF_GLOSSY);
            set_R(l, ks); l.fresnel = FRESNEL_DIELECTRIC; l.f_etaI = 1.5f; l.f_etaT = 1.f;
            float rough = m.urough;
            if (m.remap_roughness) rough = roughness_to_alpha(rough);
            set_alpha(l, rough, rough);
            add(l);
        }
        break;
    }
    case GNXR_MAT_DISNEY: {  // DisneyMaterial.cpp:467-581
        if (m.disney_scatter_distance[0] != 0 || m.disney_scatter_distance[1] != 0 || m.disney_scatter_distance[2] != 0) {
            set_error("Disney scatterDistance != 0 selects the BSSRDF branch, which no reference integrator consumes");
            return false;
        }
        RGB c = clamp0(m.kd);
        float metallicWeight = m.disney_metallic, e = m.eta[0], strans = m.disney_spec_trans;
        float diffuseWeight = (1 - metallicWeight) * (1 - strans);
        float dt = m.disney_diff_trans / 2;
        float rough = m.disney_roughness;
        float lum = 0.212671f * c.c[0] + 0.715160f * c.c[1] + 0.072169f * c.c[2];
        RGB Ctint = lum > 0 ? RGB{{c.c[0] / lum, c.c[1] / lum, c.c[2] / lum}} : RGB{{1, 1, 1}};
        float sheenWeight = m.disney_sheen;
        RGB Csheen = {{0, 0, 0}};
        if (sheenWeight > 0) for (int i = 0; i < 3; ++i) Csheen.c[i] = (1 - m.disney_sheen_tint) * 1.f + m.disney_sheen_tint * Ctint.c[i];
        bool thin = m.disney_thin != 0;
        if (diffuseWeight > 0) {
            if (thin) {
                float flat = m.disney_flatness;
                DLobe l = blank_lobe(LOBE_DISNEY_DIFFUSE, BSDF_REFLECTION | BSDF_DIFFUSE);
                set_R(l, scale_rgb(c, diffuseWeight * (1 - flat) * (1 - dt))); add(l);
                DLobe l2 = blank_lobe(LOBE_DISNEY_FAKESS, BSDF_REFLECTION | BSDF_DIFFUSE);
                set_R(l2, scale_rgb(c, diffuseWeight * flat * (1 - dt))); l2.roughness = rough; add(l2);
            } else {
                DLobe l = blank_lobe(LOBE_DISNEY_DIFFUSE, BSDF_REFLECTION | BSDF_DIFFUSE);
                set_R(l, scale_rgb(c, diffuseWeight)); add(l);
            }
            DLobe lr = blank_lobe(LOBE_DISNEY_RETRO, BSDF_REFLECTION | BSDF_DIFFUSE);
            set_R(lr, scale_rgb(c, diffuseWeight)); lr.roughness = rough; add(lr);
            if (sheenWeight > 0) {
                DLobe ls = blank_lobe(LOBE_DISNEY_SHEEN, BSDF_REFLECTION | BSDF_DIFFUSE);
                set_R(ls, scale_rgb(Csheen, diffuseWeight * sheenWeight)); add(ls);
            }
        }
        float aspect = (float)std::sqrt(1 - m.disney_anisotropic * .9);
        float ax = std::max(.001f, sqrf(rough) / aspect), ay = std::max(.001f, sqrf(rough) * aspect);
        float r0s = sqrf(e - 1) / sqrf(e + 1);
        RGB Cspec0;
        for (int i = 0; i < 3; ++i) {
            float tint = (1 - m.disney_spec_tint) * 1.f + m.disney_spec_tint * Ctint.c[i];
            Cspec0.c[i] = (1 - metallicWeight) * (r0s * tint) + metallicWeight * c.c[i];
        }
        {
            DLobe l = blank_lobe(LOBE_MICRO_REFL, BSDF_REFLECTION | BSDF_GLOSSY);
            l.R[0] = l.R[1] = l.R[2] = 1.f;
            l.fresnel = FRESNEL_DISNEY; memcpy(l.f_R0, Cspec0.c, 12); l.f_metallic = metallicWeight; l.f_eta = e;
            set_alpha(l, ax, ay); l.disney_g = 1;
            add(l);
        }
        if (m.disney_clearcoat > 0) {
            DLobe l = blank_lobe(LOBE_DISNEY_CLEARCOAT, BSDF_REFLECTION | BSDF_GLOSSY);
            l.weight = m.disney_clearcoat;
            l.gloss = lerpf(m.disney_clearcoat_gloss, (float).1, (float).001);
            add(l);
        }
        if (strans > 0) {
            RGB T = {{strans * std::sqrt(c.c[0]), strans * std::sqrt(c.c[1]), strans * std::sqrt(c.c[2])}};
            DLobe l = blank_lobe(LOBE_MICRO_TRANS, BSDF_TRANSMISSION | BSDF_GLOSSY);
            set_T(l, T); l.etaA = 1.f; l.etaB = e; l.fresnel = FRESNEL_DIELECTRIC; l.f_etaI = 1.f; l.f_etaT = e;
            if (thin) {
                float rscaled = (0.65f * e - 0.35f) * rough;
                set_alpha(l, std::max(.001f, sqrf(rscaled) / aspect), std::max(.001f, sqrf(rscaled) * aspect));
                l.disney_g = 0;
            } else { set_alpha(l, ax, ay); l.disney_g = 1; }
            add(l);
        }
        if (thin) { DLobe l = blank_lobe(LOBE_LAMBERT_TRANS, BSDF_TRANSMISSION | BSDF_DIFFUSE); set_T(l, scale_rgb(c, dt)); add(l); }
        break;
    }
    default: set_error("unknown material type %d", m.type); return false;
    }
    for (int i = 0; i < out->n_lobes; ++i) if (!(out->lobes[i].type & BSDF_SPECULAR)) out->n_nonspecular++;
    // smallest shade-kernel specialisation whose lobe set covers this material (device_bsdf.h LM_*)
    out->shade_class = 0;
    for (int i = 0; i < out->n_lobes; ++i) {
        int k = out->lobes[i].kind, f = out->lobes[i].fresnel;
        int c = (k == LOBE_LAMBERT || k == LOBE_OREN) ? 0
              : ((k == LOBE_SPEC_REFL || k == LOBE_SPEC_TRANS || k == LOBE_FRESNEL_SPEC || k == LOBE_MICRO_REFL || k == LOBE_MICRO_TRANS) && f != FRESNEL_DISNEY &&
                 !out->lobes[i].disney_g) ? 1 : 2;
        out->shade_class = std::max(out->shade_class, c);
    }
    // class 0 promises the diffuse kernels AT MOST ONE lobe (Lambert or Oren-Nayar reflection): Bsdf<LM_DIFFUSE> then needs no lobe loops
    if (out->shade_class == 0 && out->n_lobes > 1) out->shade_class = 1;
    if (textured) out->shade_class = 3;   // own shade queue: its kernel evaluates the textures and rebuilds the lobe list per hit
    return true;
}

// ------------------------------------------------------------------ sampler tables
namespace {
struct Pcg32 {  // core/RNG.h:30-110
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    uint32_t next() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31));
    }
    uint32_t bounded(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        while (true) { uint32_t r = next(); if (r >= threshold) return r % b; }
    }
};
}  // namespace

static void build_sampler_tables(CompiledScene *cs) {
    const int N = 1000;  // PrimeTableSize, samplers/LowDiscrepancy.h:17
    cs->primes.clear();
    for (int c = 2; (int)cs->primes.size() < N; ++c) {
        bool p = true;
        for (int d = 2; d * d <= c; ++d) if (c % d == 0) { p = false; break; }
        if (p) cs->primes.push_back(c);
    }
    cs->prime_sums.resize(N);
    int sum = 0;
    for (int i = 0; i < N; ++i) { cs->prime_sums[i] = sum; sum += cs->primes[i]; }
    // ComputeRadicalInversePermutations with a default-seeded RNG (HaltonSampler.cpp:36-39)
    cs->perms.resize(sum);
    Pcg32 rng;
    uint16_t *p = cs->perms.data();
    for (int i = 0; i < N; ++i) {
        int count = cs->primes[i];
        for (int j = 0; j < count; ++j) p[j] = (uint16_t)j;
        for (int j = 0; j < count; ++j) { int other = j + (int)rng.bounded(count - j); std::swap(p[j], p[other]); }
        p += count;
    }
    // exact unsigned 32-bit division by each prime: q = (mulhi(n, M) + ((n - mulhi(n, M)) >> 1)) >> (s - 1)
    // with s = ceil(log2 d), M = floor(2^32 * (2^s - d) / d) + 1  (Granlund-Montgomery, valid for all n < 2^32)
    // One 32-byte record per dimension (device_sampler.h DimInfo): the two magic words, the base, the offset of its permutation, and the
    // per-dimension constants of ScrambledRadicalInverse -- 1 / (float)base, the perm[0] tail invBase * perm[0] / (1 - invBase)
    // (LowDiscrepancy.cpp:392; float operations, IEEE on host and device alike) and ceil(2^32 / base) for the one-multiply digit division.
    cs->prime_magic.resize(8 * (size_t)N);
    for (int i = 0; i < N; ++i) {
        uint32_t d = (uint32_t)cs->primes[i];
        int s = 0;
        while ((1ull << s) < d) ++s;
        uint64_t M = ((1ull << 32) * ((1ull << s) - d)) / d + 1;
        uint32_t *rec = &cs->prime_magic[8 * (size_t)i];
        rec[0] = (uint32_t)M;
        rec[1] = (uint32_t)s;
        rec[2] = d;
        rec[3] = (uint32_t)cs->prime_sums[i];
        const float invBase = 1.f / (float)d;
        const float tail = invBase * (float)(int)cs->perms[cs->prime_sums[i]] / (1 - invBase);
        memcpy(&rec[4], &invBase, 4);
        memcpy(&rec[5], &tail, 4);
        rec[6] = (uint32_t)((0x100000000ull + d - 1) / d);
        rec[7] = 0;
    }
}

float host_radical_inverse(const CompiledScene &cs, int baseIndex, uint64_t a) {  // LowDiscrepancy.cpp:358-372,396-403
    if (baseIndex == 0) {
        uint32_t lo = (uint32_t)a, hi = (uint32_t)(a >> 32);
        auto rev = [](uint32_t n) {
            n = (n << 16) | (n >> 16);
            n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
            n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
            n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
            n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
            return n;
        };
        uint64_t r = ((uint64_t)rev(lo) << 32) | rev(hi);
        return (float)(r * 5.4210108624275222e-20);
    }
    int base = cs.primes[baseIndex];
    const float invBase = 1.f / (float)base;
    uint64_t reversedDigits = 0;
    float invBaseN = 1;
    while (a) {
        uint64_t next = a / base, digit = a - next * base;
        reversedDigits = reversedDigits * base + digit;
        invBaseN *= invBase;
        a = next;
    }
    return std::min(reversedDigits * invBaseN, 0x1.fffffep-1f);
}

// ------------------------------------------------------------------ camera / halton
DCamera make_camera(const gnxr_camera &c, int W, int H, int medium) {
    DCamera d;
    memset(&d, 0, sizeof(d));
    Xf lookat = look_at(Vec3(c.eye[0], c.eye[1], c.eye[2]), Vec3(c.look[0], c.look[1], c.look[2]), Vec3(c.up[0], c.up[1], c.up[2]));
    Mat4 c2w = lookat.inv;  // Camera2WorldStart = Inverse(lookat), RenderThread.cpp:65
    float frame = (float)W / (float)H;
    float sxmin, sxmax, symin, symax;
    if (frame > 1.f) { sxmin = -frame; sxmax = frame; symin = -1.f; symax = 1.f; }
    else { sxmin = -1.f; sxmax = 1.f; symin = -1.f / frame; symax = 1.f / frame; }
    if (c.orthographic) {   // CreateOrthographicCamera: ScreenScale = 2 (Orthographic.cpp:108-114)
        const float ScreenScale = 2.0f;
        sxmin *= ScreenScale; sxmax *= ScreenScale; symin *= ScreenScale; symax *= ScreenScale;
    }
    // Perspective(fov, 1e-2, 1000) (Perspective.cpp:18) or Orthographic(0, 10) = Scale(1, 1, 1 / (zFar - zNear)) * Translate(0, 0, -zNear)
    // (Orthographic.h:18, Transform.cpp:282-285)
    Xf c2s = c.orthographic ? xmul(scale(1, 1, 1 / (10.f - 0.f)), translate(Vec3(0, 0, -0.f))) : perspective(c.fov_deg, 1e-2f, 1000.f);
    Xf s2r = xmul(xmul(scale((float)W, (float)H, 1), scale(1 / (sxmax - sxmin), 1 / (symin - symax), 1)), translate(Vec3(-sxmin, -symax, 0)));
    Xf r2c = xmul(xinverse(c2s), xinverse(s2r));
    memcpy(d.r2c, r2c.m.m, 64);
    memcpy(d.c2w, c2w.m, 64);
    d.lens_radius = c.lens_radius;
    d.focal_distance = c.focal_distance;
    d.medium = medium;
    d.ortho = c.orthographic ? 1 : 0;
    return d;
}

static int64_t mod64(int64_t a, int64_t b) { int64_t r = a - (a / b) * b; return r < 0 ? r + b : r; }
static void ext_gcd(uint64_t a, uint64_t b, int64_t *x, int64_t *y) {
    if (b == 0) { *x = 1; *y = 0; return; }
    int64_t d = a / b, xp, yp;
    ext_gcd(b, a % b, &xp, &yp);
    *x = yp; *y = xp - (d * yp);
}
DHalton make_halton(int W, int H) {  // HaltonSampler.cpp:33-60 (kMaxResolution = 128)
    DHalton h;
    memset(&h, 0, sizeof(h));
    int res[2] = {W, H};
    for (int i = 0; i < 2; ++i) {
        int base = (i == 0) ? 2 : 3, sc = 1, ex = 0;
        while (sc < std::min(res[i], 128)) { sc *= base; ++ex; }
        h.base_scale[i] = sc; h.base_exp[i] = ex;
    }
    h.stride = h.base_scale[0] * h.base_scale[1];
    int64_t x, y;
    ext_gcd(h.base_scale[1], h.base_scale[0], &x, &y); h.mult_inv[0] = (int)mod64(x, h.base_scale[0]);
    ext_gcd(h.base_scale[0], h.base_scale[1], &x, &y); h.mult_inv[1] = (int)mod64(x, h.base_scale[1]);
    return h;
}

// ------------------------------------------------------------------ env light tables
static inline float lanczos(float x, float tau = 2) {  // core/Texture.cpp:152-161
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x *= kPi;
    float s = std::sin(x * tau) / (x * tau);
    float l = std::sin(x) / x;
    return s * l;
}
static inline int modi(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }
static inline int round_up_pow2(int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
struct RW { int first; float w[4]; };
static std::vector<RW> resample_weights(int oldRes, int newRes) {  // MIPMap.h:41-59
    std::vector<RW> wt(newRes);
    float filterwidth = 2.f;
    for (int i = 0; i < newRes; ++i) {
        float center = (i + .5f) * oldRes / newRes;
        wt[i].first = (int)std::floor((center - filterwidth) + 0.5f);
        for (int j = 0; j < 4; ++j) { float pos = wt[i].first + j + .5f; wt[i].w[j] = lanczos((pos - center) / filterwidth); }
        float inv = 1 / (wt[i].w[0] + wt[i].w[1] + wt[i].w[2] + wt[i].w[3]);
        for (int j = 0; j < 4; ++j) wt[i].w[j] *= inv;
    }
    return wt;
}
static void dist1d(const float *f, int n, float *cdf /*n+1*/, float *funcInt) {  // Sampling.h:22-35
    cdf[0] = 0;
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + f[i - 1] / n;
    *funcInt = cdf[n];
    if (*funcInt == 0) for (int i = 1; i < n + 1; ++i) cdf[i] = float(i) / float(n);
    else for (int i = 1; i < n + 1; ++i) cdf[i] /= *funcInt;
}

// flip_y: SkyBoxLight::loadImage switches stb_image to vertically flipped loading for the whole process
// (stbi_set_flip_vertically_on_load(true), lights/SkyBoxLight.cpp:19); an InfiniteAreaLight constructed after
// a SkyBoxLight -- the order of ui/RenderThread.cpp:145-151 -- therefore sees its map upside down.
static void build_env(const gnxr_scene_desc *d, const gnxr_light &l, bool flip_y, CompiledScene *cs) {
    int w = d->env_width, h = d->env_height;
    std::vector<float> tex((size_t)w * h * 3);
    for (int j = 0; j < h; ++j)
        for (int i0 = 0; i0 < w; ++i0) {
            size_t i = (size_t)j * w + i0, src = (size_t)(flip_y ? h - 1 - j : j) * w + i0;
            for (int c = 0; c < 3; ++c) {  // texel = r * Sqrt(r), r = L * rgb (InfiniteAreaLight.cpp:33-41)
                float r = l.le[c] * d->env_rgb[3 * src + c];
                tex[3 * i + c] = r * std::sqrt(r);
            }
        }
    int rx = w, ry = h;
    if ((rx & (rx - 1)) || (ry & (ry - 1))) {  // MIPMap ctor resample, MIPMap.h:93-146 (wrap = Repeat)
        int px = round_up_pow2(rx), py = round_up_pow2(ry);
        std::vector<RW> sw = resample_weights(rx, px);
        std::vector<float> res((size_t)px * py * 3, 0.f);
        for (int t = 0; t < ry; ++t)
            for (int s = 0; s < px; ++s)
                for (int c = 0; c < 3; ++c) {
                    float acc = 0.f;
                    for (int j = 0; j < 4; ++j) {
                        int o = modi(sw[s].first + j, rx);
                        acc += sw[s].w[j] * tex[((size_t)t * rx + o) * 3 + c];
                    }
                    res[((size_t)t * px + s) * 3 + c] = acc;
                }
        std::vector<RW> tw = resample_weights(ry, py);
        std::vector<float> work((size_t)py * 3);
        for (int s = 0; s < px; ++s) {
            for (int t = 0; t < py; ++t)
                for (int c = 0; c < 3; ++c) {
                    float acc = 0.f;
                    for (int j = 0; j < 4; ++j) {
                        int o = modi(tw[t].first + j, ry);
                        acc += tw[t].w[j] * res[((size_t)o * px + s) * 3 + c];
                    }
                    work[(size_t)t * 3 + c] = acc;
                }
            for (int t = 0; t < py; ++t)
                for (int c = 0; c < 3; ++c) res[((size_t)t * px + s) * 3 + c] = clampf(work[(size_t)t * 3 + c], 0.f, INFINITY);
        }
        tex.swap(res);
        rx = px; ry = py;
    }
    cs->env_texels = tex;
    cs->env_texels4.resize(tex.size() / 3 * 4);
    for (size_t i = 0; i < tex.size() / 3; ++i) { cs->env_texels4[4 * i] = tex[3 * i]; cs->env_texels4[4 * i + 1] = tex[3 * i + 1]; cs->env_texels4[4 * i + 2] = tex[3 * i + 2]; cs->env_texels4[4 * i + 3] = 0.f; }
    // InfiniteAreaLight::Power = Pi r^2 Lmap->Lookup((.5, .5), .5) (InfiniteAreaLight.cpp:84-89): the only consumer of the upper
    // MIP levels (MIPMap.h:147-170 box-filter pyramid, :225-242 Lookup, :244-256 triangle).  Only the "power" light
    // strategy reads it.
    {
        std::vector<std::vector<float>> pyr;
        std::vector<int> lw, lh;
        pyr.push_back(tex); lw.push_back(rx); lh.push_back(ry);
        int nLevels = 1;
        for (int m = std::max(rx, ry); m > 1; m >>= 1) ++nLevels;
        auto tx = [&](int level, int s_, int t_, int c) { return pyr[level][((size_t)modi(t_, lh[level]) * lw[level] + modi(s_, lw[level])) * 3 + c]; };
        for (int i = 1; i < nLevels; ++i) {
            int sRes = std::max(1, lw[i - 1] / 2), tRes = std::max(1, lh[i - 1] / 2);
            std::vector<float> lvl((size_t)sRes * tRes * 3);
            for (int t = 0; t < tRes; ++t)
                for (int s_ = 0; s_ < sRes; ++s_)
                    for (int c = 0; c < 3; ++c)
                        lvl[((size_t)t * sRes + s_) * 3 + c] = .25f * (tx(i - 1, 2 * s_, 2 * t, c) + tx(i - 1, 2 * s_ + 1, 2 * t, c) + tx(i - 1, 2 * s_, 2 * t + 1, c) +
                                                                      tx(i - 1, 2 * s_ + 1, 2 * t + 1, c));
            pyr.push_back(std::move(lvl)); lw.push_back(sRes); lh.push_back(tRes);
        }
        auto triangle = [&](int level, float sx, float ty, float *out) {
            level = std::min(std::max(level, 0), nLevels - 1);
            float s_ = sx * lw[level] - 0.5f, t_ = ty * lh[level] - 0.5f;
            int s0 = (int)std::floor(s_), t0 = (int)std::floor(t_);
            float ds = s_ - s0, dt = t_ - t0;
            for (int c = 0; c < 3; ++c)
                out[c] = (1 - ds) * (1 - dt) * tx(level, s0, t0, c) + (1 - ds) * dt * tx(level, s0, t0 + 1, c) + ds * (1 - dt) * tx(level, s0 + 1, t0, c) +
                         ds * dt * tx(level, s0 + 1, t0 + 1, c);
        };
        const float invLog2 = 1.442695040888963387004650940071f;
        float level = nLevels - 1 + std::log(std::max(.5f, 1e-8f)) * invLog2;
        float *out = cs->env_power_lookup;
        if (level < 0) triangle(0, .5f, .5f, out);
        else if (level >= nLevels - 1) { for (int c = 0; c < 3; ++c) out[c] = tx(nLevels - 1, 0, 0, c); }
        else {
            int iLevel = (int)std::floor(level);
            float delta = level - iLevel, a[3], b[3];
            triangle(iLevel, .5f, .5f, a);
            triangle(iLevel + 1, .5f, .5f, b);
            for (int c = 0; c < 3; ++c) out[c] = (1 - delta) * a[c] + delta * b[c];
        }
    }
    DEnv &e = cs->env;
    memset(&e, 0, sizeof(e));
    e.w = rx; e.h = ry; e.dw = 2 * rx; e.dh = 2 * ry;
    Mat4 l2w; memcpy(l2w.m, l.light_to_world, 64);
    Mat4 w2l = inverse(l2w);
    memcpy(e.l2w, l2w.m, 64); memcpy(e.w2l, w2l.m, 64);
    // sampling image: level-0 bilinear lookup (fwidth maps to level < 0) times sin(theta), InfiniteAreaLight.cpp:65-80
    auto texel = [&](int s, int t, int c) { return tex[((size_t)modi(t, ry) * rx + modi(s, rx)) * 3 + c]; };
    int W2 = e.dw, H2 = e.dh;
    std::vector<float> img((size_t)W2 * H2);
    for (int v = 0; v < H2; v++) {
        float vp = (v + .5f) / (float)H2;
        float sinTheta = std::sin(kPi * (v + .5f) / H2);
        for (int u = 0; u < W2; ++u) {
            float up = (u + .5f) / (float)W2;
            float s = up * rx - 0.5f, t = vp * ry - 0.5f;
            int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
            float ds = s - s0, dt = t - t0;
            float rgb[3];
            for (int c = 0; c < 3; ++c)
                rgb[c] = (1 - ds) * (1 - dt) * texel(s0, t0, c) + (1 - ds) * dt * texel(s0, t0 + 1, c) + ds * (1 - dt) * texel(s0 + 1, t0, c) +
                         ds * dt * texel(s0 + 1, t0 + 1, c);
            float y = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
            img[(size_t)u + (size_t)v * W2] = y;
            img[(size_t)u + (size_t)v * W2] *= sinTheta;
        }
    }
    cs->env_cond_func = img;
    cs->env_cond_cdf.resize((size_t)(W2 + 1) * H2);
    cs->env_cond_int.resize(H2);
    for (int v = 0; v < H2; ++v) dist1d(&img[(size_t)v * W2], W2, &cs->env_cond_cdf[(size_t)v * (W2 + 1)], &cs->env_cond_int[v]);
    cs->env_marg_func = cs->env_cond_int;
    cs->env_marg_cdf.resize(H2 + 1);
    dist1d(cs->env_marg_func.data(), H2, cs->env_marg_cdf.data(), &e.marg_func_int);
    // Guide tables for FindInterval (GNXRayTracer.h:336-349) over the marginal and conditional cdfs: for bucket b of [0, 1) the entry is
    // the number of cdf values <= b / G, so the answer for any u of the bucket lies between two neighbouring entries and the device
    // bisects a handful of values instead of 1025 / 2049 (same predicate, same result -- the partition point is unique).
    auto upper = [](const float *cdf, int size, float u) { int n = 0, len = size; while (len > 0) { int half = len >> 1; if (cdf[n + half] <= u) { n += half + 1; len -= half + 1; } else len = half; } return n; };
    cs->env_marg_guide.resize(kEnvGuideMarg + 1);
    for (int b = 0; b <= kEnvGuideMarg; ++b) cs->env_marg_guide[b] = b == kEnvGuideMarg ? (uint16_t)(H2 + 1) : (uint16_t)upper(cs->env_marg_cdf.data(), H2 + 1, (float)b / kEnvGuideMarg);
    cs->env_cond_guide.resize((size_t)H2 * (kEnvGuideCond + 1));
    for (int v = 0; v < H2; ++v)
        for (int b = 0; b <= kEnvGuideCond; ++b)
            cs->env_cond_guide[(size_t)v * (kEnvGuideCond + 1) + b] = b == kEnvGuideCond ? (uint16_t)(W2 + 1) : (uint16_t)upper(&cs->env_cond_cdf[(size_t)v * (W2 + 1)], W2 + 1, (float)b / kEnvGuideCond);
    // Preprocess: scene.WorldBound().BoundingSphere (InfiniteAreaLight.h:23-26)
    Vec3 c = (cs->world_bound.lo + cs->world_bound.hi) / 2;
    e.world_center[0] = c.x; e.world_center[1] = c.y; e.world_center[2] = c.z;
    e.world_radius = length(c - cs->world_bound.hi);
    cs->has_env = true;
}

// ------------------------------------------------------------------ image textures
// ImageTexture::GetTexture (textures/ImageTexture.cpp:50-106: y flip, convertIn) + the MIPMap constructor (core/MIPMap.h:85-200:
// Lanczos resample to powers of two with the wrap mode, clamp, box-filtered pyramid through Texel(), EWA weight table).
static float inverse_gamma_correct(float value) {   // core/GNXRayTracer.h:367-371
    if (value <= 0.04045f) return value * 1.f / 12.92f;
    return std::pow((value + 0.055f) * 1.f / 1.055f, (float)2.4f);
}
static bool build_textures(const gnxr_scene_desc *d, CompiledScene *cs) {
    cs->textures.clear(); cs->tex_texels.clear(); cs->ewa_lut.clear();
    if (d->n_textures <= 0) return true;
    if (!d->textures || !d->texels) { set_error("textures without texel data"); return false; }
    cs->ewa_lut.resize(128);
    for (int i = 0; i < 128; ++i) {   // MIPMap.h:191-198
        float alpha = 2;
        float r2 = float(i) / float(128 - 1);
        cs->ewa_lut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
    }
    for (int ti = 0; ti < d->n_textures; ++ti) {
        const gnxr_texture &t = d->textures[ti];
        if (t.width <= 0 || t.height <= 0 || t.texel_offset < 0 || t.wrap < GNXR_WRAP_REPEAT || t.wrap > GNXR_WRAP_CLAMP) { set_error("texture %d: bad description", ti); return false; }
        const float *src = d->texels + t.texel_offset;
        int rx = t.width, ry = t.height;
        std::vector<float> tex((size_t)rx * ry * 3);
        for (int y = 0; y < ry; ++y)
            for (int x = 0; x < rx; ++x)
                for (int c = 0; c < 3; ++c) {
                    float v = src[((size_t)(ry - 1 - y) * rx + x) * 3 + c];   // flipped: (0,0) is the lower left corner
                    tex[((size_t)y * rx + x) * 3 + c] = t.scale * (t.gamma ? inverse_gamma_correct(v) : v);
                }
        auto wrap = [&](int v, int res) { return t.wrap == GNXR_WRAP_REPEAT ? modi(v, res) : (t.wrap == GNXR_WRAP_CLAMP ? std::min(std::max(v, 0), res - 1) : v); };
        if ((rx & (rx - 1)) || (ry & (ry - 1))) {
            int px = round_up_pow2(rx), py = round_up_pow2(ry);
            std::vector<RW> sw = resample_weights(rx, px);
            std::vector<float> res((size_t)px * py * 3, 0.f);
            for (int tt = 0; tt < ry; ++tt)
                for (int s_ = 0; s_ < px; ++s_)
                    for (int c = 0; c < 3; ++c) {
                        float acc = 0.f;
                        for (int j = 0; j < 4; ++j) {
                            int o = wrap(sw[s_].first + j, rx);
                            if (o >= 0 && o < rx) acc += sw[s_].w[j] * tex[((size_t)tt * rx + o) * 3 + c];
                        }
                        res[((size_t)tt * px + s_) * 3 + c] = acc;
                    }
            std::vector<RW> tw = resample_weights(ry, py);
            std::vector<float> work((size_t)py * 3);
            for (int s_ = 0; s_ < px; ++s_) {
                for (int tt = 0; tt < py; ++tt)
                    for (int c = 0; c < 3; ++c) {
                        float acc = 0.f;
                        for (int j = 0; j < 4; ++j) {
                            int o = wrap(tw[tt].first + j, ry);
                            if (o >= 0 && o < ry) acc += tw[tt].w[j] * res[((size_t)o * px + s_) * 3 + c];
                        }
                        work[(size_t)tt * 3 + c] = acc;
                    }
                for (int tt = 0; tt < py; ++tt)
                    for (int c = 0; c < 3; ++c) res[((size_t)tt * px + s_) * 3 + c] = clampf(work[(size_t)tt * 3 + c], 0.f, INFINITY);
            }
            tex.swap(res);
            rx = px; ry = py;
        }
        DTexture dt;
        memset(&dt, 0, sizeof(dt));
        dt.w0 = rx; dt.h0 = ry; dt.wrap = t.wrap; dt.trilinear = t.trilinear; dt.max_aniso = t.max_aniso;
        dt.su = t.su; dt.sv = t.sv; dt.du = t.du; dt.dv = t.dv;
        int nLevels = 1;
        for (int m = std::max(rx, ry); m > 1; m >>= 1) ++nLevels;   // 1 + Log2Int(max res)
        if (nLevels > 16) { set_error("texture %d: more than 16 MIP levels", ti); return false; }
        dt.n_levels = nLevels;
        std::vector<float> prev = tex, cur;
        int pw = rx, ph = ry;
        auto push_level = [&](const std::vector<float> &lvl, int level, int w, int h) {
            dt.level_offset[level] = (int32_t)(cs->tex_texels.size() / 4);
            for (size_t i = 0; i < (size_t)w * h; ++i) { cs->tex_texels.push_back(lvl[3 * i]); cs->tex_texels.push_back(lvl[3 * i + 1]); cs->tex_texels.push_back(lvl[3 * i + 2]); cs->tex_texels.push_back(0.f); }
        };
        push_level(prev, 0, pw, ph);
        for (int i = 1; i < nLevels; ++i) {
            int sRes = std::max(1, pw / 2), tRes = std::max(1, ph / 2);
            cur.assign((size_t)sRes * tRes * 3, 0.f);
            auto tx = [&](int s_, int t_, int c) -> float {   // Texel(i - 1, s, t)
                if (t.wrap == GNXR_WRAP_BLACK && (s_ < 0 || s_ >= pw || t_ < 0 || t_ >= ph)) return 0.f;
                return prev[((size_t)wrap(t_, ph) * pw + wrap(s_, pw)) * 3 + c];
            };
            for (int tt = 0; tt < tRes; ++tt)
                for (int s_ = 0; s_ < sRes; ++s_)
                    for (int c = 0; c < 3; ++c)
                        cur[((size_t)tt * sRes + s_) * 3 + c] = .25f * (tx(2 * s_, 2 * tt, c) + tx(2 * s_ + 1, 2 * tt, c) + tx(2 * s_, 2 * tt + 1, c) + tx(2 * s_ + 1, 2 * tt + 1, c));
            push_level(cur, i, sRes, tRes);
            prev.swap(cur); pw = sRes; ph = tRes;
        }
        cs->textures.push_back(dt);
    }
    return true;
}

// ------------------------------------------------------------------ compile
bool compile_scene(const gnxr_scene_desc *d, CompiledScene *cs, HlbvhBuildFn hlbvh_build) {
    if (!d || d->abi_version != GNXR_ABI_VERSION) { set_error("scene description ABI version mismatch"); return false; }
    if (d->n_triangles <= 0 || d->n_vertices <= 0 || !d->vertices || !d->indices || !d->tri_material || !d->tri_light) {
        set_error("scene description has no geometry");
        return false;
    }
    // counts and the arrays they describe must agree (the caller's pointers are read below without further checks)
    if (d->n_materials < 0 || d->n_lights < 0 || d->n_media < 0 || d->n_spheres < 0 || d->n_textures < 0 || (d->n_materials > 0 && !d->materials) ||
        (d->n_lights > 0 && !d->lights) || (d->n_media > 0 && !d->media) || (d->n_spheres > 0 && !d->spheres) || (d->n_textures > 0 && (!d->textures || !d->texels))) {
        set_error("scene description: a table has a count but no data");
        return false;
    }
    for (int i = 0; i < d->n_materials; ++i)
        if (d->materials[i].kd_texture > d->n_textures || d->materials[i].ks_texture > d->n_textures || d->materials[i].kd_texture < 0 || d->materials[i].ks_texture < 0) {
            set_error("material %d: texture reference out of range", i);
            return false;
        }
    if (!build_textures(d, cs)) return false;
    for (int i = 0; i < d->n_spheres; ++i) {
        const int m = d->spheres[i].material;
        if (m >= 0 && m < d->n_materials && (d->materials[m].kd_texture || d->materials[m].ks_texture)) { set_error("sphere %d: image-textured materials are supported on triangles only", i); return false; }
    }
    for (int i = 0; i < 3 * d->n_triangles; ++i)
        if (d->indices[i] < 0 || d->indices[i] >= d->n_vertices) { set_error("triangle index out of range"); return false; }
    for (int i = 0; i < d->n_triangles; ++i) {
        if (d->tri_material[i] >= d->n_materials) { set_error("material index out of range"); return false; }
        if (d->tri_light[i] >= d->n_lights) { set_error("light index out of range"); return false; }
    }
    auto vert = [&](int i) { return Vec3(d->vertices[3 * i], d->vertices[3 * i + 1], d->vertices[3 * i + 2]); };
    // ---- BVH over triangle bounds (Triangle::WorldBound, Triangle.cpp:62-69)
    BvhBuilder bb;
    bb.info.resize(d->n_triangles);
    for (int i = 0; i < d->n_triangles; ++i) {
        Box3 b;
        b.grow(vert(d->indices[3 * i])); b.grow(vert(d->indices[3 * i + 1])); b.grow(vert(d->indices[3 * i + 2]));
        bb.info[i].prim = i; bb.info[i].b = b;
        bb.info[i].c = .5f * b.lo + .5f * b.hi;
    }
    bb.nodes.reserve(2 * (size_t)d->n_triangles);
    bb.ordered.reserve(d->n_triangles);
    if (d->bvh_split_method < GNXR_BVH_SAH || d->bvh_split_method > GNXR_BVH_EQUAL_COUNTS) { set_error("unknown BVH split method %d", d->bvh_split_method); return false; }
    bb.method = d->bvh_split_method;
    int root = d->bvh_split_method == GNXR_BVH_HLBVH ? bb.build_hlbvh(hlbvh_build) : bb.build(0, d->n_triangles);
    if (root < 0) return false;
    cs->nodes.clear();
    cs->nodes.reserve(bb.nodes.size());
    cs->bvh_max_depth = 0;
    flatten(bb.nodes, root, cs->nodes, 0, &cs->bvh_max_depth);
    cs->nodes4.clear();
    cs->stack4_need = 1;
    cs->root4 = collapse(cs->nodes, 0, cs->nodes4, 0, &cs->stack4_need);
    if (cs->nodes4.empty()) cs->nodes4.push_back(DNode4());
    else if (cs->root4 >= 0) {
        // Renumber the 4-wide nodes: the top of the tree in breadth-first order (indices [0, kTopNodes): what the traversal kernel keeps in
        // LDS), everything below in the depth-first order `collapse` produced (subtrees stay compact in memory).  Node numbers are only
        // addresses: the child slots, their boxes and the visiting-order tables are untouched, so the walk is the same walk.
        const size_t n = cs->nodes4.size();
        std::vector<int32_t> order;
        order.reserve(n);
        std::vector<char> placed(n, 0);
        std::vector<int32_t> frontier{cs->root4};
        for (size_t head = 0; head < frontier.size() && order.size() < (size_t)kTopNodesMax; ++head) {
            const int32_t o = frontier[head];
            order.push_back(o); placed[o] = 1;
            for (int k = 0; k < 4; ++k) if (cs->nodes4[o].child[k] >= 0 && cs->nodes4[o].child[k] != kNode4Empty) frontier.push_back(cs->nodes4[o].child[k]);
        }
        for (size_t o = 0; o < n; ++o) if (!placed[o]) order.push_back((int32_t)o);
        std::vector<int32_t> new_of(n);
        for (size_t i = 0; i < n; ++i) new_of[order[i]] = (int32_t)i;
        std::vector<DNode4> re(n);
        for (size_t i = 0; i < n; ++i) {
            re[i] = cs->nodes4[order[i]];
            for (int k = 0; k < 4; ++k) if (re[i].child[k] >= 0 && re[i].child[k] != kNode4Empty) re[i].child[k] = new_of[re[i].child[k]];
        }
        cs->nodes4.swap(re);
        cs->root4 = new_of[cs->root4];
    }
    // bounds of every leaf of the binary tree, addressed by the leaf's first triangle: BVHAccel::Intersect tests a leaf's OWN box when
    // it pops the node (BVHAccel.cpp:665), the 4-wide walk re-tests it against the tMax of that moment (trace_kernel.hip.h, phase B)
    cs->leaf_boxes.assign((size_t)d->n_triangles * 8, 0.f);
    for (const DNode &n : cs->nodes)
        if ((n.meta & 0xffffu) != 0) {
            float *lb = &cs->leaf_boxes[(size_t)n.offset * 8];
            lb[0] = n.lo[0]; lb[1] = n.lo[1]; lb[2] = n.lo[2]; lb[3] = n.hi0; lb[4] = n.hi1; lb[5] = n.hi2;
        }
    cs->world_bound.lo = Vec3(cs->nodes[0].lo[0], cs->nodes[0].lo[1], cs->nodes[0].lo[2]);
    cs->world_bound.hi = Vec3(cs->nodes[0].hi0, cs->nodes[0].hi1, cs->nodes[0].hi2);
    // ---- spheres (outside the BVH)
    cs->spheres.assign(std::max(1, d->n_spheres), DSphere());
    for (int i = 0; i < d->n_spheres; ++i) {
        const gnxr_sphere &sp = d->spheres[i];
        DSphere &ds = cs->spheres[i];
        if (!(sp.radius > 0) || sp.material >= d->n_materials || sp.medium_inside >= d->n_media || sp.medium_outside >= d->n_media) {
            set_error("sphere %d: bad radius / material / medium", i);
            return false;
        }
        memcpy(ds.c, sp.center, 12);
        ds.r = sp.radius;
        ds.material = (sp.material >= 0 && d->materials[sp.material].type == GNXR_MAT_NONE) ? -1 : sp.material;
        ds.med_in = sp.medium_inside; ds.med_out = sp.medium_outside;
        ds.prim = d->n_triangles + i;
        Box3 b;   // Scene::WorldBound covers every primitive (the light-selection grid is laid over it)
        b.grow(Vec3(sp.center[0] - sp.radius, sp.center[1] - sp.radius, sp.center[2] - sp.radius));
        b.grow(Vec3(sp.center[0] + sp.radius, sp.center[1] + sp.radius, sp.center[2] + sp.radius));
        cs->world_bound.grow(b);
    }
    cs->n_spheres = d->n_spheres;
    // ---- triangles in leaf order
    cs->tris.resize(d->n_triangles);
    cs->leaf_of_prim.assign(d->n_triangles, -1);
    for (int li = 0; li < d->n_triangles; ++li) {
        int prim = bb.ordered[li];
        cs->leaf_of_prim[prim] = li;
        DTri &t = cs->tris[li];
        Vec3 p0 = vert(d->indices[3 * prim]), p1 = vert(d->indices[3 * prim + 1]), p2 = vert(d->indices[3 * prim + 2]);
        t.p0[0] = p0.x; t.p0[1] = p0.y; t.p0[2] = p0.z; t.prim = prim;
        t.p1[0] = p1.x; t.p1[1] = p1.y; t.p1[2] = p1.z; t.material = d->tri_material[prim];
        if (t.material >= 0 && d->materials[t.material].type == GNXR_MAT_NONE) t.material = -1;   // no BSDF: medium boundary
        t.p2[0] = p2.x; t.p2[1] = p2.y; t.p2[2] = p2.z; t.light = d->tri_light[prim];
    }
    // a one-triangle leaf's bounds are Triangle::WorldBound() = Union(Bounds3f(p0, p1), p2) (shape/Triangle.cpp:60-69): the componentwise
    // min / max of the vertices, no rounding involved.  Checked here for every such leaf so that k_trace4 can rebuild the box from the
    // vertices it loads anyway; any mismatch (there should be none) sends all leaves back to the leaf_boxes table.
    cs->leaf1_from_verts = 1;
    for (const DNode &n : cs->nodes)
        if ((n.meta & 0xffffu) == 1 && (size_t)n.offset < cs->tris.size()) {
            const DTri &t = cs->tris[n.offset];
            const float hi[3] = {n.hi0, n.hi1, n.hi2};
            for (int a = 0; a < 3; ++a) {
                const float lo_v = std::min(std::min(t.p0[a], t.p1[a]), t.p2[a]), hi_v = std::max(std::max(t.p0[a], t.p1[a]), t.p2[a]);
                if (!(lo_v == n.lo[a]) || !(hi_v == hi[a])) cs->leaf1_from_verts = 0;
            }
        }
    // ---- materials
    cs->materials.resize(std::max(1, d->n_materials));
    memset(cs->materials.data(), 0, sizeof(DMaterial) * cs->materials.size());
    for (int i = 0; i < d->n_materials; ++i) if (!compile_material(d->materials[i], &cs->materials[i])) return false;
    cs->materials_single = cs->materials;   // ComputeScatteringFunctions(..., allowMultipleLobes = false): differs for smooth glass only
    for (int i = 0; i < d->n_materials; ++i) if (!compile_material(d->materials[i], &cs->materials_single[i], false)) return false;
    // ---- per-corner uvs and shading normals (TriangleMesh::uv / ::n).  A triangle whose uvs are not the GetUVs defaults or that has
    // normals gets a COPY of its material with shade class 3 (the general shade queue, which reads the attribute tables and derives
    // dpdu / dpdv, the shading frame and dndu / dndv from them); everything else stays on the kernels with the defaults folded in.
    cs->tri_uv.clear();
    cs->tri_n.clear();
    cs->tri_s.clear();
    if (d->tri_uv || d->tri_n || d->tri_s) {
        if (d->tri_uv) cs->tri_uv.assign((size_t)d->n_triangles * 8, 0.f);
        if (d->tri_n) cs->tri_n.assign((size_t)d->n_triangles * 12, 0.f);
        if (d->tri_s) cs->tri_s.assign((size_t)d->n_triangles * 12, 0.f);
        std::vector<int> attr_copy(d->n_materials, -1);
        const float def[6] = {0, 0, 1, 0, 1, 1}, zero9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int li = 0; li < d->n_triangles; ++li) {
            const int prim = cs->tris[li].prim;
            bool custom = false;
            if (d->tri_uv) {
                const float *uv = d->tri_uv + 6 * (size_t)prim;
                memcpy(&cs->tri_uv[(size_t)li * 8], uv, 24);
                custom = custom || memcmp(uv, def, 24) != 0;
            }
            if (d->tri_n) {
                const float *nn = d->tri_n + 9 * (size_t)prim;
                memcpy(&cs->tri_n[(size_t)li * 12], nn, 36);
                const bool hasN = memcmp(nn, zero9, 36) != 0;
                if (hasN && d->tri_light[prim] >= 0) { set_error("triangle %d: per-vertex normals on an emissive triangle are not supported", prim); return false; }
                custom = custom || hasN;
            }
            if (d->tri_s) {
                const float *sv = d->tri_s + 9 * (size_t)prim;
                memcpy(&cs->tri_s[(size_t)li * 12], sv, 36);
                const bool hasS = memcmp(sv, zero9, 36) != 0;
                if (hasS && d->tri_light[prim] >= 0) { set_error("triangle %d: per-vertex tangents on an emissive triangle are not supported", prim); return false; }
                custom = custom || hasS;
            }
            DTri &t = cs->tris[li];
            if (t.material < 0 || !custom) continue;
            if (attr_copy[t.material] < 0) {
                attr_copy[t.material] = (int)cs->materials.size();
                DMaterial m = cs->materials[t.material], ms = cs->materials_single[t.material];
                m.shade_class = ms.shade_class = 3;
                m.has_attr = ms.has_attr = 1;
                cs->materials.push_back(m);
                cs->materials_single.push_back(ms);
            }
            t.material = attr_copy[t.material];
        }
    }
    cs->tri_class.assign(d->n_triangles, 0);
    for (int li = 0; li < d->n_triangles; ++li) {
        const int m = cs->tris[li].material;
        if (m >= 0) cs->tri_class[li] = (uint8_t)cs->materials[m].shade_class;
    }
    // ---- lights
    cs->lights.resize(std::max(1, d->n_lights));
    memset(cs->lights.data(), 0, sizeof(DLight) * cs->lights.size());
    cs->infinite_lights.clear();
    cs->desc_lights.assign(d->lights, d->lights + d->n_lights);
    cs->has_env = false;
    for (int i = 0; i < d->n_lights; ++i) {
        const gnxr_light &l = d->lights[i];
        DLight &dl = cs->lights[i];
        dl.type = l.type; dl.two_sided = l.two_sided; dl.tri_leaf = -1;
        dl.n_samples = std::max(1, l.n_samples);
        memcpy(dl.le, l.le, 12);
        if (l.type == GNXR_LIGHT_AREA_TRI) {
            if (l.tri < 0 || l.tri >= d->n_triangles) { set_error("light %d: triangle out of range", i); return false; }
            if (d->tri_light[l.tri] != i) { set_error("light %d: tri_light[%d] does not point back", i, l.tri); return false; }
            dl.tri_leaf = cs->leaf_of_prim[l.tri];
            Vec3 p0 = vert(d->indices[3 * l.tri]), p1 = vert(d->indices[3 * l.tri + 1]), p2 = vert(d->indices[3 * l.tri + 2]);
            memcpy(dl.p0, &p0, 12); memcpy(dl.p1, &p1, 12); memcpy(dl.p2, &p2, 12);
            dl.area = (float)(0.5 * length(cross(p1 - p0, p2 - p0)));  // Triangle::Area, Triangle.cpp:455-462
            dl.inv_area = 1 / dl.area;
            Vec3 n = normalize(cross(p1 - p0, p2 - p0));                // Triangle::Sample, Triangle.cpp:473
            memcpy(dl.n, &n, 12);
        } else if (l.type == GNXR_LIGHT_INFINITE) {
            if (!d->env_rgb || d->env_width <= 0 || d->env_height <= 0) { set_error("INFINITE light without env map"); return false; }
            if (cs->has_env) { set_error("only one INFINITE light is supported"); return false; }
            dl.env = 1;
            bool flip_y = false;
            for (int k = 0; k < i; ++k) if (d->lights[k].type == GNXR_LIGHT_SKYBOX) flip_y = true;
            build_env(d, l, flip_y, cs);
            cs->infinite_lights.push_back(i);
        } else if (l.type == GNXR_LIGHT_SKYBOX) {
            memcpy(dl.center, l.center, 12);
            dl.radius = l.radius;
            cs->infinite_lights.push_back(i);
        } else if (l.type == GNXR_LIGHT_POINT || l.type == GNXR_LIGHT_SPOT || l.type == GNXR_LIGHT_DISTANT) {
            // lights/PointLight.cpp, SpotLight.cpp, DistantLight.cpp (field reuse: device_lights.h light_sample)
            Mat4 l2w, w2l;
            memcpy(&l2w.m[0][0], l.light_to_world, 64);
            w2l = inverse(l2w);                                             // Transform(const Matrix4x4 &): mInv = Inverse(m)
            Vec3 pL = xform_point(l2w, Vec3(0, 0, 0));
            memcpy(dl.p0, &pL, 12);
            for (int c = 0; c < 3; ++c) { dl.p1[c] = w2l.m[0][c]; dl.p2[c] = w2l.m[1][c]; dl.center[c] = w2l.m[2][c]; }
            dl.area = std::cos((kPi / 180) * l.radius);                     // cosTotalWidth
            dl.inv_area = std::cos((kPi / 180) * l.falloff_start);          // cosFalloffStart
            Vec3 w = normalize(xform_vector(l2w, Vec3(l.center[0], l.center[1], l.center[2])));
            memcpy(dl.n, &w, 12);
            // DistantLight::Preprocess: scene.WorldBound().BoundingSphere (Geometry.h:770-773)
            Vec3 c = (cs->world_bound.lo + cs->world_bound.hi) / 2;
            const Box3 &wb = cs->world_bound;
            bool inside = c.x >= wb.lo.x && c.x <= wb.hi.x && c.y >= wb.lo.y && c.y <= wb.hi.y && c.z >= wb.lo.z && c.z <= wb.hi.z;
            dl.radius = inside ? length(c - wb.hi) : 0.f;
        } else { set_error("light %d: unknown type %d", i, l.type); return false; }
    }
    // ---- media
    cs->media.clear();
    if (d->n_media > 0) {
        if (!d->media) { set_error("n_media > 0 without a media array"); return false; }
        cs->media.assign(d->media, d->media + d->n_media);
        int64_t total = 0;
        for (size_t i = 0; i < cs->media.size(); ++i) {   // checked before anything is copied or read
            const gnxr_medium &m = cs->media[i];
            if (m.type != GNXR_MEDIUM_GRID) continue;
            if (m.nx <= 0 || m.ny <= 0 || m.nz <= 0) { set_error("medium %d: empty density grid", (int)i); return false; }
            if (m.density_offset < 0) { set_error("medium %d: negative density_offset", (int)i); return false; }
            if (!d->grid_density) { set_error("medium %d: GRID medium without grid_density", (int)i); return false; }
            total = std::max<int64_t>(total, m.density_offset + (int64_t)m.nx * m.ny * m.nz);
        }
        if (total > 0) cs->grid_density.assign(d->grid_density, d->grid_density + total);
    }
    cs->dmedia.assign(std::max<size_t>(1, cs->media.size()), DMedium());
    for (size_t i = 0; i < cs->media.size(); ++i) {
        const gnxr_medium &m = cs->media[i];
        DMedium &dm = cs->dmedia[i];
        memset(&dm, 0, sizeof(dm));
        dm.type = m.type; dm.nx = m.nx; dm.ny = m.ny; dm.nz = m.nz; dm.g = m.g;
        memcpy(dm.sigma_a, m.sigma_a, 12); memcpy(dm.sigma_s, m.sigma_s, 12);
        dm.sigma_t = m.sigma_a[0] + m.sigma_s[0];
        if (m.type == GNXR_MEDIUM_GRID) {
            if (m.nx <= 0 || m.ny <= 0 || m.nz <= 0) { set_error("medium %d: empty density grid", (int)i); return false; }
            Mat4 m2w;
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m2w.m[r][c] = m.medium_to_world[4 * r + c];
            Mat4 inv = inverse(m2w);   // Transform(Matrix4x4) -> mInv = Inverse(m), Transform.h:110-112
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) dm.w2m[4 * r + c] = inv.m[r][c];
            float maxDensity = 0;   // GridDensityMedium.h:28-31
            const float *dd = cs->grid_density.data() + m.density_offset;
            for (int64_t k = 0; k < (int64_t)m.nx * m.ny * m.nz; ++k) maxDensity = std::max(maxDensity, dd[k]);
            dm.inv_max_density = 1 / maxDensity;
            if (m.density_offset + (int64_t)m.nx * m.ny * m.nz >= (1ll << 31)) { set_error("medium %d: density grid too large", (int)i); return false; }
            dm.density_offset = (int32_t)m.density_offset;
        } else if (m.type != GNXR_MEDIUM_HOMOGENEOUS) { set_error("medium %d: unknown type %d", (int)i, m.type); return false; }
    }
    cs->tri_media.clear();
    if (d->tri_medium_inside && d->tri_medium_outside) {
        cs->tri_media.resize(2 * (size_t)d->n_triangles);
        for (int li = 0; li < d->n_triangles; ++li) {
            int prim = cs->tris[li].prim;
            int mi = d->tri_medium_inside[prim], mo = d->tri_medium_outside[prim];
            if (mi >= d->n_media || mo >= d->n_media) { set_error("medium index out of range"); return false; }
            cs->tri_media[2 * li] = mi; cs->tri_media[2 * li + 1] = mo;
        }
    }
    cs->camera = d->camera;
    // Camera::medium: -1 == none.  A zero-initialised description (memset) says "inside medium 0", so the index is checked
    // against the media that exist; without media any value means "none".
    if (d->n_media > 0 && (d->camera_medium < -1 || d->camera_medium >= d->n_media)) { set_error("camera_medium %d out of range (%d media)", d->camera_medium, d->n_media); return false; }
    cs->camera_medium = d->n_media > 0 ? d->camera_medium : -1;
    build_sampler_tables(cs);
    return true;
}

// ------------------------------------------------------------------ light-selection table
// Host restatement of the pieces of <Light>::Sample_Li that SpatialLightDistribution::ComputeDistribution
// (core/LightDistribution.cpp:206-274) evaluates: only Li.y()/pdf from a point with no normal.
namespace {
struct GridCtx {
    const CompiledScene *cs;
    int nl;
};
static inline int find_interval(const float *cdf, int size, float u) {  // GNXRayTracer.h:336-349 with pred cdf[i] <= u
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    return std::min(std::max(first - 1, 0), size - 2);
}
static float sample_continuous(const float *func, const float *cdf, int n, float funcInt, float u, float *pdf, int *off) {
    int offset = find_interval(cdf, n + 1, u);
    if (off) *off = offset;
    float du = u - cdf[offset];
    if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
    if (pdf) *pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
    return (offset + du) / n;
}
// returns Li.y()/pdf contribution (0 when pdf == 0)
static float light_contrib(const CompiledScene &cs, int j, Vec3 ref, float u0, float u1) {
    const DLight &l = cs.lights[j];
    if (l.type == GNXR_LIGHT_AREA_TRI) {
        // Triangle::Sample + Shape::Sample(ref) + DiffuseAreaLight::Sample_Li
        float su0 = std::sqrt(u0);
        float b0 = 1 - su0, b1 = u1 * su0;
        Vec3 p0(l.p0[0], l.p0[1], l.p0[2]), p1(l.p1[0], l.p1[1], l.p1[2]), p2(l.p2[0], l.p2[1], l.p2[2]);
        Vec3 p = b0 * p0 + b1 * p1 + (1 - b0 - b1) * p2;
        Vec3 n(l.n[0], l.n[1], l.n[2]);
        float pdf = 1 / l.area;
        Vec3 wi = p - ref;
        if (dot(wi, wi) == 0) pdf = 0;
        else {
            wi = normalize(wi);
            Vec3 dd = ref - p;
            pdf *= dot(dd, dd) / std::abs(dot(n, Vec3(-wi.x, -wi.y, -wi.z)));
            if (std::isinf(pdf)) pdf = 0.f;
        }
        Vec3 d2 = p - ref;
        if (pdf == 0 || dot(d2, d2) == 0) return 0;
        Vec3 w = normalize(p - ref);
        bool dotNW = dot(n, Vec3(-w.x, -w.y, -w.z));  // DiffuseAreaLight.h:24 bool truncation
        if (!(l.two_sided || dotNW > 0)) return 0;
        float y = 0.212671f * l.le[0] + 0.715160f * l.le[1] + 0.072169f * l.le[2];
        return pdf > 0 ? y / pdf : 0;
    } else if (l.type == GNXR_LIGHT_INFINITE) {
        const DEnv &e = cs.env;
        float pdfs[2];
        int v;
        float d1 = sample_continuous(cs.env_marg_func.data(), cs.env_marg_cdf.data(), e.dh, e.marg_func_int, u1, &pdfs[1], &v);
        float d0 = sample_continuous(&cs.env_cond_func[(size_t)v * e.dw], &cs.env_cond_cdf[(size_t)v * (e.dw + 1)], e.dw, cs.env_cond_int[v], u0, &pdfs[0], nullptr);
        float mapPdf = pdfs[0] * pdfs[1];
        if (mapPdf == 0) return 0;
        float theta = d1 * kPi;
        float sinTheta = std::sin(theta);
        float pdf = mapPdf / (2 * kPi * kPi * sinTheta);
        if (sinTheta == 0) pdf = 0;
        // Lmap->Lookup(uv): level-0 bilinear
        int rx = e.w, ry = e.h;
        float s = d0 * rx - 0.5f, t = d1 * ry - 0.5f;
        int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
        float ds = s - s0, dt = t - t0;
        auto texel = [&](int ss, int tt, int c) { return cs.env_texels[((size_t)modi(tt, ry) * rx + modi(ss, rx)) * 3 + c]; };
        float rgb[3];
        for (int c = 0; c < 3; ++c)
            rgb[c] = (1 - ds) * (1 - dt) * texel(s0, t0, c) + (1 - ds) * dt * texel(s0, t0 + 1, c) + ds * (1 - dt) * texel(s0 + 1, t0, c) +
                     ds * dt * texel(s0 + 1, t0 + 1, c);
        float y = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
        return pdf > 0 ? y / pdf : 0;
    } else if (l.type == GNXR_LIGHT_POINT || l.type == GNXR_LIGHT_SPOT || l.type == GNXR_LIGHT_DISTANT) {   // pdf = 1
        float rgb[3] = {l.le[0], l.le[1], l.le[2]};
        if (l.type != GNXR_LIGHT_DISTANT) {
            Vec3 pL(l.p0[0], l.p0[1], l.p0[2]);
            Vec3 wi = normalize(pL - ref);
            Vec3 dd = pL - ref;
            float d2 = dot(dd, dd), falloff = 1;
            if (l.type == GNXR_LIGHT_SPOT) {   // SpotLight::Falloff(-wi)
                Vec3 w(-wi.x, -wi.y, -wi.z);
                Vec3 wl = normalize(Vec3(l.p1[0] * w.x + l.p1[1] * w.y + l.p1[2] * w.z, l.p2[0] * w.x + l.p2[1] * w.y + l.p2[2] * w.z,
                                         l.center[0] * w.x + l.center[1] * w.y + l.center[2] * w.z));
                float cosTheta = wl.z;
                if (cosTheta < l.area) falloff = 0;
                else if (cosTheta >= l.inv_area) falloff = 1;
                else { float delta = (cosTheta - l.area) / (l.inv_area - l.area); falloff = (delta * delta) * (delta * delta); }
                for (int c = 0; c < 3; ++c) rgb[c] = rgb[c] * falloff / d2;
            } else for (int c = 0; c < 3; ++c) rgb[c] = rgb[c] / d2;
        }
        float y = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
        return y / 1.f;
    } else {  // SKYBOX: Li = 0, pdf = 1/4pi
        return 0.f / (1.f / (4 * kPi));
    }
}
}  // namespace

void build_light_grid(const CompiledScene &cs, int strategy, DLightGrid *grid, std::vector<float> *table, bool layout_only) {
    int nl = (int)cs.desc_lights.size();
    memset(grid, 0, sizeof(*grid));
    grid->n_lights = nl;
    // floats per voxel record; up to three lights (the Cornell configurations: two light triangles, + the environment light) the record is
    // padded to whole float4s so that light_select reads it with one or two aligned 16-byte loads instead of a chain of dependent gathers
    grid->stride = nl <= 3 ? ((2 * nl + 1 + 3) & ~3) : 2 * nl + 1;
    grid->nvox[0] = grid->nvox[1] = grid->nvox[2] = 1;
    memcpy(grid->lo, &cs.world_bound.lo, 12);
    memcpy(grid->hi, &cs.world_bound.hi, 12);
    if (nl == 0) { table->assign(1, 0.f); return; }
    auto write_dist = [&](const float *func, float *dst) {
        std::vector<float> cdf(nl + 1);
        float funcInt;
        dist1d(func, nl, cdf.data(), &funcInt);
        for (int i = 0; i < nl; ++i) dst[i] = cdf[i + 1];
        for (int i = 0; i < nl; ++i) dst[nl + i] = func[i];
        dst[2 * nl] = funcInt;
    };
    // CreateLightSampleDistribution, LightDistribution.cpp:15-33
    if (strategy == GNXR_LIGHTS_UNIFORM || nl == 1) {
        std::vector<float> prob(nl, 1.f);
        table->assign(grid->stride, 0.f);
        write_dist(prob.data(), table->data());
        return;
    }
    if (strategy == GNXR_LIGHTS_POWER) {  // ComputeLightPowerDistribution, Integrator.cpp:212-220
        std::vector<float> power(nl, 0.f);
        for (int i = 0; i < nl; ++i) {
            const DLight &l = cs.lights[i];
            if (l.type == GNXR_LIGHT_AREA_TRI) {
                float s = (l.two_sided ? 2 : 1);
                float rgb[3];
                for (int c = 0; c < 3; ++c) rgb[c] = s * l.le[c] * l.area * kPi;
                power[i] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
            } else if (l.type == GNXR_LIGHT_INFINITE && cs.has_env) {   // InfiniteAreaLight::Power, InfiniteAreaLight.cpp:84-89
                float k = kPi * cs.env.world_radius * cs.env.world_radius;
                float rgb[3] = {k * cs.env_power_lookup[0], k * cs.env_power_lookup[1], k * cs.env_power_lookup[2]};
                power[i] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
            } else if (l.type == GNXR_LIGHT_POINT || l.type == GNXR_LIGHT_SPOT || l.type == GNXR_LIGHT_DISTANT) {
                float rgb[3];
                for (int c = 0; c < 3; ++c)
                    rgb[c] = l.type == GNXR_LIGHT_POINT ? 4 * kPi * l.le[c]                                         // PointLight.cpp:24
                           : l.type == GNXR_LIGHT_SPOT ? l.le[c] * 2 * kPi * (1 - .5f * (l.inv_area + l.area))      // SpotLight.cpp:42-45
                                                       : l.le[c] * kPi * l.radius * l.radius;                       // DistantLight.cpp:27-30
                power[i] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
            } else power[i] = 0;  // SkyBoxLight::Power() = 0
        }
        table->assign(grid->stride, 0.f);
        write_dist(power.data(), table->data());
        return;
    }
    // SpatialLightDistribution, LightDistribution.cpp:70-97 (maxVoxels = 64)
    grid->spatial = 1;
    Vec3 diag = cs.world_bound.diag();
    float bmax = diag[cs.world_bound.max_extent()];
    for (int i = 0; i < 3; ++i) grid->nvox[i] = std::max(1, int(std::round(diag[i] / bmax * 64)));
    size_t nv = (size_t)grid->nvox[0] * grid->nvox[1] * grid->nvox[2];
    table->assign(layout_only ? 1 : nv * grid->stride, 0.f);
    if (layout_only) return;   // the device fills the table (k_light_grid)
    // the 128 probe points of ComputeDistribution, LightDistribution.cpp:226-234
    const int nSamples = 128;
    float ri[5][nSamples];
    for (int i = 0; i < nSamples; ++i)
        for (int k = 0; k < 5; ++k) ri[k][i] = host_radical_inverse(cs, k, i);
    const int nvx = grid->nvox[0], nvy = grid->nvox[1], nvz = grid->nvox[2];
    auto work = [&](int x0, int x1) {
        std::vector<float> contrib(nl);
        for (int x = x0; x < x1; ++x)
            for (int y = 0; y < nvy; ++y)
                for (int z = 0; z < nvz; ++z) {
                    Vec3 p0(float(x) / float(nvx), float(y) / float(nvy), float(z) / float(nvz));
                    Vec3 p1(float(x + 1) / float(nvx), float(y + 1) / float(nvy), float(z + 1) / float(nvz));
                    Vec3 a = cs.world_bound.lerp(p0), b = cs.world_bound.lerp(p1);
                    Box3 vb;  // Bounds3f(p1, p2) ctor takes min/max
                    vb.lo = {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)};
                    vb.hi = {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)};
                    std::fill(contrib.begin(), contrib.end(), 0.f);
                    for (int i = 0; i < nSamples; ++i) {
                        Vec3 po = vb.lerp(Vec3(ri[0][i], ri[1][i], ri[2][i]));
                        for (int j = 0; j < nl; ++j) contrib[j] += light_contrib(cs, j, po, ri[3][i], ri[4][i]);
                    }
                    float sum = 0;
                    for (float c : contrib) sum += c;
                    float avg = sum / (nSamples * contrib.size());
                    float minC = (avg > 0) ? (float)(.001 * avg) : 1;
                    for (int j = 0; j < nl; ++j) contrib[j] = std::max(contrib[j], minC);
                    size_t idx = ((size_t)x * nvy + y) * nvz + z;
                    write_dist(contrib.data(), &(*table)[idx * grid->stride]);
                }
    };
    int nthreads = (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    nthreads = std::min(nthreads, nvx);
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(work, nvx * t / nthreads, nvx * (t + 1) / nthreads);
    for (auto &t : pool) t.join();
}

// RadicalInverse(k, i), k = 0..4, i = 0..127: the probe points / light samples of ComputeDistribution (LightDistribution.cpp:226-234)
void light_grid_probes(const CompiledScene &cs, float *ri /* [5][128] */) {
    for (int i = 0; i < 128; ++i)
        for (int k = 0; k < 5; ++k) ri[k * 128 + i] = host_radical_inverse(cs, k, i);
}

}  // namespace gnxr

// gnxr_device_types.h -- POD tables shared by the host scene compiler and the HIP kernels.
//
// Layout rules (MI355X): everything a lane touches per ray is 16-byte granular so one access is a
// dwordx4; per-scene tables are read-mostly and small enough (BVH of 100k triangles = 6.4 MB nodes
// + 4.8 MB triangles) to live in the 32 MiB aggregate L2 / 256 MiB Infinity Cache.
#pragma once
#include <stdint.h>

namespace gnxr {

// ---- BVH: the reference's 32-byte LinearBVHNode (accelerator/BVHAccel.cpp:54-65) as two dwordx4 ----
struct DNode {
    float lo[3];
    float hi0;        // hi.x
    float hi1, hi2;   // hi.y hi.z
    int32_t offset;   // leaf: first triangle (leaf order); interior: second child
    uint32_t meta;    // nPrims (low 16) | axis << 16
};
static_assert(sizeof(DNode) == 32, "DNode must be 32 bytes");

// 4-wide node made by collapsing two levels of the binary tree (children = the grandchildren A.l A.r B.l B.r of a
// binary node N with children A, B).  The split axes of N, A and B are kept so that the four children are visited in
// exactly the order BVHAccel::Intersect would reach them (near child first by dirIsNeg[axis]), which keeps hit
// records bit-identical while halving the number of dependent memory round trips per ray.  128 B = 8 dwordx4.
struct DNode4 {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    int32_t child[4];     // >= 0: DNode4 index; < 0: leaf, ~ref = first triangle | nPrims << 24; kNode4Empty: no child (box inverted)
    uint32_t order_lo, order_hi;   // visiting order per ray octant (neg0 | neg1 << 1 | neg2 << 2): byte o = 4 x 2-bit child slots, nearest first
    int32_t axes;                  // axis0 | axisA << 2 | axisB << 4 (kept for inspection)
    int32_t _pad;
};
static_assert(sizeof(DNode4) == 128, "DNode4 must be 128 bytes");
constexpr int32_t kNode4Empty = 0x7ffffffe;
constexpr int kTopNodesMax = 1024;   // DNode4[0 .. kTopNodesMax) are the top of the tree in breadth-first order (scene_compile.cpp); the traversal kernel caches a prefix of them in LDS
constexpr int32_t kRefDone = 0x7fffffff;

// Triangle in BVH-leaf order: three dwordx4.  .w lanes carry the ids the shading stage needs.
struct DTri {
    float p0[3]; int32_t prim;      // authoring index (gnxr_hit.prim)
    float p1[3]; int32_t material;  // -1 == null material
    float p2[3]; int32_t light;     // index into lights or -1
};
static_assert(sizeof(DTri) == 48, "DTri must be 48 bytes");

// Sphere (pbrt-v3 quadratic sphere; include/gnxr.h explains why): kept outside the triangle BVH, tested first.
// A sphere hit is reported as hit code -2 - sphereIndex (triangles: leaf index >= 0, miss: -1).
struct DSphere {
    float c[3]; float r;
    int32_t material, med_in, med_out, prim;
};
static_assert(sizeof(DSphere) == 32, "DSphere must be 32 bytes");

// ---- BSDF lobes, precomputed per material on the host (all textures are constants) ----
enum LobeKind : int32_t {
    LOBE_LAMBERT = 0, LOBE_OREN, LOBE_SPEC_REFL, LOBE_SPEC_TRANS, LOBE_FRESNEL_SPEC, LOBE_MICRO_REFL, LOBE_MICRO_TRANS,
    LOBE_LAMBERT_TRANS, LOBE_DISNEY_DIFFUSE, LOBE_DISNEY_FAKESS, LOBE_DISNEY_RETRO, LOBE_DISNEY_SHEEN, LOBE_DISNEY_CLEARCOAT
};
enum FresnelKind : int32_t { FRESNEL_NOOP = 0, FRESNEL_DIELECTRIC, FRESNEL_CONDUCTOR, FRESNEL_DISNEY };
enum : int32_t {
    BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31
};

struct DLobe {            // 32 dwords
    int32_t kind, type, fresnel, disney_g;
    float R[3]; float A;
    float T[3]; float B;
    float etaA, etaB, alphax, alphay;
    float f_etaI, f_etaT, f_metallic, f_eta;      // dielectric / disney fresnel
    float f_cEtaT[3]; float roughness;            // conductor etaT (etaI == 1) ; disney roughness
    float f_cK[3]; float weight;                  // conductor k ; clearcoat weight
    float f_R0[3]; float gloss;                   // disney R0 ; clearcoat gloss
};
static_assert(sizeof(DLobe) == 128, "DLobe must be 128 bytes");

struct DMaterial {
    int32_t n_lobes;
    int32_t has_bump;
    int32_t n_nonspecular;   // NumComponents(BSDF_ALL & ~BSDF_SPECULAR)
    float eta;               // BSDF::eta
    int32_t shade_class;     // which k_shade specialisation can evaluate every lobe: 0 diffuse, 1 glossy, 2 any, 3 general queue (image textures, per-corner uvs)
    int32_t kd_tex, ks_tex;  // 1 + texture index or 0: lobe 0 = Lambert / Oren (Kd), lobe 1 = microfacet (Ks), see compile_material
    int32_t has_attr;        // copy of a material for triangles with uvs / shading normals of their own (DTexTables::tri_uv, tri_n); shade class 3
    DLobe lobes[8];
};

// ---- image textures: ImageTexture + UVMapping2D + MIPMap (textures/ImageTexture.h, core/Texture.cpp:163-175, core/MIPMap.h) ----
struct DTexture {
    int32_t n_levels, w0, h0, wrap;      // level i is max(1, w0 >> i) x max(1, h0 >> i)
    int32_t trilinear; float max_aniso;
    float su, sv, du, dv;
    int32_t level_offset[16];            // first texel (float4 rgb_) of each level in DTexTables::texels
    int32_t _pad[2];
};
// ---- lights ----
struct DLight {
    int32_t type;       // gnxr_light_type
    int32_t tri_leaf;   // AREA_TRI: triangle index in leaf order
    int32_t two_sided;
    int32_t env;        // INFINITE: 1
    float le[3]; float area;
    float p0[3]; float inv_area;
    float p1[3]; float radius;
    float p2[3]; int32_t n_samples;   // max(1, Light::nSamples), core/Light.cpp:19
    float n[3]; float _pad2;   // Normalize(Cross(p1-p0, p2-p0)), Triangle.cpp:473
    float center[3]; float _pad3;
};

struct DHalton {
    int32_t base_scale[2], base_exp[2];
    int32_t stride, mult_inv[2];
    int32_t base32_max;   // radical inverses of bases <= this keep reversedDigits in 32 bits: base * (largest sample index in use) < 2^32 (0: never)
};

struct DCamera {
    float r2c[16];   // RasterToCamera
    float c2w[16];   // CameraToWorld
    float lens_radius, focal_distance;
    int32_t medium;
    int32_t ortho;    // OrthographicCamera (camera/Orthographic.cpp) instead of PerspectiveCamera
};

constexpr int kEnvGuideMarg = 1024, kEnvGuideCond = 512;   // buckets of the FindInterval guide tables (powers of two: u * G is exact)
struct DEnv {         // InfiniteAreaLight tables
    int32_t w, h;     // Lmap level-0 size
    int32_t dw, dh;   // distribution size (2w, 2h)
    float l2w[16], w2l[16];
    float world_center[3]; float world_radius;
    float marg_func_int;
};

// Medium (media/HomogeneousMedium.h, media/GridDensityMedium.h): coefficients + the grid's WorldToMedium
struct DMedium {
    int32_t type;            // gnxr_medium_type
    int32_t nx, ny, nz;
    float sigma_a[3]; float g;
    float sigma_s[3]; float sigma_t;          // GRID: (sigma_a + sigma_s)[0], GridDensityMedium.h:27
    float w2m[16];                            // Inverse(mediumToWorld), row-major
    float inv_max_density;
    int32_t density_offset;
    int32_t _pad[2];
};

struct DLightGrid {
    int32_t nvox[3];
    int32_t n_lights;
    int32_t stride;      // floats per voxel: cdf[1..n] func[0..n-1] funcInt
    int32_t spatial;     // 0: one distribution for every point (uniform/power)
    float lo[3], hi[3];
};

}  // namespace gnxr

// host_scene.h -- host side of libgnxr: scene authoring (mirror of ui/ModelList.cpp and
// ui/MaterialList.cpp) and the scene compiler that flattens a gnxr_scene_desc into device tables.
#pragma once
#include <string>
#include <vector>

#include "../../include/gnxr.h"
#include "gnxr_device_types.h"
#include "host_math.h"

namespace gnxr {

void set_error(const char *fmt, ...);
const char *get_error();

// ---------------- authoring ----------------
struct Builder {
    std::vector<float> vertices;   // world space
    std::vector<int32_t> indices, tri_material, tri_light, tri_med_in, tri_med_out;
    std::vector<gnxr_material> materials;
    std::vector<gnxr_light> lights;
    std::vector<gnxr_medium> media;
    std::vector<gnxr_sphere> spheres;
    std::vector<float> grid_density;
    std::vector<gnxr_texture> textures;
    std::vector<float> texels;
    std::vector<float> tri_uv;     // empty, or 6 floats per triangle (defaults for triangles never given uvs)
    std::vector<float> tri_n;      // empty, or 9 floats per triangle (zeros for triangles without normals)
    std::vector<float> tri_s;      // empty, or 9 floats per triangle (zeros for triangles without tangents)
    std::vector<float> env_rgb;
    int env_w = 0, env_h = 0;
    gnxr_camera camera;
    int camera_medium = -1;
    int bvh_split_method = 0;
    Builder();
    int add_mesh(const float *verts, int nv, const int32_t *idx, int nt, const Xf &o2w, int material, int med_in, int med_out);
    void fill_desc(gnxr_scene_desc *d) const;
};

bool read_model_3d(const char *path, std::vector<float> *verts, std::vector<int32_t> *idx);  // shape/plyRead.h:19-48
bool write_synthetic_3d(const char *path, int target_tris, uint32_t seed);
bool read_rgbe(const char *path, std::vector<float> *rgb, int *w, int *h);  // what stbi_loadf returns for a .hdr

// ---------------- compiled scene (host copies of the device tables) ----------------
struct CompiledScene {
    // geometry
    std::vector<DNode> nodes;
    std::vector<DNode4> nodes4;          // two-level collapse of `nodes` (same leaf visiting order)
    int32_t root4 = 0;                   // root reference (a leaf ref when the scene has a single leaf)
    int stack4_need = 1;                 // worst-case traversal stack entries for nodes4
    std::vector<DTri> tris;              // leaf order
    std::vector<float> leaf_boxes;       // 8 floats per leaf-order triangle, valid at the first triangle of each leaf: the leaf's LinearBVHNode bounds (lo.xyz hi.x | hi.yz 0 0)
    int leaf1_from_verts = 0;            // every one-triangle leaf's bounds == min / max of its triangle's vertices (checked in compile_scene)
    std::vector<uint8_t> tri_class;      // per leaf-order triangle: DMaterial::shade_class of its material (0 for null materials): the class the binning pass gives a path that hit it
    std::vector<int32_t> leaf_of_prim;   // authoring index -> leaf index
    int bvh_max_depth = 0;
    Box3 world_bound;
    // shading
    std::vector<DMaterial> materials;
    std::vector<DMaterial> materials_single;   // allowMultipleLobes == false (Whitted)
    std::vector<DTexture> textures;            // image textures: parameters + level offsets into tex_texels
    std::vector<float> tex_texels;             // float4 (rgb_) per texel, all levels of all textures
    std::vector<float> ewa_lut;                // MIPMap::weightLut
    std::vector<float> tri_uv;                 // empty, or 8 floats per leaf-order triangle: (u,v) x 3 corners + pad
    std::vector<float> tri_n;                  // empty, or 12 floats per leaf-order triangle: 3 shading normals + pad (zeros == none)
    std::vector<float> tri_s;                  // the same for TriangleMesh::s (shading tangents)
    std::vector<DLight> lights;
    std::vector<int32_t> infinite_lights;
    // sampler
    std::vector<uint16_t> perms;
    std::vector<int32_t> primes, prime_sums;
    std::vector<uint32_t> prime_magic;   // 8 words per prime: multiplier, shift (exact u32 division), base, permutation offset, 1 / base, perm[0] tail, ceil(2^32 / base), 0
    // env light
    bool has_env = false;
    DEnv env;
    std::vector<float> env_texels;       // Lmap level 0, rgb
    std::vector<float> env_texels4;      // the same as float4 (rgb_) per texel: one dwordx4 per texel of the bilinear lookup on the device
    float env_power_lookup[3] = {0, 0, 0};   // Lmap->Lookup((.5,.5), .5), for InfiniteAreaLight::Power
    std::vector<float> env_cond_func, env_cond_cdf, env_cond_int;   // Distribution2D conditional rows
    std::vector<float> env_marg_func, env_marg_cdf;
    std::vector<uint16_t> env_marg_guide, env_cond_guide;   // FindInterval guide tables (kEnvGuideMarg + 1 entries; per row kEnvGuideCond + 1)
    // media
    std::vector<gnxr_medium> media;
    std::vector<DMedium> dmedia;
    std::vector<float> grid_density;
    std::vector<DSphere> spheres;
    int n_spheres = 0;
    std::vector<int32_t> tri_media;      // leaf order, (inside, outside) per triangle; empty when no triangle is a medium boundary
    // camera description (matrices depend on the render resolution)
    gnxr_camera camera;
    int camera_medium = -1;
    // copy of the description for the light grid builder
    std::vector<gnxr_light> desc_lights;
};

// HLBVH (GNXR_BVH_HLBVH, BVHAccel.cpp:369-626) is built by the caller's device stage (api.hip + hlbvh_build.hip.h): Morton codes, radix
// sort, one LBVH per treelet and the SAH over the treelet roots.  It returns the build tree -- leaves index the sorted primitive array
// (`first`, `n`), interior nodes carry their two children and the split axis -- its root, and the sorted primitive order.
struct HlbvhNode { float b[6]; int32_t child[2]; int32_t axis, first, n; };   // bounds lo.xyz hi.xyz
typedef bool (*HlbvhBuildFn)(const float *prim_bounds6, const float *centroids3, int n, const float lo[3], const float hi[3],
                             std::vector<HlbvhNode> *nodes, int *root, uint32_t *prims_sorted);
bool compile_scene(const gnxr_scene_desc *d, CompiledScene *out, HlbvhBuildFn hlbvh_build = nullptr);
DCamera make_camera(const gnxr_camera &c, int W, int H, int medium);      // camera/Perspective.cpp:114-135, core/Camera.h:54-75
DHalton make_halton(int W, int H);                                          // samplers/HaltonSampler.cpp:33-60
// light-selection table: dense restatement of core/LightDistribution.cpp (uniform / power / spatial)
void build_light_grid(const CompiledScene &cs, int strategy, DLightGrid *grid, std::vector<float> *table, bool layout_only = false);
void light_grid_probes(const CompiledScene &cs, float *ri /* [5][128] */);

// host restatements used by probes and the light grid
float host_radical_inverse(const CompiledScene &cs, int baseIndex, uint64_t a);

}  // namespace gnxr

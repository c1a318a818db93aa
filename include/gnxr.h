/*
 * gnxr.h -- C ABI of the MI355X wavefront path-tracing core (libgnxr.so).
 *
 * This is the drop-in boundary for the reference's render hot path.  The reference
 * (zhouxuguang/GNXRayTracer) has no FFI of its own; its two seams are C++ virtuals:
 *
 *   - Integrator seam : pbr::Integrator::Render(const Scene&, double& timeConsume)
 *                       core/Integrator.h:17-23, called from ui/RenderThread.cpp:175
 *   - Aggregate seam  : pbr::Primitive::Intersect / IntersectP
 *                       core/Primitive.h:13-27, reached via core/Scene.h:32-35
 *
 * A live pbr::Scene cannot be flattened from outside (all members private), so the
 * scene crosses the boundary where it is *authored* (ui/ModelList.cpp, ui/MaterialList.cpp,
 * ui/RenderThread.cpp:46-187) as plain arrays: gnxr_scene_desc below.
 *
 * Conventions: extern "C", plain pointers and sizes, no C++/torch types.  The caller owns
 * every input and output buffer; the library copies what it needs at gnxr_scene_create and
 * owns the device memory behind the opaque handle.  Every entry point returns 0 on success
 * and a negative gnxr_status on failure; gnxr_last_error() returns a thread-local message.
 * Nothing throws across the boundary (the reference builds with exceptions disabled on
 * Apple, CMakeLists.txt:252-253).  There is NO CPU fallback: without a HIP device every
 * compute entry point fails with GNXR_ERR_NO_DEVICE.
 */
#ifndef GNXR_H
#define GNXR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNXR_ABI_VERSION 5   /* 5: gnxr_render_params::passes_in_flight; gnxr_stats::passes_in_flight / loop_iterations / state_bytes */

typedef enum gnxr_status {
    GNXR_OK = 0,
    GNXR_ERR_INVALID = -1,    /* bad argument / inconsistent description            */
    GNXR_ERR_NO_DEVICE = -2,  /* no HIP device (or no such device); there is no CPU fallback */
    GNXR_ERR_OOM = -3,        /* host or device allocation failed                     */
    GNXR_ERR_UNSUPPORTED = -4,/* feature present in the description but not built    */
    GNXR_ERR_IO = -5,         /* file could not be read / written                     */
    GNXR_ERR_RUNTIME = -6     /* a HIP call failed at run time (launch, copy, synchronise); message in gnxr_last_error() */
} gnxr_status;

/* ---- materials: materials/{Matte,Mirror,Glass,Metal,Plastic,Disney}Material.cpp ---- */
typedef enum gnxr_material_type {
    GNXR_MAT_NONE = 0,     /* null material: medium boundary, PathIntegrator.cpp:121-126 */
    GNXR_MAT_MATTE = 1,    /* MatteMaterial.cpp:14-32   (Lambert / OrenNayar)            */
    GNXR_MAT_MIRROR = 2,   /* MirrorMaterial.cpp:13-24                                   */
    GNXR_MAT_GLASS = 3,    /* GlassMaterial.cpp:14-61                                    */
    GNXR_MAT_METAL = 4,    /* MetalMaterial.cpp:28-49                                    */
    GNXR_MAT_PLASTIC = 5,  /* PlasticMaterial.cpp:15-41                                  */
    GNXR_MAT_DISNEY = 6    /* DisneyMaterial.cpp:467-581 (BSSRDF branch out of scope)    */
} gnxr_material_type;

/* Textures are ConstantTexture (textures/ConstantTexture.h:13-23) except Kd / Ks of MATTE and
 * PLASTIC, which may be an ImageTexture (kd_texture / ks_texture below; the reference's
 * getSmileFacePlasticMaterial, ui/MaterialList.cpp:31-46), so a material is a POD of
 * constants plus two texture references.  Field use per type:
 *   MATTE  : kd, sigma (degrees)
 *   MIRROR : kr
 *   GLASS  : kr, kt, eta[0] (index), urough, vrough, remap_roughness
 *   METAL  : eta (rgb), k (rgb), urough, vrough, remap_roughness
 *   PLASTIC: kd, ks, urough (roughness), remap_roughness
 *   DISNEY : kd (color), eta[0], and the disney_* block (DisneyMaterial.h:21-36)
 * bump: every reference material carries a non-null ConstantTexture<float>(0) bump map
 * (ui/RenderThread.cpp:90, ui/MaterialList.cpp:44,54,67,80) so Material::Bump
 * (core/Material.cpp:16-52) runs on every hit; has_bump=1 reproduces that.            */
typedef struct gnxr_material {
    int32_t type;
    int32_t has_bump;
    int32_t remap_roughness;
    int32_t disney_thin;
    float kd[3];
    float ks[3];
    float kr[3];
    float kt[3];
    float eta[3];
    float k[3];
    float sigma;
    float urough;
    float vrough;
    float disney_metallic;
    float disney_spec_trans;
    float disney_spec_tint;
    float disney_sheen;
    float disney_sheen_tint;
    float disney_clearcoat;
    float disney_clearcoat_gloss;
    float disney_anisotropic;
    float disney_roughness;
    float disney_flatness;
    float disney_diff_trans;
    float disney_scatter_distance[3];
    int32_t kd_texture;     /* 1 + index into desc.textures, 0 == the constant kd (MATTE, PLASTIC) */
    int32_t ks_texture;     /* 1 + index into desc.textures, 0 == the constant ks (PLASTIC)        */
} gnxr_material;

/* ---- ImageTexture<RGBSpectrum, Spectrum> + UVMapping2D + MIPMap: textures/ImageTexture.{h,cpp},
 * core/Texture.cpp:163-175, core/MIPMap.h.  The texels cross the boundary decoded (what stbi_loadf
 * returns in ImageTexture.cpp:12-38: row 0 is the TOP row; the library applies the y flip of :79-85,
 * convertIn's scale / inverse gamma, the Lanczos resample to powers of two and the pyramid).  Triangles
 * carry no per-vertex uv anywhere in the reference (ui/ModelList.cpp passes nullptr), so uv is the default
 * (0,0),(1,0),(1,1) of Triangle::GetUVs (shape/Triangle.h:60-74).  Filtering needs the camera ray
 * differentials (camera/Perspective.cpp:86-106, SurfaceInteraction::ComputeDifferentials). ---------- */
typedef enum gnxr_image_wrap { GNXR_WRAP_REPEAT = 0, GNXR_WRAP_BLACK = 1, GNXR_WRAP_CLAMP = 2 } gnxr_image_wrap; /* enum class ImageWrap, core/MIPMap.h:18 */
typedef struct gnxr_texture {
    int32_t width, height;
    int64_t texel_offset;   /* first float of this texture in desc.texels (RGB fp32, width*height*3)  */
    float su, sv, du, dv;   /* UVMapping2D(su, sv, du, dv)                                          */
    float max_aniso;        /* MIPMap::maxAnisotropy (EWA)                                          */
    float scale;            /* convertIn scale                                                      */
    int32_t trilinear;      /* doTrilinear: 0 == EWA                                                */
    int32_t wrap;           /* gnxr_image_wrap                                                      */
    int32_t gamma;          /* convertIn: InverseGammaCorrect                                       */
    int32_t _pad;
} gnxr_texture;

/* ---- lights: lights/{DiffuseAreaLight,InfiniteAreaLight,SkyBoxLight}.cpp ---------- */
typedef enum gnxr_light_type {
    GNXR_LIGHT_AREA_TRI = 1, /* one DiffuseAreaLight per emissive triangle, ModelList.cpp:140-146 */
    GNXR_LIGHT_INFINITE = 2, /* InfiniteAreaLight.cpp:12-132, uses desc.env_*                    */
    GNXR_LIGHT_SKYBOX = 3,   /* SkyBoxLight.cpp:43-85 with a failed image load (gradient)        */
    /* delta lights (LightFlags::DeltaPosition / DeltaDirection): EstimateDirect takes its IsDeltaLight branch, core/Integrator.cpp:
     * 157-158, 168.  The reference authors a spot and a distant light (AddSpotLight / AddDistLight, ui/ModelList.cpp:149-161) but
     * leaves the calls commented out (ui/RenderThread.cpp:138-141).                                                            */
    GNXR_LIGHT_POINT = 4,    /* lights/PointLight.cpp: le = I, position = light_to_world * (0,0,0)                              */
    GNXR_LIGHT_SPOT = 5,     /* lights/SpotLight.cpp: le = I, radius = totalWidth (degrees), falloff_start (degrees), axis +z   */
    GNXR_LIGHT_DISTANT = 6   /* lights/DistantLight.cpp: le = L, center = wLight in light space (Normalize(LightToWorld(w)))    */
} gnxr_light_type;

typedef struct gnxr_light {
    int32_t type;
    int32_t tri;        /* AREA_TRI: index into desc triangles (authoring order)          */
    int32_t two_sided;  /* kept for fidelity; has no effect (DiffuseAreaLight.h:24 quirk) */
    int32_t n_samples;  /* Light::nSamples (core/Light.cpp:19: max(1, n)); used by UniformSampleAllLights only */
    float le[3];        /* AREA_TRI: Lemit ; INFINITE: power scale L                      */
    float radius;       /* SKYBOX: sphere radius                                          */
    float center[3];    /* SKYBOX: sphere centre ; DISTANT: wLight                        */
    float falloff_start;/* SPOT: falloffStart in degrees (radius carries totalWidth)      */
    float light_to_world[16]; /* INFINITE: row-major 4x4 (LightToWorld, ModelList.cpp:174) */
} gnxr_light;

/* ---- camera: camera/Perspective.cpp:114-135 (fov 90, lens 0 in the reference) ------ */
typedef struct gnxr_camera {
    float eye[3];
    float look[3];
    float up[3];
    float fov_deg;
    float lens_radius;
    float focal_distance;
    int32_t orthographic;   /* 0: PerspectiveCamera (camera/Perspective.cpp) ; 1: OrthographicCamera (camera/Orthographic.cpp:
                               Orthographic(0, 10), screen window x 2 as CreateOrthographicCamera sets it; fov_deg unused) */
} gnxr_camera;

/* ---- media: media/{Homogeneous,GridDensity}Medium.cpp (VolPath, config 5) ---------- */
typedef enum gnxr_medium_type { GNXR_MEDIUM_HOMOGENEOUS = 1, GNXR_MEDIUM_GRID = 2 } gnxr_medium_type;

typedef struct gnxr_medium {
    int32_t type;
    int32_t nx, ny, nz;        /* GRID: density grid resolution                            */
    float sigma_a[3];
    float sigma_s[3];
    float g;
    float _pad;
    float medium_to_world[16]; /* GRID: row-major 4x4                                      */
    int64_t density_offset;    /* GRID: first float of this grid in desc.grid_density      */
} gnxr_medium;

/* ---- sphere: shape/Sphere.{h,cpp} is an unfinished stub in the reference (Intersect returns `discriminant > 0` and
 * fills neither tHit nor the interaction), so the quadratic sphere of pbrt-v3 -- whose file it was started from -- is
 * built instead: full sphere, ObjectToWorld = Translate(center).  Parity with the reference is UNPINNED (nothing to
 * pin against); the device is pinned against the CPU restatement and against analytic hits. ------------------------- */
typedef struct gnxr_sphere {
    float center[3];
    float radius;
    int32_t material;       /* index into materials, -1 == null material          */
    int32_t medium_inside;  /* MediumInterface, -1 == none                        */
    int32_t medium_outside;
    int32_t _pad;
} gnxr_sphere;

/* ---- scene description: what ModelList.cpp / RenderThread.cpp author --------------- */
typedef struct gnxr_scene_desc {
    int32_t abi_version;        /* GNXR_ABI_VERSION */
    int32_t n_vertices;
    int32_t n_triangles;
    int32_t n_materials;
    int32_t n_lights;
    int32_t n_media;
    int32_t env_width;          /* INFINITE light: lat-long radiance map (RGB fp32), 0 if none */
    int32_t env_height;
    const float *vertices;      /* n_vertices * 3, WORLD space (TriangleMesh ctor, Triangle.cpp:27-31) */
    const int32_t *indices;     /* n_triangles * 3; triangle order == reference prims order  */
    const int32_t *tri_material;/* n_triangles; index into materials, -1 == null material    */
    const int32_t *tri_light;   /* n_triangles; index into lights (AREA_TRI) or -1           */
    const int32_t *tri_medium_inside;  /* n_triangles or NULL; -1 == none (MediumInterface)  */
    const int32_t *tri_medium_outside; /* n_triangles or NULL                                */
    const gnxr_material *materials;
    const gnxr_light *lights;   /* order == scene.lights order (light selection index)       */
    const gnxr_medium *media;
    const float *grid_density;
    const float *env_rgb;       /* env_width*env_height*3, row-major, as decoded from .hdr   */
    gnxr_camera camera;
    int32_t camera_medium;      /* medium the camera sits in, -1 == none                     */
    int32_t n_spheres;
    const gnxr_sphere *spheres; /* tested before the triangle BVH; gnxr_hit.prim = n_triangles + sphere index */
    int32_t n_textures;
    int32_t _pad;
    const gnxr_texture *textures;
    const float *texels;
    const float *tri_uv;        /* n_triangles * 6 or NULL: (u,v) of each triangle's three corners = TriangleMesh::uv looked up through
                                   the vertex indices (Triangle::GetUVs, shape/Triangle.h:60-74); NULL == the defaults (0,0),(1,0),(1,1)
                                   every mesh of the reference gets (ui/ModelList.cpp passes uv = nullptr)                     */
    const float *tri_n;         /* n_triangles * 9 or NULL: WORLD-space shading normals of each triangle's three corners = TriangleMesh::n
                                   looked up through the vertex indices (shape/Triangle.cpp:228-297: interpolated shading normal,
                                   shading frame, dndu / dndv, geometric normal flipped onto its side); three zero vectors == the
                                   triangle has no normals.  Not allowed on emissive triangles.                                */
    const float *tri_s;         /* n_triangles * 9 or NULL: WORLD-space shading tangents of the corners = TriangleMesh::s (the `ss`
                                   of Triangle.cpp:242-250); three zero vectors == none.  Not allowed on emissive triangles.   */
    int32_t bvh_split_method;   /* gnxr_bvh_split_method: how BVHAccel(prims, 1, splitMethod) builds the tree */
    int32_t _pad2;
} gnxr_scene_desc;

/* enum class SplitMethod, accelerator/BVHAccel.h:24 (same order).  The reference builds BVHAccel(prims, 1) = SAH. */
typedef enum gnxr_bvh_split_method {
    GNXR_BVH_SAH = 0,    /* recursiveBuild with the surface-area heuristic, BVHAccel.cpp:191-367 (host)                          */
    GNXR_BVH_HLBVH = 1,  /* HLBVHBuild, BVHAccel.cpp:369-626: Morton codes + radix sort on the device, LBVH treelets and the SAH
                            upper tree over at most 4096 treelets on the host                                                   */
    GNXR_BVH_MIDDLE = 2, /* recursiveBuild, split at the midpoint of the centroid bounds, BVHAccel.cpp:243-258 (host)            */
    GNXR_BVH_EQUAL_COUNTS = 3 /* recursiveBuild, nth_element at the median, BVHAccel.cpp:259-268 (host)                          */
} gnxr_bvh_split_method;

typedef enum gnxr_integrator {
    GNXR_INTEGRATOR_PATH = 0,    /* integrators/PathIntegrator.cpp:62-208   */
    GNXR_INTEGRATOR_VOLPATH = 1, /* integrators/VolPathIntegrator.cpp:24-159 */
    GNXR_INTEGRATOR_WHITTED = 2, /* integrators/WhittedIntegrator.cpp:14-68 */
    GNXR_INTEGRATOR_DIRECT = 3   /* integrators/DirectLightingIntegrator.cpp:11-67 */
} gnxr_integrator;

/* enum class LightStrategy, integrators/DirectLightingIntegrator.h:13 (same order) */
typedef enum gnxr_direct_strategy {
    GNXR_DIRECT_SAMPLE_ALL = 0, /* UniformSampleAllLights, core/Integrator.cpp:25-55: nSamples array samples per light */
    GNXR_DIRECT_SAMPLE_ONE = 1  /* UniformSampleOneLight without a distribution, core/Integrator.cpp:57-79          */
} gnxr_direct_strategy;

typedef enum gnxr_light_strategy {
    GNXR_LIGHTS_SPATIAL = 0, /* core/LightDistribution.cpp:70-274 */
    GNXR_LIGHTS_UNIFORM = 1,
    GNXR_LIGHTS_POWER = 2
} gnxr_light_strategy;

/* Render parameters.  A render covers samples [spp_begin, spp_end) of a HaltonSampler(spp)
 * (samplers/HaltonSampler.cpp:33-60) for the image rows this shard owns:
 * row y belongs to the shard iff (y / shard_rows) % shard_count == shard_index.
 * shard_count=1 renders the whole image.                                                  */
typedef struct gnxr_render_params {
    int32_t width, height;
    int32_t spp;                /* HaltonSampler samplesPerPixel; divisor of the box average */
    int32_t spp_begin, spp_end; /* sample range rendered by this call; 0,spp == all          */
    int32_t max_depth;          /* PathIntegrator maxDepth                                   */
    float rr_threshold;         /* PathIntegrator rrThreshold                                */
    int32_t integrator;         /* gnxr_integrator                                           */
    int32_t light_strategy;     /* gnxr_light_strategy                                       */
    int32_t shard_index, shard_count, shard_rows;
    int32_t samples_per_pass;   /* 0 = auto; samples of one pixel rendered per (sub-)pass.  The image does not depend on it */
    int32_t direct_strategy;    /* gnxr_direct_strategy (GNXR_INTEGRATOR_DIRECT only)        */
    int32_t passes_in_flight;   /* GNXR_INTEGRATOR_PATH: sub-passes alive at once, each in its own region of the path state
                                   (resident state = passes_in_flight x samples_per_pass samples per pixel); staggered in time,
                                   so that a launch mixes the first bounces of one sub-pass with the thin late bounces of the
                                   others.  0 = auto (4, fewer if the call is short or memory is tight), at most 8.  The image
                                   does not depend on it.  Reference loop: core/Integrator.cpp:256-293 */
} gnxr_render_params;

typedef struct gnxr_stats {
    uint64_t rays_closest;      /* Scene::Intersect calls  (core/Scene.cpp:11-17)            */
    uint64_t rays_any;          /* Scene::IntersectP calls (core/Scene.cpp:19-24)            */
    uint64_t camera_samples;
    uint64_t nodes_visited;     /* filled only when GNXR_STATS_TRAVERSAL is requested        */
    uint64_t tris_tested;
    double seconds_render;      /* kernel pipeline, excludes scene build/upload              */
    double seconds_trace;       /* sum of HIP-event time of the traversal kernels            */
    double seconds_total;
    uint32_t kernel_launches;
    uint32_t passes;
    double seconds_closest;     /* k_closest launches (profiling bit 0)                      */
    double seconds_nee;         /* k_nee launches                                            */
    double seconds_shade;       /* k_shade launches                                          */
    uint32_t launches_closest, launches_nee;
    uint64_t rays_closest_nee;  /* closest-hit rays traced by k_nee (MIS rays)               */
    uint64_t media_segments;    /* VolPath: ray segments handed to the tracking kernel (Medium::Sample / Medium::Tr calls on rays inside a medium) */
    uint64_t media_steps;       /* VolPath: tracking-loop iterations of those segments (filled by the counting run, profiling bit 2) */
    uint64_t leaf_retests;      /* counting run, bit 2: leaf boxes re-tested against a shrunken tMax by the 4-wide walk (32 B each)  */
    uint64_t nodes_from_memory; /* counting run, bit 2: 4-wide node visits that read global memory (the rest hit the kernel's LDS copy of the top of the tree) */
    uint32_t passes_in_flight;  /* sub-passes the path loop kept alive at once (1 for the other integrators)                          */
    uint32_t loop_iterations;   /* iterations of the path loop (one trace + one shade stage each)                                     */
    uint64_t state_bytes;       /* path state resident on the device for this render (per-path arrays and queues)                      */
} gnxr_stats;

typedef struct gnxr_ray { float o[3]; float tmax; float d[3]; float _pad; } gnxr_ray;
typedef struct gnxr_hit {
    int32_t prim;               /* triangle index (authoring order) or -1                    */
    float t, b0, b1, b2;
    float n[3];                 /* geometric normal as set by Triangle::Intersect            */
} gnxr_hit;

typedef struct gnxr_scene gnxr_scene;

/* -- lifecycle ---------------------------------------------------------------------- */
int gnxr_abi_version(void);
int gnxr_abi_sizeof(int which);        /* sizeof of the structs above in declaration order (binding self-check) */
int gnxr_init(int device_id);          /* binds the calling process to one HIP device    */
/* One process, several devices (`gnxr_init(int n_devices, const int *device_ids)`): scenes created afterwards are
 * replicated on every listed device and gnxr_render / gnxr_render_device deal the image rows round-robin over them, render the
 * shards concurrently (one host thread + stream per device, no exchange during rendering) and assemble the FrameBuffer on
 * device_ids[0] -- one strided peer copy per device where peer access could be enabled both ways (recorded per pair at init), a
 * pinned host buffer otherwise; GNXR_NO_PEER=1 forces the staged route -- what the reference's single `integrator->Render(scene)` call
 * (ui/RenderThread.cpp:175, core/Integrator.h:17-23) needs to use a whole node.  Results are bit-identical to one device's
 * (pixels are independent).  The same id may be listed more than once (two shards sharing a device: how this path is tested on
 * a one-GPU box; on DISTINCT devices the path has not run yet: parity unpinned there, the pool hands out one-GPU boxes only).
 * Batched trace calls and probes run on device_ids[0].  gnxr_init(d) == gnxr_init_devices(1, &d).                              */
int gnxr_init_devices(int32_t n_devices, const int32_t *device_ids);
void gnxr_shutdown(void);
const char *gnxr_last_error(void);
/* Measurement switches (process-wide).  bit 0: bracket every traversal kernel launch with HIP events on
 * the render stream and report their summed duration in gnxr_stats.seconds_trace (+ per-kernel split in
 * seconds_closest / seconds_nee); bit 1: run the counting variant of the traversal kernel on the reference's BINARY
 * tree (BVHAccel::Intersect's own node visits: comparable with the oracle's counts) and fill nodes_visited /
 * tris_tested; bit 2: count on the walk the timed kernel performs instead -- 4-wide nodes visited, speculative visits
 * included -- and the tracking-loop iterations of the medium kernel (media_steps).  Counting runs are slower; never
 * combine them with a timed run.                                                                             */
int gnxr_set_profiling(int flags);
/* Measurement hook: the VALU issue rate of this device, in 1e9 wave64 instructions per second, reached by a kernel of
 * independent v_fma_f32 chains with 8 waves on every SIMD -- the ceiling bench.py prices the traversal kernel's
 * instruction stream against.                                                                                */
int gnxr_probe_valu_peak(double *giga_wave_insts_per_s);
/* Measurement hook: the rate at which this device takes per-lane 16-byte gathers (every lane reads the 8 dwordx4 of its own random
 * 128-byte record of an 8 MB table: the access pattern of a BVH node visit), in 1e9 lane-loads per second.  On MI355X this rate
 * (~690 G/s = 1.1 lanes per clock per CU) is the same for L1-resident and L2-resident tables and at 1 to 8 blocks per CU
 * (tools/probes/gather_probe.hip): it is a ceiling of the vector-memory path, beside HBM bandwidth and VALU issue.              */
int gnxr_probe_gather_peak(double *giga_lane_loads_per_s);

/* -- scene (replaces `Scene(make_shared<BVHAccel>(prims,1), lights)`, RenderThread.cpp:155) */
int gnxr_scene_create(const gnxr_scene_desc *desc, gnxr_scene **out);
void gnxr_scene_destroy(gnxr_scene *scene);
/* Test hook: the flattened binary BVH (the reference's LinearBVHNode[]: per node 6 floats of bounds into `bounds6`, then
 * offset / nPrimitives / axis into `meta3`) and the primitive order (`ordered`, n_triangles entries); *n_nodes receives the node
 * count, arrays are filled when node_capacity allows. */
int gnxr_scene_bvh(const gnxr_scene *scene, float *bounds6, int32_t *meta3, int32_t *ordered, int64_t node_capacity, int64_t *n_nodes);
int gnxr_scene_info(const gnxr_scene *scene, int32_t *n_bvh_nodes, int32_t *bvh_max_depth,
                    int32_t *n_light_voxels);

/* -- Integrator seam (replaces integrator->Render(*worldScene, frameTime), RenderThread.cpp:175).
 * rgba_out: width*height*4 fp32, row-major, pixel (x,y) at (x + y*width)*4, the layout of
 * FrameBuffer::fbuffer (ui/FrameBuffer.h:136); A is written as 1.  With shard_count>1 only
 * the shard's rows are written.                                                          */
int gnxr_render(gnxr_scene *scene, const gnxr_render_params *params, float *rgba_out,
                gnxr_stats *stats);
/* Same, output stays in device memory (d_rgba_out is a HIP device pointer, e.g. the data_ptr
 * of a torch tensor) and the work is enqueued on `hip_stream` (a hipStream_t, may be NULL). */
int gnxr_render_device(gnxr_scene *scene, const gnxr_render_params *params, void *d_rgba_out,
                       void *hip_stream, gnxr_stats *stats);

/* Allocates, on every device of the handle, the path state a render with these parameters needs, without rendering anything: the
 * first gnxr_render / gnxr_render_device call with them then runs at its steady-state speed (the reference has no counterpart: its
 * per-thread MemoryArena grows inside Render, core/Integrator.cpp:262).  State buffers only ever grow; they are released with the scene. */
int gnxr_render_reserve(gnxr_scene *scene, const gnxr_render_params *params);

/* -- Aggregate seam (replaces Scene::Intersect / Scene::IntersectP), batched ------------ */
int gnxr_trace_closest(gnxr_scene *scene, const gnxr_ray *rays, int64_t n, gnxr_hit *hits);
int gnxr_trace_any(gnxr_scene *scene, const gnxr_ray *rays, int64_t n, uint8_t *occluded);

/* -- sampler / camera probes (bit-exactness test hooks) --------------------------------- */
/* HaltonSampler(spp, [0,width)x[0,height)) value of dimension dim[i] for sample s[i] of pixel
 * (px[i],py[i]) -- GetIndexForSample + SampleDimension, HaltonSampler.cpp:63-94.            */
int gnxr_sample_halton(int32_t width, int32_t height, const int32_t *px, const int32_t *py,
                       const int64_t *s, const int32_t *dim, int64_t n, float *out);
/* Camera rays (Perspective.cpp:62-112 + Integrator.cpp:277-283): for sample s of pixel (px,py)
 * writes origin[3], direction[3] per ray.                                                    */
int gnxr_camera_rays(const gnxr_camera *cam, int32_t width, int32_t height, const int32_t *px,
                     const int32_t *py, const int64_t *s, int64_t n, float *o_out, float *d_out);

/* FrameBuffer::saveToFile (ui/FrameBuffer.cpp:6-9: stbi_write_png of the RGBA8 plane): writes `rgba8` (width * height * 4
 * bytes, row-major, as gnxr_framebuffer_update produces it) as an 8-bit RGBA PNG.  Host-only; the file decodes to the same
 * pixels as the reference's (the zlib stream itself is not stb's). */
int gnxr_framebuffer_save_png(const char *path, const uint8_t *rgba8, int32_t width, int32_t height);

/* Unit-test hook: the light-selection table of `strategy` (core/LightDistribution.cpp; per voxel cdf[1..n], func[0..n-1],
 * funcInt), built on the device (on_host == 0) or by the host restatement (on_host != 0).  *n_floats receives the table
 * size; the table is copied when `out` has room for it. */
int gnxr_light_grid_table(gnxr_scene *s, int32_t strategy, int32_t on_host, float *out, int64_t capacity, int64_t *n_floats);

/* Unit-test hook: the device's float libm on caller-supplied arguments.  fn: 0 logf, 1 expf, 2 sinf, 3 cosf, 4 / 5 sinf / cosf
 * through the shared-reduction pair evaluation, 6 acosf, 7 atan2f(x, x2), 8 powf(x, x2) (x2 may be NULL otherwise).  The
 * reference reaches these through std::log / std::exp / std::sin / std::cos on floats (core/Sampling.cpp:87-105,
 * media/GridDensityMedium.cpp:41,67, media/HomogeneousMedium.cpp:13,24, core/Medium.cpp:187, core/Geometry.h:1436-1443);
 * the device restates glibc 2.35's algorithms so the results carry glibc's bits. */
int gnxr_eval_libm(int32_t fn, const float *x, const float *x2, int64_t n, float *out);
/* The double-precision libm calls of the path, on float arguments widened to double, results as doubles.  fn: 0 sin, 1 cos (the pair
 * the device evaluates for `r * cos(phi)`, `r * sin(phi)` at core/MicroFacet.cpp:220-223, where the unqualified calls bind to the
 * double versions), 2 sqrt (MicroFacet.cpp:220, DisneyMaterial.cpp:236), 3 tan (MicroFacet.cpp:190-191, 297). */
int gnxr_eval_libm_f64(int32_t fn, const float *x, int64_t n, double *out);

/* -- output stage (FrameBuffer::update_f_u_c, ui/FrameBuffer.h:127-149) ------------------ */
/* Folds one Render() result into the running mean of `frame_count` previous frames and
 * tone-maps 1-exp(-x/0.25) to RGBA8 (A=255).  Device-side; host pointers in and out.        */
int gnxr_framebuffer_update(float *running_mean_rgba, const float *frame_rgba, int32_t width,
                            int32_t height, int32_t frame_count, uint8_t *rgba8_out);

/* -- host-side scene authoring (mirror of ui/ModelList.cpp, ui/MaterialList.cpp) -------- */
typedef struct gnxr_builder gnxr_builder;
int gnxr_builder_create(gnxr_builder **out);
void gnxr_builder_destroy(gnxr_builder *b);
/* material factories return a material index */
int gnxr_builder_add_material(gnxr_builder *b, const gnxr_material *m);
int gnxr_builder_matte(gnxr_builder *b, const float kd[3], float sigma_deg);          /* RenderThread.cpp:79-99 */
int gnxr_builder_mirror(gnxr_builder *b, const float kr[3]);                          /* RenderThread.cpp:102   */
int gnxr_builder_purple_plastic(gnxr_builder *b);                                     /* MaterialList.cpp:48-56 */
int gnxr_builder_yellow_metal(gnxr_builder *b);                                       /* MaterialList.cpp:58-69 */
int gnxr_builder_white_glass(gnxr_builder *b);                                        /* MaterialList.cpp:71-83 */
/* geometry: each returns the index of the first triangle added, or <0 */
int gnxr_builder_add_mesh(gnxr_builder *b, const float *vertices, int32_t n_vertices,
                          const int32_t *indices, int32_t n_triangles, const float *object_to_world16,
                          int32_t material, int32_t medium_inside, int32_t medium_outside);
int gnxr_builder_add_model_3d(gnxr_builder *b, const char *path, int32_t material);   /* ModelList.cpp:47-69, plyRead.h:19-48 */
int gnxr_builder_add_cornell(gnxr_builder *b, int32_t material1, int32_t material2,
                             int32_t material3);                                      /* ModelList.cpp:71-118 */
int gnxr_builder_add_floor(gnxr_builder *b, int32_t material);                        /* ModelList.cpp:20-45  */
int gnxr_builder_add_area_light(gnxr_builder *b, int32_t material);                   /* ModelList.cpp:120-147 */
/* AddAreaLight's pattern for any mesh: one DiffuseAreaLight(Lemit, nSamples, triangle, twoSided = false) per triangle
 * (ModelList.cpp:137-146); returns the first triangle index */
int gnxr_builder_add_emissive_mesh(gnxr_builder *b, const float *vertices, int32_t n_vertices, const int32_t *indices, int32_t n_triangles,
                                   const float *object_to_world16, int32_t material, const float lemit[3], int32_t n_samples);
int gnxr_builder_add_sky_light(gnxr_builder *b);                                      /* ModelList.cpp:163-170 */
int gnxr_builder_add_spot_light(gnxr_builder *b);                                     /* AddSpotLight, ModelList.cpp:149-154 */
int gnxr_builder_add_dist_light(gnxr_builder *b);                                     /* AddDistLight, ModelList.cpp:156-161 */
int gnxr_builder_add_light(gnxr_builder *b, const gnxr_light *l);                     /* POINT / SPOT / DISTANT with caller-chosen parameters */
int gnxr_builder_add_inf_light(gnxr_builder *b, const char *hdr_path);                /* ModelList.cpp:172-179 */
int gnxr_builder_add_inf_light_data(gnxr_builder *b, const float *rgb, int32_t w, int32_t h,
                                    const float *light_to_world16, const float power[3]);
int gnxr_builder_add_medium(gnxr_builder *b, const gnxr_medium *m, const float *density);
/* GridDensityMedium from a `.volume` text file (the format of the reference's Resources/density_render.70.volume: `nx N ny N nz N`,
 * `p0 x y z`, `p1 x y z`, `sigma_a r g b`, `sigma_s r g b`, then nx*ny*nz densities): sigma_a / sigma_s of the header times
 * sigma_scale, Henyey-Greenstein g, MediumToWorld = medium_to_world16 or, when NULL, Translate(p0) * Scale(p1 - p0) of the header.
 * The reference ships the file but no reader for it, so the file semantics (x fastest, the p0 / p1 placement) are THIS project's
 * definition -- parity unpinned for the loader, pinned for the medium it builds.  Each dimension <= 4096, at most 2^31 - 1 values;
 * GNXR_ERR_IO for a malformed file, GNXR_ERR_OOM when the values do not fit in memory.  Returns the medium index.                  */
int gnxr_builder_add_volume_file(gnxr_builder *b, const char *path, float g, float sigma_scale, const float *medium_to_world16);
/* ImageTexture: `t` carries the mapping / filter parameters (width, height, texel_offset are filled in); returns the texture
 * index.  _file decodes a Radiance .hdr like stbi_loadf; other formats (the reference's awesomeface.jpg) must be decoded by the
 * caller and passed as texels. */
int gnxr_builder_add_texture_data(gnxr_builder *b, const gnxr_texture *t, const float *rgb, int32_t w, int32_t h);
int gnxr_builder_add_texture_file(gnxr_builder *b, const gnxr_texture *t, const char *hdr_path);
int gnxr_builder_set_material_texture(gnxr_builder *b, int32_t material, int32_t slot /* 0 Kd, 1 Ks */, int32_t texture);
/* per-corner (u,v) of triangles [first_triangle, first_triangle + n_triangles): tri_uv holds n_triangles * 6 floats */
int gnxr_builder_set_triangle_uv(gnxr_builder *b, int32_t first_triangle, int32_t n_triangles, const float *tri_uv);
/* per-corner WORLD-space shading normals (n_triangles * 9 floats); a caller with object-space normals applies the mesh's
 * ObjectToWorld as TriangleMesh's constructor does (Transform::operator()(Normal3f): the inverse transpose) */
int gnxr_builder_set_triangle_normals(gnxr_builder *b, int32_t first_triangle, int32_t n_triangles, const float *tri_n);
int gnxr_builder_set_triangle_tangents(gnxr_builder *b, int32_t first_triangle, int32_t n_triangles, const float *tri_s);   /* TriangleMesh::s, world space */
int gnxr_builder_add_sphere(gnxr_builder *b, const float center[3], float radius, int32_t material, int32_t medium_inside,
                            int32_t medium_outside);                                      /* returns the sphere index */
int gnxr_builder_set_camera(gnxr_builder *b, const gnxr_camera *cam);
int gnxr_builder_set_bvh_split_method(gnxr_builder *b, int32_t method);               /* gnxr_bvh_split_method */
int gnxr_builder_set_camera_medium(gnxr_builder *b, int32_t medium);                  /* Camera::medium, core/Camera.h; -1 == none */                 /* RenderThread.cpp:60-68 */
/* The returned description points into builder-owned memory, valid until the next builder
 * call or gnxr_builder_destroy.                                                            */
int gnxr_builder_desc(gnxr_builder *b, gnxr_scene_desc *out);

/* Deterministic synthetic stand-in for the absent Resources/dragon.3d (.MISSING_LARGE_BLOBS:1):
 * writes a `.3d` text file in the format plyRead.h:19-48 parses.                           */
int gnxr_write_synthetic_3d(const char *path, int32_t target_triangles, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif /* GNXR_H */

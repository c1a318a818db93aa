"""ctypes mirror of include/gnxr.h (the C ABI of libgnxr.so).

Only data layout and prototypes live here; no compute.  The structures must match
include/gnxr.h field for field (tests/test_abi.py checks sizes and exported symbols).
"""
import ctypes as C

GNXR_ABI_VERSION = 5

# gnxr_status
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_OOM, ERR_UNSUPPORTED, ERR_IO, ERR_RUNTIME = 0, -1, -2, -3, -4, -5, -6
# gnxr_material_type
MAT_NONE, MAT_MATTE, MAT_MIRROR, MAT_GLASS, MAT_METAL, MAT_PLASTIC, MAT_DISNEY = range(7)
# gnxr_light_type
LIGHT_AREA_TRI, LIGHT_INFINITE, LIGHT_SKYBOX, LIGHT_POINT, LIGHT_SPOT, LIGHT_DISTANT = 1, 2, 3, 4, 5, 6
# gnxr_integrator / gnxr_light_strategy
INTEGRATOR_PATH, INTEGRATOR_VOLPATH, INTEGRATOR_WHITTED, INTEGRATOR_DIRECT = 0, 1, 2, 3
DIRECT_SAMPLE_ALL, DIRECT_SAMPLE_ONE = 0, 1
LIGHTS_SPATIAL, LIGHTS_UNIFORM, LIGHTS_POWER = 0, 1, 2
MEDIUM_HOMOGENEOUS, MEDIUM_GRID = 1, 2

f32, i32, u8, i64, u32, u64 = C.c_float, C.c_int32, C.c_uint8, C.c_int64, C.c_uint32, C.c_uint64


class Material(C.Structure):
    _fields_ = [
        ("type", i32), ("has_bump", i32), ("remap_roughness", i32), ("disney_thin", i32),
        ("kd", f32 * 3), ("ks", f32 * 3), ("kr", f32 * 3), ("kt", f32 * 3), ("eta", f32 * 3), ("k", f32 * 3),
        ("sigma", f32), ("urough", f32), ("vrough", f32),
        ("disney_metallic", f32), ("disney_spec_trans", f32), ("disney_spec_tint", f32), ("disney_sheen", f32),
        ("disney_sheen_tint", f32), ("disney_clearcoat", f32), ("disney_clearcoat_gloss", f32),
        ("disney_anisotropic", f32), ("disney_roughness", f32), ("disney_flatness", f32), ("disney_diff_trans", f32),
        ("disney_scatter_distance", f32 * 3), ("kd_texture", i32), ("ks_texture", i32),
    ]


class Light(C.Structure):
    _fields_ = [
        ("type", i32), ("tri", i32), ("two_sided", i32), ("n_samples", i32),
        ("le", f32 * 3), ("radius", f32), ("center", f32 * 3), ("falloff_start", f32), ("light_to_world", f32 * 16),
    ]


class Camera(C.Structure):
    _fields_ = [("eye", f32 * 3), ("look", f32 * 3), ("up", f32 * 3), ("fov_deg", f32), ("lens_radius", f32),
                ("focal_distance", f32), ("orthographic", i32)]


class Medium(C.Structure):
    _fields_ = [("type", i32), ("nx", i32), ("ny", i32), ("nz", i32), ("sigma_a", f32 * 3), ("sigma_s", f32 * 3),
                ("g", f32), ("_pad", f32), ("medium_to_world", f32 * 16), ("density_offset", i64)]


class Sphere(C.Structure):
    _fields_ = [("center", f32 * 3), ("radius", f32), ("material", i32), ("medium_inside", i32), ("medium_outside", i32), ("_pad", i32)]


class Texture(C.Structure):
    _fields_ = [("width", i32), ("height", i32), ("texel_offset", i64), ("su", f32), ("sv", f32), ("du", f32), ("dv", f32),
                ("max_aniso", f32), ("scale", f32), ("trilinear", i32), ("wrap", i32), ("gamma", i32), ("_pad", i32)]


WRAP_REPEAT, WRAP_BLACK, WRAP_CLAMP = 0, 1, 2


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", i32), ("n_vertices", i32), ("n_triangles", i32), ("n_materials", i32), ("n_lights", i32),
        ("n_media", i32), ("env_width", i32), ("env_height", i32),
        ("vertices", C.POINTER(f32)), ("indices", C.POINTER(i32)), ("tri_material", C.POINTER(i32)),
        ("tri_light", C.POINTER(i32)), ("tri_medium_inside", C.POINTER(i32)), ("tri_medium_outside", C.POINTER(i32)),
        ("materials", C.POINTER(Material)), ("lights", C.POINTER(Light)), ("media", C.POINTER(Medium)),
        ("grid_density", C.POINTER(f32)), ("env_rgb", C.POINTER(f32)),
        ("camera", Camera), ("camera_medium", i32), ("n_spheres", i32), ("spheres", C.POINTER(Sphere)),
        ("n_textures", i32), ("_pad", i32), ("textures", C.POINTER(Texture)), ("texels", C.POINTER(f32)),
        ("tri_uv", C.POINTER(f32)), ("tri_n", C.POINTER(f32)), ("tri_s", C.POINTER(f32)),
        ("bvh_split_method", i32), ("_pad2", i32),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", i32), ("height", i32), ("spp", i32), ("spp_begin", i32), ("spp_end", i32), ("max_depth", i32),
        ("rr_threshold", f32), ("integrator", i32), ("light_strategy", i32),
        ("shard_index", i32), ("shard_count", i32), ("shard_rows", i32), ("samples_per_pass", i32), ("direct_strategy", i32),
        ("passes_in_flight", i32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays_closest", u64), ("rays_any", u64), ("camera_samples", u64), ("nodes_visited", u64), ("tris_tested", u64),
        ("seconds_render", C.c_double), ("seconds_trace", C.c_double), ("seconds_total", C.c_double),
        ("kernel_launches", u32), ("passes", u32),
        ("seconds_closest", C.c_double), ("seconds_nee", C.c_double), ("seconds_shade", C.c_double),
        ("launches_closest", u32), ("launches_nee", u32), ("rays_closest_nee", u64),
        ("media_segments", u64), ("media_steps", u64), ("leaf_retests", u64), ("nodes_from_memory", u64),
        ("passes_in_flight", u32), ("loop_iterations", u32), ("state_bytes", u64),
    ]


class Ray(C.Structure):
    _fields_ = [("o", f32 * 3), ("tmax", f32), ("d", f32 * 3), ("_pad", f32)]


class Hit(C.Structure):
    _fields_ = [("prim", i32), ("t", f32), ("b0", f32), ("b1", f32), ("b2", f32), ("n", f32 * 3)]


P = C.POINTER
VP = C.c_void_p

# name -> (restype, argtypes); every symbol include/gnxr.h declares
PROTOTYPES = {
    "gnxr_abi_version": (C.c_int, []),
    "gnxr_abi_sizeof": (C.c_int, [C.c_int]),
    "gnxr_init": (C.c_int, [C.c_int]),
    "gnxr_shutdown": (None, []),
    "gnxr_last_error": (C.c_char_p, []),
    "gnxr_init_devices": (C.c_int, [i32, P(i32)]),
    "gnxr_set_profiling": (C.c_int, [C.c_int]),
    "gnxr_probe_valu_peak": (C.c_int, [P(C.c_double)]),
    "gnxr_probe_gather_peak": (C.c_int, [P(C.c_double)]),
    "gnxr_scene_create": (C.c_int, [P(SceneDesc), P(VP)]),
    "gnxr_scene_destroy": (None, [VP]),
    "gnxr_scene_info": (C.c_int, [VP, P(i32), P(i32), P(i32)]),
    "gnxr_render": (C.c_int, [VP, P(RenderParams), P(f32), P(Stats)]),
    "gnxr_render_device": (C.c_int, [VP, P(RenderParams), VP, VP, P(Stats)]),
    "gnxr_render_reserve": (C.c_int, [VP, P(RenderParams)]),
    "gnxr_trace_closest": (C.c_int, [VP, P(Ray), i64, P(Hit)]),
    "gnxr_trace_any": (C.c_int, [VP, P(Ray), i64, P(u8)]),
    "gnxr_sample_halton": (C.c_int, [i32, i32, P(i32), P(i32), P(i64), P(i32), i64, P(f32)]),
    "gnxr_camera_rays": (C.c_int, [P(Camera), i32, i32, P(i32), P(i32), P(i64), i64, P(f32), P(f32)]),
    "gnxr_framebuffer_update": (C.c_int, [P(f32), P(f32), i32, i32, i32, P(u8)]),
    "gnxr_framebuffer_save_png": (C.c_int, [C.c_char_p, P(u8), i32, i32]),
    "gnxr_light_grid_table": (C.c_int, [VP, i32, i32, P(f32), i64, P(i64)]),
    "gnxr_eval_libm": (C.c_int, [i32, P(f32), P(f32), i64, P(f32)]),
    "gnxr_eval_libm_f64": (C.c_int, [i32, P(f32), i64, P(C.c_double)]),
    "gnxr_builder_create": (C.c_int, [P(VP)]),
    "gnxr_builder_set_camera_medium": (C.c_int, [VP, i32]),
    "gnxr_builder_add_sphere": (C.c_int, [VP, P(f32), f32, i32, i32, i32]),
    "gnxr_builder_destroy": (None, [VP]),
    "gnxr_builder_add_material": (C.c_int, [VP, P(Material)]),
    "gnxr_builder_matte": (C.c_int, [VP, P(f32), f32]),
    "gnxr_builder_mirror": (C.c_int, [VP, P(f32)]),
    "gnxr_builder_purple_plastic": (C.c_int, [VP]),
    "gnxr_builder_yellow_metal": (C.c_int, [VP]),
    "gnxr_builder_white_glass": (C.c_int, [VP]),
    "gnxr_builder_add_mesh": (C.c_int, [VP, P(f32), i32, P(i32), i32, P(f32), i32, i32, i32]),
    "gnxr_builder_add_model_3d": (C.c_int, [VP, C.c_char_p, i32]),
    "gnxr_builder_add_cornell": (C.c_int, [VP, i32, i32, i32]),
    "gnxr_builder_add_floor": (C.c_int, [VP, i32]),
    "gnxr_builder_add_area_light": (C.c_int, [VP, i32]),
    "gnxr_builder_add_sky_light": (C.c_int, [VP]),
    "gnxr_builder_add_inf_light": (C.c_int, [VP, C.c_char_p]),
    "gnxr_builder_add_inf_light_data": (C.c_int, [VP, P(f32), i32, i32, P(f32), P(f32)]),
    "gnxr_builder_add_medium": (C.c_int, [VP, P(Medium), P(f32)]),
    "gnxr_builder_add_volume_file": (C.c_int, [VP, C.c_char_p, f32, f32, P(f32)]),
    "gnxr_builder_add_emissive_mesh": (C.c_int, [VP, P(f32), i32, P(i32), i32, P(f32), i32, P(f32), i32]),
    "gnxr_builder_add_spot_light": (C.c_int, [VP]),
    "gnxr_builder_add_dist_light": (C.c_int, [VP]),
    "gnxr_builder_add_light": (C.c_int, [VP, P(Light)]),
    "gnxr_builder_add_texture_data": (C.c_int, [VP, P(Texture), P(f32), i32, i32]),
    "gnxr_builder_add_texture_file": (C.c_int, [VP, P(Texture), C.c_char_p]),
    "gnxr_builder_set_material_texture": (C.c_int, [VP, i32, i32, i32]),
    "gnxr_builder_set_triangle_uv": (C.c_int, [VP, i32, i32, P(f32)]),
    "gnxr_builder_set_triangle_normals": (C.c_int, [VP, i32, i32, P(f32)]),
    "gnxr_builder_set_triangle_tangents": (C.c_int, [VP, i32, i32, P(f32)]),
    "gnxr_builder_set_bvh_split_method": (C.c_int, [VP, i32]),
    "gnxr_scene_bvh": (C.c_int, [VP, P(f32), P(i32), P(i32), i64, P(i64)]),
    "gnxr_builder_set_camera": (C.c_int, [VP, P(Camera)]),
    "gnxr_builder_desc": (C.c_int, [VP, P(SceneDesc)]),
    "gnxr_write_synthetic_3d": (C.c_int, [C.c_char_p, i32, u32]),
}


ABI_STRUCTS = [Material, Light, Camera, Medium, SceneDesc, RenderParams, Stats, Ray, Hit, Sphere, Texture]


def bind(lib):
    """Attach prototypes; raises AttributeError naming the first missing symbol."""
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib

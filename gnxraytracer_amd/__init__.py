"""gnxraytracer_amd -- MI355X-native wavefront path-tracing core (host-side Python mirror).

This package is a thin ctypes layer over ``libgnxr.so`` (HIP kernels + C ABI, ``include/gnxr.h``).
Names follow the reference's authoring layer so that tests read like the reference's own scene
set-up (ui/ModelList.cpp, ui/MaterialList.cpp, ui/RenderThread.cpp:46-187):

    b = SceneBuilder()
    white = b.MatteMaterial((0.91, 0.91, 0.91), sigma=60)
    ...
    b.AddCornell(red, blue, white); b.AddAreaLight(white)
    scene = Scene(b)                                   # Scene(make_shared<BVHAccel>(prims, 1), lights)
    img, stats = PathIntegrator(8, rrThreshold=1.0).Render(scene, 256, 256, spp=64)

There is NO CPU fallback: every compute call goes through the HIP library and raises
``GnxrError`` if it (or a GPU) is missing.  The CPU restatement under ``oracle/`` is test
infrastructure and is never imported from here.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import (Camera, Hit, Light, Material, Medium, Ray, RenderParams, SceneDesc, Stats, Texture)  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GNXR_LIB", os.path.join(_HERE, "libgnxr.so"))   # GNXR_LIB: A/B builds during tuning


class GnxrError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libgnxr.so (built by ``__graft_entry__.build()``); fail loudly when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GnxrError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _lib = _abi.bind(C.CDLL(LIB_PATH))
        if _lib.gnxr_abi_version() != _abi.GNXR_ABI_VERSION:
            raise GnxrError("libgnxr.so ABI version mismatch")
    return _lib


def _check(rc):
    if rc < 0:
        raise GnxrError(f"gnxr error {rc}: {lib().gnxr_last_error().decode(errors='replace')}")
    return rc


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def init(device_id=0):
    _check(lib().gnxr_init(int(device_id)))


def init_devices(device_ids):
    """One process, several devices: scenes created afterwards are replicated on all of them and Render() shards the image rows
    over them (gnxr_init_devices)."""
    ids = (C.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
    _check(lib().gnxr_init_devices(len(device_ids), ids))


class SceneBuilder:
    """Mirror of the scene-authoring free functions in ui/ModelList.cpp / ui/MaterialList.cpp."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(lib().gnxr_builder_create(C.byref(self._h)))
        self.hdr_path = None

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.gnxr_builder_destroy(self._h)
            self._h = None

    # ---- materials (return a material index) ----
    def MatteMaterial(self, kd, sigma=60.0):          # ui/RenderThread.cpp:79-99
        return _check(lib().gnxr_builder_matte(self._h, _f3(kd), float(sigma)))

    def MirrorMaterial(self, kr):                     # ui/RenderThread.cpp:102
        return _check(lib().gnxr_builder_mirror(self._h, _f3(kr)))

    def getPurplePlasticMaterial(self):               # ui/MaterialList.cpp:48-56
        return _check(lib().gnxr_builder_purple_plastic(self._h))

    def getYelloMetalMaterial(self):                  # ui/MaterialList.cpp:58-69
        return _check(lib().gnxr_builder_yellow_metal(self._h))

    def getWhiteGlassMaterial(self):                  # ui/MaterialList.cpp:71-83
        return _check(lib().gnxr_builder_white_glass(self._h))

    def add_material(self, **kw):
        m = Material()
        m.has_bump = 1
        for k, v in kw.items():
            if isinstance(v, (tuple, list, np.ndarray)):
                setattr(m, k, (C.c_float * len(v))(*[float(x) for x in v]))
            else:
                setattr(m, k, v)
        return _check(lib().gnxr_builder_add_material(self._h, C.byref(m)))

    def DisneyMaterial(self, color, metallic=0.0, eta=1.5, roughness=0.5, specularTint=0.0, anisotropic=0.0, sheen=0.0,
                       sheenTint=0.5, clearcoat=0.0, clearcoatGloss=1.0, specTrans=0.0, thin=False, flatness=0.0,
                       diffTrans=1.0):                 # materials/DisneyMaterial.h:21-36
        return self.add_material(type=_abi.MAT_DISNEY, kd=color, eta=(eta, 0, 0), disney_metallic=metallic,
                                 disney_roughness=roughness, disney_spec_tint=specularTint,
                                 disney_anisotropic=anisotropic, disney_sheen=sheen, disney_sheen_tint=sheenTint,
                                 disney_clearcoat=clearcoat, disney_clearcoat_gloss=clearcoatGloss,
                                 disney_spec_trans=specTrans, disney_thin=int(thin), disney_flatness=flatness,
                                 disney_diff_trans=diffTrans)

    # ---- geometry / lights ----
    def AddModel(self, path, material):               # ui/ModelList.cpp:47-69
        return _check(lib().gnxr_builder_add_model_3d(self._h, os.fsencode(path), int(material)))

    def AddCornell(self, material1, material2, material3):   # ui/ModelList.cpp:71-118
        return _check(lib().gnxr_builder_add_cornell(self._h, int(material1), int(material2), int(material3)))

    def AddFloor(self, material):                     # ui/ModelList.cpp:20-45
        return _check(lib().gnxr_builder_add_floor(self._h, int(material)))

    def AddAreaLight(self, material):                 # ui/ModelList.cpp:120-147
        return _check(lib().gnxr_builder_add_area_light(self._h, int(material)))

    def AddSkyLight(self):                            # ui/ModelList.cpp:163-170
        return _check(lib().gnxr_builder_add_sky_light(self._h))

    def add_emissive_mesh(self, vertices, indices, material, lemit, n_samples=5, object_to_world=None):
        """AddAreaLight's pattern (ui/ModelList.cpp:137-146) for any mesh: one DiffuseAreaLight per triangle."""
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
        m = None
        if object_to_world is not None:
            m = np.ascontiguousarray(object_to_world, dtype=np.float32).reshape(16).ctypes.data_as(C.POINTER(C.c_float))
        return _check(lib().gnxr_builder_add_emissive_mesh(self._h, v.ctypes.data_as(C.POINTER(C.c_float)), len(v), i.ctypes.data_as(C.POINTER(C.c_int32)),
                                                           len(i), m, int(material), _f3(lemit), int(n_samples)))

    def AddSpotLight(self):                           # ui/ModelList.cpp:149-154 (the call is commented out in RenderThread.cpp:138)
        return _check(lib().gnxr_builder_add_spot_light(self._h))

    def AddDistLight(self):                           # ui/ModelList.cpp:156-161
        return _check(lib().gnxr_builder_add_dist_light(self._h))

    def add_delta_light(self, kind, I, light_to_world=None, total_width=45.0, falloff_start=30.0, w_light=(0, 0, 1)):
        """PointLight(LightToWorld, I) / SpotLight(LightToWorld, I, totalWidth, falloffStart) / DistantLight(LightToWorld, L, wLight)."""
        l = Light()
        l.type = {"point": _abi.LIGHT_POINT, "spot": _abi.LIGHT_SPOT, "distant": _abi.LIGHT_DISTANT}[kind]
        l.tri = -1
        l.n_samples = 1
        l.le[:] = [float(v) for v in I]
        l.radius = float(total_width)
        l.falloff_start = float(falloff_start)
        l.center[:] = [float(v) for v in w_light]
        m = np.eye(4, dtype=np.float32) if light_to_world is None else np.asarray(light_to_world, dtype=np.float32).reshape(4, 4)
        l.light_to_world[:] = [float(v) for v in m.reshape(16)]
        return _check(lib().gnxr_builder_add_light(self._h, C.byref(l)))

    def AddInfLight(self, hdr_path):                  # ui/ModelList.cpp:172-179
        self.hdr_path = str(hdr_path)
        return _check(lib().gnxr_builder_add_inf_light(self._h, os.fsencode(hdr_path)))

    def AddInfLightData(self, rgb, light_to_world=None, power=(1.0, 1.0, 1.0)):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        h, w = rgb.shape[:2]
        m = None
        if light_to_world is not None:
            m = np.ascontiguousarray(light_to_world, dtype=np.float32).reshape(16).ctypes.data_as(C.POINTER(C.c_float))
        return _check(lib().gnxr_builder_add_inf_light_data(self._h, rgb.ctypes.data_as(C.POINTER(C.c_float)), w, h, m,
                                                            _f3(power)))

    def add_mesh(self, vertices, indices, material, object_to_world=None, medium_inside=-1, medium_outside=-1, uv=None, normals=None, tangents=None):
        """TriangleMesh(ObjectToWorld, nTriangles, vertexIndices, nVertices, P, S = nullptr, N = normals, UV = uv, ...): uv is the
        per-vertex (u, v) array of the mesh or None (Triangle::GetUVs defaults, what every mesh of the reference gets); normals the
        per-vertex object-space shading normals or None (flat shading)."""
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
        m = None
        if object_to_world is not None:
            m = np.ascontiguousarray(object_to_world, dtype=np.float32).reshape(16).ctypes.data_as(C.POINTER(C.c_float))
        first = _check(lib().gnxr_builder_add_mesh(self._h, v.ctypes.data_as(C.POINTER(C.c_float)), len(v),
                                                   i.ctypes.data_as(C.POINTER(C.c_int32)), len(i), m, int(material),
                                                   int(medium_inside), int(medium_outside)))
        if uv is not None:
            corner = np.ascontiguousarray(np.asarray(uv, dtype=np.float32).reshape(-1, 2)[i].reshape(-1, 6))
            _check(lib().gnxr_builder_set_triangle_uv(self._h, first, len(i), corner.ctypes.data_as(C.POINTER(C.c_float))))
        if normals is not None:
            n = np.asarray(normals, dtype=np.float32).reshape(-1, 3)
            if object_to_world is not None:   # Transform::operator()(Normal3f): n' = (M^-1)^T n, evaluated in float like Transform.h:308-315
                minv = np.linalg.inv(np.asarray(object_to_world, dtype=np.float64).reshape(4, 4)).astype(np.float32)
                x, y, z = n[:, 0].copy(), n[:, 1].copy(), n[:, 2].copy()
                n = np.stack([minv[0, 0] * x + minv[1, 0] * y + minv[2, 0] * z, minv[0, 1] * x + minv[1, 1] * y + minv[2, 1] * z,
                              minv[0, 2] * x + minv[1, 2] * y + minv[2, 2] * z], 1).astype(np.float32)
            corner = np.ascontiguousarray(n[i].reshape(-1, 9))
            _check(lib().gnxr_builder_set_triangle_normals(self._h, first, len(i), corner.ctypes.data_as(C.POINTER(C.c_float))))
        if tangents is not None:   # TriangleMesh::s: s[i] = ObjectToWorld(S[i]), a plain vector transform (Triangle.cpp:49-52)
            t = np.asarray(tangents, dtype=np.float32).reshape(-1, 3)
            if object_to_world is not None:
                mm = np.asarray(object_to_world, dtype=np.float32).reshape(4, 4)
                x, y, z = t[:, 0].copy(), t[:, 1].copy(), t[:, 2].copy()
                t = np.stack([mm[0, 0] * x + mm[0, 1] * y + mm[0, 2] * z, mm[1, 0] * x + mm[1, 1] * y + mm[1, 2] * z,
                              mm[2, 0] * x + mm[2, 1] * y + mm[2, 2] * z], 1).astype(np.float32)
            corner = np.ascontiguousarray(t[i].reshape(-1, 9))
            _check(lib().gnxr_builder_set_triangle_tangents(self._h, first, len(i), corner.ctypes.data_as(C.POINTER(C.c_float))))
        return first

    def add_medium(self, medium, density=None):
        d = None
        if density is not None:
            density = np.ascontiguousarray(density, dtype=np.float32)
            d = density.ctypes.data_as(C.POINTER(C.c_float))
        return _check(lib().gnxr_builder_add_medium(self._h, C.byref(medium), d))

    def add_volume_file(self, path, g=0.0, sigma_scale=1.0, medium_to_world=None):
        """GridDensityMedium from a `.volume` file (Resources/density_render.70.volume's format); returns the medium index."""
        m = None
        if medium_to_world is not None:
            m = np.ascontiguousarray(medium_to_world, dtype=np.float32).reshape(16).ctypes.data_as(C.POINTER(C.c_float))
        return _check(lib().gnxr_builder_add_volume_file(self._h, os.fsencode(path), float(g), float(sigma_scale), m))

    def add_image_texture(self, image, su=1.0, sv=1.0, du=0.0, dv=0.0, trilinear=False, max_aniso=8.0, wrap="repeat", scale=1.0, gamma=False):
        """ImageTexture<RGBSpectrum, Spectrum>(UVMapping2D(su, sv, du, dv), file, doTrilinear, maxAniso, wrap, scale, gamma)
        (textures/ImageTexture.h; the defaults are those of getSmileFacePlasticMaterial, ui/MaterialList.cpp:31-46).  `image` is
        the path of a Radiance .hdr or an array [H, W, 3] of decoded texels (row 0 = top row, as stbi_loadf returns them)."""
        t = Texture(0, 0, 0, su, sv, du, dv, max_aniso, scale, int(bool(trilinear)), {"repeat": 0, "black": 1, "clamp": 2}[wrap], int(bool(gamma)), 0)
        if isinstance(image, (str, bytes, os.PathLike)):
            self.texture_paths = getattr(self, "texture_paths", []) + [str(image)]
            return _check(lib().gnxr_builder_add_texture_file(self._h, C.byref(t), os.fsencode(image)))
        rgb = np.ascontiguousarray(image, dtype=np.float32)
        self.texture_paths = getattr(self, "texture_paths", []) + [""]
        return _check(lib().gnxr_builder_add_texture_data(self._h, C.byref(t), rgb.ctypes.data_as(C.POINTER(C.c_float)), rgb.shape[1], rgb.shape[0]))

    def set_material_texture(self, material, slot, texture):
        """slot "kd" / "ks": replace the material's constant Kd / Ks texture by image texture `texture`."""
        return _check(lib().gnxr_builder_set_material_texture(self._h, int(material), {"kd": 0, "ks": 1}[slot], int(texture)))

    def getSmileFacePlasticMaterial(self, image):     # ui/MaterialList.cpp:31-46 (Kd = Ks = the same ImageTexture, roughness 0.1, remap)
        t = self.add_image_texture(image)
        m = self.add_material(type=_abi.MAT_PLASTIC, kd=(0.5, 0.5, 0.5), ks=(0.5, 0.5, 0.5), urough=0.1, remap_roughness=1)
        self.set_material_texture(m, "kd", t)
        self.set_material_texture(m, "ks", t)
        return m

    def AddSphere(self, center, radius, material, medium_inside=-1, medium_outside=-1):
        """pbrt-v3 quadratic sphere (the reference's shape/Sphere.h is an unfinished stub; see include/gnxr.h)."""
        c = (C.c_float * 3)(*[float(v) for v in center])
        return _check(lib().gnxr_builder_add_sphere(self._h, c, float(radius), int(material), int(medium_inside), int(medium_outside)))

    def set_camera(self, eye=(0, 0, 5), look=(0, 0, 0), up=(0, 1, 0), fov=90.0, lens_radius=0.0, focal_distance=3.0, orthographic=False):
        """CreatePerspectiveCamera (camera/Perspective.cpp:114-135) or, orthographic=True, CreateOrthographicCamera
        (camera/Orthographic.cpp:94-121) on LookAt(eye, look, up)."""
        cam = Camera(_f3(eye), _f3(look), _f3(up), fov, lens_radius, focal_distance, int(bool(orthographic)))
        _check(lib().gnxr_builder_set_camera(self._h, C.byref(cam)))

    def set_bvh_split_method(self, method):
        """BVHAccel(prims, 1, SplitMethod::SAH | HLBVH | Middle | EqualCounts) (accelerator/BVHAccel.h:24-29); the reference uses SAH."""
        _check(lib().gnxr_builder_set_bvh_split_method(self._h, {"sah": 0, "hlbvh": 1, "middle": 2, "equal_counts": 3}[method]))

    def set_camera_medium(self, medium):
        _check(lib().gnxr_builder_set_camera_medium(self._h, int(medium)))

    def desc(self):
        """gnxr_scene_desc pointing into builder-owned memory (valid until the next builder call)."""
        d = SceneDesc()
        _check(lib().gnxr_builder_desc(self._h, C.byref(d)))
        return d


def write_synthetic_3d(path, target_triangles=100000, seed=1):
    """Deterministic stand-in for the absent Resources/dragon.3d (`.MISSING_LARGE_BLOBS`)."""
    _check(lib().gnxr_write_synthetic_3d(os.fsencode(path), int(target_triangles), int(seed)))


class Scene:
    """Device-resident scene: replaces `Scene(make_shared<BVHAccel>(prims, 1), lights)` (RenderThread.cpp:155)."""

    def __init__(self, builder_or_desc):
        self._h = C.c_void_p()
        self._keep = builder_or_desc
        d = builder_or_desc.desc() if isinstance(builder_or_desc, SceneBuilder) else builder_or_desc
        self.n_triangles = int(d.n_triangles)
        _check(lib().gnxr_scene_create(C.byref(d), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.gnxr_scene_destroy(self._h)
            self._h = None

    __del__ = close

    def info(self):
        n, dmax, nv = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().gnxr_scene_info(self._h, C.byref(n), C.byref(dmax), C.byref(nv)))
        return {"bvh_nodes": n.value, "bvh_max_depth": dmax.value, "light_voxels": nv.value}

    # Aggregate seam: Scene::Intersect / IntersectP, batched
    def bvh(self):
        """Test hook: (bounds [n, 6], meta [n, 3] = offset / nPrimitives / axis, primitive order) of the flattened binary BVH."""
        n = C.c_int64(0)
        _check(lib().gnxr_scene_bvh(self._h, None, None, None, 0, C.byref(n)))
        bounds = np.zeros((n.value, 6), np.float32); meta = np.zeros((n.value, 3), np.int32); order = np.zeros(self.n_triangles, np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        _check(lib().gnxr_scene_bvh(self._h, bounds.ctypes.data_as(C.POINTER(C.c_float)), ip(meta), ip(order), n.value, C.byref(n)))
        return bounds, meta, order

    def light_grid_table(self, strategy="spatial", on_host=False):
        """Test hook: the light-selection table (device-built or host-built)."""
        code = {"spatial": _abi.LIGHTS_SPATIAL, "uniform": _abi.LIGHTS_UNIFORM, "power": _abi.LIGHTS_POWER}[strategy]
        n = C.c_int64(0)
        _check(lib().gnxr_light_grid_table(self._h, code, int(on_host), None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.float32)
        _check(lib().gnxr_light_grid_table(self._h, code, int(on_host), out.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n)))
        return out

    def Intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), dtype=HIT_DTYPE)
        _check(lib().gnxr_trace_closest(self._h, rays.ctypes.data_as(C.POINTER(Ray)), len(rays),
                                        hits.ctypes.data_as(C.POINTER(Hit))))
        return hits

    def IntersectP(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        occ = np.zeros(len(rays), dtype=np.uint8)
        _check(lib().gnxr_trace_any(self._h, rays.ctypes.data_as(C.POINTER(Ray)), len(rays),
                                    occ.ctypes.data_as(C.POINTER(C.c_uint8))))
        return occ


HIT_DTYPE = np.dtype([("prim", np.int32), ("t", np.float32), ("b0", np.float32), ("b1", np.float32), ("b2", np.float32),
                      ("n", np.float32, 3)])


def make_rays(o, d, tmax=np.inf):
    o = np.asarray(o, dtype=np.float32).reshape(-1, 3)
    d = np.asarray(d, dtype=np.float32).reshape(-1, 3)
    r = np.zeros((len(o), 8), dtype=np.float32)
    r[:, 0:3] = o
    r[:, 3] = tmax
    r[:, 4:7] = d
    return r


def stats_dict(s):
    return {k: getattr(s, k) for k, _ in Stats._fields_}


class PathIntegrator:
    """Mirror of pbr::PathIntegrator(maxDepth, camera, sampler, bounds, fb, rrThreshold, strategy)
    (integrators/PathIntegrator.cpp:20-29); camera and sampler are implied by the scene and (W, H, spp)."""
    integrator = _abi.INTEGRATOR_PATH

    def __init__(self, maxDepth=5, rrThreshold=1.0, lightSampleStrategy="spatial"):
        self.maxDepth = int(maxDepth)
        self.rrThreshold = float(rrThreshold)
        self.strategy = {"spatial": _abi.LIGHTS_SPATIAL, "uniform": _abi.LIGHTS_UNIFORM,
                         "power": _abi.LIGHTS_POWER}.get(lightSampleStrategy, _abi.LIGHTS_SPATIAL)

    def params(self, width, height, spp, spp_begin=0, spp_end=0, shard_index=0, shard_count=1, shard_rows=1,
               samples_per_pass=0, passes_in_flight=0):
        return RenderParams(width, height, spp, spp_begin, spp_end, self.maxDepth, self.rrThreshold, self.integrator,
                            self.strategy, shard_index, shard_count, shard_rows, samples_per_pass, getattr(self, "directStrategy", 0),
                            passes_in_flight)

    def Render(self, scene, width, height, spp, **kw):
        """Integrator::Render: returns (float32 image [H, W, 4], stats dict)."""
        p = self.params(width, height, spp, **kw)
        img = np.zeros((height, width, 4), dtype=np.float32)
        st = Stats()
        _check(lib().gnxr_render(scene._h, C.byref(p), img.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st)))
        return img, stats_dict(st)

    def Reserve(self, scene, width, height, spp, **kw):
        """gnxr_render_reserve: allocate the path state of a Render / RenderDevice call with these arguments, without rendering."""
        p = self.params(width, height, spp, **kw)
        _check(lib().gnxr_render_reserve(scene._h, C.byref(p)))

    def RenderDevice(self, scene, d_ptr, width, height, spp, stream=None, **kw):
        p = self.params(width, height, spp, **kw)
        st = Stats()
        _check(lib().gnxr_render_device(scene._h, C.byref(p), C.c_void_p(int(d_ptr)),
                                        C.c_void_p(int(stream) if stream else None), C.byref(st)))
        return stats_dict(st)


class VolPathIntegrator(PathIntegrator):
    """pbr::VolPathIntegrator (integrators/VolPathIntegrator.cpp): same constructor arguments as PathIntegrator."""
    integrator = _abi.INTEGRATOR_VOLPATH


class WhittedIntegrator(PathIntegrator):
    """pbr::WhittedIntegrator(maxDepth, ...) (integrators/WhittedIntegrator.h): BASELINE config 1, which the reference runs on
    the CPU only.  On the device it is a per-path depth-first state machine (csrc/whitted_kernel.hip.h)."""
    integrator = _abi.INTEGRATOR_WHITTED

    def __init__(self, maxDepth=5):
        super().__init__(maxDepth, 1.0, "uniform")


class DirectLightingIntegrator(PathIntegrator):
    """pbr::DirectLightingIntegrator(strategy, maxDepth, ...) (integrators/DirectLightingIntegrator.h:16-38); strategy is the
    reference's LightStrategy: "all" = UniformSampleAll (Light::nSamples array samples per light and vertex), "one" =
    UniformSampleOne.  Shares the depth-first device state machine with Whitted (csrc/whitted_kernel.hip.h)."""
    integrator = _abi.INTEGRATOR_DIRECT

    def __init__(self, strategy="all", maxDepth=5):
        super().__init__(maxDepth, 1.0, "uniform")
        self.directStrategy = {"all": _abi.DIRECT_SAMPLE_ALL, "one": _abi.DIRECT_SAMPLE_ONE}[strategy]


def sample_halton(width, height, px, py, s, dim):
    px = np.ascontiguousarray(px, dtype=np.int32)
    py = np.ascontiguousarray(py, dtype=np.int32)
    s = np.ascontiguousarray(s, dtype=np.int64)
    dim = np.ascontiguousarray(dim, dtype=np.int32)
    out = np.zeros(len(px), dtype=np.float32)
    _check(lib().gnxr_sample_halton(width, height, px.ctypes.data_as(C.POINTER(C.c_int32)),
                                    py.ctypes.data_as(C.POINTER(C.c_int32)), s.ctypes.data_as(C.POINTER(C.c_int64)),
                                    dim.ctypes.data_as(C.POINTER(C.c_int32)), len(px),
                                    out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def camera_rays(camera, width, height, px, py, s):
    px = np.ascontiguousarray(px, dtype=np.int32)
    py = np.ascontiguousarray(py, dtype=np.int32)
    s = np.ascontiguousarray(s, dtype=np.int64)
    o = np.zeros((len(px), 3), dtype=np.float32)
    d = np.zeros((len(px), 3), dtype=np.float32)
    _check(lib().gnxr_camera_rays(C.byref(camera), width, height, px.ctypes.data_as(C.POINTER(C.c_int32)),
                                  py.ctypes.data_as(C.POINTER(C.c_int32)), s.ctypes.data_as(C.POINTER(C.c_int64)),
                                  len(px), o.ctypes.data_as(C.POINTER(C.c_float)),
                                  d.ctypes.data_as(C.POINTER(C.c_float))))
    return o, d


def save_png(path, rgba8):
    """FrameBuffer::saveToFile (ui/FrameBuffer.cpp:6-9): rgba8 is a [H, W, 4] uint8 array."""
    rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w = rgba8.shape[:2]
    _check(lib().gnxr_framebuffer_save_png(str(path).encode(), rgba8.ctypes.data_as(C.POINTER(C.c_uint8)), w, h))


def eval_libm(fn, x, x2=None):
    """Test hook: the device's float libm (fn = "log" | "exp" | "sin" | "cos" | "acos" | "atan2") on the array x (atan2: y = x, x = x2)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros_like(x)
    p2 = None
    if x2 is not None:
        x2 = np.ascontiguousarray(x2, dtype=np.float32)
        p2 = x2.ctypes.data_as(C.POINTER(C.c_float))
    code = {"log": 0, "exp": 1, "sin": 2, "cos": 3, "sincos.sin": 4, "sincos.cos": 5, "acos": 6, "atan2": 7, "pow": 8}[fn]
    _check(lib().gnxr_eval_libm(code, x.ctypes.data_as(C.POINTER(C.c_float)), p2, x.size, out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def eval_libm_f64(fn, x):
    """Test hook: the device's double-precision sin / cos / sqrt / tan on float arguments (widened), results as float64."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(x.shape, dtype=np.float64)
    _check(lib().gnxr_eval_libm_f64({"sin": 0, "cos": 1, "sqrt": 2, "tan": 3}[fn], x.ctypes.data_as(C.POINTER(C.c_float)), x.size,
                                    out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def probe_valu_peak():
    """Measurement hook: VALU issue rate of the device in 1e9 wave64 instructions / s (independent v_fma_f32 chains, 8 waves per SIMD)."""
    v = C.c_double(0.0)
    _check(lib().gnxr_probe_valu_peak(C.byref(v)))
    return v.value


def probe_gather_peak():
    """Measurement hook: per-lane 16-byte gather rate of the device in 1e9 lane-loads / s (8 dwordx4 of a random 128-byte record per lane)."""
    v = C.c_double(0.0)
    _check(lib().gnxr_probe_gather_peak(C.byref(v)))
    return v.value


def framebuffer_update(running_mean, frame, frame_count):
    """FrameBuffer::update_f_u_c (ui/FrameBuffer.h:127-149): running mean + 1-exp(-4x) tone map to RGBA8."""
    h, w = frame.shape[:2]
    rgba8 = np.zeros((h, w, 4), dtype=np.uint8)
    _check(lib().gnxr_framebuffer_update(running_mean.ctypes.data_as(C.POINTER(C.c_float)),
                                         np.ascontiguousarray(frame, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float)),
                                         w, h, int(frame_count), rgba8.ctypes.data_as(C.POINTER(C.c_uint8))))
    return rgba8

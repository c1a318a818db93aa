"""ctypes access to the CPU oracle (oracle/libgnx_oracle.so) and to oracle/_ref/gnx_ref.

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np

import gnxraytracer_amd as gx
from gnxraytracer_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libgnx_oracle.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "gnx_ref")

_olib = None


def olib():
    global _olib
    if _olib is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libgnx_oracle.so"])
        L = C.CDLL(ORACLE_SO)
        P, VP = C.POINTER, C.c_void_p
        f32, i32, i64, u8 = C.c_float, C.c_int32, C.c_int64, C.c_uint8
        L.gnxo_scene_create.argtypes = [P(_abi.SceneDesc), P(VP)]
        L.gnxo_scene_destroy.argtypes = [VP]
        L.gnxo_scene_info.argtypes = [VP, P(i32), P(i32)]
        L.gnxo_scene_bvh.argtypes = [VP, P(f32), P(i32), P(i32), P(i32), P(i32)]
        L.gnxo_render.argtypes = [VP, P(_abi.RenderParams), P(f32), P(_abi.Stats), C.c_int]
        L.gnxo_set_count_traversal.argtypes = [VP, C.c_int]
        L.gnxo_trace_closest.argtypes = [VP, P(_abi.Ray), i64, P(_abi.Hit)]
        L.gnxo_trace_any.argtypes = [VP, P(_abi.Ray), i64, P(u8)]
        L.gnxo_sample_halton.argtypes = [i32, i32, P(i32), P(i32), P(i64), P(i32), i64, P(f32)]
        L.gnxo_camera_rays.argtypes = [P(_abi.Camera), i32, i32, P(i32), P(i32), P(i64), i64, P(f32), P(f32)]
        L.gnxo_rng_u32.argtypes = [C.c_int, C.c_uint64, C.c_int, P(C.c_uint32)]
        L.gnxo_perm_table.argtypes = [P(C.c_uint16), i64]
        L.gnxo_perm_table.restype = i64
        L.gnxo_primes.argtypes = [P(i32), P(i32)]
        L.gnxo_bsdf_probe.argtypes = [VP, P(_abi.Ray), P(f32), P(f32), i64, C.c_int, P(f32)]
        L.gnxo_light_probe.argtypes = [VP, C.c_int, C.c_int, P(f32), P(f32), P(f32), P(f32), i64, P(f32)]
        L.gnxo_light_le.argtypes = [VP, C.c_int, P(_abi.Ray), i64, P(f32)]
        L.gnxo_framebuffer_update.argtypes = [P(f32), P(f32), i32, i32, i32, P(u8)]
        L.gnxo_libm_f64.argtypes = [i32, P(f32), i64, P(C.c_double)]
        _olib = L
    return _olib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleScene:
    def __init__(self, builder):
        self._keep = builder
        self._h = C.c_void_p()
        d = builder.desc()
        rc = olib().gnxo_scene_create(C.byref(d), C.byref(self._h))
        assert rc == 0

    def __del__(self):
        if getattr(self, "_h", None):
            olib().gnxo_scene_destroy(self._h)
            self._h = None

    def info(self):
        n, d = C.c_int32(), C.c_int32()
        olib().gnxo_scene_info(self._h, C.byref(n), C.byref(d))
        return {"bvh_nodes": n.value, "bvh_max_depth": d.value}

    def bvh(self, n_tris):
        nn = self.info()["bvh_nodes"]
        b = np.zeros((nn, 6), np.float32)
        off, npr, ax = (np.zeros(nn, np.int32) for _ in range(3))
        order = np.zeros(n_tris, np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        olib().gnxo_scene_bvh(self._h, _fp(b), ip(off), ip(npr), ip(ax), ip(order))
        return b, off, npr, ax, order

    def set_bvh(self, bounds, meta, order):
        """Traverse a BVH handed in (e.g. the one the compiled reference built with SplitMethod::HLBVH) instead of the oracle's own."""
        bounds = np.ascontiguousarray(bounds, np.float32); order = np.ascontiguousarray(order, np.int32)
        cols = [np.ascontiguousarray(meta[:, k], np.int32) for k in range(3)]
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        olib().gnxo_scene_set_bvh(self._h, _fp(bounds), ip(cols[0]), ip(cols[1]), ip(cols[2]), C.c_int64(len(bounds)), ip(order), C.c_int64(len(order)))

    def render(self, integrator, width, height, spp, threads=0, count_traversal=False, **kw):
        p = integrator.params(width, height, spp, **kw)
        img = np.zeros((height, width, 4), np.float32)
        st = _abi.Stats()
        olib().gnxo_set_count_traversal(self._h, int(count_traversal))
        rc = olib().gnxo_render(self._h, C.byref(p), _fp(img), C.byref(st), threads)
        assert rc == 0
        return img, gx.stats_dict(st)

    def Intersect(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), gx.HIT_DTYPE)
        olib().gnxo_trace_closest(self._h, rays.ctypes.data_as(C.POINTER(_abi.Ray)), len(rays), hits.ctypes.data_as(C.POINTER(_abi.Hit)))
        return hits

    def IntersectP(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        occ = np.zeros(len(rays), np.uint8)
        olib().gnxo_trace_any(self._h, rays.ctypes.data_as(C.POINTER(_abi.Ray)), len(rays), occ.ctypes.data_as(C.POINTER(C.c_uint8)))
        return occ

    def bsdf_probe(self, rays, wi, u, flags=31):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        wi = np.ascontiguousarray(wi, np.float32)
        u = np.ascontiguousarray(u, np.float32)
        out = np.zeros((len(rays), 16), np.float32)
        olib().gnxo_bsdf_probe(self._h, rays.ctypes.data_as(C.POINTER(_abi.Ray)), _fp(wi), _fp(u), len(rays), flags, _fp(out))
        return out

    def light_probe(self, light, refP, refN, u, wiQ, strategy=_abi.LIGHTS_UNIFORM):
        refP, refN, u, wiQ = (np.ascontiguousarray(a, np.float32) for a in (refP, refN, u, wiQ))
        out = np.zeros((len(refP), 12), np.float32)
        olib().gnxo_light_probe(self._h, light, strategy, _fp(refP), _fp(refN), _fp(u), _fp(wiQ), len(refP), _fp(out))
        return out

    def light_le(self, light, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros((len(rays), 3), np.float32)
        olib().gnxo_light_le(self._h, light, rays.ctypes.data_as(C.POINTER(_abi.Ray)), len(rays), _fp(out))
        return out


def oracle_framebuffer_update(running_mean, frame, frame_count):
    """FrameBuffer::update_f_u_c restated (oracle/gnx_oracle.cpp): updates running_mean in place, returns the RGBA8 plane."""
    h, w = frame.shape[:2]
    frame = np.ascontiguousarray(frame, np.float32)
    rgba8 = np.zeros((h, w, 4), np.uint8)
    rc = olib().gnxo_framebuffer_update(_fp(running_mean), _fp(frame), w, h, int(frame_count), rgba8.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0
    return rgba8


def host_libm_f64(fn, x):
    """sin / cos / sqrt / tan of the host's libm (double) on float32 arguments widened to double, all cores."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(x.shape, np.float64)
    rc = olib().gnxo_libm_f64({"sin": 0, "cos": 1, "sqrt": 2, "tan": 3}[fn], _fp(x), x.size, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out


def oracle_halton(width, height, px, py, s, dim):
    px, py = np.ascontiguousarray(px, np.int32), np.ascontiguousarray(py, np.int32)
    s, dim = np.ascontiguousarray(s, np.int64), np.ascontiguousarray(dim, np.int32)
    out = np.zeros(len(px), np.float32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    olib().gnxo_sample_halton(width, height, ip(px), ip(py), s.ctypes.data_as(C.POINTER(C.c_int64)), ip(dim), len(px), _fp(out))
    return out


def oracle_camera_rays(camera, width, height, px, py, s):
    px, py, s = np.ascontiguousarray(px, np.int32), np.ascontiguousarray(py, np.int32), np.ascontiguousarray(s, np.int64)
    o, d = np.zeros((len(px), 3), np.float32), np.zeros((len(px), 3), np.float32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    olib().gnxo_camera_rays(C.byref(camera), width, height, ip(px), ip(py), s.ctypes.data_as(C.POINTER(C.c_int64)), len(px), _fp(o), _fp(d))
    return o, d


def oracle_rng(n=64, seq=None):
    out = np.zeros(n, np.uint32)
    olib().gnxo_rng_u32(0 if seq is None else 1, 0 if seq is None else seq, n, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def oracle_perms():
    n = olib().gnxo_perm_table(None, 0)
    out = np.zeros(n, np.uint16)
    olib().gnxo_perm_table(out.ctypes.data_as(C.POINTER(C.c_uint16)), n)
    return out


def oracle_primes():
    p, s = np.zeros(1000, np.int32), np.zeros(1000, np.int32)
    olib().gnxo_primes(p.ctypes.data_as(C.POINTER(C.c_int32)), s.ctypes.data_as(C.POINTER(C.c_int32)))
    return p, s


# ---------------------------------------------------------------- reference driver (dev container only)
def have_ref():
    return os.path.exists(REF_BIN) and os.path.isdir("/root/reference")


def write_scene_file(builder, path):
    """Serialise a gnxr_scene_desc for oracle/_ref/gnx_ref (format: oracle/ref_driver.cpp readScene)."""
    d = builder.desc()
    nt, nv = d.n_triangles, d.n_vertices

    def arr(ptr, n, ct):
        if n == 0 or not ptr:
            return b""
        return C.string_at(ptr, n * C.sizeof(ct))

    with open(path, "wb") as f:
        f.write(b"GNXS" + struct.pack("<i", 6))
        f.write(struct.pack("<8i", nv, nt, d.n_materials, d.n_lights, d.n_media, d.env_width, d.env_height, d.camera_medium))
        f.write(bytes(d.camera))
        f.write(arr(d.vertices, 3 * nv, C.c_float))
        f.write(arr(d.indices, 3 * nt, C.c_int32))
        f.write(arr(d.tri_material, nt, C.c_int32))
        f.write(arr(d.tri_light, nt, C.c_int32))
        f.write(arr(d.tri_medium_inside, nt, C.c_int32))
        f.write(arr(d.tri_medium_outside, nt, C.c_int32))
        f.write(arr(d.materials, d.n_materials, _abi.Material))
        f.write(arr(d.lights, d.n_lights, _abi.Light))
        f.write(arr(d.media, d.n_media, _abi.Medium))
        nd = 0
        for i in range(d.n_media):
            m = d.media[i]
            if m.type == _abi.MEDIUM_GRID:
                nd = max(nd, m.density_offset + m.nx * m.ny * m.nz)
        f.write(struct.pack("<q", nd))
        f.write(arr(d.grid_density, nd, C.c_float))
        f.write(arr(d.env_rgb, 3 * d.env_width * d.env_height, C.c_float))
        hp = (builder.hdr_path or "").encode()
        f.write(struct.pack("<i", len(hp)) + hp)
        # image textures: parameters + the FILE the reference's ImageTexture loads (array textures cannot be handed over)
        f.write(struct.pack("<i", d.n_textures))
        f.write(arr(d.textures, d.n_textures, _abi.Texture))
        paths = getattr(builder, "texture_paths", [])
        for i in range(d.n_textures):
            tp = os.path.abspath(paths[i]).encode() if paths[i] else b""
            assert tp, "the reference loads image textures from files: use add_image_texture(path)"
            f.write(struct.pack("<i", len(tp)) + tp)
        # per-corner uvs (version 3)
        f.write(struct.pack("<i", 1 if d.tri_uv else 0))
        f.write(arr(d.tri_uv, 6 * nt, C.c_float))
        # per-corner shading normals (version 4)
        f.write(struct.pack("<i", 1 if d.tri_n else 0))
        f.write(arr(d.tri_n, 9 * nt, C.c_float))
        f.write(struct.pack("<i", 1 if d.tri_s else 0))   # per-corner shading tangents (version 5)
        f.write(arr(d.tri_s, 9 * nt, C.c_float))
        f.write(struct.pack("<i", d.bvh_split_method))     # BVHAccel SplitMethod (version 6)


def run_ref(scene_path, cmd, in_bytes, args=(), stderr=None):
    with tempfile.TemporaryDirectory() as td:
        ip, op = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        if in_bytes is None:
            ip = "-"
        else:
            with open(ip, "wb") as f:
                f.write(in_bytes)
        subprocess.check_call([REF_BIN, scene_path or "-", cmd, ip, op] + [str(a) for a in args], stdout=subprocess.DEVNULL, stderr=stderr)
        with open(op, "rb") as f:
            return f.read()

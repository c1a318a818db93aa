"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI (libgnxr.so),
against the golden vectors from the compiled reference and against the CPU oracle on the same seeded inputs.

Bars: bit-exact for integer / index work, for Halton values, camera rays, hit records, the float libm, and -- since the
device restates glibc's logf / expf / sinf / cosf / acosf / atan2f -- for whole IMAGES and ray counts of every golden
scene.  Comparisons against the oracle at sizes without a golden keep a tolerance (RMSE 1e-4, north-star bar 1e-3, ray
counts 0.1 %) because powf (Disney clearcoat) and the double-precision sin / cos of MicroFacet.cpp still go through OCML."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import scenes
from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-4      # per-pixel, linear radiance
MAXABS_TOL = 5e-3
RAYS_TOL = 1e-3


def biteq(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def rmse(a, b):
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return float(np.sqrt((d ** 2).mean())), float(np.abs(d).max())


def test_native_library_is_the_one_loaded(gpu):
    maps = open("/proc/self/maps").read()
    assert "libgnxr.so" in maps


@pytest.mark.parametrize("fn", ["log", "exp", "sin", "cos"])
def test_float_libm_carries_glibc_bits(gpu, fn):
    """std::log / exp / sin / cos on floats are glibc's logf / expf / sinf / cosf in the reference (not correctly rounded);
    the device restates those algorithms (device_math.h).  Checked against this image's libm.so.6, bit for bit, over the
    argument ranges the path produces (1 - u, -sigma_t * t, angles in [-pi, 2 pi])."""
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    f = getattr(libm, fn + "f")
    f.restype, f.argtypes = C.c_float, [C.c_float]
    rng = np.random.default_rng(11)
    n = 200000
    if fn == "log":
        x = np.concatenate([1 - rng.random(n, dtype=np.float32), np.exp(rng.uniform(-30, 30, n)).astype(np.float32), [1.0, 0.5, 2.0 ** -24, 1e-40]])
    elif fn == "exp":
        x = np.concatenate([-rng.exponential(5.0, n), rng.uniform(-110, 90, n), [0.0, -0.0, -1e-8, -103.9, -104.5, 88.7, 89.0]])
    else:
        x = np.concatenate([rng.uniform(-np.pi, 2 * np.pi, n), rng.uniform(-1e-3, 1e-3, n // 4), rng.uniform(-119, 119, n), [0.0, np.pi / 4, 0.75, 0.78]])
    x = x.astype(np.float32)
    ref = np.array([f(float(v)) for v in x], np.float32)
    assert biteq(gpu.eval_libm(fn, x), ref)
    if fn in ("sin", "cos"):
        assert biteq(gpu.eval_libm("sincos." + fn, x), ref)


def test_acosf_atan2f_carry_glibc_bits(gpu):
    """SphericalTheta / SphericalPhi (core/Geometry.h:1436-1443, InfiniteAreaLight::Le / Pdf_Li) call acosf / atan2f: glibc's
    fdlibm float versions, restated on the device."""
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.acosf.restype, libm.acosf.argtypes = C.c_float, [C.c_float]
    libm.atan2f.restype, libm.atan2f.argtypes = C.c_float, [C.c_float, C.c_float]
    rng = np.random.default_rng(12)
    x = np.concatenate([rng.uniform(-1, 1, 200000), [1, -1, 0, 0.5, -0.5, 1e-9, -1e-9]]).astype(np.float32)
    assert biteq(gpu.eval_libm("acos", x), np.array([libm.acosf(float(v)) for v in x], np.float32))
    y = np.concatenate([rng.normal(size=200000), [0, 0, 1, -1, 1e-30, 1e30, 0.0, -0.0]]).astype(np.float32)
    x = np.concatenate([rng.normal(size=200000), [1, -1, 0, 0, 1e30, 1e-30, -0.0, 1.0]]).astype(np.float32)
    assert biteq(gpu.eval_libm("atan2", y, x), np.array([libm.atan2f(float(a), float(b)) for a, b in zip(y, x)], np.float32))
    # powf(alpha^2, 1 - u): DisneyClearcoat::Sample_f
    libm.powf.restype, libm.powf.argtypes = C.c_float, [C.c_float, C.c_float]
    x = np.concatenate([rng.uniform(1e-6, 1, 150000), np.exp(rng.uniform(-20, 5, 50000)), [1.0, 0.5, 2.0, 1e-30]]).astype(np.float32)
    y = np.concatenate([1 - rng.random(150000), rng.uniform(-8, 8, 50000), [0.3, 1.0, 0.0, 3.0]]).astype(np.float32)
    assert biteq(gpu.eval_libm("pow", x, y), np.array([libm.powf(float(a), float(b)) for a, b in zip(x, y)], np.float32))


def test_double_sin_cos_carry_glibc_bits_for_every_float_angle(gpu):
    """`Float phi = 6.28318530718 * U2; r * cos(phi); r * sin(phi)` (core/MicroFacet.cpp:220-223): the unqualified calls bind to glibc's
    double __sin / __cos (0.55 ULP, not correctly rounded; OCML differs from them in ~3 % of the arguments).  The device restates them
    (device_math.h gx_sin_d / gx_cos_d).  phi is a float in [0, 2 pi], so the domain is finite: EVERY float from 0 to 6.2831855 --
    1 086 918 620 arguments, all four argument ranges of s_sin.c -- is evaluated on the device and compared with this box's libm.so.6,
    all 64 bits, plus negative and large (range-reduction) arguments on a sample.  sqrt (IEEE) rides along on a sample."""
    hi = int(np.float32(6.2831855).view(np.uint32))
    chunk = 1 << 26
    for fn in ("sin", "cos"):
        bad = 0
        for start in range(0, hi + 1, chunk):
            x = np.arange(start, min(start + chunk, hi + 1), dtype=np.uint32).view(np.float32)
            d, h = gpu.eval_libm_f64(fn, x), ol.host_libm_f64(fn, x)
            bad += int((d.view(np.uint64) != h.view(np.uint64)).sum())
        assert bad == 0, (fn, bad)
    rng = np.random.default_rng(21)
    x = np.concatenate([-rng.uniform(0, 7, 500000), rng.uniform(-1e5, 1e5, 500000), rng.uniform(-1.05e8, 1.05e8, 500000),
                        [0.0, -0.0, 0.126, 0.125999, 0.855469, 2.426265, np.pi, np.pi / 2, 1e-9, 7.4e-9, 1.5e-8]]).astype(np.float32)
    for fn in ("sin", "cos"):
        d, h = gpu.eval_libm_f64(fn, x), ol.host_libm_f64(fn, x)
        assert (d.view(np.uint64) == h.view(np.uint64)).all(), fn
    u = rng.random(2000000, dtype=np.float32)
    r = (u / (1 - u)).astype(np.float32)
    assert (gpu.eval_libm_f64("sqrt", r).view(np.uint64) == ol.host_libm_f64("sqrt", r).view(np.uint64)).all()


@pytest.mark.parametrize("res", [(256, 256), (1920, 1080), (64, 64)])
def test_halton_bit_exact(gpu, res):
    g = golden(f"halton_{res[0]}x{res[1]}.npz")
    q = g["q"]
    v = gpu.sample_halton(res[0], res[1], q[:, 0], q[:, 1], q[:, 2], q[:, 3])
    assert (v.view(np.uint32) == g["bits"]).all()
    # a larger seeded sweep against the oracle, including high dimensions and the last samples of 1024 spp
    rng = np.random.default_rng(5)
    n = 200000
    px, py = rng.integers(0, res[0], n), rng.integers(0, res[1], n)
    s, dim = rng.integers(0, 1024, n), rng.integers(0, 999, n)
    assert (gpu.sample_halton(res[0], res[1], px, py, s, dim).view(np.uint32) == ol.oracle_halton(res[0], res[1], px, py, s, dim).view(np.uint32)).all()


@pytest.mark.parametrize("res", [(256, 256), (1920, 1080)])
def test_camera_rays_bit_exact(gpu, res):
    g = golden(f"camrays_{res[0]}x{res[1]}.npz")
    b = scenes.cornell()
    o, d = gpu.camera_rays(b.desc().camera, res[0], res[1], g["q"][:, 0], g["q"][:, 1], g["q"][:, 2])
    assert biteq(np.concatenate([o, d], 1), g["od"])


def _scene(name):
    if name == "cornell":
        return scenes.cornell()
    return scenes.dragon_cornell(2000, "glass+metal", mesh_path=os.path.join(GOLDEN, "mesh_2k.3d"))


@pytest.mark.parametrize("name", ["cornell", "mesh2k"])
def test_aggregate_seam_hits(gpu, name):
    b = _scene(name)
    scene = gpu.Scene(b)
    h = golden(f"hits_{name}.npz")
    gh = scene.Intersect(h["rays"])
    assert (gh["prim"] == h["prim"]).all()
    m = h["prim"] >= 0
    assert biteq(gh["t"][m], h["t"][m]) and biteq(gh["n"][m], h["n"][m])
    assert (scene.IntersectP(h["srays"]) == h["occluded"]).all()
    # barycentrics (not kept by the reference) against the oracle, plus a bigger batch
    osc = ol.OracleScene(b)
    rays = scenes.random_rays(300000, seed=9)
    gh, oh = scene.Intersect(rays), osc.Intersect(rays)
    assert (gh["prim"] == oh["prim"]).all()
    m = oh["prim"] >= 0
    for f in ("t", "b0", "b1", "b2", "n"):
        assert biteq(gh[f][m], oh[f][m]), f
    # edge cases: empty batch, rays that leave the scene, zero-length shadow segments
    assert len(scene.Intersect(np.zeros((0, 8), np.float32))) == 0
    away = gpu.make_rays([[0, 0, 100]] * 8, [[0, 0, 1]] * 8)
    assert (scene.Intersect(away)["prim"] == -1).all() and (scene.IntersectP(away) == 0).all()


def _render_case(gpu, name):
    g = golden("render.npz")
    W, H, spp, depth = (int(v) for v in g[name + "_cfg"])
    if name in ("cornell", "cornell_uniform"):
        b = scenes.cornell()
    elif name == "zoo":
        b = scenes.material_zoo()
    elif name == "mesh2k":
        b = _scene("mesh2k")
    else:
        b = scenes.cornell(sky=True)
        b.AddInfLight(os.path.join(GOLDEN, "env_100x50.hdr"))
    integ = gpu.PathIntegrator(depth, 1.0, {"cornell_uniform": "uniform", "cornell_env_power": "power"}.get(name, "spatial"))
    return b, integ, (W, H, spp), g[name], tuple(int(v) for v in g[name + "_rays"])


@pytest.mark.parametrize("name", ["cornell", "zoo", "mesh2k", "cornell_env", "cornell_uniform", "cornell_env_power"])
def test_render_matches_reference_images(gpu, name):
    b, integ, (W, H, spp), ref_img, ref_rays = _render_case(gpu, name)
    img, st = integ.Render(gpu.Scene(b), W, H, spp)
    assert (img[..., 3] == 1).all()
    # every pixel and both ray counts carry the reference's bits
    assert (st["rays_closest"], st["rays_any"]) == ref_rays
    assert biteq(img[..., :3], ref_img[..., :3])


def test_cfg2_full_size_against_recorded_reference_run(gpu):
    """cfg 2 (Cornell 256x256 @64 spp, maxDepth 8): 16 058 662 / 12 329 468 rays and checksum 78538.576918
    were recorded from the complete reference."""
    g = golden("cfg2_recorded.npz")
    img, st = gpu.PathIntegrator(8, 1.0, "spatial").Render(gpu.Scene(scenes.cornell()), 256, 256, 64)
    assert (st["rays_closest"], st["rays_any"]) == (16058662, 12329468)
    assert abs(float(img[..., :3].astype(np.float64).sum()) - 78538.576918) < 1e-5
    assert biteq(img[::4, ::4, :3], g["thumb"])


def test_full_size_properties_1080p(gpu):
    """BASELINE-size properties that need no oracle: determinism, shard/sample-range recombination and
    linearity in the emitted radiance (doubling Le doubles every pixel exactly: all products scale by 2)."""
    b = scenes.dragon_cornell(100000, "glass+metal")
    scene = gpu.Scene(b)
    integ = gpu.PathIntegrator(8, 1.0, "spatial")
    W, H, spp = 1920, 1080, 1024
    kw = dict(spp_begin=0, spp_end=2)
    a, sa = integ.Render(scene, W, H, spp, **kw)
    a2, sa2 = integ.Render(scene, W, H, spp, **kw)
    assert (a.view(np.uint32) == a2.view(np.uint32)).all() and sa["rays_closest"] == sa2["rays_closest"]      # no atomics on pixels
    assert np.isfinite(a).all() and a[..., :3].min() >= 0
    acc = np.zeros_like(a)
    rays = 0
    for r in range(4):
        part, s = integ.Render(scene, W, H, spp, shard_index=r, shard_count=4, shard_rows=1, **kw)
        acc += part
        rays += s["rays_closest"] + s["rays_any"]
    assert (acc.view(np.uint32) == a.view(np.uint32)).all() and rays == sa["rays_closest"] + sa["rays_any"]
    one, _ = integ.Render(scene, W, H, spp, spp_begin=0, spp_end=1)
    two, _ = integ.Render(scene, W, H, spp, spp_begin=1, spp_end=2)
    assert np.allclose(one[..., :3] + two[..., :3], a[..., :3], rtol=1e-6, atol=1e-9)
    # linearity in Le
    d = b.desc()
    for i in range(d.n_lights):
        for c in range(3):
            d.lights[i].le[c] *= 2.0
    bright, sb = integ.Render(gpu.Scene(d), W, H, spp, **kw)
    assert (bright[..., :3] == 2 * a[..., :3]).all() and sb["rays_any"] == sa["rays_any"]
    del b


def test_headline_config_at_full_resolution_against_oracle(gpu):
    """BASELINE configs[2] at its own size (1920x1080, HaltonSampler(1024), 100 k-triangle mesh, Glass + Metal): two of the
    1024 samples of every pixel (16.7 M rays), images and ray counts against the oracle, bit for bit."""
    b = scenes.dragon_cornell(100000, "glass+metal")
    integ = gpu.PathIntegrator(8, 1.0, "spatial")
    kw = dict(spp_begin=500, spp_end=502)
    img, st = integ.Render(gpu.Scene(b), 1920, 1080, 1024, **kw)
    oimg, ost = ol.OracleScene(b).render(integ, 1920, 1080, 1024, **kw)
    assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
    same = (img[..., :3].view(np.uint32) == oimg[..., :3].view(np.uint32)).mean()
    r, mx = rmse(img, oimg)
    print(f"1920x1080 samples 500-501: {same * 100:.5f} % of the values bit-identical, rmse {r:.2e}")
    assert biteq(img[..., :3], oimg[..., :3])   # no margin: the double sin / cos of MicroFacet.cpp:220-223 are glibc's too (test above)


def test_cfg4_at_full_resolution_against_oracle(gpu):
    """BASELINE configs[3] at its own size: the 100 k-triangle mesh in Glass / Metal / Plastic / Disney quarters inside the Cornell
    box, lit by the area light AND an InfiniteAreaLight over a 1000 x 500 lat-long map (RLE .hdr -> builder's RGBE reader ->
    Lanczos resample to 1024 x 512 -> 2048 x 1024 Distribution2D), 1920 x 1080, HaltonSampler(1024): two of the 1024 samples of every
    pixel, image and ray counts against the oracle bit for bit, and 4 row shards recombine to the same image.  The map is the
    deterministic synthetic stand-in of tests/scenes.py (the reference's MonValley1000.hdr cannot travel to the GPU box; in the
    development container tests/test_reference_assets.py pins reader, light and materials on the real file)."""
    env = scenes.synthetic_env_path(1000, 500)
    b = scenes.dragon_cornell(100000, "zoo", env=env)
    d = b.desc()
    assert (d.env_width, d.env_height) == (1000, 500) and d.n_lights == 3
    integ = gpu.PathIntegrator(8, 1.0, "spatial")
    kw = dict(spp_begin=700, spp_end=702)
    scene = gpu.Scene(b)
    img, st = integ.Render(scene, 1920, 1080, 1024, **kw)
    oimg, ost = ol.OracleScene(b).render(integ, 1920, 1080, 1024, **kw)
    assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
    same = (img[..., :3].view(np.uint32) == oimg[..., :3].view(np.uint32)).mean()
    r, mx = rmse(img, oimg)
    print(f"cfg4 1920x1080 samples 700-701: {same * 100:.5f} % of the values bit-identical, rmse {r:.2e}, rays {st['rays_closest']}+{st['rays_any']}")
    assert biteq(img[..., :3], oimg[..., :3])
    acc = np.zeros_like(img)
    rays = 0
    for rk in range(4):
        part, s = integ.Render(scene, 1920, 1080, 1024, shard_index=rk, shard_count=4, shard_rows=1, **kw)
        acc += part
        rays += s["rays_closest"] + s["rays_any"]
    assert (acc.view(np.uint32) == img.view(np.uint32)).all() and rays == st["rays_closest"] + st["rays_any"]
    assert np.isfinite(img).all() and img[..., :3].min() >= 0 and img[..., :3].max() * (1024 / 2) > 1.0   # (two of 1024 samples, divided by 1024)


def test_dragon_scene_against_oracle(gpu):
    """The headline scene at a size the oracle finishes in seconds."""
    b = scenes.dragon_cornell(100000, "glass+metal")
    integ = gpu.PathIntegrator(8, 1.0, "spatial")
    img, st = integ.Render(gpu.Scene(b), 240, 135, 1024, spp_begin=0, spp_end=8)
    oimg, ost = ol.OracleScene(b).render(integ, 240, 135, 1024, spp_begin=0, spp_end=8)
    # Glass + Metal sample microfacet normals through the double-precision sin / cos of MicroFacet.cpp:220-223: glibc's on both sides now
    assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
    assert biteq(img[..., :3], oimg[..., :3])


@pytest.mark.parametrize("name", ["vol_synth", "vol_cfg5"])
def test_volpath_matches_reference_images(gpu, name):
    """cfg 5 (VolPathIntegrator + GridDensityMedium + HomogeneousMedium): images and ray counts produced by the
    restated VolPath loop on the reference's own classes (tests/golden/render_vol.npz)."""
    g = golden("render_vol.npz")
    W, H, spp, depth = (int(v) for v in g[name + "_cfg"])
    b = scenes.volume_cornell(sigma_a=(0.5,) * 3, sigma_s=(3.5,) * 3, g_grid=0.3) if name == "vol_synth" else scenes.volume_cornell_cfg5(0.05)
    img, st = gpu.VolPathIntegrator(depth, 1.0, "spatial").Render(gpu.Scene(b), W, H, spp)
    assert (st["rays_closest"], st["rays_any"]) == (int(g[name + "_rays"][0]), 0)
    assert biteq(img[..., :3], g[name][..., :3])


def test_volpath_cfg5_density_against_oracle(gpu):
    """cfg 5 at the volume file's own sigma_a = 10, sigma_s = 90: the delta-tracking loops run past Halton
    dimension 1000, where the reference indexes PrimeSums out of bounds (undefined); this build and the oracle
    wrap (device_sampler.h), so the oracle is the only comparison available -- parity unpinned beyond dimension 1000."""
    b = scenes.volume_cornell_cfg5(1.0)
    integ = gpu.VolPathIntegrator(8, 1.0, "spatial")
    img, st = integ.Render(gpu.Scene(b), 96, 96, 256, spp_begin=0, spp_end=8)
    oimg, ost = ol.OracleScene(b).render(integ, 96, 96, 256, spp_begin=0, spp_end=8)
    assert st["rays_closest"] == ost["rays_closest"]
    assert biteq(img[..., :3], oimg[..., :3])
    # sharded and sample-range renders recombine exactly
    acc = np.zeros_like(img)
    for rk in range(2):
        part, _ = integ.Render(gpu.Scene(b), 96, 96, 256, spp_begin=0, spp_end=8, shard_index=rk, shard_count=2, shard_rows=4)
        acc += part
    assert (acc.view(np.uint32) == img.view(np.uint32)).all()


@pytest.mark.parametrize("cap", ["0", "5", None])
def test_volpath_step_cap_and_packing_over_several_passes(gpu, monkeypatch, cap):
    """k_vol_media parks a tracking loop that is still running after GNXR_VOLMEDIA_STEP_CAP steps (default 64) and goes on with it in the
    next round; k_vol_pack moves the live paths between two sets of state arrays and leaves copies of such records behind.  Three passes
    of 262 k paths (packing engages, slots are reused by other paths) must give the oracle's image and ray count with no cap, with a cap
    that nearly every segment hits several times, and with the shipped one."""
    if cap is None: monkeypatch.delenv("GNXR_VOLMEDIA_STEP_CAP", raising=False)
    else: monkeypatch.setenv("GNXR_VOLMEDIA_STEP_CAP", cap)
    b = scenes.volume_cornell_cfg5(1.0)
    integ = gpu.VolPathIntegrator(8, 1.0, "spatial")
    img, st = integ.Render(gpu.Scene(b), 256, 256, 256, spp_begin=0, spp_end=12, samples_per_pass=4)
    oimg, ost = ol.OracleScene(b).render(integ, 256, 256, 256, spp_begin=0, spp_end=12)
    assert st["rays_closest"] == ost["rays_closest"]
    assert biteq(img[..., :3], oimg[..., :3])
    # the same range again on the same scene object: the same paths land in the same slots, where the first render left its records
    scene = gpu.Scene(b)
    for _ in range(2):
        img2, st2 = integ.Render(scene, 256, 256, 256, spp_begin=0, spp_end=12, samples_per_pass=4)
        assert st2["rays_closest"] == ost["rays_closest"] and biteq(img2[..., :3], oimg[..., :3])


def test_path_integrator_passes_through_medium_boundaries(gpu):
    """PathIntegrator on the volume scene: null-material boundaries are skipped (PathIntegrator.cpp:121-126)."""
    b = scenes.volume_cornell(sigma_a=(0.5,) * 3, sigma_s=(3.5,) * 3)
    integ = gpu.PathIntegrator(8, 1.0, "spatial")
    img, st = integ.Render(gpu.Scene(b), 64, 64, 16)
    oimg, ost = ol.OracleScene(b).render(integ, 64, 64, 16)
    r, mx = rmse(img, oimg)
    assert r < RMSE_TOL and mx < MAXABS_TOL, (r, mx)
    assert abs(st["rays_closest"] - ost["rays_closest"]) <= RAYS_TOL * ost["rays_closest"]


@pytest.mark.parametrize("name", ["cornell", "zoo", "cfg1", "sphere_mirror", "sphere_glass", "sky"])
def test_whitted_matches_reference(gpu, name):
    """cfg 1 (WhittedIntegrator) on the device: the depth-first recursion as a per-path state machine.  cornell /
    zoo / cfg1 are golden images of the restated recursion on the reference's classes (cfg1: the recorded 1 048 576 /
    2 028 213 rays); the sphere scenes ("6 quads + 1 Sphere" of BASELINE configs[0]) and the sky-lit box compare with the oracle."""
    g = golden("render_whitted.npz")
    if name == "cfg1":
        img, st = gpu.WhittedIntegrator(5).Render(gpu.Scene(scenes.cornell()), 256, 256, 16)
        assert (st["rays_closest"], st["rays_any"]) == (1048576, 2028213)
        assert abs(float(img[..., :3].astype(np.float64).sum()) - float(g["cfg1_checksum"])) < 1e-6
        assert biteq(img[::4, ::4, :3], g["cfg1_thumb"])
        return
    if name in ("cornell", "zoo"):
        W, H, spp, depth = (int(v) for v in g[name + "_cfg"])
        b = scenes.cornell() if name == "cornell" else scenes.material_zoo()
        img, st = gpu.WhittedIntegrator(depth).Render(gpu.Scene(b), W, H, spp)
        assert (st["rays_closest"], st["rays_any"]) == tuple(int(v) for v in g[name + "_rays"])
        assert biteq(img[..., :3], g[name][..., :3])
        return
    b = scenes.cornell(sky=True) if name == "sky" else scenes.cornell_sphere(name.split("_")[1])
    integ = gpu.WhittedIntegrator(5)
    img, st = integ.Render(gpu.Scene(b), 96, 96, 16)
    oimg, ost = ol.OracleScene(b).render(integ, 96, 96, 16)
    assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
    assert biteq(img[..., :3], oimg[..., :3])

@pytest.mark.parametrize("name", ["cornell_all", "cornell_one", "zoo_all", "zoo_one", "sphere_all", "env_one", "env_all", "depth2_all"])
def test_direct_lighting_matches_reference(gpu, name):
    """DirectLightingIntegrator on the device, sharing Whitted's depth-first state machine: UniformSampleAllLights
    with the sampler's 2D arrays (Light::nSamples = 5 per area light) and their Get2D fallback, or UniformSampleOneLight, each with
    full EstimateDirect records (shadow + MIS ray).  cornell / zoo are golden images of the restated integrator on the reference's
    classes; the sphere, env-lit and depth-2 (arrays used up at the third vertex) cases compare with the oracle."""
    g = golden("render_direct.npz")
    if name in g.files:
        W, H, spp, depth = (int(v) for v in g[name + "_cfg"])
        scene, strat = name.split("_")
        b = scenes.cornell() if scene == "cornell" else scenes.material_zoo()
        img, st = gpu.DirectLightingIntegrator(strat, depth).Render(gpu.Scene(b), W, H, spp)
        assert (st["rays_closest"], st["rays_any"]) == tuple(int(v) for v in g[name + "_rays"])
        assert biteq(img[..., :3], g[name][..., :3])
        return
    b = {"sphere_all": lambda: scenes.cornell_sphere("glass"), "env_one": lambda: scenes.dragon_cornell(2000, "glass+metal", env=os.path.join(GOLDEN, "env_100x50.hdr")),
         "env_all": lambda: scenes.dragon_cornell(2000, "glass+metal", env=os.path.join(GOLDEN, "env_100x50.hdr")),
         "depth2_all": scenes.material_zoo}[name]()
    integ = gpu.DirectLightingIntegrator(name.split("_")[1], 2 if name == "depth2_all" else 5)
    img, st = integ.Render(gpu.Scene(b), 96, 64, 8, samples_per_pass=3)
    oimg, ost = ol.OracleScene(b).render(integ, 96, 64, 8)
    assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"]) and st["rays_any"] > 0
    assert biteq(img[..., :3], oimg[..., :3])

@pytest.mark.parametrize("name", ["path", "whitted", "direct_all", "volpath", "direct_one", "lens_whitted", "lens_volpath"])
def test_textured_materials_match_reference(gpu, name):
    """image textures on the device (csrc/device_texture.h): ImageTexture / UVMapping2D / MIPMap EWA + trilinear
    lookups, camera ray differentials, ComputeDifferentials and their propagation through specular reflection / transmission.
    path / whitted / direct_all / volpath are golden images the reference's own ImageTexture, MIPMap, camera and Interaction code
    produced; direct_one and the thin-lens camera variants (the lens branch of the offset rays) compare with the oracle."""
    g = golden("render_textured.npz")
    W, H, spp, depth = (int(v) for v in g["cfg"])
    b = scenes.textured_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr"))
    mk = {"path": lambda: gpu.PathIntegrator(depth, 1.0, "spatial"), "whitted": lambda: gpu.WhittedIntegrator(depth),
          "direct_all": lambda: gpu.DirectLightingIntegrator("all", depth), "direct_one": lambda: gpu.DirectLightingIntegrator("one", depth),
          "volpath": lambda: gpu.VolPathIntegrator(depth, 1.0, "spatial")}
    if name in g.files:
        img, st = mk[name]().Render(gpu.Scene(b), W, H, spp)
        assert (st["rays_closest"], st["rays_any"]) == tuple(int(v) for v in g[name + "_rays"])
        assert biteq(img[..., :3], g[name][..., :3])
        return
    if name.startswith("lens_"):
        b.set_camera(eye=(0.3, 0.2, 4.5), look=(0, -0.5, 0), fov=75.0, lens_radius=0.15, focal_distance=6.0)
    integ = mk[name.split("_", 1)[1] if name.startswith("lens_") else name]()
    img, st = integ.Render(gpu.Scene(b), 80, 56, 8, samples_per_pass=3)
    oimg, ost = ol.OracleScene(b).render(integ, 80, 56, 8)
    assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
    assert biteq(img[..., :3], oimg[..., :3])

@pytest.mark.parametrize("name", ["path", "whitted", "direct_one", "volpath", "direct_all"])
def test_per_vertex_uvs_match_reference(gpu, name):
    """TriangleMesh::uv on the device: triangles with uvs of their own go through the general shade queue, which derives dpdu /
    dpdv (and with them every shading frame) and the texture coordinates from the stored corner uvs, degenerate uvs included.
    Golden images of the reference's classes; direct_all compares with the oracle."""
    g = golden("render_textured_uv.npz")
    W, H, spp, depth = (int(v) for v in g["cfg"])
    b = scenes.textured_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr"), uv_quads=True)
    integ = {"path": lambda: gpu.PathIntegrator(depth, 1.0, "spatial"), "whitted": lambda: gpu.WhittedIntegrator(depth),
             "direct_all": lambda: gpu.DirectLightingIntegrator("all", depth), "direct_one": lambda: gpu.DirectLightingIntegrator("one", depth),
             "volpath": lambda: gpu.VolPathIntegrator(depth, 1.0, "spatial")}[name]()
    img, st = integ.Render(gpu.Scene(b), W, H, spp)
    if name in g.files:
        assert (st["rays_closest"], st["rays_any"]) == tuple(int(v) for v in g[name + "_rays"])
        assert biteq(img[..., :3], g[name][..., :3])
    else:
        oimg, ost = ol.OracleScene(b).render(integ, W, H, spp)
        assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
        assert biteq(img[..., :3], oimg[..., :3])

@pytest.mark.parametrize("name", ["path", "whitted", "direct_all", "volpath", "direct_one"])
def test_per_vertex_normals_match_reference(gpu, name):
    """TriangleMesh::n on the device: the general shade queue reads the per-corner normals, builds the interpolated shading frame,
    dndu / dndv (ray differentials of the specular children) and the flipped geometric normal (etaScale, medium interfaces).
    Golden images of the reference's classes; direct_one compares with the oracle."""
    g = golden("render_smooth.npz")
    W, H, spp, depth = (int(v) for v in g["cfg"])
    b = scenes.smooth_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr"))
    integ = {"path": lambda: gpu.PathIntegrator(depth, 1.0, "spatial"), "whitted": lambda: gpu.WhittedIntegrator(depth),
             "direct_all": lambda: gpu.DirectLightingIntegrator("all", depth), "direct_one": lambda: gpu.DirectLightingIntegrator("one", depth),
             "volpath": lambda: gpu.VolPathIntegrator(depth, 1.0, "spatial")}[name]()
    scene = gpu.Scene(b)
    img, st = integ.Render(scene, W, H, spp)
    if name in g.files:
        assert (st["rays_closest"], st["rays_any"]) == tuple(int(v) for v in g[name + "_rays"])
        assert biteq(img[..., :3], g[name][..., :3])
    else:
        oimg, ost = ol.OracleScene(b).render(integ, W, H, spp)
        assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
        assert biteq(img[..., :3], oimg[..., :3])
        # Aggregate seam: isect->n of a smooth-shaded triangle is the geometric normal flipped onto the shading side
        rays = scenes.random_rays(20000, seed=4)
        gh, oh = scene.Intersect(rays), ol.OracleScene(b).Intersect(rays)
        assert (gh["prim"] == oh["prim"]).all() and biteq(gh["t"], oh["t"]) and biteq(gh["n"], oh["n"])

@pytest.mark.parametrize("name", ["path_spatial", "path_power", "path_uniform", "whitted", "direct_all", "direct_one", "volpath"])
def test_delta_lights_match_reference(gpu, name):
    """Point / Spot / Distant lights on the device (device_lights.h light_sample<LT_DELTA>, the IsDeltaLight branches of the three
    EstimateDirect restatements, Power and the device-built spatial light grid): golden images of the reference's light classes."""
    g = golden("render_delta.npz")
    W, H, spp, depth = (int(v) for v in g["cfg"])
    integ = {"path_spatial": lambda: gpu.PathIntegrator(depth, 1.0, "spatial"), "path_power": lambda: gpu.PathIntegrator(depth, 1.0, "power"),
             "path_uniform": lambda: gpu.PathIntegrator(depth, 1.0, "uniform"), "whitted": lambda: gpu.WhittedIntegrator(depth),
             "direct_all": lambda: gpu.DirectLightingIntegrator("all", depth), "direct_one": lambda: gpu.DirectLightingIntegrator("one", depth),
             "volpath": lambda: gpu.VolPathIntegrator(depth, 1.0, "spatial")}[name]()
    scene = gpu.Scene(scenes.delta_cornell())
    img, st = integ.Render(scene, W, H, spp)
    assert (st["rays_closest"], st["rays_any"]) == tuple(int(v) for v in g[name + "_rays"])
    assert biteq(img[..., :3], g[name][..., :3])
    if name == "path_spatial":   # the light grid built on the device equals the host restatement, delta lights included
        dev, host = scene.light_grid_table("spatial", on_host=False), scene.light_grid_table("spatial", on_host=True)
        assert biteq(dev, host)

@pytest.mark.parametrize("name", ["path", "whitted", "volpath", "lens_whitted", "lens_path"])
def test_orthographic_camera_matches_reference(gpu, name):
    """OrthographicCamera on the device (camera_ray / camera_ray_diff): golden images of the reference's camera class; the thin-lens
    variants (CreateOrthographicCamera hard-codes lensRadius = 0) compare with the oracle."""
    g = golden("render_ortho.npz")
    W, H, spp, depth = (int(v) for v in g["cfg"])
    b = scenes.textured_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr"))
    lens = name.startswith("lens_")
    b.set_camera(eye=(0.2, 0.1, 5.0), look=(0.0, -0.2, 0.0), orthographic=True, lens_radius=0.2 if lens else 0.0, focal_distance=6.0)
    key = name.split("_")[-1]
    integ = {"path": lambda: gpu.PathIntegrator(depth, 1.0, "spatial"), "whitted": lambda: gpu.WhittedIntegrator(depth),
             "volpath": lambda: gpu.VolPathIntegrator(depth, 1.0, "spatial")}[key]()
    img, st = integ.Render(gpu.Scene(b), W, H, spp)
    if not lens:
        assert (st["rays_closest"], st["rays_any"]) == tuple(int(v) for v in g[key + "_rays"])
        assert biteq(img[..., :3], g[key][..., :3])
    else:
        oimg, ost = ol.OracleScene(b).render(integ, W, H, spp)
        assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
        assert biteq(img[..., :3], oimg[..., :3])


@pytest.mark.parametrize("kind", ["matte", "mirror", "glass", "medium"])
def test_sphere_matches_oracle(gpu, kind):
    """pbrt-v3 quadratic sphere (parity with the reference unpinned: its Sphere is a stub).  The device is
    pinned to the CPU restatement: hit records bit for bit, images and ray counts bit for bit."""
    b = scenes.cornell_sphere(kind)
    scene, osc = gpu.Scene(b), ol.OracleScene(b)
    rays = scenes.random_rays(20000, seed=9)
    gh, oh = scene.Intersect(rays), osc.Intersect(rays)
    assert (gh["prim"] == oh["prim"]).all() and (gh["prim"] == b.desc().n_triangles).sum() > 1000
    assert biteq(gh["t"], oh["t"]) and biteq(gh["n"], oh["n"])
    seg = scenes.random_rays(20000, seed=10, tmax=1.5)
    assert (scene.IntersectP(seg) == osc.IntersectP(seg)).all()
    integ = gpu.VolPathIntegrator(8, 1.0, "spatial") if kind == "medium" else gpu.PathIntegrator(8, 1.0, "spatial")
    img, st = integ.Render(scene, 96, 96, 16)
    oimg, ost = osc.render(integ, 96, 96, 16)
    assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
    assert biteq(img[..., :3], oimg[..., :3])


@pytest.mark.parametrize("name", ["cornell", "env", "mesh"])
def test_light_grid_device_equals_host(gpu, name):
    """SpatialLightDistribution::ComputeDistribution for all voxels: the device kernel and the host restatement (which the
    golden images pinned in earlier builds) produce the same table, bit for bit."""
    if name == "cornell":
        b = scenes.cornell()
    elif name == "env":
        b = scenes.cornell(sky=True)
        b.AddInfLight(os.path.join(GOLDEN, "env_100x50.hdr"))
    else:
        b = _scene("mesh2k")
    scene = gpu.Scene(b)
    dev = scene.light_grid_table("spatial", on_host=False)
    host = scene.light_grid_table("spatial", on_host=True)
    assert dev.size == host.size and dev.size > 1000
    assert biteq(dev, host)


def test_randomised_sweep_against_oracle(gpu):
    """30 random (scene, integrator, image size, depth, rr threshold, light strategy, sample range, shard, pass size) cases:
    images and ray counts equal the oracle's bit for bit (tests/dev_sweep.py runs larger sweeps)."""
    import dev_sweep
    bad = dev_sweep.run_sweep(seed=7, ncase=30, verbose=False)
    assert not bad, bad


@pytest.mark.parametrize("name", ["cornell", "mesh2k"])
def test_pipelined_passes_equal_one_pass(gpu, name):
    """The device-driven path loop (csrc/api.hip: up to eight sub-passes alive at once in regions of the state arrays, queue counts on the
    device, k_queue_merge / k_loop_tail) against the same samples rendered as ONE pass and against the oracle: images and ray counts bit for
    bit, for sub-pass sizes that divide the sample range, that do not, for every number of sub-passes in flight, and for a sample sub-range."""
    b = scenes.cornell() if name == "cornell" else _scene("mesh2k")
    scene = gpu.Scene(b)
    integ = gpu.PathIntegrator(8, 1.0, "spatial")
    W, H, spp = 48, 40, 7
    one, st1 = integ.Render(scene, W, H, spp, samples_per_pass=spp)
    oimg, ost = ol.OracleScene(b).render(integ, W, H, spp)
    assert (st1["rays_closest"], st1["rays_any"]) == (ost["rays_closest"], ost["rays_any"]) and biteq(one, oimg)
    for k in (1, 2, 3, 4):
        img, st = integ.Render(scene, W, H, spp, samples_per_pass=k)
        assert st["passes"] == (spp + k - 1) // k
        assert (st["rays_closest"], st["rays_any"]) == (st1["rays_closest"], st1["rays_any"]), k
        assert biteq(img, one), k
    for inflight in (1, 2, 3, 5, 8):
        img, st = integ.Render(scene, W, H, spp, samples_per_pass=1, passes_in_flight=inflight)
        assert st["passes_in_flight"] == min(inflight, spp) and st["passes"] == spp and st["state_bytes"] > 0
        assert (st["rays_closest"], st["rays_any"]) == (st1["rays_closest"], st1["rays_any"]), inflight
        assert biteq(img, one), inflight
    a, sa = integ.Render(scene, W, H, spp, spp_begin=2, spp_end=6, samples_per_pass=1)
    c, sc_ = integ.Render(scene, W, H, spp, spp_begin=2, spp_end=6, samples_per_pass=4)
    assert biteq(a, c) and (sa["rays_closest"], sa["rays_any"]) == (sc_["rays_closest"], sc_["rays_any"])


def test_render_reserve(gpu):
    """gnxr_render_reserve allocates the state of a render without rendering: it succeeds for every integrator, leaves results unchanged,
    and rejects the parameters gnxr_render rejects."""
    b = scenes.cornell()
    scene = gpu.Scene(b)
    for integ in (gpu.PathIntegrator(8, 1.0, "spatial"), gpu.VolPathIntegrator(8, 1.0, "spatial"), gpu.WhittedIntegrator(5)):
        integ.Reserve(scene, 40, 30, 6, samples_per_pass=2)
    integ = gpu.PathIntegrator(8, 1.0, "spatial")
    img, st = integ.Render(scene, 40, 30, 6, samples_per_pass=2)
    oimg, ost = ol.OracleScene(b).render(integ, 40, 30, 6)
    assert biteq(img, oimg) and (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
    with pytest.raises(Exception):
        integ.Reserve(scene, 0, 30, 6)


def test_edge_cases(gpu):
    scene = gpu.Scene(scenes.cornell())
    img, st = gpu.PathIntegrator(0).Render(scene, 8, 8, 2)               # maxDepth 0
    oimg, ost = ol.OracleScene(scenes.cornell()).render(gpu.PathIntegrator(0), 8, 8, 2)
    assert st["rays_any"] == 0 and st["rays_closest"] == 128 and biteq(img, oimg)
    img, _ = gpu.PathIntegrator(8).Render(scene, 1, 1, 4)                # 1x1 image
    assert np.isfinite(img).all()
    img, _ = gpu.PathIntegrator(8).Render(scene, 33, 7, 3, samples_per_pass=2)   # ragged: spp not a multiple of the pass size
    oimg, _ = ol.OracleScene(scenes.cornell()).render(gpu.PathIntegrator(8), 33, 7, 3)
    assert rmse(img, oimg)[0] < RMSE_TOL
    with pytest.raises(gpu.GnxrError):
        gpu.PathIntegrator(8).Render(scene, 0, 8, 1)
    with pytest.raises(gpu.GnxrError):
        gpu.PathIntegrator(8).Render(scene, 8, 8, 4, spp_begin=3, spp_end=2)
    # descriptions compile_scene would otherwise read out of bounds are refused: camera medium, grid media without data
    bv = scenes.volume_cornell()
    dv = bv.desc()
    dv.camera_medium = 7
    with pytest.raises(gpu.GnxrError, match="camera_medium"):
        gpu.Scene(dv)
    dv = bv.desc()
    dv.grid_density = None
    with pytest.raises(gpu.GnxrError, match="grid_density"):
        gpu.Scene(dv)
    dv = bv.desc()
    dv.media[0].density_offset = -5
    with pytest.raises(gpu.GnxrError, match="density_offset"):
        gpu.Scene(dv)
    dv.media[0].density_offset = 0
    # a scene without lights renders black and traces no shadow rays
    b = gpu.SceneBuilder()
    w = b.MatteMaterial((0.5, 0.5, 0.5))
    b.AddCornell(w, w, w)
    img, st = gpu.PathIntegrator(3).Render(gpu.Scene(b), 16, 16, 2)
    assert (img[..., :3] == 0).all() and st["rays_any"] == 0


def test_framebuffer_update_matches_ui_framebuffer(gpu):
    """FrameBuffer::update_f_u_c (ui/FrameBuffer.h:127-149): running mean over Render() calls + 1 - expf(-4x) -> uint8, against the
    oracle's restatement of that function, BYTE for byte (the device's expf carries glibc's bits), over five folded frames whose
    values span the tone curve (0, denormal-small, around every uint8 step, saturating)."""
    rng = np.random.default_rng(3)
    H, W = 37, 53
    mean_d = np.zeros((H, W, 4), np.float32)
    mean_o = np.zeros((H, W, 4), np.float32)
    for k in range(1, 6):
        f = (rng.random((H, W, 4)) ** 3 * (4.0 if k % 2 else 0.2)).astype(np.float32)
        f[0, :8, :3] = [[0.0], [1e-30], [1e-8], [0.25 * np.log(2.0)], [10.0], [1e30], [0.001], [88.0]]   # edge values (the mean of non-negative frames stays >= 0)
        u_d = gpu.framebuffer_update(mean_d, f, k)
        u_o = ol.oracle_framebuffer_update(mean_o, f, k)
        assert biteq(mean_d[..., :3], mean_o[..., :3])
        assert (u_d == u_o).all() and (u_d[..., 3] == 255).all()
    assert len(np.unique(u_d[..., :3])) > 200   # the comparison covered the whole tone curve
    # and the closed form agrees to within the rounding of expf (sanity of the restatement itself)
    tm = (1.0 - np.exp(-mean_o[..., :3].astype(np.float64) / 0.25)) * 255
    assert np.abs(u_o[..., :3].astype(np.float64) - np.floor(tm)).max() <= 1


def test_c_caller_renders_the_default_scene(gpu, tmp_path):
    """tools/gnxr_cli.c -- RenderThread::run in plain C against the C ABI: scene authoring through gnxr_builder_*, two Render()
    iterations folded by gnxr_framebuffer_update (FrameBuffer::update_f_u_c) and written as PNG.  The file must decode to the RGBA8
    plane the same pipeline produces through the Python mirror."""
    import struct, subprocess, sys, zlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as ge
    cli = ge.build_cli()
    W, H, spp = 64, 48, 8
    for integ_name, integ in [("whitted", gpu.WhittedIntegrator(5)), ("path", gpu.PathIntegrator(5, 1.0, "spatial"))]:
        out = tmp_path / f"cli_{integ_name}.png"
        r = subprocess.run([cli, "--width", str(W), "--height", str(H), "--spp", str(spp), "--frames", "2", "--integrator", integ_name, "--sky", "--out", str(out)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        b = scenes.cornell(light_material="dragon", sky=True)
        scene = gpu.Scene(b)
        mean = np.zeros((H, W, 4), np.float32)
        for f in (1, 2):
            img, _ = integ.Render(scene, W, H, spp)
            rgba8 = gpu.framebuffer_update(mean, img, f)
        data = out.read_bytes()
        pos, idat = 8, b""
        while pos < len(data):
            n, typ = struct.unpack(">I4s", data[pos:pos + 8])
            if typ == b"IDAT": idat += data[pos + 8:pos + 8 + n]
            pos += 12 + n
        raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, W * 4 + 1)
        assert (raw[:, 1:].reshape(H, W, 4) == rgba8).all() and rgba8[..., :3].max() > 100


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_multi_device_render_behind_the_abi(gpu, devices):
    """gnxr_init_devices: one process, several devices behind one scene handle -- rows dealt round-robin, shards rendered
    concurrently on a host thread + stream per device, FrameBuffer assembled on the first device by strided peer copies.  On a one-GPU
    box the list names device 0 two / three times (independent replicas and streams on the same card: the code path of a node); the
    image and the ray counts must equal the single-device render bit for bit, for every integrator, odd heights included, and an
    externally sharded call (rank shards x device shards) must still recombine."""
    cases = [(scenes.dragon_cornell(2000, "zoo", env=os.path.join(GOLDEN, "env_100x50.hdr"), mesh_path=os.path.join(GOLDEN, "mesh_2k.3d")), gpu.PathIntegrator(8, 1.0, "spatial"), (97, 61, 6)),
             (scenes.volume_cornell(sigma_a=(0.5,) * 3, sigma_s=(3.5,) * 3, g_grid=0.3), gpu.VolPathIntegrator(6, 1.0, "spatial"), (64, 47, 4)),
             (scenes.material_zoo(), gpu.WhittedIntegrator(5), (80, 33, 4))]
    singles = []
    for b, integ, (W, H, spp) in cases:
        singles.append(integ.Render(gpu.Scene(b), W, H, spp))
    try:
        gpu.init_devices(devices)
        for (b, integ, (W, H, spp)), (ref, rst) in zip(cases, singles):
            scene = gpu.Scene(b)
            img, st = integ.Render(scene, W, H, spp)
            assert (st["rays_closest"], st["rays_any"], st["camera_samples"]) == (rst["rays_closest"], rst["rays_any"], rst["camera_samples"])
            assert (img.view(np.uint32) == ref.view(np.uint32)).all()
            acc = np.zeros_like(img)
            for rk in range(2):   # two "ranks", each spreading its rows over the device list
                part, _ = integ.Render(scene, W, H, spp, shard_index=rk, shard_count=2, shard_rows=1)
                acc += part
            assert (acc.view(np.uint32) == ref.view(np.uint32)).all()
            # the Aggregate seam answers from the primary replica
            rays = scenes.random_rays(2000, seed=3)
            assert (scene.Intersect(rays)["prim"] == gpu.Scene(b).Intersect(rays)["prim"]).all()
        with pytest.raises(gpu.GnxrError):
            integ.Render(scene, W, H, spp, shard_index=0, shard_count=2, shard_rows=4)   # multi-device deals single rows
        with pytest.raises(gpu.GnxrError):
            gpu.init_devices([0, 99])
    finally:
        gpu.init(0)


def test_multi_device_assembly_without_peer_access(gpu, monkeypatch):
    """A device pair for which peer access cannot be enabled assembles its rows through a pinned host buffer instead of a peer copy
    (gnxr_init_devices records the outcome per pair; GNXR_NO_PEER=1 forces that route, here with device 0 listed three times).  Same
    bits as one device, ragged height included."""
    b = scenes.dragon_cornell(2000, "zoo", env=os.path.join(GOLDEN, "env_100x50.hdr"), mesh_path=os.path.join(GOLDEN, "mesh_2k.3d"))
    integ, (W, H, spp) = gpu.PathIntegrator(8, 1.0, "spatial"), (97, 61, 6)
    ref, rst = integ.Render(gpu.Scene(b), W, H, spp)
    monkeypatch.setenv("GNXR_NO_PEER", "1")
    try:
        gpu.init_devices([0, 0, 0])
        img, st = integ.Render(gpu.Scene(b), W, H, spp)
        assert (st["rays_closest"], st["rays_any"]) == (rst["rays_closest"], rst["rays_any"])
        assert (img.view(np.uint32) == ref.view(np.uint32)).all()
    finally:
        monkeypatch.delenv("GNXR_NO_PEER")
        gpu.init(0)


def test_light_grid_with_many_lights(gpu):
    """More lights than the unrolled k_light_grid handles (16): the general kernel uses the voxel's table slice as scratch and must
    produce the host restatement's bits; the render (spatial light selection over 22 lights) matches the oracle."""
    b = scenes.cornell()
    white = b.MatteMaterial(scenes.WHITE, 60.0)
    quad = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    d0 = b.desc().n_lights
    for k in range(10):   # ten more emissive quads = 20 more one-triangle lights, spread over the box
        x, z = -2.0 + 0.4 * k, -1.5 + 0.3 * k
        v = np.array([[x, 2.3 - 0.05 * k, z], [x + 0.3, 2.3 - 0.05 * k, z], [x + 0.3, 2.3 - 0.05 * k, z + 0.3], [x, 2.3 - 0.05 * k, z + 0.3]], np.float32)
        b.add_emissive_mesh(v, quad, white, (3.0 + k, 4.0, 5.0 - 0.3 * k))
    assert b.desc().n_lights == d0 + 20
    scene = gpu.Scene(b)
    dev, host = scene.light_grid_table("spatial", on_host=False), scene.light_grid_table("spatial", on_host=True)
    assert biteq(dev, host)
    for integ in (gpu.PathIntegrator(5, 1.0, "spatial"), gpu.DirectLightingIntegrator("all", 3), gpu.WhittedIntegrator(3)):   # "all": 110 records per vertex
        img, st = integ.Render(scene, 64, 48, 8)
        oimg, ost = ol.OracleScene(b).render(integ, 64, 48, 8)
        assert (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
        assert biteq(img[..., :3], oimg[..., :3])


@pytest.mark.parametrize("name", ["mesh2k", "smooth", "middle", "equal", "sah"])
def test_hlbvh_build_matches_reference(gpu, name):
    """GNXR_BVH_HLBVH: Morton codes + radix sort on the device, LBVH treelets and the SAH upper tree on the host must
    give the LinearBVHNode[] and primitive order of the reference's BVHAccel(prims, 1, SplitMethod::HLBVH), bit for bit; rendering
    through it gives the reference's image and ray counts, and the traversal statistics of the oracle walking the same tree."""
    g = golden("bvh_hlbvh.npz")
    W, H, spp, depth = (int(v) for v in g["cfg"])
    b = scenes.smooth_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr")) if name == "smooth" else \
        scenes.dragon_cornell(2000, "glass+metal", mesh_path=os.path.join(GOLDEN, "mesh_2k.3d"))
    b.set_bvh_split_method({"middle": "middle", "equal": "equal_counts", "sah": "sah"}.get(name, "hlbvh"))   # the other BVHAccel::SplitMethod values ride along
    scene = gpu.Scene(b)
    bounds, meta, order = scene.bvh()
    if name == "sah":   # the default build against the dump of BVHAccel(prims, 1) that pins the oracle (bvh_mesh2k.npz)
        gs = golden("bvh_mesh2k.npz")
        assert biteq(bounds, gs["bounds"]) and (meta == gs["meta"]).all() and (order == gs["order"]).all()
        return
    assert bounds.shape == g[name + "_bounds"].shape and biteq(bounds, g[name + "_bounds"])
    assert (meta == g[name + "_meta"]).all() and (order == g[name + "_order"]).all()
    integ = gpu.PathIntegrator(depth, 1.0, "spatial")
    img, st = integ.Render(scene, W, H, spp)
    assert (st["rays_closest"], st["rays_any"]) == tuple(int(v) for v in g[name + "_rays"])
    assert biteq(img[..., :3], g[name + "_img"][..., :3])

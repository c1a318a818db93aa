"""Checks that need the reference tree itself (/root/reference) and the compiled reference (oracle/_ref/gnx_ref): they run
in the development container only and are skipped on the GPU box, where neither exists.  They pin BASELINE config 4's
inputs at their own size: the reference's Resources/MonValley1000.hdr (1000 x 500, run-length encoded) through the builder's
RGBE reader, the 1000x500 -> 1024x512 Lanczos resample + Distribution2D of InfiniteAreaLight, and Plastic / Disney / Glass /
Metal lit by it -- oracle against the reference's own classes, bit for bit.  No reference file is copied: the HDR is read
where it lies."""
import os
import struct
import tempfile

import numpy as np
import pytest

import oracle_lib as ol
import scenes
from conftest import GOLDEN

RES = "/root/reference/Resources"
needs_ref = pytest.mark.skipif(not (ol.have_ref() and os.path.isdir(RES)), reason="needs /root/reference and oracle/_ref/gnx_ref (development container only)")


def _stbi(path):
    raw = ol.run_ref(None, "hdr", None, [path])   # stbi_loadf, 3rd/stb_image.h, as lights/InfiniteAreaLight.cpp:27 calls it
    w, h = struct.unpack("<2i", raw[:8])
    return np.frombuffer(raw[8:], np.float32).reshape(h, w, 3)


def _builder_pixels(gx, path):
    b = gx.SceneBuilder()
    b.AddInfLight(path)
    d = b.desc()
    return np.ctypeslib.as_array(d.env_rgb, shape=(d.env_height, d.env_width, 3)).copy()


@needs_ref
@pytest.mark.parametrize("name", ["MonValley1000.hdr", "TropicalRuins1000.hdr"])
def test_rgbe_reader_equals_stbi_on_the_reference_hdr(gx, name):
    """The reference's own environment maps are new-style RLE Radiance files: the RLE branch of read_rgbe
    (csrc/scene_builder.cpp) must return stbi_loadf's floats bit for bit."""
    path = os.path.join(RES, name)
    mine, ref = _builder_pixels(gx, path), _stbi(path)
    assert mine.shape == ref.shape == (500, 1000, 3)
    assert (mine.view(np.uint32) == ref.view(np.uint32)).all()


@needs_ref
def test_rgbe_reader_equals_stbi_on_the_synthetic_stand_in(gx, tmp_path):
    """bench.py --workload cfg4 and the 1080p GPU test use a synthetic 1000 x 500 map written RLE-encoded by tests/scenes.py."""
    p = str(tmp_path / "env.hdr")
    scenes.write_rgbe(p, scenes.synthetic_env(1000, 500), rle=True)
    mine, ref = _builder_pixels(gx, p), _stbi(p)
    assert (mine.view(np.uint32) == ref.view(np.uint32)).all()


@needs_ref
@pytest.mark.parametrize("strategy", ["spatial", "power"])
def test_oracle_equals_reference_classes_under_the_reference_hdr(gx, strategy):
    """cfg 4 at a size the CPU finishes in seconds: 2 k-triangle mesh in Glass / Metal / Plastic / Disney quarters inside the
    Cornell box, InfiniteAreaLight(MonValley1000.hdr) with the transform of ui/ModelList.cpp:172-179 -- the oracle against the
    restated Render / Li loop on the reference's own InfiniteAreaLight, MIPMap, Distribution2D, BSDF and BVH classes."""
    b = scenes.dragon_cornell(2000, "zoo", env=os.path.join(RES, "MonValley1000.hdr"), mesh_path=os.path.join(GOLDEN, "mesh_2k.3d"))
    W, H, spp, depth = 96, 54, 4, 8
    integ = gx.PathIntegrator(depth, 1.0, strategy)
    oimg, ost = ol.OracleScene(b).render(integ, W, H, spp)
    with tempfile.TemporaryDirectory() as td:
        sp = os.path.join(td, "scene.bin")
        ol.write_scene_file(b, sp)
        raw = ol.run_ref(sp, "render", None, [W, H, spp, depth, 1.0, {"spatial": 0, "uniform": 1, "power": 2}[strategy], 4, 0])
    rimg = np.frombuffer(raw[:W * H * 16], np.float32).reshape(H, W, 4)
    cnt = np.frombuffer(raw[W * H * 16:W * H * 16 + 16], np.uint64)
    assert (int(cnt[0]), int(cnt[1])) == (ost["rays_closest"], ost["rays_any"])
    assert (oimg[..., :3].view(np.uint32) == rimg[..., :3].view(np.uint32)).all()
    assert oimg[..., :3].max() > 0.5   # the environment actually lights the scene


@needs_ref
def test_volume_reader_on_the_reference_density_file(gx):
    """Resources/density_render.70.volume (100 x 100 x 40, CRLF) through gnxr_builder_add_volume_file: header and all 400 000
    densities equal the committed data fixture tests/golden/density_70.npz (which cfg 5's goldens were rendered from)."""
    g = np.load(os.path.join(GOLDEN, "density_70.npz"))
    b = gx.SceneBuilder()
    m = b.add_volume_file(os.path.join(RES, "density_render.70.volume"))
    d = b.desc()
    md = d.media[m]
    assert (md.nx, md.ny, md.nz) == (int(g["nx"]), int(g["ny"]), int(g["nz"])) == (100, 100, 40)
    assert list(md.sigma_a) == [10.0] * 3 and list(md.sigma_s) == [90.0] * 3
    got = np.ctypeslib.as_array(d.grid_density, shape=(100 * 100 * 40,))
    assert (got.view(np.uint32) == g["density"].reshape(-1).astype(np.float32).view(np.uint32)).all()


def _ref_render(b, args, W, H):
    with tempfile.TemporaryDirectory() as td:
        sp = os.path.join(td, "scene.bin")
        ol.write_scene_file(b, sp)
        raw = ol.run_ref(sp, "render", None, args)
    img = np.frombuffer(raw[:W * H * 16], np.float32).reshape(H, W, 4)
    cnt = np.frombuffer(raw[W * H * 16:W * H * 16 + 16], np.uint64)
    return img, (int(cnt[0]), int(cnt[1]))


@needs_ref
def test_volpath_at_config_size_oracle_equals_reference_classes(gx):
    """cfg 5's geometry at cfg 5's image size (512 x 512; 2 spp of HaltonSampler(2), sigma scaled by 0.05 so that no sample passes
    Halton dimension 1000, where the reference is undefined): the two restatements of VolPathIntegrator::Li
    (integrators/VolPathIntegrator.cpp:24-159) -- the oracle's and ref_driver.cpp's on the reference's own GridDensityMedium /
    HomogeneousMedium / HenyeyGreenstein / BVH / light classes -- agree bit for bit, ray counts included.  The 64 x 64 fixture
    (render_vol.npz) pins the same pair at fixture size; this closes the gap up to the configuration's own size."""
    b = scenes.volume_cornell_cfg5(0.05, golden_dir=GOLDEN)
    W, H, spp, depth = 512, 512, 2, 8
    ol.olib().gnxo_max_dimension(1)
    oimg, ost = ol.OracleScene(b).render(gx.VolPathIntegrator(depth, 1.0, "spatial"), W, H, spp, threads=8)
    assert ol.olib().gnxo_max_dimension(1) < 1000
    rimg, rrays = _ref_render(b, [W, H, spp, depth, 1.0, 0, 8, 1], W, H)
    assert rrays == (ost["rays_closest"], ost["rays_any"])
    assert (oimg[..., :3].view(np.uint32) == rimg[..., :3].view(np.uint32)).all()
    assert rrays[0] > 2 * W * H   # media segments make several closest-hit rays per camera sample


@needs_ref
def test_direct_lighting_all_at_256_oracle_equals_reference_classes(gx):
    """DirectLightingIntegrator, UniformSampleAll, on the one-of-each-material scene at 256 x 256 (integrators/
    DirectLightingIntegrator.cpp:30-64 + the sample arrays of core/Sampler.cpp:52-72): oracle == the restatement on the reference's
    Sampler / BSDF / light / BVH classes, bit for bit -- four times the linear size of the render_direct.npz fixture."""
    b = scenes.material_zoo()
    W, H, spp, depth = 256, 256, 4, 5
    oimg, ost = ol.OracleScene(b).render(gx.DirectLightingIntegrator("all", depth), W, H, spp, threads=8)
    rimg, rrays = _ref_render(b, [W, H, spp, depth, 1.0, 0, 8, 3, 0], W, H)
    assert rrays == (ost["rays_closest"], ost["rays_any"])
    assert (oimg[..., :3].view(np.uint32) == rimg[..., :3].view(np.uint32)).all()

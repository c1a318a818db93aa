"""Host-side logic that runs without a GPU: scene authoring (ModelList/MaterialList mirror), `.3d` IO,
sharding arithmetic, argument validation of the oracle side of the harness."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
import scenes
from conftest import GOLDEN


def test_cornell_authoring_matches_modellist(gx):
    b = scenes.cornell()
    d = b.desc()
    assert (d.n_triangles, d.n_vertices, d.n_materials, d.n_lights) == (12, 36, 3, 2)       # ModelList.cpp:71-147
    v = np.ctypeslib.as_array(d.vertices, shape=(d.n_vertices, 3))
    assert v[:30].min() == -2.5 and v[:30].max() == 2.5                                       # box side 5 centred at origin
    assert np.allclose(v[30:, 1], 2.45) and np.abs(v[30:, [0, 2]]).max() == np.float32(1.4)  # light quad, ModelList.cpp:125-129
    mats = np.ctypeslib.as_array(d.tri_material, shape=(12,))
    assert list(mats) == [0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 0, 0]                                # tris 6,7 red; 8,9 blue
    lights = np.ctypeslib.as_array(d.tri_light, shape=(12,))
    assert list(lights[10:]) == [0, 1] and (lights[:10] == -1).all()
    assert d.lights[0].le[0] == 5.0 and d.lights[0].two_sided == 0
    cam = d.camera
    assert tuple(cam.eye) == (0.0, 0.0, 5.0) and cam.fov_deg == 90.0 and cam.lens_radius == 0.0  # RenderThread.cpp:60-68


def test_material_factories(gx):
    b = gx.SceneBuilder()
    p, m, g = b.getPurplePlasticMaterial(), b.getYelloMetalMaterial(), b.getWhiteGlassMaterial()
    d = b.desc()
    assert np.allclose(list(d.materials[p].kd), [0.35, 0.12, 0.48]) and d.materials[p].remap_roughness == 1
    assert np.allclose(list(d.materials[p].ks), [0.65, 0.88, 0.52])
    assert np.allclose(list(d.materials[m].eta), [0.2, 0.2, 0.8]) and d.materials[m].urough == np.float32(0.15)
    assert d.materials[g].eta[0] == 1.5 and d.materials[g].remap_roughness == 0 and d.materials[g].urough == np.float32(0.1)
    assert all(d.materials[i].has_bump == 1 for i in (p, m, g))


def test_synthetic_mesh_is_deterministic_and_loader_scales(tmp_path, gx):
    a, b2 = tmp_path / "a.3d", tmp_path / "b.3d"
    gx.write_synthetic_3d(str(a), 2000, 7)
    gx.write_synthetic_3d(str(b2), 2000, 7)
    assert a.read_bytes() == b2.read_bytes() == open(os.path.join(GOLDEN, "mesh_2k.3d"), "rb").read()
    head = a.read_text().split("\n")[:2]
    assert head[0].startswith("vertex ") and head[1].startswith("face ")                   # plyRead.h:23-28
    b = gx.SceneBuilder()
    m = b.MatteMaterial((0.5, 0.5, 0.5))
    b.AddModel(str(a), m)
    d = b.desc()
    raw = np.loadtxt(str(a), skiprows=2, max_rows=d.n_vertices, dtype=np.float32)
    v = np.ctypeslib.as_array(d.vertices, shape=(d.n_vertices, 3))
    expect = raw * np.float32(20)
    expect[:, 1] = expect[:, 1] + np.float32(-2.9)                                          # x20 (plyRead.h:38), y-2.9 (ModelList.cpp:56)
    assert (v == expect).all()
    assert v.min() > -2.5 and v.max() < 2.5                                                  # sits inside the box


def test_bad_inputs_are_rejected(gx, tmp_path):
    b = gx.SceneBuilder()
    with pytest.raises(gx.GnxrError):
        b.AddModel(str(tmp_path / "missing.3d"), 0)
    with pytest.raises(gx.GnxrError):
        b.add_mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 5]], 0)     # index out of range
    with pytest.raises(gx.GnxrError):
        b.AddInfLight(str(tmp_path / "missing.hdr"))


def test_shard_rows_partition_the_image():
    from gnxraytracer_amd.distributed import shard_row_index
    for H, world, sr in [(1080, 8, 1), (1080, 7, 1), (270, 4, 8), (5, 8, 1), (64, 2, 3)]:
        rows = [shard_row_index(H, r, world, sr) for r in range(world)]
        assert sorted(sum(rows, [])) == list(range(H))


def test_oracle_sharded_render_equals_full(gx):
    """Every pixel depends only on (x, y, sample): shards and spp ranges recombine bit-exactly."""
    b = scenes.cornell()
    osc = ol.OracleScene(b)
    integ = gx.PathIntegrator(8, 1.0, "spatial")
    full, st = osc.render(integ, 48, 40, 8)
    acc = np.zeros_like(full)
    rays = 0
    for r in range(3):
        part, s = osc.render(integ, 48, 40, 8, shard_index=r, shard_count=3, shard_rows=2)
        acc += part
        rays += s["rays_closest"] + s["rays_any"]
    assert (acc.view(np.uint32) == full.view(np.uint32)).all()
    assert rays == st["rays_closest"] + st["rays_any"]


def test_oracle_edge_cases(gx):
    integ0 = gx.PathIntegrator(0, 1.0, "spatial")
    b = scenes.cornell()
    osc = ol.OracleScene(b)
    img, st = osc.render(integ0, 8, 8, 2)                         # maxDepth 0: emission of directly seen lights only
    assert st["rays_any"] == 0 and st["rays_closest"] == 8 * 8 * 2
    img1, _ = osc.render(gx.PathIntegrator(8), 1, 1, 4)             # 1x1 image
    assert np.isfinite(img1).all()
    # rays that start outside and point away miss everything
    rays = gx.make_rays([[0, 0, 100]] * 4, [[0, 0, 1]] * 4)
    assert (osc.Intersect(rays)["prim"] == -1).all() and (osc.IntersectP(rays) == 0).all()
    # zero-length batch
    assert len(osc.Intersect(np.zeros((0, 8), np.float32))) == 0


def test_sphere_against_analytic_hits(gx):
    """the oracle's pbrt-v3 sphere against the closed-form ray/sphere solution evaluated in float64."""
    import oracle_lib as ol
    import scenes
    c, r = np.array([0.6, -1.5, 0.2], np.float32).astype(np.float64), 1.0   # the float32 centre the scene stores
    b = scenes.cornell_sphere("matte", center=tuple(c), radius=r)
    osc = ol.OracleScene(b)
    rng = np.random.default_rng(4)
    n = 20000
    o = rng.uniform(-2.3, 2.3, (n, 3)).astype(np.float32)
    tgt = (c + rng.normal(size=(n, 3)) * 0.8).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = gx.make_rays(o, d)
    hits = osc.Intersect(rays)
    nt = b.desc().n_triangles
    on_sphere = hits["prim"] == nt
    assert on_sphere.sum() > n // 4
    od, dd = o.astype(np.float64) - c, d.astype(np.float64)
    A = (dd * dd).sum(1); B = 2 * (dd * od).sum(1); Cq = (od * od).sum(1) - r * r
    disc = B * B - 4 * A * Cq
    t0 = (-B - np.sqrt(np.maximum(disc, 0))) / (2 * A); t1 = (-B + np.sqrt(np.maximum(disc, 0))) / (2 * A)
    t_exact = np.where(t0 > 0, t0, t1)
    sel = on_sphere & (disc > 1e-3)            # away from grazing incidence, where t is ill-conditioned
    assert np.allclose(hits["t"][sel], t_exact[sel], rtol=1e-6, atol=1e-6)
    p = o[sel].astype(np.float64) + dd[sel] * hits["t"][sel][:, None].astype(np.float64)
    n_exact = (p - c) / np.linalg.norm(p - c, axis=1, keepdims=True)
    assert np.abs(np.abs((hits["n"][sel] * n_exact).sum(1)) - 1).max() < 1e-5      # the normal is radial
    assert np.abs(np.linalg.norm(p - c, axis=1) - r).max() < 1e-5                  # the hit lies on the sphere
    # rays whose exact solution misses the sphere (or lies behind a wall) never report it
    miss = (disc < -1e-6) | (t1 < -1e-6)
    assert not (hits["prim"][miss] == nt).any()
    # IntersectP agrees with Intersect for segments that end inside / beyond the sphere
    seg = gx.make_rays(o, (tgt - o), 1.0)
    occ = osc.IntersectP(seg)
    full = osc.Intersect(seg)
    assert ((full["prim"] >= 0) == (occ != 0)).all()


def test_framebuffer_save_png_round_trips(gx, tmp_path):
    """FrameBuffer::saveToFile (ui/FrameBuffer.cpp:6-9): the PNG decodes to the RGBA8 plane that was handed in."""
    import struct, zlib
    rng = np.random.default_rng(2)
    for (h, w) in [(1, 1), (37, 53), (300, 420)]:     # the last one spans several stored deflate blocks
        img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        p = tmp_path / f"fb_{w}x{h}.png"
        gx.save_png(p, img)
        data = p.read_bytes()
        assert data[:8] == bytes([0x89]) + b"PNG\r\n\x1a\n"
        pos, idat, ihdr = 8, b"", None
        while pos < len(data):
            n, typ = struct.unpack(">I4s", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            crc = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
            assert zlib.crc32(typ + body) & 0xffffffff == crc
            if typ == b"IHDR": ihdr = struct.unpack(">IIBBBBB", body)
            if typ == b"IDAT": idat += body
            pos += 12 + n
        assert ihdr == (w, h, 8, 6, 0, 0, 0)
        raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * 4 + 1)
        assert (raw[:, 0] == 0).all() and (raw[:, 1:].reshape(h, w, 4) == img).all()


def test_bench_launch_plan():
    """bench.py --gpus N: alone it must start its own ranks, under a launcher it must agree with WORLD_SIZE."""
    import bench
    a = lambda *v: bench.parse(list(v))
    assert bench.launch_plan(a(), {}) == ("run", 1)
    assert bench.launch_plan(a("--gpus", "1"), {"WORLD_SIZE": "1"}) == ("run", 1)
    assert bench.launch_plan(a("--gpus", "8"), {}) == ("spawn", 8)
    assert bench.launch_plan(a("--gpus", "8"), {"WORLD_SIZE": "8", "RANK": "3"}) == ("run", 8)
    assert bench.launch_plan(a("--gpus", "8"), {"WORLD_SIZE": "4"})[0] == "error"
    assert bench.launch_plan(a("--gpus", "1"), {"WORLD_SIZE": "2"})[0] == "error"      # `bench.py` under a 2-rank launcher without --gpus 2
    assert bench.launch_plan(a("--gpus", "2"), {"WORLD_SIZE": "x"})[0] == "error"
    with pytest.raises(SystemExit):
        a("--gpus", "0")
    c5 = a("--workload", "cfg5")
    assert (c5.width, c5.height, c5.spp, c5.spp_per_step, c5.steps) == (512, 512, 256, 256, 1)
    assert a("--workload", "cfg4").workload == "cfg4" and a().workload == "cfg3"


def test_bench_spawns_its_own_ranks(tmp_path, monkeypatch):
    """`python bench.py --gpus 2` with no launcher around it starts `python -m torch.distributed.run --nproc-per-node 2 bench.py ...`
    as a child and returns the child's exit code (the ranks themselves need GPUs; here the command line is what is checked)."""
    import bench
    seen = {}
    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    rc = bench.spawn_ranks(2, ["--gpus", "2", "--steps", "4"])
    cmd = seen["cmd"]
    assert rc == 7 and cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "4"]
    assert cmd[-5] == os.path.abspath(bench.__file__) and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the whole-node case the driver may ask for: `python bench.py --gpus 8 --steps K --warmup W` with no launcher around it
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "8", "--steps", "8", "--warmup", "2"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    cmd = seen["cmd"]
    assert ex.value.code == 7   # the child's exit code is this process's
    assert cmd[0] == bench.sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[cmd.index(os.path.abspath(bench.__file__)) + 1:] == ["--gpus", "8", "--steps", "8", "--warmup", "2"]


def test_radiance_writer_round_trips_through_the_builder(gx, tmp_path):
    """tests/scenes.write_rgbe (RLE and flat) -> the builder's RGBE reader: both encodings decode to the same floats, within RGBE's
    8-bit mantissa of the source, zeros (exponent byte 0) and long runs included."""
    img = scenes.synthetic_env(200, 64)
    px = {}
    for rle in (True, False):
        p = str(tmp_path / f"e{int(rle)}.hdr")
        scenes.write_rgbe(p, img, rle=rle)
        b = gx.SceneBuilder()
        b.AddInfLight(p)
        d = b.desc()
        px[rle] = np.ctypeslib.as_array(d.env_rgb, shape=(d.env_height, d.env_width, 3)).copy()
    assert os.path.getsize(str(tmp_path / "e1.hdr")) < os.path.getsize(str(tmp_path / "e0.hdr"))   # runs were found
    assert (px[True].view(np.uint32) == px[False].view(np.uint32)).all()
    m = img.max(axis=2, keepdims=True)
    assert (np.abs(px[True] - img) <= m / 128 + 1e-30).all() and (px[True][-1] == 0).all()


def test_volume_file_reader(gx, tmp_path):
    """gnxr_builder_add_volume_file: the `.volume` text format of Resources/density_render.70.volume (CRLF line ends, header rows
    nx/ny/nz, p0, p1, sigma_a, sigma_s, then the densities x-fastest) -> a GRID medium placed by the header's own box."""
    d = scenes.synthetic_density(7, 5, 3)   # [nz, ny, nx]
    nz, ny, nx = d.shape
    p = tmp_path / "smoke.volume"
    with open(p, "w", newline="") as f:
        f.write(f"nx {nx} ny {ny} nz {nz}\r\np0 0.010000 0.020000 0.030000 \r\np1 1.990000 1.500000 0.790000 \r\nsigma_a 10 10 10\r\nsigma_s 90 80 70\r\n")
        f.write(" ".join(repr(float(v)) for v in d.reshape(-1)) + " \r\n")
    b = gx.SceneBuilder()
    w = b.MatteMaterial((0.5, 0.5, 0.5))
    b.AddCornell(w, w, w)
    m = b.add_volume_file(str(p), g=0.3, sigma_scale=0.5)
    desc = b.desc()
    md = desc.media[m]
    assert (md.type, md.nx, md.ny, md.nz) == (gx._abi.MEDIUM_GRID, nx, ny, nz) and md.g == np.float32(0.3)
    assert list(md.sigma_a) == [5.0, 5.0, 5.0] and list(md.sigma_s) == [45.0, 40.0, 35.0]
    m2w = np.array(list(md.medium_to_world), np.float32).reshape(4, 4)
    p0, p1 = np.float32([0.01, 0.02, 0.03]), np.float32([1.99, 1.5, 0.79])
    assert (np.diag(m2w)[:3] == p1 - p0).all() and (m2w[:3, 3] == p0).all() and m2w[3, 3] == 1
    got = np.ctypeslib.as_array(desc.grid_density, shape=(nx * ny * nz,))
    assert (got.view(np.uint32) == d.reshape(-1).view(np.uint32)).all()
    m3 = b.add_volume_file(str(p), medium_to_world=np.eye(4))     # explicit placement, second grid appended behind the first
    assert b.desc().media[m3].density_offset == nx * ny * nz and b.desc().media[m3].medium_to_world[0] == 1.0
    for bad in ("nx 2 ny 2\n", "nx 2 ny 2 nz 2\np0 0 0 0\np1 1 1 1\nsigma_a 1 1 1\nsigma_s 1 1 1\n1 2 3\n"):
        q = tmp_path / "bad.volume"
        q.write_text(bad)
        with pytest.raises(gx.GnxrError):
            b.add_volume_file(str(q))


def test_bench_call_ranges():
    """bench.py submits the steps of a contiguous sample range as one library call: the ranges tile the requested steps exactly, never
    cross the end of the Halton range, and wrap around it."""
    import bench
    assert bench.call_ranges(0, 8, 128, 1024) == [(0, 1024, 8)]
    assert bench.call_ranges(0, 20, 128, 1024) == [(0, 1024, 8), (0, 1024, 8), (0, 512, 4)]
    assert bench.call_ranges(0, 5, 128, 1024) == [(0, 640, 5)]
    assert bench.call_ranges(6, 4, 128, 1024) == [(768, 1024, 2), (0, 256, 2)]
    assert bench.call_ranges(0, 1, 256, 256) == [(0, 256, 1)]
    assert bench.call_ranges(0, 3, 100, 256) == [(0, 200, 2), (200, 256, 1)]   # a range the step does not divide: the last call is short
    for i0, n, sps, spp in ((0, 7, 64, 1024), (3, 40, 128, 1024), (0, 9, 32, 96)):
        r = bench.call_ranges(i0, n, sps, spp)
        assert sum(m for _, _, m in r) == n and all(0 <= a < b <= spp for a, b, _ in r)

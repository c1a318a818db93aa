"""Scene set-ups of BASELINE.json's configs, authored through the product's SceneBuilder
(the mirror of ui/ModelList.cpp / ui/MaterialList.cpp / ui/RenderThread.cpp:46-187)."""
import os

import numpy as np

import gnxraytracer_amd as gx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WHITE = (0.91, 0.91, 0.91)
RED = (0.9, 0.1, 0.17)
BLUE = (0.14, 0.21, 0.87)
DRAGON_GREEN = (0.2, 0.8, 0.2)


def cornell(light_material="white", sky=False):
    """cfg 1/2: Cornell box, 10 wall triangles + 2 light triangles, Matte sigma=60 (RenderThread.cpp:79-133)."""
    b = gx.SceneBuilder()
    white = b.MatteMaterial(WHITE, 60.0)
    red = b.MatteMaterial(RED, 60.0)
    blue = b.MatteMaterial(BLUE, 60.0)
    b.AddCornell(red, blue, white)
    lm = white if light_material == "white" else b.MatteMaterial(DRAGON_GREEN, 60.0)
    b.AddAreaLight(lm)
    if sky:
        b.AddSkyLight()
    return b


def material_zoo():
    """Cornell box whose walls carry one of each material (function-level BSDF parity)."""
    b = gx.SceneBuilder()
    white = b.MatteMaterial(WHITE, 60.0)
    lambert = b.MatteMaterial(RED, 0.0)
    mirror = b.MirrorMaterial(DRAGON_GREEN)
    glass = b.getWhiteGlassMaterial()
    metal = b.getYelloMetalMaterial()
    plastic = b.getPurplePlasticMaterial()
    sglass = b.add_material(type=gx._abi.MAT_GLASS, kr=(0.98,) * 3, kt=(0.98,) * 3, eta=(1.5, 0, 0), urough=0.0, vrough=0.0)
    disney = disney_preset(b)
    first = b.AddCornell(lambert, mirror, white)
    # floor: glass / smooth glass, ceiling: metal / plastic, back wall (tris 4,5): disney
    d = b.desc()
    mats = np.ctypeslib.as_array(d.tri_material, shape=(d.n_triangles,))
    mats[first + 0] = glass
    mats[first + 1] = sglass
    mats[first + 2] = metal
    mats[first + 3] = plastic
    mats[first + 4] = disney
    mats[first + 5] = disney_thin_preset(b)
    b.AddAreaLight(white)
    return b


def disney_preset(b):
    """Build-defined Disney preset for cfg 4 (the reference never instantiates DisneyMaterial)."""
    return b.DisneyMaterial(color=(0.8, 0.45, 0.2), metallic=0.3, eta=1.5, roughness=0.4, specularTint=0.2, anisotropic=0.3,
                            sheen=0.5, sheenTint=0.5, clearcoat=0.6, clearcoatGloss=0.8, specTrans=0.25, thin=False,
                            flatness=0.0, diffTrans=0.0)


def disney_thin_preset(b):
    return b.DisneyMaterial(color=(0.3, 0.6, 0.8), metallic=0.0, eta=1.4, roughness=0.5, specularTint=0.0, anisotropic=0.0,
                            sheen=0.0, sheenTint=0.5, clearcoat=0.0, clearcoatGloss=1.0, specTrans=0.4, thin=True,
                            flatness=0.3, diffTrans=0.8)


def synthetic_mesh_path(n_tris, seed=1, cache_dir=None):
    cache_dir = cache_dir or os.path.join(ROOT, "gpurun_out", "_meshes")
    os.makedirs(cache_dir, exist_ok=True)
    p = os.path.join(cache_dir, f"synthetic_dragon_{n_tris}_{seed}.3d")
    if not os.path.exists(p):
        gx.write_synthetic_3d(p, n_tris, seed)
    return p


def write_rgbe(path, rgb, rle=True):
    """Radiance .hdr writer (32-bit_rle_rgbe).  rle=True writes the new-style run-length-encoded scanlines every Radiance tool
    (and the reference's Resources/*.hdr) uses -- the branch of the builder's reader that stbi's hdr loader calls
    `stbi__hdr_load` RLE -- rle=False the flat layout.  Data only: no reference file is copied."""
    rgb = np.asarray(rgb, np.float64)
    h, w, _ = rgb.shape
    m = rgb.max(axis=2)
    ok = m > 1e-32
    mant, ex = np.frexp(np.where(ok, m, 1.0))           # m = mant * 2^ex, mant in [0.5, 1)
    scale = np.where(ok, mant * 256.0 / np.where(ok, m, 1.0), 0)
    px = np.zeros((h, w, 4), np.uint8)
    px[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    px[..., 3] = np.where(ok, ex + 128, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        if not rle or w < 8 or w >= 32768:
            f.write(px.tobytes())
            return
        for j in range(h):
            out = bytearray([2, 2, (w >> 8) & 0xff, w & 0xff])
            for c in range(4):
                line = px[j, :, c]
                # run boundaries, then greedy: runs of >= 4 equal bytes become (128 + n, value), the rest literal blocks of <= 128
                edges = np.flatnonzero(np.diff(line)) + 1
                starts = np.concatenate([[0], edges]); ends = np.concatenate([edges, [w]])
                lit_start = 0
                def flush_lit(a, b):
                    while a < b:
                        n = min(128, b - a)
                        out.append(n); out.extend(line[a:a + n].tobytes()); a += n
                for a, b in zip(starts.tolist(), ends.tolist()):
                    if b - a >= 4:
                        flush_lit(lit_start, a)
                        v = int(line[a])
                        while a < b:
                            n = min(127, b - a)
                            out.append(128 + n); out.append(v); a += n
                        lit_start = b
                flush_lit(lit_start, w)
            f.write(bytes(out))


def synthetic_env(w=1000, h=500, seed=4):
    """Deterministic outdoor-style lat-long radiance map at the size of the reference's Resources/MonValley1000.hdr (1000 x 500:
    not a power of two, so MIPMap's Lanczos resample to 1024 x 512 runs): graded sky, a small bright sun (the importance table must
    find it; peak 200, MonValley's is 69), ground, a few soft clouds.  Stand-in for the HDR, which cannot travel to the GPU box."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    u, v = (xx + 0.5) / w, (yy + 0.5) / h
    sky = np.stack([0.25 + 0.5 * v, 0.45 + 0.45 * v, 1.05 - 0.25 * v], 2) * (1.3 - 0.6 * v[..., None])
    ground = np.stack([0.22 + 0.05 * np.sin(40 * u), 0.19 + 0.04 * np.cos(31 * u), 0.12 + 0.0 * u], 2) * (0.6 + 0.8 * (v[..., None] - 0.5))
    img = np.where((v > 0.52)[..., None], ground, sky)
    rng = np.random.default_rng(seed)
    for _ in range(7):   # clouds
        cu, cv, r = rng.random(), 0.1 + 0.3 * rng.random(), 0.03 + 0.06 * rng.random()
        du = np.minimum(np.abs(u - cu), 1 - np.abs(u - cu))
        img += 0.9 * np.exp(-((du / (2 * r)) ** 2 + ((v - cv) / r) ** 2))[..., None] * (v < 0.5)[..., None]
    su, sv = 0.31, 0.22   # the sun: ~3 texels wide, two orders of magnitude above the sky
    img += (2.0e2 * np.exp(-(((u - su) * w) ** 2 + ((v - sv) * h) ** 2) / 4.0))[..., None] * np.array([1.0, 0.93, 0.82])
    img[h - 6:, :, :] = 0.0   # a black band (RGBE exponent 0: the `e == 0` branch of the decoder) with long runs
    return img.astype(np.float32)


def synthetic_env_path(w=1000, h=500, cache_dir=None):
    cache_dir = cache_dir or os.path.join(ROOT, "gpurun_out", "_meshes")
    os.makedirs(cache_dir, exist_ok=True)
    p = os.path.join(cache_dir, f"synthetic_env_{w}x{h}.hdr")
    if not os.path.exists(p):
        tmp = p + f".{os.getpid()}.tmp"
        write_rgbe(tmp, synthetic_env(w, h), rle=True)
        os.replace(tmp, p)
    return p


def dragon_cornell(n_tris=100000, material="glass", env=None, extra_materials=False, mesh_path=None):
    """cfg 3 (Glass + Metal halves) / cfg 4 (+ InfiniteAreaLight, Plastic, Disney).  The mesh is the seeded
    synthetic stand-in for the absent dragon.3d; AddModel comes first, as in RenderThread.cpp:119-133."""
    b = gx.SceneBuilder()
    white = b.MatteMaterial(WHITE, 60.0)
    red = b.MatteMaterial(RED, 60.0)
    blue = b.MatteMaterial(BLUE, 60.0)
    glass = b.getWhiteGlassMaterial()
    metal = b.getYelloMetalMaterial()
    path = mesh_path or synthetic_mesh_path(n_tris)
    first = b.AddModel(path, glass)
    d = b.desc()
    nt = d.n_triangles
    mats = np.ctypeslib.as_array(d.tri_material, shape=(nt,))
    if material == "glass+metal":
        mats[first + nt // 2:first + nt] = metal          # build-defined split: second half of the faces
    elif material == "metal":
        mats[first:first + nt] = metal
    elif material == "zoo":
        plastic = b.getPurplePlasticMaterial()
        dis = disney_preset(b)
        q = nt // 4
        mats[first + q:first + 2 * q] = metal
        mats[first + 2 * q:first + 3 * q] = plastic
        mats[first + 3 * q:first + nt] = dis
    b.AddCornell(red, blue, white)
    b.AddAreaLight(white)
    if env:
        b.AddInfLight(env)
    return b


def random_rays(n, seed=0, inside=2.4, tmax=np.inf):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-inside, inside, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return gx.make_rays(o, d.astype(np.float32), tmax)


def box_mesh(lo, hi):
    """12 triangles with outward geometric normals (n = Cross(p0-p2, p1-p2), shape/Triangle.cpp:223)."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    v = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]], np.float32)
    quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 4, 7, 3)]   # -z +z -y +y +x -x, CCW from outside
    idx = []
    for a, b, c, d in quads:
        idx += [[a, b, c], [a, c, d]]
    return v, np.array(idx, np.int32)


def synthetic_density(nx=24, ny=24, nz=12, seed=3):
    """Smooth seeded smoke-like density in [0, 1] (stand-in for Resources/density_render.70.volume in tests)."""
    z, y, x = np.mgrid[0:nz, 0:ny, 0:nx].astype(np.float32)
    x, y, z = (x + 0.5) / nx, (y + 0.5) / ny, (z + 0.5) / nz
    rng = np.random.default_rng(seed)
    d = np.zeros_like(x)
    for _ in range(5):
        c = rng.random(3)
        r = 0.15 + 0.2 * rng.random()
        d += np.exp(-((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / (r * r))
    d = (d / d.max()).astype(np.float32)
    return np.ascontiguousarray(d)   # [nz, ny, nx]: density[(z*ny + y)*nx + x], GridDensityMedium.h:34-38


def read_volume_file(path):
    """Resources/density_render.70.volume: `nx N ny N nz N / p0 / p1 / sigma_a / sigma_s` then nx*ny*nz floats (CRLF)."""
    toks = open(path).read().split()
    nx, ny, nz = int(toks[1]), int(toks[3]), int(toks[5])
    p0 = [float(t) for t in toks[7:10]]
    p1 = [float(t) for t in toks[11:14]]
    sa = [float(t) for t in toks[15:18]]
    ss = [float(t) for t in toks[19:22]]
    d = np.array(toks[22:22 + nx * ny * nz], np.float32)
    return dict(nx=nx, ny=ny, nz=nz, p0=p0, p1=p1, sigma_a=sa, sigma_s=ss, density=d)


def volume_cornell(density=None, sigma_a=(10, 10, 10), sigma_s=(90, 90, 90), g_grid=0.0, grid_lo=(-1.6, -2.4, -1.2), grid_hi=(0.2, -0.6, 0.4)):
    """cfg 5: Cornell + a null-material box filled with a GridDensityMedium + a second null-material box with the
    HomogeneousMedium(2.4, 1.4, 0.5) of ui/RenderThread.cpp:107.  mediumToWorld = Translate(lo) * Scale(hi - lo)."""
    b = cornell()
    if density is None:
        density = synthetic_density()
    nz, ny, nx = density.shape
    lo, hi = np.array(grid_lo, np.float32), np.array(grid_hi, np.float32)
    m2w = np.eye(4, dtype=np.float32)
    m2w[0, 0], m2w[1, 1], m2w[2, 2] = hi - lo
    m2w[0:3, 3] = lo
    grid = gx.Medium()
    grid.type = gx._abi.MEDIUM_GRID
    grid.nx, grid.ny, grid.nz = nx, ny, nz
    grid.sigma_a[:] = sigma_a
    grid.sigma_s[:] = sigma_s
    grid.g = g_grid
    grid.medium_to_world[:] = m2w.reshape(16)
    mg = b.add_medium(grid, density)
    hom = gx.Medium()
    hom.type = gx._abi.MEDIUM_HOMOGENEOUS
    hom.sigma_a[:] = (2.4, 2.4, 2.4)
    hom.sigma_s[:] = (1.4, 1.4, 1.4)
    hom.g = 0.5
    mh = b.add_medium(hom)
    v, i = box_mesh(grid_lo, grid_hi)
    b.add_mesh(v, i, -1, medium_inside=mg, medium_outside=-1)
    v, i = box_mesh((0.6, -2.4, -0.8), (1.9, -0.9, 0.6))
    b.add_mesh(v, i, -1, medium_inside=mh, medium_outside=-1)
    return b


def reference_density(golden_dir=None):
    """The reference's own density grid (Resources/density_render.70.volume, 100x100x40), kept as a data fixture."""
    g = np.load(os.path.join(golden_dir or os.path.join(ROOT, "tests", "golden"), "density_70.npz"))
    return np.ascontiguousarray(g["density"].reshape(int(g["nz"]), int(g["ny"]), int(g["nx"])))


def volume_cornell_cfg5(sigma_scale=1.0, golden_dir=None):
    """cfg 5 with the reference's density grid (sigma_a 10, sigma_s 90 from the file header) in a 2 x 2 x 0.8 box.
    sigma_scale < 1 keeps the delta-tracking loops below Halton dimension 1000, past which the reference reads
    PrimeSums out of bounds (undefined there; this build wraps, device_sampler.h)."""
    return volume_cornell(reference_density(golden_dir), sigma_a=(10 * sigma_scale,) * 3, sigma_s=(90 * sigma_scale,) * 3,
                          grid_lo=(-1.9, -2.4, -0.4), grid_hi=(0.1, -0.4, 0.4))


def cornell_sphere(kind="matte", center=(0.6, -1.5, 0.2), radius=1.0):
    """cfg 1 as BASELINE.json words it ("6 quads + 1 Sphere"): the Cornell box plus one pbrt-v3 sphere (the reference's own
    Sphere is an unfinished stub -- parity unpinned, see include/gnxr.h).  kind: matte | mirror | glass | medium."""
    b = cornell()
    if kind == "matte":
        m = b.MatteMaterial(DRAGON_GREEN, 60.0)
    elif kind == "mirror":
        m = b.MirrorMaterial((0.9, 0.9, 0.9))
    elif kind == "glass":
        m = b.add_material(type=gx._abi.MAT_GLASS, kr=(0.98,) * 3, kt=(0.98,) * 3, eta=(1.5, 0, 0), urough=0.0, vrough=0.0)
    else:
        m = -1
    mi = -1
    if kind == "medium":
        hom = gx.Medium()
        hom.type = gx._abi.MEDIUM_HOMOGENEOUS
        hom.sigma_a[:] = (0.4, 0.8, 1.2)
        hom.sigma_s[:] = (2.0, 1.6, 1.2)
        hom.g = 0.3
        mi = b.add_medium(hom)
    b.AddSphere(center, radius, m, medium_inside=mi, medium_outside=-1)
    return b


def cornell_no_lights():
    """walls only: scene.lights is empty (UniformSampleOneLight returns 0, Integrator.cpp:63; Whitted's light loop is empty)."""
    b = gx.SceneBuilder()
    white = b.MatteMaterial(WHITE, 60.0)
    b.AddCornell(b.MatteMaterial(RED, 60.0), b.MatteMaterial(BLUE, 60.0), white)
    return b


def cornell_in_fog():
    """the camera sits inside a thin HomogeneousMedium bounded by a large null-material box (Camera::medium != nullptr)."""
    b = cornell()
    hom = gx.Medium()
    hom.type = gx._abi.MEDIUM_HOMOGENEOUS
    hom.sigma_a[:] = (0.02, 0.03, 0.04)
    hom.sigma_s[:] = (0.10, 0.08, 0.06)
    hom.g = -0.2
    m = b.add_medium(hom)
    v, i = box_mesh((-6.0, -6.0, -6.0), (6.0, 6.0, 8.0))
    b.add_mesh(v, i, -1, medium_inside=m, medium_outside=-1)
    b.set_camera_medium(m)
    return b


def textured_cornell(tex_path, glass_sheet=True, uv_quads=False):
    """image-textured materials.  Back wall = the reference's getSmileFacePlasticMaterial (ui/MaterialList.cpp:31-46:
    Kd = Ks = one ImageTexture, EWA, Repeat) on `tex_path`; floor = Matte whose Kd is the same image tiled 3 x 3 through the
    trilinear filter with Clamp wrap, gamma and scale; left wall = mirror and a free-standing smooth-glass sheet so that Whitted /
    DirectLighting carry ray differentials through specular reflection and transmission onto the textures.  Triangles have no
    per-vertex uv in the reference, so every triangle shows the lower-right half of the image (Triangle::GetUVs defaults)."""
    b = gx.SceneBuilder()
    white = b.MatteMaterial(WHITE, 60.0)
    blue = b.MatteMaterial(BLUE, 60.0)
    mirror = b.MirrorMaterial((0.9, 0.9, 0.9))
    smile = b.getSmileFacePlasticMaterial(tex_path)
    floor_tex = b.add_image_texture(tex_path, su=3.0, sv=3.0, du=0.25, dv=0.1, trilinear=True, wrap="clamp", scale=0.8, gamma=True)
    floor = b.MatteMaterial(WHITE, 0.0)
    b.set_material_texture(floor, "kd", floor_tex)
    first = b.AddCornell(mirror, blue, white)
    d = b.desc()
    mats = np.ctypeslib.as_array(d.tri_material, shape=(d.n_triangles,))
    mats[first + 0] = floor
    mats[first + 1] = floor
    mats[first + 4] = smile
    mats[first + 5] = smile
    if glass_sheet:
        sglass = b.add_material(type=gx._abi.MAT_GLASS, kr=(0.98,) * 3, kt=(0.98,) * 3, eta=(1.5, 0, 0), urough=0.0, vrough=0.0)
        v = np.array([[0.3, -2.5, 0.4], [1.9, -2.5, -0.6], [1.9, 0.4, -0.6], [0.3, 0.4, 0.4]], np.float32)
        b.add_mesh(v, np.array([[0, 1, 2], [0, 2, 3]], np.int32), sglass)
    if uv_quads:
        # meshes WITH per-vertex uv (TriangleMesh::uv): a poster showing the whole image once, a Disney-coated panel whose uvs
        # run to 2.5 (Repeat) -- uv also sets dpdu / dpdv and with them the shading frame of every lobe --, and a panel whose
        # three uvs coincide (degenerate: Triangle::Intersect falls back to CoordinateSystem(ng))
        quad = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
        poster = np.array([[-2.0, -1.0, -2.2], [-0.2, -1.0, -2.3], [-0.2, 0.9, -2.3], [-2.0, 0.9, -2.2]], np.float32)
        b.add_mesh(poster, quad, smile, uv=[[0, 0], [1, 0], [1, 1], [0, 1]])
        panel = np.array([[0.9, -2.45, 1.2], [2.3, -2.45, 0.2], [2.3, -1.2, 0.0], [0.9, -1.2, 1.0]], np.float32)
        b.add_mesh(panel, quad, disney_preset(b), uv=[[0.25, 0.5], [2.5, 0.5], [2.5, 1.75], [0.25, 1.75]])
        flat = np.array([[-2.4, -2.4, 0.5], [-1.2, -2.4, 1.2], [-1.2, -1.5, 1.2], [-2.4, -1.5, 0.5]], np.float32)
        b.add_mesh(flat, quad, floor, uv=[[0.4, 0.6]] * 4)
    b.AddAreaLight(white)
    return b


def uv_sphere_mesh(center, radius, n_lat=8, n_lon=12):
    """Lat-long tessellated sphere with per-vertex uvs (phi / 2pi, theta / pi) and object-space normals (p - center) / r."""
    c = np.asarray(center, np.float64)
    verts, uvs, normals = [], [], []
    for i in range(n_lat + 1):
        th = np.pi * i / n_lat
        for j in range(n_lon + 1):
            ph = 2 * np.pi * j / n_lon
            n = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
            verts.append(c + radius * n); normals.append(n); uvs.append([j / n_lon, i / n_lat])
    tris = []
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * (n_lon + 1) + j, (i + 1) * (n_lon + 1) + j
            if i > 0: tris.append([a, b, a + 1])
            if i < n_lat - 1: tris.append([a + 1, b, b + 1])
    return np.array(verts, np.float32), np.array(tris, np.int32), np.array(uvs, np.float32), np.array(normals, np.float32)


def smooth_cornell(tex_path, medium_ball=True):
    """Per-vertex shading normals (TriangleMesh::n): coarse lat-long spheres that are smooth-shaded -- a mirror ball and a glass ball
    (the interpolated normal drives the specular directions, dndu / dndv the ray differentials of Whitted / DirectLighting, and the
    geometric normal is flipped onto the shading side, which the glass's etaScale and the medium interface read), a ball with normals
    AND uvs carrying the image-textured plastic, a scaled (non-uniform object-to-world) Disney ellipsoid, and -- for VolPath -- a
    null-material ball with normals around a HomogeneousMedium."""
    b = gx.SceneBuilder()
    white = b.MatteMaterial(WHITE, 60.0)
    red = b.MatteMaterial(RED, 60.0)
    blue = b.MatteMaterial(BLUE, 60.0)
    mirror = b.MirrorMaterial((0.9, 0.9, 0.9))
    sglass = b.add_material(type=gx._abi.MAT_GLASS, kr=(0.98,) * 3, kt=(0.98,) * 3, eta=(1.5, 0, 0), urough=0.0, vrough=0.0)
    smile = b.getSmileFacePlasticMaterial(tex_path)
    b.AddCornell(red, blue, white)
    v, t, uv, n = uv_sphere_mesh((-1.3, -1.6, -0.6), 0.85)
    b.add_mesh(v, t, mirror, normals=n)
    v, t, uv, n = uv_sphere_mesh((1.2, -1.5, 0.6), 0.9, 6, 9)
    b.add_mesh(v, t, sglass, normals=n)
    v, t, uv, n = uv_sphere_mesh((0.0, 0.3, -1.3), 0.8, 7, 10)
    b.add_mesh(v, t, smile, uv=uv, normals=n)
    v, t, uv, n = uv_sphere_mesh((0.0, 0.0, 0.0), 1.0, 6, 8)
    m = np.array([[0.7, 0, 0, 1.4], [0, 0.35, 0, 0.9], [0, 0, 0.5, -0.8], [0, 0, 0, 1]], np.float32)
    # per-vertex tangents (TriangleMesh::s) along the parallels; they vanish at the poles (the `ss.LengthSquared() > 0` fallback)
    tang = np.stack([-n[:, 2], np.zeros(len(n), np.float32), n[:, 0]], 1).astype(np.float32)
    b.add_mesh(v, t, disney_preset(b), object_to_world=m, uv=uv, normals=n, tangents=tang)
    # tangents WITHOUT normals (`mesh->n || mesh->s`): a flat-shaded metal panel whose anisotropic frame follows the given tangents
    quad = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    panel = np.array([[-2.3, -2.45, 1.9], [-1.1, -2.45, 1.9], [-1.1, -2.1, 0.9], [-2.3, -2.1, 0.9]], np.float32)
    aniso = b.add_material(type=gx._abi.MAT_METAL, eta=(0.2, 0.9, 1.1), k=(3.9, 2.4, 2.2), urough=0.05, vrough=0.3, remap_roughness=1)
    b.add_mesh(panel, quad, aniso, tangents=[[1, 0, 0.4], [0.8, 0, 0.6], [0.6, 0.1, 0.8], [0.9, 0, 0.3]])
    if medium_ball:
        hom = gx.Medium()
        hom.type = gx._abi.MEDIUM_HOMOGENEOUS
        hom.sigma_a[:] = (0.4, 0.5, 0.9)
        hom.sigma_s[:] = (1.6, 1.4, 0.8)
        hom.g = 0.3
        med = b.add_medium(hom)
        v, t, uv, n = uv_sphere_mesh((-0.3, -1.9, 1.4), 0.55, 6, 8)
        b.add_mesh(v, t, -1, medium_inside=med, medium_outside=-1, normals=n)
    b.AddAreaLight(white)
    return b


def delta_cornell(medium_ball=True):
    """Delta lights (EstimateDirect's IsDeltaLight branch, core/Integrator.cpp:157-158, 168): the reference's AddSpotLight and
    AddDistLight (ui/ModelList.cpp:149-161; their calls are commented out in RenderThread.cpp:138-141) plus a PointLight, next to the
    area light, in the material zoo -- and a HomogeneousMedium ball so that VolPath's handleMedia = true variant takes the branch too."""
    b = material_zoo()
    b.AddSpotLight()
    b.AddDistLight()
    b.add_delta_light("point", (6.0, 4.0, 2.5), light_to_world=[[1, 0, 0, -1.2], [0, 1, 0, -0.4], [0, 0, 1, 1.1], [0, 0, 0, 1]])
    b.add_delta_light("spot", (30.0, 30.0, 40.0), total_width=35.0, falloff_start=10.0,
                      light_to_world=[[0.8, 0, 0.6, 1.9], [0, 1, 0, 1.7], [-0.6, 0, 0.8, 2.0], [0, 0, 0, 1]])
    if medium_ball:
        hom = gx.Medium()
        hom.type = gx._abi.MEDIUM_HOMOGENEOUS
        hom.sigma_a[:] = (0.3, 0.4, 0.6)
        hom.sigma_s[:] = (1.2, 1.0, 0.7)
        hom.g = 0.2
        med = b.add_medium(hom)
        v, t, uv, n = uv_sphere_mesh((0.2, -1.4, 0.9), 0.7, 6, 8)
        b.add_mesh(v, t, -1, medium_inside=med, medium_outside=-1)
    return b

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

"""Ad-hoc GPU-vs-oracle check used during bring-up (the real tests are test_gpu_*.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gnxraytracer_amd as gx
import oracle_lib as ol, scenes

gx.init(0)
rng = np.random.default_rng(1)
for (W, H) in [(256, 256), (1920, 1080), (64, 64)]:
    n = 100000
    px = rng.integers(0, W, n); py = rng.integers(0, H, n); s = rng.integers(0, 1024, n); dim = rng.integers(0, 300, n)
    g = gx.sample_halton(W, H, px, py, s, dim); o = ol.oracle_halton(W, H, px, py, s, dim)
    print("halton", W, H, "biteq", (g.view(np.uint32) == o.view(np.uint32)).all(), flush=True)
b = scenes.cornell()
for (W, H) in [(256, 256), (1920, 1080)]:
    n = 50000
    px = rng.integers(0, W, n); py = rng.integers(0, H, n); s = rng.integers(0, 64, n)
    go, gd = gx.camera_rays(b.desc().camera, W, H, px, py, s); oo, od = ol.oracle_camera_rays(b.desc().camera, W, H, px, py, s)
    print("camrays", W, H, "biteq", (go.view(np.uint32) == oo.view(np.uint32)).all(), (gd.view(np.uint32) == od.view(np.uint32)).all(), flush=True)

def check_scene(b, name, sizes):
    t = time.time(); scene = gx.Scene(b); print(name, "scene create %.2fs" % (time.time() - t), scene.info(), flush=True)
    osc = ol.OracleScene(b)
    rays = scenes.random_rays(200000, seed=3)
    gh = scene.Intersect(rays); oh = osc.Intersect(rays)
    m = oh['prim'] >= 0
    print(name, "closest prim eq", (gh['prim'] == oh['prim']).mean(), "t biteq", (gh['t'][m].view(np.uint32) == oh['t'][m].view(np.uint32)).mean(),
          "b biteq", (gh['b0'][m].view(np.uint32) == oh['b0'][m].view(np.uint32)).mean(), "n biteq", (gh['n'][m].view(np.uint32) == oh['n'][m].view(np.uint32)).mean(), flush=True)
    srays = scenes.random_rays(200000, seed=4, tmax=1.5)
    print(name, "any eq", (scene.IntersectP(srays) == osc.IntersectP(srays)).mean(), flush=True)
    integ = gx.PathIntegrator(8, 1.0, "spatial")
    for (W, H, spp) in sizes:
        t = time.time(); img, st = integ.Render(scene, W, H, spp); tg = time.time() - t
        t = time.time(); oimg, ost = osc.render(integ, W, H, spp); to = time.time() - t
        d = img[..., :3] - oimg[..., :3]
        print(name, W, H, spp, "rays gpu", st['rays_closest'], st['rays_any'], "oracle", ost['rays_closest'], ost['rays_any'],
              "biteq frac %.5f" % (img.view(np.uint32) == oimg.view(np.uint32)).mean(), "rmse %.3e maxabs %.3e" % (np.sqrt((d ** 2).mean()), np.abs(d).max()),
              "t gpu %.3f (render %.3f) oracle %.2f" % (tg, st['seconds_render'], to), "launches", st['kernel_launches'], flush=True)
        nr = st['rays_closest'] + st['rays_any']
        print("   Mrays/s gpu %.1f  cpu(oracle,all cores) %.2f" % (nr / st['seconds_render'] / 1e6, (ost['rays_closest'] + ost['rays_any']) / ost['seconds_render'] / 1e6), flush=True)

check_scene(scenes.cornell(), "cornell", [(64, 64, 16), (256, 256, 64)])
check_scene(scenes.material_zoo(), "zoo", [(64, 64, 16), (128, 128, 32)])
check_scene(scenes.dragon_cornell(100000, "glass+metal"), "dragon", [(128, 72, 8), (480, 270, 16)])

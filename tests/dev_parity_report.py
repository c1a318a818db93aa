"""Bit-identity / RMSE of the device images against the reference goldens (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gnxraytracer_amd as gx, scenes
from conftest import GOLDEN, golden
gx.init(0)
g = golden("render.npz"); gv = golden("render_vol.npz")
def report(name, img, ref, st, rays):
    d = img[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64)
    same = (img[..., :3].view(np.uint32) == ref[..., :3].view(np.uint32)).mean()
    print(f"{name:16s} bit-identical {same*100:7.3f} %  rmse {np.sqrt((d**2).mean()):.3e}  maxabs {np.abs(d).max():.3e}  rays {st['rays_closest']}/{st['rays_any']} ref {tuple(int(v) for v in rays)}")
for name in ["cornell", "zoo", "mesh2k", "cornell_env", "cornell_uniform"]:
    W, H, spp, depth = (int(v) for v in g[name + "_cfg"])
    if name in ("cornell", "cornell_uniform"): b = scenes.cornell()
    elif name == "zoo": b = scenes.material_zoo()
    elif name == "mesh2k": b = scenes.dragon_cornell(2000, "glass+metal", mesh_path=os.path.join(GOLDEN, "mesh_2k.3d"))
    else:
        b = scenes.cornell(sky=True); b.AddInfLight(os.path.join(GOLDEN, "env_100x50.hdr"))
    integ = gx.PathIntegrator(depth, 1.0, "uniform" if name == "cornell_uniform" else "spatial")
    img, st = integ.Render(gx.Scene(b), W, H, spp)
    report(name, img, g[name], st, g[name + "_rays"])
for name in ["vol_synth", "vol_cfg5"]:
    W, H, spp, depth = (int(v) for v in gv[name + "_cfg"])
    b = scenes.volume_cornell(sigma_a=(0.5,) * 3, sigma_s=(3.5,) * 3, g_grid=0.3) if name == "vol_synth" else scenes.volume_cornell_cfg5(0.05)
    img, st = gx.VolPathIntegrator(depth, 1.0, "spatial").Render(gx.Scene(b), W, H, spp)
    report(name, img, gv[name], st, gv[name + "_rays"])
gc = golden("cfg2_recorded.npz")
img, st = gx.PathIntegrator(8, 1.0, "spatial").Render(gx.Scene(scenes.cornell()), 256, 256, 64)
print("cfg2 full: rays", st["rays_closest"], st["rays_any"], "(reference 16058662 / 12329468) checksum %.6f (reference 78538.576918)" % float(img[..., :3].astype(np.float64).sum()),
      "thumb bit-identical %.3f %%" % (100 * (img[::4, ::4, :3].view(np.uint32) == gc["thumb"].view(np.uint32)).mean()))

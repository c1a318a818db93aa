"""One-off larger parity checks against the oracle (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gnxraytracer_amd as gx, oracle_lib as ol, scenes
from conftest import GOLDEN
gx.init(0)
def check(name, b, integ, W, H, spp, **kw):
    t0 = time.time(); img, st = integ.Render(gx.Scene(b), W, H, spp, **kw); t1 = time.time()
    oimg, ost = ol.OracleScene(b).render(integ, W, H, spp, **kw); t2 = time.time()
    same = (img[..., :3].view(np.uint32) == oimg[..., :3].view(np.uint32))
    d = img[..., :3].astype(np.float64) - oimg[..., :3].astype(np.float64)
    print(f"{name:28s} identical {same.mean()*100:8.4f}%  rmse {np.sqrt((d**2).mean()):.2e}  rays {st['rays_closest']}/{st['rays_any']} vs {ost['rays_closest']}/{ost['rays_any']}  gpu {t1-t0:.1f}s oracle {t2-t1:.1f}s", flush=True)
env = os.path.join(GOLDEN, "env_100x50.hdr")
check("cfg3 dragon 480x270x16", scenes.dragon_cornell(100000, "glass+metal"), gx.PathIntegrator(8, 1.0, "spatial"), 480, 270, 1024, spp_begin=0, spp_end=16)
check("cfg4 zoo+env 480x270x16", scenes.dragon_cornell(100000, "zoo", env=env), gx.PathIntegrator(8, 1.0, "spatial"), 480, 270, 1024, spp_begin=0, spp_end=16)
check("cfg4 zoo+env power 240x135x8", scenes.dragon_cornell(100000, "zoo", env=env), gx.PathIntegrator(8, 1.0, "power"), 240, 135, 1024, spp_begin=100, spp_end=108)
check("cfg5 full sigma 128x128x16", scenes.volume_cornell_cfg5(1.0), gx.VolPathIntegrator(8, 1.0, "spatial"), 128, 128, 256, spp_begin=0, spp_end=16)
check("cfg5 full sigma depth 20", scenes.volume_cornell_cfg5(1.0), gx.VolPathIntegrator(20, 1.0, "spatial"), 96, 96, 256, spp_begin=30, spp_end=38)
check("cfg1 sphere whitted 256x256x16", scenes.cornell_sphere("glass"), gx.WhittedIntegrator(5), 256, 256, 16)
check("cfg2 deep path depth 30", scenes.material_zoo(), gx.PathIntegrator(30, 1.0, "spatial"), 128, 128, 64, spp_begin=0, spp_end=16)
if len(sys.argv) > 1 and sys.argv[1] == "full":
    check("cfg3 FULL 1920x1080x4 (spp 500..503)", scenes.dragon_cornell(100000, "glass+metal"), gx.PathIntegrator(8, 1.0, "spatial"), 1920, 1080, 1024, spp_begin=500, spp_end=504)
    check("cfg5 FULL 512x512x8", scenes.volume_cornell_cfg5(1.0), gx.VolPathIntegrator(8, 1.0, "spatial"), 512, 512, 256, spp_begin=100, spp_end=108)

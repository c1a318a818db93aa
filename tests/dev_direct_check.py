"""DirectLighting device-vs-oracle check (dev tool): python tests/dev_direct_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gnxraytracer_amd as gx, scenes, oracle_lib as ol
gx.init(0)
cases = [("cornell", scenes.cornell()), ("zoo", scenes.material_zoo()), ("mesh", scenes.dragon_cornell(2000, "glass+metal")),
         ("sphere", scenes.cornell_sphere("glass")), ("nolights", scenes.cornell_no_lights())]
bad = 0
for name, b in cases:
    for sname in ("all", "one"):
        for depth in (1, 3, 5):
            W, H, spp = 50, 38, 4
            integ = gx.DirectLightingIntegrator(sname, depth)
            oimg, ost = ol.OracleScene(b).render(integ, W, H, spp)
            for spass in (0, 1):
                img, st = integ.Render(gx.Scene(b), W, H, spp, samples_per_pass=spass)
                same = bool((img.view(np.uint32) == oimg.view(np.uint32)).all())
                rays = (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
                if not (same and rays): bad += 1
                print(name, sname, depth, spass, "biteq", same, "rays", (st["rays_closest"], st["rays_any"]), (ost["rays_closest"], ost["rays_any"]),
                      "maxabs", float(np.abs(img - oimg).max()), flush=True)
print("BAD", bad)

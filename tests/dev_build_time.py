"""Scene set-up time on the headline scene (dev tool): python tests/dev_build_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnxraytracer_amd as gx, scenes
gx.init(0)
for n in (100000, 1000000):
    b = scenes.dragon_cornell(n, "glass+metal")
    for rep in range(2):
        t = time.time(); s = gx.Scene(b); dt = time.time() - t
        print(f"{n} tris: gnxr_scene_create {dt*1e3:.1f} ms", flush=True)
        del s

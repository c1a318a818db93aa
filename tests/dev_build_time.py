"""Scene set-up time on the headline scene (dev tool): python tests/dev_build_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gnxraytracer_amd as gx, scenes
gx.init(0)
for n in (100000, 1000000):
    for method in ("sah", "hlbvh"):
        b = scenes.dragon_cornell(n, "glass+metal")
        b.set_bvh_split_method(method)
        for rep in range(2):
            t = time.time(); s = gx.Scene(b); dt = time.time() - t
            print(f"{n} tris, {method}: gnxr_scene_create {dt*1e3:.1f} ms, {s.bvh()[0].shape[0]} nodes", flush=True)
            if rep == 1 and n == 100000:   # what the tree costs at render time
                import torch
                integ = gx.PathIntegrator(8, 1.0, "spatial"); out = torch.zeros((1080, 1920, 4), device="cuda")
                for k in range(2): st = integ.RenderDevice(s, out.data_ptr(), 1920, 1080, 1024, spp_begin=32 * k, spp_end=32 * k + 32, samples_per_pass=32)
                print(f"   render 32 spp: {st['seconds_render']*1e3:.1f} ms, {(st['rays_closest'] + st['rays_any']) / st['seconds_render'] / 1e6:.0f} Mrays/s", flush=True)
            del s

"""Dev tool: render the headline scene with the mesh all glass / all metal / half and half (2 x 32 spp at 1080p), for a rocprofv3 run
that compares the lane utilisation and time of the glossy shade kernel between the three.   usage: python3 tests/dev_mat_split.py glass|metal|glass+metal"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gnxraytracer_amd as gx, scenes
gx.init(0)
b = scenes.dragon_cornell(100000, sys.argv[1])
scene = gx.Scene(b); integ = gx.PathIntegrator(8, 1.0, "spatial")
out = torch.zeros((1080, 1920, 4), device="cuda")
for rep in range(2):
    st = integ.RenderDevice(scene, out.data_ptr(), 1920, 1080, 1024, spp_begin=64 * rep, spp_end=64 * rep + 64, samples_per_pass=32)
print(sys.argv[1], "rays", st["rays_closest"] + st["rays_any"], "ms", round(st["seconds_render"] * 1e3, 2))

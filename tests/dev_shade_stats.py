"""k_shade wave-time per section (dev tool; needs a -DGX_SHADE_STATS build): GNXR_LIB=ab_libs/lib_sstats.so python tests/dev_shade_stats.py [cfg3|cfg4]"""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnxraytracer_amd as gx, scenes
gx.init(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
b = scenes.dragon_cornell(100000, "glass+metal") if wl == "cfg3" else scenes.dragon_cornell(100000, "zoo", env=scenes.synthetic_env_path(1000, 500))
scene = gx.Scene(b); integ = gx.PathIntegrator(8, 1.0, "spatial")
out = torch.zeros((1080, 1920, 4), device="cuda")
lib = C.CDLL(gx.LIB_PATH)
buf = (C.c_ulonglong * 16)()
integ.RenderDevice(scene, out.data_ptr(), 1920, 1080, 1024, spp_begin=0, spp_end=16, samples_per_pass=16)
lib.gnxr_debug_shade_stats(buf, 1)
integ.RenderDevice(scene, out.data_ptr(), 1920, 1080, 1024, spp_begin=16, spp_end=32, samples_per_pass=16)
lib.gnxr_debug_shade_stats(buf, 1)
v = list(buf); tot = sum(v) or 1
names = ["load+tri_test+surface_point", "Le", "light_select(+1 halton)", "4 halton", "light_sample", "bsdf f/pdf + shadow ray", "bsdf sample_f (MIS)", "light_pdf + MIS record", "NEE stores", "loop head", "continuation (2 halton, sample_f, RR, stores)"]
print(wl, json.dumps({n: round(v[i] / tot, 4) for i, n in enumerate(names)}, indent=1))

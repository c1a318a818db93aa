"""A/B timing of library builds on the headline workload (dev tool): python tests/dev_ab.py lib1.so lib2.so ..."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    import gnxraytracer_amd as gx, scenes
    gx.init(0)
    b = scenes.dragon_cornell(100000, "glass+metal")
    scene = gx.Scene(b); integ = gx.PathIntegrator(8, 1.0, "spatial")
    out = torch.zeros((1080, 1920, 4), device="cuda")
    gx.lib().gnxr_set_profiling(1)
    best = None
    for rep in range(4):
        st = integ.RenderDevice(scene, out.data_ptr(), 1920, 1080, 1024, spp_begin=8 * rep, spp_end=8 * rep + 8, samples_per_pass=8)
        if rep and (best is None or st["seconds_render"] < best["seconds_render"]): best = st
    rays = best["rays_closest"] + best["rays_any"]
    print(json.dumps({"lib": os.environ.get("GNXR_LIB", "default"), "ms": best["seconds_render"] * 1e3, "Mrays/s": rays / best["seconds_render"] / 1e6,
                      "trace_ms": best["seconds_closest"] * 1e3, "shade_ms": best["seconds_shade"] * 1e3, "combine_ms": best["seconds_nee"] * 1e3, "checksum": float(out.sum().item())}))
else:
    for lib in sys.argv[1:]:
        env = dict(os.environ, GNXR_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + lib + " " + r.stderr[-500:]), flush=True)

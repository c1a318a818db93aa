"""A/B timing of library builds / tuning knobs on the headline workload (dev tool).
usage: python tests/dev_ab.py [--spp N] [--passes P] [--workload cfg3|cfg4] variant ...   (each timed call renders P passes of N spp)      variant = name[:lib.so][:ENV=VAL,ENV=VAL]
Each variant runs in its own process (GNXR_LIB selects the build, the environment carries the knobs); prints one JSON line each."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    import gnxraytracer_amd as gx, scenes
    spp, workload, passes = int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    spp = int(os.environ.get("GNXR_AB_SPP", spp)); passes = int(os.environ.get("GNXR_AB_PASSES", passes))   # per-variant overrides
    gx.init(0)
    if workload == "tex":   # the image-textured Cornell box of the parity tests (smile-face plastic, tiled matte floor, mirror, glass sheet)
        from conftest import GOLDEN
        b = scenes.textured_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr"))
    else:
        b = scenes.dragon_cornell(100000, "glass+metal") if workload == "cfg3" else scenes.dragon_cornell(100000, "zoo", env=scenes.synthetic_env_path(1000, 500))
    scene = gx.Scene(b)
    which = os.environ.get("GNXR_AB_INTEG", "path")   # per-variant: the integrator under test
    integ = {"path": lambda: gx.PathIntegrator(8, 1.0, "spatial"), "whitted": lambda: gx.WhittedIntegrator(5), "direct": lambda: gx.DirectLightingIntegrator("all", 5),
             "direct_one": lambda: gx.DirectLightingIntegrator("one", 5), "volpath": lambda: gx.VolPathIntegrator(8, 1.0, "spatial")}[which]()
    out = torch.zeros((1080, 1920, 4), device="cuda")
    gx.lib().gnxr_set_profiling(1)
    best = None
    for rep in range(4):
        st = integ.RenderDevice(scene, out.data_ptr(), 1920, 1080, 1024, spp_begin=(spp * passes * rep) % 1024, spp_end=(spp * passes * rep) % 1024 + spp * passes, samples_per_pass=spp)
        if rep and (best is None or st["seconds_render"] < best["seconds_render"]): best = st
    rays = best["rays_closest"] + best["rays_any"]
    print(json.dumps({"variant": os.environ.get("GNXR_AB_NAME", "default"), "ms": round(best["seconds_render"] * 1e3, 3), "Mrays/s": round(rays / best["seconds_render"] / 1e6, 1),
                      "trace_ms": round(best["seconds_closest"] * 1e3, 3), "shade_ms": round(best["seconds_shade"] * 1e3, 3), "combine_ms": round(best["seconds_nee"] * 1e3, 3),
                      "checksum": float(out.double().sum().item()), "in_flight": best.get("passes_in_flight"), "iters": best.get("loop_iterations"),
                      "state_GB": round(best.get("state_bytes", 0) / 1e9, 1), "spp": spp, "passes": passes}))
else:
    args = sys.argv[1:]
    spp, workload, passes = 32, "cfg3", 1
    while args and args[0].startswith("--"):
        if args[0] == "--spp": spp = int(args[1])
        if args[0] == "--workload": workload = args[1]
        if args[0] == "--passes": passes = int(args[1])
        args = args[2:]
    for v in args:
        parts = v.split(":")
        env = dict(os.environ, GNXR_AB_NAME=parts[0])
        for p in parts[1:]:
            if "=" in p:
                for kv in p.split(","):
                    k, val = kv.split("="); env[k] = val
            elif p:
                env["GNXR_LIB"] = os.path.abspath(p)
        r = subprocess.run([sys.executable, __file__, "--child", str(spp), workload, str(passes)], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + v + " " + r.stderr[-800:]), flush=True)

"""Per-rank efficiency of the row-sharded render measured on one GPU (dev tool): rank 0's share of the full 1024-spp frame for N = 1, 2, 4, 8
ranks, with the library's own sub-pass choice and with explicit ones.  usage: python tests/dev_shard_eff.py [k:R ...]   (0:0 = auto)"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gnxraytracer_amd as gx, scenes
gx.init(0)
b = scenes.dragon_cornell(100000, "glass+metal")
scene = gx.Scene(b); integ = gx.PathIntegrator(8, 1.0, "spatial")
out = torch.zeros((1080, 1920, 4), device="cuda")
cfgs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(0, 0)]
base = None
for N in (1, 2, 4, 8):
    for k, R in cfgs:
        best = None
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            st = integ.RenderDevice(scene, out.data_ptr(), 1920, 1080, 1024, spp_begin=0, spp_end=1024, samples_per_pass=k, passes_in_flight=R, shard_index=0, shard_count=N, shard_rows=1)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if best is None or dt < best[0]: best = (dt, st)
        dt, st = best
        rays = st["rays_closest"] + st["rays_any"]
        if N == 1 and base is None: base = dt
        print(json.dumps({"N": N, "spp_per_pass": k, "in_flight_req": R, "in_flight": st["passes_in_flight"], "sub_passes": st["passes"], "iters": st["loop_iterations"], "state_GB": round(st["state_bytes"] / 1e9, 1),
                          "rank0_ms": round(dt * 1e3, 1), "rank0_Mrays/s": round(rays / dt / 1e6), "strong_scaling_eff_vs_N1": round(base / (N * dt), 3) if base else None}), flush=True)

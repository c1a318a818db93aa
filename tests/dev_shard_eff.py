"""Per-rank efficiency of the row-sharded render measured on one GPU (dev tool): rank 0's share for N = 1, 2, 4, 8."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gnxraytracer_amd as gx, scenes
gx.init(0)
b = scenes.dragon_cornell(100000, "glass+metal")
scene = gx.Scene(b); integ = gx.PathIntegrator(8, 1.0, "spatial")
out = torch.zeros((1080, 1920, 4), device="cuda")
sps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for N in (1, 2, 4, 8):
    best = None
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = integ.RenderDevice(scene, out.data_ptr(), 1920, 1080, 1024, spp_begin=sps * rep, spp_end=sps * rep + sps, samples_per_pass=sps, shard_index=0, shard_count=N, shard_rows=1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep and (best is None or dt < best[0]): best = (dt, st)
    dt, st = best
    rays = st["rays_closest"] + st["rays_any"]
    print(json.dumps({"N": N, "spp_per_pass": sps, "rank0_ms": dt * 1e3, "rank0_Mrays/s": rays / dt / 1e6, "ideal_aggregate_Mrays/s": N * rays / dt / 1e6}))

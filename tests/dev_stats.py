"""k_trace wave statistics (dev tool; needs a -DGX_TRACE_STATS build): GNXR_LIB=ab_libs/lib_stats.so python tests/dev_stats.py"""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gnxraytracer_amd as gx, scenes
gx.init(0)
b = scenes.dragon_cornell(100000, "glass+metal")
scene = gx.Scene(b); integ = gx.PathIntegrator(8, 1.0, "spatial")
out = torch.zeros((1080, 1920, 4), device="cuda")
lib = C.CDLL(gx.LIB_PATH)
buf = (C.c_ulonglong * 24)()
integ.RenderDevice(scene, out.data_ptr(), 1920, 1080, 1024, spp_begin=0, spp_end=64, samples_per_pass=32)
lib.gnxr_debug_trace_stats(buf, 1)
st = integ.RenderDevice(scene, out.data_ptr(), 1920, 1080, 1024, spp_begin=64, spp_end=128, samples_per_pass=32)
lib.gnxr_debug_trace_stats(buf, 1)
v = list(buf)
rays = st["rays_closest"] + st["rays_any"]
names = ["loop_trips", "A_wave_iters", "A_lane_iters", "B_wave_entries", "B_lanes", "B_wave_tri_iters(max leafN)", "B_lane_tris", "refill_events", "refill_lanes", "live_lanes_sum", "B_lanes_retest",
         "ticks_refill", "ticks_A", "ticks_B", "unused14", "ticks_retire"]
d = {n: v[i] for i, n in enumerate(names)}
d["rays"] = rays
d["A_util"] = v[2] / max(1, v[1]) / 64
d["B_util_lanes"] = v[4] / max(1, v[3]) / 64
d["B_util_tris"] = v[6] / max(1, v[5]) / 64
d["live_frac"] = v[9] / max(1, v[0]) / 64
d["A_wave_iters_per_ray"] = v[1] * 64 / rays
d["A_lane_iters_per_ray"] = v[2] / rays
d["B_wave_tri_iters_per_ray"] = v[5] * 64 / rays
d["B_lane_tris_per_ray"] = v[6] / rays
d["refill_events_per_ray_x64"] = v[7] * 64 / rays
d["A_live_frac"] = v[20] / max(1, v[1]) / 64; d["A_holding_leaf_frac"] = v[21] / max(1, v[1]) / 64; d["A_finished_frac"] = v[22] / max(1, v[1]) / 64; d["A_speculating_frac"] = v[23] / max(1, v[1]) / 64
d["A_visits_below"] = {"16": v[16] / max(1, v[2]), "64": v[17] / max(1, v[2]), "256": v[18] / max(1, v[2]), "1024": v[19] / max(1, v[2])}
tt = sum(v[i] for i in (11, 12, 13, 15)) or 1
d["time_share"] = {"refill": v[11] / tt, "A": v[12] / tt, "B": v[13] / tt, "retire": v[15] / tt}
d["ticks_per_A_iter"] = v[12] / max(1, v[1]); d["ticks_per_B_entry"] = v[13] / max(1, v[3]); d["ticks_per_refill_event"] = v[11] / max(1, v[7])
print(json.dumps(d, indent=1))

"""Turn a `rocprofv3 --kernel-trace --pmc SQ_...` pass into profiles/pmc_*.json (dev tool; bench.py reads profiles/pmc_latest.json).
usage: python tests/dev_pmc_json.py <pmc_dir> <out.json> --workload cfg3 --spp-per-step 128 --width 1920 --height 1080 [--traffic traffic.json]
Per kernel (template arguments dropped, dispatches averaged):
  valu_insts_per_launch = SQ_INSTS_VALU / dispatches                     (wave64 VALU instructions issued)
  lane_util             = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)   (share of the 64 lanes that are on, VALU instructions)
  valu_busy             = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * SQ_BUSY_CYCLES / 32 shader engines): the gfx94x VALUBusy formula
                          (ROCm 7.2 has no gfx950 derived-counter section, MI355X_MICROARCH.md); it charges 4 cycles per instruction
  wait_frac             = SQ_WAIT_ANY / SQ_WAVE_CYCLES                    (share of wave-cycles parked in s_waitcnt / barrier)
`--traffic` stamps a dev_traffic.py file with the same `_measured_on` block."""
import argparse, collections, csv, glob, json, subprocess, sys, os

ap = argparse.ArgumentParser()
ap.add_argument("pmc_dir"); ap.add_argument("out")
ap.add_argument("--workload", default="cfg3"); ap.add_argument("--spp-per-step", type=int, default=128)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--commit", default=None); ap.add_argument("--traffic", default=None)
ap.add_argument("--steps-profiled", type=int, default=3, help="steps the profiled command rendered (warm-up + timed): totals / this = per-step figures, which do not depend on how the library merges launches")
a = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = a.commit
if commit is None:
    try: commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        try: commit = open(os.path.join(ROOT, ".build_commit")).read().strip()
        except Exception: commit = None
sys.path.insert(0, ROOT)
try:
    import bench
    source_id = bench.kernel_source_id()
except Exception:
    source_id = None
on = {"source_id": source_id, "steps_profiled": a.steps_profiled, "workload": a.workload, "spp_per_step": a.spp_per_step, "width": a.width, "height": a.height, "commit": commit,
      "command": f"rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --workload {a.workload} --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"}

def short(name):
    return name.split("(")[0].replace("void ", "").replace("gnxr::", "").split("<")[0].strip()

files = glob.glob(a.pmc_dir + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set); variants = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"]); variants[k].add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
out = {"_measured_on": on}
for k, c in agg.items():
    n = len(disp[k]); e = {"dispatches": n, "variants": sorted(variants[k]), "raw_totals": {kk: vv for kk, vv in c.items()}}
    if "SQ_INSTS_VALU" in c: e["valu_insts_per_launch"] = c["SQ_INSTS_VALU"] / n; e["valu_insts_per_step"] = c["SQ_INSTS_VALU"] / a.steps_profiled
    if c.get("SQ_ACTIVE_INST_VALU"): e["lane_util"] = c.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * c["SQ_ACTIVE_INST_VALU"])
    if c.get("SQ_BUSY_CYCLES") and "SQ_ACTIVE_INST_VALU" in c: e["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * c["SQ_BUSY_CYCLES"] / 32)
    if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c: e["wait_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_BUSY_CYCLES") and "SQ_WAVE_CYCLES" in c: e["waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / (1024 * c["SQ_BUSY_CYCLES"] / 32)
    if c.get("SQ_INSTS_VALU") and "SQ_INSTS_SALU" in c: e["salu_per_valu"] = c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"]
    out[k] = e
json.dump(out, open(a.out, "w"), indent=1)
if a.traffic and os.path.exists(a.traffic):
    t = json.load(open(a.traffic))
    for k_, v_ in t.items():
        if isinstance(v_, dict) and "hbm_bytes_per_launch" in v_:
            v_["hbm_bytes_per_step"] = v_["hbm_bytes_per_launch"] * v_["dispatches"] / a.steps_profiled
            v_["hbm_bytes_per_step_uncorrected"] = v_["hbm_bytes_per_launch_uncorrected"] * v_["dispatches"] / a.steps_profiled
    t["_measured_on"] = dict(on, command=on["command"].replace("<counters>", "FETCH_SIZE | WRITE_SIZE (separate passes)")); json.dump(t, open(a.traffic, "w"), indent=1)
for k in ("k_trace", "k_shade", "k_vol_media", "k_vol_step"):
    if k in out: print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in out[k].items() if kk not in ("raw_totals", "variants")})

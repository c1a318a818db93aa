#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (+ wall-clock to 1024 spp) on the "dragon" Cornell scene.

Workloads (BASELINE.json `configs`):
  cfg3 (default, the headline metric)  Cornell box + ~100k-triangle mesh, Glass + Metal, PathIntegrator maxDepth 8,
        rrThreshold 1, "spatial" light sampling, HaltonSampler(1024), 1920x1080.
  cfg4  the same geometry with Metal / Plastic / Disney / Glass quarters + InfiniteAreaLight over a 1000x500 lat-long map
        (a deterministic synthetic map of MonValley1000.hdr's size: the reference's HDR cannot travel to the GPU box).
  cfg5  VolPathIntegrator, GridDensityMedium (the reference's density grid) + HomogeneousMedium, 512x512 @256 spp.
The mesh is the seeded SYNTHETIC stand-in for the reference's dragon.3d, which is absent from the snapshot
(.MISSING_LARGE_BLOBS) -- numbers are not comparable with anyone else's "dragon".

A step = one pass of the hot path over one batch: `--spp-per-step` (default 128) consecutive Halton samples of every pixel.
The default --steps 8 renders the full 1024 spp image, so `wall_to_1024spp_s` is measured, not extrapolated.  The steps of a
contiguous sample range are submitted as one library call; the library cuts the range into sub-passes (its own choice: ~64 M
paths each, four alive at once in regions of the state arrays, staggered so that a launch mixes the first bounces of one with the
thin late bounces of the others) and drives them with a device-side loop -- queue counts never come back to the host.
After the default (cfg3) run, two-step runs of cfg4 and cfg5 are appended under "also".

--gpus N > 1: one rank per GPU.  When no launcher has set WORLD_SIZE, bench.py starts its own ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process, before anything touches the GPU),
passes the child's output through and exits with its code; under a launcher it checks that the world size is N.  Image rows
are interleaved over the ranks, no collective runs during rendering, and the final FrameBuffer is gathered to rank 0 with one
RCCL gather inside the timed region.  value = rays traced by all ranks / max-over-ranks time.

Ray = one Scene::Intersect or Scene::IntersectP query (closest-hit, shadow and MIS rays), the unit the reference was
profiled in.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, MI355X_MICROARCH.md
# FP32 vector peak of the guide: 157.3 TFLOP/s = 256 CUs x 4 SIMDs x one wave64 FMA every 2 cycles at 2.4 GHz = 1228.8 G wave64 instructions/s
VALU_SPEC_GWIPS = 1228.8
VALU_SPEC_SIMDS = 1024
WORKLOADS = ("cfg3", "cfg4", "cfg5")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=128,
                    help="samples of every pixel rendered by one step; how a call's samples are cut into sub-passes is the library's business (--spp-per-pass)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024, help="HaltonSampler samplesPerPixel")
    ap.add_argument("--tris", type=int, default=100000)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--save-image", type=str, default="")
    ap.add_argument("--fuse-steps", type=int, default=0, help="steps a rank submits as one pass (default: the number of ranks)")
    ap.add_argument("--workload", choices=WORKLOADS, default="cfg3")
    ap.add_argument("--spp-per-pass", type=int, default=0, help="samples per pixel of one sub-pass of the library's path loop (0: the library's choice)")
    ap.add_argument("--passes-in-flight", type=int, default=0, help="sub-passes the path loop keeps alive at once (0: the library's choice)")
    ap.add_argument("--no-also", action="store_true", help="default cfg3 run only: skip the short cfg4 / cfg5 runs appended under \"also\"")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.workload == "cfg5":   # BASELINE.json configs[4]; explicit flags still win
        d = ap.parse_args([])
        if args.width == d.width and args.height == d.height: args.width, args.height = 512, 512
        if args.spp == d.spp: args.spp = 256
        # one pass holds all 256 samples of every pixel (67 M paths, 24 GB): VolPath has one ray in flight per path, so a pass runs ~60
        # rounds whose tail is thin; the thicker the rounds the better (64 / 128 / 256 spp per pass: 0.425 / 0.364 / 0.336 s)
        if args.spp_per_step == d.spp_per_step: args.spp_per_step = 256
        if args.steps == d.steps: args.steps = max(1, args.spp // args.spp_per_step)
    return args


# ------------------------------------------------------------------------------------------------ rank launch
def launch_plan(args, env):
    """What this process has to do about --gpus (pure function of the arguments and the environment; tests/test_host_logic.py):
       ("run", world)         render as one of `world` ranks (world == 1: no process group)
       ("spawn", n)           no launcher around us: start n ranks as a child process and relay its result
       ("error", message)     the launcher's world size contradicts --gpus"""
    ws = env.get("WORLD_SIZE")
    if ws is None:
        return ("run", 1) if args.gpus == 1 else ("spawn", args.gpus)
    try:
        world = int(ws)
    except ValueError:
        return ("error", f"WORLD_SIZE={ws!r} is not a number")
    if world != args.gpus:
        return ("error", f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    return ("run", world)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Start the ranks as a CHILD (never exec: this may run under a profiler that has initialised the GPU), relay stdout / stderr."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def call_ranges(i0, nsteps, sps, spp):
    """(s0, s1, steps) of the library calls that render steps i0 .. i0+nsteps-1 -- step i covers samples [i*sps, (i+1)*sps) of every pixel,
    modulo the sampler's range `spp`: one call per contiguous stretch of the Halton sequence (pure function; tests/test_host_logic.py)"""
    out = []
    i = i0
    while i < i0 + nsteps:
        s0 = (i * sps) % spp
        m = min(i0 + nsteps - i, max(1, (spp - s0) // sps))   # steps up to the end of the sample range
        out.append((s0, min(s0 + m * sps, spp), m))
        i += m
    return out


# ------------------------------------------------------------------------------------------------ CPU baseline
def host_cores():
    """CPU cores this job can actually use: the affinity mask, capped by the cgroup's CPU quota when there is one (a 1-GPU box hands a
    job a share of the host: its mask lists every core of the machine, its cpu.max says how many it may run on at once).  Nothing is
    hard-coded; all three figures go into the JSON."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]       # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:                                                             # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            quota = None
    cores = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return cores, {"affinity_mask": aff, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count()}


def cpu_baseline(builder, args):
    """CPU baseline on the box's host cores, on a bounded sample of the same workload.
    kind "reference": oracle/_ref/gnx_ref, the reference's own translation units (BVHAccel, Triangle, BSDFs, lights, samplers,
    media -- compiled from /root/reference in the development container, the binary travels with the snapshot) under the
    restated Render / Li loop (integrators/*.cpp and core/Integrator.cpp need Qt and cannot be built).
    kind "port": the oracle (plain CPU restatement), used when the reference binary is not there."""
    import struct

    import numpy as np

    import gnxraytracer_amd as gx
    import oracle_lib as ol

    vol = args.workload == "cfg5"
    integ = (gx.VolPathIntegrator if vol else gx.PathIntegrator)(args.max_depth, 1.0, "spatial")
    cores, core_info = host_cores()
    # same scene / camera / sampler type on a bounded sample: cfg 3/4: 1/4 of the pixels, cfg 5: all pixels; one sample per pixel is
    # timed first and the number of samples is then chosen so that each of the two baselines is about 12 s of CPU work
    w, h = (args.width, args.height) if vol else (960, 540)
    osc = ol.OracleScene(builder)
    osc.render(integ, 64, 36, args.spp, threads=cores, spp_begin=0, spp_end=1)   # touch the tables once
    t0 = time.perf_counter()
    osc.render(integ, w, h, args.spp, threads=cores, spp_begin=0, spp_end=1)
    t1 = max(1e-3, time.perf_counter() - t0)
    spp = int(max(2, min(args.spp, 12.0 / t1)))
    print(f"[bench] cpu baseline: {cores} threads ({core_info}), 1 spp took {t1:.2f} s -> timing {spp} spp at {w}x{h}", file=sys.stderr, flush=True)
    img, st = osc.render(integ, w, h, args.spp, threads=cores, spp_begin=0, spp_end=spp)
    rays = st["rays_closest"] + st["rays_any"]
    port = {"value": rays / st["seconds_render"] / 1e6, "unit": "Mrays/s", "cores": cores, "cores_detail": core_info, "kind": "port",
            "sample": f"{w}x{h} px, samples 0..{spp - 1} of HaltonSampler({args.spp}), same scene; {rays} rays in {st['seconds_render']:.1f} s; "
                      f"oracle = CPU restatement of the reference path, OpenMP over pixel columns (core/Integrator.cpp:256) on the {cores} cores this job may use, no printf"}
    def one_thread(run, label):
        """the same renderer on ONE thread, on a quarter of the pixels and as many samples as ~6 s buy (BASELINE.md asks for both figures)"""
        w1, h1 = max(16, w // 2), max(16, h // 2)
        t0_ = time.perf_counter()
        run(w1, h1, 1, 1)
        t1_ = max(1e-3, time.perf_counter() - t0_)
        spp1 = int(max(1, min(args.spp, 6.0 / t1_)))
        rays1, secs1 = run(w1, h1, spp1, 1)
        return {"value": rays1 / secs1 / 1e6, "unit": "Mrays/s", "cores": 1, "kind": label,
                "sample": f"{w1}x{h1} px, {spp1} spp, same scene; {rays1} rays in {secs1:.1f} s on one thread"}

    def run_port(w_, h_, spp_, threads_):
        _, st_ = osc.render(integ, w_, h_, args.spp, threads=threads_, spp_begin=0, spp_end=spp_)
        return st_["rays_closest"] + st_["rays_any"], st_["seconds_render"]

    if not os.path.exists(ol.REF_BIN):
        port["one_thread"] = one_thread(run_port, "port")
        return port
    try:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            sp = os.path.join(td, "scene.bin")
            ol.write_scene_file(builder, sp)

            def run_ref(w_, h_, spp_, threads_):
                raw_ = ol.run_ref(sp, "render", None, [w_, h_, spp_, args.max_depth, 1.0, 0, threads_, 1 if vol else 0])
                cnt_ = np.frombuffer(raw_[w_ * h_ * 16:w_ * h_ * 16 + 16], np.uint64)
                return int(cnt_[0]) + int(cnt_[1]), struct.unpack("<d", raw_[w_ * h_ * 16 + 16:w_ * h_ * 16 + 24])[0]

            print(f"[bench] cpu baseline: compiled reference classes, {spp} spp", file=sys.stderr, flush=True)
            rrays, secs = run_ref(w, h, spp, cores)
            single = one_thread(run_ref, "reference")
        return {"value": rrays / secs / 1e6, "unit": "Mrays/s", "cores": cores, "cores_detail": core_info, "kind": "reference",
                "sample": f"{w}x{h} px, {spp} spp (HaltonSampler({spp})), same scene; {rrays} rays in {secs:.1f} s; the reference's own classes "
                          f"(compiled from its sources) under the restated Render/Li loop, OpenMP over pixel columns on the {cores} cores this job may use, no printf",
                "one_thread": single, "port": {"value": port["value"], "sample": port["sample"]}}
    except Exception as e:   # the binary is optional: fall back to the port
        port["reference_error"] = str(e)[-120:]
        if vol:
            port["reference_note"] = ("at the volume file's own sigma_t = 100 the tracking loops pass Halton dimension 1000, where the reference "
                                      "indexes PrimeSums[] out of bounds (undefined behaviour; the compiled reference crashes here), so the "
                                      "oracle -- which wraps the dimension like the device -- is the CPU baseline for cfg 5")
        port["one_thread"] = one_thread(run_port, "port")
        return port


# ------------------------------------------------------------------------------------------------ roofline
def git_head():
    """HEAD of the tree this runs from; on the GPU box (.git does not travel) the stamp a post-commit hook leaves in .build_commit."""
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:
        try:
            return open(os.path.join(ROOT, ".build_commit")).read().strip() or None
        except Exception:
            return None


def kernel_source_id():
    """Identifies the KERNELS a counter file was measured on (__graft_entry__.source_id: a hash over csrc/ and the compile flags)."""
    import __graft_entry__ as ge
    return ge.source_id()


def library_matches_source():
    """False when the libgnxr.so in the tree was not built from the csrc/ in the tree (None: no id file beside the library)."""
    import __graft_entry__ as ge
    lid = ge.library_source_id()
    return None if lid is None else lid == ge.source_id()


def counters_stale(tj):
    """True when a profiles/ counter file was measured on other kernels than the ones in this tree (no source id in the file: compare commits)."""
    on = tj.get("_measured_on", {})
    if on.get("source_id"):
        return on["source_id"] != kernel_source_id()
    return on.get("commit") != git_head()


def load_profile_json(name, args, sps, world):
    """profiles/<name>: counter figures per launch, valid only for the launch size they were measured on (`_measured_on`)."""
    path = os.path.join(ROOT, "profiles", name.replace(".json", f"_{args.workload}.json"))   # one file per workload
    if not os.path.exists(path):
        return None
    try:
        tj = json.load(open(path))
    except Exception:
        return None
    on = tj.get("_measured_on", {})
    if (on.get("workload"), on.get("spp_per_step"), on.get("width"), on.get("height")) != (args.workload, sps, args.width, args.height) or world != 1:
        return None
    return tj


def roofline(kernel, secs, launches, units, unit_name, bytes_per_unit, byte_terms, args, sps, world, peaks, gathers_per_unit, extra, steps=None):
    """Three ceilings for the dominant kernel, each a fraction of a stated peak; `bound` names the one the kernel sits closest to.
    hbm   : bytes that actually reached HBM per launch (PMC: FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes, profiles/) over
            the HIP-event launch time vs 8 TB/s.  The ALGORITHMIC byte rate (SURVEY 8d's model, counted on the walk that is timed)
            stands beside it: nodes and triangles are served by LDS / L2 / Infinity Cache, so that rate is a cache-level figure and may
            exceed the HBM peak -- it is reported as `algorithmic_GBps`, never as a fraction of HBM.
    valu  : VALU wave-instructions per unit (SQ_INSTS_VALU of a kept --pmc pass) x units / time vs the issue rate a saturating
            v_fma_f32 loop reaches on this device (gnxr_probe_valu_peak, measured in this run).
    gather: per-lane vector-memory loads per unit (counted: 8 per node visit that reads memory, 3 per triangle, 2 per leaf re-test,
            3 per ray for its record) x units / time vs the rate at which the device takes per-lane 16-byte gathers
            (gnxr_probe_gather_peak, measured in this run: ~1.1 lanes per clock per CU, hit or miss, at any occupancy)."""
    avg_s = secs / launches
    upl = units / launches
    alg = upl * bytes_per_unit / avg_s / 1e9
    one = unit_name[:-1]
    r = {"kernel": kernel, "launches": launches, "avg_launch_ms": avg_s * 1e3, unit_name + "_per_launch": upl,
         "bytes_per_" + one: bytes_per_unit, "byte_model": byte_terms, "algorithmic_bytes_per_launch": upl * bytes_per_unit}
    r.update(extra)
    hbm = {"achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "algorithmic_GBps": alg}
    traffic = None
    tj = load_profile_json("traffic_latest.json", args, sps, world)
    if tj and kernel in tj:
        # counter totals are kept per STEP (the library merges the launches of consecutive passes, so "per launch" depends on how many
        # steps a call covers); per launch of THIS run = per step x steps / launches
        if steps and tj[kernel].get("hbm_bytes_per_step"):
            traffic = tj[kernel]["hbm_bytes_per_step"] * steps / launches
        else:
            traffic = tj[kernel].get("hbm_bytes_per_launch")
        meas = traffic / avg_s / 1e9
        raw_step = tj[kernel].get("hbm_bytes_per_step_uncorrected")
        raw = (raw_step * steps / launches) if (steps and raw_step) else None
        hbm.update({"achieved": meas, "frac": meas / HBM_PEAK_GBS, "algorithmic_over_measured": upl * bytes_per_unit / traffic,
                    # the two readings of the counters side by side: FETCH_SIZE as counted + WRITE_SIZE, and with the guide's x2 on the fetch side
                    # (documented for wide streaming reads; node gathers are not that, so the x2 figure is an upper bound)
                    "bytes_per_launch_raw": raw, "bytes_per_launch_fetch_x2": traffic,
                    "achieved_raw": (raw / avg_s / 1e9) if raw else None, "frac_raw": (raw / avg_s / 1e9 / HBM_PEAK_GBS) if raw else None,
                    "stale_counters": counters_stale(tj),
                    "traffic_source": {"file": f"profiles/traffic_latest_{args.workload}.json", "commit": tj.get("_measured_on", {}).get("commit"),
                                       "raw_bytes_per_step": tj[kernel].get("hbm_bytes_per_step_uncorrected"), "steps_profiled": tj.get("_measured_on", {}).get("steps_profiled"),
                                       "correction": "FETCH_SIZE x 2 (gfx950, MI355X_MICROARCH.md) + WRITE_SIZE"}})
    valu = None
    pj = load_profile_json("pmc_latest.json", args, sps, world)
    if pj and kernel in pj and peaks.get("valu"):
        k = pj[kernel]
        vpl = k["valu_insts_per_step"] * steps / launches if (steps and k.get("valu_insts_per_step")) else k["valu_insts_per_launch"]
        ach = vpl / avg_s / 1e9
        lane = k.get("lane_util")
        valu = {"achieved": ach, "peak": peaks["valu"], "unit": "G wave-instr/s", "frac": ach / peaks["valu"],
                # against the guide's figure instead of the in-run probe (which moves with the clock the chip settles at under load)
                "peak_spec": VALU_SPEC_GWIPS, "frac_of_spec": ach / VALU_SPEC_GWIPS,
                "probe_clock_GHz": peaks["valu"] * 2.0 / VALU_SPEC_SIMDS,   # a saturating v_fma_f32 loop issues one wave64 instruction per SIMD every 2 cycles
                "useful_lane_frac": (ach / VALU_SPEC_GWIPS * lane) if lane else None,   # issue slots x lanes that do work, of the spec peak
                "stale_counters": counters_stale(pj),
                "wave_insts_per_" + one: vpl / upl, "lane_util": k.get("lane_util"), "valu_busy_pmc": k.get("valu_busy"),
                "wait_frac": k.get("wait_frac"), "waves_per_simd": k.get("waves_per_simd"),
                "source": {"file": f"profiles/pmc_latest_{args.workload}.json", "commit": pj.get("_measured_on", {}).get("commit")},
                "peak_source": "gnxr_probe_valu_peak in this run: independent v_fma_f32 chains, 8 waves per SIMD on every CU"}
    gather = None
    if gathers_per_unit and peaks.get("gather"):
        ach = upl * gathers_per_unit / avg_s / 1e9
        gather = {"achieved": ach, "peak": peaks["gather"], "unit": "G lane-loads/s", "frac": ach / peaks["gather"], "lane_loads_per_" + one: gathers_per_unit,
                  "peak_source": "gnxr_probe_gather_peak in this run: 8 dwordx4 of a random 128-byte record per lane, 8 MB table, 5 blocks per CU"}
    cands = [(c["frac"], n, c) for n, c in (("valu", valu), ("gather", gather), ("hbm", hbm)) if c and c["frac"] is not None]
    if cands:
        _, name, pick = max(cands, key=lambda t: t[0])
        r.update({"bound": name, "achieved": pick["achieved"], "peak": pick["peak"], "unit": pick["unit"], "frac": pick["frac"]})
    else:   # no counter file for this launch size and no probe: nothing measured to price against
        r.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None})
    r.update({"traffic": traffic, "hbm": hbm, "valu": valu, "gather": gather})
    return r


def also_runs():
    """Short runs of cfg 4 and cfg 5 as children of the default run: value, wall, and the roofline of their dominant kernel."""
    out = {}
    for wl, extra in (("cfg4", ["--steps", "2", "--warmup", "1"]), ("cfg5", ["--warmup", "1"])):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--no-cpu-baseline", "--no-also"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
            j = json.loads(line)
            rf = j.get("roofline") or {}
            out[wl] = {"value": j["value"], "unit": j["unit"], "steps": j["steps"], "ms_per_step": j["ms_per_step"], "wall_to_full_spp_s": j["wall_to_full_spp_s"],
                       "workload": j["config"]["workload"], "passes_in_flight": j["config"].get("passes_in_flight"), "path_state_GB": j["config"].get("path_state_GB"),
                       "rays_per_camera_sample": j["rays"]["per_camera_sample"],
                       "roofline": {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "traffic", "kernel_seconds")},
                       "roofline_valu": {k: (rf.get("valu") or {}).get(k) for k in ("frac", "frac_of_spec", "lane_util", "useful_lane_frac", "valu_busy_pmc", "stale_counters")},
                       "roofline_hbm": {k: (rf.get("hbm") or {}).get(k) for k in ("frac", "frac_raw", "achieved", "achieved_raw", "stale_counters")},
                       "child_wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:
            out[wl] = {"error": str(e)[-200:]}
    return out


# ------------------------------------------------------------------------------------------------ the benchmark
def main():
    args = parse()
    plan = launch_plan(args, os.environ)
    if plan[0] == "error":
        print("bench.py: " + plan[1], file=sys.stderr)
        sys.exit(2)
    if plan[0] == "spawn":   # before torch / HIP are touched
        sys.exit(spawn_ranks(plan[1], sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = plan[1]
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GNXR_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share the visible devices and
    # the collectives run on host copies.  The measured configuration is always nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("GNXR_BENCH_BACKEND", "nccl")
    n_visible = torch.cuda.device_count()
    if world > 1 and backend == "nccl" and n_visible < world:
        if rank == 0:
            print(f"bench.py: --gpus {world} over RCCL needs {world} visible devices, found {n_visible} "
                  "(GNXR_BENCH_BACKEND=gloo rehearses the rank logic on fewer)", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, n_visible)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    import gnxraytracer_amd as gx
    import scenes

    gx.init(dev_index)
    W, H = args.width, args.height
    # every rank builds the same scene (replicated: ~12 MB of tables, + 22 MB of env-map tables for cfg 4); rank 0 writes the
    # synthetic input files once
    mesh_path = os.path.join(ROOT, "gpurun_out", "_meshes", f"synthetic_dragon_{args.tris}_1.3d")
    env_path = None
    if args.workload == "cfg4":
        env_path = os.path.join(ROOT, "gpurun_out", "_meshes", "synthetic_env_1000x500.hdr")
    if rank == 0 and args.workload in ("cfg3", "cfg4"):
        scenes.synthetic_mesh_path(args.tris)
        if env_path:
            scenes.synthetic_env_path(1000, 500)
    if world > 1:
        dist.barrier()
    if args.workload == "cfg5":
        builder = scenes.volume_cornell_cfg5(1.0)
        integ = gx.VolPathIntegrator(args.max_depth, 1.0, "spatial")
    elif args.workload == "cfg4":
        builder = scenes.dragon_cornell(args.tris, "zoo", env=env_path, mesh_path=mesh_path)
        integ = gx.PathIntegrator(args.max_depth, 1.0, "spatial")
    else:
        builder = scenes.dragon_cornell(args.tris, "glass+metal", mesh_path=mesh_path)
        integ = gx.PathIntegrator(args.max_depth, 1.0, "spatial")
    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    t_setup = time.perf_counter()
    scene = gx.Scene(builder)   # gnxr_scene_create: host BVH build (the reference's SAH splits), 4-wide collapse, tables, upload
    scene_setup_s = time.perf_counter() - t_setup
    shard = dict(shard_index=rank, shard_count=world, shard_rows=1)
    out = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    acc = torch.zeros_like(out)
    stream = torch.cuda.current_stream().cuda_stream
    sps = args.spp_per_step

    # With N ranks a rank owns 1/N of the rows.  The library sizes its sub-passes in PATHS (~64 M), so a rank's sub-pass simply covers N
    # times as many samples per pixel and its launches stay as thick as a single GPU's -- nothing to special-case here (cfg 5, one
    # pass per call, still fuses N steps).  Exactly `steps` x `spp_per_step` samples of every pixel are rendered inside the timed region.
    fuse = args.fuse_steps if args.fuse_steps > 0 else world
    # How a call's samples are cut into sub-passes is the library's business (gnxr_render_params.samples_per_pass = 0: sub-passes of ~16 M
    # paths, four in flight; csrc/api.hip); --spp-per-pass / --passes-in-flight override it.  VolPath (cfg 5) renders a call as one pass.
    def pass_args(s0_, s1_):
        if args.workload == "cfg5":
            return dict(samples_per_pass=min(fuse * sps, s1_ - s0_))
        return dict(samples_per_pass=args.spp_per_pass, passes_in_flight=args.passes_in_flight)

    def calls_of(i0, nsteps):
        return call_ranges(i0, nsteps, sps, args.spp)

    # path state for the largest call of the run is allocated before the warm-up (gnxr_render_reserve), so that no timed step grows it
    for s0_, s1_, _ in list(calls_of(0, args.warmup)) + list(calls_of(0, args.steps)):
        integ.Reserve(scene, W, H, args.spp, spp_begin=s0_, spp_end=s1_, **pass_args(s0_, s1_), **shard)

    def run_steps(i0, nsteps):
        """steps i0 .. i0+nsteps-1 = samples [i0*sps, (i0+nsteps)*sps) of every pixel (mod the Halton range --spp), submitted as ONE library
        call per contiguous sample range (csrc/api.hip, the device-driven path loop).  Returns the summed stats."""
        agg = {}
        for s0, s1, m in calls_of(i0, nsteps):
            st = integ.RenderDevice(scene, out.data_ptr(), W, H, args.spp, stream=stream, spp_begin=s0, spp_end=s1, **pass_args(s0, s1), **shard)
            acc.add_(out)
            for k_, v_ in st.items():
                if isinstance(v_, (int, float)):
                    agg[k_] = max(agg.get(k_, 0), v_) if k_ in ("passes_in_flight", "state_bytes") else agg.get(k_, 0) + v_
        return agg

    note(f"scene ready in {scene_setup_s:.2f} s; {args.warmup} warm-up + {args.steps} timed steps of {sps} spp")
    if args.warmup > 0:
        run_steps(0, args.warmup)
    acc.zero_()
    if not args.no_kernel_timing:
        gx.lib().gnxr_set_profiling(1)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tot = dict(rays_closest=0, rays_any=0, seconds_closest=0.0, seconds_nee=0.0, seconds_shade=0.0, launches_closest=0,
               launches_nee=0, rays_closest_nee=0, camera_samples=0, kernel_launches=0, media_segments=0, passes=0, loop_iterations=0,
               passes_in_flight=0, state_bytes=0)
    sync()
    t0 = time.perf_counter()
    st = run_steps(0, args.steps)
    for k in tot:
        tot[k] += st[k]
    # final FrameBuffer gather: each rank owns rows y with y % world == rank (one RCCL gather, timed)
    if world > 1:
        from gnxraytracer_amd.distributed import gather_framebuffer
        full = gather_framebuffer(acc.to(coll_dev), rank, world, 1, dst=0)
        if rank == 0:
            acc = full.to(dev)
    sync()
    dt = time.perf_counter() - t0
    gx.lib().gnxr_set_profiling(0)

    note(f"timed region {dt:.3f} s")
    rays = tot["rays_closest"] + tot["rays_any"]
    tvec = torch.tensor([dt, float(rays), float(tot["rays_closest"]), float(tot["rays_any"])], dtype=torch.float64, device=coll_dev)
    per_rank = None
    if world > 1:
        tmax = tvec.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tvec.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, rays_all = tmax[0].item(), tsum[1].item()
        # load balance of the row shards: every rank's own wall time and ray count
        every = [torch.zeros_like(tvec) for _ in range(world)]
        dist.all_gather(every, tvec)
        per_rank = [{"rank": i, "wall_s": round(v[0].item(), 4), "rays": int(v[1].item())} for i, v in enumerate(every)]
    else:
        dt_max, rays_all = dt, float(rays)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    spp_timed = args.steps * sps                 # samples per pixel traced inside the timed region (wraps around the Halton range beyond --spp)
    spp_done = min(spp_timed, args.spp)
    names = {"cfg3": "dragon-stand-in Cornell 1920x1080", "cfg4": "dragon-stand-in Cornell + environment light 1920x1080 (cfg 4)", "cfg5": "volume Cornell 512x512 (cfg 5)"}
    detail = "Mrays/s (path tracing, closest-hit + shadow + MIS rays), " + names[args.workload]
    metric = detail
    if args.workload == "cfg3":   # the headline metric under BASELINE.json's own name: `value` is its Mrays/s half, `wall_to_1024spp_s` the other
        try:
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            metric = "Mrays/sec + wall-clock to 1024spp, dragon Cornell 1920×1080"
    wl = {"cfg5": f"cfg5: Cornell + GridDensityMedium (reference density grid 100x100x40, sigma_a 10 sigma_s 90) + HomogeneousMedium, "
                  f"VolPathIntegrator maxDepth {args.max_depth} rr 1 spatial, Halton({args.spp}), {W}x{H}",
          "cfg4": f"cfg4: Cornell + synthetic {args.tris}-tri mesh (stand-in for absent dragon.3d) in Glass / Metal / Plastic / Disney quarters + "
                  f"InfiniteAreaLight over a synthetic 1000x500 lat-long map (stand-in for MonValley1000.hdr, which cannot travel), "
                  f"PathIntegrator maxDepth {args.max_depth} rr 1 spatial, Halton({args.spp}), {W}x{H}",
          "cfg3": f"cfg3: Cornell + synthetic {args.tris}-tri mesh (stand-in for absent dragon.3d), Glass+Metal, "
                  f"PathIntegrator maxDepth {args.max_depth} rr 1 spatial, Halton({args.spp}), {W}x{H}"}[args.workload]
    result = {
        "metric": metric,
        "metric_detail": detail,
        "value": rays_all / dt_max / 1e6,
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl, "spp_per_step": sps,
                   # reported by the library (gnxr_stats), not assumed here
                   "passes_in_flight": tot["passes_in_flight"], "sub_passes": tot["passes"], "loop_iterations": tot["loop_iterations"],
                   "path_state_GB": round(tot["state_bytes"] / 1e9, 2),
                   "submission": "one library call per contiguous sample range; the library cuts it into sub-passes and keeps several in flight, queue counts stay on the device",
                   "spp_rendered": spp_done, "spp_timed": spp_timed,
                   "sharding": f"rows y % {world} == rank" if world > 1 else "none",
                   "gather": "RCCL gather of row shards to rank 0 (in timed region)" if world > 1 else "n/a",
                   "world_size": dist.get_world_size() if world > 1 else 1, "backend": (backend if world > 1 else "none"),
                   "devices_visible": n_visible, "commit": git_head(), "kernel_source_id": kernel_source_id(),
                   # False: the libgnxr.so that ran was not built from the csrc/ in this tree (a stale or experiment library)
                   "library_matches_source": library_matches_source(), "per_rank": per_rank},
        "wall_to_1024spp_s": dt_max * (1024.0 / spp_timed),
        "wall_to_full_spp_s": dt_max * (float(args.spp) / spp_timed),
        "wall_measured_s": dt_max,
        # the metric excludes scene build / upload like the reference's timeConsume (core/Integrator.cpp:228,317);
        # the end-to-end figure is reported beside it
        "scene_setup_s": scene_setup_s,
        "wall_end_to_end_s": scene_setup_s + dt_max * (float(args.spp) / spp_timed),
        "rays": {"closest": tot["rays_closest"], "any": tot["rays_any"], "per_camera_sample": rays / max(1, tot["camera_samples"])},
    }

    # ---- roofline of the dominant kernel (rank 0's launches, HIP events on the render stream)
    if not args.no_kernel_timing and tot["launches_closest"] > 0:
        note("roofline: VALU issue and gather-rate probes + counting pass")
        peaks = {"valu": gx.probe_valu_peak(), "gather": gx.probe_gather_peak()}   # what this device reaches on each, now
        # untimed counting pass of one sample per pixel: the WIDE (4-wide, speculative) walk that the timed kernel performs,
        # and the tracking-loop steps of k_vol_media
        gx.lib().gnxr_set_profiling(4)
        stc = integ.RenderDevice(scene, out.data_ptr(), W, H, args.spp, stream=stream, spp_begin=0, spp_end=1, samples_per_pass=1, **shard)
        gx.lib().gnxr_set_profiling(0)
        torch.cuda.synchronize()
        nr = stc["rays_closest"] + stc["rays_any"]
        n4, n_tris = stc["nodes_visited"] / nr, stc["tris_tested"] / nr
        kernel_seconds = {"k_trace4": tot["seconds_closest"], ("k_vol_media" if args.workload == "cfg5" else "k_nee_combine"): tot["seconds_nee"],
                          ("k_vol_step+compaction" if args.workload == "cfg5" else "k_shade+compaction"): tot["seconds_shade"]}
        if args.workload == "cfg5" and tot["launches_nee"] > 0 and tot["seconds_nee"] >= tot["seconds_closest"]:
            # k_vol_media dominates cfg 5: unit = one medium segment (Medium::Sample / Medium::Tr of the ray in flight).
            # bytes: 32 B ray + 16 B vs record + 16 B result per segment, 8 density loads of 4 B per tracking step
            # (two Halton draws per step read the permutation table: 2 B x digits, L2-resident, not charged)
            steps = stc["media_steps"] / max(1, stc["media_segments"])
            b_seg = 64.0 + 32.0 * steps
            # per-lane loads: 8 density values per step + the permutation-table digits of its two Halton values (~2 x 5) + 4 per segment
            result["roofline"] = roofline("k_vol_media", tot["seconds_nee"], tot["launches_nee"], tot["media_segments"], "segments", b_seg,
                                          "64 B per segment (ray, state, result) + 32 B per tracking step (8 density loads)", args, sps, world, peaks,
                                          4.0 + 18.0 * steps, {"tracking_steps_per_segment": steps, "kernel_seconds": kernel_seconds}, steps=args.steps)
        else:
            n_re, n_mem = stc["leaf_retests"] / nr, stc["nodes_from_memory"] / nr
            b_ray = 136.0 + 128.0 * n4 + 48.0 * n_tris + 32.0 * n_re   # SURVEY 8(d) with the node term of the tree that is walked: 128-B DNode4
            result["roofline"] = roofline("k_trace4", tot["seconds_closest"], tot["launches_closest"], rays, "rays", b_ray,
                                          "136 B records + 128 B x 4-wide nodes visited (speculative visits included) + 48 B x triangles tested + "
                                          "32 B x leaf boxes re-tested, counted by k_trace4<COUNT> on the timed walk", args, sps, world, peaks,
                                          8.0 * n_mem + 3.0 * n_tris + 2.0 * n_re + 3.0,
                                          {"nodes4_per_ray": n4, "nodes4_from_memory_per_ray": n_mem, "tris_per_ray": n_tris, "leaf_retests_per_ray": n_re,
                                           "kernel_seconds": kernel_seconds}, steps=args.steps)
        result["roofline"]["note"] = ("nodes and triangles (11 MB) are served by LDS (the top 64 nodes: half of all visits), L2 and Infinity Cache, so the "
                                      "algorithmic byte model over-states what reaches HBM; the walk is a dependent chain per ray (load a node, test four "
                                      "boxes, pick the next) and sits below all three ceilings -- DESIGN.md section 4 has the measurements that rule each one out")
    if args.save_image and rank == 0:
        np.save(args.save_image, acc.cpu().numpy())
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(builder, args)
        result["cpu_baseline"]["reference_in_survey_container"] = "cfg 2 only: 1.58 Mrays/s as-is, 6.3 Mrays/s printf-free, 8-core Xeon 2.1 GHz (BASELINE.md)"
    if world == 1 and args.workload == "cfg3" and not args.no_also and not args.no_kernel_timing:
        # BASELINE.json's other two GPU configurations, two steps each, so that their numbers are timed by whoever times this run
        # (child processes: this one first gives its 15 GB of path state back)
        del scene, out, acc
        torch.cuda.empty_cache()
        result["also"] = also_runs()
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Randomised device-vs-oracle sweep (dev tool): scenes x integrators x odd image sizes x depths x rr thresholds x light
strategies x sample ranges x shards x sub-pass sizes x sub-passes in flight.  Every case must match the oracle bit for bit (images and ray counts)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gnxraytracer_amd as gx, oracle_lib as ol, scenes
from conftest import GOLDEN
def env():
    b = scenes.cornell(sky=True); b.AddInfLight(os.path.join(GOLDEN, "env_100x50.hdr")); return b
SCENES = {
    "cornell": scenes.cornell, "zoo": scenes.material_zoo, "env": env,
    "mesh2k": lambda: scenes.dragon_cornell(2000, "glass+metal", mesh_path=os.path.join(GOLDEN, "mesh_2k.3d")),
    "mesh2k_zoo_env": lambda: scenes.dragon_cornell(2000, "zoo", env=os.path.join(GOLDEN, "env_100x50.hdr"), mesh_path=os.path.join(GOLDEN, "mesh_2k.3d")),
    "sphere_glass": lambda: scenes.cornell_sphere("glass"), "sphere_mirror": lambda: scenes.cornell_sphere("mirror"),
    "vol_synth": lambda: scenes.volume_cornell(sigma_a=(0.5,) * 3, sigma_s=(3.5,) * 3, g_grid=0.3),
    "vol_cfg5_thin": lambda: scenes.volume_cornell_cfg5(0.05), "vol_sphere": lambda: scenes.cornell_sphere("medium"),
    "textured": lambda: scenes.textured_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr")),
    "vol_textured": lambda: scenes.textured_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr")),
    "smooth": lambda: scenes.smooth_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr")),
    "vol_smooth": lambda: scenes.smooth_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr")),
    "delta": scenes.delta_cornell, "vol_delta": scenes.delta_cornell,
    "textured_uv": lambda: scenes.textured_cornell(os.path.join(GOLDEN, "tex_smile_96x80.hdr"), uv_quads=True),
    "no_lights": scenes.cornell_no_lights, "vol_fog": scenes.cornell_in_fog, "vol_no_lights": scenes.cornell_no_lights,
}
def run_sweep(seed=1, ncase=40, verbose=True):
    """returns the list of mismatching case descriptions"""
    rng = np.random.default_rng(seed)
    bad = []
    names = list(SCENES)
    for c in range(ncase):
        name = names[c % len(names)]
        b = SCENES[name]()
        vol = name.startswith("vol")
        kind = "volpath" if vol else str(rng.choice(["path", "path", "whitted", "direct"]))
        depth = int(rng.integers(1, 11)); rr = float(rng.choice([0.25, 1.0, 4.0])); strat = str(rng.choice(["spatial", "uniform", "power"]))
        W, H = int(rng.integers(17, 90)), int(rng.integers(17, 90)); spp = int(rng.choice([4, 8, 16, 64]))
        s0 = int(rng.integers(0, spp)); s1 = int(rng.integers(s0 + 1, spp + 1))
        shards = int(rng.choice([1, 1, 2, 3])); sr = int(rng.choice([1, 2, 5])); si = int(rng.integers(0, shards))
        spp_pass = int(rng.choice([0, 1, 3])); in_flight = int(rng.choice([0, 1, 2, 3, 8]))
        if kind == "whitted": integ = gx.WhittedIntegrator(min(depth, 6))
        elif kind == "direct": integ = gx.DirectLightingIntegrator(str(rng.choice(["all", "one"])), min(depth, 6)); kind = "direct-" + {0: "all", 1: "one"}[integ.directStrategy]
        elif kind == "volpath": integ = gx.VolPathIntegrator(depth, rr, strat)
        else: integ = gx.PathIntegrator(depth, rr, strat)
        kw = dict(spp_begin=s0, spp_end=s1, shard_index=si, shard_count=shards, shard_rows=sr)
        t0 = time.time()
        img, st = integ.Render(gx.Scene(b), W, H, spp, samples_per_pass=spp_pass, passes_in_flight=in_flight, **kw)
        oimg, ost = ol.OracleScene(b).render(integ, W, H, spp, **kw)
        same = img[..., :3].view(np.uint32) == oimg[..., :3].view(np.uint32)
        ok = bool(same.all()) and (st["rays_closest"], st["rays_any"]) == (ost["rays_closest"], ost["rays_any"])
        desc = (f"{name:15s} {kind:10s} depth {depth:2d} rr {rr:4.2f} {strat:8s} {W}x{H} spp {spp} [{s0},{s1}) shard {si}/{shards}x{sr} pass {spp_pass}x{in_flight}  identical {same.mean()*100:.3f}% "
                f"rays {st['rays_closest']}/{st['rays_any']} vs {ost['rays_closest']}/{ost['rays_any']}")
        if not ok: bad.append(desc)
        if verbose: print(("ok  " if ok else "BAD ") + desc + f"  {time.time()-t0:.1f}s", flush=True)
    return bad


if __name__ == "__main__":
    gx.init(0)
    bad = run_sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    print("mismatching cases:", len(bad))

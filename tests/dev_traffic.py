"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/traffic_*.json (dev tool).
usage: python tests/dev_traffic.py <fetch_dir> <write_dir> <out.json>
gfx950: FETCH_SIZE (KB) under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM); k_trace's reads are 16-B
per-lane gathers, for which the counter is uncalibrated -- both the raw and the doubled figure are recorded."""
import csv, glob, json, sys, collections
def load(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gnxr::", "").split("<")[0]
        agg[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: (agg[k], len(n[k])) for k in agg}
fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fe:
    fkb, nd = fe[k]; wkb = wr.get(k, (0, nd))[0]
    out[k] = {"dispatches": nd, "fetch_kb_raw_per_launch": fkb / nd, "write_kb_per_launch": wkb / nd,
              "hbm_bytes_per_launch": (2 * fkb + wkb) * 1024 / nd, "hbm_bytes_per_launch_uncorrected": (fkb + wkb) * 1024 / nd,
              "note": "FETCH_SIZE x2 (gfx950 correction for wide reads) + WRITE_SIZE, KB -> bytes"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out.get("k_trace", {})))

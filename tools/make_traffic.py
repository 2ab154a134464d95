#!/usr/bin/env python3
"""HBM bytes per launch of the dominant kernel from the PMC passes of tools/prof_pmc.sh -> profiles/<tag>_<workload>_traffic.json
(what bench.py reports as roofline.traffic).  FETCH_SIZE / WRITE_SIZE are in KB (1024 B); FETCH_SIZE is doubled on gfx950 as
/opt/skills/guides/MI355X_MICROARCH.md prescribes (it reports half of wide coalesced reads).
Usage: make_traffic.py <pmc dir> <tag> <workload: genome|chr19> [kernel]"""
import csv, glob, json, os, sys
root, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
kernel = sys.argv[4] if len(sys.argv) > 4 else "k_tile_build"
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if r.get("Kernel_Name", "").startswith(kernel) and r["Counter_Name"] in vals:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
# bench.py also launches the kernel WITHOUT the text (cli_shaped_step: 4 B per base only): the roofline is stated for the launches
# of the timed step, i.e. those that write track + text -- the dispatches within 10 % of the largest WRITE_SIZE; FETCH_SIZE (the
# event buckets, the same for both kinds) is averaged over all
wmax = max(vals["WRITE_SIZE"]) if vals["WRITE_SIZE"] else 0.0
vals["WRITE_SIZE"] = [v for v in vals["WRITE_SIZE"] if v >= 0.9 * wmax]
f = sum(vals["FETCH_SIZE"]) / max(1, len(vals["FETCH_SIZE"]))
w = sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"]))
out = {"kernel": kernel, "workload": workload, "FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
       "launches_counted": [len(vals["FETCH_SIZE"]), len(vals["WRITE_SIZE"])],
       "hbm_bytes_per_launch": (2 * f + w) * 1024.0,
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py (tools/prof_pmc.sh); FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); KB = 1024 B",
       "source": "profiles/%s_pmc_summary.txt" % tag}
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "%s_%s_traffic.json" % (tag, workload)), "w"), indent=1)
print(json.dumps(out))

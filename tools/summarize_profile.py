#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh run (gpurun_out/<tag>/) into tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_pmc.json           per-launch FETCH_SIZE / WRITE_SIZE of the dominant kernel
  profiles/pmc_latest.json          {workload: {"hbm_bytes_per_launch": ...}} read by bench.py
HBM bytes follow MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of a wide (16 B/lane) coalesced read stream, so it is doubled; WRITE_SIZE is exact
(checked here against k_synth_fill, which writes a known byte count)."""
import csv, glob, json, os, shutil, sys

tag, workload = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = max(glob.glob(os.path.join(src, "kt", "*", "*kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
dom = max(rows, key=lambda r: float(r["TotalDurationNs"]))

def counter(kind, name_part, exact=None):
    f = max(glob.glob(os.path.join(src, kind, "*", "*counter_collection.csv")), key=os.path.getmtime)
    recs = list(csv.DictReader(open(f)))
    if exact is not None:        # the dominant instantiation only (the finish pass is another instantiation of the same template)
        vals = [float(r["Counter_Value"]) for r in recs if r["Kernel_Name"] == exact]
        if vals:
            return vals
    return [float(r["Counter_Value"]) for r in recs if name_part in r["Kernel_Name"]]

import re
dom_fn = re.search(r"vq::(k_\w+)", dom["Name"]).group(1)      # counters of the dominant kernel only
fetch = counter("pmc_fetch", "vq::" + dom_fn + "<", dom["Name"])
write = counter("pmc_write", "vq::" + dom_fn + "<", dom["Name"])
synth_w = counter("pmc_write", "k_synth_fill")
out = {
    "workload": workload,
    "dominant_kernel": dom["Name"], "calls": int(dom["Calls"]), "avg_ns": float(dom["AverageNs"]),
    "min_ns": float(dom["MinNs"]), "percentage": float(dom["Percentage"]),
    "FETCH_SIZE_KiB_per_launch": sum(fetch) / len(fetch), "WRITE_SIZE_KiB_per_launch": sum(write) / len(write),
    "fetch_correction": 2.0,
    "hbm_bytes_per_launch": (2.0 * sum(fetch) / len(fetch) + sum(write) / len(write)) * 1024.0,
    "calibration_k_synth_fill_WRITE_SIZE_KiB": synth_w,
}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
latest = os.path.join(dst, "pmc_latest.json")
allw = json.load(open(latest)) if os.path.exists(latest) else {}
allw[workload] = {"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "from": f"{tag}_pmc.json"}
json.dump(allw, open(latest, "w"), indent=1)
print(json.dumps(out, indent=1))

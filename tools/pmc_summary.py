"""Summarise rocprofv3 --pmc passes: per-launch FETCH_SIZE / WRITE_SIZE of the path kernel.

Counter semantics per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
in KiB (hbm_bytes = value * 1024); on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read
stream, so the corrected upper figure (x2) is printed beside the raw one.
"""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
res = {}
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out, f"prof_{name}", "**", "*counter_collection.csv"), recursive=True)
    vals = []
    for f in files:
        for row in csv.DictReader(open(f)):
            if "pt_render_kernel<false" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
    if vals:
        res[counter] = {"launches": len(vals), "mean_KiB_per_launch": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}
print(json.dumps(res, indent=1))
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    f = res["FETCH_SIZE"]["mean_KiB_per_launch"] * 1024
    w = res["WRITE_SIZE"]["mean_KiB_per_launch"] * 1024
    print(f"per launch: fetch raw {f/1e6:.2f} MB (x2 gfx950 correction {2*f/1e6:.2f} MB), write {w/1e6:.2f} MB, "
          f"total raw {(f+w)/1e6:.2f} MB, corrected {(2*f+w)/1e6:.2f} MB")
    json.dump({"hbm_bytes_per_launch": 2 * f + w, "fetch_raw_bytes": f, "write_bytes": w, "iterations_per_launch": 256,
               "note": "FETCH_SIZE*1024*2 (gfx950 correction) + WRITE_SIZE*1024, mean over pt_render_kernel launches"},
              open(os.path.join(out, "pmc_traffic.json"), "w"))

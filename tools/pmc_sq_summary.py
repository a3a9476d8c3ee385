import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "pt_render_kernel<false" in row.get("Kernel_Name", "") or "pt_render_kernelILb0" in row.get("Kernel_Name", ""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m): print(f"{k:28s} {m[k]:18.1f}  (n={len(acc[k])})")
g = m.get
if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
    print("lanes active per VALU instr (of 64): %.1f" % (g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") ))
if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
    print("VALU instrs per wave: %.0f" % (g("SQ_INSTS_VALU") / g("SQ_WAVES")))
if g("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM"):
        if g(k): print(f"{k}/SQ_WAVE_CYCLES = {g(k)/g('SQ_WAVE_CYCLES'):.3f}")
if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
    print("SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = %.3f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES")))

# machine-readable copy for bench.py (profiles/pmc_sq.json): instruction counts per launch of the path kernel
import json
if g("SQ_INSTS_VALU"):
    json.dump({"valu_insts_per_launch": g("SQ_INSTS_VALU"), "salu_insts_per_launch": g("SQ_INSTS_SALU"),
               "lds_insts_per_launch": g("SQ_INSTS_LDS"), "iterations_per_launch": 256,
               "active_lanes_per_valu_inst": (g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU")) if g("SQ_ACTIVE_INST_VALU") else None,
               "note": "rocprofv3 --pmc SQ_INSTS_* (own passes), mean over pt_render_kernel launches of bench.py"},
              open(os.path.join(out, "pmc_sq.json"), "w"))

#!/bin/bash
# HBM traffic of the path kernel on the 253k-triangle stand-in (config 5), own --pmc passes.  usage: bash tools/pmc_stress.sh <tag>
TAG=${1:-stress_pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- python tools/gpu_stress.py > $OUT/run_fetch.log 2> $OUT/prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- python tools/gpu_stress.py > $OUT/run_write.log 2> $OUT/prof_write.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/prof_l2 -o l2 -- python tools/gpu_stress.py > $OUT/run_l2.log 2> $OUT/prof_l2.err
python - <<PY
import csv, glob, collections
out = "$OUT"
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_render_kernel<false" in r["Kernel_Name"]:
            rows[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    vals = [sum(v) for v in d.values()]
    print(k, "launches", len(vals), "max per launch %.4g" % max(vals), "(the 4K 8-spp launch is the largest)")
PY
grep 3840 $OUT/run_fetch.log | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete

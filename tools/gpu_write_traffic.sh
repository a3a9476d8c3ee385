#!/bin/bash
# does the write traffic of the wide walk's suspend records cost time?  The product kernel against a build that
# writes every suspend record TWICE (var/libgpt_mirror.so: -DPT_WIDE_SUSP_MIRROR=1) and one whose drains never stop early, so no ray
# is ever suspended (var/libgpt_nostop.so: -DPT_WIDE_STOP_T=0 -DPT_WIDE_STOP_T_SMALL=0): launch time (HIP events, 3 launches) and
# WRITE_SIZE (own rocprofv3 pass) on the c5 and c4 stand-ins.   usage (GPU box): bash tools/gpu_write_traffic.sh <tag>
TAG=${1:-wt}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for w in c5 c4; do
  spp=8; [ $w = c4 ] && spp=32
  for v in product mirror nostop; do
    lib=gpu_pathtracer_amd/libgpt.so; [ $v != product ] && lib=var/libgpt_$v.so
    for rep in 1 2; do GPT_LIB_PATH=$PWD/$lib python tools/gpu_standin.py $w wide $spp 3 2>/dev/null | grep STANDIN | sed "s/^/$v $rep: /"; done
    GPT_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${w}_$v -o p -- python tools/gpu_standin.py $w wide $spp 2 > /dev/null 2>&1
    python - <<PY
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("$OUT/${w}_$v/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "pt_render_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
print("$v: $w WRITE_SIZE per launch %.2f GB (largest of %d launches)" % (max(v) * 1024 / 1e9, len(v)))
PY
  done
done 2>&1 | tee $OUT/write_traffic.log
find $OUT -name "*.csv" -delete

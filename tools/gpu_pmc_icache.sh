#!/bin/bash
# instruction-cache counters of one stand-in / of the headline kernel (own pass, --kernel-trace only).
# usage (GPU box): bash tools/gpu_pmc_icache.sh <tag> <c2|c3|c4|c5> <reference|wide|near> [iterations]
TAG=$1; WHICH=${2:-c5}; MODE=${3:-wide}; SPP=${4:-8}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ $WHICH = c2 ]; then CMD="python bench.py --no-cpu-baseline --no-counters --no-parity --no-square --no-other-configs --steps 8 --warmup 4"; else CMD="python tools/gpu_standin.py $WHICH $MODE $SPP 2"; fi
for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/ic_${WHICH}_${MODE}_$tag -o p -- $CMD > /dev/null 2> $OUT/ic_${WHICH}_${MODE}_$tag.err
done
python - <<PY | tee $OUT/ic_${WHICH}_${MODE}_summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/ic_${WHICH}_${MODE}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_render_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
big = {k: max(v) for k, v in acc.items()}
print("== instruction fetch, $WHICH $MODE (largest launch of the run)")
for k in sorted(big): print(f"   {k:28s} {big[k]:.5g}")
if "SQC_ICACHE_REQ" in big and big["SQC_ICACHE_REQ"]:
    print(f"   I-cache hit rate {big.get('SQC_ICACHE_HITS',0)/big['SQC_ICACHE_REQ']:.4f}, misses per request {big.get('SQC_ICACHE_MISSES',0)/big['SQC_ICACHE_REQ']:.4f}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete

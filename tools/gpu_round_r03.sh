#!/bin/bash
# One GPU-box session of round 3: smoke, both test suites, bench (with its in-run counters and other_configs), rocprofv3 kernel
# stats of the same command, the stand-ins on both trees and in both orders, counters of c3 / c5 in both orders, and the probe
# builds (var/libgpt_*.so: tools/build_variant.sh probe|loopprobe|asmcount|subprobe).
# Usage (from the repo root on the GPU box): bash tools/gpu_round_r03.sh <tag>
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^Load\|^Merge\|^Bvh\|^Scene'
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" > $OUT/smoke.log; tail -1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -av "$F" > $OUT/pytest_gpu.log; grep -a "passed\|failed" $OUT/pytest_gpu.log
timeout 1500 python -m pytest tests -m gpu -q --gpt-opt lds_scene=0 2>&1 | grep -av "$F" > $OUT/pytest_gpu_nolds.log; grep -a "passed\|failed" $OUT/pytest_gpu_nolds.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python bench.py --no-cpu-baseline --no-counters --no-parity --no-square --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -6 $f; done
for w in c3 c4 c5; do n=32; [ $w = c5 ] && n=8; for m in reference near wide; do for t in "" sbvh; do python tools/gpu_standin.py $w $m $n 3 $t 2>/dev/null | grep STANDIN; done; done; done > $OUT/configs.log; cat $OUT/configs.log
for w in c3 c5; do n=32; [ $w = c5 ] && n=8; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o stats -- python tools/gpu_standin.py $w wide $n 3 > /dev/null 2>&1; for f in $(find $OUT/prof_$w -name "*kernel_stats.csv"); do cp $f $OUT/${w}_wide_kernel_stats.csv; head -3 $f; done; done
for w in c3 c5; do for m in reference wide; do bash tools/gpu_pmc_standin.sh $TAG $w $m $( [ $w = c5 ] && echo 8 || echo 32 ) > /dev/null 2>&1; cat $OUT/${w}_${m}_pmc_summary.txt; done; done > $OUT/standin_pmc.txt; cat $OUT/standin_pmc.txt
if [ -f var/libgpt_probe.so ]; then
  for w in c3 c4 c5; do GPT_LIB_PATH=$PWD/var/libgpt_probe.so GPT_ALLOW_OLD_LIB=1 python tools/gpu_wide_probe.py $w 2>/dev/null | grep PROBE; done > $OUT/probes.txt
  for w in c2 c2sq c3 c4 c5; do GPT_LIB_PATH=$PWD/var/libgpt_loopprobe.so GPT_ALLOW_OLD_LIB=1 python tools/gpu_loop_probe.py $w 2>/dev/null | grep LOOP; done >> $OUT/probes.txt
  for m in reference wide; do for w in c3 c4 c5; do GPT_LIB_PATH=$PWD/var/libgpt_subprobe.so GPT_ALLOW_OLD_LIB=1 python tools/gpu_timesplit.py $w $m 2>/dev/null | grep "SPLIT\|SUB"; done; done >> $OUT/probes.txt
  cat $OUT/probes.txt
fi
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT

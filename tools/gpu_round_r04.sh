#!/bin/bash
# One GPU-box session of round 4: smoke, the GPU suite four ways (defaults, lds_scene=0, scheduler=1, scheduler=2), bench (in-run counters,
# other_configs legs default / reference / sbvh+default), rocprofv3 kernel stats of the same command, the three schedulers side by side on the
# stand-ins, counters of c3 / c5 (per-wave kernel, both orders) and of the decoupled scheduler's two trace stages on c5.
# Usage (from the repo root on the GPU box): bash tools/gpu_round_r04.sh <tag>
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^Load\|^Merge\|^Bvh\|^Scene'
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" > $OUT/smoke.log; tail -1 $OUT/smoke.log
for opt in "" "lds_scene=0" "scheduler=1" "scheduler=2"; do
  n=pytest_gpu${opt:+_}${opt/=/}
  timeout 1500 python -m pytest tests -m gpu -q ${opt:+--gpt-opt $opt} 2>&1 | grep -av "$F" > $OUT/$n.log; echo "$n: $(grep -a 'passed\|failed' $OUT/$n.log | tail -1)"
done
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python bench.py --no-cpu-baseline --no-counters --no-parity --no-square --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -6 $f; done
python tools/gpu_wavefront.py c3,c4,c5 reference,wide 2>&1 | grep "^WF" > $OUT/schedulers.log; cat $OUT/schedulers.log
for w in c3 c5; do n=32; [ $w = c5 ] && n=8; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o stats -- python tools/gpu_standin.py $w wide $n 3 > /dev/null 2>&1; for f in $(find $OUT/prof_$w -name "*kernel_stats.csv"); do cp $f $OUT/${w}_wide_kernel_stats.csv; head -3 $f; done; done
for w in c3 c5; do for m in reference wide; do bash tools/gpu_pmc_standin.sh $TAG $w $m $( [ $w = c5 ] && echo 8 || echo 32 ) > /dev/null 2>&1; cat $OUT/${w}_${m}_pmc_summary.txt; done; done > $OUT/standin_pmc.txt; cat $OUT/standin_pmc.txt
for s in 1 2; do GPT_WF_SCHEDULER=$s bash tools/gpu_pmc_wf.sh $TAG c5 wide 8 0 > /dev/null 2>&1; cp $OUT/wf_c5_wide_pmc_summary.txt $OUT/wf_c5_wide_scheduler${s}_pmc_summary.txt; cat $OUT/wf_c5_wide_scheduler${s}_pmc_summary.txt; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT

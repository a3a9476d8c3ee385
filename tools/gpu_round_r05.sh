#!/bin/bash
# One GPU-box session of round 5: smoke, the GPU suite twice (defaults, lds_scene=0), bench (in-run counters, other_configs, volpath leg),
# rocprofv3 kernel stats of the same command, counters of c3 / c5 in both orders.  The pytest RESULT lines are written to files that are
# copied into profiles/ (the tracked logs held the RCCL banner only).
# Usage (from the repo root on the GPU box): bash tools/gpu_round_r05.sh <tag> [quick]
TAG=${1:-r05}
QUICK=$2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^Load\|^Merge\|^Bvh\|^Scene'
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" > $OUT/smoke.log; tail -1 $OUT/smoke.log
for opt in "" "lds_scene=0"; do
  n=pytest_gpu${opt:+_}${opt/=/}
  timeout 1700 python -m pytest tests -m gpu -q ${opt:+--gpt-opt $opt} 2>&1 | grep -av "$F" > $OUT/$n.full.log
  (echo "# python -m pytest tests -m gpu -q ${opt:+--gpt-opt $opt}   (libgpt.so sha1 $(sha1sum gpu_pathtracer_amd/libgpt.so | cut -c1-16))"; grep -a 'passed\|failed\|error' $OUT/$n.full.log | tail -5; tail -12 $OUT/$n.full.log) > $OUT/$n.log
  echo "$n: $(grep -a 'passed\|failed' $OUT/$n.full.log | tail -1)"
  [ -n "$QUICK" ] && break
done
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python bench.py --no-cpu-baseline --no-counters --no-parity --no-square --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -6 $f; done
if [ -z "$QUICK" ]; then
for w in c3 c5; do for m in reference wide; do bash tools/gpu_pmc_standin.sh $TAG $w $m $( [ $w = c5 ] && echo 8 || echo 32 ) > /dev/null 2>&1; cat $OUT/${w}_${m}_pmc_summary.txt; done; done > $OUT/standin_pmc.txt; cat $OUT/standin_pmc.txt
fi
[ -z "$QUICK" ] && { bash tools/gpu_pmc_volpath.sh $TAG > $OUT/volpath_pmc.txt 2>&1; cat $OUT/volpath_pmc.txt | tail -12; }
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT

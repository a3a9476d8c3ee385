#!/bin/bash
# One GPU-box session of round 6: smoke, the GPU suite three times (defaults, every scene through the global-memory kernels, every Volpath
# scene through the one-ray kernel), bench.py (in-run counters, other_configs with CPU baselines, volpath leg with counters, 8-shard
# projection), rocprofv3 kernel stats of the same command.  Result lines are written to files that are copied into profiles/r06/.
# Usage (from the repo root on the GPU box): bash tools/gpu_round_r06.sh <tag> [quick]
TAG=${1:-r06}
QUICK=$2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^Load\|^Merge\|^Bvh\|^Scene'
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" > $OUT/smoke.log; tail -1 $OUT/smoke.log
for opt in "" "lds_scene=0" "vpt_walk_kernel=1"; do
  n=pytest_gpu${opt:+_}${opt/=/}
  sel=""; [ "$opt" = "vpt_walk_kernel=1" ] && sel="-k volpath"
  timeout 1700 python -m pytest tests -m gpu -q -s $sel ${opt:+--gpt-opt $opt} 2>&1 | grep -av "$F" > $OUT/$n.full.log
  (echo "# python -m pytest tests -m gpu -q $sel ${opt:+--gpt-opt $opt}   (libgpt.so sha1 $(sha1sum gpu_pathtracer_amd/libgpt.so | cut -c1-16))"; grep -a 'passed\|failed\|error' $OUT/$n.full.log | tail -5; grep -a "one rank:\|projection:\|  shard\|8 ranks on one GPU" $OUT/$n.full.log) > $OUT/$n.log
  echo "$n: $(grep -a 'passed\|failed' $OUT/$n.full.log | tail -1)"
  [ -n "$QUICK" ] && break
done
python bench.py --steps 20 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python bench.py --steps 20 --no-cpu-baseline --no-counters --no-parity --no-square --no-other-configs > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.err
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; head -6 $f; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT

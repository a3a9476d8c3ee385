#!/bin/bash
# One GPU-box session: tests, smoke, bench, rocprofv3 kernel stats + HBM counters.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag>
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 --timeout-method=thread > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
# per-kernel time of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o stats -- python bench.py --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof_stats.err
# HBM traffic counters, each in its own pass (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- python bench.py --no-cpu-baseline --steps 8 --warmup 4 > /dev/null 2> $OUT/prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- python bench.py --no-cpu-baseline --steps 8 --warmup 4 > /dev/null 2> $OUT/prof_write.err
find $OUT -name "*.csv" | head -20
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
# keep the merge small: drop the big traces, keep stats + summaries
find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT
# instruction counts of the path kernel (SQ counters, own passes) -> pmc_sq.json for bench.py's valu_issue figure
bash tools/pmc_sq.sh $TAG/sq > /dev/null 2>&1; cat $OUT/sq/sq_summary.txt 2>/dev/null | tail -8

#!/bin/bash
# the suite once more with every scene forced through the global-memory path, then the Volpath timings
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --gpt-opt lds_scene=0 --timeout 300 --timeout-method=thread > $OUT/pytest_gpu_nolds.log 2>&1; echo "pytest(no lds) rc=$?"; tail -2 $OUT/pytest_gpu_nolds.log
timeout 600 python tools/gpu_volpath.py 2>&1 | grep -v "^Bvh\|^Merge\|^Scene\|^Build" | tee $OUT/volpath.log
timeout 600 python tools/gpu_stress.py 2>&1 | grep -v "^Bvh\|^Merge\|^Scene\|^Build" | tee $OUT/stress.log
# per-kernel times of the Volpath run (which instantiation ran, how long)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_volpath -o volpath -- python tools/gpu_volpath.py > /dev/null 2> $OUT/prof_volpath.err
for f in $(find $OUT/prof_volpath -name "*kernel_stats.csv"); do head -8 $f; done
find $OUT -name "*kernel_trace.csv" -size +2M -delete

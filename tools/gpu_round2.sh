#!/bin/bash
# the suite once more with every scene forced through the global-memory path, then the Volpath timings
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
GPT_NO_LDS_SCENE=1 timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 --timeout-method=thread > $OUT/pytest_gpu_nolds.log 2>&1; echo "pytest(no lds) rc=$?"; tail -2 $OUT/pytest_gpu_nolds.log
timeout 600 python tools/gpu_volpath.py 2>&1 | grep -v "^Bvh\|^Merge\|^Scene\|^Build" | tee $OUT/volpath.log
timeout 600 python tools/gpu_stress.py 2>&1 | grep -v "^Bvh\|^Merge\|^Scene\|^Build" | tee $OUT/stress.log

#!/bin/bash
# quick GPU check of the decoupled scheduler: bash tools/gpu_r04_quick.sh <tag> [standins] [orders] [wf_paths list] [prof stand-in]
TAG=${1:-r04q}; WHICH=${2:-c5}; ORD=${3:-wide}; PATHS=${4:-1048576}; PROF=${5:-c5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^Load\|^Merge\|^Bvh\|^Scene'
timeout 120 python - > $OUT/first.log 2>&1 <<'PY'
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as ol
from gpu_pathtracer_amd import api
scene, meta = ol.load_cornell(8)
W, H, spp = 256, 128, 8
cam = ol.cornell_camera(meta, W, H)
ref, _ = ol.render(scene, cam, W, H, meta["epsilon"], 1, spp, kind="soft")
for order in ("reference", "wide"):
    with api.Renderer(scene.desc, W, H, meta["epsilon"]) as r:
        r.set_traversal_order(order)
        r.set_option("scheduler", 1)
        r.render(cam, 1, spp, reset=True)
        got = r.read_accum()
        print("FIRST", order, "scheduler 1 active", r.get_option("scheduler_active"), "paths", r.get_option("wf_paths"),
              "floats differing from the oracle (reference order):", int(np.count_nonzero(got.view(np.uint32) != ref.view(np.uint32))), "of", got.size, flush=True)
PY
grep "FIRST\|Error\|error" $OUT/first.log | tail -4
timeout 240 python tools/gpu_wavefront.py $WHICH $ORD 0 $PATHS 2>&1 | grep "^WF\|Error\|error" > $OUT/wavefront.log; cat $OUT/wavefront.log
if [ "$PROF" != "none" ]; then bash tools/gpu_wf_prof.sh $TAG $PROF wide $( [ $PROF = c5 ] && echo 8 || echo 32 ) ${PATHS%%,*} | grep -v "^WF"; fi

#!/bin/bash
# First GPU session of round 4: the decoupled scheduler against the per-wave kernel (films bit-equal?), the GPU suite through it, timings.
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F='^Load\|^Merge\|^Bvh\|^Scene'
timeout 300 python - > $OUT/first.log 2>&1 <<'PY'
import sys, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as ol
from gpu_pathtracer_amd import api
scene, meta = ol.load_cornell(8)
W, H, spp = 256, 128, 8
cam = ol.cornell_camera(meta, W, H)
ref, _ = ol.render(scene, cam, W, H, meta["epsilon"], 1, spp, kind="soft")
for order in ("reference", "wide"):
    for sched in (0, 1):
        with api.Renderer(scene.desc, W, H, meta["epsilon"]) as r:
            r.set_traversal_order(order)
            r.set_option("scheduler", sched)
            r.set_option("wf_paths", 4096)
            r.render(cam, 1, spp, reset=True)
            got = r.read_accum()
            print(order, "scheduler", sched, "active", r.get_option("scheduler_active"), "rounds", r.get_option("last_rounds"),
                  "floats differing from the oracle (reference order):", int(np.count_nonzero(got.view(np.uint32) != ref.view(np.uint32))), "of", got.size, flush=True)
PY
cat $OUT/first.log | grep -v "$F" | tail -8
timeout 1700 python -m pytest tests -m gpu -q -x --gpt-opt scheduler=1 2>&1 | grep -av "$F" > $OUT/pytest_gpu_wf.log; tail -15 $OUT/pytest_gpu_wf.log
timeout 900 python tools/gpu_wavefront.py c5,c3,c4 reference,wide 0 1048576,524288 2>&1 | grep "^WF\|Error\|error" > $OUT/wavefront.log; cat $OUT/wavefront.log

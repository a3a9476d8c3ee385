#!/bin/bash
# the shipped scene's shape (512 x 512, 17 bounces, density grid) through a list of var/libgpt_<name>.so builds of the one-ray Volpath kernel
# usage (GPU box): bash tools/gpu_walk_variants.sh name1 name2 ...   ("product" = gpu_pathtracer_amd/libgpt.so)
for v in "$@"; do
  lib=$PWD/var/libgpt_$v.so; [ $v = product ] && lib=
  GPT_LIB_PATH=$lib python - <<PY 2>/dev/null | grep WALKVAR
import sys, tempfile, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import standins
from gpu_pathtracer_amd import api
ls = api.LoadedScene(standins.write_smoke_scene(tempfile.mkdtemp()))
with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
    r.render(ls.camera, 1, 2, reset=True); r.synchronize()
    best = 1e9
    for _ in range(3):
        r.kernel_time_reset(); r.render(ls.camera, 1, 64, reset=True); r.synchronize()
        best = min(best, r.kernel_time()[1])
    print(f"WALKVAR $v: {best:.1f} ms, {ls.width * ls.height * 64 / best / 1e3:.1f} Msamples/s, film {hashlib.sha1(r.read_accum().tobytes()).hexdigest()[:12]}")
PY
done

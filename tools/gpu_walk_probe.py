"""Probe build of the one-ray Volpath kernel (bash tools/build_variant.sh walkprobe -DPT_WALK_PROBE=1): where a wave's time goes - stage
code / tracking / pool drain - and how many lanes take part.   python tools/gpu_walk_probe.py [var/libgpt_walkprobe.so]"""
import os, sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
if len(sys.argv) > 1: os.environ["GPT_LIB_PATH"] = os.path.abspath(sys.argv[1])
import numpy as np
import standins
from gpu_pathtracer_amd import api
ls = api.LoadedScene(standins.write_smoke_scene(tempfile.mkdtemp()))
spp = 8
with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
    r.render(ls.camera, 1, 2, reset=True); r.synchronize()
    r.kernel_time_reset(); r.render(ls.camera, 1, 64, reset=True); r.synchronize()
    print(f"WALKPROBE timing (counting off): {ls.width * ls.height * 64 / r.kernel_time()[1] / 1e3:.1f} Msamples/s")
    r.enable_counters(True)
    r.render(ls.camera, 1, spp, reset=True); r.synchronize()
    c = np.zeros(16, dtype=np.uint64)
    api.check(r.lib.gpt_read_probe_counters(r.ctx, c.ctypes.data))
    c = [int(x) for x in c]
n = c[5]
print(f"WALKPROBE samples {n}: rays/sample {(c[3] + c[4]) / n:.2f}, bounce iterations {c[2] / n:.2f}")
print(f"WALKPROBE stage passes per 64 samples {c[6] / n * 64:.1f}, ready lanes per pass {c[7] / max(1, c[6]):.1f}")
print(f"WALKPROBE tracking turns per 64 samples {c[8] / n * 64:.1f}, drains per 64 samples {c[9] / n * 64:.1f} ({(c[3] + c[4]) / max(1, c[9]):.1f} rays and {c[14] / max(1, c[9]):.0f} cycles each); "
      f"wave-steps per turn {c[10] / max(1, c[8]):.1f}, lanes per wave-step {c[11] / max(1, c[10]):.1f}, lane-steps per sample {c[11] / n:.0f}")
tot = c[14] + c[15]
print(f"WALKPROBE wave cycles: drain {c[14] / tot:.2f}, stage code {c[12] / tot:.2f}, tracking {c[13] / tot:.2f}, rest {(c[15] - c[12] - c[13]) / tot:.2f}; "
      f"cycles per tracking wave-step {c[13] / max(1, c[10]):.0f}, per stage pass {c[12] / max(1, c[6]):.0f}")

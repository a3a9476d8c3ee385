"""Time kernel variants (var/libgpt_*.so) on the headline workload and check that every one
produces the same film as the first (bit-exact).  Usage: python tools/bench_variants.py name1 name2 ..."""
import hashlib, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time, hashlib, json
sys.path.insert(0, %r)
import numpy as np
from gpu_pathtracer_amd import api, host
W, H, D = 1920, 1080, 8
scene, meta = host.load_baked(os.path.join(%r, "tests", "golden", "cornell_pt.npz"), D)
cam = host.camera_from_meta(meta, W, H)
r = api.Renderer(scene.desc, W, H, 0.001)
r.render(cam, 1, 8, reset=True); r.synchronize()
h = hashlib.sha1(r.read_accum().tobytes()).hexdigest()[:12]
best = 1e9
for rep in range(3):
    r.kernel_time_reset(); r.render(cam, 1, 64, reset=True); r.synchronize()
    n, ms = r.kernel_time(); best = min(best, ms)
print(json.dumps({"hash8spp": h, "ms64": best, "msamples": W*H*64/best/1e3}))
''' % (ROOT, ROOT)
ref = None
for name in sys.argv[1:]:
    env = dict(os.environ, GPT_LIB_PATH=os.path.join(ROOT, "var", f"libgpt_{name}.so"), GPT_ALLOW_OLD_LIB="1")
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
    try:
        d = json.loads(line)
        ref = ref or d["hash8spp"]
        print(f"{name:24s} {d['msamples']:9.1f} Msamples/s  {d['ms64']:8.2f} ms  {'SAME' if d['hash8spp']==ref else 'DIFFERENT!'}", flush=True)
    except Exception:
        print(name, "FAILED", line, flush=True)

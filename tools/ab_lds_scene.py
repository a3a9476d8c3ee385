import subprocess, sys, os, json
ROOT="/root/repo"
CHILD = r'''
import sys, os, hashlib, json
sys.path.insert(0, "%s")
from gpu_pathtracer_amd import api, host
W, H, D = 1920, 1080, 8
scene, meta = host.load_baked(os.path.join("%s", "tests", "golden", "cornell_pt.npz"), D)
cam = host.camera_from_meta(meta, W, H)
r = api.Renderer(scene.desc, W, H, 0.001)
r.set_option("lds_scene", int(sys.argv[1]))
r.render(cam, 1, 8, reset=True); r.synchronize()
h = hashlib.sha1(r.read_accum().tobytes()).hexdigest()[:12]
best = 1e9
for rep in range(3):
    r.kernel_time_reset(); r.render(cam, 1, 64, reset=True); r.synchronize()
    n, ms = r.kernel_time(); best = min(best, ms)
print(json.dumps({"hash": h, "ms": best, "ms_s": W*H*64/best/1e3}))
''' % (ROOT, ROOT)
for label, lds in (("lds_scene", "1"), ("global", "0"), ("lds_scene", "1"), ("global", "0")):
    o = subprocess.run([sys.executable, "-c", CHILD, lds], capture_output=True, text=True)
    print(label, o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-300:], flush=True)

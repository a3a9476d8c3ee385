"""Wall time of 1024 iterations at 1080p (full frame and one 1/8 shard) against iterations per launch (renderer option "max_batch")."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
from gpu_pathtracer_amd import api, host
W, H = 1920, 1080
scene, meta = host.load_baked(%r + "/tests/golden/cornell_pt.npz", 8)
cam = host.camera_from_meta(meta, W, H)
api.DEFAULT_OPTIONS["max_batch"] = int(sys.argv[1])
out = {}
for label, rank, n in (("full", 0, 1), ("shard", 3, 8)):
    r = api.Renderer(scene.desc, W, H, 0.001)
    r.set_tile_owner(rank, n)
    r.render(cam, 1, 64, reset=True); r.synchronize()
    t0 = time.perf_counter(); r.render(cam, 1, 1024, reset=True); r.synchronize(); dt = time.perf_counter() - t0
    out[label] = dt * 1e3
    r.close()
print(json.dumps(out))
''' % (ROOT, ROOT)
for b in sys.argv[1:]:
    o = subprocess.run([sys.executable, "-c", CHILD, b], capture_output=True, text=True)
    d = json.loads(o.stdout.strip().splitlines()[-1])
    print(f"batch {b:>4s}: full {d['full']:7.1f} ms ({1920*1080*1024/d['full']/1e3:7.1f} Msamples/s)  1/8 shard {d['shard']:6.1f} ms -> {d['full']/d['shard']:.2f}x", flush=True)

"""Chunked scheduling experiment: full frame (N=1) and one shard of an 8-way split, several chunk sizes."""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, hashlib, json
sys.path.insert(0, %r)
from gpu_pathtracer_amd import api, host
W, H, D = 1920, 1080, 8
scene, meta = host.load_baked(os.path.join(%r, "tests", "golden", "cornell_pt.npz"), D)
cam = host.camera_from_meta(meta, W, H)
if sys.argv[1] != "auto": api.DEFAULT_OPTIONS["chunk_iters"] = int(sys.argv[1])
out = {}
for label, rank, n in (("full", 0, 1), ("shard0of8", 0, 8), ("shard5of8", 5, 8)):
    r = api.Renderer(scene.desc, W, H, 0.001)
    r.set_tile_owner(rank, n)
    r.render(cam, 1, 64, reset=True); r.synchronize()
    h = hashlib.sha1(r.read_accum().tobytes()).hexdigest()[:10]
    best = 1e9
    for rep in range(3):
        r.kernel_time_reset(); r.render(cam, 1, 64, reset=True); r.synchronize()
        best = min(best, r.kernel_time()[1])
    out[label] = (round(best, 2), h)
    r.close()
print(json.dumps(out))
''' % (ROOT, ROOT)
ref = None
for c in sys.argv[1:]:
    o = subprocess.run([sys.executable, "-c", CHILD, c], capture_output=True, text=True)
    try:
        d = json.loads(o.stdout.strip().splitlines()[-1])
    except Exception:
        print(c, "FAILED", o.stderr[-400:]); continue
    ref = ref or {k: v[1] for k, v in d.items()}
    full = d["full"][0]
    print(f"chunk={c:>5s}: full {full:7.2f} ms ({1920*1080*64/full/1e3:7.1f} Ms/s) | shard0/8 {d['shard0of8'][0]:6.2f} ms  shard5/8 {d['shard5of8'][0]:6.2f} ms "
          f"-> 8-GPU kernel-time speedup {full/max(d['shard0of8'][0], d['shard5of8'][0]):.2f}x | " +
          ("SAME" if all(d[k][1] == ref[k] for k in d) else "DIFFERENT!"), flush=True)

"""First GPU bring-up: elementary-op parity, RNG parity, Cornell parity vs the soft oracle, quick timing."""
import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as ol
from gpu_pathtracer_amd import api, scene_types as st

soft = ol.load("soft")
rng = np.random.default_rng(7)
def obatch(fn, x, y=None):
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y if y is not None else x, np.float32)
    o = np.zeros_like(x); soft.oracle_math_batch(fn, st.ptr(x), st.ptr(y), st.ptr(o), len(x)); return o
N = 1 << 20
tests = {
 0: (rng.random(N)*8-0.2, None), 1: (rng.random(N)*8-0.2, None), 2: (rng.random(N)*8-0.2, None),
 3: (rng.standard_normal(N)*10, None), 4: (rng.random(N)*2-1, None),
 5: (rng.random(N)*20+1e-5, np.full(N, 1/2.2)),
 6: (rng.standard_normal(N)*np.exp(rng.standard_normal(N)*8), rng.standard_normal(N)*np.exp(rng.standard_normal(N)*8)),
 7: (np.abs(rng.standard_normal(N))*np.exp(rng.standard_normal(N)*20), None),
 8: (np.abs(rng.standard_normal(N))*np.exp(rng.standard_normal(N)*20), None),
}
names = ["sin","cos","tan","atan","acos","pow","div","sqrt","rsqrt"]
for fn,(x,y) in tests.items():
    x = x.astype(np.float32); y = None if y is None else np.asarray(y, np.float32)
    g = api.debug_math(fn, x, y); o = obatch(fn, x, y)
    bad = np.count_nonzero(g.view(np.uint32) != o.view(np.uint32))
    print(f"math {names[fn]:6s} mismatches {bad} / {N}", flush=True)
    if bad:
        i = np.nonzero(g.view(np.uint32) != o.view(np.uint32))[0][:5]
        print("   x", x[i], "gpu", g[i], "cpu", o[i])
import ctypes as C
for px, it in [(0,1),(1,2),(12345,2),(2073599,1024)]:
    s, u = api.debug_rng(px, it, 64)
    seed = C.c_uint32(); uo = np.zeros(64, np.float32); soft.oracle_rng_table(px, it, C.byref(seed), st.ptr(uo), 64)
    print("rng", px, it, s == seed.value, np.array_equal(u, uo))

def compare(W, H, spp, depth, label):
    scene, meta = ol.load_cornell(depth)
    cam = ol.cornell_camera(meta, W, H)
    t = time.time(); acc_o, col_o = ol.render(scene, cam, W, H, 0.001, 1, spp, kind="soft"); to = time.time()-t
    with api.Renderer(scene.desc, W, H, 0.001) as r:
        t = time.time(); r.render(cam, 1, spp, reset=True); r.synchronize(); tg = time.time()-t
        acc_g = r.read_accum(); col_g = r.read_color()
        n, ms = r.kernel_time()
    bad = np.count_nonzero(acc_g.view(np.uint32) != acc_o.view(np.uint32))
    badc = np.count_nonzero(col_g.view(np.uint32) != col_o.view(np.uint32))
    d = np.abs(acc_g-acc_o)
    rms = np.sqrt(np.mean((acc_g-acc_o)**2))/np.sqrt(np.mean(acc_o**2))
    print(f"{label}: {W}x{H} spp{spp} depth{depth}: acc mismatching floats {bad}/{acc_o.size} color {badc} maxabs {d.max():.3g} relRMS {rms:.3g} | gpu {ms:.2f} ms ({W*H*spp/ms/1e3:.1f} Msamples/s) cpu8 {to:.2f}s", flush=True)
    return acc_g, acc_o

compare(64, 64, 1, 4, "tiny")
compare(128, 128, 4, 4, "small")
compare(256, 256, 16, 8, "depth8")
compare(512, 512, 64, 4, "C1")
# timing at the headline shape
scene, meta = ol.load_cornell(8)
W, H = 1920, 1080
cam = ol.cornell_camera(meta, W, H)
with api.Renderer(scene.desc, W, H, 0.001) as r:
    for spp in (8, 64):
        r.kernel_time_reset()
        t = time.time(); r.render(cam, 1, spp, reset=True); r.synchronize(); tg = time.time()-t
        n, ms = r.kernel_time()
        print(f"1080p depth8 spp{spp}: kernel {ms:.2f} ms wall {tg*1e3:.2f} ms -> {W*H*spp/ms/1e3:.1f} Msamples/s", flush=True)
    r.enable_counters(True); r.render(cam, 1, 4, reset=True); r.synchronize(); print(r.read_counters())

"""VERDICT r4 item 1(a), CPU only: trips per sample of the three walks on the stand-ins, counted by the oracle.
   python tools/cpu_wide8_count.py [c5,c4,c3] [spp]
node trips = node visits (one wide node per trip); leaf trips: the 4-wide GPU loop tests two triangles of a leaf per trip
(ceil(count / 2) per leaf visit, an upper bound: a ray that ends in the leaf stops earlier), the 8-wide walk a whole leaf."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import oracle_lib as ol, scenes
from gpu_pathtracer_amd import api
import ctypes as C

which = (sys.argv[1] if len(sys.argv) > 1 else "c5").split(",")
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = ol.load("soft"); lib.oracle_leaf_visits.restype = C.c_uint64
for w in which:
    ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), w))
    W, H = ls.width // 4, ls.height // 4
    cam = ol.make_camera(*[None] * 0) if False else ls.camera
    # the stand-in's own camera at a quarter of the resolution (same framing: the camera record carries the resolution)
    import copy
    cam = copy.copy(ls.camera)
    films = {}
    for order, name in ((0, "reference order (binary)"), (2, "4-wide"), (3, "8-wide compressed")):
        t0 = time.time()
        films[order], _ = ol.render(ls, cam, ls.width, ls.height, ls.epsilon, 1, spp, kind="soft", order=order, threads=8, rank=0, n_ranks=16)
        c = ol.counters("soft"); lv = lib.oracle_leaf_visits(); n = c["samples"]
        rays = (c["closest_rays"] + c["shadow_rays"]) / n
        if hasattr(lib, "oracle_win") and os.environ.get("ORACLE_DIR_OVERRIDE"):
            lib.oracle_win.restype = C.c_uint64
            print("   K-window leaf trips/sample (K=3,4,6,8):", [round(lib.oracle_win(k) / n, 2) for k in range(4)])
        print(f"{w} {name:26s}: rays/sample {rays:.2f} node trips/sample {c['node_visits'] / n:7.2f} leaf visits {lv / n:6.2f} "
              f"triangle tests {c['prim_tests'] / n:6.2f} ({time.time() - t0:.0f} s, {n} samples)", flush=True)
    b = films[0].reshape(-1, 3).astype(np.float64)
    for o in (2, 3):
        a = films[o].reshape(-1, 3).astype(np.float64)
        rms = np.sqrt(((a - b) ** 2).mean(0)) / np.sqrt((b ** 2).mean(0))
        print(f"{w} order {o} against the reference order: floats differing {int(np.count_nonzero(films[o] != films[0]))} of {films[o].size}, relative RMS {rms}")
    ls.close()

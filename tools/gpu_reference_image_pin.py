"""Power of the pin against the reference's published render of its default scene (result/heterogeneous.png, kept as
tests/golden/reference_heterogeneous_64.npy): the GPU renders the shipped scene and deliberately wrong variants of it at
high sample counts and prints how far each lands from the reference's picture.  The thresholds of
tests/test_gpu_parity.py::test_volpath_gpu_film_against_the_reference_render are chosen from this table: the shipped scene
must pass, every wrong variant (bar the swap between two unbiased estimators) must fail.
usage (GPU box): python tools/gpu_reference_image_pin.py [spp]"""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol, scenes, refimg
from gpu_pathtracer_amd import api

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
want = refimg.load("reference_heterogeneous_64.npy")


def variant(edit):
    d = tempfile.mkdtemp()
    path = scenes.write_smoke_scene(d)
    js = json.load(open(path))
    edit(js)
    json.dump(js, open(path, "w"))
    return path


def het(js):
    return js["medium"][1]


def set_(d, **kw):
    d.update(kw)


VARIANTS = [
    ("as shipped", lambda js: None),
    ("extinction halved", lambda js: set_(het(js), sigmaA=[5.0] * 3, sigmaS=[45.0] * 3)),
    ("extinction x 0.8", lambda js: set_(het(js), sigmaA=[8.0] * 3, sigmaS=[72.0] * 3)),
    ("extinction x 1.25", lambda js: set_(het(js), sigmaA=[12.5] * 3, sigmaS=[112.5] * 3)),
    ("albedo 0.5 instead of 0.9", lambda js: set_(het(js), sigmaA=[50.0] * 3, sigmaS=[50.0] * 3)),
    ("albedo 0.8 instead of 0.9", lambda js: set_(het(js), sigmaA=[20.0] * 3, sigmaS=[80.0] * 3)),
    ("phase g = 0.8 instead of 0", lambda js: set_(het(js), g=0.8)),
    ("phase g = 0.3 instead of 0", lambda js: set_(het(js), g=0.3)),
    ("delta instead of ratio tracking", lambda js: set_(het(js), evalTransmittanceType=0)),
    ("residual ratio instead of ratio tracking", lambda js: set_(het(js), evalTransmittanceType=2)),
    ("grid box shifted by 0.1 in x", lambda js: set_(het(js), p0=[-0.53, 0.27, -0.2415], p1=[0.793, 1.593, 0.2415])),
    ("grid box shifted by 0.03 in x", lambda js: set_(het(js), p0=[-0.60, 0.27, -0.2415], p1=[0.723, 1.593, 0.2415])),
    ("maxDepth 5 instead of 17", lambda js: set_(js, maxDepth=5)),
    ("maxDepth 9 instead of 17", lambda js: set_(js, maxDepth=9)),
    ("light radiance x 0.9", lambda js: set_(js["light"][0], radiance=[15.3, 10.8, 3.6])),
    ("Path instead of Volpath (medium ignored, box removed)", lambda js: (set_(js, integrator="pt"), js["scene"].pop())),
]

print(f"# GPU film of the rebuilt shipped scene and wrong variants against the reference's picture: 512x512, {spp} spp, filmic + 8-bit,")
print("# 64x64 blocks; |frame-mean difference| (worst channel), mean and max |block difference|")
for name, edit in VARIANTS:
    ls = api.LoadedScene(variant(edit))
    W, H = ls.width, ls.height
    cam = ol.make_camera((0, 1.0, 6.8), (0, 1.0, 0), (0, 1, 0), (W, H), 19.5, 0.0, 7.0)
    cam.medium = ls.camera.medium
    t = time.time()
    with api.Renderer(ls.desc, W, H, ls.epsilon) as r:
        r.render(cam, 1, spp, reset=True)
        acc = r.read_accum()
    dt = time.time() - t
    m, bm, bx, means = refimg.compare(acc, spp, W, H, want)
    print(f"{name:55s} mean diff {m:.4f}  block mean {bm:.4f}  block max {bx:.3f}   frame means {means.round(4)}  ({dt:.1f} s)", flush=True)
    ls.close()
print("# reference frame means", want.mean(axis=(0, 1)).round(4))

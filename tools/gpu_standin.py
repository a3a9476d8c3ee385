"""Render one SURVEY stand-in at its full size (for rocprofv3 runs): python tools/gpu_standin.py c5 wide|reference [iterations [launches [sbvh]]]"""
import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenes
from gpu_pathtracer_amd import api
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
mode = sys.argv[2] if len(sys.argv) > 2 else "reference"
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
sbvh = len(sys.argv) > 5 and sys.argv[5] == "sbvh"          # the split tree (GPT_LOAD_SBVH) instead of the reference builder's
ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which), sbvh=sbvh)
with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
    r.set_traversal_order(mode)
    r.render(ls.camera, 1, 2, reset=True); r.synchronize()
    r.kernel_time_reset()
    for rep in range(reps):
        r.render(ls.camera, 1, spp, reset=True)
    r.synchronize()
    n, ms = r.kernel_time()
print(f"STANDIN {which} {mode}{' sbvh' if sbvh else ''} ({ls.desc.n_prims} primitives, {ls.desc.n_nodes} nodes): {ls.width}x{ls.height}, {spp} iterations per launch, {n} launches, {ms / n:.2f} ms per launch, "
      f"{ls.width * ls.height * spp * n / ms / 1e3:.1f} Msamples/s", flush=True)

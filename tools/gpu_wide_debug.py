"""Staged check of the wide walk's trace operator (gpt_debug_trace) against the oracle; each stage in its own process so that a
GPU fault in one does not hide the others.  usage (GPU box): python tools/gpu_wide_debug.py [stage]"""
import sys, subprocess, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
STAGES = ["cornell_few", "cornell_proper", "cornell_all", "soup_proper", "soup_all", "chain", "c5_proper", "c5_all", "render_cornell", "render_soup"]

def run(stage):
    import numpy as np
    import scenes, oracle_lib as ol
    from gpu_pathtracer_amd import api
    from test_gpu_parity import operator_rays
    if stage.startswith("render"):
        if stage == "render_cornell":
            scene, meta = ol.load_cornell(6)
        else:
            scene, meta = scenes.zoo_scene(max_depth=6, extra=scenes.random_soup(3000, 5, size=0.3))
        W, H, spp = 128, 96, 4
        cam = ol.cornell_camera(meta, W, H)
        want, _ = ol.render(scene, cam, W, H, 0.001, 1, spp, kind="soft", order=2)
        with api.Renderer(scene.desc, W, H, 0.001) as r:
            r.set_option("lds_scene", 0)
            r.set_traversal_order("wide")
            r.render(cam, 1, spp, reset=True)
            got = r.read_accum()
        bad = np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))
        print(stage, "floats differing:", bad, "of", got.size, flush=True)
        return
    if stage.startswith("cornell"):
        scene, _ = ol.load_cornell(4)
    elif stage.startswith("soup"):
        scene, _ = scenes.zoo_scene(max_depth=4, extra=scenes.random_soup(3000, 5, size=0.3))
    elif stage == "chain":
        from test_gpu_parity import chain_scene
        scene = chain_scene(36)[0] if isinstance(chain_scene(36), tuple) else chain_scene(36)
    else:
        import tempfile
        scene = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), "c5"))
    n = 64 if stage.endswith("few") else (20000 if not stage.startswith("c5") else 100000)
    rays = operator_rays(n, 17)
    if stage.endswith("proper") or stage.endswith("few") or stage == "chain":
        proper = ~np.isnan(rays).any(axis=1) & (np.abs(rays[:, 3:6]).sum(axis=1) > 0)
        rays = rays[proper]
    with api.Renderer(scene.desc, 64, 64, 0.001) as r:
        r.set_option("lds_scene", 0)
        r.set_traversal_order(2)
        prim, tb = r.trace_rays(rays)
    want_prim, want_tb = ol.trace_rays(scene, 0.001, rays, 2)
    hit = want_prim >= 0
    bad = np.nonzero(prim != want_prim)[0]
    same = (tb.view(np.uint32) == want_tb.view(np.uint32)) | (np.isnan(tb) & np.isnan(want_tb))
    badt = np.nonzero(hit & ~same.all(axis=1))[0]
    print(stage, len(rays), "rays:", len(bad), "other primitive,", len(badt), "other (t,b1,b2); hit rate %.3f" % hit.mean(), flush=True)
    for i in bad[:6]:
        print("   ray", i, rays[i], "gpu", prim[i], tb[i], "oracle", want_prim[i], want_tb[i])

if len(sys.argv) > 1:
    run(sys.argv[1])
else:
    for st in STAGES:
        p = subprocess.run(["timeout", "120", sys.executable, __file__, st], capture_output=True, text=True)
        out = [l for l in p.stdout.splitlines() if not l.startswith(("Load", "Merge", "Bvh", "Scene"))]
        print("\n".join(out) if out else f"{st}: no output", "| rc", p.returncode, flush=True)
        if p.returncode != 0:
            err = [l for l in p.stderr.splitlines() if "fault" in l or "Error" in l or "error" in l]
            print("   ", err[:3])

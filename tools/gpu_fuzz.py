"""Randomised GPU-vs-oracle soak: random triangle soups + Cornell parts, random materials, cameras, frame sizes,
iteration batching, integrator (pt / ao / vpt with random homogeneous and density-grid media, material-less boxes and
medium-filled meshes), traversal order, tree builder (the reference's or the split BVH), and memory path; every film must match the oracle bit for bit.
usage: python tools/gpu_fuzz.py <seconds> [seed]      (run under `timeout`; each case is small)"""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import scenes, oracle_lib as ol
from gpu_pathtracer_amd import api, scene_types as st

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
lib = ol.load("soft")
t_end = time.time() + budget
n_cases = n_bad = 0
while time.time() < t_end:
    n_cases += 1
    n_soup = int(rng.choice([0, 3, 40, 400, 3000]))
    mats = tuple(int(m) for m in rng.choice([0, 1, 2, 5, 6, 7, 8, 9, 10, 11, 12, 13], size=4))
    parts = []
    if n_soup:
        parts.append(scenes.random_soup(n_soup, int(rng.integers(1 << 30)), mats=mats, size=float(rng.choice([0.05, 0.15, 0.4]))))
    if rng.random() < 0.5:
        parts.append(scenes.uv_sphere((float(rng.uniform(-.5, .5)), float(rng.uniform(.3, 1.6)), float(rng.uniform(-.5, .5))),
                                      float(rng.uniform(.15, .45)), int(rng.choice(mats)), nu=int(rng.choice([6, 12, 20])), nv=int(rng.choice([4, 8, 14]))))
    # Volpath: random media; sometimes a material-less box holding one, sometimes a mesh filled with one
    vpt = bool(rng.random() < 0.35)
    media, grids, cam_medium = [], [], -1
    if vpt:
        for _ in range(int(rng.integers(1, 4))):
            sa, ss = rng.uniform(0.0, 1.5, 3), rng.uniform(0.05, 3.0, 3)
            g = float(rng.choice([0.0, 0.0, 0.5, -0.6, 0.0005]))
            if rng.random() < 0.5:
                media.append(st.make_medium(tuple(sa), tuple(ss), g, float(rng.choice([0.1, 1.0, 4.0]))))
            else:
                grid = scenes.smoke_grid(int(rng.integers(2, 20)), int(rng.integers(2, 20)), int(rng.integers(2, 20)), seed=int(rng.integers(1 << 30)))
                if grid.max() <= 0: grid[0, 0, 0] = np.float32(1)
                grids.append(grid)
                lo = rng.uniform(-1.2, -0.2, 3) + np.array([0, 1, 0])
                hi = lo + rng.uniform(0.4, 2.4, 3)
                a, sct = float(rng.uniform(0, 3)), float(rng.uniform(0.1, 12))
                media.append(st.make_het_medium((a, a, a), (sct, sct, sct), grid, tuple(lo), tuple(hi), int(rng.choice([1, 3, 40, 500])),
                                                int(rng.integers(0, 3)), g, 1.0))
        nm = len(media)
        if rng.random() < 0.6:
            lo = rng.uniform(-0.8, 0.0, 3) + np.array([0, 0.8, 0])
            hi = lo + rng.uniform(0.2, 1.0, 3)
            parts.append(scenes.box_mesh(tuple(lo), tuple(hi), -1, inside=int(rng.integers(-1, nm)), outside=int(rng.integers(-1, nm))))
        if parts and rng.random() < 0.5:
            parts[0]["triangle"]["mediumInside"] = int(rng.integers(-1, nm))
            parts[0]["triangle"]["mediumOutside"] = int(rng.integers(-1, nm))
        cam_medium = int(rng.integers(-1, nm))
    extra = scenes.concat(parts) if parts else None
    with_env = bool(rng.random() < 0.35)
    with_area = bool(rng.random() < 0.8) or not with_env
    depth = int(rng.choice([1, 2, 4, 7, 12, 20]))
    assign = {k: int(rng.choice(mats)) for k in ("short", "tall", "floor", "back", "left", "right", "ceil") if rng.random() < 0.5}
    scene, meta = scenes.zoo_scene(max_depth=depth, with_env=with_env, with_area_light=with_area, assign=assign, extra=extra)
    W, H = int(rng.choice([32, 40, 64, 97, 128, 200])), int(rng.choice([4, 9, 36, 64, 100, 131]))
    spp = int(rng.choice([1, 2, 3, 5, 9]))
    kind = rng.choice(["pinhole", "lens", "outside"])
    if kind == "pinhole":
        cam = ol.cornell_camera(meta, W, H)
    elif kind == "lens":
        cam = ol.make_camera((0, 1.0, 6.8), (0, 1.0, 0), (0, 1, 0), (W, H), 19.5, aperture=float(rng.uniform(0.02, 0.3)), focal=float(rng.uniform(4, 8)))
    else:
        cam = ol.make_camera((float(rng.uniform(-2, 2)), float(rng.uniform(0.2, 2.5)), float(rng.uniform(3, 8))), (0, 1, 0), (0, 1, 0), (W, H), float(rng.uniform(15, 70)))
    ao = bool(rng.random() < 0.2) and not vpt
    order = int(rng.choice([0, 0, -1, 2, 2]))           # reference / the product's default rule / 4-wide tree, one lane per ray
    near = order
    force_global = bool(rng.random() < 0.4)
    eps = float(rng.choice([0.001, 0.0005, 0.01]))
    if ao:
        scene.desc.set_integrator("ao", float(rng.choice([0.05, 0.5, 3.0])))
    force_walk = False
    if vpt:
        scene.set_mediums(media, keep=grids)
        scene.desc.set_integrator("vpt", depth)
        cam.medium = cam_medium
        force_walk = bool(rng.random() < 0.3)
    split_tree = bool(rng.random() < 0.25) and len(scene.prims) > 0          # the split BVH (gpt_sbvh_build) instead of the reference builder's tree
    if split_tree:
        sb_prims, sb_nodes, _, _ = api.sbvh_build(scene.prims, float(rng.choice([1e-5, 1e-3, 1.0])))
        scene.sb_keep = (sb_prims, sb_nodes)
        scene.desc.prims, scene.desc.n_prims = st.ptr(sb_prims), len(sb_prims)
        scene.desc.nodes, scene.desc.n_nodes = st.ptr(sb_nodes), len(sb_nodes)
    if order == 2 and len(scene.nodes) == 0:
        order = near = 0
    ref, _ = ol.render(scene, cam, W, H, eps, 1, spp, kind="soft", order=order)
    api.DEFAULT_OPTIONS["lds_scene"] = 0 if force_global else 1
    api.DEFAULT_OPTIONS["vpt_walk_kernel"] = 1 if force_walk else 0
    with api.Renderer(scene.desc, W, H, eps) as r:
        if len(scene.nodes) or order != 2:          # (an empty scene has no wide tree: refused, see gpt_set_traversal_order)
            r.set_traversal_order(order)
        if rng.random() < 0.5:
            r.render(cam, 1, spp, reset=True)
        else:                                   # split the iterations into two calls
            k = int(rng.integers(1, spp + 1))
            r.render(cam, 1, k, reset=True)
            if k < spp: r.render(cam, k + 1, spp - k, reset=False)
        got = r.read_accum()
    bad = int(np.count_nonzero(got.view(np.uint32) != ref.view(np.uint32)))
    tag = f"case {n_cases}: soup {n_soup} tris {len(scene.prims)} depth {depth} {W}x{H} spp {spp} cam {kind} env {with_env} area {with_area} ao {ao} vpt {vpt} media {len(media)} grids {len(grids)} walk {force_walk} order {order} global {force_global} split {split_tree}"
    if bad:
        n_bad += 1
        print("MISMATCH", bad, tag, flush=True)
    elif n_cases % 20 == 0:
        print("ok", tag, flush=True)
print(f"{n_cases} cases, {n_bad} mismatches, seed {seed}", flush=True)
sys.exit(1 if n_bad else 0)

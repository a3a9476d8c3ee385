"""The 8-wide prototype on the GPU (bash tools/build_variant.sh wide8 -DPT_WIDE8=1; GPT_LIB_PATH=var/libgpt_wide8.so): staged checks
against the oracle's walk of the same tree (include/gpt_wide8_bvh.h), each stage in its own process, then timings of the stand-ins.
   python tools/gpu_wide8.py check | time [c3,c4,c5]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np


def stage(name):
    import oracle_lib as ol, scenes
    import test_gpu_parity as tg
    from gpu_pathtracer_amd import api, scene_types as st
    if name.startswith("trace"):
        if name == "trace_soup":
            scene, _ = scenes.zoo_scene(max_depth=4, extra=scenes.random_soup(3000, 5, size=0.3)); n = 120_000
        elif name == "trace_cornell":
            scene, _ = ol.load_cornell(4); n = 120_000
        else:
            scene = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), "c5")); n = 200_000
        rays = tg.operator_rays(n, 17)
        with api.Renderer(scene.desc, 64, 64, 0.001) as r:
            r.set_option("lds_scene", 0)
            r.set_traversal_order(3)
            prim, tb = r.trace_rays(rays)
        want_prim, want_tb = ol.trace_rays(scene, 0.001, rays, 3)
        ref_prim, ref_tb = ol.trace_rays(scene, 0.001, rays, 0)
        hit = want_prim >= 0
        same = (tb.view(np.uint32) == want_tb.view(np.uint32)) | (np.isnan(tb) & np.isnan(want_tb))
        proper = ~np.isnan(rays).any(axis=1) & (np.abs(rays[:, 3:6]).sum(axis=1) > 0)
        closest = proper & (rays[:, 7] == 0)
        print(f"W8 {name}: {n} rays, hit rate {hit.mean():.3f}; GPU vs oracle(8-wide): other primitive {int(np.count_nonzero(prim != want_prim))}, "
              f"(t, b1, b2) differ on {int(np.count_nonzero(~same[hit].all(axis=1)))} hits; oracle(8-wide) vs oracle(reference order) on proper closest-hit rays: "
              f"other primitive {int(np.count_nonzero(want_prim[closest] != ref_prim[closest]))}, other (t,b1,b2) {int(np.count_nonzero((want_tb[closest] != ref_tb[closest]).any(axis=1)))}", flush=True)
        return
    if name.startswith("render"):
        integ = name.split("_")[1]
        fog = st.make_medium((0.0014, 0.0025, 0.0142), (0.70, 1.22, 1.90), 0.0, 0.3)
        scene, meta = scenes.zoo_scene(max_depth=9, with_env=True, extra=scenes.random_soup(2500, 11, size=0.25))
        W, H, spp = 160, 96, 5
        cam = ol.cornell_camera(meta, W, H)
        if integ == "ao": scene.desc.set_integrator("ao", 0.8)
        elif integ == "vpt":
            scene.set_mediums([fog]); scene.desc.set_integrator("vpt", 9); cam.medium = 0
        want, _ = ol.render(scene, cam, W, H, 0.001, 1, spp, kind="soft", order=3)
        ref, _ = ol.render(scene, cam, W, H, 0.001, 1, spp, kind="soft", order=0)
        with api.Renderer(scene.desc, W, H, 0.001) as r:
            r.set_traversal_order(3)
            r.set_option("max_batch", 2)
            r.render(cam, 1, spp, reset=True)
            got = r.read_accum()
        print(f"W8 {name}: floats differing GPU vs oracle(8-wide) {int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))} of {got.size}; "
              f"oracle(8-wide) vs oracle(reference order): {int(np.count_nonzero(want != ref))}", flush=True)
        return
    if name.startswith("standin"):
        w = name.split("_")[1]
        ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), w))
        W, H, spp = 480, 272, 4
        cam = ol.make_camera((0, 1.0, 6.8), (0, 1.0, 0), (0, 1, 0), (W, H), 19.5, 0.0, 7.0)
        want, _ = ol.render(ls, cam, W, H, ls.epsilon, 1, spp, kind="soft", order=3, threads=min(64, os.cpu_count()))
        with api.Renderer(ls.desc, W, H, ls.epsilon) as r:
            r.set_traversal_order(3)
            r.render(cam, 1, spp, reset=True)
            got = r.read_accum()
        print(f"W8 {name}: floats differing GPU vs oracle(8-wide) {int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))} of {got.size}", flush=True)
        return
    if name.startswith("time"):
        w = name.split("_")[1]
        ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), w))
        spp = 8 if w == "c5" else 32
        with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
            order = r.get_option("traversal_order")
            r.render(ls.camera, 1, 2, reset=True); r.synchronize()
            best = 1e9
            for _ in range(3):
                r.kernel_time_reset(); r.render(ls.camera, 1, spp, reset=True); r.synchronize()
                best = min(best, r.kernel_time()[1])
            import hashlib
            print(f"W8 {name}: order {order}, {ls.width}x{ls.height} x {spp}: {best:.2f} ms, {ls.width * ls.height * spp / best / 1e3:.1f} Msamples/s, film {hashlib.sha1(r.read_accum().tobytes()).hexdigest()[:12]}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "stage":
        stage(sys.argv[2]); sys.exit(0)
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what == "stages":
        stages = sys.argv[2].split(",")
    elif what == "check":
        stages = ["trace_cornell", "trace_soup", "render_pt", "render_ao", "render_vpt", "trace_c5", "standin_c5"]
    else:
        stages = ["time_" + w for w in (sys.argv[2] if len(sys.argv) > 2 else "c3,c4,c5").split(",")]
    for s in stages:
        p = subprocess.run(["timeout", "300", sys.executable, __file__, "stage", s], capture_output=True, text=True)
        out = [l for l in p.stdout.splitlines() if l.startswith("W8")]
        print("\n".join(out) if out else f"W8 {s}: FAILED rc={p.returncode} {p.stderr[-400:]}", flush=True)

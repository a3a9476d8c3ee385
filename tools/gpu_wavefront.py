"""The two schedulers side by side on the SURVEY stand-ins at full size: per (stand-in, traversal order) the path-kernel time of the
per-wave kernel ("scheduler" 0) and of the shade / trace phases over workgroup pools (1), whether the films are bit-equal, and the path slots in flight.
python tools/gpu_wavefront.py [c3,c4,c5] [reference,wide] [iterations] [-] [sbvh]"""
import hashlib
import os
import sys
import tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import scenes
from gpu_pathtracer_amd import api

which_list = (sys.argv[1] if len(sys.argv) > 1 else "c3,c4,c5").split(",")
modes = (sys.argv[2] if len(sys.argv) > 2 else "reference,wide").split(",")
spp_arg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
paths_list = [0]
sbvh = len(sys.argv) > 5 and sys.argv[5] == "sbvh"


def run(ls, mode, spp, scheduler, wf_paths=None, reps=2):
    with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
        r.set_traversal_order(mode)
        r.set_option("scheduler", scheduler)
        if not os.environ.get("GPT_WF_ONE_BATCH"):            # (counter runs: exactly one batch per scheduler)
            r.render(ls.camera, 1, 2, reset=True); r.synchronize()
        else:
            reps = 1
        best = 1e9
        for rep in range(reps):
            r.kernel_time_reset(); r.render(ls.camera, 1, spp, reset=True); r.synchronize()
            best = min(best, r.kernel_time()[1])
        film = r.read_accum()
        rounds = r.get_option("wf_paths") if scheduler else 0
        active = r.get_option("scheduler_active")
    return ls.width * ls.height * spp / best / 1e3, best, hashlib.sha1(film.tobytes()).hexdigest()[:16], rounds, active


for which in which_list:
    spp = spp_arg or {"c3": 32, "c4": 32, "c5": 8}[which]
    ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which), sbvh=sbvh)
    for mode in modes:
        base = run(ls, mode, spp, 0) if not os.environ.get("GPT_WF_ONE_BATCH") else (1.0, 0.0, "-", 0, 0)
        print(f"WF {which} {mode:9s}{' sbvh' if sbvh else ''} per-wave kernel : {base[0]:8.1f} Msamples/s ({base[1]:7.2f} ms / {spp} iterations) film {base[2]}", flush=True)
        for sched in ((1, 2) if not os.environ.get("GPT_WF_SCHEDULER") else (int(os.environ["GPT_WF_SCHEDULER"]),)):     # (counter runs: one of them)
            if sched == 2 and mode != "wide":
                continue
            wf = run(ls, mode, spp, sched, 0)
            print(f"WF {which} {mode:9s}{' sbvh' if sbvh else ''} stages, {'a lane per ray' if sched == 1 else 'ray stream   '}: {wf[0]:8.1f} Msamples/s ({wf[1]:7.2f} ms / {spp} iterations) film {wf[2]} "
                  f"{'EQUAL' if wf[2] == base[2] else 'DIFFERENT'} paths {wf[3]} active {wf[4]} x{wf[0] / base[0]:.2f}", flush=True)
    ls.close()

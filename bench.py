#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the path-tracing hot path on the BASELINE.json configuration.

Workload (config.workload): BASELINE.json configs[1] — cornell_box, 1920x1080, 8 bounces, area light + MIS,
1024 spp — rendered as K steps of SPP_PER_STEP iterations (default 16 x 64 = 1024 spp).  A "step" is SPP_PER_STEP
iterations of the path kernel over the whole frame.  Inputs (scene, camera, film) are resident in HBM before the timed
region starts.

N GPUs (launched by torch.distributed.run, one rank per GPU): every rank holds the whole scene, owns the 8x8
pixel tiles t with t % N == rank, and the timed region ends with ONE RCCL sum-reduce of the float3 accumulator
to rank 0 (disjoint supports, so the result is bit-identical to 1 GPU), issued by the library itself (gpt_reduce_film).
Total work is fixed as N grows ("strong").

What the JSON line carries besides the contract's fields (rank 0, N = 1):
  roofline      the bound of the dominant kernel on this workload is VALU issue (the 7.4 KB scene lives in LDS), so the
                headline fraction is wave64 VALU instructions per second against 1024 SIMDs x 2.4 GHz / 2 cycles
                (/opt/skills/guides/MI355X_MICROARCH.md "v_fma_f32 (wave64) 2 cyc").  The instruction count is MEASURED IN
                THIS RUN: bench.py re-runs one launch of the same binary under `rocprofv3 --pmc` (counters in their own
                passes, --kernel-trace only).  roofline.hbm keeps the HBM picture: SURVEY.md 8(d)'s algorithmic bytes (of
                the reference's algorithm and of this kernel's own ray counts), the counter-measured traffic and its
                fraction of the 8 TB/s peak.
  cpu_baseline  the oracle (CPU restatement) on one host core, bounded sample; host CPU model and core count.  The same beside every other
                leg of the line (config.other_configs.*.legs.default.cpu_baseline, config.volpath.cpu_baseline): one thread on a tile crop
                of the leg's own scene and frame, <= 8 s each.
  parity        GPU film against the pinned (glibc) oracle at 256x256 / 1024 spp / depth 8: per-channel relative RMS.
  config.other_configs   BASELINE.json configs 3 - 5 on their SURVEY.md 8(d) stand-ins (tests/standins.py, built from
                tests/golden/meshes.npz through the product loader), one full-size launch per leg (the loader's and gpt_begin's defaults; the reference's tree in the 4-wide walk and in the
                reference's order; the split tree): Msamples/s from the library's HIP events, and for two legs the VALU-issue fraction,
                active lanes and HBM-side GB/s of one extra rocprofv3 pass set.  Parity-test cases, not the headline: they are here so that their numbers are driver-witnessed.
  config.volpath         the reference's shipped Volpath scene (cornell_box/scene.json: density grid in a material-less box, 512 x 512, 17 bounces)
                rebuilt from fixtures through the product loader: one 64-iteration launch of the one-ray-at-a-time kernel, HIP events; its
                VALU-issue fraction, lanes and HBM-side traffic from rocprofv3 passes inside this run, like the c3 - c5 legs.
  config.per_rank        where every rank's wall time went: path kernel, accumulation kernel, reduce, Output, rest (HIP-event spans of the library).
  config.eight_gpu_projection   (N = 1) the job's eight tile shards one after another on this GPU, each timed like the real job minus the reduce,
                plus an ASSUMED reduce / closing-collective time: what to expect from the first real 8-GPU run.
`python bench.py --gpus N` without a torch.distributed.run environment launches itself under it (one rank per GPU).
Nothing here reads /root/reference.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, MAX_DEPTH, EPS = 1920, 1080, 8, 0.001
SPP_PER_STEP = 64
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD, CLOCK_HZ, VALU_CYCLES = 1024, 2.4e9, 2.0   # 256 CUs x 4 SIMDs, max clock, cycles per wave64 VALU instruction (same guide)
KERNEL = "pt_render_kernel<false, true, 1>"        # counting off, scene staged in LDS, Path integrator


def algorithmic_bytes_per_sample(c):
    """reference-layout bytes the reference's algorithm touches per sample:
    40 B per node visit (LinearBVHNode) + 176 B per primitive test (Primitive) + 72 B per bounce
    (Material) + 192 B per shadow ray (Area) + 60 B of film traffic (12 B kernel_color write + Output's
    12 R + 12 R + 12 W + 12 W)."""
    s = float(c["samples"])
    return (40.0 * c["node_visits"] + 176.0 * c["prim_tests"] + 72.0 * c["bounce_iters"] + 192.0 * c["shadow_rays"]) / s + 60.0


# SURVEY.md 8(d) counts for this exact workload (cornell 1920x1080, depth 8), measured with the oracle's
# counting pass (iterations 1-2): used when the CPU leg is skipped (N > 1 or --no-cpu-baseline).
B_ALG_CONFIG2 = 11917.5


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def cpu_baseline():
    """The oracle (CPU restatement of the same algorithm, same BVH) on ONE host core, bounded sample.
    Its work counters are the reference algorithm's N_node/N_prim/N_bounce/N_shadow, i.e. the inputs of
    the algorithmic-bytes figure (the GPU kernel traces fewer rays: it skips rays that cannot contribute)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    scene, meta = ol.load_cornell(MAX_DEPTH)
    cam = ol.cornell_camera(meta, WIDTH, HEIGHT)
    spp = 8
    t = time.perf_counter()
    ol.render(scene, cam, WIDTH, HEIGHT, EPS, 1, spp, kind="soft", threads=1)
    dt = time.perf_counter() - t
    cpu_baseline.b_alg = algorithmic_bytes_per_sample(ol.counters("soft"))
    model, cores = host_cpu()
    return {"value": WIDTH * HEIGHT * spp / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "host_cpu": model, "host_cores_total": cores,
            "sample": f"same scene/camera/frame, iterations 1-{spp} ({WIDTH * HEIGHT * spp} samples), "
                      f"oracle/liboracle_soft.so single thread, {dt:.1f} s"}


def cpu_crop_baseline(ls, n_crop, crop_rank, budget_s=8.0):
    """The CPU leg beside a stand-in / the Volpath scene ("a reduced-spp run of each GPU config's scene"): the oracle
    on ONE host thread, the leg's own scene, camera and frame, in the reference's traversal order on the loader's tree, restricted to the
    8x8 tiles t with t % n_crop == crop_rank (spread over the whole frame: the tile-ownership rule of the multi-GPU split) and to as
    many iterations as fit the time budget.  The oracle is the thing timed here and nowhere else."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    W, H = ls.width, ls.height
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    owned = (tiles - crop_rank + n_crop - 1) // n_crop
    kw = dict(kind="soft", rank=crop_rank, n_ranks=n_crop, threads=1, order=0)
    ol.render(ls, ls.camera, W, H, ls.epsilon, 1, 1, **kw)          # (the first call also pays the oracle's one-time scene set-up)
    t = time.perf_counter()
    ol.render(ls, ls.camera, W, H, ls.epsilon, 1, 1, **kw)
    one = time.perf_counter() - t
    spp = max(1, min(1024, int(budget_s / max(one, 1e-4))))
    t = time.perf_counter()
    ol.render(ls, ls.camera, W, H, ls.epsilon, 2, spp, **kw)
    dt = time.perf_counter() - t
    model, cores = host_cpu()
    return {"value": owned * 64 * spp / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port", "host_cpu": model, "host_cores_total": cores,
            "sample": f"same scene / camera / {W}x{H} frame, the {owned} tiles t % {n_crop} == {crop_rank} ({owned * 64} pixels), iterations 2-{spp + 1} "
                      f"({owned * 64 * spp} samples), reference traversal order, oracle/liboracle_soft.so single thread, {dt:.1f} s"}


def parity_check(api):
    """north_star's tolerance at its own sample count: GPU film vs the pinned (glibc) oracle, Cornell 256x256, depth 8.
    1024 spp when the host has the cores for it (67 M oracle samples), 128 spp otherwise.  The oracle is the checker."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    cores = os.cpu_count() or 1
    spp = 1024 if cores >= 32 else 128
    lib = ol.load("libm")
    scene, meta = ol.load_cornell(MAX_DEPTH, lib)
    cam = ol.cornell_camera(meta, 256, 256, lib)
    t = time.perf_counter()
    ref, _ = ol.render(scene, cam, 256, 256, EPS, 1, spp, kind="libm", threads=min(64, cores))
    dt = time.perf_counter() - t
    with api.Renderer(scene.desc, 256, 256, EPS) as r:
        r.render(cam, 1, spp, reset=True)
        got = r.read_accum()
    a, b = got.reshape(-1, 3).astype(np.float64), ref.reshape(-1, 3).astype(np.float64)
    rms = np.sqrt(((a - b) ** 2).mean(0)) / np.sqrt((b ** 2).mean(0))
    out = {"rms": [float(x) for x in rms], "tolerance": 1e-4, "ok": bool((rms <= 1e-4).all()),
           "what": f"GPU film vs oracle/liboracle_libm.so, cornell 256x256, {spp} spp, depth 8, per-channel relative RMS "
                   f"of the linear radiance (oracle {dt:.1f} s on {min(64, cores)} threads)",
           "gpu_frame_mean": [float(x) for x in (a / spp).mean(0)]}
    if spp == 1024:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_appendix_b.json")))["radiance_256_1024spp_depth8_mean"]
        out["oracle_mean_equals_survey_value"] = [float(f"{x:.9g}") for x in (b / spp).mean(0)] == gold
    return out


# ---- counters measured in this run -------------------------------------------------------------------------------------------

STANDINS = {"c3": ("config 3 stand-in: 3 x sphere.obj + cube-subdiv.obj, shaderball camera / light / metals, glass, substrate, checker; 1920x1080, depth 10", 32),
            "c4": ("config 4 stand-in: config-5 geometry under a procedural 1024x512 sky, no area light; 1920x1080, depth 7", 32),
            "c5": ("config 5 stand-in: Cornell walls + dragon + bunny2 + teapot + 9 spheres (248 574 triangles); 3840x2160, depth 16", 8)}


def load_standin(which, sbvh=False, reference_bvh=False):
    """The stand-in's scene directory is written, loaded through the product loader and removed.  The loader reports its progress
    on stdout like the reference's (parsescene.cpp): that goes to stderr here - stdout carries the one JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import standins
    from gpu_pathtracer_amd import api
    d = tempfile.mkdtemp(prefix="gpt_standin_")
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        return api.LoadedScene(standins.write_standin_scene(d, which), sbvh=sbvh, reference_bvh=reference_bvh)      # (the loader copies everything it reads)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        shutil.rmtree(d, ignore_errors=True)


def counter_child(which="c2", mode="reference"):
    """Run under rocprofv3 by live_counters(): two launches of the kernel to be counted (c2: 64 iterations at 1080p each)."""
    from gpu_pathtracer_amd import api, host
    if which == "c2":
        scene, meta = host.load_baked(os.path.join(ROOT, "tests", "golden", "cornell_pt.npz"), MAX_DEPTH)
        cam = host.camera_from_meta(meta, WIDTH, HEIGHT)
        with api.Renderer(scene.desc, WIDTH, HEIGHT, EPS) as r:
            r.render(cam, 1, SPP_PER_STEP, reset=True)
            r.render(cam, SPP_PER_STEP + 1, SPP_PER_STEP, reset=False)
            r.synchronize()
        return
    if which == "volpath":
        ls = load_shipped_volpath(api)
        with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
            r.render(ls.camera, 1, VOLPATH_SPP, reset=True)
            r.render(ls.camera, VOLPATH_SPP + 1, VOLPATH_SPP, reset=False)
            r.synchronize()
        return
    ls = load_standin(which, sbvh=mode.startswith("sbvh"), reference_bvh=mode.startswith("reference"))
    spp = STANDINS[which][1]
    with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
        if mode == "reference_tree+reference_order":
            r.set_traversal_order("reference")          # (the other legs: what gpt_begin chose)
        r.render(ls.camera, 1, spp, reset=True)
        r.render(ls.camera, spp + 1, spp, reset=False)
        r.synchronize()


def rocprof_pass(counters, workdir, tag, child=("c2", "reference")):
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = os.path.join(workdir, tag)
    env = dict(os.environ, TMPDIR=workdir)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", out, "-o", tag, "--",
                                                          sys.executable, os.path.abspath(__file__), "--counter-child", child[0], child[1]]
    if rocprof_pass.gave_up:
        raise RuntimeError("an earlier rocprofv3 pass of this run hung: the remaining passes are skipped")
    try:
        p = subprocess.run(cmd, cwd=workdir, env=env, capture_output=True, text=True, timeout=90)   # a pass takes 2 - 6 s; a hang must not cost the bench line
    except subprocess.TimeoutExpired:
        rocprof_pass.gave_up = True       # one hang: no further passes in this run (nine passes x the timeout would be a quarter of an hour)
        raise
    vals = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "render_kernel" in row.get("Kernel_Name", ""):
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    if not vals:
        raise RuntimeError(f"rocprofv3 pass {tag} produced no counters (rc {p.returncode}): {p.stderr[-300:]}")
    # the launches to be counted are the large ones (a stand-in child may also run small ones: keep the two largest per counter)
    big = {k: sorted(v)[-2:] for k, v in vals.items()}
    return {k: sum(v) / len(v) for k, v in big.items()}, max(len(v) for v in big.values())


rocprof_pass.gave_up = False


def live_counters():
    """SQ instruction counters and the two HBM traffic counters of the headline kernel, each set in its own rocprofv3 pass
    (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass; /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots"), on
    64-iteration launches of the same libgpt.so."""
    work = tempfile.mkdtemp(prefix="gpt_pmc_")
    try:
        sq, n = rocprof_pass(["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU",
                              "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"], work, "sq")
        fetch, _ = rocprof_pass(["FETCH_SIZE"], work, "fetch")
        write, _ = rocprof_pass(["WRITE_SIZE"], work, "write")
    finally:
        shutil.rmtree(work, ignore_errors=True)
    sq.update(fetch)
    sq.update(write)
    sq["launches_averaged"] = n
    sq["iterations_per_launch"] = SPP_PER_STEP
    return sq


def standin_leg(api, which, counters=True, cpu=True):
    """One BASELINE stand-in at full size, four legs, each one launch timed with the library's HIP events:
      "default"                         what a caller gets who calls gpt_scene_load, gpt_begin, gpt_render and nothing else: the loader's tree (the
                                        reference builder's unless it has oversized leaves - include/gpt.h) in the order gpt_begin picks (the 4-wide
                                        walk for every scene that does not fit LDS)
      "reference_tree+wide"             GPT_LOAD_REFERENCE_BVH: the reference builder's tree whatever it is, gpt_begin's order
      "reference_tree+reference_order"  ... in the reference's own traversal order (gpt_set_traversal_order)
      "sbvh+wide"                       GPT_LOAD_SBVH: the split tree, gpt_begin's order
    and for "default" and "reference_tree+wide" VALU-issue fraction, lanes and HBM-side traffic from rocprofv3 passes inside this run (SQ,
    FETCH_SIZE, WRITE_SIZE: each in its own pass)."""
    import numpy as np
    label, spp = STANDINS[which]
    out = {"workload": label, "iterations_per_launch": spp, "legs": {}}
    film = {}
    for leg in ("default", "reference_tree+wide", "reference_tree+reference_order", "sbvh+wide"):
        ls = load_standin(which, sbvh=leg.startswith("sbvh"), reference_bvh=leg.startswith("reference"))
        n_samples = ls.width * ls.height * spp
        with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
            if leg == "reference_tree+reference_order":
                r.set_traversal_order("reference")
            r.render(ls.camera, 1, spp, reset=True)             # same call as the timed one: the sample planes exist afterwards
            r.synchronize()
            best = None
            for _ in range(2):
                r.kernel_time_reset()
                r.render(ls.camera, 1, spp, reset=True)
                r.synchronize()
                n, ms = r.kernel_time()
                best = ms / max(1, n) if best is None else min(best, ms / max(1, n))
            film[leg] = r.read_accum()
            if cpu and leg == "default":
                try:
                    cpu_leg = cpu_crop_baseline(ls, 256, 37)
                except Exception as e:
                    cpu_leg = {"error": f"{type(e).__name__}: {e}"[:300]}
            out["legs"][leg] = {"value": n_samples / best / 1e3, "unit": "Msamples/s", "launch_ms": best,
                                "traversal_order": {0: "reference", 2: "wide4"}[r.get_option("traversal_order")],
                                "triangles": int(ls.desc.n_prims), "bvh_nodes": int(ls.desc.n_nodes),
                                "accumulator_sha1": hashlib.sha1(film[leg].tobytes()).hexdigest()[:16]}
            if cpu and leg == "default":
                out["legs"][leg]["cpu_baseline"] = cpu_leg
        ls.close()
    # against the reference order on the reference's tree: equal films except where two hits tie within rounding (include/gpt_wide_bvh.h)
    b = film["reference_tree+reference_order"].reshape(-1, 3).astype(np.float64)
    for leg in ("default", "reference_tree+wide", "sbvh+wide"):
        a = film[leg].reshape(-1, 3).astype(np.float64)
        out["legs"][leg]["vs_reference_order"] = {"floats_differing": int(np.count_nonzero(film[leg] != film["reference_tree+reference_order"])), "floats": int(film[leg].size),
                                                  "rel_rms": [float(x) for x in np.sqrt(((a - b) ** 2).mean(0)) / np.sqrt((b ** 2).mean(0))], "tolerance": 1e-4}
    for leg in (("default", "reference_tree+wide") if counters else ()):
        work = tempfile.mkdtemp(prefix="gpt_pmc_")
        try:
            sq, _ = rocprof_pass(["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"],
                                 work, "sq", (which, leg))
            fetch, _ = rocprof_pass(["FETCH_SIZE"], work, "fetch", (which, leg))
            write, _ = rocprof_pass(["WRITE_SIZE"], work, "write", (which, leg))
            ms = out["legs"][leg]["launch_ms"]
            peak_issue = N_SIMD * CLOCK_HZ / VALU_CYCLES
            traffic = (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0
            lanes = sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"]
            frac = sq["SQ_INSTS_VALU"] / (ms * 1e-3) / peak_issue
            out["legs"][leg]["roofline"] = {"bound": "valu_issue + memory latency (DESIGN.md section 4)",
                               "valu_issue_frac": frac, "active_lanes_of_64": lanes, "useful_frac": frac * lanes / 64.0,
                               "valu_insts_per_sample_lane": sq["SQ_INSTS_VALU"] / n_samples * 64,
                               "wait_any_over_wave_cycles": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                               "hbm_GBps": traffic / (ms * 1e-3) / 1e9, "hbm_frac_of_peak": traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "fetch_KiB_raw_per_launch": fetch["FETCH_SIZE"], "write_KiB_per_launch": write["WRITE_SIZE"],
                               "compulsory_bytes_per_launch": 16.0 * n_samples,
                               "counters": "rocprofv3 --pmc on launches of the same size of this libgpt.so, inside this run"}
        except Exception as e:
            out["legs"][leg]["roofline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            shutil.rmtree(work, ignore_errors=True)
    return out


def shard_projection(api, scene, cam, steps, out_ptr, device, one_rank):
    """What the 8-GPU job will cost, as far as ONE GPU can tell (no 8-GPU node has run this path yet): the eight ranks'
    shards (tiles t % 8 == k) of the SAME job, one after another on this GPU, each timed like the real job minus the reduce - path
    kernel by the library's events, wall clock around render + Output (root shard only) + synchronise.  The projection adds an
    ASSUMED reduce time; everything else in it is measured here."""
    rows = []
    for k in range(8):
        with api.Renderer(scene.desc, WIDTH, HEIGHT, EPS, device=device) as rs:
            rs.set_tile_owner(k, 8)
            rs.render(cam, 1, steps * SPP_PER_STEP, reset=True)          # the same call first: the sample planes exist afterwards
            rs.synchronize()
            rs.kernel_time_reset()
            t = time.perf_counter()
            rs.render(cam, 1, steps * SPP_PER_STEP, reset=True)
            if k == 0:
                rs.tonemap(steps * SPP_PER_STEP, bool(cam.filmic), out_ptr)
            rs.synchronize()
            wall = time.perf_counter() - t
            n, ms = rs.kernel_time()
            rows.append({"rank": k, "owned_tiles": rs.get_option("owned_tiles"), "launches": n, "kernel_ms": ms,
                         "output_kernel_ms": rs.get_option("output_kernel_us") / 1e3, "wall_ms": wall * 1e3})
    kernels = [x["kernel_ms"] for x in rows]
    slowest = max(x["wall_ms"] for x in rows)
    reduce_assumed_ms = 1.0        # 24.9 MB float3 frame, ring reduce to one root over xGMI at >= 25 GB/s effective + launch latency
    sync_assumed_ms = 0.3          # the closing barrier + the MAX all-reduce of the timings (two small collectives)
    t8 = slowest + reduce_assumed_ms + sync_assumed_ms
    return {"what": "the job's eight shards (tiles t % 8 == k) run one after another on this one GPU; measured: kernel and wall per shard; "
                    "assumed: reduce and closing collectives",
            "per_shard": rows, "kernel_ms_sum_over_one_rank_kernel_ms": sum(kernels) / one_rank["kernel_ms"],
            "kernel_ms_max_over_mean": max(kernels) / (sum(kernels) / 8.0),
            "wall_minus_kernel_ms_max": max(x["wall_ms"] - x["kernel_ms"] for x in rows),
            "reduce_assumed_ms": reduce_assumed_ms, "closing_collectives_assumed_ms": sync_assumed_ms,
            "projected_8gpu_ms": t8, "projected_8gpu_speedup": one_rank["wall_s"] * 1e3 / t8}


T0 = time.perf_counter()


def note(what):
    """progress to stderr (stdout carries the one JSON line): where the wall-clock time of a run goes"""
    print(f"[bench {time.perf_counter() - T0:7.1f} s] {what}", file=sys.stderr, flush=True)


def load_shipped_volpath(api):
    """scenes/cornell_box/scene.json of the reference rebuilt from fixtures (tests/standins.py: write_smoke_scene), through the product loader"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import standins
    d = tempfile.mkdtemp(prefix="gpt_smoke_")
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)                 # the loader's progress lines go to stderr: stdout carries the one JSON line
    try:
        return api.LoadedScene(standins.write_smoke_scene(d))
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        shutil.rmtree(d, ignore_errors=True)


VOLPATH_SPP = 64


def volpath_leg(api, counters=True, cpu=True):
    """The reference's shipped default scene (scenes/cornell_box/scene.json: "vpt", 17 bounces, a 100 x 100 x 40 density grid in a
    material-less box, 512 x 512) rebuilt on disk from this repository's fixtures (tests/standins.py: write_smoke_scene; where
    /root/reference exists tests/test_scene_loader.py shows it loads to the shipped scene bit for bit) and read through the product
    loader: the one-ray-at-a-time Volpath kernel, one 64-iteration launch by HIP events."""
    import numpy as np
    ls = load_shipped_volpath(api)
    spp = VOLPATH_SPP
    with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
        r.render(ls.camera, 1, 2, reset=True)
        r.synchronize()
        best = None
        for _ in range(2):
            r.kernel_time_reset()
            r.render(ls.camera, 1, spp, reset=True)
            r.synchronize()
            n, ms = r.kernel_time()
            best = ms if best is None else min(best, ms)
        film = r.read_accum()
        walk = int(r.get_option("walk_kernel_active"))
    out = {"workload": "the reference's shipped scenes/cornell_box/scene.json rebuilt from fixtures: Cornell walls + 100x100x40 density grid "
                       f"(sigmaT 100, albedo 0.9) in a material-less box, {ls.width}x{ls.height}, 17 bounces, ratio tracking, iterMax 2000, "
                       f"{spp} spp in one launch",
           "value": ls.width * ls.height * spp / best / 1e3, "unit": "Msamples/s", "launch_ms": best, "walk_kernel": walk,
           "timed": "path-kernel launch, HIP events of the library (gpt_kernel_time)",
           "accumulator_sha1": hashlib.sha1(film.tobytes()).hexdigest()[:16],
           "mean_radiance": [float(x) for x in (film.reshape(-1, 3).astype(np.float64).mean(0) / spp)]}
    if cpu:
        try:
            out["cpu_baseline"] = cpu_crop_baseline(ls, 64, 21)
        except Exception as e:
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if counters:
        # the counters of the c3 - c5 legs for this kernel (pt_render_kernel<false, true, PT_IT_VPT_WALK>): launches of the same size
        # of the same libgpt.so under rocprofv3 --pmc inside this run, each counter set in its own pass
        work = tempfile.mkdtemp(prefix="gpt_pmc_")
        try:
            sq, _ = rocprof_pass(["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"],
                                 work, "sq", ("volpath", "default"))
            fetch, _ = rocprof_pass(["FETCH_SIZE"], work, "fetch", ("volpath", "default"))
            write, _ = rocprof_pass(["WRITE_SIZE"], work, "write", ("volpath", "default"))
            n_samples = ls.width * ls.height * spp
            peak_issue = N_SIMD * CLOCK_HZ / VALU_CYCLES
            traffic = (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0
            lanes = sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"]
            frac = sq["SQ_INSTS_VALU"] / (best * 1e-3) / peak_issue
            out["roofline"] = {"bound": "valu_issue at low lane occupancy (DESIGN.md section 7)", "valu_issue_frac": frac, "active_lanes_of_64": lanes,
                               "useful_frac": frac * lanes / 64.0, "valu_insts_per_sample_lane": sq["SQ_INSTS_VALU"] / n_samples * 64,
                               "wait_any_over_wave_cycles": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                               "hbm_GBps": traffic / (best * 1e-3) / 1e9, "hbm_frac_of_peak": traffic / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "fetch_KiB_raw_per_launch": fetch["FETCH_SIZE"], "write_KiB_per_launch": write["WRITE_SIZE"],
                               "compulsory_bytes_per_launch": 16.0 * n_samples,
                               "counters": "rocprofv3 --pmc on launches of the same size of this libgpt.so, inside this run"}
        except Exception as e:
            out["roofline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            shutil.rmtree(work, ignore_errors=True)
    ls.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)       # 4 x 64 iterations = one full-size launch
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true", help="skip the rocprofv3 passes (roofline fields that need them are null)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-square", action="store_true", help="skip the square-frame figure and the counting render (a rocprofv3 --stats run then sees "
                                                             "only the headline kernel's full-size launches)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the stand-ins of BASELINE.json configs 3 - 5 (config.other_configs)")
    ap.add_argument("--counter-child", nargs=2, metavar=("WHICH", "ORDER"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.counter_child:
        return counter_child(*args.counter_child)

    import numpy as np
    import torch
    from gpu_pathtracer_amd import api, host

    rank = int(os.environ.get("RANK", "0"))
    share_gpu = bool(os.environ.get("GPT_BENCH_SHARE_GPU"))       # every rank on GPU 0: functional check of the N-rank path on a 1-GPU box
    local_rank = 0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # called as `python bench.py --gpus N`: launch the N ranks (one per GPU) the way the driver does
            port = 29500 + os.getpid() % 2000
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            raise SystemExit(subprocess.call(cmd))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is the rendezvous (it carries the RCCL unique id to the ranks) and the barrier; the framebuffer
        # reduce itself is issued by the library (gpt_reduce_film -> ncclReduce on the renderer's stream).  With
        # GPT_BENCH_SHARE_GPU=1 (all ranks on GPU 0; RCCL refuses duplicate devices) the same code path runs over gloo.
        backend = "gloo" if share_gpu else os.environ.get("GPT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    note("imports done")
    scene, meta = host.load_baked(os.path.join(ROOT, "tests", "golden", "cornell_pt.npz"), MAX_DEPTH)
    cam = host.camera_from_meta(meta, WIDTH, HEIGHT)
    n_floats = WIDTH * HEIGHT * 3
    out = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    r = api.Renderer(scene.desc, WIDTH, HEIGHT, EPS, device=local_rank)
    r.set_tile_owner(rank, world)
    comm = None
    if world > 1:
        from gpu_pathtracer_amd import distributed as gd
        # GPT_BENCH_TRY_NATIVE (test hook): attempt the library's RCCL set-up even where it must fail (two ranks on one GPU),
        # to exercise the agreed fall-back to the torch.distributed reduce
        comm = gd.FilmReducer(r, dist, rank, world, native=(backend == "nccl" or bool(os.environ.get("GPT_BENCH_TRY_NATIVE"))))

    if comm is not None:
        # RCCL builds its rings and buffers on the first collective of a communicator: do that once outside the timed region,
        # whatever --warmup says (the film is still zero here)
        comm.reduce(root=0)
        r.synchronize()

    def barrier():
        r.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    host_reduce_s = [0.0]          # the gloo stand-in synchronises inside reduce(): timed on the host; the RCCL reduce by the library's events

    def job(steps):
        # K steps of SPP_PER_STEP iterations each, handed to the renderer in one call: it cuts them into launches of
        # up to 256 iterations (a launch has a fixed cost; the reference's one-iteration-per-Render is the other extreme)
        r.render(cam, 1, steps * SPP_PER_STEP, reset=True)
        if comm is not None:
            th = time.perf_counter()
            comm.reduce(root=0)                                 # the one collective: float3 framebuffer over xGMI
            host_reduce_s[0] = time.perf_counter() - th
        if rank == 0:
            if comm is not None:
                comm.tonemap_reduced(steps * SPP_PER_STEP, bool(cam.filmic), out.data_ptr())
            else:
                r.tonemap(steps * SPP_PER_STEP, bool(cam.filmic), out.data_ptr())   # Output
        r.synchronize()

    if args.warmup > 0:
        job(args.warmup)
    barrier()
    r.kernel_time_reset()
    t0 = time.perf_counter()
    job(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    launches, kernel_ms = r.kernel_time()
    note(f"timed region done: {dt:.3f} s for {args.steps} steps")

    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    samples = WIDTH * HEIGHT * SPP_PER_STEP * args.steps
    # Where this rank's wall time went: the path kernel and the accumulation kernel behind every launch (HIP events), the reduce (the
    # library's events around ncclReduce; for the gloo stand-in, which synchronises, the host clock around it - that one includes
    # waiting for the kernel), Output on the root, and the rest (launch gaps, the two barriers, waiting for slower ranks).
    reduce_ms = None
    if comm is not None:
        reduce_ms = r.get_option("last_reduce_us") / 1e3 if comm.native else host_reduce_s[0] * 1e3
    tonemap_ms = r.get_option("last_tonemap_us") / 1e3 if rank == 0 else None
    output_ms = r.get_option("output_kernel_us") / 1e3        # pt_output_kernel: sample planes -> accumulator, once per launch (shards with the tiles)
    mine = {"rank": rank, "owned_tiles": r.get_option("owned_tiles"), "sample_plane_bytes": r.get_option("sample_plane_bytes"),
            "launches": launches, "kernel_ms": kernel_ms, "output_kernel_ms": output_ms, "wall_s": dt, "reduce_ms": reduce_ms, "tonemap_ms": tonemap_ms,
            "reduce_timed_by": None if comm is None else ("HIP events around ncclReduce" if comm.native else "host clock around the synchronising gloo reduce (includes waiting for this rank's kernel)"),
            "rest_ms": dt * 1e3 - kernel_ms - output_ms - (reduce_ms if (comm is not None and comm.native) else 0.0) - (tonemap_ms or 0.0)}
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank == 0:
        acc_host = comm.read_reduced() if comm is not None else r.read_accum()
        img = acc_host.reshape(-1, 3) / np.float32(args.steps * SPP_PER_STEP)
        finite = bool(np.isfinite(img).all())
        frame_sha1 = hashlib.sha1(acc_host.tobytes()).hexdigest()[:16]
        lib_sha1 = hashlib.sha1(open(api.LIB_PATH, "rb").read()).hexdigest()[:16]
        options = {k: r.get_option(k) for k in ("lds_scene", "lds_scene_active", "max_batch", "last_batch", "chunk_iters", "sample_plane_bytes")}
        cpu = None
        b_alg = B_ALG_CONFIG2
        single = world == 1
        if single and not args.no_cpu_baseline:
            cpu = cpu_baseline()
            note("cpu_baseline done")
            b_alg = cpu_baseline.b_alg      # algorithmic bytes of the reference algorithm on this workload
        # this rank's launches cover its own tiles: samples per launch on this rank
        samples_per_launch = samples / world / max(1, launches)
        avg_ms = kernel_ms / max(1, launches)

        # the kernel's OWN ray counts (it does not trace rays that provably cannot contribute): counting build, 2 iterations
        b_alg_kernel = None
        square = None
        if single and not args.no_square:
            r.enable_counters(True)
            r.render(cam, 1, 2, reset=True)
            b_alg_kernel = algorithmic_bytes_per_sample(r.read_counters())
            r.enable_counters(False)
            # the camera is framed for a square image - at 16:9, 44 % of the primary rays miss the box and
            # cost one traversal each.  The same scene at 1080 x 1080 (every pixel sees the box), same kernel:
            sq_cam = host.camera_from_meta(meta, 1088, 1080)
            with api.Renderer(scene.desc, 1088, 1080, EPS, device=local_rank) as rs:
                rs.render(sq_cam, 1, 256, reset=True)            # the same call as the timed one: the sample planes exist afterwards
                rs.synchronize()
                rs.kernel_time_reset()
                ts = time.perf_counter()
                rs.render(sq_cam, 1, 256, reset=True)
                rs.synchronize()
                wall = time.perf_counter() - ts
                n_sq, ms_sq = rs.kernel_time()
                square = {"frame": "1088x1080 (square framing: every primary ray enters the box)", "spp": 256,
                          "value": 1088 * 1080 * 256 / (ms_sq * 1e-3) / 1e6, "unit": "Msamples/s",
                          "timed": "path-kernel launches, HIP events of the library (gpt_kernel_time)", "launches": n_sq,
                          "wall_clock_value": 1088 * 1080 * 256 / wall / 1e6}

        note("square frame and kernel ray counts done")
        projection = None
        if single and not args.no_square:
            try:
                projection = shard_projection(api, scene, cam, args.steps, out.data_ptr(), local_rank, mine)
            except Exception as e:
                projection = {"error": f"{type(e).__name__}: {e}"[:300]}
            note("8-shard projection done")
        live, live_err = None, None
        if single and not args.no_counters:
            try:
                live = live_counters()
            except Exception as e:          # rocprofv3 missing or refused: say so, print nulls
                live_err = f"{type(e).__name__}: {e}"[:300]
        note("headline counters (3 rocprofv3 passes) done")
        per_sample = None
        if live:
            n_per_launch_samples = WIDTH * HEIGHT * live["iterations_per_launch"]
            per_sample = {k: live[k] / n_per_launch_samples for k in live if k.startswith(("SQ_", "FETCH", "WRITE"))}
        peak_issue = N_SIMD * CLOCK_HZ / VALU_CYCLES
        roof = {"bound": "valu_issue", "achieved": None, "peak": peak_issue / 1e9, "unit": "G wave64-VALU-instructions/s", "frac": None,
                "traffic": None, "kernel": f"pt::{KERNEL} (counting off, scene staged in LDS, Path integrator)",
                "avg_launch_ms": avg_ms, "launches": launches,
                "iterations_per_launch": args.steps * SPP_PER_STEP / max(1, launches), "samples_per_launch": samples_per_launch,
                "peak_is": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)",
                "counters": "rocprofv3 --pmc on 64-iteration launches of this libgpt.so, inside this run" if live else None,
                "counters_error": live_err}
        hbm = {"algorithmic_bytes_per_sample": b_alg, "algorithmic_bytes_per_sample_kernel_counts": b_alg_kernel,
               "algorithmic_GBps": b_alg * samples_per_launch / (avg_ms * 1e-3) / 1e9,
               "note": "algorithmic = reference-layout bytes of SURVEY.md 8(d): logical traffic; the 7.4 KB scene is LDS-resident, "
                       "so it exceeds the 8 TB/s HBM peak by construction and is not a fraction of a physical roof",
               "counter_traffic_bytes_per_launch": None, "counter_GBps": None, "frac_of_peak": None, "peak_GBps": HBM_PEAK_GBS}
        if per_sample:
            n_valu = per_sample["SQ_INSTS_VALU"] * samples_per_launch
            lanes = live["SQ_THREAD_CYCLES_VALU"] / live["SQ_ACTIVE_INST_VALU"]
            roof.update({"achieved": n_valu / (avg_ms * 1e-3) / 1e9, "frac": n_valu / (avg_ms * 1e-3) / peak_issue,
                         "valu_insts_per_launch": n_valu, "valu_insts_per_sample_lane": per_sample["SQ_INSTS_VALU"] * 64,
                         "salu_insts_per_launch": per_sample["SQ_INSTS_SALU"] * samples_per_launch,
                         "lds_insts_per_launch": per_sample["SQ_INSTS_LDS"] * samples_per_launch,
                         "active_lanes_of_64": lanes, "useful_frac": n_valu / (avg_ms * 1e-3) / peak_issue * lanes / 64.0,
                         "wait_any_over_wave_cycles": live["SQ_WAIT_ANY"] / live["SQ_WAVE_CYCLES"],
                         "wait_inst_any_over_wave_cycles": live["SQ_WAIT_INST_ANY"] / live["SQ_WAVE_CYCLES"]})
            # HBM: FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md "HBM")
            traffic = (2.0 * per_sample["FETCH_SIZE"] + per_sample["WRITE_SIZE"]) * 1024.0 * samples_per_launch
            roof["traffic"] = traffic
            hbm.update({"counter_traffic_bytes_per_launch": traffic, "counter_GBps": traffic / (avg_ms * 1e-3) / 1e9,
                        "frac_of_peak": traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "compulsory_bytes_per_launch": 16.0 * samples_per_launch,
                        "fetch_KiB_raw_per_launch": per_sample["FETCH_SIZE"] * samples_per_launch,
                        "write_KiB_per_launch": per_sample["WRITE_SIZE"] * samples_per_launch})
        roof["hbm"] = hbm
        others = None
        if single and not args.no_other_configs:
            others = {}
            for which in ("c3", "c4", "c5"):
                try:
                    others[which] = standin_leg(api, which, counters=not args.no_counters, cpu=not args.no_cpu_baseline)
                except Exception as e:
                    others[which] = {"error": f"{type(e).__name__}: {e}"[:300]}
                note(f"other_configs {which} done")
        # Volpath on the shape of the only scene the reference ships complete (scenes/cornell_box/scene.json: 512 x 512,
        # 17 bounces, a 100 x 100 x 40 density grid with sigmaT = 100 inside a material-less box, ratio tracking, iterMax 2000) - the
        # one-ray-at-a-time kernel (pt_render_kernel<..., PT_IT_VPT_WALK, ...>), one 64-iteration launch by HIP events
        volpath = None
        if single and not args.no_other_configs:
            try:
                volpath = volpath_leg(api, counters=not args.no_counters, cpu=not args.no_cpu_baseline)
            except Exception as e:
                volpath = {"error": f"{type(e).__name__}: {e}"[:300]}
            note("volpath leg done")
        par = None
        if single and not args.no_parity:
            par = parity_check(api)
            note("parity check done")
        line = {
            "metric": "Msamples/s at 1920x1080, 8-bounce PT",
            "value": samples / dt_max / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cornell_box 1920x1080, 8 bounces, area light + MIS, "
                                   f"{args.steps * SPP_PER_STEP} spp ({args.steps} steps x {SPP_PER_STEP} iterations)",
                       "scene": "cornell_pt (36 triangles, 27 BVH nodes)", "spp_per_step": SPP_PER_STEP,
                       "tiles": "8x8 pixels, tile % n_gpus == rank", "all_finite": finite,
                       "accumulator_sha1": frame_sha1, "libgpt_sha1": lib_sha1,
                       "renderer_options": options, "options_set": dict(r.options_set),
                       "env_overrides": dict(api.ENV_OVERRIDES, **{k: os.environ[k] for k in ("GPT_BENCH_SHARE_GPU", "GPT_BENCH_BACKEND", "GPT_BENCH_TRY_NATIVE") if os.environ.get(k)}),
                       "reduce": (comm.kind if comm is not None else None), "per_rank": per_rank,
                       "square_frame": square, "eight_gpu_projection": projection, "other_configs": others, "volpath": volpath,
                       "mean_radiance": [float(x) for x in img.astype(np.float64).mean(0)]},
            "roofline": roof,
        }
        if par is not None:
            line["parity"] = par
            line["parity_rms"] = max(par["rms"])
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    r.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

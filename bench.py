#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the path-tracing hot path on the BASELINE.json configuration.

Workload (config.workload): BASELINE.json configs[1] — cornell_box, 1920x1080, 8 bounces, area light + MIS,
1024 spp — rendered as K steps of SPP_PER_STEP iterations (default 16 x 64 = 1024 spp).  A "step" is one
gpt_render() call = one launch of the path kernel over the whole frame for SPP_PER_STEP iterations.
Inputs (scene, camera, film) are resident in HBM before the timed region starts.

N GPUs (launched by torch.distributed.run, one rank per GPU): every rank holds the whole scene, owns the 8x8
pixel tiles t with t % N == rank, and the timed region ends with ONE RCCL sum-reduce of the float3 accumulator
to rank 0 (disjoint supports, so the result is bit-identical to 1 GPU).  Total work is fixed as N grows
("strong").

Prints one JSON line on rank 0.  Nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, MAX_DEPTH, EPS = 1920, 1080, 8, 0.001
SPP_PER_STEP = 64
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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


def cpu_baseline():
    """The oracle (CPU restatement of the same algorithm, same BVH) on ONE host core, bounded sample.
    Its work counters are the reference algorithm's N_node/N_prim/N_bounce/N_shadow, i.e. the inputs of
    the algorithmic-bytes figure (the GPU kernel traces fewer rays: it skips rays that cannot contribute)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    scene, meta = ol.load_cornell(MAX_DEPTH)
    cam = ol.cornell_camera(meta, WIDTH, HEIGHT)
    spp = 2
    t = time.perf_counter()
    ol.render(scene, cam, WIDTH, HEIGHT, EPS, 1, spp, kind="soft", threads=1)
    dt = time.perf_counter() - t
    cpu_baseline.b_alg = algorithmic_bytes_per_sample(ol.counters("soft"))
    return {"value": WIDTH * HEIGHT * spp / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"same scene/camera/frame, iterations 1-{spp} ({WIDTH * HEIGHT * spp} samples), "
                      f"oracle/liboracle_soft.so single thread, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)       # 4 x 64 iterations = one full-size launch
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from gpu_pathtracer_amd import api, host

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("GPT_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  GPT_BENCH_BACKEND=gloo + GPT_BENCH_SHARE_GPU=1 runs the same N-rank code
        # path with every rank on GPU 0 (functional check on a 1-GPU box; RCCL refuses duplicate devices).
        backend = os.environ.get("GPT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    scene, meta = host.load_baked(os.path.join(ROOT, "tests", "golden", "cornell_pt.npz"), MAX_DEPTH)
    cam = host.camera_from_meta(meta, WIDTH, HEIGHT)
    n_floats = WIDTH * HEIGHT * 3
    # the film IS the reduce buffer: torch tensors bound as accumulator / last-sample planes
    acc = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
    col = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
    out = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    r = api.Renderer(scene.desc, WIDTH, HEIGHT, EPS, device=local_rank)
    r.bind_film(acc.data_ptr(), col.data_ptr())
    r.set_tile_owner(rank, world)

    def barrier():
        r.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def job(steps):
        # K steps of SPP_PER_STEP iterations each, handed to the renderer in one call: it cuts them into launches of
        # up to 256 iterations (a launch has a fixed cost; the reference's one-iteration-per-Render is the other extreme)
        r.render(cam, 1, steps * SPP_PER_STEP, reset=True)
        r.synchronize()
        if dist is not None:
            dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)     # the one collective: float3 framebuffer over xGMI
            torch.cuda.synchronize()                           # the reduce runs on torch's streams, Output on ours
        if rank == 0:
            r.tonemap(steps * SPP_PER_STEP, bool(cam.filmic), out.data_ptr())   # Output on the root
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

    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    samples = WIDTH * HEIGHT * SPP_PER_STEP * args.steps
    if rank == 0:
        img = acc.cpu().numpy().reshape(-1, 3) / np.float32(args.steps * SPP_PER_STEP)
        finite = bool(np.isfinite(img).all())
        import hashlib
        frame_sha1 = hashlib.sha1(acc.cpu().numpy().tobytes()).hexdigest()[:16]
        cpu = None
        b_alg = B_ALG_CONFIG2
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
            b_alg = cpu_baseline.b_alg      # algorithmic bytes of the reference algorithm on this workload
        # this rank's launches cover its own tiles: samples per launch on this rank
        samples_per_launch = samples / world / max(1, launches)
        avg_ms = kernel_ms / max(1, launches)
        achieved = b_alg * samples_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))          # counters were taken on 256-iteration launches: scale to this run's
                traffic = rec.get("hbm_bytes_per_launch") * (samples / world / max(1, launches)) / (WIDTH * HEIGHT * rec.get("iterations_per_launch", 256))
            except Exception:
                traffic = None
        # the compute-side picture next to the (logical) HBM figure: wave-level VALU instructions issued per second
        # against what 1024 SIMDs can issue (one fp32 wave64 instruction per ~2.43 cycles, measured with
        # tools/micro/issue_rate.hip); instruction counts come from the committed SQ counter pass
        valu = None
        sq = os.path.join(ROOT, "profiles", "pmc_sq.json")
        if os.path.exists(sq) and world == 1:
            try:
                rec = json.load(open(sq))
                n_valu = rec.get("valu_insts_per_launch") * samples_per_launch / (WIDTH * HEIGHT * rec.get("iterations_per_launch", 256))
                peak_issue = 1024 * 2.4e9 / 2.43
                valu = {"insts_per_launch": n_valu, "achieved_per_s": n_valu / (avg_ms * 1e-3), "peak_per_s": peak_issue,
                        "frac": n_valu / (avg_ms * 1e-3) / peak_issue,
                        "active_lanes_of_64": json.load(open(sq)).get("active_lanes_per_valu_inst")}
            except Exception:
                valu = None
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
                       "accumulator_sha1": frame_sha1,
                       "mean_radiance": [float(x) for x in img.astype(np.float64).mean(0)]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "pt::pt_render_kernel<false, true, 1> (counting off, scene staged in LDS, Path integrator)", "avg_launch_ms": avg_ms, "launches": launches,
                         "iterations_per_launch": args.steps * SPP_PER_STEP / max(1, launches),
                         "algorithmic_bytes_per_sample": b_alg, "valu_issue": valu,
                         "note": "algorithmic = reference-layout bytes (SURVEY.md 8d); the 7.4 KB scene is "
                                 "cache-resident, so this logical figure can exceed the HBM peak"},
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    r.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

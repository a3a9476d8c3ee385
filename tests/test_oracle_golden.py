"""Pins the CPU oracle against the reference's known answers (no GPU).

Golden sources:
  tests/golden/survey_appendix_b.json  values produced by the reference's own code
  tests/golden/rng_table.json          produced from the real thrust headers (tools/gen_rng_golden.cpp)
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import scenes
from gpu_pathtracer_amd import scene_types as st

GOLD = json.load(open(os.path.join(ol.GOLDEN, "survey_appendix_b.json")))
RNG = json.load(open(os.path.join(ol.GOLDEN, "rng_table.json")))


def nine(x):
    """the survey printed 9 significant digits, which identifies a float32"""
    return float(f"{float(x):.9g}")


@pytest.mark.parametrize("kind", ["libm", "soft"])
def test_rng_matches_thrust(kind):
    lib = ol.load(kind)
    for row in RNG:
        if "pixel" not in row:
            continue
        seed = C.c_uint32()
        u = np.zeros(16, np.float32)
        lib.oracle_rng_table(row["pixel"], row["iter"], C.byref(seed), st.ptr(u), 16)
        assert seed.value == row["seed"]
        assert u.view(np.uint32).tolist() == row["u_bits"]


def test_rng_matches_survey_table():
    lib = ol.load("libm")
    for row in GOLD["rng"]:
        seed = C.c_uint32()
        u = np.zeros(4, np.float32)
        lib.oracle_rng_table(row["pixel"], row["iter"], C.byref(seed), st.ptr(u), 4)
        assert seed.value == row["seed"]
        assert [nine(x) for x in u] == row["u"]


def test_uniform_can_reach_one():
    # float(x-1)/2^31 rounds to 1.0f for x-1 >= 2147483584
    assert np.float32(2147483584) / np.float32(2147483648.0) == np.float32(1.0)


def test_cornell_bvh_listing():
    scene, _ = ol.load_cornell(4, ol.load("libm"))
    g = GOLD["cornell_bvh"]
    assert len(scene.prims) == g["n_prims"] and len(scene.nodes) == g["n_nodes"]
    assert 32 + 176 * len(scene.prims) + 40 * len(scene.nodes) == g["cache_bytes"]
    assert np.allclose(scene.root_box, g["root_box"], atol=1e-9)
    for row in g["nodes"]:
        n = scene.nodes[row[0]]
        if row[1] == "I":
            assert not n["is_leaf"] and n["second_child_offset"] == row[2]
        else:
            assert n["is_leaf"] and (n["start"], n["end"]) == (row[2], row[3])
    tri = scene.prims["triangle"]
    assert [i for i in range(36) if tri["lightIdx"][i] >= 0] == g["light_prims"]
    assert all(tri["matIdx"][i] == g["light_matIdx"] for i in g["light_prims"])
    assert [i for i in range(36) if tri["matIdx"][i] == 0] == g["matIdx_0_prims"]
    assert [i for i in range(36) if tri["matIdx"][i] == 1] == g["matIdx_1_prims"]
    assert scene.cdf.tolist() == g["light_cdf"]
    # every primitive reachable exactly once
    seen = []
    for n in scene.nodes:
        if n["is_leaf"]:
            seen += list(range(n["start"], n["end"] + 1))
    assert sorted(seen) == list(range(36))


@pytest.mark.parametrize("case", GOLD["radiance_clang_nofma"], ids=lambda c: f"spp{c['spp']}")
def test_cornell_radiance_matches_reference_values(case):
    lib = ol.load("libm")
    scene, meta = ol.load_cornell(case["depth"], lib)
    cam = ol.cornell_camera(meta, 512, 512, lib)
    acc, _ = ol.render(scene, cam, 512, 512, meta["epsilon"], 1, case["spp"], kind="libm")
    img = acc.reshape(-1, 3) / np.float32(case["spp"])
    for p, rgb in case["pixels"].items():
        assert [nine(x) for x in img[int(p)]] == rgb, f"pixel {p}"
    mean = img.astype(np.float64).mean(0)
    assert [nine(x) for x in mean] == case["mean"]


@pytest.mark.parametrize("row,depth,W,H", [("cornell_depth4_512", 4, 512, 512), ("cornell_depth8_512", 8, 512, 512),
                                            ("cornell_depth8_1080p", 8, 1920, 1080)])
def test_work_counters_match_survey(row, depth, W, H):
    """B_alg inputs: bounce iterations, closest-hit rays, shadow rays, node visits and primitive
    tests per sample, at depth 4 and at depth 8 (square and the 1080p framing of config 2).  The survey counted with a g++
    build (right-to-left draws), so agreement is statistical (< 0.5 %), not exact."""
    lib = ol.load("libm")
    scene, meta = ol.load_cornell(depth, lib)
    cam = ol.cornell_camera(meta, W, H, lib)
    ol.render(scene, cam, W, H, meta["epsilon"], 1, 2, kind="libm")
    c = ol.counters("libm")
    g = GOLD["work_per_sample_gxx"][row]
    s = c["samples"]
    assert s == W * H * 2
    for key, name in (("bounce", "bounce_iters"), ("closest", "closest_rays"), ("shadow", "shadow_rays"),
                      ("node", "node_visits"), ("prim", "prim_tests")):
        assert abs(c[name] / s - g[key]) / g[key] < 5e-3, key


def test_depth8_1024spp_frame_mean_matches_reference_value():
    """The survey's longest run of the reference's own code: Cornell 256 x 256, 1024 spp, depth 8 (north_star's sample count
    and depth), frame mean of the linear radiance printed to nine digits.  The libm build of the
    oracle reproduces all nine digits of all three channels - 67 M samples, every bounce, light sample, MIS weight and
    roulette decision of depth-8 paths included.  The soft-math build (the GPU's bit-exact partner) differs from it only by
    last-bit roundings of sin / cos: per-channel relative RMS of the two films ~1e-6, bar 1e-4."""
    lib = ol.load("libm")
    scene, meta = ol.load_cornell(8, lib)
    cam = ol.cornell_camera(meta, 256, 256, lib)
    threads = min(32, os.cpu_count() or 1)
    acc, _ = ol.render(scene, cam, 256, 256, meta["epsilon"], 1, 1024, kind="libm", threads=threads)
    img = acc.reshape(-1, 3) / np.float32(1024)
    assert [nine(x) for x in img.astype(np.float64).mean(0)] == GOLD["radiance_256_1024spp_depth8_mean"]
    soft, _ = ol.render(scene, cam, 256, 256, meta["epsilon"], 1, 1024, kind="soft", threads=threads)
    a, b = soft.reshape(-1, 3).astype(np.float64), acc.reshape(-1, 3).astype(np.float64)
    rms = np.sqrt(((a - b) ** 2).mean(0)) / np.sqrt((b ** 2).mean(0))
    print("soft vs libm, 256x256 1024 spp depth 8: relative RMS per channel", rms)
    assert (rms <= 1e-4).all(), rms


def test_soft_and_libm_builds_agree_statistically():
    """Same algorithm, different last-bit rounding of sin/cos: images differ only in rare pixels."""
    scene, meta = ol.load_cornell(4)
    cam = ol.cornell_camera(meta, 128, 128)
    a, _ = ol.render(scene, cam, 128, 128, 0.001, 1, 16, kind="soft")
    b, _ = ol.render(scene, cam, 128, 128, 0.001, 1, 16, kind="libm")
    rms = np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2))
    assert rms < 1e-3
    assert np.count_nonzero(a != b) < 0.05 * a.size


def test_render_is_deterministic_and_thread_independent():
    scene, meta = ol.load_cornell(8)
    cam = ol.cornell_camera(meta, 64, 64)
    a, _ = ol.render(scene, cam, 64, 64, 0.001, 1, 8, threads=1)
    b, _ = ol.render(scene, cam, 64, 64, 0.001, 1, 8, threads=8)
    assert a.tobytes() == b.tobytes()


def test_batched_iterations_equal_single_calls():
    """iter_count=N in one call == N reference-style calls with iter = 1..N."""
    scene, meta = ol.load_cornell(5)
    cam = ol.cornell_camera(meta, 64, 64)
    a, ca = ol.render(scene, cam, 64, 64, 0.001, 1, 6)
    acc = np.zeros(64 * 64 * 3, np.float32)
    col = np.zeros(64 * 64 * 3, np.float32)
    for it in range(1, 7):
        ol.render(scene, cam, 64, 64, 0.001, it, 1, reset=(it == 1), acc=acc, color=col)
    assert a.tobytes() == acc.tobytes() and ca.tobytes() == col.tobytes()


def test_tile_ownership_partitions_the_frame():
    scene, meta = ol.load_cornell(4)
    W, H = 96, 64
    cam = ol.cornell_camera(meta, W, H)
    full, _ = ol.render(scene, cam, W, H, 0.001, 1, 3)
    parts = [ol.render(scene, cam, W, H, 0.001, 1, 3, rank=r, n_ranks=3)[0] for r in range(3)]
    s = parts[0] + parts[1] + parts[2]
    assert s.tobytes() == full.tobytes()
    # supports are disjoint
    nz = [(p.reshape(-1, 3) != 0).any(1) for p in parts]
    assert not (nz[0] & nz[1]).any() and not (nz[0] & nz[2]).any() and not (nz[1] & nz[2]).any()


def test_ambient_occlusion_oracle_properties():
    """Ao (pathtracer.cu:830-876) has no golden values in the reference (parity unpinned): check what the
    algorithm implies.  An unoccluded sample is cos*(1/pi)/(cos/pi) = 1 up to rounding; with a tiny maxDist
    everything the camera sees is open; a long one darkens the box; misses are 0; thread count does not matter."""
    scene, meta = ol.load_cornell(4)
    W, H = 96, 64
    cam = ol.cornell_camera(meta, W, H)
    scene.desc.set_integrator("ao", 1e-3)
    a, _ = ol.render(scene, cam, W, H, 0.001, 1, 4, kind="soft")
    img = a.reshape(H, W, 3) / np.float32(4)
    hit = img[..., 0] > 0
    assert hit.any() and (~hit).any()                         # 96x64 framing: the box in the middle, nothing beside it
    per_sample = a.reshape(H, W, 3)[..., 0]                    # every sample is 0 (miss) or 1 (open), so sums are ~integers
    assert np.abs(per_sample - np.round(per_sample)).max() < 4e-6
    assert (np.round(per_sample) == 4).sum() > 0.5 * hit.sum()
    scene.desc.set_integrator("ao", 10.0)
    b1, _ = ol.render(scene, cam, W, H, 0.001, 1, 4, kind="soft", threads=1)
    b8, _ = ol.render(scene, cam, W, H, 0.001, 1, 4, kind="soft", threads=8)
    assert b1.tobytes() == b8.tobytes()
    hit = np.round(per_sample) == 4
    inside = (b1.reshape(H, W, 3) / np.float32(4))[hit]
    assert inside.mean() < 0.25                                  # a closed box: almost every 10-unit ray is blocked
    scene.desc.set_integrator("ao", 0.5)
    c_soft, _ = ol.render(scene, cam, W, H, 0.001, 1, 4, kind="soft")
    c_libm, _ = ol.render(scene, cam, W, H, 0.001, 1, 4, kind="libm")
    assert np.abs(c_soft - c_libm).max() <= 1.0 + 1e-6 and np.mean(c_soft != c_libm) < 1e-3
    assert 0.25 < (c_soft.reshape(H, W, 3) / np.float32(4))[hit].mean() < 1.0


def test_traversal_operators_do_not_depend_on_the_order():
    """Intersect / IntersectP as operators (oracle_trace_rays) on 60 000 rays with edge cases (axis-aligned directions, origins on
    the walls, short / zero intervals): whether an any-hit ray is blocked is the same in both orders, and the wide walk finds
    the reference order's closest hit - primitive, t, b1, b2 - on every ray with a proper direction, exactly equal hits included."""
    import scenes
    import test_gpu_parity as tg
    rays = tg.operator_rays(60_000, 23)
    proper = ~np.isnan(rays).any(axis=1) & (np.abs(rays[:, 3:6]).sum(axis=1) > 0)
    closest, anyhit = proper & (rays[:, 7] == 0), proper & (rays[:, 7] != 0)
    for scene in (ol.load_cornell(4)[0], scenes.zoo_scene(max_depth=4, extra=scenes.random_soup(2000, 5, size=0.3))[0]):
        res = {order: ol.trace_rays(scene, 0.001, rays, order, threads=8) for order in (0, 2)}
        assert 0.3 < (res[0][0] >= 0).mean() < 0.999
        assert np.array_equal(res[2][0][anyhit] >= 0, res[0][0][anyhit] >= 0)
        assert np.array_equal(res[2][0][closest], res[0][0][closest])
        assert res[2][1][closest].tobytes() == res[0][1][closest].tobytes()
        hit = res[0][0] >= 0
        assert (res[0][1][hit & proper & ~np.isnan(rays[:, 6]), 0] >= 0.001).all()             # tmin = epsilon


def test_wide_traversal_agrees_with_the_reference_order():
    """include/gpt_wide_bvh.h: the 4-wide tree collapsed from the reference's tree, walked nearest child first with an
    order-free rule for equal distances (the larger primitive index).  Same boxes, same box and triangle arithmetic: the film
    is the reference order's (bar 1e-4 relative RMS; measured: every float equal), with a quarter of the node visits.  The
    structure itself: every primitive sits in exactly one leaf, leaves hold at most 16, children keep the reference's order."""
    import ctypes as C
    import scenes
    lib = ol.load("soft")
    for scene, meta, W, H, spp in ((ol.load_cornell(8) + (96, 96, 8)), (scenes.zoo_scene(max_depth=8, with_env=True) + (96, 72, 6)),
                                   (scenes.stress_scene(0.25, max_depth=12) + (96, 72, 4))):
        cam = ol.cornell_camera(meta, W, H)
        ref, _ = ol.render(scene, cam, W, H, 0.001, 1, spp, kind="soft", order=0)
        c_ref = ol.counters("soft")
        auto, _ = ol.render(scene, cam, W, H, 0.001, 1, spp, kind="soft")          # the product's rule: wide unless the scene fits LDS
        small = len(scene.prims) <= 40
        assert (ol.counters("soft")["node_visits"] == c_ref["node_visits"]) == small and auto.tobytes() == ref.tobytes()
        wide, _ = ol.render(scene, cam, W, H, 0.001, 1, spp, kind="soft", order=2)
        c_wide = ol.counters("soft")
        again, _ = ol.render(scene, cam, W, H, 0.001, 1, spp, kind="soft", threads=1, order=2)
        assert wide.tobytes() == again.tobytes()
        a, b = wide.reshape(-1, 3).astype(np.float64), ref.reshape(-1, 3).astype(np.float64)
        rms = np.sqrt(((a - b) ** 2).mean(0)) / np.sqrt((b ** 2).mean(0))
        assert (rms <= 1e-4).all()
        assert np.count_nonzero(wide != ref) == 0          # stronger than the bar, and what has been observed on every scene so far
        assert c_wide["closest_rays"] == c_ref["closest_rays"] and c_wide["shadow_rays"] == c_ref["shadow_rays"]
        assert c_wide["node_visits"] * 3 < c_ref["node_visits"]
    assert lib.oracle_set_traversal(7) == -1 and lib.oracle_set_traversal(1) == -1


def test_volpath_oracle_properties():
    """Volpath (pathtracer.cu:1025-1242) with homogeneous media.  No reference output exists for it (parity unpinned).
    Checked here: without media it IS Path (same draws, same arithmetic); fog between camera and scene dims the film;
    the thread count does not matter; a density-grid record without a grid is refused."""
    from gpu_pathtracer_amd import scene_types as st
    scene, meta = ol.load_cornell(6)
    W, H = 64, 64
    cam = ol.cornell_camera(meta, W, H)
    pt, _ = ol.render(scene, cam, W, H, 0.001, 1, 8, kind="soft")
    scene.desc.set_integrator("vpt", 6)
    vpt, _ = ol.render(scene, cam, W, H, 0.001, 1, 8, kind="soft")
    assert vpt.tobytes() == pt.tobytes()
    # fog everywhere the camera looks
    scene.set_mediums([st.make_medium((0.0014, 0.0025, 0.0142), (0.70, 1.22, 1.90), 0.0, 0.25),
                       st.make_medium((0.05, 0.05, 0.05), (0.4, 0.4, 0.4), 0.7, 1.0)])
    for med in (0, 1):
        cam.medium = med
        f1, _ = ol.render(scene, cam, W, H, 0.001, 1, 8, kind="soft", threads=1)
        f8, _ = ol.render(scene, cam, W, H, 0.001, 1, 8, kind="soft", threads=8)
        assert f1.tobytes() == f8.tobytes() and np.isfinite(f1).all()
        assert 0 < f1.mean() < pt.mean()                       # extinction on the way to the camera
    # pure absorption, depth 1, looking straight at the light from below: Li = exp(-sigmaT * t) * Le per sample
    scene.set_mediums([st.make_medium((0.3, 0.2, 0.1), (0.0, 0.0, 0.0), 0.0, 1.0)])
    scene.desc.set_integrator("vpt", 1)
    up = ol.make_camera((0.0, 0.5, 0.0), (0.0, 2.0, 0.0), (0, 0, 1), (32, 32), 4.0)
    up.medium = 0
    a, _ = ol.render(scene, up, 32, 32, 0.001, 1, 1, kind="soft")
    scene.desc.set_integrator("pt", 1)
    up.medium = -1
    b, _ = ol.render(scene, up, 32, 32, 0.001, 1, 1, kind="soft")
    a, b = a.reshape(-1, 3), b.reshape(-1, 3)
    lit = b[:, 0] > 0
    assert lit.sum() > 900
    # sigmaS = 0: the distance sample never lands inside (pdf sigma*exp(-sigma d), weight sigmaT*Tr/pdf = Tr*sigmaT/(sigma*exp(-sigma*d)))
    # - the reference's estimator for the un-scattered case, so only the ratio's sign and finiteness are asserted here
    assert np.isfinite(a).all() and (a[lit] >= 0).all()
    # a heterogeneous record needs its grid and a positive iterMax
    het = st.make_medium((1, 1, 1), (1, 1, 1))
    het["type"] = 1
    scene.set_mediums([het])
    scene.desc.set_integrator("vpt", 4)
    n = W * H * 3
    rc = ol.load("soft").oracle_render(C.byref(scene.desc), C.byref(cam), W, H, C.c_float(0.001), 1, 1, 1,
                                       st.ptr(np.zeros(n, np.float32)), st.ptr(np.zeros(n, np.float32)), None, 0, 1, 1)
    assert rc == -2


def test_volpath_oracle_density_grids_and_interfaces():
    """Density grids (medium.h:53-182) behind a material-less surface (pathtracer.cu:1117-1124).  A camera in vacuum
    looks straight up at the light through an absorbing grid of constant density rho that starts at a material-less
    sheet and contains the light: the path ray passes the sheet without a bounce, Sample() survives the stretch L to
    the light with probability T = exp(-sigma rho L), and the emitter is then attenuated by the Tr() estimate of the
    same stretch, whose mean is T for all three estimators - so the film's mean is T^2 Le."""
    from gpu_pathtracer_amd import scene_types as st
    rho, sigma = np.float32(0.7), 0.8
    grid = np.full((4, 4, 4), rho, np.float32)
    sheet = scenes.box_mesh((-0.99, 0.9, -0.99), (0.99, 2.5, 0.99), -1, inside=0, outside=-1)[4:6]     # its bottom face
    assert np.allclose([sheet["triangle"][v]["v"]["y"] for v in ("v1", "v2", "v3")], 0.9)
    scene, meta = scenes.zoo_scene(max_depth=1, extra=sheet, assign={})
    W = H = 32
    up = ol.make_camera((0.0, 0.5, 0.0), (0.0, 2.0, 0.0), (0, 0, 1), (W, H), 4.0)
    scene.desc.set_integrator("vpt", 1)
    tri = scene.prims["triangle"]
    is_sheet = tri["matIdx"] == -1
    assert is_sheet.sum() == 2
    tri["mediumInside"][is_sheet] = -1
    vac, _ = ol.render(scene, up, W, H, 0.001, 1, 1, kind="soft")            # no media: the sheet alone changes nothing
    tri["mediumInside"][is_sheet] = 0
    lit = vac.reshape(-1, 3)[:, 0] > 0
    assert lit.sum() > 900
    light_y = float(tri["v1"]["v"]["y"][tri["lightIdx"] >= 0][0])
    T = np.exp(-sigma * float(rho) * (light_y - 0.9))
    spp = 24
    for tr_type in (0, 1, 2):
        het = st.make_het_medium((sigma, sigma, sigma), (0, 0, 0), grid, (-2, -1, -2), (2, 3, 2), 1000, tr_type)
        scene.set_mediums([het], keep=[grid])
        f1, _ = ol.render(scene, up, W, H, 0.001, 1, spp, kind="soft", threads=1)
        f8, _ = ol.render(scene, up, W, H, 0.001, 1, spp, kind="soft", threads=8)
        assert f1.tobytes() == f8.tobytes() and np.isfinite(f1).all()
        ratio = f1.reshape(-1, 3)[lit].mean(axis=0) / (spp * vac.reshape(-1, 3)[lit].mean(axis=0))
        assert np.allclose(ratio, T * T, rtol=0.04), (tr_type, ratio, T * T)


# result/cornell_dof.png: the Cornell box with both boxes - the scene of BASELINE configs 1 and 2, every mesh of which ships -
# through the thin-lens camera.  Its scene file is not in the repository; the shipped scene.json carries "focalDistance": 7.0
# beside "apertureRadius": 0.0, and a GPU sweep over (maxDepth, focalDistance, apertureRadius) against the picture
# (tools/gpu_fit_cornell_dof.py, profiles/r02/cornell_dof_fit.txt) has a sharp minimum at focalDistance 7.0, apertureRadius
# 0.5 - the json's own focal distance and a round aperture - for every depth >= 7.  The picture's last six pixel columns
# are black (a window-capture artefact), so the last block column is left out.
DOF = {"aperture": 0.5, "focal": 7.0, "depth": 8}


def dof_compare(acc, spp):
    import refimg
    want = refimg.load("reference_cornell_dof_64.npy")[:, :63]
    got = refimg.blocks(refimg.filmic_png(acc, spp, 512, 512))[:, :63]
    d = np.abs(got - want)
    return float(np.abs(got.mean(axis=(0, 1)) - want.mean(axis=(0, 1))).max()), float(d.mean()), float(d.max())


def dof_camera(meta, aperture=None, focal=None):
    c = meta["camera"]
    return ol.make_camera(c["position"], c["lookat"], c["up"], (512, 512), c["fov"], DOF["aperture"] if aperture is None else aperture,
                          DOF["focal"] if focal is None else focal, c["distance"], c["filmic"])


def test_oracle_reproduces_the_reference_depth_of_field_render():
    """Pin for Path on the Cornell box of configs 1 / 2 (BVH, triangles, lambertian BSDF, area light, MIS, roulette), the
    thin-lens camera, the filmic curve and the PNG conventions against a picture the reference's author rendered: 16 spp of
    the oracle land on result/cornell_dof.png within Monte-Carlo noise (the GPU test runs 4096 spp with tight bounds)."""
    scene, meta = ol.load_cornell(DOF["depth"])
    spp = 16
    acc, _ = ol.render(scene, dof_camera(meta), 512, 512, meta["epsilon"], 1, spp, kind="soft")
    m, bm, bx = dof_compare(acc, spp)
    print("cornell_dof: frame mean diff", m, "block mean", bm, "block max", bx)
    # (at 16 spp the concave tone curve alone pulls the mean of a noisy image down by a few thousandths)
    assert m < 0.008 and bm < 0.008 and bx < 0.12, (m, bm, bx)
    pin, _ = ol.render(scene, dof_camera(meta, aperture=0.0), 512, 512, meta["epsilon"], 1, spp, kind="soft")
    assert dof_compare(pin, spp)[1] > 0.012          # the pinhole camera does not: the blur is the lens, not noise


def test_oracle_reproduces_the_reference_smoke_render(tmp_path):
    """Pin for Volpath, density-grid media, material-less surfaces, the loader and the filmic tonemap: the reference
    publishes a render of its default scene (result/heterogeneous.png = scenes/cornell_box/scene.json as shipped, density.d
    included), kept here box-filtered to 64 x 64 (tests/golden/reference_heterogeneous_64.npy,
    tools/make_reference_image_fixture.py).  The oracle renders the same scene file at 512 x 512 with 16 samples per
    pixel, applies Output's filmic curve (pathtracer.cu:199-204,2516-2531) and the PNG writer's flip and 8-bit truncation
    (imageio.cpp:61-78), and has to land on the same picture up to Monte-Carlo noise: each colour channel's frame mean
    within 0.008 (of 1; at 16 spp the concave tone curve alone pulls the mean of a noisy image down by ~0.003 against the
    converged reference), the mean absolute difference of the 64 x 64 blocks below 0.008, no block further off than 0.08.
    (A wrong phase function, tracking estimator, medium switch at the interface or light sampling shifts the frame mean
    by several times that: e.g. dropping the medium entirely gives 0.46 / 0.36 / 0.20 against 0.416 / 0.305 / 0.114.)"""
    from gpu_pathtracer_amd import api
    want = np.load(os.path.join(ol.ROOT, "tests", "golden", "reference_heterogeneous_64.npy")).astype(np.float64)
    # the shipped scene, rebuilt from this repository's fixtures (bit-identical to loading the shipped file:
    # tests/test_scene_loader.py::test_rebuilt_smoke_scene_is_the_shipped_scene), so that this runs anywhere
    ls = api.LoadedScene(scenes.write_smoke_scene(str(tmp_path / "smoke")))
    W, H = ls.width, ls.height
    assert (W, H) == (512, 512) and ls.desc.integrator_type == st.IT_VPT and ls.desc.max_depth == 17
    cam = ol.make_camera((0, 1.0, 6.8), (0, 1.0, 0), (0, 1, 0), (W, H), 19.5, 0.0, 7.0)      # the json's camera block
    cam.medium = ls.camera.medium
    spp = 16
    acc, _ = ol.render(ls, cam, W, H, ls.epsilon, 1, spp, kind="soft")
    lin = acc.reshape(H, W, 3).astype(np.float64) / spp
    c = np.maximum(0.0, lin - 0.004)
    img = (c * (6.2 * c + 0.5)) / (c * (6.2 * c + 1.7) + 0.06)
    img = np.floor(np.clip(img, 0.0, 1.0) * 255.0) / 255.0
    got = img[::-1].reshape(64, 8, 64, 8, 3).mean(axis=(1, 3))
    d = np.abs(got - want)
    assert np.abs(got.mean(axis=(0, 1)) - want.mean(axis=(0, 1))).max() < 0.008, (got.mean(axis=(0, 1)), want.mean(axis=(0, 1)))
    print("frame means", got.mean(axis=(0, 1)), want.mean(axis=(0, 1)), "block diff mean / max", d.mean(), d.max())
    assert d.mean() < 0.008 and d.max() < 0.08, (d.mean(), d.max())
    ls.close()

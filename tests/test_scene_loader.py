"""C++ scene loader (JSON + OBJ + PNG/PFM + bvh.cache) — host code only, no GPU."""
import json
import os
import shutil
import struct
import zlib

import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
import scenes
from gpu_pathtracer_amd import api, scene_types as st

SCENE = os.path.join(ol.ROOT, "scenes", "cornell_pt", "scene.json")


def tri_fields_equal(a, b):
    ok = True
    for v in ("v1", "v2", "v3"):
        for f in ("v", "n"):
            for c in "xyz":
                ok &= np.array_equal(a[v][f][c].view(np.uint32), b[v][f][c].view(np.uint32))
        ok &= np.array_equal(a[v]["uv"], b[v]["uv"])
    for f in ("matIdx", "lightIdx", "bssrdfIdx", "mediumInside", "mediumOutside"):
        ok &= np.array_equal(a[f], b[f])
    return bool(ok)


def test_cornell_scene_file_equals_baked_fixture():
    ls = api.LoadedScene(SCENE)
    scene, meta = ol.load_cornell(8)
    d = ls.desc
    assert (d.n_prims, d.n_nodes, d.n_materials, d.n_lights, d.n_light_distribution) == (36, 27, 8, 2, 3)
    assert d.integrator_type == st.IT_PT and d.max_depth == 8
    assert tri_fields_equal(ls.array("prims", "n_prims", st.PRIMITIVE)["triangle"], scene.prims["triangle"])
    assert ls.array("materials", "n_materials", st.MATERIAL).tobytes() == scene.materials.tobytes()
    nodes = ls.array("nodes", "n_nodes", st.BVH_NODE)
    assert all(np.array_equal(nodes[f], scene.nodes[f]) for f in nodes.dtype.names)
    assert ls.array("light_distribution", "n_light_distribution", np.float32).tolist() == [0.0, 0.5, 1.0]
    lights = ls.array("lights", "n_lights", st.AREA)
    assert tri_fields_equal(lights["triangle"], scene.lights["triangle"])
    assert [float(lights["radiance"][0][c]) for c in "xyz"] == [17.0, 12.0, 4.0]
    assert (ls.width, ls.height) == (512, 512) and abs(ls.epsilon - 0.001) < 1e-9
    assert bytes(ls.camera) == bytes(ol.cornell_camera(meta, 512, 512))


def write_png(path, img):
    """8-bit RGBA PNG, rows top-down (test helper: zlib from the standard library)"""
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


@pytest.fixture()
def scene_dir(tmp_path):
    d = tmp_path / "scene"
    shutil.copytree(os.path.dirname(SCENE), d)
    return d


def test_transform_defaults_textures_env_and_materials(scene_dir):
    rng = np.random.default_rng(5)
    tex = rng.integers(0, 256, (6, 5, 4), dtype=np.uint8)
    write_png(scene_dir / "checker.png", tex)
    env = scenes.sky_env(16, 8)
    # PFM is bottom-up on disk; the loader hands rows top-down like the reference's EXR reader
    api.save_pfm(str(scene_dir / "sky.pfm"), 16, 8, env[::-1].copy())
    js = json.load(open(scene_dir / "scene.json"))
    js["material"] += [
        {"name": "tex", "bsdf": "substrate", "diffuse": "checker.png", "alpha": 0.1, "remap": True, "specular": [0.04, 0.04, 0.04]},
        {"name": "aniso", "bsdf": "roughconduct", "alphaU": 0.0025, "alphaV": 0.25, "remap": True, "eta": [0.2, 0.9, 1.1], "k": [3.9, 2.4, 2.1]},
        {"name": "unknown-bsdf", "bsdf": "velvet"},
    ]
    js["scene"][5]["material"] = "tex"
    js["scene"][6].update({"material": "aniso", "scale": [0.5, 0.5, 0.5], "translate": [0.1, 0.0, -0.2], "rotate": [0, 30, 10]})
    js["light"].append({"infinite": "sky.pfm", "rotate": [0, 30, 0]})
    js.pop("screen_width")
    json.dump(js, open(scene_dir / "scene.json", "w"))
    ls = api.LoadedScene(str(scene_dir / "scene.json"))
    d = ls.desc
    assert (ls.width, ls.height) == (512, 512)             # default when either size key is missing
    mats = ls.array("materials", "n_materials", st.MATERIAL)
    assert d.n_materials == 11 and d.n_textures == 1
    assert mats[8]["type"] == st.MT_SUBSTRATE and mats[8]["textureIdx"] == 0
    x = np.log(np.float32(0.1))
    remap = np.float32(1.62142) + np.float32(0.819955) * x + np.float32(0.1734) * x * x + np.float32(0.0171201) * x * x * x + np.float32(0.000640711) * x * x * x * x
    assert abs(float(mats[8]["alphaU"]) - float(remap)) < 1e-6 and mats[8]["alphaU"] == mats[8]["alphaV"]
    assert mats[9]["alphaU"] != mats[9]["alphaV"]
    assert mats[10]["type"] == st.MT_LAMBERTIAN               # unknown bsdf name -> 0, as std::map::operator[]
    # texture: flipped vertically, sRGB->linear pow 2.2, truncated to 8 bit (reference imageio.cpp:11-59, texture.h:21-25)
    import ctypes as C
    trec = C.cast(d.textures, C.POINTER(st.Texture))[0]
    assert (trec.width, trec.height) == (5, 6)
    got = np.ctypeslib.as_array(C.cast(trec.data, C.POINTER(C.c_uint8)), shape=(6, 5, 4))
    lin = np.power((tex[::-1].astype(np.float32) * np.float32(1 / 255.0))[..., :3], np.float32(2.2), dtype=np.float32)
    want = np.concatenate([(lin * np.float32(255)).astype(np.uint8), tex[::-1][..., 3:]], -1)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    # transformed mesh: the tall box is scaled/rotated/translated; normals stay unit
    prims = ls.array("prims", "n_prims", st.PRIMITIVE)
    moved = prims[prims["triangle"]["matIdx"] == 9]["triangle"]
    assert len(moved) == 12
    n = np.stack([moved["v1"]["n"][c] for c in "xyz"], -1)
    assert np.allclose((n ** 2).sum(-1), 1.0, atol=1e-6)
    ys = np.concatenate([moved[v]["v"]["y"] for v in ("v1", "v2", "v3")])
    assert abs(ys.max() - 0.6 * 1.0) < 0.2 and ys.min() > -0.2
    # env light: cdf gets one more entry, bounding sphere from the scene box, u/v/w from "rotate"
    assert d.n_light_distribution == 4
    inf = C.cast(d.infinite, C.POINTER(st.Infinite))[0]
    assert inf.isvalid and (inf.width, inf.height) == (16, 8) and inf.radius > 1.0
    assert abs(inf.u.x - np.cos(np.pi / 6)) < 1e-6 and abs(inf.v.y - 1) < 1e-6
    data = np.ctypeslib.as_array(C.cast(inf.data, C.POINTER(C.c_float)), shape=(8, 16, 3))
    assert np.array_equal(data, env)


def write_exr(path, img, compression, pixel_type):
    """Minimal scanline OpenEXR writer (test helper, written from the file-format description):
    img (H, W, 3) float32, rows top-down; compression 0 NONE, 1 RLE, 2 ZIPS, 3 ZIP; pixel_type 1 HALF, 2 FLOAT."""
    h, w, _ = img.shape
    names = ["B", "G", "R"]                      # channels are stored in alphabetical order
    chan = {"R": img[..., 0], "G": img[..., 1], "B": img[..., 2]}
    dt = np.float16 if pixel_type == 1 else np.float32

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", pixel_type, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = (attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([compression])) +
              attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) +
              attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
              attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) +
              attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    lines_per = 16 if compression == 3 else 1
    chunks = []
    for y0 in range(0, h, lines_per):
        raw = b"".join(chan[n][y].astype(dt).tobytes() for y in range(y0, min(h, y0 + lines_per)) for n in names)
        data = raw
        if compression:
            b = np.frombuffer(raw, np.uint8)
            half = (len(b) + 1) // 2
            t = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)        # split even/odd bytes
            assert len(t[:half]) == half
            p = t.copy()
            p[1:] = (t[1:] - t[:-1] + 128 + 256) % 256                     # predictor
            pre = p.astype(np.uint8).tobytes()
            if compression == 1:                                           # RLE: runs of 3..127 equal bytes, else literals
                out, i = bytearray(), 0
                while i < len(pre):
                    run = 1
                    while i + run < len(pre) and pre[i + run] == pre[i] and run < 127:
                        run += 1
                    if run >= 3:
                        out += bytes([run - 1, pre[i]])
                        i += run
                    else:
                        j = i
                        while j < len(pre) and j - i < 127 and not (j + 2 < len(pre) and pre[j] == pre[j + 1] == pre[j + 2]):
                            j += 1
                        j = max(j, i + 1)
                        out += bytes([(256 - (j - i)) & 0xff]) + pre[i:j]
                        i = j
                comp = bytes(out)
            else:
                comp = zlib.compress(pre, 6)
            data = comp if len(comp) < len(raw) else raw
        chunks.append(struct.pack("<ii", y0, len(data)) + data)
    head = struct.pack("<ii", 20000630, 2) + header
    off = len(head) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    open(path, "wb").write(head + table + b"".join(chunks))


@pytest.mark.parametrize("compression,pixel_type", [(0, 2), (0, 1), (1, 1), (2, 2), (3, 1), (3, 2)])
def test_exr_environment_map(scene_dir, compression, pixel_type):
    """"infinite": "<file>.exr" (reference src/parsescene.cpp:543-550 -> ImageIO::LoadExr): float RGB, row 0 = top."""
    import ctypes as C
    env = scenes.sky_env(40, 21)                  # 21 rows: the last 16-line ZIP block is short
    if pixel_type == 1:
        env = env.astype(np.float16).astype(np.float32)
    write_exr(str(scene_dir / "sky.exr"), env, compression, pixel_type)
    js = json.load(open(scene_dir / "scene.json"))
    js["light"].append({"infinite": "sky.exr"})
    json.dump(js, open(scene_dir / "scene.json", "w"))
    ls = api.LoadedScene(str(scene_dir / "scene.json"))
    inf = C.cast(ls.desc.infinite, C.POINTER(st.Infinite))[0]
    assert inf.isvalid and (inf.width, inf.height) == (40, 21)
    data = np.ctypeslib.as_array(C.cast(inf.data, C.POINTER(C.c_float)), shape=(21, 40, 3))
    assert np.array_equal(data, env)
    assert (inf.u.x, inf.v.y, inf.w.z) == (1.0, 1.0, 1.0)      # no "rotate": identity axes (documented deviation)


@pytest.mark.parametrize("subsampling,quality,progressive", [(0, 95, False), (1, 90, False), (2, 85, False),
                                                              (0, 95, True), (1, 90, True), (2, 60, True)])
def test_jpeg_texture(scene_dir, subsampling, quality, progressive):
    """JPEG textures (the reference's sponza uses .jpg), baseline and progressive (libjpeg's default scan script:
    interleaved DC, spectral bands, successive-approximation refinements): decoded within a few 8-bit steps of an
    independent decoder (PIL); exact agreement with stb_image's fixed-point IDCT is not claimed.  The progressive
    file must decode to exactly what the baseline file of the same coefficients decodes to."""
    import ctypes as C
    from PIL import Image
    yy, xx = np.mgrid[0:37, 0:50]
    img = np.stack([(xx * 5) % 256, (yy * 7) % 256, ((xx + yy) * 3) % 256], -1).astype(np.uint8)
    img = (img // 2 + 60).astype(np.uint8)
    Image.fromarray(img).save(scene_dir / "tex.jpg", quality=quality, subsampling=subsampling, progressive=progressive)
    if progressive:
        assert b"\xff\xc2" in (scene_dir / "tex.jpg").read_bytes()          # really an SOF2 file
        Image.fromarray(img).save(scene_dir / "tex_b.jpg", quality=quality, subsampling=subsampling, progressive=False)
    ref = np.asarray(Image.open(scene_dir / "tex.jpg").convert("RGB")).astype(np.float32)
    js = json.load(open(scene_dir / "scene.json"))
    js["material"].append({"name": "jpg", "bsdf": "lambertian", "diffuse": "tex.jpg"})
    if progressive:
        js["material"].append({"name": "jpg_b", "bsdf": "lambertian", "diffuse": "tex_b.jpg"})
    json.dump(js, open(scene_dir / "scene.json", "w"))
    ls = api.LoadedScene(str(scene_dir / "scene.json"))
    trec = C.cast(ls.desc.textures, C.POINTER(st.Texture))[0]
    assert (trec.width, trec.height) == (50, 37)
    got = np.ctypeslib.as_array(C.cast(trec.data, C.POINTER(C.c_uint8)), shape=(37, 50, 4))[..., :3].astype(np.float32)
    if progressive:                                                       # same coefficients, scan order only differs
        tb = C.cast(ls.desc.textures, C.POINTER(st.Texture))[1]
        base = np.ctypeslib.as_array(C.cast(tb.data, C.POINTER(C.c_uint8)), shape=(37, 50, 4))[..., :3]
        assert (base == got).all()
    want = np.floor(np.power(ref[::-1] / 255.0, 2.2) * 255.0)       # flip + sRGB->linear + truncate, as LoadTexture
    tol = 6 if subsampling == 0 else 40                               # chroma replication vs PIL's smooth upsampling
    assert np.abs(got - want).mean() < (1.0 if subsampling == 0 else 3.0)
    assert np.abs(got - want).max() <= tol


def test_jpeg_progressive_with_restart_markers_and_grayscale(scene_dir):
    """Progressive scans with restart intervals (EOB runs and DC predictors reset at every RSTn) and one-component
    files: the progressive file decodes to exactly the texels of the baseline file with the same coefficients."""
    import ctypes as C
    from PIL import Image
    rng = np.random.default_rng(1)
    yy, xx = np.mgrid[0:131, 0:203]
    img = np.stack([(xx * 3 + yy) % 256, (yy * 2) % 256, (xx ^ yy) % 256], -1).astype(np.uint8)
    img = (img * 0.6 + rng.integers(0, 60, img.shape)).astype(np.uint8)
    cases = [("p_rst.jpg", dict(quality=80, subsampling=2, progressive=True, restart_marker_blocks=3)),
             ("b_rst.jpg", dict(quality=80, subsampling=2, progressive=False, restart_marker_blocks=3)),
             ("p_gray.jpg", dict(quality=70, progressive=True)), ("b_gray.jpg", dict(quality=70, progressive=False))]
    js = json.load(open(scene_dir / "scene.json"))
    for name, kw in cases:
        Image.fromarray(img if "gray" not in name else img[..., 0]).save(scene_dir / name, **kw)
        js["material"].append({"name": name, "bsdf": "lambertian", "diffuse": name})
    json.dump(js, open(scene_dir / "scene.json", "w"))
    assert (scene_dir / "p_rst.jpg").read_bytes().count(b"\xff\xd0") > 20
    ls = api.LoadedScene(str(scene_dir / "scene.json"))
    tex = []
    for i, (name, _) in enumerate(cases):
        t = C.cast(ls.desc.textures, C.POINTER(st.Texture))[i]
        assert (t.width, t.height) == (203, 131)
        a = np.ctypeslib.as_array(C.cast(t.data, C.POINTER(C.c_uint8)), shape=(131, 203, 4))[..., :3].copy()
        ref = np.asarray(Image.open(scene_dir / name).convert("RGB")).astype(np.float32)
        want = np.floor(np.power(ref[::-1] / 255.0, 2.2) * 255.0)
        assert np.abs(a - want).mean() < (0.1 if "gray" in name else 3.0)
        tex.append(a)
    assert (tex[0] == tex[1]).all() and (tex[2] == tex[3]).all()


def test_exr_writer_roundtrip_through_the_reader(scene_dir):
    """gpt_save_exr (HALF B,G,R like the reference's SaveExr) -> read back as an environment map."""
    import ctypes as C
    rng = np.random.default_rng(3)
    film = (rng.random((9, 14, 3)) * np.array([40.0, 3.0, 0.01])).astype(np.float32)      # row 0 = bottom
    film[0, 0] = [0.0, 65504.0, 1e-7]
    api.save_exr(str(scene_dir / "out.exr"), 14, 9, film)
    js = json.load(open(scene_dir / "scene.json"))
    js["light"].append({"infinite": "out.exr"})
    json.dump(js, open(scene_dir / "scene.json", "w"))
    ls = api.LoadedScene(str(scene_dir / "scene.json"))
    inf = C.cast(ls.desc.infinite, C.POINTER(st.Infinite))[0]
    data = np.ctypeslib.as_array(C.cast(inf.data, C.POINTER(C.c_float)), shape=(9, 14, 3))
    assert np.array_equal(data, film[::-1].astype(np.float16).astype(np.float32))          # top-down, half precision


def test_bvh_cache_roundtrip_and_staleness(scene_dir):
    """bvh.cache in the reference's layout (src/bvh.cpp:189-218) + a content hash: reused when the primitives
    match, rebuilt when the scene changed (the reference silently reuses a stale cache)."""
    path = str(scene_dir / "scene.json")
    api.LoadedScene(path)
    assert not (scene_dir / "bvh.cache").exists()               # opt-in: a plain load never touches the scene directory
    a = api.LoadedScene(path, use_bvh_cache=True)
    cache = scene_dir / "bvh.cache"
    assert cache.exists()
    raw = cache.read_bytes()
    n_nodes, n_prims = struct.unpack("<ii", raw[:8])
    assert (n_nodes, n_prims) == (27, 36) and len(raw) == 32 + 36 * 176 + 27 * 40 + 8
    prims_a = a.array("prims", "n_prims", st.PRIMITIVE).tobytes()
    mtime = cache.stat().st_mtime_ns
    b = api.LoadedScene(path, use_bvh_cache=True)               # second load: served from the cache
    assert cache.stat().st_mtime_ns == mtime
    assert b.array("prims", "n_prims", st.PRIMITIVE).tobytes() == prims_a
    assert b.desc.n_nodes == 27
    # change the geometry: the stale cache must not be used
    obj = (scene_dir / "geometry" / "tall.obj").read_text().replace("v 0.", "v 0.1", 1)
    (scene_dir / "geometry" / "tall.obj").write_text(obj)
    c = api.LoadedScene(path, use_bvh_cache=True)
    assert c.array("prims", "n_prims", st.PRIMITIVE).tobytes() != prims_a
    assert cache.read_bytes() != raw                             # rewritten for the new primitives
    # a cache written by the reference has no trailing hash: accepted when the primitive count matches
    cache.write_bytes(cache.read_bytes()[:-8])
    d = api.LoadedScene(path, use_bvh_cache=True)
    assert d.desc.n_nodes == c.desc.n_nodes


def test_loader_errors(tmp_path, scene_dir):
    lib = api.load()
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(tmp_path / "missing.json"))
    assert "is not good" in str(e.value)
    (tmp_path / "bad.json").write_text('{"camera": {"position": [0,0,0]}, "scene": [')
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(tmp_path / "bad.json"))
    assert "Parse scene error" in str(e.value) and "gpt error -6" in str(e.value)
    (tmp_path / "nocam.json").write_text('{"scene": []}')
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(tmp_path / "nocam.json"))
    assert "must define camera" in str(e.value)
    js = json.load(open(scene_dir / "scene.json"))
    js["scene"][0]["material"] = "DoesNotExist"
    json.dump(js, open(scene_dir / "scene.json", "w"))
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(scene_dir / "scene.json"))
    assert "no material named" in str(e.value)
    js["scene"][0] = {"sphere": True, "material": "General"}
    json.dump(js, open(scene_dir / "scene.json", "w"))
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(scene_dir / "scene.json"))
    assert "sphere" in str(e.value)


def test_crafted_image_files_are_refused_not_overrun(scene_dir):
    """The decoders are hand-written: a short IHDR, an IDAT / ZIP block that inflates far beyond the image, a 64-bit block
    offset near 2^64 and an absurd dataWindow must all come back as load errors (no out-of-bounds read, no memory blow-up)."""
    def png(ihdr, idat):
        def chunk(t, b):
            return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")
    js0 = json.load(open(scene_dir / "scene.json"))

    def load_with_texture(name):
        js = json.loads(json.dumps(js0))
        js["material"].append({"name": "tex", "bsdf": "lambertian", "diffuse": name})
        js["scene"][0]["material"] = "tex"
        json.dump(js, open(scene_dir / "scene.json", "w"))
        return api.LoadedScene(str(scene_dir / "scene.json"))

    good = struct.pack(">IIBBBBB", 4, 4, 8, 6, 0, 0, 0)
    rows = b"".join(b"\x00" + bytes(16) for _ in range(4))
    open(scene_dir / "ok.png", "wb").write(png(good, zlib.compress(rows)))
    assert load_with_texture("ok.png").desc.n_textures == 1
    open(scene_dir / "short_ihdr.png", "wb").write(png(good[:8], zlib.compress(rows)) + bytes(16))
    open(scene_dir / "bomb.png", "wb").write(png(good, zlib.compress(bytes(64 << 20), 9)))      # 64 MiB from 64 KiB
    for name in ("short_ihdr.png", "bomb.png"):
        with pytest.raises(api.GptError):
            load_with_texture(name)

    def load_with_env(name):
        js = json.loads(json.dumps(js0))
        js["light"].append({"infinite": name})
        json.dump(js, open(scene_dir / "scene.json", "w"))
        return api.LoadedScene(str(scene_dir / "scene.json"))

    env = scenes.sky_env(8, 4)
    write_exr(str(scene_dir / "ok.exr"), env, 3, 2)
    assert load_with_env("ok.exr").desc.n_light_distribution == 4
    raw = bytearray(open(scene_dir / "ok.exr", "rb").read())
    table = None
    # the single 16-line ZIP block: its offset is the only table entry; find it by value
    for pos in range(len(raw) - 8):
        if struct.unpack_from("<Q", raw, pos)[0] == pos + 8:
            table = pos
            break
    assert table is not None
    bad = bytearray(raw)
    struct.pack_into("<Q", bad, table, 2 ** 64 - 4)                  # off + 8 wraps around
    open(scene_dir / "wrap.exr", "wb").write(bad)
    bad = bytearray(raw)
    block = table + 8
    comp = zlib.compress(bytes(64 << 20), 9)                          # a ZIP block that inflates to 64 MiB
    bad[block + 4:] = struct.pack("<i", len(comp)) + comp
    open(scene_dir / "bomb.exr", "wb").write(bad)
    bad = bytearray(raw)
    dw = bad.rindex(b"dataWindow\x00box2i\x00") + len(b"dataWindow\x00box2i\x00") + 4
    struct.pack_into("<iiii", bad, dw, -2 ** 31, -2 ** 31, 2 ** 31 - 1, 2 ** 31 - 1)
    open(scene_dir / "window.exr", "wb").write(bad)
    for name in ("wrap.exr", "bomb.exr", "window.exr"):
        with pytest.raises(api.GptError):
            load_with_env(name)

    # a vertex that is not finite: the reference's bucket index would be undefined (bvh.cpp:77)
    prims = np.zeros(5, dtype=st.PRIMITIVE)
    prims["triangle"]["v1"]["v"]["x"] = np.arange(5)
    prims["triangle"]["v2"]["v"]["y"] = 1
    prims["triangle"]["v3"]["v"]["z"] = 1
    api.bvh_build(prims)
    prims["triangle"]["v3"]["v"]["z"][3] = np.inf
    with pytest.raises(api.GptError) as e:
        api.bvh_build(prims)
    assert "non-finite" in str(e.value)


def test_shipped_vpt_scene_settings(scene_dir):
    """The reference's own cornell json is a "vpt" scene with a homogeneous medium, a density grid and a material-less
    mesh around it (scenes/cornell_box/scene.json): all of it is loaded the way parsescene.cpp:72-137,340-392 does."""
    js = json.load(open(scene_dir / "scene.json"))
    js["integrator"] = "vpt"
    js["maxDepth"] = 17
    nx, ny, nz = 4, 3, 2
    vals = (np.arange(nx * ny * nz, dtype=np.float32) * np.float32(0.37) % np.float32(2.5)).astype(np.float32)
    with open(scene_dir / "geometry" / "density.d", "w") as f:
        for v in vals:
            f.write(f"{v:.6f}\n")                                    # one value per line, like the shipped file
    js["medium"] = [{"type": "homogeneous", "sigmaA": [0.0014, 0.0025, 0.0142], "sigmaS": [0.70, 1.22, 1.90], "scale": 25.0, "name": "vol"},
                    {"type": "heterogeneous", "sigmaA": [10, 10, 10], "sigmaS": [90, 90, 90], "nx": nx, "ny": ny, "nz": nz,
                     "p0": [-0.63, 0.27, -0.2415], "p1": [0.693, 1.593, 0.2415], "density": "geometry/density.d", "iterMax": 2000, "name": "hhh"}]
    js["camera"]["medium"] = "vol"
    first_mesh = js["scene"][0]["mesh"]
    js["scene"].append({"mesh": first_mesh, "inside": "hhh", "outside": "", "translate": [0, 0.5, 0]})
    json.dump(js, open(scene_dir / "scene.json", "w"))
    ls = api.LoadedScene(str(scene_dir / "scene.json"))
    assert ls.desc.integrator_type == 2 and ls.desc.max_depth == 17
    assert ls.desc.n_mediums == 2 and ls.camera.medium == 0
    med = np.ctypeslib.as_array(C.cast(ls.desc.mediums, C.POINTER(C.c_uint8)), shape=(2 * 104,)).view(st.MEDIUM)
    assert med[0]["type"] == 0 and med[1]["type"] == 1 and med[0]["g"] == 0
    assert np.allclose([med[0]["sigmaS"]["x"], med[0]["sigmaS"]["y"], med[0]["sigmaS"]["z"]], np.float32([0.70, 1.22, 1.90]) * np.float32(25))
    assert med[0]["sigmaT"]["z"] == np.float32(np.float32(0.0142) * np.float32(25)) + np.float32(np.float32(1.90) * np.float32(25))
    h = med[1]
    assert (h["nx"], h["ny"], h["nz"], h["iterMax"], h["evalTransmittanceType"]) == (nx, ny, nz, 2000, 1)
    assert h["p0"]["x"] == np.float32(-0.63) and h["p1"]["y"] == np.float32(1.593) and h["sigmaT"]["x"] == np.float32(100)
    grid = np.ctypeslib.as_array(C.cast(int(h["density"]), C.POINTER(C.c_float)), shape=(nx * ny * nz,))
    want = np.float32([float(f"{v:.6f}") for v in vals])
    assert (grid == want).all()
    assert h["invMaxDensity"] == np.float32(1) / want.max()
    # the mesh without a material: matIdx -1, the grid inside, nothing outside
    prims = ls.array("prims", "n_prims", st.PRIMITIVE)["triangle"]
    iface = prims[prims["matIdx"] == -1]
    assert len(iface) > 0 and (iface["mediumInside"] == 1).all() and (iface["mediumOutside"] == -1).all()
    assert (prims[prims["matIdx"] != -1]["mediumInside"] == -1).all()
    # "pt" cannot render material-less surfaces and says so before it looks for a device
    ls.set_integrator(st.IT_PT, 8)
    assert ls.desc.integrator_type == st.IT_PT and ls.desc.max_depth == 8
    with pytest.raises(api.GptError) as e:
        api.Renderer(ls.desc, 64, 64, 0.001)
    assert "material" in str(e.value)
    # a grid whose file is short, and a coloured attenuation (the reference exits), are load errors
    with open(scene_dir / "geometry" / "density.d", "w") as f:
        f.write("0.5 0.25\n")
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(scene_dir / "scene.json"))
    assert "holds 2 values" in str(e.value)
    js["medium"][1]["sigmaS"] = [90, 80, 90]
    json.dump(js, open(scene_dir / "scene.json", "w"))
    with pytest.raises(api.GptError) as e:
        api.LoadedScene(str(scene_dir / "scene.json"))
    assert "uniform attenuation" in str(e.value)


@pytest.mark.skipif(not os.path.exists("/root/reference/scenes/cornell_box/scene.json"), reason="the reference tree is not here")
def test_reference_cornell_box_loads_unmodified():
    """scenes/cornell_box/scene.json of the reference, as shipped (vpt, 100x100x40 density grid, material-less mesh)."""
    ls = api.LoadedScene("/root/reference/scenes/cornell_box/scene.json")
    assert ls.desc.integrator_type == st.IT_VPT and ls.desc.max_depth == 17 and ls.desc.n_mediums == 2
    med = np.ctypeslib.as_array(C.cast(ls.desc.mediums, C.POINTER(C.c_uint8)), shape=(2 * 104,)).view(st.MEDIUM)
    h = med[1]
    assert (h["nx"], h["ny"], h["nz"]) == (100, 100, 40) and h["density"] != 0 and h["invMaxDensity"] > 0
    prims = ls.array("prims", "n_prims", st.PRIMITIVE)["triangle"]
    assert (prims["matIdx"] == -1).sum() == 12 and ls.desc.n_lights == 2


def test_ao_scene_settings(scene_dir):
    """"integrator": "ao" + "maxDist" (parsescene.cpp:186-188; default 0.5) land in the union the renderer reads."""
    import struct
    js = json.load(open(scene_dir / "scene.json"))
    js["integrator"] = "ao"
    js.pop("maxDepth", None)
    json.dump(js, open(scene_dir / "scene.json", "w"))
    ls = api.LoadedScene(str(scene_dir / "scene.json"))
    assert ls.desc.integrator_type == st.IT_AO
    assert struct.unpack("<f", struct.pack("<i", ls.desc.max_depth))[0] == 0.5
    js["maxDist"] = 0.75
    json.dump(js, open(scene_dir / "scene.json", "w"))
    ls = api.LoadedScene(str(scene_dir / "scene.json"))
    assert struct.unpack("<f", struct.pack("<i", ls.desc.max_depth))[0] == 0.75


def test_png_writer_follows_savepng(tmp_path):
    """flip Y, clamp, truncate (reference src/imageio.cpp:61-78)"""
    from PIL import Image
    w, h = 7, 5
    rng = np.random.default_rng(2)
    img = (rng.random((h, w, 3)) * 1.4 - 0.2).astype(np.float32)
    api.save_png(str(tmp_path / "o.png"), w, h, img)
    got = np.asarray(Image.open(tmp_path / "o.png"))
    want = (np.clip(img[::-1], 0, 1) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(got, want)


def test_ply_meshes_load_like_obj(scene_dir):
    """veach_bidir of the reference names .ply meshes (assimp reads them there): ascii and binary little-endian PLY
    give the same triangles as the same mesh written as OBJ - with and without normals / uvs."""
    V = np.float32([[-0.3, 0.2, 0.1], [0.4, 0.2, 0.15], [0.45, 0.9, 0.1], [-0.25, 0.95, 0.2], [0.1, 1.3, 0.0]])
    N = np.float32([[0, 0, 1], [0.1, 0, 0.99], [0, 0.1, 0.99], [-0.1, 0, 0.99], [0, 0, 1]])
    T = np.float32([[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 1.5]])
    faces = [[0, 1, 2, 3], [3, 2, 4]]
    js0 = json.load(open(scene_dir / "scene.json"))

    def load(mesh_name):
        js = json.loads(json.dumps(js0))
        js["scene"] = [{"mesh": "geometry/floor.obj", "material": "General"}, {"mesh": mesh_name, "material": "General", "rotate": [0, 20, 0]}]
        json.dump(js, open(scene_dir / "s2.json", "w"))
        ls = api.LoadedScene(str(scene_dir / "s2.json"))
        tris = ls.array("prims", "n_prims", st.PRIMITIVE)["triangle"].copy()
        ls.close()
        return tris

    for with_attr in (True, False):
        with open(scene_dir / "m.obj", "w") as f:
            for v in V: f.write("v %.9g %.9g %.9g\n" % tuple(v))
            if with_attr:
                for n in N: f.write("vn %.9g %.9g %.9g\n" % tuple(n))
                for t in T: f.write("vt %.9g %.9g\n" % tuple(t))
            for fc in faces:
                f.write("f " + " ".join((f"{i+1}/{i+1}/{i+1}" if with_attr else f"{i+1}") for i in fc) + "\n")
        props = "property float x\nproperty float y\nproperty float z\n"
        if with_attr:
            props += "property float nx\nproperty float ny\nproperty float nz\nproperty float s\nproperty float t\n"
        head = lambda fmt: (f"ply\nformat {fmt} 1.0\ncomment test\nelement vertex {len(V)}\n{props}"
                            f"element face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n")
        with open(scene_dir / "a.ply", "w") as f:
            f.write(head("ascii"))
            for i in range(len(V)):
                row = list(V[i]) + (list(N[i]) + list(T[i]) if with_attr else [])
                f.write(" ".join("%.9g" % x for x in row) + "\n")
            for fc in faces: f.write(f"{len(fc)} " + " ".join(map(str, fc)) + "\n")
        with open(scene_dir / "b.ply", "wb") as f:
            f.write(head("binary_little_endian").encode())
            for i in range(len(V)):
                f.write(V[i].tobytes())
                if with_attr: f.write(N[i].tobytes() + T[i].tobytes())
            for fc in faces: f.write(struct.pack("<B", len(fc)) + np.int32(fc).tobytes())
        ref = load("m.obj")
        assert len(ref) == 2 + 3 + 2                   # floor + quad (2) + triangle + the light
        for name in ("a.ply", "b.ply"):
            got = load(name)
            assert len(got) == len(ref) and tri_fields_equal(got, ref), (name, with_attr)
    with open(scene_dir / "bad.ply", "w") as f:
        f.write("ply\nformat binary_big_endian 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(api.GptError):
        load("bad.ply")


@pytest.mark.skipif(not os.path.isdir("/root/reference/scenes"), reason="the reference tree is not here")
def test_every_reference_scene_file_parses(tmp_path):
    """The reference ships 18 scene descriptions but (except for cornell_box) not their meshes and textures.  With a
    placeholder behind every asset name, each json has to go through the loader - every key the reference's scenes use is
    understood - except vol_caustic.json, whose sphere primitive is outside the triangle path tracer (refused by name)."""
    import glob

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o

    obj = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1/1 2/2/1 3/3/1\n"
    ply = "ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\n" \
          "property list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n"
    seen = {}
    for f in sorted(glob.glob("/root/reference/scenes/*/*.json")):
        rel = os.path.relpath(f, "/root/reference/scenes")
        if rel == "cornell_box/fur.json":
            continue                                   # a list of hair segments to paste into a scene, not a scene (nor valid json)
        js = json.load(open(f))
        d = tmp_path / "scenes" / os.path.dirname(rel)
        d.mkdir(parents=True, exist_ok=True)
        for name in strings(js):
            ext = name.lower().rsplit(".", 1)[-1] if "." in name else ""
            path = os.path.normpath(os.path.join(d, name))
            if ext not in ("obj", "ply", "png", "jpg", "jpeg", "exr", "hdr", "d") or os.path.exists(path):
                continue
            os.makedirs(os.path.dirname(path), exist_ok=True)
            if ext == "obj":
                open(path, "w").write(obj)
            elif ext == "ply":
                open(path, "w").write(ply)
            elif ext == "d":
                grid = [m for m in js.get("medium", []) if m.get("density") == name][0]
                open(path, "w").write("0.5\n" * (grid["nx"] * grid["ny"] * grid["nz"]))
            elif ext in ("exr", "hdr"):
                api.save_exr(path, 4, 2, np.ones((2, 4, 3), np.float32))
            else:
                write_png(path, np.full((2, 2, 4), 128, np.uint8))     # (a .jpg name with PNG content: the reader goes by content)
        out = d / os.path.basename(rel)
        json.dump(js, open(out, "w"))
        try:
            ls = api.LoadedScene(str(out))
            seen[rel] = (ls.desc.integrator_type, ls.desc.n_prims, ls.desc.n_lights)
            ls.close()
        except api.GptError as e:
            seen[rel] = str(e)
    assert len(seen) == 18
    bad = {k: v for k, v in seen.items() if isinstance(v, str)}
    assert list(bad) == ["cornell_box/vol_caustic.json"] and "sphere" in bad["cornell_box/vol_caustic.json"], bad
    assert seen["cornell_box/scene.json"][0] == st.IT_VPT
    assert all(v[0] == st.IT_PT and v[1] > 0 for k, v in seen.items() if not isinstance(v, str) and k != "cornell_box/scene.json")


@pytest.mark.skipif(not os.path.exists("/root/reference/scenes/cornell_box/geometry/density.d"), reason="the reference tree is not here")
def test_rebuilt_smoke_scene_is_the_shipped_scene(tmp_path):
    """scenes.write_smoke_scene (what the GPU box renders, having no /root/reference) loads to the same scene as the
    reference's scenes/cornell_box/scene.json: primitives in BVH order, nodes, lights, media and density grid, bit for bit."""
    a = api.LoadedScene("/root/reference/scenes/cornell_box/scene.json")
    b = api.LoadedScene(scenes.write_smoke_scene(str(tmp_path / "smoke")))
    assert (a.width, a.height, a.epsilon, a.camera.medium) == (b.width, b.height, b.epsilon, b.camera.medium)
    assert (a.desc.integrator_type, a.desc.max_depth, a.desc.n_prims, a.desc.n_nodes, a.desc.n_lights, a.desc.n_mediums) == \
           (b.desc.integrator_type, b.desc.max_depth, b.desc.n_prims, b.desc.n_nodes, b.desc.n_lights, b.desc.n_mediums)
    assert tri_fields_equal(a.array("prims", "n_prims", st.PRIMITIVE)["triangle"], b.array("prims", "n_prims", st.PRIMITIVE)["triangle"])
    assert a.array("nodes", "n_nodes", st.BVH_NODE).tobytes() == b.array("nodes", "n_nodes", st.BVH_NODE).tobytes()
    ma = np.ctypeslib.as_array(C.cast(a.desc.mediums, C.POINTER(C.c_uint8)), shape=(2 * 104,)).view(st.MEDIUM).copy()
    mb = np.ctypeslib.as_array(C.cast(b.desc.mediums, C.POINTER(C.c_uint8)), shape=(2 * 104,)).view(st.MEDIUM).copy()
    ga = np.ctypeslib.as_array(C.cast(int(ma[1]["density"]), C.POINTER(C.c_float)), shape=(400000,))
    gb = np.ctypeslib.as_array(C.cast(int(mb[1]["density"]), C.POINTER(C.c_float)), shape=(400000,))
    assert ga.tobytes() == gb.tobytes()
    for name in st.MEDIUM.names:                     # (field by field: the records' tail padding is not defined)
        if name != "density":
            assert ma[name].tobytes() == mb[name].tobytes(), name
    a.close()
    b.close()

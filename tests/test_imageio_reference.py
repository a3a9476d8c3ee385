"""The image files either side of the path, against the REFERENCE'S OWN decoders.

The reference reads textures with stb_image, environment maps with tinyexr and writes pictures with stb_image_write / tinyexr
(src/imageio.cpp:1-9).  Those are vendored single-header libraries that compile with g++ on their own, so this part of the
reference is built here where it lies (oracle/ref_imageio.cpp -> oracle/_ref/libref_imageio.so, see oracle/Makefile) and the
product's decoders (gpu_pathtracer_amd/csrc/imageio.cpp, through the C ABI) must agree with it BIT FOR BIT: byte work.

The few lines of src/imageio.cpp around those calls (1/255, powf(x, 2.2f), the flip of SavePng, Texture::Texture's truncation)
are restated below, next to the call they follow.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from gpu_pathtracer_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_imageio.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF_LIB) and os.path.isdir("/root/reference"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    if not os.path.exists(REF_LIB):
        pytest.skip("oracle/_ref/libref_imageio.so is not built and /root/reference is not here to build it from")
    lib = C.CDLL(REF_LIB)
    lib.ref_stbi_load_flipped.restype = C.c_void_p
    lib.ref_stbi_load_flipped.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_free.argtypes = [C.c_void_p]
    lib.ref_free.restype = None
    lib.ref_stbi_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    lib.ref_load_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ref_save_exr.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    return lib


def ref_decode8(ref, path):
    """stbi_load as ImageIO::LoadTexture calls it (src/imageio.cpp:13-14) -> uint8 [H, W, components] or None"""
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    p = ref.ref_stbi_load_flipped(os.fsencode(str(path)), C.byref(w), C.byref(h), C.byref(c))
    if not p:
        return None
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), (h.value, w.value, c.value)).copy()
    ref.ref_free(p)
    return out


def ref_load_exr(ref, path):
    """LoadEXR as ImageIO::LoadExr calls it (src/imageio.cpp:84) -> float32 [H, W, 4] (row 0 = top) or None"""
    w, h, p = C.c_int(), C.c_int(), C.c_void_p()
    if ref.ref_load_exr(os.fsencode(str(path)), C.byref(p), C.byref(w), C.byref(h)) != 0:
        return None
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), (h.value, w.value, 4)).copy()
    ref.ref_free(p)
    return out


def ref_save_exr(ref, path, rgb_top_down, compression, half):
    planes = [np.ascontiguousarray(rgb_top_down[..., c], dtype=np.float32) for c in range(3)]
    h, w = planes[0].shape
    rc = ref.ref_save_exr(os.fsencode(str(path)), w, h, planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data,
                          compression, half)
    assert rc == 0


def radiance(rng, h, w, kind):
    if kind == "smooth":                                    # few distinct 16-bit values: PIZ's 14-bit wavelet
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([x / max(w - 1, 1), y / max(h - 1, 1), (x + y) % 7 / 7.0], -1).astype(np.float32)
        return (np.round(img * 40) / 8).astype(np.float32)
    img = np.exp(rng.normal(0, 3, (h, w, 3))).astype(np.float32)     # sky-like dynamic range, every value different
    img[rng.random((h, w)) < 0.05] = 0.0
    return img


NAMES = {0: "none", 1: "rle", 2: "zips", 3: "zip", 4: "piz"}


@pytest.mark.parametrize("compression", [0, 1, 2, 3, 4], ids=lambda c: NAMES[c])
@pytest.mark.parametrize("half", [1, 0], ids=["half", "float"])
def test_exr_reader_equals_tinyexr(ref, tmp_path, compression, half):
    """gpt_load_exr == the reference's LoadEXR on files written by the reference's own encoder: every compression tinyexr has,
    HALF and FLOAT channels, sizes that leave partial blocks (ZIP: 16 lines, PIZ: 32), one-pixel rows and columns."""
    rng = np.random.default_rng(compression * 2 + half)
    for h, w, kind in ((1, 1, "noise"), (1, 9, "noise"), (9, 1, "smooth"), (37, 45, "smooth"), (37, 45, "noise"), (64, 32, "noise"),
                       (33, 70, "noise"), (40, 700, "noise")):
        img = radiance(rng, h, w, kind)
        path = tmp_path / f"{NAMES[compression]}_{h}x{w}_{kind}.exr"
        ref_save_exr(ref, path, img, compression, half)
        want = ref_load_exr(ref, path)
        assert want is not None and want.shape == (h, w, 4)
        got = api.load_exr(str(path))
        assert got.shape == (h, w, 3)
        assert np.array_equal(got.view(np.uint32), want[..., :3].copy().view(np.uint32)), (h, w, kind)
        if not half:
            assert np.array_equal(got.view(np.uint32), img.view(np.uint32))          # FLOAT files are lossless


def test_piz_takes_both_wavelet_paths(ref, tmp_path):
    """PIZ's wavelet has a 14-bit and a 16-bit form, chosen by the number of distinct values in the block: make sure the files
    above exercise both (so that the equality is not vacuous for one of them)."""
    rng = np.random.default_rng(5)
    for kind, h, w, wide in (("smooth", 37, 45, False), ("noise", 40, 700, True)):
        img = radiance(rng, h, w, kind)
        with np.errstate(over="ignore"):
            block = img[:32].astype(np.float16).view(np.uint16)
        assert (len(np.unique(block)) >= (1 << 14)) == wide


def test_exr_writer_is_read_back_by_tinyexr(ref, tmp_path):
    """gpt_save_exr (HALF B,G,R, rows top-down from the bottom-up film, ImageIO::SaveExr src/imageio.cpp:104-161) through the
    reference's reader: the same values the reference's own writer stores for that film, including its float -> half rounding."""
    rng = np.random.default_rng(11)
    film = radiance(rng, 23, 31, "noise")
    film[0, :8] = np.array([1e-9, 6.0e-8, 6.1e-5, 65504.0, 65520.0, 1e9, 0.1, 1.0 / 3.0], np.float32)[:, None]
    api.save_exr(str(tmp_path / "ours.exr"), 31, 23, film)
    ref_save_exr(ref, tmp_path / "theirs.exr", film, 0, 1)        # the reference hands SaveExr its top-down copy: compare unflipped
    ours = ref_load_exr(ref, tmp_path / "ours.exr")
    theirs = ref_load_exr(ref, tmp_path / "theirs.exr")
    assert ours is not None and theirs is not None
    # gpt_save_exr takes the film as the kernel holds it (row 0 = bottom) and stores it top-down
    assert np.array_equal(ours[::-1, :, :3].copy().view(np.uint32), theirs[..., :3].copy().view(np.uint32))


def png_cases(rng, directory):
    from PIL import Image
    cases = []
    h, w = 19, 27
    grey = rng.integers(0, 256, (h, w), dtype=np.uint8)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    for name, im, kw in (("grey", Image.fromarray(grey, "L"), {}), ("grey_alpha", Image.fromarray(rgba[..., :2].copy(), "LA"), {}),
                         ("rgb", Image.fromarray(rgb, "RGB"), {}), ("rgba", Image.fromarray(rgba, "RGBA"), {}),
                         ("rgb_fast", Image.fromarray(rgb, "RGB"), {"compress_level": 1}),
                         ("rgb_stored", Image.fromarray(rgb, "RGB"), {"compress_level": 0}),
                         ("palette", Image.fromarray(rgb, "RGB").quantize(17), {}),
                         ("palette_transparent", Image.fromarray(rgb, "RGB").quantize(17), {"transparency": 3}),
                         ("smooth", Image.fromarray((np.add.outer(np.arange(64), np.arange(80)) % 256).astype(np.uint8), "L").convert("RGB"), {"optimize": True})):
        path = os.path.join(directory, name + ".png")
        im.save(path, **kw)
        cases.append(path)
    return cases


def test_png_reader_equals_stb_image(ref, tmp_path):
    """gpt_decode_image8 == stbi_load(flip) for every 8-bit PNG colour type (and channel count reported the same way)."""
    rng = np.random.default_rng(3)
    for path in png_cases(rng, str(tmp_path)):
        want = ref_decode8(ref, path)
        got = api.decode_image8(path)
        assert want is not None and got.shape == want.shape, path
        assert np.array_equal(got, want), path


def texels_like_the_reference(ref, path):
    """ImageIO::LoadTexture (src/imageio.cpp:11-59) on stb_image's bytes, then Texture::Texture (src/texture.h:15-27)."""
    px = ref_decode8(ref, path)
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    inv = np.float32(1.0) / np.float32(255.0)
    unit = (np.arange(256, dtype=np.float32) * inv).astype(np.float32)
    lin = np.array([libm.powf(float(v), 2.2) for v in unit], np.float32)           # powf(texel, 2.2f)
    to8 = lambda v: (v * np.float32(255.0)).astype(np.uint8)                        # unsigned char(x * 255)
    h, w, comp = px.shape
    out = np.empty((h, w, 4), np.uint8)
    if comp == 1:
        out[..., :3] = to8(lin[px[..., 0]])[..., None]
        out[..., 3] = 255
    else:
        out[..., :3] = to8(lin[px[..., :3]])
        out[..., 3] = to8(unit[px[..., 3]]) if comp == 4 else 255
    return out


def test_texture_texels_equal_loadtexture_on_stb_bytes(ref, tmp_path):
    rng = np.random.default_rng(4)
    for path in png_cases(rng, str(tmp_path)):
        if ref_decode8(ref, path).shape[2] == 2:
            continue        # LoadTexture leaves a two-channel texel uninitialised (src/imageio.cpp:25-41): nothing to compare with
        assert np.array_equal(api.load_texture(path), texels_like_the_reference(ref, path)), path


def test_png_writer_equals_savepng_through_stb(ref, tmp_path):
    """gpt_save_png vs ImageIO::SavePng (src/imageio.cpp:61-78: flip, clamp, truncate, stbi_write_png): the two files decode to
    the same pixels (the deflate streams differ: stb compresses, the product stores)."""
    rng = np.random.default_rng(8)
    h, w = 21, 34
    film = rng.uniform(-0.2, 1.3, (h, w, 3)).astype(np.float32)
    film[0, :4] = [[0, 0, 0], [1, 1, 1], [0.999999, 0.5, 1.0 / 255], [np.float32(1.0) - np.float32(6e-8)] * 3]
    api.save_png(str(tmp_path / "ours.png"), w, h, film)
    clamped = np.clip(film, np.float32(0), np.float32(1))
    bytes_top_down = np.ascontiguousarray((clamped[::-1] * np.float32(255.0)).astype(np.uint32).astype(np.uint8))
    assert ref.ref_stbi_write_png(os.fsencode(str(tmp_path / "theirs.png")), w, h, bytes_top_down.ctypes.data) != 0
    ours, theirs = ref_decode8(ref, tmp_path / "ours.png"), ref_decode8(ref, tmp_path / "theirs.png")
    assert ours is not None and np.array_equal(ours, theirs)
    assert np.array_equal(api.decode_image8(str(tmp_path / "theirs.png")), theirs)       # and the product reads stb's file


def jpeg_cases(rng, directory):
    from PIL import Image
    y, x = np.mgrid[0:83, 0:117]
    img = np.stack([127 + 120 * np.sin(x / 9.0), 127 + 120 * np.cos(y / 7.0), (3 * x + 2 * y) % 256], -1)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    cases = []
    for name, kw in (("444_q90", dict(quality=90, subsampling=0)), ("422_q75", dict(quality=75, subsampling=1)),
                     ("420_q60", dict(quality=60, subsampling=2)), ("420_q95_progressive", dict(quality=95, subsampling=2, progressive=True)),
                     ("444_q30_progressive", dict(quality=30, subsampling=0, progressive=True)),
                     ("420_restart", dict(quality=80, subsampling=2, restart_marker_blocks=3)),
                     ("422_q100", dict(quality=100, subsampling=1)), ("420_q5", dict(quality=5, subsampling=2))):
        path = os.path.join(directory, name + ".jpg")
        Image.fromarray(img).save(path, **kw)
        cases.append(path)
    for name, kw in (("grey_q70", dict(quality=70)), ("grey_progressive", dict(quality=85, progressive=True))):
        path = os.path.join(directory, name + ".jpg")
        Image.fromarray(img[..., 0].copy(), "L").save(path, **kw)
        cases.append(path)
    for name, size in (("tiny_1x1", (1, 1)), ("odd_17x9", (9, 17)), ("odd_8x33", (33, 8))):
        path = os.path.join(directory, name + ".jpg")
        Image.fromarray(img[:size[0], :size[1]].copy()).save(path, quality=85, subsampling=2)
        cases.append(path)
    return cases


def test_jpeg_reader_equals_stb_image(ref, tmp_path):
    """gpt_decode_image8 == stbi_load(flip) on JPEG files: stb_image's fixed-point inverse DCT, its (3,1)-tap chroma filters
    and its fixed-point YCbCr -> RGB are part of what a texel IS for the reference, so the product's reader restates them."""
    rng = np.random.default_rng(6)
    for path in jpeg_cases(rng, str(tmp_path)):
        want = ref_decode8(ref, path)
        got = api.decode_image8(path)
        assert want is not None and got.shape == want.shape, path
        diff = np.abs(got.astype(int) - want.astype(int))
        assert diff.max() == 0, (os.path.basename(path), int(diff.max()), float((diff > 0).mean()))
        assert np.array_equal(api.load_texture(path), texels_like_the_reference(ref, path)), path


def encode_baseline_jpeg(img, sampling, quality_scale=2):
    """A minimal baseline JPEG writer for the sampling factors PIL cannot produce (chroma 1x2, 4x1, 4x2 ...): float DCT,
    one flat-ish quantisation table, the Huffman tables of a PIL file (Annex K).  sampling: (h, v) of the luma component;
    both chroma components are 1x1, i.e. sub-sampled by h horizontally and v vertically."""
    import io
    from PIL import Image
    probe = io.BytesIO()
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(probe, "JPEG", quality=75, subsampling=0)
    d = probe.getvalue()
    tables, pos, dht_segments = {}, 2, b""
    while pos < len(d):
        marker, length = d[pos + 1], int.from_bytes(d[pos + 2:pos + 4], "big")
        if marker == 0xc4:
            dht_segments += d[pos:pos + 2 + length]
            q = pos + 4
            while q < pos + 2 + length:
                tc_th, bits = d[q], d[q + 1:q + 17]
                vals = d[q + 17:q + 17 + sum(bits)]
                code, k, table = 0, 0, {}
                for l in range(1, 17):
                    for _ in range(bits[l - 1]):
                        table[vals[k]] = (code, l)
                        code += 1
                        k += 1
                    code <<= 1
                tables[tc_th] = table
                q += 17 + sum(bits)
        if marker == 0xda:
            break
        pos += 2 + length
    zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    qt = np.clip((1 + np.add.outer(np.arange(8), np.arange(8))) * quality_scale, 1, 255).astype(np.int32)
    H, W = img.shape[:2]
    h, v = sampling
    f = img.astype(np.float64)
    ycc = np.stack([0.299 * f[..., 0] + 0.587 * f[..., 1] + 0.114 * f[..., 2],
                    128 - 0.168736 * f[..., 0] - 0.331264 * f[..., 1] + 0.5 * f[..., 2],
                    128 + 0.5 * f[..., 0] - 0.418688 * f[..., 1] - 0.081312 * f[..., 2]], -1)
    mw, mh = 8 * h, 8 * v
    PW, PH = -(-W // mw) * mw, -(-H // mh) * mh
    ycc = np.pad(ycc, ((0, PH - H), (0, PW - W), (0, 0)), mode="edge")
    planes = [ycc[..., 0], ycc[::v, ::h, 1], ycc[::v, ::h, 2]]          # chroma by point sampling: any choice is a valid file
    k = np.arange(8)
    basis = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * np.where(k == 0, np.sqrt(1 / 8), 0.5)[:, None]
    bits = []

    def put(code, length):
        bits.append((code, length))

    def magnitude(x):
        s = 0 if x == 0 else int(abs(x)).bit_length()
        return s, (x if x >= 0 else x + (1 << s) - 1)
    pred = [0, 0, 0]

    def block(c, by, bx):
        px = planes[c][by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] - 128.0
        coef = np.rint(basis @ px @ basis.T / qt).astype(int).reshape(64)
        dc_t, ac_t = tables[0 if c == 0 else 1], tables[0x10 if c == 0 else 0x11]
        s, m = magnitude(coef[0] - pred[c])
        pred[c] = coef[0]
        put(*dc_t[s])
        if s:
            put(m, s)
        run = 0
        last = max([i for i in range(1, 64) if coef[zz[i]] != 0], default=0)
        for i in range(1, last + 1):
            x = coef[zz[i]]
            if x == 0:
                run += 1
                continue
            while run > 15:
                put(*ac_t[0xf0])
                run -= 16
            s, m = magnitude(x)
            put(*ac_t[run << 4 | s])
            put(m, s)
            run = 0
        if last < 63:
            put(*ac_t[0])
    for my in range(PH // mh):
        for mx in range(PW // mw):
            for yy in range(v):
                for xx in range(h):
                    block(0, my * v + yy, mx * h + xx)
            block(1, my, mx)
            block(2, my, mx)
    acc, n, body = 0, 0, bytearray()
    for code, length in bits:
        acc = acc << length | code
        n += length
        while n >= 8:
            n -= 8
            byte = acc >> n & 0xff
            body.append(byte)
            if byte == 0xff:
                body.append(0)
    if n:
        byte = (acc << (8 - n) | (1 << (8 - n)) - 1) & 0xff
        body.append(byte)
        if byte == 0xff:
            body.append(0)
    seg = lambda marker, payload: bytes([0xff, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload
    out = b"\xff\xd8" + seg(0xe0, b"JFIF\0\1\1\0\0\1\0\1\0\0")
    out += seg(0xdb, bytes([0]) + bytes(int(qt.reshape(64)[zz[i]]) for i in range(64)))
    out += seg(0xc0, bytes([8]) + H.to_bytes(2, "big") + W.to_bytes(2, "big") + bytes([3, 1, h << 4 | v, 0, 2, 0x11, 0, 3, 0x11, 0]))
    out += dht_segments
    out += seg(0xda, bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0])) + bytes(body) + b"\xff\xd9"
    return out


@pytest.mark.parametrize("sampling", [(1, 2), (4, 1), (4, 2), (2, 4), (1, 1), (2, 2), (2, 1)], ids=lambda s: "%dx%d" % s)
def test_jpeg_unusual_chroma_sampling_equals_stb_image(ref, tmp_path, sampling):
    """The sampling factors PIL cannot write: vertical-only halving (stb_image's 3:1 vertical filter), factors of 4 (replication),
    and the usual ones again through this file's own encoder as a check of the encoder."""
    rng = np.random.default_rng(sampling[0] * 8 + sampling[1])
    for h, w in ((50, 70), (16, 32), (1, 1), (9, 5), (33, 3)):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 120 * np.sin(x / 5.0 + y / 11.0), 127 + 120 * np.cos(y / 4.0), (5 * x + 3 * y) % 256], -1)
        img = np.clip(img + rng.normal(0, 10, img.shape), 0, 255).astype(np.uint8)
        path = tmp_path / f"s{sampling[0]}{sampling[1]}_{h}x{w}.jpg"
        path.write_bytes(encode_baseline_jpeg(img, sampling))
        want = ref_decode8(ref, path)
        assert want is not None and want.shape == (h, w, 3), path
        assert np.abs(want[::-1].astype(int) - img.astype(int)).mean() < 40      # (the file really holds the picture)
        got = api.decode_image8(str(path))
        assert got.shape == want.shape and np.array_equal(got, want), path


def test_damaged_files_are_refused_or_decoded_never_overrun(ref, tmp_path):
    """Random damage to PIZ, JPEG and PNG files (the formats with the most pointer arithmetic in the reader): every call returns -
    an error or some picture of the declared size.  (The same mutations ran under AddressSanitizer while the readers were
    written; this keeps a short version of that in the suite.)"""
    rng = np.random.default_rng(12)
    seeds = []
    for i, (h, w, kind, half) in enumerate(((37, 45, "smooth", 1), (40, 300, "noise", 1), (33, 20, "noise", 0))):
        path = tmp_path / f"seed{i}.exr"
        ref_save_exr(ref, path, radiance(rng, h, w, kind), 4, half)
        seeds.append((path.read_bytes(), api.load_exr))
    for path in jpeg_cases(rng, str(tmp_path))[:6]:
        seeds.append((open(path, "rb").read(), api.decode_image8))
    for depth, ctype, lace, ch in ((16, 2, True, 3), (4, 3, True, 1), (1, 0, False, 1), (8, 6, False, 4)):
        plte = rng.integers(0, 256, 48).tolist() if ctype == 3 else None
        seeds.append((encode_png(rng.integers(0, 1 << depth, (13, 21, ch)), depth, ctype, lace, plte, None, rng), api.decode_image8))
    decoded = refused = 0
    target = str(tmp_path / "damaged.bin")
    for data, reader in seeds:
        for _ in range(120):
            m = bytearray(data)
            for _ in range(int(rng.integers(1, 5))):
                p = int(rng.integers(0, len(m)))
                m[p] = (int(rng.integers(0, 256)), m[p] ^ (1 << int(rng.integers(0, 8))), 0xff, 0)[int(rng.integers(0, 4))]
            if rng.random() < 0.1:
                del m[int(rng.integers(1, len(m))):]
            open(target, "wb").write(m)
            try:
                out = reader(target)
                assert out.size > 0
                decoded += 1
            except RuntimeError:
                refused += 1
    assert decoded > 100 and refused > 100


def test_piz_bit_count_and_jpeg_sampling_overreads_are_refused(ref, tmp_path):
    """Two crafted files that made the readers read past a buffer (found by review, reproduced under AddressSanitizer): a PIZ
    block whose Huffman bit count is within 7 of 2^32 (the byte count wrapped in 32 bits and passed the length check), and a
    JPEG whose sampling factors do not divide the largest one (H = 4, 3, 1: stb_image pads its planes, this reader does not)."""
    import struct
    rng = np.random.default_rng(5)
    path = tmp_path / "piz.exr"
    ref_save_exr(ref, path, radiance(rng, 20, 24, "smooth"), 4, 1)
    raw = bytearray(path.read_bytes())
    assert api.load_exr(str(path)).shape == (20, 24, 3)
    pos = 8                                              # magic, version; then the attributes up to an empty name
    while raw[pos] != 0:
        pos = raw.index(b"\0", pos) + 1                  # name
        pos = raw.index(b"\0", pos) + 1                  # type
        pos += 4 + struct.unpack_from("<i", raw, pos)[0]
    block = struct.unpack_from("<Q", raw, pos + 1)[0]    # the offset table's first entry: {y, size} {min, max non-zero, bitmap, length, Huffman block}
    assert struct.unpack_from("<i", raw, block + 4)[0] < 20 * 24 * 3 * 2      # (a block that did not shrink is stored as it is)
    lo, hi = struct.unpack_from("<HH", raw, block + 8)
    huf = block + 12 + (hi - lo + 1 if hi >= lo else 0) + 4
    im, iM, _, n_bits = struct.unpack_from("<IIII", raw, huf)
    assert im <= iM < 65537 and 0 < n_bits < 8 * len(raw)        # (this is the Huffman header)
    for bad in (0xffffffff, 0xfffffffb, 0xfffffff9):
        struct.pack_into("<I", raw, huf + 12, bad)
        (tmp_path / "piz_bad.exr").write_bytes(raw)
        with pytest.raises(RuntimeError):
            api.load_exr(str(tmp_path / "piz_bad.exr"))

    y, x = np.mgrid[0:8, 0:32]
    img = np.stack([x * 8, y * 30, x + y], -1).astype(np.uint8)
    data = bytearray(encode_baseline_jpeg(img, (1, 1)))
    sof = data.index(b"\xff\xc0")
    assert data[sof + 9] == 3                             # three components: {id, sampling, table} from sof + 10
    data[sof + 11], data[sof + 14], data[sof + 17] = 0x41, 0x31, 0x11
    (tmp_path / "h431.jpg").write_bytes(data)
    with pytest.raises(RuntimeError):
        api.decode_image8(str(tmp_path / "h431.jpg"))


def test_decoder_entry_points_report_errors_and_sizes(tmp_path):
    """The C ABI of the decoders: size-only calls, buffers that are too small, files that are not pictures (no reference needed)."""
    import ctypes as C
    from PIL import Image
    lib = api.load()
    png = str(tmp_path / "a.png")
    Image.fromarray(np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)).save(png)
    w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.gpt_decode_image8(os.fsencode(png), C.byref(w), C.byref(h), C.byref(c), None, 0) == 0
    assert (w.value, h.value, c.value) == (7, 5, 3)
    small = np.zeros(10, np.uint8)
    assert lib.gpt_decode_image8(os.fsencode(png), C.byref(w), C.byref(h), C.byref(c), small.ctypes.data, small.size) == -1
    assert lib.gpt_load_texture(os.fsencode(png), C.byref(w), C.byref(h), small.ctypes.data, 2) == -1
    assert lib.gpt_decode_image8(None, C.byref(w), C.byref(h), C.byref(c), None, 0) == -1
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"not a picture at all" * 10)
    for reader in (api.decode_image8, api.load_texture, api.load_exr):
        with pytest.raises(api.GptError):
            reader(str(junk))
        with pytest.raises(api.GptError):
            reader(str(tmp_path / "missing.file"))
    assert api.decode_image8(png).shape == (5, 7, 3) and api.load_texture(png).shape == (5, 7, 4)


def encode_png(samples, depth, ctype, interlace=False, plte=None, trns=None, rng=None):
    """A PNG writer for the corners of the format PIL does not write (Adam7, 16-bit RGB(A), 1 / 2 / 4-bit grey and palettes,
    colour keys): samples [H, W, channels] of unsigned ints below 2**depth; every scanline gets a random filter type."""
    import struct
    import zlib
    H, W, ch = samples.shape
    rng = rng or np.random.default_rng(0)

    def pack_row(row):                                   # [w, ch] -> bytes at this bit depth
        flat = row.reshape(-1).astype(np.uint32)
        if depth == 16:
            return b"".join(struct.pack(">H", int(v)) for v in flat)
        if depth == 8:
            return bytes(int(v) for v in flat)
        bits = "".join(format(int(v), "0%db" % depth) for v in flat)
        bits += "0" * (-len(bits) % 8)
        return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if pa <= pb and pa <= pc else (b if pb <= pc else c)

    def filtered(rows):                                  # list of byte rows of one pass -> filter byte + filtered bytes each
        bpp = max(1, ch * depth // 8)
        out, prev = b"", None
        for cur in rows:
            ft = int(rng.integers(0, 5))
            up = prev if prev is not None else bytes(len(cur))
            line = bytearray()
            for i, x in enumerate(cur):
                a = cur[i - bpp] if i >= bpp else 0
                b = up[i]
                c = up[i - bpp] if i >= bpp else 0
                pred = (0, a, b, (a + b) >> 1, paeth(a, b, c))[ft]
                line.append((x - pred) & 255)
            out += bytes([ft]) + bytes(line)
            prev = cur
        return out
    if interlace:
        data = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                data += filtered([pack_row(r) for r in sub])
    else:
        data = filtered([pack_row(r) for r in samples])

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))
    z = zlib.compress(data, 6)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        out += chunk(b"PLTE", bytes(plte))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    half = len(z) // 2                                    # two IDAT chunks: the stream continues across them
    return out + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")


def test_png_every_colour_type_depth_and_interlace_equals_stb_image(ref, tmp_path):
    """The whole PNG matrix: colour types 0 / 2 / 3 / 4 / 6 at every legal bit depth, plain and Adam7, with and without a
    tRNS chunk (a colour key for grey / RGB - compared at 16 bits in 16-bit files -, alpha per entry for palettes), sizes that leave
    empty Adam7 passes.  stb_image decides what a reader may decide (16 -> 8 bits by the high byte, 1 / 2 / 4-bit grey scaled
    by 255 / 85 / 17, a key adds a channel); the product's reader has to agree byte for byte and channel count for channel count."""
    import struct
    rng = np.random.default_rng(9)
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}
    depths = {0: (1, 2, 4, 8, 16), 2: (8, 16), 3: (1, 2, 4, 8), 4: (8, 16), 6: (8, 16)}
    n = 0
    for ctype in (0, 2, 3, 4, 6):
        for depth in depths[ctype]:
            for interlace in (False, True):
                for with_trns in ((False, True) if ctype in (0, 2, 3) else (False,)):
                    for h, w in ((13, 21), (1, 1), (3, 2), (9, 5)):
                        ch = channels[ctype]
                        samples = rng.integers(0, 1 << depth, (h, w, ch))
                        plte = trns = None
                        if ctype == 3:
                            n_entries = 1 << depth
                            plte = rng.integers(0, 256, 3 * n_entries).tolist()
                            if with_trns:
                                trns = rng.integers(0, 256, max(1, n_entries // 2)).tolist()      # shorter than the palette: the rest is opaque
                        elif with_trns:
                            key = samples[h // 2, w // 2]                                       # a key that does occur
                            if depth == 16:
                                samples[0, 0] = key ^ 1 if ch == 1 else key                     # same high byte, different sample
                                if ch == 1:
                                    samples[0, 0] = (int(key[0]) & 0xff00) | ((int(key[0]) + 1) & 0xff)
                            trns = b"".join(struct.pack(">H", int(v)) for v in key)
                        path = tmp_path / f"t{ctype}_d{depth}_{'i' if interlace else 'p'}_{'k' if with_trns else 'n'}_{h}x{w}.png"
                        path.write_bytes(encode_png(samples, depth, ctype, interlace, plte, trns, rng))
                        want = ref_decode8(ref, path)
                        assert want is not None, path.name
                        got = api.decode_image8(str(path))
                        assert got.shape == want.shape, (path.name, got.shape, want.shape)
                        assert np.array_equal(got, want), path.name
                        n += 1
    assert n == 4 * (2 * 2 * 5 + 2 * 2 * 2 + 2 * 2 * 4 + 2 * 2 + 2 * 2)


def bmp_bytes(pixels_bottom_up, bpp, hsz=40, compress=0, masks=None, palette=None, top_down=False, extra_gap=0):
    """A BMP file from rows of already-packed pixel values: pixels_bottom_up [H, W] of ints (palette indices or packed pixels)."""
    import struct
    h, w = pixels_bottom_up.shape
    rows = b""
    for r in pixels_bottom_up:
        if bpp == 1:
            bits = "".join(str(int(v) & 1) for v in r) + "0" * (-w % 8)
            line = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
        elif bpp == 4:
            vals = [int(v) & 15 for v in r] + [0] * (w % 2)
            line = bytes(vals[i] << 4 | vals[i + 1] for i in range(0, len(vals), 2))
        elif bpp == 8:
            line = bytes(int(v) & 255 for v in r)
        else:
            line = b"".join(int(v).to_bytes(bpp // 8, "little") for v in r)
        rows += line + b"\0" * (-len(line) % 4)
    pal = b""
    if palette is not None:
        pal = b"".join(bytes([c[2], c[1], c[0]]) + (b"" if hsz == 12 else b"\0") for c in palette)
    if hsz == 12:
        info = struct.pack("<IHHHH", 12, w, h, 1, bpp)
    else:
        info = struct.pack("<IiiHHIIiiII", hsz, w, -h if top_down else h, 1, bpp, compress, len(rows), 2835, 2835, 0, 0)
        if hsz == 40 and compress == 3:
            info += struct.pack("<III", *masks[:3])                 # the masks follow a 40-byte header
        elif hsz >= 56:
            m = list(masks or (0, 0, 0, 0)) + [0] * 4
            info += struct.pack("<IIII", *m[:4])
            if hsz == 56 and compress == 3:
                info += struct.pack("<III", *m[:3])                 # stb_image 2.19 looks for a 56-byte header's masks BEHIND it
            if hsz >= 108:
                info += b"BGRs" + b"\0" * 48
            if hsz == 124:
                info += b"\0" * 16
    offset = 14 + len(info) + len(pal) + extra_gap
    return b"BM" + struct.pack("<IHHI", offset + len(rows), 0, 0, offset) + info + pal + b"\xaa" * extra_gap + rows


def test_bmp_reader_equals_stb_image(ref, tmp_path):
    """BMP as stb_image 2.19 reads it: OS/2 and Windows headers (12 / 40 / 56 / 108 / 124 bytes), 1 / 4 / 8-bit palettes, 16-bit 5-5-5
    and bit-field layouts (5-6-5, 4-4-4-4, odd masks widened by bit replication), 24 and 32 bits, an alpha channel that is zero
    everywhere (becomes opaque), top-down files, row padding at every width, a gap before the pixels; RLE files are refused by both."""
    from PIL import Image
    rng = np.random.default_rng(14)
    files = []
    rgb = rng.integers(0, 256, (11, 13, 3), dtype=np.uint8)
    for name, im in (("pil_rgb", Image.fromarray(rgb)), ("pil_l", Image.fromarray(rgb[..., 0].copy(), "L")),
                     ("pil_p", Image.fromarray(rgb).quantize(40)), ("pil_1", Image.fromarray(rgb[..., 0].copy(), "L").convert("1")),
                     ("pil_rgba", Image.fromarray(np.dstack([rgb, rgb[..., :1]]), "RGBA"))):
        path = tmp_path / (name + ".bmp")
        im.save(path)
        files.append(path)
    pal16 = rng.integers(0, 256, (16, 3)).tolist()
    pal256 = rng.integers(0, 256, (256, 3)).tolist()
    for w in (1, 2, 3, 4, 5, 7, 8, 9, 16, 17):
        h = 5
        cases = [("p8", rng.integers(0, 256, (h, w)), dict(bpp=8, palette=pal256)), ("p4", rng.integers(0, 16, (h, w)), dict(bpp=4, palette=pal16)),
                 ("p1", rng.integers(0, 2, (h, w)), dict(bpp=1, palette=pal16[:2])), ("os2_p8", rng.integers(0, 256, (h, w)), dict(bpp=8, hsz=12, palette=pal256)),
                 ("os2_24", rng.integers(0, 1 << 24, (h, w)), dict(bpp=24, hsz=12)), ("rgb24", rng.integers(0, 1 << 24, (h, w)), dict(bpp=24)),
                 ("rgb24_topdown", rng.integers(0, 1 << 24, (h, w)), dict(bpp=24, top_down=True)), ("rgb24_gap", rng.integers(0, 1 << 24, (h, w)), dict(bpp=24, extra_gap=6)),
                 ("x555", rng.integers(0, 1 << 16, (h, w)), dict(bpp=16)), ("r565", rng.integers(0, 1 << 16, (h, w)), dict(bpp=16, compress=3, masks=(0xf800, 0x07e0, 0x001f))),
                 ("a4444_v4", rng.integers(0, 1 << 16, (h, w)), dict(bpp=16, hsz=108, compress=3, masks=(0x0f00, 0x00f0, 0x000f, 0xf000))),
                 ("bgr_a_low_v5", rng.integers(0, 1 << 32, (h, w)), dict(bpp=32, hsz=124, compress=3, masks=(0x0000ff00, 0x00ff0000, 0x7f000000, 0x000000ff))),       # (a mask reaching bit 31 outside the standard layout trips an assert in stb_image: not a case)
                 ("m565_in_32_v5", rng.integers(0, 1 << 32, (h, w)), dict(bpp=32, hsz=124, compress=3, masks=(0x00f80000, 0x0007e000, 0x00001f00, 0x000000c0))),       # (channels wider than 8 bits trip an assert in stb_image: not a case)
                 ("m233_56", rng.integers(0, 1 << 16, (h, w)), dict(bpp=16, hsz=56, compress=3, masks=(0x00c0, 0x0038, 0x0007, 0))),
                 ("bgra32", rng.integers(0, 1 << 32, (h, w)), dict(bpp=32)), ("bgrx32_alpha0", rng.integers(0, 1 << 24, (h, w)), dict(bpp=32)),
                 ("bgra32_v4", rng.integers(0, 1 << 32, (h, w)), dict(bpp=32, hsz=108, compress=3, masks=(0xff0000, 0xff00, 0xff, 0xff000000))),
                 ("rle8_refused", rng.integers(0, 256, (h, w)), dict(bpp=8, compress=1, palette=pal256)),
                 ("same_masks_refused", rng.integers(0, 1 << 16, (h, w)), dict(bpp=16, compress=3, masks=(0x1f, 0x1f, 0x1f)))]
        for name, px, kw in cases:
            path = tmp_path / f"{name}_{w}.bmp"
            path.write_bytes(bmp_bytes(px, **kw))
            files.append(path)
    accepted = 0
    for path in files:
        want = ref_decode8(ref, path)
        if want is None:
            with pytest.raises(api.GptError):
                api.decode_image8(str(path))
            continue
        accepted += 1
        got = api.decode_image8(str(path))
        assert got.shape == want.shape, (path.name, got.shape, want.shape)
        assert np.array_equal(got, want), path.name
    assert accepted >= len(files) - 25 and accepted < len(files)


def tga_bytes(pixels_top_down, bpp, image_type, top_down=False, palette=None, pal_bits=24, pal_start=0, rle=False, id_text=b"", rng=None):
    """A TGA file: pixels_top_down [H, W] of ints (packed pixels or palette indices); rows are stored bottom-up unless top_down."""
    import struct
    rng = rng or np.random.default_rng(0)
    h, w = pixels_top_down.shape
    rows = pixels_top_down if top_down else pixels_top_down[::-1]
    nbytes = (bpp + 7) // 8
    px = [int(v).to_bytes(nbytes, "little") for v in rows.reshape(-1)]
    if rle:
        body, i = b"", 0
        while i < len(px):                                # packets may run across scanlines, as the format allows
            n = int(rng.integers(1, 9))
            n = min(n, len(px) - i)
            if rng.random() < 0.5:
                body += bytes([0x80 | (n - 1)]) + px[i]
                for k in range(n):
                    px[i + k] = px[i]                     # (a run repeats its first pixel: make the expected picture say so)
            else:
                body += bytes([n - 1]) + b"".join(px[i:i + n])
            i += n
        flat = np.array([int.from_bytes(b, "little") for b in px], dtype=np.int64).reshape(rows.shape)
        pixels_top_down[...] = flat if top_down else flat[::-1]
    else:
        body = b"".join(px)
    pal = b""
    if palette is not None:
        pal = b"\xee" * (0 if not pal_start else 0) + b"".join(int(v).to_bytes((pal_bits + 7) // 8, "little") for v in palette)
    header = struct.pack("<BBBHHBHHHHBB", len(id_text), 1 if palette is not None else 0, image_type + (8 if rle else 0),
                         0, len(palette) if palette is not None else 0, pal_bits if palette is not None else 0, 0, 0, w, h, bpp,
                         0x20 if top_down else 0)
    return header + id_text + pal + body


def test_tga_reader_equals_stb_image(ref, tmp_path):
    """TGA as stb_image 2.19 reads it: true colour 15 / 16 / 24 / 32 bits, grey 8 and grey + alpha 16, colour maps with 15 / 16 / 24 / 32-bit
    entries and 8- or 16-bit indices, run-length packets (also across scanlines), both row orders, an image-id field."""
    from PIL import Image
    rng = np.random.default_rng(15)
    files = []
    rgb = rng.integers(0, 256, (9, 14, 3), dtype=np.uint8)
    rgb[2:5] = rgb[2, 0]                                   # something for PIL's run-length coder
    for name, im, kw in (("pil_rgb", Image.fromarray(rgb), {}), ("pil_rgb_rle", Image.fromarray(rgb), {"compression": "tga_rle"}),
                         ("pil_rgba_rle", Image.fromarray(np.dstack([rgb, rgb[..., :1]]), "RGBA"), {"compression": "tga_rle"}),
                         ("pil_l", Image.fromarray(rgb[..., 0].copy(), "L"), {}), ("pil_p", Image.fromarray(rgb).quantize(30), {}),
                         ("pil_rgb_topdown", Image.fromarray(rgb), {"orientation": 1}), ("pil_la", Image.fromarray(rgb[..., :2].copy(), "LA"), {})):
        path = tmp_path / (name + ".tga")
        try:
            im.save(path, **kw)
        except Exception:
            continue
        files.append(path)
    for w, h in ((1, 1), (7, 5), (16, 3)):
        for rle in (False, True):
            for top_down in (False, True):
                tag = f"{w}x{h}{'_rle' if rle else ''}{'_td' if top_down else ''}"
                cases = [("c24", rng.integers(0, 1 << 24, (h, w)), dict(bpp=24, image_type=2)), ("c32", rng.integers(0, 1 << 32, (h, w)), dict(bpp=32, image_type=2)),
                         ("c16", rng.integers(0, 1 << 16, (h, w)), dict(bpp=16, image_type=2)), ("c15", rng.integers(0, 1 << 15, (h, w)), dict(bpp=15, image_type=2)),
                         ("g8", rng.integers(0, 256, (h, w)), dict(bpp=8, image_type=3)), ("ga16", rng.integers(0, 1 << 16, (h, w)), dict(bpp=16, image_type=3)),
                         ("m24", rng.integers(0, 40, (h, w)), dict(bpp=8, image_type=1, palette=rng.integers(0, 1 << 24, 37).tolist(), pal_bits=24)),
                         ("m32", rng.integers(0, 20, (h, w)), dict(bpp=8, image_type=1, palette=rng.integers(0, 1 << 32, 20).tolist(), pal_bits=32)),
                         ("m16", rng.integers(0, 20, (h, w)), dict(bpp=8, image_type=1, palette=rng.integers(0, 1 << 16, 20).tolist(), pal_bits=16)),
                         ("m15_idx16", rng.integers(0, 300, (h, w)), dict(bpp=16, image_type=1, palette=rng.integers(0, 1 << 15, 300).tolist(), pal_bits=15)),
                         ("c24_id", rng.integers(0, 1 << 24, (h, w)), dict(bpp=24, image_type=2, id_text=b"made by the test"))]
                for name, px, kw in cases:
                    path = tmp_path / f"{name}_{tag}.tga"
                    path.write_bytes(tga_bytes(px, top_down=top_down, rle=rle, rng=rng, **kw))
                    files.append(path)
    # a file cut short: stb_image hands back uninitialised rows, the product refuses it (as it refuses a header that claims more
    # pixels than the file could hold)
    whole = (tmp_path / "c24_7x5.tga").read_bytes()
    (tmp_path / "cut_short.tga").write_bytes(whole[:len(whole) - 40])
    (tmp_path / "bomb.tga").write_bytes(whole[:12] + (16000).to_bytes(2, "little") * 2 + whole[16:])
    for name in ("cut_short.tga", "bomb.tga"):
        with pytest.raises(api.GptError):
            api.decode_image8(str(tmp_path / name))
    for path in files:
        want = ref_decode8(ref, path)
        assert want is not None, path.name
        got = api.decode_image8(str(path))
        assert got.shape == want.shape, (path.name, got.shape, want.shape)
        assert np.array_equal(got, want), path.name
        if want.shape[2] != 2:
            assert np.array_equal(api.load_texture(str(path)), texels_like_the_reference(ref, path)), path.name
    assert len(files) > 130


def test_pnm_reader_equals_stb_image(ref, tmp_path):
    """Binary PGM / PPM as stb_image reads them (8-bit, samples unscaled, comments in the header)."""
    from PIL import Image
    rng = np.random.default_rng(16)
    rgb = rng.integers(0, 256, (7, 9, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "pil.ppm")
    Image.fromarray(rgb[..., 0].copy(), "L").save(tmp_path / "pil.pgm")
    (tmp_path / "comments.ppm").write_bytes(b"P6 # a comment\n# another\n9\t7 # size\r\n255\n" + rgb.tobytes())
    (tmp_path / "max100.pgm").write_bytes(b"P5\n9 7\n100\n" + (rgb[..., 0] % 101).tobytes())
    (tmp_path / "one.pgm").write_bytes(b"P5 1 1 255 " + bytes([77]))
    for name in ("pil.ppm", "pil.pgm", "comments.ppm", "max100.pgm", "one.pgm"):
        want = ref_decode8(ref, tmp_path / name)
        got = api.decode_image8(str(tmp_path / name))
        assert want is not None and got.shape == want.shape and np.array_equal(got, want), name
    (tmp_path / "deep.pgm").write_bytes(b"P5\n2 2\n65535\n" + bytes(8))
    (tmp_path / "ascii.pgm").write_bytes(b"P2\n2 2\n255\n1 2 3 4\n")
    for name in ("deep.pgm", "ascii.pgm"):
        assert ref_decode8(ref, tmp_path / name) is None
        with pytest.raises(api.GptError):
            api.decode_image8(str(tmp_path / name))


def test_cmyk_and_ycck_jpeg_equal_stb_image(ref, tmp_path):
    """Four-component JPEG files: Adobe CMYK (transform 0), YCCK (transform 2), and four components without a recognised Adobe marker
    (stb_image ignores the fourth) - all come out as three channels."""
    from PIL import Image
    rng = np.random.default_rng(19)
    y, x = np.mgrid[0:45, 0:61]
    img = np.stack([127 + 120 * np.sin(x / 6.0), 127 + 120 * np.cos(y / 5.0), (4 * x + 3 * y) % 256, 40 + (x * y) % 200], -1)
    img = np.clip(img + rng.normal(0, 5, img.shape), 0, 255).astype(np.uint8)
    n = 0
    for name, kw in (("cmyk_444", dict(quality=90, subsampling=0)), ("cmyk_420", dict(quality=75, subsampling=2)),
                     ("cmyk_progressive", dict(quality=80, subsampling=1, progressive=True))):
        base = tmp_path / (name + ".jpg")
        Image.fromarray(img, "CMYK").save(base, **kw)
        data = bytearray(base.read_bytes())
        at = data.find(b"Adobe")
        assert at > 0 and data[at - 4:at - 2] == b"\xff\xee"
        variants = {"": data}
        for tag, value in (("_as_ycck", 2), ("_as_ycc_alpha", 1)):
            v = bytearray(data)
            v[at + 11] = value                                  # the colour-transform byte of the Adobe segment
            variants[tag] = v
        v = bytearray(data)
        v[at:at + 5] = b"Adobf"                                 # not an Adobe marker any more
        variants["_unmarked"] = v
        for tag, blob in variants.items():
            path = tmp_path / (name + tag + ".jpg")
            path.write_bytes(bytes(blob))
            want = ref_decode8(ref, path)
            assert want is not None and want.shape == (45, 61, 3), path.name
            got = api.decode_image8(str(path))
            assert got.shape == want.shape and np.array_equal(got, want), path.name
            n += 1
    assert n == 12


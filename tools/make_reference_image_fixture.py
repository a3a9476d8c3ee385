"""tests/golden/reference_{heterogeneous,cornell_dof}_64.npy: renders the reference's author published
(result/*.png), box-filtered to 64x64.  reference_heterogeneous_64.npy: the reference's own published render of its default scene
(/root/reference/result/heterogeneous.png = scenes/cornell_box/scene.json: Volpath, 17 bounces, 100x100x40 density grid
inside a material-less box), box-filtered from 512x512 to 64x64 (float32 in [0,1], row 0 = top of the image).
An output of the reference, kept as data; tests/test_oracle_golden.py renders the same scene file with the oracle and
compares.  Also tests/golden/reference_density_grid.npz: the density grid of that scene (scenes/cornell_box/geometry/
density.d, 100 x 100 x 40 values with six decimals, stored as integer millionths) - input data of the reference, so
that the GPU box, which has no /root/reference, can render the same scene (tests/scenes.py: write_smoke_scene).
Run where /root/reference exists:  python tools/make_reference_image_fixture.py"""
import os, struct, sys, zlib
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_png_rgb8(path):
    b = open(path, "rb").read()
    pos, idat = 8, b""
    while pos < len(b):
        n, = struct.unpack(">I", b[pos:pos + 4])
        t, d = b[pos + 4:pos + 8], b[pos + 8:pos + 8 + n]
        pos += 12 + n
        if t == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", d[:10])
        elif t == b"IDAT":
            idat += d
    assert depth == 8 and ctype in (2, 6)
    ch = 3 if ctype == 2 else 4
    raw, stride = zlib.decompress(idat), w * ch
    img = np.zeros((h, stride), np.int32)
    prev = np.zeros(stride, np.int32)
    p = 0
    for y in range(h):
        f = raw[p]
        line = np.frombuffer(raw[p + 1:p + 1 + stride], np.uint8).astype(np.int32)
        p += 1 + stride
        if f == 0:
            out = line
        elif f == 2:
            out = (line + prev) & 255
        else:
            out = np.zeros(stride, np.int32)
            for i in range(stride):
                a = out[i - ch] if i >= ch else 0
                up = prev[i]
                c = prev[i - ch] if i >= ch else 0
                if f == 1:
                    pr = a
                elif f == 3:
                    pr = (a + up) >> 1
                else:
                    pa, pb, pc = abs(up - c), abs(a - c), abs(a + up - 2 * c)
                    pr = a if (pa <= pb and pa <= pc) else (up if pb <= pc else c)
                out[i] = (line[i] + pr) & 255
        img[y] = out
        prev = out
    return img.reshape(h, w, ch)[:, :, :3].astype(np.uint8)


def box_filtered(name):
    img = read_png_rgb8("/root/reference/result/%s.png" % name).astype(np.float64) / 255.0
    assert img.shape == (512, 512, 3)
    small = img.reshape(64, 8, 64, 8, 3).mean(axis=(1, 3)).astype(np.float32)
    out = os.path.join(ROOT, "tests", "golden", "reference_%s_64.npy" % name)
    np.save(out, small)
    print("wrote", out, small.shape, "channel means", small.mean(axis=(0, 1)))


if __name__ == "__main__":
    box_filtered("heterogeneous")
    # result/cornell_dof.png: the Cornell box with its two boxes (the geometry of BASELINE config 1 / 2) through the
    # thin-lens camera.  (result/volume_caustic.png cannot serve as a pin: the shipped vol_caustic.json names a 5 x 4 mm light
    # mesh while the picture shows the Cornell light, and with that light (tests/scenes.py: write_vol_caustic_scene) Volpath
    # converges, for every maxDepth >= 33, to a frame 7 - 12 % darker than the picture: its scene differs in something the
    # repository does not record.)
    box_filtered("cornell_dof")
    d = np.loadtxt("/root/reference/scenes/cornell_box/geometry/density.d", dtype=np.float64)
    q = np.round(d * 1e6).astype(np.int32)
    assert q.size == 100 * 100 * 40 and np.abs(q / 1e6 - d).max() < 1e-9
    out = os.path.join(ROOT, "tests", "golden", "reference_density_grid.npz")
    np.savez_compressed(out, millionths=q, nx=100, ny=100, nz=40)
    print("wrote", out, os.path.getsize(out), "bytes")

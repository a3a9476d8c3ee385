"""ctypes binding of libgpt.so (include/gpt.h) — the C ABI behind the reference's
BeginRender / Render / EndRender (reference src/pathtracer.h:10-12).

Plumbing only.  There is no Python or CPU implementation behind these calls:
if the HIP library is missing or no GPU is visible, they raise.
"""
import ctypes as C
import os

import numpy as np

from . import scene_types as st

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPT_LIB_PATH") or os.path.join(_HERE, "libgpt.so")   # override: kernel experiments only
# The library itself reads no environment variable.  What this binding takes from the environment (the path of an
# experimental build) is recorded here, and bench.py prints it in its JSON line (config.env_overrides).
ENV_OVERRIDES = {k: os.environ[k] for k in ("GPT_LIB_PATH", "GPT_ALLOW_OLD_LIB") if os.environ.get(k)}
# Renderer options (gpt_set_option) applied to every Renderer this process creates, e.g. {"lds_scene": 0} to run a whole
# test suite through the global-memory kernels (pytest --gpt-opt lds_scene=0).  Empty = the library's defaults.
DEFAULT_OPTIONS = {}

_lib = None


class GptError(RuntimeError):
    pass


def load():
    """Load libgpt.so (built in-tree by `make -C gpu_pathtracer_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GptError(f"{LIB_PATH} is missing: build it with __graft_entry__.build(); there is no fallback path")
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.  If libgpt.so pulled in the
    # system copy first, a later torch.cuda initialisation in the same process fails ("No HIP GPUs are
    # available").  Loading torch first makes both share the runtime torch ships.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32, f32 = C.c_void_p, C.c_int32, C.c_uint32, C.c_float
    lib.gpt_last_error.restype = C.c_char_p
    lib.gpt_version.restype = C.c_char_p
    sig = {
        "gpt_begin": [vp, u32, u32, f32, C.c_int, C.POINTER(vp)],
        "gpt_set_tile_owner": [vp, C.c_int, C.c_int],
        "gpt_set_integrator": [vp, i32, i32, C.c_float],
        "gpt_set_traversal_order": [vp, i32],
        "gpt_set_option": [vp, C.c_char_p, C.c_int64],
        "gpt_get_option": [vp, C.c_char_p, C.POINTER(C.c_int64)],
        "gpt_scene_load_cached": [C.c_char_p, C.c_int, C.POINTER(vp)],
        "gpt_scene_load_ex": [C.c_char_p, C.c_int, C.POINTER(vp)],
        "gpt_sbvh_build": [vp, i32, f32, vp, i32, C.POINTER(i32), vp, vp, i32, C.POINTER(i32), vp],
        "gpt_render": [vp, vp, u32, u32, C.c_int, vp],
        "gpt_tonemap": [vp, u32, C.c_int, vp],
        "gpt_tonemap_from": [vp, vp, u32, C.c_int, vp],
        "gpt_comm_unique_id": [vp],
        "gpt_comm_init": [vp, C.c_int, C.c_int, vp],
        "gpt_reduce_film": [vp, C.c_int],
        "gpt_read_reduced": [vp, vp],
        "gpt_comm_destroy": [vp],
        "gpt_synchronize": [vp],
        "gpt_read_accum": [vp, vp],
        "gpt_read_color": [vp, vp],
        "gpt_write_state": [vp, vp, vp],
        "gpt_copy_to_host": [vp, vp, vp, C.c_size_t],
        "gpt_bind_film": [vp, vp, vp],
        "gpt_end": [vp],
        "gpt_kernel_time": [vp, C.POINTER(u32), C.POINTER(C.c_double)],
        "gpt_kernel_time_reset": [vp],
        "gpt_enable_counters": [vp, C.c_int],
        "gpt_read_counters": [vp, vp],
        "gpt_read_probe_counters": [vp, vp],
        "gpt_debug_trace": [vp, vp, C.c_int, vp, vp],
        "gpt_debug_math": [C.c_int, C.c_int, vp, vp, vp, C.c_int],
        "gpt_debug_rng": [C.c_int, u32, u32, vp, vp, C.c_int],
        "gpt_debug_bsdf": [C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp],
        "gpt_debug_fail_next_wide_alloc": [C.c_int],
        "gpt_bvh_build": [vp, i32, vp, vp, C.POINTER(i32), vp],
        "gpt_light_distribution": [vp, i32, vp, vp, C.POINTER(i32)],
        "gpt_infinite_init": [vp, vp],
        "gpt_camera_init": [vp, vp, vp, vp, f32, f32, f32, f32, f32, f32, C.c_int, C.c_int],
        "gpt_scene_load": [C.c_char_p, C.POINTER(vp)],
        "gpt_scene_get_desc": [vp, vp],
        "gpt_scene_get_config": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(f32), vp],
        "gpt_scene_set_integrator": [vp, i32, i32],
        "gpt_scene_free": [vp],
        "gpt_save_png": [C.c_char_p, i32, i32, vp],
        "gpt_save_pfm": [C.c_char_p, i32, i32, vp],
        "gpt_save_exr": [C.c_char_p, i32, i32, vp],
        "gpt_decode_image8": [C.c_char_p, vp, vp, vp, vp, C.c_int64],
        "gpt_load_texture": [C.c_char_p, vp, vp, vp, C.c_int64],
        "gpt_load_exr": [C.c_char_p, vp, vp, vp, C.c_int64],
    }
    for name, args in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            # kernel A/B runs load older builds through GPT_LIB_PATH; the product library must export everything
            if os.environ.get("GPT_LIB_PATH") and os.environ.get("GPT_ALLOW_OLD_LIB"):
                continue
            raise
        fn.argtypes = args
        fn.restype = C.c_int
    for name in ("gpt_accum_device_ptr", "gpt_color_device_ptr", "gpt_reduced_device_ptr"):
        fn = getattr(lib, name)
        fn.argtypes = [vp]
        fn.restype = vp
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise GptError(f"gpt error {rc}: {load().gpt_last_error().decode()}")


# ---- host-side preparation (CPU) ----------------------------------------------

def bvh_build(prims):
    lib = load()
    prims = np.ascontiguousarray(prims)
    n = len(prims)
    out = np.zeros(n, dtype=st.PRIMITIVE)
    nodes = np.zeros(max(1, 2 * n), dtype=st.BVH_NODE)
    box = np.zeros(6, dtype=np.float32)
    nn = C.c_int32(0)
    check(lib.gpt_bvh_build(st.ptr(prims), n, st.ptr(out), st.ptr(nodes), C.byref(nn), st.ptr(box)))
    return out, nodes[: nn.value].copy(), box


def sbvh_build(prims, alpha=1e-5, capacity=None):
    """gpt_sbvh_build: the split BVH (object + spatial splits, duplicated references) -> (prims, nodes, box, origin index)"""
    lib = load()
    prims = np.ascontiguousarray(prims)
    n = len(prims)
    cap = int(capacity or 2 * n + 64)
    out = np.zeros(cap, dtype=st.PRIMITIVE)
    orig = np.zeros(cap, dtype=np.int32)
    nodes = np.zeros(max(1, 2 * cap), dtype=st.BVH_NODE)
    box = np.zeros(6, dtype=np.float32)
    nn, npr = C.c_int32(0), C.c_int32(0)
    check(lib.gpt_sbvh_build(st.ptr(prims), n, float(alpha), st.ptr(out), cap, C.byref(npr), st.ptr(orig), st.ptr(nodes), len(nodes), C.byref(nn), st.ptr(box)))
    return out[: npr.value].copy(), nodes[: nn.value].copy(), box, orig[: npr.value].copy()


def light_distribution(lights, infinite=None):
    lib = load()
    cdf = np.zeros(len(lights) + 2, dtype=np.float32)
    n = C.c_int32(0)
    check(lib.gpt_light_distribution(st.ptr(lights), len(lights), C.byref(infinite) if infinite is not None else None,
                                     st.ptr(cdf), C.byref(n)))
    return cdf[: n.value].copy()


def camera_init(position, lookat, up=(0, 1, 0), res=(512, 512), fov=60.0, aperture=0.0, focal=0.0, distance=0.1,
                filmic=True, environment=False):
    cam = st.Camera()
    p, la, u = (C.c_float * 3)(*position), (C.c_float * 3)(*lookat), (C.c_float * 3)(*up)
    check(load().gpt_camera_init(C.byref(cam), p, la, u, float(res[0]), float(res[1]), float(distance), float(fov),
                                 float(aperture), float(focal), int(filmic), int(environment)))
    return cam


class LoadedScene:
    """gpt_scene_load: LoadScene + InitScene of the reference (src/parsescene.cpp:45, src/main.cpp:261-278)."""

    def __init__(self, json_path, use_bvh_cache=False, sbvh=False, reference_bvh=False):
        """sbvh: GPT_LOAD_SBVH, reference_bvh: GPT_LOAD_REFERENCE_BVH; neither: the reference builder's tree unless it has oversized leaves (gpt.h)"""
        self.lib = load()
        self.handle = C.c_void_p()
        check(self.lib.gpt_scene_load_ex(os.fsencode(json_path), (1 if use_bvh_cache else 0) | (2 if sbvh else 0) | (4 if reference_bvh else 0),
                                         C.byref(self.handle)))
        self.desc = st.SceneDesc()
        check(self.lib.gpt_scene_get_desc(self.handle, C.byref(self.desc)))
        w, h, eps = C.c_int32(), C.c_int32(), C.c_float()
        self.camera = st.Camera()
        check(self.lib.gpt_scene_get_config(self.handle, C.byref(w), C.byref(h), C.byref(eps), C.byref(self.camera)))
        self.width, self.height, self.epsilon = w.value, h.value, eps.value

    def set_integrator(self, integrator_type, max_depth):
        check(self.lib.gpt_scene_set_integrator(self.handle, integrator_type, max_depth))
        check(self.lib.gpt_scene_get_desc(self.handle, C.byref(self.desc)))

    def array(self, name, count_name, dtype):
        n = getattr(self.desc, count_name)
        ptr_ = getattr(self.desc, name)
        if not n or not ptr_:
            return np.zeros(0, dtype=dtype)
        raw = np.ctypeslib.as_array(C.cast(ptr_, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dtype).itemsize,))
        return raw.view(dtype)

    def close(self):
        if self.handle:
            self.lib.gpt_scene_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def save_png(path, width, height, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    check(load().gpt_save_png(os.fsencode(path), width, height, st.ptr(rgb)))


def save_exr(path, width, height, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    check(load().gpt_save_exr(os.fsencode(path), width, height, st.ptr(rgb)))


def decode_image8(path):
    """gpt_decode_image8: PNG / JPEG -> uint8 [H, W, components], row 0 = bottom (stb_image with flip-on-load)."""
    w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
    check(load().gpt_decode_image8(os.fsencode(path), C.byref(w), C.byref(h), C.byref(c), None, 0))
    out = np.empty((h.value, w.value, c.value), np.uint8)
    check(load().gpt_decode_image8(os.fsencode(path), C.byref(w), C.byref(h), C.byref(c), st.ptr(out), out.size))
    return out


def load_texture(path):
    """gpt_load_texture: the texels the kernel samples, uint8 [H, W, 4]."""
    w, h = C.c_int32(), C.c_int32()
    check(load().gpt_load_texture(os.fsencode(path), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 4), np.uint8)
    check(load().gpt_load_texture(os.fsencode(path), C.byref(w), C.byref(h), st.ptr(out), out.size // 4))
    return out


def load_exr(path):
    """gpt_load_exr: float32 [H, W, 3], row 0 = top."""
    w, h = C.c_int32(), C.c_int32()
    check(load().gpt_load_exr(os.fsencode(path), C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 3), np.float32)
    check(load().gpt_load_exr(os.fsencode(path), C.byref(w), C.byref(h), st.ptr(out), out.size))
    return out


def save_pfm(path, width, height, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    check(load().gpt_save_pfm(os.fsencode(path), width, height, st.ptr(rgb)))


# ---- renderer --------------------------------------------------------------------

class Renderer:
    """gpt_begin .. gpt_end around one scene; mirrors BeginRender/Render/EndRender."""

    def __init__(self, desc, width, height, epsilon, device=0):
        self.lib = load()
        self.width, self.height = int(width), int(height)
        self.ctx = C.c_void_p()
        check(self.lib.gpt_begin(C.byref(desc), self.width, self.height, float(epsilon), int(device), C.byref(self.ctx)))
        self.options_set = {}
        for k, v in DEFAULT_OPTIONS.items():
            self.set_option(k, v)

    def set_option(self, name, value):
        """gpt_set_option: "lds_scene", "vpt_walk_kernel", "max_batch", "chunk_iters" (include/gpt.h)"""
        check(self.lib.gpt_set_option(self.ctx, name.encode(), int(value)))
        self.options_set[name] = int(value)

    def get_option(self, name):
        v = C.c_int64(0)
        check(self.lib.gpt_get_option(self.ctx, name.encode(), C.byref(v)))
        return v.value

    def set_tile_owner(self, rank, n_ranks):
        check(self.lib.gpt_set_tile_owner(self.ctx, rank, n_ranks))

    def set_traversal_order(self, order):
        """"reference" / 0: the reference's order on its binary tree; "wide" / 2: the 4-wide tree walked one lane per ray
        (include/gpt_wide_bvh.h); "auto" / -1: gpt_begin's choice again - "wide" for scenes that do not fit LDS, "reference"
        otherwise.  get_option("traversal_order") reads the order in force."""
        code = {"reference": 0, "wide": 2, "auto": -1}.get(order, order)
        check(self.lib.gpt_set_traversal_order(self.ctx, int(code)))

    def set_integrator(self, kind, value):
        """"pt": value = maxDepth; "ao": value = maxDist (the reference reads both from the scene on every Render call)"""
        if kind == "pt":
            check(self.lib.gpt_set_integrator(self.ctx, st.IT_PT, int(value), 0.0))
        elif kind == "vpt":
            check(self.lib.gpt_set_integrator(self.ctx, st.IT_VPT, int(value), 0.0))
        elif kind == "ao":
            check(self.lib.gpt_set_integrator(self.ctx, st.IT_AO, 0, float(value)))
        else:
            raise ValueError(f"integrator {kind!r} is not supported (pt, vpt, ao)")

    def render(self, camera, iter_first, iter_count, reset=False, out_dev=None):
        check(self.lib.gpt_render(self.ctx, C.byref(camera), int(iter_first), int(iter_count), int(bool(reset)), out_dev))

    def tonemap(self, iteration, filmic, out_dev):
        check(self.lib.gpt_tonemap(self.ctx, int(iteration), int(bool(filmic)), out_dev))

    def tonemap_from(self, acc_dev, iteration, filmic, out_dev):
        check(self.lib.gpt_tonemap_from(self.ctx, acc_dev, int(iteration), int(bool(filmic)), out_dev))

    def synchronize(self):
        check(self.lib.gpt_synchronize(self.ctx))

    # ---- multi-GPU film reduce (RCCL inside the library) ----
    def comm_init(self, rank, n_ranks, unique_id):
        """unique_id: the 128 bytes rank 0 got from comm_unique_id(); also sets the tile ownership"""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        check(self.lib.gpt_comm_init(self.ctx, int(rank), int(n_ranks), buf))

    def comm_destroy(self):
        check(self.lib.gpt_comm_destroy(self.ctx))

    def reduce_film(self, root=0):
        check(self.lib.gpt_reduce_film(self.ctx, int(root)))

    def reduced_ptr(self):
        return self.lib.gpt_reduced_device_ptr(self.ctx)

    def read_reduced(self):
        a = np.empty(self.width * self.height * 3, dtype=np.float32)
        check(self.lib.gpt_read_reduced(self.ctx, st.ptr(a)))
        return a

    def accum_ptr(self):
        return self.lib.gpt_accum_device_ptr(self.ctx)

    def color_ptr(self):
        return self.lib.gpt_color_device_ptr(self.ctx)

    def read_accum(self):
        a = np.empty(self.width * self.height * 3, dtype=np.float32)
        check(self.lib.gpt_read_accum(self.ctx, st.ptr(a)))
        return a

    def read_color(self):
        a = np.empty(self.width * self.height * 3, dtype=np.float32)
        check(self.lib.gpt_read_color(self.ctx, st.ptr(a)))
        return a

    def read_device(self, dev_ptr, n_floats):
        a = np.empty(n_floats, dtype=np.float32)
        check(self.lib.gpt_copy_to_host(self.ctx, dev_ptr, st.ptr(a), n_floats))
        return a

    def write_state(self, acc, color):
        acc = np.ascontiguousarray(acc, dtype=np.float32)
        color = np.ascontiguousarray(color, dtype=np.float32)
        check(self.lib.gpt_write_state(self.ctx, st.ptr(acc), st.ptr(color)))

    def bind_film(self, acc_dev=None, color_dev=None):
        check(self.lib.gpt_bind_film(self.ctx, acc_dev, color_dev))

    def trace_rays(self, rays8):
        """gpt_debug_trace: rays8 (n, 8) float32 = origin, direction, tmax, any_hit -> (prim (n,) int32, tb (n, 3) float32)"""
        rays8 = np.ascontiguousarray(rays8, dtype=np.float32).reshape(-1, 8)
        n = len(rays8)
        prim = np.zeros(n, dtype=np.int32)
        tb = np.zeros((n, 3), dtype=np.float32)
        check(self.lib.gpt_debug_trace(self.ctx, st.ptr(rays8), n, st.ptr(prim), st.ptr(tb)))
        return prim, tb

    def kernel_time(self):
        n, ms = C.c_uint32(0), C.c_double(0)
        check(self.lib.gpt_kernel_time(self.ctx, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def kernel_time_reset(self):
        check(self.lib.gpt_kernel_time_reset(self.ctx))

    def enable_counters(self, on=True):
        check(self.lib.gpt_enable_counters(self.ctx, int(on)))

    def read_counters(self):
        c = np.zeros(6, dtype=np.uint64)
        check(self.lib.gpt_read_counters(self.ctx, st.ptr(c)))
        names = ["node_visits", "prim_tests", "bounce_iters", "shadow_rays", "closest_rays", "samples"]
        return dict(zip(names, map(int, c)))

    def read_probe_counters(self):
        c = np.zeros(16, dtype=np.uint64)
        check(self.lib.gpt_read_probe_counters(self.ctx, st.ptr(c)))
        names = ["node_visits", "prim_tests", "bounce_iters", "shadow_rays", "closest_rays", "samples",
                 "w_node", "w_prim", "w_trip", "l_trip", "cyc_direct", "cyc_hit", "cyc_regen", "unused13", "cyc_trace", "cyc_shade"]
        return dict(zip(names, map(int, c)))

    def close(self):
        if self.ctx:
            self.lib.gpt_end(self.ctx)
            self.ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0; hand the bytes to every rank)"""
    buf = (C.c_char * 128)()
    check(load().gpt_comm_unique_id(buf))
    return bytes(buf)


def debug_math(fn, x, y=None, device=0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y if y is not None else x, dtype=np.float32)
    out = np.empty_like(x)
    check(load().gpt_debug_math(device, fn, st.ptr(x), st.ptr(y), st.ptr(out), len(x)))
    return out


def debug_bsdf(material, geom11, in3, mode, texture=None, device=0):
    """SampleBSDF (mode 1, in3 = draws) / Fr (mode 0, in3 = direction) on the device; `texture`: an (H, W, 4) uint8 array or None.
    Returns (n, 7): out.xyz, fr.xyz, pdf."""
    geom11 = np.ascontiguousarray(geom11, dtype=np.float32)
    in3 = np.ascontiguousarray(in3, dtype=np.float32)
    out = np.zeros((len(geom11), 7), dtype=np.float32)
    rec = None
    if texture is not None:
        texture = np.ascontiguousarray(texture, dtype=np.uint8)
        rec = st.Texture()
        rec.data, rec.height, rec.width = texture.ctypes.data, texture.shape[0], texture.shape[1]
    check(load().gpt_debug_bsdf(device, st.ptr(material), C.byref(rec) if rec is not None else None, st.ptr(geom11), st.ptr(in3),
                                len(geom11), mode, st.ptr(out)))
    return out


def last_error():
    """the message of the last failed call on this thread (gpt_last_error); also kept by gpt_begin's low-memory fall-back"""
    return load().gpt_last_error().decode()


def debug_fail_next_wide_alloc(enable=True):
    check(load().gpt_debug_fail_next_wide_alloc(1 if enable else 0))


def debug_rng(pixel, iteration, n, device=0):
    seed = C.c_uint32(0)
    u = np.zeros(n, dtype=np.float32)
    check(load().gpt_debug_rng(device, pixel, iteration, C.byref(seed), st.ptr(u), n))
    return seed.value, u

"""Host-side scene assembly on top of the C ABI: what the reference does between LoadScene() and
BeginRender() (reference src/parsescene.cpp:492-541 light records, src/scene.h:50-83 Scene::Init).

All arithmetic happens in libgpt.so (gpt_bvh_build, gpt_light_distribution, gpt_infinite_init,
gpt_camera_init); this module only owns the numpy buffers and fills the gpt_scene_desc.
"""
import ctypes as C
import json

import numpy as np

from . import api
from . import scene_types as st


class HostScene:
    def __init__(self, prims, materials, light_radiance, max_depth, env=None, env_uvw=None, textures=None):
        prims = np.ascontiguousarray(prims)
        self.materials = np.ascontiguousarray(materials)
        lidx = prims["triangle"]["lightIdx"]
        n_lights = int(lidx.max()) + 1 if (lidx >= 0).any() else 0
        self.lights = np.zeros(n_lights, dtype=st.AREA)      # Area{radiance, triangle, medium}, parsescene.cpp:531-536
        for i in np.nonzero(lidx >= 0)[0]:
            li = int(lidx[i])
            self.lights[li]["triangle"] = prims[i]["triangle"]
            rad = light_radiance[li] if np.ndim(light_radiance) == 2 else light_radiance
            self.lights[li]["radiance"] = st.f3(rad)
            self.lights[li]["medium"] = -1
        self.prims, self.nodes, self.root_box = api.bvh_build(prims)
        self.infinite = None
        self.env = None
        if env is not None:
            self.env = np.ascontiguousarray(env, dtype=np.float32)
            inf = st.Infinite()
            inf.data = self.env.ctypes.data
            inf.height, inf.width = self.env.shape[0], self.env.shape[1]
            u, v, w = env_uvw if env_uvw is not None else ((1, 0, 0), (0, 1, 0), (0, 0, 1))
            inf.u, inf.v, inf.w = st.Float3(*u), st.Float3(*v), st.Float3(*w)
            inf.isvalid = 1
            api.check(api.load().gpt_infinite_init(C.byref(inf), st.ptr(self.root_box)))
            self.infinite = inf
        self.cdf = api.light_distribution(self.lights, self.infinite)
        self.textures = [np.ascontiguousarray(t, dtype=np.uint8) for t in (textures or [])]
        self._tex = (st.Texture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            self._tex[i].data = t.ctypes.data
            self._tex[i].height, self._tex[i].width = t.shape[0], t.shape[1]
        d = st.SceneDesc()
        d.prims, d.n_prims = st.ptr(self.prims), len(self.prims)
        d.nodes, d.n_nodes = st.ptr(self.nodes), len(self.nodes)
        d.materials, d.n_materials = st.ptr(self.materials), len(self.materials)
        d.lights, d.n_lights = st.ptr(self.lights), len(self.lights)
        d.light_distribution, d.n_light_distribution = st.ptr(self.cdf), len(self.cdf)
        d.infinite = C.addressof(self.infinite) if self.infinite is not None else None
        d.textures = C.addressof(self._tex) if self.textures else None
        d.n_textures = len(self.textures)
        d.integrator_type = st.IT_PT
        d.max_depth = int(max_depth)
        self.desc = d


def load_baked(path, max_depth):
    """A baked triangle soup (tools/bake_*.py): Primitive records before BVH ordering + material table."""
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    prims = z["prims"].view(st.PRIMITIVE)
    materials = z["materials"].view(st.MATERIAL)
    return HostScene(prims, materials, meta["light_radiance"], max_depth), meta


def camera_from_meta(meta, width, height):
    c = meta["camera"]
    return api.camera_init(c["position"], c["lookat"], c["up"], (width, height), c["fov"], c["apertureRadius"],
                           c["focalDistance"], c["distance"], c["filmic"])


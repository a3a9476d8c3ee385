"""Msamples/s of the GPT_TRAVERSAL_WIDE4 kernels on the c3 / c5 stand-ins for each var/libgpt_<name>.so given
(kernel experiments: tools/build_variant.sh <name> -DPT_WIDE_...=...).  usage (GPU box): python tools/gpu_wide_variants.py name..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, tempfile, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import scenes
from gpu_pathtracer_amd import api
out = []
for which, spp in (("c3", 16), ("c4", 16), ("c5", 4)):
    ls = api.LoadedScene(scenes.write_standin_scene(tempfile.mkdtemp(), which))
    with api.Renderer(ls.desc, ls.width, ls.height, ls.epsilon) as r:
        r.set_traversal_order(sys.argv[1])
        r.render(ls.camera, 1, 2, reset=True); r.synchronize()
        best = 1e9
        for rep in range(3):
            r.kernel_time_reset(); r.render(ls.camera, 1, spp, reset=True); r.synchronize()
            best = min(best, r.kernel_time()[1])
        h = hashlib.sha1(r.read_accum().tobytes()).hexdigest()[:8]
    out.append("%%s %%7.1f Ms/s (%%s)" %% (which, ls.width * ls.height * spp / best / 1e3, h))
print("RESULT", " | ".join(out))
''' % (ROOT, ROOT)
for name in sys.argv[1:]:
    mode = "wide"
    if ":" in name:
        name, mode = name.split(":")
    env = dict(os.environ)
    if name != "cur":
        env.update(GPT_LIB_PATH=os.path.join(ROOT, "var", f"libgpt_{name}.so"), GPT_ALLOW_OLD_LIB="1")
    o = subprocess.run([sys.executable, "-c", CHILD, mode], env=env, capture_output=True, text=True)
    res = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
    print(f"{name:24s} {mode:9s}", res[0][7:] if res else "FAILED " + o.stderr[-300:], flush=True)

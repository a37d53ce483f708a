"""Scratch (needs the instrumented variant library): wave-rounds and lane-rounds of the two phases of the mesh walk on the stress scene."""
import os, sys, ctypes as C
os.environ["WBRT_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opencl-ray-tracer_amd", "variants", "libdbg.so")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg
import numpy as np
scene = pkg.scene_stress(500, 224, 224, 16, 16)
cam = pkg.camera_stress(1920, 8, 50)
bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 11))
ctx.tune("lpt", 0)
out = (C.c_ulonglong * 4)()
for mode, name in ((pkg.MESH_BVH, "proven"), (pkg.MESH_BVH_CAPPED, "capped")):
    ctx.set_mesh_mode(mode)
    ctx.render_frame(cam)
    pkg.lib().wbrt_dbg_rounds_(out, 1)
    ctx.set_timing(True); ctx.render_frame(cam); ms, n = ctx.get_timing(); ctx.set_timing(False)
    pkg.lib().wbrt_dbg_rounds_(out, 1)
    a_w, a_l, b_w, b_l = [int(x) for x in out]
    print("%s: %.1f ms; box-tree phase: %d wave-rounds, %.1f%% lanes busy; cooperative phase: %d wave-rounds, %.1f%% lanes busy" % (
        name, ms, a_w, 100.0 * a_l / max(a_w * 64, 1), b_w, 100.0 * b_l / max(b_w * 64, 1)), flush=True)
ctx.close()

"""Scratch (needs an instrumented variant library, WBRT_LIB): wave-rounds and lane-rounds of the cooperative walk on
the f1_car scene and the stress scene.  build: make -C opencl-ray-tracer_amd/csrc ../variants/walk_stats.so; usage: WBRT_LIB=$PWD/opencl-ray-tracer_amd/variants/walk_stats.so python scripts/dbg_rounds.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, scene_main_cpp_with_mesh
import numpy as np
out = (C.c_ulonglong * 8)()
for sname in ("f1_car", "stress"):
    if sname == "stress":
        scene, cam = pkg.scene_stress(500, 224, 224, 16, 16), pkg.camera_stress(1920, 8, 50)
    else:
        scene, cam = scene_main_cpp_with_mesh("f1_car"), pkg.camera_main_cpp(1920, 8, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 11))
    ctx.tune("lpt", 0)
    for mode, name in ((pkg.MESH_BVH, "proven"), (pkg.MESH_BVH_CAPPED, "capped")):
        ctx.set_mesh_mode(mode)
        ctx.render_frame(cam)
        pkg.lib().wbrt_dbg_rounds_(out, 1)
        ctx.set_timing(True); ctx.render_frame(cam); ms, n = ctx.get_timing(); ctx.set_timing(False)
        pkg.lib().wbrt_dbg_rounds_(out, 1)
        r, busy, inner, leaf, rin, sleaf, sinner, top = [int(x) for x in out[:8]]
        print("%s %s: %.1f ms; %.2f M wave-rounds (%.2f M with inner nodes), lanes busy %.1f%%; lane visits: inner %.1f M (sphere hierarchy %.1f M), leaf slots %.1f M (spheres %.1f M); %.2f M wave-rounds over the hierarchy of the boxes" % (
            sname, name, ms, r / 1e6, rin / 1e6, 100.0 * busy / max(r * 64, 1), inner / 1e6, sinner / 1e6, leaf / 1e6, sleaf / 1e6, top / 1e6), flush=True)
    ctx.close()

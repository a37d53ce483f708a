"""Scratch: cold frames of the headline and the mesh configuration for pilot depths."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, scene_main_cpp_with_mesh
for name in ("readme", "mesh"):
    if name == "readme":
        scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50)
    else:
        scene = scene_main_cpp_with_mesh("f1_car"); cam = pkg.camera_main_cpp(1920, 100, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
    ctx.set_schedule_reuse(False)
    for pd in (3, 5, 8, 12):
        ctx.tune("pilot_depth", pd)
        for _ in range(2): ctx.render_frame(cam)
        ctx.set_timing(True)
        for _ in range(6): ctx.render_frame(cam)
        ms, n = ctx.get_timing(); ctx.set_timing(False)
        print("%s cold frame, pilot depth %d: %.3f ms" % (name, pd, ms / n), flush=True)
    ctx.close()

"""Scratch: cold frames of the mesh configuration for pilot lengths."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, scene_main_cpp_with_mesh
scene = scene_main_cpp_with_mesh("f1_car"); cam = pkg.camera_main_cpp(1920, 100, 50)
bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
ctx.set_schedule_reuse(False)
for ps in (1, 2, 3, 4, 6, 8):
    ctx.tune("pilot_spp", ps)
    for _ in range(2): ctx.render_frame(cam)
    ctx.set_timing(True)
    for _ in range(5): ctx.render_frame(cam)
    ms, n = ctx.get_timing(); ctx.set_timing(False)
    print("mesh cold frame, pilot %d spp (at most a 24th of 100 = 4): %.3f ms" % (ps, ms / n), flush=True)
ctx.close()

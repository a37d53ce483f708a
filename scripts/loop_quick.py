"""Scratch: the f1_car and burger scenes in WBRT_MESH_LOOP (the reference's triangle loop, wave-cooperative): time and rate."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, scene_main_cpp_with_mesh, bits
for mesh, spp in (("f1_car", 4), ("burger", 16)):
    scene = scene_main_cpp_with_mesh(mesh); cam = pkg.camera_main_cpp(1920, spp, 50); bw, bh = pkg.set_blocks(cam, 4)
    ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 11))
    ctx.set_mesh_mode(pkg.MESH_LOOP); ctx.set_schedule_reuse(False)
    ctx.render_frame(cam)
    ctx.set_timing(True); img = ctx.render_frame(cam); ms, n = ctx.get_timing(); ctx.set_timing(False)
    c = ctx.count_events(cam); fl = pkg.algorithmic_flops(c)
    print("%s loop mode 1920x1080 x %d spp: %.1f ms, %.3e tri tests/s, %.2f TFLOP/s algorithmic" % (mesh, spp, ms, c["tri_tests"] / ms * 1e3, fl / ms / 1e9), flush=True)
    ctx.close()

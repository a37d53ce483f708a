"""Scratch: times the mesh scene (main.cpp spheres + f1_car in one bounding box) at reduced size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, orc, scene_main_cpp_with_mesh, bits
W = int(sys.argv[1]) if len(sys.argv) > 1 else 480
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scene = scene_main_cpp_with_mesh("f1_car")
cam = pkg.camera_main_cpp(W, SPP, 50)
bw, bh = pkg.set_blocks(cam, 4)
seeds = pkg.make_seeds(bw * bh, 11)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(seeds)
img = ctx.render_frame(cam)
ctx.set_timing(True)
t = time.time(); img = ctx.render_frame(cam); dt = time.time() - t
ms, n = ctx.get_timing()
paths = cam.width * cam.height * SPP
print("C3 %dx%d %d spp: kernel %.1f ms (%d launches), %.2f Mpaths/s" % (cam.width, cam.height, SPP, ms, n, paths / ms / 1e3), flush=True)
c = ctx.count_events(cam)
fl = pkg.algorithmic_flops(c)
print("tri tests/path %.0f, flops/path %.0f, %.2f TFLOP/s algorithmic, tri tests/s %.3e" % (c["tri_tests"] / paths, fl / paths, fl / ms / 1e9, c["tri_tests"] / ms * 1e3))
print({k: v for k, v in c.items() if k.startswith(("tri", "bbox", "seg", "path"))})
if W <= 128:
    ref = orc.Oracle(scene).render_frame(cam, seeds)
    print("bit-exact vs oracle:", np.array_equal(bits(img), bits(ref)))

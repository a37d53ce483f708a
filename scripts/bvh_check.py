"""Scratch: mesh scene rendered with the triangle hierarchies and with the per-box triangle loop;
prints both kernel times and whether the two images are bit-identical."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, scene_main_cpp_with_mesh, bits
W = int(sys.argv[1]) if len(sys.argv) > 1 else 480
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mesh = sys.argv[3] if len(sys.argv) > 3 else "f1_car"
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 11
scene = scene_main_cpp_with_mesh(mesh)
cam = pkg.camera_main_cpp(W, SPP, 50)
bw, bh = pkg.set_blocks(cam, 4)
seeds = pkg.make_seeds(bw * bh, seed)
imgs = {}
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(seeds)
for mode in (pkg.MESH_BVH_CAPPED, pkg.MESH_BVH, pkg.MESH_LOOP):
    ctx.set_mesh_mode(mode)
    img = ctx.render_frame(cam)
    ctx.set_timing(True)
    img = ctx.render_frame(cam)
    ms, n = ctx.get_timing()
    paths = cam.width * cam.height * SPP
    print("mode=%s %s %dx%d %d spp: kernel %.2f ms (%d launches), %.2f Mpaths/s" % (mode, mesh, cam.width, cam.height, SPP, ms, n, paths / ms / 1e3), flush=True)
    imgs[mode] = img
    ctx.get_timing()
    ctx.set_timing(False)
ctx.close()
ok = True
for mode in (pkg.MESH_BVH_CAPPED, pkg.MESH_BVH):
    diff = int((bits(imgs[mode]) != bits(imgs[pkg.MESH_LOOP])).sum())
    print("mode %d vs loop: differing floats: %d of %d" % (mode, diff, imgs[mode].size))
    ok = ok and diff == 0
sys.exit(0 if ok else 1)

"""Scratch: times the divergence-stress scene (500 spheres + ~100k-triangle height field in 256 boxes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg
W = int(sys.argv[1]) if len(sys.argv) > 1 else 960
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 8
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 500
scene = pkg.scene_stress(NS, 224, 224, 16, 16)
cam = pkg.camera_stress(W, SPP, 50)
bw, bh = pkg.set_blocks(cam, 4)
seeds = pkg.make_seeds(bw * bh, 11)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(seeds)
for mode in (pkg.MESH_BVH, pkg.MESH_LOOP):
    ctx.set_mesh_mode(mode)
    ctx.set_timing(True)
    img = ctx.render_frame(cam)
    ms, n = ctx.get_timing()
    paths = cam.width * cam.height * SPP
    print("stress mode=%d %dx%d %d spp, %d spheres, %d tris, %d boxes: kernel %.1f ms, %.2f Mpaths/s" % (
        mode, cam.width, cam.height, SPP, scene.S, len(scene.triangles), len(scene.boxes), ms, paths / ms / 1e3), flush=True)
    if mode == pkg.MESH_BVH: a = img
print("identical:", np.array_equal(a.view(np.uint32), img.view(np.uint32)))

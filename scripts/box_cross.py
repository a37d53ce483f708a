"""Scratch: hierarchy over the boxes vs the loop over all boxes, by box count (the private box_tree_min tuning hook decides)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg
for nb in (2, 3, 4, 6, 8):
    scene = pkg.scene_stress(20, 224, 224, nb, nb)
    cam = pkg.camera_stress(960, 32, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    out = []
    for tree_min in (1 << 30, 2):
        ctx = pkg.Context(0); ctx.tune("box_tree_min", tree_min); ctx.upload_seeds(pkg.make_seeds(bw * bh, 11)); ctx.upload_scene(scene)
        ctx.render_frame(cam); ctx.render_frame(cam)
        ctx.set_timing(True); ctx.render_frame(cam); ms, n = ctx.get_timing()
        out.append(ms); ctx.close()
    print("%3d boxes: loop %.2f ms, hierarchy %.2f ms" % (nb * nb, out[0], out[1]), flush=True)

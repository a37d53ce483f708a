"""Scratch: the stress scene (1920x1080 x 16 spp, proven walk) under the hand-over threshold of the cooperative walk."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg
scene = pkg.scene_stress(500, 224, 224, 16, 16); cam = pkg.camera_stress(1920, 16, 50)
bw, bh = pkg.set_blocks(cam, 4)
for idle in (2, 4, 8, 16, 24):
    ctx = pkg.Context(0); ctx.tune("bvh_min_idle", idle); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 11))
    ctx.render_frame(cam); ctx.render_frame(cam)
    ctx.set_timing(True)
    for _ in range(3): ctx.render_frame(cam)
    ms, n = ctx.get_timing()
    print("min idle %2d: %.2f ms per frame (schedule carried over)" % (idle, ms / n), flush=True)
    ctx.close()

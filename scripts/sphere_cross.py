"""Scratch: spheres-only stress scenes of growing size, sphere hierarchy on (default threshold) vs off."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg
cam = pkg.camera_stress(960, 8, 50)
bw, bh = pkg.set_blocks(cam, 4)
seeds = pkg.make_seeds(bw * bh, 11)
for n in (16, 32, 64, 128, 256, 500, 2000):
    full = pkg.scene_stress(n, 8, 8, 1, 1)
    out = []
    for tmin in (1, 1000000):
        ctx = pkg.Context(0); ctx.tune("sphere_tree_min", tmin); ctx.upload_seeds(seeds); ctx.upload_scene(pkg.Scene(spheres=full.spheres))
        ctx.render_frame(cam); ctx.set_timing(True); ctx.render_frame(cam); ms, _ = ctx.get_timing(); ctx.close()
        out.append(ms)
    print("%5d spheres: hierarchy %.2f ms, loop %.2f ms" % (n, out[0], out[1]), flush=True)

"""Scratch: headline frame in the strict and the tolerant arithmetic mode: kernel time (schedule carried over and cold) and image error."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, image_error
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50)
bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
imgs = {}
for mode in (pkg.ARITH_STRICT, pkg.ARITH_TOLERANT):
    ctx.set_arith_mode(mode)
    for reuse in (True, False):
        ctx.set_schedule_reuse(reuse)
        for _ in range(3): imgs[mode] = ctx.render_frame(cam)
        ctx.set_timing(True)
        for _ in range(10): ctx.render_frame(cam)
        ms, n = ctx.get_timing(); ctx.set_timing(False)
        print("arith %d, schedule %s: %.3f ms per frame" % (mode, "carried over" if reuse else "cold", ms / n), flush=True)
print("tolerant vs strict: MAE %.3e PSNR %.1f dB" % image_error(imgs[1], imgs[0]))
print("fraction of channel values that differ: %.4f" % float((imgs[0] != imgs[1]).mean()))
ctx.close()

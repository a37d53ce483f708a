"""Scratch: tolerant vs strict image error on parts of the stress scene."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, image_error
full = pkg.scene_stress(60, 48, 48, 6, 6)
cases = {"full": full, "spheres": pkg.Scene(spheres=full.spheres), "mesh": pkg.Scene(triangles=full.triangles, boxes=full.boxes),
         "40 spheres": pkg.Scene(spheres=pkg.scene_stress(40, 8, 8, 1, 1).spheres)}
ctx = pkg.Context(0)
for name, sc in cases.items():
    for spp in (64, 250):
        cam = pkg.camera_stress(192, spp, 50)
        bw, bh = pkg.set_blocks(cam, 4)
        ctx.upload_scene(sc); ctx.upload_seeds(pkg.make_seeds(bw * bh, 34))
        ctx.set_arith_mode(pkg.ARITH_STRICT); a = ctx.render_frame(cam)
        ctx.set_arith_mode(pkg.ARITH_TOLERANT); b = ctx.render_frame(cam)
        mae, psnr = image_error(b, a)
        print("%-12s %3d spp: MAE %.3e PSNR %.1f, pixels differing %.3f" % (name, spp, mae, psnr, float((a != b).any(axis=2).mean())), flush=True)
ctx.close()

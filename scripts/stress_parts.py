"""Scratch: stress scene split into its parts (spheres only / few spheres + mesh / all)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg
W, SPP = 960, 8
full = pkg.scene_stress(500, 224, 224, 16, 16)
cam = pkg.camera_stress(W, SPP, 50)
bw, bh = pkg.set_blocks(cam, 4)
seeds = pkg.make_seeds(bw * bh, 11)
ctx = pkg.Context(0); ctx.upload_seeds(seeds)
few = (pkg.Sphere * 4)(*[full.spheres[i] for i in range(4)])
cases = {"all": full, "spheres only": pkg.Scene(spheres=full.spheres),
         "4 spheres + mesh": pkg.Scene(spheres=few, triangles=full.triangles, boxes=full.boxes)}
for name, sc in cases.items():
    ctx.upload_scene(sc)
    ctx.render_frame(cam)
    ctx.set_timing(True); ctx.render_frame(cam); ms, n = ctx.get_timing(); ctx.set_timing(False)
    c = ctx.count_events(cam)
    paths = cam.width * cam.height * SPP
    print("%-18s kernel %.1f ms, %.1f Mpaths/s, segments/path %.2f, sphere tests/seg %.0f, bbox tests/seg %.0f, bbox hits/seg %.2f, tri tests/seg %.1f" % (
        name, ms, paths / ms / 1e3, c["segments"] / paths, c["sphere_tests"] / max(c["segments"], 1), c["bbox_tests"] / max(c["segments"], 1),
        c["bbox_hits"] / max(c["segments"], 1), c["tri_tests"] / max(c["segments"], 1)), flush=True)

"""Scratch: headline frame with the per-wave render kernel and with the workgroup kernel that re-deals pixel streams: equality and time."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg
scene = pkg.scene_readme()
small = pkg.camera_readme(160, 7, 6); sbw, sbh = pkg.set_blocks(small, 3)
cam = pkg.camera_readme(1280, 250, 50); bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.upload_scene(scene)
variants = [(0, 0), (8, 2), (8, 3), (4, 4), (4, 6), (4, 5)]
for mode in (pkg.ARITH_STRICT, pkg.ARITH_TOLERANT):
    ctx.set_arith_mode(mode)
    ctx.upload_seeds(pkg.make_seeds(sbw * sbh, 7))
    ref = None
    for wg, per_cu in variants:
        ctx.tune("wg_kernel", wg); ctx.tune("wg_per_cu", per_cu)
        img = ctx.render_frame(small)
        if ref is None: ref = img
        if not np.array_equal(img, ref, equal_nan=True): print("SMALL FRAME DIFFERS", mode, wg, per_cu, flush=True)
    ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
    ref = None
    for wg, per_cu in variants:
        ctx.tune("wg_kernel", wg); ctx.tune("wg_per_cu", per_cu)
        for reuse in (True, False):
            ctx.set_schedule_reuse(reuse)
            for _ in range(3): img = ctx.render_frame(cam)
            if ref is None: ref = img
            ctx.set_timing(True)
            for _ in range(10): ctx.render_frame(cam)
            ms, n = ctx.get_timing(); ctx.set_timing(False)
            print("arith %d waves %d per_cu %d, schedule %s: %.3f ms per frame, equal %s" % (mode, wg, per_cu, "carried over" if reuse else "cold", ms / n, np.array_equal(img, ref, equal_nan=True)), flush=True)
ctx.close()

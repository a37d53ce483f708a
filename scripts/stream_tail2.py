"""Scratch: can a pilot (4 spp, 8 bounces) find the pixels with the longest streams of the full launch?"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.scene_readme()
lib = pkg.lib(); lib.wbrt_pixel_costs_.restype = C.c_int
def costs_of(spp, depth):
    cam = pkg.camera_readme(1280, spp, depth)
    bw, bh = pkg.set_blocks(cam, 4)
    ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
    ctx.render_frame(cam)
    c = np.zeros((cam.height, cam.width), dtype=np.int32)
    lib.wbrt_pixel_costs_(ctx._h, C.byref(cam), c.ctypes.data_as(C.c_void_p))
    ctx.close()
    return c
full = costs_of(250, 50).astype(np.float64)
order_full = np.argsort(full.ravel())[::-1]
for spp, depth in ((4, 8), (8, 8), (16, 8), (4, 50)):
    pil = costs_of(spp, depth).astype(np.float64)
    # smoothed over 3x3 and 7x7 pixels
    def smooth(a, r):
        out = np.zeros_like(a); n = 0
        for dy in range(-r, r + 1):
            for dx in range(-r, r + 1):
                out += np.roll(np.roll(a, dy, 0), dx, 1); n += 1
        return out / n
    for name, est in (("raw", pil), ("3x3", smooth(pil, 1)), ("7x7", smooth(pil, 3))):
        rank = np.empty(est.size, dtype=np.int64)
        rank[np.argsort(est.ravel())[::-1]] = np.arange(est.size)
        line = []
        for top in (4, 16, 64):
            worst = rank[order_full[:top]].max()
            line.append("true top %d within the pilot's top %d" % (top, worst + 1))
        print("pilot %d spp depth %d, %s: %s" % (spp, depth, name, "; ".join(line)), flush=True)

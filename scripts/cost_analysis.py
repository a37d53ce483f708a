"""Scratch: per-pixel iteration counts of the headline frame -> wave-level lane efficiency."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50)
bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
costs = np.zeros((cam.height, cam.width), dtype=np.int32)
lib = pkg.lib(); lib.wbrt_pixel_costs_.restype = C.c_int
rc = lib.wbrt_pixel_costs_(ctx._h, C.byref(cam), costs.ctypes.data_as(C.c_void_p)); assert rc == 0
np.save("gpurun_out/costs_readme_1280.npy", costs)
print("iterations per pixel: min %d mean %.1f max %d" % (costs.min(), costs.mean(), costs.max()))
# 8x8 tiles as the kernel forms waves (inside each 640x360 block)
t = costs.reshape(cam.height // 8, 8, cam.width // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
wmax = t.max(axis=1)
print("tile waves: lane efficiency (sum / 64*max) = %.3f" % (t.sum() / (64.0 * wmax.sum())))
s = np.sort(costs.reshape(-1))[::-1].reshape(-1, 64)
print("cost-sorted waves: lane efficiency = %.3f" % (s.sum() / (64.0 * s.max(axis=1).sum())))
print("wave iterations: tile mean %.1f, sorted mean %.1f" % (wmax.mean(), s.max(axis=1).mean()))
pct = np.percentile(costs, [1, 10, 50, 90, 99, 99.9])
print("percentiles 1/10/50/90/99/99.9:", pct)

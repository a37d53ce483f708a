"""Scratch: latency of ONE wave: the README scene at 8x8 pixels (a single tile), many samples."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.scene_readme()
spp = 4000
cam = pkg.calculate_camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 4.0, 1.0, 10.0, 0.0, 8, 8, spp, 50)
cam.blockSize = pkg.vec3(8, 8, 0)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(64, 5))
ctx.render_frame(cam)
ctx.set_timing(True); ctx.render_frame(cam); ms, n = ctx.get_timing()
lib = pkg.lib(); lib.wbrt_pixel_costs_.restype = C.c_int
costs = np.zeros((8, 8), dtype=np.int32)
lib.wbrt_pixel_costs_(ctx._h, C.byref(cam), costs.ctypes.data_as(C.c_void_p))
print("lone wave: %.3f ms, wave-max iterations %d, mean %d -> %.3f us per iteration" % (ms, costs.max(), costs.mean(), ms * 1e3 / costs.max()))

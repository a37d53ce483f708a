"""Scratch: per-iteration latency of ONE wave alone on the GPU by the number of its live lanes (README scene, narrow view of
the spheres, 8x8 / 4x4 / 2x2 / 1x1 pixels, many samples)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.scene_readme()
spp = 4000
lib = pkg.lib(); lib.wbrt_pixel_costs_.restype = C.c_int
for n in (8, 4, 2, 1):
    cam = pkg.calculate_camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 4.0 * n / 8.0, 1.0, 10.0, 0.0, n, n, spp, 50)
    cam.blockSize = pkg.vec3(n, n, 0)
    ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(n * n, 5))
    ctx.tune("lpt", 0)
    ctx.render_frame(cam)
    ctx.set_timing(True); ctx.render_frame(cam); ms, _ = ctx.get_timing()
    costs = np.zeros((n, n), dtype=np.int32)
    lib.wbrt_pixel_costs_(ctx._h, C.byref(cam), costs.ctypes.data_as(C.c_void_p))
    print("%2d live lanes: %.3f ms, longest stream %d iterations (mean %d) -> %.3f us per iteration of the wave" % (
        n * n, ms, costs.max(), costs.mean(), ms * 1e3 / costs.max()), flush=True)
    ctx.close()

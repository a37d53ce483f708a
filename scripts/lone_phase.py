"""Scratch (needs variants/phase.so): shader-clock cycles per phase of the main loop, one lone wave."""
import os, sys, ctypes as C
import numpy as np
os.environ["WBRT_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opencl-ray-tracer_amd", "variants", "phase.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.scene_readme()
cam = pkg.calculate_camera((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 4.0, 1.0, 10.0, 0.0, 8, 8, 4000, 50)
cam.blockSize = pkg.vec3(8, 8, 0)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(64, 5))
ctx.set_timing(True); ctx.render_frame(cam); ms, n = ctx.get_timing()
buf = np.zeros(8, dtype=np.uint64)
pkg.lib().wbrt_dbg_dump_(buf.ctypes.data_as(C.c_void_p), 2)
it = float(buf[6]); names = ["depth check + spheres", "surface lookup", "rejection loop", "scatter / materials", "sky", "accumulate + next camera ray"]
print("kernel %.2f ms, iterations (lane 0) %d" % (ms, it))
tot = float(buf[:6].sum())
for k in range(6):
    print("  %-30s %8.0f cycles/iteration  %5.1f%%" % (names[k], buf[k] / it, 100.0 * buf[k] / tot))
print("  total %.0f cycles/iteration; %.3f us/iteration wall -> clock %.2f GHz-equivalent" % (tot / it, ms * 1e3 / it, tot / it / (ms * 1e3 / it) / 1e3))

"""Scratch (needs the instrumented variants/clk.so): per-wave start/end clocks of the headline frame."""
import os, sys, ctypes as C
import numpy as np
os.environ["WBRT_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opencl-ray-tracer_amd", "variants", "clk.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50)
bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
ctx.render_frame(cam); ctx.render_frame(cam)
n = 14400
buf = np.zeros((n, 4), dtype=np.uint64)
pkg.lib().wbrt_dbg_dump_(buf.ctypes.data_as(C.c_void_p), n)
np.save("gpurun_out/wave_times.npy", buf)
t0 = buf[:, 0].min(); start = (buf[:, 0] - t0).astype(np.float64); end = (buf[:, 1] - t0).astype(np.float64); it = buf[:, 2].astype(np.float64)
T = end.max()
print("frame span (clock ticks) %.0f; waves %d" % (T, n))
for q in (0.5, 0.7, 0.8, 0.9, 0.95, 0.99):
    print("  %.0f%% of waves finished by %.3f of the span" % (q * 100, np.quantile(end, q) / T))
late = end > 0.85 * T
print("waves ending in the last 15%%: %d; their mean wave-max iterations %.0f (all waves %.0f); their mean start %.3f, mean duration %.3f of span" % (
    late.sum(), it[late].mean(), it.mean(), (start[late] / T).mean(), ((end - start)[late] / T).mean()))
dur = end - start
rate = it / np.maximum(dur, 1)
print("iterations per tick: early long waves (pos<288) %.5f, all %.5f, late starters (start>0.5T) %.5f, last 20 to finish %.5f" % (
    rate[:288].mean(), rate.mean(), rate[start > 0.5 * T].mean(), rate[np.argsort(-end)[:20]].mean()))
lastk = np.argsort(-end)[:12]
for i in lastk:
    print("  last to finish: pos %5d start %.3f end %.3f wave-max iters %d" % (i, start[i] / T, end[i] / T, it[i]))
pos = np.arange(n)
for lo, hi in ((0, 288), (288, 1440), (1440, 4800), (4800, 9600), (9600, 14400)):
    sel = (pos >= lo) & (pos < hi)
    print("  launch positions %5d-%5d: mean wave-max iters %.0f, max %d, mean start %.3f, mean end %.3f" % (lo, hi, it[sel].mean(), it[sel].max(), (start[sel] / T).mean(), (end[sel] / T).mean()))
order = np.argsort(-(end - start))[:10]
for i in order:
    print("  longest waves: pos %5d start %.3f end %.3f wave-max iters %d" % (i, start[i] / T, end[i] / T, it[i]))
# concurrency over time
ts = np.linspace(0, T, 21)
print("running waves at t:", [(int(((start <= t) & (end > t)).sum())) for t in ts])

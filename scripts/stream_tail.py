"""Scratch: how long the longest pixel streams of the headline frame are (loop iterations per pixel at 250 spp), and how many
tiles (waves) hold one of the K longest."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50)
bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
ctx.render_frame(cam)
lib = pkg.lib(); lib.wbrt_pixel_costs_.restype = C.c_int
costs = np.zeros((cam.height, cam.width), dtype=np.int32)
rc = lib.wbrt_pixel_costs_(ctx._h, C.byref(cam), costs.ctypes.data_as(C.c_void_p))
flat = np.sort(costs.ravel())[::-1]
print("pixels %d, mean %.0f iterations, longest %d" % (flat.size, flat.mean(), flat[0]))
for k in (1, 4, 16, 64, 256, 1024, 4096, 16384):
    print("  %5d-th longest stream: %d iterations" % (k, flat[k - 1]))
# per 8x8 tile: the wave lasts as long as its longest lane
H, W = costs.shape
tiles = costs[:H - H % 8, :W - W % 8].reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3)).ravel()
ts = np.sort(tiles)[::-1]
for k in (1, 4, 16, 64, 256, 1024):
    print("  %5d-th longest tile (its longest lane): %d iterations" % (k, ts[k - 1]))

# the tile of the longest stream: its 64 lanes, sorted
costs = costs[:H - H % 8, :W - W % 8]
y, x = np.unravel_index(np.argmax(costs), costs.shape)
tile = costs[y // 8 * 8:y // 8 * 8 + 8, x // 8 * 8:x // 8 * 8 + 8]
print("tile of the longest stream (pixel %d,%d): lanes sorted:" % (x, y), np.sort(tile.ravel())[::-1].tolist())
for k in (1, 2, 3):
    yy, xx = np.unravel_index(np.argsort(costs.ravel())[::-1][k * 5], costs.shape)
    t2 = costs[yy // 8 * 8:yy // 8 * 8 + 8, xx // 8 * 8:xx // 8 * 8 + 8]
    print("tile of the %d-th longest stream: top lanes" % (k * 5 + 1), np.sort(t2.ravel())[::-1][:12].tolist(), "median", int(np.median(t2)))

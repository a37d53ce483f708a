"""Scratch: per-pixel iteration counts of the headline frame at 4 spp (what the pilot launch sees) and 8 spp."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
scene = pkg.scene_readme()
lib = pkg.lib(); lib.wbrt_pixel_costs_.restype = C.c_int
for spp in (4, 8, 250):
    cam = pkg.camera_readme(1280, spp, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
    costs = np.zeros((cam.height, cam.width), dtype=np.int32)
    assert lib.wbrt_pixel_costs_(ctx._h, C.byref(cam), costs.ctypes.data_as(C.c_void_p)) == 0
    np.save("gpurun_out/costs_readme_1280_spp%d.npy" % spp, costs)
    print(spp, costs.mean())
    ctx.close()

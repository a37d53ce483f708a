"""Scratch: a few headline frames with one setting of the workgroup kernel (run under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg
wg = int(sys.argv[1])
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50); bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
ctx.tune("wg_kernel", wg)
for _ in range(3): ctx.render_frame(cam)
ctx.close()

"""Scratch: the stress scene (500 spheres + 100352 triangles in 256 boxes) at 1920x1080 x 16 spp in the mesh modes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, bits
scene = pkg.scene_stress(500, 224, 224, 16, 16)
cam = pkg.camera_stress(1920, 16, 50)
bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 11))
imgs = {}
for mode, name in ((pkg.MESH_BVH, "proven"), (pkg.MESH_BVH_CAPPED, "capped")):
    ctx.set_mesh_mode(mode)
    ctx.render_frame(cam); ctx.render_frame(cam)
    ctx.set_timing(True); imgs[mode] = ctx.render_frame(cam); ms, n = ctx.get_timing(); ctx.set_timing(False)
    print("%s: %.1f ms" % (name, ms), flush=True)
print("identical:", bool((bits(imgs[pkg.MESH_BVH]) == bits(imgs[pkg.MESH_BVH_CAPPED])).all()))
ctx.close()

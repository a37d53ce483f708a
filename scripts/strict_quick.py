"""Scratch: the mesh scene in the capped and the proven-margin walk (loop threshold given), kernel times at a sample count."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, scene_main_cpp_with_mesh, bits
W = int(sys.argv[1]); SPP = int(sys.argv[2]); mesh = sys.argv[3]; loop_max = int(sys.argv[4]) if len(sys.argv) > 4 else 256
scene = scene_main_cpp_with_mesh(mesh)
cam = pkg.camera_main_cpp(W, SPP, 50)
bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.tune("strict_loop_max", loop_max); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 11))
imgs = {}
for mode in (pkg.MESH_BVH_CAPPED, pkg.MESH_BVH):
    ctx.set_mesh_mode(mode)
    ctx.render_frame(cam)
    ctx.render_frame(cam)
    ctx.set_timing(True)
    for _ in range(3):
        imgs[mode] = ctx.render_frame(cam)
    ms, n = ctx.get_timing()
    ctx.set_timing(False)
    print("mode=%d %s %dx%d %d spp loop_max=%d: kernel %.2f ms per frame" % (mode, mesh, cam.width, cam.height, SPP, loop_max, ms / n), flush=True)
print("identical:", (bits(imgs[pkg.MESH_BVH_CAPPED]) == bits(imgs[pkg.MESH_BVH])).all())
ctx.close()

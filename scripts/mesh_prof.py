"""Scratch: a few frames of the f1_car scene in one mesh mode (run under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, scene_main_cpp_with_mesh
mode = int(sys.argv[1]); spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
scene = scene_main_cpp_with_mesh("f1_car"); cam = pkg.camera_main_cpp(1920, spp, 50); bw, bh = pkg.set_blocks(cam, 4)
ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 11)); ctx.set_mesh_mode(mode)
for _ in range(3): ctx.render_frame(cam)
ctx.close()

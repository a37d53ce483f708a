"""Scratch: proven-margin walk against the cooperative loop by mesh size (sub-meshes of the burger and f1_car soups)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import pkg, golden, bits
cam = pkg.camera_main_cpp(1920, 8, 50); bw, bh = pkg.set_blocks(cam, 4)
for mesh, sizes in (("burger", (32, 64, 128, 256, 512, 828)), ("f1_car", (64, 256, 1024, 4096))):
    soup = golden("meshes.npz")[mesh]
    for n in sizes:
        # spatially compact sub-mesh: the n triangles nearest to the first one
        c = soup.reshape(-1, 3, 3).mean(axis=1)
        idx = np.sort(np.argsort(((c - c[0]) ** 2).sum(axis=1))[:n])
        scene = pkg.scene_main_cpp_from_soup(soup[idx])
        ctx = pkg.Context(0); ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 11))
        res = {}
        for name, loop_max in (("loop", 1 << 30), ("walk", 0)):
            ctx.tune("strict_loop_max", loop_max)
            ctx.render_frame(cam); ctx.render_frame(cam)
            ctx.set_timing(True)
            for _ in range(3): img = ctx.render_frame(cam)
            ms, k = ctx.get_timing(); ctx.set_timing(False)
            res[name] = (ms / k, img)
        print("%s %5d triangles: loop %.2f ms, walk %.2f ms, identical %s" % (mesh, n, res["loop"][0], res["walk"][0], bool((bits(res["loop"][1]) == bits(res["walk"][1])).all())), flush=True)
        ctx.close()

"""Scratch: cold-frame time of the headline workload by pilot sample count and pilot bounce limit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50)
bw, bh = pkg.set_blocks(cam, 4)
buf = torch.zeros(cam.height * cam.width * 3, dtype=torch.float32, device="cuda:0")
for spp, depth in ((4, 0), (4, 16), (4, 8), (4, 5), (2, 8), (8, 8), (8, 5), (16, 5)):
    ctx = pkg.Context(0); ctx.tune("pilot_spp", spp); ctx.tune("pilot_depth", depth)
    ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
    ctx.set_schedule_reuse(False)
    for _ in range(3): ctx.render_frame_device(cam, buf.data_ptr())
    ctx.synchronize(); ctx.set_timing(True)
    for _ in range(12): ctx.render_frame_device(cam, buf.data_ptr())
    ctx.synchronize(); ms, n = ctx.get_timing()
    print("pilot %2d spp, depth limit %2d: cold frame %.3f ms" % (spp, depth, ms / n), flush=True)
    ctx.close()

"""Scratch: one rank's share of the headline frame at 4 and 8 ranks (V = 8 shards) by pilot size: is a 4-sample pilot still
right for launches of 31-63 samples?  Cold frames, no exchange step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util import pkg
import importlib
mg = importlib.import_module(pkg.__name__ + ".multi_gpu")
V = 8
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50); bw, bh = pkg.set_blocks(cam, 4)
spp = pkg.split_spp(250, V); cam.samplesPerPixel = max(spp)
frame = None
for pilot_spp, pilot_depth, min_spp in ((4, 8, 1), (2, 8, 1), (1, 8, 1), (2, 5, 1), (1, 5, 1), (4, 8, 100000)):
    ctx = pkg.Context(0); ctx.upload_scene(scene)
    ctx.tune("pilot_spp", pilot_spp); ctx.tune("pilot_depth", pilot_depth); ctx.tune("lpt_min_spp", min_spp)
    if frame is None: frame = mg.FrameBuffers(cam.width, cam.height, torch.device("cuda", 0))
    stream = torch.cuda.current_stream().cuda_stream
    line = []
    for world in (2, 4, 8):
        def step():
            ctx.set_schedule_reuse(True)
            mg.render_sharded_frame(ctx, cam, frame, V, 0x5EED, 0, world, stream=stream, all_reduce=None, shard_spp=spp)
        step(); step(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        line.append("world %d: %.3f ms" % (world, (time.perf_counter() - t) * 50))
    print("pilot %d spp, depth %d, %s: %s" % (pilot_spp, pilot_depth, "unscheduled" if min_spp > 1000 else "scheduled", "; ".join(line)), flush=True)
    ctx.close()

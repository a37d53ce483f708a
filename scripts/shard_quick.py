"""Scratch: what one rank of an N-GPU run computes per headline frame (shards v % N == 0 of 8), timed on one GPU without the all-reduce."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util import pkg
import importlib
mg = importlib.import_module(pkg.__name__ + ".multi_gpu")
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50); bw, bh = pkg.set_blocks(cam, 4)
spp = pkg.split_spp(250, 8); cam.samplesPerPixel = max(spp)
ctx = pkg.Context(0); ctx.upload_scene(scene)
frame = mg.FrameBuffers(cam.width, cam.height, torch.device("cuda", 0))
stream = torch.cuda.current_stream().cuda_stream
for world in (1, 2, 4, 8):
    for cold in (True, False):
        def step():
            if cold: ctx.set_schedule_reuse(True)
            mg.render_sharded_frame(ctx, cam, frame, 8, 0x5EED, 0, world, stream=stream, all_reduce=None, shard_spp=spp)
        step(); step(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize()
        print("world %d (%d shards on this rank), %s: %.3f ms per frame" % (world, 8 // world, "cold" if cold else "schedule carried over", (time.perf_counter() - t) * 100), flush=True)
ctx.close()

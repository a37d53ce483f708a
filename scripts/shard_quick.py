"""Scratch: what one rank of an N-GPU run computes per headline frame (shards v % N == 0 of V), timed on one GPU without the
exchange step.  usage: shard_quick.py [V=8] [lpt_min_spp=1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from util import pkg
import importlib
mg = importlib.import_module(pkg.__name__ + ".multi_gpu")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
min_spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50); bw, bh = pkg.set_blocks(cam, 4)
spp = pkg.split_spp(250, V); cam.samplesPerPixel = max(spp)
ctx = pkg.Context(0); ctx.upload_scene(scene)
ctx.tune("lpt_min_spp", min_spp)
frame = mg.FrameBuffers(cam.width, cam.height, torch.device("cuda", 0))
stream = torch.cuda.current_stream().cuda_stream
for world in (1, 2, 4, 8):
    for cold in (True, False):
        def step():
            if cold: ctx.set_schedule_reuse(True)
            mg.render_sharded_frame(ctx, cam, frame, V, 0x5EED, 0, world, stream=stream, all_reduce=None, shard_spp=spp)
        step(); step(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize()
        print("V=%d lpt_min_spp=%d world %d (%d shards on this rank), %s: %.3f ms per frame" % (V, min_spp, world, len(pkg.shards_of_rank(V, 0, world)), "cold" if cold else "schedule carried over", (time.perf_counter() - t) * 100), flush=True)
ctx.close()

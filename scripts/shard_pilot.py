"""Scratch: one 31/32-spp seed shard of the headline frame (what a rank of 8 renders), cold, for pilot lengths and depths."""
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
def run(label):
    def step():
        ctx.set_schedule_reuse(True)
        mg.render_sharded_frame(ctx, cam, frame, 8, 0x5EED, 0, 8, stream=stream, all_reduce=None, shard_spp=spp)
    step(); step(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print("%s: %.3f ms" % (label, (time.perf_counter() - t) * 50), flush=True)
ctx.tune("lpt", 0); run("no schedule"); ctx.tune("lpt", 1)
for ps in (1, 2, 3, 4):
    for pd in (2, 3, 5, 8):
        ctx.tune("pilot_spp", ps); ctx.tune("pilot_depth", pd)
        run("pilot %d spp, depth %d" % (ps, pd))
ctx.close()

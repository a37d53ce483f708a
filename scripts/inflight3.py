"""Scratch: stream creation recipes for two frames in flight, each in a fresh process (argv[1])."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
mg = __import__("importlib").import_module(g.PKG_NAME + ".multi_gpu")
recipe = sys.argv[1]
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50)
bw, bh = pkg.set_blocks(cam, 4)
seeds = pkg.make_seeds(bw * bh, 0x5EED)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if recipe == "streams_first":
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
elif recipe == "four_02":
    s = [torch.cuda.Stream() for _ in range(4)]; streams = [s[0], s[2]]
elif recipe == "four_03":
    s = [torch.cuda.Stream() for _ in range(4)]; streams = [s[0], s[3]]
elif recipe == "eight_04":
    s = [torch.cuda.Stream() for _ in range(8)]; streams = [s[0], s[4]]
elif recipe == "prio":
    streams = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]
ctxs, frames = [], []
for i in range(2):
    c = pkg.Context(0); c.upload_scene(scene); c.upload_seeds(seeds)
    ctxs.append(c); frames.append(mg.FrameBuffers(cam.width, cam.height, dev))
if recipe == "after":
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
if recipe == "ctx_streams":
    streams = None
def step(i):
    k = i % 2
    mg.render_step(ctxs[k], cam, frames[k], total_spp=250, stream=streams[k].cuda_stream)
for i in range(4): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 20
for i in range(K): step(i)
torch.cuda.synchronize()
print("%s: %.3f ms per frame" % (recipe, (time.perf_counter() - t0) / K * 1e3), flush=True)

"""Scratch: throughput of the headline frame with 1 vs 2 frames in flight (two contexts, two streams)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
mg = __import__("importlib").import_module(g.PKG_NAME + ".multi_gpu")
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50)
bw, bh = pkg.set_blocks(cam, 4)
seeds = pkg.make_seeds(bw * bh, 0x5EED)
dev = torch.device("cuda", 0)
for nfl in (1, 2, 3):
    ctxs, frames, streams = [], [], []
    for i in range(nfl):
        c = pkg.Context(0); c.upload_scene(scene); c.upload_seeds(seeds)
        ctxs.append(c); frames.append(mg.FrameBuffers(cam.width, cam.height, dev)); streams.append(torch.cuda.Stream())
    def step(i):
        k = i % nfl
        mg.render_step(ctxs[k], cam, frames[k], total_spp=250, stream=streams[k].cuda_stream)
    for i in range(6): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 30
    for i in range(K): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d in flight: %.3f ms per frame, %.1f Mpaths/s" % (nfl, dt / K * 1e3, cam.width * cam.height * 250 * K / dt / 1e6), flush=True)
    for c in ctxs: c.close()

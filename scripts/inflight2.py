"""Scratch: why does two-in-flight differ between scripts? toggles: with-stream context, set_device."""
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
def run(tag, use_with, warm, K):
    nfl = 2
    ctxs, frames, streams = [], [], []
    for i in range(nfl):
        c = pkg.Context(0); c.upload_scene(scene); c.upload_seeds(seeds)
        ctxs.append(c); frames.append(mg.FrameBuffers(cam.width, cam.height, dev)); streams.append(torch.cuda.Stream())
    def step(i):
        k = i % nfl
        if use_with:
            with torch.cuda.stream(streams[k]):
                mg.render_step(ctxs[k], cam, frames[k], total_spp=250, stream=streams[k].cuda_stream)
        else:
            mg.render_step(ctxs[k], cam, frames[k], total_spp=250, stream=streams[k].cuda_stream)
    for i in range(warm): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s: %.3f ms per frame" % (tag, dt / K * 1e3), flush=True)
    for c in ctxs: c.close()
run("plain warm6 K30", False, 6, 30)
run("with   warm6 K30", True, 6, 30)
run("plain warm3 K20", False, 3, 20)
run("plain warm4 K20", False, 4, 20)

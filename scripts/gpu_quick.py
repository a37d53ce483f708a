"""Scratch GPU check: parity on small scenes, then timing of the headline workload."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

pkg = g.load_package(); orc = g.load_oracle()
ctx = pkg.Context(0)
print(ctx.device_info(), flush=True)

rs = np.random.RandomState(1)
a = (rs.standard_normal(1 << 16) * 10 ** rs.uniform(-6, 6, 1 << 16)).astype(np.float32)
b = (rs.standard_normal(1 << 16) * 10 ** rs.uniform(-6, 6, 1 << 16)).astype(np.float32)
print("div exact:", np.array_equal(ctx.math_probe(0, a, b).view(np.uint32), (a / b).view(np.uint32)))
print("sqrt exact:", np.array_equal(ctx.math_probe(1, np.abs(a)).view(np.uint32), np.sqrt(np.abs(a)).view(np.uint32)))
ol = orc.oracle_lib()
at = np.array([ol.wbo_atan2f_portable(float(x), float(y)) for x, y in zip(a[:4096], b[:4096])], dtype=np.float32)
print("atan2 exact:", np.array_equal(ctx.math_probe(2, a[:4096], b[:4096]).view(np.uint32), at.view(np.uint32)))
u = rs.uniform(-1, 1, 4096).astype(np.float32)
asn = np.array([ol.wbo_asinf_portable(float(x)) for x in u], dtype=np.float32)
print("asin exact:", np.array_equal(ctx.math_probe(3, u).view(np.uint32), asn.view(np.uint32)))

def check(name, scene, cam, nblocks, seed):
    bw, bh = pkg.set_blocks(cam, nblocks)
    seeds = pkg.make_seeds(bw * bh, seed)
    ctx.upload_scene(scene); ctx.upload_seeds(seeds)
    t = time.time(); img = ctx.render_frame(cam); tg = time.time() - t
    ids = ctx.primary_ids(cam)
    O = orc.Oracle(scene)
    t = time.time(); ref = O.render_frame(cam, seeds); tc = time.time() - t
    rid = O.primary_ids(cam, seeds)
    nd = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
    print("%s: image ndiff %d (max %.3g), ids ndiff %d, gpu %.3fs cpu %.3fs" % (
        name, nd, float(np.abs(img - ref).max()), int((ids != rid).sum()), tg, tc), flush=True)
    gc = ctx.count_events(cam)
    _, oc = O.render_frame(cam, seeds, counters=True)
    oc = dict(zip(pkg.COUNTER_NAMES, map(int, oc)))
    bad = {k: (gc[k], oc[k]) for k in gc if gc[k] != oc[k]}
    print("   counters equal:", not bad, bad if bad else "")

check("readme 320x180x16 4blk", pkg.scene_readme(), pkg.camera_readme(320, 16, 50), 4, 7)
check("readme 320x180x16 1blk", pkg.scene_readme(), pkg.camera_readme(320, 16, 50), 1, 7)
check("maincpp spheres 320x180x16", pkg.scene_main_cpp(), pkg.camera_main_cpp(320, 16, 50), 4, 5)
check("stress small 128x72x4", pkg.scene_stress(40, 24, 24, 4, 4), pkg.camera_stress(128, 4, 50), 4, 3)

# headline workload
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50)
bw, bh = pkg.set_blocks(cam, 4)
ctx.upload_scene(scene); ctx.upload_seeds(pkg.make_seeds(bw * bh, 0x5EED))
import torch
buf = torch.zeros(cam.height * cam.width * 3, dtype=torch.float32, device="cuda:0")
ctx.set_timing(True)
for i in range(3):
    ctx.render_frame_device(cam, buf.data_ptr())
ctx.synchronize(); ms, n = ctx.get_timing()
print("warm: %.3f ms/frame" % (ms / n))
for i in range(5):
    ctx.render_frame_device(cam, buf.data_ptr())
ctx.synchronize(); ms, n = ctx.get_timing()
paths = cam.width * cam.height * cam.samplesPerPixel
print("C1 1280x720x250: %.3f ms/frame, %.1f Mpaths/s" % (ms / n, paths / (ms / n) / 1e3))
c = ctx.count_events(cam)
fl = pkg.algorithmic_flops(c)
print("flops/frame %.4e, flop/path %.1f -> %.2f TFLOP/s (%.1f%% of 157.3)" % (fl, fl / paths, fl / (ms / n) / 1e9, fl / (ms / n) / 1e9 / 157.3 * 100))

"""Scratch: times the headline frame under tuning knobs given on the command line (env)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
scene = pkg.scene_readme(); cam = pkg.camera_readme(1280, 250, 50)
bw, bh = pkg.set_blocks(cam, 4)
seeds = pkg.make_seeds(bw * bh, 0x5EED)
buf = torch.zeros(cam.height * cam.width * 3, dtype=torch.float32, device="cuda:0")
def run(label, **env):
    for k, v in env.items():
        os.environ[k] = str(v)
    ctx = pkg.Context(0)
    ctx.upload_scene(scene); ctx.upload_seeds(seeds)
    for _ in range(3): ctx.render_frame_device(cam, buf.data_ptr())
    ctx.synchronize(); ctx.set_timing(True)
    for _ in range(12): ctx.render_frame_device(cam, buf.data_ptr())
    ctx.synchronize(); ms, n = ctx.get_timing()
    print("%-40s %.3f ms" % (label, ms / n), flush=True)
    ctx.close()
    for k in env: os.environ.pop(k, None)
import subprocess
KNOBS = ["WBRT_LPT", "WBRT_PILOT_SPP", "WBRT_PERSISTENT", "WBRT_BLOCKS_PER_CU"]
if len(sys.argv) > 1 and sys.argv[1] == "one":
    run(os.environ.get("WBRT_LIB", "default").split("/")[-1] + " " + " ".join("%s=%s" % (k[5:], os.environ[k]) for k in KNOBS if k in os.environ))
else:
    here = os.path.dirname(os.path.abspath(__file__))
    vdir = os.path.join(os.path.dirname(here), "opencl-ray-tracer_amd", "variants")
    combos = [dict(WBRT_LPT="1")]
    for lib in sorted(os.listdir(vdir)):
        for combo in combos:
            env = dict(os.environ, WBRT_LIB=os.path.join(vdir, lib), **combo)
            subprocess.run([sys.executable, __file__, "one"], env=env, check=False)

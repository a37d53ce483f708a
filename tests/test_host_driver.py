"""The headless C++ host driver (opencl-ray-tracer_amd/host/main.cpp, mirror of the
reference's main()): builds, fails loudly without a GPU, and on a GPU writes the same
picture as the Python-side calls of the C ABI."""
import os
import subprocess

import numpy as np
import pytest

from util import pkg, ROOT, bits

EXE = os.path.join(ROOT, "opencl-ray-tracer_amd", "wbrt_render")


def test_driver_is_built_and_prints_usage():
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    r = subprocess.run([EXE, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--scene" in r.stdout


def test_driver_reports_missing_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([EXE, "--scene", "readme", "--width", "64", "--out", str(tmp_path / "x")], capture_output=True, text=True)
    assert r.returncode != 0 and "wbrt_create" in r.stderr


def _read_pfm(path):
    data = open(path, "rb").read()
    head, rest = data.split(b"\n", 3)[:3], data.split(b"\n", 3)[3]
    w, h = map(int, head[1].split())
    return np.frombuffer(rest, dtype=np.float32).reshape(h, w, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["blocks", "whole-frame"])
def test_driver_output_equals_library_calls(gpu_ctx, tmp_path, mode):
    out = str(tmp_path / "pic")
    args = [EXE, "--scene", "readme", "--width", "256", "--spp", "6", "--depth", "50", "--blocks", "4", "--seed", "77", "--out", out]
    if mode == "whole-frame":
        args.append("--whole-frame")
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    cam = pkg.camera_readme(256, 6, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    gpu_ctx.upload_scene(pkg.scene_readme())
    gpu_ctx.upload_seeds(pkg.make_seeds(bw * bh, 77))
    want = gpu_ctx.render_frame(cam)
    got = _read_pfm(out + ".pfm")
    assert np.array_equal(bits(got), bits(want))
    ppm = open(out + ".ppm", "rb").read()
    assert ppm.startswith(b"P6\n256 144\n255\n")
    assert ppm[len(b"P6\n256 144\n255\n"):] == pkg.tonemap_host(want).tobytes()
    assert os.path.getsize(out + ".bmp") == 54 + 256 * 144 * 3
    assert os.path.getsize(out + ".exr") > 256 * 144 * 12


@pytest.mark.gpu
def test_driver_sharded_frame_equals_library_shard_sum(gpu_ctx, tmp_path):
    """wbrt_render --gpus 1 --shards 8: the C++ host's sharded frame (device-generated seeds, one context, no
    collective at one GPU) is the sum of the 8 shards rendered one by one through the C ABI, and equals the Python
    binding's render_sharded_frame."""
    import importlib
    import torch
    from util import entry
    mg = importlib.import_module(entry.PKG_NAME + ".multi_gpu")
    out = str(tmp_path / "sharded")
    W, SPP, V = 256, 20, 8
    args = [EXE, "--scene", "readme", "--width", str(W), "--spp", str(SPP), "--depth", "50", "--blocks", "4", "--seed", "77",
            "--gpus", "1", "--shards", str(V), "--out", out]
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    spp = pkg.split_spp(SPP, V)
    assert spp == [3, 3, 3, 3, 2, 2, 2, 2]
    cam = pkg.camera_readme(W, SPP, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    gpu_ctx.upload_scene(pkg.scene_readme())
    acc = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
    for v in range(V):
        c = cam.copy()
        c.samplesPerPixel = spp[v]
        gpu_ctx.upload_seeds(pkg.make_seeds(bw * bh, 77, shard=v))
        acc = acc + gpu_ctx.render_frame(c, pkg.OUT_SUM)
    got = _read_pfm(out + ".pfm")
    assert np.array_equal(bits(got), bits(acc / np.float32(SPP)))
    ppm = open(out + ".ppm", "rb").read()
    assert ppm[len(b"P6\n256 144\n255\n"):] == pkg.tonemap_host(acc, 1.0 / SPP).tobytes()
    # the Python binding's shard loop gives the same sum and picture
    frame = mg.FrameBuffers(cam.width, cam.height, torch.device("cuda", 0))
    mg.render_sharded_frame(gpu_ctx, cam, frame, V, 77, 0, 1, shard_spp=spp)
    gpu_ctx.synchronize()
    assert np.array_equal(bits(frame.image().cpu().numpy()), bits(acc))
    assert np.array_equal(frame.picture().cpu().numpy(), pkg.tonemap_host(acc, 1.0 / SPP))
    # and so does the one-thread n-GPU entry point at n = 1
    assert np.array_equal(bits(pkg.render_sharded_frame([gpu_ctx], cam, V, 77, shard_spp=spp)), bits(acc))

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

"""Helpers shared by the tests: package loaders and scene construction from fixtures."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
orc = entry.load_oracle()


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_meta(z):
    return json.loads(bytes(z["meta"]).decode())


triangles_from_soup = pkg.triangles_from_soup


def scene_main_cpp_with_mesh(mesh="f1_car"):
    """main.cpp:98-129 scene built from the committed triangle soup (no OBJ file needed)."""
    return pkg.scene_main_cpp_from_soup(golden("meshes.npz")[mesh])


def scene_by_tag(tag):
    if tag.startswith("readme"):
        return pkg.scene_readme()
    if tag.startswith("maincpp_f1car"):
        return scene_main_cpp_with_mesh("f1_car")
    if tag.startswith("maincpp"):
        return pkg.scene_main_cpp()
    if tag.startswith("stress"):
        return pkg.scene_stress(60, 48, 48, 6, 6)
    raise KeyError(tag)


def camera_by_meta(meta):
    if meta["tag"].startswith("readme"):
        cam = pkg.camera_readme(meta["width"], meta["spp"], meta["depth"])
    elif meta["tag"].startswith("stress"):
        cam = pkg.camera_stress(meta["width"], meta["spp"], meta["depth"])
    else:
        cam = pkg.camera_main_cpp(meta["width"], meta["spp"], meta["depth"])
    bw, bh = pkg.set_blocks(cam, meta["blocks"])
    assert (bw, bh) == (meta["block_w"], meta["block_h"])
    return cam


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def image_error(a, b):
    """(mean abs error, PSNR in dB for peak 1.0) between two linear float images."""
    d = a.astype(np.float64) - b.astype(np.float64)
    mae = float(np.abs(d).mean())
    mse = float((d * d).mean())
    psnr = float("inf") if mse == 0 else 10.0 * np.log10(1.0 / mse)
    return mae, psnr

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


def triangles_from_soup(soup, scale, translate, mat, box_id=0):
    """Triangle array + bounding box from a raw [T,9] float32 soup, with the reference loader's
    arithmetic: vertex = v*scale + translate in float (utility.cpp:115-119) and the box
    maximum initialised to FLT_MIN (main.cpp:114-115)."""
    s = np.asarray(scale, dtype=np.float32)
    t = np.asarray(translate, dtype=np.float32)
    v = soup.reshape(-1, 3, 3).astype(np.float32)
    w = (v * s[None, None, :]).astype(np.float32) + t[None, None, :]
    w = w.astype(np.float32)
    n = w.shape[0]
    tris = (pkg.Triangle * n)()
    arr = np.frombuffer(tris, dtype=np.float32).reshape(n, 16)
    arr[:, :9] = w.reshape(n, 9)
    arr[:, 9] = mat.albedo.x
    arr[:, 10] = mat.albedo.y
    arr[:, 11] = mat.albedo.z
    arr[:, 12] = mat.fuzz
    arr[:, 13] = mat.ior
    arr.view(np.int32)[:, 14] = mat.type
    arr.view(np.int32)[:, 15] = box_id
    flat = w.reshape(-1, 3)
    mn = np.minimum(flat.min(axis=0), np.float32(np.finfo(np.float32).max))
    mx = np.maximum(flat.max(axis=0), np.float32(np.finfo(np.float32).tiny))
    return tris, mn, mx


def scene_main_cpp_with_mesh(mesh="f1_car"):
    """main.cpp:98-129 scene built from the committed triangle soup (no OBJ file needed)."""
    soup = golden("meshes.npz")[mesh]
    base = pkg.scene_main_cpp()
    tris, mn, mx = triangles_from_soup(soup, (0.15, 0.15, 0.15), (0.0, -0.5, -1.0),
                                       pkg.material((0.8, 0.6, 0.2), 0.5, 0.0, pkg.METAL), 0)
    boxes = (pkg.BoundingBox * 1)()
    boxes[0] = pkg.BoundingBox(0, pkg.vec3(*mn), pkg.vec3(*mx))
    return pkg.Scene(spheres=base.spheres, triangles=tris, boxes=boxes)


def scene_by_tag(tag):
    if tag.startswith("readme"):
        return pkg.scene_readme()
    if tag.startswith("maincpp_f1car"):
        return scene_main_cpp_with_mesh("f1_car")
    if tag.startswith("maincpp"):
        return pkg.scene_main_cpp()
    raise KeyError(tag)


def camera_by_meta(meta):
    if meta["tag"].startswith("readme"):
        cam = pkg.camera_readme(meta["width"], meta["spp"], meta["depth"])
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

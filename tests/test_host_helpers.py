"""CPU tests of the C ABI's host half: the library loads and exports every declared symbol,
and the host helpers (camera, block planner, seeds, tone map, writers, loaders) behave as
the reference's host functions do."""
import os
import re
import struct

import pytest

import numpy as np

from util import pkg, ROOT, golden


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ocl_rt_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(wbrt_[a-z0-9_]+)\s*\(", header)))
    assert declared == pkg.EXPORTED_SYMBOLS
    lib = pkg.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert os.path.dirname(pkg.LIB_PATH).endswith("opencl-ray-tracer_amd")


def test_image_height_and_block_planner():
    # globals.hpp:8-10 and utility.cpp:192-236 (values checked against the reference's rule by hand)
    assert [pkg.image_height(w) for w in (1280, 1920, 3840, 64)] == [720, 1080, 2160, 36]
    assert pkg.ideal_block_size(1280, 720, 4) == (640, 360)
    assert pkg.ideal_block_size(1920, 1080, 4) == (960, 540)
    assert pkg.ideal_block_size(3840, 2160, 4) == (1920, 1080)
    assert pkg.ideal_block_size(1280, 720, 1) == (1280, 720)
    # largest product <= ceil(pixels/target) wins, ties go to the most square pair
    bw, bh = pkg.ideal_block_size(1280, 720, 3)
    assert 1280 % bw == 0 and 720 % bh == 0 and bw * bh <= -(-1280 * 720 // 3)
    best = max(fw * fh for fw in range(1, 1281) if 1280 % fw == 0 for fh in range(1, 721)
               if 720 % fh == 0 and fw * fh <= -(-1280 * 720 // 3))
    assert bw * bh == best


def test_camera_readme_values():
    cam = pkg.camera_readme(1280, 250, 50)
    assert (cam.width, cam.height, cam.samplesPerPixel, cam.maxDepth) == (1280, 720, 250, 50)
    assert cam.origin.tuple() == (0.0, 0.0, 0.0)
    # fov 90, focus 1: viewport height 2*tan(pi/4) ~ 2, width = 1.7777777*height
    assert abs(cam.vertical.y - 2.0) < 1e-6 and abs(cam.horizontal.x - 2.0 * 1.7777777) < 1e-5
    assert abs(cam.lowerLeftCorner.x + 1.7777777) < 1e-5 and abs(cam.lowerLeftCorner.y + 1.0) < 1e-6
    assert abs(cam.lowerLeftCorner.z + 1.0) < 1e-7
    assert cam.aperture == 0.0
    m = pkg.camera_main_cpp(1280, 5, 32)
    # aperture is converted from degrees to radians and kept in the struct (cl_camera.cpp:6,44)
    assert abs(m.aperture - 1.2 * np.float32(3.141592654) / 180.0) < 1e-8
    assert m.focusDistance == 2.0
    u = np.array(m.horizontal.tuple())
    v = np.array(m.vertical.tuple())
    assert abs(np.dot(u, v)) < 1e-5
    d = np.array(m.defocusDiscU.tuple())
    assert abs(np.linalg.norm(d) - 2.0 * np.tan(m.aperture / 2)) < 1e-6


def test_seeds_are_deterministic_and_sharded():
    a = pkg.make_seeds(1000, 5, 0)
    assert np.array_equal(a, pkg.make_seeds(1000, 5, 0))
    assert not np.array_equal(a, pkg.make_seeds(1000, 5, 1))
    assert not np.array_equal(a, pkg.make_seeds(1000, 6, 0))
    assert np.array_equal(a[:10], pkg.make_seeds(10, 5, 0))
    assert len(np.unique(a)) > 990


def test_tonemap_and_writers(tmp_path):
    img = np.zeros((2, 3, 3), dtype=np.float32)
    img[0, 0] = [0.0, 0.25, 1.0]   # bottom-left pixel
    img[1, 2] = [0.5, 0.81, 0.04]  # top-right pixel
    rgb = pkg.tonemap_host(img)
    assert rgb.shape == (2, 3, 3)
    assert rgb[1, 0].tolist() == [0, 127, 255]           # bottom row comes last (top-down output)
    assert rgb[0, 2].tolist() == [int(np.sqrt(0.5) * 255), 229, 51]
    assert pkg.tonemap_host(img * 4, 0.25)[1, 0].tolist() == [0, 127, 255]
    p = str(tmp_path / "a.ppm")
    pkg.write_ppm(p, rgb)
    data = open(p, "rb").read()
    assert data.startswith(b"P6\n3 2\n255\n") and data[11:] == rgb.tobytes()
    p = str(tmp_path / "a.pfm")
    pkg.write_pfm(p, img)
    data = open(p, "rb").read()
    assert data.startswith(b"PF\n3 2\n-1.0\n") and data[len(b"PF\n3 2\n-1.0\n"):] == img.tobytes()
    p = str(tmp_path / "a.bmp")
    pkg.write_bmp(p, rgb)
    data = open(p, "rb").read()
    stride = 12  # 3 px * 3 B = 9, padded to 12
    assert data[:2] == b"BM" and struct.unpack("<I", data[2:6])[0] == 54 + stride * 2 == len(data)
    assert data[10] == 54 and data[14] == 40
    assert struct.unpack("<ii", data[18:26]) == (3, 2) and data[26] == 1 and data[28] == 24
    first_row = data[54:54 + 9]  # bottom row, B,G,R
    assert list(first_row[:3]) == [255, 127, 0]


def test_obj_loader(tmp_path):
    obj = tmp_path / "m.obj"
    obj.write_text("# test\no thing\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 -2.5\nvn 0 0 1\n"
                   "s off\nf 1//1 2//1 3//1\nf 1/1/1 3/1/1 4/1/1\n")
    mat = pkg.material((0.8, 0.6, 0.2), 0.5, 0.0, pkg.METAL)
    tris, mn, mx, nv = pkg.read_mesh(str(obj), (2, 2, 2), (0.5, -1.0, -0.25), mat, 3)
    assert nv == 4 and len(tris) == 2
    assert tris[0].p1.tuple() == (2.5, -1.0, -0.25) and tris[0].p2.tuple() == (0.5, 1.0, -0.25)
    assert tris[1].p2.tuple() == (0.5, -1.0, -5.25)
    assert tris[1].boundingBoxId == 3 and tris[1].material.type == pkg.METAL and tris[1].material.fuzz == 0.5
    assert mn.tuple() == (0.5, -1.0, -5.25)
    # maximum starts at FLT_MIN (main.cpp:114-115): an all-negative axis keeps 1.17549435e-38
    assert mx.x == 2.5 and mx.y == 1.0 and mx.z == np.float32(np.finfo(np.float32).tiny)


def test_reference_mesh_fixture_bounds():
    # the f1_car soup with main.cpp's transform reproduces the bounding box the reference's loader computes for it
    from util import scene_main_cpp_with_mesh
    sc = scene_main_cpp_with_mesh("f1_car")
    assert sc.T == 15329 and sc.B == 1 and sc.S == 8
    b = sc.boxes[0]
    assert np.allclose(b.min.tuple(), (-1.1281, -0.4965, -1.3759), atol=1e-4)
    assert abs(b.max.x - 0.9657) < 1e-4 and b.max.y == b.max.z == np.float32(np.finfo(np.float32).tiny)
    assert golden("meshes.npz")["burger"].shape == (828, 9)


def _rgbe(rgb):
    m = float(max(rgb))
    if m < 1e-32:
        return (0, 0, 0, 0)
    e = int(np.floor(np.log2(m))) + 1
    s = 256.0 / 2.0 ** e
    return (int(rgb[0] * s), int(rgb[1] * s), int(rgb[2] * s), e + 128)


def test_hdr_loader_and_sky_tonemap(tmp_path):
    w, h = 16, 3
    rs = np.random.RandomState(0)
    px = rs.uniform(0, 4, (h, w, 3))
    px[1, 3:12] = px[1, 3]  # a run
    quads = np.array([[_rgbe(px[y, x]) for x in range(w)] for y in range(h)], dtype=np.uint8)
    body = bytearray()
    for y in range(h):
        body += bytes([2, 2, (w >> 8) & 0xFF, w & 0xFF])
        for comp in range(4):
            row = quads[y, :, comp]
            x = 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    body += bytes([128 + run, int(row[x])])
                    x += run
                else:
                    lit = 1
                    while x + lit < w and lit < 128 and not (x + lit + 2 < w and row[x + lit] == row[x + lit + 1] == row[x + lit + 2]):
                        lit += 1
                    body += bytes([lit]) + bytes(int(v) for v in row[x:x + lit])
                    x += lit
    path = tmp_path / "sky.hdr"
    path.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + bytes(body))
    img = pkg.load_hdr(str(path))
    assert img.shape == (h, w, 3)
    want = quads[..., :3].astype(np.float32) / 256.0 * (2.0 ** (quads[..., 3:4].astype(np.float32) - 128))
    assert np.array_equal(img, want.astype(np.float32))
    tm = pkg.sky_tonemap(img)
    ref = (1 + 1 / np.power(-1 - img.astype(np.float64), 5)).astype(np.float32)
    assert np.allclose(tm, ref, rtol=1e-6) and tm.min() >= 0.0 and tm.max() < 1.0


def test_scene_fixtures():
    s = pkg.scene_readme()
    assert s.S == 5 and s.spheres[0].radius == 100.0 and s.spheres[4].material.type == pkg.METAL
    assert s.spheres[3].material.fuzz == 0.5 and s.spheres[0].material.albedo.tuple() == (0.0, np.float32(0.8), np.float32(0.7))
    m = pkg.scene_main_cpp()
    assert m.S == 8 and [m.spheres[i].material.type for i in range(8)] == [0, 0, 1, 1, 2, 3, 1, 1]
    st = pkg.scene_stress(50, 8, 6, 2, 3)
    assert st.S == 50 and st.T == 2 * 8 * 6 and st.B == 6
    ids = np.array([st.triangles[i].boundingBoxId for i in range(st.T)])
    assert np.all(np.diff(ids) >= 0) and set(ids) == set(range(6))
    sky = pkg.sky_gradient(64, 32)
    assert sky.shape == (32, 64, 3) and np.allclose(sky[0, 0], (0.5, 0.7, 1.0)) and np.allclose(sky[-1, 0], (1, 1, 1))


# ---- pinned against the reference's own host functions (tests/golden/host_kat.npz, generated by
# tests/golden/make_golden.py from oracle/_ref/libwbref_host.so)
def test_camera_matches_reference_bytes():
    z = golden("host_kat.npz")
    for prm, want in zip(z["cam_params"], z["cam_bytes"]):
        cam = pkg.calculate_camera(prm[0:3], prm[3:6], prm[6:9], float(prm[9]), float(prm[10]), float(prm[11]),
                                   float(prm[12]), 1280, 720, 5, 32)
        got = np.frombuffer(bytes(cam), dtype=np.uint8).copy()
        # blockSize (bytes 72..83) is not set by CalculateCamera; main.cpp:92 assigns it afterwards
        got[72:84] = 0
        want = want.copy()
        want[72:84] = 0
        assert np.array_equal(got, want), prm


def test_block_planner_matches_reference():
    z = golden("host_kat.npz")
    for t, (bw, bh) in zip(z["block_targets"], z["blocks"]):
        assert pkg.ideal_block_size(1280, 720, int(t)) == (int(bw), int(bh)), t


def test_mesh_boxes_match_reference():
    from util import triangles_from_soup
    z = golden("host_kat.npz")
    m = golden("meshes.npz")
    mat = pkg.material((0.8, 0.6, 0.2), 0.5, 0.0, pkg.METAL)
    for name in ("f1_car", "burger"):
        _, mn, mx = triangles_from_soup(m[name], (0.15, 0.15, 0.15), (0.0, -0.5, -1.0), mat, 0)
        assert np.array_equal(np.array([mn, mx], dtype=np.float32), z["box_" + name])


def test_hdr_loader_and_tonemap_match_reference(tmp_path):
    z = golden("host_kat.npz")
    p = tmp_path / "sky.hdr"
    p.write_bytes(z["hdr_file"].tobytes())
    raw = pkg.load_hdr(str(p))
    assert np.array_equal(raw.view(np.uint32), z["hdr_raw"].view(np.uint32))
    assert np.array_equal(pkg.sky_tonemap(raw).view(np.uint32), z["hdr_toned"].view(np.uint32))


def test_write_colour_and_bmp_match_reference(tmp_path):
    z = golden("host_kat.npz")
    vals = z["colour_in"]
    img = vals.reshape(1, -1, 3)
    rgb = pkg.tonemap_host(img)[0]
    assert np.array_equal(rgb[:, ::-1], z["colour_bgr"])  # the reference stores B,G,R
    bgr_bottom_up = z["bmp_in"]                              # rows bottom-up, B,G,R
    rgb_top_down = bgr_bottom_up[::-1, :, ::-1]
    p = tmp_path / "t.bmp"
    pkg.write_bmp(str(p), np.ascontiguousarray(rgb_top_down))
    assert p.read_bytes() == z["bmp_file"].tobytes()


def test_exr_writer_round_trip(tmp_path):
    """Uncompressed OpenEXR scanline file: parsed back here by the published layout (magic,
    version, attribute list, offset table, per-line y + size + channel-planar B,G,R floats)."""
    import struct
    rs = np.random.RandomState(3)
    h, w = 5, 7
    img = rs.uniform(0, 4, (h, w, 3)).astype(np.float32)  # row 0 = bottom
    path = str(tmp_path / "x.exr")
    pkg.write_exr(path, img)
    b = open(path, "rb").read()
    assert struct.unpack_from("<II", b, 0) == (20000630, 2)
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); typ = b[pos:e].decode(); pos = e + 1
        (size,) = struct.unpack_from("<i", b, pos); pos += 4
        attrs[name] = (typ, b[pos:pos + size]); pos += size
    pos += 1
    assert set(attrs) >= {"channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio",
                          "screenWindowCenter", "screenWindowWidth"}
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    assert struct.unpack("<4i", attrs["dataWindow"][1]) == (0, 0, w - 1, h - 1)
    ch, names = attrs["channels"][1], []
    q = 0
    while ch[q] != 0:
        e = ch.index(b"\0", q); names.append(ch[q:e].decode()); q = e + 1
        assert struct.unpack_from("<i", ch, q)[0] == 2 and struct.unpack_from("<ii", ch, q + 8) == (1, 1)
        q += 16
    assert names == ["B", "G", "R"]
    offsets = struct.unpack_from("<%dQ" % h, b, pos)
    out = np.zeros((h, w, 3), np.float32)
    for y, off in enumerate(offsets):
        yy, size = struct.unpack_from("<ii", b, off)
        assert (yy, size) == (y, w * 12)
        planes = np.frombuffer(b, dtype="<f4", count=3 * w, offset=off + 8).reshape(3, w)
        out[h - 1 - y, :, 2], out[h - 1 - y, :, 1], out[h - 1 - y, :, 0] = planes[0], planes[1], planes[2]
    assert offsets[-1] + 8 + w * 12 == len(b)
    assert np.array_equal(out.view(np.uint32), img.view(np.uint32))


def test_setters_reject_a_null_context():
    """Entry points that only change settings report an error for a null context without touching a GPU."""
    lib = pkg.lib()
    for name in ("wbrt_set_schedule_reuse", "wbrt_set_mesh_mode", "wbrt_set_timing"):
        assert getattr(lib, name)(None, 1) < 0
        assert lib.wbrt_last_error()


def test_shard_helpers_through_the_c_abi():
    """Shard assignment and sample split of the C ABI (wbrt_shards_of_rank, wbrt_split_spp): every shard has exactly
    one owner, ranks differ by at most one shard, the samples add up."""
    for V in (1, 2, 5, 8, 13):
        for K in (1, 2, 4, 8):
            parts = [pkg.shards_of_rank(V, r, K) for r in range(K)]
            assert sorted(sum(parts, [])) == list(range(V))
            assert all(v % K == r for r, p in enumerate(parts) for v in p)
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert pkg.split_spp(250, 8) == [32, 32, 31, 31, 31, 31, 31, 31]
    assert pkg.split_spp(1000, 8) == [125] * 8
    assert pkg.split_spp(3, 8) == [1, 1, 1, 0, 0, 0, 0, 0]
    with pytest.raises(ValueError):
        pkg.shards_of_rank(4, 2, 2)
    # a communicator over one GPU needs neither RCCL nor a device buffer exchange; without any GPU it fails cleanly
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(pkg.WbrtError):
            pkg.Comm([0])


def test_rccl_load_failure_is_a_clean_error():
    """The path wbrt_comm_create takes when RCCL cannot be loaded: WBRT_ERR_NO_DEVICE and a message that names the
    loader's reason (no crash on the loader's cleared error string)."""
    with pytest.raises(pkg.WbrtError) as e:
        pkg.rccl_probe("/nonexistent/dir/librccl_absent.so")
    assert e.value.code == pkg.ERR_NO_DEVICE
    assert "RCCL not found" in str(e.value) and "librccl_absent" in str(e.value)
    # twice in a row: the second failure reports its own reason too
    with pytest.raises(pkg.WbrtError) as e:
        pkg.rccl_probe("/nonexistent/dir/librccl_absent2.so")
    assert "librccl_absent2" in str(e.value)
    # a library that opens but is not RCCL
    with pytest.raises(pkg.WbrtError) as e:
        pkg.rccl_probe("libm.so.6")
    assert "lacks a required symbol" in str(e.value)


def test_box_splitter_preserves_order_and_the_picture():
    """wbrt_split_mesh_boxes: contiguous runs, ids ascending with the array, every triangle inside its (padded) box; the
    split scene gives the oracle (the reference's loops) the same image and primary ids as the one-box scene."""
    from util import orc, scene_main_cpp_with_mesh, bits
    one = scene_main_cpp_with_mesh("burger")
    split = scene_main_cpp_with_mesh("burger")
    boxes = pkg.split_mesh_boxes(split.triangles, 12, first_box_id=0, pad=1e-4)
    assert len(boxes) == 12
    ids = np.frombuffer(split.triangles, dtype=np.int32).reshape(-1, 16)[:, 15]
    assert (np.diff(ids) >= 0).all() and ids[0] == 0 and ids[-1] == 11
    counts = np.bincount(ids)
    assert counts.max() - counts.min() <= 1
    pts = np.frombuffer(split.triangles, dtype=np.float32).reshape(-1, 16)[:, :9].reshape(-1, 3, 3)
    for k in range(12):
        sel = pts[ids == k].reshape(-1, 3)
        assert boxes[k].id == k
        assert (sel.min(0) > np.array(boxes[k].min.tuple())).all() and (sel.max(0) < np.array(boxes[k].max.tuple())).all()
    # geometry untouched
    a = np.frombuffer(one.triangles, dtype=np.float32).reshape(-1, 16)[:, :15]
    b = np.frombuffer(split.triangles, dtype=np.float32).reshape(-1, 16)[:, :15]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    split_scene = pkg.Scene(spheres=one.spheres, triangles=split.triangles, boxes=boxes)
    cam = pkg.camera_main_cpp(96, 3, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    seeds = pkg.make_seeds(bw * bh, 19)
    want, got = orc.Oracle(one), orc.Oracle(split_scene)
    assert np.array_equal(bits(got.render_frame(cam, seeds)), bits(want.render_frame(cam, seeds)))
    assert np.array_equal(got.primary_ids(cam, seeds), want.primary_ids(cam, seeds))
    # the FLT_MIN quirk: maxima never below 1.17549435e-38
    q = pkg.split_mesh_boxes(scene_main_cpp_with_mesh("burger").triangles, 3, flt_min_quirk=True)
    assert all(b.max.y >= 1.17549435e-38 and b.max.z >= 1.17549435e-38 for b in q)

"""Generates the golden fixtures in this directory by running the REFERENCE's own kernel
file (compiled for the host by oracle/Makefile into oracle/_ref/libwbref.so).

Run in the build container, where /root/reference exists:

    make -C oracle ref && python tests/golden/make_golden.py

Only data lands here: inputs (seeds, rays, primitives, the triangle soup parsed from the
reference's OBJ assets) and the reference's outputs (states, floats, hit records, float
images, primary-hit ids).  Canonical arithmetic of these vectors: x86-64 baseline (no FMA),
-ffp-contract=off, IEEE divide/sqrt, glibc 2.35 float atan2f/asinf/powf.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
orc = g.load_oracle()
ref = orc.ref_lib()

REF_ASSETS = "/root/reference/src/assets/models"


class Ray(C.Structure):
    _fields_ = [("origin", pkg.Vec3), ("direction", pkg.Vec3), ("inv", pkg.Vec3)]


class HitRecord(C.Structure):
    _fields_ = [("p", pkg.Vec3), ("normal", pkg.Vec3), ("t", C.c_float), ("frontFace", C.c_bool),
                ("material", pkg.Material)]


assert C.sizeof(Ray) == 36 and C.sizeof(HitRecord) == 56


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote %s (%d bytes)" % (name, os.path.getsize(path)))


def mkray(o, d):
    r = Ray()
    r.origin = pkg.vec3(*o)
    r.direction = pkg.vec3(*d)
    with np.errstate(divide="ignore"):
        inv = np.float32(1.0) / np.asarray(d, dtype=np.float32)
    r.inv = pkg.vec3(*inv)
    return r


def rng_kat():
    rs = np.random.RandomState(11)
    seeds = np.concatenate([[12345, 0, 1, 0xFFFFFFFF, 0x80000000], rs.randint(0, 2 ** 32, 250, dtype=np.uint64)]).astype(np.uint64)
    steps = 8
    states = np.zeros((len(seeds), steps), dtype=np.uint64)
    floats = np.zeros((len(seeds), steps), dtype=np.float32)
    ref.wbref_rand_float.argtypes = [C.POINTER(C.c_uint64)]
    for i, s in enumerate(seeds):
        st = C.c_uint64(int(s))
        for k in range(steps):
            floats[i, k] = ref.wbref_rand_float(C.byref(st))
            states[i, k] = st.value
    assert states[0, 0] == 29803012144720 and abs(floats[0, 0] - 0.10588157) < 1e-8
    # states whose top bits force the float to round up to exactly 1.0
    edge = np.array([(0xFFFFFFFF << 16) | 0xFFFF, (0xFFFFFF80 << 16), (0xFFFFFF7F << 16)], dtype=np.uint64)
    save("rng_kat.npz", seeds=seeds, states=states, floats=floats, edge_states=edge)


def fn_kat():
    rs = np.random.RandomState(5)
    n = 4000
    ref.wbref_hit_sphere.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    ref.wbref_hit_triangle.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    ref.wbref_hit_bbox.argtypes = [C.c_void_p, C.c_void_p]
    o = rs.uniform(-2, 2, (n, 3)).astype(np.float32)
    d = rs.uniform(-1, 1, (n, 3)).astype(np.float32)
    d[::50, 0] = 0.0  # axis-parallel rays (infinite inverse direction)
    tmax = np.where(rs.rand(n) < 0.5, np.inf, rs.uniform(0.5, 6, n)).astype(np.float32)
    # spheres
    c = rs.uniform(-2, 2, (n, 3)).astype(np.float32)
    rad = rs.uniform(0.1, 2.0, n).astype(np.float32)
    sp_hit = np.zeros(n, dtype=np.int32)
    sp_rec = np.zeros((n, 8), dtype=np.float32)  # t, p, normal, front
    for i in range(n):
        s = pkg.Sphere(pkg.vec3(*c[i]), float(rad[i]), pkg.material((0.5, 0.5, 0.5)))
        r = mkray(o[i], d[i])
        h = HitRecord()
        sp_hit[i] = ref.wbref_hit_sphere(C.byref(s), C.byref(r), 0.001, float(tmax[i]), C.byref(h))
        sp_rec[i] = [h.t, h.p.x, h.p.y, h.p.z, h.normal.x, h.normal.y, h.normal.z, float(h.frontFace)]
    # triangles
    p = rs.uniform(-2, 2, (n, 9)).astype(np.float32)
    tr_hit = np.zeros(n, dtype=np.int32)
    tr_rec = np.zeros((n, 8), dtype=np.float32)
    for i in range(n):
        t = pkg.Triangle(pkg.vec3(*p[i, 0:3]), pkg.vec3(*p[i, 3:6]), pkg.vec3(*p[i, 6:9]), pkg.material((0.5, 0.5, 0.5)), 0)
        r = mkray(o[i], d[i])
        h = HitRecord()
        tr_hit[i] = ref.wbref_hit_triangle(C.byref(t), C.byref(r), 0.001, float(tmax[i]), C.byref(h))
        tr_rec[i] = [h.t, h.p.x, h.p.y, h.p.z, h.normal.x, h.normal.y, h.normal.z, float(h.frontFace)]
    # boxes
    lo = rs.uniform(-2, 1, (n, 3)).astype(np.float32)
    hi = (lo + rs.uniform(0.0, 2, (n, 3))).astype(np.float32)
    bb_hit = np.zeros(n, dtype=np.int32)
    for i in range(n):
        b = pkg.BoundingBox(0, pkg.vec3(*lo[i]), pkg.vec3(*hi[i]))
        r = mkray(o[i], d[i])
        bb_hit[i] = ref.wbref_hit_bbox(C.byref(b), C.byref(r))
    print("sphere hits %d, triangle hits %d, box hits %d of %d" % (sp_hit.sum(), tr_hit.sum(), bb_hit.sum(), n))
    save("fn_kat.npz", o=o, d=d, tmax=tmax, sph_c=c, sph_r=rad, sph_hit=sp_hit, sph_rec=sp_rec,
         tri_p=p, tri_hit=tr_hit, tri_rec=tr_rec, box_lo=lo, box_hi=hi, box_hit=bb_hit)


def scatter_kat():
    rs = np.random.RandomState(9)
    n = 1500
    ref.wbref_scatter.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]
    ref.wbref_to_uv.argtypes = [C.c_void_p, C.c_void_p]
    din = rs.uniform(-1, 1, (n, 3)).astype(np.float32)
    nrm = rs.standard_normal((n, 3)).astype(np.float32)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    pt = rs.uniform(-2, 2, (n, 3)).astype(np.float32)
    kind = rs.randint(0, 3, n).astype(np.int32)
    front = rs.randint(0, 2, n).astype(np.int32)
    fuzz = np.where(rs.rand(n) < 0.3, 0.0, rs.uniform(0, 1, n)).astype(np.float32)
    ior = rs.uniform(1.1, 2.4, n).astype(np.float32)
    seeds = rs.randint(0, 2 ** 32, n, dtype=np.uint64)
    ok = np.zeros(n, dtype=np.int32)
    att = np.zeros((n, 3), dtype=np.float32)
    dout = np.zeros((n, 3), dtype=np.float32)
    s_after = np.zeros(n, dtype=np.uint64)
    for i in range(n):
        h = HitRecord()
        h.p = pkg.vec3(*pt[i]); h.normal = pkg.vec3(*nrm[i]); h.t = 1.0; h.frontFace = bool(front[i])
        h.material = pkg.material((0.3, 0.6, 0.9), float(fuzz[i]), float(ior[i]), int(kind[i]))
        r = mkray(pt[i], din[i])
        st = C.c_uint64(int(seeds[i]))
        a = pkg.Vec3(); ro = Ray()
        ok[i] = ref.wbref_scatter(int(kind[i]), C.byref(r), C.byref(h), C.byref(st), C.byref(a), C.byref(ro))
        att[i] = a.tuple(); dout[i] = ro.direction.tuple(); s_after[i] = st.value
    dirs = rs.standard_normal((n, 3)).astype(np.float32)
    dirs[0] = [0, -1, 0]; dirs[1] = [0, 1, 0]; dirs[2] = [-1, 0, 0]; dirs[3] = [-1, 0, -0.0]
    uv = np.zeros((n, 2), dtype=np.float32)
    for i in range(n):
        v = pkg.vec3(*dirs[i]); out = (C.c_float * 2)()
        ref.wbref_to_uv(C.byref(v), out)
        uv[i] = [out[0], out[1]]
    save("scatter_kat.npz", din=din, normal=nrm, p=pt, kind=kind, front=front, fuzz=fuzz, ior=ior, seeds=seeds,
         ok=ok, atten=att, dout=dout, seed_after=s_after, uv_dirs=dirs, uv=uv)


def tri_soup(tris):
    return np.frombuffer(tris, dtype=np.float32).reshape(len(tris), 16)[:, :9].copy()


def meshes():
    out = {}
    for name in ("f1_car", "burger"):
        tris, mn, mx, nv = pkg.read_mesh(os.path.join(REF_ASSETS, name + ".obj"), (1, 1, 1), (0, 0, 0),
                                         pkg.material((0.8, 0.6, 0.2), 0.5, 0.0, pkg.METAL), 0)
        out[name] = tri_soup(tris)
        print(name, "vertices", nv, "triangles", len(tris))
    save("meshes.npz", **out)
    return out


def image_case(tag, scene, cam, nblocks, seed):
    bw, bh = pkg.set_blocks(cam, nblocks)
    seeds = pkg.make_seeds(bw * bh, seed)
    R = orc.Reference(scene)
    img = R.render_frame(cam, seeds)
    ids = R.primary_ids(cam, seeds)
    meta = dict(tag=tag, width=cam.width, height=cam.height, spp=cam.samplesPerPixel, depth=cam.maxDepth,
                blocks=nblocks, block_w=bw, block_h=bh, seed=seed, S=scene.S, T=scene.T, B=scene.B)
    save("img_%s.npz" % tag, image=img, ids=ids.astype(np.int32), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))


def write_rle_hdr(path, px):
    """Writes a Radiance RGBE file with adaptive RLE scanlines from an [h,w,3] float array."""
    h, w, _ = px.shape

    def rgbe(rgb):
        m = float(max(rgb))
        if m < 1e-32:
            return (0, 0, 0, 0)
        e = int(np.floor(np.log2(m))) + 1
        sc = 256.0 / 2.0 ** e
        return (int(rgb[0] * sc), int(rgb[1] * sc), int(rgb[2] * sc), e + 128)

    quads = np.array([[rgbe(px[y, x]) for x in range(w)] for y in range(h)], dtype=np.uint8)
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
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w) + bytes(body))


def host_kat():
    """Golden vectors from the reference's HOST functions (oracle/_ref/libwbref_host.so)."""
    import tempfile
    hl = orc.ref_host_lib()
    cfg = (C.c_int * 5)()
    hl.wbrefh_config(cfg)
    assert list(cfg) == [1280, 720, 5, 32, 4]
    f3 = lambda v: (C.c_float * 3)(*v)
    hl.wbrefh_calculate_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
    rs = np.random.RandomState(17)
    cases = [((-1.3, 0.2, 0.5), (0.2, 0.0, -1.5), (0, 1, 0), 60.0, 1.7777777, 2.0, 1.2),
             ((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 1.7777777, 1.0, 0.0)]
    for _ in range(30):
        cases.append((tuple(rs.uniform(-3, 3, 3)), tuple(rs.uniform(-3, 3, 3)), (0, 1, 0), float(rs.uniform(20, 110)),
                      1.7777777, float(rs.uniform(0.5, 8)), float(rs.uniform(0, 3))))
    params = np.zeros((len(cases), 13), dtype=np.float32)
    cams = np.zeros((len(cases), 120), dtype=np.uint8)
    for i, (fr, at, up, fov, asp, foc, ap) in enumerate(cases):
        params[i] = list(fr) + list(at) + list(up) + [fov, asp, foc, ap]
        p32 = params[i]
        cam = pkg.Camera()
        hl.wbrefh_calculate_camera(C.byref(cam), f3(p32[0:3]), f3(p32[3:6]), f3(p32[6:9]), float(p32[9]), float(p32[10]), float(p32[11]), float(p32[12]))
        cams[i] = np.frombuffer(bytes(cam), dtype=np.uint8)
    # block planner at the reference's compiled-in 1280x720
    hl.wbrefh_ideal_block_size.argtypes = [C.c_int, C.c_int, C.c_void_p]
    targets = np.array([1, 2, 3, 4, 5, 6, 8, 9, 12, 16, 25, 64, 100, 360], dtype=np.int32)
    blocks = np.zeros((len(targets), 2), dtype=np.int32)
    for i, t in enumerate(targets):
        out = (C.c_float * 3)()
        hl.wbrefh_ideal_block_size(1280 * 720, int(t), out)
        blocks[i] = [int(out[0]), int(out[1])]
    # OBJ loader: the reference's triangles must equal this repo's loader bit for bit
    hl.wbrefh_read_mesh.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    mat = pkg.material((0.8, 0.6, 0.2), 0.5, 0.0, pkg.METAL)
    mesh_boxes = {}
    for name in ("f1_car", "burger"):
        path = os.path.join(REF_ASSETS, name + ".obj").encode()
        nv, nt = C.c_int(0), C.c_int(0)
        mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
        assert hl.wbrefh_read_mesh(path, f3((0.15, 0.15, 0.15)), f3((0.0, -0.5, -1.0)), C.byref(mat), None, 0, C.byref(nv), C.byref(nt), mn, mx) == 0
        tris = (pkg.Triangle * nt.value)()
        assert hl.wbrefh_read_mesh(path, f3((0.15, 0.15, 0.15)), f3((0.0, -0.5, -1.0)), C.byref(mat), tris, nt.value, C.byref(nv), C.byref(nt), mn, mx) == 0
        mine, mmn, mmx, mnv = pkg.read_mesh(path.decode(), (0.15, 0.15, 0.15), (0.0, -0.5, -1.0), mat, 0)
        assert mnv == nv.value and len(mine) == nt.value
        assert bytes(mine) == bytes(tris), "OBJ loader differs from the reference for " + name
        assert tuple(mn) == mmn.tuple() and tuple(mx) == mmx.tuple()
        mesh_boxes[name] = np.array([list(mn), list(mx)], dtype=np.float32)
        print(name, "reference loader == this repo's loader:", nt.value, "triangles, box", list(mn), list(mx))
    # HDR loader + sky tone map on a synthetic RLE file (the reference's skybox.hdr is not in its checkout)
    px = rs.uniform(0, 6, (5, 24, 3))
    px[2, 4:17] = px[2, 4]
    px[3] = 0.0
    tmp = tempfile.mkdtemp()
    hdr = os.path.join(tmp, "sky.hdr")
    write_rle_hdr(hdr, px)
    w, h = C.c_int(0), C.c_int(0)
    hl.wbrefh_load_hdr.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    assert hl.wbrefh_load_hdr(hdr.encode(), C.byref(w), C.byref(h), None, None, 0) == 0
    raw = np.zeros((h.value, w.value, 3), dtype=np.float32)
    toned = np.zeros_like(raw)
    assert hl.wbrefh_load_hdr(hdr.encode(), C.byref(w), C.byref(h), raw.ctypes.data_as(C.c_void_p), toned.ctypes.data_as(C.c_void_p), raw.size // 3) == 0
    hdr_bytes = np.frombuffer(open(hdr, "rb").read(), dtype=np.uint8)
    # WriteColour on a ladder of values
    vals = np.concatenate([np.linspace(0, 1, 4001), rs.uniform(0, 1, 4000), (np.arange(256) / 255.0) ** 2]).astype(np.float32)
    vals = vals[: (len(vals) // 3) * 3].reshape(-1, 3)
    hl.wbrefh_write_colour.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    bgr = np.zeros((len(vals), 3), dtype=np.uint8)
    for i, v in enumerate(vals):
        out = (C.c_ubyte * 3)()
        hl.wbrefh_write_colour(i % 720, i % 1280, f3(v), out)
        bgr[i] = list(out)
    # BMP writer on a 5x3 image (row padding 1 byte)
    img = rs.randint(0, 256, (3, 5, 3)).astype(np.uint8)  # B,G,R rows bottom-up, as the reference stores it
    bmp = os.path.join(tmp, "t.bmp")
    hl.wbrefh_write_bmp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
    hl.wbrefh_write_bmp(img.ctypes.data_as(C.c_void_p), 3, 5, bmp.encode())
    bmp_bytes = np.frombuffer(open(bmp, "rb").read(), dtype=np.uint8)
    save("host_kat.npz", cam_params=params, cam_bytes=cams, block_targets=targets, blocks=blocks,
         box_f1_car=mesh_boxes["f1_car"], box_burger=mesh_boxes["burger"], hdr_file=hdr_bytes, hdr_raw=raw, hdr_toned=toned,
         colour_in=vals, colour_bgr=bgr, bmp_in=img, bmp_file=bmp_bytes)


def mesh_large():
    """Mesh configuration at sizes the reference kernel still finishes in minutes on the host:
    a 480x270, 2 spp image + ids, and the primary ids of the full 1920x1080 frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    scene = util.scene_main_cpp_with_mesh("f1_car")
    image_case("maincpp_f1car_480x270_s2_b4", scene, pkg.camera_main_cpp(480, 2, 50), 4, 13)
    cam = pkg.camera_main_cpp(1920, 1, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    seeds = pkg.make_seeds(bw * bh, 0x5EED)
    ids = orc.Reference(scene).primary_ids(cam, seeds)
    print("1920x1080 mesh primary ids: %d distinct, %d misses" % (len(np.unique(ids)), int((ids < 0).sum())))
    save("ids_maincpp_f1car_1920x1080_b4.npz", ids=ids.astype(np.int16), seed=np.array([0x5EED], dtype=np.uint64))


def tolerance_case():
    """README scene at 320x180 x 250 spp from the reference kernel: the image the tolerant arithmetic mode (and the
    strict one) is held against, MAE <= 1e-3 / PSNR >= 40 dB, with the primary ids that stay bit-exact."""
    image_case("readme_320x180_s250_b4", pkg.scene_readme(), pkg.camera_readme(320, 250, 50), 4, 33)


def stress_case():
    """The divergence-stress scene in small (60 spheres, 4608 triangles in 36 bounding boxes) from the reference kernel: many
    boxes and many spheres at once, which the other image cases do not have."""
    image_case("stress_128x72_s4_b4", pkg.scene_stress(60, 48, 48, 6, 6), pkg.camera_stress(128, 4, 50), 4, 17)


def main():
    if not orc.have_ref():
        raise SystemExit("oracle/_ref/libwbref.so missing: run `make -C oracle ref` where /root/reference exists")
    sizes = (C.c_int * 8)()
    ref.wbref_sizes(sizes)
    assert list(sizes) == [12, 24, 40, 64, 28, 120, 36, 56], list(sizes)
    if "--mesh-large" in sys.argv:
        mesh_large()
        return
    if "--tolerance" in sys.argv:
        tolerance_case()
        return
    if "--stress" in sys.argv:
        stress_case()
        return
    if "--host-only" in sys.argv:
        host_kat()
        return
    host_kat()
    rng_kat()
    fn_kat()
    scatter_kat()
    meshes()
    # images + ids: README scene, 4 blocks and 1 block; main.cpp spheres; main.cpp + mesh
    image_case("readme_160x90_s8_b4", pkg.scene_readme(), pkg.camera_readme(160, 8, 50), 4, 7)
    image_case("readme_160x90_s8_b1", pkg.scene_readme(), pkg.camera_readme(160, 8, 50), 1, 7)
    image_case("readme_64x36_s250_b4", pkg.scene_readme(), pkg.camera_readme(64, 250, 50), 4, 21)
    image_case("maincpp_160x90_s8_b4", pkg.scene_main_cpp(), pkg.camera_main_cpp(160, 8, 32), 4, 5)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    sc_obj = pkg.scene_main_cpp(os.path.join(REF_ASSETS, "f1_car.obj"))
    sc_soup = util.scene_main_cpp_with_mesh("f1_car")
    assert bytes(sc_obj.triangles) == bytes(sc_soup.triangles) and bytes(sc_obj.boxes) == bytes(sc_soup.boxes)
    image_case("maincpp_f1car_96x54_s2_b4", sc_obj, pkg.camera_main_cpp(96, 2, 50), 4, 11)
    # full-size primary ids of the headline configuration (ids only; int8 compresses well)
    scene = pkg.scene_readme()
    cam = pkg.camera_readme(1280, 1, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    seeds = pkg.make_seeds(bw * bh, 0x5EED)
    ids = orc.Reference(scene).primary_ids(cam, seeds)
    vals, counts = np.unique(ids, return_counts=True)
    print("1280x720 primary ids:", dict(zip(vals.tolist(), counts.tolist())))
    save("ids_readme_1280x720_b4.npz", ids=ids.astype(np.int8), seed=np.array([0x5EED], dtype=np.uint64))
    mesh_large()
    tolerance_case()


if __name__ == "__main__":
    main()

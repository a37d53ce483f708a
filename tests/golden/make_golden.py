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


def main():
    if not orc.have_ref():
        raise SystemExit("oracle/_ref/libwbref.so missing: run `make -C oracle ref` where /root/reference exists")
    sizes = (C.c_int * 8)()
    ref.wbref_sizes(sizes)
    assert list(sizes) == [12, 24, 40, 64, 28, 120, 36, 56], list(sizes)
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


if __name__ == "__main__":
    main()

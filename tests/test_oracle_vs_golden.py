"""CPU tests: the oracle (oracle/pt_oracle.c) against golden vectors produced by the
reference's own kernel file (tests/golden/make_golden.py), bit for bit."""
import ctypes as C

import numpy as np
import pytest

from util import pkg, orc, golden, golden_meta, scene_by_tag, camera_by_meta, bits, image_error

ol = orc.oracle_lib()


class Ray(C.Structure):
    _fields_ = [("origin", pkg.Vec3), ("direction", pkg.Vec3)]


def test_struct_sizes_match_reference():
    assert [C.sizeof(t) for t in (pkg.Vec3, pkg.Material, pkg.Sphere, pkg.Triangle, pkg.BoundingBox, pkg.Camera)] == \
        [12, 24, 40, 64, 28, 120]


def test_rng_known_answers():
    z = golden("rng_kat.npz")
    ol.wbo_rand_float.argtypes = [C.POINTER(C.c_uint64)]
    for s, states, floats in zip(z["seeds"], z["states"], z["floats"]):
        st = C.c_uint64(int(s))
        for k in range(states.shape[0]):
            f = ol.wbo_rand_float(C.byref(st))
            assert st.value == int(states[k])
            assert np.float32(f).view(np.uint32) == floats[k].view(np.uint32)
    # known answer worked by hand from gpu_kernel.cl:173-180: seed 12345 -> state 29803012144720 -> 0.10588157f
    assert ol.wbo_next_seed(12345) == 29803012144720
    st = C.c_uint64(12345)
    assert abs(ol.wbo_rand_float(C.byref(st)) - 0.10588157) < 1e-8


def test_rng_reaches_one_inclusive():
    # (s >> 16) >= 0xFFFFFF80 rounds to 2^32 in float, so the draw is exactly 1.0
    # (gpu_kernel.cl:587-591).  Invert one LCG step to get a predecessor state.
    M, A, MASK = 0x5DEECE66D, 0xB, (1 << 48) - 1
    Minv = pow(M, -1, 1 << 48)
    for target_hi, expect_one in ((0xFFFFFFFF, True), (0xFFFFFF80, True), (0xFFFFFF7F, False)):
        target = (target_hi << 16) | 0x1234
        prev = ((target - A) * Minv) & MASK
        st = C.c_uint64(prev)
        ol.wbo_rand_float.argtypes = [C.POINTER(C.c_uint64)]
        f = ol.wbo_rand_float(C.byref(st))
        assert st.value == target
        assert (f == 1.0) == expect_one


def test_intersection_known_answers():
    z = golden("fn_kat.npz")
    ol.wbo_hit_sphere.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.POINTER(C.c_float)]
    ol.wbo_hit_triangle.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.POINTER(C.c_float)]
    ol.wbo_hit_bbox.argtypes = [C.c_void_p, C.c_void_p]
    n = len(z["o"])
    for i in range(n):
        r = Ray(pkg.vec3(*z["o"][i]), pkg.vec3(*z["d"][i]))
        t = C.c_float(0)
        s = pkg.Sphere(pkg.vec3(*z["sph_c"][i]), float(z["sph_r"][i]), pkg.material((0.5, 0.5, 0.5)))
        hit = ol.wbo_hit_sphere(C.byref(s), C.byref(r), 0.001, float(z["tmax"][i]), C.byref(t))
        assert hit == z["sph_hit"][i]
        if hit:
            assert np.float32(t.value).view(np.uint32) == z["sph_rec"][i, 0].view(np.uint32)
        p = z["tri_p"][i]
        tr = pkg.Triangle(pkg.vec3(*p[0:3]), pkg.vec3(*p[3:6]), pkg.vec3(*p[6:9]), pkg.material((0.5, 0.5, 0.5)), 0)
        hit = ol.wbo_hit_triangle(C.byref(tr), C.byref(r), 0.001, float(z["tmax"][i]), C.byref(t))
        assert hit == z["tri_hit"][i]
        if hit:
            assert np.float32(t.value).view(np.uint32) == z["tri_rec"][i, 0].view(np.uint32)
        b = pkg.BoundingBox(0, pkg.vec3(*z["box_lo"][i]), pkg.vec3(*z["box_hi"][i]))
        assert ol.wbo_hit_bbox(C.byref(b), C.byref(r)) == z["box_hit"][i]


def test_scatter_known_answers():
    z = golden("scatter_kat.npz")
    ol.wbo_scatter.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                               C.POINTER(C.c_uint64), C.c_int, C.c_void_p, C.c_void_p]
    for i in range(len(z["kind"])):
        r = Ray(pkg.vec3(*z["p"][i]), pkg.vec3(*z["din"][i]))
        p = pkg.vec3(*z["p"][i])
        nrm = pkg.vec3(*z["normal"][i])
        m = pkg.material((0.3, 0.6, 0.9), float(z["fuzz"][i]), float(z["ior"][i]), int(z["kind"][i]))
        st = C.c_uint64(int(z["seeds"][i]))
        att, out = pkg.Vec3(), Ray()
        ok = ol.wbo_scatter(int(z["kind"][i]), C.byref(r), C.byref(p), C.byref(nrm), int(z["front"][i]), C.byref(m),
                            C.byref(st), orc.MATH_LIBM, C.byref(att), C.byref(out))
        assert ok == z["ok"][i]
        assert st.value == int(z["seed_after"][i])
        assert np.array_equal(bits(np.array(att.tuple())), bits(z["atten"][i]))
        assert np.array_equal(bits(np.array(out.direction.tuple())), bits(z["dout"][i])), (i, z["kind"][i])


def test_sky_uv_known_answers():
    z = golden("scatter_kat.npz")
    ol.wbo_to_uv.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    worst = 0.0
    for d, uv in zip(z["uv_dirs"], z["uv"]):
        v = pkg.vec3(*d)
        out = (C.c_float * 2)()
        ol.wbo_to_uv(C.byref(v), orc.MATH_LIBM, out)
        assert np.array_equal(bits(np.array(out[:])), bits(uv))
        ol.wbo_to_uv(C.byref(v), orc.MATH_PORTABLE, out)
        worst = max(worst, abs(out[0] - uv[0]), abs(out[1] - uv[1]))
    # the portable atan2/asin agree with libm to float rounding of a value in [0,1]
    assert worst <= 2.0 ** -23


def test_portable_math_accuracy():
    rs = np.random.RandomState(3)
    y = rs.standard_normal(20000).astype(np.float32)
    x = rs.standard_normal(20000).astype(np.float32)
    got = np.array([ol.wbo_atan2f_portable(float(a), float(b)) for a, b in zip(y, x)], dtype=np.float32)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(got - want) <= 0.5000001 * ulp)  # correctly rounded to float
    u = rs.uniform(-1, 1, 20000).astype(np.float32)
    got = np.array([ol.wbo_asinf_portable(float(a)) for a in u], dtype=np.float32)
    want = np.arcsin(u.astype(np.float64))
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert np.all(np.abs(got - want) <= 0.5000001 * ulp)
    assert ol.wbo_atan2f_portable(0.0, -1.0) == np.float32(np.pi)
    assert ol.wbo_atan2f_portable(-0.0, -1.0) == -np.float32(np.pi)
    assert ol.wbo_asinf_portable(-1.0) == -np.float32(np.pi / 2)
    got = np.array([ol.wbo_pow5f_portable(float(a)) for a in u], dtype=np.float32)
    assert np.allclose(got, u.astype(np.float64) ** 5, rtol=1e-7, atol=0)


IMAGE_CASES = ["readme_160x90_s8_b4", "readme_160x90_s8_b1", "readme_64x36_s250_b4",
               "maincpp_160x90_s8_b4", "maincpp_f1car_96x54_s2_b4", "stress_128x72_s4_b4"]


@pytest.mark.parametrize("tag", IMAGE_CASES)
def test_images_and_ids_match_reference_bitwise(tag):
    z = golden("img_%s.npz" % tag)
    meta = golden_meta(z)
    scene = scene_by_tag(tag)
    cam = camera_by_meta(meta)
    seeds = pkg.make_seeds(meta["block_w"] * meta["block_h"], meta["seed"])
    O = orc.Oracle(scene)
    img = O.render_frame(cam, seeds, math_mode=orc.MATH_LIBM)
    assert np.array_equal(bits(img), bits(z["image"]))
    assert np.array_equal(O.primary_ids(cam, seeds), z["ids"])
    # portable transcendental mode: same image up to rare sky-texel flips
    img_p = O.render_frame(cam, seeds, math_mode=orc.MATH_PORTABLE)
    mae, psnr = image_error(img_p, z["image"])
    assert mae <= 1e-5 and psnr >= 60.0
    assert (bits(img_p) != bits(z["image"])).mean() < 1e-3


def test_full_size_primary_ids_match_reference():
    z = golden("ids_readme_1280x720_b4.npz")
    scene = pkg.scene_readme()
    cam = pkg.camera_readme(1280, 1, 50)
    bw, bh = pkg.set_blocks(cam, 4)
    assert (bw, bh) == (640, 360)
    seeds = pkg.make_seeds(bw * bh, int(z["seed"][0]))
    ids = orc.Oracle(scene).primary_ids(cam, seeds)
    assert np.array_equal(ids, z["ids"].astype(np.int32))


def test_sum_mode_and_counters():
    scene = pkg.scene_readme()
    cam = pkg.camera_readme(64, 6, 50)
    bw, bh = pkg.set_blocks(cam, 1)
    seeds = pkg.make_seeds(bw * bh, 3)
    O = orc.Oracle(scene)
    mean = O.render_frame(cam, seeds)
    total = O.render_frame(cam, seeds, out_mode=orc.OUT_SUM)
    assert np.array_equal(bits(total / np.float32(6)), bits(mean))
    _, ctr = O.render_frame(cam, seeds, counters=True)
    c = dict(zip(pkg.COUNTER_NAMES, map(int, ctr)))
    assert c["paths"] == 64 * 36 * 6
    assert c["sphere_tests"] == 5 * c["segments"]
    assert c["miss"] + c["depth_exhausted"] + c["absorbed"] + c["emissive"] == c["paths"]
    assert c["rng_draws"] == 2 * c["paths"] + 3 * c["sphere_tries"] + 2 * c["disk_tries"] + c["dielectric_draw"]
    assert 300 < pkg.algorithmic_flops(c) / c["paths"] < 900


@pytest.mark.skipif(not orc.have_ref(), reason="reference build absent (needs /root/reference at build time)")
@pytest.mark.parametrize("case", [("readme", 96, 5, 4, 101), ("readme", 80, 3, 1, 102), ("maincpp", 96, 5, 4, 103)])
def test_live_against_reference_build(case):
    name, width, spp, blocks, seed = case
    scene = pkg.scene_readme() if name == "readme" else pkg.scene_main_cpp()
    cam = (pkg.camera_readme if name == "readme" else pkg.camera_main_cpp)(width, spp, 50)
    bw, bh = pkg.set_blocks(cam, blocks)
    seeds = pkg.make_seeds(bw * bh, seed)
    a = orc.Oracle(scene).render_frame(cam, seeds, math_mode=orc.MATH_LIBM)
    b = orc.Reference(scene).render_frame(cam, seeds)
    assert np.array_equal(bits(a), bits(b))
    assert np.array_equal(orc.Oracle(scene).primary_ids(cam, seeds), orc.Reference(scene).primary_ids(cam, seeds))


def test_edge_cases():
    scene = pkg.scene_readme()
    O = orc.Oracle(scene)
    # zero samples: 0/0 per channel, exactly what the reference's final divide gives
    cam = pkg.camera_readme(16, 0, 50)
    pkg.set_blocks(cam, 1)
    seeds = pkg.make_seeds(16 * 9, 1)
    assert np.isnan(O.render_frame(cam, seeds)).all()
    # depth 0: every path is "exhausted" -> black (gpu_kernel.cl:337-340)
    cam = pkg.camera_readme(16, 2, 0)
    pkg.set_blocks(cam, 1)
    assert not O.render_frame(cam, seeds).any()
    # empty scene: pure sky
    empty = pkg.Scene()
    cam = pkg.camera_readme(16, 2, 5)
    pkg.set_blocks(cam, 1)
    img = orc.Oracle(empty).render_frame(cam, seeds)
    assert img.min() >= 0.5 and img.max() <= 1.0
    assert (orc.Oracle(empty).primary_ids(cam, seeds) == -1).all()

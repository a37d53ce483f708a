"""Python binding of the MI355X path tracer's C ABI (include/ocl_rt_amd.h).

The product is the shared library ``libocl_rt_amd.so`` (HIP kernels for gfx950 + C ABI +
host helpers).  This module is a thin ctypes layer over it for the test-suite, the
benchmark and multi-GPU orchestration; it contains no rendering logic and has no CPU
fallback: if the library is missing, loading fails loudly.

The struct mirrors are byte-identical to the reference's PODs
(src/opencl/gpu_kernel.cl:3-69 == src/cpp/opencl_objects/cl_*.hpp).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WBRT_LIB") or os.path.join(_HERE, "libocl_rt_amd.so")  # override: kernel A/B experiments
CSRC = os.path.join(_HERE, "csrc")

OUT_MEAN, OUT_SUM, OUT_ACCUMULATE = 0, 1, 2
LAMBERTIAN, METAL, DIELECTRIC, EMISSIVE = 0, 1, 2, 3
NUM_COUNTERS = 26
COUNTER_NAMES = [
    "paths", "segments", "rng_draws",
    "sphere_tests", "sphere_disc_ge0", "sphere_second_root", "sphere_accept",
    "bbox_tests", "bbox_hits",
    "tri_tests", "tri_exit_parallel", "tri_exit_u", "tri_exit_v", "tri_reach_t", "tri_accept",
    "lambert", "metal", "metal_fuzz", "dielectric", "dielectric_draw", "emissive",
    "sphere_tries", "disk_tries",
    "miss", "depth_exhausted", "absorbed",
]


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tuple(self):
        return (self.x, self.y, self.z)


class Material(C.Structure):
    _fields_ = [("albedo", Vec3), ("fuzz", C.c_float), ("ior", C.c_float), ("type", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("center", Vec3), ("radius", C.c_float), ("material", Material)]


class Triangle(C.Structure):
    _fields_ = [("p0", Vec3), ("p1", Vec3), ("p2", Vec3), ("material", Material), ("boundingBoxId", C.c_int32)]


class BoundingBox(C.Structure):
    _fields_ = [("id", C.c_int32), ("min", Vec3), ("max", Vec3)]


class Camera(C.Structure):
    _fields_ = [
        ("origin", Vec3), ("horizontal", Vec3), ("vertical", Vec3), ("lowerLeftCorner", Vec3),
        ("defocusDiscU", Vec3), ("defocusDiscV", Vec3), ("blockSize", Vec3), ("blockOffset", Vec3),
        ("width", C.c_int32), ("height", C.c_int32), ("samplesPerPixel", C.c_int32), ("maxDepth", C.c_int32),
        ("focusDistance", C.c_float), ("aperture", C.c_float),
    ]

    def copy(self):
        c = Camera()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(Camera))
        return c


assert C.sizeof(Vec3) == 12 and C.sizeof(Material) == 24 and C.sizeof(Sphere) == 40
assert C.sizeof(Triangle) == 64 and C.sizeof(BoundingBox) == 28 and C.sizeof(Camera) == 120


class WbrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("wbrt error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compiles csrc/ into libocl_rt_amd.so with hipcc for gfx950 (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["make", "-C", CSRC, "all"], check=True)
    return LIB_PATH


_lib = None
MESH_LOOP, MESH_BVH, MESH_BVH_CAPPED = 0, 1, 2
ARITH_STRICT, ARITH_TOLERANT = 0, 1
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_IO, ERR_CANCELLED = -1, -2, -3, -4, -5, -6

# name -> (restype, argtypes); every symbol declared in include/ocl_rt_amd.h
_P = C.c_void_p
_SIGNATURES = {
    "wbrt_last_error": (C.c_char_p, []),
    "wbrt_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "wbrt_destroy": (C.c_int, [_P]),
    "wbrt_upload_scene": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32]),
    "wbrt_upload_sky": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "wbrt_upload_seeds": (C.c_int, [_P, _P, C.c_size_t]),
    "wbrt_render_block": (C.c_int, [_P, C.POINTER(Camera), _P]),
    "wbrt_render_blocks": (C.c_int, [_P, C.POINTER(Camera), _P, _P]),
    "wbrt_render_frame": (C.c_int, [_P, C.POINTER(Camera), C.c_int, _P]),
    "wbrt_render_frame_device": (C.c_int, [_P, C.POINTER(Camera), C.c_int, _P, _P]),
    "wbrt_render_frame_device_noalloc": (C.c_int, [_P, C.POINTER(Camera), C.c_int, _P, _P]),
    "wbrt_primary_ids_frame": (C.c_int, [_P, C.POINTER(Camera), _P]),
    "wbrt_count_events_frame": (C.c_int, [_P, C.POINTER(Camera), _P]),
    "wbrt_tonemap_device": (C.c_int, [_P, _P, C.c_float, C.c_int32, C.c_int32, _P, _P]),
    "wbrt_synchronize": (C.c_int, [_P]),
    "wbrt_shards_of_rank": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, _P]),
    "wbrt_split_spp": (C.c_int32, [C.c_int32, C.c_int32, _P]),
    "wbrt_make_seeds_device": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_size_t, _P]),
    "wbrt_render_shards": (C.c_int, [_P, C.POINTER(Camera), C.c_int32, _P, C.c_uint64, C.c_int32, C.c_int32, _P, _P]),
    "wbrt_comm_create": (C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    "wbrt_comm_destroy": (C.c_int, [_P]),
    "wbrt_comm_size": (C.c_int32, [_P]),
    "wbrt_comm_allreduce_sum": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "wbrt_comm_reduce_sum": (C.c_int, [_P, _P, C.c_size_t, C.c_int32, _P]),
    "wbrt_render_sharded_frame": (C.c_int, [_P, _P, C.c_int32, C.POINTER(Camera), C.c_int32, _P, C.c_uint64, _P]),
    "wbrt_set_mesh_mode": (C.c_int, [_P, C.c_int]),
    "wbrt_set_schedule_reuse": (C.c_int, [_P, C.c_int]),
    "wbrt_set_arith_mode": (C.c_int, [_P, C.c_int]),
    "wbrt_set_timing": (C.c_int, [_P, C.c_int]),
    "wbrt_get_timing": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int]),
    "wbrt_device_info": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "wbrt_calculate_camera": (None, [C.POINTER(Camera), Vec3, Vec3, Vec3, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "wbrt_image_height": (C.c_int32, [C.c_int32, C.c_double]),
    "wbrt_ideal_block_size": (Vec3, [C.c_int32, C.c_int32, C.c_int32]),
    "wbrt_make_seeds": (None, [C.c_uint64, C.c_uint32, _P, C.c_size_t]),
    "wbrt_tonemap_host": (None, [_P, C.c_float, C.c_int32, C.c_int32, _P]),
    "wbrt_write_ppm": (C.c_int, [C.c_char_p, _P, C.c_int32, C.c_int32]),
    "wbrt_write_pfm": (C.c_int, [C.c_char_p, _P, C.c_int32, C.c_int32]),
    "wbrt_write_bmp": (C.c_int, [C.c_char_p, _P, C.c_int32, C.c_int32]),
    "wbrt_write_exr": (C.c_int, [C.c_char_p, _P, C.c_int32, C.c_int32]),
    "wbrt_read_mesh": (C.c_int, [C.c_char_p, Vec3, Vec3, Material, C.c_int32, _P, C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.POINTER(Vec3), C.POINTER(Vec3)]),
    "wbrt_split_mesh_boxes": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P, C.c_int32]),
    "wbrt_sky_tonemap": (None, [_P, C.c_size_t, _P]),
    "wbrt_load_hdr": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_float))]),
    "wbrt_free": (None, [_P]),
    "wbrt_scene_readme": (C.c_int32, [_P, C.c_int32]),
    "wbrt_scene_main_cpp": (C.c_int32, [_P, C.c_int32]),
    "wbrt_camera_main_cpp": (None, [C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32]),
    "wbrt_camera_readme": (None, [C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32]),
    "wbrt_sky_gradient": (None, [_P, C.c_int32, C.c_int32]),
    "wbrt_scene_stress_spheres": (C.c_int32, [_P, C.c_int32, C.c_uint64]),
    "wbrt_scene_stress_mesh": (C.c_int32, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64]),
}
EXPORTED_SYMBOLS = sorted(_SIGNATURES)


def lib():
    """The loaded C-ABI library.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: build it first (python -c 'import __graft_entry__ as g; g.build()'); "
                "there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: PyTorch's wheel bundles its own libamdhip64.so.7 and a
        # second, separately initialised runtime in the same process finds no GPU.  Importing
        # torch first makes the loader resolve this library's libamdhip64.so.7 dependency to
        # the copy torch already loaded.  (A C++ host without torch uses /opt/rocm's.)
        if os.environ.get("WBRT_SKIP_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        l.wbrt_math_probe_.restype = C.c_int
        l.wbrt_math_probe_.argtypes = [_P, C.c_int, _P, _P, _P, C.c_int32]
        l.wbrt_tune_.restype = C.c_int
        l.wbrt_tune_.argtypes = [_P, C.c_char_p, C.c_int]
        l.wbrt_buffer_bytes_.restype = C.c_longlong
        l.wbrt_buffer_bytes_.argtypes = [_P, C.c_char_p]
        l.wbrt_rccl_probe_.restype = C.c_int
        l.wbrt_rccl_probe_.argtypes = [C.c_char_p]
        _lib = l
    return _lib


def _check(rc):
    if rc != 0:
        raise WbrtError(rc, lib().wbrt_last_error().decode("utf-8", "replace"))


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.cast(a, C.c_void_p)


def vec3(x, y, z):
    return Vec3(float(x), float(y), float(z))


def material(albedo, fuzz=0.0, ior=0.0, type=0):
    return Material(vec3(*albedo), float(fuzz), float(ior), int(type))


# ----------------------------------------------------------------- host helpers
def calculate_camera(look_from, look_at, v_up, v_fov, aspect, focus, aperture, width, height, spp, depth):
    cam = Camera()
    lib().wbrt_calculate_camera(C.byref(cam), vec3(*look_from), vec3(*look_at), vec3(*v_up), v_fov, aspect, focus,
                                aperture, width, height, spp, depth)
    return cam


def image_height(width, aspect=1.7777777):
    return lib().wbrt_image_height(width, aspect)


def ideal_block_size(width, height, target_blocks):
    v = lib().wbrt_ideal_block_size(width, height, target_blocks)
    return int(v.x), int(v.y)


def set_blocks(cam, target_blocks):
    """Tiles the frame like main.cpp:65,92: camera.blockSize = GetIdealBlockSize(pixels, target)."""
    bw, bh = ideal_block_size(cam.width, cam.height, target_blocks)
    cam.blockSize = vec3(bw, bh, 0)
    return bw, bh


def make_seeds(n, base_seed=0x5EED, shard=0):
    out = np.empty(n, dtype=np.uint32)
    lib().wbrt_make_seeds(base_seed, shard, _ptr(out), n)
    return out


def camera_readme(width=1280, spp=250, depth=50):
    cam = Camera()
    lib().wbrt_camera_readme(C.byref(cam), width, spp, depth)
    return cam


def camera_main_cpp(width=1920, spp=100, depth=50):
    cam = Camera()
    lib().wbrt_camera_main_cpp(C.byref(cam), width, spp, depth)
    return cam


def sky_gradient(width=64, height=32):
    out = np.empty((height, width, 3), dtype=np.float32)
    lib().wbrt_sky_gradient(_ptr(out), width, height)
    return out


def tonemap_host(image, scale=1.0):
    h, w, _ = image.shape
    img = np.ascontiguousarray(image, dtype=np.float32)
    out = np.empty((h, w, 3), dtype=np.uint8)
    lib().wbrt_tonemap_host(_ptr(img), scale, w, h, _ptr(out))
    return out


def write_ppm(path, rgb8):
    h, w, _ = rgb8.shape
    _check(lib().wbrt_write_ppm(path.encode(), _ptr(np.ascontiguousarray(rgb8)), w, h))


def write_pfm(path, image):
    h, w, _ = image.shape
    _check(lib().wbrt_write_pfm(path.encode(), _ptr(np.ascontiguousarray(image, dtype=np.float32)), w, h))


def write_exr(path, image):
    h, w, _ = image.shape
    _check(lib().wbrt_write_exr(path.encode(), _ptr(np.ascontiguousarray(image, dtype=np.float32)), w, h))


def write_bmp(path, rgb8):
    h, w, _ = rgb8.shape
    _check(lib().wbrt_write_bmp(path.encode(), _ptr(np.ascontiguousarray(rgb8)), w, h))


def read_mesh(path, scale, translate, mat, bounding_box_id=0):
    """Returns (Triangle array, box_min Vec3, box_max Vec3, vertex_count); utility.cpp:63-170."""
    nt, nv = C.c_int32(0), C.c_int32(0)
    _check(lib().wbrt_read_mesh(path.encode(), vec3(*scale), vec3(*translate), mat, bounding_box_id, None,
                                C.byref(nt), C.byref(nv), None, None))
    tris = (Triangle * max(nt.value, 1))()
    mn, mx = Vec3(), Vec3()
    cap = C.c_int32(nt.value)
    _check(lib().wbrt_read_mesh(path.encode(), vec3(*scale), vec3(*translate), mat, bounding_box_id, tris,
                                C.byref(cap), C.byref(nv), C.byref(mn), C.byref(mx)))
    return (Triangle * cap.value).from_buffer(tris) if cap.value else (Triangle * 0)(), mn, mx, nv.value


def split_mesh_boxes(triangles, target_boxes, first_box_id=0, pad=1e-4, flt_min_quirk=False):
    """Cuts one mesh (a ctypes Triangle array, modified in place: boundingBoxId) into contiguous runs with a bounding box
    each (wbrt_split_mesh_boxes); returns the BoundingBox array."""
    n = min(target_boxes, len(triangles))
    boxes = (BoundingBox * max(n, 1))()
    got = lib().wbrt_split_mesh_boxes(triangles, len(triangles), first_box_id, target_boxes, pad, 1 if flt_min_quirk else 0, boxes, n)
    if got < 0:
        raise ValueError("wbrt_split_mesh_boxes failed")
    return (BoundingBox * got).from_buffer_copy(bytes(boxes)[: got * C.sizeof(BoundingBox)])


def sky_tonemap(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.empty_like(rgb)
    lib().wbrt_sky_tonemap(_ptr(rgb), rgb.size // 3, _ptr(out))
    return out


def load_hdr(path):
    w, h = C.c_int32(0), C.c_int32(0)
    p = C.POINTER(C.c_float)()
    _check(lib().wbrt_load_hdr(path.encode(), C.byref(w), C.byref(h), C.byref(p)))
    try:
        arr = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        lib().wbrt_free(p)
    return arr


class Scene:
    """Host-side scene: the reference's arrays plus the (already tone-mapped) sky image."""

    def __init__(self, spheres=None, triangles=None, boxes=None, sky=None):
        self.spheres = spheres if spheres is not None else (Sphere * 0)()
        self.triangles = triangles if triangles is not None else (Triangle * 0)()
        self.boxes = boxes if boxes is not None else (BoundingBox * 0)()
        self.sky = np.ascontiguousarray(sky if sky is not None else sky_gradient(), dtype=np.float32)

    @property
    def S(self):
        return len(self.spheres)

    @property
    def T(self):
        return len(self.triangles)

    @property
    def B(self):
        return len(self.boxes)


def scene_readme():
    """README.md:32-38 five-sphere scene + synthetic gradient sky."""
    s = (Sphere * 5)()
    assert lib().wbrt_scene_readme(s, 5) == 5
    return Scene(spheres=s)


def scene_main_cpp(mesh_path=None):
    """main.cpp:98-129: eight spheres (+ the OBJ mesh in one bounding box when a path is given)."""
    s = (Sphere * 8)()
    assert lib().wbrt_scene_main_cpp(s, 8) == 8
    if mesh_path is None:
        return Scene(spheres=s)
    tris, mn, mx, _ = read_mesh(mesh_path, (0.15, 0.15, 0.15), (0.0, -0.5, -1.0),
                                material((0.8, 0.6, 0.2), 0.5, 0.0, METAL), 0)
    boxes = (BoundingBox * 1)()
    boxes[0] = BoundingBox(0, mn, mx)
    return Scene(spheres=s, triangles=tris, boxes=boxes)


def triangles_from_soup(soup, scale, translate, mat, box_id=0):
    """Triangle array + bounding box from a raw [T,9] float32 vertex soup, with the reference loader's
    arithmetic: vertex = v*scale + translate in float (utility.cpp:115-119) and the box
    maximum initialised to FLT_MIN (main.cpp:114-115)."""
    s = np.asarray(scale, dtype=np.float32)
    t = np.asarray(translate, dtype=np.float32)
    v = np.asarray(soup).reshape(-1, 3, 3).astype(np.float32)
    w = ((v * s[None, None, :]).astype(np.float32) + t[None, None, :]).astype(np.float32)
    n = w.shape[0]
    tris = (Triangle * n)()
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


def scene_main_cpp_from_soup(soup):
    """main.cpp:98-129 scene (eight spheres + one mesh in one bounding box) from an already parsed vertex soup
    ([T,9] float32, untransformed OBJ coordinates) instead of an OBJ file."""
    base = scene_main_cpp()
    tris, mn, mx = triangles_from_soup(soup, (0.15, 0.15, 0.15), (0.0, -0.5, -1.0),
                                       material((0.8, 0.6, 0.2), 0.5, 0.0, METAL), 0)
    boxes = (BoundingBox * 1)()
    boxes[0] = BoundingBox(0, vec3(*mn), vec3(*mx))
    return Scene(spheres=base.spheres, triangles=tris, boxes=boxes)


def scene_stress(n_spheres=500, nx=224, nz=224, boxes_x=16, boxes_z=16, base_seed=2024):
    """Divergence-stress scene: random Lambertian/metal spheres + a height-field mesh in many boxes."""
    s = (Sphere * n_spheres)()
    assert lib().wbrt_scene_stress_spheres(s, n_spheres, base_seed) == n_spheres
    nt = 2 * nx * nz
    tris = (Triangle * nt)()
    boxes = (BoundingBox * (boxes_x * boxes_z))()
    got = lib().wbrt_scene_stress_mesh(tris, boxes, nx, nz, boxes_x, boxes_z, base_seed)
    assert got == nt
    return Scene(spheres=s, triangles=tris, boxes=boxes)


def camera_stress(width=3840, spp=500, depth=50):
    return calculate_camera((0.0, 2.5, 4.0), (0.0, 0.5, -8.0), (0.0, 1.0, 0.0), 50.0, 1.7777777, 10.0, 0.0,
                            width, image_height(width), spp, depth)


# -------------------------------------------------------------------- device path
class Context:
    """One GPU's render context (wbrt_create ... wbrt_destroy)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib().wbrt_create(device, C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            lib().wbrt_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, scene):
        _check(lib().wbrt_upload_scene(self._h, _ptr(scene.spheres) if scene.S else None, scene.S,
                                       _ptr(scene.triangles) if scene.T else None, scene.T,
                                       _ptr(scene.boxes) if scene.B else None, scene.B))
        h, w, _ = scene.sky.shape
        _check(lib().wbrt_upload_sky(self._h, _ptr(scene.sky), w, h))

    def upload_seeds(self, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        _check(lib().wbrt_upload_seeds(self._h, _ptr(seeds), seeds.size))

    def render_block(self, cam):
        bw, bh = int(cam.blockSize.x), int(cam.blockSize.y)
        out = np.empty((bh, bw, 3), dtype=np.float32)
        _check(lib().wbrt_render_block(self._h, C.byref(cam), _ptr(out)))
        return out

    def render_blocks(self, cam, on_block):
        """wbrt_render_blocks: on_block(bx, by, block[h, w, 3]) per block in the reference's order; a true return stops."""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p)

        def trampoline(_user, bx, by, w, h, ptr):
            blk = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(h, w, 3)).copy()
            return 1 if on_block(bx, by, blk) else 0
        cb = proto(trampoline)
        _check(lib().wbrt_render_blocks(self._h, C.byref(cam), C.cast(cb, C.c_void_p), None))

    def render_frame(self, cam, out_mode=OUT_MEAN, into=None):
        out = into if into is not None else np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        _check(lib().wbrt_render_frame(self._h, C.byref(cam), out_mode, _ptr(out)))
        return out

    def render_frame_device(self, cam, device_ptr, out_mode=OUT_MEAN, stream=None):
        _check(lib().wbrt_render_frame_device(self._h, C.byref(cam), out_mode, C.c_void_p(device_ptr),
                                              C.c_void_p(stream) if stream else None))

    def render_frame_device_noalloc(self, cam, device_ptr, out_mode=OUT_MEAN, stream=None):
        """wbrt_render_frame_device without any allocation (schedule arrays in the library's static device array)."""
        _check(lib().wbrt_render_frame_device_noalloc(self._h, C.byref(cam), out_mode, C.c_void_p(device_ptr),
                                                      C.c_void_p(stream) if stream else None))

    def tune(self, key, value):
        """Private test/experiment hook (wbrt_tune_): thresholds that never change a result."""
        _check(lib().wbrt_tune_(self._h, key.encode(), int(value)))

    def buffer_bytes(self, name):
        """Test hook: bytes the context has allocated for one of its schedule buffers (costs, tile_costs, order, hist)."""
        v = lib().wbrt_buffer_bytes_(self._h, name.encode())
        if v < 0:
            raise ValueError(name)
        return int(v)

    def make_seeds_device(self, n, base_seed=0x5EED, shard=0, stream=None):
        """Generates the seed buffer of (base_seed, shard) on the device (== make_seeds + upload_seeds, no host copy)."""
        _check(lib().wbrt_make_seeds_device(self._h, base_seed, shard, n, C.c_void_p(stream) if stream else None))

    def render_shards(self, cam, device_ptr, num_shards, base_seed=0x5EED, rank=0, world=1, shard_spp=None, stream=None):
        """This rank's seed shards of one frame summed into device_ptr (zeroed first); asynchronous on `stream`."""
        arr = None if shard_spp is None else np.ascontiguousarray(shard_spp, dtype=np.int32)
        if arr is not None and arr.size != num_shards:
            raise ValueError("shard_spp needs one entry per shard")
        _check(lib().wbrt_render_shards(self._h, C.byref(cam), num_shards, _ptr(arr), base_seed, rank, world,
                                        C.c_void_p(device_ptr), C.c_void_p(stream) if stream else None))

    def primary_ids(self, cam):
        out = np.empty((cam.height, cam.width), dtype=np.int32)
        _check(lib().wbrt_primary_ids_frame(self._h, C.byref(cam), _ptr(out)))
        return out

    def count_events(self, cam):
        out = np.zeros(NUM_COUNTERS, dtype=np.uint64)
        _check(lib().wbrt_count_events_frame(self._h, C.byref(cam), _ptr(out)))
        return dict(zip(COUNTER_NAMES, (int(v) for v in out)))

    def tonemap_device(self, image_ptr, scale, width, height, rgb8_ptr, stream=None):
        _check(lib().wbrt_tonemap_device(self._h, C.c_void_p(image_ptr), scale, width, height, C.c_void_p(rgb8_ptr),
                                         C.c_void_p(stream) if stream else None))

    def synchronize(self):
        _check(lib().wbrt_synchronize(self._h))

    def set_mesh_mode(self, mode):
        """MESH_LOOP, MESH_BVH (default, proven equal to the loop) or MESH_BVH_CAPPED; see include/ocl_rt_amd.h."""
        _check(lib().wbrt_set_mesh_mode(self._h, int(mode)))

    def set_arith_mode(self, mode):
        """ARITH_STRICT (default, bit-exact) or ARITH_TOLERANT (fused/approximate arithmetic); see include/ocl_rt_amd.h."""
        _check(lib().wbrt_set_arith_mode(self._h, int(mode)))

    def set_schedule_reuse(self, on=True):
        """Schedule a launch from the costs its predecessor recorded (default) or always from a pilot launch."""
        _check(lib().wbrt_set_schedule_reuse(self._h, 1 if on else 0))

    def set_timing(self, on=True):
        _check(lib().wbrt_set_timing(self._h, 1 if on else 0))

    def get_timing(self, reset=True):
        ms, n = C.c_double(0), C.c_int32(0)
        _check(lib().wbrt_get_timing(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def device_info(self):
        buf = C.create_string_buffer(512)
        _check(lib().wbrt_device_info(self._h, buf, 512))
        return buf.value.decode()

    def math_probe(self, op, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
        out = np.empty_like(a)
        _check(lib().wbrt_math_probe_(self._h, op, _ptr(a), _ptr(b), _ptr(out), a.size))
        return out


def shards_of_rank(num_shards, rank, world):
    out = np.empty(max(num_shards, 1), dtype=np.int32)
    n = lib().wbrt_shards_of_rank(num_shards, rank, world, _ptr(out))
    if n < 0:
        raise ValueError("bad shard split")
    return [int(v) for v in out[:n]]


def split_spp(total_spp, num_shards):
    out = np.empty(num_shards, dtype=np.int32)
    if lib().wbrt_split_spp(total_spp, num_shards, _ptr(out)) != 0:
        raise ValueError("bad shard split")
    return [int(v) for v in out]


def render_sharded_frame(ctxs, cam, num_shards, base_seed=0x5EED, shard_spp=None, comm=None):
    """wbrt_render_sharded_frame: one frame of num_shards seed shards over the given contexts (one per GPU); returns
    the clamped per-sample sum over all shards as a host array."""
    out = np.empty((cam.height, cam.width, 3), dtype=np.float32)
    arr = None if shard_spp is None else np.ascontiguousarray(shard_spp, dtype=np.int32)
    hs = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(lib().wbrt_render_sharded_frame(comm._h if comm is not None else None, hs, len(ctxs), C.byref(cam), num_shards,
                                           _ptr(arr), base_seed, _ptr(out)))
    return out


def rccl_probe(library_name):
    """Test hook: what wbrt_comm_create does when RCCL is to be loaded from `library_name` (raises WbrtError if it cannot)."""
    _check(lib().wbrt_rccl_probe_(library_name.encode()))


class Comm:
    """One process, several GPUs: RCCL communicator of the C ABI (wbrt_comm_*)."""

    def __init__(self, devices):
        arr = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        _check(lib().wbrt_comm_create(_ptr(arr), arr.size, C.byref(h)))
        self._h = h
        self.size = arr.size

    def allreduce_sum(self, device_ptrs, count, streams=None):
        bufs = (C.c_void_p * self.size)(*[C.c_void_p(p) for p in device_ptrs])
        st = None if streams is None else (C.c_void_p * self.size)(*[C.c_void_p(x) if x else None for x in streams])
        _check(lib().wbrt_comm_allreduce_sum(self._h, bufs, count, st))

    def reduce_sum(self, device_ptrs, count, root=0, streams=None):
        bufs = (C.c_void_p * self.size)(*[C.c_void_p(p) for p in device_ptrs])
        st = None if streams is None else (C.c_void_p * self.size)(*[C.c_void_p(x) if x else None for x in streams])
        _check(lib().wbrt_comm_reduce_sum(self._h, bufs, count, root, st))

    def close(self):
        if self._h:
            lib().wbrt_comm_destroy(self._h)
            self._h = None


# ------------------------------------------------------------ FLOP accounting
def algorithmic_flops(c):
    """Algorithmic FP32 operations for a set of event counters.

    Counting rule: each FP32 add/sub/mul/div/sqrt/min/max of the reference's source counts 1;
    compares, selects, abs, negation, integer/RNG work, conversions and the internals of
    atan2/asin/pow count 0.  Per-function costs (reference lines in brackets):
      u,v jitter 10 [gpu_kernel.cl:626-627]; camera ray 18 [:575-576] (+10 disk mapping
      [:570-572]); disk try 11 [:248-254]; sphere test 23 (+3 root, +2 second root, +17
      accepted) [:457-483]; bounding box 23 [:536-554]; triangle 20/30/46/52 by exit
      (+29 accepted) [:491-525]; unit-sphere try 14 [:187-193]; Lambertian 18 [:198-201,400];
      metal 32 (+6 fuzz) [:417-422]; dielectric 47 (+1 draw) + 12 reflect or 29 refract
      (priced at 20) [:425-451]; emissive 3; miss 18 [:237-239,347]; accumulate 3 [:644].
    """
    g = lambda k: int(c.get(k, 0))
    tri_full = g("tri_reach_t")
    f = 0
    f += g("paths") * (10 + 18 + 3)
    f += g("disk_tries") * 11 + (g("paths") * 10 if g("disk_tries") else 0)
    f += g("sphere_tests") * 23 + g("sphere_disc_ge0") * 3 + g("sphere_second_root") * 2 + g("sphere_accept") * 17
    f += g("bbox_tests") * 23
    f += g("tri_exit_parallel") * 20 + g("tri_exit_u") * 30 + g("tri_exit_v") * 46 + tri_full * 52 + g("tri_accept") * 29
    f += g("sphere_tries") * 14
    f += g("lambert") * 18 + g("metal") * 32 + g("metal_fuzz") * 6
    f += g("dielectric") * (47 + 20) + g("dielectric_draw") * 1
    f += g("emissive") * 3
    f += g("miss") * 18
    return f

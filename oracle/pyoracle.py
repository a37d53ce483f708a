"""TEST INFRASTRUCTURE ONLY: ctypes bindings of the CPU checkers.

  liboracle.so        this repository's CPU restatement (pt_oracle.c)
  _ref/libwbref.so    the reference's own kernel file compiled for the host (built by
                      oracle/Makefile from /root/reference when that tree is present)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It takes the product's ctypes PODs (opencl-ray-tracer_amd/__init__.py) because
the layouts are byte-identical on all three sides.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libwbref.so")
REF_HOST_SO = os.path.join(_HERE, "_ref", "libwbref_host.so")

MATH_LIBM, MATH_PORTABLE = 0, 1
OUT_MEAN, OUT_SUM = 0, 1
NUM_COUNTERS = 26


def build():
    """Builds liboracle.so, and _ref/libwbref.so when the reference tree is present."""
    subprocess.run(["make", "-C", _HERE, "oracle", "ref", "ref_host"], check=True)


class _SceneC(C.Structure):
    _fields_ = [("spheres", C.c_void_p), ("S", C.c_int),
                ("tris", C.c_void_p), ("T", C.c_int),
                ("bbs", C.c_void_p), ("B", C.c_int),
                ("sky", C.c_void_p), ("skyw", C.c_int), ("skyh", C.c_int)]


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.cast(a, C.c_void_p)


def guarded_sky(sky):
    """Sky texels followed by width+1 zero guard texels (same layout as the device copy)."""
    h, w, _ = sky.shape
    buf = np.zeros(((h * w) + w + 1, 3), dtype=np.float32)
    buf[: h * w] = sky.reshape(-1, 3)
    return buf


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError("%s missing: run `make -C oracle oracle`" % ORACLE_SO)
        l = C.CDLL(ORACLE_SO)
        l.wbo_render_items.restype = C.c_int
        l.wbo_render_frame.restype = C.c_int
        l.wbo_primary_ids_frame.restype = C.c_int
        l.wbo_next_seed.restype = C.c_uint64
        l.wbo_next_seed.argtypes = [C.c_uint64]
        l.wbo_rand_float.restype = C.c_float
        l.wbo_atan2f_portable.restype = C.c_float
        l.wbo_atan2f_portable.argtypes = [C.c_float, C.c_float]
        l.wbo_asinf_portable.restype = C.c_float
        l.wbo_asinf_portable.argtypes = [C.c_float]
        l.wbo_pow5f_portable.restype = C.c_float
        l.wbo_pow5f_portable.argtypes = [C.c_float]
        _oracle = l
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("%s missing (needs /root/reference at build time)" % REF_SO)
        l = C.CDLL(REF_SO)
        l.wbref_run_items.restype = C.c_int
        l.wbref_next_seed.restype = C.c_uint64
        l.wbref_next_seed.argtypes = [C.c_uint64]
        l.wbref_rand_float.restype = C.c_float
        _ref = l
    return _ref


def ref_host_lib():
    """The reference's host-side functions (oracle/_ref/libwbref_host.so)."""
    if not os.path.exists(REF_HOST_SO):
        raise RuntimeError("%s missing (needs /root/reference at build time)" % REF_HOST_SO)
    return C.CDLL(REF_HOST_SO)


class Oracle:
    """Runs a scene through liboracle.so."""

    def __init__(self, scene):
        self.scene = scene
        self._sky = guarded_sky(scene.sky)
        h, w, _ = scene.sky.shape
        self._c = _SceneC(_ptr(scene.spheres) if scene.S else None, scene.S,
                          _ptr(scene.triangles) if scene.T else None, scene.T,
                          _ptr(scene.boxes) if scene.B else None, scene.B,
                          _ptr(self._sky), w, h)

    def render_frame(self, cam, seeds, out_mode=OUT_MEAN, math_mode=MATH_PORTABLE, threads=None, counters=False):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        img = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        ctr = np.zeros(NUM_COUNTERS, dtype=np.uint64) if counters else None
        rc = oracle_lib().wbo_render_frame(C.byref(self._c), C.byref(cam), _ptr(seeds), _ptr(img),
                                           C.c_int(out_mode), C.c_int(math_mode), _ptr(ctr),
                                           C.c_int(threads or os.cpu_count() or 1))
        if rc:
            raise RuntimeError("wbo_render_frame failed: %d" % rc)
        return (img, ctr) if counters else img

    def render_block(self, cam, seeds, out_mode=OUT_MEAN, math_mode=MATH_PORTABLE, threads=None):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        bw, bh = int(cam.blockSize.x), int(cam.blockSize.y)
        img = np.zeros((bh, bw, 3), dtype=np.float32)
        rc = oracle_lib().wbo_render_items(C.byref(self._c), C.byref(cam), _ptr(seeds), C.c_int(0), C.c_int(bw * bh),
                                           _ptr(img), C.c_int(out_mode), C.c_int(math_mode), None,
                                           C.c_int(threads or os.cpu_count() or 1))
        if rc:
            raise RuntimeError("wbo_render_items failed: %d" % rc)
        return img

    def primary_ids(self, cam, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        ids = np.zeros((cam.height, cam.width), dtype=np.int32)
        rc = oracle_lib().wbo_primary_ids_frame(C.byref(self._c), C.byref(cam), _ptr(seeds), _ptr(ids))
        if rc:
            raise RuntimeError("wbo_primary_ids_frame failed: %d" % rc)
        return ids


class Reference:
    """Runs a scene through the reference's own kernel (oracle/_ref/libwbref.so)."""

    def __init__(self, scene):
        self.scene = scene
        self._sky = guarded_sky(scene.sky)

    def _run(self, cam, seeds, rgb, spheres, tris, sky, count, threads):
        sc = self.scene
        h, w, _ = sc.sky.shape
        rc = ref_lib().wbref_run_items(_ptr(rgb), _ptr(sky), C.c_int(w), C.c_int(h), _ptr(seeds), C.byref(cam),
                                       _ptr(spheres) if sc.S else None, C.c_int(sc.S),
                                       _ptr(tris) if sc.T else None, C.c_int(sc.T),
                                       _ptr(sc.boxes) if sc.B else None, C.c_int(sc.B),
                                       C.c_int(0), C.c_int(count), C.c_int(threads or os.cpu_count() or 1))
        if rc:
            raise RuntimeError("wbref_run_items failed: %d" % rc)

    def render_block(self, cam, seeds, threads=None):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        bw, bh = int(cam.blockSize.x), int(cam.blockSize.y)
        rgb = np.zeros((bh, bw, 3), dtype=np.float32)
        self._run(cam, seeds, rgb, self.scene.spheres, self.scene.triangles, self._sky, bw * bh, threads)
        return rgb

    def render_frame(self, cam, seeds, threads=None, _probe=None):
        """Block loop of main.cpp:229-260 around the reference kernel."""
        bw, bh = int(cam.blockSize.x), int(cam.blockSize.y)
        nbx, nby = cam.width // bw, cam.height // bh
        img = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        for by in range(nby - 1, -1, -1):
            for bx in range(nbx):
                c = cam.copy()
                c.blockOffset.x, c.blockOffset.y, c.blockOffset.z = float(bx), float(by), 0.0
                if _probe is None:
                    blk = self.render_block(c, seeds, threads)
                else:
                    blk = np.zeros((bh, bw, 3), dtype=np.float32)
                    self._run(c, seeds, blk, _probe[0], _probe[1], _probe[2], bw * bh, threads)
                img[by * bh:(by + 1) * bh, bx * bw:(bx + 1) * bw] = blk
        return img

    def primary_ids(self, cam, seeds, threads=None):
        """Primary-hit ids from the UNMODIFIED reference kernel.

        The reference's hit record has no primitive id, so the scene is re-coloured for a
        probe render: every material becomes emissive (type 3) with its index coded exactly in
        the albedo (x = low 12 bits / 4096, y = next 12 bits / 4096, z = 1), the sky is black,
        samplesPerPixel = 1.  An emissive hit returns throughput(1,1,1) * albedo
        (gpu_kernel.cl:326-329), the mean over one sample is exact, so the id is read back
        from the pixel; z == 0 means a miss.  Geometry, camera, seeds and the first two (or
        more, with defocus) random draws are exactly those of the real render.
        """
        from copy import deepcopy  # noqa: F401  (ctypes arrays are copied below by hand)
        sc = self.scene
        S, T = sc.S, sc.T

        def coded(i):
            return ((i & 0xFFF) / 4096.0, ((i >> 12) & 0xFFF) / 4096.0, 1.0)

        sph = type(sc.spheres)()
        C.memmove(sph, sc.spheres, C.sizeof(sph))
        for i in range(S):
            a = coded(i)
            sph[i].material.albedo.x, sph[i].material.albedo.y, sph[i].material.albedo.z = a
            sph[i].material.type = 3
        tri = type(sc.triangles)()
        C.memmove(tri, sc.triangles, C.sizeof(tri))
        if T:
            # vectorised recolouring of the triangle array
            arr = np.frombuffer(tri, dtype=np.float32).reshape(T, 16)
            idx = np.arange(S, S + T)
            arr[:, 9] = (idx & 0xFFF) / 4096.0
            arr[:, 10] = ((idx >> 12) & 0xFFF) / 4096.0
            arr[:, 11] = 1.0
            arr.view(np.int32)[:, 14] = 3
        black = np.zeros_like(self._sky)
        c = cam.copy()
        c.samplesPerPixel = 1
        img = self.render_frame(c, seeds, threads, _probe=(sph, tri, black))
        ids = (np.rint(img[..., 0] * 4096.0).astype(np.int64) | (np.rint(img[..., 1] * 4096.0).astype(np.int64) << 12))
        ids = np.where(img[..., 2] > 0.5, ids, -1)
        return ids.astype(np.int32)

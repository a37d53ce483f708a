"""Host-side checks of the triangle-hierarchy builder (no GPU): every triangle of a box sits in
exactly one leaf below that box's root, every child box contains its triangles, the stored
E bounds |e0||e1| from above, and the depth never exceeds the kernel's stack."""
import ctypes as C

import numpy as np
import pytest

from util import pkg, golden, triangles_from_soup

STACK = 24  # WBRT_BVH_STACK in csrc/rt_device.h


def _build(geom, ranges, split_by_e=0):
    L = pkg.lib()
    L.wbrt_bvh_build_.restype = C.c_int
    T, B = geom.shape[0], ranges.shape[0]
    nodes = np.zeros((max(2 * T, 1), 16), np.float32)
    tri_index = np.zeros(max(T, 1), np.int32)
    roots = np.zeros((B, 8), np.float32)
    depth = C.c_int(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = L.wbrt_bvh_build_(p(geom), p(ranges), B, p(nodes), nodes.shape[0], p(tri_index), tri_index.shape[0], p(roots), C.byref(depth), split_by_e)
    assert n >= 0
    return nodes[:n], tri_index, roots, depth.value


def _geom_from_soup(soup):
    v = soup.reshape(-1, 3, 3).astype(np.float32)
    g = np.zeros((v.shape[0], 12), np.float32)
    g[:, 0:3] = v[:, 0]
    g[:, 3:6] = v[:, 1] - v[:, 0]
    g[:, 6:9] = v[:, 2] - v[:, 0]
    return g


def _check_tree(geom, ranges, nodes, tri_index, roots, depth):
    assert depth <= STACK
    p0 = geom[:, 0:3].astype(np.float64)
    verts = np.stack([p0, p0 + geom[:, 3:6], p0 + geom[:, 6:9]], axis=1)  # [T,3,3]
    e = np.linalg.norm(geom[:, 3:6].astype(np.float64), axis=1) * np.linalg.norm(geom[:, 6:9].astype(np.float64), axis=1)
    refs = nodes[:, 12:14].view(np.int32)
    seen = np.zeros(geom.shape[0], np.int32)

    def walk(ref, mn, mx, emax, level):
        assert level <= STACK
        if ref < 0:
            leaf = ~ref
            first, count = leaf >> 4, (leaf & 15) + 1
            idx = tri_index[first:first + count]
            finite = np.isfinite(verts[idx]).all(axis=(1, 2))
            assert (verts[idx][finite] >= mn - 0).all() and (verts[idx][finite] <= mx + 0).all()
            assert (e[idx][np.isfinite(e[idx])] <= emax).all()
            seen[idx] += 1
            return idx
        n = nodes[ref]
        left = walk(int(refs[ref, 0]), n[0:3].astype(np.float64), n[3:6].astype(np.float64), float(n[14]), level + 1)
        right = walk(int(refs[ref, 1]), n[6:9].astype(np.float64), n[9:12].astype(np.float64), float(n[15]), level + 1)
        both = np.concatenate([left, right])
        finite = np.isfinite(verts[both]).all(axis=(1, 2))
        assert (verts[both][finite] >= mn).all() and (verts[both][finite] <= mx).all()
        return both

    for b, (first, last) in enumerate(ranges):
        if last <= first:
            assert roots[b, 7] < 0
            continue
        got = walk(int(roots[b, 3:4].view(np.int32)[0]), roots[b, 0:3].astype(np.float64), roots[b, 4:7].astype(np.float64),
                   float(roots[b, 7]), 0)
        assert sorted(got.tolist()) == list(range(first, last))
    inside = np.zeros(geom.shape[0], bool)
    for first, last in ranges:
        inside[first:last] = True
    assert (seen[inside] == 1).all() and (seen[~inside] == 0).all()


@pytest.mark.parametrize("mesh", ["f1_car", "burger", "random", "coincident", "line", "tiny"])
def test_builder_invariants(mesh):
    rs = np.random.RandomState(5)
    z = golden("meshes.npz")
    if mesh in z.files:
        geom = _geom_from_soup(z[mesh])
    elif mesh == "random":
        c = rs.uniform(-5, 5, (3000, 1, 3))
        geom = _geom_from_soup((c + rs.uniform(-0.3, 0.3, (3000, 3, 3)) * rs.uniform(0.01, 3, (3000, 1, 1))).reshape(-1, 9))
    elif mesh == "coincident":   # identical centroids: only the median fallback can split them
        geom = _geom_from_soup(np.tile(rs.uniform(-1, 1, (1, 9)), (1000, 1)))
    elif mesh == "line":         # sorted along one axis with exponentially growing gaps: lopsided SAH splits
        x = np.cumsum(1.5 ** np.arange(60))[:, None, None] * np.array([1.0, 0, 0])
        geom = _geom_from_soup((x + rs.uniform(-0.1, 0.1, (60, 3, 3))).reshape(-1, 9))
    else:
        geom = _geom_from_soup(rs.uniform(-1, 1, (3, 9)))
    T = geom.shape[0]
    if mesh == "random":
        ranges = np.array([[0, 1000], [1000, 1000], [1000, T]], np.int32)  # one empty box
    else:
        ranges = np.array([[0, T]], np.int32)
    for split_by_e in (0, 1):
        nodes, tri_index, roots, depth = _build(geom, ranges, split_by_e)
        _check_tree(geom, ranges, nodes, tri_index, roots, depth)


def test_builder_handles_non_finite_vertices():
    rs = np.random.RandomState(6)
    soup = rs.uniform(-1, 1, (200, 9))
    soup[3, 4] = np.nan
    soup[7, 0] = np.inf
    soup[11, :] = np.nan
    geom = _geom_from_soup(soup)
    ranges = np.array([[0, 200]], np.int32)
    nodes, tri_index, roots, depth = _build(geom, ranges)
    assert depth <= STACK
    assert sorted(tri_index[:200].tolist()) == list(range(200))

"""CPU tests of the samples-per-pixel sharding logic with torch.distributed (gloo,
world_size 2).  The render call of each rank is served by the CPU oracle through a stand-in
context with the C ABI's method names, so the orchestration (shard assignment, device-side
accumulation, all-reduce, normalisation) is exercised exactly as on GPUs."""
import ctypes as C
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import pkg, orc, entry, bits

mg = importlib.import_module(entry.PKG_NAME + ".multi_gpu")


class OracleContext:
    """Same methods as pkg.Context, backed by the oracle, writing through raw pointers."""

    def __init__(self, scene):
        self.oracle = orc.Oracle(scene)
        self.seeds = None

    def upload_seeds(self, seeds):
        self.seeds = np.array(seeds, dtype=np.uint32)

    def synchronize(self):
        pass

    def render_frame_device(self, cam, ptr, out_mode, stream=None):
        n = cam.width * cam.height * 3
        dst = np.ctypeslib.as_array((C.c_float * n).from_address(ptr))
        s = self.oracle.render_frame(cam, self.seeds, out_mode=orc.OUT_SUM, threads=2).reshape(-1)
        if out_mode == pkg.OUT_ACCUMULATE:
            dst += s
        elif out_mode == pkg.OUT_SUM:
            dst[:] = s
        else:
            dst[:] = s / np.float32(cam.samplesPerPixel)

    def render_shards(self, cam, ptr, num_shards, base_seed=0x5EED, rank=0, world=1, shard_spp=None, stream=None):
        """wbrt_render_shards: zero, then += the clamped sum of every shard of this rank (host seeds stand in for the
        device generator, which test_gpu_parity checks against them)."""
        n = cam.width * cam.height * 3
        dst = np.ctypeslib.as_array((C.c_float * n).from_address(ptr))
        dst[:] = 0
        bw, bh = int(cam.blockSize.x), int(cam.blockSize.y)
        for v in pkg.shards_of_rank(num_shards, rank, world):
            c = cam.copy()
            if shard_spp is not None:
                c.samplesPerPixel = int(shard_spp[v])
            self.upload_seeds(pkg.make_seeds(bw * bh, base_seed, shard=v))
            self.render_frame_device(c, ptr, pkg.OUT_ACCUMULATE)

    def tonemap_device(self, ptr, scale, width, height, out_ptr, stream=None):
        n = width * height * 3
        src = np.ctypeslib.as_array((C.c_float * n).from_address(ptr)).reshape(height, width, 3)
        dst = np.ctypeslib.as_array((C.c_uint8 * n).from_address(out_ptr)).reshape(height, width, 3)
        dst[:] = pkg.tonemap_host(src, scale)


def _case():
    scene = pkg.scene_readme()
    cam = pkg.camera_readme(64, 3, 50)
    pkg.set_blocks(cam, 4)
    return scene, cam


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_shards, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scene, cam = _case()
        ctx = OracleContext(scene)
        frame = mg.FrameBuffers(cam.width, cam.height, torch.device("cpu"))
        n = mg.render_sharded_frame(ctx, cam, frame, num_shards, 77, rank, world,
                                    all_reduce=lambda t: dist.all_reduce(t))
        np.save(os.path.join(out_dir, "accum_%d.npy" % rank), frame.accum.numpy())
        np.save(os.path.join(out_dir, "rgb8_%d.npy" % rank), frame.rgb8.numpy())
        np.save(os.path.join(out_dir, "count_%d.npy" % rank), np.array([n]))
        # weak-scaling step used by bench.py: one shard per rank
        bw, bh = int(cam.blockSize.x), int(cam.blockSize.y)
        ctx.upload_seeds(pkg.make_seeds(bw * bh, 77, shard=rank))
        mg.render_step(ctx, cam, frame, total_spp=cam.samplesPerPixel * world,
                       all_reduce=lambda t: dist.all_reduce(t))
        np.save(os.path.join(out_dir, "weak_%d.npy" % rank), frame.accum.numpy())
    finally:
        dist.destroy_process_group()


def test_shard_assignment_partitions():
    for V in (1, 2, 5, 8, 13):
        for K in (1, 2, 4, 8):
            parts = [mg.shards_for_rank(V, r, K) for r in range(K)]
            assert sorted(sum(parts, [])) == list(range(V))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


@pytest.mark.parametrize("num_shards", [2, 5])
def test_two_rank_sharded_frame(tmp_path, num_shards):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), num_shards, str(tmp_path)), nprocs=world, join=True)
    scene, cam = _case()
    bw, bh = int(cam.blockSize.x), int(cam.blockSize.y)
    O = orc.Oracle(scene)
    shard = [O.render_frame(cam, pkg.make_seeds(bw * bh, 77, shard=v), out_mode=orc.OUT_SUM) for v in range(num_shards)]
    per_rank = []
    for r in range(world):
        acc = np.zeros_like(shard[0])
        for v in mg.shards_for_rank(num_shards, r, world):
            acc += shard[v]
        per_rank.append(acc)
    want = (per_rank[0] + per_rank[1]).reshape(-1)
    a0 = np.load(tmp_path / "accum_0.npy")
    a1 = np.load(tmp_path / "accum_1.npy")
    assert np.array_equal(bits(a0), bits(want)) and np.array_equal(bits(a1), bits(want))
    counts = [int(np.load(tmp_path / ("count_%d.npy" % r))[0]) for r in range(world)]
    assert sum(counts) == num_shards
    total = num_shards * cam.samplesPerPixel
    pic = pkg.tonemap_host(want.reshape(cam.height, cam.width, 3), 1.0 / total)
    assert np.array_equal(np.load(tmp_path / "rgb8_0.npy").reshape(pic.shape), pic)
    # the sharded frame is the mean of all V*spp clamped samples: same picture up to summation order
    mean_of_shards = sum(s.astype(np.float64) for s in shard) / total
    assert np.abs(want.reshape(mean_of_shards.shape) / total - mean_of_shards).max() <= 1e-6
    # weak step: sum of shard 0 and shard 1
    w0 = np.load(tmp_path / "weak_0.npy")
    assert np.array_equal(bits(w0), bits((shard[0] + shard[1]).reshape(-1)))


def test_single_rank_equals_reference_semantics():
    scene, cam = _case()
    ctx = OracleContext(scene)
    frame = mg.FrameBuffers(cam.width, cam.height, torch.device("cpu"))
    mg.render_sharded_frame(ctx, cam, frame, 1, 5, 0, 1)
    bw, bh = int(cam.blockSize.x), int(cam.blockSize.y)
    seeds = pkg.make_seeds(bw * bh, 5, shard=0)
    want = orc.Oracle(scene).render_frame(cam, seeds)  # the strict reference run (V = 1)
    got = frame.image().numpy() / np.float32(cam.samplesPerPixel)
    assert np.array_equal(bits(got), bits(want))


def test_uneven_shard_sample_counts():
    assert mg.split_spp(500, 8) == [63, 63, 63, 63, 62, 62, 62, 62]
    assert mg.split_spp(8, 4) == [2, 2, 2, 2]
    scene, cam = _case()
    cam.samplesPerPixel = 7  # split over 3 shards as 3, 2, 2
    spp = mg.split_spp(7, 3)
    ctx = OracleContext(scene)
    frame = mg.FrameBuffers(cam.width, cam.height, torch.device("cpu"))
    mg.render_sharded_frame(ctx, cam, frame, 3, 9, 0, 1, shard_spp=spp)
    bw, bh = int(cam.blockSize.x), int(cam.blockSize.y)
    O = orc.Oracle(scene)
    want = None
    for v in range(3):
        c = cam.copy()
        c.samplesPerPixel = spp[v]
        part = O.render_frame(c, pkg.make_seeds(bw * bh, 9, shard=v), out_mode=orc.OUT_SUM)
        want = part if want is None else want + part
    assert np.array_equal(bits(frame.accum.numpy()), bits(want.reshape(-1)))
    pic = pkg.tonemap_host(want.reshape(cam.height, cam.width, 3), 1.0 / 7)
    assert np.array_equal(frame.rgb8.numpy().reshape(pic.shape), pic)

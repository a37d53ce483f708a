"""Samples-per-pixel sharding across GPUs (one process per GPU, torch.distributed / RCCL).

The reference has one random stream per pixel and a data-dependent number of draws per
sample (gpu_kernel.cl:606,624-645), so a frame is split into V *seed shards* instead of
sample ranges: shard v is the reference kernel's semantics run with its own seed buffer
(make_seeds(base, shard=v)) and spp_per_shard samples, producing the UN-NORMALISED clamped
per-sample sum.  frame = sum_v shard_v / (V * spp_per_shard).  Rank r of K renders shards
{v : v % K == r}, accumulating them on the device; one all-reduce(sum) of the
width*height*3 float buffer per frame (or a reduce to rank 0, the only rank that needs the
picture) is the only exchange step; the ranks that hold the sum then normalise and tone-map.

torch is used for device memory, streams and the collective only; rendering goes through
the C ABI (Context).
"""
from . import OUT_SUM


def shards_for_rank(num_shards, rank, world):
    from . import shards_of_rank
    return shards_of_rank(num_shards, rank, world)


class FrameBuffers:
    """Device-resident accumulation buffer (float, row 0 = bottom) and 8-bit picture (row 0 = top)."""

    def __init__(self, width, height, device):
        import torch
        self.width, self.height = width, height
        self.accum = torch.zeros(height * width * 3, dtype=torch.float32, device=device)
        self.rgb8 = torch.zeros(height * width * 3, dtype=torch.uint8, device=device)

    def image(self):
        return self.accum.view(self.height, self.width, 3)

    def picture(self):
        return self.rgb8.view(self.height, self.width, 3)


def render_step(ctx, cam, frame, total_spp, stream=None, all_reduce=None, tonemap=True):
    """One frame with the seed shard currently uploaded in ctx: render the clamped sum into
    frame.accum, sum it over ranks when a collective is given (an all-reduce, or a reduce to the one
    rank that keeps the picture), normalise by total_spp and tone-map into frame.rgb8 (tonemap=False:
    ranks that do not hold the summed frame skip that).  Everything is enqueued on `stream`."""
    ctx.render_frame_device(cam, frame.accum.data_ptr(), OUT_SUM, stream)
    if all_reduce is not None:
        all_reduce(frame.accum)
    if tonemap:
        ctx.tonemap_device(frame.accum.data_ptr(), 1.0 / float(total_spp), cam.width, cam.height,
                           frame.rgb8.data_ptr(), stream)


def split_spp(total_spp, num_shards):
    """Samples per shard when total_spp is not a multiple of the shard count: the first
    total_spp % num_shards shards take one sample more (wbrt_split_spp)."""
    from . import split_spp as _split
    return _split(total_spp, num_shards)


def render_sharded_frame(ctx, cam, frame, num_shards, base_seed, rank, world, stream=None, all_reduce=None,
                         shard_spp=None, tonemap=True):
    """Strong-scaling form: a frame of num_shards seed shards split over `world` ranks.  Every
    shard holds cam.samplesPerPixel samples, or shard_spp[v] when a list is given (see
    split_spp).  The shard loop is the C ABI's (wbrt_render_shards: seeds generated on the device,
    no host synchronisation between shards).  Returns the number of shards this rank rendered."""
    total = num_shards * cam.samplesPerPixel if shard_spp is None else sum(shard_spp)
    ctx.render_shards(cam, frame.accum.data_ptr(), num_shards, base_seed, rank, world, shard_spp, stream)
    if all_reduce is not None:
        all_reduce(frame.accum)
    if tonemap:
        ctx.tonemap_device(frame.accum.data_ptr(), 1.0 / float(total), cam.width, cam.height,
                           frame.rgb8.data_ptr(), stream)
    return len(shards_for_rank(num_shards, rank, world))

"""The entry point that allocates nothing (wbrt_render_frame_device_noalloc) against the entry point it
is a variant of (wbrt_render_frame_device): same image bit for bit, on two calls in a row, the second
with a smaller frame, through the C ABI.  Collected last: it shares one static device array per GPU."""
import numpy as np
import pytest

from util import pkg, bits

pytestmark = pytest.mark.gpu


def _render_both(ctx, torch, scene, cam, blocks, seed, out_mode=pkg.OUT_MEAN):
    bw, bh = pkg.set_blocks(cam, blocks)
    ctx.upload_scene(scene)
    ctx.upload_seeds(pkg.make_seeds(bw * bh, seed))
    n = cam.height * cam.width * 3
    want = torch.full((n,), 0.25, dtype=torch.float32, device="cuda:0")
    got = torch.full((n,), 0.25, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ctx.render_frame_device(cam, want.data_ptr(), out_mode)
    ctx.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    ctx.render_frame_device_noalloc(cam, got.data_ptr(), out_mode)
    ctx.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before  # no device memory taken or returned by the call
    return got.cpu().numpy(), want.cpu().numpy()


def test_noalloc_entry_equals_allocating_entry_twice_in_a_row():
    import torch
    ctx = pkg.Context(0)
    try:
        # first call: 640x360 (large enough for the scheduled path: pilot + sort + ordered launch)
        got, want = _render_both(ctx, torch, pkg.scene_readme(), pkg.camera_readme(640, 40, 50), 4, 41)
        assert np.array_equal(bits(got), bits(want))
        assert np.isfinite(got).all() and got.max() > 0.0
        # second call, smaller frame, a mesh scene (the schedule also carries walk costs), accumulating output
        scene = pkg.scene_stress(40, 24, 24, 4, 4)
        got, want = _render_both(ctx, torch, scene, pkg.camera_stress(384, 32, 50), 4, 42, pkg.OUT_ACCUMULATE)
        assert np.array_equal(bits(got), bits(want))
        # third: a frame too small for the schedule (plain launch) still works
        got, want = _render_both(ctx, torch, pkg.scene_readme(), pkg.camera_readme(64, 9, 50), 1, 43)
        assert np.array_equal(bits(got), bits(want))
    finally:
        ctx.close()


def test_noalloc_entry_leaves_the_context_schedule_buffers_unallocated():
    """A context that only ever used the entry point that allocates nothing owns no schedule arrays afterwards (a
    hipMalloc/hipFree pair inside the call would not show in the free-memory figure); also on a side stream with
    kernel timing switched on, which the entry point ignores."""
    import torch
    ctx = pkg.Context(0)
    try:
        cam = pkg.camera_readme(640, 24, 50)
        bw, bh = pkg.set_blocks(cam, 4)
        ctx.upload_scene(pkg.scene_readme())
        ctx.upload_seeds(pkg.make_seeds(bw * bh, 44))
        n = cam.height * cam.width * 3
        got = torch.zeros(n, dtype=torch.float32, device="cuda:0")
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        ctx.set_timing(True)
        ctx.render_frame_device_noalloc(cam, got.data_ptr(), stream=side.cuda_stream)
        ctx.synchronize()
        assert ctx.get_timing() == (0.0, 0)  # no events of the context were touched
        ctx.set_timing(False)
        for name in ("costs", "tile_costs", "order", "hist"):
            assert ctx.buffer_bytes(name) == 0, name
        want = torch.zeros(n, dtype=torch.float32, device="cuda:0")
        ctx.render_frame_device(cam, want.data_ptr())
        ctx.synchronize()
        assert ctx.buffer_bytes("costs") > 0  # the allocating entry point is the one that grows them
        assert np.array_equal(bits(got.cpu().numpy()), bits(want.cpu().numpy()))
    finally:
        ctx.close()


def test_noalloc_entry_rejects_frames_beyond_its_static_array():
    import torch
    ctx = pkg.Context(0)
    try:
        cam = pkg.camera_readme(2560, 8, 50)  # 2560x1440 = 3.7 M pixels > 2^21
        bw, bh = pkg.set_blocks(cam, 4)
        ctx.upload_scene(pkg.scene_readme())
        ctx.upload_seeds(pkg.make_seeds(bw * bh, 1))
        buf = torch.zeros(cam.height * cam.width * 3, dtype=torch.float32, device="cuda:0")
        with pytest.raises(pkg.WbrtError) as e:
            ctx.render_frame_device_noalloc(cam, buf.data_ptr())
        assert e.value.code == -1 and "2097152" in str(e.value)
        ctx.render_frame_device(cam, buf.data_ptr())  # the allocating entry point serves it
        ctx.synchronize()
    finally:
        ctx.close()

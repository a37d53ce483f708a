#!/usr/bin/env python3
"""Headline benchmark: Mpaths/s of the README five-sphere scene, 1280x720, 250 samples per
pixel, depth 50, 4 blocks (the reference's default tiling), on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one frame: path-trace every pixel-sample on the GPU (inputs already resident in
HBM), sum the un-normalised accumulation buffers over ranks onto rank 0 (RCCL reduce, N > 1 only),
normalise + gamma to 8-bit on the device (rank 0).

  N = 1   the frame is ONE launch with the reference's semantics (one random stream per pixel, 250 samples);
          every step is a COLD frame: its launch schedule comes from its own pilot launch, nothing is carried
          over from the previous step (config.ms_per_step_schedule_carried_over reports the other case);
  N > 1   strong scaling on the SAME frame: 250 samples per pixel as 8 seed shards of 32/31 samples, shard v
          rendered by rank v % N, one reduce(sum) of the 11 MB accumulation buffer to rank 0 per frame
          (--weak: every rank renders its own 250-spp shard instead, the frame then holds 250*N samples).

The JSON line also carries
  roofline      FP32-VALU roofline of the render kernel: algorithmic FLOPs of one launch
                (counted on the device for exactly this workload) / its mean duration from HIP
                events on the launch stream, against the 157.3 TFLOP/s vector peak;
  cpu_baseline  the reference's own kernel (oracle/_ref, built from its sources) or, if that
                is absent, this repo's CPU oracle, timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

PEAK_FP32_VALU_TFLOPS = 157.3
WIDTH, SPP, DEPTH, BLOCKS = 1280, 250, 50, 4
BASE_SEED = 0x5EED
SHARDS = 8
CPU_SECONDS = 10.0


def cpu_baseline(pkg, scene, cam, seeds, whole_blocks):
    """Times the CPU checker on a bounded sample of the same workload (about CPU_SECONDS of wall time on up to 16
    threads): whole blocks of the frame where a block finishes in seconds, else 64x8-pixel strips spread over the
    frame (mesh scenes: the reference tests every triangle of a hit box for every ray segment)."""
    orc = entry.load_oracle()
    threads = max(1, min(16, os.cpu_count() or 1))
    if orc.have_ref():
        runner, kind = orc.Reference(scene), "reference"
        run = lambda c, s: runner.render_block(c, s, threads)
    else:
        runner, kind = orc.Oracle(scene), "port"
        run = lambda c, s: runner.render_block(c, s, math_mode=orc.MATH_PORTABLE, threads=threads)
    c = cam.copy()
    if whole_blocks:
        bw, bh = int(cam.blockSize.x), int(cam.blockSize.y)
        what = "blocks"
    else:
        bw, bh = 64, 8
        c.blockSize = pkg.vec3(bw, bh, 0)
        what = "64x8-pixel strips spread over the frame"
    nbx, nby = cam.width // bw, cam.height // bh
    s = np.ascontiguousarray(seeds[: bw * bh])
    order = np.random.RandomState(1).permutation(nbx * nby) if not whole_blocks else np.arange(nbx * nby)
    done, t0 = 0, time.perf_counter()
    for b in order:
        c.blockOffset = pkg.vec3(int(b) % nbx, int(b) // nbx, 0)
        run(c, s)
        done += 1
        if time.perf_counter() - t0 > CPU_SECONDS:
            break
    dt = time.perf_counter() - t0
    paths = done * bw * bh * cam.samplesPerPixel
    return {
        "value": round(paths / dt / 1e6, 4), "unit": "Mpaths/s", "cores": threads, "kind": kind,
        "sample": "%d of %d %s (%dx%d px, %d spp) of the same frame, %.1f s" % (done, nbx * nby, what, bw, bh, cam.samplesPerPixel, dt),
    }


def make_workload(pkg, config):
    """Scene, camera, samples per pixel, description and whether the frame is always cut into seed shards, for one of
    BASELINE.json's configurations."""
    if config == "mesh":
        soup = np.load(os.path.join(ROOT, "tests", "golden", "meshes.npz"))["f1_car"]  # parsed vertex soup of the reference's f1_car.obj (data)
        scene = pkg.scene_main_cpp_from_soup(soup)
        total_spp = 100
        cam = pkg.camera_main_cpp(1920, total_spp, DEPTH)
        name = "main.cpp scene: 8 spheres + f1_car mesh (15329 triangles, 1 bounding box), 1920x1080, 100 spp, depth 50, 4 blocks of 960x540 (BASELINE.json configs[2])"
        return scene, cam, total_spp, name, False
    if config == "uhd":
        total_spp = 1000
        name = "README 5-sphere scene, 3840x2160, 1000 spp, depth 50, 4 blocks of 1920x1080 (BASELINE.json configs[3])"
        return pkg.scene_readme(), pkg.camera_readme(3840, total_spp, DEPTH), total_spp, name, True
    if config == "stress":
        total_spp = 500
        name = ("divergence stress: 500 random Lambertian/metal spheres + 100352-triangle mesh in 256 bounding boxes, 3840x2160, "
                "500 spp, depth 50, 4 blocks of 1920x1080 (BASELINE.json configs[4])")
        return pkg.scene_stress(500, 224, 224, 16, 16), pkg.camera_stress(3840, total_spp, DEPTH), total_spp, name, True
    name = "README 5-sphere scene, 1280x720, 250 spp, depth 50, 4 blocks of 640x360, bit-matched 48-bit LCG streams (BASELINE.json configs[1])"
    return pkg.scene_readme(), pkg.camera_readme(WIDTH, SPP, DEPTH), SPP, name, False


def other_config_step_ms(pkg, mg, torch, config, steps, device):
    """ms per COLD step of another BASELINE configuration on one GPU, measured the way the headline is (wall clock around
    `steps` frames, inputs resident, tone map included), in a context of its own: {"ms_per_step", "steps", "Mpaths/s"}."""
    scene, cam, total_spp, _, always_sharded = make_workload(pkg, config)
    bw, bh = pkg.set_blocks(cam, BLOCKS)
    ctx = pkg.Context(device)
    try:
        ctx.upload_scene(scene)
        ctx.upload_seeds(pkg.make_seeds(bw * bh, BASE_SEED))
        frame = mg.FrameBuffers(cam.width, cam.height, torch.device("cuda", device))
        stream = torch.cuda.current_stream().cuda_stream
        shard_spp = pkg.split_spp(total_spp, SHARDS) if always_sharded else None
        if always_sharded:
            cam.samplesPerPixel = max(shard_spp)

        def step(c):
            if always_sharded:
                ctx.set_schedule_reuse(True)  # forgets the recorded costs: the frame's first shard runs a pilot
                mg.render_sharded_frame(ctx, c, frame, SHARDS, BASE_SEED, 0, 1, stream=stream, shard_spp=shard_spp)
            else:
                mg.render_step(ctx, c, frame, total_spp=total_spp, stream=stream)
        if not always_sharded:
            ctx.set_schedule_reuse(False)
        # untimed: the same frame at 1 sample per launch (loads the kernels, grows the context's buffers)
        warm = cam.copy()
        warm.samplesPerPixel = 1
        if always_sharded:
            saved, shard_spp = shard_spp, [1] * SHARDS
            step(warm)
            shard_spp = saved
        else:
            step(warm)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            step(cam)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / steps * 1e3
        return {"ms_per_step": round(ms, 3), "steps": steps, "Mpaths/s": round(cam.width * cam.height * total_spp / ms / 1e3, 1)}
    finally:
        ctx.close()


def profiled_traffic(config):
    """HBM bytes per launch and issue utilisation of the render kernel from the committed PMC summary of this
    configuration (profiles/r3/traffic.json, else r2; condensed from rocprofv3 --pmc passes), or None."""
    for rnd in ("r3", "r2"):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "traffic.json")) as f:
                entry = json.load(f).get(config)
            if entry:
                return entry
        except (OSError, ValueError):
            pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 50 (readme), 20 (mesh), 3 (uhd), 2 (stress)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (readme, mesh), 1 (uhd, stress)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra measurements after the timed region (profiling runs)")
    ap.add_argument("--weak", action="store_true",
                    help="readme/mesh with N > 1: every rank renders its own full-spp shard (weak scaling) instead of the split frame")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="throughput mode: steps alternate over this many contexts/streams so that the tail of one frame "
                         "overlaps the start of the next (default 1: one frame at a time, ms_per_step is a frame time)")
    ap.add_argument("--arith", choices=["strict", "tolerant"], default="strict",
                    help="strict (default): bit-exact IEEE arithmetic; tolerant: WBRT_ARITH_TOLERANT (fused multiply-adds, 1-ulp "
                         "reciprocal / square root), images within the stated tolerance of the reference instead of bit-exact")
    ap.add_argument("--config", choices=["readme", "mesh", "uhd", "stress"], default="readme",
                    help="readme = headline (BASELINE.json configs[1]); mesh = configs[2]: main.cpp spheres + f1_car mesh, "
                         "1920x1080, 100 spp; uhd = configs[3]: README scene 3840x2160, 1000 spp as 8 seed shards of 125 spp "
                         "split over the ranks; stress = configs[4]: 500 random spheres + 100352-triangle "
                         "mesh in 256 bounding boxes, 3840x2160, 500 spp as 8 seed shards of 63/62 spp")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"uhd": 3, "stress": 2, "mesh": 20}.get(args.config, 50)
    if args.warmup is None:
        args.warmup = {"uhd": 1, "stress": 1}.get(args.config, 3)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    # Rehearsal on a box with fewer GPUs than ranks (never used for a reported number):
    # WBRT_BENCH_SHARE_GPU=1 maps ranks onto the GPUs present, WBRT_BENCH_BACKEND=gloo replaces RCCL,
    # which refuses two ranks on one device.
    backend = os.environ.get("WBRT_BENCH_BACKEND", "nccl")
    if os.environ.get("WBRT_BENCH_SHARE_GPU") == "1":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    pkg = entry.load_package()
    mg = __import__("importlib").import_module(entry.PKG_NAME + ".multi_gpu")
    scene, cam, total_spp, name, always_sharded = make_workload(pkg, args.config)
    bw, bh = pkg.set_blocks(cam, BLOCKS)
    # how the frame is cut
    weak = args.weak and world > 1 and not always_sharded
    sharded = always_sharded or (world > 1 and not weak)
    if sharded:
        if SHARDS % world:
            raise SystemExit("the frame is %d seed shards: use 1, 2, 4 or 8 GPUs" % SHARDS)
        shard_spp = pkg.split_spp(total_spp, SHARDS)
        cam.samplesPerPixel = max(shard_spp)
        workload = name + "; frame = %d seed shards of %d/%d spp, shard v on rank v %% %d" % (SHARDS, max(shard_spp), min(shard_spp), world)
        paths_per_step = cam.width * cam.height * total_spp
        scaling = "strong"
    else:
        shard_spp = None
        workload = name + ("; one launch with the reference's semantics" if world == 1 else "; WEAK scaling: one %d-spp seed shard per rank" % total_spp)
        paths_per_step = cam.width * cam.height * total_spp * world
        scaling = "weak"
    seeds = pkg.make_seeds(bw * bh, BASE_SEED, shard=rank)  # single-launch steps: rank r renders seed shard r

    nfl = max(1, args.frames_in_flight)
    if nfl > 1 and sharded:
        raise SystemExit("--frames-in-flight applies to the one-launch-per-frame configurations")
    ctxs, frames, streams = [], [], []
    for k in range(nfl):
        c = pkg.Context(local_rank)
        c.upload_scene(scene)
        c.upload_seeds(seeds)
        if args.arith == "tolerant":
            c.set_arith_mode(pkg.ARITH_TOLERANT)
        ctxs.append(c)
        frames.append(mg.FrameBuffers(cam.width, cam.height, torch.device("cuda", local_rank)))
        # streams of equal priority were observed to share one hardware queue (no overlap); alternating the
        # priority puts consecutive frames on different queues
        streams.append(torch.cuda.current_stream() if nfl == 1 else torch.cuda.Stream(priority=-(k % 2)))
    ctx, frame, stream = ctxs[0], frames[0], streams[0].cuda_stream
    step_no = [0]
    cold = [True]  # every step schedules itself (pilot launch); False: costs carried over from the previous step

    def all_reduce(t, op=None):
        op = op or dist.ReduceOp.SUM
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:  # rehearsal backend: stage through the host
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    def frame_reduce(t):
        # only rank 0 keeps the picture: a reduce to it moves half of what an all-reduce moves over xGMI
        if backend == "nccl":
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        else:  # rehearsal backend: stage through the host
            h = t.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                t.copy_(h)

    def step():
        if sharded:
            if cold[0]:
                ctx.set_schedule_reuse(True)  # forgets the recorded costs: the frame's first shard runs a pilot, its other shards reuse it
            mg.render_sharded_frame(ctx, cam, frame, SHARDS, BASE_SEED, rank, world, stream=stream,
                                    all_reduce=frame_reduce if world > 1 else None, shard_spp=shard_spp, tonemap=rank == 0)
        elif nfl == 1:
            mg.render_step(ctx, cam, frame, total_spp=total_spp * world, stream=stream,
                           all_reduce=frame_reduce if world > 1 else None, tonemap=rank == 0)
        else:
            k = step_no[0] % nfl
            step_no[0] += 1
            with torch.cuda.stream(streams[k]):
                mg.render_step(ctxs[k], cam, frames[k], total_spp=total_spp * world, stream=streams[k].cuda_stream,
                               all_reduce=frame_reduce if world > 1 else None, tonemap=rank == 0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed(n):
        barrier()
        t = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        barrier()
        return time.perf_counter() - t

    if not sharded:
        for c in ctxs:
            c.set_schedule_reuse(False)  # cold frames: a pilot launch before every frame
    for _ in range(args.warmup):
        step()
    barrier()
    if nfl == 1:
        ctx.set_timing(True)
    dt = timed(args.steps)
    kernel_ms, launches = ctx.get_timing()
    ctx.set_timing(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        all_reduce(t, dist.ReduceOp.MAX)
        dt = float(t.item())

    extras = {}
    if nfl == 1 and not args.no_extras:
        # the same step with the launch schedule carried over from the previous (identical) frame: no pilot launch
        cold[0] = False
        if not sharded:
            ctx.set_schedule_reuse(True)
        step()
        ctx.set_timing(True)
        reps = 1 if always_sharded else 5
        carried = timed(reps) / reps * 1e3
        k_ms, k_n = ctx.get_timing()
        ctx.set_timing(False)
        extras["ms_per_step_schedule_carried_over"] = round(carried, 4)
        extras["kernel_ms_per_launch_schedule_carried_over"] = round(k_ms / max(k_n, 1), 4)
        cold[0] = True
        if not sharded:
            ctx.set_schedule_reuse(False)
        if world == 1 and not sharded and args.config in ("readme", "mesh"):
            # the N > 1 form of this workload on one GPU: the same frame as 8 seed shards (baseline of the scaling curve)
            sspp = pkg.split_spp(total_spp, SHARDS)
            c8 = cam.copy()
            c8.samplesPerPixel = max(sspp)

            def step8():
                ctx.set_schedule_reuse(True)
                mg.render_sharded_frame(ctx, c8, frame, SHARDS, BASE_SEED, 0, 1, stream=stream, shard_spp=sspp)
            step8()
            barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                step8()
            barrier()
            extras["ms_per_step_as_8_seed_shards_on_1_gpu"] = round((time.perf_counter() - t1) / 5 * 1e3, 4)
            ctx.set_schedule_reuse(False)
            ctx.upload_seeds(seeds)
        if args.arith == "strict" and not sharded:
            # the same cold step in the tolerant arithmetic mode, and how far its image is from the strict one
            step()
            torch.cuda.synchronize()
            strict_img = frame.accum.clone()
            ctx.set_arith_mode(pkg.ARITH_TOLERANT)
            step()
            ctx.set_timing(True)
            tol_ms = timed(5) / 5 * 1e3
            tk_ms, tk_n = ctx.get_timing()
            ctx.set_timing(False)
            extras["tolerant_arithmetic"] = {
                "ms_per_step": round(tol_ms, 4), "kernel_ms_per_launch": round(tk_ms / max(tk_n, 1), 4),
                "image_mae_vs_strict": float((frame.accum - strict_img).abs().mean().item() / (total_spp * world)),
                "mode": "WBRT_ARITH_TOLERANT: fused multiply-adds, hardware reciprocal / square root, no margin bookkeeping; cold frames",
            }
            ctx.set_arith_mode(pkg.ARITH_STRICT)
        if args.config == "mesh" and not sharded:
            # the opt-in capped-margin walk (NOT proven equal to the reference's triangle loop; the timed steps ran the
            # default, proven mode), for comparison
            ctx.set_mesh_mode(pkg.MESH_BVH_CAPPED)
            step()
            step()
            extras["ms_per_step_in_WBRT_MESH_BVH_CAPPED"] = round(timed(3) / 3 * 1e3, 3)
            ctx.set_mesh_mode(pkg.MESH_BVH)

    if rank == 0:
        if sharded:
            # a rank's shards run as one fused launch where the library can fuse them: samples per launch from the launches counted
            rank_spp = sum(shard_spp[v] for v in pkg.shards_of_rank(SHARDS, 0, world))
            launch_spp = rank_spp / max(launches / max(args.steps, 1), 1e-9)
        else:
            launch_spp = cam.samplesPerPixel
        launch_paths = cam.width * cam.height * launch_spp
        hw_note = None
        if args.config in ("mesh", "stress"):
            # Hardware roofline of a mesh scene: the reference's algorithm tests every triangle of a hit box, the walk skips
            # almost all of them, so counted FLOPs / walk time is not a hardware rate.  The fraction is therefore measured
            # where the kernel executes exactly the counted work: WBRT_MESH_LOOP, at a reduced frame.
            small = pkg.camera_main_cpp(1920, 4, DEPTH) if args.config == "mesh" else pkg.camera_stress(960, 2, DEPTH)
            sbw, sbh = pkg.set_blocks(small, BLOCKS)
            ctx.upload_seeds(pkg.make_seeds(sbw * sbh, BASE_SEED))
            counters = ctx.count_events(small)
            small_flops = pkg.algorithmic_flops(counters)
            small_paths = small.width * small.height * small.samplesPerPixel
            small_buf = torch.zeros(small.width * small.height * 3, dtype=torch.float32, device="cuda")
            ctx.set_mesh_mode(pkg.MESH_LOOP)
            ctx.set_schedule_reuse(False)
            ctx.render_frame_device(small, small_buf.data_ptr(), pkg.OUT_SUM, stream)
            torch.cuda.synchronize()
            ctx.set_timing(True)
            ctx.render_frame_device(small, small_buf.data_ptr(), pkg.OUT_SUM, stream)
            torch.cuda.synchronize()
            loop_ms, loop_n = ctx.get_timing()
            ctx.set_timing(False)
            ctx.set_mesh_mode(pkg.MESH_BVH)
            flops = small_flops / small_paths * launch_paths  # reference-equivalent FLOPs of one launch of the timed workload
            achieved = small_flops / (loop_ms * 1e-3) / 1e12
            hw_note = ("measured in WBRT_MESH_LOOP (the kernel then executes the reference's own triangle loop, i.e. exactly the counted "
                       "FLOPs) on %dx%d x %d spp of the same scene: %.3e FLOP in %.2f ms; the timed workload itself walks a hierarchy "
                       "and skips most of that work" % (small.width, small.height, small.samplesPerPixel, small_flops, loop_ms))
            ctx.upload_seeds(seeds)
        else:
            count_cam = cam.copy()
            count_cam.samplesPerPixel = max(1, min(int(round(launch_spp)), 125))
            counters = ctx.count_events(count_cam)  # exact event counts of (up to 125 spp of) one launch, untimed; scaled below
            flops = pkg.algorithmic_flops(counters) * (launch_spp / count_cam.samplesPerPixel)
            if nfl > 1:  # overlapping launches have no meaningful individual duration: price the frame rate instead
                kernel_ms, launches = dt * 1e3, args.steps
            achieved = flops / (kernel_ms / max(launches, 1) * 1e-3) / 1e12
        if nfl > 1:  # overlapping launches have no meaningful individual duration: the frame rate stands in
            kernel_ms, launches = dt * 1e3, args.steps
        k_ms = kernel_ms / max(launches, 1)
        traffic = profiled_traffic(args.config + ("_tolerant" if args.arith == "tolerant" else ""))
        result = {
            "metric": "Mpaths/s", "value": round(paths_per_step * args.steps / dt / 1e6, 2), "unit": "Mpaths/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": workload,
                "arithmetic": "WBRT_ARITH_STRICT (bit-exact vs a non-contracting IEEE build of the reference kernel)" if args.arith == "strict"
                              else "WBRT_ARITH_TOLERANT (images within MAE 1e-3 / PSNR 40 dB of the reference, not bit-exact)",
                "paths_per_step": paths_per_step,
                "parallelism": ("1 GPU" if world == 1 else
                                "%d ranks, spp split as seed shards + one reduce(sum) of the %.2f MB accumulation buffer to rank 0 per frame" % (
                                    world, cam.width * cam.height * 12 / 1e6)),
                "kernel_ms_per_launch": round(k_ms, 4), "render_launches_per_step": round(launches / max(args.steps, 1), 2), "frames_in_flight": nfl,
                "schedule": "waves launched costliest first; every timed step prices its pixel streams with its own pilot launch (1 to 4 spp, at most "
                            "a 24th of the launch's samples; cold frame, nothing carried over from the previous step)" + ("; the shards a rank renders share that schedule (one fused launch where their sample counts allow)" if sharded else ""),
                "kernel_ms_per_launch_includes": "pilot launch + cost sort + render launch (HIP events on the launch stream)",
            },
            "roofline": {
                "bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4),
                "traffic": None if traffic is None else traffic.get("bytes_per_launch"),
                "traffic_source": None if traffic is None else traffic.get("source"),
                # the timed kernel's own issue utilisation from the same profile (VALU issue against the 2-cycle-per-wave64 ceiling,
                # active lanes per issued instruction, share of wave-cycles spent waiting)
                "timed_kernel_utilisation": None if traffic is None else traffic.get("utilisation"),
                "flops_per_launch": flops, "flop_per_path": round(flops / launch_paths, 2),
                "note": hw_note or ("peak = 256 CUs x 4 SIMDs x 32 lanes x 2 FLOP (FMA) x 2.4 GHz; results are bit-exact with a non-contracting "
                                    "IEEE build, so a counted FLOP issues as a separate mul or add: ceiling 0.5 of peak before divergence"),
            },
        }
        result["config"].update(extras)
        if args.config == "readme":
            result["config"]["readme_published_gpu_frame_s"] = 0.38
        if args.config in ("mesh", "stress"):
            result["config"]["mesh_mode"] = "WBRT_MESH_BVH (default: hierarchy walk with proven margins, same result as the reference's triangle loop)"
            result["config"]["reference_equivalent_tflops"] = round(flops / (k_ms * 1e-3) / 1e12, 2)
        if world == 1 and args.config == "readme" and args.arith == "strict" and nfl == 1 and not args.no_extras:
            # the other BASELINE configurations on this GPU, cold steps, outside the timed region (each also has a bench line of
            # its own: --config mesh / uhd / stress)
            result["config"]["other_configs"] = {
                c: other_config_step_ms(pkg, mg, torch, c, n, local_rank) for c, n in (("mesh", 3), ("uhd", 2), ("stress", 1))}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(pkg, scene, cam.copy(), seeds, whole_blocks=args.config in ("readme", "uhd"))
        print(json.dumps(result), flush=True)
    for c in ctxs:
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

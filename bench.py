#!/usr/bin/env python3
"""bench.py — Mrays/s of the hot path (BASELINE.json metric) on N MI355X GPUs of one node.

A "step" is one frame: one pt_render of the workload (the reference's one optixLaunch per frame,
SimplePathtracer.cpp:73-97) with scene, BVH and probe already resident in HBM.  Default workload is
BASELINE config C3: the procedural 1,000,000-triangle voxel terrain, 1920x1080, 4 spp, depth 8,
Disney BSDF, 2048x1024 sky+sun probe — the frame the metric is quoted on.

N>1 (one process per GPU, torch.distributed over RCCL): the FIXED frame is tile-partitioned (interleaved
64x16 tiles, replicated scene, no data-path collective) — strong scaling, which is what "1/2/4/8 MI355X
scaling" of a 1080p frame means.  `--scaling weak` (opt-in) grows the image to N x the pixels instead.
`--workload c4_terrain1M_4k_16spp_d8` is BASELINE config C4 (3840x2160, 16 spp, tiled across the GPUs).
`value` = rays traced by all ranks / max-over-ranks time of the K timed frames, each a device-synchronised pt_render like the
reference's render() (SimplePathtracer.cpp:96) FOR EVERY N: one launch chain per frame (`subframes_per_batch` 1,
`frames_in_flight` 1) — `--frames-in-flight 3` / `--batch B` make the other schedules the timed mode.  After the timed region, for
the record and never part of `value`: the same frames with three whole frames in flight (`mrays_per_s_pipelined`,
`ms_per_frame_pipelined`; same images bit for bit), on a partitioned frame the same frames as wavefront batches of N subframes
(pt_render_batch: `batched`, `batched_pipelined`), and a loop that renders AND hands every frame over for display (pack -> one RCCL
all-gather of the packed rgba8 strips -> unpack into the display buffer, the exchange of frame k overlapping the rendering of
frame k+1): `ms_per_displayed_frame` = loop time / frames handed over, with the exchange alone as `gather_ms`.
`single_frame_launches` repeats the strict per-frame figure under one name whatever the timed mode is, so the 1/2/4/8 curve can
be read from one key; `n_ranks_seen` is the size of the communicator the collectives ran on (RCCL's for backend nccl).

`python3 bench.py --gpus N` without a launcher starts its own N ranks: N child processes of this script with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, spawned BEFORE this process imports torch or loads the
HIP library (a process that has touched the GPU is never re-executed); rank 0's JSON line is relayed, the exit status is the
worst child's.  Under `python -m torch.distributed.run` (WORLD_SIZE already set) it is a rank like before.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, camera, width, height, spp, depth)
    "c3_terrain1M_1080p_4spp_d8": ("terrain", "TERRAIN_CAMERA", 1920, 1080, 4, 8),
    "c2_cornell_1080p_4spp_d8": ("cornell", "CORNELL_CAMERA", 1920, 1080, 4, 8),
    "c1_cornell_256_1spp_d4": ("cornell", "CORNELL_CAMERA", 256, 256, 1, 4),  # BASELINE config C1's size (tests of the launcher; Disney BSDF here, Lambert is a test case)
    "c4_terrain1M_4k_16spp_d8": ("terrain", "TERRAIN_CAMERA", 3840, 2160, 16, 8),
    # second 1 M-triangle workload (scenes.stadium_scene): rotated, displaced, long thin triangles over six decades of edge length, camera inside
    "stadium1M_1080p_4spp_d8": ("stadium", "STADIUM_CAMERA", 1920, 1080, 4, 8),
    # the shape of the reference's real inputs (main.cpp:171-194: OBJ scenes with diffuse textures): the C3 terrain with texcoords and one
    # 1024^2 RGBA8 texture per height band, written as OBJ + MTL + PNG and read back through objloader (= loadOBJ, Model.cpp:137-212), so
    # every closest hit takes the tex2D branch (deviceProgram.cu:512-523) and the materials are what an MTL file can carry (Kd, Ke)
    "terrain1M_textured_1080p_4spp_d8": ("terrain_textured_obj", "TERRAIN_CAMERA", 1920, 1080, 4, 8),
    # the order of the reference's largest assets (San Miguel, ~10 M triangles): the same terrain generator on a 1500 x 1500 column grid, 9.5 M triangles
    "terrain10M_1080p_4spp_d8": ("terrain10M", "TERRAIN_CAMERA", 1920, 1080, 4, 8),
    # the reference's only published runs (HelloPathtracing_sv4_vmv23, 3840x2160, depth cutoff 4):
    # uniform 8 spp without accumulation, and the 3-region foveated schedule (radii 157/515, 1/2/8 spp)
    "sv4_uniform_terrain1M_4k_8spp_d4": ("terrain", "TERRAIN_CAMERA", 3840, 2160, 8, 4),
    "sv4_foveated_terrain1M_4k_d4": ("terrain", "TERRAIN_CAMERA", 3840, 2160, 8, 4),
}

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: FP32 vector peak
# SURVEY.md §8(d), algorithmic bytes (SoA, no padding):
BYTES_PER_RAY_FRAME = 160            # whole pipeline: 320 B per bounce (one radiance + one shadow ray) = 160 B per ray
BYTES_PER_PIXEL_FRAME = 84           # 16 B accum read + (16*4 + 4) B written per pixel per frame
BYTES_PER_RADIANCE_RAY_TRACE = 48    # traversal stage alone: read ray 32 B + write hit 16 B
BYTES_PER_SHADOW_RAY_TRACE = 36      # read ray 32 B + write 4 B
FLOPS_PER_HIT = 700                  # 2 x BSDFEval + 2 x BSDFPdf + BSDFSample + hit setup


def _strip_comments(text):
    """C/C++ source without comments and blank lines (string literals in these files hold no comment markers worth a parser)."""
    out, i, n = [], 0, len(text)
    while i < n:
        if text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
        elif text[i] == '"':
            j = i + 1
            while j < n and text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        else:
            out.append(text[i])
            i += 1
    return "\n".join(l.rstrip() for l in "".join(out).split("\n") if l.strip())


def source_hash():
    """Hash of the kernel sources WITHOUT their comments: PMC-derived numbers under profiles/ carry it and are only quoted for the
    code they were measured on (a hash of the .so would change with every rebuild, a hash of the text with every note on a
    rejected experiment)."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "optixpathtracer_amd", "csrc")
    files = [os.path.join(src, f) for f in ("pt_device.h", "pt_bvh.h", "pt_bvh8.h", "pt_kernels.h", "pt_fused.h", "pt_host.h", "pt_api.hip", "pt_bvh_build.hip")]
    files.append(os.path.join(ROOT, "include", "pt_detmath.h"))  # (the facade header and the C ABI's comments are not kernel sources)
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(_strip_comments(open(f, encoding="utf-8").read()).encode())
    flags = [l for l in open(os.path.join(src, "Makefile"), encoding="utf-8") if l.startswith("FLAGS")]
    h.update("".join(flags).encode())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3_terrain1M_1080p_4spp_d8", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--depth", type=int, default=0, help="experiments only: override the workload's depth limit (the metric's is 8); the line then says so in config.max_depth")
    ap.add_argument("--no-isolated", action="store_true", help="skip the extra single-stream frames that give per-kernel (non-overlapped) durations")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --share-device rehearses the N>1 path on a 1-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use HIP device 0 (rehearsal only)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal on a 1-GPU box: initialise the process group and run the N>1 code (collectives on device tensors, the displayed-frame loop through RCCL) with a world of one rank")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong", help="N>1: strong (default) = the metric's fixed frame split N ways; weak = per-GPU work fixed, image area grows with N")
    ap.add_argument("--simulate-world", type=int, default=0, help="single process: render only rank 0's tiles of an N-way partition (predicts per-GPU time at N GPUs)")
    ap.add_argument("--simulate-rank", type=int, default=0, help="with --simulate-world N: which rank's share to render (load balance of the interleaved tiles)")
    ap.add_argument("--tile", type=int, nargs=2, default=[64, 16], metavar=("W", "H"), help="tile size of the interleaved partition (multiples of 8)")
    ap.add_argument("--split-shadow", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0, help="concurrent pixel chunks per frame (0 = library default)")
    ap.add_argument("--kernel-timing", type=int, default=0, help="1: per-launch HIP-event timing inside the timed loop (slower; the isolated phase always has it)")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="pt_options.frames_in_flight of the TIMED loop. 0 (default): every frame synchronous like the reference's render(); 3: frame k runs whole on stream k mod 3 and pt_render(k) returns once frame k-2 is complete (three frames overlap, same images); 2: pixel chunks as in the synchronous frame, pt_render(k) waits for frame k-1")
    ap.add_argument("--batch", type=int, default=1, help="timed loop: frames are rendered as wavefront batches of this many subframes (pt_render_batch; the last batch of the loop may be shorter). 1 (default, every N) = every frame its own pt_render, like the reference's loop; the batched schedule (N subframes per launch chain on an N-way partition) is reported beside it under `batched`")
    ap.add_argument("--no-extra-schedules", action="store_true", help="skip the extra frames after the timed region (pipelined / batched figures; profiling runs: keeps the frame count at warmup + steps)")
    ap.add_argument("--launch-check", action="store_true", help="ranks only join the process group, all-reduce their rank, gather one record per rank and print {n_ranks_seen, ranks}: tests the self-launcher without a GPU")
    ap.add_argument("--launch-render", action="store_true", help="with --launch-check (implied by --backend nccl): every rank also creates a renderer on its device, renders its share of C1 (Cornell box, 256x256, 1 spp, depth 4) and the checksums of the shares are gathered and compared with rank 0's render of the whole frame — first contact with N devices in seconds")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # nothing GPU-related has been imported yet

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launch_check:
        return launch_check(rank, world, args.backend, args.launch_render or args.backend == "nccl", 0 if args.share_device else local_rank)

    import torch

    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        if world == 1:  # --force-dist without a launcher
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("MASTER_PORT", "29533")

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"  # device of the tensors handed to collectives

    from optixpathtracer_amd import renderer as R
    from optixpathtracer_amd import scenes

    scene_name, cam_name, w, h, spp, depth = WORKLOADS[args.workload]
    if args.depth > 0:
        depth = args.depth
    if world > 1 and args.scaling == "weak":
        # per-GPU work fixed: the image grows to N x the pixels (same aspect, multiples of 8) and is tile-partitioned
        f = world ** 0.5
        w, h = int(round(w * f / 8)) * 8, int(round(h * f / 8)) * 8
    model = {"terrain": scenes.voxel_terrain, "terrain10M": lambda: scenes.voxel_terrain(n=1500, target_tris=10_000_000), "stadium": scenes.stadium_scene, "cornell": scenes.cornell_box, "terrain_textured_obj": textured_terrain_through_obj}[scene_name]()
    probe = scenes.sky_probe(2048, 1024).BuildCDF()
    cam = getattr(scenes, cam_name)

    r = R.SampleRenderer(model, device=local_rank)
    r.setProbe(probe)
    opts = dict(max_depth=depth, max_paths=args.max_paths, streams=args.streams, split_shadow=args.split_shadow, kernel_timing=args.kernel_timing,
                frames_in_flight=0 if args.kernel_timing else args.frames_in_flight)
    r.setOptions(**opts)
    part_world = world if world > 1 else max(1, args.simulate_world)
    if world > 1 or args.force_dist:
        r.setPartition(rank, world, args.tile[0], args.tile[1])
    elif args.simulate_world > 1:
        r.setPartition(args.simulate_rank % args.simulate_world, args.simulate_world, args.tile[0], args.tile[1])
    r.resize((w, h))
    r.setCamera(R.make_camera(cam, w / h))
    r.launchParams.samples_per_launch = spp

    def barrier():
        r.sync()  # frames in flight finish (and report their errors) before the clock is read
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sv4 = args.workload.startswith("sv4_")
    foveated = "foveated" in args.workload

    if args.batch < 1 or (args.batch > 1 and sv4):
        raise SystemExit("--batch B needs a pt_render workload")

    def render_frame(k, count=1):
        if not sv4:
            r.launchParams.frame.subframe_index = k  # progressive accumulation, like the reference's loop
            if count == 1:
                r.render()
            else:
                r.renderBatch(count)  # subframes k .. k+count-1 as one wavefront batch, same buffers bit for bit
        elif foveated:  # sv4 FOV_ON render(): gaze = cursor; here it circles the image centre
            import math
            gaze = (w // 2 + int(300 * math.cos(0.3 * k)), h // 2 + int(200 * math.sin(0.3 * k)))
            r.renderFoveated(gaze)
        else:  # sv4 FOV_OFF render(): one full-resolution launch, 8 spp, accumulation off
            r.renderRegions([dict(launch_w=w, launch_h=h, factor_x=1, factor_y=1, fill_size=1, cx=w // 2, cy=h // 2, r_inner=0.0,
                                  r_outer=1000000000.0, offset_x=0, offset_y=0, redraw=0, spp=spp, subframe_index=0)], r.SV4_VARIANT)

    def timed_frames(first, n, count=1):
        """n frames from subframe `first` on, `count` per launch chain; returns (seconds, rays) of this rank, frames complete at both ends"""
        barrier()
        a0 = r.stats()
        t_0 = time.perf_counter()
        tk = []
        for k in range(0, n, count):
            render_frame(first + k, count)
            tk.append(time.perf_counter())
        barrier()
        t_1 = time.perf_counter()
        if os.environ.get("BENCH_DEBUG"):
            print("timed_frames", count, [round((b - a) * 1e3, 2) for a, b in zip([t_0] + tk, tk + [t_1])], file=sys.stderr)
        a1 = r.stats()
        assert a1["frames"] - a0["frames"] == n, (a1["frames"], a0["frames"], n)
        return t_1 - t_0, (a1["total_radiance_rays"] + a1["total_shadow_rays"]) - (a0["total_radiance_rays"] + a0["total_shadow_rays"])

    # untimed warm-up: whole launch chains of the timed loop's size (the path state is sized by the chain), so at least --warmup frames
    warm = ((max(args.warmup, 1) + args.batch - 1) // args.batch) * args.batch
    for k in range(0, warm, args.batch):
        render_frame(k, args.batch)
    # A context times the two bit-identical schedules of a small synchronous frame (launch chain, fused bounce loop) against each other over
    # its first frames (pt_stats.schedule bit 8 while it does): that trial belongs to the warm-up, not to the K timed steps — which are then
    # rendered by whichever schedule measured faster ON THIS BOX.  The N=1 frame of the headline metric is too large to take part.
    while not sv4 and opts["frames_in_flight"] < 2 and (r.stats()["schedule"] & 0x100) and warm < 64 * args.batch:
        render_frame(warm, args.batch)
        warm += args.batch
    if dist is not None:  # every rank enters the timed loop at the same subframe index (a trial is 2 x (1 + PT_SCHED_TRIALS) frames on every rank, so this is a no-op unless a rank was disturbed)
        wt = torch.tensor([warm], dtype=torch.int64, device=red_dev)
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        while warm < int(wt.item()):
            render_frame(warm, args.batch)
            warm += args.batch
    barrier()
    keys = ("trace_ms", "shadow_ms", "shade_ms", "other_ms", "render_ms", "trace_launches", "shadow_launches", "shade_launches", "radiance_rays", "shadow_rays", "shaded_hits", "fused_passes")
    agg = dict.fromkeys(keys, 0.0)
    # rays are counted on the device for every frame; pt_stats also keeps the totals since pt_create, so that a loop with frames in
    # flight does not have to read (= wait for) each frame's statistics
    pipelined = opts["frames_in_flight"] >= 2
    per_frame_stats = not pipelined and args.batch == 1 and bool(args.kernel_timing)  # (a pt_get_stats + dict per step is host time inside the timed region: only when its per-launch figures are wanted)
    s0 = r.stats()
    t0 = time.perf_counter()
    n_chains = 0
    stamps = [t0]
    for k in range(0, args.steps, args.batch):
        render_frame(warm + k, min(args.batch, args.steps - k))
        n_chains += 1
        stamps.append(time.perf_counter())  # synchronous schedule: the chain is complete here; frames in flight: host pacing only
        if per_frame_stats:
            st = r.stats()
            for key in agg:
                agg[key] += st[key]
    barrier()
    dt = time.perf_counter() - t0
    step_ms = np.diff(np.asarray(stamps)) * 1e3 / args.batch
    st = r.stats()
    assert st["frames"] - s0["frames"] == args.steps, (st["frames"], s0["frames"])
    rays = (st["total_radiance_rays"] + st["total_shadow_rays"]) - (s0["total_radiance_rays"] + s0["total_shadow_rays"])
    if not per_frame_stats:  # per-launch-chain figures of the LAST chain only, scaled to the loop (the frames differ only by their random numbers)
        agg = {key: st[key] * n_chains for key in keys}
    next_sub = warm + args.steps

    # Other schedules of the same frames, for the record (after the timed region, never part of `value`; same images bit for bit):
    # three whole frames in flight, and — on a partitioned frame — wavefront batches of as many subframes as there are shares
    # (a 1/N share x N subframes per launch chain carries the rays of one whole frame)
    extra = {}
    if not args.no_extra_schedules and not sv4 and not args.kernel_timing:
        n_x = max(6, min(args.steps, 30))
        if not pipelined:
            r.setOptions(**dict(opts, frames_in_flight=3))
            render_frame(next_sub); render_frame(next_sub + 1); render_frame(next_sub + 2)
            tp, rp = timed_frames(next_sub + 3, n_x)
            next_sub += 3 + n_x
            extra["pipelined"] = (tp, rp, n_x)
            r.setOptions(**opts)
        if part_world > 1:
            # the strict reading of the reference's loop on a partitioned frame: every frame its own launch chain (batch 1) — or, when that
            # is the timed mode, the batched schedule
            bc = 1 if args.batch > 1 else part_world
            r.setOptions(**dict(opts, frames_in_flight=0))
            n_b = ((n_x + bc - 1) // bc) * bc
            render_frame(next_sub, bc)
            tb, rb = timed_frames(next_sub + bc, n_b, bc)
            extra["single_frame_launches" if bc == 1 else "batched"] = (tb, rb, n_b, bc)
            next_sub += bc + n_b
            if bc > 1:
                r.setOptions(**dict(opts, frames_in_flight=3))
                render_frame(next_sub, bc); render_frame(next_sub + bc, bc)
                tb3, rb3 = timed_frames(next_sub + 2 * bc, n_b, bc)
                extra["batched_pipelined"] = (tb3, rb3, n_b, bc)
                next_sub += 2 * bc + n_b
            r.setOptions(**opts)
    if pipelined:  # everything after this runs synchronously
        r.setOptions(**dict(opts, frames_in_flight=0))

    tot = torch.tensor([dt, float(rays)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tot.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, rays_all = float(tmax[0]), float(tsum[1])
        per_rank = torch.zeros(world, dtype=torch.float64, device=red_dev)
        per_rank[rank] = dt / args.steps * 1e3
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(x), 3) for x in per_rank.tolist()]
        per_rank_b = torch.zeros(world, dtype=torch.float64, device=red_dev)  # every rank builds its own copy of the tree (replicated scene)
        per_rank_b[rank] = r.stats()["bvh_build_ms"]
        dist.all_reduce(per_rank_b, op=dist.ReduceOp.SUM)
        per_rank_build_ms = [round(float(x), 2) for x in per_rank_b.tolist()]
    else:
        dt_max, rays_all = dt, float(rays)
        per_rank_ms = per_rank_build_ms = None
    extra_out = {}
    for name, tup in extra.items():  # max time over the ranks, rays summed over the ranks
        te = torch.tensor([tup[0]], dtype=torch.float64, device=red_dev)
        re_ = torch.tensor([float(tup[1])], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            dist.all_reduce(re_, op=dist.ReduceOp.SUM)
        nfr = tup[2]
        extra_out[name] = {"ms_per_frame": round(float(te[0]) / nfr * 1e3, 3), "mrays_per_s": round(float(re_[0]) / float(te[0]) / 1e6, 2), "frames": nfr}
        if len(tup) > 3:
            extra_out[name]["subframes_per_batch"] = tup[3]

    # displayed frames: every launch chain's result is handed over for display — pack -> one all-gather of the packed rgba8 strips ->
    # scatter into the display buffer — with the exchange of chain k overlapping the rendering of chain k+1 (multigpu.HandOff over
    # pt_pack_async / pt_unpack_display; whole frames in flight, so the context's own stream is free for the hand-off)
    gather_ms = ms_displayed = None
    if dist is not None:
        from optixpathtracer_amd import multigpu

        packer = multigpu.DevicePacker(r)

        def all_gather(dst, src):
            if args.backend == "nccl":
                dist.all_gather_into_tensor(dst, src)
            else:  # rehearsal: stage through host memory
                d = torch.empty(dst.shape, dtype=dst.dtype)
                dist.all_gather_into_tensor(d, src.cpu())
                dst.copy_(d)

        bd = args.batch
        r.setOptions(**dict(opts, frames_in_flight=3))
        hand = multigpu.HandOff(packer, R.PT_BUF_FRAME, world, all_gather)
        for k in range(3):  # warm: RCCL, the display buffer, the streams
            render_frame(next_sub, bd); next_sub += bd
            hand.collect(); hand.submit()
        hand.collect()
        barrier()
        nd = max(2, min(args.steps // bd, 10))
        g_acc = 0.0
        d0 = time.perf_counter()
        for k in range(nd):
            render_frame(next_sub, bd); next_sub += bd
            g0 = time.perf_counter()
            hand.collect()
            hand.submit()
            g_acc += time.perf_counter() - g0
        hand.collect()
        r.displaySync()
        barrier()
        dd = time.perf_counter() - d0
        tt = torch.tensor([dd, g_acc], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms_displayed = float(tt[0]) / nd * 1e3  # loop time / frames actually handed over for display (one per launch chain)
        gather_ms = float(tt[1]) / nd * 1e3            # host time of one hand-over (wait for the strip, all-gather, enqueue the scatter)
        # the displayed frame must be complete and must be the last frame: rgba8 alpha is 255 wherever a rank wrote, and this rank's own pixels equal its frame buffer
        shown = r.downloadDisplay(R.PT_BUF_FRAME)
        mine = r.download(R.PT_BUF_FRAME)
        px = multigpu.pixel_lists(w, h, world, args.tile[0], args.tile[1])[rank]
        ys, xs = (px >> 16).astype(np.int64), (px & 0xFFFF).astype(np.int64)
        if not ((shown >> 24) == 255).all() or not np.array_equal(shown[ys, xs], mine[ys, xs]):
            raise SystemExit(f"rank {rank}: the displayed frame is incomplete or stale")
        r.setOptions(**opts)

    # isolated per-kernel durations: the timed schedule overlaps three frames (the synchronous one three pixel chunks) on separate streams, so its per-class
    # HIP-event sums include time spent sharing the machine.  A few extra frames with ONE chunk stream give durations that
    # add up to the frame (rank 0, N=1 only; not part of `value`).
    iso = None
    if world == 1 and not args.no_isolated and not sv4 and args.streams == 0:
        r.setOptions(**dict(opts, streams=1, kernel_timing=1, frames_in_flight=0))
        n_iso = 3
        render_frame(0)
        ia = dict.fromkeys(keys, 0.0)
        for k in range(n_iso):
            render_frame(1 + k)
            st1 = r.stats()
            for key in ia:
                ia[key] += st1[key]
        iso = {k: ia[k] / n_iso for k in ia}
        r.setOptions(**opts)

    if rank == 0:
        mrays = rays_all / dt_max / 1e6
        frame_s = dt_max / args.steps
        owned_px = r.ownedPixels()[0]
        # --- roofline.  (1) The frame against HBM with SURVEY.md §8(d)'s formula: (rays x 160 B + pixels x 84 B + scene bytes)
        # per frame / frame time.  (2) The dominant kernel class, the BVH traversal launches (k_trace8_cam for the camera rays — packets of 64 —, one
        # k_trace8<3> per bounce tracing that bounce's closest-hit rays with the previous bounce's shadow rays): §8(d)'s traversal-
        # stage bytes (48 B per closest-hit ray, 36 B per shadow ray) per launch / mean ISOLATED launch time (single-stream frames,
        # HIP events on the launch's own stream, pt_stats).
        # scene bytes read at least once per frame: the wide tree with its leaf triangles (which the shade kernel re-reads — there is no second
        # triangle array any more) and the probe's texels, pdf and cdf rows
        scene_bytes = st["bvh_bytes"] + probe.data.shape[0] * probe.data.shape[1] * (16 + 4 + 4)
        textured = any(mm.diffuseTextureID >= 0 and mm.texcoord is not None for mm in model.meshes)
        if textured:  # the texels and the 64-byte per-triangle records (vertices + texcoords) a textured closest hit reads
            scene_bytes += sum(int(t.pixel.nbytes) for t in model.textures) + 64 * model.num_triangles
        rays_frame = rays_all / args.steps
        px_frame = float(w * h) if world > 1 else float(owned_px)
        alg_frame = rays_frame * BYTES_PER_RAY_FRAME + px_frame * BYTES_PER_PIXEL_FRAME + scene_bytes * (world if world > 1 else 1)
        achieved = alg_frame / frame_s / 1e9 / max(1, world)  # per GPU, against one GPU's peak
        src = iso if iso is not None else {k: agg[k] / args.steps for k in agg}
        n_launch = max(1.0, src["trace_launches"] + src["shadow_launches"])
        trav_ms = (src["trace_ms"] + src["shadow_ms"]) / n_launch
        trav_bytes = (src["radiance_rays"] * BYTES_PER_RADIANCE_RAY_TRACE + src["shadow_rays"] * BYTES_PER_SHADOW_RAY_TRACE) / n_launch
        trav_gbs = trav_bytes / (trav_ms * 1e-3) / 1e9 if trav_ms > 0 else 0.0
        # shade kernel against the FP32 vector peak: closest hits x 700 flop (§8(d)) / isolated k_shade time
        hits = src["shaded_hits"]  # closest hits shaded with an accepted BSDF sample, counted on the device (pt_stats)
        shade = None
        if src["shade_ms"] > 0:
            tf = hits * FLOPS_PER_HIT / (src["shade_ms"] * 1e-3) / 1e12
            shade = {"kernel": "k_shade", "bound": "fp32_valu", "achieved": round(tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf / FP32_PEAK_TFLOPS, 5), "flops_per_hit": FLOPS_PER_HIT, "hits_per_frame": int(hits),
                     "isolated_ms_per_frame": round(src["shade_ms"], 3), "launches_per_frame": int(src["shade_launches"])}
        # PMC-derived figures (HBM traffic, VALU issue / lane utilisation) cannot be collected inside an unprofiled run: they are
        # read from the committed rocprofv3 passes of this same command and quoted only for the sources they were measured on
        traffic = valu = trav_traffic_frame = None
        replayed = None  # the committed file the PMC-derived fields below were read from (they are REPLAYED, not measured by this run)
        shash = source_hash()
        default_cfg = args.workload == "c3_terrain1M_1080p_4spp_d8" and world == 1 and args.simulate_world == 0 and args.batch == 1 and not pipelined
        pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
        tj = pj[-1] if pj else ""
        if default_cfg and os.path.exists(tj):
            try:
                T = json.load(open(tj))
                if T.get("src_hash") == shash:
                    traffic = T.get("traffic_bytes_per_frame")  # per frame, like `achieved` / alg_bytes_per_frame
                    trav_traffic_frame = T.get("traversal_traffic_bytes_per_frame")
                    valu = T.get("valu")
                    replayed = os.path.relpath(tj, ROOT)
                    if shade is not None and "shade" in T:
                        shade["pmc"] = T["shade"]
                        shade["pmc_replayed_from"] = replayed
            except Exception:
                traffic = valu = replayed = None
        limiter = None
        ws = ((valu or {}).get("wave_state") or {}).get("k_trace8<3>")
        if ws:
            limiter = (f"dependent-load latency at 5 waves/SIMD, not HBM: k_trace8<3> waves wait on memory {100 * ws['wait_mem']:.0f} % of their time, "
                       f"VALU pipe {100 * ws['valu_pipe']:.0f} % used, TA {100 * ws['ta_busy']:.0f} % busy, mean L1->L2 round trip {ws['l2_round_trip_cycles']:.0f} cycles")
        kname = "k_trace8<3> + k_trace8_cam"
        fused_passes = int(round(agg["fused_passes"] / args.steps))  # small synchronous frames (a 1/4 or 1/8 share): generate -> trace -> shade rounds inside ONE persistent kernel (csrc/pt_fused.h)
        if src["fused_passes"] > 0:  # (the isolated single-stream frames always run the launch chain)
            kname = "k_path_loop (fused bounce loop: its time holds generate, traversal and shading of the whole pass)"
        strong = world > 1 and args.scaling == "strong"
        out = {
            "metric": "Mrays/s (and ms/frame) at 1080p 4spp depth8; 1/2/4/8 MI355X scaling",
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(frame_s * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if (world == 1 or strong) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": args.workload + (f"_weak_x{world}_area" if (world > 1 and not strong) else ""), "triangles": model.num_triangles, "width": w, "height": h, "spp": spp,
                "max_depth": depth, "bsdf": "disney", "probe": "sky2048x1024+sun", "textures": [list(t.resolution) for t in model.textures] or None, "partition": f"tiles{args.tile[0]}x{args.tile[1]}/{part_world}",
            },
            "rays_per_frame": int(rays_frame),
            "fps": round(args.steps / dt_max, 2),
            # schedule of the timed loop: frames_in_flight 1 = every frame a device-synchronised pt_render (the reference's render()); subframes_per_batch 1 = one frame per launch chain
            "frames_in_flight": opts["frames_in_flight"] if pipelined else 1,
            "subframes_per_batch": args.batch,
            "fused_bounce_loop_passes_per_frame": fused_passes,  # 0: the launch chain (every frame of more than PT_SCHED_MAX_PATHS = 4.5 M paths; the N=1 headline always)
            # small synchronous frames: the context timed its two bit-identical schedules against each other during the warm-up (device time of
            # the trial frames, ms) and the timed steps ran the faster; warmup_frames = the frames actually rendered before the clock started
            "schedule": {"timed_steps": "fused" if (st["schedule"] & 1) else "chain", "trial_chain_ms": round(st["sched_chain_ms"], 3) or None, "trial_fused_ms": round(st["sched_fused_ms"], 3) or None,
                         "warmup_frames": warm},
            # the same frames on the library's other schedules, measured after the timed region (same images bit for bit; not `value`)
            "mrays_per_s_pipelined": extra_out.get("pipelined", {}).get("mrays_per_s"),
            "ms_per_frame_pipelined": extra_out.get("pipelined", {}).get("ms_per_frame"),
            "batched": extra_out.get("batched"),
            "batched_pipelined": extra_out.get("batched_pipelined"),
            # the strict per-frame schedule (every frame its own device-synchronised pt_render) under one key for every N: the timed loop itself unless --batch / --frames-in-flight changed it
            "single_frame_launches": extra_out.get("single_frame_launches") or ({"ms_per_frame": round(frame_s * 1e3, 3), "mrays_per_s": round(mrays, 2), "frames": args.steps, "subframes_per_batch": 1} if (args.batch == 1 and not pipelined) else None),
            # device time from a launch chain's first kernel to its last (with frames in flight: one frame's LATENCY, three frames overlap) — of the last chain of the loop
            "frame_latency_ms": round(agg["render_ms"] / n_chains, 3),
            "ms_per_step_per_rank": per_rank_ms,
            "bvh_build_ms_per_rank": per_rank_build_ms,
            "kernel_ms_per_frame_isolated": None if iso is None else {k: round(iso[k], 3) for k in ("trace_ms", "shadow_ms", "shade_ms", "other_ms", "render_ms")},
            "bvh": {"nodes": st["bvh_nodes"], "levels": st["bvh_levels"], "bytes": st["bvh_bytes"], "build_ms": round(st["bvh_build_ms"], 2), "hierarchy": ["lbvh", "ploc", "imported", "sah"][st["bvh_builder"]]},
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "ms_per_displayed_frame": None if ms_displayed is None else round(ms_displayed, 3),  # loop time / frames handed over, in a loop that hands every launch chain's result over, the exchange overlapping the next chain
            "n_ranks_seen": 1 if dist is None else dist.get_world_size(),  # size of the communicator the collectives ran on (RCCL's for backend nccl)
            # spread of the timed steps on rank 0 (host clock after each device-synchronised step): tells a box effect (boxes differ by about 2 %) from a change
            "step_ms": {"min": round(float(step_ms.min()), 3), "median": round(float(np.median(step_ms)), 3), "max": round(float(step_ms.max()), 3)},
            "roofline": {
                "kernel": "whole frame, all wavefront stages (SURVEY.md 8d: rays x 160 B + pixels x 84 B + scene bytes)", "bound": "hbm",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                # every PMC-derived field of this line (traffic, dominant_kernel.traffic, valu, measured_limiter, shade.pmc) is replayed from this committed
                # rocprofv3 pass of the same command — quoted only while its src_hash equals the hash of the kernel sources — never measured in this run
                "traffic_replayed_from": replayed if traffic is not None else None,
                "alg_bytes_per_frame": int(alg_frame), "scene_bytes": int(scene_bytes),
                # what the counters say limits the dominant kernel — quoted, like every PMC-derived figure, only for the profiled sources
                "measured_limiter": limiter, "measured_limiter_replayed_from": replayed if limiter is not None else None,
                "dominant_kernel": None if trav_ms <= 0 else {
                    "kernel": kname + " (BVH traversal: closest-hit + shadow rays per launch)", "bytes_per_ray": [BYTES_PER_RADIANCE_RAY_TRACE, BYTES_PER_SHADOW_RAY_TRACE],
                    "alg_bytes_per_launch": int(trav_bytes), "avg_launch_ms": round(trav_ms, 4), "isolated": iso is not None,
                    "achieved": round(trav_gbs, 2), "unit": "GB/s", "frac": round(trav_gbs / HBM_PEAK_GBS, 5),
                    # measured fabric traffic of the traversal kernels per frame / launches per frame of THIS (single-stream) schedule
                    "traffic": None if trav_traffic_frame is None else int(trav_traffic_frame / n_launch),
                    "traffic_replayed_from": replayed if trav_traffic_frame is not None else None,
                },
                "shade": shade, "valu": valu, "valu_replayed_from": replayed if valu is not None else None, "src_hash": shash,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, probe, cam, w, h, spp, depth, r.exportBVH())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def textured_terrain_through_obj():
    """scenes.textured_terrain() -> OBJ + MTL + PNG in a scratch directory -> objloader.load_model: the reference's own route for its scenes
    (loadOBJ semantics through the native parser pt_load_obj, every mesh with its own vertex map; untimed).  No cache: 150 MB of OBJ parse in
    a few seconds since round 5 (the Python restatement needed 26-50 s and kept a predictable file in the shared temp directory)."""
    import shutil
    import tempfile

    from optixpathtracer_amd import objloader, scenes

    d = tempfile.mkdtemp(prefix="ptamd_obj_")
    try:
        t0 = time.perf_counter()
        path = scenes.write_obj(scenes.textured_terrain(), os.path.join(d, "terrain.obj"))
        t1 = time.perf_counter()
        model = objloader.load_model(path)
        print(f"[bench] textured terrain: wrote {os.path.getsize(path) >> 20} MiB of OBJ in {t1 - t0:.1f} s, pt_load_obj + textures in {time.perf_counter() - t1:.1f} s: "
              f"{len(model.meshes)} meshes, {model.num_triangles} triangles, {len(model.textures)} textures", file=sys.stderr)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return model


def launch_ranks(n):
    """`python3 bench.py --gpus N` by itself: N children of this very command line, one per GPU, rendezvous on 127.0.0.1.  Runs before
    torch / libptamd.so are imported, so the parent never touches a GPU; no exec anywhere.  Rank 0's stdout is this process's stdout."""
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for rk in range(n):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rk == 0 else subprocess.DEVNULL, text=rk == 0))

    def relay(pipe):  # the contract is ONE JSON line on stdout: anything else rank 0 prints there (gloo's connection banner) goes to stderr
        for line in pipe:
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()

    import threading
    th = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    th.start()
    worst = 0
    try:
        # a rank that dies leaves the others in a collective: when one fails, give the rest a moment and end them
        pending = list(procs)
        while pending:
            for p in list(pending):
                try:
                    rc = p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if rc != 0:
                    worst = worst or (rc if rc > 0 else 128 - rc)
                    deadline = time.time() + 20
                    for q in pending:
                        try:
                            q.wait(timeout=max(0.1, deadline - time.time()))
                        except subprocess.TimeoutExpired:
                            q.kill()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        th.join(timeout=5)
    return worst


def launch_check(rank, world, backend, render=False, device=0):
    """What --launch-check runs in each rank: join the group, all-reduce, gather one record per rank, rank 0 prints the communicator's size and
    the records (no GPU unless backend nccl or --launch-render).  With `render`: first contact with the devices before the long run — every
    rank creates a renderer on its device (scene upload, BVH build, stream probing), renders its share of C1 through pt_set_partition and
    reports build / render times and a checksum of its own pixels; rank 0 renders the whole frame on its device and checks every share
    against it (the union of the shares is the single-GPU frame bit for bit)."""
    import socket

    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(device)
        dist.init_process_group(backend, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([float(rank + 1)], device=cdev)
    dist.all_reduce(t)
    ok = float(t[0]) == world * (world + 1) / 2
    # one record per rank: [rank, pid, device, visible devices, owned pixels, checksum of the owned accum pixels, build us, render us, expected checksums ok]
    rec = [rank, os.getpid(), device, -1, 0, 0, 0, 0, 1]
    expected = None
    if render:
        from optixpathtracer_amd import multigpu
        from optixpathtracer_amd import renderer as R
        from optixpathtracer_amd import scenes

        rec[3] = torch.cuda.device_count()
        w = h = 256
        probe = scenes.sky_probe(256, 128).BuildCDF()

        def frame(part):
            r = R.SampleRenderer(scenes.cornell_box(), device=device)
            r.setProbe(probe)
            r.setOptions(max_depth=4)
            if part:
                r.setPartition(rank, world, 64, 16)
            r.resize((w, h))
            r.setCamera(R.make_camera(scenes.CORNELL_CAMERA, w / h))
            r.launchParams.samples_per_launch = 1
            t0 = time.perf_counter()
            r.render()
            ms = (time.perf_counter() - t0) * 1e3
            acc = r.download(R.PT_BUF_ACCUM).copy()
            build_ms = r.stats()["bvh_build_ms"]
            r.close()
            return acc, build_ms, ms

        def checksum(acc, k):
            px = multigpu.pixel_lists(w, h, world, 64, 16)[k]
            ys, xs = (px >> 16).astype(np.int64), (px & 0xFFFF).astype(np.int64)
            own = acc[ys, xs].view(np.uint32).astype(np.uint64)
            return len(px), int((own * (np.arange(own.size, dtype=np.uint64).reshape(own.shape) % np.uint64(8191) + np.uint64(1))).sum() % np.uint64(1 << 53))

        acc, build_ms, ms = frame(world > 1)
        rec[4], rec[5] = checksum(acc, rank)
        rec[6], rec[7] = int(build_ms * 1e3), int(ms * 1e3)
        if rank == 0:
            whole = frame(False)[0] if world > 1 else acc
            expected = [checksum(whole, k) for k in range(world)]
    mine = torch.tensor(rec, dtype=torch.int64, device=cdev)
    allr = torch.empty(world * len(rec), dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(allr, mine)
    recs = allr.cpu().reshape(world, len(rec)).tolist()
    ok = ok and [x[0] for x in recs] == list(range(world))
    if rank == 0:
        ranks = [{"rank": x[0], "pid": x[1], "device": x[2]} for x in recs]
        if render:
            for k, x in enumerate(recs):
                ranks[k].update(devices_visible=x[3], owned_pixels=x[4], bvh_build_ms=x[6] / 1e3, first_render_ms=x[7] / 1e3,
                                share_equals_whole_frame=(x[4], x[5]) == expected[k])
                ok = ok and ranks[k]["share_equals_whole_frame"]
            ok = ok and sum(x[4] for x in recs) == 256 * 256
        print(json.dumps({"launch_check": ok, "n_gpus": world, "n_ranks_seen": dist.get_world_size(), "backend": backend, "host": socket.gethostname(), "rendered": bool(render), "ranks": ranks}), flush=True)
    flag = torch.tensor([1.0 if ok else 0.0], device=cdev)
    dist.broadcast(flag, 0)
    dist.barrier()
    dist.destroy_process_group()
    if float(flag[0]) != 1.0:
        raise SystemExit(1)


def host_cpu_share():
    """CPUs this process may really use: the scheduler affinity, capped by the cgroup CPU quota (a 1-GPU box of the pool
    shows 256 logical CPUs but a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(model, probe, cam, w, h, spp, depth, bvh8):
    """The scalar C port of the same path (oracle/, 'port') on the GPU box's host cores, on bounded samples of the same
    workload (same scene/camera/spp/depth; rays are counted, not extrapolated): all cores on the workload's own frame (a half-
    resolution sample of frames above 1080p / 4 spp), and ONE thread on a quarter-resolution frame.  The port traverses THE SAME 8-wide tree as the GPU kernels,
    exported through pt_export_bvh (scalar stack traversal, oracle/pt_oracle.c bvh8_traverse).  Timed region = the render only."""
    from oracle import orc
    from optixpathtracer_amd import scenes

    O = orc.Oracle("det")
    sc = O.make_scene(model, False)
    O.set_bvh8(sc, *bvh8)
    pr = O.make_probe(probe)
    cores = host_cpu_share()
    nthreads = min(cores, 64)

    def run(sw, sh, threads):
        U, V, W = scenes.uvw_frame(**cam, aspect=sw / sh)
        t0 = time.perf_counter()
        out = O.render(sc, pr, (U, V, W), cam["eye"], sw, sh, spp, depth, 0, 0, None, threads)
        dt = time.perf_counter() - t0
        rays = out["radiance_rays"] + out["shadow_rays"]
        return rays, dt

    # all cores: the workload's own frame when it is a 1080p / 4 spp frame (≈4 s on 16 cores), a half-resolution sample of larger ones
    sw, sh = (w, h) if w * h * spp <= 9_000_000 else (max(16, w // 2), max(9, h // 2))
    rays, dt = run(sw, sh, nthreads)
    qw, qh = max(16, w // 4), max(9, h // 4)
    rays1, dt1 = run(qw, qh, 1)
    return {
        "value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": nthreads, "kind": "port",
        "sample": f"{sw}x{sh} frame of the same scene/camera/{spp}spp/depth{depth}, {rays} rays in {dt:.2f}s (reference-order ray count); tree = the product's 8-wide BVH exported through pt_export_bvh",
        "single_thread": {"value": round(rays1 / dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                          "sample": f"{qw}x{qh} frame, {rays1} rays in {dt1:.2f}s"},
    }


if __name__ == "__main__":
    main()

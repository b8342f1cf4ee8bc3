#!/usr/bin/env python3
"""bench.py — Mrays/s of the hot path (BASELINE.json metric) on N MI355X GPUs of one node.

A "step" is one frame: one pt_render of the workload (the reference's one optixLaunch per frame,
SimplePathtracer.cpp:73-97) with scene, BVH and probe already resident in HBM.  Default workload is
BASELINE config C3: the procedural 1,000,000-triangle voxel terrain, 1920x1080, 4 spp, depth 8,
Disney BSDF, 2048x1024 sky+sun probe.  N>1: the image is tile-partitioned (interleaved 64x16 tiles,
no data-path collective).  Default for N>1 is WEAK scaling: per-GPU work is fixed, the image grows to N x the
pixels (N=4: 3840x2160, config C4's size) so each rank renders one 1080p frame's worth of paths; `--scaling strong`
splits the fixed 1080p frame N ways instead.  `value` = rays traced by all ranks / max-over-ranks time.  One RCCL all-gather of the packed frame
runs after the timed region as the display hand-off (reported as gather_ms).

Prints ONE JSON line on rank 0.
"""
import argparse
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
    "c4_terrain1M_4k_16spp_d8": ("terrain", "TERRAIN_CAMERA", 3840, 2160, 16, 8),
    # the reference's only published runs (HelloPathtracing_sv4_vmv23, 3840x2160, depth cutoff 4):
    # uniform 8 spp without accumulation, and the 3-region foveated schedule (radii 157/515, 1/2/8 spp)
    "sv4_uniform_terrain1M_4k_8spp_d4": ("terrain", "TERRAIN_CAMERA", 3840, 2160, 8, 4),
    "sv4_foveated_terrain1M_4k_d4": ("terrain", "TERRAIN_CAMERA", 3840, 2160, 8, 4),
}

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_RADIANCE_RAY_TRACE = 32 + 4 + 8  # closest-hit ray: rayO+rayD (32) + queue entry (4) + hit write (8)
BYTES_PER_SHADOW_RAY_TRACE = 32 + 4 + 16  # shadow ray: origin+direction (32) + queue entry (4) + pending contribution (16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3_terrain1M_1080p_4spp_d8", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --share-device rehearses the N>1 path on a 1-GPU box")
    ap.add_argument("--share-device", action="store_true", help="all ranks use HIP device 0 (rehearsal only)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="N>1: weak = per-GPU work fixed (image area grows with N, C4-style); strong = the fixed frame split N ways")
    ap.add_argument("--simulate-world", type=int, default=0, help="single process: render only rank 0's tiles of an N-way partition (predicts per-GPU time at N GPUs)")
    ap.add_argument("--split-shadow", type=int, default=0)
    ap.add_argument("--streams", type=int, default=0, help="concurrent pixel chunks per frame (0 = library default)")
    ap.add_argument("--bvh-kind", type=int, default=0, help="0 = 8-wide compressed BVH (default), 1 = binary BVH")
    ap.add_argument("--trace-kernel", type=int, default=0, help="0 = persistent-wave traversal (default), 1 = first grid-stride kernel")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    dist = None
    if world > 1:
        import torch.distributed as dist

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
    if world > 1 and args.scaling == "weak":
        # per-GPU work fixed: the image grows to N x the pixels (same aspect, multiples of 8; N=4 is exactly 3840x2160,
        # BASELINE config C4's size) and is tile-partitioned, so every rank still renders one 1080p frame's worth of paths
        f = world ** 0.5
        w, h = int(round(w * f / 8)) * 8, int(round(h * f / 8)) * 8
    model = scenes.voxel_terrain() if scene_name == "terrain" else scenes.cornell_box()
    probe = scenes.sky_probe(2048, 1024).BuildCDF()
    cam = getattr(scenes, cam_name)

    r = R.SampleRenderer(model, device=local_rank)
    r.setProbe(probe)
    r.setOptions(max_depth=depth, max_paths=args.max_paths, trace_kernel=args.trace_kernel, bvh_kind=args.bvh_kind, streams=args.streams, split_shadow=args.split_shadow)
    if world > 1:
        r.setPartition(rank, world, 64, 16)
    elif args.simulate_world > 1:
        r.setPartition(0, args.simulate_world, 64, 16)
    r.resize((w, h))
    r.setCamera(R.make_camera(cam, w / h))
    r.launchParams.samples_per_launch = spp

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sv4 = args.workload.startswith("sv4_")
    foveated = "foveated" in args.workload

    def render_frame(k):
        if not sv4:
            r.launchParams.frame.subframe_index = k  # progressive accumulation, like the reference's loop
            r.render()
        elif foveated:  # sv4 FOV_ON render(): gaze = cursor; here it circles the image centre
            import math
            gaze = (w // 2 + int(300 * math.cos(0.3 * k)), h // 2 + int(200 * math.sin(0.3 * k)))
            r.renderFoveated(gaze)
        else:  # sv4 FOV_OFF render(): one full-resolution launch, 8 spp, accumulation off
            r.renderRegions([dict(launch_w=w, launch_h=h, factor_x=1, factor_y=1, fill_size=1, cx=w // 2, cy=h // 2, r_inner=0.0,
                                  r_outer=1000000000.0, offset_x=0, offset_y=0, redraw=0, spp=spp, subframe_index=0)], r.SV4_VARIANT)

    for k in range(args.warmup):
        render_frame(k)
    barrier()
    rays = 0
    agg = dict(trace_ms=0.0, shadow_ms=0.0, shade_ms=0.0, other_ms=0.0, render_ms=0.0, trace_launches=0, shadow_launches=0, radiance_rays=0, shadow_rays=0)
    t0 = time.perf_counter()
    for k in range(args.steps):
        render_frame(args.warmup + k)
        st = r.stats()
        rays += st["radiance_rays"] + st["shadow_rays"]
        for key in agg:
            agg[key] += st[key]
    barrier()
    dt = time.perf_counter() - t0

    tot = torch.tensor([dt, float(rays)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tot.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, rays_all = float(tmax[0]), float(tsum[1])
    else:
        dt_max, rays_all = dt, float(rays)

    # display hand-off: one all-gather of the packed rgba8 frame (outside the timed region)
    gather_ms = None
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

        multigpu.exchange_frame(packer, R.PT_BUF_FRAME, world, all_gather)  # warm RCCL
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        multigpu.exchange_frame(packer, R.PT_BUF_ACCUM, world, all_gather)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        # the assembled frame must be complete: every pixel of accum_buffer was written by exactly one rank (alpha = 1)
        full = r.download(R.PT_BUF_ACCUM)
        if not (full[..., 3] == 1.0).all():
            raise SystemExit(f"rank {rank}: assembled frame has unwritten pixels")

    if rank == 0:
        mrays = rays_all / dt_max / 1e6
        # roofline of the dominant kernels, the BVH traversal launches (k_trace8<0> for the camera rays, then one
        # k_trace8<3> per bounce tracing that bounce's closest-hit rays together with the previous bounce's shadow
        # rays): algorithmic bytes per launch = (radiance rays x 44 B + shadow rays x 52 B) / launches,
        # duration = mean HIP-event time of those launches on their own streams (pt_stats)
        n_launch = max(1, agg["trace_launches"] + agg["shadow_launches"])
        avg_ms = (agg["trace_ms"] + agg["shadow_ms"]) / n_launch
        alg_bytes = (agg["radiance_rays"] * BYTES_PER_RADIANCE_RAY_TRACE + agg["shadow_rays"] * BYTES_PER_SHADOW_RAY_TRACE) / n_launch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic per traversal launch from the committed rocprofv3 PMC passes of this same command
        # (tools/profile.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the guide); PMC cannot be
        # collected inside an unprofiled run, so the figure is read from profiles/ when the workload matches
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r1_traffic.json")
        if os.path.exists(tj) and args.workload == "c3_terrain1M_1080p_4spp_d8" and world == 1 and args.bvh_kind == 0 and args.trace_kernel == 0:
            try:
                T = json.load(open(tj))
                num = den = 0.0
                for k, v in T.items():
                    if k.startswith("k_trace8"):
                        num += (v["fetch_bytes_per_dispatch_x2"] + v["write_bytes_per_dispatch"]) * v["dispatches"]
                        den += v["dispatches"]
                traffic = int(num / den) if den else None
            except Exception:
                traffic = None
        # what actually bounds the path (committed PMC pass, profiles/r1_04_pmc_valu.md): VALU issue slots used by the
        # frame = VALU wave-instructions x 4 SIMD cycles / (SIMDs x clock x frame time), and the lane utilisation
        valu = None
        vj = os.path.join(ROOT, "profiles", "r1_pmc_valu.json")
        if traffic is not None and os.path.exists(vj):
            try:
                P = json.load(open(vj))
                frame_s = dt_max / args.steps
                valu = {"issue_frac": round(P["valu_insts_per_frame"] * P["simd_cycles_per_valu_inst"] / (P["simds"] * P["clock_ghz"] * 1e9 * frame_s), 3),
                        "lane_util_traversal": P["valu_lane_util"]["k_trace8<3>"], "source": P["source"]}
            except Exception:
                valu = None
        kname = {(1, 0): "k_trace2", (1, 1): "k_trace", (0, 1): "k_trace"}.get((args.bvh_kind, args.trace_kernel), "k_trace8<3>/<0>")
        out = {
            "metric": "Mrays/s (and ms/frame) at 1080p 4spp depth8; 1/2/4/8 MI355X scaling",
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": args.workload + (f"_weak_x{world}_area" if (world > 1 and args.scaling == "weak") else ""), "triangles": model.num_triangles, "width": w, "height": h, "spp": spp,
                "max_depth": depth, "bsdf": "disney", "probe": "sky2048x1024+sun", "partition": f"tiles64x16/{world}",
            },
            "rays_per_frame": int(rays_all / args.steps),
            "fps": round(args.steps / dt_max, 2),
            "kernel_ms_per_frame": {k: round(agg[k] / args.steps, 3) for k in ("trace_ms", "shadow_ms", "shade_ms", "other_ms", "render_ms")},
            "bvh": {"nodes": st["bvh_nodes"], "bytes": st["bvh_bytes"], "build_ms": round(st["bvh_build_ms"], 2)},
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "roofline": {
                "kernel": kname + " (BVH traversal: closest-hit + shadow rays)", "bound": "hbm", "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "avg_launch_ms": round(avg_ms, 4), "alg_bytes_per_launch": int(alg_bytes), "valu": valu,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, probe, cam, w, h, spp, depth)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


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


def cpu_baseline(model, probe, cam, w, h, spp, depth):
    """The scalar C port of the same path (oracle/, 'port'), all host cores, on a bounded sample of the
    same workload: the same scene/camera/spp/depth at half resolution (a quarter of the frame's paths;
    rays are counted, not extrapolated).  Timed region = the render only (BVH build excluded, as on the GPU)."""
    import ctypes as C

    from oracle import orc
    from optixpathtracer_amd import scenes

    O = orc.Oracle("det")
    sc = O.make_scene(model, True)
    pr = O.make_probe(probe)
    U, V, W = scenes.uvw_frame(**cam, aspect=w / h)
    cores = host_cpu_share()
    nthreads = min(cores, 64)
    sw, sh = max(16, w // 2), max(9, h // 2)
    U, V, W = scenes.uvw_frame(**cam, aspect=sw / sh)
    t0 = time.perf_counter()
    out = O.render(sc, pr, (U, V, W), cam["eye"], sw, sh, spp, depth, 0, 0, None, nthreads)
    dt = time.perf_counter() - t0
    rays = out["radiance_rays"] + out["shadow_rays"]
    return {
        "value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": nthreads, "kind": "port",
        "sample": f"{sw}x{sh} frame of the same scene/camera/{spp}spp/depth{depth}, {rays} rays in {dt:.2f}s (reference-order ray count)",
    }


if __name__ == "__main__":
    main()

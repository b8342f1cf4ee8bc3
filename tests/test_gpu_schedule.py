"""Schedules that must not change a bit: frames in flight mixed across entry points, the overlapped display hand-off, parallel
enqueue of the multi-context renderer.  Self-comparisons against the synchronous schedule (which the parity tests pin to the checker)."""
import os

import numpy as np
import pytest

from optixpathtracer_amd import scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fif", [2, 3])
def test_render_and_foveated_frames_alternate_in_flight_fullsize(ptlib, fif):
    """pt_render and pt_render_regions alternate while frames are in flight (1920x1080, 70 k triangles, so that a frame is long enough
    for a mis-ordered resolve to show): the foveated launches overwrite / blend pixels the previous pt_render wrote and vice versa, on
    different streams — every resolve has to wait for the other kind's previous frame.  Buffers equal the synchronous run's."""
    from optixpathtracer_amd import renderer as R

    m = scenes.voxel_terrain(n=96, target_tris=70000)
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h = 1920, 1080
    out = {}
    for mode in (0, fif):
        r = R.SampleRenderer(m)
        r.setProbe(probe)
        r.setOptions(frames_in_flight=mode, max_depth=4)
        r.resize((w, h))
        r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
        r.launchParams.samples_per_launch = 2
        sf = 0
        for k in range(10):
            r.launchParams.frame.subframe_index = sf
            if k % 2 == 0:
                r.render()
                sf += 1
            else:
                r.renderFoveated((900 + 20 * k, 540), variant=dict(R.SampleRenderer.SV4_VARIANT))  # increments subframe_index itself
                sf = r.launchParams.frame.subframe_index
        r.sync()
        out[mode] = [r.download(b) for b in (R.PT_BUF_ACCUM, R.PT_BUF_FRAME)]
        r.close()
    for a, b in zip(out[0], out[fif]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_overlapped_handoff_matches_serial(ptlib):
    """pt_pack_async / pt_pack_wait / pt_unpack_display: two ranks of a partitioned frame (two contexts on this device, the all-gather done
    by concatenating their strips) hand frame k-1 over while frame k renders, three frames in flight.  What lands in the display buffer
    after every hand-over — rgba8 and float accum — equals the frame the serial protocol (render, pt_pack, pt_unpack) assembles."""
    import torch

    from optixpathtracer_amd import renderer as R

    m = scenes.voxel_terrain(n=96, target_tris=70000)
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h, spp, nframes = 320, 192, 2, 6

    def make(rank, fif):
        r = R.SampleRenderer(m)
        r.setProbe(probe)
        r.setOptions(frames_in_flight=fif)
        r.setPartition(rank, 2, 16, 8)
        r.resize((w, h))
        r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
        r.launchParams.samples_per_launch = spp
        return r

    def bufs(which, n):
        b = torch.zeros(n, dtype=torch.int32, device="cuda") if which == R.PT_BUF_FRAME else torch.zeros((n, 4), dtype=torch.float32, device="cuda")
        torch.cuda.current_stream().synchronize()  # torch fills on ITS stream; the library's streams do not wait for it (hipStreamNonBlocking)
        return b

    # serial protocol: the reference for every frame
    ser = [make(0, 0), make(1, 0)]
    padded = ser[0].ownedPixels()[1]
    want = {R.PT_BUF_FRAME: [], R.PT_BUF_ACCUM: []}
    for k in range(nframes):
        for r in ser:
            r.launchParams.frame.subframe_index = k
            r.render()
        for which in want:
            send = [bufs(which, padded) for _ in ser]
            for r, s_ in zip(ser, send):
                r.pack(which, s_.data_ptr())
            allr = torch.cat(send)
            torch.cuda.current_stream().synchronize()  # the library's streams do not wait for torch's
            ser[0].unpack(which, allr.data_ptr())
            want[which].append(ser[0].download(which))
    for r in ser:
        r.close()

    for fif in (3, 2):
        ranks = [make(0, fif), make(1, fif)]
        for which in (R.PT_BUF_FRAME, R.PT_BUF_ACCUM):
            send = [[bufs(which, padded), bufs(which, padded)] for _ in ranks]
            for k in range(nframes + 1):
                if k < nframes:
                    for r in ranks:
                        r.launchParams.frame.subframe_index = k
                        r.render()  # enqueued; returns when frame k - (fif - 1) is complete
                if k > 0:  # collect frame k-1 while frame k renders
                    slot = (k - 1) & 1
                    for r in ranks:
                        r.packWait(slot)
                    allr = torch.cat([send[0][slot], send[1][slot]])
                    torch.cuda.current_stream().synchronize()
                    for r in ranks:
                        r.unpackDisplay(which, allr.data_ptr())
                    for r in ranks:
                        r.displaySync()  # allr may be freed after this
                if k < nframes:
                    for i, r in enumerate(ranks):
                        r.packAsync(which, send[i][k & 1].data_ptr(), k & 1)
                if k > 0:
                    for r in ranks:
                        got = r.downloadDisplay(which)
                        assert np.array_equal(got.view(np.uint32), want[which][k - 1].view(np.uint32)), (fif, which, k - 1)
            for r in ranks:  # the next buffer's loop starts the accumulation over
                r.sync()
        for r in ranks:
            r.close()


def test_multi_parallel_enqueue_and_overlapped_handoff(ptlib, capsys):
    """pt_multi with 8 contexts on this device: every rank's launches are enqueued by its own host thread (pt_multi_stats.threads,
    .enqueue_ms), and with frames in flight the hand-over of frame k-1 overlaps frame k: each render(out) returns frame k-1 — rgba8 in
    `out`, float accum in the display buffer — bit-equal to the single-context frame, flush() hands over the last one."""
    from optixpathtracer_amd import renderer as R

    m = scenes.voxel_terrain(n=96, target_tris=70000)
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h, spp, nframes = 640, 360, 2, 6
    single = R.SampleRenderer(m)
    single.setProbe(probe)
    single.resize((w, h))
    single.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
    single.launchParams.samples_per_launch = spp
    want = []
    for k in range(nframes):
        single.launchParams.frame.subframe_index = k
        single.render()
        want.append((single.download(R.PT_BUF_FRAME), single.download(R.PT_BUF_ACCUM)))
    single.close()

    for fif in (0, 3, 2):
        mr = R.MultiRenderer(m, devices=(0,) * 8)
        mr.setProbe(probe)
        mr.setOptions(frames_in_flight=fif)
        mr.resize((w, h))
        mr.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
        mr.launchParams.samples_per_launch = spp
        mr.gather_mask = (1 << R.PT_BUF_FRAME) | (1 << R.PT_BUF_ACCUM)
        out = np.zeros((h, w), np.uint32)
        enq = []
        for k in range(nframes):
            mr.launchParams.frame.subframe_index = k
            out[:] = 0
            mr.render(out)
            st = mr.stats()
            enq.append(st["enqueue_ms"])
            shown = k if fif == 0 else k - 1  # overlapped: the previous frame
            if shown >= 0:
                assert np.array_equal(out, want[shown][0]), (fif, k)
                acc = mr.download(R.PT_BUF_ACCUM, 5) if fif == 0 else mr.downloadDisplay(R.PT_BUF_ACCUM, 5)
                assert np.array_equal(acc.view(np.uint32), want[shown][1].view(np.uint32)), (fif, k)
            else:
                assert not out.any()
        if fif:
            mr.flush(out)
            assert np.array_equal(out, want[-1][0])
            assert np.array_equal(mr.downloadDisplay(R.PT_BUF_ACCUM, 7).view(np.uint32), want[-1][1].view(np.uint32))
            assert mr.stats()["frames_handed_over"] == nframes
        st = mr.stats()
        assert st["threads"] == 8 and st["frames"] == nframes * 8
        with capsys.disabled():
            print(f"\n[pt_multi, 8 contexts on one device, frames_in_flight={fif}] host enqueue time per frame (slowest rank's thread): "
                  f"median {np.median(enq[1:]):.3f} ms, max {max(enq[1:]):.3f} ms")
        assert np.median(enq[1:]) < (0.5 if fif == 3 else 1.0)  # measured: 0.12 ms (whole frames, 21 launches), 0.41 ms (three pixel chunks, 63 launches)
        mr.close()


@pytest.mark.gpu
def test_bench_multi_gpu_code_path_runs_over_rccl_with_one_rank():
    """bench.py's N>1 code — process group on backend nccl (= RCCL), collectives on device tensors, the overlapped displayed-frame loop through
    all_gather_into_tensor — with a world of one rank (--force-dist): what a 1-GPU box can rehearse of the driver's multi-GPU runs.  The
    displayed frame is checked inside bench.py (complete, and the last frame)."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-isolated",
                          "--workload", "c2_cornell_1080p_4spp_d8"], capture_output=True, text=True, timeout=400, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 1 and d["ms_per_displayed_frame"] is not None and d["ms_per_displayed_frame"] > 0
    assert d["ms_per_displayed_frame"] < 1.5 * d["ms_per_step"], d  # the hand-off overlaps the rendering


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with NO launcher (the way the driver issues the N-GPU run): the script spawns its two ranks itself before
    touching the GPU, rank 0's single JSON line comes back on stdout, and the line carries the timed per-frame schedule, the batched one,
    the displayed-frame loop and the communicator's size.  Rehearsed on the one-GPU box with gloo and a shared device."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "4", "--warmup", "1",
                          "--workload", "c2_cornell_1080p_4spp_d8"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "strong"
    assert d["frames_in_flight"] == 1 and d["subframes_per_batch"] == 1  # the timed loop is one pt_render per frame for every N
    assert d["single_frame_launches"]["ms_per_frame"] == d["ms_per_step"]
    assert d["batched"]["subframes_per_batch"] == 2 and d["batched"]["ms_per_frame"] > 0 and d["batched_pipelined"]["ms_per_frame"] > 0
    assert d["ms_per_displayed_frame"] > 0 and d["gather_ms"] >= 0
    assert d["step_ms"]["min"] <= d["step_ms"]["median"] <= d["step_ms"]["max"]
    assert len(d["ms_per_step_per_rank"]) == 2


@pytest.mark.parametrize("n", [2, 3])
def test_launch_check_renders_every_share(ptlib, n):
    """`bench.py --gpus N --launch-check --launch-render` (what `--backend nccl --launch-check` does on an N-GPU node, here over gloo with the
    ranks sharing device 0): every rank creates its renderer, renders its share of C1, and the gathered checksums of the shares equal rank
    0's render of the whole frame; the line names every rank's device, build time and first-render time."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--share-device", "--launch-check", "--launch-render"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.strip()][-1])
    assert d["launch_check"] is True and d["rendered"] is True and d["n_ranks_seen"] == n
    assert sum(x["owned_pixels"] for x in d["ranks"]) == 256 * 256
    assert all(x["share_equals_whole_frame"] and x["bvh_build_ms"] > 0 and x["first_render_ms"] > 0 and x["devices_visible"] >= 1 for x in d["ranks"])


def _partition_frame(monkeypatch, env):
    """a small frame on rank 1 of a 3-way partition (three pixel chunks, launches far below the grid size) under the given environment"""
    from optixpathtracer_amd import renderer as R

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = R.SampleRenderer(scenes.voxel_terrain(n=96, target_tris=70000))
    r.setProbe(scenes.sky_probe(256, 128).BuildCDF())
    r.setPartition(1, 3, 64, 16)
    w, h = 320, 192
    r.resize((w, h))
    r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
    r.launchParams.samples_per_launch = 3
    out = []
    for k in range(2):
        r.launchParams.frame.subframe_index = k
        r.render()
        out.append({b: r.download(b).copy() for b in (R.PT_BUF_ACCUM, R.PT_BUF_COLOR, R.PT_BUF_NORMAL, R.PT_BUF_ALBEDO)} | {"frame": r.downloadPixels().copy()})
    st = r.stats()
    r.close()
    for k in env:
        monkeypatch.delenv(k)
    _partition_frame.last_stats = st
    return out, {k: st[k] for k in ("radiance_rays", "shadow_rays", "shaded_hits", "paths")}


def test_enqueue_threads_do_not_change_a_bit(ptlib, monkeypatch):
    """small synchronous frames enqueue each pixel chunk's chain from its own thread (PT_ENQUEUE_THREADS, read per context at
    pt_create since round 5); the single-thread path and the threaded path must leave the same five buffers and the same ray counts."""
    a, sa = _partition_frame(monkeypatch, {"PT_ENQUEUE_THREADS": "0", "PT_FUSED": "0"})  # (PT_FUSED=0: the launch chain in three pixel chunks)
    assert _partition_frame.last_stats["fused_passes"] == 0 and _partition_frame.last_stats["trace_launches"] > 3
    b, sb = _partition_frame(monkeypatch, {"PT_ENQUEUE_THREADS": "2", "PT_FUSED": "0"})
    assert sa == sb
    for fa, fb in zip(a, b):
        for k in fa:
            assert np.array_equal(fa[k].view(np.uint8), fb[k].view(np.uint8)), k


@pytest.mark.parametrize("cap", ["64", "128", "320"])
def test_fused_bounce_loop_is_bit_identical(ptlib, monkeypatch, cap):
    """pt_fused.h: the bounce loop of a pass as one persistent kernel (every wave runs generate -> trace -> shade rounds on a private window
    of the queue arrays, no barrier between bounces).  Per path the arithmetic, the random numbers and the order of its contributions are the
    launch chain's, so all five buffers and the three device-counted ray totals must be equal — with windows of one, two and five waves'
    worth of entries (refills at every round, paths of different depth side by side in one wave), and with the traversal stack's LDS levels
    cut down so that the spill path is exercised inside the fused kernel too."""
    a, sa = _partition_frame(monkeypatch, {"PT_FUSED": "0"})
    assert _partition_frame.last_stats["fused_passes"] == 0
    b, sb = _partition_frame(monkeypatch, {"PT_FUSED": "2", "PT_FUSED_CAP": cap})  # every pixel chunk's pass fused (three kernels)
    assert _partition_frame.last_stats["fused_passes"] == 3
    c, sc = _partition_frame(monkeypatch, {"PT_FUSED": "2", "PT_FUSED_CAP": cap, "PT_STACK_LDS_SKIP": "7"})
    d, sd = _partition_frame(monkeypatch, {"PT_FUSED_CAP": cap})  # the default policy: a frame this small is ONE fused pass
    assert _partition_frame.last_stats["fused_passes"] == 1 and _partition_frame.last_stats["trace_launches"] == 1
    if cap == "128":  # scenes whose calibration rays cost more than PT_FUSED_MAX_COST steps keep the launch chain (this terrain: 11 steps; default limit 22)
        _partition_frame(monkeypatch, {"PT_FUSED_MAX_COST": "5"})
        assert _partition_frame.last_stats["fused_passes"] == 0
    e, se = _partition_frame(monkeypatch, {})  # ... with the window size chosen by frame size
    assert _partition_frame.last_stats["fused_passes"] == 1
    assert sa == sb == sc == sd == se
    for fa, fb, fc, fd, fe in zip(a, b, c, d, e):
        for k in fa:
            for other in (fb, fc, fd, fe):
                assert np.array_equal(fa[k].view(np.uint8), other[k].view(np.uint8)), k


def test_fused_bounce_loop_soak(ptlib):
    """tools/r5_fused_soak.py for a few seconds: random frame sizes, partitions, samples, depth limits, BSDF modes, window sizes, grids of 1 .. 5120
    waves, chunkings and progressive subframes on four small scenes (terrain, stadium, Cornell box, textured) — accumulation buffer, rgba8 frame
    and ray totals of a PT_FUSED=2 context equal a PT_FUSED=0 context's on every frame (150 s on the GPU box: 5960 frames, profiles/r5_14_fused_soak.log)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "r5_fused_soak.py"), "6"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert "every buffer and every ray count equal" in res.stdout


def test_render_device_refuses_plain_host_memory(ptlib):
    """a caller written against the old render(uint32_t* h_pixels) must not reach a device-to-device copy with a
    pageable host destination: pt_render_device answers PT_ERR_INVALID for a pointer HIP does not know, before rendering anything."""
    import ctypes as C

    from optixpathtracer_amd import renderer as R

    r = R.SampleRenderer(scenes.cornell_box())
    r.setProbe(scenes.sky_probe(64, 32).BuildCDF())
    r.resize((64, 48))
    r.setCamera(R.make_camera(scenes.CORNELL_CAMERA, 64 / 48))
    host = np.zeros((48, 64), np.uint32)
    before = r.stats()["frames"]
    with pytest.raises(RuntimeError, match="not device memory"):
        r.renderDevice(host.ctypes.data_as(C.c_void_p))
    assert r.stats()["frames"] == before and not host.any()
    r.render()  # the context is still usable


@pytest.mark.parametrize("fif", [0, 3])
def test_render_into_a_caller_owned_device_buffer(ptlib, fif):
    """pt_render_device = render(sutil::CUDAOutputBuffer<uint32_t>&) (SimplePathtracer.cpp:99-107): the rgba8 frame lands in the caller's
    DEVICE buffer, complete when the call returns, equal to the frame pt_render + pt_download give; pt_stream is the context's stream."""
    import torch

    from optixpathtracer_amd import renderer as R

    m = scenes.cornell_box()
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h = 200, 120
    r = R.SampleRenderer(m)
    r.setProbe(probe)
    r.setOptions(frames_in_flight=fif)
    r.resize((w, h))
    r.setCamera(R.make_camera(scenes.CORNELL_CAMERA, w / h))
    r.launchParams.samples_per_launch = 2
    assert r.stream not in (None, 0)
    ref = []
    for k in range(3):
        r.launchParams.frame.subframe_index = k
        r.render()
        r.sync()
        ref.append(r.downloadPixels().copy())
    r.uploadAccum(np.zeros((h, w, 4), np.float32))
    dst = torch.full((h, w), 0x55, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for k in range(3):
        r.launchParams.frame.subframe_index = k
        r.renderDevice(dst.data_ptr())
        got = dst.cpu().numpy().view(np.uint32)  # no further synchronisation with the library: the call is synchronous
        assert np.array_equal(got, ref[k]), k
    r.close()


def test_wait_event_orders_the_context_behind_the_callers_stream(ptlib):
    """STREAM CONTRACT of include/pt_amd.h: the library's streams do not wait for the caller's.  A receive buffer filled on a torch side
    stream behind a long-running kernel is handed to pt_unpack after pt_wait_event(event recorded behind the fill): the unpack must see
    the filled buffer although the host never waited for it."""
    import torch

    from optixpathtracer_amd import renderer as R

    m = scenes.cornell_box()
    probe = scenes.sky_probe(64, 32).BuildCDF()
    w, h = 256, 128
    r = R.SampleRenderer(m)
    r.setProbe(probe)
    r.resize((w, h))
    r.setCamera(R.make_camera(scenes.CORNELL_CAMERA, w / h))
    r.render()
    owned, padded = r.ownedPixels()
    assert owned == w * h
    side = torch.cuda.Stream()
    src = torch.zeros(padded, dtype=torch.int32, device="cuda")
    big = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(20):
            big.add_(1.0)  # ~20 x 0.1 ms of work ahead of the fill
        src.fill_(0x01020304)
        ev = torch.cuda.Event()
        ev.record(side)
    r.waitEvent(ev.cuda_event)
    r.unpack(R.PT_BUF_FRAME, src.data_ptr())
    got = r.downloadPixels()
    assert (got == 0x01020304).all()
    s = r.stats()
    assert s["frames"] == 1
    r.close()


def test_sized_stats_getter(ptlib):
    import ctypes as C

    from optixpathtracer_amd import _lib
    from optixpathtracer_amd import renderer as R

    r = R.SampleRenderer(scenes.cornell_box())
    L = _lib.load_library()
    full = _lib.Stats()
    assert L.pt_get_stats(r._ctx, C.byref(full)) == 0
    n = L.pt_stats_size()
    assert n == C.sizeof(_lib.Stats)
    small = (C.c_ubyte * 24)()          # a caller built against a header that ended after `paths`
    big = (C.c_ubyte * (n + 32))(*([0xEE] * (n + 32)))
    assert L.pt_get_stats_n(r._ctx, small, 24) == 0 and L.pt_get_stats_n(r._ctx, big, n + 32) == 0
    assert bytes(small) == bytes(full)[:24] and bytes(big)[:n] == bytes(full) and set(bytes(big)[n:]) == {0}
    r.close()


def test_alternating_schedules_do_not_reallocate_the_path_state(ptlib, monkeypatch):
    """a fused-size synchronous frame takes ONE batch set holding the whole frame, a foveated frame or a frame of more
    samples takes three; the switch used to drain, free and re-allocate every path buffer on EVERY alternation.  The state now only grows:
    after the first cycle pt_stats.path_state_allocs stays put, and every frame equals the frame of a context that never leaves the launch
    chain (PT_FUSED=0), bit for bit."""
    from optixpathtracer_amd import renderer as R

    m = scenes.voxel_terrain(n=64, target_tris=30000)
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h = 480, 272

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = R.SampleRenderer(m)
        for k in env:
            monkeypatch.delenv(k)
        r.setProbe(probe)
        r.resize((w, h))
        r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
        frames, allocs, fused = [], [], []
        sf = 0
        for cycle in range(3):
            # (a) a small synchronous frame: one fused pass by default
            r.launchParams.samples_per_launch = 2
            r.launchParams.frame.subframe_index = sf
            r.render()
            sf += 1
            fused.append(r.stats()["fused_passes"])
            frames.append(r.download(R.PT_BUF_ACCUM).copy())
            # (b) a foveated frame: always three sets
            r.launchParams.frame.subframe_index = sf
            r.renderFoveated((240, 136), inner_radius=40, outer_radius=120, variant=dict(R.SampleRenderer.SV4_VARIANT))
            sf = r.launchParams.frame.subframe_index
            frames.append(r.download(R.PT_BUF_ACCUM).copy())
            # (c) the camera stopped: 40 samples per launch — 5.2 M paths, beyond the fused limit: three chunks
            r.launchParams.samples_per_launch = 40
            r.launchParams.frame.subframe_index = sf
            r.render()
            sf += 1
            fused.append(r.stats()["fused_passes"])
            frames.append(r.download(R.PT_BUF_ACCUM).copy())
            allocs.append(r.stats()["path_state_allocs"])
        r.close()
        return frames, allocs, fused

    fa, aa, fused_a = run({})
    fb, ab, fused_b = run({"PT_FUSED": "0"})
    assert fused_a == [1, 0] * 3 and fused_b == [0, 0] * 3
    assert aa[0] <= 3 and aa[1] == aa[0] and aa[2] == aa[0], aa  # nothing re-allocated after the first cycle
    assert ab[1] == ab[0] and ab[2] == ab[0], ab
    for k, (x, y) in enumerate(zip(fa, fb)):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), k


def _sched_ctx(monkeypatch, env, w=320, h=192, spp=2, world=1):
    from optixpathtracer_amd import renderer as R

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = R.SampleRenderer(scenes.voxel_terrain(n=64, target_tris=30000))
    for k in env:
        monkeypatch.delenv(k)
    r.setProbe(scenes.sky_probe(256, 128).BuildCDF())
    if world > 1:
        r.setPartition(0, world, 64, 16)
    r.resize((w, h))
    r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
    r.launchParams.samples_per_launch = spp
    return r


@pytest.mark.parametrize("initial,fake,want", [("fused", "1.0,2.0", 0), ("chain", "2.0,1.0", 1)])
def test_online_schedule_choice_corrects_a_wrong_first_guess(ptlib, monkeypatch, initial, fake, want):
    """the launch chain and the fused bounce loop are bit-identical, so the context times them against each other instead
    of trusting thresholds.  The logic, with the measured times replaced by constants (PT_SCHED_FAKE: chain ms, fused ms) and the WRONG schedule
    forced as the first guess (PT_SCHED_INITIAL): the frames alternate (guess first, one warm-up + three timed frames each), then every frame
    takes the faster schedule; a change of the frame configuration (samples per launch) starts a new trial; the images never change."""
    r = _sched_ctx(monkeypatch, {"PT_SCHED_TRIALS": "3", "PT_SCHED_INITIAL": initial, "PT_SCHED_FAKE": fake, "PT_SCHED_PROBE": "0"})
    ref = _sched_ctx(monkeypatch, {"PT_SCHED_TRIALS": "0", "PT_FUSED": "0"})
    first = 1 if initial == "fused" else 0
    seq = []
    for k in range(12):
        for x in (r, ref):
            x.launchParams.frame.subframe_index = k
            x.render()
        st = r.stats()
        seq.append((st["schedule"] & 1, bool(st["schedule"] & 0x100), st["fused_passes"]))
        assert np.array_equal(r.download(0).view(np.uint32), ref.download(0).view(np.uint32)), k
    assert [s[0] for s in seq[:8]] == [first, 1 - first] * 4 and all(s[1] for s in seq[:8]), seq  # the trial: alternating, flagged
    assert all(s[0] == want and not s[1] for s in seq[8:]), seq                                   # settled on the faster one
    assert all(s[2] == s[0] for s in seq), seq                                                       # and the flag says what really ran
    st = r.stats()
    c, f = (float(x) for x in fake.split(","))
    assert st["sched_chain_ms"] == c and st["sched_fused_ms"] == f
    r.launchParams.samples_per_launch = 3  # another configuration: measured again
    r.launchParams.frame.subframe_index = 12
    r.render()
    assert r.stats()["schedule"] & 0x100
    r.close()
    ref.close()


def test_online_schedule_choice_measures_real_frames(ptlib, monkeypatch):
    """The same with real device times on a 1/2 share of a small frame: after the trial the context reports both schedules' best frame
    times, runs the one that measured faster, and looks at the loser again every PT_SCHED_PROBE frames without leaving its choice for noise."""
    r = _sched_ctx(monkeypatch, {"PT_SCHED_TRIALS": "2", "PT_SCHED_PROBE": "4"}, w=640, h=360, spp=4, world=2)
    flags = []
    for k in range(6 + 12):
        r.launchParams.frame.subframe_index = k
        r.render()
        flags.append(r.stats()["schedule"])
    st = r.stats()
    assert all(f & 0x100 for f in flags[:6]) and not flags[6] & 0x100, flags
    assert st["sched_chain_ms"] > 0 and st["sched_fused_ms"] > 0
    choice = 1 if st["sched_fused_ms"] < st["sched_chain_ms"] else 0
    settled = [f & 1 for f in flags[6:] if not f & 0x100]
    assert settled.count(choice) >= len(settled) // 2, (flags, st)  # (every fourth settled frame probes the other schedule; a probe that wins by
    assert settled.count(1 - choice) >= 1, flags                    #  more than 5 % starts the trial over: sub-millisecond frames are noisy)
    r.close()

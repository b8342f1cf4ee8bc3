"""pt_render_batch: `count` subframes of the reference's progressive loop as one wavefront batch must leave exactly the buffers
`count` launches leave (deviceProgram.cu:357 seeds by (pixel, subframe); :460-466 blends in subframe order).  Checked against the CPU
checker's frame-by-frame loop at small sizes and against the library's own single launches — whole frame and a 1/8 share — at
BASELINE C3's literal size."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from optixpathtracer_amd import scenes

pytestmark = pytest.mark.gpu

BUFS = ("accum", "frame", "color", "normal", "albedo")


def _renderer(model, probe, cam, w, h, part=None, **opt):
    from optixpathtracer_amd.renderer import SampleRenderer, make_camera

    r = SampleRenderer(model)
    r.setProbe(probe)
    if opt:
        r.setOptions(**opt)
    if part:
        r.setPartition(*part)
    r.resize((w, h))
    r.setCamera(make_camera(cam, w / h))
    return r


def _buffers(r):
    from optixpathtracer_amd import renderer as R

    return dict(accum=r.download(R.PT_BUF_ACCUM), frame=r.download(R.PT_BUF_FRAME), color=r.download(R.PT_BUF_COLOR),
                normal=r.download(R.PT_BUF_NORMAL), albedo=r.download(R.PT_BUF_ALBEDO))


def _same(a, b, what):
    for k in BUFS:
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (what, k)


@pytest.fixture(scope="module")
def small_probe():
    return scenes.sky_probe(256, 128).BuildCDF()


def test_batch_equals_checker_frame_by_frame(ptlib, orc_det, small_probe):
    """4 subframes x 2 spp in one batch == the checker's four launches; then a second batch continues the accumulation (first_subframe 4)."""
    m = scenes.cornell_box()
    w, h, spp = 96, 64, 2
    cam = scenes.CORNELL_CAMERA
    r = _renderer(m, small_probe, cam, w, h)
    r.launchParams.samples_per_launch = spp
    O = orc_det
    sc, pr = O.make_scene(m), O.make_probe(small_probe)
    U, V, W = scenes.uvw_frame(**cam, aspect=w / h)
    accum, rays = None, 0
    for first, count in ((0, 4), (4, 3)):
        r.launchParams.frame.subframe_index = first
        r.renderBatch(count)
        for sf in range(first, first + count):
            o = O.render(sc, pr, (U, V, W), cam["eye"], w, h, spp, 8, sf, 0, accum)
            accum = o["accum"]
            rays += o["radiance_rays"] + o["shadow_rays"]
        g = _buffers(r)
        for k in ("accum", "color", "normal", "albedo"):
            assert_bits_equal(g[k], o[k], f"{k} after subframes {first}..{first + count - 1}")
        assert np.array_equal(g["frame"], o["frame"])
    st = r.stats()
    assert st["frames"] == 7 and st["paths"] == w * h * spp * 3
    assert st["total_radiance_rays"] + st["total_shadow_rays"] <= rays  # provably dead rays are skipped, never more than the reference order


@pytest.mark.parametrize("opts", [dict(), dict(max_paths=7000), dict(max_paths=20000, streams=1), dict(frames_in_flight=2), dict(frames_in_flight=3),
                                  dict(frames_in_flight=3, max_paths=9000)],
                         ids=["default", "passes_cut_subframes", "one_stream", "fif2", "fif3", "fif3_passes"])
def test_batch_is_schedule_invariant(ptlib, small_probe, opts):
    """Batches of 1..4 subframes, interleaved with single launches, on every schedule (passes that end inside a subframe, one stream,
    frames in flight) and on a 1/3 share: always the buffers of the plain frame-by-frame loop."""
    m = scenes.voxel_terrain(n=96, target_tris=70000)
    w, h, spp = 160, 96, 3
    for part in (None, (1, 3, 16, 8)):
        ref = _renderer(m, small_probe, scenes.TERRAIN_CAMERA, w, h, part)
        ref.launchParams.samples_per_launch = spp
        for sf in range(12):
            ref.launchParams.frame.subframe_index = sf
            ref.render()
        want, st_ref = _buffers(ref), ref.stats()
        ref.close()
        r = _renderer(m, small_probe, scenes.TERRAIN_CAMERA, w, h, part, **opts)
        r.launchParams.samples_per_launch = spp
        sf = 0
        for count in (2, 1, 4, 0, 4):  # 0 = an ordinary render() in between
            r.launchParams.frame.subframe_index = sf
            if count == 0:
                r.render()
                sf += 1
            else:
                r.renderBatch(count)
                sf += count
        assert sf == 12
        _same(_buffers(r), want, (opts, part))
        st = r.stats()
        assert st["frames"] == 12
        assert (st["total_radiance_rays"], st["total_shadow_rays"]) == (st_ref["total_radiance_rays"], st_ref["total_shadow_rays"])
        r.close()


def test_batch_with_shadow_catcher(ptlib, orc_det, small_probe):
    """Shadow-catcher scenes run one sample per pass with per-pixel carries; a batch resets them at every subframe boundary."""
    m = scenes.two_box_scene(shadow_catcher=True)
    w, h, spp = 80, 48, 2
    cam = scenes.TWO_BOX_CAMERA
    ref = _renderer(m, small_probe, cam, w, h)
    ref.launchParams.samples_per_launch = spp
    for sf in range(5):
        ref.launchParams.frame.subframe_index = sf
        ref.render()
    want = _buffers(ref)
    for opts in (dict(), dict(frames_in_flight=3)):
        r = _renderer(m, small_probe, cam, w, h, **opts)
        r.launchParams.samples_per_launch = spp
        r.launchParams.frame.subframe_index = 0
        r.renderBatch(3)
        r.launchParams.frame.subframe_index = 3
        r.renderBatch(2)
        _same(_buffers(r), want, opts)


def test_batch_argument_errors(ptlib, small_probe):
    r = _renderer(scenes.cornell_box(), small_probe, scenes.CORNELL_CAMERA, 32, 32)
    with pytest.raises(RuntimeError, match="count"):
        r.renderBatch(0)
    with pytest.raises(RuntimeError, match="count"):
        r.renderBatch(5000)
    r.launchParams.frame.subframe_index = 0xFFFFFFFE
    with pytest.raises(RuntimeError, match="32 bits"):
        r.renderBatch(3)


def test_multi_context_batch(ptlib, small_probe):
    """pt_multi_render_batch: three contexts on one device, a batch of 4 == four pt_multi_render calls == the single context."""
    from optixpathtracer_amd import renderer as R

    m = scenes.voxel_terrain(n=96, target_tris=70000)
    w, h, spp = 160, 96, 2
    single = _renderer(m, small_probe, scenes.TERRAIN_CAMERA, w, h)
    single.launchParams.samples_per_launch = spp
    for sf in range(4):
        single.launchParams.frame.subframe_index = sf
        single.render()
    want = _buffers(single)
    mr = R.MultiRenderer(m, devices=(0, 0, 0))
    mr.setProbe(small_probe)
    mr.resize((w, h))
    mr.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
    mr.launchParams.samples_per_launch = spp
    mr.gather_mask = sum(1 << b for b in range(5))
    mr.launchParams.frame.subframe_index = 0
    mr.renderBatch(4)
    for rank in (0, 2):
        got = dict(zip(BUFS, [mr.download(b, rank) for b in (R.PT_BUF_ACCUM, R.PT_BUF_FRAME, R.PT_BUF_COLOR, R.PT_BUF_NORMAL, R.PT_BUF_ALBEDO)]))
        _same(got, want, rank)
    assert mr.stats()["frames"] == 4 * 3
    mr.close()
    # batches with frames in flight and the overlapped hand-over: two batches of 2, the second call returns the first batch's last frame
    half = _renderer(m, small_probe, scenes.TERRAIN_CAMERA, w, h)
    half.launchParams.samples_per_launch = spp
    for sf in range(2):
        half.launchParams.frame.subframe_index = sf
        half.render()
    want2 = half.download(R.PT_BUF_FRAME)
    mr = R.MultiRenderer(m, devices=(0, 0, 0))
    mr.setProbe(small_probe)
    mr.setOptions(frames_in_flight=3)
    mr.resize((w, h))
    mr.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
    mr.launchParams.samples_per_launch = spp
    out = np.zeros((h, w), np.uint32)
    mr.launchParams.frame.subframe_index = 0
    mr.renderBatch(2, out)
    assert not out.any()  # nothing has been handed over yet
    mr.launchParams.frame.subframe_index = 2
    mr.renderBatch(2, out)
    assert np.array_equal(out, want2)
    mr.flush(out)
    assert np.array_equal(out, want["frame"])
    assert mr.stats()["frames"] == 4 * 3 and mr.stats()["frames_handed_over"] == 2


def test_fullsize_c3_batch_of_4_subframes(ptlib):
    """BASELINE C3 at its literal size (1 M triangles, 1920x1080, 4 spp, depth 8): a batch of 4 subframes equals 4 single launches bit for
    bit in all five buffers and in the ray counts — the whole frame (two passes per chunk: 33 M paths) and a 1/8 share (one pass of
    4.1 M paths, the case the batch exists for), synchronous and with three batches in flight."""
    from optixpathtracer_amd import renderer as R

    m = scenes.voxel_terrain()
    probe = scenes.sky_probe(2048, 1024).BuildCDF()
    w, h = 1920, 1080
    for part in (None, (3, 8, 64, 16)):
        out = {}
        for mode in ("single", "batch", "batch_fif3"):
            r = _renderer(m, probe, scenes.TERRAIN_CAMERA, w, h, part, frames_in_flight=3 if mode == "batch_fif3" else 0)
            r.launchParams.samples_per_launch = 4
            if mode == "single":
                for sf in range(8):
                    r.launchParams.frame.subframe_index = sf
                    r.render()
            else:
                for sf in (0, 4):
                    r.launchParams.frame.subframe_index = sf
                    r.renderBatch(4)
            st = r.stats()
            out[mode] = (_buffers(r), (st["total_radiance_rays"], st["total_shadow_rays"], st["frames"]))
            r.close()
        for mode in ("batch", "batch_fif3"):
            _same(out[mode][0], out["single"][0], (part, mode))
            assert out[mode][1] == out["single"][1] and out[mode][1][2] == 8

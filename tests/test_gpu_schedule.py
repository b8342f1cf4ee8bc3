"""Schedules that must not change a bit: frames in flight mixed across entry points, the overlapped display hand-off, parallel
enqueue of the multi-context renderer.  Self-comparisons against the synchronous schedule (which the parity tests pin to the checker)."""
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

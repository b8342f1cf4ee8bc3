"""Camera rays as wave packets (k_trace8_cam, pt_bvh8.h): forced on for every launch size (PT_CAM_MIN_PATHS=0) and compared bit for bit with
the checker and with the per-ray kernel (PT_CAM_PACKETS=0) — partial last packets, frames smaller than one packet, deep and degenerate
trees, tile partitions (a rank's pixel list is still made of 8 x 8 blocks), wavefront batches, frames in flight, sample passes."""
import numpy as np
import pytest

from optixpathtracer_amd import scenes
from test_gpu_parity import _compare, _gpu_render, _oracle_render, _renderer

pytestmark = pytest.mark.gpu


@pytest.fixture
def packets(monkeypatch):
    monkeypatch.setenv("PT_CAM_PACKETS", "1")
    monkeypatch.setenv("PT_CAM_MIN_PATHS", "0")


def _per_ray(monkeypatch):
    monkeypatch.setenv("PT_CAM_PACKETS", "0")


@pytest.mark.parametrize("scene,cam,size,spp", [
    ("cornell", "CORNELL_CAMERA", (96, 64), 3),
    ("cornell", "CORNELL_CAMERA", (7, 5), 1),       # 35 paths: less than one packet
    ("cornell", "CORNELL_CAMERA", (13, 9), 5),      # 585 paths: the last packet is partial
    ("terrain", "TERRAIN_CAMERA", (160, 96), 2),
    ("two_box", "TWO_BOX_CAMERA", (96, 64), 3),     # shadow-catcher materials
    ("textured", None, (120, 80), 2),
])
def test_packets_match_the_checker(ptlib, orc_det, packets, scene, cam, size, spp):
    probe = scenes.sky_probe(256, 128).BuildCDF()
    if scene == "cornell":
        m = scenes.cornell_box()
    elif scene == "terrain":
        m = scenes.voxel_terrain(n=64, target_tris=30000)
    elif scene == "two_box":
        m = scenes.two_box_scene(shadow_catcher=True)
    else:
        m = scenes.textured_scene()
    c = getattr(scenes, cam) if cam else dict(eye=(3.0, 2.5, -4.5), lookat=(0.0, 0.6, 0.5), up=(0.0, 1.0, 0.0), fovY=45.0)
    w, h = size
    g = _gpu_render(_renderer(m, probe, c, w, h), spp, subframes=2)
    o = _oracle_render(orc_det, m, probe, c, w, h, spp, subframes=2, use_bvh=None if scene == "terrain" else False)
    _compare(g, o)


@pytest.mark.parametrize("scene", ["stadium", "copies", "terrain_partition", "terrain_batch_fif", "terrain_far"])
def test_packets_match_the_per_ray_kernel(ptlib, monkeypatch, scene):
    """Self-comparison at sizes the checker would take minutes for: all five buffers equal with packets forced on and with packets off."""
    from optixpathtracer_amd import renderer as R

    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h, spp = 640, 360, 2
    if scene == "stadium":
        m, cam = scenes.stadium_scene(target_tris=200_000), scenes.STADIUM_CAMERA  # 15-level tree, needle leaves
    elif scene == "copies":
        base = np.array([[0, 0, 0], [4, 0, 0], [0, 3, 0]], np.float32)
        tri = np.repeat(base[None], 6000, 0)  # every packet tests thousands of coincident triangles: ties by primitive id
        m = scenes.Model(meshes=[scenes.TriangleMesh(vertex=tri.reshape(-1, 3).copy(), index=np.arange(18000, dtype=np.uint32).reshape(-1, 3), material=scenes.Material())])
        cam = dict(eye=(1.5, 1.0, 6.0), lookat=(1.5, 1.0, 0.0), up=(0.0, 1.0, 0.0), fovY=50.0)
        w, h, spp = 160, 96, 1
    elif scene == "terrain_far":
        # hit_in_box's tolerance grows with the distance travelled, and beyond ~16 scene sizes it exceeds the builders' box
        # padding — there rounding-noise hits may depend on which boxes were entered, and a packet lane tests triangles under nodes its own ray
        # never entered.  A camera 30 scene sizes away with a 2-degree field of view: packets and per-ray kernel must still agree on this scene
        # (genuine hits are inside every box; the caveat concerns noise hits, pt_bvh8.h k_trace8_cam).
        m = scenes.voxel_terrain(n=96, target_tris=70000)
        e = np.asarray(scenes.TERRAIN_CAMERA["eye"], np.float64)
        cam = dict(eye=tuple(float(x) for x in e / np.linalg.norm(e) * 3000.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovY=2.2)
    else:
        m, cam = scenes.voxel_terrain(n=96, target_tris=70000), scenes.TERRAIN_CAMERA
    out = []
    for mode in ("packets", "per_ray"):
        monkeypatch.setenv("PT_CAM_MIN_PATHS", "0")
        monkeypatch.setenv("PT_CAM_PACKETS", "1" if mode == "packets" else "0")
        r = R.SampleRenderer(m)
        r.setProbe(probe)
        if scene == "terrain_partition":
            r.setPartition(2, 5, 64, 16)
        if scene == "terrain_batch_fif":
            r.setOptions(frames_in_flight=3, max_paths=300_000)  # several sample passes per frame, three frames in flight
        r.resize((w, h))
        r.setCamera(R.make_camera(cam, w / h))
        r.launchParams.samples_per_launch = spp
        if scene == "terrain_batch_fif":
            r.launchParams.frame.subframe_index = 0
            r.renderBatch(3)
            r.launchParams.frame.subframe_index = 3
            r.render()
            r.sync()
        else:
            for sf in range(2):
                r.launchParams.frame.subframe_index = sf
                r.render()
        out.append([r.download(b) for b in (R.PT_BUF_ACCUM, R.PT_BUF_FRAME, R.PT_BUF_COLOR, R.PT_BUF_NORMAL, R.PT_BUF_ALBEDO)] + [r.stats()])
        r.close()
    for a, b in zip(out[0][:5], out[1][:5]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), scene
    assert out[0][5]["total_radiance_rays"] == out[1][5]["total_radiance_rays"] and out[0][5]["total_shadow_rays"] == out[1][5]["total_shadow_rays"]
    assert np.isfinite(out[0][0]).all() and (out[0][0][..., :3] > 0).any()

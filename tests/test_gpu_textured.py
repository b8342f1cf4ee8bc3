"""The path every real input of the reference takes (main.cpp:171-194: OBJ scenes with diffuse textures → loadOBJ → createTextures →
tex2D in the closest-hit program, deviceProgram.cu:512-523, SimplePathtracer.cpp:603-654), at scale: the textured terrain through the
OBJ writer and objloader, against the CPU checker."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import assert_bits_equal
from optixpathtracer_amd import objloader, scenes
from test_gpu_parity import _compare, _gpu_render, _oracle_render, _renderer

pytestmark = pytest.mark.gpu


def test_textured_terrain_through_obj_matches_oracle(ptlib, orc_det, tmp_path):
    """70 k textured triangles written as OBJ + MTL + PNG, read back by objloader (the reference's loadOBJ semantics): the whole render —
    every closest hit samples a 256^2 texture with wrapping texcoords — equals the checker bit for bit, and equals the render of the
    model the file was written from (same triangles in the same order; only the vertex numbering inside the meshes differs)."""
    direct = scenes.textured_terrain(n=96, target_tris=70000, tex_size=256)
    path = scenes.write_obj(direct, str(tmp_path / "terrain.obj"))
    m = objloader.load_model(path)  # the route for rendering: native parser, a vertex map per mesh
    assert m.num_triangles == direct.num_triangles and len(m.textures) == 8 and all(x.diffuseTextureID >= 0 for x in m.meshes)
    for a, b in zip(direct.meshes, m.meshes):
        assert a.vertex[a.index].tobytes() == b.vertex[b.index].tobytes() and a.texcoord[a.index].tobytes() == b.texcoord[b.index].tobytes()
        b.material = a.material  # MTL carries Kd / Ke only: give the loaded meshes the presets so that the two renders are comparable
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h = 192, 120
    g = _gpu_render(_renderer(m, probe, scenes.TERRAIN_CAMERA, w, h), 3, subframes=2)
    o = _oracle_render(orc_det, m, probe, scenes.TERRAIN_CAMERA, w, h, 3, subframes=2)
    _compare(g, o)
    g2 = _gpu_render(_renderer(direct, probe, scenes.TERRAIN_CAMERA, w, h), 3, subframes=2)
    _compare(g2, g)
    alb = g["albedo"][..., :3].reshape(-1, 3)
    assert len(np.unique((alb * 1024).astype(np.int32), axis=0)) > 2000  # texture detail in the first-hit albedo, not 8 material colours


@pytest.mark.parametrize("size", [(1024, 1024), (1000, 600), (7, 5), (1, 1), (64, 2048)])
def test_tex2d_layouts_bit_exact(ptlib, orc_det, size):
    """The software sampler against the checker's row-major restatement for power-of-two sizes (mask wrap, tiled layout) and odd
    sizes (the % path): 40 000 lookups over [-9, 9]^2 plus the texel-centre / edge cases."""
    from optixpathtracer_amd.renderer import SampleRenderer

    tw, th = size
    rng = np.random.default_rng(tw * 31 + th)
    m = scenes.textured_scene()
    m.textures[0] = scenes.Texture(rng.integers(0, 2**32, (th, tw), dtype=np.uint64).astype(np.uint32))
    r = SampleRenderer(m)
    st = rng.uniform(-9, 9, (40000, 2)).astype(np.float32)
    k = np.arange(64, dtype=np.float32)
    st[:64, 0] = (k + 0.5) / tw; st[:64, 1] = (k * 3 + 0.5) / th           # texel centres
    st[64:128, 0] = k / tw; st[64:128, 1] = -k / th                        # texel edges, negative side
    st[128:136] = [[0, 0], [1, 1], [-1, -1], [0.999999, 0.000001], [-1e-7, 1e-7], [8.5, -8.5], [1 - 0.5 / tw, 0.5 / th], [3, -3]]
    g = r.evalTable(7, st, 4)
    tex = m.textures[0].pixel
    ref = np.zeros((len(st), 4), np.float32)
    out = np.zeros(4, np.float32)
    for i in range(len(st)):
        orc_det.lib.orc_tex2d(tex.reshape(-1), tw, th, float(st[i, 0]), float(st[i, 1]), out)
        ref[i] = out
    assert_bits_equal(g, ref, f"tex2D {tw}x{th}")


def test_fullsize_textured_terrain_rows(ptlib, orc_det):
    """The textured 1 M-triangle workload bench.py times (1920x1080, 4 spp, depth 8, eight 1024^2 textures): rows of the full-size frame
    equal the checker's, the frame is deterministic, every camera hit shows texture detail."""
    from oracle import orc as orc_mod

    m = scenes.textured_terrain()
    assert m.num_triangles == 1_000_000 and len(m.textures) == 8 and m.textures[0].pixel.shape == (1024, 1024)
    probe = scenes.sky_probe(2048, 1024).BuildCDF()
    w, h = 1920, 1080
    r = _renderer(m, probe, scenes.TERRAIN_CAMERA, w, h)
    g1 = _gpu_render(r, 4)
    g2 = _gpu_render(r, 4)
    assert np.array_equal(g1["accum"].view(np.uint32), g2["accum"].view(np.uint32)) and np.isfinite(g1["accum"]).all()
    sc = orc_det.make_scene(m, True)
    pr = orc_det.make_probe(probe)
    U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=w / h)
    prm = orc_mod.Params()
    prm.width, prm.height, prm.subframe_index, prm.samples_per_launch, prm.max_depth, prm.bsdf_mode = w, h, 0, 4, 8, 0
    for dst, src in ((prm.eye, scenes.TERRAIN_CAMERA["eye"]), (prm.U, U), (prm.V, V), (prm.W, W)):
        for k in range(3):
            dst[k] = float(src[k])
    rows = [7, 400, 640, 1000]
    accum = np.zeros((h, w, 4), np.float32)
    orc_det.lib.orc_render_rows.argtypes = [C.c_void_p, C.POINTER(orc_mod.Probe), C.POINTER(orc_mod.Params), orc_mod.f32p, orc_mod.i32p, C.c_int, C.c_int]
    orc_det.lib.orc_render_rows(sc.h, C.byref(pr), C.byref(prm), accum.reshape(-1), np.array(rows, np.int32), len(rows), 8)
    for y in rows:
        assert_bits_equal(g1["accum"][y], accum[y], f"row {y} of the textured 1080p frame")
    alb = g1["albedo"][600:700, :, :3].reshape(-1, 3)
    assert len(np.unique((alb * 1024).astype(np.int32), axis=0)) > 20000


def test_reference_fixture_obj_renders_like_the_checker(ptlib, orc_det):
    """tests/golden/obj_fixture/basic.obj through objloader and through the whole pipeline: five small textures of odd sizes (8x5, 6x4, 3x6, 8x5
    again, 2x2 — tiles mostly padding), back-filled texcoords, a texture loaded twice, a missing one, meshes with and without texcoords.  All five
    buffers equal the checker's.  With the reference's own arrays (one vertex map per SHAPE, Model.cpp:176) a mesh of this file indexes a vertex
    it does not have — the reference would read past its vertex buffer; pt_create refuses the scene — so the render uses the loader's
    per-mesh map."""
    from optixpathtracer_amd.renderer import SampleRenderer

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "obj_fixture", "basic.obj")
    with pytest.raises(RuntimeError, match="vertex index out of range"):
        SampleRenderer(objloader.load_obj(path))
    m = objloader.load_model(path)
    assert len(m.meshes) == 8 and len(m.textures) == 5
    probe = scenes.sky_probe(128, 64).BuildCDF()
    cam = dict(eye=(2.6, 2.2, 3.4), lookat=(0.5, 0.5, 0.5), up=(0.0, 1.0, 0.0), fovY=40.0)
    w, h = 96, 64
    g = _gpu_render(_renderer(m, probe, cam, w, h), 4, subframes=2)
    o = _oracle_render(orc_det, m, probe, cam, w, h, 4, subframes=2, use_bvh=False)
    _compare(g, o)
    assert (g["albedo"][..., :3].reshape(-1, 3).sum(1) > 0).mean() > 0.1  # the unit cube is in view

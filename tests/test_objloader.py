"""loadOBJ restatement (Model.cpp:137-212) on a generated OBJ/MTL/PNG: mesh split per (shape, material), vertex
de-duplication per index triple, Kd/Ke, map_Kd with the y mirror, fan triangulation, negative indices."""
import os

import numpy as np

from optixpathtracer_amd import objloader, scenes


def _write_scene(d):
    from PIL import Image

    img = np.zeros((4, 8, 3), np.uint8)
    img[0, :, 0] = 255  # top row red
    img[3, :, 2] = 255  # bottom row blue
    Image.fromarray(img).save(os.path.join(d, "tex.png"))
    open(os.path.join(d, "m.mtl"), "w").write(
        "newmtl red\nKd 0.8 0.1 0.1\nKe 0 0 0\nnewmtl lit\nKd 0.5 0.5 0.5\nKe 2 2 2\nmap_Kd tex.png\n")
    open(os.path.join(d, "s.obj"), "w").write(
        "mtllib m.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nv 1 0 1\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 1\n"
        "o quad\nusemtl lit\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"   # a quad: fan → 2 triangles, 4 distinct vertices
        "usemtl red\nf 1 2 6\nf -6 -2 -1\n"                 # same shape, second material; negative indices
        "o second\nusemtl red\nf 1 5 6\n")
    return os.path.join(d, "s.obj")


def test_load_obj(tmp_path):
    m = objloader.load_obj(_write_scene(str(tmp_path)))
    assert len(m.meshes) == 3 and len(m.textures) == 1
    lit = [x for x in m.meshes if x.diffuseTextureID == 0][0]
    assert len(lit.index) == 2 and len(lit.vertex) == 4 and lit.texcoord.shape == (4, 2)
    assert np.allclose(lit.material["color"], 0.5) and np.allclose(lit.material["emission"], 2.0)
    reds = [x for x in m.meshes if x.diffuseTextureID == -1]
    assert sorted(len(x.index) for x in reds) == [1, 2]
    two = [x for x in reds if len(x.index) == 2][0]
    assert len(two.vertex) == 4 and two.texcoord is None  # (1,2,6) and (1,5,6): 4 distinct positions
    assert np.allclose(two.material["color"], (0.8, 0.1, 0.1))
    px = m.textures[0].pixel
    assert px.shape == (4, 8)
    assert (px[0] & 0xFFFFFF == 0xFF0000).all() and (px[3] & 0xFFFFFF == 0x0000FF).all()  # mirrored: blue row first
    v, idx, tri_mesh, mats = m.flatten()
    assert idx.max() < len(v) and len(idx) == 5

#!/usr/bin/env python3
"""Generates tests/golden/obj_fixture/* (inputs: OBJ / MTL / image files written by THIS script) and tests/golden/ref_model.npz
(outputs: what the REFERENCE's own Model.cpp — loadOBJ, addVertex, loadTexture, addBox, with its vendored tinyobjloader 2.0.0 and
stb_image, compiled from where they lie into oracle/_ref/libptref.so by oracle/Makefile + oracle/ref_build/ref_model.cpp — makes of
them).  Run in the build container only:  python tests/golden/make_model_golden.py
tests/test_objloader.py compares optixpathtracer_amd/objloader.py and scenes.add_box with the stored arrays bit for bit."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from optixpathtracer_amd import scenes  # noqa: E402

FIX = os.path.join(HERE, "obj_fixture")


def write_inputs(d=FIX):
    """The input set.  Every file is small, written here, and committed next to the reference's outputs."""
    from PIL import Image

    os.makedirs(os.path.join(d, "tex"), exist_ok=True)
    rng = np.random.default_rng(77)
    Image.fromarray(rng.integers(0, 256, (5, 8, 3), dtype=np.uint8), "RGB").save(os.path.join(d, "tex", "rgb_5rows.png"))   # odd height: the middle row stays
    Image.fromarray(rng.integers(0, 256, (4, 6, 4), dtype=np.uint8), "RGBA").save(os.path.join(d, "tex", "rgba.png"))
    Image.fromarray(rng.integers(0, 256, (6, 3), dtype=np.uint8), "L").save(os.path.join(d, "tex", "gray.png"))
    Image.fromarray(rng.integers(0, 256, (2, 2, 3), dtype=np.uint8), "RGB").save(os.path.join(d, "tex", "two.bmp"))
    W = lambda name, text, nl="\n": open(os.path.join(d, name), "w", newline="").write(text.replace("\n", nl))  # noqa: E731

    # --- basic: quads, negative indices, shared / unshared normals and texcoords, two materials per shape, Kd / Ke / map_Kd,
    # a missing texture file, g and o, the vertex map shared by a shape's materials, back-filled and zero-padded texcoords / normals
    W("basic.mtl",
      "# materials\nnewmtl red\nKd 0.8 0.1 0.1\nKe 0 0 0\n\nnewmtl lit\nKd 0.5 0.5 0.5\nKe 2 2.5 3\nmap_Kd tex/rgb_5rows.png\n"
      "newmtl missing\nKd 0.25 0.5 0.75\nmap_Kd tex/nope.png\nnewmtl alpha\n\tKd 1 1 1\n\tmap_Kd -bm 0.5 -clamp on tex\\rgba.png\n"
      "newmtl gray\nKe 0.125 0 0\nmap_Kd   tex/gray.png   \nnewmtl bmp\nKd .1 .2 .3\nmap_Kd tex/two.bmp\n")
    W("basic.obj",
      "mtllib basic.mtl\n"
      "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nv 1 0 1\nv 1 1 1\nv 0 1 1\n"
      "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.25 0.75\n"
      "vn 0 0 1\nvn 0 0 -1\nvn 1 0 0\n"
      "o quad\nusemtl lit\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"          # a quad with shared normal
      "usemtl red\nf 1 2 6\nf -8 -4 -3\nf 1/1/1 2/2/1 6\n"       # same shape, second material; negative indices; triples first seen under `lit` (shared map)
      "usemtl lit\nf 5/5 6/2 7/3\nf 5//2 6//2 8//3\n"            # back to lit: texcoord-only, then normal-only vertices (zero padding / back-fill)
      "o second\nusemtl red\nf 1 5 6\n"
      "g grp one two\nusemtl missing\nf 1/1 2/2 3/3\nf 3/3 4/4 1/1\n"
      "g\nusemtl alpha\nf 5/1/1 6/2/1 7/3/1 8/4/1\nusemtl gray\nf 1//1 2//1 3//1\nusemtl alpha\nf 2/2/3 3/3/3 7/3/3\n"
      "o again\nusemtl lit\nf 1/1 2/2 3/3\n"                      # same texture in another shape: loaded a second time
      "usemtl bmp\nf 1/1 3/3 4/4\n")

    # --- concave and non-planar polygons in different planes: the ear clipping and its choice of axes
    W("concave.mtl", "newmtl m\nKd 0.5 0.5 0.5\n")
    poly = [
        "v 0 0 0\nv 4 0 0\nv 4 1 0\nv 1 1 0\nv 1 4 0\nv 0 4 0\nf 1 2 3 4 5 6\n",                         # L in z=0
        "v 0 0 0\nv 0 4 0\nv 0 4 1\nv 0 1 1\nv 0 1 4\nv 0 0 4\nf -6 -5 -4 -3 -2 -1\n",                   # L in x=0
        "v 0 5 0\nv 2 5 1\nv 4 5 0\nv 3 5 3\nv 4 5 6\nv 2 5 5\nv 0 5 6\nv 1 5 3\nf -8 -7 -6 -5 -4 -3 -2 -1\n",  # bow tie-ish star in y=5
        "v 0 0 0\nv 2 0.5 0.1\nv 4 0 0\nv 2 3 0.2\nf -1 -2 -3 -4\n",                                       # arrow head, reversed winding, slightly non-planar
        "v 0 0 0\nv 1 0 0\nv 2 0 0\nv 2 2 0\nv 0 2 0\nf -5 -4 -3 -2 -1\n",                                 # collinear leading corner
        "v 0 0 0\nv 3 0 0\nv 3 3 0\nv 2 1 0\nv 1 2.5 0\nv 0 3 0\nv 0.5 1 0\nf -7 -6 -5 -4 -3 -2 -1\n",   # two reflex vertices
        "v 1 1 1\nv 2 2 2\nv 3 3 3\nv 4 4 4\nf -4 -3 -2 -1\n",                                              # fully degenerate (a line)
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 0.5 0\nf -5 -4 -1 -3 -2\n",                             # self-touching pentagon
    ]
    W("concave.obj", "mtllib concave.mtl\nusemtl m\n" + "".join(poly))

    # --- number formats through tryParseDouble, missing fields, CRLF line ends, tabs
    W("numbers.mtl", "newmtl n\nKd 0.1234567 1e-1 .5\nKe 12.5e-1 +3. 7E0\n", "\r\n")
    W("numbers.obj",
      "mtllib numbers.mtl\nusemtl n\n"
      "v 0.1 0.2 0.3\nv 1.23456789012 -0.000001234 123456.789\nv 1e-3 2.5E+2 -7e0\nv .5 -.25 +3.\nv 7\nv 1 2\nv 0.12345678 0.123456789 0.1234567891\n"
      "v 3.14159265358979 2.71828182845905 1.41421356237310\nv 1e10 1e-10 1.5e-45\nv 16777217 0.3333333333333333 99999.99\nv 1x 2.5abc nan\nv -0 -0.0 1e400\n"
      "vt 0.1 0.9 0.5\nvt 1.5\nvt\nvt 0\n"
      "f 1/1 2/2 3/3\n\tf  4/1  5/2\t6/3 \nf 7 8 9\nf 10 11 12\n", "\r\n")

    # --- MTL quirks: a map_Kd before any Kd takes diffuse 0.6, after one it keeps 0; duplicate names; the last, unnamed block
    W("quirks.mtl", "newmtl first\nmap_Kd tex/gray.png\nnewmtl second\nKd 0.2 0.3 0.4\nnewmtl third\nmap_Kd tex/gray.png\nnewmtl second\nKd 0.9 0.9 0.9\nnewmtl fourth\n")
    W("quirks.obj", "mtllib nothere.mtl quirks.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvt 0 1\n"
      "usemtl first\nf 1/1 2/2 3/3\nusemtl second\nf 2 4 3\nusemtl third\nf 1/1 2/2 4/3\nusemtl fourth\nf 1 4 3\nusemtlsecond\nf 1 2 4\n")
    # --- images: the other formats loadTexture's stb_image decodes (Model.cpp:88-135; the reference's assets use TGA and JPEG beside PNG):
    # TGA true colour, run-length encoded TGA with alpha, grey TGA — lossless, so any decoder must agree bit for bit — and a JPEG, where
    # two conforming decoders may differ in a texel's last bits (stb_image's IDCT and upsampling are its own)
    rng2 = np.random.default_rng(78)
    Image.fromarray(rng2.integers(0, 256, (7, 5, 3), dtype=np.uint8), "RGB").save(os.path.join(d, "tex", "rgb.tga"))
    flat = np.repeat(rng2.integers(0, 256, (4, 3, 4), dtype=np.uint8), 3, axis=1)  # runs of three equal texels: packets of both kinds
    Image.fromarray(flat, "RGBA").save(os.path.join(d, "tex", "rgba_rle.tga"), compression="tga_rle")
    Image.fromarray(rng2.integers(0, 256, (3, 4), dtype=np.uint8), "L").save(os.path.join(d, "tex", "gray.tga"))
    yy, xx = np.mgrid[0:16, 0:24]
    photo = np.stack([(xx * 10) % 256, (yy * 15) % 256, ((xx + yy) * 6) % 256], -1).astype(np.uint8)
    Image.fromarray(photo, "RGB").save(os.path.join(d, "tex", "photo.jpg"), quality=92)
    W("images.mtl", "newmtl a\nKd 1 1 1\nmap_Kd tex/rgb.tga\nnewmtl b\nKd 1 1 1\nmap_Kd tex/rgba_rle.tga\nnewmtl c\nKd 1 1 1\nmap_Kd tex/gray.tga\n"
      "newmtl d\nKd 1 1 1\nmap_Kd tex/photo.jpg\n")
    W("images.obj", "mtllib images.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvt 1 1\n"
      "usemtl a\nf 1/1 2/2 3/3\nusemtl b\nf 2/2 4/4 3/3\nusemtl c\nf 1/1 4/4 3/3\nusemtl d\nf 1/1 2/2 4/4\n")
    return ["basic.obj", "concave.obj", "numbers.obj", "quirks.obj", "images.obj"]


def write_rle_hdr(path, rgbe, header=b"#?RADIANCE\n# comment\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1\n\n"):
    """Radiance .hdr with the per-scanline run-length encoding (runs of 4+ equal bytes as runs, the rest as literals)"""
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(header + b"-Y %d +X %d\n" % (h, w))
        for y in range(h):
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for ch in range(4):
                row = rgbe[y, :, ch]
                x = 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 4:
                        f.write(bytes([128 + run, int(row[x])]))
                        x += run
                    else:
                        lit = 1
                        while x + lit < w and lit < 128 and not (x + lit + 3 < w and row[x + lit] == row[x + lit + 1] == row[x + lit + 2] == row[x + lit + 3]):
                            lit += 1
                        f.write(bytes([lit]) + row[x: x + lit].tobytes())
                        x += lit


def write_flat_hdr(path, rgbe, header=b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n"):
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(header + b"-Y %d +X %d\n" % (h, w))
        f.write(np.ascontiguousarray(rgbe, np.uint8).tobytes())


def write_hdr_inputs(d):
    """Radiance .hdr inputs of loadProbe (main.cpp:146-156): flat and run-length encoded scanlines, widths on both sides of stb_image's RLE limits
    (8 and 32768), zero exponents, the full exponent range, long runs."""
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(99)
    out = []
    a = rng.integers(0, 256, (6, 40, 4), dtype=np.uint8)
    a[2, 5:30] = (10, 20, 30, 130)
    a[3, :, 3] = 0
    a[4, :, 3] = np.arange(40) * 6 + 8  # exponents 8 .. 242
    write_rle_hdr(os.path.join(d, "rle40.hdr"), a); out.append("rle40.hdr")
    write_flat_hdr(os.path.join(d, "flat40.hdr"), a); out.append("flat40.hdr")
    b = rng.integers(0, 256, (3, 7, 4), dtype=np.uint8)  # width < 8: never run-length encoded
    write_flat_hdr(os.path.join(d, "flat7.hdr"), b); out.append("flat7.hdr")
    c = rng.integers(100, 140, (5, 300, 4), dtype=np.uint8)
    c[:, 100:260, :3] = 200  # runs longer than 127: several run records
    write_rle_hdr(os.path.join(d, "rle300.hdr"), c, b"#?RADIANCE\nGAMMA=2.2\nPRIMARIES=0 0 0 0 0 0 0 0\nFORMAT=32-bit_rle_rgbe\n\n"); out.append("rle300.hdr")
    return out


BOXES = [
    (dict(), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)),
    (dict(color=(0.2, 0.4, 0.8), roughness=0.3, flags=scenes.MATERIAL_FLAG_SHADOW_CATCHER), (0.1, -2.7, 3.3), (0.7, 0.05, 12.9)),
    (dict(emission=(1, 2, 3), metallic=1.0), (1e6, 1e-6, -3.0), (1e-3, 1e3, 0.1)),
]


def pack(prefix, meshes, textures, G):
    G[prefix + "n"] = np.array([len(meshes), len(textures)], np.int32)
    for i, m in enumerate(meshes):
        for k in ("vertex", "normal", "texcoord", "index"):
            G[f"{prefix}m{i}_{k}"] = m[k]
        G[f"{prefix}m{i}_material"] = np.frombuffer(np.array(m["material"]).tobytes(), np.uint8).copy()
        G[f"{prefix}m{i}_tex"] = np.int32(m["diffuseTextureID"])
    for i, t in enumerate(textures):
        G[f"{prefix}t{i}"] = t


def main():
    R = orc.load_ref()
    if R is None or not hasattr(R, "refm_load_obj"):
        raise SystemExit("oracle/_ref/libptref.so lacks the Model.cpp shims: run `make -C oracle ref` where /root/reference exists")
    G = {}
    for name in write_inputs():
        out = orc.ref_load_obj(R, os.path.join(FIX, name))
        assert out is not None, name
        pack(name[:-4] + "_", out[0], out[1], G)
        print(name, len(out[0]), "meshes", len(out[1]), "textures")
    boxes = orc.ref_add_boxes(R, [(scenes.Material(**kw), p, e) for kw, p, e in BOXES])
    pack("boxes_", boxes, [], G)
    for name in write_hdr_inputs(os.path.join(HERE, "hdr_fixture")):
        img = orc.ref_loadf(R, os.path.join(HERE, "hdr_fixture", name))
        assert img is not None, name
        G["hdr_" + name[:-4]] = img
        print(name, img.shape)
    out = os.path.join(HERE, "ref_model.npz")
    np.savez_compressed(out, **G)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

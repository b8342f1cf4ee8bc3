"""loadProbe (HelloPathtracing_original/main.cpp:146-156): stbi_loadf(path, &w, &h, &n, 4) of a Radiance .hdr (RGBE)
environment map → ProbeData (float4 pixels, top row first, alpha 1).  stb_image is not vendored here; this is a
self-contained reader for the two encodings .hdr files use (flat RGBE and the "new" per-scanline RLE), with
stb_image's RGBE→float conversion: f = ldexp(1, e - (128 + 8)) and rgb = mantissa * f, all-zero for e == 0."""
from __future__ import annotations

import numpy as np

from .scenes import ProbeData


def _rgbe_to_float(rgbe: np.ndarray) -> np.ndarray:
    e = rgbe[..., 3].astype(np.int32)
    f = np.ldexp(np.float32(1.0), e - (128 + 8)).astype(np.float32)
    out = np.ones(rgbe.shape[:-1] + (4,), np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * f[..., None]
    out[e == 0, :3] = 0.0
    return out


def load_hdr(path: str) -> np.ndarray:
    """Returns (h, w, 4) float32, rows top to bottom (the -Y +X orientation every common .hdr uses)."""
    data = open(path, "rb").read()
    pos = 0
    if not (data.startswith(b"#?RADIANCE") or data.startswith(b"#?RGBE")):
        raise RuntimeError("not a Radiance .hdr file")
    fmt_ok = False
    while True:
        end = data.index(b"\n", pos)
        line = data[pos:end]
        pos = end + 1
        if line == b"":
            break
        if line.startswith(b"FORMAT=32-bit_rle_rgbe"):
            fmt_ok = True
    if not fmt_ok:
        raise RuntimeError("unsupported .hdr FORMAT")
    end = data.index(b"\n", pos)
    res = data[pos:end].split()
    pos = end + 1
    if len(res) != 4 or res[0] != b"-Y" or res[2] != b"+X":
        raise RuntimeError("unsupported .hdr orientation")
    h, w = int(res[1]), int(res[3])
    buf = np.frombuffer(data, np.uint8, offset=pos)
    img = np.zeros((h, w, 4), np.uint8)
    if w < 8 or w >= 32768 or not (buf[0] == 2 and buf[1] == 2 and not (buf[2] & 0x80)):
        img[:] = buf[: h * w * 4].reshape(h, w, 4)  # flat RGBE
    else:
        p = 0
        for y in range(h):
            if buf[p] != 2 or buf[p + 1] != 2 or ((int(buf[p + 2]) << 8) | int(buf[p + 3])) != w:
                raise RuntimeError("corrupt RLE scanline")
            p += 4
            for ch in range(4):
                x = 0
                while x < w:
                    c = int(buf[p]); p += 1
                    if c > 128:
                        c -= 128
                        img[y, x : x + c, ch] = buf[p]
                        p += 1
                    else:
                        img[y, x : x + c, ch] = buf[p : p + c]
                        p += c
                    x += c
    return _rgbe_to_float(img)


def save_hdr(path: str, rgb: np.ndarray) -> None:
    """Flat (uncompressed) RGBE writer — for tests and for exporting the procedural probes."""
    rgb = np.asarray(rgb, np.float32)[..., :3]
    h, w, _ = rgb.shape
    m = rgb.max(-1)
    mant, e = np.frexp(m)
    scale = np.where(m > 1e-32, mant * 256.0 / np.maximum(m, 1e-38), 0.0)
    out = np.zeros((h, w, 4), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, e + 128, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        f.write(out.tobytes())


def load_probe(path: str) -> ProbeData:
    d = load_hdr(path)
    return ProbeData(d.shape[1], d.shape[0], np.ascontiguousarray(d))

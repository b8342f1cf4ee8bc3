"""loadProbe restatement: RGBE decode (flat and RLE scanlines), stb_image's conversion, alpha = 1."""
import numpy as np

from optixpathtracer_amd import hdrloader, scenes


def _rle_file(path, rgbe):
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\n# comment\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1\n\n-Y %d +X %d\n" % (h, w))
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
                        lit = min(w - x, 64)
                        f.write(bytes([lit]) + row[x : x + lit].tobytes())
                        x += lit


def test_flat_and_rle_roundtrip(tmp_path):
    p = scenes.sky_probe(64, 32)
    f1 = str(tmp_path / "a.hdr")
    hdrloader.save_hdr(f1, p.data)
    a = hdrloader.load_hdr(f1)
    assert a.shape == (32, 64, 4) and (a[..., 3] == 1).all()
    assert np.allclose(a[..., :3], p.data[..., :3], rtol=1 / 100, atol=1e-3)  # 8-bit mantissas
    rng = np.random.default_rng(0)
    rgbe = rng.integers(0, 256, (8, 40, 4), dtype=np.uint8)
    rgbe[2, 5:30] = (10, 20, 30, 130)  # long runs
    rgbe[3, :, 3] = 0  # zero exponent → black
    f2 = str(tmp_path / "b.hdr")
    _rle_file(f2, rgbe)
    b = hdrloader.load_hdr(f2)
    ref = np.ones((8, 40, 4), np.float32)
    ref[..., :3] = rgbe[..., :3].astype(np.float32) * np.ldexp(np.float32(1), rgbe[..., 3].astype(np.int32) - 136)[..., None]
    ref[rgbe[..., 3] == 0, :3] = 0
    assert np.array_equal(b, ref)
    pd = hdrloader.load_probe(f2).BuildCDF()
    assert pd.valid and pd.width == 40 and pd.height == 8

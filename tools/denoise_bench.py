#!/usr/bin/env python3
"""pt_denoise timing at 1080p and 4K on the C3 scene: ms for n a-trous passes and achieved GB/s against the algorithmic
64 B per pixel per pass (3 x 16 B read + 16 B written)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optixpathtracer_amd import scenes
from optixpathtracer_amd import renderer as R

m = scenes.voxel_terrain(n=96, target_tris=70000)
probe = scenes.sky_probe(512, 256).BuildCDF()
for (w, h) in ((1920, 1080), (3840, 2160)):
    r = R.SampleRenderer(m)
    r.setProbe(probe)
    r.resize((w, h))
    r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
    r.launchParams.samples_per_launch = 1
    r.render()
    for it in (1, 5):
        best = 1e9
        for _ in range(5):
            _, ms = r.denoise(iterations=it)
            best = min(best, ms)
        gb = w * h * 64 * it / best / 1e6
        print(f"{w}x{h} iterations {it}: {best:.3f} ms  {gb:.0f} GB/s algorithmic ({gb / 8000 * 100:.1f} % of 8 TB/s)", flush=True)

#!/usr/bin/env python3
"""Launch floor of the traversal kernel: mean kernel time of pt_trace for n diffuse-bounce rays of the C3 scene, n = 64 ... 1 M
(closest hit and any hit).  The slope is the bulk rate, the intercept the per-launch floor that nine launches per chunk pay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optixpathtracer_amd import scenes
from optixpathtracer_amd import renderer as R

m = scenes.voxel_terrain()
rng = np.random.default_rng(0)
w, h = 1280, 720
U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=w / h)
ys, xs = np.mgrid[0:h, 0:w]
d = (2 * (xs + 0.5) / w - 1)[..., None] * U + (2 * (ys + 0.5) / h - 1)[..., None] * V + W
d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).reshape(-1, 3).astype(np.float32)
n = len(d)
prim = np.concatenate([np.tile(np.array(scenes.TERRAIN_CAMERA["eye"], np.float32), (n, 1)), np.full((n, 1), 1e-3, np.float32), d, np.full((n, 1), 1e16, np.float32)], 1).astype(np.float32)
r = R.SampleRenderer(m)
(t, p), _ = r.trace(prim)
hit = p >= 0
P = prim[hit, :3] + t[hit, None] * prim[hit, 4:7]
k = len(P)
dd = rng.standard_normal((k, 3)).astype(np.float32)
dd /= np.linalg.norm(dd, axis=1, keepdims=True)
dd[:, 1] = np.abs(dd[:, 1])
bounce = np.concatenate([P, np.full((k, 1), 1e-2, np.float32), dd, np.full((k, 1), 1e16, np.float32)], 1).astype(np.float32)
print("rays available", k, flush=True)
for any_hit in (False, True):
    for cnt in (64, 512, 4096, 26000, 100000, 400000, min(k, 800000)):
        sub = bounce[:: max(1, k // cnt)][:cnt]  # spread over the image like a late bounce
        _, ms = r.trace(sub, any_hit=any_hit, iters=20)
        print(("any-hit " if any_hit else "closest ") + f"n={len(sub):7d}  {ms * 1e3:8.1f} us  {len(sub) / ms / 1e3:8.1f} Mrays/s", flush=True)

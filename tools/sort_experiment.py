#!/usr/bin/env python3
"""Does ray reordering pay?  Traces second/third-bounce-like ray sets through pt_trace in path order, in random
order and sorted by (direction octant, origin Morton cell); kernel time only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optixpathtracer_amd import scenes
from optixpathtracer_amd import renderer as R

m = scenes.voxel_terrain()
rng = np.random.default_rng(0)
w, h = 1920, 1080
U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=w / h)
ys, xs = np.mgrid[0:h, 0:w]
# 8x8 block order like the renderer
by, bx = ys // 8, xs // 8
order = np.lexsort(((xs % 8).ravel(), (ys % 8).ravel(), bx.ravel(), by.ravel()))
dx = 2 * (xs + 0.5) / w - 1; dy = 2 * (ys + 0.5) / h - 1
d = dx[..., None] * U + dy[..., None] * V + W
d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).reshape(-1, 3).astype(np.float32)[order]
n = len(d)
eye = np.array(scenes.TERRAIN_CAMERA["eye"], np.float32)
def mk(o, d, tmin): return np.concatenate([o, np.full((len(o), 1), tmin, np.float32), d, np.full((len(o), 1), 1e16, np.float32)], 1).astype(np.float32)
r = R.SampleRenderer(m)
rays = mk(np.tile(eye, (n, 1)), d, 1e-3)
for bounce in range(3):
    (t, p), ms = r.trace(rays, iters=3)
    print(f"bounce {bounce}: {len(rays)} rays path order {ms:.3f} ms  {len(rays)/ms/1e3:.0f} Mrays/s", flush=True)
    if bounce > 0:
        perm = rng.permutation(len(rays))
        _, ms_r = r.trace(rays[perm], iters=3)
        o = rays[:, :3]; dd = rays[:, 4:7]
        cell = np.clip(((o + 100.0) / 200.0 * 32).astype(np.int64), 0, 31)
        def part(v):
            v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
        mort = (part(cell[:, 0]) << 2) | (part(cell[:, 1]) << 1) | part(cell[:, 2])
        octant = (dd[:, 0] < 0).astype(np.int64) | ((dd[:, 1] < 0).astype(np.int64) << 1) | ((dd[:, 2] < 0).astype(np.int64) << 2)
        for name, key in (("origin-morton32", mort), ("octant|origin", (octant << 15) | mort), ("origin|octant", (mort << 3) | octant)):
            perm = np.argsort(key, kind="stable")
            _, ms_s = r.trace(rays[perm], iters=3)
            print(f"   sorted by {name}: {ms_s:.3f} ms ({ms/ms_s:.2f}x vs path order; random order {ms_r:.3f} ms)", flush=True)
    hit = p >= 0
    P = rays[hit, :3] + t[hit, None] * rays[hit, 4:7]
    k = len(P)
    dd = rng.standard_normal((k, 3)).astype(np.float32); dd /= np.linalg.norm(dd, axis=1, keepdims=True); dd[:, 1] = np.abs(dd[:, 1])
    rays = mk(P, dd, 1e-3)

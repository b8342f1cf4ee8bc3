#!/usr/bin/env python3
"""Traversal micro-benchmark through pt_trace: primary, diffuse-bounce and sun-shadow ray sets on the C3 scene,
for both tree kinds; with PT_DEBUG_COUNTS=1 prints node steps / triangle tests per ray."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optixpathtracer_amd import scenes
from optixpathtracer_amd import renderer as R

m = scenes.voxel_terrain()
rng = np.random.default_rng(0)
w, h = 960, 540
U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=w / h)
ys, xs = np.mgrid[0:h, 0:w]
dx = 2 * (xs + 0.5) / w - 1; dy = 2 * (ys + 0.5) / h - 1
d = dx[..., None] * U + dy[..., None] * V + W
d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).reshape(-1, 3).astype(np.float32)
n = len(d)
prim = np.concatenate([np.tile(np.array(scenes.TERRAIN_CAMERA["eye"], np.float32), (n, 1)), np.full((n, 1), 1e-3, np.float32), d, np.full((n, 1), 1e16, np.float32)], 1).astype(np.float32)
import os
for bk in ("lbvh", "ploc"):
    os.environ["PT_BVH_BUILDER"] = bk
    r = R.SampleRenderer(m)
    print("hierarchy", bk, flush=True)
    (t, p), ms = r.trace(prim, iters=3); print("  primary   ", n, "rays", round(ms, 3), "ms", round(n / ms / 1e3, 1), "Mrays/s  hit", round(float((p >= 0).mean()), 3), flush=True)
    hit = p >= 0
    P = prim[hit, :3] + t[hit, None] * prim[hit, 4:7]
    k = len(P)
    dd = rng.standard_normal((k, 3)).astype(np.float32); dd /= np.linalg.norm(dd, axis=1, keepdims=True); dd[:, 1] = np.abs(dd[:, 1])
    bounce = np.concatenate([P, np.full((k, 1), 1e-3, np.float32), dd, np.full((k, 1), 1e16, np.float32)], 1).astype(np.float32)
    (t2, p2), ms = r.trace(bounce, iters=3); print("  bounce    ", k, "rays", round(ms, 3), "ms", round(k / ms / 1e3, 1), "Mrays/s  hit", round(float((p2 >= 0).mean()), 3), flush=True)
    sun = np.tile(scenes.SUN_DIR.astype(np.float32), (k, 1)) + rng.normal(0, 0.01, (k, 3)).astype(np.float32)
    sh = np.concatenate([P, np.full((k, 1), 1e-2, np.float32), sun, np.full((k, 1), 1e16, np.float32)], 1).astype(np.float32)
    occ, ms = r.trace(sh, any_hit=True, iters=3); print("  sun shadow", k, "rays", round(ms, 3), "ms", round(k / ms / 1e3, 1), "Mrays/s  occluded", round(float(occ.mean()), 3), flush=True)
    (t3, p3), ms = r.trace(sh, iters=3); print("  sun closest", k, "rays", round(ms, 3), "ms", round(k / ms / 1e3, 1), "Mrays/s  hit", round(float((p3 >= 0).mean()), 3), flush=True)

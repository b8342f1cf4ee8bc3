#!/usr/bin/env python3
"""tools/r4_split_probe.py [budget]: what could reference splitting buy on the stadium scene?  An upper-bound probe with the shipped builder:
needle triangles are really subdivided (longest-edge bisection, 2^k pieces, pieces chosen by sqrt(AABB area - ideal area) under a budget of
extra triangles) and the subdivided scene is rendered: Mrays/s, node steps and triangle tests per ray against the original."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optixpathtracer_amd import renderer as R  # noqa: E402
from optixpathtracer_amd import scenes  # noqa: E402


def bisect(tri, levels):
    """tri (m,3,3) -> (m * 2^levels, 3, 3) by repeated longest-edge bisection"""
    for _ in range(levels):
        e = np.stack([np.linalg.norm(tri[:, (k + 1) % 3] - tri[:, k], axis=1) for k in range(3)], 1)
        k = e.argmax(1)
        idx = np.arange(len(tri))
        a, b, c = tri[idx, k], tri[idx, (k + 1) % 3], tri[idx, (k + 2) % 3]
        m = 0.5 * (a + b)
        tri = np.concatenate([np.stack([a, m, c], 1), np.stack([m, b, c], 1)])
    return tri


def split_model(model, budget=0.3, cap=4):
    out = scenes.Model()
    allE = []
    per = []
    for mesh in model.meshes:
        t = mesh.vertex[mesh.index].astype(np.float64)
        lo, hi = t.min(1), t.max(1)
        d = hi - lo
        A = 2 * (d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0])
        n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
        ideal = np.abs(n).sum(1)
        E = np.sqrt(np.maximum(A - ideal, 0))
        allE.append(E)
        per.append(t)
    tot = sum(e.sum() for e in allE)
    ntri = sum(len(e) for e in allE)
    tau = tot / (budget * ntri)
    extra = 0
    for mesh, t, E in zip(model.meshes, per, allE):
        s = np.minimum(1 + np.floor(E / tau), 2**cap)
        lv = np.floor(np.log2(s)).astype(int)
        parts = []
        for L in range(cap + 1):
            sel = t[lv == L]
            if len(sel):
                parts.append(bisect(sel, L))
        tt = np.concatenate(parts).astype(np.float32)
        extra += len(tt) - len(t)
        out.meshes.append(scenes.TriangleMesh(tt.reshape(-1, 3).copy(), np.arange(3 * len(tt), dtype=np.uint32).reshape(-1, 3), mesh.material))
    print(f"split: {ntri} -> {ntri + extra} triangles (+{100 * extra / ntri:.1f} %)")
    return out


def run(model, tag):
    probe = scenes.sky_probe(2048, 1024).BuildCDF()
    w, h = 1920, 1080
    r = R.SampleRenderer(model)
    r.setProbe(probe)
    r.resize((w, h))
    r.setCamera(R.make_camera(scenes.STADIUM_CAMERA, w / h))
    r.launchParams.samples_per_launch = 4
    for k in range(3):
        r.launchParams.frame.subframe_index = k
        r.render()
    s0 = r.stats()
    t0 = time.perf_counter()
    for k in range(10):
        r.launchParams.frame.subframe_index = 3 + k
        r.render()
    dt = time.perf_counter() - t0
    s1 = r.stats()
    rays = (s1["total_radiance_rays"] + s1["total_shadow_rays"]) - (s0["total_radiance_rays"] + s0["total_shadow_rays"])
    print(f"{tag}: {dt / 10 * 1e3:.2f} ms/frame, {rays / dt / 1e6:.0f} Mrays/s, {rays // 10} rays/frame, bvh {s1['bvh_nodes']} nodes {s1['bvh_levels']} levels builder {s1['bvh_builder']}")
    r.close()


if __name__ == "__main__":
    m = scenes.stadium_scene()
    run(m, "original")
    for b in [float(x) for x in sys.argv[1:]] or [0.3]:
        run(split_model(m, b), f"budget {b}")

"""Which rays get different hits from the LBVH- and the PLOC-built tree of the 1 M-triangle stadium, and what does brute force say?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optixpathtracer_amd import renderer as R, scenes

m = scenes.stadium_scene()
v, idx, _, _ = m.flatten()
tri = v[idx].astype(np.float64)
cam = scenes.STADIUM_CAMERA
w, h = 1920, 1080
U, V, W = scenes.uvw_frame(**cam, aspect=w / h)
rng = np.random.default_rng(5)
n = 3_000_000
x = rng.uniform(-1, 1, n).astype(np.float32); y = rng.uniform(-1, 1, n).astype(np.float32)
d = x[:, None] * U[None] + y[:, None] * V[None] + W[None]
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((n, 8), np.float32)
rays[:, :3] = np.asarray(cam["eye"], np.float32); rays[:, 3] = 1e-3; rays[:, 4:7] = d; rays[:, 7] = 1e16
res = {}
for b in ("lbvh", "ploc"):
    os.environ["PT_BVH_BUILDER"] = b
    r = R.SampleRenderer(m)
    res[b] = r.trace(rays)[0]
    r.close()
(t0, p0), (t1, p1) = res["lbvh"], res["ploc"]
bad = np.nonzero((p0 != p1) | (t0.view(np.uint32) != t1.view(np.uint32)))[0]
print("rays", n, "differing", len(bad))
for i in bad[:12]:
    o = rays[i, :3].astype(np.float64); dd = rays[i, 4:7].astype(np.float64)
    # double-precision Moeller-Trumbore over all triangles
    e1 = tri[:, 1] - tri[:, 0]; e2 = tri[:, 2] - tri[:, 0]
    pv = np.cross(dd, e2); det = (e1 * pv).sum(1)
    ok = np.abs(det) > 0
    inv = np.where(ok, 1.0 / np.where(ok, det, 1), 0)
    tv = o - tri[:, 0]; u = (tv * pv).sum(1) * inv
    qv = np.cross(tv, e1); vv = (qv * dd).sum(1) * inv; tt = (qv * e2).sum(1) * inv
    hit = ok & (u >= -1e-9) & (vv >= -1e-9) & (u + vv <= 1 + 1e-9) & (tt > 1e-3)
    cand = np.nonzero(hit)[0]
    order = cand[np.argsort(tt[cand])][:4]
    print(i, "lbvh", (float(t0[i]), int(p0[i])), "ploc", (float(t1[i]), int(p1[i])), "double-precision nearest:", [(float(tt[k]), int(k), float(u[k]), float(vv[k])) for k in order])

"""Why does a ray miss in the tiny_in_huge / collinear cases?  Walks the exported 8-wide tree towards the expected primitive in float64."""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_parity as T
from optixpathtracer_amd import scenes
from optixpathtracer_amd.renderer import SampleRenderer
from oracle import orc
O = orc.Oracle('det')
rng = np.random.default_rng(5)
c = rng.uniform(-5000, 5000, (4000, 1, 3))
tri = (c + rng.standard_normal((4000, 3, 3)) * 1e-2).astype(np.float32)
tri[:200] = (rng.uniform(-1, 1, (200, 1, 3)) + rng.standard_normal((200, 3, 3)) * 0.3).astype(np.float32)
m = scenes.Model(meshes=[scenes.TriangleMesh(vertex=tri.reshape(-1, 3).copy(), index=np.arange(3 * len(tri), dtype=np.uint32).reshape(-1, 3), material=scenes.Material())])
sc = O.make_scene(m, use_bvh=False)
scb = O.make_scene(m, use_bvh=True)
lo, hi = float(tri.min()) - 5, float(tri.max()) + 5
rays = T._random_rays(rng, 20000, lo, hi)
to, po = O.trace_closest(sc, rays)
tb, pb = O.trace_closest(scb, rays)
print("checker brute force vs checker's own BVH2: prim diffs", int((po != pb).sum()), "hits", int((po >= 0).sum()))
r = SampleRenderer(m)
(t, prim), _ = r.trace(rays)
bad = np.nonzero(prim != po)[0]
print("gpu vs brute force diffs", len(bad), " gpu vs checker BVH2 diffs", int((prim != pb).sum()))
nodes, tris = r.exportBVH()
O.set_bvh8(sc, nodes, tris)
t8, p8 = O.trace_closest(sc, rays)
print("cpu walk of the exported tree vs gpu diffs", int((p8 != prim).sum()), " vs brute force", int((p8 != po).sum()))
primid = tris[:, 9].view(np.int32)
def decode(n):
    w = nodes[n]
    org = w[0:3].view(np.float32).astype(np.float64)
    sx = np.array([w[3] << 16], np.uint32).view(np.float32)[0]; sy = np.array([w[3] & 0xffff0000], np.uint32).view(np.float32)[0]; sz = np.array([w[7] << 16], np.uint32).view(np.float32)[0]
    imask = int(w[7] >> 16); cb = int(w[4]); tbase = int(w[5]); leafbits = int(w[6])
    q = w[8:20].view(np.uint8).reshape(6, 8).astype(np.float64)
    step = np.array([sx, sy, sz], np.float64)
    blo = org[None, :] + q[0:3].T * step[None, :]
    bhi = org[None, :] + q[3:6].T * step[None, :]
    return blo, bhi, imask, cb, tbase, leafbits, step
# parent map
parent = {}
leaf_of = {}
for n in range(len(nodes)):
    blo, bhi, imask, cb, tbase, leafbits, step = decode(n)
    for s in range(8):
        if imask >> s & 1:
            parent[cb + bin(imask & ((1 << s) - 1)).count('1')] = (n, s)
        for k in range(3):
            if leafbits >> (3 * s + k) & 1:
                leaf_of[tbase + bin(leafbits & ((1 << (3 * s + k)) - 1)).count('1')] = (n, s)
for i in bad[:4]:
    P = po[i]
    ti = int(np.nonzero(primid == P)[0][0])
    n, s = leaf_of[ti]
    chain = [(n, s)]
    while n in parent:
        n, s = parent[n]; chain.append((n, s))
    o = rays[i, :3].astype(np.float64); d = rays[i, 4:7].astype(np.float64)
    print(f"ray {i}: o {o} d {d} expected prim {P} t {to[i]}; triangle verts {tris[ti,:9]}")
    hitp = o + d * float(to[i])
    for n, s in reversed(chain):
        blo, bhi, imask, cb, tbase, leafbits, step = decode(n)
        l, h = blo[s], bhi[s]
        with np.errstate(divide='ignore'):
            t1 = (l - o) / d; t2 = (h - o) / d
        tn = np.minimum(t1, t2).max(); tf = np.maximum(t1, t2).min()
        inside = np.all(hitp >= l) and np.all(hitp <= h)
        tv = tris[ti, :9].reshape(3, 3).astype(np.float64)
        contains = np.all(tv >= l - 0) and np.all(tv <= h + 0)
        print(f"   node {n} slot {s}: step {step} box lo {l} hi {h} extent {h-l}  f64 tnear {tn:.6f} tfar {tf:.6f} {'HIT' if tn <= tf else 'MISS'}  hit point inside box: {inside}  triangle inside box: {contains}")

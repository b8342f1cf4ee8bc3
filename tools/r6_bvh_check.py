#!/usr/bin/env python3
"""tools/r6_bvh_check.py MODE [SCENE...] — run in a process of its own with PT_LIB pointing at a library variant (tests/test_gpu_builder.py,
tools/r6_sah.sh):
  build    builds every scene with the binned-SAH hierarchy forced and by default (LBVH against SAH by calibration), checks that the wide
           tree holds every primitive exactly once, prints `ok SCENE builder ms`; meant for the bounds-checked library
           (variants/libptamd_chk.so, -DPT_BVH_CHECK=1) with PT_BVH_GUARD=1
  inject   expects pt_create to FAIL cleanly (PT_BVH_INJECT=1|2 set by the caller, SAH forced): prints `refused: <message>`
  export   prints sha256 of the canonical (numbering-free) SAH tree of every scene and the build times: two libraries that build the same
           hierarchy print the same hashes"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from optixpathtracer_amd import renderer as R, scenes  # noqa: E402


def scene(name):
    if name == "terrain70k":
        return scenes.voxel_terrain(n=96, target_tris=70000)
    if name == "terrain1M":
        return scenes.voxel_terrain()
    if name == "stadium200k":
        return scenes.stadium_scene(target_tris=200_000)
    if name == "stadium1M":
        return scenes.stadium_scene()
    if name == "copies":
        base = np.array([[0, 0, 0], [4, 0, 0], [0, 3, 0]], np.float32)
        tri = np.repeat(base[None], 6000, 0)
        return scenes.Model(meshes=[scenes.TriangleMesh(vertex=tri.reshape(-1, 3).copy(), index=np.arange(18000, dtype=np.uint32).reshape(-1, 3), material=scenes.Material())])
    if name == "line":  # 40 000 small triangles along one axis: deep, unbalanced splits, coincident bins on two axes
        t = np.array([[0, 0, 0], [0.2, 0, 0], [0, 0.2, 0]], np.float32)[None] + np.linspace(-500, 500, 40000, dtype=np.float32)[:, None, None] * np.array([1, 0, 0], np.float32)
        return scenes.Model(meshes=[scenes.TriangleMesh(vertex=t.reshape(-1, 3).copy(), index=np.arange(120000, dtype=np.uint32).reshape(-1, 3), material=scenes.Material())])
    if name == "cornell":
        return scenes.cornell_box()
    raise SystemExit("unknown scene " + name)


def canonical(nodes, tris):
    N = np.frombuffer(nodes, np.uint32).reshape(-1, 20)
    T = np.frombuffer(tris, np.uint32).reshape(-1, 12)
    h = hashlib.sha256()
    stack, count = [0], 0
    while stack:
        i = stack.pop()
        nd = N[i]
        child_base, tri_base, leafbits, imask = int(nd[4]), int(nd[5]), int(nd[6]), int(nd[7]) >> 16
        h.update(np.concatenate([nd[:4], nd[6:]]).tobytes())
        h.update(T[tri_base:tri_base + bin(leafbits).count("1")].tobytes())
        stack.extend(reversed([child_base + k for k in range(bin(imask).count("1"))]))
        count += 1
    return h.hexdigest(), count


def main():
    mode = sys.argv[1]
    names = sys.argv[2:] or ["terrain70k", "stadium200k", "copies", "line", "cornell"]
    for name in names:
        m = scene(name)
        n = m.num_triangles
        for builder in (("sah",) if mode != "build" else ("sah", None)):
            if builder:
                os.environ["PT_BVH_BUILDER"] = builder
            else:
                os.environ.pop("PT_BVH_BUILDER", None)
            try:
                r = R.SampleRenderer(m)
            except Exception as e:  # pt_create refused the build
                if mode == "inject":
                    print("refused:", str(e).replace("\n", " ")[:200])
                    continue
                raise
            if mode == "inject":
                raise SystemExit(f"{name}: the build was expected to be refused")
            nodes, tris = (np.asarray(x).tobytes() for x in r.exportBVH()[:2])
            st = r.stats()
            prims = np.sort(np.frombuffer(tris, np.uint32).reshape(-1, 12)[:, 9])
            assert len(prims) == n and np.array_equal(prims, np.arange(n, dtype=np.uint32)), name
            digest, walked = canonical(nodes, tris)
            assert st["bvh_challengers_skipped"] == 0, st
            r.close()
            if mode == "export":
                r2 = R.SampleRenderer(m)  # the second build of a process is the warm one
                ms2 = r2.stats()["bvh_build_ms"]
                r2.close()
                print(f"{name} {digest} nodes {walked} first_build_ms {st['bvh_build_ms']:.2f} second_build_ms {ms2:.2f}")
            else:
                print(f"ok {name} {builder or 'default'} builder {st['bvh_builder']} {st['bvh_build_ms']:.2f} ms")


if __name__ == "__main__":
    main()

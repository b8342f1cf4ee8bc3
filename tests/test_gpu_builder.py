"""The on-GPU builder (row a16, replaces optixAccelBuild + optixAccelCompact, SimplePathtracer.cpp:561-591): the level-synchronous
refit / collapse-cost passes must produce the very tree the atomic climb produced, and the build has a time budget."""
import os

import numpy as np
import pytest

from optixpathtracer_amd import scenes

pytestmark = pytest.mark.gpu


def _canonical(nodes_bytes, tris_bytes):
    """The exported tree without its node numbering: k_collapse8 hands out child and triangle ranges with atomic counters, so the order
    of the nodes inside a level (and of the leaf-triangle groups) differs from build to build.  Walk from the root in slot order and
    emit every node's content (grid, masks, planes) and its leaf triangles: equal trees give equal bytes."""
    N = np.frombuffer(nodes_bytes, np.uint32).reshape(-1, 20)
    T = np.frombuffer(tris_bytes, np.uint32).reshape(-1, 12)
    out_nodes, out_tris = [], []
    stack = [0]
    while stack:
        i = stack.pop()
        nd = N[i]
        child_base, tri_base, leafbits, imask = int(nd[4]), int(nd[5]), int(nd[6]), int(nd[7]) >> 16
        out_nodes.append(np.concatenate([nd[:4], nd[6:]]))
        ntri = bin(leafbits).count("1")
        out_tris.append(T[tri_base:tri_base + ntri])
        kids = [child_base + k for k in range(bin(imask).count("1"))]
        stack.extend(reversed(kids))
    return np.concatenate(out_nodes).tobytes(), np.concatenate(out_tris).tobytes(), len(out_nodes)


def _export(model, monkeypatch, **env):
    from optixpathtracer_amd.renderer import SampleRenderer

    for k in ("PT_BVH_CLIMB", "PT_BVH_BUILDER", "PT_PLOC_TAIL"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = SampleRenderer(model)
    nodes, tris = r.exportBVH()[:2]
    st = r.stats()
    r.close()
    return np.asarray(nodes).tobytes(), np.asarray(tris).tobytes(), st


@pytest.mark.parametrize("scene", ["terrain1M", "stadium200k", "copies", "small"])
def test_level_passes_reproduce_the_climb(ptlib, monkeypatch, scene):
    """PT_BVH_CLIMB=1 runs the bottom-up passes the way rounds 1-3 did (one thread per leaf climbing with atomic counters and agent-scope
    fences); the default runs them level by level.  Nodes and leaf triangles of the wide tree must be identical up to the numbering of the nodes, for both
    hierarchies (the calibration then picks the same one)."""
    if scene == "terrain1M":
        m = scenes.voxel_terrain()
    elif scene == "stadium200k":
        m = scenes.stadium_scene(target_tris=200_000)
    elif scene == "copies":  # thousands of copies of one triangle: the deepest hierarchies the builder meets (Morton ties broken by primitive id)
        base = np.array([[0, 0, 0], [4, 0, 0], [0, 3, 0]], np.float32)
        tri = np.repeat(base[None], 6000, 0)
        m = scenes.Model(meshes=[scenes.TriangleMesh(vertex=tri.reshape(-1, 3).copy(), index=np.arange(18000, dtype=np.uint32).reshape(-1, 3), material=scenes.Material())])
    else:
        m = scenes.cornell_box()
    for builder in ("lbvh", "ploc", None):
        env = {} if builder is None else {"PT_BVH_BUILDER": builder}
        a = _export(m, monkeypatch, **env)
        b = _export(m, monkeypatch, PT_BVH_CLIMB="1", **env)
        ca, cb = _canonical(a[0], a[1]), _canonical(b[0], b[1])
        assert len(a[0]) == len(b[0]) and len(a[1]) == len(b[1]) and ca[2] * 80 == len(a[0])
        assert ca[0] == cb[0] and ca[1] == cb[1], f"{scene}/{builder}: the level-synchronous build differs from the climb"
        assert a[2]["bvh_builder"] == b[2]["bvh_builder"] and a[2]["bvh_levels"] == b[2]["bvh_levels"]


@pytest.mark.parametrize("scene", ["terrain70k", "stadium200k", "copies"])
def test_ploc_tail_in_one_workgroup_builds_the_same_hierarchy(ptlib, monkeypatch, scene):
    """The last rounds of the PLOC clustering run in one workgroup (k_ploc_tail) instead of five launches and a host wait per round:
    same search, same acceptance rule, same node numbering — the wide tree over it must be the same as with PT_PLOC_TAIL=0."""
    if scene == "terrain70k":
        m = scenes.voxel_terrain(n=96, target_tris=70000)
    elif scene == "stadium200k":
        m = scenes.stadium_scene(target_tris=200_000)
    else:
        base = np.array([[0, 0, 0], [4, 0, 0], [0, 3, 0]], np.float32)
        tri = np.repeat(base[None], 6000, 0)
        m = scenes.Model(meshes=[scenes.TriangleMesh(vertex=tri.reshape(-1, 3).copy(), index=np.arange(18000, dtype=np.uint32).reshape(-1, 3), material=scenes.Material())])
    a = _export(m, monkeypatch, PT_BVH_BUILDER="ploc")
    b = _export(m, monkeypatch, PT_BVH_BUILDER="ploc", PT_PLOC_TAIL="0")
    ca, cb = _canonical(a[0], a[1]), _canonical(b[0], b[1])
    assert ca[0] == cb[0] and ca[1] == cb[1]
    assert a[2]["bvh_builder"] == 1


@pytest.mark.parametrize("scene", ["terrain70k", "stadium200k", "copies", "cornell", "five"])
def test_sah_hierarchy_holds_every_triangle_once(ptlib, monkeypatch, scene):
    """The binned-SAH hierarchy (round 5, pt_bvh_build.hip build_sah): level-synchronous splits of the large nodes, one thread per subtree of at most
    eight leaves; coincident centroids (the `copies` scene: 6000 copies of one triangle) fall back to halving by position.  Whatever the shape, the
    wide tree over it must hold every primitive exactly once, and a render through it must equal the LBVH's bit for bit."""
    from optixpathtracer_amd import renderer as R

    cam = scenes.TERRAIN_CAMERA
    if scene == "terrain70k":
        m = scenes.voxel_terrain(n=96, target_tris=70000)
    elif scene == "stadium200k":
        m, cam = scenes.stadium_scene(target_tris=200_000), scenes.STADIUM_CAMERA
    elif scene == "copies":
        base = np.array([[0, 0, 0], [4, 0, 0], [0, 3, 0]], np.float32)
        tri = np.repeat(base[None], 6000, 0)
        m = scenes.Model(meshes=[scenes.TriangleMesh(vertex=tri.reshape(-1, 3).copy(), index=np.arange(18000, dtype=np.uint32).reshape(-1, 3), material=scenes.Material())])
        cam = dict(eye=(1.5, 1.0, 6.0), lookat=(1.5, 1.0, 0.0), up=(0.0, 1.0, 0.0), fovY=50.0)
    elif scene == "cornell":
        m, cam = scenes.cornell_box(), scenes.CORNELL_CAMERA  # 32 triangles: one level of large nodes, then small subtrees
    else:
        rng = np.random.default_rng(4)  # five triangles: the whole hierarchy is one small subtree
        v = rng.standard_normal((15, 3)).astype(np.float32)
        m = scenes.Model(meshes=[scenes.TriangleMesh(vertex=v, index=np.arange(15, dtype=np.uint32).reshape(-1, 3), material=scenes.Material())])
        cam = dict(eye=(0.0, 0.0, 6.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fovY=50.0)
    n = m.num_triangles
    probe = scenes.sky_probe(128, 64).BuildCDF()
    frames = {}
    for builder in ("sah", "lbvh"):
        for k in ("PT_BVH_CLIMB", "PT_PLOC_TAIL"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("PT_BVH_BUILDER", builder)
        r = R.SampleRenderer(m)
        if builder == "sah":
            nodes, tris = r.exportBVH()[:2]
            assert r.stats()["bvh_builder"] == 3
            prims = np.sort(np.frombuffer(np.asarray(tris).tobytes(), np.uint32).reshape(-1, 12)[:, 9])
            assert len(prims) == n and np.array_equal(prims, np.arange(n, dtype=np.uint32))
            walked = _canonical(np.asarray(nodes).tobytes(), np.asarray(tris).tobytes())  # every leaf triangle is reachable from the root
            assert len(walked[1]) == n * 48
        r.setProbe(probe)
        r.resize((160, 96))
        r.setCamera(R.make_camera(cam, 160 / 96))
        r.launchParams.samples_per_launch = 2
        r.render()
        frames[builder] = r.download(R.PT_BUF_ACCUM).copy()
        r.close()
    assert np.array_equal(frames["sah"].view(np.uint32), frames["lbvh"].view(np.uint32))


def test_build_time_budget(ptlib, monkeypatch):
    """pt_stats.bvh_build_ms for a million triangles (the LBVH and the binned-SAH hierarchy, a wide tree over each, calibration): rounds 1-3 took
    43-60 ms, round 4 (LBVH | PLOC) 13.8 / 16.9 ms."""
    for k in ("PT_BVH_CLIMB", "PT_BVH_BUILDER"):
        monkeypatch.delenv(k, raising=False)
    from optixpathtracer_amd.renderer import SampleRenderer

    out = {}
    for name, m in (("terrain", scenes.voxel_terrain()), ("stadium", scenes.stadium_scene())):
        best = 1e9
        for _ in range(3):
            r = SampleRenderer(m)
            best = min(best, r.stats()["bvh_build_ms"])
            r.close()
        out[name] = best
    print(f"\n[bvh build, 1 M triangles] terrain {out['terrain']:.1f} ms, stadium {out['stadium']:.1f} ms")
    assert out["terrain"] < 25 and out["stadium"] < 25, out


def test_imported_hierarchy_is_validated(ptlib, monkeypatch, tmp_path):
    """PT_BVH_IMPORT (experiment hook: a binary hierarchy built elsewhere) reads an untrusted file.  A well-formed tree — here a right comb,
    the worst shape there is — must give the image the GPU builders give (ray search is defined by geometry); a file
    that names a node twice, leaves one unreferenced, refers to the root or to an id out of range must be refused by pt_create
    (the level-synchronous passes read parent[] of every internal node)."""
    from optixpathtracer_amd import renderer as R

    m = scenes.cornell_box()
    n = sum(len(mesh.index) for mesh in m.meshes)
    probe = scenes.sky_probe(64, 32).BuildCDF()

    def render():
        r = R.SampleRenderer(m)
        r.setProbe(probe)
        r.resize((96, 64))
        r.setCamera(R.make_camera(scenes.CORNELL_CAMERA, 96 / 64))
        r.launchParams.samples_per_launch = 2
        r.render()
        a = r.download(R.PT_BUF_ACCUM).copy()
        r.close()
        return a

    monkeypatch.delenv("PT_BVH_IMPORT", raising=False)
    ref = render()
    comb = np.empty((n - 1, 2), np.int32)  # node j = (primitive j, node j + 1); the last node holds the last two primitives
    for j in range(n - 1):
        comb[j] = (~j, j + 1)
    comb[n - 2] = (~(n - 2), ~(n - 1))

    def write(tree, name):
        p = tmp_path / name
        with open(p, "wb") as f:
            f.write(np.int32(n).tobytes())
            f.write(np.ascontiguousarray(tree, np.int32).tobytes())
        return str(p)

    monkeypatch.setenv("PT_BVH_IMPORT", write(comb, "comb.tree"))
    assert np.array_equal(render().view(np.uint32), ref.view(np.uint32))
    bad = {}
    t = comb.copy(); t[3, 1] = 5; bad["node 5 named twice, node 4 never"] = t
    t = comb.copy(); t[2, 0] = ~0; bad["primitive 0 named twice, primitive 2 never"] = t
    t = comb.copy(); t[n - 2, 1] = 0; bad["the root as a child"] = t
    t = comb.copy(); t[1, 1] = n - 1; bad["node id out of range"] = t
    t = comb.copy(); t[1, 0] = ~n; bad["primitive out of range"] = t
    t = comb.copy(); t[4, 1] = 2; bad["a child with a smaller id than its parent"] = t
    for what, tree in bad.items():
        monkeypatch.setenv("PT_BVH_IMPORT", write(tree, "bad.tree"))
        with pytest.raises(RuntimeError):
            R.SampleRenderer(m)


def _run_tool(args, env, timeout=600):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("PT_BVH_CLIMB", "PT_BVH_BUILDER", "PT_PLOC_TAIL", "PT_BVH_INJECT", "PT_BVH_GUARD", "PT_LIB"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(root, "tools", "r6_bvh_check.py")] + args, capture_output=True, text=True, timeout=timeout, env=e)


def _chk_lib():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "optixpathtracer_amd", "variants", "libptamd_chk.so")
    assert os.path.exists(lib), "variants/libptamd_chk.so is missing: __graft_entry__.build() (or tools/variants.sh chk \"-DPT_BVH_CHECK=1\") builds it"
    return lib


def test_bounds_checked_builder_on_every_shape(ptlib):
    """the SAH kernels compiled with every data-derived index checked against its array (-DPT_BVH_CHECK=1) and the build
    arena's slices separated by guard bands (PT_BVH_GUARD=1), on the shapes that stress the hierarchy (terrain, needle geometry, 6000
    coincident triangles, 40 000 triangles on a line, the Cornell box): no check fires, no band is touched, every tree holds every primitive
    once.  A process of its own: the library variant must not share a process with the product library."""
    res = _run_tool(["build"], {"PT_LIB": _chk_lib(), "PT_BVH_GUARD": "1"})
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert res.stdout.count("ok ") == 10 and "ARENA GUARD" not in res.stderr and "bounds check failed" not in res.stderr, res.stdout + res.stderr[-2000:]


@pytest.mark.parametrize("inject,what", [("1", "bounds check failed"), ("2", "ARENA GUARD")])
def test_bounds_checks_and_guard_bands_fire(ptlib, inject, what):
    """... and they do fire: with a leaf range pushed past its array (PT_BVH_INJECT=1) the next level's range check refuses the build; with 96
    words written past the end of an arena slice (PT_BVH_INJECT=2) the guard band behind it is found overwritten — pt_create fails with an
    error instead of a GPU memory access fault or a silently corrupted neighbour slice."""
    res = _run_tool(["inject", "terrain70k"], {"PT_LIB": _chk_lib(), "PT_BVH_GUARD": "1", "PT_BVH_INJECT": inject})
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert "refused:" in res.stdout and what in res.stderr, res.stdout + res.stderr[-2000:]

"""GPU parity: the HIP path (through the C ABI) against the CPU checker built with the same
deterministic math (oracle 'det').  Integer / index work and — because both sides execute the same
IEEE operation sequence — every float here is compared BIT-EXACT."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from optixpathtracer_amd import scenes

pytestmark = pytest.mark.gpu


def _renderer(model, probe, cam, w, h, **opt):
    from optixpathtracer_amd.renderer import SampleRenderer, make_camera

    r = SampleRenderer(model)
    r.setProbe(probe)
    r.resize((w, h))
    r.setCamera(make_camera(cam, w / h))
    if opt:
        r.setOptions(**opt)
    return r


def _oracle_render(O, model, probe, cam, w, h, spp, subframes=1, max_depth=8, bsdf_mode=0, use_bvh=None):
    sc = O.make_scene(model, use_bvh)
    pr = O.make_probe(probe)
    U, V, W = scenes.uvw_frame(**cam, aspect=w / h)
    out, accum = None, None
    rays = 0
    for sf in range(subframes):
        out = O.render(sc, pr, (U, V, W), cam["eye"], w, h, spp, max_depth, sf, bsdf_mode, accum)
        accum = out["accum"]
        rays += out["radiance_rays"] + out["shadow_rays"]
    return out


def _gpu_render(r, spp, subframes=1):
    from optixpathtracer_amd import renderer as R

    r.launchParams.samples_per_launch = spp
    for sf in range(subframes):
        r.launchParams.frame.subframe_index = sf
        r.render()
    return dict(
        accum=r.download(R.PT_BUF_ACCUM), frame=r.download(R.PT_BUF_FRAME), normal=r.download(R.PT_BUF_NORMAL),
        color=r.download(R.PT_BUF_COLOR), albedo=r.download(R.PT_BUF_ALBEDO), stats=r.stats(),
    )


def _compare(g, o):
    assert_bits_equal(g["accum"], o["accum"], "accum_buffer")
    assert_bits_equal(g["color"], o["color"], "color_buffer")
    assert_bits_equal(g["normal"], o["normal"], "normal_buffer")
    assert_bits_equal(g["albedo"], o["albedo"], "albedo_buffer")
    assert np.array_equal(g["frame"], o["frame"]), "frame_buffer (rgba8)"


@pytest.fixture(scope="module")
def small_probe():
    return scenes.sky_probe(256, 128).BuildCDF()


# Both schedules of a synchronous frame meet the checker on the edge cases: "chain" = one launch per stage and bounce
# (k_generate / k_trace8 / k_shade, PT_FUSED=0), "fused" = the bounce loop as one persistent kernel (k_path_loop, PT_FUSED=1: what a frame this
# small takes by default).  The switch is read per context at pt_create.
@pytest.fixture(params=["chain", "fused"])
def sched(request, monkeypatch):
    monkeypatch.setenv("PT_FUSED", "0" if request.param == "chain" else "1")
    monkeypatch.setenv("PT_SCHED_TRIALS", "0")  # (the on-line choice between the two would otherwise pick per frame)
    monkeypatch.setenv("PT_FUSED_MAX_COST", "1e9")  # (... and the first guess keeps scenes with expensive rays on the chain)
    return request.param


def _check_sched(g, sched):
    st = g["stats"]
    if sched == "fused":
        assert st["fused_passes"] >= 1 and st["shade_launches"] == 0, st
    else:
        assert st["fused_passes"] == 0 and st["shade_launches"] >= 1, st


# ---------------------------------------------------------------- function tables
def test_division_sqrt_correctly_rounded(ptlib):
    """The numerics contract: device fp32 / and sqrt are IEEE correctly rounded (same bits as the host)."""
    from optixpathtracer_amd.renderer import SampleRenderer

    r = SampleRenderer(scenes.cornell_box())
    rng = np.random.default_rng(3)
    n = 200000
    x = (rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n))).astype(np.float32)
    y = (rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n))).astype(np.float32)
    out = r.evalTable(5, np.stack([np.full(n, 6, np.float32), x, y], 1), 1)[:, 0]
    assert_bits_equal(out, (x / y).astype(np.float32), "fp32 division")
    ax = np.abs(x)
    out = r.evalTable(5, np.stack([np.full(n, 7, np.float32), ax, y], 1), 1)[:, 0]
    assert_bits_equal(out, np.sqrt(ax, dtype=np.float32), "fp32 sqrt")


def test_detmath_tables(ptlib, orc_det):
    from optixpathtracer_amd.renderer import SampleRenderer

    r = SampleRenderer(scenes.cornell_box())
    rng = np.random.default_rng(4)
    n = 100000
    cases = {
        0: (rng.uniform(0, 6.2832, n), None), 1: (rng.uniform(0, 6.2832, n), None), 2: (rng.uniform(-1, 1, n), None),
        3: (rng.standard_normal(n), rng.standard_normal(n)), 4: (np.exp(rng.uniform(-14, 0.5, n)), None),
        5: (rng.uniform(0, 1, n), np.full(n, 1 / 2.4)),
    }
    for fn, (x, y) in cases.items():
        x = x.astype(np.float32)
        y = np.zeros(n, np.float32) if y is None else y.astype(np.float32)
        g = r.evalTable(5, np.stack([np.full(n, fn, np.float32), x, y], 1), 1)[:, 0]
        assert_bits_equal(g, orc_det.math_table(fn, x, y), f"detmath fn {fn}")


def test_rng_tables(ptlib, orc_det):
    import ctypes as C

    from optixpathtracer_amd.renderer import SampleRenderer

    r = SampleRenderer(scenes.cornell_box())
    rng = np.random.default_rng(5)
    n = 5000
    ab = rng.integers(0, 2**32, (n, 2), dtype=np.uint64).astype(np.uint32)
    g = r.evalTable(6, ab.view(np.float32), 8).view(np.uint32)
    L = orc_det.lib
    for i in range(n):
        s = L.orc_tea4(int(ab[i, 0]), int(ab[i, 1]))
        assert g[i, 0] == s
        l = C.c_uint32(s)
        rn = np.float32(L.orc_rnd(C.byref(l)))
        assert g[i, 1] == rn.view(np.uint32) and g[i, 2] == l.value
        st = np.zeros(2, np.uint32)
        L.orc_random_init(st, s)
        f1 = np.float32(L.orc_randf(st)); f2 = np.float32(L.orc_randf(st))
        assert g[i, 3] == f1.view(np.uint32) and g[i, 4] == f2.view(np.uint32)
        assert g[i, 5] == st[0] and g[i, 6] == st[1]
        assert g[i, 7] == L.orc_rand(st)


@pytest.mark.parametrize("mode", [0, 1])
def test_bsdf_tables(ptlib, orc_det, mode):
    import ctypes as C

    from optixpathtracer_amd.renderer import SampleRenderer

    r = SampleRenderer(scenes.cornell_box())
    rng = np.random.default_rng(6)
    n = 4000

    def unit(k):
        v = rng.standard_normal((k, 3)).astype(np.float32)
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    mats = scenes.material_presets() + [scenes.Material(), scenes.Material(transmission=1.0, roughness=0.2, eta=1.33)]
    for mat in mats:
        N, V, Lv = unit(n), unit(n), unit(n)
        V = np.where((np.sum(N * V, 1) < 0)[:, None], -V, V).astype(np.float32)  # view above the surface
        eta = np.where(rng.random(n) < 0.5, 1.0, 1.5).astype(np.float32)
        etaO = np.where(eta == 1.0, 1.5, 1.0).astype(np.float32)
        inp = np.concatenate([N, V, Lv, eta[:, None], etaO[:, None]], 1).astype(np.float32)
        g = r.evalTable(0, inp, 4, material=mat, bsdf_mode=mode)
        ref = np.zeros((n, 4), np.float32)
        albedo = np.ascontiguousarray(mat["color"], np.float32)
        f = np.zeros(3, np.float32)
        for i in range(n):
            orc_det.lib.orc_bsdf_eval(mode, mat.ctypes.data, albedo, float(eta[i]), float(etaO[i]), N[i].copy(), V[i].copy(), Lv[i].copy(), f)
            ref[i, :3] = f
            ref[i, 3] = orc_det.lib.orc_bsdf_pdf(mode, mat.ctypes.data, float(eta[i]), float(etaO[i]), N[i].copy(), V[i].copy(), Lv[i].copy())
        assert_bits_equal(g, ref, "BSDFEval/BSDFPdf")
        seeds = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        inp = np.concatenate([N, V, eta[:, None], etaO[:, None], seeds.view(np.float32)[:, None]], 1).astype(np.float32)
        g = r.evalTable(1, inp, 6, material=mat, bsdf_mode=mode)
        ref = np.zeros((n, 6), np.float32)
        Lo = np.zeros(3, np.float32); pdf = C.c_float(); st = np.zeros(2, np.uint32)
        for i in range(n):
            orc_det.lib.orc_bsdf_sample(mode, mat.ctypes.data, float(eta[i]), float(etaO[i]), N[i].copy(), V[i].copy(), int(seeds[i]), Lo, C.byref(pdf), st)
            ref[i, :3] = Lo
            ref[i, 3] = pdf.value
            ref[i, 4:] = st.view(np.float32)
        assert_bits_equal(g, ref, "BSDFSample")


def test_probe_tables(ptlib, orc_det):
    import ctypes as C

    from optixpathtracer_amd.renderer import SampleRenderer

    # widths around the 6-column line of the device layout (5, 6, 7, 13), flat stretches and NaN rows (spots), the reference's test probes
    probes = [scenes.disc_probe(), scenes.sky_probe(512, 256), scenes.constant_probe(), scenes.spots_probe(1000, 64), scenes.spots_probe(4099, 8, fill=0.002),
              scenes.spots_probe(7, 5, fill=0.5), scenes.spots_probe(13, 3, fill=0.3), scenes.constant_probe(5, 4), scenes.constant_probe(6, 64)]
    for probe in (p.BuildCDF() for p in probes):
        r = SampleRenderer(scenes.cornell_box())
        r.setProbe(probe)
        pr = orc_det.make_probe(probe)
        rng = np.random.default_rng(7)
        n = 5000
        seeds = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        g = r.evalTable(2, seeds.view(np.float32)[:, None], 9)
        ref = np.zeros((n, 9), np.float32)
        d = np.zeros(3, np.float32); c = np.zeros(3, np.float32); pdf = C.c_float(); st = np.zeros(2, np.uint32)
        for i in range(n):
            orc_det.lib.orc_probe_sample(C.byref(pr), int(seeds[i]), d, c, C.byref(pdf), st)
            ref[i, :3] = d; ref[i, 3:6] = c; ref[i, 6] = pdf.value; ref[i, 7:] = st.view(np.float32)
        assert_bits_equal(g, ref, "ProbeSample")
        dirs = rng.standard_normal((n, 3)).astype(np.float32)
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        dirs[0] = (0, 1, 0); dirs[1] = (0, -1, 0); dirs[2] = (1, 0, 0); dirs[3] = (0, 0, -1)
        dirs = dirs.astype(np.float32)
        g = r.evalTable(3, dirs, 6)
        ref = np.zeros((n, 6), np.float32)
        uv = np.zeros(2, np.float32); px = np.zeros(4, np.float32)
        for i in range(n):
            orc_det.lib.orc_probe_dir_to_uv(dirs[i].copy(), uv)
            orc_det.lib.orc_probe_eval(C.byref(pr), uv, px)
            ref[i, :2] = uv; ref[i, 2:] = px
        assert_bits_equal(g, ref, "ProbeEval(ProbeDirToUV)")
        # ProbePdf (Probe.cuh:69-93; pinned to the reference in tests/test_oracle_golden.py::test_probe_pdf): poles and their neighbourhood included
        pd2 = np.concatenate([dirs[:2000], (rng.standard_normal((300, 3)) * np.array([1e-3, 1, 1e-3])).astype(np.float32)])
        g = r.evalTable(8, pd2, 1)[:, 0]
        ref = np.array([orc_det.lib.orc_probe_pdf(C.byref(pr), pd2[i].copy()) for i in range(len(pd2))], np.float32)
        assert_bits_equal(g, ref, "ProbePdf")


def test_make_color_table(ptlib, orc_det):
    from optixpathtracer_amd.renderer import SampleRenderer

    r = SampleRenderer(scenes.cornell_box())
    rng = np.random.default_rng(8)
    c = (rng.random((20000, 3)) * 1.4 - 0.2).astype(np.float32)
    c[:300, 0] = np.linspace(0, 0.005, 300)
    g = r.evalTable(4, c, 1).view(np.uint32)[:, 0]
    ref = np.array([orc_det.lib.orc_make_color(c[i].copy()) for i in range(len(c))], np.uint32)
    assert np.array_equal(g, ref)


# ---------------------------------------------------------------- ray search
def _random_rays(rng, n, lo, hi, tmin=1e-3):
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, np.full((n, 1), tmin, np.float32), d, np.full((n, 1), 1e16, np.float32)], 1).astype(np.float32)


def test_trace_cornell_vs_bruteforce(ptlib, orc_det):
    from optixpathtracer_amd.renderer import SampleRenderer

    m = scenes.cornell_box()
    r = SampleRenderer(m)
    sc = orc_det.make_scene(m, use_bvh=False)  # brute force: independent of any tree
    rng = np.random.default_rng(9)
    rays = _random_rays(rng, 50000, -100, 700)
    # adversarial: rays aimed exactly at vertices and edge midpoints (shared-edge watertightness, ties)
    v, idx, _, _ = m.flatten()
    tri = v[idx]
    targets = np.concatenate([tri.reshape(-1, 3), 0.5 * (tri[:, 0] + tri[:, 1]), 0.5 * (tri[:, 1] + tri[:, 2]), tri.mean(1)]).astype(np.float32)
    o = np.array([278.0, 273.0, -900.0], np.float32) + rng.uniform(-50, 50, (len(targets), 3)).astype(np.float32)
    d = targets - o
    adv = np.concatenate([o, np.full((len(o), 1), 1e-3, np.float32), d, np.full((len(o), 1), 1e16, np.float32)], 1).astype(np.float32)
    rays = np.concatenate([rays, adv]).astype(np.float32)
    (t, prim), _ = r.trace(rays)
    to, po = orc_det.trace_closest(sc, rays)
    assert np.array_equal(prim, po)
    assert_bits_equal(t, to, "closest-hit t")
    occ, _ = r.trace(rays, any_hit=True)
    assert np.array_equal(occ, orc_det.trace_any(sc, rays))


def test_far_camera_keeps_hits_on_flat_triangles(ptlib, orc_det):
    """A floor quad in the plane y = 0 (a bounding box of zero thickness) seen from 5 to 60 scene sizes away along oblique directions: the
    computed hit point o + t d is off the plane by up to ~2^-22 of the distance travelled, more than the fixed half padding hp from about
    16 scene sizes on — hit_in_box's tolerance grows with t d for that reason (pt_bvh.h).  Every ray aimed at the interior of the quad
    must hit it; kernel and brute-force checker agree bit for bit."""
    from optixpathtracer_amd.renderer import SampleRenderer

    S = 100.0
    m = scenes.Model()
    m.meshes.append(scenes._quads_to_mesh([[(-S, 0, -S), (S, 0, -S), (S, 0, S), (-S, 0, S)]], scenes.Material()))
    scenes.add_box(m, scenes.Material(), (0.0, 5.0, 0.0), (5.0, 5.0, 5.0))
    r = SampleRenderer(m)
    sc = orc_det.make_scene(m, use_bvh=False)
    rng = np.random.default_rng(77)
    n = 40000
    targets = np.zeros((n, 3), np.float32)
    targets[:, 0] = rng.uniform(-0.95 * S, 0.95 * S, n)
    targets[:, 2] = rng.uniform(-0.95 * S, 0.95 * S, n)
    keep = (np.abs(targets[:, 0]) > 6.0) | (np.abs(targets[:, 2]) > 6.0)  # not under the box
    dist = rng.choice([5.0, 16.0, 32.0, 60.0], n) * S
    elev = rng.uniform(0.05, 1.2, n)  # radians above the floor: grazing to steep
    azim = rng.uniform(0, 2 * np.pi, n)
    back = np.stack([np.cos(elev) * np.cos(azim), np.sin(elev), np.cos(elev) * np.sin(azim)], 1)
    o = (targets + back * dist[:, None]).astype(np.float32)
    d = (targets - o).astype(np.float64)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([o, np.full((n, 1), 1e-3, np.float32), d, np.full((n, 1), 1e16, np.float32)], 1).astype(np.float32)
    (t, prim), _ = r.trace(rays)
    to, po = orc_det.trace_closest(sc, rays)
    assert np.array_equal(prim, po)
    assert_bits_equal(t, to, "closest-hit t from far away")
    # box shadows aside, every such ray reaches the floor (primitives 0 and 1) unless the box is in the way
    floor_or_box = prim >= 0
    assert floor_or_box[keep].all(), f"{(~floor_or_box[keep]).sum()} of {keep.sum()} far rays fell through the floor"
    assert (prim[keep] <= 1).mean() > 0.9


def test_trace_terrain_vs_oracle(ptlib, orc_det):
    from optixpathtracer_amd.renderer import SampleRenderer

    m = scenes.voxel_terrain(n=96, target_tris=70000)
    r = SampleRenderer(m)
    sc = orc_det.make_scene(m, use_bvh=True)  # the oracle's own median-split tree (validated vs brute force on CPU)
    rng = np.random.default_rng(10)
    rays = _random_rays(rng, 200000, -110, 110)
    rays[:, 1] = rng.uniform(-30, 60, len(rays))
    # secondary-like rays starting ON surfaces: re-launch from hit points
    (t, prim), _ = r.trace(rays)
    to, po = orc_det.trace_closest(sc, rays)
    assert np.array_equal(prim, po) and (prim >= 0).mean() > 0.2
    assert_bits_equal(t, to, "closest-hit t")
    hit = prim >= 0
    P = rays[hit, :3] + t[hit, None] * rays[hit, 4:7]
    r2 = _random_rays(rng, int(hit.sum()), 0, 1, tmin=1e-2)
    r2[:, :3] = P
    occ, _ = r.trace(r2, any_hit=True)
    assert np.array_equal(occ, orc_det.trace_any(sc, r2))
    (t2, p2), _ = r.trace(r2)
    to2, po2 = orc_det.trace_closest(sc, r2)
    assert np.array_equal(p2, po2)
    assert_bits_equal(t2, to2, "closest-hit t (surface origins)")


def _soup_model(rng):
    """Triangle soup with the cases a regular scene never has: sizes over five decades, zero-area and duplicated
    triangles (ties → lowest primitive index), coplanar overlaps, scene-spanning and far-away triangles."""
    def tris(n, lo, hi, size):
        c = rng.uniform(lo, hi, (n, 1, 3))
        return (c + rng.standard_normal((n, 3, 3)) * size).astype(np.float32)

    parts = [tris(3000, -100, 100, np.exp(rng.uniform(np.log(1e-3), np.log(50.0), (3000, 1, 1))))]
    deg = tris(200, -100, 100, 5.0)
    deg[:100, 2] = deg[:100, 1]                      # two equal vertices
    deg[100:, 2] = 0.5 * (deg[100:, 0] + deg[100:, 1])  # collinear
    parts.append(deg)
    parts.append(parts[0][rng.integers(0, 3000, 100)])  # exact duplicates
    parts.append(tris(100, -20, 20, 300.0))              # scene-spanning
    parts.append(tris(50, 9000, 11000, 40.0))            # far away: stretches the root box and the 8-bit grids
    cop = tris(60, -50, 50, 20.0)
    cop[:, :, 1] = 7.0                                    # coplanar overlapping triangles in the plane y = 7
    parts.append(cop)
    tri = np.concatenate(parts).astype(np.float32)
    perm = rng.permutation(len(tri))
    tri = tri[perm]
    mesh = scenes.TriangleMesh(vertex=tri.reshape(-1, 3).copy(), index=np.arange(3 * len(tri), dtype=np.uint32).reshape(-1, 3), material=scenes.Material())
    return scenes.Model(meshes=[mesh]), tri


def test_trace_triangle_soup_adversarial(ptlib, orc_det, monkeypatch):
    from optixpathtracer_amd.renderer import SampleRenderer

    rng = np.random.default_rng(77)
    m, tri = _soup_model(rng)
    sc = orc_det.make_scene(m, use_bvh=False)
    rays = [_random_rays(rng, 30000, -120, 120)]
    ax = _random_rays(rng, 6000, -110, 110)                # axis-parallel rays: exact +0 / -0 direction components
    k = rng.integers(0, 3, len(ax))
    sgn = rng.choice([-1.0, 1.0], len(ax)).astype(np.float32)
    ax[:, 4:7] = np.where(rng.random((len(ax), 3)) < 0.5, 0.0, -0.0).astype(np.float32)
    ax[np.arange(len(ax)), 4 + k] = sgn
    rays.append(ax)
    pl = _random_rays(rng, 6000, -110, 110)                # one zero component, in the plane of the coplanar group
    pl[:, 1] = 7.0
    pl[:, 5] = 0.0
    rays.append(pl)
    sc_ = _random_rays(rng, 6000, -110, 110)               # unnormalised directions and finite [tmin, tmax] windows
    sc_[:, 4:7] *= np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (len(sc_), 1))).astype(np.float32)
    sc_[:, 3] = rng.uniform(0, 50, len(sc_)) / np.linalg.norm(sc_[:, 4:7], axis=1)
    sc_[:, 7] = sc_[:, 3] + rng.uniform(0, 150, len(sc_)) / np.linalg.norm(sc_[:, 4:7], axis=1)
    rays.append(sc_)
    tg = tri[rng.integers(0, len(tri), 6000)]               # aimed at vertices / edge midpoints of the soup
    targets = np.where(rng.random((len(tg), 1)) < 0.5, tg[:, 0], 0.5 * (tg[:, 0] + tg[:, 1]))
    o = rng.uniform(-150, 150, (len(tg), 3)).astype(np.float32)
    aimed = np.concatenate([o, np.full((len(o), 1), 1e-3, np.float32), (targets - o).astype(np.float32), np.full((len(o), 1), 1e16, np.float32)], 1)
    rays.append(aimed)
    rays = np.concatenate(rays).astype(np.float32)
    to, po = orc_det.trace_closest(sc, rays)
    occ_o = orc_det.trace_any(sc, rays)
    assert (po >= 0).mean() > 0.3
    for builder in ("lbvh", "ploc", "sah"):  # every hierarchy the builder can put under the 8-wide tree
        monkeypatch.setenv("PT_BVH_BUILDER", builder)
        r = SampleRenderer(m)
        (t, prim), _ = r.trace(rays)
        assert np.array_equal(prim, po), f"{builder}: {(prim != po).sum()} primitive ids differ"
        assert_bits_equal(t, to, f"closest-hit t, {builder}")
        occ, _ = r.trace(rays, any_hit=True)
        assert np.array_equal(occ, occ_o)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["one_triangle", "three_triangles", "four_triangles", "five_triangles", "thousands_of_copies", "collinear_centroids", "flat_scene", "tiny_in_huge", "many_meshes"])
def test_builder_degenerate_scenes(ptlib, orc_det, case, monkeypatch, small_probe):
    """Scenes that stress the on-GPU builder rather than the traversal: Morton codes that all tie, root boxes with zero extent
    on one or two axes (quantisation exponents), a single primitive, triangles far smaller than one 8-bit grid cell of the
    root box, and hundreds of one-triangle meshes (primitive ids across mesh boundaries)."""
    from optixpathtracer_amd.renderer import SampleRenderer

    rng = np.random.default_rng(5)
    base = np.array([[0, 0, 0], [4, 0, 0], [0, 3, 0]], np.float32)
    if case == "one_triangle":
        tri = base[None]
    elif case in ("three_triangles", "four_triangles", "five_triangles"):  # around the single-node limit (3 triangles in one leaf child)
        k = {"three_triangles": 3, "four_triangles": 4, "five_triangles": 5}[case]
        tri = (rng.uniform(-3, 3, (k, 1, 3)) + rng.standard_normal((k, 3, 3)) * 2).astype(np.float32)
    elif case == "thousands_of_copies":
        tri = np.repeat(base[None], 5000, 0)  # above the 4096 triangles from which the builder also tries a PLOC hierarchy
    elif case == "collinear_centroids":
        tri = base[None] * 0.2 + np.linspace(-50, 50, 4000, dtype=np.float32)[:, None, None] * np.array([1, 0, 0], np.float32)
    elif case == "flat_scene":  # every vertex in the plane z = 2: the root box has zero extent in z
        tri = rng.uniform(-30, 30, (2500, 3, 3)).astype(np.float32)
        tri[:, :, 2] = 2.0
    elif case == "tiny_in_huge":
        c = rng.uniform(-500, 500, (4000, 1, 3))  # triangles of ≈0.05 units: 1/80 of one cell of the root's 8-bit grid, ≈10^3 float ulps across
        tri = (c + rng.standard_normal((4000, 3, 3)) * 0.05).astype(np.float32)
        tri[:200] = (rng.uniform(-1, 1, (200, 1, 3)) + rng.standard_normal((200, 3, 3)) * 0.3).astype(np.float32)  # a dense cluster at the origin
    else:
        tri = (rng.uniform(-20, 20, (600, 1, 3)) + rng.standard_normal((600, 3, 3)) * 2).astype(np.float32)
    if case == "many_meshes":
        meshes = [scenes.TriangleMesh(vertex=t.copy(), index=np.array([[0, 1, 2]], np.uint32), material=scenes.Material()) for t in tri]
    else:
        meshes = [scenes.TriangleMesh(vertex=tri.reshape(-1, 3).copy(), index=np.arange(3 * len(tri), dtype=np.uint32).reshape(-1, 3), material=scenes.Material())]
    m = scenes.Model(meshes=meshes)
    sc = orc_det.make_scene(m, use_bvh=False)
    lo, hi = float(tri.min()) - 5, float(tri.max()) + 5
    rays = _random_rays(rng, 20000, lo, hi)
    tg = tri[rng.integers(0, len(tri), 20000)]                # aimed at the triangles (random barycentrics), from outside and inside
    bc = rng.dirichlet([1, 1, 1], len(tg)).astype(np.float32)
    targets = (tg * bc[:, :, None]).sum(1)
    # origins stay within a few scene sizes: from hundreds of diameters away the float triangle test itself reports hits for rays that
    # pass nowhere near a small triangle (tools/diag_degenerate2.py: the checker's own BVH2 disagrees with its brute force there too),
    # and exact ties between coplanar copies are then decided by rounding noise larger than the builder's 2^-16 box padding
    o = (targets + rng.standard_normal((len(tg), 3)) * rng.choice([0.5, 20.0], (len(tg), 1))).astype(np.float32)
    aimed = np.concatenate([o, np.full((len(o), 1), 1e-3, np.float32), (targets - o).astype(np.float32), np.full((len(o), 1), 1e16, np.float32)], 1)
    rays = np.concatenate([rays, aimed]).astype(np.float32)
    to, po = orc_det.trace_closest(sc, rays)
    occ_o = orc_det.trace_any(sc, rays)
    assert (po >= 0).mean() > 0.2
    for builder in ("lbvh", "ploc", "sah"):
        monkeypatch.setenv("PT_BVH_BUILDER", builder)
        r = SampleRenderer(m)
        (t, prim), _ = r.trace(rays)
        assert np.array_equal(prim, po), f"{case}, {builder}: {(prim != po).sum()} primitive ids differ"
        assert_bits_equal(t, to, f"{case}: closest-hit t, {builder}")
        occ, _ = r.trace(rays, any_hit=True)
        assert np.array_equal(occ, occ_o)
    # ... and a small frame of the same scene through BOTH schedules of a synchronous frame (launch chain, fused bounce loop) against the checker
    monkeypatch.delenv("PT_BVH_BUILDER")
    ctr = 0.5 * (tri.reshape(-1, 3).min(0) + tri.reshape(-1, 3).max(0))
    ext = float(np.abs(tri.reshape(-1, 3) - ctr).max()) + 1.0
    cam = dict(eye=tuple(float(x) for x in ctr + np.array([0.9, 0.7, 1.6], np.float32) * ext), lookat=tuple(float(x) for x in ctr), up=(0.0, 1.0, 0.0), fovY=50.0)
    w, h = 40, 24
    o = _oracle_render(orc_det, m, small_probe, cam, w, h, 2, use_bvh=False)
    for fused in ("0", "1"):
        monkeypatch.setenv("PT_FUSED", fused)
        monkeypatch.setenv("PT_SCHED_TRIALS", "0")
        monkeypatch.setenv("PT_FUSED_MAX_COST", "1e9")  # (5000 coincident triangles cost a calibration ray thousands of steps: the policy would keep the chain)
        g = _gpu_render(_renderer(m, small_probe, cam, w, h), 2)
        _check_sched(g, "fused" if fused == "1" else "chain")
        _compare(g, o)


# ---------------------------------------------------------------- whole renders
def test_render_cornell_c1_lambert(ptlib, orc_det, small_probe):
    """BASELINE config 1: Cornell 256x256, 1 spp, depth 4, Lambert."""
    m = scenes.cornell_box()
    w = h = 256
    r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h, max_depth=4, bsdf_mode=1)
    g = _gpu_render(r, 1)
    o = _oracle_render(orc_det, m, small_probe, scenes.CORNELL_CAMERA, w, h, 1, max_depth=4, bsdf_mode=1)
    _compare(g, o)


def test_render_cornell_disney_4spp_depth8(ptlib, orc_det, small_probe):
    """BASELINE config 2 semantics at a size the CPU finishes in seconds."""
    m = scenes.cornell_box()
    w, h = 240, 136
    r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h)
    g = _gpu_render(r, 4)
    o = _oracle_render(orc_det, m, small_probe, scenes.CORNELL_CAMERA, w, h, 4)
    _compare(g, o)
    # rays: the GPU skips provably dead rays (see DESIGN.md "ray accounting"), never traces more than the reference
    assert g["stats"]["radiance_rays"] <= o["radiance_rays"] and g["stats"]["shadow_rays"] <= o["shadow_rays"]
    assert g["stats"]["radiance_rays"] > 0.8 * o["radiance_rays"]
    # shaded hits (pt_stats): every live shadow ray comes from one, and every one was reached by a traced closest-hit ray
    assert g["stats"]["shadow_rays"] <= g["stats"]["shaded_hits"] <= g["stats"]["radiance_rays"]


def test_c2_cornell_1080p_4spp_depth8_rows(ptlib, orc_det):
    """BASELINE config C2 at its literal size (Cornell box, 1920x1080, 4 spp, depth 8, Disney BSDF, the 2048x1024 sky+sun probe):
    rows of the frame re-rendered by the checker (brute-force ray search over the 32 triangles) match bit for bit."""
    import ctypes as C

    from oracle import orc as orc_mod

    m = scenes.cornell_box()
    probe = scenes.sky_probe(2048, 1024).BuildCDF()
    w, h = 1920, 1080
    g = _gpu_render(_renderer(m, probe, scenes.CORNELL_CAMERA, w, h), 4)
    assert g["stats"]["paths"] == w * h * 4
    sc = orc_det.make_scene(m, False)
    pr = orc_det.make_probe(probe)
    U, V, W = scenes.uvw_frame(**scenes.CORNELL_CAMERA, aspect=w / h)
    prm = orc_mod.Params()
    prm.width, prm.height, prm.subframe_index, prm.samples_per_launch, prm.max_depth, prm.bsdf_mode = w, h, 0, 4, 8, 0
    for dst, src in ((prm.eye, scenes.CORNELL_CAMERA["eye"]), (prm.U, U), (prm.V, V), (prm.W, W)):
        for k in range(3):
            dst[k] = float(src[k])
    rows = np.array([0, 300, 540, 811, 1079], np.int32)
    accum = np.zeros((h, w, 4), np.float32)
    orc_det.lib.orc_render_rows.argtypes = [C.c_void_p, C.POINTER(orc_mod.Probe), C.POINTER(orc_mod.Params), orc_mod.f32p, orc_mod.i32p, C.c_int, C.c_int]
    orc_det.lib.orc_render_rows(sc.h, C.byref(pr), C.byref(prm), accum.reshape(-1), rows, len(rows), 16)
    for y in rows:
        assert_bits_equal(g["accum"][y], accum[y], f"row {y} of the C2 frame")


def test_both_traversal_kernels_agree(ptlib, orc_det, small_probe, monkeypatch):
    """Every traversal schedule over either hierarchy gives the checker's bits; the removed A/B options are refused."""
    m = scenes.voxel_terrain(n=96, target_tris=70000)
    w, h = 128, 72
    o = _oracle_render(orc_det, m, small_probe, scenes.TERRAIN_CAMERA, w, h, 2)
    # unified launches (default), closest-hit and shadow launches apart on 2 streams, 3 concurrent pixel chunks, asynchronous shadow records
    for builder in ("lbvh", "ploc", "sah"):
        monkeypatch.setenv("PT_BVH_BUILDER", builder)
        for opt in (dict(), dict(split_shadow=1), dict(streams=3), dict(split_shadow=2)):
            r = _renderer(m, small_probe, scenes.TERRAIN_CAMERA, w, h, **opt)
            _compare(_gpu_render(r, 2), o)
    with pytest.raises(RuntimeError, match="reserved"):
        r.setOptions(bvh_kind=1)
    with pytest.raises(RuntimeError, match="reserved"):
        r.setOptions(trace_kernel=1)


def test_render_progressive_subframes(ptlib, orc_det, small_probe):
    """deviceProgram.cu:460-467: clamp + running lerp over subframes 0..3."""
    m = scenes.cornell_box()
    w, h = 96, 64
    r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h)
    g = _gpu_render(r, 2, subframes=4)
    o = _oracle_render(orc_det, m, small_probe, scenes.CORNELL_CAMERA, w, h, 2, subframes=4)
    _compare(g, o)


@pytest.mark.parametrize("opts", [dict(), dict(max_paths=6000), dict(streams=1), dict(split_shadow=2)], ids=["default", "many_chunks", "one_stream", "async_shadows"])
def test_pipelined_frames_are_bit_identical(ptlib, small_probe, opts):
    """pt_options.frames_in_flight = 2: frame k+1 is enqueued before frame k is waited for.  Progressive accumulation makes every frame
    depend on the previous one through accum_buffer, so any mis-ordering between frames shows; the buffers, the per-frame and the
    cumulative ray counts must equal the synchronous run's, also across a camera change, a resize and a partition change in mid-flight."""
    from optixpathtracer_amd import renderer as R

    m = scenes.voxel_terrain(n=96, target_tris=70000)
    w, h = 160, 96

    def run(fif):
        r = _renderer(m, small_probe, scenes.TERRAIN_CAMERA, w, h, frames_in_flight=fif, **opts)
        r.launchParams.samples_per_launch = 2
        per_frame = []
        for sf in range(7):
            if sf == 4:  # host-side change between frames: passed by value with the next frame
                cam = dict(scenes.TERRAIN_CAMERA)
                cam["eye"] = tuple(c * 1.05 for c in cam["eye"])
                r.setCamera(R.make_camera(cam, w / h))
            r.launchParams.frame.subframe_index = sf
            r.render()
            if fif < 2:
                per_frame.append(r.stats()["radiance_rays"] + r.stats()["shadow_rays"])
        out = dict(accum=r.download(R.PT_BUF_ACCUM), frame=r.download(R.PT_BUF_FRAME), normal=r.download(R.PT_BUF_NORMAL))
        st = r.stats()
        # a resize and a partition with frames still in flight: the library finishes them first
        r.launchParams.frame.subframe_index = 0
        r.render()
        r.resize((96, 64))
        r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, 96 / 64))
        r.render()
        r.setPartition(1, 3, 16, 8)
        r.render()
        r.render()
        r.sync()
        out["small"] = r.download(R.PT_BUF_ACCUM)
        out["st2"] = r.stats()
        return out, st, per_frame

    sync, st_s, per_frame = run(0)
    for fif in (2, 3):  # 2: the frame's pixel chunks, no frame-wide start; 3: whole frames on alternating streams, resolves chained
        pipe, st_p, _ = run(fif)
        for k in ("accum", "frame", "normal", "small"):
            assert np.array_equal(sync[k].view(np.uint32), pipe[k].view(np.uint32)), (fif, k)
        assert st_p["frames"] == st_s["frames"] == 7
        assert st_p["total_radiance_rays"] + st_p["total_shadow_rays"] == st_s["total_radiance_rays"] + st_s["total_shadow_rays"] == sum(per_frame)
        assert (st_p["radiance_rays"], st_p["shadow_rays"], st_p["shaded_hits"]) == (st_s["radiance_rays"], st_s["shadow_rays"], st_s["shaded_hits"])  # last frame
        assert pipe["st2"]["frames"] == sync["st2"]["frames"] == 11
        assert pipe["st2"]["total_radiance_rays"] == sync["st2"]["total_radiance_rays"]


@pytest.mark.parametrize("fif", [2, 3])
def test_pipelined_frames_report_errors_late_but_loudly(ptlib, small_probe, monkeypatch, fif):
    """A traversal-stack overflow in frame k is reported by the call that waits for frame k (the next pt_render or pt_sync)."""
    from optixpathtracer_amd import renderer as R

    monkeypatch.setenv("PT_STACK_LDS_SKIP", "11")
    monkeypatch.setenv("PT_STACK_CAP", "2")
    monkeypatch.setenv("PT_STACK_NOCHECK", "1")
    r = _renderer(scenes.voxel_terrain(n=96, target_tris=70000), small_probe, scenes.TERRAIN_CAMERA, 96, 64, frames_in_flight=fif)
    r.launchParams.samples_per_launch = 1
    r.render()  # enqueued, not yet waited for
    with pytest.raises(RuntimeError, match="traversal stack overflow"):
        r.sync()
    for _ in range(fif - 1):
        r.render()
    with pytest.raises(RuntimeError, match="traversal stack overflow"):
        r.render()  # waits for the oldest frame in flight


def test_render_sample_chunking_invariant(ptlib, orc_det, small_probe):
    """Splitting a launch into pixel/sample chunks (max_paths) must not change a single bit."""
    m = scenes.cornell_box()
    w, h = 96, 64
    o = _oracle_render(orc_det, m, small_probe, scenes.CORNELL_CAMERA, w, h, 5)
    for max_paths in (0, 96 * 64 * 2, 1000, 64):
        r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h, max_paths=max_paths)
        _compare(_gpu_render(r, 5), o)


def test_render_terrain_all_material_branches(ptlib, orc_det, sched):
    """~70k-triangle voxel terrain, 8 material presets (transmission, subsurface, clearcoat, metal ...)."""
    m = scenes.voxel_terrain(n=96, target_tris=70000)
    probe = scenes.sky_probe(512, 256).BuildCDF()
    w, h = 160, 90
    r = _renderer(m, probe, scenes.TERRAIN_CAMERA, w, h)
    g = _gpu_render(r, 4)
    _check_sched(g, sched)
    o = _oracle_render(orc_det, m, probe, scenes.TERRAIN_CAMERA, w, h, 4)
    _compare(g, o)


def test_render_shadow_catcher_two_box(ptlib, orc_det):
    """The reference's own procedural scene (main.cpp:165-169) with the SHADOW_CATCHER ground:
    pass-through of secondary rays, alpha from occluded shadow rays, the depth==max trace that is NOT dead."""
    m = scenes.two_box_scene(True)
    probe = scenes.disc_probe().BuildCDF()
    w, h = 96, 64
    for depth in (8, 2):
        r = _renderer(m, probe, scenes.TWO_BOX_CAMERA, w, h, max_depth=depth)
        g = _gpu_render(r, 3, subframes=2)
        o = _oracle_render(orc_det, m, probe, scenes.TWO_BOX_CAMERA, w, h, 3, subframes=2, max_depth=depth, use_bvh=False)
        _compare(g, o)


def test_shadow_catcher_scene_with_frames_in_flight(ptlib, orc_det):
    """The shadow-catcher scene (one sample per pass, the extra pass-through iterations) rendered with 2 and 3 frames in flight
    against the checker: five progressive subframes."""
    m = scenes.two_box_scene(True)
    probe = scenes.disc_probe().BuildCDF()
    w, h = 96, 64
    o = _oracle_render(orc_det, m, probe, scenes.TWO_BOX_CAMERA, w, h, 3, subframes=5, use_bvh=False)
    for fif in (2, 3):
        r = _renderer(m, probe, scenes.TWO_BOX_CAMERA, w, h, frames_in_flight=fif)
        _compare(_gpu_render(r, 3, subframes=5), o)


def test_render_shadow_catcher_pass_through_chains(ptlib, orc_det):
    """Paths that cross several shadow-catcher faces and then keep bouncing: a pass-through does not consume depth
    (deviceProgram.cu:503-508), so the reference's raygen loop (:411-443) traces such a path more than max_depth+1 times.
    The wavefront schedule must keep iterating until no path is left (it used to stop after max_depth+1 launches)."""
    m = scenes.catcher_stack_scene()
    probe = scenes.disc_probe().BuildCDF()
    w, h = 96, 64
    for depth in (8, 3, 1):
        r = _renderer(m, probe, scenes.TWO_BOX_CAMERA, w, h, max_depth=depth)
        g = _gpu_render(r, 3, subframes=2)
        o = _oracle_render(orc_det, m, probe, scenes.TWO_BOX_CAMERA, w, h, 3, subframes=2, max_depth=depth, use_bvh=False)
        _compare(g, o)
        if depth < 8:  # 3 sample passes of depth+1 closest-hit launches each, plus the extra iterations
            assert g["stats"]["trace_launches"] > 3 * (depth + 1), g["stats"]
    # several chunks / streams and sample passes: the extra iterations of one chunk must not disturb the others
    r = _renderer(m, probe, scenes.TWO_BOX_CAMERA, w, h, max_depth=3, max_paths=2048, streams=2)
    g = _gpu_render(r, 3, subframes=2)
    o = _oracle_render(orc_det, m, probe, scenes.TWO_BOX_CAMERA, w, h, 3, subframes=2, max_depth=3, use_bvh=False)
    _compare(g, o)


def test_render_constant_probe_and_edge_sizes(ptlib, orc_det, sched):
    """Constant-white probe (loadColor, sv4 main.cpp:167-180); odd sizes that do not fill 8x8 blocks; 1x1."""
    m = scenes.cornell_box()
    probe = scenes.constant_probe().BuildCDF()
    for (w, h) in ((1, 1), (7, 5), (33, 17)):
        r = _renderer(m, probe, scenes.CORNELL_CAMERA, w, h)
        g = _gpu_render(r, 2)
        _check_sched(g, sched)
        o = _oracle_render(orc_det, m, probe, scenes.CORNELL_CAMERA, w, h, 2)
        _compare(g, o)


def test_partition_reproduces_single_gpu_bits(ptlib, small_probe):
    """Tile partition (the multi-GPU path): the union of every rank's pixels equals the unpartitioned image,
    bit for bit (seeds depend only on pixel index and subframe, deviceProgram.cu:357)."""
    from optixpathtracer_amd import renderer as R

    m = scenes.cornell_box()
    w, h = 200, 120
    full = _gpu_render(_renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h), 3)
    acc = np.zeros_like(full["accum"])
    cover = np.zeros((h, w), np.int32)
    world = 3
    for rank in range(world):
        r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h)
        r.setPartition(rank, world, 16, 8)
        g = _gpu_render(r, 3)
        mine = g["accum"][..., 3] == 1.0  # alpha channel of accum is written as 1 by rendered pixels only
        cover += mine
        acc[mine] = g["accum"][mine]
        owned, padded = r.ownedPixels()
        assert owned == mine.sum() and padded >= owned
    assert (cover == 1).all()
    assert_bits_equal(acc, full["accum"], "partitioned accum")


def test_tonemap_sqrt_epilogue(ptlib, orc_det, small_probe):
    """toneMap.cu:41-58 computeFinalPixelColorsKernel."""
    m = scenes.cornell_box()
    w, h = 64, 48
    r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h)
    g = _gpu_render(r, 2)
    out = r.tonemapSqrt()
    ref = np.zeros(w * h, np.uint32)
    orc_det.lib.orc_tonemap_sqrt(np.ascontiguousarray(g["accum"]).reshape(-1), ref, w * h)
    assert np.array_equal(out.reshape(-1), ref)


def test_error_behaviour(ptlib, small_probe):
    from optixpathtracer_amd.renderer import SampleRenderer

    r = SampleRenderer(scenes.cornell_box())
    r.render()  # before resize: silently nothing (SimplePathtracer.cpp:77)
    r.resize((0, 0))  # ignored (:112)
    r.resize((32, 16))
    with pytest.raises(RuntimeError):
        r.render()  # no probe
    with pytest.raises(RuntimeError):
        r.setProbe(scenes.constant_probe())  # BuildCDF not run: "Probe Data is not valid" (Probe.h:104)
    bad = scenes.Model([scenes.TriangleMesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 5]], np.uint32), scenes.Material())])
    with pytest.raises(RuntimeError):
        SampleRenderer(bad)


# ---------------------------------------------------------------- full-size properties (BASELINE sizes)
def test_fullsize_1080p_terrain_properties(ptlib, orc_det):
    """C3 size (1M triangles, 1920x1080, 4 spp, depth 8): determinism, finiteness, ray bound, and a
    bit-exact window against the oracle (rows of the same full-size launch rendered on the CPU)."""
    m = scenes.voxel_terrain()
    assert m.num_triangles == 1_000_000
    probe = scenes.sky_probe(2048, 1024).BuildCDF()
    w, h = 1920, 1080
    r = _renderer(m, probe, scenes.TERRAIN_CAMERA, w, h)
    g1 = _gpu_render(r, 4)
    g2 = _gpu_render(r, 4)  # subframe 0 again: idempotent
    assert np.array_equal(g1["accum"].view(np.uint32), g2["accum"].view(np.uint32))
    assert np.isfinite(g1["accum"]).all()
    st = g1["stats"]
    assert st["paths"] == w * h * 4
    assert st["radiance_rays"] <= st["paths"] * 8 and st["shadow_rays"] <= st["radiance_rays"]
    # window: the oracle renders a full-size frame's pixel subset = same seeds (pixel index uses the full width)
    sc = orc_det.make_scene(m, True)
    pr = orc_det.make_probe(probe)
    U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=w / h)
    from oracle import orc as orc_mod
    import ctypes as C

    prm = orc_mod.Params()
    prm.width, prm.height, prm.subframe_index, prm.samples_per_launch, prm.max_depth, prm.bsdf_mode = w, h, 0, 4, 8, 0
    for dst, src in ((prm.eye, scenes.TERRAIN_CAMERA["eye"]), (prm.U, U), (prm.V, V), (prm.W, W)):
        for k in range(3):
            dst[k] = float(src[k])
    rows = [5, 333, 540, 777, 1079]
    accum = np.zeros((h, w, 4), np.float32)
    orc_det.lib.orc_render_rows.argtypes = [C.c_void_p, C.POINTER(orc_mod.Probe), C.POINTER(orc_mod.Params), orc_mod.f32p, orc_mod.i32p, C.c_int, C.c_int]
    orc_det.lib.orc_render_rows(sc.h, C.byref(pr), C.byref(prm), accum.reshape(-1), np.array(rows, np.int32), len(rows), 8)
    for y in rows:
        assert_bits_equal(g1["accum"][y], accum[y], f"row {y} of the 1080p frame")
    # the schedule never changes a bit at full size either: one stream / five streams / a third of the path slots
    # (samples split into passes) / separate shadow launches all reproduce the default frame
    for opt in (dict(streams=1), dict(streams=5), dict(max_paths=3_000_000), dict(split_shadow=1), dict(split_shadow=2), dict(split_shadow=2, streams=1)):
        r2 = _renderer(m, probe, scenes.TERRAIN_CAMERA, w, h, **opt)
        g3 = _gpu_render(r2, 4)
        for k in ("accum", "color", "normal", "albedo"):
            assert np.array_equal(g3[k].view(np.uint32), g1[k].view(np.uint32)), f"{opt}: {k} differs from the default schedule"
        assert np.array_equal(g3["frame"], g1["frame"])
        assert g3["stats"]["radiance_rays"] == st["radiance_rays"] and g3["stats"]["shadow_rays"] == st["shadow_rays"]
        del r2


def test_c4_4k_16spp_8way_partition_rank_by_rank(ptlib, orc_det):
    """BASELINE config C4 at its literal size: the 1 M-triangle scene, 3840x2160, 16 spp, depth 8, image tile-partitioned
    8 ways (interleaved 64x16 tiles).  The eight shares are rendered one after the other on this GPU; their union must cover
    every pixel exactly once and equal the checker's rows of the unpartitioned 4K frame bit for bit."""
    from oracle import orc as orc_mod
    import ctypes as C

    m = scenes.voxel_terrain()
    probe = scenes.sky_probe(2048, 1024).BuildCDF()
    w, h, spp, world = 3840, 2160, 16, 8
    r = _renderer(m, probe, scenes.TERRAIN_CAMERA, w, h)
    r.launchParams.samples_per_launch = spp
    r.launchParams.frame.subframe_index = 0
    from optixpathtracer_amd import renderer as R

    union = np.zeros((h, w, 4), np.float32)
    cover = np.zeros((h, w), np.int32)
    rays = 0
    owned_total = 0
    for rank in range(world):
        r.setPartition(rank, world, 64, 16)  # re-applies to the 4K frame: buffers are cleared, only this rank's tiles get written
        r.render()
        a = r.download(R.PT_BUF_ACCUM)
        mine = a[..., 3] == 1.0
        cover += mine
        union[mine] = a[mine]
        st = r.stats()
        rays += st["radiance_rays"] + st["shadow_rays"]
        owned_total += r.ownedPixels()[0]
        assert st["paths"] == r.ownedPixels()[0] * spp
        assert abs(int(mine.sum()) - w * h // world) <= 64 * 16 * 8  # shares are balanced to within a few tiles
    assert owned_total == w * h and (cover == 1).all(), "every pixel is rendered by exactly one rank"
    assert np.isfinite(union).all() and rays > 4 * w * h * spp
    sc = orc_det.make_scene(m, True)
    pr = orc_det.make_probe(probe)
    U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=w / h)
    prm = orc_mod.Params()
    prm.width, prm.height, prm.subframe_index, prm.samples_per_launch, prm.max_depth, prm.bsdf_mode = w, h, 0, spp, 8, 0
    for dst, src in ((prm.eye, scenes.TERRAIN_CAMERA["eye"]), (prm.U, U), (prm.V, V), (prm.W, W)):
        for k in range(3):
            dst[k] = float(src[k])
    rows = [3, 1040, 1519, 2159]  # rows owned by different ranks (tile rows 0, 65, 94, 134)
    accum = np.zeros((h, w, 4), np.float32)
    orc_det.lib.orc_render_rows.argtypes = [C.c_void_p, C.POINTER(orc_mod.Probe), C.POINTER(orc_mod.Params), orc_mod.f32p, orc_mod.i32p, C.c_int, C.c_int]
    orc_det.lib.orc_render_rows(sc.h, C.byref(pr), C.byref(prm), accum.reshape(-1), np.array(rows, np.int32), len(rows), 16)
    for y in rows:
        assert_bits_equal(union[y], accum[y], f"row {y} of the 4K / 16 spp frame assembled from 8 shares")


def test_converged_image_vs_reference_pinned_checker(ptlib, orc_libm, capsys):
    """north_star's tolerance, measured against the REFERENCE-PINNED side: the checker's "libm" build is the one whose RNG,
    samplers, probe functions and make_color are bit-equal to the reference's own headers compiled here
    (tests/test_oracle_golden.py); the HIP kernels use the deterministic-math twin.  Accumulate 1024 spp (64 subframes of
    16 spp through the progressive blend, deviceProgram.cu:456-466) and require relative L2 <= 1e-3 between the GPU's
    accum_buffer and the libm checker's."""
    from optixpathtracer_amd import renderer as R

    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h, spp, sub = 48, 32, 16, 64
    report = []
    for name, model, cam, bvh in (("cornell", scenes.cornell_box(), scenes.CORNELL_CAMERA, False),
                                  ("terrain70k", scenes.voxel_terrain(n=96, target_tris=70000), scenes.TERRAIN_CAMERA, True)):
        r = _renderer(model, probe, cam, w, h)
        r.launchParams.samples_per_launch = spp
        for sf in range(sub):
            r.launchParams.frame.subframe_index = sf
            r.render()
        g = r.download(R.PT_BUF_ACCUM)[..., :3].astype(np.float64)
        sc = orc_libm.make_scene(model, bvh)
        pr = orc_libm.make_probe(probe)
        U, V, W = scenes.uvw_frame(**cam, aspect=w / h)
        accum = None
        for sf in range(sub):
            accum = orc_libm.render(sc, pr, (U, V, W), cam["eye"], w, h, spp, 8, sf, 0, accum)["accum"]
        o = accum[..., :3].astype(np.float64)
        l2 = float(np.sqrt(((g - o) ** 2).sum() / (o ** 2).sum()))
        report.append((name, l2))
        assert l2 <= 1e-3, f"{name}: relative L2 {l2:.3e} after {spp * sub} spp exceeds 1e-3"
    with capsys.disabled():
        print("\n[converged image, GPU vs reference-pinned libm checker, %d spp] " % (spp * sub) + ", ".join(f"{n}: rel L2 {v:.2e}" for n, v in report))


def test_partition_matches_host_mirror(ptlib, small_probe):
    """The library's pixel partition equals optixpathtracer_amd.multigpu.pixel_lists (the layout contract of
    the all-gather), checked through pack(): packed accum == accum gathered with the host list."""
    import torch

    from optixpathtracer_amd import multigpu
    from optixpathtracer_amd import renderer as R

    m = scenes.cornell_box()
    w, h, world = 100, 60, 4
    lists = multigpu.pixel_lists(w, h, world, 16, 8)
    for rank in (0, 3):
        r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h)
        r.setPartition(rank, world, 16, 8)
        g = _gpu_render(r, 1)
        owned, padded = r.ownedPixels()
        assert owned == len(lists[rank]) and padded == max(len(l) for l in lists)
        buf = torch.zeros((padded, 4), dtype=torch.float32, device="cuda")
        r.pack(R.PT_BUF_ACCUM, buf.data_ptr())
        px = lists[rank]
        exp = g["accum"][(px >> 16).astype(np.int64), (px & 0xFFFF).astype(np.int64)]
        assert np.array_equal(buf.cpu().numpy()[:owned], exp)


def test_c5_progressive_64_subframes_and_tonemap(ptlib, orc_det):
    """BASELINE config C5 semantics at a CPU-checkable size: 64 subframes x 1 spp progressive accumulation
    (clamp + running lerp every frame) then the toneMap.cu epilogue — bit-exact against the checker, so the
    "per-pixel L2 vs the reference image" is 0 at every subframe, not only after convergence."""
    m = scenes.voxel_terrain(n=64, target_tris=30000)
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h = 80, 45
    r = _renderer(m, probe, scenes.TERRAIN_CAMERA, w, h)
    g = _gpu_render(r, 1, subframes=64)
    o = _oracle_render(orc_det, m, probe, scenes.TERRAIN_CAMERA, w, h, 1, subframes=64)
    _compare(g, o)
    rel_l2 = np.linalg.norm(g["accum"][..., :3].astype(np.float64) - o["accum"][..., :3]) / np.linalg.norm(o["accum"][..., :3])
    assert rel_l2 <= 1e-3  # north_star's tolerance; actual value is exactly 0
    out = r.tonemapSqrt()
    ref = np.zeros(w * h, np.uint32)
    orc_det.lib.orc_tonemap_sqrt(np.ascontiguousarray(g["accum"]).reshape(-1), ref, w * h)
    assert np.array_equal(out.reshape(-1), ref)


def test_gpu_cdf_build_bit_exact(ptlib, orc_det, small_probe):
    """BuildCDF on the GPU equals the host BuildCDF (Probe.h:29-77) bit for bit, and a render
    through it equals a render through setProbe."""
    from optixpathtracer_amd.renderer import SampleRenderer, make_camera

    m = scenes.cornell_box()
    for probe in (scenes.sky_probe(512, 256), scenes.disc_probe(), scenes.constant_probe()):
        r = SampleRenderer(m)
        r.setProbeImage(probe.data)
        got = r.probeCDF()
        ref = orc_det.build_cdf(probe.data, probe.width, probe.height)
        for a, b in zip(got, ref):
            assert_bits_equal(a, b, "GPU BuildCDF")
    w, h = 64, 40
    r = SampleRenderer(m)
    r.setProbeImage(small_probe.data)
    r.resize((w, h))
    r.setCamera(make_camera(scenes.CORNELL_CAMERA, w / h))
    g = _gpu_render(r, 2)
    o = _oracle_render(orc_det, m, small_probe, scenes.CORNELL_CAMERA, w, h, 2)
    _compare(g, o)


def test_probe_8k_by_4k_maximum_size(ptlib, orc_det):
    """the largest probe the reference's scenes load is 8192 x 4096 (537 MB of texels, 2 x 134 MB of tables).  At that size:
    BuildCDF on the GPU == the host BuildCDF (Probe.h:29-77) bit for bit, and ProbeSample / ProbeEval / ProbePdf through the device's guide and
    line tables == the checker, with a sun of a few thousand texels, per-texel noise, and black rows (NaN row CDFs) in it."""
    import ctypes as C

    from optixpathtracer_amd.renderer import SampleRenderer

    W, H = 8192, 4096
    rng = np.random.default_rng(11)
    v = np.linspace(0, 1, H, dtype=np.float32)[:, None]
    lum = ((np.float32(0.1) + (np.float32(1) - v) * np.float32(0.9)) * (np.float32(0.75) + np.float32(0.5) * rng.random((H, W), dtype=np.float32))).astype(np.float32)
    lum[1000:1040, 3000:3060] = 5000.0
    lum[H - 64:] = 0.0
    lum[2000] = 0.0
    data = np.empty((H, W, 4), np.float32)
    data[..., 0] = lum; data[..., 1] = lum * np.float32(0.9); data[..., 2] = lum * np.float32(1.1); data[..., 3] = 1.0
    del lum
    ref = orc_det.build_cdf(data, W, H)
    r = SampleRenderer(scenes.cornell_box())
    r.setProbeImage(data)
    got = r.probeCDF()
    for a, b, name in zip(got, ref, ("pdfX", "cdfX", "pdfY", "cdfY")):
        assert_bits_equal(a, b, "GPU BuildCDF 8k " + name)
    del got
    probe = scenes.ProbeData(W, H, data)
    probe.pdfValuesX, probe.cdfValuesX, probe.pdfValuesY, probe.cdfValuesY = ref
    pr = orc_det.make_probe(probe)
    n = 4000
    seeds = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    g = r.evalTable(2, seeds.view(np.float32)[:, None], 9)
    want = np.zeros((n, 9), np.float32)
    d = np.zeros(3, np.float32); c = np.zeros(3, np.float32); pdf = C.c_float(); st = np.zeros(2, np.uint32)
    for i in range(n):
        orc_det.lib.orc_probe_sample(C.byref(pr), int(seeds[i]), d, c, C.byref(pdf), st)
        want[i, :3] = d; want[i, 3:6] = c; want[i, 6] = pdf.value; want[i, 7:] = st.view(np.float32)
    assert_bits_equal(g, want, "ProbeSample 8k")
    assert (want[:, 3] == 5000.0).sum() > n // 4  # the sun is found: importance sampling really goes through the tables
    dirs = rng.standard_normal((n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = np.ascontiguousarray(dirs, np.float32)
    g = r.evalTable(3, dirs, 6)
    want = np.zeros((n, 6), np.float32)
    uv = np.zeros(2, np.float32); px = np.zeros(4, np.float32)
    for i in range(n):
        orc_det.lib.orc_probe_dir_to_uv(dirs[i].copy(), uv)
        orc_det.lib.orc_probe_eval(C.byref(pr), uv, px)
        want[i, :2] = uv; want[i, 2:] = px
    assert_bits_equal(g, want, "ProbeEval 8k")
    g = r.evalTable(8, dirs, 1)[:, 0]
    want = np.array([orc_det.lib.orc_probe_pdf(C.byref(pr), dirs[i].copy()) for i in range(n)], np.float32)
    assert_bits_equal(g, want, "ProbePdf 8k")
    r.close()


@pytest.mark.parametrize("variant_name", ["SV4_VARIANT", "SV3_VARIANT"])
def test_foveated_sv4_three_launches(ptlib, orc_det, variant_name):
    """SURVEY §8f row 1 — the foveated variants' render() (HelloPathtracing_sv4_vmv23/SimplePathtracer.cpp:132-216):
    periphery at 1/4 resolution accumulating over subframes, annulus at 1/2 resolution and fovea at full
    resolution redrawn every frame; sv4 device semantics (seed from the launch index, annulus early-out, fillSize^2
    splat, tmin .01, back-face-culled occlusion rays, depth 4, exposure 4 + Reinhard + make_color).  Several frames
    with a moving gaze point, bit-exact accum_buffer and frame_buffer against the checker.  SV3_VARIANT = the sv3 directory's
    epilogue (exposure 2^3, no Reinhard in the write that wins)."""
    from optixpathtracer_amd.renderer import SampleRenderer, make_camera

    m = scenes.voxel_terrain(n=64, target_tris=30000)
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h = 192, 128
    r = SampleRenderer(m)
    r.setProbe(probe)
    # the sv3 run also squeezes the path-slot budget so that every launch is cut into several passes of launch indices
    r.setOptions(max_depth=4, max_paths=(1500 if variant_name == "SV3_VARIANT" else 0))
    r.resize((w, h))
    r.setCamera(make_camera(scenes.TERRAIN_CAMERA, w / h))
    sc = orc_det.make_scene(m, True)
    pr = orc_det.make_probe(probe)
    U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=w / h)
    accum = np.zeros((h, w, 4), np.float32)
    frame = np.zeros((h, w), np.uint32)
    for k, gaze in enumerate([(96, 64), (100, 60), (70, 80)]):
        assert r.launchParams.frame.subframe_index == k
        regs = r.foveatedRegions((w, h), gaze, k, inner_radius=14, outer_radius=44, spp=(1, 2, 4))
        variant = getattr(r, variant_name)
        r.renderFoveated(gaze, inner_radius=14, outer_radius=44, spp=(1, 2, 4), variant=variant)
        orc_det.render_regions(sc, pr, (U, V, W), scenes.TERRAIN_CAMERA["eye"], w, h, regs, variant, 4, accum, frame)
        g_acc = r.download(R_ACCUM)
        g_frm = r.download(R_FRAME)
        assert_bits_equal(g_acc, accum, f"foveated accum_buffer, frame {k}")
        assert np.array_equal(g_frm, frame), f"foveated frame_buffer, frame {k}"
    # the three regions really differ in sampling density: fovea pixels are all distinct renders, periphery is 4x4 blocks
    blk = g_acc[0:4, 0:4, :3].reshape(-1, 3)
    assert (blk == blk[0]).all()
    st = r.stats()
    assert st["radiance_rays"] > 0 and st["paths"] == (w // 4) * (h // 4) * 1 + 46 * 46 * 2 + 30 * 30 * 4


def test_foveated_frames_in_flight(ptlib, orc_det):
    """Six foveated frames with a moving gaze and nothing read back in between, three frames in flight (all launches of a frame
    on that frame's stream, resolves chained from frame to frame), then two uniform frames on top: final buffers against the
    checker and against the synchronous run."""
    from optixpathtracer_amd.renderer import SampleRenderer, make_camera

    m = scenes.voxel_terrain(n=64, target_tris=30000)
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h = 192, 128
    # the regions stay inside the image: where a splat crosses the border the reference clamps the pixel index (sv4 deviceProgram.cu:528-534),
    # so several launch indices write the same border pixel — a data race of the reference itself, kept as it is
    gazes = [(96, 64), (100, 60), (70, 80), (50, 50), (140, 70), (96, 64)]
    out = {}
    for fif in (0, 2, 3):
        r = SampleRenderer(m)
        r.setProbe(probe)
        r.setOptions(max_depth=4, frames_in_flight=fif)
        r.resize((w, h))
        r.setCamera(make_camera(scenes.TERRAIN_CAMERA, w / h))
        for gaze in gazes:
            r.renderFoveated(gaze, inner_radius=14, outer_radius=44, spp=(1, 2, 4), variant=r.SV4_VARIANT)
        fov = (r.download(R_ACCUM), r.download(R_FRAME))
        r.launchParams.samples_per_launch = 2
        for sf in (6, 7):  # the uniform renderer continues the accumulation in the same pipeline
            r.launchParams.frame.subframe_index = sf
            r.render()
        out[fif] = fov + (r.download(R_ACCUM), r.stats()["frames"])
    sc = orc_det.make_scene(m, True)
    pr = orc_det.make_probe(probe)
    U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=w / h)
    accum = np.zeros((h, w, 4), np.float32)
    frame = np.zeros((h, w), np.uint32)
    r0 = SampleRenderer(m)
    for k, gaze in enumerate(gazes):
        regs = r0.foveatedRegions((w, h), gaze, k, inner_radius=14, outer_radius=44, spp=(1, 2, 4))
        orc_det.render_regions(sc, pr, (U, V, W), scenes.TERRAIN_CAMERA["eye"], w, h, regs, r0.SV4_VARIANT, 4, accum, frame)
    for fif in (0, 2, 3):
        assert_bits_equal(out[fif][0], accum, f"foveated accum_buffer after six frames, frames_in_flight={fif}")
        assert np.array_equal(out[fif][1], frame)
    for fif in (2, 3):
        assert_bits_equal(out[fif][2], out[0][2], "uniform frames on top of the foveated ones")
        assert out[fif][3] == out[0][3] == 8


R_ACCUM, R_FRAME = 0, 1


def test_foveated_sv_sv2_initial_depth_and_aov_writes(ptlib, orc_det):
    """The sv / sv2 directories (one device program): prd.depth starts at 1 with the cutoff `depth >= 3`
    (HelloPathtracing_sv/deviceProgram.cu:428,483) — all radiance goes to indirectLight (a different float association than
    direct + indirect), nothing is added to normal/albedo — canonical tmin/occlusion/make_color, and every launch also writes
    normal_buffer (zeros, w = 1), color_buffer and albedo_buffer (:553-555).  Three frames of the sv host schedule's three
    launches with a moving gaze: all five buffers bit-exact against the checker."""
    from optixpathtracer_amd import renderer as R
    from optixpathtracer_amd.renderer import SampleRenderer, make_camera

    m = scenes.voxel_terrain(n=64, target_tris=30000)
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h = 192, 128
    r = SampleRenderer(m)
    r.setProbe(probe)
    r.setOptions(max_depth=SampleRenderer.SV_MAX_DEPTH)
    r.resize((w, h))
    r.setCamera(make_camera(scenes.TERRAIN_CAMERA, w / h))
    sc = orc_det.make_scene(m, True)
    pr = orc_det.make_probe(probe)
    U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=w / h)
    accum = np.zeros((h, w, 4), np.float32)
    frame = np.zeros((h, w), np.uint32)
    aov = [np.zeros((h, w, 4), np.float32) for _ in range(3)]
    variant = SampleRenderer.SV_VARIANT
    for k, gaze in enumerate([(96, 64), (100, 60), (70, 80)]):
        regs = r.foveatedRegions((w, h), gaze, k, inner_radius=14, outer_radius=44, spp=(1, 2, 4))
        r.renderFoveated(gaze, inner_radius=14, outer_radius=44, spp=(1, 2, 4), variant=variant)
        orc_det.render_regions(sc, pr, (U, V, W), scenes.TERRAIN_CAMERA["eye"], w, h, regs, variant, SampleRenderer.SV_MAX_DEPTH, accum, frame, aov=aov)
        assert_bits_equal(r.download(R.PT_BUF_ACCUM), accum, f"sv accum_buffer, frame {k}")
        assert np.array_equal(r.download(R.PT_BUF_FRAME), frame), f"sv frame_buffer, frame {k}"
        assert_bits_equal(r.download(R.PT_BUF_NORMAL), aov[0], f"sv normal_buffer, frame {k}")
        assert_bits_equal(r.download(R.PT_BUF_COLOR), aov[1], f"sv color_buffer, frame {k}")
        assert_bits_equal(r.download(R.PT_BUF_ALBEDO), aov[2], f"sv albedo_buffer, frame {k}")
    n = r.download(R.PT_BUF_NORMAL)
    assert (n[..., :3] == 0).all() and (n[..., 3] == 1).all()  # depth never 0: the AOVs stay empty, as in the reference
    # initial depth 0 with AOV writes (the canonical raygen in launch form): first-hit sums are written
    v2 = dict(variant, initial_depth=0)
    accum[:] = 0; frame[:] = 0
    for a in aov:
        a[:] = 0
    r2 = SampleRenderer(m)
    r2.setProbe(probe)
    r2.setOptions(max_depth=4)
    r2.resize((w, h))
    r2.setCamera(make_camera(scenes.TERRAIN_CAMERA, w / h))
    regs = r2.foveatedRegions((w, h), (96, 64), 0, inner_radius=14, outer_radius=44, spp=(1, 2, 4))
    r2.renderFoveated((96, 64), inner_radius=14, outer_radius=44, spp=(1, 2, 4), variant=v2)
    orc_det.render_regions(sc, pr, (U, V, W), scenes.TERRAIN_CAMERA["eye"], w, h, regs, v2, 4, accum, frame, aov=aov)
    assert_bits_equal(r2.download(R.PT_BUF_ACCUM), accum, "accum_buffer, initial depth 0 + AOVs")
    assert_bits_equal(r2.download(R.PT_BUF_NORMAL), aov[0], "normal_buffer, initial depth 0 + AOVs")
    assert_bits_equal(r2.download(R.PT_BUF_ALBEDO), aov[2], "albedo_buffer, initial depth 0 + AOVs")
    assert (aov[0][..., :3] != 0).any()


def test_textured_meshes(ptlib, orc_det, sched):
    """deviceProgram.cu:512-523 + createTextures (SimplePathtracer.cpp:603-654): albedo replaced by a wrap/bilinear
    tex2D of an RGBA8 texture at the barycentric texcoord; a mesh with a texture id but no texcoords keeps its colour."""
    import ctypes as C

    from optixpathtracer_amd.renderer import SampleRenderer

    m = scenes.textured_scene()
    probe = scenes.sky_probe(256, 128).BuildCDF()
    # the sampler itself
    r = SampleRenderer(m)
    rng = np.random.default_rng(21)
    st = rng.uniform(-3, 3, (20000, 2)).astype(np.float32)
    st[:8] = [[0, 0], [1, 1], [0.5, 0.5], [-0.25, 1.75], [1.0 / 128, 1.0 / 64], [0.999999, 0.000001], [-1, -1], [2, -3]]
    g = r.evalTable(7, st, 4)
    tex = m.textures[0].pixel
    ref = np.zeros((len(st), 4), np.float32)
    out = np.zeros(4, np.float32)
    for i in range(len(st)):
        orc_det.lib.orc_tex2d(tex.reshape(-1), tex.shape[1], tex.shape[0], float(st[i, 0]), float(st[i, 1]), out)
        ref[i] = out
    assert_bits_equal(g, ref, "tex2D wrap/bilinear")
    assert 0.0 <= g.min() and g.max() <= 1.0
    # whole render
    w, h = 160, 100
    cam = dict(eye=(3.0, 2.5, -4.5), lookat=(0.0, 0.6, 0.5), up=(0.0, 1.0, 0.0), fovY=45.0)
    rr = _renderer(m, probe, cam, w, h)
    gg = _gpu_render(rr, 3, subframes=2)
    _check_sched(gg, sched)
    oo = _oracle_render(orc_det, m, probe, cam, w, h, 3, subframes=2, use_bvh=False)
    _compare(gg, oo)
    # the texture really shows up in the first-hit albedo buffer: many distinct albedos, not 3 material colours
    alb = gg["albedo"][..., :3].reshape(-1, 3)
    assert len(np.unique(alb.round(4), axis=0)) > 100


@pytest.mark.parametrize("spp,nsub", [(4, 12), (1, 64)])
def test_fullsize_c5_progressive_rows(ptlib, orc_det, spp, nsub):
    """C5 at full size: 1 M triangles, 1920x1080, progressive accumulation — the literal configuration (64 subframes x 1 spp,
    then the HIP tone-map epilogue) and 12 subframes x 4 spp; rows of the final accum_buffer are re-rendered by the checker
    through all subframes and must match bit for bit (per-pixel L2 vs the checker: exactly 0), the epilogue likewise."""
    import ctypes as C

    from oracle import orc as orc_mod

    m = scenes.voxel_terrain()
    probe = scenes.sky_probe(2048, 1024).BuildCDF()
    w, h = 1920, 1080
    r = _renderer(m, probe, scenes.TERRAIN_CAMERA, w, h)
    g = _gpu_render(r, spp, subframes=nsub)
    sc = orc_det.make_scene(m, True)
    pr = orc_det.make_probe(probe)
    U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=w / h)
    rows = np.array([0, 271, 540, 803, 1079], np.int32)
    accum = np.zeros((h, w, 4), np.float32)
    orc_det.lib.orc_render_rows.argtypes = [C.c_void_p, C.POINTER(orc_mod.Probe), C.POINTER(orc_mod.Params), orc_mod.f32p, orc_mod.i32p, C.c_int, C.c_int]
    for sf in range(nsub):
        prm = orc_mod.Params()
        prm.width, prm.height, prm.subframe_index, prm.samples_per_launch, prm.max_depth, prm.bsdf_mode = w, h, sf, spp, 8, 0
        for dst, src in ((prm.eye, scenes.TERRAIN_CAMERA["eye"]), (prm.U, U), (prm.V, V), (prm.W, W)):
            for k in range(3):
                dst[k] = float(src[k])
        orc_det.lib.orc_render_rows(sc.h, C.byref(pr), C.byref(prm), accum.reshape(-1), rows, len(rows), 16)
    for y in rows:
        assert_bits_equal(g["accum"][y], accum[y], f"row {y} after {nsub} subframes")
    tm = r.tonemapSqrt()  # toneMap.cu epilogue on the accumulated frame
    ref = np.zeros(w, np.uint32)
    for y in rows:
        orc_det.lib.orc_tonemap_sqrt(np.ascontiguousarray(accum[y]).reshape(-1), ref, w)
        assert np.array_equal(tm[y], ref), f"tone-mapped row {y}"
    assert (g["accum"][..., :3] <= 50.0 + 1e-3).all()  # subframe 0 is unclamped (sun radiance 50), later ones clamp to 10


def test_denoise_atrous_bit_exact(ptlib, orc_det, small_probe):
    """f4: the AOV-guided a-trous pass (pt_denoise) against the checker's restatement, bit for bit: rendered Cornell AOVs
    (noisy 1-spp colour, first-hit normal and albedo), both inputs, 0..5 iterations, and the two rgba8 epilogues."""
    from optixpathtracer_amd import renderer as R

    m = scenes.cornell_box()
    w, h = 160, 96
    r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h)
    g = _gpu_render(r, 1, subframes=3)
    assert np.isfinite(g["color"]).all()
    for inp, src in ((R.PT_BUF_COLOR, g["color"]), (R.PT_BUF_ACCUM, g["accum"])):
        for it in (0, 1, 2, 5):
            d, ms = r.denoise(iterations=it, sigma_color=0.8, sigma_normal=0.3, sigma_albedo=0.15, input=inp)
            o = orc_det.denoise(src, g["normal"], g["albedo"], it, 0.8, 0.3, 0.15)
            assert_bits_equal(d, o, f"denoised, input {inp}, {it} iterations")
    d, _ = r.denoise(iterations=5, epilogue=1)
    o = orc_det.denoise(g["color"], g["normal"], g["albedo"], 5, 1.0, 0.25, 0.1)
    assert_bits_equal(d, o, "denoised (defaults)")
    import ctypes as C
    px = np.empty(w * h, np.uint32)
    orc_det.lib.orc_tonemap_sqrt.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    orc_det.lib.orc_tonemap_sqrt(o.ctypes.data, px.ctypes.data, w * h)
    assert np.array_equal(r.download(R.PT_BUF_FRAME).reshape(-1), px), "computeFinalPixelColors of the denoised buffer"
    # the filter must actually do something useful: less variance inside the flat back wall, walls' colour edge kept
    noisy, den = g["color"][..., :3], d[..., :3]
    flat = (np.abs(g["normal"][..., 2] + 1.0) < 1e-6) & (np.abs(g["albedo"][..., 0] - g["albedo"][..., 1]) < 1e-6) & (g["albedo"][..., 0] > 0)
    assert flat.sum() > 500 and den[flat].var(0).sum() < 0.5 * noisy[flat].var(0).sum()
    with pytest.raises(RuntimeError):
        r.denoise(iterations=9)
    with pytest.raises(RuntimeError):
        r.denoise(sigma_color=0.0)


def test_denoise_edge_sizes_and_synthetic(ptlib, orc_det, small_probe):
    """a-trous on tiny / odd frames (taps falling outside the image at every spacing) with synthetic AOVs."""
    from optixpathtracer_amd import renderer as R

    m = scenes.cornell_box()
    rng = np.random.default_rng(5)
    for (w, h) in ((1, 1), (3, 5), (33, 9), (70, 41)):
        r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h)
        g = _gpu_render(r, 2)
        d, _ = r.denoise(iterations=5, sigma_color=2.0, sigma_normal=0.5, sigma_albedo=0.5, input=R.PT_BUF_ACCUM)
        o = orc_det.denoise(g["accum"], g["normal"], g["albedo"], 5, 2.0, 0.5, 0.5)
        assert_bits_equal(d, o, f"denoised {w}x{h}")
        assert np.isfinite(d).all()


def test_traversal_stack_spill_path(ptlib, orc_det, small_probe, monkeypatch):
    """The traversal stack keeps 12 levels in LDS and spills deeper ones to a per-lane global area that the test scenes
    (depth <= 7) never reach.  PT_STACK_LDS_SKIP moves the LDS/global boundary down so that the spill path runs on
    every ray with 2 or more pending groups; the image and the batch queries must not change by a bit."""
    m = scenes.voxel_terrain(n=96, target_tris=70000)
    w, h = 160, 96
    ref = _gpu_render(_renderer(m, small_probe, scenes.TERRAIN_CAMERA, w, h), 2)
    rng = np.random.default_rng(21)
    rays = _random_rays(rng, 60000, -110, 110)
    from optixpathtracer_amd.renderer import SampleRenderer

    (t0, p0), _ = SampleRenderer(m).trace(rays)
    for skip in ("11", "12"):
        monkeypatch.setenv("PT_STACK_LDS_SKIP", skip)
        r = _renderer(m, small_probe, scenes.TERRAIN_CAMERA, w, h)
        g = _gpu_render(r, 2)
        _compare(g, ref)
        (t1, p1), _ = r.trace(rays)
        assert np.array_equal(p0, p1)
        assert_bits_equal(t0, t1, f"closest-hit t with PT_STACK_LDS_SKIP={skip}")
        occ0, _ = SampleRenderer(m).trace(rays, any_hit=True)
    monkeypatch.delenv("PT_STACK_LDS_SKIP")
    o = _oracle_render(orc_det, m, small_probe, scenes.TERRAIN_CAMERA, w, h, 2, use_bvh=True)
    _compare(ref, o)


def test_cpu_traversal_of_the_exported_product_tree(ptlib, orc_det, small_probe):
    """pt_export_bvh hands the 8-wide tree to the host; the checker's scalar traversal of THAT tree (oracle/pt_oracle.c
    bvh8_traverse — what bench.py times as the CPU baseline "on the same BVH") must (a) find every triangle of the scene exactly
    once in the leaf array, (b) agree with brute force over all triangles on closest hits (t and primitive bit-exact) and on
    occlusion, and (c) render the same image as the GPU."""
    from optixpathtracer_amd.renderer import SampleRenderer

    m = scenes.voxel_terrain(n=96, target_tris=70000)
    r = _renderer(m, small_probe, scenes.TERRAIN_CAMERA, 96, 64)
    nodes, tris = r.exportBVH()
    st = r.stats()
    assert len(nodes) == st["bvh_nodes"] and len(tris) == m.num_triangles
    prim = tris[:, 9].view(np.int32)
    assert np.array_equal(np.sort(prim), np.arange(m.num_triangles))
    rng = np.random.default_rng(77)
    rays = _random_rays(rng, 3000, -110, 110)
    brute = orc_det.make_scene(m, False)
    t0, p0 = orc_det.trace_closest(brute, rays)
    o0 = orc_det.trace_any(brute, rays)
    sc = orc_det.make_scene(m, False)
    orc_det.set_bvh8(sc, nodes, tris)
    t1, p1 = orc_det.trace_closest(sc, rays)
    o1 = orc_det.trace_any(sc, rays)
    assert np.array_equal(p0, p1) and np.array_equal(o0, o1)
    assert_bits_equal(t0, t1, "closest-hit t: CPU traversal of the exported tree vs brute force")
    (tg, pg), _ = r.trace(rays)
    assert np.array_equal(pg, p1)
    g = _gpu_render(r, 2)
    U, V, W = scenes.uvw_frame(**scenes.TERRAIN_CAMERA, aspect=96 / 64)
    o = orc_det.render(sc, orc_det.make_probe(small_probe), (U, V, W), scenes.TERRAIN_CAMERA["eye"], 96, 64, 2)
    _compare(g, o)
    with pytest.raises(RuntimeError, match="buffer sizes"):
        r._ck(r._L.pt_export_bvh(r._ctx, nodes.ctypes.data, 16, tris.ctypes.data, tris.nbytes, None, None), "pt_export_bvh")


def test_work_stealing_results_do_not_depend_on_timing(ptlib, small_probe):
    """Which lane steals which subtree, and when, depends on run-time timing; the image must not.  The same frame is rendered
    several hundred times (small frames, where most launches are all tail and stealing is busiest, and full-size frames of the
    1 M-triangle scene) and every repetition must reproduce the first bit for bit, ray counts included."""
    import zlib

    from optixpathtracer_amd import renderer as R

    for model, (w, h), spp, reps in ((scenes.voxel_terrain(n=96, target_tris=70000), (160, 96), 2, 300),
                                     (scenes.voxel_terrain(), (1920, 1080), 4, 40)):
        r = _renderer(model, small_probe, scenes.TERRAIN_CAMERA, w, h)
        r.launchParams.samples_per_launch = spp
        r.launchParams.frame.subframe_index = 0
        first = None
        for k in range(reps):
            r.render()
            st = r.stats()
            sig = (zlib.crc32(r.download(R.PT_BUF_ACCUM).tobytes()), st["radiance_rays"], st["shadow_rays"])
            if first is None:
                first = sig
            assert sig == first, f"repetition {k} of the {w}x{h} frame differs: {sig} vs {first}"


def test_traversal_stack_overflow_fails_loudly(ptlib, small_probe, monkeypatch):
    """A tree deeper than the traversal stack must never give a silently wrong image.  (a) pt_create compares the wide tree's
    level count with the stack capacity and refuses the scene with PT_ERR_UNSUPPORTED (-4); (b) with that check bypassed
    (test hook PT_STACK_NOCHECK) the traversal kernel's push reports the overflow and pt_render / pt_trace return
    PT_ERR_UNSUPPORTED.  The LBVH's 58-bit keys bound real trees at far fewer levels than the 64 the stack holds, so the
    capacity is lowered by the test hooks PT_STACK_LDS_SKIP / PT_STACK_CAP to reach both paths."""
    from optixpathtracer_amd.renderer import SampleRenderer

    m = scenes.voxel_terrain(n=96, target_tris=70000)
    w, h = 96, 64
    monkeypatch.setenv("PT_STACK_LDS_SKIP", "11")
    monkeypatch.setenv("PT_STACK_CAP", "2")  # 1 level in LDS + 1 spill level; the tree has 5 or more
    with pytest.raises(RuntimeError, match=r"pt_create failed \(-4\).*levels.*traversal stack holds 2"):
        SampleRenderer(m)
    monkeypatch.setenv("PT_STACK_NOCHECK", "1")
    r = _renderer(m, small_probe, scenes.TERRAIN_CAMERA, w, h)
    r.launchParams.samples_per_launch = 1
    with pytest.raises(RuntimeError, match=r"\(-4\).*traversal stack overflow"):
        r.render()
    rays = _random_rays(np.random.default_rng(5), 20000, -110, 110)
    with pytest.raises(RuntimeError, match=r"\(-4\).*traversal stack overflow"):
        r.trace(rays)
    # the full-size stack on the same scene: no fault, and the context stays usable after a reported fault
    monkeypatch.delenv("PT_STACK_CAP")
    monkeypatch.delenv("PT_STACK_NOCHECK")
    monkeypatch.delenv("PT_STACK_LDS_SKIP")
    r2 = _renderer(m, small_probe, scenes.TERRAIN_CAMERA, w, h)
    _gpu_render(r2, 1)
    assert r2.stats()["bvh_levels"] >= 3


def test_cxx_facade_demo_matches_python(ptlib, small_probe, tmp_path):
    """The C++ facade (csrc/SampleRenderer.h) over the C ABI, compiled with a host compiler alone and run as its own process
    (no Python, no torch): the reference application's call sequence on the main.cpp two-box scene gives the same bits as
    the Python facade."""
    import os
    import shutil
    import struct
    import subprocess

    from conftest import ROOT

    if not shutil.which("g++"):
        pytest.skip("no host C++ compiler")
    m = scenes.two_box_scene(shadow_catcher=False)
    cam = scenes.TWO_BOX_CAMERA
    w, h, spp, nsub = 96, 64, 2, 3
    scene = tmp_path / "scene.bin"
    with open(scene, "wb") as f:
        f.write(struct.pack("<I", len(m.meshes)))
        for mesh in m.meshes:
            v = np.ascontiguousarray(mesh.vertex, np.float32)
            idx = np.ascontiguousarray(mesh.index, np.uint32)
            f.write(struct.pack("<II", len(v), len(idx)))
            f.write(np.asarray(mesh.material).tobytes())
            f.write(v.tobytes())
            f.write(idx.tobytes())
        f.write(struct.pack("<II", small_probe.width, small_probe.height))
        f.write(np.ascontiguousarray(small_probe.data, np.float32).tobytes())
        f.write(np.asarray(list(cam["eye"]) + list(cam["lookat"]) + list(cam["up"]) + [cam["fovY"]], np.float32).tobytes())
        f.write(struct.pack("<IIII", w, h, spp, nsub))
    exe = tmp_path / "facade_demo"
    libdir = os.path.join(ROOT, "optixpathtracer_amd")
    subprocess.run(["g++", "-std=c++17", "-I", ROOT, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "facade_demo.cpp"),
                    "-L", libdir, "-lptamd", "-L", "/opt/rocm/lib", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    out = tmp_path / "out.bin"
    res = subprocess.run([str(exe), str(scene), str(out)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    raw = np.fromfile(out, np.uint8)
    frame = raw[: w * h * 4].view(np.uint32).reshape(h, w)
    accum = raw[w * h * 4 :].view(np.float32).reshape(h, w, 4)
    g = _gpu_render(_renderer(m, small_probe, cam, w, h), spp, subframes=nsub)
    assert_bits_equal(accum, g["accum"], "accum_buffer from the C++ process")
    assert np.array_equal(frame, g["frame"])
    # the same loop with three frames in flight (SampleRenderer::setFramesInFlight): render() no longer waits for its own frame
    outp = tmp_path / "outp.bin"
    res = subprocess.run([str(exe), str(scene), str(outp), "0", "3"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    raw = np.fromfile(outp, np.uint8)
    assert np.array_equal(raw[: w * h * 4].view(np.uint32).reshape(h, w), g["frame"])
    assert_bits_equal(raw[w * h * 4 :].view(np.float32).reshape(h, w, 4), g["accum"], "accum_buffer from the pipelined C++ process")
    # the same application on MultiSampleRenderer: 3 contexts in one process (all on device 0 here), frame assembled by the
    # library's own exchange; rank 2's accum_buffer after an explicit gather
    out3 = tmp_path / "out3.bin"
    res = subprocess.run([str(exe), str(scene), str(out3), "3"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    raw = np.fromfile(out3, np.uint8)
    assert np.array_equal(raw[: w * h * 4].view(np.uint32).reshape(h, w), g["frame"])
    assert_bits_equal(raw[w * h * 4 :].view(np.float32).reshape(h, w, 4), g["accum"], "accum_buffer from the 3-context C++ process")
    # ... and with three frames in flight: every render(pixels) hands over the previous frame while the next one renders, flush() the last
    out3p = tmp_path / "out3p.bin"
    res = subprocess.run([str(exe), str(scene), str(out3p), "3", "3"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    raw = np.fromfile(out3p, np.uint8)
    assert np.array_equal(raw[: w * h * 4].view(np.uint32).reshape(h, w), g["frame"])
    assert_bits_equal(raw[w * h * 4 :].view(np.float32).reshape(h, w, 4), g["accum"], "accum_buffer from the 3-context C++ process, frames in flight")


def test_multi_context_one_process(ptlib, small_probe, monkeypatch):
    """pt_create_multi: N contexts of ONE process render interleaved tiles of one frame concurrently and assemble it with the
    library's own exchange.  On this one-GPU box all contexts live on device 0 (direct device-to-device copies; RCCL refuses
    duplicate devices) — the partition, the enqueue-all-then-wait schedule, pack / exchange / unpack and the per-rank buffers
    are the multi-GPU code path.  Every rank's assembled buffers must equal the single-context render bit for bit; the
    forced-RCCL run (one rank, its own communicator) exercises the ncclAllGather branch."""
    from optixpathtracer_amd import renderer as R

    for model, cam, (w, h), spp in ((scenes.cornell_box(), scenes.CORNELL_CAMERA, (100, 60), 2),
                                    (scenes.voxel_terrain(n=96, target_tris=70000), scenes.TERRAIN_CAMERA, (200, 120), 3)):
        ref = _gpu_render(_renderer(model, small_probe, cam, w, h), spp, subframes=2)
        for ndev in (3, 8):
            mr = R.MultiRenderer(model, devices=[0] * ndev)
            assert mr.world == ndev
            mr.setProbe(small_probe)
            mr.resize((w, h), tile=(16, 8))
            mr.setCamera(R.make_camera(cam, w / h))
            mr.launchParams.samples_per_launch = spp
            mr.gather_mask = sum(1 << b for b in (R.PT_BUF_ACCUM, R.PT_BUF_FRAME, R.PT_BUF_COLOR, R.PT_BUF_NORMAL, R.PT_BUF_ALBEDO))
            for sf in range(2):
                mr.launchParams.frame.subframe_index = sf
                mr.render()
            st = mr.stats()
            assert st["exchange"] == "peer_copy" and st["ndev"] == ndev
            assert st["radiance_rays"] == ref["stats"]["radiance_rays"] and st["shadow_rays"] == ref["stats"]["shadow_rays"]
            assert st["paths"] == w * h * spp
            for rank in (0, ndev - 1):
                assert_bits_equal(mr.download(R.PT_BUF_ACCUM, rank), ref["accum"], f"accum_buffer assembled on rank {rank} of {ndev}")
                assert_bits_equal(mr.download(R.PT_BUF_NORMAL, rank), ref["normal"], f"normal_buffer on rank {rank}")
                assert_bits_equal(mr.download(R.PT_BUF_ALBEDO, rank), ref["albedo"], f"albedo_buffer on rank {rank}")
                assert np.array_equal(mr.download(R.PT_BUF_FRAME, rank), ref["frame"])
            mr.close()
    # a throughput loop: nothing handed over per frame, three frames in flight on every context, one gather at the end
    ref5 = _gpu_render(_renderer(model, small_probe, cam, w, h), spp, subframes=5)
    mr = R.MultiRenderer(model, devices=[0] * 3)
    mr.setOptions(frames_in_flight=3)
    mr.setProbe(small_probe)
    mr.resize((w, h), tile=(16, 8))
    mr.setCamera(R.make_camera(cam, w / h))
    mr.launchParams.samples_per_launch = spp
    mr.gather_mask = 0
    for sf in range(5):
        mr.launchParams.frame.subframe_index = sf
        mr.render()
    mr.gather(R.PT_BUF_ACCUM)
    mr.gather(R.PT_BUF_FRAME)
    for rank in (0, 2):
        assert_bits_equal(mr.download(R.PT_BUF_ACCUM, rank), ref5["accum"], f"accum_buffer after the pipelined loop, rank {rank}")
        assert np.array_equal(mr.download(R.PT_BUF_FRAME, rank), ref5["frame"])
    st = mr.stats()
    assert st["frames"] == 3 * 5 and st["radiance_rays"] == ref5["stats"]["radiance_rays"]
    mr.close()
    # RCCL branch: a single-rank communicator (the only RCCL configuration a one-GPU box can run)
    monkeypatch.setenv("PT_MULTI_EXCHANGE", "rccl")
    model, cam, (w, h), spp = scenes.cornell_box(), scenes.CORNELL_CAMERA, (100, 60), 2
    ref = _gpu_render(_renderer(model, small_probe, cam, w, h), spp)
    mr = R.MultiRenderer(model, devices=[0])
    mr.setProbe(small_probe)
    mr.resize((w, h))
    mr.setCamera(R.make_camera(cam, w / h))
    mr.launchParams.samples_per_launch = spp
    out = np.zeros((h, w), np.uint32)
    mr.render(out)
    assert mr.stats()["exchange"] == "rccl"
    assert np.array_equal(out, ref["frame"])
    mr.gather(R.PT_BUF_ACCUM)
    assert_bits_equal(mr.download(R.PT_BUF_ACCUM), ref["accum"], "accum_buffer through ncclAllGather")
    mr.close()
    # error behaviour of the multi-context layer: status + message naming the rank, never a crash
    with pytest.raises(RuntimeError, match=r"pt_create_multi failed \(-1\).*rank 1.*bad device ordinal"):
        R.MultiRenderer(model, devices=[0, 99])
    m3 = R.MultiRenderer(model, devices=[0, 0])
    with pytest.raises(RuntimeError, match="not resized"):
        m3.gather(R.PT_BUF_FRAME)
    with pytest.raises(RuntimeError, match=r"rank 0.*no probe set"):
        m3.resize((32, 16))
        m3.render()
    m3.close()
    monkeypatch.setenv("PT_MULTI_EXCHANGE", "rccl")
    with pytest.raises(RuntimeError, match="distinct devices"):
        m2 = R.MultiRenderer(model, devices=[0, 0])
        m2.setProbe(small_probe)
        m2.resize((w, h))
        m2.setCamera(R.make_camera(cam, w / h))
        m2.render()


def test_multi_context_foveated_launches(ptlib):
    """The foveated launches on a partitioned image (pt_render_regions no longer refuses world != 1): a launch index is traced
    by every rank that owns a pixel of its splat, each rank writes only its own pixels, and the assembled accum_buffer /
    frame_buffer equal the single-context frames bit for bit over several frames with a moving gaze (periphery accumulating,
    annulus and fovea redrawn, odd offsets so that 2x2 splats straddle tile borders)."""
    from optixpathtracer_amd import renderer as R

    m = scenes.voxel_terrain(n=64, target_tris=30000)
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h = 192, 128
    single = R.SampleRenderer(m)
    single.setProbe(probe)
    single.setOptions(max_depth=4)
    single.resize((w, h))
    single.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
    mr = R.MultiRenderer(m, devices=[0, 0, 0])
    mr.setProbe(probe)
    mr.setOptions(max_depth=4)
    mr.resize((w, h), tile=(16, 8))
    mr.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
    mr.gather_mask = (1 << R.PT_BUF_ACCUM) | (1 << R.PT_BUF_FRAME)
    for k, gaze in enumerate([(96, 64), (101, 59), (71, 80)]):
        single.renderFoveated(gaze, inner_radius=14, outer_radius=44, spp=(1, 2, 4))
        mr.renderFoveated(gaze, inner_radius=14, outer_radius=44, spp=(1, 2, 4))
        for rank in (0, 2):
            assert_bits_equal(mr.download(R.PT_BUF_ACCUM, rank), single.download(R.PT_BUF_ACCUM), f"foveated accum_buffer, frame {k}, rank {rank}")
            assert np.array_equal(mr.download(R.PT_BUF_FRAME, rank), single.download(R.PT_BUF_FRAME)), f"foveated frame_buffer, frame {k}"
    assert mr.stats()["radiance_rays"] >= single.stats()["radiance_rays"]  # straddling splats are traced by both neighbours


def test_async_shadow_schedule_matches_oracle(ptlib, orc_det, small_probe):
    """split_shadow = 2: per-bounce shadow records traced on side streams, visibility bits, contributions summed in bounce
    order by the resolve.  Same bits as the checker for whole frames, progressive accumulation, sample/pixel chunking,
    depth cutoffs, the Lambert mode, the foveated launches, and (falling back to the synchronous schedule) a shadow catcher."""
    m = scenes.cornell_box()
    w, h = 96, 64
    o = _oracle_render(orc_det, m, small_probe, scenes.CORNELL_CAMERA, w, h, 5)
    for max_paths in (0, 1000, 64):
        r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h, max_paths=max_paths, split_shadow=2)
        _compare(_gpu_render(r, 5), o)
    o = _oracle_render(orc_det, m, small_probe, scenes.CORNELL_CAMERA, w, h, 2, subframes=3, max_depth=3, bsdf_mode=1)
    r = _renderer(m, small_probe, scenes.CORNELL_CAMERA, w, h, max_depth=3, bsdf_mode=1, split_shadow=2)
    _compare(_gpu_render(r, 2, subframes=3), o)
    t = scenes.voxel_terrain(n=96, target_tris=70000)
    probe = scenes.sky_probe(512, 256).BuildCDF()
    r = _renderer(t, probe, scenes.TERRAIN_CAMERA, 160, 90, split_shadow=2)
    g = _gpu_render(r, 4)
    _compare(g, _oracle_render(orc_det, t, probe, scenes.TERRAIN_CAMERA, 160, 90, 4))
    assert g["stats"]["shadow_launches"] == g["stats"]["shade_launches"]
    # foveated launches (sv4 settings, depth 4)
    r = _renderer(t, probe, scenes.TERRAIN_CAMERA, 160, 90, max_depth=4, split_shadow=2)
    r2 = _renderer(t, probe, scenes.TERRAIN_CAMERA, 160, 90, max_depth=4)
    for k, gaze in enumerate([(80, 45), (60, 50)]):
        r.renderFoveated(gaze, inner_radius=12, outer_radius=36, spp=(1, 2, 4))
        r2.renderFoveated(gaze, inner_radius=12, outer_radius=36, spp=(1, 2, 4))
        assert_bits_equal(r.download(R_ACCUM), r2.download(R_ACCUM), f"foveated accum, frame {k}")
        assert np.array_equal(r.download(R_FRAME), r2.download(R_FRAME))
    # shadow catcher: the asynchronous schedule does not apply, the result must still be the reference's
    c = scenes.two_box_scene(shadow_catcher=True)
    r = _renderer(c, small_probe, scenes.TWO_BOX_CAMERA, 96, 64, split_shadow=2)
    _compare(_gpu_render(r, 3), _oracle_render(orc_det, c, small_probe, scenes.TWO_BOX_CAMERA, 96, 64, 3))


def test_bench_line_contract(ptlib):
    """bench.py as the driver runs it (one process, default workload, few steps): ONE JSON line with the contract's keys, the
    metric's configuration, a roofline and a CPU-baseline object, and figures that are consistent with each other."""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT

    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2"], capture_output=True, text=True, timeout=400)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), (key, d[key])
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "Mrays/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "strong"
    assert d["config"]["workload"] == "c3_terrain1M_1080p_4spp_d8" and (d["config"]["width"], d["config"]["height"], d["config"]["spp"], d["config"]["max_depth"]) == (1920, 1080, 4, 8)
    assert "model" not in d["config"]
    # value = rays / time of the timed frames
    assert abs(d["value"] - d["rays_per_frame"] / d["ms_per_step"] / 1e3) / d["value"] < 1e-3
    # the headline is the reference's semantics: every frame a device-synchronised pt_render; the other schedules ride along
    assert d["frames_in_flight"] == 1 and d["subframes_per_batch"] == 1
    assert d["mrays_per_s_pipelined"] > 0.8 * d["value"] and d["ms_per_frame_pipelined"] > 0 and d["batched"] is None
    assert d["n_ranks_seen"] == 1 and d["single_frame_launches"]["ms_per_frame"] == d["ms_per_step"]
    sm = d["step_ms"]
    assert sm["min"] <= sm["median"] <= sm["max"] and abs(sm["median"] - d["ms_per_step"]) < 0.25 * d["ms_per_step"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0.0 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > rf["alg_bytes_per_frame"] * 0.5
    dk = rf["dominant_kernel"]
    assert dk["isolated"] and dk["avg_launch_ms"] > 0 and abs(dk["frac"] - dk["achieved"] / 8000.0) < 1e-4
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mrays/s" and cb["cores"] >= 1 and cb["value"] > 0 and isinstance(cb["sample"], str)
    assert cb["single_thread"]["cores"] == 1 and 0 < cb["single_thread"]["value"] <= cb["value"] * 1.05


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_frames_in_flight_api_fuzz(ptlib, small_probe, seed):
    """A random sequence of API calls — uniform and foveated frames, camera / size / partition / option changes, read-backs, epilogues —
    on a synchronous renderer and on one with frames in flight (the mode itself changes along the way): every read-back must agree
    bit for bit.  Exercises the "wait for the frames in flight first" rule of every entry point."""
    from optixpathtracer_amd import renderer as R

    rng = np.random.default_rng(seed)
    m = scenes.voxel_terrain(n=64, target_tris=30000)
    size = [(128, 80)]

    def make(fif):
        r = R.SampleRenderer(m)
        r.setProbe(small_probe)
        r.setOptions(max_depth=4, frames_in_flight=fif)
        r.resize(size[0])
        r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, size[0][0] / size[0][1]))
        r.launchParams.samples_per_launch = 2
        return r

    a, b = make(0), make(3)
    ops = rng.choice(["render", "render", "render", "render", "foveated", "camera", "resize", "partition", "download", "stats", "options", "tonemap", "denoise", "sync"], 70)
    sub = 0
    checks = 0
    for op in ops:
        if op == "render":
            for r in (a, b):
                r.launchParams.frame.subframe_index = sub
                r.render()
            sub += 1
        elif op == "foveated":
            w, h = size[0]
            gaze = (int(rng.integers(32, w - 32)), int(rng.integers(32, h - 32)))  # regions stay inside the image (outer radius 30)
            for r in (a, b):
                r.launchParams.frame.subframe_index = sub
                r.renderFoveated(gaze, inner_radius=10, outer_radius=30, spp=(1, 2, 2), variant=r.SV4_VARIANT)
            sub += 1
        elif op == "camera":
            cam = dict(scenes.TERRAIN_CAMERA)
            cam["eye"] = tuple(float(c * rng.uniform(0.9, 1.1)) for c in cam["eye"])
            for r in (a, b):
                r.setCamera(R.make_camera(cam, size[0][0] / size[0][1]))
        elif op == "resize":
            size[0] = [(128, 80), (96, 72), (160, 88)][int(rng.integers(0, 3))]
            for r in (a, b):
                r.resize(size[0])
                r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, size[0][0] / size[0][1]))
            sub = 0
        elif op == "partition":
            rank, world = [(0, 1), (1, 2), (2, 3)][int(rng.integers(0, 3))]
            for r in (a, b):
                r.setPartition(rank, world, 16, 8)
            sub = 0
        elif op == "options":
            fif = int(rng.choice([2, 3]))
            spp = int(rng.choice([1, 2, 3]))
            b.setOptions(max_depth=4, frames_in_flight=fif)
            for r in (a, b):
                r.launchParams.samples_per_launch = spp
        elif op == "tonemap":
            assert np.array_equal(a.tonemapSqrt(), b.tonemapSqrt())
            checks += 1
        elif op == "denoise":
            da, _ = a.denoise(iterations=2)
            db, _ = b.denoise(iterations=2)
            assert np.array_equal(da.view(np.uint32), db.view(np.uint32))
            checks += 1
        elif op == "sync":
            b.sync()
        elif op == "stats":
            sa, sb = a.stats(), b.stats()
            assert (sa["frames"], sa["total_radiance_rays"], sa["total_shadow_rays"], sa["radiance_rays"], sa["shaded_hits"]) == (sb["frames"], sb["total_radiance_rays"], sb["total_shadow_rays"], sb["radiance_rays"], sb["shaded_hits"])
            checks += 1
        else:
            which = [R.PT_BUF_ACCUM, R.PT_BUF_FRAME, R.PT_BUF_NORMAL, R.PT_BUF_ALBEDO][int(rng.integers(0, 4))]
            assert np.array_equal(a.download(which).view(np.uint32), b.download(which).view(np.uint32)), (op, which)
            checks += 1
    assert np.array_equal(a.download(R.PT_BUF_ACCUM).view(np.uint32), b.download(R.PT_BUF_ACCUM).view(np.uint32))
    assert np.array_equal(a.download(R.PT_BUF_FRAME), b.download(R.PT_BUF_FRAME))
    assert checks >= 5


def test_fullsize_c3_frames_in_flight_match_synchronous(ptlib):
    """The schedule bench.py times (C3: 1 M triangles, 1920x1080, 4 spp, depth 8, three whole frames in flight, batch-sized traversal
    grid) against the synchronous schedule at the literal size: five progressive subframes, every buffer bit for bit, equal ray counts —
    also for a 1/8 share of the frame."""
    from optixpathtracer_amd import renderer as R

    m = scenes.voxel_terrain()
    probe = scenes.sky_probe(2048, 1024).BuildCDF()
    w, h = 1920, 1080
    for part in (None, (3, 8)):
        out = {}
        for fif in (0, 3):
            r = R.SampleRenderer(m)
            r.setProbe(probe)
            r.setOptions(frames_in_flight=fif)
            if part:
                r.setPartition(part[0], part[1], 64, 16)
            r.resize((w, h))
            r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
            r.launchParams.samples_per_launch = 4
            for sf in range(5):
                r.launchParams.frame.subframe_index = sf
                r.render()
            st = r.stats()
            out[fif] = [r.download(b) for b in (R.PT_BUF_ACCUM, R.PT_BUF_FRAME, R.PT_BUF_NORMAL, R.PT_BUF_ALBEDO, R.PT_BUF_COLOR)] + [st["total_radiance_rays"], st["total_shadow_rays"], st["frames"]]
            r.close()
        for k in range(5):
            assert np.array_equal(out[0][k].view(np.uint32), out[3][k].view(np.uint32)), (part, k)
        assert out[0][5:] == out[3][5:] and out[0][7] == 5

"""Properties of the restated parts that the reference cannot pin (it has no tests; Disney.cuh and
deviceProgram.cu do not compile without OptiX headers): the checks its own commented-out BSDFTest
(Disney.cuh:430-503) sketches, plus internal consistency of the checker itself."""
import ctypes as C

import numpy as np
import pytest

from optixpathtracer_amd import scenes


def _unit(rng, n):
    v = rng.standard_normal((n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def test_detmath_accuracy_vs_float64(orc_det):
    rng = np.random.default_rng(1)
    n = 200000

    def ulps(got, exact):
        exact32 = exact.astype(np.float32)
        return np.abs(got.astype(np.float64) - exact) / np.spacing(np.maximum(np.abs(exact32), np.float32(1e-30))).astype(np.float64)

    x = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
    # sin/cos near their zeros lose relative accuracy like every float implementation; bound the absolute error there
    assert np.abs(orc_det.math_table(0, x).astype(np.float64) - np.sin(x.astype(np.float64))).max() < 2.5e-7
    assert np.abs(orc_det.math_table(1, x).astype(np.float64) - np.cos(x.astype(np.float64))).max() < 2.5e-7
    x = rng.uniform(-1, 1, n).astype(np.float32)
    assert ulps(orc_det.math_table(2, x), np.arccos(x.astype(np.float64))).max() <= 4
    a = rng.standard_normal(n).astype(np.float32); b = rng.standard_normal(n).astype(np.float32)
    assert np.abs(orc_det.math_table(3, a, b).astype(np.float64) - np.arctan2(a.astype(np.float64), b.astype(np.float64))).max() < 1e-6
    x = np.exp(rng.uniform(-14, 1, n)).astype(np.float32)
    assert np.abs(orc_det.math_table(4, x).astype(np.float64) - np.log(x.astype(np.float64))).max() < 2e-6
    x = rng.uniform(0.003, 1, n).astype(np.float32)
    y = np.full(n, 1 / 2.4, np.float32)
    assert ulps(orc_det.math_table(5, x, y), np.power(x.astype(np.float64), y.astype(np.float64))).max() <= 16


def test_libm_and_det_oracles_agree_statistically(orc_det, orc_libm):
    """The two math modes are different roundings of the same algorithm: almost every pixel equal to 1e-4,
    the rest are path flips (chaotic), and the image-level relative L2 is small."""
    m = scenes.cornell_box()
    probe = scenes.sky_probe(256, 128).BuildCDF()
    w, h = 96, 64
    U, V, W = scenes.uvw_frame(**scenes.CORNELL_CAMERA, aspect=w / h)
    imgs = []
    for O in (orc_libm, orc_det):
        r = O.render(O.make_scene(m), O.make_probe(probe), (U, V, W), scenes.CORNELL_CAMERA["eye"], w, h, 8)
        imgs.append(r["accum"][..., :3].astype(np.float64))
    a, b = imgs
    rel = np.abs(a - b).max(-1) / (np.abs(a).max(-1) + 1e-3)
    assert (rel < 1e-4).mean() > 0.995
    assert np.linalg.norm(a - b) / np.linalg.norm(a) < 0.05


@pytest.mark.parametrize("mat_i", range(9))
def test_bsdf_sample_pdf_consistency(orc_det, mat_i):
    """BSDFTest's idea: every sampled direction has pdf > 0 there (or the sample is rejected), BSDFPdf of
    the sampled direction equals the pdf BSDFSample returned (non-specular lobes), values are finite."""
    mats = scenes.material_presets() + [scenes.Material()]
    mat = mats[mat_i]
    rng = np.random.default_rng(100 + mat_i)
    n = 3000
    N = _unit(rng, n); V = _unit(rng, n)
    V = np.where((np.sum(N * V, 1) < 0)[:, None], -V, V).astype(np.float32)
    L = orc_det.lib
    Lo = np.zeros(3, np.float32); pdf = C.c_float(); st = np.zeros(2, np.uint32); f = np.zeros(3, np.float32)
    albedo = np.ascontiguousarray(mat["color"], np.float32)
    n_pos = 0
    for i in range(n):
        L.orc_bsdf_sample(0, mat.ctypes.data, 1.0, 1.5, N[i].copy(), V[i].copy(), 12345 + i, Lo, C.byref(pdf), st)
        assert np.isfinite(pdf.value)
        if pdf.value <= 0:
            continue
        n_pos += 1
        assert abs(np.linalg.norm(Lo) - 1) < 1e-3
        p2 = L.orc_bsdf_pdf(0, mat.ctypes.data, 1.0, 1.5, N[i].copy(), V[i].copy(), Lo.copy())
        if float(mat["transmission"]) == 0.0:
            assert p2 == pdf.value  # BSDFSample ends in pdf = BSDFPdf(...) (Disney.cuh:312)
        L.orc_bsdf_eval(0, mat.ctypes.data, albedo, 1.0, 1.5, N[i].copy(), V[i].copy(), Lo.copy(), f)
        assert np.isfinite(f).all() and (f >= 0).all()
    assert n_pos > 0.5 * n


def test_bsdf_pdf_integrates_to_at_most_one(orc_det):
    """Monte-Carlo integral of BSDFPdf over the sphere (uniform directions): <= 1 for the default material
    (it is 0.5*cosine + 0.5*GGX-half-vector pdf above the surface, 0 below)."""
    mat = scenes.Material()
    rng = np.random.default_rng(7)
    n = 40000
    Lv = _unit(rng, n)
    N = np.array([0, 0, 1], np.float32); V = np.array([0.3, 0.1, 0.948], np.float32); V /= np.linalg.norm(V)
    s = 0.0
    for i in range(n):
        s += orc_det.lib.orc_bsdf_pdf(0, mat.ctypes.data, 1.0, 1.5, N, V, Lv[i].copy())
    integral = s / n * 4 * np.pi
    assert 0.5 < integral < 1.05  # < 1: half-vector reflections that land below the surface carry no pdf there


def test_lambert_mode_matches_simple_bsdf(orc_det):
    """USE_SIMPLE_BSDF (Disney.cuh:125-147): pdf 1/2pi above, 0 below; f = color/pi."""
    mat = scenes.Material(color=(0.2, 0.5, 0.9))
    N = np.array([0, 1, 0], np.float32); V = np.array([0, 1, 0], np.float32)
    f = np.zeros(3, np.float32)
    up = np.array([0.6, 0.8, 0], np.float32); dn = np.array([0.6, -0.8, 0], np.float32)
    assert abs(orc_det.lib.orc_bsdf_pdf(1, mat.ctypes.data, 1, 1.5, N, V, up) - 1 / (2 * np.pi)) < 1e-7
    assert orc_det.lib.orc_bsdf_pdf(1, mat.ctypes.data, 1, 1.5, N, V, dn) == 0
    orc_det.lib.orc_bsdf_eval(1, mat.ctypes.data, np.ascontiguousarray(mat["color"]), 1, 1.5, N, V, up, f)
    assert np.allclose(f, np.array([0.2, 0.5, 0.9]) / np.pi, rtol=1e-6)


def test_oracle_bvh_equals_bruteforce(orc_det):
    m = scenes.voxel_terrain(n=40, target_tris=12000)
    s1 = orc_det.make_scene(m, False); s2 = orc_det.make_scene(m, True)
    rng = np.random.default_rng(11)
    n = 4000
    o = rng.uniform(-120, 120, (n, 3)).astype(np.float32); o[:, 1] = rng.uniform(-10, 60, n)
    d = _unit(rng, n)
    rays = np.concatenate([o, np.full((n, 1), 1e-3, np.float32), d, np.full((n, 1), 1e16, np.float32)], 1).astype(np.float32)
    t1, p1 = orc_det.trace_closest(s1, rays); t2, p2 = orc_det.trace_closest(s2, rays)
    assert np.array_equal(p1, p2) and np.array_equal(t1.view(np.uint32), t2.view(np.uint32)) and (p1 >= 0).mean() > 0.2
    assert np.array_equal(orc_det.trace_any(s1, rays), orc_det.trace_any(s2, rays))


def test_watertight_shared_edges(orc_det):
    """Rays aimed at points on the shared diagonal of a quad's two triangles never slip through."""
    m = scenes.Model([scenes._quads_to_mesh([[(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1)]], scenes.Material())])
    sc = orc_det.make_scene(m, False)
    rng = np.random.default_rng(12)
    n = 20000
    s = rng.random(n).astype(np.float32)
    target = np.stack([s, np.zeros(n, np.float32), s], 1)  # on the diagonal (0,0,0)-(1,0,1)
    o = np.stack([rng.uniform(-2, 3, n), rng.uniform(0.5, 5, n), rng.uniform(-2, 3, n)], 1).astype(np.float32)
    d = target - o
    rays = np.concatenate([o, np.full((n, 1), 1e-3, np.float32), d, np.full((n, 1), 1e16, np.float32)], 1).astype(np.float32)
    t, p = orc_det.trace_closest(sc, rays)
    inner = (s > 1e-3) & (s < 1 - 1e-3)
    assert (p[inner] >= 0).all()


def test_render_is_thread_count_invariant_and_progressive(orc_det):
    m = scenes.cornell_box()
    probe = scenes.constant_probe().BuildCDF()
    w, h = 40, 24
    U, V, W = scenes.uvw_frame(**scenes.CORNELL_CAMERA, aspect=w / h)
    sc, pr = orc_det.make_scene(m), orc_det.make_probe(probe)
    a = orc_det.render(sc, pr, (U, V, W), scenes.CORNELL_CAMERA["eye"], w, h, 3, nthreads=1)
    b = orc_det.render(sc, pr, (U, V, W), scenes.CORNELL_CAMERA["eye"], w, h, 3, nthreads=5)
    assert np.array_equal(a["accum"], b["accum"]) and a["radiance_rays"] == b["radiance_rays"]
    # subframe 1 blends: accum1 = lerp(accum0, clamp(cur,0,10), 1/2)  (deviceProgram.cu:460-466)
    c = orc_det.render(sc, pr, (U, V, W), scenes.CORNELL_CAMERA["eye"], w, h, 3, subframe=1, accum=a["accum"])
    cur = orc_det.render(sc, pr, (U, V, W), scenes.CORNELL_CAMERA["eye"], w, h, 3, subframe=1, accum=np.zeros_like(a["accum"]))
    # with accum_prev = 0: result = 0 + .5*(clamp(cur)-0) → clamp(cur) = 2*that
    cl = 2 * cur["accum"][..., :3]
    exp = a["accum"][..., :3] + np.float32(0.5) * (cl - a["accum"][..., :3])
    assert np.allclose(c["accum"][..., :3], exp, rtol=1e-6, atol=1e-7)
    assert (c["accum"][..., 3] == 1).all()


def test_closed_white_furnace_energy_bound(orc_det):
    """Inside a closed box under NEE-only lighting nothing reaches the probe: every shadow ray is occluded,
    so radiance is exactly the emission seen on primary hits (quirks 1 and 3 of SURVEY.md §8a)."""
    m = scenes.Model()
    scenes.add_box(m, scenes.Material(color=(0.9, 0.9, 0.9), emission=(0.25, 0.5, 0.75)), (0, 0, 0), (5, 5, 5))
    probe = scenes.constant_probe().BuildCDF()
    cam = dict(eye=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0), fovY=60.0)
    w, h = 24, 16
    U, V, W = scenes.uvw_frame(**cam, aspect=w / h)
    r = orc_det.render(orc_det.make_scene(m), orc_det.make_probe(probe), (U, V, W), cam["eye"], w, h, 4)
    # ... except for samples whose BSDF sample fails at the first hit (pdf <= 0 → DONE): the raygen loop
    # breaks BEFORE adding that hit's radiance (quirk 2, deviceProgram.cu:429-437), so a pixel holds k/4 of it
    ratio = r["accum"][..., :3] / np.array([0.25, 0.5, 0.75], np.float32)
    k = ratio[..., 0] * 4
    # a few pixels see hits within tmin=.01 of a box edge, where the shadow ray legitimately leaks (quirk 8)
    exact = np.isclose(k, np.round(k), atol=1e-5) & np.isclose(ratio, ratio[..., :1], rtol=1e-6).all(-1) & (k <= 4)
    assert exact.mean() > 0.97 and (np.round(k[exact]) == 4).mean() > 0.25 and (np.round(k[exact]) >= 2).mean() > 0.9
    assert r["shadow_rays"] > 0


@pytest.mark.parametrize("kind", ["libm", "det"])
def test_atrous_filter_properties(kind, orc_libm, orc_det):
    """The checker's a-trous restatement (pt_denoise semantics): constants are fixed points, weights are a partition of
    unity (output within the input range), guide edges stop the blur, 0 iterations is the identity."""
    O = orc_libm if kind == "libm" else orc_det
    rng = np.random.default_rng(3)
    h, w = 40, 56
    const = np.full((h, w, 4), 0.37, np.float32)
    zeros = np.zeros((h, w, 4), np.float32)
    out = O.denoise(const, zeros, zeros, 5)
    assert np.abs(out - const).max() < 1e-6
    noisy = rng.uniform(0.2, 0.8, (h, w, 4)).astype(np.float32)
    assert np.array_equal(O.denoise(noisy, zeros, zeros, 0), noisy)
    out = O.denoise(noisy, zeros, zeros, 4, sigma_color=10.0)
    assert out[..., :3].min() >= 0.2 - 1e-6 and out[..., :3].max() <= 0.8 + 1e-6
    assert np.array_equal(out[..., 3], noisy[..., 3])  # alpha carried through
    assert out[..., :3].var() < 0.1 * noisy[..., :3].var()
    # two half-planes with different albedo and colour: the albedo guide keeps the step sharp
    step = np.zeros((h, w, 4), np.float32)
    step[:, : w // 2, :3] = 0.2
    step[:, w // 2 :, :3] = 0.9
    alb = step.copy()
    noise = rng.normal(0, 0.02, (h, w, 3)).astype(np.float32)
    col = step.copy()
    col[..., :3] += noise
    out = O.denoise(col, zeros, alb, 5, sigma_color=10.0, sigma_albedo=0.05)
    assert abs(out[:, : w // 2, :3].mean() - 0.2) < 0.01 and abs(out[:, w // 2 :, :3].mean() - 0.9) < 0.01
    assert out[:, w // 2 - 1, :3].max() < 0.3 and out[:, w // 2, :3].min() > 0.8


def test_probe_line_search_model_equals_lower_bound():
    """The device layout of ProbeSample's column search (pt_device.h: 128-byte lines of six columns + a u16 guide per row) restated in
    numpy: for every row kind the reference's BuildCDF can produce — smooth, flat stretches, all-NaN — and widths around the line size,
    guide -> candidate lines -> count gives the index of the reference's LowerBound (Probe.cuh:119-136).  Pins the ALGORITHM on the CPU; the
    kernel itself is compared bit for bit on the GPU (test_probe_tables)."""
    from optixpathtracer_amd import scenes

    G = 6

    def lower_bound_ref(row, v):  # Probe.cuh:119-136
        lo, hi = 0, len(row)
        while lo < hi:
            mid = lo + (hi - lo) // 2
            if row[mid] < v:
                lo = mid + 1
            else:
                hi = mid
        return lo

    def model(row, v, gk):
        w = len(row)
        lpr = (w + G - 1) // G
        cdf = np.full(lpr * G, np.inf, np.float32)
        cdf[:w] = row
        last = cdf.reshape(lpr, G)[:, G - 1]
        with np.errstate(invalid="ignore"):
            guide = np.array([int(np.sum(last < np.float32(k / gk))) for k in range(gk + 1)])  # counts: NaN compares false
            k = min(int(np.float32(v) * np.float32(gk)), gk - 1)
            lo, hi = min(guide[k], lpr - 1), min(guide[k + 1], lpr - 1)
            while hi - lo > 2:
                mid = lo + (hi - lo) // 2
                if last[mid] < v:
                    lo = mid + 1
                else:
                    hi = mid
            i1 = min(lo + 1, hi)
            l = lo
            if i1 > lo and last[lo] < v:
                l = i1
                if hi > i1 and last[i1] < v:
                    l = hi
            return l * G + int(np.sum(cdf[l * G:(l + 1) * G] < v))

    rng = np.random.default_rng(11)
    probes = [scenes.spots_probe(1000, 16), scenes.spots_probe(4099, 4, fill=0.002), scenes.spots_probe(7, 5, fill=0.5), scenes.spots_probe(13, 3, fill=0.3),
              scenes.constant_probe(5, 4), scenes.constant_probe(6, 8), scenes.sky_probe(256, 8), scenes.disc_probe(100, 10)]
    for pr in probes:
        p = pr.BuildCDF()
        w, h = p.width, p.height
        cdfx = np.asarray(p.cdfValuesX, np.float32).reshape(h, w)
        gk = 64
        while gk < 4096 and gk * 2 < w:
            gk *= 2
        for row in cdfx:
            vs = np.concatenate([rng.random(40, dtype=np.float32), row[np.isfinite(row)][:8], np.float32([0.0, np.nextafter(np.float32(1), np.float32(0))])])
            for v in vs:
                if not (0.0 <= v < 1.0):
                    continue
                with np.errstate(invalid="ignore"):
                    ref = lower_bound_ref(row, np.float32(v))
                got = model(row, np.float32(v), gk)
                assert min(got, w) == ref or (got >= w and ref == w), (w, v, got, ref)


def _census_rays(orc, sc, cam, n_cam, rng):
    """camera rays through random points of a 1080p image plane, then — from their hit points — as many bounce rays (uniform directions
    about the up axis, tmin 1e-3) and as many shadow-like rays (upper hemisphere, tmin 1e-2): the three kinds of rays a frame traces"""
    U, V, W = scenes.uvw_frame(**cam, aspect=1920 / 1080)
    x = rng.uniform(-1, 1, n_cam).astype(np.float32)
    y = rng.uniform(-1, 1, n_cam).astype(np.float32)
    d = x[:, None] * U[None] + y[:, None] * V[None] + W[None]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    cam_rays = np.zeros((n_cam, 8), np.float32)
    cam_rays[:, :3] = np.asarray(cam["eye"], np.float32)
    cam_rays[:, 3] = 1e-3
    cam_rays[:, 4:7] = d
    cam_rays[:, 7] = 1e16
    t, prim = orc.trace_closest(sc, cam_rays)
    hit = prim >= 0
    P = (cam_rays[hit, :3] + t[hit, None] * cam_rays[hit, 4:7]).astype(np.float32)
    out = [cam_rays]
    for tmin, up_only in ((1e-3, False), (1e-2, True)):
        dd = rng.standard_normal((len(P), 3)).astype(np.float32)
        dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        if up_only:
            dd[:, 1] = np.abs(dd[:, 1])
        r = np.zeros((len(P), 8), np.float32)
        r[:, :3] = P
        r[:, 3] = tmin
        r[:, 4:7] = dd
        r[:, 7] = 1e16
        out.append(r)
    return np.concatenate(out)


def _census(orc, model, cam, n_cam, seed=7, threads=8):
    from concurrent.futures import ThreadPoolExecutor

    sc = orc.make_scene(model, use_bvh=True)
    rays = _census_rays(orc, sc, cam, n_cam, np.random.default_rng(seed))
    parts = np.array_split(rays, threads)
    with ThreadPoolExecutor(threads) as ex:  # the C loop runs without the GIL
        res = list(ex.map(lambda p: orc.hit_census(sc, p), parts))
    tot = {k: sum(r[k] for r in res) for k in res[0]}
    return tot


def test_hit_in_box_census_against_exact_geometry(orc_det):
    """VERDICT round 4, "What's weak" 10: how often does the float acceptance rule (sign-consistent triple products + hit_in_box — this
    repository's definition of optixTrace's answer, the same bits in kernel and checker) return a different closest hit than exact geometry
    (Moller-Trumbore in double precision on the same vertices)?  Small scenes here; PT_CENSUS_FULL=1 runs the 1 M-triangle terrain and
    stadium with ~1.5 M rays each and prints the table DESIGN.md section 2 quotes (profiles/r5_07_hit_census.md)."""
    import os

    full = os.environ.get("PT_CENSUS_FULL") == "1"
    cases = [("terrain", scenes.voxel_terrain() if full else scenes.voxel_terrain(n=96, target_tris=70000), scenes.TERRAIN_CAMERA),
             ("stadium", scenes.stadium_scene() if full else scenes.stadium_scene(60000), scenes.STADIUM_CAMERA)]
    for name, model, cam in cases:
        c = _census(orc_det, model, cam, 600_000 if full else 40_000)
        per_m = {k: 1e6 * c[k] / c["rays"] for k in ("accepted_but_inexact", "rejected_but_exact", "order_only")}
        print(f"\n[census] {name}: {model.num_triangles} triangles, {c['rays']} rays, {c['candidates'] / c['rays']:.1f} candidate triangles per ray; closest hit equal for "
              f"{100.0 * c['same_closest_hit'] / c['rays']:.5f} %; per million rays: accepted-but-inexact {per_m['accepted_but_inexact']:.1f}, "
              f"rejected-but-exact {per_m['rejected_but_exact']:.1f}, same two triangles in another order {per_m['order_only']:.1f}; "
              f"candidates classified differently {c['candidates_classified_differently']} ({1e6 * c['candidates_classified_differently'] / c['candidates']:.2f} per million)")
        assert c["rays"] > 0 and c["same_closest_hit"] + c["accepted_but_inexact"] + c["rejected_but_exact"] + c["order_only"] == c["rays"]
        assert c["same_closest_hit"] / c["rays"] > 0.999  # the rule and exact geometry disagree on rays that graze an edge or a needle, nowhere else

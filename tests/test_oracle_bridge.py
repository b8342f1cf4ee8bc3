"""Bridges between the three statements of the algorithm (all CPU, no GPU needed):

  reference headers  --bit-exact (tests/test_oracle_golden.py)-->  checker "libm"
  checker "libm"     --per function, a few ulp (THIS FILE)------>  checker "det"  --bit-exact (-m gpu tests)--> HIP kernels

and, for the part of the reference that cannot be compiled here (Disney.cuh includes LaunchParams.h -> <optix.h>),
a SECOND, independent transcription of BSDFPdf / BSDFEval (Disney.cuh:151-192, 317-426) in vectorised float64 numpy,
written from the reference text alone and structured differently from oracle/pt_oracle.c, against which the checker is
compared: a misreading would have to be made twice, in two different shapes, to pass.

The GPU tests compare the HIP kernels with "det" bit for bit; "det" and "libm" are the same C source with two sets of
sinf/cosf/acosf/atan2f/logf/powf (include/pt_detmath.h vs glibc), so a function-level bound in ulp between them ties
the kernels to the reference-pinned build function by function (the image-level link is the converged-image test in
tests/test_gpu_parity.py)."""
import ctypes as C

import numpy as np
import pytest

from optixpathtracer_amd import scenes


def _unit(rng, n):
    v = rng.standard_normal((n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def _ulp_diff(a, b, floor=1e-6):
    """difference in units of the float32 spacing at max(|a|,|b|, floor)"""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    ref = np.maximum(np.maximum(np.abs(a), np.abs(b)), np.float32(floor))
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.spacing(ref).astype(np.float64)


def _mats():
    return scenes.material_presets() + [scenes.Material()]


def _geometry(rng, n):
    N = _unit(rng, n)
    V = _unit(rng, n)
    V = np.where((np.sum(N * V, 1) < 0)[:, None], -V, V).astype(np.float32)  # the viewer is on the normal's side
    L = _unit(rng, n)
    return N, V, L


def _eval_pdf(O, mat, N, V, L, etaI=1.0, etaO=1.5):
    lib = O.lib
    n = len(N)
    f = np.zeros((n, 3), np.float32)
    pdf = np.zeros(n, np.float32)
    albedo = np.ascontiguousarray(mat["color"], np.float32).reshape(3)
    tmp = np.zeros(3, np.float32)
    for i in range(n):
        lib.orc_bsdf_eval(0, mat.ctypes.data, albedo, etaI, etaO, N[i].copy(), V[i].copy(), L[i].copy(), tmp)
        f[i] = tmp
        pdf[i] = lib.orc_bsdf_pdf(0, mat.ctypes.data, etaI, etaO, N[i].copy(), V[i].copy(), L[i].copy())
    return f, pdf


@pytest.mark.parametrize("mat_i", range(9))
def test_det_vs_libm_bsdf_eval_pdf_ulp(orc_det, orc_libm, mat_i):
    """BSDFEval / BSDFPdf: the two builds differ only through logf (GTR1's clearcoat term) — every other operation is a
    single correctly rounded IEEE op on both sides — so they agree to a few ulp, and exactly where no clearcoat is set."""
    mat = _mats()[mat_i]
    rng = np.random.default_rng(300 + mat_i)
    N, V, L = _geometry(rng, 1500)
    for eta in ((1.0, 1.5), (1.5, 1.0)):
        fd, pd = _eval_pdf(orc_det, mat, N, V, L, *eta)
        fl, pl = _eval_pdf(orc_libm, mat, N, V, L, *eta)
        assert np.isfinite(fd).all() and np.isfinite(fl).all()
        assert np.array_equal(pd.view(np.uint32), pl.view(np.uint32)), "BSDFPdf uses no transcendental: bit-equal"
        u = _ulp_diff(fd, fl)
        if float(mat["clearcoat"]) == 0.0:
            assert u.max() == 0.0, f"BSDFEval without clearcoat must be bit-equal, max {u.max()} ulp"
        else:
            assert u.max() <= 4.0, f"BSDFEval with clearcoat: {u.max()} ulp between det and libm"


@pytest.mark.parametrize("mat_i", range(9))
def test_det_vs_libm_bsdf_sample(orc_det, orc_libm, mat_i):
    """BSDFSample with the same seeds: the same lobe decisions (they compare Randf with material constants and the
    Fresnel term, none of which involves a transcendental), directions within 1e-6 (a few ulp of sin/cos carried through the tangent basis),
    pdfs within 16 ulp, RNG state identical."""
    mat = _mats()[mat_i]
    rng = np.random.default_rng(400 + mat_i)
    N, V, _ = _geometry(rng, 1500)
    out = []
    for O in (orc_det, orc_libm):
        Lo = np.zeros(3, np.float32)
        pdf = C.c_float()
        st = np.zeros(2, np.uint32)
        res = np.zeros((len(N), 6), np.float64)
        for i in range(len(N)):
            O.lib.orc_bsdf_sample(0, mat.ctypes.data, 1.0, 1.5, N[i].copy(), V[i].copy(), 777 + i, Lo, C.byref(pdf), st)
            res[i] = [Lo[0], Lo[1], Lo[2], pdf.value, st[0], st[1]]
        out.append(res)
    d, l = out
    assert np.array_equal(d[:, 4:], l[:, 4:]), "RNG streams diverged: a lobe decision flipped between the builds"
    assert np.array_equal(d[:, 3] > 0, l[:, 3] > 0)
    ok = d[:, 3] > 0
    assert np.abs(d[ok, :3] - l[ok, :3]).max() <= 1e-6
    # the pdf of a mirror-like lobe is steep in the direction: a relative bound, with the ulp bound where it is smooth
    rel = np.abs(d[ok, 3] - l[ok, 3]) / np.maximum(np.abs(l[ok, 3]), 1e-6)
    assert rel.max() <= 2e-4, rel.max()
    assert np.median(_ulp_diff(d[ok, 3], l[ok, 3])) <= 2.0


def test_det_vs_libm_probe_sample_and_eval(orc_det, orc_libm):
    """ProbeSample with the same seeds picks the same texel (the search compares floats of identical CDF arrays; BuildCDF
    has no transcendental) — colour and RNG state bit-equal, direction within 3e-7 (sin/cos), pdf within 8 ulp (sin);
    ProbeDirToUV (acos/atan2) within 2e-7 and the texel ProbeEval returns identical away from texel borders."""
    probe = scenes.sky_probe(256, 128).BuildCDF()
    res = []
    for O in (orc_det, orc_libm):
        pr = O.make_probe(probe)
        d = np.zeros(3, np.float32)
        c = np.zeros(3, np.float32)
        pdf = C.c_float()
        st = np.zeros(2, np.uint32)
        rows = np.zeros((4000, 9), np.float64)
        for i in range(len(rows)):
            O.lib.orc_probe_sample(C.byref(pr), 9000 + i, d, c, C.byref(pdf), st)
            rows[i] = [*d, *c, pdf.value, st[0], st[1]]
        res.append(rows)
    a, b = res
    assert np.array_equal(a[:, 3:6], b[:, 3:6]) and np.array_equal(a[:, 7:], b[:, 7:])
    assert np.abs(a[:, :3] - b[:, :3]).max() <= 3e-7
    assert _ulp_diff(a[:, 6], b[:, 6]).max() <= 8.0
    rng = np.random.default_rng(12)
    dirs = _unit(rng, 4000)
    uv = []
    for O in (orc_det, orc_libm):
        o = np.zeros((len(dirs), 2), np.float32)
        t = np.zeros(2, np.float32)
        for i in range(len(dirs)):
            O.lib.orc_probe_dir_to_uv(dirs[i].copy(), t)
            o[i] = t
        uv.append(o)
    assert np.abs(uv[0].astype(np.float64) - uv[1]).max() <= 2e-7
    same_texel = (np.floor(uv[0] * [256, 128]) == np.floor(uv[1] * [256, 128])).all(1)
    assert same_texel.mean() > 0.999


def test_det_vs_libm_make_color(orc_det, orc_libm):
    """make_color (powf): at most one 8-bit step apart, and only at quantisation edges."""
    rng = np.random.default_rng(3)
    c = np.concatenate([rng.uniform(0, 1.2, (20000, 3)), rng.uniform(0, 0.01, (2000, 3))]).astype(np.float32)
    a = np.array([orc_det.lib.orc_make_color(x.copy()) for x in c], np.uint32)
    b = np.array([orc_libm.lib.orc_make_color(x.copy()) for x in c], np.uint32)
    da = np.stack([(a >> s) & 255 for s in (0, 8, 16, 24)], 1).astype(int)
    db = np.stack([(b >> s) & 255 for s in (0, 8, 16, 24)], 1).astype(int)
    assert np.abs(da - db).max() <= 1
    assert (da != db).any(1).mean() < 2e-3


# ---------------------------------------------------------------------------------------------------------------------
# Independent float64 transcription of Disney.cuh (vectorised; names follow the reference).

K_PI = float(np.float32(3.141592653589793))  # maths.h:29 (a float literal)


def _dot(a, b):
    return np.sum(a * b, axis=-1)


def _lerp(a, b, t):  # sutil/vec_math.h:98-101
    return a + t * (b - a)


def _fr(VDotN, etaI, etaT):  # Disney.cuh:75-93
    sin2 = (etaI / etaT) ** 2 * (1.0 - VDotN * VDotN)
    LDotN = np.sqrt(np.maximum(0.0, 1.0 - sin2))
    eta = etaT / etaI
    r1 = (VDotN - eta * LDotN) / (VDotN + eta * LDotN)
    r2 = (LDotN - eta * VDotN) / (LDotN + eta * VDotN)
    return np.where(sin2 > 1.0, 1.0, 0.5 * (r1 * r1 + r2 * r2))


def _schlick(u):  # :50-55
    m = np.clip(1.0 - u, 0.0, 1.0)
    return m ** 5


def _gtr1(NDotH, a):  # :57-63
    a2 = a * a
    t = 1.0 + (a2 - 1.0) * NDotH * NDotH
    return (a2 - 1.0) / (K_PI * np.log(a2) * t) if a < 1 else np.full_like(NDotH, 1.0 / K_PI)


def _gtr2(NDotH, a):  # :65-70
    a2 = a * a
    t = 1.0 + (a2 - 1.0) * NDotH * NDotH
    return a2 / (K_PI * t * t)


def _smith(NDotv, alphaG):  # :72-77
    a = alphaG * alphaG
    b = NDotv * NDotv
    return 1.0 / (NDotv + np.sqrt(a + b - a * b))


def disney_pdf_f64(m, etaI, etaO, n, V, L):  # Disney.cuh:151-192
    LdotN = _dot(L, n)
    below = _lerp(1.0 / (2.0 * K_PI) * m["subsurface"] * 0.5, 0.0, m["transmission"])
    F = _fr(_dot(n, V), etaI, etaO)
    a = max(0.001, m["roughness"])
    h = L + V
    hl = np.sqrt(_dot(h, h))[:, None]
    half = np.where(hl > 0, h / np.where(hl > 0, hl, 1.0), 0.0)  # SafeNormalize, maths.h:144-156
    cth = np.abs(_dot(half, n))
    pdfHalf = _gtr2(cth, a) * cth
    pdfSpec = 0.25 * pdfHalf / np.maximum(1e-6, _dot(L, half))
    pdfDiff = np.abs(LdotN) / K_PI * (1.0 - m["subsurface"])
    above = _lerp(_lerp(pdfDiff, pdfSpec, 0.5), pdfSpec * F, m["transmission"])
    return np.where(LdotN <= 0.0, below, above)


def disney_eval_f64(m, albedo, etaI, etaO, N, V, L):  # Disney.cuh:317-426
    NDotL, NDotV = _dot(N, L), _dot(N, V)
    H = L + V
    H = H / np.sqrt(_dot(H, H))[:, None]
    NDotH, LDotH = _dot(N, H), _dot(L, H)
    Cd = np.asarray(albedo, np.float64)
    lum = 0.3 * Cd[0] + 0.6 * Cd[1] + 0.1 * Cd[2]
    Ctint = Cd / lum if lum > 0 else np.ones(3)
    Cspec0 = _lerp(m["specular"] * 0.08 * _lerp(np.ones(3), Ctint, m["specularTint"]), Cd, m["metallic"])
    a = max(0.001, m["roughness"])
    up = (NDotL > 0)[:, None]
    bsdf = np.zeros((len(N), 3))
    if m["transmission"] > 0:
        F = _fr(NDotV, etaI, etaO)
        down_val = (m["transmission"] * (1.0 - F) / np.abs(NDotL) * (1.0 - m["metallic"]))[:, None] * np.ones(3)
        FH = _fr(LDotH, etaI, etaO)[:, None]
        up_val = (_smith(NDotV, a) * _smith(NDotL, a) * _gtr2(NDotH, a))[:, None] * _lerp(Cspec0[None, :], 1.0, FH)
        bsdf = np.where(up, up_val, down_val)
    brdf = np.zeros((len(N), 3))
    if m["transmission"] < 1:
        down_val = np.zeros((len(N), 3))
        if m["subsurface"] > 0:
            s = np.sqrt(np.asarray(m["color"], np.float64).reshape(3))  # the MATERIAL colour, not the (textured) albedo (:373)
            Fd = (1.0 - 0.5 * _schlick(np.abs(NDotL))) * (1.0 - 0.5 * _schlick(NDotV))
            down_val = (1.0 / K_PI) * s[None, :] * m["subsurface"] * Fd[:, None] * (1.0 - m["metallic"])
        FH = _schlick(LDotH)
        Fs = _lerp(Cspec0[None, :], 1.0, FH[:, None])
        Gs = _smith(NDotV, a) * _smith(NDotL, a)
        Fd90 = 0.5 + 2.0 * LDotH * LDotH * m["roughness"]
        Fd = _lerp(1.0, Fd90, _schlick(NDotL)) * _lerp(1.0, Fd90, _schlick(NDotV))
        Dr = _gtr1(NDotH, _lerp(0.1, 0.001, m["clearcoatGloss"]))
        Fc = _lerp(0.04, 1.0, FH)
        Gr = _smith(NDotL, 0.25) * _smith(NDotV, 0.25)
        up_val = ((1.0 / K_PI) * Fd[:, None] * Cd[None, :] * (1.0 - m["metallic"]) * (1.0 - m["subsurface"])
                  + (Gs * _gtr2(NDotH, a))[:, None] * Fs + (m["clearcoat"] * Gr * Fc * Dr)[:, None])
        brdf = np.where(up, up_val, down_val)
    return _lerp(brdf, bsdf, m["transmission"])


@pytest.mark.parametrize("mat_i", range(9))
def test_checker_disney_matches_independent_float64_transcription(orc_libm, mat_i):
    """The checker's BSDFEval / BSDFPdf (float32, the reference's operation order) against the float64 transcription
    above: relative difference <= 5e-4 of the value scale (float32 rounding of the checker; measured maximum 2.1e-4) for every preset (clearcoat, metal, subsurface, transmission,
    specular tint, default), both medium orders, directions on both sides of the surface."""
    mat = _mats()[mat_i]
    m = {k: (float(mat[k]) if np.ndim(mat[k]) == 0 else np.asarray(mat[k], np.float64).reshape(-1)) for k in mat.dtype.names}
    rng = np.random.default_rng(500 + mat_i)
    N, V, L = _geometry(rng, 2000)
    # keep away from the |N.L| -> 0 and L+V -> 0 poles, where float32 cancellation (not the formulas) sets the error
    keep = (np.abs(np.sum(N * L, 1)) > 0.02) & (np.linalg.norm(L + V, axis=1) > 0.05) & (np.sum(N * V, 1) > 0.02)
    N, V, L = N[keep], V[keep], L[keep]
    N64, V64, L64 = N.astype(np.float64), V.astype(np.float64), L.astype(np.float64)
    for etaI, etaO in ((1.0, 1.5), (1.5, 1.0)):
        f, pdf = _eval_pdf(orc_libm, mat, N, V, L, etaI, etaO)
        f64 = disney_eval_f64(m, m["color"], etaI, etaO, N64, V64, L64)
        p64 = disney_pdf_f64(m, etaI, etaO, N64, V64, L64)
        scale_f = np.maximum(np.abs(f64).max(1), 1e-3)[:, None]
        assert (np.abs(f - f64) / scale_f).max() <= 5e-4, (np.abs(f - f64) / scale_f).max()
        scale_p = np.maximum(np.abs(p64), 1e-3)
        assert (np.abs(pdf - p64) / scale_p).max() <= 5e-4, (np.abs(pdf - p64) / scale_p).max()

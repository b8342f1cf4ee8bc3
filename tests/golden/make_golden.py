#!/usr/bin/env python3
"""Generates tests/golden/ref_tables.npz from the REFERENCE's own code: oracle/_ref/libptref.so is
built by oracle/Makefile from the reference headers where they lie under /root/reference (cuda/random.h,
maths.h, sample.h, Probe.cuh, Material.h, cuda/helpers.h, sutil/Camera.cpp, sutil/vec_math.h).
Run in the build container only (the reference does not travel):  python tests/golden/make_golden.py
Every table stores its inputs next to the reference's outputs, so the fixture is self-contained."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from optixpathtracer_amd import scenes  # noqa: E402


def main():
    R = orc.load_ref()
    if R is None:
        raise SystemExit("oracle/_ref/libptref.so missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(20241022)
    G = {}
    # --- integer RNG streams: tea<4>, lcg/rnd, Random
    n = 512
    ab = rng.integers(0, 2**32, (n, 2), dtype=np.uint64).astype(np.uint32)
    ab[:4] = [[0, 0], [1, 0], [0xFFFFFFFF, 0xFFFFFFFF], [1920 * 1080 - 1, 63]]
    tea = np.array([R.ref_tea4(int(a), int(b)) for a, b in ab], np.uint32)
    lcg_state = np.zeros((n, 4), np.uint32)
    rnd_val = np.zeros((n, 4), np.float32)
    rand_u = np.zeros((n, 6), np.uint32)
    randf = np.zeros((n, 6), np.float32)
    rand_state = np.zeros((n, 2), np.uint32)
    for i in range(n):
        s = C.c_uint32(int(tea[i]))
        for k in range(4):
            rnd_val[i, k] = R.ref_rnd(C.byref(s))
            lcg_state[i, k] = s.value
        st = np.zeros(2, np.uint32)
        R.ref_random_init(st, int(tea[i]))
        for k in range(6):
            rand_u[i, k] = R.ref_rand(st)
        R.ref_random_init(st, int(tea[i]))
        for k in range(6):
            randf[i, k] = R.ref_randf(st)
        rand_state[i] = st
    G.update(rng_ab=ab, rng_tea=tea, rng_lcg_state=lcg_state, rng_rnd=rnd_val, rng_rand=rand_u, rng_randf=randf, rng_state=rand_state)
    # --- vector helpers / samplers
    n = 2000
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:6] = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]]
    d = d.astype(np.float32)
    bu = np.zeros((n, 3), np.float32); bv = np.zeros((n, 3), np.float32)
    uv = np.zeros((n, 2), np.float32); back = np.zeros((n, 3), np.float32)
    nrm = np.zeros((n, 3), np.float32); ff = np.zeros((n, 3), np.float32)
    other = rng.standard_normal((n, 3)).astype(np.float32)
    for i in range(n):
        R.ref_basis_from_vector(d[i].copy(), bu[i], bv[i])
        R.ref_probe_dir_to_uv(d[i].copy(), uv[i])
        R.ref_probe_uv_to_dir(uv[i].copy(), back[i])
        R.ref_normalize(other[i].copy(), nrm[i])
        R.ref_faceforward(d[i].copy(), other[i].copy(), ff[i])
    G.update(vec_dir=d, vec_basis_u=bu, vec_basis_v=bv, vec_uv=uv, vec_uv_dir=back, vec_other=other, vec_normalize=nrm, vec_faceforward=ff)
    seeds = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    ush = np.zeros((n, 3), np.float32); csh = np.zeros((n, 3), np.float32)
    u12 = rng.random((n, 2)).astype(np.float32)
    for i in range(n):
        R.ref_uniform_sample_hemisphere(int(seeds[i]), ush[i])
        R.ref_cosine_sample_hemisphere(float(u12[i, 0]), float(u12[i, 1]), csh[i])
    G.update(smp_seeds=seeds, smp_uniform_hemi=ush, smp_u12=u12, smp_cosine_hemi=csh)
    # --- probe: the reference's ProbeCreateTest-style disc probe (Probe.cuh:207-242) + CDF arrays as inputs
    pd = scenes.disc_probe(100, 50)
    O = orc.Oracle("libm")
    pd.BuildCDF(O.build_cdf)
    ps = np.zeros((n, 7), np.float32); pstate = np.zeros((n, 2), np.uint32); pe = np.zeros((n, 4), np.float32)
    lum = np.zeros(pd.width * pd.height, np.float32)
    flat = pd.data.reshape(-1, 4)
    for i in range(len(lum)):
        lum[i] = R.ref_luminance(flat[i].copy())
    for i in range(n):
        pdf = C.c_float()
        R.ref_probe_sample(pd.width, pd.height, pd.data.reshape(-1), pd.pdfValuesX.reshape(-1), pd.cdfValuesX.reshape(-1), pd.pdfValuesY, pd.cdfValuesY,
                           int(seeds[i]), ps[i, :3], ps[i, 3:6], C.byref(pdf), pstate[i])
        ps[i, 6] = pdf.value
        R.ref_probe_eval(pd.width, pd.height, pd.data.reshape(-1), uv[i].copy(), pe[i])
    G.update(probe_w=np.int32(pd.width), probe_h=np.int32(pd.height), probe_data=pd.data, probe_pdfX=pd.pdfValuesX, probe_cdfX=pd.cdfValuesX,
             probe_pdfY=pd.pdfValuesY, probe_cdfY=pd.cdfValuesY, probe_sample=ps, probe_sample_state=pstate, probe_eval=pe, probe_luminance=lum)
    # --- make_color
    c = (rng.random((4000, 3)) * 1.4 - 0.2).astype(np.float32)
    c[:400, 0] = np.linspace(0, 0.006, 400)
    mc = np.array([R.ref_make_color(c[i].copy()) for i in range(len(c))], np.uint32)
    G.update(color_in=c, color_out=mc)
    # --- Material defaults / IOR
    buf = np.zeros(104, np.uint8)
    R.ref_material_default(buf.ctypes.data)
    spec = np.linspace(0, 1, 41).astype(np.float32)
    ior = np.zeros(41, np.float32)
    for i, s in enumerate(spec):
        m = scenes.Material(specular=s)
        ior[i] = R.ref_material_ior(m.ctypes.data)
    m = scenes.Material(eta=1.33)
    G.update(mat_default_bytes=buf, mat_sizeof=np.int32(R.ref_sizeof_material()), mat_specular=spec, mat_ior=ior, mat_ior_eta=np.float32(R.ref_material_ior(m.ctypes.data)))
    # --- UVWFrame
    cams = []
    for cam, asp in ((scenes.CORNELL_CAMERA, 1920 / 1080), (scenes.TERRAIN_CAMERA, 1920 / 1080), (scenes.TWO_BOX_CAMERA, 1.5), (scenes.CORNELL_CAMERA, 1.0)):
        U = np.zeros(3, np.float32); V = np.zeros(3, np.float32); W = np.zeros(3, np.float32)
        R.ref_uvw_frame(np.array(cam["eye"], np.float32), np.array(cam["lookat"], np.float32), np.array(cam["up"], np.float32), cam["fovY"], asp, U, V, W)
        cams.append(np.concatenate([cam["eye"], cam["lookat"], cam["up"], [cam["fovY"], asp], U, V, W]).astype(np.float32))
    G.update(cam_table=np.array(cams, np.float32))
    # --- round 3: SafeNormalize (maths.h:144-156: a * (1.0 / sqrt(m)) with a DOUBLE quotient), lerp / clamp (sutil/vec_math.h:500-516),
    # toSRGB alone (cuda/helpers.h:34-42).  SafeNormalize inputs span 60 binades of |a|^2 plus the zero vector and denormal lengths, so that a
    # float division that differed from the double quotient rounded to float would show.
    n = 6000
    sa = (rng.standard_normal((n, 3)) * np.exp(rng.uniform(-20, 20, (n, 1)))).astype(np.float32)
    sa[:4] = [[0, 0, 0], [1e-23, 0, 0], [3, 4, 0], [1e-30, 1e-30, 1e-30]]
    sn = np.zeros((n, 3), np.float32)
    la = rng.standard_normal((n, 3)).astype(np.float32); lb = rng.standard_normal((n, 3)).astype(np.float32)
    lt = rng.uniform(-0.25, 1.25, n).astype(np.float32)
    lo = np.zeros((n, 3), np.float32)
    cv = (rng.standard_normal((n, 3)) * 6).astype(np.float32)
    cv[:3] = [[np.nan, 0.5, 11], [-0.0, 10.0, 0.0], [np.inf, -np.inf, 5]]
    cl = np.zeros((n, 3), np.float32)
    sc = (rng.random((n, 3)) * 1.2).astype(np.float32)
    sc[:600, 0] = np.linspace(0.0, 0.0062616, 600)  # both sides of the 0.0031308 knee
    so = np.zeros((n, 3), np.float32)
    for i in range(n):
        R.ref_safe_normalize(sa[i].copy(), sn[i])
        R.ref_lerp3(la[i].copy(), lb[i].copy(), float(lt[i]), lo[i])
        R.ref_clamp3(cv[i].copy(), 0.0, 10.0, cl[i])
        R.ref_to_srgb(sc[i].copy(), so[i])
    G.update(safe_in=sa, safe_out=sn, lerp_a=la, lerp_b=lb, lerp_t=lt, lerp_out=lo, clamp_in=cv, clamp_out=cl, srgb_in=sc, srgb_out=so)
    # --- round 4: ProbePdf (Probe.cuh:69-93) on the disc probe above: the 2000 unit directions of vec_dir plus directions at and next to the poles
    # (|sin(theta)| < 1e-4 gives pdf 0) — appended with its own generator so that every earlier table stays byte-identical
    rng4 = np.random.default_rng(404)
    pdirs = np.concatenate([d, np.array([[0, 1, 0], [0, -1, 0], [1e-5, 1, 0], [0, -1, 1e-5], [3e-4, 1, 0], [1, 0, 0], [0, 0, -1]], np.float32),
                            (rng4.standard_normal((500, 3)) * np.array([1e-3, 1, 1e-3])).astype(np.float32)])
    ppdf = np.array([R.ref_probe_pdf(pd.width, pd.height, pd.data.reshape(-1), pd.pdfValuesX.reshape(-1), pd.pdfValuesY, pdirs[i].copy()) for i in range(len(pdirs))], np.float32)
    G.update(probe_pdf_dirs=pdirs, probe_pdf=ppdf)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_tables.npz")
    np.savez_compressed(out, **G)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generates tests/golden/restatement_frames.npz: small whole frames rendered by the CPU restatement built with the
deterministic math header (oracle 'det': only IEEE + - * / sqrt, so the bits do not depend on the machine or libm).
These are NOT reference outputs (the reference's device program cannot be built here): they freeze the
restatement's semantics so that a later edit of oracle/ or include/pt_detmath.h that changes a single bit of a frame is
noticed (tests/test_oracle_golden.py::test_restatement_frames_frozen).  Run:  python tests/golden/make_restatement_frames.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from optixpathtracer_amd import scenes  # noqa: E402

CASES = {
    # name: (scene factory, camera, probe factory, w, h, spp, subframes, max_depth, bsdf_mode)
    "cornell_c1_lambert": (scenes.cornell_box, scenes.CORNELL_CAMERA, lambda: scenes.sky_probe(256, 128), 48, 48, 1, 1, 4, 1),
    "cornell_disney_progressive": (scenes.cornell_box, scenes.CORNELL_CAMERA, lambda: scenes.sky_probe(256, 128), 48, 32, 2, 3, 8, 0),
    "two_box_shadow_catcher": (lambda: scenes.two_box_scene(shadow_catcher=True), scenes.TWO_BOX_CAMERA, lambda: scenes.disc_probe(), 48, 32, 2, 1, 8, 0),
    "terrain_all_materials": (lambda: scenes.voxel_terrain(n=48, target_tris=15000), scenes.TERRAIN_CAMERA, lambda: scenes.sky_probe(256, 128), 40, 24, 2, 1, 8, 0),
}


def render_case(O, case):
    make_scene, cam, make_probe, w, h, spp, nsub, depth, mode = case
    model, probe = make_scene(), make_probe().BuildCDF()
    sc = O.make_scene(model, None)
    pr = O.make_probe(probe)
    U, V, W = scenes.uvw_frame(**cam, aspect=w / h)
    out, accum = None, None
    for sf in range(nsub):
        out = O.render(sc, pr, (U, V, W), cam["eye"], w, h, spp, depth, sf, mode, accum, 4)
        accum = out["accum"]
    return out


def main():
    O = orc.Oracle("det")
    G = {}
    for name, case in CASES.items():
        out = render_case(O, case)
        for k in ("accum", "color", "normal", "albedo"):
            G[f"{name}.{k}"] = np.ascontiguousarray(out[k], np.float32).view(np.uint32)
        G[f"{name}.frame"] = out["frame"]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "restatement_frames.npz")
    np.savez_compressed(path, **G)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""pt_set_probe_image (GPU BuildCDF) against the host loop pt_build_cdf + upload, 2k x 1k and 8k x 4k probes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from optixpathtracer_amd import scenes
from optixpathtracer_amd import renderer as R

m = scenes.cornell_box()
r = R.SampleRenderer(m)
for (w, h) in ((2048, 1024), (8192, 4096)):
    p = scenes.sky_probe(w, h)
    for k in range(3):
        t0 = time.perf_counter(); r.setProbeImage(np.asarray(p.data).reshape(h, w, 4)); t1 = time.perf_counter()
    t2 = time.perf_counter(); q = p.BuildCDF(); r.setProbe(q); t3 = time.perf_counter()
    print(f"{w}x{h}: GPU pt_set_probe_image {1e3*(t1-t0):.1f} ms (incl. {w*h*16/1e6:.0f} MB upload), host BuildCDF + pt_set_probe {1e3*(t3-t2):.1f} ms", flush=True)

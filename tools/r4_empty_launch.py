"""Fixed cost of a launch chain: the 1 M-triangle terrain seen by a camera that looks at the sky — every camera ray misses, so the eight bounces'
k_shade / k_trace8 launches run on empty queues.  Run under rocprofv3 --kernel-trace --stats: the per-kernel averages are the launches' floors."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from optixpathtracer_amd import scenes, renderer as R
m = scenes.voxel_terrain()
probe = scenes.sky_probe(256, 128).BuildCDF()
r = R.SampleRenderer(m); r.setProbe(probe); r.resize((960, 270))
cam = dict(eye=(0.0, 300.0, 0.0), lookat=(0.0, 600.0, 1.0), up=(0.0, 0.0, 1.0), fovY=45.0)
r.setCamera(R.make_camera(cam, 960 / 270))
r.launchParams.samples_per_launch = 4
for k in range(3): r.render()
t0 = time.time()
for k in range(50):
    r.launchParams.frame.subframe_index = k
    r.render()
dt = (time.time() - t0) / 50
st = r.stats()
print("all-miss frame of %d paths: %.3f ms per frame, radiance rays %d, shadow rays %d, shaded hits %d" % (st["paths"], dt * 1e3, st["radiance_rays"], st["shadow_rays"], st["shaded_hits"]))

"""Soak (argument: frames in flight, default 3): 40 create/render/destroy cycles (device memory delta is the runtime's first-use pools: tools/leak_check.py shows no drift afterwards) and 1500 progressive 1080p frames of the 1 M-triangle scene (stable ms/frame, finite accumulation)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
FIF = int(sys.argv[1]) if len(sys.argv) > 1 else 3
from optixpathtracer_amd import scenes, renderer as R
m = scenes.voxel_terrain(n=96, target_tris=70000)
probe = scenes.sky_probe(256, 128).BuildCDF()
free0 = torch.cuda.mem_get_info()[0]
for k in range(40):
    r = R.SampleRenderer(m); r.setProbe(probe); r.resize((320, 200)); r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, 1.6))
    r.setOptions(frames_in_flight=FIF); r.launchParams.samples_per_launch = 2; r.render(); r.render(); r.close()  # destroyed with frames in flight
    if k % 10 == 9:
        mr = R.MultiRenderer(m, devices=[0, 0]); mr.setProbe(probe); mr.resize((320, 200)); mr.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, 1.6)); mr.render(); mr.close()
free1 = torch.cuda.mem_get_info()[0]
print("create/destroy x40: device memory delta %.1f MB" % ((free0 - free1) / 1e6), flush=True)
big = scenes.voxel_terrain()
r = R.SampleRenderer(big); r.setProbe(scenes.sky_probe(2048, 1024).BuildCDF()); r.resize((1920, 1080)); r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, 16 / 9))
r.launchParams.samples_per_launch = 4
r.setOptions(frames_in_flight=FIF)
t0 = time.time()
for k in range(1500):
    r.launchParams.frame.subframe_index = k
    r.render()
r.sync()
dt = time.time() - t0
a = r.download(R.PT_BUF_ACCUM)
print("1500 progressive frames, %d in flight: %.2f ms/frame, accum finite %s, mean %.4f" % (FIF, dt / 1500 * 1e3, bool(np.isfinite(a).all()), float(a[..., :3].mean())), flush=True)

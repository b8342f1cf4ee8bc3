"""Does the order in which schedules are used change their speed (path-state re-allocation)?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optixpathtracer_amd import renderer as R, scenes

model = scenes.voxel_terrain()
probe = scenes.sky_probe(2048, 1024).BuildCDF()
w, h = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1

def make():
    r = R.SampleRenderer(model)
    r.setProbe(probe)
    if world > 1:
        r.setPartition(0, world, 64, 16)
    r.resize((w, h))
    r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, w / h))
    r.launchParams.samples_per_launch = 4
    return r

def run(r, fif, n=24, tag=""):
    r.setOptions(frames_in_flight=fif)
    for k in range(4):
        r.launchParams.frame.subframe_index = k
        r.render()
    r.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        r.launchParams.frame.subframe_index = 4 + k
        r.render()
    r.sync(); torch.cuda.synchronize()
    print(f"{tag} fif={fif}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/frame", flush=True)

r = make(); run(r, 3, tag="fresh"); run(r, 0, tag="then"); run(r, 3, tag="then"); run(r, 0, tag="then"); r.close()
r = make(); run(r, 0, tag="fresh"); run(r, 3, tag="then"); run(r, 3, tag="again"); r.close()

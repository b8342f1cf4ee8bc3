"""Device-memory leak check: create / render / destroy contexts (single and multi) repeatedly; free memory must not drift."""
import sys
sys.path.insert(0, '.')
import torch
from optixpathtracer_amd import scenes, renderer as R
m = scenes.voxel_terrain(n=96, target_tris=70000)
probe = scenes.sky_probe(256, 128).BuildCDF()
def cycle(n):
    for k in range(n):
        r = R.SampleRenderer(m); r.setProbe(probe); r.resize((320, 200)); r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, 1.6))
        r.launchParams.samples_per_launch = 2; r.render(); r.close()
def mcycle(n):
    for k in range(n):
        mr = R.MultiRenderer(m, devices=[0, 0, 0]); mr.setProbe(probe); mr.resize((320, 200)); mr.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, 1.6))
        mr.launchParams.samples_per_launch = 2; mr.render()
        # round 3: the rank threads, the overlapped hand-off with its display buffers and strips, a batch — destroyed with a hand-over still pending
        mr.setOptions(frames_in_flight=3); mr.render(); mr.renderBatch(3); mr.render(); mr.close()
def tcycle(n):
    # round 4: textured scenes (tiled texels, 64-byte per-triangle records), the builder's arena, the chunk-enqueue threads of small frames
    tm = scenes.textured_terrain(n=96, target_tris=70000, tex_size=128)
    for k in range(n):
        r = R.SampleRenderer(tm); r.setProbe(probe); r.resize((320, 200)); r.setCamera(R.make_camera(scenes.TERRAIN_CAMERA, 1.6))
        r.launchParams.samples_per_launch = 2; r.render(); r.render(); r.close()
cycle(3)
mcycle(2)
tcycle(2)
f0 = torch.cuda.mem_get_info()[0]
for n in (20, 20, 40):
    cycle(n)
    f1 = torch.cuda.mem_get_info()[0]
    print("after %d more cycles: delta %.1f MB" % (n, (f0 - f1) / 1e6), flush=True)
for n in (10, 20):
    mcycle(n)
    f1 = torch.cuda.mem_get_info()[0]
    print("after %d more 3-context cycles: delta %.1f MB" % (n, (f0 - f1) / 1e6), flush=True)
for n in (20, 20):
    tcycle(n)
    f1 = torch.cuda.mem_get_info()[0]
    print("after %d more textured cycles: delta %.1f MB" % (n, (f0 - f1) / 1e6), flush=True)
import threading
print("host threads alive at the end:", threading.active_count(), "(python's own); native enqueue threads are joined by pt_destroy")

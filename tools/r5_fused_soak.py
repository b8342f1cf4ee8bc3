"""tools/r5_fused_soak.py [seconds] — soak of the fused bounce loop (csrc/pt_fused.h) against the launch chain: random frame sizes, partitions,
samples per launch, depth limits, BSDF modes, window sizes, cameras and progressive subframes on four small scenes; every frame is rendered by
a PT_FUSED=2 context and a PT_FUSED=0 context and the accumulation buffer, the rgba8 frame and the three device-counted totals must be equal.
Any difference, a fault bit or a hang (the caller wraps this in `timeout`) fails.  GPU box only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optixpathtracer_amd import renderer as R  # noqa: E402
from optixpathtracer_amd import scenes  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20261005)
SCENES = [("terrain70k", scenes.voxel_terrain(n=96, target_tris=70000), scenes.TERRAIN_CAMERA), ("stadium20k", scenes.stadium_scene(20000), scenes.STADIUM_CAMERA),
          ("cornell", scenes.cornell_box(), scenes.CORNELL_CAMERA), ("textured", scenes.textured_scene(), scenes.CORNELL_CAMERA)]
probe = scenes.sky_probe(256, 128).BuildCDF()


def make(model, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        r = R.SampleRenderer(model)  # the switches are read at pt_create
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    r.setProbe(probe)
    return r


t_end = time.time() + budget
frames = configs = 0
while time.time() < t_end:
    name, model, cam = SCENES[int(rng.integers(len(SCENES)))]
    cap = int(rng.choice([64, 128, 192, 448]))
    grid = int(rng.choice([0, 0, 1, 7, 300]))  # few waves: every wave goes through many refills
    a = make(model, {"PT_FUSED": "2", "PT_FUSED_CAP": str(cap), "PT_FUSED_GRID": str(grid)})
    b = make(model, {"PT_FUSED": "0"})
    for _ in range(4):
        w, h = int(rng.integers(8, 400)), int(rng.integers(8, 300))
        world = int(rng.choice([1, 1, 2, 3, 8]))
        rank = int(rng.integers(world))
        spp = int(rng.integers(1, 5))
        depth = int(rng.integers(1, 9))
        mode = int(rng.choice([R.PT_BSDF_DISNEY, R.PT_BSDF_DISNEY, R.PT_BSDF_LAMBERT]))
        streams = int(rng.choice([0, 1, 3]))
        max_paths = int(rng.choice([0, 0, 5000, 70000]))
        eye = np.asarray(cam["eye"], np.float64) * (1.0 + 0.3 * rng.standard_normal(3))
        c = dict(cam, eye=tuple(eye))
        for r in (a, b):
            r.setOptions(max_depth=depth, bsdf_mode=mode, streams=streams, max_paths=max_paths)
            r.setPartition(rank, world, 64, 16)
            r.resize((w, h))
            r.setCamera(R.make_camera(c, w / h))
            r.launchParams.samples_per_launch = spp
        for sf in range(int(rng.integers(1, 4))):
            out = []
            for r in (a, b):
                r.launchParams.frame.subframe_index = sf
                r.render()
                st = r.stats()
                out.append((r.download(R.PT_BUF_ACCUM).copy(), r.downloadPixels().copy(), tuple(st[k] for k in ("radiance_rays", "shadow_rays", "shaded_hits", "paths")), st["fused_passes"]))
            (fa, pa, sa, na), (fb, pb, sb, nb) = out
            ctx = f"{name} {w}x{h} rank {rank}/{world} spp {spp} depth {depth} mode {mode} streams {streams} max_paths {max_paths} cap {cap} grid {grid} subframe {sf}"
            assert nb == 0 and (na > 0 or a.ownedPixels()[0] == 0), (ctx, na, nb)
            assert sa == sb, (ctx, sa, sb)
            assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)) and np.array_equal(pa, pb), ctx
            frames += 1
        configs += 1
    a.close()
    b.close()
print(f"fused soak: {frames} frames of {configs} configurations on {len(SCENES)} scenes, every buffer and every ray count equal to the launch chain's")

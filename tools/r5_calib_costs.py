"""tools/r5_calib_costs.py — the calibration cost (node steps + 0.6 x triangle tests per calibration ray, PT_DEBUG_BVH) of the scenes the
tests and the bench use: what PT_FUSED_MAX_COST (22) is compared with."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PT_DEBUG_BVH"] = "1"
from optixpathtracer_amd import scenes
from optixpathtracer_amd.renderer import SampleRenderer

for name, m in (("terrain 70k", scenes.voxel_terrain(n=96, target_tris=70000)), ("terrain 1M", scenes.voxel_terrain()), ("stadium 60k", scenes.stadium_scene(60000)),
                ("stadium 1M", scenes.stadium_scene()), ("terrain 10M", scenes.voxel_terrain(n=1500, target_tris=10_000_000)), ("cornell", scenes.cornell_box())):
    print("==", name, m.num_triangles, flush=True)
    sys.stderr.flush()
    r = SampleRenderer(m)
    r.close()

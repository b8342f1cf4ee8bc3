"""two pt_create calls on the C3 terrain (the second build is the warm one) — what tools/r5_sah_prof.sh, tools/r6_s1.sh and tools/r6_s2.sh profile
(round 5 also ran it under an experiment script, r5_sah_exp.sh, that was never committed and is lost: profiles/r6_01_sah_fault.md)"""
import os, sys
sys.path.insert(0, os.getcwd())
from optixpathtracer_amd import renderer as R, scenes
m = scenes.voxel_terrain()
for k in range(2):
    r = R.SampleRenderer(m); print(r.stats()["bvh_build_ms"], r.stats()["bvh_builder"]); r.close()

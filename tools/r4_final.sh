#!/bin/bash
# tools/r4_final.sh: the round's evidence on one box — full GPU tests, rocprofv3 passes for C3 and the stadium (kernel stats, HBM traffic, wave state),
# the textured workload beside C3, and every bench line.  Judged files land under gpurun_out/*/judged (gpurun merges gpurun_out/ back).
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_t_final.log 2>&1
tail -3 gpurun_out/r4_t_final.log
bash tools/profile_r4.sh r4_10 c3_terrain1M_1080p_4spp_d8 r4_pmc.json 2>&1 | tail -3
bash tools/profile_r4.sh r4_11_stadium stadium1M_1080p_4spp_d8 stadium_r4_pmc.json 2>&1 | tail -3
R4_NOCPU=1 bash tools/r4_textured.sh r4_12_tex > gpurun_out/r4_12_tex.log 2>&1; tail -3 gpurun_out/r4_12_tex.log
bash tools/r4_lines.sh r4_13_lines 2>&1 | tail -14

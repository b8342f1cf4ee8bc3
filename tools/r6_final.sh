#!/bin/bash
# tools/r6_final.sh: the round's evidence on one box — full GPU tests, rocprofv3 passes for C3 and the stadium (kernel stats, HBM traffic, wave state), every bench line.
# Judged files land under gpurun_out/*/judged (gpurun merges gpurun_out/ back; tools/collect_r6.py copies them into profiles/).
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6_t_final.log 2>&1
tail -3 gpurun_out/r6_t_final.log
bash tools/profile_r6.sh r6_10 c3_terrain1M_1080p_4spp_d8 r6_pmc.json 2>&1 | tail -3
bash tools/profile_r6.sh r6_11_stadium stadium1M_1080p_4spp_d8 stadium_r6_pmc.json 2>&1 | tail -3
bash tools/r6_lines.sh r6_13_lines 2>&1 | tail -34

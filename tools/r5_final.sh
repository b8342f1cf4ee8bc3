#!/bin/bash
# tools/r5_final.sh: the round's evidence on one box — full GPU tests, rocprofv3 passes for C3 and the stadium (kernel stats, HBM traffic, wave state), every bench line.
# Judged files land under gpurun_out/*/judged (gpurun merges gpurun_out/ back; tools/collect_r5.py copies them into profiles/).
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5_t_final.log 2>&1
tail -3 gpurun_out/r5_t_final.log
bash tools/profile_r5.sh r5_10 c3_terrain1M_1080p_4spp_d8 r5_pmc.json 2>&1 | tail -3
bash tools/profile_r5.sh r5_11_stadium stadium1M_1080p_4spp_d8 stadium_r5_pmc.json 2>&1 | tail -3
bash tools/r5_lines.sh r5_13_lines 2>&1 | tail -24

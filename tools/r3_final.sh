cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_t13.log 2>&1
tail -4 gpurun_out/r3_t13.log
bash tools/profile_r3.sh r3_10 c3_terrain1M_1080p_4spp_d8 r3_pmc.json 2>&1 | tail -3
bash tools/profile_r3.sh r3_11_stadium stadium1M_1080p_4spp_d8 stadium_r3_pmc.json 2>&1 | tail -3
bash tools/r3_lines.sh 2>&1 | tail -16

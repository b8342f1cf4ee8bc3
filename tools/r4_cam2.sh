#!/bin/bash
V=$PWD/optixpathtracer_amd/variants
python -m pytest tests/test_gpu_parity.py -x -q -k "render or fullsize" > gpurun_out/r4_cam_tests2.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4_cam_tests2.log
for WL in c2_cornell_1080p_4spp_d8 c3_terrain1M_1080p_4spp_d8 stadium1M_1080p_4spp_d8; do
  echo "== $WL"
  ROUNDS=2 BENCH_ARGS="--no-cpu-baseline --no-isolated --workload $WL" bash tools/r3_ab_env.sh "perray PT_CAM_PACKETS=0" "g3072 PT_CAM_GRID=3072" "g5120 PT_CAM_GRID=5120" "g8192 PT_CAM_GRID=8192" "p1g5120 PT_CAM_GRID=5120 PT_LIB=$V/libptamd_per1.so" "p4g5120 PT_CAM_GRID=5120 PT_LIB=$V/libptamd_per4.so" 2>&1 | tail -6
done

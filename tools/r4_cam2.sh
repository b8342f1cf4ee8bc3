#!/bin/bash
V=$PWD/optixpathtracer_amd/variants
python -m pytest tests/test_gpu_packets.py tests/test_gpu_parity.py -x -q -k "packets or fullsize or stadium" > gpurun_out/r4_cam_tests4.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4_cam_tests4.log
for WL in c3_terrain1M_1080p_4spp_d8 stadium1M_1080p_4spp_d8 c2_cornell_1080p_4spp_d8; do
  echo "== $WL"
  ROUNDS=3 BENCH_ARGS="--no-cpu-baseline --workload $WL" bash tools/r3_ab_env.sh "nopk PT_LIB=$V/libptamd_nopk.so" "pk X=1" 2>&1 | tail -2
done

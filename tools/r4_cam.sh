#!/bin/bash
# tools/r4_cam.sh: packet traversal of camera rays (k_trace8_cam) against the per-ray kernel, parity first, then A/B on three workloads
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stadium.py tests/test_gpu_batch.py tests/test_gpu_textured.py -x -q > gpurun_out/r4_cam_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_cam_tests.log
for WL in c3_terrain1M_1080p_4spp_d8 stadium1M_1080p_4spp_d8 c2_cornell_1080p_4spp_d8; do
  echo "== $WL"
  ROUNDS=2 BENCH_ARGS="--no-cpu-baseline --workload $WL" bash tools/r3_ab_env.sh "perray_$WL PT_CAM_PACKETS=0" "packet_$WL PT_CAM_PACKETS=1" 2>&1 | tail -2
done
python - <<'PY'
import json
for wl in ("c3_terrain1M_1080p_4spp_d8","stadium1M_1080p_4spp_d8"):
    for t in ("perray","packet"):
        d=json.loads(open(f"gpurun_out/ab/{t}_{wl}.2.json").read().strip().splitlines()[-1])
        print(wl, t, d["ms_per_step"], d["kernel_ms_per_frame_isolated"])
PY

#!/bin/bash
# tools/r5_sah.sh: the GPU binned-SAH hierarchy — build phases and calibration (PT_DEBUG_BVH), parity tests with the hierarchy forced, frame times
# with LBVH | PLOC (PT_BVH_SAH=0), with the three candidates (default) and with SAH forced, on C3 and the stadium.
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
for wl in c3_terrain1M_1080p_4spp_d8 stadium1M_1080p_4spp_d8; do
  echo "== $wl: build log (second build of the process is not shown; first build)"
  PT_DEBUG_BVH=1 python bench.py $B --steps 2 --warmup 1 --workload $wl 2>&1 >/dev/null | grep "pt_bvh" | grep -v "slots used" | tail -22
done
echo "== parity with the SAH hierarchy forced"
PT_BVH_BUILDER=sah timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stadium.py tests/test_gpu_packets.py -m gpu -x -q 2>&1 | tail -3
for wl in c3_terrain1M_1080p_4spp_d8 stadium1M_1080p_4spp_d8 c2_cornell_1080p_4spp_d8; do
  echo "== $wl"
  ROUNDS=2 BENCH_ARGS="$B --steps 30 --workload $wl" bash tools/r3_ab_env.sh "two PT_BVH_SAH=0" "three X=1" "threeseg PT_BVH_CALIB=segments" "sah PT_BVH_BUILDER=sah" "lbvh PT_BVH_BUILDER=lbvh" "ploc PT_BVH_BUILDER=ploc" 2>&1 | tail -6
done

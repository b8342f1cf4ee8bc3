#!/bin/bash
# round-2 GPU session 1: parity tests, bench, timelines (N=1, 1/8 share), step histogram
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/s1_tests.log
tail -5 gpurun_out/s1_tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err && tail -c 3000 gpurun_out/s1_bench.json
for sw in 2 4 8; do
  timeout -k 10 120 python bench.py --steps 20 --warmup 5 --simulate-world $sw --no-cpu-baseline --no-isolated > gpurun_out/s1_sim$sw.json 2>> gpurun_out/s1_bench.err
  python - <<PY
import json
d=json.load(open("gpurun_out/s1_sim$sw.json")); print("simulate-world $sw:", d["ms_per_step"], "ms", d["value"], "Mrays/s")
PY
done
timeout -k 10 200 tools/timeline.sh n1 && timeout -k 10 200 tools/timeline.sh w8 --simulate-world 8 && timeout -k 10 200 tools/timeline.sh w8s1 --simulate-world 8 --streams 1
PT_LIB=$PWD/optixpathtracer_amd/variants/libptamd_dbg.so PT_DEBUG_COUNTS=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --simulate-world 8 --streams 1 > gpurun_out/s1_dbg_w8.log 2>&1
PT_LIB=$PWD/optixpathtracer_amd/variants/libptamd_dbg.so PT_DEBUG_COUNTS=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --streams 1 > gpurun_out/s1_dbg_n1.log 2>&1
grep "pt_render" gpurun_out/s1_dbg_w8.log | tail -8
echo done

#!/bin/bash
# tools/r5_wavelog.sh: per-wave logs of the traversal launches (PT_DEBUG_WAVELOG variant `wlog`) for a 1/8 share and the full C3 frame, plus
# the shipped build's times at both sizes on the same box.  Output: gpurun_out/r5_wavelog_*.txt
set -e
V=$PWD/optixpathtracer_amd/variants
mkdir -p gpurun_out
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
for W in 8 1; do
  if [ $W = 1 ]; then A=""; else A="--simulate-world $W"; fi
  python bench.py $B --steps 30 --warmup 5 $A > gpurun_out/r5_base_w$W.json 2> gpurun_out/r5_base_w$W.err
  PT_LIB=$V/libptamd_wlog.so PT_DEBUG_COUNTS=1 PT_WAVELOG=$PWD/gpurun_out/r5_wavelog_w$W.bin python bench.py $B --steps 2 --warmup 2 $A > gpurun_out/r5_stats_w$W.json 2> gpurun_out/r5_stats_w$W.err
  python tools/r5_wavelog.py gpurun_out/r5_wavelog_w$W.bin > gpurun_out/r5_wavelog_w$W.txt
  rm -f gpurun_out/r5_wavelog_w$W.bin
done
python - <<'PY'
import json
for w in (8, 1):
    d = json.loads(open(f'gpurun_out/r5_base_w{w}.json').read().strip().splitlines()[-1])
    print(w, d['ms_per_step'], d.get('step_ms'))
PY

#!/bin/bash
# tools/ab.sh "<variant names or 'default'>" [bench args]: ms/frame for N=1 and the 1/8, 1/4, 1/2 shares per library variant
cd "$(dirname "$0")/.."
VARS=$1; shift
for v in $VARS; do
  if [ "$v" = default ]; then unset PT_LIB; else export PT_LIB=$PWD/optixpathtracer_amd/variants/libptamd_$v.so; fi
  for sw in ${AB_WORLDS:-0 8}; do
    timeout -k 10 150 python bench.py --steps 20 --warmup 5 --simulate-world $sw --no-cpu-baseline --no-isolated "$@" > /tmp/ab.json 2>/tmp/ab.err || { echo "$v w$sw FAILED"; tail -3 /tmp/ab.err; continue; }
    python - "$v" "$sw" <<'PY'
import json,sys
d=json.load(open("/tmp/ab.json"))
print(f"{sys.argv[1]:>12s} world {sys.argv[2]}: {d['ms_per_step']:.3f} ms  {d['value']:.0f} Mrays/s  (device {d['frame_latency_ms']:.3f} ms)", flush=True)
PY
  done
done

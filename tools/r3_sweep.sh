#!/bin/bash
# tools/r3_sweep.sh "<variant names>": synchronous N=1, 1/8 share batched, stadium — ms per frame per library variant (default first and last)
cd "$(dirname "$0")/.."
one() { # label, extra bench args...
  lab=$1; shift
  python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-isolated --no-extra-schedules "$@" > /tmp/sw.json 2>/dev/null && python -c "import json;d=json.load(open('/tmp/sw.json'));print('%-10s %-28s %.3f ms  %.0f Mrays/s' % ('$VARIANT','$lab',d['ms_per_step'],d['value']), flush=True)" || echo "$VARIANT $lab FAILED"
}
for v in default $1 default; do
  if [ "$v" = default ]; then unset PT_LIB; else export PT_LIB=$PWD/optixpathtracer_amd/variants/libptamd_$v.so; fi
  VARIANT=$v
  one "c3 sync"
  one "c3 1/8 batch8" --simulate-world 8
  one "stadium sync" --workload stadium1M_1080p_4spp_d8
done

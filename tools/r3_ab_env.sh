#!/bin/bash
# tools/r3_ab_env.sh "TAG ENV=VAL ..." ...  → one bench line per quoted configuration (first word = tag, rest = environment), gpurun_out/ab/
mkdir -p gpurun_out/ab
for cfg in "$@"; do
  set -- $cfg; tag=$1; shift
  env "$@" python bench.py --no-extra-schedules --steps 20 --warmup 5 > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || { tail -5 gpurun_out/ab/$tag.err; exit 1; }
  python - "$tag" <<'PY'
import json,sys
t=sys.argv[1]
d=json.loads(open(f'gpurun_out/ab/{t}.json').read().strip().splitlines()[-1])
r=d.get('roofline') or {}
print(t, 'ms', d['ms_per_step'], 'Mrays', d['value'], 'trace_iso_ms', (r.get('dominant_kernel') or {}).get('avg_launch_ms'), 'shade_TF', (r.get('shade') or {}).get('achieved'), flush=True)
PY
done

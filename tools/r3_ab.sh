#!/bin/bash
# tools/r3_ab.sh NAME...  → bench lines of the default build and of variants/libptamd_NAME.so on the same box (gpurun_out/ab/)
mkdir -p gpurun_out/ab
python bench.py --no-extra-schedules --steps 20 --warmup 5 > gpurun_out/ab/base.json 2> gpurun_out/ab/base.err || exit 1
for n in "$@"; do
  PT_LIB=$PWD/optixpathtracer_amd/variants/libptamd_$n.so python bench.py --no-extra-schedules --steps 20 --warmup 5 > gpurun_out/ab/$n.json 2> gpurun_out/ab/$n.err || exit 1
done
python bench.py --no-extra-schedules --steps 20 --warmup 5 > gpurun_out/ab/base2.json 2> gpurun_out/ab/base2.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    r=d.get('roofline') or {}
    print(f.split('/')[-1], 'ms', d['ms_per_step'], 'Mrays', d['value'], 'trace_iso_ms', (r.get('dominant_kernel') or {}).get('avg_launch_ms'), 'shade', json.dumps({k:v for k,v in (r.get('shade') or {}).items() if k in ('isolated_ms','ms_per_frame','avg_launch_ms','achieved')}))
PY

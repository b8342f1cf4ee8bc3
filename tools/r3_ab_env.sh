#!/bin/bash
# tools/r3_ab_env.sh "TAG ENV=VAL ..." ...  → bench lines per quoted configuration (first word = tag, rest = environment), interleaved over
# ROUNDS rounds (default 3) so that box drift hits every configuration alike; prints every run and the per-tag medians.  gpurun_out/ab/
mkdir -p gpurun_out/ab
ROUNDS=${ROUNDS:-3}
CFGS=("$@")
for r in $(seq 1 $ROUNDS); do
for cfg in "${CFGS[@]}"; do
  set -- $cfg; tag=$1; shift
  env "$@" python bench.py --no-extra-schedules --steps 20 --warmup 5 $BENCH_ARGS > gpurun_out/ab/$tag.$r.json 2> gpurun_out/ab/$tag.$r.err || { tail -5 gpurun_out/ab/$tag.$r.err; exit 1; }
done
done
python - "$ROUNDS" "${CFGS[@]}" <<'PY'
import json,sys,statistics as st
R=int(sys.argv[1])
for cfg in sys.argv[2:]:
    t=cfg.split()[0]
    ms=[];sh=[];tr=[]
    for r in range(1,R+1):
        d=json.loads(open(f'gpurun_out/ab/{t}.{r}.json').read().strip().splitlines()[-1])
        rf=d.get('roofline') or {}
        ms.append(d['ms_per_step']); sh.append((rf.get('shade') or {}).get('achieved') or 0); tr.append((rf.get('dominant_kernel') or {}).get('avg_launch_ms') or 0)
    print(f"{t:10s} ms median {st.median(ms):.3f} all {ms}  shade_TF {st.median(sh):.3f}  trace_iso_ms {st.median(tr):.4f}", flush=True)
PY

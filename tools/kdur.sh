#!/bin/bash
# tools/kdur.sh <lib or ""> : per-dispatch durations of the first frame's kernels under rocprofv3 --kernel-trace
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/kdur_$$; rm -rf $OUT; mkdir -p $OUT
[ -n "$1" ] && export PT_LIB=$1
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline $KDUR_ARGS > $OUT/log 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
def d(r): return round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for key in ('k_shade','k_trace8<0>','k_trace8<1>','k_trace8<3>','k_generate','k_resolve'):
    rr=[r for r in rows if key in r['Kernel_Name']]
    print(key, [d(r) for r in rr[-8:]], 'vgpr', rr[-1]['VGPR_Count'] if rr else '')
PY
rm -rf $OUT

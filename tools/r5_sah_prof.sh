#!/bin/bash
# tools/r5_sah_prof.sh: kernel trace of two pt_create calls on the C3 terrain (the second build is the warm one): where the SAH hierarchy's time goes
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sah_prof; rm -rf "$OUT"; mkdir -p "$OUT"
PT_DEBUG_BVH=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/sah_prof.py > "$OUT/log" 2>&1
grep "pt_bvh\|^[0-9]" "$OUT/log" | tail -30
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):5d} total {float(r["TotalDurationNs"])/1e6:8.3f} ms avg {float(r["AverageNs"])/1e3:8.1f} us')
PY

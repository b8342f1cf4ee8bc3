#!/bin/bash
# tools/pmc.sh <tag> "<space separated counters>" [bench args]  — one PMC pass, per-kernel sums
set -e
TAG=$1; CTRS=$2; shift 2 || true
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-schedules "$@" > "$OUT/bench.log" 2>&1 || { tail -20 "$OUT/bench.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void ", "").split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for k in agg for c in agg[k]})
print("kernel," + ",".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
    print(k + "," + ",".join(f"{v.get(n,0):.4g}" for n in names))
PY
find "$OUT" -name "*counter_collection.csv" -size +1M -delete || true

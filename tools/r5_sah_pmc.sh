#!/bin/bash
# tools/r5_sah_pmc.sh: counters of the SAH builder's kernels over two builds of the C3 terrain (one PMC pass per group; counters only)
export TMPDIR=/tmp
for grp in "state SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "mem TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"; do
  set -- $grp; tag=$1; shift
  OUT=$PWD/gpurun_out/sah_pmc_$tag; rm -rf "$OUT"; mkdir -p "$OUT"
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python3 tools/sah_prof.py > "$OUT/log" 2>&1 || tail -3 "$OUT/log"
  python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
fs = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
if not fs: sys.exit("no counters: " + sys.argv[1])
agg = defaultdict(lambda: defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0][:24]
    if "sah" in k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for k in agg for c in agg[k]})
print("kernel," + ",".join(names))
for k, v in agg.items(): print(k + "," + ",".join(f"{v.get(n,0):.4g}" for n in names))
PY
  find "$OUT" -name "*counter_collection.csv" -delete
done

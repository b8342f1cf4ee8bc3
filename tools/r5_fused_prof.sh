#!/bin/bash
# tools/r5_fused_prof.sh: rocprofv3 kernel stats and wave-state counters of a 1/8 share rendered as one fused pass per frame (k_path_loop) and as
# the launch chain (PT_FUSED=0); --pmc in its own passes.  Output: gpurun_out/prof_fused/*.csv + a summary
set -e
OUT=$PWD/gpurun_out/prof_fused
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --simulate-world 8 --steps 10 --warmup 2 --no-cpu-baseline --no-isolated --no-extra-schedules"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fused" -- python3 $ARGS > "$OUT/fused.log" 2>&1
PT_FUSED=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/chain" -- python3 $ARGS > "$OUT/chain.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_state" -- python3 $ARGS > "$OUT/state.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_valu" -- python3 $ARGS > "$OUT/valu.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("fused", "chain"):
    f = glob.glob(f"{out}/{tag}/**/*kernel_stats.csv", recursive=True)[0]
    print(f"== {tag}: kernel stats (frames: 12)")
    for r in list(csv.DictReader(open(f)))[:8]:
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6} avg {float(r['AverageNs']) / 1e3:9.1f} us total {float(r['TotalDurationNs']) / 1e6:8.2f} ms {float(r['Percentage']):5.1f} %")
for tag in ("pmc_state", "pmc_valu"):
    f = glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_path_loop" in r["Kernel_Name"]:
            acc["k_path_loop"][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in acc.items():
        print(f"== {tag} {k}: " + ", ".join(f"{n} {v:.3g}" for n, v in sorted(c.items())))
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
            w = c["SQ_WAVE_CYCLES"]
            print(f"   waiting {100 * c['SQ_WAIT_ANY'] / w:.1f} % of wave cycles, issuing {100 * c['SQ_ACTIVE_INST_ANY'] / w:.1f} % (VALU {100 * c['SQ_ACTIVE_INST_VALU'] / w:.1f} %), waiting-for-instruction {100 * c['SQ_WAIT_INST_ANY'] / w:.1f} %")
        if "SQ_THREAD_CYCLES_VALU" in c:
            print(f"   lane utilisation {100 * c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):.1f} % of 64 lanes")
PY

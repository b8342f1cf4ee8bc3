#!/bin/bash
# tools/r6_s2.sh: second GPU session of round 6 — suite, SAH builder after the fix (decision records + thread-per-node children), cold pt_create
# (phases, HIP API trace), k_shade traffic attribution by ablation builds.   Output: gpurun_out/r6_s2/*
D=gpurun_out/r6_s2; mkdir -p $D
V=$PWD/optixpathtracer_amd/variants
export TMPDIR=/tmp
echo "== GPU suite"; timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $D/gpu_tests.log 2>&1; tail -3 $D/gpu_tests.log
echo "== SAH hierarchy: round-5 kernels against round 6"
SC="terrain70k stadium200k copies line cornell terrain1M stadium1M"
PT_LIB=$V/libptamd_sahr5.so timeout -k 10 300 python tools/r6_bvh_check.py export $SC > $D/sah_r5.txt 2> $D/sah_r5.err
timeout -k 10 300 python tools/r6_bvh_check.py export $SC > $D/sah_r6.txt 2> $D/sah_r6.err
paste -d'\n' $D/sah_r5.txt $D/sah_r6.txt | cut -c1-30,76-
python - $D <<'PY'
import sys
a = {l.split()[0]: l.split()[1] for l in open(sys.argv[1] + "/sah_r5.txt") if l.strip()}
b = {l.split()[0]: l.split()[1] for l in open(sys.argv[1] + "/sah_r6.txt") if l.strip()}
print("same canonical SAH trees:", {k: a.get(k) == b.get(k) for k in b})
PY
echo "== build phases, first and second pt_create of a process (C3 terrain)"
PT_DEBUG_BVH=1 timeout -k 10 200 python tools/sah_prof.py > $D/build_phases.txt 2>&1; grep "pt_bvh\|^[0-9]" $D/build_phases.txt | grep -v "slots used\|SAH cost" | tail -30
echo "== cold pt_create: HIP API + kernel trace"
timeout -k 10 200 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $D/cold -- python3 tools/sah_prof.py > $D/cold.log 2>&1; tail -2 $D/cold.log
python3 - $D/cold <<'PY'
import csv, glob, os, sys
for pat, n in (("*hip_api_stats.csv", 14), ("*kernel_stats.csv", 14)):
    fs = glob.glob(os.path.join(sys.argv[1], "**", pat), recursive=True)
    if not fs: print("missing", pat); continue
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: -float(r["TotalDurationNs"]))
    print("--", pat)
    for r in rows[:n]: print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} total {float(r["TotalDurationNs"])/1e6:9.3f} ms max {float(r["MaxNs"])/1e6:8.3f} ms')
PY
find $D/cold -name "*_trace.csv" -size +3M -delete
echo "== k_shade traffic attribution (FETCH_SIZE / WRITE_SIZE per kernel; depth 1 = the first shade launch only, identical inputs in every build)"
for depth in 1 8; do
  for v in base abl_probe abl_trinrm abl_state abl_shadow; do
    if [ $depth = 8 ] && [ $v != base ] && [ $v != abl_probe ]; then continue; fi
    L=$V/libptamd_$v.so; [ $v = base ] && L=$PWD/optixpathtracer_amd/libptamd.so
    for c in FETCH_SIZE WRITE_SIZE; do
      echo "-- depth $depth $v $c"
      PT_LIB=$L PMC_TIMEOUT=120 bash tools/pmc.sh r6abl_${v}_d${depth}_$c "$c" --depth $depth --no-isolated 2>&1 | grep -E "^kernel|k_shade|k_trace8<|k_trace8_cam|k_generate|k_resolve" 
    done
  done
done > $D/shade_traffic.txt 2>&1
tail -80 $D/shade_traffic.txt

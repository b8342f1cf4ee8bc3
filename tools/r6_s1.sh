#!/bin/bash
# tools/r6_s1.sh: first GPU session of round 6 — the GPU suite on the new sources, the SAH builder before / after (same trees? build times, cold and
# warm, with phases), the active-lane histogram of the traversal loop (sizing a several-lanes-per-ray mode), C3 and C4 shares.
# Output: gpurun_out/r6_s1/*
D=gpurun_out/r6_s1; mkdir -p $D
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
echo "== GPU suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $D/gpu_tests.log 2>&1; tail -3 $D/gpu_tests.log
echo "== SAH hierarchy: round-5 kernels against round 6 (canonical tree hashes, build times)"
SC="terrain70k stadium200k copies line cornell terrain1M stadium1M"
PT_LIB=$V/libptamd_sahr5.so timeout -k 10 300 python tools/r6_bvh_check.py export $SC > $D/sah_r5.txt 2> $D/sah_r5.err
timeout -k 10 300 python tools/r6_bvh_check.py export $SC > $D/sah_r6.txt 2> $D/sah_r6.err
paste -d'\n' $D/sah_r5.txt $D/sah_r6.txt
python - $D <<'PY'
import sys
a = {l.split()[0]: l.split()[1] for l in open(sys.argv[1] + "/sah_r5.txt") if l.strip()}
b = {l.split()[0]: l.split()[1] for l in open(sys.argv[1] + "/sah_r6.txt") if l.strip()}
print("same canonical SAH trees:", {k: a.get(k) == b.get(k) for k in b})
PY
echo "== build phases, first and second pt_create of a process (C3 terrain)"
PT_DEBUG_BVH=1 timeout -k 10 200 python tools/sah_prof.py > $D/build_phases.txt 2>&1; grep "pt_bvh\|^[0-9]" $D/build_phases.txt | grep -v "slots used" | tail -40
echo "== traversal iterations by active lanes (PT_DEBUG_WAVELOG=3)"
for cfg in "share8_fused --simulate-world 8" "share8_chain --simulate-world 8" "full"; do
  set -- $cfg; name=$1; shift
  F=1; [ $name = share8_chain ] && F=0
  PT_SCHED_TRIALS=0 PT_FUSED=$F PT_LIB=$V/libptamd_wlog3.so PT_DEBUG_COUNTS=1 timeout -k 10 200 python bench.py $B --steps 2 --warmup 2 "$@" > $D/hist_$name.json 2> $D/hist_$name.err
  echo "$name: $(grep 'iterations by lanes' $D/hist_$name.err | tail -1)"
done
echo "== bench lines"
timeout -k 10 300 python bench.py --steps 30 --warmup 5 > $D/n1_default.json 2> $D/n1_default.err
for N in 2 4 8; do timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --simulate-world $N > $D/simworld$N.json 2> $D/simworld$N.err; done
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload c4_terrain1M_4k_16spp_d8 > $D/c4_one_gpu.json 2> $D/c4.err
for N in 2 4 8; do timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-isolated --workload c4_terrain1M_4k_16spp_d8 --simulate-world $N > $D/c4_simworld$N.json 2> $D/c4_simworld$N.err; done
python - $D <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    if "hist_" in f: continue
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"FAIL",e); continue
    print(f"{os.path.basename(f)[:-5]:16s} {d['value']:9.1f} Mrays/s {d['ms_per_step']:8.3f} ms step {d.get('step_ms')} pipelined {d.get('ms_per_frame_pipelined')} batched {(d.get('batched') or {}).get('ms_per_frame')} build {d['bvh']['build_ms']} {d['bvh']['hierarchy']} sched {d.get('schedule')}")
PY

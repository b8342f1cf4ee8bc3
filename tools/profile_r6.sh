#!/bin/bash
# tools/profile_r6.sh <tag> [workload] [pmc json name]   (GPU box, from the repo root)
# rocprofv3 evidence for the bench line: kernel-trace stats of the default schedule (synchronous frames) and of a single chunk stream
# (isolated per-kernel durations), then separate PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; VALU counters 4 at a
# time; wave-state, TA/L1 and L1->L2 latency counters one pass each).  --pmc passes never carry trace domains other than the counters.
# Writes gpurun_out/prof_<tag>/ and the judged summaries: profiles/<tag>_*.  Every pass profiles the same command.
set -e
TAG=${1:-r6}
WL=${2:-c3_terrain1M_1080p_4spp_d8}
PMC=${3:-r6_pmc.json}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" profiles
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-isolated --no-extra-schedules"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/bench_stats.log" 2>&1
echo "stats done"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1" -- python3 $ARGS --streams 1 > "$OUT/bench_stats1.log" 2>&1
echo "stats1 done"
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/bench_fetch.log" 2>&1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/bench_write.log" 2>&1
echo "traffic done"
timeout -k 10 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_valu" -- python3 $ARGS > "$OUT/bench_valu.log" 2>&1
timeout -k 10 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_busy" -- python3 $ARGS > "$OUT/bench_busy.log" 2>&1
echo "valu done"
timeout -k 10 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_state" -- python3 $ARGS > "$OUT/bench_state.log" 2>&1
timeout -k 10 240 rocprofv3 --pmc TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$OUT/pmc_ta" -- python3 $ARGS > "$OUT/bench_ta.log" 2>&1
timeout -k 10 240 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/pmc_lat" -- python3 $ARGS > "$OUT/bench_lat.log" 2>&1
echo "state done"
python3 tools/summarize_r3.py "$OUT" "$TAG" "$PMC" "$WL"
mkdir -p "$OUT/judged" && cp profiles/${TAG}_* "profiles/$PMC" "$OUT/judged/"   # gpurun merges gpurun_out/ back, not profiles/
find "$OUT" -name "*_kernel_trace.csv" -size +2M -delete || true
find "$OUT" -name "*counter_collection.csv" -size +2M -delete || true
ls profiles | grep "$TAG"

#!/bin/bash
# tools/r5_node64_pmc.sh: why the one-line node is slower — traversal steps per ray (PT_DEBUG_STATS builds), wave-state / TA / L1 / fabric counters
# of k_trace8<3> with 80-byte and 64-byte nodes (one PMC pass each, counters only), C3.  Output: gpurun_out/r5_n64_*.txt
V=$PWD/optixpathtracer_amd/variants
for cfg in n80 n64; do
  PT_LIB=$V/libptamd_${cfg}stats.so PT_DEBUG_COUNTS=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --no-extra-schedules 2>&1 >/dev/null | grep "pt_render" | tail -4 > gpurun_out/r5_n64_steps_$cfg.txt
  if [ "$cfg" != n80 ]; then export PT_LIB=$V/libptamd_$cfg.so; else unset PT_LIB; fi
  bash tools/pmc.sh n64_state_$cfg "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" --no-isolated > gpurun_out/r5_n64_state_$cfg.txt 2>&1
  bash tools/pmc.sh n64_ta_$cfg "TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" --no-isolated > gpurun_out/r5_n64_ta_$cfg.txt 2>&1
  bash tools/pmc.sh n64_mem_$cfg "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" --no-isolated > gpurun_out/r5_n64_mem_$cfg.txt 2>&1
done
tail -n +1 gpurun_out/r5_n64_*_n80.txt gpurun_out/r5_n64_*_n64.txt

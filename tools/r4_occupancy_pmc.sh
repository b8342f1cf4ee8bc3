#!/bin/bash
# tools/r4_occupancy_pmc.sh: wave-state and TA counters of k_trace8<3> for the shipped build (5 waves per SIMD) and the PT8_PARK build at
# 6 waves per SIMD (tools/patches/r4_pt8_park.patch, variants/libptamd_w6p4.so) — one PMC pass each, counters only.
V=$PWD/optixpathtracer_amd/variants
for cfg in "base" "w6p4"; do
  if [ "$cfg" != base ]; then export PT_LIB=$V/libptamd_$cfg.so; else unset PT_LIB; fi
  bash tools/pmc.sh occ_state_$cfg "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" --no-isolated > gpurun_out/occ_state_$cfg.txt 2>&1
  bash tools/pmc.sh occ_ta_$cfg "TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" --no-isolated > gpurun_out/occ_ta_$cfg.txt 2>&1
  bash tools/pmc.sh occ_lds_$cfg "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD" --no-isolated > gpurun_out/occ_lds_$cfg.txt 2>&1
done
tail -n +1 gpurun_out/occ_*_base.txt gpurun_out/occ_*_w6p4.txt

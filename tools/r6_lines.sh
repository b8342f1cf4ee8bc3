#!/bin/bash
# tools/r6_lines.sh <dir>: the round's bench lines on one box (gpurun_out/<dir>/*.json): N=1 default, the other workloads, the predicted shares of
# C3 AND of C4 (the configuration BASELINE.json names for 8 GPUs), the same shares with each schedule forced, the gloo rehearsals
D=gpurun_out/${1:-r6_lines}; mkdir -p $D
python bench.py --steps 30 --warmup 5 > $D/n1_default.json 2> $D/n1_default.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload stadium1M_1080p_4spp_d8 > $D/stadium.json 2> $D/stadium.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload terrain1M_textured_1080p_4spp_d8 > $D/textured.json 2> $D/textured.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload c2_cornell_1080p_4spp_d8 > $D/c2_cornell.json 2> $D/c2.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload c4_terrain1M_4k_16spp_d8 > $D/c4_one_gpu.json 2> $D/c4.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload terrain10M_1080p_4spp_d8 > $D/terrain10M.json 2> $D/t10.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload sv4_uniform_terrain1M_4k_8spp_d4 > $D/sv4_uniform.json 2> $D/sv4u.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload sv4_foveated_terrain1M_4k_d4 > $D/sv4_foveated.json 2> $D/sv4f.err
for N in 2 4 8; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --simulate-world $N > $D/simworld$N.json 2> $D/simworld$N.err; done
for N in 2 4 8; do python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-isolated --workload c4_terrain1M_4k_16spp_d8 --simulate-world $N > $D/c4_simworld$N.json 2> $D/c4_simworld$N.err; done
for N in 4 8; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --workload stadium1M_1080p_4spp_d8 --simulate-world $N > $D/stadium_simworld$N.json 2> $D/stadium_simworld$N.err; done
# the same C3 shares with each schedule forced (the on-line choice off): what the choice is between
for N in 2 4 8; do for F in 0 1; do PT_SCHED_TRIALS=0 PT_FUSED=$F PT_FUSED_MAX_PATHS=5000000 PT_FUSED_MAX_COST=1e9 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --no-extra-schedules --simulate-world $N > $D/simworld${N}_forced$F.json 2> $D/simworld${N}_forced$F.err; done; done
python bench.py --gpus 2 --backend gloo --share-device --steps 10 --warmup 3 --no-cpu-baseline --no-isolated > $D/gloo2_selflaunch.json 2> $D/gloo2.err
python bench.py --gpus 2 --backend gloo --share-device --launch-check --launch-render > $D/launch_check2.json 2> $D/lc2.err
python - $D <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"FAIL",e); continue
    if "value" not in d: print(os.path.basename(f)[:-5], d.get("launch_check"), [ (x.get("share_equals_whole_frame"), x.get("bvh_build_ms")) for x in d.get("ranks", [])]); continue
    sc = d.get("schedule") or {}
    print(f"{os.path.basename(f)[:-5]:22s} {d['value']:9.1f} Mrays/s {d['ms_per_step']:8.3f} ms  pipelined {d.get('ms_per_frame_pipelined')}  batched {(d.get('batched') or {}).get('ms_per_frame')}  displayed {d.get('ms_per_displayed_frame')} build {d['bvh']['build_ms']} {d['bvh']['hierarchy']} sched {sc.get('timed_steps')} {sc.get('trial_chain_ms')}/{sc.get('trial_fused_ms')}")
PY

#!/bin/bash
# tools/r5_lines.sh <dir>: the round's bench lines on one box (gpurun_out/<dir>/*.json): N=1 default, the other workloads, the predicted shares
D=gpurun_out/${1:-r5_lines}; mkdir -p $D
python bench.py --steps 30 --warmup 5 > $D/n1_default.json 2> $D/n1_default.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload stadium1M_1080p_4spp_d8 > $D/stadium.json 2> $D/stadium.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload terrain1M_textured_1080p_4spp_d8 > $D/textured.json 2> $D/textured.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload c2_cornell_1080p_4spp_d8 > $D/c2_cornell.json 2> $D/c2.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload c4_terrain1M_4k_16spp_d8 > $D/c4_one_gpu.json 2> $D/c4.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload terrain10M_1080p_4spp_d8 > $D/terrain10M.json 2> $D/t10.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload sv4_uniform_terrain1M_4k_8spp_d4 > $D/sv4_uniform.json 2> $D/sv4u.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload sv4_foveated_terrain1M_4k_d4 > $D/sv4_foveated.json 2> $D/sv4f.err
for N in 2 4 8; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --simulate-world $N > $D/simworld$N.json 2> $D/simworld$N.err; done
# the same shares as launch chains (PT_FUSED=0), for the before / after of the fused bounce loop
for N in 4 8; do PT_FUSED=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --no-extra-schedules --simulate-world $N > $D/simworld${N}_chain.json 2> $D/simworld${N}_chain.err; done
python bench.py --gpus 2 --backend gloo --share-device --steps 10 --warmup 3 --no-cpu-baseline --no-isolated > $D/gloo2_selflaunch.json 2> $D/gloo2.err
python bench.py --gpus 2 --backend gloo --share-device --launch-check --launch-render > $D/launch_check2.json 2> $D/lc2.err
# the round-4 hierarchy pair (LBVH | PLOC) on the same box, for the before / after of the SAH hierarchy
PT_BVH_SAH=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --no-extra-schedules > $D/n1_r4pair.json 2> $D/n1_r4pair.err
PT_BVH_SAH=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --no-extra-schedules --workload stadium1M_1080p_4spp_d8 > $D/stadium_r4pair.json 2> $D/stadium_r4pair.err
PT_BVH_SAH=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --no-extra-schedules --workload terrain1M_textured_1080p_4spp_d8 > $D/textured_r4pair.json 2> $D/textured_r4pair.err
PT_BVH_SAH=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-isolated --no-extra-schedules --workload terrain10M_1080p_4spp_d8 > $D/terrain10M_r4pair.json 2> $D/t10_r4pair.err
PT_BVH_SAH=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --no-extra-schedules --simulate-world 8 > $D/simworld8_r4pair.json 2> $D/sw8_r4pair.err
python - $D <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"FAIL",e); continue
    if "value" not in d: print(os.path.basename(f)[:-5], d.get("launch_check"), [ (x.get("share_equals_whole_frame"), x.get("bvh_build_ms")) for x in d.get("ranks", [])]); continue
    print(f"{os.path.basename(f)[:-5]:20s} {d['value']:9.1f} Mrays/s  {d['ms_per_step']:8.3f} ms  step {d.get('step_ms')}  pipelined {d.get('ms_per_frame_pipelined')}  batched {(d.get('batched') or {}).get('ms_per_frame')}  displayed {d.get('ms_per_displayed_frame')} build {d['bvh']['build_ms']} {d['bvh']['hierarchy']}")
PY

#!/bin/bash
# tools/r3_lines.sh: the bench lines quoted in DESIGN.md section 7 (one GPU box); writes gpurun_out/r3_lines/*.json
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_lines; mkdir -p $O
python bench.py > $O/n1_default.json 2> $O/n1_default.err
for w in 2 4 8; do python bench.py --steps 32 --warmup 8 --simulate-world $w --no-cpu-baseline --no-isolated > $O/simworld$w.json 2>/dev/null; done
python bench.py --steps 24 --warmup 4 --simulate-world 8 --batch 1 --no-cpu-baseline --no-isolated > $O/simworld8_batch1.json 2>/dev/null
python bench.py --workload stadium1M_1080p_4spp_d8 > $O/stadium.json 2>/dev/null
python bench.py --workload c2_cornell_1080p_4spp_d8 --no-cpu-baseline > $O/c2.json 2>/dev/null
python bench.py --workload c4_terrain1M_4k_16spp_d8 --steps 6 --warmup 2 --no-cpu-baseline > $O/c4.json 2>/dev/null
python bench.py --workload sv4_uniform_terrain1M_4k_8spp_d4 --steps 10 --warmup 3 --no-cpu-baseline > $O/sv4_uniform.json 2>/dev/null
python bench.py --workload sv4_foveated_terrain1M_4k_d4 --steps 30 --warmup 5 --no-cpu-baseline > $O/sv4_foveated.json 2>/dev/null
python bench.py --workload sv4_foveated_terrain1M_4k_d4 --steps 30 --warmup 5 --no-cpu-baseline --frames-in-flight 2 > $O/sv4_foveated_fif2.json 2>/dev/null
for n in 2 4; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 24 --warmup 4 --backend gloo --share-device --no-cpu-baseline > $O/gloo_n$n.json 2> $O/gloo_n$n.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3_lines/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], d['ms_per_step'], 'batch', d.get('subframes_per_batch'), 'pipe', d.get('ms_per_frame_pipelined'), 'single', (d.get('single_frame_launches') or {}).get('ms_per_frame'), 'disp', d.get('ms_per_displayed_frame'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'rf', d['roofline']['frac'], (d['roofline'].get('dominant_kernel') or {}).get('frac'))
    except Exception as e: print(f, 'FAIL', e)
PY

cd $GRAFT_REPO_ROOT
for wl in c3_terrain1M_1080p_4spp_d8 stadium1M_1080p_4spp_d8 c2_cornell_1080p_4spp_d8; do
PT_DEBUG_BVH=1 python bench.py --workload $wl --steps 20 --warmup 4 --no-cpu-baseline --no-isolated 2> /tmp/e.txt > /tmp/o.json
grep "calibration" /tmp/e.txt | cut -c1-300
python -c "import json;d=json.load(open('/tmp/o.json'));print('$wl ms/step',d['ms_per_step'],'Mrays/s',d['value'],'pipelined',d['ms_per_frame_pipelined'],'bvh',d['bvh'])"
done
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_t9.log 2>&1
tail -8 gpurun_out/r3_t9.log

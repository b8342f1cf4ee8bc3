cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_t11.log 2>&1
tail -5 gpurun_out/r3_t11.log
run() { lab=$1; shift
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline "$@" > /tmp/o.json 2>/dev/null
  python -c "import json;d=json.load(open('/tmp/o.json'));i=d['kernel_ms_per_frame_isolated'] or {};print('$lab','sync',d['ms_per_step'],d['value'],'pipelined',d['ms_per_frame_pipelined'],'iso shade',i.get('shade_ms'),'trace',i.get('trace_ms'),'other',i.get('other_ms'))"
}
run c3
run stadium --workload stadium1M_1080p_4spp_d8
run c3_w8 --simulate-world 8 --no-isolated
run c3

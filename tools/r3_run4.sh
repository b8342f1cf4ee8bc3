cd $GRAFT_REPO_ROOT
for probe in 1 0; do for k in 0 1 2 3 5; do
PT_STREAM_PROBE=$probe PT_STREAM_SHIFT=$k python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-isolated > /tmp/o.json 2>/dev/null
python -c "import json;d=json.load(open('/tmp/o.json'));print('probe $probe shift $k: sync',d['ms_per_step'],'pipelined',d['ms_per_frame_pipelined'])"
done; done

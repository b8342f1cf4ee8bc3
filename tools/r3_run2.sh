cd $GRAFT_REPO_ROOT
for e in "" "PT_SIDE_STREAMS=1"; do
for w in 0 8; do
env $e python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-isolated --no-extra-schedules --simulate-world $w > /tmp/o.json 2>/dev/null
python -c "import json;d=json.load(open('/tmp/o.json'));print('$e','w$w',d['ms_per_step'],d['value'])"
done; done
KDUR_ARGS="--no-extra-schedules --streams 1" bash tools/kdur.sh ""

cd $GRAFT_REPO_ROOT
run() { lab=$1; shift
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-isolated --no-extra-schedules "$@" > /tmp/o.json 2>/dev/null
  python -c "import json;d=json.load(open('/tmp/o.json'));print('$lab','sync',d['ms_per_step'],d['value'])"
}
run default
run mp4M --max-paths 4194304
run mp5.5M --max-paths 5600000
run mp2.8M --max-paths 2800000
run streams2 --streams 2
run streams4 --streams 4
run streams4_mp --streams 4 --max-paths 5600000
run default

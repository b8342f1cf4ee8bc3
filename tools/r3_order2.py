import sys, time, os, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for extra in ([], ["--simulate-world", "8"]):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "4", "--no-cpu-baseline", "--no-isolated"] + extra, capture_output=True, text=True)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    print(extra, d["ms_per_step"], d["ms_per_frame_pipelined"], d["batched"], d["batched_pipelined"], flush=True)
    print(out.stderr[-600:])

"""tools/collect_r5.py — after `gpurun -- bash tools/r6_final.sh`: copy the judged profile files that came back under gpurun_out/ into profiles/
and bundle the bench lines into profiles/r6_13_bench_lines.json."""
import glob, json, os, shutil

for tag in ("r6_10", "r6_11_stadium"):
    for f in glob.glob(f"gpurun_out/prof_{tag}/judged/*"):
        shutil.copy(f, "profiles/")
out = {}
for f in sorted(glob.glob("gpurun_out/r6_13_lines/*.json")):
    try:
        out[os.path.basename(f)[:-5]] = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAIL", e)
json.dump(out, open("profiles/r6_13_bench_lines.json", "w"), indent=1)
for k, d in out.items():
    if "value" not in d:
        print(k, d)
        continue
    print(f"{k:18s} {d['value']:9.1f} Mrays/s {d['ms_per_step']:8.3f} ms  pipelined {d.get('ms_per_frame_pipelined')}  batched {(d.get('batched') or {}).get('ms_per_frame')}  displayed {d.get('ms_per_displayed_frame')}")
d = out.get("n1_default")
if d:
    r = d["roofline"]
    print("roofline", r["achieved"], r["frac"], "traffic", r["traffic"], "dominant", r["dominant_kernel"], "shade", r["shade"], "limiter", r["measured_limiter"])
    print("cpu", d.get("cpu_baseline"))

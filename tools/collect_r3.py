"""tools/collect_r3.py TAG_C3 TAG_STADIUM LINES_NAME — after `gpurun -- bash tools/r3_final.sh`: copy the judged profile files that came back
under gpurun_out/ into profiles/, name the builder kernels the stats CSVs list as '(anonymous namespace)::...', and bundle the bench lines."""
import csv, glob, json, os, re, shutil, sys

c3, st, lines = sys.argv[1:4]
for tag in (c3, st):
    for f in glob.glob(f"gpurun_out/prof_{tag}/judged/*"):
        shutil.copy(f, "profiles/")
out = {}
for f in sorted(glob.glob("gpurun_out/r3_lines/*.json")):
    try:
        out[os.path.basename(f)[:-5]] = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAIL", e)
json.dump(out, open(f"profiles/{lines}.json", "w"), indent=1)


def fix(tag):
    md = open(f"profiles/{tag}_summary.md").read().split("\n")
    tabs = {"default": f"profiles/{tag}_kernel_stats.csv", "one": f"profiles/{tag}_kernel_stats_streams1.csv"}
    rows = {k: [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))] for k, f in tabs.items()}
    cur, o = None, []
    for line in md:
        if line.startswith("## kernel time, default"): cur = "default"
        elif line.startswith("## kernel time, one chunk"): cur = "one"
        elif line.startswith("## "): cur = None
        m = re.match(r"\|  \| (\d+) \| ([\d.]+) \|", line)
        if cur and m:
            for n, c, t in rows[cur]:
                if c == int(m.group(1)) and abs(t - float(m.group(2))) < 0.002:
                    line = line.replace("|  |", "| " + n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60] + " |", 1)
                    break
        elif m:
            line = line.replace("|  |", "| builder kernels (anonymous namespace, merged) |", 1)
        o.append(line)
    open(f"profiles/{tag}_summary.md", "w").write("\n".join(o))


fix(c3); fix(st)
d = out["n1_default"]; r = d["roofline"]
print("n1", d["ms_per_step"], d["value"], "pipe", d.get("ms_per_frame_pipelined"), "traffic", r["traffic"], "frac", r["frac"], "dom", r["dominant_kernel"]["avg_launch_ms"], r["dominant_kernel"]["achieved"], r["dominant_kernel"]["frac"], r["dominant_kernel"].get("traffic"), "shade", r["shade"]["achieved"], r["shade"]["frac"], r["shade"]["isolated_ms_per_frame"], r["shade"]["pmc"])
print("cpu", d.get("cpu_baseline"))
s = out["stadium"]
print("stadium", s["ms_per_step"], s["value"], s.get("ms_per_frame_pipelined"), s["roofline"]["dominant_kernel"]["avg_launch_ms"], s["roofline"]["shade"]["isolated_ms_per_frame"], (s.get("cpu_baseline") or {}).get("value"))

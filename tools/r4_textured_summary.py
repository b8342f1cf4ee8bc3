#!/usr/bin/env python3
"""tools/r4_textured_summary.py <dir> <tag>: profiles/<tag>_summary.md + profiles/<tag>_lines.json from a tools/r4_textured.sh run — the textured
1 M-triangle workload beside the untextured C3: bench lines, isolated kernel durations (one chunk stream), k_shade's measured traffic per
dispatch (FETCH_SIZE x 2 + WRITE_SIZE, KiB, separate passes, gfx950 correction as in MI355X_MICROARCH.md) and its L1->L2 round trip."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from summarize_r3 import bench_line, counters, find, short  # noqa: E402


def main():
    out, tag = sys.argv[1], sys.argv[2]
    from bench import source_hash

    wls = ("c3_terrain1M_1080p_4spp_d8", "terrain1M_textured_1080p_4spp_d8")
    md = [f"# {tag}: textured 1 M-triangle workload beside C3 (kernel sources {source_hash()})\n",
          "Textured: the C3 terrain with texcoords and eight 1024^2 RGBA8 textures, written as OBJ + MTL + PNG and read back through objloader",
          "(loadOBJ semantics): every closest hit samples a texture (deviceProgram.cu:512-523); materials are what MTL carries (Kd, Ke).\n",
          "| workload | Mrays/s | ms/frame (min / median / max step) | 3 frames in flight | rays/frame | isolated trace / shade ms per frame |", "|---|---|---|---|---|---|"]
    lines = {}
    for wl in wls:
        b = bench_line(os.path.join(out, f"line_{wl}.json"))
        lines[wl] = b
        if b:
            k = b.get("kernel_ms_per_frame_isolated") or {}
            sm = b.get("step_ms") or {}
            md.append(f"| {wl} | {b['value']} | {b['ms_per_step']} ({sm.get('min')} / {sm.get('median')} / {sm.get('max')}) | {b.get('ms_per_frame_pipelined')} | {b['rays_per_frame']} | {k.get('trace_ms')} / {k.get('shade_ms')} |")
    json.dump(lines, open(os.path.join(ROOT, "profiles", f"{tag}_lines.json"), "w"), indent=1)
    frames = 7
    for wl in wls:
        st = find(os.path.join(out, f"stats1_{wl}"), "*kernel_stats.csv")
        md.append(f"\n## {wl}: kernel time, one chunk stream (--kernel-trace --stats, 7 frames)\n")
        if st:
            shutil.copy(st, os.path.join(ROOT, "profiles", f"{tag}_{'tex' if 'textured' in wl else 'c3'}_kernel_stats_streams1.csv"))
            md += ["| kernel | calls | total ms | avg us | ms / frame |", "|---|---|---|---|---|"]
            for r in list(csv.DictReader(open(st)))[:8]:
                ms = float(r["TotalDurationNs"]) / 1e6
                md.append(f"| {short(r['Name'])} | {r['Calls']} | {ms:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {ms / frames:.3f} |")
        fetch, fcnt = counters(os.path.join(out, f"fetch_{wl}"))
        write, _ = counters(os.path.join(out, f"write_{wl}"))
        lat, _ = counters(os.path.join(out, f"lat_{wl}"))
        md += ["\n| kernel | dispatches | FETCH MiB / dispatch (raw) | x2 | WRITE MiB / dispatch | FETCH x2 + WRITE, MB / dispatch | MB / frame | L1->L2 round trip, cycles |", "|---|---|---|---|---|---|---|---|"]
        for k in sorted(fetch, key=lambda k: -fetch[k]["FETCH_SIZE"])[:6]:
            n = max(1, fcnt[k]["FETCH_SIZE"])
            f = fetch[k]["FETCH_SIZE"] / n / 1024
            w = write.get(k, {}).get("WRITE_SIZE", 0.0) / n / 1024
            tot = (2 * f + w) * 2**20 / 1e6
            l = lat.get(k, {})
            rt = l.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / l["TCP_TCC_READ_REQ_sum"] if l.get("TCP_TCC_READ_REQ_sum") else 0.0
            md.append(f"| {k} | {n} | {f:.2f} | {2 * f:.2f} | {w:.2f} | {tot:.1f} | {tot * n / frames:.0f} | {rt:.0f} |")
    open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w").write("\n".join(md) + "\n")
    os.makedirs(os.path.join(out, "judged"), exist_ok=True)
    for f in os.listdir(os.path.join(ROOT, "profiles")):
        if f.startswith(tag + "_"):
            shutil.copy(os.path.join(ROOT, "profiles", f), os.path.join(out, "judged"))
    print("\n".join(md))


if __name__ == "__main__":
    main()

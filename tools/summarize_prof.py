#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel time (rocprofv3 --kernel-trace --stats) and
per-kernel HBM traffic from the two PMC passes.  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM) — both the raw and the x2 figure are
printed; gathers of 16-byte records (this kernel's pattern) are not calibrated in the guide."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:70]


def main():
    out, tag = sys.argv[1], sys.argv[2]
    print(f"# rocprofv3 summary — {tag}\n")
    for log in ("bench_stats.log",):
        p = os.path.join(out, log)
        if os.path.exists(p):
            lines = [l for l in open(p) if l.startswith("{")]
            if lines:
                print("bench line under the profiler (profiled runs clock lower; do not compare with unprofiled):\n")
                print("```\n" + lines[-1].strip() + "\n```\n")
    st = find(os.path.join(out, "stats"), "*kernel_stats.csv")
    if st:
        print("## kernel time (--kernel-trace --stats)\n")
        print("| kernel | calls | total ms | avg us | % |")
        print("|---|---|---|---|---|")
        rows = list(csv.DictReader(open(st)))
        for r in rows[:14]:
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
        print()
    traffic = {}
    for cname, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        cc = find(os.path.join(out, sub), "*counter_collection.csv")
        if not cc:
            continue
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(cc)):
            if r.get("Counter_Name") != cname:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        print(f"## {cname} per kernel (KiB summed over dispatches; separate pass)\n")
        print("| kernel | dispatches | total MiB | MiB / dispatch |")
        print("|---|---|---|---|")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
            print(f"| {k} | {n} | {v/1024:.1f} | {v/1024/max(n,1):.2f} |")
            traffic.setdefault(k, {})[cname] = {"dispatches": n, "KiB": v}
        print()
    # machine-readable per-kernel traffic for bench.py's roofline.traffic (bytes per dispatch; FETCH_SIZE doubled as
    # MI355X_MICROARCH.md §HBM prescribes for 16-B-per-lane loads, WRITE_SIZE as is; KiB -> bytes)
    import json
    out_t = {}
    for k, d in traffic.items():
        f, w_ = d.get("FETCH_SIZE"), d.get("WRITE_SIZE")
        if f and w_:
            out_t[k] = {"dispatches": f["dispatches"], "fetch_bytes_per_dispatch_x2": 2 * f["KiB"] * 1024 / f["dispatches"],
                        "write_bytes_per_dispatch": w_["KiB"] * 1024 / w_["dispatches"]}
    json.dump(out_t, open(os.path.join(out, f"traffic_{tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

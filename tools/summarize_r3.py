#!/usr/bin/env python3
"""Summarise a tools/profile_r3.sh directory into profiles/<tag>_summary.md, profiles/<tag>_kernel_stats{,_streams1}.csv and
profiles/<pmc name>.json (default r3_pmc.json) (the PMC-derived numbers bench.py quotes, tagged with the hash of the kernel sources they were measured on).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide reads by 2x (MI355X_MICROARCH.md, HBM section): the
raw and the doubled figure are both given, the doubled one is what `roofline.traffic` uses."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    # k_trace8<MODE, XW>: the default instantiation (XW = false, since round 5) keeps the name earlier rounds' files and bench.py use
    import re
    return re.sub(r"k_trace8<(\d), false>", r"k_trace8<\1>", name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60])


def bench_line(path):
    if not os.path.exists(path):
        return None
    lines = [l for l in open(path) if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


def counters(d):
    cc = find(d, "*counter_collection.csv")
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    if cc:
        for r in csv.DictReader(open(cc)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    return agg, cnt


def main():
    out, tag = sys.argv[1], sys.argv[2]
    pmc_name = sys.argv[3] if len(sys.argv) > 3 else "r3_pmc.json"
    workload = sys.argv[4] if len(sys.argv) > 4 else "c3_terrain1M_1080p_4spp_d8"
    from bench import source_hash

    md = [f"# rocprofv3 summary — {tag} (kernel sources {source_hash()})\n",
          f"Command of every pass: `rocprofv3 <mode> -- python3 bench.py --workload {workload} --steps 5 --warmup 2 --no-cpu-baseline --no-isolated --no-extra-schedules`",
          "(1 M triangles, 1920x1080, 4 spp, depth 8; 7 frames, every frame a device-synchronised pt_render).  Profiled runs clock lower than unprofiled ones; PMC passes serialise kernels.\n"]
    frames = 7
    for name, sub, log in (("default schedule of bench.py (synchronous frames: three pixel chunks on three streams)", "stats", "bench_stats.log"), ("one chunk stream (isolated kernel durations)", "stats1", "bench_stats1.log")):
        st = find(os.path.join(out, sub), "*kernel_stats.csv")
        b = bench_line(os.path.join(out, log))
        md.append(f"## kernel time, {name} (--kernel-trace --stats)\n")
        if b:
            md.append(f"bench line under the profiler: {b['ms_per_step']} ms/frame, {b['value']} Mrays/s\n")
        if st:
            dst = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats{'_streams1' if sub == 'stats1' else ''}.csv")
            shutil.copy(st, dst)
            md += ["| kernel | calls | total ms | avg us | ms / frame | % |", "|---|---|---|---|---|---|"]
            tot = 0.0
            for r in list(csv.DictReader(open(st)))[:12]:
                ms = float(r["TotalDurationNs"]) / 1e6
                tot += ms
                md.append(f"| {short(r['Name'])} | {r['Calls']} | {ms:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {ms / frames:.3f} | {float(r['Percentage']):.1f} |")
            md.append(f"\nsum of kernel durations per frame: {tot / frames:.3f} ms" + (f" (frame under the profiler: {b['ms_per_step']} ms)" if b else "") + "\n")
    fetch, fcnt = counters(os.path.join(out, "pmc_fetch"))
    write, _ = counters(os.path.join(out, "pmc_write"))
    md += ["## HBM traffic per kernel (FETCH_SIZE and WRITE_SIZE, separate passes; KiB -> MiB)\n",
           "| kernel | dispatches | FETCH MiB / dispatch (raw) | x2 (gfx950 correction) | WRITE MiB / dispatch | FETCH x2 + WRITE, MB |", "|---|---|---|---|---|---|"]
    traffic = {}
    for k in sorted(fetch, key=lambda k: -fetch[k]["FETCH_SIZE"])[:10]:
        n = max(1, fcnt[k]["FETCH_SIZE"])
        f = fetch[k]["FETCH_SIZE"] / n / 1024
        w = write.get(k, {}).get("WRITE_SIZE", 0.0) / n / 1024
        traffic[k] = {"dispatches": n, "fetch_raw_bytes": f * 2**20, "write_bytes": w * 2**20, "total_bytes_x2": (2 * f + w) * 2**20}
        md.append(f"| {k} | {n} | {f:.2f} | {2 * f:.2f} | {w:.2f} | {(2 * f + w) * 2**20 / 1e6:.1f} |")
    md.append("")
    valu, vcnt = counters(os.path.join(out, "pmc_valu"))
    busy, _ = counters(os.path.join(out, "pmc_busy"))
    md += ["## VALU counters (sums over all dispatches of the pass)\n", "| kernel | dispatches | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU | SQ_THREAD_CYCLES_VALU | lane utilisation | SQ_WAVE_CYCLES | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES |", "|---|---|---|---|---|---|---|---|"]
    lane = {}
    insts_frame = 0.0
    for k in sorted(valu, key=lambda k: -valu[k]["SQ_INSTS_VALU"])[:8]:
        v = valu[k]
        lu = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64) if v["SQ_ACTIVE_INST_VALU"] else 0.0
        if k.startswith("k_"):
            lane[k] = round(lu, 4)
        insts_frame += v["SQ_INSTS_VALU"] / frames
        wait = busy.get(k, {}).get("SQ_WAIT_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"] if v["SQ_WAVE_CYCLES"] else 0.0
        md.append(f"| {k} | {vcnt[k]['SQ_INSTS_VALU']} | {v['SQ_INSTS_VALU']:.4g} | {v['SQ_ACTIVE_INST_VALU']:.4g} | {v['SQ_THREAD_CYCLES_VALU']:.4g} | {lu * 100:.1f} % | {v['SQ_WAVE_CYCLES']:.4g} | {wait * 100:.0f} % |")
    md.append(f"\nVALU wave-instructions per frame: {insts_frame:.4g} (x 2 SIMD cycles each on the 32-lane SIMDs — one wave alone issues every 4 —, 1024 SIMDs x 2.4 GHz available)\n")
    # where a wave's time goes (SQ wave-state counters, one pass) and how loaded the vector-memory path is (TA / L1, two passes)
    state, _ = counters(os.path.join(out, "pmc_state"))
    ta, _ = counters(os.path.join(out, "pmc_ta"))
    lat, _ = counters(os.path.join(out, "pmc_lat"))
    wave_state = {}
    if state:
        md += ["## Wave-state split (SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY = SQ_WAVE_CYCLES) and vector-memory path\n",
               "| kernel | waiting on memory | ready, not issued | issuing | of which VALU | VALU pipe use (5 waves/SIMD, 2 cycles per wave64 instruction) | TA busy / L1 clocked | L1 stalled on pending misses | mean L1->L2 read round trip, cycles |",
               "|---|---|---|---|---|---|---|---|---|"]
        for k in sorted(state, key=lambda k: -state[k]["SQ_WAVE_CYCLES"]):
            r = state[k]
            w = r["SQ_WAVE_CYCLES"]
            if not k.startswith("k_") or not w:
                continue
            e = {"wait_mem": r["SQ_WAIT_ANY"] / w, "wait_issue": r["SQ_WAIT_INST_ANY"] / w, "active": r["SQ_ACTIVE_INST_ANY"] / w, "valu": r["SQ_ACTIVE_INST_VALU"] / w}
            e["valu_pipe"] = 5 * e["valu"] * 2 / 4
            t, l = ta.get(k), lat.get(k)
            if t and t["TCP_GATE_EN1_sum"]:
                e["ta_busy"] = t["TA_TA_BUSY_sum"] / t["TCP_GATE_EN1_sum"]
                e["pending_stall"] = t["TCP_PENDING_STALL_CYCLES_sum"] / t["TCP_GATE_EN1_sum"]
                e["acc_per_cycle"] = t["TCP_TOTAL_CACHE_ACCESSES_sum"] / t["TCP_GATE_EN1_sum"]
            if l and l["TCP_TCC_READ_REQ_sum"]:
                e["l2_round_trip_cycles"] = l["TCP_TCC_READ_REQ_LATENCY_sum"] / l["TCP_TCC_READ_REQ_sum"]
            wave_state[k] = {a: round(b, 3) for a, b in e.items()}
            md.append(f"| {k} | {e['wait_mem']:.1%} | {e['wait_issue']:.1%} | {e['active']:.1%} | {e['valu']:.1%} | {e['valu_pipe']:.0%} | "
                      + (f"{e['ta_busy']:.0%} | {e['pending_stall']:.0%} | " if "ta_busy" in e else "- | - | ") + (f"{e['l2_round_trip_cycles']:.0f} |" if "l2_round_trip_cycles" in e else "- |"))
        md.append("")
    b0 = bench_line(os.path.join(out, "bench_stats.log"))
    trav = [k for k in traffic if k.startswith("k_trace8")]
    tnum = sum(traffic[k]["total_bytes_x2"] * traffic[k]["dispatches"] for k in trav)
    tden = sum(traffic[k]["dispatches"] for k in trav)
    shade_k = [k for k in traffic if k.startswith("k_shade")]
    pmc = {
        "source": f"profiles/{tag}_summary.md", "src_hash": source_hash(), "frames": frames,
        "traffic_bytes_per_traversal_launch": int(tnum / tden) if tden else None,
        # whole-frame figures (FETCH x2 + WRITE of every per-frame kernel) — what bench.py sets against its per-frame algorithmic bytes
        "traffic_bytes_per_frame": int(sum(v["total_bytes_x2"] * v["dispatches"] for k, v in traffic.items() if k.startswith(("k_trace8", "k_shade", "k_generate", "k_resolve", "k_accum"))) / frames),
        "traversal_traffic_bytes_per_frame": int(sum(v["total_bytes_x2"] * v["dispatches"] for k, v in traffic.items() if k.startswith("k_trace8")) / frames),
        "per_kernel_traffic": {k: {kk: (int(vv) if kk != "dispatches" else vv) for kk, vv in v.items()} for k, v in traffic.items()},
        "valu": {"lane_util": lane, "valu_insts_per_frame": insts_frame, "simd_cycles_per_valu_inst": 2, "single_wave_issue_cycles": 4, "simds": 1024, "clock_ghz": 2.4,
                 "issue_frac_at_profiled_frame_ms": (round(insts_frame * 2 / (1024 * 2.4e9 * b0["ms_per_step"] * 1e-3), 3) if b0 else None),
                 "wave_state": wave_state or None, "source": f"profiles/{tag}_summary.md"},
        "shade": ({"fetch_x2_plus_write_bytes_per_dispatch": int(traffic[shade_k[0]]["total_bytes_x2"]), "lane_util": lane.get(shade_k[0]),
                   "source": f"profiles/{tag}_summary.md"} if shade_k else None),
    }
    pmc["workload"] = workload
    json.dump(pmc, open(os.path.join(ROOT, "profiles", pmc_name), "w"), indent=1)
    open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md[-14:]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/make_profile_docs.py <tag> <frame_ms>: turn the outputs of tools/profile.sh <tag> and the three tools/pmc.sh passes
(valu, busy, mem; text tables under gpurun_out/) into the committed artefacts under profiles/:
<tag>_final.md, <tag>_kernel_stats.csv, <tag>_pmc_valu.md, r1_traffic.json, r1_pmc_valu.json (the two bench.py reads)."""
import glob, json, shutil, sys

tag, ms = sys.argv[1], float(sys.argv[2])
src = f"gpurun_out/prof_{tag}"
shutil.copy(f"{src}/summary_{tag}.md", f"profiles/{tag}_final.md")
shutil.copy(f"{src}/traffic_{tag}.json", "profiles/r1_traffic.json")
ks = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)
if ks:
    shutil.copy(ks[0], f"profiles/{tag}_kernel_stats.csv")


def table(path):
    rows = [l.strip().split(",") for l in open(path) if l.strip() and not l.startswith("/opt")]
    hdr, out = rows[0], []
    for r in rows[1:]:
        extra = len(r) - len(hdr)  # kernel names containing commas
        out.append((",".join(r[: 1 + extra]), dict(zip(hdr[1:], map(float, r[1 + extra :])))))
    return hdr[1:], out


md = [f"# rocprofv3 PMC passes — {tag}: what bounds the hot path\n",
      "Command per pass (`tools/pmc.sh`): `rocprofv3 --pmc <4 counters> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`",
      "(C3, 3 frames = 63 `k_trace8<3>`, 72 `k_shade`, 9 `k_trace8<0>` dispatches; kernels are serialised under PMC). Sums over all dispatches:\n"]
allv = {}
for f in ("pmc_valu", "pmc_busy", "pmc_mem"):
    hdr, rows = table(f"gpurun_out/{f}.txt")
    md.append("| kernel | " + " | ".join(hdr) + " |")
    md.append("|---|" + "---|" * len(hdr))
    for name, v in rows:
        if not name or name.startswith("rocprim"):
            continue
        md.append(f"| {name} | " + " | ".join(f"{v[h]:.4g}" for h in hdr) + " |")
        allv.setdefault(name, {}).update(v)
    md.append("")
frames = 3
lane = {k: allv[k]["SQ_THREAD_CYCLES_VALU"] / (allv[k]["SQ_ACTIVE_INST_VALU"] * 64) for k in allv if "SQ_ACTIVE_INST_VALU" in allv[k]}
valu_frame = sum(allv[k]["SQ_INSTS_VALU"] for k in allv if k) / frames
trav = sum(allv[k]["SQ_INSTS_VALU"] for k in allv if k.startswith("k_trace8")) / frames
avail = 1024 * 2.4e9 * ms * 1e-3
t3 = allv["k_trace8<3>"]
md += ["## Derived\n",
       f"* **VALU lane utilisation** = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU × 64): `k_trace8<3>` {lane['k_trace8<3>']*100:.1f} %, `k_trace8<0>` (coherent camera rays) {lane['k_trace8<0>']*100:.1f} %, `k_shade` {lane['k_shade<0, false>']*100:.1f} % (first correct kernel: 26 %, `r1_01_bvh2_baseline.md`).",
       f"* **VALU issue occupancy of a frame**: {valu_frame:.3g} VALU wave-instructions per frame ({trav:.3g} in the traversal kernels) × 4 SIMD cycles (a lower bound: IEEE divide/sqrt sequences, FP64 and transcendentals take longer) = {valu_frame*4:.3g} SIMD-cycles of the 1024 SIMDs × 2.4 GHz × {ms} ms = {avail:.3g} available: **≥ {valu_frame*4/avail*100:.0f} % of all VALU issue slots of the frame are used**. This, not HBM, is the roof the path runs against; the rest is dependent-load latency (80-byte node fetches that mostly miss L2; `k_shade`'s state/triangle/probe-CDF chains).",
       f"* `k_trace8<3>`: {t3['SQ_INSTS_VMEM_RD']/t3['SQ_INSTS_VALU']*1000:.0f} VMEM reads per 1000 VALU instructions, LDS bank conflicts {t3['SQ_LDS_BANK_CONFLICT']:.3g} cycles in total, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {t3['SQ_WAIT_INST_ANY']/t3['SQ_WAVE_CYCLES']*100:.0f} %.",
       ""]
open(f"profiles/{tag}_pmc_valu.md", "w").write("\n".join(md))
json.dump({"source": f"profiles/{tag}_pmc_valu.md", "frames": frames,
           "valu_lane_util": {"k_trace8<3>": round(lane["k_trace8<3>"], 3), "k_trace8<0>": round(lane["k_trace8<0>"], 3), "k_shade": round(lane["k_shade<0, false>"], 3)},
           "valu_insts_per_frame": valu_frame, "simd_cycles_per_valu_inst": 4, "simds": 1024, "clock_ghz": 2.4},
          open("profiles/r1_pmc_valu.json", "w"), indent=1)
print(open(f"profiles/{tag}_pmc_valu.md").read()[-1600:])

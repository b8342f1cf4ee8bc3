#!/usr/bin/env python3
"""tools/r5_wavelog.py LOG.bin — per-launch summary of the per-wave log a PT_DEBUG_STATS build writes (PT_DEBUG_COUNTS=1 PT_WAVELOG=LOG.bin):
rays, waves that took work, launch span, wave durations (mean / p50 / p90 / max), share of the span an average wave spent working, and the
share of wave time spent in the stealing phase.  max/mean of the wave durations is what balancing ACROSS waves could recover."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
if len(a) == 0:
    sys.exit("empty log")
key = a[:, 0]
order = {}
rows = []
for k in np.unique(key):
    m = a[key == k]
    t0 = m[:, 2].astype(np.int64)
    t1 = m[:, 3].astype(np.int64)
    tex = m[:, 4].astype(np.int64)
    rows.append((t0.min(), k, m, t0, t1, tex))
rows.sort(key=lambda r: r[0])
T0 = rows[0][0]
print(f"{'start us':>9} {'mode':>4} {'rays':>8} {'waves':>6} {'span us':>8} {'mean':>7} {'p50':>7} {'p90':>7} {'max':>7} {'max/mean':>8} {'busy %':>7} {'steal %':>8} {'iters mean/max':>15} {'linger %':>8} {'taken':>6} {'donated':>7}")
tot_span = tot_ideal = 0.0
for (s, k, m, t0, t1, tex) in rows:
    mode = int(m[0, 1] >> np.uint64(32))
    n = int(m[0, 1] & np.uint64(0xffffffff))
    d = (t1 - t0) / 100.0  # 100 MHz -> us
    if mode == 7:  # a fused bounce-loop kernel (pt_fused.h): one entry per wave: rounds, ticks in traversal, ticks in shading
        span = (t1.max() - t0.min()) / 100.0
        live = m[:, 5] > 0
        rounds = m[live, 5].astype(np.int64)
        tr = m[live, 6].astype(np.int64) / 100.0
        sh = m[live, 7].astype(np.int64) / 100.0
        dl = d[live]
        tp = (tex[live] - t0.min()) / 100.0
        print(f"{(s - T0) / 100.0:9.1f} fused {n:8d} paths, {len(m)} waves ({int(live.sum())} with work): span {span:.1f} us, wave mean {dl.mean():.1f} p90 {np.percentile(dl, 90):.1f} max {dl.max():.1f} us; "
              f"rounds mean {rounds.mean():.1f} max {rounds.max()}; per round: {dl.sum() / rounds.sum():.1f} us = traversal {tr.sum() / rounds.sum():.1f} + shading {sh.sum() / rounds.sum():.1f} + refill/other {(dl.sum() - tr.sum() - sh.sum()) / rounds.sum():.1f}; "
              f"pool empty (as waves saw it) at {np.percentile(tp, 5):.0f}..{np.percentile(tp, 95):.0f} us")
        tot_span += span
        tot_ideal += dl.mean()
        continue
    span = (t1.max() - t0.min()) / 100.0
    steal = np.where(tex > 0, (t1 - tex) / 100.0, 0.0)
    it = m[:, 5].astype(np.int64)
    linger = np.zeros(len(m))
    taken = donated = 0
    # shader-clock cycles of the wave's traversal steps (PT_DEBUG_WAVELOG builds since the cross-wave experiment left): per iteration, over all waves
    c01 = (m[:, 6] >> np.uint64(32)).astype(np.int64).sum(); c12 = (m[:, 6] & np.uint64(0xffffffff)).astype(np.int64).sum()
    c23 = (m[:, 7] >> np.uint64(32)).astype(np.int64).sum(); cout = (m[:, 7] & np.uint64(0xffffffff)).astype(np.int64).sum()
    nit = max(1, int(m[:, 5].astype(np.int64).sum()))
    cyc = f"  cycles/iteration: select {c01 / nit:.0f} + wait {c12 / nit:.0f} + arithmetic {c23 / nit:.0f} + outside {cout / nit:.0f}"
    if "--fine" in sys.argv:  # -DPT_DEBUG_WAVELOG=2 builds: the same words hold the outer loop's parts
        cyc = f"  outer passes/wave {cout / len(m):.1f}, iterations per pass {nit / max(1, cout):.1f}; cycles per PASS: write-back {c01 / max(1, cout):.0f} + refill {c12 / max(1, cout):.0f} + steal round {c23 / max(1, cout):.0f}"
    print(f"{(s - T0) / 100.0:9.1f} {mode:4d} {n:8d} {len(m):6d} {span:8.1f} {d.mean():7.1f} {np.percentile(d, 50):7.1f} {np.percentile(d, 90):7.1f} {d.max():7.1f} "
          f"{d.max() / max(d.mean(), 1e-9):8.2f} {100 * d.mean() / max(span, 1e-9):7.1f} {100 * steal.sum() / max(d.sum(), 1e-9):8.1f} {it.mean():7.1f}/{it.max():5d} {100 * linger.sum() / max(d.sum(), 1e-9):8.1f} {taken:6d} {donated:7d}" + cyc)
    tot_span += span
    tot_ideal += d.mean()
print(f"# sum of launch spans {tot_span:.1f} us, sum of mean wave durations {tot_ideal:.1f} us ({100 * tot_ideal / tot_span:.1f} %)")

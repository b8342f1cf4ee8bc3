#!/usr/bin/env python3
"""Timeline of the last stretch of GPU activity that contains a k_resolve in a rocprofv3 kernel trace (tools/timeline.sh; back-to-back
frames form one stretch): every dispatch with its start offset, duration and queue, then per-kernel sums."""
import csv, glob, sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:34]


# segments of activity separated by > 50 us of idle; the last frame = the last segment that holds a k_resolve dispatch
segs, cur, max_end = [], [], None
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if max_end is not None and st - max_end > 50000:
        segs.append(cur)
        cur = []
    cur.append(r)
    max_end = en if max_end is None else max(max_end, en)
segs.append(cur)
fr = [sg for sg in segs if any("k_resolve" in r["Kernel_Name"] for r in sg)][-1]
t0 = int(fr[0]["Start_Timestamp"])
tend = max(int(r["End_Timestamp"]) for r in fr)
print(f"# last frame: {len(fr)} dispatches, span {(tend - t0) / 1e3:.1f} us")
busy = 0
events = []
for r in fr:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    events.append((s, e))
    print(f"{s / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size_X', '?'):>8}  {short(r['Kernel_Name'])}")
# union of busy intervals
events.sort()
cur_s, cur_e = events[0]
for s, e in events[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"# GPU busy (union of dispatches) {busy / 1e3:.1f} us of {(tend - t0) / 1e3:.1f} us; sum of durations {sum(e - s for s, e in events) / 1e3:.1f} us")
by = {}
for r in fr:
    k = short(r["Kernel_Name"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    by.setdefault(k, []).append(d)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"# {k:36s} n={len(v):3d} sum {sum(v) / 1e3:9.1f} us  mean {sum(v) / len(v) / 1e3:8.1f} us  max {max(v) / 1e3:8.1f} us")

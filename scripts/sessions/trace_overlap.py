#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of the bench command: per match_walkers launch (one per cycle), the serve launches that ran INSIDE its
interval — the evidence that evaluation and merge launches overlap the persistent walkers.
usage: trace_overlap.py <dir with *_kernel_trace.csv> [out.txt]"""
import csv, glob, os, sys
src = sys.argv[1]
files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort(key=lambda r: r[1])
walk = [r for r in rows if "match_walkers" in r[0]]
out = []
for k, (name, s, e, q) in enumerate(walk):
    inside = [r for r in rows if ("match_serve_eval" in r[0] or "match_serve_merge" in r[0]) and r[1] >= s and r[2] <= e]
    ev = [r for r in inside if "serve_eval" in r[0]]
    mg = [r for r in inside if "serve_merge" in r[0]]
    queues = sorted({r[3] for r in inside})
    # how many serve launches are in flight at the same time (the three servers side by side)
    events = sorted([(r[1], 1) for r in inside] + [(r[2], -1) for r in inside])
    cur = peak = 0
    busy = 0
    last = None
    for t, d in events:
        if cur > 0 and last is not None:
            busy += t - last
        cur += d
        peak = max(peak, cur)
        last = t
    out.append(f"walkers launch {k}: {(e - s) / 1e6:.2f} ms on queue/stream {q}; inside it {len(ev)} match_serve_eval ({sum(r[2] - r[1] for r in ev) / 1e6:.2f} ms summed) and "
               f"{len(mg)} match_serve_merge ({sum(r[2] - r[1] for r in mg) / 1e6:.2f} ms summed) launches on queues/streams {queues}; at least one serve launch running "
               f"{busy / 1e6:.2f} ms of it, up to {peak} at once")
if walk:
    s, e = walk[-1][1], walk[-1][2]
    out.append("")
    out.append("the first 60 dispatches inside the last walkers launch (microseconds from its start: start, end, queue/stream, kernel):")
    n = 0
    for r in rows:
        if r[1] >= s and r[2] <= e and "match_walkers" not in r[0]:
            out.append(f"  {(r[1] - s) / 1e3:9.1f} {(r[2] - s) / 1e3:9.1f}  {r[3]:>4}  {r[0][:60]}")
            n += 1
            if n >= 60:
                break
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")

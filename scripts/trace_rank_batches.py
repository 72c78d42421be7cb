#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of the bench command: the rank phase of a cycle — the batched launches (cook_multi<&kernel, ...>,
cook_amd/csrc/multi.hpp) between the end of one match_walkers launch and the start of the next — per stream: how many launches, how long
they run summed, how much of the phase at least one / several of the batches' launches are running (the rank parts of a GPU's
pools as four batches of two pools, each a sequence of dependent launches on its own stream).
usage: trace_rank_batches.py <dir with *_kernel_trace.csv> [out.txt]"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short_kernel_name

src = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort(key=lambda r: r[1])
walk = [r for r in rows if "match_walkers" in r[0]]
out = []
for k in range(1, len(walk)):
    lo, hi = walk[k - 1][2], walk[k][1]
    ph = [r for r in rows if "cook_multi" in r[0] and r[1] >= lo and r[2] <= hi]
    if not ph:
        continue
    t0, t1 = min(r[1] for r in ph), max(r[2] for r in ph)
    streams = sorted({r[3] for r in ph})
    events = sorted([(r[1], 1) for r in ph] + [(r[2], -1) for r in ph])
    cur = 0
    last = None
    at_least = {1: 0, 2: 0, 3: 0}
    peak = 0
    for t, d in events:
        if last is not None:
            for n in at_least:
                if cur >= n:
                    at_least[n] += t - last
        cur += d
        peak = max(peak, cur)
        last = t
    out.append(f"rank phase before walkers launch {k}: {len(ph)} batched launches on streams {streams}, first start to last end {(t1 - t0) / 1e3:.1f} us; "
               f"their durations summed {sum(r[2] - r[1] for r in ph) / 1e3:.1f} us; at least one running {at_least[1] / 1e3:.1f} us, at least two "
               f"{at_least[2] / 1e3:.1f} us, at least three {at_least[3] / 1e3:.1f} us, up to {peak} at once")
    for s in streams:
        mine = [r for r in ph if r[3] == s]
        out.append(f"    stream {s}: {len(mine)} launches, {sum(r[2] - r[1] for r in mine) / 1e3:.1f} us summed, from {(min(r[1] for r in mine) - t0) / 1e3:.1f} to "
                   f"{(max(r[2] for r in mine) - t0) / 1e3:.1f} us")
if len(walk) >= 2:
    lo, hi = walk[-2][2], walk[-1][1]
    ph = [r for r in rows if "cook_multi" in r[0] and r[1] >= lo and r[2] <= hi]
    if ph:
        t0 = min(r[1] for r in ph)
        out.append("")
        out.append("the first 80 batched launches of the last rank phase (microseconds from its first launch: start, end, stream, kernel):")
        for r in ph[:80]:
            out.append(f"  {(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {r[3]:>5}  {short_kernel_name(r[0])}")
text = "\n".join(out) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text[:6000])

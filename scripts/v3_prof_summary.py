#!/usr/bin/env python3
"""Prints the phase table of a COOK_V3_PROF run (the JSON lines libcookmatch writes to COOK_V3_PROF_FILE): last call of the file."""
import json
import sys

NAMES = {0: "h: job record / constraint form / group hosts", 1: "h: block bounds", 2: "h: block selection", 3: "h: free-resource look-up",
         4: "h: offer evaluation", 5: "h: list merge", 6: "h: publish", 7: "h: waiting for the window", 8: "h: insertions (count)",
         16: "w: waiting / skipping", 17: "w: job load + touched offers", 18: "w: fast decision", 19: "w: general decision",
         20: "w: commit, touched lane", 21: "w: commit, new lane", 22: "w: unmatched"}
d = [json.loads(l) for l in open(sys.argv[1]) if l.strip()][-1]
print("K", d["K"], "M", d["M"], "total_us", d["total_us"])
for i, (c, n) in enumerate(zip(d["cyc"], d["cnt"])):
    if c or n:
        print(f"{i:3d} {NAMES.get(i, ''):48s} events {n:9d}  cycles {c:13d}  per event {c / max(n, 1):9.0f}")

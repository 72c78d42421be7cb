#!/usr/bin/env python3
"""Summarises a COOK_ROUND_LOG csv (one line per placement round of the last match): per phase / per stop reason."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = len(rows)
def agg(rs, tag):
    if not rs: return
    f = lambda k: sum(float(r[k]) for r in rs)
    print(tag, "rounds", len(rs), "resolved", int(f("resolved")), "visited", int(f("n_list")), "matched", int(f("matched")), "touched", int(f("touched")),
          "setup_ms %.1f seq_ms %.1f" % (f("setup_us")/1e3, f("seq_us")/1e3), "avg wcur %.0f" % (f("wcur")/len(rs)), "us/visit %.2f" % (f("seq_us")/max(1,f("n_list"))))
agg(rows, "all")
ph1 = [r for r in rows if int(r["head"]) < 47000]; ph2 = [r for r in rows if int(r["head"]) >= 47000]
agg(ph1, "phase1(head<47k)"); agg(ph2, "phase2")
for s in range(5):
    agg([r for r in rows if int(r["stop"]) == s], "stop=%d" % s)
for i in list(range(0, 10)) + list(range(200, 206)) + list(range(n-4, n)):
    print(rows[i])

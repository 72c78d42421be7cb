#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection CSV per kernel: mean of every counter per dispatch (+ dispatch count)."""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short_kernel_name  # (the batched path's kernels are instances of one template: the inner name is the kernel)

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in files:
    for r in csv.DictReader(open(f)):
        k = short_kernel_name(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[: int(sys.argv[2]) if len(sys.argv) > 2 else 6]:
    print(k, {c: round(v / cnt[k][c], 1) for c, v in agg[k].items()}, "dispatches", max(cnt[k].values()))

#!/usr/bin/env python3
"""Rebalancer preemption sweep (BASELINE.json configs[4]): R running tasks + P pending jobs examined,
one cook_rebalance_run per step with inputs resident in HBM.  Prints one JSON line (reported beside bench.py's headline,
never instead of it).  `--check` compares the decisions with the CPU oracle (bit-exact) on the same inputs and times it.

Algorithmic bytes: B_rebal(R, P) = 52·R (init: keys + resources in, order + DRU out) + P·40·R (per pending
job every scored task's dru/mem/cpus/gpus/host/user is read once).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--running", type=int, default=1_000_000)
    ap.add_argument("--pending", type=int, default=128, help="pending jobs examined (= max-preemption)")
    ap.add_argument("--users", type=int, default=10_000)
    ap.add_argument("--hosts", type=int, default=50_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spare-frac", type=float, default=0.0, help="hosts with spare capacity; 0 = every decision has to preempt")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--kernels", action="store_true", help="one more profiled call: per-kernel microseconds and launches")
    args = ap.parse_args()
    import torch
    assert torch.cuda.is_available(), "needs an MI355X (no CPU fallback)"
    from cook_amd.engine import Engine
    from tests import parity_cases as P

    b = P.make_rebalance_case(seed=0xC00C0005, n_running=args.running, n_pending=args.pending, n_users=args.users,
                              n_hosts=args.hosts, max_preemption=args.pending, quota_frac=0.02, spare_frac=args.spare_frac)
    with Engine(b["params"]) as e:
        t0 = time.perf_counter()
        e.rebalance_stage(b["running"], b["pending"], b["pending_job_id"], b["pending_priority"], b["users"], b["spare"],
                          b["rparams"], host_attrs=b["host_attrs"], groups=b["groups"])
        stage_s = time.perf_counter() - t0
        for _ in range(args.warmup):
            e.rebalance_run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev = []
        for _ in range(args.steps):
            e.rebalance_run()
            ev.append(e.rebalance_timing())
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps
        got = e.rebalance_fetch()
        kern = None
        if args.kernels:
            e.set_profiling(True)
            e.rebalance_run()
            kern = {k: [round(v[0] * 1e3, 1), v[1]] for k, v in sorted(e.kernel_timings().items(), key=lambda kv: -kv[1][0])[:14]}
            e.set_profiling(False)
    nd = len(got["decisions"])
    nbytes = 52 * args.running + args.pending * 40 * args.running
    out = {"metric": "rebalancer sweep: cook_rebalance_run calls/sec", "value": 1.0 / wall, "unit": "calls/s",
           "ms_per_call": wall * 1e3, "hip_event_ms": sorted(ev)[len(ev) // 2], "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{args.running} running tasks + {args.pending} pending jobs examined, {args.users} users, "
                                  f"{args.hosts} hosts", "decisions": nd,
                      "preempted": sum(len(d["tasks"]) for d in got["decisions"])},
           "stage_s": stage_s, "kernels_us_and_launches_per_call": kern,
           "roofline": {"bound": "hbm", "achieved": nbytes / wall / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": nbytes / wall / 1e9 / 8000.0, "algorithmic_bytes_per_call": nbytes, "traffic": None}}
    if args.check:
        from oracle import pyoracle
        t0 = time.perf_counter()
        want = pyoracle.rebalance(b["params"], b["running"], b["pending"], b["pending_job_id"], b["pending_priority"], b["users"],
                                  b["spare"], b["rparams"], host_attrs=b["host_attrs"], groups=b["groups"])
        cpu_s = time.perf_counter() - t0
        P._rebal_equal(got, want, "C5")
        out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "calls/s", "cores": 1, "kind": "port",
                               "sample": "the oracle on the same inputs, whole call"}
        out["speedup_vs_cpu_baseline"] = cpu_s / wall
        out["parity"] = "decisions, preempted tasks and pending DRUs bit-identical to the oracle"
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Offer construction (cook_offers_run) on one MI355X: the whole 50k-node cluster of BASELINE.json's configs[3] with 400k pods
(= its running tasks), inputs resident in HBM, kernels only.  Prints one JSON line; --check compares with the oracle."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cook_amd import _abi as A  # noqa: E402
from cook_amd import synth  # noqa: E402
from cook_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=50000)
    ap.add_argument("--pods", type=int, default=400000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    nodes, pods, op = synth.make_cluster_state(seed=0xC00C, n_nodes=a.nodes, n_pods=a.pods, disk=True, n_attr_keys=8, max_pods=110)
    # algorithmic bytes: every input column once + the offer rows once (cookmatch.h cook_nodes / cook_pods / cook_node_offers)
    b_in = a.nodes * (8 + 8 + 4 + 4 + 8 + 4 + 1 + 4 + 8 * 4) + a.pods * (4 + 8 + 8 + 4 + 4 + 8 + 4 + 1)
    with Engine(A.default_params()) as e:
        e.offers_stage(nodes, pods, op)
        for _ in range(a.warmup):
            e.offers_run()
        e.set_profiling(True)
        ms = []
        for _ in range(a.steps):
            e.offers_run()
            ms.append(e.offers_timing())
        kt = e.kernel_timings()
        got = e.offers_fetch()
        b_out = got.n * (4 + 4 + 8 + 8 + 4 + 8 + 4 + 8 + 4 + 8 * 4) + a.nodes
        out = dict(what="cook_offers_run", nodes=a.nodes, pods=a.pods, offers=int(got.n), ms_per_call=float(np.median(ms)),
                   algorithmic_bytes=b_in + b_out, achieved_GBps=(b_in + b_out) / (float(np.median(ms)) * 1e-3) / 1e9,
                   peak_GBps=8000.0, kernels_us_per_call={k: round(v[0] * 1e3 / a.steps, 2) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])
                                                          if k.startswith(("offers", "radix", "iota"))})
        if a.check:
            from oracle import k8s_offers
            t0 = time.time()
            want = k8s_offers.build_rows(nodes, pods, op)
            out["oracle_s"] = round(time.time() - t0, 2)
            ok = all(np.array_equal(getattr(got, k), v) for k, v in want["rows"].items()) and np.array_equal(got.node_status, want["status"])
            ok = ok and all(got.totals[k] == v for k, v in want["totals"].items())
            out["identical_to_oracle"] = bool(ok)
            out["speedup_vs_oracle"] = round(out["oracle_s"] * 1e3 / out["ms_per_call"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Class-ordered best fit (match_algo 3) against the oracle: seeded random configurations, then BASELINE's C2 / one C4 pool / K = 1000 with
timings and the walk's own statistics.  `--emu` runs the SIMT-emulator build on the CPU (small sizes only), else libcookmatch.so on the GPU.
TEST TOOL (uses the oracle)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--fuzz", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--big", action="store_true", help="C2, one C4 pool (all jobs and K = 1000), timed")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--lib", default="", help="a library variant: cook_amd/libcookmatch_<name>.so (scripts/build_variant.sh)")
    ap.add_argument("--only", default="", help="substring of the big cases to run")
    args = ap.parse_args()
    from cook_amd import _abi as A
    from cook_amd import synth, workload
    from cook_amd.engine import Engine
    from oracle import pyoracle
    if args.emu:
        from tests.simt_emu import build_emu
        so = build_emu.build()
    else:
        from cook_amd import build
        so = build.build()
        if args.lib:
            so = os.path.join(ROOT, "cook_amd", f"libcookmatch_{args.lib}.so")
    rng = np.random.default_rng(args.seed)
    used = 0
    sc = args.scale
    for it in range(args.fuzz):
        kw = dict(seed=int(rng.integers(1, 1 << 30)), n_pending=int(rng.integers(1, int(700 * sc))), n_running=int(rng.integers(0, int(150 * sc))),
                  n_users=int(rng.integers(1, 30)), n_offers=int(rng.integers(1, int(500 * sc))), gpus=bool(rng.integers(0, 2)),
                  constraints=bool(rng.integers(0, 2)), fractional=bool(rng.integers(0, 8) == 0), tie_heavy=bool(rng.integers(0, 2)))
        pool = synth.make_pool(**kw)
        p = A.default_params(good_enough_fitness=1.0, match_algo=3)
        with Engine(p, lib_path=so) as e:
            j2o, fail, head = e.match(pool.pending_jobs, pool.offers, pool.groups, ())
            stt = e.match_stats()
        o = pyoracle.match(p, pool.pending_jobs, pool.offers, pool.groups, ())
        ok = np.array_equal(j2o, o[0]) and np.array_equal(fail, o[1]) and head == o[2]
        used += stt.get("placement_form") == 3
        if not ok:
            bad = np.nonzero((j2o != o[0]) | (fail != o[1]))[0]
            print(f"MISMATCH configuration {it} {kw}: first at {bad[:8]}: {j2o[bad[:8]]} vs {o[0][bad[:8]]}, fail {fail[bad[:8]]} vs {o[1][bad[:8]]}; stats {stt}")
            sys.exit(1)
    print(f"fuzz: {args.fuzz} configurations identical to the oracle, {used} of them placed by class-ordered best fit (seed {args.seed})")
    if not args.big:
        return
    p = A.default_params(good_enough_fitness=1.0, match_algo=3)

    def ranked_jobs(pool, K=None):
        ranked, _ = pyoracle.rank(p, pool.tasks, pool.users)
        idx = (np.cumsum(pool.tasks.pending) - 1)[ranked]
        return pool.pending_jobs.take(idx if K is None else idx[:K])

    cases = []
    c2 = synth.make_pool(seed=0xC00C0002, n_pending=50_000, n_running=20_000, n_users=1000, n_offers=5000)
    cases.append(("C2 50k x 5k", ranked_jobs(c2), c2.offers, c2.groups))
    c4 = workload.make_pool(workload.ClusterSpec(), 0)
    cases.append(("C4 pool 0", ranked_jobs(c4), c4.offers, c4.groups))
    cases.append(("C4 pool 0, K = 1000", ranked_jobs(c4, 1000), c4.offers, c4.groups))
    for tag, jobs, offers, groups in cases:
        if args.only and args.only not in tag:
            continue
        o = pyoracle.match(p, jobs, offers, groups, ())
        for algo in (3, 2):
            pp = A.default_params(good_enough_fitness=1.0, match_algo=algo)
            with Engine(pp, lib_path=so) as e:
                e.match_stage(jobs, offers, groups, ())
                ts = []
                for _ in range(args.reps):
                    t0 = time.perf_counter()
                    e.match_run()
                    ts.append((time.perf_counter() - t0) * 1e3)
                j2o, fail, head = e.match_fetch()
                stt = e.match_stats()
            ok = np.array_equal(j2o, o[0]) and np.array_equal(fail, o[1]) and head == o[2]
            cf = {k: v for k, v in stt.items() if k.startswith("cf_") or k in ("placement_form", "classfit_refused")}
            print(f"{tag}: match_algo {algo}: {'identical to the oracle' if ok else 'MISMATCH'}; match_run ms {['%.2f' % t for t in ts]} (min {min(ts):.2f}); {cf if algo == 3 else ''}")
            if not ok:
                sys.exit(1)


if __name__ == "__main__":
    main()

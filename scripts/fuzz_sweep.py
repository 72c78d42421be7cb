#!/usr/bin/env python3
"""Seeded random sweep of the engine against the oracle (TEST TOOL): small random configurations of rank / cycle / match / explain /
cycle update and of the rebalancer, both match_algo values, eval split caps, ports / named scalars, k8s gpu maps with several entries.
`--emu` runs the SIMT-emulator build on the CPU, otherwise libcookmatch.so on the GPU.  Prints one line; exit 1 on the first
difference (with the configuration that produced it)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--match", type=int, default=60, help="match / cycle configurations")
    ap.add_argument("--rebalance", type=int, default=60, help="rebalancer configurations")
    ap.add_argument("--multi", type=int, default=0, help="multi-pool configurations (2-8 random pools through ONE cook_cycle_match_multi: served walkers with "
                                                          "1-3 serve streams, or lockstep launches; pools that disagree on good-enough / K)")
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--scale", type=float, default=1.0, help="size factor of the configurations")
    ap.add_argument("--algo", type=int, default=-1, help="force cook_params.match_algo (default: drawn per configuration)")
    ap.add_argument("--ge", type=float, default=-1.0, help="force good-enough-fitness (default: drawn per configuration)")
    ap.add_argument("--guard", action="store_true", help="run under COOK_GUARD=1 (every device buffer between two bands of a pattern) and fail on any write found outside a buffer")
    ap.add_argument("--only", type=int, default=-1, help="run only that match configuration (the random draws of the others are still made)")
    args = ap.parse_args()
    if args.guard:
        os.environ["COOK_GUARD"] = "1"  # read once when the library is loaded
    from cook_amd import _abi as A
    from cook_amd import synth
    from cook_amd.engine import Engine
    from tests import parity_cases as P
    if args.emu:
        from tests.simt_emu import build_emu
        so = build_emu.build()
    else:
        from cook_amd import build
        so = build.build()
    make_engine = lambda params: Engine(params, lib_path=so)  # noqa: E731
    rng = np.random.default_rng(args.seed)
    sc = args.scale
    for it in range(args.match):
        kw = dict(seed=int(rng.integers(1, 1 << 30)), n_pending=int(rng.integers(1, int(600 * sc))), n_running=int(rng.integers(0, int(150 * sc))),
                  n_users=int(rng.integers(1, 30)), n_offers=int(rng.integers(1, int(400 * sc))), gpus=bool(rng.integers(0, 2)),
                  constraints=bool(rng.integers(0, 2)), fractional=bool(rng.integers(0, 2)), tie_heavy=bool(rng.integers(0, 2)),
                  no_shares=bool(rng.integers(0, 4) == 0), quota_frac=float(rng.choice([0.0, 0.02, 0.5])))
        p = A.default_params(good_enough_fitness=float(rng.choice([1.0, 1.0, 0.8, 0.5, 0.3])), match_algo=int(rng.choice([0, 0, 0, 0, 1])),
                             max_over_quota_jobs=int(rng.choice([0, 3, 100])))
        if args.algo >= 0:
            p.match_algo = args.algo
        if args.ge >= 0:
            p.good_enough_fitness = args.ge
        os.environ["COOK_EVAL_SPLIT"] = str(int(rng.choice([1, 2, 4])))
        pool = synth.make_pool(**kw)
        if rng.integers(0, 3) == 0:  # ports / named scalars
            n, m = pool.pending_jobs.n, pool.offers.n
            s2 = rng.integers(1, 30, (n, 2)).astype(np.float64) * 0.5
            s2[rng.random((n, 2)) < 0.5] = np.nan
            pool.pending_jobs.ports = np.where(rng.random(n) < 0.3, rng.integers(1, 4, n), 0).astype(np.int32)
            pool.pending_jobs.scalars, pool.pending_jobs.n_scalars = s2, 2
            pool.offers.ports = rng.integers(0, 7, m).astype(np.int32)
            pool.offers.scalars, pool.offers.n_scalars = rng.integers(0, 120, (m, 2)) * 0.5, 2
        try:
            k_cycle = int(rng.integers(1, kw["n_pending"] + 1))
            reserved = tuple(int(x) for x in rng.integers(0, kw["n_offers"], int(rng.integers(0, 3))))
            if args.only >= 0 and it != args.only:
                continue
            P.rank_parity(make_engine, pool, p)
            P.cycle_parity(make_engine, pool, p, k_cycle)
            j2o = P.match_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, p, reserved=reserved)
            if (j2o < 0).any():
                P.explain_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, p, max_pos=6, tag=str(kw))
            if it % 4 == 0:  # cook_cycle_update against a restage of the updated arrays and the oracle
                npd, nrn = int(rng.integers(1, int(500 * sc))), int(rng.integers(0, int(200 * sc)))
                P.cycle_update_parity(make_engine, seed=int(rng.integers(1, 1 << 30)), n_pending=npd, n_running=nrn, n_users=int(rng.integers(1, 30)),
                                      n_offers=int(rng.integers(1, int(200 * sc))), n_remove=int(rng.integers(0, npd + nrn + 1)),
                                      n_add=int(rng.integers(0, int(300 * sc))), new_offers=bool(rng.integers(0, 2)),
                                      k=int(rng.choice([10 ** 9, 50])))
        except AssertionError as ex:
            print("FAIL match", it, kw, p.match_algo, p.good_enough_fitness, os.environ["COOK_EVAL_SPLIT"], str(ex)[:300])
            sys.exit(1)
    os.environ.pop("COOK_EVAL_SPLIT", None)
    for it in range(args.multi):
        n = int(rng.integers(2, 9))
        pools, params, ks = [], [], []
        ge_all = float(rng.choice([1.0, 1.0, 0.8, 0.5]))
        mixed = bool(rng.integers(0, 3) == 0)
        for i in range(n):
            npd = int(rng.integers(1, int(1500 * sc)))
            pools.append(synth.make_pool(seed=int(rng.integers(1, 1 << 30)), n_pending=npd, n_running=int(rng.integers(0, int(300 * sc))),
                                         n_users=int(rng.integers(1, 30)), n_offers=int(rng.integers(1, int(500 * sc))), gpus=bool(rng.integers(0, 2)),
                                         constraints=bool(rng.integers(0, 2)), fractional=bool(rng.integers(0, 2)), tie_heavy=bool(rng.integers(0, 2))))
            # (the form per pool: window rounds, the engine's choice — class-ordered best fit from six engines on —, class-ordered best fit where the pool allows it:
            #  one call then holds pools of both forms)
            params.append(A.default_params(good_enough_fitness=float(rng.choice([1.0, 0.8, 0.5])) if mixed else ge_all, match_algo=int(rng.choice([2, 2, 0, 3]))))
            ks.append(int(rng.choice([10 ** 9, 10 ** 9, int(rng.integers(1, npd + 1))])))
        os.environ["COOK_MATCH_SERVED"] = str(int(rng.integers(0, 4) != 0))
        os.environ["COOK_SERVE_STREAMS"] = str(int(rng.integers(1, 4)))
        try:
            P.mixed_chain_parity(make_engine, pools, params, ks, rank_batched=bool(it % 2))  # (every other one: the rank parts as one pool batch)
        except AssertionError as ex:
            print("FAIL multi", it, n, os.environ["COOK_MATCH_SERVED"], os.environ["COOK_SERVE_STREAMS"], it % 2, str(ex)[:300])
            sys.exit(1)
    os.environ.pop("COOK_MATCH_SERVED", None)
    os.environ.pop("COOK_SERVE_STREAMS", None)
    for it in range(args.rebalance):
        kw = dict(seed=int(rng.integers(1, 1 << 30)), n_running=int(rng.integers(0, int(900 * sc))), n_pending=int(rng.integers(1, 40)),
                  n_users=int(rng.integers(1, 25)), n_hosts=int(rng.integers(1, int(70 * sc))), fractional=bool(rng.integers(0, 2)),
                  constraints=bool(rng.integers(0, 2)), gpus=bool(rng.integers(0, 2)), spare_frac=float(rng.choice([0.0, 0.3, 1.0])),
                  quota_frac=float(rng.choice([0.0, 0.1, 0.5])), max_preemption=int(rng.choice([4, 16, 64])), dru_mode=int(rng.integers(0, 2)),
                  gpu_slots=int(rng.choice([1, 1, 2])))
        if not kw["constraints"]:
            kw["gpu_slots"] = 1
        try:
            P.rebalance_parity(make_engine, P.make_rebalance_case(**kw))
        except AssertionError as ex:
            print("FAIL rebalance", it, kw, str(ex)[:300])
            sys.exit(1)
    guard = ""
    if args.guard:
        import gc
        gc.collect()  # the bands are looked at when a buffer is freed
        e = make_engine(A.default_params())
        hits = e.match_stats().get("guard_hits", -1)
        del e
        if hits != 0:
            print(f"FAIL guard: {hits} writes outside a device buffer (COOK_GUARD lines on stderr)")
            sys.exit(1)
        guard = ", COOK_GUARD=1: no write outside a device buffer"
    print(f"fuzz ok: {args.match} match / cycle configurations, {args.multi} multi-pool configurations, {args.rebalance} rebalancer configurations, seed {args.seed}, "
          f"{'emulator' if args.emu else 'gpu'}{guard}")


if __name__ == "__main__":
    main()

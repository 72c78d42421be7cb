#!/usr/bin/env python3
"""Gate for the class-ordered best-fit placement (Next 1): run the CPU model (classfit_model.cpp) on BASELINE's C2, C3, one C4
pool and a seeded fuzz, compare every placement and fail code with the oracle, print the model's wave-step statistics.
TEST TOOL (uses the oracle).  `python tests/classfit_model/run_model.py [--fuzz N] [--big]`"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from cook_amd import _abi as A  # noqa: E402
from cook_amd import synth, workload  # noqa: E402
from oracle import pyoracle  # noqa: E402

NAMES = ["walked", "presettled", "matched", "overlay_wins", "opens", "gpu_places", "epochs", "scans", "empty_scans", "crit_scans_matched", "band_events",
         "literal_evals", "tightens", "slow_jobs", "slow_scans", "classes", "chunks", "waves", "walked_unmatched", "crit_scans_unmatched", "max_crit_scans",
         "dead_drops", "open_dead", "kc", "km", "gpu_unmatched", "crit_steps_matched", "crit_steps_unmatched", "slow_unmatched"]


def lib():
    so = os.path.join(HERE, "libclassfit_model.so")
    src = os.path.join(HERE, "classfit_model.cpp")
    deps = [src, os.path.join(ROOT, "oracle", "cook_oracle.cpp"), os.path.join(ROOT, "include", "cookmatch.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so, src])
    return C.CDLL(so)


def model_match(params, jobs, offers, groups=None, reserved=()):
    j2o = np.full(max(1, jobs.n), -1, dtype=np.int32)
    fail = np.zeros(max(1, jobs.n), dtype=np.uint32)
    head = C.c_uint8(0)
    res = np.array(list(reserved) or [0], dtype=np.uint32)
    stats = np.zeros(64, dtype=np.uint64)
    js, os_ = jobs.as_struct(), offers.as_struct()
    gs = groups.as_struct() if groups is not None else None
    rc = lib().classfit_model_match(C.byref(params), C.byref(js), C.byref(os_), C.byref(gs) if gs is not None else None,
                                    res.ctypes.data_as(C.POINTER(C.c_uint32)), len(reserved), j2o.ctypes.data_as(C.POINTER(C.c_int32)),
                                    fail.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(head), stats.ctypes.data_as(C.POINTER(C.c_uint64)), 64)
    return rc, j2o[: jobs.n], fail[: jobs.n], bool(head.value), {n: int(stats[i]) for i, n in enumerate(NAMES)}


def ranked_jobs(params, pool, K=None):
    ranked, _ = pyoracle.rank(params, pool.tasks, pool.users)
    pend_ord = np.cumsum(pool.tasks.pending) - 1
    idx = pend_ord[ranked]
    if K is not None:
        idx = idx[:K]
    return pool.pending_jobs.take(idx)


def check(tag, params, jobs, offers, groups, reserved=(), verbose=True):
    t0 = time.time()
    want = pyoracle.match(params, jobs, offers, groups, reserved_hosts=reserved)
    t1 = time.time()
    rc, j2o, fail, head, st = model_match(params, jobs, offers, groups, reserved)
    t2 = time.time()
    if rc != 0:
        if verbose:
            print(f"{tag}: not eligible (falls back to the window rounds)")
        return None
    ok = np.array_equal(j2o, want[0]) and np.array_equal(fail, want[1]) and head == want[2]
    if not ok:
        bad = np.nonzero((j2o != want[0]) | (fail != want[1]))[0]
        print(f"{tag}: MISMATCH at {bad[:10]} model {j2o[bad[:10]]} / {fail[bad[:10]]} oracle {want[0][bad[:10]]} / {want[1][bad[:10]]}")
        return False
    if verbose:
        wm = max(1, st["matched"])
        wu = max(1, st["walked_unmatched"])
        print(f"{tag}: identical to the oracle ({jobs.n} jobs x {offers.n} offers; oracle {t1 - t0:.2f} s, model {t2 - t1:.2f} s)")
        print(f"   classes {st['classes']} chunks {st['chunks']} class waves {st['waves']} fixed point 2^-{st['kc']} / 2^-{st['km']}")
        print(f"   walked {st['walked']} (matched {st['matched']}, unmatched {st['walked_unmatched']}: {st['gpu_unmatched']} gpu jobs, {st['slow_unmatched']} with host / attribute / group constraints), settled by the batch pre-check {st['presettled']}")
        print(f"   matched: overlay lane wins {st['overlay_wins']}, offers opened {st['opens']} (dead at once {st['open_dead']}), gpu-class placements {st['gpu_places']}, "
              f"dead lanes dropped {st['dead_drops']}, epochs {st['epochs']}")
        print(f"   chunk scans {st['scans']} (empty {st['empty_scans']}, by jobs with host / attribute / group constraints {st['slow_scans']} for {st['slow_jobs']} such jobs), "
              f"tightenings {st['tightens']}")
        print(f"   critical path: scans per matched job {st['crit_scans_matched'] / wm:.3f}, per walked unmatched job {st['crit_scans_unmatched'] / wu:.3f}, worst {st['max_crit_scans']}; "
              f"wave-steps per matched job {st['crit_steps_matched'] / wm:.2f}, per walked unmatched {st['crit_steps_unmatched'] / wu:.2f}, "
              f"per unmatched job incl. pre-settled {(st['crit_steps_unmatched'] + st['presettled'] / 64.0) / max(1, st['walked_unmatched'] + st['presettled']):.3f}")
        print(f"   guard band: {st['band_events']} jobs with several offers inside 2^-37 ({st['literal_evals']} literal evaluations)")
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fuzz", type=int, default=1000)
    ap.add_argument("--big", action="store_true", help="C2, C3 and one C4 pool at full size")
    ap.add_argument("--seed", type=int, default=7)
    args = ap.parse_args()
    p = A.default_params(good_enough_fitness=1.0)
    ok = True
    if args.big:
        c2 = synth.make_pool(seed=0xC00C0002, n_pending=50_000, n_running=20_000, n_users=1000, n_offers=5000)
        ok &= check("C2 50k x 5k", p, ranked_jobs(p, c2), c2.offers, c2.groups) is True
        c3 = synth.make_pool(seed=0xC00C0003, n_pending=200_000, n_running=80_000, n_users=2000, n_offers=20_000, gpus=True, constraints=True)
        ok &= check("C3 200k x 20k", p, ranked_jobs(p, c3), c3.offers, c3.groups) is not False
        spec = workload.ClusterSpec()
        c4 = workload.make_pool(spec, 0)
        ok &= check("C4 pool 0 125k x 6250", p, ranked_jobs(p, c4), c4.offers, c4.groups) is True
        ok &= check("C4 pool 0, K = 1000", p, ranked_jobs(p, c4, 1000), c4.offers, c4.groups) is True
    rng = np.random.default_rng(args.seed)
    n_ok = n_skip = 0
    for it in range(args.fuzz):
        kw = dict(seed=int(rng.integers(1, 1 << 30)), n_pending=int(rng.integers(1, 600)), n_running=int(rng.integers(0, 150)),
                  n_users=int(rng.integers(1, 30)), n_offers=int(rng.integers(1, 400)), gpus=bool(rng.integers(0, 2)),
                  constraints=bool(rng.integers(0, 2)), fractional=bool(rng.integers(0, 4) == 0), tie_heavy=bool(rng.integers(0, 2)))
        pool = synth.make_pool(**kw)
        reserved = tuple(int(x) for x in rng.integers(0, kw["n_offers"], int(rng.integers(0, 3))))
        r = check(f"fuzz {it} {kw}", p, pool.pending_jobs, pool.offers, pool.groups, reserved, verbose=False)
        if r is False:
            ok = False
            break
        n_ok += r is True
        n_skip += r is None
    print(f"fuzz: {n_ok} configurations identical to the oracle, {n_skip} not eligible (fractional resources), seed {args.seed}")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

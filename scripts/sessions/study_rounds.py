#!/usr/bin/env python3
"""Rounds per match against the placement's compile-time shape (evaluated window, jobs per segment, merged list length) on the SIMT
emulator built with the shipped launch shapes: hardware-independent statistics (rounds, why they ended) for design decisions.
TEST INFRASTRUCTURE (builds emulator variants into /tmp); results in DESIGN.md §15.

  python scripts/study_rounds.py --scale 0.25 --variants 960:384:24 960:384:48 960:192:64 1920:384:48      (WEVAL:WSEG:LM)
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
EMU = os.path.join(ROOT, "tests", "simt_emu")


def build(weval, wseg, lm):
    out = f"/tmp/libcookmatch_emu_w{weval}_s{wseg}_lm{lm}.so"
    if not os.path.exists(out):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-I", EMU, "-DCOOK_EMU_SHIPPED_SHAPES",
                               f"-DCOOK_MV_WEVAL={weval}", f"-DCOOK_MV_WSEG={wseg}", f"-DCOOK_MV_LM={lm}", "-x", "c++",
                               os.path.join(ROOT, "cook_amd", "csrc", "engine.hip"), os.path.join(EMU, "emu.cpp"), "-o", out])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.25, help="fraction of a C4 pool (125k pending x 6250 offers)")
    ap.add_argument("--variants", nargs="+", default=["960:384:48"])
    a = ap.parse_args()
    from cook_amd import _abi as A
    from cook_amd import synth
    from cook_amd.engine import Engine
    n_pend, n_off = int(125000 * a.scale), int(6250 * a.scale)
    pool = synth.make_pool(seed=0xC00C0004, n_pending=n_pend, n_running=int(50000 * a.scale), n_users=max(10, int(10000 * a.scale)),
                           n_offers=n_off, gpus=True, constraints=True)
    p = A.default_params(good_enough_fitness=1.0)
    ref = None
    for v in a.variants:
        weval, wseg, lm = (int(x) for x in v.split(":"))
        so = build(weval, wseg, lm)
        t0 = time.time()
        with Engine(p, lib_path=so) as e:
            e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
            e.cycle_run(n_pend)
            _, j2o, _ = e.cycle_fetch()
            st = e.match_stats()
        if ref is None:
            ref = j2o
        same = bool((ref == j2o).all())
        print(f"WEVAL={weval} WSEG={wseg} LM={lm}: rounds {st['rounds']} (list {st['stop_list']} full {st['stop_full']} window {st['stop_window']} "
              f"segments {st['segments']}) visited {st['visited']} matched {st['matched']} same_result {same}  [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    main()

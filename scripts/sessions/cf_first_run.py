#!/usr/bin/env python3
"""Class-ordered best fit: the FIRST run after staging against the oracle, then a second one (a run that reads memory it has not written shows here:
the second run finds the first one's leftovers).  TEST TOOL (uses the oracle)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cook_amd import _abi as A, synth
from cook_amd.engine import Engine
from oracle import pyoracle
lib = sys.argv[1] if len(sys.argv) > 1 else ""
from cook_amd import build
so = build.build() if not lib else os.path.join(ROOT, "cook_amd", f"libcookmatch_{lib}.so")
p = A.default_params(good_enough_fitness=1.0, match_algo=3)
c2 = synth.make_pool(seed=0xC00C0002, n_pending=50_000, n_running=20_000, n_users=1000, n_offers=5000)
ranked, _ = pyoracle.rank(p, c2.tasks, c2.users)
jobs = c2.pending_jobs.take((np.cumsum(c2.tasks.pending) - 1)[ranked])
o = pyoracle.match(p, jobs, c2.offers, c2.groups, ())
if os.environ.get("CF_WARM"):  # other engines' matches first: their freed device buffers (and the LDS they left) are what this one starts on
    for sd in range(6):
        pl = synth.make_pool(seed=77 + sd, n_pending=300 + 200 * sd, n_running=100, n_users=20, n_offers=100 + 60 * sd, gpus=bool(sd % 2), constraints=bool(sd % 3 == 0))
        with Engine(p, lib_path=so) as e0:
            e0.match(pl.pending_jobs, pl.offers, pl.groups, ())
with Engine(p, lib_path=so) as e:
    e.match_stage(jobs, c2.offers, c2.groups, ())
    for rep in range(3):
        e.match_run()
        j2o, fail, head = e.match_fetch()
        st = e.match_stats()
        ok = np.array_equal(j2o, o[0]) and np.array_equal(fail, o[1])
        bad = np.nonzero(j2o != o[0])[0]
        print(f"run {rep}: {'identical' if ok else 'MISMATCH first at ' + str(bad[:5])}; epochs {st.get('cf_epochs')} exact {st.get('cf_exact_turns')} ticks {st.get('cf_ticks')}")

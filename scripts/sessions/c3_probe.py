"""BASELINE.json configs[1] / configs[2] (C2, C3) as single pools: cycle time and placement statistics (COOK_LIB picks a library build)."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from cook_amd import _abi as A, synth
from cook_amd.engine import Engine
for name, kw in (("C2", dict(seed=0xC00C0002, n_pending=50000, n_running=20000, n_users=1000, n_offers=5000)),
                 ("C3", dict(seed=0xC00C0003, n_pending=200000, n_running=80000, n_users=2000, n_offers=20000, gpus=True, constraints=True))):
    pool = synth.make_pool(**kw)
    with Engine(A.default_params(good_enough_fitness=float(os.environ.get('PROBE_GE', '1.0'))), device=0) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            e.cycle_run(pool.n_pending)
            ts.append((time.perf_counter() - t0) * 1e3)
        _, j2o, _ = e.cycle_fetch()
        st = e.match_stats()
        print(name, "ms", [round(t, 2) for t in ts], "matched", int((j2o >= 0).sum()), {k: st[k] for k in ("rounds", "stop_list", "stop_full", "stop_window", "segments", "setup_us", "seq_us")})

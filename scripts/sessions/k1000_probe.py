"""One pool of the benchmark cluster at K = 1000 (config.clj:113): placement statistics and the round log."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from cook_amd import _abi as A, workload
from cook_amd.engine import Engine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
pool = workload.make_pool(workload.ClusterSpec(), 0)
e = Engine(A.default_params(), device=0)
e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
for it in range(3):
    t0 = time.perf_counter()
    e.cycle_run(K)
    dt = (time.perf_counter() - t0) * 1e3
print("cycle ms", round(dt, 3), e.last_timing(), e.match_stats())
log = os.environ.get("COOK_ROUND_LOG")
if log and os.path.exists(log):
    print(open(log).read()[:3000])

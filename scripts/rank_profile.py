#!/usr/bin/env python3
"""Per-kernel times of ONE pool's rank + small-K cycle at the bench's pool shape (hipEvent pairs around every launch: the sums include
the events' own cost, so read them as shares, and the wall time without profiling as the total)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cook_amd import _abi as A  # noqa: E402
from cook_amd import workload  # noqa: E402
from cook_amd.engine import Engine  # noqa: E402

spec = workload.ClusterSpec(pools=8, pending=1_000_000, running=400_000, offers=50_000, users=10_000, constraints=True)
pool = workload.make_pool(spec, 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ALGO = int(sys.argv[2]) if len(sys.argv) > 2 else 0
GE = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0  # 1.0 = best fit (the timed configuration); the reference's default is 0.8
e = Engine(A.default_params(match_algo=ALGO, good_enough_fitness=GE))
e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
for _ in range(3):
    e.cycle_run(K)
ts = []
for _ in range(20 if K <= 5000 else 4):
    t0 = time.perf_counter()
    e.cycle_run(K)
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(f"cycle K={K} algo {ALGO} good-enough {GE}: p50 {ts[len(ts) // 2]:.3f} ms  stage {e.last_timing()}")
print("  placement", {k: v for k, v in e.match_stats().items() if v})
e.set_profiling(True)
n = 5
for _ in range(n):
    e.cycle_run(K)
kt = e.kernel_timings()
tot = sum(ms for ms, _ in kt.values())
print(f"profiled: {tot / n:.3f} ms of kernels per cycle, {sum(c for _, c in kt.values()) / n:.0f} launches")
for name, (ms, c) in sorted(kt.items(), key=lambda kv: -kv[1][0]):
    print(f"  {name:28s} {ms / n * 1e3:9.1f} us/cycle  {c / n:6.1f} launches  {ms / max(1, c) * 1e3:7.1f} us each")

"""One pool's cook_cycle_update alone: host wall time per call (incl. the stream sync at its end) and the kernels / copies it issues.
Usage: python scripts/update_probe.py [--lib path] [--small]"""
import argparse, sys, time
import numpy as np
sys.path.insert(0, ".")
from cook_amd import _abi as A, synth, workload
from cook_amd.engine import Engine, PinnedArena

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--small", action="store_true")
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
spec = workload.ClusterSpec()
if a.small:
    spec = workload.ClusterSpec(pools=8, pending=16000, running=8000, offers=800, users=200)
n_pend, n_run, n_off = spec.per_pool
pool = workload.make_pool(spec, 0)
n_delta = max(1, (n_pend + n_run) // 100)
extra = synth.make_pool(seed=0xD0000, n_pending=n_delta // 2, n_running=n_delta - n_delta // 2, n_users=spec.users, n_offers=n_off, gpus=True,
                        constraints=True, id_base=27_592_186_044_416)
aj = extra.pending_jobs
ng = pool.groups.n if pool.groups is not None else 0
if aj.group is not None:
    aj.group = np.where((aj.group != A.NONE_U32) & (ng > 0), aj.group % max(1, ng), A.NONE_U32).astype(np.uint32)
arena = PinnedArena(a.lib)
rem = arena.copy(np.sort(np.random.default_rng(7).choice(n_pend + n_run, size=n_delta, replace=False)).astype(np.uint32))
delta = (rem, arena.pin(extra.tasks), arena.pin(aj), arena.pin(pool.offers))
e = Engine(A.default_params(), device=0, lib_path=a.lib)
ts = []
for it in range(a.iters):
    e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
    e.cycle_run(1000)
    e.set_profiling(it == a.iters - 1)
    t0 = time.perf_counter()
    e.cycle_update(*delta)
    ts.append((time.perf_counter() - t0) * 1e3)
print("update ms per call:", [round(t, 3) for t in ts])
kt = e.kernel_timings()
tot = sum(v[0] if isinstance(v, (tuple, list)) else v for v in kt.values()) if kt else 0
print("profiled scopes of the last call:", len(kt), "sum ms", round(tot, 3))
for k, v in sorted(kt.items(), key=lambda kv: -(kv[1][0] if isinstance(kv[1], (tuple, list)) else kv[1]))[:12]:
    print("  ", k, v)

#!/usr/bin/env python3
"""The boundary leg of bench.py (eight pools: restage, cook_cycle_update of every pool through the cluster's thread pool, the cycle, the fetch)
many times over, to catch the occasional slow update (one 8.9 ms call among fifty of 1 ms: profiles/r05aa_bench.json) with the library's own
account of the call (cook_match_stats_ex [26..28]) and the host's (per-thread wall time of every engine's call).
usage: update_outliers8.py [iterations] [serial]     (serial: the updates one after the other in the main thread)"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from cook_amd import _abi as A, synth, workload, sharding
from cook_amd.engine import Engine, PinnedArena

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
serial = len(sys.argv) > 2 and sys.argv[2] == "serial"
spec = workload.ClusterSpec()
n_pend, n_run, n_off = spec.per_pool
pools = workload.make_pools(spec, range(8))
engines = {p: Engine(A.default_params(good_enough_fitness=1.0), device=0) for p in pools}
arena = PinnedArena()
rng = np.random.default_rng(7)
n_delta = (n_pend + n_run) // 100
deltas, pinned = {}, {}
for p in pools:
    ex = synth.make_pool(seed=0xD0000 + p, n_pending=n_delta // 2, n_running=n_delta - n_delta // 2, n_users=spec.users, n_offers=n_off, gpus=True,
                         constraints=True, id_base=27_592_186_044_416)
    aj, ng = ex.pending_jobs, pools[p].groups.n
    aj.group = np.where((aj.group != A.NONE_U32) & (ng > 0), aj.group % max(1, ng), A.NONE_U32).astype(np.uint32)
    deltas[p] = (arena.copy(np.sort(rng.choice(n_pend + n_run, size=n_delta, replace=False)).astype(np.uint32)), arena.pin(ex.tasks), arena.pin(aj),
                 arena.pin(pools[p].offers))
    pinned[p] = (arena.pin(pools[p].tasks), arena.pin(pools[p].pending_jobs), arena.pin(pools[p].offers))


def restage():
    for p in pools:
        engines[p].cycle_stage(pinned[p][0], pools[p].users, pinned[p][1], pinned[p][2], pools[p].groups)
    torch.cuda.synchronize()


restage()
cl = sharding.ShardedCluster(engines, workload.quota_groups(spec))
K = spec.per_pool[0]
cl.cycle(K); torch.cuda.synchronize()
cl.update(deltas); torch.cuda.synchronize()
restage(); cl.cycle(K); torch.cuda.synchronize()
wall = {}
orig = {p: engines[p].cycle_update for p in pools}
for p in pools:  # the host's own clock around every engine's call (inside its pool thread)
    def timed(*a, _p=p, **k):
        t0 = time.perf_counter()
        r = orig[_p](*a, **k)
        wall[_p] = (time.perf_counter() - t0) * 1e3
        return r
    engines[p].cycle_update = timed
gc.collect(); gc.disable()
ts = []
for it in range(iters):
    t0 = time.perf_counter()
    if serial:
        for p in pools:
            engines[p].cycle_update(*deltas[p])
    else:
        cl.update(deltas)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    ts.append(ms)
    if it > 5 and ms > 2.0 * float(np.median(ts)):
        st = {p: engines[p].match_stats() for p in pools}
        print(f"sample {it}: {ms:.2f} ms  host wall per pool {[round(wall[p], 2) for p in pools]}  library: in call {[st[p]['update_us'] for p in pools]} us, "
              f"in stream syncs {[st[p]['update_sync_us'] for p in pools]} us, device allocations {[st[p]['update_allocs'] for p in pools]}", flush=True)
    cl.cycle(K); torch.cuda.synchronize()
    restage()
ts = np.array(ts)
print(f"{'serial' if serial else 'thread pool'}: {iters} updates of eight pools: median {np.median(ts):.3f} ms, p99 {np.percentile(ts, 99):.3f}, max {ts.max():.3f}, over 2 x median: {(ts > 2 * np.median(ts)).sum()}")

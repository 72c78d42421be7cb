"""Looks for the occasional slow cook_cycle_update (10+ ms against 0.2): one engine, many (stage, cycle, update) rounds; variants of the delta."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cook_amd import _abi as A, synth, workload
from cook_amd.engine import Engine, PinnedArena
import gc
spec = workload.ClusterSpec()
n_pend, n_run, n_off = spec.per_pool
pool = workload.make_pool(spec, 0)
n_delta = (n_pend + n_run) // 100
extra = synth.make_pool(seed=0xD0000, n_pending=n_delta // 2, n_running=n_delta - n_delta // 2, n_users=spec.users, n_offers=n_off, gpus=True,
                        constraints=True, id_base=27_592_186_044_416)
aj = extra.pending_jobs
ng = pool.groups.n
aj.group = np.where((aj.group != A.NONE_U32) & (ng > 0), aj.group % max(1, ng), A.NONE_U32).astype(np.uint32)
arena = PinnedArena()
rem = arena.copy(np.sort(np.random.default_rng(7).choice(n_pend + n_run, size=n_delta, replace=False)).astype(np.uint32))
T, J, O = arena.pin(extra.tasks), arena.pin(aj), arena.pin(pool.offers)
PT, PJ, PO = arena.pin(pool.tasks), arena.pin(pool.pending_jobs), arena.pin(pool.offers)
e = Engine(A.default_params(), device=0)
gc.collect(); gc.disable()
for name, delta, restage in (("full delta after restage", (rem, T, J, O), True), ("offers only, no restage", (np.zeros(0, np.uint32), None, None, O), False),
                             ("rows only after restage", (rem, T, J, None), True)):
    e.cycle_stage(PT, pool.users, PJ, PO, pool.groups)
    e.cycle_run(1000)
    ts = []
    for it in range(40):
        if restage and it:
            e.cycle_stage(PT, pool.users, PJ, PO, pool.groups)
            e.cycle_run(1000)
        t0 = time.perf_counter()
        e.cycle_update(*delta)
        ts.append((time.perf_counter() - t0) * 1e3)
        e.cycle_run(1000)
    ts = np.array(ts)
    print(f"{name}: median {np.median(ts):.3f} ms, max {ts.max():.3f}, over 3 ms: {np.nonzero(ts > 3)[0].tolist()} {[round(x, 1) for x in ts[ts > 3]]}")

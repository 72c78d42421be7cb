"""Pools over GPUs: one process per GPU, pool p -> rank p mod world.

Every structure on the hot path is per pool (scheduler.clj:2167-2194 ranks pool by pool, :2488-2506 makes one handler
and one Fenzo per pool), so the pools of a cluster shard over ranks with NO data-path collective.  The one cross-pool
reduction is quota-group usage: `aggregate-quota-groups` (scheduler.clj:2125-2132) sums the running usage
{count, cpus, mem, gpus} of all pools mapped to the same quota group, and `filter-based-on-quota` (:2134-2157) then
filters each member pool's queue against the group quota.  Here every rank reduces the usage of its own pools on its
device, and ONE all-reduce(SUM) of a [n_groups x 4] f64 matrix (RCCL over xGMI on the GPU box, gloo in the CPU tests)
gives every rank the group totals.  The payload is 32 bytes per group: latency-bound, one per cycle.

The per-pool compute is behind the tiny `PoolEngine` protocol so that the same code drives `cook_amd.engine.Engine`
(bench.py, GPU) and a checker in the tests.
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Protocol, Sequence

import numpy as np

from . import _abi as A


def pools_of_rank(n_pools: int, world: int, rank: int) -> List[int]:
    """Pool p lives on rank p mod world (one pool per GPU when world == n_pools: BASELINE.json configs[3])."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return [p for p in range(n_pools) if p % world == rank]


class PoolEngine(Protocol):
    def rank_pool_usage(self) -> A.CookUsage: ...
    def rank_set_quota(self, quota: Optional[A.CookPoolQuota]) -> None: ...
    def cycle_run(self, num_considerable: int) -> None: ...


@dataclass
class QuotaGroups:
    """quota-grouping config: pool -> group name, group -> quota (scheduler.clj:2125-2157; config `quota-grouping`)."""
    pool_group: Dict[int, int] = field(default_factory=dict)      # pool id -> dense group id; absent = no group
    group_quota: Dict[int, A.CookUsage] = field(default_factory=dict)
    pool_quota: Dict[int, A.CookUsage] = field(default_factory=dict)  # tools/global-pool-quota; absent = nil

    @property
    def n_groups(self) -> int:
        return (max(self.pool_group.values()) + 1) if self.pool_group else 0


def group_usage_matrix(groups: QuotaGroups, local_usage: Dict[int, Sequence[float]]) -> np.ndarray:
    """This rank's contribution: [n_groups, 4] f64, row g = sum of the running usage of the local pools in group g."""
    m = np.zeros((max(1, groups.n_groups), 4), dtype=np.float64)
    for p, u in sorted(local_usage.items()):
        g = groups.pool_group.get(p)
        if g is not None:
            m[g] += np.asarray(u, dtype=np.float64)
    return m


def all_reduce_group_usage(local: np.ndarray, world: int, device=None, force: bool = False) -> np.ndarray:
    """The collective the reference's path needs: all-reduce(SUM) of the [n_groups, 4] matrix.  Integer-valued usages (count,
    and the benchmark's cpus/mem/gpus) sum exactly in any order; the reference sums pools in map order (merge-with +).
    `force` runs the collective even at world 1 (the single-rank RCCL test on the GPU box)."""
    if world <= 1 and not force:
        return local
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def all_reduce_user_usage(engines: Sequence, n_users: int, world: int, device=None, force: bool = False, to_host: bool = True):
    """BASELINE.json north_star's collective: the cross-pool per-user usage totals.  Every local pool writes its [U, 3] vector
    {cpus, mem, gpus of the user's running tasks} (cook_rank_user_usage) — on a GPU rank straight into a device tensor, so the
    payload never visits the host —, the rank adds its pools, and ONE all-reduce(SUM) over RCCL / gloo gives every rank the
    totals (10k users = 240 KB: latency-bound on xGMI).  Dividing by the users' shares gives the cross-pool DRU totals."""
    import torch
    import torch.distributed as dist

    on_gpu = device is not None and getattr(device, "type", "cpu") == "cuda"
    engines = list(engines)
    if on_gpu:
        # one slice per engine: every engine writes its own rows on its own stream (and synchronises that stream before returning),
        # torch reads them only afterwards — no buffer is reused while another stream may still be reading it
        parts = torch.empty((max(1, len(engines)), n_users, 3), dtype=torch.float64, device=device)
        for i, e in enumerate(engines):
            e.rank_user_usage(n_users, device_ptr=parts[i].data_ptr())
        acc = parts[:len(engines)].sum(dim=0) if engines else torch.zeros((n_users, 3), dtype=torch.float64, device=device)
    else:
        acc = torch.zeros((n_users, 3), dtype=torch.float64)
        for e in engines:
            acc += torch.from_numpy(np.ascontiguousarray(e.rank_user_usage(n_users)))
    if world > 1 or force:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return acc.cpu().numpy() if to_host else acc  # to_host=False: the totals stay where the collective left them (no sync, no copy)


def reduce_user_usage_parts(parts, pools, n_users: int, world: int, on_gpu: bool):
    """The tail of all_reduce_user_usage when the pools' [U, 3] vectors have been extracted already (ShardedCluster.cycle does that
    in the pools' threads): sum the local pools, ONE all-reduce(SUM); the totals stay where the collective left them."""
    import torch
    import torch.distributed as dist

    if on_gpu:
        acc = parts.sum(dim=0)
    else:
        acc = torch.zeros((n_users, 3), dtype=torch.float64)
        for p in pools:
            acc += torch.from_numpy(np.ascontiguousarray(parts[p]))
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return acc


class _SerialExecutor:
    """map / shutdown of a ThreadPoolExecutor, on the calling thread"""

    @staticmethod
    def map(fn, xs):
        return [fn(x) for x in xs]

    @staticmethod
    def shutdown(wait=True):
        return None


class ShardedCluster:
    """The pools of one cluster that live on this rank, and one match cycle over them.

    cycle(K): 1. per-pool running usage (device reduction)  2. all-reduce into quota-group usage
              3. per pool: set quota inputs; rank + take K of all local pools in ONE cook_cycle_run_rank_multi call (the pools' flows
                 side by side on one stream, the same kernel of several pools in one launch; COOK_RANK_BATCH=0: pools concurrently, one
                 stream each), then the placements of all
                 local pools in ONE cook_cycle_match_multi call (served walkers: a persistent walker workgroup per pool beside
                 serve launches); with COOK_MATCH_SERVED=0 as lockstep chains of launches, a single pool through cook_cycle_run.
    """

    def __init__(self, engines: Dict[int, PoolEngine], groups: QuotaGroups, world: int = 1, rank: int = 0, device=None, serial: bool = False):
        """serial: the pools of this rank take turns on the calling thread instead of running on one thread each (engines that
        cannot be called concurrently: the single-process SIMT emulator of the tests)."""
        self.engines = dict(engines)
        self.pools = sorted(self.engines)
        self.groups = groups
        self.world, self.rank, self.device = world, rank, device
        self.max_chains = int(os.environ.get("COOK_MAX_CHAINS", "4"))
        if serial:
            self._tp = self._tp_rank = _SerialExecutor()
        else:
            self._tp = ThreadPoolExecutor(max_workers=max(1, len(self.pools)))
            self._tp_rank = ThreadPoolExecutor(max_workers=max(1, min(len(self.pools), int(os.environ.get("COOK_MAX_RANK_CHAINS", str(self.max_chains))))))
        self.last_group_usage: Optional[np.ndarray] = None
        self.last_pool_usage: Dict[int, Sequence[float]] = {}
        self.n_users = 0                      # > 0: every cycle also all-reduces the cross-pool per-user usage [U, 3]
        self._last_user_usage = None          # torch tensor on the collective's device (or numpy); see last_user_usage
        self._user_parts = [None, None]       # two [pools, U, 3] device tensors the pools write their usage vectors into, used in turn:
        self._user_parts_turn = 0             # the sum / all-reduce of cycle c (torch's stream) may still read one while the engines'
                                              # own streams fill the other in cycle c + 1
        self.last_phase_ms = (0.0, 0.0, 0.0, 0.0)
        self.chain_whole_cycle = os.environ.get("COOK_CHAIN_WHOLE_CYCLE", "0") != "0"
        self.force_multi = os.environ.get("COOK_FORCE_MULTI", "0") != "0"  # every pool through the multi-pool launch path, one per chain (measurement)
        # served walkers (match_v2.hpp): ALL pools of the rank in one cook_cycle_match_multi call — one persistent walker workgroup per
        # pool beside serve launches, two streams per GPU whatever the number of pools.  COOK_MATCH_SERVED=0: lockstep chains as before.
        self.served = os.environ.get("COOK_MATCH_SERVED", "1") != "0"
        self._usage_warm = False  # the engines have summed their pools' usage for the tables they hold (see cycle)
        # the rank parts of all local pools in ONE cook_cycle_run_rank_multi call (one thread, one stream, the same kernel of several pools
        # in one launch) instead of a thread per pool: the stage is bound by the number of launches the host makes
        self.rank_batch = os.environ.get("COOK_RANK_BATCH", "1") != "0"
        # ... in up to four batches of at least two pools, each from a thread of its own (measured on MI355X, eight pools: 1 / 2 / 4 batches =
        # 2.6 / 2.3 / 2.2 ms of the cycle's rank phase against 3.3 for a thread and a stream per pool: the batched launches are bound by the GPU,
        # and two or four sequences fill what one leaves idle between dependent kernels — profiles/r05rd_rank_batch.txt)
        self.rank_batches = int(os.environ.get("COOK_RANK_BATCHES", "4"))

    @property
    def last_user_usage(self) -> Optional[np.ndarray]:
        """cross-pool per-user usage [U, 3] of the last cycle (copied to the host on demand: the cycle leaves it on the device)"""
        u = self._last_user_usage
        if u is None or isinstance(u, np.ndarray):
            return u
        return u.cpu().numpy()

    def close(self):
        self._tp.shutdown(wait=True)
        self._tp_rank.shutdown(wait=True)

    def quota_inputs(self, pool: int, pool_usage: Sequence[float], group_usage: np.ndarray) -> Optional[A.CookPoolQuota]:
        pq = self.groups.pool_quota.get(pool)
        g = self.groups.pool_group.get(pool)
        gq = self.groups.group_quota.get(g) if g is not None else None
        # pool-name->usage only holds pools WITH running tasks, so a group none of whose pools runs anything has no usage
        # entry and filter-based-on-quota skips the group filter: (and quota-group-quota quota-group-usage),
        # scheduler.clj:2134-2157.  The reduced task count tells the two cases apart.
        if gq is not None and not (group_usage[g][0] > 0):
            gq = None
        if pq is None and gq is None:
            return None
        return A.pool_quota(pool_quota=pq, group_quota=gq,
                            group_usage=A.usage(*group_usage[g].tolist()) if gq is not None else None,
                            pool_usage=A.usage(*pool_usage))

    def update(self, deltas: Dict[int, tuple]):
        """cook_cycle_update for the local pools, deltas[pool] = the arguments of Engine.cycle_update.  The update is a short chain of
        small kernels per pool: at most max_chains of them at a time, like the rank stages (eight at once measured SLOWER than one after
        the other on MI355X, 5.6 against 5.1 ms for eight pools: more than four streams of small kernels serialise)."""
        self._usage_warm = False
        list(self._tp_rank.map(lambda p: self.engines[p].cycle_update(*deltas[p]), [p for p in self.pools if p in deltas]))

    def cycle(self, num_considerable: int):
        t0 = time.perf_counter()
        # (the library keeps a pool's usage until its task table changes: after the first cycle on a table the calls return at once, and
        #  handing them to the thread pool would cost more than making them — 0.25 ms of the benchmark's cycle)
        native = len(self.pools) > 1 and self.rank_batch and all(hasattr(self.engines[p], "_h") for p in self.pools)
        if native:  # the pools' sums side by side in ONE call (cook_rank_pool_usage_multi: one launch per kernel, one synchronisation)
            from .engine import rank_pool_usage_multi
            usages = dict(zip(self.pools, rank_pool_usage_multi([self.engines[p] for p in self.pools])))
        elif self._usage_warm:
            usages = {p: self.engines[p].rank_pool_usage().as_tuple() for p in self.pools}
        else:
            usages = dict(zip(self.pools, self._tp.map(lambda p: self.engines[p].rank_pool_usage().as_tuple(), self.pools)))
            self._usage_warm = True
        total = all_reduce_group_usage(group_usage_matrix(self.groups, usages), self.world, self.device)
        self.last_group_usage = total
        self.last_pool_usage = usages

        t1 = time.perf_counter()
        multi = all(hasattr(self.engines[p], "cycle_run_rank") for p in self.pools)
        # (measured on MI355X, profiles/r05q_pools_per_gpu.txt: 8 pools 52.9 ms served against 59.8 in lockstep pairs; 4 pools 45.9 against 44.9
        #  as four chains of one pool each, 2 and 1 pools the same either way — up to max_chains pools every pool has a chain to itself)
        served = multi and self.served and self.max_chains < len(self.pools) <= 16
        lockstep = multi and (len(self.pools) > self.max_chains or self.force_multi or served)

        # the per-user usage vectors of the local pools (north_star's collective payload) are extracted by the pools' own threads right
        # after their rank stage — in parallel, overlapped with the other pools' work — so that the end of the cycle only sums and reduces
        want_users = bool(self.n_users) and all(hasattr(self.engines[p], "rank_user_usage") for p in self.pools)
        on_gpu = self.device is not None and getattr(self.device, "type", "cpu") == "cuda"
        user_parts = None
        if want_users:
            if on_gpu:
                import torch
                turn = self._user_parts_turn = self._user_parts_turn ^ 1
                if self._user_parts[turn] is None or tuple(self._user_parts[turn].shape) != (len(self.pools), self.n_users, 3):
                    self._user_parts[turn] = torch.empty((len(self.pools), self.n_users, 3), dtype=torch.float64, device=self.device)
                user_parts = self._user_parts[turn]
            else:
                user_parts = {}

        def run(p):
            self.engines[p].rank_set_quota(self.quota_inputs(p, usages[p], total))
            if lockstep:
                self.engines[p].cycle_run_rank(num_considerable)  # rank part per pool, in parallel
            else:
                self.engines[p].cycle_run(num_considerable)
            if want_users:
                if on_gpu:
                    self.engines[p].rank_user_usage(self.n_users, device_ptr=user_parts[self.pools.index(p)].data_ptr())
                else:
                    user_parts[p] = self.engines[p].rank_user_usage(self.n_users)

        batch_rank = lockstep and self.rank_batch and len(self.pools) > 1 and all(hasattr(self.engines[p], "_h") for p in self.pools)

        def rank_all():
            nonlocal user_parts
            if not batch_rank:
                list(self._tp_rank.map(run, self.pools))  # (the rank stages are chains of small kernels: at most max_chains at a time)
                return
            from .engine import cycle_run_rank_multi
            for p in self.pools:
                self.engines[p].rank_set_quota(self.quota_inputs(p, usages[p], total))
            # COOK_RANK_BATCHES > 1: the pools in that many batches, each from a thread of its own (every batch has its own stream): two
            # sequences of launches can overlap where one leaves the GPU idle between dependent kernels (measured: DESIGN.md 3a)
            nb = max(1, min(self.rank_batches, len(self.pools) // 2)) if hasattr(self._tp_rank, "submit") else 1  # (a serial executor only maps)
            host_parts = {}

            def one(b):
                idx = list(range(b, len(self.pools), nb))
                engs = [self.engines[self.pools[i]] for i in idx]
                if want_users and on_gpu:
                    cycle_run_rank_multi(engs, num_considerable, user_usage_ptrs=[user_parts[i].data_ptr() for i in idx])
                elif want_users:
                    for i, u in zip(idx, cycle_run_rank_multi(engs, num_considerable, n_users=self.n_users)):
                        host_parts[self.pools[i]] = u
                else:
                    cycle_run_rank_multi(engs, num_considerable)

            if nb == 1:
                one(0)
            else:  # (the calling thread takes a batch itself: handing one to the pool costs 30-45 us before its first launch, per-dispatch trace)
                futs = [self._tp_rank.submit(one, b) for b in range(1, nb)]
                try:
                    one(0)
                finally:  # (no batch may still be running on its engines when an error leaves this call: the caller closes them)
                    import concurrent.futures
                    concurrent.futures.wait(futs)
                for f in futs:
                    f.result()
            if want_users and not on_gpu:
                user_parts = host_parts

        n_chains = max(1, min(len(self.pools), self.max_chains))
        if served:
            from .engine import cycle_match_multi
            rank_all()
            t2 = time.perf_counter()
            cycle_match_multi([self.engines[p] for p in self.pools])  # (falls back to lockstep launches inside the library if it must)
            lead_e = self.engines[self.pools[0]]
            if hasattr(lead_e, "match_stats") and lead_e.match_stats().get("served_fell_back"):
                # correct, but a latency spike of the walkers' time-out (250 ms): something stalled the serve launches — another engine of the device
                # allocating or freeing, a descheduled host thread
                self.served_fell_back_cycles = getattr(self, "served_fell_back_cycles", 0) + 1
                import warnings
                warnings.warn("cook_cycle_match_multi: the served walkers gave up and the cycle was redone in lockstep launches", RuntimeWarning)
        elif lockstep and self.chain_whole_cycle:
            # MI355X runs about four independent chains of small kernels at full speed (beyond that the hardware queues
            # share dispatch pipes: 4 pools 113 ms, 6 or 8 pools 186 ms per cycle), while pools in lockstep pay for the
            # slowest pool of every round (8 in lockstep: 215 ms).  So: at most MAX_CHAINS chains, pools spread over them;
            # a chain ranks its pools one after the other and goes straight on to their placement rounds.
            from .engine import cycle_match_multi

            def chain(c):
                mine = self.pools[c::n_chains]
                for p in mine:
                    run(p)
                cycle_match_multi([self.engines[p] for p in mine])

            list(self._tp.map(chain, range(n_chains)))
            t2 = time.perf_counter()
        else:
            rank_all()
            t2 = time.perf_counter()
            if lockstep:
                from .engine import cycle_match_multi
                groups = [[self.engines[p] for p in self.pools[c::n_chains]] for c in range(n_chains)]
                list(self._tp.map(cycle_match_multi, groups))
        t3 = time.perf_counter()
        if want_users:
            self._last_user_usage = reduce_user_usage_parts(user_parts, self.pools, self.n_users, self.world, on_gpu)
        # host wall time of the phases: pool usage + all-reduce, rank (+ the whole cycle of pools that run on their own chain),
        # lockstep placement, per-user usage all-reduce
        self.last_phase_ms = tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, time.perf_counter() - t3))

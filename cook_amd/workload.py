"""The benchmark workload (BASELINE.json configs[3]) as ONE definition shared by bench.py and the parity tests, so that the
configuration that is timed is the configuration that is checked ("What's weak" 1).

8 pools x (125k pending + 50k running tasks, 6 250 offers), 10k users, gpu dimension + EQUALS / novel-host / unique-group
constraints; every pool has a (non-binding) pool quota and all pools belong to ONE quota group whose usage is the sum over the
pools of the cluster (scheduler.clj:2125-2157) -> the one cross-rank all-reduce of the path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Sequence

from . import _abi as A
from . import sharding, synth

SEED_BASE = 0xC00C0004


@dataclass
class ClusterSpec:
    pools: int = 8
    pending: int = 1_000_000   # over all pools
    running: int = 400_000
    offers: int = 50_000
    users: int = 10_000
    constraints: bool = True

    @property
    def per_pool(self):
        return self.pending // self.pools, self.running // self.pools, self.offers // self.pools


def make_pool(spec: ClusterSpec, p: int) -> synth.Pool:
    n_pend, n_run, n_off = spec.per_pool
    return synth.make_pool(seed=SEED_BASE + p, n_pending=n_pend, n_running=n_run, n_users=spec.users, n_offers=n_off,
                           gpus=spec.constraints, constraints=spec.constraints)


def make_pools(spec: ClusterSpec, which: Sequence[int]) -> Dict[int, synth.Pool]:
    return {p: make_pool(spec, p) for p in which}


def quota_groups(spec: ClusterSpec) -> sharding.QuotaGroups:
    P = spec.pools
    return sharding.QuotaGroups(pool_group={p: 0 for p in range(P)},
                                group_quota={0: A.quota(count=80_000_000, cpus=1e10, mem=1e14, gpus=1e9)},
                                pool_quota={p: A.quota(count=10_000_000, cpus=1e9, mem=1e13, gpus=1e8) for p in range(P)})

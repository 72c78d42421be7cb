"""TEST INFRASTRUCTURE: whole-pool parity check of one match cycle against the oracle, shared by tests/ and by bench.py's
check leg (which runs after the timed region; the product path never imports this package)."""
from __future__ import annotations

import numpy as np

from . import pyoracle


def check_pool_against_oracle(params, pool, quota, ranked, j2o, k: int, threads: int = 16):
    """TEST / BENCH-CHECK ONLY (imports the oracle): rank order and every assignment of one pool's cycle, bit-exact.
    Returns the oracle's (ranked, j2o) so that a caller can reuse them (bench.py's cpu_baseline leg times the same calls)."""
    o_ranked, _ = pyoracle.rank(params, pool.tasks, pool.users, quota=quota)
    if not np.array_equal(ranked, o_ranked):
        bad = np.nonzero(ranked[: len(o_ranked)] != o_ranked[: len(ranked)])[0]
        raise AssertionError(f"rank order differs from the oracle (lengths {len(ranked)} / {len(o_ranked)}, first at {bad[:3]})")
    kk = min(k, len(o_ranked))
    pend_ord = np.cumsum(pool.tasks.pending) - 1
    o_j2o, _, _ = pyoracle.match(params, pool.pending_jobs.take(pend_ord[o_ranked[:kk]]), pool.offers, pool.groups,
                                 nthreads=threads if params.good_enough_fitness >= 1.0 else 1)
    if not np.array_equal(j2o, o_j2o):
        bad = np.nonzero(j2o[: len(o_j2o)] != o_j2o[: len(j2o)])[0]
        raise AssertionError(f"assignments differ from the oracle (lengths {len(j2o)} / {len(o_j2o)}, first at rank position {bad[:3]})")
    return o_ranked, o_j2o

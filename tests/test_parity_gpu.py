"""Parity tests proper: the HIP engine (cook_amd/libcookmatch.so, gfx950) behind the C ABI vs the CPU oracle and the
reference's golden vectors.  Run on the GPU box with `pytest -m gpu`.  No CPU fallback: if the extension or the GPU
is missing these tests FAIL."""
import os

import numpy as np
import pytest

from cook_amd import _abi as A
from cook_amd import synth
from cook_amd.engine import Engine
from oracle import pyoracle
from tests import golden_util as _G
from tests import parity_cases as P

G_EXPLAIN = _G.load("explain")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def make_engine():
    from cook_amd import build
    so = build.build()  # no-op when the in-tree .so is current (hipcc is present on the GPU box too)
    return lambda params: Engine(params, lib_path=so)


def test_engine_is_hip_build(make_engine):
    with make_engine(A.default_params()) as e:
        assert "hip gfx950" in e.version


def test_rank_golden(make_engine):
    P.check_rank_golden(make_engine)


def test_rank_group_golden(make_engine):
    P.check_rank_group_golden(make_engine)


def test_match_golden(make_engine):
    P.check_match_golden(make_engine)


@pytest.mark.parametrize("kw", [
    dict(seed=1, n_pending=700, n_running=300, n_users=40, n_offers=50),
    dict(seed=2, n_pending=50000, n_running=20000, n_users=1000, n_offers=50),          # BASELINE config C2 rank shape
    dict(seed=3, n_pending=20000, n_running=5000, n_users=200, n_offers=50, tie_heavy=True),
    dict(seed=4, n_pending=20000, n_running=5000, n_users=200, n_offers=50, fractional=True),
    dict(seed=5, n_pending=20000, n_running=5000, n_users=200, n_offers=50, no_shares=True),
    dict(seed=6, n_pending=6000, n_running=0, n_users=7, n_offers=10, tie_heavy=True, quota_frac=0.5),
    dict(seed=7, n_pending=125000, n_running=50000, n_users=10000, n_offers=50),        # one C4 pool
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items() if k != "n_offers"))
def test_rank_parity_random(make_engine, kw):
    pool = synth.make_pool(**kw)
    P.rank_parity(make_engine, pool, A.default_params(max_over_quota_jobs=10))


def test_rank_parity_gpu_mode_and_quota(make_engine):
    pool = synth.make_pool(seed=11, n_pending=8000, n_running=4000, n_users=300, n_offers=20, gpus=True)
    pool.tasks.gpus[:] = np.maximum(pool.tasks.gpus, 1.0)
    P.rank_parity(make_engine, pool, A.default_params(dru_mode=1))
    q = A.pool_quota(pool_quota=A.quota(count=6000, cpus=25000.0), group_quota=A.quota(mem=4.0e7),
                     group_usage=A.usage(count=10, cpus=100, mem=1.0e6))
    P.rank_parity(make_engine, pool, A.default_params(offensive_max_mem_mb=16000.0, offensive_max_cpus=6.0), quota=q)


def test_rank_edge_cases(make_engine):
    p = A.default_params()
    empty = A.Tasks(cpus=np.zeros(0), mem=np.zeros(0), user=np.zeros(0), priority=np.zeros(0), start_ms=np.zeros(0),
                    task_id=np.zeros(0), job_id=np.zeros(0), pending=np.zeros(0))
    users = A.Users(div_cpus=np.ones(1), div_mem=np.ones(1))
    with make_engine(p) as e:
        ranked, _ = e.rank(empty, users)
    assert len(ranked) == 0
    pool = synth.make_pool(seed=3, n_pending=0, n_running=50, n_users=5, n_offers=4)
    assert len(P.rank_parity(make_engine, pool, p)) == 0
    pool = synth.make_pool(seed=3, n_pending=1, n_running=0, n_users=1, n_offers=4)
    assert len(P.rank_parity(make_engine, pool, p)) == 1
    pool = synth.make_pool(seed=9, n_pending=3000, n_running=500, n_users=3, n_offers=4, quota_frac=1.0)
    P.rank_parity(make_engine, pool, A.default_params(max_over_quota_jobs=0))


ALGOS = pytest.mark.parametrize("algo", [0, 1, 3], ids=["default", "serial", "classfit"])  # window rounds (shipped) / the one-job-at-a-time sweep / class-ordered best fit


@ALGOS
@pytest.mark.parametrize("ge", [1.0, 0.8, 0.5])
def test_match_parity_c2(make_engine, ge, algo):
    # BASELINE config C2: cpus+mem only, 5k offers; K = first 3000 pending jobs in input order
    pool = synth.make_pool(seed=21, n_pending=3000, n_running=100, n_users=200, n_offers=5000)
    P.match_parity(make_engine, pool.pending_jobs, pool.offers, None, A.default_params(good_enough_fitness=ge, match_algo=algo))


@ALGOS
def test_match_parity_c3_constraints(make_engine, algo):
    # BASELINE config C3 shape: host/attribute constraints + gpu dimension + unique groups (scaled to oracle-seconds)
    pool = synth.make_pool(seed=22, n_pending=4000, n_running=1000, n_users=200, n_offers=2000, gpus=True, constraints=True)
    j2o = P.match_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups,
                         A.default_params(good_enough_fitness=1.0, match_algo=algo), reserved=(3, 7, 150))
    assert (j2o >= 0).sum() > 500


@ALGOS
def test_match_fills_cluster_then_fails(make_engine, algo):
    # more demand than capacity: the tail of the queue must fail exactly like the oracle (fail codes included)
    pool = synth.make_pool(seed=23, n_pending=6000, n_running=0, n_users=50, n_offers=200)
    p = A.default_params(good_enough_fitness=1.0, match_algo=algo)
    j2o = P.match_parity(make_engine, pool.pending_jobs, pool.offers, None, p)
    assert (j2o < 0).sum() > 1000


def test_match_long_windows(make_engine, algo=0):
    # a cluster that is full after a few hundred jobs: from then on nearly every job is settled in the parallel phase of the resolve
    # kernel, the window grows past the LDS-staged size (MV_WLONG) and only the few jobs that still need the walk are staged —
    # gpu jobs, constrained jobs and group members keep some of those in every window
    pool = synth.make_pool(seed=29, n_pending=60000, n_running=0, n_users=300, n_offers=1500, gpus=True, constraints=True)
    pool.offers.cpus[:] = np.minimum(pool.offers.cpus, 12.0)  # small hosts: the cluster is full after a few thousand jobs
    pool.offers.mem[:] = np.minimum(pool.offers.mem, 40000.0)
    p = A.default_params(good_enough_fitness=1.0, match_algo=algo)
    P.match_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, p)
    with make_engine(p) as e:
        e.match(pool.pending_jobs, pool.offers, pool.groups)
        long_rounds = e.match_stats()["rounds"]
    os.environ["COOK_WLONG"] = "0"
    try:
        with make_engine(p) as e:
            e.match(pool.pending_jobs, pool.offers, pool.groups)
            short_rounds = e.match_stats()["rounds"]
    finally:
        del os.environ["COOK_WLONG"]
    assert long_rounds * 5 < short_rounds * 4, (long_rounds, short_rounds)


@ALGOS
def test_match_group_types(make_engine, algo):
    rng = np.random.default_rng(5)
    n, m = 1200, 300
    attr = np.zeros((m, 2), dtype=np.uint32)
    attr[:, 0] = rng.integers(1, 4, m)
    attr[:, 1] = rng.integers(0, 3, m)
    offers = A.Offers(cpus=np.full(m, 8.0), mem=np.full(m, 16000.0), attr=attr, k8s=np.ones(m, dtype=np.uint8))
    group = rng.integers(0, 6, n).astype(np.uint32)
    group[rng.random(n) < 0.6] = A.NONE_U32
    jobs = A.Jobs(cpus=rng.integers(1, 4, n).astype(float), mem=rng.integers(1, 4, n) * 1000.0, group=group)
    groups = A.Groups(type=np.array([1, 2, 2, 3, 3, 0], dtype=np.uint8),
                      attr_key=np.array([A.NONE_U32, 0, A.NONE_U32, 1, 0, 0], dtype=np.uint32),
                      minimum=np.array([0, 3, 10, 0, 0, 0], dtype=np.int32),
                      run_hosts=[[1, 2], [3], [], [], [5, 6], []],
                      run_attrs=[[0, 0], [int(attr[3, 0])], [], [], [int(attr[5, 0]), int(attr[6, 0])], []])
    P.match_parity(make_engine, jobs, offers, groups, A.default_params(good_enough_fitness=1.0, match_algo=algo))


@ALGOS
def test_match_constraints_beyond_the_fast_paths(make_engine, algo):
    jobs, offers, groups = P.slow_constraint_case(9, 3000, 800)
    j2o = P.match_parity(make_engine, jobs, offers, groups, A.default_params(good_enough_fitness=1.0, match_algo=algo))
    assert (j2o >= 0).sum() > 300


def test_match_slot_table_and_touched_set_limits(make_engine):
    p = A.default_params(good_enough_fitness=1.0)
    jobs, offers = P.pinned_jobs_case(7, 3000, 6000, 64)
    P.match_parity(make_engine, jobs, offers, None, p)
    with make_engine(p) as e:
        e.match(jobs, offers)
        st_ = e.match_stats()
        assert st_["segments"] >= st_["rounds"] > 0, st_
        # ... and touches more offers per round than the walk has lanes: lanes of dead offers (full to the smallest job) are given away
        assert st_["touched"] > 64 * st_["rounds"] or st_["rounds"] > 2, st_
    jobs, offers = P.pinned_jobs_case(8, 3000, 4000, 0)
    P.match_parity(make_engine, jobs, offers, None, p)
    with make_engine(p) as e:
        e.match(jobs, offers)
        assert e.match_stats()["stop_full"] > 0


def test_cycle_parity(make_engine):
    pool = synth.make_pool(seed=31, n_pending=20000, n_running=8000, n_users=500, n_offers=2000, gpus=True, constraints=True)
    P.cycle_parity(make_engine, pool, A.default_params(good_enough_fitness=1.0), k=1000)


def test_size_independent_properties_full_pool(make_engine):
    """One full C4 pool (125k pending x 6250 offers, K=2000): properties that need no oracle at full size."""
    pool = synth.make_pool(seed=41, n_pending=125000, n_running=50000, n_users=10000, n_offers=6250)
    p = A.default_params(good_enough_fitness=1.0)
    with make_engine(p) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e.cycle_run(2000)
        ranked, j2o, head = e.cycle_fetch()
        _, dru = e.rank_fetch(want_dru=True)
    # ranked is a permutation of (a subset of) the pending tasks, DRU non-decreasing along it
    assert len(set(ranked.tolist())) == len(ranked) and pool.tasks.pending[ranked].all()
    assert np.all(np.diff(dru[ranked]) >= 0)
    # no offer is over-committed
    pend_ord = np.cumsum(pool.tasks.pending) - 1
    jobs = pend_ord[ranked[: len(j2o)]]
    used_c = np.bincount(j2o[j2o >= 0], weights=pool.pending_jobs.cpus[jobs][j2o >= 0], minlength=pool.offers.n)
    used_m = np.bincount(j2o[j2o >= 0], weights=pool.pending_jobs.mem[jobs][j2o >= 0], minlength=pool.offers.n)
    assert np.all(used_c <= pool.offers.cpus) and np.all(used_m <= pool.offers.mem)
    # idempotence: the same staged inputs give the same answer again
    with make_engine(p) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e.cycle_run(2000)
        ranked2, j2o2, _ = e.cycle_fetch()
    assert np.array_equal(ranked, ranked2) and np.array_equal(j2o, j2o2)


def test_rebalance_golden(make_engine):
    P.check_rebalance_golden(make_engine)


@pytest.mark.parametrize("kw", [
    dict(seed=51, n_running=4000, n_pending=64, n_users=40, n_hosts=300),
    dict(seed=52, n_running=4000, n_pending=64, n_users=40, n_hosts=300, fractional=True),
    dict(seed=53, n_running=10240, n_pending=128, n_users=50, n_hosts=16, max_preemption=48),    # reference stress shape (:1152-1187)
    dict(seed=60, n_running=8000, n_pending=60, n_users=40, n_hosts=80, max_preemption=32, spare_frac=0.0),    # 65..128 items per host: lists in LDS
    dict(seed=54, n_running=5000, n_pending=128, n_users=60, n_hosts=400, constraints=True, gpus=True, max_preemption=128),
    dict(seed=55, n_running=3000, n_pending=40, n_users=20, n_hosts=250, dru_mode=1),
    dict(seed=56, n_running=0, n_pending=10, n_users=3, n_hosts=8, spare_frac=1.0),
    dict(seed=707730441, n_running=2, n_pending=29, n_users=9, n_hosts=23, fractional=True, gpus=True, spare_frac=1.0),  # hosts without running tasks that take placed jobs (found by the fuzz sweep)
    dict(seed=57, n_running=60000, n_pending=32, n_users=500, n_hosts=3000, max_preemption=32),  # multi-block scans / sorts
    dict(seed=58, n_running=30000, n_pending=16, n_users=4, n_hosts=900, fractional=True),       # heavy users: many re-scoring tiles, redone sequentially
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_rebalance_parity_random(make_engine, kw):
    P.rebalance_parity(make_engine, P.make_rebalance_case(**kw))


def test_rebalance_rescoring_paths(make_engine):
    P.rebalance_paths(make_engine)


def test_considerable_golden(make_engine):
    P.check_considerable_golden(make_engine)


@pytest.mark.parametrize("kw", [
    dict(seed=61, n=20000, n_users=200),
    dict(seed=62, n=125000, n_users=10000, fractional=True),               # one C4 pool's queue
    dict(seed=63, n=20000, n_users=200, tokens=False, pool_quota=False, eligible=False),
    dict(seed=64, n=20000, n_users=50, enforce=False),
    dict(seed=65, n=1, n_users=1),
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_considerable_parity_random(make_engine, kw):
    queue, st = P.make_considerable_case(**kw)
    for k in (1, 1000, 10 ** 6):
        P.considerable_parity(make_engine, queue, st, k)


def test_considerable_empty_queue(make_engine):
    queue, st = P.make_considerable_case(seed=66, n=0, n_users=3)
    assert len(P.considerable_parity(make_engine, queue, st, 10)) == 0


def test_rank_equal_dru_runs(make_engine):
    P.equal_dru_run_cases(make_engine)


def test_rank_user_usage(make_engine):
    got = P.user_usage_parity(make_engine, synth.make_pool(seed=81, n_pending=30000, n_running=50000, n_users=3000, n_offers=10, gpus=True), 3000)
    assert got[:, 0].sum() > 0 and got[:, 2].sum() > 0
    P.user_usage_parity(make_engine, synth.make_pool(seed=82, n_pending=500, n_running=40000, n_users=7, n_offers=10, fractional=True), 7)


def test_cycle_with_considerable_filters(make_engine):
    pool = synth.make_pool(seed=32, n_pending=20000, n_running=8000, n_users=500, n_offers=2000, gpus=True, constraints=True)
    _, st = P.make_considerable_case(seed=67, n=10, n_users=500)
    rng = np.random.default_rng(3)
    elig = (rng.random(pool.n_pending) < 0.9).astype(np.uint8)
    pos, j2o = P.cycle_considerable_parity(make_engine, pool, A.default_params(good_enough_fitness=1.0), 1000, st, elig)
    assert 0 < len(pos) <= 1000 and not np.array_equal(pos, np.arange(len(pos)))


@pytest.mark.parametrize("n,ge", [(2, 1.0), (4, 1.0), (6, 1.0), (4, 0.8), (6, 0.8)])
def test_multi_pool_context_forms(make_engine, n, ge, multi_mode):
    # the lockstep launches with their pools' contexts in the kernel arguments (PoolPack<2> / <4>) and, beyond four pools, read from memory
    pools = [synth.make_pool(seed=170 + i, n_pending=2500 + 500 * i, n_running=600, n_users=40, n_offers=300 + 250 * i, gpus=(i % 2 == 1),
                             constraints=(i % 3 == 0)) for i in range(n)]
    P.multi_pool_parity(make_engine, pools, A.default_params(good_enough_fitness=ge, match_algo=2), k=10 ** 9)


def test_lockstep_chain_of_pools_that_disagree(make_engine, multi_mode):
    # good-enough 0.8 next to best fit, K = 120 next to all pending, in ONE lockstep chain (two pools: contexts in the kernel arguments;
    # five: from memory)
    for n in (2, 5):
        pools = [synth.make_pool(seed=270 + i, n_pending=260 + 50 * i, n_running=60, n_users=12, n_offers=60 + 35 * i, gpus=(i % 2 == 0),
                                 constraints=(i % 2 == 1)) for i in range(n)]
        params = [A.default_params(good_enough_fitness=(0.8 if i % 2 == 0 else 1.0), match_algo=2) for i in range(n)]
        P.mixed_chain_parity(make_engine, pools, params, [120 if i % 3 == 0 else 10 ** 9 for i in range(n)])
        P.mixed_chain_parity(make_engine, pools, params, [120 if i % 3 == 0 else 10 ** 9 for i in range(n)], rank_batched=True)  # every pool its own K in ONE rank call


@pytest.mark.parametrize("seed", [611, 612])
def test_cycle_update_with_every_optional_column(make_engine, seed):
    P.cycle_update_xres_parity(make_engine, seed)


def test_cycle_update_moves_the_eligible_mask(make_engine):
    P.cycle_update_mask_parity(make_engine, seed=77)


def test_multi_pool(make_engine, multi_mode, algo=2):
    pools = [synth.make_pool(seed=71, n_pending=6000, n_running=2000, n_users=100, n_offers=3000, gpus=True, constraints=True),
             synth.make_pool(seed=72, n_pending=3000, n_running=500, n_users=50, n_offers=400),
             synth.make_pool(seed=73, n_pending=0, n_running=30, n_users=5, n_offers=20),
             synth.make_pool(seed=74, n_pending=5000, n_running=0, n_users=80, n_offers=150)]
    P.multi_pool_parity(make_engine, pools, A.default_params(good_enough_fitness=1.0, match_algo=algo), k=4000)


@pytest.mark.parametrize("kw", [
    dict(seed=401),
    dict(seed=403, n_remove=300, n_add=0, new_offers=False),
    dict(seed=406, n_pending=60000, n_running=25000, n_users=800, n_offers=3000, n_remove=9000, n_add=8000, k=20000),
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_cycle_update(make_engine, kw):
    P.cycle_update_parity(make_engine, **kw)


def test_edge_cases(make_engine):
    P.edge_cases(make_engine)


# ---- offer construction from node state (cook_offers_*) ------------------------------------------------------------------
def test_offers_golden(make_engine):
    P.check_offers_golden(make_engine)


@pytest.mark.parametrize("kw", [
    dict(seed=1, n_nodes=300, n_pods=2500),
    dict(seed=2, n_nodes=6250, n_pods=60000, disk=True, corrupt=0.05, n_attr_keys=8),     # one C4 pool's nodes
    dict(seed=3, n_nodes=64, n_pods=3000, disk=True, max_pods=40),
    dict(seed=7, n_nodes=50000, n_pods=400000, disk=True, corrupt=0.01, max_pods=110),    # the whole 50k-node cluster
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_offers_parity_random(make_engine, kw):
    nodes, pods, op = synth.make_cluster_state(**kw)
    P.offers_parity(make_engine, nodes, pods, op, str(kw))


def test_offers_edge_cases(make_engine):
    P.offers_edge_cases(make_engine)


def test_offers_feed_the_match(make_engine):
    P.offers_feed_match(make_engine, n_nodes=2000, n_pods=12000, n_jobs=6000)


# ---- why-unscheduled summaries and match-cycle metrics (cook_match_explain / cook_match_metrics) ------------------------------
@pytest.mark.parametrize("algo", [0, 1], ids=["default", "serial"])
def test_explain_parity(make_engine, algo):
    p = A.default_params(good_enough_fitness=1.0, match_algo=algo)
    pool = synth.make_pool(seed=22, n_pending=3000, n_running=100, n_users=20, n_offers=1500, gpus=True, constraints=True)
    pos, counts = P.explain_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, p, reserved=(3, 7, 90), tag="constraints")
    assert counts[:, 0].any() and counts[:, 7].any()
    pool = synth.make_pool(seed=23, n_pending=2000, n_running=0, n_users=10, n_offers=100)
    P.explain_parity(make_engine, pool.pending_jobs, pool.offers, None, p, tag="over-committed")
    jobs, offers, groups = P.slow_constraint_case(9, 600, 200)
    pos, counts = P.explain_parity(make_engine, jobs, offers, groups, p, tag="slow constraints")
    assert counts[:, 5].any() and counts[:, 8].any()


def test_metrics_known_answers(make_engine):
    P.metrics_known_answers(make_engine)


def test_metrics_parity(make_engine):
    p = A.default_params(good_enough_fitness=1.0)
    pool = synth.make_pool(seed=24, n_pending=6000, n_running=0, n_users=50, n_offers=2000, gpus=True, constraints=True)
    m = P.metrics_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, p, n_users=50, tag="integers")
    assert 0 < m["matched"] < 6000
    pool = synth.make_pool(seed=25, n_pending=20000, n_running=0, n_users=40, n_offers=1000, fractional=True)
    pool.pending_jobs.cpus[:] = pool.pending_jobs.cpus + 0.1
    P.metrics_parity(make_engine, pool.pending_jobs, pool.offers, None, p, n_users=40, tag="fractional")
    P.metrics_parity(make_engine, A.Jobs(cpus=np.zeros(0), mem=np.zeros(0)), pool.offers, None, p, tag="no jobs")


def test_explain_after_a_cycle(make_engine):
    pool = synth.make_pool(seed=31, n_pending=6000, n_running=2000, n_users=30, n_offers=500, gpus=True, constraints=True)
    P.cycle_explain_parity(make_engine, pool, A.default_params(good_enough_fitness=1.0), k=3000, n_users=30)


# ---- BASELINE.json's configurations at FULL size: rank + placement of every pending job, bit-exact against the oracle -------
def _full_cycle_parity(make_engine, pool, ge=1.0, threads=16, k=None, stats=None, algo=0):
    """rank of every task + placement of the first k ranked jobs (default: all pending) of one pool, bit-exact against the oracle"""
    p = A.default_params(good_enough_fitness=ge, match_algo=algo)
    with make_engine(p) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e.cycle_run(pool.n_pending if k is None else k)
        ranked, j2o, head = e.cycle_fetch()
        _, dru = e.rank_fetch(want_dru=True)
        if stats is not None:
            stats.update(e.match_stats())
    o_ranked, o_dru = pyoracle.rank(p, pool.tasks, pool.users)
    assert np.array_equal(ranked, o_ranked) and np.array_equal(dru, o_dru, equal_nan=True)
    pend_ord = np.cumsum(pool.tasks.pending) - 1
    kk = len(o_ranked) if k is None else min(k, len(o_ranked))
    # (the oracle's threaded sweep buckets the hosts per job; first-fit-above-threshold is order-dependent: one thread below 1.0)
    o_j2o, _, o_head = pyoracle.match(p, pool.pending_jobs.take(pend_ord[o_ranked[:kk]]), pool.offers, pool.groups,
                                      nthreads=threads if ge >= 1.0 else 1)
    assert len(j2o) == len(o_j2o)
    bad = np.nonzero(j2o != o_j2o)[0]
    assert len(bad) == 0 and head == o_head, f"assignment differs first at rank position {bad[:5]}"
    return j2o


def test_c2_full_size(make_engine):
    """configs[1]: 50k pending x 5k offers, cpus + mem only, single pool (+ 20k running tasks, 1000 users)"""
    pool = synth.make_pool(seed=0xC00C0002, n_pending=50000, n_running=20000, n_users=1000, n_offers=5000)
    j2o = _full_cycle_parity(make_engine, pool)
    assert 5000 < (j2o >= 0).sum() < 50000


def test_c3_full_size(make_engine):
    """configs[2]: 200k pending x 20k offers with host / attribute constraints + the gpu dimension (80k running, 2000 users)"""
    pool = synth.make_pool(seed=0xC00C0003, n_pending=200000, n_running=80000, n_users=2000, n_offers=20000, gpus=True, constraints=True)
    j2o = _full_cycle_parity(make_engine, pool)
    assert 20000 < (j2o >= 0).sum() < 200000


def test_c4_one_pool_full_size(make_engine):
    """configs[3], one of its 8 pools = what one GPU of the 8-GPU configuration runs: 125k pending x 6250 offers, 10k users"""
    pool = synth.make_pool(seed=0xC00C0004, n_pending=125000, n_running=50000, n_users=10000, n_offers=6250, gpus=True, constraints=True)
    stats = {}
    j2o = _full_cycle_parity(make_engine, pool, stats=stats)
    assert 10000 < (j2o >= 0).sum() < 125000
    # the merged lists of MV_LM entries out of per-chunk lists of MV_L: the early stop on a full chunk list (JL_TRUNC) is exercised at
    # the shipped shapes, and rounds do end on such lists
    assert stats["trunc_lists"] > 0 and stats["stop_list"] > 0, stats


def _c4_pool():
    return synth.make_pool(seed=0xC00C0004, n_pending=125000, n_running=50000, n_users=10000, n_offers=6250, gpus=True, constraints=True)


def test_c4_one_pool_good_enough_08_full_size(make_engine):
    """The reference's DEFAULT operating point at size (config.clj:110-113: good-enough-fitness 0.8): one full C4 pool, every pending job,
    against the single-thread oracle (first offer in array order whose fitness exceeds 0.8 wins outright, scheduler.clj:2312-2314;
    which of several good-enough hosts wins is oracle-defined).  VERDICT r2 item 1a."""
    j2o = _full_cycle_parity(make_engine, _c4_pool(), ge=0.8)
    assert 10000 < (j2o >= 0).sum() < 125000


@pytest.mark.parametrize("ge", [1.0, 0.8])
def test_c4_one_pool_k1000(make_engine, ge):
    """fenzo-max-jobs-considered 1000 (config.clj:113) on a full C4 pool: the cycle Cook runs on day one.  VERDICT r2 item 1b."""
    j2o = _full_cycle_parity(make_engine, _c4_pool(), ge=ge, k=1000)
    assert len(j2o) == 1000 and (j2o >= 0).sum() > 500


def test_rank_c5_queue_size(make_engine):
    """configs[4]'s queue: the rebalancer takes its pending jobs from the RANKED queue (rebalancer.clj:574-590), i.e. the rank of
    1M running + 500k pending tasks of one pool — order and DRUs bit-exact against the oracle.  VERDICT r2 item 1c."""
    pool = synth.make_pool(seed=0xC00C0005, n_pending=500_000, n_running=1_000_000, n_users=10_000, n_offers=64)
    ranked = P.rank_parity(make_engine, pool, A.default_params())
    assert len(ranked) > 100_000


@pytest.mark.parametrize("form", ["engine-choice", "served", "lockstep"])
def test_timed_configuration_parity(make_engine, form, monkeypatch):
    """The configuration bench.py TIMES, driven exactly as bench.py drives it (cook_amd/workload.py builds it for both): the 8
    pools of configs[3] on one rank, ShardedCluster.cycle = quota-group all-reduce inputs + rank per pool + the placements of
    all pools through the rank's multi-pool path — served walkers (the default: ONE persistent walker launch for the eight pools
    beside three streams of evaluation launches, LIVE on the GPU) and launch chains x lockstep slots (COOK_MATCH_SERVED=0).
    Checked bit-exact against the oracle on a repeated cycle: pools of different servers / chains, first and second slots.
    "engine-choice" = match_algo 0, what bench.py times: with eight engines on the device the engine places every eligible pool by class-ordered
    best fit (ONE cf_walk launch, a workgroup per pool); the other two forms pin the window rounds (match_algo 2)."""
    from cook_amd import sharding, workload
    from oracle import checks
    multi_mode = "served" if form == "engine-choice" else form
    monkeypatch.setenv("COOK_MATCH_SERVED", "1" if multi_mode == "served" else "0")
    spec = workload.ClusterSpec()
    params = A.default_params(good_enough_fitness=1.0, match_algo=0 if form == "engine-choice" else 2)
    pools = workload.make_pools(spec, range(spec.pools))
    engines = {}
    try:
        for p, pool in pools.items():
            engines[p] = make_engine(params)
            engines[p].cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        cl = sharding.ShardedCluster(engines, workload.quota_groups(spec))
        K = spec.per_pool[0]
        cl.cycle(K)
        cl.cycle(K)  # the timed region repeats cycles on resident inputs: check a repeat, not the first call
        st0 = engines[0].match_stats()
        if form == "engine-choice":
            assert all(e.match_stats()["placement_form"] == 3 for e in engines.values()), st0  # every pool of the benchmark's cluster is eligible
            assert st0["rank_batch_pools"] >= 2 and st0["rank_batch_single_ops"] <= 3 * st0["rank_batch_pools"], st0  # (alone: the three fills of the class tables' set-up)
        else:
            assert st0["placement_form"] == 0 and st0["served_mode"] == (1 if multi_mode == "served" else 0) and st0["served_fell_back"] == 0
            assert st0["rank_batch_pools"] >= 2 and st0["rank_batch_single_ops"] == 0, st0  # the ranks ran as joint sequences of launches (read-backs too)
        n_chains = min(spec.pools, cl.max_chains)
        check = sorted({0, 1 % spec.pools, (n_chains + 1) % spec.pools, spec.pools - 1})  # chains 0, 1, 1 (second slot), last (second slot)
        for p in check:
            ranked, j2o, _ = engines[p].cycle_fetch()
            q = cl.quota_inputs(p, cl.last_pool_usage[p], cl.last_group_usage)
            checks.check_pool_against_oracle(params, pools[p], q, ranked, j2o, K)
            assert 10000 < (j2o >= 0).sum() < K
        cl.close()
    finally:
        for e in engines.values():
            e.close()


def test_rank_batch_diverging_flows(make_engine):
    """cook_cycle_run_rank_multi on the GPU: eight pools whose rank flows differ (130k tasks beside 12k; tie-heavy, fractional, equal-DRU
    runs, gpu mode, quotas + offensive filter, no running task, considerable filters) in ONE call — ranked order, considerable positions,
    placements and per-user usage equal to cook_cycle_run on a fresh engine per pool, the order equal to the oracle's."""
    stats = P.rank_batch_parity(make_engine, P.rank_batch_cases(scale=20), k=3000, n_users=300, min_grouped=40)
    assert stats[-1]["rank_batch_single_ops"] <= 3 * 8, stats[-1]  # (alone: the usage vectors' copies to pageable memory, fills of the considerable filters / a second refinement)


def test_rank_batch_small_and_failing_flows(make_engine, monkeypatch):
    P.rank_batch_parity(make_engine, P.rank_batch_cases(), k=300, n_users=300)
    P.rank_batch_one_flow_fails(make_engine)
    monkeypatch.setenv("COOK_RANK_RADIX", "1")
    P.rank_batch_parity(make_engine, P.rank_batch_cases(scale=4)[1:5], k=10 ** 9)


def test_served_walkers_ragged_pools_many_cycles(make_engine, monkeypatch):
    """Served walkers LIVE, the case the emulator cannot run: six pools of very different sizes (walkers that finish at different
    times, serve iterations with one to three pools, long stretches where a server only polls), 1 to 3 serve streams, eight cycles
    each — every cycle of every pool against the lockstep launches' result of the same inputs (which the rest of the suite pins to
    the oracle).  This is the shape that exposed the late-workgroup race of the first served version."""
    from cook_amd.engine import cycle_match_multi
    sizes = [(30000, 1500), (900, 60), (12000, 3000), (150, 20), (20000, 800), (5000, 5000)]
    pools = [synth.make_pool(seed=900 + i, n_pending=npd, n_running=npd // 4, n_users=200, n_offers=m, gpus=(i % 2 == 0), constraints=(i % 3 != 1))
             for i, (npd, m) in enumerate(sizes)]
    params = A.default_params(good_enough_fitness=1.0, match_algo=2)  # (the window rounds: with six engines on the device match_algo 0 would pick class-ordered best fit)
    engines = [make_engine(params) for _ in pools]
    try:
        for e, pool in zip(engines, pools):
            e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)

        def cycle():
            for e in engines:
                e.cycle_run_rank(10 ** 9)
            cycle_match_multi(engines)
            return [e.cycle_fetch() for e in engines]
        monkeypatch.setenv("COOK_MATCH_SERVED", "0")
        want = cycle()
        assert engines[0].match_stats()["served_mode"] == 0
        monkeypatch.setenv("COOK_MATCH_SERVED", "1")
        for streams in (1, 2, 3):
            monkeypatch.setenv("COOK_SERVE_STREAMS", str(streams))
            for c in range(8):
                got = cycle()
                st_ = engines[0].match_stats()
                assert st_["served_mode"] == 1 and st_["served_fell_back"] == 0 and st_["serve_streams"] == streams
                for i, (a, b) in enumerate(zip(got, want)):
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], (streams, c, i)
    finally:
        for e in engines:
            e.close()


def test_c5_full_size(make_engine):
    """configs[4] at full size: 1M running tasks + 128 pending jobs examined, 50k hosts, NO spare capacity anywhere, so every
    decision comes out of the preemption-candidate scan (rebalancer.clj:320-407) — decisions, preempted task lists and pending
    DRUs bit-identical to the oracle."""
    b = P.make_rebalance_case(seed=0xC00C0005, n_running=1_000_000, n_pending=128, n_users=10_000, n_hosts=50_000,
                              max_preemption=128, quota_frac=0.02, spare_frac=0.0)
    got = P.rebalance_parity(make_engine, b, min_decisions=16)
    assert sum(len(d["tasks"]) for d in got["decisions"]) >= 16


def test_rccl_single_rank_collectives(make_engine):
    """The `nccl` (= RCCL) branch of the sharding layer on ONE rank: device tensors through dist.all_reduce, world size 1
    (VERDICT r1 item 6).  Checks the quota-group matrix and the per-user [U, 3] vector (written by the engine straight into
    the collective's device buffer) against the oracle's sums."""
    import os
    import torch
    import torch.distributed as dist
    from cook_amd import sharding
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        pools = [synth.make_pool(seed=91 + p, n_pending=3000, n_running=5000 + 100 * p, n_users=200, n_offers=10, gpus=True) for p in range(3)]
        groups = sharding.QuotaGroups(pool_group={0: 0, 1: 1, 2: 0})
        engines = [make_engine(A.default_params()) for _ in pools]
        for e, pool in zip(engines, pools):
            e.rank(pool.tasks, pool.users)
        local = sharding.group_usage_matrix(groups, {p: engines[p].rank_pool_usage().as_tuple() for p in range(3)})
        total = sharding.all_reduce_group_usage(local, 1, dev, force=True)
        want = sharding.group_usage_matrix(groups, {p: pyoracle.pool_usage(pools[p].tasks).as_tuple() for p in range(3)})
        assert np.array_equal(total, want) and total[0][0] == 5000 + 5200
        uu = sharding.all_reduce_user_usage(engines, 200, 1, dev, force=True)
        assert np.array_equal(uu, sum(pyoracle.user_usage(pool.tasks, 200) for pool in pools))
        for e in engines:
            e.close()
    finally:
        dist.destroy_process_group()


def test_offers_many_models_and_types(make_engine):
    P.offers_many_models_and_types(make_engine)


# ---- ports, named scalars, several entries per host in the k8s "gpus" / "disk" maps (scheduler.clj:456-471, 177-189) -----------
def test_xres_known_answers(make_engine):
    P.xres_known_answers(make_engine)


@ALGOS
@pytest.mark.parametrize("kw", [
    dict(seed=71, n=6000, m=400),
    dict(seed=72, n=6000, m=900, groups=True, constraints=True),
    dict(seed=73, n=6000, m=400, ports=False, scalars=3),
    dict(seed=74, n=6000, m=900, slots=3, scalars=0, ports=False),
    dict(seed=75, n=6000, m=900, slots=4, groups=True),
], ids=["ports+scalars", "with-groups-and-constraints", "three-scalars", "gpu-and-disk-maps", "everything"])
def test_match_ports_scalars_maps(make_engine, kw, algo):
    kw = dict(kw)
    jobs, offers, groups = P.xres_random_case(kw.pop("seed"), kw.pop("n"), kw.pop("m"), **kw)
    j2o = P.match_parity(make_engine, jobs, offers, groups, A.default_params(match_algo=algo))
    assert (j2o >= 0).sum() > 20 and (j2o < 0).sum() > 5


@pytest.mark.parametrize("ge", [0.8, 0.4])
def test_match_ports_scalars_good_enough_and_explain(make_engine, ge):
    jobs, offers, groups = P.xres_random_case(76, 6000, 400, groups=True, slots=2)
    p = A.default_params(good_enough_fitness=ge)
    P.match_parity(make_engine, jobs, offers, groups, p)
    P.explain_parity(make_engine, jobs, offers, groups, p, tag="xres")
    P.metrics_parity(make_engine, jobs, offers, groups, p, n_models=4, tag="xres")


def test_offers_slot_tables(make_engine):
    P.offers_slot_tables(make_engine)


def test_rebalance_gpu_maps_with_several_models(make_engine):
    got = P.rebalance_parity(make_engine, P.make_rebalance_case(seed=59, n_running=500, n_pending=60, n_users=15, n_hosts=40, constraints=True,
                                                                gpus=True, gpu_slots=2))
    assert len(got["decisions"]) > 0


@pytest.mark.parametrize("split", [1, 2, 4])
def test_match_eval_offer_split_levels(make_engine, monkeypatch, split):
    # idle rows of the eval grid take shares of the offers (eval_split): every cap gives the oracle's placement
    monkeypatch.setenv("COOK_EVAL_SPLIT", str(split))
    pool = synth.make_pool(seed=83, n_pending=5000, n_running=100, n_users=20, n_offers=3000, gpus=True, constraints=True)
    for ge in (1.0, 0.6):
        P.match_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, A.default_params(good_enough_fitness=ge))


def test_rank_tie_rule_in_tiles_and_as_radix_passes(make_engine, monkeypatch):
    P.tie_rule_forms(make_engine, monkeypatch, n_users=9000, per_user=3)  # one tie group per position, 9 000 items each: beyond a tile
    P.tie_rule_forms(make_engine, monkeypatch, n_users=150, per_user=5)


def test_guarded_fuzz_sweep():
    """A seeded sweep of small random configurations (rank / cycle / match / update / multi-pool served walkers / rebalancer) with every
    device buffer between two guard bands (COOK_GUARD=1, read when the library is loaded: hence a process of its own): bit-identical to
    the oracle and no write outside a buffer.  2 050 configurations of the same sweep: profiles/r05z_fuzz_guard_2050.txt."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_sweep.py"), "--guard", "--match", "120", "--rebalance", "40", "--multi", "30",
                        "--seed", "5150"], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "no write outside a device buffer" in r.stdout and "COOK_GUARD: " not in r.stderr, (r.stdout[-500:], r.stderr[-1500:])


def test_bench_two_ranks_on_one_gpu():
    """The multi-process path END TO END on real HIP engines: bench.py as two ranks (torch.distributed.run), both on cuda:0
    (COOK_BENCH_ONE_DEVICE=1, a test aid), the cycle's collectives over gloo — pools sharded p mod world, the quota-group all-reduce, the
    per-user all-reduce, the max-over-ranks timing, the parity check of every rank's pools against the oracle after the timed region,
    and the line's `collective` record.  (RCCL needs a GPU per rank; the single-rank RCCL collectives are test_rccl_single_rank_collectives.)"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, COOK_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo", "--no-cpu-baseline", "--no-extras",
           "--no-adjacent", "--no-roofline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["parity_checked"] is True
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2
    assert c["pools_of_rank"] == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert c["group_usage_equals_sum_over_all_pools"] is True
    assert d["last_cycle"]["considered"] > 900_000 and d["last_cycle"]["matched"] > 300_000


@pytest.mark.parametrize("case", [c for c in G_EXPLAIN if "engine" in c], ids=[c["name"] for c in G_EXPLAIN if "engine" in c])
def test_explain_reference_cases(make_engine, case):
    # the reference's own cases of the why-unscheduled reducer (tests/golden/explain.json) through cook_match / cook_match_explain
    P.explain_golden_engine(make_engine, case)


def test_pool_usage_multi(make_engine):
    P.pool_usage_multi_parity(make_engine, n=8)

"""C-ABI parity on a machine WITHOUT a GPU: the same cook_amd/csrc sources compiled against the SIMT emulator
(tests/simt_emu).  Checks kernel logic + host orchestration against the oracle; small sizes (the emulator is slow)."""
import os

import numpy as np
import pytest

from cook_amd import _abi as A
from cook_amd import synth
from cook_amd.engine import Engine
from tests import golden_util as _G
from tests import parity_cases as P

G_EXPLAIN = _G.load("explain")


@pytest.fixture(scope="module")
def make_engine():
    from tests.simt_emu import build_emu
    so = build_emu.build()
    return lambda params: Engine(params, lib_path=so)


def test_rank_golden(make_engine):
    P.check_rank_golden(make_engine)


def test_rank_group_golden(make_engine):
    P.check_rank_group_golden(make_engine)


def test_match_golden(make_engine):
    P.check_match_golden(make_engine)


@pytest.mark.parametrize("kw", [
    dict(seed=1, n_pending=700, n_running=300, n_users=40, n_offers=50),
    dict(seed=2, n_pending=5000, n_running=1500, n_users=300, n_offers=50),          # multi-block scans / sorts
    dict(seed=3, n_pending=900, n_running=300, n_users=25, n_offers=50, tie_heavy=True),
    dict(seed=4, n_pending=900, n_running=300, n_users=25, n_offers=50, fractional=True),
    dict(seed=5, n_pending=900, n_running=300, n_users=25, n_offers=50, no_shares=True),
    dict(seed=6, n_pending=600, n_running=0, n_users=7, n_offers=10, tie_heavy=True, quota_frac=0.5),
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items() if k != "n_offers"))
def test_rank_parity_random(make_engine, kw):
    pool = synth.make_pool(**kw)
    P.rank_parity(make_engine, pool, A.default_params(max_over_quota_jobs=10))


def test_rank_parity_gpu_mode_and_quota(make_engine):
    pool = synth.make_pool(seed=11, n_pending=800, n_running=400, n_users=30, n_offers=20, gpus=True)
    # give every task a gpu so the gpu-mode scores increase strictly
    pool.tasks.gpus[:] = np.maximum(pool.tasks.gpus, 1.0)
    P.rank_parity(make_engine, pool, A.default_params(dru_mode=1))
    q = A.pool_quota(pool_quota=A.quota(count=600, cpus=2500.0), group_quota=A.quota(mem=4.0e6),
                     group_usage=A.usage(count=10, cpus=100, mem=1.0e6))
    P.rank_parity(make_engine, pool, A.default_params(offensive_max_mem_mb=16000.0, offensive_max_cpus=6.0), quota=q)


def test_rank_equal_dru_runs(make_engine):
    P.equal_dru_run_cases(make_engine)


def test_rank_tie_rule_in_tiles_and_as_radix_passes(make_engine, monkeypatch):
    P.tie_rule_forms(make_engine, monkeypatch)
    P.tie_rule_forms(make_engine, monkeypatch, n_users=150, per_user=5)


def test_rank_user_usage(make_engine):
    got = P.user_usage_parity(make_engine, synth.make_pool(seed=81, n_pending=900, n_running=700, n_users=40, n_offers=10, gpus=True), 40)
    assert got[:, 0].sum() > 0 and got[:, 2].sum() > 0
    P.user_usage_parity(make_engine, synth.make_pool(seed=82, n_pending=500, n_running=2500, n_users=7, n_offers=10, fractional=True), 7)
    P.user_usage_parity(make_engine, synth.make_pool(seed=83, n_pending=50, n_running=0, n_users=5, n_offers=10), 5)


def test_rank_edge_cases(make_engine):
    p = A.default_params()
    # empty input
    empty = A.Tasks(cpus=np.zeros(0), mem=np.zeros(0), user=np.zeros(0), priority=np.zeros(0), start_ms=np.zeros(0),
                    task_id=np.zeros(0), job_id=np.zeros(0), pending=np.zeros(0))
    users = A.Users(div_cpus=np.ones(1), div_mem=np.ones(1))
    with make_engine(p) as e:
        ranked, _ = e.rank(empty, users)
    assert len(ranked) == 0
    # only running tasks -> empty queue; a single pending job
    pool = synth.make_pool(seed=3, n_pending=0, n_running=50, n_users=5, n_offers=4)
    assert len(P.rank_parity(make_engine, pool, p)) == 0
    pool = synth.make_pool(seed=3, n_pending=1, n_running=0, n_users=1, n_offers=4)
    assert len(P.rank_parity(make_engine, pool, p)) == 1
    # limiter that drops almost everything
    pool = synth.make_pool(seed=9, n_pending=300, n_running=50, n_users=3, n_offers=4, quota_frac=1.0)
    P.rank_parity(make_engine, pool, A.default_params(max_over_quota_jobs=0))


ALGOS = pytest.mark.parametrize("algo", [0, 1, 3], ids=["default", "serial", "classfit"])  # window rounds (shipped) / the one-job-at-a-time sweep / class-ordered best fit


@ALGOS
@pytest.mark.parametrize("ge", [1.0, 0.8, 0.5])
def test_match_parity_random(make_engine, ge, algo):
    pool = synth.make_pool(seed=21, n_pending=400, n_running=100, n_users=20, n_offers=300)
    P.match_parity(make_engine, pool.pending_jobs, pool.offers, None, A.default_params(good_enough_fitness=ge, match_algo=algo))


@ALGOS
def test_match_parity_constraints(make_engine, algo):
    pool = synth.make_pool(seed=22, n_pending=400, n_running=100, n_users=20, n_offers=200, gpus=True, constraints=True)
    j2o = P.match_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups,
                         A.default_params(good_enough_fitness=1.0, match_algo=algo), reserved=(3, 7, 150))
    assert (j2o >= 0).sum() > 50


@ALGOS
def test_match_overcommitted_cluster(make_engine, algo):
    # demand >> capacity: hosts fill up, lists run out, the tail of the queue fails (fail codes must match too)
    pool = synth.make_pool(seed=23, n_pending=700, n_running=0, n_users=10, n_offers=48)  # more offers than a list holds
    p = A.default_params(good_enough_fitness=1.0, match_algo=algo)
    j2o = P.match_parity(make_engine, pool.pending_jobs, pool.offers, None, p)
    assert (j2o < 0).sum() > 100


def test_match_long_windows(make_engine, algo=0):
    # a cluster that is full after a few hundred jobs: from then on nearly every job is settled in the parallel phase of the resolve
    # kernel, the window grows past the LDS-staged size (MV_WLONG) and only the few jobs that still need the walk are staged —
    # gpu jobs, constrained jobs and group members keep some of those in every window
    # (more than MV_T = 64 gpu hosts: a job that only fits hosts of the wrong kind must fail on MORE offers than a round can touch
    # for its summary to be final, see the `trivial` rule of resolve_round)
    pool = synth.make_pool(seed=29, n_pending=9000, n_running=0, n_users=30, n_offers=800, gpus=True, constraints=True)
    pool.offers.cpus[:] = np.minimum(pool.offers.cpus, 12.0)  # small hosts: the cluster is full after ~2000 jobs
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
    # balanced / attribute-equals / unique groups incl. running cotasks (constraints.clj:586-644)
    rng = np.random.default_rng(5)
    n, m = 120, 40
    attr = np.zeros((m, 2), dtype=np.uint32)
    attr[:, 0] = rng.integers(1, 4, m)
    attr[:, 1] = rng.integers(0, 3, m)  # 0 = attribute absent on some hosts
    offers = A.Offers(cpus=np.full(m, 8.0), mem=np.full(m, 16000.0), attr=attr, k8s=np.ones(m, dtype=np.uint8))
    group = rng.integers(0, 6, n).astype(np.uint32)
    group[rng.random(n) < 0.3] = A.NONE_U32
    jobs = A.Jobs(cpus=rng.integers(1, 4, n).astype(float), mem=rng.integers(1, 4, n) * 1000.0, group=group)
    groups = A.Groups(type=np.array([1, 2, 2, 3, 3, 0], dtype=np.uint8),
                      attr_key=np.array([A.NONE_U32, 0, A.NONE_U32, 1, 0, 0], dtype=np.uint32),
                      minimum=np.array([0, 3, 10, 0, 0, 0], dtype=np.int32),
                      run_hosts=[[1, 2], [3], [], [], [5, 6], []],
                      run_attrs=[[0, 0], [int(attr[3, 0])], [], [], [int(attr[5, 0]), int(attr[6, 0])], []])
    P.match_parity(make_engine, jobs, offers, groups, A.default_params(good_enough_fitness=1.0, match_algo=algo))


@ALGOS
def test_match_constraints_beyond_the_fast_paths(make_engine, algo):
    jobs, offers, groups = P.slow_constraint_case(9, 200, 60)
    j2o = P.match_parity(make_engine, jobs, offers, groups, A.default_params(good_enough_fitness=1.0, match_algo=algo))
    assert (j2o >= 0).sum() > 20


def test_match_many_distinct_candidates_and_touched_set_limits(make_engine):
    # many distinct candidate offers per window (the walk continues through several segments of one evaluated window) / every job
    # lands on its own host (each commit touches a new offer: rounds end on the 64 lanes): the result must stay exact
    jobs, offers = P.pinned_jobs_case(7, 300, 900, 48)
    p = A.default_params(good_enough_fitness=1.0)
    P.match_parity(make_engine, jobs, offers, None, p)
    with make_engine(p) as e:
        e.match(jobs, offers)
        st_ = e.match_stats()
        assert st_["segments"] >= st_["rounds"] > 0, st_
        # ... and touches more offers per round than the walk has lanes: lanes of dead offers (full to the smallest job) are given away
        assert st_["touched"] > 64 * st_["rounds"] or st_["rounds"] > 2, st_
    jobs, offers = P.pinned_jobs_case(8, 300, 400, 0)
    P.match_parity(make_engine, jobs, offers, None, p)
    with make_engine(p) as e:
        e.match(jobs, offers)
        assert e.match_stats()["stop_full"] > 0


def test_match_refuses_pools_beyond_the_offer_table(make_engine):
    # the walk's owner table is one LDS byte per offer of the pool: a pool of 200 000 offers is refused, loudly, before any kernel runs
    from cook_amd.engine import CookError
    pool = synth.make_pool(seed=5, n_pending=4, n_running=0, n_users=2, n_offers=200_000)
    with make_engine(A.default_params()) as e:
        with pytest.raises(CookError, match="too many offers"):
            e.match(pool.pending_jobs, pool.offers)


def test_cycle_parity(make_engine):
    pool = synth.make_pool(seed=31, n_pending=600, n_running=200, n_users=30, n_offers=100, gpus=True, constraints=True)
    P.cycle_parity(make_engine, pool, A.default_params(good_enough_fitness=1.0), k=150)


def test_rebalance_golden(make_engine):
    P.check_rebalance_golden(make_engine)


@pytest.mark.parametrize("kw", [
    dict(seed=51, n_running=400, n_pending=24, n_users=12, n_hosts=30),
    dict(seed=52, n_running=400, n_pending=24, n_users=12, n_hosts=30, fractional=True),           # exact fix-up paths
    dict(seed=53, n_running=600, n_pending=30, n_users=20, n_hosts=3, max_preemption=12),
    dict(seed=60, n_running=800, n_pending=30, n_users=20, n_hosts=8, max_preemption=16, spare_frac=0.0),     # 65..128 items per host: lists in LDS          # hosts beyond the LDS cap
    dict(seed=54, n_running=500, n_pending=40, n_users=15, n_hosts=40, constraints=True, gpus=True),
    dict(seed=55, n_running=300, n_pending=20, n_users=8, n_hosts=25, dru_mode=1),
    dict(seed=56, n_running=0, n_pending=10, n_users=3, n_hosts=8, spare_frac=1.0),
    dict(seed=707730441, n_running=2, n_pending=29, n_users=9, n_hosts=23, fractional=True, gpus=True, spare_frac=1.0),  # hosts without running tasks that take placed jobs (found by the fuzz sweep)                # spare resources only
    dict(seed=57, n_running=3600, n_pending=10, n_users=2, n_hosts=90),                            # users of several re-scoring tiles
    dict(seed=58, n_running=2600, n_pending=8, n_users=2, n_hosts=70, fractional=True),            # ... redone sequentially
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_rebalance_parity_random(make_engine, kw):
    P.rebalance_parity(make_engine, P.make_rebalance_case(**kw))


def test_rebalance_rescoring_paths(make_engine):
    P.rebalance_paths(make_engine)


def test_considerable_golden(make_engine):
    P.check_considerable_golden(make_engine)


@pytest.mark.parametrize("kw", [
    dict(seed=61, n=900, n_users=12),
    dict(seed=62, n=3000, n_users=40, fractional=True),                    # exact fix-up path, multi-block scans
    dict(seed=63, n=900, n_users=12, tokens=False, pool_quota=False, eligible=False),
    dict(seed=64, n=900, n_users=5, enforce=False),
    dict(seed=65, n=1, n_users=1),
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_considerable_parity_random(make_engine, kw):
    queue, st = P.make_considerable_case(**kw)
    for k in (1, 50, 10 ** 6):
        P.considerable_parity(make_engine, queue, st, k)


def test_considerable_empty_queue(make_engine):
    queue, st = P.make_considerable_case(seed=66, n=0, n_users=3)
    assert len(P.considerable_parity(make_engine, queue, st, 10)) == 0


def test_cycle_with_considerable_filters(make_engine):
    pool = synth.make_pool(seed=32, n_pending=600, n_running=200, n_users=30, n_offers=100, gpus=True, constraints=True)
    _, st = P.make_considerable_case(seed=67, n=10, n_users=30)
    rng = np.random.default_rng(3)
    elig = (rng.random(pool.n_pending) < 0.9).astype(np.uint8)
    pos, j2o = P.cycle_considerable_parity(make_engine, pool, A.default_params(good_enough_fitness=1.0), 150, st, elig)
    assert 0 < len(pos) <= 150 and not np.array_equal(pos, np.arange(len(pos)))


def test_lockstep_chain_of_pools_that_disagree(make_engine, multi_mode):
    # good-enough 0.8 next to best fit, K = 120 next to all pending, in ONE lockstep chain (two pools: contexts in the kernel arguments;
    # five: from memory)
    for n in (2, 5):
        pools = [synth.make_pool(seed=270 + i, n_pending=260 + 50 * i, n_running=60, n_users=12, n_offers=60 + 35 * i, gpus=(i % 2 == 0),
                                 constraints=(i % 2 == 1)) for i in range(n)]
        params = [A.default_params(good_enough_fitness=(0.8 if i % 2 == 0 else 1.0), match_algo=2) for i in range(n)]
        P.mixed_chain_parity(make_engine, pools, params, [120 if i % 3 == 0 else 10 ** 9 for i in range(n)])
        P.mixed_chain_parity(make_engine, pools, params, [120 if i % 3 == 0 else 10 ** 9 for i in range(n)], rank_batched=True)  # every pool its own K in ONE rank call


def test_cycle_update_rejects_rows_it_cannot_append(make_engine):
    # cook_cycle_update validates the appended rows before it touches the resident columns: a user id beyond the staged users, a
    # pending job without its user column when the staged jobs carry one, a pending-job count that disagrees with add_tasks
    from cook_amd.engine import CookError
    p = A.default_params()
    pool = synth.make_pool(seed=91, n_pending=120, n_running=40, n_users=9, n_offers=20, constraints=True)
    extra = synth.make_pool(seed=92, n_pending=6, n_running=4, n_users=9, n_offers=4, constraints=True, id_base=27_592_186_044_416)
    extra.pending_jobs.group = None
    with make_engine(p) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e.cycle_run(10 ** 9)
        want = e.cycle_fetch()
        bad = synth.make_pool(seed=92, n_pending=6, n_running=4, n_users=9, n_offers=4, constraints=True, id_base=27_592_186_044_416)
        bad.tasks.user = bad.tasks.user.copy()
        bad.tasks.user[0] = 500  # no such user
        with pytest.raises(CookError, match="user id out of range"):
            e.cycle_update((), bad.tasks, bad.pending_jobs, None)
        nouser = extra.pending_jobs.take(np.arange(extra.pending_jobs.n))
        nouser.user = None
        with pytest.raises(CookError, match="user column"):
            e.cycle_update((), extra.tasks, nouser, None)
        with pytest.raises(CookError, match="must equal"):
            e.cycle_update((), extra.tasks, extra.pending_jobs.take(np.arange(3)), None)
        # a removal list is checked on the device, after the compactions ran into the columns' second buffers: nothing is swapped in
        with pytest.raises(CookError, match="out of range or twice"):
            e.cycle_update(np.array([3, 17, 3], np.uint32), extra.tasks, extra.pending_jobs, pool.offers)
        with pytest.raises(CookError, match="out of range or twice"):
            e.cycle_update(np.array([5, pool.tasks.n], np.uint32), None, None, None)
        # fresh offers the engine refuses (a gpu model table without its counts; offers without mem): checked BEFORE the task / job
        # delta is applied — a caller that retries must not remove the same rows twice
        import copy
        bad_offers = copy.copy(pool.offers)
        bad_offers.gpu_model, bad_offers.gpu_slots, bad_offers.gpu_count = np.ones((pool.offers.n, 1), np.uint32), 1, None
        with pytest.raises(CookError, match="gpu_model without gpu_count"):
            e.cycle_update(np.array([3, 17], np.uint32), extra.tasks, extra.pending_jobs, bad_offers)
        # CSR constraint columns of the delta: offsets that decrease, or values missing behind a non-empty list -> refused, not a fault
        broken = extra.pending_jobs.take(np.arange(extra.pending_jobs.n))
        broken.eq_off = broken.eq_off.copy()
        broken.eq_off[0] = 1
        with pytest.raises(CookError, match="eq_off must start at 0"):
            e.cycle_update((), extra.tasks, broken, None)
        e.cycle_run(10 ** 9)  # the refused updates left the resident state as it was
        again = e.cycle_fetch()
        assert np.array_equal(want[0], again[0]) and np.array_equal(want[1], again[1])
        # ... and the same delta with good offers is then applied once, exactly as on an engine that never saw the refused calls
        e.cycle_update(np.array([3, 17], np.uint32), extra.tasks, extra.pending_jobs, pool.offers)
        e.cycle_run(10 ** 9)
        after = e.cycle_fetch()
    with make_engine(p) as e2:
        e2.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e2.cycle_update(np.array([3, 17], np.uint32), extra.tasks, extra.pending_jobs, pool.offers)
        e2.cycle_run(10 ** 9)
        clean = e2.cycle_fetch()
    assert np.array_equal(after[0], clean[0]) and np.array_equal(after[1], clean[1])


@pytest.mark.parametrize("seed", [611, 612])
def test_cycle_update_with_every_optional_column(make_engine, seed):
    P.cycle_update_xres_parity(make_engine, seed)


def test_cycle_update_moves_the_eligible_mask(make_engine):
    P.cycle_update_mask_parity(make_engine, seed=77)


def test_multi_pool(make_engine, multi_mode, algo=2):
    # three pools of different sizes (different numbers of offer chunks, rounds and K, one of them with nothing pending): in lockstep
    # launches (blockIdx.z = pool)
    pools = [synth.make_pool(seed=71, n_pending=400, n_running=100, n_users=20, n_offers=300, gpus=True, constraints=True),
             synth.make_pool(seed=72, n_pending=150, n_running=50, n_users=10, n_offers=40),
             synth.make_pool(seed=73, n_pending=0, n_running=30, n_users=5, n_offers=20)]
    P.multi_pool_parity(make_engine, pools, A.default_params(good_enough_fitness=1.0, match_algo=algo), k=300)


@pytest.mark.parametrize("n,ge", [(2, 1.0), (4, 1.0), (6, 1.0), (4, 0.8), (6, 0.8)])
def test_multi_pool_context_forms(make_engine, n, ge, multi_mode):
    # how the lockstep launches get their pools' contexts: in the kernel arguments for up to four pools (PoolPack<2> / PoolPack<4>,
    # picked by blockIdx.z), from a context record in memory beyond that — same placements either way, best fit and good-enough
    pools = [synth.make_pool(seed=170 + i, n_pending=220 + 40 * i, n_running=60, n_users=12, n_offers=50 + 30 * i, gpus=(i % 2 == 1),
                             constraints=(i % 3 == 0)) for i in range(n)]
    P.multi_pool_parity(make_engine, pools, A.default_params(good_enough_fitness=ge, match_algo=2), k=10 ** 9)


@pytest.mark.parametrize("whole", [True, False, None, "engine-choice"], ids=["chain-runs-rank-and-placement", "rank-barrier-placement", "served-walkers", "engine-choice"])
def test_sharded_cluster_lockstep_chains(make_engine, whole, monkeypatch):
    # ShardedCluster.cycle as bench.py drives it, five pools on two launch chains (slots 3 + 2): quota inputs, rank per pool,
    # lockstep placement per chain; every pool against the oracle, on a repeated cycle
    from cook_amd import sharding, workload
    from oracle import checks
    # "engine-choice": match_algo 0 with six engines on the device = class-ordered best fit for every eligible pool (one cf_walk launch, a workgroup per pool), the
    # others by the served walkers; the first three forms pin the window rounds (match_algo 2)
    auto = whole == "engine-choice"
    if auto:
        whole = None
    monkeypatch.setenv("COOK_MATCH_SERVED", "1" if whole is None else "0")  # (read by ShardedCluster and by the library)
    spec = workload.ClusterSpec(pools=6 if auto else 5, pending=1500, running=500, offers=400, users=40)
    params = A.default_params(good_enough_fitness=1.0, match_algo=0 if auto else 2)
    pools = workload.make_pools(spec, range(spec.pools))
    engines = {}
    try:
        for p, pool in pools.items():
            engines[p] = make_engine(params)
            engines[p].cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        cl = sharding.ShardedCluster(engines, workload.quota_groups(spec))
        cl.max_chains, cl.chain_whole_cycle = 2, bool(whole)
        assert cl.served == (whole is None)  # (None: all five pools in one served call)
        cl.close()

        class Serial:  # the emulator runs one launch at a time (one process-wide fiber scheduler): the chains take turns here
            map = staticmethod(lambda fn, xs: [fn(x) for x in xs])
            shutdown = staticmethod(lambda wait=True: None)
        cl._tp = cl._tp_rank = Serial()
        K = spec.per_pool[0]
        cl.cycle(K)
        cl.cycle(K)
        assert cl.last_phase_ms[1] > 0.0
        if auto:
            forms = [engines[p].match_stats()["placement_form"] for p in pools]
            assert forms.count(3) >= 3 and set(forms) <= {0, 3}, forms
        else:
            assert engines[0].match_stats()["served_mode"] == (2 if whole is None else 0)  # (2: the stepping form, all the emulator can run)
        for p in pools:
            ranked, j2o, _ = engines[p].cycle_fetch()
            q = cl.quota_inputs(p, cl.last_pool_usage[p], cl.last_group_usage)
            checks.check_pool_against_oracle(params, pools[p], q, ranked, j2o, K)
        cl.close()
    finally:
        for e in engines.values():
            e.close()


def test_many_pools_good_enough(make_engine, multi_mode):
    # five pools in one lockstep chain (contexts from memory: more than a PoolPack holds), good-enough 0.8 (the reference's default),
    # groups of every type in some of the pools
    pools = [synth.make_pool(seed=171 + i, n_pending=250 + 60 * i, n_running=40, n_users=12, n_offers=90 + 40 * i, gpus=(i % 2 == 0),
                             constraints=(i % 2 == 0)) for i in range(5)]
    P.multi_pool_parity(make_engine, pools, A.default_params(good_enough_fitness=0.8, match_algo=2), k=400)


@pytest.mark.parametrize("kw", [
    dict(seed=401),
    dict(seed=402, n_remove=0, n_add=60),                        # only submissions
    dict(seed=403, n_remove=300, n_add=0, new_offers=False),     # only departures, the offers stay
    dict(seed=404, n_pending=60, n_running=0, n_remove=60, n_add=40, n_offers=20),  # everything staged goes
    dict(seed=405, n_pending=2500, n_running=900, n_remove=700, n_add=650, k=400),
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_cycle_update(make_engine, kw):
    P.cycle_update_parity(make_engine, **kw)


def test_edge_cases(make_engine):
    P.edge_cases(make_engine)


# ---- offer construction from node state ---------------------------------------------------------------------------------------
def test_offers_golden(make_engine):
    P.check_offers_golden(make_engine)


@pytest.mark.parametrize("kw", [
    dict(seed=1, n_nodes=300, n_pods=2500),
    dict(seed=2, n_nodes=700, n_pods=9000, disk=True, corrupt=0.05, n_attr_keys=5),       # multi-block sort, foreign models / types
    dict(seed=3, n_nodes=64, n_pods=3000, disk=True, max_pods=40),                        # long pod lists, pod limit
    dict(seed=4, n_nodes=500, n_pods=40, gpus=False, fractional=False),                   # mostly idle nodes
], ids=lambda kw: "-".join(f"{k}{v}" for k, v in kw.items()))
def test_offers_parity_random(make_engine, kw):
    nodes, pods, op = synth.make_cluster_state(**kw)
    P.offers_parity(make_engine, nodes, pods, op, str(kw))


def test_offers_edge_cases(make_engine):
    P.offers_edge_cases(make_engine)


def test_offers_feed_the_match(make_engine):
    P.offers_feed_match(make_engine)


# ---- why-unscheduled summaries and match-cycle metrics ------------------------------------------------------------------------
def _group_case():
    rng = np.random.default_rng(5)
    n, m = 160, 40
    attr = np.zeros((m, 2), dtype=np.uint32)
    attr[:, 0] = rng.integers(1, 4, m)
    attr[:, 1] = rng.integers(0, 3, m)
    offers = A.Offers(cpus=np.full(m, 8.0), mem=np.full(m, 16000.0), attr=attr, k8s=np.ones(m, dtype=np.uint8),
                      max_tasks=np.full(m, 6, np.int32), num_tasks=rng.integers(0, 5, m).astype(np.int32))
    group = rng.integers(0, 6, n).astype(np.uint32)
    group[rng.random(n) < 0.3] = A.NONE_U32
    jobs = A.Jobs(cpus=rng.integers(1, 4, n).astype(float), mem=rng.integers(1, 4, n) * 1000.0, group=group)
    groups = A.Groups(type=np.array([1, 2, 2, 3, 3, 0], dtype=np.uint8),
                      attr_key=np.array([A.NONE_U32, 0, A.NONE_U32, 1, 0, 0], dtype=np.uint32),
                      minimum=np.array([0, 3, 10, 0, 0, 0], dtype=np.int32),
                      run_hosts=[[1, 2], [3], [], [], [5, 6], []],
                      run_attrs=[[0, 0], [int(attr[3, 0])], [], [], [int(attr[5, 0]), int(attr[6, 0])], []])
    return jobs, offers, groups


@pytest.mark.parametrize("algo", [0, 1], ids=["default", "serial"])
def test_explain_parity(make_engine, algo):
    p = A.default_params(good_enough_fitness=1.0, match_algo=algo)
    pool = synth.make_pool(seed=22, n_pending=400, n_running=100, n_users=20, n_offers=120, gpus=True, constraints=True)
    pos, counts = P.explain_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, p, reserved=(3, 7, 90), tag="constraints")
    assert counts[:, 0].any() and counts[:, 7].any()  # resources and the gpu-host constraint both occur
    pool = synth.make_pool(seed=23, n_pending=500, n_running=0, n_users=10, n_offers=24)
    P.explain_parity(make_engine, pool.pending_jobs, pool.offers, None, p, tag="over-committed")
    jobs, offers, groups = _group_case()
    pos, counts = P.explain_parity(make_engine, jobs, offers, groups, p, tag="groups")
    assert counts[:, 9].any() and counts[:, 11:14].any()  # max-tasks-per-host and group constraints occur
    jobs, offers, groups = P.slow_constraint_case(9, 200, 60)
    pos, counts = P.explain_parity(make_engine, jobs, offers, groups, p, tag="slow constraints")
    assert counts[:, 5].any() and counts[:, 8].any()


def test_explain_is_the_references_map():
    # unscheduled.clj's vector (test/cook/test/unscheduled.clj:57-72) and fenzo_utils' reduction (test/.../fenzo_utils.clj:76-91)
    row = np.zeros(A.WHY_SLOTS, np.uint32)
    row[[0, 1, 8]] = (8, 14, 3)
    assert A.why_summary(row) == {":constraints": {"novel_host_constraint": 3}, ":resources": {"mem": 14, "cpus": 8}}
    assert A.why_summary(np.zeros(A.WHY_SLOTS, np.uint32)) == {}
    # ... and what the user sees (test/cook/test/unscheduled.clj:66-72; the vector lists mem before cpus)
    shown = A.why_reasons({":resources": {"mem": 14, "cpus": 8}, ":constraints": {"novel_host_constraint": 3}})
    assert shown == [dict(reason="Not enough mem available.", host_count=14), dict(reason="Not enough cpus available.", host_count=8),
                     dict(reason="Job already ran on this host.", host_count=3)]


def test_metrics_known_answers(make_engine):
    P.metrics_known_answers(make_engine)


def test_metrics_parity(make_engine):
    p = A.default_params(good_enough_fitness=1.0)
    pool = synth.make_pool(seed=24, n_pending=700, n_running=0, n_users=25, n_offers=90, gpus=True, constraints=True)
    m = P.metrics_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, p, n_users=25, tag="integers")
    assert 0 < m["matched"] < 700
    pool = synth.make_pool(seed=25, n_pending=3000, n_running=0, n_users=40, n_offers=150, fractional=True)
    pool.pending_jobs.cpus[:] = pool.pending_jobs.cpus + 0.1    # non-dyadic: the totals need the in-order fold
    P.metrics_parity(make_engine, pool.pending_jobs, pool.offers, None, p, n_users=40, tag="fractional")
    # nothing considered / nothing offered
    none = A.Jobs(cpus=np.zeros(0), mem=np.zeros(0))
    P.metrics_parity(make_engine, none, pool.offers, None, p, tag="no jobs")
    one = A.Jobs(cpus=np.array([1.0]), mem=np.array([1.0]))
    P.metrics_parity(make_engine, one, A.Offers(cpus=np.zeros(0), mem=np.zeros(0)), None, p, tag="no offers")


def test_explain_after_a_cycle(make_engine):
    pool = synth.make_pool(seed=31, n_pending=600, n_running=200, n_users=30, n_offers=60, gpus=True, constraints=True)
    P.cycle_explain_parity(make_engine, pool, A.default_params(good_enough_fitness=1.0), k=300, n_users=30)


def test_new_entry_points_fail_loudly(make_engine):
    from cook_amd.engine import CookError
    p = A.default_params(good_enough_fitness=1.0)
    with make_engine(p) as e:
        for call in (lambda: e.match_explain([0]), lambda: e.match_metrics(), lambda: e.offers_run()):
            with pytest.raises(CookError) as ei:  # nothing staged / no match ran: COOK_E_STATE, outputs untouched
                call()
            assert ei.value.code == -4
        jobs = A.Jobs(cpus=np.array([1.0, 2.0]), mem=np.array([1.0, 2.0]))
        e.match(jobs, A.Offers(cpus=np.array([4.0]), mem=np.array([4.0])))
        with pytest.raises(CookError) as ei:  # a position beyond the jobs of the last match
            e.match_explain([2])
        assert ei.value.code == -1
        with pytest.raises(CookError) as ei:  # per-user counts without the jobs' user column
            e.match_metrics(n_users=3)
        assert ei.value.code == -1
        assert e.match_explain([]).shape == (0, A.WHY_SLOTS)
        assert e.match_metrics()["matched"] == 2  # the engine is still usable after the errors


def test_offers_many_models_and_types(make_engine):
    P.offers_many_models_and_types(make_engine)


def test_fuzz_rank_cycle_match_explain(make_engine):
    """seeded sweep over small random configurations (sizes, gpus / constraints / fractional / tie-heavy / no-shares inputs,
    quotas, good-enough 1.0-0.3, every match_algo, reserved hosts): rank, cycle, match and explain against the oracle"""
    rng = np.random.default_rng(20260923)
    for _ in range(24):
        kw = dict(seed=int(rng.integers(1, 1 << 30)), n_pending=int(rng.integers(1, 300)), n_running=int(rng.integers(0, 150)),
                  n_users=int(rng.integers(1, 30)), n_offers=int(rng.integers(1, 100)), gpus=bool(rng.integers(0, 2)),
                  constraints=bool(rng.integers(0, 2)), fractional=bool(rng.integers(0, 2)), tie_heavy=bool(rng.integers(0, 2)),
                  no_shares=bool(rng.integers(0, 4) == 0), quota_frac=float(rng.choice([0.0, 0.02, 0.5])))
        p = A.default_params(good_enough_fitness=float(rng.choice([1.0, 0.8, 0.5, 0.3])), match_algo=int(rng.choice([0, 0, 0, 1])),
                             max_over_quota_jobs=int(rng.choice([0, 3, 100])))
        pool = synth.make_pool(**kw)
        if rng.integers(0, 3) == 0:  # ports and named scalars on a third of the configurations (also through the cycle's job columns)
            n, m = pool.pending_jobs.n, pool.offers.n
            sc = rng.integers(1, 30, (n, 2)).astype(np.float64) * 0.5
            sc[rng.random((n, 2)) < 0.5] = np.nan
            pool.pending_jobs.ports = np.where(rng.random(n) < 0.3, rng.integers(1, 4, n), 0).astype(np.int32)
            pool.pending_jobs.scalars, pool.pending_jobs.n_scalars = sc, 2
            pool.offers.ports = rng.integers(0, 7, m).astype(np.int32)
            pool.offers.scalars, pool.offers.n_scalars = rng.integers(0, 120, (m, 2)) * 0.5, 2
        P.rank_parity(make_engine, pool, p)
        P.cycle_parity(make_engine, pool, p, int(rng.integers(1, kw["n_pending"] + 1)))
        reserved = tuple(int(x) for x in rng.integers(0, kw["n_offers"], int(rng.integers(0, 3))))
        j2o = P.match_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, p, reserved=reserved)
        if (j2o < 0).any():
            P.explain_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, p, max_pos=6, tag=str(kw))


def test_fuzz_offers_considerable_rebalance(make_engine):
    """seeded sweep over small random configurations of the other entry points (this sweep found the scratch-region collision of
    hosts without running tasks in the rebalancer)"""
    rng = np.random.default_rng(20260924)
    for _ in range(20):
        seed = int(rng.integers(1, 1 << 30))
        nodes, pods, op = synth.make_cluster_state(seed=seed, n_nodes=int(rng.integers(1, 300)), n_pods=int(rng.integers(0, 2000)),
                                                   gpus=bool(rng.integers(0, 2)), disk=bool(rng.integers(0, 2)),
                                                   fractional=bool(rng.integers(0, 2)), n_attr_keys=int(rng.choice([0, 3, 8])),
                                                   max_pods=int(rng.choice([4, 16, 110])), corrupt=float(rng.choice([0.0, 0.05])))
        P.offers_parity(make_engine, nodes, pods, op, f"offers {seed}")
        n = int(rng.integers(1, 1000))
        q, st = P.make_considerable_case(seed, n, int(rng.integers(1, 50)), fractional=bool(rng.integers(0, 2)),
                                         tokens=bool(rng.integers(0, 2)), enforce=bool(rng.integers(0, 2)),
                                         pool_quota=bool(rng.integers(0, 2)), eligible=bool(rng.integers(0, 2)))
        P.considerable_parity(make_engine, q, st, int(rng.integers(0, n + 5)))
        P.rebalance_parity(make_engine, P.make_rebalance_case(
            seed=seed, n_running=int(rng.integers(0, 500)), n_pending=int(rng.integers(1, 30)), n_users=int(rng.integers(1, 20)),
            n_hosts=int(rng.integers(1, 60)), fractional=bool(rng.integers(0, 2)), constraints=bool(rng.integers(0, 2)),
            gpus=bool(rng.integers(0, 2)), dru_mode=int(rng.integers(0, 2)), spare_frac=float(rng.choice([0.0, 0.2, 1.0]))))


def test_fuzz_groups_constraints_metrics_replay(make_engine):
    """seeded sweep: random group tables (unique / balanced / attribute-equals with running cotasks, task limits), constraints beyond
    the fast paths, metrics, and the replay loop over random traces — engine vs oracle"""
    from tests.test_replay import CONFIG, TIGHT
    rng = np.random.default_rng(20260925)
    for _ in range(10):
        seed = int(rng.integers(1, 1 << 30))
        p = A.default_params(good_enough_fitness=float(rng.choice([1.0, 0.8, 0.5])), match_algo=int(rng.choice([0, 0, 0, 1])))
        n, m = int(rng.integers(5, 200)), int(rng.integers(3, 60))
        attr = np.zeros((m, 2), dtype=np.uint32)
        attr[:, 0] = rng.integers(1, 4, m)
        attr[:, 1] = rng.integers(0, 3, m)
        offers = A.Offers(cpus=rng.integers(2, 12, m).astype(float), mem=rng.integers(2, 20, m) * 1000.0, attr=attr, k8s=np.ones(m, dtype=np.uint8),
                          max_tasks=np.full(m, int(rng.integers(2, 8)), np.int32), num_tasks=rng.integers(0, 4, m).astype(np.int32))
        group = rng.integers(0, 6, n).astype(np.uint32)
        group[rng.random(n) < 0.3] = A.NONE_U32
        jobs = A.Jobs(cpus=rng.integers(1, 4, n).astype(float), mem=rng.integers(1, 4, n) * 1000.0, group=group)
        keys = [A.NONE_U32, 0, A.NONE_U32, 1, 0, 0]
        rh = [[int(x) for x in rng.integers(0, m, int(rng.integers(0, 3)))] for _ in range(6)]
        ra = [[int(attr[h, k]) if k != A.NONE_U32 else 0 for h in hs] for hs, k in zip(rh, keys)]
        groups = A.Groups(type=np.array([1, 2, 2, 3, 3, 0], dtype=np.uint8), attr_key=np.array(keys, dtype=np.uint32),
                          minimum=np.array([0, int(rng.integers(0, 5)), 10, 0, 0, 0], dtype=np.int32), run_hosts=rh, run_attrs=ra)
        j2o = P.match_parity(make_engine, jobs, offers, groups, p)
        if (j2o < 0).any():
            P.explain_parity(make_engine, jobs, offers, groups, p, max_pos=5, tag=f"groups {seed}")
        jobs, offers, groups = P.slow_constraint_case(seed, int(rng.integers(5, 250)), int(rng.integers(12, 80)))
        j2o = P.match_parity(make_engine, jobs, offers, groups, p)
        if (j2o < 0).any():
            P.explain_parity(make_engine, jobs, offers, groups, p, max_pos=5, tag=f"slow {seed}")
        pool = synth.make_pool(seed=seed, n_pending=int(rng.integers(1, 500)), n_running=0, n_users=int(rng.integers(1, 30)),
                               n_offers=int(rng.integers(1, 90)), gpus=bool(rng.integers(0, 2)), constraints=bool(rng.integers(0, 2)),
                               fractional=bool(rng.integers(0, 2)))
        P.metrics_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, A.default_params(good_enough_fitness=1.0), n_users=30)
        trace, hosts = P.make_trace(seed, int(rng.integers(5, 60)), int(rng.integers(1, 5)), span_ms=int(rng.integers(60_000, 600_000)))
        P.replay_parity(make_engine, trace, hosts, TIGHT if rng.integers(0, 2) else CONFIG, min_matched=0)


# ---- ports, named scalars, several entries per host in the k8s "gpus" / "disk" maps (scheduler.clj:456-471, 177-189) -----------
def test_xres_known_answers(make_engine):
    P.xres_known_answers(make_engine)


@ALGOS
@pytest.mark.parametrize("kw", [
    dict(seed=71, n=500, m=60),
    dict(seed=72, n=500, m=90, groups=True, constraints=True),
    dict(seed=73, n=500, m=60, ports=False, scalars=3),
    dict(seed=74, n=500, m=90, slots=3, scalars=0, ports=False),
    dict(seed=75, n=500, m=90, slots=4, groups=True),
], ids=["ports+scalars", "with-groups-and-constraints", "three-scalars", "gpu-and-disk-maps", "everything"])
def test_match_ports_scalars_maps(make_engine, kw, algo):
    kw = dict(kw)
    jobs, offers, groups = P.xres_random_case(kw.pop("seed"), kw.pop("n"), kw.pop("m"), **kw)
    j2o = P.match_parity(make_engine, jobs, offers, groups, A.default_params(match_algo=algo))
    assert (j2o >= 0).sum() > 20 and (j2o < 0).sum() > 5


@pytest.mark.parametrize("ge", [0.8, 0.4])
def test_match_ports_scalars_good_enough_and_explain(make_engine, ge):
    jobs, offers, groups = P.xres_random_case(76, 500, 60, groups=True, slots=2)
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


def test_match_more_offer_chunks_than_lanes(make_engine):
    # more than 64 chunks of 128 offers: a merge lane folds several chunk lists (the first as it is, the rest by insertion)
    rng = np.random.default_rng(81)
    m, n = 8500, 160
    offers = A.Offers(cpus=rng.integers(1, 9, m).astype(float), mem=rng.integers(1, 9, m) * 1024.0)
    jobs = A.Jobs(cpus=rng.integers(1, 5, n).astype(float), mem=rng.integers(1, 5, n) * 1024.0)
    for ge in (1.0, 0.7):
        P.match_parity(make_engine, jobs, offers, None, A.default_params(good_enough_fitness=ge))


@pytest.mark.parametrize("split", [1, 2, 4])
def test_match_eval_offer_split_levels(make_engine, monkeypatch, split):
    # idle rows of the eval grid take shares of the offers (eval_split): every cap gives the oracle's placement
    monkeypatch.setenv("COOK_EVAL_SPLIT", str(split))
    pool = synth.make_pool(seed=83, n_pending=300, n_running=100, n_users=20, n_offers=260, gpus=True, constraints=True)
    for ge in (1.0, 0.6):
        P.match_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, A.default_params(good_enough_fitness=ge))


def test_guard_bands_report_a_write_past_a_buffer():
    """COOK_GUARD=1 (scripts/fuzz_sweep.py --guard): every device buffer sits between two bands of a pattern that are looked at when
    the buffer is freed.  A clean match reports nothing; the built-in self-test (one byte past the placement column at destroy) is
    reported on stderr — so a sweep that prints no COOK_GUARD line really had no write outside a buffer."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests.simt_emu import build_emu\n"
            "from cook_amd import _abi as A, synth\n"
            "from cook_amd.engine import Engine\n"
            "pool = synth.make_pool(seed=5, n_pending=200, n_running=50, n_users=9, n_offers=60, constraints=True, gpus=True)\n"
            "e = Engine(A.default_params(), lib_path=build_emu.build())\n"
            "e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)\n"
            "e.cycle_run(200)\n"
            "e.cycle_fetch()\n"
            "print('hits', e.match_stats()['guard_hits'])\n"
            "del e\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for selftest, want_line in (("", False), ("1", True)):
        env = dict(os.environ, COOK_GUARD="1")
        env.pop("COOK_GUARD_SELFTEST", None)
        if selftest:
            env["COOK_GUARD_SELFTEST"] = selftest
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "hits 0" in r.stdout
        assert ("COOK_GUARD: " in r.stderr) == want_line, r.stderr[-2000:]
        if want_line:
            assert "PAST its end" in r.stderr


# ---- the rank parts of several pools in one call (cook_cycle_run_rank_multi: pool batches, engine.hip) ------------------------------------
def test_rank_batch_diverging_flows(make_engine):
    stats = P.rank_batch_parity(make_engine, P.rank_batch_cases(), k=300, n_users=300, min_grouped=20)
    assert stats[-1]["rank_batch_single_ops"] <= stats[-1]["rank_batch_launches"], stats[-1]


def test_rank_batch_tie_rule_as_radix_passes(make_engine, monkeypatch):
    monkeypatch.setenv("COOK_RANK_RADIX", "1")
    P.rank_batch_parity(make_engine, P.rank_batch_cases()[1:5], k=10 ** 9)


def test_rank_batch_of_one_pool(make_engine):
    """a call for ONE engine is cook_cycle_run_rank (+ cook_rank_user_usage)"""
    P.rank_batch_parity(make_engine, P.rank_batch_cases()[2:3], k=200, n_users=40)


def test_rank_batch_one_flow_fails(make_engine):
    P.rank_batch_one_flow_fails(make_engine)


def test_rank_batch_switched_off_in_a_fresh_process():
    """COOK_RANK_BATCH=0 / COOK_BATCH_COPY_KERNEL=0 are read when the library is loaded: in a process of their own the multi call is a loop
    over its engines (no batch statistics), resp. the read-backs are recorded copies issued alone — same results either way."""
    import json
    import subprocess
    import sys
    code = r'''
import json, sys
sys.path.insert(0, %r)
import numpy as np
from cook_amd import _abi as A, synth
from cook_amd.engine import Engine, cycle_run_rank_multi, cycle_match_multi
from tests.simt_emu import build_emu
so = build_emu.build()
p = A.default_params(good_enough_fitness=1.0, match_algo=2)
pools = [synth.make_pool(seed=300 + i, n_pending=500 + 40 * i, n_running=200, n_users=20, n_offers=40, gpus=True, constraints=True) for i in range(3)]
engines = [Engine(p, lib_path=so) for _ in pools]
for e, pl in zip(engines, pools):
    e.cycle_stage(pl.tasks, pl.users, pl.pending_jobs, pl.offers, pl.groups)
cycle_run_rank_multi(engines, [10 ** 9, 200, 10 ** 9])
st = engines[0].match_stats()
cycle_match_multi(engines)
out = [[a.tolist() for a in e.cycle_fetch()[:2]] for e in engines]
print(json.dumps({"stats": {k: v for k, v in st.items() if k.startswith("rank_batch")}, "out": out}))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for name, env in (("batch", {}), ("off", {"COOK_RANK_BATCH": "0"}), ("copies", {"COOK_BATCH_COPY_KERNEL": "0"})):
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["batch"]["stats"]["rank_batch_pools"] == 3 and got["batch"]["stats"]["rank_batch_single_ops"] == 0
    assert got["off"]["stats"]["rank_batch_pools"] == 0
    assert got["copies"]["stats"]["rank_batch_pools"] == 3 and got["copies"]["stats"]["rank_batch_single_ops"] > 0
    assert got["batch"]["out"] == got["off"]["out"] == got["copies"]["out"]


@pytest.mark.parametrize("case", [c for c in G_EXPLAIN if "engine" in c], ids=[c["name"] for c in G_EXPLAIN if "engine" in c])
def test_explain_reference_cases(make_engine, case):
    # the reference's own cases of the why-unscheduled reducer (tests/golden/explain.json) through cook_match / cook_match_explain
    P.explain_golden_engine(make_engine, case)


def test_pool_usage_multi(make_engine):
    P.pool_usage_multi_parity(make_engine)


def test_engine_choice_of_the_placement_form_in_fresh_processes():
    """match_algo 0 with six engines on the device: class-ordered best fit for the pools that allow it; COOK_CLASSFIT=0 (read once per process) keeps the
    window rounds, COOK_CLASSFIT=1 takes the class-ordered form for ONE engine too — same assignments every way."""
    import json
    import subprocess
    import sys
    code = r'''
import json, sys
sys.path.insert(0, %r)
from cook_amd import _abi as A, synth
from cook_amd.engine import Engine, cycle_run_rank_multi, cycle_match_multi
from tests.simt_emu import build_emu
so = build_emu.build()
n = int(sys.argv[1])
p = A.default_params(good_enough_fitness=1.0)
pools = [synth.make_pool(seed=500 + i, n_pending=300 + 40 * i, n_running=100, n_users=20, n_offers=60 + 10 * i, gpus=(i %% 2 == 0), constraints=True) for i in range(n)]
engines = [Engine(p, lib_path=so) for _ in pools]
for e, pl in zip(engines, pools):
    e.cycle_stage(pl.tasks, pl.users, pl.pending_jobs, pl.offers, pl.groups)
cycle_run_rank_multi(engines, 10 ** 9)
cycle_match_multi(engines)
print(json.dumps({"forms": [e.match_stats()["placement_form"] for e in engines], "out": [e.cycle_fetch()[1].tolist() for e in engines]}))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for name, n, env in (("six", 6, {}), ("six-forbidden", 6, {"COOK_CLASSFIT": "0"}), ("five", 5, {}), ("one", 1, {}), ("one-forced", 1, {"COOK_CLASSFIT": "1"})):
        envx = {k: v for k, v in os.environ.items() if k != "COOK_CLASSFIT"}
        r = subprocess.run([sys.executable, "-c", code, str(n)], env={**envx, **env}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["six"]["forms"].count(3) >= 3 and set(got["six"]["forms"]) <= {0, 3}, got["six"]["forms"]
    assert set(got["six-forbidden"]["forms"]) == {0} and set(got["five"]["forms"]) == {0} and got["one"]["forms"] == [0] and got["one-forced"]["forms"] == [3]
    assert got["six"]["out"] == got["six-forbidden"]["out"] and got["five"]["out"] == got["six"]["out"][:5] and got["one"]["out"] == got["one-forced"]["out"] == got["six"]["out"][:1]

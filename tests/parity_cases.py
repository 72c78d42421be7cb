"""Parity checks shared by the CPU-emulated build (tests/test_parity_emu.py) and the real HIP build
(tests/test_parity_gpu.py, -m gpu): the engine behind the C ABI vs the oracle on the same seeded inputs,
and vs the reference's own golden vectors.  Bit-exact: ranked order (uint32), DRU (fp64 ==), assignments (int32)."""
import dataclasses

import numpy as np

from cook_amd import _abi as A
from cook_amd import synth
from oracle import pyoracle
from tests import golden_util as G


def check_rank_golden(make_engine):
    for case in G.load("rank"):
        tasks, users, names, _ = G.build_rank_inputs(case["jobs"], case["shares"], case.get("quotas"))
        p = A.default_params(dru_mode=case.get("dru_mode", 0), max_over_quota_jobs=case.get("max_over_quota_jobs", 100))
        with make_engine(p) as e:
            ranked, dru = e.rank(tasks, users)
        o_ranked, o_dru = pyoracle.rank(p, tasks, users)
        assert list(ranked) == list(o_ranked), case["name"]
        assert np.array_equal(dru, o_dru, equal_nan=True), case["name"]
        if "expect_ranked" in case:
            assert [names[i] for i in ranked] == case["expect_ranked"], (case["name"], case["ref"])


def check_rank_group_golden(make_engine):
    for case in G.load("rank_group"):
        built, usages = {}, {}
        for pool, spec in case["pools"].items():
            tasks, users, names, _ = G.build_rank_inputs(spec["jobs"], case["shares"])
            built[pool] = (tasks, users, names)
            with make_engine(A.default_params()) as e:
                e.rank_stage(tasks, users)
                usages[pool] = e.rank_pool_usage()  # device reduction of the pool's running usage
            assert usages[pool].as_tuple() == pyoracle.pool_usage(tasks).as_tuple()
        gu = A.usage(*[sum(getattr(u, f) for u in usages.values()) for f in ("count", "cpus", "mem", "gpus")])
        for pool, spec in case["pools"].items():
            tasks, users, names = built[pool]
            q = A.pool_quota(pool_quota=G.usage_of(spec["quota"]), group_quota=G.usage_of(case["group_quota"]), group_usage=gu)
            with make_engine(A.default_params()) as e:
                ranked, _ = e.rank(tasks, users, quota=q)
            assert [names[i] for i in ranked] == spec["expect_ranked"], (case["name"], pool)


def check_match_golden(make_engine):
    for case in G.load("match"):
        J, O, names, x = G.build_match_all(case)
        p = A.default_params(good_enough_fitness=case["good_enough"], **x["params"])
        with make_engine(p) as e:
            j2o, fail, head = e.match(J, O, x["groups"], x["reserved"])
        o_j2o, o_fail, o_head = pyoracle.match(p, J, O, x["groups"], x["reserved"])
        assert np.array_equal(j2o, o_j2o), case["name"]
        assert np.array_equal(fail, o_fail), case["name"]
        assert head == o_head, case["name"]
        G.check_match_expectations(case, names, x["host_names"], j2o, head)
        # The reference's expectations were produced by a Fenzo that ALSO saw what Cook puts on every request and lease: ports 0 and
        # the named scalars "cpus" / "mem" on the jobs (scheduler.clj:177-189, 466), the test offers' 1001-port range
        # (31000-32000, e.g. test/cook/test/scheduler/constraints.clj:69) and their scalar resources (offer.clj:57-65).  With them
        # spelled out the outcome must not move.
        if J.n and O.n:
            J2 = dataclasses.replace(J, ports=np.zeros(J.n, np.int32), scalars=np.stack([J.cpus, J.mem], axis=1))
            O2 = dataclasses.replace(O, ports=np.full(O.n, 1001, np.int32), scalars=np.stack([O.cpus, O.mem], axis=1))
            with make_engine(p) as e:
                j2o_x, fail_x, head_x = e.match(J2, O2, x["groups"], x["reserved"])
            assert np.array_equal(j2o_x, j2o) and np.array_equal(fail_x, fail) and head_x == head, case["name"]
            assert np.array_equal(pyoracle.match(p, J2, O2, x["groups"], x["reserved"])[0], o_j2o), case["name"]


def rank_parity(make_engine, pool: synth.Pool, params, quota=None):
    with make_engine(params) as e:
        ranked, dru = e.rank(pool.tasks, pool.users, quota=quota)
    o_ranked, o_dru = pyoracle.rank(params, pool.tasks, pool.users, quota=quota)
    assert len(ranked) == len(o_ranked)
    bad = np.nonzero(ranked != o_ranked)[0]
    assert len(bad) == 0, f"rank order differs first at {bad[:5]} of {len(ranked)}"
    assert np.array_equal(dru, o_dru, equal_nan=True)
    return ranked


def tie_rule_forms(make_engine, monkeypatch, n_users=400, per_user=6):
    """The sorted-merge tie rule in both of the engine's forms (cook_amd/csrc/tile_sort.hpp: tie groups sorted in LDS tiles; engine.hip
    tie_refine_radix: radix passes over the tied items), on the same pools.  Many users with the SAME tasks in the same order make tie
    groups of n_users items that stay tied until the users' name order decides (deep doubling rounds); with the emulated suite's small
    tiles such a group does not fit one and the call has to fall back on its own."""
    pool = synth.make_pool(seed=81, n_pending=n_users * per_user, n_running=0, n_users=n_users, n_offers=4, tie_heavy=True)
    t = pool.tasks
    t.user[:] = np.arange(t.n) % n_users
    order = np.argsort(t.user, kind="stable")
    within = np.zeros(t.n, dtype=np.int64)
    for u in range(n_users):
        idx = order[t.user[order] == u]
        within[idx] = np.arange(len(idx))
    t.cpus[:] = 1.0 + (within % 3)
    t.mem[:] = 512.0 * (1 + (within % 2))
    t.priority[:] = 50
    pool.users.div_cpus[:] = 40.0  # equal shares: the k-th tasks of all users carry the same DRU
    pool.users.div_mem[:] = 16384.0
    params = A.default_params(max_over_quota_jobs=10_000)
    a = rank_parity(make_engine, pool, params)
    monkeypatch.setenv("COOK_RANK_RADIX", "1")
    b = rank_parity(make_engine, pool, params)
    monkeypatch.delenv("COOK_RANK_RADIX")
    assert np.array_equal(a, b)
    # an ordinary tie-heavy pool (short groups: the tiles take it) through both forms as well
    pool = synth.make_pool(seed=82, n_pending=1200, n_running=400, n_users=60, n_offers=4, tie_heavy=True)
    a = rank_parity(make_engine, pool, A.default_params())
    monkeypatch.setenv("COOK_RANK_RADIX", "1")
    b = rank_parity(make_engine, pool, A.default_params())
    monkeypatch.delenv("COOK_RANK_RADIX")
    assert np.array_equal(a, b)


def equal_dru_run_cases(make_engine):
    """Users with runs of EQUAL consecutive DRUs (rank_run refused such pools).  The literal merge (dru.clj:92-94)
    re-conses the emitting user at the front, so a run is emitted back to back; the oracle's literal and heap forms agree."""
    # (a) zero-resource and absorbed (1e-17) requests in the default mode, tie-heavy so that runs sit inside big tie groups
    pool = synth.make_pool(seed=71, n_pending=900, n_running=300, n_users=25, n_offers=10, tie_heavy=True)
    rng = np.random.default_rng(5)
    z = rng.random(pool.tasks.n) < 0.15
    pool.tasks.cpus[z] = 1e-17
    pool.tasks.mem[z] = 1e-17
    z2 = rng.random(pool.tasks.n) < 0.05
    pool.tasks.cpus[z2] = 0.0
    pool.tasks.mem[z2] = 0.0
    rank_parity(make_engine, pool, A.default_params(max_over_quota_jobs=10))
    o_lit, _ = pyoracle.rank(A.default_params(max_over_quota_jobs=10), pool.tasks, pool.users, literal_merge=True)
    o_heap, _ = pyoracle.rank(A.default_params(max_over_quota_jobs=10), pool.tasks, pool.users)
    assert np.array_equal(o_lit, o_heap)
    # (b) gpu mode with gpu-less tasks: cumulative gpus stay put over long runs, incl. runs at a user's head (DRU 0)
    pool = synth.make_pool(seed=72, n_pending=1500, n_running=500, n_users=40, n_offers=10, gpus=True)
    rank_parity(make_engine, pool, A.default_params(dru_mode=1))
    # (c) runs cut by the over-quota limiter, and a quota filter behind them
    pool = synth.make_pool(seed=73, n_pending=700, n_running=200, n_users=9, n_offers=10, tie_heavy=True, quota_frac=0.5)
    z = np.random.default_rng(6).random(pool.tasks.n) < 0.3
    pool.tasks.cpus[z] = 0.0
    pool.tasks.mem[z] = 0.0
    q = A.pool_quota(pool_quota=A.quota(count=500, cpus=900.0))
    rank_parity(make_engine, pool, A.default_params(max_over_quota_jobs=3), quota=q)
    # (d) everything equal: one user, all-zero tasks; and two users with nothing but zeros
    for n_users in (1, 2):
        pool = synth.make_pool(seed=74, n_pending=130, n_running=70, n_users=n_users, n_offers=10)
        pool.tasks.cpus[:] = 0.0
        pool.tasks.mem[:] = 0.0
        rank_parity(make_engine, pool, A.default_params())


def user_usage_parity(make_engine, pool: synth.Pool, n_users):
    with make_engine(A.default_params()) as e:
        e.rank(pool.tasks, pool.users)
        got = e.rank_user_usage(n_users)
    want = pyoracle.user_usage(pool.tasks, n_users)
    assert np.array_equal(got, want), np.nonzero((got != want).any(axis=1))[0][:5]
    return got


def match_parity(make_engine, jobs, offers, groups, params, reserved=()):
    with make_engine(params) as e:
        j2o, fail, head = e.match(jobs, offers, groups, reserved)
    o_j2o, o_fail, o_head = pyoracle.match(params, jobs, offers, groups, reserved)
    bad = np.nonzero(j2o != o_j2o)[0]
    assert len(bad) == 0, f"assignment differs first at job {bad[:5]}: {j2o[bad[:5]]} vs {o_j2o[bad[:5]]}"
    assert np.array_equal(fail, o_fail)
    assert head == o_head
    return j2o


def cycle_parity(make_engine, pool: synth.Pool, params, k):
    with make_engine(params) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e.cycle_run(k)
        ranked, j2o, head = e.cycle_fetch()
    o_ranked, _ = pyoracle.rank(params, pool.tasks, pool.users)
    assert np.array_equal(ranked, o_ranked)
    pend_ord = np.cumsum(pool.tasks.pending) - 1
    considerable = pool.pending_jobs.take(pend_ord[o_ranked[:k]])
    o_j2o, _, o_head = pyoracle.match(params, considerable, pool.offers, pool.groups)
    assert np.array_equal(j2o, o_j2o)
    assert head == o_head
    return ranked, j2o


def pinned_jobs_case(seed, n_jobs, n_offers, cardinality):
    """Jobs pinned by an EQUALS constraint (constraints.clj:356-377) to one value of an attribute with `cardinality`
    values (0 = host-unique): spreads the candidates of a window over many distinct offers, which is what fills the
    placement kernel's candidate-slot table (cardinality ~ 64) or its touched set (host-unique)."""
    rng = np.random.default_rng(seed)
    attr = np.zeros((n_offers, 1), dtype=np.uint32)
    attr[:, 0] = (np.arange(n_offers) + 1) if cardinality == 0 else rng.integers(1, cardinality + 1, n_offers)
    offers = A.Offers(cpus=rng.integers(4, 17, n_offers).astype(float), mem=rng.integers(4, 17, n_offers) * 4096.0, attr=attr,
                      k8s=np.ones(n_offers, dtype=np.uint8))
    hi = n_offers if cardinality == 0 else cardinality
    equals = [[(0, int(rng.integers(1, hi + 1)))] for _ in range(n_jobs)]
    jobs = A.Jobs.with_constraints(rng.integers(1, 4, n_jobs).astype(float), rng.integers(1, 4, n_jobs) * 1024.0,
                                   equals=equals, novel=[[] for _ in range(n_jobs)])
    return jobs, offers


# ---- rebalancer ------------------------------------------------------------------------------------------------------
def _rebal_equal(got, want, tag):
    assert len(got["decisions"]) == len(want["decisions"]), (tag, len(got["decisions"]), len(want["decisions"]))
    for i, (d, o) in enumerate(zip(got["decisions"], want["decisions"])):
        assert d == o, (tag, i, d, o)  # host, dru, resources (fp64 ==), preempted task lists: all exact
    assert np.array_equal(got["pending_dru"], want["pending_dru"], equal_nan=True), tag


def check_rebalance_golden(make_engine):
    """The reference's own rebalancer vectors (tests/golden/rebalance.json) through the C ABI.  Cases that drive the oracle's
    test hooks (a forced decision, a State that already holds preempted tasks) have no ABI equivalent and are skipped."""
    from tests.test_oracle_golden import check_rebalance_case
    n = 0
    for case in G.load("rebalance"):
        if "forced" in case or case.get("init_preempted_hosts"):
            continue
        b = G.build_rebalance_inputs(case)
        with make_engine(b["params"]) as e:
            res = e.rebalance(b["running"], b["pending"], b["pending_job_id"], b["pending_priority"], b["users"], b["spare"],
                              b["rparams"], host_attrs=b["host_attrs"], groups=b["groups"], attrs_cached=b["slave_known"])
        check_rebalance_case(case, res, b)
        want = pyoracle.rebalance(b["params"], b["running"], b["pending"], b["pending_job_id"], b["pending_priority"], b["users"],
                                  b["spare"], b["rparams"], host_attrs=b["host_attrs"], groups=b["groups"],
                                  slave_known=b["slave_known"])
        _rebal_equal(res, want, case["name"])
        n += 1
    assert n >= 25


def make_rebalance_case(seed, n_running, n_pending, n_users, n_hosts, *, fractional=False, constraints=False, gpus=False,
                        spare_frac=0.3, quota_frac=0.1, max_preemption=64, min_dru_diff=0.05, safe_dru=0.0, dru_mode=0, gpu_slots=1):
    """Random pool for cook_rebalance in the shape of the reference's stress generator (test/cook/test/rebalancer.clj:1152-1187)
    and BASELINE.json C5: running tasks spread over hosts, pending jobs of the same users, some spare capacity."""
    rng = np.random.default_rng(seed)
    R, P = n_running, n_pending
    n = R + P
    cpus = np.clip(np.rint(rng.normal(3.0, 1.0, n)), 1, 8)
    mem = np.clip(np.rint(rng.normal(10240.0, 4096.0, n)), 512, 65536)
    if fractional:
        cpus = cpus + rng.integers(0, 10, n) / 10.0
        mem = mem + rng.integers(0, 10, n) / 10.0
    g = np.zeros(n)
    gmodel = np.zeros(n, dtype=np.uint32)
    if gpus:
        has = rng.random(n) < 0.15
        g[has] = rng.choice([1.0, 2.0, 4.0], size=int(has.sum()))
        gmodel[has] = rng.choice([1, 2], size=int(has.sum()))
    if dru_mode == 1:  # gpu pools: every running task holds gpus; the pending jobs keep theirs (0 without `gpus`)
        g[:R] = np.maximum(g[:R], 1.0)
    p = 1.0 / np.arange(1, n_users + 1) ** 1.1
    user = rng.permutation(n_users)[rng.choice(n_users, size=n, p=p / p.sum())].astype(np.uint32)
    prio = rng.integers(0, 101, n).astype(np.int32)
    base = 17_592_186_044_416
    running = A.Tasks(cpus=cpus[:R], mem=mem[:R], gpus=g[:R], user=user[:R], priority=prio[:R],
                      start_ms=(1_600_000_000_000 + rng.integers(0, 86_400_000, R)).astype(np.int64),
                      task_id=(base + 2_000_000_000 + rng.permutation(R)).astype(np.int64),
                      job_id=(base + rng.permutation(R)).astype(np.int64), pending=np.zeros(R, dtype=np.uint8),
                      host=rng.integers(0, n_hosts, R).astype(np.uint32))
    div_c, div_m, div_g = np.full(n_users, 64.0), np.full(n_users, 262144.0), np.full(n_users, 8.0)
    big = rng.random(n_users) < 0.1
    div_c[big] *= 4
    div_m[big] *= 4
    qcount = np.full(n_users, 2.0 ** 31 - 1)
    qcount[rng.random(n_users) < quota_frac] = float(max(2, R // max(1, n_users)))
    users = A.Users(div_cpus=div_c, div_mem=div_m, div_gpus=div_g, quota_count=qcount)
    kw = {}
    host_attrs, groups = None, None
    if constraints:
        n_keys = 4
        card = [2, 3, 8, 0]
        attr = np.zeros((n_hosts, n_keys), dtype=np.uint32)
        for k, c in enumerate(card):
            attr[:, k] = (np.arange(n_hosts) + 1) if c == 0 else rng.integers(0, c + 1, n_hosts)  # 0 = absent on some hosts
        cached = rng.random(n_hosts) < 0.9  # some hosts are not in the agent-attributes-cache
        rows = np.nonzero(cached)[0]
        o_gm = np.zeros(n_hosts, dtype=np.uint32)
        o_gc = np.zeros(n_hosts)
        if gpus:
            gh = rng.random(n_hosts) < 0.3
            o_gm[gh] = rng.choice([1, 2], size=int(gh.sum()))
            o_gc[gh] = rng.choice([1.0, 2.0, 4.0], size=int(gh.sum()))
        if gpu_slots > 1:  # hosts whose "gpus" map holds several models (constraints.clj:136-142 reads it per model)
            o_gm = np.stack([o_gm] + [np.where((o_gm > 0) & (rng.random(n_hosts) < 0.5), 3 - o_gm.astype(np.int64), 0).astype(np.uint32)
                                      for _ in range(gpu_slots - 1)], axis=1)
            o_gm[:, 2:] = 0
            o_gc = np.stack([o_gc] + [rng.choice([1.0, 2.0, 4.0], size=n_hosts) for _ in range(gpu_slots - 1)], axis=1)
        host_attrs = A.Offers(cpus=np.zeros(len(rows)), mem=np.zeros(len(rows)), host=rows.astype(np.uint32),
                              k8s=(rng.random(len(rows)) < 0.8).astype(np.uint8), gpu_model=o_gm[rows], gpu_count=o_gc[rows],
                              attr=attr[rows])
        equals = [[(int(rng.integers(0, 3)), int(rng.integers(1, 4)))] if rng.random() < 0.2 else [] for _ in range(P)]
        novel = [[int(h) for h in rng.integers(0, n_hosts, 2)] if rng.random() < 0.1 else [] for _ in range(P)]
        grp = np.full(P, A.NONE_U32, dtype=np.uint32)
        n_g = 6
        in_g = rng.random(P) < 0.3
        grp[in_g] = rng.integers(0, n_g, int(in_g.sum()))
        groups = A.Groups(type=np.array([1, 2, 2, 3, 3, 0], dtype=np.uint8),
                          attr_key=np.array([A.NONE_U32, 0, 2, 1, 0, 0], dtype=np.uint32),
                          minimum=np.array([0, 2, 10, 0, 0, 0], dtype=np.int32),
                          run_hosts=[[int(h) for h in rng.integers(0, n_hosts, int(rng.integers(0, 4)))] for _ in range(n_g)])
        kw.update(equals=equals, novel=novel, group=grp)
    jobs = A.Jobs.with_constraints(cpus[R:], mem[R:], gpus=g[R:], gpu_model=gmodel[R:], user=user[R:], **kw)
    sh = np.nonzero(rng.random(n_hosts) < spare_frac)[0].astype(np.uint32)
    spare = A.HostSpare(host=sh, cpus=rng.integers(0, 6, len(sh)).astype(np.float64) + (0.5 if fractional else 0.0),
                        mem=rng.integers(0, 16, len(sh)) * 1024.0, gpus=rng.integers(0, 3, len(sh)).astype(np.float64))
    rp = A.CookRebalanceParams(safe_dru, min_dru_diff, max_preemption, 0)
    return dict(params=A.default_params(dru_mode=dru_mode), running=running, pending=jobs,
                pending_job_id=(base + 10_000_000 + rng.permutation(P)).astype(np.int64), pending_priority=prio[R:], users=users,
                spare=spare, rparams=rp, host_attrs=host_attrs, groups=groups)


def rebalance_parity(make_engine, b, min_decisions=0):
    with make_engine(b["params"]) as e:
        e.rebalance_stage(b["running"], b["pending"], b["pending_job_id"], b["pending_priority"], b["users"], b["spare"],
                          b["rparams"], host_attrs=b["host_attrs"], groups=b["groups"])
        e.rebalance_run()
        got = e.rebalance_fetch()
        e.rebalance_run()  # a run must be repeatable on the same staged inputs (it updates spare resources internally)
        again = e.rebalance_fetch()
    want = pyoracle.rebalance(b["params"], b["running"], b["pending"], b["pending_job_id"], b["pending_priority"], b["users"],
                              b["spare"], b["rparams"], host_attrs=b["host_attrs"], groups=b["groups"])
    _rebal_equal(got, want, "random")
    _rebal_equal(again, want, "random, second run")
    assert len(got["decisions"]) >= min_decisions, len(got["decisions"])
    return got


def rebalance_paths(make_engine):
    """Which re-scoring path ran (cook_kernel_timings): integer-valued resources -> one launch from the flipped slots
    (rebal_rs_delta, the next job prepared inside rebal_apply); a fractional resource -> tile scans + left-to-right redo; and the
    general path forced on the integer inputs gives the same decisions."""
    import os

    def run(b):
        with make_engine(b["params"]) as e:
            e.set_profiling(True)
            e.rebalance_stage(b["running"], b["pending"], b["pending_job_id"], b["pending_priority"], b["users"], b["spare"],
                              b["rparams"], host_attrs=b["host_attrs"], groups=b["groups"])
            e.rebalance_run()
            got = e.rebalance_fetch()
            names = set(e.kernel_timings())
        want = pyoracle.rebalance(b["params"], b["running"], b["pending"], b["pending_job_id"], b["pending_priority"], b["users"],
                                  b["spare"], b["rparams"], host_attrs=b["host_attrs"], groups=b["groups"])
        _rebal_equal(got, want, "paths")
        assert len(got["decisions"]) >= 3
        return names
    for kw in (dict(seed=57, n_running=3600, n_pending=10, n_users=2, n_hosts=90),
               dict(seed=54, n_running=500, n_pending=40, n_users=15, n_hosts=40, constraints=True, gpus=True)):
        integer = make_rebalance_case(**kw)
        names = run(integer)
        assert "rebal_rs_delta" in names and "rebal_rs_local" not in names, names
        os.environ["COOK_REBAL_GENERAL"] = "1"
        try:
            names = run(integer)
        finally:
            del os.environ["COOK_REBAL_GENERAL"]
        assert "rebal_rs_local" in names and "rebal_rs_delta" not in names, names
    names = run(make_rebalance_case(seed=58, n_running=2600, n_pending=8, n_users=2, n_hosts=70, fractional=True))
    assert "rebal_rs_local" in names and "rebal_rs_delta" not in names, names


# ---- considerable jobs ------------------------------------------------------------------------------------------------
def check_considerable_golden(make_engine):
    from tests.test_oracle_golden import check_considerable_case
    for case in G.load("considerable"):
        queue, st, names, unames = G.build_considerable_inputs(case)
        with make_engine(A.default_params()) as e:
            idx, rl, ps = e.considerable(queue, st, case["num_considerable"])
        check_considerable_case(case, idx, rl, names, unames)
        o_idx, o_rl, o_ps = pyoracle.considerable(queue, st, case["num_considerable"])
        assert np.array_equal(idx, o_idx) and np.array_equal(rl, o_rl) and np.array_equal(ps, o_ps), case["name"]


def make_considerable_case(seed, n, n_users, *, fractional=False, tokens=True, enforce=True, pool_quota=True, eligible=True):
    rng = np.random.default_rng(seed)
    cpus = np.clip(np.rint(rng.normal(3.0, 1.0, n)), 1, 8)
    mem = np.clip(np.rint(rng.normal(10240.0, 4096.0, n)), 512, 65536)
    if fractional:
        cpus = cpus + rng.integers(0, 10, n) / 10.0
        mem = mem + rng.integers(0, 10, n) / 10.0
    gpus = np.where(rng.random(n) < 0.1, rng.choice([1.0, 2.0, 4.0], size=n), 0.0)
    p = 1.0 / np.arange(1, n_users + 1) ** 1.1
    user = rng.permutation(n_users)[rng.choice(n_users, size=n, p=p / p.sum())].astype(np.uint32)
    queue = A.Queue(cpus=cpus, mem=mem, gpus=gpus, user=user,
                    eligible=(rng.random(n) < 0.9).astype(np.uint8) if eligible else None)
    per_user = max(1, n // n_users)
    ucount = rng.integers(0, 20, n_users).astype(np.float64)
    ucpus = ucount * 3.0 + (0.1 if fractional else 0.0)
    umem = ucount * 10240.0
    ugpus = np.zeros(n_users)
    qcount = np.where(rng.random(n_users) < 0.5, ucount + rng.integers(0, 3 * per_user + 2, n_users), 2.0 ** 31 - 1)
    qcpus = np.where(rng.random(n_users) < 0.3, ucpus + rng.integers(0, 10 * per_user + 5, n_users), A.DMAX)
    qmem = np.full(n_users, A.DMAX)
    qgpus = np.where(rng.random(n_users) < 0.2, 2.0, A.DMAX)
    st = A.UserState(quota_count=qcount, quota_cpus=qcpus, quota_mem=qmem, quota_gpus=qgpus, usage_count=ucount, usage_cpus=ucpus,
                     usage_mem=umem, usage_gpus=ugpus,
                     tokens_left=rng.integers(0, 2 * per_user + 3, n_users).astype(np.int64) if tokens else None,
                     enforce_rate_limit=enforce,
                     pool_quota=A.quota(count=float(ucount.sum() + n // 3), cpus=float(ucpus.sum() + n)) if pool_quota else None)
    return queue, st


def considerable_parity(make_engine, queue, st, k):
    with make_engine(A.default_params()) as e:
        idx, rl, ps = e.considerable(queue, st, k)
    o_idx, o_rl, o_ps = pyoracle.considerable(queue, st, k)
    assert np.array_equal(idx, o_idx), (len(idx), len(o_idx))
    assert np.array_equal(rl, o_rl) and np.array_equal(ps, o_ps)
    return idx


def cycle_considerable_parity(make_engine, pool: synth.Pool, params, k, st, eligible_by_pending):
    """rank -> considerable filters -> match on the device vs the same three steps through the oracle."""
    with make_engine(params) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e.cycle_set_considerable(st, eligible_by_pending)
        e.cycle_run(k)
        ranked, j2o, head = e.cycle_fetch()
        pos = e.cycle_fetch_considerable()
    o_ranked, _ = pyoracle.rank(params, pool.tasks, pool.users)
    assert np.array_equal(ranked, o_ranked)
    pend_ord = np.cumsum(pool.tasks.pending) - 1
    jobs_of_queue = pend_ord[o_ranked]
    J = pool.pending_jobs
    queue = A.Queue(cpus=J.cpus[jobs_of_queue], mem=J.mem[jobs_of_queue], gpus=J.gpus[jobs_of_queue] if J.gpus is not None else None,
                    user=J.user[jobs_of_queue],
                    eligible=np.asarray(eligible_by_pending, dtype=np.uint8)[jobs_of_queue] if eligible_by_pending is not None else None)
    o_pos, _, _ = pyoracle.considerable(queue, st, k)
    assert np.array_equal(pos, o_pos), (len(pos), len(o_pos))
    o_j2o, _, o_head = pyoracle.match(params, J.take(jobs_of_queue[o_pos]), pool.offers, pool.groups)
    assert np.array_equal(j2o, o_j2o) and head == o_head
    return pos, j2o


def slow_constraint_case(seed, n, m):
    """Jobs whose constraints overflow the eval loop's register/LDS fast paths: more than 4 EQUALS pairs, attribute keys
    beyond the 8 staged in LDS, more than 4 novel hosts, unique groups with more than 8 hosts to avoid."""
    rng = np.random.default_rng(seed)
    n_keys = 10
    attr = rng.integers(1, 3, (m, n_keys)).astype(np.uint32)
    attr[:, 9] = rng.integers(0, 3, m)  # key 9 absent on some hosts
    offers = A.Offers(cpus=rng.integers(4, 17, m).astype(float), mem=rng.integers(8, 33, m) * 1024.0, attr=attr,
                      k8s=np.ones(m, dtype=np.uint8))
    equals, novel = [], []
    for _ in range(n):
        r = rng.random()
        if r < 0.25:
            ks = rng.choice(n_keys, size=int(rng.integers(5, 7)), replace=False)      # > 4 pairs
            equals.append([(int(k), int(rng.integers(1, 3))) for k in ks[:2]] + [(int(k), int(attr[rng.integers(0, m), k])) for k in ks[2:]])
        elif r < 0.5:
            equals.append([(int(rng.integers(8, 12)), int(rng.integers(0, 3)))])        # keys 8, 9 (beyond LDS), 10, 11 (beyond the table)
        elif r < 0.7:
            equals.append([(int(rng.integers(0, 8)), int(rng.integers(1, 3)))])
        else:
            equals.append([])
        novel.append([int(h) for h in rng.integers(0, m, int(rng.integers(5, 8)))] if rng.random() < 0.3 else [])
    group = np.full(n, A.NONE_U32, dtype=np.uint32)
    members = rng.permutation(n)[: n // 3]
    group[members] = 0
    jobs = A.Jobs.with_constraints(rng.integers(1, 4, n).astype(float), rng.integers(1, 5, n) * 1024.0, equals=equals, novel=novel,
                                   group=group)
    groups = A.Groups(type=np.array([1], dtype=np.uint8), run_hosts=[[int(h) for h in rng.choice(m, 10, replace=False)]])
    return jobs, offers, groups


def multi_pool_parity(make_engine, pools, params, k, rank_batched=True):
    """The rank part of every pool — ONE cook_cycle_run_rank_multi for all of them (rank_batched: the pools' flows side by side, the same
    kernel of several pools in one launch) or cook_cycle_run_rank per pool — + ONE cook_cycle_match_multi for all of them
    == cook_cycle_run on each pool == oracle."""
    from cook_amd.engine import cycle_match_multi, cycle_run_rank_multi
    engines = [make_engine(params) for _ in pools]
    try:
        for e, pool in zip(engines, pools):
            e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
            if not rank_batched:
                e.cycle_run_rank(k)
        if rank_batched:
            cycle_run_rank_multi(engines, k)
            st = engines[0].match_stats()
            assert st["rank_batch_pools"] == len(engines) or len(engines) == 1, st
        cycle_match_multi(engines)
        got = [e.cycle_fetch() for e in engines]
        cycle_match_multi(engines)  # nothing deferred any more: a no-op, not an error
    finally:
        for e in engines:
            e.close()
    for (ranked, j2o, head), pool in zip(got, pools):
        o_ranked, _ = pyoracle.rank(params, pool.tasks, pool.users)
        assert np.array_equal(ranked, o_ranked)
        pend_ord = np.cumsum(pool.tasks.pending) - 1
        kk = min(k, len(o_ranked))
        o_j2o, _, o_head = pyoracle.match(params, pool.pending_jobs.take(pend_ord[o_ranked[:kk]]), pool.offers, pool.groups)
        assert np.array_equal(j2o, o_j2o) and head == o_head
    return got


def rank_batch_parity(make_engine, cases, k, n_users=0, min_grouped=1):
    """ONE cook_cycle_run_rank_multi over pools whose rank flows DIFFER (sizes, tie rounds, equal-DRU runs, quotas, considerable filters,
    DRU mode ...) + one cook_cycle_match_multi == cook_cycle_run on a fresh engine per pool (ranked order, considerable positions,
    placement, per-user usage) — and the rank order == oracle.  cases: dicts with pool, params, and optionally quota, cons = (user state,
    eligible mask by pending job)."""
    from cook_amd.engine import cycle_match_multi, cycle_run_rank_multi
    want = []
    for c in cases:
        with make_engine(c["params"]) as e:
            pool = c["pool"]
            e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
            if c.get("cons"):
                e.cycle_set_considerable(*c["cons"])
            e.rank_set_quota(c.get("quota"))
            e.cycle_run(k)
            uu = e.rank_user_usage(n_users) if n_users else None
            want.append((e.cycle_fetch(), e.cycle_fetch_considerable(), uu))
    engines = [make_engine(c["params"]) for c in cases]
    try:
        for e, c in zip(engines, cases):
            pool = c["pool"]
            e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
            if c.get("cons"):
                e.cycle_set_considerable(*c["cons"])
            e.rank_set_quota(c.get("quota"))
        stats = []
        for _ in range(2):  # (the second call finds every buffer at its size: no flow has to wait for a reallocation)
            uus = cycle_run_rank_multi(engines, k, n_users=n_users)
            stats.append(engines[0].match_stats())
            cycle_match_multi(engines)
            got = [(e.cycle_fetch(), e.cycle_fetch_considerable(), uu) for e, uu in zip(engines, uus or [None] * len(engines))]
            for i, ((g_f, g_pos, g_uu), (w_f, w_pos, w_uu), c) in enumerate(zip(got, want, cases)):
                assert np.array_equal(g_f[0], w_f[0]), ("ranked", i)
                assert np.array_equal(g_pos, w_pos), ("considerable positions", i)
                assert np.array_equal(g_f[1], w_f[1]) and g_f[2] == w_f[2], ("placement", i)
                if n_users:
                    assert np.array_equal(g_uu, w_uu), ("user usage", i)
    finally:
        for e in engines:
            e.close()
    for c, (w_f, _, _) in zip(cases, want):
        o_ranked, _ = pyoracle.rank(c["params"], c["pool"].tasks, c["pool"].users, quota=c.get("quota"))
        assert np.array_equal(w_f[0], o_ranked)
    st = stats[-1]
    if len(cases) > 1:  # (a call for ONE engine is cook_cycle_run_rank)
        assert st["rank_batch_pools"] == len(cases) and st["rank_batch_grouped_launches"] >= min_grouped, st
    return stats


def rank_batch_cases(scale=1):
    """pools whose rank flows differ: multi-block sizes, tie-heavy, fractional (inexact prefixes), equal-DRU runs, gpu mode with
    good-enough-fitness below 1, pool / group quota + offensive filter, no running task, considerable filters"""
    rng = np.random.default_rng(5)
    cases = []
    cases.append(dict(pool=synth.make_pool(seed=201, n_pending=5000 * scale, n_running=1500 * scale, n_users=300, n_offers=60, gpus=True, constraints=True),
                      params=A.default_params(good_enough_fitness=1.0, match_algo=2, max_over_quota_jobs=10)))  # multi-block scans / sorts
    cases.append(dict(pool=synth.make_pool(seed=202, n_pending=900 * scale, n_running=300 * scale, n_users=25, n_offers=40, tie_heavy=True),
                      params=A.default_params(good_enough_fitness=1.0, match_algo=2)))
    cases.append(dict(pool=synth.make_pool(seed=203, n_pending=700 * scale, n_running=300 * scale, n_users=25, n_offers=40, fractional=True),
                      params=A.default_params(good_enough_fitness=1.0, match_algo=2)))
    runs = synth.make_pool(seed=204, n_pending=900 * scale, n_running=300 * scale, n_users=25, n_offers=40, tie_heavy=True)  # equal-DRU runs: the collapse path
    z = rng.random(runs.tasks.n) < 0.2
    runs.tasks.cpus[z] = 1e-17
    runs.tasks.mem[z] = 1e-17
    cases.append(dict(pool=runs, params=A.default_params(good_enough_fitness=1.0, match_algo=2, max_over_quota_jobs=10)))
    gp = synth.make_pool(seed=205, n_pending=800 * scale, n_running=400 * scale, n_users=30, n_offers=40, gpus=True)
    cases.append(dict(pool=gp, params=A.default_params(good_enough_fitness=0.8, match_algo=2, dru_mode=1)))
    cases.append(dict(pool=synth.make_pool(seed=206, n_pending=800 * scale, n_running=400 * scale, n_users=30, n_offers=40, gpus=True),
                      params=A.default_params(good_enough_fitness=1.0, match_algo=2, offensive_max_mem_mb=16000.0, offensive_max_cpus=6.0),
                      quota=A.pool_quota(pool_quota=A.quota(count=600 * scale, cpus=2500.0 * scale), group_quota=A.quota(mem=4.0e6 * scale),
                                         group_usage=A.usage(count=10, cpus=100, mem=1.0e6))))
    cases.append(dict(pool=synth.make_pool(seed=207, n_pending=600 * scale, n_running=0 * scale, n_users=7, n_offers=20, tie_heavy=True, quota_frac=0.5),
                      params=A.default_params(good_enough_fitness=1.0, match_algo=2)))
    cp = synth.make_pool(seed=208, n_pending=700 * scale, n_running=300 * scale, n_users=12, n_offers=40)
    _, st = make_considerable_case(seed=61, n=10, n_users=12)
    elig = (rng.random(700 * scale) < 0.8).astype(np.uint8)
    cases.append(dict(pool=cp, params=A.default_params(good_enough_fitness=1.0, match_algo=2), cons=(st, elig)))
    return cases


class _raises:
    def __init__(self, exc, text):
        self.exc, self.text = exc, text

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        assert et is not None and issubclass(et, self.exc) and self.text in str(ev), (et, ev)
        return True


def rank_batch_one_flow_fails(make_engine):
    """an engine that was never staged fails in ITS flow; the others' results are complete and right"""
    from cook_amd.engine import CookError, cycle_match_multi, cycle_run_rank_multi
    cases = rank_batch_cases()[1:4]
    engines = [make_engine(c["params"]) for c in cases] + [make_engine(A.default_params())]
    try:
        for e, c in zip(engines, cases):
            pool = c["pool"]
            e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        with _raises(CookError, "before cook_cycle_stage"):
            cycle_run_rank_multi(engines, 10 ** 9)
        with _raises(CookError, "before cook_cycle_stage"):  # ... also when the failing engine leads the call (its message survives the call's own epilogue)
            cycle_run_rank_multi(engines[::-1], 10 ** 9)
        cycle_match_multi(engines[:3])
        for e, c in zip(engines, cases):
            ranked, j2o, head = e.cycle_fetch()
            o_ranked, _ = pyoracle.rank(c["params"], c["pool"].tasks, c["pool"].users)
            assert np.array_equal(ranked, o_ranked)
    finally:
        for e in engines:
            e.close()


def mixed_chain_parity(make_engine, pools, params_list, ks, rank_batched=False):
    """One lockstep chain whose pools DISAGREE: good-enough-fitness below 1 next to best fit (the whole chain then runs the good-enough
    launches: a best-fit pool is placed by best fit all the same, from that flavour's shorter best-fit lists), and different numbers
    of considerable jobs.  Every pool against the oracle under ITS parameters."""
    from cook_amd.engine import cycle_match_multi, cycle_run_rank_multi
    engines = [make_engine(p) for p in params_list]
    try:
        for e, pool, k in zip(engines, pools, ks):
            e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
            if not rank_batched:
                e.cycle_run_rank(k)
        if rank_batched:  # ONE call for the rank parts, every pool with its own K (cook_cycle_run_rank_multi)
            cycle_run_rank_multi(engines, ks)
        cycle_match_multi(engines)
        got = [e.cycle_fetch() for e in engines]
    finally:
        for e in engines:
            e.close()
    for (ranked, j2o, head), pool, p, k in zip(got, pools, params_list, ks):
        o_ranked, _ = pyoracle.rank(p, pool.tasks, pool.users)
        assert np.array_equal(ranked, o_ranked)
        pend_ord = np.cumsum(pool.tasks.pending) - 1
        kk = min(k, len(o_ranked))
        o_j2o, _, o_head = pyoracle.match(p, pool.pending_jobs.take(pend_ord[o_ranked[:kk]]), pool.offers, pool.groups)
        assert np.array_equal(j2o, o_j2o) and head == o_head, (p.good_enough_fitness, k)
    return got


def cycle_update_mask_parity(make_engine, seed, n_pending=700, n_running=300, n_users=25, n_offers=100, n_remove=120, n_add=160):
    """The eligible mask of cook_cycle_set_considerable travels through cook_cycle_update: rows of removed pending jobs leave it, added
    pending jobs are eligible (1) — the next cycle equals a restage of the updated arrays with the moved mask set explicitly."""
    rng = np.random.default_rng(seed)
    p = A.default_params(good_enough_fitness=1.0)
    pool = synth.make_pool(seed=seed, n_pending=n_pending, n_running=n_running, n_users=n_users, n_offers=n_offers, gpus=True, constraints=True)
    extra = synth.make_pool(seed=seed + 1000, n_pending=n_add // 2, n_running=n_add - n_add // 2, n_users=n_users, n_offers=8, gpus=True,
                            constraints=True, id_base=27_592_186_044_416)
    ng = pool.groups.n if pool.groups is not None else 0
    add_jobs = extra.pending_jobs
    if add_jobs.group is not None:
        add_jobs.group = np.where((add_jobs.group != A.NONE_U32) & (ng > 0), add_jobs.group % max(1, ng), A.NONE_U32).astype(np.uint32)
    remove = np.sort(rng.choice(pool.tasks.n, size=min(n_remove, pool.tasks.n), replace=False)).astype(np.uint32)
    mask = (rng.random(pool.n_pending) < 0.7).astype(np.uint8)  # by pending ordinal
    keep = np.ones(pool.tasks.n, bool)
    keep[remove] = False
    T, X = pool.tasks, extra.tasks
    cat = lambda a, b: np.concatenate([a[keep], b])  # noqa: E731
    tasks2 = A.Tasks(cpus=cat(T.cpus, X.cpus), mem=cat(T.mem, X.mem), gpus=cat(T.gpus, X.gpus), user=cat(T.user, X.user),
                     priority=cat(T.priority, X.priority), start_ms=cat(T.start_ms, X.start_ms), task_id=cat(T.task_id, X.task_id),
                     job_id=cat(T.job_id, X.job_id), pending=cat(T.pending, X.pending))
    keep_p = keep[np.nonzero(T.pending)[0]]
    jobs2 = _concat_jobs(pool.pending_jobs.take(np.nonzero(keep_p)[0]), add_jobs)
    mask2 = np.concatenate([mask[keep_p], np.ones(add_jobs.n, np.uint8)])
    k = 10 ** 9
    big = np.full(n_users, 1e15)
    st = A.UserState(quota_count=big, quota_cpus=big, quota_mem=big, quota_gpus=big, usage_count=np.zeros(n_users), usage_cpus=np.zeros(n_users),
                     usage_mem=np.zeros(n_users), usage_gpus=np.zeros(n_users))  # (quotas that never bind: the mask alone filters)
    with make_engine(p) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e.cycle_set_considerable(st, mask)
        e.cycle_run(k)
        e.cycle_update(remove, extra.tasks, add_jobs, None)
        e.cycle_run(k)
        got = e.cycle_fetch()
        got_pos = e.cycle_fetch_considerable()
    with make_engine(p) as e:
        e.cycle_stage(tasks2, pool.users, jobs2, pool.offers, pool.groups)
        e.cycle_set_considerable(st, mask2)
        e.cycle_run(k)
        want = e.cycle_fetch()
        want_pos = e.cycle_fetch_considerable()
    assert np.array_equal(got_pos, want_pos), "the considerable jobs after the update differ from a restage with the moved mask"
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    # ... and the mask really bites: fewer considerable jobs than ranked ones
    assert 0 < len(got_pos) < len(got[0])
    return got


def edge_cases(make_engine):
    """Empty and degenerate inputs through the C ABI: they must behave like the oracle, not crash."""
    prm = A.default_params(good_enough_fitness=1.0)
    pool = synth.make_pool(seed=81, n_pending=40, n_running=10, n_users=5, n_offers=6)
    none_off = A.Offers(cpus=np.zeros(0), mem=np.zeros(0))
    none_job = A.Jobs(cpus=np.zeros(0), mem=np.zeros(0))
    # no offers: every job unmatched, failure code "nothing evaluated" (8); head-matched-or-no-matches is true
    j2o = match_parity(make_engine, pool.pending_jobs, none_off, None, prm)
    assert (j2o < 0).all()
    # no jobs
    with make_engine(prm) as e:
        j2o, fail, head = e.match(none_job, pool.offers)
    assert len(j2o) == 0 and head
    # offers nobody fits: zero capacity
    zero = A.Offers(cpus=np.zeros(4), mem=np.zeros(4))
    assert (match_parity(make_engine, pool.pending_jobs, zero, None, prm) < 0).all()
    # a cycle that considers nothing
    with make_engine(prm) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e.cycle_run(0)
        ranked, j2o, head = e.cycle_fetch()
    assert len(ranked) == pool.n_pending and len(j2o) == 0 and head
    # rebalancer: nothing pending / no budget
    b = make_rebalance_case(seed=82, n_running=50, n_pending=0, n_users=4, n_hosts=5)
    assert rebalance_parity(make_engine, b)["decisions"] == []
    b = make_rebalance_case(seed=83, n_running=50, n_pending=6, n_users=4, n_hosts=5, max_preemption=0)
    assert rebalance_parity(make_engine, b)["decisions"] == []
    # considerable: K = 0
    queue, st = make_considerable_case(seed=84, n=30, n_users=4)
    assert len(considerable_parity(make_engine, queue, st, 0)) == 0


# ---- offer construction from node state (cook_offers_build vs oracle/k8s_offers.py) ---------------------------------------
def _offers_equal(got: A.BuiltOffers, want, tag):
    for k, v in want["rows"].items():
        g = getattr(got, k)
        v = v.reshape(g.shape) if v.size == 0 else v
        assert g.dtype == v.dtype and np.array_equal(g, v), (tag, k, g[:8], v[:8])  # fp64 ==: bit-exact sums
    assert np.array_equal(got.node_status, want["status"]), (tag, "status")
    for k, v in want["totals"].items():
        assert got.totals[k] == v, (tag, k, got.totals[k], v)
    for k in ("gpu_capacity_by_model", "gpu_consumed_by_model", "disk_capacity_by_type", "disk_consumed_by_type"):
        assert np.array_equal(getattr(got, k), want[k]), (tag, k, getattr(got, k), want[k])


def offers_parity(make_engine, nodes, pods, oparams, tag=""):
    from oracle import k8s_offers
    with make_engine(A.default_params()) as e:
        got = e.offers_build(nodes, pods, oparams)
        e.offers_run()  # repeatable on staged inputs
        again = e.offers_fetch()
    want = k8s_offers.build_rows(nodes, pods, oparams)
    _offers_equal(got, want, tag)
    _offers_equal(again, want, tag + " (second run)")
    if nodes.attr is not None:
        assert np.array_equal(got.attr, nodes.attr[got.node]), (tag, "attr rows")
    return got


def check_offers_golden(make_engine):
    """the reference's own generate-offers / get-consumption vectors through the C ABI"""
    from oracle import k8s_offers
    gold = G.load("offers")
    for case in gold["generate"]:
        nodes, pods, op, names, gm, dt = G.build_offers_inputs(case)
        got = offers_parity(make_engine, nodes, pods, op, case["name"])
        assert got.n == case["n_offers"], case["name"]
        by = {names[v]: r for r, v in enumerate(got.node)}
        for host, exp in case["expect"].items():
            r = by[host]
            assert got.cpus[r] == exp["cpus"] and got.mem[r] == exp["mem"], (case["name"], host, got.cpus[r], got.mem[r])
            gpus = {gm[got.gpu_model[r] - 1]: got.gpu_count[r]} if got.gpu_model[r] else {}
            disk = {dt[got.disk_type[r] - 1]: got.disk_space[r]} if got.disk_type[r] else {}
            assert gpus == exp["gpus"] and disk == exp["disk"], (case["name"], host, gpus, disk)
    # get-consumption vectors: every host gets an ample node, so consumption = capacity - available
    for case in gold["consumption"]:
        hosts = sorted({p["node"] for p in case["pods"] if p["node"]})
        big = dict(cpu=64.0, memory=65536.0)
        nodes_l = []
        for h in hosts:
            exp = case["expect"].get(h, {})
            alloc = dict(big)
            gt = next(iter(exp.get("gpus", {})), None)
            dty = next(iter(exp.get("disk", {})), None)
            if gt:
                alloc["nvidia.com/gpu"] = 16
            if dty:
                alloc["ephemeral-storage"] = 1048576.0
            nodes_l.append(dict(name=h, allocatable=alloc, gpu_type=gt, disk_type=dty))
        nodes, pods, op, names, gm, dt = G.build_offers_inputs(dict(nodes=nodes_l, pods=case["pods"], clobber=case["clobber"]))
        got = offers_parity(make_engine, nodes, pods, op, case["name"])
        for r, v in enumerate(got.node):
            exp = case["expect"][names[v]]
            assert 64.0 - got.cpus[r] == exp["cpus"] and 65536.0 - got.mem[r] == exp["mem"], (case["name"], names[v])
            if exp.get("gpus"):
                assert 16 - got.gpu_count[r] == next(iter(exp["gpus"].values())), (case["name"], names[v])
            if exp.get("disk"):
                assert 1048576.0 - got.disk_space[r] == next(iter(exp["disk"].values())), (case["name"], names[v])
    for case in gold["schedulable"]:
        n = dict(case["node"])
        n["allocatable"] = dict(n.get("allocatable") or {}, cpu=1.0, memory=1.0)
        nodes, pods, op, *_ = G.build_offers_inputs(dict(nodes=[n], pods=[dict(name="p", node=n["name"], containers=[{"cpu": 0.5}])],
                                                         max_pods=30, filter_unsound=case["filter_unsound"]))
        got = offers_parity(make_engine, nodes, pods, op, case["ref"])
        assert (got.n == 1) == case["expect"], case["ref"]
    assert k8s_offers is not None


def offers_edge_cases(make_engine):
    op = A.offer_params(max_pods_per_node=4, n_gpu_models=2, n_disk_types=2)
    none = A.Pods(node=np.zeros(0), cpus=np.zeros(0), mem=np.zeros(0))
    # no nodes at all; nodes without pods; pods without nodes; a node exactly at / below the pod limit
    got = offers_parity(make_engine, A.Nodes(cpus=np.zeros(0), mem=np.zeros(0)), none, op, "no nodes")
    assert got.n == 0 and got.totals["cpus_capacity"] == 0.0
    nodes = A.Nodes(cpus=[4.0, 8.0, 2.0], mem=[1024.0, 2048.0, 512.0], gpus=[0, 2, 0], gpu_model=[0, 1, 0])
    assert offers_parity(make_engine, nodes, none, op, "no pods").n == 3
    stray = A.Pods(node=[A.NONE_U32, 7, 3], cpus=[1.0, 1.0, 1.0], mem=[1.0, 1.0, 1.0])
    got = offers_parity(make_engine, nodes, stray, op, "pods without a node of the pool")
    assert got.n == 3 and got.totals["cpus_consumed"] == 0.0
    limit = A.Pods(node=[0, 0, 0, 0, 1, 1, 1], cpus=[0.1] * 7, mem=[0.3] * 7)
    got = offers_parity(make_engine, nodes, limit, op, "pod limit")
    assert list(got.node) == [1, 2] and got.num_pods[0] == 3
    # over-committed node: negative availability clamps to 0.0 for cpus / mem, gpu count goes negative as in the reference
    over = A.Pods(node=[1, 1], cpus=[6.0, 6.0], mem=[4096.0, 1.0], gpus=[2, 1], gpu_model=[1, 1])
    got = offers_parity(make_engine, nodes, over, op, "over-committed")
    assert got.cpus[1] == 0.0 and got.mem[1] == 0.0 and got.gpu_count[1] == -1.0


def offers_feed_match(make_engine, n_nodes=120, n_pods=700, n_jobs=300):
    """offer rows built on the device are valid match input — through the host (BuiltOffers.as_offers) and IN PLACE
    (cook_match_stage_built_offers / cook_cycle_stage_built_offers): both give the assignments the oracle's match gives on the
    oracle's offers"""
    from oracle import k8s_offers
    nodes, pods, op = synth.make_cluster_state(seed=11, n_nodes=n_nodes, n_pods=n_pods, n_attr_keys=8, fractional=False, max_pods=12)
    pool = synth.make_pool(seed=12, n_pending=n_jobs, n_running=n_jobs // 3, n_users=30, n_offers=10, gpus=True)
    pend = pool.pending_jobs
    p = A.default_params(good_enough_fitness=1.0)
    with make_engine(p) as e:
        built = e.offers_build(nodes, pods, op)
        j2o, _, head = e.match(pend, built.as_offers(), None)
        e.offers_run()
        e.match_stage_built_offers(pend)                      # device columns in place, no task limits
        e.match_run()
        j2o_dev, _, head_dev = e.match_fetch()
        e.match_stage_built_offers(pend, with_task_limits=True)  # + COOK_MAX_TASKS_PER_HOST / COOK_NUM_TASKS_ON_HOST
        e.match_run()
        j2o_lim, _, _ = e.match_fetch()
        e.cycle_stage_built_offers(pool.tasks, pool.users, pend, None, with_task_limits=True)
        e.cycle_run(n_jobs)
        ranked, j2o_cyc, _ = e.cycle_fetch()
    want = k8s_offers.build_rows(nodes, pods, op)["rows"]
    kw = dict(cpus=want["cpus"], mem=want["mem"], host=want["host"], k8s=np.ones(len(want["cpus"]), np.uint8),
              gpu_model=want["gpu_model"], gpu_count=want["gpu_count"], disk_type=want["disk_type"], disk_space=want["disk_space"],
              attr=nodes.attr[want["node"]])
    o_j2o, _, o_head = pyoracle.match(p, pend, A.Offers(**kw), None)
    assert np.array_equal(j2o, o_j2o) and head == o_head and (j2o >= 0).sum() > n_jobs // 10
    assert np.array_equal(j2o_dev, o_j2o) and head_dev == o_head
    lim = A.Offers(max_tasks=np.full(len(want["cpus"]), op.max_pods_per_node, np.int32), num_tasks=want["num_pods"], **kw)
    o_lim, _, _ = pyoracle.match(p, pend, lim, None)
    assert np.array_equal(j2o_lim, o_lim) and not np.array_equal(o_lim, o_j2o)  # the pod limit binds somewhere
    o_ranked, _ = pyoracle.rank(p, pool.tasks, pool.users)
    pend_ord = np.cumsum(pool.tasks.pending) - 1
    o_cyc, _, _ = pyoracle.match(p, pend.take(pend_ord[o_ranked]), lim, None)
    assert np.array_equal(ranked, o_ranked) and np.array_equal(j2o_cyc, o_cyc)


# ---- why-unscheduled summaries and match-cycle metrics (consumers of the placement's by-products) --------------------------
def explain_parity(make_engine, jobs, offers, groups, params, reserved=(), max_pos=48, tag=""):
    """cook_match_explain after a match == the oracle's summary taken DURING its sweep (fenzo_utils.clj:33-55)"""
    with make_engine(params) as e:
        j2o, fail, _ = e.match(jobs, offers, groups, reserved)
        unm = np.nonzero(j2o < 0)[0]
        mat = np.nonzero(j2o >= 0)[0]
        pos = np.concatenate([unm[:: max(1, len(unm) // max_pos)][:max_pos], mat[:: max(1, len(mat) // 8)][:8], unm[-1:]]).astype(np.uint32)
        counts = e.match_explain(pos)
        again = e.match_explain(pos[::-1].copy())[::-1]
    o_j2o, o_counts = pyoracle.match_explain(params, jobs, offers, groups, reserved, pos)
    assert np.array_equal(j2o, o_j2o), tag
    bad = np.nonzero((counts != o_counts).any(axis=1))[0]
    assert len(bad) == 0, (tag, pos[bad[:3]], counts[bad[:3]], o_counts[bad[:3]])
    assert np.array_equal(again, counts), (tag, "order of the positions must not matter")
    # every host shows up in exactly one class, except that a host short of several resources counts under each of them
    res = [A.WHY_SLOTS - 20 + x for x in (0, 1, 14, 15, 16, 17)]
    for q, k in enumerate(pos):
        row = counts[q]
        n_con = int(row[2:14].sum())
        assert n_con + int(row[res].max()) <= offers.n, (tag, k, row)
        if j2o[k] < 0:  # the fail code of the match is the OR of the classes seen, and no host accepted the job
            assert n_con + int(row[res].sum()) >= offers.n, (tag, k, row)
            want = (1 if row[res].any() else 0) | (2 if row[3:14].any() else 0) | (4 if row[2] else 0)
            assert fail[k] == (want if want else 8), (tag, k, fail[k], row)
    return pos, counts


def metrics_parity(make_engine, jobs, offers, groups, params, n_users=0, n_models=2, tag=""):
    """cook_match_metrics == handle-match-cycle-metrics' numbers recomputed from the oracle's placement"""
    with make_engine(params) as e:
        j2o, _, head = e.match(jobs, offers, groups)
        m = e.match_metrics(n_users=n_users if jobs.user is not None else 0, n_gpu_models=n_models)
    o_j2o, _, o_head = pyoracle.match(params, jobs, offers, groups)
    assert np.array_equal(j2o, o_j2o)
    assert m["considerable"] == jobs.n and m["matched"] == int((o_j2o >= 0).sum()) and m["unmatched"] == int((o_j2o < 0).sum()), tag
    assert m["offers"] == offers.n and m["offers_scheduled"] == len(set(o_j2o[o_j2o >= 0].tolist())), tag
    assert m["head_matched"] == (bool(o_j2o[0] >= 0) if jobs.n else False), tag
    for got, cols in ((m["jobs"], (jobs.cpus, jobs.mem)), (m["offers_stats"], (offers.cpus, offers.mem))):
        want = pyoracle.resource_stats(*cols)
        for k, v in want.items():
            assert got[k] == v or (np.isnan(v) and np.isnan(got[k])), (tag, k, got[k], v)
    if jobs.user is not None and n_users:
        assert np.array_equal(m["user_considerable"], np.bincount(jobs.user, minlength=n_users)), tag
        assert np.array_equal(m["user_matched"], np.bincount(jobs.user[o_j2o >= 0], minlength=n_users)), tag
    if jobs.gpus is not None:
        jm = jobs.gpu_model if jobs.gpu_model is not None else np.zeros(jobs.n, np.uint32)
        want = np.bincount(jm[jobs.gpus > 0], weights=jobs.gpus[jobs.gpus > 0], minlength=n_models + 1)[: n_models + 1]
        assert np.array_equal(m["job_gpus_by_model"], want.astype(np.int64)), tag
    if offers.gpu_model is not None:
        gm, gc = offers.gpu_model.reshape(-1), offers.gpu_count.reshape(-1)  # every entry of every host's map
        sel = gm != 0
        want = np.bincount(gm[sel], weights=gc[sel], minlength=n_models + 1)[: n_models + 1]
        assert np.array_equal(m["offer_gpus_by_model"], want.astype(np.int64)), tag
    return m


def cycle_explain_parity(make_engine, pool: synth.Pool, params, k, n_users):
    """explain / metrics after cook_cycle_run: the jobs of the match are the first k ranked jobs (j_index indirection)"""
    with make_engine(params) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e.cycle_run(k)
        ranked, j2o, _ = e.cycle_fetch()
        unm = np.nonzero(j2o < 0)[0]
        pos = np.concatenate([unm[:24], [0]]).astype(np.uint32)
        counts = e.match_explain(pos)
        m = e.match_metrics(n_users=n_users, n_gpu_models=2)
    pend_ord = np.cumsum(pool.tasks.pending) - 1
    considerable = pool.pending_jobs.take(pend_ord[ranked[:k]])
    o_j2o, o_counts = pyoracle.match_explain(params, considerable, pool.offers, pool.groups, (), pos)
    assert np.array_equal(j2o, o_j2o) and np.array_equal(counts, o_counts)
    assert m["matched"] == int((o_j2o >= 0).sum()) and m["considerable"] == len(o_j2o)
    assert np.array_equal(m["user_considerable"], np.bincount(considerable.user, minlength=n_users))
    want = pyoracle.resource_stats(considerable.cpus, considerable.mem)
    assert all(m["jobs"][key] == v for key, v in want.items())


# ---- trace replay (cook_amd/replay.py): the simulator loop over the engine vs the same loop over the oracle ----------------
class OracleBackend:
    """the CPU oracle behind the replay harness' backend interface (test infrastructure)"""

    def rank(self, params, tasks, users):
        return pyoracle.rank(params, tasks, users)[0]

    def match(self, params, jobs, offers):
        return pyoracle.match(params, jobs, offers)[0]

    def rebalance(self, params, running, pending, job_ids, priorities, users, spare, rparams):
        return pyoracle.rebalance(params, running, pending, job_ids, priorities, users, spare, rparams)["decisions"]


def make_trace(seed, n_jobs, n_hosts, n_users=5, span_ms=3_600_000, fail_frac=0.05):
    """a trace / hosts pair in the format of the reference's simulator_files (example-trace.json, example-hosts.json);
    job shapes as simulator/src/main/cook/sim/schedule.clj:58-83"""
    rng = np.random.default_rng(seed)
    submit = np.sort(rng.integers(0, span_ms, n_jobs))
    users = [chr(ord("a") + i) for i in range(n_users)]
    weights = 1.0 / np.arange(1, n_users + 1)
    trace = []
    for i in range(n_jobs):
        trace.append({
            "run-time-ms": int(rng.integers(60_000, 1_800_000)), "submit-time-ms": int(submit[i]), "job/priority": int(rng.integers(0, 101)),
            "job/resource": [{"resource/type": "resource.type/cpus", "resource/amount": float(rng.integers(1, 5))},
                             {"resource/type": "resource.type/mem", "resource/amount": float(rng.integers(512, 4096))}],
            "job/max-retries": 5, "job/name": "dummy_job", "job/uuid": "591c97d8-0000-4000-8000-%012d" % i,
            "job/user": users[int(rng.choice(n_users, p=weights / weights.sum()))], "job/expected-runtime": 100,
            "status": "failed" if rng.random() < fail_frac else "finished"})
    hosts = [{"hostname": "%03d" % h, "attributes": {}, "slave-id": "slave-%03d" % h,
              "resources": {"cpus": {"*": 10}, "mem": {"*": 10000}}} for h in range(n_hosts)]
    return trace, hosts


def replay_parity(make_engine, trace, hosts, config, min_preempted=0, max_cycles=10 ** 9, min_matched=None):
    from cook_amd import replay
    with make_engine(A.default_params()) as e:
        got = replay.simulate(trace, hosts, config, replay.EngineBackend(e), max_cycles)
    want = replay.simulate(trace, hosts, config, OracleBackend(), max_cycles)
    assert got.cycles == want.cycles and got.log == want.log, [(a, b) for a, b in zip(got.log, want.log) if a != b][:3]
    rg, rw = got.rows(), want.rows()
    assert rg == rw, next((a, b) for a, b in zip(rg, rw) if a != b)
    assert sum(r["matched"] for r in got.log) >= (min(len(trace) // 2, 10) if min_matched is None else min_matched)
    assert sum(r["preempted"] for r in got.log) >= min_preempted
    assert set(rg[0].keys()) == set(replay.CSV_HEADERS)
    return got


def offers_many_models_and_types(make_engine):
    """more gpu models than the kernel totals in LDS (global-atomic path) and more disk types than one pass of the gauge
    kernel folds (several passes of 8 quantities)"""
    rng = np.random.default_rng(77)
    n, p, n_models, n_types = 900, 5000, 70, 6
    gm = rng.integers(0, n_models + 1, n).astype(np.uint32)
    gp = np.where(gm > 0, rng.integers(1, 9, n), 0).astype(np.int32)
    dt = rng.integers(0, n_types + 1, n).astype(np.uint32)
    nodes = A.Nodes(cpus=rng.integers(8, 65, n).astype(float), mem=rng.integers(8, 65, n) * 1024.0, gpus=gp, gpu_model=gm,
                    disk=np.where(dt > 0, 100000.0 + rng.integers(0, 50, n) * 0.1, -1.0), disk_type=dt)
    pn = rng.integers(0, n, p).astype(np.uint32)
    want_g = (gm[pn] > 0) & (rng.random(p) < 0.5)
    want_d = (dt[pn] > 0) & (rng.random(p) < 0.5)
    pods = A.Pods(node=pn, cpus=rng.integers(1, 4, p) + 0.1, mem=rng.integers(100, 900, p) + 0.3,
                  gpus=np.where(want_g, 1, 0).astype(np.int32), gpu_model=np.where(want_g, gm[pn], 0).astype(np.uint32),
                  disk=np.where(want_d, 10.7, -1.0), disk_type=np.where(want_d, dt[pn], 0).astype(np.uint32))
    op = A.offer_params(max_pods_per_node=64, n_gpu_models=n_models, n_disk_types=n_types)
    got = offers_parity(make_engine, nodes, pods, op, "many models / types")
    assert (got.gpu_consumed_by_model > 0).sum() > 40 and (got.disk_consumed_by_type > 0).sum() == n_types


def check_replay_recorded(backend, max_cycles=10 ** 9):
    """The reference's OWN recorded simulator run (tests/golden/replay_example.json <- simulator_files/example-*): the replay
    reproduces every recorded task row — job, hostname, slave id, user, resources, status, the cycle it started in and the cycle it
    was seen finished in (up to `max_cycles` cycles of the run)."""
    import json
    import math
    import os
    from cook_amd import replay
    g = json.load(open(os.path.join(G.GOLDEN, "replay_example.json")))
    sim = replay.simulate(g["trace"], g["hosts"], g["config"], backend, max_cycles)
    step = g["config"]["cycle-step-ms"]
    rows = {r["job_id"]: r for r in sim.rows()}
    t0 = min(r["start_time_ms"] for r in rows.values()) if rows else 0  # cycles are counted from the first task start, as in the fixture
    n = 0
    for job_id, e in g["expect"].items():
        if e["start_cycle"] >= max_cycles - 1:
            continue
        r = rows[job_id]
        sc = round((r["start_time_ms"] - t0) / step)
        assert (r["hostname"], r["slave_id"], r["user"], r["mem"], r["cpus"], sc) == \
               (e["hostname"], e["slave_id"], e["user"], e["mem"], e["cpus"], e["start_cycle"]), (job_id, r, e)
        if max_cycles >= 10 ** 9:
            ec = math.ceil((r["end_time_ms"] - t0) / step) if r["end_time_ms"] != "" else None
            assert (r["status"], ec) == (e["status"], e["end_cycle"]), (job_id, r, e)
        n += 1
    if max_cycles >= 10 ** 9:
        assert n == len(g["expect"]) == len(rows) == 115
    return n


def _concat_jobs(a: A.Jobs, b: A.Jobs) -> A.Jobs:
    """rows of a followed by rows of b (optional columns: taken from whoever has them, defaults for the other)"""
    kw = {}
    na, nb = a.n, b.n
    defaults = dict(gpus=0.0, gpu_model=0, user=0, group=A.NONE_U32, reserved_host=-1, ckpt_location=0, est_end_ms=0, disk_request=-1.0, disk_type=0)
    for name in ("cpus", "mem", "gpus", "gpu_model", "user", "group", "reserved_host", "ckpt_location", "est_end_ms", "disk_request", "disk_type"):
        xa, xb = getattr(a, name), getattr(b, name)
        if xa is None and xb is None:
            continue
        ref = xa if xa is not None else xb
        fa = xa if xa is not None else np.full(na, defaults[name], dtype=ref.dtype)
        fb = xb if xb is not None else np.full(nb, defaults[name], dtype=ref.dtype)
        kw[name] = np.concatenate([fa, fb])
    for off, vals in (("eq_off", ("eq_key", "eq_val")), ("novel_off", ("novel_host",))):
        oa, ob = getattr(a, off), getattr(b, off)
        if oa is None and ob is None:
            continue
        oa = oa if oa is not None else np.zeros(na + 1, np.uint32)
        ob = ob if ob is not None else np.zeros(nb + 1, np.uint32)
        kw[off] = np.concatenate([oa[:na + 1], ob[1:nb + 1] + oa[na]]).astype(np.uint32)
        for v in vals:
            va = getattr(a, v) if getattr(a, off) is not None else np.zeros(0, np.uint32)
            vb_ = getattr(b, v) if getattr(b, off) is not None else np.zeros(0, np.uint32)
            kw[v] = np.concatenate([va[:oa[na]], vb_[:ob[nb]]]).astype(np.uint32)
    for name, dflt in (("ports", 0), ("scalars", np.nan)):
        xa, xb = getattr(a, name), getattr(b, name)
        if xa is None and xb is None:
            continue
        ref = xa if xa is not None else xb
        shape = lambda n: (n,) + ref.shape[1:]  # noqa: E731
        fa = xa if xa is not None else np.full(shape(na), dflt, dtype=ref.dtype)
        fb = xb if xb is not None else np.full(shape(nb), dflt, dtype=ref.dtype)
        kw[name] = np.concatenate([fa, fb])
    return A.Jobs(**kw)


def cycle_update_xres_parity(make_engine, seed, n=260, m=60, n_add=70, n_remove=50):
    """cook_cycle_update with every optional column on both sides of the delta: jobs with ports, two named scalars, gpu models, disk
    requests, EQUALS / novel-host constraints; fresh offers with ports, scalars, slot tables of two gpu models / disk types and an
    attribute column.  The delta travels as ONE block (cycle_update.hpp): a column laid out wrongly in it shows here.
    == cook_cycle_stage of the updated arrays (fresh engine) == oracle."""
    rng = np.random.default_rng(seed)
    p = A.default_params()
    J1, O1, _ = xres_random_case(seed, n, m, ports=True, scalars=2, slots=2, constraints=True)
    J2, O2, _ = xres_random_case(seed + 1, n_add, m + 9, ports=True, scalars=2, slots=2, constraints=True)
    n_users = 9
    J1.user = rng.integers(0, n_users, n).astype(np.uint32)
    J2.user = rng.integers(0, n_users, n_add).astype(np.uint32)

    def tasks_of(J, base):
        k = J.n
        return A.Tasks(cpus=J.cpus, mem=J.mem, gpus=J.gpus, user=J.user, priority=rng.integers(0, 100, k).astype(np.int32),
                       start_ms=np.zeros(k, np.int64), task_id=np.zeros(k, np.int64), job_id=(base + np.arange(k)).astype(np.int64),
                       pending=np.ones(k, np.uint8))
    T1, T2 = tasks_of(J1, 1000), tasks_of(J2, 9000)
    users = A.Users(div_cpus=np.full(n_users, 40.0), div_mem=np.full(n_users, 65536.0), div_gpus=np.full(n_users, 4.0))
    remove = np.sort(rng.choice(n, size=n_remove, replace=False)).astype(np.uint32)
    keep = np.ones(n, bool)
    keep[remove] = False
    cat = lambda a, b: np.concatenate([a[keep], b])  # noqa: E731
    T12 = A.Tasks(cpus=cat(T1.cpus, T2.cpus), mem=cat(T1.mem, T2.mem), gpus=cat(T1.gpus, T2.gpus), user=cat(T1.user, T2.user),
                  priority=cat(T1.priority, T2.priority), start_ms=cat(T1.start_ms, T2.start_ms), task_id=cat(T1.task_id, T2.task_id),
                  job_id=cat(T1.job_id, T2.job_id), pending=cat(T1.pending, T2.pending))
    J12 = _concat_jobs(J1.take(np.nonzero(keep)[0]), J2)
    with make_engine(p) as e:
        e.cycle_stage(T1, users, J1, O1, None)
        e.cycle_run(10 ** 9)
        e.cycle_update(remove, T2, J2, O2)
        e.cycle_run(10 ** 9)
        got = e.cycle_fetch()
        e.cycle_update(np.zeros(0, np.uint32), None, None, O1)  # the offers alone, back to the first set
        e.cycle_run(10 ** 9)
        got_b = e.cycle_fetch()
    for off, got_x in ((O2, got), (O1, got_b)):
        with make_engine(p) as e:
            e.cycle_stage(T12, users, J12, off, None)
            e.cycle_run(10 ** 9)
            want = e.cycle_fetch()
        assert np.array_equal(got_x[0], want[0]) and np.array_equal(got_x[1], want[1]) and got_x[2] == want[2]
        o_ranked, _ = pyoracle.rank(p, T12, users)
        assert np.array_equal(got_x[0], o_ranked)
        o_j2o, _, o_head = pyoracle.match(p, J12.take(o_ranked), off, None)
        assert np.array_equal(got_x[1], o_j2o) and got_x[2] == o_head
    assert (got[1] >= 0).sum() > 0
    return got


def cycle_update_parity(make_engine, seed, n_pending=900, n_running=400, n_users=30, n_offers=120, n_remove=150, n_add=200, new_offers=True, k=10 ** 9):
    """cook_cycle_update(delta) on resident inputs == cook_cycle_stage of the updated arrays (fresh engine) == oracle."""
    rng = np.random.default_rng(seed)
    p = A.default_params(good_enough_fitness=1.0)
    pool = synth.make_pool(seed=seed, n_pending=n_pending, n_running=n_running, n_users=n_users, n_offers=n_offers, gpus=True, constraints=True)
    extra = synth.make_pool(seed=seed + 1000, n_pending=n_add // 2, n_running=n_add - n_add // 2, n_users=n_users, n_offers=max(8, n_offers // 2 + 7),
                            gpus=True, constraints=True, id_base=27_592_186_044_416)
    # group ids of the added jobs must exist in the staged group table: map them into it (or none)
    ng = pool.groups.n if pool.groups is not None else 0
    add_jobs = extra.pending_jobs
    if add_jobs.group is not None:
        add_jobs.group = np.where((add_jobs.group != A.NONE_U32) & (ng > 0), add_jobs.group % max(1, ng), A.NONE_U32).astype(np.uint32)
    remove = np.sort(rng.choice(pool.tasks.n, size=min(n_remove, pool.tasks.n), replace=False)).astype(np.uint32)
    offers2 = extra.offers if new_offers else None
    # ---- the updated arrays, on the host --------------------------------------------------------------------------------------
    keep = np.ones(pool.tasks.n, bool)
    keep[remove] = False
    T, X = pool.tasks, extra.tasks
    cat = lambda a, b: np.concatenate([a[keep], b])
    tasks2 = A.Tasks(cpus=cat(T.cpus, X.cpus), mem=cat(T.mem, X.mem), gpus=cat(T.gpus, X.gpus), user=cat(T.user, X.user),
                     priority=cat(T.priority, X.priority), start_ms=cat(T.start_ms, X.start_ms), task_id=cat(T.task_id, X.task_id),
                     job_id=cat(T.job_id, X.job_id), pending=cat(T.pending, X.pending))
    pend_idx = np.nonzero(T.pending)[0]
    keep_p = keep[pend_idx]
    jobs2 = _concat_jobs(pool.pending_jobs.take(np.nonzero(keep_p)[0]), add_jobs)
    off_final = offers2 if offers2 is not None else pool.offers
    with make_engine(p) as e:
        e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
        e.cycle_run(k)       # a cycle on the old state first: the update must not depend on a fresh stage
        e.cycle_update(remove, extra.tasks, add_jobs, offers2)
        e.cycle_run(k)
        got = e.cycle_fetch()
    with make_engine(p) as e:
        e.cycle_stage(tasks2, pool.users, jobs2, off_final, pool.groups)
        e.cycle_run(k)
        want = e.cycle_fetch()
    assert np.array_equal(got[0], want[0]), "ranked order after the update differs from a restage"
    assert np.array_equal(got[1], want[1]) and got[2] == want[2], "placements after the update differ from a restage"
    o_ranked, _ = pyoracle.rank(p, tasks2, pool.users)
    assert np.array_equal(got[0], o_ranked)
    pend_ord = np.cumsum(tasks2.pending) - 1
    kk = min(k, len(o_ranked))
    o_j2o, _, o_head = pyoracle.match(p, jobs2.take(pend_ord[o_ranked[:kk]]), off_final, pool.groups)
    assert np.array_equal(got[1], o_j2o) and got[2] == o_head
    return got


# ---- Fenzo's other resource dimensions: ports and named scalars (scheduler.clj:456-471, 177-189; offer.clj:57-73) and hosts whose
# ---- k8s "gpus" / "disk" maps hold several entries (constraints.clj:122-199) ----------------------------------------------------
def xres_known_answers(make_engine):
    """small cases whose answers follow from the reference's rules by hand"""
    p = A.default_params()
    nan = float("nan")
    # (1) VERDICT r1's example: a job with :ports 2 against an offer without a ports resource -> no match, a resource failure
    jobs = A.Jobs(cpus=[1.0, 1.0], mem=[100.0, 100.0], ports=[2, 0])
    offers = A.Offers(cpus=[4.0], mem=[1000.0])
    with make_engine(p) as e:
        j2o, fail, head = e.match(jobs, offers, None)
        counts = e.match_explain(np.array([0], np.uint32))
    assert j2o.tolist() == [-1, 0] and fail.tolist() == [1, 0] and not head
    assert counts[0, A.WHY_PORTS] == 1 and counts[0].sum() == 1
    assert A.why_summary(counts[0]) == {}  # Fenzo's PORTS failure has no message: the reference's summary skips it
    # (2) ports are consumed: ranges of 3 / 1 / 2 ports; cpuMemBinPacker prefers the fullest host that still fits
    jobs = A.Jobs(cpus=[1.0] * 5, mem=[100.0] * 5, ports=[2, 2, 1, 1, 1])
    offers = A.Offers(cpus=[8.0, 4.0, 4.0], mem=[8000.0, 1000.0, 1000.0], ports=[3, 1, 2])
    want = pyoracle.match(p, jobs, offers, None)[0]
    # job0 -> offer 2 (smallest host with 2 ports), job1 -> offer 0, job2: offer 2 is out of ports, offer 1 (0.175) beats offer 0
    assert want.tolist() == [2, 0, 1, 0, -1], want
    with make_engine(p) as e:
        j2o, fail, _ = e.match(jobs, offers, None)
    assert j2o.tolist() == want.tolist() and fail.tolist() == [0, 0, 0, 0, 1]
    # (3) named scalars: "disk" 10 / none / 25 against leases holding 30 / 0; a job without a request under the name is not tested
    jobs = A.Jobs(cpus=[1.0] * 4, mem=[100.0] * 4, scalars=np.array([[10.0], [nan], [25.0], [15.0]]))
    offers = A.Offers(cpus=[2.0, 8.0], mem=[1000.0, 8000.0], scalars=np.array([[0.0], [30.0]]))
    with make_engine(p) as e:
        j2o, fail, _ = e.match(jobs, offers, None)
        counts = e.match_explain(np.array([2], np.uint32))
    assert j2o.tolist() == [1, 0, -1, 1] and fail.tolist() == [0, 0, 1, 0]  # 10 + 25 > 30; 10 + 15 <= 30
    assert A.why_summary(counts[0], scalar_names=("disk",)) == {":resources": {"disk": 2}}
    # (4) a pool adjuster (scheduler.clj:473-479): TaskRequest cpus 2.0 but the named "cpus" request stays 3.0
    jobs = A.Jobs(cpus=[2.0, 2.0], mem=[100.0, 100.0], scalars=np.array([[3.0, 100.0], [3.0, 100.0]]))
    offers = A.Offers(cpus=[5.0], mem=[1000.0], scalars=np.array([[5.0, 1000.0]]))
    with make_engine(p) as e:
        j2o, fail, _ = e.match(jobs, offers, None)
        counts = e.match_explain(np.array([1], np.uint32))
    assert j2o.tolist() == [0, -1]  # 2 + 2 <= 5 cpus, but 3 + 3 > 5 of the scalar named "cpus"
    assert A.why_summary(counts[0], scalar_names=("cpus", "mem")) == {":resources": {"cpus": 1}}
    # (5) a host whose "gpus" map has two entries (tools.clj:985): (get model->count model 0) per model, (count map) for the rest
    jobs = A.Jobs(cpus=[1.0] * 4, mem=[100.0] * 4, gpus=[2.0, 4.0, 4.0, 0.0], gpu_model=[1, 2, 1, 0])
    offers = A.Offers(cpus=[8.0, 8.0, 8.0], mem=[8000.0] * 3, k8s=[1, 1, 1], gpu_model=[[1, 2], [0, 2], [0, 0]], gpu_count=[[2.0, 4.0], [0.0, 4.0], [0.0, 0.0]])
    with make_engine(p) as e:
        j2o, _, _ = e.match(jobs, offers, None)
        m = e.match_metrics(n_gpu_models=2)
    assert j2o.tolist() == [0, 1, -1, 2]  # job1: host 0 is taken (one gpu job per VM), host 1 has 4 of model 2; job2: nobody has 4 of model 1
    assert m["offer_gpus_by_model"].tolist() == [0, 2, 8]
    # (6) a "disk" map with two types (constraints.clj:164-199): >= on the requested type, 0 when the host lacks it
    jobs = A.Jobs(cpus=[1.0] * 3, mem=[100.0] * 3, disk_request=[40.0, 40.0, 10.0], disk_type=[2, 1, 3])
    offers = A.Offers(cpus=[8.0, 4.0], mem=[8000.0, 4000.0], k8s=[1, 1], disk_type=[[1, 2], [2, 0]], disk_space=[[30.0, 100.0], [50.0, 0.0]])
    with make_engine(p) as e:
        j2o, fail, _ = e.match(jobs, offers, None)
    assert j2o.tolist() == [1, -1, -1] and fail.tolist() == [0, 2, 2]


def xres_random_case(seed, n, m, *, ports=True, scalars=2, groups=False, slots=1, constraints=False):
    rng = np.random.default_rng(seed)
    kw = {}
    if ports:
        kw["ports"] = np.where(rng.random(n) < 0.3, rng.integers(1, 4, n), 0)
    if scalars:
        sc = rng.integers(1, 40, (n, scalars)).astype(np.float64) * 0.25
        sc[rng.random((n, scalars)) < 0.6] = np.nan
        kw["scalars"] = sc
    okw = dict(k8s=np.ones(m, np.uint8))
    if slots > 1:  # gpu hosts with several models, some with a zero-count entry; disk maps with several types
        gm = np.zeros((m, slots), np.uint32)
        gc = np.zeros((m, slots))
        dt = np.zeros((m, slots), np.uint32)
        ds = np.zeros((m, slots))
        for v in range(m):
            if rng.random() < 0.4:
                k = int(rng.integers(1, slots + 1))
                gm[v, :k] = rng.permutation(np.arange(1, 5))[:k]
                gc[v, :k] = rng.integers(0, 3, k) * 2.0
            k = int(rng.integers(0, slots + 1))
            dt[v, :k] = rng.permutation(np.arange(1, 5))[:k]
            ds[v, :k] = rng.integers(1, 9, k) * 10.0
        okw.update(gpu_model=gm, gpu_count=gc, disk_type=dt, disk_space=ds)
        g = np.where(rng.random(n) < 0.3, rng.integers(1, 3, n) * 2.0, 0.0)
        kw.update(gpus=g, gpu_model=np.where(g > 0, rng.integers(1, 5, n), 0),
                  disk_request=np.where(rng.random(n) < 0.5, rng.integers(1, 9, n) * 10.0, -1.0), disk_type=rng.integers(1, 4, n))
    grp = None
    if groups:
        n_groups = max(1, n // 12)
        kw["group"] = np.where(rng.random(n) < 0.25, rng.integers(0, n_groups, n), A.NONE_U32).astype(np.uint32)
        grp = A.Groups(type=rng.choice([0, 1, 1, 2, 3], n_groups).astype(np.uint8), attr_key=np.zeros(n_groups, np.uint32),
                       minimum=np.full(n_groups, 2, np.int32))
        okw["attr"] = rng.integers(1, 4, (m, 1)).astype(np.uint32)
    cpus = rng.integers(1, 5, n).astype(float)
    mem = rng.integers(1, 9, n) * 512.0
    if constraints:
        jobs = A.Jobs.with_constraints(cpus, mem, equals=[[(0, int(rng.integers(1, 4)))] if rng.random() < 0.2 else [] for _ in range(n)],
                                       novel=[[int(rng.integers(0, m))] if rng.random() < 0.1 else [] for _ in range(n)], **kw)
        okw.setdefault("attr", rng.integers(1, 4, (m, 1)).astype(np.uint32))
    else:
        jobs = A.Jobs(cpus=cpus, mem=mem, **kw)
    offers = A.Offers(cpus=rng.integers(4, 17, m).astype(float), mem=rng.integers(4, 17, m) * 2048.0,
                      ports=rng.integers(0, 9, m) if ports else None,
                      scalars=(rng.integers(0, 200, (m, scalars)) * 0.25) if scalars else None, **okw)
    return jobs, offers, grp


def offers_slot_tables(make_engine):
    """(:gpus available) / (:disk available) as maps with several entries: a model only the pods name becomes a key of its own
    (deep-merge-with keeps the consumed count as it is, util.clj:208-225) -> slot tables instead of COOK_NODE_ST_FOREIGN_*"""
    from oracle import k8s_offers
    # node 0: 4 x model 1; pods consume 1 x model 1, 2 x model 2, 1 x model 3, 1 x model 2; node 1: no gpus of its own, one pod under model 3
    nodes = A.Nodes(cpus=[32.0, 16.0], mem=[65536.0, 32768.0], gpus=[4, 0], gpu_model=[1, 0], disk=[1000.0, -1.0], disk_type=[1, 0])
    pods = A.Pods(node=[0, 0, 0, 0, 1], cpus=[1.0] * 5, mem=[100.0] * 5, gpus=[1, 2, 1, 1, 2], gpu_model=[1, 2, 3, 2, 3],
                  disk=[10.5, 20.25, -1.0, 0.5, 7.0], disk_type=[1, 2, 0, 2, 2])
    op = A.offer_params(n_gpu_models=3, n_disk_types=2, gpu_slots=3, disk_slots=2)
    got = offers_parity(make_engine, nodes, pods, op, "slot tables")
    assert got.gpu_model.tolist() == [[1, 2, 3], [3, 0, 0]] and got.gpu_count.tolist() == [[3.0, 3.0, 1.0], [2.0, 0.0, 0.0]]
    assert got.disk_type.tolist() == [[1, 2], [2, 0]] and got.disk_space.tolist() == [[989.5, 20.75], [7.0, 0.0]]
    assert got.node_status.tolist() == [3, 3]  # nothing left for the host to rebuild
    # the order of the entries follows the pods that CONSUME: a pod without resource requests (api.clj:908-913) names type 1 first
    # but brings nothing in, so type 2 takes the node's one slot and type 1 overflows
    n1 = A.Nodes(cpus=[8.0], mem=[1024.0])
    p1 = A.Pods(node=[0, 0, 0], cpus=[1.0] * 3, mem=[1.0] * 3, disk=[5.0, 50.25, 7.0], disk_type=[1, 2, 1], flags=[A.POD_NO_REQUESTS, 0, 0])
    g1 = offers_parity(make_engine, n1, p1, A.offer_params(n_disk_types=2), "consuming order")
    assert g1.disk_type.tolist() == [2] and g1.disk_space.tolist() == [50.25] and g1.node_status.tolist() == [1 | 2 | 8]
    # the same nodes through one-entry rows: the second model does not fit and the node is flagged
    one = offers_parity(make_engine, nodes, pods, A.offer_params(n_gpu_models=3, n_disk_types=2), "one slot")
    assert one.gpu_model.tolist() == [1, 3] and one.node_status.tolist() == [3 | 4 | 8, 3]
    # the rows feed the match unchanged, in place: a job asking for 3 x model 2 lands on node 0 (the reference's quirk: what the
    # pods CONSUME under a model the node does not list reads as available), a job without gpus finds no host without a gpus map
    jobs = A.Jobs(cpus=[1.0, 1.0, 1.0], mem=[10.0] * 3, gpus=[3.0, 2.0, 0.0], gpu_model=[2, 3, 0])
    with make_engine(A.default_params()) as e:
        e.offers_build(nodes, pods, op)
        e.match_stage_built_offers(jobs)
        e.match_run()
        j2o, _, _ = e.match_fetch()
    want = pyoracle.match(A.default_params(), jobs, got.as_offers(), None)[0]
    assert j2o.tolist() == want.tolist() == [0, 1, -1]
    # random clusters with "corrupt" pods (models / types their node does not list), every slot count
    rng = np.random.default_rng(91)
    for slots in (1, 2, 4):
        n, p = 300, 2500
        gm = rng.integers(0, 4, n).astype(np.uint32)
        dt = rng.integers(0, 4, n).astype(np.uint32)
        nodes = A.Nodes(cpus=rng.integers(8, 65, n).astype(float), mem=rng.integers(8, 65, n) * 1024.0,
                        gpus=np.where(gm > 0, rng.integers(1, 9, n), 0).astype(np.int32), gpu_model=gm,
                        disk=np.where(dt > 0, 1000.0 + rng.integers(0, 50, n) * 0.1, -1.0), disk_type=dt)
        pn = rng.integers(0, n, p).astype(np.uint32)
        wg, wd = rng.random(p) < 0.4, rng.random(p) < 0.4
        pods = A.Pods(node=pn, cpus=rng.integers(1, 4, p) + 0.1, mem=rng.integers(100, 900, p) + 0.3,
                      gpus=np.where(wg, rng.integers(1, 3, p), 0).astype(np.int32),
                      gpu_model=np.where(wg, np.where(rng.random(p) < 0.7, gm[pn], rng.integers(1, 6, p)), 0).astype(np.uint32),
                      disk=np.where(wd, 10.7, -1.0), disk_type=np.where(wd, np.where(rng.random(p) < 0.7, dt[pn], rng.integers(1, 5, p)), 0).astype(np.uint32))
        got = offers_parity(make_engine, nodes, pods, A.offer_params(max_pods_per_node=64, n_gpu_models=5, n_disk_types=4, gpu_slots=slots,
                                                                     disk_slots=slots), "random slots=%d" % slots)
        flagged = int(((got.node_status & 12) != 0).sum())
        assert (flagged > 0) == (slots < 4), (slots, flagged)
    assert k8s_offers is not None


def metrics_known_answers(make_engine):
    """cook_match_metrics' :percentiles / :totals / :largest-by against HAND-DERIVED answers (the reference holds no vector of
    resource-maps->stats, scheduler.clj:547-582, or task-stats/percentiles, task_stats.clj:59-80: the values follow from the code —
    see tests/test_oracle_golden.py::test_resource_stats_known_answers, which pins the oracle on the same cases)."""
    cases = [
        ([35.0, 20.0, 15.0, 50.0, 40.0], [1.0, 2.0, 3.0, 4.0, 5.0], dict(p50_cpus=35.0, p95_cpus=50.0, p100_cpus=50.0, p50_mem=3.0, p95_mem=5.0,
                                                                        p100_mem=5.0, total_cpus=160.0, total_mem=15.0, largest_by_cpus=3, largest_by_mem=4)),
        ([float(x) for x in range(1, 21)], [float(x) for x in range(20, 0, -1)], dict(p50_cpus=10.0, p95_cpus=19.0, p100_cpus=20.0, p50_mem=10.0,
                                                                                      p95_mem=19.0, largest_by_cpus=19, largest_by_mem=0)),
        ([4.0, 9.0, 9.0, 1.0], [7.0, 7.0, 2.0, 7.0], dict(largest_by_cpus=2, largest_by_mem=3, p50_cpus=4.0, p100_mem=7.0)),
        ([0.1, 0.2, 0.3], [0.3, 0.2, 0.1], dict(total_cpus=(0.1 + 0.2) + 0.3, total_mem=(0.3 + 0.2) + 0.1)),
    ]
    p = A.default_params(good_enough_fitness=1.0)
    for cpus, mem, want in cases:
        jobs = A.Jobs(cpus=cpus, mem=mem)
        offers = A.Offers(cpus=[1000.0], mem=[1e6], host=np.arange(1, dtype=np.uint32))
        with make_engine(p) as e:
            e.match(jobs, offers)
            m = e.match_metrics()
        for k, v in want.items():
            assert m["jobs"][k] == v, (cpus, k, m["jobs"][k], v)


# ---- the reference's own cases of the why-unscheduled reducer (tests/golden/explain.json <- test/cook/test/scheduler/fenzo_utils.clj:56-100) -------------
_WHY_SLOT_OF = {name: slot for slot, (_, name) in A.WHY_NAMES.items()}


def _explain_expect(case):
    """the case's expected map with "other_constraint" (no constraint of Cook's) renamed to the one that stands for it"""
    out = {k: dict(v) for k, v in case["expect"].items()}
    if "other_constraint" in out.get(":constraints", {}):
        out[":constraints"]["user_defined_constraint"] = out[":constraints"].pop("other_constraint")
    return out


def explain_golden_reduce(case):
    """the reducer itself: every host's result counted into the COOK_WHY_* row the engine fills, the row turned into the reference's map"""
    names = []
    for _, lack in case["results"]:
        for r in lack:
            if r not in ("cpus", "mem") and r not in names:
                names.append(r)
    assert len(names) <= 3
    row = np.zeros(A.WHY_SLOTS, np.uint32)
    for cons, lack in case["results"]:
        for r in lack:
            row[{"cpus": 0, "mem": 1}.get(r, A.WHY_SCALAR0 + (names.index(r) if r in names else 0))] += 1
        if cons:
            row[_WHY_SLOT_OF["user_defined_constraint" if cons == "other_constraint" else cons]] += 1
    assert A.why_summary(row, scalar_names=tuple(names)) == _explain_expect(case), case["ref"]


def explain_golden_inputs(case):
    """the case's hosts as a placement's inputs: one job, a host per result (see tests/golden/make_golden.py EXPLAIN) -> (jobs, offers, scalar names)"""
    eng = case["engine"]
    hosts, names = eng["hosts"], list(eng["scalars"])
    H = len(hosts)
    novel = [h for h, d in enumerate(hosts) if d.get("constraint") == "novel_host_constraint"]
    other = [h for h, d in enumerate(hosts) if d.get("constraint") == "other_constraint"]
    kw = {}
    if names:
        kw["scalars"] = np.ones((1, len(names)))
    jobs = A.Jobs.with_constraints(np.array([1.0]), np.array([100.0]), equals=[[(0, 1)] if other else []], novel=[novel], **kw)
    attr = np.ones((H, 1), np.uint32)
    attr[other, 0] = 2
    okw = dict(attr=attr) if other else {}
    if names:
        sc = np.full((H, len(names)), 5.0)
        for h, d in enumerate(hosts):
            for r in d.get("lack", []):
                sc[h, names.index(r)] = 0.0
        okw["scalars"] = sc
    offers = A.Offers(cpus=np.full(H, 4.0), mem=np.full(H, 1000.0), host=np.arange(H, dtype=np.uint32), **okw)
    return jobs, offers, tuple(names)


def explain_golden_engine(make_engine, case):
    """cook_match_explain on the case's inputs == the reference's expected map"""
    jobs, offers, names = explain_golden_inputs(case)
    with make_engine(A.default_params(good_enough_fitness=1.0)) as e:
        j2o, _, _ = e.match(jobs, offers, None)
        counts = e.match_explain(np.array([0], np.uint32))
    assert (j2o[0] >= 0) == (case["expect"] == {}), case["ref"]
    assert A.why_summary(counts[0], scalar_names=names) == _explain_expect(case), case["ref"]


def pool_usage_multi_parity(make_engine, n=5):
    """cook_rank_pool_usage_multi == cook_rank_pool_usage of every engine (bit for bit: a pool's reduction order does not depend on its
    neighbours) == the running rows summed on the host; an engine twice is COOK_E_INVALID; an empty pool and a pool without running tasks."""
    from cook_amd.engine import CookError, rank_pool_usage_multi
    shapes = [(900, 700), (40, 0), (0, 0), (3000, 5000), (200, 1)] + [(150 * i, 90 * i) for i in range(5, n)]
    pools = [synth.make_pool(seed=4100 + i, n_pending=npd, n_running=nr, n_users=25, n_offers=16, gpus=(i % 2 == 0), fractional=(i % 3 == 0))
             for i, (npd, nr) in enumerate(shapes[:n])]
    engines = [make_engine(A.default_params()) for _ in pools]
    try:
        for e, pl in zip(engines, pools):
            e.rank_stage(pl.tasks, pl.users)
        multi = rank_pool_usage_multi(engines)
        for e in engines:  # (a second table: the memo of the first call must not answer for it)
            e.rank_stage(pools[0].tasks, pools[0].users)
        for e, pl in zip(engines, pools):
            e.rank_stage(pl.tasks, pl.users)
        single = [e.rank_pool_usage().as_tuple() for e in engines]
        assert multi == single, (multi, single)
        assert rank_pool_usage_multi(engines[::-1]) == single[::-1]
        for u, pl in zip(multi, pools):
            run = pl.tasks.pending == 0
            assert u[0] == float(run.sum())
            assert abs(u[1] - float(pl.tasks.cpus[run].sum())) <= 1e-9 * max(1.0, u[1])
        try:
            rank_pool_usage_multi([engines[0], engines[1], engines[0]])
        except CookError as ex:
            assert ex.code == -1, ex  # COOK_E_INVALID
        else:
            raise AssertionError("an engine twice in cook_rank_pool_usage_multi was accepted")
    finally:
        for e in engines:
            e.close()

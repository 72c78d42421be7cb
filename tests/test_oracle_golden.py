"""Pins the CPU oracle (oracle/cook_oracle.cpp) to the known answers of the reference's own unit tests
(tests/golden/*.json, transcribed by tests/golden/make_golden.py with reference file:line per case)."""
import numpy as np
import pytest

from cook_amd import _abi as A
from tests import golden_util as G

RANK = G.load("rank")
RANK_GROUP = G.load("rank_group")
MATCH = G.load("match")


def _params(case, **kw):
    return A.default_params(dru_mode=case.get("dru_mode", 0),
                            max_over_quota_jobs=case.get("max_over_quota_jobs", 100), **kw)


@pytest.mark.parametrize("literal", [False, True], ids=["heap-merge", "literal-merge"])
@pytest.mark.parametrize("case", RANK, ids=[c["name"] for c in RANK])
def test_rank_golden(oracle, case, literal):
    tasks, users, names, unames = G.build_rank_inputs(case["jobs"], case["shares"], case.get("quotas"))
    p = _params(case)
    if "expect_ranked" in case:
        ranked, _ = oracle.rank(p, tasks, users, literal_merge=literal)
        assert [names[i] for i in ranked] == case["expect_ranked"], case["ref"]
    idx, drus = oracle.rank_merged(p, tasks, users, literal_merge=literal)
    if "expect_merged_drus" in case:
        assert list(drus) == case["expect_merged_drus"], case["ref"]  # exact equality on doubles, as the reference
    if "expect_merged_names" in case:
        assert [names[i] for i in idx] == case["expect_merged_names"], case["ref"]
    if "expect_merged_users" in case:
        assert [unames[tasks.user[i]] for i in idx] == case["expect_merged_users"], case["ref"]


@pytest.mark.parametrize("case", RANK_GROUP, ids=[c["name"] for c in RANK_GROUP])
def test_rank_quota_group_golden(oracle, case):
    # aggregate-quota-groups (scheduler.clj:2125-2132): group usage = sum of the member pools' running usage.
    usages = {}
    built = {}
    for pool, spec in case["pools"].items():
        tasks, users, names, _ = G.build_rank_inputs(spec["jobs"], case["shares"])
        built[pool] = (tasks, users, names)
        usages[pool] = oracle.pool_usage(tasks)
    gu = A.usage(*[sum(getattr(u, f) for u in usages.values()) for f in ("count", "cpus", "mem", "gpus")])
    for pool, spec in case["pools"].items():
        tasks, users, names = built[pool]
        q = A.pool_quota(pool_quota=G.usage_of(spec["quota"]), group_quota=G.usage_of(case["group_quota"]), group_usage=gu)
        ranked, _ = oracle.rank(A.default_params(), tasks, users, quota=q)
        assert [names[i] for i in ranked] == spec["expect_ranked"], (case["ref"], pool)


def test_quota_group_aggregate_golden():
    c = G.load("quota_group_agg")
    agg = {}
    for pool, u in c["usage"].items():
        g = c["groups"].get(pool)
        if g is None:
            continue
        for k, v in u.items():
            agg.setdefault(g, {}).setdefault(k, 0)
            agg[g][k] += v
    assert agg == c["expect"], c["ref"]


@pytest.mark.parametrize("case", MATCH, ids=[c["name"] for c in MATCH])
def test_match_golden(oracle, case):
    J, O, names, x = G.build_match_all(case)
    p = A.default_params(good_enough_fitness=case["good_enough"], **x["params"])
    j2o, fail, head = oracle.match(p, J, O, x["groups"], x["reserved"])
    G.check_match_expectations(case, names, x["host_names"], j2o, head)
    if "expect_assignment" in case and "expect_n_matched" not in case:
        assert {names[k]: int(j2o[k]) for k in range(J.n) if j2o[k] >= 0} == case["expect_assignment"], case["ref"]
    # multi-thread CPU baseline variant gives the same placement when good-enough is disabled
    if case["good_enough"] >= 1.0 and J.n:
        j2o8, _, _ = oracle.match(p, J, O, x["groups"], x["reserved"], nthreads=4)
        assert np.array_equal(j2o, j2o8)


REBAL = G.load("rebalance")


def check_rebalance_case(case, result, b):
    """Shared by the oracle test and the GPU parity tests: `result` = dict(decisions, pending_dru[, final])."""
    names, hosts, pnames = b["task_names"], b["host_names"], b["pending_names"]
    R = b["running"].n
    dec = result["decisions"]

    def tname(i):
        return names[i] if i != A.NONE_U32 else "<placed-this-cycle>"

    if "expect_decisions" in case:
        assert len(dec) == len(case["expect_decisions"]), case["ref"]
        for d, e in zip(dec, case["expect_decisions"]):
            assert pnames[d["pending_index"]] == e["job"], case["ref"]
            assert hosts[d["host"]] == e["host"], case["ref"]
            assert d["dru"] == (A.DMAX if e["dru"] == "MAX" else e["dru"]), case["ref"]  # exact, as the reference's (is (= ...))
            assert [tname(t) for t in d["tasks"]] == e["tasks"], case["ref"]
            assert (d["mem"], d["cpus"], d["gpus"]) == (e["mem"], e["cpus"], e["gpus"]), case["ref"]
    if "expect_decision_host_in" in case:
        assert len(dec) == 1 and hosts[dec[0]["host"]] in case["expect_decision_host_in"], (case["ref"], dec)
    if "expect_pending_dru" in case:
        for n, v in case["expect_pending_dru"].items():
            assert result["pending_dru"][pnames.index(n)] == v, (case["ref"], n)
    if "expect_jobs_to_run" in case:
        assert [pnames[d["pending_index"]] for d in dec] == case["expect_jobs_to_run"], case["ref"]
        assert [tname(t) for d in dec for t in d["tasks"]] == case["expect_tasks_to_preempt"], case["ref"]
    if "expect_final_order" in case and result.get("final") is not None:
        order, drus = result["final"]
        assert [names[i] for i in order] == case["expect_final_order"], case["ref"]
        assert list(drus) == case["expect_final_drus"], case["ref"]


@pytest.mark.parametrize("case", REBAL, ids=[c["name"] for c in REBAL])
def test_rebalance_golden(oracle, case):
    b = G.build_rebalance_inputs(case)
    res = oracle.rebalance(b["params"], b["running"], b["pending"], b["pending_job_id"], b["pending_priority"], b["users"],
                           b["spare"], b["rparams"], host_attrs=b["host_attrs"], groups=b["groups"], forced=b["forced"],
                           want_final=True, slave_known=b["slave_known"], init_preempted_hosts=b["init_preempted_hosts"])
    check_rebalance_case(case, res, b)


CONS = G.load("considerable")


def check_considerable_case(case, idx, rate_limited, names, unames):
    assert [names[i] for i in idx] == case["expect"], (case["name"], case["ref"])
    if "expect_rate_limited" in case:
        got = {u: int(c) for u, c in zip(unames, rate_limited) if c}
        assert got == case["expect_rate_limited"], (case["name"], case["ref"])


@pytest.mark.parametrize("case", CONS, ids=[c["name"] for c in CONS])
def test_considerable_golden(oracle, case):
    queue, st, names, unames = G.build_considerable_inputs(case)
    idx, rl, _ = oracle.considerable(queue, st, case["num_considerable"])
    check_considerable_case(case, idx, rl, names, unames)


# ---- offer construction from node state (oracle/k8s_offers.py vs the reference's kubernetes tests) -------------------------
def _n2p(pods):
    out = {}
    for p in pods:
        out.setdefault(p["node"], []).append(p)
    return out


def test_k8s_consumption_golden():
    from oracle import k8s_offers as K
    for c in G.load("offers")["consumption"]:
        assert K.get_consumption(c["clobber"], _n2p(c["pods"])) == c["expect"], c["name"]


def test_k8s_capacity_golden():
    from oracle import k8s_offers as K
    for c in G.load("offers")["capacity"]:
        assert K.get_capacity({n["name"]: n for n in c["nodes"]}) == c["expect"], c["name"]


def test_k8s_node_schedulable_golden():
    from oracle import k8s_offers as K
    for c in G.load("offers")["schedulable"]:
        n2p = {c["node"]["name"]: [dict(name="p")]}  # num-pods-on-node redefined to 1, capacity 30 (api.clj:844)
        assert K.node_schedulable(c["node"], 30, n2p, c["filter_unsound"]) == c["expect"], c["ref"]
    assert K.node_schedulable(None, 30, {}) is False  # api.clj:789-790


def test_k8s_generate_offers_golden():
    from oracle import k8s_offers as K
    for c in G.load("offers")["generate"]:
        offers, gauges, _ = K.generate_offers({n["name"]: n for n in c["nodes"]}, _n2p(c["pods"]), max_pods_per_node=c["max_pods"])
        assert len(offers) == c["n_offers"], c["name"]
        by = {o["hostname"]: o for o in offers}
        for host, exp in c["expect"].items():
            got = {k: by[host][k] for k in ("mem", "cpus", "disk", "gpus")}
            assert got == exp, (c["name"], host, got)
        assert gauges["nodes_total"] == 5 and gauges["nodes_schedulable"] == 5


def test_resource_stats_known_answers():
    """oracle/pyoracle.resource_stats (row n3) pinned BY RULE: the reference holds no vector of resource-maps->stats
    (scheduler/src/cook/scheduler/scheduler.clj:547-582) or task-stats/percentiles (task_stats.clj:59-80), so the expected values are
    derived by hand from the code: percentile p of the sorted values = (nth sorted (dec (ceil (* (/ p 100) n)))) in exact ratio
    arithmetic (Nearest Rank), :totals = (reduce +) left to right, :largest-by = the LAST element of a stable sort by the resource
    (among equal maxima the one that comes last in the input)."""
    from oracle import pyoracle
    # five values (the Nearest Rank article's example): n = 5 -> p50 index ceil(2.5) - 1 = 2, p95 index ceil(4.75) - 1 = 4, p100 index 4
    s = pyoracle.resource_stats([35.0, 20.0, 15.0, 50.0, 40.0], [1.0, 2.0, 3.0, 4.0, 5.0])
    assert (s["p50_cpus"], s["p95_cpus"], s["p100_cpus"]) == (35.0, 50.0, 50.0)
    assert (s["p50_mem"], s["p95_mem"], s["p100_mem"]) == (3.0, 5.0, 5.0)
    assert s["total_cpus"] == 160.0 and s["total_mem"] == 15.0
    assert s["largest_by_cpus"] == 3 and s["largest_by_mem"] == 4
    # ten values: p50 -> index ceil(5) - 1 = 4 (the LOWER median: (* 1/2 10) is exactly 5, no float fuzz), p95 -> ceil(9.5) - 1 = 9
    v = [3.0, 6.0, 7.0, 8.0, 8.0, 10.0, 13.0, 15.0, 16.0, 20.0]
    s = pyoracle.resource_stats(v[::-1], v)
    assert (s["p50_cpus"], s["p95_cpus"], s["p100_cpus"]) == (8.0, 20.0, 20.0)
    assert s["largest_by_cpus"] == 0 and s["largest_by_mem"] == 9
    # 20 values 1..20: p95 -> index ceil(19) - 1 = 18 -> 19 (95/100 * 20 is exactly 19 as a ratio; as a double product it is
    # 19.000000000000004 -> 20: the oracle must use the ratio, as Clojure does)
    v = [float(x) for x in range(1, 21)]
    s = pyoracle.resource_stats(v, v)
    assert (s["p50_cpus"], s["p95_cpus"], s["p100_cpus"]) == (10.0, 19.0, 20.0)
    # a tie at the maximum: (last (sort-by ...)) of a stable sort is the later one
    s = pyoracle.resource_stats([4.0, 9.0, 9.0, 1.0], [7.0, 7.0, 2.0, 7.0])
    assert s["largest_by_cpus"] == 2 and s["largest_by_mem"] == 3
    # totals are left-to-right fp64 sums (reduce +): 0.1 + 0.2 + 0.3 is 0.6000000000000001, not 0.6
    s = pyoracle.resource_stats([0.1, 0.2, 0.3], [0.3, 0.2, 0.1])
    assert s["total_cpus"] == (0.1 + 0.2) + 0.3 and s["total_mem"] == (0.3 + 0.2) + 0.1 and s["total_cpus"] != s["total_mem"]
    # one value; no value
    s = pyoracle.resource_stats([2.5], [8.0])
    assert (s["p50_cpus"], s["p95_cpus"], s["p100_cpus"], s["largest_by_cpus"]) == (2.5, 2.5, 2.5, 0)
    s = pyoracle.resource_stats([], [])
    assert s["total_cpus"] == 0.0 and s["p50_cpus"] != s["p50_cpus"]


EXPLAIN = G.load("explain")


@pytest.mark.parametrize("case", EXPLAIN, ids=[c["name"] for c in EXPLAIN])
def test_explain_reducer_golden(oracle, case):
    # the reference's own test of summarize-placement-failure (test/cook/test/scheduler/fenzo_utils.clj:56-100): the reducer, and — where Fenzo can
    # produce the hosts' results in a real placement — the oracle's summary of that placement
    from tests import parity_cases as P
    P.explain_golden_reduce(case)
    if "engine" in case:
        jobs, offers, names = P.explain_golden_inputs(case)
        j2o, counts = oracle.match_explain(A.default_params(good_enough_fitness=1.0), jobs, offers, None, (), np.array([0], np.uint32))
        assert (j2o[0] >= 0) == (case["expect"] == {}), case["ref"]
        assert A.why_summary(counts[0], scalar_names=names) == P._explain_expect(case), case["ref"]

#!/usr/bin/env python3
"""Transcribes the known-answer vectors of the reference's OWN unit tests for the fair-share match path into
tests/golden/*.json.  The reference is Clojure (no JVM in this image), so the vectors are transcribed by hand
from the cited test forms — every case names the reference test file:line it restates (paths relative to
/root/reference/scheduler/).  Run `python tests/golden/make_golden.py` to regenerate the JSON files.

Conventions of the transcription
  * entities are listed in creation order; :db/id and :instance/start-time grow with creation order
    (testutil.clj:234-345: start-time defaults to (java.util.Date.) at creation), so `seq` doubles as both.
  * users are referred to by NAME; the loader assigns user ids in ascending name order (cookmatch.h contract).
  * "MAX" stands for Double/MAX_VALUE (share.clj:95 default share, quota defaults).
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
T_DRU = "test/cook/test/scheduler/dru.clj"
T_SCHED = "test/cook/test/scheduler/scheduler.clj"
T_REB = "test/cook/test/rebalancer.clj"


def job(name, user, cpus, mem, gpus=0.0, priority=50, running=False, **kw):
    d = dict(name=name, user=user, cpus=cpus, mem=mem, gpus=gpus, priority=priority, running=running)
    d.update(kw)
    return d


RANK = [
    dict(
        name="compute-task-scored-task-pairs", ref=f"{T_DRU}:30-59", dru_mode=0,
        jobs=[job("t1", "ljin", 10.0, 10.0, running=True), job("t2", "ljin", 5.0, 5.0, running=True),
              job("t3", "ljin", 25.0, 15.0, running=True), job("t4", "ljin", 15.0, 25.0, running=True)],
        shares={"ljin": dict(cpus=25.0, mem=25.0)},
        expect_merged_names=["t1", "t2", "t3", "t4"], expect_merged_drus=[0.4, 0.6, 1.6, 2.2],
    ),
    dict(
        name="sorted-task-scored-task-pairs", ref=f"{T_DRU}:85-123", dru_mode=0,
        jobs=[job("l1", "ljin", 10.0, 10.0, running=True), job("l2", "ljin", 5.0, 5.0, running=True),
              job("l3", "ljin", 25.0, 15.0, running=True), job("l4", "ljin", 15.0, 25.0, running=True),
              job("w1", "wzhao", 10.0, 10.0, running=True), job("s1", "sunil", 10.0, 10.0, running=True)],
        shares={u: dict(cpus=10.0, mem=10.0) for u in ("ljin", "wzhao", "sunil")},
        expect_merged_drus=[1.0, 1.0, 1.0, 1.5, 4.0, 5.5],
    ),
    dict(
        name="sorted-task-scored-task-pairs-with-running", ref=f"{T_DRU}:129-162", dru_mode=0,
        # job1..job5 created in order; instances created for job4 then job2 (so job4's task sorts first)
        jobs=[job("job1", "ljin", 10.0, 10.0), job("job2", "ljin", 20.0, 20.0, running=True, inst_seq=2),
              job("job3", "ljin", 40.0, 40.0), job("job4", "ljin", 80.0, 80.0, running=True, inst_seq=1),
              job("job5", "ljin", 160.0, 160.0)],
        shares={"ljin": dict(cpus=10.0, mem=10.0)},
        expect_merged_names=["job4", "job2", "job1", "job3", "job5"],
        expect_merged_drus=[8.0, 10.0, 11.0, 15.0, 31.0],
    ),
    dict(
        name="compute-sorted-task-cumulative-gpu-score-pairs", ref=f"{T_DRU}:165-188", dru_mode=1,
        jobs=[job("t1", "ljin", 1.0, 10.0, gpus=10.0, running=True), job("t2", "ljin", 1.0, 10.0, gpus=5.0, running=True),
              job("t3", "ljin", 1.0, 10.0, gpus=25.0, running=True), job("t4", "ljin", 1.0, 10.0, gpus=15.0, running=True)],
        shares={"ljin": dict(cpus="MAX", mem="MAX", gpus=10.0)},
        expect_merged_names=["t1", "t2", "t3", "t4"], expect_merged_drus=[1.0, 1.5, 4.0, 5.5],
    ),
    dict(
        name="sorted-task-cumulative-gpu-score-pairs", ref=f"{T_DRU}:190-218", dru_mode=1,
        jobs=[job("l1", "ljin", 1.0, 10.0, gpus=10.0, running=True), job("l2", "ljin", 1.0, 10.0, gpus=5.0, running=True),
              job("l3", "ljin", 1.0, 10.0, gpus=25.0, running=True), job("l4", "ljin", 1.0, 10.0, gpus=15.0, running=True),
              job("w1", "wzhao", 1.0, 10.0, gpus=10.0, running=True), job("s1", "sunil", 1.0, 10.0, gpus=10.0, running=True)],
        shares={"ljin": dict(cpus="MAX", mem="MAX", gpus=5.0), "wzhao": dict(cpus="MAX", mem="MAX", gpus=10.0),
                "sunil": dict(cpus="MAX", mem="MAX", gpus=2.5)},
        expect_merged_users=["wzhao", "ljin", "ljin", "sunil", "ljin", "ljin"],
        expect_merged_drus=[1.0, 2.0, 3.0, 4.0, 8.0, 11.0],
    ),
    dict(
        name="sort-jobs-by-dru default share", ref=f"{T_SCHED}:232-253", dru_mode=0,
        jobs=[job("j1", "ljin", 1.0, 3.0, running=True, inst_seq=1), job("j2", "ljin", 1.0, 5.0),
              job("j3", "ljin", 1.0, 2.0), job("j4", "ljin", 5.0, 5.0),
              job("j5", "wzhao", 6.0, 6.0, running=True, inst_seq=2), job("j6", "wzhao", 5.0, 5.0),
              job("j7", "sunil", 5.0, 10.0, running=True, inst_seq=3), job("j8", "sunil", 5.0, 10.0)],
        shares={u: dict(cpus=10.0, mem=10.0) for u in ("ljin", "wzhao", "sunil")},
        expect_ranked=["j2", "j3", "j6", "j4", "j8"],
    ),
    dict(
        name="sort-jobs-by-dru one user has non-default share", ref=f"{T_SCHED}:255-260", dru_mode=0,
        jobs=[job("j1", "ljin", 1.0, 3.0, running=True, inst_seq=1), job("j2", "ljin", 1.0, 5.0),
              job("j3", "ljin", 1.0, 2.0), job("j4", "ljin", 5.0, 5.0),
              job("j5", "wzhao", 6.0, 6.0, running=True, inst_seq=2), job("j6", "wzhao", 5.0, 5.0),
              job("j7", "sunil", 5.0, 10.0, running=True, inst_seq=3), job("j8", "sunil", 5.0, 10.0)],
        shares={"ljin": dict(cpus=10.0, mem=10.0), "wzhao": dict(cpus=10.0, mem=10.0),
                "sunil": dict(cpus=100.0, mem=100.0)},
        expect_ranked=["j8", "j2", "j3", "j6", "j4"],
    ),
    dict(
        name="sort-jobs-by-dru normal jobs, Double.MAX_VALUE divisors, priority", ref=f"{T_SCHED}:262-271", dru_mode=0,
        jobs=[job("j1n", "u1", 1.0, 1000.0), job("j2n", "u1", 1.0, 1000.0, priority=90),
              job("j3n", "u2", 1.0, 1500.0), job("j4n", "u2", 1.0, 1500.0, priority=30)],
        shares={"u1": dict(cpus="MAX", mem="MAX"), "u2": dict(cpus="MAX", mem="MAX")},
        expect_ranked=["j2n", "j3n", "j1n", "j4n"],
    ),
    dict(
        name="sort-jobs-by-dru gpu jobs", ref=f"{T_SCHED}:273-283", dru_mode=1,
        jobs=[job("j1g", "u1", 1.0, 1000.0, gpus=10.0), job("j2g", "u1", 1.0, 1000.0, gpus=25.0, priority=90),
              job("j3g", "u2", 1.0, 1500.0, gpus=20.0), job("j4g", "u2", 1.0, 1500.0, gpus=10.0, priority=30)],
        shares={"u1": dict(cpus="MAX", mem="MAX", gpus="MAX"), "u2": dict(cpus="MAX", mem="MAX", gpus="MAX")},
        expect_ranked=["j3g", "j2g", "j4g", "j1g"],
    ),
    dict(
        name="sort-jobs-by-dru limit-quota (count quota 1, max-over-quota-jobs 3)", ref=f"{T_SCHED}:300-316",
        dru_mode=0, max_over_quota_jobs=3,
        jobs=[job("rj", "test", 1.0, 10.0, running=True, inst_seq=1)] + [job(f"wj{i}", "test", 1.0, 10.0) for i in range(1, 6)],
        shares={"test": dict(cpus="MAX", mem="MAX")},
        quotas={"test": dict(count=1)},
        expect_ranked=["wj1", "wj2", "wj3"],
    ),
    dict(
        name="limit-over-quota-jobs 25 jobs, count 5, limit 10 -> 15", ref=f"{T_SCHED}:2353-2368",
        dru_mode=0, max_over_quota_jobs=10,
        jobs=[job(f"j{i}", "u", 1.0, 10.0) for i in range(25)],
        shares={"u": dict(cpus="MAX", mem="MAX")}, quotas={"u": dict(count=5)},
        expect_ranked=[f"j{i}" for i in range(15)],
    ),
    dict(
        name="limit-over-quota-jobs no quota -> 25", ref=f"{T_SCHED}:2353-2364",
        dru_mode=0, max_over_quota_jobs=10,
        jobs=[job(f"j{i}", "u", 1.0, 10.0) for i in range(25)],
        shares={"u": dict(cpus="MAX", mem="MAX")}, quotas={"u": dict(count="MAX")},
        expect_ranked=[f"j{i}" for i in range(25)],
    ),
    dict(
        name="gpu share prioritization: ljin double gpu share", ref=f"{T_SCHED}:1172-1193", dru_mode=1,
        jobs=[job("ljin-1", "ljin", 5.0, 5.0, gpus=1.0, running=True, inst_seq=1)] +
             [job(f"ljin-{i}", "ljin", 5.0, 5.0, gpus=1.0) for i in (2, 3, 4)] +
             [job("wzhao-1", "wzhao", 5.0, 5.0, gpus=1.0), job("wzhao-2", "wzhao", 5.0, 5.0, gpus=1.0)],
        shares={"ljin": dict(cpus=1.0, mem=2.0, gpus=2.0), "wzhao": dict(cpus=1.0, mem=2.0, gpus=1.0)},
        expect_ranked=["ljin-2", "wzhao-1", "ljin-3", "ljin-4", "wzhao-2"],
    ),
    dict(
        name="gpu share prioritization: single gpu share (pins the sorted-merge tie rule)", ref=f"{T_SCHED}:1194-1200",
        dru_mode=1,
        jobs=[job("ljin-1", "ljin", 5.0, 5.0, gpus=1.0, running=True, inst_seq=1)] +
             [job(f"ljin-{i}", "ljin", 5.0, 5.0, gpus=1.0) for i in (2, 3, 4)] +
             [job("wzhao-1", "wzhao", 5.0, 5.0, gpus=1.0), job("wzhao-2", "wzhao", 5.0, 5.0, gpus=1.0)],
        shares={"ljin": dict(cpus=1.0, mem=2.0, gpus=1.0), "wzhao": dict(cpus=1.0, mem=2.0, gpus=1.0)},
        expect_ranked=["wzhao-1", "wzhao-2", "ljin-2", "ljin-3", "ljin-4"],
    ),
]

# test-rank-obeys-group-global-quota (scheduler.clj:318-402): pools a and b share quota group "s".
_POOL_A = [job("j1", "ljin", 1.0, 30.0, running=True, inst_seq=1), job("j2", "ljin", 1.0, 50.0),
           job("j3", "ljin", 1.0, 20.0), job("j4", "ljin", 5.0, 5.0)]
_POOL_B = [job("j5", "ljin", 6.0, 6.0, running=True, inst_seq=2), job("j6", "ljin", 5.0, 5.0),
           job("j7", "ljin", 5.0, 10.0, running=True, inst_seq=3), job("j8", "ljin", 5.0, 10.0)]
_BIG = dict(count=10, mem=10000, cpus=10000, gpus=1000)


def _group_case(title, qa, qb, qs, ea, eb, line):
    return dict(name=f"rank-obeys-group-global-quota: {title}", ref=f"{T_SCHED}:{line}",
                pools={"a": dict(jobs=_POOL_A, quota=qa, expect_ranked=ea),
                       "b": dict(jobs=_POOL_B, quota=qb, expect_ranked=eb)},
                group_quota=qs, shares={"ljin": dict(cpus="MAX", mem="MAX")})


RANK_GROUP = [
    _group_case("no limits hit", _BIG, _BIG, _BIG, ["j2", "j3", "j4"], ["j6", "j8"], "336-343"),
    _group_case("group count 4", _BIG, _BIG, dict(_BIG, count=4), ["j2"], ["j6"], "344-353"),
    _group_case("per-pool mem/cpus", dict(_BIG, mem=80, cpus=2), dict(_BIG, mem=21, cpus=16), _BIG, ["j2"], ["j6"], "354-361"),
    _group_case("group mem 96.1 cpus 17.1", _BIG, _BIG, dict(_BIG, mem=96.1, cpus=17.1), ["j2"], ["j6"], "363-372"),
    _group_case("pool a count 4", dict(_BIG, count=4), _BIG, _BIG, ["j2", "j3", "j4"], ["j6", "j8"], "374-381"),
    _group_case("pool b count 3", _BIG, dict(_BIG, count=3), _BIG, ["j2", "j3", "j4"], ["j6"], "382-389"),
    _group_case("both pools", dict(_BIG, count=4), dict(_BIG, count=3), _BIG, ["j2", "j3", "j4"], ["j6"], "390-397"),
]

QUOTA_GROUP_AGG = dict(
    ref=f"{T_SCHED}:222-230", groups={"a": "s", "b": "s"},
    usage={"a": dict(mem=1, cpus=10, count=100), "b": dict(mem=2, cpus=20, count=200),
           "c": dict(mem=4, cpus=40, count=400), "d": dict(mem=8, cpus=80, count=800)},
    expect={"s": dict(count=300, cpus=30, mem=3)},
)


def offer(cpus, mem, **kw):
    d = dict(cpus=cpus, mem=mem)
    d.update(kw)
    return d


# Placement (Fenzo scheduleOnce through Cook).  `expect_matched`: set of matched job names; `expect_offers_used`:
# number of distinct offers with >=1 task; `expect_assignment` only where feasibility forces the mapping.
_J4 = [job(f"j{i}", "u", 1.0, 1000.0) for i in range(1, 5)]
_HRO_JOBS = [job("job-1", "u", 3, 2048), job("job-2", "u", 13, 1024), job("job-3", "u", 7, 4096),
             job("job-4", "u", 11, 1024), job("job-5", "u", 5, 2048, gpus=2, gpu_model="nvidia-tesla-p100"),
             job("job-6", "u", 19, 1024, gpus=4, gpu_model="nvidia-tesla-p100")]
_HRO_OFFERS = [offer(10, 2048), offer(20, 16384), offer(30, 8192)]
MATCH = [
    dict(name="match-offer-to-schedule: consume nothing (no jobs)", ref=f"{T_SCHED}:560-563", good_enough=1.0,
         jobs=[], offers=[offer(2, 2000)], expect_matched=[]),
] + [
    dict(name=f"match-offer-to-schedule: consume nothing offer({c},{m})", ref=f"{T_SCHED}:560-567", good_enough=1.0,
         jobs=_J4, offers=[offer(c, m)], expect_matched=[])
    for c, m in ((0, 0), (0.5, 100), (0.5, 1000), (1, 500))
] + [
    dict(name=f"match-offer-to-schedule: partial offer({c},{m})", ref=f"{T_SCHED}:569-575", good_enough=1.0,
         jobs=_J4, offers=[offer(c, m)], expect_n_matched=1, expect_matched=["j1"])
    for c, m in ((1, 1000), (1.5, 1500))
] + [
    dict(name=f"match-offer-to-schedule: full offer({c},{m})", ref=f"{T_SCHED}:577-584", good_enough=1.0,
         jobs=_J4, offers=[offer(c, m)], expect_matched=["j1", "j2", "j3", "j4"])
    for c, m in ((4, 4000), (5, 5000))
] + [
    dict(name=f"checkpoint locality offer={ol} ckpt={ck} job-loc={jl}", ref=f"{T_SCHED}:586-658", good_enough=1.0,
         jobs=[job("j", "u", 1.0, 1000.0, ckpt_location=(jl if ck else None))],
         offers=[offer(1.0, 1000.0, location=ol)], expect_matched=(["j"] if ok else []))
    for ol, ck, jl, ok in (("a", True, "a", True), ("a", False, "a", True), ("b", True, "a", False),
                           ("b", False, "a", True), ("a", True, "b", False), ("b", True, "b", True))
] + [
    dict(name="match ordering: rank order respected on a 1-cpu host", ref=f"{T_SCHED}:660-706", good_enough=0.8,
         jobs=[job("high-priority", "u", 1.0, 1000.0)] + [job(f"low-{i}", "u", 1.0, 1000.0) for i in range(8)],
         offers=[offer(1.0, 200000.0)], expect_matched=["high-priority"], expect_assignment={"high-priority": 0}),
    dict(name="handle-resource-offers: enough offers for all normal jobs (K=6)", ref=f"{T_SCHED}:1957-1964",
         good_enough=0.8, jobs=_HRO_JOBS, offers=_HRO_OFFERS,
         expect_matched=["job-1", "job-2", "job-3", "job-4"], expect_offers_used=3, expect_head_matched=True,
         # hand-evaluated in SURVEY.md §8(c): cpuMemBinPacker + strict max
         expect_assignment={"job-1": 0, "job-2": 1, "job-3": 1, "job-4": 2}),
    dict(name="handle-resource-offers: K=1", ref=f"{T_SCHED}:1966-1973", good_enough=0.8,
         jobs=_HRO_JOBS[:1], offers=_HRO_OFFERS, expect_matched=["job-1"], expect_offers_used=1),
    dict(name="handle-resource-offers: K=2", ref=f"{T_SCHED}:1975-1982", good_enough=0.8,
         jobs=_HRO_JOBS[:2], offers=_HRO_OFFERS, expect_matched=["job-1", "job-2"], expect_offers_used=2),
]

# ---------------------------------------------------------------------------------------------------------------
# Rebalancer (test/cook/test/rebalancer.clj).  running: instances in creation order (`seq` = :instance/start-time and
# :db/id order); pending: jobs to make room for, in the order given to rebalance; `job_seq` = creation order of the
# job entity among ALL jobs of the test (the pending job's sort key, tools.clj:614-641).
# host_attrs: the agent-attributes-cache by hostname.  `slave_cached: false` on a running task = its (random) slave
# id is not in the cache (testutil create-dummy-instance gives every instance its own slave id); cotask hosts named
# "<host>#nocache" likewise stand for cotasks whose slave id has no cached attribute map.
def run(name, user, cpus, mem, host="localhost", gpus=0.0, priority=50, **kw):
    d = dict(name=name, user=user, cpus=cpus, mem=mem, host=host, gpus=gpus, priority=priority)
    d.update(kw)
    return d


def pend(name, user, cpus, mem, job_seq, gpus=0.0, priority=50, **kw):
    d = dict(name=name, user=user, cpus=cpus, mem=mem, gpus=gpus, priority=priority, job_seq=job_seq)
    d.update(kw)
    return d


_DEF25 = {"default": dict(cpus=25.0, mem=25.0, gpus=1.0)}
_R8 = [run("t1", "ljin", 10.0, 10.0), run("t2", "ljin", 5.0, 5.0), run("t3", "ljin", 25.0, 15.0), run("t4", "ljin", 15.0, 25.0),
       run("t5", "wzhao", 8.0, 8.0), run("t6", "wzhao", 10.0, 10.0), run("t7", "wzhao", 10.0, 10.0), run("t8", "wzhao", 10.0, 10.0)]
# test-compute-preemption-decision "without group constraints": no job5/job6
_RD = [run("t1", "ljin", 10.0, 10.0, "hostA"), run("t2", "ljin", 5.0, 5.0, "hostA"), run("t3", "ljin", 25.0, 15.0, "hostB"),
       run("t4", "ljin", 15.0, 25.0, "hostB"), run("t7", "wzhao", 10.0, 10.0, "hostA"), run("t8", "wzhao", 10.0, 10.0, "hostB")]
_PD = {"job9": pend("job9", "wzhao", 15.0, 15.0, 7), "job10": pend("job10", "sunil", 15.0, 15.0, 8),
       "job11": pend("job11", "ljin", 15.0, 15.0, 9), "job12": pend("job12", "sunil", 40.0, 40.0, 10),
       "job13": pend("job13", "sunil", 45.0, 45.0, 11), "job14": pend("job14", "sunil", 80.0, 80.0, 12)}


def _dec(job, spare, min_diff, expect, line):
    return dict(name=f"compute-preemption-decision {job} spare={spare} min-dru-diff={min_diff}", ref=f"{T_REB}:{line}",
                shares=_DEF25, running=_RD, pending=[_PD[job]], spare=spare,
                params=dict(max_preemption=128, safe_dru_threshold=1.0, min_dru_diff=min_diff),
                expect_decisions=([dict(job=job, **expect)] if expect else []))


_PIGS4 = {"pig1": "straw", "pig2": "sticks", "pig3": "bricks", "pig4": "rebar"}
_HN4 = {h: {"HOSTNAME": h} for h in _PIGS4.values()}
_AZ4 = {"straw": {"az": "east", "HOSTNAME": "straw"}, "sticks": {"az": "east", "HOSTNAME": "sticks"},
        "bricks": {"az": "east", "HOSTNAME": "bricks"}, "rebar": {"az": "west", "HOSTNAME": "rebar"}}
_PIGS8 = {"pig1": "straw", "pig2": "sticks", "pig3": "bricks", "pig4": "rebar", "pig5": "concrete", "pig6": "steel",
          "pig7": "gold", "pig8": "titanium"}
_AZ8 = {"straw": "east", "sticks": "west", "bricks": "south", "rebar": "north", "concrete": "east", "steel": "west",
        "gold": "south", "titanium": "north"}
_P05 = dict(max_preemption=128, safe_dru_threshold=1.0, min_dru_diff=0.05)


def _pigs(pigs, ncpus):
    return [run(f"{u}-task", u, ncpus, 10.0, h) for u, h in pigs.items()]


def _balanced_case(title, group_hosts, preempted_hosts, expect_in, line):
    cot = [run(f"cotask{i}", "diego", 1.0, 10.0, h) for i, h in enumerate(group_hosts)]
    return dict(name=f"compute-preemption-decision balanced group: {title}", ref=f"{T_REB}:{line}",
                shares={"default": dict(cpus=20.0, mem=20.0, gpus=1.0)}, running=_pigs(_PIGS8, 200.0) + cot,
                pending=[pend("pending", "diego", 1.0, 10.0, 100, group="g")],
                host_attrs={h: {"az": z, "HOSTNAME": h} for h, z in _AZ8.items()},
                groups={"g": dict(type="balanced", attribute="az", minimum=4, running_hosts=list(group_hosts))},
                init_preempted_hosts=preempted_hosts, spare={}, params=_P05, expect_decision_host_in=expect_in)


_RL = [run("t1", "ljin", 10.0, 10.0, "hostA"), run("t2", "ljin", 5.0, 5.0, "hostA"), run("t3", "ljin", 25.0, 15.0, "hostB"),
       run("t4", "ljin", 15.0, 25.0, "hostB"), run("t5", "wzhao", 8.0, 8.0, "hostA"), run("t6", "wzhao", 10.0, 10.0, "hostB"),
       run("t7", "wzhao", 10.0, 10.0, "hostA"), run("t8", "wzhao", 10.0, 10.0, "hostB")]
_PW = [pend(f"job{i}", "wzhao", 5.0, 5.0, i) for i in range(9, 19)]
_PS = [pend(f"job{i}", "sunil", 5.0, 5.0, i) for i in range(19, 29)]
_PL = dict(max_preemption=128, safe_dru_threshold=1.0, min_dru_diff=0.0)
_SH25 = {"default": dict(cpus=25.0, mem=25.0)}
# test-next-state: hosts A/B alternate
_RN = [run("t1", "ljin", 10.0, 10.0, "hostA"), run("t2", "ljin", 5.0, 5.0, "hostA"), run("t3", "ljin", 25.0, 15.0, "hostB"),
       run("t4", "ljin", 15.0, 25.0, "hostB"), run("t5", "wzhao", 8.0, 8.0, "hostA"), run("t6", "wzhao", 10.0, 10.0, "hostB"),
       run("t7", "wzhao", 10.0, 10.0, "hostA"), run("t8", "wzhao", 10.0, 10.0, "hostB")]

REBALANCE = [
    dict(name="init-state: priority-map order and DRUs", ref=f"{T_REB}:55-113", shares=_DEF25, running=_R8, pending=[], spare={},
         params=_PL, expect_decisions=[],
         expect_final_order=["t4", "t3", "t8", "t7", "t6", "t2", "t1", "t5"],
         expect_final_drus=[2.2, 1.6, 1.52, 1.12, 0.72, 0.6, 0.4, 0.32]),
    dict(name="compute-pending-default-job-dru", ref=f"{T_REB}:115-160", shares=_DEF25,
         # job4 and job11 are created with the misspelt key :ucpus, so they get the default 1.0 cpus (testutil.clj:234-250)
         running=[run("t1", "ljin", 10.0, 10.0), run("t2", "ljin", 5.0, 5.0), run("t3", "ljin", 25.0, 15.0), run("t4", "ljin", 1.0, 25.0),
                  run("t5", "wzhao", 8.0, 8.0), run("t6", "wzhao", 10.0, 10.0), run("t7", "wzhao", 10.0, 10.0), run("t8", "wzhao", 10.0, 10.0)],
         pending=[pend("job9", "wzhao", 10.0, 10.0, 9), pend("job10", "sunil", 20.0, 20.0, 10), pend("job11", "ljin", 1.0, 10.0, 11)],
         spare={}, params=dict(max_preemption=128, safe_dru_threshold="MAX", min_dru_diff=0.0),
         expect_decisions=[], expect_pending_dru={"job9": 1.92, "job10": 0.8, "job11": 2.6}),
    dict(name="compute-pending-gpu-job-dru", ref=f"{T_REB}:162-196", dru_mode=1, shares=_DEF25,
         running=[run(f"t{i}", "ljin" if i <= 4 else "wzhao", 1.0, 10.0, gpus=1.0) for i in range(1, 9)],
         pending=[pend("job9", "wzhao", 10.0, 10.0, 9, gpus=1.0), pend("job10", "sunil", 20.0, 20.0, 10, gpus=1.0),
                  pend("job11", "ljin", 1.0, 10.0, 11, gpus=2.0)],
         spare={}, params=dict(max_preemption=128, safe_dru_threshold="MAX", min_dru_diff=0.0),
         expect_decisions=[], expect_pending_dru={"job9": 5.0, "job10": 1.0, "job11": 6.0}),
    _dec("job9", {}, 0.05, dict(host="hostB", dru=2.2, tasks=["t4"], mem=25.0, cpus=15.0, gpus=0.0), "253-266"),
    _dec("job9", {"hostB": dict(mem=15.0, cpus=15.0)}, 0.5, dict(host="hostB", dru="MAX", tasks=[], mem=15.0, cpus=15.0, gpus=0.0), "268-281"),
    _dec("job9", {"hostA": dict(mem=20.0, cpus=20.0), "hostB": dict(mem=10.0, cpus=10.0)}, 0.5,
         dict(host="hostA", dru="MAX", tasks=[], mem=20.0, cpus=20.0, gpus=0.0), "283-297"),
    _dec("job9", {"hostA": dict(mem=10.0, cpus=10.0), "hostB": dict(mem=10.0, cpus=10.0)}, 0.0,
         dict(host="hostB", dru=2.2, tasks=["t4"], mem=35.0, cpus=25.0, gpus=0.0), "299-313"),
    _dec("job10", {}, 0.5, dict(host="hostB", dru=2.2, tasks=["t4"], mem=25.0, cpus=15.0, gpus=0.0), "315-328"),
    _dec("job11", {}, 0.5, None, "330-343"),
    _dec("job12", {}, 0.0, None, "345-358"),
    _dec("job12", {"hostA": dict(mem=40.0, cpus=40.0)}, 0.5, dict(host="hostA", dru="MAX", tasks=[], mem=40.0, cpus=40.0, gpus=0.0), "360-373"),
    _dec("job12", {"hostA": dict(mem=35.0, cpus=35.0)}, 0.0, None, "375-388"),
    _dec("job12", {"hostA": dict(mem=35.0, cpus=35.0), "hostB": dict(mem=30.0, cpus=30.0)}, 0.5,
         dict(host="hostB", dru=2.2, tasks=["t4"], mem=55.0, cpus=45.0, gpus=0.0), "390-404"),
    _dec("job13", {}, 0.5, None, "406-419"),
    _dec("job13", {}, 2.0, None, "421-434"),
    _dec("job14", {}, 0.5, None, "436-440"),
    dict(name="compute-preemption-decision novel-host: failed everywhere except rebar", ref=f"{T_REB}:441-482",
         shares={"default": dict(cpus=10.0, mem=10.0, gpus=10.0)}, running=_pigs(_PIGS4, 100.0),
         pending=[pend("pending", "diego", 1.0, 10.0, 100, novel=["straw", "sticks", "bricks"])],
         host_attrs=_HN4, spare={}, params=_P05, expect_decision_host_in=["rebar"]),
    dict(name="compute-preemption-decision novel-host: unconstrained job finds a host", ref=f"{T_REB}:511-530",
         shares={"default": dict(cpus=10.0, mem=10.0, gpus=10.0)}, running=_pigs(_PIGS4, 100.0),
         pending=[pend("pending", "diego", 1.0, 10.0, 100)], host_attrs=_HN4, spare={}, params=_P05,
         expect_decision_host_in=["straw", "sticks", "bricks", "rebar"]),
    dict(name="compute-preemption-decision unique group: one unconstrained host", ref=f"{T_REB}:532-575",
         shares={"default": dict(cpus=10.0, mem=10.0, gpus=1.0)},
         running=_pigs(_PIGS4, 100.0) + [run(f"cotask-{h}", "diego", 1.0, 10.0, h, slave_cached=False) for h in ("straw", "sticks", "bricks")],
         pending=[pend("pending", "diego", 1.0, 10.0, 100, group="g")], host_attrs=_HN4,
         groups={"g": dict(type="unique", running_hosts=["straw#nocache", "sticks#nocache", "bricks#nocache"])},
         spare={}, params=_P05, expect_decision_host_in=["rebar"]),
    dict(name="compute-preemption-decision unique group: no unconstrained host", ref=f"{T_REB}:577-600",
         shares={"default": dict(cpus=10.0, mem=10.0, gpus=1.0)},
         running=_pigs(_PIGS4, 100.0) + [run(f"cotask-{h}", "diego", 1.0, 10.0, h, slave_cached=False) for h in ("straw", "sticks", "bricks", "rebar")],
         pending=[pend("pending", "diego", 1.0, 10.0, 100, group="g")], host_attrs=_HN4,
         groups={"g": dict(type="unique", running_hosts=["straw#nocache", "sticks#nocache", "bricks#nocache", "rebar#nocache"])},
         spare={}, params=_P05, expect_decisions=[]),
    dict(name="compute-preemption-decision attribute-equals group: az=west", ref=f"{T_REB}:602-650",
         shares={"default": dict(cpus=10.0, mem=10.0, gpus=1.0)},
         running=_pigs(_PIGS4, 100.0) + [run("cotask-steel", "diego", 1.0, 10.0, "steel")],
         pending=[pend("pending", "diego", 1.0, 10.0, 100, group="g")],
         host_attrs=dict(_AZ4, steel={"az": "west", "HOSTNAME": "steel"}),
         groups={"g": dict(type="attribute-equals", attribute="az", running_hosts=["steel"])},
         spare={}, params=_P05, expect_decision_host_in=["rebar"]),
    dict(name="compute-preemption-decision attribute-equals group: az=south has no host", ref=f"{T_REB}:652-678",
         shares={"default": dict(cpus=10.0, mem=10.0, gpus=1.0)},
         running=_pigs(_PIGS4, 100.0) + [run("cotask-steel", "diego", 1.0, 10.0, "steel")],
         pending=[pend("pending", "diego", 1.0, 10.0, 100, group="g")],
         host_attrs=dict(_AZ4, steel={"az": "south", "HOSTNAME": "steel"}),
         groups={"g": dict(type="attribute-equals", attribute="az", running_hosts=["steel"])},
         spare={}, params=_P05, expect_decisions=[]),
    _balanced_case("only north is open", ["straw", "sticks", "bricks", "rebar", "concrete", "steel", "gold", "titanium", "straw",
                                          "sticks", "bricks", "rebar", "concrete", "steel", "gold"], [], ["titanium", "rebar"], "698-738"),
    _balanced_case("titanium already preempted this cycle -> south", ["straw", "sticks", "bricks", "rebar", "concrete", "steel", "gold",
                                                                      "titanium", "straw", "sticks", "bricks", "rebar", "concrete", "steel"],
                   ["titanium"], ["bricks", "gold"], "740-786"),
    dict(name="compute-preemption-decision user over quota preempts only its own task", ref=f"{T_REB}:788-823",
         shares={"testA": dict(cpus=1.0, mem=1.0), "testB": dict(cpus=1.0, mem=1.0)}, quotas={"testA": dict(count=1)},
         running=[run("t1", "testA", 100.0, 100.0, "hostA", priority=1), run("t2", "testB", 200.0, 200.0, "hostA")],
         pending=[pend("job3", "testA", 1.0, 1.0, 3)], spare={}, params=dict(max_preemption=128, safe_dru_threshold=1.0, min_dru_diff=0.5),
         expect_decisions=[dict(job="job3", host="hostA", dru=100.0, tasks=["t1"], mem=100.0, cpus=100.0, gpus=0.0)]),
    dict(name="next-state: job9 preempts t6,t8 on hostB", ref=f"{T_REB}:895-918", shares=_DEF25, running=_RN,
         pending=[pend("job9", "wzhao", 15.0, 15.0, 9)], spare={"hostA": dict(mem=50.0, cpus=50.0)}, params=_PL,
         forced={"job9": dict(host="hostB", tasks=["t6", "t8"], mem=20.0, cpus=20.0, gpus=0.0)},
         expect_final_order=["t4", "t3", "job9", "t7", "t2", "t1", "t5"], expect_final_drus=[2.2, 1.6, 1.32, 0.72, 0.6, 0.4, 0.32]),
    dict(name="next-state: job10 preempts t2,t7 on hostA", ref=f"{T_REB}:920-944", shares=_DEF25, running=_RN,
         pending=[pend("job10", "sunil", 15.0, 15.0, 10)], spare={"hostA": dict(mem=50.0, cpus=50.0)}, params=_PL,
         forced={"job10": dict(host="hostA", tasks=["t2", "t7"], mem=65.0, cpus=65.0, gpus=0.0)},
         expect_final_order=["t4", "t3", "t8", "t6", "job10", "t1", "t5"], expect_final_drus=[2.0, 1.4, 1.12, 0.72, 0.6, 0.4, 0.32]),
    dict(name="next-state: job12 takes spare resources only (equal DRUs order by user)", ref=f"{T_REB}:946-988", shares=_DEF25, running=_RN,
         pending=[pend("job12", "sunil", 40.0, 40.0, 12)], spare={"hostA": dict(mem=50.0, cpus=50.0)}, params=_PL,
         forced={"job12": dict(host="hostA", tasks=[], mem=50.0, cpus=50.0, gpus=0.0)},
         expect_final_order=["t4", "t3", "job12", "t8", "t7", "t6", "t2", "t1", "t5"],
         expect_final_drus=[2.2, 1.6, 1.6, 1.52, 1.12, 0.72, 0.6, 0.4, 0.32]),
    dict(name="rebalance: simple test", ref=f"{T_REB}:1079-1085", shares=_SH25, running=_RL, pending=_PW, spare={}, params=_PL,
         expect_jobs_to_run=["job9", "job10", "job11"], expect_tasks_to_preempt=["t4"]),
    dict(name="rebalance: simple test with available resources", ref=f"{T_REB}:1086-1092", shares=_SH25, running=_RL, pending=_PW,
         spare={"hostB": dict(mem=0.0, cpus=10.0)}, params=_PL,
         expect_jobs_to_run=["job9", "job10", "job11", "job12", "job13"], expect_tasks_to_preempt=["t4"]),
    dict(name="rebalance: simple test 2", ref=f"{T_REB}:1093-1100", shares=_SH25, running=_RL, pending=_PS, spare={}, params=_PL,
         expect_jobs_to_run=[f"job{i}" for i in range(19, 27)], expect_tasks_to_preempt=["t4", "t3"]),
    dict(name="rebalance: simple test 2 with available resources", ref=f"{T_REB}:1101-1108", shares=_SH25, running=_RL, pending=_PS,
         spare={"hostB": dict(mem=25.0, cpus=25.0)}, params=_PL,
         expect_jobs_to_run=[f"job{i}" for i in range(19, 27)], expect_tasks_to_preempt=["t4"]),
    dict(name="rebalance: test with share change", ref=f"{T_REB}:1109-1118", shares=dict(_SH25, sunil=dict(cpus=50.0, mem=50.0)),
         running=_RL, pending=_PS, spare={}, params=_PL,
         expect_jobs_to_run=[f"job{i}" for i in range(19, 29)], expect_tasks_to_preempt=["t4", "t3", "t8"]),
]

# constraints truth tables (test/cook/test/scheduler/constraints.clj): one job, one VM with ample cpus/mem; the job is matched
# iff the constraint under test passes.  Extra job fields: equals [[attribute, pattern]], novel [hostnames], disk {request, type}
# (after the pool's defaulting / type-map, constraints.clj:105-120), est_end_ms, group; offer fields: host, attrs, disk {type: MiB},
# host_start_s, run_count (tasks running or assigned on the VM); case fields: groups, reserved_hosts, host_lifetime_mins.
T_CON = "test/cook/test/scheduler/constraints.clj"
_BIG = dict(cpus=40.0, mem=5000.0)
_UD = [["is_spot", "true"], ["instance_type", "mem.large"]]
_P100, _K80 = "nvidia-tesla-p100", "nvidia-tesla-k80"


def _one(name, ref, j, o, ok, **kw):
    return dict(name=name, ref=ref, good_enough=0.8, jobs=[dict(job("j", "u", 5.0, 5.0), **j)], offers=[dict(_BIG, **o)],
                expect_matched=(["j"] if ok else []), **kw)


CONSTRAINTS = [
    _one(f"user-defined EQUALS vs {attrs}", f"{T_CON}:43-57", dict(equals=_UD), dict(attrs=attrs), ok)
    for attrs, ok in (({"is_spot": "true", "instance_type": "mem.large"}, True), ({"is_spot": "true", "instance_type": "cpu.large"}, False),
                      ({"is_spot": "false", "instance_type": "mem.large"}, False), ({"is_spot": "true"}, False),
                      ({"instance_type": "mem.large"}, False), ({}, False))
] + [
    _one("gpu: 1 gpu asked, VM has 4 (count must match exactly)", f"{T_CON}:112-122", dict(gpus=1, gpu_model=_P100),
         dict(k8s=True, gpu_model=_P100, gpu_count=4), False),
    _one("gpu: 8 gpus asked, VM has 4", f"{T_CON}:123-132", dict(gpus=8, gpu_model=_P100), dict(k8s=True, gpu_model=_P100, gpu_count=4), False),
    _one("gpu: right count, wrong model", f"{T_CON}:133-142", dict(gpus=4, gpu_model=_K80), dict(k8s=True, gpu_model=_P100, gpu_count=4), False),
    _one("gpu: right count and model", f"{T_CON}:143-152", dict(gpus=4, gpu_model=_P100), dict(k8s=True, gpu_model=_P100, gpu_count=4), True),
    _one("gpu: one GPU job per VM (a task is already assigned)", f"{T_CON}:153-163", dict(gpus=4, gpu_model=_P100),
         dict(k8s=True, gpu_model=_P100, gpu_count=4, run_count=1), False),
    _one("gpu: non-gpu job on a gpu VM", f"{T_CON}:164-173", dict(), dict(k8s=True, gpu_model=_P100, gpu_count=4), False),
    _one("gpu: gpu job on a k8s VM without gpus", f"{T_CON}:174-183", dict(gpus=1, gpu_model=_P100), dict(k8s=True), False),
    _one("gpu: non-gpu job on a k8s VM without gpus", f"{T_CON}:184-193", dict(), dict(k8s=True), True),
    _one("gpu: gpu job on a mesos VM", f"{T_CON}:194-203", dict(gpus=1.0), dict(), False),
    _one("gpu: non-gpu job on a mesos VM", f"{T_CON}:204-213", dict(), dict(), True),
    _one("disk: enough space, right type (type-map standard -> pd-standard)", f"{T_CON}:262-271", dict(disk=dict(request=10.0, type="pd-standard")),
         dict(k8s=True, disk={"pd-standard": 50}), True),
    _one("disk: request == space, default type", f"{T_CON}:272-281", dict(disk=dict(request=50.0, type="pd-standard")),
         dict(k8s=True, disk={"pd-standard": 50}), True),
    _one("disk: not enough space", f"{T_CON}:282-291", dict(disk=dict(request=100.0, type="pd-standard")), dict(k8s=True, disk={"pd-standard": 50}), False),
    _one("disk: wrong type", f"{T_CON}:292-300", dict(disk=dict(request=10.0, type="pd-ssd")), dict(k8s=True, disk={"pd-standard": 50}), False),
    _one("rebalancer reservation: VM reserved for another job", f"{T_CON}:332-340", dict(), dict(host="hostB"), False, reserved_hosts=["hostB"]),
    _one("rebalancer reservation: VM not reserved", f"{T_CON}:341-349", dict(), dict(host="hostA"), True, reserved_hosts=["hostB"]),
    _one("estimated completion: VM without host-start-time", f"{T_CON}:432-437", dict(est_end_ms=100000), dict(), True, host_lifetime_mins=1),
    _one("estimated completion: host started at 0 s, dies at 60 s < end 100 s", f"{T_CON}:438", dict(est_end_ms=100000), dict(host_start_s=0), False,
         host_lifetime_mins=1),
    _one("estimated completion: host started at 51 s, dies at 111 s > end 100 s", f"{T_CON}:439", dict(est_end_ms=100000), dict(host_start_s=51), True,
         host_lifetime_mins=1),
]

# group host-placement through Fenzo (test/cook/test/scheduler/scheduler.clj:982-1155): dummy jobs are 1 cpu / 10 MB, test VMs
# 100 cpus / 100 000 MB, good-enough 0.8 (make-dummy-scheduler, :79-92)
_VM = dict(cpus=100.0, mem=100000.0)


def _dj(i, group=None, cpus=1.0):
    return dict(job(f"g{i}", "u", cpus, 10.0), group=group)


GROUPS_FENZO = [
    dict(name="unique group: conflicting jobs, different cycles (cotask already runs on the host)", ref=f"{T_SCHED}:986-1013", good_enough=0.8,
         jobs=[_dj(1, "G")], offers=[dict(_VM, host="test-host")], groups={"G": dict(type="unique", running_hosts=["test-host"])},
         expect_matched=[]),
    dict(name="unique group: conflicting jobs, same cycle", ref=f"{T_SCHED}:1014-1041", good_enough=0.8,
         jobs=[_dj(1, "G"), _dj(2, "G")], offers=[dict(_VM, host="test-host")], groups={"G": dict(type="unique")}, expect_matched=["g1"]),
    dict(name="no group: both jobs fit the host", ref=f"{T_SCHED}:1042-1049", good_enough=0.8, jobs=[_dj(1), _dj(2)],
         offers=[dict(_VM, host="test-host")], expect_matched=["g1", "g2"]),
    dict(name="balanced group (HOSTNAME, minimum 3): 9 jobs on 3 hosts -> 3 each", ref=f"{T_SCHED}:1055-1073", good_enough=0.8,
         jobs=[_dj(i, "B") for i in range(9)], offers=[dict(_VM, host=h) for h in ("straw", "sticks", "bricks")],
         groups={"B": dict(type="balanced", attribute="HOSTNAME", minimum=3)}, expect_matched=[f"g{i}" for i in range(9)],
         expect_counts={"straw": 3, "sticks": 3, "bricks": 3}),
    dict(name="no group: 9 jobs on 3 hosts are bin-packed, not balanced", ref=f"{T_SCHED}:1074-1086", good_enough=0.8,
         jobs=[_dj(i) for i in range(9)], offers=[dict(_VM, host=h) for h in ("straw", "sticks", "bricks")],
         expect_matched=[f"g{i}" for i in range(9)], expect_not_counts={"straw": 3, "sticks": 3, "bricks": 3}),
    dict(name="attribute-equals group: cotask runs on az=east, 20 jobs pile up on the one east VM (5 cpus)", ref=f"{T_SCHED}:1103-1136", good_enough=0.8,
         jobs=[_dj(i, "A") for i in range(20)],
         offers=[dict(cpus=1.0, mem=100000.0, host=f"w{i:02d}", attrs={"az": "west"}) for i in range(20)] + [dict(cpus=5.0, mem=100000.0, host="east1", attrs={"az": "east"})],
         groups={"A": dict(type="attribute-equals", attribute="az", running_hosts=["east0"], running_attrs=["east"])},
         expect_n_matched=5, expect_counts={"east1": 5}),
    dict(name="no group: 20 jobs use every VM", ref=f"{T_SCHED}:1137-1153", good_enough=0.8, jobs=[_dj(i) for i in range(20)],
         offers=[dict(cpus=1.0, mem=100000.0, host=f"w{i:02d}", attrs={"az": "west"}) for i in range(15)] + [dict(cpus=5.0, mem=100000.0, host="east1", attrs={"az": "east"})],
         expect_n_matched=20),
]

# handle-resource-offers! end to end (scheduler.clj:1803-2256): 8 jobs in rank order; with the pool's disk constraint enabled every job
# carries one (default request 10 000 of type standard -> pd-standard, constraints.clj:105-120)
_PDS = dict(request=10000.0, type="pd-standard")
_HRO8 = [dict(job("job-1", "u", 3, 2048), disk=_PDS), dict(job("job-2", "u", 13, 1024), disk=_PDS), dict(job("job-3", "u", 7, 4096), disk=_PDS),
         dict(job("job-4", "u", 11, 1024), disk=_PDS), dict(job("job-5", "u", 5, 2048, gpus=2, gpu_model=_P100), disk=_PDS),
         dict(job("job-6", "u", 19, 1024, gpus=4, gpu_model=_P100), disk=_PDS),
         dict(job("job-7", "u", 1, 2048), disk=dict(request=250000.0, type="pd-ssd")), dict(job("job-8", "u", 2, 2048), disk=dict(request=10000.0, type="pd-ssd"))]


def _k8s(c, m, gpus=None, disk=None, host=None):
    d = dict(cpus=c, mem=m, k8s=True, disk=disk or {"pd-standard": 512000})
    if gpus:
        d.update(gpu_model=_P100, gpu_count=gpus)
    if host:
        d.update(host=host)
    return d


_KO = {1: _k8s(10, 2048), 2: _k8s(20, 16384), 3: _k8s(30, 8192), 4: _k8s(4, 2048), 5: _k8s(4, 1024), 6: _k8s(10, 4096, gpus=2),
       7: _k8s(20, 4096, gpus=4), 8: _k8s(30, 16384, gpus=1), 9: _k8s(100, 200000), 10: _k8s(30, 2048, disk={"pd-ssd": 500000}),
       11: _k8s(30, 2048, disk={"pd-ssd": 200000})}
_MO = {i: dict(cpus=c, mem=m) for i, (c, m) in enumerate([(10, 2048), (20, 16384), (30, 8192), (4, 2048), (4, 1024), (10, 4096), (20, 4096),
                                                          (30, 16384), (100, 200000)], start=1)}


def _hro(name, line, offers, expect, jobs=None, head=None, **kw):
    d = dict(name=f"handle-resource-offers: {name}", ref=f"{T_SCHED}:{line}", good_enough=0.8, jobs=jobs or _HRO8, offers=offers,
             expect_matched=expect, **kw)
    if head is not None:
        d["expect_head_matched"] = head
    return d


HRO = [
    _hro("offer for single job", "2054-2061", [_KO[4]], ["job-1"]),
    _hro("offer for first three jobs", "2063-2070", [_KO[3]], ["job-1", "job-2", "job-3"]),
    _hro("offer not fit for any job", "2072-2078", [_KO[5]], []),
    _hro("will not launch jobs on reserved host", "2110-2118", [dict(_KO[1], host="h1")], [], reserved_hosts=["h1"]),
    _hro("only launches reserved jobs on reserved host", "2120-2137", [dict(_KO[9], host="h9")], ["job-1", "job-2"], reserved_hosts=["h9"],
         jobs=[dict(j, reserved_host="h9") if j["name"] in ("job-1", "job-2") else j for j in _HRO8]),
    _hro("mesos: all offers for all jobs (gpu jobs never match)", "2161-2168", [_MO[i] for i in range(1, 10)],
         ["job-1", "job-2", "job-3", "job-4", "job-7", "job-8"]),
    _hro("k8s: all offers for all jobs", "2199-2206", [_KO[i] for i in range(1, 12)], [f"job-{i}" for i in range(1, 9)]),
    _hro("k8s gpu offers for all gpu jobs", "2208-2215", [_KO[6], _KO[7]], ["job-5", "job-6"], head=False),
    _hro("k8s gpu offer for single gpu job", "2217-2224", [_KO[6]], ["job-5"], head=False),
    _hro("k8s gpu offer matching no gpu job", "2226-2232", [_KO[8]], [], head=True),
    _hro("disk offer matching job requesting same disk type", "2235-2242", [_KO[10]], ["job-7"], head=False),
    _hro("disk offer matching no job (K = 7)", "2244-2250", [_KO[11]], [], jobs=_HRO8[:7], head=True),
]


# ---- considerable jobs (scheduler.clj:729-762; tools.clj:903-973) -------------------------------------------------------
T_TOOLS = "test/cook/test/tools.clj"


def qj(name, user, cpus, mem, gpus=0.0, eligible=True):
    return dict(name=name, user=user, cpus=cpus, mem=mem, gpus=gpus, eligible=eligible)


_Q4 = [qj("job-1", "john", 2, 2048), qj("job-2", "john", 1, 1024), qj("job-3", "john", 3, 4096), qj("job-4", "john", 1, 1024)]
_USE = {"john": dict(count=1, cpus=2, mem=1024)}
_NG = [qj("job-1", "u", 3, 2048), qj("job-2", "u", 13, 1024), qj("job-3", "u", 7, 4096), qj("job-4", "u", 11, 1024)]
_GJ = [qj("job-5", "u", 5, 2048, gpus=2), qj("job-6", "u", 19, 1024, gpus=4)]
_UU = {"u": dict(count=1, cpus=2, mem=1024, gpus=0)}
_UQ = {"u": dict(count=10, cpus=50, mem=32768, gpus=10)}


def _cons(name, line, queue, quota, expect, k=5, usage=None, **kw):
    return dict(name=name, ref=f"{T_SCHED}:{line}", queue=queue, user_usage=usage or _UU, user_quota=quota, num_considerable=k,
                expect=expect, **kw)


CONSIDERABLE = [
    # tools.clj tests: the pool filter alone (no user quota), the user filter alone, and both (user filter first)
    dict(name="filter-based-on-pool-quota: no jobs included", ref=f"{T_TOOLS}:763-772", queue=_Q4, user_usage={}, user_quota={},
         pool_quota=dict(count=1, cpus=2, mem=1024), pool_usage=dict(count=1, cpus=2, mem=1024), num_considerable=100, expect=[]),
    dict(name="filter-based-on-pool-quota: all jobs included", ref=f"{T_TOOLS}:773-775", queue=_Q4, user_usage={}, user_quota={},
         pool_quota=dict(count=10, cpus=20, mem=32768), pool_usage=dict(count=1, cpus=2, mem=1024), num_considerable=100,
         expect=["job-1", "job-2", "job-3", "job-4"]),
    dict(name="filter-based-on-pool-quota: room for later jobs not included", ref=f"{T_TOOLS}:776-778", queue=_Q4, user_usage={},
         user_quota={}, pool_quota=dict(count=4, cpus=20, mem=6144), pool_usage=dict(count=1, cpus=2, mem=1024),
         num_considerable=100, expect=["job-1", "job-2"]),
    dict(name="filter-based-on-user-quota: no jobs included", ref=f"{T_TOOLS}:780-791", queue=_Q4, user_usage=_USE,
         user_quota={"john": dict(count=1, cpus=2, mem=1024)}, num_considerable=100, expect=[]),
    dict(name="filter-based-on-user-quota: all jobs included", ref=f"{T_TOOLS}:792-794", queue=_Q4, user_usage=_USE,
         user_quota={"john": dict(count=10, cpus=20, mem=32768)}, num_considerable=100, expect=["job-1", "job-2", "job-3", "job-4"]),
    dict(name="filter-based-on-user-quota: room for later jobs not included", ref=f"{T_TOOLS}:795-797", queue=_Q4, user_usage=_USE,
         user_quota={"john": dict(count=4, cpus=20, mem=6144)}, num_considerable=100, expect=["job-1", "job-2"]),
    dict(name="filter-pending-jobs-for-quota: user quota filters first", ref=f"{T_TOOLS}:799-817",
         queue=[qj("job-1", "john", 1, 1), qj("job-2", "john", 1, 1), qj("job-3", "john", 1, 1), qj("job-4", "bob", 1, 1)],
         user_usage={"john": dict(count=1, cpus=1, mem=1), "bob": dict(count=1, cpus=1, mem=1)},
         user_quota={"john": dict(count=2, cpus=100, mem=100), "bob": dict(count=2, cpus=100, mem=100)},
         pool_quota=dict(count=4, cpus=100, mem=100), num_considerable=100, expect=["job-1", "job-4"]),
    # scheduler.clj test-pending-jobs->considerable-jobs
    _cons("all jobs deferred by the launch plugin", "1586-1598", [dict(j, eligible=False) for j in _NG], _UQ, []),
    _cons("jobs inside usage quota", "1601-1611", _NG, _UQ, ["job-1", "job-2", "job-3", "job-4"], expect_rate_limited={}),
    _cons("gpu jobs inside usage quota", "1612-1618", _GJ, _UQ, ["job-5", "job-6"], expect_rate_limited={}),
    _cons("inside usage quota, beyond rate limit", "1620-1635", _NG, _UQ, ["job-1"], tokens={"u": 1}, enforce=True,
          expect_rate_limited={"u": 3}),
    _cons("gpu jobs inside usage quota, beyond rate limit", "1637-1644", _GJ, _UQ, ["job-5"], tokens={"u": 1}, enforce=True,
          expect_rate_limited={"u": 1}),
    _cons("num-considerable 3", "1646-1656", _NG, _UQ, ["job-1", "job-2", "job-3"], k=3),
    _cons("num-considerable 2", "1658-1668", _NG, _UQ, ["job-1", "job-2"], k=2),
    _cons("num-considerable 2, gpu jobs", "1664-1668", _GJ, _UQ, ["job-5", "job-6"], k=2),
    _cons("num-considerable 1", "1670-1680", _NG, _UQ, ["job-1"], k=1),
    _cons("num-considerable 1, gpu jobs", "1676-1680", _GJ, _UQ, ["job-5"], k=1),
    _cons("some jobs inside usage quota", "1682-1692", _NG, {"u": dict(count=5, cpus=10, mem=4096, gpus=10)}, ["job-1"]),
    _cons("some gpu jobs inside usage quota", "1688-1692", _GJ, {"u": dict(count=5, cpus=10, mem=4096, gpus=10)}, ["job-5"]),
    _cons("quota gpus not ignored", "1694-1704", _GJ, {"u": dict(count=5, cpus=10, mem=4096, gpus=0)}, []),
    _cons("all jobs exceed quota", "1706-1716", _NG, {"u": dict(count=5, cpus=3, mem=4096, gpus=10)}, []),
    _cons("all gpu jobs exceed quota", "1712-1716", _GJ, {"u": dict(count=5, cpus=3, mem=4096, gpus=10)}, []),
]


# ---- offer construction from node state (kubernetes/compute_cluster.clj:68-190) ------------------------------------------
T_KAPI = "test/cook/test/kubernetes/api.clj"
T_KCC = "test/cook/test/kubernetes/compute_cluster.clj"
P100, K80 = "nvidia-tesla-p100", "nvidia-tesla-k80"


def kpod(name, node, *requests, synthetic=False):
    """testutil.clj:527-567 pod-helper: one container per request map; gpus / disk also set the pod's nodeSelector."""
    containers, gpu_model, disk_type = [], None, None
    for r in requests:
        c = {}
        if r.get("mem") is not None:
            c["memory"] = float(r["mem"])
        if r.get("cpus") is not None:
            c["cpu"] = float(r["cpus"])
        if r.get("gpus") is not None and int(r["gpus"]) > 0:
            c["nvidia.com/gpu"] = int(r["gpus"])
            gpu_model = r.get("gpu-model") or P100
        if r.get("disk") is not None:
            c["ephemeral-storage"] = float(r["disk"].get("disk-request", 10000))
            disk_type = r["disk"].get("disk-type") or "standard"
        containers.append(c)
    return dict(name=name, node=node, containers=containers, gpu_model=gpu_model, disk_type=disk_type, synthetic=synthetic)


def knode(name, cpus, mem, gpus=None, gpu_model=None, disk=None, **flags):
    """testutil.clj:586-628 node-helper"""
    alloc = {}
    if cpus is not None:
        alloc["cpu"] = float(cpus)
    if mem is not None:
        alloc["memory"] = float(mem)
    gt = None
    if gpus is not None and gpus > 0:
        alloc["nvidia.com/gpu"] = int(gpus)
        gt = gpu_model or P100
    dt = None
    if disk is not None:
        alloc["ephemeral-storage"] = float(disk["disk-amount"])
        dt = disk.get("disk-type") or "standard"
    d = dict(name=name, allocatable=alloc, gpu_type=gt, disk_type=dt, unschedulable=False, other_taints=False,
             blocklist_label=False, gpu_taint=False)
    d.update(flags)
    return d


_AGG_PODS = [
    kpod("cook-synthetic-pod-podA", "hostA", {"cpus": 1.0, "mem": 100.0, "gpus": "2", "gpu-model": P100}, synthetic=True),
    kpod("podB", "hostA", {"cpus": 1.0}),
    kpod("podA", "hostA", {"gpus": "1", "gpu-model": P100}),
    kpod("podC", "hostB", {"cpus": 1.0}, {"mem": 100.0}),
    kpod("podC", "hostB", {"cpus": 2.0}, {"mem": 30.0, "gpus": "1", "gpu-model": K80,
                                        "disk": {"disk-request": 10.0, "disk-limit": 50.0, "disk-type": "standard"}}),
    kpod("podD", "hostC", {"cpus": 1.0, "disk": {"disk-request": 100.0, "disk-type": "pd-ssd"}}),
    kpod("podD", "hostC", {"cpus": 1.0, "disk": {"disk-type": "pd-ssd"}}),
    kpod("podE", None, {"cpus": 12.0}),
]
_AGG_REST = {"hostB": {"cpus": 3.0, "mem": 130.0, "gpus": {K80: 1}, "disk": {"standard": 10.0}},
             "hostC": {"cpus": 2.0, "mem": 0.0, "disk": {"pd-ssd": 10100.0}}}

K8S_CONSUMPTION = [
    dict(name="single pod without gpus", ref=f"{T_KAPI}:23-29", clobber=False,
         pods=[kpod("podA", "hostA", {"cpus": 1.0, "mem": 100.0})], expect={"hostA": {"cpus": 1.0, "mem": 100.0}}),
    dict(name="single pod with gpus", ref=f"{T_KAPI}:31-38", clobber=False,
         pods=[kpod("podA", "hostA", {"cpus": 1.0, "mem": 100.0, "gpus": "2", "gpu-model": P100})],
         expect={"hostA": {"cpus": 1.0, "mem": 100.0, "gpus": {P100: 2}}}),
    dict(name="multiple containers without gpus", ref=f"{T_KAPI}:40-48", clobber=False,
         pods=[kpod("podA", "hostA", {"cpus": 1.0, "mem": 100.0, "gpus": "0"}, {"cpus": 1.0, "mem": 0.0}, {"mem": 100.0})],
         expect={"hostA": {"cpus": 2.0, "mem": 200.0}}),
    dict(name="multiple containers with gpus", ref=f"{T_KAPI}:50-59", clobber=False,
         pods=[kpod("podA", "hostA", {"cpus": 1.0, "mem": 100.0, "gpus": "1", "gpu-model": P100},
                    {"cpus": 1.0, "mem": 0.0, "gpus": "4", "gpu-model": P100}, {"mem": 100.0})],
         expect={"hostA": {"cpus": 2.0, "mem": 200.0, "gpus": {P100: 5}}}),
    dict(name="aggregates pods by node name", ref=f"{T_KAPI}:61-93", clobber=False, pods=_AGG_PODS,
         expect=dict({"hostA": {"cpus": 2.0, "mem": 100.0, "gpus": {P100: 3}}}, **_AGG_REST)),
    dict(name="aggregates pods by node name, synthetic pods clobbered", ref=f"{T_KAPI}:94-98", clobber=True, pods=_AGG_PODS,
         expect=dict({"hostA": {"cpus": 1.0, "mem": 0.0, "gpus": {P100: 1}}}, **_AGG_REST)),
]

K8S_CAPACITY = [
    dict(name="capacity with and without gpus", ref=f"{T_KAPI}:100-109",
         nodes=[knode("nodeA", 1.0, 100.0, 2, P100), knode("nodeB", 1.0, None), knode("nodeC", None, 100.0, 5, P100),
                knode("nodeD", None, None, 7, P100)],
         expect={"nodeA": {"cpus": 1.0, "mem": 100.0, "gpus": {P100: 2}}, "nodeB": {"cpus": 1.0, "mem": 0.0},
                 "nodeC": {"cpus": 0.0, "mem": 100.0, "gpus": {P100: 5}}, "nodeD": {"cpus": 0.0, "mem": 0.0, "gpus": {P100: 7}}}),
    dict(name="capacity with disk", ref=f"{T_KAPI}:110-114",
         nodes=[knode("nodeA", 1.0, 100.0, 2, P100, {"disk-amount": 10000, "disk-type": "standard"}),
                knode("nodeB", 2.0, 100.0, None, None, {"disk-amount": 10000, "disk-type": "pd-ssd"})],
         expect={"nodeA": {"cpus": 1.0, "mem": 100.0, "gpus": {P100: 2}, "disk": {"standard": 10000.0}},
                 "nodeB": {"cpus": 2.0, "mem": 100.0, "disk": {"pd-ssd": 10000.0}}}),
]


def _bare(**flags):
    d = dict(name="NodeName", allocatable=None, gpu_type=None, disk_type=None, unschedulable=False, other_taints=False,
             blocklist_label=False, gpu_taint=False)
    d.update(flags)
    return d


# api/num-pods-on-node is redefined to 1 and the pod-count capacity is 30 throughout (api.clj:844); label / taint matching is
# the host's job, so each case carries the predicate values the cited form sets up
K8S_SCHEDULABLE = [
    dict(ref=f"{T_KAPI}:855", node=_bare(blocklist_label=True), filter_unsound=False, expect=False),
    dict(ref=f"{T_KAPI}:856", node=_bare(), filter_unsound=False, expect=True),
    dict(ref=f"{T_KAPI}:872", node=_bare(), filter_unsound=False, expect=True),            # the pool's own taint
    dict(ref=f"{T_KAPI}:873", node=_bare(other_taints=True), filter_unsound=False, expect=False),
    dict(ref=f"{T_KAPI}:889", node=_bare(other_taints=True), filter_unsound=False, expect=False),
    dict(ref=f"{T_KAPI}:908", node=_bare(gpu_taint=True, allocatable={"nvidia.com/gpu": 1}), filter_unsound=False, expect=True),
    dict(ref=f"{T_KAPI}:918", node=_bare(unschedulable=True), filter_unsound=False, expect=False),
    dict(ref=f"{T_KAPI}:928", node=_bare(unschedulable=None), filter_unsound=False, expect=True),
    dict(ref=f"{T_KAPI}:937", node=_bare(), filter_unsound=False, expect=True),
    dict(ref=f"{T_KAPI}:946", node=_bare(gpu_taint=True), filter_unsound=False, expect=True),
    dict(ref=f"{T_KAPI}:948", node=_bare(gpu_taint=True), filter_unsound=True, expect=False),
    dict(ref=f"{T_KAPI}:952", node=_bare(gpu_taint=True, allocatable={"nvidia.com/gpu": 0}), filter_unsound=True, expect=False),
]

_GO_NODES = [knode("nodeA", 1.0, 1000.0, 10, P100), knode("nodeB", 1.0, 1000.0, 25, P100), knode("nodeC", 1.0, 1000.0),
             knode("nodeE", 2.0, 1100.0, None, None, {"disk-amount": 256000, "disk-type": "pd-standard"}),
             knode("my.fake.host", 1.0, 1000.0)]
# pods of the test + the two starting pods that (cc/launch-tasks ...) registered for task-1 / task-2 on "my.fake.host"
# (dummy jobs of 0.1 and 0.2 cpus, 10 MiB each: add-starting-pods, compute_cluster.clj:148-170 of the test)
_GO_PODS = [
    kpod("podA", "nodeA", {"cpus": 0.25, "mem": 250.0, "gpus": "9", "gpu-model": P100}, {"cpus": 0.1, "mem": 100.0}),
    kpod("podB", "nodeA", {"cpus": 0.25, "mem": 250.0, "gpus": "1", "gpu-model": P100}),
    kpod("podC", "nodeB", {"cpus": 1.0, "mem": 1100.0, "gpus": "10", "gpu-model": P100}),
    kpod("podD", "nodeD", {"cpus": 1.0, "mem": 1100.0, "gpus": "10", "gpu-model": P100}),
    kpod("podE", "nodeE", {"cpus": 1.0, "mem": 1100.0, "disk": {"disk-request": 50, "disk-limit": 70, "disk-type": "pd-standard"}}),
    kpod("task-1", "my.fake.host", {"cpus": 0.1, "mem": 10.0}),
    kpod("task-2", "my.fake.host", {"cpus": 0.2, "mem": 10.0}),
]
_GO_A_FULL = dict(mem=1000.0, cpus=1.0, disk={}, gpus={P100: 10})

K8S_OFFERS = [
    dict(name="test-generate-offers", ref=f"{T_KCC}:120-215", nodes=_GO_NODES, pods=_GO_PODS, max_pods=3, n_offers=5,
         expect={"nodeA": dict(mem=400.0, cpus=0.4, disk={}, gpus={P100: 0}),
                 "nodeB": dict(mem=0.0, cpus=0.0, disk={}, gpus={P100: 15}),
                 "my.fake.host": dict(mem=980.0, cpus=0.7, disk={}, gpus={}),
                 "nodeE": dict(mem=0.0, cpus=1.0, disk={"pd-standard": 255950.0}, gpus={})}),
    dict(name="node with empty pod list", ref=f"{T_KCC}:221-231", nodes=_GO_NODES, pods=[p for p in _GO_PODS if p["node"] != "nodeA"],
         max_pods=3, n_offers=5, expect={"nodeA": _GO_A_FULL}),
    dict(name="node whose pods have no resource requests", ref=f"{T_KCC}:243-253", nodes=_GO_NODES,
         pods=[p for p in _GO_PODS if p["node"] != "nodeA"] +
              [dict(name="bare1", node="nodeA", containers=[], gpu_model=None, disk_type=None, synthetic=False),
               dict(name="bare2", node="nodeA", containers=[], gpu_model=None, disk_type=None, synthetic=False)],
         max_pods=3, n_offers=5, expect={"nodeA": _GO_A_FULL}),
]



# ---- why-unscheduled: the reducer of Fenzo's per-host results (fenzo_utils.clj:33-55), the reference's own six cases --------------------------------
# test/cook/test/scheduler/fenzo_utils.clj:56-100 (test-summarize-placement-failures).  A host's result = (constraint that failed or null, resources
# that failed); the test's failures carry the resource's name as their message (its helper at :36-45), which is what the reducer counts by.
# "engine": the same hosts as inputs of a real placement where Fenzo can produce them — it evaluates hard constraints only on a host whose resources fit
# (AssignableVirtualMachine.tryRequest), so a host that fails a constraint AND resources exists in the reducer's unit test only; the named resources
# are named scalars (their failure's message is the name, fenzo_utils.clj:21-45; at most COOK_MAX_SCALARS = 3 per call), "other_constraint" (not a
# constraint Cook has) stands for user_defined_constraint in the engine form.
_EX = "scheduler/test/cook/test/scheduler/fenzo_utils.clj"
EXPLAIN = [
    dict(name="empty accumulator, a host that takes the job", ref=f"{_EX}:58-59", results=[[None, []]], expect={},
         engine=dict(hosts=[dict()], scalars=[])),
    dict(name="one host short of ports", ref=f"{_EX}:61-64", results=[[None, ["ports"]]], expect={":resources": {"ports": 1}},
         engine=dict(hosts=[dict(lack=["ports"])], scalars=["ports"])),
    dict(name="one host the job ran on", ref=f"{_EX}:66-69", results=[["novel_host_constraint", []]], expect={":constraints": {"novel_host_constraint": 1}},
         engine=dict(hosts=[dict(constraint="novel_host_constraint")], scalars=[])),
    dict(name="a constraint and two resources on one host", ref=f"{_EX}:71-76", results=[["novel_host_constraint", ["cpus", "mem"]]],
         expect={":constraints": {"novel_host_constraint": 1}, ":resources": {"cpus": 1, "mem": 1}}),
    dict(name="six hosts reduced", ref=f"{_EX}:78-96",
         results=[["novel_host_constraint", []], ["other_constraint", ["cpus", "mem"]], [None, ["cpus"]], [None, ["gpus"]], [None, ["ports", "disk"]],
                  ["novel_host_constraint", ["mem", "cpus", "ports"]]],
         expect={":constraints": {"novel_host_constraint": 2, "other_constraint": 1}, ":resources": {"cpus": 3, "gpus": 1, "mem": 2, "ports": 2, "disk": 1}}),
    dict(name="the six hosts' separable results as a placement", ref=f"{_EX}:78-96 (hosts 1, 3, 4, 5 and a host failing other_constraint alone)",
         results=[["novel_host_constraint", []], ["other_constraint", []], [None, ["cpus"]], [None, ["gpus"]], [None, ["ports", "gpus"]]],
         expect={":constraints": {"novel_host_constraint": 1, "other_constraint": 1}, ":resources": {"cpus": 1, "gpus": 2, "ports": 1}},
         engine=dict(hosts=[dict(constraint="novel_host_constraint"), dict(constraint="other_constraint"), dict(lack=["cpus"]), dict(lack=["gpus"]), dict(lack=["ports", "gpus"])],
                     scalars=["cpus", "gpus", "ports"])),
]

def main():
    out = dict(rank=RANK, rank_group=RANK_GROUP, quota_group_agg=QUOTA_GROUP_AGG, match=MATCH + CONSTRAINTS + GROUPS_FENZO + HRO, rebalance=REBALANCE, considerable=CONSIDERABLE,
               offers=dict(consumption=K8S_CONSUMPTION, capacity=K8S_CAPACITY, schedulable=K8S_SCHEDULABLE, generate=K8S_OFFERS), explain=EXPLAIN)
    for k, v in out.items():
        with open(os.path.join(HERE, f"{k}.json"), "w") as f:
            json.dump(v, f, indent=1, sort_keys=True)
    print("wrote", ", ".join(f"{k}.json" for k in out))


if __name__ == "__main__":
    main()

"""Trace replay: Cook's simulator loop over the MI355X engine.

Mirrors the cycle order of the reference's simulator (scheduler/test/cook/test/zz_simulator.clj:435-549):
    submit the jobs whose submit time has come -> complete the tasks whose run time is over -> rank -> match -> launch ->
    (every `time-ms-between-rebalancing`) rebalance -> advance the clock by cycle-step-ms, until the trace is used up,
reads its input files (simulator_files/*-trace.json, *-hosts.json; the config as a dict of the .edn's keys) and returns /
writes one row per task instance in the schema of example-out-trace.csv (zz_simulator.clj:197-246), so that the analysis
notebook of the reference (simulator_files/analysis/) reads the result unchanged.

What the harness keeps for itself is what Datomic, the Mesos mock (mesos/mock.clj) and the scheduler's Clojure glue hold in the
reference: the job / instance tables, host occupancy, the head-matched feedback on the number of considerable jobs
(scheduler.clj:1613-1651).  Every ranking, placement and preemption DECISION comes from the backend: `EngineBackend` =
libcookmatch.so (no CPU fallback); the tests run the same loop over the oracle and compare the traces row by row.
"""
from __future__ import annotations

import csv
import json
import uuid as _uuid
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import _abi as A

CSV_HEADERS = ["job_id", "instance_id", "group_id", "submit_time_ms", "mesos_start_time_ms", "start_time_ms", "end_time_ms",
               "hostname", "slave_id", "status", "reason", "user", "mem", "cpus", "job_name", "requested_run_time",
               "expected_run_time", "requested_status"]

# zz_simulator.clj:73-78 default-rebalancer-config, :80-87 default-schedulers-config, :371-375 simulate's defaults
DEFAULTS = dict(cycle_step_ms=30000, time_ms_between_rebalancing=30 * 60 * 1000, max_considerable=2000, scaleback=0.95,
                floor_iterations_before_reset=1000, good_enough_fitness=1.0, safe_dru_threshold=1.0, min_dru_diff=0.5,
                max_preemption=100, default_share=dict(mem=4000.0, cpus=4.0, gpus=1.0), max_retries_default=5)


def load_trace(path: str) -> List[dict]:
    """simulator_files/*-trace.json: a list of job maps sorted by submit-time-ms (zz_simulator.clj:447-450 insists on it)."""
    with open(path) as f:
        trace = json.load(f)
    times = [j["submit-time-ms"] for j in trace]
    if times != sorted(times):
        raise ValueError("Trace jobs are expected to be sorted by submit-time-ms")
    return trace


def load_hosts(path: str) -> List[dict]:
    with open(path) as f:
        return json.load(f)


def config_from_edn_keys(cfg: dict) -> dict:
    """{:shares [...] :cycle-step-ms n :scheduler-config {:rebalancer-config {...} :fenzo-config {...}}} as a dict"""
    out = dict(DEFAULTS)
    out["default_share"] = dict(DEFAULTS["default_share"])
    if "cycle-step-ms" in cfg:
        out["cycle_step_ms"] = int(cfg["cycle-step-ms"])
    if "time-ms-between-rebalancing" in cfg:
        out["time_ms_between_rebalancing"] = int(cfg["time-ms-between-rebalancing"])
    for s in cfg.get("shares", []):
        if s.get("user") == "default":
            out["default_share"] = {k: float(s[k]) for k in ("mem", "cpus", "gpus") if k in s}
        else:
            out.setdefault("user_shares", {})[s["user"]] = {k: float(s[k]) for k in ("mem", "cpus", "gpus") if k in s}
    sc = cfg.get("scheduler-config", {})
    rb = sc.get("rebalancer-config", {})
    for k_edn, k in (("max-preemption", "max_preemption"), ("safe-dru-threshold", "safe_dru_threshold"), ("min-dru-diff", "min_dru_diff")):
        if k_edn in rb:
            out[k] = rb[k_edn]
    fz = sc.get("fenzo-config", {})
    if "fenzo-max-jobs-considered" in fz:
        out["max_considerable"] = int(fz["fenzo-max-jobs-considered"])
    if "good-enough-fitness" in fz:
        out["good_enough_fitness"] = float(fz["good-enough-fitness"])
    return out


def _resource(job: dict, kind: str) -> float:
    for r in job.get("job/resource", []):
        if r["resource/type"] == "resource.type/" + kind:
            return float(r["resource/amount"])
    return 0.0


@dataclass
class _Job:
    idx: int                      # :db/id order = submission order
    uuid: str
    user: str
    cpus: float
    mem: float
    priority: int
    run_time_ms: int
    status: str                   # requested status: "finished" | "failed"
    name: str
    group: str
    expected_runtime: Optional[int]
    max_retries: int
    submit_ms: int = 0
    state: str = "waiting"        # waiting | running | completed
    instances: List[dict] = field(default_factory=list)


class EngineBackend:
    """The product path: every decision through libcookmatch.so (one engine = one pool)."""

    def __init__(self, engine):
        self.e = engine

    def rank(self, params, tasks, users):
        self.e.set_params(params)
        return self.e.rank(tasks, users, want_dru=False)[0]

    def match(self, params, jobs, offers):
        self.e.set_params(params)
        return self.e.match(jobs, offers)[0]

    def rebalance(self, params, running, pending, job_ids, priorities, users, spare, rparams):
        self.e.set_params(params)
        return self.e.rebalance(running, pending, job_ids, priorities, users, spare, rparams)["decisions"]


class Simulator:
    def __init__(self, trace: List[dict], hosts: List[dict], config: dict, backend, offer_order: str = "descending"):
        self.cfg = config_from_edn_keys(config) if any("-" in k for k in config) else {**DEFAULTS, **config}
        self.backend = backend
        self.trace = trace
        hosts = sorted(hosts, key=lambda h: h["hostname"])  # host ids = hostname ranks (cookmatch.h)
        self.host_names = [h["hostname"] for h in hosts]
        self.slave_ids = [h.get("slave-id", "") for h in hosts]
        self.host_cpus = np.array([float(h["resources"]["cpus"]["*"]) for h in hosts])
        self.host_mem = np.array([float(h["resources"]["mem"]["*"]) for h in hosts])
        self.used_cpus = np.zeros(len(hosts))
        self.used_mem = np.zeros(len(hosts))
        self.count = np.zeros(len(hosts), np.int32)
        # offer index -> host id (see _offers): "descending" hostname order is the binding's contract; "ascending" exists
        # so that a test can show what the other order does to the recorded run
        self.offer_order = np.arange(len(hosts))[::-1].copy() if offer_order == "descending" else np.arange(len(hosts))
        self.user_names = sorted({j["job/user"] for j in trace})  # user ids = name ranks
        self.uid = {u: i for i, u in enumerate(self.user_names)}
        self.jobs: List[_Job] = []
        self.next_task_id = 1
        self.num_considerable = self.cfg["max_considerable"]
        self.floor_iterations = 0
        self.cycles = 0
        self.log: List[dict] = []     # per cycle: what was submitted / completed / matched / preempted
        self.params = A.default_params(good_enough_fitness=self.cfg["good_enough_fitness"])

    # ---- tables -> SoA -----------------------------------------------------------------------------------------------------
    def _users(self) -> A.Users:
        sh = self.cfg["default_share"]
        us = self.cfg.get("user_shares", {})
        col = lambda k: np.array([us.get(u, sh).get(k, sh.get(k, A.DMAX)) for u in self.user_names])  # noqa: E731
        return A.Users(div_cpus=col("cpus"), div_mem=col("mem"), div_gpus=col("gpus"))

    def _tasks(self):
        """running instances ++ the synthetic tasks of the waiting jobs (tools.clj:582-588), plus the row -> job map"""
        rows = []
        for j in self.jobs:
            if j.state == "running":
                inst = j.instances[-1]
                rows.append((j, 0, inst["start_ms"], inst["task_id"], inst["host"]))
            elif j.state == "waiting":
                rows.append((j, 1, 0, 0, 0))
        t = A.Tasks(cpus=[r[0].cpus for r in rows], mem=[r[0].mem for r in rows], user=[self.uid[r[0].user] for r in rows],
                    priority=[r[0].priority for r in rows], start_ms=[r[2] for r in rows], task_id=[r[3] for r in rows],
                    job_id=[r[0].idx for r in rows], pending=[r[1] for r in rows], host=[r[4] for r in rows])
        return t, [r[0] for r in rows]

    def _offers(self) -> A.Offers:
        """what the Mesos mock offers: the unused part of every host; Fenzo's view of the tasks it placed there.
        Offer ORDER: the engine breaks fitness ties by the lowest offer index; the recorded run of the reference
        (simulator_files/example-out-trace.csv: five identical hosts) resolves them towards the LAST hostname, so the offers are
        presented in descending hostname order (`offer_order`; the host ids in the `host` column stay the name ranks)."""
        o = self.offer_order
        return A.Offers(cpus=(self.host_cpus - self.used_cpus)[o], mem=(self.host_mem - self.used_mem)[o], host=o.astype(np.uint32),
                        run_cpus=self.used_cpus[o], run_mem=self.used_mem[o], run_count=self.count[o])

    def _jobs_soa(self, jobs: List[_Job]) -> A.Jobs:
        novel = [sorted({i["host"] for i in j.instances if i["reason"] != "preempted-by-rebalancer"}) for j in jobs]
        return A.Jobs.with_constraints(np.array([j.cpus for j in jobs]), np.array([j.mem for j in jobs]),
                                       novel=novel, equals=[[] for _ in jobs], user=np.array([self.uid[j.user] for j in jobs], np.uint32))

    # ---- one cycle (zz_simulator.clj:435-549) ---------------------------------------------------------------------------------
    def _submit(self, now: int) -> int:
        n = 0
        while self.trace and self.trace[0]["submit-time-ms"] <= now:
            t = self.trace.pop(0)
            n += 1
            self.jobs.append(_Job(idx=len(self.jobs) + 1, uuid=t["job/uuid"], user=t["job/user"], cpus=_resource(t, "cpus"),
                                  mem=_resource(t, "mem"), priority=int(t.get("job/priority", 50)), run_time_ms=int(t["run-time-ms"]),
                                  status=t.get("status", "finished"), name=t.get("job/name", ""), group=t.get("job/group", ""),
                                  expected_runtime=t.get("job/expected-runtime"),
                                  max_retries=int(t.get("job/max-retries", self.cfg["max_retries_default"])),
                                  submit_ms=now + n))  # the clock is advanced 1 ms per submitted job (:455-457)
        return n

    def _finish(self, job: _Job, now: int, status: str, reason: str):
        inst = job.instances[-1]
        inst.update(end_ms=now, status=status, reason=reason)
        h = inst["host"]
        self.used_cpus[h] -= job.cpus
        self.used_mem[h] -= job.mem
        self.count[h] -= 1
        if status == "success":
            job.state = "completed"
        else:  # failed / preempted: back to waiting while attempts remain (mea-culpa failures are not consumed)
            attempts = sum(1 for i in job.instances if i["reason"] != "preempted-by-rebalancer")
            job.state = "waiting" if attempts < job.max_retries else "completed"

    def _complete(self, now: int) -> int:
        n = 0
        for j in self.jobs:
            if j.state == "running" and j.instances[-1]["start_ms"] + j.run_time_ms <= now:
                ok = j.status != "failed"
                self._finish(j, j.instances[-1]["start_ms"] + j.run_time_ms, "success" if ok else "failed", "" if ok else "command-failed")
                n += 1
        return n

    def step(self, now: int, rebalance: bool) -> dict:
        rec = dict(time=now, submitted=self._submit(now), completed=self._complete(now), matched=0, preempted=0, considerable=0)
        users = self._users()
        tasks, row_job = self._tasks()
        # rank
        ranked = self.backend.rank(self.params, tasks, users) if tasks.n else np.zeros(0, np.uint32)
        queue = [row_job[i] for i in ranked]
        # match: the first num-considerable ranked jobs against the hosts' free resources
        considerable = queue[: self.num_considerable]
        rec["considerable"] = len(considerable)
        matched_head = True
        if considerable:
            j2o = self.backend.match(self.params, self._jobs_soa(considerable), self._offers())
            matched_head = bool(j2o[0] >= 0) or not (j2o >= 0).any()  # scheduler.clj:1495
            for job, v in zip(considerable, j2o):
                if v < 0:
                    continue
                v = int(self.offer_order[v])  # offer index -> host id
                job.state = "running"
                job.instances.append(dict(task_id=self.next_task_id, instance_id=str(_uuid.UUID(int=self.next_task_id)), host=int(v),
                                          start_ms=now, end_ms=None, status="running", reason=""))
                self.next_task_id += 1
                self.used_cpus[v] += job.cpus
                self.used_mem[v] += job.mem
                self.count[v] += 1
                rec["matched"] += 1
        # head-matched feedback on the number of considerable jobs (scheduler.clj:1613-1651)
        nxt = self.cfg["max_considerable"] if matched_head else max(1, int(self.cfg["scaleback"] * self.num_considerable))
        self.floor_iterations = self.floor_iterations + 1 if nxt == 1 else 0
        self.num_considerable = self.cfg["max_considerable"] if self.floor_iterations >= self.cfg["floor_iterations_before_reset"] else nxt
        # rebalance (rebalancer.clj:559-597): decisions for the first max-preemption waiting jobs in rank order
        if rebalance:
            tasks, row_job = self._tasks()
            run_rows = [i for i in range(tasks.n) if not tasks.pending[i]]
            ranked = self.backend.rank(self.params, tasks, users) if tasks.n else np.zeros(0, np.uint32)
            pend = [row_job[i] for i in ranked]
            if run_rows and pend:
                sel = np.array(run_rows)
                running = A.Tasks(cpus=tasks.cpus[sel], mem=tasks.mem[sel], user=tasks.user[sel], priority=tasks.priority[sel],
                                  start_ms=tasks.start_ms[sel], task_id=tasks.task_id[sel], job_id=tasks.job_id[sel],
                                  pending=np.zeros(len(sel), np.uint8), host=tasks.host[sel])
                free_c, free_m = self.host_cpus - self.used_cpus, self.host_mem - self.used_mem
                has = np.nonzero((free_c > 0) | (free_m > 0))[0]
                spare = A.HostSpare(host=has, cpus=free_c[has], mem=free_m[has])
                rp = A.CookRebalanceParams(float(self.cfg["safe_dru_threshold"]), float(self.cfg["min_dru_diff"]),
                                           int(self.cfg["max_preemption"]), 0)
                decisions = self.backend.rebalance(self.params, running, self._jobs_soa(pend), [j.idx for j in pend],
                                                   [j.priority for j in pend], users, spare, rp)
                for d in decisions:
                    for t in d["tasks"]:
                        if t == A.NONE_U32:
                            continue  # a job placed earlier in this call: nothing to kill (rebalancer.clj:529)
                        job = row_job[run_rows[t]]
                        if job.state == "running":
                            self._finish(job, now, "failed", "preempted-by-rebalancer")
                            rec["preempted"] += 1
        self.cycles += 1
        self.log.append(rec)
        return rec

    def run(self, max_cycles: int = 10 ** 9) -> List[dict]:
        """the loop of zz_simulator.clj:435-549: ends with the cycle that submits the last job of the trace"""
        if not self.trace:
            return self.rows()
        now = self.trace[0]["submit-time-ms"]
        since_rebalance = 0
        while self.cycles < max_cycles:
            do_rb = since_rebalance > self.cfg["time_ms_between_rebalancing"]
            more = bool(self.trace)  # (when (seq trace) (recur ...)) tests the trace BEFORE this cycle's batch is dropped (:535):
            self.step(now, do_rb)    # the loop runs one last cycle with nothing left to submit
            if not more:
                break
            since_rebalance = 0 if do_rb else since_rebalance + self.cfg["cycle_step_ms"]
            now += self.cfg["cycle_step_ms"]
        return self.rows()

    # ---- output: one row per task instance, the columns of dump-jobs-to-csv (zz_simulator.clj:235-246) ---------------------------
    def rows(self) -> List[dict]:
        out = []
        for j in self.jobs:
            for i in j.instances:
                out.append(dict(job_id=j.uuid, instance_id=i["instance_id"], group_id=j.group, submit_time_ms=j.submit_ms,
                                mesos_start_time_ms=i["start_ms"], start_time_ms=i["start_ms"], end_time_ms=i["end_ms"] if i["end_ms"] is not None else "",
                                hostname=self.host_names[i["host"]], slave_id=self.slave_ids[i["host"]],
                                status=":instance.status/" + i["status"], reason=i["reason"], user=j.user, mem=j.mem, cpus=j.cpus,
                                job_name=j.name, requested_run_time=j.run_time_ms,
                                expected_run_time=j.expected_runtime if j.expected_runtime is not None else "", requested_status=j.status))
        return out

    def write_csv(self, path: str):
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=CSV_HEADERS)
            w.writeheader()
            w.writerows(self.rows())


def simulate(trace, hosts, config, backend, max_cycles: int = 10 ** 9, offer_order: str = "descending") -> Simulator:
    sim = Simulator([dict(j) for j in trace], hosts, config, backend, offer_order=offer_order)
    sim.run(max_cycles)
    return sim

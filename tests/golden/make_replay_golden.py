#!/usr/bin/env python3
"""Builds tests/golden/replay_example.json from the reference's OWN recorded simulator run
(/root/reference/scheduler/simulator_files/: example-trace.json + example-hosts.json + example-config.edn in,
example-out-trace.csv out — the output of the real Clojure scheduler with the real Fenzo over that trace).

The fixture holds what the replay needs of the inputs (per job: uuid, user, cpus, mem, priority, submit / run time; per host:
name, slave id, cpus, mem; the config's keys) and, per task row of the recorded output: hostname, status, and the cycle (of
cycle-step-ms = 30 s) in which the task started / was seen finished.  Times are compared by cycle because the reference's
frozen clock advances 1 ms per operation inside a cycle (zz_simulator.clj:455-480), which is bookkeeping, not a decision.
Run here (the reference checkout is not present on the GPU box): python tests/golden/make_replay_golden.py"""
import csv
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scheduler/simulator_files"
STEP = 30000  # :cycle-step-ms of example-config.edn


def main():
    trace = json.load(open(os.path.join(REF, "example-trace.json")))
    hosts = json.load(open(os.path.join(REF, "example-hosts.json")))
    rows = list(csv.DictReader(open(os.path.join(REF, "example-out-trace.csv"))))
    t0 = min(int(r["start_time_ms"]) for r in rows)
    first_cycle = None
    expect = {}
    for r in rows:
        start_c = round((int(r["start_time_ms"]) - t0) / STEP)
        # a task still running when the simulation ends carries the dump time as its end: not a decision
        end_c = round((int(r["end_time_ms"]) - t0) / STEP) if r["end_time_ms"] and not r["status"].endswith("running") else None
        expect[r["job_id"]] = dict(hostname=r["hostname"], slave_id=r["slave_id"], status=r["status"], user=r["user"],
                                   mem=float(r["mem"]), cpus=float(r["cpus"]), start_cycle=start_c, end_cycle=end_c)
    keep = ("run-time-ms", "submit-time-ms", "job/priority", "job/resource", "job/max-retries", "job/name", "job/uuid", "job/user",
            "job/expected-runtime", "job/group", "status")
    out = dict(
        source="scheduler/simulator_files/example-{trace.json,hosts.json,config.edn,out-trace.csv} of the reference",
        config={"shares": [{"user": "default", "mem": 60000.0, "cpus": 600.0, "gpus": 1.0}], "cycle-step-ms": STEP,
                "scheduler-config": {"rebalancer-config": {"max-preemption": 10.0}, "fenzo-config": {"fenzo-max-jobs-considered": 200}}},
        trace=[{k: j[k] for k in keep if k in j} for j in trace],
        hosts=[{"hostname": h["hostname"], "slave-id": h["slave-id"],
                "resources": {"cpus": {"*": h["resources"]["cpus"]["*"]}, "mem": {"*": h["resources"]["mem"]["*"]}}} for h in hosts],
        expect=expect)
    with open(os.path.join(HERE, "replay_example.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote replay_example.json:", len(out["trace"]), "jobs,", len(out["hosts"]), "hosts,", len(expect), "recorded task rows")


if __name__ == "__main__":
    main()

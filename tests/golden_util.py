"""Turns the transcribed reference test vectors (tests/golden/*.json) into the SoA inputs of the C ABI."""
from __future__ import annotations

import json
import os

import numpy as np

from cook_amd import _abi as A

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def _v(x):
    return A.DMAX if x == "MAX" else float(x)


def build_rank_inputs(jobs, shares, quotas=None):
    """-> (Tasks, Users, names list aligned with task index, user names sorted)."""
    unames = sorted({j["user"] for j in jobs} | set(shares.keys()))
    uid = {u: i for i, u in enumerate(unames)}
    n = len(jobs)
    # creation order: job ids by listing order; instances by inst_seq (default: listing order among running)
    run_order = sorted([i for i, j in enumerate(jobs) if j["running"]], key=lambda i: jobs[i].get("inst_seq", i + 1))
    inst_rank = {i: r for r, i in enumerate(run_order)}
    tasks = A.Tasks(
        cpus=np.array([j["cpus"] for j in jobs], dtype=np.float64),
        mem=np.array([j["mem"] for j in jobs], dtype=np.float64),
        gpus=np.array([j.get("gpus", 0.0) for j in jobs], dtype=np.float64),
        user=np.array([uid[j["user"]] for j in jobs], dtype=np.uint32),
        priority=np.array([j.get("priority", 50) for j in jobs], dtype=np.int32),
        start_ms=np.array([1_600_000_000_000 + inst_rank.get(i, 0) for i in range(n)], dtype=np.int64),
        task_id=np.array([17_592_186_050_000 + inst_rank.get(i, 0) for i in range(n)], dtype=np.int64),
        job_id=np.array([17_592_186_045_000 + i for i in range(n)], dtype=np.int64),
        pending=np.array([0 if j["running"] else 1 for j in jobs], dtype=np.uint8),
    )
    quotas = quotas or {}
    users = A.Users(
        div_cpus=np.array([_v(shares.get(u, {}).get("cpus", "MAX")) for u in unames]),
        div_mem=np.array([_v(shares.get(u, {}).get("mem", "MAX")) for u in unames]),
        div_gpus=np.array([_v(shares.get(u, {}).get("gpus", "MAX")) for u in unames]),
        quota_count=np.array([_v(quotas.get(u, {}).get("count", 2.0 ** 31 - 1)) for u in unames]),
        quota_cpus=np.array([_v(quotas.get(u, {}).get("cpus", "MAX")) for u in unames]),
        quota_mem=np.array([_v(quotas.get(u, {}).get("mem", "MAX")) for u in unames]),
        quota_gpus=np.array([_v(quotas.get(u, {}).get("gpus", "MAX")) for u in unames]),
    )
    return tasks, users, [j["name"] for j in jobs], unames


def usage_of(q):
    return A.CookUsage(float(q.get("count", 0)), float(q.get("cpus", 0)), float(q.get("mem", 0)), float(q.get("gpus", 0)))


def build_match_inputs(case):
    jobs, offers = case["jobs"], case["offers"]
    models, locs = {}, {}

    def intern(table, key):
        if key is None:
            return 0
        return table.setdefault(key, len(table) + 1)

    K = len(jobs)
    J = A.Jobs(
        cpus=np.array([float(j["cpus"]) for j in jobs], dtype=np.float64),
        mem=np.array([float(j["mem"]) for j in jobs], dtype=np.float64),
        gpus=np.array([float(j.get("gpus", 0.0)) for j in jobs], dtype=np.float64),
        gpu_model=np.array([intern(models, j.get("gpu_model")) for j in jobs], dtype=np.uint32),
        ckpt_location=np.array([intern(locs, j.get("ckpt_location")) for j in jobs], dtype=np.uint32),
    ) if K else A.Jobs(cpus=np.zeros(0), mem=np.zeros(0))
    M = len(offers)
    O = A.Offers(
        cpus=np.array([float(o["cpus"]) for o in offers], dtype=np.float64),
        mem=np.array([float(o["mem"]) for o in offers], dtype=np.float64),
        k8s=np.array([1 if o.get("k8s") else 0 for o in offers], dtype=np.uint8),
        gpu_model=np.array([intern(models, o.get("gpu_model")) for o in offers], dtype=np.uint32),
        gpu_count=np.array([float(o.get("gpu_count", 0.0)) for o in offers], dtype=np.float64),
        location=np.array([intern(locs, o.get("location")) for o in offers], dtype=np.uint32),
    )
    return J, O, [j["name"] for j in jobs]

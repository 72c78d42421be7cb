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
    """-> (Jobs, Offers, job names) — plus, for the extended vectors, build_match_extras(case)."""
    J, O, names, _ = build_match_all(case)
    return J, O, names


def build_match_all(case):
    """-> (Jobs, Offers, job names, extras) with extras = dict(groups, reserved (host ids), host_names, params overrides)."""
    jobs, offers = case["jobs"], case["offers"]
    models, locs, dtypes, keys, vals = {}, {}, {}, {}, {}

    def intern(table, key):
        if key is None:
            return 0
        return table.setdefault(key, len(table) + 1)

    # hosts: named offers first, then hosts that only appear in constraints; ids = name ranks (cookmatch.h contract)
    hnames = [o.get("host", f"~offer{i:04d}") for i, o in enumerate(offers)]
    extra_hosts = set(case.get("reserved_hosts", []))
    for j in jobs:
        extra_hosts |= set(j.get("novel", []))
        if j.get("reserved_host"):
            extra_hosts.add(j["reserved_host"])
    for gdef in case.get("groups", {}).values():
        extra_hosts |= set(gdef.get("running_hosts", []))
    all_hosts = sorted(set(hnames) | extra_hosts)
    hid = {h: i for i, h in enumerate(all_hosts)}
    for o in offers:
        for k_ in o.get("attrs", {}):
            keys.setdefault(k_, len(keys))
    for j in jobs:
        for k_, _v in j.get("equals", []):
            keys.setdefault(k_, len(keys))
    for gdef in case.get("groups", {}).values():
        if gdef.get("attribute") not in (None, "HOSTNAME"):
            keys.setdefault(gdef["attribute"], len(keys))
    gnames = sorted(case.get("groups", {}))
    gid = {n: i for i, n in enumerate(gnames)}
    K = len(jobs)
    if K:
        J = A.Jobs.with_constraints(
            np.array([float(j["cpus"]) for j in jobs], dtype=np.float64), np.array([float(j["mem"]) for j in jobs], dtype=np.float64),
            equals=[[(keys[k_], intern(vals, v_)) for k_, v_ in j.get("equals", [])] for j in jobs],
            novel=[[hid[h] for h in j.get("novel", [])] for j in jobs],
            gpus=np.array([float(j.get("gpus", 0.0)) for j in jobs], dtype=np.float64),
            gpu_model=np.array([intern(models, j.get("gpu_model")) for j in jobs], dtype=np.uint32),
            ckpt_location=np.array([intern(locs, j.get("ckpt_location")) for j in jobs], dtype=np.uint32),
            group=np.array([gid[j["group"]] if j.get("group") else A.NONE_U32 for j in jobs], dtype=np.uint32),
            reserved_host=np.array([hid[j["reserved_host"]] if j.get("reserved_host") else -1 for j in jobs], dtype=np.int32),
            est_end_ms=np.array([int(j.get("est_end_ms", 0)) for j in jobs], dtype=np.int64),
            disk_request=np.array([float(j["disk"]["request"]) if j.get("disk") else -1.0 for j in jobs], dtype=np.float64),
            disk_type=np.array([intern(dtypes, j["disk"]["type"]) if j.get("disk") else 0 for j in jobs], dtype=np.uint32))
    else:
        J = A.Jobs(cpus=np.zeros(0), mem=np.zeros(0))
    M = len(offers)
    n_keys = max(1, len(keys))
    attr = np.zeros((M, n_keys), dtype=np.uint32)
    for i, o in enumerate(offers):
        for k_, v_ in o.get("attrs", {}).items():
            attr[i, keys[k_]] = intern(vals, v_)

    def one_disk(o):
        d = o.get("disk") or {}
        assert len(d) <= 1
        return next(iter(d.items())) if d else (None, 0.0)

    O = A.Offers(
        cpus=np.array([float(o["cpus"]) for o in offers], dtype=np.float64),
        mem=np.array([float(o["mem"]) for o in offers], dtype=np.float64),
        host=np.array([hid[h] for h in hnames], dtype=np.uint32),
        k8s=np.array([1 if o.get("k8s") else 0 for o in offers], dtype=np.uint8),
        gpu_model=np.array([intern(models, o.get("gpu_model")) for o in offers], dtype=np.uint32),
        gpu_count=np.array([float(o.get("gpu_count", 0.0)) for o in offers], dtype=np.float64),
        disk_type=np.array([intern(dtypes, one_disk(o)[0]) for o in offers], dtype=np.uint32),
        disk_space=np.array([float(one_disk(o)[1]) for o in offers], dtype=np.float64),
        attr=attr,
        location=np.array([intern(locs, o.get("location")) for o in offers], dtype=np.uint32),
        host_start_s=np.array([int(o["host_start_s"]) if "host_start_s" in o else -1 for o in offers], dtype=np.int64),
        run_count=np.array([int(o.get("run_count", 0)) for o in offers], dtype=np.int32),
    )
    G = None
    if gnames:
        gd = case["groups"]

        def key_of(g_):
            a_ = gd[g_].get("attribute")
            return A.NONE_U32 if a_ in (None, "HOSTNAME") else keys[a_]

        G = A.Groups(type=np.array([GROUP_TYPE[gd[g_]["type"]] for g_ in gnames], dtype=np.uint8),
                     attr_key=np.array([key_of(g_) for g_ in gnames], dtype=np.uint32),
                     minimum=np.array([gd[g_].get("minimum", 0) for g_ in gnames], dtype=np.int32),
                     run_hosts=[[hid[h] for h in gd[g_].get("running_hosts", [])] for g_ in gnames],
                     run_attrs=[[intern(vals, v_) for v_ in gd[g_].get("running_attrs", [])] or [0] * len(gd[g_].get("running_hosts", []))
                                for g_ in gnames])
    extras = dict(groups=G, reserved=[hid[h] for h in case.get("reserved_hosts", [])], host_names=hnames,
                  params=dict(host_lifetime_mins=int(case["host_lifetime_mins"])) if "host_lifetime_mins" in case else {})
    return J, O, [j["name"] for j in jobs], extras


def check_match_expectations(case, names, host_names, j2o, head):
    """The reference's assertions on a placement: which jobs matched, how many per host, the head-matched flag."""
    matched = sorted(names[k] for k in range(len(names)) if j2o[k] >= 0)
    if "expect_matched" in case and "expect_n_matched" not in case:
        assert matched == sorted(case["expect_matched"]), (case["name"], case["ref"], matched)
    if "expect_n_matched" in case:
        assert len(matched) == case["expect_n_matched"], (case["name"], case["ref"], matched)
    counts = {}
    for k in range(len(names)):
        if j2o[k] >= 0:
            counts[host_names[j2o[k]]] = counts.get(host_names[j2o[k]], 0) + 1
    if "expect_counts" in case:
        assert counts == case["expect_counts"], (case["name"], case["ref"], counts)
    if "expect_not_counts" in case:
        assert counts != case["expect_not_counts"], (case["name"], case["ref"], counts)
    if "expect_head_matched" in case:
        assert bool(head) == case["expect_head_matched"], (case["name"], case["ref"])
    if "expect_offers_used" in case:
        assert len(counts) == case["expect_offers_used"], (case["name"], case["ref"])
    if "expect_assignment" in case:
        for n_, o_ in case["expect_assignment"].items():
            assert j2o[names.index(n_)] == o_, (case["name"], case["ref"])


GROUP_TYPE = {"all": 0, "unique": 1, "balanced": 2, "attribute-equals": 3}


def build_rebalance_inputs(case):
    """-> dict of everything cook_rebalance / oracle_rebalance take, plus name tables (tests/golden/rebalance.json)."""
    running, pending = case["running"], case["pending"]
    shares, quotas = case["shares"], case.get("quotas", {})
    unames = sorted(({t["user"] for t in running} | {j["user"] for j in pending} | set(shares) | set(quotas)) - {"default"})
    uid = {u: i for i, u in enumerate(unames)}
    groups = case.get("groups", {})
    hnames = {t["host"] for t in running} | set(case.get("spare", {})) | set(case.get("host_attrs", {}))
    hnames |= {h for j in pending for h in j.get("novel", [])} | set(case.get("init_preempted_hosts", []))
    hnames |= {h for g in groups.values() for h in g.get("running_hosts", [])}
    hnames = sorted(hnames)
    hid = {h: i for i, h in enumerate(hnames)}
    R, P = len(running), len(pending)
    tasks = A.Tasks(
        cpus=np.array([t["cpus"] for t in running], dtype=np.float64), mem=np.array([t["mem"] for t in running], dtype=np.float64),
        gpus=np.array([t.get("gpus", 0.0) for t in running], dtype=np.float64),
        user=np.array([uid[t["user"]] for t in running], dtype=np.uint32),
        priority=np.array([t.get("priority", 50) for t in running], dtype=np.int32),
        start_ms=np.array([1_600_000_000_000 + i for i in range(R)], dtype=np.int64),
        task_id=np.array([17_592_186_050_000 + i for i in range(R)], dtype=np.int64),
        job_id=np.array([17_592_186_045_000 + i for i in range(R)], dtype=np.int64),
        pending=np.zeros(R, dtype=np.uint8), host=np.array([hid[t["host"]] for t in running], dtype=np.uint32))

    def share(u, k):
        return _v(shares.get(u, shares.get("default", {})).get(k, "MAX"))

    def quota(u, k, dflt):
        return _v(quotas.get(u, {}).get(k, dflt))

    users = A.Users(div_cpus=np.array([share(u, "cpus") for u in unames]), div_mem=np.array([share(u, "mem") for u in unames]),
                    div_gpus=np.array([share(u, "gpus") for u in unames]),
                    quota_count=np.array([quota(u, "count", 2.0 ** 31 - 1) for u in unames]),
                    quota_cpus=np.array([quota(u, "cpus", "MAX") for u in unames]),
                    quota_mem=np.array([quota(u, "mem", "MAX") for u in unames]),
                    quota_gpus=np.array([quota(u, "gpus", "MAX") for u in unames]))
    # attribute interning: key ids by first appearance (HOSTNAME is implicit), value id 0 = absent
    keys, vals = {}, {}
    host_attrs = case.get("host_attrs")
    for m in (host_attrs or {}).values():
        for k_, v_ in m.items():
            if k_ != "HOSTNAME":
                keys.setdefault(k_, len(keys))
                vals.setdefault(v_, len(vals) + 1)
    attrs = None
    if host_attrs is not None:
        hs = sorted(host_attrs)
        attr = np.zeros((len(hs), max(1, len(keys))), dtype=np.uint32)
        for r, h in enumerate(hs):
            assert host_attrs[h].get("HOSTNAME", h) == h
            for k_, v_ in host_attrs[h].items():
                if k_ != "HOSTNAME":
                    attr[r, keys[k_]] = vals[v_]
        attrs = A.Offers(cpus=np.zeros(len(hs)), mem=np.zeros(len(hs)), host=np.array([hid[h] for h in hs], dtype=np.uint32), attr=attr)
    gnames = sorted(groups)
    gid = {g: i for i, g in enumerate(gnames)}
    G = None
    if gnames:
        G = A.Groups(type=np.array([GROUP_TYPE[groups[g]["type"]] for g in gnames], dtype=np.uint8),
                     attr_key=np.array([keys[groups[g]["attribute"]] if "attribute" in groups[g] else A.NONE_U32 for g in gnames],
                                       dtype=np.uint32),
                     minimum=np.array([groups[g].get("minimum", 0) for g in gnames], dtype=np.int32),
                     run_hosts=[[hid[h] for h in groups[g].get("running_hosts", [])] for g in gnames])
    jobs = A.Jobs.with_constraints(
        np.array([j["cpus"] for j in pending], dtype=np.float64), np.array([j["mem"] for j in pending], dtype=np.float64),
        novel=[[hid[h] for h in j.get("novel", [])] for j in pending],
        gpus=np.array([j.get("gpus", 0.0) for j in pending], dtype=np.float64),
        user=np.array([uid[j["user"]] for j in pending], dtype=np.uint32),
        group=np.array([gid[j["group"]] if j.get("group") else A.NONE_U32 for j in pending], dtype=np.uint32))
    spare = case.get("spare", {})
    sh = sorted(spare)
    S = A.HostSpare(host=np.array([hid[h] for h in sh], dtype=np.uint32), cpus=np.array([spare[h].get("cpus", 0.0) for h in sh], dtype=np.float64),
                    mem=np.array([spare[h].get("mem", 0.0) for h in sh], dtype=np.float64),
                    gpus=np.array([spare[h].get("gpus", 0.0) for h in sh], dtype=np.float64))
    prm = case["params"]
    rp = A.CookRebalanceParams(_v(prm["safe_dru_threshold"]), _v(prm["min_dru_diff"]), int(prm["max_preemption"]), 0)
    tname = {t["name"]: i for i, t in enumerate(running)}
    pname = {j["name"]: i for i, j in enumerate(pending)}
    forced = None
    if "forced" in case:
        forced = {pname[n]: (None if d is None else (hid[d["host"]], [tname[t] if t in tname else R + pname[t] for t in d["tasks"]],
                                                     (d["cpus"], d["mem"], d.get("gpus", 0.0))))
                  for n, d in case["forced"].items()}
    return dict(params=A.default_params(dru_mode=case.get("dru_mode", 0)), running=tasks, pending=jobs,
                pending_job_id=np.array([17_592_186_045_000 + 1000 + j["job_seq"] for j in pending], dtype=np.int64),
                pending_priority=np.array([j.get("priority", 50) for j in pending], dtype=np.int32),
                users=users, spare=S, rparams=rp, host_attrs=attrs, groups=G, forced=forced,
                slave_known=np.array([0 if t.get("slave_cached") is False else 1 for t in running], dtype=np.uint8),
                init_preempted_hosts=[hid[h] for h in case.get("init_preempted_hosts", [])],
                task_names=[t["name"] for t in running] + [j["name"] for j in pending], host_names=hnames,
                pending_names=[j["name"] for j in pending])


def build_considerable_inputs(case):
    """-> (Queue, UserState, job names, user names) for tests/golden/considerable.json."""
    q = case["queue"]
    unames = sorted({j["user"] for j in q} | set(case["user_usage"]) | set(case["user_quota"]))
    uid = {u: i for i, u in enumerate(unames)}
    queue = A.Queue(cpus=np.array([j["cpus"] for j in q], dtype=np.float64), mem=np.array([j["mem"] for j in q], dtype=np.float64),
                    gpus=np.array([j.get("gpus", 0.0) for j in q], dtype=np.float64),
                    user=np.array([uid[j["user"]] for j in q], dtype=np.uint32),
                    eligible=np.array([1 if j.get("eligible", True) else 0 for j in q], dtype=np.uint8))

    def col(table, key, dflt):
        return np.array([float(table.get(u, {}).get(key, dflt)) for u in unames], dtype=np.float64)

    uq, uu = case["user_quota"], case["user_usage"]
    # a user without a quota entry: (user->quota user) is nil and below-quota? reads every key as 0 -> nothing passes; the
    # pool-filter-only vectors have no user filter at all, which an unbounded quota expresses
    no_user_filter = not uq
    big = A.DMAX
    tokens = case.get("tokens")
    st = A.UserState(
        quota_count=col(uq, "count", big if no_user_filter else 0.0), quota_cpus=col(uq, "cpus", big if no_user_filter else 0.0),
        quota_mem=col(uq, "mem", big if no_user_filter else 0.0), quota_gpus=col(uq, "gpus", big if no_user_filter else 0.0),
        usage_count=col(uu, "count", 0.0), usage_cpus=col(uu, "cpus", 0.0), usage_mem=col(uu, "mem", 0.0), usage_gpus=col(uu, "gpus", 0.0),
        tokens_left=np.array([tokens.get(u, 1 << 40) for u in unames], dtype=np.int64) if tokens is not None else None,
        enforce_rate_limit=bool(case.get("enforce", False)),
        pool_quota=usage_of(case["pool_quota"]) if "pool_quota" in case else None,
        pool_usage=usage_of(case["pool_usage"]) if "pool_usage" in case else None)
    return queue, st, [j["name"] for j in q], unames


# ---- offer construction (tests/golden/offers.json) -----------------------------------------------------------------------
def build_offers_inputs(case):
    """map-level vector -> (Nodes, Pods, CookOfferParams, node names in row order, gpu model names, disk type names).
    The host's part of the boundary: nodes in ascending name order, names interned to dense ids (0 = none), one row per
    pod with the containers' requests summed (merge-with +, api.clj:904-911)."""
    nodes = sorted(case["nodes"], key=lambda n: n["name"])
    names = [n["name"] for n in nodes]
    idx = {n: i for i, n in enumerate(names)}
    pods = case["pods"]
    gm = sorted({n["gpu_type"] for n in nodes if n.get("gpu_type")} | {p["gpu_model"] for p in pods if p.get("gpu_model")})
    dt = sorted({n["disk_type"] for n in nodes if n.get("disk_type")} | {p["disk_type"] for p in pods if p.get("disk_type")})
    gid = {m: i + 1 for i, m in enumerate(gm)}
    did = {t: i + 1 for i, t in enumerate(dt)}
    alloc = lambda n, k, d: (n.get("allocatable") or {}).get(k, d)  # noqa: E731
    flags = [(A.NODE_UNSCHEDULABLE if n.get("unschedulable") else 0) | (A.NODE_OTHER_TAINTS if n.get("other_taints") else 0) |
             (A.NODE_BLOCKLIST_LABEL if n.get("blocklist_label") else 0) | (A.NODE_GPU_TAINT if n.get("gpu_taint") else 0) for n in nodes]
    N = A.Nodes(cpus=[alloc(n, "cpu", 0.0) for n in nodes], mem=[alloc(n, "memory", 0.0) for n in nodes],
                gpus=[alloc(n, "nvidia.com/gpu", 0) for n in nodes], gpu_model=[gid.get(n.get("gpu_type"), 0) for n in nodes],
                disk=[alloc(n, "ephemeral-storage", -1.0) for n in nodes], disk_type=[did.get(n.get("disk_type"), 0) for n in nodes],
                flags=flags)
    rows = dict(node=[], cpus=[], mem=[], gpus=[], gpu_model=[], disk=[], disk_type=[], flags=[])
    for p in pods:
        cs = [c for c in (p.get("containers") or []) if c is not None]
        tot = lambda k: sum(c[k] for c in cs if k in c)  # noqa: E731  (exact for the vectors' values, any order)
        rows["node"].append(idx.get(p.get("node"), A.NONE_U32))
        rows["cpus"].append(float(tot("cpu")))
        rows["mem"].append(float(tot("memory")))
        rows["gpus"].append(int(tot("nvidia.com/gpu")))
        rows["gpu_model"].append(gid.get(p.get("gpu_model"), 0))
        rows["disk"].append(float(tot("ephemeral-storage")) if any("ephemeral-storage" in c for c in cs) else -1.0)
        rows["disk_type"].append(did.get(p.get("disk_type"), 0))
        rows["flags"].append((A.POD_SYNTHETIC if p.get("synthetic") else 0) | (0 if cs else A.POD_NO_REQUESTS))
    P = A.Pods(**{k: np.array(v) for k, v in rows.items()})
    params = A.offer_params(clobber_synthetic_pods=case.get("clobber", False), max_pods_per_node=case.get("max_pods", 2 ** 31 - 1),
                            filter_out_unsound_gpu_nodes=case.get("filter_unsound", False), n_gpu_models=len(gm), n_disk_types=len(dt))
    return N, P, params, names, gm, dt

"""ctypes front-end of oracle/libcookoracle.so — the CPU restatement of the reference algorithm.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by
cook_amd/.  Struct layouts come from cook_amd._abi (= include/cookmatch.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from cook_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libcookoracle.so")
    src = os.path.join(_HERE, "cook_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "cookmatch.h")
    def stale():
        return (not os.path.exists(so)) or any(os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale():
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lk:  # (pytest -n: several workers find the library stale at once; one builds, the others wait)
            fcntl.flock(lk, fcntl.LOCK_EX)
            if force or stale():
                subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_version.restype = C.c_char_p
    return _LIB


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _f64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def rank(params, tasks: A.Tasks, users: A.Users, quota=None, literal_merge=False):
    """-> (ranked pending task indices, dru per task)."""
    out = np.zeros(max(1, tasks.n), dtype=np.uint32)
    dru = np.zeros(max(1, tasks.n), dtype=np.float64)
    n = C.c_uint32(0)
    ts, us = tasks.as_struct(), users.as_struct()
    rc = lib().oracle_rank(C.byref(params), C.byref(ts), C.byref(us), C.byref(quota) if quota is not None else None,
                           _u32p(out), C.byref(n), _f64p(dru), int(literal_merge))
    assert rc == 0
    return out[: n.value].copy(), dru[: tasks.n].copy()


def rank_merged(params, tasks: A.Tasks, users: A.Users, literal_merge=False):
    """-> (task indices, drus) of the full merged sequence (running and pending), dru.clj:114-126."""
    idx = np.zeros(max(1, tasks.n), dtype=np.uint32)
    dru = np.zeros(max(1, tasks.n), dtype=np.float64)
    n = C.c_uint32(0)
    ts, us = tasks.as_struct(), users.as_struct()
    rc = lib().oracle_rank_merged(C.byref(params), C.byref(ts), C.byref(us), _u32p(idx), _f64p(dru), C.byref(n),
                                  int(literal_merge))
    assert rc == 0
    return idx[: n.value].copy(), dru[: n.value].copy()


def user_usage(tasks: A.Tasks, n_users: int) -> np.ndarray:
    """[U, 3] = {cpus, mem, gpus} summed over every user's RUNNING tasks, left to right in the user's task order
    (tools.clj:614-641: -priority, start time, task id).  Oracle-defined order: the reference's per-user usage maps reduce in
    query order (unpinned); the cross-pool sum of this vector is BASELINE.json north_star's all-reduce payload."""
    out = np.zeros((n_users, 3), dtype=np.float64)
    run = np.nonzero(np.asarray(tasks.pending) == 0)[0]
    order = sorted(run.tolist(), key=lambda i: (int(tasks.user[i]), -int(tasks.priority[i]), int(tasks.start_ms[i]), int(tasks.task_id[i]), int(tasks.job_id[i])))
    seen = set()
    g = tasks.gpus if tasks.gpus is not None else None
    for i in order:
        u = int(tasks.user[i])
        c, m, gg = float(tasks.cpus[i]), float(tasks.mem[i]), (float(g[i]) if g is not None else 0.0)
        if u not in seen:
            out[u] = (c, m, gg)
            seen.add(u)
        else:
            out[u, 0] += c
            out[u, 1] += m
            out[u, 2] += gg
    return out


def pool_usage(tasks: A.Tasks) -> A.CookUsage:
    u = A.CookUsage()
    ts = tasks.as_struct()
    lib().oracle_pool_usage(C.byref(ts), C.byref(u))
    return u


def considerable(queue: A.Queue, users: A.UserState, num_considerable: int):
    """-> (queue positions of the considerable jobs, rate_limited per user, passed per user)   (scheduler.clj:729-762)."""
    out = np.zeros(max(1, min(num_considerable, queue.n)), dtype=np.uint32)
    rl = np.zeros(max(1, users.n), dtype=np.uint32)
    ps = np.zeros(max(1, users.n), dtype=np.uint32)
    n = C.c_uint32(0)
    qs, us = queue.as_struct(), users.as_struct()
    rc = lib().oracle_considerable(C.byref(qs), C.byref(us), int(num_considerable), _u32p(out), C.byref(n), _u32p(rl), _u32p(ps))
    assert rc == 0
    return out[: n.value].copy(), rl[: users.n].copy(), ps[: users.n].copy()


def sorted_merge(colls, literal=False):
    """colls: list of sorted key lists -> sequence of coll indices in emission order (dru.clj:82-104)."""
    off = np.zeros(len(colls) + 1, dtype=np.uint32)
    for i, c in enumerate(colls):
        off[i + 1] = off[i] + len(c)
    keys = np.array([k for c in colls for k in c] or [0.0], dtype=np.float64)
    out = np.zeros(max(1, int(off[-1])), dtype=np.uint32)
    lib().oracle_sorted_merge(len(colls), _u32p(off), _f64p(keys), int(literal), _u32p(out))
    return out[: int(off[-1])].copy()


def match(params, jobs: A.Jobs, offers: A.Offers, groups: A.Groups = None, reserved_hosts=(), nthreads=1):
    """-> (job_to_offer int32[K], fail_code uint32[K], head_matched bool)  — scheduleOnce restatement."""
    j2o = np.full(max(1, jobs.n), -1, dtype=np.int32)
    fail = np.zeros(max(1, jobs.n), dtype=np.uint32)
    head = C.c_uint8(0)
    res = np.array(list(reserved_hosts) or [0], dtype=np.uint32)
    js, os_ = jobs.as_struct(), offers.as_struct()
    gs = groups.as_struct() if groups is not None else None
    rc = lib().oracle_match(C.byref(params), C.byref(js), C.byref(os_), C.byref(gs) if gs is not None else None,
                            _u32p(res), len(reserved_hosts), j2o.ctypes.data_as(C.POINTER(C.c_int32)), _u32p(fail),
                            C.byref(head), int(nthreads))
    assert rc == 0
    return j2o[: jobs.n].copy(), fail[: jobs.n].copy(), bool(head.value)


def cycle(params, tasks: A.Tasks, users: A.Users, pending_jobs: A.Jobs, offers: A.Offers, groups: A.Groups = None, quota=None,
          K=None, nthreads=1, deadline_s=0.0):
    """One pool's whole match cycle in ONE library call (oracle_cycle: rank -> first K ranked jobs gathered -> placement), for
    bench.py's cpu_baseline leg: the interpreter lock is released for the whole call, so pool threads run side by side.
    -> (ranked pending task indices, job_to_offer int32[min(K, ranked)], {rank, gather, match} seconds); with deadline_s > 0 the
    placement gives up after that many seconds: -> (None, None, seconds)."""
    n_pend = pending_jobs.n
    K = n_pend if K is None else int(K)
    ranked = np.zeros(max(1, tasks.n), dtype=np.uint32)
    j2o = np.full(max(1, min(K, n_pend)), -1, dtype=np.int32)
    nr, nk = C.c_uint32(0), C.c_uint32(0)
    phase = np.zeros(3, dtype=np.float64)
    ts, us, js, os_ = tasks.as_struct(), users.as_struct(), pending_jobs.as_struct(), offers.as_struct()
    gs = groups.as_struct() if groups is not None else None
    rc = lib().oracle_cycle(C.byref(params), C.byref(ts), C.byref(us), C.byref(quota) if quota is not None else None, C.byref(js),
                            C.byref(os_), C.byref(gs) if gs is not None else None, C.c_uint32(K), int(nthreads), _u32p(ranked),
                            C.byref(nr), j2o.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nk), _f64p(phase), C.c_double(deadline_s))
    assert rc in (0, 1)
    if rc == 1:
        return None, None, {"rank": float(phase[0]), "gather": float(phase[1]), "match": float(phase[2])}
    return ranked[: nr.value].copy(), j2o[: nk.value].copy(), {"rank": float(phase[0]), "gather": float(phase[1]), "match": float(phase[2])}


def match_explain(params, jobs: A.Jobs, offers: A.Offers, groups: A.Groups = None, reserved_hosts=(), job_pos=()):
    """-> (job_to_offer, counts uint32[n, COOK_WHY_SLOTS]): the placement plus, per job position, the placement-failure summary of
    fenzo_utils.clj:33-55 in the COOK_WHY_* slots of include/cookmatch.h."""
    j2o = np.full(max(1, jobs.n), -1, dtype=np.int32)
    res = np.array(list(reserved_hosts) or [0], dtype=np.uint32)
    pos = np.ascontiguousarray(job_pos, dtype=np.uint32)
    counts = np.zeros((max(1, len(pos)), 20), dtype=np.uint32)
    js, os_ = jobs.as_struct(), offers.as_struct()
    gs = groups.as_struct() if groups is not None else None
    rc = lib().oracle_match_explain(C.byref(params), C.byref(js), C.byref(os_), C.byref(gs) if gs is not None else None,
                                    _u32p(res), len(reserved_hosts), j2o.ctypes.data_as(C.POINTER(C.c_int32)),
                                    _u32p(pos if len(pos) else np.zeros(1, np.uint32)), len(pos), _u32p(counts.reshape(-1)))
    assert rc == 0
    return j2o[: jobs.n].copy(), counts[: len(pos)].copy()


def resource_stats(cpus, mem) -> dict:
    """resource-maps->stats (scheduler.clj:547-582) of the "cpus" and "mem" columns: :totals = left-to-right sums,
    :percentiles by nearest rank (task_stats.clj:59-80: index ceil(p/100 * n) - 1 in exact ratio arithmetic),
    :largest-by = the last element of the stable sort by that resource."""
    from fractions import Fraction
    import math
    out = {}
    n = len(cpus)
    for name, col in (("cpus", np.asarray(cpus, np.float64)), ("mem", np.asarray(mem, np.float64))):
        if n == 0:
            out.update({f"total_{name}": 0.0, f"p50_{name}": float("nan"), f"p95_{name}": float("nan"), f"p100_{name}": float("nan"),
                        f"largest_by_{name}": A.NONE_U32})
            continue
        order = np.argsort(col, kind="stable")
        srt = col[order]
        out[f"total_{name}"] = float(np.add.accumulate(col)[-1])  # sequential, like (reduce (partial merge-with +))
        for p in (50, 95, 100):
            out[f"p{p}_{name}"] = float(srt[math.ceil(Fraction(p, 100) * n) - 1])
        out[f"largest_by_{name}"] = int(order[-1])
    return out


class RebalHooks(C.Structure):
    _fields_ = [("running_slave_known", C.POINTER(C.c_uint8)), ("init_preempted_hosts", C.POINTER(C.c_uint32)),
                ("n_init_preempted", C.c_uint32), ("forced_host", C.POINTER(C.c_int32)), ("forced_off", C.POINTER(C.c_uint32)),
                ("forced_task", C.POINTER(C.c_uint32)), ("forced_res", C.POINTER(A.CookUsage)),
                ("final_order", C.POINTER(C.c_uint32)), ("final_dru", C.POINTER(C.c_double)), ("n_final", C.POINTER(C.c_uint32))]


def rebalance(params, running: A.Tasks, pending: A.Jobs, pending_job_id, pending_priority, users: A.Users,
              spare: A.HostSpare, rparams: A.CookRebalanceParams, host_attrs: A.Offers = None, groups: A.Groups = None,
              forced=None, want_final=False, slave_known=None, init_preempted_hosts=()):
    """-> dict(decisions=[...], pending_dru=array, final=(order, dru))   (rebalancer.clj:434-467).
    Test hooks: forced = {pending index: None | (host, [task ids], (cpus, mem, gpus))} applies that decision instead of
    computing one; slave_known / init_preempted_hosts: see oracle_rebal_hooks in cook_oracle.cpp."""
    P, R = pending.n, running.n
    dec = (A.CookPreemption * max(1, P))()
    pre = np.zeros(max(1, R + P), dtype=np.uint32)
    nd, npre = C.c_uint32(0), C.c_uint32(0)
    jid = np.ascontiguousarray(pending_job_id, dtype=np.int64)
    pri = np.ascontiguousarray(pending_priority, dtype=np.int32)
    pdru = np.zeros(max(1, P), dtype=np.float64)
    rs, ps, us, ss = running.as_struct(), pending.as_struct(), users.as_struct(), spare.as_struct()
    hs = host_attrs.as_struct() if host_attrs is not None else None
    gs = groups.as_struct() if groups is not None else None
    hooks = RebalHooks()
    keep = []
    i32p, u32p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    if slave_known is not None:
        sk = np.ascontiguousarray(slave_known, dtype=np.uint8)
        keep.append(sk)
        hooks.running_slave_known = sk.ctypes.data_as(u8p)
    if len(init_preempted_hosts):
        ip = np.ascontiguousarray(init_preempted_hosts, dtype=np.uint32)
        keep.append(ip)
        hooks.init_preempted_hosts = ip.ctypes.data_as(u32p)
        hooks.n_init_preempted = len(ip)
    if forced is not None:
        f_host = np.full(max(1, P), -2, dtype=np.int32)
        f_off = np.zeros(P + 1, dtype=np.uint32)
        flat = []
        f_res = (A.CookUsage * max(1, P))()
        for pj in range(P):
            if pj in forced:
                d = forced[pj]
                if d is None:
                    f_host[pj] = -1
                else:
                    f_host[pj] = d[0]
                    flat.extend(d[1])
                    f_res[pj] = A.usage(0, *d[2])
            f_off[pj + 1] = len(flat)
        f_task = np.array(flat or [0], dtype=np.uint32)
        keep += [f_host, f_off, f_task, f_res]
        hooks.forced_host = f_host.ctypes.data_as(i32p)
        hooks.forced_off = f_off.ctypes.data_as(u32p)
        hooks.forced_task = f_task.ctypes.data_as(u32p)
        hooks.forced_res = f_res
    fin_o = np.zeros(max(1, R + P), dtype=np.uint32)
    fin_d = np.zeros(max(1, R + P), dtype=np.float64)
    nf = C.c_uint32(0)
    if want_final:
        hooks.final_order = fin_o.ctypes.data_as(u32p)
        hooks.final_dru = fin_d.ctypes.data_as(C.POINTER(C.c_double))
        hooks.n_final = C.pointer(nf)
    rc = lib().oracle_rebalance(C.byref(params), C.byref(rs), C.byref(ps), jid.ctypes.data_as(C.POINTER(C.c_int64)),
                                pri.ctypes.data_as(i32p), C.byref(us), C.byref(ss),
                                C.byref(hs) if hs is not None else None, C.byref(gs) if gs is not None else None,
                                C.byref(rparams), dec, C.byref(nd), _u32p(pre), C.byref(npre), _f64p(pdru), C.byref(hooks))
    assert rc == 0
    out = []
    for i in range(nd.value):
        d = dec[i]
        out.append(dict(pending_index=d.pending_index, host=d.host, dru=d.dru, cpus=d.cpus, mem=d.mem, gpus=d.gpus,
                        tasks=[int(x) for x in pre[d.task_off: d.task_off + d.task_n]]))
    return dict(decisions=out, pending_dru=pdru[:P].copy(),
                final=(fin_o[: nf.value].copy(), fin_d[: nf.value].copy()) if want_final else None)

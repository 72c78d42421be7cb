#!/usr/bin/env python3
"""bench.py — match-cycles/sec + p50 cycle latency at 1M pending x 50k offers (BASELINE.json metric).

One "step" = one match cycle of the whole synthetic cluster: for every pool, cook_rank over all running+pending tasks
of the pool followed by cook_match of the first K ranked jobs against all of the pool's offers.
Workload = BASELINE.json configs[3]: 8 pools x (125k pending + 50k running tasks, 6 250 offers), 10k users, gpu
dimension + attribute/novel-host/unique-group constraints, quota group over all pools.  K defaults to "all pending"
(the literal 1M x 50k jobs x offers problem; the reference caps K at 1000 only because its CPU path cannot afford
more, config.clj:113) — `--considerable 1000` reproduces the reference default.

Scaling is STRONG: the cluster (8 pools) is fixed; N ranks take pools r, r+N, ... (one pool per GPU at N=8, exactly
configs[3]).  The only cross-rank exchange is the all-reduce of per-pool running usage into the quota-group usage
(scheduler.clj:2125-2157) over torch.distributed (RCCL).  Inputs are resident in HBM before the timed region.

Launch: `python bench.py` (N=1) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pools", type=int, default=8)
    ap.add_argument("--pending", type=int, default=1_000_000, help="pending jobs over all pools")
    ap.add_argument("--running", type=int, default=400_000)
    ap.add_argument("--offers", type=int, default=50_000)
    ap.add_argument("--users", type=int, default=10_000)
    ap.add_argument("--considerable", type=int, default=0, help="K per pool; 0 = all ranked pending jobs")
    ap.add_argument("--good-enough", type=float, default=1.0, help="1.0 = parity setting (zz_simulator.clj:84)")
    ap.add_argument("--no-constraints", action="store_true")
    ap.add_argument("--match-algo", type=int, default=0, help="cook_params.match_algo: 0 / 2 window rounds (eval, merge, resolve launches), 1 = the one-job-at-a-time sweep by a single workgroup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-adjacent", action="store_true", help="skip the timings of the rows next to the hot path (offers, explain, metrics)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="cap on the oracle threads of the cpu_baseline leg; 0 = os.cpu_count()")
    ap.add_argument("--no-extras", action="store_true", help="skip SURVEY.md §8d's reporting matrix (K = 1000 / 1e5, good-enough 0.8, C2, C3) under extra_configs")
    ap.add_argument("--boundary", action="store_true", help="run the boundary leg (cook_cycle_update from page-locked columns) even with --no-extras")
    ap.add_argument("--as-rank-of", type=int, default=0, help="single process: time only the pools rank 0 of an N-GPU job would hold (the per-GPU load behind DESIGN.md's scaling prediction; not a bench line)")
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="strong: the cluster of --pools pools over N GPUs (the default: BASELINE's metric); weak: --pools pools PER GPU — a cluster of N x --pools "
                         "pools with N x the jobs, tasks and offers (what uses a node: one pool is one chain, DESIGN.md 8)")
    ap.add_argument("--no-check", action="store_true", help="skip the parity check of the timed configuration (rank 0's first and last pool vs the oracle, after the timed region)")
    # rehearsal of the multi-process path on a machine without GPUs (tests/test_sharding_gloo.py): the engines load the given build of
    # the library (the SIMT emulator, test infrastructure) and the collectives run over gloo.  Never a bench line: `data` says so.
    ap.add_argument("--engine-lib", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"), help=argparse.SUPPRESS)
    return ap.parse_args()


# match_walkers / match_serve_*: the served form of the same three phases (one persistent walker workgroup per pool, ONE launch per cycle,
# beside evaluation + merge launches for the pools that asked: cook_amd/csrc/match_v2.hpp "served walkers")
PLACEMENT_KERNELS = ("match_resolve2", "match_eval2", "match_merge2", "match_serial", "match_walkers", "match_serve_eval", "match_serve_merge", "cf_walk")


def rank_batch_report(stats):
    """what the rank's pool batches of the last cycle did, summed over the batches' lead engines (sharding.py: up to four batches)"""
    leads = [st for st in stats if st.get("rank_batch_pools", 0) > 0]
    out = {"batches": len(leads)}
    for k in ("pools", "launches", "grouped_launches", "single_ops", "syncs"):
        out[k] = sum(st["rank_batch_" + k] for st in leads)
    return out


def algorithmic_bytes(kernel, n_tasks, k, m, launches_per_match=1.0, pools_per_launch=1.0):
    """Inputs read once + outputs written once per launch, per pool.

    Placement: the per-job figure of B_feas + B_assign (40 B job vector in, 4 B assignment out, the job's M/8-byte row of the
    feasibility bit matrix) x the jobs one launch resolves (K / launches of that kernel per match call, empty over-launched
    rounds included) + the 64 B offer records, which every launch reads once (per pool it serves: a rank with more than four
    pools runs them in lockstep groups, blockIdx.z = pool)."""
    if kernel in PLACEMENT_KERNELS:
        jobs_per_launch = k / max(1e-9, launches_per_match)  # over all the pools a launch serves
        return jobs_per_launch * (40 + 4 + m / 8.0) + 64 * m * pools_per_launch
    table = {
        "radix_scatter": 16 * n_tasks,                  # 8 B key gather + 4 B perm in + 4 B perm out
        "radix_hist": 12 * n_tasks,
        "user_usage_scan": 73 * n_tasks,                # SumU4 in + out + head flag
        "rank_gather": 57 * n_tasks,
    }
    return table.get(kernel)


def cpu_stat():
    """nr_throttled / throttled_usec of the container's CPU controller (a process over its quota is stopped until the period ends)"""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            kv = dict(l.split() for l in f.read().strip().splitlines())
        return {k: int(kv[k]) for k in ("nr_periods", "nr_throttled", "throttled_usec") if k in kv}
    except (OSError, ValueError):
        return None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc passes (profiles/r*_pmc_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate passes, scripts/profile_round2.sh + scripts/make_pmc_traffic.py).  The file records the revision
    of the kernel sources it was taken at (scripts/kernel_rev.py); counters of ANOTHER binary are not quoted: (None, reason)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, "no profiles/r*_pmc_traffic.json"
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from kernel_rev import kernel_rev
        rev = kernel_rev()
        docs = []
        for fn in files:
            with open(fn) as f:
                docs.append((fn, json.load(f)))
    except (OSError, ValueError, ImportError) as ex:
        return None, f"unreadable: {ex}"
    match = [(fn, d) for fn, d in docs if d.get("kernel_rev") == rev]  # (the passes of THIS revision, whatever the file is called)
    if not match:
        fn, d = max(docs, key=lambda x: os.path.getmtime(x[0]))
        return None, f"{os.path.relpath(fn, ROOT)} was taken at kernel revision {d.get('kernel_rev')}, this binary is {rev} (no file of {len(docs)} matches)"
    fn, doc = match[-1]
    name = os.path.relpath(fn, ROOT)
    rec = doc.get("kernels", {}).get(kernel)
    if rec is None:
        return None, f"{name} holds no pass for {kernel}"
    return rec, name


def extra_c5(device, check=True, steps=3, sizes=(1_000_000, 500_000, 10_000, 50_000, 128)):
    """BASELINE.json configs[4] as the reference runs it (rebalancer.clj:574-590): the rebalancer takes its jobs from the RANKED
    queue of the pool — cook_rank over 1M running + 500k pending tasks — keeps the first max-preemption (128) of them, then the
    preemption sweep (init-state + compute-preemption-decision + next-state per job, rebalancer.clj:222-467) over the 1M running
    tasks on 50k hosts.  Both legs timed with resident inputs, both compared with the oracle."""
    import torch
    from cook_amd import _abi as A
    from cook_amd import synth
    from cook_amd.engine import Engine
    R, PEND, U, H, MAXP = sizes
    pool = synth.make_pool(seed=0xC00C0005, n_pending=PEND, n_running=R, n_users=U, n_offers=H)
    params = A.default_params()
    out = {"what": f"BASELINE.json configs[4]: rank of {R} running + {PEND} pending tasks of one pool ({U} users), then the preemption sweep "
                   f"for the first {MAXP} ranked pending jobs over the running tasks on {H} hosts (no spare capacity: every decision preempts)"}
    with Engine(params, device=device) as e:
        e.rank_stage(pool.tasks, pool.users)
        ts = []
        for _ in range(steps + 1):
            e.rank_run()
            torch.cuda.synchronize()
            ts.append(e.last_timing()[0])
        ranked, dru = e.rank_fetch(want_dru=True)
        rank_ms = sorted(ts[1:])[len(ts[1:]) // 2]
        n_tasks = pool.tasks.n
        out["rank"] = {"ms": rank_ms, "tasks": n_tasks, "ranked_pending": int(len(ranked)), "algorithmic_bytes": 52 * n_tasks,
                       "GBps_algorithmic": 52 * n_tasks / (rank_ms * 1e-3) / 1e9, "frac_of_hbm_peak": 52 * n_tasks / (rank_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # the sweep: the pool's running tasks, the first MAXP ranked pending jobs
        run_idx = np.nonzero(pool.tasks.pending == 0)[0]
        t = pool.tasks
        running = A.Tasks(cpus=t.cpus[run_idx], mem=t.mem[run_idx], gpus=None, user=t.user[run_idx], priority=t.priority[run_idx],
                          start_ms=t.start_ms[run_idx], task_id=t.task_id[run_idx], job_id=t.job_id[run_idx],
                          pending=np.zeros(len(run_idx), dtype=np.uint8), host=t.host[run_idx])
        head = ranked[:MAXP]
        pend_ord = np.cumsum(t.pending) - 1
        pending = pool.pending_jobs.take(pend_ord[head])
        spare = A.HostSpare(host=np.zeros(0, dtype=np.uint32), cpus=np.zeros(0), mem=np.zeros(0), gpus=np.zeros(0))
        rp = A.CookRebalanceParams(0.0, 0.05, MAXP, 0)
        e.rebalance_stage(running, pending, t.job_id[head], t.priority[head], pool.users, spare, rp)
        e.rebalance_run()
        torch.cuda.synchronize()
        ev = []
        for _ in range(steps):
            e.rebalance_run()
            torch.cuda.synchronize()
            ev.append(e.rebalance_timing())
        got = e.rebalance_fetch()
        sweep_ms = sorted(ev)[len(ev) // 2]
        nbytes = 52 * R + MAXP * 40 * R
        out["sweep"] = {"ms": sweep_ms, "decisions": len(got["decisions"]), "preempted": sum(len(d["tasks"]) for d in got["decisions"]),
                        "algorithmic_bytes": nbytes, "GBps_algorithmic": nbytes / (sweep_ms * 1e-3) / 1e9,
                        "frac_of_hbm_peak": nbytes / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        out["ms_total"] = rank_ms + sweep_ms
    out["parity_checked"] = False
    if check:
        from oracle import pyoracle
        c0 = time.perf_counter()
        o_ranked, o_dru = pyoracle.rank(params, pool.tasks, pool.users)
        c1 = time.perf_counter()
        assert np.array_equal(ranked, o_ranked) and np.array_equal(dru, o_dru, equal_nan=True), "PARITY: C5 rank differs from the oracle"
        c2 = time.perf_counter()
        want = pyoracle.rebalance(params, running, pending, t.job_id[head], t.priority[head], pool.users, spare, rp)
        c3 = time.perf_counter()
        out["cpu_baseline"] = {"rank_s": c1 - c0, "sweep_s": c3 - c2, "cycle_s": (c1 - c0) + (c3 - c2), "cores": 1, "kind": "port",
                               "sample": "the oracle's rank and sweep on the same inputs, whole calls, one thread"}
        out["speedup_vs_cpu_baseline"] = {"rank": (c1 - c0) * 1e3 / out["rank"]["ms"] if isinstance(out.get("rank"), dict) and out["rank"].get("ms") else None,
                                          "sweep": (c3 - c2) * 1e3 / out["sweep"]["ms"], "total": ((c1 - c0) + (c3 - c2)) * 1e3 / out["ms_total"]}
        assert len(want["decisions"]) == len(got["decisions"]), "PARITY: C5 decisions differ from the oracle"
        for a, b in zip(got["decisions"], want["decisions"]):
            assert a == b, f"PARITY: C5 decision differs from the oracle: {a} / {b}"  # host, dru, resources (fp64 ==), preempted tasks
        assert np.array_equal(got["pending_dru"], want["pending_dru"], equal_nan=True), "PARITY: C5 pending DRUs differ from the oracle"
        out["parity_checked"] = True
    return out


def host_core_budget():
    """-> (cores this process may really use at once, logical CPUs, note): the affinity mask, cut to a container's CPU quota (the GPU boxes of this
    pool: 256 logical CPUs, quota 16)."""
    host_cores = os.cpu_count() or 1
    try:
        host_cores = min(host_cores, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    logical_cpus, quota_note = host_cores, None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
            if quota < host_cores:
                host_cores, quota_note = quota, f"cgroup cpu.max {q} {per} = {quota} CPUs of {logical_cpus} logical"
    except (OSError, ValueError):
        pass
    return host_cores, logical_cpus, quota_note


def cpu_leg_pools(params_x, cluster, pools, my_pools, K_x, nt):
    """The oracle's whole cycle (rank -> gather -> placement; one library call per pool, the pools side by side on threads of their own) for an
    extra operating point of the line.  -> {cycle_s, rank_s, match_s (slowest pool's), cores, sample}"""
    import threading
    from oracle import pyoracle
    quota = {p: cluster.quota_inputs(p, cluster.last_pool_usage[p], cluster.last_group_usage) for p in my_pools}
    out = {}

    def one(p):
        out[p] = pyoracle.cycle(params_x, pools[p].tasks, pools[p].users, pools[p].pending_jobs, pools[p].offers, pools[p].groups, quota=quota[p], K=K_x, nthreads=nt)

    ths = [threading.Thread(target=one, args=(p,)) for p in my_pools]
    a = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    wall = time.perf_counter() - a
    assert len(out) == len(my_pools), "an oracle thread failed"
    return {"cycle_s": wall, "rank_s": max(o[2]["rank"] for o in out.values()), "match_s": max(o[2]["match"] for o in out.values()),
            "cores": len(my_pools) * nt, "kind": "port",
            "sample": f"the whole cycle (not a sample) of the {len(my_pools)} pools at once, {nt} evaluator thread(s) per pool, one library call per pool thread"}


def cpu_baseline_leg(args, params, cluster, pools, my_pools, K, n_off):
    """The oracle ("port": C++ restatement of the reference algorithm, TEST INFRASTRUCTURE used here only as the timed CPU leg and as
    the checker) on this box's host cores, over the whole cycle of the pools held by this process.  -> (cpu_baseline dict,
    {pool: (ranked, j2o)} of the oracle for the parity check).

    The reference runs one match handler per pool concurrently (tools.clj:799-806, scheduler.clj:2425-2435) and Fenzo spreads a
    job's host evaluation over an executor per CPU (built at scheduler.clj:2301-2324), so the forms timed are `a` pools at once x
    `t` evaluator threads per pool for every (a, t) with a x t <= the host's cores; `value` = the fastest.  Inside the timed region
    a pool thread makes ONE library call (oracle_cycle: rank -> gather -> placement) — the interpreter lock is released for all of
    it; the call's own clocks (rank / gather / match seconds per pool) are reported next to the wall time."""
    import threading
    from oracle import pyoracle
    host_cores, logical_cpus, quota_note = host_core_budget()
    P = len(my_pools)
    quota = {p: cluster.quota_inputs(p, cluster.last_pool_usage[p], cluster.last_group_usage) for p in my_pools}
    threaded_ok = args.good_enough >= 1.0  # (the oracle's host-bucketed form is best fit only)
    cap = max(1, min(args.cpu_threads or host_cores, host_cores))
    for p in my_pools:  # ctypes views of the columns built once, outside every timed region
        pools[p].tasks.as_struct(), pools[p].pending_jobs.as_struct(), pools[p].offers.as_struct()

    def one_pool(p, nt, out, deadline):
        out[p] = pyoracle.cycle(params, pools[p].tasks, pools[p].users, pools[p].pending_jobs, pools[p].offers, pools[p].groups,
                                quota=quota[p], K=K, nthreads=nt, deadline_s=deadline)

    def batch(which, nt, deadline):
        out = {}
        ths = [threading.Thread(target=one_pool, args=(p, nt, out, deadline)) for p in which]
        a = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        wall = time.perf_counter() - a
        assert len(out) == len(which), "an oracle thread failed"
        return wall, out

    def run_form(at_once, nt, deadline):
        """All of this process's pools, `at_once` of them side by side, the batches one after the other.  -> (wall, outputs, wall over the
        slowest library call) or None when a pool's placement ran into the deadline (a form that oversubscribes the cores the process may
        really use: its evaluator threads meet once per job)."""
        wall, out, slack = 0.0, {}, 0.0
        for b in range(0, P, at_once):
            w, o = batch(my_pools[b:b + at_once], nt, deadline)
            if any(o[p][0] is None for p in o):
                return None
            wall += w
            out.update(o)
            slack = max(slack, w / max(1e-9, max(sum(o[p][2].values()) for p in o)))
        return wall, out, slack

    # the forms: every pool at once x t evaluator threads per pool (t = 1, 2, 4, ... while P x t fits the cores, at most 32: the reference's
    # shape on a many-core host), then fewer pools at once with the threads that frees (P/2 x 2t_max, ..., 1 pool x up to 64 threads)
    forms = []
    a0 = min(P, cap)
    t = 1
    while a0 * t <= cap and t <= 32:
        if threaded_ok or t == 1:
            forms.append((a0, t))
        t *= 2
    a, tmax = a0 // 2, (forms[-1][1] if forms else 1) * 2
    while a >= 1 and threaded_ok:
        if a * tmax <= cap and tmax <= 64:
            forms.append((a, tmax))
            tmax *= 2
        a //= 2
    variants, skipped, ref, budget_s = [], [], None, 75.0
    t_begin = time.perf_counter()
    base_s = None
    for at_once, nt in forms:
        if variants and time.perf_counter() - t_begin > budget_s:
            skipped.append({"pools_at_once": at_once, "threads_per_pool": nt, "why": "the leg's time budget was used up"})
            continue
        deadline = 0.0 if base_s is None else max(8.0, 2.5 * base_s)  # (per library call; the first form — 1 thread per pool — has none)
        print(f"bench.py: cpu_baseline form {at_once} pool(s) at once x {nt} thread(s)", file=sys.stderr, flush=True)
        res = run_form(at_once, nt, deadline)
        if res is not None and res[2] > 1.1:  # the wall time must be the slowest pool's library call, not the harness: once more before it is reported as is
            res = run_form(at_once, nt, deadline) or res
        if res is None:
            skipped.append({"pools_at_once": at_once, "threads_per_pool": nt, "why": f"a pool's placement had not finished after {deadline:.0f} s (oversubscribed cores)"})
            continue
        wall, out, slack = res
        if base_s is None:
            base_s = wall
        if ref is None:
            ref = out
        else:
            for p in my_pools:
                if not (np.array_equal(out[p][0], ref[p][0]) and np.array_equal(out[p][1], ref[p][1])):
                    raise AssertionError("oracle: threaded and single-thread placements differ")
        variants.append({"form": "pools concurrent" if at_once >= P else f"{at_once} pool(s) at a time", "pools_at_once": at_once,
                         "threads_per_pool": nt, "cores": min(at_once * nt, host_cores), "cycle_s": wall, "cycles_per_s": 1.0 / wall,
                         "wall_over_slowest_library_call": round(slack, 3),
                         "slowest_pool_s": {k: round(max(out[p][2][k] for p in my_pools), 4) for k in ("rank", "gather", "match")}})
    best = max(variants, key=lambda v: v["cycles_per_s"])
    p0 = ref[my_pools[0]][2]
    cpu = {"value": best["cycles_per_s"], "unit": "cycles/s", "cores": best["cores"], "kind": "port",
           "sample": f"the whole cycle (not a sample): oracle rank + placement of all K = {K} considerable jobs x {n_off} offers of each of the {P} "
                     f"pools; fastest form: {best['pools_at_once']} pool(s) at once x {best['threads_per_pool']} thread(s) per pool = "
                     f"{best['cores']} cores, {best['cycle_s']:.3f} s per cycle",
           "variants": variants, "forms_not_reported": skipped, "host_cores": host_cores, "host_logical_cpus": logical_cpus, "host_cpu_quota": quota_note, "harness_overhead_ok": all(v["wall_over_slowest_library_call"] <= 1.1 for v in variants),
           "rank_s_pool0": p0["rank"], "gather_s_pool0": p0["gather"], "match_s_pool0_single_thread": p0["match"],
           "note": "C++ restatement (-O2) of the reference algorithm; the JVM reference cannot run here (no JDK, Fenzo jar absent). One library "
                   "call per pool thread inside the timed region (rank -> gather -> placement, interpreter lock released); the threads-per-pool "
                   "forms bucket a job's hosts over persistent workers that meet once per job (Fenzo's executor-per-CPU evaluation)."}
    return cpu, {p: (ref[p][0], ref[p][1]) for p in my_pools}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` from a plain shell: launch ourselves as one process per GPU (the driver's form is
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`, which arrives here with WORLD_SIZE set)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    os.environ.setdefault("COOK_POOL_USAGE_MEMO", "0")  # the timed cycles sum the pools' running usage every time, as a live cycle (a new task table) does
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # one hardware queue per pool stream (read at HIP initialisation; see cook_amd/engine.py)
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # (the default of this ROCm build; see cook_amd/engine.py)
    import torch
    import torch.distributed as dist

    rehearsal = args.engine_lib is not None  # the emulator build on CPU: plumbing only, no number
    if not rehearsal:
        assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    on_gpu = torch.cuda.is_available() and not rehearsal
    # (COOK_BENCH_ONE_DEVICE=1, a TEST aid: every rank on cuda:0 — with --dist-backend gloo the whole multi-process path, real HIP engines
    #  and real collectives, runs on a box with one GPU: tests/test_parity_gpu.py.  Not a measurement.)
    dev_index = 0 if os.environ.get("COOK_BENCH_ONE_DEVICE") == "1" else local_rank
    if on_gpu:
        torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index) if on_gpu else torch.device("cpu")

    def device_sync():
        if on_gpu:
            torch.cuda.synchronize()

    cdev = dev if (on_gpu and args.dist_backend == "nccl") else torch.device("cpu")  # where the collectives' tensors live (gloo: on the host)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu and args.dist_backend == "nccl":
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    from cook_amd import _abi as A
    from cook_amd import synth
    from cook_amd.engine import Engine

    weak = args.scaling == "weak"
    grow = world if weak else 1  # weak scaling: every GPU brings --pools pools of the same size with it
    P = args.pools * grow
    from cook_amd import workload
    from cook_amd.sharding import pools_of_rank
    my_pools = pools_of_rank(P, world, rank) if not args.as_rank_of else pools_of_rank(P, args.as_rank_of, 0)
    spec = workload.ClusterSpec(pools=P, pending=args.pending * grow, running=args.running * grow, offers=args.offers * grow, users=args.users,
                                constraints=not args.no_constraints)
    n_pend, n_run, n_off = spec.per_pool
    params = A.default_params(good_enough_fitness=args.good_enough, match_algo=args.match_algo)
    K = args.considerable if args.considerable > 0 else n_pend

    # ---- synthetic inputs, staged into HBM before the timed region ------------------------------------------
    t0 = time.time()
    pools, engines = {}, {}
    for p in my_pools:
        pools[p] = workload.make_pool(spec, p)
        e = Engine(params, device=0, lib_path=args.engine_lib) if rehearsal else Engine(params, device=dev_index)  # (the emulator has one device)
        e.cycle_stage(pools[p].tasks, pools[p].users, pools[p].pending_jobs, pools[p].offers, pools[p].groups)
        engines[p] = e
    gen_s = time.time() - t0
    # quota: every pool has a (non-binding) pool quota and belongs to ONE quota group "s" whose usage is the sum over
    # all pools of the cluster -> the cross-rank all-reduce (scheduler.clj:2125-2157); cook_amd/sharding.py
    from cook_amd import sharding
    qg = workload.quota_groups(spec)
    cluster = sharding.ShardedCluster(engines, qg, world=world, rank=rank, device=cdev, serial=rehearsal)  # (the emulator runs one launch at a time)
    cluster.n_users = args.users  # every timed cycle runs north_star's collective: the all-reduce of the cross-pool per-user usage [U x 3]

    def cycle():
        cluster.cycle(K)

    def fence():
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()

    # (the interpreter's cyclic collector stays out of the timed regions, as in timeit: a generation-2 pass over this process's object
    #  graph is a 10+ ms pause of whichever thread holds the lock; the host of a deployment is a JVM, not this harness)
    import gc
    def mark(what):
        if rank == 0:
            print(f"bench.py [{time.time() - t0:7.1f} s] {what}", file=sys.stderr, flush=True)

    mark("inputs staged; warm-up + timed region")
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        cycle()
    fence()
    lat = []
    phases = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        cycle()
        lat.append(time.perf_counter() - t1)
        phases.append(cluster.last_phase_ms)
    fence()
    elapsed = time.perf_counter() - t_start
    gc.enable()
    tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())

    # results of the last cycle (for the record + sanity)
    matched = considered = ranked_n = 0
    stage_ms = {}
    fetched = {}
    for p in my_pools:
        r, j2o, head = engines[p].cycle_fetch()
        fetched[p] = (r, j2o)
        ranked_n += len(r)
        considered += len(j2o)
        matched += int((j2o >= 0).sum())
        stage_ms[p] = engines[p].last_timing()
    timed_stats = engines[my_pools[0]].match_stats()  # of the LAST TIMED cycle (the extras below run other forms)
    timed_form = timed_stats.get("placement_form", 0)
    cnt = torch.tensor([matched, considered, ranked_n], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    matched, considered, ranked_n = [int(x) for x in cnt.tolist()]

    # ---- what the collectives saw (the line proves that N ranks took part): every rank reports its pools and their running usage, rank 0
    #      re-adds them on the host and compares with what the timed cycle's all-reduce left on it (scheduler.clj:2125-2157: the group
    #      usage is the sum over ALL pools of the group, whichever rank holds them; :2488-2506: one handler per pool) -----------------
    mine = {"rank": rank, "pools": list(my_pools), "device": str(dev),
            "pool_usage": {int(p): [float(x) for x in cluster.last_pool_usage[p]] for p in my_pools}}
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
    else:
        everyone = [mine]
    collective = None
    if rank == 0 and not args.as_rank_of:  # (--as-rank-of times one rank's share of the pools in a single process: no collective to report)
        all_usage = {int(p): u for r in everyone for p, u in r["pool_usage"].items()}
        assert sorted(all_usage) == list(range(P)), f"the ranks hold pools {sorted(all_usage)}, the cluster has {P}"
        want = sharding.group_usage_matrix(qg, all_usage)
        assert np.array_equal(want, np.asarray(cluster.last_group_usage)), "the all-reduced quota-group usage is not the sum over all pools"
        collective = {"backend": (dist.get_backend() if world > 1 else None), "world_size": (dist.get_world_size() if world > 1 else 1),
                      "pools_of_rank": [r["pools"] for r in sorted(everyone, key=lambda r: r["rank"])],
                      "devices": [r["device"] for r in sorted(everyone, key=lambda r: r["rank"])],
                      "group_usage_allreduced": np.asarray(cluster.last_group_usage).tolist(), "group_usage_equals_sum_over_all_pools": True,
                      "per_cycle": ["all_reduce(SUM) f64 [n_groups x 4] quota-group usage", f"all_reduce(SUM) f64 [{args.users} x 3] per-user usage"]}

    mark("timed region done; roofline pass")
    # ---- roofline of the dominant kernel: second pass with per-kernel HIP events on each engine's own stream ----
    roofline = None
    if not args.no_roofline:  # every rank runs the pass (cycle() holds a collective); rank 0 reports
        for p in my_pools:
            engines[p].set_profiling(True)
        for _ in range(max(1, min(args.steps, 3))):
            cycle()
        agg = {}
        for p in my_pools:
            for name, (ms, n) in engines[p].kernel_timings().items():
                a = agg.setdefault(name, [0.0, 0])
                a[0] += ms
                a[1] += n
            engines[p].set_profiling(False)
        if agg:
            dom = max(agg, key=lambda k: agg[k][0])
            if "match_walkers" in agg:
                # served walkers: the kernel that places the jobs is the ONE persistent walker launch of the cycle.  (The serve launches'
                # event durations include the bounded wait of their latch for the next request — idle polling by one wave, up to
                # COOK_SERVE_POLL_US per empty iteration —, so their sums say nothing about work.)
                dom = "match_walkers"
            avg_ms = agg[dom][0] / max(1, agg[dom][1])
            n_cycles = max(1, min(args.steps, 3))
            launches_per_match = agg[dom][1] / (n_cycles * max(1, len(my_pools)))
            n_chains = min(len(my_pools), cluster.max_chains) if len(my_pools) > cluster.max_chains else len(my_pools)
            pools_per_launch = len(my_pools) / max(1, n_chains)
            served_mode = engines[my_pools[0]].match_stats().get("served_mode", 0)
            if served_mode and dom == "match_walkers":
                pools_per_launch = float(len(my_pools))  # ONE walker launch per cycle places every pool of the rank
            if dom == "cf_walk":
                pools_per_launch = 1.0 / max(1e-9, launches_per_match)  # class-ordered best fit: ONE ordinary launch, a workgroup per pool (blockIdx.x = pool)
            nbytes = algorithmic_bytes(dom, n_pend + n_run, min(K, n_pend), n_off, launches_per_match, pools_per_launch)
            achieved = (nbytes / (avg_ms * 1e-3) / 1e9) if (nbytes and avg_ms > 0) else None
            total_ms = sum(v[0] for v in agg.values())
            tr, tr_src = pmc_traffic(dom)
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                        "traffic": tr["hbm_bytes_per_launch"] if tr else None, "traffic_source": tr_src,
                        "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": nbytes, "launches_per_match": launches_per_match, "pools_per_launch": pools_per_launch,
                        "share_of_kernel_time": agg[dom][0] / total_ms if total_ms else None,
                        "note": "placement is a sequential dependency chain (job i+1 sees job i's commitment): "
                                "latency-bound, not bandwidth-bound; see DESIGN.md"
                                + ("; match_walkers is ONE persistent launch per cycle (a walker workgroup per pool) whose duration includes the time its "
                                   "walkers wait for their windows to be evaluated" if dom == "match_walkers" else "")
                                + ("; cf_walk (class-ordered best fit, DESIGN.md 4b) is ONE ordinary launch per cycle, one workgroup of 16 waves per pool, all state in "
                                   "LDS: the K x M pair evaluation of the reference is never made, so the bytes it moves are far below the algorithmic figure" if dom == "cf_walk" else ""),
                        "kernels_ms_per_cycle": {k: round(v[0] / max(1, min(args.steps, 3)), 4) for k, v in
                                                 sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]}}

    # ---- CPU baseline: the oracle (kind "port") on the box's host cores, the WHOLE cycle of rank 0's pools ------------------
    # The reference runs every pool's match handler on its own thread (tools.clj:799-806 chime-at per pool, scheduler.clj:2425-2435)
    # and Fenzo evaluates a job's hosts on an executor per CPU (scheduler.clj:2301-2324): the forms timed are `a` pools at once x `t`
    # evaluator threads per pool, a x t <= the host's cores, every one the FULL K of every pool; `value` = the fastest
    # (cpu_baseline_leg).  One library call per pool thread: no interpreter work inside the timed region.
    # ---- + parity of the TIMED configuration: the results of the last timed cycle (as fetched above, before the profiled pass)
    #      against the oracle, bit-exact — every pool of rank 0 when the concurrent baseline ran (its outputs are reused), else the
    #      first pool (first slot of a launch chain) and the last (last slot of another chain).  A fast wrong answer must not
    #      produce a number: a mismatch raises.
    mark("cpu_baseline leg + parity of the timed configuration")
    cpu = None
    parity_checked, parity_pools = False, []
    if rank == 0 and not (args.no_cpu_baseline and args.no_check):
        cpu, oracle_out = (None, {})
        if not args.no_cpu_baseline and world == 1 and not args.as_rank_of:
            cpu, oracle_out = cpu_baseline_leg(args, params, cluster, pools, my_pools, K, n_off)
        if not args.no_check:
            from oracle import checks
            check_pools = list(my_pools) if oracle_out else sorted({my_pools[0], my_pools[-1]})
            for pc in check_pools:
                r, j2o = fetched[pc]
                if pc in oracle_out:
                    o_ranked, o_j2o = oracle_out[pc]
                    assert np.array_equal(r, o_ranked), f"PARITY: rank of pool {pc} differs from the oracle"
                    assert np.array_equal(j2o, o_j2o), f"PARITY: assignments of pool {pc} differ from the oracle"
                else:
                    qc = cluster.quota_inputs(pc, cluster.last_pool_usage[pc], cluster.last_group_usage)
                    checks.check_pool_against_oracle(params, pools[pc], qc, r, j2o, K, threads=min(16, os.cpu_count() or 1))
                parity_pools.append({"pool": pc, "jobs_checked": int(len(j2o))})
            parity_checked = True

    # ---- SURVEY.md §8d's reporting matrix, beside the headline (never instead of it): the reference's default cap K = 1000
    #      (config.clj:113: launch / latency-bound, p50 and p95 in microseconds), K = 1e5, the reference's DEFAULT good-enough
    #      fitness 0.8 (config.clj:111), and BASELINE.json configs[1] / configs[2] as single pools.  N = 1 only.
    mark("extra configurations")
    extra = None
    if rank == 0 and world == 1 and not args.no_extras:
        extra = {}

        def timed(fn, n, warm=1):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(n):
                a = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - a)
            ts.sort()
            return ts

        def pct(ts, q):
            return ts[min(len(ts) - 1, int(q * len(ts)))]

        ts = timed(lambda: cluster.cycle(1000), 40, warm=3)
        extra["K=1000"] = {"what": f"the same {P}-pool cluster, 1000 considerable jobs per pool (config.clj:113)", "cycles": len(ts),
                           "p50_cycle_us": pct(ts, 0.5) * 1e6, "p95_cycle_us": pct(ts, 0.95) * 1e6,
                           "stage_ms_pool0": dict(zip(("rank", "match"), engines[my_pools[0]].last_timing()))}
        hc, _, _ = host_core_budget()
        nt_small = max(1, min(hc // max(1, P), 4))

        def with_cpu(row, gpu_s, leg):  # a CPU figure beside every operating point of the line (north_star: "next to the reference timed on the box's own host cores")
            if leg is not None:
                row["cpu_baseline"] = leg
                row["speedup_vs_cpu_baseline"] = leg["cycle_s"] / gpu_s
            return row

        no_cpu = args.no_cpu_baseline
        with_cpu(extra["K=1000"], pct(ts, 0.5), None if no_cpu else cpu_leg_pools(params, cluster, pools, my_pools, 1000, 1))
        ts = timed(lambda: cluster.cycle(100_000), 4)
        extra["K=1e5"] = {"what": f"the same cluster, 100000 considerable jobs per pool", "cycles": len(ts), "p50_cycle_ms": pct(ts, 0.5) * 1e3,
                          "p95_cycle_ms": pct(ts, 0.95) * 1e3}
        with_cpu(extra["K=1e5"], pct(ts, 0.5), None if no_cpu else cpu_leg_pools(params, cluster, pools, my_pools, 100_000, nt_small))
        p08 = A.default_params(good_enough_fitness=0.8, match_algo=args.match_algo)
        for e in engines.values():
            e.set_params(p08)
        ts = timed(lambda: cluster.cycle(1000), 40, warm=3)
        extra["K=1000, good_enough=0.8"] = with_cpu({"what": f"the same cluster at the reference's shipped defaults (config.clj:110-116): 1000 considerable jobs per pool, good-enough-fitness 0.8",
                                                     "cycles": len(ts), "p50_cycle_us": pct(ts, 0.5) * 1e6, "p95_cycle_us": pct(ts, 0.95) * 1e6}, pct(ts, 0.5),
                                                    None if no_cpu else cpu_leg_pools(p08, cluster, pools, my_pools, 1000, 1))
        ts = timed(lambda: cluster.cycle(K), 3)
        f08 = {p: engines[p].cycle_fetch() for p in my_pools}
        m08 = sum(int((f08[p][1] >= 0).sum()) for p in my_pools)
        extra["good_enough=0.8"] = {"what": f"the same cluster, K = {K} per pool, good-enough-fitness 0.8 (the reference's default, config.clj:111; "
                                            "the winner among equally good-enough hosts is oracle-defined: first in offer order)",
                                    "cycles": len(ts), "p50_cycle_ms": pct(ts, 0.5) * 1e3, "matched": m08, "parity_checked": False,
                                    "placement_stats_pool0": engines[my_pools[0]].match_stats()}
        # (good-enough < 1: the oracle's placement is its single-thread form — the first offer in array order above the threshold, an oracle-defined rule)
        with_cpu(extra["good_enough=0.8"], pct(ts, 0.5), None if no_cpu else cpu_leg_pools(p08, cluster, pools, my_pools, K, 1))
        if not args.no_check:  # the timed 0.8 cycle of rank 0's first pool against the single-thread oracle, every assignment
            from oracle import checks
            pc = my_pools[0]
            qc = cluster.quota_inputs(pc, cluster.last_pool_usage[pc], cluster.last_group_usage)
            checks.check_pool_against_oracle(p08, pools[pc], qc, f08[pc][0], f08[pc][1], K, threads=1)
            extra["good_enough=0.8"]["parity_checked"] = True
            extra["good_enough=0.8"]["parity"] = {"pool": pc, "jobs_checked": int(len(f08[pc][1])), "against": "oracle, single thread, bit-exact"}
        for e in engines.values():
            e.set_params(params)
        for name, kw in (("C2", dict(seed=0xC00C0002, n_pending=50000, n_running=20000, n_users=1000, n_offers=5000)),
                         ("C3", dict(seed=0xC00C0003, n_pending=200000, n_running=80000, n_users=2000, n_offers=20000, gpus=True,
                                     constraints=True))):
            pool_x = synth.make_pool(**kw)
            with Engine(params, device=dev_index) as ex:
                ex.cycle_stage(pool_x.tasks, pool_x.users, pool_x.pending_jobs, pool_x.offers, pool_x.groups)
                ts = timed(lambda: ex.cycle_run(pool_x.n_pending), 3)
                _, j2o_x, _ = ex.cycle_fetch()
                extra[name] = {"what": f"BASELINE.json configs[{1 if name == 'C2' else 2}]: single pool, {kw['n_pending']} pending x {kw['n_offers']} offers"
                                       + ("" if name == "C2" else ", gpu dimension + host / attribute / group constraints") + ", K = all pending",
                               "cycles": len(ts), "p50_cycle_ms": pct(ts, 0.5) * 1e3, "matched": int((j2o_x >= 0).sum()),
                               "pair_evaluations": int(len(j2o_x)) * kw["n_offers"],
                               "stage_ms": dict(zip(("rank", "match"), ex.last_timing())), "placement_stats": ex.match_stats()}
                # the class-ordered best fit (match_algo 3) on the same pool, where the call is eligible (C3's 20 000 offers are not)
                ex.set_params(A.default_params(good_enough_fitness=args.good_enough, match_algo=3))
                ts3 = timed(lambda: ex.cycle_run(pool_x.n_pending), 3)
                _, j2o_3, _ = ex.cycle_fetch()
                st3 = ex.match_stats()
                assert np.array_equal(j2o_3, j2o_x), f"PARITY: {name}: class-ordered best fit differs from the window rounds"
                extra[name]["classfit"] = {"p50_cycle_ms": pct(ts3, 0.5) * 1e3, "placed_by_classfit": st3.get("placement_form") == 3, "refused_bits": st3.get("classfit_refused"),
                                           "identical_to_window_rounds": True}
            if not no_cpu:
                from oracle import pyoracle
                nt_x = max(1, min(hc, 16))
                a = time.perf_counter()
                o_r, o_j, o_ph = pyoracle.cycle(params, pool_x.tasks, pool_x.users, pool_x.pending_jobs, pool_x.offers, pool_x.groups, K=pool_x.n_pending, nthreads=nt_x)
                cpu_s = time.perf_counter() - a
                if not args.no_check:
                    assert np.array_equal(o_j, j2o_x), f"PARITY: {name}: assignments differ from the oracle"
                    extra[name]["parity_checked"] = True
                with_cpu(extra[name], pct(ts, 0.5), {"cycle_s": cpu_s, "rank_s": o_ph["rank"], "match_s": o_ph["match"], "cores": nt_x, "kind": "port",
                                                     "sample": f"the whole cycle (not a sample), one library call, {nt_x} evaluator threads"})
            del pool_x
        try:
            extra["C5"] = extra_c5(dev_index, check=not args.no_check)
        except AssertionError:
            raise  # a parity failure must not produce a bench line
        except Exception as ex:  # (an extra must never cost the headline its line)
            extra["C5"] = {"error": repr(ex)}
        # ---- both placement forms on the headline cluster, whichever the engine's own choice (match_algo 0) took in the timed cycles: class-ordered best fit
        #      (match_algo 3; DESIGN.md §4b: one launch, a workgroup per pool, no evaluation launches) and the window rounds (2; served walkers)
        for key, algo in (("classfit", 3), ("window_rounds", 2)):
            px = A.default_params(good_enough_fitness=args.good_enough, match_algo=algo)
            for e in engines.values():
                e.set_params(px)
            ts = timed(lambda: cluster.cycle(K), 3)
            fx = {p: engines[p].cycle_fetch() for p in my_pools}
            stx = engines[my_pools[0]].match_stats()
            same = all(np.array_equal(fx[p][1], fetched[p][1]) for p in my_pools)
            assert same, f"PARITY: match_algo {algo} differs from the timed cycle's assignments"
            extra[key] = {"what": f"the same cluster and K = {K}, cook_params.match_algo = {algo} ({'class-ordered best fit' if algo == 3 else 'window rounds'})", "cycles": len(ts),
                          "p50_cycle_ms": pct(ts, 0.5) * 1e3, "placement_form_pool0": stx.get("placement_form"), "identical_to_timed_cycle": True,
                          "stage_ms_pool0": dict(zip(("rank", "match"), engines[my_pools[0]].last_timing())), "placement_stats_pool0": stx}
            if algo == 3:
                extra[key]["placed_by_classfit_pool0"] = stx.get("placement_form") == 3
            ts = timed(lambda: cluster.cycle(1000), 40, warm=3)
            extra[key]["K=1000"] = {"p50_cycle_us": pct(ts, 0.5) * 1e6, "p95_cycle_us": pct(ts, 0.95) * 1e6}
        for e in engines.values():
            e.set_params(params)

    # ---- the boundary, not just the core (never `value`): what a cycle costs when the host hands over what CHANGED since the last
    #      one and takes the assignments back.  cook_cycle_update per pool (1 % of the tasks leave, as many arrive — half of them new
    #      submissions —, fresh offers) from page-locked columns, the cycle, cook_cycle_fetch into page-locked buffers; beside it the
    #      cost of restaging EVERYTHING from pageable and from page-locked memory (cook_cycle_stage).
    mark("boundary leg")
    boundary = None
    if rank == 0 and world == 1 and (args.boundary or not args.no_extras):
        from cook_amd.engine import PinnedArena
        arena = PinnedArena()
        try:
            rng = np.random.default_rng(7)
            n_delta = max(1, (n_pend + n_run) // 100)
            deltas, pinned_pools, outs = {}, {}, {}
            for p in my_pools:
                extra_pool = synth.make_pool(seed=0xD0000 + p, n_pending=n_delta // 2, n_running=n_delta - n_delta // 2, n_users=args.users,
                                             n_offers=n_off, gpus=not args.no_constraints, constraints=not args.no_constraints,
                                             id_base=27_592_186_044_416)
                aj = extra_pool.pending_jobs
                ng = pools[p].groups.n if pools[p].groups is not None else 0
                if aj.group is not None:
                    aj.group = np.where((aj.group != A.NONE_U32) & (ng > 0), aj.group % max(1, ng), A.NONE_U32).astype(np.uint32)
                deltas[p] = (arena.copy(np.sort(rng.choice(n_pend + n_run, size=n_delta, replace=False)).astype(np.uint32)),
                             arena.pin(extra_pool.tasks), arena.pin(aj), arena.pin(pools[p].offers))
                pinned_pools[p] = (arena.pin(pools[p].tasks), arena.pin(pools[p].pending_jobs), arena.pin(pools[p].offers))
                outs[p] = (arena.empty(n_pend + n_delta, np.uint32), arena.empty(n_pend + n_delta, np.int32))

            def restage(pinned):
                for p in my_pools:
                    t_, j_, o_ = pinned_pools[p] if pinned else (pools[p].tasks, pools[p].pending_jobs, pools[p].offers)
                    engines[p].cycle_stage(t_, pools[p].users, j_, o_, pools[p].groups)
                torch.cuda.synchronize()

            b0 = time.perf_counter()
            restage(False)
            b1 = time.perf_counter()
            restage(True)
            b2 = time.perf_counter()
            staged_bytes = sum(sum(a.nbytes for a in vars(x).values() if isinstance(a, np.ndarray))
                               for p in my_pools for x in pinned_pools[p])
            cluster.cycle(K)
            torch.cuda.synchronize()
            cluster.update(deltas)  # one untimed update: the first call of an engine allocates the columns' second buffers
            torch.cuda.synchronize()
            restage(True)
            cluster.cycle(K)
            torch.cuda.synchronize()
            t_upd = t_cyc = t_fetch = 0.0
            upd_samples = []
            upd_outliers = []
            cpu_stat0 = cpu_stat()
            n_b = 50  # (fifty samples of the update: an occasional slow call — 10 ms against 1 — was seen in round 4; max / median is in the line)
            gc.collect()
            gc.disable()
            for it in range(n_b):
                c0 = time.perf_counter()
                cluster.update(deltas)
                torch.cuda.synchronize()
                c1 = time.perf_counter()
                cluster.cycle(K)
                torch.cuda.synchronize()
                c2 = time.perf_counter()
                for p in my_pools:
                    engines[p].cycle_fetch(out=outs[p])
                c3 = time.perf_counter()
                t_upd += c1 - c0
                upd_samples.append(round((c1 - c0) * 1e3, 3))
                if len(upd_samples) > 5 and upd_samples[-1] > 2.0 * float(np.median(upd_samples)):
                    # a slow call: where its time went, per pool (cook_match_stats_ex [26..28]: the library times every update call)
                    st_ = {p: engines[p].match_stats() for p in my_pools}
                    upd_outliers.append({"sample": it, "ms": upd_samples[-1],
                                         "per_pool_us_in_call": [st_[p].get("update_us") for p in my_pools],
                                         "per_pool_us_in_stream_syncs": [st_[p].get("update_sync_us") for p in my_pools],
                                         "per_pool_device_allocations": [st_[p].get("update_allocs") for p in my_pools],
                                         "per_pool_slowest_phase": [st_[p].get("update_slowest_phase") for p in my_pools],
                                         "per_pool_slowest_phase_us": [st_[p].get("update_slowest_phase_us") for p in my_pools],
                                         "cgroup_cpu_stat": cpu_stat()})
                t_cyc += c2 - c1
                t_fetch += c3 - c2
                if it + 1 < n_b:  # back to the benchmark's state for the next measurement: a full restage (not timed)
                    restage(True)
            gc.enable()
            upd_med = float(np.median(upd_samples))  # (the median; the samples are in the line)
            boundary = {"ms_per_step_incl_transfers": upd_med + (t_cyc + t_fetch) / n_b * 1e3,
                        "update_ms": upd_med, "update_ms_mean": t_upd / n_b * 1e3, "update_ms_max": float(max(upd_samples)), "update_ms_max_over_median": float(max(upd_samples)) / max(1e-9, upd_med), "update_outliers": upd_outliers, "cgroup_cpu_stat_before_after": [cpu_stat0, cpu_stat()], "update_ms_samples": upd_samples, "cycle_ms": t_cyc / n_b * 1e3, "fetch_ms": t_fetch / n_b * 1e3,
                        "delta": f"per pool: {n_delta} task rows leave, {n_delta} arrive ({n_delta // 2} of them pending jobs), {n_off} fresh offers",
                        "restage_all_pageable_ms": (b1 - b0) * 1e3, "restage_all_pinned_ms": (b2 - b1) * 1e3,
                        "restage_bytes": int(staged_bytes), "restage_pinned_GBps": staged_bytes / max(1e-9, b2 - b1) / 1e9,
                        "note": "host wall time around the C ABI calls (cook_cycle_update / the cycle / cook_cycle_fetch) after one untimed "
                                "update; page-locked memory from cook_host_alloc"}
            restage(True)  # leave the engines on the benchmark's own inputs, with a finished cycle (the rows below read its results)
            cluster.cycle(K)
            torch.cuda.synchronize()
        finally:
            arena.close()  # (the engine reads host arrays only during a call)

    mark("adjacent rows")
    # ---- the rows either side of the path, timed once on rank 0's first pool; not part of `value` ----
    adjacent = None
    if rank == 0 and not args.no_adjacent:
        e0 = engines[my_pools[0]]
        _, j2o0, _ = e0.cycle_fetch()
        unm = np.nonzero(j2o0 < 0)[0][:64].astype(np.uint32)
        e0.match_explain(unm[:1])  # first call allocates
        a0 = time.perf_counter()
        why = e0.match_explain(unm)
        a1 = time.perf_counter()
        e0.match_metrics(n_users=args.users, n_gpu_models=2)
        a2 = time.perf_counter()
        met = e0.match_metrics(n_users=args.users, n_gpu_models=2)
        a3 = time.perf_counter()
        nodes, pods_, op = synth.make_cluster_state(seed=0xC00C, n_nodes=n_off, n_pods=n_run, disk=True, n_attr_keys=8, max_pods=110)
        e0.offers_stage(nodes, pods_, op)
        e0.offers_run()
        e0.offers_run()
        adjacent = {"explain_64_unmatched_jobs_ms": (a1 - a0) * 1e3, "explain_hosts_refusing_first_job": int(why[0].sum()) if len(why) else 0,
                    "metrics_ms": (a3 - a2) * 1e3, "metrics_matched": met["matched"],
                    "offers_build_ms": e0.offers_timing(), "offers_build_shape": f"{n_off} nodes x {n_run} pods (one pool)",
                    "note": "cook_match_explain / cook_match_metrics / cook_offers_run on pool 0 after the timed region (host wall time incl. sync; offers: HIP events)"}

    if rank == 0:
        value = args.steps / elapsed
        lat_ms = sorted(x * 1e3 for x in lat)
        out = {
            "metric": "match-cycles/sec at 1M pending x 50k offers" if not weak else f"match-cycles/sec at {world}M pending x {50 * world}k offers (weak scaling: 1M x 50k per GPU)",
            "value": value, "unit": "cycles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "p50_cycle_latency_ms": lat_ms[len(lat_ms) // 2], "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL on the SIMT emulator (CPU, gloo): plumbing only, not a measurement",
            "config": {"workload": f"{P} pools x ({n_pend} pending + {n_run} running tasks, {n_off} offers), {args.users} users; "
                                   f"rank all tasks + match K={K} per pool"
                                   + ("" if args.no_constraints else "; gpu dim + EQUALS/novel-host/unique-group constraints"),
                       "pools": P, "pending_total": n_pend * P, "running_total": n_run * P, "offers_total": n_off * P,
                       "users": args.users, "considerable_per_pool": K, "good_enough_fitness": args.good_enough, "match_algo": args.match_algo,
                       "placement_form": {0: "window rounds", 1: "serial sweep", 3: "class-ordered best fit"}.get(timed_form, timed_form),
                       "parallelism": f"pools sharded over {world} GPU(s); per rank up to {cluster.max_chains} pools as launch chains of window rounds of their own; six or more pools of a "
                                      f"GPU by class-ordered best fit (ONE launch, a workgroup of 16 waves per pool, no evaluation launches) where a pool's numbers and constraints allow it, "
                                      f"else as served walkers (one persistent walker workgroup per pool beside evaluation launches)" + ("" if cluster.served else "; COOK_MATCH_SERVED=0: lockstep groups instead"), "pair_evaluations_per_cycle": considered * n_off},
            # (weak scaling: the cycle grows with the node — the figure that compares across N is pairs per second, not cycles)
            "pair_evaluations_per_s": considered * n_off * value,
            "last_cycle": {"ranked": ranked_n, "considered": considered, "matched": matched,
                           "stage_ms_pool0": {"rank": stage_ms[my_pools[0]][0], "match": stage_ms[my_pools[0]][1]},
                           "placement_stats_pool0": timed_stats},
            "phase_ms": dict(zip(("pool_usage_allreduce", "rank", "placement", "user_usage_allreduce"),
                                 (float(np.median([ph[x] for ph in phases])) for x in range(4)))),
            # the rank parts of the rank's pools as ONE joint sequence of launches (cook_cycle_run_rank_multi; COOK_RANK_BATCH=0: a thread and a
            # stream per pool): launches made, of them for more than one pool, operations issued on their own, stream synchronisations
            "rank_batch": rank_batch_report([engines[p].match_stats() for p in my_pools]),
            "setup_s": gen_s, "pool_usage_memo": os.environ.get("COOK_POOL_USAGE_MEMO") != "0",
            "collective": collective, "roofline": roofline, "cpu_baseline": cpu, "adjacent_rows": adjacent, "extra_configs": extra, "boundary": boundary,
            "parity_checked": parity_checked, "parity": {"against": "oracle (bit-exact rank order + every assignment)", "pools": parity_pools},
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(out))
    for e in engines.values():
        e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

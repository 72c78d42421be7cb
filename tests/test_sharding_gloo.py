"""N>1 path on CPU: world_size 2 over gloo.  Pools shard over ranks (pool p -> rank p mod world), the ONLY collective is
the all-reduce of per-pool running usage into quota-group usage (scheduler.clj:2125-2157), and the ranked queues every
rank produces must equal the ones a single process produces for the same cluster.

The per-pool compute here is the CPU oracle behind the PoolEngine protocol (tests may use the oracle; the product path,
cook_amd.engine.Engine, needs a GPU) — what is under test is cook_amd/sharding.py: partitioning, the contribution
matrix, the collective, and the quota inputs each pool's rank stage receives.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cook_amd import _abi as A  # noqa: E402
from cook_amd import sharding, synth  # noqa: E402

N_POOLS = 4


class OracleEngine:
    """PoolEngine backed by the oracle: rank + take K (the match is not needed to test the sharding)."""

    def __init__(self, pool, params):
        from oracle import pyoracle
        self.o, self.pool, self.params = pyoracle, pool, params
        self.quota = None
        self.ranked = None

    def rank_pool_usage(self):
        return self.o.pool_usage(self.pool.tasks)

    def rank_set_quota(self, q):
        self.quota = q

    def rank_user_usage(self, n_users, device_ptr=None):
        return self.o.user_usage(self.pool.tasks, n_users)

    def cycle_run(self, k):
        r, _ = self.o.rank(self.params, self.pool.tasks, self.pool.users, self.quota)
        self.ranked = r[:k]


def make_cluster():
    pools = {p: synth.make_pool(seed=0x5A4D + p, n_pending=600, n_running=300 + 50 * p, n_users=25, n_offers=32)
             for p in range(N_POOLS)}
    # pools 0,1,3 share quota group 0 with a BINDING count quota; pool 2 has only its own pool quota
    total_run = sum(pools[p].n_running for p in (0, 1, 3))
    groups = sharding.QuotaGroups(
        pool_group={0: 0, 1: 0, 3: 0},
        group_quota={0: A.quota(count=total_run + 150)},
        pool_quota={2: A.quota(count=pools[2].n_running + 40), 0: A.quota(count=10_000_000)})
    return pools, groups


def run_rank(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pools, groups = make_cluster()
    params = A.default_params()
    mine = sharding.pools_of_rank(N_POOLS, world, rank)
    cl = sharding.ShardedCluster({p: OracleEngine(pools[p], params) for p in mine}, groups, world=world, rank=rank)
    cl.n_users = 25
    cl.cycle(500)
    np.savez(os.path.join(out_dir, f"w{world}_r{rank}.npz"), group_usage=cl.last_group_usage, user_usage=cl.last_user_usage,
             **{f"ranked_{p}": cl.engines[p].ranked for p in mine})
    cl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_pools_of_rank_partition():
    for world in (1, 2, 3, 4, 8):
        seen = sorted(p for r in range(world) for p in sharding.pools_of_rank(8, world, r))
        assert seen == list(range(8))
    assert sharding.pools_of_rank(8, 8, 5) == [5]
    assert sharding.pools_of_rank(8, 2, 1) == [1, 3, 5, 7]
    with pytest.raises(ValueError):
        sharding.pools_of_rank(8, 2, 2)


def test_group_usage_matrix():
    g = sharding.QuotaGroups(pool_group={0: 1, 2: 1, 5: 0})
    m = sharding.group_usage_matrix(g, {0: (1, 2, 3, 4), 2: (10, 20, 30, 40), 3: (7, 7, 7, 7), 5: (1, 1, 1, 1)})
    assert m.shape == (2, 4)
    assert m[1].tolist() == [11, 22, 33, 44] and m[0].tolist() == [1, 1, 1, 1]


@pytest.mark.timeout(300)
def test_world2_gloo_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path)
    run_rank(0, 1, _free_port(), out)  # single process: all pools local, no collective
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=run_rank, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    one = np.load(os.path.join(out, "w1_r0.npz"))
    got = {}
    for r in range(2):
        z = np.load(os.path.join(out, f"w2_r{r}.npz"))
        assert np.array_equal(z["group_usage"], one["group_usage"])  # every rank holds the cluster-wide group usage
        assert np.array_equal(z["user_usage"], one["user_usage"])    # ... and the cross-pool per-user usage totals [U, 3]
        got.update({k: z[k] for k in z.files if k.startswith("ranked_")})
    assert sorted(got) == [f"ranked_{p}" for p in range(N_POOLS)]
    pools, groups = make_cluster()
    from oracle import pyoracle
    want = sum(pyoracle.user_usage(pools[p].tasks, 25) for p in range(N_POOLS))
    assert one["user_usage"].shape == (25, 3) and np.array_equal(one["user_usage"], want) and want[:, 0].sum() > 0
    for p in range(N_POOLS):
        assert np.array_equal(got[f"ranked_{p}"], one[f"ranked_{p}"]), f"pool {p}"
    # the group quota is binding: the three member pools together keep at most 150 pending jobs ... per pool the filter
    # is a prefix condition on that pool's queue against the cluster-wide usage, so each keeps at most 150
    for p in (0, 1, 3):
        assert 0 < len(one[f"ranked_{p}"]) <= 150
    assert 0 < len(one["ranked_2"]) <= 40


def test_group_without_running_tasks_skips_the_group_filter():
    """filter-based-on-quota applies the group filter only when (and quota-group-quota quota-group-usage)
    (scheduler.clj:2134-2157); pool-name->usage has no entry for pools without running tasks, so at cold start the group
    usage is nil and the queue is NOT cut at the group quota."""
    pools = {p: synth.make_pool(seed=0x77 + p, n_pending=300, n_running=0, n_users=9, n_offers=8) for p in range(2)}
    groups = sharding.QuotaGroups(pool_group={0: 0, 1: 0}, group_quota={0: A.quota(count=50)})
    params = A.default_params()
    cl = sharding.ShardedCluster({p: OracleEngine(pools[p], params) for p in pools}, groups)
    cl.cycle(10_000)
    for p in pools:
        assert cl.quota_inputs(p, cl.last_pool_usage[p], cl.last_group_usage) is None
        assert len(cl.engines[p].ranked) == 300
    cl.close()
    # one running task anywhere in the group switches the filter on for every member pool
    pools[1] = synth.make_pool(seed=0x79, n_pending=300, n_running=1, n_users=9, n_offers=8)
    cl = sharding.ShardedCluster({p: OracleEngine(pools[p], params) for p in pools}, groups)
    cl.cycle(10_000)
    assert all(len(cl.engines[p].ranked) == 49 for p in pools)
    cl.close()


def test_rank_batches_from_several_threads(monkeypatch):
    """ShardedCluster.cycle with eight pools: the rank parts as FOUR pool batches of two pools (cook_cycle_run_rank_multi), one on the calling
    thread and three handed to the pool's threads, every pool's per-user usage collected — against the same cluster with one batch of
    eight.  Real engines on the SIMT emulator (one launch at a time: a lock stands in for the GPU's concurrency)."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from cook_amd import engine as E, workload
    from tests.simt_emu import build_emu
    so = build_emu.build()
    spec = workload.ClusterSpec(pools=8, pending=4000, running=1600, offers=320, users=60)
    pools = workload.make_pools(spec, range(spec.pools))
    params = A.default_params(good_enough_fitness=1.0)
    lock, calls = threading.Lock(), []
    real = E.cycle_run_rank_multi

    def guarded(engines, *a, **kw):
        with lock:
            calls.append((threading.get_ident(), len(engines)))
            return real(engines, *a, **kw)
    monkeypatch.setattr(E, "cycle_run_rank_multi", guarded)
    results = {}
    for batches in (4, 1):
        monkeypatch.setenv("COOK_RANK_BATCHES", str(batches))
        engines = {p: E.Engine(params, lib_path=so) for p in pools}
        try:
            for p, pool in pools.items():
                engines[p].cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)
            cl = sharding.ShardedCluster(engines, workload.quota_groups(spec), serial=True)
            if batches > 1:
                cl._tp_rank = ThreadPoolExecutor(max_workers=4)  # (everything else of the cycle stays on the calling thread)
            cl.n_users = spec.users
            del calls[:]
            cl.cycle(spec.per_pool[0])
            assert sorted(n for _, n in calls) == ([2, 2, 2, 2] if batches == 4 else [8]), calls
            if batches == 4:
                assert len({t for t, _ in calls}) >= 2 and threading.get_ident() in {t for t, _ in calls}, calls
            results[batches] = ([engines[p].cycle_fetch() for p in pools], np.array(cl.last_user_usage), np.array(cl.last_group_usage))
            cl.close()
        finally:
            for e in engines.values():
                e.close()
    for (a, b) in zip(results[4][0], results[1][0]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert np.array_equal(results[4][1], results[1][1]) and np.array_equal(results[4][2], results[1][2])
    from oracle import pyoracle
    want = sum(pyoracle.user_usage(pools[p].tasks, spec.users) for p in pools)
    assert np.array_equal(results[4][1], want)


@pytest.mark.timeout(900)
def test_bench_self_launches_world2_over_gloo():
    """`python bench.py --gpus 2` from a plain shell (no WORLD_SIZE): bench.py re-executes itself under torch.distributed.run, one
    process per rank; both ranks run the sharded cycle (quota-group + per-user all-reduces over gloo here, RCCL on the GPU box), rank 0
    prints the one JSON line with the whole-job rate and checks its pools against the oracle.  The engines are the SIMT-emulator build
    of the library (test infrastructure) at a small cluster: what is rehearsed is the launch / reporting / collective plumbing."""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests", "simt_emu"))
    import build_emu
    lib = build_emu.build()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pools", "4", "--pending", "1600",
           "--running", "800", "--offers", "96", "--users", "20", "--no-extras", "--no-adjacent", "--engine-lib", lib, "--dist-backend", "gloo"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line, rank 1 none
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["steps"] == 2 and doc["value"] > 0 and doc["scaling"] == "strong"
    assert doc["parity_checked"] is True and [p["pool"] for p in doc["parity"]["pools"]] == [0, 2]  # rank 0 of 2 holds pools 0 and 2
    assert doc["cpu_baseline"] is None  # N = 1 only
    assert "REHEARSAL" in doc["data"]
    assert doc["last_cycle"]["considered"] > 0 and doc["phase_ms"]["user_usage_allreduce"] >= 0
    # the line proves who took part: backend, world size as the process group saw it, every rank's pools, and that the all-reduced
    # quota-group matrix is the sum over ALL pools' running usage (gathered from every rank, re-added on rank 0)
    co = doc["collective"]
    assert co["backend"] == "gloo" and co["world_size"] == 2 and co["pools_of_rank"] == [[0, 2], [1, 3]]
    assert co["group_usage_equals_sum_over_all_pools"] is True and co["group_usage_allreduced"][0][0] == 800  # 4 pools x 200 running tasks
    # N = 1 through the same entry point: all pools local, the concurrent cpu_baseline leg and the parity of every pool
    cmd1 = [c for c in cmd]
    cmd1[cmd1.index("--gpus") + 1] = "1"
    r1 = subprocess.run(cmd1, env=env, capture_output=True, text=True, timeout=800)
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["n_gpus"] == 1 and [p["pool"] for p in d1["parity"]["pools"]] == [0, 1, 2, 3]
    cb = d1["cpu_baseline"]
    assert cb["kind"] == "port" and cb["variants"][0]["form"] == "pools concurrent" and cb["variants"][0]["pools_at_once"] == 4
    assert cb["variants"][0]["threads_per_pool"] == 1 and all(v["pools_at_once"] * v["threads_per_pool"] <= cb["host_cores"] for v in cb["variants"])
    assert all(set(v["slowest_pool_s"]) == {"rank", "gather", "match"} for v in cb["variants"])  # the library call's own clocks, per form
    assert d1["collective"]["world_size"] == 1 and d1["collective"]["pools_of_rank"] == [[0, 1, 2, 3]]
    assert cb["value"] == max(v["cycles_per_s"] for v in cb["variants"]) and cb["cores"] >= 1
    assert d1["last_cycle"]["matched"] == doc["last_cycle"]["matched"]  # the sharded job places what the single process places
    # weak scaling: --pools pools PER rank — two ranks hold a cluster of 8 pools with twice the jobs, tasks and offers; the collective's record lists them all
    cmdw = cmd + ["--scaling", "weak", "--no-cpu-baseline"]
    rw = subprocess.run(cmdw, env=env, capture_output=True, text=True, timeout=800)
    assert rw.returncode == 0, rw.stderr[-3000:]
    dw = json.loads([ln for ln in rw.stdout.splitlines() if ln.startswith("{")][0])
    assert dw["scaling"] == "weak" and dw["n_gpus"] == 2 and dw["config"]["pools"] == 8 and dw["config"]["pending_total"] == 2 * doc["config"]["pending_total"]
    assert dw["collective"]["pools_of_rank"] == [[0, 2, 4, 6], [1, 3, 5, 7]] and dw["collective"]["group_usage_equals_sum_over_all_pools"] is True
    assert dw["parity_checked"] is True and dw["pair_evaluations_per_s"] > 0

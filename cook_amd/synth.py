"""Synthetic pool generator for the benchmark and the parity tests (SURVEY.md §8d / BASELINE.md §4).

Shapes follow the reference's own generators: simulator/src/main/cook/sim/schedule.clj:58-83,
simulator/config/larger_cluster_simulation.edn, scheduler/test/cook/test/benchmark.clj:41-45.
Everything is seeded (numpy PCG64) so the CPU oracle and the HIP engine see identical inputs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _abi as A


@dataclass
class Pool:
    tasks: A.Tasks          # running ++ pending (synthetic tasks)
    users: A.Users
    pending_jobs: A.Jobs    # one entry per pending task, in pending-ordinal order
    offers: A.Offers
    groups: Optional[A.Groups]
    n_running: int
    n_pending: int


def _zipf_users(rng, n, n_users, s=1.1):
    p = 1.0 / np.arange(1, n_users + 1) ** s
    p /= p.sum()
    perm = rng.permutation(n_users)
    return perm[rng.choice(n_users, size=n, p=p)].astype(np.uint32)


def make_pool(seed: int, n_pending: int, n_running: int, n_users: int, n_offers: int, *, gpus: bool = False,
              constraints: bool = False, fractional: bool = False, no_shares: bool = False, quota_frac: float = 0.02,
              n_attr_keys: int = 8, tie_heavy: bool = False, id_base: int = 17_592_186_044_416) -> Pool:
    rng = np.random.default_rng(seed)
    n = n_running + n_pending
    # jobs: cpus integer 1-8 ~ clamp(round(N(3,1))); mem MiB integer ~ clamp(round(N(10240,4096)), 512, 65536)
    cpus = np.clip(np.rint(rng.normal(3.0, 1.0, n)), 1, 8)
    mem = np.clip(np.rint(rng.normal(10240.0, 4096.0, n)), 512, 65536)
    if tie_heavy:  # few distinct shapes -> many equal DRUs -> exercises the sorted-merge tie rule
        cpus = rng.integers(1, 3, n).astype(np.float64)
        mem = (rng.integers(1, 3, n) * 1024).astype(np.float64)
    if fractional:  # non-dyadic values: prefix sums round, the exact sequential fix-up path must kick in
        cpus = cpus + rng.integers(0, 10, n) / 10.0
        mem = mem + rng.integers(0, 10, n) / 10.0
    g = np.zeros(n)
    gmodel = np.zeros(n, dtype=np.uint32)
    if gpus:
        has = rng.random(n) < 0.10
        g[has] = rng.choice([1.0, 2.0, 4.0, 8.0], size=int(has.sum()), p=[.5, .25, .15, .1])
        gmodel[has] = rng.choice([1, 2], size=int(has.sum()), p=[.7, .3])
    user = _zipf_users(rng, n, n_users)
    priority = rng.integers(0, 101, n).astype(np.int32)
    pending = np.zeros(n, dtype=np.uint8)
    pending[n_running:] = 1
    day_ms = 24 * 3600 * 1000
    start = np.where(pending == 1, 0, 1_600_000_000_000 + rng.integers(0, day_ms, n)).astype(np.int64)
    task_id = (id_base + 2_000_000_000 + rng.permutation(n)).astype(np.int64)  # unique
    job_id = (id_base + rng.permutation(n)).astype(np.int64)
    host = rng.integers(0, max(1, n_offers), n).astype(np.uint32)
    # interleave running and pending in the input arrays (the engine must not rely on their order)
    order = rng.permutation(n)
    tasks = A.Tasks(cpus=cpus[order], mem=mem[order], gpus=g[order] if gpus else None, user=user[order],
                    priority=priority[order], start_ms=start[order], task_id=task_id[order], job_id=job_id[order],
                    pending=pending[order], host=host[order])
    # users: default share {cpus 64, mem 262144, gpus 8}; 5% of users x4; quotas: 2% of users count=50
    div_c = np.full(n_users, 64.0)
    div_m = np.full(n_users, 262144.0)
    div_g = np.full(n_users, 8.0)
    big = rng.random(n_users) < 0.05
    div_c[big] *= 4
    div_m[big] *= 4
    div_g[big] *= 4
    if no_shares:
        div_c[:] = A.DMAX
        div_m[:] = A.DMAX
        div_g[:] = A.DMAX
    qcount = np.full(n_users, 2.0 ** 31 - 1)
    qcount[rng.random(n_users) < quota_frac] = 50.0
    users = A.Users(div_cpus=div_c, div_mem=div_m, div_gpus=div_g, quota_count=qcount)
    # offers: host cpus in {16,32,64,96}, mem = cpus*4096; residual = floor(total * U(0.05,1))
    tot_c = rng.choice([16.0, 32.0, 64.0, 96.0], size=n_offers, p=[.2, .4, .3, .1])
    frac = rng.uniform(0.05, 1.0, n_offers)
    oc = np.floor(tot_c * frac)
    om = np.floor(tot_c * 4096.0 * frac)
    run_c, run_m = tot_c - oc, tot_c * 4096.0 - om
    run_n = np.rint(run_c / 3.0).astype(np.int32)
    o_gm = np.zeros(n_offers, dtype=np.uint32)
    o_gc = np.zeros(n_offers)
    attr = None
    if gpus:
        gh = rng.random(n_offers) < 0.10
        o_gm[gh] = rng.choice([1, 2], size=int(gh.sum()), p=[.7, .3])
        o_gc[gh] = rng.choice([1.0, 2.0, 4.0, 8.0], size=int(gh.sum()))
        run_n[gh] = np.where(rng.random(int(gh.sum())) < 0.5, 0, run_n[gh])  # gpu hosts must be empty to take a gpu job
    if constraints:
        card = [2, 3, 4, 8, 16, 32, 64, 0][:n_attr_keys]
        attr = np.zeros((n_offers, n_attr_keys), dtype=np.uint32)
        for k, c in enumerate(card):
            attr[:, k] = (np.arange(n_offers) + 1) if c == 0 else rng.integers(1, c + 1, n_offers)
    offers = A.Offers(cpus=oc, mem=om, host=np.arange(n_offers, dtype=np.uint32), k8s=np.ones(n_offers, dtype=np.uint8),
                      gpu_model=o_gm if gpus else None, gpu_count=o_gc if gpus else None, attr=attr,
                      run_cpus=run_c, run_mem=run_m, run_count=run_n)
    # pending jobs (aligned with pending ordinal = order of appearance in `tasks`)
    pidx = np.nonzero(tasks.pending)[0]
    P = len(pidx)
    groups = None
    kw = {}
    if constraints:
        equals, novel = [], []
        grp = np.full(P, A.NONE_U32, dtype=np.uint32)
        for q in range(P):
            e = []
            if rng.random() < 0.20:
                for _ in range(int(rng.integers(1, 3))):
                    k = int(rng.integers(0, min(6, n_attr_keys)))
                    e.append((k, int(rng.integers(1, [2, 3, 4, 8, 16, 32][k] + 1))))
            equals.append(e)
            novel.append([int(h) for h in rng.integers(0, n_offers, int(rng.integers(1, 4)))] if rng.random() < 0.02 else [])
        # 5% of jobs in unique-placement groups of size 2-6
        n_g = 0
        gtypes = []
        q = 0
        members = rng.permutation(P)[: int(0.05 * P)]
        while q < len(members):
            size = int(rng.integers(2, 7))
            grp[members[q:q + size]] = n_g
            gtypes.append(1)
            n_g += 1
            q += size
        if n_g:
            run_hosts = [[int(h) for h in rng.integers(0, n_offers, int(rng.integers(0, 2)))] for _ in range(n_g)]
            groups = A.Groups(type=np.array(gtypes, dtype=np.uint8), run_hosts=run_hosts)
        kw.update(group=grp)
        jobs = A.Jobs.with_constraints(tasks.cpus[pidx], tasks.mem[pidx], equals=equals, novel=novel,
                                       gpus=tasks.gpus[pidx] if gpus else None,
                                       gpu_model=gmodel[order][pidx] if gpus else None, user=tasks.user[pidx], **kw)
    else:
        jobs = A.Jobs(cpus=tasks.cpus[pidx], mem=tasks.mem[pidx], gpus=tasks.gpus[pidx] if gpus else None,
                      gpu_model=gmodel[order][pidx] if gpus else None, user=tasks.user[pidx])
    return Pool(tasks=tasks, users=users, pending_jobs=jobs, offers=offers, groups=groups, n_running=n_running,
                n_pending=n_pending)


def make_cluster_state(seed: int, n_nodes: int, n_pods: int, *, gpus: bool = True, disk: bool = False, fractional: bool = True,
                       n_attr_keys: int = 0, max_pods: int = 32, corrupt: float = 0.0):
    """Node / pod state of one Kubernetes pool as the offer construction sees it (kubernetes/compute_cluster.clj:68-190).
    Node shapes follow SURVEY.md §8d's offers (cpus in {16,32,64,96}, mem = cpus x 4096 MiB, 10 % gpu hosts); pod requests
    follow the job shapes, with the 0.1-cpu sidecar the reference adds to every pod (kubernetes/api.clj:1346-1349 shapes) when
    `fractional`, so that consumption sums are NOT exact in every order.  `corrupt` = fraction of gpu / disk pods placed on
    nodes of another model / type.  -> (Nodes, Pods, CookOfferParams)"""
    rng = np.random.default_rng(seed)
    cpus = rng.choice([16.0, 32.0, 64.0, 96.0], size=n_nodes, p=[0.2, 0.4, 0.3, 0.1])
    mem = cpus * 4096.0
    n_models, n_types = (2 if gpus else 0), (2 if disk else 0)
    ngpu = np.zeros(n_nodes, np.int32)
    nmodel = np.zeros(n_nodes, np.uint32)
    if gpus:
        is_g = rng.random(n_nodes) < 0.10
        ngpu[is_g] = rng.choice([1, 2, 4, 8], size=int(is_g.sum()))
        nmodel[is_g] = rng.choice([1, 2], size=int(is_g.sum()), p=[0.7, 0.3])
        # a few unsound nodes: gpu taint / label without allocatable gpus
        odd = rng.random(n_nodes) < 0.01
        ngpu[odd] = 0
    ndisk = np.full(n_nodes, -1.0)
    ndtype = np.zeros(n_nodes, np.uint32)
    if disk:
        has = rng.random(n_nodes) < 0.8
        ndisk[has] = rng.choice([256000.0, 512000.0], size=int(has.sum()))
        ndtype[has] = rng.choice([1, 2], size=int(has.sum()))
    flags = np.zeros(n_nodes, np.uint8)
    for bit, frac in ((A.NODE_UNSCHEDULABLE, 0.02), (A.NODE_OTHER_TAINTS, 0.02), (A.NODE_BLOCKLIST_LABEL, 0.01)):
        flags[rng.random(n_nodes) < frac] |= bit
    flags[(nmodel != 0) | (rng.random(n_nodes) < 0.005)] |= A.NODE_GPU_TAINT
    attr = rng.integers(0, 5, size=(n_nodes, n_attr_keys)).astype(np.uint32) if n_attr_keys else None
    nodes = A.Nodes(cpus=cpus, mem=mem, host=np.arange(n_nodes) * 2 + 1, gpus=ngpu, gpu_model=nmodel, disk=ndisk, disk_type=ndtype,
                    flags=flags, attr=attr)
    # pods: node uniform (a few without a node of this pool), a hot spot so that some nodes hit the pod limit
    pnode = rng.integers(0, max(1, n_nodes), size=n_pods).astype(np.uint32)
    hot = rng.random(n_pods) < 0.05
    pnode[hot] = rng.integers(0, max(1, n_nodes // 50 + 1), size=int(hot.sum()))
    pnode[rng.random(n_pods) < 0.02] = A.NONE_U32
    if n_nodes == 0:
        pnode[:] = A.NONE_U32
    pc = np.clip(np.round(rng.normal(3, 1, n_pods)), 1, 8)
    pm = np.clip(np.round(rng.normal(10240, 4096, n_pods)), 512, 65536)
    if fractional:
        pc = pc + 0.1                      # sidecar request: 0.1 is not dyadic
        pm = pm + rng.choice([0.0, 0.5, 100.3], size=n_pods)
    pg = np.zeros(n_pods, np.int32)
    pgm = np.zeros(n_pods, np.uint32)
    if gpus and n_nodes:
        on = pnode < n_nodes
        want = on & (nmodel[np.minimum(pnode, n_nodes - 1)] != 0) & (rng.random(n_pods) < 0.5)
        pg[want] = rng.choice([1, 2, 4], size=int(want.sum()))
        pgm[want] = nmodel[pnode[want]]
        bad = want & (rng.random(n_pods) < corrupt)
        pgm[bad] = 3 - pgm[bad]            # the other model
        stray = on & ~want & (rng.random(n_pods) < corrupt * 0.1)
        pg[stray], pgm[stray] = 1, 1       # gpu pods on nodes without (matching) gpus
    pd = np.full(n_pods, -1.0)
    pdt = np.zeros(n_pods, np.uint32)
    if disk and n_nodes:
        on = pnode < n_nodes
        want = on & (rng.random(n_pods) < 0.4)
        pd[want] = rng.choice([10000.0, 50.25, 1000.1], size=int(want.sum()))
        pdt[want] = ndtype[pnode[want]]
        pdt[want & (pdt == 0)] = 1
        bad = want & (rng.random(n_pods) < corrupt)
        pdt[bad] = 3 - pdt[bad]
    pf = np.zeros(n_pods, np.uint8)
    pf[rng.random(n_pods) < 0.05] |= A.POD_SYNTHETIC
    noreq = (rng.random(n_pods) < 0.03) & (pgm == 0)
    pf[noreq] |= A.POD_NO_REQUESTS
    pods = A.Pods(node=pnode, cpus=pc, mem=pm, gpus=pg, gpu_model=pgm, disk=pd, disk_type=pdt, flags=pf)
    params = A.offer_params(clobber_synthetic_pods=bool(seed & 1), filter_out_unsound_gpu_nodes=bool(seed & 2),
                            max_pods_per_node=max_pods, n_gpu_models=n_models, n_disk_types=n_types)
    return nodes, pods, params

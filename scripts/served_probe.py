#!/usr/bin/env python3
"""A/B probe of the served walkers on one MI355X: the 8-pool benchmark cluster staged once per library variant, the cycle timed under
a list of environment settings (read by the library at every match).  usage: served_probe.py lib1.so,lib2.so 'K=V K2=V2;K=V;...' [pools]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from cook_amd import _abi as A, workload, sharding
from cook_amd.engine import Engine

libs = sys.argv[1].split(",")
settings = [s for s in (sys.argv[2] if len(sys.argv) > 2 else "").split(";")]
n_pools = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ge = float(os.environ.get("PROBE_GE", "1.0"))
spec = workload.ClusterSpec()
pools = workload.make_pools(spec, range(n_pools))
params = A.default_params(good_enough_fitness=ge)
K = int(os.environ.get("PROBE_K", spec.per_pool[0]))
ref = None
for lib in libs:
    path = os.path.join(ROOT, "cook_amd", lib) if lib != "default" else None
    engines = {p: Engine(params, device=0, lib_path=path) for p in pools}
    for p, e in engines.items():
        e.cycle_stage(pools[p].tasks, pools[p].users, pools[p].pending_jobs, pools[p].offers, pools[p].groups)
    for st in settings:
        saved = {}
        for kv in st.split():
            k, v = kv.split("=")
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        cl = sharding.ShardedCluster(engines, workload.quota_groups(spec))
        cl.cycle(K)
        torch.cuda.synchronize()
        ts, bad = [], []
        saved_logs = False
        n_cycles = int(os.environ.get("PROBE_CYCLES", "4"))
        for cyc in range(n_cycles):
            a = time.perf_counter()
            cl.cycle(K)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - a) * 1e3)
            out = [engines[p].cycle_fetch()[1] for p in pools]
            rl = os.environ.get("COOK_ROUND_LOG")
            if rl and ref is None and cyc == 0:  # the reference run's round logs
                import glob, shutil
                for f in glob.glob(rl[:-1] + ".*"):
                    shutil.copy(f, f.replace("rlog.", "ref_rlog."))
            if ref is not None:  # EVERY cycle against the reference run
                for p, (x, y) in enumerate(zip(out, ref)):
                    if not np.array_equal(x, y):
                        bad.append((cyc, p, int(np.nonzero(x != y)[0][0]), int((x != y).sum())))
            if rl and bad and not saved_logs:
                import glob, shutil
                saved_logs = True
                tag = st.replace(" ", "_").replace("=", "")
                for f in glob.glob(rl[:-1] + ".*"):
                    shutil.copy(f, f.replace("rlog.", f"bad_{tag}_c{cyc}_rlog."))
        if bad:
            print("  MISMATCH (cycle, pool, first differing job, differing jobs):", bad[:12], flush=True)
        if ref is None:
            ref = out
            from oracle import checks  # (the first run against the oracle: first and last pool; every later run against the first)
            for pc in sorted({0, n_pools - 1}):
                r, j2o, _ = engines[pc].cycle_fetch()
                checks.check_pool_against_oracle(params, pools[pc], cl.quota_inputs(pc, cl.last_pool_usage[pc], cl.last_group_usage), r, j2o, K, threads=8)
            print("first run == oracle (pools 0 and %d)" % (n_pools - 1), flush=True)
        same = all(np.array_equal(a, b) for a, b in zip(out, ref))
        s = engines[0].match_stats()
        print(json.dumps({"lib": lib, "env": st, "ms": round(sorted(ts)[1], 2), "ms_all": [round(t, 1) for t in ts], "same_as_first": same,
                          "rounds": s["rounds"], "seq_us": s["seq_us"], "setup_us": s["setup_us"], "mode": s["served_mode"], "iters": s["serve_iterations"],
                          "empty": s["serve_empty_iterations"], "windows": s["serve_pool_windows"], "latch_wait_us": s["serve_latch_wait_us"], "fell_back": s["served_fell_back"]}), flush=True)
        cl.close()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for e in engines.values():
        e.close()

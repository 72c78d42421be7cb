#!/usr/bin/env python3
"""Timing study of the evaluation (a -DCOOK_EVAL_TRACE build of the library, COOK_LIB): one C4 pool, the waves' phase stamps of the
evaluation of the given rounds (stderr).  usage: COOK_LIB=cook_amd/libcookmatch_evtrace.so python scripts/eval_trace.py 3 20 45 70"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in sys.argv[1:]:
    env = dict(os.environ, COOK_EVAL_TRACE_ROUND=r, COOK_MATCH_SERVED="0", PYTHONPATH=ROOT)
    code = ("from cook_amd import _abi as A, workload\nfrom cook_amd.engine import Engine\nimport torch\n"
            "spec = workload.ClusterSpec(); pool = workload.make_pool(spec, 0)\n"
            "e = Engine(A.default_params(good_enough_fitness=1.0), device=0)\n"
            "e.cycle_stage(pool.tasks, pool.users, pool.pending_jobs, pool.offers, pool.groups)\ne.cycle_run(10**9)\n")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("\n".join(l for l in p.stderr.splitlines() if "EVALTRACE" in l) or p.stderr[-500:], flush=True)

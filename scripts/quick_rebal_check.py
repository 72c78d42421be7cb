#!/usr/bin/env python3
"""Seconds-long GPU check of the rebalancer against the oracle on a few small cases (incl. the fuzz-found one)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cook_amd.engine import Engine  # noqa: E402
from tests import parity_cases as P  # noqa: E402

mk = lambda params: Engine(params)  # noqa: E731
for kw in (dict(seed=707730441, n_running=2, n_pending=29, n_users=9, n_hosts=23, fractional=True, gpus=True, spare_frac=1.0),
           dict(seed=51, n_running=4000, n_pending=64, n_users=40, n_hosts=300),
           dict(seed=58, n_running=30000, n_pending=16, n_users=4, n_hosts=900, fractional=True)):
    P.rebalance_parity(mk, P.make_rebalance_case(**kw))
    print("ok", kw["seed"], flush=True)

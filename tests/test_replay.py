"""Trace replay harness (cook_amd/replay.py): the reference simulator's cycle order over the engine.
CPU tests drive the SIMT-emulated build; `-m gpu` drives the real library.  Both must produce the oracle-driven trace row by row."""
import os

import pytest

from cook_amd import _abi as A
from cook_amd import replay
from tests import parity_cases as P

CONFIG = {"shares": [{"user": "default", "mem": 60000.0, "cpus": 600.0, "gpus": 1.0}], "cycle-step-ms": 30000,
          "scheduler-config": {"rebalancer-config": {"max-preemption": 10.0}, "fenzo-config": {"fenzo-max-jobs-considered": 200}}}
# a cluster that fills up + a rebalancer that runs often and preempts across users
TIGHT = {"shares": [{"user": "default", "mem": 20000.0, "cpus": 20.0, "gpus": 1.0}, {"user": "a", "mem": 2000.0, "cpus": 2.0}],
         "cycle-step-ms": 30000, "time-ms-between-rebalancing": 120000,
         "scheduler-config": {"rebalancer-config": {"max-preemption": 8.0, "min-dru-diff": 0.05, "safe-dru-threshold": 0.1},
                              "fenzo-config": {"fenzo-max-jobs-considered": 50}}}


@pytest.fixture(scope="module")
def emu_engine():
    from cook_amd.engine import Engine
    from tests.simt_emu import build_emu
    so = build_emu.build()
    return lambda params: Engine(params, lib_path=so)


def test_config_parsing_matches_the_example_edn():
    c = replay.config_from_edn_keys(CONFIG)  # simulator_files/example-config.edn
    assert c["cycle_step_ms"] == 30000 and c["max_considerable"] == 200 and c["max_preemption"] == 10.0
    assert c["default_share"] == {"mem": 60000.0, "cpus": 600.0, "gpus": 1.0}
    assert c["time_ms_between_rebalancing"] == 30 * 60 * 1000 and c["min_dru_diff"] == 0.5  # zz_simulator.clj:73-78, 371-375


def test_replay_oracle_is_deterministic_and_conserves_resources():
    trace, hosts = P.make_trace(1, 150, 5)
    a = replay.simulate(trace, hosts, CONFIG, P.OracleBackend())
    b = replay.simulate(trace, hosts, CONFIG, P.OracleBackend())
    assert a.rows() == b.rows() and a.cycles > 100
    assert (a.used_cpus <= a.host_cpus).all() and (a.used_mem <= a.host_mem).all() and (a.used_cpus >= 0).all()
    # the loop ends with the cycle after the last submission (zz_simulator.clj:535): every job was submitted
    assert len(a.jobs) == len(trace)
    # a job never runs twice at once and only restarts after a failure / preemption
    for j in a.jobs:
        for x, y in zip(j.instances, j.instances[1:]):
            assert x["end_ms"] is not None and x["end_ms"] <= y["start_ms"] and x["status"] == "failed"


def test_replay_parity_emulated(emu_engine, tmp_path):
    trace, hosts = P.make_trace(2, 120, 4)
    sim = P.replay_parity(emu_engine, trace, hosts, CONFIG)
    out = tmp_path / "out-trace.csv"
    sim.write_csv(str(out))
    head = out.read_text().splitlines()[0].split(",")
    assert head == replay.CSV_HEADERS  # the reference's dump-jobs-to-csv schema (zz_simulator.clj:235-246)


def test_replay_parity_emulated_with_preemption(emu_engine):
    trace, hosts = P.make_trace(3, 160, 3, span_ms=1_200_000)
    P.replay_parity(emu_engine, trace, hosts, TIGHT, min_preempted=1)


def test_replay_fixture_is_the_reference_checkout():
    """tests/golden/replay_example.json == what tests/golden/make_replay_golden.py derives from the reference's files now
    (this container only; the GPU box has no checkout and uses the committed fixture)"""
    base = "/root/reference/scheduler/simulator_files"
    if not os.path.exists(os.path.join(base, "example-out-trace.csv")):
        pytest.skip("reference checkout not present")
    import csv
    import json
    from tests import golden_util as G
    fx = json.load(open(os.path.join(G.GOLDEN, "replay_example.json")))
    trace = replay.load_trace(os.path.join(base, "example-trace.json"))
    assert [j["job/uuid"] for j in fx["trace"]] == [j["job/uuid"] for j in trace] and len(trace) == 119
    assert [j["submit-time-ms"] for j in fx["trace"]] == [j["submit-time-ms"] for j in trace]
    rows = list(csv.DictReader(open(os.path.join(base, "example-out-trace.csv"))))
    assert len(rows) == 115 and {r["job_id"] for r in rows} == set(fx["expect"])
    assert all(fx["expect"][r["job_id"]]["hostname"] == r["hostname"] and fx["expect"][r["job_id"]]["status"] == r["status"] for r in rows)


def test_replay_reproduces_the_recorded_reference_run():
    """every task row of the reference's recorded example-out-trace.csv (real Clojure scheduler + real Fenzo), oracle-driven"""
    assert P.check_replay_recorded(P.OracleBackend()) == 115


def test_offer_order_contract_against_the_recorded_run():
    """INTEGRATION.md 3 "Offer order": the engine (and the oracle) break equal-fitness ties towards the lowest offer INDEX; the
    reference's recorded run (real Fenzo, five identical hosts) breaks them towards the LAST hostname.  Rows of `cook_offers` in
    DESCENDING hostname order reproduce all 115 recorded rows (the test above); in ASCENDING order the replay places every task in
    the same cycle but on the mirror-image host — this test counts the rows that diverge, so that a binding that forgets the rule is
    caught by the recorded run and not in production."""
    import json
    from tests import golden_util as G
    g = json.load(open(os.path.join(G.GOLDEN, "replay_example.json")))
    names = sorted(h["hostname"] for h in g["hosts"])
    out = {}
    for order in ("descending", "ascending"):
        sim = replay.simulate(g["trace"], g["hosts"], g["config"], P.OracleBackend(), offer_order=order)
        rows = {r["job_id"]: r for r in sim.rows()}
        assert set(rows) == set(g["expect"])
        out[order] = rows
    diverging = [j for j, e in g["expect"].items() if out["ascending"][j]["hostname"] != e["hostname"]]
    assert not [j for j, e in g["expect"].items() if out["descending"][j]["hostname"] != e["hostname"]]
    assert len(diverging) > 57, len(diverging)  # far more than half of the 115 rows (the rest sit on the middle host of five, or are forced)
    for j in diverging:  # ... and every diverging row is the MIRROR host: host h <-> host 4 - h, same start
        assert names.index(out["ascending"][j]["hostname"]) == len(names) - 1 - names.index(g["expect"][j]["hostname"]), j
        assert out["ascending"][j]["start_time_ms"] == out["descending"][j]["start_time_ms"]


def test_replay_recorded_run_through_the_emulated_engine(emu_engine):
    with emu_engine(A.default_params()) as e:
        assert P.check_replay_recorded(replay.EngineBackend(e)) == 115  # all 243 cycles, all 115 recorded rows


@pytest.mark.gpu
def test_replay_recorded_run_gpu():
    """... and through libcookmatch.so on the MI355X: all 243 cycles, all 115 recorded rows"""
    from cook_amd import build
    from cook_amd.engine import Engine
    with Engine(A.default_params(), lib_path=build.build()) as e:
        assert P.check_replay_recorded(replay.EngineBackend(e)) == 115


@pytest.mark.gpu
def test_replay_parity_gpu():
    from cook_amd import build
    from cook_amd.engine import Engine
    so = build.build()
    mk = lambda params: Engine(params, lib_path=so)  # noqa: E731
    trace, hosts = P.make_trace(2, 400, 12)
    P.replay_parity(mk, trace, hosts, CONFIG)
    trace, hosts = P.make_trace(3, 500, 8, span_ms=1_800_000)
    P.replay_parity(mk, trace, hosts, TIGHT, min_preempted=1)

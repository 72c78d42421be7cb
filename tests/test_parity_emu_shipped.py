"""The emulated build compiled with the SHIPPED launch shapes (window 320 / 384 candidate slots / 768-thread resolve workgroup /
1024-slot re-scoring tiles ...): the day-to-day emulated suite (test_parity_emu.py) runs small shapes so that small inputs take many
rounds; this module runs a few parity cases through exactly the constants the GPU executes, sized so that the shapes matter (more
than one window, more candidate offers than slots, users of several tiles).  What only the GPU suite covers after that is the
target-specific code of cook_amd/csrc/gpu_prims.hpp (DPP / readlane reductions, scoped fences, the write-through stores)."""
import numpy as np
import pytest

from cook_amd import _abi as A
from cook_amd import synth
from cook_amd.engine import Engine
from tests import parity_cases as P


@pytest.fixture(scope="module")
def make_engine():
    from tests.simt_emu import build_emu
    so = build_emu.build(shipped_shapes=True)
    return lambda params: Engine(params, lib_path=so)


def test_the_library_reports_its_shapes(make_engine):
    with make_engine(A.default_params()) as e:
        assert "shipped launch shapes" in e.version


def test_match_spans_several_windows(make_engine, algo=0):
    pool = synth.make_pool(seed=31, n_pending=1800, n_running=100, n_users=40, n_offers=420, gpus=True, constraints=True)
    j2o = P.match_parity(make_engine, pool.pending_jobs, pool.offers, pool.groups, A.default_params(match_algo=algo), reserved=(3, 7))
    assert (j2o >= 0).sum() > 200 and (j2o < 0).sum() > 100


def test_match_more_candidate_offers_than_slots(make_engine):
    jobs, offers = P.pinned_jobs_case(seed=32, n_jobs=900, n_offers=700, cardinality=0)  # host-unique pins: one distinct offer per job
    P.match_parity(make_engine, jobs, offers, None, A.default_params())


def test_match_good_enough_and_ports(make_engine):
    jobs, offers, groups = P.xres_random_case(33, 1500, 200, groups=True, slots=2)
    P.match_parity(make_engine, jobs, offers, groups, A.default_params(good_enough_fitness=0.8))


def test_cycle_and_multi_pool(make_engine):
    pools = [synth.make_pool(seed=40 + i, n_pending=700, n_running=300, n_users=25, n_offers=150, constraints=(i % 2 == 0)) for i in range(3)]
    P.multi_pool_parity(make_engine, pools, A.default_params(match_algo=2), k=10 ** 9)


def test_rebalance_users_of_several_tiles(make_engine):
    got = P.rebalance_parity(make_engine, P.make_rebalance_case(seed=34, n_running=6000, n_pending=24, n_users=3, n_hosts=400, fractional=True))
    assert len(got["decisions"]) > 0
    P.rebalance_parity(make_engine, P.make_rebalance_case(seed=35, n_running=3000, n_pending=40, n_users=30, n_hosts=20, constraints=True, gpus=True))


def test_rank_tie_tiles_of_several_items_per_thread(make_engine, monkeypatch):
    """the tie rule's tiles at their shipped size (1 024 threads, up to 8 192 items: a thread moves up to eight items of a tile, read
    before the first is written — tile_sort.hpp): tie-heavy pools whose groups spill over the nominal stretch, and the same pools with the
    tie rule as radix passes"""
    for seed, nu in ((51, 300), (52, 40)):
        pool = synth.make_pool(seed=seed, n_pending=9000, n_running=3000, n_users=nu, n_offers=16, tie_heavy=True)
        a = P.rank_parity(make_engine, pool, A.default_params(max_over_quota_jobs=10))
        monkeypatch.setenv("COOK_RANK_RADIX", "1")
        b = P.rank_parity(make_engine, pool, A.default_params(max_over_quota_jobs=10))
        monkeypatch.delenv("COOK_RANK_RADIX")
        assert np.array_equal(a, b)

"""ctypes binding of libcookmatch.so (include/cookmatch.h).

The product path is the HIP library and nothing else: if the shared object is missing or no MI355X is visible,
construction raises — there is NO CPU fallback (the CPU oracle under oracle/ is test infrastructure and is never
imported from this package).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import _abi as A
from ._protos import PROTOS

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.environ.get("COOK_LIB") or os.path.join(_HERE, "libcookmatch.so")  # COOK_LIB: a tuning variant of the same library

EXPORTS = list(PROTOS)  # every function include/cookmatch.h declares (cook_amd/_protos.py is generated from the header)


class CookError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"cookmatch error {code}: {msg}")
        self.code = code


_LIBS = {}


def load_library(path: Optional[str] = None):
    path = os.path.abspath(path or DEFAULT_LIB)
    if path in _LIBS:
        return _LIBS[path]
    # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64 (same SONAME as /opt/rocm's).  Whichever HIP
    # runtime is loaded FIRST serves the whole process, and mixing the two fails at hsa_init.  Every caller of this
    # package (bench.py, smoke(), the tests) also uses torch for device plumbing, so load torch's runtime first.
    # One HIP stream per pool: ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share
    # a queue serialise.  A rank that drives 8 pools wants 8 queues; the setting is read when the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # Kernel arguments in device memory: the launch-latency setting of this ROCm build (a cycle is ~500 dependent launches per chain;
    # with it switched off the eight-pool cycle measured 63.5 ms against 58.9).  Already the default here; pinned for hosts where it is not.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build the HIP extension first (python -m cook_amd.build). "
            "cook_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (restype, argtypes) in PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = restype
        fn.argtypes = argtypes   # pointers are c_void_p: ctypes checks the argument COUNT and pointer-vs-scalar for every call
    # the struct layouts of cook_amd._abi mirror include/cookmatch.h at COOK_ABI_VERSION: a library of another layout is refused before
    # the first call passes it a struct
    if lib.cook_abi_version() != A.ABI_VERSION:
        raise CookError(-1,
                        f"{path} has ABI version {lib.cook_abi_version()}, this binding was written for {A.ABI_VERSION}")
    _LIBS[path] = lib
    return lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Engine:
    """One engine per pool (one HIP stream); not re-entrant (cookmatch.h conventions)."""

    def __init__(self, params: Optional[A.CookParams] = None, device: int = 0, lib_path: Optional[str] = None):
        self._lib = load_library(lib_path)
        self.params = params or A.default_params()
        h = C.c_void_p()
        rc = self._lib.cook_engine_create(C.byref(self.params), int(device), C.byref(h))
        if rc != 0 or not h:
            raise CookError(rc, "cook_engine_create failed (no visible MI355X / HIP runtime error); "
                                "cook_amd has no CPU fallback")
        self._h = h
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cook_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def version(self) -> str:
        return self._lib.cook_version().decode()

    def _chk(self, rc):
        if rc != 0:
            raise CookError(rc, self._lib.cook_last_error(self._h).decode())

    def set_params(self, params: A.CookParams):
        self.params = params
        self._chk(self._lib.cook_engine_set_params(self._h, C.byref(params)))

    # ---- rank --------------------------------------------------------------------------------------------
    def rank_stage(self, tasks: A.Tasks, users: A.Users):
        ts, us = tasks.as_struct(), users.as_struct()
        self._rank_n, self._rank_np = tasks.n, int(tasks.pending.sum()) if tasks.n else 0
        self._chk(self._lib.cook_rank_stage(self._h, C.byref(ts), C.byref(us)))

    def rank_set_quota(self, quota: Optional[A.CookPoolQuota]):
        self._chk(self._lib.cook_rank_set_quota(self._h, C.byref(quota) if quota is not None else None))

    def rank_pool_usage(self) -> A.CookUsage:
        u = A.CookUsage()
        self._chk(self._lib.cook_rank_pool_usage(self._h, C.byref(u)))
        return u

    def rank_user_usage(self, n_users: int, device_ptr: Optional[int] = None):
        """[U, 3] = {cpus, mem, gpus} of every user's running tasks in this pool (cook_rank_user_usage).  With `device_ptr` (the
        address of a device buffer of U x 3 doubles, e.g. a torch tensor's data_ptr()) nothing comes back to the host."""
        if device_ptr is not None:
            self._chk(self._lib.cook_rank_user_usage(self._h, C.c_void_p(device_ptr), 1))
            return None
        out = np.zeros((max(1, n_users), 3), dtype=np.float64)
        self._chk(self._lib.cook_rank_user_usage(self._h, _p(out, C.c_double), 0))
        return out[:n_users]

    def rank_run(self):
        self._chk(self._lib.cook_rank_run(self._h))

    def rank_fetch(self, want_dru: bool = True):
        out = np.zeros(max(1, self._rank_np), dtype=np.uint32)
        dru = np.zeros(max(1, self._rank_n), dtype=np.float64) if want_dru else None
        n = C.c_uint32(0)
        self._chk(self._lib.cook_rank_fetch(self._h, _p(out, C.c_uint32), C.byref(n),
                                            _p(dru, C.c_double) if want_dru else None))
        return out[: n.value].copy(), (dru[: self._rank_n].copy() if want_dru else None)

    def rank(self, tasks: A.Tasks, users: A.Users, quota: Optional[A.CookPoolQuota] = None, want_dru: bool = True):
        """sort-jobs-by-dru-helper + filter-based-on-quota + filter-offensive-jobs -> (ranked task idx, dru per task)."""
        self.rank_stage(tasks, users)
        self.rank_set_quota(quota)
        self.rank_run()
        return self.rank_fetch(want_dru)

    # ---- match -------------------------------------------------------------------------------------------
    def match_stage(self, jobs: A.Jobs, offers: A.Offers, groups: Optional[A.Groups] = None,
                    reserved_hosts: Sequence[int] = ()):
        js, os_ = jobs.as_struct(), offers.as_struct()
        gs = groups.as_struct() if groups is not None else None
        res = np.array(list(reserved_hosts) or [0], dtype=np.uint32)
        self._match_k = jobs.n
        self._chk(self._lib.cook_match_stage(self._h, C.byref(js), C.byref(os_), C.byref(gs) if gs is not None else None,
                                             _p(res, C.c_uint32), len(reserved_hosts)))

    def match_run(self):
        self._chk(self._lib.cook_match_run(self._h))

    def match_count(self) -> int:
        """jobs of the engine's last match = the length cook_match_fetch / cook_cycle_fetch write"""
        n = C.c_uint32(0)
        self._chk(self._lib.cook_match_count(self._h, C.byref(n)))
        return n.value

    def match_fetch(self, k: Optional[int] = None):
        n = self.match_count()  # the engine's own count sizes the buffers (a cycle takes at most num_considerable jobs)
        k = n if k is None else min(k, n)
        j2o = np.full(max(1, n), -1, dtype=np.int32)
        fail = np.zeros(max(1, n), dtype=np.uint32)
        head = C.c_uint8(0)
        self._chk(self._lib.cook_match_fetch(self._h, _p(j2o, C.c_int32), _p(fail, C.c_uint32), C.byref(head)))
        return j2o[:k].copy(), fail[:k].copy(), bool(head.value)

    def match(self, jobs: A.Jobs, offers: A.Offers, groups: Optional[A.Groups] = None, reserved_hosts: Sequence[int] = ()):
        """Body of match-offer-to-schedule: -> (job_to_offer, fail_code, head_matched)."""
        self.match_stage(jobs, offers, groups, reserved_hosts)
        self.match_run()
        return self.match_fetch()

    # ---- rank + match without a host round trip --------------------------------------------------------------
    def cycle_stage(self, tasks: A.Tasks, users: A.Users, pending_jobs: A.Jobs, offers: A.Offers,
                    groups: Optional[A.Groups] = None, reserved_hosts: Sequence[int] = ()):
        ts, us, js, os_ = tasks.as_struct(), users.as_struct(), pending_jobs.as_struct(), offers.as_struct()
        gs = groups.as_struct() if groups is not None else None
        res = np.array(list(reserved_hosts) or [0], dtype=np.uint32)
        self._rank_n, self._rank_np = tasks.n, int(tasks.pending.sum()) if tasks.n else 0
        self._chk(self._lib.cook_cycle_stage(self._h, C.byref(ts), C.byref(us), C.byref(js), C.byref(os_),
                                             C.byref(gs) if gs is not None else None, _p(res, C.c_uint32),
                                             len(reserved_hosts)))

    def cycle_update(self, remove_task=(), add_tasks: Optional[A.Tasks] = None, add_pending: Optional[A.Jobs] = None,
                     offers: Optional[A.Offers] = None):
        """What changed since the last cycle (cook_cycle_update): task rows to remove (indices into the CURRENT arrays), task /
        pending-job rows to append, optionally fresh offers.  The resident columns are edited on the device."""
        rem = np.ascontiguousarray(np.asarray(list(remove_task) if not isinstance(remove_task, np.ndarray) else remove_task, dtype=np.uint32))
        ts = add_tasks.as_struct() if add_tasks is not None else None
        js = add_pending.as_struct() if add_pending is not None else None
        os_ = offers.as_struct() if offers is not None else None
        d = A.CookCycleDelta(len(rem), _p(rem, C.c_uint32) if len(rem) else None, C.pointer(ts) if ts is not None else None,
                             C.pointer(js) if js is not None else None, C.pointer(os_) if os_ is not None else None)
        self._chk(self._lib.cook_cycle_update(self._h, C.byref(d)))
        n_add = add_tasks.n if add_tasks is not None else 0
        p_add = int(add_tasks.pending.sum()) if n_add else 0
        # the mirror's sizes for the fetch buffers: upper bounds (removed rows only shrink them)
        self._rank_n = self._rank_n + n_add
        self._rank_np = self._rank_np + p_add

    def cycle_run(self, num_considerable: int):
        self._chk(self._lib.cook_cycle_run(self._h, int(num_considerable)))

    def cycle_run_rank(self, num_considerable: int):
        """The rank / considerable / take-K part of cycle_run; the placement then runs in cycle_match_multi()."""
        self._chk(self._lib.cook_cycle_run_rank(self._h, int(num_considerable)))

    def cycle_fetch(self, out=None):
        """-> (ranked task indices, job_to_offer by rank position, head matched).  `out` = (u32 buffer, i32 buffer) to fetch into
        (e.g. page-locked arrays of a PinnedArena, each with room for every pending task): views of them are returned."""
        if out is None:
            ranked = np.zeros(max(1, self._rank_np), dtype=np.uint32)
            j2o = np.full(max(1, self._rank_np), -1, dtype=np.int32)
        else:
            ranked, j2o = out
            assert len(ranked) >= self._rank_np and len(j2o) >= self._rank_np
        n, k = C.c_uint32(0), C.c_uint32(0)
        head = C.c_uint8(0)
        self._chk(self._lib.cook_cycle_fetch(self._h, _p(ranked, C.c_uint32), C.byref(n), _p(j2o, C.c_int32),
                                             C.byref(k), C.byref(head)))
        if out is None:
            return ranked[: n.value].copy(), j2o[: k.value].copy(), bool(head.value)
        return ranked[: n.value], j2o[: k.value], bool(head.value)

    # ---- considerable jobs -----------------------------------------------------------------------------------
    def considerable(self, queue: A.Queue, users: A.UserState, num_considerable: int):
        """pending-jobs->considerable-jobs (scheduler.clj:729-762) -> (queue positions, rate_limited per user, passed per user)."""
        out = np.zeros(max(1, min(int(num_considerable), queue.n)), dtype=np.uint32)
        rl = np.zeros(max(1, users.n), dtype=np.uint32)
        ps = np.zeros(max(1, users.n), dtype=np.uint32)
        n = C.c_uint32(0)
        qs, us = queue.as_struct(), users.as_struct()
        self._chk(self._lib.cook_considerable(self._h, C.byref(qs), C.byref(us), int(num_considerable), _p(out, C.c_uint32),
                                              C.byref(n), _p(rl, C.c_uint32), _p(ps, C.c_uint32)))
        return out[: n.value].copy(), rl[: users.n].copy(), ps[: users.n].copy()

    def cycle_set_considerable(self, users: Optional[A.UserState], eligible_by_pending=None):
        el = np.ascontiguousarray(eligible_by_pending, dtype=np.uint8) if eligible_by_pending is not None else None
        us = users.as_struct() if users is not None else None
        self._chk(self._lib.cook_cycle_set_considerable(self._h, C.byref(us) if us is not None else None,
                                                        _p(el, C.c_uint8) if el is not None else None))

    def cycle_fetch_considerable(self):
        out = np.zeros(max(1, self._rank_np), dtype=np.uint32)
        n = C.c_uint32(0)
        self._chk(self._lib.cook_cycle_fetch_considerable(self._h, _p(out, C.c_uint32), C.byref(n)))
        return out[: n.value].copy()

    # ---- rebalancer ------------------------------------------------------------------------------------------
    def rebalance_stage(self, running: A.Tasks, pending: A.Jobs, pending_job_id, pending_priority, users: A.Users,
                        spare: A.HostSpare, rparams: A.CookRebalanceParams, host_attrs: Optional[A.Offers] = None,
                        groups: Optional[A.Groups] = None, attrs_cached=None):
        rs, ps, us, ss = running.as_struct(), pending.as_struct(), users.as_struct(), spare.as_struct()
        hs = host_attrs.as_struct() if host_attrs is not None else None
        gs = groups.as_struct() if groups is not None else None
        jid = np.ascontiguousarray(pending_job_id, dtype=np.int64)
        pri = np.ascontiguousarray(pending_priority, dtype=np.int32)
        ck = np.ascontiguousarray(attrs_cached, dtype=np.uint8) if attrs_cached is not None else None
        self._rb_p, self._rb_r = pending.n, running.n
        self._chk(self._lib.cook_rebalance_stage(
            self._h, C.byref(rs), _p(ck, C.c_uint8) if ck is not None else None, C.byref(ps), _p(jid, C.c_int64),
            _p(pri, C.c_int32), C.byref(us), C.byref(ss), C.byref(hs) if hs is not None else None,
            C.byref(gs) if gs is not None else None, C.byref(rparams)))

    def rebalance_run(self):
        self._chk(self._lib.cook_rebalance_run(self._h))

    def rebalance_fetch(self):
        P, R = self._rb_p, self._rb_r
        dec = (A.CookPreemption * max(1, P))()
        pre = np.zeros(max(1, R + P), dtype=np.uint32)
        pdru = np.zeros(max(1, P), dtype=np.float64)
        nd, npre = C.c_uint32(0), C.c_uint32(0)
        self._chk(self._lib.cook_rebalance_fetch(self._h, dec, C.byref(nd), _p(pre, C.c_uint32), C.byref(npre),
                                                 _p(pdru, C.c_double)))
        out = []
        for i in range(nd.value):
            d = dec[i]
            out.append(dict(pending_index=d.pending_index, host=d.host, dru=d.dru, cpus=d.cpus, mem=d.mem, gpus=d.gpus,
                            tasks=[int(x) for x in pre[d.task_off: d.task_off + d.task_n]]))
        return dict(decisions=out, pending_dru=pdru[:P].copy(), final=None)

    def rebalance(self, running, pending, pending_job_id, pending_priority, users, spare, rparams, host_attrs=None,
                  groups=None, attrs_cached=None):
        """init-state + the rebalance loop (rebalancer.clj:222-467) -> dict(decisions=[...], pending_dru=array)."""
        self.rebalance_stage(running, pending, pending_job_id, pending_priority, users, spare, rparams, host_attrs,
                             groups, attrs_cached)
        self.rebalance_run()
        return self.rebalance_fetch()

    def rebalance_timing(self) -> float:
        ms = C.c_double(0)
        self._lib.cook_rebalance_timing(self._h, C.byref(ms))
        return ms.value

    # ---- consumers of the placement's by-products ----------------------------------------------------------------
    def match_explain(self, job_pos) -> np.ndarray:
        """fenzo-utils/summarize-placement-failure (fenzo_utils.clj:33-55) for the given job positions of the LAST match:
        -> uint32 [n, WHY_SLOTS] host counts (A.why_summary turns a row into the reference's map)."""
        pos = np.ascontiguousarray(job_pos, dtype=np.uint32)
        out = np.zeros((max(1, len(pos)), A.WHY_SLOTS), dtype=np.uint32)
        self._chk(self._lib.cook_match_explain(self._h, _p(pos, C.c_uint32) if len(pos) else None, len(pos),
                                               _p(out.reshape(-1), C.c_uint32)))
        return out[: len(pos)].copy()

    def match_metrics(self, n_users: int = 0, n_gpu_models: int = 0) -> dict:
        """handle-match-cycle-metrics' numbers (scheduler.clj:1210-1280) for the LAST match."""
        m = A.CookCycleMetrics()
        uc = np.zeros(max(1, n_users), np.uint32)
        um = np.zeros(max(1, n_users), np.uint32)
        jg = np.zeros(n_gpu_models + 1, np.int64)
        og = np.zeros(n_gpu_models + 1, np.int64)
        self._chk(self._lib.cook_match_metrics(self._h, C.byref(m), _p(uc, C.c_uint32) if n_users else None,
                                               _p(um, C.c_uint32) if n_users else None, n_users, _p(jg, C.c_int64), _p(og, C.c_int64),
                                               n_gpu_models))
        return dict(considerable=m.considerable, matched=m.matched, unmatched=m.unmatched, offers=m.offers,
                    offers_scheduled=m.offers_scheduled, head_matched=bool(m.head_matched), jobs=m.jobs.as_dict(),
                    offers_stats=m.offer_stats.as_dict(), user_considerable=uc[:n_users].copy(), user_matched=um[:n_users].copy(),
                    job_gpus_by_model=jg, offer_gpus_by_model=og)

    # ---- offer construction from node state --------------------------------------------------------------------
    def offers_stage(self, nodes: A.Nodes, pods: A.Pods, oparams: A.CookOfferParams):
        ns, ps = nodes.as_struct(), pods.as_struct()
        self._of_n, self._of_attr, self._of_params = nodes.n, nodes.n_attr_keys, oparams
        self._chk(self._lib.cook_offers_stage(self._h, C.byref(ns), C.byref(ps), C.byref(oparams)))

    def offers_run(self):
        self._chk(self._lib.cook_offers_run(self._h))

    def offers_fetch(self) -> A.BuiltOffers:
        n, na, op = self._of_n, self._of_attr, self._of_params
        cap = max(1, n)
        gs, ds = max(1, int(op.gpu_slots)), max(1, int(op.disk_slots))
        tab = lambda k, dt: np.zeros(cap, dt) if k == 1 else np.zeros((cap, k), dt)  # noqa: E731
        cols = dict(node=np.zeros(cap, np.uint32), host=np.zeros(cap, np.uint32), cpus=np.zeros(cap), mem=np.zeros(cap),
                    gpu_model=tab(gs, np.uint32), gpu_count=tab(gs, np.float64), disk_type=tab(ds, np.uint32),
                    disk_space=tab(ds, np.float64), num_pods=np.zeros(cap, np.int32))
        attr = np.zeros((cap, na), np.uint32) if na else None
        o = A.CookNodeOffers(_p(cols["node"], C.c_uint32), _p(cols["host"], C.c_uint32), _p(cols["cpus"], C.c_double),
                             _p(cols["mem"], C.c_double), _p(cols["gpu_model"].reshape(-1), C.c_uint32),
                             _p(cols["gpu_count"].reshape(-1), C.c_double), _p(cols["disk_type"].reshape(-1), C.c_uint32),
                             _p(cols["disk_space"].reshape(-1), C.c_double), _p(cols["num_pods"], C.c_int32),
                             _p(attr.reshape(-1), C.c_uint32) if na else None)
        status = np.zeros(cap, np.uint8)
        tot = A.CookOfferTotals()
        gcap, gcons = np.zeros(op.n_gpu_models + 1, np.int64), np.zeros(op.n_gpu_models + 1, np.int64)
        dcap, dcons = np.zeros(op.n_disk_types + 1), np.zeros(op.n_disk_types + 1)
        r = C.c_uint32(0)
        self._chk(self._lib.cook_offers_fetch(self._h, C.byref(o), C.byref(r), _p(status, C.c_uint8), C.byref(tot),
                                              _p(gcap, C.c_int64), _p(gcons, C.c_int64), _p(dcap, C.c_double), _p(dcons, C.c_double)))
        k = r.value
        totals = {f: getattr(tot, f) for f, _ in A.CookOfferTotals._fields_}
        return A.BuiltOffers(attr=attr[:k].copy() if na else None, node_status=status[:n].copy(), totals=totals,
                             gpu_capacity_by_model=gcap, gpu_consumed_by_model=gcons, disk_capacity_by_type=dcap,
                             disk_consumed_by_type=dcons, **{c: v[:k].copy() for c, v in cols.items()})

    def offers_build(self, nodes: A.Nodes, pods: A.Pods, oparams: A.CookOfferParams) -> A.BuiltOffers:
        """generate-offers' numeric core (kubernetes/compute_cluster.clj:68-190): available = capacity - consumption per
        node, the schedulable filter, the offer rows in node order and the capacity / consumption gauges."""
        self.offers_stage(nodes, pods, oparams)
        self.offers_run()
        return self.offers_fetch()

    def match_stage_built_offers(self, jobs: A.Jobs, groups: Optional[A.Groups] = None, reserved_hosts: Sequence[int] = (),
                                 with_task_limits: bool = False):
        """cook_match_stage with the rows of the last offers_run as offers, in place on the device (no host round trip)."""
        js = jobs.as_struct()
        gs = groups.as_struct() if groups is not None else None
        res = np.array(list(reserved_hosts) or [0], dtype=np.uint32)
        self._match_k = jobs.n
        self._chk(self._lib.cook_match_stage_built_offers(self._h, C.byref(js), C.byref(gs) if gs is not None else None,
                                                          _p(res, C.c_uint32), len(reserved_hosts), int(bool(with_task_limits))))

    def cycle_stage_built_offers(self, tasks: A.Tasks, users: A.Users, pending_jobs: A.Jobs, groups: Optional[A.Groups] = None,
                                 reserved_hosts: Sequence[int] = (), with_task_limits: bool = False):
        """cook_cycle_stage with the rows of the last offers_run as offers, in place on the device."""
        ts, us, js = tasks.as_struct(), users.as_struct(), pending_jobs.as_struct()
        gs = groups.as_struct() if groups is not None else None
        res = np.array(list(reserved_hosts) or [0], dtype=np.uint32)
        self._rank_n, self._rank_np = tasks.n, int(tasks.pending.sum()) if tasks.n else 0
        self._chk(self._lib.cook_cycle_stage_built_offers(self._h, C.byref(ts), C.byref(us), C.byref(js),
                                                          C.byref(gs) if gs is not None else None, _p(res, C.c_uint32),
                                                          len(reserved_hosts), int(bool(with_task_limits))))

    def offers_timing(self) -> float:
        ms = C.c_double(0)
        self._lib.cook_offers_timing(self._h, C.byref(ms))
        return ms.value

    # ---- measurement -----------------------------------------------------------------------------------------
    def last_timing(self):
        r, m = C.c_double(0), C.c_double(0)
        self._lib.cook_last_timing(self._h, C.byref(r), C.byref(m))
        return r.value, m.value

    def match_stats(self):
        out = (C.c_uint32 * 64)()
        n = self._lib.cook_match_stats_ex(self._h, out, 64)
        keys = ("rounds", "matched", "stop_list", "stop_full", "stop_group", "stop_window", "segments", "resolved", "setup_us", "seq_us", "touched", "visited",
                "_12", "_13", "_14", "_15", "trunc_lists", "served_mode", "served_pools", "serve_iterations", "serve_empty_iterations",
                "serve_pool_windows", "serve_latch_wait_us", "served_fell_back", "serve_streams", "guard_hits", "update_us", "update_sync_us", "update_allocs", "update_slowest_phase", "update_slowest_phase_us", "_31",
                "rank_batch_pools", "rank_batch_launches", "rank_batch_grouped_launches", "rank_batch_single_ops", "rank_batch_syncs",
                "placement_form", "classfit_refused", "_39", "cf_walked", "cf_matched", "cf_overlay_wins", "cf_opened", "cf_opened_full", "cf_gpu_places", "cf_epochs",
                "cf_scans", "cf_exact_turns", "cf_retightened", "_50", "cf_batches", "cf_dead_lanes", "cf_ticks", "cf_ticks_prologue", "cf_ticks_epochs", "cf_ticks_books",
                "cf_spins", "cf_ticks_walk", "cf_ticks_phase1", "cf_rewinds", "cf_flips", "cf_hwid_decider", "cf_hwid_books")
        return {k: int(x) for k, x in zip(keys, out[:max(0, n)]) if not k.startswith("_")}

    def set_profiling(self, on: bool):
        self._lib.cook_set_profiling(self._h, int(bool(on)))

    def kernel_timings(self):
        cap = 128
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        launches = (C.c_uint32 * cap)()
        n = self._lib.cook_kernel_timings(self._h, names, ms, launches, cap)
        return {names[i].decode(): (ms[i], launches[i]) for i in range(max(0, n))}


class PinnedArena:
    """numpy arrays in page-locked host memory (cook_host_alloc): copies to and from the device run at link speed.  The
    arena owns the memory; arrays made by it must not outlive it."""

    def __init__(self, lib_path: Optional[str] = None):
        self._lib = load_library(lib_path)
        self._blocks = []

    def empty(self, shape, dtype) -> np.ndarray:
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
        nbytes = max(1, n * dt.itemsize)
        p = self._lib.cook_host_alloc(nbytes)
        if not p:
            raise MemoryError("cook_host_alloc failed")
        self._blocks.append(p)
        buf = (C.c_char * nbytes).from_address(p)
        return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

    def copy(self, a: np.ndarray) -> np.ndarray:
        out = self.empty(a.shape, a.dtype)
        out[...] = a
        return out

    def pin(self, obj):
        """A copy of a dataclass of columns (A.Tasks, A.Jobs, A.Offers, ...) whose numpy arrays live in this arena."""
        import copy as _copy
        import dataclasses
        new = _copy.copy(obj)
        for f in dataclasses.fields(obj):
            v = getattr(obj, f.name)
            if isinstance(v, np.ndarray):
                object.__setattr__(new, f.name, self.copy(np.ascontiguousarray(v)))
        return new

    def close(self):
        for p in self._blocks:
            self._lib.cook_host_free(p)
        self._blocks = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def rank_pool_usage_multi(engines: Sequence[Engine]):
    """rank_pool_usage of several engines of one device in ONE call (cook_rank_pool_usage_multi: the pools' sums as pool batches, one stream
    synchronisation) -> a list of (count, cpus, mem, gpus) tuples, engines' order."""
    if not engines:
        return []
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    out = (A.CookUsage * len(engines))()
    rc = engines[0]._lib.cook_rank_pool_usage_multi(arr, len(engines), out)
    if rc != 0:
        for e in engines:
            if e._lib.cook_last_error(e._h):
                e._chk(rc)
        engines[0]._chk(rc)
    return [u.as_tuple() for u in out]


def cycle_run_rank_multi(engines: Sequence[Engine], num_considerable, user_usage_ptrs: Optional[Sequence[int]] = None,
                         n_users: int = 0):
    """cycle_run_rank of several engines (pools of one rank, same device) in ONE call: the pools' rank flows side by side on one stream,
    the same kernel of several pools in one launch (cook_cycle_run_rank_multi; same results as the calls one by one).
    num_considerable: one K for all, or one per engine.  user_usage_ptrs: device addresses of one [U, 3] float64 buffer per engine -> rank_user_usage(device_ptr=...) of each, in the same
    call; n_users > 0 without pointers: the usage comes back as a list of [U, 3] host arrays."""
    if not engines:
        return None
    lib = engines[0]._lib
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    ks = [int(num_considerable)] * len(engines) if np.isscalar(num_considerable) else [int(k) for k in num_considerable]
    assert len(ks) == len(engines)
    ks = (C.c_uint32 * len(engines))(*[min(k, 0xFFFFFFFF) for k in ks])
    outs = None
    if user_usage_ptrs is not None:
        uu = (C.c_void_p * len(engines))(*[C.c_void_p(int(p)) for p in user_usage_ptrs])
        rc = lib.cook_cycle_run_rank_multi(arr, len(engines), ks, uu, 1)
    elif n_users:
        outs = [np.zeros((max(1, n_users), 3), dtype=np.float64) for _ in engines]
        uu = (C.c_void_p * len(engines))(*[o.ctypes.data for o in outs])
        rc = lib.cook_cycle_run_rank_multi(arr, len(engines), ks, uu, 0)
    else:
        rc = lib.cook_cycle_run_rank_multi(arr, len(engines), ks, None, 0)
    if rc != 0:
        for e in engines:  # the message is with the engine whose flow failed
            if e._lib.cook_last_error(e._h):
                e._chk(rc)
        engines[0]._chk(rc)
    return [o[:n_users] for o in outs] if outs is not None else None


def cycle_match_multi(engines: Sequence[Engine]):
    """The placements of several engines (pools of one rank, same device) in lockstep rounds: one sequence of launches with
    blockIdx.z = pool instead of one stream of small kernels per pool (cook_cycle_match_multi)."""
    if not engines:
        return
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    lead = engines[0]
    lead._chk(lead._lib.cook_cycle_match_multi(arr, len(engines)))

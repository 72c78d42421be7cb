"""ctypes mirror of include/cookmatch.h (struct layouts and SoA containers).

The containers hold numpy arrays (host SoA) and build the C structs on demand; they keep the arrays alive for the
duration of a call.  Field names follow the reference's domain: tasks, users, shares (divisors), quotas, offers,
leases, groups (see include/cookmatch.h for the reference file:line each field comes from).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

NONE_U32 = 0xFFFFFFFF
ABI_VERSION = 4  # COOK_ABI_VERSION of include/cookmatch.h whose struct layouts this module mirrors (tests/test_abi.py compares)
DMAX = float(np.finfo(np.float64).max)

_f64p = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)


class CookParams(C.Structure):
    _fields_ = [
        ("dru_mode", C.c_int32),
        ("max_over_quota_jobs", C.c_int32),
        ("offensive_max_mem_mb", C.c_double),
        ("offensive_max_cpus", C.c_double),
        ("good_enough_fitness", C.c_double),
        ("host_lifetime_mins", C.c_int64),
        ("match_algo", C.c_int32),
        ("reserved", C.c_int32),
    ]


class CookUsage(C.Structure):
    _fields_ = [("count", C.c_double), ("cpus", C.c_double), ("mem", C.c_double), ("gpus", C.c_double)]

    def as_tuple(self):
        return (self.count, self.cpus, self.mem, self.gpus)


class CookTasks(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("cpus", _f64p), ("mem", _f64p), ("gpus", _f64p),
        ("user", _u32p), ("priority", _i32p),
        ("start_ms", _i64p), ("task_id", _i64p), ("job_id", _i64p),
        ("pending", _u8p), ("host", _u32p),
    ]


class CookUsers(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("div_cpus", _f64p), ("div_mem", _f64p), ("div_gpus", _f64p),
        ("quota_count", _f64p), ("quota_cpus", _f64p), ("quota_mem", _f64p), ("quota_gpus", _f64p),
    ]


class CookPoolQuota(C.Structure):
    _fields_ = [
        ("has_pool_quota", C.c_int32), ("has_group_quota", C.c_int32),
        ("pool_quota", CookUsage), ("group_quota", CookUsage), ("group_usage", CookUsage),
        ("pool_usage_given", C.c_int32), ("reserved", C.c_int32),
        ("pool_usage", CookUsage),
    ]


class CookQueue(C.Structure):
    _fields_ = [("n", C.c_uint32), ("cpus", _f64p), ("mem", _f64p), ("gpus", _f64p), ("user", _u32p), ("eligible", _u8p)]


class CookUserState(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("quota_count", _f64p), ("quota_cpus", _f64p), ("quota_mem", _f64p), ("quota_gpus", _f64p),
        ("usage_count", _f64p), ("usage_cpus", _f64p), ("usage_mem", _f64p), ("usage_gpus", _f64p),
        ("tokens_left", _i64p), ("enforce_rate_limit", C.c_int32), ("has_pool_quota", C.c_int32),
        ("pool_quota", CookUsage), ("pool_usage_given", C.c_int32), ("reserved", C.c_int32), ("pool_usage", CookUsage),
    ]


class CookJobs(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("cpus", _f64p), ("mem", _f64p), ("gpus", _f64p),
        ("gpu_model", _u32p), ("user", _u32p), ("group", _u32p),
        ("eq_off", _u32p), ("eq_key", _u32p), ("eq_val", _u32p),
        ("novel_off", _u32p), ("novel_host", _u32p),
        ("reserved_host", _i32p), ("ckpt_location", _u32p), ("est_end_ms", _i64p),
        ("disk_request", _f64p), ("disk_type", _u32p),
        ("ports", _i32p), ("n_scalars", C.c_uint32), ("reserved_", C.c_uint32), ("scalars", _f64p),
    ]


class CookOffers(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("cpus", _f64p), ("mem", _f64p), ("host", _u32p), ("k8s", _u8p),
        ("gpu_model", _u32p), ("gpu_count", _f64p),
        ("disk_type", _u32p), ("disk_space", _f64p),
        ("n_attr_keys", C.c_uint32), ("attr", _u32p),
        ("max_tasks", _i32p), ("num_tasks", _i32p),
        ("location", _u32p), ("host_start_s", _i64p),
        ("run_cpus", _f64p), ("run_mem", _f64p), ("run_count", _i32p),
        ("gpu_slots", C.c_uint32), ("disk_slots", C.c_uint32), ("ports", _i32p),
        ("n_scalars", C.c_uint32), ("reserved_", C.c_uint32), ("scalars", _f64p),
    ]


class CookCycleDelta(C.Structure):  # cook_cycle_update
    _fields_ = [("n_remove", C.c_uint32), ("remove_task", C.POINTER(C.c_uint32)), ("add_tasks", C.POINTER(CookTasks)),
                ("add_pending", C.POINTER(CookJobs)), ("offers", C.POINTER(CookOffers))]


class CookGroups(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("type", _u8p), ("attr_key", _u32p), ("minimum", _i32p),
        ("run_off", _u32p), ("run_host", _u32p), ("run_attr", _u32p),
    ]


class CookRebalanceParams(C.Structure):
    _fields_ = [("safe_dru_threshold", C.c_double), ("min_dru_diff", C.c_double),
                ("max_preemption", C.c_int32), ("reserved", C.c_int32)]


class CookHostSpare(C.Structure):
    _fields_ = [("n", C.c_uint32), ("host", _u32p), ("cpus", _f64p), ("mem", _f64p), ("gpus", _f64p)]


class CookPreemption(C.Structure):
    _fields_ = [
        ("pending_index", C.c_uint32), ("host", C.c_uint32),
        ("dru", C.c_double), ("cpus", C.c_double), ("mem", C.c_double), ("gpus", C.c_double),
        ("task_off", C.c_uint32), ("task_n", C.c_uint32),
    ]


MAX_SCALARS, MAX_RES_SLOTS = 3, 4  # COOK_MAX_SCALARS, COOK_MAX_RES_SLOTS
_DT = {_f64p: np.float64, _u32p: np.uint32, _i32p: np.int32, _i64p: np.int64, _u8p: np.uint8}


def _ptr(arr: Optional[np.ndarray], ptype):
    if arr is None:
        return ptype()
    assert arr.dtype == _DT[ptype] and arr.flags["C_CONTIGUOUS"], (arr.dtype, ptype)
    return arr.ctypes.data_as(ptype)


def _memo_struct(obj, build):
    """The ctypes view of a column set, rebuilt only when one of its arrays was replaced (a struct is ~20 pointer conversions, 55 us of
    interpreter time under the GIL; a rank drives eight pools per cycle).  In-place edits of the arrays keep the pointers valid; the
    memo holds the arrays, so an address cannot be reused while it is cached."""
    if getattr(obj, "scalars", None) is not None:  # (passed as a transposed COPY: an edit in place would not reach a cached one)
        return build()
    arrays = [v for k, v in vars(obj).items() if isinstance(v, np.ndarray) and not k.startswith("_")]
    # the key: the object itself (a copy.copy() carries the memo over, its struct must not), its arrays, and the plain fields the
    # builders bake into the struct (n_attr_keys, gpu_slots, disk_slots, n_scalars, ...)
    plain = tuple((k, v) for k, v in vars(obj).items() if isinstance(v, (int, bool, np.integer)) and not k.startswith("_"))
    key = (id(obj), tuple(map(id, arrays)), plain)
    memo = obj.__dict__.get("_struct_memo")
    if memo is not None and memo[0] == key:
        return memo[1]
    st = build()
    obj.__dict__["_struct_memo"] = (key, st, arrays)
    return st


def _arr(x, dtype, n=None):
    if x is None:
        return None
    a = np.ascontiguousarray(x, dtype=dtype)
    if n is not None:
        assert a.shape == (n,), (a.shape, n)
    return a


def _table(x, dtype, n, max_cols):
    """[n] or [n, cols] -> ([n, cols] C-contiguous, cols)"""
    if x is None:
        return None, 0
    a = np.ascontiguousarray(x, dtype=dtype)
    if a.ndim == 1:
        a = a.reshape(n, 1)
    assert a.ndim == 2 and a.shape[0] == n and 1 <= a.shape[1] <= max_cols, (a.shape, n, max_cols)
    return a, a.shape[1]


def default_params(**kw) -> CookParams:
    """Reference defaults: config.clj:108-116 (fenzo), :413-416 (max-over-quota-jobs 100), :398-407 (task-constraints)."""
    p = CookParams(dru_mode=0, max_over_quota_jobs=100, offensive_max_mem_mb=float("inf"),
                   offensive_max_cpus=float("inf"), good_enough_fitness=0.8, host_lifetime_mins=0, match_algo=0,
                   reserved=0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


@dataclass
class Tasks:
    """running instances ++ synthetic tasks of pending jobs (tools.clj:582-588)."""
    cpus: np.ndarray
    mem: np.ndarray
    user: np.ndarray
    priority: np.ndarray
    start_ms: np.ndarray
    task_id: np.ndarray
    job_id: np.ndarray
    pending: np.ndarray
    gpus: Optional[np.ndarray] = None
    host: Optional[np.ndarray] = None

    def __post_init__(self):
        n = len(self.cpus)
        self.cpus = _arr(self.cpus, np.float64, n)
        self.mem = _arr(self.mem, np.float64, n)
        self.gpus = _arr(self.gpus, np.float64, n)
        self.user = _arr(self.user, np.uint32, n)
        self.priority = _arr(self.priority, np.int32, n)
        self.start_ms = _arr(self.start_ms, np.int64, n)
        self.task_id = _arr(self.task_id, np.int64, n)
        self.job_id = _arr(self.job_id, np.int64, n)
        self.pending = _arr(self.pending, np.uint8, n)
        self.host = _arr(self.host, np.uint32, n)

    @property
    def n(self):
        return len(self.cpus)

    def as_struct(self) -> CookTasks:
        return _memo_struct(self, self._build_struct)

    def _build_struct(self) -> CookTasks:
        return CookTasks(self.n, _ptr(self.cpus, _f64p), _ptr(self.mem, _f64p), _ptr(self.gpus, _f64p),
                         _ptr(self.user, _u32p), _ptr(self.priority, _i32p), _ptr(self.start_ms, _i64p),
                         _ptr(self.task_id, _i64p), _ptr(self.job_id, _i64p), _ptr(self.pending, _u8p),
                         _ptr(self.host, _u32p))


@dataclass
class Users:
    """DRU divisors = shares (share.clj:75-119) and quotas (quota.clj:272-295), indexed by user id (= name rank)."""
    div_cpus: np.ndarray
    div_mem: np.ndarray
    div_gpus: Optional[np.ndarray] = None
    quota_count: Optional[np.ndarray] = None
    quota_cpus: Optional[np.ndarray] = None
    quota_mem: Optional[np.ndarray] = None
    quota_gpus: Optional[np.ndarray] = None

    def __post_init__(self):
        n = len(self.div_cpus)
        full = lambda v: np.full(n, v, dtype=np.float64)  # noqa: E731
        self.div_cpus = _arr(self.div_cpus, np.float64, n)
        self.div_mem = _arr(self.div_mem, np.float64, n)
        self.div_gpus = _arr(self.div_gpus if self.div_gpus is not None else full(DMAX), np.float64, n)
        self.quota_count = _arr(self.quota_count if self.quota_count is not None else full(2.0 ** 31 - 1), np.float64, n)
        self.quota_cpus = _arr(self.quota_cpus if self.quota_cpus is not None else full(DMAX), np.float64, n)
        self.quota_mem = _arr(self.quota_mem if self.quota_mem is not None else full(DMAX), np.float64, n)
        self.quota_gpus = _arr(self.quota_gpus if self.quota_gpus is not None else full(DMAX), np.float64, n)

    @property
    def n(self):
        return len(self.div_cpus)

    def as_struct(self) -> CookUsers:
        return CookUsers(self.n, _ptr(self.div_cpus, _f64p), _ptr(self.div_mem, _f64p), _ptr(self.div_gpus, _f64p),
                         _ptr(self.quota_count, _f64p), _ptr(self.quota_cpus, _f64p), _ptr(self.quota_mem, _f64p),
                         _ptr(self.quota_gpus, _f64p))


@dataclass
class Queue:
    """The pool's pending jobs in rank order (input of pending-jobs->considerable-jobs, scheduler.clj:729-762)."""
    cpus: np.ndarray
    mem: np.ndarray
    user: np.ndarray
    gpus: Optional[np.ndarray] = None
    eligible: Optional[np.ndarray] = None

    def __post_init__(self):
        n = len(self.cpus)
        self.cpus = _arr(self.cpus, np.float64, n)
        self.mem = _arr(self.mem, np.float64, n)
        self.user = _arr(self.user, np.uint32, n)
        self.gpus = _arr(self.gpus, np.float64, n)
        self.eligible = _arr(self.eligible, np.uint8, n)

    @property
    def n(self):
        return len(self.cpus)

    def as_struct(self) -> CookQueue:
        return CookQueue(self.n, _ptr(self.cpus, _f64p), _ptr(self.mem, _f64p), _ptr(self.gpus, _f64p), _ptr(self.user, _u32p),
                         _ptr(self.eligible, _u8p))


@dataclass
class UserState:
    """user->quota, user->usage, launch-rate tokens and the pool quota (tools.clj:903-973), indexed by user id."""
    quota_count: np.ndarray
    quota_cpus: np.ndarray
    quota_mem: np.ndarray
    quota_gpus: np.ndarray
    usage_count: np.ndarray
    usage_cpus: np.ndarray
    usage_mem: np.ndarray
    usage_gpus: np.ndarray
    tokens_left: Optional[np.ndarray] = None
    enforce_rate_limit: bool = False
    pool_quota: Optional[CookUsage] = None
    pool_usage: Optional[CookUsage] = None

    def __post_init__(self):
        n = len(self.quota_count)
        for f in ("quota_count", "quota_cpus", "quota_mem", "quota_gpus", "usage_count", "usage_cpus", "usage_mem", "usage_gpus"):
            setattr(self, f, _arr(getattr(self, f), np.float64, n))
        self.tokens_left = _arr(self.tokens_left, np.int64, n)

    @property
    def n(self):
        return len(self.quota_count)

    def as_struct(self) -> CookUserState:
        s = CookUserState()
        s.n = self.n
        for f in ("quota_count", "quota_cpus", "quota_mem", "quota_gpus", "usage_count", "usage_cpus", "usage_mem", "usage_gpus"):
            setattr(s, f, _ptr(getattr(self, f), _f64p))
        s.tokens_left = _ptr(self.tokens_left, _i64p)
        s.enforce_rate_limit = int(bool(self.enforce_rate_limit))
        s.has_pool_quota = int(self.pool_quota is not None)
        if self.pool_quota is not None:
            s.pool_quota = self.pool_quota
        s.pool_usage_given = int(self.pool_usage is not None)
        if self.pool_usage is not None:
            s.pool_usage = self.pool_usage
        return s


def usage(count=0.0, cpus=0.0, mem=0.0, gpus=0.0) -> CookUsage:
    return CookUsage(float(count), float(cpus), float(mem), float(gpus))


def quota(count=2.0 ** 31 - 1, cpus=DMAX, mem=DMAX, gpus=DMAX) -> CookUsage:
    return CookUsage(float(count), float(cpus), float(mem), float(gpus))


def pool_quota(pool_quota: Optional[CookUsage] = None, group_quota: Optional[CookUsage] = None,
               group_usage: Optional[CookUsage] = None, pool_usage: Optional[CookUsage] = None) -> CookPoolQuota:
    q = CookPoolQuota()
    q.has_pool_quota = int(pool_quota is not None)
    q.has_group_quota = int(group_quota is not None and group_usage is not None)
    if pool_quota is not None:
        q.pool_quota = pool_quota
    if q.has_group_quota:
        q.group_quota = group_quota
        q.group_usage = group_usage
    q.pool_usage_given = int(pool_usage is not None)
    if pool_usage is not None:
        q.pool_usage = pool_usage
    return q


def _csr(lists, n):
    off = np.zeros(n + 1, dtype=np.uint32)
    flat = []
    for i in range(n):
        flat.extend(lists[i] if lists is not None else [])
        off[i + 1] = len(flat)
    return off, flat


def _csr_take(off, cols, idx):
    """Rows `idx` of a CSR table (offsets `off`, payload columns `cols`) -> (new offsets, new columns); no per-row Python."""
    off = off.astype(np.int64)
    starts, lens = off[idx], off[idx + 1] - off[idx]
    new_off = np.zeros(len(idx) + 1, dtype=np.int64)
    np.cumsum(lens, out=new_off[1:])
    pos = np.arange(new_off[-1], dtype=np.int64) + np.repeat(starts - new_off[:-1], lens)
    return new_off.astype(np.uint32), tuple(np.ascontiguousarray(c[pos], dtype=np.uint32) for c in cols)


@dataclass
class Jobs:
    """Considerable jobs in rank order = Fenzo TaskRequests (scheduler.clj:456-509)."""
    cpus: np.ndarray
    mem: np.ndarray
    gpus: Optional[np.ndarray] = None
    gpu_model: Optional[np.ndarray] = None
    user: Optional[np.ndarray] = None
    group: Optional[np.ndarray] = None
    eq_off: Optional[np.ndarray] = None
    eq_key: Optional[np.ndarray] = None
    eq_val: Optional[np.ndarray] = None
    novel_off: Optional[np.ndarray] = None
    novel_host: Optional[np.ndarray] = None
    reserved_host: Optional[np.ndarray] = None
    ckpt_location: Optional[np.ndarray] = None
    est_end_ms: Optional[np.ndarray] = None
    disk_request: Optional[np.ndarray] = None
    disk_type: Optional[np.ndarray] = None
    ports: Optional[np.ndarray] = None    # number of ports asked for (scheduler.clj:466)
    scalars: Optional[np.ndarray] = None  # [n, n_scalars] named scalar requests, NaN = none (scheduler.clj:177-189)

    def __post_init__(self):
        n = len(self.cpus)
        self.cpus = _arr(self.cpus, np.float64, n)
        self.mem = _arr(self.mem, np.float64, n)
        self.ports = _arr(self.ports, np.int32, n)
        self.scalars, self.n_scalars = _table(self.scalars, np.float64, n, MAX_SCALARS)
        self.gpus = _arr(self.gpus, np.float64, n)
        self.gpu_model = _arr(self.gpu_model, np.uint32, n)
        self.user = _arr(self.user, np.uint32, n)
        self.group = _arr(self.group, np.uint32, n)
        self.eq_off = _arr(self.eq_off, np.uint32)
        self.eq_key = _arr(self.eq_key, np.uint32)
        self.eq_val = _arr(self.eq_val, np.uint32)
        self.novel_off = _arr(self.novel_off, np.uint32)
        self.novel_host = _arr(self.novel_host, np.uint32)
        self.reserved_host = _arr(self.reserved_host, np.int32, n)
        self.ckpt_location = _arr(self.ckpt_location, np.uint32, n)
        self.est_end_ms = _arr(self.est_end_ms, np.int64, n)
        self.disk_request = _arr(self.disk_request, np.float64, n)
        self.disk_type = _arr(self.disk_type, np.uint32, n)
        if self.eq_off is not None:
            assert len(self.eq_off) == n + 1
            if self.eq_key is None or len(self.eq_key) == 0:  # keep non-NULL pointers for empty CSR payloads
                self.eq_key = np.zeros(1, np.uint32)
                self.eq_val = np.zeros(1, np.uint32)
        if self.novel_off is not None:
            assert len(self.novel_off) == n + 1
            if self.novel_host is None or len(self.novel_host) == 0:
                self.novel_host = np.zeros(1, np.uint32)

    @staticmethod
    def with_constraints(cpus, mem, equals=None, novel=None, **kw) -> "Jobs":
        """equals: per job list of (key, value); novel: per job list of host ids."""
        n = len(cpus)
        if equals is not None:
            off, flat = _csr(equals, n)
            kw.update(eq_off=off, eq_key=np.array([k for k, _ in flat], dtype=np.uint32),
                      eq_val=np.array([v for _, v in flat], dtype=np.uint32))
        if novel is not None:
            off, flat = _csr(novel, n)
            kw.update(novel_off=off, novel_host=np.array(flat, dtype=np.uint32))
        return Jobs(cpus=cpus, mem=mem, **kw)

    @property
    def n(self):
        return len(self.cpus)

    def take(self, idx) -> "Jobs":
        """Jobs re-ordered/subset by index array (used to turn pending jobs into considerable-in-rank-order)."""
        idx = np.asarray(idx, dtype=np.int64)
        kw = {}
        for name in ("cpus", "mem", "gpus", "gpu_model", "user", "group", "reserved_host", "ckpt_location",
                     "est_end_ms", "disk_request", "disk_type", "ports", "scalars"):
            a = getattr(self, name)
            kw[name] = None if a is None else a[idx]
        if self.eq_off is not None:
            kw["eq_off"], (kw["eq_key"], kw["eq_val"]) = _csr_take(self.eq_off, (self.eq_key, self.eq_val), idx)
        if self.novel_off is not None:
            kw["novel_off"], (kw["novel_host"],) = _csr_take(self.novel_off, (self.novel_host,), idx)
        return Jobs(**kw)

    def _scalar_cols(self):
        if self.scalars is None:
            return None
        self._cols = np.ascontiguousarray(self.scalars.T).reshape(-1)  # the ABI takes one contiguous column per name
        return self._cols

    def as_struct(self) -> CookJobs:
        return _memo_struct(self, self._build_struct)

    def _build_struct(self) -> CookJobs:
        return CookJobs(self.n, _ptr(self.cpus, _f64p), _ptr(self.mem, _f64p), _ptr(self.gpus, _f64p),
                        _ptr(self.gpu_model, _u32p), _ptr(self.user, _u32p), _ptr(self.group, _u32p),
                        _ptr(self.eq_off, _u32p), _ptr(self.eq_key, _u32p), _ptr(self.eq_val, _u32p),
                        _ptr(self.novel_off, _u32p), _ptr(self.novel_host, _u32p),
                        _ptr(self.reserved_host, _i32p), _ptr(self.ckpt_location, _u32p),
                        _ptr(self.est_end_ms, _i64p), _ptr(self.disk_request, _f64p), _ptr(self.disk_type, _u32p),
                        _ptr(self.ports, _i32p), self.n_scalars, 0,
                        _ptr(self._scalar_cols(), _f64p))


@dataclass
class Offers:
    """One lease per host (offer.clj:31-76) plus Fenzo's running-task view of the host."""
    cpus: np.ndarray
    mem: np.ndarray
    host: Optional[np.ndarray] = None
    k8s: Optional[np.ndarray] = None
    gpu_model: Optional[np.ndarray] = None
    gpu_count: Optional[np.ndarray] = None
    disk_type: Optional[np.ndarray] = None
    disk_space: Optional[np.ndarray] = None
    attr: Optional[np.ndarray] = None  # [n, n_attr_keys] uint32, 0 = absent
    max_tasks: Optional[np.ndarray] = None
    num_tasks: Optional[np.ndarray] = None
    location: Optional[np.ndarray] = None
    host_start_s: Optional[np.ndarray] = None
    run_cpus: Optional[np.ndarray] = None
    run_mem: Optional[np.ndarray] = None
    run_count: Optional[np.ndarray] = None
    ports: Optional[np.ndarray] = None    # ports in the lease's ranges (offer.clj:71-73)
    scalars: Optional[np.ndarray] = None  # [n, n_scalars] lease getScalarValues under the jobs' scalar names (offer.clj:57-65)

    def __post_init__(self):
        n = len(self.cpus)
        self.cpus = _arr(self.cpus, np.float64, n)
        self.mem = _arr(self.mem, np.float64, n)
        self.host = _arr(self.host if self.host is not None else np.arange(n), np.uint32, n)
        self.k8s = _arr(self.k8s, np.uint8, n)
        self.ports = _arr(self.ports, np.int32, n)
        self.scalars, self.n_scalars = _table(self.scalars, np.float64, n, MAX_SCALARS)
        # gpu_model / gpu_count (disk_type / disk_space): [n], or [n, slots] for hosts whose k8s map has several entries
        self.gpu_model, self.gpu_slots = _table(self.gpu_model, np.uint32, n, MAX_RES_SLOTS)
        self.gpu_count, gc = _table(self.gpu_count, np.float64, n, MAX_RES_SLOTS)
        if self.gpu_model is not None and self.gpu_count is None:
            self.gpu_count = np.zeros((n, self.gpu_slots))
        assert self.gpu_model is None or self.gpu_count.shape == self.gpu_model.shape
        self.disk_type, self.disk_slots = _table(self.disk_type, np.uint32, n, MAX_RES_SLOTS)
        self.disk_space, ds = _table(self.disk_space, np.float64, n, MAX_RES_SLOTS)
        assert (self.disk_type is None) == (self.disk_space is None) and (self.disk_type is None or self.disk_space.shape == self.disk_type.shape)
        for name in ("gpu_model", "gpu_count", "disk_type", "disk_space"):  # the plain per-host columns stay 1-d
            a = getattr(self, name)
            if a is not None and a.shape[1] == 1:
                setattr(self, name, a.reshape(n))
        if self.attr is not None:
            self.attr = np.ascontiguousarray(self.attr, dtype=np.uint32)
            assert self.attr.ndim == 2 and self.attr.shape[0] == n
        self.max_tasks = _arr(self.max_tasks, np.int32, n)
        self.num_tasks = _arr(self.num_tasks, np.int32, n)
        if self.max_tasks is not None and self.num_tasks is None:
            self.num_tasks = np.zeros(n, np.int32)
        self.location = _arr(self.location, np.uint32, n)
        self.host_start_s = _arr(self.host_start_s, np.int64, n)
        self.run_cpus = _arr(self.run_cpus, np.float64, n)
        self.run_mem = _arr(self.run_mem, np.float64, n)
        self.run_count = _arr(self.run_count, np.int32, n)

    @property
    def n(self):
        return len(self.cpus)

    @property
    def n_attr_keys(self):
        return 0 if self.attr is None else self.attr.shape[1]

    def _scalar_cols(self):
        if self.scalars is None:
            return None
        self._cols = np.ascontiguousarray(self.scalars.T).reshape(-1)
        return self._cols

    def as_struct(self) -> CookOffers:
        return _memo_struct(self, self._build_struct)

    def _build_struct(self) -> CookOffers:
        attr = None if self.attr is None else self.attr.reshape(-1)
        flat = lambda a: None if a is None else a.reshape(-1)
        return CookOffers(self.n, _ptr(self.cpus, _f64p), _ptr(self.mem, _f64p), _ptr(self.host, _u32p),
                          _ptr(self.k8s, _u8p), _ptr(flat(self.gpu_model), _u32p), _ptr(flat(self.gpu_count), _f64p),
                          _ptr(flat(self.disk_type), _u32p), _ptr(flat(self.disk_space), _f64p),
                          self.n_attr_keys, _ptr(attr, _u32p),
                          _ptr(self.max_tasks, _i32p), _ptr(self.num_tasks, _i32p),
                          _ptr(self.location, _u32p), _ptr(self.host_start_s, _i64p),
                          _ptr(self.run_cpus, _f64p), _ptr(self.run_mem, _f64p), _ptr(self.run_count, _i32p),
                          self.gpu_slots if self.gpu_model is not None else 0, self.disk_slots if self.disk_type is not None else 0,
                          _ptr(self.ports, _i32p), self.n_scalars, 0, _ptr(self._scalar_cols(), _f64p))


@dataclass
class Groups:
    """Job groups with host-placement constraints (constraints.clj:519-678)."""
    type: np.ndarray  # 0 all, 1 unique, 2 balanced, 3 attribute-equals
    attr_key: Optional[np.ndarray] = None
    minimum: Optional[np.ndarray] = None
    run_hosts: Optional[list] = None  # per group: host ids of cotasks already running
    run_attrs: Optional[list] = None  # per group: attr value id of each running cotask's host
    _run_off: np.ndarray = field(init=False, default=None)
    _run_host: np.ndarray = field(init=False, default=None)
    _run_attr: np.ndarray = field(init=False, default=None)

    def __post_init__(self):
        n = len(self.type)
        self.type = _arr(self.type, np.uint8, n)
        self.attr_key = _arr(self.attr_key if self.attr_key is not None else np.full(n, NONE_U32), np.uint32, n)
        self.minimum = _arr(self.minimum if self.minimum is not None else np.zeros(n), np.int32, n)
        off, flat = _csr(self.run_hosts, n)
        _, flat_a = _csr(self.run_attrs, n)
        if self.run_attrs is None:
            flat_a = [0] * len(flat)
        assert len(flat_a) == len(flat)
        self._run_off = off
        self._run_host = np.array(flat if flat else [0], dtype=np.uint32)
        self._run_attr = np.array(flat_a if flat_a else [0], dtype=np.uint32)

    @property
    def n(self):
        return len(self.type)

    def as_struct(self) -> CookGroups:
        return CookGroups(self.n, _ptr(self.type, _u8p), _ptr(self.attr_key, _u32p), _ptr(self.minimum, _i32p),
                          _ptr(self._run_off, _u32p), _ptr(self._run_host, _u32p), _ptr(self._run_attr, _u32p))


@dataclass
class HostSpare:
    host: np.ndarray
    cpus: np.ndarray
    mem: np.ndarray
    gpus: Optional[np.ndarray] = None

    def __post_init__(self):
        n = len(self.host)
        self.host = _arr(self.host, np.uint32, n)
        self.cpus = _arr(self.cpus, np.float64, n)
        self.mem = _arr(self.mem, np.float64, n)
        self.gpus = _arr(self.gpus if self.gpus is not None else np.zeros(n), np.float64, n)

    def as_struct(self) -> CookHostSpare:
        return CookHostSpare(len(self.host), _ptr(self.host, _u32p), _ptr(self.cpus, _f64p), _ptr(self.mem, _f64p),
                             _ptr(self.gpus, _f64p))


# ---- offer construction from node state (kubernetes/compute_cluster.clj:68-190) ---------------------------------------
NODE_UNSCHEDULABLE, NODE_OTHER_TAINTS, NODE_BLOCKLIST_LABEL, NODE_GPU_TAINT = 1, 2, 4, 8
POD_SYNTHETIC, POD_NO_REQUESTS = 1, 2
NODE_ST_OFFER, NODE_ST_CONSUMED, NODE_ST_FOREIGN_GPU, NODE_ST_FOREIGN_DISK = 1, 2, 4, 8


class CookNodes(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("host", _u32p), ("cpus", _f64p), ("mem", _f64p), ("gpus", _i32p), ("gpu_model", _u32p),
        ("disk", _f64p), ("disk_type", _u32p), ("flags", _u8p),
        ("n_attr_keys", C.c_uint32), ("attr", _u32p),
    ]


class CookPods(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("node", _u32p), ("cpus", _f64p), ("mem", _f64p), ("gpus", _i32p), ("gpu_model", _u32p),
        ("disk", _f64p), ("disk_type", _u32p), ("flags", _u8p),
    ]


class CookOfferParams(C.Structure):
    _fields_ = [
        ("clobber_synthetic_pods", C.c_int32), ("filter_out_unsound_gpu_nodes", C.c_int32),
        ("max_pods_per_node", C.c_int32), ("n_gpu_models", C.c_uint32), ("n_disk_types", C.c_uint32),
        ("gpu_slots", C.c_uint32), ("disk_slots", C.c_uint32),
    ]


class CookNodeOffers(C.Structure):
    _fields_ = [
        ("node", _u32p), ("host", _u32p), ("cpus", _f64p), ("mem", _f64p), ("gpu_model", _u32p), ("gpu_count", _f64p),
        ("disk_type", _u32p), ("disk_space", _f64p), ("num_pods", _i32p), ("attr", _u32p),
    ]


class CookOfferTotals(C.Structure):
    _fields_ = [
        ("cpus_capacity", C.c_double), ("mem_capacity", C.c_double), ("cpus_consumed", C.c_double), ("mem_consumed", C.c_double),
        ("nodes_total", C.c_uint32), ("nodes_schedulable", C.c_uint32),
    ]


@dataclass
class Nodes:
    """node-name->node of one pool in ascending node-name order (api.clj:874-884 get-capacity inputs + the host-evaluated
    predicates of node-schedulable?, api.clj:782-847)."""
    cpus: np.ndarray
    mem: np.ndarray
    host: Optional[np.ndarray] = None
    gpus: Optional[np.ndarray] = None
    gpu_model: Optional[np.ndarray] = None
    disk: Optional[np.ndarray] = None       # < 0 = no allocatable ephemeral-storage
    disk_type: Optional[np.ndarray] = None
    flags: Optional[np.ndarray] = None
    attr: Optional[np.ndarray] = None       # [n, n_attr_keys] label table in the Offers.attr encoding

    def __post_init__(self):
        n = len(self.cpus)
        self.cpus = _arr(self.cpus, np.float64, n)
        self.mem = _arr(self.mem, np.float64, n)
        self.host = _arr(self.host if self.host is not None else np.arange(n), np.uint32, n)
        self.gpus = _arr(self.gpus, np.int32, n)
        self.gpu_model = _arr(self.gpu_model, np.uint32, n)
        self.disk = _arr(self.disk, np.float64, n)
        self.disk_type = _arr(self.disk_type, np.uint32, n)
        self.flags = _arr(self.flags, np.uint8, n)
        if self.attr is not None:
            self.attr = np.ascontiguousarray(self.attr, dtype=np.uint32)
            assert self.attr.ndim == 2 and self.attr.shape[0] == n

    @property
    def n(self):
        return len(self.cpus)

    @property
    def n_attr_keys(self):
        return 0 if self.attr is None else self.attr.shape[1]

    def as_struct(self) -> CookNodes:
        attr = None if self.attr is None else self.attr.reshape(-1)
        return CookNodes(self.n, _ptr(self.host, _u32p), _ptr(self.cpus, _f64p), _ptr(self.mem, _f64p), _ptr(self.gpus, _i32p),
                         _ptr(self.gpu_model, _u32p), _ptr(self.disk, _f64p), _ptr(self.disk_type, _u32p), _ptr(self.flags, _u8p),
                         self.n_attr_keys, _ptr(attr, _u32p))


@dataclass
class Pods:
    """Every pod of node-name->pods (api.clj:886-930 get-consumption inputs): per-pod sums of the containers' requests."""
    node: np.ndarray                         # index into Nodes, NONE_U32 = no node of this pool
    cpus: np.ndarray
    mem: np.ndarray
    gpus: Optional[np.ndarray] = None
    gpu_model: Optional[np.ndarray] = None
    disk: Optional[np.ndarray] = None        # < 0 = no container asks for ephemeral-storage
    disk_type: Optional[np.ndarray] = None
    flags: Optional[np.ndarray] = None

    def __post_init__(self):
        n = len(self.node)
        self.node = _arr(self.node, np.uint32, n)
        self.cpus = _arr(self.cpus, np.float64, n)
        self.mem = _arr(self.mem, np.float64, n)
        self.gpus = _arr(self.gpus, np.int32, n)
        self.gpu_model = _arr(self.gpu_model, np.uint32, n)
        self.disk = _arr(self.disk, np.float64, n)
        self.disk_type = _arr(self.disk_type, np.uint32, n)
        self.flags = _arr(self.flags, np.uint8, n)

    @property
    def n(self):
        return len(self.node)

    def as_struct(self) -> CookPods:
        return CookPods(self.n, _ptr(self.node, _u32p), _ptr(self.cpus, _f64p), _ptr(self.mem, _f64p), _ptr(self.gpus, _i32p),
                        _ptr(self.gpu_model, _u32p), _ptr(self.disk, _f64p), _ptr(self.disk_type, _u32p), _ptr(self.flags, _u8p))


def offer_params(clobber_synthetic_pods=False, filter_out_unsound_gpu_nodes=False, max_pods_per_node=2 ** 31 - 1,
                 n_gpu_models=0, n_disk_types=0, gpu_slots=1, disk_slots=1) -> CookOfferParams:
    """gpu_slots / disk_slots: entries per offer row of the "gpus" / "disk" maps (the node's own model plus the models only its
    pods name); 1 = the plain per-host columns"""
    return CookOfferParams(int(bool(clobber_synthetic_pods)), int(bool(filter_out_unsound_gpu_nodes)), int(max_pods_per_node),
                           int(n_gpu_models), int(n_disk_types), int(gpu_slots), int(disk_slots))


@dataclass
class BuiltOffers:
    """What cook_offers_fetch returns: the offer rows (the cook_offers columns of the same names), the per-node status
    bits, the gauges and the per-model / per-type totals."""
    node: np.ndarray
    host: np.ndarray
    cpus: np.ndarray
    mem: np.ndarray
    gpu_model: np.ndarray
    gpu_count: np.ndarray
    disk_type: np.ndarray
    disk_space: np.ndarray
    num_pods: np.ndarray
    attr: Optional[np.ndarray]
    node_status: np.ndarray
    totals: dict
    gpu_capacity_by_model: np.ndarray
    gpu_consumed_by_model: np.ndarray
    disk_capacity_by_type: np.ndarray
    disk_consumed_by_type: np.ndarray

    @property
    def n(self):
        return len(self.node)

    def as_offers(self, **kw) -> "Offers":
        """The rows as match input (offer.clj:31-76: Kubernetes leases carry compute-cluster-type = kubernetes)."""
        return Offers(cpus=self.cpus, mem=self.mem, host=self.host, k8s=np.ones(self.n, np.uint8), gpu_model=self.gpu_model,
                      gpu_count=self.gpu_count, disk_type=self.disk_type, disk_space=self.disk_space, attr=self.attr, **kw)


# ---- why-unscheduled summaries and match-cycle metrics ------------------------------------------------------------------------
WHY_SLOTS = 20
WHY_SCALAR0, WHY_PORTS = 14, 17
WHY_NAMES = {  # slot -> the key of fenzo-utils/summarize-placement-failure's map (fenzo_utils.clj:33-55)
    0: (":resources", "cpus"), 1: (":resources", "mem"), 2: (":resources", "fitness"),
    3: (":constraints", "checkpoint_locality_constraint"), 4: (":constraints", "estimated_completion_constraint"),
    5: (":constraints", "user_defined_constraint"), 6: (":constraints", "disk_host_constraint"),
    7: (":constraints", "gpu_host_constraint"), 8: (":constraints", "novel_host_constraint"),
    9: (":constraints", "max_tasks_per_host"), 10: (":constraints", "rebalancer_reservation_constraint"),
    11: (":constraints", "unique_host_placement_group_constraint"),
    12: (":constraints", "balanced_host_placement_group_constraint"),
    13: (":constraints", "attribute_equals_host_placement_group_constraint"),
}


def why_summary(row, scalar_names=()) -> dict:
    """one COOK_WHY_* row -> the reference's {:resources {...} :constraints {...}} map (zero counts omitted).  scalar_names: the
    caller's named-scalar table (cook_jobs.scalars columns); a name of "cpus" / "mem" replaces the cpus / mem slot (the summary's
    "cpus" / "mem" entries ARE the named-scalar failures: the message field, fenzo_utils.clj:21-45)."""
    out: dict = {}
    named = {name: WHY_SCALAR0 + s for s, name in enumerate(scalar_names)}
    for slot, (kind, name) in WHY_NAMES.items():
        src = named.get(name, slot) if kind == ":resources" else slot
        if row[src]:
            out.setdefault(kind, {})[name] = int(row[src])
    for name, slot in named.items():
        if name not in ("cpus", "mem") and row[slot]:
            out.setdefault(":resources", {})[name] = int(row[slot])
    return out


class CookResourceStats(C.Structure):
    _fields_ = [("total_cpus", C.c_double), ("total_mem", C.c_double),
                ("p50_cpus", C.c_double), ("p95_cpus", C.c_double), ("p100_cpus", C.c_double),
                ("p50_mem", C.c_double), ("p95_mem", C.c_double), ("p100_mem", C.c_double),
                ("largest_by_cpus", C.c_uint32), ("largest_by_mem", C.c_uint32)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class CookCycleMetrics(C.Structure):
    _fields_ = [("considerable", C.c_uint32), ("matched", C.c_uint32), ("unmatched", C.c_uint32),
                ("offers", C.c_uint32), ("offers_scheduled", C.c_uint32), ("head_matched", C.c_uint32),
                ("reserved", C.c_uint32 * 2), ("jobs", CookResourceStats), ("offer_stats", CookResourceStats)]


CONSTRAINT_MESSAGES = {  # unscheduled.clj:71-75 constraint-name->message
    "novel_host_constraint": "Job already ran on this host.",
    "gpu_host_constraint": "Host has no GPU support.",
    "non_gpu_host_constraint": "Host is reserved for jobs that need GPU support.",
    "attribute-equals-host-placement-group-constraint": "Host had a different attribute than other jobs in the group.",
}


def why_reasons(summary: dict) -> list:
    """unscheduled/fenzo-failures-for-user (unscheduled.clj:77-93): the summary map -> [{:reason .. :host_count ..} ...],
    resources first, then constraints (unknown constraint names are shown as they are)."""
    out = [dict(reason=f"Not enough {k} available.", host_count=v) for k, v in summary.get(":resources", {}).items()]
    out += [dict(reason=CONSTRAINT_MESSAGES.get(k, k), host_count=v) for k, v in summary.get(":constraints", {}).items()]
    return out

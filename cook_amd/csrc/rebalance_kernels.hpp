// rebalance_kernels.hpp — device side of cook_rebalance: the rebalancer's preemption decisions for one pool
// (rebalancer.clj:222-266 init-state, :157-220 pending-job DRU / job-below-quota, :320-407 compute-preemption-decision,
//  :270-309 next-state, :434-467 rebalance; dru.clj:128-144).
//
// State.  Slots: s < R = running task s, slot R + pj = the task pending job pj becomes when a decision places it.  All
// S = R + P slots are sorted ONCE into per-user order (index space B, same keys as the rank path: tools.clj:614-641; a
// pending slot sits exactly where `(conj task-set synthetic-task)` would insert it) and carry an `act` bit: running slots
// start active, a preempted task is masked out, a placed job is masked in.  Adding 0.0 is exact, so "recompute the DRUs of
// the changed users" (dru.clj:128-144) is a masked re-run of the segmented prefix scan of the rank path (exactness fix-up
// included), and the priority map (rebalancer.clj:252-256) never has to be materialised: its order (-dru, user, position
// in the user's list) is only needed WITHIN one host, where the decision kernel sorts the few candidates in place.
//
// One pending job = three kernels (+ the masked re-scan):
//   rebal_job_prep : ONE wave.  job-below-quota (the job first, then the user's active tasks, left to right), the
//                    pending-job DRU (nearest active task at or before the job's slot), the group's cohost values.
//   rebal_decide   : wave per host.  Filters the host's tasks (running, in host-sorted order, ++ jobs placed on it this
//                    cycle), sorts the candidates by (dru desc, B position asc) by counting, forms the prefix aggregates
//                    seeded with the host's spare resources exactly left to right, keeps the feasible prefix with the
//                    greatest last DRU (ties -> longest), evaluates the job / group constraints on the host's attribute
//                    row (through the "last scored task of the host" slave-id quirk, rebalancer.clj:369-375).
//   rebal_apply    : ONE workgroup.  arg-max over hosts (ties -> last host in name order, rebalancer.clj:404), then
//                    next-state: mask the preempted tasks out, the job in, host spare := aggregate - job, remaining - 1.
// The whole loop runs on the device; the host only enqueues.
#pragma once
#include "../../include/cookmatch.h"
#include "common.hpp"
#include "scan.hpp"

constexpr int RB_CAP = 128;    // candidates per host kept in LDS (larger hosts use the global scratch)
constexpr int RB_WAVES = 4;    // hosts per decide block

struct RebalCtl {
  int remaining;        // max-preemption budget left (rebalancer.clj:442)
  unsigned nd, np;      // decisions / preempted tasks emitted
  unsigned n_pre_hosts; // hosts of the tasks preempted so far whose attribute map is known (constraints.clj:686-689)
  unsigned n_x;         // jobs placed so far this cycle (x_pj, sorted by host)
  unsigned n_changed;   // users whose active set the last decision changed (chg[])
  unsigned n_tiles;     // tiles of RB_RS_TILE slots their segments are cut into (chg_tile[])
  unsigned pad[1];
};

struct RebalJob {  // context of the pending job being decided (written by rebal_job_prep)
  unsigned active;  // 0: the loop is over (budget spent)
  unsigned pj, us;
  unsigned below;   // job-below-quota
  unsigned has_gpus;  // (:gpus resources) of the job is positive
  unsigned gtype, gkey, n_co;  // group type (0 none), attribute key, number of cohosts staged in co_val
  int minim, maxfreq;          // balanced: minim as constraints.clj:611-615 computes it, max frequency
  unsigned pad0, pad1;
  double pdru, c, m, g;
};

struct RebalIn {
  unsigned R, P, S, U, H;
  int dru_mode;
  long long host_lifetime_mins;
  double safe_dru, min_diff;
  // slots, A space
  const uint32_t* slot_user;
  const double *slot_cpus, *slot_mem, *slot_gpus;
  const uint32_t* posB;
  const uint8_t* attrs_cached;  // [R] or null (= all cached)
  // B space
  const SumU4* s_use;
  const uint32_t *seg_start, *seg_end;
  uint8_t* act;
  const double* dru;
  const SumU4* pre;            // masked per-user inclusive prefix sums (exact), refreshed after every decision
  const uint32_t* user_safe;   // [U] 1: every resource of the user's slots is a small multiple of 2^-10 -> sums exact in ANY order
  // users
  const double *q_count, *q_cpus, *q_mem, *q_gpus, *div_cpus, *div_mem, *div_gpus;
  // hosts
  const uint32_t* hperm;          // [R] running slots grouped by host
  const uint32_t *hstart, *hend;  // [H]
  const uint32_t* hbase;          // [H] running tasks on the hosts before this one (exclusive scan of the hosts' sizes)
  // the running slots' columns mirrored in HOST order (index = position in hperm): rebal_decide streams a host's tasks
  // instead of chasing slot -> position-in-user-order -> value through three dependent random loads
  const uint32_t *h_pb, *h_user;  // position in per-user order (static), user (static)
  const double *h_cpus, *h_mem, *h_gpus;
  double* h_dru;                  // refreshed by the re-scoring of the users a decision touched
  uint8_t* h_act;                 // cleared when a task is preempted
  const uint32_t* hidx;           // [S] B position -> index in host order (running slots), COOK_NONE otherwise
  uint32_t* chg;                  // [S + 1] users to re-score after the last decision (each once)
  uint32_t* chg_tile;             // [S + 2] first tile of each of them, total at [n_changed]
  uint32_t* chg_bad;              // [S + 1] an addition rounded while re-scoring that user
  uint32_t* chg_mark;             // [U] stamp of the decision that last listed the user (deduplication)
  SumU4 *tile_agg, *tile_carry;   // [S / RB_RS_TILE + S + 2] tile totals / carries
  SumU4* pre_w;                   // writable aliases of pre / dru
  double* dru_w;
  const int32_t* row_of_host;     // [H] row of the host in the attribute table, -1 = not cached
  double *spare_c, *spare_m, *spare_g;
  uint8_t* has_spare;
  // host attribute table (the agent-attributes-cache, scheduler.clj:1586-1597), one row per cached host
  unsigned n_attr;
  const uint32_t* a_host;
  const uint8_t* a_k8s;
  const uint32_t* a_gpu_model;
  const double* a_gpu_count;
  const uint32_t* a_disk_type;
  const double* a_disk_space;
  unsigned a_gpu_slots, a_disk_slots;  // entries per host in the k8s "gpus" / "disk" maps (>= 1)
  const uint32_t* a_attr;
  const uint32_t* a_location;
  const int64_t* a_host_start;
  // pending jobs
  const double *j_cpus, *j_mem, *j_gpus;
  const uint32_t *j_gpu_model, *j_user, *j_group, *j_eq_off, *j_eq_key, *j_eq_val, *j_novel_off, *j_novel_host, *j_ckpt, *j_disk_type;
  const int64_t* j_est_end;
  const double* j_disk_req;
  // groups
  unsigned G;
  const uint8_t* g_type;
  const uint32_t* g_attr_key;
  const int32_t* g_min;
  const uint32_t *g_run_off, *g_run_host;
  // dynamic lists
  uint32_t* x_pj;       // [P] placed jobs sorted by host
  uint32_t* x_host;     // [P] by pj
  uint8_t* x_known;     // [P] by pj: the placed task carries a slave id whose attributes are cached
  uint32_t *x_before, *x_cnt;  // [H] placed jobs on hosts before this one / on this one
  uint32_t* pre_hosts;  // [S]
  uint32_t* co_val;     // cohost values of the current job's group
  // per-host results of rebal_decide
  unsigned long long* hres_key;  // f64_key(dru) of the host's best prefix, 0 = none
  uint32_t *hres_len, *hres_base;
  double *hres_dru, *hres_c, *hres_m, *hres_g;
  uint32_t* srt_slot;  // [S] the host's candidates in priority order at [hres_base ...]
  // global scratch for hosts with more than RB_CAP items (same addressing as srt_slot)
  double *gs_dru, *gs_cpus, *gs_mem, *gs_gpus;
  uint32_t *gs_posB, *gs_slot, *gs_ord;
  // outputs
  cook_preemption* decisions;
  uint32_t* preempted;
  double* pending_dru;
  RebalCtl* ctl;
  RebalJob* job;
};

struct LoadMaskedU4 {  // usage of slot i if active, else zeros (x + 0.0 == x: masked-out slots do not perturb the sums)
  const SumU4* p;
  const uint8_t* act;
  __device__ __forceinline__ SumU4 operator()(unsigned i) const { return act[i] ? p[i] : SumU4::zero(); }
};

static __device__ __forceinline__ SumU4 wave_incl_scan_u4(SumU4 v) {
  const unsigned lane = lane_id();
  for (unsigned d = 1; d < COOK_WAVE; d <<= 1) {
    const SumU4 p = shfl_up_v(v, d);
    if (lane >= d) v = combine(p, v);
  }
  return v;
}
static __device__ __forceinline__ SumU4 wave_bcast_u4(const SumU4& v, int src) {
  SumU4 r;
  r.count = __shfl(v.count, src, COOK_WAVE);
  r.cpus = __shfl(v.cpus, src, COOK_WAVE);
  r.mem = __shfl(v.mem, src, COOK_WAVE);
  r.gpus = __shfl(v.gpus, src, COOK_WAVE);
  r.bad = __shfl(v.bad, src, COOK_WAVE);
  return r;
}

// value of attribute `key` in the map of row r (r < 0: nil map -> 0 = absent); COOK_NONE_U32 = "HOSTNAME"
static __device__ __forceinline__ uint32_t rebal_attr(const RebalIn& in, int r, uint32_t key) {
  if (r < 0) return 0u;
  if (key == 0xFFFFFFFFu) return in.a_host[r] + 1u;
  if (!in.a_attr || key >= in.n_attr) return 0u;
  return in.a_attr[(size_t)r * in.n_attr + key];
}

// ---- exact re-scoring after a decision: masked sequential fix-up + DRU -------------------------------------------------
__global__ void __launch_bounds__(256) rebal_fix_inexact(const SumU4* __restrict__ s_use, const uint8_t* __restrict__ act,
                                                         SumU4* __restrict__ pre, const uint32_t* __restrict__ seg_start,
                                                         const uint32_t* __restrict__ seg_end, uint32_t* __restrict__ inexact_user,
                                                         unsigned n_users) {
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_users || !inexact_user[u]) return;
  inexact_user[u] = 0u;  // ready for the next re-scan
  double c = 0.0, cp = 0.0, m = 0.0, g = 0.0;
  bool first = true;
  for (unsigned i = seg_start[u]; i < seg_end[u]; ++i) {
    if (act[i]) {
      const SumU4 x = s_use[i];
      if (first) {
        c = x.count, cp = x.cpus, m = x.mem, g = x.gpus;
        first = false;
      } else {
        c += x.count, cp += x.cpus, m += x.mem, g += x.gpus;
      }
    }
    pre[i] = SumU4{c, cp, m, g, 0u};
  }
}

__global__ void __launch_bounds__(256) rebal_score(const SumU4* __restrict__ pre, const uint32_t* __restrict__ s_user, unsigned n,
                                                   int dru_mode, const double* __restrict__ div_cpus,
                                                   const double* __restrict__ div_mem, const double* __restrict__ div_gpus,
                                                   double* __restrict__ dru) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned u = s_user[i];
  const SumU4 s = pre[i];
  double d;
  if (dru_mode == 1) {
    d = s.gpus / div_gpus[u];
  } else {
    const double a = s.mem / div_mem[u], b = s.cpus / div_cpus[u];
    d = a > b ? a : b;
  }
  dru[i] = d;
}

// A user is "safe" when every resource value of its slots is a non-negative multiple of 2^-10 below 2^22 and it has fewer than
// 2^20 slots: every partial sum of such values is exactly representable, so sums are the same in any association and
// job-below-quota (the job first, then the user's tasks left to right) may reuse the prefix scan's total.
__global__ void __launch_bounds__(256) rebal_user_safe(const uint32_t* __restrict__ user, const double* __restrict__ cpus,
                                                       const double* __restrict__ mem, const double* __restrict__ gpus, unsigned n,
                                                       const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_end,
                                                       uint32_t* __restrict__ user_safe) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  auto ok = [](double v) {
    const double s = v * 1024.0;
    return v >= 0.0 && v < 4194304.0 && s == (double)(long long)s;
  };
  const unsigned u = user[i];
  if (!(ok(cpus[i]) && ok(mem[i]) && ok(gpus[i])) || seg_end[u] - seg_start[u] >= (1u << 20)) user_safe[u] = 0u;
}

// posB[slot] = position of the slot in per-user order; act[pos] = slot is a running task
__global__ void __launch_bounds__(256) rebal_invert_perm(const uint32_t* __restrict__ permB, unsigned n, unsigned R,
                                                         uint32_t* __restrict__ posB, uint8_t* __restrict__ act) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned s = permB[i];
  posB[s] = i;
  act[i] = s < R ? 1 : 0;
}

__global__ void __launch_bounds__(256) rebal_host_keys(const uint32_t* __restrict__ host, unsigned n, uint64_t* __restrict__ key) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) key[i] = host[i];
}
__global__ void __launch_bounds__(256) rebal_host_sizes(const uint32_t* __restrict__ hstart, const uint32_t* __restrict__ hend, unsigned H,
                                                        uint32_t* __restrict__ size) {
  const unsigned h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h < H) size[h] = hend[h] - hstart[h];
}
__global__ void __launch_bounds__(256) rebal_host_bounds(const uint32_t* __restrict__ hperm, const uint32_t* __restrict__ host,
                                                         unsigned n, uint32_t* __restrict__ hstart, uint32_t* __restrict__ hend) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned h = host[hperm[i]];
  if (i == 0 || host[hperm[i - 1]] != h) hstart[h] = i;
  if (i == n - 1 || host[hperm[i + 1]] != h) hend[h] = i + 1;
}

// host-ordered mirrors of the running slots (built once per run, after both orders are known)
__global__ void __launch_bounds__(256) rebal_host_mirror(const uint32_t* __restrict__ hperm, const uint32_t* __restrict__ posB,
                                                         const uint32_t* __restrict__ slot_user, const double* __restrict__ cpus,
                                                         const double* __restrict__ mem, const double* __restrict__ gpus, unsigned R,
                                                         uint32_t* __restrict__ h_pb, uint32_t* __restrict__ h_user,
                                                         double* __restrict__ h_cpus, double* __restrict__ h_mem, double* __restrict__ h_gpus,
                                                         uint8_t* __restrict__ h_act, uint32_t* __restrict__ hidx) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const unsigned slot = hperm[i], pb = posB[slot];
  h_pb[i] = pb;
  h_user[i] = slot_user[slot];
  h_cpus[i] = cpus[slot];
  h_mem[i] = mem[slot];
  h_gpus[i] = gpus[slot];
  h_act[i] = 1;
  hidx[pb] = i;
}
__global__ void __launch_bounds__(256) rebal_mirror_dru(const uint32_t* __restrict__ h_pb, const double* __restrict__ dru, unsigned R,
                                                        double* __restrict__ h_dru) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R) h_dru[i] = dru[h_pb[i]];
}

// Re-scoring after a decision (dru.clj:128-144 recomputes the changed users only).  rebal_apply lists the users whose active
// set changed (in.chg) and cuts their segments of the per-user order into tiles of RB_RS_TILE slots (in.chg_tile: first tile of
// each user, in.ctl->n_tiles in all): a heavy user's hundred thousand slots are scanned by many workgroups, a light user's by
// one.  Masked prefix sums with exactness tracking in three steps (tile-local scans, per-user scan of the tile totals, carry +
// DRU) — any association is the left-to-right sum when no addition rounded; a user where one did is redone sequentially
// (rebal_rs_fix, as rebal_fix_inexact does).  The DRUs also go to the host-ordered mirror.
#ifdef __HIP_EMU__
constexpr int RB_RS_TILE = 256, RB_RS_GRID = 4, RB_RS_USERS = 4;  // few fibers per launch; small tiles = many-tile users in small tests
#else
constexpr int RB_RS_TILE = 1024, RB_RS_GRID = 256, RB_RS_USERS = 64;
#endif
static __device__ __forceinline__ unsigned rebal_rs_user_of_tile(const RebalIn& in, unsigned n_chg, unsigned tile) {
  unsigned lo = 0, hi = n_chg;  // last x with chg_tile[x] <= tile
  while (hi - lo > 1) {
    const unsigned mid = (lo + hi) >> 1;
    if (in.chg_tile[mid] <= tile)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}
static __device__ __forceinline__ double rebal_dru_of(const RebalIn& in, unsigned u, const SumU4& sm) {
  if (in.dru_mode == 1) return sm.gpus / in.div_gpus[u];
  const double a = sm.mem / in.div_mem[u], b = sm.cpus / in.div_cpus[u];
  return a > b ? a : b;
}
// step 1: tile-local masked inclusive scans -> pre (without the carry), tile totals
__global__ void __launch_bounds__(RB_RS_TILE) rebal_rs_local(RebalIn in) {
  __shared__ SumU4 s_tot[RB_RS_TILE / COOK_WAVE];
  const unsigned n_chg = in.ctl->n_changed, n_tiles = in.ctl->n_tiles;
  if (n_chg == 0u) return;
  const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
  for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const unsigned x = rebal_rs_user_of_tile(in, n_chg, tile), u = in.chg[x];
    const unsigned s1 = in.seg_end[u], i = in.seg_start[u] + (tile - in.chg_tile[x]) * RB_RS_TILE + tid;
    const bool in_seg = i < s1;
    const SumU4 v = wave_incl_scan_u4((in_seg && in.act[i]) ? in.s_use[i] : SumU4::zero());
    if (lane == COOK_WAVE - 1) s_tot[w] = v;
    __syncthreads();
    SumU4 pre = SumU4::zero();
    for (unsigned k = 0; k < w; ++k) pre = combine(pre, s_tot[k]);
    const SumU4 t = combine(pre, v);
    if (in_seg) in.pre_w[i] = t;
    if (tid == RB_RS_TILE - 1) in.tile_agg[tile] = t;  // the tile's total (zeros beyond the segment)
    if (t.bad) in.chg_bad[x] = 1u;
    __syncthreads();
  }
}
// step 2: per changed user, exclusive scan of its tile totals (one wave; a user has few tiles)
__global__ void __launch_bounds__(COOK_WAVE) rebal_rs_carry(RebalIn in) {
  const unsigned n_chg = in.ctl->n_changed;
  const unsigned lane = lane_id();
  for (unsigned x = blockIdx.x; x < n_chg; x += gridDim.x) {
    const unsigned t0 = in.chg_tile[x], t1 = in.chg_tile[x + 1];
    SumU4 carry = SumU4::zero();
    unsigned bad = 0u;
    for (unsigned base = t0; base < t1; base += COOK_WAVE) {
      const unsigned t = base + lane;
      const SumU4 v = t < t1 ? in.tile_agg[t] : SumU4::zero();
      const SumU4 inc = combine(carry, wave_incl_scan_u4(v));
      // exclusive = inclusive of the previous lane
      SumU4 ex = shfl_up_v(inc, 1);
      if (lane == 0) ex = carry;
      if (t < t1) in.tile_carry[t] = ex;
      bad |= inc.bad;
      carry = wave_bcast_u4(inc, COOK_WAVE - 1);
    }
    if (__any(bad != 0u) && lane == 0) in.chg_bad[x] = 1u;
  }
}
// step 3: carry + local prefix -> pre, DRU, host-ordered mirror
__global__ void __launch_bounds__(RB_RS_TILE) rebal_rs_finish(RebalIn in) {
  const unsigned n_chg = in.ctl->n_changed, n_tiles = in.ctl->n_tiles;
  if (n_chg == 0u) return;
  const unsigned tid = threadIdx.x;
  for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const unsigned x = rebal_rs_user_of_tile(in, n_chg, tile), u = in.chg[x];
    const unsigned i = in.seg_start[u] + (tile - in.chg_tile[x]) * RB_RS_TILE + tid;
    if (i >= in.seg_end[u]) continue;
    const SumU4 t = combine(in.tile_carry[tile], in.pre_w[i]);
    if (t.bad) in.chg_bad[x] = 1u;
    in.pre_w[i] = SumU4{t.count, t.cpus, t.mem, t.gpus, 0u};
    const double d = rebal_dru_of(in, u, t);
    in.dru_w[i] = d;
    const unsigned hi = in.hidx[i];
    if (hi != 0xFFFFFFFFu) in.h_dru[hi] = d;
  }
}
// step 4: users where an addition rounded: left to right, exactly as the reference's reductions, then their DRUs again
__global__ void __launch_bounds__(256) rebal_rs_fix(RebalIn in) {
  const unsigned n_chg = in.ctl->n_changed;
  for (unsigned x = blockIdx.x; x < n_chg; x += gridDim.x) {
    if (!in.chg_bad[x]) continue;  // block-uniform
    const unsigned u = in.chg[x], s0 = in.seg_start[u], s1 = in.seg_end[u];
    if (threadIdx.x == 0) {
      double c = 0.0, cp = 0.0, m = 0.0, g = 0.0;
      bool first = true;
      for (unsigned i = s0; i < s1; ++i) {
        if (in.act[i]) {
          const SumU4 v = in.s_use[i];
          if (first) {
            c = v.count, cp = v.cpus, m = v.mem, g = v.gpus;
            first = false;
          } else {
            c += v.count, cp += v.cpus, m += v.mem, g += v.gpus;
          }
        }
        in.pre_w[i] = SumU4{c, cp, m, g, 0u};
      }
    }
    __syncthreads();
    for (unsigned i = s0 + threadIdx.x; i < s1; i += blockDim.x) {
      const double d = rebal_dru_of(in, u, in.pre_w[i]);
      in.dru_w[i] = d;
      const unsigned hi = in.hidx[i];
      if (hi != 0xFFFFFFFFu) in.h_dru[hi] = d;
    }
    __syncthreads();
  }
}

// ---- per pending job: quota test, pending DRU, group cohosts -------------------------------------------------------------
__global__ void __launch_bounds__(COOK_WAVE) rebal_job_prep(RebalIn in, unsigned pj) {
  const unsigned lane = lane_id();
  RebalJob jb;
  jb.active = 0;
  jb.pj = pj;
  jb.us = 0;
  jb.below = 0;
  jb.has_gpus = 0;
  jb.gtype = jb.gkey = jb.n_co = 0;
  jb.minim = jb.maxfreq = 0;
  jb.pad0 = jb.pad1 = 0;
  jb.pdru = jb.c = jb.m = jb.g = 0.0;
  if (lane == 0) in.ctl->n_changed = 0u;  // nothing to re-score unless rebal_apply takes a decision
  if (in.ctl->remaining <= 0) {
    if (lane == 0) *in.job = jb;
    return;
  }
  const unsigned us = in.j_user[pj];
  const double jc = in.j_cpus[pj], jm = in.j_mem[pj], jg = in.j_gpus ? in.j_gpus[pj] : 0.0;
  const unsigned s0 = in.seg_start[us], s1 = in.seg_end[us];
  const unsigned ppos = in.posB[in.R + pj];
  // rebalancer.clj:210-220: usage of (conj running-jobs job) = the job first, then the user's tasks in order
  const SumU4 seed{1.0, jc, jm, jg, 0u};
  SumU4 fu = seed;
  int last = -1;
  auto okv = [](double v) {
    const double s = v * 1024.0;
    return v >= 0.0 && v < 4194304.0 && s == (double)(long long)s;
  };
  if (in.user_safe[us] && okv(jc) && okv(jm) && okv(jg)) {
    // every partial sum is exact whatever the association: the job + the scan's total over the user's active tasks
    if (s1 > s0) {
      const SumU4 t = in.pre[s1 - 1];
      fu = SumU4{1.0 + t.count, jc + t.cpus, jm + t.mem, jg + t.gpus, 0u};
    }
    // nearest active slot before the job's own: almost always in the first chunk (only preempted tasks are inactive)
    for (unsigned hi = ppos; hi > s0 && last < 0;) {
      const unsigned lo = hi - s0 > COOK_WAVE ? hi - COOK_WAVE : s0;
      const unsigned i = lo + lane;
      const bool a = i < hi && in.act[i] != 0;
      const unsigned long long mk = __ballot(a);
      if (mk != 0ull) last = (int)(lo + 63u - (unsigned)__clzll((unsigned long long)mk));
      hi = lo;
    }
  } else {
    SumU4 carry = seed;
    unsigned bad = 0u;
    for (unsigned base = s0; base < s1; base += COOK_WAVE) {
      const unsigned i = base + lane;
      const bool a = i < s1 && in.act[i] != 0;
      const SumU4 x = a ? in.s_use[i] : SumU4::zero();
      const SumU4 t = combine(carry, wave_incl_scan_u4(x));
      bad |= t.bad;
      if (a && i < ppos && (int)i > last) last = (int)i;
      carry = wave_bcast_u4(t, COOK_WAVE - 1);
    }
    bad = __any(bad != 0u) ? 1u : 0u;
    for (int d = 32; d >= 1; d >>= 1) {
      const int o = __shfl_xor(last, d, COOK_WAVE);
      last = o > last ? o : last;
    }
    fu = carry;
    if (bad) {  // a partial sum rounded: redo left to right like the reference (all lanes compute the same thing)
      double c = 1.0, cp = jc, m = jm, g = jg;
      for (unsigned i = s0; i < s1; ++i)
        if (in.act[i]) {
          const SumU4 x = in.s_use[i];
          c += x.count, cp += x.cpus, m += x.mem, g += x.gpus;
        }
      fu = SumU4{c, cp, m, g, 0u};
    }
  }
  jb.active = 1;
  jb.us = us;
  jb.below = below_quota4(in.q_count[us], in.q_cpus[us], in.q_mem[us], in.q_gpus[us], Usage4{fu.count, fu.cpus, fu.mem, fu.gpus}) ? 1u : 0u;
  jb.has_gpus = (in.j_gpus && jg > 0) ? 1u : 0u;
  jb.c = jc;
  jb.m = jm;
  jb.g = jg;
  // rebalancer.clj:157-208: nearest task at or before the synthetic pending task in the user's order
  const double near = last >= 0 ? in.dru[last] : 0.0;
  if (in.dru_mode == 1) {
    jb.pdru = near + jg / in.div_gpus[us];
  } else {
    const double a = near + jm / in.div_mem[us], b = near + jc / in.div_cpus[us];
    jb.pdru = a > b ? a : b;
  }
  // group cohosts: every task preempted so far this cycle ++ the group's running cotasks (constraints.clj:680-697)
  const unsigned g = (in.j_group && in.G) ? in.j_group[pj] : 0xFFFFFFFFu;
  if (g != 0xFFFFFFFFu && in.g_type[g] != 0) {
    jb.gtype = in.g_type[g];
    jb.gkey = jb.gtype == 1 ? 0xFFFFFFFFu : in.g_attr_key[g];
    const unsigned n_pre = in.ctl->n_pre_hosts;
    const unsigned r0 = in.g_run_off ? in.g_run_off[g] : 0u, r1 = in.g_run_off ? in.g_run_off[g + 1] : 0u;
    const unsigned n_co = n_pre + (r1 - r0);
    jb.n_co = n_co;
    for (unsigned x = lane; x < n_co; x += COOK_WAVE) {
      const unsigned h = x < n_pre ? in.pre_hosts[x] : in.g_run_host[r0 + (x - n_pre)];
      const int row = h < in.H ? in.row_of_host[h] : -1;
      st_agent(&in.co_val[x], rebal_attr(in, row, jb.gkey));
    }
    __threadfence();
    wave_sync();
    if (jb.gtype >= 2 && n_co) {  // frequencies of the attribute over the cohosts (nil = 0 is a legal value)
      unsigned distinct = 0, mn = 0xFFFFFFFFu, mx = 0;
      for (unsigned x = lane; x < n_co; x += COOK_WAVE) {
        const unsigned v = ld_agent(&in.co_val[x]);
        unsigned cnt = 0;
        bool first = true;
        for (unsigned y = 0; y < n_co; ++y)
          if (ld_agent(&in.co_val[y]) == v) {
            ++cnt;
            if (y < x) first = false;
          }
        if (first) {
          ++distinct;
          mn = cnt < mn ? cnt : mn;
          mx = cnt > mx ? cnt : mx;
        }
      }
      for (int d = 32; d >= 1; d >>= 1) {
        distinct += __shfl_xor(distinct, d, COOK_WAVE);
        const unsigned a = __shfl_xor(mn, d, COOK_WAVE), b = __shfl_xor(mx, d, COOK_WAVE);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
      }
      jb.minim = ((unsigned)(in.g_min[g] > 0 ? in.g_min[g] : 0) > distinct) ? 0 : (int)mn;
      jb.maxfreq = (int)mx;
    }
  }
  if (lane == 0) {
    *in.job = jb;
    if (in.pending_dru) in.pending_dru[pj] = jb.pdru;
  }
}

// job constraints of the rebalancer on the attribute map of row r (r < 0: nil map); constraints.clj:459-466 through the
// 3-arity evaluate, i.e. with NO tasks assigned on the VM
static __device__ __forceinline__ bool rebal_job_constraints(const RebalIn& in, unsigned k, int r, double jg) {
  if (r >= 0 && in.j_novel_off)  // novel-host: (get nil "HOSTNAME") is nil, never in the set
    for (unsigned x = in.j_novel_off[k]; x < in.j_novel_off[k + 1]; ++x)
      if (in.j_novel_host[x] == in.a_host[r]) return false;
  const bool k8s = r >= 0 && in.a_k8s && in.a_k8s[r];
  if (k8s) {  // gpu-host, constraints.clj:122-157
    if (jg > 0) {
      const double avail = map_get_dev(in.a_gpu_model, in.a_gpu_count, in.a_gpu_slots, (unsigned)r, in.j_gpu_model ? in.j_gpu_model[k] : 0u);
      if (!(avail == jg)) return false;
    } else if (map_count_dev(in.a_gpu_model, in.a_gpu_slots, (unsigned)r) != 0u) {
      return false;
    }
  } else if (!(jg == 0)) {
    return false;
  }
  if (in.j_disk_req && in.j_disk_req[k] >= 0 && k8s) {  // disk-host, constraints.clj:164-199
    const double space = map_get_dev(in.a_disk_type, in.a_disk_space, in.a_disk_slots, (unsigned)r, in.j_disk_type[k]);
    if (!(space >= in.j_disk_req[k])) return false;
  }
  if (in.j_eq_off)  // user-defined EQUALS: (= pattern (get nil attribute)) is false
    for (unsigned x = in.j_eq_off[k]; x < in.j_eq_off[k + 1]; ++x)
      if (rebal_attr(in, r, in.j_eq_key[x]) != in.j_eq_val[x]) return false;
  if (r >= 0 && in.j_est_end && in.j_est_end[k] != 0 && in.a_host_start && in.a_host_start[r] >= 0) {
    const long long death = 1000ll * in.a_host_start[r] + 60ll * 1000ll * in.host_lifetime_mins;
    if (!(in.j_est_end[k] < death)) return false;
  }
  if (in.j_ckpt && in.j_ckpt[k] != 0) {  // checkpoint-locality
    const unsigned loc = (r >= 0 && in.a_location) ? in.a_location[r] : 0u;
    if (loc != in.j_ckpt[k]) return false;
  }
  return true;
}

// group constraint (constraints.clj:586-644) against the cohost values staged by rebal_job_prep; whole wave cooperates
static __device__ __forceinline__ bool rebal_group_constraint(const RebalIn& in, const RebalJob& jb, int row) {
  const unsigned lane = lane_id();
  const unsigned target = rebal_attr(in, row, jb.gkey);
  if (jb.gtype == 1 && target == 0u) return false;  // unique: the target hostname must be present
  unsigned cnt = 0;
  for (unsigned x = lane; x < jb.n_co; x += COOK_WAVE) cnt += in.co_val[x] == target ? 1u : 0u;
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, COOK_WAVE);
  if (jb.gtype == 1) return cnt == 0u;
  if (jb.n_co == 0u) return true;
  if (jb.gtype == 2) return cnt == 0u || jb.minim == jb.maxfreq || (int)cnt < jb.maxfreq;
  return cnt != 0u;  // attribute-equals
}

// first index i in [0, n) with host(x_pj[i]) >= h
static __device__ __forceinline__ unsigned rebal_x_lower(const RebalIn& in, unsigned n, unsigned h) {
  unsigned lo = 0, hi = n;
  while (lo < hi) {
    const unsigned mid = (lo + hi) >> 1;
    if (in.x_host[in.x_pj[mid]] < h)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// ---- per host: candidates, priority order, prefix aggregates, best feasible prefix ------------------------------------------
__global__ void __launch_bounds__(COOK_WAVE* RB_WAVES) rebal_decide(RebalIn in) {
  __shared__ double l_dru[RB_WAVES][RB_CAP], l_cpus[RB_WAVES][RB_CAP], l_mem[RB_WAVES][RB_CAP], l_gpus[RB_WAVES][RB_CAP];
  __shared__ uint32_t l_posB[RB_WAVES][RB_CAP], l_slot[RB_WAVES][RB_CAP], l_ord[RB_WAVES][RB_CAP];
  const RebalJob jb = *in.job;
  if (!jb.active) return;
  const unsigned lane = lane_id(), w = wave_id();
  const unsigned h = blockIdx.x * RB_WAVES + w;
  if (h >= in.H) return;
  const unsigned hs = in.hstart[h], n_seg = in.hend[h] - hs;
  // jobs placed on this host earlier in the cycle = x_pj[xs, xe): the placed list is sorted by host and rebal_apply keeps, per
  // host, the number of entries before it (two dependent binary searches per wave cost more than the rest of the kernel)
  const unsigned xs = in.x_before[h], xe = xs + in.x_cnt[h];
  const unsigned n = n_seg + (xe - xs);
  const bool sp = in.has_spare[h] != 0;
  if (lane == 0) in.hres_key[h] = 0ull;
  if (n == 0 && !sp) return;
  // the host's region of the scratch arrays starts after the running tasks of the hosts before it and the jobs placed on them.
  // (hstart is only meaningful for hosts that HAVE running tasks: an empty host's 0 made its region collide with another host's.)
  const unsigned base = in.hbase[h] + xs;
  const bool big = n > (unsigned)RB_CAP;
  double *c_dru = big ? in.gs_dru + base : l_dru[w], *c_cpus = big ? in.gs_cpus + base : l_cpus[w];
  double *c_mem = big ? in.gs_mem + base : l_mem[w], *c_gpus = big ? in.gs_gpus + base : l_gpus[w];
  uint32_t *c_posB = big ? in.gs_posB + base : l_posB[w], *c_slot = big ? in.gs_slot + base : l_slot[w];
  uint32_t* c_ord = big ? in.gs_ord + base : l_ord[w];
  // ---- pass A: filter (rebalancer.clj:339-349) + the host's last scored task in priority-map order (:369-375) ----------
  unsigned n_c = 0;
  double last_d = 0.0;
  unsigned last_pb = 0, last_slot = 0xFFFFFFFFu;
  for (unsigned t0 = 0; t0 < n; t0 += COOK_WAVE) {
    const unsigned t = t0 + lane;
    const bool valid = t < n;
    unsigned slot = 0, pb = 0, usr = 0;
    bool a = false;
    double d = 0.0;
    const bool mirrored = valid && t < n_seg;  // a running slot: its columns lie at hs + t of the host-ordered mirrors
    if (mirrored) {
      slot = in.hperm[hs + t];
      pb = in.h_pb[hs + t];
      a = in.h_act[hs + t] != 0;
      usr = in.h_user[hs + t];
      if (a) d = in.h_dru[hs + t];
    } else if (valid) {  // a job placed earlier in this cycle
      slot = in.R + in.x_pj[xs + (t - n_seg)];
      pb = in.posB[slot];
      a = in.act[pb] != 0;
      usr = in.slot_user[slot];
      if (a) d = in.dru[pb];
    }
    if (a && (last_slot == 0xFFFFFFFFu || d < last_d || (d == last_d && pb > last_pb))) {
      last_d = d;
      last_pb = pb;
      last_slot = slot;
    }
    const bool cand = a && (jb.below || usr == jb.us) && !(d < in.safe_dru) && (d - jb.pdru > in.min_diff);
    const unsigned long long mk = __ballot(cand);
    if (cand) {
      const unsigned idx = n_c + (unsigned)__popcll(mk & lanemask_lt());
      st_agent(&c_dru[idx], d);
      st_agent(&c_posB[idx], pb);
      st_agent(&c_slot[idx], slot);
      const double xc = mirrored ? in.h_cpus[hs + t] : in.slot_cpus[slot], xm = mirrored ? in.h_mem[hs + t] : in.slot_mem[slot];
      const double xg = mirrored ? in.h_gpus[hs + t] : in.slot_gpus[slot];
      st_agent(&c_cpus[idx], xc);
      st_agent(&c_mem[idx], xm);
      st_agent(&c_gpus[idx], xg);
    }
    n_c += (unsigned)__popcll(mk);
  }
  if (n_c == 0 && !sp) return;  // nothing to preempt and nothing spare: no prefix exists (wave-uniform)
  // the LAST scored task of the host decides which slave id (hence attribute map) the host resolves to
  for (int dd = 32; dd >= 1; dd >>= 1) {
    const double od = __shfl_xor(last_d, dd, COOK_WAVE);
    const unsigned opb = __shfl_xor(last_pb, dd, COOK_WAVE), osl = __shfl_xor(last_slot, dd, COOK_WAVE);
    if (osl != 0xFFFFFFFFu && (last_slot == 0xFFFFFFFFu || od < last_d || (od == last_d && opb > last_pb))) {
      last_d = od;
      last_pb = opb;
      last_slot = osl;
    }
  }
  bool known = false;
  if (last_slot != 0xFFFFFFFFu)
    known = last_slot < in.R ? (in.attrs_cached ? in.attrs_cached[last_slot] != 0 : true) : in.x_known[last_slot - in.R] != 0;
  const int row = known ? in.row_of_host[h] : -1;
  if (!rebal_job_constraints(in, jb.pj, row, jb.g)) return;
  if (jb.gtype && !rebal_group_constraint(in, jb, row)) return;
  if (big) __threadfence();
  wave_sync();
  // ---- pass B: priority-map order inside the host = (dru desc, position in B asc), by counting -----------------------------
  for (unsigned i0 = 0; i0 < n_c; i0 += COOK_WAVE) {
    const unsigned i = i0 + lane;
    const bool vi = i < n_c;
    const double di = vi ? ld_agent(&c_dru[i]) : 0.0;
    const unsigned pi = vi ? ld_agent(&c_posB[i]) : 0u;
    unsigned r = 0;
    for (unsigned j = 0; j < n_c; ++j) {
      const double dj = ld_agent(&c_dru[j]);
      const unsigned pjx = ld_agent(&c_posB[j]);
      r += (dj > di || (dj == di && pjx < pi)) ? 1u : 0u;
    }
    if (vi) {
      st_agent(&c_ord[r], i);
      in.srt_slot[base + r] = ld_agent(&c_slot[i]);
    }
  }
  if (big) __threadfence();
  wave_sync();
  // ---- pass C: prefix aggregates seeded with the spare resources (rebalancer.clj:384-403), best feasible prefix ------------
  const double jc = jb.c, jm = jb.m, jg = jb.g;
  const bool need_g = jb.has_gpus != 0;
  SumU4 carry = SumU4::zero();
  if (sp) carry = SumU4{0.0, 0.0 + in.spare_c[h], 0.0 + in.spare_m[h], 0.0 + in.spare_g[h], 0u};
  const SumU4 seed = carry;
  // per-lane best: key (f64_key(dru), len) lexicographic max; len 0 = the spare pseudo-entry alone
  unsigned long long bk = 0ull;
  unsigned bl = 0;
  double bd = 0.0, bc = 0.0, bm = 0.0, bg = 0.0;
  const double DMAXV = 1.7976931348623157e308;
  if (lane == 0 && sp && seed.mem >= jm && seed.cpus >= jc && (need_g ? seed.gpus >= jg : true)) {
    bk = f64_key(DMAXV);
    bd = DMAXV;
    bc = seed.cpus;
    bm = seed.mem;
    bg = seed.gpus;
  }
  unsigned bad = 0u;
  for (unsigned k0 = 0; k0 < n_c; k0 += COOK_WAVE) {
    const unsigned k = k0 + lane;
    const bool vk = k < n_c;
    const unsigned i = vk ? ld_agent(&c_ord[k]) : 0u;
    SumU4 x = SumU4::zero();
    double d = 0.0;
    if (vk) {
      x = SumU4{0.0, ld_agent(&c_cpus[i]), ld_agent(&c_mem[i]), ld_agent(&c_gpus[i]), 0u};
      d = ld_agent(&c_dru[i]);
    }
    const SumU4 t = combine(carry, wave_incl_scan_u4(x));
    if (vk) bad |= t.bad;
    const bool enough = vk && t.mem >= jm && t.cpus >= jc && (need_g ? t.gpus >= jg : true) && d >= 0.0;
    if (enough) {
      const unsigned long long key = f64_key(d);
      if (key >= bk) {  // later prefix wins ties (max-key, rebalancer.clj:404)
        bk = key;
        bl = k + 1;
        bd = d;
        bc = t.cpus;
        bm = t.mem;
        bg = t.gpus;
      }
    }
    carry = wave_bcast_u4(t, COOK_WAVE - 1);
  }
  if (__any(bad != 0u)) {  // a partial sum rounded: left to right, exactly as the reference's reductions
    bk = 0ull;
    bl = 0;
    if (lane == 0) {
      double ac = seed.cpus, am = seed.mem, ag = seed.gpus;
      if (sp && am >= jm && ac >= jc && (need_g ? ag >= jg : true)) {
        bk = f64_key(DMAXV);
        bd = DMAXV;
        bc = ac, bm = am, bg = ag;
      }
      for (unsigned k = 0; k < n_c; ++k) {
        const unsigned i = ld_agent(&c_ord[k]);
        ac += ld_agent(&c_cpus[i]);
        am += ld_agent(&c_mem[i]);
        ag += ld_agent(&c_gpus[i]);
        const double d = ld_agent(&c_dru[i]);
        if (am >= jm && ac >= jc && (need_g ? ag >= jg : true) && d >= 0.0 && f64_key(d) >= bk) {
          bk = f64_key(d);
          bl = k + 1;
          bd = d;
          bc = ac, bm = am, bg = ag;
        }
      }
    }
  }
  // wave arg-max of (bk, bl)
  unsigned long long mk = bk;
  unsigned ml = bl;
  for (int dd = 32; dd >= 1; dd >>= 1) {
    const unsigned long long ok = __shfl_xor(mk, dd, COOK_WAVE);
    const unsigned ol = __shfl_xor(ml, dd, COOK_WAVE);
    if (ok > mk || (ok == mk && ol > ml)) {
      mk = ok;
      ml = ol;
    }
  }
  if (mk != 0ull && bk == mk && bl == ml) {  // exactly one lane holds (mk, ml): prefix lengths are distinct per lane
    in.hres_key[h] = mk;
    in.hres_len[h] = ml;
    in.hres_base[h] = base;
    in.hres_dru[h] = bd;
    in.hres_c[h] = bc;
    in.hres_m[h] = bm;
    in.hres_g[h] = bg;
  }
}

// ---- arg-max over hosts + next-state (rebalancer.clj:270-309, 404) -----------------------------------------------------------
constexpr int RB_APPLY_THREADS = 1024;
__global__ void __launch_bounds__(RB_APPLY_THREADS) rebal_apply(RebalIn in) {
  __shared__ unsigned long long s_key[RB_APPLY_THREADS];
  __shared__ unsigned s_host[RB_APPLY_THREADS];
  const RebalJob jb = *in.job;
  if (!jb.active) return;
  const unsigned tid = threadIdx.x;
  unsigned long long bk = 0ull;
  unsigned bh = 0;
  for (unsigned h = tid; h < in.H; h += RB_APPLY_THREADS) {
    const unsigned long long k = in.hres_key[h];
    if (k != 0ull && k >= bk) {  // hosts ascend with h: the later host wins ties
      bk = k;
      bh = h;
    }
  }
  s_key[tid] = bk;
  s_host[tid] = bh;
  __syncthreads();
  for (unsigned s = RB_APPLY_THREADS / 2; s >= 1; s >>= 1) {
    if (tid < s) {
      const unsigned long long ok = s_key[tid + s];
      const unsigned oh = s_host[tid + s];
      if (ok > s_key[tid] || (ok == s_key[tid] && ok != 0ull && oh > s_host[tid])) {
        s_key[tid] = ok;
        s_host[tid] = oh;
      }
    }
    __syncthreads();
  }
  if (s_key[0] == 0ull) return;  // no host can take the job: no decision, state unchanged (rebalancer.clj:455-458)
  const unsigned h = s_host[0];
  for (unsigned hh = h + 1 + tid; hh < in.H; hh += RB_APPLY_THREADS) in.x_before[hh] += 1u;  // the job is about to join x_pj at host h
  if (tid != 0) return;
  in.x_cnt[h] += 1u;
  const unsigned len = in.hres_len[h], base = in.hres_base[h];
  RebalCtl c = *in.ctl;
  cook_preemption d;
  d.pending_index = jb.pj;
  d.host = h;
  d.dru = in.hres_dru[h];
  d.cpus = in.hres_c[h];
  d.mem = in.hres_m[h];
  d.gpus = in.hres_g[h];
  d.task_off = c.np;
  d.task_n = len;
  in.decisions[c.nd++] = d;
  bool first_known = false;
  unsigned n_chg = 0, n_tiles = 0;
  const unsigned stamp = c.nd;  // decisions are numbered from 1 here (nd was just incremented): 0 = never listed
  auto changed = [&](unsigned u) {
    if (in.chg_mark[u] == stamp) return;  // listed already by this decision
    in.chg_mark[u] = stamp;
    in.chg[n_chg] = u;
    in.chg_bad[n_chg] = 0u;
    in.chg_tile[n_chg] = n_tiles;
    n_tiles += (in.seg_end[u] - in.seg_start[u] + RB_RS_TILE - 1) / RB_RS_TILE;
    ++n_chg;
  };
  changed(jb.us);
  for (unsigned k = 0; k < len; ++k) {
    const unsigned slot = in.srt_slot[base + k];
    const unsigned pbk = in.posB[slot];
    in.act[pbk] = 0;
    if (slot < in.R) in.h_act[in.hidx[pbk]] = 0;
    changed(in.slot_user[slot]);
    in.preempted[c.np++] = slot < in.R ? slot : 0xFFFFFFFFu;  // a task placed this cycle is reported as NONE (rebalancer.clj:529)
    const bool known = slot < in.R ? (in.attrs_cached ? in.attrs_cached[slot] != 0 : true) : in.x_known[slot - in.R] != 0;
    if (k == 0) first_known = known;
    if (known) in.pre_hosts[c.n_pre_hosts++] = h;
  }
  // the job becomes a task of its user on that host, carrying the slave id of the first preempted task (:279-281)
  in.act[in.posB[in.R + jb.pj]] = 1;
  in.x_host[jb.pj] = h;
  in.x_known[jb.pj] = (len > 0 && first_known) ? 1 : 0;
  unsigned pos = c.n_x;
  while (pos > 0 && in.x_host[in.x_pj[pos - 1]] > h) {
    in.x_pj[pos] = in.x_pj[pos - 1];
    --pos;
  }
  in.x_pj[pos] = jb.pj;
  c.n_x += 1;
  in.spare_c[h] = d.cpus - jb.c;  // rebalancer.clj:302-305
  in.spare_m[h] = d.mem - jb.m;
  in.spare_g[h] = d.gpus - jb.g;
  in.has_spare[h] = 1;
  c.remaining -= 1;
  in.chg_tile[n_chg] = n_tiles;
  c.n_changed = n_chg;
  c.n_tiles = n_tiles;
  *in.ctl = c;
}

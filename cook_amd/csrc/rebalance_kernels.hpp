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
#ifndef RB_WAVES_N
#define RB_WAVES_N 8  // (measured on the 50k-host sweep: 4 waves / 32 pairs 7.84 ms, 8 / 64 6.91, 16 / 128 6.90, 8 / 256 7.92)
#endif
#ifndef RB_PAIRS_N
#define RB_PAIRS_N 64
#endif
constexpr int RB_WAVES = RB_WAVES_N;  // waves of a decide block
constexpr int RB_PAIRS = RB_PAIRS_N;  // pairs of hosts a decide block looks at (a thread each), the ones to evaluate shared by its waves

struct RebalCtl {
  int remaining;        // max-preemption budget left (rebalancer.clj:442)
  unsigned nd, np;      // decisions / preempted tasks emitted
  unsigned n_pre_hosts; // hosts of the tasks preempted so far whose attribute map is known (constraints.clj:686-689)
  unsigned n_x;         // jobs placed so far this cycle (x_pj)
  unsigned n_changed;   // users whose active set the last decision changed (chg[])
  unsigned n_tiles;     // tiles of RB_RS_TILE slots their segments are cut into (chg_tile[])
  unsigned n_big;       // hosts listed in big_list
  unsigned max_items;   // running tasks + placed jobs of the fullest host so far (the host decides from it whether rebal_decide_big
                        // can have work: read back with the budget)
  unsigned n_delta;     // slots whose active flag the last decision flipped (dl_pos / dl_sign)
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
  uint32_t* dl_pos;               // [S + 1] B positions whose active flag the last decision flipped ...
  int32_t* dl_sign;               // [S + 1] ... -1: a preempted task left, +1: the placed job joined (rebal_rs_delta)
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
  uint32_t* x_pj;       // [P] placed jobs in placement order
  uint32_t* x_host;     // [P] by pj
  uint8_t* x_known;     // [P] by pj: the placed task carries a slave id whose attributes are cached
  uint32_t *x_head, *x_cnt;    // [H] newest placed job on the host (chain through x_next), number of them
  uint32_t* x_next;            // [P] by pj: the next older placed job on the same host
  uint32_t* big_list;          // [H] hosts that hold (or held) more than 64 items: rebal_decide_big's work list
  uint32_t* pre_hosts;  // [S]
  uint32_t* co_val;     // cohost values of the current job's group
  // per-host results of rebal_decide
  unsigned long long* hres_key;  // f64_key(dru) of the host's best prefix, 0 = none
  // the best host of every rebal_decide workgroup (greatest key, the later host on ties): what rebal_apply scans instead of all hosts
  unsigned long long* blk_key;
  uint32_t* blk_host;
  unsigned n_blk;
  // pruning of the per-host evaluation (rebal_decide's two phases): an upper bound on the key a host can reach
  unsigned long long* hmax_key;   // [H] f64_key of a value >= the DRU of every active item of the host (only ever raised; all ones: a job placed this cycle sits there)
  const uint32_t* h_host;         // [R] host of a position in host order
  unsigned long long* best_key;   // [1] greatest key a host evaluated so far for the current job reached (0 none; rebal_apply clears it)
  unsigned long long thr_key;     // hosts whose bound is at least this are evaluated first (0: every host in one phase)
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

// inclusive wave scan of usage vectors whose count field is not needed, by DPP row shifts (common.hpp scan_fetch): `steps` = 4, 5 or 6
// covers the first 16, 32 or 64 lanes (wave-uniform)
// SAFE: every value that can meet in a sum is a non-negative multiple of 2^-10 below 2^22 (rebal_user_safe over the slots, the spare
// resources checked at staging; differences of such values stay such values) and a host holds at most 64 of them: no addition can
// round, so the TwoSum exactness bit (25 fp64 instructions per combine against 3) and the left-to-right fall-back are compiled out.
template <bool SAFE>
static __device__ __forceinline__ SumU4 combine_t(const SumU4& a, const SumU4& b) {
  if (SAFE) return SumU4{a.count + b.count, a.cpus + b.cpus, a.mem + b.mem, a.gpus + b.gpus, 0u};
  return combine(a, b);
}
template <int STEP, bool SAFE = false>
static __device__ __forceinline__ SumU4 scan_step_u4(const SumU4& v) {
  SumU4 p;
  p.count = 0.0;
  p.cpus = scan_fetch_f64<STEP>(v.cpus);
  p.mem = scan_fetch_f64<STEP>(v.mem);
  p.gpus = scan_fetch_f64<STEP>(v.gpus);
  p.bad = SAFE ? 0u : (unsigned)scan_fetch_u32<STEP>((int)v.bad);
  return combine_t<SAFE>(p, v);  // lanes without a source fetched zeros: x + 0.0 == x, nothing rounds
}
template <bool SAFE = false>
static __device__ __forceinline__ SumU4 wave_incl_scan_u4_rows(SumU4 v, unsigned lanes) {
  v = scan_step_u4<0, SAFE>(v);
  v = scan_step_u4<1, SAFE>(v);
  v = scan_step_u4<2, SAFE>(v);
  v = scan_step_u4<3, SAFE>(v);
  if (lanes > 16u) v = scan_step_u4<4, SAFE>(v);
  if (lanes > 32u) v = scan_step_u4<5, SAFE>(v);
  return v;
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
// host of every position in host order, and the hosts' first bounds: the greatest DRU among a host's running tasks
__global__ void __launch_bounds__(256) rebal_host_bound_init(const uint32_t* __restrict__ hstart, const uint32_t* __restrict__ hend, unsigned H,
                                                             const double* __restrict__ h_dru, uint32_t* __restrict__ h_host,
                                                             unsigned long long* __restrict__ hmax_key) {
  const unsigned h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  unsigned long long k = 0ull;
  for (unsigned i = hstart[h]; i < hend[h]; ++i) {
    h_host[i] = h;
    const unsigned long long ki = f64_key(h_dru[i]);
    k = ki > k ? ki : k;
  }
  hmax_key[h] = k;
}
// a re-scored slot's DRU into the host-ordered mirror; the host's bound follows it upwards
static __device__ __forceinline__ void rebal_mirror_store(const RebalIn& in, unsigned hi, double d) {
  in.h_dru[hi] = d;
  const unsigned h = in.h_host[hi];
  const unsigned long long k = f64_key(d);
  if (in.hmax_key[h] < k) atomicMax(&in.hmax_key[h], k);
}

// Re-scoring after a decision (dru.clj:128-144 recomputes the changed users only).  rebal_apply lists the users whose active
// set changed (in.chg) and cuts their segments of the per-user order into tiles of RB_RS_TILE slots (in.chg_tile: first tile of
// each user, in.ctl->n_tiles in all): a heavy user's hundred thousand slots are scanned by many workgroups, a light user's by
// one.  Masked prefix sums with exactness tracking in three steps (tile-local scans, per-user scan of the tile totals, carry +
// DRU) — any association is the left-to-right sum when no addition rounded; a user where one did is redone sequentially
// (rebal_rs_fix, as rebal_fix_inexact does).  The DRUs also go to the host-ordered mirror.
// (the emulated tests: few fibers per launch; small tiles = many-tile users in small tests)
#ifndef RB_RS_GRID_N
#define RB_RS_GRID_N 256
#endif
constexpr int RB_RS_TILE = COOK_SHAPE(1024, 256), RB_RS_GRID = COOK_SHAPE(RB_RS_GRID_N, 4), RB_RS_USERS = COOK_SHAPE(64, 4);
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
__global__ void __launch_bounds__(RB_RS_TILE) rebal_rs_local(const RebalIn* __restrict__ inp) {
  const RebalIn& in = *inp;
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
    const bool single = in.chg_tile[x + 1] - in.chg_tile[x] == 1u;  // the user's only tile: no carry to wait for, finish here
    if (in_seg) {
      if (single) {
        in.pre_w[i] = SumU4{t.count, t.cpus, t.mem, t.gpus, 0u};
        const double d = rebal_dru_of(in, u, t);
        in.dru_w[i] = d;
        const unsigned hi = in.hidx[i];
        if (hi != 0xFFFFFFFFu) rebal_mirror_store(in, hi, d);
      } else {
        in.pre_w[i] = t;
      }
    }
    if (tid == RB_RS_TILE - 1) in.tile_agg[tile] = t;  // the tile's total (zeros beyond the segment)
    if (t.bad) in.chg_bad[x] = 1u;
    __syncthreads();
  }
}
// step 2: per changed user, exclusive scan of its tile totals (one wave; a user has few tiles)
__global__ void __launch_bounds__(COOK_WAVE) rebal_rs_carry(const RebalIn* __restrict__ inp) {
  const RebalIn& in = *inp;
  const unsigned n_chg = in.ctl->n_changed;
  const unsigned lane = lane_id();
  for (unsigned x = blockIdx.x; x < n_chg; x += gridDim.x) {
    const unsigned t0 = in.chg_tile[x], t1 = in.chg_tile[x + 1];
    if (t1 - t0 <= 1u) continue;  // finished by rebal_rs_local
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
__global__ void __launch_bounds__(RB_RS_TILE) rebal_rs_finish(const RebalIn* __restrict__ inp) {
  const RebalIn& in = *inp;
  const unsigned n_chg = in.ctl->n_changed, n_tiles = in.ctl->n_tiles;
  if (n_chg == 0u) return;
  const unsigned tid = threadIdx.x;
  for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const unsigned x = rebal_rs_user_of_tile(in, n_chg, tile), u = in.chg[x];
    if (in.chg_tile[x + 1] - in.chg_tile[x] <= 1u) continue;  // finished by rebal_rs_local
    const unsigned i = in.seg_start[u] + (tile - in.chg_tile[x]) * RB_RS_TILE + tid;
    if (i >= in.seg_end[u]) continue;
    const SumU4 t = combine(in.tile_carry[tile], in.pre_w[i]);
    if (t.bad) in.chg_bad[x] = 1u;
    in.pre_w[i] = SumU4{t.count, t.cpus, t.mem, t.gpus, 0u};
    const double d = rebal_dru_of(in, u, t);
    in.dru_w[i] = d;
    const unsigned hi = in.hidx[i];
    if (hi != 0xFFFFFFFFu) rebal_mirror_store(in, hi, d);
  }
}
// step 4: users where an addition rounded: left to right, exactly as the reference's reductions, then their DRUs again
__global__ void __launch_bounds__(256) rebal_rs_fix(const RebalIn* __restrict__ inp) {
  const RebalIn& in = *inp;
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
      if (hi != 0xFFFFFFFFu) rebal_mirror_store(in, hi, d);
    }
    __syncthreads();
  }
}

// Re-scoring when EVERY user is safe (rebal_user_safe: all sums exact in any association — integer-valued resources, the usual
// case): the masked prefix sums after a decision are the ones before it plus the usage of the slots that joined, minus the usage of
// the slots that left, at or before each position — exact, so bit-identical to scanning again.  One launch over the tiles of the
// changed users instead of four (local scans, carries, finish, fix), no dependency between tiles, and tiles before a user's first
// flipped slot are skipped.  The flipped slots (one per preempted task + the placed job: a handful) are listed by rebal_apply.
constexpr int RB_DL_LDS = 64;  // flips staged in LDS; a longer list is read from memory
static __device__ __forceinline__ SumU4 rebal_delta_upto(const RebalIn& in, const uint32_t* pos, const int32_t* sign, unsigned n_dl, unsigned s0,
                                                         unsigned upto) {
  // flips of the user whose segment starts at s0, at positions s0..upto (a segment is contiguous in B: position alone tells the user)
  SumU4 add = SumU4::zero();
  for (unsigned k = 0; k < n_dl; ++k) {
    const unsigned p = pos[k];
    if (p < s0 || p > upto) continue;
    const SumU4 v = in.s_use[p];
    if (sign[k] < 0) add.count -= v.count, add.cpus -= v.cpus, add.mem -= v.mem, add.gpus -= v.gpus;
    else add.count += v.count, add.cpus += v.cpus, add.mem += v.mem, add.gpus += v.gpus;
  }
  return add;
}
__global__ void __launch_bounds__(RB_RS_TILE) rebal_rs_delta(const RebalIn* __restrict__ inp) {
  const RebalIn& in = *inp;
  __shared__ uint32_t s_pos[RB_DL_LDS];
  __shared__ int32_t s_sign[RB_DL_LDS];
  __shared__ SumU4 s_val[RB_DL_LDS];
  const unsigned n_chg = in.ctl->n_changed, n_tiles = in.ctl->n_tiles, n_dl = in.ctl->n_delta;
  if (n_chg == 0u || n_dl == 0u) return;
  const unsigned tid = threadIdx.x;
  const bool staged = n_dl <= (unsigned)RB_DL_LDS;
  if (staged && tid < n_dl) {
    const unsigned p = in.dl_pos[tid];
    s_pos[tid] = p;
    s_sign[tid] = in.dl_sign[tid];
    s_val[tid] = in.s_use[p];
  }
  __syncthreads();
  for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const unsigned x = rebal_rs_user_of_tile(in, n_chg, tile), u = in.chg[x];
    const unsigned s0 = in.seg_start[u], s1 = in.seg_end[u];
    const unsigned i0 = s0 + (tile - in.chg_tile[x]) * RB_RS_TILE, i = i0 + tid;
    SumU4 add = SumU4::zero();
    bool any = false;
    if (staged) {
      for (unsigned k = 0; k < n_dl; ++k) {
        const unsigned p = s_pos[k];
        if (p < s0 || p >= s1 || p > i) continue;  // another user's, or behind this slot
        any = true;
        const SumU4 v = s_val[k];
        if (s_sign[k] < 0) add.count -= v.count, add.cpus -= v.cpus, add.mem -= v.mem, add.gpus -= v.gpus;
        else add.count += v.count, add.cpus += v.cpus, add.mem += v.mem, add.gpus += v.gpus;
      }
    } else {
      for (unsigned k = 0; k < n_dl; ++k) {
        const unsigned p = in.dl_pos[k];
        if (p < s0 || p >= s1 || p > i) continue;
        any = true;
        const SumU4 v = in.s_use[p];
        if (in.dl_sign[k] < 0) add.count -= v.count, add.cpus -= v.cpus, add.mem -= v.mem, add.gpus -= v.gpus;
        else add.count += v.count, add.cpus += v.cpus, add.mem += v.mem, add.gpus += v.gpus;
      }
    }
    if (i < s1 && any) {
      const SumU4 o = in.pre[i];
      const SumU4 t{o.count + add.count, o.cpus + add.cpus, o.mem + add.mem, o.gpus + add.gpus, 0u};
      in.pre_w[i] = t;
      const double d = rebal_dru_of(in, u, t);
      in.dru_w[i] = d;
      const unsigned hi = in.hidx[i];
      if (hi != 0xFFFFFFFFu) rebal_mirror_store(in, hi, d);
    }
  }
}

// ---- per pending job: quota test, pending DRU, group cohosts -------------------------------------------------------------
// n_dl > 0 (only from rebal_apply, every user safe): the prefix sums and DRUs in memory are the ones BEFORE the decision just taken
// — rebal_rs_delta runs after this — so the two values read from them are corrected by the flips at or before their positions.
static __device__ __forceinline__ void rebal_job_prep_dev(const RebalIn& in, unsigned pj, unsigned n_dl) {
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
      SumU4 t = in.pre[s1 - 1];
      if (n_dl) {
        const SumU4 a = rebal_delta_upto(in, in.dl_pos, in.dl_sign, n_dl, s0, s1 - 1);
        t = SumU4{t.count + a.count, t.cpus + a.cpus, t.mem + a.mem, t.gpus + a.gpus, 0u};
      }
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
  double near = last >= 0 ? in.dru[last] : 0.0;
  if (n_dl && last >= 0) {
    const SumU4 a = rebal_delta_upto(in, in.dl_pos, in.dl_sign, n_dl, s0, (unsigned)last);
    const SumU4 o = in.pre[last];
    near = rebal_dru_of(in, us, SumU4{o.count + a.count, o.cpus + a.cpus, o.mem + a.mem, o.gpus + a.gpus, 0u});
  }
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
__global__ void __launch_bounds__(COOK_WAVE) rebal_job_prep(const RebalIn* __restrict__ inp, unsigned pj) {
  const RebalIn& in = *inp;
  if (lane_id() == 0) in.ctl->n_changed = 0u, in.ctl->n_delta = 0u;  // nothing to re-score unless rebal_apply takes a decision
  rebal_job_prep_dev(in, pj, 0u);
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

// ---- per host: candidates, priority order, prefix aggregates, best feasible prefix ------------------------------------------
struct HostBest {  // best feasible prefix of a host for the current job
  unsigned long long key;  // f64_key(dru), 0 = none
  unsigned len;
  double dru, c, m, g;
};
// Jobs placed earlier in this cycle form one chain per host (x_head[h] -> x_next[pj] -> ...), newest first; x_cnt[h] = its length.
// The order inside a host does not matter: the host's items are ordered by (dru desc, position in B asc) anyway.
static __device__ __forceinline__ unsigned rebal_chain_at(const RebalIn& in, unsigned h, unsigned q) {
  unsigned cur = in.x_head[h];
  for (unsigned s = 0; s < q; ++s) cur = in.x_next[cur];
  return cur;
}
// placed jobs on hosts before h (the host's offset into the global scratch, slow path only): a count over the <= P placed jobs
static __device__ __forceinline__ unsigned rebal_x_before(const RebalIn& in, unsigned h) {
  const unsigned n_x = in.ctl->n_x;
  unsigned c = 0;
  for (unsigned i = lane_id(); i < n_x; i += COOK_WAVE) c += in.x_host[in.x_pj[i]] < h ? 1u : 0u;
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, COOK_WAVE);
  return c;
}
// the host's last scored task in priority-map order (rebalancer.clj:369-375) decides which attribute map the host resolves to
static __device__ __forceinline__ int rebal_host_row(const RebalIn& in, unsigned h, double last_d, unsigned last_pb, unsigned last_slot) {
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
  return known ? in.row_of_host[h] : -1;
}

// A host with at most 64 items (running tasks + jobs placed this cycle), one item per lane, everything in registers: filter
// (rebalancer.clj:339-349), rank by (dru desc, position in B asc) with wave broadcasts, prefix aggregates seeded with the spare
// resources (:384-403), best feasible prefix (:404).  l_rank = 64 words of LDS owned by this wave.  -> out.key != 0 when the host can
// take the job; sorted_slot = lane k's slot of the k-th candidate in priority order (the preempted tasks are a prefix of it).
template <bool SAFE = false>
static __device__ __forceinline__ void rebal_host_small(const RebalIn& in, const RebalJob& jb, unsigned h, unsigned hs, unsigned n_seg, unsigned n_here,
                                                       bool sp, uint32_t* l_rank, HostBest& out, unsigned& sorted_slot) {
  const unsigned lane = lane_id(), t = lane, n = n_seg + n_here;
  out.key = 0ull;
  out.len = 0;
  out.dru = out.c = out.m = out.g = 0.0;
  sorted_slot = 0xFFFFFFFFu;
  const bool valid = t < n, mirrored = t < n_seg;
  unsigned my_pj = 0;
  if (n_here) {  // wave-uniform
    unsigned cur = in.x_head[h];
    for (unsigned q = 0; q < n_here; ++q) {
      if (t == n_seg + q) my_pj = cur;
      cur = in.x_next[cur];
    }
  }
  unsigned slot = 0, pb = 0, usr = 0;
  bool a = false;
  double d = 0.0, xc = 0.0, xm = 0.0, xg = 0.0;
  if (mirrored) {  // a running slot: its columns lie at hs + t of the host-ordered mirrors
    slot = in.hperm[hs + t];
    pb = in.h_pb[hs + t];
    a = in.h_act[hs + t] != 0;
    usr = in.h_user[hs + t];
    d = in.h_dru[hs + t];
    xc = in.h_cpus[hs + t], xm = in.h_mem[hs + t], xg = in.h_gpus[hs + t];
  } else if (valid) {  // a job placed earlier in this cycle
    slot = in.R + my_pj;
    pb = in.posB[slot];
    a = in.act[pb] != 0;
    usr = in.slot_user[slot];
    d = in.dru[pb];
    xc = in.slot_cpus[slot], xm = in.slot_mem[slot], xg = in.slot_gpus[slot];
  }
  if (!a) d = 0.0;
#if defined(RB_CUT) && RB_CUT == 1
  if (d == 123.456) out.key = 1ull;
  return;
#endif
  const bool cand = a && (jb.below || usr == jb.us) && !(d < in.safe_dru) && (d - jb.pdru > in.min_diff);
  const unsigned long long mask = __ballot(cand);
  const unsigned n_c = (unsigned)__popcll(mask);
  if (n_c == 0 && !sp) return;  // nothing to preempt and nothing spare: no prefix exists (wave-uniform)
  // which attribute map the host resolves to: its last scored task's (rebalancer.clj:369-375).  When every running task's slave id is
  // cached and no placed job sits on the host, any active item gives the same answer
  int row;
  if (!in.attrs_cached && n_here == 0)
    row = __ballot(a) != 0ull ? in.row_of_host[h] : -1;
  else
    row = rebal_host_row(in, h, d, pb, a ? slot : 0xFFFFFFFFu);
  if (!rebal_job_constraints(in, jb.pj, row, jb.g)) return;
  if (jb.gtype && !rebal_group_constraint(in, jb, row)) return;
#if defined(RB_CUT) && RB_CUT == 2
  if (d == 123.456) out.key = 1ull;
  return;
#endif
  // priority-map order inside the host = (dru desc, position in B asc): rank by counting, the other items broadcast one by one
  unsigned r = 0;
  for (unsigned long long m = mask; m != 0ull; m &= m - 1ull) {
    const int j = __ffsll((unsigned long long)m) - 1;
    const double dj = wave_read_lane_f64(d, j);
    const unsigned pjx = (unsigned)wave_read_lane((int)pb, j);
    r += (dj > d || (dj == d && pjx < pb)) ? 1u : 0u;
  }
  if (cand) l_rank[r] = lane;
  wave_sync();
  const unsigned src = lane < n_c ? l_rank[lane] : 0u;
  wave_sync();  // (the next call of this wave reuses l_rank)
  const bool vk = lane < n_c;
  const double sd = __shfl(d, (int)src, COOK_WAVE);
  SumU4 x = SumU4{0.0, __shfl(xc, (int)src, COOK_WAVE), __shfl(xm, (int)src, COOK_WAVE), __shfl(xg, (int)src, COOK_WAVE), 0u};
  sorted_slot = (unsigned)__shfl((int)slot, (int)src, COOK_WAVE);
  if (!vk) {
    x = SumU4::zero();
    sorted_slot = 0xFFFFFFFFu;
  }
#if defined(RB_CUT) && RB_CUT == 3
  if (sd == 123.456) out.key = 1ull;
  return;
#endif
  // prefix aggregates seeded with the spare resources, best feasible prefix: key (f64_key(dru), len) lexicographic max; len 0 = the
  // spare pseudo-entry alone
  const double jc = jb.c, jm = jb.m, jg = jb.g;
  const bool need_g = jb.has_gpus != 0;
  SumU4 seed = SumU4::zero();
  if (sp) seed = SumU4{0.0, 0.0 + in.spare_c[h], 0.0 + in.spare_m[h], 0.0 + in.spare_g[h], 0u};
  const double DMAXV = 1.7976931348623157e308;
  unsigned long long bk = 0ull;
  unsigned bl = 0;
  double bd = 0.0, bc = 0.0, bm = 0.0, bg = 0.0;
  const bool spare_alone = sp && seed.mem >= jm && seed.cpus >= jc && (need_g ? seed.gpus >= jg : true);
  const SumU4 tt = combine_t<SAFE>(seed, wave_incl_scan_u4_rows<SAFE>(x, n_c));
  if (!SAFE && __any(vk && tt.bad != 0u)) {  // a partial sum rounded: left to right, exactly as the reference's reductions (all lanes alike)
    double ac = seed.cpus, am = seed.mem, ag = seed.gpus;
    if (spare_alone) bk = f64_key(DMAXV), bd = DMAXV, bc = ac, bm = am, bg = ag;
    for (unsigned k = 0; k < n_c; ++k) {
      ac += __shfl(x.cpus, (int)k, COOK_WAVE);
      am += __shfl(x.mem, (int)k, COOK_WAVE);
      ag += __shfl(x.gpus, (int)k, COOK_WAVE);
      const double dk = __shfl(sd, (int)k, COOK_WAVE);
      if (am >= jm && ac >= jc && (need_g ? ag >= jg : true) && dk >= 0.0 && f64_key(dk) >= bk) bk = f64_key(dk), bl = k + 1, bd = dk, bc = ac, bm = am, bg = ag;
    }
  } else {
    if (lane == 0 && spare_alone) bk = f64_key(DMAXV), bd = DMAXV, bc = seed.cpus, bm = seed.mem, bg = seed.gpus;
    const bool enough = vk && tt.mem >= jm && tt.cpus >= jc && (need_g ? tt.gpus >= jg : true) && sd >= 0.0;
    if (enough && f64_key(sd) >= bk) bk = f64_key(sd), bl = lane + 1, bd = sd, bc = tt.cpus, bm = tt.mem, bg = tt.gpus;  // later prefix wins ties (max-key, :404)
    // wave arg-max of (bk, bl): the greatest key, and among equal keys the longest prefix = the highest lane (lane k holds prefix
    // k + 1; the spare pseudo-entry, length 0, sits in lane 0 under the greatest key there is)
    const unsigned long long mk = wave_max_u64(bk);
    if (mk == 0ull) return;
    const unsigned long long own = __ballot(bk == mk);
    const int wl = 63 - __clzll((unsigned long long)own);
    bk = mk;
    bl = (unsigned)wave_read_lane((int)bl, wl);
    bd = wave_read_lane_f64(bd, wl), bc = wave_read_lane_f64(bc, wl), bm = wave_read_lane_f64(bm, wl), bg = wave_read_lane_f64(bg, wl);
  }
  out.key = bk;
  out.len = bl;
  out.dru = bd, out.c = bc, out.m = bm, out.g = bg;
}

// Two hosts with at most 32 items each in ONE wave, a half per host (lanes 0..31: host h0, lanes 32..63: host h0 + 1): the launch is
// bound by the instructions its 50k waves issue, and a typical host fills a third of a wave.  Same steps as rebal_host_small with every
// cross-lane operation confined to the half.  The job must not belong to a constrained group (that check is wave-cooperative).
// -> lanes 0 and 32 hold the result of their host.
template <bool SAFE = false>
static __device__ __forceinline__ void rebal_host_pair(const RebalIn& in, const RebalJob& jb, unsigned h0, uint32_t* l_rank, HostBest& out) {
  const unsigned lane = lane_id(), sub = lane >> 5, t = lane & 31u, hb = sub << 5;
  const unsigned h = h0 + sub;
  const bool hv = h < in.H;
  const unsigned hs = hv ? in.hstart[h] : 0u, n_seg = hv ? in.hend[h] - hs : 0u, n_here = hv ? in.x_cnt[h] : 0u;
  const unsigned n = n_seg + n_here;
  const bool sp = hv && in.has_spare[h] != 0;
  out.key = 0ull;
  out.len = 0;
  out.dru = out.c = out.m = out.g = 0.0;
  const bool valid = t < n, mirrored = t < n_seg;
  unsigned slot = 0, pb = 0, usr = 0;
  bool a = false;
  double d = 0.0, xc = 0.0, xm = 0.0, xg = 0.0;
  if (mirrored) {
    slot = in.hperm[hs + t];
    pb = in.h_pb[hs + t];
    a = in.h_act[hs + t] != 0;
    usr = in.h_user[hs + t];
    d = in.h_dru[hs + t];
    xc = in.h_cpus[hs + t], xm = in.h_mem[hs + t], xg = in.h_gpus[hs + t];
  } else if (valid) {  // a job placed earlier in this cycle: the (t - n_seg)-th of the host's chain
    slot = in.R + rebal_chain_at(in, h, t - n_seg);
    pb = in.posB[slot];
    a = in.act[pb] != 0;
    usr = in.slot_user[slot];
    d = in.dru[pb];
    xc = in.slot_cpus[slot], xm = in.slot_mem[slot], xg = in.slot_gpus[slot];
  }
  if (!a) d = 0.0;
  const bool cand = a && (jb.below || usr == jb.us) && !(d < in.safe_dru) && (d - jb.pdru > in.min_diff);
  const unsigned long long mask = __ballot(cand);
  const unsigned hmask = (unsigned)(mask >> hb);
  const unsigned n_c = (unsigned)__popc(hmask);
  bool alive = hv && !(n_c == 0 && !sp);
  // which attribute map the host resolves to (see rebal_host_small)
  int row = -1;
  if (__any(in.attrs_cached != nullptr || n_here != 0u)) {  // wave-uniform: the general rule, reduced over the half
    double last_d = d;
    unsigned last_pb = pb, last_slot = a ? slot : 0xFFFFFFFFu;
    for (int dd = 16; dd >= 1; dd >>= 1) {
      const double od = __shfl_xor(last_d, dd, COOK_WAVE);
      const unsigned opb = __shfl_xor(last_pb, dd, COOK_WAVE), osl = __shfl_xor(last_slot, dd, COOK_WAVE);
      if (osl != 0xFFFFFFFFu && (last_slot == 0xFFFFFFFFu || od < last_d || (od == last_d && opb > last_pb))) last_d = od, last_pb = opb, last_slot = osl;
    }
    bool known = false;
    if (last_slot != 0xFFFFFFFFu)
      known = last_slot < in.R ? (in.attrs_cached ? in.attrs_cached[last_slot] != 0 : true) : in.x_known[last_slot - in.R] != 0;
    row = (known && hv) ? in.row_of_host[h] : -1;
  } else {
    const unsigned amask = (unsigned)(__ballot(a) >> hb);
    row = (amask != 0u && hv) ? in.row_of_host[h] : -1;
  }
  if (alive && !rebal_job_constraints(in, jb.pj, row, jb.g)) alive = false;
  if (!__any(alive)) return;
  // rank inside the half by counting; item jj of both halves is broadcast in one step
  unsigned r = 0;
  const unsigned either = (unsigned)mask | (unsigned)(mask >> 32);
  for (unsigned m = either; m != 0u; m &= m - 1u) {
    const int jj = __ffs((int)m) - 1;
    const int src = (int)hb + jj;
    const double dj = __shfl(d, src, COOK_WAVE);
    const unsigned pjx = (unsigned)__shfl((int)pb, src, COOK_WAVE);
    const bool cj = ((hmask >> jj) & 1u) != 0u;
    r += (cj && (dj > d || (dj == d && pjx < pb))) ? 1u : 0u;
  }
  if (cand) l_rank[hb + r] = lane;
  wave_sync();
  const bool vk = t < n_c;
  const unsigned src = vk ? l_rank[hb + t] : lane;
  wave_sync();
  const double sd = __shfl(d, (int)src, COOK_WAVE);
  SumU4 x = SumU4{0.0, __shfl(xc, (int)src, COOK_WAVE), __shfl(xm, (int)src, COOK_WAVE), __shfl(xg, (int)src, COOK_WAVE), 0u};
  if (!vk) x = SumU4::zero();
  const double jc = jb.c, jm = jb.m, jg = jb.g;
  const bool need_g = jb.has_gpus != 0;
  SumU4 seed = SumU4::zero();
  if (sp) seed = SumU4{0.0, 0.0 + in.spare_c[h], 0.0 + in.spare_m[h], 0.0 + in.spare_g[h], 0u};
  const double DMAXV = 1.7976931348623157e308;
  unsigned long long bk = 0ull;
  unsigned bl = 0;
  double bd = 0.0, bc = 0.0, bm = 0.0, bg = 0.0;
  const bool spare_alone = sp && seed.mem >= jm && seed.cpus >= jc && (need_g ? seed.gpus >= jg : true);
  // inclusive scan inside the half: four row steps + the row-to-row step (rows 0 -> 1 and 2 -> 3)
  SumU4 sc = scan_step_u4<0, SAFE>(x);
  sc = scan_step_u4<1, SAFE>(sc);
  sc = scan_step_u4<2, SAFE>(sc);
  sc = scan_step_u4<3, SAFE>(sc);
  sc = scan_step_u4<4, SAFE>(sc);
  const SumU4 tt = combine_t<SAFE>(seed, sc);
  const unsigned hbad = SAFE ? 0u : (unsigned)(__ballot(vk && tt.bad != 0u) >> hb);
  if (!SAFE && __any(hbad != 0u)) {  // a partial sum rounded in some half: that half redoes it left to right like the reference (every lane of it alike)
    double ac = seed.cpus, am = seed.mem, ag = seed.gpus;
    unsigned long long k2 = 0ull;
    unsigned l2 = 0;
    double d2 = 0.0, c2 = 0.0, m2 = 0.0, g2 = 0.0;
    if (spare_alone) k2 = f64_key(DMAXV), d2 = DMAXV, c2 = ac, m2 = am, g2 = ag;
    for (unsigned k = 0; k < 32u; ++k) {
      if (!__any(k < n_c)) break;
      const double ck = __shfl(x.cpus, (int)(hb + k), COOK_WAVE), mk2 = __shfl(x.mem, (int)(hb + k), COOK_WAVE), gk = __shfl(x.gpus, (int)(hb + k), COOK_WAVE);
      const double dk = __shfl(sd, (int)(hb + k), COOK_WAVE);
      if (k < n_c) {
        ac += ck, am += mk2, ag += gk;
        if (am >= jm && ac >= jc && (need_g ? ag >= jg : true) && dk >= 0.0 && f64_key(dk) >= k2) k2 = f64_key(dk), l2 = k + 1, d2 = dk, c2 = ac, m2 = am, g2 = ag;
      }
    }
    if (hbad != 0u) {
      if (alive && t == 0) out.key = k2, out.len = l2, out.dru = d2, out.c = c2, out.m = m2, out.g = g2;
      alive = false;  // this half is done
    }
  }
  if (t == 0 && spare_alone) bk = f64_key(DMAXV), bd = DMAXV, bc = seed.cpus, bm = seed.mem, bg = seed.gpus;
  const bool enough = vk && tt.mem >= jm && tt.cpus >= jc && (need_g ? tt.gpus >= jg : true) && sd >= 0.0;
  if (enough && f64_key(sd) >= bk) bk = f64_key(sd), bl = t + 1, bd = sd, bc = tt.cpus, bm = tt.mem, bg = tt.gpus;  // later prefix wins ties (:404)
  // arg-max of (bk, bl) over the half: the greatest key, among equal keys the highest lane (= the longest prefix)
  const unsigned long long mk = half_max_u64(bk);
  const unsigned own = (unsigned)(__ballot(bk == mk) >> hb);
  const int wl = (int)hb + 31 - __clz((int)own);  // own != 0: the lane holding the maximum is among them
  const unsigned wlen = (unsigned)__shfl((int)bl, wl, COOK_WAVE);
  const double wd = __shfl(bd, wl, COOK_WAVE), wc = __shfl(bc, wl, COOK_WAVE), wm = __shfl(bm, wl, COOK_WAVE), wg = __shfl(bg, wl, COOK_WAVE);
  if (alive && t == 0 && mk != 0ull) out.key = mk, out.len = wlen, out.dru = wd, out.c = wc, out.m = wm, out.g = wg;
}

// hosts with at most 64 items (running tasks + jobs placed this cycle): a wave takes two neighbouring hosts, half a wave each when
// both hold at most 32 items (rebal_host_pair), else one after the other.  No LDS beyond 64 words per wave, few registers.  The launch
// is bound by the instructions its waves issue.  Larger hosts are left to rebal_decide_big.
// Two launches per job when in.thr_key != 0 (the decision is the arg-max of the hosts' keys, and a host's key is the DRU of one of its active items, or
// the greatest key there is when its spare resources alone hold the job): phase 0 evaluates the hosts whose BOUND (in.hmax_key; all ones for a host
// whose spare resources hold the job) reaches in.thr_key and raises in.best_key to the best key found; phase 1 evaluates the other hosts whose bound
// reaches that key (an equal key still matters: the later host wins ties) — usually a few per cent of them.  A host left out cannot be the arg-max.
template <bool SAFE>
__global__ void __launch_bounds__(COOK_WAVE* RB_WAVES) rebal_decide(const RebalIn* __restrict__ inp, unsigned phase) {
  const RebalIn& in = *inp;
  __shared__ uint32_t l_rank[RB_WAVES][COOK_WAVE];
  __shared__ unsigned long long s_bk[RB_WAVES];
  __shared__ unsigned s_bh[RB_WAVES];
  __shared__ unsigned s_q[RB_PAIRS], s_nq;
  // the second phase has nothing to do when the first one's best key reaches the threshold (every host left has a bound below it): the usual case, and
  // rebal_apply then does not read this phase's entries
  if (phase != 0u && *in.best_key >= in.thr_key) return;
  const RebalJob jb = *in.job;
  if (!jb.active) return;
  const unsigned lane = lane_id(), w = wave_id();
  // (1) which of the workgroup's RB_PAIRS pairs of hosts are this phase's: a thread per pair, the pairs to evaluate into a list.  (Evaluating a host that
  // is not this phase's does no harm — its result is the same — so a pair goes by its better host.)
  if (threadIdx.x == 0) s_nq = 0u;
  __syncthreads();
  if (threadIdx.x < (unsigned)RB_PAIRS) {
    const unsigned pr = blockIdx.x * (unsigned)RB_PAIRS + threadIdx.x;
    bool mine = false;
    if (2u * pr < in.H) {
      if (in.thr_key == 0ull) {
        mine = true;
      } else {
        const unsigned long long best = phase ? *in.best_key : 0ull;
        for (unsigned q = 0; q < 2u && 2u * pr + q < in.H; ++q) {
          const unsigned h = 2u * pr + q;
          unsigned long long bd = in.hmax_key[h];
          if (in.has_spare[h] != 0) {
            const double sc = 0.0 + in.spare_c[h], sm = 0.0 + in.spare_m[h], sg = 0.0 + in.spare_g[h];
            if (sm >= jb.m && sc >= jb.c && (jb.has_gpus != 0 ? sg >= jb.g : true)) bd = ~0ull;
          }
          mine = mine || (phase == 0u ? bd >= in.thr_key : (bd < in.thr_key && bd >= best && bd != 0ull));
        }
      }
    }
    if (mine) s_q[atomicAdd(&s_nq, 1u)] = pr;
  }
  __syncthreads();
  const unsigned nq = s_nq;
  // (2) the listed pairs, a wave each in turn.  The wave's better host (the later one on ties), then the workgroup's: rebal_apply's arg-max reads one
  // entry per workgroup and phase
  unsigned long long wk = 0ull;
  unsigned wh = 0u;
  auto take = [&](unsigned long long k, unsigned h) {
    if (k != 0ull && (k > wk || (k == wk && h > wh))) wk = k, wh = h;  // (wave-uniform)
  };
  auto eval_pair = [&](unsigned h0) {
    const bool two = h0 + 1u < in.H;
#ifdef RB_COUNT  // study build: waves that evaluate, per phase (rebalance_run prints them)
    if (lane == 0) atomicAdd(&in.best_key[1u + phase], 1ull);
    if (lane == 0 && jb.below) atomicAdd(&in.best_key[3u], 1ull);
#endif
    const unsigned n0 = in.hend[h0] - in.hstart[h0] + in.x_cnt[h0];
    const unsigned n1 = two ? in.hend[h0 + 1u] - in.hstart[h0 + 1u] + in.x_cnt[h0 + 1u] : 0u;
    if (n0 <= 32u && n1 <= 32u && jb.gtype == 0u) {
      HostBest hb;
      rebal_host_pair<SAFE>(in, jb, h0, l_rank[w], hb);
      if ((lane & 31u) == 0u && h0 + (lane >> 5) < in.H) {
        const unsigned h = h0 + (lane >> 5);
        in.hres_key[h] = hb.key;
        if (hb.key != 0ull) {
          in.hres_len[h] = hb.len;
          in.hres_base[h] = 0xFFFFFFFFu;  // rebal_apply re-derives the preempted prefix of a small host itself
          in.hres_dru[h] = hb.dru;
          in.hres_c[h] = hb.c;
          in.hres_m[h] = hb.m;
          in.hres_g[h] = hb.g;
        }
      }
      const unsigned long long k0 = wave_read_lane_u64(hb.key, 0), k1 = two ? wave_read_lane_u64(hb.key, 32) : 0ull;
      take(k0, h0);
      take(k1, h0 + 1u);
      return;
    }
    for (unsigned q = 0; q < (two ? 2u : 1u); ++q) {
      const unsigned h = h0 + q;
      const unsigned hs = in.hstart[h], n_seg = in.hend[h] - hs;
      const unsigned n_here = in.x_cnt[h];  // jobs placed on this host earlier in the cycle
      const unsigned n = n_seg + n_here;
      const bool sp = in.has_spare[h] != 0;
      if (n > (unsigned)COOK_WAVE) continue;  // rebal_decide_big's
      HostBest hb;
      hb.key = 0ull;
      if (n != 0 || sp) {
        unsigned ss;
        rebal_host_small<SAFE>(in, jb, h, hs, n_seg, n_here, sp, l_rank[w], hb, ss);
      }
      if (lane == 0) {
        in.hres_key[h] = hb.key;
        if (hb.key != 0ull) {
          in.hres_len[h] = hb.len;
          in.hres_base[h] = 0xFFFFFFFFu;
          in.hres_dru[h] = hb.dru;
          in.hres_c[h] = hb.c;
          in.hres_m[h] = hb.m;
          in.hres_g[h] = hb.g;
        }
      }
      take(hb.key, h);
    }
  };
  for (unsigned i = w; i < nq; i += (unsigned)RB_WAVES) eval_pair(2u * s_q[i]);
  if (lane == 0) s_bk[w] = wk, s_bh[w] = wh;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long bk = 0ull;
    unsigned bh = 0u;
    for (int k = 0; k < RB_WAVES; ++k)
      if (s_bk[k] != 0ull && (s_bk[k] > bk || (s_bk[k] == bk && s_bh[k] > bh))) bk = s_bk[k], bh = s_bh[k];
    in.blk_key[phase * in.n_blk + blockIdx.x] = bk;
    in.blk_host[phase * in.n_blk + blockIdx.x] = bh;
    if (in.thr_key != 0ull && phase == 0u && bk != 0ull) atomicMax(in.best_key, bk);
  }
}

// hosts with more than 64 items: lists in LDS (<= RB_CAP) or in the host's region of the global scratch.  Such hosts are few or none
// (a million tasks on 50k hosts: none): they are kept in a list (big_list: the hosts whose running tasks alone exceed 64, plus the ones
// rebal_apply pushes over that mark) that a small fixed grid walks, a wave per host.
struct BigLds {
  double dru[RB_CAP], cpus[RB_CAP], mem[RB_CAP], gpus[RB_CAP];
  uint32_t posB[RB_CAP], slot[RB_CAP], ord[RB_CAP];
};
__global__ void __launch_bounds__(256) rebal_big_init(const uint32_t* __restrict__ hstart, const uint32_t* __restrict__ hend, unsigned H,
                                                      uint32_t* __restrict__ big_list, RebalCtl* __restrict__ ctl) {
  const unsigned h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h < H && hend[h] - hstart[h] > (unsigned)COOK_WAVE) big_list[atomicAdd(&ctl->n_big, 1u)] = h;
}
static __device__ void rebal_host_big(const RebalIn& in, const RebalJob& jb, unsigned h, BigLds& L) {
  const unsigned lane = lane_id();
  const unsigned hs = in.hstart[h], n_seg = in.hend[h] - hs;
  const unsigned n_here = in.x_cnt[h];
  const unsigned n = n_seg + n_here;
  const bool sp = in.has_spare[h] != 0;
  if (lane == 0) in.hres_key[h] = 0ull;
  // the region starts after the running tasks of the hosts before it and the jobs placed on them.
  // (hstart is only meaningful for hosts that HAVE running tasks: an empty host's 0 made its region collide with another host's.)
  const bool big = n > (unsigned)RB_CAP;
  const unsigned base = in.hbase[h] + rebal_x_before(in, h);
  double *c_dru = big ? in.gs_dru + base : L.dru, *c_cpus = big ? in.gs_cpus + base : L.cpus;
  double *c_mem = big ? in.gs_mem + base : L.mem, *c_gpus = big ? in.gs_gpus + base : L.gpus;
  uint32_t *c_posB = big ? in.gs_posB + base : L.posB, *c_slot = big ? in.gs_slot + base : L.slot;
  uint32_t* c_ord = big ? in.gs_ord + base : L.ord;
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
      slot = in.R + rebal_chain_at(in, h, t - n_seg);
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
  const int row = rebal_host_row(in, h, last_d, last_pb, last_slot);
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

__global__ void __launch_bounds__(COOK_WAVE* RB_WAVES) rebal_decide_big(const RebalIn* __restrict__ inp) {
  const RebalIn& in = *inp;
  __shared__ BigLds s_l[RB_WAVES];
  const RebalJob jb = *in.job;
  if (!jb.active) return;
  const unsigned n_big = in.ctl->n_big;
  for (unsigned x = blockIdx.x * RB_WAVES + wave_id(); x < n_big; x += gridDim.x * RB_WAVES) {
    const unsigned h = in.big_list[x];
    if (in.hend[h] - in.hstart[h] + in.x_cnt[h] > (unsigned)COOK_WAVE) rebal_host_big(in, jb, h, s_l[wave_id()]);
    wave_sync();  // the wave's lists are free again
  }
}

// ---- arg-max over hosts + next-state (rebalancer.clj:270-309, 404) -----------------------------------------------------------
constexpr int RB_APPLY_THREADS = 1024;
// pj_next != COOK_NONE (every user safe, rebalance_run): the workgroup goes on to prepare the NEXT pending job (rebal_job_prep_dev
// with this decision's flips) — one launch less per pending job; the re-scoring of the changed users follows as rebal_rs_delta.
__global__ void __launch_bounds__(RB_APPLY_THREADS) rebal_apply(const RebalIn* __restrict__ inp, unsigned pj_next) {
  const RebalIn& in = *inp;
  __shared__ unsigned long long s_key[RB_APPLY_THREADS / COOK_WAVE];
  __shared__ unsigned s_host[RB_APPLY_THREADS / COOK_WAVE];
  __shared__ uint32_t s_rank[COOK_WAVE], s_pre[COOK_WAVE];
  const RebalJob jb = *in.job;
  if (!jb.active) return;  // the budget is spent: the next job stays inactive too
  const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
  // (both phases of this job's rebal_decide are behind this launch: the next job's start from no best key)
  const unsigned long long best0 = in.thr_key != 0ull ? *in.best_key : 0ull;
  __syncthreads();
  if (tid == 0 && in.thr_key != 0ull) *in.best_key = 0ull;
  // arg-max of the hosts' keys, the later host winning ties (max-key, rebalancer.clj:404).  Eight independent loads in flight per
  // thread: a dependent one-load-per-iteration loop over 50k hosts was the larger half of this kernel.
  unsigned long long bk = 0ull;
  unsigned bh = 0;
  const unsigned n_ent = (in.thr_key != 0ull && best0 < in.thr_key) ? 2u * in.n_blk : in.n_blk;  // one entry per rebal_decide workgroup and phase that ran
  for (unsigned b0 = tid; b0 < n_ent; b0 += 8u * RB_APPLY_THREADS) {
    unsigned long long k[8];
    unsigned hh[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned b = b0 + (unsigned)q * RB_APPLY_THREADS;
      k[q] = b < n_ent ? in.blk_key[b] : 0ull;
      hh[q] = b < n_ent ? in.blk_host[b] : 0u;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (k[q] != 0ull && (k[q] > bk || (k[q] == bk && hh[q] >= bh))) bk = k[q], bh = hh[q];  // the later host wins ties
  }
  // hosts of more than 64 items are rebal_decide_big's: their keys are not in any workgroup's entry
  for (unsigned x = tid; x < in.ctl->n_big; x += RB_APPLY_THREADS) {
    const unsigned h = in.big_list[x];
    const unsigned long long k = in.hres_key[h];
    if (k != 0ull && in.hend[h] - in.hstart[h] + in.x_cnt[h] > (unsigned)COOK_WAVE && (k > bk || (k == bk && h > bh))) bk = k, bh = h;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long ok = __shfl_xor(bk, d, COOK_WAVE);
    const unsigned oh = __shfl_xor(bh, d, COOK_WAVE);
    if (ok > bk || (ok == bk && ok != 0ull && oh > bh)) bk = ok, bh = oh;
  }
  if (lane == 0) s_key[w] = bk, s_host[w] = bh;
  __syncthreads();
  if (w != 0) return;
  bk = lane < RB_APPLY_THREADS / COOK_WAVE ? s_key[lane] : 0ull;
  bh = lane < RB_APPLY_THREADS / COOK_WAVE ? s_host[lane] : 0u;
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long ok = __shfl_xor(bk, d, COOK_WAVE);
    const unsigned oh = __shfl_xor(bh, d, COOK_WAVE);
    if (ok > bk || (ok == bk && ok != 0ull && oh > bh)) bk = ok, bh = oh;
  }
  unsigned n_dl = 0;
  if (bk != 0ull) {  // (wave-uniform)  0: no host can take the job — no decision, state unchanged (rebalancer.clj:455-458)
    const unsigned h = bh;
    const unsigned len = in.hres_len[h], base = in.hres_base[h];
    // the preempted tasks = the first `len` candidates of the host in priority order: a small host's are re-derived here (one wave,
    // registers), a large host's were listed by rebal_decide
    if (base == 0xFFFFFFFFu) {
      const unsigned hs = in.hstart[h];
      HostBest hb;
      unsigned ss;
      rebal_host_small(in, jb, h, hs, in.hend[h] - hs, in.x_cnt[h], in.has_spare[h] != 0, s_rank, hb, ss);
      s_pre[lane] = ss;
    }
    wave_sync();
    if (len <= (unsigned)COOK_WAVE) {
      // The usual case (a host of more than 64 items is rebal_decide_big's): FIRST every load the commit needs — lane k the facts of the k-th preempted task, lane 0
      // the control block, the host's and the job's words; every one of these loads is in flight before the first store (the arrays behind RebalIn's pointers may
      // alias for the compiler: a store between two loads made the old form a chain of ~25 dependent round trips by one lane) —, THEN lane 0 writes, in the old order.
      unsigned g_slot = 0, g_pbk = 0, g_user = 0xFFFFFFFFu, g_hx = 0, g_nt = 0;
      bool g_known = false;
      if (lane < len) {
        g_slot = base == 0xFFFFFFFFu ? s_pre[lane] : in.srt_slot[base + lane];
        g_pbk = in.posB[g_slot];
        g_user = in.slot_user[g_slot];
        g_known = g_slot < in.R ? (in.attrs_cached ? in.attrs_cached[g_slot] != 0 : true) : in.x_known[g_slot - in.R] != 0;
        g_hx = g_slot < in.R ? in.hidx[g_pbk] : 0u;
        g_nt = (in.seg_end[g_user] - in.seg_start[g_user] + RB_RS_TILE - 1) / RB_RS_TILE;
      }
      RebalCtl c;
      cook_preemption d;
      unsigned pbj = 0, xh = 0, xc = 0, items0 = 0, nt_us = 0;
      if (lane == 0) {
        c = *in.ctl;
        d.dru = in.hres_dru[h], d.cpus = in.hres_c[h], d.mem = in.hres_m[h], d.gpus = in.hres_g[h];
        pbj = in.posB[in.R + jb.pj];
        xh = in.x_head[h], xc = in.x_cnt[h];
        items0 = in.hend[h] - in.hstart[h];
        nt_us = (in.seg_end[jb.us] - in.seg_start[jb.us] + RB_RS_TILE - 1) / RB_RS_TILE;
      }
      // a user is listed once per decision: the job's own first, then the preempted tasks' in their order (the old form's stamps in chg_mark)
      s_rank[lane] = g_user;  // (rebal_host_small is done with s_rank)
      wave_sync();
      bool g_dup = g_user == jb.us;
      for (unsigned j = 0; j < lane && lane < len; ++j) g_dup = g_dup || s_rank[j] == g_user;
      const unsigned long long m_dup = cook_ballot(g_dup), m_known = cook_ballot(g_known);
      // lane 0 takes lane k's words from the registers they were loaded into (no staging: wave_read_lane of a uniform k)
      bool first_known = false;
      unsigned n_chg = 0, n_tiles = 0;
      if (lane == 0) {
        d.pending_index = jb.pj;
        d.host = h;
        d.task_off = c.np;
        d.task_n = len;
        in.decisions[c.nd++] = d;
        in.chg[0] = jb.us, in.chg_bad[0] = 0u, in.chg_tile[0] = 0u;
        n_chg = 1u, n_tiles = nt_us;
      }
      for (unsigned k = 0; k < len; ++k) {  // (wave-uniform trip count; the reads below are cross-lane, the writes lane 0's)
        const unsigned slot = (unsigned)wave_read_lane((int)g_slot, (int)k), pbk = (unsigned)wave_read_lane((int)g_pbk, (int)k), u = (unsigned)wave_read_lane((int)g_user, (int)k),
                       hx = (unsigned)wave_read_lane((int)g_hx, (int)k), nt = (unsigned)wave_read_lane((int)g_nt, (int)k);
        const bool dup = (m_dup >> k) & 1ull, known = (m_known >> k) & 1ull;
        if (lane == 0) {
          in.act[pbk] = 0;
          in.dl_pos[n_dl] = pbk, in.dl_sign[n_dl] = -1;
          if (slot < in.R) in.h_act[hx] = 0;
          if (!dup) {
            in.chg[n_chg] = u, in.chg_bad[n_chg] = 0u, in.chg_tile[n_chg] = n_tiles;
            n_tiles += nt;
            ++n_chg;
          }
          in.preempted[c.np++] = slot < in.R ? slot : 0xFFFFFFFFu;  // a task placed this cycle is reported as NONE (rebalancer.clj:529)
          if (k == 0) first_known = known;
          if (known) in.pre_hosts[c.n_pre_hosts++] = h;
        }
        ++n_dl;
      }
      if (lane == 0) {
        // the job becomes a task of its user on that host, carrying the slave id of the first preempted task (:279-281)
        in.act[pbj] = 1;
        in.dl_pos[n_dl] = pbj, in.dl_sign[n_dl] = 1;
        in.x_host[jb.pj] = h;
        in.x_known[jb.pj] = (len > 0 && first_known) ? 1 : 0;
        in.x_next[jb.pj] = xh;  // joins the host's chain of placed jobs
        in.x_head[h] = jb.pj;
        in.x_cnt[h] = xc + 1u;
        in.hmax_key[h] = ~0ull;  // (the placed job's DRU is not in the host-ordered mirror: the host is evaluated for every job from now on)
        if (items0 + xc + 1u > c.max_items) c.max_items = items0 + xc + 1u;
        if (items0 <= (unsigned)COOK_WAVE && items0 + xc + 1u == (unsigned)COOK_WAVE + 1u) in.big_list[c.n_big++] = h;  // this placement takes the host past 64 items
        in.x_pj[c.n_x] = jb.pj;  // ... and the list of all of them (placement order)
        c.n_x += 1;
        in.spare_c[h] = d.cpus - jb.c;  // rebalancer.clj:302-305
        in.spare_m[h] = d.mem - jb.m;
        in.spare_g[h] = d.gpus - jb.g;
        in.has_spare[h] = 1;
        c.remaining -= 1;
        in.chg_tile[n_chg] = n_tiles;
        c.n_changed = n_chg;
        c.n_tiles = n_tiles;
        c.n_delta = n_dl + 1u;
        *in.ctl = c;
      }
      ++n_dl;  // (the job's own flip; every lane counts alike)
    } else
    if (lane == 0) {
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
        const unsigned slot = base == 0xFFFFFFFFu ? s_pre[k] : in.srt_slot[base + k];
        const unsigned pbk = in.posB[slot];
        in.act[pbk] = 0;
        in.dl_pos[n_dl] = pbk, in.dl_sign[n_dl] = -1;
        ++n_dl;
        if (slot < in.R) in.h_act[in.hidx[pbk]] = 0;
        changed(in.slot_user[slot]);
        in.preempted[c.np++] = slot < in.R ? slot : 0xFFFFFFFFu;  // a task placed this cycle is reported as NONE (rebalancer.clj:529)
        const bool known = slot < in.R ? (in.attrs_cached ? in.attrs_cached[slot] != 0 : true) : in.x_known[slot - in.R] != 0;
        if (k == 0) first_known = known;
        if (known) in.pre_hosts[c.n_pre_hosts++] = h;
      }
      // the job becomes a task of its user on that host, carrying the slave id of the first preempted task (:279-281)
      const unsigned pbj = in.posB[in.R + jb.pj];
      in.act[pbj] = 1;
      in.dl_pos[n_dl] = pbj, in.dl_sign[n_dl] = 1;
      ++n_dl;
      in.x_host[jb.pj] = h;
      in.x_known[jb.pj] = (len > 0 && first_known) ? 1 : 0;
      in.x_next[jb.pj] = in.x_head[h];  // joins the host's chain of placed jobs
      in.x_head[h] = jb.pj;
      in.x_cnt[h] += 1u;
      in.hmax_key[h] = ~0ull;  // (the placed job's DRU is not in the host-ordered mirror: the host is evaluated for every job from now on)
      if (in.hend[h] - in.hstart[h] + in.x_cnt[h] > c.max_items) c.max_items = in.hend[h] - in.hstart[h] + in.x_cnt[h];
      if (in.hend[h] - in.hstart[h] <= (unsigned)COOK_WAVE && in.hend[h] - in.hstart[h] + in.x_cnt[h] == (unsigned)COOK_WAVE + 1u)
        in.big_list[c.n_big++] = h;  // this placement takes the host past 64 items
      in.x_pj[c.n_x] = jb.pj;            // ... and the list of all of them (placement order)
      c.n_x += 1;
      in.spare_c[h] = d.cpus - jb.c;  // rebalancer.clj:302-305
      in.spare_m[h] = d.mem - jb.m;
      in.spare_g[h] = d.gpus - jb.g;
      in.has_spare[h] = 1;
      c.remaining -= 1;
      in.chg_tile[n_chg] = n_tiles;
      c.n_changed = n_chg;
      c.n_tiles = n_tiles;
      c.n_delta = n_dl;
      *in.ctl = c;
    }
  } else if (lane == 0 && pj_next != 0xFFFFFFFFu) {
    in.ctl->n_changed = 0u;  // (the stand-alone rebal_job_prep resets these before the decision)
    in.ctl->n_delta = 0u;
  }
  if (pj_next == 0xFFFFFFFFu) return;
  __threadfence();  // lane 0's writes (active flags, the flip list, the control block) before the whole wave reads them
  n_dl = (unsigned)wave_read_lane((int)n_dl, 0);
  wave_sync();
  rebal_job_prep_dev(in, pj_next, n_dl);
}

// match_kernels.hpp — device side of cook_match: rank-ordered bin-pack placement, i.e. what Cook delegates to
// Fenzo's TaskScheduler.scheduleOnce (scheduler.clj:617-687, 2301-2324) with cpuMemBinPacker fitness (config.clj:108)
// and Cook's hard constraints (constraints.clj).
//
// Semantics (DESIGN.md "placement"): for each job in rank order, among the
// offers that still have room and pass every constraint pick the one with the strictly greatest fitness
//   ((run_cpus + assigned_cpus + job.cpus) / (offer.cpus + run_cpus) + (same for mem)) / 2
// (lowest offer index on ties; the first offer in array order whose fitness exceeds good-enough wins outright), then
// commit the job to it.  Job i+1 sees job i's commitment: the placement is a sequential chain.
#pragma once
#include "common.hpp"

struct MatchIn {
  // jobs (rank order)
  unsigned K;
  const double *j_cpus, *j_mem, *j_gpus;
  const uint32_t *j_gpu_model, *j_group, *j_eq_off, *j_eq_key, *j_eq_val, *j_novel_off, *j_novel_host;
  const int32_t* j_reserved_host;
  const uint32_t* j_ckpt;
  const int64_t* j_est_end;
  const double* j_disk_req;
  const uint32_t* j_disk_type;
  // optional indirection: job k of the match is entry j_index[k] of the job arrays (rank output feeds the match directly)
  const uint32_t* j_index;
  // offers
  unsigned M;
  const double *o_cpus, *o_mem;
  const uint32_t* o_host;
  const uint8_t* o_k8s;
  const uint32_t* o_gpu_model;
  const double* o_gpu_count;
  const uint32_t* o_disk_type;
  const double* o_disk_space;
  unsigned n_attr;
  const uint32_t* o_attr;
  const int32_t *o_max_tasks, *o_num_tasks;
  const uint32_t* o_location;
  const int64_t* o_host_start;
  const double *o_run_cpus, *o_run_mem;
  const int32_t* o_run_count;
  // groups
  unsigned G;
  const uint8_t* g_type;
  const uint32_t* g_attr_key;
  const int32_t* g_min;
  const uint32_t *g_run_off, *g_run_host, *g_run_attr;
  // reserved hosts bitmap (1 bit per host id) and its size in 32-bit words
  const uint32_t* reserved_bits;
  unsigned reserved_words;
  // params
  double good_enough;
  long long host_lifetime_mins;
  // != 0: two offers of the call share a host (then a placement can forbid, for a unique host-placement group, an offer OTHER
  // than the one it was made on, and the placement walk sends every group member through its general path)
  unsigned host_dup;
  // entries per host in the k8s "gpus" / "disk" maps (>= 1; o_gpu_model / o_gpu_count are [M][gpu_slots], model 0 = empty slot)
  unsigned gpu_slots, disk_slots;
  // Fenzo's other additive resources (cookmatch.h cook_jobs.ports / .scalars): has_x == 0 when no job of the call asks for any
  unsigned has_x, n_scal;
  const int32_t *j_ports, *o_ports;
  const double *j_scal[3], *o_scal[3];  // one column per named scalar
};

struct MatchState {
  double *ac, *am;       // [M] resources assigned in this call (touched only by the offer's owner thread)
  int32_t* acount;       // [M]
  int32_t* group_last;   // [G] last job (match index) of the group placed in this call, -1 none
  int32_t* job_prev;     // [K] previous placed job of the same group
  int32_t* job_to_offer; // [K] output
  uint32_t* fail_code;   // [K] output
  unsigned* summary;     // [0] matched count, [1] head matched flag
  int cutoff;            // unique-group check ignores placements by jobs with match index >= cutoff (INT_MAX = none)
  // bit v%64 of alive[v/64]: offer v can still take the smallest job of this call.  Cleared for good once
  // assigned + min job > lease in cpus or mem (placements only add), so the eval loop never looks at the offer again.
  unsigned long long* alive;
  const double* jmin;    // [2] minimum cpus / mem over the jobs of this call
  int32_t* xports;       // [M] ports assigned in this call       } only when MatchIn::has_x; written by the one thread that
  double* xscal;         // [3][M] named scalars assigned in this call } commits a job, with agent-scope accesses
};

// Fenzo's resource fit beyond cpus / mem for (job jj, offer v) under the call's placements so far: ports (the request's port COUNT
// against the free ports of the lease's ranges; scheduler.clj:466, offer.clj:71-73) and each named scalar request as
// used + request > total against the lease's scalar of that name (scheduler.clj:177-189, offer.clj:57-65).
// -> bit 0: ports do not fit, bit 1 + s: named scalar s does not fit. 
static __device__ __forceinline__ unsigned xres_fail_bits(const MatchIn& in, const MatchState& st, unsigned jj, unsigned v) {
  unsigned bits = 0;
  const int jp = in.j_ports ? in.j_ports[jj] : 0;
  if (jp > 0 && (long long)ld_agent(&st.xports[v]) + jp > (long long)(in.o_ports ? in.o_ports[v] : 0)) bits |= 1u;
  _Pragma("unroll") for (unsigned s = 0; s < 3u; ++s) {
    if (s >= in.n_scal) break;
    const double r = in.j_scal[s][jj];
    if (r != r) continue;  // no request under this name
    const double t = in.o_scal[s] ? in.o_scal[s][v] : 0.0;
    if (ld_agent(&st.xscal[(size_t)s * in.M + v]) + r > t) bits |= 2u << s;
  }
  return bits;
}
static __device__ __forceinline__ bool job_has_xres(const MatchIn& in, unsigned jj) {
  bool x = in.j_ports && in.j_ports[jj] > 0;
  _Pragma("unroll") for (unsigned s = 0; s < 3u; ++s) {
    if (s >= in.n_scal) break;
    const double r = in.j_scal[s][jj];
    x = x || r == r;
  }
  return x;
}
// the one thread that commits job jj to offer v
static __device__ __forceinline__ void xres_commit(const MatchIn& in, const MatchState& st, unsigned jj, unsigned v) {
  const int jp = in.j_ports ? in.j_ports[jj] : 0;
  if (jp > 0) st_agent(&st.xports[v], ld_agent(&st.xports[v]) + jp);
  _Pragma("unroll") for (unsigned s = 0; s < 3u; ++s) {
    if (s >= in.n_scal) break;
    const double r = in.j_scal[s][jj];
    if (r == r) st_agent(&st.xscal[(size_t)s * in.M + v], ld_agent(&st.xscal[(size_t)s * in.M + v]) + r);
  }
}

static __device__ __forceinline__ uint32_t offer_attr_val(const MatchIn& in, unsigned v, uint32_t key) {
  if (key == 0xFFFFFFFFu) return in.o_host[v] + 1;  // "HOSTNAME"
  if (key >= in.n_attr || !in.o_attr) return 0;
  return in.o_attr[(size_t)v * in.n_attr + key];
}

// The constraints of constraints.clj for (job jj, offer v), split by what they depend on:
//   static_pass  : job x offer only (novel-host, gpu model/count, disk, EQUALS, estimated completion, checkpoint
//                  locality, rebalancer reservation)                       -> never changes during a match call
//   dyn_pass     : also the number of tasks placed on v in this call (gpu "VM must be empty", max-tasks-per-host)
//   group_pass   : also where the job's cotasks were placed in this call (group host-placement)
static __device__ __forceinline__ bool static_pass(const MatchIn& in, unsigned jj, unsigned v) {
  const unsigned host = in.o_host[v];
  if (in.j_novel_off) {  // novel-host, constraints.clj:68-94
    for (unsigned x = in.j_novel_off[jj]; x < in.j_novel_off[jj + 1]; ++x)
      if (in.j_novel_host[x] == host) return false;
  }
  const double jg = in.j_gpus ? in.j_gpus[jj] : 0.0;
  const bool k8s = in.o_k8s && in.o_k8s[v];
  if (k8s) {  // gpu-host, constraints.clj:122-157 (model / count part)
    if (jg > 0) {
      const double avail = map_get_dev(in.o_gpu_model, in.o_gpu_count, in.gpu_slots, v, in.j_gpu_model ? in.j_gpu_model[jj] : 0u);
      if (!(avail == jg)) return false;
    } else if (map_count_dev(in.o_gpu_model, in.gpu_slots, v) != 0u) {
      return false;
    }
  } else if (!(jg == 0)) {
    return false;
  }
  if (in.j_disk_req && in.j_disk_req[jj] >= 0 && k8s) {  // disk-host, constraints.clj:164-199
    const double space = map_get_dev(in.o_disk_type, in.o_disk_space, in.disk_slots, v, in.j_disk_type[jj]);
    if (!(space >= in.j_disk_req[jj])) return false;
  }
  if (in.j_eq_off) {  // user-defined EQUALS, constraints.clj:356-377
    for (unsigned x = in.j_eq_off[jj]; x < in.j_eq_off[jj + 1]; ++x)
      if (offer_attr_val(in, v, in.j_eq_key[x]) != in.j_eq_val[x]) return false;
  }
  if (in.j_est_end && in.j_est_end[jj] != 0 && in.o_host_start && in.o_host_start[v] >= 0) {  // constraints.clj:385-401
    const long long death = 1000ll * in.o_host_start[v] + 60ll * 1000ll * in.host_lifetime_mins;
    if (!(in.j_est_end[jj] < death)) return false;
  }
  if (in.j_ckpt && in.j_ckpt[jj] != 0) {  // checkpoint-locality, constraints.clj:218-240
    const unsigned loc = in.o_location ? in.o_location[v] : 0u;
    if (loc != in.j_ckpt[jj]) return false;
  }
  if (in.reserved_bits && (host >> 5) < in.reserved_words && ((in.reserved_bits[host >> 5] >> (host & 31)) & 1u)) {
    // rebalancer-reservation, constraints.clj:242-252 + scheduler.clj:645-653
    if (!(in.j_reserved_host && in.j_reserved_host[jj] == (int)host)) return false;
  }
  return true;
}

static __device__ __forceinline__ bool dyn_pass(const MatchIn& in, unsigned jj, unsigned v, int acount_v) {
  const double jg = in.j_gpus ? in.j_gpus[jj] : 0.0;
  if (jg > 0 && in.o_k8s && in.o_k8s[v]) {  // gpu-host: no task (running or assigned this cycle) on the VM
    if ((in.o_run_count ? in.o_run_count[v] : 0) + acount_v != 0) return false;
  }
  if (in.o_max_tasks && in.o_max_tasks[v] >= 0) {  // max-tasks-per-host, constraints.clj:433-456
    if (!((in.o_num_tasks ? in.o_num_tasks[v] : 0) + acount_v < in.o_max_tasks[v])) return false;
  }
  return true;
}

// group host-placement, constraints.clj:586-644
static __device__ __forceinline__ bool group_pass(const MatchIn& in, const MatchState& st, unsigned jj, unsigned v) {
  if (!(in.j_group && in.j_group[jj] != 0xFFFFFFFFu)) return true;
  const unsigned host = in.o_host[v];
  const unsigned g = in.j_group[jj];
  const unsigned type = in.g_type[g];
  if (type == 0) return true;
  const unsigned r0 = in.g_run_off ? in.g_run_off[g] : 0u, r1 = in.g_run_off ? in.g_run_off[g + 1] : 0u;
  const unsigned key = in.g_attr_key[g];
  if (type == 1) {  // unique
    for (unsigned x = r0; x < r1; ++x)
      if (in.g_run_host[x] == host) return false;
    for (int c = ld_agent(&st.group_last[g]); c >= 0; c = ld_agent(&st.job_prev[c]))
      if (c < st.cutoff && in.o_host[ld_agent(&st.job_to_offer[c])] == host) return false;
    return true;
  }
  // frequencies of the attribute over cotask hosts (running ++ placed in this call); nil (0) is a legal value
  const unsigned target = offer_attr_val(in, v, key);
  const unsigned n_run = r1 - r0;
  unsigned n_cyc = 0;
  for (int c = ld_agent(&st.group_last[g]); c >= 0; c = ld_agent(&st.job_prev[c])) ++n_cyc;
  const unsigned total = n_run + n_cyc;
  if (total == 0) return true;
  // value of cotask number x (running first, then this call's in reverse placement order)
  auto val_at = [&](unsigned x) -> unsigned {
    if (x < n_run) return key == 0xFFFFFFFFu ? in.g_run_host[r0 + x] + 1 : in.g_run_attr[r0 + x];
    int c = ld_agent(&st.group_last[g]);
    for (unsigned s = n_run; s < x; ++s) c = ld_agent(&st.job_prev[c]);
    return offer_attr_val(in, (unsigned)ld_agent(&st.job_to_offer[c]), key);
  };
  unsigned tfreq = 0, mn = 0xFFFFFFFFu, mx = 0, distinct = 0;
  for (unsigned a = 0; a < total; ++a) {
    const unsigned va = val_at(a);
    if (va == target) ++tfreq;
    bool first = true;
    unsigned cnt = 0;
    for (unsigned b = 0; b < total; ++b) {
      const unsigned vb = val_at(b);
      if (vb == va) {
        if (b < a) first = false;
        ++cnt;
      }
    }
    if (first) {
      ++distinct;
      mn = cnt < mn ? cnt : mn;
      mx = cnt > mx ? cnt : mx;
    }
  }
  if (type == 2) {  // balanced
    if (tfreq != 0) {
      const unsigned minim = ((unsigned)(in.g_min[g] > 0 ? in.g_min[g] : 0) > distinct) ? 0u : mn;
      if (!(minim == mx || tfreq < mx)) return false;
    }
  } else {  // attribute-equals
    if (tfreq == 0) return false;
  }
  return true;
}

// All static and dynamic constraints for (job jj, offer v).  `acount_v` = tasks placed on v in this call.
static __device__ __forceinline__ bool constraints_pass(const MatchIn& in, const MatchState& st, unsigned jj, unsigned v,
                                                        int acount_v) {
  return static_pass(in, jj, v) && dyn_pass(in, jj, v, acount_v) && group_pass(in, st, jj, v);
}

struct Cand {
  double fit;
  int idx;
};
static __device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b) {  // a strictly better than b
  return a.fit > b.fit || (a.fit == b.fit && a.idx >= 0 && (b.idx < 0 || a.idx < b.idx));
}

// Exact serial placement: ONE workgroup walks the K jobs; for each job every thread evaluates its offers
// (v = tid, tid+THREADS, ...), the workgroup reduces to the winner, and the winner's owner thread commits.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) match_serial(MatchIn in, MatchState st) {
  constexpr int NW = THREADS / COOK_WAVE;
  __shared__ double s_fit[NW];
  __shared__ int s_idx[NW];
  __shared__ int s_ge[NW];
  __shared__ unsigned s_fail[NW];
  __shared__ int s_win;
  const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
  unsigned matched = 0;
  int head = 0;
  for (unsigned k = 0; k < in.K; ++k) {
    const unsigned jj = in.j_index ? in.j_index[k] : k;
    const double c = in.j_cpus[jj], m = in.j_mem[jj];
    Cand best{-1.0, -1};
    int ge_idx = 0x7FFFFFFF;
    unsigned fail = 0;
    for (unsigned v = tid; v < in.M; v += THREADS) {
      const double ac = st.ac[v], am = st.am[v];
      if (ac + c > in.o_cpus[v] || am + m > in.o_mem[v] || (in.has_x && xres_fail_bits(in, st, jj, v) != 0u)) {
        fail |= 1u;
        continue;
      }
      if (!constraints_pass(in, st, jj, v, st.acount[v])) {
        fail |= 2u;
        continue;
      }
      const double rc = in.o_run_cpus ? in.o_run_cpus[v] : 0.0, rm = in.o_run_mem ? in.o_run_mem[v] : 0.0;
      const double fit = ((rc + ac + c) / (in.o_cpus[v] + rc) + (rm + am + m) / (in.o_mem[v] + rm)) / 2.0;
      if (!(fit > 0.0)) {
        fail |= 4u;
        continue;
      }
      if (fit > best.fit) {
        best.fit = fit;
        best.idx = (int)v;
      }
      if (fit > in.good_enough && (int)v < ge_idx) ge_idx = (int)v;
    }
    // wave reduction (butterfly), then across waves through LDS
    for (int d = 32; d >= 1; d >>= 1) {
      Cand o{__shfl_xor(best.fit, d, COOK_WAVE), __shfl_xor(best.idx, d, COOK_WAVE)};
      if (cand_better(o, best)) best = o;
      const int og = __shfl_xor(ge_idx, d, COOK_WAVE);
      ge_idx = og < ge_idx ? og : ge_idx;
      fail |= __shfl_xor(fail, d, COOK_WAVE);
    }
    if (lane == 0) {
      s_fit[w] = best.fit;
      s_idx[w] = best.idx;
      s_ge[w] = ge_idx;
      s_fail[w] = fail;
    }
    __syncthreads();
    if (tid == 0) {
      Cand b{s_fit[0], s_idx[0]};
      int g = s_ge[0];
      unsigned f = s_fail[0];
      for (int q = 1; q < NW; ++q) {
        Cand o{s_fit[q], s_idx[q]};
        if (cand_better(o, b)) b = o;
        g = s_ge[q] < g ? s_ge[q] : g;
        f |= s_fail[q];
      }
      const int win = (g != 0x7FFFFFFF) ? g : b.idx;  // scheduler.clj:2312-2314 early exit at the first good-enough VM
      s_win = win;
      st_agent(&st.job_to_offer[k], win);
      if (st.fail_code) st.fail_code[k] = win >= 0 ? 0u : (f ? f : 8u);
      if (win >= 0) {
        ++matched;
        if (k == 0) head = 1;
        if (in.j_group && in.j_group[jj] != 0xFFFFFFFFu) {
          const unsigned g2 = in.j_group[jj];
          st_agent(&st.job_prev[k], ld_agent(&st.group_last[g2]));
          st_agent(&st.group_last[g2], (int)k);
        }
      }
    }
    __syncthreads();
    const int win = s_win;
    if (win >= 0 && (unsigned)win % THREADS == tid) {  // owner of the offer commits
      st.ac[win] += c;
      st.am[win] += m;
      st.acount[win] += 1;
      if (in.has_x) xres_commit(in, st, jj, (unsigned)win);
    }
    // next iteration's first LDS write happens after its own __syncthreads pair; s_win is re-read only after the next barrier
  }
  if (tid == 0) {
    st.summary[0] = matched;
    st.summary[1] = (matched == 0 || head) ? 1u : 0u;  // scheduler.clj:1495 matched-head-or-no-matches?
  }
}

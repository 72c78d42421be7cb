// explain_kernels.hpp — consumers of the placement's by-products:
//  * "why unscheduled": the placement-failure summary the reference builds from Fenzo's TaskAssignmentResults
//    (fenzo_utils.clj:33-55 summarize-placement-failure; read back by unscheduled.clj:95-110) — per investigated job and per
//    host either the resources that do not fit or the name of the first failing hard constraint — recomputed AFTER a match
//    from its result: the state a job saw = the placements of the jobs ranked before it.
//  * the match-cycle metrics (scheduler.clj:1210-1280 handle-match-cycle-metrics, :547-600 jobs->stats / offers->stats).
#pragma once
#include "common.hpp"
#include "match_kernels.hpp"

// slots of a summary row (cookmatch.h COOK_WHY_*)
constexpr int WHY_CPUS = 0, WHY_MEM = 1, WHY_FITNESS = 2, WHY_CKPT = 3, WHY_EST = 4, WHY_USER = 5, WHY_DISK = 6, WHY_GPU = 7,
              WHY_NOVEL = 8, WHY_MAX_TASKS = 9, WHY_RESERVED = 10, WHY_GROUP_UNIQUE = 11, WHY_SCALAR0 = 14, WHY_PORTS = 17, WHY_SLOTS = 20;

// sort key of a job position = the offer it was placed on; unmatched jobs go behind every offer
__global__ void __launch_bounds__(256) explain_offer_keys(const int32_t* __restrict__ j2o, unsigned K, unsigned M, uint64_t* __restrict__ key) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) {
    const int v = j2o[k];
    key[k] = (v >= 0 && (unsigned)v < M) ? (unsigned)v : M;
  }
}

// group host-placement (constraints.clj:586-644) as job `cutoff` saw it: cotasks placed by jobs ranked before it only
static __device__ __forceinline__ bool group_pass_at(const MatchIn& in, const MatchState& st, unsigned jj, unsigned v, int cutoff) {
  if (!(in.j_group && in.j_group[jj] != 0xFFFFFFFFu)) return true;
  const unsigned host = in.o_host[v];
  const unsigned g = in.j_group[jj];
  const unsigned type = in.g_type[g];
  if (type == 0) return true;
  const unsigned r0 = in.g_run_off ? in.g_run_off[g] : 0u, r1 = in.g_run_off ? in.g_run_off[g + 1] : 0u;
  const unsigned key = in.g_attr_key[g];
  auto first = [&]() {
    int c = st.group_last[g];
    while (c >= 0 && c >= cutoff) c = st.job_prev[c];
    return c;
  };
  auto next = [&](int c) {
    c = st.job_prev[c];
    while (c >= 0 && c >= cutoff) c = st.job_prev[c];
    return c;
  };
  if (type == 1) {  // unique
    for (unsigned x = r0; x < r1; ++x)
      if (in.g_run_host[x] == host) return false;
    for (int c = first(); c >= 0; c = next(c))
      if (in.o_host[st.job_to_offer[c]] == host) return false;
    return true;
  }
  const unsigned target = offer_attr_val(in, v, key);
  const unsigned n_run = r1 - r0;
  unsigned n_cyc = 0;
  for (int c = first(); c >= 0; c = next(c)) ++n_cyc;
  const unsigned total = n_run + n_cyc;
  if (total == 0) return true;
  auto val_at = [&](unsigned x) -> unsigned {
    if (x < n_run) return key == 0xFFFFFFFFu ? in.g_run_host[r0 + x] + 1 : in.g_run_attr[r0 + x];
    int c = first();
    for (unsigned s = n_run; s < x; ++s) c = next(c);
    return offer_attr_val(in, (unsigned)st.job_to_offer[c], key);
  };
  unsigned tfreq = 0, mn = 0xFFFFFFFFu, mx = 0, distinct = 0;
  for (unsigned a = 0; a < total; ++a) {
    const unsigned va = val_at(a);
    if (va == target) ++tfreq;
    bool fst = true;
    unsigned cnt = 0;
    for (unsigned b = 0; b < total; ++b) {
      const unsigned vb = val_at(b);
      if (vb == va) {
        if (b < a) fst = false;
        ++cnt;
      }
    }
    if (fst) {
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

// The hard constraints in the order Fenzo walks them = (into (list) constraints) of make-task-request (scheduler.clj:493-501):
// checkpoint-locality, estimated-completion, user-defined, disk-host, gpu-host, novel-host (constraints.clj:459-464 reversed),
// max_tasks_per_host, rebalancer-reservation, group.  -> slot of the first failing one, -1 if all pass.
static __device__ __forceinline__ int first_failed_constraint(const MatchIn& in, const MatchState& st, unsigned jj, unsigned v, int acount_v,
                                                              int cutoff) {
  const unsigned host = in.o_host[v];
  const bool k8s = in.o_k8s && in.o_k8s[v];
  if (in.j_ckpt && in.j_ckpt[jj] != 0) {
    const unsigned loc = in.o_location ? in.o_location[v] : 0u;
    if (loc != in.j_ckpt[jj]) return WHY_CKPT;
  }
  if (in.j_est_end && in.j_est_end[jj] != 0 && in.o_host_start && in.o_host_start[v] >= 0) {
    const long long death = 1000ll * in.o_host_start[v] + 60ll * 1000ll * in.host_lifetime_mins;
    if (!(in.j_est_end[jj] < death)) return WHY_EST;
  }
  if (in.j_eq_off) {
    for (unsigned x = in.j_eq_off[jj]; x < in.j_eq_off[jj + 1]; ++x)
      if (offer_attr_val(in, v, in.j_eq_key[x]) != in.j_eq_val[x]) return WHY_USER;
  }
  if (in.j_disk_req && in.j_disk_req[jj] >= 0 && k8s) {
    const double space = map_get_dev(in.o_disk_type, in.o_disk_space, in.disk_slots, v, in.j_disk_type[jj]);
    if (!(space >= in.j_disk_req[jj])) return WHY_DISK;
  }
  {
    const double jg = in.j_gpus ? in.j_gpus[jj] : 0.0;
    if (k8s) {
      if (jg > 0) {
        const double avail = map_get_dev(in.o_gpu_model, in.o_gpu_count, in.gpu_slots, v, in.j_gpu_model ? in.j_gpu_model[jj] : 0u);
        const int on_vm = (in.o_run_count ? in.o_run_count[v] : 0) + acount_v;
        if (!(avail == jg && on_vm == 0)) return WHY_GPU;
      } else if (map_count_dev(in.o_gpu_model, in.gpu_slots, v) != 0u) {
        return WHY_GPU;
      }
    } else if (!(jg == 0)) {
      return WHY_GPU;
    }
  }
  if (in.j_novel_off) {
    for (unsigned x = in.j_novel_off[jj]; x < in.j_novel_off[jj + 1]; ++x)
      if (in.j_novel_host[x] == host) return WHY_NOVEL;
  }
  if (in.o_max_tasks && in.o_max_tasks[v] >= 0) {
    if (!((in.o_num_tasks ? in.o_num_tasks[v] : 0) + acount_v < in.o_max_tasks[v])) return WHY_MAX_TASKS;
  }
  if (in.reserved_bits && (host >> 5) < in.reserved_words && ((in.reserved_bits[host >> 5] >> (host & 31)) & 1u)) {
    if (!(in.j_reserved_host && in.j_reserved_host[jj] == (int)host)) return WHY_RESERVED;
  }
  if (!group_pass_at(in, st, jj, v, cutoff)) return WHY_GROUP_UNIQUE - 1 + (int)in.g_type[in.j_group[jj]];
  return -1;
}

// grid = (offer blocks, investigated jobs).  Thread = one offer v for job position k = pos[blockIdx.y]: it rebuilds what k saw on
// v by folding, in rank order, the jobs placed on v before k (plist = job positions stably partitioned by offer), then classifies.
__global__ void __launch_bounds__(256) explain_classify(MatchIn in, MatchState st, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ plist,
                                                        const uint32_t* __restrict__ ostart, const uint32_t* __restrict__ oend,
                                                        uint32_t* __restrict__ counts) {
  __shared__ unsigned s_cnt[WHY_SLOTS];
  if (threadIdx.x < WHY_SLOTS) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const unsigned k = pos[blockIdx.y];
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < in.M && k < in.K) {
    const unsigned jj = in.j_index ? in.j_index[k] : k;
    const double c = in.j_cpus[jj], m = in.j_mem[jj];
    double ac = 0.0, am = 0.0, as[3] = {0.0, 0.0, 0.0};
    int acount = 0;
    long long aports = 0;
    for (unsigned i = ostart[v]; i < oend[v]; ++i) {
      const unsigned kk = plist[i];
      if (kk >= k) break;
      const unsigned j2 = in.j_index ? in.j_index[kk] : kk;
      ac += in.j_cpus[j2];  // the placement accumulated them in this very order
      am += in.j_mem[j2];
      ++acount;
      if (in.has_x) {
        aports += in.j_ports ? in.j_ports[j2] : 0;
        _Pragma("unroll") for (unsigned s = 0; s < 3u; ++s) {
          if (s >= in.n_scal) break;
          const double r = in.j_scal[s][j2];
          if (r == r) as[s] += r;
        }
      }
    }
    const bool fc = ac + c > in.o_cpus[v], fm = am + m > in.o_mem[v];
    unsigned fx = 0;
    if (in.has_x) {
      const int jp = in.j_ports ? in.j_ports[jj] : 0;
      if (jp > 0 && aports + jp > (long long)(in.o_ports ? in.o_ports[v] : 0)) fx |= 1u;
      _Pragma("unroll") for (unsigned s = 0; s < 3u; ++s) {
        if (s >= in.n_scal) break;
        const double r = in.j_scal[s][jj];
        if (r == r && as[s] + r > (in.o_scal[s] ? in.o_scal[s][v] : 0.0)) fx |= 2u << s;
      }
    }
    if (fc || fm || fx) {
      if (fc) atomicAdd(&s_cnt[WHY_CPUS], 1u);
      if (fm) atomicAdd(&s_cnt[WHY_MEM], 1u);
      if (fx & 1u) atomicAdd(&s_cnt[WHY_PORTS], 1u);
      for (unsigned s = 0; s < 3u; ++s)
        if ((fx >> (1u + s)) & 1u) atomicAdd(&s_cnt[WHY_SCALAR0 + s], 1u);
    } else {
      const int why = first_failed_constraint(in, st, jj, v, acount, (int)k);
      if (why >= 0) {
        atomicAdd(&s_cnt[why], 1u);
      } else {
        const double rc = in.o_run_cpus ? in.o_run_cpus[v] : 0.0, rm = in.o_run_mem ? in.o_run_mem[v] : 0.0;
        const double fit = ((rc + ac + c) / (in.o_cpus[v] + rc) + (rm + am + m) / (in.o_mem[v] + rm)) / 2.0;
        if (!(fit > 0.0)) atomicAdd(&s_cnt[WHY_FITNESS], 1u);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < WHY_SLOTS && s_cnt[threadIdx.x]) atomicAdd(&counts[(size_t)blockIdx.y * WHY_SLOTS + threadIdx.x], s_cnt[threadIdx.x]);
}

// ---- match-cycle metrics (scheduler.clj:1210-1280) ------------------------------------------------------------------------
struct ResourceStatsDev {  // cook_resource_stats
  double total_cpus, total_mem, p50_cpus, p95_cpus, p100_cpus, p50_mem, p95_mem, p100_mem;
  uint32_t largest_by_cpus, largest_by_mem;
};

// the considerable jobs' resource columns in match order (jobs->resource-maps, scheduler.clj:511-545) + their sort keys
__global__ void __launch_bounds__(256) metrics_gather_jobs(MatchIn in, double* __restrict__ cpus, double* __restrict__ mem,
                                                           uint64_t* __restrict__ kc, uint64_t* __restrict__ km) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= in.K) return;
  const unsigned jj = in.j_index ? in.j_index[k] : k;
  const double c = in.j_cpus[jj], m = in.j_mem[jj];
  cpus[k] = c;
  mem[k] = m;
  kc[k] = f64_key(c);
  km[k] = f64_key(m);
}
__global__ void __launch_bounds__(256) metrics_keys(const double* __restrict__ a, const double* __restrict__ b, unsigned n,
                                                    uint64_t* __restrict__ ka, uint64_t* __restrict__ kb) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    ka[i] = f64_key(a[i]);
    kb[i] = f64_key(b[i]);
  }
}
struct LoadPair {  // (cpus, mem) as a tracked SumU4 element
  const double *a, *b;
  __device__ __forceinline__ SumU4 operator()(unsigned i) const { return SumU4{1.0, a[i], b[i], 0.0, 0u}; }
};
// nearest-rank percentiles of a stably sorted permutation (task_stats.clj:59-80): index ceil(p n / 100) - 1 in exact
// arithmetic, as the reference's ratio arithmetic gives; :largest-by = the LAST of the stable sort (scheduler.clj:563-568)
__global__ void metrics_pick(const uint32_t* __restrict__ perm, const double* __restrict__ val, unsigned n, double* __restrict__ p50,
                             double* __restrict__ p95, double* __restrict__ p100, uint32_t* __restrict__ largest) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned long long nn = n;
  *p50 = val[perm[(unsigned)((50ull * nn + 99ull) / 100ull) - 1u]];
  *p95 = val[perm[(unsigned)((95ull * nn + 99ull) / 100ull) - 1u]];
  *p100 = val[perm[n - 1]];
  *largest = perm[n - 1];
}
// :totals (reduce (partial merge-with +)) in collection order.  The tracked scan formed EVERY prefix; if none of its additions
// rounded, every prefix is exact, hence equal to the left-to-right sum, and the last element is the answer.  Otherwise the
// in-order fold below.
constexpr int MT_TILE = 2048;
__global__ void __launch_bounds__(1024) metrics_totals(const SumU4* __restrict__ scan, const double* __restrict__ a,
                                                       const double* __restrict__ b, unsigned n, double* __restrict__ ta,
                                                       double* __restrict__ tb) {
  __shared__ unsigned s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  unsigned bad = 0;
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x) bad |= scan[i].bad;
  if (bad) atomicOr(&s_bad, 1u);
  __syncthreads();
  if (!s_bad) {
    if (threadIdx.x == 0) {
      *ta = scan[n - 1].cpus;
      *tb = scan[n - 1].mem;
    }
    return;
  }
  // the in-order fold: the workgroup stages tiles of both columns in LDS, wave 0 / 1 fold one column each (one fp64 add
  // latency per element; LDS broadcast reads)
  __shared__ double s_val[2][MT_TILE];
  const unsigned w = wave_id();
  double acc = -0.0;
  for (unsigned base = 0; base < n; base += MT_TILE) {
    const unsigned cnt = n - base < (unsigned)MT_TILE ? n - base : (unsigned)MT_TILE;
    for (unsigned x = threadIdx.x; x < 2u * MT_TILE; x += blockDim.x) {
      const unsigned col = x / MT_TILE, i = x % MT_TILE;
      s_val[col][i] = i < cnt ? (col ? b[base + i] : a[base + i]) : 0.0;
    }
    __syncthreads();
    if (w < 2) {
      acc = fold_lds_in_order(&s_val[w][0], cnt, acc);
    }
    __syncthreads();
  }
  if (w < 2 && lane_id() == 0) *(w ? tb : ta) = acc;
}
// frequencies of users over the considerable / matched jobs (scheduler.clj:1216-1227), gpus per model over the jobs
__global__ void __launch_bounds__(256) metrics_job_counts(MatchIn in, const int32_t* __restrict__ j2o, const uint32_t* __restrict__ j_user,
                                                          unsigned n_users, uint32_t* __restrict__ user_considerable,
                                                          uint32_t* __restrict__ user_matched, unsigned n_models,
                                                          unsigned long long* __restrict__ job_gpus_by_model) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= in.K) return;
  const unsigned jj = in.j_index ? in.j_index[k] : k;
  if (j_user) {
    const unsigned u = j_user[jj];
    if (u < n_users) {
      atomicAdd(&user_considerable[u], 1u);
      if (j2o[k] >= 0) atomicAdd(&user_matched[u], 1u);
    }
  }
  if (job_gpus_by_model && in.j_gpus) {
    const double g = in.j_gpus[jj];
    const unsigned md = in.j_gpu_model ? in.j_gpu_model[jj] : 0u;
    // job gpu counts are whole numbers (schema: :resource.type/gpus is a count): summed exactly as integers
    if (g > 0 && md <= n_models) atomicAdd(&job_gpus_by_model[md], (unsigned long long)g);
  }
}
// offers-scheduled = leases Fenzo used (scheduler.clj:1372-1374); "gpus/<model>" totals of offers->resource-maps (tools.clj:1032-1058)
__global__ void __launch_bounds__(256) metrics_offer_counts(const int32_t* __restrict__ acount, unsigned M, unsigned* __restrict__ scheduled,
                                                            const uint32_t* __restrict__ o_gpu_model, const double* __restrict__ o_gpu_count,
                                                            unsigned gpu_slots, unsigned n_models,
                                                            unsigned long long* __restrict__ offer_gpus_by_model) {
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  const bool used = v < M && acount[v] > 0;
  const unsigned long long b = __ballot(used);
  if (lane_id() == 0 && b) atomicAdd(scheduled, (unsigned)__popcll(b));
  if (v < M && offer_gpus_by_model && o_gpu_model && o_gpu_count) {
    for (unsigned q = 0; q < gpu_slots; ++q) {  // every entry of the host's map (tools.clj:1032-1058)
      const unsigned md = o_gpu_model[(size_t)v * gpu_slots + q];
      // whole numbers (possibly negative on an over-committed node): summed exactly as two's-complement integers
      if (md != 0 && md <= n_models) atomicAdd(&offer_gpus_by_model[md], (unsigned long long)(long long)o_gpu_count[(size_t)v * gpu_slots + q]);
    }
  }
}

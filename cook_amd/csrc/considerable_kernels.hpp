// considerable_kernels.hpp — device side of cook_considerable: the filters between rank and match
// (scheduler.clj:729-762 pending-jobs->considerable-jobs; tools.clj:903-973 filter-pending-jobs-for-quota;
//  tools.clj:654-668 filter-sequential: the state advances on rejected jobs too).
//
// The reference threads per-user state through the queue sequentially.  Here the queue positions are stably partitioned
// by user (one radix sort), the per-user usage prefixes are ONE segmented scan seeded with the users' running usage (with
// the rank path's exactness fix-up), the rate-limit index is a segmented count of the survivors, and the pool-quota
// filter is the queue-order seeded scan of the rank path (queue_filter_quota).  Everything is compaction in between.
#pragma once
#include "common.hpp"
#include "scan.hpp"

COOK_KERNEL void cons_user_keys(const uint32_t* __restrict__ user, unsigned n, uint64_t* __restrict__ key) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) key[i] = user[i];
}

// queue arrays of a cycle: position q of the ranked queue is pending job pend_ord[ranked[q]]
COOK_KERNEL void cons_gather_queue(const uint32_t* __restrict__ ranked, const uint32_t* __restrict__ pend_ord,
                                                         unsigned n, const double* __restrict__ j_cpus, const double* __restrict__ j_mem,
                                                         const double* __restrict__ j_gpus, const uint32_t* __restrict__ j_user,
                                                         const uint8_t* __restrict__ elig_by_pending, double* __restrict__ q_cpus,
                                                         double* __restrict__ q_mem, double* __restrict__ q_gpus,
                                                         uint32_t* __restrict__ q_user, uint8_t* __restrict__ q_elig) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const unsigned jj = pend_ord[ranked[q]];
  q_cpus[q] = j_cpus[jj];
  q_mem[q] = j_mem[jj];
  q_gpus[q] = j_gpus ? j_gpus[jj] : 0.0;
  q_user[q] = j_user[jj];
  q_elig[q] = elig_by_pending ? elig_by_pending[jj] : 1;
}

// per-user order of the queue: usage of each job, segment heads and bounds
COOK_KERNEL void cons_gather(const uint32_t* __restrict__ permU, unsigned n, const uint32_t* __restrict__ user,
                                                   const double* __restrict__ cpus, const double* __restrict__ mem,
                                                   const double* __restrict__ gpus, uint32_t* __restrict__ g_user,
                                                   SumU4* __restrict__ g_use, uint8_t* __restrict__ head,
                                                   uint32_t* __restrict__ seg_start, uint32_t* __restrict__ seg_end) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned t = permU[i];
  const unsigned u = user[t];
  g_user[i] = u;
  g_use[i] = SumU4{1.0, cpus[t], mem[t], gpus ? gpus[t] : 0.0, 0u};
  const bool h = (i == 0) || (user[permU[i - 1]] != u);
  head[i] = h ? 1 : 0;
  if (h) seg_start[u] = i;
  if (i == n - 1 || user[permU[i + 1]] != u) seg_end[u] = i + 1;
}

struct LoadUserSeeded {  // (merge-with + job-usage usage[user]), tools.clj:908: the first job of a user adds to the user's running usage
  const SumU4* use;
  const uint8_t* head;
  const uint32_t* g_user;
  const double *uc, *ucpus, *umem, *ugpus;
  __device__ __forceinline__ SumU4 operator()(unsigned i) const {
    const SumU4 x = use[i];
    if (!head[i]) return x;
    const unsigned u = g_user[i];
    return combine(x, SumU4{uc[u], ucpus[u], umem[u], ugpus[u], 0u});
  }
};

COOK_KERNEL void cons_fix_inexact(const SumU4* __restrict__ g_use, SumU4* __restrict__ pre,
                                                        const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_end,
                                                        const uint32_t* __restrict__ inexact_user, unsigned n_users,
                                                        const double* __restrict__ uc, const double* __restrict__ ucpus,
                                                        const double* __restrict__ umem, const double* __restrict__ ugpus) {
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_users || !inexact_user[u]) return;
  double c = uc[u], cp = ucpus[u], m = umem[u], g = ugpus[u];
  for (unsigned i = seg_start[u]; i < seg_end[u]; ++i) {
    const SumU4 x = g_use[i];
    c = x.count + c;  // job usage first, as merge-with + does
    cp = x.cpus + cp;
    m = x.mem + m;
    g = x.gpus + g;
    pre[i] = SumU4{c, cp, m, g, 0u};
  }
}

COOK_KERNEL void cons_user_quota_flag(const SumU4* __restrict__ pre, const uint32_t* __restrict__ g_user, unsigned n,
                                                            const double* __restrict__ q_count, const double* __restrict__ q_cpus,
                                                            const double* __restrict__ q_mem, const double* __restrict__ q_gpus,
                                                            int* __restrict__ flag) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned u = g_user[i];
  const SumU4 s = pre[i];
  flag[i] = below_quota4(q_count[u], q_cpus[u], q_mem[u], q_gpus[u], Usage4{s.count, s.cpus, s.mem, s.gpus}) ? 1 : 0;
}

// tools.clj:935-955: the n-th job of a user that reaches this stage is rate limited iff n > tokens-left; it is dropped only
// when the limiter is enforcing.  Writes the verdict back in QUEUE order.
COOK_KERNEL void cons_rate_limit(const int* __restrict__ flag1, const SumI* __restrict__ idx1,
                                                       const uint32_t* __restrict__ g_user, const uint32_t* __restrict__ permU, unsigned n,
                                                       const int64_t* __restrict__ tokens, int enforce, int* __restrict__ keep_q,
                                                       uint32_t* __restrict__ rate_limited, uint32_t* __restrict__ passed) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int keep = 0;
  if (flag1[i]) {
    const unsigned u = g_user[i];
    const bool limited = tokens ? ((long long)idx1[i].v > (long long)tokens[u]) : false;
    if (limited)
      atomicAdd(&rate_limited[u], 1u);
    else
      atomicAdd(&passed[u], 1u);
    keep = (limited && enforce) ? 0 : 1;
  }
  keep_q[permU[i]] = keep;
}

// survivors in queue order -> (queue position, usage) lists for the pool-quota scan
COOK_KERNEL void cons_compact_queue(const int* __restrict__ flag, const SumI* __restrict__ incl, unsigned n,
                                                          const double* __restrict__ cpus, const double* __restrict__ mem,
                                                          const double* __restrict__ gpus, uint32_t* __restrict__ qitem,
                                                          SumU4* __restrict__ quse, unsigned* __restrict__ len_out) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  if (flag[q]) {
    const unsigned o = (unsigned)incl[q].v - 1;
    qitem[o] = q;
    quse[o] = SumU4{1.0, cpus[q], mem[q], gpus ? gpus[q] : 0.0, 0u};
  }
  if (q == n - 1) *len_out = (unsigned)incl[q].v;
}

COOK_KERNEL void cons_eligible_flag(const uint32_t* __restrict__ qitem, unsigned len, const uint8_t* __restrict__ elig,
                                                          int* __restrict__ flag) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < len) flag[q] = elig[qitem[q]] ? 1 : 0;
}

// tools.clj:966 pool-usage = sum of the users' usage (user-id order); one workgroup, exactness tracked
COOK_KERNEL void cons_pool_usage(const double* __restrict__ uc, const double* __restrict__ ucpus,
                                                        const double* __restrict__ umem, const double* __restrict__ ugpus, unsigned n,
                                                        SumU4* __restrict__ out) {
  __shared__ SumU4 ws[1024 / COOK_WAVE];
  SumU4 acc = SumU4::zero();
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x) acc = combine(acc, SumU4{uc[i], ucpus[i], umem[i], ugpus[i], 0u});
  for (int d = 32; d >= 1; d >>= 1) {
    SumU4 o;
    o.count = __shfl_xor(acc.count, d, COOK_WAVE);
    o.cpus = __shfl_xor(acc.cpus, d, COOK_WAVE);
    o.mem = __shfl_xor(acc.mem, d, COOK_WAVE);
    o.gpus = __shfl_xor(acc.gpus, d, COOK_WAVE);
    o.bad = __shfl_xor(acc.bad, d, COOK_WAVE);
    acc = combine(acc, o);
  }
  if (lane_id() == 0) ws[wave_id()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    SumU4 t = ws[0];
    for (unsigned k = 1; k < blockDim.x / COOK_WAVE; ++k) t = combine(t, ws[k]);
    if (t.bad) {
      t = SumU4::zero();
      for (unsigned i = 0; i < n; ++i) {
        if (i == 0) {
          t = SumU4{uc[0], ucpus[0], umem[0], ugpus[0], 0u};
        } else {
          t.count += uc[i];
          t.cpus += ucpus[i];
          t.mem += umem[i];
          t.gpus += ugpus[i];
        }
      }
    }
    *out = t;
  }
}

// job k of the match = pending job pend_ord[ranked[cons_idx[k]]]
COOK_KERNEL void cons_job_index(const uint32_t* __restrict__ cons_idx, const uint32_t* __restrict__ ranked,
                                                      const uint32_t* __restrict__ pend_ord, unsigned k, uint32_t* __restrict__ j_index) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) j_index[i] = pend_ord[ranked[cons_idx[i]]];
}

// rank_kernels.hpp — device side of cook_rank: the reference's rank cycle for one pool
// (scheduler.clj:2057-2091 sort-jobs-by-dru-helper, dru.clj:50-126, scheduler.clj:2134-2157, 2198-2229).
//
// Index spaces:  A = caller's task index;  B = position in per-user order (sorted by user, then feature vector);
//                C = position in global DRU order.
#pragma once
#include "common.hpp"
#include "scan.hpp"

// ---- what a rank run wants cleared or preset, in ONE launch (each memset is a launch: the stage is bound by their number) -----------
// scratch64[0..6] = all ones (the key minima and the words' agreeing bits), counters[0..n_counters) = 0, per-user words = 0
COOK_KERNEL void rank_init(unsigned long long* __restrict__ scratch64, unsigned* __restrict__ counters, unsigned n_counters,
                                                 uint32_t* __restrict__ inexact_user, uint32_t* __restrict__ seg_end, unsigned n_users,
                                                 unsigned* __restrict__ tie_ctl, unsigned tie_ctl_words, unsigned nblk /* blocks of this launch */) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 7) scratch64[i] = ~0ull;
  if (i < n_counters) counters[i] = 0u;
  if (i < tie_ctl_words) tie_ctl[i] = 0u;
  for (unsigned u = i; u < n_users; u += nblk * blockDim.x) {
    inexact_user[u] = 0u;
    seg_end[u] = 0u;  // stays 0 for a user without tasks (rank_gather writes the others): cook_rank_user_usage reads it as "absent"
  }
}

// ---- A.2 per-user order keys (tools.clj:614-641) --------------------------------------------------------
// mins[0] = min start over running, mins[1] = min job id over pending, mins[2] = min task id over running
COOK_KERNEL void rank_key_mins(const int64_t* __restrict__ start_ms, const int64_t* __restrict__ task_id,
                                                     const int64_t* __restrict__ job_id, const uint8_t* __restrict__ pending,
                                                     unsigned n, unsigned long long* __restrict__ mins /*[3] as i64_key*/, unsigned nblk) {
  unsigned long long m0 = ~0ull, m1 = ~0ull, m2 = ~0ull;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nblk * blockDim.x) {
    if (pending[i]) {
      const unsigned long long k = i64_key(job_id[i]);
      m1 = k < m1 ? k : m1;
    } else {
      const unsigned long long a = i64_key(start_ms[i]), b = i64_key(task_id[i]);
      m0 = a < m0 ? a : m0;
      m2 = b < m2 ? b : m2;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long a = __shfl_xor(m0, d, COOK_WAVE), b = __shfl_xor(m1, d, COOK_WAVE), c = __shfl_xor(m2, d, COOK_WAVE);
    m0 = a < m0 ? a : m0;
    m1 = b < m1 ? b : m1;
    m2 = c < m2 ? c : m2;
  }
  // one set of atomics per BLOCK (a few dozen blocks): thousands of same-address atomics from every wave cost 96 us for 175k tasks
  __shared__ unsigned long long s_m[3][256 / COOK_WAVE];
  if (lane_id() == 0) s_m[0][wave_id()] = m0, s_m[1][wave_id()] = m1, s_m[2][wave_id()] = m2;
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned long long m = s_m[threadIdx.x][0];
    for (unsigned w = 1; w < blockDim.x / COOK_WAVE; ++w) m = s_m[threadIdx.x][w] < m ? s_m[threadIdx.x][w] : m;
    atomicMin(&mins[threadIdx.x], m);
  }
}

// Three key words, most significant first.  Comparing (w0,w1,w2) lexicographically == comparing
// [user, -priority, start|MAX, task|nil, job] (pending tasks all share start=MAX and task=nil, so they order by job id;
// running tasks have unique task ids, so the job id never decides).  Subtracting the per-class minimum keeps the number
// of varying bytes (= radix passes) small.
COOK_KERNEL void rank_build_keys(const uint32_t* __restrict__ user, const int32_t* __restrict__ priority,
                                                       const int64_t* __restrict__ start_ms, const int64_t* __restrict__ task_id,
                                                       const int64_t* __restrict__ job_id, const uint8_t* __restrict__ pending,
                                                       unsigned n, const unsigned long long* __restrict__ mins,
                                                       uint64_t* __restrict__ w0, uint64_t* __restrict__ w1, uint64_t* __restrict__ w2,
                                                       unsigned long long* __restrict__ same /*[3], preset to all ones*/) {
  // same[k] keeps the bits of word k on which every task agrees with task 0 (the radix passes skip them, sort.hpp)
  auto words = [&](unsigned i, uint64_t& a, uint64_t& b, uint64_t& c) {
    const bool p = pending[i] != 0;
    const uint32_t np = ((uint32_t)(0x40000000 - priority[i])) & 0x7FFFFFFFu;  // -priority, biased (|priority| < 2^30)
    a = ((uint64_t)user[i] << 32) | ((uint64_t)np << 1) | (p ? 1u : 0u);
    b = p ? (i64_key(job_id[i]) - mins[1]) : (i64_key(start_ms[i]) - mins[0]);
    c = p ? 0ull : (i64_key(task_id[i]) - mins[2]);
  };
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t r0, r1, r2;
  words(0, r0, r1, r2);
  unsigned long long d0 = 0, d1 = 0, d2 = 0;
  if (i < n) {
    uint64_t a, b, c;
    words(i, a, b, c);
    w0[i] = a, w1[i] = b, w2[i] = c;
    d0 = a ^ r0, d1 = b ^ r1, d2 = c ^ r2;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    d0 |= __shfl_xor(d0, d, COOK_WAVE);
    d1 |= __shfl_xor(d1, d, COOK_WAVE);
    d2 |= __shfl_xor(d2, d, COOK_WAVE);
  }
  __shared__ unsigned long long s_d[3][256 / COOK_WAVE];  // one set of atomics per block
  if (lane_id() == 0) s_d[0][wave_id()] = d0, s_d[1][wave_id()] = d1, s_d[2][wave_id()] = d2;
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned long long m = 0ull;
    for (unsigned w = 0; w < blockDim.x / COOK_WAVE; ++w) m |= s_d[threadIdx.x][w];
    if (m) atomicAnd(&same[threadIdx.x], ~m);
  }
}

// ---- gather into per-user order, segment heads and bounds ------------------------------------------------
COOK_KERNEL void rank_gather(const uint32_t* __restrict__ permB, unsigned n, const uint32_t* __restrict__ user,
                                                   const double* __restrict__ cpus, const double* __restrict__ mem,
                                                   const double* __restrict__ gpus, const uint8_t* __restrict__ pending,
                                                   uint32_t* __restrict__ s_user, SumU4* __restrict__ s_use,
                                                   uint8_t* __restrict__ s_pending, uint8_t* __restrict__ head,
                                                   uint32_t* __restrict__ seg_start, uint32_t* __restrict__ seg_end) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned t = permB[i];
  const unsigned u = user[t];
  s_user[i] = u;
  s_use[i] = SumU4{1.0, cpus[t], mem[t], gpus ? gpus[t] : 0.0, 0u};
  s_pending[i] = pending[t];
  const bool h = (i == 0) || (user[permB[i - 1]] != u);
  head[i] = h ? 1 : 0;
  if (h) seg_start[u] = i;
  if (i == n - 1 || user[permB[i + 1]] != u) seg_end[u] = i + 1;
}

struct LoadU4 {
  const SumU4* p;
  __device__ __forceinline__ SumU4 operator()(unsigned i) const { return p[i]; }
};
struct LoadI {
  const int* p;
  __device__ __forceinline__ SumI operator()(unsigned i) const { return SumI{p[i]}; }
};

// users whose prefix sums involved an inexact addition
COOK_KERNEL void rank_mark_inexact(const SumU4* __restrict__ pre, const uint32_t* __restrict__ s_user,
                                                         unsigned n, uint32_t* __restrict__ inexact_user) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && pre[i].bad) inexact_user[s_user[i]] = 1u;
}
// ... recomputed left to right exactly as the reference's `reductions` (dru.clj:43-48); one thread per flagged user.
COOK_KERNEL void rank_fix_inexact(const SumU4* __restrict__ s_use, SumU4* __restrict__ pre,
                                                        const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_end,
                                                        const uint32_t* __restrict__ inexact_user, unsigned n_users) {
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_users || !inexact_user[u]) return;
  SumU4 acc = SumU4::zero();
  for (unsigned i = seg_start[u]; i < seg_end[u]; ++i) {
    const SumU4 x = s_use[i];
    if (i == seg_start[u]) {
      acc = x;
    } else {
      acc.count += x.count;
      acc.cpus += x.cpus;
      acc.mem += x.mem;
      acc.gpus += x.gpus;
    }
    acc.bad = 0;
    pre[i] = acc;
  }
}

// ---- A.3 limiter (scheduler.clj:2057-2071) and A.4 DRU (dru.clj:50-80) ------------------------------------
COOK_KERNEL void rank_over_flag(const SumU4* __restrict__ pre, const uint32_t* __restrict__ s_user, unsigned n,
                                                      const double* __restrict__ q_count, const double* __restrict__ q_cpus,
                                                      const double* __restrict__ q_mem, const double* __restrict__ q_gpus,
                                                      int* __restrict__ over) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned u = s_user[i];
  const SumU4 s = pre[i];
  const Usage4 us{s.count, s.cpus, s.mem, s.gpus};
  over[i] = below_quota4(q_count[u], q_cpus[u], q_mem[u], q_gpus[u], us) ? 0 : 1;
}

// keep while the running count of over-quota prefixes <= max-over-quota-jobs; score the survivors.
COOK_KERNEL void rank_score(const SumU4* __restrict__ pre, const SumI* __restrict__ over_cnt,
                                                  const uint32_t* __restrict__ s_user, unsigned n, int limit, int dru_mode,
                                                  const double* __restrict__ div_cpus, const double* __restrict__ div_mem,
                                                  const double* __restrict__ div_gpus, double* __restrict__ dru,
                                                  uint64_t* __restrict__ dkey, uint8_t* __restrict__ keep,
                                                  unsigned* __restrict__ counters /*[0]=n_kept, [1]=equal-run violations*/,
                                                  unsigned long long* __restrict__ or_and /*[0] = OR of the kept keys, [1] = OR of their complements*/) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  bool k = false;
  uint64_t key = ~0ull;
  if (i < n) {
    const unsigned u = s_user[i];
    const SumU4 s = pre[i];
    k = over_cnt[i].v <= limit;
    double d;
    if (dru_mode == 1)
      d = s.gpus / div_gpus[u];  // dru.clj:76-77
    else {
      const double a = s.mem / div_mem[u], b = s.cpus / div_cpus[u];  // dru.clj:58-59 (max (/ mem md) (/ cpus cd))
      d = a > b ? a : b;
    }
    dru[i] = d;
    keep[i] = k ? 1 : 0;
    if (k) key = f64_key(d);
    dkey[i] = key;
  }
  const unsigned long long kept = __ballot(k);
  unsigned long long o = k ? key : 0ull, a = k ? key : ~0ull;
  for (int dd = 32; dd >= 1; dd >>= 1) {
    o |= __shfl_xor(o, dd, COOK_WAVE);
    a &= __shfl_xor(a, dd, COOK_WAVE);
  }
  // per block, not per wave (same-address atomics serialise)
  __shared__ unsigned s_n[256 / COOK_WAVE];
  __shared__ unsigned long long s_o[256 / COOK_WAVE], s_a[256 / COOK_WAVE];
  if (lane_id() == 0) s_n[wave_id()] = (unsigned)__popcll(kept), s_o[wave_id()] = o, s_a[wave_id()] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned nk = 0;
    unsigned long long bo = 0ull, ba = ~0ull;
    for (unsigned w = 0; w < blockDim.x / COOK_WAVE; ++w) nk += s_n[w], bo |= s_o[w], ba &= s_a[w];
    if (nk) {
      atomicAdd(&counters[0], nk);
      atomicOr(&or_and[0], bo);
      atomicOr(&or_and[1], ~ba);
    }
  }
}

// dropped tasks must sort behind every kept one: one extra 1-bit pass key
COOK_KERNEL void rank_notkept_key(const uint8_t* __restrict__ keep, unsigned n, uint64_t* __restrict__ k) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) k[i] = keep[i] ? 0ull : 1ull;
}

// ---- A.5 global order: tie groups and the sorted-merge tie rule (dru.clj:82-104) ---------------------------
// Sorting by DRU alone leaves groups of equal keys.  The reference's merge orders a tie group by which user emitted most
// recently, i.e. by the final position of each item's predecessor in its user's list, later predecessor first; users
// that have not emitted yet come last in name order.  Unfolding the recursion, item j of user u sorts by the sequence
//   ( d_j , -d_{j-1} , +d_{j-2} , -d_{j-3} , ... , (+/-) v_u )      v_u = -(BIG + name rank of u)  ("virtual" item)
// compared lexicographically.  We resolve it by prefix doubling over ranks (as in suffix-array construction):
//   rank_0 = tie group by d;  key_1 = (rank_0(j), MAXR - rank_0(j-1));  key_{k+1} = (rank_k(j), rank_k(j - 2^k)), k >= 1.
// Ranks are "U + first C-position of the item's tie group"; virtual items take ranks U-1-u (all below any real rank).
COOK_KERNEL void tie_heads(const uint32_t* __restrict__ permC, const uint64_t* __restrict__ dkey, unsigned n_kept,
                                                 const uint32_t* __restrict__ s_user, uint8_t* __restrict__ thead,
                                                 uint8_t* __restrict__ dhead /*a second copy that stays, or null*/,
                                                 int* __restrict__ ones /*or null*/, unsigned* __restrict__ equal_runs) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_kept) return;
  const unsigned i = permC[p];
  const uint8_t h = (p == 0 || dkey[i] != dkey[permC[p - 1]]) ? 1 : 0;
  thead[p] = h;
  if (dhead) dhead[p] = h;
  if (ones) ones[p] = 1;
  // a user's DRUs must increase strictly along its list (positive resources); equal neighbours are counted so the
  // host can refuse instead of mis-ordering them ("equal consecutive DRUs")
  if (i > 0 && s_user[i - 1] == s_user[i] && dkey[i - 1] == dkey[i]) atomicAdd(equal_runs, 1u);
}

// idx_in_group (1-based, from the segmented scan of ones) -> group start, rank of the item, tied flag, tied count
COOK_KERNEL void tie_assign(const uint32_t* __restrict__ permC, const uint8_t* __restrict__ thead,
                                                  const SumI* __restrict__ idx_in_group, unsigned n_kept, unsigned n_users,
                                                  uint32_t* __restrict__ rank_of_item /*[B]*/, uint32_t* __restrict__ gstart /*[C]*/,
                                                  int* __restrict__ tied /*[C]*/, unsigned* __restrict__ n_tied) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  bool t = false;
  if (p < n_kept) {
    const unsigned start = p - (unsigned)(idx_in_group[p].v - 1);
    gstart[p] = start;
    rank_of_item[permC[p]] = n_users + start;
    const bool next_same = (p + 1 < n_kept) && !thead[p + 1];
    t = !thead[p] || next_same;
    tied[p] = t ? 1 : 0;
  }
  const unsigned long long m = __ballot(t);
  if (lane_id() == 0 && m) atomicAdd(n_tied, (unsigned)__popcll(m));
}

// compact the tied C-positions (excl = exclusive prefix of `tied`) and build their composite keys for round k
COOK_KERNEL void tie_build(const uint32_t* __restrict__ permC, const int* __restrict__ tied,
                                                 const SumI* __restrict__ tied_incl, const uint32_t* __restrict__ gstart,
                                                 unsigned n_kept, unsigned n_users, unsigned n_items, int round, unsigned key_bits,
                                                 const uint32_t* __restrict__ rank_of_item, const uint32_t* __restrict__ s_user,
                                                 const uint32_t* __restrict__ seg_start, uint32_t* __restrict__ tpos,
                                                 uint32_t* __restrict__ titem, uint64_t* __restrict__ ckey) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_kept || !tied[p]) return;
  const unsigned j = (unsigned)tied_incl[p].v - 1;
  const unsigned i = permC[p];
  const unsigned u = s_user[i];
  const unsigned idx = i - seg_start[u];  // position in the user's list
  const unsigned maxr = n_users + n_items;
  unsigned sec;
  if (round == 0) {
    const unsigned pr = idx >= 1 ? rank_of_item[i - 1] : (n_users - 1 - u);
    sec = maxr - pr;  // later predecessor first
  } else {
    const unsigned step = 1u << round;
    if (idx >= step)
      sec = rank_of_item[i - step];
    else if (idx + 1 == step)
      sec = n_users - 1 - u;
    else
      sec = 0;  // the item's sequence already ended inside rank_k: it is unique in its group
  }
  tpos[j] = p;
  titem[j] = i;
  ckey[j] = ((uint64_t)gstart[p] << key_bits) | sec;  // both below 2^key_bits
}

// after sorting the tied items by composite key: write them back into their (contiguous) group slots and split groups
COOK_KERNEL void tie_writeback(const uint32_t* __restrict__ sorted_j, const uint32_t* __restrict__ tpos,
                                                     const uint32_t* __restrict__ titem, const uint64_t* __restrict__ ckey,
                                                     unsigned n_tied, uint32_t* __restrict__ permC, uint8_t* __restrict__ thead) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_tied) return;
  const unsigned sj = sorted_j[j];
  const unsigned p = tpos[j];  // tied slots ascend with j; the sort key's high word (group start) ascends too
  permC[p] = titem[sj];
  if (j > 0 && ckey[sorted_j[j - 1]] != ckey[sj]) thead[p] = 1;
}

// ---- equal consecutive DRUs inside one user (a zero-resource task, a gpu-less task in gpu mode, a request absorbed by the sum) ---
// The literal merge (dru.clj:92-94) re-conses an emitting coll at the FRONT: when the next head of that user carries the SAME
// key it is the first minimum of the next stable sort, so a run of equal keys inside a user is emitted back to back, in the
// user's task order.  The prefix-doubling scheme above assumes strictly increasing keys per user (an item's predecessor lives
// in an EARLIER tie group), so such runs are collapsed first: only a run's head takes part in the tie refinement (in an index
// space without the followers), and the followers are re-inserted right behind their head afterwards.
// follower flag over B: kept, same user and same key as the item before it
COOK_KERNEL void run_follower_flag(const uint32_t* __restrict__ s_user, const uint64_t* __restrict__ dkey,
                                                         const uint8_t* __restrict__ keep, unsigned n, int* __restrict__ isf,
                                                         int* __restrict__ nonf) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool f = keep[i] && i > 0 && s_user[i - 1] == s_user[i] && dkey[i - 1] == dkey[i];
  isf[i] = f ? 1 : 0;
  nonf[i] = f ? 0 : 1;
}
// compacted copies of the per-item arrays (index space B' = B without followers)
COOK_KERNEL void run_compact_items(const int* __restrict__ nonf, const SumI* __restrict__ nonf_incl, unsigned n,
                                                         const uint32_t* __restrict__ s_user, const uint64_t* __restrict__ dkey,
                                                         const uint8_t* __restrict__ head, uint32_t* __restrict__ c_user,
                                                         uint64_t* __restrict__ c_dkey, uint32_t* __restrict__ c_orig,
                                                         uint32_t* __restrict__ c_seg_start, uint32_t* __restrict__ b_to_c) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned ci = (unsigned)nonf_incl[i].v - 1u;  // for a follower: the compact index of its run head
  b_to_c[i] = ci;
  if (!nonf[i]) return;
  c_user[ci] = s_user[i];
  c_dkey[ci] = dkey[i];
  c_orig[ci] = i;
  if (head[i]) c_seg_start[s_user[i]] = ci;  // a user's first item is never a follower
}
// sentinel behind the last compact item (its follower count = n - c_orig[last] - 1)
COOK_KERNEL void run_compact_sentinel(const SumI* __restrict__ nonf_incl, unsigned n, uint32_t* __restrict__ c_orig) {
  if (blockIdx.x == 0 && threadIdx.x == 0) c_orig[(unsigned)nonf_incl[n - 1].v] = n;
}
// flag over C positions [0, n_kept): the item at this position is a follower
COOK_KERNEL void run_flag_positions(const uint32_t* __restrict__ permC, const int* __restrict__ isf, unsigned n_kept,
                                                          int* __restrict__ posf) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_kept) posf[p] = isf[permC[p]];
}
// permC' = permC without the followers, in compact indices
COOK_KERNEL void run_compact_positions(const uint32_t* __restrict__ permC, const int* __restrict__ posf,
                                                             const SumI* __restrict__ posf_incl, unsigned n_kept,
                                                             const uint32_t* __restrict__ b_to_c, uint32_t* __restrict__ permC2) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_kept || posf[p]) return;
  permC2[p - (unsigned)posf_incl[p].v] = b_to_c[permC[p]];
}
// followers per position of the refined compact order (input of the expansion scan)
COOK_KERNEL void run_count_followers(const uint32_t* __restrict__ permC2, const uint32_t* __restrict__ c_orig,
                                                           unsigned n2, int* __restrict__ nfol) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n2) return;
  const unsigned ci = permC2[p];
  // everything between two non-followers of B is a follower of the first (followers are kept by definition)
  nfol[p] = (int)(c_orig[ci + 1] - c_orig[ci] - 1u);
}
// final order: every run head followed by its followers
COOK_KERNEL void run_expand(const uint32_t* __restrict__ permC2, const uint32_t* __restrict__ c_orig,
                                                  const int* __restrict__ nfol, const SumI* __restrict__ nfol_incl, unsigned n2,
                                                  uint32_t* __restrict__ permC) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n2) return;
  const unsigned i = c_orig[permC2[p]];
  const unsigned f = (unsigned)nfol[p];
  const unsigned o = p + (unsigned)nfol_incl[p].v - f;  // p + exclusive prefix of the follower counts
  for (unsigned t = 0; t <= f; ++t) permC[o + t] = i + t;
}

// ---- A.6 queue of pending jobs in rank order and the quota filters ------------------------------------------
COOK_KERNEL void queue_flag_pending(const uint32_t* __restrict__ permC, const uint8_t* __restrict__ s_pending,
                                                          unsigned n_kept, int* __restrict__ flag) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_kept) flag[p] = s_pending[permC[p]] ? 1 : 0;
}

// L[q] = B-index of the q-th pending job in rank order; its usage for the quota scans
COOK_KERNEL void queue_compact_pending(const uint32_t* __restrict__ permC, const int* __restrict__ flag,
                                                             const SumI* __restrict__ incl, unsigned n_kept,
                                                             const SumU4* __restrict__ s_use, uint32_t* __restrict__ qitem,
                                                             SumU4* __restrict__ quse, unsigned* __restrict__ qlen) {
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_kept) return;
  if (flag[p]) {
    const unsigned q = (unsigned)incl[p].v - 1;
    const unsigned i = permC[p];
    qitem[q] = i;
    quse[q] = s_use[i];
  }
  if (p == n_kept - 1) *qlen = (unsigned)incl[p].v;
}

struct LoadQueueUse {  // element 0 carries the starting usage: ((base + j0) + j1) + ... as filter-sequential does
  const SumU4* p;
  SumU4 base;
  __device__ __forceinline__ SumU4 operator()(unsigned i) const {
    const SumU4 x = p[i];
    return i == 0 ? combine(base, x) : x;
  }
};

// tools.clj:917-933: keep iff the updated usage is below-quota?.  Also notes whether any prefix was inexact.
COOK_KERNEL void queue_quota_flag(const SumU4* __restrict__ pre, unsigned len, Usage4 quota, int* __restrict__ flag,
                                                        unsigned* __restrict__ any_bad) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= len) return;
  const SumU4 s = pre[q];
  flag[q] = below_quota4(quota.count, quota.cpus, quota.mem, quota.gpus, Usage4{s.count, s.cpus, s.mem, s.gpus}) ? 1 : 0;
  if (s.bad) *any_bad = 1u;
}

// exact sequential recomputation of a queue prefix (one thread; only runs when a parallel partial sum was inexact)
COOK_KERNEL void queue_quota_fix(const SumU4* __restrict__ quse, unsigned len, SumU4 base, Usage4 quota, const unsigned* __restrict__ any_bad,
                                int* __restrict__ flag) {
  if (blockIdx.x != 0 || threadIdx.x != 0 || !*any_bad) return;
  double c = base.count, cp = base.cpus, m = base.mem, g = base.gpus;
  for (unsigned q = 0; q < len; ++q) {
    const SumU4 x = quse[q];
    c = x.count + c;  // (merge-with + job-usage usage), tools.clj:927
    cp = x.cpus + cp;
    m = x.mem + m;
    g = x.gpus + g;
    flag[q] = below_quota4(quota.count, quota.cpus, quota.mem, quota.gpus, Usage4{c, cp, m, g}) ? 1 : 0;
  }
}

// offensive filter folded into the last stage (scheduler.clj:2198-2229): applied AFTER the quota filters saw the job
COOK_KERNEL void queue_offensive_flag(const SumU4* __restrict__ quse, unsigned len, double max_mem, double max_cpus,
                                                            int* __restrict__ flag) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= len) return;
  const SumU4 x = quse[q];
  flag[q] = (x.mem > max_mem || x.cpus > max_cpus) ? 0 : 1;
}

COOK_KERNEL void queue_compact(const uint32_t* __restrict__ qitem_in, const SumU4* __restrict__ quse_in,
                                                     const int* __restrict__ flag, const SumI* __restrict__ incl, unsigned len,
                                                     uint32_t* __restrict__ qitem_out, SumU4* __restrict__ quse_out,
                                                     unsigned* __restrict__ len_out) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= len) return;
  if (flag[q]) {
    const unsigned o = (unsigned)incl[q].v - 1;
    qitem_out[o] = qitem_in[q];
    quse_out[o] = quse_in[q];
  }
  if (q == len - 1) *len_out = (unsigned)incl[q].v;
}

// final: B-index -> caller's task index; DRU back into A space
COOK_KERNEL void queue_emit(const uint32_t* __restrict__ qitem, unsigned len, const uint32_t* __restrict__ permB,
                                                  uint32_t* __restrict__ ranked) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < len) ranked[q] = permB[qitem[q]];
}
COOK_KERNEL void dru_to_task_space(const double* __restrict__ dru, const uint8_t* __restrict__ keep,
                                                         const uint32_t* __restrict__ permB, unsigned n, double* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[permB[i]] = keep[i] ? dru[i] : __longlong_as_double(0x7FF8000000000000ll);
}

// ---- per-user running usage of the pool: the [U x 3] vector of the cross-pool all-reduce (BASELINE.json north_star; ----------
// SURVEY.md §8e).  Sum of {cpus, mem, gpus} over the user's RUNNING tasks in the user's task order (tools.clj:614-641; the
// reference's own per-user usage maps reduce in query order, which is not defined): a masked segmented scan over the
// per-user order the rank already holds, exactness tracked and fixed up like the DRU prefixes.
struct LoadRunningU4 {
  const SumU4* use;
  const uint8_t* pending;
  __device__ __forceinline__ SumU4 operator()(unsigned i) const {
    if (pending[i]) return SumU4::zero();
    return use[i];
  }
};
COOK_KERNEL void user_usage_extract(const SumU4* __restrict__ run_pre, const SumU4* __restrict__ s_use,
                                                          const uint8_t* __restrict__ s_pending, const uint32_t* __restrict__ seg_start,
                                                          const uint32_t* __restrict__ seg_end, unsigned n_users,
                                                          double* __restrict__ out /*[U][3]*/) {
  const unsigned u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_users) return;
  double c = 0.0, m = 0.0, g = 0.0;
  if (seg_end[u] != 0u) {  // (rank_init: 0 = the user has no task in this pool)
    const unsigned a = seg_start[u], b = seg_end[u];
    const SumU4 t = run_pre[b - 1];
    if (!t.bad) {
      c = t.cpus, m = t.mem, g = t.gpus;
    } else {  // an addition of the parallel scan rounded: left to right, as a sequential reduce would
      bool first = true;
      for (unsigned i = a; i < b; ++i) {
        if (s_pending[i]) continue;
        const SumU4 x = s_use[i];
        if (first) {
          c = x.cpus, m = x.mem, g = x.gpus;
          first = false;
        } else {
          c += x.cpus, m += x.mem, g += x.gpus;
        }
      }
    }
  }
  out[(size_t)u * 3 + 0] = c;
  out[(size_t)u * 3 + 1] = m;
  out[(size_t)u * 3 + 2] = g;
}
// ---- pool running usage (scheduler.clj:2118-2123, 2173): one workgroup, exactness tracked ----------------------
// stage 1: POOL_USAGE_BLOCKS blocks fold strided slices (a single 1024-thread block took 153 us for 175k tasks: 171 dependent
// iterations); stage 2 (pool_usage_reduce) combines the partial sums, or redoes the sum left to right when one of them rounded
constexpr int POOL_USAGE_BLOCKS = 64;
COOK_KERNEL void pool_usage_partial(const double* __restrict__ cpus, const double* __restrict__ mem,
                                                          const double* __restrict__ gpus, const uint8_t* __restrict__ pending,
                                                          unsigned n, SumU4* __restrict__ part, unsigned nblk) {
  __shared__ SumU4 ws[256 / COOK_WAVE];
  SumU4 acc = SumU4::zero();
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nblk * blockDim.x)
    if (!pending[i]) acc = combine(acc, SumU4{1.0, cpus[i], mem[i], gpus ? gpus[i] : 0.0, 0u});
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
    part[blockIdx.x] = t;
  }
}
COOK_KERNEL void pool_usage_reduce(const double* __restrict__ cpus, const double* __restrict__ mem,
                                                               const double* __restrict__ gpus, const uint8_t* __restrict__ pending,
                                                               unsigned n, const SumU4* __restrict__ part, unsigned n_part,
                                                               SumU4* __restrict__ out) {
  if (threadIdx.x != 0) return;
  SumU4 t = part[0];
  for (unsigned k = 1; k < n_part; ++k) t = combine(t, part[k]);
  if (t.bad) {  // some partial sum rounded: redo left to right like the reference
    t = SumU4::zero();
    for (unsigned i = 0; i < n; ++i)
      if (!pending[i]) {
        t.count += 1.0;
        t.cpus += cpus[i];
        t.mem += mem[i];
        t.gpus += gpus ? gpus[i] : 0.0;
      }
  }
  *out = t;
}

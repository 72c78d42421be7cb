// match_window.hpp — exact rank-ordered placement without a K-long chain of full offer sweeps.
//
// Placement is sequential by definition (job i+1 sees job i's commitment), but a commitment changes ONE offer.  So for
// a window of W consecutive jobs we split the work in two launches per round:
//
//   match_window_eval  (W workgroups, one job each, all M offers, fully parallel, the heavy part):
//       against the snapshot S of per-offer assignments at the start of the round, find the job's L best feasible
//       offers (fitness desc, index asc) — a PREFIX of its true ranking — plus, when good-enough < 1, the first L
//       offers in array order whose fitness exceeds good-enough, plus per-reason failure counts.
//   match_window_resolve  (ONE wave, the sequential part, touches only "touched" offers):
//       walks the window in rank order keeping the offers committed to in this round ("touched", <= 64, one per
//       lane, state in registers).  For job j the true winner under the current state S' is
//           max( best UNTOUCHED offer under S  ,  best TOUCHED offer re-evaluated under S' )
//       and the first entry of j's list that is untouched — or touched and still feasible (its fitness only grew, so it
//       beats every untouched offer) — settles the left term.  If the list (length L, more candidates may exist) runs
//       out, or a 65th offer would be touched, the round ends at j and the next round re-snapshots from there.
//
// The result is bit-identical to the one-job-at-a-time sweep (match_serial) for every input; only speed depends on L/W.
// Jobs of balanced / attribute-equals groups change the feasibility of UNTOUCHED offers when a cotask is placed, so a
// round never resolves a second member of such a group after the first one was placed.
#pragma once
#include "common.hpp"
#include "match_kernels.hpp"

constexpr int MW_L = 4;        // candidate list length per job
constexpr int MW_THREADS = 256;
constexpr int MW_T = COOK_WAVE;  // touched slots = lanes of the resolving wave

struct WinCtl {
  unsigned head;          // first unresolved job
  unsigned wcur;          // window size for the next round
  unsigned rounds;
  unsigned matched;
  unsigned head_matched;  // job 0 was matched
  unsigned stop_list, stop_full, stop_group, stop_window;  // why rounds ended (statistics)
};

struct WinBuf {           // per-window outputs of match_window_eval, indexed by (job - head)
  double* cand_fit;       // [W][L]
  int* cand_idx;          // [W][L]
  int* ncand;             // [W]  entries valid in the list (== L: more feasible offers may exist)
  int* ge_idx;            // [W][L] first offers (array order) with fitness > good-enough under S
  int* nge;               // [W]
  unsigned* failcnt;      // [W][3] number of offers failing on resources / constraints / zero fitness under S
  WinCtl* ctl;
  unsigned wmax;
};

struct PairEval {
  double fit;      // valid when bits == 0
  unsigned bits;   // 0 feasible; 1 resources, 2 constraints, 4 zero fitness (first failing check, as Fenzo reports)
};

// One (job, offer) evaluation under explicit assignment state (ac, am, acount of the offer).  st.cutoff: placements by jobs
// with match index >= cutoff are ignored by the group-unique check (re-creates the snapshot view of a touched offer).
static __device__ __forceinline__ PairEval eval_pair(const MatchIn& in, const MatchState& st, unsigned jj, double c, double m,
                                                     unsigned v, double ac, double am, int acount) {
  PairEval r{0.0, 0u};
  if (ac + c > in.o_cpus[v] || am + m > in.o_mem[v]) {
    r.bits = 1u;
    return r;
  }
  if (!constraints_pass(in, st, jj, v, acount)) {
    r.bits = 2u;
    return r;
  }
  const double rc = in.o_run_cpus ? in.o_run_cpus[v] : 0.0, rm = in.o_run_mem ? in.o_run_mem[v] : 0.0;
  r.fit = ((rc + ac + c) / (in.o_cpus[v] + rc) + (rm + am + m) / (in.o_mem[v] + rm)) / 2.0;
  if (!(r.fit > 0.0)) r.bits = 4u;
  return r;
}

// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(MW_THREADS) match_window_eval(MatchIn in, MatchState st, WinBuf wb) {
  constexpr int NW = MW_THREADS / COOK_WAVE;
  __shared__ double s_fit[NW];
  __shared__ int s_idx[NW];
  __shared__ unsigned s_cnt[3];
  const unsigned head = wb.ctl->head, wcur = wb.ctl->wcur;
  const unsigned b = blockIdx.x;
  if (b >= wcur || head + b >= in.K) return;
  const unsigned k = head + b;
  const unsigned jj = in.j_index ? in.j_index[k] : k;
  const double c = in.j_cpus[jj], m = in.j_mem[jj];
  const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
  if (tid < 3) s_cnt[tid] = 0;
  // per-thread top-L (fitness desc, index asc; offers are visited in ascending index so ties keep the earlier one)
  double tf[MW_L];
  int ti[MW_L];
  int gi[MW_L];
#pragma unroll
  for (int q = 0; q < MW_L; ++q) {
    tf[q] = -1.0;
    ti[q] = -1;
    gi[q] = 0x7FFFFFFF;
  }
  unsigned c1 = 0, c2 = 0, c4 = 0;
  const bool use_ge = in.good_enough < 1.0;
  for (unsigned v = tid; v < in.M; v += MW_THREADS) {
    const PairEval pe = eval_pair(in, st, jj, c, m, v, st.ac[v], st.am[v], st.acount[v]);
    if (pe.bits) {
      c1 += pe.bits & 1u;
      c2 += (pe.bits >> 1) & 1u;
      c4 += (pe.bits >> 2) & 1u;
      continue;
    }
    if (pe.fit > tf[MW_L - 1]) {  // insert, keeping order; strict > keeps the earlier index on equal fitness
      tf[MW_L - 1] = pe.fit;
      ti[MW_L - 1] = (int)v;
#pragma unroll
      for (int q = MW_L - 1; q > 0; --q) {
        if (tf[q] > tf[q - 1]) {
          const double a = tf[q];
          tf[q] = tf[q - 1];
          tf[q - 1] = a;
          const int x = ti[q];
          ti[q] = ti[q - 1];
          ti[q - 1] = x;
        }
      }
    }
    if (use_ge && pe.fit > in.good_enough) {
#pragma unroll
      for (int q = 0; q < MW_L; ++q) {
        if (gi[q] == 0x7FFFFFFF) {
          gi[q] = (int)v;
          break;
        }
      }
    }
  }
  __syncthreads();
  if (c1) atomicAdd(&s_cnt[0], c1);
  if (c2) atomicAdd(&s_cnt[1], c2);
  if (c4) atomicAdd(&s_cnt[2], c4);
  // merge: L rounds of workgroup arg-max over the threads' list heads
  int n_out = 0;
  for (int round = 0; round < MW_L; ++round) {
    Cand best{tf[0], ti[0]};
    for (int d = 32; d >= 1; d >>= 1) {
      Cand o{__shfl_xor(best.fit, d, COOK_WAVE), __shfl_xor(best.idx, d, COOK_WAVE)};
      if (cand_better(o, best)) best = o;
    }
    if (lane == 0) {
      s_fit[w] = best.fit;
      s_idx[w] = best.idx;
    }
    __syncthreads();
    Cand win{s_fit[0], s_idx[0]};
    for (int q = 1; q < NW; ++q) {
      Cand o{s_fit[q], s_idx[q]};
      if (cand_better(o, win)) win = o;
    }
    __syncthreads();
    if (win.idx < 0) break;
    if (tid == 0) {
      wb.cand_fit[(size_t)b * MW_L + round] = win.fit;
      wb.cand_idx[(size_t)b * MW_L + round] = win.idx;
    }
    ++n_out;
    if (ti[0] == win.idx) {  // the owner pops its head
#pragma unroll
      for (int q = 0; q < MW_L - 1; ++q) {
        tf[q] = tf[q + 1];
        ti[q] = ti[q + 1];
      }
      tf[MW_L - 1] = -1.0;
      ti[MW_L - 1] = -1;
    }
  }
  int n_ge = 0;
  if (use_ge) {
    for (int round = 0; round < MW_L; ++round) {
      int best = gi[0];
      for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(best, d, COOK_WAVE);
        best = o < best ? o : best;
      }
      if (lane == 0) s_idx[w] = best;
      __syncthreads();
      int win = s_idx[0];
      for (int q = 1; q < NW; ++q) win = s_idx[q] < win ? s_idx[q] : win;
      __syncthreads();
      if (win == 0x7FFFFFFF) break;
      if (tid == 0) wb.ge_idx[(size_t)b * MW_L + round] = win;
      ++n_ge;
      if (gi[0] == win) {
#pragma unroll
        for (int q = 0; q < MW_L - 1; ++q) gi[q] = gi[q + 1];
        gi[MW_L - 1] = 0x7FFFFFFF;
      }
    }
  }
  if (tid == 0) {
    wb.ncand[b] = n_out;
    wb.nge[b] = n_ge;
    wb.failcnt[(size_t)b * 3 + 0] = s_cnt[0];
    wb.failcnt[(size_t)b * 3 + 1] = s_cnt[1];
    wb.failcnt[(size_t)b * 3 + 2] = s_cnt[2];
  }
}

// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(COOK_WAVE) match_window_resolve(MatchIn in, MatchState st, WinBuf wb) {
  const unsigned lane = lane_id();
  WinCtl ctl = *wb.ctl;
  const unsigned head = ctl.head;
  if (head >= in.K) return;
  const unsigned wend = (head + ctl.wcur < in.K) ? head + ctl.wcur : in.K;
  const bool use_ge = in.good_enough < 1.0;
  // touched slot of this lane
  int t_v = -1;
  double t_ac = 0, t_am = 0, t_ac0 = 0, t_am0 = 0;
  int t_acount = 0, t_acount0 = 0;
  unsigned nT = 0;
  unsigned k = head;
  unsigned stop = 0;  // 1 list exhausted, 2 touched set full, 3 group barrier
  unsigned matched = 0, head_matched = ctl.head_matched;
  for (; k < wend; ++k) {
    const unsigned b = k - head;
    const unsigned jj = in.j_index ? in.j_index[k] : k;
    const double c = in.j_cpus[jj], m = in.j_mem[jj];
    // a second member of a balanced / attribute-equals group after one was placed in this round: re-snapshot first
    unsigned g = 0xFFFFFFFFu, gtype = 0;
    if (in.j_group && in.j_group[jj] != 0xFFFFFFFFu) {
      g = in.j_group[jj];
      gtype = in.g_type[g];
      if (gtype >= 2) {
        const int last = ld_agent(&st.group_last[g]);
        if (last >= (int)head) {
          stop = 3;
          break;
        }
      }
    }
    // every touched offer re-evaluated under the current state
    PairEval pe{0.0, 8u};
    if (t_v >= 0) pe = eval_pair(in, st, jj, c, m, (unsigned)t_v, t_ac, t_am, t_acount);
    const bool t_feas = (t_v >= 0) && pe.bits == 0;
    // --- arg-max path: first list entry that is untouched, or touched and still feasible ------------------------
    const int nc = wb.ncand[b];
    Cand ucand{-1.0, -1};
    bool settled = false;
    for (int q = 0; q < nc; ++q) {
      const int idx = wb.cand_idx[(size_t)b * MW_L + q];
      const unsigned long long owner = __ballot(t_v == idx);
      if (owner == 0ull) {
        ucand = Cand{wb.cand_fit[(size_t)b * MW_L + q], idx};
        settled = true;
        break;
      }
      const int ol = __ffsll((unsigned long long)owner) - 1;
      if (__shfl((int)t_feas, ol, COOK_WAVE)) {
        settled = true;  // a touched, still feasible offer dominates every untouched one
        break;
      }
    }
    if (!settled && nc == MW_L) {
      stop = 1;
      break;
    }
    // --- good-enough path: lowest index with fitness > good-enough ------------------------------------------------
    int ge_pick = 0x7FFFFFFF;
    if (use_ge) {
      const int ng = wb.nge[b];
      bool ge_settled = false;
      for (int q = 0; q < ng; ++q) {
        const int idx = wb.ge_idx[(size_t)b * MW_L + q];
        if (__ballot(t_v == idx) == 0ull) {
          ge_pick = idx;
          ge_settled = true;
          break;
        }
      }
      int tg = (t_feas && pe.fit > in.good_enough) ? t_v : 0x7FFFFFFF;
      for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(tg, d, COOK_WAVE);
        tg = o < tg ? o : tg;
      }
      if (!ge_settled && ng == MW_L && tg > wb.ge_idx[(size_t)b * MW_L + MW_L - 1]) {
        // untouched good-enough offers beyond the list may exist with an index below the best touched one
        stop = 1;
        break;
      }
      ge_pick = tg < ge_pick ? tg : ge_pick;
    }
    // --- best touched ------------------------------------------------------------------------------------------------
    Cand best{t_feas ? pe.fit : -1.0, t_feas ? t_v : -1};
    for (int d = 32; d >= 1; d >>= 1) {
      Cand o{__shfl_xor(best.fit, d, COOK_WAVE), __shfl_xor(best.idx, d, COOK_WAVE)};
      if (cand_better(o, best)) best = o;
    }
    if (cand_better(ucand, best)) best = ucand;
    const int win = (ge_pick != 0x7FFFFFFF) ? ge_pick : best.idx;
    // --- commit ------------------------------------------------------------------------------------------------------
    const unsigned long long owner = __ballot(t_v == win && win >= 0);
    if (win >= 0 && owner == 0ull && nT == MW_T) {
      stop = 2;  // no free slot to track a new touched offer: end the round before this job
      break;
    }
    if (win >= 0) {
      if (owner == 0ull) {
        if (lane == nT) {
          t_v = win;
          t_ac0 = st.ac[win];
          t_am0 = st.am[win];
          t_acount0 = st.acount[win];
          t_ac = t_ac0 + c;
          t_am = t_am0 + m;
          t_acount = t_acount0 + 1;
        }
        ++nT;
      } else if (t_v == win) {
        t_ac += c;
        t_am += m;
        t_acount += 1;
      }
      ++matched;
      if (k == 0) head_matched = 1;
      if (lane == 0) {
        st_agent(&st.job_to_offer[k], win);
        if (st.fail_code) st.fail_code[k] = 0u;
        if (g != 0xFFFFFFFFu) {
          st_agent(&st.job_prev[k], ld_agent(&st.group_last[g]));
          st_agent(&st.group_last[g], (int)k);
        }
      }
      // the unique-group check of later jobs reads job_to_offer / group lists written by lane 0: order them
      wave_sync();
    } else {
      // unmatched: failure summary = OR over offers of the first failing check under the CURRENT state.  Start from the
      // snapshot counts and swap each touched offer's snapshot verdict for its current one.
      unsigned f1 = wb.failcnt[(size_t)b * 3 + 0], f2 = wb.failcnt[(size_t)b * 3 + 1], f4 = wb.failcnt[(size_t)b * 3 + 2];
      int d1 = 0, d2 = 0, d4 = 0;
      if (t_v >= 0) {
        // snapshot view of this offer: state at round start, group placements of this round ignored via job index cutoff
        MatchState st0 = st;
        st0.cutoff = (int)head;
        const PairEval p0 = eval_pair(in, st0, jj, c, m, (unsigned)t_v, t_ac0, t_am0, t_acount0);
        d1 = (int)(pe.bits & 1u) - (int)(p0.bits & 1u);
        d2 = (int)((pe.bits >> 1) & 1u) - (int)((p0.bits >> 1) & 1u);
        d4 = (int)((pe.bits >> 2) & 1u) - (int)((p0.bits >> 2) & 1u);
      }
      for (int d = 32; d >= 1; d >>= 1) {
        d1 += __shfl_xor(d1, d, COOK_WAVE);
        d2 += __shfl_xor(d2, d, COOK_WAVE);
        d4 += __shfl_xor(d4, d, COOK_WAVE);
      }
      const unsigned bits = (((int)f1 + d1) > 0 ? 1u : 0u) | (((int)f2 + d2) > 0 ? 2u : 0u) | (((int)f4 + d4) > 0 ? 4u : 0u);
      if (lane == 0) {
        st_agent(&st.job_to_offer[k], -1);
        if (st.fail_code) st.fail_code[k] = bits ? bits : 8u;
      }
    }
  }
  // write the touched offers' state back and publish the new head
  if (t_v >= 0) {
    st.ac[t_v] = t_ac;
    st.am[t_v] = t_am;
    st.acount[t_v] = t_acount;
  }
  if (lane == 0) {
    const unsigned resolved = k - head;
    ctl.head = k;
    ctl.rounds += 1;
    ctl.matched += matched;
    ctl.head_matched = head_matched;
    if (stop == 1) ctl.stop_list += 1;
    if (stop == 2) ctl.stop_full += 1;
    if (stop == 3) ctl.stop_group += 1;
    if (stop == 0) ctl.stop_window += 1;
    // adapt the window: aim at ~2x what a round resolves, within [32, wmax]
    unsigned wn = stop == 0 ? ctl.wcur * 2 : (resolved * 2 > 32 ? resolved * 2 : 32);
    if (wn < 32) wn = 32;
    if (wn > wb.wmax) wn = wb.wmax;  // never beyond the eval grid
    ctl.wcur = wn;
    *wb.ctl = ctl;
  }
}

// match_v2.hpp — exact rank-ordered placement (Fenzo scheduleOnce semantics, scheduler.clj:617-687) as a pipeline of
// window rounds.  Placement is sequential by definition — job i+1 sees job i's commitment — but one commitment changes
// ONE offer.  For a window of W consecutive jobs a round is three launches:
//
//   match_eval2    (grid = offer chunks x job groups; lane = job, offers walked in a wave-uniform loop so the offer record
//                   comes through scalar loads): against the snapshot S of per-offer assignments at round start, every job
//                   gets the top-L feasible offers of each chunk (fitness desc, index asc), the first LG offers whose
//                   fitness exceeds good-enough, failure counts, and — for every (job, offer) — one bit "static
//                   constraints pass" (a ballot over the 64 jobs of the wave = one u64 per offer: colbits[offer][group]).
//                   The two fp64 divides of the fitness are only executed for pairs whose cheap upper bound
//                   (multiply by a precomputed reciprocal) can still enter the lane's top-L.
//   match_merge2   (one wave per job): chunk lists -> the job's global top-L / first-LG / failure counts.
//   match_resolve2 (ONE workgroup; after a parallel set-up phase wave 0 walks the window in rank order): all the data
//                   the sequential walk needs is first staged in LDS — job records, candidate lists, and for every
//                   DISTINCT candidate offer of the window ("slot") its record, snapshot state and colbits column — so
//                   the per-job critical path is LDS + registers only.  Lanes own the offers committed to in this round
//                   ("touched"); for job j the winner under the current state S' is
//                       max( best UNTOUCHED offer under S , best TOUCHED offer re-evaluated under S' )
//                   and the first entry of j's list that is untouched — or touched and still feasible (its fitness only
//                   grew, so it dominates every untouched offer) — settles the left term.  If the list (length L, more
//                   candidates may exist) runs out, a 65th offer would be touched, or the slot table overflowed, the round
//                   ends there and the next round re-snapshots.
//
// The result is bit-identical to the one-job-at-a-time sweep (match_serial) for every input; only speed depends on L/W.
// Jobs of balanced / attribute-equals groups change the feasibility of UNTOUCHED offers when a cotask is placed, so a
// round never resolves a second member of such a group after the first one was placed.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "match_kernels.hpp"

// waves per SIMD the eval kernels are compiled for (-DCOOK_EVAL_WAVES=n builds a tuning variant: fewer registers, more waves)
#ifdef COOK_EVAL_WAVES
#define COOK_EVAL_OCCUPANCY __attribute__((amdgpu_waves_per_eu(COOK_EVAL_WAVES, COOK_EVAL_WAVES)))
#else
#define COOK_EVAL_OCCUPANCY
#endif

#ifndef COOK_MV_L
#define COOK_MV_L 8
#endif
constexpr int MV_L = COOK_MV_L;            // candidate list length per job and chunk (-DCOOK_MV_L=n builds a variant for tuning runs)
// Length of a job's MERGED list (what the walk sees).  A job's per-chunk top-L lists determine its global top-LM exactly as long as no
// chunk has contributed all L of its entries (that chunk may hide an (L+1)-th): the merge stops there and marks the list truncated.
// 12 entries with a window of 384 jobs is what fits the walk's 160 KB of LDS: a C4 pool takes 365 rounds instead of the 476 of
// (8 entries, 512 jobs), the eight-pool cycle 82.9 ms instead of 92.6.  -DCOOK_MV_LM=8 builds the old layout's lists.
#ifndef COOK_MV_LM
#define COOK_MV_LM 12
#endif
constexpr int MV_LM = COOK_MV_LM;
constexpr bool MV_LM_EXT = MV_LM > MV_L;
static_assert(MV_LM >= MV_L && MV_LM <= 64, "the walk holds one merged-list entry per lane");
#ifndef COOK_MV_LG
#define COOK_MV_LG 4
#endif
constexpr int MV_LG = COOK_MV_LG;          // good-enough list length per job
#ifndef COOK_MV_OCW
#define COOK_MV_OCW 32
#endif
constexpr int MV_OCW = COOK_MV_OCW;        // offers per eval wave (a power of two <= 64; -DCOOK_MV_OCW=n builds a tuning variant)
#ifndef COOK_MV_EW
#define COOK_MV_EW 4
#endif
constexpr int MV_EW = COOK_MV_EW;          // waves per eval block (same 64 jobs, consecutive offer sub-chunks)
constexpr int MV_OCB = MV_OCW * MV_EW;     // offers per eval block
constexpr int MV_T = COOK_WAVE;            // touched offers per round = lanes of the walking wave
#ifndef COOK_MV_RTHREADS
#define COOK_MV_RTHREADS COOK_SHAPE(768, 256)  // (the emulated tests: fewer fibers per block; the strides are blockDim.x either way)
#endif
constexpr int MV_RTHREADS = COOK_MV_RTHREADS;  // threads of the resolve workgroup: the set-up phase is parallel over them (256 -> 768: 9.2 -> 5.1 ms
                                               // per C4 pool), wave 0 walks.  resolve_round strides by blockDim.x, so the persistent
                                               // kernel may run it with its own (eval-tile) block shape.
constexpr int MV_RWAVES_MAX = (MV_RTHREADS > COOK_WAVE * MV_EW ? MV_RTHREADS : COOK_WAVE * MV_EW) / COOK_WAVE;
#if defined(COOK_MV_WMAX)  // a study build (rounds per match against list length / window / slot table; scripts/study_rounds.py)
constexpr int MV_WMAX = COOK_MV_WMAX;
constexpr int MV_S = COOK_MV_S;
constexpr int MV_HASH = 4 * COOK_MV_S;
#else
constexpr int MV_WMAX = COOK_SHAPE(384, 128);  // jobs per round (the emulated tests: small, so that small inputs run many rounds)
constexpr int MV_S = COOK_SHAPE(256, 128);     // distinct candidate offers staged per round
constexpr int MV_HASH = COOK_SHAPE(1024, 512);
#endif
constexpr int MV_JG = MV_WMAX / 64;        // job groups (waves of jobs) per window whose walk data fits the resolve workgroup's LDS
// A window may grow to MV_WLONG jobs once next to nothing of it has to be WALKED: when the cluster is full almost every job is
// settled in the parallel phase of the resolve kernel (no feasible offer under the snapshot, however the jobs before it fare) and
// needs no LDS at all — only the jobs the walk visits are staged (at most MV_WMAX of them, by walk position).  One C4 pool spent
// 152 of its 604 rounds resolving 512 such jobs each; with long windows that tail takes about 20 rounds.
constexpr int MV_WLONG = MV_WMAX * 8;
constexpr int MV_JGL = MV_WLONG / 64;      // job groups of a long window (stride of colbits)
constexpr int MV_EPJ_MAX = (MV_LM + MV_LG) > 16 ? (MV_LM + MV_LG) : 16;
constexpr int MV_JSTEP = MV_S / MV_EPJ_MAX;  // jobs inserted into the slot table per step (at most MV_EPJ_MAX entries each)
static_assert(MV_JSTEP >= 1, "slot-table step sizing");
static_assert(MV_OCW <= COOK_WAVE, "one lane stages one offer");
static_assert(MV_OCW == 64 || MV_OCW == 32 || MV_OCW == 16 || MV_OCW == 8, "a wave's alive bits are an aligned slice of one 64-bit word");

struct OfferA {  // resources of an offer (offer.clj:55-61) + Fenzo's running view; 48 B, read wave-uniformly
  double oc, om;          // lease cpus / mem
  double rc, rm;          // resources of tasks Fenzo tracks as running on the host
  double inv_dc, inv_dm;  // 1 / (oc + rc), 1 / (om + rm): only for the pruning bound, never for the fitness itself
};
struct OfferB {  // what the cheap constraint checks need; 32 B
  uint32_t host, gpu_model;
  double gpu_count;
  int32_t run_count, task_slack;  // task_slack = COOK_MAX_TASKS_PER_HOST - COOK_NUM_TASKS_ON_HOST (INT_MAX when absent)
  uint32_t flags, pad;            // bit0 kubernetes VM, bit1 host is in the rebalancer's reserved set, bit2 the host's "gpus" map has
                                  // several entries (gpu_model = one of them; the constraint reads the table)
};
struct JobRec {  // one considerable job in match order; 40 B
  double c, m, g;
  uint32_t gpu_model;
  int32_t reserved_host;
  uint32_t group;  // COOK_NONE_U32 or group id
  uint32_t flags;  // bit0 has constraints that need the slow static check, bit1 member of a constrained group,
                   // bits 8..9 group type
};
constexpr uint32_t JF_SLOW = 1u, JF_GROUPED = 2u, JF_FASTC = 4u, JF_XRES = 8u;  // JF_XRES: asks for ports / named scalars
// The common job constraints in a form the eval loop checks from registers + LDS only: up to MV_NC user-defined EQUALS
// pairs on attribute keys < MV_NA (or HOSTNAME) and up to MV_NC novel-host entries.  Jobs with more, or with a disk /
// estimated-completion / checkpoint constraint, carry JF_SLOW and go through static_pass (global-memory CSR walk).
constexpr int MV_NC = 4;   // fast constraint slots per kind
constexpr int MV_NA = 8;   // attribute keys staged in LDS per offer
constexpr int MV_FH = 8;   // hosts a unique-group job must avoid, kept in registers per tile
struct JobCons {
  uint32_t eq_key[MV_NC], eq_val[MV_NC], novel[MV_NC];
  uint32_t n_eq, n_novel;
};

struct WinCtl {
  unsigned head;          // first unresolved job
  unsigned wcur;          // window size for the next round
  unsigned rounds;
  unsigned matched;
  unsigned head_matched;  // job 0 was matched
  unsigned stop_list, stop_full, stop_group, stop_window, stop_slots;  // why rounds ended (statistics)
  unsigned touched_sum;   // sum over rounds of touched offers
  unsigned visited_sum;   // sum over rounds of jobs the walk had to visit (the rest were settled in parallel)
  unsigned long long t_setup, t_seq;  // resolve kernel: ticks (100 MHz wall clock) spent in the set-up / sequential phase
  unsigned reeval_max;    // list-exhausted jobs re-evaluated in place per round before the round ends (0 = end the round at once)
  unsigned reevals;       // jobs re-evaluated in place (statistics)
  unsigned trunc_lists;   // walked jobs whose merged list carried the truncated flag (MV_LM > MV_L: the merge stopped on a full chunk list)
  unsigned trunc_stops;   // rounds that ended on such a list running out
  unsigned wgrow_pct;     // next window = this percentage of what the round resolved (window ended early) / of the window (it did not)
  unsigned wlong_cap;     // largest window the launch sequence allows (MV_WLONG, or MV_WMAX when long windows are switched off)
  unsigned long long t_eval, t_merge;  // persistent kernel: ticks spent in the eval / merge phases (as seen by workgroup 0)
#ifdef COOK_WALK_PROF  // measurement build: shader cycles / jobs of the walk by outcome (0 shortcut, 1 touched offer wins, 2 new lane,
                       // 3 walked and unmatched, 4 member of a constrained group, 5 exact path ran)
  unsigned long long prof_cyc[8];
  unsigned prof_cnt[8];
#endif
};

struct RoundLog {  // one record per round (diagnostics; only written when V2Buf::round_log is set)
  unsigned head, wcur, resolved, n_list, touched, stop, matched, setup_ticks, seq_ticks, nslots, pad0, pad1;
};
constexpr unsigned MV_ROUND_LOG_CAP = 8192;

// One offer chunk's candidates for one job, as ONE aligned record (128 bytes at L = 8, LG = 4) that the evaluating lane writes
// and the merging lane reads in 16-byte pieces: whole lines, so the persistent kernel can publish it with write-through stores
// (MI355X_MICROARCH.md "publish-large": write-through + drained flag beats plain stores + an L2 write-back fence per producer).
struct alignas(16) ChunkRec {
  double fit[MV_L];   // fitness desc, offer index asc
  int idx[MV_L];      // -1 = no entry
  int ge[MV_LG];      // first offers (ascending index) whose fitness exceeds good-enough; 0x7FFFFFFF = none
  unsigned cnt[4];    // n | nge << 8, offers failing on resources / constraints / zero fitness
};
static_assert(sizeof(ChunkRec) % 16 == 0, "ChunkRec is moved in 16-byte pieces");
static_assert(offsetof(ChunkRec, cnt) % 16 == 0 && offsetof(ChunkRec, cnt) + 16 == sizeof(ChunkRec), "the counts are the record's last 16-byte piece");
constexpr unsigned CHUNK_COUNT_PIECE = offsetof(ChunkRec, cnt) / 16;  // an empty list is stored from here on (chunk_store): the merge reads n = 0 and ignores the rest
// (chunk_store: platform.hpp)

struct V2Buf {
  RoundLog* round_log;
  unsigned split_max;  // cap of eval_split (1 = never cut a wave's offer batch)
#ifdef COOK_EVAL_TRACE
  unsigned long long* eval_trace;  // timing study build: per eval block [start, end] ticks of the 100 MHz clock + HW_ID
#endif
  const OfferA* oa;
  const OfferB* ob;
  const JobRec* jr;
  const JobCons* jcons;  // [K] fast constraint slots of the jobs flagged JF_FASTC
  ChunkRec* prec;      // [wmax][C]      chunk lists: one record per (job of the window, offer chunk)
  uint64_t* colbits;   // [M][JGL]       static-constraints-pass bit of (offer, job of the window)
  unsigned* jfh;       // [wlong][MV_FH + 2]  group members: the hosts their cotasks occupy under the snapshot (unique groups), how many
                       //                (int: -1 not gathered, -2 more than MV_FH), the group's last placed job — what the walk's fast
                       //                path needs, gathered ONCE by the evaluation (the tile of chunk 0 writes it)
  double* cand_fit;    // [wmax][L]
  int* cand_idx;       // [wmax][L]
  int* ge_idx;         // [wmax][LG]
  uint32_t* cinfo;     // [wmax][4]      ncand | nge << 8, c1, c2, c4
  WinCtl* ctl;
  const MatchIn* in_dev;  // the MatchIn of this call in device memory (the walk only needs it for constrained groups)
  unsigned C;          // eval blocks along the offers
};

// ---- once per match call: pack offers and jobs -----------------------------------------------------------------------
__global__ void __launch_bounds__(256) match_pack_offers(MatchIn in, OfferA* __restrict__ oa, OfferB* __restrict__ ob) {
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= in.M) return;
  OfferA a;
  a.oc = in.o_cpus[v];
  a.om = in.o_mem[v];
  a.rc = in.o_run_cpus ? in.o_run_cpus[v] : 0.0;
  a.rm = in.o_run_mem ? in.o_run_mem[v] : 0.0;
  a.inv_dc = 1.0 / (a.oc + a.rc);
  a.inv_dm = 1.0 / (a.om + a.rm);
  oa[v] = a;
  OfferB b;
  b.host = in.o_host[v];
  b.gpu_model = 0u;  // the one entry of the host's "gpus" map (or, bit2, one of several)
  b.gpu_count = 0.0;
  unsigned n_keys = 0;
  for (unsigned q = 0; in.o_gpu_model && q < in.gpu_slots; ++q) {
    const unsigned md = in.o_gpu_model[(size_t)v * in.gpu_slots + q];
    if (md != 0u) {
      if (n_keys == 0) {
        b.gpu_model = md;
        b.gpu_count = in.o_gpu_count ? in.o_gpu_count[(size_t)v * in.gpu_slots + q] : 0.0;
      }
      ++n_keys;
    }
  }
  b.run_count = in.o_run_count ? in.o_run_count[v] : 0;
  b.task_slack = (in.o_max_tasks && in.o_max_tasks[v] >= 0) ? in.o_max_tasks[v] - (in.o_num_tasks ? in.o_num_tasks[v] : 0) : 0x7FFFFFFF;
  const bool k8s = in.o_k8s && in.o_k8s[v];
  const bool rsv = in.reserved_bits && (b.host >> 5) < in.reserved_words && ((in.reserved_bits[b.host >> 5] >> (b.host & 31)) & 1u);
  b.flags = (k8s ? 1u : 0u) | (rsv ? 2u : 0u) | (n_keys > 1u ? 4u : 0u);
  b.pad = 0;
  ob[v] = b;
}

__global__ void __launch_bounds__(256) match_pack_jobs(MatchIn in, JobRec* __restrict__ jr, JobCons* __restrict__ jcons) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= in.K) return;
  const unsigned jj = in.j_index ? in.j_index[k] : k;
  JobRec j;
  j.c = in.j_cpus[jj];
  j.m = in.j_mem[jj];
  j.g = in.j_gpus ? in.j_gpus[jj] : 0.0;
  j.gpu_model = in.j_gpu_model ? in.j_gpu_model[jj] : 0u;
  j.reserved_host = in.j_reserved_host ? in.j_reserved_host[jj] : -1;
  j.group = in.j_group ? in.j_group[jj] : 0xFFFFFFFFu;
  unsigned f = 0;
  JobCons jc;
  jc.n_eq = jc.n_novel = 0;
#pragma unroll
  for (int q = 0; q < MV_NC; ++q) jc.eq_key[q] = jc.eq_val[q] = jc.novel[q] = 0u;
  const unsigned n0 = in.j_novel_off ? in.j_novel_off[jj] : 0u, n1 = in.j_novel_off ? in.j_novel_off[jj + 1] : 0u;
  const unsigned e0 = in.j_eq_off ? in.j_eq_off[jj] : 0u, e1 = in.j_eq_off ? in.j_eq_off[jj + 1] : 0u;
  bool fits = (n1 - n0) <= (unsigned)MV_NC && (e1 - e0) <= (unsigned)MV_NC;
  for (unsigned x = e0; x < e1 && fits; ++x) {
    const unsigned key = in.j_eq_key[x];
    if (key != 0xFFFFFFFFu && key >= (unsigned)MV_NA && key < in.n_attr) fits = false;  // beyond the keys staged in LDS
  }
  if (fits) {
    for (unsigned x = n0; x < n1; ++x) {
#pragma unroll
      for (int q = 0; q < MV_NC; ++q)
        if ((unsigned)q == x - n0) jc.novel[q] = in.j_novel_host[x];
    }
    for (unsigned x = e0; x < e1; ++x) {
#pragma unroll
      for (int q = 0; q < MV_NC; ++q)
        if ((unsigned)q == x - e0) {
          jc.eq_key[q] = in.j_eq_key[x];
          jc.eq_val[q] = in.j_eq_val[x];
        }
    }
    jc.n_novel = n1 - n0;
    jc.n_eq = e1 - e0;
    if (jc.n_novel || jc.n_eq) f |= JF_FASTC;
  } else {
    f |= JF_SLOW;
  }
  if (in.j_disk_req && in.j_disk_req[jj] >= 0) f |= JF_SLOW;
  if (in.j_est_end && in.j_est_end[jj] != 0) f |= JF_SLOW;
  if (in.j_ckpt && in.j_ckpt[jj] != 0) f |= JF_SLOW;
  if (in.has_x && job_has_xres(in, jj)) f |= JF_XRES;
  if (j.group != 0xFFFFFFFFu) {
    const unsigned t = in.g_type[j.group];
    if (t != 0) f |= JF_GROUPED | (t << 8);
  }
  j.flags = f;
  jr[k] = j;
  jcons[k] = jc;
}

// minimum cpus / mem over the jobs of the call (positive doubles order like their bit patterns; jmin starts at +inf)
__global__ void __launch_bounds__(256) match_job_minima(const JobRec* __restrict__ jr, unsigned K, unsigned long long* __restrict__ jmin_bits) {
  double c = __longlong_as_double(0x7FF0000000000000ll), m = c;
  bool odd = false;  // a negative or non-finite request (jmin_bits[2]: match_v3 leaves such calls to the window rounds)
  for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
    const JobRec j = jr[k];
    c = j.c < c ? j.c : c;
    m = j.m < m ? j.m : m;
    odd = odd || !(j.c >= 0.0 && j.m >= 0.0 && j.c < 1e300 && j.m < 1e300);
  }
  if (__any(odd) && lane_id() == 0) atomicOr(&jmin_bits[2], 1ull);
  // negative or NaN resources would break the ordering trick: such inputs switch the dead-offer shortcut off (minimum 0)
  if (!(c >= 0.0)) c = 0.0;
  if (!(m >= 0.0)) m = 0.0;
  for (int d = 32; d >= 1; d >>= 1) {
    const double oc = __shfl_xor(c, d, COOK_WAVE), om = __shfl_xor(m, d, COOK_WAVE);
    c = oc < c ? oc : c;
    m = om < m ? om : m;
  }
  if (lane_id() == 0) {
    atomicMin(&jmin_bits[0], (unsigned long long)__double_as_longlong(c));
    atomicMin(&jmin_bits[1], (unsigned long long)__double_as_longlong(m));
  }
}
// alive bits at the start of a call (nothing assigned yet)
__global__ void __launch_bounds__(256) match_init_alive(const OfferA* __restrict__ oa, unsigned M, const double* __restrict__ jmin,
                                                        unsigned long long* __restrict__ alive) {
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  bool a = false;
  if (v < M) {
    const OfferA o = oa[v];
    a = !(0.0 + jmin[0] > o.oc || 0.0 + jmin[1] > o.om);
  }
  const unsigned long long bits = __ballot(a);
  if (lane_id() == 0 && (v >> 6) < (M + 63u) / 64u) alive[v >> 6] = bits;
}

// ---- the cheap parts of the constraint check, from the packed records only ---------------------------------------------
// gpu-host model/count (constraints.clj:122-157) + rebalancer reservation (constraints.clj:242-252)
static __device__ __forceinline__ bool static_fast(const JobRec& j, const OfferB& o, const MatchIn& in, unsigned v) {
  bool ok;
  if (o.flags & 1u) {
    if (j.g > 0) {
      double avail = (o.gpu_model != 0 && o.gpu_model == j.gpu_model) ? o.gpu_count : 0.0;
      if (o.flags & 4u) avail = map_get_dev(in.o_gpu_model, in.o_gpu_count, in.gpu_slots, v, j.gpu_model);
      ok = avail == j.g;
    } else {
      ok = o.gpu_model == 0;
    }
  } else {
    ok = j.g == 0;
  }
  if ((o.flags & 2u) && j.reserved_host != (int)o.host) ok = false;
  return ok;
}
// gpu-host "no task on the VM" + max-tasks-per-host (constraints.clj:433-456) under `acount` placements of this call
static __device__ __forceinline__ bool dyn_fast(const JobRec& j, const OfferB& o, int acount) {
  if (j.g > 0 && (o.flags & 1u) && o.run_count + acount != 0) return false;
  return acount < o.task_slack;
}
// cpuMemBinPacker (config.clj:108), operation for operation as the oracle computes it
static __device__ __forceinline__ double fitness_of(const OfferA& a, double ac, double am, double c, double m) {
  return ((a.rc + ac + c) / (a.oc + a.rc) + (a.rm + am + m) / (a.om + a.rm)) / 2.0;
}

template <int N>
static __device__ __forceinline__ void topl_insert(double (&tf)[N], int (&ti)[N], double fit, int idx) {
  // precondition: (fit, idx) is better than the last entry; bubble it up (strictly better only: earlier index stays first)
  tf[N - 1] = fit;
  ti[N - 1] = idx;
#pragma unroll
  for (int q = N - 1; q > 0; --q) {
    const bool sw = tf[q] > tf[q - 1] || (tf[q] == tf[q - 1] && ti[q] >= 0 && (ti[q - 1] < 0 || ti[q] < ti[q - 1]));
    if (sw) {
      const double a = tf[q];
      tf[q] = tf[q - 1];
      tf[q - 1] = a;
      const int x = ti[q];
      ti[q] = ti[q - 1];
      ti[q - 1] = x;
    }
  }
}

// The same for a lane that meets its offers in ASCENDING index order (a wave's walk over its offer batch): a new entry only passes
// entries it beats strictly, so position = number of entries it beats — N independent compares and a shift by selects, no
// dependent compare-swap chain (the insertion was a third of the eval wave's time).
template <int N>
static __device__ __forceinline__ void topl_insert_ascending(double (&tf)[N], int (&ti)[N], double fit, int idx) {
  bool g[N];
#pragma unroll
  for (int q = 0; q < N; ++q) g[q] = fit > tf[q];  // monotone in q: the list descends (empty entries hold -1)
#pragma unroll
  for (int q = N - 1; q > 0; --q) {
    tf[q] = g[q - 1] ? tf[q - 1] : (g[q] ? fit : tf[q]);
    ti[q] = g[q - 1] ? ti[q - 1] : (g[q] ? idx : ti[q]);
  }
  tf[0] = g[0] ? fit : tf[0];
  ti[0] = g[0] ? idx : ti[0];
}

// The rare paths of the offer loops as real calls on the device copy of MatchIn: inlined, their CSR walks kept some forty kernel
// arguments alive across the loop and the compiler spilled scalar registers into VGPR lanes (281 v_readlane restores per offer
// iteration of the eval kernel).
static __device__ __attribute__((noinline)) bool group_pass_dev(const MatchIn* in, MatchState st, unsigned jj, unsigned v) {
  return group_pass(*in, st, jj, v);
}
static __device__ __attribute__((noinline)) bool static_pass_dev(const MatchIn* in, unsigned jj, unsigned v) { return static_pass(*in, jj, v); }
static __device__ __attribute__((noinline)) unsigned xres_fail_dev(const MatchIn* in, MatchState st, unsigned jj, unsigned v) {
  return xres_fail_bits(*in, st, jj, v);
}

// ---- eval ------------------------------------------------------------------------------------------------------------------
struct EvalWaveLds {  // what ONE wave stages for the offers it walks (MV_OCW at a time): the offer loop then reads LDS broadcasts only
  OfferA oa[MV_OCW];
  OfferB ob[MV_OCW];
  double oac[MV_OCW], oam[MV_OCW];
  int oacount[MV_OCW];
  uint32_t attr[MV_OCW][MV_NA];  // the first MV_NA attribute values of the offers (0 = absent)
};
struct EvalLds {
  double fit[MV_EW][COOK_WAVE][MV_L];
  int idx[MV_EW][COOK_WAVE][MV_L];
  int ge[MV_EW][COOK_WAVE][MV_LG];
  unsigned cnt[MV_EW][COOK_WAVE][3];
  EvalWaveLds wave[MV_EW];
};

// the job of one lane and its running results over the offers seen so far
struct EvalLane {
  bool valid, slow, grouped, fastc, use_ge;
  JobRec j;
  unsigned jj;
  unsigned k;  // the job's index in match order (vb.jr / vb.jcons)
  unsigned fh[MV_FH];
  int n_fh;
  int glast;  // the group's last placed job under the snapshot (-1 none; members of a group only)
  double ge, ge_lo;
  double tf[MV_L];
  int ti[MV_L];
  int gi[MV_LG];
  int n_ge;
  double thr;  // pruning threshold: (1 - 2^-40) * current L-th best, valid once the list is full
  unsigned c1, c2, c4;
};

// lane = job `b` of the window (64 consecutive jobs per wave): load it and gather what its constraints need
// The job's fast constraints (JobCons) in the form the offer loop checks without a per-lane LDS look-up: per attribute key staged in LDS
// the required value and an all-ones mask when the key is constrained (the offer's values are wave-uniform), the required HOSTNAME
// value, the hosts to avoid (0xFFFFFFFF = unused), and "cannot be satisfied by any offer".  Lives only inside the constraint pass of
// eval_scan_offers (22 registers that the fitness pass does not carry).
struct EvalCons {
  unsigned req[MV_NA], wild[MV_NA];
  unsigned req_host, wild_host;
  unsigned novel[MV_NC];
  bool impossible;
};
static __device__ __forceinline__ void eval_cons_setup(EvalCons& E, bool fastc, const V2Buf& vb, unsigned k) {
#pragma unroll
  for (int q = 0; q < MV_NA; ++q) E.req[q] = E.wild[q] = 0u;
  E.req_host = E.wild_host = 0u;
#pragma unroll
  for (int q = 0; q < MV_NC; ++q) E.novel[q] = 0xFFFFFFFFu;
  E.impossible = false;
  if (fastc) {
    const JobCons jc = vb.jcons[k];
#pragma unroll
    for (int q = 0; q < MV_NC; ++q) {
      if ((unsigned)q < jc.n_novel) E.novel[q] = jc.novel[q];
      if ((unsigned)q < jc.n_eq) {
        const unsigned key = jc.eq_key[q], val = jc.eq_val[q];
        if (key == 0xFFFFFFFFu) {  // "HOSTNAME" (value = host id + 1)
          if (E.wild_host && E.req_host != val) E.impossible = true;
          E.req_host = val;
          E.wild_host = 0xFFFFFFFFu;
        } else if (key >= (unsigned)MV_NA) {  // beyond the offers' attribute table: every offer reads as absent (0)
          if (val != 0u) E.impossible = true;
        } else {
#pragma unroll
          for (int a = 0; a < MV_NA; ++a)
            if ((unsigned)a == key) {
              if (E.wild[a] && E.req[a] != val) E.impossible = true;
              E.req[a] = val;
              E.wild[a] = 0xFFFFFFFFu;
            }
        }
      }
    }
  }
}

// GE = false: the launch was made for good-enough-fitness 1.0 (plain best fit, the parity setting): the good-enough list, its
// threshold and counters are compiled out of the offer loop (10 vector registers)
template <bool GE = true>
static __device__ __forceinline__ void eval_lane_setup(EvalLane& E, const MatchIn& in, const MatchState& st, const V2Buf& vb, unsigned head,
                                                       unsigned wcur, unsigned jg) {
  const unsigned lane = lane_id();
  const unsigned b = jg * COOK_WAVE + lane, k = head + b;
  E.valid = b < wcur && k < in.K;
  E.j.c = E.j.m = E.j.g = 0.0;
  E.j.gpu_model = 0;
  E.j.reserved_host = -1;
  E.j.group = 0xFFFFFFFFu;
  E.j.flags = 0;
  E.jj = 0;
  E.k = k;
  if (E.valid) {
    E.j = vb.jr[k];
    E.jj = in.j_index ? in.j_index[k] : k;
  }
  E.slow = (E.j.flags & JF_SLOW) != 0;
  E.grouped = (E.j.flags & JF_GROUPED) != 0;
  E.fastc = !E.slow && (E.j.flags & JF_FASTC) != 0;
  // unique host-placement groups (constraints.clj:586-598): the hosts to avoid = running cotasks ++ cotasks placed by
  // earlier rounds of this call, gathered ONCE per tile into registers (n_fh = -1: not such a job, -2: too many -> slow path)
  E.n_fh = -1;
  E.glast = -1;
#pragma unroll
  for (int q = 0; q < MV_FH; ++q) E.fh[q] = 0xFFFFFFFFu;
  if (E.valid && E.j.group != 0xFFFFFFFFu) E.glast = ld_agent(&st.group_last[E.j.group]);
  if (E.grouped && ((E.j.flags >> 8) & 3u) == 1u) {
    E.n_fh = 0;
    const unsigned g = E.j.group;
    const unsigned r0 = in.g_run_off ? in.g_run_off[g] : 0u, r1 = in.g_run_off ? in.g_run_off[g + 1] : 0u;
    auto push = [&](unsigned h) {
      if (E.n_fh >= 0 && E.n_fh < MV_FH) {
#pragma unroll
        for (int q = 0; q < MV_FH; ++q)
          if (q == E.n_fh) E.fh[q] = h;
        ++E.n_fh;
      } else {
        E.n_fh = -2;
      }
    };
    for (unsigned x = r0; x < r1 && E.n_fh >= 0; ++x) push(in.g_run_host[x]);
    for (int c = E.glast; c >= 0 && E.n_fh >= 0; c = ld_agent(&st.job_prev[c]))
      if (c < st.cutoff) push(in.o_host[ld_agent(&st.job_to_offer[c])]);
  }
  E.use_ge = GE && in.good_enough < 1.0;
  E.ge = in.good_enough;
  E.ge_lo = in.good_enough * (1.0 - 0x1p-40);
#pragma unroll
  for (int q = 0; q < MV_L; ++q) {
    E.tf[q] = -1.0;
    E.ti[q] = -1;
  }
#pragma unroll
  for (int q = 0; q < MV_LG; ++q) E.gi[q] = 0x7FFFFFFF;
  E.n_ge = 0;
  E.thr = -1.0;
  E.c1 = E.c2 = E.c4 = 0;
}

// When a window has fewer job groups than the eval grid has rows (the filling phase resolves ~100 jobs per round: 2 of 8 rows), the
// idle rows take a share of the OFFERS instead: with A active job groups, row gy serves job group gy % A and part gy / A of the
// R = eval_split(wcur) parts every wave's offer batch is cut into, and a chunk contributes R partial lists per job ("virtual
// chunks" ch * R + part; the merge kernel derives the same R from the same window).  R = 1 is the plain layout.
constexpr int MV_SPLIT_MAX = 4;  // a wave keeps at least MV_OCW / 4 offers; V2Buf::split_max (host) caps it: sharing a GPU with other pools'
                                 // launches, the extra blocks and the R-fold chunk lists cost more than the shorter tiles save
static __device__ __forceinline__ unsigned eval_split(unsigned wcur, unsigned split_max) {
  const unsigned active = (wcur + COOK_WAVE - 1) / COOK_WAVE;
  unsigned r = 1;
  while (r * 2u <= split_max && r * 2u * active <= (unsigned)MV_JG && (unsigned)MV_OCW / (r * 2u) >= 8u) r *= 2u;
  return r;
}

// the offers [v0, v0 + nsub) against the wave's 64 jobs (nsub = MV_OCW, or a power-of-two share of it): stage them in the wave's LDS,
// then walk them in a wave-uniform loop
template <bool THROUGH, bool GE = true>
static __device__ __forceinline__ void eval_scan_offers(EvalLane& E, EvalWaveLds& W, const MatchIn& in, const MatchState& st, const V2Buf& vb,
                                                        unsigned v0, unsigned jg, unsigned nsub = MV_OCW) {
  const unsigned lane = lane_id();
  const unsigned v1 = (v0 + nsub < in.M) ? v0 + nsub : in.M;
  if (v0 + lane < v1) {
    W.oa[lane] = vb.oa[v0 + lane];
    W.ob[lane] = vb.ob[v0 + lane];
    W.oac[lane] = st.ac[v0 + lane];
    W.oam[lane] = st.am[v0 + lane];
    W.oacount[lane] = st.acount[v0 + lane];
#pragma unroll
    for (int q = 0; q < MV_NA; ++q)
      W.attr[lane][q] = (in.o_attr && (unsigned)q < in.n_attr) ? in.o_attr[(size_t)(v0 + lane) * in.n_attr + q] : 0u;
  }
  wave_sync();
  const bool valid = E.valid;
  const JobRec& j = E.j;
  // offers that cannot take even the smallest job of the call any more fail every job on resources: count, never evaluate
  unsigned long long live = 0ull;
  if (v0 < v1) {
    live = (st.alive[v0 >> 6] >> (v0 & 63u)) & (nsub == 64u ? ~0ull : ((1ull << (nsub & 63u)) - 1ull));  // an aligned slice of one word
    if (v1 - v0 < nsub) live &= (1ull << (v1 - v0)) - 1ull;
  }
  // Two passes over the live offers, so that neither carries the other's registers (one loop held 197 VGPRs = two waves per SIMD
  // while 57 % of its wave cycles were waits): the CONSTRAINT pass — resources under the snapshot, the static checks, the colbits
  // ballot — leaves a bit per offer in two lane masks; the FITNESS pass reads the masks and never sees the constraint form.
  unsigned long long resm = 0ull, statm = 0ull;  // bit vi: the lane's job fits offer v0 + vi on resources / also passes the static checks
  {
    EvalCons Cn;
    eval_cons_setup(Cn, E.fastc, vb, E.k);
    for (unsigned long long m = live; m != 0ull;) {  // wave-uniform
      const unsigned vi = (unsigned)__ffsll((unsigned long long)m) - 1u;
      m &= m - 1ull;
      const unsigned v = v0 + vi;
      // every LDS read of this offer is issued here, in one batch
      const double oc = W.oa[vi].oc, om = W.oa[vi].om;
      const double ac = W.oac[vi], am = W.oam[vi];
      const OfferB o = W.ob[vi];
      unsigned arow[MV_NA];
#pragma unroll
      for (int x = 0; x < MV_NA; ++x) arow[x] = W.attr[vi][x];
      bool res = valid && !(ac + j.c > oc || am + j.m > om);
      if (in.has_x) {  // ports / named scalars (rare): the jobs that ask for any read the offer's counters
        if (res && (j.flags & JF_XRES)) res = xres_fail_dev(vb.in_dev, st, E.jj, v) == 0u;
      }
      if (!__any(res)) {
        if (lane == 0) {
          if (THROUGH) st_agent(&vb.colbits[(size_t)v * MV_JGL + jg], (uint64_t)0ull);
          else vb.colbits[(size_t)v * MV_JGL + jg] = 0ull;
        }
        continue;
      }
      bool stat = res && static_fast(j, o, in, v);
      {  // novel-host (constraints.clj:68-94) and user-defined EQUALS (:356-377): the offer's host and attribute values are wave-uniform
        unsigned diff = (Cn.req_host ^ (o.host + 1u)) & Cn.wild_host;
#pragma unroll
        for (int x = 0; x < MV_NA; ++x) diff |= (Cn.req[x] ^ arow[x]) & Cn.wild[x];
        bool hit = Cn.impossible;
#pragma unroll
        for (int q = 0; q < MV_NC; ++q) hit = hit | (Cn.novel[q] == o.host);
        stat = stat && diff == 0u && !hit;
      }
      if (stat && E.slow) stat = static_pass_dev(vb.in_dev, E.jj, v);
      const unsigned long long bits = __ballot(stat);
      if (lane == 0) {
        if (THROUGH) st_agent(&vb.colbits[(size_t)v * MV_JGL + jg], (uint64_t)bits);
        else vb.colbits[(size_t)v * MV_JGL + jg] = bits;
      }
      resm |= res ? 1ull << vi : 0ull;
      statm |= stat ? 1ull << vi : 0ull;
    }
  }
  unsigned long long feasm = 0ull;
  for (unsigned long long m = live; m != 0ull;) {  // wave-uniform
    const unsigned vi = (unsigned)__ffsll((unsigned long long)m) - 1u;
    m &= m - 1ull;
    const bool stat = ((statm >> vi) & 1ull) != 0ull;
    if (!__any(stat)) continue;
    const unsigned v = v0 + vi;
    const OfferA a = W.oa[vi];
    const double ac = W.oac[vi], am = W.oam[vi];
    const OfferB o = W.ob[vi];
    const int acount = W.oacount[vi];
    bool feas = stat && dyn_fast(j, o, acount);
    {  // unique host-placement groups: the hosts to avoid sit in registers (0xFFFFFFFF for everybody else)
      bool taken = false;
#pragma unroll
      for (int q = 0; q < MV_FH; ++q) taken = taken | (E.fh[q] == o.host);
      feas = feas && !taken;
    }
    if (__any(E.grouped && E.n_fh < 0)) {  // (wave-uniform) balanced / attribute-equals groups, or too many hosts: the general walk
      if (feas && E.grouped && E.n_fh < 0) feas = group_pass_dev(vb.in_dev, st, E.jj, v);
    }
    feasm |= feas ? 1ull << vi : 0ull;
    if (feas) {
      const double t1 = (a.rc + ac + j.c) * a.inv_dc, t2 = (a.rm + am + j.m) * a.inv_dm;
      const double ub = (t1 + t2) * 0.5;
      bool prune = E.ti[MV_L - 1] >= 0 && t1 >= 0.0 && t2 >= 0.0 && ub < E.thr;
      if (GE && E.use_ge && E.n_ge < MV_LG && !(ub < E.ge_lo)) prune = false;
      if (!prune) {
        const double fit = fitness_of(a, ac, am, j.c, j.m);
        if (!(fit > 0.0)) {
          E.c4 += 1u;
        } else {
          if (fit > E.tf[MV_L - 1]) {
            topl_insert_ascending<MV_L>(E.tf, E.ti, fit, (int)v);
            if (E.ti[MV_L - 1] >= 0) E.thr = E.tf[MV_L - 1] * (1.0 - 0x1p-40);
          }
          if (GE && E.use_ge && fit > E.ge && E.n_ge < MV_LG) {
#pragma unroll
            for (int q = 0; q < MV_LG; ++q)
              if (q == E.n_ge) E.gi[q] = (int)v;
            ++E.n_ge;
          }
        }
      }
    }
  }
  // failure classes: offers failing on resources (the dead ones too), offers fitting on resources but infeasible (a constraint)
  const unsigned n_res = (unsigned)__popcll(resm);
  E.c1 += valid ? (v1 > v0 ? v1 - v0 : 0u) - n_res : 0u;
  E.c2 += n_res - (unsigned)__popcll(feasm);
  wave_sync();  // every lane is done with the staged offers before the wave stages the next ones
}

// the group data of the lane's job for the walk (the tile of chunk 0 writes it, once per round)
template <bool THROUGH>
static __device__ __forceinline__ void eval_store_group(const EvalLane& E, const V2Buf& vb, unsigned b) {
  if (E.j.group == 0xFFFFFFFFu) return;
  unsigned* row = vb.jfh + (size_t)b * (MV_FH + 2);
#pragma unroll
  for (int q = 0; q < MV_FH; ++q) {
    if (THROUGH) st_agent(&row[q], E.fh[q]);
    else row[q] = E.fh[q];
  }
  if (THROUGH) {
    st_agent(&row[MV_FH], (unsigned)E.n_fh);
    st_agent(&row[MV_FH + 1], (unsigned)E.glast);
  } else {
    row[MV_FH] = (unsigned)E.n_fh;
    row[MV_FH + 1] = (unsigned)E.glast;
  }
}

// One tile = 64 jobs (job group jg of the window) x MV_OCB offers (chunk ch); the whole workgroup (MV_EW waves) takes part.
// Ends with every thread past its last LDS access only after the caller's next __syncthreads().
// The MV_EW waves may be a whole workgroup (w = wave_id(), sync = __syncthreads) or a TEAM of waves inside a larger workgroup of
// the persistent kernel (match_world.hpp: w = wave in team, sync = the team's LDS barrier, THROUGH = write-through stores).
template <bool THROUGH, bool GE = true, class Sync>
static __device__ __forceinline__ void eval_tile_t(char* lds, const MatchIn& in, const MatchState& st, const V2Buf& vb, unsigned head,
                                                   unsigned wcur, unsigned ch, unsigned jg, unsigned w, Sync sync, unsigned part = 0,
                                                   unsigned split = 1) {
  EvalLds& L = *reinterpret_cast<EvalLds*>(lds);
  auto& s_fit = L.fit;
  auto& s_idx = L.idx;
  auto& s_ge = L.ge;
  auto& s_cnt = L.cnt;
  if (jg * COOK_WAVE >= wcur || head + jg * COOK_WAVE >= in.K) return;  // uniform over the waves of the tile
  const unsigned lane = lane_id();
  const unsigned b = jg * COOK_WAVE + lane;
  EvalLane E;
#ifdef COOK_EVAL_TRACE
  unsigned long long* trp = vb.eval_trace ? vb.eval_trace + (size_t)vb.C * MV_JG * 3 + ((size_t)jg * vb.C + ch) * 16 + w * 4 : nullptr;
  if (trp && lane == 0) trp[0] = cook_ticks();
#endif
  eval_lane_setup<GE>(E, in, st, vb, head, wcur, jg);
#ifdef COOK_EVAL_TRACE
  if (trp && lane == 0) trp[1] = cook_ticks();
#endif
  eval_scan_offers<THROUGH, GE>(E, L.wave[w], in, st, vb, ch * MV_OCB + w * MV_OCW + part * ((unsigned)MV_OCW / split), jg, (unsigned)MV_OCW / split);
#ifdef COOK_EVAL_TRACE
  if (trp && lane == 0) trp[2] = cook_ticks();
#endif
  const bool valid = E.valid, use_ge = GE && E.use_ge;
  // ---- merge the block's MV_EW wave lists per job through LDS -------------------------------------------------------
#pragma unroll
  for (int q = 0; q < MV_L; ++q) {
    s_fit[w][lane][q] = E.tf[q];
    s_idx[w][lane][q] = E.ti[q];
  }
#pragma unroll
  for (int q = 0; q < MV_LG; ++q) s_ge[w][lane][q] = E.gi[q];
  s_cnt[w][lane][0] = E.c1;
  s_cnt[w][lane][1] = E.c2;
  s_cnt[w][lane][2] = E.c4;
  sync();
  if (w != 0 || !valid) return;  // (the caller synchronises the waves before the LDS is reused)
  if (ch == 0 && part == 0) eval_store_group<THROUGH>(E, vb, b);
  int p[MV_EW];
#pragma unroll
  for (int x = 0; x < MV_EW; ++x) p[x] = 0;
  ChunkRec R;
  int n_out = 0;
#pragma unroll
  for (int q = 0; q < MV_L; ++q) {
    R.fit[q] = -1.0;
    R.idx[q] = -1;
  }
#pragma unroll
  for (int q = 0; q < MV_LG; ++q) R.ge[q] = 0x7FFFFFFF;
  {
    bool more = true;
#pragma unroll
    for (int q = 0; q < MV_L; ++q) {
      Cand best{-1.0, -1};
      int bx = -1;
      if (more) {
#pragma unroll
        for (int x = 0; x < MV_EW; ++x) {
          if (p[x] < MV_L) {
            const Cand o{s_fit[x][lane][p[x]], s_idx[x][lane][p[x]]};
            if (o.idx >= 0 && cand_better(o, best)) {
              best = o;
              bx = x;
            }
          }
        }
      }
      if (bx < 0) {
        more = false;
      } else {
        R.fit[q] = best.fit;
        R.idx[q] = best.idx;
        ++n_out;
#pragma unroll
        for (int x = 0; x < MV_EW; ++x)
          if (x == bx) ++p[x];
      }
    }
  }
  int n_g = 0;
  if (use_ge) {
#pragma unroll
    for (int x = 0; x < MV_EW; ++x) p[x] = 0;
    bool more = true;
#pragma unroll
    for (int q = 0; q < MV_LG; ++q) {
      int best = 0x7FFFFFFF, bx = -1;
      if (more) {
#pragma unroll
        for (int x = 0; x < MV_EW; ++x) {
          if (p[x] < MV_LG) {
            const int o = s_ge[x][lane][p[x]];
            if (o < best) {
              best = o;
              bx = x;
            }
          }
        }
      }
      if (bx < 0) {
        more = false;
      } else {
        R.ge[q] = best;
        ++n_g;
#pragma unroll
        for (int x = 0; x < MV_EW; ++x)
          if (x == bx) ++p[x];
      }
    }
  }
  unsigned t1 = 0, t2 = 0, t4 = 0;
#pragma unroll
  for (int x = 0; x < MV_EW; ++x) {
    t1 += s_cnt[x][lane][0];
    t2 += s_cnt[x][lane][1];
    t4 += s_cnt[x][lane][2];
  }
  R.cnt[0] = (unsigned)n_out | ((unsigned)n_g << 8);
  R.cnt[1] = t1;
  R.cnt[2] = t2;
  R.cnt[3] = t4;
  chunk_store(&vb.prec[(size_t)b * (vb.C * split) + ch * split + part], R, THROUGH, (n_out | n_g) == 0 ? CHUNK_COUNT_PIECE : 0u);
#ifdef COOK_EVAL_TRACE
  if (trp && lane == 0) trp[3] = cook_ticks();
#endif
}
template <bool GE = true>
static __device__ __forceinline__ void eval_tile(char* lds, const MatchIn& in, const MatchState& st, const V2Buf& vb, unsigned head,
                                                 unsigned wcur, unsigned ch, unsigned jg, unsigned part = 0, unsigned split = 1) {
  eval_tile_t<false, GE>(lds, in, st, vb, head, wcur, ch, jg, wave_id(), [] { __syncthreads(); }, part, split);
}

// The same tile by ONE wave on its own (the persistent placement kernel's evaluator waves, match_world.hpp): 64 jobs x the
// MV_OCB offers of chunk ch in MV_EW batches of MV_OCW; no workgroup barrier anywhere, the chunk list goes straight to HBM.
template <bool THROUGH, bool GE = true>
static __device__ __forceinline__ void eval_tile_wave(EvalWaveLds& W, const MatchIn& in, const MatchState& st, const V2Buf& vb, unsigned head,
                                                      unsigned wcur, unsigned ch, unsigned jg) {
  if (jg * COOK_WAVE >= wcur || head + jg * COOK_WAVE >= in.K) return;  // wave-uniform
  const unsigned lane = lane_id();
  const unsigned b = jg * COOK_WAVE + lane;
  EvalLane E;
  eval_lane_setup<GE>(E, in, st, vb, head, wcur, jg);
  for (int s = 0; s < MV_EW; ++s) {
    const unsigned v0 = ch * MV_OCB + (unsigned)s * MV_OCW;
    if (v0 >= in.M) break;
    eval_scan_offers<THROUGH, GE>(E, W, in, st, vb, v0, jg);
  }
  if (!E.valid) return;
  if (ch == 0) eval_store_group<THROUGH>(E, vb, b);
  ChunkRec R;
  int n_out = 0, n_g = 0;
#pragma unroll
  for (int q = 0; q < MV_L; ++q) {
    R.fit[q] = E.tf[q];
    R.idx[q] = E.ti[q];
    n_out += E.ti[q] >= 0 ? 1 : 0;
  }
#pragma unroll
  for (int q = 0; q < MV_LG; ++q) {
    R.ge[q] = E.gi[q];
    n_g += E.gi[q] != 0x7FFFFFFF ? 1 : 0;
  }
  R.cnt[0] = (unsigned)n_out | ((unsigned)n_g << 8);
  R.cnt[1] = E.c1;
  R.cnt[2] = E.c2;
  R.cnt[3] = E.c4;
  chunk_store(&vb.prec[(size_t)b * vb.C + ch], R, THROUGH, (n_out | n_g) == 0 ? CHUNK_COUNT_PIECE : 0u);
}

// What one block of the eval grid (offer chunks x MV_JG) does.  A window of the usual size: the block's MV_EW waves share ONE tile
// (job group gy of chunk ch, a batch of offers each).  A LONG window (more job groups than the grid has rows; nearly all its offers
// are dead by then, so a tile is little more than its prologue): every wave takes a job group of its own and walks the whole chunk,
// eval_tile_wave — MV_EW job groups per pass instead of one.
template <bool GE = true>
static __device__ __forceinline__ void eval_block(char* lds, const MatchIn& in, const MatchState& st, const V2Buf& vb, unsigned head,
                                                  unsigned wcur, unsigned ch, unsigned gy, unsigned ny) {
  if (wcur <= ny * COOK_WAVE) {
    const unsigned split = ny == (unsigned)MV_JG ? eval_split(wcur, vb.split_max) : 1u;  // (the grid's rows are MV_JG in every launch path)
    if (split == 1u) {
      eval_tile<GE>(lds, in, st, vb, head, wcur, ch, gy);
    } else {
      const unsigned active = (wcur + COOK_WAVE - 1) / COOK_WAVE;
      if (gy < active * split) eval_tile<GE>(lds, in, st, vb, head, wcur, ch, gy % active, gy / active, split);
    }
    return;
  }
  EvalLds& L = *reinterpret_cast<EvalLds*>(lds);
  const unsigned w = wave_id();
  for (unsigned jg = gy * MV_EW + w; jg * COOK_WAVE < wcur; jg += ny * MV_EW) eval_tile_wave<false, GE>(L.wave[w], in, st, vb, head, wcur, ch, jg);
}
template <bool GE>
__global__ void __launch_bounds__(COOK_WAVE* MV_EW) COOK_EVAL_OCCUPANCY match_eval2(MatchIn in, MatchState st, V2Buf vb) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(EvalLds)];
#ifdef COOK_EVAL_TRACE
  const unsigned long long t0 = cook_ticks();
#endif
  eval_block<GE>(lds, in, st, vb, vb.ctl->head, vb.ctl->wcur, blockIdx.x, blockIdx.y, gridDim.y);
#ifdef COOK_EVAL_TRACE
  __syncthreads();
  if (vb.eval_trace && threadIdx.x == 0) {
    const unsigned blk = blockIdx.y * gridDim.x + blockIdx.x;
    vb.eval_trace[blk * 3 + 0] = t0;
    vb.eval_trace[blk * 3 + 1] = cook_ticks();
    vb.eval_trace[blk * 3 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
  }
#endif
}

// ---- merge: one wave per job ---------------------------------------------------------------------------------------------
template <bool THROUGH>
static __device__ __forceinline__ void merge_job(const MatchIn& in, const V2Buf& vb, unsigned head, unsigned wcur, unsigned b,
                                                 unsigned split = 1) {  // split: eval_split(wcur) behind the launch path's eval grid
  if (b >= wcur || head + b >= in.K) return;
  const unsigned lane = lane_id();
  const bool use_ge = in.good_enough < 1.0;
  double tf[MV_L];
  int ti[MV_L];
  int gi[MV_LG];
#pragma unroll
  for (int q = 0; q < MV_L; ++q) {
    tf[q] = -1.0;
    ti[q] = -1;
  }
#pragma unroll
  for (int q = 0; q < MV_LG; ++q) gi[q] = 0x7FFFFFFF;
  int n_ge = 0;
  unsigned c1 = 0, c2 = 0, c4 = 0;
  int n_seen = 0;     // (MV_LM_EXT) entries of all the lane's chunks
  bool hide = false;  // (MV_LM_EXT) see MV_LM
  const unsigned cv = vb.C * split;  // chunk lists per job (virtual chunks, eval_split)
  for (unsigned ch = lane; ch < cv; ch += COOK_WAVE) {
    const ChunkRec R = vb.prec[(size_t)b * cv + ch];  // eight 16-byte loads, all in flight together
    const unsigned info = R.cnt[0];
    c1 += R.cnt[1];
    c2 += R.cnt[2];
    c4 += R.cnt[3];
    const int n = (int)(info & 0xFFu), ng = (int)((info >> 8) & 0xFFu);
    if (MV_LM_EXT) {  // this lane's list may end before the chunk's (or chunks') feasible offers do
      n_seen += n;
      hide = hide || n == MV_L || n_seen > MV_L;
    }
    if (ch < (unsigned)COOK_WAVE) {  // the lane's first chunk (its only one up to 64 chunks = 8 192 offers): the sorted list as it is
#pragma unroll
      for (int q = 0; q < MV_L; ++q)
        if (q < n) tf[q] = R.fit[q], ti[q] = R.idx[q];
    } else {  // a later (virtual) chunk holds higher offer indices than everything the lane has seen: an entry only passes entries it
              // beats strictly, and equal-fitness entries of its own list arrive in index order
#pragma unroll
      for (int q = 0; q < MV_L; ++q) {
        if (q >= n) break;
        if (!(R.fit[q] > tf[MV_L - 1])) break;  // chunk list is sorted: nothing further can enter
        topl_insert_ascending<MV_L>(tf, ti, R.fit[q], R.idx[q]);
      }
    }
    if (use_ge)
#pragma unroll
      for (int q = 0; q < MV_LG; ++q) {  // chunks ascend with ch, entries ascend inside a chunk
        if (q >= ng || n_ge >= MV_LG) break;
        const int o = R.ge[q];
#pragma unroll
        for (int x = 0; x < MV_LG; ++x)
          if (x == n_ge) gi[x] = o;
        ++n_ge;
      }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    c1 += __shfl_xor(c1, d, COOK_WAVE);
    c2 += __shfl_xor(c2, d, COOK_WAVE);
    c4 += __shfl_xor(c4, d, COOK_WAVE);
  }
  int n_out = 0;
  bool trunc = false;  // (MV_LM_EXT) the merged list may not hold every feasible offer
  for (int round = 0; round < MV_LM; ++round) {
    // the best head over the lanes: greatest fitness (positive doubles order like their bit patterns), lowest offer index among
    // equal ones — two DPP reductions instead of six rounds of three ds_bpermute shuffles
    const unsigned long long key = ti[0] >= 0 ? (unsigned long long)__double_as_longlong(tf[0]) : 0ull;
    const unsigned long long mk = wave_max_u64(key);
    if (mk == 0ull) break;  // wave-uniform
    const unsigned long long tie = __ballot(key == mk);
    Cand best{__longlong_as_double((long long)mk), 0};
    if ((tie & (tie - 1ull)) == 0ull)
      best.idx = wave_read_lane(ti[0], __ffsll((unsigned long long)tie) - 1);
    else
      best.idx = (int)(0x7FFFFFFFu - wave_max_u32(key == mk ? 0x7FFFFFFFu - (unsigned)ti[0] : 0u));
    if (lane == 0) {
      if (THROUGH) {
        st_agent(&vb.cand_fit[(size_t)b * MV_LM + round], best.fit);
        st_agent(&vb.cand_idx[(size_t)b * MV_LM + round], best.idx);
      } else {
        vb.cand_fit[(size_t)b * MV_LM + round] = best.fit;
        vb.cand_idx[(size_t)b * MV_LM + round] = best.idx;
      }
    }
    ++n_out;
    bool emptied = false;
    if (ti[0] == best.idx) {  // the owner pops its head
#pragma unroll
      for (int q = 0; q < MV_L - 1; ++q) {
        tf[q] = tf[q + 1];
        ti[q] = ti[q + 1];
      }
      tf[MV_L - 1] = -1.0;
      ti[MV_L - 1] = -1;
      emptied = ti[0] < 0 && hide;
    }
    if (MV_LM_EXT && __any(emptied)) {  // a list that may continue beyond what the lane holds just ran out: stop here
      trunc = true;
      break;
    }
  }
  if (MV_LM_EXT && !trunc) trunc = __any(ti[0] >= 0);  // LM entries emitted and some lane still holds more
  int n_g = 0;
  if (use_ge) {
    for (int round = 0; round < MV_LG; ++round) {
      int best = gi[0];
      for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(best, d, COOK_WAVE);
        best = o < best ? o : best;
      }
      if (best == 0x7FFFFFFF) break;
      if (lane == 0) {
        if (THROUGH) st_agent(&vb.ge_idx[(size_t)b * MV_LG + round], best);
        else vb.ge_idx[(size_t)b * MV_LG + round] = best;
      }
      ++n_g;
      if (gi[0] == best) {
#pragma unroll
        for (int q = 0; q < MV_LG - 1; ++q) gi[q] = gi[q + 1];
        gi[MV_LG - 1] = 0x7FFFFFFF;
      }
    }
  }
  if (lane == 0) {
    if (THROUGH) {
      st_agent(&vb.cinfo[(size_t)b * 4 + 0], (uint32_t)((unsigned)n_out | ((unsigned)n_g << 8) | ((MV_LM_EXT && trunc) ? 1u << 16 : 0u)));
      st_agent(&vb.cinfo[(size_t)b * 4 + 1], (uint32_t)c1);
      st_agent(&vb.cinfo[(size_t)b * 4 + 2], (uint32_t)c2);
      st_agent(&vb.cinfo[(size_t)b * 4 + 3], (uint32_t)c4);
    } else {
      vb.cinfo[(size_t)b * 4 + 0] = (unsigned)n_out | ((unsigned)n_g << 8) | ((MV_LM_EXT && trunc) ? 1u << 16 : 0u);
      vb.cinfo[(size_t)b * 4 + 1] = c1;
      vb.cinfo[(size_t)b * 4 + 2] = c2;
      vb.cinfo[(size_t)b * 4 + 3] = c4;
    }
  }
}

// one wave per job; a block of MV_MW waves takes MV_MW jobs per pass (a long window needs several passes)
constexpr int MV_MW = 4;
__global__ void __launch_bounds__(COOK_WAVE* MV_MW) match_merge2(MatchIn in, V2Buf vb) {
  const unsigned head = vb.ctl->head, wcur = vb.ctl->wcur;
  const unsigned split = wcur <= (unsigned)MV_WMAX ? eval_split(wcur, vb.split_max) : 1u;  // as match_eval2's grid cut the offers
  for (unsigned b = blockIdx.x * MV_MW + wave_id(); b < wcur; b += gridDim.x * MV_MW) merge_job<false>(in, vb, head, wcur, b, split);
}

// ---- resolve -----------------------------------------------------------------------------------------------------------------
struct SlotRec {  // one distinct candidate offer of the window, staged in LDS
  OfferA a;
  OfferB o;
  double ac, am;  // snapshot state
  int acount;
  int offer;
};
struct JobL {  // a job of the window as the walk reads it (one 32-byte LDS record)
  double c, m;
  unsigned info;  // bits 0-7 ncand, 8-15 nge, 16 gpu job, 17 member of a constrained group, 18-19 group type
  unsigned group;
  unsigned short f1, f2, f4;  // saturated counts of offers failing on resources / constraints / zero fitness under S
  unsigned short b;           // window position of the job (the record itself sits at its WALK position)
};
struct EntL {  // candidate-list entry (16 bytes): fitness under S, offer, slot
  double fit;
  int off;
  unsigned short slot, pad;
};
struct GEntL {  // good-enough list entry
  int off;
  unsigned short slot, pad;
};
constexpr unsigned JL_GPU = 1u << 16, JL_GROUPED = 1u << 17, JL_HASGROUP = 1u << 20;  // (bits 18-19: group type)
constexpr unsigned JL_XRES = 1u << 28;  // asks for ports / named scalars: general path only
constexpr unsigned JL_TRUNC = 1u << 29;  // (MV_LM_EXT) the merged list may not hold every feasible offer (cinfo bit 16)
// "entries may exist beyond the job's list" / "the list holds every feasible offer" inside the walk (cinfo_u, nc: the walk's locals)
#if COOK_MV_LM > COOK_MV_L
#define COOK_L_TRUNC() ((cinfo_u & JL_TRUNC) != 0u)
#define COOK_L_COMPLETE() ((cinfo_u & JL_TRUNC) == 0u)
#else
#define COOK_L_TRUNC() (nc == MV_L)
#define COOK_L_COMPLETE() (nc < MV_L)
#endif
constexpr unsigned JL_GSLOT_SHIFT = 21, JL_GSLOT_NONE = 0x7Fu;  // bits 21-27: the job's row of ResolveLds::gfh, or none
constexpr int MV_GMAX = 64;  // group members per round whose hosts-to-avoid are staged for the walk's fast path



// (WALK_STAT: platform.hpp — counters of the emulated build's design studies, nothing on the GPU)

struct ResolveLds {
  JobL job[MV_WMAX];          // the jobs the walk visits, in rank order (walk position i; JobL::b = window position)
  EntL ent[MV_WMAX][MV_LM];   // their candidate lists, by walk position
  GEntL gent[MV_WMAX][MV_LG];
  SlotRec slot[MV_S];
  unsigned long long col[MV_S][MV_JG];  // static-constraints-pass bits of (slot, walked job), by WALK position (bit i & 63 of word i >> 6)
  unsigned long long visit[MV_JGL];
  double tac[MV_T], tam[MV_T];  // current state of the touched offers, by owner lane (published for a re-evaluation)
  double rfit[MV_RWAVES_MAX];
  int hkey[MV_HASH];
  int j2o[MV_WMAX];                 // results of the walk BY WALK POSITION, flushed to HBM once per round: a global store inside
  int tacount[MV_T];                // the walk would stall later s_waitcnt vmcnt(0) on its acknowledgement
  int ridx[MV_RWAVES_MAX], rge[MV_RWAVES_MAX];
  unsigned rc[MV_RWAVES_MAX][3];
  unsigned vbase[MV_JGL + 1];       // walk position of the first visited job of each 64-job group
  // members of unique (or unconstrained) host-placement groups among the walked jobs: the hosts their cotasks occupied when the round
  // began (running ++ placed by earlier rounds; 0xFFFFFFFF = unused) and the group's last placed job then
  unsigned gfh[MV_GMAX][MV_FH];
  int glast[MV_GMAX];
  unsigned n_gslots;
  unsigned nslots, minbad;
  int cmd;                          // window index of the job to re-evaluate, -1 = the walk is over
  unsigned short hslot[MV_HASH];
  unsigned char slot_lane[MV_S];
  unsigned char fail[MV_WMAX];      // by walk position
  // ports / named scalars assigned on a touched offer when the round began, by owner lane: saved by the first job of the round
  // that moves them (the failure summary of an unmatched job compares against the round's snapshot)
  double x0s[MV_T][3];
  int x0p[MV_T];
  unsigned char x0set[MV_T];
};

// One round of the window walk by ONE workgroup of MV_RTHREADS threads (all of them must call it).
// REEVAL compiles the in-place re-evaluation of list-exhausted jobs in (match_algo 3); without it the helper waves leave after
// the set-up phase and the walk loop carries none of that machinery (it cost the default path ~10 % of the walk).
//
// The walk is one dependent chain run by a single wave, so what it costs per job is latency: measured on MI355X
// (scripts/ubench_wave.hip) a dependent LDS read is 60-68 cycles, a 6-step DPP reduction 166 (compiler form), a ballot -> ffs ->
// readlane hop 62, a wave-uniform branch ~25, against 48 for the fp64 evaluation of a touched offer itself.  The loop is
// therefore organised as (1) a two-deep software pipeline over walk records that are laid out by WALK position (no dependent
// address chain: record and list entries of job i+2 and the owner look-up of job i+1 are in flight while job i is decided),
// (2) a FAST PATH for the common job — no constrained group, good-enough disabled, finite positive fitness values — that
// orders the touched offers by an fp32 image of the approximate fitness (one hand-placed DPP reduction, common.hpp) and falls
// back to (3) the GENERAL PATH below it whenever the order is not certain at fp32 resolution (two touched offers within 2^-20,
// touched and untouched best within 2^-38), the job is unmatched, or anything unusual is involved.  Both paths produce the same
// decision; only the general path knows every rule.
template <bool REEVAL>
static __device__ void resolve_round(char* lds, MatchState st, const V2Buf& vb) {
  ResolveLds& L = *reinterpret_cast<ResolveLds*>(lds);
  auto& s_job = L.job;
  auto& s_ent = L.ent;
  auto& s_gent = L.gent;
  auto& s_slot = L.slot;
  auto& s_col = L.col;
  auto& s_slot_lane = L.slot_lane;
  auto& s_hkey = L.hkey;
  auto& s_hslot = L.hslot;
  auto& s_j2o = L.j2o;
  auto& s_fail = L.fail;
  auto& s_visit = L.visit;
  auto& s_vbase = L.vbase;
  auto& s_gfh = L.gfh;
  auto& s_glast = L.glast;
  unsigned& s_ngslots = L.n_gslots;
  unsigned& s_nslots = L.nslots;
  unsigned& s_minbad = L.minbad;
  auto& s_tac = L.tac;
  auto& s_tam = L.tam;
  auto& s_tacount = L.tacount;
  int& s_cmd = L.cmd;
  auto& s_rfit = L.rfit;
  auto& s_ridx = L.ridx;
  auto& s_rge = L.rge;
  auto& s_rc = L.rc;
  const unsigned tid = threadIdx.x, lane = lane_id(), NT = blockDim.x;
  WinCtl ctl = *vb.ctl;
  const unsigned head = ctl.head;
  const unsigned K = vb.in_dev->K;
  if (head >= K) return;
  const unsigned long long tk0 = cook_ticks();
  const unsigned wend = (head + ctl.wcur < K) ? head + ctl.wcur : K;
  const unsigned nwin = wend - head;
  const double good_enough = vb.in_dev->good_enough;
  const bool use_ge = good_enough < 1.0;
  const uint32_t* const j_index = vb.in_dev->j_index;
  // ---- set-up phase (all threads): stage the window in LDS -------------------------------------------------------------
  for (unsigned x = tid; x < MV_HASH; x += NT) s_hkey[x] = -1;
  if (tid < MV_JGL) s_visit[tid] = 0ull;
  if (tid < (unsigned)MV_T) L.x0set[tid] = 0;
  if (tid == 0) {
    s_nslots = 0;
    s_minbad = 0xFFFFFFFFu;
    s_ngslots = 0;
  }
  __syncthreads();
  // A job without any feasible offer under S stays unmatched whatever the jobs before it do (placements only take
  // capacity away; constrained groups excepted), and its failure summary cannot change when every class it reports is
  // backed by more offers than a round can touch: such jobs are settled here, in parallel, and the walk skips them.
  for (unsigned b = tid; b < nwin; b += NT) {
    const unsigned flags = vb.jr[head + b].flags;
    const unsigned info = vb.cinfo[(size_t)b * 4 + 0];
    const unsigned c1 = vb.cinfo[(size_t)b * 4 + 1], c2 = vb.cinfo[(size_t)b * 4 + 2], c4 = vb.cinfo[(size_t)b * 4 + 3];
    // members of balanced / attribute-equals groups excepted: a cotask's placement can make an offer FEASIBLE for them; a unique
    // group only ever takes hosts away (constraints.clj:586-598), like a resource
    const bool opens = (flags & JF_GROUPED) != 0 && ((flags >> 8) & 3u) != 1u;
    const bool trivial = (info & 0xFFFFu) == 0u && !opens && c1 > 0u && (c2 == 0u || c2 > (unsigned)MV_T) &&
                         (c4 == 0u || c4 > (unsigned)MV_T);
    if (trivial) {
      // final whatever this round does, also for a job behind the point where the round stops: job_to_offer keeps the -1 it was
      // initialised with; should the job still be unresolved next round, its summary is simply rewritten under the newer snapshot
      if (st.fail_code) st.fail_code[head + b] = 1u | (c2 ? 2u : 0u) | (c4 ? 4u : 0u);
    } else {
      atomicOr(&s_visit[b >> 6], 1ull << (b & 63u));
    }
  }
  __syncthreads();
  if (tid == 0) {
    unsigned acc = 0;
    const unsigned ng = (nwin + COOK_WAVE - 1) / COOK_WAVE;
    for (unsigned g = 0; g < ng; ++g) {
      s_vbase[g] = acc;
      acc += (unsigned)__popcll(s_visit[g]);
    }
    for (unsigned g = ng; g <= (unsigned)MV_JGL; ++g) s_vbase[g] = acc;  // (walkpos_to_b scans on; [MV_JGL] = the total)
  }
  __syncthreads();
  const unsigned n_list = s_vbase[MV_JGL];  // jobs the walk has to visit
  const unsigned n_walk = n_list < (unsigned)MV_WMAX ? n_list : (unsigned)MV_WMAX;  // ... and can stage in this round
  // walk records + candidate lists of the visited jobs -> LDS, by walk position (one parallel pass; the slot-table passes below
  // then never touch HBM)
  constexpr int EPJ = MV_LM + MV_LG;
  for (unsigned e = tid; e < nwin * (EPJ + 1); e += NT) {
    const unsigned b = e / (EPJ + 1), q = e % (EPJ + 1);
    const unsigned long long vw = s_visit[b >> 6];
    if (!((vw >> (b & 63u)) & 1ull)) continue;
    const unsigned i = s_vbase[b >> 6] + (unsigned)__popcll(vw & ((1ull << (b & 63u)) - 1ull));
    if (i >= n_walk) continue;
    const unsigned info = vb.cinfo[(size_t)b * 4 + 0];
    if (q == (unsigned)EPJ) {
      s_fail[i] = 0;  // a visited job that gets matched leaves it at that
      const JobRec j = vb.jr[head + b];
      const unsigned c1 = vb.cinfo[(size_t)b * 4 + 1], c2 = vb.cinfo[(size_t)b * 4 + 2], c4 = vb.cinfo[(size_t)b * 4 + 3];
      JobL r;
      r.c = j.c;
      r.m = j.m;
      const bool grouped = (j.flags & JF_GROUPED) != 0;
      r.info = (info & 0xFFFFu) | (j.g > 0 ? JL_GPU : 0u) | (grouped ? JL_GROUPED : 0u) | (((j.flags >> 8) & 3u) << 18) |
               (j.group != 0xFFFFFFFFu ? JL_HASGROUP : 0u) | ((j.flags & JF_XRES) ? JL_XRES : 0u) |
               ((MV_LM_EXT && (info & (1u << 16))) ? JL_TRUNC : 0u);
      // a member of a unique (type 1) or unconstrained (type 0) group: stage what the walk's fast path needs — the hosts to avoid
      // as the round begins and the group's last placed job (for the chain link) — so that it never has to go to HBM for them
      unsigned gslot = JL_GSLOT_NONE;
      const unsigned gt = (j.flags >> 8) & 3u;
      if (j.group != 0xFFFFFFFFu && gt <= 1u && !use_ge && vb.in_dev->host_dup == 0u && !(j.flags & JF_XRES)) {
        const unsigned* row = vb.jfh + (size_t)b * (MV_FH + 2);  // gathered by the evaluation of this round
        const int nfh = (int)row[MV_FH];
        if (gt == 0u || (nfh >= 0 && nfh <= MV_FH)) {
          const unsigned gs = atomicAdd(&s_ngslots, 1u);
          if (gs < (unsigned)MV_GMAX) {
#pragma unroll
            for (int x = 0; x < MV_FH; ++x) s_gfh[gs][x] = gt == 1u ? row[x] : 0xFFFFFFFFu;
            s_glast[gs] = (int)row[MV_FH + 1];
            gslot = gs;
          }
        }
      }
      r.info |= gslot << JL_GSLOT_SHIFT;
      r.group = j.group;
      r.f1 = (unsigned short)(c1 < 0xFFFFu ? c1 : 0xFFFFu);
      r.f2 = (unsigned short)(c2 < 0xFFFFu ? c2 : 0xFFFFu);
      r.f4 = (unsigned short)(c4 < 0xFFFFu ? c4 : 0xFFFFu);
      r.b = (unsigned short)b;
      s_job[i] = r;
    } else if (q < (unsigned)MV_LM) {
      EntL x;
      x.fit = -1.0;
      x.off = -1;
      x.slot = 0;
      x.pad = 0;
      if (q < (info & 0xFFu)) {
        x.off = vb.cand_idx[(size_t)b * MV_LM + q];
        x.fit = vb.cand_fit[(size_t)b * MV_LM + q];
      }
      s_ent[i][q] = x;
    } else {
      GEntL x;
      x.off = -1;
      x.slot = 0;
      x.pad = 0;
      if (use_ge && q - MV_LM < ((info >> 8) & 0xFFu)) x.off = vb.ge_idx[(size_t)b * MV_LG + (q - MV_LM)];
      s_gent[i][q - MV_LM] = x;
    }
  }
  __syncthreads();
  // slot table = the DISTINCT candidate offers.  Optimistic pass: insert every entry of the window at once; if the table
  // overflows (rare) redo it MV_JSTEP jobs at a time so that the overflow cuts the walk at a job boundary (every job
  // before the cut has all its candidates staged).
  for (int pass = 0; pass < 2; ++pass) {
    const unsigned step = pass == 0 ? (n_walk ? n_walk : 1u) : (unsigned)MV_JSTEP;
    bool overflow = false;
    for (unsigned s0 = 0; s0 < n_walk; s0 += step) {
      const unsigned e1 = ((s0 + step < n_walk) ? s0 + step : n_walk) * EPJ;
      for (unsigned e = s0 * EPJ + tid; e < e1; e += NT) {
        const unsigned i = e / EPJ, q = e % EPJ;
        const int idx = q < (unsigned)MV_LM ? s_ent[i][q].off : s_gent[i][q - MV_LM].off;
        if (idx < 0) continue;
        unsigned h = ((unsigned)idx * 2654435761u) % MV_HASH;
        for (;;) {
          const int old = atomicCAS(&s_hkey[h], -1, idx);
          if (old == -1) {  // creator: allocate the slot
            const unsigned s = atomicAdd(&s_nslots, 1u);
            s_hslot[h] = (unsigned short)(s < (unsigned)MV_S ? s : 0xFFFFu);
            if (s < (unsigned)MV_S) {
              s_slot[s].offer = idx;
            } else {
              atomicMin(&s_minbad, s0);
            }
            break;
          }
          if (old == idx) break;
          h = (h + 1) % MV_HASH;
          if (pass == 0 && s_nslots > (unsigned)MV_S) break;  // the optimistic pass already failed: stop filling the table
        }
      }
      __syncthreads();
      overflow = s_nslots > (unsigned)MV_S;  // block-uniform: read between two barriers
      __syncthreads();
      if (overflow) break;
    }
    if (!overflow || pass == 1) break;
    // overflow in the optimistic pass: reset the table and go stepwise
    for (unsigned x = tid; x < MV_HASH; x += NT) s_hkey[x] = -1;
    if (tid == 0) {
      s_nslots = 0;
      s_minbad = 0xFFFFFFFFu;
    }
    __syncthreads();
  }
  const unsigned n_eff = s_minbad < n_walk ? s_minbad : n_walk;  // walk positions resolvable in this round
  for (unsigned e = tid; e < n_eff * EPJ; e += NT) {  // candidate offer -> slot
    const unsigned i = e / EPJ, q = e % EPJ;
    const int idx = q < (unsigned)MV_LM ? s_ent[i][q].off : s_gent[i][q - MV_LM].off;
    if (idx < 0) continue;
    unsigned h = ((unsigned)idx * 2654435761u) % MV_HASH;
    while (s_hkey[h] != idx) h = (h + 1) % MV_HASH;
    if (q < (unsigned)MV_LM)
      s_ent[i][q].slot = s_hslot[h];
    else
      s_gent[i][q - MV_LM].slot = s_hslot[h];
  }
  const unsigned nslots = s_nslots < (unsigned)MV_S ? s_nslots : (unsigned)MV_S;
  for (unsigned s = tid; s < nslots; s += NT) {
    const int v = s_slot[s].offer;
    s_slot[s].a = vb.oa[v];
    s_slot[s].o = vb.ob[v];
    s_slot[s].ac = st.ac[v];
    s_slot[s].am = st.am[v];
    s_slot[s].acount = st.acount[v];
    s_slot_lane[s] = 0xFF;
  }
  // colbits columns of the slots.  A window of the usual size: as they are, by window position.  A LONG window (its job groups
  // do not fit the LDS array): compacted to the WALKED jobs — word g of a column holds the 64 jobs of window group g; the bits of
  // the visited ones (s_visit[g]) go to walk positions s_vbase[g] ... in order.
  const bool lw = nwin > (unsigned)MV_WMAX;
  const unsigned ngrp = (nwin + COOK_WAVE - 1) / COOK_WAVE;
  if (!lw) {
    for (unsigned x = tid; x < nslots * MV_JG; x += NT) {
      const unsigned sl = x / MV_JG, g = x % MV_JG;
      s_col[sl][g] = (g * COOK_WAVE < nwin) ? vb.colbits[(size_t)s_slot[sl].offer * MV_JGL + g] : 0ull;
    }
  }
  for (unsigned sl = tid; lw && sl < nslots; sl += NT) {  // one thread per slot: no atomics, no extra barrier
    unsigned long long acc[MV_JG];
#pragma unroll
    for (int w = 0; w < MV_JG; ++w) acc[w] = 0ull;
    const uint64_t* colp = vb.colbits + (size_t)s_slot[sl].offer * MV_JGL;
    for (unsigned g = 0; g < ngrp; ++g) {
      const unsigned long long V = s_visit[g];
      const unsigned base = s_vbase[g];
      if (V == 0ull || base >= n_walk) continue;
      const unsigned long long Wd = colp[g];
      unsigned long long packed;
      if (V == ~0ull) {
        packed = Wd;
      } else {  // parallel bit extract of Wd under V
        packed = 0ull;
        unsigned o = 0;
        for (unsigned long long m = V; m != 0ull; m &= m - 1ull, ++o) packed |= ((Wd >> (__ffsll((unsigned long long)m) - 1)) & 1ull) << o;
      }
      const unsigned w0 = base >> 6, sh = base & 63u;
#pragma unroll
      for (int w = 0; w < MV_JG; ++w) {
        if ((unsigned)w == w0) acc[w] |= packed << sh;
        if (sh != 0u && (unsigned)w == w0 + 1u) acc[w] |= packed >> (64u - sh);
      }
    }
#pragma unroll
    for (int w = 0; w < MV_JG; ++w) s_col[sl][w] = acc[w];
  }
  if (tid == 0) s_cmd = -1;
  __syncthreads();
  // All offers x one job under the CURRENT state (snapshot for untouched offers, s_t* for touched ones): per-wave partial
  // results go to s_r*; every thread of the workgroup takes part (wave 0 asks for it through s_cmd + two barriers).
  auto reeval = [&](unsigned b) {
    const MatchIn& in = *vb.in_dev;
    const unsigned k = head + b;
    const JobRec j = vb.jr[k];
    const unsigned jj = in.j_index ? in.j_index[k] : k;
    const bool slow = (j.flags & (JF_SLOW | JF_FASTC)) != 0, grouped = (j.flags & JF_GROUPED) != 0;  // any CSR constraint
    Cand best{-1.0, -1};
    int ge_idx = 0x7FFFFFFF;
    unsigned c1 = 0, c2 = 0, c4 = 0;
    for (unsigned v = tid; v < in.M; v += NT) {
      double ac = st.ac[v], am = st.am[v];
      int acount = st.acount[v];
      unsigned h = (v * 2654435761u) % MV_HASH;
      for (;;) {
        const int key = s_hkey[h];
        if (key == -1) break;
        if (key == (int)v) {
          const unsigned sl = s_hslot[h];
          if (sl != 0xFFFFu) {
            const unsigned ln = s_slot_lane[sl];
            if (ln != 0xFFu) {
              ac = s_tac[ln];
              am = s_tam[ln];
              acount = s_tacount[ln];
            }
          }
          break;
        }
        h = (h + 1) % MV_HASH;
      }
      const OfferA a = vb.oa[v];
      if (ac + j.c > a.oc || am + j.m > a.om || ((j.flags & JF_XRES) && xres_fail_dev(vb.in_dev, st, jj, v) != 0u)) {
        ++c1;
        continue;
      }
      const OfferB o = vb.ob[v];
      bool ok = static_fast(j, o, in, v) && dyn_fast(j, o, acount);
      if (ok && slow) ok = static_pass(in, jj, v);
      if (ok && grouped) ok = group_pass(in, st, jj, v);
      if (!ok) {
        ++c2;
        continue;
      }
      const double fit = fitness_of(a, ac, am, j.c, j.m);
      if (!(fit > 0.0)) {
        ++c4;
        continue;
      }
      if (fit > best.fit) {  // offers ascend with v inside a thread: the first maximum keeps the lowest index
        best.fit = fit;
        best.idx = (int)v;
      }
      if (fit > good_enough && (int)v < ge_idx) ge_idx = (int)v;
    }
    for (int d = 32; d >= 1; d >>= 1) {
      const Cand o{__shfl_xor(best.fit, d, COOK_WAVE), __shfl_xor(best.idx, d, COOK_WAVE)};
      if (cand_better(o, best)) best = o;
      const int og = __shfl_xor(ge_idx, d, COOK_WAVE);
      ge_idx = og < ge_idx ? og : ge_idx;
      c1 += __shfl_xor(c1, d, COOK_WAVE);
      c2 += __shfl_xor(c2, d, COOK_WAVE);
      c4 += __shfl_xor(c4, d, COOK_WAVE);
    }
    if (lane == 0) {
      const unsigned w = tid >> 6;
      s_rfit[w] = best.fit;
      s_ridx[w] = best.idx;
      s_rge[w] = ge_idx;
      s_rc[w][0] = c1;
      s_rc[w][1] = c2;
      s_rc[w][2] = c4;
    }
  };
  if (tid >= COOK_WAVE) {  // helper waves: sleep at the barrier until wave 0 asks for a re-evaluation or finishes the walk
    if constexpr (!REEVAL) return;
    for (;;) {
      EMU_SITE("resolve: helper waiting");
      __syncthreads();
      const int cmd = s_cmd;
      if (cmd < 0) break;
      reeval((unsigned)cmd);
      __syncthreads();
    }
    return;
  }
  // wave 0 walks the window
  const unsigned long long tk1 = cook_ticks();
  // ---- sequential phase ---------------------------------------------------------------------------------------------------
  // Lanes own the offers touched in this round (state in registers).  Cross-lane traffic is ballots, v_readlane and DPP
  // reductions (no ds_bpermute); fitness values are first compared through a reciprocal-multiply approximation (relative error
  // < 2^-50) and the two fp64 divides are only executed when candidates are closer than 2^-38 relative — exactness is unaffected.
  int t_slot = -1, t_v = -1;
  double t_oc = 0, t_om = 0, t_rc = 0, t_rm = 0, t_invc = 0, t_invm = 0;
  double t_ac = 0, t_am = 0, t_basec = 0, t_basem = 0;
  int t_acount = 0, t_run = 0, t_slack = 0;
  unsigned t_k8s = 0, t_host = 0;
  unsigned long long t_col = 0ull;
  // group members placed in THIS round, one per lane in placement order (group, host, match index): what a later member of the same
  // group has to avoid / link to, without asking HBM.  n_log > 64: the log overflowed, no fast path for group members any more
  unsigned lg_group = 0xFFFFFFFFu, lg_host = 0u, n_log = 0u;
  int lg_k = -1;
  unsigned cur_g = 0xFFFFFFFFu;
  unsigned nT = 0;
  unsigned stop = 0;  // 1 list exhausted, 2 touched set full, 3 group barrier, 4 slot table cut the window
  unsigned matched = 0, head_matched = ctl.head_matched;
  // window position of walk position i (for the one position the records do not hold: the first job beyond the staged ones)
  auto walkpos_to_b = [&](unsigned i) {
    unsigned g = 0;
    while (g + 1 < (unsigned)MV_JGL && s_vbase[g + 1] <= i) ++g;
    unsigned long long m = s_visit[g];
    for (unsigned r = i - s_vbase[g]; r > 0; --r) m &= m - 1ull;
    return g * COOK_WAVE + (unsigned)__ffsll((unsigned long long)m) - 1u;
  };
  unsigned resolved = n_eff < n_walk ? (unsigned)s_job[n_eff].b : (n_eff < n_list ? walkpos_to_b(n_eff) : nwin);
  unsigned nslots_cur = nslots;  // slots staged so far (re-evaluations may add some)
  unsigned n_exhaust = 0;        // jobs whose list ran out and were re-evaluated
  unsigned n_trunc = 0;          // walked jobs with a truncated merged list (statistics)
  bool stop_on_trunc = false;    // the round ended on a TRUNCATED list running out (an untruncated full list cannot: it is complete)
  constexpr double EPS_HI = 1.0 + 0x1p-38, EPS_LO = 1.0 - 0x1p-38;
  struct JobRegs {   // exactly what the LDS loads deliver: nothing is decoded before the job's own iteration (a decode right after
                     // the load would wait for it)
    double c, m;
    unsigned info, group;
    unsigned f4b;      // JobL::f4 | JobL::b << 16
    double e_fit;      // list entry `lane` (lanes >= MV_LM: none)
    int e_off;
    unsigned e_slotw;  // EntL::slot | pad << 16
    unsigned owner;    // lane owning the entry's slot, 0xFF untouched, 0xFE no entry
  };
  // record + list entry of walk position i: addresses depend on i only, so the loads of job i+2 are issued two iterations
  // ahead and nothing waits for them (OPAQUE_V: see common.hpp)
  auto load_rec = [&](unsigned i) {
    JobRegs r;
    unsigned ii = i < n_eff ? i : 0u;
    OPAQUE_V(ii);
    const JobL* jp = &s_job[ii];
    r.c = jp->c;
    r.m = jp->m;
    r.info = jp->info;
    r.group = jp->group;
    r.f4b = *reinterpret_cast<const unsigned*>(&jp->f4);
    r.e_fit = -1.0;
    r.e_off = -1;
    r.e_slotw = 0;
    r.owner = 0xFEu;
    if (lane < (unsigned)MV_LM) {
      const EntL* ep = &s_ent[ii][lane];
      r.e_fit = ep->fit;
      r.e_off = ep->off;
      r.e_slotw = *reinterpret_cast<const unsigned*>(&ep->slot);
    }
    return r;
  };
  // the owner look-up needs the entry's slot: issued one iteration ahead (a commit in between patches it, see below)
  auto load_owner = [&](JobRegs& r) {
    if (lane < (unsigned)MV_LM && r.e_off >= 0) r.owner = s_slot_lane[r.e_slotw & 0xFFFFu];
  };
  JobRegs cur = load_rec(0);
  load_owner(cur);
  JobRegs nxt = load_rec(1);
  WAIT_LDS();  // nothing pending at loop entry either (the loop's own waits sit at the END of its iterations)
  unsigned i = 0;  // walk position; after the loop: the number of walk positions done
  for (; i < n_eff; ++i) {
    EMU_SITE("resolve: walk loop");
#ifdef COOK_WALK_PROF
    const unsigned long long pk0 = __builtin_readcyclecounter();
    unsigned pcat = 0;
#define WALK_END(cat)                                                \
  do {                                                               \
    const unsigned long long pk1_ = __builtin_readcyclecounter();    \
    ctl.prof_cyc[cat] += pk1_ - pk0;                                 \
    ctl.prof_cnt[cat] += 1u;                                         \
  } while (0)
#else
    unsigned pcat = 0;
    (void)pcat;
#define WALK_END(cat) ((void)0)
#endif
    JobRegs nn = load_rec(i + 2);  // in flight while job i is decided
    load_owner(nxt);
    const unsigned cinfo_u = wave_uniform_u32(cur.info), cb_u = wave_uniform_u32(cur.f4b) >> 16;
    const bool cur_no_zero_fit = (wave_uniform_u32(cur.f4b) & 0xFFFFu) == 0u;  // no offer had zero fitness for this job under S
    const unsigned cur_slot = cur.e_slotw & 0xFFFFu;
    const unsigned b = cb_u, k = head + b;
    const unsigned cpos = lw ? i : b, bl = cpos & 63u;  // the job's bit in the staged colbits columns: by window position, or (long window) by walk position
    const double c = cur.c, m = cur.m;
    const bool grouped = (cinfo_u & JL_GROUPED) != 0;
    const bool job_gpu = (cinfo_u & JL_GPU) != 0;
    const bool has_group = (cinfo_u & JL_HASGROUP) != 0;
    const unsigned g = has_group ? wave_uniform_u32(cur.group) : 0xFFFFFFFFu, gtype = (cinfo_u >> 18) & 3u;
    const int nc = (int)(cinfo_u & 0xFFu);
    n_trunc += (cinfo_u & JL_TRUNC) ? 1u : 0u;  // (scalar: cinfo_u is wave-uniform)

    if ((cpos >> 6) != cur_g) {  // next word of the columns: the touched lanes fetch theirs
      cur_g = cpos >> 6;
      if (t_slot >= 0) t_col = s_col[t_slot][cur_g];
      WAIT_LDS();
    }
    const bool t_on = t_slot >= 0;
    // ======== FAST PATH ======================================================================================================
    // self-contained: decision AND commit, then straight on to the next job (its control flow never joins the general path's).
    // Two instantiations: plain jobs, and members of unique / unconstrained groups whose hosts-to-avoid the set-up phase staged
    // (JL_GSLOT) — kept apart so that the group code costs the plain jobs nothing.
    const unsigned gslot = (cinfo_u >> JL_GSLOT_SHIFT) & JL_GSLOT_NONE;
    auto fast_path = [&](auto group_tag) -> bool {
      constexpr bool GROUP = decltype(group_tag)::value;
      const bool res_ok = t_on && !(t_ac + c > t_oc || t_am + m > t_om);
      bool con_ok = ((t_col >> bl) & 1ull) != 0 && t_acount < t_slack;
      if (job_gpu && t_k8s && t_run + t_acount != 0) con_ok = false;
      unsigned long long ghits = 0ull;  // log entries of this job's group
      if constexpr (GROUP) {
        ghits = __ballot(lane < n_log && lg_group == g);
        if (gtype == 1u) {  // unique host placement (constraints.clj:586-598): not where a cotask runs or was placed
          unsigned fhv[MV_FH];
#pragma unroll
          for (int q = 0; q < MV_FH; ++q) fhv[q] = s_gfh[gslot][q];
          bool forb = false;
#pragma unroll
          for (int q = 0; q < MV_FH; ++q) forb = forb | (t_host == fhv[q]);
          for (unsigned long long hm = ghits; hm != 0ull; hm &= hm - 1ull) {
            const unsigned h = (unsigned)wave_read_lane((int)lg_host, __ffsll((unsigned long long)hm) - 1);  // (every lane takes part)
            forb = forb | (t_host == h);
          }
          con_ok = con_ok & !forb;
        }
      }
      // publish a placed group member: the chain in HBM (later rounds' evaluation and the general path read it) and the round's log
      auto publish_member = [&](int w_offer, unsigned w_host) {
        const int prev = ghits != 0ull ? wave_read_lane(lg_k, 63 - __clzll((long long)ghits)) : s_glast[gslot];
        if (lane == 0) {
          st_agent(&st.job_to_offer[k], w_offer);
          st_agent(&st.job_prev[k], prev);
          st_agent(&st.group_last[g], (int)k);
        }
        if (lane == n_log) {
          lg_group = g;
          lg_host = w_host;
          lg_k = (int)k;
        }
        ++n_log;
      };
      const double a1 = (t_basec + c) * t_invc, a2 = (t_basem + m) * t_invm;
      const double fa = (a1 + a2) * 0.5;
      const bool cand = res_ok && con_ok;
      // fp32 image of the approximate fitness: monotone in fa; a candidate whose approximation cannot be trusted for ordering
      // (negative terms, zero, below fp32's normal range) takes +inf, which sends the job to the general path
      const bool sane = a1 >= 0.0 && a2 >= 0.0 && fa > 0x1p-100;
      const float kf = cand ? (sane ? (float)fa : __int_as_float(0x7F800000)) : 0.0f;
      const float mx = wave_max_f32(kf);
      // first untouched entry of the list: the best untouched offer under S (a touched entry that is still feasible and sits in
      // front of it only gained fitness: it beats this one in the comparison below, so "first untouched" is all the list has to give)
      const unsigned long long untouched_mask = __ballot(cur.owner == 0xFFu);
      double u_fit = -1.0;
      int u_off = -1, u_slot = -1;
      if (untouched_mask != 0ull) {
        const int qs = __ffsll((unsigned long long)untouched_mask) - 1;
        u_fit = wave_read_lane_f64(cur.e_fit, qs);
        u_off = wave_read_lane(cur.e_off, qs);
        u_slot = wave_read_lane((int)cur_slot, qs);
      }
      int f_lane = -1;      // >= 0: that touched offer wins
      bool f_new = false;   // the untouched offer (u_off, u_slot) wins
      if (mx == 0.0f) {  // no touched offer can take the job
        f_new = u_off >= 0;  // else: unmatched or list exhausted -> general path
      } else if (mx < __int_as_float(0x7F800000)) {
        const unsigned long long near = __ballot(kf >= mx * (1.0f - 0x1p-20f));
        if ((near & (near - 1ull)) == 0ull) {  // one touched offer clearly ahead of the other touched ones
          const int wl = __ffsll((unsigned long long)near) - 1;
          const double fw = wave_read_lane_f64(fa, wl);
          if (u_off < 0) {
            // no untouched entry: fine unless the list is full and none of its entries is still a candidate (then better
            // untouched offers may exist beyond the list: exhausted, general path)
            bool ok = COOK_L_COMPLETE();
            if (!ok) {
              const unsigned long long cand_mask = __ballot(cand);
              const bool e_live = cur.owner < 0xFEu && ((cand_mask >> (cur.owner & 63u)) & 1ull);
              ok = __any(e_live);
            }
            if (ok) f_lane = wl;
          } else if (fw * EPS_LO > u_fit) {
            f_lane = wl;
          } else if (fw * EPS_HI < u_fit) {
            f_new = true;
          }
        }
      }
      if (f_lane >= 0) {  // an offer touched earlier in this round takes the job
        if ((int)lane == f_lane) {
          t_ac += c;
          t_am += m;
          t_acount += 1;
          t_basec = t_rc + t_ac;
          t_basem = t_rm + t_am;
        }
        const int w = wave_read_lane(t_v, f_lane);
        ++matched;
        if (k == 0) head_matched = 1;
        if (lane == 0) s_j2o[i] = w;  // (s_fail[i] = 0 since the set-up)
        if constexpr (GROUP) publish_member(w, (unsigned)wave_read_lane((int)t_host, f_lane));
        WALK_STAT(3, 1);
        WALK_STAT(8, 1);
        WALK_END(GROUP ? 4u : 1u);
        return true;
      }
      if (f_new && nT < (unsigned)MV_T) {  // an untouched offer: the next free lane takes ownership
        if (lane == nT) {
          const SlotRec r = s_slot[u_slot];
          t_slot = u_slot;
          t_v = u_off;
          t_oc = r.a.oc;
          t_om = r.a.om;
          t_rc = r.a.rc;
          t_rm = r.a.rm;
          t_invc = r.a.inv_dc;
          t_invm = r.a.inv_dm;
          t_k8s = r.o.flags & 1u;
          t_host = r.o.host;
          t_run = r.o.run_count;
          t_slack = r.o.task_slack;
          t_ac = r.ac + c;
          t_am = r.am + m;
          t_acount = r.acount + 1;
          t_basec = t_rc + t_ac;
          t_basem = t_rm + t_am;
          s_slot_lane[u_slot] = (unsigned char)nT;
        }
        if (lane == nT) t_col = s_col[u_slot][cur_g];
        WAIT_LDS();
        if constexpr (GROUP) publish_member(u_off, (unsigned)wave_read_lane((int)t_host, (int)nT));
        // the owner look-up of the next job was issued before this commit: patch it
        if (nxt.owner == 0xFFu && (nxt.e_slotw & 0xFFFFu) == (unsigned)u_slot) nxt.owner = nT;
        ++nT;
        ++matched;
        if (k == 0) head_matched = 1;
        if (lane == 0) s_j2o[i] = u_off;
        wave_sync();  // the owner table update is visible to the whole wave before the next look-up reads it
        WALK_STAT(4, 1);
        WALK_STAT(8, 1);
        WALK_END(GROUP ? 4u : 2u);
        return true;
      }
      return false;
    };
    if (!use_ge) {
      bool fast_done = false;
      if (!(cinfo_u & (JL_GROUPED | JL_HASGROUP | JL_XRES)))
        fast_done = fast_path(std::false_type{});
      else if (gslot != JL_GSLOT_NONE && n_log < (unsigned)COOK_WAVE)
        fast_done = fast_path(std::true_type{});
      if (fast_done) {
        WAIT_LDS_BUT_LAST();  // the prefetches of this iteration have arrived (see common.hpp); the result store may still fly
        cur = nxt;
        nxt = nn;
        continue;
      }
    }
    int win = -1, win_slot = -1, win_lane = -1;  // win_lane >= 0: a touched offer wins
    bool need_exact = false;
    bool exhausted = false;  // the job's list ran out: re-evaluate it against the current state (below)
    unsigned pe_bits = 8u;   // exact verdict of this lane's offer (only when the exact path ran)
    double pe_fit = 0.0;
    unsigned jj = 0;
    // values of the general path that the unmatched branch of the commit reads
    bool res_ok_g = false, con_ok_g = false;
    double nc_g = 0.0, nm_g = 0.0;
    // ======== GENERAL PATH ===================================================================================================
    {
      bool gok = true;
      if (grouped) {
        jj = j_index ? j_index[k] : k;
        // a second member of a balanced / attribute-equals group after one was placed in this round: re-snapshot first
        if (gtype >= 2 && ld_agent(&st.group_last[g]) >= (int)head) {
          stop = 3;
          resolved = b;
          break;
        }
        if (t_slot >= 0) gok = group_pass_dev(vb.in_dev, st, jj, (unsigned)t_v);
      }
      // every touched offer re-evaluated under the current state: verdict + approximate fitness
      bool res_ok = t_on && !(t_ac + c > t_oc || t_am + m > t_om);
      if (cinfo_u & JL_XRES) {  // ports / named scalars: the counters of the call live in HBM (only such jobs move them)
        jj = j_index ? j_index[k] : k;
        if (res_ok) res_ok = xres_fail_dev(vb.in_dev, st, jj, (unsigned)t_v) == 0u;
      }
      bool con_ok = ((t_col >> bl) & 1ull) != 0 && t_acount < t_slack && gok;
      if (job_gpu && t_k8s && t_run + t_acount != 0) con_ok = false;
      const double nc_ = t_basec + c, nm_ = t_basem + m;  // (rc + ac) + c, (rm + am) + m
      const double a1 = nc_ * t_invc, a2 = nm_ * t_invm;
      const double fa = (a1 + a2) * 0.5;
      const bool cand = res_ok && con_ok;
      res_ok_g = res_ok, con_ok_g = con_ok, nc_g = nc_, nm_g = nm_;
      // the approximation is trusted for ordering only when both terms are non-negative and the result is positive
      const bool sane = a1 >= 0.0 && a2 >= 0.0 && fa > 0.0;
      need_exact = use_ge || __any(cand && !sane);
      const unsigned long long cand_mask = __ballot(cand);
      double u_fit = -1.0;     // best untouched candidate: fitness under S, offer, slot
      int u_off = -1, u_slot = -1;
      bool decided = false;
      do {
        // No feasible offer under S, no zero-fitness offer, no constrained group: placements only take capacity away and the
        // job's constraints can only get worse on a touched offer, so it stays unmatched whatever happened in this round;
        // only its failure summary may change (handled below from the touched offers' current verdicts).
        WALK_STAT(0, 1);
        WALK_STAT(6, nT);
        if (nc == 0 && !grouped && cur_no_zero_fit) {
          WALK_STAT(1, 1);
          break;
        }
        // --- arg-max path: first list entry that is untouched, or touched and still a candidate -------------------------------
        // (a touched offer that is still feasible only gained fitness, so it dominates every untouched offer behind it; a
        //  zero-fitness verdict cannot appear on an offer that was feasible under S)
        const bool e_valid = cur.owner != 0xFEu;
        const bool e_untouched = cur.owner == 0xFFu;
        const bool e_live = e_valid && !e_untouched && ((cand_mask >> (cur.owner & 63u)) & 1ull);
        const unsigned long long settle_mask = __ballot(e_untouched || e_live), untouched_mask = __ballot(e_untouched);
        if (settle_mask == 0ull && COOK_L_TRUNC()) {
          exhausted = true;
          break;
        }
        if (settle_mask != 0ull) {
          const int qs = __ffsll((unsigned long long)settle_mask) - 1;
          if ((untouched_mask >> qs) & 1ull) {
            u_fit = wave_read_lane_f64(cur.e_fit, qs);
            u_off = wave_read_lane(cur.e_off, qs);
            u_slot = wave_read_lane((int)cur_slot, qs);
          }
        }
        // --- best touched candidate ----------------------------------------------------------------------------------------------
        if (!need_exact) {
          if (cand_mask == 0ull) {
            win = u_off;
            win_slot = u_slot;
            decided = true;
          } else {
            const unsigned long long key = cand ? (unsigned long long)__double_as_longlong(fa) : 0ull;  // positive doubles
            const double mx = __longlong_as_double((long long)wave_max_u64(key));
            const unsigned long long near = __ballot(cand && fa >= mx * EPS_LO);
            if ((near & (near - 1ull)) == 0ull) {  // one touched offer clearly ahead of the other touched ones
              if (u_off < 0 || mx * EPS_LO > u_fit) {
                win_lane = __ffsll((unsigned long long)near) - 1;
                decided = true;
              } else if (mx * EPS_HI < u_fit) {
                win = u_off;
                win_slot = u_slot;
                decided = true;
              }
            }
            if (!decided) need_exact = true;
          }
        }
        if (need_exact) {
          WALK_STAT(2, 1);
          if (t_on) {
            pe_bits = 0u;
            if (!res_ok) {
              pe_bits = 1u;
            } else if (!con_ok) {
              pe_bits = 2u;
            } else {
              pe_fit = (nc_ / (t_oc + t_rc) + nm_ / (t_om + t_rm)) / 2.0;
              if (!(pe_fit > 0.0)) pe_bits = 4u;
            }
          }
          const bool t_feas = t_on && pe_bits == 0u;
          const unsigned long long feas_mask = __ballot(t_feas);
          // with exact verdicts a list entry settles only if its owner is still FEASIBLE (zero fitness excluded)
          const bool e_live2 = e_valid && !e_untouched && ((feas_mask >> (cur.owner & 63u)) & 1ull);
          const unsigned long long settle2 = __ballot(e_untouched || e_live2);
          if (settle2 == 0ull && COOK_L_TRUNC()) {
            exhausted = true;
            break;
          }
          u_fit = -1.0;
          u_off = u_slot = -1;
          if (settle2 != 0ull) {
            const int qs = __ffsll((unsigned long long)settle2) - 1;
            if ((untouched_mask >> qs) & 1ull) {
              u_fit = wave_read_lane_f64(cur.e_fit, qs);
              u_off = wave_read_lane(cur.e_off, qs);
              u_slot = wave_read_lane((int)cur_slot, qs);
            }
          }
          // good-enough path: lowest offer index with fitness > good-enough (scheduler.clj:2312-2314)
          int ge_pick = 0x7FFFFFFF, ge_slot = -1, ge_lane = -1;
          if (use_ge) {
            const int ng = (int)((cinfo_u >> 8) & 0xFFu);
            GEntL ge;
            ge.off = -1;
            ge.slot = 0;
            ge.pad = 0;
            unsigned g_owner = 0xFEu;
            if ((int)lane < ng) {
              ge = s_gent[i][lane];
              g_owner = s_slot_lane[ge.slot];
            }
            const unsigned long long gun = __ballot(g_owner == 0xFFu);
            int last_idx = -1;
            if (ng > 0) last_idx = wave_read_lane(ge.off, ng - 1);
            if (gun != 0ull) {
              const int q = __ffsll((unsigned long long)gun) - 1;
              ge_pick = wave_read_lane(ge.off, q);
              ge_slot = wave_read_lane((int)ge.slot, q);
            }
            // lowest-index touched offer that is feasible with fitness > good-enough
            const unsigned long long tkey = (t_feas && pe_fit > good_enough)
                                                ? (((unsigned long long)(unsigned)(0x7FFFFFFF - t_v) << 32) | (unsigned long long)lane)
                                                : 0ull;
            const unsigned long long tmx = feas_mask != 0ull ? wave_max_u64(tkey) : 0ull;
            const int tg = tmx != 0ull ? 0x7FFFFFFF - (int)(unsigned)(tmx >> 32) : 0x7FFFFFFF;
            if (gun == 0ull && ng == MV_LG && tg > last_idx) {
              // untouched good-enough offers beyond the list may exist with an index below the best touched one
              exhausted = true;
              break;
            }
            if (tg < ge_pick) {
              ge_pick = tg;
              ge_lane = (int)(unsigned)(tmx & 63ull);
            }
          }
          if (ge_pick != 0x7FFFFFFF) {
            if (ge_lane >= 0) {
              win_lane = ge_lane;
            } else {
              win = ge_pick;
              win_slot = ge_slot;
            }
          } else {
            // best touched (max fitness, lowest offer index on ties) vs best untouched
            Cand best{-1.0, -1};
            int best_lane = -1;
            if (feas_mask != 0ull) {
              const unsigned long long key = t_feas ? (unsigned long long)__double_as_longlong(pe_fit) : 0ull;
              const unsigned long long mx = wave_max_u64(key);
              unsigned long long tie = __ballot(t_feas && key == mx);
              int wl = __ffsll((unsigned long long)tie) - 1;
              int wv = wave_read_lane(t_v, wl);
              tie &= tie - 1ull;
              while (tie != 0ull) {  // equal fitness on several touched offers: the lowest offer index wins
                const int l2 = __ffsll((unsigned long long)tie) - 1;
                const int v2 = wave_read_lane(t_v, l2);
                if (v2 < wv) {
                  wv = v2;
                  wl = l2;
                }
                tie &= tie - 1ull;
              }
              best = Cand{__longlong_as_double((long long)mx), wv};
              best_lane = wl;
            }
            if (u_off >= 0 && cand_better(Cand{u_fit, u_off}, best)) {
              win = u_off;
              win_slot = u_slot;
            } else if (best_lane >= 0) {
              win_lane = best_lane;
            }
          }
        }
      } while (0);
    }
    // --- list exhausted: the whole workgroup evaluates this one job against the current state ---------------------------------
    int re_bits = -1;  // >= 0: the exact failure summary of an unmatched re-evaluated job
    if (exhausted) {
      if (!REEVAL || n_exhaust >= ctl.reeval_max) {  // end the round here: the next round evaluates the rest of the window afresh
        stop = 1;
        resolved = b;
        stop_on_trunc = (cinfo_u & JL_TRUNC) != 0u;
        break;
      }
      if constexpr (REEVAL) {
      if (t_slot >= 0) {
        s_tac[lane] = t_ac;
        s_tam[lane] = t_am;
        s_tacount[lane] = t_acount;
      }
      if (lane == 0) s_cmd = (int)b;
      EMU_SITE("resolve: walker asks for a re-evaluation");
      __syncthreads();
      reeval(b);
      EMU_SITE("resolve: walker after re-evaluation");
      __syncthreads();
      ++n_exhaust;
      Cand rb{s_rfit[0], s_ridx[0]};
      int rg = s_rge[0];
      unsigned rc1 = s_rc[0][0], rc2 = s_rc[0][1], rc4 = s_rc[0][2];
      for (int q = 1; q < (int)(NT / COOK_WAVE); ++q) {
        const Cand o{s_rfit[q], s_ridx[q]};
        if (cand_better(o, rb)) rb = o;
        rg = s_rge[q] < rg ? s_rge[q] : rg;
        rc1 += s_rc[q][0];
        rc2 += s_rc[q][1];
        rc4 += s_rc[q][2];
      }
      win = win_slot = win_lane = -1;
      const int pick = rg != 0x7FFFFFFF ? rg : rb.idx;  // scheduler.clj:2312-2314: the first good-enough offer wins outright
      if (pick < 0) {
        re_bits = (int)((rc1 ? 1u : 0u) | (rc2 ? 2u : 0u) | (rc4 ? 4u : 0u));
      } else {
        // the winner may be touched, staged but untouched, or not staged at all (then it gets a slot now)
        unsigned h = ((unsigned)pick * 2654435761u) % MV_HASH;
        int slot = -1;
        for (;;) {
          const int key = s_hkey[h];
          if (key == -1 || key == pick) {
            if (key == pick && s_hslot[h] != 0xFFFFu) slot = (int)s_hslot[h];
            break;
          }
          h = (h + 1) % MV_HASH;
        }
        const int owner_lane = slot >= 0 ? (int)s_slot_lane[slot] : 0xFF;
        wave_sync();  // every lane has looked the offer up before lane 0 edits the tables
        if (owner_lane != 0xFF) {
          win_lane = owner_lane;
        } else {
          if (slot < 0) {
            if (nslots_cur >= (unsigned)MV_S || nslots_cur + 1u >= (unsigned)MV_HASH) {
              stop = 4;  // no room to stage another offer: end the round before this job
              resolved = b;
              break;
            }
            slot = (int)nslots_cur++;
            if (lane == 0) {
              SlotRec r;
              r.a = vb.oa[pick];
              r.o = vb.ob[pick];
              r.ac = st.ac[pick];
              r.am = st.am[pick];
              r.acount = st.acount[pick];
              r.offer = pick;
              s_slot[slot] = r;
              s_slot_lane[slot] = 0xFF;
              s_hkey[h] = pick;
              s_hslot[h] = (unsigned short)slot;
            }
            if (lane < (unsigned)MV_JG)
              s_col[slot][lane] = (!lw && lane * COOK_WAVE < nwin) ? vb.colbits[(size_t)pick * MV_JGL + lane] : 0ull;
            wave_sync();
            for (unsigned g = lane; lw && g < ngrp; g += COOK_WAVE) {  // (long window) the new slot's column, compacted like the others
              const unsigned long long V = s_visit[g];
              const unsigned base = s_vbase[g];
              if (V == 0ull || base >= n_walk) continue;
              const unsigned long long Wd = vb.colbits[(size_t)pick * MV_JGL + g];
              unsigned long long packed = 0ull;
              unsigned o = 0;
              for (unsigned long long m = V; m != 0ull; m &= m - 1ull, ++o) packed |= ((Wd >> (__ffsll((unsigned long long)m) - 1)) & 1ull) << o;
              const unsigned w0 = base >> 6, sh = base & 63u;
              atomicOr(&s_col[slot][w0], packed << sh);
              if (sh != 0u && w0 + 1u < (unsigned)MV_JG) atomicOr(&s_col[slot][w0 + 1u], packed >> (64u - sh));
            }
            wave_sync();
          }
          win = pick;
          win_slot = slot;
        }
      }
      }  // if constexpr (REEVAL)
    }
    // --- commit --------------------------------------------------------------------------------------------------------------
    if (win_lane >= 0) WALK_STAT(3, 1);
    else if (win >= 0) WALK_STAT(4, 1);
    else WALK_STAT(5, 1);
    WALK_STAT_PREV_LANE(i, win_lane, win, nT);
#ifdef COOK_WALK_PROF
    pcat = grouped ? 4u : (win >= 0 || win_lane >= 0 ? 5u : 3u);
#endif
    if (win_lane >= 0) {  // an offer touched earlier in this round takes the job
      if ((int)lane == win_lane) {
        t_ac += c;
        t_am += m;
        t_acount += 1;
        t_basec = t_rc + t_ac;
        t_basem = t_rm + t_am;
      }
      win = wave_read_lane(t_v, win_lane);
    } else if (win >= 0) {  // an untouched offer: the next free lane takes ownership
      if (nT == (unsigned)MV_T) {
        stop = 2;  // no free lane to track a new touched offer: end the round before this job
        resolved = b;
        break;
      }
      if (lane == nT) {
        const SlotRec r = s_slot[win_slot];
        t_slot = win_slot;
        t_v = win;
        t_oc = r.a.oc;
        t_om = r.a.om;
        t_rc = r.a.rc;
        t_rm = r.a.rm;
        t_invc = r.a.inv_dc;
        t_invm = r.a.inv_dm;
        t_k8s = r.o.flags & 1u;
        t_host = r.o.host;
        t_run = r.o.run_count;
        t_slack = r.o.task_slack;
        t_ac = r.ac + c;
        t_am = r.am + m;
        t_acount = r.acount + 1;
        t_basec = t_rc + t_ac;
        t_basem = t_rm + t_am;
        s_slot_lane[win_slot] = (unsigned char)nT;
      }
      if (lane == nT) t_col = s_col[win_slot][cur_g];
      WAIT_LDS();
      // the owner look-up of the next job was issued before this commit: patch it
      if (nxt.owner == 0xFFu && (nxt.e_slotw & 0xFFFFu) == (unsigned)win_slot) nxt.owner = nT;
      ++nT;
      wave_sync();  // the owner table update is visible to the whole wave before the next look-up reads it
    }
    if (win >= 0) {
      ++matched;
      if (k == 0) head_matched = 1;
      if (cinfo_u & JL_XRES) {  // the offer's owner lane books the job's ports / named scalars
        const int ol = win_lane >= 0 ? win_lane : (int)nT - 1;
        if ((int)lane == ol) {
          const MatchIn& in = *vb.in_dev;
          if (!L.x0set[lane]) {
            L.x0set[lane] = 1;
            L.x0p[lane] = ld_agent(&st.xports[win]);
            _Pragma("unroll") for (unsigned sc = 0; sc < 3u; ++sc)
              if (sc < in.n_scal) L.x0s[lane][sc] = ld_agent(&st.xscal[(size_t)sc * in.M + (unsigned)win]);
          }
          xres_commit(in, st, jj, (unsigned)win);
        }
      }
      if (lane == 0) {
        s_j2o[i] = win;
        s_fail[i] = 0;
        if (g != 0xFFFFFFFFu) {  // cotasks look each other up through HBM (group_pass): publish at once
          st_agent(&st.job_to_offer[k], win);
          st_agent(&st.job_prev[k], ld_agent(&st.group_last[g]));
          st_agent(&st.group_last[g], (int)k);
        }
      }
      if (g != 0xFFFFFFFFu) {
        wave_sync();  // later cotasks of this wave read what lane 0 just published
        // ... and the round's log, for the members that take the fast path (the owner lane of the winning offer knows its host)
        const int ol = win_lane >= 0 ? win_lane : (int)nT - 1;
        const unsigned w_host = (unsigned)wave_read_lane((int)t_host, ol);
        if (n_log < (unsigned)COOK_WAVE) {
          if (lane == n_log) {
            lg_group = g;
            lg_host = w_host;
            lg_k = (int)k;
          }
          ++n_log;
        } else {
          n_log = COOK_WAVE + 1u;  // overflow: the log is incomplete from here on
        }
      }
    } else {
      // unmatched: failure summary = OR over offers of the first failing check under the CURRENT state.  Start from the
      // snapshot counts and swap each touched offer's snapshot verdict for its current one (exact verdicts needed).
      // (only the general path gets here: the fast path never leaves a job unmatched)
      const JobL jl = s_job[i];
      int d1 = 0, d2 = 0, d4 = 0;
      if (nT != 0) {  // wave-uniform
        if (pe_bits == 8u && t_on) {  // the exact path did not run for this job
          pe_bits = 0u;
          if (!res_ok_g) {
            pe_bits = 1u;
          } else if (!con_ok_g) {
            pe_bits = 2u;
          } else {
            pe_fit = (nc_g / (t_oc + t_rc) + nm_g / (t_om + t_rm)) / 2.0;
            if (!(pe_fit > 0.0)) pe_bits = 4u;
          }
        }
        unsigned p0 = 0u;  // snapshot verdict: state at round start, group placements of this round ignored via the cutoff
        if (t_on) {
          const SlotRec r = s_slot[t_slot];
          bool x0_fail = false;
          if (cinfo_u & JL_XRES) {  // ports / named scalars as the round began: saved if a job of this round moved them, else current
            const MatchIn& in = *vb.in_dev;
            const bool sv = L.x0set[lane] != 0;
            const int jp = in.j_ports ? in.j_ports[jj] : 0;
            const long long up = sv ? L.x0p[lane] : ld_agent(&st.xports[t_v]);
            if (jp > 0 && up + jp > (long long)(in.o_ports ? in.o_ports[t_v] : 0)) x0_fail = true;
            _Pragma("unroll") for (unsigned sc = 0; sc < 3u; ++sc) {
              if (sc >= in.n_scal) break;
              const double rq = in.j_scal[sc][jj];
              const double us = sv ? L.x0s[lane][sc] : ld_agent(&st.xscal[(size_t)sc * in.M + (unsigned)t_v]);
              if (rq == rq && us + rq > (in.o_scal[sc] ? in.o_scal[sc][t_v] : 0.0)) x0_fail = true;
            }
          }
          if (r.ac + c > t_oc || r.am + m > t_om || x0_fail) {
            p0 = 1u;
          } else {
            bool ok = ((t_col >> bl) & 1ull) != 0 && r.acount < t_slack;
            if (job_gpu && t_k8s && t_run + r.acount != 0) ok = false;
            if (ok && grouped) {
              MatchState st0 = st;
              st0.cutoff = (int)head;
              ok = group_pass_dev(vb.in_dev, st0, jj, (unsigned)t_v);
            }
            if (!ok) {
              p0 = 2u;
            } else {
              const double f0 = ((t_rc + r.ac + c) / (t_oc + t_rc) + (t_rm + r.am + m) / (t_om + t_rm)) / 2.0;
              if (!(f0 > 0.0)) p0 = 4u;
            }
          }
        }
        d1 = __popcll(__ballot(t_on && (pe_bits & 1u))) - __popcll(__ballot(t_on && (p0 & 1u)));
        d2 = __popcll(__ballot(t_on && (pe_bits & 2u))) - __popcll(__ballot(t_on && (p0 & 2u)));
        d4 = __popcll(__ballot(t_on && (pe_bits & 4u))) - __popcll(__ballot(t_on && (p0 & 4u)));
      }
      unsigned bits = (((int)jl.f1 + d1) > 0 ? 1u : 0u) | (((int)jl.f2 + d2) > 0 ? 2u : 0u) | (((int)jl.f4 + d4) > 0 ? 4u : 0u);
      if (re_bits >= 0) bits = (unsigned)re_bits;  // exact counts from the re-evaluation
      if (lane == 0) {
        s_j2o[i] = -1;
        s_fail[i] = (unsigned char)(bits ? bits : 8u);
      }
    }
    WALK_END(pcat);
    WAIT_ALL_MEM();
    cur = nxt;
    nxt = nn;
  }
  if (stop == 0 && n_eff < n_list) stop = 4;
  if constexpr (REEVAL) {
    if (lane == 0) s_cmd = -1;  // release the helper waves
    EMU_SITE("resolve: walker done");
    __syncthreads();
  }
  // flush the results of the jobs resolved, write the touched offers' state back and publish the new head
  wave_sync();
  for (unsigned x = lane; x < i; x += COOK_WAVE) {  // the walked jobs (the others were settled, and written, in the set-up phase)
    const unsigned bx = s_job[x].b;
    st.job_to_offer[head + bx] = s_j2o[x];
    if (st.fail_code) st.fail_code[head + bx] = s_fail[x];
  }
  if (t_slot >= 0) {
    st.ac[t_v] = t_ac;
    st.am[t_v] = t_am;
    st.acount[t_v] = t_acount;
    if (t_ac + st.jmin[0] > t_oc || t_am + st.jmin[1] > t_om)  // full for every job of this call, for good
      atomicAnd(&st.alive[(unsigned)t_v >> 6], ~(1ull << ((unsigned)t_v & 63u)));
  }
  if (lane == 0) {
    ctl.head = head + resolved;
    ctl.rounds += 1;
    ctl.matched += matched;
    ctl.head_matched = head_matched;
    ctl.touched_sum += nT;
    ctl.visited_sum += n_list;
    ctl.t_setup += tk1 - tk0;
    ctl.t_seq += cook_ticks() - tk1;
    if (stop == 1) ctl.stop_list += 1;
    ctl.reevals += n_exhaust;
    ctl.trunc_lists += n_trunc;
    ctl.trunc_stops += stop_on_trunc ? 1u : 0u;
    if (vb.round_log && ctl.rounds <= MV_ROUND_LOG_CAP) {
      RoundLog r;
      r.head = head, r.wcur = ctl.wcur, r.resolved = resolved, r.n_list = n_list, r.touched = nT, r.stop = stop, r.matched = matched;
      r.setup_ticks = (unsigned)(tk1 - tk0), r.seq_ticks = (unsigned)(cook_ticks() - tk1), r.nslots = nslots_cur, r.pad0 = r.pad1 = 0;
      vb.round_log[ctl.rounds - 1] = r;
    }
    if (stop == 2) ctl.stop_full += 1;
    if (stop == 3) ctl.stop_group += 1;
    if (stop == 4) ctl.stop_slots += 1;
    if (stop == 0) ctl.stop_window += 1;
    // adapt the window: a multiple of what a round resolves (more = fewer rounds, less = fewer jobs evaluated twice)
    unsigned wn = stop == 0 ? ctl.wcur * 2 : (unsigned)(((unsigned long long)resolved * ctl.wgrow_pct + 99ull) / 100ull);
    if (wn < 64) wn = 64;
    // past MV_WMAX only while next to nothing of a window has to be walked (see MV_WLONG), and never beyond what this launch
    // sequence sized its buffers and grids for
    unsigned cap = (unsigned)MV_WMAX;
    if (stop == 0 && nwin >= (unsigned)MV_WMAX && n_list * 8u <= nwin) cap = ctl.wlong_cap > cap ? ctl.wlong_cap : cap;
    if (wn > cap) wn = cap;
    ctl.wcur = wn;
    *vb.ctl = ctl;
  }
}

__global__ void __launch_bounds__(MV_RTHREADS) match_resolve2(MatchState st, V2Buf vb) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(ResolveLds)];
  resolve_round<false>(lds, st, vb);
}
__global__ void __launch_bounds__(MV_RTHREADS) match_resolve2_reeval(MatchState st, V2Buf vb) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(ResolveLds)];
  resolve_round<true>(lds, st, vb);
}

// ---- persistent placement kernel ---------------------------------------------------------------------------------------------
// The three phases of a round above, looped inside ONE launch: G workgroups share the eval tiles and the merge jobs of a
// round, workgroup 0 resolves it, and grid-wide barriers (a counter + a generation word in HBM, agent-scope atomics)
// separate the phases.  No host round trip and no launch per round: a match call is one kernel however many rounds it
// takes.  All G workgroups must be resident at the same time (the host sizes G for that); should a barrier ever time out
// the kernel gives up (PersistCtl::error) and the host re-runs the match with one launch per phase.
struct PersistCtl {
  unsigned bar_count, bar_gen;
  unsigned error;   // 1: a grid barrier timed out (not all workgroups were resident)
  unsigned rounds;  // rounds executed by this launch
};
constexpr unsigned long long MV_BARRIER_TIMEOUT_TICKS = 200000000ull;  // 2 s of the 100 MHz clock

union PersistLds {
  EvalLds e;
  ResolveLds r;
};

// returns false on time-out / error (every workgroup then leaves the kernel)
static __device__ __forceinline__ bool grid_barrier(PersistCtl* pc, unsigned nblocks) {
  __threadfence();  // every wave: its own stores of the phase have reached L2 before the workgroup arrives
  __syncthreads();
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    int ok = 1;
    if (nblocks > 1) {
      __threadfence();  // release: this workgroup's writes of the phase are visible device-wide before it arrives
      const unsigned gen = ld_agent(&pc->bar_gen);
      if (atomicAdd(&pc->bar_count, 1u) == nblocks - 1u) {
        st_agent(&pc->bar_count, 0u);
        __threadfence();
        atomicAdd(&pc->bar_gen, 1u);
      } else {
        const unsigned long long t0 = cook_ticks();
        while (ld_agent(&pc->bar_gen) == gen) {
          if (ld_agent(&pc->error) != 0u) {
            ok = 0;
            break;
          }
          SPIN_PAUSE_LONG();
          if (cook_ticks() - t0 > MV_BARRIER_TIMEOUT_TICKS) {
            st_agent(&pc->error, 1u);
            ok = 0;
            break;
          }
        }
      }
    }
    if (ld_agent(&pc->error) != 0u) ok = 0;
    s_ok = ok;
  }
  __syncthreads();
  __threadfence();  // acquire: drop stale L1 lines before reading what the other workgroups wrote
  return s_ok != 0;
}

__global__ void __launch_bounds__(COOK_WAVE* MV_EW) match_persist(MatchIn in, MatchState st, V2Buf vb, PersistCtl* pc, unsigned max_rounds) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(PersistLds)];
  const unsigned nb = gridDim.x, wg = blockIdx.x;
  unsigned rounds = 0;
  for (; rounds < max_rounds; ++rounds) {
    const unsigned head = ld_agent(&vb.ctl->head), wcur = ld_agent(&vb.ctl->wcur);
    if (head >= in.K) break;  // the same value in every workgroup: written before the last barrier
    const unsigned long long t0 = cook_ticks();
    const unsigned nwin = (head + wcur < in.K) ? wcur : in.K - head;
    const unsigned njg = (nwin + COOK_WAVE - 1) / COOK_WAVE;
    const unsigned ntiles = vb.C * njg;
    for (unsigned t = wg; t < ntiles; t += nb) {
      eval_tile(lds, in, st, vb, head, wcur, t % vb.C, t / vb.C);
      __syncthreads();
    }
    if (!grid_barrier(pc, nb)) return;
    const unsigned long long t1 = cook_ticks();
    for (unsigned b = wg * MV_EW + wave_id(); b < nwin; b += nb * MV_EW) merge_job<false>(in, vb, head, wcur, b);
    if (!grid_barrier(pc, nb)) return;
    if (wg == 0) {
      const unsigned long long t2 = cook_ticks();
      resolve_round<false>(lds, st, vb);
      __syncthreads();
      if (threadIdx.x == 0) {
        vb.ctl->t_eval += t1 - t0;
        vb.ctl->t_merge += t2 - t1;
      }
    }
    if (!grid_barrier(pc, nb)) return;
  }
  if (wg == 0 && threadIdx.x == 0) pc->rounds = rounds;
}


// ---- several pools in lockstep: the same three phases with blockIdx.z = pool ------------------------------------------------
// A rank that owns several pools runs their placements as ONE sequence of launches (the pools' rounds advance together, each on
// its own WinCtl; a pool that has finished exits at once).  Eight independent streams of small kernels interfere badly beyond
// four streams on MI355X (kernel averages double, profiles/README.md); one stream of 8-pool launches does not.
struct PoolCtx {
  MatchIn in;
  MatchState st;
  V2Buf vb;
};
template <bool GE>
__global__ void __launch_bounds__(COOK_WAVE* MV_EW) COOK_EVAL_OCCUPANCY match_eval2_multi(const PoolCtx* __restrict__ ctx) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(EvalLds)];
  const PoolCtx& c = ctx[blockIdx.z];
  if (blockIdx.x >= c.vb.C) return;  // pools may differ in their number of offers
  eval_block<GE>(lds, c.in, c.st, c.vb, c.vb.ctl->head, c.vb.ctl->wcur, blockIdx.x, blockIdx.y, gridDim.y);
}
__global__ void __launch_bounds__(COOK_WAVE* MV_MW) match_merge2_multi(const PoolCtx* __restrict__ ctx) {
  const PoolCtx& c = ctx[blockIdx.z];
  const unsigned head = c.vb.ctl->head, wcur = c.vb.ctl->wcur;
  const unsigned split = wcur <= (unsigned)MV_WMAX ? eval_split(wcur, c.vb.split_max) : 1u;
  for (unsigned b = blockIdx.x * MV_MW + wave_id(); b < wcur; b += gridDim.x * MV_MW) merge_job<false>(c.in, c.vb, head, wcur, b, split);
}
__global__ void __launch_bounds__(MV_RTHREADS) match_resolve2_multi(const PoolCtx* __restrict__ ctx) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(ResolveLds)];
  const PoolCtx& c = ctx[blockIdx.z];
  resolve_round<false>(lds, c.st, c.vb);
}
__global__ void __launch_bounds__(MV_RTHREADS) match_resolve2_multi_reeval(const PoolCtx* __restrict__ ctx) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(ResolveLds)];
  const PoolCtx& c = ctx[blockIdx.z];
  resolve_round<true>(lds, c.st, c.vb);
}

// match_v2.hpp — exact rank-ordered placement (Fenzo scheduleOnce semantics, scheduler.clj:617-687) as a pipeline of
// window rounds.  Placement is sequential by definition — job i+1 sees job i's commitment — but one commitment changes
// ONE offer.  For a window of W consecutive jobs a round is three launches:
//
//   match_eval2    (grid = offer chunks x job groups; lane = job, offers walked in a wave-uniform loop over records staged in the
//                   wave's LDS): against the snapshot S of per-offer assignments at round start, every job gets the top-L
//                   feasible offers of each chunk (fitness desc, index asc), the first offers whose fitness exceeds
//                   good-enough, failure counts, and — for every (job, offer) — one bit "static constraints pass" (a ballot
//                   over the 64 jobs of the wave = one u64 per offer: colbits[offer][group]).  The two fp64 divides of the
//                   fitness are only executed for pairs whose cheap upper bound can still enter the lane's top-L.
//   match_merge2   (one wave per job): chunk lists -> the job's global top-LM / first-LG / failure counts.
//   match_resolve2 (ONE workgroup; wave 0 walks the window in rank order, SEGMENT by segment): the jobs the walk has to visit
//                   and their candidate lists are staged in LDS by walk position, a segment (up to MV_WSEG jobs) at a time;
//                   lanes own the offers committed to in this round ("touched", <= 64, state in registers) and an LDS table
//                   indexed by OFFER tells the owner lane of every offer.  For job j the winner under the current state S' is
//                       max( best UNTOUCHED offer under S , best TOUCHED offer re-evaluated under S' )
//                   and the first entry of j's list that is untouched — or touched and still feasible (its fitness only
//                   grew, so it dominates every untouched offer) — settles the left term.  A lane that opens an offer reads
//                   its record, snapshot state and colbits word from global memory (one round trip per opened offer; round 3
//                   staged every DISTINCT candidate of the window in an LDS slot table: 37 % of the workgroup's LDS, a hash
//                   build per round, and rounds that ended because the table was full).  When a segment is used up with the
//                   lists intact and lanes to spare the workgroup stages the next segment of the SAME evaluated window and the
//                   walker goes on with its lanes — no launch, no re-evaluation.  If a list runs out (truncated: more
//                   candidates may exist) or a 65th offer would be touched, the round ends there and the next re-snapshots.
//
// The result is bit-identical to the one-job-at-a-time sweep (match_serial) for every input; only speed depends on the shapes.
// Jobs of balanced / attribute-equals groups change the feasibility of UNTOUCHED offers when a cotask is placed, so a
// round never resolves a second member of such a group after the first one was placed.
#pragma once
#include <type_traits>

#include "common.hpp"
#include "match_kernels.hpp"

// waves per SIMD the eval kernels are compiled for (-DCOOK_EVAL_WAVES=n builds a tuning variant).  Four since round 5: the block's LDS
// is 27.6 KB (EvalLds), so a fourth wave per SIMD is there for the taking at 128 VGPRs; the compiler spills 39 (best fit) / 62 (good-enough
// launches) of the 158 / 161 registers it would like, and the cycle is still faster — eight pools 57.8 against 59.8 ms in lockstep pairs,
// 52.2 against 52.9 ms with served walkers (profiles/r05q_probe8.txt); 0 = the compiler's choice (three waves)
#ifndef COOK_EVAL_WAVES
#define COOK_EVAL_WAVES 4
#endif
#if COOK_EVAL_WAVES > 0
#define COOK_EVAL_OCCUPANCY COOK_WAVES_PER_SIMD(COOK_EVAL_WAVES)
#else
#define COOK_EVAL_OCCUPANCY
#endif

#ifndef COOK_MV_L
#define COOK_MV_L 8
#endif
constexpr int MV_L = COOK_MV_L;            // candidate list length per job and chunk (-DCOOK_MV_L=n builds a variant for tuning runs)
// Two LIST SHAPES of the merged lists (what the walk sees), chosen by the launch's template flag GE:
//   best fit (good-enough-fitness >= 1, the parity setting): 24 best-fit entries, no good-enough list.  A job's per-chunk top-L
//     lists determine its global top-LM exactly as long as no chunk has contributed all L of its entries (that chunk may hide an
//     (L+1)-th): the merge stops there and marks the list truncated.  Rounds per quarter-scale C4 pool against LM with nothing else
//     in the way (emulator): 12 -> 90 (the round-3 layout, 384 slots), 24 -> 62, 32 -> 60, 48 -> 59.
//   GE (good-enough-fitness < 1; config.clj:111 ships 0.8): there the "first offers above the threshold" list is the one that runs
//     out: every job of a window wants the SAME lowest-index offers above the threshold, so a round gets as far as that list reaches —
//     64 entries of it (one per lane of the walk), 32 best-fit entries for the jobs nothing clears the threshold for (rounds per quarter-scale
//     C4 pool at 0.8: 76 with round 3's lists of 16 / 12, 63 with 64 / 12, 41 with 64 / 24, 36 with 64 / 32).  The evaluation
//     hands the offers above the threshold over as a BIT per offer and chunk (complete: the merged list is exact to its last entry;
//     round 3's per-chunk lists of 12 cut the merged list at the first chunk with more than 12 such offers, usually the first).
// (64 since the end of round 5 — one entry per lane of the walk, 48 before: the reference's default K = 1000 4.79 -> 4.66 ms (7 -> 5 rounds per
//  pool: on an empty cluster best fit piles consecutive jobs onto the same offers and a list is stale after ~200 jobs), one C4 pool alone — one
//  GPU of the 8-GPU configuration — 40.9 -> 39.7 ms, eight pools on one GPU +- 0, C2 -1 %, C3 +1 %: profiles/r05zk_lm64_probe.txt,
//  r05y_variant_sweeps.txt; a staged job's LDS row grows from 613 to 805 bytes, segments get shorter and more)
#ifndef COOK_MV_LM
#define COOK_MV_LM 64
#endif
#ifndef COOK_MV_LM_GE
#define COOK_MV_LM_GE 32
#endif
template <bool GE>
struct VShape {
  static constexpr int LM = GE ? COOK_MV_LM_GE : COOK_MV_LM;  // merged best-fit entries per job
  static constexpr int LG = GE ? 64 : 0;           // merged good-enough entries per job
  static constexpr int LGS = GE ? 64 : 1;          // (array bound: never zero)
};
constexpr int MV_LM_MAX = COOK_MV_LM > COOK_MV_LM_GE ? COOK_MV_LM : COOK_MV_LM_GE, MV_LG_MAX = 64;
static_assert(VShape<false>::LM <= MV_LM_MAX && VShape<true>::LM <= MV_LM_MAX && VShape<true>::LG <= MV_LG_MAX, "buffer sizing");
static_assert(MV_LM_MAX <= 64 && MV_LG_MAX <= 64, "the walk holds one merged-list entry per lane");
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
constexpr int MV_RTHREADS = COOK_MV_RTHREADS;  // threads of the resolve workgroup: the staging is parallel over them, wave 0 walks
// Jobs staged in LDS per SEGMENT of the walk (at most; the offer-owner table shares the LDS: resolve_wseg).  The emulated tests: small,
// so that small inputs run many segments and rounds.
#ifndef COOK_MV_WSEG
#define COOK_MV_WSEG COOK_SHAPE(384, 96)
#endif
constexpr int MV_WSEG = COOK_MV_WSEG;
// Largest window the TILE path of the evaluation serves (rows of the eval grid = MV_WEVAL / 64): a round evaluates up to that many
// jobs against one snapshot and the walk consumes them segment by segment.
#ifndef COOK_MV_WEVAL
#define COOK_MV_WEVAL COOK_SHAPE(960, 256)
#endif
constexpr int MV_WEVAL = COOK_MV_WEVAL;
constexpr int MV_JG = MV_WEVAL / 64;       // job groups (waves of jobs) of such a window = rows of the eval grid
static_assert(MV_WEVAL % 64 == 0 && MV_JG >= 1, "whole job groups");
// A window may grow to MV_WLONG jobs once next to nothing of it has to be WALKED: when the cluster is full almost every job is
// settled in the parallel phase of the resolve kernel (no feasible offer under the snapshot, however the jobs before it fare) and
// needs no LDS at all.  One C4 pool spent 152 of its 604 rounds resolving 512 such jobs each; with long windows that tail takes
// about 20 rounds.
// (round 5: 10 240 instead of 2 560 — the tail of a C4 pool, 60 000 jobs that no offer can take any more, is 5 rounds instead of 25; one pool
//  42.1 -> 41.1 ms, eight pools 52.2 -> 50.7 ms; 5 120 / 20 480 measured 41.4 / 41.4 and 51.1 / 50.7: profiles/r05w_wlong_sweep.txt)
#ifndef COOK_MV_WLONG
#define COOK_MV_WLONG COOK_SHAPE(10240, 1024)
#endif
constexpr int MV_WLONG = COOK_MV_WLONG;
constexpr int MV_JGL = MV_WLONG / 64;      // job groups of a long window (stride of colbits)
static_assert(MV_WLONG % 64 == 0 && MV_WLONG >= MV_WEVAL && MV_WLONG < 65536, "JobL::b is 16 bits");
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
// What a lane of the placement walk needs when it becomes the owner of an offer, as ONE cache line: the offer's record and its state as
// of the last round's end (the resolve kernel keeps the state fields current next to MatchState's arrays, which the evaluation reads).
// Opening an offer was five cache lines (OfferA, OfferB, three state arrays) and ~1 000 cycles of the one walking wave per opened offer.
struct alignas(128) OfferW {
  double oc, om, rc, rm, inv_dc, inv_dm;  // = OfferA
  uint32_t host, k8s;                     // OfferB::host, flags bit 0
  int32_t run_count, task_slack;
  double ac, am;                          // assigned by the rounds so far
  int32_t acount;
  uint32_t pad[11];
};
static_assert(sizeof(OfferW) == 128, "one cache line per offer");
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
  unsigned stop_list, stop_full, stop_group, stop_window;  // why rounds ended (statistics): a truncated list ran out, 64 offers touched,
                                                           // second member of a group whose constraint can open offers, window used up
  unsigned segments;      // segments staged (the excess over the rounds that walked anything = continuations without a launch)
  unsigned touched_sum;   // sum over rounds of touched offers
  unsigned visited_sum;   // sum over rounds of jobs the walk had to visit (the rest were settled in parallel)
  unsigned long long t_setup, t_seq;  // resolve kernel: ticks (100 MHz wall clock) spent staging / in the sequential phase
  unsigned trunc_lists;   // walked jobs whose merged list carried the truncated flag (the merge stopped on a full chunk list)
  unsigned wgrow_pct;     // next window = this percentage of what the round resolved (window ended early) / of the window (it did not)
  unsigned wlong_cap;     // largest window the launch sequence allows (MV_WLONG, or MV_WEVAL when long windows are switched off)
  unsigned no_retire;     // the next round gives no lanes of dead offers away (the last one used few lanes: see resolve_round)
#ifdef COOK_WALK_PROF  // measurement build: shader cycles / jobs of the walk by outcome (0 shortcut, 1 touched offer wins, 2 new lane,
                       // 3 walked and unmatched, 4 member of a constrained group, 5 exact path ran)
  unsigned long long prof_cyc[8];
  unsigned prof_cnt[8];
#endif
};

struct RoundLog {  // one record per round (diagnostics; only written when V2Buf::round_log is set)
  unsigned head, wcur, resolved, n_list, touched, stop, matched, setup_ticks, seq_ticks, segments;
  // what the round was GIVEN, as checksums (only computed when a log is kept): the merged lists' summary words of the window, the offer
  // state and the alive bits the window was evaluated against, the static-constraint bits of the window's job groups
  unsigned h_cinfo, h_state, h_alive, h_col;
};
constexpr unsigned MV_ROUND_LOG_CAP = 8192;

// One offer chunk's candidates for one job, as ONE aligned record (128 bytes for best fit, 144 with the good-enough bits) that the
// evaluating lane writes and the merging lane reads in 16-byte pieces.
template <bool GE>
struct alignas(16) ChunkRecT {
  double fit[MV_L];         // fitness desc, offer index asc
  int idx[MV_L];            // -1 = no entry
  unsigned long long gm[GE ? MV_EW : 2];  // (GE) bit i of word w: the fitness of offer chunk * MV_OCB + w * MV_OCW + i exceeds good-enough
  unsigned cnt[4];          // n | nge << 8, offers failing on resources / constraints / zero fitness
};
static_assert(sizeof(ChunkRecT<false>) % 16 == 0 && sizeof(ChunkRecT<true>) % 16 == 0, "ChunkRec is moved in 16-byte pieces");
static_assert(offsetof(ChunkRecT<false>, cnt) + 16 == sizeof(ChunkRecT<false>) && offsetof(ChunkRecT<true>, cnt) + 16 == sizeof(ChunkRecT<true>),
              "the counts are the record's last 16-byte piece");
// an empty list is stored from this piece on (chunk_store): the merge reads n = 0 and ignores the rest
template <bool GE>
constexpr unsigned chunk_count_piece() { return (unsigned)(offsetof(ChunkRecT<GE>, cnt) / 16); }
// (chunk_store: platform.hpp)

struct V2Buf {
  RoundLog* round_log;
  unsigned split_max;  // cap of eval_split (1 = never cut a wave's offer batch)
#ifdef COOK_EVAL_TRACE
  unsigned long long* eval_trace;  // timing study build: per eval block [start, end] ticks of the 100 MHz clock + HW_ID
#endif
  const OfferA* oa;
  const OfferB* ob;
  OfferW* ow;          // [M] the walk's one-line records (state fields written by the resolve kernel)
  const JobRec* jr;
  const JobCons* jcons;  // [K] fast constraint slots of the jobs flagged JF_FASTC
  void* prec;          // [wlong][C]     chunk lists: one ChunkRecT<GE> per (job of the window, offer chunk)
  uint64_t* colbits;   // [M][JGL]       static-constraints-pass bit of (offer, job of the window)
  unsigned* jfh;       // [wlong][MV_FH + 2]  group members: the hosts their cotasks occupy under the snapshot (unique groups), how many
                       //                (int: -1 not gathered, -2 more than MV_FH), the group's last placed job — what the walk's fast
                       //                path needs, gathered ONCE by the evaluation (the tile of chunk 0 writes it)
  double* cand_fit;    // [wlong][LM]
  int* cand_idx;       // [wlong][LM]
  int* ge_idx;         // [wlong][LG]
  uint32_t* cinfo;     // [wlong][4]     ncand | nge << 8 | truncated << 16 | good-enough list truncated << 17, c1, c2, c4
  WinCtl* ctl;
  const MatchIn* in_dev;  // the MatchIn of this call in device memory (the walk only needs it for constrained groups)
  unsigned C;          // eval blocks along the offers
};

// ---- once per match call: pack offers and jobs -----------------------------------------------------------------------
COOK_KERNEL void match_pack_offers(const MatchIn* __restrict__ inp /* device copy of the call's MatchIn */, OfferA* __restrict__ oa,
                                   OfferB* __restrict__ ob, OfferW* __restrict__ ow) {
  const MatchIn& in = *inp;
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
  OfferW w;
  w.oc = a.oc, w.om = a.om, w.rc = a.rc, w.rm = a.rm, w.inv_dc = a.inv_dc, w.inv_dm = a.inv_dm;
  w.host = b.host, w.k8s = b.flags & 1u, w.run_count = b.run_count, w.task_slack = b.task_slack;
  w.ac = 0.0, w.am = 0.0, w.acount = 0;  // (nothing assigned yet: a match call starts from the offers as staged)
#pragma unroll
  for (int q = 0; q < 11; ++q) w.pad[q] = 0u;
  ow[v] = w;
}

COOK_KERNEL void match_pack_jobs(const MatchIn* __restrict__ inp, JobRec* __restrict__ jr, JobCons* __restrict__ jcons) {
  const MatchIn& in = *inp;
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
COOK_KERNEL void match_job_minima(const JobRec* __restrict__ jr, unsigned K, unsigned long long* __restrict__ jmin_bits, unsigned nblk /* blocks of this launch */) {
  double c = __longlong_as_double(0x7FF0000000000000ll), m = c;
  bool odd = false;  // a negative or non-finite request (jmin_bits[2]: match_v3 leaves such calls to the window rounds)
  for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += nblk * blockDim.x) {
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
COOK_KERNEL void match_init_alive(const OfferA* __restrict__ oa, unsigned M, const double* __restrict__ jmin,
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
// One workgroup's LDS: the offers its waves stage while they scan, and — in the SAME bytes, behind a workgroup barrier — the waves'
// lists for the tile's epilogue (as two regions a block took 46.6 KB: three blocks per CU whatever the register count).
template <bool GE>
struct EvalLds {
  union {
    EvalWaveLds wave[MV_EW];
    struct {
      double fit[MV_EW][COOK_WAVE][MV_L];
      int idx[MV_EW][COOK_WAVE][MV_L];
      unsigned long long ge[MV_EW][COOK_WAVE];  // (GE) the waves' good-enough bits
      unsigned cnt[MV_EW][COOK_WAVE][3];
    };
  };
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
  unsigned long long gm[MV_EW];  // (GE launches only) good-enough bits of the batches this lane's wave walked (one batch in a shared tile)
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
  for (int q = 0; q < MV_EW; ++q) E.gm[q] = 0ull;
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
                                                        unsigned v0, unsigned jg, unsigned nsub = MV_OCW, unsigned slot = 0,  // slot: E.gm word of this batch
                                                        unsigned long long* trp = nullptr) {  // (COOK_EVAL_TRACE builds: where the wave's time stamps go)
  (void)trp;
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
#ifdef COOK_EVAL_TRACE
  if (trp && lane == 0) trp[2] = cook_ticks();
#endif
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
#ifdef COOK_EVAL_TRACE
  if (trp && lane == 0) trp[3] = cook_ticks();
#endif
  unsigned long long feasm = 0ull, gem = 0ull;  // gem: bit vi = the fitness on offer v0 + vi exceeds good-enough
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
      if (GE && E.use_ge && !(ub < E.ge_lo)) prune = false;  // (it may clear the threshold: the exact value decides)
      if (!prune) {
        const double fit = fitness_of(a, ac, am, j.c, j.m);
        if (!(fit > 0.0)) {
          E.c4 += 1u;
        } else {
          if (fit > E.tf[MV_L - 1]) {
            topl_insert_ascending<MV_L>(E.tf, E.ti, fit, (int)v);
            if (E.ti[MV_L - 1] >= 0) E.thr = E.tf[MV_L - 1] * (1.0 - 0x1p-40);
          }
          if (GE && E.use_ge && fit > E.ge) gem |= 1ull << vi;
        }
      }
    }
  }
  // failure classes: offers failing on resources (the dead ones too), offers fitting on resources but infeasible (a constraint)
  const unsigned n_res = (unsigned)__popcll(resm);
  E.c1 += valid ? (v1 > v0 ? v1 - v0 : 0u) - n_res : 0u;
  E.c2 += n_res - (unsigned)__popcll(feasm);
  if (GE) {
#pragma unroll
    for (int q = 0; q < MV_EW; ++q)
      if ((unsigned)q == slot) E.gm[q] = gem;
  }
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
// a larger workgroup (w = wave in team, sync = the team's barrier; THROUGH = write-through stores: no shipped launch uses either).
template <bool THROUGH, bool GE = true, class Sync>
static __device__ __forceinline__ void eval_tile_t(char* lds, const MatchIn& in, const MatchState& st, const V2Buf& vb, unsigned head,
                                                   unsigned wcur, unsigned ch, unsigned jg, unsigned w, Sync sync, unsigned part = 0,
                                                   unsigned split = 1) {
  EvalLds<GE>& L = *reinterpret_cast<EvalLds<GE>*>(lds);
  auto& s_fit = L.fit;
  auto& s_idx = L.idx;
  auto& s_ge = L.ge;
  auto& s_cnt = L.cnt;
  if (jg * COOK_WAVE >= wcur || head + jg * COOK_WAVE >= in.K) return;  // uniform over the waves of the tile
  const unsigned lane = lane_id();
  const unsigned b = jg * COOK_WAVE + lane;
  EvalLane E;
#ifdef COOK_EVAL_TRACE
  unsigned long long* trp = vb.eval_trace ? vb.eval_trace + (size_t)vb.C * MV_JG * 3 + ((size_t)jg * vb.C + ch) * 32 + w * 8 : nullptr;
  if (trp && lane == 0) trp[0] = cook_ticks();
#endif
  eval_lane_setup<GE>(E, in, st, vb, head, wcur, jg);
#ifdef COOK_EVAL_TRACE
  if (trp && lane == 0) trp[1] = cook_ticks();
#endif
#ifdef COOK_EVAL_TRACE
  eval_scan_offers<THROUGH, GE>(E, L.wave[w], in, st, vb, ch * MV_OCB + w * MV_OCW + part * ((unsigned)MV_OCW / split), jg, (unsigned)MV_OCW / split, 0, trp);
  if (trp && lane == 0) trp[4] = cook_ticks();
#else
  eval_scan_offers<THROUGH, GE>(E, L.wave[w], in, st, vb, ch * MV_OCB + w * MV_OCW + part * ((unsigned)MV_OCW / split), jg, (unsigned)MV_OCW / split);
#endif
  const bool valid = E.valid, use_ge = GE && E.use_ge;
  // ---- merge the block's MV_EW wave lists per job through LDS -------------------------------------------------------
  sync();  // (the lists go where the waves' staged offers were: every wave of the tile is done scanning)
#pragma unroll
  for (int q = 0; q < MV_L; ++q) {
    s_fit[w][lane][q] = E.tf[q];
    s_idx[w][lane][q] = E.ti[q];
  }
  if constexpr (GE) s_ge[w][lane] = E.gm[0];
  s_cnt[w][lane][0] = E.c1;
  s_cnt[w][lane][1] = E.c2;
  s_cnt[w][lane][2] = E.c4;
  sync();
  if (w != 0 || !valid) return;  // (the caller synchronises the waves before the LDS is reused)
  if (ch == 0 && part == 0) eval_store_group<THROUGH>(E, vb, b);
  int p[MV_EW];
#pragma unroll
  for (int x = 0; x < MV_EW; ++x) p[x] = 0;
  ChunkRecT<GE> R;
  int n_out = 0;
#pragma unroll
  for (int q = 0; q < MV_L; ++q) {
    R.fit[q] = -1.0;
    R.idx[q] = -1;
  }
#pragma unroll
  for (int q = 0; q < (GE ? MV_EW : 2); ++q) R.gm[q] = 0ull;
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
  if constexpr (GE) {
    if (use_ge) {
#pragma unroll
      for (int x = 0; x < MV_EW; ++x) {
        R.gm[x] = s_ge[x][lane];
        n_g += __popcll(R.gm[x]);
      }
      n_g = n_g < 255 ? n_g : 255;
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
  chunk_store(&reinterpret_cast<ChunkRecT<GE>*>(vb.prec)[(size_t)b * (vb.C * split) + ch * split + part], R, THROUGH, (n_out | n_g) == 0 ? chunk_count_piece<GE>() : 0u);
#ifdef COOK_EVAL_TRACE
  if (trp && lane == 0) trp[5] = cook_ticks();
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
    eval_scan_offers<THROUGH, GE>(E, W, in, st, vb, v0, jg, MV_OCW, (unsigned)s);
  }
  if (!E.valid) return;
  if (ch == 0) eval_store_group<THROUGH>(E, vb, b);
  ChunkRecT<GE> R;
  int n_out = 0, n_g = 0;
#pragma unroll
  for (int q = 0; q < MV_L; ++q) {
    R.fit[q] = E.tf[q];
    R.idx[q] = E.ti[q];
    n_out += E.ti[q] >= 0 ? 1 : 0;
  }
#pragma unroll
  for (int q = 0; q < (GE ? MV_EW : 2); ++q) {
    R.gm[q] = GE ? E.gm[q < MV_EW ? q : 0] : 0ull;
    n_g += GE ? __popcll(R.gm[q]) : 0;
  }
  n_g = n_g < 255 ? n_g : 255;
  R.cnt[0] = (unsigned)n_out | ((unsigned)n_g << 8);
  R.cnt[1] = E.c1;
  R.cnt[2] = E.c2;
  R.cnt[3] = E.c4;
  chunk_store(&reinterpret_cast<ChunkRecT<GE>*>(vb.prec)[(size_t)b * vb.C + ch], R, THROUGH, (n_out | n_g) == 0 ? chunk_count_piece<GE>() : 0u);
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
  EvalLds<GE>& L = *reinterpret_cast<EvalLds<GE>*>(lds);
  const unsigned w = wave_id();
  for (unsigned jg = gy * MV_EW + w; jg * COOK_WAVE < wcur; jg += ny * MV_EW) eval_tile_wave<false, GE>(L.wave[w], in, st, vb, head, wcur, ch, jg);
}
template <bool GE>
__global__ void __launch_bounds__(COOK_WAVE* MV_EW) COOK_EVAL_OCCUPANCY match_eval2(MatchIn in, MatchState st, V2Buf vb) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(EvalLds<GE>)];
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
// lane = offer chunk (its per-chunk list as it is; further chunks of the same lane by the ascending insertion).  The merged best-fit
// list is cut — and marked truncated — the moment a lane whose chunk list(s) may continue beyond what it holds pops its last entry.
// The good-enough list is the first LG set bits of the chunks' masks in offer order (a prefix sum of the chunks' bit counts gives
// every lane the list positions of its offers): exact to its last entry, truncated only when more than LG offers clear the threshold.
template <bool GE>
static __device__ __forceinline__ void merge_job(const MatchIn& in, const V2Buf& vb, unsigned head, unsigned wcur, unsigned b,
                                                 unsigned split = 1) {  // split: eval_split(wcur) behind the launch path's eval grid
  if (b >= wcur || head + b >= in.K) return;
  constexpr int LM = VShape<GE>::LM, LG = VShape<GE>::LG;
  const unsigned lane = lane_id();
  const bool use_ge = GE && in.good_enough < 1.0;
  double tf[MV_L];
  int ti[MV_L];
#pragma unroll
  for (int q = 0; q < MV_L; ++q) {
    tf[q] = -1.0;
    ti[q] = -1;
  }
  unsigned n_g_total = 0;  // (GE) offers above the threshold seen so far, over all chunks (wave-uniform)
  unsigned c1 = 0, c2 = 0, c4 = 0;
  int n_seen = 0;     // entries of all the lane's chunks
  bool hide = false;  // the lane's best-fit list may end before its chunks' feasible offers do
  const unsigned cv = vb.C * split;  // chunk lists per job (virtual chunks, eval_split)
  const ChunkRecT<GE>* const prec = reinterpret_cast<const ChunkRecT<GE>*>(vb.prec);
  for (unsigned ch0 = 0; ch0 < cv; ch0 += COOK_WAVE) {  // (wave-uniform: the good-enough part scans over the lanes)
    const unsigned ch = ch0 + lane;
    const bool have = ch < cv;
    ChunkRecT<GE> R;
    if (have) R = prec[(size_t)b * cv + ch];  // 16-byte loads, all in flight together
    const unsigned info = have ? R.cnt[0] : 0u;
    if (have) {
      c1 += R.cnt[1];
      c2 += R.cnt[2];
      c4 += R.cnt[3];
    }
    const int n = (int)(info & 0xFFu);
    n_seen += n;
    hide = hide || n == MV_L || n_seen > MV_L;
    if (ch0 == 0u) {  // the lane's first chunk (its only one up to 64 chunks = 8 192 offers): the sorted list as it is
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
    if constexpr (GE) {
      if (use_ge && n_g_total < (unsigned)LG) {  // (wave-uniform) chunks ascend with the lane, offers with the word and the bit
        const bool any_bits = have && ((info >> 8) & 0xFFu) != 0u;
        unsigned cnt = 0;
#pragma unroll
        for (int x = 0; x < MV_EW; ++x) cnt += any_bits ? (unsigned)__popcll(R.gm[x]) : 0u;
        unsigned incl = cnt;  // inclusive prefix sum over the lanes
        for (unsigned d = 1; d < (unsigned)COOK_WAVE; d <<= 1) {
          const unsigned o = (unsigned)__shfl_up((int)incl, d, COOK_WAVE);
          if (lane >= d) incl += o;
        }
        unsigned pos = n_g_total + incl - cnt;  // list position of this lane's first offer
        if (cnt != 0u && pos < (unsigned)LG) {
#pragma unroll
          for (int x = 0; x < MV_EW; ++x) {
            for (unsigned long long m = R.gm[x]; m != 0ull && pos < (unsigned)LG; m &= m - 1ull, ++pos)
              vb.ge_idx[(size_t)b * LG + pos] = (int)(ch * (unsigned)MV_OCB + (unsigned)x * (unsigned)MV_OCW + (unsigned)__ffsll((unsigned long long)m) - 1u);
          }
        }
        n_g_total += (unsigned)__shfl((int)incl, COOK_WAVE - 1, COOK_WAVE);
      }
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    c1 += __shfl_xor(c1, d, COOK_WAVE);
    c2 += __shfl_xor(c2, d, COOK_WAVE);
    c4 += __shfl_xor(c4, d, COOK_WAVE);
  }
  int n_out = 0;
  bool trunc = false;  // the merged list may not hold every feasible offer
  for (int round = 0; round < LM; ++round) {
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
      vb.cand_fit[(size_t)b * LM + round] = best.fit;
      vb.cand_idx[(size_t)b * LM + round] = best.idx;
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
    if (__any(emptied)) {  // a list that may continue beyond what the lane holds just ran out: stop here
      trunc = true;
      break;
    }
  }
  if (!trunc) trunc = __any(ti[0] >= 0);  // LM entries emitted and some lane still holds more
  const unsigned n_g = GE ? (n_g_total < (unsigned)LG ? n_g_total : (unsigned)LG) : 0u;
  // (the chunks' bit counts saturate at 255 only in the record's count byte, never in the masks; once LG offers are listed the scan
  //  above stops, so "more than LG" is all n_g_total can say beyond that point)
  const bool gtrunc = GE && n_g_total >= (unsigned)LG && LG > 0;
  if (lane == 0) {
    vb.cinfo[(size_t)b * 4 + 0] = (unsigned)n_out | (n_g << 8) | (trunc ? 1u << 16 : 0u) | (gtrunc ? 1u << 17 : 0u);
    vb.cinfo[(size_t)b * 4 + 1] = c1;
    vb.cinfo[(size_t)b * 4 + 2] = c2;
    vb.cinfo[(size_t)b * 4 + 3] = c4;
  }
}

// one wave per job; a block of MV_MW waves takes MV_MW jobs per pass
constexpr int MV_MW = 4;
constexpr int MV_MERGE_BLOCKS = COOK_SHAPE(240, 16);  // blocks of the merge grid (x MV_MW waves: one pass for the windows of the tile path)
template <bool GE>
static __device__ __forceinline__ void merge_block(const MatchIn& in, const V2Buf& vb) {
  const unsigned head = vb.ctl->head, wcur = vb.ctl->wcur;
  const unsigned split = wcur <= (unsigned)MV_WEVAL ? eval_split(wcur, vb.split_max) : 1u;  // as match_eval2's grid cut the offers
  for (unsigned b = blockIdx.x * MV_MW + wave_id(); b < wcur; b += gridDim.x * MV_MW) merge_job<GE>(in, vb, head, wcur, b, split);
}
template <bool GE>
__global__ void __launch_bounds__(COOK_WAVE* MV_MW) match_merge2(MatchIn in, V2Buf vb) {
  merge_block<GE>(in, vb);
}

// ---- resolve -----------------------------------------------------------------------------------------------------------------
struct JobL {  // a job of the window as the walk reads it (one 32-byte LDS record)
  double c, m;
  unsigned info;  // bits 0-7 ncand, 8-15 nge, 16 gpu job, 17 member of a constrained group, 18-19 group type
  unsigned group;
  unsigned short f1, f2, f4;  // saturated counts of offers failing on resources / constraints / zero fitness under S
  unsigned short b;           // window position of the job (the record itself sits at its WALK position)
};
constexpr unsigned JL_GPU = 1u << 16, JL_GROUPED = 1u << 17, JL_HASGROUP = 1u << 20;  // (bits 18-19: group type)
constexpr unsigned JL_XRES = 1u << 28;    // asks for ports / named scalars: general path only
constexpr unsigned JL_TRUNC = 1u << 29;   // the merged list may not hold every feasible offer (cinfo bit 16)
constexpr unsigned JL_GTRUNC = 1u << 30;  // the good-enough list may not hold every offer above the threshold (cinfo bit 17)
// "entries may exist beyond the job's list" / "the list holds every feasible offer" inside the walk (cinfo_u: the walk's local)
#define COOK_L_TRUNC() ((cinfo_u & JL_TRUNC) != 0u)
#define COOK_L_COMPLETE() ((cinfo_u & JL_TRUNC) == 0u)
constexpr unsigned JL_GSLOT_SHIFT = 21, JL_GSLOT_NONE = 0x7Fu;  // bits 21-27: the job's row of ResolveFixed::gfh, or none
constexpr int MV_GMAX = 64;  // group members per segment whose hosts-to-avoid are staged for the walk's fast path
// Offers a round may touch beyond its 64 lanes: when every lane is taken, a lane whose offer is DEAD — it cannot take even the smallest
// job of the call any more, so no later job can go there — is given to the next offer (the dead offer's state is written back at once,
// its byte in the owner table says "dead": list entries that name it are skipped like touched offers that do not fit).  On the
// benchmark's pools 40-50 of the 64 lanes are dead when the 65th offer is asked for (best fit fills offers to the brim).
// List entries per job whose OfferW line and colbits word the STAGING of a segment touches, so that the walk's open_lane finds them in
// the L2 of the XCD the workgroup runs on (they were last written / read by evaluation blocks all over the chip): 70 % of the offers a
// walk opens are among the first four entries of the job's list, 80 % among the first eight (emulator, C4 pool).  Costs the walking
// wave nothing: the other waves of the workgroup issue the loads while they stage.
#ifndef COOK_MV_PF
#define COOK_MV_PF 8
#endif
constexpr int MV_PF = COOK_MV_PF;
constexpr unsigned MV_RETIRE_CAP = 192;
constexpr unsigned MV_TMAX = (unsigned)MV_T + MV_RETIRE_CAP;  // offers one round can touch at most
constexpr unsigned OWNER_UNTOUCHED = 0xFFu, OWNER_NONE = 0xFEu, OWNER_DEAD = 0xFDu;  // values of the owner table / of JobRegs::owner beside lane numbers

// (WALK_STAT: platform.hpp — counters of the emulated build's design studies, nothing on the GPU)

// The resolve workgroup's LDS: this fixed part, then — sized at run time from the number of offers (resolve_wseg) — the segment's
// job records, candidate lists and results BY WALK POSITION, and the owner table: one byte per OFFER of the pool, the lane that
// owns it in this round, 0xFF = untouched.
struct ResolveFixed {
  unsigned long long visit[MV_JGL];  // bit b of the window: the walk has to visit job b (the others are settled in parallel)
  unsigned vbase[MV_JGL + 1];        // walk position of the first visited job of each 64-job group
  // members of unique (or unconstrained) host-placement groups among the segment's jobs: the hosts their cotasks occupied when the
  // round began (running ++ placed by earlier rounds; 0xFFFFFFFF = unused) and the group's last placed job then
  unsigned gfh[MV_GMAX][MV_FH];
  int glast[MV_GMAX];
  unsigned n_gslots;
  unsigned dbg_h[4];                // (round log only) checksums of the round's inputs, see RoundLog
  int cmd;                          // the walker's word to the other waves: 1 = stage the next segment, 0 = the round is over
  unsigned seg_lo;                  // first walk position of the segment being staged
  int sink[COOK_WAVE];              // where lanes 1..63 put their copy of a result the walk stores (see store_result)
  unsigned char sinkb[COOK_WAVE];
  // what a lane WITHOUT a list entry loads instead of one (the walk's loads are select-on-the-address, never a branch on the lane)
  double fit_none;                  // -1
  int off_none;                     // -1
  unsigned char owner_none[4];      // 0xFE = "no entry"
  // ports / named scalars assigned on a touched offer when the round began, by owner lane: saved by the first job of the round
  // that moves them (the failure summary of an unmatched job compares against the round's snapshot)
  double x0s[MV_T][3];
  int x0p[MV_T];
  unsigned char x0set[MV_T];
  // the state of a touched offer as the round began, by owner lane (the failure summary of an unmatched job swaps each touched offer's
  // verdict under the snapshot for its current one)
  double ac0[MV_T], am0[MV_T];
  int acount0[MV_T];
};
constexpr unsigned MV_RLDS_BYTES = 160u * 1024u - 2048u;  // the workgroup's static LDS array (the CU has 160 KB)
template <bool GE>
constexpr unsigned resolve_job_bytes() {  // LDS per staged job: record, best-fit entries (fitness + offer), good-enough entries, result, failure code
  return (unsigned)sizeof(JobL) + 12u * (unsigned)VShape<GE>::LM + 4u * (unsigned)VShape<GE>::LG + 4u + 1u;
}
constexpr unsigned resolve_fixed_bytes() { return ((unsigned)sizeof(ResolveFixed) + 15u) / 16u * 16u + 128u; }  // (+ alignment slack of the carved arrays)
// jobs per segment for a pool of M offers (0 = the owner table alone does not fit: the host refuses such a pool)
template <bool GE>
static __host__ __device__ __forceinline__ unsigned resolve_wseg(unsigned M) {
  const unsigned owner = (M + 16u) / 16u * 16u;
  if (resolve_fixed_bytes() + owner >= MV_RLDS_BYTES) return 0u;
  const unsigned w = (MV_RLDS_BYTES - resolve_fixed_bytes() - owner) / resolve_job_bytes<GE>();
  return w < (unsigned)MV_WSEG ? w : (unsigned)MV_WSEG;
}
constexpr unsigned MV_WSEG_MIN = 16;  // pools whose owner table leaves less than that per segment are refused (about 150 000 offers)

// The run-time part of the resolve workgroup's LDS, carved behind ResolveFixed (resolve_wseg sizes it)
template <bool GE>
struct SegLds {
  JobL* job;             // [wseg] the segment's jobs, in rank order (walk position - seg_lo; JobL::b = window position)
  double* efit;          // [wseg][LM] fitness under S of the candidate entries, by walk position
  int* eoff;             // [wseg][LM] offer of the entry, -1 = none
  int* goff;             // [wseg][LG] (GE) good-enough entries: offer, -1 = none
  int* j2o;              // [wseg] results of the walk BY WALK POSITION, flushed to HBM once per segment: a global store inside the
                         //        walk would stall later s_waitcnt vmcnt(0) on its acknowledgement
  unsigned char* fail;   // [wseg] failure codes, by walk position
  unsigned char* owner;  // [M] owner lane of an offer, OWNER_UNTOUCHED / OWNER_DEAD
  unsigned wseg;
  __device__ __forceinline__ SegLds(char* lds, unsigned M) {
    constexpr int LM = VShape<GE>::LM, LG = VShape<GE>::LG;
    wseg = resolve_wseg<GE>(M);
    char* carve = lds + ((sizeof(ResolveFixed) + 15u) / 16u * 16u);
    job = reinterpret_cast<JobL*>(carve);
    carve += (size_t)wseg * sizeof(JobL);
    efit = reinterpret_cast<double*>(carve);
    carve += (size_t)wseg * LM * 8u;
    eoff = reinterpret_cast<int*>(carve);
    carve += (size_t)wseg * LM * 4u;
    goff = reinterpret_cast<int*>(carve);
    carve += (size_t)wseg * LG * 4u;
    j2o = reinterpret_cast<int*>(carve);
    carve += (size_t)wseg * 4u;
    fail = reinterpret_cast<unsigned char*>(carve);
    carve += ((size_t)wseg + 15u) / 16u * 16u;
    owner = reinterpret_cast<unsigned char*>(carve);
  }
};

// ---- once per round (all threads): the owner table, the jobs the walk can skip ---------------------------------------------------
// A job without any feasible offer under S stays unmatched whatever the jobs before it do (placements only take capacity away;
// constrained groups excepted), and its failure summary cannot change when every class it reports is backed by more offers than the
// round can touch (t_max): such jobs are settled here, in parallel, and the walk skips them.  -> the number of jobs the walk must visit
// (L.visit: their bits by window position, L.vbase: walk position of the first visited job of each 64-job group).
template <bool GE>
static __device__ __forceinline__ unsigned resolve_settle(ResolveFixed& L, const SegLds<GE>& S, const MatchState& st, const V2Buf& vb, unsigned head,
                                                          unsigned nwin, unsigned M, unsigned t_max) {
  const unsigned tid = threadIdx.x, NT = blockDim.x;
  for (unsigned x = tid; x < (M + 3u) / 4u; x += NT) reinterpret_cast<unsigned*>(S.owner)[x] = 0xFFFFFFFFu;
  if (tid < MV_JGL) L.visit[tid] = 0ull;
  if (tid < (unsigned)MV_T) L.x0set[tid] = 0;
  if (tid == 0) {
    L.fit_none = -1.0;
    L.off_none = -1;
    L.owner_none[0] = L.owner_none[1] = L.owner_none[2] = L.owner_none[3] = (unsigned char)OWNER_NONE;
    L.cmd = 0;
    L.n_gslots = 0;
    L.dbg_h[0] = L.dbg_h[1] = L.dbg_h[2] = L.dbg_h[3] = 0u;
  }
  __syncthreads();
  if (vb.round_log) {  // diagnostics: what this round was given
    const unsigned ngrp = (nwin + COOK_WAVE - 1) / COOK_WAVE;
    unsigned hs = 0, ha = 0, hc = 0;
    for (unsigned v = tid; v < M; v += NT) {
      const unsigned long long a = (unsigned long long)__double_as_longlong(st.ac[v]), m2 = (unsigned long long)__double_as_longlong(st.am[v]);
      hs += (unsigned)(a >> 20) * (v + 1u) + (unsigned)(m2 >> 20) * (v + 7u) + (unsigned)st.acount[v] * 131u;
      for (unsigned g = 0; g < ngrp; ++g) {
        const unsigned long long w = vb.colbits[(size_t)v * MV_JGL + g];
        hc += ((unsigned)w ^ (unsigned)(w >> 32)) * (v * 31u + g + 1u);
      }
    }
    for (unsigned w2 = tid; w2 < (M + 63u) / 64u; w2 += NT) {
      const unsigned long long w = st.alive[w2];
      ha += ((unsigned)w ^ (unsigned)(w >> 32)) * (w2 + 1u);
    }
    atomicAdd(&L.dbg_h[1], hs);
    atomicAdd(&L.dbg_h[2], ha);
    atomicAdd(&L.dbg_h[3], hc);
  }
  for (unsigned b = tid; b < nwin; b += NT) {
    const unsigned flags = vb.jr[head + b].flags;
    const unsigned info = vb.cinfo[(size_t)b * 4 + 0];
    const unsigned c1 = vb.cinfo[(size_t)b * 4 + 1], c2 = vb.cinfo[(size_t)b * 4 + 2], c4 = vb.cinfo[(size_t)b * 4 + 3];
    if (vb.round_log) atomicAdd(&L.dbg_h[0], (info * 31u + c1 * 7u + c2 * 3u + c4) * (b + 1u));
    // members of balanced / attribute-equals groups excepted: a cotask's placement can make an offer FEASIBLE for them; a unique
    // group only ever takes hosts away (constraints.clj:586-598), like a resource
    const bool opens = (flags & JF_GROUPED) != 0 && ((flags >> 8) & 3u) != 1u;
    const bool trivial = (info & 0xFFFFu) == 0u && !opens && c1 > 0u && (c2 == 0u || c2 > t_max) && (c4 == 0u || c4 > t_max);
    if (trivial) {
      // final whatever this round does, also for a job behind the point where the round stops: job_to_offer keeps the -1 it was
      // initialised with; should the job still be unresolved next round, its summary is simply rewritten under the newer snapshot
      if (st.fail_code) st.fail_code[head + b] = 1u | (c2 ? 2u : 0u) | (c4 ? 4u : 0u);
    } else {
      atomicOr(&L.visit[b >> 6], 1ull << (b & 63u));
    }
  }
  __syncthreads();
  if (tid <= (unsigned)MV_JGL) {  // every thread sums its own prefix ([MV_JGL] = the total)
    unsigned acc = 0;
    for (unsigned g = 0; g < tid; ++g) acc += (unsigned)__popcll(L.visit[g]);  // (groups beyond the window hold no bits)
    L.vbase[tid] = acc;
  }
  __syncthreads();
  return wave_uniform_u32(L.vbase[MV_JGL]);
}

// ---- the segment [lo, lo + n) of walk positions -> LDS, by walk position (all threads; L.n_gslots = 0 and a barrier behind it are
// ---- the caller's) -> n ------------------------------------------------------------------------------------------------------
template <bool GE>
static __device__ __forceinline__ unsigned resolve_stage_segment(ResolveFixed& L, const SegLds<GE>& S, const V2Buf& vb, unsigned head, unsigned nwin,
                                                                 unsigned n_list, unsigned lo, double good_enough) {
  constexpr int LM = VShape<GE>::LM, LG = VShape<GE>::LG;
  const unsigned tid = threadIdx.x, NT = blockDim.x;
  const bool use_ge = GE && good_enough < 1.0;
  const unsigned ngrp = (nwin + COOK_WAVE - 1) / COOK_WAVE;
  const unsigned hi = lo + S.wseg < n_list ? lo + S.wseg : n_list;
  // the job groups of the window that hold walk positions of the segment
  unsigned g0 = 0;
  while (g0 + 1 < ngrp && L.vbase[g0 + 1] <= lo) ++g0;
  // pass 1, thread = window position: the records of the visited jobs, compacted to walk positions
  for (unsigned b = g0 * COOK_WAVE + tid; b < nwin && L.vbase[b >> 6] < hi; b += NT) {
    const unsigned long long vw = L.visit[b >> 6];
    if (!((vw >> (b & 63u)) & 1ull)) continue;
    const unsigned i = L.vbase[b >> 6] + (unsigned)__popcll(vw & ((1ull << (b & 63u)) - 1ull));
    if (i < lo || i >= hi) continue;
    const unsigned x = i - lo;
    const unsigned info = vb.cinfo[(size_t)b * 4 + 0];
    const unsigned c1 = vb.cinfo[(size_t)b * 4 + 1], c2 = vb.cinfo[(size_t)b * 4 + 2], c4 = vb.cinfo[(size_t)b * 4 + 3];
    const JobRec j = vb.jr[head + b];
    S.fail[x] = 0;  // a visited job that gets matched leaves it at that
    JobL r;
    r.c = j.c;
    r.m = j.m;
    const bool grouped = (j.flags & JF_GROUPED) != 0;
    r.info = (info & 0xFFFFu) | (j.g > 0 ? JL_GPU : 0u) | (grouped ? JL_GROUPED : 0u) | (((j.flags >> 8) & 3u) << 18) |
             (j.group != 0xFFFFFFFFu ? JL_HASGROUP : 0u) | ((j.flags & JF_XRES) ? JL_XRES : 0u) |
             ((info & (1u << 16)) ? JL_TRUNC : 0u) | ((info & (1u << 17)) ? JL_GTRUNC : 0u);
    // a member of a unique (type 1) or unconstrained (type 0) group: stage what the walk's fast path needs — the hosts to avoid
    // as the round begins and the group's last placed job (for the chain link) — so that it never has to go to HBM for them
    unsigned gslot = JL_GSLOT_NONE;
    const unsigned gt = (j.flags >> 8) & 3u;
    if (j.group != 0xFFFFFFFFu && gt <= 1u && (GE || !(good_enough < 1.0)) && vb.in_dev->host_dup == 0u && !(j.flags & JF_XRES)) {
      const unsigned* row = vb.jfh + (size_t)b * (MV_FH + 2);  // gathered by the evaluation of this round
      const int nfh = (int)row[MV_FH];
      if (gt == 0u || (nfh >= 0 && nfh <= MV_FH)) {
        const unsigned gs = atomicAdd(&L.n_gslots, 1u);
        if (gs < (unsigned)MV_GMAX) {
#pragma unroll
          for (int y = 0; y < MV_FH; ++y) L.gfh[gs][y] = gt == 1u ? row[y] : 0xFFFFFFFFu;
          L.glast[gs] = (int)row[MV_FH + 1];
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
    S.job[x] = r;
  }
  __syncthreads();
  // pass 2, thread = list entry: the candidate lists by walk position.  The loads do not wait for the job's counts (the arrays are
  // sized for every entry of every job of a window: entries beyond a list hold stale values, replaced by "none" behind the load)
  const unsigned n = hi - lo;
#pragma unroll 4
  for (unsigned e = tid; e < n * (unsigned)LM; e += NT) {
    const unsigned x = e / (unsigned)LM, q = e % (unsigned)LM;
    const JobL* jl = &S.job[x];
    const unsigned b = jl->b, nl = jl->info & 0xFFu;
    const int o = vb.cand_idx[(size_t)b * LM + q];
    const double f = vb.cand_fit[(size_t)b * LM + q];
    S.efit[e] = q < nl ? f : -1.0;
    S.eoff[e] = q < nl ? o : -1;
  }
  if constexpr (LG > 0) {
#pragma unroll 4
    for (unsigned e = tid; e < n * (unsigned)LG; e += NT) {
      const unsigned x = e / (unsigned)LG, q = e % (unsigned)LG;
      const JobL* jl = &S.job[x];
      const unsigned b = jl->b, ngl = (jl->info >> 8) & 0xFFu;
      const int o = vb.ge_idx[(size_t)b * LG + q];
      S.goff[e] = (use_ge && q < ngl) ? o : -1;
    }
  }
  __syncthreads();
  return n;
}

// (MV_PF) the waves that do not walk touch what opening the first entries of the segment's lists would read — behind the staging's
// last barrier, i.e. while wave 0 already walks
template <bool GE>
static __device__ __forceinline__ void resolve_prefetch_segment(const SegLds<GE>& S, const V2Buf& vb, unsigned n) {
  constexpr int LM = VShape<GE>::LM;
  const unsigned tid = threadIdx.x, NT = blockDim.x;
  unsigned pf_sink = 0u;
  for (unsigned e = tid - COOK_WAVE; e < n * (unsigned)MV_PF; e += NT - COOK_WAVE) {
    const unsigned x = e / (unsigned)MV_PF, q = e % (unsigned)MV_PF;
    const int o = S.eoff[(size_t)x * LM + q];
    if (o < 0) continue;
    PREFETCH_WORD(pf_sink, &vb.ow[(unsigned)o]);
    PREFETCH_WORD(pf_sink, &vb.colbits[(size_t)(unsigned)o * MV_JGL + ((unsigned)S.job[x].b >> 6)]);
  }
  PREFETCH_DRAIN(pf_sink);
}

// ---- the end of a round (lane 0 of the walking wave): statistics, the window of the next round, the control block back to HBM -----
static __device__ __forceinline__ void resolve_finish(WinCtl& ctl, const V2Buf& vb, unsigned head, unsigned nwin, unsigned resolved, unsigned stop,
                                                      unsigned matched, unsigned head_matched, unsigned touched, unsigned n_list,
                                                      unsigned n_segments, unsigned n_trunc, unsigned long long t_stage, unsigned long long t_all,
                                                      const unsigned* dbg_h) {
  ctl.head = head + resolved;
  ctl.rounds += 1;
  ctl.matched += matched;
  ctl.head_matched = head_matched;
  ctl.touched_sum += touched;
  ctl.visited_sum += n_list;
  ctl.segments += n_segments;
  ctl.t_setup += t_stage;
  ctl.t_seq += t_all - t_stage;
  ctl.trunc_lists += n_trunc;
  if (vb.round_log && ctl.rounds <= MV_ROUND_LOG_CAP) {
    RoundLog r;
    r.head = head, r.wcur = ctl.wcur, r.resolved = resolved, r.n_list = n_list, r.touched = touched, r.stop = stop, r.matched = matched;
    r.setup_ticks = (unsigned)t_stage, r.seq_ticks = (unsigned)(t_all - t_stage), r.segments = n_segments;
    r.h_cinfo = dbg_h[0], r.h_state = dbg_h[1], r.h_alive = dbg_h[2], r.h_col = dbg_h[3];
    vb.round_log[ctl.rounds - 1] = r;
  }
  if (stop == 1) ctl.stop_list += 1;
  if (stop == 2 || stop == 5) ctl.stop_full += 1;
  if (stop == 3) ctl.stop_group += 1;
  if (stop == 0) ctl.stop_window += 1;
  // adapt the window: a multiple of what a round resolves (more = fewer rounds, less = fewer jobs evaluated twice)
  unsigned wn = stop == 0 ? ctl.wcur * 2 : (unsigned)(((unsigned long long)resolved * ctl.wgrow_pct + 99ull) / 100ull);
  if (wn < 64) wn = 64;
  // past MV_WEVAL only while next to nothing of a window has to be walked (see MV_WLONG), and never beyond what this launch
  // sequence sized its buffers and grids for
  unsigned cap = (unsigned)MV_WEVAL;
  if (stop == 0 && nwin >= (unsigned)MV_WEVAL && n_list * 8u <= nwin) cap = ctl.wlong_cap > cap ? ctl.wlong_cap : cap;
  if (wn > cap) wn = cap;
  ctl.wcur = wn;
  ctl.no_retire = touched < (unsigned)MV_T * 3u / 4u ? 1u : 0u;
  *vb.ctl = ctl;
}

// One round of the window walk by ONE workgroup of MV_RTHREADS threads (all of them must call it): resolve_settle, then per segment
// resolve_stage_segment (all threads) / resolve_prefetch_segment (the waves that do not walk) and the walk below (wave 0),
// resolve_finish at the end.
//
// The walk is one dependent chain run by a single wave.  What it costs per job is the number of INSTRUCTIONS on the job's path — a
// wave issues one every fourth cycle or so: 945 cycles for the ~200 instructions of a job that goes to an offer touched before
// (-DCOOK_WALK_PROF) — not the latencies of scripts/ubench_wave.hip one by one (dependent LDS read 60-68 cycles,
// compiler-form DPP reduction 166, ballot -> ffs -> readlane 62): those are hidden behind the issue of the rest.  The loop is
// organised as (1) a look-ahead of ONE job over walk records that are laid out by WALK position (no dependent address chain: record
// and list entries of job i+1 are loaded while job i is decided, the owner look-up of its entries at the end of job i's turn), the
// fast loop unrolled by two over two register sets so that the look-ahead costs no register rotation; (2) a FAST PATH for the
// common job — no constrained group, finite positive fitness values — that orders the touched offers by an fp32 image of the
// approximate fitness (one hand-placed DPP reduction, gpu_prims.hpp), written without a branch on the lane number (stores, loads and
// bookings are selects: see store_result / open_lane / take_job for what such a branch does to the whole loop), and falls back to
// (3) the GENERAL PATH below it whenever the order is not certain at fp32 resolution (two touched offers within 2^-20, touched and
// untouched best within 2^-38), the job is unmatched, or anything unusual is involved.  Both paths produce the same decision; only
// the general path knows every rule.
template <bool GE>  // GE: the launch was made for good-enough-fitness < 1 (list shape VShape<true>; the fast path knows the "first offer above the threshold" rule)
static __device__ void resolve_round(char* lds, MatchState st, const V2Buf& vb) {
  constexpr int LM = VShape<GE>::LM, LG = VShape<GE>::LG;
  constexpr bool GEF = GE;
  ResolveFixed& L = *reinterpret_cast<ResolveFixed*>(lds);
  auto& s_gfh = L.gfh;
  auto& s_glast = L.glast;
  unsigned& s_ngslots = L.n_gslots;
  const unsigned tid = threadIdx.x, lane = lane_id();
  WinCtl ctl = *vb.ctl;
  // (the launch's scalars through scalar registers, explicitly: in the multi-pool kernels `vb` and `st` are read from a context record
  //  in memory, their pointers are generic pointers to the compiler, and whatever is loaded through a generic pointer counts as a
  //  per-lane value — the walk loop built on them would run under execution masks with its counters in vector registers)
  ctl.head = wave_uniform_u32(ctl.head);
  ctl.wcur = wave_uniform_u32(ctl.wcur);
  const unsigned head = ctl.head;
  const unsigned K = wave_uniform_u32(vb.in_dev->K);
  if (head >= K) return;
  const unsigned M = wave_uniform_u32(vb.in_dev->M);
  const unsigned long long tk0 = cook_ticks();
  const unsigned wend = (head + ctl.wcur < K) ? head + ctl.wcur : K;
  const unsigned nwin = wend - head;
  const double good_enough = wave_uniform_f64(vb.in_dev->good_enough);
  const bool use_ge = GE && good_enough < 1.0;
  const uint32_t* const j_index = wave_uniform_ptr(vb.in_dev->j_index);
  // dead lanes are given away (MV_RETIRE_CAP) unless jobs of the call move ports / named scalars (their per-lane snapshots would go with the lane)
  // ... and unless the previous round used few of its lanes: a round that may touch MV_TMAX offers must WALK every unmatched job whose
  // failure classes are backed by no more offers than that, and once the cluster is full (rounds that touch a handful of offers, windows
  // of thousands of jobs settled in parallel) those would be most of the queue
  const bool can_retire = wave_uniform_u32(vb.in_dev->has_x) == 0u && wave_uniform_u32(ctl.no_retire) == 0u;
  const unsigned t_max = can_retire ? MV_TMAX : (unsigned)MV_T;  // offers this round can touch at most
  const double jmin_c = wave_uniform_f64(st.jmin[0]), jmin_m = wave_uniform_f64(st.jmin[1]);
  // the run-time part of the LDS
  const SegLds<GE> S(lds, M);
  JobL* const s_job = S.job;
  double* const s_efit = S.efit;
  int* const s_eoff = S.eoff;
  int* const s_goff = S.goff;
  int* const s_j2o = S.j2o;
  unsigned char* const s_fail = S.fail;
  unsigned char* const s_owner = S.owner;
  // ---- once per round (all threads): what the walk can skip, the owner table -------------------------------------------------------
  const unsigned n_list = resolve_settle<GE>(L, S, st, vb, head, nwin, M, t_max);  // jobs the walk has to visit
  auto stage_segment = [&](unsigned lo) -> unsigned { return resolve_stage_segment<GE>(L, S, vb, head, nwin, n_list, lo, good_enough); };
  auto prefetch_segment = [&](unsigned n) { resolve_prefetch_segment<GE>(S, vb, n); };
  unsigned seg_lo = 0;
  unsigned n_eff = stage_segment(0);  // walk positions of the segment
  unsigned n_segments = 1;
  if (tid >= COOK_WAVE) {  // the other waves: asleep at the barrier until wave 0 asks for the next segment or ends the round
    for (;;) {
      prefetch_segment(n_eff);
      EMU_SITE("resolve: helper waiting");
      __syncthreads();
      if (L.cmd == 0) break;
      seg_lo = wave_uniform_u32(L.seg_lo);
      n_eff = stage_segment(seg_lo);
    }
    return;
  }
  // wave 0 walks the window
  unsigned long long tk1 = cook_ticks();
  unsigned long long t_stage = tk1 - tk0;
  // ---- sequential phase ---------------------------------------------------------------------------------------------------
  // Lanes own the offers touched in this round (state in registers).  Cross-lane traffic is ballots, v_readlane and DPP
  // reductions (no ds_bpermute); fitness values are first compared through a reciprocal-multiply approximation (relative error
  // < 2^-50) and the two fp64 divides are only executed when candidates are closer than 2^-38 relative — exactness is unaffected.
  int t_v = -1;  // the lane's offer (-1: the lane owns none yet)
  double t_oc = 0, t_om = 0, t_rc = 0, t_rm = 0, t_invc = 0, t_invm = 0;
  double t_ac = 0, t_am = 0, t_basec = 0, t_basem = 0;
  int t_acount = 0, t_run = 0, t_slack = 0;
  unsigned t_k8s = 0, t_host = 0;
  unsigned long long t_col = 0ull;  // the offer's static-constraints-pass bits of job group cur_g of the window (colbits)
  unsigned long long t_coln = 0ull;  // ... and of group col_next, fetched when the walk entered cur_g (nothing waits for it)
  unsigned col_next = 0xFFFFFFFFu;
  // group members placed in THIS round, one per lane in placement order (group, host, match index): what a later member of the same
  // group has to avoid / link to, without asking HBM.  n_log > 64: the log overflowed, no fast path for group members any more
  unsigned lg_group = 0xFFFFFFFFu, lg_host = 0u, n_log = 0u;
  int lg_k = -1;
  unsigned cur_g = 0xFFFFFFFFu;
  unsigned nT = 0;
  unsigned n_retired = 0;  // dead offers whose lanes were given away in this round
  unsigned stop = 0;  // 1 list exhausted, 2 touched set full, 3 group barrier, 5 an unmatched job's summary needs a fresh snapshot
  unsigned matched = 0, head_matched = ctl.head_matched;
  unsigned resolved = nwin;
  unsigned n_trunc = 0;          // walked jobs with a truncated merged list (statistics)
  // guard bands of the approximate fitness (relative 2^-38; the approximation is good to ~2^-50): x - x * 2^-38 and x + x * 2^-38 through
  // v_ldexp_f64 with an inline exponent — as multiplications by 1 -+ 2^-38 the two fp64 constants lived in VGPRs, were spilled, and the
  // walk's fast path reloaded them from scratch memory for every job (two dependent scratch loads on the critical path)
  // fp32 threshold below which a touched offer's approximate fitness cannot be "above good-enough" nor "maybe above" (rounding to fp32 is
  // monotone; the margin of 2^-30 covers the 2^-38 guard band): the fast path's first look in launches with the good-enough rule
  const float ge_near_f = (float)(good_enough - ldexp(good_enough < 0.0 ? -good_enough : good_enough, -30));
  auto eps_lo = [](double x) { return x - ldexp(x, -38); };
  auto eps_hi = [](double x) { return x + ldexp(x, -38); };
  struct JobRegs {   // exactly what the LDS loads deliver: nothing is decoded before the job's own iteration (a decode right after
                     // the load would wait for it)
    double c, m;
    unsigned info, group;
    unsigned f4b;      // JobL::f4 | JobL::b << 16
    double e_fit;      // list entry `lane` (lanes >= LM: none)
    int e_off;
    unsigned owner;    // lane owning the entry's offer, 0xFF untouched, 0xFE no entry
    int g_off;         // (GEF) good-enough list entry `lane` (lanes >= LG: none): offer, owner as above
    unsigned g_owner;
  };
  // record + list entry of walk position i of the segment: addresses depend on i only, so the loads of job i+1 are issued a whole turn
  // ahead and nothing waits for them (OPAQUE_V: see gpu_prims.hpp)
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
    r.owner = 0xFEu;
    r.g_off = -1;
    r.g_owner = 0xFEu;
    // (no branch on the lane number in the walk loop, here or below — see take_job for what one costs the whole loop — and no select
    //  BEHIND a load either, which would wait for it on the spot: a lane without an entry loads the "none" record instead)
    {
      const bool has = lane < (unsigned)LM;
      const double* fp = has ? &s_efit[(size_t)ii * LM + lane] : &L.fit_none;
      const int* op = has ? &s_eoff[(size_t)ii * LM + lane] : &L.off_none;
      r.e_fit = *fp;
      r.e_off = *op;
    }
    if constexpr (GEF) {
      const int* gp = (use_ge && lane < (unsigned)LG) ? &s_goff[(size_t)ii * LG + lane] : &L.off_none;
      r.g_off = *gp;
    }
    return r;
  };
  // the owner look-up needs the entry's offer: issued at the end of the turn before the job's own (a commit in between patches it, see below)
  auto load_owner = [&](JobRegs& r) {
    {
      const unsigned char* op = (lane < (unsigned)LM && r.e_off >= 0) ? &s_owner[(unsigned)r.e_off] : &L.owner_none[0];
      r.owner = *op;
    }
    if constexpr (GEF) {
      const unsigned char* op = (use_ge && lane < (unsigned)LG && r.g_off >= 0) ? &s_owner[(unsigned)r.g_off] : &L.owner_none[0];
      r.g_owner = *op;
    }
  };
  // The fast path's result store, by ALL lanes: lane 0 into the result row, the others into a sink.  As `if (lane == 0) store` it is a
  // divergent branch whose join is the block where the fast path's exits meet, and the compiler's uniformity analysis then takes every
  // value that meets there for divergent — the walk position, the count of touched offers, the fast path's verdict itself: the loop
  // became a divergent loop (execution masks, its counters in vector registers, the walker's state copied at every join).
  auto store_result = [&](unsigned pos, int w) {
    int* const p = lane == 0 ? &s_j2o[pos] : &L.sink[lane];
    *p = w;
  };
  // The colbits word of job group g of the window for the lane's offer (0 for a lane without one): one gather from global memory when
  // the walk enters a new group of 64 window positions (at most nwin / 64 times per round).
  auto fetch_col = [&](unsigned g) -> unsigned long long {
    const unsigned v = t_v >= 0 ? (unsigned)t_v : 0u;
    const unsigned long long w = vb.colbits[(size_t)v * MV_JGL + g];
    return t_v >= 0 ? w : 0ull;
  };
  // the walk enters group g of the window: its word from the prefetch if that is the group fetched ahead, and the word of the group
  // after it ordered now (a long window's visited jobs may skip groups: then the word is fetched on the spot)
  auto enter_group = [&](unsigned g) {
    if (__builtin_expect(g == col_next, 1)) {
      t_col = t_coln;
    } else if (nT != 0u) {
      t_col = fetch_col(g);
      WAIT_ALL_MEM();
    }
    cur_g = g;
    col_next = g + 1u < (unsigned)MV_JGL ? g + 1u : g;
    if (nT != 0u) t_coln = fetch_col(col_next);
  };
  // Lane nT becomes the owner of the untouched offer `off` that takes a job of (c, m).  Its record, snapshot state and colbits word come
  // from GLOBAL memory (wave-uniform addresses: every lane reads them, one transaction each, a single round trip for all of them).
  // Branch-free: every lane keeps its own state through selects unless it is the new owner.  As `if (lane == nT) { state = record }`
  // the loads were masked writes into a second set of registers: the compiler kept the walker's state in two homes from then on and
  // moved all of it from one to the other and back in every iteration (~45 v_mov per job on the path of a job that goes to an offer
  // touched before, which never opens a lane).
  auto open_lane = [&](unsigned nl, int off, double jc, double jm) {
    const unsigned v = wave_uniform_u32((unsigned)off);
    const OfferW w = vb.ow[v];  // one cache line (pulled into this XCD's L2 when the segment was staged)
    const unsigned long long colw = vb.colbits[(size_t)v * MV_JGL + cur_g], colwn = vb.colbits[(size_t)v * MV_JGL + col_next];
    const bool me = lane == nl;
    t_v = me ? off : t_v;
    t_oc = me ? w.oc : t_oc;
    t_om = me ? w.om : t_om;
    t_rc = me ? w.rc : t_rc;
    t_rm = me ? w.rm : t_rm;
    t_invc = me ? w.inv_dc : t_invc;
    t_invm = me ? w.inv_dm : t_invm;
    t_k8s = me ? w.k8s : t_k8s;
    t_host = me ? w.host : t_host;
    t_run = me ? w.run_count : t_run;
    t_slack = me ? w.task_slack : t_slack;
    t_ac = me ? w.ac + jc : t_ac;
    t_am = me ? w.am + jm : t_am;
    t_acount = me ? w.acount + 1 : t_acount;
    t_basec = t_rc + t_ac;
    t_basem = t_rm + t_am;
    t_col = me ? colw : t_col;
    t_coln = me ? colwn : t_coln;
    L.ac0[nl] = w.ac;  // (every lane, same value)
    L.am0[nl] = w.am;
    L.acount0[nl] = w.acount;
    unsigned char* const p = me ? &s_owner[v] : &L.sinkb[lane];
    *p = (unsigned char)nl;
  };
  // the lane for a newly touched offer: the next free one, or — all 64 taken — the lane of a DEAD offer, which is retired first: its
  // state goes to global memory now (nothing reads it before the next round), its alive bit is cleared, the owner table says "dead".
  // -> MV_T: none (the round ends).  `nxt`'s owner look-up was issued before: patched here.
  auto alloc_lane = [&](JobRegs& nxt) -> unsigned {
    if (__builtin_expect(nT < (unsigned)MV_T, 1)) return nT;
    if (!can_retire || n_retired >= MV_RETIRE_CAP) return (unsigned)MV_T;
    const bool dead = t_ac + jmin_c > t_oc || t_am + jmin_m > t_om;  // (all 64 lanes own an offer)
    const unsigned long long dm = __ballot(dead);
    if (dm == 0ull) return (unsigned)MV_T;
    const unsigned nl = (unsigned)__ffsll((unsigned long long)dm) - 1u;
    if (lane == nl) {
      st.ac[t_v] = t_ac;
      st.am[t_v] = t_am;
      st.acount[t_v] = t_acount;
      vb.ow[t_v].ac = t_ac;
      vb.ow[t_v].am = t_am;
      vb.ow[t_v].acount = t_acount;
      atomicAnd(&st.alive[(unsigned)t_v >> 6], ~(1ull << ((unsigned)t_v & 63u)));
      s_owner[(unsigned)t_v] = (unsigned char)OWNER_DEAD;
    }
    if (nxt.owner == nl) nxt.owner = OWNER_DEAD;
    if constexpr (GEF) {
      if (nxt.g_owner == nl) nxt.g_owner = OWNER_DEAD;
    }
    ++n_retired;
    wave_sync();
    return nl;
  };
  // The owner lane of a touched offer books a job of (jc, jm) — through selects as well: a branch on the lane number anywhere in the fast
  // path makes its whole region "divergent control flow" for the compiler, which then rebuilds it with flow blocks whose undefined
  // inputs keep the register coalescer from giving the walker's state ONE home (the v_mov trains of open_lane's comment).
  auto take_job = [&](bool me, double jc, double jm) {
    t_ac = me ? t_ac + jc : t_ac;
    t_am = me ? t_am + jm : t_am;
    t_acount = me ? t_acount + 1 : t_acount;
    t_basec = t_rc + t_ac;
    t_basem = t_rm + t_am;
  };
  // publish a placed member of a group whose hosts the staging gathered (fast paths): the chain in HBM (later rounds' evaluation and the
  // general path read it) and the round's log.  ghits = the log entries of the job's group.
  auto publish_group_member = [&](unsigned long long ghits, unsigned gslot, unsigned g, unsigned k, int w_offer, unsigned w_host) {
    const int prev = ghits != 0ull ? wave_read_lane(lg_k, 63 - __clzll((long long)ghits)) : s_glast[gslot];
    if (lane == 0) {
      st_agent(&st.job_to_offer[k], w_offer);
      st_agent(&st.job_prev[k], prev);
      st_agent(&st.group_last[g], (int)k);
    }
    const bool me = lane == n_log;
    lg_group = me ? g : lg_group;
    lg_host = me ? w_host : lg_host;
    lg_k = me ? (int)k : lg_k;
    ++n_log;
  };
  auto store_fail = [&](unsigned pos, unsigned char f) {
    unsigned char* const p = lane == 0 ? &s_fail[pos] : &L.sinkb[lane];
    *p = f;
  };
  // ---- the segments of the round ---------------------------------------------------------------------------------------------
  for (;;) {
  JobRegs cur = load_rec(0);
  load_owner(cur);
  JobRegs nxt = cur;
  WAIT_LDS();  // nothing pending at loop entry either (the loop's own waits sit at the END of its iterations)
  unsigned i = 0;  // walk position in the segment; after the loop: the number of the segment's walk positions done
  // The decoded form of the job in `cur` (wave-uniform values in scalar registers).  Declared by a macro because both loops below
  // need it in their own scope: the walker's state must not flow through a join of the two paths (see the loop comment).
#define WALK_DECODE()                                                                                                            \
  const unsigned cinfo_u = wave_uniform_u32(cur.info), cb_u = wave_uniform_u32(cur.f4b) >> 16;                                    \
  const bool cur_no_zero_fit = (wave_uniform_u32(cur.f4b) & 0xFFFFu) == 0u; /* no offer had zero fitness for this job under S */ \
  const unsigned b = cb_u, k = head + b;                                                                                          \
  /* the job's bit in the colbits words: by window position */                                                                   \
  const unsigned bl = b & 63u;                                                                                                    \
  const double c = cur.c, m = cur.m;                                                                                              \
  const bool grouped = (cinfo_u & JL_GROUPED) != 0;                                                                               \
  const bool job_gpu = (cinfo_u & JL_GPU) != 0;                                                                                   \
  const bool has_group = (cinfo_u & JL_HASGROUP) != 0;                                                                            \
  const unsigned gtype = (cinfo_u >> 18) & 3u;                                                                                    \
  const int nc = (int)(cinfo_u & 0xFFu);                                                                                          \
  const bool t_on = t_v >= 0;                                                                                                     \
  (void)cur_no_zero_fit, (void)k, (void)bl, (void)c, (void)m, (void)grouped, (void)job_gpu, (void)has_group, (void)gtype, (void)nc, (void)t_on
#ifdef COOK_WALK_PROF
#define WALK_PROF_BEGIN() const unsigned long long pk0 = __builtin_readcyclecounter()
#define WALK_END(cat)                                                \
  do {                                                               \
    const unsigned long long pk1_ = __builtin_readcyclecounter();    \
    ctl.prof_cyc[cat] += pk1_ - pk0;                                 \
    ctl.prof_cnt[cat] += 1u;                                         \
  } while (0)
#else
#define WALK_PROF_BEGIN() ((void)0)
#define WALK_END(cat) ((void)0)
#endif
  // TWO loops: the inner one holds nothing but the fast path and runs from job to job while that settles them; a job it cannot settle
  // leaves it for one turn of the outer loop's general path.  As ONE loop body (fast path, else general path, one latch) every variable
  // of the walker's state reached the latch through a join of the two paths, and the compiler resolved those joins with register
  // copies: ~70 v_mov per job on the fast path (state -> temporaries -> state), a quarter of its time.
  // The fast loop is unrolled by two with the job records in two register sets that swap roles (`cur` / `nxt` of one turn are `nxt` /
  // `cur` of the next): the look-ahead costs no register rotation.  fast_turn = one job: 0 = settled, on to the next; 1 = the segment
  // is used up; 2 = not settled (or settled by an untouched offer: open_off), this job leaves the loop.
  int open_off = -1;  // >= 0: the fast path gave the job to an untouched offer (committed below the loop)
  bool open_group = false;
  auto fast_turn = [&](JobRegs& cur, JobRegs& nxt) __attribute__((always_inline)) -> int {
      if (__builtin_expect(i >= n_eff, 0)) return 1;
      EMU_SITE("resolve: walk loop");
      WALK_PROF_BEGIN();
      nxt = load_rec(i + 1);  // in flight while job i is decided (its owner look-up follows at the end of this turn, when the entry's offer is there)
      WALK_DECODE();
      if (__builtin_expect((b >> 6) != cur_g, 0)) enter_group(b >> 6);  // next word of the columns
    // ======== FAST PATH ======================================================================================================
    // self-contained: decision AND commit, then straight on to the next job (its control flow never joins the general path's).
    // Two instantiations: plain jobs, and members of unique / unconstrained groups whose hosts-to-avoid the staging gathered
    // (JL_GSLOT) — kept apart so that the group code costs the plain jobs nothing.
    const unsigned gslot = (cinfo_u >> JL_GSLOT_SHIFT) & JL_GSLOT_NONE;
    auto fast_path = [&](auto group_tag) -> bool {
      constexpr bool GROUP = decltype(group_tag)::value;
      const unsigned g = GROUP ? wave_uniform_u32(cur.group) : 0xFFFFFFFFu;  // (jobs without a group never read the word)
      (void)g;
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
      auto publish_member = [&](int w_offer, unsigned w_host) { publish_group_member(ghits, gslot, g, k, w_offer, w_host); };
      const double a1 = (t_basec + c) * t_invc, a2 = (t_basem + m) * t_invm;
      const double fa = (a1 + a2) * 0.5;
      const bool cand = res_ok && con_ok;
      // fp32 image of the approximate fitness: monotone in fa; a candidate whose approximation cannot be trusted for ordering
      // (negative terms, zero, below fp32's normal range) takes +inf, which sends the job to the general path
      const bool sane = a1 >= 0.0 && a2 >= 0.0 && fa > 0x1p-100;
      const float kf = cand ? (sane ? (float)fa : __int_as_float(0x7F800000)) : 0.0f;
      // (best-fit launches reduce here; launches with the good-enough rule only once that rule has left the job undecided — most of
      //  their jobs go to an offer above the threshold, and a reduction they never look at is ~25 instructions of the walking wave)
      float mx = 0.0f;
      if constexpr (!GEF) mx = wave_max_f32(kf);
      // first untouched entry of the list: the best untouched offer under S (a touched entry that is still feasible and sits in
      // front of it only gained fitness: it beats this one in the comparison below, so "first untouched" is all the list has to give)
      const unsigned long long untouched_mask = __ballot(cur.owner == 0xFFu);
      double u_fit = -1.0;
      int u_off = -1;
      if (__builtin_expect(untouched_mask != 0ull, 1)) {
        const int qs = __ffsll((unsigned long long)untouched_mask) - 1;
        u_fit = wave_read_lane_f64(cur.e_fit, qs);
        u_off = wave_read_lane(cur.e_off, qs);
      }
      int f_lane = -1;      // >= 0: that touched offer wins
      bool f_new = false;   // the untouched offer u_off wins
      bool ge_done = false;  // (GEF) the good-enough rule settled the job
      if constexpr (GEF) {
        if (use_ge) {
          // scheduler.clj:2312-2314: the first offer in array order whose fitness exceeds good-enough wins outright.  Untouched offers keep
          // the fitness they had under S, so their part of that order is the job's good-enough list; a touched offer is above the
          // threshold for sure when its approximate fitness clears it with a margin, below for sure the other way round — anything in
          // between (or a list that may not reach far enough) goes to the general path and its exact divisions
          // (first a look through the fp32 image the best-fit reduction uses anyway: while no touched offer comes near the threshold — the
          //  filling phase of a cycle: two thirds of the walked jobs of a C4 pool at 0.8 — the exact tests and the index reduction are skipped)
          int tg = 0x7FFFFFFF;  // lowest offer index among the touched offers above the threshold
          int tl = -1;          // ... and its lane
          if (__ballot(kf >= ge_near_f) != 0ull) {
            const bool above = cand && sane && eps_lo(fa) > good_enough;
            const bool maybe = cand && !above && (!sane || eps_hi(fa) > good_enough);
            if (__builtin_expect(__any(maybe), 0)) return false;
            const unsigned long long above_mask = __ballot(above);
            if (above_mask != 0ull) {
              if ((above_mask & (above_mask - 1ull)) == 0ull) {  // one touched offer above the threshold — the common case — needs no reduction
                tl = __ffsll((unsigned long long)above_mask) - 1;
                tg = wave_read_lane(t_v, tl);
              } else {
                const unsigned tkey = above ? 0x7FFFFFFFu - (unsigned)t_v : 0u;
                const unsigned tmx = wave_max_u32(tkey);
                tg = 0x7FFFFFFF - (int)tmx;
                tl = __ffsll((unsigned long long)__ballot(above && tkey == tmx)) - 1;
              }
            }
          }
          const int ng = (int)((cinfo_u >> 8) & 0xFFu);
          int ge_pick = 0x7FFFFFFF;
          if (ng != 0) {
            const unsigned long long gun = __ballot((int)lane < ng && cur.g_owner == 0xFFu);
            if (__builtin_expect(gun != 0ull, 1)) {
              const int q = __ffsll((unsigned long long)gun) - 1;
              ge_pick = wave_read_lane(cur.g_off, q);
            } else if (cinfo_u & JL_GTRUNC) {
              // every listed offer is touched by now: untouched ones above the threshold may exist beyond the list, below the best touched index or not
              if (tg > wave_read_lane(cur.g_off, ng - 1)) return false;
            }
          } else if (cinfo_u & JL_GTRUNC) {
            return false;
          }
          if (tg < ge_pick) {
            f_lane = tl;
            ge_done = true;
          } else if (ge_pick != 0x7FFFFFFF) {
            f_new = true;
            u_off = ge_pick;
            ge_done = true;
          }  // else: nobody above the threshold — best fit among what is below it
        }
      }
      if constexpr (GEF) {
        if (!ge_done) mx = wave_max_f32(kf);
      }
      if (ge_done) {
        // (decided above)
      } else if (__builtin_expect(mx == 0.0f, 0)) {  // no touched offer can take the job
        f_new = u_off >= 0;  // else: unmatched or list exhausted -> general path
      } else if (__builtin_expect(mx < __int_as_float(0x7F800000), 1)) {
        const unsigned long long near = __ballot(kf >= mx * (1.0f - 0x1p-20f));
        if (__builtin_expect((near & (near - 1ull)) == 0ull, 1)) {  // one touched offer clearly ahead of the other touched ones
          const int wl = __ffsll((unsigned long long)near) - 1;
          const double fw = wave_read_lane_f64(fa, wl);
          if (__builtin_expect(u_off < 0, 0)) {
            // no untouched entry: fine unless the list is truncated and none of its entries is still a candidate (then better
            // untouched offers may exist beyond the list: exhausted, general path)
            bool ok = COOK_L_COMPLETE();
            if (!ok) {
              const unsigned long long cand_mask = __ballot(cand);
              const bool e_live = cur.owner < (unsigned)MV_T && ((cand_mask >> (cur.owner & 63u)) & 1ull);
              ok = __any(e_live);
            }
            if (ok) f_lane = wl;
          } else if (__builtin_expect(eps_lo(fw) > u_fit, 1)) {
            f_lane = wl;
          } else if (eps_hi(fw) < u_fit) {
            f_new = true;
          }
        }
      }
      // the booking by the winner's lane, on EVERY way out of here (f_lane = -1: no lane): as a statement of the branch below the booked
      // fields met their unbooked selves where the fast path's exits join, and were moved between two sets of registers for it
      take_job((int)lane == f_lane, c, m);
      if (__builtin_expect(f_lane >= 0, 1)) {  // an offer touched earlier in this round takes the job
        const int w = wave_read_lane(t_v, f_lane);
        store_result(i, w);  // (s_fail[i] = 0 since the staging)
        if constexpr (GROUP) publish_member(w, (unsigned)wave_read_lane((int)t_host, f_lane));
        WALK_STAT(3, 1);
        WALK_STAT(8, 1);
        WALK_END(GROUP ? 4u : ((GEF && ge_done) ? 6u : 1u));
        return true;
      }
      if (__builtin_expect(f_new && (nT < (unsigned)MV_T || can_retire), 1)) {  // an untouched offer: a lane takes ownership — outside this loop (see below)
        open_off = u_off;
        open_group = GROUP;
      }
      return false;
    };
    bool fast_done = false;
    if (GEF || !(good_enough < 1.0)) {
      if (__builtin_expect(!(cinfo_u & (JL_GROUPED | JL_HASGROUP | JL_XRES)), 1))
        fast_done = fast_path(std::false_type{});
      else if (gslot != JL_GSLOT_NONE && n_log < (unsigned)COOK_WAVE)
        fast_done = fast_path(std::true_type{});
    }
    load_owner(nxt);  // (before a commit of the paths below: they patch it)
    if (__builtin_expect(!fast_done, 0)) {
      WALK_END(7u);  // (measurement build: the turn's share of a job that the paths below the loop finish)
      return 2;
    }
    WAIT_LDS_BUT_2();  // the record of the next job has arrived (see common.hpp); the result store and the owner look-up may still fly
    ++i;
    return 0;
  };
  for (;;) {
    bool walk_over = false;
    open_off = -1;
    open_group = false;
    for (;;) {  // ---- fast loop ----
      int r = fast_turn(cur, nxt);
      if (__builtin_expect(r == 0, 1)) r = fast_turn(nxt, cur) | 4;  // (bit 2: the register sets are swapped)
      if (__builtin_expect((r & 3) == 0, 1)) continue;
      walk_over = (r & 3) == 1;
      if (r & 4) {  // back to `cur` = this job, `nxt` = the next one
        const JobRegs t = cur;
        cur = nxt;
        nxt = t;
      }
      break;
    }  // ---- fast loop ----
    if (__builtin_expect(walk_over, 0)) break;
    // (prefetches and the column word of the job in `cur` are in place: the fast loop's turn for it issued them)
    WALK_PROF_BEGIN();
    unsigned pcat = 0;
    (void)pcat;
    WALK_DECODE();
    if (__builtin_expect(open_off >= 0, 1)) {
      // ---- the fast path's other verdict: an untouched offer takes the job and the next free lane becomes its owner.  Committed HERE,
      // outside the fast loop: the fields open_lane writes (an offer's totals, reciprocals, host ...) are then loop-invariant inside
      // it, and only there does the compiler keep them in ONE set of registers without moving them about
      const unsigned g = open_group ? wave_uniform_u32(cur.group) : 0xFFFFFFFFu;
      const unsigned gslot = (cinfo_u >> JL_GSLOT_SHIFT) & JL_GSLOT_NONE;
      const unsigned long long ghits = open_group ? __ballot(lane < n_log && lg_group == g) : 0ull;
      const unsigned nl = alloc_lane(nxt);
      if (__builtin_expect(nl == (unsigned)MV_T, 0)) {
        stop = 2;  // no lane to track a new touched offer: end the round before this job (the fast path booked nothing for it)
        resolved = b;
        break;
      }
      open_lane(nl, open_off, c, m);
      WAIT_ALL_MEM();
      if (open_group) publish_group_member(ghits, gslot, g, k, open_off, (unsigned)wave_read_lane((int)t_host, (int)nl));
      // the owner look-up of the next job was issued before this commit: patch it
      if (nxt.owner == 0xFFu && nxt.e_off == open_off) nxt.owner = nl;
      if constexpr (GEF) {
        if (nxt.g_owner == 0xFFu && nxt.g_off == open_off) nxt.g_owner = nl;
      }
      nT += nT < (unsigned)MV_T ? 1u : 0u;
      store_result(i, open_off);
      wave_sync();  // the owner table update is visible to the whole wave before the next look-up reads it
      WALK_STAT(4, 1);
      WALK_STAT(8, 1);
      WALK_END(open_group ? 4u : 2u);
      WAIT_LDS_BUT_LAST();
      cur = nxt;
      ++i;
      continue;
    }
    const unsigned g = has_group ? wave_uniform_u32(cur.group) : 0xFFFFFFFFu;
    int win = -1, win_lane = -1;  // win_lane >= 0: a touched offer wins; else win >= 0: that untouched offer
    bool need_exact = false;
    bool exhausted = false;  // the job's list ran out: the round ends here
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
        // (the loaded word through a scalar register: to the compiler a load through a generic pointer is a per-lane value, the branch on
        //  it a divergent exit of the walk loop, and everything the loop carries — walk position, touched count — divergent with it)
        if (gtype >= 2 && (int)wave_uniform_u32((unsigned)ld_agent(&st.group_last[g])) >= (int)head) {
          stop = 3;
          resolved = b;
          break;
        }
        if (t_v >= 0) gok = group_pass_dev(vb.in_dev, st, jj, (unsigned)t_v);
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
      need_exact = (good_enough < 1.0) || __any(cand && !sane);
      const unsigned long long cand_mask = __ballot(cand);
      double u_fit = -1.0;     // best untouched candidate: fitness under S, offer
      int u_off = -1;
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
        const bool e_untouched = cur.owner == 0xFFu;
        const bool e_live = cur.owner < (unsigned)MV_T && ((cand_mask >> (cur.owner & 63u)) & 1ull);
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
          }
        }
        // --- best touched candidate ----------------------------------------------------------------------------------------------
        if (!need_exact) {
          if (cand_mask == 0ull) {
            win = u_off;
            decided = true;
          } else {
            const unsigned long long key = cand ? (unsigned long long)__double_as_longlong(fa) : 0ull;  // positive doubles
            const double mx = __longlong_as_double((long long)wave_max_u64(key));
            const unsigned long long near = __ballot(cand && fa >= eps_lo(mx));
            if ((near & (near - 1ull)) == 0ull) {  // one touched offer clearly ahead of the other touched ones
              if (u_off < 0 || eps_lo(mx) > u_fit) {
                win_lane = __ffsll((unsigned long long)near) - 1;
                decided = true;
              } else if (eps_hi(mx) < u_fit) {
                win = u_off;
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
          const bool e_live2 = cur.owner < (unsigned)MV_T && ((feas_mask >> (cur.owner & 63u)) & 1ull);
          const unsigned long long settle2 = __ballot(e_untouched || e_live2);
          if (settle2 == 0ull && COOK_L_TRUNC()) {
            exhausted = true;
            break;
          }
          u_fit = -1.0;
          u_off = -1;
          if (settle2 != 0ull) {
            const int qs = __ffsll((unsigned long long)settle2) - 1;
            if ((untouched_mask >> qs) & 1ull) {
              u_fit = wave_read_lane_f64(cur.e_fit, qs);
              u_off = wave_read_lane(cur.e_off, qs);
            }
          }
          // good-enough path: lowest offer index with fitness > good-enough (scheduler.clj:2312-2314)
          int ge_pick = 0x7FFFFFFF, ge_lane = -1;
          if (good_enough < 1.0) {
            if constexpr (!GE) {  // (launches for good-enough-fitness < 1 are GE launches: the host sees to it)
              exhausted = true;
              break;
            } else {
              const int ng = (int)((cinfo_u >> 8) & 0xFFu);
              int ge_off = -1;
              unsigned g_owner = 0xFEu;
              if ((int)lane < ng) {
                ge_off = s_goff[(size_t)i * LG + lane];
                g_owner = ge_off >= 0 ? (unsigned)s_owner[(unsigned)ge_off] : 0xFEu;
              }
              const unsigned long long gun = __ballot(g_owner == 0xFFu);
              int last_idx = -1;
              if (ng > 0) last_idx = wave_read_lane(ge_off, ng - 1);
              if (gun != 0ull) {
                const int q = __ffsll((unsigned long long)gun) - 1;
                ge_pick = wave_read_lane(ge_off, q);
              }
              // lowest-index touched offer that is feasible with fitness > good-enough
              const unsigned long long tkey = (t_feas && pe_fit > good_enough)
                                                  ? (((unsigned long long)(unsigned)(0x7FFFFFFF - t_v) << 32) | (unsigned long long)lane)
                                                  : 0ull;
              const unsigned long long tmx = feas_mask != 0ull ? wave_max_u64(tkey) : 0ull;
              const int tg = tmx != 0ull ? 0x7FFFFFFF - (int)(unsigned)(tmx >> 32) : 0x7FFFFFFF;
              if (gun == 0ull && (cinfo_u & JL_GTRUNC) && tg > last_idx) {
                // untouched good-enough offers beyond the list may exist with an index below the best touched one
                exhausted = true;
                break;
              }
              if (tg < ge_pick) {
                ge_pick = tg;
                ge_lane = (int)(unsigned)(tmx & 63ull);
              }
            }
          }
          if (ge_pick != 0x7FFFFFFF) {
            if (ge_lane >= 0) {
              win_lane = ge_lane;
            } else {
              win = ge_pick;
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
            } else if (best_lane >= 0) {
              win_lane = best_lane;
            }
          }
        }
      } while (0);
    }
    if (exhausted) {  // end the round here: the next round evaluates the rest of the window afresh
      stop = 1;
      resolved = b;
      break;
    }
    // --- commit --------------------------------------------------------------------------------------------------------------
    if (win_lane >= 0) WALK_STAT(3, 1);
    else if (win >= 0) WALK_STAT(4, 1);
    else WALK_STAT(5, 1);
    WALK_STAT_PREV_LANE(i, win_lane, win, nT);
#ifdef COOK_WALK_PROF
    pcat = grouped ? 4u : (win >= 0 || win_lane >= 0 ? 5u : 3u);
#endif
    unsigned new_lane = (unsigned)MV_T;  // the lane an untouched winner was given
    if (win_lane >= 0) {  // an offer touched earlier in this round takes the job
      take_job((int)lane == win_lane, c, m);
      win = wave_read_lane(t_v, win_lane);
    } else if (win >= 0) {  // an untouched offer: a lane takes ownership
      new_lane = alloc_lane(nxt);
      if (new_lane == (unsigned)MV_T) {
        stop = 2;  // no lane to track a new touched offer: end the round before this job
        resolved = b;
        break;
      }
      open_lane(new_lane, win, c, m);
      WAIT_ALL_MEM();
      // the owner look-up of the next job was issued before this commit: patch it
      if (nxt.owner == 0xFFu && nxt.e_off == win) nxt.owner = new_lane;
      if constexpr (GEF) {
        if (nxt.g_owner == 0xFFu && nxt.g_off == win) nxt.g_owner = new_lane;
      }
      nT += nT < (unsigned)MV_T ? 1u : 0u;
      wave_sync();  // the owner table update is visible to the whole wave before the next look-up reads it
    }
    if (win >= 0) {
      if (cinfo_u & JL_XRES) {  // the offer's owner lane books the job's ports / named scalars
        const int ol = win_lane >= 0 ? win_lane : (int)new_lane;
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
      store_result(i, win);
      store_fail(i, 0);
      if (g != 0xFFFFFFFFu) {
        if (lane == 0) {  // cotasks look each other up through HBM (group_pass): publish at once
          st_agent(&st.job_to_offer[k], win);
          st_agent(&st.job_prev[k], ld_agent(&st.group_last[g]));
          st_agent(&st.group_last[g], (int)k);
        }
        wave_sync();  // later cotasks of this wave read what lane 0 just published
        // ... and the round's log, for the members that take the fast path (the owner lane of the winning offer knows its host)
        const int ol = win_lane >= 0 ? win_lane : (int)new_lane;
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
          const double ac0 = L.ac0[lane], am0 = L.am0[lane];  // the offer's state as the round began
          const int acount0 = L.acount0[lane];
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
          if (ac0 + c > t_oc || am0 + m > t_om || x0_fail) {
            p0 = 1u;
          } else {
            bool ok = ((t_col >> bl) & 1ull) != 0 && acount0 < t_slack;
            if (job_gpu && t_k8s && t_run + acount0 != 0) ok = false;
            if (ok && grouped) {
              MatchState st0 = st;
              st0.cutoff = (int)head;
              ok = group_pass_dev(vb.in_dev, st0, jj, (unsigned)t_v);
            }
            if (!ok) {
              p0 = 2u;
            } else {
              const double f0 = ((t_rc + ac0 + c) / (t_oc + t_rc) + (t_rm + am0 + m) / (t_om + t_rm)) / 2.0;
              if (!(f0 > 0.0)) p0 = 4u;
            }
          }
        }
        d1 = __popcll(__ballot(t_on && (pe_bits & 1u))) - __popcll(__ballot(t_on && (p0 & 1u)));
        d2 = __popcll(__ballot(t_on && (pe_bits & 2u))) - __popcll(__ballot(t_on && (p0 & 2u)));
        d4 = __popcll(__ballot(t_on && (pe_bits & 4u))) - __popcll(__ballot(t_on && (p0 & 4u)));
      }
      // ... and the RETIRED offers of the round: each fails on resources now (dead), so class 1 is not empty; what class each was in
      // under S for THIS job nobody kept, so classes 2 / 4 are only certain when the snapshot count is zero (no retired offer can have
      // been in the class) or larger than every offer that may have left it.  Anything in between needs a fresh snapshot: the round
      // ends before this job (rare: unmatched jobs that are walked at all are, and only after a round's 65th offer).
      if (n_retired != 0u) {
        const int hi2 = (int)jl.f2 + d2, hi4 = (int)jl.f4 + d4;  // (upper bounds: the retired offers can only take away)
        const bool amb2 = jl.f2 != 0 && hi2 > 0 && hi2 - (int)n_retired <= 0, amb4 = jl.f4 != 0 && hi4 > 0 && hi4 - (int)n_retired <= 0;
        if (amb2 || amb4) {
          stop = 5;
          resolved = b;
          break;
        }
        d1 += (int)n_retired;
      }
      const unsigned bits = (((int)jl.f1 + d1) > 0 ? 1u : 0u) | (((int)jl.f2 + d2) > 0 ? 2u : 0u) | (((int)jl.f4 + d4) > 0 ? 4u : 0u);
      store_result(i, -1);  // (branch-free like the fast path's: this is the last statement before the paths of the iteration meet)
      store_fail(i, (unsigned char)(bits ? bits : 8u));
    }
    WALK_END(pcat);
    WAIT_ALL_MEM();
    cur = nxt;
    ++i;
  }
#undef WALK_DECODE
#undef WALK_PROF_BEGIN
#undef WALK_END
  // ---- the segment is over (used up, or the round stopped inside it): flush its results ----------------------------------------
  wave_sync();
  // the segment's counts, read off the results (counters carried through the walk loop cost it instructions in every job): matched
  // jobs, "the head of the queue was matched", walked jobs with a truncated list (incl. the job the round stopped at, if any)
  {
    const unsigned n_seen = i < n_eff ? i + 1u : i;
    for (unsigned x0 = 0; x0 < n_seen; x0 += COOK_WAVE) {
      const unsigned x = x0 + lane;
      const bool got = x < i && s_j2o[x] >= 0;
      matched += (unsigned)__popcll(__ballot(got));
      if (__ballot(got && head + (unsigned)s_job[x < n_seen ? x : 0u].b == 0u) != 0ull) head_matched = 1;
      n_trunc += (unsigned)__popcll(__ballot(x < n_seen && (s_job[x < n_seen ? x : 0u].info & JL_TRUNC) != 0u));
    }
  }
  for (unsigned x = lane; x < i; x += COOK_WAVE) {  // the walked jobs (the others were settled, and written, in the parallel phase)
    const unsigned bx = s_job[x].b;
    st.job_to_offer[head + bx] = s_j2o[x];
    if (st.fail_code) st.fail_code[head + bx] = s_fail[x];
  }
  // The next segment of the same window, if the round did not stop and there is one: the lists of its jobs were computed against the
  // same snapshot, the walker keeps its lanes (the offers it touched are exactly the ones whose list entries it re-evaluates), so the
  // walk simply goes on — the workgroup stages the segment, no launch and no evaluation in between.
  if (stop != 0 || seg_lo + n_eff >= n_list) break;
  seg_lo += n_eff;
  if (lane == 0) {
    L.seg_lo = seg_lo;
    L.cmd = 1;
    s_ngslots = 0;
  }
  {
    const unsigned long long ts0 = cook_ticks();
    EMU_SITE("resolve: walker asks for the next segment");
    __syncthreads();  // (releases the other waves into stage_segment)
    n_eff = stage_segment(seg_lo);
    if (lane == 0) L.cmd = 0;  // (read by the others only behind the next barrier)
    ++n_segments;
    t_stage += cook_ticks() - ts0;
  }
  }  // ---- segments ----
  // ---- the round is over: release the other waves, write the touched offers' state back and publish the new head ------------------
  if (lane == 0) L.cmd = 0;
  EMU_SITE("resolve: walker done");
  __syncthreads();
  if (t_v >= 0) {
    st.ac[t_v] = t_ac;
    st.am[t_v] = t_am;
    st.acount[t_v] = t_acount;
    vb.ow[t_v].ac = t_ac;
    vb.ow[t_v].am = t_am;
    vb.ow[t_v].acount = t_acount;
    if (t_ac + st.jmin[0] > t_oc || t_am + st.jmin[1] > t_om)  // full for every job of this call, for good
      atomicAnd(&st.alive[(unsigned)t_v >> 6], ~(1ull << ((unsigned)t_v & 63u)));
  }
  if (lane == 0)
    resolve_finish(ctl, vb, head, nwin, resolved, stop, matched, head_matched, nT + n_retired, n_list, n_segments, n_trunc, t_stage,
                   cook_ticks() - tk0, L.dbg_h);
}

template <bool GE>
__global__ void __launch_bounds__(MV_RTHREADS) match_resolve2(MatchState st, V2Buf vb) {
  __shared__ __attribute__((aligned(16))) char lds[MV_RLDS_BYTES];
  resolve_round<GE>(lds, st, vb);
}

// ---- several pools in one launch ----------------------------------------------------------------------------------------------
// A rank that holds more pools than the GPU runs launch chains at full speed (about four) places them in LOCKSTEP: the
// three launches of a round with blockIdx.z = pool, every pool on its own WinCtl.  Up to MV_PACK pools travel IN the kernel
// arguments (PoolPack: the compiler sees kernel-argument loads — scalar, uniform — where a context record in memory gives it generic
// pointers); more than that read their contexts from a device array.
struct PoolCtx {
  MatchIn in;
  MatchState st;
  V2Buf vb;
};
template <bool GE>
__global__ void __launch_bounds__(COOK_WAVE* MV_EW) COOK_EVAL_OCCUPANCY match_eval2_multi(const PoolCtx* __restrict__ ctx) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(EvalLds<GE>)];
  const PoolCtx& c = ctx[blockIdx.z];
  if (blockIdx.x >= c.vb.C) return;
  eval_block<GE>(lds, c.in, c.st, c.vb, c.vb.ctl->head, c.vb.ctl->wcur, blockIdx.x, blockIdx.y, gridDim.y);
}
template <bool GE>
__global__ void __launch_bounds__(COOK_WAVE* MV_MW) match_merge2_multi(const PoolCtx* __restrict__ ctx) {
  const PoolCtx& c = ctx[blockIdx.z];
  merge_block<GE>(c.in, c.vb);
}
template <bool GE>
__global__ void __launch_bounds__(MV_RTHREADS) match_resolve2_multi(const PoolCtx* __restrict__ ctx) {
  __shared__ __attribute__((aligned(16))) char lds[MV_RLDS_BYTES];
  const PoolCtx& c = ctx[blockIdx.z];
  resolve_round<GE>(lds, c.st, c.vb);
}
constexpr int MV_PACK = 4;
template <int N>
struct PoolPack {
  PoolCtx c[N];
};
template <bool GE, int N>
__global__ void __launch_bounds__(COOK_WAVE* MV_EW) COOK_EVAL_OCCUPANCY match_eval2_pack(const PoolPack<N> p) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(EvalLds<GE>)];
  const PoolCtx& c = p.c[blockIdx.z];
  if (blockIdx.x >= c.vb.C) return;
  eval_block<GE>(lds, c.in, c.st, c.vb, c.vb.ctl->head, c.vb.ctl->wcur, blockIdx.x, blockIdx.y, gridDim.y);
}
template <bool GE, int N>
__global__ void __launch_bounds__(COOK_WAVE* MV_MW) match_merge2_pack(const PoolPack<N> p) {
  const PoolCtx& c = p.c[blockIdx.z];
  merge_block<GE>(c.in, c.vb);
}
template <bool GE, int N>
__global__ void __launch_bounds__(MV_RTHREADS) match_resolve2_pack(const PoolPack<N> p) {
  __shared__ __attribute__((aligned(16))) char lds[MV_RLDS_BYTES];
  const PoolCtx& c = p.c[blockIdx.z];
  resolve_round<GE>(lds, c.st, c.vb);
}

// ---- served walkers: the pools of a rank DECOUPLED -------------------------------------------------------------------------------
// The lockstep launches above make every pool of a chain wait for the slowest walk of the round and for the evaluation of all its
// neighbours.  Here every pool has ONE persistent walker workgroup — match_walkers, one launch per match of the whole rank, on a
// stream of its own — that runs resolve_round after resolve_round; between two rounds it posts "my next window wants evaluating"
// (ServeSlot::req) and waits for "lists ready" (ServeSlot::ready).  A second stream carries SERVE ITERATIONS — match_serve_eval +
// match_serve_merge, the same evaluation and merge as above — for whichever pools had a request open when the iteration was put
// together (the LATCH: taken by the last workgroup of the previous iteration's merge, so that every workgroup of an iteration
// agrees on the pools it serves).  That workgroup also publishes the iteration's results and then WAITS (bounded) for the next
// request, so the chain of serve launches is paced by the walkers: two streams per GPU, whatever the number of pools.
//
// Hand-offs (MI355X_MICROARCH.md, inter-workgroup visibility): walker -> server: plain stores, agent_release(), relaxed agent store
// of req; the latch reads req with agent loads and the NEXT launch's kernel-start acquire makes the walker's stores visible to its
// workgroups on every XCD.  server -> walker: every merge workgroup releases before it takes its arrival ticket, the last arriver
// stores ready, the walker polls it, acquires once, and reads with plain loads.  Every wait is bounded: a walker that is not served
// within `spin_ticks` raises ServeCtl::error and leaves, the pool's state is consistent (the last finished round), and the host
// finishes the match with lockstep launches.
//
// STEPPING form (spin_ticks == 0; the SIMT emulator of the test suite, which runs one launch at a time, and COOK_SERVE_STEP=1 on the
// GPU): nothing waits — a walker that finds its window not served yet returns, and the host alternates walker launches, latch
// launches and serve iterations.  Same kernels, same words, same results.
struct alignas(128) ServeSlot {  // per pool; the walker's words and the server's on lines of their own
  unsigned req;   // [walker -> server] windows asked for so far (the first one by the host: 1)
  unsigned done;  // [walker -> server] 1 = every job of the pool is resolved, 2 = the walker gave up (ServeCtl::error)
  // the walker's own account (100 MHz ticks; COOK_SERVE_TRACE=1 prints it): from posting a request to seeing its lists, and from the end
  // of a round to the posting of the next request (drain, barrier, L2 write-back)
  unsigned long long wait_ticks, post_ticks;
  unsigned waits, pad0[25];
  unsigned ready;  // [server -> walker] windows served so far
  unsigned claim;  // [server <-> server] the last request of this pool that a server has taken (dynamic assignment: whichever latch sees a
                   // request first takes it with a compare-and-swap from req - 1 to req)
  unsigned pad1[30];
};
constexpr unsigned MV_SERVE_MAX = 16;  // pools per served call
// What ONE serve iteration works on: the pools that had a request open when the iteration was put together, the request numbers it
// serves, and the arrival count that completes its merge.  Two of them, used in turn (iteration `it` reads latch[it & 1], its last
// merge workgroup writes latch[(it + 1) & 1]): workgroups of an iteration that START late — behind the latch, which happens when another
// server's evaluation holds the chip's wave slots — still find the iteration's own list.  (With one list a late workgroup read the NEXT
// iteration's pools, merged windows nobody had evaluated into lists a walker was reading, and took an arrival ticket it was not counted
// for: profiles/r05i_probe.txt.)
struct ServeLatch {
  unsigned n;
  unsigned ticket_target;  // cumulative: the arrival counter is never reset
  unsigned pool[MV_SERVE_MAX], seq[MV_SERVE_MAX];
};
// one per SERVER (a stream of serve iterations): it serves the pools first, first + stride, ... (n_pools of them).  On cache lines of
// its own, the arrival counter on another.
struct alignas(256) ServeCtl {
  unsigned n_pools, pool_first, pool_stride;
  unsigned claim_max;     // > 0: DYNAMIC assignment — this server looks at every pool of the call (first 0, stride 1) and takes up to claim_max
                          // open requests per iteration, first come first served among the servers (ServeSlot::claim); 0: its own pools only
  unsigned dbg_fence;     // (diagnostics, COOK_SERVE_FENCE=1) every hand-off with full agent-scope fences by every workgroup
  unsigned dbg_delay[3];  // (diagnostics) 100 MHz ticks to wait [0] before publishing, [1] between seeing ready and the acquire, [2] behind the acquire
  ServeLatch latch[2];
  unsigned served[MV_SERVE_MAX];  // = ServeSlot::ready of every pool (the latch's own copy)
  unsigned all_done;              // no walker is left
  unsigned error;                 // a walker gave up
  unsigned iterations, empty_iterations, pools_served;  // statistics
  unsigned long long wait_ticks;  // 100 MHz ticks the latch spent waiting for a request
  unsigned long long formed_tick, busy_ticks;  // when the running iteration's list was put together; ticks from there to the publishing of its results (iterations with work)
  alignas(128) unsigned ticket;   // arrivals of merge workgroups so far (agent-scope atomics only; zeroed by the host)
};
struct alignas(128) ServeHost {  // page-locked host memory, written by the latch with system-scope stores, polled by the host
  unsigned iter_done;  // serve iterations finished
  unsigned all_done, error, pad;
};

// the latch of iteration `it` (one wave): publish what the iteration served, then put the next iteration together
static __device__ __forceinline__ void serve_latch(ServeCtl* sc, ServeSlot* slots, ServeHost* host, unsigned long long poll_ticks, unsigned it) {
  const unsigned lane = lane_id();
  const ServeLatch& cur = sc->latch[it & 1u];
  ServeLatch& nxt = sc->latch[(it + 1u) & 1u];
  const unsigned n = wave_uniform_u32(sc->n_pools), nl = wave_uniform_u32(cur.n);
  const unsigned my_pool = wave_uniform_u32(sc->pool_first) + lane * wave_uniform_u32(sc->pool_stride);  // lane = the server's lane-th pool
  unsigned mine = lane < n ? sc->served[lane] : 0u;  // (the previous latch's values)
  for (unsigned x = 0; x < nl; ++x) {                  // ... brought up to date from the iteration that just ran, without a trip through memory
    const unsigned p = wave_uniform_u32(cur.pool[x]), q = wave_uniform_u32(cur.seq[x]);
    if (my_pool == p) mine = q;
  }
  if (lane < n) sc->served[lane] = mine;
  if (sc->dbg_delay[0] != 0u) {
    const unsigned long long d0 = cook_ticks();
    while (cook_ticks() - d0 < sc->dbg_delay[0]) SPIN_PAUSE_FAR();
  }
  if (lane < nl) st_agent(&slots[cur.pool[lane]].ready, cur.seq[lane]);
  const unsigned long long t0 = cook_ticks();
  if (lane == 0 && nl != 0u && sc->formed_tick != 0ull) sc->busy_ticks += t0 - sc->formed_tick;
  unsigned rq = 0, dn = 0;
  unsigned long long pend, alive;
  const unsigned claim_max = wave_uniform_u32(sc->claim_max);
  for (;;) {
    unsigned cl = mine;
    if (lane < n) {
      rq = ld_agent(&slots[my_pool].req);
      dn = ld_agent(&slots[my_pool].done);
      if (claim_max != 0u) cl = ld_agent(&slots[my_pool].claim);
    }
    alive = __ballot(lane < n && dn == 0u);
    pend = __ballot(lane < n && dn == 0u && rq != cl);
    if (claim_max != 0u && pend != 0ull) {  // dynamic: take what nobody has taken yet, at most claim_max of them (the others are some other server's)
      const bool mine_to_try = ((pend >> lane) & 1ull) != 0ull && (unsigned)__popcll(pend & ((1ull << lane) - 1ull)) < claim_max;
      bool won = false;
      if (mine_to_try) won = atomicCAS(&slots[my_pool].claim, cl, rq) == cl;
      pend = __ballot(won);
    }
    if (pend != 0ull || alive == 0ull || poll_ticks == 0ull || cook_ticks() - t0 > poll_ticks) break;
    SPIN_PAUSE_FAR();
  }
  const unsigned long long waited = cook_ticks() - t0;
  if ((pend >> lane) & 1ull) {
    const unsigned x = (unsigned)__popcll(pend & ((1ull << lane) - 1ull));
    nxt.pool[x] = my_pool;
    nxt.seq[x] = rq;
  }
  if (lane == 0) {
    const unsigned np = (unsigned)__popcll(pend);
    nxt.n = np;
    nxt.ticket_target = cur.ticket_target + np * (unsigned)MV_MERGE_BLOCKS;
    sc->iterations += 1u;
    sc->empty_iterations += nl == 0u ? 1u : 0u;
    sc->pools_served += nl;
    sc->wait_ticks += waited;
    sc->formed_tick = cook_ticks();
    const unsigned err = ld_agent(&sc->error);
    if (alive == 0ull) sc->all_done = 1u;
    if (alive == 0ull) st_system(&host->all_done, 1u);
    if (err != 0u) st_system(&host->error, err);
    st_system(&host->iter_done, sc->iterations);
  }
}
// (stepping form) behind a walker launch: the latch of iteration `it` once more — it publishes the same numbers again and now finds the
// requests the walkers have just posted
__global__ void __launch_bounds__(COOK_WAVE) match_serve_latch(ServeCtl* sc, ServeSlot* slots, ServeHost* host, unsigned it) {
  serve_latch(sc, slots, host, 0ull, it);
}
template <bool GE>
__global__ void __launch_bounds__(COOK_WAVE* MV_EW) COOK_EVAL_OCCUPANCY match_serve_eval(const PoolCtx* __restrict__ ctx, const ServeCtl* __restrict__ sc, unsigned it) {
  __shared__ __attribute__((aligned(16))) char lds[sizeof(EvalLds<GE>)];
  const ServeLatch& cur = sc->latch[it & 1u];
  if (blockIdx.z >= cur.n) return;
  const PoolCtx& c = ctx[cur.pool[blockIdx.z]];
  if (blockIdx.x >= c.vb.C) return;
  const bool dbg = sc->dbg_fence != 0u;
  if (dbg) {
    agent_acquire();
    __syncthreads();
  }
  eval_block<GE>(lds, c.in, c.st, c.vb, c.vb.ctl->head, c.vb.ctl->wcur, blockIdx.x, blockIdx.y, gridDim.y);
  if (dbg) {
    drain_stores();
    agent_release();
  }
}
template <bool GE>
__global__ void __launch_bounds__(COOK_WAVE* MV_MW) match_serve_merge(const PoolCtx* __restrict__ ctx, ServeCtl* sc, ServeSlot* slots, ServeHost* host,
                                                                      unsigned long long poll_ticks, unsigned it) {
  __shared__ unsigned s_last;
  const ServeLatch& cur = sc->latch[it & 1u];  // (stable for the whole launch: the latch writes the OTHER one)
  const unsigned nl = cur.n;
  if (nl == 0u ? (blockIdx.x | blockIdx.z) != 0u : blockIdx.z >= nl) return;  // (an empty iteration: block 0 is the latch)
  if (nl != 0u) {
    const PoolCtx& c = ctx[cur.pool[blockIdx.z]];
    if (sc->dbg_fence != 0u) {
      agent_acquire();
      __syncthreads();
    }
    merge_block<GE>(c.in, c.vb);
    if (sc->dbg_fence != 0u) {
      drain_stores();
      agent_release();
    }
  }
  drain_stores();  // (every wave: its list entries are in the L2 before thread 0 writes the L2 back)
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned last = 1u;
    if (nl != 0u) {
      agent_release();  // this workgroup's lists are in memory before its arrival counts
      last = atomicAdd(&sc->ticket, 1u) + 1u == cur.ticket_target ? 1u : 0u;
    }
    s_last = last;
  }
  __syncthreads();
  if (s_last == 0u || threadIdx.x >= (unsigned)COOK_WAVE) return;
  serve_latch(sc, slots, host, poll_ticks, it);
}

struct WalkCtx {  // what resolve_round needs of a pool (MatchIn through vb.in_dev): small enough for MV_WALK_PACK of them in the kernel arguments
  MatchState st;
  V2Buf vb;
};
constexpr int MV_WALK_PACK = 8;
template <int N>
struct WalkPack {
  WalkCtx c[N];
};
// one walker workgroup's life: rounds until the pool is placed (or, stepping form, until a window is not served yet)
template <bool GE>
static __device__ __forceinline__ void walk_pool(char* lds, int& s_go, const MatchState& st, const V2Buf& vb, ServeSlot* slot, ServeCtl* sc,
                                                 unsigned long long spin_ticks) {
  const unsigned K = vb.in_dev->K;
  for (;;) {
    if (threadIdx.x == 0) {
      int go = 0;
      const unsigned want = ld_agent(&slot->req);  // (this workgroup's own word, or the host's first request)
      if (ld_agent(&slot->done) == 0u) {
        const unsigned long long t0 = cook_ticks();
        for (;;) {
          if (ld_agent(&slot->ready) == want) {
            go = 1;
            break;
          }
          if (spin_ticks == 0ull) break;  // stepping form: come back when served
          if (ld_agent(&sc->error) != 0u || cook_ticks() - t0 > spin_ticks) {
            st_agent(&sc->error, 1u);
            st_agent(&slot->done, 2u);
            break;
          }
          SPIN_PAUSE_FAR();
        }
        if (go == 1) {
          slot->wait_ticks += cook_ticks() - t0;
          slot->waits += 1u;
          if (sc->dbg_delay[1] != 0u) {
            const unsigned long long d0 = cook_ticks();
            while (cook_ticks() - d0 < sc->dbg_delay[1]) SPIN_PAUSE_FAR();
          }
          agent_acquire();  // ONE acquire for the workgroup: the merged lists, colbits, group rows
          if (sc->dbg_delay[2] != 0u) {
            const unsigned long long d0 = cook_ticks();
            while (cook_ticks() - d0 < sc->dbg_delay[2]) SPIN_PAUSE_FAR();
          }
        }
      }
      s_go = go;
    }
    EMU_SITE("walker: served?");
    __syncthreads();
    if (s_go != 1) return;
    if (sc->dbg_fence != 0u) {
      agent_acquire();
      __syncthreads();
    }
    resolve_round<GE>(lds, st, vb);
    const unsigned long long t_round_end = cook_ticks();
    EMU_SITE("walker: round done");
    drain_stores();   // (every wave: see drain_stores)
    if (sc->dbg_fence != 0u) agent_release();
    __syncthreads();  // the walk is over (the helper waves wait here), every store of the round has been acknowledged
    if (threadIdx.x == 0) {
      const unsigned head = vb.ctl->head;  // (written by this thread, resolve_finish)
      agent_release();  // offer state, results, group chains, the control block: in memory before the request is
      if (head >= K) {
        st_agent(&slot->done, 1u);
        s_go = 0;
      } else {
        slot->post_ticks += cook_ticks() - t_round_end;
        st_agent(&slot->req, ld_agent(&slot->req) + 1u);
      }
    }
    __syncthreads();
    if (s_go != 1) return;
  }
}
template <bool GE, int N>
__global__ void __launch_bounds__(MV_RTHREADS) match_walkers_pack(const WalkPack<N> p, ServeSlot* slots, ServeCtl* sc, unsigned long long spin_ticks) {
  __shared__ __attribute__((aligned(16))) char lds[MV_RLDS_BYTES];
  __shared__ int s_go;
  const WalkCtx& c = p.c[blockIdx.x];
  walk_pool<GE>(lds, s_go, c.st, c.vb, &slots[blockIdx.x], sc, spin_ticks);
}
template <bool GE>
__global__ void __launch_bounds__(MV_RTHREADS) match_walkers(const PoolCtx* __restrict__ ctx, ServeSlot* slots, ServeCtl* sc, unsigned long long spin_ticks) {
  __shared__ __attribute__((aligned(16))) char lds[MV_RLDS_BYTES];
  __shared__ int s_go;
  const PoolCtx& c = ctx[blockIdx.x];
  walk_pool<GE>(lds, s_go, c.st, c.vb, &slots[blockIdx.x], sc, spin_ticks);
}

// match_v3.hpp — the same exact rank-ordered placement as match_v2.hpp (Fenzo scheduleOnce semantics, scheduler.clj:617-687), as ONE
// persistent workgroup per pool: no launch per round, no brute-force evaluation of every (job, offer) pair.
//
// What makes it cheap: cpuMemBinPacker (config.clj:108) is "fullness after the placement", and in exact arithmetic
//     2 * fitness(j, o) = G(o) + c_j / D_c(o) + m_j / D_m(o),      G(o) = used_cpus / D_c + used_mem / D_m  (job-independent),
// while "o has room for j" implies G(o) <= 2 - (c_j / D_c + m_j / D_m).  So the workgroup keeps, in LDS, ONE order of the live
// offers — grouped by (D_c, D_m, gpu signature), fullest first inside a group — with, per position, the sort key and the offer's free
// cpus / mem (two bf16, rounded up), and per block of 64 positions a summary (key range, largest free cpus / mem, range of the
// denominators).  For a job the summaries bound the best fitness any offer of a block could reach and tell which blocks cannot
// hold an offer with room; a HELPER wave looks at the few blocks that can matter, best bounds first: the per-position free
// resources (LDS) say which lanes are worth a look, only those read the offer's records (lane = offer, every value exact fp64,
// operation for operation as the oracle computes it), and the wave stops as soon as the entries it holds beat the bound of
// everything it has not looked at.  Bounds only PRUNE: every decision is taken on exact values, so the result is bit-identical to
// the one-job-at-a-time sweep for every input.
//
// Wave 0 WALKS the jobs in rank order exactly as match_v2's resolve kernel does — lanes own the offers committed to since the
// order was last updated ("touched", state in registers), the winner is max(best untouched offer of the job's list, best touched
// offer re-evaluated under the current state) — but it is fed through an LDS ring by the helper waves of its own workgroup, which
// run a bounded number of jobs ahead of it.  A helper leaves the offers that are touched when it looks OUT of the list (they are
// the walker's business), so a list goes stale only through the few offers opened between its preparation and its use.  When a
// list does run out, the ring is flushed and prepared again (an "epoch": two barriers, the walker keeps its lanes); when the
// touched set is full the walker writes its lanes back IN PLACE (a "generation": the <= 64 offers keep their positions, their keys
// and free resources are replaced and the summaries of their blocks recomputed — the bounds stay valid for any arrangement, and
// since best fit fills the offers roughly in the order they stand in, the order decays slowly); every few generations the order
// is rebuilt by a sort.  The offers' records live in global memory IN POSITION ORDER as a structure of arrays (V3Pos), so a
// helper's "lane = position" reads are coalesced: a gather of 64 scattered offer records costs the CU's one texture-address
// unit about a thousand cache-line requests per block, the same block in position order under a hundred.
//
// Scope (the host checks it and runs match_v2 otherwise): best fit (good-enough-fitness >= 1), no ports / named scalars, no
// balanced / attribute-equals groups, one offer per host, at most V3_MMAX offers.
#pragma once
#include "match_v2.hpp"

#ifndef COOK_V3_THREADS
#define COOK_V3_THREADS COOK_SHAPE(256, 128)
#endif
constexpr int V3_THREADS = COOK_V3_THREADS;           // threads of a workgroup (workgroup 0: wave 0 walks, the others feed it; the rest: helpers)
constexpr int V3_WAVES = V3_THREADS / COOK_WAVE;
constexpr int V3_MMAX = 8192;                          // offers per pool (the offer index is 13 bits of the sort key)
constexpr int V3_NBMAX = V3_MMAX / COOK_WAVE;          // blocks of the order
constexpr int V3_BPL = V3_NBMAX / COOK_WAVE;           // blocks per lane when a wave looks at every summary (2)
constexpr int V3_PPT = (V3_MMAX + V3_THREADS - 1) / V3_THREADS;  // positions per thread when the whole order moves
#ifndef COOK_V3_L
#define COOK_V3_L 8
#endif
constexpr int V3_L = COOK_V3_L;                        // list entries per job (<= 16: the list lives in the first row of a wave)
#ifndef COOK_V3_R
#define COOK_V3_R COOK_SHAPE(64, 8)
#endif
constexpr int V3_R = COOK_V3_R;                        // ring entries (the emulated tests: small, so that the ring wraps all the time)
constexpr int V3_T = COOK_WAVE;                        // touched offers per generation = lanes of the walking wave
#ifndef COOK_V3_GR
#define COOK_V3_GR COOK_SHAPE(256, 16)
#endif
constexpr int V3_GR = COOK_V3_GR;                      // entries of a bank of the global ring (helpers -> walker workgroup)
constexpr int V3_NBANK = 4;                            // banks: epoch e uses bank e % 4, so a straggler of epoch e - 1 .. e - 3 cannot clash
constexpr int V3_HW_MAX = 512;                         // helper waves of a launch, at most
constexpr unsigned V3_E_DONE = 0xFFFFFFFFu;            // "epoch" that tells the helpers the call is over
constexpr int V3_JOB_WORDS = 40;                       // sizeof(V3Job) / 8
#ifndef V3_FORCE_FULL
#define V3_FORCE_FULL 0
#endif
constexpr int V3_DLOG = 16;                            // generations whose dirty-block masks are kept for helpers that fell behind
constexpr int V3_BATCH = 4;                            // blocks a helper looks at per step (their loads are in flight together)
static_assert(V3_L <= 16 && V3_BPL == 2 && V3_R <= COOK_WAVE, "shapes");

struct V3Ent {  // candidate-list entry: exact fitness under the generation's snapshot, offer, its position in the order
  double fit;
  int off;
  unsigned pos;
};
constexpr unsigned V3I_TRUNC = 1u << 8;     // feasible untouched offers may exist beyond the list
constexpr unsigned V3I_NOFEAS = 1u << 9;    // no offer at all (touched ones included) was feasible under the snapshot
constexpr unsigned V3I_PLAIN = 1u << 10;    // no gpus, no constraints of its own, no group, no reserved host: the walker's short path
constexpr unsigned V3I_GPU = 1u << 16, V3I_GROUPED = 1u << 17, V3I_HASGROUP = 1u << 20, V3I_FASTC = 1u << 21, V3I_SLOW = 1u << 22,
                   V3I_GFAST = 1u << 23;  // bits 18-19: group type; GFAST: the hosts to avoid are staged (gfh / n_fh / glast)
struct alignas(16) V3Job {  // one prepared job (a ring entry)
  V3Ent ent[V3_L];
  double c, m, g;
  unsigned info;          // bits 0-7 entries, V3I_*
  unsigned k;             // match position
  unsigned gpu_model;
  int reserved_host;
  unsigned group, jj;
  unsigned short f1, f2, f4, pad;  // saturated counts of offers failing on resources / constraints / zero fitness under the snapshot
  unsigned req[MV_NA], wild[MV_NA], req_host, wild_host, novel[MV_NC], impossible;  // (V3I_FASTC) EvalCons
  unsigned gfh[MV_FH];    // (V3I_GFAST) hosts the job's cotasks occupied when the generation began
  int n_fh, glast;
};

struct V3BlockSum {  // summary of one block of 64 positions of the order (floats rounded towards the safe side)
  float kmax, kmin;      // >= the greatest / <= the smallest key G of the block
  float maxc, maxm;      // >= the greatest free cpus / mem under the snapshot
  float ri_dc, ri_dm;    // >= 1 / (smallest denominator)
  float rx_dc, rx_dm;    // <= 1 / (greatest denominator)
  unsigned sig, pad;     // which kinds of gpu request the block's offers can serve (v3_offer_sig, OR over the block)
};
// bit 0: takes jobs without gpus; bit 1: a k8s "gpus" map with several entries (may serve any gpu job); bits 2..31: a hash of the
// one (model, count) a gpu job must ask for (constraints.clj:122-157 as static_fast evaluates it)
static __device__ __forceinline__ unsigned v3_gpu_bit(unsigned model, double count) {
  unsigned long long h = ((unsigned long long)__double_as_longlong(count) + model) * 0x9E3779B97F4A7C15ull;
  return 1u << (2u + (unsigned)(h >> 59) % 30u);
}
static __device__ __forceinline__ unsigned v3_offer_sig(unsigned flags, unsigned gpu_model, double gpu_count) {
  if (!(flags & 1u)) return 1u;
  if (flags & 4u) return 2u;
  return gpu_model == 0u ? 1u : v3_gpu_bit(gpu_model, gpu_count);
}

constexpr int V3_NPROF = 24;
struct V3Ctl {  // in global memory: results and statistics of one call
  unsigned head, matched, head_matched, generations;
  unsigned stop_full, stop_list, stop_log, stop_other;
  unsigned walked, settled, scan_steps, opens;
  unsigned long long t_total, t_regen, t_walk_wait;  // 100 MHz ticks
  unsigned error;  // != 0: the kernel refused the input before placing anything (the host runs match_v2)
  unsigned epochs; // ring flushes + generations
  unsigned fast, visits;  // jobs decided by the walker's fast path; blocks whose offers' records a helper read
  unsigned long long t_flush;
  unsigned long long prof_cyc[V3_NPROF];  // (COOK_V3_PROF builds) shader cycles / events by phase, summed over the waves
  unsigned long long prof_cnt[V3_NPROF];
};

// -DCOOK_V3_PROF: a measurement build — shader cycles by phase.  Helper phases: 0 job record + constraint form + group hosts, 1 block
// bounds, 2 block selection, 3 free-resource look-up (LDS), 4 offer evaluation (records from HBM), 5 list merge, 6 publish, 7 waiting
// for the walker (look-ahead window), 8 insertions (count only).  Walker phases: 16 waiting / skipping, 17 job load + touched offers,
// 18 fast decision, 19 general decision, 20 commit on a touched lane, 21 commit on a new lane, 22 unmatched.
#ifdef COOK_V3_PROF
#define V3P_DECL() unsigned long long pk_ = __builtin_readcyclecounter()
#define V3P_MARK(pc, pn, i)                                      \
  do {                                                           \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    (pc)[i] += now_ - pk_;                                       \
    (pn)[i] += 1u;                                               \
    pk_ = now_;                                                  \
  } while (0)
#define V3P_COUNT(pn, i, v) ((pn)[i] += (v))
#else
#define V3P_DECL() ((void)0)
#define V3P_MARK(pc, pn, i) ((void)0)
#define V3P_COUNT(pn, i, v) ((void)0)
#endif

struct V3Pos {  // the offers of the order BY POSITION (structure of arrays, V3_MMAX entries each)
  double *oc, *om, *rc, *rm;  // lease cpus / mem, Fenzo's running cpus / mem
  double *ac, *am;            // assigned in this call (as of the last generation change)
  double* gpu_count;
  unsigned *off, *host, *gpu_model, *flags;
  int *run_count, *slack, *acount;
  unsigned* attr;             // [MV_NA][V3_MMAX]
};
constexpr size_t V3_POS_BYTES = (size_t)V3_MMAX * (7 * 8 + 4 * 4 + 3 * 4 + MV_NA * 4);
static inline V3Pos v3_pos_carve(char* base) {  // (host) the arrays inside one allocation of V3_POS_BYTES
  V3Pos P;
  double* d = reinterpret_cast<double*>(base);
  P.oc = d, P.om = d + V3_MMAX, P.rc = d + 2 * V3_MMAX, P.rm = d + 3 * V3_MMAX, P.ac = d + 4 * V3_MMAX, P.am = d + 5 * V3_MMAX, P.gpu_count = d + 6 * V3_MMAX;
  unsigned* u = reinterpret_cast<unsigned*>(d + 7 * V3_MMAX);
  P.off = u, P.host = u + V3_MMAX, P.gpu_model = u + 2 * V3_MMAX, P.flags = u + 3 * V3_MMAX;
  P.run_count = reinterpret_cast<int*>(u + 4 * V3_MMAX), P.slack = reinterpret_cast<int*>(u + 5 * V3_MMAX), P.acount = reinterpret_cast<int*>(u + 6 * V3_MMAX);
  P.attr = u + 7 * V3_MMAX;
  return P;
}

// What the walker workgroup and the helper workgroups share (global memory, one allocation per engine).  Payload that changes only at
// a generation change (fcm .. gen_first) is written with plain stores and handed over by an agent-scope release / acquire pair; the
// ring, its flags and the control words are sc1 (write-through, L1-bypassing) accesses on both sides: no fences on the per-job path.
struct V3Glob {
  unsigned* fcm;                // [V3_MMAX] master copies of the helpers' LDS arrays
  float* okey;                  // [V3_MMAX]
  V3BlockSum* bsum;             // [V3_NBMAX]
  unsigned long long* tbits;    // [V3_NBMAX] positions touched in this generation (sc1)
  unsigned long long* dlog;     // [V3_DLOG][V3_BPL] blocks a generation changed (all ones: everything)
  unsigned* hdr;                // [0] n_pos, [1] gen_first  (payload)
  V3Job* ring;                  // [V3_NBANK][V3_GR]
  unsigned long long* rflag;    // [V3_NBANK][V3_GR]: (job + 1) | (epoch << 5 | failure bits << 2 | 1 ready / 2 settled) << 32
  unsigned long long* next64;   // epoch << 32 | next job position (helpers fetch-and-add it); epoch 0: not started
  unsigned* walk_pos;           // first job the walker has not consumed (published by the walker workgroup's feeder wave)
  unsigned* gen;                // generation number
  unsigned* ack;                // [V3_HW_MAX] the epoch each helper wave last took a job in
};
constexpr size_t V3_GLOB_CTL_OFF = (size_t)V3_MMAX * 8 + sizeof(V3BlockSum) * V3_NBMAX + 8 * V3_NBMAX + 8 * V3_DLOG * V3_BPL + 64 + sizeof(V3Job) * V3_NBANK * V3_GR;
constexpr size_t V3_GLOB_CTL_BYTES = 8 * (size_t)V3_NBANK * V3_GR + 64 + 4 * V3_HW_MAX;  // flags + control words: zeroed before every call
constexpr size_t V3_GLOB_BYTES = V3_GLOB_CTL_OFF + V3_GLOB_CTL_BYTES;
static inline V3Glob v3_glob_carve(char* base) {  // (host)
  V3Glob G;
  char* q = base;
  G.fcm = reinterpret_cast<unsigned*>(q), q += (size_t)V3_MMAX * 4;
  G.okey = reinterpret_cast<float*>(q), q += (size_t)V3_MMAX * 4;
  G.bsum = reinterpret_cast<V3BlockSum*>(q), q += sizeof(V3BlockSum) * V3_NBMAX;
  G.tbits = reinterpret_cast<unsigned long long*>(q), q += 8 * V3_NBMAX;
  G.dlog = reinterpret_cast<unsigned long long*>(q), q += 8 * V3_DLOG * V3_BPL;
  G.hdr = reinterpret_cast<unsigned*>(q), q += 64;
  G.ring = reinterpret_cast<V3Job*>(q), q += sizeof(V3Job) * V3_NBANK * V3_GR;
  // ---- zeroed before every call from here on (V3_GLOB_CTL_OFF) ----
  G.rflag = reinterpret_cast<unsigned long long*>(q), q += 8 * (size_t)V3_NBANK * V3_GR;
  G.next64 = reinterpret_cast<unsigned long long*>(q);
  G.walk_pos = reinterpret_cast<unsigned*>(q + 8);
  G.gen = reinterpret_cast<unsigned*>(q + 12), q += 64;
  G.ack = reinterpret_cast<unsigned*>(q);
  return G;
}

struct V3Buf {
  V3Pos P;
  V3Glob G;
  const OfferA* oa;
  const OfferB* ob;
  const JobRec* jr;
  const JobCons* jcons;
  const MatchIn* in_dev;
  V3Ctl* ctl;
  int32_t* group_snap;          // [G] st.group_last as the generation began (the helpers' view; the walker publishes to the live array)
  const unsigned long long* job_flags;  // [0..1] jmin bits, [2] != 0: some job has a negative / non-finite request
  unsigned look_ahead;          // jobs the helpers may run ahead of the walker (1 .. V3_R)
  unsigned rebuild_gens;        // the order is rebuilt by a sort every that many generations (>= 1)
  unsigned n_helper_waves;      // (gridDim.x - 1) * waves per workgroup
  unsigned pad;
};

static_assert(sizeof(V3Job) * V3_R <= sizeof(unsigned long long) * V3_MMAX, "the ring lives in the sort buffer");
static_assert(sizeof(V3Job) == 8 * V3_JOB_WORDS, "a ring entry moves as 40 8-byte words");
struct V3Lds {
  union {
    unsigned long long skey[V3_MMAX];   // sort buffer of a rebuild: class hash (19) | NOT key bits (32: fullest first) | offer (13)
    V3Job ring[V3_R];                   // between rebuilds the same bytes hold the ring
  };
  float okey[V3_MMAX];                  // position -> key (>= G of the offer under the snapshot; < 0: dead)
  unsigned fcm[V3_MMAX];              // position -> free cpus << 16 | free mem under the snapshot, two bf16 rounded UP (0 0: dead)
  unsigned char owner[V3_MMAX];       // offer -> lane of the walker that owns it in this generation, 0xFF none
  V3BlockSum bsum[V3_NBMAX];
  unsigned long long tbits[V3_NBMAX];    // block -> positions touched in this generation
  unsigned long long dirty[V3_BPL];      // blocks whose summaries the generation change must recompute
  unsigned rstate[V3_R];              // (position + 1) << 2 | 1 ready / 2 settled
  int res_j2o[V3_R];                  // the walker's verdicts by ring slot; a feeder wave writes them out behind the walker
  unsigned res_fail[V3_R];            // 0xFFFFFFFF: nothing to write (settled by a helper / a group member, written at once)
  unsigned flushed;                   // jobs below this position have their results in HBM
  unsigned n_pos, n_blocks;           // positions of the order (dead ones included until the next rebuild)
  unsigned next;                      // next job position a helper takes
  unsigned walk_pos;                  // first job the walker has not consumed
  unsigned gen_first;                 // first job of this generation (cutoff of the group chains)
  unsigned gen_stop, done;
  unsigned sort_n;
  unsigned stop_reason;               // why the epoch ended: 1 list ran out, 2 touched set full, 3 group log full
  unsigned abort;                     // the walker waited for a prepared job longer than V3_WAIT_TICKS (a bug, never the input): give up loudly
};
struct V3HLds {  // a helper workgroup: its copy of what the search reads, refreshed at every generation change
  float okey[V3_MMAX];
  unsigned fcm[V3_MMAX];
  V3BlockSum bsum[V3_NBMAX];
  V3Job stage[V3_WAVES];                     // a wave builds its entry here before it goes out
  unsigned short cand[V3_WAVES][COOK_WAVE];  // a wave's candidate positions of a pass
  unsigned n_pos, n_blocks, gen_first;
  unsigned gen_loaded;                       // generation the copy stands for
  unsigned lock;                             // the wave that refreshes the copy holds it
};
constexpr unsigned long long V3_WAIT_TICKS = 150000000ull;  // 1.5 s of the 100 MHz clock

// One lane reads a word other waves write, every lane gets that value: on the GPU a wave-wide load of one address is uniform anyway,
// but lanes that are fibers (the emulated build) would each read at their own time and could take different branches.
static __device__ __forceinline__ unsigned v3_ld_agent_u(const unsigned* p) {
  unsigned v = 0;
  if (lane_id() == 0) v = ld_agent(p);
  return (unsigned)__shfl((int)v, 0, COOK_WAVE);
}
static __device__ __forceinline__ unsigned long long v3_ld_agent_u64(const unsigned long long* p) {
  unsigned long long v = 0;
  if (lane_id() == 0) v = ld_agent(p);
  const unsigned lo = (unsigned)__shfl((int)(unsigned)v, 0, COOK_WAVE), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), 0, COOK_WAVE);
  return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}
static __device__ __forceinline__ unsigned v3_ld_wg_u(const unsigned* p) {
  unsigned v = 0;
  if (lane_id() == 0) v = ld_wg(p);
  return (unsigned)__shfl((int)v, 0, COOK_WAVE);
}
// float >= d / <= d (the summaries must err on the safe side)
static __device__ __forceinline__ float v3_f32_up(double d) {
  float f = (float)d;
  if ((double)f < d) f = __int_as_float(__float_as_int(f) + (f >= 0.0f ? 1 : -1));
  return f;
}
static __device__ __forceinline__ float v3_f32_down(double d) {
  float f = (float)d;
  if ((double)f > d) f = __int_as_float(__float_as_int(f) + (f > 0.0f ? -1 : 1));
  return f;
}
// bf16 >= f for f >= 0 (anything else -> 0); an overflow becomes +inf, which is on the safe side
static __device__ __forceinline__ unsigned v3_bf16_up(float f) {
  if (!(f > 0.0f)) return 0u;
  const unsigned b = (unsigned)__float_as_int(f);
  return (b >> 16) + ((b & 0xFFFFu) ? 1u : 0u);
}
static __device__ __forceinline__ unsigned v3_pack_free(double fc, double fm) { return (v3_bf16_up(v3_f32_up(fc)) << 16) | v3_bf16_up(v3_f32_up(fm)); }
static __device__ __forceinline__ float v3_free_c(unsigned w) { return __int_as_float((int)(w & 0xFFFF0000u)); }
static __device__ __forceinline__ float v3_free_m(unsigned w) { return __int_as_float((int)(w << 16)); }
static __device__ __forceinline__ float v3_key_f32(unsigned long long sk) { return __int_as_float((int)~(unsigned)((sk >> 13) & 0xFFFFFFFFull)); }

static __device__ __forceinline__ unsigned v3_hash(unsigned long long a, unsigned long long b, unsigned c, unsigned long long d) {
  unsigned long long h = a * 0x9E3779B97F4A7C15ull;
  h ^= (b + 0x7F4A7C159E3779B9ull) * 0xC2B2AE3D27D4EB4Full;
  h ^= ((unsigned long long)c + 0x165667B19E3779F9ull) * 0x9E3779B97F4A7C15ull;
  h ^= (d + 0x27D4EB2F165667C5ull) * 0xC2B2AE3D27D4EB4Full;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return (unsigned)h;
}
// key of an offer in the order: group hash (19 bits) | NOT key bits (32: fullest first) | offer (13)
static __device__ __forceinline__ unsigned long long v3_sort_key(const OfferA& a, const OfferB& b, double ac, double am, unsigned v) {
  const double dc = a.oc + a.rc, dm = a.om + a.rm;
  const double G = (a.rc + ac) / dc + (a.rm + am) / dm;
  float kf = v3_f32_up(G * (1.0 + 0x1p-40));
  if (!(kf >= 0.0f)) kf = 0.0f;
  const unsigned h = v3_hash((unsigned long long)__double_as_longlong(dc), (unsigned long long)__double_as_longlong(dm),
                             b.gpu_model * 2u + (b.flags & 1u), (unsigned long long)__double_as_longlong(b.gpu_count)) & 0x7FFFFu;
  return ((unsigned long long)h << 45) | ((unsigned long long)(~(unsigned)__float_as_int(kf)) << 13) | (unsigned long long)v;
}

// ---- bitonic sort of L.skey[0 .. n2) (n2 a power of two), ascending; every thread of the workgroup takes part --------------------
static __device__ __forceinline__ void v3_sort(V3Lds& L, unsigned n2) {
  const unsigned tid = threadIdx.x, NT = blockDim.x;
  for (unsigned k = 2; k <= n2; k <<= 1) {
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
      for (unsigned x = tid; x < (n2 >> 1); x += NT) {
        const unsigned lo = ((x & ~(j - 1)) << 1) | (x & (j - 1)), hi = lo | j;
        const unsigned long long a = L.skey[lo], b = L.skey[hi];
        const bool up = (lo & k) == 0;
        if ((a > b) == up) {
          L.skey[lo] = b;
          L.skey[hi] = a;
        }
      }
      __syncthreads();
    }
  }
}

// ---- block summaries of the order (one wave per block, lane = position); ends with a barrier -------------------------------------------
// full: every block, denominators and gpu signatures included (after a rebuild); else only the blocks of L.dirty, and only what a
// placement changes (keys, free resources).
static __device__ __forceinline__ void v3_summaries(V3Lds& L, const V3Buf& vb, bool full) {
  const unsigned lane = lane_id(), NW = blockDim.x / COOK_WAVE;
  const V3Pos& P = vb.P;
  const unsigned n = L.n_pos, nb = (n + COOK_WAVE - 1) / COOK_WAVE;
  for (unsigned b = wave_id(); b < nb; b += NW) {
    if (!full && !((L.dirty[b >> 6] >> (b & 63u)) & 1ull)) continue;  // (wave-uniform)
    const unsigned p = b * COOK_WAVE + lane;
    float key = 0.0f, fc = 0.0f, fm = 0.0f;
    unsigned nkey = 0u;  // NOT bits of the values whose minimum is wanted (non-negative floats order like their bits)
    if (p < n) {
      const float k0 = L.okey[p];
      const unsigned w = L.fcm[p];
      if (k0 >= 0.0f) {
        key = k0;
        nkey = ~(unsigned)__float_as_int(k0);
        fc = v3_free_c(w);
        fm = v3_free_m(w);
      }
    }
    const float kmax = wave_max_f32(key), maxc = wave_max_f32(fc), maxm = wave_max_f32(fm);
    const unsigned xkey = wave_max_u32(nkey);
    V3BlockSum s;
    if (full) {
      float idc = 0.0f, idm = 0.0f;
      unsigned nidc = 0u, nidm = 0u, sig = 0u;
      if (p < n) {
        const double rdc = 1.0 / (P.oc[p] + P.rc[p]), rdm = 1.0 / (P.om[p] + P.rm[p]);
        idc = v3_f32_up(rdc * (1.0 + 0x1p-50));
        idm = v3_f32_up(rdm * (1.0 + 0x1p-50));
        nidc = ~(unsigned)__float_as_int(v3_f32_down(rdc * (1.0 - 0x1p-50)));
        nidm = ~(unsigned)__float_as_int(v3_f32_down(rdm * (1.0 - 0x1p-50)));
        sig = v3_offer_sig(P.flags[p], P.gpu_model[p], P.gpu_count[p]);
      }
      s.ri_dc = wave_max_f32(idc), s.ri_dm = wave_max_f32(idm);
      s.rx_dc = __int_as_float((int)~wave_max_u32(nidc)), s.rx_dm = __int_as_float((int)~wave_max_u32(nidm));
      for (int d = 32; d >= 1; d >>= 1) sig |= (unsigned)__shfl_xor((int)sig, d, COOK_WAVE);
      s.sig = sig, s.pad = 0u;
    } else {
      s = L.bsum[b];
    }
    s.kmax = kmax, s.kmin = __int_as_float((int)~xkey), s.maxc = maxc, s.maxm = maxm;  // (a block of dead offers: kmin = NaN bits of ~0 -> no room)
    if (lane == 0) L.bsum[b] = s;
  }
  if (threadIdx.x == 0) L.n_blocks = nb;
  __syncthreads();
}

// ---- a rebuild: every live offer sorted into the order, its records written out in position order -----------------------------------
static __device__ __forceinline__ void v3_build(V3Lds& L, const MatchIn& in, const MatchState& st, const V3Buf& vb) {
  const unsigned tid = threadIdx.x, NT = blockDim.x, lane = lane_id();
  const unsigned M = in.M;
  const V3Pos& P = vb.P;
  if (tid == 0) L.sort_n = 0;
  for (unsigned v = tid; v < (unsigned)V3_MMAX; v += NT) L.owner[v] = 0xFF;
  for (unsigned b = tid; b < (unsigned)V3_NBMAX; b += NT) L.tbits[b] = 0ull;
  if (tid < (unsigned)V3_BPL) L.dirty[tid] = 0ull;
  __syncthreads();
  // live offers -> sort keys (dead ones cannot take the smallest job of the call: they never come back)
  for (unsigned v0 = 0; v0 < M; v0 += NT) {
    const unsigned v = v0 + tid;
    bool live = false;
    unsigned long long sk = ~0ull;
    if (v < M) {
      live = ((st.alive[v >> 6] >> (v & 63u)) & 1ull) != 0ull;
      if (live) sk = v3_sort_key(vb.oa[v], vb.ob[v], st.ac[v], st.am[v], v);
    }
    const unsigned long long bal = __ballot(live);
    unsigned base = 0;
    if (lane == 0 && bal) base = atomicAdd(&L.sort_n, (unsigned)__popcll(bal));
    base = (unsigned)__shfl((int)base, 0, COOK_WAVE);
    if (live) L.skey[base + (unsigned)__popcll(bal & lanemask_lt())] = sk;
  }
  __syncthreads();
  const unsigned n = L.sort_n;
  unsigned n2 = 64;
  while (n2 < n) n2 <<= 1;
  for (unsigned x = n + tid; x < n2; x += NT) L.skey[x] = ~0ull;
  __syncthreads();
  v3_sort(L, n2);
  for (unsigned p = tid; p < n; p += NT) {
    const unsigned long long sk = L.skey[p];
    const unsigned v = (unsigned)(sk & 0x1FFFull);
    const OfferA a = vb.oa[v];
    const OfferB o = vb.ob[v];
    const double ac = st.ac[v], am = st.am[v];
    P.oc[p] = a.oc, P.om[p] = a.om, P.rc[p] = a.rc, P.rm[p] = a.rm;
    P.ac[p] = ac, P.am[p] = am;
    P.gpu_count[p] = o.gpu_count;
    P.off[p] = v, P.host[p] = o.host, P.gpu_model[p] = o.gpu_model, P.flags[p] = o.flags;
    P.run_count[p] = o.run_count, P.slack[p] = o.task_slack, P.acount[p] = st.acount[v];
#pragma unroll
    for (int x = 0; x < MV_NA; ++x) P.attr[(size_t)x * V3_MMAX + p] = (in.o_attr && (unsigned)x < in.n_attr) ? in.o_attr[(size_t)v * in.n_attr + x] : 0u;
    L.okey[p] = v3_key_f32(sk);
    L.fcm[p] = v3_pack_free(a.oc - ac, a.om - am);
  }
  if (tid == 0) L.n_pos = n;
  __threadfence_block();
  __syncthreads();
  v3_summaries(L, vb, true);
}

// ---- a helper wave prepares job k: candidate list + failure counts under the generation's snapshot ----------------------------------
static __device__ __forceinline__ void v3_prepare(V3HLds& H, const MatchIn& in, const MatchState& st, const V3Buf& vb, unsigned k, V3Job& J, unsigned* steps_out,
                                  unsigned* visits_out, unsigned long long* pc, unsigned long long* pn) {
  const unsigned lane = lane_id();
  V3P_DECL();
  const JobRec j = vb.jr[k];
  const unsigned jj = in.j_index ? in.j_index[k] : k;
  const bool slow = (j.flags & JF_SLOW) != 0, grouped = (j.flags & JF_GROUPED) != 0, fastc = !slow && (j.flags & JF_FASTC) != 0;
  const unsigned gtype = (j.flags >> 8) & 3u;
  // the constraint form (EvalCons): built from the packed JobCons
  EvalCons E;
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
        if (key == 0xFFFFFFFFu) {
          if (E.wild_host && E.req_host != val) E.impossible = true;
          E.req_host = val;
          E.wild_host = 0xFFFFFFFFu;
        } else if (key >= (unsigned)MV_NA) {
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
  // unique host-placement groups: hosts of the running cotasks and of the cotasks placed before this generation
  unsigned fh[MV_FH];
  int n_fh = -1, glast = -1;
#pragma unroll
  for (int q = 0; q < MV_FH; ++q) fh[q] = 0xFFFFFFFFu;
  const int cutoff = (int)H.gen_first;
  if (j.group != 0xFFFFFFFFu) glast = vb.group_snap[j.group];  // the group's last job placed BEFORE this generation (later ones: the walker's log)
  if (grouped && gtype == 1u) {
    n_fh = 0;
    const unsigned g = j.group;
    const unsigned r0 = in.g_run_off ? in.g_run_off[g] : 0u, r1 = in.g_run_off ? in.g_run_off[g + 1] : 0u;
    auto push = [&](unsigned h) {
      if (n_fh >= 0 && n_fh < MV_FH) {
#pragma unroll
        for (int q = 0; q < MV_FH; ++q)
          if (q == n_fh) fh[q] = h;
        ++n_fh;
      } else {
        n_fh = -2;
      }
    };
    for (unsigned x = r0; x < r1 && n_fh >= 0; ++x) push(in.g_run_host[x]);
    for (int c = glast; c >= 0 && n_fh >= 0; c = ld_agent(&st.job_prev[c])) push(in.o_host[ld_agent(&st.job_to_offer[c])]);
  }
  MatchState st_cut = st;  // the general group check under the snapshot: the chains as the generation began
  st_cut.cutoff = cutoff;
  st_cut.group_last = vb.group_snap;
  V3P_MARK(pc, pn, 0);
  // ---- which blocks can matter, and how good an offer of each could be (floats, every rounding towards "may matter") ------------------
  const unsigned nb = H.n_blocks, n_pos = H.n_pos;
  const V3Pos& P = vb.P;
  const float c_up = v3_f32_up(j.c), m_up = v3_f32_up(j.m), c_dn = v3_f32_down(j.c), m_dn = v3_f32_down(j.m);
  const unsigned need_sig = j.g > 0 ? (2u | v3_gpu_bit(j.gpu_model, j.g)) : 1u;
  float ub[V3_BPL], nh[V3_BPL], th[V3_BPL];  // the lane's blocks: bound of the best fitness (0: cannot matter), need (upper), key threshold of "room"
  bool dfr[V3_BPL];                          // room by the summary, but no offer of the block serves the job's kind of gpu request
  bool any_nores = n_pos < in.M;             // some offer fails on resources (the dead ones do)
#pragma unroll
  for (int q = 0; q < V3_BPL; ++q) {
    const unsigned b = (unsigned)q * COOK_WAVE + lane;
    ub[q] = nh[q] = th[q] = 0.0f;
    dfr[q] = false;
    if (b < nb) {
      const V3BlockSum s = H.bsum[b];
      const float need_hi = (c_up * s.ri_dc + m_up * s.ri_dm) * (1.0f + 0x1p-20f);
      const float need_lo = (c_dn * s.rx_dc + m_dn * s.rx_dm) * (1.0f - 0x1p-20f);
      // an offer with room has G <= 2 - need (used = D - free): a block whose smallest key exceeds that holds none
      const float thr = (2.0f - need_lo) * (1.0f + 0x1p-20f) + 0x1p-20f;
      const bool room = s.maxc >= c_dn && s.maxm >= m_dn && s.kmin <= thr;
      if (room) {
        float u = (fminf(s.kmax, thr) + need_hi) * 0.5f * (1.0f + 0x1p-20f) + 0x1p-100f;
        if (!(u < 1.0f)) u = 1.0f;  // (room implies fitness <= 1; also catches a NaN)
        nh[q] = need_hi;
        th[q] = thr;
        if (s.sig & need_sig)
          ub[q] = u;  // > 0
        else
          dfr[q] = true;  // every offer of the block fails the gpu-host constraint (or on resources): only the failure counts care
      } else {
        any_nores = true;
      }
    }
  }
  any_nores = __any(any_nores);
  V3P_MARK(pc, pn, 1);
  // ---- the search: lane = CANDIDATE.  Blocks are loaded a group at a time (free resources + keys from LDS give every position an
  // upper bound `a` of its fitness, 0 = no room); the positions that can still enter the list are packed into the wave's lanes,
  // evaluated exactly from the position-ordered records, and ranked together with the entries the list already holds.
  constexpr int NBL = 8;                       // blocks of a group
  constexpr unsigned CAP = COOK_WAVE - V3_L;   // candidates of a pass (the last V3_L lanes carry the list so far)
  unsigned short* const cbuf = H.cand[wave_id()];
  unsigned n_list = 0;
  double t8 = -1.0;   // (fitness, offer) of the list's last entry once it is full
  int t8_off = -1;
  bool more = false;  // feasible untouched offers exist (or may exist) beyond the list
  unsigned n_res = 0, n_feas = 0, n_zero = 0, steps = 0, visits = 0;
  bool counting = false;  // second phase: the deferred blocks, for the failure counts only
  for (unsigned guard = 0; guard < 4u * (unsigned)V3_NBMAX; ++guard) {
    const float mx = wave_max_f32(fmaxf(ub[0], ub[1]));
    if (!(mx > 0.0f)) {
      // every block that could hold a candidate has been dealt with.  The failure counts of a list that claims to be complete must be
      // exact up to V3_T: the blocks left out for their gpu signature hold offers that fail on constraints if they have room
      const bool trunc0 = n_list == (unsigned)V3_L && more;
      if (counting || trunc0) break;
      const unsigned c2now = n_res - n_feas - n_zero;
      if (c2now > (unsigned)V3_T && any_nores) break;
      if (!__any(dfr[0] || dfr[1])) break;
      counting = true;
#pragma unroll
      for (int q = 0; q < V3_BPL; ++q)
        if (dfr[q]) ub[q] = 0x1p-100f, dfr[q] = false;
      continue;
    }
    if (!counting && n_list == (unsigned)V3_L && (double)mx < t8) {  // nothing left can enter the list (an equal bound could, through a lower offer index)
      more = true;
      break;
    }
    if (counting && (n_res - n_feas - n_zero) > (unsigned)V3_T && any_nores) break;  // the counts are settled
    // ---- a group: the blocks whose bound is close to the best one, at most NBL of them -----------------------------------------------
    const float cut = mx * (1.0f - 0x1p-5f);
    const double lo0 = (!counting && n_list == (unsigned)V3_L) ? t8 : -1.0;
    unsigned long long s0 = __ballot(ub[0] >= cut && ub[0] > 0.0f && (double)ub[0] >= lo0), s1 = __ballot(ub[1] >= cut && ub[1] > 0.0f && (double)ub[1] >= lo0);
    unsigned bsel[NBL];
    int nsel = 0;
#pragma unroll
    for (int s = 0; s < NBL; ++s) {
      bsel[s] = 0u;
      if (s0 != 0ull) {
        bsel[s] = (unsigned)__ffsll((unsigned long long)s0) - 1u;
        s0 &= s0 - 1ull;
        nsel = s + 1;
      } else if (s1 != 0ull) {
        bsel[s] = 64u + (unsigned)__ffsll((unsigned long long)s1) - 1u;
        s1 &= s1 - 1ull;
        nsel = s + 1;
      }
    }
    float a[NBL];  // bound of the fitness of position (block s, this lane); 0: no room / nothing there
#pragma unroll
    for (int s = 0; s < NBL; ++s) {
      a[s] = 0.0f;
      if (s < nsel) {  // (wave-uniform)
        const unsigned b = bsel[s], src = b & 63u;
        const float nh_b = __int_as_float(wave_read_lane(__float_as_int(b < 64u ? nh[0] : nh[1]), (int)src));
        const float th_b = __int_as_float(wave_read_lane(__float_as_int(b < 64u ? th[0] : th[1]), (int)src));
        if (src == lane) {
          if (b < 64u)
            ub[0] = 0.0f;
          else
            ub[1] = 0.0f;
        }
        const unsigned p = b * COOK_WAVE + lane;
        bool room = false;
        if (p < n_pos) {
          const unsigned w = H.fcm[p];
          const float kf = H.okey[p];
          room = v3_free_c(w) >= c_dn && v3_free_m(w) >= m_dn && kf >= 0.0f;
          if (room) {
            float u = (fminf(kf, th_b) + nh_b) * 0.5f * (1.0f + 0x1p-20f) + 0x1p-100f;
            if (!(u < 1.0f)) u = 1.0f;
            a[s] = u;
          }
        }
        if (__any(p < n_pos && !room)) any_nores = true;
        ++steps;
      }
    }
    V3P_MARK(pc, pn, 2);
    // ---- passes over the group: whoever can still enter the list -------------------------------------------------------------------------
    for (unsigned pass = 0; pass < 2u * (unsigned)NBL * COOK_WAVE; ++pass) {
      if (counting && (n_res - n_feas - n_zero) > (unsigned)V3_T && any_nores) break;  // the counts are settled
      // (a >= fitness >= t8 is needed to enter a full list; the counting phase wants every position that has room)
      const float lo_f = (!counting && n_list == (unsigned)V3_L) ? v3_f32_down(t8) : 0.0f;
      unsigned long long pm[NBL];
      unsigned total = 0;
#pragma unroll
      for (int s = 0; s < NBL; ++s) {
        pm[s] = __ballot(a[s] > 0.0f && a[s] >= lo_f);
        total += (unsigned)__popcll(pm[s]);
      }
      if (total == 0u) break;
      if (total > 12u && !counting) {  // the best ones first: a threshold near the group's best bound, widened until enough positions pass
        float theta = mx * (1.0f - 0x1p-7f);
        for (int w = 0; w < 3; ++w) {
          unsigned cnt = 0;
#pragma unroll
          for (int s = 0; s < NBL; ++s) cnt += (unsigned)__popcll(__ballot(a[s] >= theta && a[s] > 0.0f && a[s] >= lo_f));
          if (cnt >= 8u) break;
          theta = w == 0 ? mx * (1.0f - 0x1p-5f) : (w == 1 ? mx * (1.0f - 0x1p-3f) : 0.0f);
        }
#pragma unroll
        for (int s = 0; s < NBL; ++s) pm[s] = __ballot(a[s] >= theta && a[s] > 0.0f && a[s] >= lo_f);
      }
      // pack the chosen positions into the lanes (at most CAP; the rest waits for the next pass)
      unsigned base = 0;
#pragma unroll
      for (int s = 0; s < NBL; ++s) {
        const unsigned r = base + (unsigned)__popcll(pm[s] & lanemask_lt());
        const bool take = ((pm[s] >> lane) & 1ull) != 0ull && r < CAP;
        if (take) {
          cbuf[r] = (unsigned short)(bsel[s] * COOK_WAVE + lane);
          a[s] = 0.0f;  // dealt with
        }
        base += (unsigned)__popcll(pm[s]);
      }
      const unsigned ncand = base < CAP ? base : CAP;
      wave_sync();
      V3P_MARK(pc, pn, 3);
      // ---- exact evaluation: lane = candidate (records by position: a handful of cache lines per array) --------------------------------
      bool res = false, feas = false, zero = false, pv = false;
      double fit = -1.0;
      int off = -1;
      unsigned pos = 0u;
      if (lane < ncand) {
        const unsigned p = cbuf[lane];
        pos = p;
        OfferA oa_;
        oa_.oc = P.oc[p], oa_.om = P.om[p], oa_.rc = P.rc[p], oa_.rm = P.rm[p];
        const double ac = P.ac[p], am = P.am[p];
        OfferB o;
        o.host = P.host[p], o.gpu_model = P.gpu_model[p], o.gpu_count = P.gpu_count[p], o.run_count = P.run_count[p], o.task_slack = P.slack[p],
        o.flags = P.flags[p], o.pad = 0u;
        const int acount = P.acount[p];
        const unsigned v = P.off[p];
        const bool touched = ((ld_agent(&vb.G.tbits[p >> 6]) >> (p & 63u)) & 1ull) != 0ull;
        unsigned av[MV_NA];
#pragma unroll
        for (int x = 0; x < MV_NA; ++x) av[x] = 0u;
        if (fastc) {
#pragma unroll
          for (int x = 0; x < MV_NA; ++x) av[x] = P.attr[(size_t)x * V3_MMAX + p];
        }
        res = !(ac + j.c > oa_.oc || am + j.m > oa_.om);
        if (res) {
          bool ok = static_fast(j, o, in, v);
          if (ok && fastc) {
            unsigned diff = (E.req_host ^ (o.host + 1u)) & E.wild_host;
#pragma unroll
            for (int x = 0; x < MV_NA; ++x) diff |= (E.req[x] ^ av[x]) & E.wild[x];
            bool hit = E.impossible;
#pragma unroll
            for (int q = 0; q < MV_NC; ++q) hit = hit | (E.novel[q] == o.host);
            ok = diff == 0u && !hit;
          }
          if (ok && slow) ok = static_pass_dev(vb.in_dev, jj, v);
          if (ok) ok = dyn_fast(j, o, acount);
          if (ok && n_fh > 0) {
            bool taken = false;
#pragma unroll
            for (int q = 0; q < MV_FH; ++q) taken = taken | (fh[q] == o.host);
            ok = !taken;
          }
          if (ok && grouped && n_fh == -2) ok = group_pass_dev(vb.in_dev, st_cut, jj, v);
          if (ok) {
            fit = fitness_of(oa_, ac, am, j.c, j.m);
            if (fit > 0.0) {
              feas = true;
              off = (int)v;
              pv = !touched;  // a touched offer is the walker's business: it stays out of the list
            } else {
              zero = true;
            }
          }
        }
      }
      const unsigned n_res_p = (unsigned)__popcll(__ballot(res));
      n_res += n_res_p;
      n_feas += (unsigned)__popcll(__ballot(feas));
      n_zero += (unsigned)__popcll(__ballot(zero));
      if (n_res_p < ncand) any_nores = true;
      ++visits;
      V3P_MARK(pc, pn, 4);
      // ---- the list so far joins in the last lanes; rank the survivors; the best V3_L are the new list ----------------------------------
      if (lane >= CAP && lane - CAP < n_list) {
        const V3Ent x = J.ent[lane - CAP];
        fit = x.fit, off = x.off, pos = x.pos;
        pv = true;
      }
      const bool beats = pv && (n_list < (unsigned)V3_L || lane >= CAP || fit > t8 || (fit == t8 && off < t8_off));
      const unsigned long long sm = __ballot(beats);
      if (__any(pv && !beats)) more = true;  // a feasible untouched offer that did not make the list
      if ((sm & ((1ull << CAP) - 1ull)) != 0ull) {  // (some candidate enters: otherwise the list stays as it is)
        unsigned rank = 0;
        for (unsigned long long mm = sm; mm != 0ull; mm &= mm - 1ull) {
          const int src = __ffsll((unsigned long long)mm) - 1;
          const double bf = wave_read_lane_f64(fit, src);
          const int bo = wave_read_lane(off, src);
          rank += (beats && (int)lane != src && (bf > fit || (bf == fit && bo < off))) ? 1u : 0u;
          V3P_COUNT(pn, 8, 1u);
        }
        const unsigned n_surv = (unsigned)__popcll(sm);
        wave_sync();  // (the old entries have been read)
        if (beats && rank < (unsigned)V3_L) {
          V3Ent x;
          x.fit = fit, x.off = off, x.pos = pos;
          J.ent[rank] = x;
        }
        if (n_surv > (unsigned)V3_L) more = true;
        n_list = n_surv < (unsigned)V3_L ? n_surv : (unsigned)V3_L;
        if (n_list == (unsigned)V3_L) {
          const unsigned long long lm = __ballot(beats && rank == (unsigned)V3_L - 1u);
          const int src = __ffsll((unsigned long long)lm) - 1;
          t8 = wave_read_lane_f64(fit, src);
          t8_off = wave_read_lane(off, src);
        }
        wave_sync();
      }
      V3P_MARK(pc, pn, 5);
    }
  }
  // a list that is not full holds EVERY feasible untouched offer: the search only leaves positions out once the list is full
  const bool trunc = n_list == (unsigned)V3_L && more;
  // failure counts under the snapshot, as far as the walker needs them (exact when !trunc): does ANY offer fail on resources; offers
  // that have room but fail a constraint — exact up to V3_T, "more than V3_T" beyond; offers of zero fitness
  const unsigned c1 = any_nores ? 1u : 0u, c2 = n_res - n_feas - n_zero, c4 = n_zero;
  if (lane < (unsigned)V3_L && lane >= n_list) {
    V3Ent x;
    x.fit = -1.0, x.off = -1, x.pos = 0u;
    J.ent[lane] = x;
  }
  if (lane == 0) {
    J.c = j.c, J.m = j.m, J.g = j.g;
    J.k = k, J.jj = jj;
    J.gpu_model = j.gpu_model;
    J.reserved_host = j.reserved_host;
    J.group = j.group;
    const bool plain = !(j.g > 0) && !(j.g < 0) && !fastc && !slow && j.group == 0xFFFFFFFFu && j.reserved_host < 0;
    J.info = n_list | (trunc ? V3I_TRUNC : 0u) | ((!trunc && n_feas == 0u) ? V3I_NOFEAS : 0u) | (plain ? V3I_PLAIN : 0u) | (j.g > 0 ? V3I_GPU : 0u) | (grouped ? V3I_GROUPED : 0u) |
             (gtype << 18) | (j.group != 0xFFFFFFFFu ? V3I_HASGROUP : 0u) | (fastc ? V3I_FASTC : 0u) | (slow ? V3I_SLOW : 0u) |
             ((j.group != 0xFFFFFFFFu && gtype <= 1u && (gtype == 0u || n_fh >= 0)) ? V3I_GFAST : 0u);
    J.f1 = (unsigned short)(c1 < 0xFFFFu ? c1 : 0xFFFFu);
    J.f2 = (unsigned short)(c2 < 0xFFFFu ? c2 : 0xFFFFu);
    J.f4 = (unsigned short)(c4 < 0xFFFFu ? c4 : 0xFFFFu);
    J.pad = 0;
#pragma unroll
    for (int q = 0; q < MV_NA; ++q) J.req[q] = E.req[q], J.wild[q] = E.wild[q];
    J.req_host = E.req_host, J.wild_host = E.wild_host;
#pragma unroll
    for (int q = 0; q < MV_NC; ++q) J.novel[q] = E.novel[q];
    J.impossible = E.impossible ? 1u : 0u;
#pragma unroll
    for (int q = 0; q < MV_FH; ++q) J.gfh[q] = fh[q];
    J.n_fh = n_fh, J.glast = glast;
  }
  *steps_out = steps;
  *visits_out = visits;
  V3P_MARK(pc, pn, 6);
}

// ---- a helper wave (any workgroup but 0): take job positions, prepare them, publish them through the global ring ---------------------
// the workgroup's copy of the order: brought up to generation g by the first wave that notices (the others wait for it; a wave still
// searching under the old generation reads a torn copy, but its job belongs to an epoch that is over and is thrown away)
static __device__ __forceinline__ void v3_helper_refresh(V3HLds& H, const V3Buf& vb, unsigned g) {
  const unsigned lane = lane_id();
  const V3Glob& G = vb.G;
  unsigned got = 0;
  if (lane == 0) got = atomicCAS(&H.lock, 0u, 1u) == 0u ? 1u : 0u;
  got = (unsigned)__shfl((int)got, 0, COOK_WAVE);
  if (!got) {
    while (v3_ld_wg_u(&H.gen_loaded) != g) {
      EMU_SITE("v3 helper: waiting for the workgroup's copy");
      SPIN_PAUSE();
    }
    lds_acquire();
    return;
  }
  const unsigned have = v3_ld_wg_u(&H.gen_loaded);
  if (have != g) {
    agent_acquire();
    // the blocks that changed since generation `have` (everything when the log no longer reaches back that far)
    unsigned long long d0 = 0ull, d1 = 0ull;
    if (g - have > (unsigned)V3_DLOG || V3_FORCE_FULL) {
      d0 = d1 = ~0ull;
    } else {
      for (unsigned x = have + 1u; x != g + 1u; ++x) {
        d0 |= G.dlog[(x % (unsigned)V3_DLOG) * V3_BPL + 0];
        d1 |= G.dlog[(x % (unsigned)V3_DLOG) * V3_BPL + 1];
      }
    }
    const unsigned n_pos = G.hdr[0], nb = (n_pos + COOK_WAVE - 1) / COOK_WAVE;
    for (unsigned b = 0; b < nb; ++b) {
      if (!(((b < 64u ? d0 : d1) >> (b & 63u)) & 1ull)) continue;  // (wave-uniform)
      const unsigned p = b * COOK_WAVE + lane;
      H.fcm[p] = G.fcm[p];
      H.okey[p] = G.okey[p];
    }
    for (unsigned b = lane; b < nb; b += COOK_WAVE) H.bsum[b] = G.bsum[b];
    if (lane == 0) {
      H.n_pos = n_pos;
      H.n_blocks = nb;
      H.gen_first = G.hdr[1];
    }
    wave_sync();  // (every lane's part of the copy is in place)
    lds_release();
    if (lane == 0) st_wg(&H.gen_loaded, g);
  }
  if (lane == 0) st_wg(&H.lock, 0u);
}

static __device__ __forceinline__ void v3_helper(V3HLds& H, const MatchIn& in, const MatchState& st, const V3Buf& vb, unsigned my, unsigned* n_steps,
                                                 unsigned* n_visits, unsigned* n_settled, unsigned long long* pc, unsigned long long* pn) {
  const unsigned lane = lane_id();
  const unsigned K = in.K;
  const V3Glob& G = vb.G;
  const unsigned la = vb.look_ahead < 1u ? 1u : (vb.look_ahead > (unsigned)V3_GR ? (unsigned)V3_GR : vb.look_ahead);
  unsigned e_seen = 0u;
  V3Job& J = H.stage[wave_id()];
  const unsigned long long t_start = cook_ticks();
  for (;;) {
    // a job position of the current epoch (one fetch-and-add gives both)
    unsigned long long t = 0ull;
    if (lane == 0) {
      t = ld_agent(G.next64);
      if ((unsigned)(t >> 32) != 0u && (unsigned)(t >> 32) != V3_E_DONE) t = atomicAdd(G.next64, 1ull);
    }
    const unsigned e = (unsigned)__shfl((int)(unsigned)(t >> 32), 0, COOK_WAVE), p = (unsigned)__shfl((int)(unsigned)t, 0, COOK_WAVE);
    if (e == V3_E_DONE) return;
    if (e == 0u) {  // the walker workgroup has not started the first epoch yet
      if (cook_ticks() - t_start > V3_WAIT_TICKS) return;  // (it never came up: the launch was not co-resident; the host sees the walker's verdict)
      EMU_SITE("v3 helper: waiting for the first epoch");
      SPIN_PAUSE_FAR();
      continue;
    }
    if (e != e_seen) {
      e_seen = e;
      if (lane == 0) st_agent(&G.ack[my], e);
      const unsigned g = v3_ld_agent_u(G.gen);
      if (g != v3_ld_wg_u(&H.gen_loaded)) v3_helper_refresh(H, vb, g);
    }
    auto epoch_now = [&]() -> unsigned { return (unsigned)(v3_ld_agent_u64(G.next64) >> 32); };
    if (p >= K) {  // nothing left to prepare in this epoch
      while (epoch_now() == e) {
        EMU_SITE("v3 helper: idle");
        SPIN_PAUSE_FAR();
      }
      continue;
    }
    // not too far ahead of the walker
    bool stale = false;
    V3P_DECL();
    while (p - v3_ld_agent_u(G.walk_pos) >= la) {
      if (epoch_now() != e) {
        stale = true;
        break;
      }
      EMU_SITE("v3 helper: window");
      SPIN_PAUSE_FAR();
    }
    if (stale) continue;
    V3P_MARK(pc, pn, 7);
    unsigned steps = 0, visits = 0;
    v3_prepare(H, in, st, vb, p, J, &steps, &visits, pc, pn);
    wave_sync();
    *n_steps += steps;
    *n_visits += visits;
    // settled here, for good: no feasible offer under the snapshot (placements only take capacity away; unique groups only take hosts
    // away), every failure class backed by more offers than a generation can touch
    unsigned info = 0, c1 = 0, c2 = 0, c4 = 0;
    if (lane == 0) {
      info = J.info;
      c1 = J.f1, c2 = J.f2, c4 = J.f4;
    }
    info = (unsigned)__shfl((int)info, 0, COOK_WAVE);
    c1 = (unsigned)__shfl((int)c1, 0, COOK_WAVE);
    c2 = (unsigned)__shfl((int)c2, 0, COOK_WAVE);
    c4 = (unsigned)__shfl((int)c4, 0, COOK_WAVE);
    const bool trivial = (info & V3I_NOFEAS) != 0u && c1 > 0u && (c2 == 0u || c2 > (unsigned)V3_T) && c4 == 0u;
    if (epoch_now() != e) continue;  // (the epoch ended meanwhile: nobody will look at the entry)
    // (a settled job's failure summary travels in the flag: the walker workgroup writes every result, so that a straggler of an old
    //  epoch can never overwrite a newer verdict)
    const unsigned fbits = 1u | (c2 ? 2u : 0u) | (c4 ? 4u : 0u);
    if (trivial) *n_settled += 1u;
    // the entry goes out: sc1 stores, drained, then the flag
    const unsigned slot = (e % (unsigned)V3_NBANK) * (unsigned)V3_GR + p % (unsigned)V3_GR;
    if (!trivial && lane < (unsigned)V3_JOB_WORDS)
      st_agent(reinterpret_cast<unsigned long long*>(&G.ring[slot]) + lane, reinterpret_cast<const unsigned long long*>(&J)[lane]);
    drain_stores();
    wave_sync();  // (every lane's words are out)
    if (lane == 0) st_agent(&G.rflag[slot], (unsigned long long)(p + 1u) | ((unsigned long long)((e << 5) | (fbits << 2) | (trivial ? 2u : 1u)) << 32));
    wave_sync();
    V3P_MARK(pc, pn, 6);
  }
}

// ---- the feeder waves of the walker workgroup (waves 1 ..): global ring -> LDS ring, in job order ------------------------------------
// wave 1 + x takes the jobs q with q % n_feeders == x; the walker's own window (V3_R LDS slots) bounds how far they run ahead.
static __device__ __forceinline__ void v3_feeder(V3Lds& L, const MatchIn& in, const MatchState& st, const V3Buf& vb, unsigned epoch) {
  const unsigned lane = lane_id();
  const unsigned K = in.K;
  const V3Glob& G = vb.G;
  const unsigned nf = blockDim.x / COOK_WAVE - 1u, me = wave_id() - 1u;
  unsigned q = L.walk_pos;
  q += (me + nf - q % nf) % nf;  // first job >= walk_pos that is this wave's
  const unsigned bank = (epoch % (unsigned)V3_NBANK) * (unsigned)V3_GR;
  // feeder 0 also writes the walker's verdicts out (the walker itself never waits for a store to HBM) and tells the helpers where the
  // walker is; a ring slot is reused only once its verdict is out
  auto flush = [&](unsigned upto) {
    unsigned fl = L.flushed;  // (only this wave writes it)
    while (fl < upto) {
      const unsigned x = fl + lane;
      if (x < upto) {
        const unsigned code = L.res_fail[x % (unsigned)V3_R];
        if (code != 0xFFFFFFFFu) {
          st.job_to_offer[x] = L.res_j2o[x % (unsigned)V3_R];
          if (st.fail_code) st.fail_code[x] = code;
        }
      }
      fl = upto - fl > (unsigned)COOK_WAVE ? fl + (unsigned)COOK_WAVE : upto;
    }
    wave_sync();
    if (lane == 0) st_wg(&L.flushed, upto);
  };
  for (;;) {
    const unsigned stopped = v3_ld_wg_u(&L.gen_stop);
    const unsigned wp = v3_ld_wg_u(&L.walk_pos);
    if (me == 0u) {
      if (lane == 0) st_agent(G.walk_pos, wp);  // (the helpers' window follows the walker)
      if (v3_ld_wg_u(&L.flushed) < wp) flush(wp);
    }
    if (stopped != 0u) return;  // (the walker's last verdicts of the epoch are out: walk_pos was read after the stop flag)
    if (q >= K || q - v3_ld_wg_u(&L.flushed) >= (unsigned)V3_R) {
      EMU_SITE("v3 feeder: window");
      SPIN_PAUSE();
      continue;
    }
    const unsigned long long f = v3_ld_agent_u64(&G.rflag[bank + q % (unsigned)V3_GR]);
    const unsigned state = (unsigned)(f >> 32);
    if ((unsigned)f != q + 1u || (state >> 5) != epoch) {
      EMU_SITE("v3 feeder: waiting for a helper");
      SPIN_PAUSE_SHORT();
      continue;
    }
    if ((state & 3u) == 1u) {
      unsigned long long w = 0ull;
      if (lane < (unsigned)V3_JOB_WORDS) w = ld_agent(reinterpret_cast<const unsigned long long*>(&G.ring[bank + q % (unsigned)V3_GR]) + lane);
      if (lane < (unsigned)V3_JOB_WORDS) reinterpret_cast<unsigned long long*>(&L.ring[q % (unsigned)V3_R])[lane] = w;
      if (lane == 0) L.res_fail[q % (unsigned)V3_R] = 0xFFFFFFFFu;  // (the walker's verdict goes here)
    } else if (lane == 0) {
      L.res_j2o[q % (unsigned)V3_R] = -1;  // settled by the helper: its summary travelled in the flag
      L.res_fail[q % (unsigned)V3_R] = (state >> 2) & 7u;
    }
    wave_sync();  // (every lane's words are in the LDS ring)
    lds_release();
    if (lane == 0) st_wg(&L.rstate[q % (unsigned)V3_R], ((q + 1u) << 2) | (state & 3u));
    q += nf;
  }
}

// ---- the walker ------------------------------------------------------------------------------------------------------------------------------
struct V3Walker {  // the walking wave's registers: they live across epochs, a generation change writes them back
  int t_v;         // the touched offer of this lane, -1 none
  unsigned t_pos;  // its position in the order
  double t_oc, t_om, t_rc, t_rm, t_invc, t_invm;
  double t_ac, t_am, t_basec, t_basem, t_ac0, t_am0;
  int t_acount, t_acount0;
  OfferB t_o;
  bool t_plain_ok;  // the offer takes jobs without gpus / constraints / reserved host (static_fast for such a job)
  unsigned t_attr[MV_NA];
  unsigned lg_group, lg_host, n_log;  // group members placed in this generation (one per lane, in placement order)
  int lg_k;
  unsigned nT;
  unsigned matched, head_matched, walked, opens, fast;
  unsigned long long wait_ticks;
};
static __device__ __forceinline__ void v3_walker_reset(V3Walker& W) {
  W.t_v = -1;
  W.t_pos = 0u;
  W.t_oc = W.t_om = W.t_rc = W.t_rm = W.t_invc = W.t_invm = 0.0;
  W.t_ac = W.t_am = W.t_basec = W.t_basem = W.t_ac0 = W.t_am0 = 0.0;
  W.t_acount = W.t_acount0 = 0;
  W.t_o.host = 0, W.t_o.gpu_model = 0, W.t_o.gpu_count = 0.0, W.t_o.run_count = 0, W.t_o.task_slack = 0x7FFFFFFF, W.t_o.flags = 0, W.t_o.pad = 0;
  W.t_plain_ok = false;
#pragma unroll
  for (int x = 0; x < MV_NA; ++x) W.t_attr[x] = 0u;
  W.lg_group = 0xFFFFFFFFu, W.lg_host = 0u, W.n_log = 0u;
  W.lg_k = -1;
  W.nT = 0u;
}

// the touched offers' state back to HBM (by offer: the call's result; by position: the helpers' view), new keys and free resources
// in place, their blocks marked for new summaries
static __device__ __forceinline__ void v3_walker_writeback(V3Lds& L, const MatchState& st, const V3Buf& vb, V3Walker& W) {
  const V3Pos& P = vb.P;
  if (W.t_v >= 0) {
    st.ac[W.t_v] = W.t_ac;
    st.am[W.t_v] = W.t_am;
    st.acount[W.t_v] = W.t_acount;
    P.ac[W.t_pos] = W.t_ac;
    P.am[W.t_pos] = W.t_am;
    P.acount[W.t_pos] = W.t_acount;
    L.owner[W.t_v] = 0xFF;
    if (W.t_ac + st.jmin[0] > W.t_oc || W.t_am + st.jmin[1] > W.t_om) {
      atomicAnd(&st.alive[(unsigned)W.t_v >> 6], ~(1ull << ((unsigned)W.t_v & 63u)));
      L.okey[W.t_pos] = -1.0f;
      L.fcm[W.t_pos] = 0u;
    } else {
      OfferA a;
      a.oc = W.t_oc, a.om = W.t_om, a.rc = W.t_rc, a.rm = W.t_rm, a.inv_dc = W.t_invc, a.inv_dm = W.t_invm;
      L.okey[W.t_pos] = v3_key_f32(v3_sort_key(a, W.t_o, W.t_ac, W.t_am, (unsigned)W.t_v));
      L.fcm[W.t_pos] = v3_pack_free(W.t_oc - W.t_ac, W.t_om - W.t_am);
    }
    const unsigned b = W.t_pos >> 6;
    atomicOr(&L.dirty[b >> 6], 1ull << (b & 63u));
    L.tbits[b] = 0ull;
  }
  v3_walker_reset(W);
}

// One epoch of the walk: until the jobs are used up (L.done), a list ran out (stop 1), the touched set (2) or the group log (3) is full.
static __device__ __forceinline__ void v3_walk(V3Lds& L, const MatchIn& in, MatchState st, const V3Buf& vb, V3Walker& W, unsigned long long* pc,
                                               unsigned long long* pn) {
  const unsigned lane = lane_id();
  const unsigned K = in.K;
  unsigned p = L.walk_pos;
  constexpr double EPS_HI = 1.0 + 0x1p-38, EPS_LO = 1.0 - 0x1p-38;
  unsigned stop = 0;
  // The walk is a software pipeline over the LDS ring: the record of job p + 2 and the owner look-up of job p + 1 are in flight
  // while job p is decided.  A record is read speculatively, its state word first (LDS operations retire in order): it is used
  // only if every lane saw "ready" for exactly that job; otherwise the pipeline is refilled the slow way (scan, wait).
  struct V3Rec {
    unsigned state, info;
    double c, m;
    double e_fit;
    int e_off;
    unsigned e_pos, owner;
  };
  auto load_rec = [&](unsigned q) -> V3Rec {
    V3Rec r;
    const V3Job& Q = L.ring[q % (unsigned)V3_R];
    r.state = L.rstate[q % (unsigned)V3_R];
    asm volatile("" ::: "memory");  // (the compiler keeps the state read in front of the record's)
    r.info = Q.info;
    r.c = Q.c, r.m = Q.m;
    r.e_fit = -1.0, r.e_off = -1, r.e_pos = 0u, r.owner = 0xFEu;
    if (lane < (unsigned)V3_L) {
      const V3Ent x = Q.ent[lane];
      r.e_fit = x.fit, r.e_off = x.off, r.e_pos = x.pos;
    }
    return r;
  };
  auto load_owner = [&](V3Rec& r) {
    if (lane < (unsigned)V3_L && r.e_off >= 0 && r.e_off < V3_MMAX) r.owner = L.owner[r.e_off];
  };
  bool have = false;
  V3Rec cur, nxt;
  cur.state = cur.info = 0u, cur.c = cur.m = 0.0, cur.e_fit = -1.0, cur.e_off = -1, cur.e_pos = 0u, cur.owner = 0xFEu;
  nxt = cur;
  while (p < K) {
    V3P_DECL();
    if (!have) {
      // ---- skip the jobs the helpers settled; wait for the next prepared one ----------------------------------------------------------
      const unsigned long long t0 = cook_ticks();
      bool waited = false;
      for (;;) {
        const unsigned q = p + lane;
        unsigned s = 0;
        if (lane < (unsigned)V3_R && q < K) s = ld_wg(&L.rstate[q % (unsigned)V3_R]);
        const bool mine = (s >> 2) == q + 1u;
        const unsigned long long settled = __ballot(mine && (s & 3u) == 2u), ready = __ballot(mine && (s & 3u) == 1u);
        const unsigned run = settled == ~0ull ? 64u : (unsigned)__ffsll((unsigned long long)~settled) - 1u;
        if (run > 0u) {
          p += run;
          if (lane == 0) st_wg(&L.walk_pos, p);
          if (p >= K) break;
          continue;
        }
        if (ready & 1ull) break;
        waited = true;
        if (cook_ticks() - t0 > V3_WAIT_TICKS) {
          if (lane == 0) {
            st_wg(&L.abort, 1u);
            st_wg(&L.done, 1u);
            st_wg(&L.gen_stop, 1u);
          }
          return;
        }
        EMU_SITE("v3 walker: waiting for a prepared job");
        SPIN_PAUSE_SHORT();
      }
      if (waited) W.wait_ticks += cook_ticks() - t0;
      if (p >= K) break;
      lds_acquire();
      wave_sync();
      cur = load_rec(p);
      load_owner(cur);
      nxt = load_rec(p + 1u);
    }
    V3Rec nn = load_rec(p + 2u);  // in flight while job p is decided
    load_owner(nxt);
    V3P_MARK(pc, pn, 16);
    const V3Job& J = L.ring[p % (unsigned)V3_R];
    const unsigned info = wave_uniform_u32(cur.info), k = p;
    const double c = cur.c, m = cur.m;
    const int nc = (int)(info & 0xFFu);
    const bool grouped = (info & V3I_GROUPED) != 0u, has_group = (info & V3I_HASGROUP) != 0u;
    const unsigned g = has_group ? wave_uniform_u32(J.group) : 0xFFFFFFFFu;
    const bool trunc = (info & V3I_TRUNC) != 0u;
    W.walked += 1u;
    // list entry `lane` (unused entries hold offer -1)
    const double e_fit = cur.e_fit;
    const int e_off = cur.e_off;
    const unsigned e_pos = cur.e_pos, owner = cur.owner;
    // ---- every touched offer under the current state ----------------------------------------------------------------------------------
    const bool t_on = W.t_v >= 0;
    const bool res_ok = t_on && !(W.t_ac + c > W.t_oc || W.t_am + m > W.t_om);
    bool con_ok;
    unsigned long long ghits = 0ull;  // log entries of this job's group
    bool g_general = false;            // the group check goes through the chains in HBM
    JobRec jr;
    jr.c = c, jr.m = m, jr.g = 0.0, jr.gpu_model = 0u, jr.reserved_host = -1, jr.group = 0xFFFFFFFFu, jr.flags = 0;
    if (info & V3I_PLAIN) {  // (wave-uniform) nothing of the job's own to check: the lane knows whether it takes such jobs
      con_ok = t_on && W.t_plain_ok && W.t_acount < W.t_o.task_slack;
    } else {
      jr.g = J.g, jr.gpu_model = J.gpu_model, jr.reserved_host = J.reserved_host, jr.group = J.group;
      con_ok = t_on && static_fast(jr, W.t_o, in, (unsigned)(t_on ? W.t_v : 0)) && dyn_fast(jr, W.t_o, W.t_acount);
      if (info & V3I_FASTC) {  // (wave-uniform)
        unsigned diff = (J.req_host ^ (W.t_o.host + 1u)) & J.wild_host;
#pragma unroll
        for (int x = 0; x < MV_NA; ++x) diff |= (J.req[x] ^ W.t_attr[x]) & J.wild[x];
        bool hit = J.impossible != 0u;
#pragma unroll
        for (int q = 0; q < MV_NC; ++q) hit = hit | (J.novel[q] == W.t_o.host);
        con_ok = con_ok && diff == 0u && !hit;
      }
      if ((info & V3I_SLOW) && con_ok) con_ok = static_pass_dev(vb.in_dev, J.jj, (unsigned)W.t_v);
      if (has_group) {
        ghits = __ballot(lane < W.n_log && W.lg_group == g);
        if (grouped) {
          if ((info & V3I_GFAST) && J.n_fh >= 0 && W.n_log <= (unsigned)COOK_WAVE) {
            bool forb = false;
#pragma unroll
            for (int q = 0; q < MV_FH; ++q) forb = forb | (W.t_o.host == J.gfh[q]);
            for (unsigned long long hm = ghits; hm != 0ull; hm &= hm - 1ull) {
              const unsigned h = (unsigned)wave_read_lane((int)W.lg_host, __ffsll((unsigned long long)hm) - 1);
              forb = forb | (W.t_o.host == h);
            }
            con_ok = con_ok && !forb;
          } else {
            g_general = true;
            if (con_ok) con_ok = group_pass_dev(vb.in_dev, st, J.jj, (unsigned)W.t_v);
          }
        }
      }
    }
    const double nc_ = W.t_basec + c, nm_ = W.t_basem + m;
    const double a1 = nc_ * W.t_invc, a2 = nm_ * W.t_invm;
    const double fa = (a1 + a2) * 0.5;
    const bool cand = res_ok && con_ok;
    int win = -1, win_lane = -1;
    unsigned win_pos = 0u;
    bool exhausted = false;
    unsigned pe_bits = 8u;
    double pe_fit = 0.0;
    bool decided = false;
    V3P_MARK(pc, pn, 17);
    // ======== FAST PATH: the touched offers ordered by an fp32 image of the approximate fitness (one DPP max chain) ==================
    {
      const bool sane = a1 >= 0.0 && a2 >= 0.0 && fa > 0x1p-100;
      const float kf = cand ? (sane ? (float)fa : __int_as_float(0x7F800000)) : 0.0f;
      const float mx = wave_max_f32(kf);
      const unsigned long long untouched_mask = __ballot(owner == 0xFFu);
      double u_fit = -1.0;
      int u_off = -1;
      unsigned u_pos = 0u;
      if (untouched_mask != 0ull) {
        const int qs = __ffsll((unsigned long long)untouched_mask) - 1;
        u_fit = wave_read_lane_f64(e_fit, qs);
        u_off = wave_read_lane(e_off, qs);
        u_pos = (unsigned)wave_read_lane((int)e_pos, qs);
      }
      if (mx == 0.0f) {  // no touched offer can take the job
        if (u_off >= 0) {
          win = u_off;
          win_pos = u_pos;
          decided = true;
        }  // else: unmatched or list exhausted -> general path
      } else if (mx < __int_as_float(0x7F800000)) {
        const unsigned long long near = __ballot(kf >= mx * (1.0f - 0x1p-20f));
        if ((near & (near - 1ull)) == 0ull) {  // one touched offer clearly ahead of the other touched ones
          const int wl = __ffsll((unsigned long long)near) - 1;
          const double fw = wave_read_lane_f64(fa, wl);
          if (u_off < 0) {
            // no untouched entry: fine unless untouched offers may exist beyond the list and none of its entries is still a candidate
            bool ok = !trunc;
            if (!ok) {
              const unsigned long long cand_mask = __ballot(cand);
              const bool e_live = owner < 0xFEu && ((cand_mask >> (owner & 63u)) & 1ull);
              ok = __any(e_live);
            }
            if (ok) {
              win_lane = wl;
              decided = true;
            }
          } else if (fw * EPS_LO > u_fit) {
            win_lane = wl;
            decided = true;
          } else if (fw * EPS_HI < u_fit) {
            win = u_off;
            win_pos = u_pos;
            decided = true;
          }
        }
      }
      if (decided) W.fast += 1u;
    }
    V3P_MARK(pc, pn, 18);
    // ======== GENERAL PATH ===================================================================================================================
    if (!decided) {
      const bool sane = a1 >= 0.0 && a2 >= 0.0 && fa > 0.0;
      bool need_exact = __any(cand && !sane);
      const unsigned long long cand_mask = __ballot(cand);
      double u_fit = -1.0;
      int u_off = -1;
      unsigned u_pos = 0u;
      do {
        // no feasible offer under the snapshot and none of zero fitness (a placement could lift that one): stays unmatched, only the
        // summary may move
        if ((info & V3I_NOFEAS) && !grouped && J.f4 == 0) break;
        const bool e_valid = owner != 0xFEu, e_untouched = owner == 0xFFu;
        const bool e_live = e_valid && !e_untouched && ((cand_mask >> (owner & 63u)) & 1ull);
        const unsigned long long settle_mask = __ballot(e_untouched || e_live), untouched_mask = __ballot(e_untouched);
        if (settle_mask == 0ull && trunc) {
          exhausted = true;
          break;
        }
        if (settle_mask != 0ull) {
          const int qs = __ffsll((unsigned long long)settle_mask) - 1;
          if ((untouched_mask >> qs) & 1ull) {
            u_fit = wave_read_lane_f64(e_fit, qs);
            u_off = wave_read_lane(e_off, qs);
            u_pos = (unsigned)wave_read_lane((int)e_pos, qs);
          }
        }
        bool dec2 = false;
        if (!need_exact) {
          if (cand_mask == 0ull) {
            win = u_off;
            win_pos = u_pos;
            dec2 = true;
          } else {
            const unsigned long long key = cand ? (unsigned long long)__double_as_longlong(fa) : 0ull;
            const double mx = __longlong_as_double((long long)wave_max_u64(key));
            const unsigned long long near = __ballot(cand && fa >= mx * EPS_LO);
            if ((near & (near - 1ull)) == 0ull) {
              if (u_off < 0 || mx * EPS_LO > u_fit) {
                win_lane = __ffsll((unsigned long long)near) - 1;
                dec2 = true;
              } else if (mx * EPS_HI < u_fit) {
                win = u_off;
                win_pos = u_pos;
                dec2 = true;
              }
            }
          }
          if (!dec2) need_exact = true;
        }
        if (need_exact) {
          if (t_on) {
            pe_bits = 0u;
            if (!res_ok) {
              pe_bits = 1u;
            } else if (!con_ok) {
              pe_bits = 2u;
            } else {
              pe_fit = (nc_ / (W.t_oc + W.t_rc) + nm_ / (W.t_om + W.t_rm)) / 2.0;
              if (!(pe_fit > 0.0)) pe_bits = 4u;
            }
          }
          const bool t_feas = t_on && pe_bits == 0u;
          const unsigned long long feas_mask = __ballot(t_feas);
          const bool e_live2 = e_valid && !e_untouched && ((feas_mask >> (owner & 63u)) & 1ull);
          const unsigned long long settle2 = __ballot(e_untouched || e_live2);
          if (settle2 == 0ull && trunc) {
            exhausted = true;
            break;
          }
          u_fit = -1.0;
          u_off = -1;
          if (settle2 != 0ull) {
            const int qs = __ffsll((unsigned long long)settle2) - 1;
            if ((untouched_mask >> qs) & 1ull) {
              u_fit = wave_read_lane_f64(e_fit, qs);
              u_off = wave_read_lane(e_off, qs);
              u_pos = (unsigned)wave_read_lane((int)e_pos, qs);
            }
          }
          Cand best{-1.0, -1};
          int best_lane = -1;
          if (feas_mask != 0ull) {
            const unsigned long long key = t_feas ? (unsigned long long)__double_as_longlong(pe_fit) : 0ull;
            const unsigned long long mx = wave_max_u64(key);
            unsigned long long tie = __ballot(t_feas && key == mx);
            int wl = __ffsll((unsigned long long)tie) - 1;
            int wv = wave_read_lane(W.t_v, wl);
            tie &= tie - 1ull;
            while (tie != 0ull) {
              const int l2 = __ffsll((unsigned long long)tie) - 1;
              const int v2 = wave_read_lane(W.t_v, l2);
              if (v2 < wv) wv = v2, wl = l2;
              tie &= tie - 1ull;
            }
            best = Cand{__longlong_as_double((long long)mx), wv};
            best_lane = wl;
          }
          if (u_off >= 0 && cand_better(Cand{u_fit, u_off}, best))
            win = u_off, win_pos = u_pos, win_lane = -1;
          else if (best_lane >= 0)
            win_lane = best_lane, win = -1;
          else
            win = win_lane = -1;
        }
      } while (0);
    }
    if (!decided || exhausted) V3P_MARK(pc, pn, 19);
    if (exhausted) {
      stop = 1;
      break;
    }
    const bool prof_new_ = win_lane < 0 && win >= 0;
    (void)prof_new_;
    // ---- commit ---------------------------------------------------------------------------------------------------------------------------
    if (win_lane < 0 && win >= 0 && W.nT == (unsigned)V3_T) {
      stop = 2;  // no free lane for another touched offer: a new generation starts with this job
      break;
    }
    if ((win_lane >= 0 || win >= 0) && has_group && W.n_log >= (unsigned)COOK_WAVE) {
      stop = 3;  // the log of this generation's group placements is full
      break;
    }
    if (win_lane >= 0) {
      if ((int)lane == win_lane) {
        W.t_ac += c;
        W.t_am += m;
        W.t_acount += 1;
        W.t_basec = W.t_rc + W.t_ac;
        W.t_basem = W.t_rm + W.t_am;
      }
      win = wave_read_lane(W.t_v, win_lane);
    } else if (win >= 0) {
      if (lane == W.nT) {  // the next free lane takes ownership: the offer's records by position (untouched: the snapshot is its state)
        const V3Pos& P = vb.P;
        const unsigned q = win_pos;
        W.t_v = win;
        W.t_pos = q;
        W.t_oc = P.oc[q], W.t_om = P.om[q], W.t_rc = P.rc[q], W.t_rm = P.rm[q];
        W.t_ac0 = P.ac[q], W.t_am0 = P.am[q], W.t_acount0 = P.acount[q];
        W.t_o.host = P.host[q], W.t_o.gpu_model = P.gpu_model[q], W.t_o.gpu_count = P.gpu_count[q], W.t_o.run_count = P.run_count[q],
        W.t_o.task_slack = P.slack[q], W.t_o.flags = P.flags[q], W.t_o.pad = 0u;
#pragma unroll
        for (int x = 0; x < MV_NA; ++x) W.t_attr[x] = P.attr[(size_t)x * V3_MMAX + q];
        W.t_invc = 1.0 / (W.t_oc + W.t_rc), W.t_invm = 1.0 / (W.t_om + W.t_rm);  // (as match_pack_offers computes OfferA::inv_dc / inv_dm)
        W.t_plain_ok = ((W.t_o.flags & 1u) ? W.t_o.gpu_model == 0u : true) && !(W.t_o.flags & 2u);
        W.t_ac = W.t_ac0 + c;
        W.t_am = W.t_am0 + m;
        W.t_acount = W.t_acount0 + 1;
        W.t_basec = W.t_rc + W.t_ac;
        W.t_basem = W.t_rm + W.t_am;
        L.owner[win] = (unsigned char)W.nT;
        const unsigned long long tb_ = L.tbits[win_pos >> 6] | (1ull << (win_pos & 63u));  // (this wave is the only writer)
        L.tbits[win_pos >> 6] = tb_;
        st_agent(&vb.G.tbits[win_pos >> 6], tb_);  // the helpers leave touched offers out of their lists
      }
      // the owner look-up of the next job was issued before this commit: patch it
      if (nxt.owner == 0xFFu && nxt.e_off == win) nxt.owner = W.nT;
      win_lane = (int)W.nT;
      ++W.nT;
      W.opens += 1u;
      wave_sync();  // the owner table update is visible to the whole wave before the next look-up reads it
    }
    if (win >= 0) {
      W.matched += 1u;
      if (k == 0) W.head_matched = 1u;
      if (has_group) {  // publish a placed group member: the chains in HBM and the generation's log
        const int prev = ghits != 0ull ? wave_read_lane(W.lg_k, 63 - __clzll((long long)ghits)) : J.glast;
        const unsigned w_host = (unsigned)wave_read_lane((int)W.t_o.host, win_lane);
        if (lane == 0) {
          st_agent(&st.job_to_offer[k], win);
          st_agent(&st.job_prev[k], prev);
          st_agent(&st.group_last[g], (int)k);
        }
        if (lane == W.n_log) {
          W.lg_group = g;
          W.lg_host = w_host;
          W.lg_k = (int)k;
        }
        ++W.n_log;
        if (g_general) wave_sync();
      }
      if (lane == 0) {  // (the verdict goes to HBM behind the walker: a feeder wave writes the ring's results out)
        L.res_j2o[k % (unsigned)V3_R] = win;
        L.res_fail[k % (unsigned)V3_R] = 0u;
      }
    } else {
      // unmatched: failure summary = OR over offers of the first failing check under the CURRENT state: the snapshot counts, with each
      // touched offer's snapshot verdict swapped for its current one.  A touched offer only got fuller, so "fails on resources" can
      // only have been added; the other two classes need the snapshot verdicts of the touched offers only when the snapshot count
      // is small enough for the touched offers to matter (0 < count <= V3_T)
      const unsigned f1 = J.f1, f2 = J.f2, f4 = J.f4;
      const unsigned long long now1 = __ballot(t_on && !res_ok), now2 = __ballot(t_on && res_ok && !con_ok);
      unsigned long long now4 = 0ull;  // passes both, fitness not positive
      if (__any(t_on && res_ok && con_ok)) {
        const double pf = (nc_ / (W.t_oc + W.t_rc) + nm_ / (W.t_om + W.t_rm)) / 2.0;
        now4 = __ballot(t_on && res_ok && con_ok && !(pf > 0.0));
      }
      unsigned bits = ((f1 > 0u || now1 != 0ull) ? 1u : 0u);
      const bool exact2 = f2 > 0u && f2 <= (unsigned)V3_T, exact4 = f4 > 0u && f4 <= (unsigned)V3_T;
      if (W.nT != 0u && (exact2 || exact4)) {
        if (info & V3I_PLAIN) jr.g = 0.0;  // (jr holds the plain job's values already)
        unsigned p0 = 0u;
        if (t_on) {
          if (W.t_ac0 + c > W.t_oc || W.t_am0 + m > W.t_om) {
            p0 = 1u;
          } else {
            bool ok = static_fast(jr, W.t_o, in, (unsigned)W.t_v) && dyn_fast(jr, W.t_o, W.t_acount0);
            if (ok && (info & V3I_FASTC)) {
              unsigned diff = (J.req_host ^ (W.t_o.host + 1u)) & J.wild_host;
#pragma unroll
              for (int x = 0; x < MV_NA; ++x) diff |= (J.req[x] ^ W.t_attr[x]) & J.wild[x];
              bool hit = J.impossible != 0u;
#pragma unroll
              for (int q = 0; q < MV_NC; ++q) hit = hit | (J.novel[q] == W.t_o.host);
              ok = diff == 0u && !hit;
            }
            if (ok && (info & V3I_SLOW)) ok = static_pass_dev(vb.in_dev, J.jj, (unsigned)W.t_v);
            if (ok && grouped) {  // the group check as the generation began
              MatchState st0 = st;
              st0.cutoff = (int)L.gen_first;
              ok = group_pass_dev(vb.in_dev, st0, J.jj, (unsigned)W.t_v);
            }
            if (!ok) {
              p0 = 2u;
            } else {
              const double f0 = ((W.t_rc + W.t_ac0 + c) / (W.t_oc + W.t_rc) + (W.t_rm + W.t_am0 + m) / (W.t_om + W.t_rm)) / 2.0;
              if (!(f0 > 0.0)) p0 = 4u;
            }
          }
        }
        const int d2 = __popcll(now2) - __popcll(__ballot(t_on && (p0 & 2u))), d4 = __popcll(now4) - __popcll(__ballot(t_on && (p0 & 4u)));
        bits |= (((int)f2 + d2) > 0 ? 2u : 0u) | (((int)f4 + d4) > 0 ? 4u : 0u);
      } else {
        // no snapshot count in the delicate range: a zero count can only grow by what the touched offers show now, a large one
        // cannot be used up by them
        bits |= ((f2 > (unsigned)V3_T || (f2 == 0u && now2 != 0ull) || (exact2 && W.nT == 0u)) ? 2u : 0u) |
                ((f4 > (unsigned)V3_T || (f4 == 0u && now4 != 0ull) || (exact4 && W.nT == 0u)) ? 4u : 0u);
      }
      if (lane == 0) {
        L.res_j2o[k % (unsigned)V3_R] = -1;
        L.res_fail[k % (unsigned)V3_R] = bits ? bits : 8u;
      }
    }
    ++p;
    if (lane == 0) st_wg(&L.walk_pos, p);
    if (win < 0) {  // (constant indices: the counters stay in registers)
      V3P_MARK(pc, pn, 22);
    } else if (prof_new_) {
      V3P_MARK(pc, pn, 21);
    } else {
      V3P_MARK(pc, pn, 20);
    }
    // ---- the pipeline moves on: is the record read ahead the next job, and was it ready when it was read? -----------------------------
    have = __all((nxt.state >> 2) == p + 1u && (nxt.state & 3u) == 1u) != 0;
    cur = nxt;
    nxt = nn;
  }
  // ---- end of the epoch -----------------------------------------------------------------------------------------------------------------------
  if (stop >= 2u || p >= K) v3_walker_writeback(L, st, vb, W);  // (at the end of the call too: the state arrays are the call's result)
  if (lane == 0) {
    st_wg(&L.walk_pos, p);
    if (stop >= 2u) st_wg(&L.gen_first, p);
    st_wg(&L.stop_reason, stop);
    if (p >= K) st_wg(&L.done, 1u);
    lds_release();
    st_wg(&L.gen_stop, 1u);
  }
}

struct PoolCtx3 {
  MatchIn in;
  MatchState st;
  V3Buf vb;
};

// ---- one launch per match call: workgroup 0 walks (wave 0) and feeds the walk from the global ring (waves 1 ..), every other
// workgroup is helper waves.  (The context travels BY VALUE: pointers that arrive as kernel arguments are known to be global memory,
// pointers loaded from a context record in memory are not, and every access through them would be a flat one.)
constexpr size_t V3_LDS_BYTES = sizeof(V3Lds) > sizeof(V3HLds) ? sizeof(V3Lds) : sizeof(V3HLds);
__global__ void __launch_bounds__(V3_THREADS) match_v3(const PoolCtx3 C) {
  COOK_BLOCK_LDS(lds, V3_LDS_BYTES);
  const MatchIn& in = C.in;
  const MatchState st = C.st;
  const V3Buf vb = C.vb;
  const V3Glob& G = vb.G;
  const unsigned tid = threadIdx.x, NT = blockDim.x, lane = lane_id();
  const unsigned long long tk0 = cook_ticks();
  unsigned long long pc[V3_NPROF], pn[V3_NPROF];
  for (int i = 0; i < V3_NPROF; ++i) pc[i] = pn[i] = 0ull;
  if (blockIdx.x != 0u) {
    // ======== a helper workgroup ==================================================================================================
    V3HLds& H = *reinterpret_cast<V3HLds*>(lds);
    if (tid == 0) {
      H.gen_loaded = 0u;  // (generation numbers start at 1)
      H.lock = 0u;
      H.n_pos = H.n_blocks = H.gen_first = 0u;
    }
    __syncthreads();
    unsigned n_steps = 0, n_visits = 0, n_settled = 0;
    v3_helper(H, in, st, vb, (blockIdx.x - 1u) * (NT / COOK_WAVE) + wave_id(), &n_steps, &n_visits, &n_settled, pc, pn);
    if (lane == 0) {
      atomicAdd(&vb.ctl->scan_steps, n_steps);
      atomicAdd(&vb.ctl->visits, n_visits);
      atomicAdd(&vb.ctl->settled, n_settled);
#ifdef COOK_V3_PROF
      for (int i = 0; i < V3_NPROF; ++i) {
        if (pc[i]) atomicAdd(&vb.ctl->prof_cyc[i], pc[i]);
        if (pn[i]) atomicAdd(&vb.ctl->prof_cnt[i], pn[i]);
      }
#endif
    }
    return;
  }
  // ======== the walker workgroup ======================================================================================================
  V3Lds& L = *reinterpret_cast<V3Lds*>(lds);
  auto finish = [&](unsigned err) {  // tell the helpers the call is over (err != 0: before anything was placed)
    if (tid == 0) {
      if (err) vb.ctl->error = err;
      st_agent(G.next64, (unsigned long long)V3_E_DONE << 32);
    }
  };
  // ---- refuse what the bounds do not cover (the host then runs match_v2): negative / non-finite resources ------------------------------
  if (tid == 0) {
    L.gen_stop = 0u;
    L.done = 0u;
    L.next = 0u;
    L.walk_pos = 0u;
    L.gen_first = 0u;
    L.abort = 0u;
    L.stop_reason = 0u;
    L.flushed = 0u;
    L.sort_n = vb.job_flags[2] != 0ull ? 1u : 0u;  // (borrowed as the "bad input" flag until the first generation)
  }
  __syncthreads();
  {
    bool bad = false;
    for (unsigned v = tid; v < in.M; v += NT) {
      const OfferA a = vb.oa[v];
      const double dc = a.oc + a.rc, dm = a.om + a.rm;
      if (!(a.oc >= 0.0 && a.om >= 0.0 && a.rc >= 0.0 && a.rm >= 0.0 && dc > 0.0 && dm > 0.0 && dc < 1e300 && dm < 1e300)) bad = true;
    }
    if (__any(bad) && lane == 0) atomicOr(&L.sort_n, 1u);
  }
  __syncthreads();
  if (L.sort_n != 0u || in.M > (unsigned)V3_MMAX || NT < 2u * COOK_WAVE || vb.n_helper_waves == 0u || vb.n_helper_waves > (unsigned)V3_HW_MAX) {
    finish(1u);
    return;
  }
  __syncthreads();
  unsigned gens = 0, epochs = 0, s_full = 0, s_list = 0, s_log = 0;
  unsigned long long t_regen = 0ull, t_flush = 0ull;
  const unsigned rebuild_gens = vb.rebuild_gens < 1u ? 1u : vb.rebuild_gens;
  unsigned w_matched = 0, w_head = 0, w_walked = 0, w_opens = 0, w_fast = 0;
  unsigned long long w_wait = 0ull;
  // between epochs: a new generation (order rebuilt, or the touched offers' keys replaced in place) or only a ring flush; what the
  // helpers need goes out to global memory, then the next epoch is opened for them.  Ends with a barrier.
  auto between = [&](unsigned reason) {
    const unsigned long long tr0 = cook_ticks();
    if (reason >= 2u) {
      for (unsigned x = tid; x < in.G; x += NT) vb.group_snap[x] = ld_agent(&st.group_last[x]);
      const bool rebuild = gens % rebuild_gens == 0u;
      if (rebuild) {
        v3_build(L, in, st, vb);  // (ends with a barrier)
      } else {
        v3_summaries(L, vb, false);
      }
      ++gens;
      // the helpers' view: changed blocks of fcm / okey, every summary, the touched bits cleared, the header, the log of what changed
      const unsigned n = L.n_pos, nb = (n + COOK_WAVE - 1) / COOK_WAVE;
      const unsigned long long d0 = rebuild ? ~0ull : L.dirty[0], d1 = rebuild ? ~0ull : L.dirty[1];
      for (unsigned pp = tid; pp < nb * COOK_WAVE; pp += NT) {
        const unsigned b = pp >> 6;
        if (!(((b < 64u ? d0 : d1) >> (b & 63u)) & 1ull)) continue;
        G.fcm[pp] = pp < n ? L.fcm[pp] : 0u;
        G.okey[pp] = pp < n ? L.okey[pp] : -1.0f;
      }
      for (unsigned b = tid; b < (unsigned)V3_NBMAX; b += NT) {
        if (b < nb) G.bsum[b] = L.bsum[b];
        st_agent(&G.tbits[b], 0ull);
      }
      if (tid == 0) {
        G.hdr[0] = n;
        G.hdr[1] = L.gen_first;
        G.dlog[(gens % (unsigned)V3_DLOG) * V3_BPL + 0] = d0;
        G.dlog[(gens % (unsigned)V3_DLOG) * V3_BPL + 1] = d1;
      }
      __syncthreads();  // (every thread's stores have left the CU)
      if (tid < (unsigned)V3_BPL) L.dirty[tid] = 0ull;
      if (tid == 0) {
        agent_release();
        st_agent(G.gen, gens);
      }
    }
    for (unsigned x = tid; x < (unsigned)V3_R; x += NT) L.rstate[x] = 0u;
    ++epochs;
    if (tid == 0) {
      L.next = L.walk_pos;
      L.gen_stop = 0u;
      // no helper may still be working for the epoch that used this bank last (epochs - V3_NBANK): everyone has taken a job since
      if (epochs > (unsigned)V3_NBANK) {
        const unsigned long long t0 = cook_ticks();
        for (unsigned h = 0; h < vb.n_helper_waves; ++h)
          while (ld_agent(&G.ack[h]) + (unsigned)V3_NBANK <= epochs) {
            if (cook_ticks() - t0 > V3_WAIT_TICKS) {
              L.abort = 1u;
              break;
            }
            EMU_SITE("v3 walker workgroup: waiting for a straggling helper");
            SPIN_PAUSE();
          }
      }
      st_agent(G.walk_pos, L.walk_pos);
      drain_stores();
      st_agent(G.next64, ((unsigned long long)epochs << 32) | (unsigned long long)L.walk_pos);  // the helpers may go
    }
    __syncthreads();
    if (reason >= 2u)
      t_regen += cook_ticks() - tr0;
    else
      t_flush += cook_ticks() - tr0;
  };
  auto after = [&]() -> unsigned {  // -> why the epoch ended
    __syncthreads();
    const unsigned reason = L.stop_reason;
    s_list += reason == 1u ? 1u : 0u;
    s_full += reason == 2u ? 1u : 0u;
    s_log += reason == 3u ? 1u : 0u;
    return reason;
  };
  // (one loop for both roles of the workgroup: the walker's lanes are dead weight in the feeders' registers, but the feeders' code is
  //  small, and the generation change — the bulk of the code — exists once)
  V3Walker W;
  v3_walker_reset(W);
  W.matched = W.head_matched = W.walked = W.opens = W.fast = 0u;
  W.wait_ticks = 0ull;
  unsigned reason = 2u;  // (the first epoch builds the order)
  for (;;) {
    between(reason);
    if (L.abort != 0u) break;
    if (tid < COOK_WAVE)
      v3_walk(L, in, st, vb, W, pc, pn);
    else
      v3_feeder(L, in, st, vb, epochs);
    reason = after();
    if (L.done != 0u) break;
    __syncthreads();
  }
  w_matched = W.matched, w_head = W.head_matched, w_walked = W.walked, w_opens = W.opens, w_fast = W.fast, w_wait = W.wait_ticks;
  finish(L.abort != 0u ? 2u : 0u);
  if (L.abort != 0u) return;
  // ---- statistics ---------------------------------------------------------------------------------------------------------------------
#ifdef COOK_V3_PROF
  if (lane == 0)
    for (int i = 0; i < V3_NPROF; ++i) {
      if (pc[i]) atomicAdd(&vb.ctl->prof_cyc[i], pc[i]);
      if (pn[i]) atomicAdd(&vb.ctl->prof_cnt[i], pn[i]);
    }
#endif
  if (tid == 0) {
    V3Ctl* c = vb.ctl;
    c->head = in.K;
    c->matched = w_matched;
    c->head_matched = w_head;
    c->generations = gens;
    c->epochs = epochs;
    c->stop_full = s_full, c->stop_list = s_list, c->stop_log = s_log, c->stop_other = 0;
    c->walked = w_walked;
    c->opens = w_opens;
    c->fast = w_fast;
    c->t_total = cook_ticks() - tk0;
    c->t_regen = t_regen;
    c->t_flush = t_flush;
    c->t_walk_wait = w_wait;
  }
}

// classfit.hpp — class-ordered best fit (`cook_params.match_algo` 0 / 3): the placement of match_kernels.hpp WITHOUT a K x M evaluation.
//
// Semantics are those of match_serial (Fenzo scheduleOnce + cpuMemBinPacker, scheduler.clj:617-687, 2301-2324, config.clj:108): for each job in
// rank order the feasible offer of greatest fitness, lowest index on ties.  The observation (Next 1): the fitness of offer v for
// job (c, m) is  1 - E_v / (2 Tc Tm) + (c / Tc + m / Tm) / 2  with Tc / Tm = lease + running totals and E_v = free_c * Tm + free_m * Tc —
// offers of equal totals (a CLASS; four in BASELINE's cluster, a handful per node pool anywhere) are ordered the same way for EVERY job.  So:
//   * all resources of the call in FIXED POINT (u32 multiples of 2^-kc cpus / 2^-km MiB; calls whose numbers are not such multiples keep
//     the window rounds of match_v2.hpp): room and the order inside a class are exact integer arithmetic;
//   * per class a sorted array (E ascending = fullest first) in LDS, cut into chunks of 64 positions; per chunk CF_LV level summaries
//     (greatest free mem among members with free cpus >= level, the levels = the jobs' cpus values), kept EXACT;
//   * one workgroup per pool: wave 0 holds the OVERLAY — the offers this call has placed on, state in registers, one per lane — and DECIDES,
//     six class waves hold the chunks of their classes (lane = chunk: summaries in registers), one wave keeps the books of the jobs nobody
//     has to visit (classfit_walk.hpp);
//   * a job: the class waves ballot their chunk summaries, scan the first candidate chunk (LDS, lane = position) and publish its first
//     feasible member ahead of the decider; the decider evaluates its overlay lanes and the published candidates in one go and commits
//     (overlay lane update / the member is zeroed in its chunk and moves into an overlay lane);
//   * candidates within 2^-37 of the best (equal E, rounding) are decided by the oracle's literal expression (the "exact" turn, rare);
//   * jobs that no offer of their kind has room for are settled 64 at a time from per-wave level maxima (placements only take room away);
//   * 64 live overlay lanes end an EPOCH: the lanes are merged back into their classes' arrays.
// No evaluation launches, no candidate lists, no rounds.  tests/classfit_model/ is the CPU model this file follows (the gate: 4.3 wave-steps
// per matched job, 1.0 per unmatched one on a BASELINE C4 pool, 32 epochs).
#pragma once
#include "common.hpp"
#include "match_kernels.hpp"

constexpr int CF_LV = 8;                  // level summaries per chunk
#ifndef CF_NT
#define CF_NT 1024
#endif
constexpr int CF_THREADS = COOK_SHAPE(CF_NT, 512);  // wave 0 the decider, wave 4 the bookkeeper, the others class waves (classfit_walk.hpp)
constexpr int CF_CW = 6;                  // sets of classes (a set = the classes one class wave's lanes hold; several waves may share a set's jobs)
constexpr int CF_WAVES = CF_THREADS / COOK_WAVE;
constexpr int CF_MAXCLS = 48;             // classes per call (totals x gpu kind)
constexpr int CF_MAXKIND = 32;            // gpu kinds incl. kind 0 = hosts without gpus
constexpr unsigned CF_OV_CAP = COOK_SHAPE(64, 8);  // live overlay lanes that end an epoch
constexpr unsigned CF_SORT_N = 8192;      // offers per call at most (the prepare kernel sorts them in LDS; ids are 13 bits)
constexpr unsigned CF_LDS_BYTES = 160u * 1024u - 2048u;
constexpr unsigned CF_MAXG = 4096;        // groups per call
constexpr unsigned CF_GMEM = 16;          // pending members per unique host-placement group
constexpr unsigned CF_OCC = 0x8000u, CF_TIE = 0x4000u, CF_IDMASK = 0x3FFFu;  // pos_cid = class << 16 | occupied gpu host << 15 | the NEXT member of the
                                                                             // class may round to the same fitness << 14 | offer
constexpr unsigned CF_KIND_NONE = 0xFFu;
constexpr double CF_BAND = 1.0 / 137438953472.0;  // 2^-37

// why a call keeps the window rounds (CfCtl::inelig)
enum : unsigned {
  CF_X_NUMBERS = 1u,      // a resource is negative / not finite / not a multiple of 2^-20 / too large for 31 bits
  CF_X_JOB_SLOW = 2u,     // a job with a constraint outside {EQUALS on keys < 8 with byte values, <= 4 novel hosts, unique group, gpu}
  CF_X_XRES = 4u,         // ports / named scalars
  CF_X_GROUP = 8u,        // balanced / attribute-equals groups, too many groups or members
  CF_X_OFFER = 16u,       // gpu maps with several entries, max-tasks-per-host, reserved hosts, two offers on one host, attribute values >= 256
  CF_X_SHAPE = 32u,       // too many classes / kinds / offers, a class beyond 64 chunks, LDS
  CF_X_LEVELS = 64u,      // job cpus values not on the 8 levels
  CF_X_ZERO = 128u,       // a job asking for nothing (fitness 0 is a failure in Fenzo), totals of 0
};

struct CfClass {
  uint32_t Tc, Tm, kind, n, off, wave, pad0, pad1;
  double hTc, hTm;  // 0.5 / Tc, 0.5 / Tm (fixed-point units): the approximate fitness
  uint64_t dE;      // offers of the class whose E differ by at most dE may round to the same fitness
};
struct CfCtl {
  // cf_scan (atomics; zeroed before)
  uint32_t inelig, fb_c, fb_m, pad0;
  uint64_t max_c_bits, max_m_bits;  // greatest cpus / mem value of the call (jobs, leases, totals), as double bits
  uint64_t jmax_c_bits;             // greatest job cpus
  uint32_t eq_keys, attr_max[8];    // attribute keys (< 8) some job's EQUALS names; greatest value id of each key over the offers
  uint32_t pad1[3];
  // cf_prepare
  uint32_t kc, km, n_cls, n_kind, M, minfc_all, minfm_all, cmin, mmin, any_eq, any_group, n_grouped;  // n_grouped: jobs in unique groups
  uint32_t t[CF_LV];
  uint64_t kind_sig[CF_MAXKIND];  // gpu model << 32 | count
  CfClass cls[CF_MAXCLS];
  // cf_walk
  uint32_t stats[48];
};
enum { CFS_SPINS = 17,
       CFS_WALKED = 0, CFS_MATCHED, CFS_OV_WIN, CFS_OPEN, CFS_OPEN_DEAD, CFS_GPU_PLACE, CFS_EPOCHS, CFS_SCANS, CFS_EXACT, CFS_TIGHTEN, CFS_PRESETTLED, CFS_BATCHES, CFS_DEAD_DROP,
       CFS_TICKS_TOTAL, CFS_TICKS_PROLOGUE, CFS_TICKS_EPOCH, CFS_TICKS_PRECHECK, CFS_TICKS_WALK = 18, CFS_TICKS_PHASE1, CFS_REWINDS, CFS_FLIPS, CFS_HWID_DECIDER, CFS_HWID_BOOKS };
struct CfJob {  // one job as the walk reads it (32 B)
  uint32_t c, m;       // fixed point
  uint32_t meta;       // kind | level << 8 | n_eq << 12 | n_nov << 16 | grouped << 20
  uint32_t grp;
  uint32_t eq[2];      // 4 x u16: key << 8 | value
  uint32_t nov[2];     // 4 x u16: offers on the hosts the job already ran on (0xFFFF none)
};
struct CfBuf {
  CfCtl* ctl;
  const JobRec* jr;
  const JobCons* jcons;
  const OfferA* oa;
  const OfferB* ob;
  uint64_t* attr8;       // [M] the first 8 attribute values of an offer, one byte each
  uint32_t* h2o;         // [max_host + 1] host -> offer
  uint32_t max_host;
  uint32_t *pos_fc, *pos_fm, *pos_cid;  // [M] class arrays as cf_prepare sorted them
  uint32_t *scr_fc, *scr_fm, *scr_cid;  // [M] scratch of an epoch's merge
  CfJob* jobs;           // [K]
  uint32_t* gcount;      // [G] pending members of a unique group in this call
  uint32_t* gmem;        // [G][CF_GMEM] their match indices
};

static __device__ __forceinline__ unsigned cf_fbits(double x, bool& bad) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  if ((b << 1) == 0ull) return 0u;  // +-0
  const int e = (int)((b >> 52) & 0x7FFull);
  if ((b >> 63) || e == 0 || e == 0x7FF) {
    bad = true;
    return 0u;
  }
  const unsigned long long mant = (b & ((1ull << 52) - 1ull)) | (1ull << 52);
  const int low = e - 1075 + (__ffsll(mant) - 1);  // weight of the lowest set bit
  return low >= 0 ? 0u : (unsigned)(-low);
}
static __host__ __device__ __forceinline__ unsigned cf_min(unsigned a, unsigned b) { return a < b ? a : b; }
static __host__ __device__ __forceinline__ unsigned cf_max(unsigned a, unsigned b) { return a > b ? a : b; }
static __device__ __forceinline__ double cf_pow2(int k) { return __longlong_as_double((long long)(1023 + k) << 52); }
static __device__ __forceinline__ uint32_t cf_fx(double v, unsigned k) { return (uint32_t)(v * cf_pow2((int)k)); }

// ---- 0. the call's tables cleared (one launch for all of them — and for every pool of a batch — where three fills were issued alone)
COOK_KERNEL void cf_init(CfBuf b, unsigned n_h2o, unsigned G) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  uint32_t* cw = reinterpret_cast<uint32_t*>(b.ctl);
  for (unsigned x = i; x < (unsigned)(sizeof(CfCtl) / 4u); x += stride) cw[x] = 0u;
  for (unsigned x = i; x < n_h2o; x += stride) b.h2o[x] = 0xFFFFFFFFu;
  for (unsigned x = i; x < G; x += stride) b.gcount[x] = 0u;
}

// ---- 1. what the numbers of the call look like (grid over max(K, M)) ------------------------------------------------------------------------
COOK_KERNEL void cf_scan(const MatchIn* __restrict__ inp, CfBuf b, unsigned K, unsigned M) {
  const MatchIn& in = *inp;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned inelig = 0, fbc = 0, fbm = 0;
  double mxc = 0.0, mxm = 0.0, jmx = 0.0;
  bool bad = false;
  if (i < K) {
    const JobRec j = b.jr[i];
    fbc = cf_fbits(j.c, bad), fbm = cf_fbits(j.m, bad);
    mxc = j.c, mxm = j.m, jmx = j.c;
    if (!(j.c > 0.0 || j.m > 0.0)) inelig |= CF_X_ZERO;
    if (!(j.g >= 0.0) || !(j.g < 4294967296.0) || j.g != (double)(unsigned)j.g) inelig |= CF_X_NUMBERS;
    if (j.flags & JF_SLOW) inelig |= CF_X_JOB_SLOW;
    if (j.flags & JF_XRES) inelig |= CF_X_XRES;
    if ((j.flags & JF_GROUPED) && ((j.flags >> 8) & 3u) != 1u) inelig |= CF_X_GROUP;
    if (j.flags & JF_FASTC) {
      const JobCons c = b.jcons[i];
      unsigned keys = 0;
      for (unsigned q = 0; q < (unsigned)MV_NC; ++q)
        if (q < c.n_eq) {
          if (c.eq_key[q] >= 8u || c.eq_val[q] >= 256u) inelig |= CF_X_JOB_SLOW;
          else keys |= 1u << c.eq_key[q];
        }
      if (keys && (keys & ~ld_agent(&b.ctl->eq_keys))) atomicOr(&b.ctl->eq_keys, keys);  // (read first: every wave would queue at ONE address)
    }
  }
  if (i < M) {
    const OfferA a = b.oa[i];
    const OfferB o = b.ob[i];
    const double Tc = a.oc + a.rc, Tm = a.om + a.rm;
    fbc = cf_max(fbc, cf_max(cf_fbits(a.oc, bad), cf_max(cf_fbits(a.rc, bad), cf_fbits(Tc, bad))));
    fbm = cf_max(fbm, cf_max(cf_fbits(a.om, bad), cf_max(cf_fbits(a.rm, bad), cf_fbits(Tm, bad))));
    mxc = mxc > Tc ? mxc : Tc, mxm = mxm > Tm ? mxm : Tm;
    if (!(Tc > 0.0) || !(Tm > 0.0)) inelig |= CF_X_ZERO;
    if ((o.flags & 6u) || o.task_slack != 0x7FFFFFFF) inelig |= CF_X_OFFER;
    if (!(o.gpu_count >= 0.0) || !(o.gpu_count < 4294967296.0) || o.gpu_count != (double)(unsigned)o.gpu_count) inelig |= CF_X_NUMBERS;
    uint64_t a8 = 0;
    for (unsigned key = 0; key < 8u && key < in.n_attr; ++key) {
      const uint32_t val = in.o_attr[(size_t)i * in.n_attr + key];
      if (val >= 256u) atomicMax(&b.ctl->attr_max[key], val);  // (only the keys some job names must fit a byte: cf_prepare)
      a8 |= (uint64_t)(val & 255u) << (8u * key);
    }
    b.attr8[i] = a8;
    if (o.host <= b.max_host) b.h2o[o.host] = i;
    else inelig |= CF_X_OFFER;
  }
  if (bad) inelig |= CF_X_NUMBERS;
  if (inelig && (inelig & ~ld_agent(&b.ctl->inelig))) atomicOr(&b.ctl->inelig, inelig);
  // (positive doubles order like their bit patterns)
  fbc = wave_max_u32(fbc), fbm = wave_max_u32(fbm);
  const unsigned long long c64 = wave_max_u64((unsigned long long)__double_as_longlong(mxc)), m64 = wave_max_u64((unsigned long long)__double_as_longlong(mxm)),
                           j64 = wave_max_u64((unsigned long long)__double_as_longlong(jmx));
  if (lane_id() == 0) {  // (a maximum already there needs no atomic: after the first waves nearly none is issued; a stale read only costs one)
    if (fbc > ld_agent(&b.ctl->fb_c)) atomicMax(&b.ctl->fb_c, fbc);
    if (fbm > ld_agent(&b.ctl->fb_m)) atomicMax(&b.ctl->fb_m, fbm);
    if (c64 > ld_agent((const unsigned long long*)&b.ctl->max_c_bits)) atomicMax((unsigned long long*)&b.ctl->max_c_bits, c64);
    if (m64 > ld_agent((const unsigned long long*)&b.ctl->max_m_bits)) atomicMax((unsigned long long*)&b.ctl->max_m_bits, m64);
    if (j64 > ld_agent((const unsigned long long*)&b.ctl->jmax_c_bits)) atomicMax((unsigned long long*)&b.ctl->jmax_c_bits, j64);
  }
}

// ---- 2. classes, the sorted class arrays, the call's constants (ONE workgroup of 1024) --------------------------------------------------------
static __device__ __forceinline__ int cf_tab_find64(unsigned long long* tab, unsigned long long key) {  // wait-free insert-or-find, 64 slots
  unsigned s = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 58);
  for (unsigned probe = 0; probe < 64u; ++probe) {
    // (a look first: a wave's lanes mostly ask for the same few keys, and 64 compare-and-swaps on ONE LDS word take their turns; a slot that holds a key keeps it)
    unsigned long long cur = ld_wg(&tab[s]);
    if (cur == ~0ull) {
      cur = atomicCAS(&tab[s], ~0ull, key);
      if (cur == ~0ull) return (int)s;
    }
    if (cur == key) return (int)s;
    s = (s + 1u) & 63u;
  }
  return -1;
}
static __device__ __forceinline__ void cf_cmpx(unsigned long long* key, unsigned i, unsigned p, unsigned N) {
  if (p < N) {  // (positions beyond N hold +inf: they never move)
    const unsigned long long a = key[i], c = key[p];
    if (c < a) key[i] = c, key[p] = a;
  }
}
COOK_KERNEL void cf_prepare(const MatchIn* __restrict__ inp, CfBuf b, const double* __restrict__ jmin, unsigned K, unsigned M, unsigned G, unsigned host_dup,
                            unsigned has_reserved) {
  constexpr unsigned B = 1024;
  __shared__ unsigned long long s_key[CF_SORT_N];
  __shared__ unsigned long long s_shape[64], s_sig[64];
  __shared__ unsigned s_pair[256];  // (shape slot << 8 | sig slot) -> class, 0xFFFFFFFF free
  __shared__ unsigned s_pair_cls[256];
  __shared__ unsigned s_cnt[CF_MAXCLS], s_off[CF_MAXCLS];
  __shared__ unsigned s_bad, s_minfc, s_minfm, s_ncls;
  __shared__ unsigned long long s_dE[CF_MAXCLS];
  __shared__ unsigned s_kind[CF_MAXCLS], s_wave[CF_MAXCLS];  // (thread 0's wave assignment reads these, not ctl->cls in global memory)
  __shared__ unsigned long long sigs[CF_MAXKIND];            // thread 0's small tables: indexed at run time, so as locals they would live in scratch = global memory
  __shared__ unsigned ck[CF_MAXCLS], cslot[CF_MAXCLS], load[CF_CW + 1], cnt[CF_CW + 1];
  const unsigned tid = threadIdx.x;
  CfCtl* ctl = b.ctl;
  if (tid == 0) {
    unsigned bad = ctl->inelig;
    const unsigned kc = ctl->fb_c, km = ctl->fb_m;
    if (kc > 20u || km > 20u) bad |= CF_X_NUMBERS;
    const double mc = __longlong_as_double((long long)ctl->max_c_bits), mm = __longlong_as_double((long long)ctl->max_m_bits);
    if (!(bad & CF_X_NUMBERS) && (!(mc * cf_pow2((int)kc) < 1073741824.0) || !(mm * cf_pow2((int)km) < 1073741824.0))) bad |= CF_X_NUMBERS;
    if (M == 0u || M > CF_SORT_N || K == 0u || K > (1u << 20) || G > CF_MAXG) bad |= CF_X_SHAPE;  // (K: a board entry's tag holds the job in 20 bits)
    if (host_dup || has_reserved) bad |= CF_X_OFFER;
    for (unsigned key = 0; key < 8u; ++key)
      if (((ctl->eq_keys >> key) & 1u) && ctl->attr_max[key] >= 256u) bad |= CF_X_OFFER;
    if (__double_as_longlong(jmin[2]) != 0ll) bad |= CF_X_NUMBERS;  // match_job_minima saw a negative / non-finite request
    s_bad = bad;
    s_minfc = 0xFFFFFFFFu, s_minfm = 0xFFFFFFFFu;
    ctl->kc = kc, ctl->km = km, ctl->M = M;
  }
  for (unsigned x = tid; x < 64u; x += B) s_shape[x] = ~0ull, s_sig[x] = ~0ull;
  for (unsigned x = tid; x < 256u; x += B) s_pair[x] = 0xFFFFFFFFu;
  for (unsigned x = tid; x < (unsigned)CF_MAXCLS; x += B) s_cnt[x] = 0u;
  __syncthreads();
  if (s_bad) {
    if (tid == 0) ctl->inelig = s_bad;
    return;
  }
  const unsigned kc = ctl->fb_c, km = ctl->fb_m;
  // classes: (totals, gpu kind) through three small insert-or-find tables
  unsigned bad = 0;
  for (unsigned v = tid; v < M; v += B) {
    const OfferA a = b.oa[v];
    const OfferB o = b.ob[v];
    const uint32_t Lc = cf_fx(a.oc, kc), Lm = cf_fx(a.om, km), Tc = cf_fx(a.oc + a.rc, kc), Tm = cf_fx(a.om + a.rm, km);
    if (Tc != Lc + cf_fx(a.rc, kc) || Tm != Lm + cf_fx(a.rm, km)) bad |= CF_X_NUMBERS;  // (the totals must be exact sums)
    const unsigned long long sig = ((o.flags & 1u) && o.gpu_model != 0u) ? ((unsigned long long)o.gpu_model << 32 | (unsigned long long)(unsigned)o.gpu_count) : 0ull;
    const int ss = cf_tab_find64(s_shape, (unsigned long long)Tc << 32 | Tm), sg = cf_tab_find64(s_sig, sig);
    if (ss < 0 || sg < 0) {
      bad |= CF_X_SHAPE;
      continue;
    }
    const unsigned pk = (unsigned)ss << 8 | (unsigned)sg;
    unsigned s = (pk * 2654435761u) >> 24;
    bool placed = false;
    for (unsigned probe = 0; probe < 256u && !placed; ++probe) {
      unsigned old = ld_wg(&s_pair[s]);
      if (old == 0xFFFFFFFFu) old = atomicCAS(&s_pair[s], 0xFFFFFFFFu, pk);
      if (old == 0xFFFFFFFFu || old == pk) placed = true;
      else s = (s + 1u) & 255u;
    }
    if (!placed) bad |= CF_X_SHAPE;
    if (Lc < ld_wg(&s_minfc)) atomicMin(&s_minfc, Lc);
    if (Lm < ld_wg(&s_minfm)) atomicMin(&s_minfm, Lm);
  }
  if (bad) atomicOr(&s_bad, bad);
  __syncthreads();
  if (tid == 0 && !s_bad) {  // a canonical order: kinds by signature (0 = no gpus first), classes by (kind, totals)
    unsigned nk = 1;
    sigs[0] = 0ull;
    for (unsigned x = 0; x < 64u; ++x)
      if (s_sig[x] != ~0ull && s_sig[x] != 0ull) {
        if (nk >= (unsigned)CF_MAXKIND) {
          s_bad |= CF_X_SHAPE;
          break;
        }
        unsigned p = nk++;
        while (p > 1u && sigs[p - 1u] > s_sig[x]) sigs[p] = sigs[p - 1u], --p;
        sigs[p] = s_sig[x];
      }
    for (unsigned x = 0; x < nk; ++x) ctl->kind_sig[x] = sigs[x];
    ctl->n_kind = nk;
    unsigned nc = 0;
    // (ck = a class's kind; kind << 56 is too little for sorting with the totals: sort the slots by (kind, Tc, Tm) directly)
    for (unsigned x = 0; x < 256u && !s_bad; ++x)
      if (s_pair[x] != 0xFFFFFFFFu) {
        if (nc >= (unsigned)CF_MAXCLS) {
          s_bad |= CF_X_SHAPE;
          break;
        }
        const unsigned long long shape = s_shape[s_pair[x] >> 8], sig = s_sig[s_pair[x] & 255u];
        unsigned kind = 0;
        for (unsigned y = 0; y < nk; ++y)
          if (sigs[y] == sig) kind = y;
        unsigned p = nc++;
        // insertion by (kind, shape)
        while (p > 0u) {
          const unsigned long long pshape = s_shape[s_pair[cslot[p - 1u]] >> 8];
          const unsigned pkind = ck[p - 1u];
          if (pkind < kind || (pkind == kind && pshape < shape)) break;
          ck[p] = ck[p - 1u], cslot[p] = cslot[p - 1u], --p;
        }
        ck[p] = kind, cslot[p] = x;
      }
    for (unsigned c = 0; c < nc && !s_bad; ++c) {
      s_pair_cls[cslot[c]] = c;
      const unsigned long long shape = s_shape[s_pair[cslot[c]] >> 8];
      CfClass& cl = ctl->cls[c];
      cl.Tc = (uint32_t)(shape >> 32), cl.Tm = (uint32_t)shape, cl.kind = (unsigned)ck[c], cl.n = 0u, cl.off = 0u, cl.wave = 0u, cl.pad0 = cl.pad1 = 0u;
      cl.hTc = 0.5 / (double)cl.Tc, cl.hTm = 0.5 / (double)cl.Tm;
      cl.dE = (uint64_t)(CF_BAND * 2.0 * (double)cl.Tc * (double)cl.Tm);
      s_dE[c] = cl.dE;
      s_kind[c] = (unsigned)ck[c];
      if (!(2.0 * (double)cl.Tc * (double)cl.Tm < 35184372088832.0)) s_bad |= CF_X_SHAPE;  // E below 2^45: the sort key is class | E | offer
    }
    s_ncls = nc;
    ctl->n_cls = nc;
  }
  __syncthreads();
  if (s_bad) {
    if (tid == 0) ctl->inelig = s_bad;
    return;
  }
  // keys: class << 58 | E << 13 | offer
  for (unsigned v = tid; v < M; v += B) {
    const OfferA a = b.oa[v];
    const OfferB o = b.ob[v];
    const uint32_t Lc = cf_fx(a.oc, kc), Lm = cf_fx(a.om, km), Tc = cf_fx(a.oc + a.rc, kc), Tm = cf_fx(a.om + a.rm, km);
    const unsigned long long sig = ((o.flags & 1u) && o.gpu_model != 0u) ? ((unsigned long long)o.gpu_model << 32 | (unsigned long long)(unsigned)o.gpu_count) : 0ull;
    const unsigned pk = (unsigned)cf_tab_find64(s_shape, (unsigned long long)Tc << 32 | Tm) << 8 | (unsigned)cf_tab_find64(s_sig, sig);
    unsigned s = (pk * 2654435761u) >> 24;
    while (s_pair[s] != pk) s = (s + 1u) & 255u;
    const unsigned c = s_pair_cls[s];
    atomicAdd(&s_cnt[c], 1u);
    const unsigned long long E = (unsigned long long)Lc * Tm + (unsigned long long)Lm * Tc;
    s_key[v] = (unsigned long long)c << 58 | E << 13 | (unsigned long long)v;
  }
  __syncthreads();
  // normalized bitonic sort (every comparison ascending, so the virtual +inf beyond M stay where they are)
  unsigned n2 = 1;
  while (n2 < M) n2 <<= 1;
  for (unsigned k = 2; k <= n2; k <<= 1) {
    for (unsigned t = tid; t < n2 / 2u; t += B) {
      const unsigned blk = t / (k / 2u), r = t % (k / 2u);
      const unsigned i = blk * k + r, p = blk * k + (k - 1u - r);
      if (i < M) cf_cmpx(s_key, i, p, M);
    }
    __syncthreads();
    for (unsigned j = k / 4u; j >= 1u; j >>= 1) {
      for (unsigned t = tid; t < n2 / 2u; t += B) {
        const unsigned i = (t / j) * 2u * j + (t % j);
        if (i < M) cf_cmpx(s_key, i, i + j, M);
      }
      __syncthreads();
    }
  }
  for (unsigned q = tid; q < M; q += B) {
    const unsigned long long key = s_key[q];
    const unsigned v = (unsigned)(key & 8191ull), c = (unsigned)(key >> 58);
    const OfferA a = b.oa[v];
    const OfferB o = b.ob[v];
    const bool gpu_host = (o.flags & 1u) && o.gpu_model != 0u;
    b.pos_fc[q] = cf_fx(a.oc, kc), b.pos_fm[q] = cf_fx(a.om, km);
    // the next member of the class inside the guard band of this one: whoever takes this one as the best must look at the literal fitness
    const unsigned long long nk = q + 1u < M ? s_key[q + 1u] : ~0ull;
    const bool tie = (unsigned)(nk >> 58) == c && ((nk >> 13) & ((1ull << 45) - 1ull)) <= ((key >> 13) & ((1ull << 45) - 1ull)) + s_dE[c];
    b.pos_cid[q] = c << 16 | ((gpu_host && o.run_count != 0) ? CF_OCC : 0u) | (tie ? CF_TIE : 0u) | v;
  }
  if (tid == 0) {
    // offsets; class waves (logical 1..CF_CW; a wave's lanes hold 64 chunks).  A job asks EVERY class of its kind, and a wave answers its classes one after
    // the other: the classes of hosts without gpus are spread by COUNT over the waves the gpu classes leave (largest first, the wave with the fewest
    // classes, then the fewest chunks), the gpu classes go by kind onto one wave, two when the others leave them
    unsigned off = 0, bad2 = 0;
    const unsigned nc = s_ncls;
    unsigned n0 = 0, any_gpu = 0;
    for (unsigned c = 0; c < nc; ++c) {
      s_off[c] = off, s_wave[c] = 0u;
      off += s_cnt[c];
      if ((s_cnt[c] + 63u) / 64u > 64u) bad2 |= CF_X_SHAPE;
      if (s_kind[c] == 0u) ++n0;
      else any_gpu = 1u;
    }
    const unsigned gw = any_gpu ? ((n0 + 2u <= (unsigned)CF_CW && ctl->n_kind > 2u) ? 2u : 1u) : 0u;
    const unsigned zw = (unsigned)CF_CW - gw;
    for (unsigned x = 0; x <= (unsigned)CF_CW; ++x) load[x] = 0u, cnt[x] = 0u;
    for (unsigned round = 0; round < n0 && !bad2; ++round) {
      unsigned best = 0xFFFFFFFFu, bn = 0;
      for (unsigned c = 0; c < nc; ++c)  // the largest class without a wave
        if (s_kind[c] == 0u && s_wave[c] == 0u && (best == 0xFFFFFFFFu || s_cnt[c] > bn)) best = c, bn = s_cnt[c];
      const unsigned nch = (bn + 63u) / 64u;
      unsigned wsel = 0;
      for (unsigned x = 1; x <= zw; ++x)
        if (load[x] + nch <= 64u && (wsel == 0u || cnt[x] < cnt[wsel] || (cnt[x] == cnt[wsel] && load[x] < load[wsel]))) wsel = x;
      if (wsel == 0u) {
        bad2 |= CF_X_SHAPE;
        break;
      }
      s_wave[best] = wsel, load[wsel] += nch, ++cnt[wsel];
    }
    for (unsigned c = 0; c < nc && !bad2; ++c) {
      if (s_kind[c] == 0u) continue;
      const unsigned wsel = zw + 1u + (s_kind[c] - 1u) % gw;
      s_wave[c] = wsel, load[wsel] += (s_cnt[c] + 63u) / 64u;
      if (load[wsel] > 64u) bad2 |= CF_X_SHAPE;
    }
    for (unsigned c = 0; c < nc; ++c) {
      CfClass& cl = ctl->cls[c];
      cl.n = s_cnt[c], cl.off = s_off[c], cl.wave = s_wave[c];
    }
    // levels: 8 values between the smallest and the greatest cpus request (exact for up to 8 evenly spaced values: 1..8 cores)
    const uint32_t cmin = cf_fx(jmin[0], kc), cmax = cf_fx(__longlong_as_double((long long)ctl->jmax_c_bits), kc);
    for (int i = 0; i < CF_LV; ++i) ctl->t[i] = cmin + (uint32_t)(((uint64_t)i * (cmax - cmin)) / (CF_LV - 1));
    ctl->cmin = cmin, ctl->mmin = cf_fx(jmin[1], km);
    ctl->minfc_all = s_minfc, ctl->minfm_all = s_minfm;
    ctl->any_eq = 0u, ctl->any_group = 0u, ctl->n_grouped = 0u;
    ctl->inelig = bad2;
  }
}

// ---- 3. the jobs as the walk reads them (grid over K) -------------------------------------------------------------------------------------------
COOK_KERNEL void cf_pack_jobs(const MatchIn* __restrict__ inp, CfBuf b, unsigned K) {
  const MatchIn& in = *inp;
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  CfCtl* ctl = b.ctl;
  unsigned grouped = 0;
  if (k < K && ctl->inelig == 0u) {  // (no early exit: the wave counts its grouped jobs together below; a call found ineligible is not packed)
  const JobRec j = b.jr[k];
  CfJob o;
  o.c = cf_fx(j.c, ctl->kc), o.m = cf_fx(j.m, ctl->km);
  unsigned L = 0, bad = CF_X_LEVELS;
  for (int i = 0; i < CF_LV; ++i)
    if (ctl->t[i] == o.c) L = (unsigned)i, bad = 0u;
  unsigned kind = 0;
  if (j.g > 0.0) {  // a gpu job runs on the hosts whose map holds exactly (its model -> its count) (constraints.clj:122-157)
    kind = CF_KIND_NONE;
    const unsigned long long sig = (unsigned long long)j.gpu_model << 32 | (unsigned long long)(unsigned)j.g;
    if (j.gpu_model != 0u)
      for (unsigned y = 1; y < ctl->n_kind; ++y)
        if (ctl->kind_sig[y] == sig) kind = y;
  }
  o.eq[0] = o.eq[1] = 0u, o.nov[0] = o.nov[1] = 0xFFFFFFFFu;
  unsigned n_eq = 0, n_nov = 0;
  if (j.flags & JF_FASTC) {
    const JobCons c = b.jcons[k];
    n_eq = c.n_eq;
    for (unsigned q = 0; q < (unsigned)MV_NC; ++q) {
      if (q < c.n_eq) o.eq[q >> 1] |= ((c.eq_key[q] << 8 | c.eq_val[q]) & 0xFFFFu) << (16u * (q & 1u));
      if (q < c.n_novel) {  // hosts without an offer in this call forbid nothing
        const uint32_t h = c.novel[q];
        const uint32_t v = h <= b.max_host ? b.h2o[h] : 0xFFFFFFFFu;
        if (v != 0xFFFFFFFFu) {
          o.nov[n_nov >> 1] = (o.nov[n_nov >> 1] & ~(0xFFFFu << (16u * (n_nov & 1u)))) | (v & 0xFFFFu) << (16u * (n_nov & 1u));
          ++n_nov;
        }
      }
    }
    if (n_eq) ctl->any_eq = 1u;
  }
  o.grp = 0xFFFFFFFFu;
  if ((j.flags & JF_GROUPED) && ((j.flags >> 8) & 3u) == 1u) {
    grouped = 1u;
    o.grp = j.group;
    const unsigned slot = atomicAdd(&b.gcount[j.group], 1u);
    if (slot < CF_GMEM) b.gmem[(size_t)j.group * CF_GMEM + slot] = k;
    else bad |= CF_X_GROUP;
    ctl->any_group = 1u;
  }
  o.meta = kind | L << 8 | n_eq << 12 | n_nov << 16 | grouped << 20;
  b.jobs[k] = o;
  if (bad) atomicOr(&ctl->inelig, bad);
  }
  // one addition per wave (a few thousand grouped jobs would otherwise queue at ONE address)
  const unsigned long long gm = cook_ballot(grouped != 0u);
  if (gm != 0ull && lane_id() == (unsigned)__ffsll((long long)gm) - 1u) atomicAdd(&ctl->n_grouped, (unsigned)__popcll(gm));
  (void)in;
}

// ---- 5. the placement chains of the unique groups, as cook_match_explain reads them (grid over G) ---------------------------------------------------
__global__ void cf_group_chains(CfBuf b, MatchState st, unsigned G) {
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const unsigned n = cf_min(b.gcount[g], CF_GMEM);
  unsigned ks[CF_GMEM];
  unsigned m = 0;
  for (unsigned x = 0; x < n; ++x) {
    const unsigned k = b.gmem[(size_t)g * CF_GMEM + x];
    if (st.job_to_offer[k] < 0) continue;
    unsigned p = m++;
    while (p > 0u && ks[p - 1u] > k) ks[p] = ks[p - 1u], --p;
    ks[p] = k;
  }
  int prev = -1;
  for (unsigned x = 0; x < m; ++x) st.job_prev[ks[x]] = prev, prev = (int)ks[x];
  st.group_last[g] = prev;
}

// ---- 4. the walk ------------------------------------------------------------------------------------------------------------------------------------------
#include "classfit_walk.hpp"

struct CfPoolCtx {  // one pool of a launch
  const MatchIn* in;
  MatchState st;
  CfBuf b;
};
constexpr int CF_PACK = 8;
struct CfPack {
  CfPoolCtx c[CF_PACK];
};
__global__ void __launch_bounds__(CF_THREADS) cf_walk(const CfPack p) {
  __shared__ __attribute__((aligned(16))) char lds[CF_LDS_BYTES];
  const CfPoolCtx& c = p.c[blockIdx.x];
  cf_walk_pool(lds, c.in, c.st, c.b);
}

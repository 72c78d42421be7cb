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
//   * one workgroup per pool: wave 0 holds the OVERLAY — the offers this call has placed on, state in registers, one per lane —, waves
//     1..6 hold the chunks of their classes (lane = chunk: summaries in registers), wave 7 keeps the books of the jobs nobody has to visit;
//   * a job: the class waves ballot their chunk summaries, scan the first candidate chunk (LDS, lane = position) and post its first
//     feasible member; the overlay wave evaluates its 64 lanes and posts the best; one barrier; every wave reads the posts and knows the
//     winner; the source commits (overlay lane update / member moves from its chunk into an overlay lane);
//   * candidates within 2^-37 of the best (equal E, rounding) are decided by the oracle's literal expression (the "exact" turn, rare);
//   * jobs that no offer of their kind has room for are settled 64 at a time from per-wave level maxima (placements only take room away);
//   * 64 live overlay lanes end an EPOCH: the lanes are merged back into their classes' arrays.
// No evaluation launches, no candidate lists, no rounds.  tests/classfit_model/ is the CPU model this file follows (the gate: 4.3 wave-steps
// per matched job, 1.0 per unmatched one on a BASELINE C4 pool, 32 epochs).
#pragma once
#include "common.hpp"
#include "match_kernels.hpp"

constexpr int CF_LV = 8;                  // level summaries per chunk
constexpr int CF_THREADS = 512;           // wave 0 overlay, waves 1..CF_CW class waves, wave 7 bookkeeper
constexpr int CF_CW = 6;
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
       CFS_TICKS_TOTAL, CFS_TICKS_PROLOGUE, CFS_TICKS_EPOCH, CFS_TICKS_PRECHECK };
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
      if (keys) atomicOr(&b.ctl->eq_keys, keys);
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
  if (inelig) atomicOr(&b.ctl->inelig, inelig);
  // (positive doubles order like their bit patterns)
  fbc = wave_max_u32(fbc), fbm = wave_max_u32(fbm);
  const unsigned long long c64 = wave_max_u64((unsigned long long)__double_as_longlong(mxc)), m64 = wave_max_u64((unsigned long long)__double_as_longlong(mxm)),
                           j64 = wave_max_u64((unsigned long long)__double_as_longlong(jmx));
  if (lane_id() == 0) {
    atomicMax(&b.ctl->fb_c, fbc), atomicMax(&b.ctl->fb_m, fbm);
    atomicMax((unsigned long long*)&b.ctl->max_c_bits, c64), atomicMax((unsigned long long*)&b.ctl->max_m_bits, m64);
    atomicMax((unsigned long long*)&b.ctl->jmax_c_bits, j64);
  }
}

// ---- 2. classes, the sorted class arrays, the call's constants (ONE workgroup of 1024) --------------------------------------------------------
static __device__ __forceinline__ int cf_tab_find64(unsigned long long* tab, unsigned long long key) {  // wait-free insert-or-find, 64 slots
  unsigned s = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 58);
  for (unsigned probe = 0; probe < 64u; ++probe) {
    const unsigned long long old = atomicCAS(&tab[s], ~0ull, key);
    if (old == ~0ull || old == key) return (int)s;
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
  const unsigned tid = threadIdx.x;
  CfCtl* ctl = b.ctl;
  if (tid == 0) {
    unsigned bad = ctl->inelig;
    const unsigned kc = ctl->fb_c, km = ctl->fb_m;
    if (kc > 20u || km > 20u) bad |= CF_X_NUMBERS;
    const double mc = __longlong_as_double((long long)ctl->max_c_bits), mm = __longlong_as_double((long long)ctl->max_m_bits);
    if (!(bad & CF_X_NUMBERS) && (!(mc * cf_pow2((int)kc) < 1073741824.0) || !(mm * cf_pow2((int)km) < 1073741824.0))) bad |= CF_X_NUMBERS;
    if (M == 0u || M > CF_SORT_N || K == 0u || G > CF_MAXG) bad |= CF_X_SHAPE;
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
      const unsigned old = atomicCAS(&s_pair[s], 0xFFFFFFFFu, pk);
      if (old == 0xFFFFFFFFu || old == pk) placed = true;
      else s = (s + 1u) & 255u;
    }
    if (!placed) bad |= CF_X_SHAPE;
    atomicMin(&s_minfc, Lc), atomicMin(&s_minfm, Lm);
  }
  if (bad) atomicOr(&s_bad, bad);
  __syncthreads();
  if (tid == 0 && !s_bad) {  // a canonical order: kinds by signature (0 = no gpus first), classes by (kind, totals)
    unsigned long long sigs[CF_MAXKIND];
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
    unsigned long long ck[CF_MAXCLS];  // kind << 56 is too little for sorting with the totals: sort the slots by (kind, Tc, Tm) directly
    unsigned cslot[CF_MAXCLS];
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
          const unsigned pkind = (unsigned)ck[p - 1u];
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
    // offsets; waves: the classes of hosts without gpus first (a wave's lanes hold 64 chunks), all gpu classes in ONE wave
    unsigned off = 0, w = 1, load = 0, bad2 = 0;
    const unsigned nc = s_ncls;
    for (unsigned c = 0; c < nc; ++c) {
      CfClass& cl = ctl->cls[c];
      cl.n = s_cnt[c], cl.off = off;
      off += cl.n;
      const unsigned nch = (cl.n + 63u) / 64u;
      if (nch > 64u) bad2 |= CF_X_SHAPE;
      if (cl.kind == 0u) {
        if (load + nch > 64u) ++w, load = 0;
        cl.wave = w, load += nch;
      }
    }
    if (load) ++w, load = 0;
    for (unsigned c = 0; c < nc; ++c) {
      CfClass& cl = ctl->cls[c];
      if (cl.kind != 0u) cl.wave = w, load += (cl.n + 63u) / 64u;
    }
    if (load > 64u || w > (unsigned)CF_CW) bad2 |= CF_X_SHAPE;
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
  if (k >= K) return;
  CfCtl* ctl = b.ctl;
  if (ctl->inelig) return;
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
  unsigned n_eq = 0, n_nov = 0, grouped = 0;
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
    atomicAdd(&ctl->n_grouped, 1u);
  }
  o.meta = kind | L << 8 | n_eq << 12 | n_nov << 16 | grouped << 20;
  b.jobs[k] = o;
  if (bad) atomicOr(&ctl->inelig, bad);
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
#ifdef __HIP_EMU__  // CF_TRACE=1 in the environment of an emulated run: what the waves of the walk do, to stderr
#include <cstdlib>
static inline bool cf_trace_on() {
  static const bool on = std::getenv("CF_TRACE") != nullptr;
  return on;
}
#define CF_TRACE(...) do { if (cf_trace_on() && lane_id() == 0) std::fprintf(stderr, __VA_ARGS__); } while (0)
#else
#define CF_TRACE(...) ((void)0)
#endif
#ifdef CF_PROF  // timing-study build: ticks of the 100 MHz clock per phase of a step (query, wait at the barrier, verdict, commit) of waves 0 / 1..6 / 7
#define CF_PROF_T(x) const unsigned long long x = cook_ticks()
#define CF_PROF_ADD(i, d) prof[i] += (unsigned)(d)
#else
#define CF_PROF_T(x) ((void)0)
#define CF_PROF_ADD(i, d) ((void)0)
#endif
struct CfPost {  // what a wave says about a job (32 B)
  double fa;     // approximate fitness of its best candidate, 0 = none
  uint32_t w0;   // offer | ambiguous << 31
  uint32_t pos;  // class waves: position; overlay: lane
  uint32_t fc, fm, cls, aux;  // aux: class waves: chunk lane; overlay: live lanes
};
struct CfJobU {  // the job of a step, wave-uniform
  unsigned c, m, kind, L, n_eq, n_nov, grouped, grp, eq0, eq1, nov0, nov1;
};
struct CfCand;
struct CfCmd;
struct CfVlog;
struct CfLds {  // the workgroup's LDS, carved at run time
  uint32_t *fc, *fm;
  uint16_t* cid;        // occupied gpu host << 15 | offer (the class follows from the position)
  CfCand* board;        // [CF_BOARD][CF_WAVES] the class waves' candidates for the steps ahead
  CfCmd* cmd;           // [CF_WAVES] the decider's last command to a class wave
  CfVlog* vlog;         // [CF_VLOG] the decider's verdicts for the bookkeeper
  uint32_t* ack;        // [CF_WAVES] the step of the last command a class wave has obeyed
  uint64_t* attr8;
  uint16_t *goff, *gcnt, *gids;
  CfJob* ring;          // [2][64]
  CfPost* post2;        // [CF_WAVES] exact turns
  CfClass* cls;         // [CF_MAXCLS] the class table (n / off as of the last epoch)
  uint32_t* pw;         // [CF_WAVES][CF_LV] greatest level summaries of a wave's chunks of hosts without gpus
  uint32_t* aw;         // [CF_WAVES][CF_LV] ... of all its chunks, occupied gpu hosts included (a wave re-writes its rows only once the bookkeeper has
                        //   read the last change: CFX_BK_DONE)
  uint32_t* gk;         // [CF_MAXKIND][CF_LV] ... of a gpu kind's chunks
  uint32_t* ovm;        // [64][2] the overlay's free values as of the last batch end
  uint32_t* ovl;        // [64][3] an epoch's overlay list (cid, fc, fm), sorted
  uint32_t* ckept;      // [CF_MAXCLS] kept members / [CF_MAXCLS] inserted / [CF_MAXCLS] new offsets
  uint32_t* misc;       // [0..1] walk mask, [2] sequence number of the last change of level maxima, [6] the batch lane of the step that made it,
                        // [3] overlay list length, [4..5] overlay valid mask as of the last batch end, [7] an exact turn ends the epoch, [8..14] CFX_*
};
static __device__ __forceinline__ unsigned cf_level_of(const uint32_t (&t)[CF_LV], uint32_t fc) {  // greatest level whose threshold fc reaches; CF_LV = none... 0-based count
  unsigned n = 0;
#pragma unroll
  for (int i = 0; i < CF_LV; ++i) n += fc >= t[i] ? 1u : 0u;
  return n;  // members with n levels: levels 0 .. n-1 (t is ascending)
}

// eight level values in eight REGISTERS: as an array inside a structure the compiler kept the whole structure in scratch memory and turned the
// select chain into an indexed scratch load (~1 us each; seen in the ISA of the first build)
struct CfLv8 {
  uint32_t v0, v1, v2, v3, v4, v5, v6, v7;
};
#define CF_FOR8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
static __device__ __forceinline__ uint32_t cf_lv_get(const CfLv8& a, unsigned i) {  // a wave-uniform index
  // (the values pass through OPAQUE_V: a select between LOADS of neighbouring members is folded into one indexed load, and the structure then
  //  stays in scratch memory for good)
  uint32_t x0 = a.v0, x1 = a.v1, x2 = a.v2, x3 = a.v3, x4 = a.v4, x5 = a.v5, x6 = a.v6, x7 = a.v7;
  OPAQUE_V(x0);
  OPAQUE_V(x1);
  OPAQUE_V(x2);
  OPAQUE_V(x3);
  OPAQUE_V(x4);
  OPAQUE_V(x5);
  OPAQUE_V(x6);
  OPAQUE_V(x7);
  uint32_t r = x0;
  r = i == 1u ? x1 : r, r = i == 2u ? x2 : r, r = i == 3u ? x3 : r, r = i == 4u ? x4 : r, r = i == 5u ? x5 : r, r = i == 6u ? x6 : r, r = i == 7u ? x7 : r;
  return r;
}
struct CfChunkLane {  // a class wave's lane = one chunk
  unsigned cls, kind, pos0, n, Tc, Tm;
  unsigned long long pres, dE;
  double hTc, hTm;
  CfLv8 lv, la;
};

// lanes of a class wave <- the chunks of the wave's classes, in class order
static __device__ __forceinline__ void cf_setup_chunks(const CfClass* cls, unsigned nc, unsigned w, unsigned lane, CfChunkLane& c, unsigned& nch_wave) {
  c.cls = 0xFFu, c.kind = 0xFEu, c.pos0 = 0u, c.n = 0u, c.Tc = 1u, c.Tm = 1u, c.pres = 0ull, c.dE = 0ull, c.hTc = 0.0, c.hTm = 0.0;
  unsigned acc = 0;
  for (unsigned ci = 0; ci < nc; ++ci) {
    const CfClass& cl = cls[ci];
    if (cl.wave != w) continue;
    const unsigned nch = (cl.n + 63u) / 64u;
    if (lane >= acc && lane < acc + nch) {
      const unsigned x = lane - acc;
      c.cls = ci, c.kind = cl.kind, c.pos0 = cl.off + 64u * x, c.n = cf_min(64u, cl.n - 64u * x), c.Tc = cl.Tc, c.Tm = cl.Tm, c.dE = cl.dE, c.hTc = cl.hTc, c.hTm = cl.hTm;
      c.pres = c.n >= 64u ? ~0ull : ((1ull << c.n) - 1ull);
    }
    acc += nch;
  }
  nch_wave = acc;
}

// the level summaries of chunk `ch` (wave-uniform) from its members; lanes = positions
static __device__ __forceinline__ void cf_tighten(const CfLds& S, const uint32_t (&t)[CF_LV], unsigned lane, unsigned ch, CfChunkLane& c) {
  const unsigned pos0 = (unsigned)wave_read_lane((int)c.pos0, (int)ch), n = (unsigned)wave_read_lane((int)c.n, (int)ch);
  const unsigned long long pres = wave_read_lane_u64(c.pres, (int)ch);
  const bool in = lane < n && ((pres >> lane) & 1ull);
  const uint32_t fc = S.fc[pos0 + lane], fm = S.fm[pos0 + lane], cid = S.cid[pos0 + lane];
  const unsigned nl = in ? cf_level_of(t, fc) : 0u;
  const bool free_host = !(cid & CF_OCC);
#define CF_TIGHTEN_LEVEL(i)                                                                   \
  {                                                                                           \
    const uint32_t va = wave_max_u32(nl > (unsigned)i ? fm + 1u : 0u);                        \
    const uint32_t vr = wave_max_u32((nl > (unsigned)i && free_host) ? fm + 1u : 0u);         \
    if (lane == ch) c.la.v##i = va, c.lv.v##i = vr;                                           \
  }
  CF_FOR8(CF_TIGHTEN_LEVEL)
#undef CF_TIGHTEN_LEVEL
}
// the wave's rows of the level-maxima tables
static __device__ __forceinline__ void cf_wave_tables(const CfLds& S, unsigned w, unsigned lane, const CfChunkLane& c, bool gpu_wave, unsigned n_kind) {
#define CF_TABLE_LEVEL(i)                                                                                                          \
  {                                                                                                                                \
    const uint32_t p = wave_max_u32(c.kind == 0u ? c.lv.v##i : 0u), a = wave_max_u32(c.kind < 0xFEu ? c.la.v##i : 0u);            \
    if (lane == 0) S.pw[w * CF_LV + i] = p, S.aw[w * CF_LV + i] = a;                                         \
  }
  CF_FOR8(CF_TABLE_LEVEL)
#undef CF_TABLE_LEVEL
  if (gpu_wave)
    for (unsigned k = 1; k < n_kind; ++k) {
#define CF_KIND_LEVEL(i)                                          \
  {                                                               \
    const uint32_t g = wave_max_u32(c.kind == k ? c.lv.v##i : 0u); \
    if (lane == 0) S.gk[k * CF_LV + i] = g;                       \
  }
      CF_FOR8(CF_KIND_LEVEL)
#undef CF_KIND_LEVEL
    }
}

static __device__ __forceinline__ CfJobU cf_job_uniform(const CfJob* jp) {
  CfJobU J;
  const CfJob j = *jp;
  J.c = wave_uniform_u32(j.c), J.m = wave_uniform_u32(j.m);
  const unsigned meta = wave_uniform_u32(j.meta);
  J.kind = meta & 255u, J.L = (meta >> 8) & 15u, J.n_eq = (meta >> 12) & 15u, J.n_nov = (meta >> 16) & 15u, J.grouped = (meta >> 20) & 1u;
  J.grp = wave_uniform_u32(j.grp);
  J.eq0 = wave_uniform_u32(j.eq[0]), J.eq1 = wave_uniform_u32(j.eq[1]), J.nov0 = wave_uniform_u32(j.nov[0]), J.nov1 = wave_uniform_u32(j.nov[1]);
  return J;
}
// the job's constraints other than the gpu kind against offer `id` (lane-parallel; the branches are wave-uniform)
static __device__ __forceinline__ bool cf_cons_ok(const CfLds& S, const CfJobU& J, unsigned id, bool active) {
  bool ok = true;
  if (J.n_eq) {  // user-defined EQUALS (constraints.clj:356-377) on the byte table
    const uint64_t a8 = S.attr8[active ? id : 0u];
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q) {
      const unsigned e = ((q < 2u ? J.eq0 : J.eq1) >> (16u * (q & 1u))) & 0xFFFFu;
      if (q < J.n_eq) ok = ok && (unsigned)((a8 >> (8u * (e >> 8))) & 255ull) == (e & 255u);
    }
  }
  if (J.n_nov) {  // novel-host (constraints.clj:68-94)
#pragma unroll
    for (unsigned q = 0; q < 4u; ++q) {
      const unsigned h = ((q < 2u ? J.nov0 : J.nov1) >> (16u * (q & 1u))) & 0xFFFFu;
      if (q < J.n_nov) ok = ok && id != h;
    }
  }
  if (J.grouped) {  // unique host-placement group (constraints.clj:586-598): cotasks running or placed earlier in this call
    const unsigned g0 = S.goff[J.grp], gn = S.gcnt[J.grp];
    for (unsigned x = 0; x < gn; ++x) ok = ok && id != (unsigned)S.gids[g0 + x];
  }
  return ok;
}
static __device__ __forceinline__ double cf_literal(unsigned Tc, unsigned Tm, unsigned fc, unsigned fm, unsigned jc, unsigned jm, double sc, double sm) {
  // the oracle's expression (cook_oracle.cpp match_impl; match_kernels.hpp fitness_of) on the exact values the fixed-point numbers stand for:
  // running + assigned = total - free, lease + running = total
  const double A = (double)(Tc - fc) * sc, Bm = (double)(Tm - fm) * sm, c = (double)jc * sc, m = (double)jm * sm;
  return ((A + c) / ((double)Tc * sc) + (Bm + m) / ((double)Tm * sm)) / 2.0;
}

// A class wave's answer for job J.  EXACT = false: per relevant class the first feasible member of the first chunk that can hold one; the best of
// them by approximate fitness; `amb` when another member may round to the same fitness.  EXACT = true: the literal fitness of every feasible
// member within the band below fmax; the greatest, lowest offer on ties.
template <bool EXACT>
static __device__ __forceinline__ void cf_class_query(const CfLds& S, const CfJobU& J, unsigned lane, const CfChunkLane& c, double fmax, double sc, double sm, CfPost& out,
                                                       unsigned& scans) {
  out.fa = 0.0, out.w0 = 0u, out.pos = 0u, out.fc = 0u, out.fm = 0u, out.cls = 0u, out.aux = 0u;
  const uint32_t lvL = cf_lv_get(c.lv, J.L);
  unsigned long long m = __ballot(c.kind == J.kind && lvL > J.m);
  bool amb = false;
  unsigned long long best_lit = 0ull;
  while (m) {
    const unsigned ch = (unsigned)__ffsll(m) - 1u;
    ++scans;
    const unsigned pos0 = (unsigned)wave_read_lane((int)c.pos0, (int)ch), n = (unsigned)wave_read_lane((int)c.n, (int)ch);
    const unsigned long long pres = wave_read_lane_u64(c.pres, (int)ch);
    const unsigned Tc = (unsigned)wave_read_lane((int)c.Tc, (int)ch), Tm = (unsigned)wave_read_lane((int)c.Tm, (int)ch), cls = (unsigned)wave_read_lane((int)c.cls, (int)ch);
    const double hTc = wave_read_lane_f64(c.hTc, (int)ch), hTm = wave_read_lane_f64(c.hTm, (int)ch);
    const unsigned long long dE = wave_read_lane_u64(c.dE, (int)ch);
    const bool in = lane < n && ((pres >> lane) & 1ull);
    const unsigned q = pos0 + lane;
    const uint32_t fc = S.fc[q], fm = S.fm[q], cid = S.cid[q];
    const bool room = in && fc >= J.c && fm >= J.m && !(cid & CF_OCC);
    const bool ok = room && cf_cons_ok(S, J, cid & CF_IDMASK, room);
    const unsigned long long b = __ballot(ok);
    const double fa = ok ? 1.0 - ((double)(fc - J.c) * hTc + (double)(fm - J.m) * hTm) : 0.0;
    if (!EXACT) {
      if (b) {
        const unsigned q0 = (unsigned)__ffsll(b) - 1u;
        const unsigned long long E = (unsigned long long)fc * Tm + (unsigned long long)fm * Tc;
        const unsigned long long E0 = wave_read_lane_u64(E, (int)q0);
        // another feasible member of the chunk inside the band, or the band reaching the chunk's end (then the next chunk may hold one)
        const bool a2 = (__ballot(ok && lane != q0 && E <= E0 + dE) != 0ull) || (wave_read_lane_u64(E, (int)(n - 1u)) <= E0 + dE && n == 64u);
        const double f0 = wave_read_lane_f64(fa, (int)q0);
        if (f0 > out.fa + CF_BAND) {
          amb = a2;
        } else if (f0 >= out.fa - CF_BAND) {
          amb = true;
        }
        if (f0 > out.fa) {
          out.fa = f0, out.w0 = (unsigned)wave_read_lane((int)(cid & CF_IDMASK), (int)q0), out.pos = pos0 + q0, out.fc = (unsigned)wave_read_lane((int)fc, (int)q0),
          out.fm = (unsigned)wave_read_lane((int)fm, (int)q0), out.cls = cls, out.aux = ch;
        } else if (f0 >= out.fa - CF_BAND) {
          amb = true;
        }
        m &= ~__ballot(c.cls == cls);  // the class is answered
      } else {
        m &= ~(1ull << ch);
      }
    } else {
      const bool cand = ok && fa >= fmax - CF_BAND;
      const double lit = cand ? cf_literal(Tc, Tm, fc, fm, J.c, J.m, sc, sm) : 0.0;
      const unsigned long long lb = (unsigned long long)__double_as_longlong(lit);
      const unsigned long long mx = wave_max_u64(lb);
      if (mx != 0ull) {
        const unsigned idmin = ~wave_max_u32((cand && lb == mx) ? ~(cid & CF_IDMASK) : 0u);
        if (mx > best_lit || (mx == best_lit && idmin < out.w0)) {
          best_lit = mx;
          const unsigned q0 = (unsigned)__ffsll(__ballot(cand && lb == mx && (cid & CF_IDMASK) == idmin)) - 1u;
          out.fa = __longlong_as_double((long long)mx), out.w0 = idmin, out.pos = pos0 + q0, out.fc = (unsigned)wave_read_lane((int)fc, (int)q0),
          out.fm = (unsigned)wave_read_lane((int)fm, (int)q0), out.cls = cls, out.aux = ch;
        }
      }
      m &= ~(1ull << ch);
    }
  }
  if (!EXACT && amb) out.w0 |= 0x80000000u;
}

// A class wave's answer for the decider: per relevant class the FIRST feasible member of the first chunk that can hold one (sorted by E: the class's
// best); of several classes of the wave the one of greatest approximate fitness.  w0 bit 31: another member / class may round to the same fitness.
// Written for the decider's critical path: nothing is computed that only the winner needs, a wave with one relevant class computes no fitness at all.
static __device__ __forceinline__ void cf_class_answer(const CfLds& S, const CfJobU& J, unsigned lane, const CfChunkLane& c, CfPost& out, unsigned& scans) {
  out.fa = 0.0, out.w0 = 0u, out.pos = 0u, out.fc = 0u, out.fm = 0u, out.cls = 0u, out.aux = 0u;
  const uint32_t lvL = cf_lv_get(c.lv, J.L);
  unsigned long long m = __ballot(c.kind == J.kind && lvL > J.m);
  bool amb = false, have = false;
  while (m) {
    const unsigned ch = (unsigned)__ffsll(m) - 1u;
    ++scans;
    const unsigned pos0 = (unsigned)wave_read_lane((int)c.pos0, (int)ch), n = (unsigned)wave_read_lane((int)c.n, (int)ch);
    const unsigned long long pres = wave_read_lane_u64(c.pres, (int)ch);
    const bool in = lane < n && ((pres >> lane) & 1ull);
    const unsigned q = pos0 + lane;
    const uint32_t fc = S.fc[q], fm = S.fm[q], cid = S.cid[q];
    const bool room = in && fc >= J.c && fm >= J.m && !(cid & CF_OCC);
    const bool ok = room && cf_cons_ok(S, J, cid & CF_IDMASK, room);
    const unsigned long long b = __ballot(ok);
    if (b == 0ull) {
      m &= ~(1ull << ch);
      continue;
    }
    const unsigned q0 = (unsigned)__ffsll(b) - 1u;
    const unsigned fc0 = (unsigned)wave_read_lane((int)fc, (int)q0), fm0 = (unsigned)wave_read_lane((int)fm, (int)q0), cid0 = (unsigned)wave_read_lane((int)cid, (int)q0);
    const unsigned cls = (unsigned)wave_read_lane((int)c.cls, (int)ch);
    m &= ~__ballot(c.cls == cls);  // the class is answered
    const bool tie = (cid0 & CF_TIE) != 0u;
    if (!have && m == 0ull) {  // the wave's only class with a candidate: the decider works the fitness out itself
      out.fa = 1.0, out.w0 = cid0 & CF_IDMASK, out.pos = pos0 + q0, out.fc = fc0, out.fm = fm0, out.cls = cls, out.aux = ch;
      amb = tie;
      break;
    }
    const double hTc = wave_read_lane_f64(c.hTc, (int)ch), hTm = wave_read_lane_f64(c.hTm, (int)ch);
    const double f0 = 1.0 - ((double)(fc0 - J.c) * hTc + (double)(fm0 - J.m) * hTm);
    if (!have || f0 > out.fa + CF_BAND) amb = tie;
    else if (f0 >= out.fa - CF_BAND) amb = true;
    if (!have || f0 > out.fa) out.fa = f0, out.w0 = cid0 & CF_IDMASK, out.pos = pos0 + q0, out.fc = fc0, out.fm = fm0, out.cls = cls, out.aux = ch;
    have = true;
  }
  if (amb) out.w0 |= 0x80000000u;
}

struct CfOvLane {  // the overlay wave's lane = one offer this call has placed on
  unsigned valid, id, cls, fc, fm, Tc, Tm;
  double hTc, hTm;
};
template <bool EXACT>
static __device__ __forceinline__ void cf_overlay_query(const CfLds& S, const CfJobU& J, unsigned lane, const CfOvLane& o, double fmax, double sc, double sm, CfPost& out) {
  out.fa = 0.0, out.w0 = 0u, out.pos = 0u, out.fc = 0u, out.fm = 0u, out.cls = 0u;
  out.aux = (unsigned)__popcll(__ballot(o.valid != 0u));
  if (J.kind != 0u) return;  // (the overlay holds hosts without gpus only: gpu hosts take one job and stay in their chunk)
  const bool room = o.valid && o.fc >= J.c && o.fm >= J.m;
  const bool ok = room && cf_cons_ok(S, J, o.id, room);
  const double fa = ok ? 1.0 - ((double)(o.fc - J.c) * o.hTc + (double)(o.fm - J.m) * o.hTm) : 0.0;
  if (!EXACT) {
    const unsigned long long fb = (unsigned long long)__double_as_longlong(fa);
    const unsigned long long mx = wave_max_u64(fb);
    if (mx == 0ull) return;
    const double fmx = __longlong_as_double((long long)mx);
    const unsigned long long near = __ballot(ok && fa >= fmx - CF_BAND);
    const unsigned l0 = (unsigned)__ffsll(__ballot(ok && fb == mx)) - 1u;
    out.fa = fmx, out.w0 = (unsigned)wave_read_lane((int)o.id, (int)l0) | ((near & (near - 1ull)) ? 0x80000000u : 0u), out.pos = l0,
    out.fc = (unsigned)wave_read_lane((int)o.fc, (int)l0), out.fm = (unsigned)wave_read_lane((int)o.fm, (int)l0), out.cls = (unsigned)wave_read_lane((int)o.cls, (int)l0);
  } else {
    const bool cand = ok && fa >= fmax - CF_BAND;
    const double lit = cand ? cf_literal(o.Tc, o.Tm, o.fc, o.fm, J.c, J.m, sc, sm) : 0.0;
    const unsigned long long lb = (unsigned long long)__double_as_longlong(lit);
    const unsigned long long mx = wave_max_u64(lb);
    if (mx == 0ull) return;
    const unsigned idmin = ~wave_max_u32((cand && lb == mx) ? ~o.id : 0u);
    const unsigned l0 = (unsigned)__ffsll(__ballot(cand && lb == mx && o.id == idmin)) - 1u;
    out.fa = __longlong_as_double((long long)mx), out.w0 = idmin, out.pos = l0, out.fc = (unsigned)wave_read_lane((int)o.fc, (int)l0),
    out.fm = (unsigned)wave_read_lane((int)o.fm, (int)l0), out.cls = (unsigned)wave_read_lane((int)o.cls, (int)l0);
  }
}

// every wave reads the posts of a step and comes to the same verdict: src = the winning wave (-1 none), amb = an exact turn is needed
struct CfVerdict {
  int src;
  bool amb;
  double fmax;
  unsigned id, pos, fc, fm, cls, aux, ov_live;
};
template <bool EXACT>
static __device__ __forceinline__ CfVerdict cf_verdict(const CfPost* posts, unsigned lane) {
  CfVerdict v;
  const bool has = lane < (unsigned)(CF_CW + 1);
  CfPost p;
  p.fa = 0.0, p.w0 = 0u, p.pos = p.fc = p.fm = p.cls = p.aux = 0u;
  if (has) p = posts[lane];
  const unsigned long long fb = (unsigned long long)__double_as_longlong(p.fa);
  const unsigned long long mx = wave_max_u64(fb);
  v.ov_live = (unsigned)wave_read_lane((int)p.aux, 0);
  v.fmax = __longlong_as_double((long long)mx);
  v.src = -1, v.amb = false, v.id = v.pos = v.fc = v.fm = v.cls = v.aux = 0u;
  if (mx == 0ull) return v;
  unsigned wl;
  if (!EXACT) {
    const unsigned long long near = __ballot(has && p.fa > 0.0 && p.fa >= v.fmax - CF_BAND);
    v.amb = (near & (near - 1ull)) != 0ull || __ballot(has && p.fa > 0.0 && p.fa >= v.fmax - CF_BAND && (p.w0 >> 31)) != 0ull;
    wl = (unsigned)__ffsll(__ballot(has && fb == mx)) - 1u;
  } else {  // greatest literal fitness, lowest offer
    const unsigned idmin = ~wave_max_u32((has && fb == mx) ? ~p.w0 : 0u);
    wl = (unsigned)__ffsll(__ballot(has && fb == mx && p.w0 == idmin)) - 1u;
  }
  v.src = (int)wl;
  v.id = (unsigned)wave_read_lane((int)(p.w0 & 0x7FFFFFFFu), (int)wl), v.pos = (unsigned)wave_read_lane((int)p.pos, (int)wl), v.fc = (unsigned)wave_read_lane((int)p.fc, (int)wl),
  v.fm = (unsigned)wave_read_lane((int)p.fm, (int)wl), v.cls = (unsigned)wave_read_lane((int)p.cls, (int)wl), v.aux = (unsigned)wave_read_lane((int)p.aux, (int)wl);
  return v;
}

// the books of one job's "an offer of the cluster has room for it" from the class arrays' side: the level maxima over every chunk of every wave
static __device__ __forceinline__ bool cf_chunks_have_room(const CfLds& S, unsigned L, unsigned m) {
  bool r = false;
  for (unsigned x = 1; x <= (unsigned)CF_CW; ++x) r = r || S.aw[x * CF_LV + L] > m;
  return r;
}
// a word another wave of the workgroup writes, the same value in every lane (lane 0 reads it)
static __device__ __forceinline__ unsigned cf_poll(const uint32_t* p) { return (unsigned)wave_read_lane((int)ld_wg(p), 0); }

// ---- the walk ---------------------------------------------------------------------------------------------------------------------------------------------
// Wave 0 DECIDES alone: its lanes 0..57 hold the overlay, lanes 58..63 take, for the job of the step, the candidate each class wave has PUBLISHED on
// the board (LDS) — the class waves answer the walked jobs of the batch ahead of the decider, up to CF_BOARD steps, and answer again from the step
// after every member the decider takes out of their arrays (a command + a version number per class wave; a candidate counts when it carries the
// step and the version the decider expects).  One evaluation of 64 lanes, one wave maximum, the commit in registers: no barrier in a step.  The
// bookkeeper (wave 7) follows the decider's verdict log.  Steps that need every wave in lockstep — several candidates inside the guard band (the
// literal fitness decides), the end of an epoch, the end of the batch — are COLLECTIVE turns: the decider raises a mode word, every wave comes to a
// barrier, the turn runs as in the first (lockstep) form of this kernel.
constexpr unsigned CF_BOARD = 8;    // steps the class waves may run ahead of the decider (a power of two)
constexpr unsigned CF_VLOG = 16;    // verdicts the bookkeeper may lag behind
constexpr unsigned CF_OVL = 58;     // overlay lanes (lanes 58..63 are the candidates of class waves 1..6)
constexpr unsigned CF_EPOCH_AT = COOK_SHAPE(58, 8);  // live overlay lanes that end an epoch
enum : unsigned { CFM_EXACT = 1u, CFM_EPOCH = 2u, CFM_BATCH_END = 3u };
enum : unsigned { CFC_REMOVE = 1u, CFC_GPU_PLACE = 2u, CFC_NONE = 3u };
enum : unsigned { CFX_HEAD_SEQ = 8, CFX_MODE = 9, CFX_DRAIN = 10, CFX_BK_DONE = 11, CFX_EX_LANE = 12, CFX_FMAX_LO = 13, CFX_FMAX_HI = 14 };  // words of CfLds::misc
struct CfCand {  // a class wave's answer for one step (40 B); tag = step << 8 | version, stored LAST
  uint32_t tag, pos, fc, fm, cid, flags;  // flags: 1 another member may round to the same fitness, 2 no candidate
  double hTc, hTm;
};
struct CfCmd {  // decider -> class wave (32 B); ver stored LAST
  uint32_t ver, kind, pos, nfc, nfm, lane_s, seq, pad;
};
struct CfVlog {  // decider -> bookkeeper (32 B); seq stored LAST
  uint32_t seq, info, id, ofc, ofm, nfc, nfm, pad;  // info: batch lane | matched << 8 | from the overlay << 9 | opens a lane << 10 | class wave << 12
};

static __device__ __forceinline__ void cf_walk_pool(char* lds, const MatchIn* __restrict__ inp, const MatchState& st, const CfBuf& b) {
  const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
  CfCtl* ctl = b.ctl;
  const unsigned K = inp->K, M = ctl->M, G = inp->G;
  const unsigned NP = (M + 63u) & ~63u;
  const unsigned long long t_start = cook_ticks();
  uint32_t t[CF_LV];
#pragma unroll
  for (int i = 0; i < CF_LV; ++i) t[i] = ctl->t[i];
  const unsigned kc = ctl->kc, km = ctl->km, cmin = ctl->cmin, mmin = ctl->mmin, n_kind = ctl->n_kind, n_cls = ctl->n_cls;
  const double sc = cf_pow2(-(int)kc), sm = cf_pow2(-(int)km);
  const bool any_eq = ctl->any_eq != 0u, any_group = ctl->any_group != 0u;
  // ---- group table sizes (needed for the layout): entries per unique group = running cotasks on hosts of this call + pending members
  __shared__ unsigned s_total, s_wsum[CF_WAVES];
  constexpr unsigned GPT = CF_MAXG / CF_THREADS;
  unsigned gsz[GPT];
  unsigned gsum = 0;
#pragma unroll
  for (unsigned x = 0; x < GPT; ++x) {
    const unsigned g = tid * GPT + x;
    unsigned sz = 0;
    if (any_group && g < G && b.gcount[g] != 0u) sz = (inp->g_run_off ? inp->g_run_off[g + 1] - inp->g_run_off[g] : 0u) + b.gcount[g];
    gsz[x] = sz, gsum += sz;
  }
  unsigned incl = gsum;  // inclusive scan over the workgroup
  for (unsigned d = 1; d < 64u; d <<= 1) {
    const unsigned y = shfl_up_t<unsigned>(incl, d);
    if (lane >= d) incl += y;
  }
  if (lane == 63u) s_wsum[w] = incl;
  __syncthreads();
  unsigned wbase = 0;
  for (unsigned x = 0; x < w; ++x) wbase += s_wsum[x];
  if (tid == CF_THREADS - 1) s_total = wbase + incl;
  __syncthreads();
  const unsigned Stot = any_group ? s_total : 0u, Gl = any_group ? G : 0u;
  // ---- layout
  CfLds S;
  {
    char* p = lds;
    S.fc = (uint32_t*)p, p += NP * 4u;
    S.fm = (uint32_t*)p, p += NP * 4u;
    S.cid = (uint16_t*)p, p += NP * 2u;
    p = lds + (((unsigned)(p - lds) + 7u) & ~7u);
    S.attr8 = (uint64_t*)p;
    if (any_eq) p += M * 8u;
    S.goff = (uint16_t*)p, p += (Gl + 1u) * 2u;
    S.gcnt = (uint16_t*)p, p += Gl * 2u;
    S.gids = (uint16_t*)p, p += Stot * 2u;
    p = lds + (((unsigned)(p - lds) + 15u) & ~15u);
    S.ring = (CfJob*)p, p += 2u * 64u * sizeof(CfJob);
    S.board = (CfCand*)p, p += CF_BOARD * CF_WAVES * sizeof(CfCand);
    S.cmd = (CfCmd*)p, p += CF_WAVES * sizeof(CfCmd);
    S.vlog = (CfVlog*)p, p += CF_VLOG * sizeof(CfVlog);
    S.post2 = (CfPost*)p, p += CF_WAVES * sizeof(CfPost);
    S.cls = (CfClass*)p, p += CF_MAXCLS * sizeof(CfClass);
    S.pw = (uint32_t*)p, p += CF_WAVES * CF_LV * 4u;
    S.aw = (uint32_t*)p, p += CF_WAVES * CF_LV * 4u;
    S.gk = (uint32_t*)p, p += CF_MAXKIND * CF_LV * 4u;
    S.ack = (uint32_t*)p, p += CF_WAVES * 4u;
    S.ovm = (uint32_t*)p, p += 128u * 4u;
    S.ovl = (uint32_t*)p, p += 192u * 4u;
    S.ckept = (uint32_t*)p, p += 3u * CF_MAXCLS * 4u;
    S.misc = (uint32_t*)p, p += 16u * 4u;
    if ((unsigned)(p - lds) > CF_LDS_BYTES) {  // (the host checks the same sum before it launches: cf_lds_bytes_host)
      if (tid == 0) atomicOr(&ctl->inelig, (unsigned)CF_X_SHAPE), st.summary[3] = 0xDEADu;
      return;
    }
  }
  // ---- prologue: class arrays, byte table, group table, job ring
  for (unsigned q = tid; q < NP; q += CF_THREADS) {
    S.fc[q] = q < M ? b.pos_fc[q] : 0u, S.fm[q] = q < M ? b.pos_fm[q] : 0u, S.cid[q] = q < M ? (uint16_t)b.pos_cid[q] : (uint16_t)0xFFFFu;
  }
  if (any_eq)
    for (unsigned v = tid; v < M; v += CF_THREADS) S.attr8[v] = b.attr8[v];
  if (any_group) {
    unsigned off = wbase + incl - gsum;
#pragma unroll
    for (unsigned x = 0; x < GPT; ++x) {
      const unsigned g = tid * GPT + x;
      if (g <= G) S.goff[g] = (uint16_t)off;
      if (g < G) {
        unsigned cnt = 0;
        if (gsz[x]) {
          const unsigned r0 = inp->g_run_off ? inp->g_run_off[g] : 0u, r1 = inp->g_run_off ? inp->g_run_off[g + 1] : 0u;
          for (unsigned r = r0; r < r1; ++r) {
            const uint32_t h = inp->g_run_host[r];
            const uint32_t v = h <= b.max_host ? b.h2o[h] : 0xFFFFFFFFu;
            if (v != 0xFFFFFFFFu) S.gids[off + cnt++] = (uint16_t)v;
          }
        }
        S.gcnt[g] = (uint16_t)cnt;
        off += gsz[x];
      }
    }
  }
  if (w == CF_WAVES - 1 && lane < cf_min(64u, K)) S.ring[lane] = b.jobs[lane];
  for (unsigned x = tid; x < n_cls; x += CF_THREADS) S.cls[x] = ctl->cls[x];
  for (unsigned x = tid; x < 16u; x += CF_THREADS) S.misc[x] = 0u;
  for (unsigned x = tid; x < CF_WAVES * CF_LV; x += CF_THREADS) S.pw[x] = 0u, S.aw[x] = 0u;
  for (unsigned x = tid; x < CF_WAVES; x += CF_THREADS) S.ack[x] = 0u, S.cmd[x].ver = 0u;
  for (unsigned x = tid; x < CF_MAXKIND * CF_LV; x += CF_THREADS) S.gk[x] = 0u;
  for (unsigned x = tid; x < CF_BOARD * CF_WAVES; x += CF_THREADS) S.board[x].tag = 0xFFFFFFFFu;
  for (unsigned x = tid; x < CF_VLOG; x += CF_THREADS) S.vlog[x].seq = 0u;
  __syncthreads();
  // ---- wave state
  CfChunkLane c;
  unsigned nch_wave = 0;
  cf_setup_chunks(S.cls, n_cls, w, lane, c, nch_wave);
  c.lv = CfLv8{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, c.la = c.lv;
  unsigned gpu_wave = 0;  // the wave that holds the gpu classes
  for (unsigned ci = 0; ci < n_cls; ++ci)
    if (S.cls[ci].kind != 0u) gpu_wave = S.cls[ci].wave;
  const bool is_class_wave = w >= 1u && w <= (unsigned)CF_CW;
  const bool is_books = w == (unsigned)CF_WAVES - 1u;
  unsigned live_waves = 0;  // class waves that hold classes (the others sleep through the batches)
  for (unsigned ci = 0; ci < n_cls; ++ci) live_waves |= 1u << S.cls[ci].wave;
  if (is_class_wave) {
    for (unsigned ch = 0; ch < nch_wave; ++ch) cf_tighten(S, t, lane, ch, c);
    cf_wave_tables(S, w, lane, c, w == gpu_wave, n_kind);
  }
  CfOvLane o;
  o.valid = 0u, o.id = 0u, o.cls = 0u, o.fc = 0u, o.fm = 0u, o.Tc = 1u, o.Tm = 1u, o.hTc = 0.0, o.hTm = 0.0;
  unsigned o_ver = 0;  // decider lanes 58..63 / class waves: the version a candidate must carry (commands AND collective turns move it on)
  unsigned o_cmd = 0;  // decider lanes 58..63: commands given to "their" class wave so far; class waves: commands obeyed
  unsigned retable_seq = 0;  // class wave: the step of its last change of level maxima (the next one waits until the bookkeeper has read this one)
  // bookkeeper (wave 7): lanes = the jobs of the batch
  unsigned bk_c = 0, bk_m = 0, bk_L = 0, bk_kind = 0, bk_cnt = 0;  // cnt: overlay lanes with room for the job
  bool bk_cha = false, bk_b1 = false, bk_walk = false;
  int bk_res = -1;
  unsigned bk_seen = 0, bk_done = 0;
  unsigned matched = 0, head = 0;
  unsigned minfc_all = ctl->minfc_all, minfm_all = ctl->minfm_all;
  unsigned st_scans = 0, st_exact = 0, st_open = 0, st_ovwin = 0, st_gpu = 0, st_epochs = 0, st_tight = 0, st_walked = 0, st_dead = 0, st_opendead = 0, st_spins = 0;
  unsigned long long tk_epoch = 0, tk_pre = 0, tk_wait = 0;
  __syncthreads();
  const unsigned long long t_loop = cook_ticks();
  CfJob nxt;  // the stager's registers: the job of lane `lane` in the next batch
  nxt.c = nxt.m = nxt.meta = nxt.grp = nxt.eq[0] = nxt.eq[1] = nxt.nov[0] = nxt.nov[1] = 0u;
  unsigned seq_base = 1;  // the step number of the batch's first walked job (steps count from 1 over the whole call)
#ifdef CF_PROF
  unsigned prof[4] = {0, 0, 0, 0};
#endif
  // the bookkeeper follows a change of level maxima: the jobs BEHIND the step that made it see the new ones
  auto follow_tables = [&]() {
    const unsigned sq = S.misc[2];
    if (sq != bk_seen) {
      bk_seen = sq;
      if (lane > S.misc[6]) bk_cha = cf_chunks_have_room(S, bk_L, bk_m);
    }
  };
  // the books of the batch's jobs behind lane s after a placement (ofc, ofm) -> (nfc, nfm)
  auto books_placement = [&](unsigned s, unsigned id, bool from_overlay, bool opens, unsigned ofc, unsigned ofm, unsigned nfc, unsigned nfm) {
    minfc_all = cf_min(minfc_all, nfc), minfm_all = cf_min(minfm_all, nfm);  // (the bookkeeper's: it sees every placement)
    if (lane == s) bk_res = (int)id;
    if (lane > s) {
      bk_b1 = bk_b1 || bk_c > nfc || bk_m > nfm;
      if (from_overlay) bk_cnt -= (ofc >= bk_c && ofm >= bk_m && !(nfc >= bk_c && nfm >= bk_m)) ? 1u : 0u;
      else if (opens) bk_cnt += (nfc >= bk_c && nfm >= bk_m) ? 1u : 0u;
    }
  };
  // a class wave takes a member out of chunk lane `ch` (position pos; gpu hosts stay, occupied) and keeps its summaries exact
  auto class_remove = [&](unsigned pos, unsigned ch, bool gpu_place, unsigned ofc, unsigned ofm, unsigned nfc, unsigned nfm, unsigned s, unsigned seq) {
    bool retable = false;
    if (gpu_place) {
      if (lane == 0) S.fc[pos] = nfc, S.fm[pos] = nfm, S.cid[pos] = (uint16_t)(S.cid[pos] | CF_OCC);
      wave_sync();
      retable = true;
    } else {
      const unsigned q0 = pos - (unsigned)wave_read_lane((int)c.pos0, (int)ch);
      if (lane == ch) c.pres &= ~(1ull << q0);
      CF_TRACE("class wave %u: position %u leaves chunk lane %u (member %u): present %llx\n", w, pos, ch, q0, c.pres);
      const unsigned nl = cf_level_of(t, ofc);
#define CF_WAS_MAX(i) retable = retable || ((unsigned)i < nl && (unsigned)wave_read_lane((int)c.lv.v##i, (int)ch) == ofm + 1u);
      CF_FOR8(CF_WAS_MAX)
#undef CF_WAS_MAX
    }
    if (retable) {
      while (retable_seq != 0u && cf_poll(&S.misc[CFX_BK_DONE]) < retable_seq) SPIN_PAUSE_NEAR();  // (the bookkeeper reads the rows of the last change)
      cf_tighten(S, t, lane, ch, c);
      cf_wave_tables(S, w, lane, c, w == gpu_wave, n_kind);
      if (lane == 0) S.misc[6] = s, st_wg(&S.misc[2], seq);
      retable_seq = seq;
      ++st_tight;
    }
  };
  // the chunk lane of a position of this class wave
  auto chunk_of = [&](unsigned pos) -> unsigned { return (unsigned)__ffsll(__ballot(pos >= c.pos0 && pos < c.pos0 + c.n)) - 1u; };

  for (unsigned base = 0; base < K; base += 64u) {
    const unsigned bn = cf_min(64u, K - base);
    const unsigned slot = (base >> 6) & 1u;
    // ---- batch pre-check (bookkeeper): who must be visited?
    if (is_books) {
      const unsigned long long t0 = cook_ticks();
      if (base + 64u + lane < K) nxt = b.jobs[base + 64u + lane];  // (arrives while the batch is walked)
      const CfJob j = S.ring[slot * 64u + (lane < bn ? lane : 0u)];
      bk_c = j.c, bk_m = j.m, bk_kind = j.meta & 255u, bk_L = (j.meta >> 8) & 15u;
      bk_seen = S.misc[2];
      bool chr = false;
      bk_cha = cf_chunks_have_room(S, bk_L, bk_m);
      if (bk_kind == 0u) {
        for (unsigned x = 1; x <= (unsigned)CF_CW; ++x) chr = chr || S.pw[x * CF_LV + bk_L] > bk_m;
      } else if (bk_kind != CF_KIND_NONE) {
        chr = S.gk[bk_kind * CF_LV + bk_L] > bk_m;
      }
      bk_cnt = 0;
      const unsigned long long ovv = (unsigned long long)S.misc[4] | (unsigned long long)S.misc[5] << 32;
      for (unsigned long long mm = ovv; mm; mm &= mm - 1ull) {
        const unsigned l = (unsigned)__ffsll(mm) - 1u;
        bk_cnt += (S.ovm[2u * l] >= bk_c && S.ovm[2u * l + 1u] >= bk_m) ? 1u : 0u;
      }
      bk_b1 = !(bk_c <= minfc_all && bk_m <= minfm_all);
      bk_walk = lane < bn && (chr || (bk_kind == 0u && bk_cnt != 0u));
      bk_res = -1;
      const unsigned long long wm0 = __ballot(bk_walk);
      if (lane == 0) S.misc[0] = (unsigned)wm0, S.misc[1] = (unsigned)(wm0 >> 32);
      tk_pre += cook_ticks() - t0;
    }
    EMU_SITE("classfit: batch");
    __syncthreads();
    const unsigned long long walkmask = wave_uniform_u64((unsigned long long)S.misc[0] | (unsigned long long)S.misc[1] << 32);
    const unsigned nw = (unsigned)__popcll(walkmask);
    st_walked += nw;
    auto seq_of = [&](unsigned s) -> unsigned { return seq_base + (unsigned)__popcll(walkmask & ((1ull << s) - 1ull)); };
    unsigned long long todo = walkmask;       // decider: the walked jobs not decided yet; class waves: not answered yet
    bool batch_done = false;
    while (!batch_done) {
      unsigned md = 0;  // the collective turn this wave leaves its loop for
      if (w == 0) {
        // ================================================= the decider =================================================
        while (md == 0u) {
          if (todo == 0ull) {
            md = (seq_base + nw - 1u) << 4 | CFM_BATCH_END;
            if (lane == 0) S.misc[CFX_DRAIN] = seq_base + nw - 1u, st_wg(&S.misc[CFX_MODE], md);
            break;
          }
          const unsigned s = (unsigned)__ffsll(todo) - 1u;
          const unsigned seq = seq_of(s);
          CF_TRACE("decider: step %u lane %u\n", seq, s);
          CF_PROF_T(p0);
          if (lane == 0) st_wg(&S.misc[CFX_HEAD_SEQ], seq);
          const CfJobU J = cf_job_uniform(&S.ring[slot * 64u + s]);
          // the candidates of the six class waves into lanes 58..63
          unsigned cpos = 0, cflags = 2u;
          if (lane >= CF_OVL && ((live_waves >> (lane - CF_OVL + 1u)) & 1u)) {
            const CfCand* e = &S.board[(seq & (CF_BOARD - 1u)) * CF_WAVES + (lane - CF_OVL + 1u)];
            const unsigned want = seq << 8 | (o_ver & 255u);
            for (;;) {
              const unsigned tg = ld_wg(&e->tag);
              COMPILER_FENCE();
              if (tg == want) break;
              ++st_spins;
              SPIN_PAUSE_NEAR();
            }
            const CfCand cd = *e;
            cpos = cd.pos, cflags = cd.flags;
            o.valid = (cd.flags & 2u) ? 0u : 1u, o.id = cd.cid & CF_IDMASK, o.cls = cd.cid >> 16, o.fc = cd.fc, o.fm = cd.fm, o.hTc = cd.hTc, o.hTm = cd.hTm;
          }
          else if (lane >= CF_OVL) o.valid = 0u;
          wave_sync();
          CF_TRACE("decider: step %u has its candidates\n", seq);
          CF_PROF_T(p1);
          // one evaluation of the 64 lanes
          const bool isov = lane < CF_OVL;
          const bool room = o.valid && o.fc >= J.c && o.fm >= J.m && (!isov || J.kind == 0u);
          const bool ok = room && (!isov || cf_cons_ok(S, J, o.id, room));
          const double fa = ok ? 1.0 - ((double)(o.fc - J.c) * o.hTc + (double)(o.fm - J.m) * o.hTm) : 0.0;
          const float ff = (float)fa;
          const float mx = wave_max_f32(ff);
          unsigned l0 = 0;
          bool amb = false, any = mx > 0.0f;
          if (any) {
            l0 = (unsigned)__ffsll(__ballot(ok && ff == mx)) - 1u;
            const double f0 = wave_read_lane_f64(fa, (int)l0);
            const unsigned long long near = __ballot(ok && fa >= f0 - CF_BAND);  // (a lane above f0 is in here too: one bit = l0 is the greatest alone)
            amb = (near & (near - 1ull)) != 0ull || (__ballot(ok && !isov && (cflags & 1u) && fa >= f0 - CF_BAND) != 0ull);
          }
          CF_PROF_T(p2);
#ifdef __HIP_EMU__
          if (cf_trace_on() && (ok || (lane >= CF_OVL))) std::fprintf(stderr, "  decider lane %u: valid %u offer %u fc %u fm %u fa %.17g ff %.9g mx %.9g l0 %u\n", lane, o.valid, o.id, o.fc, o.fm, fa, (double)ff, (double)mx, l0);
#endif
          if (amb) {  // the literal fitness decides: every wave in lockstep
            md = seq << 4 | CFM_EXACT;
            const unsigned long long fb = (unsigned long long)__double_as_longlong(wave_read_lane_f64(fa, (int)l0));
            if (lane == 0)
              S.misc[CFX_EX_LANE] = s, S.misc[CFX_FMAX_LO] = (unsigned)fb, S.misc[CFX_FMAX_HI] = (unsigned)(fb >> 32), S.misc[CFX_DRAIN] = seq - 1u, st_wg(&S.misc[CFX_MODE], md);
            break;
          }
          todo &= todo - 1ull;
          // ---- commit
          while (seq - cf_poll(&S.misc[CFX_BK_DONE]) >= CF_VLOG) SPIN_PAUSE_NEAR();  // (the bookkeeper is this far behind: never seen)
          CfVlog* vl = &S.vlog[seq & (CF_VLOG - 1u)];
          if (!any) {
            if (lane == 0) {
              vl->info = s;
              COMPILER_FENCE();
              st_wg(&vl->seq, seq);
            }
          } else {
            const unsigned ofc = (unsigned)wave_read_lane((int)o.fc, (int)l0), ofm = (unsigned)wave_read_lane((int)o.fm, (int)l0), id = (unsigned)wave_read_lane((int)o.id, (int)l0);
            const unsigned nfc = ofc - J.c, nfm = ofm - J.m;
            const bool dead = nfc < cmin || nfm < mmin;
            const bool from_ov = l0 < CF_OVL;
            const bool gpu_place = !from_ov && J.kind != 0u;
            const bool opens = !from_ov && !gpu_place && !dead;
            ++matched;
            if (base + s == 0u) head = 1u;
            unsigned live = (unsigned)__popcll(__ballot(isov && o.valid != 0u));
            if (from_ov) {
              ++st_ovwin;
              if (lane == l0) {
                o.fc = nfc, o.fm = nfm;
                if (dead) o.valid = 0u;
              }
              if (dead) ++st_dead, --live;
            } else {
              // the member leaves its class wave's arrays: a command, and the wave's candidates for the later steps count no more
              if (lane == l0) {
                ++o_ver, ++o_cmd;
                CfCmd* cm = &S.cmd[l0 - CF_OVL + 1u];
                cm->kind = gpu_place ? CFC_GPU_PLACE : CFC_REMOVE, cm->pos = cpos, cm->nfc = nfc, cm->nfm = nfm, cm->lane_s = s, cm->seq = seq;
                COMPILER_FENCE();
                st_wg(&cm->ver, o_cmd);
              }
              if (opens) {
                ++st_open;
                const unsigned lf = (unsigned)__ffsll(~__ballot(o.valid != 0u || !isov)) - 1u;  // (a free overlay lane: a full overlay ended the epoch at once)
                const unsigned cls2 = (unsigned)wave_read_lane((int)o.cls, (int)l0);
                const double hTc2 = wave_read_lane_f64(o.hTc, (int)l0), hTm2 = wave_read_lane_f64(o.hTm, (int)l0);
                if (lane == lf) o.valid = 1u, o.id = id, o.cls = cls2, o.fc = nfc, o.fm = nfm, o.Tc = S.cls[cls2].Tc, o.Tm = S.cls[cls2].Tm, o.hTc = hTc2, o.hTm = hTm2;
                ++live;
              } else if (gpu_place) {
                ++st_gpu;
              } else {
                ++st_opendead;
              }
            }
            if (J.grouped) {  // the group's next members must not land on this offer: the table, and EVERY class wave answers the later steps again
              if (lane == 0) {
                const unsigned g0 = S.goff[J.grp], gn = S.gcnt[J.grp];
                S.gids[g0 + gn] = (uint16_t)id, S.gcnt[J.grp] = (uint16_t)(gn + 1u);
              }
              wave_sync();
              if (lane >= CF_OVL && (from_ov || lane != l0)) {
                ++o_ver, ++o_cmd;
                CfCmd* cm = &S.cmd[lane - CF_OVL + 1u];
                cm->kind = CFC_NONE, cm->pos = 0u, cm->nfc = 0u, cm->nfm = 0u, cm->lane_s = s, cm->seq = seq;
                COMPILER_FENCE();
                st_wg(&cm->ver, o_cmd);
              }
            }
            if (lane == 0) {
              vl->info = s | 1u << 8 | (from_ov ? 1u : 0u) << 9 | (opens ? 1u : 0u) << 10 | (from_ov ? 0u : l0 - CF_OVL + 1u) << 12 | (J.grouped ? 1u : 0u) << 16;
              vl->id = id, vl->ofc = ofc, vl->ofm = ofm, vl->nfc = nfc, vl->nfm = nfm;
              COMPILER_FENCE();
              st_wg(&vl->seq, seq);
            }
            if (opens && live >= CF_EPOCH_AT) {  // the overlay is full of live offers: back into their classes' arrays, every wave in lockstep
              md = seq << 4 | CFM_EPOCH;
              if (lane == 0) S.misc[CFX_EX_LANE] = s, S.misc[CFX_DRAIN] = seq, st_wg(&S.misc[CFX_MODE], md);
            }
          }
          CF_PROF_T(p3);
          CF_PROF_ADD(0, p1 - p0);
          CF_PROF_ADD(1, p2 - p1);
          CF_PROF_ADD(2, p3 - p2);
        }
      } else if (is_class_wave) {
        // ================================================= a class wave: answers ahead of the decider =================================================
        while (md == 0u) {
          CF_PROF_T(q0);
          const unsigned ver = cf_poll(&S.cmd[w].ver);
          if (ver != o_cmd) {  // a member of ours was taken (or the group table changed): obey, then answer the steps behind that one again
            COMPILER_FENCE();
            CF_TRACE("class wave %u: command %u\n", w, ver);
            const unsigned kind = cf_poll(&S.cmd[w].kind), pos = cf_poll(&S.cmd[w].pos), nfc = cf_poll(&S.cmd[w].nfc), nfm = cf_poll(&S.cmd[w].nfm), cs = cf_poll(&S.cmd[w].lane_s),
                           cseq = cf_poll(&S.cmd[w].seq);
            if (kind != CFC_NONE) {
              const unsigned ch = chunk_of(pos);
              const unsigned ofc = S.fc[pos], ofm = S.fm[pos];
              class_remove(pos, ch, kind == CFC_GPU_PLACE, ofc, ofm, nfc, nfm, cs, cseq);
            }
            o_cmd = ver, ++o_ver;
            todo = walkmask & ~((2ull << cs) - 1ull);
            if (lane == 0) st_wg(&S.ack[w], cseq);
            CF_TRACE("class wave %u: command %u obeyed (step %u)\n", w, ver, cseq);
            CF_PROF_T(q4);
            CF_PROF_ADD(3, q4 - q0);
            continue;
          }
          const unsigned mode = cf_poll(&S.misc[CFX_MODE]);
          if (mode != 0u) {
            if (cf_poll(&S.cmd[w].ver) != o_cmd) continue;  // (a command given before the mode was raised comes first)
            md = mode;
            break;
          }
          if (todo != 0ull && ((live_waves >> w) & 1u)) {
            const unsigned s = (unsigned)__ffsll(todo) - 1u;
            const unsigned seq = seq_of(s);
            if (seq < cf_poll(&S.misc[CFX_HEAD_SEQ]) + CF_BOARD) {
              CF_PROF_T(q1);
              const CfJobU J = cf_job_uniform(&S.ring[slot * 64u + s]);
              CfPost mine;
              cf_class_answer(S, J, lane, c, mine, st_scans);
              CF_PROF_T(q2);
              CF_TRACE("class wave %u: answer for step %u lane %u: fa %.17g offer %u\n", w, seq, s, mine.fa, mine.w0);
              if (lane == 0) {
                CfCand* e = &S.board[(seq & (CF_BOARD - 1u)) * CF_WAVES + w];
                const bool none = !(mine.fa > 0.0);
                const CfClass* cl = &S.cls[none ? 0u : mine.cls];
                e->pos = mine.pos, e->fc = mine.fc, e->fm = mine.fm, e->cid = mine.cls << 16 | (mine.w0 & CF_IDMASK), e->flags = (mine.w0 >> 31) | (none ? 2u : 0u);
                e->hTc = cl->hTc, e->hTm = cl->hTm;
                COMPILER_FENCE();
                st_wg(&e->tag, seq << 8 | (o_ver & 255u));
              }
              todo &= todo - 1ull;
              CF_PROF_T(q3);
              CF_PROF_ADD(0, q1 - q0);
              CF_PROF_ADD(1, q2 - q1);
              CF_PROF_ADD(2, q3 - q2);
              continue;
            }
          }
          if ((live_waves >> w) & 1u) SPIN_PAUSE_NEAR();
          else SPIN_PAUSE_IDLE();
        }
      } else {
        // ================================================= the bookkeeper: follows the verdict log =================================================
        while (md == 0u) {
          const unsigned nx = bk_done + 1u;
          const CfVlog* vl = &S.vlog[nx & (CF_VLOG - 1u)];
          if (cf_poll(&vl->seq) == nx) {
            COMPILER_FENCE();
            const unsigned info = cf_poll(&vl->info);
            const unsigned s = info & 255u;
            if ((info >> 8) & 1u) {
              const unsigned cw = (info >> 12) & 15u;
              if (cw != 0u) {  // a member left class wave cw: its summaries (and the level maxima) are up to date once it says so
                while (cf_poll(&S.ack[cw]) < nx) SPIN_PAUSE_NEAR();
              }
              const unsigned id = cf_poll(&vl->id), ofc = cf_poll(&vl->ofc), ofm = cf_poll(&vl->ofm), nfc = cf_poll(&vl->nfc), nfm = cf_poll(&vl->nfm);
              follow_tables();
              books_placement(s, id, ((info >> 9) & 1u) != 0u, ((info >> 10) & 1u) != 0u, ofc, ofm, nfc, nfm);
            }
            bk_done = nx;
            CF_TRACE("bookkeeper: step %u done (info %x)\n", nx, info);
            if (lane == 0) st_wg(&S.misc[CFX_BK_DONE], nx);
            continue;
          }
          const unsigned mode = cf_poll(&S.misc[CFX_MODE]);
          if (mode != 0u && bk_done >= cf_poll(&S.misc[CFX_DRAIN])) {
            md = mode;
            break;
          }
          SPIN_PAUSE_IDLE();
        }
      }
      // ================================================= a collective turn: every wave =================================================
      CF_TRACE("wave %u: to the collective turn %x\n", w, md);
      EMU_SITE("classfit: collective");
      __syncthreads();
      md = wave_uniform_u32(md);  // (every wave left its loop with the mode word the decider raised)
      const unsigned kind = md & 15u, cseq = md >> 4;
      bool epoch = kind == CFM_EPOCH;
      unsigned s = S.misc[CFX_EX_LANE];
      if (kind == CFM_EXACT) {
        ++st_exact;
        const CfJobU J = cf_job_uniform(&S.ring[slot * 64u + s]);
        const double fmax = __longlong_as_double((long long)((unsigned long long)S.misc[CFX_FMAX_LO] | (unsigned long long)S.misc[CFX_FMAX_HI] << 32));
        CfPost mine;
        if (w == 0) {
          CfOvLane ov = o;  // (lanes 58..63 hold this step's candidates: the class waves answer for their members themselves)
          if (lane >= CF_OVL) ov.valid = 0u;
          cf_overlay_query<true>(S, J, lane, ov, fmax, sc, sm, mine);
          if (lane == 0) S.post2[0] = mine;
        } else if (is_class_wave) {
          cf_class_query<true>(S, J, lane, c, fmax, sc, sm, mine, st_scans);
          if (lane == 0) S.post2[w] = mine;
        }
        EMU_SITE("classfit: exact turn");
        __syncthreads();
        const CfVerdict v = cf_verdict<true>(S.post2, lane);
        // (an exact turn is raised because candidates exist: v.src >= 0)
        const unsigned nfc = v.fc - J.c, nfm = v.fm - J.m;
        const bool dead = nfc < cmin || nfm < mmin;
        const bool from_ov = v.src == 0;
        const bool gpu_place = !from_ov && J.kind != 0u;
        const bool opens = !from_ov && !gpu_place && !dead;
        ++matched;
        if (base + s == 0u) head = 1u;
        if (w == 0) {
          unsigned live = (unsigned)__popcll(__ballot(lane < CF_OVL && o.valid != 0u));
          if (from_ov) {
            ++st_ovwin;
            if (lane == v.pos) {
              o.fc = nfc, o.fm = nfm;
              if (dead) o.valid = 0u;
            }
            if (dead) ++st_dead, --live;
          } else {
            if (lane == CF_OVL - 1u + (unsigned)v.src) ++o_ver;  // (the class wave takes its member out itself, below; no command)
            if (opens) {
              ++st_open;
              const unsigned lf = (unsigned)__ffsll(~__ballot(o.valid != 0u || lane >= CF_OVL)) - 1u;
              const CfClass cl = S.cls[v.cls];
              if (lane == lf) o.valid = 1u, o.id = v.id, o.cls = v.cls, o.fc = nfc, o.fm = nfm, o.Tc = cl.Tc, o.Tm = cl.Tm, o.hTc = cl.hTc, o.hTm = cl.hTm;
              ++live;
            } else if (gpu_place) {
              ++st_gpu;
            } else {
              ++st_opendead;
            }
          }
          if (J.grouped) {
            if (lane == 0) {
              const unsigned g0 = S.goff[J.grp], gn = S.gcnt[J.grp];
              S.gids[g0 + gn] = (uint16_t)v.id, S.gcnt[J.grp] = (uint16_t)(gn + 1u);
            }
            if (lane >= CF_OVL && (from_ov || lane != CF_OVL - 1u + (unsigned)v.src)) ++o_ver;
          }
          todo &= ~(1ull << s);
          epoch = opens && live >= CF_EPOCH_AT;
          if (lane == 0) S.misc[7] = epoch ? 1u : 0u;
        } else if (is_class_wave) {
          const bool mine_src = (int)w == v.src;
          if (mine_src) class_remove(v.pos, v.aux, gpu_place, v.fc, v.fm, nfc, nfm, s, cseq);
          if (mine_src || J.grouped) ++o_ver, todo = walkmask & ~((2ull << s) - 1ull);
          if (lane == 0) st_wg(&S.ack[w], cseq);
        } else {
          follow_tables();
          books_placement(s, v.id, from_ov, opens, v.fc, v.fm, nfc, nfm);
          bk_done = cseq;
          if (lane == 0) st_wg(&S.misc[CFX_BK_DONE], cseq);
        }
        EMU_SITE("classfit: exact turn done");
        __syncthreads();
        epoch = S.misc[7] != 0u;
        if (is_books) follow_tables();  // (a change of level maxima made in this turn)
      }
      if (epoch) {  // ---- the overlay is full of live offers: back into their classes' arrays
        const unsigned long long te = cook_ticks();
        ++st_epochs;
        // (1) the overlay's lanes, sorted by (class, E, offer), into LDS
        if (w == 0) {
          const bool live = lane < CF_OVL && o.valid != 0u;
          const unsigned long long key = live ? ((unsigned long long)o.cls << 58 | ((unsigned long long)o.fc * o.Tm + (unsigned long long)o.fm * o.Tc) << 13 | (unsigned long long)o.id) : ~0ull;
          unsigned rank = 0;
          for (unsigned l = 0; l < 64u; ++l) rank += wave_read_lane_u64(key, (int)l) < key ? 1u : 0u;
          if (live) S.ovl[3u * rank] = o.cls << 16 | o.id, S.ovl[3u * rank + 1u] = o.fc, S.ovl[3u * rank + 2u] = o.fm;
          const unsigned nlive = (unsigned)__popcll(__ballot(live));
          if (lane == 0) S.misc[3] = nlive;
          o.valid = 0u;
        }
        for (unsigned x = tid; x < 3u * CF_MAXCLS; x += CF_THREADS) S.ckept[x] = 0u;
        EMU_SITE("classfit: epoch 1");
        __syncthreads();
        // (2) members kept / inserted per class
        if (is_class_wave && c.cls != 0xFFu) atomicAdd(&S.ckept[c.cls], (unsigned)__popcll(c.pres));
        if (w == 0 && lane < S.misc[3]) atomicAdd(&S.ckept[CF_MAXCLS + (S.ovl[3u * lane] >> 16)], 1u);
        __syncthreads();
        if (tid == 0) {
          unsigned off = 0;
          for (unsigned ci = 0; ci < n_cls; ++ci) S.ckept[2 * CF_MAXCLS + ci] = off, off += S.ckept[ci] + S.ckept[CF_MAXCLS + ci];
        }
        __syncthreads();
        // (3) every class wave merges its classes into the scratch arrays: kept members keep their order, the list's entries go between them
        if (is_class_wave) {
          unsigned li = 0;  // first list entry of the class being merged
          for (unsigned ci = 0; ci < n_cls; ++ci) {
            const unsigned ni = S.ckept[CF_MAXCLS + ci];
            if (S.cls[ci].wave == w) {
              const unsigned noff = S.ckept[2 * CF_MAXCLS + ci];
              const unsigned Tc = S.cls[ci].Tc, Tm = S.cls[ci].Tm;
              unsigned kept_before = 0, ip = li;
              const unsigned long long chunks = __ballot(c.cls == ci);
              for (unsigned long long mm = chunks; mm; mm &= mm - 1ull) {
                const unsigned ch = (unsigned)__ffsll(mm) - 1u;
                const unsigned pos0 = (unsigned)wave_read_lane((int)c.pos0, (int)ch), n = (unsigned)wave_read_lane((int)c.n, (int)ch);
                const unsigned long long pres = wave_read_lane_u64(c.pres, (int)ch);
                const bool in = lane < n;
                const bool keep = in && ((pres >> lane) & 1ull);
                const uint32_t fc = S.fc[pos0 + lane], fm = S.fm[pos0 + lane], cid = ci << 16 | (uint32_t)S.cid[pos0 + lane];
                const unsigned long long E = (unsigned long long)fc * Tm + (unsigned long long)fm * Tc;
                const unsigned idq = cid & CF_IDMASK;
                const unsigned long long keepm = __ballot(keep);
                unsigned ins_before = ip - li;  // list entries of the class in front of this member
                // entries whose key is below the chunk's last member go in here (members that left keep their old key: the order stands)
                const unsigned long long Elast = wave_read_lane_u64(E, (int)(n - 1u));
                const unsigned idlast = (unsigned)wave_read_lane((int)idq, (int)(n - 1u));
                while (ip < li + ni) {
                  const unsigned ecid = S.ovl[3u * ip], efc = S.ovl[3u * ip + 1u], efm = S.ovl[3u * ip + 2u];
                  const unsigned long long Ee = (unsigned long long)efc * Tm + (unsigned long long)efm * Tc;
                  const unsigned ide = ecid & CF_IDMASK;
                  if (!(Ee < Elast || (Ee == Elast && ide < idlast))) break;
                  const bool before = in && (E < Ee || (E == Ee && idq < ide));  // the member stays in front of the entry
                  const unsigned long long bm = __ballot(before);
                  if (in && !before) ++ins_before;
                  const unsigned np = noff + kept_before + (unsigned)__popcll(keepm & bm) + (ip - li);
                  if (lane == 0) st_agent(&b.scr_fc[np], efc), st_agent(&b.scr_fm[np], efm), st_agent(&b.scr_cid[np], ecid);
                  ++ip;
                }
                if (keep) {
                  const unsigned np = noff + kept_before + (unsigned)__popcll(keepm & lanemask_lt()) + ins_before;
                  st_agent(&b.scr_fc[np], fc), st_agent(&b.scr_fm[np], fm), st_agent(&b.scr_cid[np], cid);
                }
                kept_before += (unsigned)__popcll(keepm);
              }
              for (; ip < li + ni; ++ip) {  // entries behind the class's last member
                const unsigned np = noff + kept_before + (ip - li);
                if (lane == 0) st_agent(&b.scr_fc[np], S.ovl[3u * ip + 1u]), st_agent(&b.scr_fm[np], S.ovl[3u * ip + 2u]), st_agent(&b.scr_cid[np], S.ovl[3u * ip]);
              }
            }
            li += ni;
          }
        }
        drain_stores();
        EMU_SITE("classfit: epoch 3");
        __syncthreads();
        // (4) the merged arrays back into LDS, the class table
        unsigned newM = 0;
        for (unsigned ci = 0; ci < n_cls; ++ci) newM += S.ckept[ci] + S.ckept[CF_MAXCLS + ci];
        for (unsigned q = tid; q < NP; q += CF_THREADS) {
          const bool inq = q < newM;
          const uint32_t fcq = inq ? ld_agent(&b.scr_fc[q]) : 0u, fmq = inq ? ld_agent(&b.scr_fm[q]) : 0u, cq = inq ? ld_agent(&b.scr_cid[q]) : 0xFFFFFFFFu;
          bool tie = false;  // the next member of the class inside the guard band of this one
          if (q + 1u < newM) {
            const uint32_t fcn = ld_agent(&b.scr_fc[q + 1u]), fmn = ld_agent(&b.scr_fm[q + 1u]), cn = ld_agent(&b.scr_cid[q + 1u]);
            if ((cn >> 16) == (cq >> 16)) {
              const CfClass* cl = &S.cls[cq >> 16];
              tie = (unsigned long long)fcn * cl->Tm + (unsigned long long)fmn * cl->Tc <= (unsigned long long)fcq * cl->Tm + (unsigned long long)fmq * cl->Tc + cl->dE;
            }
          }
          S.fc[q] = fcq, S.fm[q] = fmq, S.cid[q] = inq ? (uint16_t)((cq & (CF_OCC | CF_IDMASK)) | (tie ? CF_TIE : 0u)) : (uint16_t)0xFFFFu;
        }
        if (tid < n_cls) S.cls[tid].n = S.ckept[tid] + S.ckept[CF_MAXCLS + tid], S.cls[tid].off = S.ckept[2 * CF_MAXCLS + tid];
        EMU_SITE("classfit: epoch 4");
        __syncthreads();
        // (5) lanes, summaries, tables, books; every candidate on the board is void: a new version everywhere
        if (is_class_wave) {
          cf_setup_chunks(S.cls, n_cls, w, lane, c, nch_wave);
          for (unsigned ch = 0; ch < nch_wave; ++ch) cf_tighten(S, t, lane, ch, c);
          cf_wave_tables(S, w, lane, c, w == gpu_wave, n_kind);
          if (w == 1u && lane == 0) S.misc[6] = s, S.misc[2] = cseq + 0x40000000u;  // (a sequence number no step's change uses)
          ++o_ver;
          todo = walkmask & ~((2ull << s) - 1ull);
          retable_seq = 0u;
        }
        if (w == 0 && lane >= CF_OVL) ++o_ver;
        if (is_books && lane > s) bk_cnt = 0u;
        EMU_SITE("classfit: epoch 5");
        __syncthreads();
        if (is_books) follow_tables();
        tk_epoch += cook_ticks() - te;
      }
      if (kind == CFM_BATCH_END) batch_done = true;
      if (tid == 0) st_wg(&S.misc[CFX_MODE], 0u);
      EMU_SITE("classfit: collective done");
      __syncthreads();
      CF_TRACE("wave %u: collective turn done, batch_done %d\n", w, (int)batch_done);
    }
    seq_base += nw;
    if (w == 0) {  // the overlay's free values for the next pre-check
      const bool live = lane < CF_OVL && o.valid != 0u;
      S.ovm[2u * lane] = live ? o.fc : 0u, S.ovm[2u * lane + 1u] = live ? o.fm : 0u;
      const unsigned long long vm = __ballot(live);
      if (lane == 0) S.misc[4] = (unsigned)vm, S.misc[5] = (unsigned)(vm >> 32);
    }
    // ---- batch end: results out, the next batch's jobs in
    if (is_books) {
      follow_tables();
      if (lane < bn) {
        const unsigned k = base + lane;
        st.job_to_offer[k] = bk_res;
        unsigned fail = 0u;
        if (bk_res < 0) {
          // failure code as match_serial's: 1 = an offer lacks room, 2 = an offer with room refuses on a constraint (every offer with room does:
          // the job stayed unmatched), 8 = no offer at all
          fail = (bk_b1 ? 1u : 0u) | ((bk_cha || bk_cnt != 0u) ? 2u : 0u);
          if (fail == 0u) fail = 8u;
        }
        if (st.fail_code) st.fail_code[k] = fail;
      }
      if (base + 64u + lane < K) S.ring[(slot ^ 1u) * 64u + lane] = nxt;
    }
    EMU_SITE("classfit: batch end");
    __syncthreads();
    CF_TRACE("wave %u: batch at %u ended\n", w, base);
  }
  if (w == 0 && lane == 0) {
    st.summary[0] = matched;
    st.summary[1] = (matched == 0u || head) ? 1u : 0u;
    st.summary[2] = st_epochs;
    const unsigned long long t_end = cook_ticks();
    uint32_t* sx = ctl->stats;
    sx[CFS_MATCHED] = matched, sx[CFS_OV_WIN] = st_ovwin, sx[CFS_OPEN] = st_open, sx[CFS_OPEN_DEAD] = st_opendead, sx[CFS_GPU_PLACE] = st_gpu, sx[CFS_EPOCHS] = st_epochs,
    sx[CFS_EXACT] = st_exact, sx[CFS_WALKED] = st_walked, sx[CFS_DEAD_DROP] = st_dead, sx[CFS_BATCHES] = (K + 63u) / 64u, sx[CFS_PRESETTLED] = K - st_walked;
    sx[CFS_TICKS_TOTAL] = (uint32_t)(t_end - t_start), sx[CFS_TICKS_PROLOGUE] = (uint32_t)(t_loop - t_start), sx[CFS_TICKS_EPOCH] = (uint32_t)tk_epoch;
  }
  if (w == 0) {
    const unsigned sp = wave_max_u32(lane >= CF_OVL ? st_spins : 0u);
    if (lane == 0) ctl->stats[CFS_SPINS] = sp;
  }
  if (is_class_wave && lane == 0) atomicAdd(&ctl->stats[CFS_SCANS], st_scans), atomicAdd(&ctl->stats[CFS_TIGHTEN], st_tight);
  if (is_books && lane == 0) ctl->stats[CFS_TICKS_PRECHECK] = (uint32_t)tk_pre;
  (void)tk_wait;
#ifdef CF_PROF
  if (lane == 0 && w <= 2u)
    for (int i = 0; i < 4; ++i) ctl->stats[20 + 4 * w + i] = prof[i];
#endif
}

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

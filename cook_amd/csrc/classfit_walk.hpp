// classfit_walk.hpp — the walk of the class-ordered best fit (classfit.hpp §4), second form.
//
// One workgroup of 8 waves per pool:
//   wave 0         the DECIDER: lanes 0..57 hold the overlay (the offers this call has placed on, state in registers), lanes 58..63 take, for the job of
//                  the step, the candidate each class wave has published on the board.  One evaluation of the 64 lanes, one wave maximum, the commit
//                  in registers.  Nothing in a step waits for another wave unless a candidate is missing.
//   waves 1-3,5-7  CLASS waves (logical 1..6): lane = chunk of 64 positions of the wave's classes, level summaries in registers.  They answer the walked
//                  jobs of the batch AHEAD of the decider (up to CF_BOARD steps): per relevant class the first feasible member (the arrays are sorted:
//                  the class's best), of several classes the one of greatest approximate fitness.  Nobody tells them about a placement: the decider
//                  ZEROES the member it takes in LDS and counts the wave's removals; an answer carries the count it was computed under, and the decider
//                  takes an older one as long as it does not name a member removed since (the first feasible member of a sorted array stays the
//                  first when another one leaves).  A wave that sees its count move answers the steps in flight again.
//   wave 4         the BOOKKEEPER (shares SIMD 0 with the decider and sleeps through the walk): between batches it settles who must be visited at
//                  all (level maxima over every chunk: placements only take room away) and writes the failure codes.
// Steps that need every wave — several candidates inside the guard band (the literal fitness decides), the end of an epoch (the overlay's live lanes
// go back into their classes' arrays), the end of a batch of 64 jobs — are COLLECTIVE turns behind a mode word and barriers.
// Exactness of the failure codes (cook_match_explain reads them): the level maxima are exact at every batch boundary (the class waves apply the batch's
// removals from the decider's log there); a job without a place has room somewhere at its turn iff it has at the batch's end, or had at its start and
// some LATER placement of the batch found room for it in the offer it took (the log's old values).
#pragma once
#if defined(CF_PROF) && !defined(CF_STATS)
#define CF_STATS 1
#endif
#ifdef CF_STATS  // counters and phase timers of the walk's hot paths (a study build: each costs scalar registers the walk is short of)
#define CF_STAT(x) x
#define CF_TICKS() cook_ticks()
#else
#define CF_STAT(x) ((void)0)
#define CF_TICKS() 0ull
#endif
#ifdef CF_PROF  // timing-study build: shader cycles (s_memtime) per phase of the decider's steps and of class wave 1's answers, to CfCtl::stats[24..]
#define CF_PROF_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define CF_PROF_ADD(i, d) prof[i] += (unsigned long long)(d)
#else
#define CF_PROF_T(x) ((void)0)
#define CF_PROF_ADD(i, d) ((void)0)
#endif

#ifndef CF_AHEAD
#define CF_AHEAD 10  // (measured: a C4 pool 50.4 ms at 4, 49.1 at 6, 48.1 at 8; later 44.05 at 8, 43.5 at 10; with 24 rows 43.45 / 43.41 / 43.98 at 12 / 16 / 22: the decider is the limit)
#endif
constexpr unsigned CF_BOARD = CF_AHEAD;  // steps the class waves may run ahead of the decider (at most CF_SLOTS - 2)
#ifndef CF_SLOTS_N
#define CF_SLOTS_N 12
#endif
constexpr unsigned CF_SLOTS = CF_SLOTS_N;  // entries per column of the board: walked ordinal modulo 12 (24).  The waves that share a set's jobs take the ordinals in turn and
                                           // their number divides it, so a slot has ONE writer: a late answer of a slow wave lands where only that wave's next one goes
static_assert(CF_BOARD + 2u <= CF_SLOTS, "a slot is free again before its wave writes it two rows on");
static_assert(CF_SLOTS == 12u || CF_SLOTS == 24u, "cf_slot and classfit_asm.hpp know these two");
static __device__ __forceinline__ unsigned cf_slot(unsigned ord) { return ord - CF_SLOTS * ((ord * 43u) >> (CF_SLOTS == 12u ? 9 : 10)); }  // ord % CF_SLOTS for ord < 64
constexpr unsigned CF_OVL = 58;     // overlay lanes (lanes 58..63 are the candidates of logical class waves 1..6)
constexpr unsigned CF_EPOCH_AT = COOK_SHAPE(58, 8);  // live overlay lanes that end an epoch
constexpr unsigned CFW_BOOKS = 4;   // the bookkeeper's wave
enum : unsigned { CFM_EXACT = 1u, CFM_EPOCH = 2u, CFM_BATCH_END = 3u };
enum : unsigned { CFX_SPARE0 = 0, CFX_SPARE1, CFX_LOGN, CFX_LOG_APPLIED, CFX_WALK_LO, CFX_WALK_HI, CFX_EX_LANE, CFX_FMAX_LO, CFX_FMAX_HI, CFX_EPOCH, CFX_MATCH_LO, CFX_MATCH_HI,
                  CFX_B1_LO, CFX_B1_HI, CFX_OVN, CFX_MINFC, CFX_MINFM, CFX_N = 24 };  // words of CfLds::misc
constexpr uint32_t CF_ENT_NONE = 0x80000000u, CF_ENT_AMB = 0x40000000u;  // CfEnt::cid: no candidate / another member or class may round to the same fitness

struct __attribute__((aligned(8))) CfFree {  // free cpus / mem of a position (fixed point), one 8-byte LDS access
  uint32_t c, m;
};
struct CfEnt {  // a class wave's answer for one step (32 B).  tag = job << 12 | generation << 8 | the wave's removal count (mod 256) the answer knows;
                // the writer voids the tag, stores the fields, stores the tag; the reader reads tag, fields, tag
  uint32_t tag, pad, pos, cid;  // pos: position | chunk lane << 16; cid: offer | class << 16 | CF_ENT_*
  uint32_t fc, fm;
  double fa;                    // approximate fitness
};
struct CfLog {  // a placement of the batch (32 B)
  uint32_t info, pos;           // info: batch lane | logical class wave << 8 (0: an overlay lane won) | chunk lane << 12 | gpu placement << 20
  uint32_t ofc, ofm, nfc, nfm, pad0, pad1;
};
struct CfPost {  // what a wave says in an exact turn (32 B)
  double fa;     // literal fitness of its best candidate, 0 = none
  uint32_t w0;   // offer
  uint32_t pos;  // class waves: position; overlay: lane
  uint32_t fc, fm, cls, aux;  // aux: class waves: chunk lane
};
struct CfJobU {  // the job of a step, wave-uniform
  unsigned c, m, kind, L, n_eq, n_nov, grouped, grp, eq0, eq1, nov0, nov1;
};
struct CfLds {  // the workgroup's LDS, carved at run time
  CfFree* fcm;           // [NP] free cpus / mem of a position (0 / 0: the member has left, or padding)
  uint16_t* cid;        // [NP] occupied gpu host << 15 | the next member may round to the same fitness << 14 | offer
  CfEnt* board;         // [CF_SLOTS][8]
  CfLog* log;           // [64]
  uint32_t* ctrl;       // [8] [0] the mode word of a collective turn, [1] the batch lane of the decider's step, [1 + w] removals from logical class wave w so far
  uint64_t* attr8;
  uint16_t *goff, *gcnt, *gids;
  CfJob* ring;          // [2][64]
  CfPost* post2;        // [8] exact turns
  CfClass* cls;         // [CF_MAXCLS] the class table (n / off as of the last epoch)
  uint32_t* pw;         // [8][CF_LV] greatest level summaries of a logical wave's chunks of hosts without gpus
  uint32_t* aw;         // [8][CF_LV] ... of all its chunks, occupied gpu hosts included
  uint32_t* gk;         // [CF_MAXKIND][CF_LV] ... of a gpu kind's chunks
  uint32_t* ovt;        // [CF_LV] ... of the overlay's lanes, as of the last batch end
  uint32_t* ovl;        // [64][3] an epoch's overlay list (cid, fc, fm), sorted
  uint32_t* ckept;      // [CF_MAXCLS] kept members / [CF_MAXCLS] inserted / [CF_MAXCLS] new offsets
  uint32_t* misc;       // CFX_*
  const uint32_t* t;    // [CF_LV] the call's cpus levels
  const uint32_t* envw; // CFE_*: the call's constants
};

static __device__ __forceinline__ unsigned cf_level_of(const uint32_t* t, uint32_t fc) {  // (t: the call's levels, in LDS)
   // how many of the levels fc reaches (t ascending)
  unsigned n = 0;
#pragma unroll
  for (int i = 0; i < CF_LV; ++i) n += fc >= t[i] ? 1u : 0u;
  return n;
}
// eight level values in eight REGISTERS (as an array inside a structure, or as a vector type indexed by a variable, the compiler keeps them in scratch
// memory: a store of the whole set and an indexed load per access, seen in the ISA)
struct CfLv8 {
  uint32_t v0, v1, v2, v3, v4, v5, v6, v7;
};
#define CF_FOR8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
static __device__ __forceinline__ uint32_t cf_lv_get(const CfLv8& a, unsigned i) {  // a wave-uniform index: a tree of selects on its three bits
  const bool b0 = (i & 1u) != 0u, b1 = (i & 2u) != 0u, b2 = (i & 4u) != 0u;
  const uint32_t r01 = b0 ? a.v1 : a.v0, r23 = b0 ? a.v3 : a.v2, r45 = b0 ? a.v5 : a.v4, r67 = b0 ? a.v7 : a.v6;
  const uint32_t r03 = b1 ? r23 : r01, r47 = b1 ? r67 : r45;
  return b2 ? r47 : r03;
}
struct CfChunkLane {  // a class wave's lane = one chunk
  unsigned cls, kind, pos0, n, Tc, Tm;
  unsigned long long dE;
  double hTc, hTm;
  CfLv8 lv, la;  // lv: members that can take a job (not an occupied gpu host); la: all members
};

// lanes of a class wave <- the chunks of the wave's classes, in class order
static __device__ __forceinline__ void cf_setup_chunks(const CfClass* cls, unsigned nc, unsigned lw, unsigned lane, CfChunkLane& c, unsigned& nch_wave) {
  c.cls = 0xFFu, c.kind = 0xFEu, c.pos0 = 0u, c.n = 0u, c.Tc = 1u, c.Tm = 1u, c.dE = 0ull, c.hTc = 0.0, c.hTm = 0.0;
  unsigned acc = 0;
  for (unsigned ci = 0; ci < nc; ++ci) {
    const CfClass& cl = cls[ci];
    if (cl.wave != lw) continue;
    const unsigned nch = (cl.n + 63u) / 64u;
    if (lane >= acc && lane < acc + nch) {
      const unsigned x = lane - acc;
      c.cls = ci, c.kind = cl.kind, c.pos0 = cl.off + 64u * x, c.n = cf_min(64u, cl.n - 64u * x), c.Tc = cl.Tc, c.Tm = cl.Tm, c.dE = cl.dE, c.hTc = cl.hTc, c.hTm = cl.hTm;
    }
    acc += nch;
  }
  nch_wave = acc;
}

// the level summaries of chunk `ch` (wave-uniform) from its members; lanes = positions
static __device__ __forceinline__ void cf_tighten(const CfLds& S, unsigned lane, unsigned ch, CfChunkLane& c) {
  const uint32_t* t = S.t;
  const unsigned pos0 = (unsigned)wave_read_lane((int)c.pos0, (int)ch), n = (unsigned)wave_read_lane((int)c.n, (int)ch);
  const bool in = lane < n;
  const CfFree f = S.fcm[pos0 + lane];
  const uint32_t cid = S.cid[pos0 + lane];
  const unsigned nl = in ? cf_level_of(t, f.c) : 0u;
  const bool free_host = !(cid & CF_OCC);
#define CF_TI_A(i) unsigned a##i = nl > (unsigned)i ? f.m + 1u : 0u;
#define CF_TI_R(i) unsigned r##i = (nl > (unsigned)i && free_host) ? f.m + 1u : 0u;
  CF_FOR8(CF_TI_A)
  CF_FOR8(CF_TI_R)
#undef CF_TI_A
#undef CF_TI_R
  wave_max8_u32(a0, a1, a2, a3, a4, a5, a6, a7);
  wave_max8_u32(r0, r1, r2, r3, r4, r5, r6, r7);
#define CF_TI_S(i) \
  if (lane == ch) c.la.v##i = a##i, c.lv.v##i = r##i;
  CF_FOR8(CF_TI_S)
#undef CF_TI_S
}
// the wave's rows of the level-maxima tables
static __device__ __forceinline__ void cf_wave_tables(const CfLds& S, unsigned lw, unsigned lane, const CfChunkLane& c, unsigned kinds, unsigned n_kind) {
#define CF_TB_P(i) unsigned p##i = c.kind == 0u ? c.lv.v##i : 0u;
#define CF_TB_A(i) unsigned a##i = c.kind < 0xFEu ? c.la.v##i : 0u;
  CF_FOR8(CF_TB_P)
  CF_FOR8(CF_TB_A)
#undef CF_TB_P
#undef CF_TB_A
  wave_max8_u32(p0, p1, p2, p3, p4, p5, p6, p7);
  wave_max8_u32(a0, a1, a2, a3, a4, a5, a6, a7);
#define CF_TB_S(i) \
  if (lane == 0) S.pw[lw * CF_LV + i] = p##i, S.aw[lw * CF_LV + i] = a##i;
  CF_FOR8(CF_TB_S)
#undef CF_TB_S
  for (unsigned k = 1; k < n_kind; ++k) {
    if (!((kinds >> k) & 1u)) continue;
#define CF_TB_G(i) unsigned g##i = c.kind == k ? c.lv.v##i : 0u;
    CF_FOR8(CF_TB_G)
#undef CF_TB_G
    wave_max8_u32(g0, g1, g2, g3, g4, g5, g6, g7);
#define CF_TB_GS(i) \
  if (lane == 0) S.gk[k * CF_LV + i] = g##i;
    CF_FOR8(CF_TB_GS)
#undef CF_TB_GS
  }
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
    const unsigned g0 = S.goff[J.grp], gn = ld_wg(&S.gcnt[J.grp]);
    for (unsigned x = 0; x < gn; ++x) ok = ok && id != (unsigned)S.gids[g0 + x];
  }
  return ok;
}
static __device__ __forceinline__ double cf_literal(unsigned Tc, unsigned Tm, unsigned fc, unsigned fm, unsigned jc, unsigned jm, double sc, double sm) {
  // the literal expression of the placement (match_kernels.hpp fitness_of; the reference: cpuMemBinPacker, config.clj:108) on the exact values the fixed-point numbers stand for:
  // running + assigned = total - free, lease + running = total
  const double A = (double)(Tc - fc) * sc, Bm = (double)(Tm - fm) * sm, c = (double)jc * sc, m = (double)jm * sm;
  return ((A + c) / ((double)Tc * sc) + (Bm + m) / ((double)Tm * sm)) / 2.0;
}

// A class wave's answer for the decider: per relevant class the FIRST feasible member of the first chunk that can hold one (sorted by E: the class's
// best); of several classes of the wave the one of greatest approximate fitness.  amb: another member / class may round to the same fitness.
// A chunk whose summary promised room and that holds none gets its summaries recomputed on the spot (they are upper bounds between batches).
struct CfAns {
  bool have, amb;
  unsigned pos, ch, cid, fc, fm, cls;
  double fa;
};
static __device__ __forceinline__ void cf_class_answer(const CfLds& S, const CfJobU& J, unsigned lane, CfChunkLane& c, CfAns& out, unsigned& scans,
                                                       unsigned& tightened) {
  out.have = false, out.amb = false, out.pos = 0u, out.ch = 0u, out.cid = 0u, out.fc = 0u, out.fm = 0u, out.cls = 0u, out.fa = 0.0;
  const uint32_t lvL = cf_lv_get(c.lv, J.L);
  unsigned long long m = __ballot(c.kind == J.kind && lvL > J.m);
  while (m) {
    const unsigned ch = (unsigned)__ffsll(m) - 1u;
    ++scans;
    const unsigned pos0 = (unsigned)wave_read_lane((int)c.pos0, (int)ch), n = (unsigned)wave_read_lane((int)c.n, (int)ch);
    const unsigned q = pos0 + lane;
    const CfFree f = S.fcm[q];
    const uint32_t cid = S.cid[q];
    const bool room = lane < n && f.c >= J.c && f.m >= J.m && !(cid & CF_OCC);
    const bool ok = room && cf_cons_ok(S, J, cid & CF_IDMASK, room);
    const unsigned long long b = __ballot(ok);
    if (b == 0ull) {
      if (__ballot(room) == 0ull) {  // a stale summary (members have left since): exact again
        cf_tighten(S, lane, ch, c);
        ++tightened;
      }
      m &= ~(1ull << ch);
      continue;
    }
    const unsigned q0 = (unsigned)__ffsll(b) - 1u;
    const unsigned fc0 = (unsigned)wave_read_lane((int)f.c, (int)q0), fm0 = (unsigned)wave_read_lane((int)f.m, (int)q0), cid0 = (unsigned)wave_read_lane((int)cid, (int)q0);
    const unsigned cls = (unsigned)wave_read_lane((int)c.cls, (int)ch);
    m &= ~__ballot(c.cls == cls);  // the class is answered
    const bool tie = (cid0 & CF_TIE) != 0u;
    const double hTc = wave_read_lane_f64(c.hTc, (int)ch), hTm = wave_read_lane_f64(c.hTm, (int)ch);
    const double f0 = 1.0 - ((double)(fc0 - J.c) * hTc + (double)(fm0 - J.m) * hTm);
    if (!out.have || f0 > out.fa + CF_BAND) out.amb = tie;
    else if (f0 >= out.fa - CF_BAND) out.amb = true;
    if (!out.have || f0 > out.fa) out.fa = f0, out.cid = cid0 & CF_IDMASK, out.pos = pos0 + q0, out.fc = fc0, out.fm = fm0, out.cls = cls, out.ch = ch;
    out.have = true;
  }
}
// An exact turn's query: the literal fitness of every feasible member within the band below fmax; the greatest, lowest offer on ties.
static __device__ __forceinline__ void cf_class_query_exact(const CfLds& S, const CfJobU& J, unsigned lane, const CfChunkLane& c, double fmax, double sc, double sm, CfPost& out,
                                                            unsigned& scans) {
  out.fa = 0.0, out.w0 = 0u, out.pos = 0u, out.fc = 0u, out.fm = 0u, out.cls = 0u, out.aux = 0u;
  const uint32_t lvL = cf_lv_get(c.lv, J.L);
  unsigned long long m = __ballot(c.kind == J.kind && lvL > J.m);
  unsigned long long best_lit = 0ull;
  while (m) {
    const unsigned ch = (unsigned)__ffsll(m) - 1u;
    ++scans;
    const unsigned pos0 = (unsigned)wave_read_lane((int)c.pos0, (int)ch), n = (unsigned)wave_read_lane((int)c.n, (int)ch);
    const unsigned Tc = (unsigned)wave_read_lane((int)c.Tc, (int)ch), Tm = (unsigned)wave_read_lane((int)c.Tm, (int)ch), cls = (unsigned)wave_read_lane((int)c.cls, (int)ch);
    const double hTc = wave_read_lane_f64(c.hTc, (int)ch), hTm = wave_read_lane_f64(c.hTm, (int)ch);
    const unsigned q = pos0 + lane;
    const CfFree f = S.fcm[q];
    const uint32_t cid = S.cid[q];
    const bool room = lane < n && f.c >= J.c && f.m >= J.m && !(cid & CF_OCC);
    const bool ok = room && cf_cons_ok(S, J, cid & CF_IDMASK, room);
    const double fa = ok ? 1.0 - ((double)(f.c - J.c) * hTc + (double)(f.m - J.m) * hTm) : 0.0;
    const bool cand = ok && fa >= fmax - CF_BAND;
    const double lit = cand ? cf_literal(Tc, Tm, f.c, f.m, J.c, J.m, sc, sm) : 0.0;
    const unsigned long long lb = (unsigned long long)__double_as_longlong(lit);
    const unsigned long long mx = wave_max_u64(lb);
    if (mx != 0ull) {
      const unsigned idmin = ~wave_max_u32((cand && lb == mx) ? ~(cid & CF_IDMASK) : 0u);
      if (mx > best_lit || (mx == best_lit && idmin < out.w0)) {
        best_lit = mx;
        const unsigned q0 = (unsigned)__ffsll(__ballot(cand && lb == mx && (cid & CF_IDMASK) == idmin)) - 1u;
        out.fa = __longlong_as_double((long long)mx), out.w0 = idmin, out.pos = pos0 + q0, out.fc = (unsigned)wave_read_lane((int)f.c, (int)q0),
        out.fm = (unsigned)wave_read_lane((int)f.m, (int)q0), out.cls = cls, out.aux = ch;
      }
    }
    m &= ~(1ull << ch);
  }
}

struct CfOvLane {  // the decider's lane: an offer this call has placed on (lanes 0..57) / the step's candidate of a class wave (58..63)
  unsigned valid, id, cls, fc, fm;
  double hTc, hTm;
};
static __device__ __forceinline__ void cf_overlay_query_exact(const CfLds& S, const CfJobU& J, unsigned lane, const CfOvLane& o, double fmax, double sc, double sm, CfPost& out) {
  out.fa = 0.0, out.w0 = 0u, out.pos = 0u, out.fc = 0u, out.fm = 0u, out.cls = 0u, out.aux = 0u;
  if (J.kind != 0u) return;  // (the overlay holds hosts without gpus only: gpu hosts take one job and stay in their chunk)
  const bool isov = lane < CF_OVL;
  const bool room = isov && o.valid && o.fc >= J.c && o.fm >= J.m;
  const bool ok = room && cf_cons_ok(S, J, o.id, room);
  const double fa = ok ? 1.0 - ((double)(o.fc - J.c) * o.hTc + (double)(o.fm - J.m) * o.hTm) : 0.0;
  const bool cand = ok && fa >= fmax - CF_BAND;
  const unsigned Tc = S.cls[isov && o.valid ? o.cls : 0u].Tc, Tm = S.cls[isov && o.valid ? o.cls : 0u].Tm;
  const double lit = cand ? cf_literal(Tc, Tm, o.fc, o.fm, J.c, J.m, sc, sm) : 0.0;
  const unsigned long long lb = (unsigned long long)__double_as_longlong(lit);
  const unsigned long long mx = wave_max_u64(lb);
  if (mx == 0ull) return;
  const unsigned idmin = ~wave_max_u32((cand && lb == mx) ? ~o.id : 0u);
  const unsigned l0 = (unsigned)__ffsll(__ballot(cand && lb == mx && o.id == idmin)) - 1u;
  out.fa = __longlong_as_double((long long)mx), out.w0 = idmin, out.pos = l0, out.fc = (unsigned)wave_read_lane((int)o.fc, (int)l0),
  out.fm = (unsigned)wave_read_lane((int)o.fm, (int)l0), out.cls = (unsigned)wave_read_lane((int)o.cls, (int)l0);
}
// the posts of an exact turn -> the winner: greatest literal fitness, lowest offer.  src: 0 the overlay, 1..6 a logical class wave, -1 none
struct CfVerdict {
  int src;
  unsigned id, pos, fc, fm, cls, aux;
};
static __device__ __forceinline__ CfVerdict cf_verdict_exact(const CfPost* posts, unsigned lane, unsigned used_sets) {  // used_sets: bit g = set g has classes (and a wave that posts)
  CfVerdict v;
  const bool has = lane == 0u || (lane < 7u && ((used_sets >> lane) & 1u));
  CfPost p;
  p.fa = 0.0, p.w0 = 0u, p.pos = p.fc = p.fm = p.cls = p.aux = 0u;
  if (has) p = posts[lane];
  const unsigned long long fb = (unsigned long long)__double_as_longlong(p.fa);
  const unsigned long long mx = wave_max_u64(fb);
  v.src = -1, v.id = v.pos = v.fc = v.fm = v.cls = v.aux = 0u;
  if (mx == 0ull) return v;
  const unsigned idmin = ~wave_max_u32((has && fb == mx) ? ~p.w0 : 0u);
  const unsigned wl = (unsigned)__ffsll(__ballot(has && fb == mx && p.w0 == idmin)) - 1u;
  v.src = (int)wl;
  v.id = (unsigned)wave_read_lane((int)p.w0, (int)wl), v.pos = (unsigned)wave_read_lane((int)p.pos, (int)wl), v.fc = (unsigned)wave_read_lane((int)p.fc, (int)wl),
  v.fm = (unsigned)wave_read_lane((int)p.fm, (int)wl), v.cls = (unsigned)wave_read_lane((int)p.cls, (int)wl), v.aux = (unsigned)wave_read_lane((int)p.aux, (int)wl);
  return v;
}
// a word another wave of the workgroup writes, the same value in every lane (lane 0 reads it)
static __device__ __forceinline__ unsigned cf_poll(const uint32_t* p) { return (unsigned)wave_read_lane((int)ld_wg(p), 0); }
static __device__ __forceinline__ unsigned long long cf_below(unsigned s) { return (1ull << s) - 1ull; }  // s < 64

// ---- the call's constants every role reads
enum : unsigned { CFE_K = 0, CFE_M, CFE_NP, CFE_NCLS, CFE_NKIND, CFE_CMIN, CFE_MMIN, CFE_KC, CFE_KM, CFE_MINFC, CFE_MINFM, CFE_N = 12 };  // CfFixed::envw
constexpr float CF_BAND32 = 1.0f / 2097152.0f;  // 2^-21: two fitness values this close as floats MAY be inside the 2^-37 band as doubles

// fixed part of the LDS at constant offsets (the arrays whose size follows the call come behind it)
struct CfFixed {
  CfEnt board[CF_SLOTS * 8u];
  CfLog log[64];
  CfJob ring[2u * 64u];
  CfPost post2[8];
  CfClass cls[CF_MAXCLS];
  uint32_t pw[8u * CF_LV], aw[8u * CF_LV], gk[CF_MAXKIND * CF_LV], ovt[CF_LV];
  uint32_t ctrl[8];
  uint32_t ovl[192], ckept[3u * CF_MAXCLS], misc[CFX_N];
  uint32_t env_t[CF_LV], envw[CFE_N];
};
static_assert(sizeof(CfFixed) % 16 == 0, "the arrays behind it are 16-byte aligned");
#if COOK_HAS_ASM_WALK
#define CF_ASM_EPOCH_LIVE "57"
#include "classfit_asm.hpp"
static_assert(CF_EPOCH_AT == 58 && CF_OVL == 58, "classfit_asm.hpp: the overlay's size");
static_assert(offsetof(CfFixed, ctrl) == 13728 + (CF_SLOTS - 12) * 256 && offsetof(CfFixed, cls) == 9472 + (CF_SLOTS - 12) * 256 && sizeof(CfFixed) == 15280 + (CF_SLOTS - 12) * 256 && sizeof(CfClass) == 56 && offsetof(CfClass, hTc) == 32 &&
                  offsetof(CfClass, hTm) == 40 && offsetof(CfEnt, pos) == 8 && offsetof(CfEnt, cid) == 12 && offsetof(CfEnt, fc) == 16 && offsetof(CfEnt, fa) == 24 && sizeof(CfEnt) == 32 &&
                  sizeof(CfLog) == 32 && offsetof(CfLog, ofc) == 8,
              "classfit_asm.hpp: offsets of the LDS records");
typedef unsigned cf_u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned cf_u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned cf_u32x4 __attribute__((ext_vector_type(4)));
#endif

template <int ROLE>  // 0 the decider, 1 a class wave, 2 the bookkeeper
static __device__ __forceinline__ void cf_walk_role(const CfLds& S, const MatchState& st, const CfBuf& b, const unsigned lw, const unsigned rep, const unsigned nrep,
                                                    const unsigned long long t_start) {
  constexpr bool is_decider = ROLE == 0, is_class_wave = ROLE == 1, is_books = ROLE == 2;
  const unsigned tid = threadIdx.x, lane = lane_id();
  CfCtl* ctl = b.ctl;
  // (the call's constants live in LDS: what a rare path needs is read there, not kept in scalar registers across the walk)
  const unsigned K = wave_uniform_u32(S.envw[CFE_K]), cmin = wave_uniform_u32(S.envw[CFE_CMIN]), mmin = wave_uniform_u32(S.envw[CFE_MMIN]);
  const uint32_t* t = S.t;
#define n_cls wave_uniform_u32(S.envw[CFE_NCLS])
#define n_kind wave_uniform_u32(S.envw[CFE_NKIND])
#define NP wave_uniform_u32(S.envw[CFE_NP])
  // ---- role state
  CfChunkLane c;
  unsigned nch_wave = 0;
  cf_setup_chunks(S.cls, n_cls, is_class_wave ? lw : 0xFFu, lane, c, nch_wave);
  c.lv = CfLv8{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, c.la = c.lv;
  unsigned my_kinds = 0;  // class wave: the kinds of its classes; decider lanes 58..63: those of "their" class wave
  {
    const unsigned forw = is_class_wave ? lw : (is_decider && lane >= CF_OVL ? lane - CF_OVL + 1u : 0xFFu);
    for (unsigned ci = 0; ci < n_cls; ++ci)
      if (S.cls[ci].wave == forw && S.cls[ci].kind < 32u) my_kinds |= 1u << S.cls[ci].kind;
  }
  const unsigned wk = is_class_wave ? wave_uniform_u32(my_kinds) : 0u;  // a class wave's kinds, in a scalar register
  // a class wave whose lanes hold ONE class answers from scalar registers (the plain case: a wave per class of hosts without gpus)
  unsigned one_cls = 0xFFFFFFFFu, u_off = 0, u_n = 0;
  double u_hTc = 0.0, u_hTm = 0.0;
  auto class_setup = [&] {
    unsigned cnt = 0, ci0 = 0;
    for (unsigned ci = 0; ci < n_cls; ++ci)
      if (S.cls[ci].wave == lw) ++cnt, ci0 = ci;
    one_cls = cnt == 1u ? ci0 : 0xFFFFFFFFu;
    const CfClass* cl = &S.cls[ci0];
    u_off = cl->off, u_n = cl->n, u_hTc = cl->hTc, u_hTm = cl->hTm;
  };
  if (is_class_wave) {
    class_setup();
    for (unsigned ch = 0; ch < nch_wave; ++ch) cf_tighten(S, lane, ch, c);
    if (rep == 0u && lw != 0u) cf_wave_tables(S, lw, lane, c, wk, n_kind);
  }
  // the decider's lanes
  CfOvLane o;
  o.valid = 0u, o.id = 0u, o.cls = 0u, o.fc = 0u, o.fm = 0u, o.hTc = 0.0, o.hTm = 0.0;
  unsigned nrm = 0, rm1 = 0xFFFFFFFFu, rm2 = 0xFFFFFFFFu;  // lanes 58..63: removals from "their" class wave so far, the positions of the last two
  unsigned minfc_all = wave_uniform_u32(S.envw[CFE_MINFC]), minfm_all = wave_uniform_u32(S.envw[CFE_MINFM]);
  unsigned matched = 0, head = 0;
  // every wave: the jobs of the batch, lane = batch slot
  unsigned jc = 0, jm = 0, jmeta = 0, jgrp = 0, jeq0 = 0, jeq1 = 0, jnov0 = 0, jnov1 = 0;
  int res = -1;
  unsigned long long b1m = 0ull;  // decider: the batch's jobs some offer lacks room for at their turn (one bit per batch lane)
  unsigned seen = 0, gen = 0;
  unsigned tight_applied = 0;  // st_tight as of the last time the wave's table rows were written (a summary recomputed since: the rows are stale)
  bool bk_room0 = false, nx_room0 = false, nx_walk = false;
  unsigned st_scans = 0, st_exact = 0, st_open = 0, st_ovwin = 0, st_gpu = 0, st_epochs = 0, st_tight = 0, st_walked = 0, st_dead = 0, st_opendead = 0, st_spins = 0, st_flips = 0,
           st_rewinds = 0;
  unsigned long long tk_epoch = 0, tk_books = 0, tk_walk = 0, tk_phase1 = 0;
#ifdef CF_PROF
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  auto job_of = [&](unsigned s) -> CfJobU {
    CfJobU J;
    J.c = (unsigned)wave_read_lane((int)jc, (int)s), J.m = (unsigned)wave_read_lane((int)jm, (int)s);
    const unsigned meta = (unsigned)wave_read_lane((int)jmeta, (int)s);
    J.kind = meta & 255u, J.L = (meta >> 8) & 15u, J.n_eq = (meta >> 12) & 15u, J.n_nov = (meta >> 16) & 15u, J.grouped = (meta >> 20) & 1u;
    J.grp = 0u, J.eq0 = J.eq1 = 0u, J.nov0 = J.nov1 = 0xFFFFFFFFu;
    if (meta >> 12) {
      J.grp = (unsigned)wave_read_lane((int)jgrp, (int)s), J.eq0 = (unsigned)wave_read_lane((int)jeq0, (int)s), J.eq1 = (unsigned)wave_read_lane((int)jeq1, (int)s),
      J.nov0 = (unsigned)wave_read_lane((int)jnov0, (int)s), J.nov1 = (unsigned)wave_read_lane((int)jnov1, (int)s);
    }
    return J;
  };
  // an unmatched job's "some offer has room" / "an offer that may take it has room" from the level maxima (exact at a batch boundary)
  auto tables_room_any = [&](unsigned L, unsigned m) -> bool {
    bool r = S.ovt[L] > m;
    for (unsigned x = 1; x <= (unsigned)CF_CW; ++x) r = r || S.aw[x * CF_LV + L] > m;
    return r;
  };
  auto tables_walk = [&](unsigned kind, unsigned L, unsigned m) -> bool {
    bool r = false;
    if (kind == 0u) {
      r = S.ovt[L] > m;
      for (unsigned x = 1; x <= (unsigned)CF_CW; ++x) r = r || S.pw[x * CF_LV + L] > m;
    } else if (kind < (unsigned)CF_MAXKIND) {
      r = S.gk[kind * CF_LV + L] > m;
    }
    return r;
  };
  // the bookkeeper reads the next batch's jobs (registers), settles who is walked; lane = batch slot
  CfJob nxt;
  nxt.c = nxt.m = nxt.meta = nxt.grp = nxt.eq[0] = nxt.eq[1] = nxt.nov[0] = nxt.nov[1] = 0u;
  auto books_next = [&](unsigned nbase) {  // tables are exact: the walk mask and the "room at its start" of the batch at nbase, its jobs into the ring
    const unsigned nn = nbase < K ? cf_min(64u, K - nbase) : 0u;
    const unsigned nslot = (nbase >> 6) & 1u;
    const bool in = lane < nn;
    if (in) S.ring[nslot * 64u + lane] = nxt;
    const unsigned kind = nxt.meta & 255u, L = (nxt.meta >> 8) & 15u;
    nx_room0 = in && tables_room_any(L, nxt.m);
    nx_walk = in && tables_walk(kind, L, nxt.m);
    const unsigned long long wm = __ballot(nx_walk);
    if (lane == 0)
      S.misc[CFX_WALK_LO] = (unsigned)wm, S.misc[CFX_WALK_HI] = (unsigned)(wm >> 32), S.misc[CFX_LOGN] = 0u, S.misc[CFX_LOG_APPLIED] = 0u,
      S.ctrl[1] = wm ? (unsigned)__ffsll(wm) - 1u : 0u;  // (the class waves run ahead of THIS step)
  };
  __syncthreads();  // (the tables of the prologue are written)
  if (is_books) {
    if (lane < cf_min(64u, K)) nxt = b.jobs[lane];
    books_next(0u);
    bk_room0 = nx_room0;
    if (64u + lane < K) nxt = b.jobs[64u + lane];
  }
  if (is_decider) cook_set_prio_high();
  const unsigned long long t_loop = cook_ticks();
  __syncthreads();
  const bool isov = lane < CF_OVL;
  const unsigned clw = isov ? 0u : lane - CF_OVL + 1u;  // decider: the lane's column of the board

  for (unsigned base = 0; base < K; base += 64u) {
    const unsigned bn = cf_min(64u, K - base);
    const unsigned slot = (base >> 6) & 1u;
    const unsigned long long walkmask = wave_uniform_u64((unsigned long long)S.misc[CFX_WALK_LO] | (unsigned long long)S.misc[CFX_WALK_HI] << 32);
    const unsigned nw = (unsigned)__popcll(walkmask);
    st_walked += nw;
    if (!is_books) {
      const CfJob j = S.ring[slot * 64u + (lane < bn ? lane : 0u)];
      const bool in = lane < bn;
      jc = in ? j.c : 0u, jm = in ? j.m : 0u, jmeta = in ? j.meta : CF_KIND_NONE, jgrp = j.grp, jeq0 = j.eq[0], jeq1 = j.eq[1], jnov0 = j.nov[0], jnov1 = j.nov[1];
    }
    res = -1;
    if (is_decider) b1m = __ballot(!(jc <= minfc_all && jm <= minfm_all));
    unsigned logn = 0;  // decider: placements of the batch so far
    unsigned long long todo = walkmask;  // decider: the walked jobs not decided yet; class waves: not answered yet
    unsigned cur_ord = 0;                // the walked ordinal (0, 1, ... over the batch's walked jobs) of the first job of todo
    unsigned ph = 0;                     // class waves: cur_ord modulo the waves that share the set's jobs (wave `rep` of `nrep` answers the ordinals = rep)
    bool batch_done = nw == 0u;
    const unsigned long long tw0 = CF_TICKS();
    CF_PROF_T(cw0);
    while (!batch_done) {
      unsigned md = 0;  // the collective turn this wave leaves its loop for
      if (is_decider) {
        // ================================================= the decider =================================================
#if COOK_HAS_ASM_WALK
        // the lane's state and the batch's jobs as the hand-placed step holds them (classfit_asm.hpp)
        cf_u32x16 ST = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        cf_u32x8 JB = {jc, jm, jmeta, clw * 32u, lane, jeq0, jeq1, 0u};
        const unsigned long long rel0 = cook_ballot(!isov & ((my_kinds & 1u) != 0u)), ovm = cf_below(CF_OVL);
        const cf_u32x8 MK = {wave_uniform_u32((unsigned)rel0), wave_uniform_u32((unsigned)(rel0 >> 32)), wave_uniform_u32((unsigned)ovm), wave_uniform_u32((unsigned)(ovm >> 32)),
                             wave_uniform_u32(cook_lds_off(S.attr8)), 0u, 0u, 0u};
        const unsigned a_board = wave_uniform_u32(cook_lds_off(S.board)), a_log = wave_uniform_u32(cook_lds_off(S.log));
        auto st_pack = [&] {
          const unsigned long long hc = (unsigned long long)__double_as_longlong(o.hTc), hm = (unsigned long long)__double_as_longlong(o.hTm);
          ST[0] = o.valid, ST[1] = o.id, ST[2] = (unsigned)hc, ST[3] = (unsigned)(hc >> 32), ST[4] = (unsigned)hm, ST[5] = (unsigned)(hm >> 32), ST[6] = o.cls, ST[7] = o.fc, ST[8] = o.fm,
          ST[9] = nrm, ST[10] = rm1, ST[11] = rm2, ST[12] = (unsigned)res;
        };
        auto st_unpack = [&] {
          o.valid = ST[0], o.id = ST[1], o.hTc = __longlong_as_double((long long)((unsigned long long)ST[3] << 32 | ST[2])),
          o.hTm = __longlong_as_double((long long)((unsigned long long)ST[5] << 32 | ST[4])), o.cls = ST[6], o.fc = ST[7], o.fm = ST[8], nrm = ST[9], rm1 = ST[10], rm2 = ST[11],
          res = (int)ST[12];
        };
#endif
        while (md == 0u) {
          // (scalars for the compiler too: one of them in a vector register turns this loop into an exec-mask loop and every counter into vector arithmetic)
          todo = wave_uniform_u64(todo), cur_ord = wave_uniform_u32(cur_ord), logn = wave_uniform_u32(logn), matched = wave_uniform_u32(matched), gen = wave_uniform_u32(gen);
          minfc_all = wave_uniform_u32(minfc_all), minfm_all = wave_uniform_u32(minfm_all), b1m = wave_uniform_u64(b1m);
#if COOK_HAS_ASM_WALK
          {  // plain steps, one behind the other, as long as they are plain (classfit_asm.hpp); the step that is not is the C++ step's below
            st_pack();
            CF_PROF_T(f0);
            while (todo != 0ull) {
              const unsigned fs = (unsigned)__ffsll(todo) - 1u;
#define CF_U(x) wave_uniform_u32((unsigned)(x))
              cf_u32x8 SC = {CF_U(matched), CF_U(minfc_all), CF_U(b1m), CF_U(b1m >> 32), CF_U(minfm_all), 1u, 0u, 0u};
              const cf_u32x8 AR = {CF_U(fs), CF_U((base + fs) << 12 | (gen & 15u) << 8), CF_U(a_board + cf_slot(cur_ord) * 8u * (unsigned)sizeof(CfEnt)), CF_U(fs | cur_ord << 8),
                                   CF_U(a_log + logn * (unsigned)sizeof(CfLog)), CF_U(cmin), CF_U(mmin), a_board};
#undef CF_U
              asm volatile(CF_ASM_DECIDER_STEP
                           : "+{v[64:79]}"(ST), "+{s[36:43]}"(SC)
                           : "{v[80:87]}"(JB), "{s[44:51]}"(AR), "{s[52:59]}"(MK)
                           : "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108",
                             "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69",
                             "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "vcc", "scc", "memory");
#ifdef CF_DELAY_D  // robustness study: a slow decider
              __builtin_amdgcn_s_sleep(6);
#endif
              const unsigned status = wave_uniform_u32(SC[5]);
              if (status == 1u) break;
              matched = wave_uniform_u32(SC[0]), minfc_all = wave_uniform_u32(SC[1]), minfm_all = wave_uniform_u32(SC[4]);
              b1m = wave_uniform_u64((unsigned long long)SC[3] << 32 | SC[2]);
              logn += status >> 1;
              todo &= todo - 1ull, ++cur_ord;
              CF_PROF_ADD(4, 1);
            }
            CF_PROF_T(f1);
            CF_PROF_ADD(5, f1 - f0);
            WAIT_LDS();  // (the hand-placed step's LDS stores are not on the compiler's books: nothing of them is in flight behind this point)
            st_unpack();
          }
#endif
          if (todo == 0ull) {
            md = CFM_BATCH_END;
            st_lane0_b32(&S.ctrl[0], md);
            break;
          }
          CF_PROF_T(p0);
          const unsigned s = (unsigned)__ffsll(todo) - 1u;
          const unsigned ord = cur_ord;
          st_lane0_b32(&S.ctrl[1], s | ord << 8);
          const unsigned Jc = (unsigned)wave_read_lane((int)jc, (int)s), Jm = (unsigned)wave_read_lane((int)jm, (int)s), meta = (unsigned)wave_read_lane((int)jmeta, (int)s);
          const unsigned kind = meta & 255u;
          const unsigned want = (base + s) << 12 | (gen & 15u) << 8;
          // the candidates of the class waves into lanes 58..63 (every lane reads an entry: no branch on the lane)
          const CfEnt* e = &S.board[cf_slot(ord) * 8u + clw];
          const bool relevant = !isov && kind < 32u && ((my_kinds >> (kind & 31u)) & 1u);
          unsigned cpos, ccid, cfc, cfm;
          double cfa;
          bool acc;
          // the overlay's lanes that can take the job (room, and the job's constraints: a class wave's candidate has passed them)
          bool ok_ov = isov & (o.valid != 0u) & (kind == 0u) & (o.fc >= Jc) & (o.fm >= Jm);
          if (meta >> 12) {
            const CfJobU J = job_of(s);
            ok_ov = ok_ov & cf_cons_ok(S, J, o.id, ok_ov);
          }
          for (;;) {
            const unsigned t1 = ld_wg(&e->tag);
            COMPILER_FENCE();
            cpos = e->pos, ccid = e->cid, cfc = e->fc, cfm = e->fm, cfa = e->fa;
            COMPILER_FENCE();
            const unsigned t2 = ld_wg(&e->tag);
            // behind: removals of the wave the answer does not know.  An answer that knows a removal cannot name its member (zeroed in LDS before the
            // count moved), so: up to two unknown removals, and neither of the last two removed positions
            const unsigned behind = (nrm - t1) & 255u, p = cpos & 0xFFFFu;
            const bool ours = (t1 == t2) & ((t1 & ~255u) == want);
            acc = ours & (((ccid & CF_ENT_NONE) != 0u) | ((behind <= 2u) & (p != rm1) & (p != rm2)));
            if (cook_ballot(relevant & !acc) == 0ull) break;
            // An answer for this job that names a member taken since: that offer is in the overlay now, FULLER than the answer knew it.  If its lane can
            // take the job it beats whatever the wave would answer today (the members left are the ones the old answer ranked below the offer as it
            // was), so the step does not wait: the wave has no candidate.  (within an epoch an offer only moves from the arrays to the overlay.)
            for (unsigned long long rq = cook_ballot(relevant & !acc & ours); rq != 0ull; rq &= rq - 1ull) {
              const unsigned L = (unsigned)__ffsll(rq) - 1u;
              const unsigned xid = (unsigned)wave_read_lane((int)(ccid & CF_IDMASK), (int)L);
              if (cook_ballot(ok_ov & (o.id == xid)) != 0ull && lane == L) acc = true, ccid = CF_ENT_NONE;
            }
            if (cook_ballot(relevant & !acc) == 0ull) break;
            CF_STAT(++st_spins);
            SPIN_PAUSE_NEAR();  // an answer is missing: wait for it
          }
          CF_PROF_T(p1);
          CF_PROF_ADD(0, p1 - p0);
          // one evaluation of the 64 lanes
          const unsigned vfc = isov ? o.fc : cfc, vfm = isov ? o.fm : cfm, vid = isov ? o.id : (ccid & CF_IDMASK);
          const bool vvalid = isov ? (o.valid != 0u && kind == 0u) : (relevant && !(ccid & CF_ENT_NONE));
          const bool room = vvalid & (vfc >= Jc) & (vfm >= Jm);
          const bool ok = isov ? ok_ov : room;
          double fov = 1.0 - ((double)(vfc - Jc) * o.hTc + (double)(vfm - Jm) * o.hTm);
          OPAQUE_V(fov);
          const double fa = ok ? (isov ? fov : cfa) : 0.0;
          const float ff = (float)fa;
          const float mx = wave_max_f32(ff);
          const unsigned long long eqm = cook_ballot(ok & (ff == mx)), nearm = cook_ballot(ok & (ff >= mx - CF_BAND32)),
                                   ambm = cook_ballot(ok & !isov & ((ccid & CF_ENT_AMB) != 0u) & (ff >= mx - CF_BAND32));
          CF_PROF_T(p2);
          CF_PROF_ADD(1, p2 - p1);
          CF_PROF_ADD(3, 1);
          if (!(mx > 0.0f)) {  // nobody takes it
            todo &= todo - 1ull, ++cur_ord;
            continue;
          }
          unsigned l0 = (unsigned)__ffsll(eqm) - 1u;
          if ((nearm & (nearm - 1ull)) != 0ull || ambm != 0ull) {  // rare: the doubles decide whether the guard band holds several
            const unsigned long long mxb = wave_max_u64((unsigned long long)__double_as_longlong(fa));
            const double f0 = __longlong_as_double((long long)mxb);
            l0 = (unsigned)__ffsll(__ballot(ok && (unsigned long long)__double_as_longlong(fa) == mxb)) - 1u;
            const unsigned long long near = __ballot(ok && fa >= f0 - CF_BAND);
            const bool amb = (near & (near - 1ull)) != 0ull || (__ballot(ok && !isov && (ccid & CF_ENT_AMB) && fa >= f0 - CF_BAND) != 0ull);
            if (amb) {  // the literal fitness decides: every wave in lockstep
              md = CFM_EXACT;
              if (lane == 0) S.misc[CFX_EX_LANE] = s, S.misc[CFX_FMAX_LO] = (unsigned)mxb, S.misc[CFX_FMAX_HI] = (unsigned)(mxb >> 32), st_wg(&S.ctrl[0], md);
              break;
            }
          }
          todo &= todo - 1ull, ++cur_ord;
          // ---- commit
          const unsigned ofc = (unsigned)wave_read_lane((int)vfc, (int)l0), ofm = (unsigned)wave_read_lane((int)vfm, (int)l0), id = (unsigned)wave_read_lane((int)vid, (int)l0);
          const unsigned nfc = ofc - Jc, nfm = ofm - Jm;
          const bool dead = nfc < cmin || nfm < mmin;
          const bool from_ov = l0 < CF_OVL;
          ++matched;
          res = lane == s ? (int)id : res;
          b1m |= cook_ballot((jc > nfc) | (jm > nfm)) & ~cf_below(s) & ~(1ull << s);
          minfc_all = cf_min(minfc_all, nfc), minfm_all = cf_min(minfm_all, nfm);
          unsigned lpos = 0, linfo = s;
          if (from_ov) {
            CF_STAT(++st_ovwin);
            const bool me = lane == l0;
            o.fc = me ? nfc : o.fc, o.fm = me ? nfm : o.fm, o.valid = (me && dead) ? 0u : o.valid;
            if (dead) CF_STAT(++st_dead);
          } else {
            // the member leaves its class wave's arrays (a gpu host stays, occupied): zeroed in place, the wave's removal count moves on
            const bool gpu_place = kind != 0u;
            const unsigned pos = (unsigned)wave_read_lane((int)cpos, (int)l0);
            const unsigned cls2 = ((unsigned)wave_read_lane((int)ccid, (int)l0) >> 16) & 255u;
            const unsigned cw = l0 - CF_OVL + 1u;
            const unsigned cnt2 = (unsigned)wave_read_lane((int)nrm, (int)l0) + 1u;
            lpos = pos & 0xFFFFu, linfo = s | cw << 8 | (pos >> 16) << 12 | (gpu_place ? 1u : 0u) << 20;
            if (gpu_place) {
              st_lane0_b64(&S.fcm[lpos], nfc, nfm);
              if (lane == 0) S.cid[lpos] = (uint16_t)(S.cid[lpos] | CF_OCC);
              CF_STAT(++st_gpu);
            } else {
              st_lane0_b64(&S.fcm[lpos], 0u, 0u);
            }
            st_lane0_b32(&S.ctrl[1u + cw], cnt2);
            const bool me = lane == l0;
            rm2 = me ? rm1 : rm2, rm1 = me ? lpos : rm1, nrm = me ? cnt2 : nrm;
            if (!gpu_place && !dead) {  // a new overlay lane
              CF_STAT(++st_open);
              const unsigned long long vm = cook_ballot(isov & (o.valid != 0u));
              const unsigned lf = (unsigned)__ffsll(~(vm | ~cf_below(CF_OVL))) - 1u;  // (a free overlay lane: a full overlay ended the epoch at once)
              const CfClass* cl = &S.cls[cls2];
              const double hTc2 = cl->hTc, hTm2 = cl->hTm;
              const bool nl = lane == lf;
              o.valid = nl ? 1u : o.valid, o.id = nl ? id : o.id, o.cls = nl ? cls2 : o.cls, o.fc = nl ? nfc : o.fc, o.fm = nl ? nfm : o.fm, o.hTc = nl ? hTc2 : o.hTc,
              o.hTm = nl ? hTm2 : o.hTm;
              const unsigned live1 = wave_uniform_u32((unsigned)__popcll(vm) + 1u);
              if (live1 >= CF_EPOCH_AT) md = CFM_EPOCH;  // the overlay is full of live offers: back into their classes' arrays
            } else if (!gpu_place) {
              CF_STAT(++st_opendead);
            }
          }
          st_lane0_b128(&S.log[logn].info, linfo, lpos, ofc, ofm);
          ++logn;
          if ((meta >> 20) & 1u) {  // the group's next members must not land on this offer
            const unsigned grp = (unsigned)wave_read_lane((int)jgrp, (int)s);
            if (lane == 0) {
              const unsigned g0 = S.goff[grp], gn = S.gcnt[grp];
              S.gids[g0 + gn] = (uint16_t)id;
              COMPILER_FENCE();
              st_wg(&S.gcnt[grp], (uint16_t)(gn + 1u));
            }
            wave_sync();
          }
          md = wave_uniform_u32(md);  // (a scalar for the compiler too: as a vector value it makes this loop's exit divergent and every counter carried out of it a vector register)
          if (md == CFM_EPOCH) {
            if (lane == 0) S.misc[CFX_EX_LANE] = s, S.misc[CFX_LOGN] = logn, st_wg(&S.ctrl[0], md);
          }
          CF_PROF_T(p3);
          CF_PROF_ADD(2, p3 - p2);
        }
      } else if (is_class_wave) {
        // ================================================= a class wave: answers ahead of the decider =================================================
        while (md == 0u) {
#if COOK_HAS_ASM_WALK
          if (one_cls != 0xFFFFFFFFu) {  // plain answers, one behind the other (classfit_asm.hpp); what is not plain — and every event — is the C++ body's below
            const cf_u32x8 LV = {c.lv.v0, c.lv.v1, c.lv.v2, c.lv.v3, c.lv.v4, c.lv.v5, c.lv.v6, c.lv.v7};
            const cf_u32x8 JC = {jc, jm, jmeta, lane, 0u, 0u, 0u, 0u};
#define CF_U(x) wave_uniform_u32((unsigned)(x))
            const unsigned long long hcb = (unsigned long long)__double_as_longlong(u_hTc), hmb = (unsigned long long)__double_as_longlong(u_hTm);
            const cf_u32x8 AR = {CF_U(cook_lds_off(S.board)), CF_U(lw), CF_U(base), CF_U((gen & 15u) << 8), CF_U(u_off), CF_U(u_n), CF_U(cook_lds_off(S.cid)), CF_U(one_cls << 16)};
            const cf_u32x8 A2 = {CF_U(hcb), CF_U(hcb >> 32), CF_U(hmb), CF_U(hmb >> 32), CF_U(rep), CF_U(nrep), CF_U(S.cls[one_cls].kind), CF_BOARD};
            for (;;) {
              cf_u32x8 SC = {CF_U(todo), CF_U(todo >> 32), CF_U(cur_ord), CF_U(ph), CF_U(seen), 3u, 0u, 0u};
              asm volatile(CF_ASM_CLASS_STEP
                           : "+{s[36:43]}"(SC)
                           : "{v[64:71]}"(LV), "{v[72:79]}"(JC), "{s[44:51]}"(AR), "{s[52:59]}"(A2)
                           : "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "s60", "s61", "s62",
                             "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "vcc", "scc",
                             "memory");
              todo = wave_uniform_u64((unsigned long long)SC[1] << 32 | SC[0]), cur_ord = wave_uniform_u32(SC[2]), ph = wave_uniform_u32(SC[3]);
              const unsigned status = wave_uniform_u32(SC[5]);
              if (status == 0u) continue;
              if (status == 2u) {
                SPIN_PAUSE_NEAR();
                continue;
              }
              break;
            }
#undef CF_U
            WAIT_LDS();
          }
#endif
          CF_PROF_T(q0);
          unsigned mode, hw, cnt;
          {  // one trip to the LDS for the three words the decider writes
            const unsigned a0 = ld_wg(&S.ctrl[0]), a1 = ld_wg(&S.ctrl[1]), a2 = ld_wg(&S.ctrl[1u + lw]);
            mode = (unsigned)wave_read_lane((int)a0, 0), hw = (unsigned)wave_read_lane((int)a1, 0), cnt = (unsigned)wave_read_lane((int)a2, 0);
          }
          if (mode != 0u) {
            md = mode;
            break;
          }
          COMPILER_FENCE();
          // the answer of step (s, ord) onto the board.  again: the step was answered before members of ours left — an entry whose member is still in its
          // array stands (the first feasible member of a sorted array stays the first when ANOTHER one leaves; "none" stays none: placements only take
          // room away) and only gets the new count; one that names a member zeroed (or a gpu host occupied) since is computed anew
          auto answer = [&](unsigned s, unsigned ord, bool again) {
            const unsigned Jc = (unsigned)wave_read_lane((int)jc, (int)s), Jm = (unsigned)wave_read_lane((int)jm, (int)s), meta = (unsigned)wave_read_lane((int)jmeta, (int)s);
            const unsigned kind = meta & 255u;
            if (!(kind < 32u && ((wk >> kind) & 1u))) return;  // (none of our classes: the decider does not ask)
            CF_PROF_T(q1);
#ifdef CF_DELAY_C  // robustness study: slow class waves (every third answer of a wave very slow)
            __builtin_amdgcn_s_sleep(4);
            if ((ord + rep) % 3u == 0u) __builtin_amdgcn_s_sleep(40);
#endif
            CfEnt* e = &S.board[cf_slot(ord) * 8u + lw];
            const unsigned tag = (base + s) << 12 | (gen & 15u) << 8 | (seen & 255u);
            if (again) {
              // (only this wave writes the slot; the decider zeroes members: lane 0's reading counts for the wave)
              const unsigned t0 = ld_wg(&e->tag), p0 = e->pos & 0xFFFFu, c0 = e->cid;
              if (((t0 ^ tag) & ~255u) == 0u) {  // (our entry for this job)
                bool stands = (c0 & CF_ENT_NONE) != 0u;
                if (!stands) {
                  const CfFree f0 = S.fcm[p0];
                  const uint32_t cd0 = S.cid[p0];
                  stands = wave_read_lane(((f0.c | f0.m) != 0u && !(cd0 & CF_OCC)) ? 1 : 0, 0) != 0;
                }
                if (stands) {
                  st_lane0_b32(&e->tag, tag);
                  return;
                }
              }
            }
            bool done = false;
            if ((meta >> 12) == 0u && one_cls != 0xFFFFFFFFu) {
              // the plain case — one class, a job without constraints: the first member with room of the first chunk that promises one publishes itself
              const uint32_t lvL = cf_lv_get(c.lv, (meta >> 8) & 15u);
              const unsigned long long mk = cook_ballot(lvL > Jm);
              if (mk == 0ull) {
                st_lane0_b32(&e->tag, 0xFFFFFFFFu);
                st_lane0_b64(&e->pos, 0u, CF_ENT_NONE);
                st_lane0_b32(&e->tag, tag);
                done = true;
              } else {
                const unsigned ch = (unsigned)__ffsll(mk) - 1u;
                const unsigned pos0 = u_off + 64u * ch, n = u_n - 64u * ch;
                const CfFree f = S.fcm[pos0 + lane];
                const uint32_t cid = S.cid[pos0 + lane];
                const unsigned long long bb = cook_ballot((lane < n) & (f.c >= Jc) & (f.m >= Jm) & !(cid & CF_OCC));
                double fa = 1.0 - ((double)(f.c - Jc) * u_hTc + (double)(f.m - Jm) * u_hTm);
                if (bb != 0ull) {
                  const unsigned long long pm = bb & (0ull - bb);  // the first feasible lane
                  const unsigned long long fb = (unsigned long long)__double_as_longlong(fa);
                  st_mask_b32(&e->tag, pm, 0xFFFFFFFFu);
                  st_mask_b128(&e->fc, pm, f.c, f.m, (unsigned)fb, (unsigned)(fb >> 32));
                  st_mask_b64(&e->pos, pm, (pos0 + lane) | ch << 16, (cid & CF_IDMASK) | one_cls << 16 | ((cid & CF_TIE) ? CF_ENT_AMB : 0u));
                  st_mask_b32(&e->tag, pm, tag);
                  done = true;
                  CF_STAT(++st_scans);
                }
              }
            }
            if (!done) {  // constraints, several classes, a chunk whose summary promised too much
              const CfJobU J = job_of(s);
              CfAns a;
              cf_class_answer(S, J, lane, c, a, st_scans, st_tight);
              const unsigned long long fb = (unsigned long long)__double_as_longlong(a.fa);
              st_lane0_b32(&e->tag, 0xFFFFFFFFu);
              st_lane0_b128(&e->fc, a.fc, a.fm, (unsigned)fb, (unsigned)(fb >> 32));
              st_lane0_b64(&e->pos, a.pos | a.ch << 16, a.have ? (a.cid | a.cls << 16 | (a.amb ? CF_ENT_AMB : 0u)) : CF_ENT_NONE);
              st_lane0_b32(&e->tag, tag);
            }
            CF_PROF_T(q2);
            CF_PROF_ADD(0, q1 - q0);
            CF_PROF_ADD(1, q2 - q1);
            CF_PROF_ADD(3, 1);
          };
          if (cnt != seen) {  // members of ours have left: this wave's steps in flight (from the decider's step up to where the wave stands) are looked at again
            seen = cnt;
            CF_STAT(++st_rewinds);
            unsigned long long t2 = walkmask & ~cf_below(hw & 255u);
            unsigned o2 = hw >> 8, p2 = o2;
            while (p2 >= nrep) p2 -= nrep;
            for (; t2 != 0ull && o2 < cur_ord; t2 &= t2 - 1ull, ++o2, p2 = p2 + 1u == nrep ? 0u : p2 + 1u)
              if (p2 == rep && wk != 0u) answer((unsigned)__ffsll(t2) - 1u, o2, true);
          }
          if (cur_ord < (hw >> 8)) {  // the decider has gone past this wave (it takes a step without an answer that names a member it holds itself): on from its step
            todo = walkmask & ~cf_below(hw & 255u);
            cur_ord = hw >> 8;
            ph = cur_ord;
            while (ph >= nrep) ph -= nrep;
          }
          while (todo != 0ull && ph != rep) {  // (the set's other waves answer these)
            todo &= todo - 1ull, ++cur_ord;
            ph = ph + 1u == nrep ? 0u : ph + 1u;
          }
          if (todo != 0ull && wk != 0u && cur_ord < (hw >> 8) + CF_BOARD) {
            const unsigned s = (unsigned)__ffsll(todo) - 1u;
            const unsigned ord = cur_ord;
            todo &= todo - 1ull, ++cur_ord;
            ph = ph + 1u == nrep ? 0u : ph + 1u;
            answer(s, ord, false);
            continue;
          }
          if (wk != 0u) SPIN_PAUSE_NEAR();
          else SPIN_PAUSE_IDLE();
          CF_PROF_T(q4);
          CF_PROF_ADD(4, q4 - q0);
          CF_PROF_ADD(5, 1);
        }
      } else {
        // ================================================= the bookkeeper sleeps through the walk =================================================
        while (md == 0u) {
          const unsigned mode = cf_poll(&S.ctrl[0]);
          if (mode != 0u) {
            md = mode;
            break;
          }
          SPIN_PAUSE_IDLE();
        }
      }
      // ================================================= a collective turn: every wave =================================================
      EMU_SITE("classfit: collective");
      WAIT_LDS();  // (stores issued by inline asm — st_lane0 / st_mask, the plain step — have landed before the barrier lets the other waves read)
      __syncthreads();
      md = wave_uniform_u32(md);  // (every wave left its loop with the mode word the decider raised)
      bool epoch = md == CFM_EPOCH;
      const unsigned s = wave_uniform_u32(S.misc[CFX_EX_LANE]);
      if (md == CFM_EXACT) {
        ++st_exact;
        const double fmax = __longlong_as_double((long long)((unsigned long long)S.misc[CFX_FMAX_LO] | (unsigned long long)S.misc[CFX_FMAX_HI] << 32));
        CfPost mine;
        if (is_decider) {
          const CfJobU J = job_of(s);
          cf_overlay_query_exact(S, J, lane, o, fmax, cf_pow2(-(int)wave_uniform_u32(S.envw[CFE_KC])), cf_pow2(-(int)wave_uniform_u32(S.envw[CFE_KM])), mine);
          if (lane == 0) S.post2[0] = mine;
        } else if (is_class_wave && rep == 0u && lw != 0u) {
          const CfJobU J = job_of(s);
          cf_class_query_exact(S, J, lane, c, fmax, cf_pow2(-(int)wave_uniform_u32(S.envw[CFE_KC])), cf_pow2(-(int)wave_uniform_u32(S.envw[CFE_KM])), mine, st_scans);
          if (lane == 0) S.post2[lw] = mine;
        }
        EMU_SITE("classfit: exact turn");
        __syncthreads();
        if (is_decider) {
          const CfJobU J = job_of(s);
          unsigned used_sets = 0;
          for (unsigned ci = 0; ci < n_cls; ++ci) used_sets |= 1u << wave_uniform_u32(S.cls[ci].wave);
          const CfVerdict v = cf_verdict_exact(S.post2, lane, used_sets);
          // (an exact turn is raised because candidates exist: v.src >= 0)
          const unsigned nfc = v.fc - J.c, nfm = v.fm - J.m;
          const bool dead = nfc < cmin || nfm < mmin;
          const bool from_ov = v.src == 0;
          const bool gpu_place = !from_ov && J.kind != 0u;
          const bool opens = !from_ov && !gpu_place && !dead;
          ++matched;
          if (lane == s) res = (int)v.id;
          b1m |= __ballot((jc > nfc) | (jm > nfm)) & ~cf_below(s) & ~(1ull << s);
          minfc_all = cf_min(minfc_all, nfc), minfm_all = cf_min(minfm_all, nfm);
          unsigned live = (unsigned)__popcll(__ballot(isov && o.valid != 0u));
          unsigned lpos = 0, linfo = s;
          if (from_ov) {
            CF_STAT(++st_ovwin);
            if (lane == v.pos) {
              o.fc = nfc, o.fm = nfm;
              if (dead) o.valid = 0u;
            }
            if (dead) --live;
          } else {
            lpos = v.pos, linfo = s | (unsigned)v.src << 8 | v.aux << 12 | (gpu_place ? 1u : 0u) << 20;
            if (lane == CF_OVL - 1u + (unsigned)v.src) {
              if (gpu_place) S.fcm[lpos] = CfFree{nfc, nfm}, S.cid[lpos] = (uint16_t)(S.cid[lpos] | CF_OCC);
              else S.fcm[lpos] = CfFree{0u, 0u};
              ++nrm, rm2 = rm1, rm1 = lpos;
              S.ctrl[1u + (unsigned)v.src] = nrm;
            }
            if (opens) {
              CF_STAT(++st_open);
              const unsigned lf = (unsigned)__ffsll(~__ballot((isov && o.valid != 0u) || !isov)) - 1u;
              const CfClass* cl = &S.cls[v.cls];
              const double hTc2 = cl->hTc, hTm2 = cl->hTm;
              if (lane == lf) o.valid = 1u, o.id = v.id, o.cls = v.cls, o.fc = nfc, o.fm = nfm, o.hTc = hTc2, o.hTm = hTm2;
              ++live;
            } else if (gpu_place) {
              CF_STAT(++st_gpu);
            } else {
              CF_STAT(++st_opendead);
            }
          }
          if (lane == 0) {
            CfLog* lg = &S.log[logn];
            lg->info = linfo, lg->pos = lpos, lg->ofc = v.fc, lg->ofm = v.fm, lg->nfc = nfc, lg->nfm = nfm;
          }
          ++logn;
          if (J.grouped && lane == 0) {
            const unsigned g0 = S.goff[J.grp], gn = S.gcnt[J.grp];
            S.gids[g0 + gn] = (uint16_t)v.id, S.gcnt[J.grp] = (uint16_t)(gn + 1u);
          }
          todo &= ~(1ull << s), ++cur_ord;
          epoch = opens && live >= CF_EPOCH_AT;
          if (lane == 0) S.misc[CFX_EPOCH] = epoch ? 1u : 0u, S.misc[CFX_LOGN] = logn;
        }
        EMU_SITE("classfit: exact turn done");
        __syncthreads();
        epoch = wave_uniform_u32(S.misc[CFX_EPOCH]) != 0u;
      }
      if (epoch) {  // ---- the overlay is full of live offers: back into their classes' arrays
        const unsigned long long te = CF_TICKS();
        ++st_epochs;
        // (1) the overlay's lanes, sorted by (class, E, offer), into LDS
        if (is_decider) {
          const bool live = isov && o.valid != 0u;
          const CfClass* cl = &S.cls[live ? o.cls : 0u];
          const unsigned long long key = live ? ((unsigned long long)o.cls << 58 | ((unsigned long long)o.fc * cl->Tm + (unsigned long long)o.fm * cl->Tc) << 13 | (unsigned long long)o.id) : ~0ull;
          unsigned rank = 0;
          for (unsigned l = 0; l < 64u; ++l) rank += wave_read_lane_u64(key, (int)l) < key ? 1u : 0u;
          if (live) S.ovl[3u * rank] = o.cls << 16 | o.id, S.ovl[3u * rank + 1u] = o.fc, S.ovl[3u * rank + 2u] = o.fm;
          const unsigned nlive = (unsigned)__popcll(__ballot(live));
          if (lane == 0) S.misc[CFX_OVN] = nlive;
          if (isov) o.valid = 0u;
          rm1 = rm2 = 0xFFFFFFFFu;  // (positions of the old arrays: an answer from the new ones may name any position)
        }
        for (unsigned x = tid; x < 3u * CF_MAXCLS; x += CF_THREADS) S.ckept[x] = 0u;
        EMU_SITE("classfit: epoch 1");
        __syncthreads();
        // (2) members kept / inserted per class
        if (is_class_wave && rep == 0u) {
          unsigned mykept = 0;
          for (unsigned ch = 0; ch < nch_wave; ++ch) {
            const unsigned pos0 = (unsigned)wave_read_lane((int)c.pos0, (int)ch), n = (unsigned)wave_read_lane((int)c.n, (int)ch);
            const CfFree f = S.fcm[pos0 + lane];
            const unsigned kept = (unsigned)__popcll(__ballot(lane < n && (f.c | f.m) != 0u));
            if (lane == ch) mykept = kept;
          }
          if (c.cls != 0xFFu) atomicAdd(&S.ckept[c.cls], mykept);
        }
        if (is_decider && lane < S.misc[CFX_OVN]) atomicAdd(&S.ckept[CF_MAXCLS + (S.ovl[3u * lane] >> 16)], 1u);
        __syncthreads();
        if (tid == 0) {
          unsigned off = 0;
          for (unsigned ci = 0; ci < n_cls; ++ci) S.ckept[2 * CF_MAXCLS + ci] = off, off += S.ckept[ci] + S.ckept[CF_MAXCLS + ci];
        }
        __syncthreads();
        // (3) the first wave of every set merges the set's classes into the scratch arrays: kept members keep their order, the list's entries go between them
        if (is_class_wave && rep == 0u) {
          unsigned li = 0;  // first list entry of the class being merged
          for (unsigned ci = 0; ci < n_cls; ++ci) {
            const unsigned ni = wave_uniform_u32(S.ckept[CF_MAXCLS + ci]);
            if (S.cls[ci].wave == lw) {
              const unsigned noff = wave_uniform_u32(S.ckept[2 * CF_MAXCLS + ci]);
              const unsigned Tc = S.cls[ci].Tc, Tm = S.cls[ci].Tm;
              unsigned kept_before = 0, ip = li;
              const unsigned long long chunks = __ballot(c.cls == ci);
              for (unsigned long long mm = chunks; mm; mm &= mm - 1ull) {
                const unsigned ch = (unsigned)__ffsll(mm) - 1u;
                const unsigned pos0 = (unsigned)wave_read_lane((int)c.pos0, (int)ch), n = (unsigned)wave_read_lane((int)c.n, (int)ch);
                const bool in = lane < n;
                const CfFree f = S.fcm[pos0 + lane];
                const uint32_t fc = f.c, fm = f.m, cid = ci << 16 | (uint32_t)S.cid[pos0 + lane];
                const bool keep = in && (fc | fm) != 0u;
                const unsigned long long Ek = (unsigned long long)fc * Tm + (unsigned long long)fm * Tc;
                const unsigned idq = cid & CF_IDMASK;
                const unsigned long long keepm = __ballot(keep);
                if (keepm == 0ull) continue;
                // entries whose key is below the chunk's last kept member go in here
                const unsigned lastk = 63u - (unsigned)__clzll(keepm);
                const unsigned long long Elast = wave_read_lane_u64(Ek, (int)lastk);
                const unsigned idlast = (unsigned)wave_read_lane((int)idq, (int)lastk);
                unsigned ins_before = ip - li;  // list entries of the class in front of this member
                while (ip < li + ni) {
                  const unsigned ecid = S.ovl[3u * ip], efc = S.ovl[3u * ip + 1u], efm = S.ovl[3u * ip + 2u];
                  const unsigned long long Ee = (unsigned long long)efc * Tm + (unsigned long long)efm * Tc;
                  const unsigned ide = ecid & CF_IDMASK;
                  if (!(Ee < Elast || (Ee == Elast && ide < idlast))) break;
                  const bool before = keep && (Ek < Ee || (Ek == Ee && idq < ide));  // the member stays in front of the entry
                  const unsigned long long bm = __ballot(before);
                  if (keep && !before) ++ins_before;
                  const unsigned np = noff + kept_before + (unsigned)__popcll(bm) + (ip - li);
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
          const uint32_t fcq = inq ? ld_agent(&b.scr_fc[q]) : 0u, fmq = inq ? ld_agent(&b.scr_fm[q]) : 0u, cq = inq ? ld_agent(&b.scr_cid[q]) : 0u;
          bool tie = false;  // the next member of the class inside the guard band of this one
          if (q + 1u < newM) {
            const uint32_t fcn = ld_agent(&b.scr_fc[q + 1u]), fmn = ld_agent(&b.scr_fm[q + 1u]), cn = ld_agent(&b.scr_cid[q + 1u]);
            if ((cn >> 16) == (cq >> 16)) {
              const CfClass* cl = &S.cls[cq >> 16];
              tie = (unsigned long long)fcn * cl->Tm + (unsigned long long)fmn * cl->Tc <= (unsigned long long)fcq * cl->Tm + (unsigned long long)fmq * cl->Tc + cl->dE;
            }
          }
          S.fcm[q] = CfFree{fcq, fmq}, S.cid[q] = inq ? (uint16_t)((cq & (CF_OCC | CF_IDMASK)) | (tie ? CF_TIE : 0u)) : (uint16_t)0u;
        }
        if (tid < n_cls) S.cls[tid].n = S.ckept[tid] + S.ckept[CF_MAXCLS + tid], S.cls[tid].off = S.ckept[2 * CF_MAXCLS + tid];
        EMU_SITE("classfit: epoch 4");
        __syncthreads();
        // (5) lanes, summaries, tables; every removal of the batch so far is in the new arrays
        if (is_class_wave) {
          cf_setup_chunks(S.cls, n_cls, lw, lane, c, nch_wave);
          class_setup();
          for (unsigned ch = 0; ch < nch_wave; ++ch) cf_tighten(S, lane, ch, c);
          if (rep == 0u && lw != 0u) cf_wave_tables(S, lw, lane, c, wk, n_kind);
          tight_applied = st_tight;
        }
        if (tid == 0) S.misc[CFX_LOG_APPLIED] = S.misc[CFX_LOGN];
        tk_epoch += CF_TICKS() - te;
      }
      if (md == CFM_BATCH_END) batch_done = true;
      if (md != CFM_BATCH_END) {  // every answer on the board is void: a new generation; the class waves go on behind the step of the turn
        ++gen;
        if (is_class_wave) {
          todo = walkmask & ~cf_below(s) & ~(1ull << s), cur_ord = (unsigned)__popcll(walkmask & cf_below(s)) + 1u;
          ph = cur_ord;
          while (ph >= nrep) ph -= nrep;
        }
      }
      if (tid == 0) st_wg(&S.ctrl[0], 0u);
      EMU_SITE("classfit: collective done");
      __syncthreads();
      if (is_class_wave) seen = cf_poll(&S.ctrl[1u + lw]);
    }
    tk_walk += CF_TICKS() - tw0;
    CF_PROF_T(cw1);
    CF_PROF_ADD(7, cw1 - cw0);
    // ---- batch end, phase 1: the decider's books of the batch, the class waves make their summaries exact
    const unsigned long long tp0 = CF_TICKS();
    if (is_decider) {
      if (lane < bn) st.job_to_offer[base + lane] = res;
      const unsigned long long mm = __ballot(res >= 0);
      if (base == 0u) head = (unsigned)(mm & 1ull);
      const bool live = isov && o.valid != 0u;
      const unsigned nl = live ? cf_level_of(t, o.fc) : 0u;
#define CF_OV_T(i) unsigned v##i = nl > (unsigned)i ? o.fm + 1u : 0u;
      CF_FOR8(CF_OV_T)
#undef CF_OV_T
      wave_max8_u32(v0, v1, v2, v3, v4, v5, v6, v7);
      if (lane == 0) {
        S.misc[CFX_MATCH_LO] = (unsigned)mm, S.misc[CFX_MATCH_HI] = (unsigned)(mm >> 32), S.misc[CFX_B1_LO] = (unsigned)b1m, S.misc[CFX_B1_HI] = (unsigned)(b1m >> 32);
        S.misc[CFX_LOGN] = logn, S.misc[CFX_MINFC] = minfc_all, S.misc[CFX_MINFM] = minfm_all;
#define CF_OV_S(i) S.ovt[i] = v##i;
        CF_FOR8(CF_OV_S)
#undef CF_OV_S
      }
    }
    if (nw != 0u) {
      EMU_SITE("classfit: phase 1");
      __syncthreads();
      if (is_class_wave) {  // the batch's removals from this wave's chunks: summaries a member that left was the maximum of are recomputed
        const unsigned n_log = wave_uniform_u32(S.misc[CFX_LOGN]), a0 = wave_uniform_u32(S.misc[CFX_LOG_APPLIED]);
        const bool mine_e = lane >= a0 && lane < n_log && ((S.log[lane].info >> 8) & 15u) == lw;
        bool dirty = st_tight != tight_applied;
        for (unsigned long long mm = __ballot(mine_e); mm; mm &= mm - 1ull) {
          const unsigned x = (unsigned)__ffsll(mm) - 1u;
          const CfLog lg = S.log[x];
          const unsigned ch = (lg.info >> 12) & 255u;
          bool retable = ((lg.info >> 20) & 1u) != 0u;  // (a gpu placement changes the member in place)
          const unsigned nl = cf_level_of(t, lg.ofc);
#define CF_WAS_MAX(i) retable = retable || (__ballot(lane == ch && (unsigned)i < nl && c.lv.v##i == lg.ofm + 1u) != 0ull);
          CF_FOR8(CF_WAS_MAX)
#undef CF_WAS_MAX
          if (retable) {
            cf_tighten(S, lane, ch, c);
            ++st_tight;
            dirty = true;
          }
        }
        if (dirty && rep == 0u && lw != 0u) cf_wave_tables(S, lw, lane, c, wk, n_kind);
        tight_applied = st_tight;
      }
    }
    EMU_SITE("classfit: phase 2");
    __syncthreads();
    tk_phase1 += CF_TICKS() - tp0;
    // ---- phase 2: the bookkeeper: failure codes of this batch, the next batch
    if (is_books) {
      const unsigned long long tb0 = CF_TICKS();
      const unsigned long long mm = (unsigned long long)S.misc[CFX_MATCH_LO] | (unsigned long long)S.misc[CFX_MATCH_HI] << 32,
                               bb = (unsigned long long)S.misc[CFX_B1_LO] | (unsigned long long)S.misc[CFX_B1_HI] << 32;
      const unsigned n_log = wave_uniform_u32(S.misc[CFX_LOGN]);
      const CfJob j = S.ring[slot * 64u + (lane < bn ? lane : 0u)];
      const bool unm = lane < bn && !((mm >> lane) & 1ull);
      const unsigned L = (j.meta >> 8) & 15u;
      bool room = unm && tables_room_any(L, j.m);
      const bool flip = unm && !room && bk_room0 && n_log != 0u;  // room at the batch's start, none at its end
      if (__ballot(flip) != 0ull) {
        ++st_flips;
        for (unsigned x = 0; x < n_log; ++x) {
          const CfLog lg = S.log[x];
          if (flip && (lg.info & 255u) > lane && lg.ofc >= j.c && lg.ofm >= j.m) room = true;
        }
      }
      if (lane < bn && st.fail_code) {
        // failure code as match_serial's: 1 = an offer lacks room, 2 = an offer with room refuses on a constraint (every offer with room does:
        // the job stayed unmatched), 8 = no offer at all
        unsigned fail = 0u;
        if (unm) {
          fail = (((bb >> lane) & 1ull) ? 1u : 0u) | (room ? 2u : 0u);
          if (fail == 0u) fail = 8u;
        }
        st.fail_code[base + lane] = fail;
      }
      books_next(base + 64u);
      bk_room0 = nx_room0;
      if (base + 128u + lane < K) nxt = b.jobs[base + 128u + lane];
      tk_books += CF_TICKS() - tb0;
    }
    EMU_SITE("classfit: batch end");
    __syncthreads();
  }
  if (is_decider) {
    if (lane == 0) {
      st.summary[0] = matched;
      st.summary[1] = (matched == 0u || head) ? 1u : 0u;
      st.summary[2] = st_epochs;
      const unsigned long long t_end = cook_ticks();
      uint32_t* sx = ctl->stats;
      sx[CFS_MATCHED] = matched, sx[CFS_OV_WIN] = st_ovwin, sx[CFS_OPEN] = st_open, sx[CFS_OPEN_DEAD] = st_opendead, sx[CFS_GPU_PLACE] = st_gpu, sx[CFS_EPOCHS] = st_epochs,
      sx[CFS_EXACT] = st_exact, sx[CFS_WALKED] = st_walked, sx[CFS_DEAD_DROP] = st_dead, sx[CFS_BATCHES] = (K + 63u) / 64u, sx[CFS_PRESETTLED] = K - st_walked;
      sx[CFS_TICKS_TOTAL] = (uint32_t)(t_end - t_start), sx[CFS_TICKS_PROLOGUE] = (uint32_t)(t_loop - t_start), sx[CFS_TICKS_EPOCH] = (uint32_t)tk_epoch;
      sx[CFS_TICKS_WALK] = (uint32_t)tk_walk, sx[CFS_TICKS_PHASE1] = (uint32_t)tk_phase1;
      sx[CFS_HWID_DECIDER] = cook_hw_id();
    }
    const unsigned sp = wave_max_u32(!isov ? st_spins : 0u);
    if (lane == 0) ctl->stats[CFS_SPINS] = sp;
  }
  if (is_class_wave && lane == 0) atomicAdd(&ctl->stats[CFS_SCANS], st_scans), atomicAdd(&ctl->stats[CFS_TIGHTEN], st_tight), atomicAdd(&ctl->stats[CFS_REWINDS], st_rewinds);
  if (is_books && lane == 0) ctl->stats[CFS_TICKS_PRECHECK] = (uint32_t)tk_books, ctl->stats[CFS_FLIPS] = st_flips, ctl->stats[CFS_HWID_BOOKS] = cook_hw_id();
#ifdef CF_PROF
  if (lane == 0 && (is_decider || (is_class_wave && lw >= 1u && lw <= 2u && rep == 0u)))
    for (int i = 0; i < 8; ++i) ctl->stats[24 + 8 * (is_decider ? 0u : lw) + i] = (uint32_t)(prof[i] >> 4);  // units of 16 cycles
#endif
#undef n_cls
#undef n_kind
#undef NP
}

static __device__ __forceinline__ void cf_walk_pool(char* lds, const MatchIn* __restrict__ inp, const MatchState& st, const CfBuf& b) {
  const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
  CfCtl* ctl = b.ctl;
  const unsigned K = inp->K, M = ctl->M, G = inp->G;
  const unsigned NP = (M + 63u) & ~63u;
  const unsigned long long t_start = cook_ticks();
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
  // ---- layout (cf_lds_bytes_host in engine.hip computes the same sum)
  CfLds S;
  {
    CfFixed* F = (CfFixed*)lds;
    S.board = F->board, S.log = F->log, S.ring = F->ring, S.post2 = F->post2, S.cls = F->cls, S.pw = F->pw, S.aw = F->aw, S.gk = F->gk, S.ovt = F->ovt, S.ctrl = F->ctrl, S.ovl = F->ovl,
    S.ckept = F->ckept, S.misc = F->misc, S.t = F->env_t, S.envw = F->envw;
    char* p = lds + sizeof(CfFixed);
    S.fcm = (CfFree*)p, p += NP * 8u;
    S.cid = (uint16_t*)p, p += NP * 2u;
    p = lds + (((unsigned)(p - lds) + 7u) & ~7u);
    S.attr8 = (uint64_t*)p;
    if (any_eq) p += M * 8u;
    S.goff = (uint16_t*)p, p += (Gl + 1u) * 2u;
    S.gcnt = (uint16_t*)p, p += Gl * 2u;
    S.gids = (uint16_t*)p, p += Stot * 2u;
    if ((unsigned)(p - lds) > CF_LDS_BYTES) {  // (the host checks the same sum before it launches)
      if (tid == 0) atomicOr(&ctl->inelig, (unsigned)CF_X_SHAPE), st.summary[3] = 0xDEADu;
      return;
    }
  }
#ifdef CF_DIAG_CLEAR_LDS
  for (unsigned x = tid; x < CF_LDS_BYTES / 4u; x += CF_THREADS) ((uint32_t*)lds)[x] = 0u;
  __syncthreads();
#endif
  // ---- prologue: class arrays, byte table, group table
  for (unsigned q = tid; q < NP; q += CF_THREADS) {
    S.fcm[q] = q < M ? CfFree{b.pos_fc[q], b.pos_fm[q]} : CfFree{0u, 0u}, S.cid[q] = q < M ? (uint16_t)b.pos_cid[q] : (uint16_t)0u;
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
  {
    CfFixed* F = (CfFixed*)lds;
    if (tid < (unsigned)CF_LV) F->env_t[tid] = ctl->t[tid];
    if (tid == 0) {
      uint32_t* ew = F->envw;
      ew[CFE_K] = K, ew[CFE_M] = M, ew[CFE_NP] = NP, ew[CFE_NCLS] = ctl->n_cls, ew[CFE_NKIND] = ctl->n_kind, ew[CFE_CMIN] = ctl->cmin, ew[CFE_MMIN] = ctl->mmin, ew[CFE_KC] = ctl->kc,
      ew[CFE_KM] = ctl->km, ew[CFE_MINFC] = ctl->minfc_all, ew[CFE_MINFM] = ctl->minfm_all;
    }
  }
  for (unsigned x = tid; x < ctl->n_cls; x += CF_THREADS) S.cls[x] = ctl->cls[x];
  for (unsigned x = tid; x < CFX_N; x += CF_THREADS) S.misc[x] = 0u;
  for (unsigned x = tid; x < 8u * CF_LV; x += CF_THREADS) S.pw[x] = 0u, S.aw[x] = 0u;
  for (unsigned x = tid; x < 8u; x += CF_THREADS) S.ctrl[x] = 0u;
  for (unsigned x = tid; x < CF_MAXKIND * CF_LV; x += CF_THREADS) S.gk[x] = 0u;
  for (unsigned x = tid; x < (unsigned)CF_LV; x += CF_THREADS) S.ovt[x] = 0u;
  for (unsigned x = tid; x < CF_SLOTS * 8u; x += CF_THREADS) S.board[x].tag = 0xFFFFFFFFu;
  __syncthreads();
  if (w == 0u) {
    cf_walk_role<0>(S, st, b, 0u, 0u, 1u, t_start);
  } else if (w == CFW_BOOKS) {
    cf_walk_role<2>(S, st, b, 0u, 0u, 1u, t_start);
  } else {
    // the class waves: every set of classes (cf_prepare: CfClass::wave = 1..CF_CW) gets a wave; the waves left over go, one after the other, to the sets
    // of hosts without gpus (every job asks them), else to all: the waves of a set take its jobs in turn
    const unsigned p = w < CFW_BOOKS ? w - 1u : w - 2u, np = (unsigned)CF_WAVES - 2u;
    unsigned used = 0, zsets = 0;
    for (unsigned ci = 0; ci < ctl->n_cls; ++ci) {
      const unsigned g = S.cls[ci].wave;
      used |= 1u << g;
      if (S.cls[ci].kind == 0u) zsets |= 1u << g;
    }
    used = wave_uniform_u32(used), zsets = wave_uniform_u32(zsets);
    unsigned myset = 0, rep = 0, nrep = 0, k = 0;
    for (unsigned pass = 0; pass < np && k < np; ++pass) {
      const unsigned from = pass == 0u ? used : (zsets ? zsets : used);
      if (from == 0u) break;
      for (unsigned g = 1; g <= (unsigned)CF_CW && k < np; ++g)
        if ((from >> g) & 1u) {
          if (k == p) myset = g;
          ++k;
        }
    }
    k = 0;
    for (unsigned pass = 0; pass < np && k < np; ++pass) {
      const unsigned from = pass == 0u ? used : (zsets ? zsets : used);
      if (from == 0u) break;
      for (unsigned g = 1; g <= (unsigned)CF_CW && k < np; ++g)
        if ((from >> g) & 1u) {
          if (g == myset) {
            if (k < p) ++rep;
            ++nrep;
          }
          ++k;
        }
    }
    if (nrep == 0u) nrep = 1u;
    {  // the waves of a set: a divisor of CF_SLOTS (a wave beyond it idles)
      const unsigned cap = nrep >= 12u ? 12u : nrep >= 6u ? 6u : nrep == 5u ? 4u : nrep;
      if (rep >= cap) myset = 0u, rep = 0u, nrep = 1u;
      else nrep = cap;
    }
    cf_walk_role<1>(S, st, b, wave_uniform_u32(myset), wave_uniform_u32(rep), wave_uniform_u32(nrep), t_start);
  }
}

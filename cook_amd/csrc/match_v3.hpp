// match_v3.hpp — the same exact rank-ordered placement as match_v2.hpp (Fenzo scheduleOnce semantics, scheduler.clj:617-687), as ONE
// persistent workgroup per pool: no launch per round, no brute-force evaluation of every (job, offer) pair.
//
// What makes it cheap: cpuMemBinPacker (config.clj:108) is "fullness after the placement", and in exact arithmetic
//     2 * fitness(j, o) = G(o) + c_j / D_c(o) + m_j / D_m(o),      G(o) = used_cpus / D_c + used_mem / D_m  (job-independent),
// while "o has room for j" implies G(o) <= 2 - (c_j / D_c + m_j / D_m).  So the workgroup keeps, in LDS, ONE order of the live
// offers — grouped by (D_c, D_m, gpu signature), fullest first inside a group — cut into blocks of 64 with a summary each (key
// range, largest free cpus / mem, range of the denominators).  For a job, the summaries bound the best fitness any offer of a block
// could reach and tell which blocks cannot hold an offer with room; a HELPER wave visits the few blocks that can matter in the
// order of their bounds (lane = offer, every value exact fp64, operation for operation as the oracle computes it) and stops as soon
// as the entries it holds beat the bound of everything unvisited.  Bounds only PRUNE: every decision is taken on exact values,
// so the result is bit-identical to the one-job-at-a-time sweep for every input.
//
// Wave 0 WALKS the jobs in rank order exactly as match_v2's resolve kernel does — lanes own the offers committed to since the
// order was built ("touched", state in registers), the winner is max(best untouched offer of the job's list, best touched offer
// re-evaluated under the current state) — but it is fed through an LDS ring by the helper waves of its own workgroup, which run
// ahead of it against the same snapshot.  When the touched set is full (or a list ran out) the workgroup writes the touched state
// back, rebuilds the order (a "generation") and goes on: a generation boundary costs a sort in LDS, not three kernel launches.
//
// Scope (the host checks it and runs match_v2 otherwise): best fit (good-enough-fitness >= 1), no ports / named scalars, no
// balanced / attribute-equals groups, one offer per host, at most V3_MMAX offers, no reserved hosts / multi-entry gpu maps.
#pragma once
#include "match_v2.hpp"

#ifndef COOK_V3_THREADS
#define COOK_V3_THREADS COOK_SHAPE(1024, 128)
#endif
constexpr int V3_THREADS = COOK_V3_THREADS;           // wave 0 walks, the others prepare jobs
constexpr int V3_WAVES = V3_THREADS / COOK_WAVE;
constexpr int V3_MMAX = 8192;                          // offers per pool (positions are 13 bits of the sort key, u16 in the order)
constexpr int V3_NBMAX = V3_MMAX / COOK_WAVE;          // blocks of the order
constexpr int V3_BPL = V3_NBMAX / COOK_WAVE;           // blocks per lane when a wave looks at every summary (2)
#ifndef COOK_V3_L
#define COOK_V3_L 8
#endif
constexpr int V3_L = COOK_V3_L;                        // list entries per job
constexpr int V3_R = COOK_SHAPE(128, 8);               // ring entries (the emulated tests: small, so that the ring wraps all the time)
constexpr int V3_T = COOK_WAVE;                        // touched offers per generation = lanes of the walking wave
constexpr int V3_BATCH = 4;                            // blocks a helper visits per batch (their loads are in flight together)
static_assert(V3_L <= COOK_WAVE && V3_BPL >= 1, "shapes");

struct V3Ent {  // candidate-list entry: exact fitness under the generation's snapshot, offer
  double fit;
  int off;
  unsigned pad;
};
constexpr unsigned V3I_TRUNC = 1u << 8;     // feasible offers may exist beyond the list
constexpr unsigned V3I_GPU = 1u << 16, V3I_GROUPED = 1u << 17, V3I_HASGROUP = 1u << 20, V3I_FASTC = 1u << 21, V3I_SLOW = 1u << 22,
                   V3I_GFAST = 1u << 23;  // bits 18-19: group type; GFAST: the hosts to avoid are staged (gfh / n_fh / glast)
struct V3Job {  // one prepared job (a ring entry)
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
  V3Ent ent[V3_L];
};

struct V3BlockSum {  // summary of one block of 64 positions of the order (floats rounded towards the safe side)
  float kmax, kmin;      // >= the greatest / smallest key G of the block
  float maxc, maxm;      // >= the greatest free cpus / mem under the snapshot
  float min_dc, min_dm;  // <= the smallest denominators
  float max_dc, max_dm;  // >= the greatest
};

struct V3Ctl {  // in global memory: results and statistics of one call
  unsigned head, matched, head_matched, generations;
  unsigned stop_full, stop_list, stop_log, stop_other;
  unsigned walked, settled, scan_steps, opens;
  unsigned long long t_total, t_regen, t_walk_wait;  // 100 MHz ticks
  unsigned error;  // != 0: the kernel refused the input before placing anything (the host runs match_v2)
  unsigned pad;
};

struct V3Buf {
  const OfferA* oa;
  const OfferB* ob;
  const JobRec* jr;
  const JobCons* jcons;
  const MatchIn* in_dev;
  V3Ctl* ctl;
  int32_t* group_snap;          // [G] st.group_last as the generation began (the helpers' view; the walker publishes to the live array)
  const unsigned long long* job_flags;  // [0..1] jmin bits, [2] != 0: some job has a negative / non-finite request
};

struct V3Lds {
  unsigned long long skey[V3_MMAX];   // sort buffer; between regenerations its space holds the ring (see v3_ring)
  unsigned short ord[V3_MMAX];        // position -> offer
  float okey[V3_MMAX];                // position -> key (>= G of the offer under the snapshot)
  unsigned char owner[V3_MMAX];       // offer -> lane of the walker that owns it in this generation, 0xFF none
  V3BlockSum bsum[V3_NBMAX];
  unsigned rstate[V3_R];              // (position + 1) << 2 | 1 ready / 2 settled
  unsigned n_pos, n_blocks;           // live offers in the order
  unsigned next;                      // next job position a helper takes
  unsigned walk_pos;                  // first job the walker has not consumed
  unsigned gen_first;                 // first job of this generation (cutoff of the group chains)
  unsigned gen_stop, done;
  unsigned sort_n;
  unsigned abort;                     // the walker waited for a prepared job longer than V3_WAIT_TICKS (a bug, never the input): give up loudly
};
constexpr unsigned long long V3_WAIT_TICKS = 150000000ull;  // 1.5 s of the 100 MHz clock
static_assert(sizeof(V3Job) * V3_R <= sizeof(unsigned long long) * V3_MMAX, "the ring lives in the sort buffer");
static __device__ __forceinline__ V3Job* v3_ring(V3Lds& L) { return reinterpret_cast<V3Job*>(L.skey); }

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
static __device__ __forceinline__ unsigned long long v3_sort_key(const OfferA& a, const OfferB& b, double ac, double am, unsigned v, float* key_out) {
  const double dc = a.oc + a.rc, dm = a.om + a.rm;
  const double G = (a.rc + ac) / dc + (a.rm + am) / dm;
  float kf = v3_f32_up(G * (1.0 + 0x1p-40));
  if (!(kf >= 0.0f)) kf = 0.0f;
  *key_out = kf;
  const unsigned h = v3_hash((unsigned long long)__double_as_longlong(dc), (unsigned long long)__double_as_longlong(dm),
                             b.gpu_model * 2u + (b.flags & 1u), (unsigned long long)__double_as_longlong(b.gpu_count)) & 0x7FFFFu;
  return ((unsigned long long)h << 45) | ((unsigned long long)(~(unsigned)__float_as_int(kf)) << 13) | (unsigned long long)v;
}

// ---- bitonic sort of L.skey[0 .. n2) (n2 a power of two), ascending; every thread of the workgroup takes part --------------------
static __device__ void v3_sort(V3Lds& L, unsigned n2) {
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

// ---- a generation: snapshot the state, (re)build the order and its block summaries -------------------------------------------------
// The touched offers' state has been written back to st.ac / st.am / st.acount / st.alive before (by the walker).
static __device__ void v3_regen(V3Lds& L, const MatchIn& in, const MatchState& st, const V3Buf& vb) {
  const unsigned tid = threadIdx.x, NT = blockDim.x, lane = lane_id();
  const unsigned M = in.M;
  if (tid == 0) L.sort_n = 0;
  for (unsigned v = tid; v < (unsigned)V3_MMAX; v += NT) L.owner[v] = 0xFF;
  __syncthreads();
  // live offers -> sort keys (dead ones cannot take the smallest job of the call: they never come back)
  for (unsigned v0 = 0; v0 < M; v0 += NT) {
    const unsigned v = v0 + tid;
    bool live = false;
    unsigned long long sk = ~0ull;
    if (v < M) {
      live = ((st.alive[v >> 6] >> (v & 63u)) & 1ull) != 0ull;
      if (live) {
        float kf;
        sk = v3_sort_key(vb.oa[v], vb.ob[v], st.ac[v], st.am[v], v, &kf);
      }
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
    L.ord[p] = (unsigned short)(sk & 0x1FFFull);
    L.okey[p] = __int_as_float((int)~(unsigned)((sk >> 13) & 0xFFFFFFFFull));
  }
  if (tid == 0) {
    L.n_pos = n;
    L.n_blocks = (n + COOK_WAVE - 1) / COOK_WAVE;
  }
  __syncthreads();
  // block summaries: one wave per block, lane = position
  const unsigned nb = (n + COOK_WAVE - 1) / COOK_WAVE;
  for (unsigned b = wave_id(); b < nb; b += NT / COOK_WAVE) {
    const unsigned p = b * COOK_WAVE + lane;
    float kmax = 0.0f, kmin = 3.0e38f, maxc = 0.0f, maxm = 0.0f, ndc = 3.0e38f, ndm = 3.0e38f, xdc = 0.0f, xdm = 0.0f;
    if (p < n) {
      const unsigned v = L.ord[p];
      const OfferA a = vb.oa[v];
      const double ac = st.ac[v], am = st.am[v];
      kmax = kmin = L.okey[p];
      maxc = v3_f32_up(a.oc - ac);
      maxm = v3_f32_up(a.om - am);
      ndc = v3_f32_down(a.oc + a.rc);
      ndm = v3_f32_down(a.om + a.rm);
      xdc = v3_f32_up(a.oc + a.rc);
      xdm = v3_f32_up(a.om + a.rm);
    }
    for (int d = 32; d >= 1; d >>= 1) {
      kmax = fmaxf(kmax, __shfl_xor(kmax, d, COOK_WAVE));
      kmin = fminf(kmin, __shfl_xor(kmin, d, COOK_WAVE));
      maxc = fmaxf(maxc, __shfl_xor(maxc, d, COOK_WAVE));
      maxm = fmaxf(maxm, __shfl_xor(maxm, d, COOK_WAVE));
      ndc = fminf(ndc, __shfl_xor(ndc, d, COOK_WAVE));
      ndm = fminf(ndm, __shfl_xor(ndm, d, COOK_WAVE));
      xdc = fmaxf(xdc, __shfl_xor(xdc, d, COOK_WAVE));
      xdm = fmaxf(xdm, __shfl_xor(xdm, d, COOK_WAVE));
    }
    if (lane == 0) {
      V3BlockSum s;
      s.kmax = kmax, s.kmin = kmin, s.maxc = maxc, s.maxm = maxm, s.min_dc = ndc, s.min_dm = ndm, s.max_dc = xdc, s.max_dm = xdm;
      L.bsum[b] = s;
    }
  }
  __syncthreads();  // (the sort buffer is free from here on: the ring may be written)
}

// ---- a helper wave prepares job k: candidate list + failure counts under the generation's snapshot ----------------------------------
// Returns false when the generation was stopped while the wave was waiting for a ring slot (nothing was published).
static __device__ void v3_prepare(V3Lds& L, const MatchIn& in, const MatchState& st, const V3Buf& vb, unsigned k, V3Job& J, unsigned* steps_out) {
  const unsigned lane = lane_id();
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
  const int cutoff = (int)L.gen_first;
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
  // ---- which blocks can matter, and how good an offer of each could be -----------------------------------------------------------
  const unsigned nb = L.n_blocks;
  double ub[V3_BPL];
#pragma unroll
  for (int q = 0; q < V3_BPL; ++q) {
    const unsigned b = (unsigned)q * COOK_WAVE + lane;
    ub[q] = -1.0;
    if (b < nb) {
      const V3BlockSum s = L.bsum[b];
      const double need_lo = (j.c / (double)s.max_dc + j.m / (double)s.max_dm) * (1.0 - 0x1p-30);
      const double need_hi = (j.c / (double)s.min_dc + j.m / (double)s.min_dm) * (1.0 + 0x1p-30);
      const double thr = 2.0 - need_lo;
      // an offer with room has G <= 2 - need (used = D - free): a block whose smallest key exceeds that holds none
      const bool room = (double)s.maxc >= j.c && (double)s.maxm >= j.m && (double)s.kmin * (1.0 - 0x1p-21) <= thr;
      if (room) {
        double u = ((double)s.kmax + need_hi) * 0.5;
        if (u > 1.0) u = 1.0;  // (room implies fitness <= 1)
        ub[q] = u * (1.0 + 0x1p-30) + 0x1p-60;
      }
    }
  }
  // ---- visit the blocks in the order of their bounds -----------------------------------------------------------------------------
  double lf[V3_BATCH];  // this lane's candidates of the current batch
  int lv[V3_BATCH];
  double tf[V3_L];      // the job's list so far (wave-uniform)
  int ti[V3_L];
#pragma unroll
  for (int q = 0; q < V3_L; ++q) tf[q] = -1.0, ti[q] = -1;
  unsigned n_res = 0, n_feas = 0, n_zero = 0, steps = 0;
  bool more = false;  // blocks were left unvisited that may hold feasible offers
  for (unsigned guard = 0; guard < 4u * (unsigned)V3_NBMAX; ++guard) {
    // the next (up to) V3_BATCH blocks by bound
    unsigned bsel[V3_BATCH];
    int nsel = 0;
    double next_ub = -1.0;
#pragma unroll
    for (int s = 0; s < V3_BATCH + 1; ++s) {
      double best = -1.0;
      int bq = -1;
#pragma unroll
      for (int q = 0; q < V3_BPL; ++q)
        if (ub[q] > best) best = ub[q], bq = q;
      const unsigned long long key = best > 0.0 ? (((unsigned long long)__double_as_longlong(best)) & ~0xFFull) | (unsigned long long)(255u - (unsigned)(bq * COOK_WAVE + (int)lane)) : 0ull;
      const unsigned long long mk = wave_max_u64(key);
      if (mk == 0ull) break;  // wave-uniform
      const unsigned b = 255u - (unsigned)(mk & 0xFFull);
      const double bu = __longlong_as_double((long long)(mk & ~0xFFull)) ;
      if (s == V3_BATCH) {  // the best of what stays behind: only its bound is needed
        next_ub = bu + 0x1p-40;
        break;
      }
      // enough entries that all beat this block's bound: nothing left can enter the list
      if (ti[V3_L - 1] >= 0 && tf[V3_L - 1] > bu + 0x1p-40) {
        next_ub = bu + 0x1p-40;
        break;
      }
      bsel[nsel++] = b;
      if ((b & 63u) == lane) {
#pragma unroll
        for (int q = 0; q < V3_BPL; ++q)
          if ((unsigned)q == (b >> 6)) ub[q] = -1.0;
      }
    }
    if (nsel == 0) {
      more = next_ub > 0.0;
      break;
    }
    // evaluate the selected blocks: lane = offer, exact values
#pragma unroll
    for (int s = 0; s < V3_BATCH; ++s) {
      lf[s] = -1.0;
      lv[s] = -1;
      if (s < nsel) {
        const unsigned p = bsel[s] * COOK_WAVE + lane;
        bool res = false, feas = false, zero = false;
        if (p < L.n_pos) {
          const unsigned v = L.ord[p];
          const OfferA a = vb.oa[v];
          const double ac = st.ac[v], am = st.am[v];
          res = !(ac + j.c > a.oc || am + j.m > a.om);
          if (res) {
            const OfferB o = vb.ob[v];
            bool ok = static_fast(j, o, in, v);
            if (ok && fastc) {
              unsigned diff = (E.req_host ^ (o.host + 1u)) & E.wild_host;
#pragma unroll
              for (int x = 0; x < MV_NA; ++x) {
                const unsigned av = (in.o_attr && (unsigned)x < in.n_attr) ? in.o_attr[(size_t)v * in.n_attr + x] : 0u;
                diff |= (E.req[x] ^ av) & E.wild[x];
              }
              bool hit = E.impossible;
#pragma unroll
              for (int q = 0; q < MV_NC; ++q) hit = hit | (E.novel[q] == o.host);
              ok = diff == 0u && !hit;
            }
            if (ok && slow) ok = static_pass_dev(vb.in_dev, jj, v);
            if (ok) ok = dyn_fast(j, o, st.acount[v]);
            if (ok && n_fh > 0) {
              bool taken = false;
#pragma unroll
              for (int q = 0; q < MV_FH; ++q) taken = taken | (fh[q] == o.host);
              ok = !taken;
            }
            if (ok && grouped && n_fh == -2) ok = group_pass_dev(vb.in_dev, st_cut, jj, v);
            if (ok) {
              const double fit = fitness_of(a, ac, am, j.c, j.m);
              if (fit > 0.0) {
                feas = true;
                lf[s] = fit;
                lv[s] = (int)v;
              } else {
                zero = true;
              }
            }
          }
        }
        n_res += (unsigned)__popcll(__ballot(res));
        n_feas += (unsigned)__popcll(__ballot(feas));
        n_zero += (unsigned)__popcll(__ballot(zero));
        ++steps;
      }
    }
    // merge the batch into the list: repeatedly the best candidate over lanes and batch slots (fitness desc, offer asc)
    for (int r = 0; r < V3_L; ++r) {
      double best = -1.0;
      int bv = -1, bs = -1;
#pragma unroll
      for (int s = 0; s < V3_BATCH; ++s)
        if (lv[s] >= 0 && (lf[s] > best || (lf[s] == best && lv[s] < bv))) best = lf[s], bv = lv[s], bs = s;
      const unsigned long long key = bv >= 0 ? (unsigned long long)__double_as_longlong(best) : 0ull;
      const unsigned long long mk = wave_max_u64(key);
      if (mk == 0ull) break;  // no candidate left
      const double mf = __longlong_as_double((long long)mk);
      if (!(ti[V3_L - 1] < 0 || mf > tf[V3_L - 1])) {
        // cannot enter a full list (an equal fitness with a lower offer index could: compare indices)
        if (!(mf == tf[V3_L - 1])) break;
      }
      const unsigned long long tie = __ballot(key == mk);
      int widx;
      if ((tie & (tie - 1ull)) == 0ull)
        widx = wave_read_lane(bv, __ffsll((unsigned long long)tie) - 1);
      else
        widx = (int)(0x7FFFFFFFu - wave_max_u32(key == mk ? 0x7FFFFFFFu - (unsigned)bv : 0u));
      if (key == mk && bv == widx) {  // the owner drops it
#pragma unroll
        for (int s = 0; s < V3_BATCH; ++s)
          if (s == bs) lv[s] = -1;
      }
      // insert (mf, widx) into the uniform list if it beats the last entry
      const bool enters = ti[V3_L - 1] < 0 || mf > tf[V3_L - 1] || (mf == tf[V3_L - 1] && widx < ti[V3_L - 1]);
      if (!enters) break;
      tf[V3_L - 1] = mf;
      ti[V3_L - 1] = widx;
#pragma unroll
      for (int q = V3_L - 1; q > 0; --q) {
        const bool sw = ti[q - 1] < 0 || tf[q] > tf[q - 1] || (tf[q] == tf[q - 1] && ti[q] < ti[q - 1]);
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
    // candidates of this batch that did not make the list are feasible offers beyond it
    {
      bool left = false;
#pragma unroll
      for (int s = 0; s < V3_BATCH; ++s) left = left | (lv[s] >= 0);
      if (__any(left)) more = true;
    }
    if (next_ub <= 0.0) {  // nothing was left behind by the selection
      bool any = false;
#pragma unroll
      for (int q = 0; q < V3_BPL; ++q) any = any | (ub[q] > 0.0);
      if (!__any(any)) break;
    }
    // stop once the list is full and its last entry beats every unvisited block's bound
    if (ti[V3_L - 1] >= 0) {
      double rest = -1.0;
#pragma unroll
      for (int q = 0; q < V3_BPL; ++q) rest = ub[q] > rest ? ub[q] : rest;
      const unsigned long long rk = wave_max_u64(rest > 0.0 ? (unsigned long long)__double_as_longlong(rest) : 0ull);
      if (rk == 0ull) break;
      if (tf[V3_L - 1] > __longlong_as_double((long long)rk)) {
        more = true;
        break;
      }
    }
  }
  // a list that is not full holds EVERY feasible offer only if no block that might hold one was skipped: the loop above only
  // skips blocks once the list is full, so a short list is complete
  int n_out = 0;
#pragma unroll
  for (int q = 0; q < V3_L; ++q) n_out += ti[q] >= 0 ? 1 : 0;
  const bool trunc = n_out == V3_L && more;
  const unsigned M = in.M;
  const unsigned c1 = M - n_res, c2 = n_res - n_feas - n_zero, c4 = n_zero;  // exact when the scan visited every block with room (n_out < V3_L)
  if (lane == 0) {
    J.c = j.c, J.m = j.m, J.g = j.g;
    J.k = k, J.jj = jj;
    J.gpu_model = j.gpu_model;
    J.reserved_host = j.reserved_host;
    J.group = j.group;
    J.info = (unsigned)n_out | (trunc ? V3I_TRUNC : 0u) | (j.g > 0 ? V3I_GPU : 0u) | (grouped ? V3I_GROUPED : 0u) | (gtype << 18) |
             (j.group != 0xFFFFFFFFu ? V3I_HASGROUP : 0u) | (fastc ? V3I_FASTC : 0u) | (slow ? V3I_SLOW : 0u) |
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
#pragma unroll
    for (int q = 0; q < V3_L; ++q) {
      V3Ent e;
      e.fit = tf[q], e.off = ti[q], e.pad = 0;
      J.ent[q] = e;
    }
  }
  *steps_out = steps;
}

// ---- the helper waves of a generation: take job positions, prepare them, publish them through the ring ---------------------------
static __device__ void v3_helper(V3Lds& L, const MatchIn& in, const MatchState& st, const V3Buf& vb, unsigned* n_steps, unsigned* n_settled) {
  const unsigned lane = lane_id();
  const unsigned K = in.K;
  for (;;) {
    if (ld_wg(&L.gen_stop) != 0u) return;
    unsigned p = 0;
    if (lane == 0) p = atomicAdd(&L.next, 1u);
    p = (unsigned)__shfl((int)p, 0, COOK_WAVE);
    if (p >= K) {  // nothing left to prepare: wait for the end of the generation
      while (ld_wg(&L.gen_stop) == 0u) {
        EMU_SITE("v3 helper: idle");
        SPIN_PAUSE();
      }
      return;
    }
    // a free ring slot: the walker is at most V3_R - 1 jobs behind
    bool stopped = false;
    while (p - ld_wg(&L.walk_pos) >= (unsigned)V3_R) {
      if (ld_wg(&L.gen_stop) != 0u) {
        stopped = true;
        break;
      }
      EMU_SITE("v3 helper: ring full");
      SPIN_PAUSE();
    }
    if (stopped) return;
    V3Job& J = v3_ring(L)[p % (unsigned)V3_R];
    unsigned steps = 0;
    v3_prepare(L, in, st, vb, p, J, &steps);
    wave_sync();
    *n_steps += steps;
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
    const bool trivial = (info & 0xFFu) == 0u && c1 > 0u && (c2 == 0u || c2 > (unsigned)V3_T) && c4 == 0u;
    if (trivial) {
      if (lane == 0 && st.fail_code) st.fail_code[p] = 1u | (c2 ? 2u : 0u) | (c4 ? 4u : 0u);
      *n_settled += 1u;
    }
    lds_release();
    if (lane == 0) st_wg(&L.rstate[p % (unsigned)V3_R], ((p + 1u) << 2) | (trivial ? 2u : 1u));
  }
}

// ---- the walker: one generation ------------------------------------------------------------------------------------------------------
struct V3WalkStats {
  unsigned matched, head_matched, walked, opens, stop;  // stop: 1 list ran out, 2 touched set full, 3 group log full, 0 all jobs done
  unsigned long long wait_ticks;
};
static __device__ void v3_walk(V3Lds& L, const MatchIn& in, MatchState st, const V3Buf& vb, V3WalkStats& ws) {
  const unsigned lane = lane_id();
  const unsigned K = in.K;
  const uint32_t* const j_index = in.j_index;
  // the touched offer of this lane
  int t_v = -1;
  double t_oc = 0, t_om = 0, t_rc = 0, t_rm = 0, t_invc = 0, t_invm = 0;
  double t_ac = 0, t_am = 0, t_basec = 0, t_basem = 0, t_ac0 = 0, t_am0 = 0;
  int t_acount = 0, t_acount0 = 0;
  OfferB t_o;
  t_o.host = 0, t_o.gpu_model = 0, t_o.gpu_count = 0.0, t_o.run_count = 0, t_o.task_slack = 0x7FFFFFFF, t_o.flags = 0, t_o.pad = 0;
  unsigned t_attr[MV_NA];
#pragma unroll
  for (int x = 0; x < MV_NA; ++x) t_attr[x] = 0u;
  // group members placed in this generation (one per lane, in placement order)
  unsigned lg_group = 0xFFFFFFFFu, lg_host = 0u, n_log = 0u;
  int lg_k = -1;
  unsigned nT = 0;
  unsigned p = L.walk_pos;
  constexpr double EPS_HI = 1.0 + 0x1p-38, EPS_LO = 1.0 - 0x1p-38;
  ws.stop = 0;
  while (p < K) {
    // ---- skip the jobs the helpers settled; wait for the next prepared one ----------------------------------------------------------
    {
      const unsigned long long t0 = cook_ticks();
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
      ws.wait_ticks += cook_ticks() - t0;
      if (p >= K) break;
    }
    lds_acquire();
    const V3Job& J = v3_ring(L)[p % (unsigned)V3_R];
    const unsigned info = J.info, k = p;
    const double c = J.c, m = J.m;
    const int nc = (int)(info & 0xFFu);
    const bool job_gpu = (info & V3I_GPU) != 0u, grouped = (info & V3I_GROUPED) != 0u, has_group = (info & V3I_HASGROUP) != 0u;
    const unsigned gtype = (info >> 18) & 3u, g = has_group ? J.group : 0xFFFFFFFFu;
    const bool trunc = (info & V3I_TRUNC) != 0u;
    ws.walked += 1u;
    // list entry `lane`
    double e_fit = -1.0;
    int e_off = -1;
    unsigned owner = 0xFEu;
    if ((int)lane < nc) {
      e_fit = J.ent[lane].fit;
      e_off = J.ent[lane].off;
      owner = L.owner[e_off];
    }
    // ---- every touched offer under the current state ----------------------------------------------------------------------------------
    const bool t_on = t_v >= 0;
    JobRec jr;
    jr.c = c, jr.m = m, jr.g = J.g, jr.gpu_model = J.gpu_model, jr.reserved_host = J.reserved_host, jr.group = J.group, jr.flags = 0;
    const bool res_ok = t_on && !(t_ac + c > t_oc || t_am + m > t_om);
    bool con_ok = t_on && static_fast(jr, t_o, in, (unsigned)(t_on ? t_v : 0)) && dyn_fast(jr, t_o, t_acount);
    if (info & V3I_FASTC) {  // (wave-uniform)
      unsigned diff = (J.req_host ^ (t_o.host + 1u)) & J.wild_host;
#pragma unroll
      for (int x = 0; x < MV_NA; ++x) diff |= (J.req[x] ^ t_attr[x]) & J.wild[x];
      bool hit = J.impossible != 0u;
#pragma unroll
      for (int q = 0; q < MV_NC; ++q) hit = hit | (J.novel[q] == t_o.host);
      con_ok = con_ok && diff == 0u && !hit;
    }
    if ((info & V3I_SLOW) && con_ok) con_ok = static_pass_dev(vb.in_dev, J.jj, (unsigned)t_v);
    unsigned long long ghits = 0ull;  // log entries of this job's group
    bool g_general = false;            // the group check goes through the chains in HBM
    if (has_group) {
      ghits = __ballot(lane < n_log && lg_group == g);
      if (grouped) {
        if ((info & V3I_GFAST) && J.n_fh >= 0 && n_log <= (unsigned)COOK_WAVE) {
          bool forb = false;
#pragma unroll
          for (int q = 0; q < MV_FH; ++q) forb = forb | (t_o.host == J.gfh[q]);
          for (unsigned long long hm = ghits; hm != 0ull; hm &= hm - 1ull) {
            const unsigned h = (unsigned)wave_read_lane((int)lg_host, __ffsll((unsigned long long)hm) - 1);
            forb = forb | (t_o.host == h);
          }
          con_ok = con_ok && !forb;
        } else {
          g_general = true;
          if (con_ok) con_ok = group_pass_dev(vb.in_dev, st, J.jj, (unsigned)t_v);
        }
      }
    }
    const double nc_ = t_basec + c, nm_ = t_basem + m;
    const double a1 = nc_ * t_invc, a2 = nm_ * t_invm;
    const double fa = (a1 + a2) * 0.5;
    const bool cand = res_ok && con_ok;
    const bool sane = a1 >= 0.0 && a2 >= 0.0 && fa > 0.0;
    bool need_exact = __any(cand && !sane);
    const unsigned long long cand_mask = __ballot(cand);
    int win = -1, win_lane = -1;
    double u_fit = -1.0;
    int u_off = -1;
    bool exhausted = false;
    unsigned pe_bits = 8u;
    double pe_fit = 0.0;
    do {
      // no feasible offer under the snapshot and none of zero fitness (a placement could lift that one): stays unmatched, only the
      // summary may move
      if (nc == 0 && !grouped && J.f4 == 0) break;
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
        }
      }
      bool decided = false;
      if (!need_exact) {
        if (cand_mask == 0ull) {
          win = u_off;
          decided = true;
        } else {
          const unsigned long long key = cand ? (unsigned long long)__double_as_longlong(fa) : 0ull;
          const double mx = __longlong_as_double((long long)wave_max_u64(key));
          const unsigned long long near = __ballot(cand && fa >= mx * EPS_LO);
          if ((near & (near - 1ull)) == 0ull) {
            if (u_off < 0 || mx * EPS_LO > u_fit) {
              win_lane = __ffsll((unsigned long long)near) - 1;
              decided = true;
            } else if (mx * EPS_HI < u_fit) {
              win = u_off;
              decided = true;
            }
          }
        }
        if (!decided) need_exact = true;
      }
      if (need_exact) {
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
          }
        }
        Cand best{-1.0, -1};
        int best_lane = -1;
        if (feas_mask != 0ull) {
          const unsigned long long key = t_feas ? (unsigned long long)__double_as_longlong(pe_fit) : 0ull;
          const unsigned long long mx = wave_max_u64(key);
          unsigned long long tie = __ballot(t_feas && key == mx);
          int wl = __ffsll((unsigned long long)tie) - 1;
          int wv = wave_read_lane(t_v, wl);
          tie &= tie - 1ull;
          while (tie != 0ull) {
            const int l2 = __ffsll((unsigned long long)tie) - 1;
            const int v2 = wave_read_lane(t_v, l2);
            if (v2 < wv) wv = v2, wl = l2;
            tie &= tie - 1ull;
          }
          best = Cand{__longlong_as_double((long long)mx), wv};
          best_lane = wl;
        }
        if (u_off >= 0 && cand_better(Cand{u_fit, u_off}, best))
          win = u_off, win_lane = -1;
        else if (best_lane >= 0)
          win_lane = best_lane, win = -1;
        else
          win = win_lane = -1;
      }
    } while (0);
    if (exhausted) {
      ws.stop = 1;
      break;
    }
    // ---- commit ---------------------------------------------------------------------------------------------------------------------------
    if (win_lane < 0 && win >= 0 && nT == (unsigned)V3_T) {
      ws.stop = 2;  // no free lane for another touched offer: a new generation starts with this job
      break;
    }
    if ((win_lane >= 0 || win >= 0) && has_group && n_log >= (unsigned)COOK_WAVE) {
      ws.stop = 3;  // the log of this generation's group placements is full
      break;
    }
    if (win_lane >= 0) {
      if ((int)lane == win_lane) {
        t_ac += c;
        t_am += m;
        t_acount += 1;
        t_basec = t_rc + t_ac;
        t_basem = t_rm + t_am;
      }
      win = wave_read_lane(t_v, win_lane);
    } else if (win >= 0) {
      if (lane == nT) {  // the next free lane takes ownership: the offer's record from HBM (untouched: the snapshot is its state)
        const OfferA a = vb.oa[win];
        t_o = vb.ob[win];
        t_v = win;
        t_oc = a.oc, t_om = a.om, t_rc = a.rc, t_rm = a.rm, t_invc = a.inv_dc, t_invm = a.inv_dm;
        t_ac0 = st.ac[win], t_am0 = st.am[win], t_acount0 = st.acount[win];
        t_ac = t_ac0 + c;
        t_am = t_am0 + m;
        t_acount = t_acount0 + 1;
        t_basec = t_rc + t_ac;
        t_basem = t_rm + t_am;
#pragma unroll
        for (int x = 0; x < MV_NA; ++x) t_attr[x] = (in.o_attr && (unsigned)x < in.n_attr) ? in.o_attr[(size_t)win * in.n_attr + x] : 0u;
        L.owner[win] = (unsigned char)nT;
      }
      win_lane = (int)nT;
      ++nT;
      ws.opens += 1u;
      wave_sync();  // the owner table update is visible to the whole wave before the next look-up reads it
    }
    if (win >= 0) {
      ws.matched += 1u;
      if (k == 0) ws.head_matched = 1u;
      if (has_group) {  // publish a placed group member: the chains in HBM and the generation's log
        const int prev = ghits != 0ull ? wave_read_lane(lg_k, 63 - __clzll((long long)ghits)) : J.glast;
        const unsigned w_host = (unsigned)wave_read_lane((int)t_o.host, win_lane);
        if (lane == 0) {
          st_agent(&st.job_to_offer[k], win);
          st_agent(&st.job_prev[k], prev);
          st_agent(&st.group_last[g], (int)k);
        }
        if (lane == n_log) {
          lg_group = g;
          lg_host = w_host;
          lg_k = (int)k;
        }
        ++n_log;
        if (g_general) wave_sync();
      } else if (lane == 0) {
        st.job_to_offer[k] = win;
      }
      if (lane == 0 && st.fail_code) st.fail_code[k] = 0u;
    } else {
      // unmatched: failure summary = OR over offers of the first failing check under the CURRENT state: the snapshot counts, with each
      // touched offer's snapshot verdict swapped for its current one
      int d1 = 0, d2 = 0, d4 = 0;
      if (nT != 0u) {
        if (pe_bits == 8u && t_on) {
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
        unsigned p0 = 0u;
        if (t_on) {
          if (t_ac0 + c > t_oc || t_am0 + m > t_om) {
            p0 = 1u;
          } else {
            bool ok = static_fast(jr, t_o, in, (unsigned)t_v) && dyn_fast(jr, t_o, t_acount0);
            if (ok && (info & V3I_FASTC)) {
              unsigned diff = (J.req_host ^ (t_o.host + 1u)) & J.wild_host;
#pragma unroll
              for (int x = 0; x < MV_NA; ++x) diff |= (J.req[x] ^ t_attr[x]) & J.wild[x];
              bool hit = J.impossible != 0u;
#pragma unroll
              for (int q = 0; q < MV_NC; ++q) hit = hit | (J.novel[q] == t_o.host);
              ok = diff == 0u && !hit;
            }
            if (ok && (info & V3I_SLOW)) ok = static_pass_dev(vb.in_dev, J.jj, (unsigned)t_v);
            if (ok && grouped) {  // the group check as the generation began
              MatchState st0 = st;
              st0.cutoff = (int)L.gen_first;
              ok = group_pass_dev(vb.in_dev, st0, J.jj, (unsigned)t_v);
            }
            if (!ok) {
              p0 = 2u;
            } else {
              const double f0 = ((t_rc + t_ac0 + c) / (t_oc + t_rc) + (t_rm + t_am0 + m) / (t_om + t_rm)) / 2.0;
              if (!(f0 > 0.0)) p0 = 4u;
            }
          }
        }
        d1 = __popcll(__ballot(t_on && (pe_bits & 1u))) - __popcll(__ballot(t_on && (p0 & 1u)));
        d2 = __popcll(__ballot(t_on && (pe_bits & 2u))) - __popcll(__ballot(t_on && (p0 & 2u)));
        d4 = __popcll(__ballot(t_on && (pe_bits & 4u))) - __popcll(__ballot(t_on && (p0 & 4u)));
      }
      const unsigned bits = (((int)J.f1 + d1) > 0 ? 1u : 0u) | (((int)J.f2 + d2) > 0 ? 2u : 0u) | (((int)J.f4 + d4) > 0 ? 4u : 0u);
      if (lane == 0) {
        st.job_to_offer[k] = -1;
        if (st.fail_code) st.fail_code[k] = bits ? bits : 8u;
      }
    }
    ++p;
    if (lane == 0) st_wg(&L.walk_pos, p);
  }
  // ---- end of the generation: the touched offers' state back to HBM ----------------------------------------------------------------------
  if (t_v >= 0) {
    st.ac[t_v] = t_ac;
    st.am[t_v] = t_am;
    st.acount[t_v] = t_acount;
    if (t_ac + st.jmin[0] > t_oc || t_am + st.jmin[1] > t_om) atomicAnd(&st.alive[(unsigned)t_v >> 6], ~(1ull << ((unsigned)t_v & 63u)));
  }
  if (lane == 0) {
    st_wg(&L.walk_pos, p);
    st_wg(&L.gen_first, p);
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

// ---- one workgroup per pool (blockIdx.x = pool), one launch per match call -----------------------------------------------------------
__global__ void __launch_bounds__(V3_THREADS) match_v3(const PoolCtx3* __restrict__ ctx) {
  COOK_BLOCK_LDS(lds, sizeof(V3Lds));
  V3Lds& L = *reinterpret_cast<V3Lds*>(lds);
  const PoolCtx3& C = ctx[blockIdx.x];
  const MatchIn& in = C.in;
  const MatchState st = C.st;
  const V3Buf vb = C.vb;
  const unsigned tid = threadIdx.x, NT = blockDim.x, lane = lane_id();
  const unsigned long long tk0 = cook_ticks();
  // ---- refuse what the bounds do not cover (the host then runs match_v2): negative / non-finite resources ------------------------------
  if (tid == 0) {
    L.gen_stop = 0u;
    L.done = 0u;
    L.next = 0u;
    L.walk_pos = 0u;
    L.gen_first = 0u;
    L.abort = 0u;
    L.sort_n = vb.job_flags[2] != 0ull ? 1u : 0u;  // (borrowed as the "bad input" flag until the first regeneration)
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
  if (L.sort_n != 0u || in.M > (unsigned)V3_MMAX) {
    if (tid == 0) vb.ctl->error = 1u;
    return;
  }
  __syncthreads();
  V3WalkStats ws;
  ws.matched = ws.head_matched = ws.walked = ws.opens = ws.stop = 0u;
  ws.wait_ticks = 0ull;
  unsigned n_steps = 0, n_settled = 0, gens = 0, s_full = 0, s_list = 0, s_log = 0;
  unsigned long long t_regen = 0ull;
  for (;;) {
    // ---- a generation: snapshot, order, ring reset ------------------------------------------------------------------------------------
    const unsigned long long tr0 = cook_ticks();
    for (unsigned x = tid; x < in.G; x += NT) vb.group_snap[x] = ld_agent(&st.group_last[x]);
    for (unsigned x = tid; x < (unsigned)V3_R; x += NT) L.rstate[x] = 0u;
    if (tid == 0) {
      L.next = L.walk_pos;
      L.gen_stop = 0u;
    }
    v3_regen(L, in, st, vb);  // (ends with a barrier)
    t_regen += cook_ticks() - tr0;
    ++gens;
    if (tid < COOK_WAVE)
      v3_walk(L, in, st, vb, ws);
    else
      v3_helper(L, in, st, vb, &n_steps, &n_settled);
    __syncthreads();
    if (tid < COOK_WAVE) {
      s_list += ws.stop == 1u ? 1u : 0u;
      s_full += ws.stop == 2u ? 1u : 0u;
      s_log += ws.stop == 3u ? 1u : 0u;
    }
    if (L.done != 0u) break;
    __syncthreads();
  }
  if (L.abort != 0u) {
    if (tid == 0) vb.ctl->error = 2u;
    return;
  }
  // ---- statistics ---------------------------------------------------------------------------------------------------------------------
  for (int d = 32; d >= 1; d >>= 1) {
    n_steps += (unsigned)__shfl_xor((int)n_steps, d, COOK_WAVE);
    n_settled += (unsigned)__shfl_xor((int)n_settled, d, COOK_WAVE);
  }
  if (lane == 0 && tid >= COOK_WAVE) {  // (every lane of a helper wave carries the same counts: the wave's)
    atomicAdd(&vb.ctl->scan_steps, n_steps / COOK_WAVE);
    atomicAdd(&vb.ctl->settled, n_settled / COOK_WAVE);
  }
  if (tid == 0) {
    V3Ctl* c = vb.ctl;
    c->head = in.K;
    c->matched = ws.matched;
    c->head_matched = ws.head_matched;
    c->generations = gens;
    c->stop_full = s_full, c->stop_list = s_list, c->stop_log = s_log, c->stop_other = 0;
    c->walked = ws.walked;
    c->opens = ws.opens;
    c->t_total = cook_ticks() - tk0;
    c->t_regen = t_regen;
    c->t_walk_wait = ws.wait_ticks;
  }
}

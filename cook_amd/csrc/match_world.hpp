// match_world.hpp — the placements of ALL pools of a rank in ONE persistent launch.
//
// Why: a pool's placement is a chain of ~600 rounds (evaluate a window -> merge -> resolve).  With one launch per phase the chain
// pays three dependent kernel boundaries per round, MI355X runs only about four such chains at full speed (hardware queues share
// dispatch pipes), so a rank with eight pools had to run them as 4 chains x 2 pools in LOCKSTEP and every round lasted as long as
// the slower pool's walk.  Here nothing is launched per round and every pool advances on its own:
//
//   workgroup p < n_pools   WALKER of pool p: publishes "evaluate round r" / "merge round r" for its window, waits for the
//                           evaluator waves' completion count, then resolves the round (resolve_round, match_v2.hpp) in its LDS.
//   the other workgroups    EVALUATORS: TEAMS of MV_EW waves (two per workgroup), each the equivalent of one match_eval2 workgroup:
//                           the team's leader polls the pools' published words in HBM, the team takes the tiles (64 jobs x MV_OCB
//                           offers: one MV_OCW batch per wave, merged through LDS, eval_tile_t) or merge jobs (merge_job, one per
//                           wave) that the static map (item + pool offset) mod teams assigns to it, for whichever pool has a phase
//                           open.  The waves of a team meet at an LDS barrier; no workgroup barrier, no queue, no atomics in HBM
//                           besides one completion add per team and phase.  (A first version gave every WAVE a whole tile of four
//                           batches: an evaluation phase then lasted as long as its slowest wave, 89 us on average and up to 353 us
//                           — profiles/r02c_world_v1_phase_profile.txt — against ~45 us for the four-wave tile.)
//
// Synchronisation is the tested hand-off form of MI355X_MICROARCH.md: producer stores -> agent-scope release fence -> flag store;
// consumer relaxed poll -> ONE agent-scope acquire -> loads.  A walker only ever waits for evaluators and evaluators only for
// walkers' flags, every wait is bounded (MW_TIMEOUT_TICKS), and a launch that does not get all its workgroups resident gives up
// through WorldCtl::error — the host then re-runs the match with one launch per phase (match_rounds_multi), which stays tested.
#pragma once
#include "match_v2.hpp"

#ifndef COOK_MW_THREADS
// 512: 2 waves per SIMD, 256 VGPRs each.  At 768 threads (168 VGPRs) the evaluator's lane state spilled (752 bytes of scratch per
// lane).  (The emulated tests: 256 = ONE team; the emulator runs every block of this kernel at the same time.)
#define COOK_MW_THREADS COOK_SHAPE(512, 256)
#endif
constexpr int MW_THREADS = COOK_MW_THREADS;
constexpr int MW_WAVES = MW_THREADS / COOK_WAVE;
constexpr unsigned MW_MAX_POOLS = 32;
constexpr unsigned MW_DONE = 0x3FFFFFFFu;  // (30 bits: see world_pack)
constexpr unsigned long long MW_TIMEOUT_TICKS = 300000000ull;  // 3 s of the 100 MHz clock

// What a walker publishes is ONE 64-bit word per pool — phase and window together, so that a reader can never pair the phase of
// one round with the window of another (a worker wave that has nothing to do in a round may look at it arbitrarily late):
//   bits 63..34 phase: 2r-1 = evaluate the window of round r, 2r = merge it, MW_DONE = the pool is finished
//   bits 33..22 window size - 1, bits 21..0 head (first unresolved job)  -> K < 2^22, windows <= 4096 (the host checks)
static __host__ __device__ __forceinline__ unsigned long long world_pack(unsigned phase, unsigned head, unsigned wcur) {
  return ((unsigned long long)(phase & 0x3FFFFFFFu) << 34) | ((unsigned long long)((wcur - 1u) & 4095u) << 22) | (unsigned long long)(head & 0x3FFFFFu);
}
static __host__ __device__ __forceinline__ unsigned world_phase(unsigned long long pb) { return (unsigned)(pb >> 34); }
struct WorldPool {  // per pool, 128 bytes: completion counters of two pools never share a line
  unsigned done;    // items completed over ALL phases so far, cumulative (evaluators -> walker): never reset, so no reset can race an add
  unsigned pad[31];
};
struct WorldCtl {
  unsigned long long pub[MW_MAX_POOLS];  // the published words of all pools, side by side: one poll reads them all
  unsigned error;      // != 0: give up (1 a wait timed out)
  unsigned n_pools, n_eval_wg;
  unsigned pad;
  unsigned long long t_wait_eval, t_wait_merge;  // ticks pool 0's walker spent waiting for the two phases (statistics)
#ifdef COOK_WORLD_PROF  // measurement build (100 MHz ticks): [phase 0 eval / 1 merge][0 items, 1 sum notice delay, 2 max notice delay,
                        // 3 sum work, 4 max work, 5 sum drain, 6 max (publish -> item finished)]
  unsigned long long t_pub[MW_MAX_POOLS];
  unsigned long long prof[2][8];
#endif
};

constexpr int MW_TEAMS = MW_WAVES / MV_EW;  // teams per evaluator workgroup
static_assert(MW_TEAMS >= 1, "an evaluator workgroup holds at least one team");
struct TeamCtl {
  unsigned cnt, gen;                      // the team's barrier: arrivals of the current generation / generation
  unsigned err, pad;
  unsigned long long view[MW_MAX_POOLS];  // the leader's last poll of WorldCtl::pub: every wave of the team decides from THIS copy
};
struct WorldEvalLds {
  TeamCtl team[MW_TEAMS];
  EvalLds eval[MW_TEAMS];
};
struct WorldLds {
  union {
    ResolveLds r;
    WorldEvalLds e;
  };
  unsigned go;  // walker: this round runs (tid 0 -> workgroup)
};

// barrier of the MV_EW waves of a team through LDS (waves of one workgroup see each other's LDS operations in order)
static __device__ __forceinline__ void team_sync(TeamCtl& t, unsigned& my_gen) {
  wave_sync();
  if (lane_id() == 0) {
    lds_release();
    if (atomicAdd(&t.cnt, 1u) == (unsigned)MV_EW - 1u) {  // last arrival opens the next generation
      st_wg(&t.cnt, 0u);
      st_wg(&t.gen, my_gen + 1u);
    } else {
      EMU_SITE("world: team barrier");
      while (ld_wg(&t.gen) == my_gen) SPIN_PAUSE_SHORT();
    }
    lds_acquire();
  }
  wave_sync();
  ++my_gen;
}

static __device__ __forceinline__ void world_walker(char* lds, const PoolCtx& c, WorldPool* wp, WorldCtl* wc, unsigned pool) {
  unsigned& s_go = reinterpret_cast<WorldLds*>(lds)->go;
  const unsigned tid = threadIdx.x;
  const unsigned K = c.in.K, C = c.vb.C;
  unsigned round = 0, target = 0;
  unsigned long long w_eval = 0, w_merge = 0;
  for (;;) {
    EMU_SITE("world: walker round top");
    if (tid == 0) {
      unsigned go = 0;
      const unsigned head = c.vb.ctl->head, wcur = c.vb.ctl->wcur;  // written by this workgroup (resolve_round) or the host
      if (head < K && ld_agent(&wc->error) == 0u) {
        ++round;
        const unsigned nwin = (head + wcur < K) ? wcur : K - head;
        const unsigned njg = (nwin + COOK_WAVE - 1) / COOK_WAVE;
        const unsigned ntiles = C * njg;
        // everything the previous round wrote (offer state, results, group chains): released, then the window is published
        agent_release();
#ifdef COOK_WORLD_PROF
        st_agent(&wc->t_pub[pool], (unsigned long long)cook_ticks());
#endif
        st_agent(&wc->pub[pool], world_pack(2u * round - 1u, head, wcur));
        target += ntiles;
        const unsigned long long t0 = cook_ticks();
        bool ok = true;
        while ((int)(ld_agent(&wp->done) - target) < 0) {
          SPIN_PAUSE();
          if (ld_agent(&wc->error) != 0u || cook_ticks() - t0 > MW_TIMEOUT_TICKS) {
            ok = false;
            break;
          }
        }
        const unsigned long long t1 = cook_ticks();
        if (ok) {
#ifdef COOK_WORLD_PROF
          st_agent(&wc->t_pub[pool], (unsigned long long)cook_ticks());
#endif
          st_agent(&wc->pub[pool], world_pack(2u * round, head, wcur));  // the chunk lists were written through by their waves
          target += nwin;
          while ((int)(ld_agent(&wp->done) - target) < 0) {
            SPIN_PAUSE();
            if (ld_agent(&wc->error) != 0u || cook_ticks() - t1 > MW_TIMEOUT_TICKS) {
              ok = false;
              break;
            }
          }
        }
        w_eval += t1 - t0;
        w_merge += cook_ticks() - t1;
        if (ok) {
          go = 1;
          agent_acquire();  // ONE acquire for the workgroup: the merged lists, colbits (and whatever else the waves wrote)
        } else {
          st_agent(&wc->error, 1u);
        }
      }
      s_go = go;
    }
    __syncthreads();
    if (!s_go) break;
    resolve_round<false>(lds, c.st, c.vb);
    __syncthreads();  // the walk is over (the helper waves wait here), its stores are issued
  }
  if (tid == 0) {
    st_agent(&wc->pub[pool], world_pack(MW_DONE, 0u, 1u));
    if (pool == 0) {
      wc->t_wait_eval = w_eval;
      wc->t_wait_merge = w_merge;
    }
  }
}

static __device__ __forceinline__ void world_evaluator(char* lds, const PoolCtx* __restrict__ ctx, WorldPool* wp, WorldCtl* wc, unsigned ewg) {
  WorldEvalLds& L = *reinterpret_cast<WorldEvalLds*>(lds);
  const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const unsigned nP = wc->n_pools, nE = wc->n_eval_wg;
  const unsigned team = w / MV_EW, tw = w % MV_EW;  // my team, my wave in it (0 = leader)
  if (team >= (unsigned)MW_TEAMS) return;            // (a workgroup size that is not a multiple of the team size)
  TeamCtl& T = L.team[team];
  if (tid < (unsigned)MW_TEAMS) {
    L.team[tid].cnt = 0u;
    L.team[tid].gen = 0u;
    L.team[tid].err = 0u;
  }
  __syncthreads();
  char* const elds = reinterpret_cast<char*>(&L.eval[team]);
  const unsigned NT = nE * (unsigned)MW_TEAMS;            // teams of the launch
  const unsigned tid_team = ewg * (unsigned)MW_TEAMS + team;
  unsigned my_gen = 0u;
  unsigned seen = 0u;              // lane p: the last phase of pool p this team has dealt with (the same in all its waves)
  unsigned long long last = 0ull;  // leader, lane p: the last word polled for pool p
  const unsigned long long t0 = cook_ticks();
  auto tsync = [&] { team_sync(T, my_gen); };
  for (;;) {
    EMU_SITE("world: team loop");
    if (tw == 0) {  // the leader polls; its view is what the whole team acts on
      const unsigned long long pb = lane < nP ? ld_agent(&wc->pub[lane]) : world_pack(MW_DONE, 0u, 1u);
      unsigned err = ld_agent(&wc->error);
      if (cook_ticks() - t0 > 20ull * MW_TIMEOUT_TICKS) err = 1u;
      if (__any(lane < nP && pb != last)) {
        agent_acquire();  // before anybody of the team reads what the publishing walker wrote
        last = pb;
      }
      if (lane < nP) T.view[lane] = pb;
      if (lane == 0) T.err = err;
    }
    tsync();
    const unsigned long long pb_l = lane < nP ? T.view[lane] : 0ull;
    const unsigned ph_l = lane < nP ? world_phase(pb_l) : MW_DONE;
    if (T.err != 0u || !__any(lane < nP && ph_l != MW_DONE)) break;  // error, or every pool is finished
    const unsigned long long todo = __ballot(lane < nP && ph_l != seen && ph_l != 0u);
    if (todo == 0ull) {
      tsync();  // the view is not rewritten before everybody has read it
      if (tw == 0) SPIN_PAUSE();
      continue;
    }
    // the pools in rotating order, so that no pool is always served last
    const unsigned rot = tid_team % nP;
    const unsigned long long hi = todo >> rot << rot;
    const unsigned p = (unsigned)__ffsll((unsigned long long)(hi ? hi : todo)) - 1u;
    const unsigned ph = (unsigned)wave_read_lane((int)ph_l, (int)p);
    const unsigned lo = (unsigned)wave_read_lane((int)(unsigned)pb_l, (int)p);
    const unsigned hi32 = (unsigned)wave_read_lane((int)(unsigned)(pb_l >> 32), (int)p);
    tsync();  // (as above)
    if (ph != MW_DONE) {
      const PoolCtx& c = ctx[p];
      // the window comes out of the SAME word as the phase: if this team is late and the round is over, it had no item in it
      // (the walker waits for every item), and the loops below find none
      const unsigned head = lo & 0x3FFFFFu, wcur = (((lo >> 22) | (hi32 << 10)) & 4095u) + 1u;
      const unsigned K = c.in.K;
      const unsigned nwin = head < K ? ((head + wcur < K) ? wcur : K - head) : 0u;
      const unsigned off = (unsigned)(((unsigned long long)p * NT) / nP);  // pools start their item -> team map at different teams
      const unsigned first = (tid_team + NT - off % NT) % NT;
      unsigned count = 0;
#ifdef COOK_WORLD_PROF
      const unsigned long long pt0 = cook_ticks(), ptp = ld_agent(&wc->t_pub[p]);
#endif
      if (ph & 1u) {
        const unsigned ntiles = c.vb.C * ((nwin + COOK_WAVE - 1u) / COOK_WAVE);
        for (unsigned t = first; t < ntiles; t += NT) {
          eval_tile_t<true, true>(elds, c.in, c.st, c.vb, head, wcur, t % c.vb.C, t / c.vb.C, tw, tsync);
          tsync();  // the leader is done with the team's lists before the next tile overwrites them
          ++count;
        }
      } else {
        const unsigned nquads = (nwin + (unsigned)MV_EW - 1u) / (unsigned)MV_EW;  // a team merges MV_EW jobs at a time, one per wave
        for (unsigned q = first; q < nquads; q += NT) {
          const unsigned b = q * (unsigned)MV_EW + tw;
          if (b < nwin) merge_job<true>(c.in, c.vb, head, wcur, b);
          const unsigned left = nwin - q * (unsigned)MV_EW;
          count += left < (unsigned)MV_EW ? left : (unsigned)MV_EW;
        }
      }
      if (count) {
#ifdef COOK_WORLD_PROF
        const unsigned long long pt1 = cook_ticks();
#endif
        drain_stores();  // this wave's records went out write-through: in memory before the team's completion count is
        tsync();
        if (tw == 0 && lane == 0) atomicAdd(&wp[p].done, count);
#ifdef COOK_WORLD_PROF
        if (tw == 0 && lane == 0) {
          const unsigned long long pt2 = cook_ticks();
          unsigned long long* q = wc->prof[(ph & 1u) ? 0 : 1];
          atomicAdd(&q[0], (unsigned long long)count);
          atomicAdd(&q[1], pt0 - ptp);
          atomicMax(&q[2], pt0 - ptp);
          atomicAdd(&q[3], pt1 - pt0);
          atomicMax(&q[4], pt1 - pt0);
          atomicAdd(&q[5], pt2 - pt1);
          atomicMax(&q[6], pt2 - ptp);
        }
#endif
      }
    }
    if (lane == p) seen = ph;
  }
}

__global__ void __launch_bounds__(MW_THREADS) match_world(const PoolCtx* __restrict__ ctx, WorldPool* wp, WorldCtl* wc) {
  COOK_BLOCK_LDS(lds, sizeof(WorldLds));
  const unsigned nP = wc->n_pools;
  if (blockIdx.x < nP)
    world_walker(lds, ctx[blockIdx.x], &wp[blockIdx.x], wc, blockIdx.x);
  else
    world_evaluator(lds, ctx, wp, wc, blockIdx.x - nP);
}

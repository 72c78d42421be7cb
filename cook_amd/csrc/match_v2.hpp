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

// The kernels exist in two LIST SHAPES (merged best-fit entries LM / good-enough entries LG per job):
//   default (12 / 4): launches made for best fit (good-enough-fitness >= 1, the parity setting) — most rounds end because a job's
//                     best-fit list ran out, so the walk's LDS image spends its bytes on that list;
//   v2ge   (8 / 12): launches made for good-enough-fitness < 1 (config.clj:111 ships 0.8) — there the "first offer above the
//                     threshold" list is the one that runs out (four entries lasted ~50 jobs: 812 rounds per C4 pool against 442
//                     with twelve, measured on MI355X), and the best-fit list only matters once nothing clears the threshold.
// Same source, compiled twice: the second time inside namespace v2ge with the two constants changed (the records and the walk's LDS
// image are sized by them).  The host picks the set per match call (engine.hip).
#include "match_v2_body.inc"

#ifndef COOK_V2GE_L
#define COOK_V2GE_L 8    // (tuning builds may set the second shape: per-chunk / merged best-fit entries, good-enough entries)
#define COOK_V2GE_LM 8
#define COOK_V2GE_LG 12
#endif
#pragma push_macro("COOK_MV_L")
#pragma push_macro("COOK_MV_LM")
#pragma push_macro("COOK_MV_LG")
#undef COOK_MV_L
#undef COOK_MV_LM
#undef COOK_MV_LG
#define COOK_MV_L COOK_V2GE_L
#define COOK_MV_LM COOK_V2GE_LM
#define COOK_MV_LG COOK_V2GE_LG
#define COOK_V2_BODY_SECOND
namespace v2ge {
#include "match_v2_body.inc"
}
#undef COOK_V2_BODY_SECOND
#pragma pop_macro("COOK_MV_LG")
#pragma pop_macro("COOK_MV_LM")
#pragma pop_macro("COOK_MV_L")
// A third shape for pools with many offers (BASELINE.json configs[2]: 20 000): there most rounds ended because the window's jobs named
// more DISTINCT candidate offers than the walk's slot table holds (729 of 1 124 rounds at 256 slots); 512 slots with a window of 256
// jobs fit the same LDS (155 KB): 1 124 -> 903 rounds, 218 -> 180 ms on MI355X.  Best fit only.
#pragma push_macro("COOK_MV_WMAX")
#pragma push_macro("COOK_MV_S")
#undef COOK_MV_WMAX
#undef COOK_MV_S
#define COOK_MV_WMAX COOK_SHAPE(256, 64)
#define COOK_MV_S COOK_SHAPE(512, 192)
#define COOK_V2_BODY_SECOND
namespace v2big {
#include "match_v2_body.inc"
}
#undef COOK_V2_BODY_SECOND
#pragma pop_macro("COOK_MV_S")
#pragma pop_macro("COOK_MV_WMAX")
constexpr unsigned V2BIG_MIN_OFFERS = COOK_SHAPE(12288, 450);  // pools with at least that many offers take the v2big shape
// A fourth shape for calls with FEW considerable jobs (config.clj:113 ships fenzo-max-jobs-considered 1000): on a cluster whose offers are
// mostly full every job opens an offer of its own, so a job's list dies with its predecessors' placements and rounds end on an
// exhausted list long before the 64 lanes are used up (20 of 25 rounds at K = 1000).  A short window leaves the LDS image room for
// merged lists of 32 entries and 512 slots.  Best fit only.
#pragma push_macro("COOK_MV_WMAX")
#pragma push_macro("COOK_MV_S")
#pragma push_macro("COOK_MV_LM")
#undef COOK_MV_WMAX
#undef COOK_MV_S
#undef COOK_MV_LM
#define COOK_MV_WMAX COOK_SHAPE(128, 64)
#define COOK_MV_S COOK_SHAPE(512, 192)
#define COOK_MV_LM 32
#define COOK_V2_BODY_SECOND
namespace v2small {
#include "match_v2_body.inc"
}
#undef COOK_V2_BODY_SECOND
#pragma pop_macro("COOK_MV_LM")
#pragma pop_macro("COOK_MV_S")
#pragma pop_macro("COOK_MV_WMAX")
constexpr unsigned V2SMALL_MAX_JOBS = COOK_SHAPE(4096, 150);  // calls with at most that many considerable jobs take the v2small shape
#undef COOK_L_TRUNC
#undef COOK_L_COMPLETE
#if COOK_MV_LM > COOK_MV_L
#define COOK_L_TRUNC() ((cinfo_u & JL_TRUNC) != 0u)
#define COOK_L_COMPLETE() ((cinfo_u & JL_TRUNC) == 0u)
#else
#define COOK_L_TRUNC() (nc == MV_L)
#define COOK_L_COMPLETE() (nc < MV_L)
#endif
static_assert(sizeof(v2ge::V2Buf) == sizeof(V2Buf) && sizeof(v2ge::PoolCtx) == sizeof(PoolCtx) && sizeof(v2ge::WinCtl) == sizeof(WinCtl) &&
                  sizeof(v2big::V2Buf) == sizeof(V2Buf) && sizeof(v2big::WinCtl) == sizeof(WinCtl) && sizeof(v2small::V2Buf) == sizeof(V2Buf) &&
                  sizeof(v2small::PoolCtx) == sizeof(PoolCtx),
              "the shapes share their argument records");
#ifndef COOK_MV_WMAX  // (a study build may shrink the default window below the other shapes': it must not run those calls)
static_assert(v2big::MV_WLONG <= MV_WLONG && v2big::MV_JGL <= MV_JGL && v2ge::MV_WLONG == MV_WLONG && v2small::MV_WLONG <= MV_WLONG && v2small::MV_JGL <= MV_JGL, "the host sizes the buffers for the default shape");
#endif

// emu_prims.hpp — TEST INFRASTRUCTURE: what the primitives of cook_amd/csrc/gpu_prims.hpp mean in the SIMT emulator
// (tests/simt_emu: lanes are fibers, a wave is a rendezvous group, memory is sequentially consistent).  Included by
// cook_amd/csrc/platform.hpp in the emulated build only; never part of libcookmatch.so.
#pragma once

// launch shapes: small ones keep the emulated suite fast and make small tests run MANY rounds / tiles; -DCOOK_EMU_SHIPPED_SHAPES
// builds the emulator with the shipped constants instead (tests/test_parity_emu_shipped.py)
#ifdef COOK_EMU_SHIPPED_SHAPES
#define COOK_SHAPE(gpu, emu) (gpu)
#define COOK_BUILD_NAME "simt-emu test build, shipped launch shapes"
#else
#define COOK_SHAPE(gpu, emu) (emu)
#define COOK_BUILD_NAME "simt-emu test build"
#endif

#define COOK_HAS_ASM_WALK 0
#define COOK_WAVES_PER_SIMD(n)  // (an occupancy request to the GPU compiler: nothing to emulate)

// ---- wave-level rendezvous ---------------------------------------------------------------------------
// On the GPU the 64 lanes of a wave run in lockstep and LDS operations of one wave retire in order, so this is a
// compiler scheduling barrier only.  (tests/simt_emu runs lanes as independent fibers and maps it to a rendezvous.)
static inline void wave_sync() { emu::arrive(emu::wave_group()); }

// agent-scope relaxed accesses for the few words that one wave writes and other waves of the same launch read later
// (placement bookkeeping of job groups): they bypass the per-CU L1 (sc1), see MI355X_MICROARCH.md §visibility.
template <class T>
static inline T ld_agent(const T* p) { return *p; }
template <class T>
static inline void st_agent(T* p, T v) { *p = v; emu::progress(); }

// LDS words one wave writes and another polls (classfit.hpp): lanes are fibers that switch only at rendezvous points and in emu::yield()
template <class T>
static inline T ld_wg(const T* p) { return *p; }
template <class T>
static inline void st_wg(T* p, T v) { *p = v; emu::progress(); }
#define COMPILER_FENCE() ((void)0)
#define SPIN_PAUSE_NEAR() emu::yield()
#define SPIN_PAUSE_IDLE() emu::yield()

// hand-off between workgroups of different launches (the served walkers): memory is sequentially consistent here, and launches run
// one after the other, so nothing ever waits — the served path runs in its STEPPING form (match_v2.hpp)
static inline void agent_release() {}
static inline void agent_acquire() {}
static inline void drain_stores() {}
#define SPIN_PAUSE_FAR() emu::yield()
template <class T>
static inline void st_system(T* p, T v) { *p = v; }

// constant-rate (100 MHz) device clock for in-kernel phase timing
static inline unsigned long long cook_ticks() { return 0ull; }

// Scheduling helpers of the placement walk.  OPAQUE_V hides a value's origin from the compiler (a wave-uniform LDS address would
// otherwise turn the loaded record into scalar registers through v_readfirstlane RIGHT AFTER the load, i.e. a full LDS round
// trip on the critical path instead of a prefetch); wave_uniform_u32 moves a value every lane holds into a scalar register where
// the code wants it (branch conditions).
// WAIT_LDS: an explicit s_waitcnt lgkmcnt(0) inside a RARE branch that reloads a loop-carried register from LDS, so that the
// compiler does not put a conservative full wait in front of the register's use on the common path (where it would also wait
// for the prefetches just issued).  The compiler places waits lazily, right before the first use: for a software pipeline
// that means at the TOP of the next iteration, behind the next prefetches.  An explicit wait at the END of an iteration (when
// the prefetches issued at its top have long arrived) tells it that nothing is pending across the back edge.
#define OPAQUE_V(x) ((void)0)
#define WAIT_LDS() ((void)0)
#define WAIT_LDS_BUT_LAST() ((void)0)
#define WAIT_LDS_BUT_2() ((void)0)
#define WAIT_ALL_MEM() ((void)0)
#define PREFETCH_WORD(sink, ptr) ((void)(sink), (void)(ptr))  // (a cache hint: nothing to emulate)
#define PREFETCH_DRAIN(sink) ((void)(sink))
static inline unsigned wave_uniform_u32(unsigned v) { return v; }
static inline unsigned long long wave_uniform_u64(unsigned long long v) { return v; }
static inline double wave_uniform_f64(double v) { return v; }
template <class T>
static inline T* wave_uniform_ptr(T* p) { return p; }

// ---- wave-wide max of a u64 key / lane reads without going through LDS ------------------------------------------------
// ds_bpermute-based shuffles cost ~100+ cycles of latency each; the placement walk is a dependent chain, so its
// reductions use DPP (row-level VALU data movement) and v_readlane instead.
static inline unsigned wave_max_u32(unsigned x) {
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned y = __shfl_xor(x, d, COOK_WAVE);
    x = y > x ? y : x;
  }
  return x;
}
static inline unsigned long long wave_max_u64(unsigned long long x) {
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long y = __shfl_xor(x, d, COOK_WAVE);
    x = y > x ? y : x;
  }
  return x;
}
static inline void wave_max8_u32(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, unsigned& a4, unsigned& a5, unsigned& a6, unsigned& a7) {
  a0 = wave_max_u32(a0), a1 = wave_max_u32(a1), a2 = wave_max_u32(a2), a3 = wave_max_u32(a3), a4 = wave_max_u32(a4), a5 = wave_max_u32(a5), a6 = wave_max_u32(a6),
  a7 = wave_max_u32(a7);
}
static inline void st_lane0_b32(void* p, unsigned a) {
  if (lane_id() == 0) *(unsigned*)p = a, emu::progress();
}
static inline void st_lane0_b64(void* p, unsigned a, unsigned b) {
  if (lane_id() == 0) ((unsigned*)p)[0] = a, ((unsigned*)p)[1] = b, emu::progress();
}
static inline void st_lane0_b128(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
  if (lane_id() == 0) ((unsigned*)p)[0] = a, ((unsigned*)p)[1] = b, ((unsigned*)p)[2] = c, ((unsigned*)p)[3] = d, emu::progress();
}
static inline void st_mask_b32(void* p, unsigned long long mask, unsigned a) {
  if ((mask >> lane_id()) & 1ull) *(unsigned*)p = a, emu::progress();
}
static inline void st_mask_b64(void* p, unsigned long long mask, unsigned a, unsigned b) {
  if ((mask >> lane_id()) & 1ull) ((unsigned*)p)[0] = a, ((unsigned*)p)[1] = b, emu::progress();
}
static inline void st_mask_b128(void* p, unsigned long long mask, unsigned a, unsigned b, unsigned c, unsigned d) {
  if ((mask >> lane_id()) & 1ull) ((unsigned*)p)[0] = a, ((unsigned*)p)[1] = b, ((unsigned*)p)[2] = c, ((unsigned*)p)[3] = d, emu::progress();
}
static inline unsigned long long cook_ballot(bool x) { return __ballot(x); }
static inline unsigned cook_hw_id() { return 0u; }
static inline void cook_set_prio_high() {}
static inline int wave_read_lane(int v, int src) { return __shfl(v, src, COOK_WAVE); }
static inline float wave_max_f32(float x) {
  for (int d = 32; d >= 1; d >>= 1) {
    const float y = __shfl_xor(x, d, COOK_WAVE);
    x = y > x ? y : x;
  }
  return x;
}

// ---- row-shift steps of a wave scan ---------------------------------------------------------------------------------------------------
// scan_fetch<STEP>(x): the value a Kogge-Stone step combines into this lane, fetched with a DPP move (a few cycles; a ds_bpermute
// shuffle costs ~100): steps 0..3 = the lane 1, 2, 4, 8 places down INSIDE its row of 16; step 4 = lane 15 of the previous row for the
// odd rows; step 5 = lane 31 for the upper half.  Lanes without a source get 0 bits (the identity of the sums scanned with it).
// After the six steps every lane holds the inclusive scan of the wave.
template <int STEP>
static inline int scan_fetch_u32(int x) {
  const unsigned lane = lane_id();
  if (STEP < 4) {
    const int v = __shfl_up(x, 1u << STEP, COOK_WAVE);
    return (lane & 15u) >= (1u << STEP) ? v : 0;
  }
  if (STEP == 4) {
    const int v = __shfl(x, (int)((lane & ~15u) - 1u) & 63, COOK_WAVE);
    return ((lane >> 4) & 1u) ? v : 0;
  }
  const int v = __shfl(x, 31, COOK_WAVE);
  return lane >= 32u ? v : 0;
}
// maximum of x over the lane's HALF of the wave (lanes 0..31 / 32..63), returned to every lane of that half; all 64 lanes active
static inline unsigned half_max_u32(unsigned x) {
  for (int d = 16; d >= 1; d >>= 1) {
    const unsigned y = __shfl_xor(x, d, COOK_WAVE);
    x = y > x ? y : x;
  }
  return x;
}

template <class Rec>
static inline void chunk_store(Rec* dst, const Rec& r, bool, unsigned first = 0u) {
  __builtin_memcpy(reinterpret_cast<char*>(dst) + 16u * first, reinterpret_cast<const char*>(&r) + 16u * first, sizeof(Rec) - 16u * first);
}

// walk statistics of the emulated build (design studies, scripts/study_rounds.py): [0] jobs walked, [1] settled by the shortcut
// (no candidate under S), [2] went through the exact path, [3] won by an offer touched earlier in the round, [4] won by an
// untouched offer (a new touched lane), [5] walked and unmatched, [6] sum of touched lanes at decision time, [7] won by the very
// lane that took the previous walked job, [8] decided by the fast path
inline unsigned long long g_walk_stats[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
inline int g_walk_prev_lane = -1;  // lane that took the previous walked job of the round ([7]: a touched offer won AND it is that lane)
#define WALK_STAT(i, v) do { if (lane == 0) g_walk_stats[i] += (v); } while (0)
#define WALK_STAT_PREV_LANE(i, win_lane, win, nT)                                              \
  do {                                                                                         \
    if (lane == 0) {                                                                           \
      if ((i) == 0) g_walk_prev_lane = -1;                                                     \
      if ((win_lane) >= 0 && (win_lane) == g_walk_prev_lane) g_walk_stats[7] += 1;             \
      g_walk_prev_lane = (win_lane) >= 0 ? (win_lane) : ((win) >= 0 ? (int)(nT) : -1);         \
    }                                                                                          \
  } while (0)
// exported by the emulated library only: the walk statistics, cumulative; reset != 0 clears them afterwards
#define COOK_EMU_EXTRA_EXPORTS                                               \
  int cook_emu_walk_stats(unsigned long long out[12], int reset) {           \
    for (int i = 0; i < 12; ++i) out[i] = g_walk_stats[i];                   \
    if (reset)                                                               \
      for (int i = 0; i < 12; ++i) g_walk_stats[i] = 0;                      \
    return COOK_OK;                                                          \
  }

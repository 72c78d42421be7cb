// common.hpp — shared device/host helpers of libcookmatch (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#define COOK_WAVE 64
#ifndef __HIP_EMU__
#define EMU_SITE(s) ((void)0)  // deadlock diagnostics of the SIMT emulator (tests/simt_emu); nothing on the GPU
#endif

// ---- wave-level rendezvous ---------------------------------------------------------------------------
// On the GPU the 64 lanes of a wave run in lockstep and LDS operations of one wave retire in order, so this is a
// compiler scheduling barrier only.  (tests/simt_emu runs lanes as independent fibers and maps it to a rendezvous.)
#ifdef __HIP_EMU__
static inline void wave_sync() { emu::arrive(emu::wave_group()); }
#else
static __device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
#endif

// agent-scope relaxed accesses for the few words that one wave writes and other waves of the same launch read later
// (placement bookkeeping of job groups): they bypass the per-CU L1 (sc1), see MI355X_MICROARCH.md §visibility.
#ifdef __HIP_EMU__
template <class T>
static inline T ld_agent(const T* p) { return *p; }
template <class T>
static inline void st_agent(T* p, T v) { *p = v; emu::progress(); }
#else
template <class T>
static __device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
static __device__ __forceinline__ void st_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// ---- cross-workgroup hand-off inside one launch (the persistent placement kernel, match_world.hpp) ---------------------------------
// The tested forms of MI355X_MICROARCH.md: producer = plain stores -> agent_release() -> relaxed agent-scope flag store;
// consumer = relaxed poll of the flag -> ONE agent_acquire() -> plain loads.  The inline-asm wait is deliberate: ROCm 7.2 drops the
// s_waitcnt after buffer_wbl2 when it can prove the wave's vmcnt scoreboard empty, and the flag then overtakes the write-back.
#ifdef __HIP_EMU__
static inline void agent_release() {}
static inline void agent_acquire() {}
static inline void drain_stores() {}
template <class T>
static inline T ld_wg(const T* p) { return *p; }
template <class T>
static inline void st_wg(T* p, T v) { *p = v; emu::progress(); }
#define SPIN_PAUSE() emu::yield()
#define SPIN_PAUSE_SHORT() emu::yield()
static inline void lds_release() {}
static inline void lds_acquire() {}
#define COOK_BLOCK_LDS(name, bytes) char* name = emu::block_lds(bytes)
#define COOK_LAUNCH_COOP(kernel, grid, block, stream, ...) emuLaunchCoop(kernel, grid, block, __VA_ARGS__)
#else
static __device__ __forceinline__ void agent_release() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
static __device__ __forceinline__ void agent_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// after write-through (sc1) stores: once the wave's store counter drains they are in memory — no L2 write-back fence needed
static __device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// LDS words that waves of one workgroup exchange WITHOUT a barrier (the poller's mirror of the phase words)
template <class T>
static __device__ __forceinline__ T ld_wg(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <class T>
static __device__ __forceinline__ void st_wg(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define SPIN_PAUSE() __builtin_amdgcn_s_sleep(2)
#define SPIN_PAUSE_SHORT() __builtin_amdgcn_s_sleep(1)
// LDS hand-off between waves of one workgroup without a workgroup barrier (the evaluator teams' barrier)
static __device__ __forceinline__ void lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
static __device__ __forceinline__ void lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
#define COOK_BLOCK_LDS(name, bytes) __shared__ __attribute__((aligned(16))) char name[bytes]
// every workgroup of the grid must be resident at once; the host sizes the grid for that (one workgroup per CU)
#define COOK_LAUNCH_COOP(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif

// constant-rate (100 MHz) device clock for in-kernel phase timing
#ifdef __HIP_EMU__
static inline unsigned long long cook_ticks() { return 0ull; }
#else
static __device__ __forceinline__ unsigned long long cook_ticks() { return wall_clock64(); }
#endif

// Scheduling helpers of the placement walk.  OPAQUE_V hides a value's origin from the compiler (a wave-uniform LDS address would
// otherwise turn the loaded record into scalar registers through v_readfirstlane RIGHT AFTER the load, i.e. a full LDS round
// trip on the critical path instead of a prefetch); wave_uniform_u32 moves a value every lane holds into a scalar register where
// the code wants it (branch conditions).
// WAIT_LDS: an explicit s_waitcnt lgkmcnt(0) inside a RARE branch that reloads a loop-carried register from LDS, so that the
// compiler does not put a conservative full wait in front of the register's use on the common path (where it would also wait
// for the prefetches just issued).  The compiler places waits lazily, right before the first use: for a software pipeline
// that means at the TOP of the next iteration, behind the next prefetches.  An explicit wait at the END of an iteration (when
// the prefetches issued at its top have long arrived) tells it that nothing is pending across the back edge.
#ifdef __HIP_EMU__
#define OPAQUE_V(x) ((void)0)
#define WAIT_LDS() ((void)0)
#define WAIT_LDS_BUT_LAST() ((void)0)
#define WAIT_ALL_MEM() ((void)0)
static inline unsigned wave_uniform_u32(unsigned v) { return v; }
#else
#define OPAQUE_V(x) asm volatile("" : "+v"(x))
#define WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F)
#define WAIT_LDS_BUT_LAST() __builtin_amdgcn_s_waitcnt(0xC17F)  // lgkmcnt(1): LDS operations retire in order, the newest may still fly
#define WAIT_ALL_MEM() __builtin_amdgcn_s_waitcnt(0x0070)     // vmcnt(0) lgkmcnt(0)
static __device__ __forceinline__ unsigned wave_uniform_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
#endif

static __device__ __forceinline__ unsigned lane_id() { return threadIdx.x & (COOK_WAVE - 1); }
static __device__ __forceinline__ unsigned wave_id() { return threadIdx.x >> 6; }

static __device__ __forceinline__ unsigned long long lanemask_lt() {
  const unsigned l = lane_id();
  return l == 0 ? 0ull : (~0ull >> (64 - l));
}

// ---- wave-wide max of a u64 key / lane reads without going through LDS ------------------------------------------------
// ds_bpermute-based shuffles cost ~100+ cycles of latency each; the placement walk is a dependent chain, so its
// reductions use DPP (row-level VALU data movement) and v_readlane instead.
#ifdef __HIP_EMU__
static inline unsigned long long wave_max_u64(unsigned long long x) {
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long y = __shfl_xor(x, d, COOK_WAVE);
    x = y > x ? y : x;
  }
  return x;
}
static inline int wave_read_lane(int v, int src) { return __shfl(v, src, COOK_WAVE); }
static inline float wave_max_f32(float x) {
  for (int d = 32; d >= 1; d >>= 1) {
    const float y = __shfl_xor(x, d, COOK_WAVE);
    x = y > x ? y : x;
  }
  return x;
}
#else
// all 64 lanes must be active.  A u64 max has no DPP form (each step = two DPP moves, a 64-bit compare and two selects: 54
// instructions on the walk's critical path); a u32 max does (v_max_u32 with a DPP source).  So: the maximum of the high words
// first, then — among the lanes that hold it — of the low words (read from the one lane when the high word is unique).
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ unsigned dpp_max_u32(unsigned x) {
  const unsigned y = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xF, false);
  return y > x ? y : x;
}
static __device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
  x = dpp_max_u32<0xB1, 0xF>(x);   // quad_perm [1,0,3,2]
  x = dpp_max_u32<0x4E, 0xF>(x);   // quad_perm [2,3,0,1]
  x = dpp_max_u32<0x141, 0xF>(x);  // row_half_mirror
  x = dpp_max_u32<0x140, 0xF>(x);  // row_mirror
  x = dpp_max_u32<0x142, 0xA>(x);  // row_bcast:15 into rows 1 and 3
  x = dpp_max_u32<0x143, 0xC>(x);  // row_bcast:31 into rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
static __device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long x) {
  const unsigned hi = (unsigned)(x >> 32), lo = (unsigned)x;
  const unsigned mh = wave_max_u32(hi);
  const unsigned long long top = __ballot(hi == mh);
  unsigned ml;
  if ((top & (top - 1ull)) == 0ull)  // wave-uniform: one lane holds the greatest high word
    ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)top) - 1));
  else
    ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | (unsigned long long)ml;
}
// Wave-wide maximum of non-negative floats (all 64 lanes active).  Hand-placed: one fused v_max_f32 with a DPP source per
// step and the two wait states a DPP read of a just-written VGPR needs — the compiler's form (copy, nop, v_mov_dpp, v_max per
// step) measured 166 cycles for the six steps on MI355X, against ~25 per step here (scripts/ubench_wave.hip).
static __device__ __forceinline__ float wave_max_f32(float x) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      : "+v"(x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
// value of v in lane src; src must be wave-uniform
static __device__ __forceinline__ int wave_read_lane(int v, int src) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}
#endif

static __device__ __forceinline__ double wave_read_lane_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)wave_read_lane((int)(unsigned)(unsigned long long)b, src);
  const unsigned hi = (unsigned)wave_read_lane((int)(unsigned)((unsigned long long)b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}

// order-preserving map fp64 -> u64 (full 64 bits: DRUs of magnitude 1e-305 must still order, share.clj:95)
static __host__ __device__ __forceinline__ uint64_t f64_key(double d) {
  uint64_t b;
#ifdef __HIP_DEVICE_COMPILE__
  b = (uint64_t)__double_as_longlong(d);
#else
  __builtin_memcpy(&b, &d, 8);
#endif
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
static __host__ __device__ __forceinline__ uint64_t i64_key(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }

// ---- exact-sum tracking ---------------------------------------------------------------------------------
// The reference accumulates fp64 usage left-to-right (dru.clj:43-48 reductions / merge-with +).  A parallel scan
// associates differently; it is bit-identical to the sequential sum iff every partial sum it forms is exact.
// two_sum_err returns the rounding error of a+b (Knuth TwoSum): 0.0 <=> the addition was exact.
static __host__ __device__ __forceinline__ double two_sum_err(double a, double b, double s) {
  const double bb = s - a;
  return (a - (s - bb)) + (b - bb);
}

struct Usage4 {  // {count, cpus, mem, gpus} (tools.clj:883-889 job->usage)
  double count, cpus, mem, gpus;
};

static __host__ __device__ __forceinline__ bool below_quota4(double qc, double qcpus, double qmem, double qgpus, const Usage4& u) {
  return u.count <= qc && u.cpus <= qcpus && u.mem <= qmem && u.gpus <= qgpus;  // tools.clj:876-881
}

template <class T>
static __device__ __forceinline__ T shfl_up_t(T v, unsigned d);
template <>
__device__ __forceinline__ double shfl_up_t<double>(double v, unsigned d) { return __shfl_up(v, d, COOK_WAVE); }
template <>
__device__ __forceinline__ int shfl_up_t<int>(int v, unsigned d) { return __shfl_up(v, d, COOK_WAVE); }
template <>
__device__ __forceinline__ unsigned shfl_up_t<unsigned>(unsigned v, unsigned d) { return __shfl_up(v, d, COOK_WAVE); }

// ---- host side ------------------------------------------------------------------------------------------
#define COOK_HIP(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) {                                                                              \
      char _b[512];                                                                                      \
      std::snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      throw cook_error(COOK_E_DEVICE, _b);                                                               \
    }                                                                                                    \
  } while (0)

struct cook_error {
  int code;
  std::string msg;
  cook_error(int c, std::string m) : code(c), msg(std::move(m)) {}
};

static inline unsigned div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

// (get model->count model 0) / (count model->count) over a host's k8s "gpus" or "disk" map (constraints.clj:136-142, 178): the
// map is the row's non-empty slots; a job without a model asks for key nil, which no map holds
static __device__ __forceinline__ double map_get_dev(const uint32_t* __restrict__ keys, const double* __restrict__ vals, unsigned slots,
                                                     unsigned v, unsigned key) {
  double r = 0.0;
  if (keys && vals && key != 0u)
    for (unsigned s = 0; s < slots; ++s)
      if (keys[(size_t)v * slots + s] == key) r = vals[(size_t)v * slots + s];
  return r;
}
static __device__ __forceinline__ unsigned map_count_dev(const uint32_t* __restrict__ keys, unsigned slots, unsigned v) {
  unsigned n = 0;
  if (keys)
    for (unsigned s = 0; s < slots; ++s) n += keys[(size_t)v * slots + s] != 0u ? 1u : 0u;
  return n;
}

// ---- row-shift steps of a wave scan ---------------------------------------------------------------------------------------------------
// scan_fetch<STEP>(x): the value a Kogge-Stone step combines into this lane, fetched with a DPP move (a few cycles; a ds_bpermute
// shuffle costs ~100): steps 0..3 = the lane 1, 2, 4, 8 places down INSIDE its row of 16; step 4 = lane 15 of the previous row for the
// odd rows; step 5 = lane 31 for the upper half.  Lanes without a source get 0 bits (the identity of the sums scanned with it).
// After the six steps every lane holds the inclusive scan of the wave.
#ifdef __HIP_EMU__
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
#else
template <int STEP>
static __device__ __forceinline__ int scan_fetch_u32(int x) {
  if (STEP == 0) return __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
  if (STEP == 1) return __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
  if (STEP == 2) return __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
  if (STEP == 3) return __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
  if (STEP == 4) return __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  return __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);                 // row_bcast:31 into rows 2 and 3
}
#endif
template <int STEP>
static __device__ __forceinline__ double scan_fetch_f64(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)scan_fetch_u32<STEP>((int)(unsigned)(unsigned long long)b);
  const unsigned hi = (unsigned)scan_fetch_u32<STEP>((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}

// maximum of x over the lane's HALF of the wave (lanes 0..31 / 32..63), returned to every lane of that half; all 64 lanes active
#ifdef __HIP_EMU__
static inline unsigned half_max_u32(unsigned x) {
  for (int d = 16; d >= 1; d >>= 1) {
    const unsigned y = __shfl_xor(x, d, COOK_WAVE);
    x = y > x ? y : x;
  }
  return x;
}
#else
static __device__ __forceinline__ unsigned half_max_u32(unsigned x) {
  x = dpp_max_u32<0xB1, 0xF>(x);   // quad_perm [1,0,3,2]
  x = dpp_max_u32<0x4E, 0xF>(x);   // quad_perm [2,3,0,1]
  x = dpp_max_u32<0x141, 0xF>(x);  // row_half_mirror
  x = dpp_max_u32<0x140, 0xF>(x);  // row_mirror: every lane holds its row's maximum
  x = dpp_max_u32<0x142, 0xA>(x);  // row_bcast:15 into rows 1 and 3: lanes 31 / 63 hold their half's
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)x, 31), hi = (unsigned)__builtin_amdgcn_readlane((int)x, 63);
  return lane_id() < 32u ? lo : hi;
}
#endif
static __device__ __forceinline__ unsigned long long half_max_u64(unsigned long long x) {
  const unsigned hi = (unsigned)(x >> 32), lo = (unsigned)x;
  const unsigned mh = half_max_u32(hi);
  const unsigned ml = half_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | (unsigned long long)ml;
}


// gpu_prims.hpp — the gfx950 (CDNA4, wave64) forms of the primitives the kernels are written against: wave rendezvous, agent- and
// workgroup-scope accesses, DPP reductions and row shifts, scheduling hints, and the launch shapes of
// the shipped build.  Included through platform.hpp, which is the only place that knows about the emulated test build.
#pragma once

#define EMU_SITE(s) ((void)0)  // deadlock diagnostics of the SIMT emulator (tests/simt_emu); nothing on the GPU
#define COOK_SHAPE(gpu, emu) (gpu)  // a launch shape: the shipped value (the emulated build may substitute a small one)
#define COOK_BUILD_NAME "hip gfx950"
#define COOK_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))  // a kernel compiled for exactly n waves per SIMD
// walk statistics are an emulated-build facility (design studies)
#define WALK_STAT(i, v) ((void)0)
#define WALK_STAT_PREV_LANE(i, win_lane, win, nT) ((void)0)
#define COOK_EMU_EXTRA_EXPORTS
#define COOK_HAS_ASM_WALK 1  // classfit_asm.hpp: the hand-placed form of the class-ordered walk's plain step (the emulated build runs the C++ step only)

// ---- wave-level rendezvous ---------------------------------------------------------------------------
// On the GPU the 64 lanes of a wave run in lockstep and LDS operations of one wave retire in order, so this is a
// compiler scheduling barrier only.  (tests/simt_emu runs lanes as independent fibers and maps it to a rendezvous.)
static __device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// agent-scope relaxed accesses for the few words that one wave writes and other waves of the same launch read later
// (placement bookkeeping of job groups): they bypass the per-CU L1 (sc1), see MI355X_MICROARCH.md §visibility.
template <class T>
static __device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
static __device__ __forceinline__ void st_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// words in LDS that one wave of a workgroup writes and another polls (classfit.hpp: the candidate board, commands, the verdict log).  Relaxed accesses the
// compiler neither removes nor moves across COMPILER_FENCE(); the LDS executes one wave's operations in program order, so a reader that finds the
// tag a writer stored LAST finds the fields the writer stored before it.
template <class T>
static __device__ __forceinline__ T ld_wg(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <class T>
static __device__ __forceinline__ void st_wg(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define COMPILER_FENCE() asm volatile("" ::: "memory")
#define SPIN_PAUSE_NEAR() __builtin_amdgcn_s_sleep(1)  // between polls of an LDS word
#define SPIN_PAUSE_IDLE() __builtin_amdgcn_s_sleep(16) // ... by a wave nobody waits for (its polls take issue slots and LDS cycles from the waves that are)

// ---- hand-off between workgroups of DIFFERENT launches that run side by side (the served walkers, match_v2.hpp) ----------------------
// The tested forms of MI355X_MICROARCH.md: producer = plain stores -> agent_release() -> relaxed agent-scope flag store; consumer =
// relaxed poll of the flag -> ONE agent_acquire() -> plain loads.  The inline-asm wait is deliberate: ROCm 7.2 drops the s_waitcnt
// after buffer_wbl2 when it can prove the wave's vmcnt scoreboard empty, and the flag then overtakes the write-back.
static __device__ __forceinline__ void agent_release() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
static __device__ __forceinline__ void agent_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// Every wave of a workgroup, BEFORE the barrier that precedes one thread's agent_release(): the wave's own stores have been acknowledged.
// __syncthreads() waits for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier — seen in the ISA), so another wave's global stores can still
// be in flight when the releasing thread's buffer_wbl2 runs; they then land in the XCD's L2 BEHIND the write-back and stay there.  Under
// light load nobody notices; with four serve streams at once the walkers read stale list entries (profiles/r05c_probe.txt).
static __device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#define SPIN_PAUSE_FAR() __builtin_amdgcn_s_sleep(8)  // between polls of a word in global memory that another workgroup writes
// a word in page-locked HOST memory that the host polls
template <class T>
static __device__ __forceinline__ void st_system(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// constant-rate (100 MHz) device clock for in-kernel phase timing
static __device__ __forceinline__ unsigned long long cook_ticks() { return wall_clock64(); }

// Scheduling helpers of the placement walk.  OPAQUE_V hides a value's origin from the compiler (a wave-uniform LDS address would
// otherwise turn the loaded record into scalar registers through v_readfirstlane RIGHT AFTER the load, i.e. a full LDS round
// trip on the critical path instead of a prefetch); wave_uniform_u32 moves a value every lane holds into a scalar register where
// the code wants it (branch conditions).
// WAIT_LDS: an explicit s_waitcnt lgkmcnt(0) inside a RARE branch that reloads a loop-carried register from LDS, so that the
// compiler does not put a conservative full wait in front of the register's use on the common path (where it would also wait
// for the prefetches just issued).  The compiler places waits lazily, right before the first use: for a software pipeline
// that means at the TOP of the next iteration, behind the next prefetches.  An explicit wait at the END of an iteration (when
// the prefetches issued at its top have long arrived) tells it that nothing is pending across the back edge.
#define OPAQUE_V(x) asm volatile("" : "+v"(x))
#define WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F)
#define WAIT_LDS_BUT_LAST() __builtin_amdgcn_s_waitcnt(0xC17F)  // lgkmcnt(1): LDS operations retire in order, the newest may still fly
#define WAIT_LDS_BUT_2() __builtin_amdgcn_s_waitcnt(0xC27F)     // lgkmcnt(2)
#define WAIT_ALL_MEM() __builtin_amdgcn_s_waitcnt(0x0070)     // vmcnt(0) lgkmcnt(0)
// Fire-and-forget load of one word: pulls the word's cache line towards the CU (L2 of its XCD, L1) for a LATER reader and waits for
// nothing.  The compiler does not track a load issued by inline asm, so the destination is a register the caller keeps for nothing else
// (`sink`, read-write in every use: one live range, one physical register) until PREFETCH_DRAIN has waited for the loads.
#define PREFETCH_WORD(sink, ptr) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(ptr) : "memory")
#define PREFETCH_DRAIN(sink) asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory")
static __device__ __forceinline__ unsigned wave_uniform_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
static __device__ __forceinline__ unsigned long long wave_uniform_u64(unsigned long long v) {
  return ((unsigned long long)wave_uniform_u32((unsigned)(v >> 32)) << 32) | (unsigned long long)wave_uniform_u32((unsigned)v);
}
static __device__ __forceinline__ double wave_uniform_f64(double v) { return __longlong_as_double((long long)wave_uniform_u64((unsigned long long)__double_as_longlong(v))); }
template <class T>
static __device__ __forceinline__ T* wave_uniform_ptr(T* p) { return reinterpret_cast<T*>(wave_uniform_u64(reinterpret_cast<unsigned long long>(p))); }

// ---- wave-wide max of a u64 key / lane reads without going through LDS ------------------------------------------------
// ds_bpermute-based shuffles cost ~100+ cycles of latency each; the placement walk is a dependent chain, so its
// reductions use DPP (row-level VALU data movement) and v_readlane instead.
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
// Eight wave-wide maxima at once (all 64 lanes active): the six DPP steps of eight independent values interleaved, so that no step waits
// for its own predecessor (classfit_walk.hpp: the level summaries of a chunk).  Every lane of the result holds nothing useful but lane 63;
// the values come back wave-uniform.
static __device__ __forceinline__ void wave_max8_u32(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, unsigned& a4, unsigned& a5, unsigned& a6, unsigned& a7) {
#define COOK_M8(ctrl)                                   \
  "v_max_u32_dpp %0, %0, %0 " ctrl "\n\t"              \
  "v_max_u32_dpp %1, %1, %1 " ctrl "\n\t"              \
  "v_max_u32_dpp %2, %2, %2 " ctrl "\n\t"              \
  "v_max_u32_dpp %3, %3, %3 " ctrl "\n\t"              \
  "v_max_u32_dpp %4, %4, %4 " ctrl "\n\t"              \
  "v_max_u32_dpp %5, %5, %5 " ctrl "\n\t"              \
  "v_max_u32_dpp %6, %6, %6 " ctrl "\n\t"              \
  "v_max_u32_dpp %7, %7, %7 " ctrl "\n\t"
  asm volatile("s_nop 1\n\t" COOK_M8("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") COOK_M8("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                   COOK_M8("row_half_mirror row_mask:0xf bank_mask:0xf") COOK_M8("row_mirror row_mask:0xf bank_mask:0xf")
                       COOK_M8("row_bcast:15 row_mask:0xa bank_mask:0xf") COOK_M8("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1\n\t"
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#undef COOK_M8
  a0 = (unsigned)__builtin_amdgcn_readlane((int)a0, 63), a1 = (unsigned)__builtin_amdgcn_readlane((int)a1, 63), a2 = (unsigned)__builtin_amdgcn_readlane((int)a2, 63),
  a3 = (unsigned)__builtin_amdgcn_readlane((int)a3, 63), a4 = (unsigned)__builtin_amdgcn_readlane((int)a4, 63), a5 = (unsigned)__builtin_amdgcn_readlane((int)a5, 63),
  a6 = (unsigned)__builtin_amdgcn_readlane((int)a6, 63), a7 = (unsigned)__builtin_amdgcn_readlane((int)a7, 63);
}
// the wave's hardware placement (HW_REG_HW_ID: bits 4..5 the SIMD of the CU), for the profile notes
static __device__ __forceinline__ unsigned cook_hw_id() { return (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
static __device__ __forceinline__ void cook_set_prio_high() { __builtin_amdgcn_s_setprio(3); }
// One lane's store of wave-uniform words to LDS WITHOUT a vector compare: `if (lane == 0) *p = v` costs a v_cmp, an exec save and a branch on it (a
// VALU -> SALU -> branch hop of 50-80 cycles on a walk's critical path); here the exec mask is set from a constant.  All 64 lanes must be active.
typedef __attribute__((address_space(3))) char* cook_lds_ptr;
static __device__ __forceinline__ unsigned cook_lds_off(const void* p) { return (unsigned)(__SIZE_TYPE__)(cook_lds_ptr)(p); }
static __device__ __forceinline__ void st_lane0_b32(void* p, unsigned a) {
  unsigned long long sv;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(sv) : "v"(cook_lds_off(p)), "v"(a) : "memory");
}
static __device__ __forceinline__ void st_lane0_b64(void* p, unsigned a, unsigned b) {
  unsigned long long sv;
  const unsigned long long v = (unsigned long long)b << 32 | a;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b64 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(sv) : "v"(cook_lds_off(p)), "v"(v) : "memory");
}
static __device__ __forceinline__ void st_lane0_b128(void* p, unsigned a, unsigned b, unsigned c, unsigned d) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  unsigned long long sv;
  const u32x4 v = {a, b, c, d};
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b128 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(sv) : "v"(cook_lds_off(p)), "v"(v) : "memory");
}
// the lanes of `mask` (wave-uniform, non-empty) store their own words to `p` (wave-uniform); all 64 lanes must be active
static __device__ __forceinline__ void st_mask_b32(void* p, unsigned long long mask, unsigned a) {
  unsigned long long sv;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(sv) : "v"(cook_lds_off(p)), "v"(a), "s"(mask) : "memory");
}
static __device__ __forceinline__ void st_mask_b64(void* p, unsigned long long mask, unsigned a, unsigned b) {
  unsigned long long sv;
  const unsigned long long v = (unsigned long long)b << 32 | a;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\tds_write_b64 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(sv) : "v"(cook_lds_off(p)), "v"(v), "s"(mask) : "memory");
}
static __device__ __forceinline__ void st_mask_b128(void* p, unsigned long long mask, unsigned a, unsigned b, unsigned c, unsigned d) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  unsigned long long sv;
  const u32x4 v = {a, b, c, d};
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\tds_write_b128 %1, %2\n\ts_mov_b64 exec, %0" : "=&s"(sv) : "v"(cook_lds_off(p)), "v"(v), "s"(mask) : "memory");
}
// ballot of a condition every active lane has computed (the builtin folds into the compare that made it; HIP's __ballot goes through a 0 / 1 register)
static __device__ __forceinline__ unsigned long long cook_ballot(bool x) { return __builtin_amdgcn_ballot_w64(x); }
// value of v in lane src; src must be wave-uniform
static __device__ __forceinline__ int wave_read_lane(int v, int src) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}

// ---- row-shift steps of a wave scan ---------------------------------------------------------------------------------------------------
// scan_fetch<STEP>(x): the value a Kogge-Stone step combines into this lane, fetched with a DPP move (a few cycles; a ds_bpermute
// shuffle costs ~100): steps 0..3 = the lane 1, 2, 4, 8 places down INSIDE its row of 16; step 4 = lane 15 of the previous row for the
// odd rows; step 5 = lane 31 for the upper half.  Lanes without a source get 0 bits (the identity of the sums scanned with it).
// After the six steps every lane holds the inclusive scan of the wave.
template <int STEP>
static __device__ __forceinline__ int scan_fetch_u32(int x) {
  if (STEP == 0) return __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
  if (STEP == 1) return __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
  if (STEP == 2) return __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
  if (STEP == 3) return __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
  if (STEP == 4) return __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  return __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);                 // row_bcast:31 into rows 2 and 3
}
// maximum of x over the lane's HALF of the wave (lanes 0..31 / 32..63), returned to every lane of that half; all 64 lanes active
static __device__ __forceinline__ unsigned half_max_u32(unsigned x) {
  x = dpp_max_u32<0xB1, 0xF>(x);   // quad_perm [1,0,3,2]
  x = dpp_max_u32<0x4E, 0xF>(x);   // quad_perm [2,3,0,1]
  x = dpp_max_u32<0x141, 0xF>(x);  // row_half_mirror
  x = dpp_max_u32<0x140, 0xF>(x);  // row_mirror: every lane holds its row's maximum
  x = dpp_max_u32<0x142, 0xA>(x);  // row_bcast:15 into rows 1 and 3: lanes 31 / 63 hold their half's
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)x, 31), hi = (unsigned)__builtin_amdgcn_readlane((int)x, 63);
  return lane_id() < 32u ? lo : hi;
}

// a record of 16-byte pieces to global memory; through: write-through (sc1) stores — the record is in memory, visible to every
// XCD, once the wave's vmcnt drains.  first: the lane's first piece to store (an EMPTY candidate list goes out as its 16-byte
// count piece alone: late in a cycle most (job, chunk) lists are empty, and 128 bytes each made the evaluation write 8x its
// algorithmic bytes)
template <class Rec>
static __device__ __forceinline__ void chunk_store(Rec* dst, const Rec& r, bool through, unsigned first = 0u) {
  static_assert(sizeof(Rec) % 16 == 0, "moved in 16-byte pieces");
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned NP = sizeof(Rec) / 16;
  u32x4 piece[NP];
  __builtin_memcpy(piece, &r, sizeof(Rec));  // (not a pointer cast: the record's fields are doubles and ints)
  u32x4* d = reinterpret_cast<u32x4*>(dst);
#pragma unroll
  for (unsigned x = 0; x < NP; ++x) {
    if (x < first) continue;
    if (through) {
      u32x4* a = d + x;
      // s_nop: a VMEM store of more than 64 bits reads its data registers AFTER issue, and the compiler's hazard recogniser does
      // not see into inline asm — without the wait state it re-used the data registers for the next address (seen in the ISA:
      // v_lshl_add_u64 into v[4:5] right behind a store of v[4:7]) and the records went out corrupted
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(a), "v"(piece[x]) : "memory");
    } else {
      d[x] = piece[x];
    }
  }
}

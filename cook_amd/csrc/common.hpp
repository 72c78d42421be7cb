// common.hpp — shared device/host helpers of libcookmatch (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#define COOK_WAVE 64

// ---- wave-level rendezvous ---------------------------------------------------------------------------
// On the GPU the 64 lanes of a wave run in lockstep and LDS operations of one wave retire in order, so this is a
// compiler scheduling barrier only.  (tests/simt_emu runs lanes as independent fibers and maps it to a rendezvous.)
#ifdef __HIP_EMU__
static inline void wave_sync() { emu::arrive(emu::S().waves[emu::wave()]); }
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
static inline void st_agent(T* p, T v) { *p = v; }
#else
template <class T>
static __device__ __forceinline__ T ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
static __device__ __forceinline__ void st_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// constant-rate (100 MHz) device clock for in-kernel phase timing
#ifdef __HIP_EMU__
static inline unsigned long long cook_ticks() { return 0ull; }
#else
static __device__ __forceinline__ unsigned long long cook_ticks() { return wall_clock64(); }
#endif

static __device__ __forceinline__ unsigned lane_id() { return threadIdx.x & (COOK_WAVE - 1); }
static __device__ __forceinline__ unsigned wave_id() { return threadIdx.x >> 6; }

static __device__ __forceinline__ unsigned long long lanemask_lt() {
  const unsigned l = lane_id();
  return l == 0 ? 0ull : (~0ull >> (64 - l));
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

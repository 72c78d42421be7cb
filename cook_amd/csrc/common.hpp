// common.hpp — shared device/host helpers of libcookmatch (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#define COOK_WAVE 64

static __device__ __forceinline__ unsigned lane_id() { return threadIdx.x & (COOK_WAVE - 1); }

static __device__ __forceinline__ unsigned long long lanemask_lt() {
  const unsigned l = lane_id();
  return l == 0 ? 0ull : (~0ull >> (64 - l));
}

#include "platform.hpp"

// the wave's index in its workgroup, IN A SCALAR REGISTER: threadIdx.x >> 6 is the same in all 64 lanes, but the compiler cannot
// know that, and everything derived from it (which offers / hosts / jobs the wave works on) would be per-lane address arithmetic,
// vector loads and exec-mask loops instead of scalar loads and scalar branches
static __device__ __forceinline__ unsigned wave_id() { return wave_uniform_u32(threadIdx.x >> 6); }

static __device__ __forceinline__ double wave_read_lane_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)wave_read_lane((int)(unsigned)(unsigned long long)b, src);
  const unsigned hi = (unsigned)wave_read_lane((int)(unsigned)((unsigned long long)b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}

static __device__ __forceinline__ unsigned long long wave_read_lane_u64(unsigned long long v, int src) {
  const unsigned lo = (unsigned)wave_read_lane((int)(unsigned)v, src);
  const unsigned hi = (unsigned)wave_read_lane((int)(unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | (unsigned long long)lo;
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

// ---- row-shift steps of a wave scan: scan_fetch_u32<STEP> comes from platform.hpp ---------------------------------------------
template <int STEP>
static __device__ __forceinline__ double scan_fetch_f64(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)scan_fetch_u32<STEP>((int)(unsigned)(unsigned long long)b);
  const unsigned hi = (unsigned)scan_fetch_u32<STEP>((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}

static __device__ __forceinline__ unsigned long long half_max_u64(unsigned long long x) {
  const unsigned hi = (unsigned)(x >> 32), lo = (unsigned)x;
  const unsigned mh = half_max_u32(hi);
  const unsigned ml = half_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | (unsigned long long)ml;
}


// ubench_wave.hip — single-wave latency table for the placement walk's building blocks on gfx950 (measurement tool; not part of
// libcookmatch).  One wave, dependent chains, cycles per step from s_memtime.  Build: hipcc --offload-arch=gfx950 -O3 -o
// scripts/ubench_wave scripts/ubench_wave.hip ; run on the GPU box; prints one JSON object.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
      std::exit(1);                                                               \
    }                                                                             \
  } while (0)

constexpr int NIT = 512;
constexpr int NTEST = 32;

static __device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ unsigned dpp_max_u32(unsigned x) {
  const unsigned y = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xF, false);
  return y > x ? y : x;
}
static __device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
  x = dpp_max_u32<0xB1, 0xF>(x);
  x = dpp_max_u32<0x4E, 0xF>(x);
  x = dpp_max_u32<0x141, 0xF>(x);
  x = dpp_max_u32<0x140, 0xF>(x);
  x = dpp_max_u32<0x142, 0xA>(x);
  x = dpp_max_u32<0x143, 0xC>(x);
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ float dpp_max_f32(float x) {
  const float y = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
  return y > x ? y : x;
}
static __device__ __forceinline__ float wave_max_f32(float x) {
  x = dpp_max_f32<0xB1, 0xF>(x);
  x = dpp_max_f32<0x4E, 0xF>(x);
  x = dpp_max_f32<0x141, 0xF>(x);
  x = dpp_max_f32<0x140, 0xF>(x);
  x = dpp_max_f32<0x142, 0xA>(x);
  x = dpp_max_f32<0x143, 0xC>(x);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
static __device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long x) {
  const unsigned hi = (unsigned)(x >> 32), lo = (unsigned)x;
  const unsigned mh = wave_max_u32(hi);
  const unsigned long long top = __ballot(hi == mh);
  unsigned ml;
  if ((top & (top - 1ull)) == 0ull)
    ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)top) - 1));
  else
    ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | (unsigned long long)ml;
}

#define BEGIN(id)                    \
  {                                  \
    __builtin_amdgcn_s_barrier();    \
    const unsigned long long t0_ = now();
#define END(id, steps)                                               \
    const unsigned long long t1_ = now();                            \
    if (lane == 0) out[id] = (double)(t1_ - t0_) / (double)(steps);  \
  }

__global__ void __launch_bounds__(64) ubench(double* out, double* sink, unsigned* gbuf, unsigned seed) {
  __shared__ unsigned lds32[4096];
  __shared__ __attribute__((aligned(16))) unsigned long long lds64[2048];
  const unsigned lane = threadIdx.x;
  // LDS pointer-chase tables: a permutation cycle over 1024 entries (per lane a different start, same cycle)
  for (unsigned i = lane; i < 4096; i += 64) lds32[i] = (i * 1237u + 331u) & 1023u;
  for (unsigned i = lane; i < 2048; i += 64) lds64[i] = (i * 1237u + 331u) & 1023u;
  __syncthreads();
  double acc = 0.0;
  // 0: empty loop (loop overhead with a trivially dependent integer add)
  {
    unsigned x = seed + lane;
    BEGIN(0)
    for (int i = 0; i < NIT; ++i) x = x * 3u + 1u;
    END(0, NIT)
    acc += x;
  }
  // 1..4: fp64 dependent chains
  {
    double x = 1.0 + seed * 1e-9 + lane;
    BEGIN(1)
    for (int i = 0; i < NIT; ++i) x = x + 1.25;
    END(1, NIT)
    acc += x;
  }
  {
    double x = 1.0 + seed * 1e-9 + lane * 1e-3;
    BEGIN(2)
    for (int i = 0; i < NIT; ++i) x = x * 1.0000001;
    END(2, NIT)
    acc += x;
  }
  {
    double x = 1.5 + seed * 1e-9 + lane * 1e-3;
    BEGIN(3)
    for (int i = 0; i < NIT; ++i) x = 2.0 / x;
    END(3, NIT)
    acc += x;
  }
  {
    double x = 1.5 + seed * 1e-9 + lane * 1e-3;
    BEGIN(4)
    for (int i = 0; i < NIT; ++i) x = __builtin_fma(x, 0.999999, 0.25);
    END(4, NIT)
    acc += x;
  }
  // 5: ds_read_b32 pointer chase
  {
    unsigned x = (seed + lane) & 1023u;
    BEGIN(5)
    for (int i = 0; i < NIT; ++i) x = lds32[x];
    END(5, NIT)
    acc += x;
  }
  // 6: ds_read_b64 pointer chase
  {
    unsigned x = (seed + lane) & 1023u;
    BEGIN(6)
    for (int i = 0; i < NIT; ++i) x = (unsigned)lds64[x];
    END(6, NIT)
    acc += x;
  }
  // 7: ds_read_b128 pointer chase
  {
    unsigned x = (seed + lane) & 1022u;
    BEGIN(7)
    for (int i = 0; i < NIT; ++i) {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds64[x & 1022u]);
      x = (unsigned)(v.x ^ (v.y & 0ull));
    }
    END(7, NIT)
    acc += x;
  }
  // 8: wave max u32 (6 DPP steps + readlane), dependent
  {
    unsigned x = seed * 7u + lane * 2654435761u;
    BEGIN(8)
    for (int i = 0; i < NIT; ++i) x = wave_max_u32(x ^ lane) * 2654435761u + lane;
    END(8, NIT)
    acc += x;
  }
  // 9: wave max u64 as the walk uses it
  {
    unsigned long long x = seed * 7ull + lane * 0x9E3779B97F4A7C15ull;
    BEGIN(9)
    for (int i = 0; i < NIT; ++i) x = wave_max_u64(x ^ lane) * 0x9E3779B97F4A7C15ull + lane;
    END(9, NIT)
    acc += (double)x;
  }
  // 10: wave max f32 through DPP (cvt f64->f32 included)
  {
    double x = 1.0 + lane * 0.001 + seed * 1e-9;
    BEGIN(10)
    for (int i = 0; i < NIT; ++i) {
      const float m = wave_max_f32((float)x);
      x = x * 0.5 + (double)m * 0.25 + lane * 1e-6;
    }
    END(10, NIT)
    acc += x;
  }
  // 11: ballot -> ffs -> readlane (one hop through SGPRs), dependent
  {
    unsigned x = seed + lane * 977u;
    BEGIN(11)
    for (int i = 0; i < NIT; ++i) {
      const unsigned long long m = __ballot((x & 64u) != 0u) | 1ull;
      const int l = __ffsll((unsigned long long)(m >> 1 | 1ull << 63)) - 1;
      x = (unsigned)__builtin_amdgcn_readlane((int)x, __builtin_amdgcn_readfirstlane(l)) * 13u + lane;
    }
    END(11, NIT)
    acc += x;
  }
  // 12: readlane of an f64 (two v_readlane) from a uniform lane index that depends on the previous value
  {
    double x = 1.0 + lane;
    unsigned l = seed & 63u;
    BEGIN(12)
    for (int i = 0; i < NIT; ++i) {
      const long long b = __double_as_longlong(x);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, __builtin_amdgcn_readfirstlane((int)l));
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), __builtin_amdgcn_readfirstlane((int)l));
      const double y = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
      x = y + lane;
      l = (lo >> 3) & 63u;
    }
    END(12, NIT)
    acc += x;
  }
  // 13: LDS write then read of a neighbour's word (one wave: in-order), dependent
  {
    unsigned x = seed + lane;
    BEGIN(13)
    for (int i = 0; i < NIT; ++i) {
      lds32[2048 + lane] = x;
      x = lds32[2048 + ((lane + 1u) & 63u)] + 1u;
    }
    END(13, NIT)
    acc += x;
  }
  // 14: ds_bpermute shuffle, dependent
  {
    unsigned x = seed + lane;
    BEGIN(14)
    for (int i = 0; i < NIT; ++i) x = (unsigned)__shfl((int)x, (int)((x + 1u) & 63u), 64) + 1u;
    END(14, NIT)
    acc += x;
  }
  // 15: global load pointer chase, plain (L1/L2 resident: 4 KB table)
  {
    unsigned x = (seed + lane) & 1023u;
    BEGIN(15)
    for (int i = 0; i < 128; ++i) x = gbuf[x];
    END(15, 128)
    acc += x;
  }
  // 16: global load pointer chase with agent-scope (sc1) loads: L2 latency
  {
    unsigned x = (seed + lane) & 1023u;
    BEGIN(16)
    for (int i = 0; i < 128; ++i) x = __hip_atomic_load(&gbuf[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    END(16, 128)
    acc += x;
  }
  // 17: wave-uniform branch chain: value -> readfirstlane -> s_cmp -> branch
  {
    unsigned x = seed;
    BEGIN(17)
    for (int i = 0; i < NIT; ++i) {
      const unsigned u = (unsigned)__builtin_amdgcn_readfirstlane((int)x);
      if (u & 1u)
        x = x * 3u + 1u;
      else
        x = (x >> 1) + 7u;
      if (u & 2u) x ^= 0x55u;
    }
    END(17, NIT)
    acc += x;
  }
  // 18: v_cmp_f64 + ballot + s_cmp chain (decision from a vector compare)
  {
    double x = 1.0 + lane * 0.01;
    unsigned cnt = 0;
    BEGIN(18)
    for (int i = 0; i < NIT; ++i) {
      const unsigned long long m = __ballot(x > 1.3);
      cnt += (unsigned)__popcll(m);
      x = (m & 1ull) ? x * 0.99 : x * 1.01;
    }
    END(18, NIT)
    acc += x + cnt;
  }
  // 19: the touched-offer evaluation of the walk: fits + approximate fitness (fp64: 2 add/cmp pairs, 2 add, 2 mul, add, mul)
  {
    double ac = lane, am = lane * 100.0, oc = 64.0 + lane, om = 262144.0, basec = 3.0 + lane, basem = 1000.0 + lane;
    const double invc = 1.0 / (oc + 3.0), invm = 1.0 / (om + 7.0);
    double c = 1.0 + (seed & 3), m = 512.0;
    double s = 0.0;
    BEGIN(19)
    for (int i = 0; i < NIT; ++i) {
      const bool ok = !(ac + c > oc || am + m > om);
      const double fa = ((basec + c) * invc + (basem + m) * invm) * 0.5;
      s += ok ? fa : 0.0;
      c = c + fa * 1e-9;  // dependency into the next step
    }
    END(19, NIT)
    acc += s;
  }
  // 20: same evaluation followed by the f32 wave max + near-ballot (the proposed reduction)
  {
    double oc = 64.0 + lane, om = 262144.0, basec = 3.0 + lane, basem = 1000.0 + lane, ac = lane, am = lane * 100.0;
    const double invc = 1.0 / (oc + 3.0), invm = 1.0 / (om + 7.0);
    double c = 1.0 + (seed & 3), m = 512.0;
    unsigned wins = 0;
    BEGIN(20)
    for (int i = 0; i < NIT; ++i) {
      const bool ok = !(ac + c > oc || am + m > om);
      const double fa = ((basec + c) * invc + (basem + m) * invm) * 0.5;
      const float k = ok ? (float)fa : 0.0f;
      const float mx = wave_max_f32(k);
      const unsigned long long near = __ballot(k >= mx * 0.999999f);
      const int wl = __ffsll((unsigned long long)near) - 1;
      wins += (unsigned)wl;
      if ((int)lane == wl) {
        ac += c;
        basec += c;
      }
      c = 1.0 + (double)(wl & 3);
    }
    END(20, NIT)
    acc += wins + ac;
  }
  // 21: same with the u64 max the walk uses today
  {
    double oc = 64.0 + lane, om = 262144.0, basec = 3.0 + lane, basem = 1000.0 + lane, ac = lane, am = lane * 100.0;
    const double invc = 1.0 / (oc + 3.0), invm = 1.0 / (om + 7.0);
    double c = 1.0 + (seed & 3), m = 512.0;
    unsigned wins = 0;
    BEGIN(21)
    for (int i = 0; i < NIT; ++i) {
      const bool ok = !(ac + c > oc || am + m > om);
      const double fa = ((basec + c) * invc + (basem + m) * invm) * 0.5;
      const unsigned long long key = ok ? (unsigned long long)__double_as_longlong(fa) : 0ull;
      const double mx = __longlong_as_double((long long)wave_max_u64(key));
      const unsigned long long near = __ballot(ok && fa >= mx * (1.0 - 0x1p-38));
      const int wl = __ffsll((unsigned long long)near) - 1;
      wins += (unsigned)wl;
      if ((int)lane == wl) {
        ac += c;
        basec += c;
      }
      c = 1.0 + (double)(wl & 3);
    }
    END(21, NIT)
    acc += wins + ac;
  }
  // 22: s_sleep(1) granularity and 23: relaxed agent store + load of the same word (one hop to L2 and back)
  {
    BEGIN(22)
    for (int i = 0; i < 64; ++i) __builtin_amdgcn_s_sleep(1);
    END(22, 64)
  }
  {
    unsigned x = seed;
    BEGIN(23)
    for (int i = 0; i < 64; ++i) {
      if (lane == 0) __hip_atomic_store(&gbuf[1024], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      x = __hip_atomic_load(&gbuf[1024], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    }
    END(23, 64)
    acc += x;
  }
  // 24: 8 independent ds_read_b128 issued together then one wait (prefetch pattern)
  {
    unsigned x = (seed + lane) & 1022u;
    unsigned long long s = 0;
    BEGIN(24)
    for (int i = 0; i < NIT / 8; ++i) {
      ulonglong2 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const ulonglong2*>(&lds64[(x + 2u * q) & 1022u]);
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q].x + v[q].y;
      x = (unsigned)s & 1022u;
    }
    END(24, NIT / 8)
    acc += (double)s;
  }
  sink[lane] = acc;
}

// two workgroups ping-pong a flag through L2 (agent-scope relaxed atomics): one-way hand-off latency between CUs
__global__ void __launch_bounds__(64) pingpong(unsigned* flag, double* out, int iters) {
  const unsigned me = blockIdx.x, lane = threadIdx.x;
  if (lane != 0) return;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    const unsigned want = 2u * i + me;  // block 0 waits for even, writes odd ...
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
      if (wall_clock64() - t0 > 200000000ull) return;  // 2 s guard
    }
    __hip_atomic_store(flag, want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const unsigned long long t1 = wall_clock64();
  if (me == 0) out[0] = (double)(t1 - t0) * 10.0 / (2.0 * iters);  // ns per one-way hop (100 MHz clock)
}

int main() {
  double *out, *sink, *pp;
  unsigned *gbuf, *flag;
  CHECK(hipMalloc(&out, NTEST * sizeof(double)));
  CHECK(hipMalloc(&pp, sizeof(double)));
  CHECK(hipMalloc(&sink, 64 * sizeof(double)));
  CHECK(hipMalloc(&gbuf, 2048 * sizeof(unsigned)));
  CHECK(hipMalloc(&flag, 64));
  std::vector<unsigned> h(2048);
  for (unsigned i = 0; i < 2048; ++i) h[i] = (i * 1237u + 331u) & 1023u;
  CHECK(hipMemcpy(gbuf, h.data(), 2048 * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(out, 0, NTEST * sizeof(double)));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(ubench, dim3(1), dim3(64), 0, 0, out, sink, gbuf, 12345u + rep);
    CHECK(hipDeviceSynchronize());
  }
  std::vector<double> r(NTEST);
  CHECK(hipMemcpy(r.data(), out, NTEST * sizeof(double), hipMemcpyDeviceToHost));
  CHECK(hipMemset(flag, 0, 64));
  CHECK(hipMemset(pp, 0, 8));
  hipLaunchKernelGGL(pingpong, dim3(2), dim3(64), 0, 0, flag, pp, 2000);
  CHECK(hipDeviceSynchronize());
  double hop = 0;
  CHECK(hipMemcpy(&hop, pp, 8, hipMemcpyDeviceToHost));
  const char* names[] = {"int_mad_chain", "f64_add", "f64_mul", "f64_div", "f64_fma", "ds_read_b32_chase", "ds_read_b64_chase",
                         "ds_read_b128_chase", "wave_max_u32_dpp", "wave_max_u64", "cvt+wave_max_f32_dpp", "ballot_ffs_readlane",
                         "readlane_f64_dynamic", "lds_write_read_neighbour", "ds_bpermute", "global_chase_plain", "global_chase_sc1",
                         "uniform_branch_chain", "vcmp_ballot_select", "offer_eval", "offer_eval+f32max+commit", "offer_eval+u64max+commit",
                         "s_sleep_1", "agent_store_load_roundtrip", "8x_ds_read_b128_batch"};
  std::printf("{\"unit\": \"shader cycles per step (one wave, dependent chain)\"");
  for (int i = 0; i < 25; ++i) std::printf(", \"%s\": %.1f", names[i], r[i]);
  std::printf(", \"flag_hop_ns_between_two_workgroups\": %.0f}\n", hop);
  return 0;
}

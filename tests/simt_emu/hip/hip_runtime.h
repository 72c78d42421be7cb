// tests/simt_emu/hip/hip_runtime.h — TEST INFRASTRUCTURE.  A tiny single-process SIMT emulator that stands in
// for <hip/hip_runtime.h> so the SAME .hip sources of cook_amd/csrc can be compiled with g++ and exercised on
// a machine without a GPU (this container).  It is NOT a compatibility layer of the product: libcookmatch.so is
// built by hipcc for gfx950 only; this header exists so that kernel LOGIC (indexing, scans, sort passes, the
// placement loop) is checked against the oracle before GPU minutes are spent.  It does not model memory
// ordering, LDS banking, occupancy or timing.
//
// Model: one OS thread; every GPU thread of a block is a ucontext fiber; blocks of a grid run one after another
// (so kernels must not wait on other blocks) — except for a CO-SCHEDULED launch (emu::launch with coop = true, used for the
// persistent placement kernel), whose blocks all run concurrently, each with its own LDS (emu::block_lds), and whose spin loops
// hand the processor on through emu::yield(); a wave is 64 consecutive threads; wave collectives and __syncthreads are
// rendezvous points at which fibers yield.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __HIP_EMU__ 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __constant__ static

struct alignas(16) uint4 {
  unsigned x, y, z, w;
};
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct emuStream {
  int dummy;
};
struct emuEvent {
  double t;
};
typedef emuStream* hipStream_t;
typedef emuEvent* hipEvent_t;
struct hipDeviceProp_t {
  char name[64];
  int multiProcessorCount;
  size_t totalGlobalMem;
  char gcnArchName[64];
};
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0
#define hipEventDefault 0

namespace emu {
struct Group {
  unsigned size = 0, count = 0;
  uint64_t gen = 0;
};
// Context switches: glibc's swapcontext saves and restores the signal mask = two system calls per switch, and the emulator
// switches at every rendezvous of every thread.  On x86-64 a fiber is just a saved stack pointer (callee-saved registers on its
// stack, emu.cpp); elsewhere ucontext.
#if defined(__x86_64__)
#define EMU_FAST_SWITCH 1
#endif
struct BlockCtx {  // one resident block
  dim3 bidx;
  Group block;
  std::vector<Group> waves;
  std::vector<uint64_t> xchg;  // 64 slots per wave
  std::vector<char> lds;       // emu::block_lds storage (co-scheduled kernels: `static` shared arrays would be shared by all blocks)
};
struct Fiber {
  ucontext_t ctx;
  void* sp = nullptr;  // EMU_FAST_SWITCH: the fiber's saved stack pointer
  char* stack = nullptr;
  bool done = true;
  unsigned tid = 0;
  dim3 tidx;
  BlockCtx* blk = nullptr;
  const char* site = "";  // last EMU_SITE() the fiber passed (deadlock diagnostics)
};
struct State {
  ucontext_t main_ctx;
  void* main_sp = nullptr;
  std::vector<Fiber> fibers;
  Fiber* cur = nullptr;
  dim3 blockDim_, gridDim_;
  std::vector<BlockCtx> blocks;
  const std::function<void()>* body = nullptr;
  uint64_t events = 0;  // rendezvous completions + thread exits + global stores of spin-waited words (progress detector)
};
State& S();
void launch(dim3 grid, dim3 block, const std::function<void()>& body, bool coop = false);
void arrive(Group& g);
void yield();                    // a spin loop hands the processor to the other fibers (co-scheduled launches)
inline void progress() { ++S().events; }
char* block_lds(size_t bytes);   // this block's LDS (same pointer for every thread of the block)
inline BlockCtx& B() { return *S().cur->blk; }
inline unsigned lane() { return S().cur->tid & 63u; }
inline unsigned wave() { return S().cur->tid >> 6; }
inline unsigned wave_size() { return B().waves[wave()].size; }
inline Group& wave_group() { return B().waves[wave()]; }

template <class T>
inline uint64_t bits_of(T v) {
  uint64_t b = 0;
  static_assert(sizeof(T) <= 8, "shfl payload");
  std::memcpy(&b, &v, sizeof(T));
  return b;
}
template <class T>
inline T from_bits(uint64_t b) {
  T v;
  std::memcpy(&v, &b, sizeof(T));
  return v;
}
template <class T>
inline T shfl_idx(T v, int src) {
  BlockCtx& s = B();
  const unsigned w = wave(), l = lane();
  s.xchg[w * 64 + l] = bits_of(v);
  arrive(s.waves[w]);
  const unsigned n = s.waves[w].size;
  T r = (src >= 0 && (unsigned)src < n) ? from_bits<T>(s.xchg[w * 64 + (unsigned)src]) : v;
  arrive(s.waves[w]);
  return r;
}
}  // namespace emu

#define EMU_SITE(s) (emu::S().cur->site = (s))
#define threadIdx (emu::S().cur->tidx)
#define blockIdx (emu::B().bidx)
#define blockDim (emu::S().blockDim_)
#define gridDim (emu::S().gridDim_)
static const int warpSize = 64;

inline void __syncthreads() { emu::arrive(emu::B().block); }
inline void __threadfence() {}
inline void __threadfence_block() {}

template <class T>
inline T __shfl(T v, int src, int width = 64) {
  (void)width;
  return emu::shfl_idx(v, src);
}
template <class T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
  (void)width;
  int l = (int)emu::lane();
  return emu::shfl_idx(v, l - (int)d >= 0 ? l - (int)d : l);
}
template <class T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
  (void)width;
  int l = (int)emu::lane();
  return emu::shfl_idx(v, l + (int)d < 64 ? l + (int)d : l);
}
template <class T>
inline T __shfl_xor(T v, int m, int width = 64) {
  (void)width;
  return emu::shfl_idx(v, (int)emu::lane() ^ m);
}
inline unsigned long long __ballot(int pred) {
  emu::BlockCtx& s = emu::B();
  const unsigned w = emu::wave(), l = emu::lane();
  s.xchg[w * 64 + l] = pred ? 1 : 0;
  emu::arrive(s.waves[w]);
  unsigned long long m = 0;
  for (unsigned i = 0; i < s.waves[w].size; ++i) m |= (unsigned long long)(s.xchg[w * 64 + i] & 1) << i;
  emu::arrive(s.waves[w]);
  return m;
}
inline int __any(int p) { return __ballot(p) != 0; }
inline int __all(int p) {
  unsigned n = emu::wave_size();
  unsigned long long full = n >= 64 ? ~0ull : ((1ull << n) - 1);
  return __ballot(p) == full;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline long long __double_as_longlong(double d) { return emu::from_bits<long long>(emu::bits_of(d)); }
inline double __longlong_as_double(long long v) { return emu::from_bits<double>(emu::bits_of(v)); }
inline unsigned __lane_id() { return emu::lane(); }
inline float __int_as_float(int v) { return emu::from_bits<float>(emu::bits_of(v)); }
inline int __float_as_int(float v) { return emu::from_bits<int>(emu::bits_of(v)); }

template <class T>
inline T atomicAdd(T* p, T v) {
  emu::progress();
  T o = *p;
  *p = o + v;
  return o;
}
template <class T>
inline T atomicMax(T* p, T v) {
  T o = *p;
  if (v > o) *p = v;
  return o;
}
template <class T>
inline T atomicMin(T* p, T v) {
  T o = *p;
  if (v < o) *p = v;
  return o;
}
template <class T>
inline T atomicExch(T* p, T v) {
  T o = *p;
  *p = v;
  return o;
}
template <class T>
inline T atomicCAS(T* p, T cmp, T v) {
  emu::progress();
  T o = *p;
  if (o == cmp) *p = v;
  return o;
}
template <class T>
inline T atomicOr(T* p, T v) {
  T o = *p;
  *p = o | v;
  return o;
}
template <class T>
inline T atomicAnd(T* p, T v) {
  T o = *p;
  *p = o & v;
  return o;
}

// ---- runtime API ------------------------------------------------------------------------------------
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess(emu)" : "hipError(emu)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof(*p));
  std::snprintf(p->name, sizeof(p->name), "simt-emu");
  std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "emu");
  p->multiProcessorCount = 4;
  p->totalGlobalMem = 1ull << 32;
  return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) {
  *p = std::malloc(n ? n : 1);
  if (*p) std::memset(*p, 0xCD, n);  // poison: catches reads of uninitialised device memory
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T>
inline hipError_t hipMalloc(T** p, size_t n) {
  return hipMalloc((void**)p, n);
}
inline hipError_t hipFree(void* p) {
  std::free(p);
  return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) {
  *p = std::malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T>
inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) {
  return hipHostMalloc((void**)p, n, f);
}
inline hipError_t hipHostFree(void* p) {
  std::free(p);
  return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  if (n) std::memmove(d, s, n);
  return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t = nullptr) {
  return hipMemcpy(d, s, n, k);
}
inline hipError_t hipMemset(void* d, int v, size_t n) {
  if (n) std::memset(d, v, n);
  return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { return hipMemset(d, v, n); }
inline hipError_t hipStreamCreate(hipStream_t* s) {
  *s = new emuStream{0};
  return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) {
  *least = 0, *greatest = -1;
  return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t s) {
  delete s;
  return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
double emu_now_ms();
inline hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new emuEvent{0};
  return hipSuccess;
}
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
  e->t = emu_now_ms();
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = (float)(b->t - a->t);
  return hipSuccess;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)              \
  do {                                                                           \
    std::function<void()> _emu_body = [&]() { kernel(__VA_ARGS__); };            \
    emu::launch(dim3(grid), dim3(block), _emu_body);                             \
  } while (0)
// all blocks of the grid resident at once (what the persistent placement kernel needs)
#define emuLaunchCoop(kernel, grid, block, ...)                                  \
  do {                                                                           \
    std::function<void()> _emu_body = [&]() { kernel(__VA_ARGS__); };            \
    emu::launch(dim3(grid), dim3(block), _emu_body, true);                       \
  } while (0)

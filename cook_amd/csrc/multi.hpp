// multi.hpp — one launch for the same kernel of several pools (the rank stage of a GPU that holds more than one pool).
//
// A pool's rank is a chain of ~100 small dependent launches, and at 175k tasks per pool the stage is bound by the NUMBER of launches
// the host has to make (≈ 6 µs each, ≈ 10 with four threads launching at once: DESIGN.md §3), not by their work.  Eight pools on one GPU
// therefore cost eight times the host time of one.  The kernels of that path are written as plain device functions (COOK_KERNEL) over a
// one-dimensional grid; `cook_multi` is the one __global__ entry for all of them: blockIdx.y names the pool, the pool's own argument
// list is read from the kernel arguments with scalar loads (blockIdx.y is uniform), and a pool whose grid is shorter than the launch's
// leaves at once.  blockIdx.x / threadIdx.x mean what they always meant, so a kernel body does not know whether it runs alone.
// The host side (engine.hip "pool batches") records the launches of each pool's flow and issues those of the same kernel together.
#pragma once

// a kernel of the batched path: a device function over a 1-D grid that never reads gridDim (its launch may be wider than its own grid)
#define COOK_KERNEL static __device__ __attribute__((always_inline)) inline

constexpr unsigned COOK_MULTI_MAX = 8;           // pools per launch at most
constexpr unsigned COOK_MULTI_ARG_BYTES = 3968;  // (kernel arguments may take 4 KB)

template <class... A>
struct ArgPack;
template <>
struct ArgPack<> {};
template <class H, class... T>
struct ArgPack<H, T...> {
  H h;
  ArgPack<T...> t;
  // from the values of a launch site, converted implicitly as a call of the kernel would convert them
  template <class X0, class... X>
  static ArgPack make(const X0& x0, const X&... x) {
    static_assert(sizeof...(X) == sizeof...(T), "launch: number of kernel arguments");
    ArgPack p{};
    p.h = x0;
    if constexpr (sizeof...(T) > 0) p.t = ArgPack<T...>::make(x...);
    return p;
  }
};

template <class... A>
struct MultiArgs {
  static constexpr unsigned per_raw = (COOK_MULTI_ARG_BYTES - 4u * COOK_MULTI_MAX) / (unsigned)(sizeof(ArgPack<A...>) ? sizeof(ArgPack<A...>) : 1);
  static constexpr unsigned PER = per_raw < COOK_MULTI_MAX ? per_raw : COOK_MULTI_MAX;  // pools one launch of this kernel can take
  static_assert(PER >= 1, "a batched kernel's arguments must fit the kernel argument segment");
  unsigned grid[COOK_MULTI_MAX];
  ArgPack<A...> a[PER];
};

template <auto F, class... X>
static __device__ __forceinline__ void pack_call(const ArgPack<>&, const X&... x) {
  F(x...);
}
template <auto F, class H, class... T, class... X>
static __device__ __forceinline__ void pack_call(const ArgPack<H, T...>& p, const X&... x) {
  pack_call<F>(p.t, x..., p.h);
}

template <auto F, int B, class... A>
__global__ void __launch_bounds__(B) cook_multi(const MultiArgs<A...> m) {
  const unsigned pool = blockIdx.y;
  if (blockIdx.x >= m.grid[pool]) return;
  pack_call<F>(m.a[pool]);
}

// scan.hpp — multi-block segmented inclusive scan (scan-then-propagate, three launches) for wave64.
//
// Serves every prefix sum of the rank path: per-user usage prefixes (scheduler.clj:2057-2071 limit-over-quota-jobs,
// dru.clj:43-66 accumulate-resources), the over-quota counter, and the queue-order usage prefixes of the pool / quota
// group filters (tools.clj:917-933).  Value types carry a `bad` bit: a parallel scan associates additions differently
// from the reference's left-to-right `reductions`; it is bit-identical iff every addition it performs is exact
// (TwoSum error == 0).  Elements whose prefix involved an inexact addition are flagged and recomputed sequentially by
// the caller's fix-up kernel, so results are exact for ANY fp64 inputs, not only integer-valued ones.
//
//   seg_scan_local     : per-block scan of a tile (thread-sequential -> wave shuffles -> cross-wave in LDS); writes
//                        the locally scanned values, the block aggregate and the block's first segment head.
//   seg_scan_blocksums : ONE workgroup scans the block aggregates (exclusive carry per block).
//   seg_scan_propagate : adds the carry to the elements in front of the block's first head.
#pragma once
#include "common.hpp"

struct SumU4 {  // usage 4-vector with exactness tracking
  double count, cpus, mem, gpus;
  unsigned bad;
  static __host__ __device__ __forceinline__ SumU4 zero() { return SumU4{0.0, 0.0, 0.0, 0.0, 0u}; }
};
static __host__ __device__ __forceinline__ SumU4 combine(const SumU4& a, const SumU4& b) {
  SumU4 r;
  r.count = a.count + b.count;  // integer-valued < 2^53: always exact
  r.cpus = a.cpus + b.cpus;
  r.mem = a.mem + b.mem;
  r.gpus = a.gpus + b.gpus;
  const bool inexact = two_sum_err(a.cpus, b.cpus, r.cpus) != 0.0 || two_sum_err(a.mem, b.mem, r.mem) != 0.0 ||
                       two_sum_err(a.gpus, b.gpus, r.gpus) != 0.0;
  r.bad = a.bad | b.bad | (inexact ? 1u : 0u);
  return r;
}
static __device__ __forceinline__ SumU4 shfl_up_v(const SumU4& v, unsigned d) {
  SumU4 r;
  r.count = __shfl_up(v.count, d, COOK_WAVE);
  r.cpus = __shfl_up(v.cpus, d, COOK_WAVE);
  r.mem = __shfl_up(v.mem, d, COOK_WAVE);
  r.gpus = __shfl_up(v.gpus, d, COOK_WAVE);
  r.bad = __shfl_up(v.bad, d, COOK_WAVE);
  return r;
}

struct SumI {  // integer counter (over-quota count, stream-compaction offsets)
  int v;
  static __host__ __device__ __forceinline__ SumI zero() { return SumI{0}; }
};
static __host__ __device__ __forceinline__ SumI combine(const SumI& a, const SumI& b) { return SumI{a.v + b.v}; }
static __device__ __forceinline__ SumI shfl_up_v(const SumI& v, unsigned d) { return SumI{__shfl_up(v.v, d, COOK_WAVE)}; }

constexpr int SS_THREADS = 256;
constexpr int SS_IPT = 4;
constexpr int SS_TILE = SS_THREADS * SS_IPT;

template <class T>
struct SegAgg {
  T v;
  unsigned f;  // tile/thread contains a segment head
};

// Inclusive segmented scan across the threads of a block of (f, v) pairs; returns the EXCLUSIVE prefix of this thread
// (ef = some earlier thread of the block has a head) and, through block_out, the aggregate of the whole block.
template <class T, int THREADS>
static __device__ __forceinline__ SegAgg<T> block_seg_exclusive(SegAgg<T> mine, SegAgg<T>* lds_wave /*[THREADS/64]*/,
                                                                SegAgg<T>& block_out) {
  const unsigned lane = lane_id(), w = wave_id();
  SegAgg<T> inc = mine;
  for (unsigned d = 1; d < COOK_WAVE; d <<= 1) {
    T pv = shfl_up_v(inc.v, d);
    const unsigned pf = __shfl_up(inc.f, d, COOK_WAVE);
    if (lane >= d) {
      if (!inc.f) inc.v = combine(pv, inc.v);
      inc.f |= pf;
    }
  }
  if (lane == COOK_WAVE - 1) lds_wave[w] = inc;
  __syncthreads();
  // exclusive prefix over earlier waves (<= 16 entries)
  SegAgg<T> wex{T::zero(), 0u};
  SegAgg<T> all{T::zero(), 0u};
  constexpr int NW = THREADS / COOK_WAVE;
  for (int k = 0; k < NW; ++k) {
    const SegAgg<T> a = lds_wave[k];
    if ((unsigned)k == w) wex = all;
    if (a.f)
      all = a;
    else
      all.v = combine(all.v, a.v);
    all.f |= a.f;
  }
  block_out = all;
  // exclusive within wave
  SegAgg<T> lex;
  lex.v = shfl_up_v(inc.v, 1);
  lex.f = __shfl_up(inc.f, 1, COOK_WAVE);
  if (lane == 0) lex = SegAgg<T>{T::zero(), 0u};
  SegAgg<T> ex;
  if (lex.f)
    ex = lex;
  else
    ex.v = combine(wex.v, lex.v);
  ex.f = wex.f | lex.f;
  __syncthreads();
  return ex;
}

// head[i] != 0 starts a new segment at i (head == nullptr: one segment = plain scan).
template <class T, class Load>
COOK_KERNEL void seg_scan_local(Load load, const uint8_t* __restrict__ head, unsigned n,
                                                             T* __restrict__ out, SegAgg<T>* __restrict__ block_agg,
                                                             unsigned* __restrict__ block_first_head) {
  __shared__ SegAgg<T> lds_wave[SS_THREADS / COOK_WAVE];
  __shared__ unsigned first_head;
  if (threadIdx.x == 0) first_head = 0xFFFFFFFFu;
  __syncthreads();
  const unsigned base = blockIdx.x * SS_TILE + threadIdx.x * SS_IPT;
  T loc[SS_IPT];
  SegAgg<T> agg{T::zero(), 0u};
  unsigned my_first = 0xFFFFFFFFu;
#pragma unroll
  for (int k = 0; k < SS_IPT; ++k) {
    const unsigned i = base + k;
    if (i < n) {
      const T x = load(i);
      const bool h = head ? head[i] != 0 : (i == 0);
      if (h) {
        agg.v = x;
        agg.f = 1u;
        if (my_first == 0xFFFFFFFFu) my_first = i;
      } else {
        agg.v = (k == 0) ? x : combine(agg.v, x);
      }
      loc[k] = agg.v;
    }
  }
  if (my_first != 0xFFFFFFFFu) atomicMin(&first_head, my_first);
  SegAgg<T> blk;
  const SegAgg<T> ex = block_seg_exclusive<T, SS_THREADS>(agg, lds_wave, blk);
  // apply the exclusive prefix to local items in front of this thread's first head
#pragma unroll
  for (int k = 0; k < SS_IPT; ++k) {
    const unsigned i = base + k;
    if (i < n) {
      T v = loc[k];
      if (i < my_first && (ex.f || threadIdx.x > 0)) v = combine(ex.v, v);
      out[i] = v;
    }
  }
  if (threadIdx.x == 0) {
    block_agg[blockIdx.x] = blk;
    block_first_head[blockIdx.x] = first_head;
  }
}

// carry[b] = exclusive segmented prefix over block aggregates; single workgroup, tiles of SS_THREADS blocks.
template <class T>
COOK_KERNEL void seg_scan_blocksums(const SegAgg<T>* __restrict__ block_agg, unsigned nblocks,
                                                                 SegAgg<T>* __restrict__ carry) {
  __shared__ SegAgg<T> lds_wave[SS_THREADS / COOK_WAVE];
  __shared__ SegAgg<T> run_s;
  if (threadIdx.x == 0) run_s = SegAgg<T>{T::zero(), 0u};
  __syncthreads();
  for (unsigned tile = 0; tile < nblocks; tile += SS_THREADS) {
    const unsigned b = tile + threadIdx.x;
    SegAgg<T> mine{T::zero(), 0u};
    if (b < nblocks) mine = block_agg[b];
    SegAgg<T> blk;
    const SegAgg<T> ex = block_seg_exclusive<T, SS_THREADS>(mine, lds_wave, blk);
    const SegAgg<T> run = run_s;
    if (b < nblocks) {
      SegAgg<T> c;
      if (ex.f)
        c = ex;
      else
        c.v = combine(run.v, ex.v);
      c.f = run.f | ex.f;
      carry[b] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      SegAgg<T> nr;
      if (blk.f)
        nr = blk;
      else
        nr.v = combine(run.v, blk.v);
      nr.f = run.f | blk.f;
      run_s = nr;
    }
    __syncthreads();
  }
}

template <class T>
COOK_KERNEL void seg_scan_propagate(T* __restrict__ out, unsigned n,
                                                                 const SegAgg<T>* __restrict__ carry,
                                                                 const unsigned* __restrict__ block_first_head) {
  const unsigned b = blockIdx.x;
  if (b == 0) return;
  const SegAgg<T> c = carry[b];
  const unsigned fh = block_first_head[b];
  const unsigned base = b * SS_TILE;
#pragma unroll
  for (int k = 0; k < SS_IPT; ++k) {
    const unsigned i = base + k * SS_THREADS + threadIdx.x;
    if (i < n && i < fh) out[i] = combine(c.v, out[i]);
  }
}

// the same for up to SS_THREADS blocks without the launch in between: every block folds the aggregates in front of it itself
// (171 of them for a pool's 175k items; the association differs from seg_scan_blocksums', which is free: a partial sum that is not
// exact is flagged whatever the tree, common.hpp "exact-sum tracking")
template <class T>
COOK_KERNEL void seg_scan_propagate_fused(T* __restrict__ out, unsigned n,
                                                                       const SegAgg<T>* __restrict__ block_agg,
                                                                       const unsigned* __restrict__ block_first_head) {
  __shared__ SegAgg<T> lds_wave[SS_THREADS / COOK_WAVE];
  const unsigned b = blockIdx.x;
  if (b == 0) return;
  SegAgg<T> mine{T::zero(), 0u};
  if (threadIdx.x < b) mine = block_agg[threadIdx.x];
  SegAgg<T> c;
  (void)block_seg_exclusive<T, SS_THREADS>(mine, lds_wave, c);
  const unsigned fh = block_first_head[b];
  const unsigned base = b * SS_TILE;
#pragma unroll
  for (int k = 0; k < SS_IPT; ++k) {
    const unsigned i = base + k * SS_THREADS + threadIdx.x;
    if (i < n && i < fh) out[i] = combine(c.v, out[i]);
  }
}


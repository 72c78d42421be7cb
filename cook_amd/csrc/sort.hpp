// sort.hpp — stable LSD radix sort of a permutation by 64-bit key words, 8-bit digits, wave64.
//
// Used for the per-user task order (tools.clj:614-641 feature vector, three key words) and for the global DRU order
// (dru.clj:82-126, one order-preserving fp64 key word).  Keys stay in place; only the u32 permutation moves, so a
// pass streams 4 B/item in + 4 B/item out plus an 8 B gather that hits L2 (N*8 B <= 12 MB for 1.4M tasks).
//
// One pass of a LARGE input = three launches:
//   radix_hist    : per-block 256-bin digit histogram (LDS atomics)              -> hist[digit][block]
//   radix_scan    : exclusive scan of hist in (digit-major, block-minor) order    (single workgroup)
//   radix_scatter : stable placement.  A block's tile is split into one contiguous chunk per wave; per-wave digit
//                   counts give each wave its base, then every wave walks its chunk 64 items at a time, ranking equal
//                   digits with 8 ballots (match-any) so earlier positions keep earlier slots.
// One pass of an input of up to 2M items (RS_FUSED_BLOCKS tiles of 2048: a pool's 175k tasks are 86; or RS_FUSED_BLOCKS_LARGE tiles of
// 4096) = two launches: the histogram is
// kept [block][digit] and every scatter block sums the columns itself (the counts of each digit over the earlier blocks and over all
// blocks: 16-byte loads, a wave per fourth of the rows) — the single-workgroup scan and its launch gap were a third of a pass
// (9.6 + ~5 us of 45 + 15, profiles/r02k), and a chain of passes is what the rank stage is made of.
// The host skips digits that are constant over the whole input (radix_varying_bits, or the kernel that builds the keys:
// rank_build_keys) and starts every digit at the lowest
// varying bit not sorted yet, so sparse varying bits do not cost a pass per byte they touch.
#pragma once
#include "common.hpp"

constexpr int RS_THREADS = 256;                     // 4 waves
constexpr int RS_WAVES = RS_THREADS / COOK_WAVE;    // waves per block
constexpr int RS_IPL_LARGE = 16;                    // items per lane: 4096 items per block
constexpr int RS_IPL_SMALL = 8;                     // pool-sized inputs: 2048 items per block, twice the blocks (86 on 256 CUs for 175k)
constexpr unsigned RS_FUSED_BLOCKS = COOK_SHAPE(128, 2);  // up to this many small tiles the scatter derives its bases itself ...
constexpr unsigned RS_FUSED_BLOCKS_LARGE = COOK_SHAPE(512, 3);  // ... and up to this many large ones (the rebalancer's million slots are 245:
                                                                // 245 KB of L2 reads per block against a 13 us scan launch and its gap)
constexpr unsigned rs_tile(int ipl) { return (unsigned)(RS_WAVES * COOK_WAVE * ipl); }

static __device__ __forceinline__ unsigned rs_digit(const uint64_t* __restrict__ key, const uint32_t* __restrict__ perm_in,
                                                    unsigned i, unsigned shift) {
  const unsigned src = perm_in ? perm_in[i] : i;
  return (unsigned)(key[src] >> shift) & 0xFFu;
}

template <int IPL>
COOK_KERNEL void radix_hist(const uint64_t* __restrict__ key, const uint32_t* __restrict__ perm_in,
                                                         unsigned n, unsigned shift, unsigned nblocks, unsigned fused,
                                                         uint32_t* __restrict__ hist) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned base = blockIdx.x * rs_tile(IPL);
  unsigned d[rs_tile(IPL) / RS_THREADS];
#pragma unroll
  for (unsigned k = 0; k < rs_tile(IPL) / RS_THREADS; ++k) {  // every gather in flight before the first LDS atomic
    const unsigned i = base + k * RS_THREADS + threadIdx.x;
    d[k] = i < n ? rs_digit(key, perm_in, i, shift) : 0xFFFFFFFFu;
  }
#pragma unroll
  for (unsigned k = 0; k < rs_tile(IPL) / RS_THREADS; ++k)
    if (d[k] != 0xFFFFFFFFu) atomicAdd(&h[d[k]], 1u);
  __syncthreads();
  hist[fused ? blockIdx.x * 256u + threadIdx.x : threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Exclusive scan of a u32 array by ONE workgroup of 1024 threads (len = 256 * nblocks: 11k entries for a pool's 175k tasks, 63k for a
// million).  The entries were written by the previous kernel on other XCDs, so every load is a trip to memory (~2 us): a step that
// loads, scans and stores 1024 entries at a time spent 57 us on the 63k entries (rocprofv3, profiles/r02i).  Here a thread takes
// SCAN1_IPT consecutive entries per step and the loads of SCAN1_NS steps are issued before the first step runs: one memory latency
// per 64k entries instead of 62.
constexpr int SCAN1_THREADS = 1024;
constexpr int SCAN1_IPT = 8;
constexpr int SCAN1_NS = 8;
static_assert(SCAN1_IPT == 8, "a thread moves its entries as two 16-byte vectors");
COOK_KERNEL void excl_scan_u32_single(uint32_t* __restrict__ data, unsigned len,
                                                                      uint32_t* __restrict__ total_out) {
  __shared__ unsigned wsum[2][SCAN1_THREADS / COOK_WAVE];
  unsigned carry = 0;  // every thread tracks the running total itself (read from LDS once per step)
  const unsigned lane = lane_id(), w = wave_id();
  constexpr unsigned STEP = SCAN1_THREADS * SCAN1_IPT;
  for (unsigned super = 0; super < len; super += STEP * SCAN1_NS) {
    unsigned v[SCAN1_NS][SCAN1_IPT];
#pragma unroll
    for (int s = 0; s < SCAN1_NS; ++s) {
      const unsigned i0 = super + (unsigned)s * STEP + threadIdx.x * SCAN1_IPT;
      if (i0 + SCAN1_IPT <= len) {  // two 16-byte loads (the thread's 8 entries are 32-byte aligned): 4-byte accesses at a 32-byte
                                    // stride between lanes are one memory transaction each — 63k of them were the 57 us
        const uint4 a = *reinterpret_cast<const uint4*>(data + i0), b = *reinterpret_cast<const uint4*>(data + i0 + 4);
        v[s][0] = a.x, v[s][1] = a.y, v[s][2] = a.z, v[s][3] = a.w;
        v[s][4] = b.x, v[s][5] = b.y, v[s][6] = b.z, v[s][7] = b.w;
      } else {
#pragma unroll
        for (int q = 0; q < SCAN1_IPT; ++q) v[s][q] = i0 + q < len ? data[i0 + q] : 0u;
      }
    }
#pragma unroll
    for (int s = 0; s < SCAN1_NS; ++s) {
      if (super + (unsigned)s * STEP >= len) break;  // block-uniform
      const unsigned i0 = super + (unsigned)s * STEP + threadIdx.x * SCAN1_IPT;
      unsigned tot = 0;
#pragma unroll
      for (int q = 0; q < SCAN1_IPT; ++q) tot += v[s][q];
      unsigned inc = tot;
      for (unsigned d = 1; d < COOK_WAVE; d <<= 1) {
        const unsigned t = __shfl_up(inc, d, COOK_WAVE);
        if (lane >= d) inc += t;
      }
      if (lane == COOK_WAVE - 1) wsum[s & 1][w] = inc;  // double-buffered: one barrier per step
      __syncthreads();
      unsigned wbase = 0, all = 0;
      for (unsigned k = 0; k < SCAN1_THREADS / COOK_WAVE; ++k) {
        const unsigned x = wsum[s & 1][k];
        wbase += k < w ? x : 0u;
        all += x;
      }
      unsigned run = carry + wbase + inc - tot;  // exclusive prefix of this thread's first entry
      unsigned o[SCAN1_IPT];
#pragma unroll
      for (int q = 0; q < SCAN1_IPT; ++q) {
        o[q] = run;
        run += v[s][q];
      }
      if (i0 + SCAN1_IPT <= len) {
        *reinterpret_cast<uint4*>(data + i0) = uint4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<uint4*>(data + i0 + 4) = uint4{o[4], o[5], o[6], o[7]};
      } else {
#pragma unroll
        for (int q = 0; q < SCAN1_IPT; ++q)
          if (i0 + q < len) data[i0 + q] = o[q];
      }
      carry += all;
    }
  }
  if (total_out && threadIdx.x == 0) *total_out = carry;
}

template <int IPL>
COOK_KERNEL void radix_scatter(const uint64_t* __restrict__ key, const uint32_t* __restrict__ perm_in,
                                                            uint32_t* __restrict__ perm_out, unsigned n, unsigned shift,
                                                            unsigned nblocks, unsigned fused, const uint32_t* __restrict__ hist) {
  __shared__ unsigned whist[RS_WAVES][256];
  __shared__ unsigned wtot[RS_WAVES];
  __shared__ uint4 s_tot[RS_WAVES][COOK_WAVE], s_bel[RS_WAVES][COOK_WAVE];
  const unsigned lane = lane_id(), w = wave_id();
  for (int k = 0; k < RS_WAVES; ++k) whist[k][threadIdx.x] = 0;
  // fused: the count of every digit in the blocks before this one, and in all blocks.  Wave w takes the rows w, w + RS_WAVES, ...;
  // a lane four digits as one 16-byte load (a thread per digit issued nblocks 4-byte loads: 245 of them for a million items took
  // longer than the scan launch they replaced), eight rows in flight; the waves' partial sums meet in LDS.
  unsigned below = 0, total = 0;
  if (fused) {
    uint4 tot{0u, 0u, 0u, 0u}, bel{0u, 0u, 0u, 0u};
#pragma unroll 8
    for (unsigned t = w; t < nblocks; t += RS_WAVES) {
      const uint4 v = *reinterpret_cast<const uint4*>(hist + (size_t)t * 256u + 4u * lane);
      tot.x += v.x, tot.y += v.y, tot.z += v.z, tot.w += v.w;
      if (t < blockIdx.x) bel.x += v.x, bel.y += v.y, bel.z += v.z, bel.w += v.w;
    }
    s_tot[w][lane] = tot;
    s_bel[w][lane] = bel;
  }
  __syncthreads();
  if (fused) {  // digit d = threadIdx.x
    const unsigned* pt = reinterpret_cast<const unsigned*>(&s_tot[0][0]);
    const unsigned* pb = reinterpret_cast<const unsigned*>(&s_bel[0][0]);
    for (int k = 0; k < RS_WAVES; ++k) {
      total += pt[k * 256 + threadIdx.x];
      below += pb[k * 256 + threadIdx.x];
    }
  }
  const unsigned wbase = blockIdx.x * rs_tile(IPL) + w * (COOK_WAVE * IPL);
  // phase 1: the wave's items (kept in registers for phase 3) and its digit counts
  unsigned src[IPL], dg[IPL];
#pragma unroll
  for (int k = 0; k < IPL; ++k) {
    const unsigned i = wbase + k * COOK_WAVE + lane;
    src[k] = i < n ? (perm_in ? perm_in[i] : i) : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int k = 0; k < IPL; ++k) dg[k] = src[k] != 0xFFFFFFFFu ? (unsigned)(key[src[k]] >> shift) & 0xFFu : 0u;
#pragma unroll
  for (int k = 0; k < IPL; ++k)
    if (src[k] != 0xFFFFFFFFu) atomicAdd(&whist[w][dg[k]], 1u);
  if (fused) {  // exclusive scan of the digit totals over the 256 threads
    unsigned inc = total;
    for (unsigned d = 1; d < COOK_WAVE; d <<= 1) {
      const unsigned t = __shfl_up(inc, d, COOK_WAVE);
      if (lane >= d) inc += t;
    }
    if (lane == COOK_WAVE - 1) wtot[w] = inc;
    __syncthreads();
    unsigned before = 0;
    for (unsigned k = 0; k < (unsigned)RS_WAVES; ++k) before += k < w ? wtot[k] : 0u;
    below += before + inc - total;
  } else {
    __syncthreads();
    below = hist[threadIdx.x * nblocks + blockIdx.x];
  }
  // phase 2: digit d = threadIdx.x; turn counts into each wave's starting slot
  {
    unsigned base = below;
    for (int k = 0; k < RS_WAVES; ++k) {
      const unsigned t = whist[k][threadIdx.x];
      whist[k][threadIdx.x] = base;
      base += t;
    }
  }
  __syncthreads();
  // phase 3: stable ranking inside the wave, 64 consecutive positions per step
  const unsigned long long lt = lanemask_lt();
#pragma unroll
  for (int k = 0; k < IPL; ++k) {
    const bool valid = src[k] != 0xFFFFFFFFu;
    const unsigned d = dg[k];
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot(valid && ((d >> b) & 1u));
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    unsigned slot = 0;
    if (valid) slot = whist[w][d];
    wave_sync();
    if (valid) {
      const unsigned rank = (unsigned)__popcll(peers & lt);
      if (rank == 0) whist[w][d] = slot + (unsigned)__popcll(peers);  // lowest lane of the peer set
      perm_out[slot + rank] = src[k];
    }
    wave_sync();
  }
}

// OR over all items of (key[i] ^ key[0]) : bits that differ somewhere.  One atomicOr per block; launched with a few dozen blocks.
COOK_KERNEL void radix_varying_bits(const uint64_t* __restrict__ key, unsigned n,
                                                          unsigned long long* __restrict__ out_mask, unsigned nblk) {
  const uint64_t k0 = key[0];
  unsigned long long m = 0;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nblk * blockDim.x) m |= key[i] ^ k0;
  for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d, COOK_WAVE);
  __shared__ unsigned long long s_m[256 / COOK_WAVE];  // one atomic per block
  if (lane_id() == 0) s_m[wave_id()] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long b = 0ull;
    for (unsigned w = 0; w < blockDim.x / COOK_WAVE; ++w) b |= s_m[w];
    if (b) atomicOr(out_mask, b);
  }
}

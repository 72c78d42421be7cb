// tile_sort.hpp — sorting INSIDE segments, a tile of whole segments per workgroup, in LDS.
//
// The sorted-merge tie rule of cook_rank (dru.clj:82-104, rank_kernels.hpp) orders the items of one tie group by the rank of a
// predecessor, once per doubling round.  As a global LSD radix sort of the tied items (sort.hpp) a round was 5 pass pairs of ~16 us
// plus two scans and the compaction around them — 4 rounds x 22 launches, half of the stage's launches, and the stage is bound by
// its launch count (profiles/r04p_kernel_stats_one_pool.csv: 200 launches, 1.3 ms).  Tie groups are short next to the input (the
// benchmark pool: 16 734 groups of 138k tied items, the longest 2 271), so:
// Here a workgroup takes the segments that START in its nominal stretch of TS_NOMINAL positions (so tiles are made of whole segments
// and a tile is at most TS_NOMINAL + longest segment - 1 long), sorts them in LDS with one bitonic network whose most significant key
// is the segment, and writes them back: one launch.  An input with a segment too long for a tile raises a flag and the host takes the
// radix path for it (engine.hip) — the result is the same order either way, the tests run both.
// (The per-user task order was tried the same way and dropped: a user of the benchmark pool holds 26 375 of its 175 000 tasks, and
// the counting sort by user in front of it spent 300 us per launch on same-address atomics.)
#pragma once
#include "common.hpp"

constexpr unsigned TS_THREADS = COOK_SHAPE(1024, 256);
constexpr unsigned TS_NOMINAL = COOK_SHAPE(1024, 64);   // positions per workgroup before the spill of its last segment
constexpr unsigned TS_CAP = COOK_SHAPE(8192, 256);      // 8 B of LDS per item (tiles of 512 / 4096 with 512 threads: 54 us per launch against 39 — the launch lasts as long as the tile with the longest group)
constexpr unsigned TS_LIDX_BITS = 13;
static_assert(TS_CAP <= (1u << TS_LIDX_BITS), "a local index is 13 bits of the sort keys");
static_assert((TS_CAP & (TS_CAP - 1)) == 0, "bitonic network sizes");
static_assert(TS_CAP % TS_THREADS == 0, "the write-back holds TS_CAP / TS_THREADS items per thread");
constexpr unsigned TS_MAX_GROUP = TS_CAP - TS_NOMINAL + 1;  // the longest segment a tile takes for certain

// last position <= pos that starts a segment (position 0 always does)
static __device__ __forceinline__ unsigned ts_prev_head(const uint8_t* __restrict__ head, unsigned pos) {
  for (;;) {
    const unsigned l = lane_id();
    const bool in = l <= pos;
    const unsigned long long m = __ballot(in && (pos - l == 0 || head[pos - l] != 0));
    if (m) return pos - ((unsigned)__ffsll(m) - 1u);
    pos -= COOK_WAVE;  // no head among 64 positions, none of them position 0: pos >= 64
  }
}

// the tile of workgroup b: [lo, hi) = the segments whose first position lies in [b * nominal, (b + 1) * nominal).  The whole workgroup
// looks for the two heads, blockDim positions per step: a wave stepping through a 2 271-item group 64 positions at a time was a chain
// of 36 dependent loads, 25 of the launch's 39 us (the launch lasts as long as its slowest tile).
static __device__ __forceinline__ unsigned ts_block_next_head(const uint8_t* __restrict__ head, unsigned pos, unsigned n, unsigned* s_min) {
  for (;; pos += blockDim.x) {
    if (threadIdx.x == 0) *s_min = 0xFFFFFFFFu;
    __syncthreads();
    const unsigned p = pos + threadIdx.x;
    const bool hit = p >= n || head[p] != 0;
    const unsigned long long m = __ballot(hit);
    if (m && lane_id() == (unsigned)__ffsll(m) - 1u) atomicMin(s_min, p < n ? p : n);  // one atomic per wave
    __syncthreads();
    const unsigned r = *s_min;
    __syncthreads();  // (everyone has read it before the next step resets it)
    if (r != 0xFFFFFFFFu) return r;
  }
}
static __device__ __forceinline__ void ts_tile_bounds(const uint8_t* __restrict__ head, unsigned n, unsigned nominal, unsigned* s_b /*[3] LDS*/) {
  const unsigned a = blockIdx.x * nominal;
  const unsigned lo = blockIdx.x == 0 ? 0u : ts_block_next_head(head, a, n, &s_b[2]);
  const unsigned hi = a + nominal >= n ? n : ts_block_next_head(head, a + nominal, n, &s_b[2]);
  if (threadIdx.x == 0) s_b[0] = lo, s_b[1] = hi;
  __syncthreads();
}

// ascending bitonic sort of s[0, n2), n2 a power of two; every thread of the workgroup takes part.  Thread t's pair of a step with
// distance j is (i, i | j), i = t with a zero inserted at bit log2(j): for j <= 64 the 64 pairs of a wave lie in the wave's own 128
// consecutive elements, so those steps need no workgroup barrier between them (a wave's LDS accesses complete in order) — 25 barriers
// instead of 78 for 4096 elements.
template <class Swap>
static __device__ __forceinline__ void ts_bitonic(unsigned n2, Swap cmp_swap) {
  for (unsigned k = 2; k <= n2; k <<= 1) {
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
      for (unsigned t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
        const unsigned i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
        cmp_swap(i, i | j, (i & k) == 0);
      }
      if (j > COOK_WAVE) __syncthreads();
      else wave_sync();
    }
    if (k >= 2 * COOK_WAVE && k < n2) __syncthreads();  // the next stage starts with pairs across waves
  }
  __syncthreads();
}
static __device__ __forceinline__ unsigned ts_pow2_at_least(unsigned x) {
  unsigned p = 2;
  while (p < x) p <<= 1;
  return p;
}

// ---- the sorted-merge tie rule, one doubling round = two launches --------------------------------------------------------------
// (what the round computes: rank_kernels.hpp, "A.5 global order")
struct TieCtl {
  unsigned overflow;        // some tie group does not fit a tile: the host restarts on the radix path
  unsigned equal_runs;      // a user with equal consecutive keys (tie_heads): the host collapses the runs first
  unsigned tied_after[32];  // tied items left by round r
};

// rank of every item under the current groups: U + first position of its group
COOK_KERNEL void tie_rank_assign(const uint32_t* __restrict__ perm, const uint8_t* __restrict__ thead, unsigned nk,
                                                       unsigned n_users, int round, const TieCtl* __restrict__ ctl,
                                                       uint32_t* __restrict__ rank_of_item) {
  if (ctl->equal_runs || (round > 0 && ctl->tied_after[round - 1] == 0)) return;
  __shared__ unsigned s_last[256 / COOK_WAVE];
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned w = wave_id(), l = lane_id();
  const bool h = p < nk && thead[p] != 0;
  const unsigned long long m = __ballot(h);
  if (l == 0) s_last[w] = m ? (p - l) + 63u - (unsigned)__clzll(m) : 0xFFFFFFFFu;
  __syncthreads();
  const unsigned long long le = m & (lanemask_lt() | (1ull << l));
  unsigned start;
  if (le) {
    start = (p - l) + 63u - (unsigned)__clzll(le);
  } else {
    start = 0xFFFFFFFFu;
    for (unsigned k = 0; k < w; ++k) start = s_last[k] != 0xFFFFFFFFu ? s_last[k] : start;
    // nothing in this workgroup before the lane: the search goes back from the workgroup's first position (wave-uniform inputs;
    // lanes that have their start idle through it)
  }
  const bool need = start == 0xFFFFFFFFu && p - l < nk;
  if (__any(need)) {
    const unsigned first = blockIdx.x * blockDim.x;  // > 0: position 0 is a head
    const unsigned back = ts_prev_head(thead, first - 1u);
    if (need) start = back;
  }
  if (p < nk) rank_of_item[perm[p]] = n_users + start;
}

COOK_KERNEL void tie_sort_tiles(uint32_t* __restrict__ perm, uint8_t* __restrict__ thead,
                                                             const uint8_t* __restrict__ dhead, unsigned nk, unsigned n_users,
                                                             unsigned n_items, int round, const uint32_t* __restrict__ rank_of_item,
                                                             const uint32_t* __restrict__ user_of,
                                                             const uint32_t* __restrict__ seg_first, TieCtl* __restrict__ ctl) {
  if (ctl->equal_runs || (round > 0 && ctl->tied_after[round - 1] == 0)) return;
  // 8 B of LDS per item = 64 KB: TWO workgroups per CU (the items themselves stay in `perm`: a key carries its position in the tile, and the
  // write-back reads through it before anything is written — with a third array of 4 B per item a CU held one workgroup, and a launch
  // for the eight pools of a GPU, 1 360 tiles, went through the chip in six waves: 142 us)
  __shared__ uint64_t s_key[TS_CAP];
  __shared__ unsigned s_b[3], s_any;
  if (threadIdx.x == 0) s_any = 0;
  ts_tile_bounds(dhead, nk, TS_NOMINAL, s_b);  // tiles follow the groups of EQUAL KEYS: the refined heads move while other tiles look
  const unsigned lo = s_b[0], hi = s_b[1];
  if (lo >= hi) return;
  const unsigned len = hi - lo;
  if (len > TS_CAP) {
    if (threadIdx.x == 0) atomicOr(&ctl->overflow, 1u);
    return;
  }
  bool any = false;
  for (unsigned r = threadIdx.x; r < len; r += blockDim.x) any |= thead[lo + r] == 0;
  if (__any(any) && lane_id() == 0) s_any = 1;
  __syncthreads();
  if (!s_any) return;  // every group of the tile is a single item
  const unsigned n2 = ts_pow2_at_least(len);
  const unsigned maxr = n_users + n_items;
  for (unsigned r = threadIdx.x; r < n2; r += blockDim.x) {
    uint64_t key = ~0ull;
    if (r < len) {
      const unsigned i = perm[lo + r];
      const unsigned u = user_of[i];
      const unsigned idx = i - seg_first[u];  // position in the user's list
      const unsigned gl = rank_of_item[i] - n_users - lo;  // the group's first position in the tile
      unsigned sec;
      if (round == 0) {
        const unsigned pr = idx >= 1 ? rank_of_item[i - 1] : (n_users - 1 - u);
        sec = maxr - pr;  // later predecessor first
      } else {
        const unsigned step = 1u << round;
        if (idx >= step)
          sec = rank_of_item[i - step];
        else if (idx + 1 == step)
          sec = n_users - 1 - u;
        else
          sec = 0;  // the item's sequence already ended inside the rank: it is alone in its group
      }
      key = ((uint64_t)gl << (32 + TS_LIDX_BITS)) | ((uint64_t)sec << TS_LIDX_BITS) | (uint64_t)r;
    }
    s_key[r] = key;
  }
  __syncthreads();
  ts_bitonic(n2, [&](unsigned a, unsigned b, bool up) {
    const uint64_t x = s_key[a], y = s_key[b];
    if ((x > y) == up) s_key[a] = y, s_key[b] = x;
  });
  constexpr unsigned PER = TS_CAP / TS_THREADS;  // items of a tile per thread
  uint32_t moved[PER];                          // the items that end up at this thread's positions, read before any position is written
#pragma unroll
  for (unsigned q = 0; q < PER; ++q) {
    const unsigned r = q * TS_THREADS + threadIdx.x;
    moved[q] = r < len ? perm[lo + ((unsigned)s_key[r] & ((1u << TS_LIDX_BITS) - 1u))] : 0u;
  }
  __syncthreads();
  unsigned tied = 0;
#pragma unroll
  for (unsigned q = 0; q < PER; ++q) {  // uniform trip count: the ballot below wants whole waves
    const unsigned r = q * TS_THREADS + threadIdx.x;
    bool t = false;
    if (r < len) {
      const uint64_t k = s_key[r];
      const bool h = r == 0 || (k >> TS_LIDX_BITS) != (s_key[r - 1] >> TS_LIDX_BITS);
      const bool next_same = r + 1 < len && (s_key[r + 1] >> TS_LIDX_BITS) == (k >> TS_LIDX_BITS);
      perm[lo + r] = moved[q];
      thead[lo + r] = h ? 1 : 0;
      t = !h || next_same;
    }
    tied += (unsigned)__popcll(__ballot(t));
  }
  if (lane_id() == 0 && tied) atomicAdd(&ctl->tied_after[round], tied);
}


// offers_kernels.hpp — offer construction from node state (the step before the match path): the numeric core of
// kubernetes.compute-cluster/generate-offers (kubernetes/compute_cluster.clj:68-190) with api/get-capacity
// (kubernetes/api.clj:874-884), api/get-consumption (:886-930) and the boolean of api/node-schedulable? (:782-847).
//
// HBM-bound streaming work: pods are stably partitioned by node (one radix sort of a u32 permutation), every node then
// folds its pods LEFT TO RIGHT in the list order the reference's (apply deep-merge-with + ...) uses, so the fp64 sums are
// bit-identical for any requests (0.1-cpu pods are not dyadic: no re-association is allowed).  Parallelism is across nodes.
#pragma once
#include "common.hpp"

// acc + p[0] + p[1] + ... + p[n-1], left to right, p in LDS (all lanes read the same words: broadcasts).  The reads of the next
// 32 values are issued before the 32 dependent adds of the current group, so the chain pays the fp64 add latency only.
static __device__ __forceinline__ double fold_lds_in_order(const double* p, unsigned n, double acc) {
  constexpr int G = 16;
  double cur[G], nxt[G];
  unsigned i = 0;
  if (n >= (unsigned)G) {
#pragma unroll
    for (int k = 0; k < G; ++k) cur[k] = p[k];
    for (; i + 2 * G <= n; i += G) {
#pragma unroll
      for (int k = 0; k < G; ++k) nxt[k] = p[i + G + k];
#pragma unroll
      for (int k = 0; k < G; ++k) acc = acc + cur[k];
#pragma unroll
      for (int k = 0; k < G; ++k) cur[k] = nxt[k];
    }
#pragma unroll
    for (int k = 0; k < G; ++k) acc = acc + cur[k];
    i += G;
  }
  for (; i < n; ++i) acc = acc + p[i];
  return acc;
}

struct NodeCols {
  const double* cpus;
  const double* mem;
  const int32_t* gpus;
  const uint32_t* gpu_model;
  const double* disk;
  const uint32_t* disk_type;
  const uint8_t* flags;
  unsigned n;
};
struct PodCols {
  const uint32_t* node;
  const double* cpus;
  const double* mem;
  const int32_t* gpus;
  const uint32_t* gpu_model;
  const double* disk;
  const uint32_t* disk_type;
  const uint8_t* flags;
  unsigned n;
};
struct NodeAvail {  // per node, before compaction
  double* cpus;       // (:cpus available), unclamped
  double* mem;
  double* cons_cpus;  // node-name->consumed entries (0.0 when the node has none)
  double* cons_mem;
  uint32_t* gpu_model;  // [n][gpu_slots] (:gpus available) as a small table: the node's own model first, then the models only its
  double* gpu_count;    //   pods name, in the order the pods list them (model 0 = empty slot)
  uint32_t* disk_type;  // [n][disk_slots] (:disk available), likewise
  double* disk_space;
  double* disk_cons;  // consumption under the node's own disk type
  int32_t* num_pods;
  uint8_t* status;    // COOK_NODE_ST_*
};

// sort key of a pod = its node index; pods without a node of this pool go behind every node
__global__ void __launch_bounds__(256) offers_pod_keys(const uint32_t* __restrict__ pod_node, unsigned n, unsigned n_nodes,
                                                       uint64_t* __restrict__ key) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const unsigned v = pod_node[i];
    key[i] = v < n_nodes ? v : n_nodes;
  }
}

// segment [seg_start, seg_end) of every node in the node-sorted pod order (both pre-set to 0 = no pods)
__global__ void __launch_bounds__(256) offers_seg_bounds(const uint32_t* __restrict__ perm, const uint64_t* __restrict__ key, unsigned n,
                                                         unsigned n_nodes, uint32_t* __restrict__ seg_start,
                                                         uint32_t* __restrict__ seg_end) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t k = key[perm[i]];
  if (k >= n_nodes) return;
  if (i == 0 || key[perm[i - 1]] != k) seg_start[k] = i;
  if (i + 1 == n || key[perm[i + 1]] != k) seg_end[k] = i + 1;
}

// a pod's columns packed in node-sorted order: the per-node fold then reads 40 contiguous bytes per pod instead of chasing the
// permutation through eight arrays (the gather itself is fully parallel over the pods)
struct PodRec {
  double cpus, mem, disk;
  int32_t gpus;
  uint32_t gpu_model, disk_type, flags;
};
__global__ void __launch_bounds__(256) offers_gather_pods(PodCols pd, const uint32_t* __restrict__ perm, PodRec* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pd.n) return;
  const unsigned p = perm[i];
  PodRec r;
  r.cpus = pd.cpus[p];
  r.mem = pd.mem[p];
  r.disk = pd.disk ? pd.disk[p] : -1.0;
  r.gpus = pd.gpus ? pd.gpus[p] : 0;
  r.gpu_model = pd.gpu_model ? pd.gpu_model[p] : 0u;
  r.disk_type = pd.disk_type ? pd.disk_type[p] : 0u;
  r.flags = pd.flags ? pd.flags[p] : 0u;
  out[i] = r;
}

// Clojure's (max 0.0 x) on a boxed double (clojure.lang.Numbers/max): NaN propagates, -0.0 survives
static __host__ __device__ __forceinline__ double clj_max0(double x) { return 0.0 > x ? 0.0 : x; }

// One lane per node: capacity (get-capacity), consumption (get-consumption: the node's pods folded in list order),
// available = deep-merge-with - (compute_cluster.clj:91), node-schedulable?.  The pods of a block's 256 consecutive nodes
// are one contiguous range of the node-sorted records: the block streams it through LDS in chunks (coalesced loads, all in
// flight at once) and every lane folds the part of its own segment that lies in the chunk — no lane chases global memory.
constexpr int ON_CHUNK = 1024;  // pod records per LDS chunk (40 KB)
constexpr int ON_MODELS = 64;   // gpu models totalled in LDS (more than that: global atomics)
constexpr int ON_SLOTS = 4;     // COOK_MAX_RES_SLOTS: entries of a host's "gpus" / "disk" map
__global__ void __launch_bounds__(256) offers_node_eval(NodeCols nd, const PodRec* __restrict__ pods,
                                                        const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_end,
                                                        int clobber_synthetic, int filter_unsound_gpu, int max_pods, unsigned n_gpu_models,
                                                        unsigned gpu_slots, unsigned disk_slots, NodeAvail out, unsigned long long* __restrict__ gpu_cap_by_model,
                                                        unsigned long long* __restrict__ gpu_cons_by_model,
                                                        uint32_t* __restrict__ block_offers) {
  __shared__ PodRec s_pod[ON_CHUNK];
  __shared__ unsigned s_cnt, s_lo, s_hi;
  __shared__ unsigned long long s_gcap[ON_MODELS], s_gcons[ON_MODELS];  // per-block totals: one global atomic per model and block
  if (threadIdx.x < ON_MODELS) s_gcap[threadIdx.x] = s_gcons[threadIdx.x] = 0ull;
  const bool lds_models = n_gpu_models < (unsigned)ON_MODELS;
  if (threadIdx.x == 0) {
    s_cnt = 0;
    s_lo = 0xFFFFFFFFu;
    s_hi = 0;
  }
  __syncthreads();
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = v < nd.n;
  const unsigned s = live ? seg_start[v] : 0u, t = live ? seg_end[v] : 0u;
  if (t > s) {
    atomicMin(&s_lo, s);
    atomicMax(&s_hi, t);
  }
  // ---- capacity (api.clj:849-884) ----
  const double cap_c = live ? nd.cpus[v] : 0.0, cap_m = live ? nd.mem[v] : 0.0;
  const int node_g = (live && nd.gpus) ? nd.gpus[v] : 0;
  const unsigned node_gm = (live && nd.gpu_model) ? nd.gpu_model[v] : 0u;
  const bool cap_has_gpu = node_gm != 0u && node_g > 0;  // force-gpu-model-in-resource-map (:849-855)
  const double node_d = (live && nd.disk) ? nd.disk[v] : -1.0;
  const unsigned node_dt = (live && nd.disk_type) ? nd.disk_type[v] : 0u;
  const bool cap_has_disk = node_d >= 0.0 && node_dt != 0u;  // force-disk-type-in-resource-map (:866-872)
  const unsigned nflags = (live && nd.flags) ? nd.flags[v] : 0u;
  // ---- consumption (api.clj:886-930) ----
  bool any = false, any_d = false, foreign_g = false, foreign_d = false;
  double cc = 0.0, cm = 0.0, own_d = 0.0;
  long long own_g = 0;
  // what the pods consume under a model / type the node's capacity does not list: deep-merge-with finds the key in ONE map only and
  // keeps the value as it is (util.clj:208-225), so (:gpus available) gains an entry {model consumed-count}.  Up to ON_SLOTS of
  // them, in first-appearance order; more than the caller's table holds sets COOK_NODE_ST_FOREIGN_*.
  unsigned fg_model[ON_SLOTS], fd_type[ON_SLOTS], n_fg = 0, n_fd = 0;
  long long fg_cnt[ON_SLOTS];
  double fd_val[ON_SLOTS];
#pragma unroll
  for (int q = 0; q < ON_SLOTS; ++q) fg_model[q] = fd_type[q] = 0u, fg_cnt[q] = 0, fd_val[q] = 0.0;
  const unsigned fg_room = live ? gpu_slots - (cap_has_gpu ? 1u : 0u) : 0u, fd_room = live ? disk_slots - (cap_has_disk ? 1u : 0u) : 0u;
  __syncthreads();
  const unsigned lo = s_lo, hi = s_hi;  // block-uniform; lo > hi when the block's nodes have no pods
  for (unsigned c0 = lo; c0 < hi; c0 += ON_CHUNK) {
    const unsigned c1 = c0 + ON_CHUNK < hi ? c0 + ON_CHUNK : hi;
    for (unsigned i = c0 + threadIdx.x; i < c1; i += blockDim.x) s_pod[i - c0] = pods[i];
    __syncthreads();
    const unsigned a = s > c0 ? s : c0, b = t < c1 ? t : c1;
    for (unsigned i = a; i < b; ++i) {
      const PodRec pr = s_pod[i - c0];
      const unsigned pf = pr.flags;
      if (clobber_synthetic && (pf & 1u)) continue;  // (remove synthetic-pod?) (:899)
      if (pf & 2u) continue;                         // nil resource map: (remove nil?) (:919)
      const double pc = pr.cpus, pm = pr.mem;
      if (!any) {  // merge-with keeps the first value as it is and adds from the second on
        cc = pc;
        cm = pm;
        any = true;
      } else {
        cc = cc + pc;
        cm = cm + pm;
      }
      const int pg = pr.gpus;
      const unsigned pgm = pr.gpu_model;
      if (pgm != 0u && pg > 0) {
        if (cap_has_gpu && pgm == node_gm) {
          own_g += pg;
        } else {
          bool found = false;
#pragma unroll
          for (int q = 0; q < ON_SLOTS; ++q)
            if ((unsigned)q < n_fg && fg_model[q] == pgm) fg_cnt[q] += pg, found = true;
          if (!found) {
            if (n_fg < fg_room) {
#pragma unroll
              for (int q = 0; q < ON_SLOTS; ++q)
                if ((unsigned)q == n_fg) fg_model[q] = pgm, fg_cnt[q] = pg;
              ++n_fg;
            } else {
              foreign_g = true;
            }
          }
        }
        if (gpu_cons_by_model && pgm <= n_gpu_models) atomicAdd(lds_models ? &s_gcons[pgm] : &gpu_cons_by_model[pgm], (unsigned long long)pg);
      }
      const double pdk = pr.disk;
      const unsigned pdt = pr.disk_type;
      if (pdk >= 0.0 && pdt != 0u) {
        if (cap_has_disk && pdt == node_dt) {
          own_d = any_d ? own_d + pdk : pdk;
          any_d = true;
        } else {
          bool found = false;
#pragma unroll
          for (int q = 0; q < ON_SLOTS; ++q)
            if ((unsigned)q < n_fd && fd_type[q] == pdt) fd_val[q] = fd_val[q] + pdk, found = true;  // merge-with +: first kept, then added
          if (!found) {
            if (n_fd < fd_room) {
#pragma unroll
              for (int q = 0; q < ON_SLOTS; ++q)
                if ((unsigned)q == n_fd) fd_type[q] = pdt, fd_val[q] = pdk;
              ++n_fd;
            } else {
              foreign_d = true;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  bool sched = false;
  if (live) {
    // ---- available = (deep-merge-with - capacity consumed) (compute_cluster.clj:91) ----
    out.cpus[v] = any ? cap_c - cc : cap_c;
    out.mem[v] = any ? cap_m - cm : cap_m;
    out.cons_cpus[v] = any ? cc : 0.0;
    out.cons_mem[v] = any ? cm : 0.0;
    {
      unsigned w = 0;
      if (cap_has_gpu) {
        out.gpu_model[(size_t)v * gpu_slots] = node_gm;
        out.gpu_count[(size_t)v * gpu_slots] = (double)((long long)node_g - own_g);
        w = 1;
      }
#pragma unroll
      for (int q = 0; q < ON_SLOTS; ++q)
        if ((unsigned)q < n_fg) out.gpu_model[(size_t)v * gpu_slots + w] = fg_model[q], out.gpu_count[(size_t)v * gpu_slots + w] = (double)fg_cnt[q], ++w;
      for (; w < gpu_slots; ++w) out.gpu_model[(size_t)v * gpu_slots + w] = 0u, out.gpu_count[(size_t)v * gpu_slots + w] = 0.0;
      w = 0;
      if (cap_has_disk) {
        out.disk_type[(size_t)v * disk_slots] = node_dt;
        out.disk_space[(size_t)v * disk_slots] = any_d ? node_d - own_d : node_d;
        w = 1;
      }
#pragma unroll
      for (int q = 0; q < ON_SLOTS; ++q)
        if ((unsigned)q < n_fd) out.disk_type[(size_t)v * disk_slots + w] = fd_type[q], out.disk_space[(size_t)v * disk_slots + w] = fd_val[q], ++w;
      for (; w < disk_slots; ++w) out.disk_type[(size_t)v * disk_slots + w] = 0u, out.disk_space[(size_t)v * disk_slots + w] = 0.0;
    }
    out.disk_cons[v] = (cap_has_disk && any_d) ? own_d : 0.0;
    if (cap_has_gpu && gpu_cap_by_model && node_gm <= n_gpu_models)
      atomicAdd(lds_models ? &s_gcap[node_gm] : &gpu_cap_by_model[node_gm], (unsigned long long)node_g);
    // ---- node-schedulable? (api.clj:782-847) ----
    const int npods = (int)(t - s);
    sched = !(nflags & 1u) && !(nflags & 2u) && npods < max_pods && !(nflags & 4u) &&
            !((nflags & 8u) && !(node_g > 0) && filter_unsound_gpu);
    out.num_pods[v] = npods;
    out.status[v] = (uint8_t)((sched ? 1u : 0u) | (any ? 2u : 0u) | (foreign_g ? 4u : 0u) | (foreign_d ? 8u : 0u));
  }
  // offers of this block of nodes (the compaction's first level)
  const unsigned long long bal = __ballot(sched);
  if (lane_id() == 0 && bal) atomicAdd(&s_cnt, (unsigned)__popcll(bal));
  __syncthreads();
  if (threadIdx.x == 0) block_offers[blockIdx.x] = s_cnt;
  if (lds_models && threadIdx.x <= n_gpu_models) {
    if (gpu_cap_by_model && s_gcap[threadIdx.x]) atomicAdd(&gpu_cap_by_model[threadIdx.x], s_gcap[threadIdx.x]);
    if (gpu_cons_by_model && s_gcons[threadIdx.x]) atomicAdd(&gpu_cons_by_model[threadIdx.x], s_gcons[threadIdx.x]);
  }
}

struct OfferRows {  // compacted offer rows (device)
  uint32_t* node;
  uint32_t* host;
  double* cpus;
  double* mem;
  uint32_t* gpu_model;
  double* gpu_count;
  uint32_t* disk_type;
  double* disk_space;
  int32_t* num_pods;
  uint32_t* attr;
};

// offer rows of the schedulable nodes, in node order (compute_cluster.clj:163-190).  Order-preserving compaction without a
// scan launch: a block sums the offer counts of the blocks before it (a few hundred words), ranks its own nodes by ballots.
__global__ void __launch_bounds__(256) offers_emit(NodeCols nd, const uint32_t* __restrict__ node_host, NodeAvail av,
                                                   const uint32_t* __restrict__ block_offers, const uint32_t* __restrict__ node_attr,
                                                   unsigned n_attr, unsigned gpu_slots, unsigned disk_slots, OfferRows o,
                                                   unsigned* __restrict__ total) {
  __shared__ unsigned s_base, s_wave[4];
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  unsigned part = 0;
  for (unsigned b = threadIdx.x; b < blockIdx.x; b += blockDim.x) part += block_offers[b];
  for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, COOK_WAVE);
  if (lane_id() == 0 && part) atomicAdd(&s_base, part);
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  const bool mine = v < nd.n && (av.status[v] & 1u);
  const unsigned long long bal = __ballot(mine);
  if (lane_id() == 0) s_wave[wave_id()] = (unsigned)__popcll(bal);
  __syncthreads();
  unsigned wbase = s_base;
  for (unsigned k = 0; k < wave_id(); ++k) wbase += s_wave[k];
  if (blockIdx.x + 1 == gridDim.x && threadIdx.x == 0) *total = s_base + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  if (!mine) return;
  const unsigned r = wbase + (unsigned)__popcll(bal & lanemask_lt());
  o.node[r] = v;
  o.host[r] = node_host ? node_host[v] : v;
  o.cpus[r] = clj_max0(av.cpus[v]);
  o.mem[r] = clj_max0(av.mem[v]);
  for (unsigned q = 0; q < gpu_slots; ++q) {
    o.gpu_model[(size_t)r * gpu_slots + q] = av.gpu_model[(size_t)v * gpu_slots + q];
    o.gpu_count[(size_t)r * gpu_slots + q] = av.gpu_count[(size_t)v * gpu_slots + q];
  }
  for (unsigned q = 0; q < disk_slots; ++q) {
    o.disk_type[(size_t)r * disk_slots + q] = av.disk_type[(size_t)v * disk_slots + q];
    o.disk_space[(size_t)r * disk_slots + q] = av.disk_space[(size_t)v * disk_slots + q];
  }
  o.num_pods[r] = av.num_pods[v];
  for (unsigned q = 0; q < n_attr; ++q) o.attr[(size_t)r * n_attr + q] = node_attr[(size_t)v * n_attr + q];
}

// The gauges (compute_cluster.clj:113-160): fp64 totals in NODE ORDER, left to right, so that they are reproducible for
// non-dyadic values (a 0.1-cpu request makes every partial sum round; a left-to-right fp64 sum is a dependent chain by
// definition).  One workgroup; it stages tiles of OT_TILE nodes of OT_Q quantities in LDS (coalesced loads, the next tile
// in flight during the fold), wave q folds quantity q over the tile reading LDS broadcasts: the chain costs one fp64 add latency
// per node and the chains of the quantities run on different waves.  Quantities: 0..3 = cpus / mem capacity, cpus / mem
// consumed; 4 + 2 t, 5 + 2 t = capacity / consumption of disk type 1 + t (further passes of OT_Q).  Accumulators start at -0.0, the additive identity of
// round-to-nearest fp64, so the first addition reproduces (reduce + coll)'s "start from the first element".
struct OfferTotalsDev {
  double cpus_capacity, mem_capacity, cpus_consumed, mem_consumed;
};
constexpr int OT_TILE = 1024, OT_Q = 8, OT_THREADS = 1024;  // quantities per pass, one folding wave each
static __device__ __forceinline__ double offers_total_value(const NodeCols& nd, const NodeAvail& av, unsigned q, unsigned v) {
  if (q == 0) return nd.cpus[v];
  if (q == 1) return nd.mem[v];
  if (q < 4) return (av.status[v] & 2u) ? (q == 2 ? av.cons_cpus[v] : av.cons_mem[v]) : 0.0;  // absent entries add 0.0 (exact)
  const unsigned ty = 1u + (q - 4u) / 2u;
  const double nd_d = nd.disk ? nd.disk[v] : -1.0;
  const bool mine = nd_d >= 0.0 && nd.disk_type && nd.disk_type[v] == ty;
  return mine ? (((q - 4u) & 1u) ? av.disk_cons[v] : nd_d) : 0.0;
}
__global__ void __launch_bounds__(OT_THREADS) offers_totals(NodeCols nd, NodeAvail av, unsigned n_disk_types, OfferTotalsDev* __restrict__ tot,
                                                            double* __restrict__ disk_cap_by_type, double* __restrict__ disk_cons_by_type) {
  __shared__ double s_val[2][OT_Q][OT_TILE];  // double-buffered: tile k+1 is fetched while the folding waves walk tile k
  const unsigned w = wave_id();
  const unsigned n = nd.n;
  const bool want_disk = disk_cap_by_type || disk_cons_by_type;
  const unsigned n_q = 4u + (want_disk ? 2u * n_disk_types : 0u);
  const unsigned n_tiles = (n + OT_TILE - 1) / OT_TILE;
  for (unsigned q0 = 0; q0 < n_q; q0 += OT_Q) {
    const unsigned nq = n_q - q0 < (unsigned)OT_Q ? n_q - q0 : (unsigned)OT_Q;
    double acc = -0.0;
    double r[OT_Q];  // thread i of the workgroup fetches element i of every quantity of a tile
    auto fetch = [&](unsigned tile) {
      const unsigned v = tile * OT_TILE + threadIdx.x;
#pragma unroll
      for (int ql = 0; ql < OT_Q; ++ql) r[ql] = ((unsigned)ql < nq && v < n) ? offers_total_value(nd, av, q0 + ql, v) : 0.0;
    };
    auto stash = [&](unsigned buf) {
#pragma unroll
      for (int ql = 0; ql < OT_Q; ++ql) s_val[buf][ql][threadIdx.x] = r[ql];
    };
    if (n_tiles) {
      fetch(0);
      stash(0);
    }
    __syncthreads();
    for (unsigned tile = 0; tile < n_tiles; ++tile) {
      const unsigned buf = tile & 1u;
      if (tile + 1 < n_tiles) fetch(tile + 1);  // global loads in flight during the fold
      if (w < nq) {
        const unsigned cnt = n - tile * OT_TILE < (unsigned)OT_TILE ? n - tile * OT_TILE : (unsigned)OT_TILE;
        acc = fold_lds_in_order(&s_val[buf][w][0], cnt, acc);
      }
      if (tile + 1 < n_tiles) stash(buf ^ 1u);
      __syncthreads();
    }
    if (w < nq && lane_id() == 0) {
      const unsigned q = q0 + w;
      const double res = acc + 0.0;  // an empty node map sums to 0
      if (q == 0) tot->cpus_capacity = res;
      else if (q == 1) tot->mem_capacity = res;
      else if (q == 2) tot->cpus_consumed = res;
      else if (q == 3) tot->mem_consumed = res;
      else if ((q - 4u) & 1u) {
        if (disk_cons_by_type) disk_cons_by_type[1u + (q - 4u) / 2u] = res;
      } else {
        if (disk_cap_by_type) disk_cap_by_type[1u + (q - 4u) / 2u] = res;
      }
    }
    __syncthreads();
  }
}

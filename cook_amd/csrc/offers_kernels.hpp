// offers_kernels.hpp — offer construction from node state (the step before the match path): the numeric core of
// kubernetes.compute-cluster/generate-offers (kubernetes/compute_cluster.clj:68-190) with api/get-capacity
// (kubernetes/api.clj:874-884), api/get-consumption (:886-930) and the boolean of api/node-schedulable? (:782-847).
//
// HBM-bound streaming work: pods are stably partitioned by node (one radix sort of a u32 permutation), every node then
// folds its pods LEFT TO RIGHT in the list order the reference's (apply deep-merge-with + ...) uses, so the fp64 sums are
// bit-identical for any requests (0.1-cpu pods are not dyadic: no re-association is allowed).  Parallelism is across nodes.
#pragma once
#include "common.hpp"

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
  double* gpu_count;
  double* disk_space;
  double* disk_cons;  // consumption under the node's own disk type
  uint32_t* flag;     // 1 = schedulable (input of the compaction scan)
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

// Clojure's (max 0.0 x) on a boxed double (clojure.lang.Numbers/max): NaN propagates, -0.0 survives
static __host__ __device__ __forceinline__ double clj_max0(double x) { return 0.0 > x ? 0.0 : x; }

// One lane per node: capacity (get-capacity), consumption (get-consumption: the node's pods folded in list order),
// available = deep-merge-with - (compute_cluster.clj:91), node-schedulable?.
__global__ void __launch_bounds__(256) offers_node_eval(NodeCols nd, PodCols pd, const uint32_t* __restrict__ perm,
                                                        const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_end,
                                                        int clobber_synthetic, int filter_unsound_gpu, int max_pods, unsigned n_gpu_models,
                                                        NodeAvail out, unsigned long long* __restrict__ gpu_cap_by_model,
                                                        unsigned long long* __restrict__ gpu_cons_by_model) {
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nd.n) return;
  const unsigned s = seg_start[v], t = seg_end[v];
  // ---- capacity (api.clj:849-884) ----
  const double cap_c = nd.cpus[v], cap_m = nd.mem[v];
  const int node_g = nd.gpus ? nd.gpus[v] : 0;
  const unsigned node_gm = nd.gpu_model ? nd.gpu_model[v] : 0u;
  const bool cap_has_gpu = node_gm != 0u && node_g > 0;  // force-gpu-model-in-resource-map (:849-855)
  const double node_d = nd.disk ? nd.disk[v] : -1.0;
  const unsigned node_dt = nd.disk_type ? nd.disk_type[v] : 0u;
  const bool cap_has_disk = node_d >= 0.0 && node_dt != 0u;  // force-disk-type-in-resource-map (:866-872)
  const unsigned nflags = nd.flags ? nd.flags[v] : 0u;
  // ---- consumption (api.clj:886-930) ----
  bool any = false, any_d = false, foreign_g = false, foreign_d = false;
  double cc = 0.0, cm = 0.0, own_d = 0.0;
  long long own_g = 0;
  for (unsigned i = s; i < t; ++i) {
    const unsigned p = perm[i];
    const unsigned pf = pd.flags ? pd.flags[p] : 0u;
    if (clobber_synthetic && (pf & 1u)) continue;  // (remove synthetic-pod?) (:899)
    if (pf & 2u) continue;                         // nil resource map: (remove nil?) (:919)
    const double pc = pd.cpus[p], pm = pd.mem[p];
    if (!any) {  // merge-with keeps the first value as it is and adds from the second on
      cc = pc;
      cm = pm;
      any = true;
    } else {
      cc = cc + pc;
      cm = cm + pm;
    }
    const int pg = pd.gpus ? pd.gpus[p] : 0;
    const unsigned pgm = pd.gpu_model ? pd.gpu_model[p] : 0u;
    if (pgm != 0u && pg > 0) {
      if (cap_has_gpu && pgm == node_gm) own_g += pg;
      else foreign_g = true;
      if (gpu_cons_by_model && pgm <= n_gpu_models) atomicAdd(&gpu_cons_by_model[pgm], (unsigned long long)pg);
    }
    const double pdk = pd.disk ? pd.disk[p] : -1.0;
    const unsigned pdt = pd.disk_type ? pd.disk_type[p] : 0u;
    if (pdk >= 0.0 && pdt != 0u) {
      if (cap_has_disk && pdt == node_dt) {
        own_d = any_d ? own_d + pdk : pdk;
        any_d = true;
      } else {
        foreign_d = true;
      }
    }
  }
  // ---- available = (deep-merge-with - capacity consumed) (compute_cluster.clj:91) ----
  out.cpus[v] = any ? cap_c - cc : cap_c;
  out.mem[v] = any ? cap_m - cm : cap_m;
  out.cons_cpus[v] = any ? cc : 0.0;
  out.cons_mem[v] = any ? cm : 0.0;
  out.gpu_count[v] = cap_has_gpu ? (double)((long long)node_g - own_g) : 0.0;
  out.disk_space[v] = cap_has_disk ? (any_d ? node_d - own_d : node_d) : 0.0;
  out.disk_cons[v] = (cap_has_disk && any_d) ? own_d : 0.0;
  if (cap_has_gpu && gpu_cap_by_model && node_gm <= n_gpu_models) atomicAdd(&gpu_cap_by_model[node_gm], (unsigned long long)node_g);
  // ---- node-schedulable? (api.clj:782-847) ----
  const int npods = (int)(t - s);
  const bool sched = !(nflags & 1u) && !(nflags & 2u) && npods < max_pods && !(nflags & 4u) &&
                     !((nflags & 8u) && !(node_g > 0) && filter_unsound_gpu);
  out.flag[v] = sched ? 1u : 0u;
  out.num_pods[v] = npods;
  out.status[v] = (uint8_t)((sched ? 1u : 0u) | (any ? 2u : 0u) | (foreign_g ? 4u : 0u) | (foreign_d ? 8u : 0u));
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

// offer rows of the schedulable nodes, in node order (compute_cluster.clj:163-190); pos = exclusive scan of flag
__global__ void __launch_bounds__(256) offers_emit(NodeCols nd, const uint32_t* __restrict__ node_host, NodeAvail av,
                                                   const uint32_t* __restrict__ pos, const uint32_t* __restrict__ node_attr, unsigned n_attr,
                                                   OfferRows o) {
  const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nd.n || !(av.status[v] & 1u)) return;
  const unsigned r = pos[v];
  o.node[r] = v;
  o.host[r] = node_host ? node_host[v] : v;
  o.cpus[r] = clj_max0(av.cpus[v]);
  o.mem[r] = clj_max0(av.mem[v]);
  const int node_g = nd.gpus ? nd.gpus[v] : 0;
  const unsigned node_gm = nd.gpu_model ? nd.gpu_model[v] : 0u;
  const bool cap_has_gpu = node_gm != 0u && node_g > 0;
  o.gpu_model[r] = cap_has_gpu ? node_gm : 0u;
  o.gpu_count[r] = av.gpu_count[v];
  const double node_d = nd.disk ? nd.disk[v] : -1.0;
  const unsigned node_dt = nd.disk_type ? nd.disk_type[v] : 0u;
  const bool cap_has_disk = node_d >= 0.0 && node_dt != 0u;
  o.disk_type[r] = cap_has_disk ? node_dt : 0u;
  o.disk_space[r] = av.disk_space[v];
  o.num_pods[r] = av.num_pods[v];
  for (unsigned q = 0; q < n_attr; ++q) o.attr[(size_t)r * n_attr + q] = node_attr[(size_t)v * n_attr + q];
}

// The gauges (compute_cluster.clj:113-160): fp64 totals in NODE ORDER, left to right, so that they are reproducible for
// non-dyadic values (a 0.1-cpu request makes every partial sum round).  One workgroup; wave w owns one chain of sums:
// wave 0 = {cpus, mem} capacity and consumption, waves 1.. = disk capacity / consumption of one disk type each.  A wave loads
// 64 nodes coalesced, then folds them in lane order through v_readlane (a chunk = 64 dependent adds per chain, the chains of
// a wave interleave).  Accumulators start at -0.0, the additive identity of round-to-nearest fp64, which makes the first
// addition reproduce (reduce + coll)'s "start from the first element".
struct OfferTotalsDev {
  double cpus_capacity, mem_capacity, cpus_consumed, mem_consumed;
};
template <int NV>
static __device__ __forceinline__ void offers_fold_chunk(double (&acc)[NV], const double (&x)[NV], unsigned cnt) {
  if (cnt == (unsigned)COOK_WAVE) {
#pragma unroll
    for (int k = 0; k < COOK_WAVE; ++k) {
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = acc[q] + wave_read_lane_f64(x[q], k);
    }
  } else {
    for (unsigned k = 0; k < cnt; ++k) {
#pragma unroll
      for (int q = 0; q < NV; ++q) acc[q] = acc[q] + wave_read_lane_f64(x[q], (int)k);
    }
  }
}
__global__ void __launch_bounds__(1024) offers_totals(NodeCols nd, NodeAvail av, unsigned n_disk_types, OfferTotalsDev* __restrict__ tot,
                                                      double* __restrict__ disk_cap_by_type, double* __restrict__ disk_cons_by_type) {
  const unsigned lane = lane_id(), w = wave_id(), nw = blockDim.x / COOK_WAVE;
  const unsigned n = nd.n;
  if (w == 0) {
    double acc[4] = {-0.0, -0.0, -0.0, -0.0};
    for (unsigned base = 0; base < n; base += COOK_WAVE) {
      const unsigned v = base + lane;
      const bool in = v < n;
      const bool has = in && (av.status[v] & 2u);
      const double x[4] = {in ? nd.cpus[v] : 0.0, in ? nd.mem[v] : 0.0, has ? av.cons_cpus[v] : 0.0, has ? av.cons_mem[v] : 0.0};
      offers_fold_chunk<4>(acc, x, n - base < (unsigned)COOK_WAVE ? n - base : (unsigned)COOK_WAVE);
    }
    if (lane == 0) {
      tot->cpus_capacity = acc[0] + 0.0;  // an empty node map sums to 0
      tot->mem_capacity = acc[1] + 0.0;
      tot->cpus_consumed = acc[2] + 0.0;
      tot->mem_consumed = acc[3] + 0.0;
    }
    return;
  }
  if (!disk_cap_by_type && !disk_cons_by_type) return;
  for (unsigned ty = w; ty <= n_disk_types; ty += nw - 1) {  // disk type ids 1..n_disk_types
    double acc[2] = {-0.0, -0.0};
    for (unsigned base = 0; base < n; base += COOK_WAVE) {
      const unsigned v = base + lane;
      const bool in = v < n;
      const double nd_d = (in && nd.disk) ? nd.disk[v] : -1.0;
      const unsigned nd_t = (in && nd.disk_type) ? nd.disk_type[v] : 0u;
      const bool mine = nd_d >= 0.0 && nd_t == ty;
      const double x[2] = {mine ? nd_d : 0.0, mine ? av.disk_cons[v] : 0.0};
      offers_fold_chunk<2>(acc, x, n - base < (unsigned)COOK_WAVE ? n - base : (unsigned)COOK_WAVE);
    }
    if (lane == 0) {
      if (disk_cap_by_type) disk_cap_by_type[ty] = acc[0] + 0.0;
      if (disk_cons_by_type) disk_cons_by_type[ty] = acc[1] + 0.0;
    }
  }
}

// cycle_update.hpp — cook_cycle_update: what changed between two match cycles of a pool whose inputs are resident in HBM.
//
// handle-resource-offers! (scheduler.clj:1339-1385) sees, cycle after cycle, almost the same pool: a few tasks finished or were
// killed, a few jobs were submitted or launched, the offers are new.  cook_cycle_stage copies everything (140 MB for the benchmark
// cluster); this entry point takes the DELTA — task rows to remove, task / pending-job rows to append, optionally a fresh set of
// offers — and edits the resident columns on the device: a stable compaction (rows keep their relative order, so task indices stay
// meaningful to the host: removed rows close up, new rows go to the end) followed by the new rows.
//
// The call is a chain of small launches and copies, and that — not the bytes — is what it costs (rounds 2-4: one compaction launch
// and one copy per column, four read-backs: 110 launches and copies, 0.64 ms per pool, eight pools 5-6.7 ms).  Now:
//   * everything the delta brings is packed by the host into ONE page-locked block and copied once; the kernels read it there;
//   * ALL task columns are compacted by one launch and all pending-job columns by another (a table of column descriptors in the
//     kernel arguments), into second buffers that are swapped in only when the call has succeeded — a removal list that names a row
//     twice or out of range leaves the resident state as it was;
//   * row counts stay on the device until one read-back at the end (buffers are sized by bounds the host knows whatever the removal
//     list holds: every old row plus the new ones, every old value of a CSR column plus the new ones);
//   * the fresh offers are one block too, used in place (offers_block_plan / offers_block_commit).
// Included by engine.hip (uses its DArr / KL / seg_scan helpers).
#pragma once

constexpr unsigned UPD_MAX_COLS = 24;
struct UpdCol {
  const void* src;  // resident column (old rows)
  void* dst;        // its second buffer
  const void* add;  // the new rows in the delta block (device), or null: dflt
  unsigned long long dflt;  // bit pattern of the default element
  unsigned esize;           // 1, 4 or 8
};
struct UpdColSet {
  UpdCol c[UPD_MAX_COLS];
  unsigned n_cols;
};
struct UpdOut {  // what the host reads back at the end
  unsigned n_keep, p_keep, bad, kept_vals[2];
};

static __device__ __forceinline__ void upd_copy_elem(void* dst, size_t di, const void* src, size_t si, unsigned esize) {
  if (esize == 8) ((uint64_t*)dst)[di] = ((const uint64_t*)src)[si];
  else if (esize == 4) ((uint32_t*)dst)[di] = ((const uint32_t*)src)[si];
  else ((uint8_t*)dst)[di] = ((const uint8_t*)src)[si];
}
static __device__ __forceinline__ void upd_store_elem(void* dst, size_t di, unsigned long long v, unsigned esize) {
  if (esize == 8) ((uint64_t*)dst)[di] = v;
  else if (esize == 4) ((uint32_t*)dst)[di] = (uint32_t)v;
  else ((uint8_t*)dst)[di] = (uint8_t)v;
}

// rm[i] = 1 for the rows the delta removes (rm zeroed before); out->bad counts indices out of range or named twice
__global__ void __launch_bounds__(256) upd_mark_removed(const uint32_t* __restrict__ rem, unsigned n_rem, unsigned n, int* __restrict__ rm,
                                                        UpdOut* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rem) return;
  const unsigned r = rem[i];
  if (r >= n || atomicExch(&rm[r], 1) != 0) atomicAdd(&out->bad, 1u);
}
struct LoadKeep {  // 1 for a row that stays
  const int* rm;
  __device__ __forceinline__ SumI operator()(unsigned i) const { return SumI{rm[i] ? 0 : 1}; }
};
// removal flags of the pending jobs (by pending ordinal) from those of their tasks (rm_p zeroed before)
__global__ void __launch_bounds__(256) upd_pending_removed(const uint8_t* __restrict__ pending, const uint32_t* __restrict__ pend_ord,
                                                           const int* __restrict__ rm, unsigned n, int* __restrict__ rm_p) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && pending[i] && rm[i]) rm_p[pend_ord[i]] = 1;
}
// every column of one table at once: thread i < n_old moves row i (if it stays) to its new place, thread n_old + r appends new row r
__global__ void __launch_bounds__(256) upd_compact_cols(UpdColSet cs, const int* __restrict__ rm, const SumI* __restrict__ incl, unsigned n_old,
                                                        unsigned n_add, unsigned which /*0 tasks, 1 pending jobs*/, UpdOut* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned n_keep = n_old ? (unsigned)incl[n_old - 1].v : 0u;
  if (i == 0) (which ? out->p_keep : out->n_keep) = n_keep;
  if (i < n_old) {
    if (rm[i]) return;
    const unsigned o = (unsigned)incl[i].v - 1u;
    for (unsigned k = 0; k < cs.n_cols; ++k) upd_copy_elem(cs.c[k].dst, o, cs.c[k].src, i, cs.c[k].esize);
  } else if (i < n_old + n_add) {
    const unsigned r = i - n_old;
    for (unsigned k = 0; k < cs.n_cols; ++k) {
      if (cs.c[k].add) upd_copy_elem(cs.c[k].dst, (size_t)n_keep + r, cs.c[k].add, r, cs.c[k].esize);
      else upd_store_elem(cs.c[k].dst, (size_t)n_keep + r, cs.c[k].dflt, cs.c[k].esize);
    }
  }
}
struct LoadPendingFlag {
  const uint8_t* pending;
  __device__ __forceinline__ SumI operator()(unsigned i) const { return SumI{pending[i] ? 1 : 0}; }
};
__global__ void __launch_bounds__(256) upd_pend_ord(const uint8_t* __restrict__ pending, const SumI* __restrict__ incl, unsigned n,
                                                    uint32_t* __restrict__ pend_ord) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pend_ord[i] = (uint32_t)(incl[i].v - (pending[i] ? 1 : 0));
}
// a CSR column pair (offsets per pending job, one or two value arrays): lengths of the rows that stay
struct LoadCsrLen {
  const uint32_t* off;
  const int* rm_p;
  __device__ __forceinline__ SumI operator()(unsigned i) const { return SumI{rm_p[i] ? 0 : (int)(off[i + 1] - off[i])}; }
};
// thread i < p_old: row i (if it stays) -> its new offset and its values; thread p_old + r, r <= p_add: offset of new row r (shifted by
// the values that stay; r == p_add: the end); thread p_old + p_add + 1 + x: new value x
__global__ void __launch_bounds__(256) upd_csr(const uint32_t* __restrict__ off, const int* __restrict__ rm_p, const SumI* __restrict__ row_incl,
                                               const SumI* __restrict__ len_incl, unsigned p_old, const uint32_t* __restrict__ a,
                                               const uint32_t* __restrict__ b, const uint32_t* __restrict__ add_off, const uint32_t* __restrict__ add_a,
                                               const uint32_t* __restrict__ add_b, unsigned p_add, unsigned add_vals, uint32_t* __restrict__ off_out,
                                               uint32_t* __restrict__ a_out, uint32_t* __restrict__ b_out, unsigned which, UpdOut* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned p_keep = p_old ? (unsigned)row_incl[p_old - 1].v : 0u;
  const unsigned kept_vals = p_old ? (unsigned)len_incl[p_old - 1].v : 0u;
  if (i == 0) out->kept_vals[which] = kept_vals;
  if (i < p_old) {
    if (rm_p[i]) return;
    const unsigned len = off[i + 1] - off[i];
    const unsigned dst = (unsigned)len_incl[i].v - len;
    off_out[(unsigned)row_incl[i].v - 1u] = dst;
    for (unsigned x = 0; x < len; ++x) {
      a_out[dst + x] = a[off[i] + x];
      if (b) b_out[dst + x] = b[off[i] + x];
    }
  } else if (i <= p_old + p_add) {
    const unsigned r = i - p_old;
    off_out[p_keep + r] = kept_vals + (add_off ? add_off[r] : 0u);
  } else if (i < p_old + p_add + 1u + add_vals) {
    const unsigned x = i - (p_old + p_add + 1u);
    a_out[kept_vals + x] = add_a[x];
    if (b_out && add_b) b_out[kept_vals + x] = add_b[x];
  }
}

struct UpdateBufs {
  DArr<int> rm, rm_p;
  DArr<SumI> incl, incl_p, len_incl, fincl;
  DArr<UpdOut> out;
  DBuf alt[2 * UPD_MAX_COLS];  // second buffers of the columns (tasks: [0, UPD_MAX_COLS), pending jobs behind them)
  DArr<uint32_t> n_off[2], n_a[2], n_b[2], pend_ord_alt;
  unsigned csr_vals[2] = {0, 0};  // values the two CSR columns hold (eq, novel); 0xFFFFFFFF = not known yet (read from the offsets)
  bool csr_known = false;
  // the delta, packed: page-locked on the host, one copy to the device
  void* h_block = nullptr;
  size_t h_cap = 0;
  DBuf d_block;
  // the fresh offers, packed (used in place by the match until the next delta brings others)
  void* h_offers = nullptr;
  size_t h_offers_cap = 0;
  DBuf d_offers[2];    // two blocks, used in turn: a call that fails leaves the resident offers (in d_offers[d_offers_cur]) as they were
  int d_offers_cur = 0;
  ~UpdateBufs() {
    if (h_block) (void)hipHostFree(h_block);
    if (h_offers) (void)hipHostFree(h_offers);
  }
};

// (included inside engine.hip's anonymous namespace)
struct BlockWriter {  // lays arrays out in a host block, 16-byte aligned; first pass (base == nullptr) only measures
  char* base;
  size_t used = 0;
  explicit BlockWriter(char* b) : base(b) {}
  template <class T>
  size_t put(const T* src, size_t n) {  // returns the offset, or (size_t)-1 for a null source
    if (!src) return (size_t)-1;
    const size_t at = used;
    if (base && n) std::memcpy(base + at, src, n * sizeof(T));
    used = (used + n * sizeof(T) + 15) & ~(size_t)15;
    return at;
  }
};
// A block of page-locked HOST memory into device memory by a KERNEL that reads it over the link.  hipMemcpyAsync does these transfers on the
// SDMA engines, and once in a while such a copy — and every other one issued at that moment, from any thread — sits in the call for 7–8 ms
// (profiles/r05ac_update_outliers.txt: 3 of 300 updates of eight pools; none with HSA_ENABLE_SDMA=0, which costs every copy of the process its
// engine instead).  The blocks of a delta are a few hundred KB: a kernel moves them in the time the engine needs to start.
__global__ void __launch_bounds__(256) upd_block_in(uint4* __restrict__ dst, const uint4* __restrict__ src_host, unsigned n16) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) dst[i] = src_host[i];
}
static void block_to_device(cook_engine* e, void* dst, const void* src_host, size_t bytes) {  // (both sides hold 16 bytes beyond `bytes`)
  const unsigned n16 = (unsigned)((bytes + 15) / 16);
  if (n16) KL("upd_block_in", upd_block_in, std::min(div_up(n16, 256u), 512u), 256, (uint4*)dst, (const uint4*)src_host, n16);
}
static void pinned_reserve(void** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap) return;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = bytes + bytes / 2 + 4096;
  COOK_HIP(hipHostMalloc(p, want, hipHostMallocDefault));
  *cap = want;
}

// the offers of a delta: every column the host gave into one block, one copy, used where it lands.  Two steps so that
// cook_cycle_update stays all-or-nothing: offers_block_plan validates the struct and reserves every byte the staging needs (it may
// refuse or fail; nothing resident has been touched), offers_block_commit copies and points the match at the new block.
struct OffersPlan {
  size_t off[32];
  unsigned gs, ds, n_attr;
};
template <class W>
static void offers_block_layout(W& w, const cook_offers* o, OffersPlan& pl) {
  const unsigned M = o->n;
  int k = 0;
  pl.off[k++] = w.put(o->cpus, M);
  pl.off[k++] = w.put(o->mem, M);
  pl.off[k++] = w.put(o->host, M);
  pl.off[k++] = w.put(o->k8s, M);
  pl.off[k++] = w.put(o->gpu_model, (size_t)M * pl.gs);
  pl.off[k++] = w.put(o->gpu_count, (size_t)M * pl.gs);
  pl.off[k++] = w.put(o->disk_type, (size_t)M * pl.ds);
  pl.off[k++] = w.put(o->disk_space, (size_t)M * pl.ds);
  pl.off[k++] = w.put(o->ports, M);
  for (unsigned sc = 0; sc < COOK_MAX_SCALARS; ++sc)
    pl.off[k++] = w.put((o->scalars && sc < o->n_scalars) ? o->scalars + (size_t)sc * M : (const double*)nullptr, M);
  pl.off[k++] = w.put(o->attr, (size_t)M * pl.n_attr);
  pl.off[k++] = w.put(o->max_tasks, M);
  pl.off[k++] = w.put(o->num_tasks, M);
  pl.off[k++] = w.put(o->location, M);
  pl.off[k++] = w.put(o->host_start_s, M);
  pl.off[k++] = w.put(o->run_cpus, M);
  pl.off[k++] = w.put(o->run_mem, M);
  pl.off[k++] = w.put(o->run_count, M);
}
OffersPlan offers_block_plan(cook_engine* e, UpdateBufs& ub, const cook_offers* o) {
  const unsigned M = o->n;
  if (M && (!o->cpus || !o->mem || !o->host)) e->fail(COOK_E_INVALID, "cook_match_stage: offers need cpus, mem and host");
  if (o->scalars && o->n_scalars > COOK_MAX_SCALARS) e->fail(COOK_E_INVALID, "cook_match_stage: more than COOK_MAX_SCALARS named scalars");
  OffersPlan pl{};
  pl.gs = res_slots(e, o->gpu_slots, "cook_match_stage: gpu_slots");
  pl.ds = res_slots(e, o->disk_slots, "cook_match_stage: disk_slots");
  if (o->gpu_model && !o->gpu_count) e->fail(COOK_E_INVALID, "cook_match_stage: gpu_model without gpu_count");
  if (o->disk_type && !o->disk_space) e->fail(COOK_E_INVALID, "cook_match_stage: disk_type without disk_space");
  match_check_offer_count(e, M);  // the walk's owner table holds one byte per offer of the pool
  pl.n_attr = o->attr ? o->n_attr_keys : 0;
  BlockWriter w(nullptr);
  offers_block_layout(w, o, pl);
  pinned_reserve(&ub.h_offers, &ub.h_offers_cap, w.used + 16);
  ub.d_offers[ub.d_offers_cur ^ 1].ensure(w.used + 16);
  return pl;
}
void offers_block_commit(cook_engine* e, UpdateBufs& ub, const cook_offers* o, OffersPlan& pl) {
  MatchIn& in = e->min;
  const unsigned M = o->n;
  DBuf& blk = ub.d_offers[ub.d_offers_cur ^ 1];
  BlockWriter w((char*)ub.h_offers);
  offers_block_layout(w, o, pl);
  block_to_device(e, blk.p, ub.h_offers, w.used);
  ub.d_offers_cur ^= 1;
  const char* d = (const char*)blk.p;
  const size_t* off = pl.off;
  const unsigned gs = pl.gs, ds = pl.ds, n_attr = pl.n_attr;
  auto at = [&](size_t o_) -> const void* { return o_ == (size_t)-1 ? nullptr : (const void*)(d + o_); };
  int k = 0;
  in.M = M;
  e->M = M;
  in.o_cpus = (const double*)at(off[k++]);
  in.o_mem = (const double*)at(off[k++]);
  in.o_host = (const uint32_t*)at(off[k++]);
  in.o_k8s = (const uint8_t*)at(off[k++]);
  in.gpu_slots = gs;
  in.disk_slots = ds;
  in.o_gpu_model = (const uint32_t*)at(off[k++]);
  in.o_gpu_count = (const double*)at(off[k++]);
  in.o_disk_type = (const uint32_t*)at(off[k++]);
  in.o_disk_space = (const double*)at(off[k++]);
  in.o_ports = (const int32_t*)at(off[k++]);
  for (unsigned sc = 0; sc < COOK_MAX_SCALARS; ++sc) in.o_scal[sc] = (const double*)at(off[k++]);
  in.n_attr = n_attr;
  in.o_attr = (const uint32_t*)at(off[k++]);
  in.o_max_tasks = (const int32_t*)at(off[k++]);
  in.o_num_tasks = (const int32_t*)at(off[k++]);
  in.o_location = (const uint32_t*)at(off[k++]);
  in.o_host_start = (const int64_t*)at(off[k++]);
  in.o_run_cpus = (const double*)at(off[k++]);
  in.o_run_mem = (const double*)at(off[k++]);
  in.o_run_count = (const int32_t*)at(off[k++]);
  in.host_dup = 0;
  if (M) {  // two offers on one host?
    std::vector<uint32_t> hs(o->host, o->host + M);
    std::sort(hs.begin(), hs.end());
    in.host_dup = std::adjacent_find(hs.begin(), hs.end()) != hs.end() ? 1u : 0u;
  }
}

void cycle_update(cook_engine* e, UpdateBufs& ub, const cook_cycle_delta* d) {
  // the host's time by phase (cook_match_stats_ex [29..30]: the longest one): 0 checks, 1 the delta's block (host copies, one transfer), 2 marks + scans,
  // 3 column compactions, 4 the CSR columns (one stream synchronisation), 5 the look at the device (the other), 6 swaps + offers
  auto t_prev = std::chrono::steady_clock::now();
  for (unsigned& x : e->upd_phase_us) x = 0u;
  auto stamp = [&](int k) {
    const auto t = std::chrono::steady_clock::now();
    e->upd_phase_us[k] += (unsigned)std::chrono::duration<double, std::micro>(t - t_prev).count();
    t_prev = t;
  };
  if (!d) e->fail(COOK_E_INVALID, "cook_cycle_update: null delta");
  if (!e->cycle_staged) e->fail(COOK_E_STATE, "cook_cycle_update before cook_cycle_stage");
  const unsigned N = e->N, P = e->n_pending, U = e->U;
  const cook_tasks* at = d->add_tasks;
  const cook_jobs* aj = d->add_pending;
  const unsigned n_add = at ? at->n : 0u;
  if (n_add && (!at->cpus || !at->mem || !at->user || !at->priority || !at->start_ms || !at->task_id || !at->job_id || !at->pending))
    e->fail(COOK_E_INVALID, "cook_cycle_update: a required array of add_tasks is NULL");
  unsigned p_add = 0;
  for (unsigned i = 0; i < n_add; ++i) {
    if (at->user[i] >= U) e->fail(COOK_E_INVALID, "cook_cycle_update: user id out of range");
    p_add += at->pending[i] ? 1u : 0u;
  }
  if (p_add != (aj ? aj->n : 0u)) e->fail(COOK_E_INVALID, "cook_cycle_update: add_pending->n must equal the number of pending tasks of add_tasks");
  if (p_add && (!aj->cpus || !aj->mem)) e->fail(COOK_E_INVALID, "cook_cycle_update: add_pending needs cpus and mem");
  if (p_add && e->has_j_user && !aj->user)  // (the considerable filters read the staged jobs' users: a missing column would read as user 0)
    e->fail(COOK_E_INVALID, "cook_cycle_update: the staged jobs carry a user column, add_pending must too");
  for (unsigned r = 0; p_add && aj->ports && r < p_add; ++r)
    if (aj->ports[r] < 0) e->fail(COOK_E_INVALID, "cook_cycle_update: negative port count");
  if (d->n_remove && !d->remove_task) e->fail(COOK_E_INVALID, "cook_cycle_update: remove_task is NULL");
  if (d->n_remove && !N) e->fail(COOK_E_INVALID, "cook_cycle_update: nothing staged to remove from");
  if (d->n_remove > N) e->fail(COOK_E_INVALID, "cook_cycle_update: remove_task holds an index out of range or twice");
  MatchIn& in = e->min;
  // a column the delta brings but the stage did not have cannot be added row-wise: the host restages (cook_cycle_stage)
  if (p_add && ((aj->gpus && !in.j_gpus) || (aj->gpu_model && !in.j_gpu_model) || (aj->group && !in.j_group) || (aj->eq_off && !in.j_eq_off) ||
                (aj->novel_off && !in.j_novel_off) || (aj->reserved_host && !in.j_reserved_host) || (aj->ckpt_location && !in.j_ckpt) ||
                (aj->est_end_ms && !in.j_est_end) || (aj->disk_request && !in.j_disk_req) || (aj->user && !e->has_j_user) ||
                (aj->ports && !in.j_ports) || (aj->scalars && aj->n_scalars > in.n_scal)))
    e->fail(COOK_E_INVALID, "cook_cycle_update: add_pending carries a column the staged jobs do not have (restage with cook_cycle_stage)");
  if (n_add && at->gpus && !e->has_gpus) e->fail(COOK_E_INVALID, "cook_cycle_update: add_tasks carries gpus but the staged tasks do not");
  if (p_add && aj->group)
    for (unsigned r = 0; r < p_add; ++r)
      if (aj->group[r] != COOK_NONE_U32 && aj->group[r] >= e->G) e->fail(COOK_E_INVALID, "cook_cycle_update: group id out of range");
  if (in.j_disk_req && p_add && aj->disk_request && !aj->disk_type) e->fail(COOK_E_INVALID, "cook_cycle_update: disk_request without disk_type");
  // the CSR constraint columns of the delta: offsets from 0, non-decreasing, and values behind every non-empty list
  auto csr_ok = [&](const uint32_t* off, const void* a, const void* b, const char* what) {
    if (!off) return;
    if (off[0] != 0) e->fail(COOK_E_INVALID, what);
    for (unsigned r = 0; r < p_add; ++r)
      if (off[r + 1] < off[r]) e->fail(COOK_E_INVALID, what);
    if (off[p_add] && (!a || !b)) e->fail(COOK_E_INVALID, what);
  };
  if (p_add && in.j_eq_off) csr_ok(aj->eq_off, aj->eq_key, aj->eq_val, "cook_cycle_update: add_pending eq_off must start at 0 and not decrease, with eq_key / eq_val behind it");
  if (p_add && in.j_novel_off) csr_ok(aj->novel_off, aj->novel_host, aj->novel_host, "cook_cycle_update: add_pending novel_off must start at 0 and not decrease, with novel_host behind it");
  // the fresh offers are checked, and everything their staging needs is reserved, BEFORE anything resident changes: a refused
  // offers struct (or an allocation that fails) leaves tasks, jobs and offers as they were
  OffersPlan offers_plan{};
  if (d->offers) offers_plan = offers_block_plan(e, ub, d->offers);
  const unsigned N2 = N - d->n_remove + n_add;  // (when the removal list is valid; the device says at the end)
  const unsigned N_hi = N + n_add;              // rows the compaction can write whatever the list holds (an entry named twice removes one row)
  const unsigned P_hi = P + p_add;              // the pending jobs can only be bounded until then
  stamp(0);
  // ---- the delta as one block -----------------------------------------------------------------------------------------------
  struct Offs {
    size_t rem, t[9], j[12 + COOK_MAX_SCALARS], eq_off, eq_key, eq_val, nv_off, nv_host;
  } o{};
  const unsigned add_eq = (p_add && aj->eq_off && in.j_eq_off) ? aj->eq_off[p_add] : 0u;
  const unsigned add_nv = (p_add && aj->novel_off && in.j_novel_off) ? aj->novel_off[p_add] : 0u;
  for (int pass = 0; pass < 2; ++pass) {
    BlockWriter w(pass ? (char*)ub.h_block : nullptr);
    o.rem = w.put(d->remove_task, d->n_remove);
    if (n_add) {
      o.t[0] = w.put(at->cpus, n_add), o.t[1] = w.put(at->mem, n_add), o.t[2] = w.put(at->gpus, n_add), o.t[3] = w.put(at->user, n_add);
      o.t[4] = w.put(at->priority, n_add), o.t[5] = w.put(at->start_ms, n_add), o.t[6] = w.put(at->task_id, n_add);
      o.t[7] = w.put(at->job_id, n_add), o.t[8] = w.put(at->pending, n_add);
    }
    if (p_add) {
      int k = 0;
      o.j[k++] = w.put(aj->cpus, p_add), o.j[k++] = w.put(aj->mem, p_add), o.j[k++] = w.put(aj->gpus, p_add);
      o.j[k++] = w.put(aj->gpu_model, p_add), o.j[k++] = w.put(aj->user, p_add), o.j[k++] = w.put(aj->group, p_add);
      o.j[k++] = w.put(aj->reserved_host, p_add), o.j[k++] = w.put(aj->ckpt_location, p_add), o.j[k++] = w.put(aj->est_end_ms, p_add);
      o.j[k++] = w.put(aj->disk_request, p_add), o.j[k++] = w.put(aj->disk_type, p_add), o.j[k++] = w.put(aj->ports, p_add);
      for (unsigned sc = 0; sc < COOK_MAX_SCALARS; ++sc)
        o.j[k++] = w.put((aj->scalars && sc < aj->n_scalars) ? aj->scalars + (size_t)sc * p_add : (const double*)nullptr, p_add);
      o.eq_off = w.put(in.j_eq_off ? aj->eq_off : (const uint32_t*)nullptr, p_add + 1);
      o.eq_key = w.put(add_eq ? aj->eq_key : (const uint32_t*)nullptr, add_eq);
      o.eq_val = w.put(add_eq ? aj->eq_val : (const uint32_t*)nullptr, add_eq);
      o.nv_off = w.put(in.j_novel_off ? aj->novel_off : (const uint32_t*)nullptr, p_add + 1);
      o.nv_host = w.put(add_nv ? aj->novel_host : (const uint32_t*)nullptr, add_nv);
    }
    if (!pass) {
      pinned_reserve(&ub.h_block, &ub.h_cap, w.used + 16);
      ub.d_block.ensure(w.used + 16);
    } else if (w.used) {
      block_to_device(e, ub.d_block.p, ub.h_block, w.used);
    }
  }
  const char* blk = (const char*)ub.d_block.p;
  auto dev = [&](size_t off, bool have) -> const void* { return (!have || off == (size_t)-1) ? nullptr : (const void*)(blk + off); };
  stamp(1);
  // ---- which rows stay, and where they go -----------------------------------------------------------------------------------
  int* rm = ub.rm.ensure(std::max(1u, N));
  int* rm_p = ub.rm_p.ensure(std::max(1u, P));
  SumI* incl = ub.incl.ensure(std::max(1u, N));
  SumI* incl_p = ub.incl_p.ensure(std::max(1u, P));
  UpdOut* out = ub.out.ensure(1);
  memset_async(e, out, 0, sizeof(UpdOut));
  if (N) memset_async(e, rm, 0, (size_t)N * 4);
  if (P) memset_async(e, rm_p, 0, (size_t)P * 4);
  if (d->n_remove) {
    KL("upd_mark_removed", upd_mark_removed, div_up(d->n_remove, 256), 256, (const uint32_t*)dev(o.rem, true), d->n_remove, N, rm, out);
    if (P)
      KL("upd_pending_removed", upd_pending_removed, div_up(N, 256), 256, (const uint8_t*)e->t_pending.ptr(), (const uint32_t*)e->pend_ord.ptr(),
         (const int*)rm, N, rm_p);
  }
  if (N) seg_scan<SumI>(e, "upd_scan", LoadKeep{rm}, (const uint8_t*)nullptr, N, incl, e->tmpI);
  if (P) seg_scan<SumI>(e, "upd_scan", LoadKeep{rm_p}, (const uint8_t*)nullptr, P, incl_p, e->tmpI);
  stamp(2);
  // ---- the columns, into their second buffers ---------------------------------------------------------------------------------
  struct Swap {
    DBuf* col;
    DBuf* alt;
  };
  std::vector<Swap> swaps;
  auto add_col = [&](UpdColSet& cs, unsigned base, DBuf& col, unsigned esize, const void* add, unsigned long long dflt, size_t rows_hi) {
    if (cs.n_cols >= UPD_MAX_COLS) e->fail(COOK_E_STATE, "cook_cycle_update: column table full");
    DBuf& alt = ub.alt[base + cs.n_cols];
    alt.ensure(std::max<size_t>(1, rows_hi) * esize);
    cs.c[cs.n_cols++] = UpdCol{col.p, alt.p, add, dflt, esize};
    swaps.push_back(Swap{&col, &alt});
  };
  auto bits64 = [](double v) {
    unsigned long long b;
    std::memcpy(&b, &v, 8);
    return b;
  };
  UpdColSet ts{};
  add_col(ts, 0, e->t_cpus.b, 8, dev(o.t[0], n_add), 0, N_hi);
  add_col(ts, 0, e->t_mem.b, 8, dev(o.t[1], n_add), 0, N_hi);
  if (e->has_gpus) add_col(ts, 0, e->t_gpus.b, 8, dev(o.t[2], n_add), bits64(0.0), N_hi);
  add_col(ts, 0, e->t_user.b, 4, dev(o.t[3], n_add), 0, N_hi);
  add_col(ts, 0, e->t_prio.b, 4, dev(o.t[4], n_add), 0, N_hi);
  add_col(ts, 0, e->t_start.b, 8, dev(o.t[5], n_add), 0, N_hi);
  add_col(ts, 0, e->t_task.b, 8, dev(o.t[6], n_add), 0, N_hi);
  add_col(ts, 0, e->t_job.b, 8, dev(o.t[7], n_add), 0, N_hi);
  add_col(ts, 0, e->t_pending.b, 1, dev(o.t[8], n_add), 0, N_hi);
  if (N + n_add) KL("upd_compact_cols", upd_compact_cols, div_up(N + n_add, 256), 256, ts, (const int*)rm, (const SumI*)incl, N, n_add, 0u, out);
  // pending ordinals of the new task array: exclusive count of pending rows in front (reads the NEW pending column)
  uint32_t* pend_ord_new = ub.pend_ord_alt.ensure(std::max(1u, N_hi));
  const uint8_t* pending_new = (const uint8_t*)ub.alt[ts.n_cols - 1].p;
  if (N2) {
    SumI* fincl = ub.fincl.ensure(N2);
    seg_scan<SumI>(e, "upd_scan", LoadPendingFlag{pending_new}, (const uint8_t*)nullptr, N2, fincl, e->tmpI);
    KL("upd_pend_ord", upd_pend_ord, div_up(N2, 256), 256, pending_new, (const SumI*)fincl, N2, pend_ord_new);
  }
  UpdColSet js{};
  {
    int k = 0;
    const bool pa = p_add != 0;
    add_col(js, UPD_MAX_COLS, e->j_cpus.b, 8, dev(o.j[k++], pa), 0, P_hi);
    add_col(js, UPD_MAX_COLS, e->j_mem.b, 8, dev(o.j[k++], pa), 0, P_hi);
    if (in.j_gpus) add_col(js, UPD_MAX_COLS, e->j_gpus.b, 8, dev(o.j[k], pa), bits64(0.0), P_hi);
    ++k;
    if (in.j_gpu_model) add_col(js, UPD_MAX_COLS, e->j_gpu_model.b, 4, dev(o.j[k], pa), 0, P_hi);
    ++k;
    if (e->has_j_user) add_col(js, UPD_MAX_COLS, e->j_user.b, 4, dev(o.j[k], pa), 0, P_hi);
    ++k;
    if (in.j_group) add_col(js, UPD_MAX_COLS, e->j_group.b, 4, dev(o.j[k], pa), 0xFFFFFFFFull, P_hi);
    ++k;
    if (in.j_reserved_host) add_col(js, UPD_MAX_COLS, e->j_reserved_host.b, 4, dev(o.j[k], pa), 0xFFFFFFFFull /* -1 */, P_hi);
    ++k;
    if (in.j_ckpt) add_col(js, UPD_MAX_COLS, e->j_ckpt.b, 4, dev(o.j[k], pa), 0, P_hi);
    ++k;
    if (in.j_est_end) add_col(js, UPD_MAX_COLS, e->j_est_end.b, 8, dev(o.j[k], pa), 0, P_hi);
    ++k;
    if (in.j_disk_req) add_col(js, UPD_MAX_COLS, e->j_disk_req.b, 8, dev(o.j[k], pa), bits64(-1.0), P_hi);
    ++k;
    if (in.j_disk_req) add_col(js, UPD_MAX_COLS, e->j_disk_type.b, 4, dev(o.j[k], pa), 0, P_hi);
    ++k;
    if (in.j_ports) add_col(js, UPD_MAX_COLS, e->j_ports.b, 4, dev(o.j[k], pa), 0, P_hi);
    ++k;
    for (unsigned sc = 0; sc < COOK_MAX_SCALARS; ++sc, ++k)
      if (sc < in.n_scal) add_col(js, UPD_MAX_COLS, e->j_scal[sc].b, 8, dev(o.j[k], pa), bits64(__builtin_nan("")), P_hi);
    // the eligible mask of cook_cycle_set_considerable is indexed by pending ordinal like the job columns: it moves with them; the
    // jobs the delta adds are eligible until the host says otherwise (a fresh mask through cook_cycle_set_considerable)
    if (e->cb && e->cb->has_elig_by_pending) add_col(js, UPD_MAX_COLS, e->cb->elig_by_pending.b, 1, nullptr, 1ull, P_hi);
  }
  if (P + p_add) KL("upd_compact_cols", upd_compact_cols, div_up(P + p_add, 256), 256, js, (const int*)rm_p, (const SumI*)incl_p, P, p_add, 1u, out);
  stamp(3);
  // ---- the two CSR columns (EQUALS constraints; hosts to avoid) ----------------------------------------------------------------
  if (!ub.csr_known) {  // how many values the staged columns hold: the last offset (once per stage)
    ub.csr_vals[0] = ub.csr_vals[1] = 0;
    uint32_t* h = (uint32_t*)e->h_scratch;
    h[0] = h[1] = 0;
    if (in.j_eq_off && P) copy_async(e, &h[0], e->j_eq_off.ptr() + P, 4, hipMemcpyDeviceToHost);
    if (in.j_novel_off && P) copy_async(e, &h[1], e->j_novel_off.ptr() + P, 4, hipMemcpyDeviceToHost);
    sync(e);
    ub.csr_vals[0] = h[0], ub.csr_vals[1] = h[1];
    ub.csr_known = true;
  }
  auto csr = [&](unsigned which, DArr<uint32_t>& off, DArr<uint32_t>& va, DArr<uint32_t>* vb, size_t o_off, size_t o_a, size_t o_b, unsigned add_vals) {
    const unsigned old_vals = ub.csr_vals[which];
    SumI* len_incl = ub.len_incl.ensure(std::max(1u, P));
    uint32_t* n_off = ub.n_off[which].ensure((size_t)P_hi + 1);
    uint32_t* n_a = ub.n_a[which].ensure(std::max<size_t>(1, (size_t)old_vals + add_vals));
    uint32_t* n_b = vb ? ub.n_b[which].ensure(std::max<size_t>(1, (size_t)old_vals + add_vals)) : nullptr;
    if (P) seg_scan<SumI>(e, "upd_scan", LoadCsrLen{off.ptr(), rm_p}, (const uint8_t*)nullptr, P, len_incl, e->tmpI);
    KL("upd_csr", upd_csr, div_up(P + p_add + 1 + add_vals, 256), 256, (const uint32_t*)off.ptr(), (const int*)rm_p, (const SumI*)incl_p,
       (const SumI*)len_incl, P, (const uint32_t*)va.ptr(), vb ? (const uint32_t*)vb->ptr() : (const uint32_t*)nullptr,
       (const uint32_t*)dev(o_off, p_add != 0), (const uint32_t*)dev(o_a, add_vals != 0), (const uint32_t*)dev(o_b, add_vals != 0 && vb), p_add, add_vals,
       n_off, n_a, n_b, which, out);
  };
  if (in.j_eq_off) csr(0, e->j_eq_off, e->j_eq_key, &e->j_eq_val, o.eq_off, o.eq_key, o.eq_val, add_eq);
  if (in.j_novel_off) csr(1, e->j_novel_off, e->j_novel_host, nullptr, o.nv_off, o.nv_host, (size_t)-1, add_nv);
  stamp(4);
  // ---- the one look at the device --------------------------------------------------------------------------------------------
  UpdOut h{};
  copy_async(e, e->h_scratch, out, sizeof(UpdOut), hipMemcpyDeviceToHost);
  sync(e);
  std::memcpy(&h, e->h_scratch, sizeof(UpdOut));
  stamp(5);
  if (h.bad) e->fail(COOK_E_INVALID, "cook_cycle_update: remove_task holds an index out of range or twice");  // nothing was swapped in
  const unsigned n_keep = N ? h.n_keep : 0u, p_keep = P ? h.p_keep : 0u;
  if (n_keep + n_add != N2) e->fail(COOK_E_STATE, "cook_cycle_update: row count mismatch");
  const unsigned P2 = p_keep + p_add;
  for (const Swap& s : swaps) std::swap(s.col->p, s.alt->p), std::swap(s.col->cap, s.alt->cap);
  std::swap(e->pend_ord.b.p, ub.pend_ord_alt.b.p), std::swap(e->pend_ord.b.cap, ub.pend_ord_alt.b.cap);
  auto swap_arr = [](DArr<uint32_t>& a, DArr<uint32_t>& b) { std::swap(a.b.p, b.b.p), std::swap(a.b.cap, b.b.cap); };
  if (in.j_eq_off) {
    swap_arr(e->j_eq_off, ub.n_off[0]), swap_arr(e->j_eq_key, ub.n_a[0]), swap_arr(e->j_eq_val, ub.n_b[0]);
    ub.csr_vals[0] = h.kept_vals[0] + add_eq;
    in.j_eq_off = e->j_eq_off.ptr(), in.j_eq_key = e->j_eq_key.ptr(), in.j_eq_val = e->j_eq_val.ptr();
  }
  if (in.j_novel_off) {
    swap_arr(e->j_novel_off, ub.n_off[1]), swap_arr(e->j_novel_host, ub.n_a[1]);
    ub.csr_vals[1] = h.kept_vals[1] + add_nv;
    in.j_novel_off = e->j_novel_off.ptr(), in.j_novel_host = e->j_novel_host.ptr();
  }
  in.j_cpus = e->j_cpus.ptr();
  in.j_mem = e->j_mem.ptr();
  if (in.j_gpus) in.j_gpus = e->j_gpus.ptr();
  if (in.j_gpu_model) in.j_gpu_model = e->j_gpu_model.ptr();
  if (in.j_group) in.j_group = e->j_group.ptr();
  if (in.j_reserved_host) in.j_reserved_host = e->j_reserved_host.ptr();
  if (in.j_ckpt) in.j_ckpt = e->j_ckpt.ptr();
  if (in.j_est_end) in.j_est_end = e->j_est_end.ptr();
  if (in.j_disk_req) in.j_disk_req = e->j_disk_req.ptr(), in.j_disk_type = e->j_disk_type.ptr();
  if (in.j_ports) in.j_ports = e->j_ports.ptr();
  for (unsigned sc = 0; sc < in.n_scal; ++sc) in.j_scal[sc] = e->j_scal[sc].ptr();
  // ports / named scalars: a new job asking for one switches the extra resource tests on
  for (unsigned sc = 0; sc < in.n_scal && !in.has_x; ++sc) {
    const double* col = (aj && aj->scalars && sc < aj->n_scalars) ? aj->scalars + (size_t)sc * p_add : nullptr;
    for (unsigned r = 0; col && r < p_add && !in.has_x; ++r) in.has_x = col[r] == col[r];
  }
  for (unsigned r = 0; aj && aj->ports && r < p_add && !in.has_x; ++r) in.has_x = aj->ports[r] > 0;
  e->N = N2;
  e->pool_usage_known = false;
  e->n_pending = P2;
  e->Kjobs = P2;
  e->K = P2;
  in.K = P2;
  e->rank_done = false;
  e->match_done = false;
  e->has_deferred = false;
  if (d->offers) {
    offers_block_commit(e, ub, d->offers, offers_plan);
    sync(e);
  }
  stamp(6);
}

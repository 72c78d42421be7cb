// cycle_update.hpp — cook_cycle_update: what changed between two match cycles of a pool whose inputs are resident in HBM.
//
// handle-resource-offers! (scheduler.clj:1339-1385) sees, cycle after cycle, almost the same pool: a few tasks finished or were
// killed, a few jobs were submitted or launched, the offers are new.  cook_cycle_stage copies everything (140 MB for the benchmark
// cluster); this entry point takes the DELTA — task rows to remove, task / pending-job rows to append, optionally a fresh set of
// offers — and edits the resident columns on the device: a stable compaction (rows keep their relative order, so task indices stay
// meaningful to the host: removed rows close up, new rows go to the end) followed by the copy of the new rows only.
// Included by engine.hip (uses its DArr / KL / seg_scan helpers).
#pragma once

__global__ void __launch_bounds__(256) upd_fill_ones(int* p, unsigned n) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 1;
}
__global__ void __launch_bounds__(256) upd_mark_removed(const uint32_t* __restrict__ rem, unsigned n_rem, unsigned n, int* __restrict__ keep,
                                                        unsigned* __restrict__ bad) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rem) return;
  const unsigned r = rem[i];
  if (r >= n || atomicExch(&keep[r], 0) == 0) atomicAdd(bad, 1u);  // out of range, or named twice
}
// keep flags of the pending jobs (by pending ordinal) from the keep flags of their tasks
__global__ void __launch_bounds__(256) upd_pending_keep(const uint8_t* __restrict__ pending, const uint32_t* __restrict__ pend_ord,
                                                        const int* __restrict__ keep, unsigned n, int* __restrict__ keep_p) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && pending[i]) keep_p[pend_ord[i]] = keep[i];
}
template <class T>
__global__ void __launch_bounds__(256) upd_compact(const T* __restrict__ in, const int* __restrict__ keep, const SumI* __restrict__ incl,
                                                   unsigned n, T* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep[i]) out[(unsigned)incl[i].v - 1u] = in[i];
}
// rows of width `w` (the offer attribute table is not touched here; this is for per-job tables should one appear)
__global__ void __launch_bounds__(256) upd_csr_len(const uint32_t* __restrict__ off, const int* __restrict__ keep, unsigned n, int* __restrict__ len) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) len[i] = keep[i] ? (int)(off[i + 1] - off[i]) : 0;
}
// new offsets of the kept rows (exclusive prefix of their lengths) + the total behind the last one
__global__ void __launch_bounds__(256) upd_csr_off(const int* __restrict__ keep, const SumI* __restrict__ row_incl, const int* __restrict__ len,
                                                   const SumI* __restrict__ len_incl, unsigned n, uint32_t* __restrict__ off_out,
                                                   unsigned n_rows_out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && keep[i]) off_out[(unsigned)row_incl[i].v - 1u] = (uint32_t)(len_incl[i].v - len[i]);
  if (i == n - 1) off_out[n_rows_out] = (uint32_t)len_incl[i].v;
}
__global__ void __launch_bounds__(256) upd_csr_vals(const uint32_t* __restrict__ off, const int* __restrict__ keep, const int* __restrict__ len,
                                                    const SumI* __restrict__ len_incl, unsigned n, const uint32_t* __restrict__ a,
                                                    const uint32_t* __restrict__ b, uint32_t* __restrict__ a_out, uint32_t* __restrict__ b_out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !keep[i]) return;
  const unsigned dst = (unsigned)(len_incl[i].v - len[i]);
  for (unsigned x = 0; x < (unsigned)len[i]; ++x) {
    a_out[dst + x] = a[off[i] + x];
    if (b) b_out[dst + x] = b[off[i] + x];
  }
}
__global__ void __launch_bounds__(256) upd_pending_flag(const uint8_t* __restrict__ pending, unsigned n, int* __restrict__ flag) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = pending[i] ? 1 : 0;
}
__global__ void __launch_bounds__(256) upd_pend_ord(const int* __restrict__ flag, const SumI* __restrict__ incl, unsigned n, uint32_t* __restrict__ pend_ord) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pend_ord[i] = (uint32_t)(incl[i].v - flag[i]);
}
template <class T>
__global__ void __launch_bounds__(256) upd_fill(T* p, unsigned n, T v) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

struct UpdateBufs {
  DArr<int> keep, keep_p, len, flag;
  DArr<SumI> incl, incl_p, len_incl;
  DArr<unsigned> bad;
  DBuf tmp;  // the compacted copy of one column (swapped with the column afterwards)
  // buffers that used to be allocated and freed inside every call (hipFree synchronises the whole device: eight pools updating at once
  // serialised on it): the removal list and the new CSR arrays (swapped with the columns like `tmp`)
  DArr<uint32_t> rem, n_off, n_a, n_b;
};

// (included inside engine.hip's anonymous namespace)
// stable compaction of one resident column + the new rows behind it; `col` ends up with n_keep + n_add rows
template <class T>
void upd_column(cook_engine* e, UpdateBufs& ub, DArr<T>& col, const int* keep, const SumI* incl, unsigned n_old, unsigned n_keep,
                const T* add, unsigned n_add, bool have_default, T dflt) {
  const unsigned n_new = n_keep + n_add;
  ub.tmp.ensure((size_t)(n_new ? n_new : 1) * sizeof(T));
  T* out = (T*)ub.tmp.p;
  if (n_old) {
    auto k = upd_compact<T>;
    KL("upd_compact", k, div_up(n_old, 256), 256, (const T*)col.ptr(), keep, incl, n_old, out);
  }
  if (n_add) {
    if (add) {
      COOK_HIP(hipMemcpyAsync(out + n_keep, add, (size_t)n_add * sizeof(T), hipMemcpyHostToDevice, e->stream));
    } else if (have_default) {
      auto k = upd_fill<T>;
      KL("upd_fill", k, div_up(n_add, 256), 256, out + n_keep, n_add, dflt);
    }
  }
  std::swap(col.b.p, ub.tmp.p);
  std::swap(col.b.cap, ub.tmp.cap);
}

void upd_csr(cook_engine* e, UpdateBufs& ub, DArr<uint32_t>& off, DArr<uint32_t>& va, DArr<uint32_t>* vb, const int* keep_p, const SumI* incl_p,
             unsigned p_old, unsigned p_keep, const uint32_t* add_off, const uint32_t* add_a, const uint32_t* add_b, unsigned p_add) {
  // lengths of the kept rows -> new offsets; values gathered row by row; then the new rows with their offsets shifted
  const unsigned p_new = p_keep + p_add;
  int* len = ub.len.ensure(std::max(1u, p_old));
  SumI* len_incl = ub.len_incl.ensure(std::max(1u, p_old));
  unsigned kept_vals = 0;
  DArr<uint32_t>&n_off = ub.n_off, &n_a = ub.n_a, &n_b = ub.n_b;
  n_off.ensure(p_new + 1);
  if (p_old) {
    KL("upd_csr_len", upd_csr_len, div_up(p_old, 256), 256, (const uint32_t*)off.ptr(), keep_p, p_old, len);
    seg_scan<SumI>(e, "upd_scan", LoadI{len}, (const uint8_t*)nullptr, p_old, len_incl, e->tmpI);
    COOK_HIP(hipMemcpyAsync(e->h_scratch, &len_incl[p_old - 1], 4, hipMemcpyDeviceToHost, e->stream));
    sync(e);
    int t = 0;
    std::memcpy(&t, e->h_scratch, 4);
    kept_vals = (unsigned)t;
  }
  const unsigned add_vals = (p_add && add_off) ? add_off[p_add] : 0u;
  n_a.ensure(std::max(1u, kept_vals + add_vals));
  if (vb) n_b.ensure(std::max(1u, kept_vals + add_vals));
  if (p_old) {
    KL("upd_csr_off", upd_csr_off, div_up(p_old, 256), 256, keep_p, incl_p, (const int*)len, (const SumI*)len_incl, p_old, n_off.ptr(), p_keep);
    KL("upd_csr_vals", upd_csr_vals, div_up(p_old, 256), 256, (const uint32_t*)off.ptr(), keep_p, (const int*)len, (const SumI*)len_incl, p_old,
       (const uint32_t*)va.ptr(), vb ? (const uint32_t*)vb->ptr() : (const uint32_t*)nullptr, n_a.ptr(), vb ? n_b.ptr() : (uint32_t*)nullptr);
  } else {
    COOK_HIP(hipMemsetAsync(n_off.ptr(), 0, 4, e->stream));
  }
  std::vector<uint32_t> shifted(p_add + 1);
  for (unsigned r = 0; r <= p_add; ++r) shifted[r] = kept_vals + (add_off ? add_off[r] : 0u);
  COOK_HIP(hipMemcpyAsync(n_off.ptr() + p_keep, shifted.data(), (size_t)(p_add + 1) * 4, hipMemcpyHostToDevice, e->stream));
  if (add_vals) {
    COOK_HIP(hipMemcpyAsync(n_a.ptr() + kept_vals, add_a, (size_t)add_vals * 4, hipMemcpyHostToDevice, e->stream));
    if (vb) COOK_HIP(hipMemcpyAsync(n_b.ptr() + kept_vals, add_b, (size_t)add_vals * 4, hipMemcpyHostToDevice, e->stream));
  }
  sync(e);  // `shifted` is a host temporary
  std::swap(off.b.p, n_off.b.p), std::swap(off.b.cap, n_off.b.cap);
  std::swap(va.b.p, n_a.b.p), std::swap(va.b.cap, n_a.b.cap);
  if (vb) std::swap(vb->b.p, n_b.b.p), std::swap(vb->b.cap, n_b.b.cap);
}


void cycle_update(cook_engine* e, UpdateBufs& ub, const cook_cycle_delta* d) {
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
  // ---- keep flags and their prefix sums --------------------------------------------------------------------------------
  int* keep = ub.keep.ensure(std::max(1u, N));
  int* keep_p = ub.keep_p.ensure(std::max(1u, P));
  SumI* incl = ub.incl.ensure(std::max(1u, N));
  SumI* incl_p = ub.incl_p.ensure(std::max(1u, P));
  unsigned* bad = ub.bad.ensure(1);
  COOK_HIP(hipMemsetAsync(bad, 0, 4, e->stream));
  unsigned n_keep = 0, p_keep = 0;
  if (N) {
    KL("upd_fill_ones", upd_fill_ones, div_up(N, 256), 256, keep, N);
    if (P) KL("upd_fill_ones", upd_fill_ones, div_up(P, 256), 256, keep_p, P);
    if (d->n_remove) {
      DArr<uint32_t>& rem = ub.rem;
      h2d(e, rem, d->remove_task, d->n_remove);
      KL("upd_mark_removed", upd_mark_removed, div_up(d->n_remove, 256), 256, (const uint32_t*)rem.ptr(), d->n_remove, N, keep, bad);
    }
    KL("upd_pending_keep", upd_pending_keep, div_up(N, 256), 256, (const uint8_t*)e->t_pending.ptr(), (const uint32_t*)e->pend_ord.ptr(),
       (const int*)keep, N, keep_p);
    seg_scan<SumI>(e, "upd_scan", LoadI{keep}, (const uint8_t*)nullptr, N, incl, e->tmpI);
    if (P) seg_scan<SumI>(e, "upd_scan", LoadI{keep_p}, (const uint8_t*)nullptr, P, incl_p, e->tmpI);
    COOK_HIP(hipMemcpyAsync(e->h_scratch, &incl[N - 1], 4, hipMemcpyDeviceToHost, e->stream));
    if (P) COOK_HIP(hipMemcpyAsync((char*)e->h_scratch + 4, &incl_p[P - 1], 4, hipMemcpyDeviceToHost, e->stream));
    COOK_HIP(hipMemcpyAsync((char*)e->h_scratch + 8, bad, 4, hipMemcpyDeviceToHost, e->stream));
    sync(e);
    int t[3] = {0, 0, 0};
    std::memcpy(t, e->h_scratch, 12);
    if (t[2]) e->fail(COOK_E_INVALID, "cook_cycle_update: remove_task holds an index out of range or twice");
    n_keep = (unsigned)t[0];
    p_keep = P ? (unsigned)t[1] : 0u;
  } else if (d->n_remove) {
    e->fail(COOK_E_INVALID, "cook_cycle_update: nothing staged to remove from");
  }
  const unsigned N2 = n_keep + n_add, P2 = p_keep + p_add;
  // ---- task columns (rank inputs) ----------------------------------------------------------------------------------------
  upd_column<double>(e, ub, e->t_cpus, keep, incl, N, n_keep, at ? at->cpus : nullptr, n_add, false, 0.0);
  upd_column<double>(e, ub, e->t_mem, keep, incl, N, n_keep, at ? at->mem : nullptr, n_add, false, 0.0);
  if (e->has_gpus) upd_column<double>(e, ub, e->t_gpus, keep, incl, N, n_keep, at ? at->gpus : nullptr, n_add, true, 0.0);
  upd_column<uint32_t>(e, ub, e->t_user, keep, incl, N, n_keep, at ? at->user : nullptr, n_add, false, 0u);
  upd_column<int32_t>(e, ub, e->t_prio, keep, incl, N, n_keep, at ? at->priority : nullptr, n_add, false, 0);
  upd_column<int64_t>(e, ub, e->t_start, keep, incl, N, n_keep, at ? at->start_ms : nullptr, n_add, false, 0);
  upd_column<int64_t>(e, ub, e->t_task, keep, incl, N, n_keep, at ? at->task_id : nullptr, n_add, false, 0);
  upd_column<int64_t>(e, ub, e->t_job, keep, incl, N, n_keep, at ? at->job_id : nullptr, n_add, false, 0);
  upd_column<uint8_t>(e, ub, e->t_pending, keep, incl, N, n_keep, at ? at->pending : nullptr, n_add, false, (uint8_t)0);
  // pending ordinals of the new array: exclusive count of pending rows in front
  e->pend_ord.ensure(std::max(1u, N2));
  if (N2) {
    int* flag = ub.flag.ensure(N2);
    SumI* fincl = ub.incl.ensure(N2);
    KL("upd_pending_flag", upd_pending_flag, div_up(N2, 256), 256, (const uint8_t*)e->t_pending.ptr(), N2, flag);
    seg_scan<SumI>(e, "upd_scan", LoadI{flag}, (const uint8_t*)nullptr, N2, fincl, e->tmpI);
    KL("upd_pend_ord", upd_pend_ord, div_up(N2, 256), 256, (const int*)flag, (const SumI*)fincl, N2, e->pend_ord.ptr());
  }
  // ---- pending-job columns (match inputs, by pending ordinal) ------------------------------------------------------------
  upd_column<double>(e, ub, e->j_cpus, keep_p, incl_p, P, p_keep, aj ? aj->cpus : nullptr, p_add, false, 0.0);
  upd_column<double>(e, ub, e->j_mem, keep_p, incl_p, P, p_keep, aj ? aj->mem : nullptr, p_add, false, 0.0);
  in.j_cpus = e->j_cpus.ptr();
  in.j_mem = e->j_mem.ptr();
  if (in.j_gpus) upd_column<double>(e, ub, e->j_gpus, keep_p, incl_p, P, p_keep, aj ? aj->gpus : nullptr, p_add, true, 0.0), in.j_gpus = e->j_gpus.ptr();
  if (in.j_gpu_model)
    upd_column<uint32_t>(e, ub, e->j_gpu_model, keep_p, incl_p, P, p_keep, aj ? aj->gpu_model : nullptr, p_add, true, 0u), in.j_gpu_model = e->j_gpu_model.ptr();
  if (e->has_j_user) upd_column<uint32_t>(e, ub, e->j_user, keep_p, incl_p, P, p_keep, aj ? aj->user : nullptr, p_add, true, 0u);
  if (in.j_group)
    upd_column<uint32_t>(e, ub, e->j_group, keep_p, incl_p, P, p_keep, aj ? aj->group : nullptr, p_add, true, 0xFFFFFFFFu), in.j_group = e->j_group.ptr();
  if (in.j_reserved_host)
    upd_column<int32_t>(e, ub, e->j_reserved_host, keep_p, incl_p, P, p_keep, aj ? aj->reserved_host : nullptr, p_add, true, -1),
        in.j_reserved_host = e->j_reserved_host.ptr();
  if (in.j_ckpt) upd_column<uint32_t>(e, ub, e->j_ckpt, keep_p, incl_p, P, p_keep, aj ? aj->ckpt_location : nullptr, p_add, true, 0u), in.j_ckpt = e->j_ckpt.ptr();
  if (in.j_est_end)
    upd_column<int64_t>(e, ub, e->j_est_end, keep_p, incl_p, P, p_keep, aj ? aj->est_end_ms : nullptr, p_add, true, (int64_t)0), in.j_est_end = e->j_est_end.ptr();
  if (in.j_disk_req) {
    upd_column<double>(e, ub, e->j_disk_req, keep_p, incl_p, P, p_keep, aj ? aj->disk_request : nullptr, p_add, true, -1.0);
    upd_column<uint32_t>(e, ub, e->j_disk_type, keep_p, incl_p, P, p_keep, aj ? aj->disk_type : nullptr, p_add, true, 0u);
    in.j_disk_req = e->j_disk_req.ptr();
    in.j_disk_type = e->j_disk_type.ptr();
  }
  // ports / named scalars (a new job asking for one switches the extra resource tests on)
  if (in.j_ports) upd_column<int32_t>(e, ub, e->j_ports, keep_p, incl_p, P, p_keep, aj ? aj->ports : nullptr, p_add, true, 0), in.j_ports = e->j_ports.ptr();
  for (unsigned sc = 0; sc < in.n_scal; ++sc) {
    const double* col = (aj && aj->scalars && sc < aj->n_scalars) ? aj->scalars + (size_t)sc * p_add : nullptr;
    upd_column<double>(e, ub, e->j_scal[sc], keep_p, incl_p, P, p_keep, col, p_add, true, __builtin_nan(""));
    in.j_scal[sc] = e->j_scal[sc].ptr();
    for (unsigned r = 0; col && r < p_add && !in.has_x; ++r) in.has_x = col[r] == col[r];
  }
  for (unsigned r = 0; aj && aj->ports && r < p_add && !in.has_x; ++r) in.has_x = aj->ports[r] > 0;
  if (in.j_eq_off) {
    upd_csr(e, ub, e->j_eq_off, e->j_eq_key, &e->j_eq_val, keep_p, incl_p, P, p_keep, aj ? aj->eq_off : nullptr, aj ? aj->eq_key : nullptr,
            aj ? aj->eq_val : nullptr, p_add);
    in.j_eq_off = e->j_eq_off.ptr(), in.j_eq_key = e->j_eq_key.ptr(), in.j_eq_val = e->j_eq_val.ptr();
  }
  if (in.j_novel_off) {
    upd_csr(e, ub, e->j_novel_off, e->j_novel_host, nullptr, keep_p, incl_p, P, p_keep, aj ? aj->novel_off : nullptr, aj ? aj->novel_host : nullptr,
            nullptr, p_add);
    in.j_novel_off = e->j_novel_off.ptr(), in.j_novel_host = e->j_novel_host.ptr();
  }
  // the eligible mask of cook_cycle_set_considerable is indexed by pending ordinal like the job columns: it moves with them; the
  // jobs the delta adds are eligible until the host says otherwise (a fresh mask through cook_cycle_set_considerable)
  if (e->cb && e->cb->has_elig_by_pending)
    upd_column<uint8_t>(e, ub, e->cb->elig_by_pending, keep_p, incl_p, P, p_keep, nullptr, p_add, true, (uint8_t)1);
  sync(e);
  e->N = N2;
  e->n_pending = P2;
  e->Kjobs = P2;
  e->K = P2;
  in.K = P2;
  e->rank_done = false;
  e->match_done = false;
  e->has_deferred = false;
  if (d->offers) match_stage_offers(e, d->offers);
}

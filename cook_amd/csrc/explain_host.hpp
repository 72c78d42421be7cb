// explain_host.hpp — host orchestration of cook_match_explain / cook_match_metrics (included by engine.hip inside its anonymous
// namespace).  Both read the result of the LAST match of the engine (cook_match_run, cook_cycle_run or the lockstep pair) in
// place on the device: job_to_offer, the group placement chains and the staged job / offer columns.
#pragma once

struct ExplainBufs {
  DArr<uint64_t> key, kc, km;
  DArr<uint32_t> permA, permB, ostart, oend, pos, counts, user_cons, user_match;
  DArr<double> jc, jm, out;  // out: 16 doubles of ResourceStats x 2
  DArr<uint32_t> largest;
  DArr<SumU4> scan;
  DArr<unsigned long long> jgpus, ogpus;
};

MatchState explain_state(cook_engine* e, unsigned K) {
  MatchState st{};
  st.ac = e->m_ac.ptr();
  st.am = e->m_am.ptr();
  st.acount = e->m_acount.ptr();
  st.group_last = e->m_group_last.ptr();
  st.job_prev = e->m_job_prev.ptr();
  st.job_to_offer = e->m_j2o.ptr();
  st.fail_code = e->m_fail.ptr();
  st.summary = e->m_summary.ptr();
  st.cutoff = 0x7FFFFFFF;
  st.alive = e->m_alive.ptr();
  st.jmin = (const double*)e->m_jmin.ptr();
  (void)K;
  return st;
}

void match_explain(cook_engine* e, ExplainBufs& x, const uint32_t* job_pos, unsigned n, uint32_t* counts) {
  if (!e->match_done || !e->last_in_valid) e->fail(COOK_E_STATE, "cook_match_explain before a match ran");
  if (n && (!job_pos || !counts)) e->fail(COOK_E_INVALID, "cook_match_explain: null job positions / counts");
  const MatchIn in = e->last_in;
  const unsigned K = in.K, M = in.M;
  for (unsigned q = 0; q < n; ++q)
    if (job_pos[q] >= K) e->fail(COOK_E_INVALID, "cook_match_explain: job position beyond the jobs of the last match");
  if (n == 0) return;
  std::memset(counts, 0, (size_t)n * WHY_SLOTS * 4);
  if (M == 0) return;  // no host refused the job: the summary is empty
  const MatchState st = explain_state(e, K);
  // job positions stably partitioned by the offer they were placed on: an offer's segment is in rank order
  x.key.ensure(K);
  x.permA.ensure(K);
  x.permB.ensure(K);
  x.ostart.ensure(M);
  x.oend.ensure(M);
  memset_async(e, x.ostart.ptr(), 0, (size_t)M * 4);
  memset_async(e, x.oend.ptr(), 0, (size_t)M * 4);
  const unsigned gK = div_up(K, 256);
  KL("explain_offer_keys", explain_offer_keys, gK, 256, (const int32_t*)st.job_to_offer, K, M, x.key.ptr());
  KM<iota_u32, 256>(e, "iota", gK, x.permA.ptr(), K);
  unsigned long long mask = 0;
  for (unsigned long long t = M; t; t >>= 1) mask = (mask << 1) | 1ull;
  const uint32_t* plist = radix_sort_masked(e, x.key.ptr(), mask, x.permA.ptr(), x.permA.ptr(), x.permB.ptr(), K);
  KL("explain_seg_bounds", offers_seg_bounds, gK, 256, plist, (const uint64_t*)x.key.ptr(), K, M, x.ostart.ptr(), x.oend.ptr());
  h2d(e, x.pos, job_pos, n);
  x.counts.ensure((size_t)n * WHY_SLOTS);
  memset_async(e, x.counts.ptr(), 0, (size_t)n * WHY_SLOTS * 4);
  for (unsigned q0 = 0; q0 < n; q0 += 65535u) {  // gridDim.y limit
    const unsigned nq = std::min(65535u, n - q0);
    KL("explain_classify", explain_classify, dim3(div_up(M, 256), nq), 256, in, st, (const uint32_t*)x.pos.ptr() + q0, plist,
       (const uint32_t*)x.ostart.ptr(), (const uint32_t*)x.oend.ptr(), x.counts.ptr() + (size_t)q0 * WHY_SLOTS);
  }
  copy_async(e, counts, x.counts.ptr(), (size_t)n * WHY_SLOTS * 4, hipMemcpyDeviceToHost);
  sync(e);
}

// resource-maps->stats of two columns (cpus, mem) of n rows resident on the device -> 8 doubles + 2 indices at out / largest
void resource_stats(cook_engine* e, ExplainBufs& x, const double* a, const double* b, const uint64_t* ka, const uint64_t* kb, unsigned n,
                    double* out /* device: total a, total b, p50 a, p95 a, p100 a, p50 b, p95 b, p100 b */, uint32_t* largest) {
  x.scan.ensure(n);
  seg_scan<SumU4>(e, "metrics_total_scan", LoadPair{a, b}, (const uint8_t*)nullptr, n, x.scan.ptr(), e->tmpU4);
  KL("metrics_totals", metrics_totals, 1, 1024, (const SumU4*)x.scan.ptr(), a, b, n, out + 0, out + 1);
  x.permA.ensure(n);
  x.permB.ensure(n);
  const unsigned g = div_up(n, 256);
  for (int col = 0; col < 2; ++col) {
    const uint64_t* key = col ? kb : ka;
    unsigned long long* dmask = e->d_scratch64.ensure(8);
    memset_async(e, dmask, 0, 8);
    KM<radix_varying_bits, 256>(e, "radix_varying_bits", std::min(g, 64u), key, n, dmask, std::min(g, 64u));
    readback64(e, 1);
    const unsigned long long mask = e->h_scratch[0];
    KM<iota_u32, 256>(e, "iota", g, x.permA.ptr(), n);
    const uint32_t* perm = radix_sort_masked(e, key, mask, x.permA.ptr(), x.permA.ptr(), x.permB.ptr(), n);
    KL("metrics_pick", metrics_pick, 1, 64, perm, col ? b : a, n, out + 2 + 3 * col, out + 3 + 3 * col, out + 4 + 3 * col, largest + col);
  }
}

void match_metrics(cook_engine* e, ExplainBufs& x, cook_cycle_metrics* out, uint32_t* user_considerable, uint32_t* user_matched,
                   unsigned n_users, int64_t* job_gpus_by_model, int64_t* offer_gpus_by_model, unsigned n_models) {
  if (!e->match_done || !e->last_in_valid) e->fail(COOK_E_STATE, "cook_match_metrics before a match ran");
  if (!out) e->fail(COOK_E_INVALID, "cook_match_metrics: null output");
  const MatchIn in = e->last_in;
  const unsigned K = in.K, M = in.M;
  const bool want_users = user_considerable || user_matched;
  if (want_users && !e->has_j_user) e->fail(COOK_E_INVALID, "cook_match_metrics: per-user counts need the jobs' user column staged");
  const MatchState st = explain_state(e, K);
  std::memset(out, 0, sizeof(*out));
  const double nan = std::numeric_limits<double>::quiet_NaN();
  cook_resource_stats empty{0.0, 0.0, nan, nan, nan, nan, nan, nan, COOK_NONE_U32, COOK_NONE_U32};
  out->jobs = out->offer_stats = empty;
  x.out.ensure(16);
  x.largest.ensure(4);
  x.user_cons.ensure(std::max(1u, n_users));
  x.user_match.ensure(std::max(1u, n_users));
  x.jgpus.ensure(n_models + 1u);
  x.ogpus.ensure(n_models + 1u);
  memset_async(e, x.user_cons.ptr(), 0, (size_t)std::max(1u, n_users) * 4);
  memset_async(e, x.user_match.ptr(), 0, (size_t)std::max(1u, n_users) * 4);
  memset_async(e, x.jgpus.ptr(), 0, (size_t)(n_models + 1u) * 8);
  memset_async(e, x.ogpus.ptr(), 0, (size_t)(n_models + 1u) * 8);
  unsigned* d_sched = e->d_counters.ptr() + 14;
  memset_async(e, d_sched, 0, 4);
  if (K) {
    x.jc.ensure(K);
    x.jm.ensure(K);
    x.kc.ensure(std::max(K, M));
    x.km.ensure(std::max(K, M));
    KL("metrics_gather_jobs", metrics_gather_jobs, div_up(K, 256), 256, in, x.jc.ptr(), x.jm.ptr(), x.kc.ptr(), x.km.ptr());
    resource_stats(e, x, x.jc.ptr(), x.jm.ptr(), x.kc.ptr(), x.km.ptr(), K, x.out.ptr(), x.largest.ptr());
    KL("metrics_job_counts", metrics_job_counts, div_up(K, 256), 256, in, (const int32_t*)st.job_to_offer,
       want_users ? (const uint32_t*)e->j_user.ptr() : (const uint32_t*)nullptr, n_users, x.user_cons.ptr(), x.user_match.ptr(), n_models,
       job_gpus_by_model ? x.jgpus.ptr() : (unsigned long long*)nullptr);
  }
  if (M) {
    x.kc.ensure(std::max(K, M));
    x.km.ensure(std::max(K, M));
    KL("metrics_keys", metrics_keys, div_up(M, 256), 256, in.o_cpus, in.o_mem, M, x.kc.ptr(), x.km.ptr());
    resource_stats(e, x, in.o_cpus, in.o_mem, x.kc.ptr(), x.km.ptr(), M, x.out.ptr() + 8, x.largest.ptr() + 2);
    KL("metrics_offer_counts", metrics_offer_counts, div_up(M, 256), 256, (const int32_t*)st.acount, M, d_sched, in.o_gpu_model,
       in.o_gpu_count, in.gpu_slots ? in.gpu_slots : 1u, n_models, offer_gpus_by_model ? x.ogpus.ptr() : (unsigned long long*)nullptr);
  }
  double h[16];
  uint32_t hl[4];
  unsigned hs[2] = {0, 0};
  copy_async(e, h, x.out.ptr(), sizeof(h), hipMemcpyDeviceToHost);
  copy_async(e, hl, x.largest.ptr(), sizeof(hl), hipMemcpyDeviceToHost);
  copy_async(e, &hs[0], d_sched, 4, hipMemcpyDeviceToHost);
  copy_async(e, &hs[1], st.summary, 4, hipMemcpyDeviceToHost);
  int head_offer = -1;  // matched-considerable-jobs-head? = the first considerable job is among the matched (scheduler.clj:1381)
  if (K) copy_async(e, &head_offer, st.job_to_offer, 4, hipMemcpyDeviceToHost);
  if (user_considerable && n_users)
    copy_async(e, user_considerable, x.user_cons.ptr(), (size_t)n_users * 4, hipMemcpyDeviceToHost);
  if (user_matched && n_users) copy_async(e, user_matched, x.user_match.ptr(), (size_t)n_users * 4, hipMemcpyDeviceToHost);
  if (job_gpus_by_model) copy_async(e, job_gpus_by_model, x.jgpus.ptr(), (size_t)(n_models + 1u) * 8, hipMemcpyDeviceToHost);
  if (offer_gpus_by_model)
    copy_async(e, offer_gpus_by_model, x.ogpus.ptr(), (size_t)(n_models + 1u) * 8, hipMemcpyDeviceToHost);
  sync(e);
  auto fill = [&](cook_resource_stats& r, const double* d, const uint32_t* l) {
    r.total_cpus = d[0];
    r.total_mem = d[1];
    r.p50_cpus = d[2];
    r.p95_cpus = d[3];
    r.p100_cpus = d[4];
    r.p50_mem = d[5];
    r.p95_mem = d[6];
    r.p100_mem = d[7];
    r.largest_by_cpus = l[0];
    r.largest_by_mem = l[1];
  };
  if (K) fill(out->jobs, h, hl);
  if (M) fill(out->offer_stats, h + 8, hl + 2);
  out->considerable = K;
  out->matched = K ? hs[1] : 0u;
  out->unmatched = K - out->matched;
  out->offers = M;
  out->offers_scheduled = hs[0];
  out->head_matched = head_offer >= 0 ? 1u : 0u;
}

// considerable_host.hpp — host orchestration of cook_considerable / the considerable stage of cook_cycle_run
// (included by engine.hip inside its anonymous namespace).
#pragma once

struct ConsBufs {
  // user state (staged by cook_considerable / cook_cycle_set_considerable)
  bool users_staged = false;
  unsigned U = 0;
  DArr<double> qcount, qcpus, qmem, qgpus, ucount, ucpus, umem, ugpus;
  DArr<int64_t> tokens;
  bool has_tokens = false;
  int enforce = 0;
  bool has_pool_quota = false, pool_usage_given = false;
  cook_usage pool_quota{}, pool_usage{};
  // cycle mode
  bool cycle_on = false;
  DArr<uint8_t> elig_by_pending;
  bool has_elig_by_pending = false;
  // queue (device)
  DArr<double> q_cpus, q_mem, q_gpus;
  DArr<uint32_t> q_user;
  DArr<uint8_t> q_elig;
  // work
  DArr<uint64_t> ukey;
  DArr<uint32_t> permA, permB, g_user, seg_start, seg_end, inexact, rate_limited, passed, qitemA, qitemB;
  DArr<uint8_t> head;
  DArr<SumU4> g_use, pre, quseA, quseB, pusage;
  DArr<int> flag1, keep_q;
  DArr<SumI> scan;
  // result
  uint32_t* result = nullptr;  // device: queue positions of the considerable jobs
  unsigned n_result = 0;
};

void cons_stage_users(cook_engine* e, ConsBufs& c, const cook_user_state* us) {
  if (!us) e->fail(COOK_E_INVALID, "cook_considerable: null user state");
  const unsigned U = us->n;
  if (U && (!us->quota_count || !us->quota_cpus || !us->quota_mem || !us->quota_gpus || !us->usage_count || !us->usage_cpus ||
            !us->usage_mem || !us->usage_gpus))
    e->fail(COOK_E_INVALID, "cook_considerable: user quota / usage arrays are required");
  c.U = U;
  h2d(e, c.qcount, us->quota_count, U);
  h2d(e, c.qcpus, us->quota_cpus, U);
  h2d(e, c.qmem, us->quota_mem, U);
  h2d(e, c.qgpus, us->quota_gpus, U);
  h2d(e, c.ucount, us->usage_count, U);
  h2d(e, c.ucpus, us->usage_cpus, U);
  h2d(e, c.umem, us->usage_mem, U);
  h2d(e, c.ugpus, us->usage_gpus, U);
  c.has_tokens = us->tokens_left != nullptr;
  if (c.has_tokens) h2d(e, c.tokens, us->tokens_left, U);
  c.enforce = us->enforce_rate_limit;
  c.has_pool_quota = us->has_pool_quota != 0;
  c.pool_quota = us->pool_quota;
  c.pool_usage_given = us->pool_usage_given != 0;
  c.pool_usage = us->pool_usage;
  sync(e);
  c.users_staged = true;
}

// the filters over a device-resident queue of n jobs; leaves the first min(K, survivors) queue positions in c.result
void cons_run_device(cook_engine* e, ConsBufs& c, unsigned n, const double* q_cpus, const double* q_mem, const double* q_gpus,
                     const uint32_t* q_user, const uint8_t* q_elig, unsigned K) {
  const unsigned U = c.U;
  c.rate_limited.ensure(std::max(1u, U));
  c.passed.ensure(std::max(1u, U));
  memset_async(e, c.rate_limited.ptr(), 0, (size_t)std::max(1u, U) * 4);
  memset_async(e, c.passed.ptr(), 0, (size_t)std::max(1u, U) * 4);
  c.n_result = 0;
  c.result = c.qitemA.ensure(std::max(1u, n));
  if (n == 0) return;  // K == 0 still runs the filters: the per-user rate-limit counters cover the whole queue
  const unsigned gN = div_up(n, 256);
  // ---- stable partition of the queue positions by user -----------------------------------------------------------------
  c.ukey.ensure(n);
  c.permA.ensure(n);
  c.permB.ensure(n);
  KM<cons_user_keys, 256>(e, "cons_user_keys", gN, q_user, n, c.ukey.ptr());
  KM<iota_u32, 256>(e, "iota", gN, c.permA.ptr(), n);
  unsigned long long umask = 0;
  for (unsigned long long x = U ? U - 1 : 0; x; x >>= 1) umask = (umask << 1) | 1ull;
  const uint32_t* permU = radix_sort_masked(e, c.ukey.ptr(), umask, c.permA.ptr(), c.permA.ptr(), c.permB.ptr(), n);
  c.g_user.ensure(n);
  c.g_use.ensure(n);
  c.head.ensure(n);
  c.seg_start.ensure(std::max(1u, U));
  c.seg_end.ensure(std::max(1u, U));
  c.inexact.ensure(std::max(1u, U));
  c.pre.ensure(n);
  memset_async(e, c.inexact.ptr(), 0, (size_t)std::max(1u, U) * 4);
  KM<cons_gather, 256>(e, "cons_gather", gN, permU, n, q_user, q_cpus, q_mem, q_gpus, c.g_user.ptr(), c.g_use.ptr(), c.head.ptr(), c.seg_start.ptr(),
      c.seg_end.ptr());
  // ---- (i) per-user quota filter, seeded with the users' running usage (tools.clj:903-915) -------------------------------
  LoadUserSeeded ld{c.g_use.ptr(), c.head.ptr(), c.g_user.ptr(), c.ucount.ptr(), c.ucpus.ptr(), c.umem.ptr(), c.ugpus.ptr()};
  seg_scan<SumU4>(e, "cons_user_usage_scan", ld, (const uint8_t*)c.head.ptr(), n, c.pre.ptr(), e->tmpU4);
  KM<rank_mark_inexact, 256>(e, "rank_mark_inexact", gN, (const SumU4*)c.pre.ptr(), (const uint32_t*)c.g_user.ptr(), n, c.inexact.ptr());
  KM<cons_fix_inexact, 256>(e, "cons_fix_inexact", div_up(std::max(1u, U), 256), (const SumU4*)c.g_use.ptr(), c.pre.ptr(),
      (const uint32_t*)c.seg_start.ptr(), (const uint32_t*)c.seg_end.ptr(), (const uint32_t*)c.inexact.ptr(), U, (const double*)c.ucount.ptr(),
      (const double*)c.ucpus.ptr(), (const double*)c.umem.ptr(), (const double*)c.ugpus.ptr());
  c.flag1.ensure(n);
  c.keep_q.ensure(n);
  c.scan.ensure(n);
  KM<cons_user_quota_flag, 256>(e, "cons_user_quota_flag", gN, (const SumU4*)c.pre.ptr(), (const uint32_t*)c.g_user.ptr(), n,
      (const double*)c.qcount.ptr(), (const double*)c.qcpus.ptr(), (const double*)c.qmem.ptr(), (const double*)c.qgpus.ptr(), c.flag1.ptr());
  // ---- (ii) launch-rate limit: index of the job among its user's survivors (tools.clj:935-955) -----------------------------
  seg_scan<SumI>(e, "cons_user_index_scan", LoadI{c.flag1.ptr()}, (const uint8_t*)c.head.ptr(), n, c.scan.ptr(), e->tmpI);
  KM<cons_rate_limit, 256>(e, "cons_rate_limit", gN, (const int*)c.flag1.ptr(), (const SumI*)c.scan.ptr(), (const uint32_t*)c.g_user.ptr(), permU, n,
      c.has_tokens ? (const int64_t*)c.tokens.ptr() : (const int64_t*)nullptr, c.enforce, c.keep_q.ptr(), c.rate_limited.ptr(), c.passed.ptr());
  // ---- survivors back in queue order ---------------------------------------------------------------------------------------------
  uint32_t* qitem = c.qitemA.ensure(n);
  uint32_t* qitem_o = c.qitemB.ensure(n);
  SumU4* quse = c.quseA.ensure(n);
  SumU4* quse_o = c.quseB.ensure(n);
  seg_scan<SumI>(e, "cons_compact_scan", LoadI{c.keep_q.ptr()}, (const uint8_t*)nullptr, n, c.scan.ptr(), e->tmpI);
  unsigned* dlen = e->d_counters.ptr() + 12;
  memset_async(e, dlen, 0, 4);
  KM<cons_compact_queue, 256>(e, "cons_compact_queue", gN, (const int*)c.keep_q.ptr(), (const SumI*)c.scan.ptr(), n, q_cpus, q_mem, q_gpus, qitem, quse, dlen);
  // ---- (iii) pool quota, seeded with the pool usage (tools.clj:917-933, 966) -------------------------------------------------------
  cook_usage base = c.pool_usage;
  if (c.has_pool_quota && !c.pool_usage_given) {
    c.pusage.ensure(1);
    if (U) {
      KM<cons_pool_usage, 1024>(e, "cons_pool_usage", 1, (const double*)c.ucount.ptr(), (const double*)c.ucpus.ptr(), (const double*)c.umem.ptr(),
          (const double*)c.ugpus.ptr(), U, c.pusage.ptr());
      pinned_copy(e, e->h_scratch + 8, c.pusage.ptr(), sizeof(SumU4), hipMemcpyDeviceToHost);
    }
  }
  pinned_copy(e, e->h_scratch, dlen, 4, hipMemcpyDeviceToHost);
  sync(e);
  unsigned len = 0;
  std::memcpy(&len, e->h_scratch, 4);
  if (c.has_pool_quota && !c.pool_usage_given) {
    SumU4 h = SumU4::zero();
    if (U) std::memcpy(&h, e->h_scratch + 8, sizeof(SumU4));
    base = cook_usage{h.count, h.cpus, h.mem, h.gpus};
  }
  if (len && c.has_pool_quota) len = queue_filter_quota(e, 2, len, c.pool_quota, base, qitem, quse, qitem_o, quse_o);
  // ---- job-allowed-to-start? + launch plugin (host-evaluated mask), then take K (scheduler.clj:747-749) --------------------------------
  if (len && q_elig) {
    e->iflag.ensure(len);
    e->scanI.ensure(len);
    KM<cons_eligible_flag, 256>(e, "cons_eligible_flag", div_up(len, 256), (const uint32_t*)qitem, len, q_elig, e->iflag.ptr());
    seg_scan<SumI>(e, "queue_compact_scan", LoadI{e->iflag.ptr()}, (const uint8_t*)nullptr, len, e->scanI.ptr(), e->tmpI);
    unsigned* len_out = e->d_counters.ptr() + 9;
    KM<queue_compact, 256>(e, "queue_compact", div_up(len, 256), (const uint32_t*)qitem, (const SumU4*)quse, (const int*)e->iflag.ptr(),
        (const SumI*)e->scanI.ptr(), len, qitem_o, quse_o, len_out);
    pinned_copy(e, e->h_scratch, len_out, 4, hipMemcpyDeviceToHost);
    sync(e);
    std::memcpy(&len, e->h_scratch, 4);
    std::swap(qitem, qitem_o);
    std::swap(quse, quse_o);
  }
  c.result = qitem;
  c.n_result = std::min(len, K);
}

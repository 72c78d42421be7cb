// rebalance_host.hpp — host orchestration of cook_rebalance (included by engine.hip inside its anonymous namespace).
// Stage: concatenate running tasks ++ the pending jobs' task slots, dense per-host tables, H2D.  Run: per-user order and
// host grouping by the rank path's radix sort, masked DRU scan, then the decision loop (three kernels + the masked
// re-scan per pending job) enqueued without host round trips; the budget is read back every RB_CHECK jobs to stop early.
#pragma once

constexpr unsigned RB_CHECK = 32;
constexpr unsigned RB_FIRST_MIN = COOK_SHAPE(256u, 2u);  // hosts of the first phase of rebal_decide at least (rebalance_run)

struct RebalBufs {
  bool staged = false, done = false;
  unsigned R = 0, P = 0, S = 0, U = 0, H = 0, G = 0, n_attr = 0, co_cap = 0;
  bool has_attrs = false;
  cook_rebalance_params rp{};
  // slot inputs (A space)
  DArr<double> cpus, mem, gpus;
  DArr<uint32_t> user, host;
  DArr<int32_t> prio;
  DArr<int64_t> start, task, job;
  DArr<uint8_t> pending, cached;
  bool has_cached = false;
  // users
  DArr<double> divc, divm, divg, qcount, qcpus, qmem, qgpus;
  // B space
  DArr<uint32_t> permA, permB2, posB, s_user, seg_start, seg_end, inexact, user_safe;
  DArr<uint8_t> s_pending, head, act;
  DArr<SumU4> s_use, pre;
  DArr<double> dru;
  // hosts
  DArr<uint64_t> hkey;
  DArr<uint32_t> hpermA, hpermB, hstart, hend, hbase, h_pb, h_user, hidx, chg, chg_tile, chg_bad, chg_mark;
  DArr<SumU4> tile_agg, tile_carry;
  DArr<uint32_t> dl_pos;
  DArr<int32_t> dl_sign;
  bool spare_safe = false;
  std::vector<uint32_t> user_safe_h;  // read back once per run: all ones = the re-scoring takes the one-launch path (rebal_rs_delta)
  DArr<uint32_t> x_head, x_cnt, x_next, big_list;
  DArr<double> h_cpus, h_mem, h_gpus, h_dru;
  DArr<uint8_t> h_act;
  uint32_t* hperm = nullptr;
  DArr<int32_t> row_of_host;
  DArr<double> spare_c, spare_m, spare_g, spare0_c, spare0_m, spare0_g;  // spare0_*: as staged (a run updates spare_*)
  DArr<uint8_t> has_spare, has_spare0;
  // attribute table
  DArr<uint32_t> a_host, a_gpu_model, a_disk_type, a_attr, a_location;
  DArr<uint8_t> a_k8s;
  DArr<double> a_gpu_count, a_disk_space;
  DArr<int64_t> a_host_start;
  bool ha_k8s = false, ha_gpu = false, ha_disk = false, ha_attr = false, ha_loc = false, ha_start = false;
  // pending jobs
  DArr<double> j_cpus, j_mem, j_gpus, j_disk_req;
  DArr<uint32_t> j_gpu_model, j_user, j_group, j_eq_off, j_eq_key, j_eq_val, j_novel_off, j_novel_host, j_ckpt, j_disk_type;
  DArr<int64_t> j_est_end;
  unsigned a_gpu_slots = 1, a_disk_slots = 1;
  unsigned max_seg = 0;  // running tasks on the fullest host
  bool hj_gpus = false, hj_gpu_model = false, hj_group = false, hj_eq = false, hj_novel = false, hj_ckpt = false, hj_disk = false,
       hj_est = false;
  // groups
  DArr<uint8_t> g_type;
  DArr<uint32_t> g_attr_key, g_run_off, g_run_host;
  DArr<int32_t> g_min;
  bool hg_run = false;
  // dynamic
  DArr<uint32_t> x_pj, x_host, pre_hosts, co_val, hres_len, hres_base, srt_slot, gs_posB, gs_slot, gs_ord;
  DArr<uint8_t> x_known;
  DArr<unsigned long long> hres_key, blk_key, hmax_key, best_key;
  DArr<uint32_t> blk_host, h_host;
  RebalIn in_host{};
  DArr<RebalIn> in_dev;
  unsigned long long thr_key = 0ull;  // this run's first-phase threshold (0: one phase)
  std::vector<unsigned long long> hmax_h;
  DArr<double> hres_dru, hres_c, hres_m, hres_g, gs_dru, gs_cpus, gs_mem, gs_gpus, pending_dru;
  DArr<cook_preemption> decisions;
  DArr<uint32_t> preempted;
  DArr<RebalCtl> ctl;
  DArr<RebalJob> jobctx;
  RebalCtl last{};
  double ms = 0;
};

void rebalance_stage(cook_engine* e, RebalBufs& b, const cook_tasks* run, const uint8_t* cached, const cook_jobs* pend,
                     const int64_t* pend_job_id, const int32_t* pend_prio, const cook_users* u, const cook_host_spare* spare,
                     const cook_offers* attrs, const cook_groups* groups, const cook_rebalance_params* rp) {
  if (!run || !pend || !u || !rp) e->fail(COOK_E_INVALID, "cook_rebalance: null running/pending/users/params");
  const unsigned R = run->n, P = pend->n, U = u->n, S = R + P;
  if (R && (!run->cpus || !run->mem || !run->user || !run->priority || !run->start_ms || !run->task_id || !run->job_id || !run->host))
    e->fail(COOK_E_INVALID, "cook_rebalance: running tasks need cpus, mem, user, priority, start_ms, task_id, job_id, host");
  if (P && (!pend->cpus || !pend->mem || !pend->user || !pend_job_id || !pend_prio))
    e->fail(COOK_E_INVALID, "cook_rebalance: pending jobs need cpus, mem, user, job ids and priorities");
  if (S && U == 0) e->fail(COOK_E_INVALID, "cook_rebalance: no users");
  const unsigned G = groups ? groups->n : 0;
  b.staged = b.done = false;
  b.R = R, b.P = P, b.S = S, b.U = U, b.G = G;
  b.rp = *rp;
  // ---- slots: running ++ one slot per pending job (the task it becomes when placed) ---------------------------------
  std::vector<double> c(S ? S : 1), m(S ? S : 1), g(S ? S : 1, 0.0);
  std::vector<uint32_t> us(S ? S : 1);
  std::vector<int32_t> pr(S ? S : 1);
  std::vector<int64_t> st(S ? S : 1, 0), tk(S ? S : 1, 0), jb(S ? S : 1);
  std::vector<uint8_t> pe(S ? S : 1, 0);
  unsigned maxh = 0;
  bool any_host = false;
  auto see_host = [&](uint32_t h) {
    maxh = std::max(maxh, h);
    any_host = true;
  };
  for (unsigned i = 0; i < R; ++i) {
    if (run->user[i] >= U) e->fail(COOK_E_INVALID, "cook_rebalance: user id out of range");
    c[i] = run->cpus[i], m[i] = run->mem[i], g[i] = run->gpus ? run->gpus[i] : 0.0;
    us[i] = run->user[i], pr[i] = run->priority[i], st[i] = run->start_ms[i], tk[i] = run->task_id[i], jb[i] = run->job_id[i];
    see_host(run->host[i]);
  }
  for (unsigned p = 0; p < P; ++p) {
    if (pend->user[p] >= U) e->fail(COOK_E_INVALID, "cook_rebalance: user id out of range");
    const unsigned s = R + p;
    c[s] = pend->cpus[p], m[s] = pend->mem[p], g[s] = pend->gpus ? pend->gpus[p] : 0.0;
    us[s] = pend->user[p], pr[s] = pend_prio[p], jb[s] = pend_job_id[p];
    pe[s] = 1;
  }
  if (spare)
    for (unsigned i = 0; i < spare->n; ++i) see_host(spare->host[i]);
  if (attrs)
    for (unsigned i = 0; i < attrs->n; ++i) see_host(attrs->host[i]);
  if (groups && groups->run_off)
    for (unsigned i = 0; i < groups->run_off[G]; ++i) see_host(groups->run_host[i]);
  const unsigned H = any_host ? maxh + 1 : 0;
  b.H = H;
  {  // running tasks on the fullest host: decides whether rebal_decide_big is ever needed
    std::vector<uint32_t> cnt(H ? H : 1, 0u);
    for (unsigned i = 0; i < R; ++i) cnt[run->host[i]] += 1u;
    b.max_seg = *std::max_element(cnt.begin(), cnt.end());
  }
  h2d(e, b.cpus, c.data(), S);
  h2d(e, b.mem, m.data(), S);
  h2d(e, b.gpus, g.data(), S);
  h2d(e, b.user, us.data(), S);
  h2d(e, b.prio, pr.data(), S);
  h2d(e, b.start, st.data(), S);
  h2d(e, b.task, tk.data(), S);
  h2d(e, b.job, jb.data(), S);
  h2d(e, b.pending, pe.data(), S);
  h2d(e, b.host, run->host, R);
  b.has_cached = cached != nullptr;
  if (cached) h2d(e, b.cached, cached, R);
  h2d(e, b.divc, u->div_cpus, U);
  h2d(e, b.divm, u->div_mem, U);
  h2d(e, b.divg, u->div_gpus, U);
  h2d(e, b.qcount, u->quota_count, U);
  h2d(e, b.qcpus, u->quota_cpus, U);
  h2d(e, b.qmem, u->quota_mem, U);
  h2d(e, b.qgpus, u->quota_gpus, U);
  // ---- dense per-host tables -------------------------------------------------------------------------------------------------
  std::vector<int32_t> row(H ? H : 1, -1);
  std::vector<double> sc(H ? H : 1, 0.0), sm(H ? H : 1, 0.0), sg(H ? H : 1, 0.0);
  std::vector<uint8_t> hs(H ? H : 1, 0);
  if (attrs)
    for (unsigned i = 0; i < attrs->n; ++i) row[attrs->host[i]] = (int32_t)i;
  if (spare)
    for (unsigned i = 0; i < spare->n; ++i) {
      const unsigned h = spare->host[i];
      sc[h] = spare->cpus[i], sm[h] = spare->mem[i], sg[h] = spare->gpus ? spare->gpus[i] : 0.0;
      hs[h] = 1;
    }
  {  // spare resources that cannot make a sum round (rebal_decide<SAFE>, with every user safe)
    auto okv = [](double v) {
      const double s = v * 1024.0;
      return v >= 0.0 && v < 4194304.0 && s == (double)(long long)s;
    };
    b.spare_safe = true;
    for (unsigned h = 0; h < H && b.spare_safe; ++h) b.spare_safe = okv(sc[h]) && okv(sm[h]) && okv(sg[h]);
  }
  h2d(e, b.row_of_host, row.data(), H);
  h2d(e, b.spare0_c, sc.data(), H);
  h2d(e, b.spare0_m, sm.data(), H);
  h2d(e, b.spare0_g, sg.data(), H);
  h2d(e, b.has_spare0, hs.data(), H);
  b.spare_c.ensure(H), b.spare_m.ensure(H), b.spare_g.ensure(H), b.has_spare.ensure(H);
  // ---- attribute table ----------------------------------------------------------------------------------------------------------
  b.has_attrs = attrs != nullptr && attrs->n > 0;
  b.ha_k8s = b.ha_gpu = b.ha_disk = b.ha_attr = b.ha_loc = b.ha_start = false;
  b.n_attr = 0;
  if (b.has_attrs) {
    const unsigned n = attrs->n;
    h2d(e, b.a_host, attrs->host, n);
    if ((b.ha_k8s = attrs->k8s != nullptr)) h2d(e, b.a_k8s, attrs->k8s, n);
    if ((b.ha_gpu = attrs->gpu_model != nullptr)) {
      if (!attrs->gpu_count) e->fail(COOK_E_INVALID, "cook_rebalance: host_attrs gpu_model without gpu_count");
      b.a_gpu_slots = res_slots(e, attrs->gpu_slots, "cook_rebalance: host_attrs gpu_slots");
      h2d(e, b.a_gpu_model, attrs->gpu_model, (size_t)n * b.a_gpu_slots);
      h2d(e, b.a_gpu_count, attrs->gpu_count, (size_t)n * b.a_gpu_slots);
    }
    if ((b.ha_disk = attrs->disk_type != nullptr && attrs->disk_space != nullptr)) {
      b.a_disk_slots = res_slots(e, attrs->disk_slots, "cook_rebalance: host_attrs disk_slots");
      h2d(e, b.a_disk_type, attrs->disk_type, (size_t)n * b.a_disk_slots);
      h2d(e, b.a_disk_space, attrs->disk_space, (size_t)n * b.a_disk_slots);
    }
    if ((b.ha_attr = attrs->attr != nullptr && attrs->n_attr_keys > 0)) {
      b.n_attr = attrs->n_attr_keys;
      h2d(e, b.a_attr, attrs->attr, (size_t)n * b.n_attr);
    }
    if ((b.ha_loc = attrs->location != nullptr)) h2d(e, b.a_location, attrs->location, n);
    if ((b.ha_start = attrs->host_start_s != nullptr)) h2d(e, b.a_host_start, attrs->host_start_s, n);
  }
  // ---- pending jobs ------------------------------------------------------------------------------------------------------------
  h2d(e, b.j_cpus, pend->cpus, P);
  h2d(e, b.j_mem, pend->mem, P);
  h2d(e, b.j_user, pend->user, P);
  if ((b.hj_gpus = pend->gpus != nullptr)) h2d(e, b.j_gpus, pend->gpus, P);
  if ((b.hj_gpu_model = pend->gpu_model != nullptr)) h2d(e, b.j_gpu_model, pend->gpu_model, P);
  if ((b.hj_group = pend->group != nullptr && G > 0)) h2d(e, b.j_group, pend->group, P);
  if ((b.hj_eq = pend->eq_off != nullptr && P > 0)) {
    h2d(e, b.j_eq_off, pend->eq_off, P + 1);
    h2d(e, b.j_eq_key, pend->eq_key, std::max(1u, pend->eq_off[P]));
    h2d(e, b.j_eq_val, pend->eq_val, std::max(1u, pend->eq_off[P]));
  }
  if ((b.hj_novel = pend->novel_off != nullptr && P > 0)) {
    h2d(e, b.j_novel_off, pend->novel_off, P + 1);
    h2d(e, b.j_novel_host, pend->novel_host, std::max(1u, pend->novel_off[P]));
  }
  if ((b.hj_ckpt = pend->ckpt_location != nullptr)) h2d(e, b.j_ckpt, pend->ckpt_location, P);
  if ((b.hj_est = pend->est_end_ms != nullptr)) h2d(e, b.j_est_end, pend->est_end_ms, P);
  if ((b.hj_disk = pend->disk_request != nullptr && pend->disk_type != nullptr)) {
    h2d(e, b.j_disk_req, pend->disk_request, P);
    h2d(e, b.j_disk_type, pend->disk_type, P);
  }
  if (pend->group && G)
    for (unsigned p = 0; p < P; ++p)
      if (pend->group[p] != COOK_NONE_U32 && pend->group[p] >= G) e->fail(COOK_E_INVALID, "cook_rebalance: group id out of range");
  // ---- groups --------------------------------------------------------------------------------------------------------------------
  unsigned max_run = 0;
  b.hg_run = false;
  if (G) {
    if (!groups->type || !groups->attr_key || !groups->minimum) e->fail(COOK_E_INVALID, "cook_rebalance: groups need type, attr_key, minimum");
    h2d(e, b.g_type, groups->type, G);
    h2d(e, b.g_attr_key, groups->attr_key, G);
    h2d(e, b.g_min, groups->minimum, G);
    if ((b.hg_run = groups->run_off != nullptr)) {
      h2d(e, b.g_run_off, groups->run_off, G + 1);
      h2d(e, b.g_run_host, groups->run_host, std::max(1u, groups->run_off[G]));
      for (unsigned x = 0; x < G; ++x) max_run = std::max(max_run, groups->run_off[x + 1] - groups->run_off[x]);
    }
  }
  b.co_cap = S + max_run + 1;
  sync(e);  // host temporaries
  b.staged = true;
}

RebalIn rebalance_args(cook_engine* e, RebalBufs& b) {
  RebalIn in;
  std::memset(&in, 0, sizeof(in));
  in.R = b.R, in.P = b.P, in.S = b.S, in.U = b.U, in.H = b.H;
  in.dru_mode = e->params.dru_mode;
  in.host_lifetime_mins = e->params.host_lifetime_mins;
  in.safe_dru = b.rp.safe_dru_threshold;
  in.min_diff = b.rp.min_dru_diff;
  in.slot_user = b.user.ptr();
  in.slot_cpus = b.cpus.ptr();
  in.slot_mem = b.mem.ptr();
  in.slot_gpus = b.gpus.ptr();
  in.posB = b.posB.ptr();
  in.attrs_cached = b.has_cached ? b.cached.ptr() : nullptr;
  in.s_use = b.s_use.ptr();
  in.seg_start = b.seg_start.ptr();
  in.seg_end = b.seg_end.ptr();
  in.act = b.act.ptr();
  in.dru = b.dru.ptr();
  in.pre = b.pre.ptr();
  in.user_safe = b.user_safe.ptr();
  in.q_count = b.qcount.ptr(), in.q_cpus = b.qcpus.ptr(), in.q_mem = b.qmem.ptr(), in.q_gpus = b.qgpus.ptr();
  in.div_cpus = b.divc.ptr(), in.div_mem = b.divm.ptr(), in.div_gpus = b.divg.ptr();
  in.hperm = b.hperm;
  in.hstart = b.hstart.ptr(), in.hend = b.hend.ptr();
  in.row_of_host = b.row_of_host.ptr();
  in.spare_c = b.spare_c.ptr(), in.spare_m = b.spare_m.ptr(), in.spare_g = b.spare_g.ptr();
  in.has_spare = b.has_spare.ptr();
  in.n_attr = b.n_attr;
  if (b.has_attrs) {
    in.a_host = b.a_host.ptr();
    in.a_k8s = b.ha_k8s ? b.a_k8s.ptr() : nullptr;
    in.a_gpu_model = b.ha_gpu ? b.a_gpu_model.ptr() : nullptr;
    in.a_gpu_count = b.ha_gpu ? b.a_gpu_count.ptr() : nullptr;
    in.a_disk_type = b.ha_disk ? b.a_disk_type.ptr() : nullptr;
    in.a_disk_space = b.ha_disk ? b.a_disk_space.ptr() : nullptr;
    in.a_gpu_slots = b.a_gpu_slots;
    in.a_disk_slots = b.a_disk_slots;
    in.a_attr = b.ha_attr ? b.a_attr.ptr() : nullptr;
    in.a_location = b.ha_loc ? b.a_location.ptr() : nullptr;
    in.a_host_start = b.ha_start ? b.a_host_start.ptr() : nullptr;
  }
  in.j_cpus = b.j_cpus.ptr(), in.j_mem = b.j_mem.ptr();
  in.j_gpus = b.hj_gpus ? b.j_gpus.ptr() : nullptr;
  in.j_gpu_model = b.hj_gpu_model ? b.j_gpu_model.ptr() : nullptr;
  in.j_user = b.j_user.ptr();
  in.j_group = b.hj_group ? b.j_group.ptr() : nullptr;
  in.j_eq_off = b.hj_eq ? b.j_eq_off.ptr() : nullptr;
  in.j_eq_key = b.hj_eq ? b.j_eq_key.ptr() : nullptr;
  in.j_eq_val = b.hj_eq ? b.j_eq_val.ptr() : nullptr;
  in.j_novel_off = b.hj_novel ? b.j_novel_off.ptr() : nullptr;
  in.j_novel_host = b.hj_novel ? b.j_novel_host.ptr() : nullptr;
  in.j_ckpt = b.hj_ckpt ? b.j_ckpt.ptr() : nullptr;
  in.j_est_end = b.hj_est ? b.j_est_end.ptr() : nullptr;
  in.j_disk_req = b.hj_disk ? b.j_disk_req.ptr() : nullptr;
  in.j_disk_type = b.hj_disk ? b.j_disk_type.ptr() : nullptr;
  in.G = b.G;
  if (b.G) {
    in.g_type = b.g_type.ptr();
    in.g_attr_key = b.g_attr_key.ptr();
    in.g_min = b.g_min.ptr();
    in.g_run_off = b.hg_run ? b.g_run_off.ptr() : nullptr;
    in.g_run_host = b.hg_run ? b.g_run_host.ptr() : nullptr;
  }
  in.x_pj = b.x_pj.ptr(), in.x_host = b.x_host.ptr(), in.x_known = b.x_known.ptr();
  in.pre_hosts = b.pre_hosts.ptr(), in.co_val = b.co_val.ptr();
  in.hres_key = b.hres_key.ptr(), in.hres_len = b.hres_len.ptr(), in.hres_base = b.hres_base.ptr();
  in.blk_key = b.blk_key.ptr(), in.blk_host = b.blk_host.ptr(), in.n_blk = in.H ? div_up(div_up(in.H, 2u), (unsigned)RB_PAIRS) : 0u;
  in.hmax_key = b.hmax_key.ptr(), in.h_host = b.h_host.ptr(), in.best_key = b.best_key.ptr(), in.thr_key = b.thr_key;
  in.hres_dru = b.hres_dru.ptr(), in.hres_c = b.hres_c.ptr(), in.hres_m = b.hres_m.ptr(), in.hres_g = b.hres_g.ptr();
  in.srt_slot = b.srt_slot.ptr();
  in.gs_dru = b.gs_dru.ptr(), in.gs_cpus = b.gs_cpus.ptr(), in.gs_mem = b.gs_mem.ptr(), in.gs_gpus = b.gs_gpus.ptr();
  in.gs_posB = b.gs_posB.ptr(), in.gs_slot = b.gs_slot.ptr(), in.gs_ord = b.gs_ord.ptr();
  in.decisions = b.decisions.ptr();
  in.preempted = b.preempted.ptr();
  in.pending_dru = b.pending_dru.ptr();
  in.ctl = b.ctl.ptr();
  in.job = b.jobctx.ptr();
  in.hbase = b.hbase.ptr();
  in.h_pb = b.h_pb.ptr(), in.h_user = b.h_user.ptr();
  in.h_cpus = b.h_cpus.ptr(), in.h_mem = b.h_mem.ptr(), in.h_gpus = b.h_gpus.ptr();
  in.h_dru = b.h_dru.ptr();
  in.h_act = b.h_act.ptr();
  in.hidx = b.hidx.ptr();
  in.chg = b.chg.ptr();
  in.chg_tile = b.chg_tile.ptr();
  in.chg_bad = b.chg_bad.ptr();
  in.chg_mark = b.chg_mark.ptr();
  in.tile_agg = b.tile_agg.ptr();
  in.tile_carry = b.tile_carry.ptr();
  in.x_head = b.x_head.ptr();
  in.x_next = b.x_next.ptr();
  in.big_list = b.big_list.ptr();
  in.x_cnt = b.x_cnt.ptr();
  in.pre_w = b.pre.ptr();
  in.dru_w = b.dru.ptr();
  in.dl_pos = b.dl_pos.ptr();
  in.dl_sign = b.dl_sign.ptr();
  return in;
}

// masked per-user prefix sums -> DRUs (dru.clj:50-80 over the active slots), exact for any fp64 input
void rebalance_rescore(cook_engine* e, RebalBufs& b) {
  const unsigned S = b.S, U = b.U;
  seg_scan<SumU4>(e, "rebal_usage_scan", LoadMaskedU4{b.s_use.ptr(), b.act.ptr()}, (const uint8_t*)b.head.ptr(), S, b.pre.ptr(), e->tmpU4);
  KM<rank_mark_inexact, 256>(e, "rank_mark_inexact", div_up(S, 256), (const SumU4*)b.pre.ptr(), (const uint32_t*)b.s_user.ptr(), S, b.inexact.ptr());
  KL("rebal_fix_inexact", rebal_fix_inexact, div_up(U, 256), 256, (const SumU4*)b.s_use.ptr(), (const uint8_t*)b.act.ptr(), b.pre.ptr(),
     (const uint32_t*)b.seg_start.ptr(), (const uint32_t*)b.seg_end.ptr(), b.inexact.ptr(), U);
  KL("rebal_score", rebal_score, div_up(S, 256), 256, (const SumU4*)b.pre.ptr(), (const uint32_t*)b.s_user.ptr(), S, (int)e->params.dru_mode,
     (const double*)b.divc.ptr(), (const double*)b.divm.ptr(), (const double*)b.divg.ptr(), b.dru.ptr());
}

void rebalance_run(cook_engine* e, RebalBufs& b) {
  if (!b.staged) e->fail(COOK_E_STATE, "cook_rebalance_run before cook_rebalance_stage");
  const unsigned R = b.R, P = b.P, S = b.S, U = b.U, H = b.H;
  b.done = false;
  std::memset(&b.last, 0, sizeof(b.last));
  b.decisions.ensure(std::max(1u, P));
  b.preempted.ensure(std::max(1u, S));
  b.pending_dru.ensure(std::max(1u, P));
  if (P == 0 || b.rp.max_preemption <= 0) {
    if (P) {
      std::vector<double> nanv(P, std::numeric_limits<double>::quiet_NaN());
      copy_async(e, b.pending_dru.ptr(), nanv.data(), (size_t)P * 8, hipMemcpyHostToDevice);
      sync(e);
    }
    b.done = true;
    return;
  }
  const unsigned gS = div_up(S, 256);
  e->d_scratch64.ensure(64);
  // ---- per-user order of all slots (tools.clj:614-641; rebalancer.clj:241-246) ---------------------------------------------
  unsigned long long* mins = e->d_scratch64.ptr();
  unsigned long long* same = e->d_scratch64.ptr() + 4;  // bits on which all keys of a word agree
  memset_async(e, mins, 0xFF, 7 * 8);
  e->w0.ensure(S);
  e->w1.ensure(S);
  e->w2.ensure(S);
  KM<rank_key_mins, 256>(e, "rank_key_mins", std::min(gS, 128u), (const int64_t*)b.start.ptr(), (const int64_t*)b.task.ptr(),
      (const int64_t*)b.job.ptr(), (const uint8_t*)b.pending.ptr(), S, mins, std::min(gS, 128u));
  KM<rank_build_keys, 256>(e, "rank_build_keys", gS, (const uint32_t*)b.user.ptr(), (const int32_t*)b.prio.ptr(), (const int64_t*)b.start.ptr(),
      (const int64_t*)b.task.ptr(), (const int64_t*)b.job.ptr(), (const uint8_t*)b.pending.ptr(), S, (const unsigned long long*)mins, e->w0.ptr(),
      e->w1.ptr(), e->w2.ptr(), same);
  readback64(e, 8);
  const unsigned long long mk0 = ~e->h_scratch[4], mk1 = ~e->h_scratch[5], mk2 = ~e->h_scratch[6];
  b.permA.ensure(S);
  b.permB2.ensure(S);
  const uint32_t* cur = nullptr;  // the identity
  cur = radix_sort_masked(e, e->w2.ptr(), mk2, cur, b.permA.ptr(), b.permB2.ptr(), S);
  cur = radix_sort_masked(e, e->w1.ptr(), mk1, cur, b.permA.ptr(), b.permB2.ptr(), S);
  cur = radix_sort_masked(e, e->w0.ptr(), mk0, cur, b.permA.ptr(), b.permB2.ptr(), S);
  if (!cur) {
    KM<iota_u32, 256>(e, "iota", gS, b.permA.ptr(), S);
    cur = b.permA.ptr();
  }
  const uint32_t* permB = cur;
  b.posB.ensure(S);
  b.act.ensure(S);
  b.s_user.ensure(S);
  b.s_use.ensure(S);
  b.s_pending.ensure(S);
  b.head.ensure(S);
  b.seg_start.ensure(U);
  b.seg_end.ensure(U);
  b.pre.ensure(S);
  b.dru.ensure(S);
  b.inexact.ensure(U);
  memset_async(e, b.seg_start.ptr(), 0, (size_t)U * 4);
  memset_async(e, b.seg_end.ptr(), 0, (size_t)U * 4);
  memset_async(e, b.inexact.ptr(), 0, (size_t)U * 4);
  KL("rebal_invert_perm", rebal_invert_perm, gS, 256, permB, S, R, b.posB.ptr(), b.act.ptr());
  KM<rank_gather, 256>(e, "rank_gather", gS, permB, S, (const uint32_t*)b.user.ptr(), (const double*)b.cpus.ptr(), (const double*)b.mem.ptr(),
      (const double*)b.gpus.ptr(), (const uint8_t*)b.pending.ptr(), b.s_user.ptr(), b.s_use.ptr(), b.s_pending.ptr(), b.head.ptr(),
      b.seg_start.ptr(), b.seg_end.ptr());
  {  // users whose sums are exact in any order (all of them for integer-valued resources)
    std::vector<uint32_t> ones(std::max(1u, U), 1u);
    b.user_safe.ensure(std::max(1u, U));
    copy_async(e, b.user_safe.ptr(), ones.data(), (size_t)std::max(1u, U) * 4, hipMemcpyHostToDevice);
    sync(e);
    KL("rebal_user_safe", rebal_user_safe, gS, 256, (const uint32_t*)b.user.ptr(), (const double*)b.cpus.ptr(), (const double*)b.mem.ptr(),
       (const double*)b.gpus.ptr(), S, (const uint32_t*)b.seg_start.ptr(), (const uint32_t*)b.seg_end.ptr(), b.user_safe.ptr());
    b.user_safe_h.assign(std::max(1u, U), 1u);
    copy_async(e, b.user_safe_h.data(), b.user_safe.ptr(), (size_t)std::max(1u, U) * 4, hipMemcpyDeviceToHost);  // read after the sync below
  }
  // ---- running tasks grouped by host (the group-by of rebalancer.clj:349, done once) --------------------------------------------
  b.hstart.ensure(std::max(1u, H));
  b.hend.ensure(std::max(1u, H));
  memset_async(e, b.hstart.ptr(), 0, (size_t)std::max(1u, H) * 4);
  memset_async(e, b.hend.ptr(), 0, (size_t)std::max(1u, H) * 4);
  b.hpermA.ensure(std::max(1u, R));
  b.hpermB.ensure(std::max(1u, R));
  b.hperm = b.hpermA.ptr();
  if (R) {
    b.hkey.ensure(R);
    const unsigned gR = div_up(R, 256);
    KL("rebal_host_keys", rebal_host_keys, gR, 256, (const uint32_t*)b.host.ptr(), R, b.hkey.ptr());
    KM<iota_u32, 256>(e, "iota", gR, b.hpermA.ptr(), R);
    unsigned long long hmask = 0;
    for (unsigned long long x = H ? H - 1 : 0; x; x >>= 1) hmask = (hmask << 1) | 1ull;
    b.hperm = radix_sort_masked(e, b.hkey.ptr(), hmask, b.hpermA.ptr(), b.hpermA.ptr(), b.hpermB.ptr(), R);
    KL("rebal_host_bounds", rebal_host_bounds, gR, 256, (const uint32_t*)b.hperm, (const uint32_t*)b.host.ptr(), R, b.hstart.ptr(),
       b.hend.ptr());
  }
  b.hbase.ensure(std::max(1u, H));
  if (H) {
    KL("rebal_host_sizes", rebal_host_sizes, div_up(H, 256), 256, (const uint32_t*)b.hstart.ptr(), (const uint32_t*)b.hend.ptr(), H, b.hbase.ptr());
    KM<excl_scan_u32_single, SCAN1_THREADS>(e, "rebal_hbase_scan", 1, b.hbase.ptr(), H, (uint32_t*)nullptr);
  }
  // host-ordered mirrors of the running slots' columns (static ones now, the DRUs after the first scoring)
  b.h_pb.ensure(std::max(1u, R)), b.h_user.ensure(std::max(1u, R));
  b.h_cpus.ensure(std::max(1u, R)), b.h_mem.ensure(std::max(1u, R)), b.h_gpus.ensure(std::max(1u, R)), b.h_dru.ensure(std::max(1u, R));
  b.h_act.ensure(std::max(1u, R));
  b.hidx.ensure(S);
  b.chg.ensure(S + 1), b.chg_tile.ensure(S + 2), b.chg_bad.ensure(S + 1), b.chg_mark.ensure(std::max(1u, U));
  b.tile_agg.ensure(S / RB_RS_TILE + S + 2), b.tile_carry.ensure(S / RB_RS_TILE + S + 2);  // every listed user adds at most one partial tile
  b.dl_pos.ensure(S + 1), b.dl_sign.ensure(S + 1);
  b.x_head.ensure(std::max(1u, H)), b.x_cnt.ensure(std::max(1u, H)), b.x_next.ensure(std::max(1u, P)), b.big_list.ensure(std::max(1u, H));
  memset_async(e, b.chg_mark.ptr(), 0, (size_t)std::max(1u, U) * 4);
  memset_async(e, b.x_head.ptr(), 0xFF, (size_t)std::max(1u, H) * 4);
  memset_async(e, b.x_cnt.ptr(), 0, (size_t)std::max(1u, H) * 4);
  memset_async(e, b.hidx.ptr(), 0xFF, (size_t)S * 4);
  if (R)
    KL("rebal_host_mirror", rebal_host_mirror, div_up(R, 256), 256, (const uint32_t*)b.hperm, (const uint32_t*)b.posB.ptr(),
       (const uint32_t*)b.user.ptr(), (const double*)b.cpus.ptr(), (const double*)b.mem.ptr(), (const double*)b.gpus.ptr(), R, b.h_pb.ptr(),
       b.h_user.ptr(), b.h_cpus.ptr(), b.h_mem.ptr(), b.h_gpus.ptr(), b.h_act.ptr(), b.hidx.ptr());
  // ---- dynamic state -----------------------------------------------------------------------------------------------------------
  b.x_pj.ensure(P);
  b.x_host.ensure(P);
  b.x_known.ensure(P);
  b.pre_hosts.ensure(S);
  b.co_val.ensure(b.co_cap);
  b.hres_key.ensure(std::max(1u, H));
  b.blk_key.ensure(2u * std::max(1u, div_up(div_up(std::max(1u, H), 2u), (unsigned)RB_PAIRS)));  // (an entry per workgroup and phase of rebal_decide)
  b.blk_host.ensure(2u * std::max(1u, div_up(div_up(std::max(1u, H), 2u), (unsigned)RB_PAIRS)));
  b.hmax_key.ensure(std::max(1u, H)), b.h_host.ensure(std::max(1u, R)), b.best_key.ensure(4);
  b.hres_len.ensure(std::max(1u, H));
  b.hres_base.ensure(std::max(1u, H));
  b.hres_dru.ensure(std::max(1u, H));
  b.hres_c.ensure(std::max(1u, H));
  b.hres_m.ensure(std::max(1u, H));
  b.hres_g.ensure(std::max(1u, H));
  b.srt_slot.ensure(S);
  b.gs_dru.ensure(S), b.gs_cpus.ensure(S), b.gs_mem.ensure(S), b.gs_gpus.ensure(S);
  b.gs_posB.ensure(S), b.gs_slot.ensure(S), b.gs_ord.ensure(S);
  b.ctl.ensure(1);
  b.jobctx.ensure(1);
  memset_async(e, b.hres_key.ptr(), 0, (size_t)std::max(1u, H) * 8);
  if (H) {  // host->spare-resources as staged: a run is repeatable on the same staged inputs
    copy_async(e, b.spare_c.ptr(), b.spare0_c.ptr(), (size_t)H * 8, hipMemcpyDeviceToDevice);
    copy_async(e, b.spare_m.ptr(), b.spare0_m.ptr(), (size_t)H * 8, hipMemcpyDeviceToDevice);
    copy_async(e, b.spare_g.ptr(), b.spare0_g.ptr(), (size_t)H * 8, hipMemcpyDeviceToDevice);
    copy_async(e, b.has_spare.ptr(), b.has_spare0.ptr(), (size_t)H, hipMemcpyDeviceToDevice);
  }
  RebalCtl c0;
  std::memset(&c0, 0, sizeof(c0));
  c0.remaining = b.rp.max_preemption;
  c0.max_items = b.max_seg;
  std::memcpy(e->h_scratch, &c0, sizeof(c0));
  copy_async(e, b.ctl.ptr(), e->h_scratch, sizeof(c0), hipMemcpyHostToDevice);
  std::vector<double> nanv(P, std::numeric_limits<double>::quiet_NaN());
  copy_async(e, b.pending_dru.ptr(), nanv.data(), (size_t)P * 8, hipMemcpyHostToDevice);
  sync(e);  // nanv / h_scratch are reused below
  if (H) KL("rebal_big_init", rebal_big_init, div_up(H, 256), 256, (const uint32_t*)b.hstart.ptr(), (const uint32_t*)b.hend.ptr(), H, b.big_list.ptr(), b.ctl.ptr());
  rebalance_rescore(e, b);  // every user once; after a decision only the users it touched (rebal_rescore_users)
  if (R) KL("rebal_mirror_dru", rebal_mirror_dru, div_up(R, 256), 256, (const uint32_t*)b.h_pb.ptr(), (const double*)b.dru.ptr(), R, b.h_dru.ptr());
  // the hosts' bounds (the greatest DRU a host holds), and the threshold of the first phase of rebal_decide: the bound about 3 % of the hosts reach.
  // Any threshold gives the same decisions (rebal_decide); this one makes the first phase small and its best key usually the decision's.
  b.thr_key = 0ull;
  memset_async(e, b.best_key.ptr(), 0, 32);
  if (H) {
    KL("rebal_host_bound_init", rebal_host_bound_init, div_up(H, 256), 256, (const uint32_t*)b.hstart.ptr(), (const uint32_t*)b.hend.ptr(), H,
       (const double*)b.h_dru.ptr(), b.h_host.ptr(), b.hmax_key.ptr());
    const unsigned n_first = std::max(RB_FIRST_MIN, H / 32u);
    if (H > 4u * n_first && std::getenv("COOK_REBAL_ONE_PHASE") == nullptr) {
      b.hmax_h.resize(H);
      copy_async(e, b.hmax_h.data(), b.hmax_key.ptr(), (size_t)H * 8, hipMemcpyDeviceToHost);
      sync(e);
      std::nth_element(b.hmax_h.begin(), b.hmax_h.begin() + (n_first - 1u), b.hmax_h.end(), std::greater<unsigned long long>());
      b.thr_key = b.hmax_h[n_first - 1u];
    }
  }
  // ---- the decision loop (rebalancer.clj:442-458) ---------------------------------------------------------------------------------
  // the kernels' argument block lives on the device: a launch passes its address (the block by value is ~1 KB of kernel arguments per launch, and the
  // decision loop is bound by the rate at which the host can enqueue)
  b.in_host = rebalance_args(e, b);
  b.in_dev.ensure(1);
  copy_async(e, b.in_dev.ptr(), &b.in_host, sizeof(RebalIn), hipMemcpyHostToDevice);
  const RebalIn* in = b.in_dev.ptr();
  unsigned known_max = b.max_seg, known_at = 0;  // items of the fullest host when the control block was last read back
  // every user safe (integer-valued resources, the usual case): the changed users are re-scored by ONE launch from the slots the decision
  // flipped (rebal_rs_delta) and rebal_apply prepares the next job itself — 3 launches per pending job instead of 7.  Otherwise the general
  // path: tile scans with exactness tracking, left-to-right redo of the users where an addition rounded.
  bool all_safe = std::getenv("COOK_REBAL_GENERAL") == nullptr;
  for (unsigned u = 0; u < U && all_safe; ++u) all_safe = b.user_safe_h[u] != 0u;
  const bool trace_loop = std::getenv("COOK_REBAL_TRACE") != nullptr;
  if (trace_loop) sync(e);
  const auto t_loop0 = std::chrono::steady_clock::now();
  if (all_safe && P) KL("rebal_job_prep", rebal_job_prep, 1, COOK_WAVE, in, 0u);
  for (unsigned pj = 0; pj < P; ++pj) {
    if (!all_safe) KL("rebal_job_prep", rebal_job_prep, 1, COOK_WAVE, in, pj);
    for (unsigned phase = 0; H && phase < (b.thr_key != 0ull ? 2u : 1u); ++phase) {
      if (all_safe && b.spare_safe) KL("rebal_decide", rebal_decide<true>, div_up(div_up(H, 2u), (unsigned)RB_PAIRS), COOK_WAVE * RB_WAVES, in, phase);
      else KL("rebal_decide", rebal_decide<false>, div_up(div_up(H, 2u), (unsigned)RB_PAIRS), COOK_WAVE * RB_WAVES, in, phase);
    }
    // hosts beyond 64 items: only when one can exist (the fullest host as last read back + the jobs placed since then)
    if (H && known_max + (pj - known_at) > (unsigned)COOK_WAVE) KL("rebal_decide_big", rebal_decide_big, 32, COOK_WAVE * RB_WAVES, in);
    if (all_safe) {
      KL("rebal_apply", rebal_apply, 1, RB_APPLY_THREADS, in, pj + 1 < P ? pj + 1 : 0xFFFFFFFFu);
      KL("rebal_rs_delta", rebal_rs_delta, RB_RS_GRID, RB_RS_TILE, in);
    } else {
      KL("rebal_apply", rebal_apply, 1, RB_APPLY_THREADS, in, 0xFFFFFFFFu);
      KL("rebal_rs_local", rebal_rs_local, RB_RS_GRID, RB_RS_TILE, in);
      KL("rebal_rs_carry", rebal_rs_carry, RB_RS_USERS, COOK_WAVE, in);
      KL("rebal_rs_finish", rebal_rs_finish, RB_RS_GRID, RB_RS_TILE, in);
      KL("rebal_rs_fix", rebal_rs_fix, RB_RS_USERS, 256, in);
    }
    if ((pj + 1) % RB_CHECK == 0 && pj + 1 < P) {
      copy_async(e, e->h_scratch, b.ctl.ptr(), sizeof(RebalCtl), hipMemcpyDeviceToHost);
      sync(e);
      RebalCtl c;
      std::memcpy(&c, e->h_scratch, sizeof(c));
      if (c.remaining <= 0) break;
      known_max = c.max_items;
      known_at = pj + 1;
    }
  }
  const auto t_loop1 = std::chrono::steady_clock::now();
  copy_async(e, e->h_scratch, b.ctl.ptr(), sizeof(RebalCtl), hipMemcpyDeviceToHost);
  sync(e);
  if (trace_loop)
    std::fprintf(stderr, "COOK_REBAL_TRACE: the decision loop: %.3f ms to enqueue, %.3f ms until the device is done (two-phase threshold %s)\n",
                 std::chrono::duration<double, std::milli>(t_loop1 - t_loop0).count(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop0).count(), b.thr_key ? "set" : "none");
  std::memcpy(&b.last, e->h_scratch, sizeof(RebalCtl));
#ifdef RB_COUNT
  {
    unsigned long long cnt[4];
    copy_async(e, cnt, b.best_key.ptr(), 32, hipMemcpyDeviceToHost);
    sync(e);
    std::fprintf(stderr, "RB_COUNT: waves that evaluated: first phase %llu, second phase %llu (of %u per phase and job; %u jobs), of them for jobs below quota %llu; threshold key %llx\n", cnt[1], cnt[2],
                 div_up(H, 2u), P, cnt[3], b.thr_key);
  }
#endif
  b.done = true;
}

void rebalance_fetch(cook_engine* e, RebalBufs& b, cook_preemption* decisions, uint32_t* n_decisions, uint32_t* preempted,
                     uint32_t* n_preempted, double* pending_dru) {
  if (!b.done) e->fail(COOK_E_STATE, "cook_rebalance_fetch before cook_rebalance_run");
  if (n_decisions) *n_decisions = b.last.nd;
  if (n_preempted) *n_preempted = b.last.np;
  if (decisions && b.last.nd)
    copy_async(e, decisions, b.decisions.ptr(), (size_t)b.last.nd * sizeof(cook_preemption), hipMemcpyDeviceToHost);
  if (preempted && b.last.np)
    copy_async(e, preempted, b.preempted.ptr(), (size_t)b.last.np * 4, hipMemcpyDeviceToHost);
  if (pending_dru && b.P)
    copy_async(e, pending_dru, b.pending_dru.ptr(), (size_t)b.P * 8, hipMemcpyDeviceToHost);
  sync(e);
}

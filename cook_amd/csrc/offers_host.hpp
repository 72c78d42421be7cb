// offers_host.hpp — host orchestration of cook_offers_* (included by engine.hip inside its anonymous namespace).
#pragma once

struct OfferBufs {
  bool staged = false, done = false;
  unsigned Nn = 0, Np = 0, n_attr = 0, n_offers = 0;
  cook_offer_params params{};
  double ms = 0;
  // inputs
  DArr<double> n_cpus, n_mem, n_disk, p_cpus, p_mem, p_disk;
  DArr<int32_t> n_gpus, p_gpus;
  DArr<uint32_t> n_host, n_gpu_model, n_disk_type, n_attr_tab, p_node, p_gpu_model, p_disk_type;
  DArr<uint8_t> n_flags, p_flags;
  NodeCols nd{};
  PodCols pd{};
  const uint32_t* d_host = nullptr;
  const uint32_t* d_attr = nullptr;
  // work
  DArr<uint64_t> key;
  DArr<uint32_t> permA, permB, seg_start, seg_end, block_offers;
  DArr<PodRec> podrec;
  DArr<double> a_cpus, a_mem, a_cons_cpus, a_cons_mem, a_gpu_count, a_disk_space, a_disk_cons;
  DArr<uint32_t> a_gpu_model, a_disk_type;
  unsigned gpu_slots = 1, disk_slots = 1;
  DArr<int32_t> a_num_pods;
  DArr<uint8_t> a_status;
  DArr<unsigned long long> gpu_cap, gpu_cons;
  DArr<double> disk_cap, disk_cons;
  DArr<OfferTotalsDev> totals;
  // output rows
  DArr<uint32_t> o_node, o_host, o_gpu_model, o_disk_type, o_attr;
  DArr<double> o_cpus, o_mem, o_gpu_count, o_disk_space;
  DArr<int32_t> o_num_pods, o_max_tasks;
  DArr<uint8_t> o_k8s;  // the two columns every Kubernetes lease carries, filled when the rows feed a match in place
};

void offers_stage(cook_engine* e, OfferBufs& b, const cook_nodes* nodes, const cook_pods* pods, const cook_offer_params* params) {
  if (!nodes || !pods || !params) e->fail(COOK_E_INVALID, "cook_offers_stage: null nodes / pods / params");
  const unsigned Nn = nodes->n, Np = pods->n;
  if (Nn && (!nodes->cpus || !nodes->mem)) e->fail(COOK_E_INVALID, "cook_offers_stage: node cpus / mem are required");
  if (Np && (!pods->node || !pods->cpus || !pods->mem)) e->fail(COOK_E_INVALID, "cook_offers_stage: pod node / cpus / mem are required");
  if (nodes->n_attr_keys && Nn && !nodes->attr) e->fail(COOK_E_INVALID, "cook_offers_stage: n_attr_keys without a label table");
  if (params->max_pods_per_node < 0) e->fail(COOK_E_INVALID, "cook_offers_stage: negative max_pods_per_node");
  b.staged = b.done = false;
  b.Nn = Nn;
  b.Np = Np;
  b.params = *params;
  b.gpu_slots = res_slots(e, params->gpu_slots, "cook_offers_stage: gpu_slots");
  b.disk_slots = res_slots(e, params->disk_slots, "cook_offers_stage: disk_slots");
  b.n_attr = nodes->attr ? nodes->n_attr_keys : 0u;
  h2d(e, b.n_cpus, nodes->cpus, Nn);
  h2d(e, b.n_mem, nodes->mem, Nn);
  b.nd = NodeCols{b.n_cpus.ptr(), b.n_mem.ptr(), h2d_opt(e, b.n_gpus, nodes->gpus, Nn), h2d_opt(e, b.n_gpu_model, nodes->gpu_model, Nn),
                  h2d_opt(e, b.n_disk, nodes->disk, Nn), h2d_opt(e, b.n_disk_type, nodes->disk_type, Nn),
                  h2d_opt(e, b.n_flags, nodes->flags, Nn), Nn};
  b.d_host = h2d_opt(e, b.n_host, nodes->host, Nn);
  b.d_attr = b.n_attr ? h2d_opt(e, b.n_attr_tab, nodes->attr, (size_t)Nn * b.n_attr) : nullptr;
  h2d(e, b.p_node, pods->node, Np);
  h2d(e, b.p_cpus, pods->cpus, Np);
  h2d(e, b.p_mem, pods->mem, Np);
  b.pd = PodCols{b.p_node.ptr(), b.p_cpus.ptr(), b.p_mem.ptr(), h2d_opt(e, b.p_gpus, pods->gpus, Np),
                 h2d_opt(e, b.p_gpu_model, pods->gpu_model, Np), h2d_opt(e, b.p_disk, pods->disk, Np),
                 h2d_opt(e, b.p_disk_type, pods->disk_type, Np), h2d_opt(e, b.p_flags, pods->flags, Np), Np};
  sync(e);
  b.staged = true;
}

void offers_run(cook_engine* e, OfferBufs& b) {
  if (!b.staged) e->fail(COOK_E_STATE, "cook_offers_run before cook_offers_stage");
  const unsigned Nn = b.Nn, Np = b.Np;
  const unsigned G = b.params.n_gpu_models + 1u, D = b.params.n_disk_types + 1u;
  b.done = false;
  b.n_offers = 0;
  b.gpu_cap.ensure(G);
  b.gpu_cons.ensure(G);
  b.disk_cap.ensure(D);
  b.disk_cons.ensure(D);
  b.totals.ensure(1);
  memset_async(e, b.gpu_cap.ptr(), 0, (size_t)G * 8);
  memset_async(e, b.gpu_cons.ptr(), 0, (size_t)G * 8);
  memset_async(e, b.disk_cap.ptr(), 0, (size_t)D * 8);
  memset_async(e, b.disk_cons.ptr(), 0, (size_t)D * 8);
  memset_async(e, b.totals.ptr(), 0, sizeof(OfferTotalsDev));
  if (Nn == 0) {
    b.done = true;
    return;
  }
  const unsigned gN = div_up(Nn, 256);
  // ---- pods stably partitioned by node: within a node the list order of node-name->pods survives ----------------------------
  b.seg_start.ensure(Nn);
  b.seg_end.ensure(Nn);
  memset_async(e, b.seg_start.ptr(), 0, (size_t)Nn * 4);
  memset_async(e, b.seg_end.ptr(), 0, (size_t)Nn * 4);
  PodRec* podrec = b.podrec.ensure(Np);
  if (Np) {
    const uint32_t* perm = nullptr;
    const unsigned gP = div_up(Np, 256);
    b.key.ensure(Np);
    b.permA.ensure(Np);
    b.permB.ensure(Np);
    KL("offers_pod_keys", offers_pod_keys, gP, 256, b.pd.node, Np, Nn, b.key.ptr());
    KM<iota_u32, 256>(e, "iota", gP, b.permA.ptr(), Np);
    unsigned long long mask = 0;
    for (unsigned long long x = Nn; x; x >>= 1) mask = (mask << 1) | 1ull;  // keys are 0..Nn
    perm = radix_sort_masked(e, b.key.ptr(), mask, b.permA.ptr(), b.permA.ptr(), b.permB.ptr(), Np);
    KL("offers_seg_bounds", offers_seg_bounds, gP, 256, perm, (const uint64_t*)b.key.ptr(), Np, Nn, b.seg_start.ptr(), b.seg_end.ptr());
    KL("offers_gather_pods", offers_gather_pods, gP, 256, b.pd, perm, podrec);
  }
  // ---- per node: capacity, consumption, available, schedulable --------------------------------------------------------------
  const unsigned GS = b.gpu_slots, DS = b.disk_slots;
  NodeAvail av{b.a_cpus.ensure(Nn),
               b.a_mem.ensure(Nn),
               b.a_cons_cpus.ensure(Nn),
               b.a_cons_mem.ensure(Nn),
               b.a_gpu_model.ensure((size_t)Nn * GS),
               b.a_gpu_count.ensure((size_t)Nn * GS),
               b.a_disk_type.ensure((size_t)Nn * DS),
               b.a_disk_space.ensure((size_t)Nn * DS),
               b.a_disk_cons.ensure(Nn),
               b.a_num_pods.ensure(Nn),
               b.a_status.ensure(Nn)};
  b.block_offers.ensure(gN);
  KL("offers_node_eval", offers_node_eval, gN, 256, b.nd, (const PodRec*)podrec, (const uint32_t*)b.seg_start.ptr(),
     (const uint32_t*)b.seg_end.ptr(), b.params.clobber_synthetic_pods, b.params.filter_out_unsound_gpu_nodes, b.params.max_pods_per_node,
     b.params.n_gpu_models, GS, DS, av, b.gpu_cap.ptr(), b.gpu_cons.ptr(), b.block_offers.ptr());
  // ---- gauges ------------------------------------------------------------------------------------------------------------------
  KL("offers_totals", offers_totals, 1, OT_THREADS, b.nd, av, b.params.n_disk_types, b.totals.ptr(), b.disk_cap.ptr(), b.disk_cons.ptr());
  // ---- offer rows of the schedulable nodes, node order ----------------------------------------------------------------------------
  unsigned* d_total = e->d_counters.ptr() + 13;
  OfferRows rows{b.o_node.ensure(Nn),      b.o_host.ensure(Nn),       b.o_cpus.ensure(Nn),     b.o_mem.ensure(Nn),
                 b.o_gpu_model.ensure((size_t)Nn * GS), b.o_gpu_count.ensure((size_t)Nn * GS),  b.o_disk_type.ensure((size_t)Nn * DS), b.o_disk_space.ensure((size_t)Nn * DS),
                 b.o_num_pods.ensure(Nn),  b.o_attr.ensure((size_t)Nn * std::max(1u, b.n_attr))};
  KL("offers_emit", offers_emit, gN, 256, b.nd, b.d_host, av, (const uint32_t*)b.block_offers.ptr(), b.d_attr, b.n_attr, GS, DS, rows, d_total);
  copy_async(e, e->h_scratch, d_total, 4, hipMemcpyDeviceToHost);
  sync(e);
  std::memcpy(&b.n_offers, e->h_scratch, 4);
  b.done = true;
}

template <class T>
void offers_d2h(cook_engine* e, T* dst, const T* src, size_t n) {
  if (dst && n) copy_async(e, dst, src, n * sizeof(T), hipMemcpyDeviceToHost);
}

void offers_fetch(cook_engine* e, OfferBufs& b, cook_node_offers* o, uint32_t* n_offers, uint8_t* node_status, cook_offer_totals* totals,
                  int64_t* gpu_cap, int64_t* gpu_cons, double* disk_cap, double* disk_cons) {
  if (!b.done) e->fail(COOK_E_STATE, "cook_offers_fetch before cook_offers_run");
  const unsigned R = b.n_offers, G = b.params.n_gpu_models + 1u, D = b.params.n_disk_types + 1u;
  OfferTotalsDev t{};
  if (o && b.Nn) {
    offers_d2h(e, o->node, (const uint32_t*)b.o_node.ptr(), R);
    offers_d2h(e, o->host, (const uint32_t*)b.o_host.ptr(), R);
    offers_d2h(e, o->cpus, (const double*)b.o_cpus.ptr(), R);
    offers_d2h(e, o->mem, (const double*)b.o_mem.ptr(), R);
    offers_d2h(e, o->gpu_model, (const uint32_t*)b.o_gpu_model.ptr(), (size_t)R * b.gpu_slots);
    offers_d2h(e, o->gpu_count, (const double*)b.o_gpu_count.ptr(), (size_t)R * b.gpu_slots);
    offers_d2h(e, o->disk_type, (const uint32_t*)b.o_disk_type.ptr(), (size_t)R * b.disk_slots);
    offers_d2h(e, o->disk_space, (const double*)b.o_disk_space.ptr(), (size_t)R * b.disk_slots);
    offers_d2h(e, o->num_pods, (const int32_t*)b.o_num_pods.ptr(), R);
    if (b.n_attr) offers_d2h(e, o->attr, (const uint32_t*)b.o_attr.ptr(), (size_t)R * b.n_attr);
  }
  if (b.Nn) offers_d2h(e, node_status, (const uint8_t*)b.a_status.ptr(), b.Nn);
  static_assert(sizeof(unsigned long long) == sizeof(int64_t), "gpu totals are copied as 64-bit words");
  offers_d2h(e, (unsigned long long*)gpu_cap, (const unsigned long long*)b.gpu_cap.ptr(), G);
  offers_d2h(e, (unsigned long long*)gpu_cons, (const unsigned long long*)b.gpu_cons.ptr(), G);
  offers_d2h(e, disk_cap, (const double*)b.disk_cap.ptr(), D);
  offers_d2h(e, disk_cons, (const double*)b.disk_cons.ptr(), D);
  if (totals) copy_async(e, &t, b.totals.ptr(), sizeof(t), hipMemcpyDeviceToHost);
  sync(e);
  if (n_offers) *n_offers = R;
  if (totals) {
    totals->cpus_capacity = t.cpus_capacity;
    totals->mem_capacity = t.mem_capacity;
    totals->cpus_consumed = t.cpus_consumed;
    totals->mem_consumed = t.mem_consumed;
    totals->nodes_total = b.Nn;
    totals->nodes_schedulable = R;
  }
}

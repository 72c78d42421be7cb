// cook_oracle.cpp — CPU restatement of the reference algorithm for the fair-share match path.
//
// TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library; the product path (cook_amd/, libcookmatch.so) never does.
//
// Parity status: the Clojure/JVM reference (and the un-vendored jar com.netflix.fenzo:fenzo-core:0.10.0,
// scheduler/project.clj:46-50) cannot be built or run in this image (no JDK/lein/network), so this
// restatement is pinned against the known-answer vectors of the reference's own unit tests
// (tests/golden/*.json, transcribed by tests/golden/make_golden.py from scheduler/test/cook/test/...).
// Where the reference leaves behaviour undefined (Fenzo VM iteration order, ties between equal-fitness
// hosts, hash-map order of gpu-mode merges, priority-map order of equal (-dru,user) keys) this file
// DEFINES it and says so at the spot ("UNPINNED").
//
// Every function cites the reference file:line it follows (paths relative to reference scheduler/).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <queue>
#include <set>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../include/cookmatch.h"

namespace {

constexpr double DMAX = std::numeric_limits<double>::max();

// ---------------------------------------------------------------------------------------------------
// A.2 per-user order.  tools.clj:614-641 task->feature-vector / same-user-task-comparator:
//   [(- priority) start-time(or Date Long/MAX) (:db/id task)(nil for synthetic) (:db/id job)], Clojure
//   vector `compare` = lexicographic, nil sorts before any number.
struct FeatureKey {
  int64_t negprio, start, task, job;
};
inline FeatureKey feature_key(const cook_tasks* t, uint32_t i) {
  FeatureKey k;
  k.negprio = -(int64_t)t->priority[i];
  bool pend = t->pending[i] != 0;
  k.start = pend ? std::numeric_limits<int64_t>::max() : t->start_ms[i];
  k.task = pend ? std::numeric_limits<int64_t>::min() : t->task_id[i];  // nil < everything
  k.job = t->job_id[i];
  return k;
}
inline bool key_less(const FeatureKey& a, const FeatureKey& b) {
  if (a.negprio != b.negprio) return a.negprio < b.negprio;
  if (a.start != b.start) return a.start < b.start;
  if (a.task != b.task) return a.task < b.task;
  return a.job < b.job;
}

// tools.clj:876-881 below-quota?: every usage key <= quota key.
inline bool below_quota(const cook_usage& q, const cook_usage& u) {
  return u.count <= q.count && u.cpus <= q.cpus && u.mem <= q.mem && u.gpus <= q.gpus;
}
inline double gpus_of(const cook_tasks* t, uint32_t i) { return t->gpus ? t->gpus[i] : 0.0; }

struct Scored {
  uint32_t task;
  double dru;
};

// ---------------------------------------------------------------------------------------------------
// A.5 sorted-merge, LITERAL restatement of dru.clj:82-104: state = list of non-empty colls; each step
// stable-sorts the list by head key (Clojure sort-by = java.util.Arrays/sort with comparator = stable
// merge sort), emits the head of the first coll and conses that coll's remainder at the FRONT.
// O(N * U log U): used for small inputs and to validate merge_heap.
void merge_literal(const std::vector<std::vector<Scored>>& colls_in, std::vector<Scored>& out) {
  struct C {
    const std::vector<Scored>* v;
    size_t pos;
  };
  std::vector<C> colls;
  for (auto& c : colls_in)
    if (!c.empty()) colls.push_back({&c, 0});
  while (!colls.empty()) {
    std::stable_sort(colls.begin(), colls.end(),
                     [](const C& a, const C& b) { return (*a.v)[a.pos].dru < (*b.v)[b.pos].dru; });
    C first = colls.front();
    out.push_back((*first.v)[first.pos]);
    colls.erase(colls.begin());
    if (first.pos + 1 < first.v->size()) colls.insert(colls.begin(), {first.v, first.pos + 1});
  }
}

// The same order in O(N log U).  Derived tie rule (DESIGN.md "sorted-merge tie rule"): among heads with equal
// key, the coll that emitted most recently comes first; colls that never emitted come last, in initial
// (user-name) order.  Validated against merge_literal by tests/test_oracle_merge.py.
void merge_heap(const std::vector<std::vector<Scored>>& colls_in, std::vector<Scored>& out) {
  struct H {
    double dru;
    int64_t recency;  // step of last emission; never emitted: -1 - initial index
    uint32_t coll;
  };
  auto worse = [](const H& a, const H& b) {  // priority_queue keeps the "largest": invert
    if (a.dru != b.dru) return a.dru > b.dru;
    return a.recency < b.recency;
  };
  std::priority_queue<H, std::vector<H>, decltype(worse)> pq(worse);
  std::vector<size_t> pos(colls_in.size(), 0);
  for (uint32_t c = 0; c < colls_in.size(); ++c)
    if (!colls_in[c].empty()) pq.push({colls_in[c][0].dru, -1 - (int64_t)c, c});
  int64_t step = 0;
  while (!pq.empty()) {
    H h = pq.top();
    pq.pop();
    out.push_back(colls_in[h.coll][pos[h.coll]]);
    if (++pos[h.coll] < colls_in[h.coll].size()) pq.push({colls_in[h.coll][pos[h.coll]].dru, step, h.coll});
    ++step;
  }
}

struct RankResult {
  std::vector<uint32_t> ranked;  // task indices of surviving pending jobs in rank order
  std::vector<double> dru;       // per input task, NaN if cut by the limiter
  std::vector<Scored> merged;    // full merged sequence (running + pending)
  cook_usage pool_usage;
};

// scheduler.clj:2118-2123 task-ents->usage over the pool's running tasks (sequential merge-with +).
cook_usage running_usage(const cook_tasks* t) {
  cook_usage u{0, 0, 0, 0};
  for (uint32_t i = 0; i < t->n; ++i)
    if (!t->pending[i]) {
      u.count += 1;
      u.cpus += t->cpus[i];
      u.mem += t->mem[i];
      u.gpus += gpus_of(t, i);
    }
  return u;
}

// tools.clj:917-933 filter-based-on-pool-quota via filter-sequential (tools.clj:654-668): the state adds EVERY
// job seen, kept or not; a job is kept iff the updated usage is below-quota?.
void pool_quota_filter(const cook_tasks* t, const cook_usage& quota, cook_usage usage, std::vector<uint32_t>& q) {
  std::vector<uint32_t> keep;
  for (uint32_t i : q) {
    usage.count += 1;
    usage.cpus += t->cpus[i];
    usage.mem += t->mem[i];
    usage.gpus += gpus_of(t, i);
    if (below_quota(quota, usage)) keep.push_back(i);
  }
  q.swap(keep);
}

void rank_impl(const cook_params* p, const cook_tasks* t, const cook_users* u, const cook_pool_quota* pq, bool literal,
               RankResult& r) {
  const uint32_t N = t->n, U = u->n;
  r.dru.assign(N, std::numeric_limits<double>::quiet_NaN());
  // scheduler.clj:2076,2084-2085: tasks = running ++ pending; group-by user; sort each by the comparator.
  std::vector<std::vector<uint32_t>> by_user(U);
  for (uint32_t i = 0; i < N; ++i) by_user[t->user[i]].push_back(i);
  std::vector<std::vector<Scored>> colls(U);
  for (uint32_t us = 0; us < U; ++us) {
    auto& v = by_user[us];
    if (v.empty()) continue;
    std::sort(v.begin(), v.end(),
              [&](uint32_t a, uint32_t b) { return key_less(feature_key(t, a), feature_key(t, b)); });
    // scheduler.clj:2057-2071 limit-over-quota-jobs: running prefix of job->usage; count prefixes that are
    // not below-quota?; keep tasks while that count <= max-over-quota-jobs.
    cook_usage q{u->quota_count[us], u->quota_cpus[us], u->quota_mem[us], u->quota_gpus[us]};
    cook_usage tot{0, 0, 0, 0};
    int64_t over = 0;
    size_t kept = 0;
    for (uint32_t i : v) {
      tot.count += 1;
      tot.cpus += t->cpus[i];
      tot.mem += t->mem[i];
      tot.gpus += gpus_of(t, i);
      if (!below_quota(q, tot)) ++over;
      if (over > p->max_over_quota_jobs) break;
      ++kept;
    }
    // dru.clj:50-66 (default) / :68-80 (gpu): inclusive prefix sums left to right, divide, max.
    double cs = 0, ms = 0, gs = 0;
    auto& c = colls[us];
    for (size_t k = 0; k < kept; ++k) {
      uint32_t i = v[k];
      double d;
      if (p->dru_mode == 1) {
        gs = (k == 0) ? gpus_of(t, i) : gs + gpus_of(t, i);
        d = gs / u->div_gpus[us];
      } else {
        cs = (k == 0) ? t->cpus[i] : cs + t->cpus[i];
        ms = (k == 0) ? t->mem[i] : ms + t->mem[i];
        d = std::max(ms / u->div_mem[us], cs / u->div_cpus[us]);
      }
      r.dru[i] = d;
      c.push_back({i, d});
    }
  }
  // dru.clj:122-126: (sort-by first) users — ids ARE name ranks — then sorted-merge on :dru.
  // UNPINNED: the gpu-mode merge (dru.clj:106-112) has no name sort (hash-map order); we use name order too.
  if (literal)
    merge_literal(colls, r.merged);
  else
    merge_heap(colls, r.merged);
  // scheduler.clj:2089-2090 keep only pending.
  std::vector<uint32_t> q;
  for (auto& s : r.merged)
    if (t->pending[s.task]) q.push_back(s.task);
  // scheduler.clj:2134-2157 filter-based-on-quota: pool quota, then quota-group quota on the survivors.
  r.pool_usage = (pq && pq->pool_usage_given) ? pq->pool_usage : running_usage(t);
  if (pq && pq->has_pool_quota) pool_quota_filter(t, pq->pool_quota, r.pool_usage, q);
  if (pq && pq->has_group_quota) pool_quota_filter(t, pq->group_quota, pq->group_usage, q);
  // scheduler.clj:2198-2229 filter-offensive-jobs.
  for (uint32_t i : q)
    if (!(t->mem[i] > p->offensive_max_mem_mb || t->cpus[i] > p->offensive_max_cpus)) r.ranked.push_back(i);
}

// ---------------------------------------------------------------------------------------------------
// A.7/A.8 placement.  Restates Fenzo 0.10.0 TaskScheduler.scheduleOnce as Cook drives it (scheduler.clj:617-687,
// 2301-2324) with BinPackingFitnessCalculators.cpuMemBinPacker (config.clj:108).  Fenzo's source is not in the
// reference tree: this is its published algorithm as recalled in SURVEY.md Appendix A.7, anchored on Cook's
// call sites and on the known answers in tests/golden/match_*.json.
// UNPINNED (defined here): VMs are visited in offer-array order in ONE bucket; best = first strictly greatest
// fitness; early exit at the first VM whose fitness > good-enough.
struct MatchState {
  std::vector<double> ac, am;              // resources assigned this call per offer
  std::vector<int32_t> acount;             // tasks assigned this call per offer
  std::vector<int64_t> aports;             // ports assigned this call per offer
  std::vector<double> ascalar;             // [M][COOK_MAX_SCALARS] named scalars assigned this call per offer
  std::vector<std::vector<uint32_t>> ghost, gattr;  // per group: hosts / attr values of same-cycle cotasks
};

inline uint32_t offer_attr(const cook_offers* o, uint32_t v, uint32_t key) {
  if (key == COOK_NONE_U32) return o->host[v] + 1;  // "HOSTNAME"
  if (key >= o->n_attr_keys || !o->attr) return 0;
  return o->attr[(size_t)v * o->n_attr_keys + key];
}

// (get model->count model 0) and (count model->count) over a host's k8s "gpus" / "disk" map (constraints.clj:136-142, 178):
// the map is the row's non-empty slots; a job without a model asks for key nil, which no map holds.
inline double map_get(const uint32_t* keys, const double* vals, uint32_t slots, uint32_t v, uint32_t key) {
  if (!keys || !vals || key == 0) return 0.0;
  const uint32_t S = slots ? slots : 1;
  for (uint32_t s2 = 0; s2 < S; ++s2)
    if (keys[(size_t)v * S + s2] == key) return vals[(size_t)v * S + s2];
  return 0.0;
}
inline uint32_t map_count(const uint32_t* keys, uint32_t slots, uint32_t v) {
  if (!keys) return 0;
  const uint32_t S = slots ? slots : 1;
  uint32_t n = 0;
  for (uint32_t s2 = 0; s2 < S; ++s2) n += keys[(size_t)v * S + s2] != 0 ? 1 : 0;
  return n;
}

// ⚠ Fenzo AssignableVirtualMachine.tryRequest beyond cpus / mem: ports (PortRanges.hasPorts: the request's port COUNT against
// the free ports of the lease's ranges; TaskRequestAdapter getPorts, scheduler.clj:466; lease portRanges, offer.clj:71-73) and
// every named scalar request (getScalarRequests = job->scalar-request, scheduler.clj:177-189) as used + request > total against
// the lease's getScalarValues (offer.clj:57-65).  Returns bit 0 = ports do not fit, bit 1 + s = named scalar s does not fit.
inline uint32_t xres_fail_bits(const cook_jobs* j, uint32_t k, const cook_offers* o, uint32_t v, const MatchState& st) {
  uint32_t bits = 0;
  const int64_t jp = j->ports ? j->ports[k] : 0;
  if (jp > 0 && st.aports[v] + jp > (int64_t)(o->ports ? o->ports[v] : 0)) bits |= 1u;
  for (uint32_t s2 = 0; j->scalars && s2 < j->n_scalars && s2 < COOK_MAX_SCALARS; ++s2) {
    const double r = j->scalars[(size_t)s2 * j->n + k];
    if (r != r) continue;  // the job has no request under this name
    const double t = (o->scalars && s2 < o->n_scalars) ? o->scalars[(size_t)s2 * o->n + v] : 0.0;
    if (st.ascalar[(size_t)v * COOK_MAX_SCALARS + s2] + r > t) bits |= 2u << s2;
  }
  return bits;
}
inline void xres_commit(const cook_jobs* j, uint32_t k, uint32_t v, MatchState& st) {
  st.aports[v] += (j->ports && j->ports[k] > 0) ? j->ports[k] : 0;  // (the ABI rejects negative counts; only positive ones are requests)
  for (uint32_t s2 = 0; j->scalars && s2 < j->n_scalars && s2 < COOK_MAX_SCALARS; ++s2) {
    const double r = j->scalars[(size_t)s2 * j->n + k];
    if (r == r) st.ascalar[(size_t)v * COOK_MAX_SCALARS + s2] += r;
  }
}

// constraints.clj: job constraints in make-fenzo-job-constraints order do not matter for pass/fail.
bool job_constraints_pass(const cook_params* p, const cook_jobs* j, uint32_t k, const cook_offers* o, uint32_t v,
                          const MatchState& st, const std::set<uint32_t>& reserved) {
  const uint32_t host = o->host[v];
  // novel-host (constraints.clj:68-94)
  if (j->novel_off)
    for (uint32_t x = j->novel_off[k]; x < j->novel_off[k + 1]; ++x)
      if (j->novel_host[x] == host) return false;
  // gpu-host (constraints.clj:122-157)
  const double jg = j->gpus ? j->gpus[k] : 0.0;
  const bool k8s = o->k8s && o->k8s[v];
  if (k8s) {
    if (jg > 0) {
      const double avail = map_get(o->gpu_model, o->gpu_count, o->gpu_slots, v, j->gpu_model ? j->gpu_model[k] : 0);
      const int32_t on_vm = (o->run_count ? o->run_count[v] : 0) + st.acount[v];
      if (!(avail == jg && on_vm == 0)) return false;
    } else if (map_count(o->gpu_model, o->gpu_slots, v) != 0) {
      return false;
    }
  } else if (!(jg == 0)) {
    return false;
  }
  // disk-host (constraints.clj:164-199): only when the pool enables it (disk_request >= 0)
  if (j->disk_request && j->disk_request[k] >= 0 && k8s) {
    const double space = map_get(o->disk_type, o->disk_space, o->disk_slots, v, j->disk_type[k]);
    if (!(space >= j->disk_request[k])) return false;
  }
  // user-defined EQUALS (constraints.clj:356-377)
  if (j->eq_off)
    for (uint32_t x = j->eq_off[k]; x < j->eq_off[k + 1]; ++x)
      if (offer_attr(o, v, j->eq_key[x]) != j->eq_val[x]) return false;
  // estimated-completion (constraints.clj:385-401)
  if (j->est_end_ms && j->est_end_ms[k] != 0 && o->host_start_s && o->host_start_s[v] >= 0) {
    const int64_t death = 1000 * o->host_start_s[v] + 60 * 1000 * p->host_lifetime_mins;
    if (!(j->est_end_ms[k] < death)) return false;
  }
  // checkpoint-locality (constraints.clj:218-240)
  if (j->ckpt_location && j->ckpt_location[k] != 0) {
    const uint32_t loc = o->location ? o->location[v] : 0;
    if (loc != j->ckpt_location[k]) return false;
  }
  // max-tasks-per-host (constraints.clj:433-456)
  if (o->max_tasks && o->max_tasks[v] >= 0) {
    if (!((o->num_tasks ? o->num_tasks[v] : 0) + st.acount[v] < o->max_tasks[v])) return false;
  }
  // rebalancer-reservation (constraints.clj:242-252, scheduler.clj:645-653): hosts reserved for OTHER jobs
  if (!reserved.empty() && reserved.count(host)) {
    if (!(j->reserved_host && j->reserved_host[k] == (int32_t)host)) return false;
  }
  return true;
}

// constraints.clj:586-644 group constraints over cotasks = DB-running ++ assigned earlier in this call.
bool group_constraint_pass(const cook_jobs* j, uint32_t k, const cook_offers* o, uint32_t v, const cook_groups* g,
                           const MatchState& st) {
  if (!g || !j->group || j->group[k] == COOK_NONE_U32) return true;
  const uint32_t gi = j->group[k];
  const uint8_t type = g->type[gi];
  if (type == 0) return true;
  const uint32_t r0 = g->run_off ? g->run_off[gi] : 0, r1 = g->run_off ? g->run_off[gi + 1] : 0;
  if (type == 1) {  // unique (constraints.clj:586-598): hostname present and not used by a cotask
    const uint32_t host = o->host[v];
    for (uint32_t x = r0; x < r1; ++x)
      if (g->run_host[x] == host) return false;
    for (uint32_t h : st.ghost[gi])
      if (h == host) return false;
    return true;
  }
  const uint32_t key = g->attr_key[gi];
  const uint32_t target = offer_attr(o, v, key);
  std::map<uint32_t, int> freq;  // nil (0) is a legal key of `frequencies`
  for (uint32_t x = r0; x < r1; ++x) freq[key == COOK_NONE_U32 ? g->run_host[x] + 1 : g->run_attr[x]]++;
  for (uint32_t a : st.gattr[gi]) freq[a]++;
  if (freq.empty()) return true;
  if (type == 2) {  // balanced (constraints.clj:600-626)
    int mn = std::numeric_limits<int>::max(), mx = 0;
    for (auto& kv : freq) {
      mn = std::min(mn, kv.second);
      mx = std::max(mx, kv.second);
    }
    const int minim = (g->minimum[gi] > (int)freq.size()) ? 0 : mn;
    auto it = freq.find(target);
    if (it == freq.end()) return true;
    return minim == mx || it->second < mx;
  }
  // attribute-equals (constraints.clj:628-644)
  return freq.count(target) != 0;
}

// The hard constraints in the order Fenzo walks them: make-task-request builds (into (list) constraints) from
// (conj (make-fenzo-job-constraints job) reservation) ++ group constraints (scheduler.clj:493-501); conj on a seq prepends
// and (into (list) ...) reverses, so the list Fenzo sees is
//   checkpoint-locality, estimated-completion, user-defined, disk-host, gpu-host, novel-host   (constraints.clj:459-464 reversed)
//   max_tasks_per_host (constraints.clj:433-456), rebalancer-reservation (:242-252), then the job's group constraint.
// Returns the COOK_WHY_* slot of the FIRST failing constraint (the one Fenzo's ConstraintFailure names), -1 if all pass.
int first_failed_constraint(const cook_params* p, const cook_jobs* j, uint32_t k, const cook_offers* o, uint32_t v,
                            const cook_groups* g, const MatchState& st, const std::set<uint32_t>& reserved) {
  const uint32_t host = o->host[v];
  const bool k8s = o->k8s && o->k8s[v];
  if (j->ckpt_location && j->ckpt_location[k] != 0) {  // checkpoint_locality_constraint
    const uint32_t loc = o->location ? o->location[v] : 0;
    if (loc != j->ckpt_location[k]) return 3;
  }
  if (j->est_end_ms && j->est_end_ms[k] != 0 && o->host_start_s && o->host_start_s[v] >= 0) {  // estimated_completion_constraint
    const int64_t death = 1000 * o->host_start_s[v] + 60 * 1000 * p->host_lifetime_mins;
    if (!(j->est_end_ms[k] < death)) return 4;
  }
  if (j->eq_off)  // user_defined_constraint
    for (uint32_t x = j->eq_off[k]; x < j->eq_off[k + 1]; ++x)
      if (offer_attr(o, v, j->eq_key[x]) != j->eq_val[x]) return 5;
  if (j->disk_request && j->disk_request[k] >= 0 && k8s) {  // disk_host_constraint
    const double space = map_get(o->disk_type, o->disk_space, o->disk_slots, v, j->disk_type[k]);
    if (!(space >= j->disk_request[k])) return 6;
  }
  {  // gpu_host_constraint
    const double jg = j->gpus ? j->gpus[k] : 0.0;
    if (k8s) {
      if (jg > 0) {
        const double avail = map_get(o->gpu_model, o->gpu_count, o->gpu_slots, v, j->gpu_model ? j->gpu_model[k] : 0);
        const int32_t on_vm = (o->run_count ? o->run_count[v] : 0) + st.acount[v];
        if (!(avail == jg && on_vm == 0)) return 7;
      } else if (map_count(o->gpu_model, o->gpu_slots, v) != 0) {
        return 7;
      }
    } else if (!(jg == 0)) {
      return 7;
    }
  }
  if (j->novel_off)  // novel_host_constraint
    for (uint32_t x = j->novel_off[k]; x < j->novel_off[k + 1]; ++x)
      if (j->novel_host[x] == host) return 8;
  if (o->max_tasks && o->max_tasks[v] >= 0)  // max_tasks_per_host
    if (!((o->num_tasks ? o->num_tasks[v] : 0) + st.acount[v] < o->max_tasks[v])) return 9;
  if (!reserved.empty() && reserved.count(host))  // rebalancer_reservation_constraint
    if (!(j->reserved_host && j->reserved_host[k] == (int32_t)host)) return 10;
  if (!group_constraint_pass(j, k, o, v, g, st)) return 10 + g->type[j->group[k]];  // 11 unique, 12 balanced, 13 attribute-equals
  return -1;
}

// deadline_s > 0 (bench.py's cpu_baseline leg only): give up — return false, outputs incomplete — once the call has run that long
bool match_impl(const cook_params* p, const cook_jobs* j, const cook_offers* o, const cook_groups* g,
                const uint32_t* reserved_hosts, uint32_t n_reserved, int32_t* job_to_offer, uint32_t* fail_code,
                uint8_t* head_matched, int nthreads, const uint32_t* explain_pos = nullptr, uint32_t n_explain = 0,
                uint32_t* explain_counts = nullptr, double deadline_s = 0.0) {
  const auto t_begin = std::chrono::steady_clock::now();
  bool finished = true;
  const uint32_t K = j->n, M = o->n;
  MatchState st;
  std::map<uint32_t, std::vector<uint32_t>> explain_rows;  // job position -> rows of explain_counts
  for (uint32_t q = 0; q < n_explain; ++q) {
    explain_rows[explain_pos[q]].push_back(q);
    for (int s2 = 0; s2 < COOK_WHY_SLOTS; ++s2) explain_counts[(size_t)q * COOK_WHY_SLOTS + s2] = 0;
  }
  st.aports.assign(M, 0);
  st.ascalar.assign((size_t)M * COOK_MAX_SCALARS, 0.0);
  st.ac.assign(M, 0.0);
  st.am.assign(M, 0.0);
  st.acount.assign(M, 0);
  if (g) {
    st.ghost.resize(g->n);
    st.gattr.resize(g->n);
  }
  std::set<uint32_t> reserved(reserved_hosts, reserved_hosts + n_reserved);
  const double ge = p->good_enough_fitness;
  uint32_t matched = 0;
  struct Best {
    double fit;
    int32_t v;
    uint32_t fail;
  };
  auto eval_range = [&](uint32_t k, uint32_t v0, uint32_t v1) -> Best {
    Best b;  // a local of the evaluating thread: published once per job, never updated in place (no shared cache line in the loop)
    const double c = j->cpus[k], m = j->mem[k];
    b.fit = -1.0;
    b.v = -1;
    b.fail = 0;
    for (uint32_t v = v0; v < v1; ++v) {
      // ⚠ Fenzo AssignableVirtualMachine.tryRequest: resources (cpus, mem, named scalars) vs the lease totals
      // minus what this call already assigned to the VM, then hard constraints, then fitness.
      if (st.ac[v] + c > o->cpus[v] || st.am[v] + m > o->mem[v] || xres_fail_bits(j, k, o, v, st) != 0) {
        b.fail |= 1;
        continue;
      }
      if (!job_constraints_pass(p, j, k, o, v, st, reserved) || !group_constraint_pass(j, k, o, v, g, st)) {
        b.fail |= 2;
        continue;
      }
      // ⚠ cpuMemBinPacker: (cpuFit + memFit)/2 with fit = (running + assigned-this-call + request) /
      // (lease total + running) per resource.
      const double rc = o->run_cpus ? o->run_cpus[v] : 0.0, rm = o->run_mem ? o->run_mem[v] : 0.0;
      const double fit = ((rc + st.ac[v] + c) / (o->cpus[v] + rc) + (rm + st.am[v] + m) / (o->mem[v] + rm)) / 2.0;
      if (!(fit > 0.0)) {  // ⚠ fitness 0.0 is a failure in Fenzo
        b.fail |= 4;
        continue;
      }
      if (fit > b.fit) {
        b.fit = fit;
        b.v = (int32_t)v;
        if (fit > ge) break;  // scheduler.clj:2312-2314
      }
    }
    return b;
  };
  // multi-thread CPU baseline: hosts bucketed across PERSISTENT worker threads per job (mirrors Fenzo's evaluator
  // pool, built at scheduler.clj:2301-2324); identical result to the single bucket when good-enough is disabled (argmax, lowest
  // index on ties).  Every worker evaluates its bucket into a local and publishes it ONCE into its own 128-byte slot; the
  // generation / completion counters sit on cache lines of their own.
  const bool mt = nthreads > 1 && !(ge < 1.0) && M >= 1024;
  const int T = mt ? nthreads : 1;
  struct alignas(128) Slot {
    Best b;
  };
  struct alignas(128) Counter {
    std::atomic<uint32_t> v{0};
  };
  std::vector<Slot> parts(T);
  Counter gen, done;
  std::atomic<bool> quit{false};
  uint32_t cur_k = 0;
  const uint32_t chunk = (M + T - 1) / T;
  std::vector<std::thread> workers;
  for (int tix = 1; tix < T; ++tix)
    workers.emplace_back([&, tix] {
      uint32_t seen = 0;
      for (;;) {
        for (unsigned spins = 0; gen.v.load(std::memory_order_acquire) == seen; ++spins) {
          if (quit.load(std::memory_order_relaxed)) return;
          if (spins < 4096u) __builtin_ia32_pause();
          else std::this_thread::yield();  // (more threads than the process may run at once: do not spin the others out of their turn)
        }
        ++seen;
        parts[tix].b = eval_range(cur_k, std::min(M, tix * chunk), std::min(M, (tix + 1) * chunk));
        done.v.fetch_add(1, std::memory_order_release);
      }
    });
  for (uint32_t k = 0; k < K; ++k) {
    if (deadline_s > 0.0 && (k & 255u) == 0u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > deadline_s) {
      finished = false;
      break;
    }
    // fenzo-utils/summarize-placement-failure (fenzo_utils.clj:33-55) over the TaskAssignmentResults of job k: per host the
    // resources that do not fit (message "cpus" / "mem", one count each) or else the first failing hard constraint's name
    auto ex = explain_rows.find(k);
    if (ex != explain_rows.end()) {
      uint32_t cnt[COOK_WHY_SLOTS] = {0};
      const double c = j->cpus[k], m = j->mem[k];
      for (uint32_t v = 0; v < M; ++v) {
        const bool fc = st.ac[v] + c > o->cpus[v], fm = st.am[v] + m > o->mem[v];
        const uint32_t fx = xres_fail_bits(j, k, o, v, st);
        if (fc || fm || fx) {  // ⚠ every resource that does not fit adds its own AssignmentFailure
          cnt[COOK_WHY_CPUS] += fc ? 1 : 0;
          cnt[COOK_WHY_MEM] += fm ? 1 : 0;
          cnt[COOK_WHY_PORTS] += (fx & 1u) ? 1 : 0;
          for (int s2 = 0; s2 < COOK_MAX_SCALARS; ++s2) cnt[COOK_WHY_SCALAR0 + s2] += ((fx >> (1 + s2)) & 1u) ? 1 : 0;
          continue;
        }
        const int why = first_failed_constraint(p, j, k, o, v, g, st, reserved);
        const bool pass = job_constraints_pass(p, j, k, o, v, st, reserved) && group_constraint_pass(j, k, o, v, g, st);
        if ((why < 0) != pass) std::abort();  // the ordered walk and the unordered check must agree
        if (why >= 0) {
          cnt[why] += 1;
          continue;
        }
        const double rc = o->run_cpus ? o->run_cpus[v] : 0.0, rm = o->run_mem ? o->run_mem[v] : 0.0;
        const double fit = ((rc + st.ac[v] + c) / (o->cpus[v] + rc) + (rm + st.am[v] + m) / (o->mem[v] + rm)) / 2.0;
        if (!(fit > 0.0)) cnt[2] += 1;
      }
      for (uint32_t q : ex->second)
        for (int s2 = 0; s2 < COOK_WHY_SLOTS; ++s2) explain_counts[(size_t)q * COOK_WHY_SLOTS + s2] = cnt[s2];
    }
    Best b;
    if (!mt) {
      b = eval_range(k, 0, M);
    } else {
      cur_k = k;
      done.v.store(0, std::memory_order_relaxed);
      gen.v.fetch_add(1, std::memory_order_release);
      b = eval_range(k, 0, std::min(M, chunk));
      for (unsigned spins = 0; done.v.load(std::memory_order_acquire) != (uint32_t)(T - 1); ++spins) {
        if (spins < 4096u) __builtin_ia32_pause();
        else std::this_thread::yield();
      }
      for (int tix = 1; tix < T; ++tix) {
        const Best& q = parts[tix].b;
        b.fail |= q.fail;
        if (q.v >= 0 && q.fit > b.fit) {
          b.fit = q.fit;
          b.v = q.v;
        }
      }
    }
    job_to_offer[k] = b.v;
    if (fail_code) fail_code[k] = b.v >= 0 ? 0u : (b.fail ? b.fail : 8u);
    if (b.v >= 0) {
      ++matched;
      st.ac[b.v] += j->cpus[k];
      st.am[b.v] += j->mem[k];
      st.acount[b.v] += 1;
      xres_commit(j, k, (uint32_t)b.v, st);
      if (g && j->group && j->group[k] != COOK_NONE_U32) {
        const uint32_t gi = j->group[k];
        st.ghost[gi].push_back(o->host[b.v]);
        st.gattr[gi].push_back(g->type[gi] >= 2 ? offer_attr(o, b.v, g->attr_key[gi]) : 0);
      }
    }
  }
  quit.store(true);
  for (auto& w : workers) w.join();
  // scheduler.clj:1495: matched-head-or-no-matches?
  if (head_matched) *head_matched = (matched == 0 || (K > 0 && job_to_offer[0] >= 0)) ? 1 : 0;
  return finished;
}

// ---------------------------------------------------------------------------------------------------
}  // namespace

extern "C" {

const char* oracle_version(void) { return "cook-oracle 1 (CPU restatement; pinned to reference unit-test vectors)"; }

// merge_mode: 0 = O(N log U) heap form, 1 = literal dru.clj:82-104 restatement.
int oracle_rank(const cook_params* p, const cook_tasks* t, const cook_users* u, const cook_pool_quota* pq,
                uint32_t* ranked_pending_idx, uint32_t* n_out, double* dru_of_task, int merge_mode) {
  RankResult r;
  rank_impl(p, t, u, pq, merge_mode == 1, r);
  if (ranked_pending_idx) std::copy(r.ranked.begin(), r.ranked.end(), ranked_pending_idx);
  if (n_out) *n_out = (uint32_t)r.ranked.size();
  if (dru_of_task) std::copy(r.dru.begin(), r.dru.end(), dru_of_task);
  return 0;
}

// Full merged order (running + pending) — used by tests that pin the DRU sequence itself (dru.clj tests).
int oracle_rank_merged(const cook_params* p, const cook_tasks* t, const cook_users* u, uint32_t* merged_idx,
                       double* merged_dru, uint32_t* n_out, int merge_mode) {
  RankResult r;
  rank_impl(p, t, u, nullptr, merge_mode == 1, r);
  for (size_t i = 0; i < r.merged.size(); ++i) {
    merged_idx[i] = r.merged[i].task;
    merged_dru[i] = r.merged[i].dru;
  }
  *n_out = (uint32_t)r.merged.size();
  return 0;
}

// scheduler.clj:729-762 pending-jobs->considerable-jobs; tools.clj:903-973 filter-pending-jobs-for-quota;
// tools.clj:654-668 filter-sequential (the state advances on rejected elements too).
// The reference pipeline is lazy: stages run interleaved and stop once `take` is satisfied.  The surviving jobs are the
// same either way; the per-user rate-limit counters are computed here over the WHOLE queue (UNPINNED: the lazy original
// counts only the jobs it consumed).
int oracle_considerable(const cook_queue* q, const cook_user_state* us, uint32_t num_considerable, uint32_t* out_idx,
                        uint32_t* n_out, uint32_t* rate_limited, uint32_t* passed) {
  const uint32_t n = q->n, U = us->n;
  std::vector<cook_usage> usage(U);
  for (uint32_t u = 0; u < U; ++u) usage[u] = cook_usage{us->usage_count[u], us->usage_cpus[u], us->usage_mem[u], us->usage_gpus[u]};
  // tools.clj:966: pool-usage = (reduce (partial merge-with +) (vals user->usage)); map order is UNPINNED -> user-id order
  cook_usage pool = us->pool_usage;
  if (!us->pool_usage_given) {
    pool = cook_usage{0, 0, 0, 0};
    for (uint32_t u = 0; u < U; ++u) {
      if (u == 0) {
        pool = usage[0];
      } else {
        pool.count += usage[u].count;
        pool.cpus += usage[u].cpus;
        pool.mem += usage[u].mem;
        pool.gpus += usage[u].gpus;
      }
    }
  }
  std::vector<uint32_t> seen(U, 0);
  if (rate_limited) std::fill(rate_limited, rate_limited + U, 0u);
  if (passed) std::fill(passed, passed + U, 0u);
  uint32_t k = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t u = q->user[i];
    const cook_usage ju{1.0, q->cpus[i], q->mem[i], q->gpus ? q->gpus[i] : 0.0};
    // tools.clj:903-915 filter-based-on-user-quota: usage' = (merge-with + job-usage usage[user])
    cook_usage& uu = usage[u];
    uu = cook_usage{ju.count + uu.count, ju.cpus + uu.cpus, ju.mem + uu.mem, ju.gpus + uu.gpus};
    const cook_usage quota{us->quota_count[u], us->quota_cpus[u], us->quota_mem[u], us->quota_gpus[u]};
    if (!below_quota(quota, uu)) continue;
    // tools.clj:935-955 filter-pending-jobs-for-ratelimit
    const uint32_t so_far = ++seen[u];
    const bool limited = us->tokens_left ? ((int64_t)so_far > us->tokens_left[u]) : false;
    if (limited) {
      if (rate_limited) rate_limited[u]++;
    } else if (passed) {
      passed[u]++;
    }
    if (limited && us->enforce_rate_limit) continue;
    // tools.clj:917-933 filter-based-on-pool-quota
    if (us->has_pool_quota) {
      pool = cook_usage{ju.count + pool.count, ju.cpus + pool.cpus, ju.mem + pool.mem, ju.gpus + pool.gpus};
      if (!below_quota(us->pool_quota, pool)) continue;
    }
    // scheduler.clj:747-749: job-allowed-to-start?, launch plugin, take
    if (q->eligible && !q->eligible[i]) continue;
    if (k < num_considerable) out_idx[k++] = i;
  }
  *n_out = k;
  return 0;
}

int oracle_pool_usage(const cook_tasks* t, cook_usage* out) {
  *out = running_usage(t);
  return 0;
}

// Bare sorted-merge over explicit per-coll key lists: coll c holds keys[off[c] .. off[c+1]).  out_coll[i] is
// the coll the i-th emitted item came from.
int oracle_sorted_merge(uint32_t n_colls, const uint32_t* off, const double* keys, int literal, uint32_t* out_coll) {
  std::vector<std::vector<Scored>> colls(n_colls);
  for (uint32_t c = 0; c < n_colls; ++c)
    for (uint32_t i = off[c]; i < off[c + 1]; ++i) colls[c].push_back({c, keys[i]});
  std::vector<Scored> out;
  if (literal)
    merge_literal(colls, out);
  else
    merge_heap(colls, out);
  for (size_t i = 0; i < out.size(); ++i) out_coll[i] = out[i].task;
  return 0;
}

int oracle_match(const cook_params* p, const cook_jobs* j, const cook_offers* o, const cook_groups* g,
                 const uint32_t* reserved_hosts, uint32_t n_reserved, int32_t* job_to_offer, uint32_t* fail_code,
                 uint8_t* head_matched, int nthreads) {
  match_impl(p, j, o, g, reserved_hosts, n_reserved, job_to_offer, fail_code, head_matched, nthreads);
  return 0;
}

// One pool's whole match cycle in ONE call (bench.py's cpu_baseline leg: no interpreter between the phases, so pool threads never
// meet on Python's lock): rank (scheduler.clj:2073-2091 ...) -> the first K ranked pending jobs gathered into considerable order
// (scheduler.clj:729-762's take) -> placement (scheduler.clj:617-687).  `pending_jobs` holds the pool's pending jobs in the order
// their tasks appear among t's pending rows (the q-th pending task is job q).  phase_s = {rank, gather, match} seconds.
int oracle_cycle(const cook_params* p, const cook_tasks* t, const cook_users* u, const cook_pool_quota* pq,
                 const cook_jobs* pending_jobs, const cook_offers* o, const cook_groups* g, uint32_t K, int nthreads,
                 uint32_t* ranked_pending_idx, uint32_t* n_ranked, int32_t* job_to_offer, uint32_t* n_considerable,
                 double* phase_s, double deadline_s) {
  using clk = std::chrono::steady_clock;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const auto t0 = clk::now();
  RankResult r;
  rank_impl(p, t, u, pq, false, r);
  const auto t1 = clk::now();
  const uint32_t kk = std::min<uint32_t>(K, (uint32_t)r.ranked.size());
  std::vector<uint32_t> pend_ord(t->n);
  uint32_t q = 0;
  for (uint32_t i = 0; i < t->n; ++i) pend_ord[i] = t->pending[i] ? q++ : 0u;
  std::vector<uint32_t> idx(kk);
  for (uint32_t i = 0; i < kk; ++i) idx[i] = pend_ord[r.ranked[i]];
  const cook_jobs* pj = pending_jobs;
  cook_jobs c = *pj;
  c.n = kk;
  auto take = [&](auto* src, auto& dst) {
    using T = std::remove_cv_t<std::remove_pointer_t<decltype(src)>>;
    if (!src) return (const T*)nullptr;
    dst.resize(std::max<uint32_t>(1, kk));
    for (uint32_t i = 0; i < kk; ++i) dst[i] = src[idx[i]];
    return (const T*)dst.data();
  };
  std::vector<double> v_cpus, v_mem, v_gpus, v_disk, v_scal;
  std::vector<uint32_t> v_model, v_user, v_group, v_ckpt, v_dtype, v_eqoff, v_eqk, v_eqv, v_nvoff, v_nvh;
  std::vector<int32_t> v_res, v_ports;
  std::vector<int64_t> v_est;
  c.cpus = take(pj->cpus, v_cpus);
  c.mem = take(pj->mem, v_mem);
  c.gpus = take(pj->gpus, v_gpus);
  c.gpu_model = take(pj->gpu_model, v_model);
  c.user = take(pj->user, v_user);
  c.group = take(pj->group, v_group);
  c.reserved_host = take(pj->reserved_host, v_res);
  c.ckpt_location = take(pj->ckpt_location, v_ckpt);
  c.est_end_ms = take(pj->est_end_ms, v_est);
  c.disk_request = take(pj->disk_request, v_disk);
  c.disk_type = take(pj->disk_type, v_dtype);
  c.ports = take(pj->ports, v_ports);
  if (pj->scalars && pj->n_scalars) {
    v_scal.resize((size_t)std::max<uint32_t>(1, kk) * pj->n_scalars);
    for (uint32_t s2 = 0; s2 < pj->n_scalars; ++s2)
      for (uint32_t i = 0; i < kk; ++i) v_scal[(size_t)s2 * kk + i] = pj->scalars[(size_t)s2 * pj->n + idx[i]];
    c.scalars = v_scal.data();
  }
  auto take_csr = [&](const uint32_t* off, std::vector<uint32_t>& noff, std::initializer_list<std::pair<const uint32_t*, std::vector<uint32_t>*>> cols) {
    noff.assign(kk + 1, 0);
    for (uint32_t i = 0; i < kk; ++i) noff[i + 1] = noff[i] + (off[idx[i] + 1] - off[idx[i]]);
    for (auto& cv : cols) {
      cv.second->resize(std::max<uint32_t>(1, noff[kk]));
      for (uint32_t i = 0; i < kk; ++i)
        std::copy(cv.first + off[idx[i]], cv.first + off[idx[i] + 1], cv.second->begin() + noff[i]);
    }
  };
  if (pj->eq_off) {
    take_csr(pj->eq_off, v_eqoff, {{pj->eq_key, &v_eqk}, {pj->eq_val, &v_eqv}});
    c.eq_off = v_eqoff.data();
    c.eq_key = v_eqk.data();
    c.eq_val = v_eqv.data();
  }
  if (pj->novel_off) {
    take_csr(pj->novel_off, v_nvoff, {{pj->novel_host, &v_nvh}});
    c.novel_off = v_nvoff.data();
    c.novel_host = v_nvh.data();
  }
  const auto t2 = clk::now();
  const bool finished = match_impl(p, &c, o, g, nullptr, 0, job_to_offer, nullptr, nullptr, nthreads, nullptr, 0, nullptr, deadline_s);
  const auto t3 = clk::now();
  if (ranked_pending_idx) std::copy(r.ranked.begin(), r.ranked.end(), ranked_pending_idx);
  if (n_ranked) *n_ranked = (uint32_t)r.ranked.size();
  if (n_considerable) *n_considerable = kk;
  if (phase_s) {
    phase_s[0] = secs(t0, t1);
    phase_s[1] = secs(t1, t2);
    phase_s[2] = secs(t2, t3);
  }
  return finished ? 0 : 1;  // 1: gave up at the deadline (the outputs are incomplete)
}

// the same placement, plus for each job position of explain_pos the 16 COOK_WHY_* counts of cook_match_explain
int oracle_match_explain(const cook_params* p, const cook_jobs* j, const cook_offers* o, const cook_groups* g,
                         const uint32_t* reserved_hosts, uint32_t n_reserved, int32_t* job_to_offer, const uint32_t* explain_pos,
                         uint32_t n_explain, uint32_t* explain_counts) {
  match_impl(p, j, o, g, reserved_hosts, n_reserved, job_to_offer, nullptr, nullptr, 1, explain_pos, n_explain, explain_counts);
  return 0;
}

// rebalancer.clj:222-266 init-state, :320-407 compute-preemption-decision, :270-309 next-state, :434-467 rebalance.
//
// host_attrs (optional): the agent-attributes-cache (scheduler.clj:1586-1597) as a cook_offers table, one row per host
//   whose attribute map is cached (`host[i]` = host id; resource columns are ignored).  A host's attribute map is nil when
//   it is not cached OR when its hostname cannot be resolved to a slave id: the reference resolves hostname -> slave-id
//   through the scored tasks (rebalancer.clj:369-375, `into {}` = the LAST task of the host in priority-map order wins);
//   real instances always carry a slave id, a task placed by an earlier decision of this cycle carries the slave id of
//   that decision's first preempted task (rebalancer.clj:279-281) — nil for a spare-resources-only decision.
// groups (optional): pending->group[p] indexes it; run_host = hosts of the group's running cotasks per the DB, minus the
//   job's own instances (constraints.clj:531-537); their attributes are looked up in host_attrs (run_attr is ignored).
// The rebalancer evaluates job-constraint-constructors only (constraints.clj:459-466: novel-host, gpu-host, disk-host,
// user-defined, estimated-completion, checkpoint-locality) through the 3-arity evaluate, i.e. with NO tasks assigned,
// and group constraints with cohosts = hosts of EVERY task preempted so far this cycle ++ cotask hosts
// (constraints.clj:680-697).
// forced_* (test hook, NULL in normal use): apply the given decision for pending job p instead of computing one
//   (pins next-state, rebalancer.clj:270-309, against the reference's test-next-state vectors).
// final_* (optional): the scored tasks after the loop in priority-map order (index >= running->n: R + pending index).
// UNPINNED: order among scored tasks with equal (-dru, user) (priority-map value sets are hash sets): we use the
// user's task order.  gpu-mode pools: compute-preemption-decision reads (:dru score) of a bare number and throws in the
// reference (rebalancer.clj:252-256, 346); only the pending-job DRU (rebalancer.clj:157-180) is pinned for that mode.
// test hooks of oracle_rebalance (all optional)
struct oracle_rebal_hooks {
  const uint8_t* running_slave_known;  // 0: the instance's slave id has no cached attribute map (a test artefact: random slave ids)
  const uint32_t* init_preempted_hosts;  // State :preempted-tasks at entry (hosts of tasks preempted earlier this cycle)
  uint32_t n_init_preempted;
  const int32_t* forced_host;  // per pending job: -2 compute, -1 forced "no decision", >= 0 forced decision on that host
  const uint32_t* forced_off;  // CSR of forced task ids (index into running, or R + pending index)
  const uint32_t* forced_task;
  const cook_usage* forced_res;  // :cpus :mem :gpus of the forced decision
  uint32_t* final_order;  // scored tasks after the loop in priority-map order
  double* final_dru;
  uint32_t* n_final;
};

struct RebalAttrs {
  const cook_offers* t;
  std::unordered_map<uint32_t, uint32_t> row_of_host;
  explicit RebalAttrs(const cook_offers* t_) : t(t_) {
    if (t)
      for (uint32_t i = 0; i < t->n; ++i) row_of_host[t->host[i]] = i;
  }
  int row(uint32_t host) const {
    auto it = row_of_host.find(host);
    return it == row_of_host.end() ? -1 : (int)it->second;
  }
  // value id of attribute `key` in the map of row r (r < 0: nil map); COOK_NONE_U32 = "HOSTNAME"
  uint32_t attr(int r, uint32_t key) const {
    if (r < 0) return 0;
    if (key == COOK_NONE_U32) return t->host[r] + 1;
    if (!t->attr || key >= t->n_attr_keys) return 0;
    return t->attr[(size_t)r * t->n_attr_keys + key];
  }
};

// job constraints of the rebalancer on the attribute map of row r (r < 0: nil map)
static bool rebal_job_constraints_pass(const cook_params* p, const cook_jobs* j, uint32_t k, const RebalAttrs& A, int r) {
  const cook_offers* o = A.t;
  // novel-host (constraints.clj:68-94): (get nil "HOSTNAME") is nil, never in the set
  if (r >= 0 && j->novel_off)
    for (uint32_t x = j->novel_off[k]; x < j->novel_off[k + 1]; ++x)
      if (j->novel_host[x] == o->host[r]) return false;
  // gpu-host (constraints.clj:122-157) with vm-tasks-assigned = []
  const double jg = j->gpus ? j->gpus[k] : 0.0;
  const bool k8s = r >= 0 && o->k8s && o->k8s[r];
  if (k8s) {
    if (jg > 0) {
      const double avail = map_get(o->gpu_model, o->gpu_count, o->gpu_slots, (uint32_t)r, j->gpu_model ? j->gpu_model[k] : 0);
      if (!(avail == jg)) return false;
    } else if (map_count(o->gpu_model, o->gpu_slots, (uint32_t)r) != 0) {
      return false;
    }
  } else if (!(jg == 0)) {
    return false;
  }
  // disk-host (constraints.clj:164-199)
  if (j->disk_request && j->disk_request[k] >= 0 && k8s) {
    const double space = map_get(o->disk_type, o->disk_space, o->disk_slots, (uint32_t)r, j->disk_type[k]);
    if (!(space >= j->disk_request[k])) return false;
  }
  // user-defined EQUALS (constraints.clj:356-377): (= pattern (get nil attribute)) is false
  if (j->eq_off)
    for (uint32_t x = j->eq_off[k]; x < j->eq_off[k + 1]; ++x)
      if (A.attr(r, j->eq_key[x]) != j->eq_val[x]) return false;
  // estimated-completion (constraints.clj:385-401): no "host-start-time" -> passes
  if (r >= 0 && j->est_end_ms && j->est_end_ms[k] != 0 && o->host_start_s && o->host_start_s[r] >= 0) {
    const int64_t death = 1000 * o->host_start_s[r] + 60 * 1000 * p->host_lifetime_mins;
    if (!(j->est_end_ms[k] < death)) return false;
  }
  // checkpoint-locality (constraints.clj:218-240)
  if (j->ckpt_location && j->ckpt_location[k] != 0) {
    const uint32_t loc = (r >= 0 && o->location) ? o->location[r] : 0;
    if (loc != j->ckpt_location[k]) return false;
  }
  return true;
}

// group constraint of the rebalancer (constraints.clj:586-644 through :680-697); cohost_rows = attribute-map rows (-1 nil)
static bool rebal_group_constraint_pass(const cook_groups* g, uint32_t gi, const RebalAttrs& A, int r,
                                        const std::vector<int>& cohost_rows) {
  const uint8_t type = g->type[gi];
  if (type == 0) return true;
  if (type == 1) {  // unique: target hostname must be present and not among the cohosts' hostnames
    const uint32_t target = A.attr(r, COOK_NONE_U32);
    if (target == 0) return false;
    for (int c : cohost_rows)
      if (A.attr(c, COOK_NONE_U32) == target) return false;
    return true;
  }
  const uint32_t key = g->attr_key[gi];
  std::map<uint32_t, int> freq;
  for (int c : cohost_rows) freq[A.attr(c, key)]++;
  if (freq.empty()) return true;
  const uint32_t target = A.attr(r, key);
  if (type == 2) {
    int mn = std::numeric_limits<int>::max(), mx = 0;
    for (auto& kv : freq) {
      mn = std::min(mn, kv.second);
      mx = std::max(mx, kv.second);
    }
    const int minim = (g->minimum[gi] > (int)freq.size()) ? 0 : mn;
    auto it = freq.find(target);
    if (it == freq.end()) return true;
    return minim == mx || it->second < mx;
  }
  return freq.count(target) != 0;
}

int oracle_rebalance(const cook_params* p, const cook_tasks* running, const cook_jobs* pending,
                     const int64_t* pending_job_id, const int32_t* pending_priority, const cook_users* u,
                     const cook_host_spare* spare_in, const cook_offers* host_attrs, const cook_groups* groups,
                     const cook_rebalance_params* rp, cook_preemption* decisions, uint32_t* n_decisions,
                     uint32_t* preempted, uint32_t* n_preempted, double* pending_dru, const oracle_rebal_hooks* hooks) {
  const uint32_t R = running->n, P = pending->n, U = u->n;
  const int32_t* forced_host = hooks ? hooks->forced_host : nullptr;
  const uint32_t* forced_off = hooks ? hooks->forced_off : nullptr;
  const uint32_t* forced_task = hooks ? hooks->forced_task : nullptr;
  const cook_usage* forced_res = hooks ? hooks->forced_res : nullptr;
  uint32_t* final_order = hooks ? hooks->final_order : nullptr;
  double* final_dru = hooks ? hooks->final_dru : nullptr;
  uint32_t* n_final = hooks ? hooks->n_final : nullptr;
  const bool gpu_mode = p->dru_mode == 1;
  const RebalAttrs A(host_attrs);
  struct RT {
    FeatureKey key;
    uint32_t user, host;
    double cpus, mem, gpus, dru;
    uint32_t id;  // < R: index into `running`; >= R: task placed for pending job id - R
    bool slave_known;
  };
  // user -> ordered tasks (sorted-set-by same-user-task-comparator, rebalancer.clj:241-246)
  std::vector<std::vector<RT>> ut(U);
  for (uint32_t i = 0; i < R; ++i) {
    RT x;
    x.key = feature_key(running, i);
    x.user = running->user[i];
    x.host = running->host[i];
    x.cpus = running->cpus[i];
    x.mem = running->mem[i];
    x.gpus = gpus_of(running, i);
    x.dru = 0;
    x.id = i;
    x.slave_known = !(hooks && hooks->running_slave_known) || hooks->running_slave_known[i] != 0;
    ut[x.user].push_back(x);
  }
  auto rescore = [&](uint32_t us) {  // dru.clj:50-80 over the user's ordered tasks
    double cs = 0, ms = 0, gs = 0;
    bool first = true;
    for (auto& x : ut[us]) {
      if (gpu_mode) {
        gs = first ? x.gpus : gs + x.gpus;
        x.dru = gs / u->div_gpus[us];
      } else {
        cs = first ? x.cpus : cs + x.cpus;
        ms = first ? x.mem : ms + x.mem;
        x.dru = std::max(ms / u->div_mem[us], cs / u->div_cpus[us]);
      }
      first = false;
    }
  };
  for (uint32_t us = 0; us < U; ++us) {
    std::sort(ut[us].begin(), ut[us].end(), [](const RT& a, const RT& b) { return key_less(a.key, b.key); });
    rescore(us);
  }
  std::map<uint32_t, cook_usage> spare;  // host -> spare {cpus,mem,gpus}; count unused
  for (uint32_t i = 0; i < spare_in->n; ++i)
    spare[spare_in->host[i]] =
        cook_usage{0, spare_in->cpus[i], spare_in->mem[i], spare_in->gpus ? spare_in->gpus[i] : 0.0};
  // priority-map order: (-dru, user) ascending (rebalancer.clj:252-256); equal keys in the user's task order (UNPINNED)
  struct Ref {
    double dru;
    uint32_t user, order;
    const RT* t;
  };
  auto priority_order = [&](std::vector<Ref>& all) {
    all.clear();
    for (uint32_t w = 0; w < U; ++w) {
      uint32_t ord = 0;
      for (auto& x : ut[w]) all.push_back({x.dru, w, ord++, &x});
    }
    std::stable_sort(all.begin(), all.end(), [](const Ref& a, const Ref& b) {
      if (a.dru != b.dru) return a.dru > b.dru;
      if (a.user != b.user) return a.user < b.user;
      return a.order < b.order;
    });
  };
  std::vector<uint32_t> preempted_hosts;  // hosts of every task preempted so far whose slave id is known
  if (hooks && hooks->init_preempted_hosts)
    preempted_hosts.assign(hooks->init_preempted_hosts, hooks->init_preempted_hosts + hooks->n_init_preempted);
  std::vector<Ref> all;
  uint32_t nd = 0, np = 0;
  int32_t remaining = rp->max_preemption;
  if (pending_dru)
    for (uint32_t pj = 0; pj < P; ++pj) pending_dru[pj] = std::numeric_limits<double>::quiet_NaN();
  for (uint32_t pj = 0; pj < P && remaining > 0; ++pj) {
    const uint32_t us = pending->user[pj];
    const double jc = pending->cpus[pj], jm = pending->mem[pj], jg = pending->gpus ? pending->gpus[pj] : 0.0;
    const bool job_has_gpus = pending->gpus && pending->gpus[pj] > 0;  // (:gpus resources) present
    // rebalancer.clj:210-220 job-below-quota: (conj running-jobs job) -> the job first, then the user's running jobs
    cook_usage fu{1, jc, jm, jg};
    for (auto& x : ut[us]) {
      fu.count += 1;
      fu.cpus += x.cpus;
      fu.mem += x.mem;
      fu.gpus += x.gpus;
    }
    const cook_usage q{u->quota_count[us], u->quota_cpus[us], u->quota_mem[us], u->quota_gpus[us]};
    const bool below = below_quota(q, fu);
    // rebalancer.clj:157-208 pending job dru: nearest task <= synthetic pending task in the user's order
    FeatureKey pk;
    pk.negprio = -(int64_t)pending_priority[pj];
    pk.start = std::numeric_limits<int64_t>::max();
    pk.task = std::numeric_limits<int64_t>::min();
    pk.job = pending_job_id[pj];
    double near = 0.0;
    for (auto& x : ut[us]) {
      if (key_less(pk, x.key)) break;  // x > pending
      near = x.dru;
    }
    const double pdru = gpu_mode ? near + jg / u->div_gpus[us]
                                 : std::max(near + jm / u->div_mem[us], near + jc / u->div_cpus[us]);
    if (pending_dru) pending_dru[pj] = pdru;
    priority_order(all);
    // hostname -> slave id through the scored tasks, last one wins (rebalancer.clj:369-375)
    std::unordered_map<uint32_t, bool> host_slave_known;
    for (auto& r : all) host_slave_known[r.t->host] = r.t->slave_known;
    // rebalancer.clj:339-349 candidates in priority-map order, grouped by host
    std::map<uint32_t, std::vector<const RT*>> by_host;  // sorted by host id = hostname order (:383)
    for (auto& kv : spare) by_host[kv.first];
    for (auto& c : all) {
      if (!(below || c.user == us)) continue;
      if (c.dru < rp->safe_dru_threshold) continue;
      if (!(c.dru - pdru > rp->min_dru_diff)) continue;
      by_host[c.t->host].push_back(c.t);
    }
    // group cohosts: every task preempted so far ++ the group's running cotasks (constraints.clj:686-689)
    const bool has_group = groups && pending->group && pending->group[pj] != COOK_NONE_U32;
    std::vector<int> cohost_rows;
    if (has_group) {
      const uint32_t gi = pending->group[pj];
      for (uint32_t h : preempted_hosts) cohost_rows.push_back(A.row(h));
      if (groups->run_off)
        for (uint32_t x = groups->run_off[gi]; x < groups->run_off[gi + 1]; ++x) cohost_rows.push_back(A.row(groups->run_host[x]));
    }
    // rebalancer.clj:384-404 per-host prefix aggregates; keep those with enough resources; max-key :dru, ties -> last
    bool have = false;
    double best_dru = 0.0;  // (fnil :dru {:dru 0.0}) nil: the nil seed has dru 0.0
    uint32_t best_host = 0;
    size_t best_len = 0;
    cook_usage best_res{0, 0, 0, 0};
    const bool forced = forced_host && forced_host[pj] != -2;
    if (!forced) {
      for (auto& kv : by_host) {
        auto hk = host_slave_known.find(kv.first);
        const int row = (hk != host_slave_known.end() && hk->second) ? A.row(kv.first) : -1;
        if (!rebal_job_constraints_pass(p, pending, pj, A, row)) continue;
        if (has_group && !rebal_group_constraint_pass(groups, pending->group[pj], A, row, cohost_rows)) continue;
        cook_usage agg{0, 0.0, 0.0, 0.0};
        auto sp = spare.find(kv.first);
        auto consider = [&](double d, size_t len) {
          const bool enough = agg.mem >= jm && agg.cpus >= jc && (job_has_gpus ? agg.gpus >= jg : true);
          if (enough && d >= best_dru) {
            have = true;
            best_dru = d;
            best_host = kv.first;
            best_len = len;
            best_res = agg;
          }
        };
        if (sp != spare.end()) {
          agg.cpus += sp->second.cpus;
          agg.mem += sp->second.mem;
          agg.gpus += sp->second.gpus;
          consider(DMAX, 0);
        }
        size_t len = 0;
        for (const RT* x : kv.second) {
          agg.cpus += x->cpus;
          agg.mem += x->mem;
          agg.gpus += x->gpus;
          consider(x->dru, ++len);
        }
      }
      if (!have) continue;
    }
    // the tasks of the decision, in candidate order
    std::vector<const RT*> chosen;
    if (forced) {
      if (forced_host[pj] < 0) continue;  // forced "no decision"
      best_host = (uint32_t)forced_host[pj];
      for (uint32_t x = forced_off[pj]; x < forced_off[pj + 1]; ++x)
        for (auto& r : all)
          if (r.t->id == forced_task[x]) chosen.push_back(r.t);
      best_res = forced_res[pj];  // :mem :cpus :gpus of the decision as given
      best_dru = 0.0;
    } else {
      auto& lst = by_host[best_host];
      chosen.assign(lst.begin(), lst.begin() + best_len);
    }
    // decision + next-state (rebalancer.clj:270-309)
    cook_preemption& d = decisions[nd++];
    d.pending_index = pj;
    d.host = best_host;
    d.dru = best_dru;
    d.cpus = best_res.cpus;
    d.mem = best_res.mem;
    d.gpus = best_res.gpus;
    d.task_off = np;
    d.task_n = 0;
    std::set<uint32_t> changed;
    changed.insert(us);
    std::set<uint32_t> gone;
    const bool new_slave_known = !chosen.empty() && chosen[0]->slave_known;  // slave id of the first preempted task
    for (const RT* x : chosen) {
      changed.insert(x->user);
      gone.insert(x->id);
      preempted[np++] = x->id < R ? x->id : COOK_NONE_U32;  // tasks placed this cycle are reported as NONE (:529 skips them)
      d.task_n++;
      if (x->slave_known) preempted_hosts.push_back(x->host);
    }
    for (uint32_t w : changed) {
      auto& v = ut[w];
      v.erase(std::remove_if(v.begin(), v.end(), [&](const RT& x) { return gone.count(x.id) != 0; }), v.end());
    }
    RT nt;
    nt.key = pk;
    nt.user = us;
    nt.host = best_host;
    nt.cpus = jc;
    nt.mem = jm;
    nt.gpus = jg;
    nt.dru = 0;
    nt.id = R + pj;
    nt.slave_known = new_slave_known;
    {
      auto& v = ut[us];
      auto it = std::upper_bound(v.begin(), v.end(), nt, [](const RT& a, const RT& b) { return key_less(a.key, b.key); });
      v.insert(it, nt);
    }
    for (uint32_t w : changed) rescore(w);
    spare[best_host] = cook_usage{0, best_res.cpus - jc, best_res.mem - jm, best_res.gpus - jg};
    --remaining;
  }
  *n_decisions = nd;
  *n_preempted = np;
  if (final_order && final_dru && n_final) {
    priority_order(all);
    for (size_t i = 0; i < all.size(); ++i) {
      final_order[i] = all[i].t->id;
      final_dru[i] = all[i].dru;
    }
    *n_final = (uint32_t)all.size();
  }
  return 0;
}

}  // extern "C"

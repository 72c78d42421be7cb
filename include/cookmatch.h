/*
 * cookmatch.h — C ABI of libcookmatch.so, the MI355X (gfx950) fair-share match engine.
 *
 * This is the drop-in boundary for ONE path of twosigma/Cook: per-cycle DRU ranking, the jobs x offers
 * feasibility/constraint evaluation, rank-ordered bin-pack placement (what Cook delegates to Netflix Fenzo's
 * TaskScheduler.scheduleOnce) and the rebalancer's preemption decisions.  The reference has no FFI on this
 * path; each entry point below names the Clojure function(s) it replaces (paths relative to the reference's
 * scheduler/ directory).  INTEGRATION.md shows the JNI binding a Cook maintainer would add.
 *
 * Conventions
 *  - Plain C.  No exceptions, no callbacks, no torch types.  All buffers are caller-allocated SoA arrays.
 *  - Identity: users, hosts, attribute keys/values, gpu models, disk types, groups, locations are dense
 *    uint32 ids interned by the host (Clojure keeps id->entity tables).  The engine never sees strings.
 *    USER IDS AND HOST IDS MUST BE ASSIGNED IN ASCENDING NAME ORDER (java String.compareTo): the reference
 *    breaks ties by user name (dru.clj:123, rebalancer.clj:252-256) and orders hosts by name
 *    (rebalancer.clj:383).
 *  - Every function returns COOK_OK (0) or a negative COOK_E_* code; cook_last_error() has the message.
 *    On error the outputs are "no ranking / no matches / no decisions", so the Clojure caller's
 *    catch-Throwable path (scheduler.clj:1521-1535) can restore its offers exactly as today.
 *  - A handle is NOT re-entrant: hold the mutual exclusion the reference holds on the Fenzo object
 *    ((locking fenzo ...), scheduler.clj:665).  Distinct handles (pools) may be used from distinct threads.
 *  - The engine is stateless across calls except for buffers it caches on the device; everything that
 *    Fenzo's TaskTracker would remember (tasks already running on a host) is passed in by the caller.
 *  - All arithmetic on the path is IEEE fp64, round-to-nearest, no flush-to-zero (share.clj:95 makes
 *    DRUs of magnitude 1e-305 legal).
 */
#ifndef COOKMATCH_H
#define COOKMATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the functions declared here are the library's ONLY exports: libcookmatch.so is built with -fvisibility=hidden */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define COOK_OK 0
#define COOK_E_INVALID (-1) /* bad argument / inconsistent sizes            */
#define COOK_E_DEVICE (-2)  /* HIP runtime error (message has the hipError) */
#define COOK_E_NOMEM (-3)
#define COOK_E_STATE (-4)   /* call order violated (run before stage ...)   */

#define COOK_NONE_U32 0xFFFFFFFFu

typedef struct cook_engine cook_engine; /* opaque; one per pool */

/* ---- knobs (every hot-path configuration value of the reference) ------------------------- */
typedef struct cook_params {
  int32_t dru_mode;            /* 0 = :pool.dru-mode/default (cpus,mem), 1 = :pool.dru-mode/gpu (scheduler.clj:2178-2183) */
  int32_t max_over_quota_jobs; /* config.clj:413-416, default 100 (scheduler.clj:2057-2071)                     */
  double offensive_max_mem_mb; /* task-constraints :memory-gb * 1024.0 (scheduler.clj:2219); +inf disables      */
  double offensive_max_cpus;   /* task-constraints :cpus (scheduler.clj:2198-2203); +inf disables               */
  double good_enough_fitness;  /* config.clj:111 default 0.8; (> fitness x) at scheduler.clj:2312-2314; >=1 = off */
  int64_t host_lifetime_mins;  /* estimated-completion-config :host-lifetime-mins (constraints.clj:392-397)     */
  int32_t match_algo;          /* 0 = engine default (= 2; = 3 when six or more engines share the device, COOK_CLASSFIT=0 / 1 forbids /
                                  forces that), 1 serial sweep (one workgroup, one job at a time: the reference form of the
                                  chain), 2 window rounds (eval / merge / resolve launches), 3 class-ordered best fit (one workgroup per pool,
                                  no evaluation launches) where the call's numbers and constraints allow it, else as 2.
                                  Other values: COOK_E_INVALID.  Identical results */
  int32_t reserved;
} cook_params;

/* ---- resource 4-vector used for quotas and usage: {count, cpus, mem, gpus} (tools.clj:883-889) ---------- */
typedef struct cook_usage {
  double count, cpus, mem, gpus;
} cook_usage;

/* ---- tasks: running instances ++ synthetic tasks for pending jobs (tools.clj:582-588) ------------------- */
typedef struct cook_tasks {
  uint32_t n;
  const double* cpus;      /* job resources (tools.clj:247-273)                                        */
  const double* mem;
  const double* gpus;      /* may be NULL (all 0.0)                                                    */
  const uint32_t* user;    /* user id = rank of the user name                                          */
  const int32_t* priority; /* :job/priority, default 50 (tools.clj:612)                                */
  const int64_t* start_ms; /* :instance/start-time in ms; ignored for pending (treated as Long.MAX)    */
  const int64_t* task_id;  /* :db/id of the instance; ignored for pending (nil sorts first)            */
  const int64_t* job_id;   /* :db/id of the job                                                        */
  const uint8_t* pending;  /* 1 = synthetic task of a waiting job, 0 = running instance                */
  const uint32_t* host;    /* running: host id (only used by cook_rebalance); may be NULL for cook_rank */
} cook_tasks;

/* ---- users: DRU divisors (share.clj:75-119,189-210) and quotas (quota.clj:272-295) ---------------------- */
typedef struct cook_users {
  uint32_t n;
  const double* div_cpus; /* share, Double.MAX_VALUE when unset                                         */
  const double* div_mem;
  const double* div_gpus;
  const double* quota_count; /* quota per resource, Double.MAX_VALUE (count: 2^31-1) when unset          */
  const double* quota_cpus;
  const double* quota_mem;
  const double* quota_gpus;
} cook_users;

/* ---- pool-level quota inputs of filter-based-on-quota (scheduler.clj:2134-2157) ------------------------- */
typedef struct cook_pool_quota {
  int32_t has_pool_quota; /* 0: (tools/global-pool-quota pool) is nil -> no filtering (tools.clj:925)    */
  int32_t has_group_quota;
  cook_usage pool_quota;
  cook_usage group_quota;
  cook_usage group_usage; /* aggregate-quota-groups over the member pools (scheduler.clj:2125-2132);
                             the cross-pool sum is the only collective on the path (RCCL all-reduce)   */
  int32_t pool_usage_given; /* 0: engine computes the pool's running usage itself (scheduler.clj:2173)  */
  int32_t reserved;
  cook_usage pool_usage;
} cook_pool_quota;

/* ---- the ranked queue and the per-user state of pending-jobs->considerable-jobs (scheduler.clj:729-762) --- */
typedef struct cook_queue { /* the pool's pending jobs in rank order (the output of cook_rank) */
  uint32_t n;
  const double* cpus;
  const double* mem;
  const double* gpus;      /* may be NULL */
  const uint32_t* user;
  const uint8_t* eligible; /* job-allowed-to-start? (scheduler.clj:747) AND the launch-plugin filter (:748), both
                              evaluated by the host; NULL = all eligible */
} cook_queue;

typedef struct cook_user_state {
  uint32_t n;                 /* users */
  const double* quota_count;  /* user->quota (quota.clj:272-295) */
  const double* quota_cpus;
  const double* quota_mem;
  const double* quota_gpus;
  const double* usage_count;  /* user->usage of the pool's running jobs (scheduler.clj:715-727); 0 for users without any */
  const double* usage_cpus;
  const double* usage_mem;
  const double* usage_gpus;
  const int64_t* tokens_left; /* ratelimit/get-token-count! of the per-user-per-pool launch rate limiter
                                 (tools.clj:943-945); NULL = no limiter (every job counts as passed) */
  int32_t enforce_rate_limit; /* ratelimit/enforce? (tools.clj:936) */
  int32_t has_pool_quota;     /* 0: (tools/global-pool-quota pool) is nil -> no pool filtering (tools.clj:925) */
  cook_usage pool_quota;
  int32_t pool_usage_given;   /* 0: the engine sums the users' usage itself in user-id order (tools.clj:966; the
                                 reference sums in hash-map order, which only matters for non-integer usages) */
  int32_t reserved;
  cook_usage pool_usage;
} cook_user_state;

/* ---- considerable jobs, in rank order (scheduler.clj:729-762, 456-509) ---------------------------------- */
typedef struct cook_jobs {
  uint32_t n;
  const double* cpus;
  const double* mem;
  const double* gpus;            /* may be NULL                                                         */
  const uint32_t* gpu_model;     /* requested model id (constraints.clj:96-103); 0 = none; may be NULL  */
  const uint32_t* user;          /* may be NULL for cook_match                                          */
  const uint32_t* group;         /* group id or COOK_NONE_U32; may be NULL                              */
  const uint32_t* eq_off;        /* CSR [n+1] of user-defined EQUALS constraints (constraints.clj:356)  */
  const uint32_t* eq_key;        /*   attribute key id                                                  */
  const uint32_t* eq_val;        /*   required value id                                                 */
  const uint32_t* novel_off;     /* CSR [n+1] of hosts the job already ran on (constraints.clj:68-94)   */
  const uint32_t* novel_host;
  const int32_t* reserved_host;  /* host reserved FOR this job by the rebalancer, -1 none (scheduler.clj:645-653) */
  const uint32_t* ckpt_location; /* location id of last checkpoint, 0 = none (constraints.clj:201-240)  */
  const int64_t* est_end_ms;     /* estimated end time, 0 = no constraint (constraints.clj:385-431)     */
  const double* disk_request;    /* MiB, <0 = constraint not in effect (constraints.clj:164-199)        */
  const uint32_t* disk_type;
  /* Fenzo's remaining additive resource dimensions (TaskRequestAdapter getPorts / getScalarRequests, scheduler.clj:456-471) */
  const int32_t* ports;          /* (:ports resources) = :job/ports, the NUMBER of ports asked for (tools.clj:271); may be
                                    NULL (all 0)                                                        */
  uint32_t n_scalars;            /* named scalar requests (job->scalar-request, scheduler.clj:177-189): column s is the
                                    request under name s of the caller's name table (<= COOK_MAX_SCALARS names)        */
  uint32_t reserved_;
  const double* scalars;         /* n_scalars columns of n doubles (column s at scalars + s * n); NaN = the job has no
                                    request under that name; may be NULL                                               */
} cook_jobs;
/* Which named scalars a binding has to pass: job->scalar-request yields every :job/resource that carries :resource/amount
 * except gpus, i.e. "cpus" and "mem" (the disk resource stores :resource.disk/request, not an amount: api.clj:818-830), and
 * whatever legacy or custom resource types a deployment's database holds ("disk" among them).  Fenzo tests each as
 * used + request > total against the lease's getScalarValues (offer.clj:57-65) and sums the placed requests by name.  For "cpus"
 * and "mem" that is the SAME comparison on the SAME operands as the cpus / mem test when the TaskRequest's cpus / mem are the
 * job's own (no job-resource-adjustments for the pool, scheduler.clj:473-479): such columns are redundant and need not be
 * passed.  With an adjuster, pass the un-adjusted amounts as scalars 0 / 1 and the adjusted ones as cpus / mem. */
#define COOK_MAX_SCALARS 3

/* ---- offers, one per host (offer.clj:31-76), plus Fenzo's view of tasks already on that host ------------ */
typedef struct cook_offers {
  uint32_t n;
  const double* cpus;        /* cpuCores of the lease (offer.clj:55)                                    */
  const double* mem;         /* memoryMB                                                                */
  const uint32_t* host;      /* host id (hostname rank)                                                 */
  const uint8_t* k8s;        /* attr "compute-cluster-type" == "kubernetes"; NULL = all 0               */
  const uint32_t* gpu_model; /* k8s "gpus" text->scalar map, one model per host; 0 = no gpus; may be NULL */
  const double* gpu_count;
  const uint32_t* disk_type; /* k8s "disk" map, one type per host; may be NULL                          */
  const double* disk_space;
  uint32_t n_attr_keys;      /* attribute table is [n][n_attr_keys], value id 0 = attribute absent      */
  const uint32_t* attr;
  const int32_t* max_tasks;  /* COOK_MAX_TASKS_PER_HOST, -1 absent; may be NULL                         */
  const int32_t* num_tasks;  /* COOK_NUM_TASKS_ON_HOST                                                  */
  const uint32_t* location;  /* COOK_COMPUTE_CLUSTER_LOCATION id; may be NULL                           */
  const int64_t* host_start_s; /* "host-start-time", -1 absent; may be NULL                             */
  const double* run_cpus;    /* sum over tasks Fenzo tracks as running on the host (getTaskAssigner,    */
  const double* run_mem;     /*   scheduler.clj:877-881); NULL = 0                                      */
  const int32_t* run_count;
  /* more than one entry in the k8s "gpus" / "disk" maps of a host (constraints.clj:122-157 reads (get model->count model 0)
     and (count model->count); :164-199 likewise for disk): gpu_model / gpu_count are then [n][gpu_slots] row-major, model 0 =
     empty slot, models distinct within a row; 0 means 1 (the plain per-host columns).  At most COOK_MAX_RES_SLOTS. */
  uint32_t gpu_slots;
  uint32_t disk_slots;       /* likewise disk_type / disk_space as [n][disk_slots]                      */
  const int32_t* ports;      /* number of ports in the lease's "ports" ranges (portRanges, offer.clj:71-73: sum of
                                end - begin + 1); may be NULL (no ports: a job asking for any never fits) */
  uint32_t n_scalars;        /* lease getScalarValues (offer.clj:57-65) under the names of cook_jobs.scalars */
  uint32_t reserved_;
  const double* scalars;     /* n_scalars columns of n doubles: the totals, 0.0 = the lease has no scalar of that name;
                                may be NULL (all 0)                                                      */
} cook_offers;
#define COOK_MAX_RES_SLOTS 4

/* ---- job groups with host-placement constraints (constraints.clj:519-678) ------------------------------- */
typedef struct cook_groups {
  uint32_t n;
  const uint8_t* type;      /* 0 all (none), 1 unique, 2 balanced, 3 attribute-equals                    */
  const uint32_t* attr_key; /* balanced / attribute-equals attribute key id                              */
  const int32_t* minimum;   /* :host-placement.balanced/minimum                                          */
  const uint32_t* run_off;  /* CSR [n+1]: cotasks of the group already running per the DB + Fenzo tracker */
  const uint32_t* run_host; /*   host id of each running cotask                                          */
  const uint32_t* run_attr; /*   value id of attr_key on that host (0 = absent)                          */
} cook_groups;

/* ---- rebalancer ---------------------------------------------------------------------------------------- */
typedef struct cook_rebalance_params { /* rebalancer.clj:535-557 (Datomic :rebalancer/config) */
  double safe_dru_threshold;
  double min_dru_diff;
  int32_t max_preemption;
  int32_t reserved;
} cook_rebalance_params;

typedef struct cook_host_spare { /* host->spare-resources from view-incubating-offers (rebalancer.clj:577-582) */
  uint32_t n;
  const uint32_t* host;
  const double* cpus;
  const double* mem;
  const double* gpus;
} cook_host_spare;

typedef struct cook_preemption { /* one preemption decision (rebalancer.clj:384-404) */
  uint32_t pending_index; /* index into the pending jobs passed in                                       */
  uint32_t host;          /* :hostname                                                                   */
  double dru;             /* :dru of the decision (Double.MAX_VALUE = spare resources only)              */
  double cpus, mem, gpus; /* resources freed on the host                                                 */
  uint32_t task_off;      /* preempted tasks = preempted[task_off .. task_off+task_n)                    */
  uint32_t task_n;
} cook_preemption;

/* ---- lifecycle ------------------------------------------------------------------------------------------ */
int cook_engine_create(const cook_params* params, int device_id, cook_engine** out);
void cook_engine_destroy(cook_engine* e);
int cook_engine_set_params(cook_engine* e, const cook_params* params);
const char* cook_last_error(const cook_engine* e);
const char* cook_version(void);
/* Layout version of the structs and buffer sizes of this header (cook_jobs / cook_offers / cook_offer_params / COOK_WHY_SLOTS changed
 * in 2: ports, named scalars, gpu / disk slot tables, 20 why-slots; 3 adds cook_match_stats_ex and the mask rule of
 * cook_cycle_update; 4: cook_params.match_algo takes 0 / 1 / 2 only, the words [6], [12..17] of the placement statistics changed).
 * A binding compares its compiled-in COOK_ABI_VERSION with the library's before the first call (cook_amd/engine.py load_library,
 * bindings/jni/cookmatch_jni.c Native.create). */
#define COOK_ABI_VERSION 4
int cook_abi_version(void);

/* ---- RANK: replaces sort-jobs-by-dru-helper + filter-based-on-quota + filter-offensive-jobs --------------
 * (scheduler.clj:2073-2091, 2134-2157, 2198-2229; dru.clj:50-126; tools.clj:614-641, 917-933).
 * ranked_pending_idx receives indices into `tasks` of the surviving pending jobs in rank order (capacity =
 * number of pending tasks); dru_of_task (optional, len tasks->n) receives each task's DRU score, NaN for
 * tasks cut by the over-quota limiter.
 * The staged form keeps inputs resident in HBM between cycles: stage (H2D) -> run (kernels only) -> fetch (D2H). */
int cook_rank(cook_engine* e, const cook_tasks* tasks, const cook_users* users, const cook_pool_quota* quota,
              uint32_t* ranked_pending_idx, uint32_t* n_out, double* dru_of_task);
int cook_rank_stage(cook_engine* e, const cook_tasks* tasks, const cook_users* users);
int cook_rank_set_quota(cook_engine* e, const cook_pool_quota* quota);
/* running usage of the pool {count,cpus,mem,gpus} (scheduler.clj:2118-2123, 2173): input to the cross-pool all-reduce */
int cook_rank_pool_usage(cook_engine* e, cook_usage* out);
/* ... of n engines of one device in ONE call from one thread: the pools' sums side by side as pool batches (one launch per kernel for all of them, one
 * stream synchronisation; cook_cycle_run_rank_multi has the mechanism), out[i] for engines[i].  Same numbers as the calls one by one (the reduction order
 * of a pool does not depend on its neighbours).  An engine twice: COOK_E_INVALID. */
int cook_rank_pool_usage_multi(cook_engine** engines, uint32_t n, cook_usage* out /* [n] */);
int cook_rank_run(cook_engine* e);
/* per-user running usage of the pool after cook_rank_run / cook_cycle_run*: usage[u*3 + {0,1,2}] = {cpus, mem, gpus} summed over user
 * u's RUNNING tasks in the user's task order (tools.clj:614-641).  This is the [U x 3] vector BASELINE.json's north_star
 * all-reduces across pools ("cross-pool per-user DRU totals": divide by the user's share, share.clj:189-210); Cook itself
 * keeps usage per pool (scheduler.clj:2167-2194), so no reference function consumes the cross-pool sum.  usage_is_device != 0:
 * `usage` is a DEVICE pointer (e.g. the buffer of the collective): nothing is copied to the host. */
int cook_rank_user_usage(cook_engine* e, double* usage, int usage_is_device);
int cook_rank_fetch(cook_engine* e, uint32_t* ranked_pending_idx, uint32_t* n_out, double* dru_of_task);

/* ---- CONSIDERABLE: replaces pending-jobs->considerable-jobs + tools/filter-pending-jobs-for-quota ----------
 * (scheduler.clj:729-762; tools.clj:654-668, 903-973).  In queue order: per-user quota filter seeded with the user's
 * running usage (state advances on rejected jobs too), launch-rate-limit filter (the n-th surviving job of a user is
 * limited iff n > tokens_left; dropped only when enforcing), pool quota filter seeded with the pool usage, eligible
 * mask, take num_considerable.  considerable_idx receives queue positions (capacity min(num_considerable, queue->n)).
 * rate_limited / passed (optional, len users): per-user counts of the rate-limit stage over the WHOLE queue.  These are UPPER
 * BOUNDS of what the reference stores in pool->user->num-rate-limited-jobs: its lazy pipeline only counts the jobs it consumed
 * (in chunks of 32) before `take num-considerable` was satisfied, so for a queue longer than that the reference's counts stop
 * early.  The considerable jobs themselves do not depend on it.  A caller that shows the counts (the /unscheduled_jobs reason)
 * should present them as "at least one job rate-limited" rather than as exact numbers (oracle-defined). */
int cook_considerable(cook_engine* e, const cook_queue* queue, const cook_user_state* users, uint32_t num_considerable,
                      uint32_t* considerable_idx, uint32_t* n_out, uint32_t* rate_limited, uint32_t* passed);
/* Same filters inside cook_cycle_run, between rank and match, with no host round trip: `users` as above (copied to the
 * device now); eligible_by_pending (optional) is indexed by pending ordinal like cook_cycle_stage's pending_jobs, whose
 * `user` array must be present.  NULL users = plain (take num-considerable) again. */
int cook_cycle_set_considerable(cook_engine* e, const cook_user_state* users, const uint8_t* eligible_by_pending);
/* rank positions of the jobs the last cook_cycle_run considered: cook_cycle_fetch's job_to_offer[k] belongs to the job at
 * ranked_pending_idx[rank_pos[k]] (identity when the considerable filters are off). */
int cook_cycle_fetch_considerable(cook_engine* e, uint32_t* rank_pos, uint32_t* n_out);

/* ---- MATCH: replaces the body of match-offer-to-schedule, i.e. TaskScheduler.scheduleOnce -----------------
 * (scheduler.clj:617-687; Fenzo 0.10.0 pinned at project.clj:46-50; constraints.clj).
 * job_to_offer[k] = offer index or -1.  head_matched mirrors scheduler.clj:1495 (first considerable job matched,
 * or nothing matched at all).  fail_code (optional, len K): 0 matched, else first reason no offer accepted it
 * (bit 0 resources, bit 1 constraints).  reserved_hosts: hosts reserved by the rebalancer for ANY job.
 * Ties between offers of equal fitness go to the lowest offer INDEX (array position).  The reference's recorded simulator run
 * (simulator_files/example-out-trace.csv) is reproduced row for row when the offers of a cycle are passed in DESCENDING hostname
 * order (ascending order yields the mirror image on identical hosts): that is the order a binding should use. */
int cook_match(cook_engine* e, const cook_jobs* considerable, const cook_offers* offers, const cook_groups* groups,
               const uint32_t* reserved_hosts, uint32_t n_reserved, int32_t* job_to_offer, uint32_t* fail_code,
               uint8_t* head_matched);
int cook_match_stage(cook_engine* e, const cook_jobs* considerable, const cook_offers* offers,
                     const cook_groups* groups, const uint32_t* reserved_hosts, uint32_t n_reserved);
int cook_match_run(cook_engine* e);
int cook_match_fetch(cook_engine* e, int32_t* job_to_offer, uint32_t* fail_code, uint8_t* head_matched);

/* ---- CYCLE: rank followed by match of the first K ranked jobs, without a host round trip ------------------
 * (pending-jobs->considerable-jobs "take num-considerable", scheduler.clj:751).  `jobs` must describe the same
 * pending tasks as the staged rank input, indexed by position among the pending tasks (pending ordinal).
 * job_to_offer is indexed by rank position (len = min(K, n_ranked)). */
int cook_cycle_stage(cook_engine* e, const cook_tasks* tasks, const cook_users* users, const cook_jobs* pending_jobs,
                     const cook_offers* offers, const cook_groups* groups, const uint32_t* reserved_hosts,
                     uint32_t n_reserved);

/* What changed between two match cycles of a pool whose inputs are RESIDENT (cook_cycle_stage once, then one delta per cycle):
 * handle-resource-offers! (scheduler.clj:1339-1385) meets almost the same pool every cycle — instances finished or were killed
 * (their task rows leave), jobs were submitted (new pending rows) or launched (the pending row leaves, a running row arrives), and
 * the offers are fresh.  The columns are edited on the device: a STABLE compaction, then the new rows at the end.  Task indices
 * reported afterwards (cook_cycle_fetch's ranked_pending_idx) refer to the updated arrays: row i of the old arrays that was kept is
 * now at i minus the number of removed rows in front of it; add_tasks[r] is at n_kept + r.  add_pending describes the pending tasks
 * of add_tasks, in order, and may only carry optional columns that the staged jobs carry too (else restage); cpus and mem are
 * required, and so is `user` when the staged jobs carry one.  Users, groups and reserved hosts stay as staged.  The eligible mask
 * of cook_cycle_set_considerable moves with the job rows (the jobs a delta adds start out eligible; send a fresh mask to say
 * otherwise).  The call is all-or-nothing: a delta it refuses (COOK_E_INVALID — a row it cannot append, a remove_task entry out of
 * range or named twice, which the device finds) leaves the resident state as it was; the host arrays are read during the call only.
 * ABI note: the struct layouts of this header are versioned by COOK_ABI_VERSION (cook_abi_version()). */
typedef struct cook_cycle_delta {
  uint32_t n_remove;
  const uint32_t* remove_task;  /* indices into the current task arrays, each at most once */
  const cook_tasks* add_tasks;  /* may be NULL */
  const cook_jobs* add_pending; /* may be NULL when no added task is pending */
  const cook_offers* offers;    /* NULL: the staged offers stay; else they are replaced wholesale */
} cook_cycle_delta;
int cook_cycle_update(cook_engine* e, const cook_cycle_delta* delta);

/* Page-locked host memory for input columns and result buffers: copies from / to it run at link speed (pageable memory goes
 * through a bounce buffer at a fraction of it).  NULL on failure. */
void* cook_host_alloc(size_t bytes);
void cook_host_free(void* p);
int cook_cycle_run(cook_engine* e, uint32_t num_considerable);
/* Several pools of one rank (same device) in ONE placement call: cook_cycle_run_rank does the rank / considerable / take-K part of
 * cook_cycle_run for ONE engine (call it for each engine, from any threads), then cook_cycle_match_multi places all of them; results
 * are fetched per engine with cook_cycle_fetch as usual.  Same results as cook_cycle_run on each engine.  How: served walkers — one
 * persistent walker workgroup per pool (one launch per call) beside up to three streams of evaluation launches for whichever pools
 * have a window waiting — or, with COOK_MATCH_SERVED=0 in the environment and as the library's own fall-back, one
 * sequence of launches with blockIdx.z = pool on engines[0]'s stream (pools in lockstep).  Many independent streams of small kernels
 * interfere on one GPU; either form keeps to four.  The call occupies the calling thread until every pool is placed.  Hold every
 * engine's lock across both calls. */
int cook_cycle_run_rank(cook_engine* e, uint32_t num_considerable);
/* cook_cycle_run_rank for n engines of one device in ONE call from ONE thread (replaces scheduler.clj:2425-2435's thread per pool for
 * the rank part; same results per engine as n separate calls; num_considerable[i] is engine i's K — the head-of-queue scaleback,
 * scheduler.clj:1613-1651, moves it per pool).  A pool's rank is a chain of about a hundred small launches and the
 * stage is bound by their number: here the pools' flows run side by side on engines[0]'s stream and a kernel that stands at the same
 * point of several flows is launched ONCE for all of them (blockIdx.y = pool), with one stream synchronisation where each flow would
 * have had its own.  user_usage != NULL: user_usage[i] also receives engine i's per-user running usage [U x 3] exactly as
 * cook_rank_user_usage(engines[i], user_usage[i], usage_is_device) would deliver it after the rank (the collective's payload), inside the
 * same joint sequence.  COOK_RANK_BATCH=0 in the environment: the engines one after another, as cook_cycle_run_rank (+ cook_rank_user_usage).
 * Every engine ONCE: a handle twice in the array is COOK_E_INVALID (here and in cook_cycle_match_multi).
 * Returns the first engine's error that is not COOK_OK; every engine keeps its own message (cook_last_error). */
int cook_cycle_run_rank_multi(cook_engine** engines, uint32_t n, const uint32_t* num_considerable /* [n]: every pool its own K */,
                              double* const* user_usage, int usage_is_device);
int cook_cycle_match_multi(cook_engine** engines, uint32_t n);
int cook_cycle_fetch(cook_engine* e, uint32_t* ranked_pending_idx, uint32_t* n_ranked, int32_t* job_to_offer,
                     uint32_t* n_considered, uint8_t* head_matched);

/* ---- REBALANCE: replaces init-state + the rebalance loop's decisions ---------------------------------------
 * (rebalancer.clj:222-266, 320-407, 270-309, 434-467; dru.clj:128-144).
 * running: the pool's running tasks with host ids.  running_attrs_cached (optional, len running->n): 0 = the instance's
 * slave id has no entry in the agent-attributes-cache (its host then resolves to a nil attribute map when that task is
 * the last scored task of the host, rebalancer.clj:369-375); NULL = all cached.
 * pending: the allowed-to-start pending jobs in rank order (rebalancer.clj:588-590); the loop stops after
 * params->max_preemption decisions.  pending->reserved_host is ignored (the rebalancer evaluates
 * job-constraint-constructors only, constraints.clj:459-466).
 * host_attrs (optional): the agent-attributes-cache as a cook_offers table, one row per cached host (host[i] = host id;
 * cpus/mem and the run_* / *_tasks columns are ignored).  groups (optional): pending->group[p] indexes it; run_host =
 * hosts of the group's running cotasks (run_attr is ignored: attributes come from host_attrs).
 * decisions capacity = pending->n; preempted capacity = running->n + pending->n (task indices into `running`;
 * COOK_NONE_U32 = a job placed earlier in this call, which the caller skips, rebalancer.clj:529).
 * pending_dru (optional, len pending->n): the pending-job DRU of every job examined (rebalancer.clj:157-208), NaN otherwise. */
int cook_rebalance(cook_engine* e, const cook_tasks* running, const uint8_t* running_attrs_cached, const cook_jobs* pending,
                   const int64_t* pending_job_id, const int32_t* pending_priority, const cook_users* users,
                   const cook_host_spare* spare, const cook_offers* host_attrs, const cook_groups* groups,
                   const cook_rebalance_params* params, cook_preemption* decisions, uint32_t* n_decisions,
                   uint32_t* preempted, uint32_t* n_preempted, double* pending_dru);
int cook_rebalance_stage(cook_engine* e, const cook_tasks* running, const uint8_t* running_attrs_cached,
                         const cook_jobs* pending, const int64_t* pending_job_id, const int32_t* pending_priority,
                         const cook_users* users, const cook_host_spare* spare, const cook_offers* host_attrs,
                         const cook_groups* groups, const cook_rebalance_params* params);
int cook_rebalance_run(cook_engine* e);
int cook_rebalance_fetch(cook_engine* e, cook_preemption* decisions, uint32_t* n_decisions, uint32_t* preempted,
                         uint32_t* n_preempted, double* pending_dru);
/* HIP-event time of the last cook_rebalance_run in milliseconds */
int cook_rebalance_timing(cook_engine* e, double* ms);

/* ---- EXPLAIN: the placement-failure summary of a job ("why unscheduled") --------------------------------------------
 * Replaces fenzo-utils/summarize-placement-failure over the TaskAssignmentResults Fenzo returns for an unassigned task
 * (scheduler/fenzo_utils.clj:33-55, written to :job/last-fenzo-placement-failure at :71-89 and read back by
 * unscheduled.clj:95-110).  For every job position of job_pos (an index into the jobs of the engine's LAST match: cook_match_run,
 * cook_cycle_run or the lockstep pair) and every offer, against the state that job saw (the placements of the jobs ranked
 * before it): the resources that did not fit ("cpus" / "mem", one count each per host) or else the FIRST failing hard constraint
 * in the order Fenzo walks them ((into (list) constraints), scheduler.clj:493-501: checkpoint-locality, estimated-completion,
 * user-defined, disk-host, gpu-host, novel-host, max_tasks_per_host, rebalancer-reservation, the group constraint) or else a zero
 * fitness.  counts is [n][COOK_WHY_SLOTS] host counts; the reference's map is {:resources {"cpus" c0 "mem" c1} :constraints
 * {<name> count ...}} over the non-zero slots.  For a job that was matched the row describes the hosts that refused it. */
#define COOK_WHY_CPUS 0                /* :resources "cpus"                                        */
#define COOK_WHY_MEM 1                 /* :resources "mem"                                         */
#define COOK_WHY_FITNESS 2             /* fitness calculator returned 0.0                          */
#define COOK_WHY_CHECKPOINT_LOCALITY 3 /* "checkpoint_locality_constraint" (constraints.clj:218)  */
#define COOK_WHY_ESTIMATED_COMPLETION 4 /* "estimated_completion_constraint" (:385)                */
#define COOK_WHY_USER_DEFINED 5        /* "user_defined_constraint" (:356)                         */
#define COOK_WHY_DISK_HOST 6           /* "disk_host_constraint" (:164)                            */
#define COOK_WHY_GPU_HOST 7            /* "gpu_host_constraint" (:122)                             */
#define COOK_WHY_NOVEL_HOST 8          /* "novel_host_constraint" (:68)                            */
#define COOK_WHY_MAX_TASKS 9           /* "max_tasks_per_host" (:438)                              */
#define COOK_WHY_RESERVATION 10        /* "rebalancer_reservation_constraint" (:242)               */
#define COOK_WHY_GROUP_UNIQUE 11       /* "unique_host_placement_group_constraint" (:586)          */
#define COOK_WHY_GROUP_BALANCED 12     /* "balanced_host_placement_group_constraint" (:600)        */
#define COOK_WHY_GROUP_ATTR_EQUALS 13  /* "attribute_equals_host_placement_group_constraint" (:628) */
#define COOK_WHY_SCALAR0 14            /* :resources <name of scalar 0>; 15, 16: scalars 1, 2 (the failure's message is the name,
                                          fenzo_utils.clj:21-45).  COOK_WHY_CPUS / _MEM are the cpus / mem tests, which are
                                          the named "cpus" / "mem" tests unless a pool adjuster makes them differ */
#define COOK_WHY_PORTS 17              /* ports did not fit.  Fenzo's PORTS failure carries no message, so the reference's
                                          summary has no entry for it (count-resource-failure skips it); kept for operators */
#define COOK_WHY_SLOTS 20
int cook_match_explain(cook_engine* e, const uint32_t* job_pos, uint32_t n, uint32_t* counts);

/* ---- METRICS: the numbers of handle-match-cycle-metrics (scheduler.clj:1210-1280) from the last match, on the device -----
 * resource-maps->stats (scheduler.clj:547-582) of "cpus" and "mem": :totals summed in collection order (bit-identical to the
 * reference's reduce), :percentiles by nearest rank over the sorted values (task_stats.clj:59-80), :largest-by = the last
 * element of the stable sort by that resource (index in collection order).  An empty collection gives NaN / COOK_NONE_U32. */
typedef struct cook_resource_stats {
  double total_cpus, total_mem;
  double p50_cpus, p95_cpus, p100_cpus;
  double p50_mem, p95_mem, p100_mem;
  uint32_t largest_by_cpus, largest_by_mem;
} cook_resource_stats;
typedef struct cook_cycle_metrics {
  uint32_t considerable, matched, unmatched; /* number-considerable-jobs / -matched-jobs / -unmatched-jobs (:1383-1385)     */
  uint32_t offers, offers_scheduled;         /* (count offers), (count offers-scheduled) = leases used (:1372-1374)        */
  uint32_t head_matched;                     /* matched-considerable-jobs-head? (:1381): job 0 is among the matched         */
  uint32_t reserved[2];
  cook_resource_stats jobs;                  /* jobs->stats of the considerable jobs (:594-600)                            */
  cook_resource_stats offer_stats;           /* offers->stats (:584-592)                                                   */
} cook_cycle_metrics;
/* user_considerable / user_matched (optional, len n_users): frequencies of the jobs' users (:1216-1227; needs the jobs' user
 * column).  job_gpus_by_model / offer_gpus_by_model (optional, len n_gpu_models + 1, by model id; jobs without a model under
 * 0): the "gpus/<model>" entries of :totals.  match-percent, queue-was-full? and the unmatched-cycles bookkeeping
 * (:1404-1486) are host arithmetic on these numbers. */
int cook_match_metrics(cook_engine* e, cook_cycle_metrics* out, uint32_t* user_considerable, uint32_t* user_matched, uint32_t n_users,
                       int64_t* job_gpus_by_model, int64_t* offer_gpus_by_model, uint32_t n_gpu_models);

/* ---- OFFERS: replaces the numeric core of kubernetes.compute-cluster/generate-offers ----------------------
 * (kubernetes/compute_cluster.clj:68-190: available = capacity - consumption per node, the schedulable filter, the
 * offer resources and the capacity / consumption totals it publishes; kubernetes/api.clj:747-765 convert-resource-map,
 * :782-847 node-schedulable?, :849-884 get-capacity, :886-930 get-consumption).  The step BEFORE the match path: its
 * output rows are the cook_offers columns of the same names.  Quantity parsing, label / taint inspection and name
 * interning stay with the host; everything numeric is here. */
#define COOK_NODE_UNSCHEDULABLE 1u   /* .getSpec .getUnschedulable is true (api.clj:795-801)                        */
#define COOK_NODE_OTHER_TAINTS 2u    /* a taint other than the pool / deletion-candidate / gpu / tenured taints (:803-814) */
#define COOK_NODE_BLOCKLIST_LABEL 4u /* carries a label of node-blocklist-labels (:825-832)                         */
#define COOK_NODE_GPU_TAINT 8u       /* carries the gpu-node-taint (:836)                                            */
typedef struct cook_nodes { /* node-name->node of one pool, rows in ascending node-name order (offers come out in this order) */
  uint32_t n;
  const uint32_t* host;      /* host id of the node (hostname rank); copied into the offer rows                  */
  const double* cpus;        /* allocatable "cpu" (api.clj:752-754), 0.0 when absent                             */
  const double* mem;         /* allocatable "memory" / memory-multiplier in MiB (:749-751), 0.0 when absent      */
  const int32_t* gpus;       /* allocatable "nvidia.com/gpu" to-int (:756-758), 0 when absent; may be NULL       */
  const uint32_t* gpu_model; /* id of the "gpu-type" label, 0 = no label (:879); may be NULL                     */
  const double* disk;        /* allocatable "ephemeral-storage" / disk-multiplier (:760-762), < 0 = absent; may be NULL */
  const uint32_t* disk_type; /* id of the pool's disk-type label value, 0 = no label (:880); may be NULL         */
  const uint8_t* flags;      /* COOK_NODE_* bits, evaluated by the host; may be NULL (all 0)                     */
  uint32_t n_attr_keys;      /* label table [n][n_attr_keys] in the cook_offers.attr encoding (labels ++ the     */
  const uint32_t* attr;      /*   "compute-cluster-type" attribute, compute_cluster.clj:165-185); may be NULL    */
} cook_nodes;

#define COOK_POD_SYNTHETIC 1u   /* pod name has the synthetic-pod prefix (api.clj:77)                                */
#define COOK_POD_NO_REQUESTS 2u /* no container carries resource requests: the pod's resource map is nil (:908-913) */
typedef struct cook_pods { /* every pod of node-name->pods; pods of one node must appear in that node's list order */
  uint32_t n;
  const uint32_t* node;      /* index into cook_nodes; COOK_NONE_U32 (or >= nodes->n) = no node assigned, or a node
                                without capacity in this pool: dropped (api.clj:894, compute_cluster.clj:88-90)    */
  const double* cpus;        /* sum over containers of the "cpu" requests (merge-with +, api.clj:904-911)         */
  const double* mem;         /* likewise "memory" / memory-multiplier                                             */
  const int32_t* gpus;       /* likewise "nvidia.com/gpu"; may be NULL                                            */
  const uint32_t* gpu_model; /* id of nodeSelector "cloud.google.com/gke-accelerator", 0 = none (:914); may be NULL */
  const double* disk;        /* likewise "ephemeral-storage" / disk-multiplier, < 0 = no container asks; may be NULL */
  const uint32_t* disk_type; /* id of the nodeSelector disk-type label value, 0 = none (:915); may be NULL         */
  const uint8_t* flags;      /* COOK_POD_* bits; may be NULL                                                      */
} cook_pods;

typedef struct cook_offer_params {
  int32_t clobber_synthetic_pods;       /* (:clobber-synthetic-pods (config/kubernetes)) (compute_cluster.clj:71)  */
  int32_t filter_out_unsound_gpu_nodes; /* (:filter-out-unsound-gpu-nodes? (config/kubernetes)) (api.clj:839)       */
  int32_t max_pods_per_node;            /* (cc/max-tasks-per-host compute-cluster) (compute_cluster.clj:49)         */
  uint32_t n_gpu_models;                /* model ids are 1..n_gpu_models (sizes the per-model totals)               */
  uint32_t n_disk_types;                /* disk type ids are 1..n_disk_types                                        */
  uint32_t gpu_slots;                   /* entries per row of cook_node_offers.gpu_model / gpu_count (0 = 1, at most
                                           COOK_MAX_RES_SLOTS): see cook_node_offers                                 */
  uint32_t disk_slots;                  /* likewise disk_type / disk_space                                           */
} cook_offer_params;

#define COOK_NODE_ST_OFFER 1u         /* the node is schedulable: an offer row was emitted                           */
#define COOK_NODE_ST_CONSUMED 2u      /* the node has an entry in node-name->consumed                                */
#define COOK_NODE_ST_FOREIGN_GPU 4u   /* pods consume gpus under more models than the row's gpu_slots hold (see below): the
                                         row lists the first ones; ask again with more slots (COOK_MAX_RES_SLOTS covers
                                         three models beyond the node's own)                                         */
#define COOK_NODE_ST_FOREIGN_DISK 8u  /* likewise for disk types                                                     */
/* (:gpus available) / (:disk available) of a node are MAPS (compute_cluster.clj:91, 180-181): the node's own model -> capacity
 * minus what its pods consume under that model, plus one entry per model only the pods name -- deep-merge-with finds such a
 * key in the consumption map alone and keeps its value as it is (util.clj:208-225), i.e. {model consumed-count}.  A row holds
 * them as gpu_slots (model, count) pairs: the node's own model first, then the others in the order the node's pod list names
 * them, model 0 = empty slot.  The rows feed cook_offers.gpu_model / gpu_count / gpu_slots unchanged. */
typedef struct cook_node_offers { /* caller-allocated columns, capacity nodes->n rows; any column may be NULL */
  uint32_t* node;      /* row -> index into cook_nodes (ascending)                                                  */
  uint32_t* host;      /* nodes->host of that node (:hostname / :slave-id, compute_cluster.clj:175-176)             */
  double* cpus;        /* (max 0.0 (:cpus available)) (:179)                                                        */
  double* mem;         /* (max 0.0 (:mem available)) (:178)                                                         */
  uint32_t* gpu_model; /* [rows][gpu_slots] keys of (:gpus available), 0 = empty slot (:181)                        */
  double* gpu_count;   /*   their values (capacity - consumption; may be negative, the reference does not clamp it) */
  uint32_t* disk_type; /* [rows][disk_slots] keys of (:disk available) (:180)                                       */
  double* disk_space;
  int32_t* num_pods;   /* pods on the node (api.clj:816, the pod-limit test; = COOK_NUM_TASKS_ON_HOST's count)      */
  uint32_t* attr;      /* [rows][nodes->n_attr_keys]: the nodes' label rows, gathered                               */
} cook_node_offers;

typedef struct cook_offer_totals { /* the gauges generate-offers publishes (compute_cluster.clj:113-160) */
  double cpus_capacity, mem_capacity;  /* total-resource over node-name->capacity, summed in node order (the reference
                                          sums in hash-map order: unpinned for non-integer values)                  */
  double cpus_consumed, mem_consumed;  /* total-resource over node-name->consumed                                    */
  uint32_t nodes_total, nodes_schedulable;
} cook_offer_totals;

/* node_status (optional, len nodes->n): COOK_NODE_ST_* bits.  gpu_capacity_by_model / gpu_consumed_by_model (optional,
 * len n_gpu_models + 1, indexed by model id): total-map-resource of :gpus (compute_cluster.clj:95-96).
 * disk_capacity_by_type / disk_consumed_by_type (optional, len n_disk_types + 1): likewise for :disk, node order;
 * consumption under a type the node's capacity does not list is not included (such nodes carry COOK_NODE_ST_FOREIGN_DISK). */
int cook_offers_build(cook_engine* e, const cook_nodes* nodes, const cook_pods* pods, const cook_offer_params* params,
                      cook_node_offers* offers, uint32_t* n_offers, uint8_t* node_status, cook_offer_totals* totals,
                      int64_t* gpu_capacity_by_model, int64_t* gpu_consumed_by_model, double* disk_capacity_by_type,
                      double* disk_consumed_by_type);
int cook_offers_stage(cook_engine* e, const cook_nodes* nodes, const cook_pods* pods, const cook_offer_params* params);
int cook_offers_run(cook_engine* e);
int cook_offers_fetch(cook_engine* e, cook_node_offers* offers, uint32_t* n_offers, uint8_t* node_status,
                      cook_offer_totals* totals, int64_t* gpu_capacity_by_model, int64_t* gpu_consumed_by_model,
                      double* disk_capacity_by_type, double* disk_consumed_by_type);
/* HIP-event time of the last cook_offers_run in milliseconds */
int cook_offers_timing(cook_engine* e, double* ms);
/* The rows of the engine's last cook_offers_run as the offers of a match / cycle IN PLACE (device columns, no host round trip and
 * no copy): cpus, mem, host, compute-cluster-type = kubernetes, the gpu / disk columns and the label rows; with_task_limits != 0
 * adds COOK_MAX_TASKS_PER_HOST = max_pods_per_node and COOK_NUM_TASKS_ON_HOST = the node's pod count (offer.clj:38-46); Fenzo's
 * running-task view (run_*) is empty.  Otherwise exactly cook_match_stage / cook_cycle_stage.  The rows must stay untouched until
 * the match has run: a later cook_offers_run on the same engine invalidates the staging. */
int cook_match_stage_built_offers(cook_engine* e, const cook_jobs* considerable, const cook_groups* groups,
                                  const uint32_t* reserved_hosts, uint32_t n_reserved, int with_task_limits);
int cook_cycle_stage_built_offers(cook_engine* e, const cook_tasks* tasks, const cook_users* users, const cook_jobs* pending_jobs,
                                  const cook_groups* groups, const uint32_t* reserved_hosts, uint32_t n_reserved,
                                  int with_task_limits);

/* Number of jobs of the engine's last match (cook_match_run: the staged jobs; cook_cycle_run / the lockstep calls: the
 * considerable jobs actually taken, <= num_considerable): the length cook_match_fetch / cook_cycle_fetch write. */
int cook_match_count(cook_engine* e, uint32_t* n_jobs);

/* ---- measurement hooks (bench.py): HIP-event time of the last *_run, per stage, in milliseconds ----------- */
int cook_last_timing(cook_engine* e, double* rank_ms, double* match_ms);
/* named kernel timings of the last run: fills up to cap entries, returns count */
int cook_kernel_timings(cook_engine* e, const char** names, double* ms, uint32_t* launches, uint32_t cap);
int cook_set_profiling(cook_engine* e, int enabled);
/* placement statistics of the last match: [0] rounds, [1] matched, [2..5] rounds ended by list-exhausted / touched-set-full /
   group barrier / window end, [6] segments of windows staged for the walk (>= rounds; the excess went on without a launch), [7] jobs resolved, [8] microseconds the resolve phase spent staging
   windows, [9] ... walking them, [10] offers touched (sum over rounds), [11] jobs the walk visited (the rest were settled in
   parallel), [12..15] reserved (0) */
int cook_match_stats(cook_engine* e, uint32_t out[16]);
/* the same, open-ended: fills min(cap, COOK_MATCH_STATS_EX_N) words and returns how many.  [0..15] as cook_match_stats;
   [16] walked jobs whose merged candidate list was cut short because one offer chunk had contributed all its entries (the list
   may not hold every feasible offer); [17..24] the last cook_cycle_match_multi LED by this engine: [17] how it ran (0 lockstep
   launches, 1 served walkers — one persistent walker workgroup per pool beside serve launches —, 2 the same in its stepping form), [18] pools,
   [19] serve iterations, [20] of them empty, [21] pool windows served, [22] microseconds the latch waited for requests, [23] 1 = the
   served match gave up and lockstep launches finished it, [24] streams of serve iterations; [25] with COOK_GUARD=1 in the
   environment (diagnostics: every device buffer sits between two bands of a pattern) the writes found outside a buffer so far, process-wide —
   the call looks at this engine's bands first —, else 0; [26..28] the last cook_cycle_update of this engine: microseconds in the call, microseconds of those the host waited in
   stream synchronisations, device buffers it had to (re)allocate, [29..30] the phase of the call that took the host longest (0 checks, 1 the delta's block,
   2 marks and scans, 3 column compactions, 4 CSR columns, 5 the look at the device, 6 swaps and offers) and its microseconds; [31] reserved (0);
   [32..36] the last cook_cycle_run_rank_multi LED by this engine: pools, launches made, of them for more than one pool, operations
   issued on their own (copies, fills, kernels outside the batched path), stream synchronisations; [37] how the last match was placed (0 window
   rounds, 1 serial sweep, 3 class-ordered best fit), [38] why a match that could have been placed by class-ordered best fit was not (0 = it was; bits:
   1 resources that are not multiples of 2^-20 below 2^30, 2 a job constraint outside {EQUALS on the first 8 attribute keys with values < 256, <= 4 novel
   hosts, unique group, gpu}, 4 ports / named scalars, 8 balanced / attribute-equals groups or more than 16 pending members of a group, 16 gpu maps with
   several entries / max-tasks-per-host / reserved hosts / two offers of one host / attribute values >= 256, 32 too many classes, gpu kinds or offers
   for one workgroup's LDS, 64 job cpus values outside the 8 levels, 128 a job asking for nothing, 0x10000 good-enough-fitness < 1, COOK_CLASSFIT=0 or
   offers built on the device), [39] reserved (0); [40..59] class-ordered best fit, the last match: jobs visited, matched, of them on an offer the call
   had placed on before (overlay lane), offers opened, of them full at once, placements on gpu hosts, epochs, chunk scans, exact turns (several offers
   within 2^-37 of the best: the literal fitness decided), summary re-computations, [50] reserved, batches of 64 jobs, overlay lanes dropped full,
   [53..56] 100 MHz ticks: the launch, its prologue, the epochs' merges, the bookkeeper's batch pre-checks; [57..63] reserved (0) */
#define COOK_MATCH_STATS_EX_N 64
int cook_match_stats_ex(cook_engine* e, uint32_t* out, uint32_t cap);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* COOKMATCH_H */

/*
 * cookmatch_jni.c — JNI shim between twosigma/Cook (Clojure on the JVM) and libcookmatch.so (include/cookmatch.h).
 *
 * The reference has no FFI on the fair-share match path; this is the binding a Cook maintainer would
 * add.  Java side: `package cook.hip; final class Native { static native ... }`, loaded with
 * System.loadLibrary("cookmatch_jni") which links against libcookmatch.so.
 *
 * Marshalling rule (one rule for every call): each cook_* input struct crosses as ONE jobjectArray of direct
 * java.nio.ByteBuffers (native byte order), one element per pointer field of the struct IN THE FIELD ORDER OF
 * include/cookmatch.h; a null element = a NULL (optional) pointer.  Scalars of the struct (n, n_attr_keys, flags) are
 * explicit jint arguments.  Plain-data structs (cook_params, cook_pool_quota, cook_rebalance_params) cross as one
 * direct buffer holding the struct itself.  Outputs are direct buffers sized by the caller as the header documents.
 * GetDirectBufferAddress never copies or pins: the SoA arrays the Clojure side fills are read in place by the
 * H2D copies of the engine.
 *
 * No JDK exists in the build image: the file is compiled in CI against tests/jni_stub/jni.h (type-checks every call
 * against cookmatch.h) and against a real <jni.h> wherever a JDK is present:
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude bindings/jni/cookmatch_jni.c \
 *      -Lcook_amd -lcookmatch -o libcookmatch_jni.so
 */
#include <jni.h>
#include <stdint.h>

#include "cookmatch.h"

#define H(h) ((cook_engine*)(intptr_t)(h))
#define BUF(T, b) ((b) ? (T*)(*env)->GetDirectBufferAddress(env, (b)) : (T*)0)
/* element i of an array of direct buffers (null array or null element -> NULL) */
static void* elem(JNIEnv* env, jobjectArray a, jsize i) {
  jobject b;
  if (!a || i >= (*env)->GetArrayLength(env, a)) return 0;
  b = (*env)->GetObjectArrayElement(env, a, i);
  return b ? (*env)->GetDirectBufferAddress(env, b) : 0;
}
#define EL(T, a, i) ((T*)elem(env, (a), (i)))

static cook_tasks tasks_of(JNIEnv* env, jint n, jobjectArray a) {
  cook_tasks t;
  t.n = (uint32_t)n;
  t.cpus = EL(const double, a, 0);
  t.mem = EL(const double, a, 1);
  t.gpus = EL(const double, a, 2);
  t.user = EL(const uint32_t, a, 3);
  t.priority = EL(const int32_t, a, 4);
  t.start_ms = EL(const int64_t, a, 5);
  t.task_id = EL(const int64_t, a, 6);
  t.job_id = EL(const int64_t, a, 7);
  t.pending = EL(const uint8_t, a, 8);
  t.host = EL(const uint32_t, a, 9);
  return t;
}
static cook_users users_of(JNIEnv* env, jint n, jobjectArray a) {
  cook_users u;
  u.n = (uint32_t)n;
  u.div_cpus = EL(const double, a, 0);
  u.div_mem = EL(const double, a, 1);
  u.div_gpus = EL(const double, a, 2);
  u.quota_count = EL(const double, a, 3);
  u.quota_cpus = EL(const double, a, 4);
  u.quota_mem = EL(const double, a, 5);
  u.quota_gpus = EL(const double, a, 6);
  return u;
}
static cook_jobs jobs_of(JNIEnv* env, jint n, jobjectArray a) {
  cook_jobs j;
  j.n = (uint32_t)n;
  j.cpus = EL(const double, a, 0);
  j.mem = EL(const double, a, 1);
  j.gpus = EL(const double, a, 2);
  j.gpu_model = EL(const uint32_t, a, 3);
  j.user = EL(const uint32_t, a, 4);
  j.group = EL(const uint32_t, a, 5);
  j.eq_off = EL(const uint32_t, a, 6);
  j.eq_key = EL(const uint32_t, a, 7);
  j.eq_val = EL(const uint32_t, a, 8);
  j.novel_off = EL(const uint32_t, a, 9);
  j.novel_host = EL(const uint32_t, a, 10);
  j.reserved_host = EL(const int32_t, a, 11);
  j.ckpt_location = EL(const uint32_t, a, 12);
  j.est_end_ms = EL(const int64_t, a, 13);
  j.disk_request = EL(const double, a, 14);
  j.disk_type = EL(const uint32_t, a, 15);
  return j;
}
static cook_offers offers_of(JNIEnv* env, jint n, jint n_attr_keys, jobjectArray a) {
  cook_offers o;
  o.n = (uint32_t)n;
  o.cpus = EL(const double, a, 0);
  o.mem = EL(const double, a, 1);
  o.host = EL(const uint32_t, a, 2);
  o.k8s = EL(const uint8_t, a, 3);
  o.gpu_model = EL(const uint32_t, a, 4);
  o.gpu_count = EL(const double, a, 5);
  o.disk_type = EL(const uint32_t, a, 6);
  o.disk_space = EL(const double, a, 7);
  o.n_attr_keys = (uint32_t)n_attr_keys;
  o.attr = EL(const uint32_t, a, 8);
  o.max_tasks = EL(const int32_t, a, 9);
  o.num_tasks = EL(const int32_t, a, 10);
  o.location = EL(const uint32_t, a, 11);
  o.host_start_s = EL(const int64_t, a, 12);
  o.run_cpus = EL(const double, a, 13);
  o.run_mem = EL(const double, a, 14);
  o.run_count = EL(const int32_t, a, 15);
  return o;
}
static cook_groups groups_of(JNIEnv* env, jint n, jobjectArray a) {
  cook_groups g;
  g.n = (uint32_t)n;
  g.type = EL(const uint8_t, a, 0);
  g.attr_key = EL(const uint32_t, a, 1);
  g.minimum = EL(const int32_t, a, 2);
  g.run_off = EL(const uint32_t, a, 3);
  g.run_host = EL(const uint32_t, a, 4);
  g.run_attr = EL(const uint32_t, a, 5);
  return g;
}

/* ---- lifecycle ------------------------------------------------------------------------------------------------ */
JNIEXPORT jlong JNICALL Java_cook_hip_Native_create(JNIEnv* env, jclass c, jobject params, jint device) {
  cook_engine* e = 0;
  int rc = cook_engine_create(BUF(const cook_params, params), device, &e);
  (void)c;
  return rc == COOK_OK ? (jlong)(intptr_t)e : (jlong)rc; /* negative = COOK_E_* */
}
JNIEXPORT void JNICALL Java_cook_hip_Native_destroy(JNIEnv* env, jclass c, jlong h) {
  (void)env, (void)c;
  cook_engine_destroy(H(h));
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_setParams(JNIEnv* env, jclass c, jlong h, jobject params) {
  (void)c;
  return cook_engine_set_params(H(h), BUF(const cook_params, params));
}
JNIEXPORT jstring JNICALL Java_cook_hip_Native_lastError(JNIEnv* env, jclass c, jlong h) {
  (void)c;
  return (*env)->NewStringUTF(env, cook_last_error(H(h)));
}

/* ---- rank: scheduler/sort-jobs-by-dru-helper + filter-based-on-quota + filter-offensive-jobs ---------------------- */
JNIEXPORT jint JNICALL Java_cook_hip_Native_rank(JNIEnv* env, jclass c, jlong h, jint n, jobjectArray tasks, jint n_users,
                                                 jobjectArray users, jobject quota, jobject ranked_out, jobject n_out,
                                                 jobject dru_out) {
  cook_tasks t = tasks_of(env, n, tasks);
  cook_users u = users_of(env, n_users, users);
  (void)c;
  return cook_rank(H(h), &t, &u, BUF(const cook_pool_quota, quota), BUF(uint32_t, ranked_out), BUF(uint32_t, n_out),
                   BUF(double, dru_out));
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_rankPoolUsage(JNIEnv* env, jclass c, jlong h, jint n, jobjectArray tasks,
                                                          jint n_users, jobjectArray users, jobject usage_out) {
  cook_tasks t = tasks_of(env, n, tasks);
  cook_users u = users_of(env, n_users, users);
  int rc = cook_rank_stage(H(h), &t, &u);
  (void)c;
  return rc ? rc : cook_rank_pool_usage(H(h), BUF(cook_usage, usage_out));
}

/* ---- considerable: scheduler/pending-jobs->considerable-jobs ------------------------------------------------------- */
JNIEXPORT jint JNICALL Java_cook_hip_Native_considerable(JNIEnv* env, jclass c, jlong h, jint n, jobjectArray queue,
                                                         jint n_users, jobjectArray user_state, jobject tokens,
                                                         jboolean enforce, jobject pool_quota, jobject pool_usage, jint k,
                                                         jobject idx_out, jobject n_out, jobject limited_out,
                                                         jobject passed_out) {
  cook_queue q;
  cook_user_state s;
  (void)c;
  q.n = (uint32_t)n;
  q.cpus = EL(const double, queue, 0);
  q.mem = EL(const double, queue, 1);
  q.gpus = EL(const double, queue, 2);
  q.user = EL(const uint32_t, queue, 3);
  q.eligible = EL(const uint8_t, queue, 4);
  s.n = (uint32_t)n_users;
  s.quota_count = EL(const double, user_state, 0);
  s.quota_cpus = EL(const double, user_state, 1);
  s.quota_mem = EL(const double, user_state, 2);
  s.quota_gpus = EL(const double, user_state, 3);
  s.usage_count = EL(const double, user_state, 4);
  s.usage_cpus = EL(const double, user_state, 5);
  s.usage_mem = EL(const double, user_state, 6);
  s.usage_gpus = EL(const double, user_state, 7);
  s.tokens_left = BUF(const int64_t, tokens);
  s.enforce_rate_limit = enforce ? 1 : 0;
  s.has_pool_quota = pool_quota ? 1 : 0;
  if (pool_quota) s.pool_quota = *BUF(const cook_usage, pool_quota);
  s.pool_usage_given = pool_usage ? 1 : 0;
  s.reserved = 0;
  if (pool_usage) s.pool_usage = *BUF(const cook_usage, pool_usage);
  return cook_considerable(H(h), &q, &s, (uint32_t)k, BUF(uint32_t, idx_out), BUF(uint32_t, n_out),
                           BUF(uint32_t, limited_out), BUF(uint32_t, passed_out));
}

/* ---- match: the body of scheduler/match-offer-to-schedule (TaskScheduler.scheduleOnce) ----------------------------- */
JNIEXPORT jint JNICALL Java_cook_hip_Native_match(JNIEnv* env, jclass c, jlong h, jint k, jobjectArray jobs, jint m,
                                                  jint n_attr_keys, jobjectArray offers, jint n_groups, jobjectArray groups,
                                                  jobject reserved_hosts, jint n_reserved, jobject job_to_offer_out,
                                                  jobject fail_code_out, jobject head_matched_out) {
  cook_jobs j = jobs_of(env, k, jobs);
  cook_offers o = offers_of(env, m, n_attr_keys, offers);
  cook_groups g = groups_of(env, n_groups, groups);
  (void)c;
  return cook_match(H(h), &j, &o, n_groups ? &g : 0, BUF(const uint32_t, reserved_hosts), (uint32_t)n_reserved,
                    BUF(int32_t, job_to_offer_out), BUF(uint32_t, fail_code_out), BUF(uint8_t, head_matched_out));
}

/* ---- cycle: rank -> considerable -> match with inputs resident on the device ------------------------------------------ */
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleStage(JNIEnv* env, jclass c, jlong h, jint n, jobjectArray tasks, jint n_users,
                                                       jobjectArray users, jint n_pending, jobjectArray pending_jobs, jint m,
                                                       jint n_attr_keys, jobjectArray offers, jint n_groups,
                                                       jobjectArray groups, jobject reserved_hosts, jint n_reserved) {
  cook_tasks t = tasks_of(env, n, tasks);
  cook_users u = users_of(env, n_users, users);
  cook_jobs j = jobs_of(env, n_pending, pending_jobs);
  cook_offers o = offers_of(env, m, n_attr_keys, offers);
  cook_groups g = groups_of(env, n_groups, groups);
  (void)c;
  return cook_cycle_stage(H(h), &t, &u, &j, &o, n_groups ? &g : 0, BUF(const uint32_t, reserved_hosts), (uint32_t)n_reserved);
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleRun(JNIEnv* env, jclass c, jlong h, jobject quota, jint num_considerable) {
  int rc = cook_rank_set_quota(H(h), BUF(const cook_pool_quota, quota));
  (void)c;
  return rc ? rc : cook_cycle_run(H(h), (uint32_t)num_considerable);
}
/* several pools of one rank in lockstep: cycleRunRank per engine (any threads), then one cycleMatchMulti(handles) */
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleRunRank(JNIEnv* env, jclass c, jlong h, jobject quota, jint num_considerable) {
  int rc = cook_rank_set_quota(H(h), BUF(const cook_pool_quota, quota));
  (void)c;
  return rc ? rc : cook_cycle_run_rank(H(h), (uint32_t)num_considerable);
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleMatchMulti(JNIEnv* env, jclass c, jobject handles /* direct buffer of n jlong */, jint n) {
  cook_engine* es[64];
  const int64_t* hs = BUF(const int64_t, handles);
  jint i;
  (void)c;
  if (!hs || n <= 0 || n > 64) return COOK_E_INVALID;
  for (i = 0; i < n; ++i) es[i] = H(hs[i]);
  return cook_cycle_match_multi(es, (uint32_t)n);
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleFetch(JNIEnv* env, jclass c, jlong h, jobject ranked_out, jobject n_ranked_out,
                                                       jobject job_to_offer_out, jobject n_considered_out,
                                                       jobject head_matched_out, jobject rank_pos_out) {
  int rc = cook_cycle_fetch(H(h), BUF(uint32_t, ranked_out), BUF(uint32_t, n_ranked_out), BUF(int32_t, job_to_offer_out),
                            BUF(uint32_t, n_considered_out), BUF(uint8_t, head_matched_out));
  (void)c;
  if (rc || !rank_pos_out) return rc;
  return cook_cycle_fetch_considerable(H(h), BUF(uint32_t, rank_pos_out), 0);
}

/* ---- rebalance: rebalancer/init-state + the rebalance loop's decisions ------------------------------------------------- */
JNIEXPORT jint JNICALL Java_cook_hip_Native_rebalance(JNIEnv* env, jclass c, jlong h, jint r, jobjectArray running,
                                                      jobject running_attrs_cached, jint p, jobjectArray pending,
                                                      jobject pending_job_id, jobject pending_priority, jint n_users,
                                                      jobjectArray users, jint n_spare, jobjectArray spare, jint n_attr_rows,
                                                      jint n_attr_keys, jobjectArray host_attrs, jint n_groups,
                                                      jobjectArray groups, jobject rparams, jobject decisions_out,
                                                      jobject n_decisions_out, jobject preempted_out, jobject n_preempted_out,
                                                      jobject pending_dru_out) {
  cook_tasks t = tasks_of(env, r, running);
  cook_jobs j = jobs_of(env, p, pending);
  cook_users u = users_of(env, n_users, users);
  cook_offers a = offers_of(env, n_attr_rows, n_attr_keys, host_attrs);
  cook_groups g = groups_of(env, n_groups, groups);
  cook_host_spare s;
  (void)c;
  s.n = (uint32_t)n_spare;
  s.host = EL(const uint32_t, spare, 0);
  s.cpus = EL(const double, spare, 1);
  s.mem = EL(const double, spare, 2);
  s.gpus = EL(const double, spare, 3);
  return cook_rebalance(H(h), &t, BUF(const uint8_t, running_attrs_cached), &j, BUF(const int64_t, pending_job_id),
                        BUF(const int32_t, pending_priority), &u, &s, host_attrs ? &a : 0, n_groups ? &g : 0,
                        BUF(const cook_rebalance_params, rparams), BUF(cook_preemption, decisions_out),
                        BUF(uint32_t, n_decisions_out), BUF(uint32_t, preempted_out), BUF(uint32_t, n_preempted_out),
                        BUF(double, pending_dru_out));
}

/* ---- offers: the numeric core of kubernetes.compute-cluster/generate-offers (compute_cluster.clj:68-190) ------------ */
static cook_nodes nodes_of(JNIEnv* env, jint n, jint n_attr_keys, jobjectArray a) {
  cook_nodes v;
  v.n = (uint32_t)n;
  v.host = EL(const uint32_t, a, 0);
  v.cpus = EL(const double, a, 1);
  v.mem = EL(const double, a, 2);
  v.gpus = EL(const int32_t, a, 3);
  v.gpu_model = EL(const uint32_t, a, 4);
  v.disk = EL(const double, a, 5);
  v.disk_type = EL(const uint32_t, a, 6);
  v.flags = EL(const uint8_t, a, 7);
  v.n_attr_keys = (uint32_t)n_attr_keys;
  v.attr = EL(const uint32_t, a, 8);
  return v;
}
static cook_pods pods_of(JNIEnv* env, jint n, jobjectArray a) {
  cook_pods p;
  p.n = (uint32_t)n;
  p.node = EL(const uint32_t, a, 0);
  p.cpus = EL(const double, a, 1);
  p.mem = EL(const double, a, 2);
  p.gpus = EL(const int32_t, a, 3);
  p.gpu_model = EL(const uint32_t, a, 4);
  p.disk = EL(const double, a, 5);
  p.disk_type = EL(const uint32_t, a, 6);
  p.flags = EL(const uint8_t, a, 7);
  return p;
}
/* offer_cols: the ten output columns of cook_node_offers as direct buffers, header field order; totals: one direct buffer
 * holding a cook_offer_totals; by_model_type: {gpu capacity, gpu consumed, disk capacity, disk consumed} buffers or nulls. */
JNIEXPORT jint JNICALL Java_cook_hip_Native_offersBuild(JNIEnv* env, jclass c, jlong h, jint n_nodes, jint n_attr_keys,
                                                        jobjectArray nodes, jint n_pods, jobjectArray pods, jobject oparams,
                                                        jobjectArray offer_cols, jobject n_offers_out, jobject node_status_out,
                                                        jobject totals_out, jobjectArray by_model_type) {
  cook_nodes nd = nodes_of(env, n_nodes, n_attr_keys, nodes);
  cook_pods pd = pods_of(env, n_pods, pods);
  cook_node_offers o;
  (void)c;
  o.node = EL(uint32_t, offer_cols, 0);
  o.host = EL(uint32_t, offer_cols, 1);
  o.cpus = EL(double, offer_cols, 2);
  o.mem = EL(double, offer_cols, 3);
  o.gpu_model = EL(uint32_t, offer_cols, 4);
  o.gpu_count = EL(double, offer_cols, 5);
  o.disk_type = EL(uint32_t, offer_cols, 6);
  o.disk_space = EL(double, offer_cols, 7);
  o.num_pods = EL(int32_t, offer_cols, 8);
  o.attr = EL(uint32_t, offer_cols, 9);
  return cook_offers_build(H(h), &nd, &pd, BUF(const cook_offer_params, oparams), &o, BUF(uint32_t, n_offers_out),
                           BUF(uint8_t, node_status_out), BUF(cook_offer_totals, totals_out), EL(int64_t, by_model_type, 0),
                           EL(int64_t, by_model_type, 1), EL(double, by_model_type, 2), EL(double, by_model_type, 3));
}
/* staged form: node / pod state stays resident between cycles, run = kernels only */
JNIEXPORT jint JNICALL Java_cook_hip_Native_offersStage(JNIEnv* env, jclass c, jlong h, jint n_nodes, jint n_attr_keys, jobjectArray nodes,
                                                        jint n_pods, jobjectArray pods, jobject oparams) {
  cook_nodes nd = nodes_of(env, n_nodes, n_attr_keys, nodes);
  cook_pods pd = pods_of(env, n_pods, pods);
  (void)c;
  return cook_offers_stage(H(h), &nd, &pd, BUF(const cook_offer_params, oparams));
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_offersRun(JNIEnv* env, jclass c, jlong h) {
  (void)env, (void)c;
  return cook_offers_run(H(h));
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_offersFetch(JNIEnv* env, jclass c, jlong h, jobjectArray offer_cols, jobject n_offers_out,
                                                        jobject node_status_out, jobject totals_out, jobjectArray by_model_type) {
  cook_node_offers o;
  (void)c;
  o.node = EL(uint32_t, offer_cols, 0);
  o.host = EL(uint32_t, offer_cols, 1);
  o.cpus = EL(double, offer_cols, 2);
  o.mem = EL(double, offer_cols, 3);
  o.gpu_model = EL(uint32_t, offer_cols, 4);
  o.gpu_count = EL(double, offer_cols, 5);
  o.disk_type = EL(uint32_t, offer_cols, 6);
  o.disk_space = EL(double, offer_cols, 7);
  o.num_pods = EL(int32_t, offer_cols, 8);
  o.attr = EL(uint32_t, offer_cols, 9);
  return cook_offers_fetch(H(h), &o, BUF(uint32_t, n_offers_out), BUF(uint8_t, node_status_out), BUF(cook_offer_totals, totals_out),
                           EL(int64_t, by_model_type, 0), EL(int64_t, by_model_type, 1), EL(double, by_model_type, 2),
                           EL(double, by_model_type, 3));
}

/* ---- consumers of the placement's by-products ----------------------------------------------------------------------- */
/* fenzo-utils/summarize-placement-failure (fenzo_utils.clj:33-55): counts_out = direct buffer of n x COOK_WHY_SLOTS uint32 */
JNIEXPORT jint JNICALL Java_cook_hip_Native_matchExplain(JNIEnv* env, jclass c, jlong h, jobject job_pos, jint n, jobject counts_out) {
  (void)c;
  return cook_match_explain(H(h), BUF(const uint32_t, job_pos), (uint32_t)n, BUF(uint32_t, counts_out));
}
/* handle-match-cycle-metrics (scheduler.clj:1210-1280): metrics_out = direct buffer holding a cook_cycle_metrics */
JNIEXPORT jint JNICALL Java_cook_hip_Native_matchMetrics(JNIEnv* env, jclass c, jlong h, jobject metrics_out, jobject user_considerable_out,
                                                         jobject user_matched_out, jint n_users, jobject job_gpus_out,
                                                         jobject offer_gpus_out, jint n_gpu_models) {
  (void)c;
  return cook_match_metrics(H(h), BUF(cook_cycle_metrics, metrics_out), BUF(uint32_t, user_considerable_out),
                            BUF(uint32_t, user_matched_out), (uint32_t)n_users, BUF(int64_t, job_gpus_out), BUF(int64_t, offer_gpus_out),
                            (uint32_t)n_gpu_models);
}
/* the rows of the last offersRun as the offers of a match / cycle, in place on the device */
JNIEXPORT jint JNICALL Java_cook_hip_Native_matchStageBuiltOffers(JNIEnv* env, jclass c, jlong h, jint k, jobjectArray jobs, jint n_groups,
                                                                  jobjectArray groups, jobject reserved_hosts, jint n_reserved,
                                                                  jint with_task_limits) {
  cook_jobs j = jobs_of(env, k, jobs);
  cook_groups g = groups_of(env, n_groups, groups);
  (void)c;
  return cook_match_stage_built_offers(H(h), &j, groups ? &g : 0, BUF(const uint32_t, reserved_hosts), (uint32_t)n_reserved, with_task_limits);
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleStageBuiltOffers(JNIEnv* env, jclass c, jlong h, jint n, jobjectArray tasks, jint n_users,
                                                                  jobjectArray users, jint n_pending, jobjectArray pending_jobs,
                                                                  jint n_groups, jobjectArray groups, jobject reserved_hosts,
                                                                  jint n_reserved, jint with_task_limits) {
  cook_tasks t = tasks_of(env, n, tasks);
  cook_users u = users_of(env, n_users, users);
  cook_jobs j = jobs_of(env, n_pending, pending_jobs);
  cook_groups g = groups_of(env, n_groups, groups);
  (void)c;
  return cook_cycle_stage_built_offers(H(h), &t, &u, &j, groups ? &g : 0, BUF(const uint32_t, reserved_hosts), (uint32_t)n_reserved,
                                       with_task_limits);
}

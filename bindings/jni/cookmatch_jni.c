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
 * cook_jobs / cook_offers take their three extra scalars (n_scalars; gpu_slots, disk_slots) as explicit jints too.
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
/* Every buffer the JVM hands over is checked against the bytes the call will read or write (GetDirectBufferCapacity of a
 * java.nio.ByteBuffer is in bytes): a short buffer sets *bad and the call returns COOK_E_INVALID before the engine sees a
 * pointer.  `need` = 0 skips the check (sizes only the engine knows; the header documents them). */
static void* buf_n(JNIEnv* env, jobject b, uint64_t need, int* bad) {
  void* p;
  if (!b) return 0;
  p = (*env)->GetDirectBufferAddress(env, b);
  if (!p || (need && (uint64_t)(*env)->GetDirectBufferCapacity(env, b) < need)) {
    *bad = 1;
    return 0;
  }
  return p;
}
#define BUFN(T, b, count) ((T*)buf_n(env, (b), (uint64_t)(count) * sizeof(T), &bad))
#define BUF(T, b) ((T*)buf_n(env, (b), sizeof(T), &bad))
/* element i of an array of direct buffers (null array or null element -> NULL); the local reference GetObjectArrayElement
 * creates is released at once: a cycleStage call walks ~60 elements, the JVM only promises room for 16 */
static void* elem_n(JNIEnv* env, jobjectArray a, jsize i, uint64_t need, int* bad) {
  jobject b;
  void* p;
  if (!a || i >= (*env)->GetArrayLength(env, a)) return 0;
  b = (*env)->GetObjectArrayElement(env, a, i);
  if (!b) return 0;
  p = buf_n(env, b, need, bad);
  (*env)->DeleteLocalRef(env, b);
  return p;
}
#define EL(T, a, i, count) ((T*)elem_n(env, (a), (i), (uint64_t)(count) * sizeof(T), bad))
#define CHECKED(call) (bad ? COOK_E_INVALID : (call))

static cook_tasks tasks_of(JNIEnv* env, jint n, jobjectArray a, int* bad) {
  cook_tasks t;
  t.n = (uint32_t)n;
  t.cpus = EL(const double, a, 0, n);
  t.mem = EL(const double, a, 1, n);
  t.gpus = EL(const double, a, 2, n);
  t.user = EL(const uint32_t, a, 3, n);
  t.priority = EL(const int32_t, a, 4, n);
  t.start_ms = EL(const int64_t, a, 5, n);
  t.task_id = EL(const int64_t, a, 6, n);
  t.job_id = EL(const int64_t, a, 7, n);
  t.pending = EL(const uint8_t, a, 8, n);
  t.host = EL(const uint32_t, a, 9, n);
  return t;
}
static cook_users users_of(JNIEnv* env, jint n, jobjectArray a, int* bad) {
  cook_users u;
  u.n = (uint32_t)n;
  u.div_cpus = EL(const double, a, 0, n);
  u.div_mem = EL(const double, a, 1, n);
  u.div_gpus = EL(const double, a, 2, n);
  u.quota_count = EL(const double, a, 3, n);
  u.quota_cpus = EL(const double, a, 4, n);
  u.quota_mem = EL(const double, a, 5, n);
  u.quota_gpus = EL(const double, a, 6, n);
  return u;
}
/* a CSR pair: offsets [n + 1], then payload columns of offsets[n] entries */
static const uint32_t* csr_off(JNIEnv* env, jobjectArray a, jsize i, jint n, uint32_t* total, int* bad) {
  const uint32_t* off = EL(const uint32_t, a, i, (uint64_t)n + 1u);
  *total = (off && n >= 0) ? off[n] : 0u;
  return off;
}
static cook_jobs jobs_of(JNIEnv* env, jint n, jint n_scalars, jobjectArray a, int* bad) {
  cook_jobs j;
  uint32_t ne = 0, nn = 0;
  j.n = (uint32_t)n;
  j.cpus = EL(const double, a, 0, n);
  j.mem = EL(const double, a, 1, n);
  j.gpus = EL(const double, a, 2, n);
  j.gpu_model = EL(const uint32_t, a, 3, n);
  j.user = EL(const uint32_t, a, 4, n);
  j.group = EL(const uint32_t, a, 5, n);
  j.eq_off = csr_off(env, a, 6, n, &ne, bad);
  j.eq_key = EL(const uint32_t, a, 7, ne);
  j.eq_val = EL(const uint32_t, a, 8, ne);
  j.novel_off = csr_off(env, a, 9, n, &nn, bad);
  j.novel_host = EL(const uint32_t, a, 10, nn);
  j.reserved_host = EL(const int32_t, a, 11, n);
  j.ckpt_location = EL(const uint32_t, a, 12, n);
  j.est_end_ms = EL(const int64_t, a, 13, n);
  j.disk_request = EL(const double, a, 14, n);
  j.disk_type = EL(const uint32_t, a, 15, n);
  j.ports = EL(const int32_t, a, 16, n);
  j.n_scalars = (uint32_t)n_scalars;
  j.reserved_ = 0;
  j.scalars = EL(const double, a, 17, (uint64_t)n * (uint64_t)(n_scalars > 0 ? n_scalars : 0));
  if (n_scalars < 0 || n_scalars > COOK_MAX_SCALARS) *bad = 1;
  return j;
}
/* dims = {n_attr_keys, gpu_slots, disk_slots, n_scalars} */
static cook_offers offers_of(JNIEnv* env, jint n, const jint dims[4], jobjectArray a, int* bad) {
  cook_offers o;
  const uint64_t gs = dims[1] > 0 ? (uint64_t)dims[1] : 1u, ds = dims[2] > 0 ? (uint64_t)dims[2] : 1u;
  o.n = (uint32_t)n;
  o.cpus = EL(const double, a, 0, n);
  o.mem = EL(const double, a, 1, n);
  o.host = EL(const uint32_t, a, 2, n);
  o.k8s = EL(const uint8_t, a, 3, n);
  o.gpu_model = EL(const uint32_t, a, 4, (uint64_t)n * gs);
  o.gpu_count = EL(const double, a, 5, (uint64_t)n * gs);
  o.disk_type = EL(const uint32_t, a, 6, (uint64_t)n * ds);
  o.disk_space = EL(const double, a, 7, (uint64_t)n * ds);
  o.n_attr_keys = (uint32_t)dims[0];
  o.attr = EL(const uint32_t, a, 8, (uint64_t)n * (uint64_t)(dims[0] > 0 ? dims[0] : 0));
  o.max_tasks = EL(const int32_t, a, 9, n);
  o.num_tasks = EL(const int32_t, a, 10, n);
  o.location = EL(const uint32_t, a, 11, n);
  o.host_start_s = EL(const int64_t, a, 12, n);
  o.run_cpus = EL(const double, a, 13, n);
  o.run_mem = EL(const double, a, 14, n);
  o.run_count = EL(const int32_t, a, 15, n);
  o.gpu_slots = (uint32_t)dims[1];
  o.disk_slots = (uint32_t)dims[2];
  o.ports = EL(const int32_t, a, 16, n);
  o.n_scalars = (uint32_t)dims[3];
  o.reserved_ = 0;
  o.scalars = EL(const double, a, 17, (uint64_t)n * (uint64_t)(dims[3] > 0 ? dims[3] : 0));
  if (dims[0] < 0 || dims[1] < 0 || dims[1] > COOK_MAX_RES_SLOTS || dims[2] < 0 || dims[2] > COOK_MAX_RES_SLOTS || dims[3] < 0 ||
      dims[3] > COOK_MAX_SCALARS)
    *bad = 1;
  return o;
}
/* the jint[4] of offers_of as a direct buffer of four native ints */
static const jint* dims_of(JNIEnv* env, jobject b, int* bad_out) {
  static const jint none[4] = {0, 0, 0, 0};
  int bad = 0;
  const jint* d = BUFN(const jint, b, 4);
  if (bad) *bad_out = 1;
  return d ? d : none;
}
static cook_groups groups_of(JNIEnv* env, jint n, jobjectArray a, int* bad) {
  cook_groups g;
  uint32_t nr = 0;
  g.n = (uint32_t)n;
  g.type = EL(const uint8_t, a, 0, n);
  g.attr_key = EL(const uint32_t, a, 1, n);
  g.minimum = EL(const int32_t, a, 2, n);
  g.run_off = csr_off(env, a, 3, n, &nr, bad);
  g.run_host = EL(const uint32_t, a, 4, nr);
  g.run_attr = EL(const uint32_t, a, 5, nr);
  return g;
}
#undef EL
#define EL(T, a, i, count) ((T*)elem_n(env, (a), (i), (uint64_t)(count) * sizeof(T), &bad))

/* ---- lifecycle ------------------------------------------------------------------------------------------------ */
JNIEXPORT jlong JNICALL Java_cook_hip_Native_create(JNIEnv* env, jclass c, jobject params, jint device) {
  cook_engine* e = 0;
  int bad = 0, rc;
  const cook_params* p = BUF(const cook_params, params);
  (void)c;
  if (bad) return (jlong)COOK_E_INVALID;
  if (cook_abi_version() != COOK_ABI_VERSION) return (jlong)COOK_E_INVALID; /* a library of another struct layout */
  rc = cook_engine_create(p, device, &e);
  return rc == COOK_OK ? (jlong)(intptr_t)e : (jlong)rc; /* negative = COOK_E_* */
}
JNIEXPORT void JNICALL Java_cook_hip_Native_destroy(JNIEnv* env, jclass c, jlong h) {
  (void)env, (void)c;
  cook_engine_destroy(H(h));
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_setParams(JNIEnv* env, jclass c, jlong h, jobject params) {
  int bad = 0;
  const cook_params* p = BUF(const cook_params, params);
  (void)c;
  return CHECKED(cook_engine_set_params(H(h), p));
}
JNIEXPORT jstring JNICALL Java_cook_hip_Native_lastError(JNIEnv* env, jclass c, jlong h) {
  (void)c;
  return (*env)->NewStringUTF(env, cook_last_error(H(h)));
}
/* page-locked host memory as a direct ByteBuffer (cook_host_alloc): columns filled into it cross the link at full speed */
JNIEXPORT jobject JNICALL Java_cook_hip_Native_hostAlloc(JNIEnv* env, jclass c, jlong bytes) {
  void* p = bytes > 0 ? cook_host_alloc((size_t)bytes) : 0;
  (void)c;
  return p ? (*env)->NewDirectByteBuffer(env, p, bytes) : 0;
}
JNIEXPORT void JNICALL Java_cook_hip_Native_hostFree(JNIEnv* env, jclass c, jobject buffer) {
  (void)c;
  if (buffer) cook_host_free((*env)->GetDirectBufferAddress(env, buffer));
}

/* ---- rank: scheduler/sort-jobs-by-dru-helper + filter-based-on-quota + filter-offensive-jobs ---------------------- */
JNIEXPORT jint JNICALL Java_cook_hip_Native_rank(JNIEnv* env, jclass c, jlong h, jint n, jobjectArray tasks, jint n_users,
                                                 jobjectArray users, jobject quota, jobject ranked_out, jobject n_out,
                                                 jobject dru_out) {
  int bad = 0;
  cook_tasks t = tasks_of(env, n, tasks, &bad);
  cook_users u = users_of(env, n_users, users, &bad);
  const cook_pool_quota* q = BUF(const cook_pool_quota, quota);
  uint32_t* ranked = BUFN(uint32_t, ranked_out, n); /* at most every task is pending */
  uint32_t* n_ranked = BUF(uint32_t, n_out);
  double* dru = BUFN(double, dru_out, n);
  (void)c;
  return CHECKED(cook_rank(H(h), &t, &u, q, ranked, n_ranked, dru));
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_rankPoolUsage(JNIEnv* env, jclass c, jlong h, jint n, jobjectArray tasks,
                                                          jint n_users, jobjectArray users, jobject usage_out) {
  int bad = 0, rc;
  cook_tasks t = tasks_of(env, n, tasks, &bad);
  cook_users u = users_of(env, n_users, users, &bad);
  cook_usage* out = BUF(cook_usage, usage_out);
  (void)c;
  if (bad) return COOK_E_INVALID;
  rc = cook_rank_stage(H(h), &t, &u);
  return rc ? rc : cook_rank_pool_usage(H(h), out);
}
/* the staged tasks' running usage per user, [n_users][3] doubles (count, cpus, mem; gpu pools: count, gpus, -), for the
 * cross-rank sums of a sharded deployment */
JNIEXPORT jint JNICALL Java_cook_hip_Native_rankUserUsage(JNIEnv* env, jclass c, jlong h, jint n_users, jobject usage_out, jint clear) {
  int bad = 0;
  double* out = BUFN(double, usage_out, (uint64_t)(n_users > 0 ? n_users : 0) * 3u);
  (void)c;
  return CHECKED(cook_rank_user_usage(H(h), out, clear));
}

/* ---- considerable: scheduler/pending-jobs->considerable-jobs ------------------------------------------------------- */
JNIEXPORT jint JNICALL Java_cook_hip_Native_considerable(JNIEnv* env, jclass c, jlong h, jint n, jobjectArray queue,
                                                         jint n_users, jobjectArray user_state, jobject tokens,
                                                         jboolean enforce, jobject pool_quota, jobject pool_usage, jint k,
                                                         jobject idx_out, jobject n_out, jobject limited_out,
                                                         jobject passed_out) {
  int bad = 0;
  cook_queue q;
  cook_user_state s;
  const cook_usage *pq = BUF(const cook_usage, pool_quota), *pu = BUF(const cook_usage, pool_usage);
  uint32_t* idx = BUFN(uint32_t, idx_out, (k < n ? k : n) > 0 ? (k < n ? k : n) : 0);
  uint32_t* n_idx = BUF(uint32_t, n_out);
  uint32_t* limited = BUFN(uint32_t, limited_out, n_users);
  uint32_t* passed = BUFN(uint32_t, passed_out, n_users);
  (void)c;
  q.n = (uint32_t)n;
  q.cpus = EL(const double, queue, 0, n);
  q.mem = EL(const double, queue, 1, n);
  q.gpus = EL(const double, queue, 2, n);
  q.user = EL(const uint32_t, queue, 3, n);
  q.eligible = EL(const uint8_t, queue, 4, n);
  s.n = (uint32_t)n_users;
  s.quota_count = EL(const double, user_state, 0, n_users);
  s.quota_cpus = EL(const double, user_state, 1, n_users);
  s.quota_mem = EL(const double, user_state, 2, n_users);
  s.quota_gpus = EL(const double, user_state, 3, n_users);
  s.usage_count = EL(const double, user_state, 4, n_users);
  s.usage_cpus = EL(const double, user_state, 5, n_users);
  s.usage_mem = EL(const double, user_state, 6, n_users);
  s.usage_gpus = EL(const double, user_state, 7, n_users);
  s.tokens_left = BUFN(const int64_t, tokens, n_users);
  s.enforce_rate_limit = enforce ? 1 : 0;
  s.has_pool_quota = pq ? 1 : 0;
  if (pq) s.pool_quota = *pq;
  s.pool_usage_given = pu ? 1 : 0;
  s.reserved = 0;
  if (pu) s.pool_usage = *pu;
  return CHECKED(cook_considerable(H(h), &q, &s, (uint32_t)k, idx, n_idx, limited, passed));
}

/* ---- match: the body of scheduler/match-offer-to-schedule (TaskScheduler.scheduleOnce) ----------------------------- */
JNIEXPORT jint JNICALL Java_cook_hip_Native_match(JNIEnv* env, jclass c, jlong h, jint k, jint n_scalars, jobjectArray jobs, jint m,
                                                  jobject offer_dims, jobjectArray offers, jint n_groups, jobjectArray groups,
                                                  jobject reserved_hosts, jint n_reserved, jobject job_to_offer_out,
                                                  jobject fail_code_out, jobject head_matched_out) {
  int bad = 0;
  cook_jobs j = jobs_of(env, k, n_scalars, jobs, &bad);
  cook_offers o = offers_of(env, m, dims_of(env, offer_dims, &bad), offers, &bad);
  cook_groups g = groups_of(env, n_groups, groups, &bad);
  const uint32_t* res = BUFN(const uint32_t, reserved_hosts, n_reserved);
  int32_t* j2o = BUFN(int32_t, job_to_offer_out, k);
  uint32_t* fail = BUFN(uint32_t, fail_code_out, k);
  uint8_t* head = BUF(uint8_t, head_matched_out);
  (void)c;
  return CHECKED(cook_match(H(h), &j, &o, n_groups ? &g : 0, res, (uint32_t)n_reserved, j2o, fail, head));
}
/* jobs of the engine's last match: the length matchFetch-style outputs need */
JNIEXPORT jint JNICALL Java_cook_hip_Native_matchCount(JNIEnv* env, jclass c, jlong h, jobject n_out) {
  int bad = 0;
  uint32_t* n = BUF(uint32_t, n_out);
  (void)c;
  return CHECKED(cook_match_count(H(h), n));
}

/* ---- cycle: rank -> considerable -> match with inputs resident on the device ------------------------------------------ */
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleStage(JNIEnv* env, jclass c, jlong h, jint n, jobjectArray tasks, jint n_users,
                                                       jobjectArray users, jint n_pending, jint n_scalars, jobjectArray pending_jobs,
                                                       jint m, jobject offer_dims, jobjectArray offers, jint n_groups,
                                                       jobjectArray groups, jobject reserved_hosts, jint n_reserved) {
  int bad = 0;
  cook_tasks t = tasks_of(env, n, tasks, &bad);
  cook_users u = users_of(env, n_users, users, &bad);
  cook_jobs j = jobs_of(env, n_pending, n_scalars, pending_jobs, &bad);
  cook_offers o = offers_of(env, m, dims_of(env, offer_dims, &bad), offers, &bad);
  cook_groups g = groups_of(env, n_groups, groups, &bad);
  const uint32_t* res = BUFN(const uint32_t, reserved_hosts, n_reserved);
  (void)c;
  return CHECKED(cook_cycle_stage(H(h), &t, &u, &j, &o, n_groups ? &g : 0, res, (uint32_t)n_reserved));
}
/* what changed since the last cycle (cook_cycle_update): rows to drop, rows to append, optionally fresh offers (offers == null:
 * the staged ones stay) */
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleUpdate(JNIEnv* env, jclass c, jlong h, jint n_remove, jobject remove_task, jint n_add,
                                                        jobjectArray add_tasks, jint n_add_pending, jint n_scalars,
                                                        jobjectArray add_pending, jint m, jobject offer_dims, jobjectArray offers) {
  int bad = 0;
  cook_tasks t = tasks_of(env, n_add, add_tasks, &bad);
  cook_jobs j = jobs_of(env, n_add_pending, n_scalars, add_pending, &bad);
  cook_offers o = offers_of(env, m, dims_of(env, offer_dims, &bad), offers, &bad);
  cook_cycle_delta d;
  (void)c;
  d.n_remove = (uint32_t)n_remove;
  d.remove_task = BUFN(const uint32_t, remove_task, n_remove);
  d.add_tasks = add_tasks ? &t : 0;
  d.add_pending = add_pending ? &j : 0;
  d.offers = offers ? &o : 0;
  return CHECKED(cook_cycle_update(H(h), &d));
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleRun(JNIEnv* env, jclass c, jlong h, jobject quota, jint num_considerable) {
  int bad = 0, rc;
  const cook_pool_quota* q = BUF(const cook_pool_quota, quota);
  (void)c;
  if (bad) return COOK_E_INVALID;
  rc = cook_rank_set_quota(H(h), q);
  return rc ? rc : cook_cycle_run(H(h), (uint32_t)num_considerable);
}
/* several pools of one rank: cycleRunRank per engine (any threads) or one cycleRunRankMulti(handles), then one cycleMatchMulti(handles) */
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleRunRank(JNIEnv* env, jclass c, jlong h, jobject quota, jint num_considerable) {
  int bad = 0, rc;
  const cook_pool_quota* q = BUF(const cook_pool_quota, quota);
  (void)c;
  if (bad) return COOK_E_INVALID;
  rc = cook_rank_set_quota(H(h), q);
  return rc ? rc : cook_cycle_run_rank(H(h), (uint32_t)num_considerable);
}
/* the rank parts of n engines in ONE call from one thread (cook_cycle_run_rank_multi: the pools' flows side by side, the same kernel of
 * several pools in one launch): quotas = direct buffer of n cook_pool_quota, or NULL when no pool has one; a pool without quota inputs
 * carries has_pool_quota = has_group_quota = 0 in its entry */
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleRunRankMulti(JNIEnv* env, jclass c, jobject handles /* direct buffer of n jlong */, jint n,
                                                              jobject quotas, jobject num_considerable /* direct buffer of n jint: every pool its own K */) {
  cook_engine* es[64];
  int bad = 0, rc;
  const int64_t* hs = BUFN(const int64_t, handles, n > 0 ? n : 0);
  const cook_pool_quota* qs = BUFN(const cook_pool_quota, quotas, n > 0 ? n : 0);
  const uint32_t* ks = BUFN(const uint32_t, num_considerable, n > 0 ? n : 0);
  jint i;
  (void)c;
  if (bad || !hs || !ks || n <= 0 || n > 64) return COOK_E_INVALID;
  for (i = 0; i < n; ++i) {
    es[i] = H(hs[i]);
    rc = cook_rank_set_quota(es[i], qs ? &qs[i] : 0);
    if (rc) return rc;
  }
  return cook_cycle_run_rank_multi(es, (uint32_t)n, ks, 0, 0);
}
/* the running usage of n staged engines of one device in ONE call (cook_rank_pool_usage_multi): usage_out = direct buffer of n cook_usage */
JNIEXPORT jint JNICALL Java_cook_hip_Native_rankPoolUsageMulti(JNIEnv* env, jclass c, jobject handles /* direct buffer of n jlong */, jint n, jobject usage_out) {
  cook_engine* es[64];
  int bad = 0;
  const int64_t* hs = BUFN(const int64_t, handles, n > 0 ? n : 0);
  cook_usage* out = BUFN(cook_usage, usage_out, n > 0 ? n : 0);
  jint i;
  (void)c;
  if (bad || !hs || !out || n <= 0 || n > 64) return COOK_E_INVALID;
  for (i = 0; i < n; ++i) es[i] = H(hs[i]);
  return cook_rank_pool_usage_multi(es, (uint32_t)n, out);
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleMatchMulti(JNIEnv* env, jclass c, jobject handles /* direct buffer of n jlong */, jint n) {
  cook_engine* es[64];
  int bad = 0;
  const int64_t* hs = BUFN(const int64_t, handles, n > 0 ? n : 0);
  jint i;
  (void)c;
  if (bad || !hs || n <= 0 || n > 64) return COOK_E_INVALID;
  for (i = 0; i < n; ++i) es[i] = H(hs[i]);
  return cook_cycle_match_multi(es, (uint32_t)n);
}
/* n_pending = the pending tasks staged (sizes every output: a cycle ranks at most that many and considers no more) */
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleFetch(JNIEnv* env, jclass c, jlong h, jint n_pending, jobject ranked_out,
                                                       jobject n_ranked_out, jobject job_to_offer_out, jobject n_considered_out,
                                                       jobject head_matched_out, jobject rank_pos_out) {
  int bad = 0, rc;
  uint32_t* ranked = BUFN(uint32_t, ranked_out, n_pending);
  uint32_t* n_ranked = BUF(uint32_t, n_ranked_out);
  int32_t* j2o = BUFN(int32_t, job_to_offer_out, n_pending);
  uint32_t* n_cons = BUF(uint32_t, n_considered_out);
  uint8_t* head = BUF(uint8_t, head_matched_out);
  uint32_t* pos = BUFN(uint32_t, rank_pos_out, n_pending);
  (void)c;
  if (bad) return COOK_E_INVALID;
  rc = cook_cycle_fetch(H(h), ranked, n_ranked, j2o, n_cons, head);
  if (rc || !pos) return rc;
  return cook_cycle_fetch_considerable(H(h), pos, 0);
}

/* ---- rebalance: rebalancer/init-state + the rebalance loop's decisions ------------------------------------------------- */
JNIEXPORT jint JNICALL Java_cook_hip_Native_rebalance(JNIEnv* env, jclass c, jlong h, jint r, jobjectArray running,
                                                      jobject running_attrs_cached, jint p, jobjectArray pending,
                                                      jobject pending_job_id, jobject pending_priority, jint n_users,
                                                      jobjectArray users, jint n_spare, jobjectArray spare, jint n_attr_rows,
                                                      jobject attr_dims, jobjectArray host_attrs, jint n_groups,
                                                      jobjectArray groups, jobject rparams, jint max_preemption, jobject decisions_out,
                                                      jobject n_decisions_out, jobject preempted_out, jobject n_preempted_out,
                                                      jobject pending_dru_out) {
  int bad = 0;
  cook_tasks t = tasks_of(env, r, running, &bad);
  cook_jobs j = jobs_of(env, p, 0, pending, &bad);
  cook_users u = users_of(env, n_users, users, &bad);
  cook_offers a = offers_of(env, n_attr_rows, dims_of(env, attr_dims, &bad), host_attrs, &bad);
  cook_groups g = groups_of(env, n_groups, groups, &bad);
  cook_host_spare s;
  const uint8_t* cached = BUFN(const uint8_t, running_attrs_cached, r);
  const int64_t* job_id = BUFN(const int64_t, pending_job_id, p);
  const int32_t* prio = BUFN(const int32_t, pending_priority, p);
  const cook_rebalance_params* rp = BUF(const cook_rebalance_params, rparams);
  /* at most max_preemption decisions (one per pending job examined) and r preempted tasks */
  cook_preemption* dec = BUFN(cook_preemption, decisions_out, (max_preemption < p ? max_preemption : p) > 0 ? (max_preemption < p ? max_preemption : p) : 0);
  uint32_t* n_dec = BUF(uint32_t, n_decisions_out);
  uint32_t* pre = BUFN(uint32_t, preempted_out, r);
  uint32_t* n_pre = BUF(uint32_t, n_preempted_out);
  double* pdru = BUFN(double, pending_dru_out, p);
  (void)c;
  s.n = (uint32_t)n_spare;
  s.host = EL(const uint32_t, spare, 0, n_spare);
  s.cpus = EL(const double, spare, 1, n_spare);
  s.mem = EL(const double, spare, 2, n_spare);
  s.gpus = EL(const double, spare, 3, n_spare);
  return CHECKED(cook_rebalance(H(h), &t, cached, &j, job_id, prio, &u, &s, host_attrs ? &a : 0, n_groups ? &g : 0, rp, dec, n_dec, pre,
                                n_pre, pdru));
}

/* ---- offers: the numeric core of kubernetes.compute-cluster/generate-offers (compute_cluster.clj:68-190) ------------ */
static cook_nodes nodes_of(JNIEnv* env, jint n, jint n_attr_keys, jobjectArray a, int* bad_out) {
  cook_nodes v;
  int bad = 0;
  v.n = (uint32_t)n;
  v.host = EL(const uint32_t, a, 0, n);
  v.cpus = EL(const double, a, 1, n);
  v.mem = EL(const double, a, 2, n);
  v.gpus = EL(const int32_t, a, 3, n);
  v.gpu_model = EL(const uint32_t, a, 4, n);
  v.disk = EL(const double, a, 5, n);
  v.disk_type = EL(const uint32_t, a, 6, n);
  v.flags = EL(const uint8_t, a, 7, n);
  v.n_attr_keys = (uint32_t)n_attr_keys;
  v.attr = EL(const uint32_t, a, 8, (uint64_t)n * (uint64_t)(n_attr_keys > 0 ? n_attr_keys : 0));
  if (bad || n_attr_keys < 0) *bad_out = 1;
  return v;
}
static cook_pods pods_of(JNIEnv* env, jint n, jobjectArray a, int* bad_out) {
  cook_pods p;
  int bad = 0;
  p.n = (uint32_t)n;
  p.node = EL(const uint32_t, a, 0, n);
  p.cpus = EL(const double, a, 1, n);
  p.mem = EL(const double, a, 2, n);
  p.gpus = EL(const int32_t, a, 3, n);
  p.gpu_model = EL(const uint32_t, a, 4, n);
  p.disk = EL(const double, a, 5, n);
  p.disk_type = EL(const uint32_t, a, 6, n);
  p.flags = EL(const uint8_t, a, 7, n);
  if (bad) *bad_out = 1;
  return p;
}
/* the ten output columns of cook_node_offers as direct buffers, header field order, each with room for every node */
static cook_node_offers offer_cols_of(JNIEnv* env, jint n_nodes, jint n_attr_keys, const cook_offer_params* op, jobjectArray cols,
                                      int* bad_out) {
  cook_node_offers o;
  int bad = 0;
  const uint64_t gs = (op && op->gpu_slots) ? op->gpu_slots : 1u, ds = (op && op->disk_slots) ? op->disk_slots : 1u;
  o.node = EL(uint32_t, cols, 0, n_nodes);
  o.host = EL(uint32_t, cols, 1, n_nodes);
  o.cpus = EL(double, cols, 2, n_nodes);
  o.mem = EL(double, cols, 3, n_nodes);
  o.gpu_model = EL(uint32_t, cols, 4, (uint64_t)n_nodes * gs);
  o.gpu_count = EL(double, cols, 5, (uint64_t)n_nodes * gs);
  o.disk_type = EL(uint32_t, cols, 6, (uint64_t)n_nodes * ds);
  o.disk_space = EL(double, cols, 7, (uint64_t)n_nodes * ds);
  o.num_pods = EL(int32_t, cols, 8, n_nodes);
  o.attr = EL(uint32_t, cols, 9, (uint64_t)n_nodes * (uint64_t)(n_attr_keys > 0 ? n_attr_keys : 0));
  if (bad) *bad_out = 1;
  return o;
}
/* totals: one direct buffer holding a cook_offer_totals; by_model_type: {gpu capacity, gpu consumed, disk capacity, disk consumed}
 * buffers (n_gpu_models + 1 / n_disk_types + 1 entries) or nulls */
JNIEXPORT jint JNICALL Java_cook_hip_Native_offersBuild(JNIEnv* env, jclass c, jlong h, jint n_nodes, jint n_attr_keys,
                                                        jobjectArray nodes, jint n_pods, jobjectArray pods, jobject oparams,
                                                        jobjectArray offer_cols, jobject n_offers_out, jobject node_status_out,
                                                        jobject totals_out, jobjectArray by_model_type) {
  int bad = 0;
  cook_nodes nd = nodes_of(env, n_nodes, n_attr_keys, nodes, &bad);
  cook_pods pd = pods_of(env, n_pods, pods, &bad);
  const cook_offer_params* op = BUF(const cook_offer_params, oparams);
  cook_node_offers o = offer_cols_of(env, n_nodes, n_attr_keys, op, offer_cols, &bad);
  uint32_t* n_off = BUF(uint32_t, n_offers_out);
  uint8_t* status = BUFN(uint8_t, node_status_out, n_nodes);
  cook_offer_totals* tot = BUF(cook_offer_totals, totals_out);
  const uint64_t ng = op ? (uint64_t)op->n_gpu_models + 1u : 0u, nt = op ? (uint64_t)op->n_disk_types + 1u : 0u;
  int64_t *gcap = EL(int64_t, by_model_type, 0, ng), *gcons = EL(int64_t, by_model_type, 1, ng);
  double *dcap = EL(double, by_model_type, 2, nt), *dcons = EL(double, by_model_type, 3, nt);
  (void)c;
  return CHECKED(cook_offers_build(H(h), &nd, &pd, op, &o, n_off, status, tot, gcap, gcons, dcap, dcons));
}
/* staged form: node / pod state stays resident between cycles, run = kernels only */
JNIEXPORT jint JNICALL Java_cook_hip_Native_offersStage(JNIEnv* env, jclass c, jlong h, jint n_nodes, jint n_attr_keys, jobjectArray nodes,
                                                        jint n_pods, jobjectArray pods, jobject oparams) {
  int bad = 0;
  cook_nodes nd = nodes_of(env, n_nodes, n_attr_keys, nodes, &bad);
  cook_pods pd = pods_of(env, n_pods, pods, &bad);
  const cook_offer_params* op = BUF(const cook_offer_params, oparams);
  (void)c;
  return CHECKED(cook_offers_stage(H(h), &nd, &pd, op));
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_offersRun(JNIEnv* env, jclass c, jlong h) {
  (void)env, (void)c;
  return cook_offers_run(H(h));
}
/* oparams: the cook_offer_params the rows were staged with (sizes the columns) */
JNIEXPORT jint JNICALL Java_cook_hip_Native_offersFetch(JNIEnv* env, jclass c, jlong h, jint n_nodes, jint n_attr_keys, jobject oparams,
                                                        jobjectArray offer_cols, jobject n_offers_out, jobject node_status_out,
                                                        jobject totals_out, jobjectArray by_model_type) {
  int bad = 0;
  const cook_offer_params* op = BUF(const cook_offer_params, oparams);
  cook_node_offers o = offer_cols_of(env, n_nodes, n_attr_keys, op, offer_cols, &bad);
  uint32_t* n_off = BUF(uint32_t, n_offers_out);
  uint8_t* status = BUFN(uint8_t, node_status_out, n_nodes);
  cook_offer_totals* tot = BUF(cook_offer_totals, totals_out);
  const uint64_t ng = op ? (uint64_t)op->n_gpu_models + 1u : 0u, nt = op ? (uint64_t)op->n_disk_types + 1u : 0u;
  int64_t *gcap = EL(int64_t, by_model_type, 0, ng), *gcons = EL(int64_t, by_model_type, 1, ng);
  double *dcap = EL(double, by_model_type, 2, nt), *dcons = EL(double, by_model_type, 3, nt);
  (void)c;
  return CHECKED(cook_offers_fetch(H(h), offer_cols ? &o : 0, n_off, status, tot, gcap, gcons, dcap, dcons));
}

/* ---- consumers of the placement's by-products ----------------------------------------------------------------------- */
/* fenzo-utils/summarize-placement-failure (fenzo_utils.clj:33-55): counts_out = direct buffer of n x COOK_WHY_SLOTS uint32 */
JNIEXPORT jint JNICALL Java_cook_hip_Native_matchExplain(JNIEnv* env, jclass c, jlong h, jobject job_pos, jint n, jobject counts_out) {
  int bad = 0;
  const uint32_t* pos = BUFN(const uint32_t, job_pos, n);
  uint32_t* counts = BUFN(uint32_t, counts_out, (uint64_t)(n > 0 ? n : 0) * COOK_WHY_SLOTS);
  (void)c;
  return CHECKED(cook_match_explain(H(h), pos, (uint32_t)n, counts));
}
/* handle-match-cycle-metrics (scheduler.clj:1210-1280): metrics_out = direct buffer holding a cook_cycle_metrics */
JNIEXPORT jint JNICALL Java_cook_hip_Native_matchMetrics(JNIEnv* env, jclass c, jlong h, jobject metrics_out, jobject user_considerable_out,
                                                         jobject user_matched_out, jint n_users, jobject job_gpus_out,
                                                         jobject offer_gpus_out, jint n_gpu_models) {
  int bad = 0;
  cook_cycle_metrics* m = BUF(cook_cycle_metrics, metrics_out);
  uint32_t *uc = BUFN(uint32_t, user_considerable_out, n_users), *um = BUFN(uint32_t, user_matched_out, n_users);
  int64_t *jg = BUFN(int64_t, job_gpus_out, (uint64_t)n_gpu_models + 1u), *og = BUFN(int64_t, offer_gpus_out, (uint64_t)n_gpu_models + 1u);
  (void)c;
  return CHECKED(cook_match_metrics(H(h), m, uc, um, (uint32_t)n_users, jg, og, (uint32_t)n_gpu_models));
}
/* the rows of the last offersRun as the offers of a match / cycle, in place on the device */
JNIEXPORT jint JNICALL Java_cook_hip_Native_matchStageBuiltOffers(JNIEnv* env, jclass c, jlong h, jint k, jint n_scalars, jobjectArray jobs,
                                                                  jint n_groups, jobjectArray groups, jobject reserved_hosts,
                                                                  jint n_reserved, jint with_task_limits) {
  int bad = 0;
  cook_jobs j = jobs_of(env, k, n_scalars, jobs, &bad);
  cook_groups g = groups_of(env, n_groups, groups, &bad);
  const uint32_t* res = BUFN(const uint32_t, reserved_hosts, n_reserved);
  (void)c;
  return CHECKED(cook_match_stage_built_offers(H(h), &j, groups ? &g : 0, res, (uint32_t)n_reserved, with_task_limits));
}
JNIEXPORT jint JNICALL Java_cook_hip_Native_cycleStageBuiltOffers(JNIEnv* env, jclass c, jlong h, jint n, jobjectArray tasks, jint n_users,
                                                                  jobjectArray users, jint n_pending, jint n_scalars,
                                                                  jobjectArray pending_jobs, jint n_groups, jobjectArray groups,
                                                                  jobject reserved_hosts, jint n_reserved, jint with_task_limits) {
  int bad = 0;
  cook_tasks t = tasks_of(env, n, tasks, &bad);
  cook_users u = users_of(env, n_users, users, &bad);
  cook_jobs j = jobs_of(env, n_pending, n_scalars, pending_jobs, &bad);
  cook_groups g = groups_of(env, n_groups, groups, &bad);
  const uint32_t* res = BUFN(const uint32_t, reserved_hosts, n_reserved);
  (void)c;
  return CHECKED(cook_cycle_stage_built_offers(H(h), &t, &u, &j, groups ? &g : 0, res, (uint32_t)n_reserved, with_task_limits));
}

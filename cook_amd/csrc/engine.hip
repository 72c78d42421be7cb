// engine.hip — libcookmatch.so: C ABI (include/cookmatch.h) + host orchestration of the HIP kernels.
// One engine = one pool = one HIP stream.  Built by hipcc for gfx950 only (cook_amd/build.py).
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <functional>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/cookmatch.h"
#include "common.hpp"
#include "multi.hpp"
#include "considerable_kernels.hpp"
#include "match_kernels.hpp"
#include "match_v2.hpp"
#include "classfit.hpp"
#include "offers_kernels.hpp"
#include "explain_kernels.hpp"
#include "rank_kernels.hpp"
#include "tile_sort.hpp"
#include "rebalance_kernels.hpp"
#include "scan.hpp"
#include "sort.hpp"

#ifndef HIP_KERNEL_NAME
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#endif

namespace {

// ---- grow-only device buffers ------------------------------------------------------------------------------
// COOK_GUARD=1 (diagnostics; scripts/fuzz_sweep.py runs under it): every buffer is allocated at exactly the size asked for between two
// 4 KB bands of a pattern, and the bands are looked at when the buffer is freed or grown — a kernel that writes before or past a buffer
// is named on stderr ("COOK_GUARD") even when the write lands in mapped memory and faults nothing.
static const bool g_guard = [] {
  const char* s = std::getenv("COOK_GUARD");
  return s && std::atoi(s) != 0;
}();
static std::atomic<unsigned> g_guard_hits{0};
static thread_local unsigned tl_dbuf_allocs = 0;  // device allocations made by this thread (cook_match_stats_ex [28]: a call that grows a buffer pays hipFree + hipMalloc)
constexpr size_t GUARD_BYTES = 4096;
// a pool batch (below, "pool batches") holds launches back until its pools meet at a synchronisation: a buffer must not be freed under them
static void batch_drain_before_free();
struct DBuf {
  void* p = nullptr;
  size_t cap = 0;
  void check_guard() {
    if (!g_guard || !p) return;
    std::vector<unsigned char> h(2 * GUARD_BYTES);
    if (hipMemcpy(h.data(), (char*)p - GUARD_BYTES, GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(h.data() + GUARD_BYTES, (char*)p + cap, GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess)
      return;
    for (size_t x = 0; x < 2 * GUARD_BYTES; ++x)
      if (h[x] != 0xA5) {
        std::fprintf(stderr, "COOK_GUARD: a buffer of %zu bytes was written %s (guard byte %zu)\n", cap, x < GUARD_BYTES ? "BEFORE its start" : "PAST its end",
                     x < GUARD_BYTES ? x : x - GUARD_BYTES);
        g_guard_hits.fetch_add(1);
        (void)hipMemset((char*)p - GUARD_BYTES, 0xA5, GUARD_BYTES);  // re-armed: one report per overrun, not one per look
        (void)hipMemset((char*)p + cap, 0xA5, GUARD_BYTES);
        break;
      }
  }
  void free_now() {
    if (!p) return;
    batch_drain_before_free();
    check_guard();
    (void)hipFree(g_guard ? (void*)((char*)p - GUARD_BYTES) : p);
    p = nullptr;
    cap = 0;
  }
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    ++tl_dbuf_allocs;
    free_now();
    if (g_guard) {
      const size_t want = (bytes + 15) & ~(size_t)15;
      void* base = nullptr;
      COOK_HIP(hipMalloc(&base, want + 2 * GUARD_BYTES));
      COOK_HIP(hipMemset(base, 0xA5, GUARD_BYTES));
      COOK_HIP(hipMemset((char*)base + GUARD_BYTES + want, 0xA5, GUARD_BYTES));
      p = (char*)base + GUARD_BYTES;
      cap = want;
      return;
    }
    size_t want = bytes + bytes / 4 + 256;
    COOK_HIP(hipMalloc(&p, want));
    cap = want;
  }
  void release() { free_now(); }
  DBuf() = default;
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  ~DBuf() { release(); }
};
template <class T>
struct DArr {
  DBuf b;
  T* ptr() { return (T*)b.p; }
  const T* ptr() const { return (const T*)b.p; }
  T* ensure(size_t n) {
    b.ensure((n ? n : 1) * sizeof(T));
    return ptr();
  }
  void release() { b.release(); }
};

template <class T>
struct ScanTmp {
  DArr<SegAgg<T>> agg, carry;
  DArr<unsigned> first_head;
};

struct KernelStat {
  double ms = 0;
  unsigned launches = 0;
};

struct RebalBufs;  // rebalance_host.hpp
struct ConsBufs;   // considerable_host.hpp
struct OfferBufs;  // offers_host.hpp
struct ExplainBufs;  // explain_host.hpp
struct UpdateBufs;   // cycle_update.hpp

}  // namespace

struct cook_engine {
  cook_params params;
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // profiling
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct Pending {
    const char* name;
    hipEvent_t a, b;
  };
  std::vector<Pending> ev_pending;
  std::map<std::string, KernelStat> kstats;
  std::vector<std::string> kstat_names;  // stable storage for cook_kernel_timings
  hipEvent_t ev_stage[4] = {nullptr, nullptr, nullptr, nullptr};
  double rank_ms = 0, match_ms = 0;
  // pinned readback scratch
  unsigned long long* h_scratch = nullptr;  // 64 words
  DArr<unsigned long long> d_scratch64;
  DArr<unsigned> d_counters;

  // ---- rank state ----
  bool rank_staged = false, rank_done = false;
  unsigned upd_phase_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ... by phase (cycle_update.hpp)
  uint32_t upd_us = 0, upd_sync_us = 0, upd_allocs = 0;  // the last cook_cycle_update: microseconds in the call, of them in stream synchronisations, device buffers (re)allocated
  bool pool_usage_known = false;  // pool_usage_val is the running usage of the resident task table (it only changes when the table does)
  cook_usage pool_usage_val{0, 0, 0, 0};
  unsigned N = 0, U = 0, n_pending = 0;
  bool has_gpus = false;
  cook_pool_quota quota{};
  DArr<double> t_cpus, t_mem, t_gpus, u_divc, u_divm, u_divg, u_qcount, u_qcpus, u_qmem, u_qgpus;
  DArr<uint32_t> t_user, permA, permB2, s_user, seg_start, seg_end, inexact_user, rank_of_item, gstart, tpos, titem, tsorted,
      tsorted2, qitemA, qitemB, ranked, pend_ord, hist;
  DArr<int32_t> t_prio;
  DArr<int64_t> t_start, t_task, t_job;
  DArr<uint8_t> t_pending, s_pending, head, keep, thead, dhead;
  DArr<TieCtl> tie_ctl;
  DArr<uint64_t> w0, w1, w2, dkey, nkkey, ckey;
  DArr<SumU4> s_use, pre, quseA, quseB, qpre, pool_usage;
  DArr<SumI> scanI;
  DArr<int> iflag, ones_buf, tied_buf, run_isf, run_nonf, run_posf;
  DArr<SumI> run_scan, run_scan2;
  DArr<uint32_t> run_user, run_orig, run_seg, run_b2c, run_perm;
  DArr<uint64_t> run_dkey;
  DArr<double> uu_out;
  DArr<SumU4> uu_pre;
  DArr<double> dru, dru_out;
  ScanTmp<SumU4> tmpU4;
  ScanTmp<SumI> tmpI;
  uint32_t* permB = nullptr;  // final per-user order (points into permA or permB2)
  uint32_t* permC = nullptr;  // final global order
  // the last cook_cycle_run_rank_multi led by this engine: pools, launches, of them for several pools, operations issued alone, synchronisations
  unsigned batch_stats[5] = {0, 0, 0, 0, 0};
  DArr<uint32_t> permC1, permC2;
  unsigned n_ranked = 0;

  // ---- match state ----
  bool match_staged = false, match_done = false;
  unsigned K = 0, M = 0, G = 0, Kjobs = 0;
  DArr<double> j_scal[3], o_scal[3], m_xscal;
  DArr<int32_t> j_ports, o_ports, m_xports;
  DArr<double> j_cpus, j_mem, j_gpus, j_disk_req, o_cpus, o_mem, o_gpu_count, o_disk_space, o_run_cpus, o_run_mem, m_ac, m_am;
  DArr<uint32_t> j_gpu_model, j_group, j_eq_off, j_eq_key, j_eq_val, j_novel_off, j_novel_host, j_ckpt, j_disk_type, j_index,
      o_host, o_gpu_model, o_disk_type, o_attr, o_location, g_attr_key, g_run_off, g_run_host, g_run_attr, reserved_bits,
      m_fail;
  DArr<int32_t> j_reserved_host, o_max_tasks, o_num_tasks, o_run_count, g_min, m_acount, m_group_last, m_job_prev, m_j2o;
  DArr<int64_t> j_est_end, o_host_start;
  DArr<uint8_t> o_k8s, g_type;
  DArr<unsigned> m_summary;
  DArr<OfferA> v_oa;
  DArr<OfferB> v_ob;
  DArr<OfferW> v_ow;
  DArr<JobRec> v_jr;
  DArr<JobCons> v_jcons;
  DArr<unsigned long long> m_alive, m_jmin;
  DArr<double> v_cand_fit;
  DArr<char> v_prec;
  DArr<int> v_cand_idx, v_ge_idx;
  DArr<uint32_t> v_cinfo, v_jfh;
  DArr<uint64_t> v_colbits;
  DArr<WinCtl> w_ctl;
  DArr<RoundLog> w_rlog;
  DArr<PoolCtx> w_pctx;      // contexts of a multi-pool match led by this engine
  PoolCtx deferred{};        // this engine's match, set up but not run (cook_cycle_run_rank)
  bool has_deferred = false;
  unsigned deferred_k = 0;
  WinCtl deferred_c0{};
  WinCtl* h_multi = nullptr;  // pinned: the pools' WinCtl read-backs
  // served walkers (match_rounds_served): the two streams of a served match led by this engine, its control blocks, what it did
  static constexpr unsigned kMaxServers = 4;
  hipStream_t s_walk = nullptr, s_serve[kMaxServers] = {};
  DArr<ServeSlot> w_slots;
  DArr<ServeCtl> w_sctl;
  ServeHost* h_serve = nullptr;  // pinned, one per server
  struct ServedStats {
    unsigned mode = 0;  // 0 not served, 1 walkers beside serve launches, 2 stepping form
    unsigned pools = 0, servers = 0, iterations = 0, empty_iterations = 0, pools_served = 0, fell_back = 0;
    double latch_wait_ms = 0;
  } served;
  int n_cus = 256;
  DArr<MatchIn> v_in;
  void* h_inbuf = nullptr;  // pinned staging copy of MatchIn
  WinCtl last_ctl{};
  bool groups_simple = true;  // no balanced / attribute-equals group staged (cook_match_stage)
  bool deferred_ge = false;   // the deferred call needs the GE launches (good-enough-fitness < 1)
  MatchIn min{};
  bool cycle_staged = false;
  unsigned cycle_considered = 0;

  // ---- rebalancer state (allocated on first use) ----
  RebalBufs* rb = nullptr;
  // ---- considerable-jobs filters (allocated on first use) ----
  ConsBufs* cb = nullptr;
  // ---- offer construction (allocated on first use) ----
  OfferBufs* ofb = nullptr;
  // ---- why-unscheduled summaries / match-cycle metrics (allocated on first use) ----
  ExplainBufs* xb = nullptr;
  UpdateBufs* ub = nullptr;  // cook_cycle_update (allocated on first use)
  MatchIn last_in{};  // the MatchIn of the last match run (K, j_index as used)
  bool last_in_valid = false;
  unsigned rlog_id = 0;  // suffix of this engine's COOK_ROUND_LOG file
  DArr<uint32_t> j_user;
  bool has_j_user = false;
  // ---- class-ordered best fit (classfit.hpp)
  DArr<CfCtl> cf_ctl;
  DArr<uint64_t> cf_attr8;
  DArr<uint32_t> cf_h2o, cf_pos[3], cf_scr[3], cf_gcount, cf_gmem;
  DArr<CfJob> cf_jobs;
  uint32_t cf_max_host = 0xFFFFFFFFu;   // greatest host id of the staged offers; 0xFFFFFFFF: not known (offers built on the device)
  uint32_t cf_group_run_total = 0;      // running cotasks over all staged groups
  bool has_deferred_cf = false;         // this engine's match is set up for cf_walk and waits for cook_cycle_match_multi
  CfPoolCtx deferred_cf{};
  unsigned last_form = 0;               // how the last match was placed: 0 window rounds, 1 serial sweep, 3 class-ordered best fit
  unsigned cf_inelig = 0;               // why the last match that asked for class-ordered best fit did not get it (CF_X_* bits; 0x10000: switched off / the host's checks)
  uint32_t cf_stats[48] = {};
  char* h_cf = nullptr;                 // pinned: summaries and statistics of the pools of a cf_run led by this engine

  void fail(int code, const std::string& m) { throw cook_error(code, m); }
};

namespace {

// ---- launch wrapper with optional per-kernel HIP-event timing ------------------------------------------------
hipEvent_t take_event(cook_engine* e) {
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t ev;
    COOK_HIP(hipEventCreate(&ev));
    e->ev_pool.push_back(ev);
  }
  return e->ev_pool[e->ev_used++];
}
struct ProfScope {
  cook_engine* e;
  hipEvent_t a = nullptr, b = nullptr;
  const char* name;
  hipStream_t stream;
  ProfScope(cook_engine* e_, const char* n, hipStream_t s = nullptr) : e(e_), name(n), stream(s ? s : e_->stream) {
    if (e->profiling) {
      a = take_event(e);
      b = take_event(e);
      (void)hipEventRecord(a, stream);
    }
  }
  ~ProfScope() {
    if (e->profiling) {
      (void)hipEventRecord(b, stream);
      e->ev_pending.push_back({name, a, b});
    }
  }
};
void prof_collect(cook_engine* e) {
  if (!e->profiling) return;
  for (auto& p : e->ev_pending) {
    float ms = 0;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      auto& s = e->kstats[p.name];
      s.ms += ms;
      s.launches += 1;
    }
  }
  e->ev_pending.clear();
  e->ev_used = 0;
}

// ---- pool batches: the flows of several pools on ONE stream, the same kernel of all of them in ONE launch ---------------------------
// cook_cycle_run_rank_multi runs the rank part of a cycle for every pool of a GPU.  Each pool's flow is the code a single pool runs
// (rank_run, the considerable filters, the set-up of the match), on a fiber of its own: while a flow runs, KM / KL / copy_async /
// memset_async RECORD what they would enqueue, and sync() — every point at which the host needs to read something back — parks the flow.
// When every flow is parked (or finished) the scheduler issues what was recorded: operations of the same kernel that stand at the front
// of several flows become one `cook_multi` launch (multi.hpp: blockIdx.y = pool), everything else is issued as recorded, each flow's
// order kept; then ONE stream synchronisation, and the flows go on.  The flows' decisions (radix digits, tie rounds, queue lengths)
// stay per pool: a pool that needs a pass the others do not simply has a record of its own at that point.
constexpr unsigned BATCH_ARG_BYTES = 496;
struct BatchOp {
  const void* key = nullptr;  // the group launcher of (kernel, block size); null: an operation issued on its own
  void (*launch)(cook_engine* lead, hipStream_t s, const char* name, const BatchOp* const* ops, unsigned n) = nullptr;
  const char* name = "";
  unsigned grid = 0;
  alignas(16) unsigned char args[BATCH_ARG_BYTES];
  std::function<void(cook_engine*, hipStream_t)> generic;
};
struct PoolFlow {
  cook_engine* e = nullptr;
  ucontext_t ctx;
  char* stack = nullptr;
  std::vector<BatchOp> ops;
  size_t cur = 0;
  int state = 0;  // 0 ready to run, 1 parked at a synchronisation, 2 finished
  int rc = COOK_OK;
  std::function<void()> body;
};
struct PoolBatch {
  cook_engine* lead = nullptr;
  hipStream_t stream = nullptr;
  std::vector<PoolFlow> flows;
  ucontext_t main_ctx;
  unsigned launches = 0, grouped = 0, singles = 0, syncs = 0;  // launches made, of them for more than one pool; operations issued alone
};
static thread_local PoolBatch* tl_batch = nullptr;
static thread_local PoolFlow* tl_flow = nullptr;  // the flow running on this thread (null: none, or the scheduler itself)
static inline bool recording() { return tl_flow != nullptr; }
static BatchOp& batch_new_op() {
  tl_flow->ops.emplace_back();
  return tl_flow->ops.back();
}
static void batch_park() {  // the running flow waits until everything recorded so far has run
  PoolFlow* f = tl_flow;
  f->state = 1;
  tl_flow = nullptr;
  swapcontext(&f->ctx, &tl_batch->main_ctx);
}
static void batch_drain_before_free() {
  if (recording() && !tl_flow->ops.empty()) batch_park();
}

template <class Fp>
struct KernelSig;
template <class... A>
struct KernelSig<void (*)(A...)> {
  using Pack = ArgPack<A...>;
  using Args = MultiArgs<A...>;
  template <auto F, int B>
  static void launch_group(cook_engine* lead, hipStream_t s, const char* name, const BatchOp* const* ops, unsigned n) {
    for (unsigned i0 = 0; i0 < n; i0 += Args::PER) {
      const unsigned c = std::min<unsigned>(Args::PER, n - i0);
      Args m{};
      unsigned gmax = 0;
      for (unsigned i = 0; i < c; ++i) {
        m.grid[i] = ops[i0 + i]->grid;
        std::memcpy(&m.a[i], ops[i0 + i]->args, sizeof(Pack));
        gmax = std::max(gmax, m.grid[i]);
      }
      ProfScope _ps(lead, name, s);
      hipLaunchKernelGGL((cook_multi<F, B, A...>), dim3(gmax, c), dim3(B), 0, s, m);
    }
  }
};
// launch of a COOK_KERNEL (a 1-D grid of `grid` blocks of B threads) on the engine's stream — or its record, inside a pool batch
template <auto F, int B, class... X>
void KM(cook_engine* e, const char* name, unsigned grid, const X&... x) {
  using Sig = KernelSig<decltype(F)>;
  static_assert(sizeof(typename Sig::Pack) <= BATCH_ARG_BYTES, "a batched kernel's arguments: pass large structures by pointer");
  if (grid == 0) return;
  const typename Sig::Pack p = Sig::Pack::make(x...);
  BatchOp local;
  BatchOp& op = recording() ? batch_new_op() : local;
  op.key = (const void*)&Sig::template launch_group<F, B>;
  op.launch = &Sig::template launch_group<F, B>;
  op.name = name;
  op.grid = grid;
  std::memcpy(op.args, &p, sizeof(p));
  if (recording()) return;
  const BatchOp* one[1] = {&op};
  op.launch(e, e->stream, name, one, 1);
}

// a __global__ kernel of its own (arguments evaluated here and now; inside a pool batch the launch is recorded and issued alone)
#define KL(name_, kern, grid, block, ...)                                                                     \
  do {                                                                                                       \
    if (recording()) {                                                                                       \
      const auto _a = std::make_tuple(__VA_ARGS__);                                                          \
      const dim3 _g(grid), _b(block);                                                                        \
      const char* _n = name_;                                                                                 \
      BatchOp& _op = batch_new_op();                                                                         \
      _op.name = _n;                                                                                         \
      _op.generic = [=](cook_engine* lead_, hipStream_t s_) {                                                \
        ProfScope _ps(lead_, _n, s_);                                                                        \
        std::apply([&](const auto&... x_) { hipLaunchKernelGGL(kern, _g, _b, 0, s_, x_...); }, _a);          \
      };                                                                                                     \
    } else {                                                                                                 \
      ProfScope _ps(e, name_);                                                                               \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, e->stream, __VA_ARGS__);                          \
    }                                                                                                        \
  } while (0)

// the same on a given stream (timed, when profiling, with events on THAT stream); never part of a pool batch
#define KLS(name, stream_, kern, grid, block, ...)                               \
  do {                                                                        \
    ProfScope _ps(e, name, stream_);                                          \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, stream_, __VA_ARGS__); \
  } while (0)

// copies and fills on the engine's stream (recorded inside a pool batch: a source in host memory must stay as it is until the flow's
// next synchronisation, which is what an asynchronous copy asks for anyway)
void copy_async(cook_engine* e, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  if (!bytes) return;
  if (recording()) {
    BatchOp& op = batch_new_op();
    op.name = "copy";
    op.generic = [=](cook_engine*, hipStream_t s_) { COOK_HIP(hipMemcpyAsync(dst, src, bytes, kind, s_)); };
    return;
  }
  COOK_HIP(hipMemcpyAsync(dst, src, bytes, kind, e->stream));
}
void memset_async(cook_engine* e, void* dst, int value, size_t bytes) {
  if (!bytes) return;
  if (recording()) {
    BatchOp& op = batch_new_op();
    op.name = "fill";
    op.generic = [=](cook_engine*, hipStream_t s_) { COOK_HIP(hipMemsetAsync(dst, value, bytes, s_)); };
    return;
  }
  COOK_HIP(hipMemsetAsync(dst, value, bytes, e->stream));
}

// a few words between device memory and PAGE-LOCKED host memory (read-backs of counters into h_scratch, a control block on its way in).
// Inside a pool batch they are moved by a kernel — the device reads and writes page-locked host memory over the link — so that the eight
// copies of eight pools are one launch and not eight calls of the runtime (COOK_BATCH_COPY_KERNEL=0: recorded copies, issued one by one)
COOK_KERNEL void copy_words_k(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, unsigned nwords) {
  for (unsigned i = threadIdx.x; i < nwords; i += blockDim.x) dst[i] = src[i];
}
// (a plain function: hipcc gave a second namespace-scope lambda initialiser in this anonymous namespace the body of the first — COOK_GUARD's —,
//  found in the disassembly of the library's static initialisers after the switch had read as "off" on the GPU box)
static bool env_switch_on_unless_zero(const char* name) {
  const char* s = std::getenv(name);
  return !(s && std::atoi(s) == 0);
}
static const bool g_batch_copy_kernel = env_switch_on_unless_zero("COOK_BATCH_COPY_KERNEL");
void pinned_copy(cook_engine* e, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  if (recording() && g_batch_copy_kernel && bytes % 4 == 0 && bytes <= 4096 && ((uintptr_t)dst | (uintptr_t)src) % 4 == 0) {
    KM<copy_words_k, COOK_WAVE>(e, "copy_words", 1, (uint32_t*)dst, (const uint32_t*)src, (unsigned)(bytes / 4));
    return;
  }
  copy_async(e, dst, src, bytes, kind);
}

template <class T>
void h2d(cook_engine* e, DArr<T>& d, const T* h, size_t n) {
  d.ensure(n);
  copy_async(e, d.ptr(), h, n * sizeof(T), hipMemcpyHostToDevice);
}
template <class T>
const T* h2d_opt(cook_engine* e, DArr<T>& d, const T* h, size_t n) {
  if (!h) return nullptr;
  h2d(e, d, h, n);
  return d.ptr();
}

// COOK_SYNC_TRACE=1: what the stream synchronisations of a call cost the host (stderr, per cook_rank_run)
static const bool g_sync_trace = std::getenv("COOK_SYNC_TRACE") != nullptr;
static thread_local double tl_sync_ms = 0.0;
static thread_local unsigned tl_syncs = 0;
void sync(cook_engine* e) {  // (always timed: two clock reads against a stream synchronisation)
  if (recording()) {  // inside a pool batch: the flow goes on once every pool's flow has come to such a point and the stream has drained
    batch_park();
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  COOK_HIP(hipStreamSynchronize(e->stream));
  tl_sync_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ++tl_syncs;
}

// issues what the flows have recorded: operations without a key as they stand, the same kernel at the front of several flows as one launch
// COOK_BATCH_TRACE=1: every operation a pool batch issues, to stderr (name x pools; "alone" = issued on its own)
static const bool g_batch_trace = std::getenv("COOK_BATCH_TRACE") != nullptr;
static void batch_flush(PoolBatch& b) {
  const unsigned P = (unsigned)b.flows.size();
  const BatchOp* group[COOK_MULTI_MAX * 8];
  for (;;) {
    for (auto& f : b.flows)
      while (f.cur < f.ops.size() && !f.ops[f.cur].key) {
        if (g_batch_trace) std::fprintf(stderr, "batch: %s alone\n", f.ops[f.cur].name);
        f.ops[f.cur].generic(b.lead, b.stream);
        ++f.cur;
        ++b.singles;
      }
    const void* best = nullptr;
    unsigned best_n = 0;
    for (unsigned i = 0; i < P; ++i) {
      const PoolFlow& f = b.flows[i];
      if (f.cur >= f.ops.size()) continue;
      const void* k = f.ops[f.cur].key;
      unsigned c = 0;
      for (unsigned j = 0; j < P; ++j) c += (b.flows[j].cur < b.flows[j].ops.size() && b.flows[j].ops[b.flows[j].cur].key == k) ? 1u : 0u;
      if (c > best_n) best_n = c, best = k;
    }
    if (!best) break;
    unsigned n = 0;
    const BatchOp* first = nullptr;
    for (auto& f : b.flows)
      if (f.cur < f.ops.size() && f.ops[f.cur].key == best && n < COOK_MULTI_MAX * 8) {
        group[n++] = &f.ops[f.cur];
        if (!first) first = &f.ops[f.cur];
        ++f.cur;
      }
    if (g_batch_trace) std::fprintf(stderr, "batch: %s x %u (grid %u)\n", first->name, n, first->grid);
    first->launch(b.lead, b.stream, first->name, group, n);
    ++b.launches;
    if (n > 1) ++b.grouped;
  }
  for (auto& f : b.flows) f.ops.clear(), f.cur = 0;
}

static void flow_entry() {
  PoolFlow* f = tl_flow;
  cook_engine* e = f->e;
  try {
    f->body();
    e->err.clear();
    f->rc = COOK_OK;
  } catch (const cook_error& ce) {
    e->err = ce.msg;
    f->rc = ce.code;
  } catch (const std::exception& ex) {
    e->err = ex.what();
    f->rc = COOK_E_NOMEM;
  } catch (...) {
    e->err = "unknown exception";
    f->rc = COOK_E_STATE;
  }
  f->state = 2;
  tl_flow = nullptr;
  swapcontext(&f->ctx, &tl_batch->main_ctx);  // (never resumed)
}
constexpr size_t FLOW_STACK_BYTES = 2u << 20;
constexpr size_t FLOW_GUARD_BYTES = 64u << 10;  // below the stack, no access: an overflow faults instead of writing into the heap
// a thread's flow stacks: kept for its next batch, unmapped when the thread ends (an executor's or a JVM's pool thread that once led a batch)
struct FlowStacks {
  std::vector<char*> maps;  // mapping = guard + stack
  ~FlowStacks() {
    for (char* m : maps) munmap(m, FLOW_GUARD_BYTES + FLOW_STACK_BYTES);
  }
  char* stack(size_t i) { return maps[i] + FLOW_GUARD_BYTES; }
};
static thread_local FlowStacks tl_flow_stacks;
// runs the flows to completion; returns the first flow's error code that is not COOK_OK (every engine keeps its own message)
static int batch_run(PoolBatch& b) {
  const unsigned P = (unsigned)b.flows.size();
  while (tl_flow_stacks.maps.size() < P) {
    void* m = mmap(nullptr, FLOW_GUARD_BYTES + FLOW_STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
    if (m == MAP_FAILED) throw cook_error(COOK_E_NOMEM, "pool batch: no memory for a flow's stack");
    (void)mprotect(m, FLOW_GUARD_BYTES, PROT_NONE);
    tl_flow_stacks.maps.push_back((char*)m);
  }
  for (unsigned i = 0; i < P; ++i) {
    PoolFlow& f = b.flows[i];
    f.stack = tl_flow_stacks.stack(i);
    f.state = 0;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = FLOW_STACK_BYTES;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, flow_entry, 0);
  }
  struct Reset {
    ~Reset() { tl_batch = nullptr, tl_flow = nullptr; }
  } reset;
  tl_batch = &b;
  // an error on the scheduler's own side (a flush, the synchronisation): the parked flows are never resumed — every engine of the batch is left
  // failed, with the clean-up guarded() gives an engine whose own call threw
  auto abandon = [&](int code, const std::string& msg) {
    for (auto& f : b.flows) {
      if (f.state == 2 && f.rc != COOK_OK) continue;  // (keeps its own message)
      f.rc = code;
      f.e->err = msg;
      f.e->ev_pending.clear();
      f.e->ev_used = 0;
    }
  };
  try {
  for (;;) {
    for (auto& f : b.flows)
      if (f.state == 0) {
        tl_flow = &f;
        swapcontext(&b.main_ctx, &f.ctx);
        tl_flow = nullptr;
      }
    batch_flush(b);
    bool parked = false;
    for (auto& f : b.flows) parked = parked || f.state == 1;
    const auto t0 = std::chrono::steady_clock::now();
    if (g_batch_trace) std::fprintf(stderr, "batch: synchronise\n");
    COOK_HIP(hipStreamSynchronize(b.stream));
    tl_sync_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ++tl_syncs;
    ++b.syncs;
    if (!parked) break;
    for (auto& f : b.flows)
      if (f.state == 1) f.state = 0;
  }
  } catch (const cook_error& ce) {
    abandon(ce.code, ce.msg);
    throw;
  } catch (const std::exception& ex) {
    abandon(COOK_E_NOMEM, ex.what());
    throw;
  }
  for (auto& f : b.flows)
    if (f.rc != COOK_OK) return f.rc;
  return COOK_OK;
}

// entries per host of a k8s "gpus" / "disk" map column pair (cookmatch.h cook_offers.gpu_slots): 0 means 1
// The placement walk keeps one LDS byte per offer of the pool.  A pool in a lockstep chain runs the good-enough flavour of the kernels
// whenever ANY pool of its chain has good-enough < 1 (match_rounds_multi), so the table must leave room for segments in both.
void match_check_offer_count(cook_engine* e, unsigned M) {
  if (e->params.match_algo == 1) return;  // (the one-job-at-a-time sweep has no such table)
  if (std::min(resolve_wseg<true>(M), resolve_wseg<false>(M)) < MV_WSEG_MIN)
    e->fail(COOK_E_INVALID, "cook_match: too many offers in one pool for the placement walk's offer table (about 150 000)");
}
unsigned res_slots(cook_engine* e, uint32_t slots, const char* what) {
  if (slots > COOK_MAX_RES_SLOTS) e->fail(COOK_E_INVALID, std::string(what) + " > COOK_MAX_RES_SLOTS");
  return slots ? slots : 1u;
}

// read back `words` 64-bit words from d_scratch64 (synchronises the stream)
void readback64(cook_engine* e, unsigned words) {
  pinned_copy(e, e->h_scratch, e->d_scratch64.ptr(), words * 8, hipMemcpyDeviceToHost);
  sync(e);
}
void readback_counters(cook_engine* e, unsigned* out, unsigned words) {
  pinned_copy(e, e->h_scratch, e->d_counters.ptr(), words * 4, hipMemcpyDeviceToHost);
  sync(e);
  std::memcpy(out, e->h_scratch, words * 4);
}

// ---- segmented scan driver ------------------------------------------------------------------------------------
template <class T, class Load>
void seg_scan(cook_engine* e, const char* tag, Load load, const uint8_t* head, unsigned n, T* out, ScanTmp<T>& tmp) {
  if (n == 0) return;
  const unsigned nb = div_up(n, SS_TILE);
  tmp.agg.ensure(nb);
  tmp.carry.ensure(nb);
  tmp.first_head.ensure(nb);
  KM<seg_scan_local<T, Load>, SS_THREADS>(e, tag, nb, load, head, n, out, tmp.agg.ptr(), tmp.first_head.ptr());
  if (nb > 1 && nb <= (unsigned)SS_THREADS) {
    KM<seg_scan_propagate_fused<T>, SS_THREADS>(e, "seg_scan_propagate", nb, out, n, (const SegAgg<T>*)tmp.agg.ptr(), (const unsigned*)tmp.first_head.ptr());
  } else if (nb > 1) {
    KM<seg_scan_blocksums<T>, SS_THREADS>(e, "seg_scan_blocksums", 1, (const SegAgg<T>*)tmp.agg.ptr(), nb, tmp.carry.ptr());
    KM<seg_scan_propagate<T>, SS_THREADS>(e, "seg_scan_propagate", nb, out, n, (const SegAgg<T>*)tmp.carry.ptr(), (const unsigned*)tmp.first_head.ptr());
  }
}

// ---- radix sort driver: one stable pass of `perm` by the 8 key bits from `shift` up --------------------------------------
template <int IPL>
static void radix_pass_t(cook_engine* e, const uint64_t* key, const uint32_t* in, uint32_t* out, unsigned n, unsigned shift, bool fused) {
  const unsigned nb = div_up(n, rs_tile(IPL));
  e->hist.ensure((size_t)256 * nb);
  KM<radix_hist<IPL>, RS_THREADS>(e, "radix_hist", nb, key, in, n, shift, nb, fused ? 1u : 0u, e->hist.ptr());
  if (!fused) KM<excl_scan_u32_single, SCAN1_THREADS>(e, "radix_scan", 1, e->hist.ptr(), 256u * nb, (uint32_t*)nullptr);
  KM<radix_scatter<IPL>, RS_THREADS>(e, "radix_scatter", nb, key, in, out, n, shift, nb, fused ? 1u : 0u, (const uint32_t*)e->hist.ptr());
}
void radix_pass(cook_engine* e, const uint64_t* key, const uint32_t* in, uint32_t* out, unsigned n, unsigned shift) {
  if (div_up(n, rs_tile(RS_IPL_SMALL)) <= RS_FUSED_BLOCKS) radix_pass_t<RS_IPL_SMALL>(e, key, in, out, n, shift, true);
  else radix_pass_t<RS_IPL_LARGE>(e, key, in, out, n, shift, div_up(n, rs_tile(RS_IPL_LARGE)) <= RS_FUSED_BLOCKS_LARGE);
}
// sort by the bits of `key` selected by `mask` (bits that vary); ping-pongs between a and b; returns final buffer.  A digit starts at
// the lowest varying bit not sorted yet (bits that never vary in between cost nothing).
uint32_t* radix_sort_masked(cook_engine* e, const uint64_t* key, unsigned long long mask, const uint32_t* cur, uint32_t* a,
                            uint32_t* b, unsigned n) {
  const uint32_t* in = cur;  // null: the identity (the first pass reads positions instead of a permutation)
  uint32_t* last = const_cast<uint32_t*>(cur);
  while (mask) {
    const unsigned shift = (unsigned)__builtin_ctzll(mask);
    uint32_t* out = (in == a) ? b : a;
    radix_pass(e, key, in, out, n, shift);
    in = out;
    last = out;
    mask = shift + 8 >= 64 ? 0ull : mask & ~((1ull << (shift + 8)) - 1ull);
  }
  return last;
}

COOK_KERNEL void fill_i32(int32_t* p, unsigned n, int32_t v) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
COOK_KERNEL void iota_u32(uint32_t* p, unsigned n) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

// =================================================================================================================
// RANK
// =================================================================================================================
void rank_stage(cook_engine* e, const cook_tasks* t, const cook_users* u) {
  if (!t || !u) e->fail(COOK_E_INVALID, "cook_rank_stage: null tasks/users");
  const unsigned N = t->n, U = u->n;
  if (N && (!t->cpus || !t->mem || !t->user || !t->priority || !t->start_ms || !t->task_id || !t->job_id || !t->pending))
    e->fail(COOK_E_INVALID, "cook_rank_stage: a required task array is NULL");
  if (U == 0 && N) e->fail(COOK_E_INVALID, "cook_rank_stage: no users");
  unsigned np = 0;
  std::vector<uint32_t> pend_ord(N ? N : 1);
  for (unsigned i = 0; i < N; ++i) {
    if (t->user[i] >= U) e->fail(COOK_E_INVALID, "cook_rank_stage: user id out of range");
    pend_ord[i] = np;
    np += t->pending[i] ? 1u : 0u;
  }
  // (nothing of the previous table counts from here on: a stage that fails half-way must not leave its usage or its rank behind)
  e->rank_staged = false, e->pool_usage_known = false, e->rank_done = false;
  e->N = N;
  e->U = U;
  e->n_pending = np;
  e->has_gpus = t->gpus != nullptr;
  h2d(e, e->t_cpus, t->cpus, N);
  h2d(e, e->t_mem, t->mem, N);
  if (t->gpus) h2d(e, e->t_gpus, t->gpus, N);
  h2d(e, e->t_user, t->user, N);
  h2d(e, e->t_prio, t->priority, N);
  h2d(e, e->t_start, t->start_ms, N);
  h2d(e, e->t_task, t->task_id, N);
  h2d(e, e->t_job, t->job_id, N);
  h2d(e, e->t_pending, t->pending, N);
  h2d(e, e->pend_ord, pend_ord.data(), N);
  h2d(e, e->u_divc, u->div_cpus, U);
  h2d(e, e->u_divm, u->div_mem, U);
  h2d(e, e->u_divg, u->div_gpus, U);
  h2d(e, e->u_qcount, u->quota_count, U);
  h2d(e, e->u_qcpus, u->quota_cpus, U);
  h2d(e, e->u_qmem, u->quota_mem, U);
  h2d(e, e->u_qgpus, u->quota_gpus, U);
  sync(e);  // pend_ord is a host temporary
  e->rank_staged = true;
  e->pool_usage_known = false;
  e->rank_done = false;
}

void rank_pool_usage(cook_engine* e, cook_usage* out) {
  if (!e->rank_staged) e->fail(COOK_E_STATE, "cook_rank_pool_usage before cook_rank_stage");
  // (COOK_POOL_USAGE_MEMO=0: sum it every time — bench.py's timed cycles keep one table resident, a live cycle arrives with a new one)
  static const bool memo = [] { const char* v = std::getenv("COOK_POOL_USAGE_MEMO"); return !(v && v[0] == '0'); }();
  if (e->pool_usage_known && memo) {  // summed for this very table already (cook_rank_stage / cook_cycle_update forget it)
    *out = e->pool_usage_val;
    return;
  }
  e->pool_usage.ensure(1);
  if (e->N == 0) {
    *out = cook_usage{0, 0, 0, 0};
    return;
  }
  e->pool_usage.ensure(1 + POOL_USAGE_BLOCKS);
  KM<pool_usage_partial, 256>(e, "pool_usage_partial", POOL_USAGE_BLOCKS, (const double*)e->t_cpus.ptr(), (const double*)e->t_mem.ptr(),
      e->has_gpus ? (const double*)e->t_gpus.ptr() : (const double*)nullptr, (const uint8_t*)e->t_pending.ptr(), e->N, e->pool_usage.ptr() + 1,
      (unsigned)POOL_USAGE_BLOCKS);
  KM<pool_usage_reduce, COOK_WAVE>(e, "pool_usage_reduce", 1, (const double*)e->t_cpus.ptr(), (const double*)e->t_mem.ptr(),
      e->has_gpus ? (const double*)e->t_gpus.ptr() : (const double*)nullptr, (const uint8_t*)e->t_pending.ptr(), e->N,
      (const SumU4*)(e->pool_usage.ptr() + 1), (unsigned)POOL_USAGE_BLOCKS, e->pool_usage.ptr());
  SumU4 h;
  pinned_copy(e, e->h_scratch, e->pool_usage.ptr(), sizeof(SumU4), hipMemcpyDeviceToHost);
  sync(e);
  std::memcpy(&h, e->h_scratch, sizeof(SumU4));
  *out = cook_usage{h.count, h.cpus, h.mem, h.gpus};
  e->pool_usage_val = *out;
  e->pool_usage_known = true;
}

// per-user running usage [U x 3] of the pool, from the per-user order of the last rank run (rank_kernels.hpp)
void rank_user_usage(cook_engine* e, double* out, bool out_is_device) {
  if (!e->rank_done) e->fail(COOK_E_STATE, "cook_rank_user_usage before cook_rank_run");
  if (!out) e->fail(COOK_E_INVALID, "cook_rank_user_usage: null output");
  const unsigned N = e->N, U = e->U;
  if (U == 0) return;
  double* dst = out_is_device ? out : e->uu_out.ensure((size_t)U * 3);
  if (N) {
    SumU4* rp = e->uu_pre.ensure(N);
    seg_scan<SumU4>(e, "user_running_scan", LoadRunningU4{e->s_use.ptr(), e->s_pending.ptr()}, (const uint8_t*)e->head.ptr(), N, rp,
                    e->tmpU4);
    KM<user_usage_extract, 256>(e, "user_usage_extract", div_up(U, 256), (const SumU4*)rp, (const SumU4*)e->s_use.ptr(),
        (const uint8_t*)e->s_pending.ptr(), (const uint32_t*)e->seg_start.ptr(), (const uint32_t*)e->seg_end.ptr(), U, dst);
  } else {
    memset_async(e, dst, 0, (size_t)U * 24);
  }
  if (!out_is_device) copy_async(e, out, dst, (size_t)U * 24, hipMemcpyDeviceToHost);
  sync(e);
}

// one quota filter stage over the queue (tools.clj:917-933); returns new queue length
unsigned queue_filter_quota(cook_engine* e, unsigned stage, unsigned len, const cook_usage& quota, const cook_usage& base, uint32_t*& qitem,
                            SumU4*& quse, uint32_t*& qitem_other, SumU4*& quse_other) {
  if (len == 0) return 0;
  e->qpre.ensure(len);
  e->iflag.ensure(len);
  e->scanI.ensure(len);
  LoadQueueUse ld{quse, SumU4{base.count, base.cpus, base.mem, base.gpus, 0u}};
  seg_scan<SumU4>(e, "queue_usage_scan", ld, (const uint8_t*)nullptr, len, e->qpre.ptr(), e->tmpU4);
  // [32 + 2 * stage]: a prefix rounded, [33 + 2 * stage]: the new length.  Stages 0 / 1 are rank_run's (zeroed by rank_init), stage 2 is
  // the considerable filters' (which may run without a rank before them: cleared here)
  unsigned* any_bad = e->d_counters.ptr() + 32 + 2 * stage;
  if (stage >= 2) memset_async(e, any_bad, 0, 8);
  Usage4 q{quota.count, quota.cpus, quota.mem, quota.gpus};
  KM<queue_quota_flag, 256>(e, "queue_quota_flag", div_up(len, 256), (const SumU4*)e->qpre.ptr(), len, q, e->iflag.ptr(), any_bad);
  KM<queue_quota_fix, 64>(e, "queue_quota_fix", 1, (const SumU4*)quse, len, SumU4{base.count, base.cpus, base.mem, base.gpus, 0u}, q,
      (const unsigned*)any_bad, e->iflag.ptr());
  seg_scan<SumI>(e, "queue_compact_scan", LoadI{e->iflag.ptr()}, (const uint8_t*)nullptr, len, e->scanI.ptr(), e->tmpI);
  unsigned* len_out = any_bad + 1;
  KM<queue_compact, 256>(e, "queue_compact", div_up(len, 256), (const uint32_t*)qitem, (const SumU4*)quse, (const int*)e->iflag.ptr(),
      (const SumI*)e->scanI.ptr(), len, qitem_other, quse_other, len_out);
  unsigned h[2];
  pinned_copy(e, e->h_scratch, len_out, 4, hipMemcpyDeviceToHost);
  sync(e);
  std::memcpy(h, e->h_scratch, 4);
  std::swap(qitem, qitem_other);
  std::swap(quse, quse_other);
  return h[0];
}

void rank_run(cook_engine* e) {
  if (!e->rank_staged) e->fail(COOK_E_STATE, "cook_rank_run before cook_rank_stage");
  const auto t_call = std::chrono::steady_clock::now();
  if (g_sync_trace) tl_sync_ms = 0.0, tl_syncs = 0;
  const unsigned N = e->N, U = e->U;
  e->n_ranked = 0;
  e->rank_done = false;
  e->ranked.ensure(std::max(1u, e->n_pending));
  if (N == 0) {
    e->rank_done = true;
    return;
  }
  const unsigned gN = div_up(N, 256);
  e->d_scratch64.ensure(64);
  e->d_counters.ensure(64);
  // --- per-user order keys -------------------------------------------------------------------------------
  const bool radix_only = std::getenv("COOK_RANK_RADIX") != nullptr;  // the tie rule as radix passes (the tests run both forms)
  unsigned long long* mins = e->d_scratch64.ptr();      // [0..2]
  unsigned long long* same = e->d_scratch64.ptr() + 4;  // [4..6] bits on which all keys of a word agree
  e->seg_start.ensure(U);
  e->seg_end.ensure(U);
  e->inexact_user.ensure(U);
  TieCtl* tie_ctl0 = e->tie_ctl.ensure(1);
  bool tie_ctl_clean = true;  // until the first refinement has used it
  KM<rank_init, 256>(e, "rank_init", std::max(1u, std::min(div_up(U, 256), 64u)), e->d_scratch64.ptr(), e->d_counters.ptr(), 40u,
      e->inexact_user.ptr(), e->seg_end.ptr(), U, reinterpret_cast<unsigned*>(tie_ctl0), (unsigned)(sizeof(TieCtl) / 4), std::max(1u,
      std::min(div_up(U, 256), 64u)));
  e->w0.ensure(N);
  e->w1.ensure(N);
  e->w2.ensure(N);
  KM<rank_key_mins, 256>(e, "rank_key_mins", std::min(gN, 64u), (const int64_t*)e->t_start.ptr(), (const int64_t*)e->t_task.ptr(),
      (const int64_t*)e->t_job.ptr(), (const uint8_t*)e->t_pending.ptr(), N, mins, std::min(gN, 64u));
  KM<rank_build_keys, 256>(e, "rank_build_keys", gN, (const uint32_t*)e->t_user.ptr(), (const int32_t*)e->t_prio.ptr(),
      (const int64_t*)e->t_start.ptr(), (const int64_t*)e->t_task.ptr(), (const int64_t*)e->t_job.ptr(), (const uint8_t*)e->t_pending.ptr(), N,
      (const unsigned long long*)mins, e->w0.ptr(), e->w1.ptr(), e->w2.ptr(), same);
  readback64(e, 8);
  const unsigned long long mk0 = ~e->h_scratch[4], mk1 = ~e->h_scratch[5], mk2 = ~e->h_scratch[6];
  e->permA.ensure(N);
  e->permB2.ensure(N);
  const uint32_t* cur = nullptr;  // the identity
  cur = radix_sort_masked(e, e->w2.ptr(), mk2, cur, e->permA.ptr(), e->permB2.ptr(), N);
  cur = radix_sort_masked(e, e->w1.ptr(), mk1, cur, e->permA.ptr(), e->permB2.ptr(), N);
  cur = radix_sort_masked(e, e->w0.ptr(), mk0, cur, e->permA.ptr(), e->permB2.ptr(), N);
  if (!cur) {  // every task has the same key words
    KM<iota_u32, 256>(e, "iota", gN, e->permA.ptr(), N);
    cur = e->permA.ptr();
  }
  e->permB = const_cast<uint32_t*>(cur);
  unsigned n_kept = 0;
  unsigned long long vor = 0, vand = 0;
  unsigned* counters = e->d_counters.ptr();  // [0] n_kept [1] equal-run [2] n_tied
  // --- gather, per-user prefix sums ----------------------------------------------------------------------
  e->s_user.ensure(N);
  e->s_use.ensure(N);
  e->s_pending.ensure(N);
  e->head.ensure(N);
  e->seg_start.ensure(U);
  e->seg_end.ensure(U);
  e->pre.ensure(N);
  KM<rank_gather, 256>(e, "rank_gather", gN, (const uint32_t*)e->permB, N, (const uint32_t*)e->t_user.ptr(), (const double*)e->t_cpus.ptr(),
      (const double*)e->t_mem.ptr(), e->has_gpus ? (const double*)e->t_gpus.ptr() : (const double*)nullptr, (const uint8_t*)e->t_pending.ptr(),
      e->s_user.ptr(), e->s_use.ptr(), e->s_pending.ptr(), e->head.ptr(), e->seg_start.ptr(), e->seg_end.ptr());
  seg_scan<SumU4>(e, "user_usage_scan", LoadU4{e->s_use.ptr()}, (const uint8_t*)e->head.ptr(), N, e->pre.ptr(), e->tmpU4);
  KM<rank_mark_inexact, 256>(e, "rank_mark_inexact", gN, (const SumU4*)e->pre.ptr(), (const uint32_t*)e->s_user.ptr(), N, e->inexact_user.ptr());
  KM<rank_fix_inexact, 256>(e, "rank_fix_inexact", div_up(U, 256), (const SumU4*)e->s_use.ptr(), e->pre.ptr(), (const uint32_t*)e->seg_start.ptr(),
      (const uint32_t*)e->seg_end.ptr(), (const uint32_t*)e->inexact_user.ptr(), U);
  // --- limiter + DRU ---------------------------------------------------------------------------------------
  e->iflag.ensure(N);
  e->scanI.ensure(N);
  KM<rank_over_flag, 256>(e, "rank_over_flag", gN, (const SumU4*)e->pre.ptr(), (const uint32_t*)e->s_user.ptr(), N, (const double*)e->u_qcount.ptr(),
      (const double*)e->u_qcpus.ptr(), (const double*)e->u_qmem.ptr(), (const double*)e->u_qgpus.ptr(), e->iflag.ptr());
  seg_scan<SumI>(e, "over_quota_scan", LoadI{e->iflag.ptr()}, (const uint8_t*)e->head.ptr(), N, e->scanI.ptr(), e->tmpI);
  e->dru.ensure(N);
  e->dkey.ensure(N);
  e->keep.ensure(N);
  unsigned long long* orand = reinterpret_cast<unsigned long long*>(counters + 8);  // [0] OR of the kept keys, [1] OR of their complements
  KM<rank_score, 256>(e, "rank_score", gN, (const SumU4*)e->pre.ptr(), (const SumI*)e->scanI.ptr(), (const uint32_t*)e->s_user.ptr(), N,
      (int)e->params.max_over_quota_jobs, (int)e->params.dru_mode, (const double*)e->u_divc.ptr(), (const double*)e->u_divm.ptr(),
      (const double*)e->u_divg.ptr(), e->dru.ptr(), e->dkey.ptr(), e->keep.ptr(), counters, orand);
  pinned_copy(e, e->h_scratch, counters, 12 * 4, hipMemcpyDeviceToHost);  // the counts and, behind them, the two key words
  sync(e);
  vor = e->h_scratch[4], vand = ~e->h_scratch[5];
  unsigned hc[2];
  std::memcpy(hc, e->h_scratch, 8);
  n_kept = hc[0];
  // --- global DRU order -------------------------------------------------------------------------------------
  e->permC1.ensure(N);
  e->permC2.ensure(N);
  uint32_t* pc = nullptr;  // the identity
  if (n_kept) pc = radix_sort_masked(e, e->dkey.ptr(), vor & ~vand, pc, e->permC1.ptr(), e->permC2.ptr(), N);
  if (n_kept < N) {  // limiter dropped tasks: one extra 1-bit pass moves them behind every kept task
    e->nkkey.ensure(N);
    KM<rank_notkept_key, 256>(e, "rank_notkept_key", gN, (const uint8_t*)e->keep.ptr(), N, e->nkkey.ptr());
    pc = radix_sort_masked(e, e->nkkey.ptr(), 1ull, pc, e->permC1.ptr(), e->permC2.ptr(), N);
  }
  if (!pc) {  // all kept keys equal
    KM<iota_u32, 256>(e, "iota", gN, e->permC1.ptr(), N);
    pc = e->permC1.ptr();
  }
  e->permC = pc;
  unsigned qlen = 0;
  uint32_t* qitem = e->qitemA.ensure(std::max(1u, e->n_pending));
  uint32_t* qitem_o = e->qitemB.ensure(std::max(1u, e->n_pending));
  SumU4* quse = e->quseA.ensure(std::max(1u, e->n_pending));
  SumU4* quse_o = e->quseB.ensure(std::max(1u, e->n_pending));
  if (n_kept) {
    const unsigned gK = div_up(n_kept, 256);
    // --- tie groups + sorted-merge tie rule (prefix doubling) ------------------------------------------------
    unsigned bits = 1;
    while ((1ull << bits) <= (unsigned long long)U + N) ++bits;  // rank values <= U + N
    // composite key of a tied item = (start of its group, secondary rank), the two packed back to back: 2 * bits key bits, 36 for a
    // pool's 175k tasks = 5 radix passes (two 32-bit halves cost a pass more)
    const unsigned long long cmask = 2 * bits >= 64 ? ~0ull : (1ull << (2 * bits)) - 1ull;
    // refines `perm` (nk items of an index space with n_items items, per-user lists contiguous) in place; returns false when a
    // user has consecutive items with equal keys (the caller collapses those runs and calls again on the collapsed space)
    auto tie_refine_radix = [&](uint32_t* perm, const uint64_t* key, const uint32_t* user_of, const uint32_t* seg_first, unsigned nk,
                                unsigned n_items) -> bool {
      const unsigned gK = div_up(nk, 256);
      e->thead.ensure(nk);
      e->rank_of_item.ensure(n_items);
      e->gstart.ensure(nk);
      int* ones = e->ones_buf.ensure(nk);
      int* tied = e->tied_buf.ensure(nk);
      e->scanI.ensure(nk);
      memset_async(e, counters + 1, 0, 4);
      KM<tie_heads, 256>(e, "tie_heads", gK, (const uint32_t*)perm, key, nk, user_of, e->thead.ptr(), (uint8_t*)nullptr, ones, counters + 1);
      for (int round = 0;; ++round) {
        seg_scan<SumI>(e, "tie_group_scan", LoadI{ones}, (const uint8_t*)e->thead.ptr(), nk, e->scanI.ptr(), e->tmpI);
        memset_async(e, counters + 2, 0, 4);
        KM<tie_assign, 256>(e, "tie_assign", gK, (const uint32_t*)perm, (const uint8_t*)e->thead.ptr(), (const SumI*)e->scanI.ptr(), nk, U,
            e->rank_of_item.ptr(), e->gstart.ptr(), tied, counters + 2);
        unsigned h3[3];
        readback_counters(e, h3, 3);
        if (h3[1]) return false;
        const unsigned n_tied = h3[2];
        if (std::getenv("COOK_TIE_TRACE")) std::fprintf(stderr, "tie round %d: %u tied of %u\n", round, n_tied, nk);
        if (n_tied == 0) break;
        if (round > 31) e->fail(COOK_E_INVALID, "cook_rank: tie refinement did not converge");
        // compact tied slots, sort them by (group start, secondary), write back, split groups
        e->tpos.ensure(n_tied);
        e->titem.ensure(n_tied);
        e->ckey.ensure(n_tied);
        e->tsorted.ensure(n_tied);
        e->tsorted2.ensure(n_tied);
        seg_scan<SumI>(e, "tie_compact_scan", LoadI{tied}, (const uint8_t*)nullptr, nk, e->scanI.ptr(), e->tmpI);
        KM<tie_build, 256>(e, "tie_build", gK, (const uint32_t*)perm, (const int*)tied, (const SumI*)e->scanI.ptr(),
            (const uint32_t*)e->gstart.ptr(), nk, U, n_items, round, bits, (const uint32_t*)e->rank_of_item.ptr(), user_of, seg_first, e->tpos.ptr(),
            e->titem.ptr(), e->ckey.ptr());
        KM<iota_u32, 256>(e, "iota", div_up(n_tied, 256), e->tsorted.ptr(), n_tied);
        uint32_t* ts = radix_sort_masked(e, e->ckey.ptr(), cmask, e->tsorted.ptr(), e->tsorted.ptr(), e->tsorted2.ptr(), n_tied);
        KM<tie_writeback, 256>(e, "tie_writeback", div_up(n_tied, 256), (const uint32_t*)ts, (const uint32_t*)e->tpos.ptr(),
            (const uint32_t*)e->titem.ptr(), (const uint64_t*)e->ckey.ptr(), n_tied, perm, e->thead.ptr());
      }
      return true;
    };
    // the same refinement with the groups sorted in LDS tiles (tile_sort.hpp): two launches per doubling round, four rounds enqueued
    // per look at the counters (rounds past the last one exit at once); a tie group too long for a tile sends the call to the radix form,
    // which starts over from the keys (the order inside a group of equal keys is free when the refinement starts)
    auto tie_refine = [&](uint32_t* perm, const uint64_t* key, const uint32_t* user_of, const uint32_t* seg_first, unsigned nk,
                          unsigned n_items) -> bool {
      if (radix_only) return tie_refine_radix(perm, key, user_of, seg_first, nk, n_items);
      const unsigned gK = div_up(nk, 256);
      e->thead.ensure(nk);
      e->dhead.ensure(nk);
      e->rank_of_item.ensure(n_items);
      TieCtl* ctl = tie_ctl0;
      if (!tie_ctl_clean) memset_async(e, ctl, 0, sizeof(TieCtl));  // (rank_init cleared it for the first refinement)
      tie_ctl_clean = false;
      KM<tie_heads, 256>(e, "tie_heads", gK, (const uint32_t*)perm, key, nk, user_of, e->thead.ptr(), e->dhead.ptr(), (int*)nullptr, &ctl->equal_runs);
      constexpr int LOOK = 4;
      for (int r0 = 0; r0 < 32; r0 += LOOK) {
        for (int round = r0; round < r0 + LOOK; ++round) {
          KM<tie_rank_assign, 256>(e, "tie_rank_assign", gK, (const uint32_t*)perm, (const uint8_t*)e->thead.ptr(), nk, U, round, (const TieCtl*)ctl,
              e->rank_of_item.ptr());
          KM<tie_sort_tiles, TS_THREADS>(e, "tie_sort_tiles", div_up(nk, TS_NOMINAL), perm, e->thead.ptr(), (const uint8_t*)e->dhead.ptr(), nk, U,
              n_items, round, (const uint32_t*)e->rank_of_item.ptr(), user_of, seg_first, ctl);
        }
        TieCtl h;
        pinned_copy(e, e->h_scratch, ctl, sizeof(TieCtl), hipMemcpyDeviceToHost);
        sync(e);
        std::memcpy(&h, e->h_scratch, sizeof(TieCtl));
        if (std::getenv("COOK_TIE_TRACE"))
          for (int round = r0; round < r0 + LOOK; ++round) std::fprintf(stderr, "tie round %d: %u tied after, of %u\n", round, h.tied_after[round], nk);
        if (h.equal_runs) return false;
        if (h.overflow) return tie_refine_radix(perm, key, user_of, seg_first, nk, n_items);
        if (h.tied_after[r0 + LOOK - 1] == 0) return true;
      }
      e->fail(COOK_E_INVALID, "cook_rank: tie refinement did not converge");
      return false;
    };
    if (!tie_refine(e->permC, e->dkey.ptr(), e->s_user.ptr(), e->seg_start.ptr(), n_kept, N)) {
      // some user has a run of equal DRUs (a zero-resource task, a gpu-less task in gpu mode, a request absorbed by the sum): the
      // merge emits such a run back to back (rank_kernels.hpp, run_*), so collapse the runs, refine the heads, re-insert the rest
      int* isf = e->run_isf.ensure(N);
      int* nonf = e->run_nonf.ensure(N);
      SumI* nonf_incl = e->run_scan.ensure(N);
      KM<run_follower_flag, 256>(e, "run_follower_flag", gN, (const uint32_t*)e->s_user.ptr(), (const uint64_t*)e->dkey.ptr(),
          (const uint8_t*)e->keep.ptr(), N, isf, nonf);
      seg_scan<SumI>(e, "run_scan", LoadI{nonf}, (const uint8_t*)nullptr, N, nonf_incl, e->tmpI);
      pinned_copy(e, e->h_scratch, &nonf_incl[N - 1], 4, hipMemcpyDeviceToHost);
      sync(e);
      int n2i = 0;
      std::memcpy(&n2i, e->h_scratch, 4);
      const unsigned N2 = (unsigned)n2i, n_kept2 = n_kept - (N - N2);  // followers are kept items
      uint32_t* c_user = e->run_user.ensure(N2);
      uint64_t* c_dkey = e->run_dkey.ensure(N2);
      uint32_t* c_orig = e->run_orig.ensure(N2 + 1);
      uint32_t* c_seg = e->run_seg.ensure(U);
      uint32_t* b_to_c = e->run_b2c.ensure(N);
      KM<run_compact_items, 256>(e, "run_compact_items", gN, (const int*)nonf, (const SumI*)nonf_incl, N, (const uint32_t*)e->s_user.ptr(),
          (const uint64_t*)e->dkey.ptr(), (const uint8_t*)e->head.ptr(), c_user, c_dkey, c_orig, c_seg, b_to_c);
      KM<run_compact_sentinel, 1>(e, "run_compact_sentinel", 1, (const SumI*)nonf_incl, N, c_orig);
      int* posf = e->run_posf.ensure(n_kept);
      SumI* posf_incl = e->run_scan2.ensure(n_kept);
      KM<run_flag_positions, 256>(e, "run_flag_positions", gK, (const uint32_t*)e->permC, (const int*)isf, n_kept, posf);
      seg_scan<SumI>(e, "run_scan", LoadI{posf}, (const uint8_t*)nullptr, n_kept, posf_incl, e->tmpI);
      uint32_t* perm2 = e->run_perm.ensure(n_kept2);
      KM<run_compact_positions, 256>(e, "run_compact_positions", gK, (const uint32_t*)e->permC, (const int*)posf, (const SumI*)posf_incl, n_kept,
          (const uint32_t*)b_to_c, perm2);
      if (!tie_refine(perm2, c_dkey, c_user, c_seg, n_kept2, N2)) e->fail(COOK_E_STATE, "cook_rank: equal-DRU runs survived the collapse");
      const unsigned gK2 = div_up(n_kept2, 256);
      KM<run_count_followers, 256>(e, "run_count_followers", gK2, (const uint32_t*)perm2, (const uint32_t*)c_orig, n_kept2, posf);
      seg_scan<SumI>(e, "run_scan", LoadI{posf}, (const uint8_t*)nullptr, n_kept2, posf_incl, e->tmpI);
      KM<run_expand, 256>(e, "run_expand", gK2, (const uint32_t*)perm2, (const uint32_t*)c_orig, (const int*)posf, (const SumI*)posf_incl, n_kept2, e->permC);
    }
    // --- queue of pending jobs in rank order ---------------------------------------------------------------
    int* flag = e->iflag.ptr();
    KM<queue_flag_pending, 256>(e, "queue_flag_pending", gK, (const uint32_t*)e->permC, (const uint8_t*)e->s_pending.ptr(), n_kept, flag);
    seg_scan<SumI>(e, "queue_pending_scan", LoadI{flag}, (const uint8_t*)nullptr, n_kept, e->scanI.ptr(), e->tmpI);
    unsigned* dq = e->d_counters.ptr() + 38;  // (zeroed by rank_init)
    KM<queue_compact_pending, 256>(e, "queue_compact_pending", gK, (const uint32_t*)e->permC, (const int*)flag, (const SumI*)e->scanI.ptr(), n_kept,
        (const SumU4*)e->s_use.ptr(), qitem, quse, dq);
    pinned_copy(e, e->h_scratch, dq, 4, hipMemcpyDeviceToHost);
    sync(e);
    std::memcpy(&qlen, e->h_scratch, 4);
  }
  // --- quota filters (scheduler.clj:2134-2157) -------------------------------------------------------------
  if (qlen && e->quota.has_pool_quota) {
    cook_usage base = e->quota.pool_usage;
    if (!e->quota.pool_usage_given) rank_pool_usage(e, &base);
    qlen = queue_filter_quota(e, 0, qlen, e->quota.pool_quota, base, qitem, quse, qitem_o, quse_o);
  }
  if (qlen && e->quota.has_group_quota)
    qlen = queue_filter_quota(e, 1, qlen, e->quota.group_quota, e->quota.group_usage, qitem, quse, qitem_o, quse_o);
  // --- offensive filter (scheduler.clj:2198-2229) -----------------------------------------------------------
  const bool offensive_on = std::isfinite(e->params.offensive_max_mem_mb) || std::isfinite(e->params.offensive_max_cpus);
  if (qlen && offensive_on) {
    e->iflag.ensure(qlen);
    e->scanI.ensure(qlen);
    KM<queue_offensive_flag, 256>(e, "queue_offensive_flag", div_up(qlen, 256), (const SumU4*)quse, qlen, e->params.offensive_max_mem_mb,
        e->params.offensive_max_cpus, e->iflag.ptr());
    seg_scan<SumI>(e, "queue_compact_scan", LoadI{e->iflag.ptr()}, (const uint8_t*)nullptr, qlen, e->scanI.ptr(), e->tmpI);
    unsigned* len_out = e->d_counters.ptr() + 9;
    KM<queue_compact, 256>(e, "queue_compact", div_up(qlen, 256), (const uint32_t*)qitem, (const SumU4*)quse, (const int*)e->iflag.ptr(),
        (const SumI*)e->scanI.ptr(), qlen, qitem_o, quse_o, len_out);
    pinned_copy(e, e->h_scratch, len_out, 4, hipMemcpyDeviceToHost);
    sync(e);
    std::memcpy(&qlen, e->h_scratch, 4);
    std::swap(qitem, qitem_o);
    std::swap(quse, quse_o);
  }
  if (qlen)
    KM<queue_emit, 256>(e, "queue_emit", div_up(qlen, 256), (const uint32_t*)qitem, qlen, (const uint32_t*)e->permB, e->ranked.ptr());
  e->n_ranked = qlen;
  e->rank_done = true;
  if (g_sync_trace) {
    std::fprintf(stderr, "cook_rank_run: %u stream synchronisations, %.3f ms waiting in them, %.3f ms in the call\n", tl_syncs, tl_sync_ms,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    tl_sync_ms = 0.0, tl_syncs = 0;
  }
}

void rank_fetch(cook_engine* e, uint32_t* ranked, uint32_t* n_out, double* dru_of_task) {
  if (!e->rank_done) e->fail(COOK_E_STATE, "cook_rank_fetch before cook_rank_run");
  if (n_out) *n_out = e->n_ranked;
  if (ranked && e->n_ranked)
    copy_async(e, ranked, e->ranked.ptr(), (size_t)e->n_ranked * 4, hipMemcpyDeviceToHost);
  if (dru_of_task && e->N) {
    e->dru_out.ensure(e->N);
    KM<dru_to_task_space, 256>(e, "dru_to_task_space", div_up(e->N, 256), (const double*)e->dru.ptr(), (const uint8_t*)e->keep.ptr(),
        (const uint32_t*)e->permB, e->N, e->dru_out.ptr());
    copy_async(e, dru_of_task, e->dru_out.ptr(), (size_t)e->N * 8, hipMemcpyDeviceToHost);
  }
  sync(e);
}

// =================================================================================================================
// MATCH
// =================================================================================================================
// the offer columns of a staged match (cook_match_stage / cook_cycle_stage / cook_cycle_update)
void match_stage_offers(cook_engine* e, const cook_offers* o, bool offers_dev) {
  MatchIn& in = e->min;
  const unsigned M = o->n;
  if (M && (!o->cpus || !o->mem || !o->host)) e->fail(COOK_E_INVALID, "cook_match_stage: offers need cpus, mem and host");
  in.M = M;
  e->M = M;
  in.o_cpus = (offers_dev ? o->cpus : h2d_opt(e, e->o_cpus, o->cpus, M));
  in.o_mem = (offers_dev ? o->mem : h2d_opt(e, e->o_mem, o->mem, M));
  in.o_host = (offers_dev ? o->host : h2d_opt(e, e->o_host, o->host, M));
  in.o_k8s = (offers_dev ? o->k8s : h2d_opt(e, e->o_k8s, o->k8s, M));
  in.gpu_slots = res_slots(e, o->gpu_slots, "cook_match_stage: gpu_slots");
  in.disk_slots = res_slots(e, o->disk_slots, "cook_match_stage: disk_slots");
  in.o_gpu_model = (offers_dev ? o->gpu_model : h2d_opt(e, e->o_gpu_model, o->gpu_model, (size_t)M * in.gpu_slots));
  in.o_gpu_count = (offers_dev ? o->gpu_count : h2d_opt(e, e->o_gpu_count, o->gpu_count, (size_t)M * in.gpu_slots));
  if (in.o_gpu_model && !in.o_gpu_count) e->fail(COOK_E_INVALID, "cook_match_stage: gpu_model without gpu_count");
  in.o_disk_type = (offers_dev ? o->disk_type : h2d_opt(e, e->o_disk_type, o->disk_type, (size_t)M * in.disk_slots));
  in.o_disk_space = (offers_dev ? o->disk_space : h2d_opt(e, e->o_disk_space, o->disk_space, (size_t)M * in.disk_slots));
  // ports / named scalars of the leases (offer.clj:57-73); jobs' names beyond the offers' columns find a total of 0
  if (o->scalars && o->n_scalars > COOK_MAX_SCALARS) e->fail(COOK_E_INVALID, "cook_match_stage: more than COOK_MAX_SCALARS named scalars");
  in.o_ports = (offers_dev ? o->ports : h2d_opt(e, e->o_ports, o->ports, M));
  for (unsigned sc = 0; sc < COOK_MAX_SCALARS; ++sc) {
    const double* col = (o->scalars && sc < o->n_scalars) ? o->scalars + (size_t)sc * M : nullptr;
    in.o_scal[sc] = (offers_dev ? col : h2d_opt(e, e->o_scal[sc], col, M));
  }
  in.n_attr = o->attr ? o->n_attr_keys : 0;
  in.o_attr = (offers_dev ? o->attr : h2d_opt(e, e->o_attr, o->attr, (size_t)M * in.n_attr));
  in.o_max_tasks = (offers_dev ? o->max_tasks : h2d_opt(e, e->o_max_tasks, o->max_tasks, M));
  in.o_num_tasks = (offers_dev ? o->num_tasks : h2d_opt(e, e->o_num_tasks, o->num_tasks, M));
  in.o_location = (offers_dev ? o->location : h2d_opt(e, e->o_location, o->location, M));
  in.o_host_start = (offers_dev ? o->host_start_s : h2d_opt(e, e->o_host_start, o->host_start_s, M));
  in.o_run_cpus = (offers_dev ? o->run_cpus : h2d_opt(e, e->o_run_cpus, o->run_cpus, M));
  in.o_run_mem = (offers_dev ? o->run_mem : h2d_opt(e, e->o_run_mem, o->run_mem, M));
  in.o_run_count = (offers_dev ? o->run_count : h2d_opt(e, e->o_run_count, o->run_count, M));
  // two offers on one host?  (offers built on the device are one per node: never)
  in.host_dup = 0;
  e->cf_max_host = 0xFFFFFFFFu;
  if (!offers_dev && M) {
    std::vector<uint32_t> hs(o->host, o->host + M);
    std::sort(hs.begin(), hs.end());
    in.host_dup = std::adjacent_find(hs.begin(), hs.end()) != hs.end() ? 1u : 0u;
    e->cf_max_host = hs.back();
  }
}
void match_stage_offers(cook_engine* e, const cook_offers* o) {
  match_stage_offers(e, o, false);
  sync(e);
}

// offers_dev: the pointers of `o` are DEVICE columns (the rows of cook_offers_run): used in place, nothing is copied
void match_stage_inputs(cook_engine* e, const cook_jobs* j, const cook_offers* o, const cook_groups* g,
                        const uint32_t* reserved_hosts, uint32_t n_reserved, bool offers_dev = false) {
  if (!j || !o) e->fail(COOK_E_INVALID, "cook_match_stage: null jobs/offers");
  const unsigned K = j->n, M = o->n, G = g ? g->n : 0;
  if (K && (!j->cpus || !j->mem)) e->fail(COOK_E_INVALID, "cook_match_stage: jobs need cpus and mem");
  if (M && (!o->cpus || !o->mem || !o->host)) e->fail(COOK_E_INVALID, "cook_match_stage: offers need cpus, mem and host");
  if (j->group && !g) {
    for (unsigned k = 0; k < K; ++k)
      if (j->group[k] != COOK_NONE_U32) e->fail(COOK_E_INVALID, "cook_match_stage: job has a group but no groups table given");
  }
  if (j->group && g)
    for (unsigned k = 0; k < K; ++k)
      if (j->group[k] != COOK_NONE_U32 && j->group[k] >= G) e->fail(COOK_E_INVALID, "cook_match_stage: group id out of range");
  MatchIn& in = e->min;
  std::memset(&in, 0, sizeof(in));
  in.K = K;
  in.M = M;
  in.G = G;
  e->Kjobs = K;
  e->cf_group_run_total = 0;
  in.j_cpus = h2d_opt(e, e->j_cpus, j->cpus, K);
  in.j_mem = h2d_opt(e, e->j_mem, j->mem, K);
  in.j_gpus = h2d_opt(e, e->j_gpus, j->gpus, K);
  in.j_gpu_model = h2d_opt(e, e->j_gpu_model, j->gpu_model, K);
  e->has_j_user = h2d_opt(e, e->j_user, j->user, K) != nullptr;
  in.j_group = h2d_opt(e, e->j_group, j->group, K);
  if (j->eq_off) {
    in.j_eq_off = h2d_opt(e, e->j_eq_off, j->eq_off, K + 1);
    const unsigned ne = K ? j->eq_off[K] : 0;
    in.j_eq_key = h2d_opt(e, e->j_eq_key, j->eq_key, std::max(1u, ne));
    in.j_eq_val = h2d_opt(e, e->j_eq_val, j->eq_val, std::max(1u, ne));
  }
  if (j->novel_off) {
    in.j_novel_off = h2d_opt(e, e->j_novel_off, j->novel_off, K + 1);
    const unsigned nn = K ? j->novel_off[K] : 0;
    in.j_novel_host = h2d_opt(e, e->j_novel_host, j->novel_host, std::max(1u, nn));
  }
  in.j_reserved_host = h2d_opt(e, e->j_reserved_host, j->reserved_host, K);
  in.j_ckpt = h2d_opt(e, e->j_ckpt, j->ckpt_location, K);
  in.j_est_end = h2d_opt(e, e->j_est_end, j->est_end_ms, K);
  in.j_disk_req = h2d_opt(e, e->j_disk_req, j->disk_request, K);
  in.j_disk_type = h2d_opt(e, e->j_disk_type, j->disk_type, K);
  if (in.j_disk_req && !in.j_disk_type) e->fail(COOK_E_INVALID, "cook_match_stage: disk_request without disk_type");
  // ports / named scalar requests (scheduler.clj:466, 177-189): has_x = some job asks for any
  if (j->scalars && j->n_scalars > COOK_MAX_SCALARS) e->fail(COOK_E_INVALID, "cook_match_stage: more than COOK_MAX_SCALARS named scalars");
  unsigned has_x = 0;
  in.j_ports = h2d_opt(e, e->j_ports, j->ports, K);
  if (j->ports)
    for (unsigned k = 0; k < K; ++k) {
      if (j->ports[k] < 0) e->fail(COOK_E_INVALID, "cook_match_stage: negative port count");
      has_x |= j->ports[k] > 0;
    }
  const unsigned n_scal = j->scalars ? j->n_scalars : 0u;
  for (unsigned sc = 0; sc < n_scal; ++sc) {
    const double* col = j->scalars + (size_t)sc * K;
    in.j_scal[sc] = h2d_opt(e, e->j_scal[sc], col, K);
    for (unsigned k = 0; k < K && !has_x; ++k) has_x = col[k] == col[k];
  }
  match_stage_offers(e, o, offers_dev);
  in.n_scal = n_scal;
  in.has_x = has_x;
  e->groups_simple = true;
  for (unsigned x = 0; x < G; ++x)
    if (g->type && g->type[x] >= 2) e->groups_simple = false;
  if (G) {
    in.g_type = h2d_opt(e, e->g_type, g->type, G);
    in.g_attr_key = h2d_opt(e, e->g_attr_key, g->attr_key, G);
    in.g_min = h2d_opt(e, e->g_min, g->minimum, G);
    if (!in.g_type || !in.g_attr_key || !in.g_min) e->fail(COOK_E_INVALID, "cook_match_stage: groups need type, attr_key, minimum");
    e->cf_group_run_total = g->run_off ? g->run_off[G] : 0u;
    if (g->run_off) {
      in.g_run_off = h2d_opt(e, e->g_run_off, g->run_off, G + 1);
      const unsigned nr = g->run_off[G];
      in.g_run_host = h2d_opt(e, e->g_run_host, g->run_host, std::max(1u, nr));
      in.g_run_attr = h2d_opt(e, e->g_run_attr, g->run_attr, std::max(1u, nr));
    }
  }
  std::vector<uint32_t> bits;
  if (n_reserved) {
    uint32_t mx = 0;
    for (unsigned i = 0; i < n_reserved; ++i) mx = std::max(mx, reserved_hosts[i]);
    bits.assign(mx / 32 + 1, 0u);
    for (unsigned i = 0; i < n_reserved; ++i) bits[reserved_hosts[i] >> 5] |= 1u << (reserved_hosts[i] & 31);
    in.reserved_bits = h2d_opt(e, e->reserved_bits, bits.data(), bits.size());
    in.reserved_words = (unsigned)bits.size();
  }
  in.good_enough = e->params.good_enough_fitness;
  in.host_lifetime_mins = e->params.host_lifetime_mins;
  sync(e);  // `bits` is a host temporary
  e->K = K;
  e->M = M;
  e->G = G;
  e->match_staged = true;
  e->match_done = false;
}

// engines alive per device: sizes the persistent placement kernel so that the kernels of all pools sharing a GPU are resident
static std::atomic<int> g_engines_on_device[64];

// the state a match call starts from, in ONE launch (nine memsets before round 5: each is a launch, and the set-up of a pool's match sits
// in the chain of small launches a cycle begins with): nothing assigned, no job placed, jmin = {max, max, 0, 0}
COOK_KERNEL void match_init_state_kernel(MatchState st, unsigned long long* __restrict__ jmin, unsigned K, unsigned M, unsigned G, unsigned nblk) {
  const unsigned stride = nblk * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += stride) {
    st.ac[i] = 0.0, st.am[i] = 0.0, st.acount[i] = 0;
    if (st.xports) {
      st.xports[i] = 0;
      for (unsigned s = 0; s < (unsigned)COOK_MAX_SCALARS; ++s) st.xscal[(size_t)s * M + i] = 0.0;
    }
  }
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < G; i += stride) st.group_last[i] = -1;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += stride) st.job_prev[i] = -1, st.job_to_offer[i] = -1;
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    st.summary[threadIdx.x] = 0u;
    jmin[threadIdx.x] = threadIdx.x < 2 ? 0x7F7F7F7F7F7F7F7Full : 0ull;  // [0..1] > every finite double's bit pattern; [2] a job with a
                                                                             // negative / non-finite request was seen
  }
}
void match_init_state(cook_engine* e, const MatchState& st, unsigned K, unsigned M, unsigned G) {
  const unsigned n = std::max(std::max(K, M), std::max(G, 1u));
  KM<match_init_state_kernel, 256>(e, "match_init_state", std::min(div_up(n, 256), 512u), st, e->m_jmin.ptr(), K, M, G, std::min(div_up(n, 256), 512u));
}

void match_finish_rounds(cook_engine* e, const MatchState& st, const V2Buf& vb, const WinCtl& hc, hipStream_t stream);
// good-enough-fitness < 1: the resolve kernel whose fast path knows the rule (COOK_GE_FAST=0: the general path decides every job, as
// before — kept for A/B measurements)
// rounds launched between two looks at the pools' progress.  The estimate comes from the rate of the last batch; when the cluster
// fills up in the middle of a batch the rest of the queue settles thousands of jobs per round and what is left of the batch are
// launches that exit at once (145 of a chain's 520 rounds at a cap of 256): cheap each, not free together.
static unsigned batch_cap() {
  static const unsigned cap = [] {
    const char* s = std::getenv("COOK_BATCH_CAP");
    const long v = s ? std::atol(s) : 0;
    return (unsigned)(v >= 2 ? v : 64);  // (256 / 96 / 48 / 24: eight pools 82.7 / 81.9 / 82.0 / 82.8 ms, one pool 57.5 / 57.1 / 56.9 / 56.6)
  }();
  return cap;
}
// COOK_PACK_ARGS=0: the multi-pool launches read their contexts from memory even when they would fit the kernel arguments (A/B switch)
static bool pack_args() {
  static const bool on = [] {
    const char* s = std::getenv("COOK_PACK_ARGS");
    return !(s && std::atoi(s) == 0);
  }();
  return on;
}
void match_rounds_multi(cook_engine** es, unsigned n);
// one round of launches on the engine's stream, for ONE pool: GE = the call runs with good-enough-fitness < 1
template <bool GE>
static void launch_round(cook_engine* e, const MatchIn& in, const MatchState& st, const V2Buf& vb) {
  KL("match_eval2", match_eval2<GE>, dim3(vb.C, MV_JG), COOK_WAVE * MV_EW, in, st, vb);
  KL("match_merge2", match_merge2<GE>, MV_MERGE_BLOCKS, COOK_WAVE * MV_MW, in, vb);
  KL("match_resolve2", match_resolve2<GE>, 1, MV_RTHREADS, st, vb);
}

// ---- class-ordered best fit (classfit.hpp): set-up, eligibility, launch --------------------------------------------------------------------
// match_algo 3 asks for the class-ordered best fit.  match_algo 0 (the engine's choice) takes it when six or more engines share the device: its walks need no
// evaluation launches, so eight pools cost what one costs (measured on MI355X, profiles/r06*: eight C4 pools 48.5 against 49.8 ms as served walkers, K = 1000 4.95
// against 5.11 ms), while a pool that has the GPU (nearly) to itself is faster in window rounds (one C4 pool 38.3 against 44.5 ms).  Measured per pool count (profiles/r06n_pools_5_6_7.txt,
// C4 pools, served walkers against class-ordered): 5 pools 45.4 / 46.1 ms, 6 pools 49.0 / 46.5, 7 pools 50.9 / 46.6, 8 pools 49.9 / 47.1: the rule turns at six.  COOK_CLASSFIT=1 / 0 forces / forbids
// it for match_algo 0.  (A plain function, not a namespace-scope lambda initialiser: hipcc gave the second such initialiser the body of the first)
static int classfit_env() {
  static const int v = [] {
    const char* s = std::getenv("COOK_CLASSFIT");
    return s && (s[0] == '0' || s[0] == '1') ? s[0] - '0' : -1;
  }();
  return v;
}
static bool classfit_by_default(const cook_engine* e) {
  const int f = classfit_env();
  if (f >= 0) return f == 1;
  return g_engines_on_device[e->device & 63].load() >= 6;
}
static size_t cf_lds_bytes_host(unsigned NP, unsigned M, bool eq, unsigned G, unsigned S) {  // the layout of cf_walk_pool (classfit_walk.hpp)
  size_t n = sizeof(CfFixed) + (size_t)NP * 10u;
  n = (n + 7u) & ~(size_t)7u;
  if (eq) n += (size_t)M * 8u;
  n += ((size_t)G + 1u) * 2u + (size_t)G * 2u + (size_t)S * 2u;
  return n + 64u;
}
// the three set-up kernels of a call and the look at what they found -> true: the call can be placed by cf_walk (ctx filled in)
bool cf_setup(cook_engine* e, const MatchIn& in, const MatchIn* in_dev, const MatchState& st, const JobRec* jr, const JobCons* jcons, const OfferA* oa, const OfferB* ob,
              CfPoolCtx& ctx) {
  const unsigned K = in.K, M = in.M, G = in.G;
  e->cf_inelig = 0x10000u;
  if (K == 0 || M == 0 || M > CF_SORT_N || G > CF_MAXG || in.good_enough < 1.0 || in.has_x || in.reserved_bits || in.host_dup) return false;
  if (e->cf_max_host == 0xFFFFFFFFu || (size_t)e->cf_max_host > 8u * (size_t)M + 65536u) return false;
  CfBuf b{};
  b.ctl = e->cf_ctl.ensure(1);
  b.jr = jr, b.jcons = jcons, b.oa = oa, b.ob = ob;
  b.attr8 = e->cf_attr8.ensure(M);
  b.max_host = e->cf_max_host;
  b.h2o = e->cf_h2o.ensure((size_t)b.max_host + 1u);
  b.pos_fc = e->cf_pos[0].ensure(M), b.pos_fm = e->cf_pos[1].ensure(M), b.pos_cid = e->cf_pos[2].ensure(M);
  b.scr_fc = e->cf_scr[0].ensure(M), b.scr_fm = e->cf_scr[1].ensure(M), b.scr_cid = e->cf_scr[2].ensure(M);
  b.jobs = e->cf_jobs.ensure(K);
  b.gcount = e->cf_gcount.ensure(std::max(1u, G));
  b.gmem = e->cf_gmem.ensure((size_t)std::max(1u, G) * CF_GMEM);
  static_assert(sizeof(CfCtl) % 4 == 0, "cf_init clears the control block word by word");
  KM<cf_init, 256>(e, "cf_init", std::min(div_up(b.max_host + 1u, 1024u), 512u), b, b.max_host + 1u, std::max(1u, G));
  KM<cf_scan, 256>(e, "cf_scan", div_up(std::max(K, M), 256), in_dev, b, K, M);
  KM<cf_prepare, 1024>(e, "cf_prepare", 1u, in_dev, b, st.jmin, K, M, G, in.host_dup, in.reserved_bits ? 1u : 0u);
  KM<cf_pack_jobs, 256>(e, "cf_pack_jobs", div_up(K, 256), in_dev, b, K);
  static_assert(offsetof(CfCtl, t) <= 512, "the control block's head is read back through the 512-byte scratch");
  pinned_copy(e, e->h_scratch, b.ctl, offsetof(CfCtl, t), hipMemcpyDeviceToHost);
  sync(e);
  CfCtl hc;
  std::memcpy((void*)&hc, e->h_scratch, offsetof(CfCtl, t));
  e->cf_inelig = hc.inelig;
  if (hc.inelig) return false;
  const unsigned NP = (M + 63u) & ~63u;
  const unsigned S = hc.any_group ? e->cf_group_run_total + hc.n_grouped : 0u;
  if (cf_lds_bytes_host(NP, M, hc.any_eq != 0u, hc.any_group ? G : 0u, S) > CF_LDS_BYTES || S > 60000u) {
    e->cf_inelig = CF_X_SHAPE;
    return false;
  }
  ctx.in = in_dev;
  ctx.st = st;
  ctx.b = b;
  return true;
}
// cf_walk for the given engines (pools of one device) on `stream`, their group chains, the books of each
void cf_run(cook_engine* lead, cook_engine* const* es, unsigned n, hipStream_t stream) {
  cook_engine* e = lead;
  for (unsigned i0 = 0; i0 < n; i0 += (unsigned)CF_PACK) {
    const unsigned c = std::min<unsigned>(CF_PACK, n - i0);
    CfPack pk{};
    for (unsigned x = 0; x < (unsigned)CF_PACK; ++x) pk.c[x] = es[i0 + (x < c ? x : 0u)]->deferred_cf;
    KLS("cf_walk", stream, cf_walk, c, CF_THREADS, pk);
  }
  for (unsigned i = 0; i < n; ++i) {
    const CfPoolCtx& c = es[i]->deferred_cf;
    const unsigned G = es[i]->last_in.G;
    if (G) KLS("cf_group_chains", stream, cf_group_chains, div_up(G, 256), 256, c.b, c.st, G);
  }
  constexpr size_t SLOT = 16 + 48 * 4;  // a pool's summary words and statistics
  if (!lead->h_cf) COOK_HIP(hipHostMalloc((void**)&lead->h_cf, 64 * SLOT, hipHostMallocDefault));
  if (n > 64) lead->fail(COOK_E_INVALID, "cook_cycle_match_multi: at most 64 pools per call");
  for (unsigned i = 0; i < n; ++i) {
    char* slot = lead->h_cf + i * SLOT;
    COOK_HIP(hipMemcpyAsync(slot, es[i]->deferred_cf.st.summary, 16, hipMemcpyDeviceToHost, stream));
    COOK_HIP(hipMemcpyAsync(slot + 16, es[i]->deferred_cf.b.ctl->stats, 48 * 4, hipMemcpyDeviceToHost, stream));
  }
  COOK_HIP(hipStreamSynchronize(stream));
  for (unsigned i = 0; i < n; ++i) {
    cook_engine* x = es[i];
    const unsigned* sum = (const unsigned*)(lead->h_cf + i * SLOT);
    if (sum[3] == 0xDEADu) lead->fail(COOK_E_STATE, "cf_walk: the pool's tables do not fit the workgroup's LDS (the host's check let it through)");
    std::memcpy(x->cf_stats, lead->h_cf + i * SLOT + 16, 48 * 4);
#ifdef CF_PROF
    {
      const uint32_t* q = x->cf_stats;
      std::fprintf(stderr, "CFPROF (x16 shader cycles) decider: slow steps %u (candidates %u evaluation %u commit %u) plain steps %u in %u walk-total %u | class wave 1: poll %u answer %u answers %u idle %u idles %u | class wave 2: poll %u answer %u answers %u idle %u idles %u | walk ticks(100MHz) %u\n", q[27], q[24], q[25], q[26], q[28], q[29], q[31], q[32], q[33], q[35], q[36], q[37], q[40], q[41], q[43], q[44], q[45], q[CFS_TICKS_WALK]);
    }
#endif
    WinCtl c{};
    c.matched = sum[0], c.head_matched = sum[1], c.rounds = sum[2], c.head = x->last_in.K, c.visited_sum = x->cf_stats[CFS_WALKED];
    c.t_seq = x->cf_stats[CFS_TICKS_TOTAL], c.t_setup = x->cf_stats[CFS_TICKS_PROLOGUE];
    x->last_ctl = c;
    x->last_form = 3;
    x->has_deferred_cf = false;
    x->has_deferred = false;
    x->match_done = true;
  }
}

void match_run_device(cook_engine* e, unsigned K, const uint32_t* j_index, bool defer = false) {
  MatchIn in = e->min;
  in.K = K;
  in.j_index = j_index;
  in.good_enough = e->params.good_enough_fitness;
  in.host_lifetime_mins = e->params.host_lifetime_mins;
  const unsigned M = in.M, G = in.G;
  e->last_in = in;
  e->last_in_valid = true;
  MatchState st;
  st.ac = e->m_ac.ensure(M);
  st.am = e->m_am.ensure(M);
  st.acount = e->m_acount.ensure(M);
  st.group_last = e->m_group_last.ensure(G);
  st.job_prev = e->m_job_prev.ensure(K);
  st.job_to_offer = e->m_j2o.ensure(K);
  st.fail_code = e->m_fail.ensure(K);
  st.summary = e->m_summary.ensure(4);
  st.alive = e->m_alive.ensure((M + 63u) / 64u + 1u);
  st.jmin = (const double*)e->m_jmin.ensure(4);
  st.xports = in.has_x ? e->m_xports.ensure(M) : nullptr;
  st.xscal = in.has_x ? e->m_xscal.ensure((size_t)M * COOK_MAX_SCALARS) : nullptr;
  match_init_state(e, st, K, M, G);
  st.cutoff = 0x7FFFFFFF;
  e->has_deferred = false;
  const int algo = e->params.match_algo;
  if (!(algo == 0 || algo == 1 || algo == 2 || algo == 3))
    e->fail(COOK_E_INVALID, "cook_params.match_algo: 0 = engine default (window rounds; class-ordered best fit where the call allows it when six or more engines share the device), 1 = serial sweep, 2 = window rounds, 3 = class-ordered best fit where the call allows it, else window rounds");
  if (defer && !(algo != 1 && K > 0)) defer = false;  // only the window rounds run several pools in one launch
  const bool ge = in.good_enough < 1.0;
  if (algo == 1) {  // one-job-at-a-time sweep by a single workgroup (reference implementation of the chain)
    constexpr int SERIAL_THREADS = COOK_SHAPE(1024, 256);
    auto k_match = match_serial<SERIAL_THREADS>;
    KL("match_serial", k_match, 1, SERIAL_THREADS, in, st);
    e->last_form = 1;
    e->has_deferred_cf = false;
  } else if (K > 0) {  // window rounds: eval -> merge -> resolve (match_v2.hpp)
    match_check_offer_count(e, M);
    V2Buf vb;
    const char* rlog_path = std::getenv("COOK_ROUND_LOG");  // diagnostics: one CSV line per round of the last match
    vb.round_log = rlog_path ? e->w_rlog.ensure(MV_ROUND_LOG_CAP) : nullptr;
    const unsigned C = div_up(M ? M : 1u, MV_OCB);
    vb.C = C;
    OfferA* oa = e->v_oa.ensure(M);
    OfferB* ob = e->v_ob.ensure(M);
    JobRec* jr = e->v_jr.ensure(K);
    vb.oa = oa;
    vb.ob = ob;
    vb.ow = e->v_ow.ensure(std::max(1u, M));
    vb.jr = jr;
    JobCons* jcons = e->v_jcons.ensure(K);
    vb.jcons = jcons;
    // sized for a LONG window (MV_WLONG jobs, match_v2.hpp) and the eval grid's largest offer split: 128 / 160 bytes per (job, offer chunk)
    vb.prec = e->v_prec.ensure((size_t)MV_WLONG * C * (sizeof(ChunkRecT<true>) > sizeof(ChunkRecT<false>) ? sizeof(ChunkRecT<true>) : sizeof(ChunkRecT<false>)));  // (a split window holds at most MV_WEVAL / split jobs)
    vb.colbits = e->v_colbits.ensure((size_t)(M ? M : 1u) * MV_JGL);
    vb.cand_fit = e->v_cand_fit.ensure((size_t)MV_WLONG * MV_LM_MAX);
    vb.cand_idx = e->v_cand_idx.ensure((size_t)MV_WLONG * MV_LM_MAX);
    vb.ge_idx = e->v_ge_idx.ensure((size_t)MV_WLONG * MV_LG_MAX);
    vb.cinfo = e->v_cinfo.ensure((size_t)MV_WLONG * 4);
    vb.jfh = e->v_jfh.ensure((size_t)MV_WLONG * (MV_FH + 2));
    vb.ctl = e->w_ctl.ensure(1);
    {  // idle rows of the eval grid take a share of the offers (eval_split) when the pool has the GPU to itself; measured on
       // MI355X: one C4 pool 66.3 -> 64.7 ms with splits up to 4, eight pools on the GPU 103 -> 113 ms (twice the chunk lists to merge,
       // more blocks than fit beside the other chains)
      const int sharing = std::max(1, g_engines_on_device[e->device & 63].load());
      vb.split_max = sharing == 1 ? (unsigned)MV_SPLIT_MAX : (sharing <= 4 ? 2u : 1u);  // (2 / 4 pools on the GPU: 69.4 -> 67.5, 72.9 -> 72.0 ms with 2)
      if (const char* ev = std::getenv("COOK_EVAL_SPLIT")) vb.split_max = (unsigned)std::max(1, std::min(MV_SPLIT_MAX, std::atoi(ev)));
      if (ge) vb.split_max = 1u;  // (the good-enough bits of a chunk are laid out for whole wave batches: match_v2.hpp ChunkRecT::gm)
    }
    {
      MatchIn* din = e->v_in.ensure(1);
      MatchIn* hin = (MatchIn*)e->h_inbuf;
      *hin = in;
      pinned_copy(e, din, hin, sizeof(MatchIn), hipMemcpyHostToDevice);
      vb.in_dev = din;
    }
    if (M) KM<match_pack_offers, 256>(e, "match_pack_offers", div_up(M, 256), (const MatchIn*)vb.in_dev, oa, ob, vb.ow);
    KM<match_pack_jobs, 256>(e, "match_pack_jobs", div_up(K, 256), (const MatchIn*)vb.in_dev, jr, jcons);
    KM<match_job_minima, 256>(e, "match_job_minima", std::min(div_up(K, 256), 256u), (const JobRec*)jr, K, e->m_jmin.ptr(), std::min(div_up(K, 256), 256u));
    if (M) KM<match_init_alive, 256>(e, "match_init_alive", div_up(M, 256), (const OfferA*)oa, M, st.jmin, st.alive);
    e->last_form = 0;
    e->has_deferred_cf = false;
    if (algo == 3 || (algo == 0 && classfit_by_default(e))) {  // class-ordered best fit when the call's numbers and constraints allow it (classfit.hpp)
      if (cf_setup(e, in, (const MatchIn*)vb.in_dev, st, jr, jcons, oa, ob, e->deferred_cf)) {
        e->cycle_considered = K;
        e->match_done = false;
        e->has_deferred_cf = true;
        if (defer) return;  // cook_cycle_match_multi runs the walks of a device's pools in one launch
        cook_engine* one[1] = {e};
        cf_run(e, one, 1, e->stream);
        return;
      }
    }
    WinCtl c0;
    std::memset(&c0, 0, sizeof(c0));
    // the first window: a call of few jobs (config.clj:113 ships fenzo-max-jobs-considered 1000) in one go — a round that stops early costs it
    // little —, a long queue with a short one (the window then follows what the rounds resolve)
    // (up to two windows' worth: the default 1000 is forty jobs more than one evaluation covers)
    c0.wcur = K <= 2u * (unsigned)MV_WEVAL ? std::max(std::min<unsigned>(K, MV_WEVAL), 1u) : std::min<unsigned>(MV_WEVAL, 128u);
    {
      // window growth: with several pools on one GPU the eval phase is compute-bound (evaluate few jobs twice); a pool
      // that has the GPU to itself is bound by the chain of rounds (prefer fewer, larger rounds)
      const int sharing = std::max(1, g_engines_on_device[e->device & 63].load());
      c0.wgrow_pct = sharing >= 4 ? 150u : 200u;
      if (const char* ev = std::getenv("COOK_WGROW_PCT")) c0.wgrow_pct = (unsigned)std::max(100, std::atoi(ev));
    }
    c0.wlong_cap = (unsigned)MV_WLONG;
    if (const char* ev = std::getenv("COOK_WLONG")) c0.wlong_cap = std::atoi(ev) ? (unsigned)MV_WLONG : (unsigned)MV_WEVAL;
    WinCtl hc = c0;
    std::memcpy(e->h_scratch, &c0, sizeof(c0));
    pinned_copy(e, vb.ctl, e->h_scratch, sizeof(WinCtl), hipMemcpyHostToDevice);
    if (defer) {  // set up only: cook_cycle_match_multi runs the rounds of several pools together
      sync(e);
      e->deferred.in = in;
      e->deferred.st = st;
      e->deferred.vb = vb;
      e->deferred_k = K;
      e->deferred_c0 = c0;
      e->deferred_ge = ge;
      e->has_deferred = true;
      e->cycle_considered = K;
      e->match_done = false;
      return;
    }
    unsigned batch = 8;
    unsigned guard = 0;
#ifdef COOK_EVAL_TRACE  // timing-study build: the waves' phase stamps of the evaluation of round COOK_EVAL_TRACE_ROUND, to stderr
    const int trace_round = std::getenv("COOK_EVAL_TRACE_ROUND") ? std::atoi(std::getenv("COOK_EVAL_TRACE_ROUND")) : -1;
    const size_t trace_words = (size_t)C * MV_JG * 3 + (size_t)C * MV_JG * 32;
    vb.eval_trace = nullptr;
    DArr<unsigned long long> d_trace;
    if (trace_round >= 0) batch = 1;
#endif
    while (hc.head < K) {
      for (unsigned r = 0; r < batch; ++r) {
#ifdef COOK_EVAL_TRACE
        if (trace_round >= 0 && (int)hc.rounds == trace_round) {
          vb.eval_trace = d_trace.ensure(trace_words);
          memset_async(e, vb.eval_trace, 0, trace_words * 8);
        } else {
          vb.eval_trace = nullptr;
        }
#endif
        if (ge) launch_round<true>(e, in, st, vb);
        else launch_round<false>(e, in, st, vb);
#ifdef COOK_EVAL_TRACE
        if (vb.eval_trace) {
          std::vector<unsigned long long> h(trace_words);
          COOK_HIP(hipMemcpy(h.data(), vb.eval_trace, trace_words * 8, hipMemcpyDeviceToHost));
          double ph[5] = {0, 0, 0, 0, 0}, tot = 0, kmin = 1e30, kmax = 0;
          unsigned nw = 0, nb = 0;
          unsigned long long k0 = ~0ull, k1 = 0;
          for (unsigned blk = 0; blk < C * (unsigned)MV_JG; ++blk) {
            const unsigned long long a = h[blk * 3], b2 = h[blk * 3 + 1];
            if (!a || !b2) continue;
            k0 = std::min(k0, a), k1 = std::max(k1, b2);
            kmin = std::min(kmin, (double)(b2 - a)), kmax = std::max(kmax, (double)(b2 - a));
            ++nb;
          }
          for (size_t t = 0; t < (size_t)C * MV_JG * 4; ++t) {
            const unsigned long long* w8 = &h[(size_t)C * MV_JG * 3 + t * 8];
            if (!w8[0] || !w8[4]) continue;
            for (int x = 0; x < 4; ++x) ph[x] += (double)(w8[x + 1] - w8[x]);
            if (w8[5]) ph[4] += (double)(w8[5] - w8[4]);
            tot += (double)((w8[5] ? w8[5] : w8[4]) - w8[0]);
            ++nw;
          }
          std::fprintf(stderr, "EVALTRACE round %d head %u wcur %u: %u blocks over %.2f us (block %.2f..%.2f us); %u waves, mean us: lane setup %.2f, stage offers %.2f, "
                               "constraint pass %.2f, fitness pass %.2f, epilogue (wave 0; /4 waves) %.2f, wave total %.2f\n",
                       trace_round, hc.head, hc.wcur, nb, (k1 - k0) / 100.0, kmin / 100.0, kmax / 100.0, nw, ph[0] / nw / 100.0, ph[1] / nw / 100.0, ph[2] / nw / 100.0,
                       ph[3] / nw / 100.0, ph[4] / nw / 100.0, tot / nw / 100.0);
        }
#endif
      }
      copy_async(e, e->h_scratch, vb.ctl, sizeof(WinCtl), hipMemcpyDeviceToHost);
      sync(e);
      const unsigned prev_head = hc.head, prev_rounds = hc.rounds;
      std::memcpy(&hc, e->h_scratch, sizeof(WinCtl));
      if (hc.head >= K) break;
      // size the next batch from the observed jobs-per-round
      const double per_round = (double)(hc.head - prev_head) / std::max(1u, hc.rounds - prev_rounds);
      const double est = (K - hc.head) / std::max(1.0, per_round);
      batch = (unsigned)std::min((double)batch_cap(), std::max(2.0, est * 1.05 + 2.0));  // over-launching is cheap: finished rounds exit at once
#ifdef COOK_EVAL_TRACE
      if (trace_round >= 0) batch = 1;  // (one round per look: the round whose stamps are wanted is found by its number)
#endif
      if (++guard > 4u * K + 64u) e->fail(COOK_E_STATE, "cook_match: window placement made no progress");
    }
    match_finish_rounds(e, st, vb, hc, e->stream);
  } else {
    unsigned sum[4] = {0u, 1u, 0u, 0u};
    std::memcpy(e->h_scratch, sum, 16);
    copy_async(e, st.summary, e->h_scratch, 16, hipMemcpyHostToDevice);
    sync(e);
  }
  e->cycle_considered = K;
  e->match_done = true;
}


// after the last round: statistics, the optional per-round log, the summary words of cook_*_fetch
void match_finish_rounds(cook_engine* e, const MatchState& st, const V2Buf& vb, const WinCtl& hc, hipStream_t stream) {
  e->last_ctl = hc;
  const char* rlog_path = std::getenv("COOK_ROUND_LOG");
  if (rlog_path && vb.round_log) {
    std::vector<RoundLog> h(std::min(hc.rounds, MV_ROUND_LOG_CAP));
    if (!h.empty()) COOK_HIP(hipMemcpy(h.data(), vb.round_log, h.size() * sizeof(RoundLog), hipMemcpyDeviceToHost));
    // one file per engine when the path ends in '@' (several pools in one process): "<path minus @>.<engine number>"
    static std::atomic<unsigned> g_rlog_seq{0};
    std::string path = rlog_path;
    if (!path.empty() && path.back() == '@') {
      if (e->rlog_id == 0) e->rlog_id = ++g_rlog_seq;
      path = path.substr(0, path.size() - 1) + "." + std::to_string(e->rlog_id);
    }
    if (FILE* f = std::fopen(path.c_str(), "w")) {
      std::fprintf(f, "head,wcur,resolved,n_list,touched,stop,matched,setup_us,seq_us,segments,h_cinfo,h_state,h_alive,h_col\n");
      for (auto& r : h)
        std::fprintf(f, "%u,%u,%u,%u,%u,%u,%u,%.2f,%.2f,%u,%08x,%08x,%08x,%08x\n", r.head, r.wcur, r.resolved, r.n_list, r.touched, r.stop, r.matched,
                     r.setup_ticks / 100.0, r.seq_ticks / 100.0, r.segments, r.h_cinfo, r.h_state, r.h_alive, r.h_col);
      std::fclose(f);
    }
  }
#ifdef COOK_WALK_PROF
  {
    static const char* cat[8] = {"shortcut", "touched_wins", "new_lane", "unmatched", "grouped", "exact", "touched_wins_by_good_enough", "fast_turn_that_left_the_loop"};
    std::fprintf(stderr, "WALKPROF rounds=%u", hc.rounds);
    for (int i = 0; i < 8; ++i)
      std::fprintf(stderr, " %s:n=%u,cyc/job=%.0f", cat[i], hc.prof_cnt[i], hc.prof_cnt[i] ? (double)hc.prof_cyc[i] / hc.prof_cnt[i] : 0.0);
    std::fprintf(stderr, "\n");
  }
#endif
  unsigned sum[4] = {hc.matched, (hc.matched == 0 || hc.head_matched) ? 1u : 0u, hc.rounds, 0u};
  std::memcpy(e->h_scratch, sum, 16);
  COOK_HIP(hipMemcpyAsync(st.summary, e->h_scratch, 16, hipMemcpyHostToDevice, stream));
  COOK_HIP(hipStreamSynchronize(stream));
}

// The placements of n engines (pools of one rank, same device) in lockstep rounds on the lead engine's stream.
void match_rounds_multi(cook_engine** es, unsigned n) {
  cook_engine* lead = es[0];
  std::vector<unsigned> live;  // engines with a deferred match
  for (unsigned i = 0; i < n; ++i) {
    if (!es[i] || es[i]->device != lead->device) lead->fail(COOK_E_INVALID, "cook_cycle_match_multi: engines must share one device");
    if (es[i]->has_deferred) live.push_back(i);
    else if (!es[i]->match_done) lead->fail(COOK_E_STATE, "cook_cycle_match_multi before cook_cycle_run_rank");
  }
  const unsigned L = (unsigned)live.size();
  if (L == 0) return;
  if (!lead->h_multi) COOK_HIP(hipHostMalloc((void**)&lead->h_multi, 64 * sizeof(WinCtl), hipHostMallocDefault));
  if (L > 64) lead->fail(COOK_E_INVALID, "cook_cycle_match_multi: at most 64 pools per call");
  std::vector<PoolCtx> hctx(L);
  std::vector<WinCtl> hc(L);
  unsigned cmax = 1;
  for (unsigned x = 0; x < L; ++x) {
    cook_engine* e = es[live[x]];
    hctx[x] = e->deferred;
    hc[x] = e->deferred_c0;
    cmax = std::max(cmax, e->deferred.vb.C);
  }
  PoolCtx* dctx = lead->w_pctx.ensure(L);
  COOK_HIP(hipMemcpyAsync(dctx, hctx.data(), L * sizeof(PoolCtx), hipMemcpyHostToDevice, lead->stream));
  COOK_HIP(hipStreamSynchronize(lead->stream));  // hctx is pageable
  cook_engine* e = lead;                         // KL times / launches on the lead engine
  bool any_ge = false;  // some pool of the launch runs with good-enough-fitness below 1: the GE launches for the whole chain (a pool at
                        // 1.0 in it is placed by best fit all the same, from the GE shape's shorter best-fit lists)
  for (unsigned x = 0; x < L; ++x) any_ge = any_ge || es[live[x]]->deferred_ge;
  unsigned batch = 8, guard = 0;
  auto all_done = [&] {
    for (unsigned x = 0; x < L; ++x)
      if (hc[x].head < es[live[x]]->deferred_k) return false;
    return true;
  };
  // up to MV_PACK pools: their contexts travel in the kernel arguments (match_v2.hpp: PoolPack)
  const bool packed = L <= (unsigned)MV_PACK && pack_args();
  PoolPack<2> pk2{};
  PoolPack<MV_PACK> pk4{};
  for (unsigned x = 0; x < (unsigned)MV_PACK; ++x) {
    if (x < 2) pk2.c[x] = hctx[x < L ? x : 0];
    pk4.c[x] = hctx[x < L ? x : 0];
  }
  auto round = [&](auto ge_tag) {
    constexpr bool GE = decltype(ge_tag)::value;
    if (packed && L <= 2u) {
      KL("match_eval2", (match_eval2_pack<GE, 2>), dim3(cmax, MV_JG, L), COOK_WAVE * MV_EW, pk2);
      KL("match_merge2", (match_merge2_pack<GE, 2>), dim3(MV_MERGE_BLOCKS, 1, L), COOK_WAVE * MV_MW, pk2);
      KL("match_resolve2", (match_resolve2_pack<GE, 2>), dim3(1, 1, L), MV_RTHREADS, pk2);
    } else if (packed) {
      KL("match_eval2", (match_eval2_pack<GE, MV_PACK>), dim3(cmax, MV_JG, L), COOK_WAVE * MV_EW, pk4);
      KL("match_merge2", (match_merge2_pack<GE, MV_PACK>), dim3(MV_MERGE_BLOCKS, 1, L), COOK_WAVE * MV_MW, pk4);
      KL("match_resolve2", (match_resolve2_pack<GE, MV_PACK>), dim3(1, 1, L), MV_RTHREADS, pk4);
    } else {
      KL("match_eval2", match_eval2_multi<GE>, dim3(cmax, MV_JG, L), COOK_WAVE * MV_EW, (const PoolCtx*)dctx);
      KL("match_merge2", match_merge2_multi<GE>, dim3(MV_MERGE_BLOCKS, 1, L), COOK_WAVE * MV_MW, (const PoolCtx*)dctx);
      KL("match_resolve2", match_resolve2_multi<GE>, dim3(1, 1, L), MV_RTHREADS, (const PoolCtx*)dctx);
    }
  };
  while (!all_done()) {
    for (unsigned r = 0; r < batch; ++r) {
      if (any_ge) round(std::true_type{});
      else round(std::false_type{});
    }
    const std::vector<WinCtl> prev = hc;
    for (unsigned x = 0; x < L; ++x)
      COOK_HIP(hipMemcpyAsync(&lead->h_multi[x], hctx[x].vb.ctl, sizeof(WinCtl), hipMemcpyDeviceToHost, lead->stream));
    COOK_HIP(hipStreamSynchronize(lead->stream));
    double est = 0;
    for (unsigned x = 0; x < L; ++x) {
      hc[x] = lead->h_multi[x];
      const unsigned K = es[live[x]]->deferred_k;
      if (hc[x].head >= K) continue;
      const double per_round = (double)(hc[x].head - prev[x].head) / std::max(1u, hc[x].rounds - prev[x].rounds);
      est = std::max(est, (K - hc[x].head) / std::max(1.0, per_round));
    }
    batch = (unsigned)std::min((double)batch_cap(), std::max(2.0, est * 1.05 + 2.0));
    if (++guard > 1000000u) lead->fail(COOK_E_STATE, "cook_cycle_match_multi: placement made no progress");
  }
  for (unsigned x = 0; x < L; ++x) {
    cook_engine* ex = es[live[x]];
    match_finish_rounds(ex, hctx[x].st, hctx[x].vb, hc[x], lead->stream);
    ex->has_deferred = false;
    ex->match_done = true;
  }
}

// COOK_MATCH_SERVED=0: cook_cycle_match_multi always runs its pools in lockstep launches (match_rounds_multi); default: served walkers
static bool served_enabled() {
  const char* s = std::getenv("COOK_MATCH_SERVED");
  return !(s && std::atoi(s) == 0);
}
// the stepping form (nothing waits on the device; the host alternates walker launches, latches and serve iterations): always in the
// emulated build, whose launches run one after the other; COOK_SERVE_STEP=1 forces it on the GPU (A/B, debugging)
static bool served_stepping() {
#ifdef __HIP_EMU__
  return true;
#else
  const char* s = std::getenv("COOK_SERVE_STEP");
  return s && std::atoi(s) != 0;
#endif
}
static unsigned long long env_ticks(const char* name, double dflt_us) {
  const char* s = std::getenv(name);
  const double us = s ? std::atof(s) : dflt_us;
  return (unsigned long long)(std::max(0.0, us) * 100.0);  // 100 MHz
}

// The placements of n engines (pools of one rank, same device) by persistent walkers — one workgroup per pool, ONE launch — beside
// serve iterations (evaluation + merge for the pools that asked) on a second stream: match_v2.hpp "served walkers".  -> false: the
// served match gave up (a walker was not served in time); the pools are in a consistent state and the caller finishes them in lockstep.
bool match_rounds_served(cook_engine** es, unsigned n) {
  cook_engine* lead = es[0];
  cook_engine* e = lead;  // KL / KLS time and launch on the lead engine
  std::vector<unsigned> live;
  for (unsigned i = 0; i < n; ++i) {
    if (!es[i] || es[i]->device != lead->device) lead->fail(COOK_E_INVALID, "cook_cycle_match_multi: engines must share one device");
    if (es[i]->has_deferred) live.push_back(i);
    else if (!es[i]->match_done) lead->fail(COOK_E_STATE, "cook_cycle_match_multi before cook_cycle_run_rank");
  }
  const unsigned L = (unsigned)live.size();
  lead->served = cook_engine::ServedStats{};
  if (L == 0) return true;
  if (L > MV_SERVE_MAX) return false;
  constexpr unsigned MAXS = cook_engine::kMaxServers;
  if (!lead->s_walk) {
    // The walkers' stream must never share a HARDWARE queue with a serve stream: a serve launch queued behind the persistent walker
    // launch would wait for walkers that wait for it (seen with eight serve streams on GPU_MAX_HW_QUEUES=8: every cycle ran into the
    // walkers' time-out).  HIP hands streams of different priorities queues of different pools, so the walkers get the only
    // high-priority stream of the process; the serve streams are ordinary ones (two of them on one queue would only take turns).
    int prio_least = 0, prio_greatest = 0;
    COOK_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    COOK_HIP(hipStreamCreateWithPriority(&lead->s_walk, hipStreamNonBlocking, prio_greatest));
    COOK_HIP(hipHostMalloc((void**)&lead->h_serve, MAXS * sizeof(ServeHost), hipHostMallocDefault));
  }
  if (!lead->h_multi) COOK_HIP(hipHostMalloc((void**)&lead->h_multi, 64 * sizeof(WinCtl), hipHostMallocDefault));
  // SERVERS: streams of serve iterations, each for its own share of the pools (pool x -> server x mod S).  An iteration is a chain of
  // latency-bound launches that leaves most of the chip idle (a window of 300 jobs is 980 waves for 4 096 slots), so two or three of
  // them side by side serve the walkers sooner than one; the walkers' launch makes S + 1 streams.
  unsigned S = 3;  // (eight pools on MI355X: 1 / 2 / 3 / 4 servers 56.6 / 53.7 / 52.7 / 57.3 ms: walkers + three servers are the four streams the part runs at full speed)
  if (const char* ev = std::getenv("COOK_SERVE_STREAMS")) S = (unsigned)std::max(1, std::atoi(ev));
  S = std::min(std::min(S, MAXS), L);
  for (unsigned sv = 0; sv < S; ++sv)
    if (!lead->s_serve[sv]) COOK_HIP(hipStreamCreateWithFlags(&lead->s_serve[sv], hipStreamNonBlocking));
  std::vector<PoolCtx> hctx(L);
  unsigned cmax = 1;
  bool any_ge = false;
  for (unsigned x = 0; x < L; ++x) {
    hctx[x] = es[live[x]]->deferred;
    cmax = std::max(cmax, hctx[x].vb.C);
    any_ge = any_ge || es[live[x]]->deferred_ge;
  }
  PoolCtx* dctx = lead->w_pctx.ensure(L);
  ServeSlot* slots = lead->w_slots.ensure(L);
  ServeCtl* sctl = lead->w_sctl.ensure(MAXS);
  std::vector<ServeSlot> hslots(L);
  for (unsigned x = 0; x < L; ++x) {
    std::memset((void*)&hslots[x], 0, sizeof(ServeSlot));
    hslots[x].req = 1u;  // the first window of every pool: asked for here
    hslots[x].claim = 1u;  // ... and given to a server here (the first iterations' lists below)
  }
  // DYNAMIC assignment (default; COOK_SERVE_DYNAMIC=0: pool x belongs to server x mod S for the whole call): every server looks at every pool and
  // takes the open requests it sees first — a walker's request no longer queues behind its neighbours' on ONE server while another polls an empty list
  // (measured: 133 us from request to lists on the servers with three pools, 107-123 on the one with two: profiles/r05zz_serve_trace.txt)
  const bool dynamic = !(std::getenv("COOK_SERVE_DYNAMIC") && std::atoi(std::getenv("COOK_SERVE_DYNAMIC")) == 0);
  const unsigned claim_max = std::max(1u, div_up(L, S));
  std::vector<ServeCtl> hs(S);
  unsigned zmax = 1;
  for (unsigned sv = 0; sv < S; ++sv) {
    std::memset(&hs[sv], 0, sizeof(ServeCtl));
    hs[sv].pool_first = sv;
    hs[sv].pool_stride = S;
    hs[sv].dbg_fence = std::getenv("COOK_SERVE_FENCE") && std::atoi(std::getenv("COOK_SERVE_FENCE")) ? 1u : 0u;
    unsigned cnt = 0;
    for (unsigned x = sv; x < L; x += S) hs[sv].latch[0].pool[cnt] = x, hs[sv].latch[0].seq[cnt] = 1u, ++cnt;
    hs[sv].n_pools = hs[sv].latch[0].n = cnt;
    hs[sv].latch[0].ticket_target = cnt * (unsigned)MV_MERGE_BLOCKS;
    if (dynamic) hs[sv].n_pools = L, hs[sv].pool_first = 0u, hs[sv].pool_stride = 1u, hs[sv].claim_max = claim_max;
    hs[sv].dbg_delay[0] = (unsigned)env_ticks("COOK_SERVE_DELAY_PUBLISH_US", 0.0);
    hs[sv].dbg_delay[1] = (unsigned)env_ticks("COOK_SERVE_DELAY_ACQ_US", 0.0);
    hs[sv].dbg_delay[2] = (unsigned)env_ticks("COOK_SERVE_DELAY_READ_US", 0.0);
    zmax = std::max(zmax, cnt);
  }
  if (dynamic) zmax = std::max(zmax, std::min(L, claim_max));
  ServeHost* hh = lead->h_serve;
  std::memset(hh, 0, MAXS * sizeof(ServeHost));
  hipStream_t s0 = lead->s_serve[0];
  COOK_HIP(hipMemcpyAsync(dctx, hctx.data(), L * sizeof(PoolCtx), hipMemcpyHostToDevice, s0));
  COOK_HIP(hipMemcpyAsync(slots, hslots.data(), L * sizeof(ServeSlot), hipMemcpyHostToDevice, s0));
  COOK_HIP(hipMemcpyAsync(sctl, hs.data(), S * sizeof(ServeCtl), hipMemcpyHostToDevice, s0));
  COOK_HIP(hipStreamSynchronize(s0));  // (pageable sources; and the walkers must find their slots initialised)
  WalkPack<MV_WALK_PACK> wp{};
  const bool packed = L <= (unsigned)MV_WALK_PACK && pack_args();
  for (unsigned x = 0; x < (unsigned)MV_WALK_PACK; ++x) {
    wp.c[x].st = hctx[x < L ? x : 0].st;
    wp.c[x].vb = hctx[x < L ? x : 0].vb;
  }
  const bool stepping = served_stepping();
  const bool one_stream = std::getenv("COOK_SERVE_ONE_STREAM") && std::atoi(std::getenv("COOK_SERVE_ONE_STREAM"));  // (diagnostics: the servers' iterations all on one stream)
  const unsigned long long spin = stepping ? 0ull : env_ticks("COOK_SERVE_WALK_TIMEOUT_US", 2.5e5);  // a walker not served for 250 ms gives up (a cycle is 50)
  const unsigned long long poll = stepping ? 0ull : env_ticks("COOK_SERVE_POLL_US", 40.0);          // the latch waits that long for a request
  std::vector<unsigned> launched(S, 0u);  // serve iterations launched, per server
  auto walkers = [&](auto ge_tag) {
    constexpr bool GE = decltype(ge_tag)::value;
    if (packed) KLS("match_walkers", lead->s_walk, (match_walkers_pack<GE, MV_WALK_PACK>), L, MV_RTHREADS, wp, slots, sctl, spin);
    else KLS("match_walkers", lead->s_walk, match_walkers<GE>, L, MV_RTHREADS, (const PoolCtx*)dctx, slots, sctl, spin);
  };
  auto serve = [&](auto ge_tag, unsigned sv) {
    constexpr bool GE = decltype(ge_tag)::value;
    hipStream_t st_ = lead->s_serve[one_stream ? 0u : sv];
    const unsigned it = launched[sv];  // the iteration's number picks its latch list (ServeLatch)
    KLS("match_serve_eval", st_, match_serve_eval<GE>, dim3(cmax, MV_JG, zmax), COOK_WAVE * MV_EW, (const PoolCtx*)dctx, (const ServeCtl*)(sctl + sv), it);
    KLS("match_serve_merge", st_, match_serve_merge<GE>, dim3(MV_MERGE_BLOCKS, 1, zmax), COOK_WAVE * MV_MW, (const PoolCtx*)dctx, sctl + sv, slots, hh + sv, poll, it);
    ++launched[sv];
  };
  auto launch_walkers = [&] { any_ge ? walkers(std::true_type{}) : walkers(std::false_type{}); };
  auto launch_serve = [&](unsigned sv) { any_ge ? serve(std::true_type{}, sv) : serve(std::false_type{}, sv); };
  volatile ServeHost* vh = hh;
  auto all_done = [&] {
    for (unsigned sv = 0; sv < S; ++sv)
      if (!vh[sv].all_done) return false;
    return true;
  };
  auto any_error = [&] {
    for (unsigned sv = 0; sv < S; ++sv)
      if (vh[sv].error) return true;
    return false;
  };
  auto sync_servers = [&] {
    for (unsigned sv = 0; sv < S; ++sv) COOK_HIP(hipStreamSynchronize(lead->s_serve[sv]));
  };
  bool stuck = false;
  if (stepping) {
    unsigned guard = 0;
    for (;;) {
      // (one phase at a time, on the GPU too: the latch launch publishes nothing and expects to find every open request unlatched)
      for (unsigned sv = 0; sv < S; ++sv) launch_serve(sv);  // evaluates what the latch put together (first: every pool's first window), publishes
      sync_servers();
      launch_walkers();   // every pool walks the windows it has been served, asks for the next, returns
      COOK_HIP(hipStreamSynchronize(lead->s_walk));
      for (unsigned sv = 0; sv < S; ++sv) KLS("match_serve_latch", lead->s_serve[sv], match_serve_latch, 1, COOK_WAVE, sctl + sv, slots, hh + sv, launched[sv] - 1u);
      sync_servers();
      if (all_done() || any_error()) break;
      if (++guard > 4000000u) lead->fail(COOK_E_STATE, "cook_cycle_match_multi: served placement made no progress");
    }
  } else {
    launch_walkers();
    // serve iterations, a few ahead of the device: each ends with the latch waiting (bounded) for the next request, so every server's chain
    // is paced by its walkers; iter_done / all_done arrive in page-locked memory
    constexpr unsigned DEPTH = 3;
    const auto t_begin = std::chrono::steady_clock::now();
    unsigned long long spins = 0;
    while (!all_done() && !any_error()) {
      bool any = false;
      for (unsigned sv = 0; sv < S; ++sv) {
        if (vh[sv].all_done || launched[sv] - vh[sv].iter_done >= DEPTH) continue;
        launch_serve(sv);
        any = true;
      }
      if (!any && (++spins & 0xFFFFull) == 0ull && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > 30.0) {
        stuck = true;  // (the walkers give up on their own after COOK_SERVE_WALK_TIMEOUT_US without being served)
        break;
      }
    }
    sync_servers();
    COOK_HIP(hipStreamSynchronize(lead->s_walk));
    if (stuck) lead->fail(COOK_E_STATE, "cook_cycle_match_multi: the serve launches stopped finishing");
  }
  // what the pools reached
  std::vector<WinCtl> hc(L);
  for (unsigned x = 0; x < L; ++x) COOK_HIP(hipMemcpyAsync(&lead->h_multi[x], hctx[x].vb.ctl, sizeof(WinCtl), hipMemcpyDeviceToHost, s0));
  COOK_HIP(hipMemcpyAsync(hs.data(), sctl, S * sizeof(ServeCtl), hipMemcpyDeviceToHost, s0));
  static const bool serve_trace = std::getenv("COOK_SERVE_TRACE") != nullptr;
  if (serve_trace) COOK_HIP(hipMemcpyAsync(hslots.data(), slots, L * sizeof(ServeSlot), hipMemcpyDeviceToHost, s0));
  COOK_HIP(hipStreamSynchronize(s0));
  if (serve_trace) {  // the walkers' and the servers' own accounts of the call (100 MHz ticks -> microseconds)
    for (unsigned x = 0; x < L; ++x)
      std::fprintf(stderr, "SERVETRACE pool %u: %u windows waited for, %.1f us each from request to lists, %.1f us from the end of a round to its request\n", x,
                   hslots[x].waits, hslots[x].waits ? hslots[x].wait_ticks / 100.0 / hslots[x].waits : 0.0,
                   hslots[x].waits ? hslots[x].post_ticks / 100.0 / hslots[x].waits : 0.0);
    for (unsigned sv = 0; sv < S; ++sv) {
      const unsigned work = hs[sv].iterations - hs[sv].empty_iterations;
      std::fprintf(stderr, "SERVETRACE server %u: %u iterations with work (%u pool windows), %.1f us each from its list to its results; %u empty iterations, %.1f ms waiting for requests\n",
                   sv, work, hs[sv].pools_served, work ? hs[sv].busy_ticks / 100.0 / work : 0.0, hs[sv].empty_iterations, hs[sv].wait_ticks / 1.0e5);
    }
  }
  bool complete = true;
  for (unsigned x = 0; x < L; ++x) {
    hc[x] = lead->h_multi[x];
    es[live[x]]->deferred_c0 = hc[x];  // (where a lockstep continuation would start)
    complete = complete && hc[x].head >= es[live[x]]->deferred_k;
  }
  lead->served.mode = stepping ? 2u : 1u;
  lead->served.pools = L;
  lead->served.servers = S;
  for (unsigned sv = 0; sv < S; ++sv) {
    lead->served.iterations += hs[sv].iterations;
    lead->served.empty_iterations += hs[sv].empty_iterations;
    lead->served.pools_served += hs[sv].pools_served;
    lead->served.latch_wait_ms += (double)hs[sv].wait_ticks / 1.0e5;
  }
  if (!complete) {
    lead->served.fell_back = 1;
    return false;
  }
  for (unsigned x = 0; x < L; ++x) {
    cook_engine* ex = es[live[x]];
    match_finish_rounds(ex, hctx[x].st, hctx[x].vb, hc[x], s0);
    ex->has_deferred = false;
    ex->match_done = true;
  }
  return true;
}

void match_fetch(cook_engine* e, unsigned K, int32_t* job_to_offer, uint32_t* fail_code, uint8_t* head_matched) {
  if (!e->match_done) e->fail(COOK_E_STATE, "cook_match_fetch before cook_match_run");
  if (job_to_offer && K) copy_async(e, job_to_offer, e->m_j2o.ptr(), (size_t)K * 4, hipMemcpyDeviceToHost);
  if (fail_code && K) copy_async(e, fail_code, e->m_fail.ptr(), (size_t)K * 4, hipMemcpyDeviceToHost);
  copy_async(e, e->h_scratch, e->m_summary.ptr(), 16, hipMemcpyDeviceToHost);
  sync(e);
  unsigned s[4];
  std::memcpy(s, e->h_scratch, 16);
  if (head_matched) *head_matched = (uint8_t)s[1];
}

COOK_KERNEL void cycle_job_index(const uint32_t* __restrict__ ranked, const uint32_t* __restrict__ pend_ord, unsigned k, uint32_t* __restrict__ j_index) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) j_index[i] = pend_ord[ranked[i]];
}

struct StageTimer {
  cook_engine* e;
  int slot;
  double* out;
  StageTimer(cook_engine* e_, int s, double* o) : e(e_), slot(s), out(o) { (void)hipEventRecord(e->ev_stage[slot], e->stream); }
  void stop() {
    (void)hipEventRecord(e->ev_stage[slot + 1], e->stream);
    (void)hipEventSynchronize(e->ev_stage[slot + 1]);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e->ev_stage[slot], e->ev_stage[slot + 1]);
    *out = ms;
  }
};

#include "considerable_host.hpp"
#include "rebalance_host.hpp"
#include "offers_host.hpp"
#include "explain_host.hpp"
#include "cycle_update.hpp"

ConsBufs& cons_bufs(cook_engine* e) {
  if (!e->cb) e->cb = new ConsBufs();
  return *e->cb;
}

ExplainBufs& explain_bufs(cook_engine* e) {
  if (!e->xb) e->xb = new ExplainBufs();
  return *e->xb;
}

OfferBufs& offer_bufs(cook_engine* e) {
  if (!e->ofb) e->ofb = new OfferBufs();
  return *e->ofb;
}

RebalBufs& rebal_bufs(cook_engine* e) {
  if (!e->rb) e->rb = new RebalBufs();
  return *e->rb;
}

template <class F>
int guarded(cook_engine* e, F&& f) {
  if (!e) return COOK_E_INVALID;
  try {
    COOK_HIP(hipSetDevice(e->device));
    f();
    e->err.clear();
    return COOK_OK;
  } catch (const cook_error& ce) {
    e->err = ce.msg;
    (void)hipStreamSynchronize(e->stream);
    e->ev_pending.clear();
    e->ev_used = 0;
    return ce.code;
  } catch (const std::exception& ex) {
    e->err = ex.what();
    return COOK_E_NOMEM;
  } catch (...) {  // nothing may cross the C boundary
    e->err = "unknown exception";
    return COOK_E_STATE;
  }
}

}  // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
// every device buffer the engine itself owns (the plans' buffers free themselves)
static std::vector<DBuf*> engine_bufs(cook_engine* e) {
  return {&e->d_scratch64.b, &e->d_counters.b, &e->t_cpus.b, &e->t_mem.b, &e->t_gpus.b, &e->u_divc.b, &e->u_divm.b, &e->u_divg.b,
                  &e->u_qcount.b, &e->u_qcpus.b, &e->u_qmem.b, &e->u_qgpus.b, &e->t_user.b, &e->permA.b, &e->permB2.b, &e->s_user.b,
                  &e->seg_start.b, &e->seg_end.b, &e->inexact_user.b, &e->rank_of_item.b, &e->gstart.b, &e->tpos.b, &e->titem.b,
                  &e->tsorted.b, &e->tsorted2.b, &e->qitemA.b, &e->qitemB.b, &e->ranked.b, &e->pend_ord.b, &e->hist.b, &e->t_prio.b,
                  &e->t_start.b, &e->t_task.b, &e->t_job.b, &e->t_pending.b, &e->s_pending.b, &e->head.b, &e->keep.b, &e->thead.b,
                  &e->w0.b, &e->w1.b, &e->w2.b, &e->dkey.b, &e->nkkey.b, &e->ckey.b, &e->s_use.b, &e->pre.b, &e->quseA.b, &e->quseB.b,
                  &e->qpre.b, &e->pool_usage.b, &e->scanI.b, &e->iflag.b, &e->ones_buf.b, &e->tied_buf.b, &e->dru.b, &e->dru_out.b, &e->tmpU4.agg.b, &e->tmpU4.carry.b,
                  &e->tmpU4.first_head.b, &e->tmpI.agg.b, &e->tmpI.carry.b, &e->tmpI.first_head.b, &e->permC1.b, &e->permC2.b,
                  &e->j_cpus.b, &e->j_mem.b, &e->j_gpus.b, &e->j_disk_req.b, &e->o_cpus.b, &e->o_mem.b, &e->o_gpu_count.b,
                  &e->o_disk_space.b, &e->o_run_cpus.b, &e->o_run_mem.b, &e->m_ac.b, &e->m_am.b, &e->j_gpu_model.b, &e->j_group.b,
                  &e->j_eq_off.b, &e->j_eq_key.b, &e->j_eq_val.b, &e->j_novel_off.b, &e->j_novel_host.b, &e->j_ckpt.b, &e->j_disk_type.b,
                  &e->j_index.b, &e->o_host.b, &e->o_gpu_model.b, &e->o_disk_type.b, &e->o_attr.b, &e->o_location.b, &e->g_attr_key.b,
                  &e->g_run_off.b, &e->g_run_host.b, &e->g_run_attr.b, &e->reserved_bits.b, &e->m_fail.b, &e->j_reserved_host.b,
                  &e->o_max_tasks.b, &e->o_num_tasks.b, &e->o_run_count.b, &e->g_min.b, &e->m_acount.b, &e->m_group_last.b,
                  &e->m_job_prev.b, &e->m_j2o.b, &e->j_est_end.b, &e->o_host_start.b, &e->o_k8s.b, &e->g_type.b, &e->m_summary.b, &e->v_oa.b, &e->v_ob.b, &e->v_ow.b, &e->v_jr.b, &e->v_prec.b, &e->v_cand_fit.b, &e->v_cand_idx.b, &e->v_ge_idx.b, &e->v_cinfo.b, &e->v_colbits.b, &e->w_ctl.b, &e->v_in.b,
                  &e->dhead.b, &e->tie_ctl.b, &e->cf_ctl.b, &e->cf_attr8.b, &e->cf_h2o.b, &e->cf_pos[0].b, &e->cf_pos[1].b, &e->cf_pos[2].b, &e->cf_scr[0].b,
                  &e->cf_scr[1].b, &e->cf_scr[2].b, &e->cf_gcount.b, &e->cf_gmem.b, &e->cf_jobs.b};
}
extern "C" {

const char* cook_version(void) {
  return "cookmatch 0.3.0 (" COOK_BUILD_NAME ")";
}
int cook_abi_version(void) { return COOK_ABI_VERSION; }

int cook_engine_create(const cook_params* params, int device_id, cook_engine** out) {
  if (!params || !out) return COOK_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return COOK_E_DEVICE;  // no GPU: fail loudly, no CPU fallback
  if (device_id < 0 || device_id >= ndev) return COOK_E_INVALID;
  cook_engine* e = nullptr;
  try {
    e = new cook_engine();
    e->params = *params;
    e->device = device_id;
    COOK_HIP(hipSetDevice(device_id));
    COOK_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) e->n_cus = prop.multiProcessorCount;
    }
    for (int i = 0; i < 4; ++i) COOK_HIP(hipEventCreate(&e->ev_stage[i]));
    COOK_HIP(hipHostMalloc((void**)&e->h_scratch, 64 * 8, hipHostMallocDefault));
    COOK_HIP(hipHostMalloc((void**)&e->h_inbuf, sizeof(MatchIn), hipHostMallocDefault));
    e->d_scratch64.ensure(64);
    e->d_counters.ensure(64);
  } catch (const cook_error&) {
    delete e;
    return COOK_E_DEVICE;
  } catch (...) {
    delete e;
    return COOK_E_NOMEM;
  }
  g_engines_on_device[device_id & 63].fetch_add(1);
  *out = e;
  return COOK_OK;
}

void cook_engine_destroy(cook_engine* e) {
  if (!e) return;
  g_engines_on_device[e->device & 63].fetch_sub(1);
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  if (g_guard && std::getenv("COOK_GUARD_SELFTEST") && e->m_j2o.b.p)  // the guard's own test: one byte past the end of the placement column
    (void)hipMemset((char*)e->m_j2o.b.p + e->m_j2o.b.cap, 0, 1);
  const std::vector<DBuf*> bufs = engine_bufs(e);
  for (DBuf* b : bufs) b->release();
  for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
  for (int i = 0; i < 4; ++i)
    if (e->ev_stage[i]) (void)hipEventDestroy(e->ev_stage[i]);
  if (e->h_scratch) (void)hipHostFree(e->h_scratch);
  if (e->h_inbuf) (void)hipHostFree(e->h_inbuf);
  if (e->h_multi) (void)hipHostFree(e->h_multi);
  if (e->h_cf) (void)hipHostFree(e->h_cf);
  if (e->h_serve) (void)hipHostFree(e->h_serve);
  if (e->s_walk) (void)hipStreamDestroy(e->s_walk);
  for (hipStream_t sv : e->s_serve)
    if (sv) (void)hipStreamDestroy(sv);
  delete e->rb;
  e->rb = nullptr;
  delete e->cb;
  e->cb = nullptr;
  delete e->ofb;
  e->ofb = nullptr;
  delete e->xb;
  e->xb = nullptr;
  delete e->ub;
  e->ub = nullptr;
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int cook_engine_set_params(cook_engine* e, const cook_params* p) {
  if (!e || !p) return COOK_E_INVALID;
  e->params = *p;
  return COOK_OK;
}

const char* cook_last_error(const cook_engine* e) { return e ? e->err.c_str() : "null engine"; }

int cook_rank_stage(cook_engine* e, const cook_tasks* tasks, const cook_users* users) {
  return guarded(e, [&] { rank_stage(e, tasks, users); });
}
int cook_rank_set_quota(cook_engine* e, const cook_pool_quota* q) {
  if (!e) return COOK_E_INVALID;
  if (q)
    e->quota = *q;
  else
    std::memset(&e->quota, 0, sizeof(e->quota));
  return COOK_OK;
}
int cook_rank_pool_usage(cook_engine* e, cook_usage* out) {
  if (!out) return COOK_E_INVALID;
  return guarded(e, [&] { rank_pool_usage(e, out); });
}
int cook_rank_user_usage(cook_engine* e, double* usage, int usage_is_device) {
  return guarded(e, [&] { rank_user_usage(e, usage, usage_is_device != 0); });
}
int cook_rank_run(cook_engine* e) {
  return guarded(e, [&] {
    StageTimer t(e, 0, &e->rank_ms);
    rank_run(e);
    t.stop();
    prof_collect(e);
  });
}
int cook_rank_fetch(cook_engine* e, uint32_t* ranked, uint32_t* n_out, double* dru) {
  return guarded(e, [&] { rank_fetch(e, ranked, n_out, dru); });
}
int cook_rank(cook_engine* e, const cook_tasks* tasks, const cook_users* users, const cook_pool_quota* quota, uint32_t* ranked,
              uint32_t* n_out, double* dru) {
  if (n_out) *n_out = 0;
  int rc = cook_rank_stage(e, tasks, users);
  if (rc) return rc;
  cook_rank_set_quota(e, quota);
  rc = cook_rank_run(e);
  if (rc) return rc;
  return cook_rank_fetch(e, ranked, n_out, dru);
}

int cook_match_stage(cook_engine* e, const cook_jobs* j, const cook_offers* o, const cook_groups* g, const uint32_t* reserved_hosts,
                     uint32_t n_reserved) {
  return guarded(e, [&] {
    match_stage_inputs(e, j, o, g, reserved_hosts, n_reserved);
    e->cycle_staged = false;
  });
}
namespace {
// the rows of the last cook_offers_run as a cook_offers of device columns (offer.clj:31-76: Kubernetes leases; COOK_MAX_TASKS_PER_HOST
// = the cluster's max pods per node, COOK_NUM_TASKS_ON_HOST = the node's pod count)
cook_offers built_offers_view(cook_engine* e, int with_task_limits) {
  if (!e->ofb || !e->ofb->done) e->fail(COOK_E_STATE, "built offers requested before cook_offers_run");
  OfferBufs& b = *e->ofb;
  const unsigned M = b.n_offers;
  b.o_k8s.ensure(std::max(1u, M));
  b.o_max_tasks.ensure(std::max(1u, M));
  if (M) {
    memset_async(e, b.o_k8s.ptr(), 1, M);
    KM<fill_i32, 256>(e, "fill_i32", div_up(M, 256), b.o_max_tasks.ptr(), M, (int32_t)b.params.max_pods_per_node);
  }
  cook_offers o;
  std::memset(&o, 0, sizeof(o));
  o.n = M;
  o.cpus = b.o_cpus.ptr();
  o.mem = b.o_mem.ptr();
  o.host = b.o_host.ptr();
  o.k8s = b.o_k8s.ptr();
  o.gpu_model = b.o_gpu_model.ptr();
  o.gpu_count = b.o_gpu_count.ptr();
  o.disk_type = b.o_disk_type.ptr();
  o.disk_space = b.o_disk_space.ptr();
  o.gpu_slots = b.gpu_slots;
  o.disk_slots = b.disk_slots;
  o.n_attr_keys = b.n_attr;
  o.attr = b.n_attr ? b.o_attr.ptr() : nullptr;
  if (with_task_limits) {
    o.max_tasks = b.o_max_tasks.ptr();
    o.num_tasks = b.o_num_pods.ptr();
  }
  return o;
}
}  // namespace

int cook_match_stage_built_offers(cook_engine* e, const cook_jobs* j, const cook_groups* g, const uint32_t* reserved_hosts,
                                  uint32_t n_reserved, int with_task_limits) {
  return guarded(e, [&] {
    const cook_offers o = built_offers_view(e, with_task_limits);
    match_stage_inputs(e, j, &o, g, reserved_hosts, n_reserved, true);
    e->cycle_staged = false;
  });
}
int cook_cycle_stage_built_offers(cook_engine* e, const cook_tasks* tasks, const cook_users* users, const cook_jobs* pending_jobs,
                                  const cook_groups* groups, const uint32_t* reserved_hosts, uint32_t n_reserved, int with_task_limits) {
  return guarded(e, [&] {
    const cook_offers o = built_offers_view(e, with_task_limits);
    rank_stage(e, tasks, users);
    if (!pending_jobs || pending_jobs->n != e->n_pending)
      e->fail(COOK_E_INVALID, "cook_cycle_stage_built_offers: pending_jobs->n must equal the number of pending tasks");
    match_stage_inputs(e, pending_jobs, &o, groups, reserved_hosts, n_reserved, true);
    e->cycle_staged = true;
    if (e->ub) e->ub->csr_known = false;  // (cycle_update.hpp: the staged CSR columns' sizes are looked up again)
  });
}
int cook_match_run(cook_engine* e) {
  return guarded(e, [&] {
    if (!e->match_staged) e->fail(COOK_E_STATE, "cook_match_run before cook_match_stage");
    StageTimer t(e, 2, &e->match_ms);
    match_run_device(e, e->K, nullptr);
    t.stop();
    prof_collect(e);
  });
}
int cook_match_fetch(cook_engine* e, int32_t* job_to_offer, uint32_t* fail_code, uint8_t* head_matched) {
  return guarded(e, [&] { match_fetch(e, e->cycle_considered, job_to_offer, fail_code, head_matched); });
}
int cook_match_count(cook_engine* e, uint32_t* n_jobs) {
  return guarded(e, [&] {
    if (!n_jobs) e->fail(COOK_E_INVALID, "cook_match_count: null n_jobs");
    if (!e->match_done) e->fail(COOK_E_STATE, "cook_match_count before a match has run");
    *n_jobs = e->cycle_considered;
  });
}
int cook_match(cook_engine* e, const cook_jobs* j, const cook_offers* o, const cook_groups* g, const uint32_t* reserved_hosts,
               uint32_t n_reserved, int32_t* job_to_offer, uint32_t* fail_code, uint8_t* head_matched) {
  if (j && job_to_offer)
    for (uint32_t k = 0; k < j->n; ++k) job_to_offer[k] = -1;  // "no matches" on any error path
  if (head_matched) *head_matched = 1;
  int rc = cook_match_stage(e, j, o, g, reserved_hosts, n_reserved);
  if (rc) return rc;
  rc = cook_match_run(e);
  if (rc) return rc;
  return cook_match_fetch(e, job_to_offer, fail_code, head_matched);
}

int cook_cycle_stage(cook_engine* e, const cook_tasks* tasks, const cook_users* users, const cook_jobs* pending_jobs,
                     const cook_offers* offers, const cook_groups* groups, const uint32_t* reserved_hosts, uint32_t n_reserved) {
  return guarded(e, [&] {
    rank_stage(e, tasks, users);
    if (!pending_jobs || pending_jobs->n != e->n_pending)
      e->fail(COOK_E_INVALID, "cook_cycle_stage: pending_jobs->n must equal the number of pending tasks");
    match_stage_inputs(e, pending_jobs, offers, groups, reserved_hosts, n_reserved);
    e->cycle_staged = true;
    if (e->ub) e->ub->csr_known = false;  // (cycle_update.hpp: the staged CSR columns' sizes are looked up again)
  });
}
// rank -> (considerable filters) -> take K -> the job index array of the match; returns K
static unsigned cycle_rank_part(cook_engine* e, uint32_t num_considerable) {
  if (!e->cycle_staged) e->fail(COOK_E_STATE, "cook_cycle_run before cook_cycle_stage");
  if (recording()) {  // (a pool batch times its joint sequence of launches itself)
    rank_run(e);
  } else {
    StageTimer tr(e, 0, &e->rank_ms);
    rank_run(e);
    tr.stop();
  }
  unsigned K = std::min<unsigned>(num_considerable, e->n_ranked);  // (take num-considerable), scheduler.clj:751
  if (e->cb && e->cb->cycle_on) {  // pending-jobs->considerable-jobs between rank and match (scheduler.clj:729-762)
    ConsBufs& c = *e->cb;
    if (!e->has_j_user) e->fail(COOK_E_INVALID, "cook_cycle_run: the considerable filters need pending_jobs->user");
    const unsigned n = e->n_ranked;
    c.q_cpus.ensure(n), c.q_mem.ensure(n), c.q_gpus.ensure(n), c.q_user.ensure(n), c.q_elig.ensure(n);
    if (n)
      KM<cons_gather_queue, 256>(e, "cons_gather_queue", div_up(n, 256), (const uint32_t*)e->ranked.ptr(), (const uint32_t*)e->pend_ord.ptr(), n,
          e->min.j_cpus, e->min.j_mem, e->min.j_gpus, (const uint32_t*)e->j_user.ptr(),
          c.has_elig_by_pending ? (const uint8_t*)c.elig_by_pending.ptr() : (const uint8_t*)nullptr, c.q_cpus.ptr(), c.q_mem.ptr(), c.q_gpus.ptr(),
          c.q_user.ptr(), c.q_elig.ptr());
    cons_run_device(e, c, n, c.q_cpus.ptr(), c.q_mem.ptr(), c.q_gpus.ptr(), c.q_user.ptr(), c.q_elig.ptr(), num_considerable);
    K = c.n_result;
    e->j_index.ensure(K);
    if (K)
      KM<cons_job_index, 256>(e, "cons_job_index", div_up(K, 256), (const uint32_t*)c.result, (const uint32_t*)e->ranked.ptr(),
          (const uint32_t*)e->pend_ord.ptr(), K, e->j_index.ptr());
  } else {
    e->j_index.ensure(K);
    if (K)
      KM<cycle_job_index, 256>(e, "cycle_job_index", div_up(K, 256), (const uint32_t*)e->ranked.ptr(), (const uint32_t*)e->pend_ord.ptr(), K, e->j_index.ptr());
  }
  return K;
}
int cook_cycle_update(cook_engine* e, const cook_cycle_delta* delta) {
  // where the call's time went, for cook_match_stats_ex [26..28] (an occasional 9 ms call among 1 ms ones: bench.py boundary.update_ms_samples)
  const auto t0 = std::chrono::steady_clock::now();
  const double sync0 = tl_sync_ms;
  const unsigned alloc0 = tl_dbuf_allocs;
  const int rc = guarded(e, [&] {
    if (!e->ub) e->ub = new UpdateBufs();
    cycle_update(e, *e->ub, delta);
    prof_collect(e);
  });
  if (e) {
    e->upd_us = (uint32_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    e->upd_sync_us = (uint32_t)((tl_sync_ms - sync0) * 1000.0);
    e->upd_allocs = tl_dbuf_allocs - alloc0;
  }
  return rc;
}
void* cook_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void cook_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
int cook_cycle_run(cook_engine* e, uint32_t num_considerable) {
  return guarded(e, [&] {
    const unsigned K = cycle_rank_part(e, num_considerable);
    StageTimer tm(e, 2, &e->match_ms);
    match_run_device(e, K, K ? e->j_index.ptr() : nullptr);
    tm.stop();
    prof_collect(e);
  });
}
int cook_cycle_run_rank(cook_engine* e, uint32_t num_considerable) {
  return guarded(e, [&] {
    const unsigned K = cycle_rank_part(e, num_considerable);
    match_run_device(e, K, K ? e->j_index.ptr() : nullptr, /*defer=*/true);  // set up; the rounds run in cook_cycle_match_multi
    prof_collect(e);
  });
}
// the rank part of a cycle for every pool of a GPU: one flow per pool in a pool batch (above)
static const bool g_rank_batch = env_switch_on_unless_zero("COOK_RANK_BATCH");
int cook_cycle_run_rank_multi(cook_engine** engines, uint32_t n, const uint32_t* num_considerable, double* const* user_usage, int usage_is_device) {
  if (!engines || n == 0 || !num_considerable) return COOK_E_INVALID;
  for (uint32_t i = 0; i < n; ++i)
    if (!engines[i] || (user_usage && !user_usage[i])) return COOK_E_INVALID;
  for (uint32_t i = 0; i < n; ++i)  // (an engine twice: two flows would record launches against one engine's buffers)
    for (uint32_t k = 0; k < i; ++k)
      if (engines[i] == engines[k]) return COOK_E_INVALID;
  cook_engine* lead = engines[0];
  bool same_device = true;
  for (uint32_t i = 1; i < n; ++i) same_device = same_device && engines[i]->device == lead->device;
  if (n == 1 || !g_rank_batch || !same_device || g_sync_trace || tl_flow) {
    int first = COOK_OK;
    for (uint32_t i = 0; i < n; ++i) {
      int rc = cook_cycle_run_rank(engines[i], num_considerable[i]);
      if (rc == COOK_OK && user_usage) rc = cook_rank_user_usage(engines[i], user_usage[i], usage_is_device);
      if (rc != COOK_OK && first == COOK_OK) first = rc;
    }
    return first;
  }
  int flows_rc = COOK_OK;
  std::vector<std::pair<int, std::string>> flow_err(n, {COOK_OK, std::string()});  // (guarded() clears the lead's message on its way out)
  const int rc = guarded(lead, [&] {
    for (uint32_t i = 0; i < n; ++i) COOK_HIP(hipStreamSynchronize(engines[i]->stream));  // (whatever a call before this one left running)
    PoolBatch b;
    b.lead = lead;
    b.stream = lead->stream;
    b.flows.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
      cook_engine* e = engines[i];
      double* uu = user_usage ? user_usage[i] : nullptr;
      const uint32_t nc = num_considerable[i];
      b.flows[i].e = e;
      b.flows[i].body = [e, nc, uu, usage_is_device] {
        const unsigned K = cycle_rank_part(e, nc);
        if (uu) rank_user_usage(e, uu, usage_is_device != 0);
        match_run_device(e, K, K ? e->j_index.ptr() : nullptr, /*defer=*/true);
      };
    }
    StageTimer tr(lead, 0, &lead->rank_ms);
    flows_rc = batch_run(b);
    for (uint32_t i = 0; i < n; ++i)
      if (b.flows[i].rc != COOK_OK) flow_err[i] = {b.flows[i].rc, engines[i]->err};
    tr.stop();
    for (uint32_t i = 1; i < n; ++i) engines[i]->rank_ms = lead->rank_ms;  // one joint sequence of launches
    lead->batch_stats[0] = n, lead->batch_stats[1] = b.launches, lead->batch_stats[2] = b.grouped, lead->batch_stats[3] = b.singles,
    lead->batch_stats[4] = b.syncs;
    prof_collect(lead);
  });
  for (uint32_t i = 0; i < n; ++i)
    if (flow_err[i].first != COOK_OK) engines[i]->err = flow_err[i].second;  // every engine whose flow failed keeps its own message
  return rc != COOK_OK ? rc : flows_rc;
}
int cook_rank_pool_usage_multi(cook_engine** engines, uint32_t n, cook_usage* out) {
  if (!engines || n == 0 || !out) return COOK_E_INVALID;
  for (uint32_t i = 0; i < n; ++i) {
    if (!engines[i]) return COOK_E_INVALID;
    for (uint32_t k = 0; k < i; ++k)
      if (engines[i] == engines[k]) return COOK_E_INVALID;
  }
  cook_engine* lead = engines[0];
  bool same_device = true;
  for (uint32_t i = 1; i < n; ++i) same_device = same_device && engines[i]->device == lead->device;
  if (n == 1 || !g_rank_batch || !same_device || g_sync_trace || tl_flow) {
    int first = COOK_OK;
    for (uint32_t i = 0; i < n; ++i) {
      const int rc = cook_rank_pool_usage(engines[i], &out[i]);
      if (rc != COOK_OK && first == COOK_OK) first = rc;
    }
    return first;
  }
  int flows_rc = COOK_OK;
  std::vector<std::pair<int, std::string>> flow_err(n, {COOK_OK, std::string()});
  const int rc = guarded(lead, [&] {
    for (uint32_t i = 0; i < n; ++i) COOK_HIP(hipStreamSynchronize(engines[i]->stream));
    PoolBatch b;
    b.lead = lead;
    b.stream = lead->stream;
    b.flows.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
      cook_engine* e = engines[i];
      cook_usage* o = &out[i];
      b.flows[i].e = e;
      b.flows[i].body = [e, o] { rank_pool_usage(e, o); };
    }
    flows_rc = batch_run(b);
    for (uint32_t i = 0; i < n; ++i)
      if (b.flows[i].rc != COOK_OK) flow_err[i] = {b.flows[i].rc, engines[i]->err};
  });
  for (uint32_t i = 0; i < n; ++i)
    if (flow_err[i].first != COOK_OK) engines[i]->err = flow_err[i].second;
  return rc != COOK_OK ? rc : flows_rc;
}
int cook_cycle_match_multi(cook_engine** engines, uint32_t n) {
  if (!engines || n == 0 || !engines[0]) return COOK_E_INVALID;
  for (uint32_t i = 0; i < n; ++i)  // (an engine twice: its rounds would be run twice over one set of buffers)
    for (uint32_t k = 0; k < i; ++k)
      if (engines[i] && engines[i] == engines[k]) return COOK_E_INVALID;
  cook_engine* lead = engines[0];
  return guarded(lead, [&] {
    StageTimer tm(lead, 2, &lead->match_ms);
    // the pools that are placed by class-ordered best fit (classfit.hpp): ONE launch, a workgroup per pool
    {
      std::vector<cook_engine*> cf;
      for (uint32_t i = 0; i < n; ++i) {
        if (!engines[i] || engines[i]->device != lead->device) lead->fail(COOK_E_INVALID, "cook_cycle_match_multi: engines must share one device");
        if (engines[i]->has_deferred_cf) cf.push_back(engines[i]);
      }
      if (!cf.empty()) cf_run(lead, cf.data(), (unsigned)cf.size(), lead->stream);
    }
    // served walkers (one persistent walker workgroup per pool beside serve launches); lockstep launches when switched off, for more
    // pools than a served call takes, or to finish a served match that gave up
    if (!(served_enabled() && match_rounds_served(engines, n))) match_rounds_multi(engines, n);
    tm.stop();
    for (uint32_t i = 1; i < n; ++i) engines[i]->match_ms = lead->match_ms;  // one joint sequence of launches
    prof_collect(lead);
  });
}
int cook_cycle_fetch(cook_engine* e, uint32_t* ranked, uint32_t* n_ranked, int32_t* job_to_offer, uint32_t* n_considered,
                     uint8_t* head_matched) {
  return guarded(e, [&] {
    rank_fetch(e, ranked, n_ranked, nullptr);
    if (n_considered) *n_considered = e->cycle_considered;
    match_fetch(e, e->cycle_considered, job_to_offer, nullptr, head_matched);
  });
}

int cook_considerable(cook_engine* e, const cook_queue* q, const cook_user_state* us, uint32_t num_considerable, uint32_t* out_idx,
                      uint32_t* n_out, uint32_t* rate_limited, uint32_t* passed) {
  if (n_out) *n_out = 0;
  return guarded(e, [&] {
    if (!q || !n_out || (!out_idx && num_considerable && q->n)) e->fail(COOK_E_INVALID, "cook_considerable: null queue / outputs");
    const unsigned n = q->n;
    if (n && (!q->cpus || !q->mem || !q->user)) e->fail(COOK_E_INVALID, "cook_considerable: the queue needs cpus, mem, user");
    ConsBufs& c = cons_bufs(e);
    cons_stage_users(e, c, us);
    for (unsigned i = 0; i < n; ++i)
      if (q->user[i] >= c.U) e->fail(COOK_E_INVALID, "cook_considerable: user id out of range");
    h2d(e, c.q_cpus, q->cpus, n);
    h2d(e, c.q_mem, q->mem, n);
    if (q->gpus) h2d(e, c.q_gpus, q->gpus, n);
    h2d(e, c.q_user, q->user, n);
    if (q->eligible) h2d(e, c.q_elig, q->eligible, n);
    cons_run_device(e, c, n, c.q_cpus.ptr(), c.q_mem.ptr(), q->gpus ? (const double*)c.q_gpus.ptr() : (const double*)nullptr,
                    c.q_user.ptr(), q->eligible ? (const uint8_t*)c.q_elig.ptr() : (const uint8_t*)nullptr, num_considerable);
    if (c.n_result) copy_async(e, out_idx, c.result, (size_t)c.n_result * 4, hipMemcpyDeviceToHost);
    if (rate_limited && c.U) copy_async(e, rate_limited, c.rate_limited.ptr(), (size_t)c.U * 4, hipMemcpyDeviceToHost);
    if (passed && c.U) copy_async(e, passed, c.passed.ptr(), (size_t)c.U * 4, hipMemcpyDeviceToHost);
    sync(e);
    *n_out = c.n_result;
    prof_collect(e);
  });
}
int cook_cycle_set_considerable(cook_engine* e, const cook_user_state* us, const uint8_t* eligible_by_pending) {
  return guarded(e, [&] {
    ConsBufs& c = cons_bufs(e);
    if (!us) {
      c.cycle_on = false;
      return;
    }
    if (!e->rank_staged) e->fail(COOK_E_STATE, "cook_cycle_set_considerable before cook_cycle_stage");
    cons_stage_users(e, c, us);
    if (c.U < e->U) e->fail(COOK_E_INVALID, "cook_cycle_set_considerable: fewer users than the staged rank input");
    c.has_elig_by_pending = eligible_by_pending != nullptr;
    if (eligible_by_pending) {
      h2d(e, c.elig_by_pending, eligible_by_pending, e->n_pending);
      sync(e);
    }
    c.cycle_on = true;
  });
}
int cook_cycle_fetch_considerable(cook_engine* e, uint32_t* rank_pos, uint32_t* n_out) {
  if (n_out) *n_out = 0;
  return guarded(e, [&] {
    if (!e->match_done) e->fail(COOK_E_STATE, "cook_cycle_fetch_considerable before cook_cycle_run");
    const unsigned K = e->cycle_considered;
    if (e->cb && e->cb->cycle_on) {
      if (K && rank_pos) copy_async(e, rank_pos, e->cb->result, (size_t)K * 4, hipMemcpyDeviceToHost);
      sync(e);
    } else if (rank_pos) {
      for (unsigned k = 0; k < K; ++k) rank_pos[k] = k;
    }
    if (n_out) *n_out = K;
  });
}

int cook_rebalance_stage(cook_engine* e, const cook_tasks* running, const uint8_t* running_attrs_cached, const cook_jobs* pending,
                         const int64_t* pending_job_id, const int32_t* pending_priority, const cook_users* users,
                         const cook_host_spare* spare, const cook_offers* host_attrs, const cook_groups* groups,
                         const cook_rebalance_params* params) {
  return guarded(e, [&] {
    rebalance_stage(e, rebal_bufs(e), running, running_attrs_cached, pending, pending_job_id, pending_priority, users, spare, host_attrs,
                    groups, params);
  });
}
int cook_rebalance_run(cook_engine* e) {
  return guarded(e, [&] {
    RebalBufs& b = rebal_bufs(e);
    StageTimer t(e, 0, &b.ms);
    rebalance_run(e, b);
    t.stop();
    prof_collect(e);
  });
}
int cook_rebalance_fetch(cook_engine* e, cook_preemption* decisions, uint32_t* n_decisions, uint32_t* preempted, uint32_t* n_preempted,
                         double* pending_dru) {
  if (n_decisions) *n_decisions = 0;
  if (n_preempted) *n_preempted = 0;
  return guarded(e, [&] { rebalance_fetch(e, rebal_bufs(e), decisions, n_decisions, preempted, n_preempted, pending_dru); });
}
int cook_rebalance(cook_engine* e, const cook_tasks* running, const uint8_t* running_attrs_cached, const cook_jobs* pending,
                   const int64_t* pending_job_id, const int32_t* pending_priority, const cook_users* users, const cook_host_spare* spare,
                   const cook_offers* host_attrs, const cook_groups* groups, const cook_rebalance_params* params,
                   cook_preemption* decisions, uint32_t* n_decisions, uint32_t* preempted, uint32_t* n_preempted, double* pending_dru) {
  if (n_decisions) *n_decisions = 0;  // "no decisions" on any error path
  if (n_preempted) *n_preempted = 0;
  int rc = cook_rebalance_stage(e, running, running_attrs_cached, pending, pending_job_id, pending_priority, users, spare, host_attrs,
                                groups, params);
  if (rc) return rc;
  rc = cook_rebalance_run(e);
  if (rc) return rc;
  return cook_rebalance_fetch(e, decisions, n_decisions, preempted, n_preempted, pending_dru);
}
int cook_rebalance_timing(cook_engine* e, double* ms) {
  if (!e || !ms) return COOK_E_INVALID;
  *ms = e->rb ? e->rb->ms : 0.0;
  return COOK_OK;
}

int cook_match_explain(cook_engine* e, const uint32_t* job_pos, uint32_t n, uint32_t* counts) {
  return guarded(e, [&] {
    match_explain(e, explain_bufs(e), job_pos, n, counts);
    prof_collect(e);
  });
}
int cook_match_metrics(cook_engine* e, cook_cycle_metrics* out, uint32_t* user_considerable, uint32_t* user_matched, uint32_t n_users,
                       int64_t* job_gpus_by_model, int64_t* offer_gpus_by_model, uint32_t n_gpu_models) {
  return guarded(e, [&] {
    match_metrics(e, explain_bufs(e), out, user_considerable, user_matched, n_users, job_gpus_by_model, offer_gpus_by_model, n_gpu_models);
    prof_collect(e);
  });
}

int cook_offers_stage(cook_engine* e, const cook_nodes* nodes, const cook_pods* pods, const cook_offer_params* params) {
  return guarded(e, [&] { offers_stage(e, offer_bufs(e), nodes, pods, params); });
}
int cook_offers_run(cook_engine* e) {
  return guarded(e, [&] {
    OfferBufs& b = offer_bufs(e);
    StageTimer t(e, 0, &b.ms);
    offers_run(e, b);
    t.stop();
    prof_collect(e);
  });
}
int cook_offers_fetch(cook_engine* e, cook_node_offers* offers, uint32_t* n_offers, uint8_t* node_status, cook_offer_totals* totals,
                      int64_t* gpu_capacity_by_model, int64_t* gpu_consumed_by_model, double* disk_capacity_by_type,
                      double* disk_consumed_by_type) {
  if (n_offers) *n_offers = 0;
  return guarded(e, [&] {
    offers_fetch(e, offer_bufs(e), offers, n_offers, node_status, totals, gpu_capacity_by_model, gpu_consumed_by_model,
                 disk_capacity_by_type, disk_consumed_by_type);
  });
}
int cook_offers_build(cook_engine* e, const cook_nodes* nodes, const cook_pods* pods, const cook_offer_params* params,
                      cook_node_offers* offers, uint32_t* n_offers, uint8_t* node_status, cook_offer_totals* totals,
                      int64_t* gpu_capacity_by_model, int64_t* gpu_consumed_by_model, double* disk_capacity_by_type,
                      double* disk_consumed_by_type) {
  if (n_offers) *n_offers = 0;  // "no offers" on any error path
  int rc = cook_offers_stage(e, nodes, pods, params);
  if (rc) return rc;
  rc = cook_offers_run(e);
  if (rc) return rc;
  return cook_offers_fetch(e, offers, n_offers, node_status, totals, gpu_capacity_by_model, gpu_consumed_by_model, disk_capacity_by_type,
                           disk_consumed_by_type);
}
int cook_offers_timing(cook_engine* e, double* ms) {
  if (!e || !ms) return COOK_E_INVALID;
  *ms = e->ofb ? e->ofb->ms : 0.0;
  return COOK_OK;
}

int cook_last_timing(cook_engine* e, double* rank_ms, double* match_ms) {
  if (!e) return COOK_E_INVALID;
  if (rank_ms) *rank_ms = e->rank_ms;
  if (match_ms) *match_ms = e->match_ms;
  return COOK_OK;
}
int cook_match_stats(cook_engine* e, uint32_t out[16]) {
  if (!e || !out) return COOK_E_INVALID;
  const WinCtl& c = e->last_ctl;
  out[0] = c.rounds;
  out[1] = c.matched;
  out[2] = c.stop_list;
  out[3] = c.stop_full;
  out[4] = c.stop_group;
  out[5] = c.stop_window;
  out[6] = c.segments;
  out[7] = c.head;
  out[8] = (uint32_t)(c.t_setup / 100ull);  // microseconds
  out[9] = (uint32_t)(c.t_seq / 100ull);
  out[10] = c.touched_sum;
  out[11] = c.visited_sum;
  out[12] = out[13] = out[14] = out[15] = 0u;
  return COOK_OK;
}
int cook_match_stats_ex(cook_engine* e, uint32_t* out, uint32_t cap) {
  if (!e || !out) return COOK_E_INVALID;
  uint32_t v[COOK_MATCH_STATS_EX_N];
  for (auto& x : v) x = 0u;
  const int rc = cook_match_stats(e, v);
  if (rc) return rc;
  const WinCtl& c = e->last_ctl;
  v[16] = c.trunc_lists;
  v[17] = e->served.mode, v[18] = e->served.pools, v[19] = e->served.iterations, v[20] = e->served.empty_iterations;
  v[21] = e->served.pools_served, v[22] = (uint32_t)(e->served.latch_wait_ms * 1000.0), v[23] = e->served.fell_back, v[24] = e->served.servers;
  v[26] = e->upd_us, v[27] = e->upd_sync_us, v[28] = e->upd_allocs;
  for (unsigned k = 0; k < 8u; ++k)
    if (e->upd_phase_us[k] > v[30]) v[29] = k, v[30] = e->upd_phase_us[k];
  for (unsigned k = 0; k < 5u; ++k) v[32 + k] = e->batch_stats[k];
  v[37] = e->last_form, v[38] = e->cf_inelig;
  if (e->last_form == 3u)
    for (unsigned k = 0; k < 24u; ++k) v[40 + k] = e->cf_stats[k];
  if (g_guard) {  // COOK_GUARD=1: look at this engine's bands now; the count is process-wide and includes buffers already freed
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (DBuf* b : engine_bufs(e)) b->check_guard();
    v[25] = g_guard_hits.load();
  }
  uint32_t n = 0;
  for (; n < cap && n < (uint32_t)COOK_MATCH_STATS_EX_N; ++n) out[n] = v[n];
  return (int)n;
}
COOK_EMU_EXTRA_EXPORTS
int cook_set_profiling(cook_engine* e, int enabled) {
  if (!e) return COOK_E_INVALID;
  e->profiling = enabled != 0;
  e->kstats.clear();
  return COOK_OK;
}
int cook_kernel_timings(cook_engine* e, const char** names, double* ms, uint32_t* launches, uint32_t cap) {
  if (!e) return COOK_E_INVALID;
  e->kstat_names.clear();
  for (auto& kv : e->kstats) e->kstat_names.push_back(kv.first);
  uint32_t n = 0;
  for (auto& nm : e->kstat_names) {
    if (n >= cap) break;
    names[n] = nm.c_str();
    ms[n] = e->kstats[nm].ms;
    launches[n] = e->kstats[nm].launches;
    ++n;
  }
  return (int)n;
}

}  // extern "C"

// classfit_model.cpp — CPU MODEL of the class-ordered best-fit placement (`match_algo = 3`): the gate VERDICT r5 asks for
// before any HIP is written.  TEST TOOL: it includes the oracle's translation unit for the constraint evaluators and is compared
// with the oracle's placement; nothing under cook_amd/ uses it.
//
// What it models, wave-step for wave-step (the HIP kernel in cook_amd/csrc/classfit.hpp follows the same structure):
//   * every resource of the call in FIXED POINT (u32; the call is eligible only if cpus / mem of jobs and offers are multiples of
//     2^-k below 2^31 / 2^k): feasibility and the order inside a class are exact integer arithmetic;
//   * offers of equal (lease + running) totals and equal gpu kind are a CLASS: cpuMemBinPacker's fitness (config.clj:108) is
//     1 - E / (2 Tc Tm) + (c / Tc + m / Tm) / 2 with E = free_c * Tm + free_m * Tc, so inside a class the order is the same for every job;
//   * per class a sorted array (E ascending) cut into chunks of 64 positions; per chunk 8 level summaries (max free mem among members
//     with free cpus >= level), upper bounds, tightened when a scan comes back empty;
//   * the offers a placement touched live in an OVERLAY of 64 lanes (one wave's registers); when it is full of live offers the epoch
//     ends: they are merged back into their classes' arrays;
//   * a job = ballot over chunk summaries -> scan of the first candidate chunk -> first feasible lane; overlay lanes evaluated in
//     parallel; exchange; commit.  Offers inside a 2^-37 band of the best go through the oracle's literal expression.
// Statistics: scans per walked job on the critical path (max over the waves), opens, epochs, empty scans, literal evaluations.
#include "../../oracle/cook_oracle.cpp"

namespace cf {

constexpr int LV = 8, CH = 64, OVN = 64;
constexpr double BAND = 1.0 / 137438953472.0;  // 2^-37

struct Cls {
  uint32_t Tc, Tm, gk, off, n, chunk0, nchunks;
  int wave;
  double iTc, iTm;
  uint64_t dE;
};
struct OvEntry {
  bool valid;
  uint32_t id, cls, fc, fm;
};
enum { ST_WALKED, ST_PRESETTLED, ST_MATCHED, ST_OV_WIN, ST_OPEN, ST_GPU_PLACE, ST_EPOCHS, ST_SCANS, ST_EMPTY_SCANS, ST_CRIT_SCANS, ST_BAND, ST_LITERAL, ST_TIGHTEN,
       ST_SLOW_JOBS, ST_SLOW_SCANS, ST_CLASSES, ST_CHUNKS, ST_WAVES, ST_WALKED_UNMATCHED, ST_CRIT_SCANS_UNMATCHED, ST_MAX_CRIT, ST_DEAD_DROPS, ST_OPEN_DEAD,
       ST_KC, ST_KM, ST_GPU_UNMATCHED, ST_CRIT_STEPS_MATCHED, ST_CRIT_STEPS_UNMATCHED, ST_SLOW_UNMATCHED, ST_N };

static int fixed_shift(const std::vector<const double*>& cols, const std::vector<uint32_t>& ns) {
  for (int k = 0; k <= 20; ++k) {
    bool ok = true;
    for (size_t a = 0; a < cols.size() && ok; ++a)
      for (uint32_t i = 0; i < ns[a] && ok; ++i) {
        const double v = cols[a] ? std::ldexp(cols[a][i], k) : 0.0;
        if (!(v >= 0.0) || !(v < 2147483648.0) || v != std::floor(v)) ok = false;
      }
    if (ok) return k;
  }
  return -1;
}

struct Model {
  const cook_params* p;
  const cook_jobs* j;
  const cook_offers* o;
  const cook_groups* g;
  uint32_t K, M;
  int kc, km;
  std::vector<uint32_t> Lc, Lm, oTc, oTm, ocls;  // per offer (fixed point): lease, totals, class
  std::vector<uint32_t> cur_fc, cur_fm;          // per offer: the TRUE current free values (bookkeeping for checks / fail bits)
  std::vector<uint8_t> occupied;
  std::vector<Cls> cls;
  std::vector<uint32_t> pfc, pfm, pid;  // position arrays
  std::vector<uint8_t> present;
  std::vector<std::array<uint32_t, LV>> lv;  // per chunk: 1 + max free mem among present members with free cpus >= t[i]; 0 = none
  std::vector<uint32_t> chunk_cls;
  uint32_t t[LV];
  uint32_t cmin, cmax, mmin;
  OvEntry ov[OVN];
  std::vector<std::pair<uint32_t, double>> sigs;  // gpu kinds: (model, count)
  MatchState st;                                  // the oracle's view of the call (constraint evaluators read it)
  std::set<uint32_t> reserved;
  uint64_t stats[ST_N] = {0};
  int nwaves = 0;

  uint32_t fx(double v, int k) const { return (uint32_t)std::ldexp(v, k); }
  uint64_t E_of(const Cls& c, uint32_t fc, uint32_t fm) const { return (uint64_t)fc * c.Tm + (uint64_t)fm * c.Tc; }
  int level_of(uint32_t c) const {
    int L = 0;
    for (int i = 1; i < LV; ++i)
      if (t[i] <= c) L = i;
    return L;
  }
  bool dead(uint32_t fc, uint32_t fm) const { return fc < cmin || fm < mmin; }
  bool occ_now(uint32_t v) const { return ((o->run_count ? o->run_count[v] : 0) + st.acount[v]) != 0; }

  void tighten(uint32_t ch) {
    const Cls& c = cls[chunk_cls[ch]];
    const uint32_t p0 = c.off + (ch - c.chunk0) * CH, p1 = std::min(c.off + c.n, p0 + CH);
    for (int i = 0; i < LV; ++i) lv[ch][i] = 0;
    for (uint32_t q = p0; q < p1; ++q)
      if (present[q] && !(c.gk != 0 && occ_now(pid[q])))  // (a gpu host takes a gpu job only while nothing runs on it: constraints.clj:122-157)
        for (int i = 0; i < LV; ++i)
          if (pfc[q] >= t[i]) lv[ch][i] = std::max(lv[ch][i], pfm[q] + 1);
    stats[ST_TIGHTEN]++;
  }

  // (re)build the class arrays from `members` (offer ids with their current free values): sort by (class, E, id)
  void build(std::vector<std::array<uint32_t, 3>>& mem) {  // {id, fc, fm}
    std::sort(mem.begin(), mem.end(), [&](const std::array<uint32_t, 3>& a, const std::array<uint32_t, 3>& b) {
      const uint32_t ca = ocls[a[0]], cb = ocls[b[0]];
      if (ca != cb) return ca < cb;
      const uint64_t ea = E_of(cls[ca], a[1], a[2]), eb = E_of(cls[cb], b[1], b[2]);
      if (ea != eb) return ea < eb;
      return a[0] < b[0];
    });
    const size_t N = mem.size();
    pfc.assign(N, 0), pfm.assign(N, 0), pid.assign(N, 0), present.assign(N, 1);
    for (auto& c : cls) c.n = 0;
    for (size_t q = 0; q < N; ++q) {
      pid[q] = mem[q][0], pfc[q] = mem[q][1], pfm[q] = mem[q][2];
      cls[ocls[mem[q][0]]].n++;
    }
    uint32_t off = 0, ch = 0;
    chunk_cls.clear();
    for (size_t k = 0; k < cls.size(); ++k) {
      cls[k].off = off, cls[k].chunk0 = ch, cls[k].nchunks = (cls[k].n + CH - 1) / CH;
      off += cls[k].n, ch += cls[k].nchunks;
      for (uint32_t x = 0; x < cls[k].nchunks; ++x) chunk_cls.push_back((uint32_t)k);
    }
    lv.assign(ch, {});
    for (uint32_t x = 0; x < ch; ++x) tighten(x);
    stats[ST_TIGHTEN] -= ch;
  }

  double literal(uint32_t v, uint32_t k) const {  // the oracle's expression on the oracle's own state (cook_oracle.cpp match_impl)
    const double c = j->cpus[k], m = j->mem[k];
    const double rc = o->run_cpus ? o->run_cpus[v] : 0.0, rm = o->run_mem ? o->run_mem[v] : 0.0;
    return ((rc + st.ac[v] + c) / (o->cpus[v] + rc) + (rm + st.am[v] + m) / (o->mem[v] + rm)) / 2.0;
  }
  double approx(const Cls& c, uint32_t fc, uint32_t fm, uint32_t jc, uint32_t jm) const {
    return 1.0 - ((double)(fc - jc) * c.iTc + (double)(fm - jm) * c.iTm) * 0.5;
  }
  bool cons_ok(uint32_t k, uint32_t v, bool slow) const {
    if (!slow) return true;
    return job_constraints_pass(p, j, k, o, v, st, reserved) && group_constraint_pass(j, k, o, v, g, st);
  }

  int run(const uint32_t* reserved_hosts, uint32_t n_reserved, int32_t* j2o, uint32_t* fail_code, uint8_t* head_matched) {
    K = j->n, M = o->n;
    const double ge = p->good_enough_fitness;
    if (ge < 1.0) return 1;
    if (j->ports || j->scalars) {
      bool x = false;
      for (uint32_t k = 0; k < K && j->ports; ++k) x = x || j->ports[k] > 0;
      for (uint32_t s2 = 0; j->scalars && s2 < j->n_scalars; ++s2)
        for (uint32_t k = 0; k < K; ++k) x = x || (j->scalars[(size_t)s2 * K + k] == j->scalars[(size_t)s2 * K + k]);
      if (x) return 1;
    }
    if (o->gpu_slots > 1 || M == 0 || K == 0 || M > 65000) return 1;
    kc = fixed_shift({j->cpus, o->cpus, o->run_cpus}, {K, M, o->run_cpus ? M : 0});
    km = fixed_shift({j->mem, o->mem, o->run_mem}, {K, M, o->run_mem ? M : 0});
    if (kc < 0 || km < 0) return 1;
    stats[ST_KC] = kc, stats[ST_KM] = km;
    for (uint32_t k = 0; k < K; ++k) {
      if (!(j->cpus[k] > 0.0 || j->mem[k] > 0.0)) return 1;
      if (j->gpus && !(j->gpus[k] >= 0.0)) return 1;
    }
    reserved = std::set<uint32_t>(reserved_hosts, reserved_hosts + n_reserved);
    st.aports.assign(M, 0), st.ascalar.assign((size_t)M * COOK_MAX_SCALARS, 0.0), st.ac.assign(M, 0.0), st.am.assign(M, 0.0), st.acount.assign(M, 0);
    if (g) st.ghost.resize(g->n), st.gattr.resize(g->n);
    // offers -> fixed point, classes
    Lc.resize(M), Lm.resize(M), oTc.resize(M), oTm.resize(M), ocls.resize(M), occupied.assign(M, 0);
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, uint32_t> cmap;
    std::map<std::tuple<uint32_t, uint32_t>, uint32_t> shape_n;
    for (uint32_t v = 0; v < M; ++v) {
      const double rc = o->run_cpus ? o->run_cpus[v] : 0.0, rm = o->run_mem ? o->run_mem[v] : 0.0;
      shape_n[std::make_tuple((uint32_t)std::ldexp(o->cpus[v] + rc, kc), (uint32_t)std::ldexp(o->mem[v] + rm, km))]++;
    }
    for (uint32_t v = 0; v < M; ++v) {
      const double rc = o->run_cpus ? o->run_cpus[v] : 0.0, rm = o->run_mem ? o->run_mem[v] : 0.0;
      const double Tc = o->cpus[v] + rc, Tm = o->mem[v] + rm;
      if (!(Tc > 0.0) || !(Tm > 0.0) || !(std::ldexp(Tc, kc) < 2147483648.0) || !(std::ldexp(Tm, km) < 2147483648.0)) return 1;
      Lc[v] = fx(o->cpus[v], kc), Lm[v] = fx(o->mem[v], km), oTc[v] = fx(Tc, kc), oTm[v] = fx(Tm, km);
      uint32_t gk = 0;
      const bool k8s = o->k8s && o->k8s[v];
      if (k8s && o->gpu_model && o->gpu_model[v] != 0) {
        const std::pair<uint32_t, double> sg{o->gpu_model[v], o->gpu_count ? o->gpu_count[v] : 0.0};
        auto it = std::find(sigs.begin(), sigs.end(), sg);
        if (it == sigs.end()) sigs.push_back(sg), it = sigs.end() - 1;
        gk = 1 + (uint32_t)(it - sigs.begin());
        occupied[v] = (o->run_count ? o->run_count[v] : 0) != 0;
      }
      const uint32_t nsub = (shape_n[std::make_tuple(oTc[v], oTm[v])] + 3583) / 3584;  // a wave's lanes hold at most 64 chunks
      auto key = std::make_tuple(oTc[v], oTm[v], gk, gk == 0 ? v % nsub : 0u);
      auto it = cmap.find(key);
      if (it == cmap.end()) {
        Cls c{};
        c.Tc = oTc[v], c.Tm = oTm[v], c.gk = gk;
        c.iTc = 1.0 / (double)c.Tc, c.iTm = 1.0 / (double)c.Tm;
        c.dE = (uint64_t)(BAND * 2.0 * (double)c.Tc * (double)c.Tm);
        it = cmap.emplace(key, (uint32_t)cls.size()).first;
        cls.push_back(c);
      }
      ocls[v] = it->second;
    }
    // levels
    cmin = 0xFFFFFFFFu, cmax = 0, mmin = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < K; ++k) {
      cmin = std::min(cmin, fx(j->cpus[k], kc)), cmax = std::max(cmax, fx(j->cpus[k], kc)), mmin = std::min(mmin, fx(j->mem[k], km));
    }
    for (int i = 0; i < LV; ++i) t[i] = cmin + (uint32_t)(((uint64_t)i * (cmax - cmin)) / (LV - 1));
    cur_fc = Lc, cur_fm = Lm;
    std::vector<std::array<uint32_t, 3>> mem;
    for (uint32_t v = 0; v < M; ++v) mem.push_back({v, Lc[v], Lm[v]});
    build(mem);
    // waves: classes of more than 64 chunks cannot be held by one wave's lanes (the kernel would split them); plain classes get a wave
    // each, greedily packed otherwise
    {
      std::vector<uint32_t> load;  // chunks held per wave; a plain class has a wave to itself, gpu classes are packed
      for (auto& c : cls) {
        if (c.nchunks > 64) return 1;
        int w = -1;
        if (c.gk != 0)
          for (size_t x = 0; x < load.size() && w < 0; ++x)
            if (load[x] + c.nchunks <= 64) w = (int)x;
        if (w < 0) {
          load.push_back(0);
          w = (int)load.size() - 1;
        }
        load[w] += c.gk != 0 ? c.nchunks : 64;
        c.wave = w;
      }
      nwaves = (int)load.size();
    }
    stats[ST_CLASSES] = cls.size(), stats[ST_CHUNKS] = lv.size(), stats[ST_WAVES] = nwaves;
    for (auto& e : ov) e.valid = false;
    uint32_t matched = 0;
    // slow flags / job kinds
    const bool offers_dyn = o->max_tasks != nullptr;
    auto job_slow = [&](uint32_t k) {
      if (offers_dyn || !reserved.empty()) return true;
      if (j->novel_off && j->novel_off[k + 1] > j->novel_off[k]) return true;
      if (j->eq_off && j->eq_off[k + 1] > j->eq_off[k]) return true;
      if (g && j->group && j->group[k] != COOK_NONE_U32 && g->type[j->group[k]] != 0) return true;
      if (j->disk_request && j->disk_request[k] >= 0) return true;
      if (j->est_end_ms && j->est_end_ms[k] != 0) return true;
      if (j->ckpt_location && j->ckpt_location[k] != 0) return true;
      return false;
    };
    auto job_gk = [&](uint32_t k) -> uint32_t {
      const double jg = j->gpus ? j->gpus[k] : 0.0;
      if (!(jg > 0)) return 0;
      const std::pair<uint32_t, double> sg{j->gpu_model ? j->gpu_model[k] : 0u, jg};
      if (sg.first == 0) return 0xFFFFu;  // (get model->count nil 0) = 0 != jg
      auto it = std::find(sigs.begin(), sigs.end(), sg);
      return it == sigs.end() ? 0xFFFFu : 1 + (uint32_t)(it - sigs.begin());
    };
    auto any_room = [&](uint32_t jc, uint32_t jm, uint32_t gk_rel, bool relevant) {  // exact: an offer of a (ir)relevant class with room
      for (uint32_t v = 0; v < M; ++v)
        if ((cls[ocls[v]].gk == gk_rel) == relevant && cur_fc[v] >= jc && cur_fm[v] >= jm) return true;
      return false;
    };
    uint32_t min_fc_all = 0xFFFFFFFFu, min_fm_all = 0xFFFFFFFFu;
    for (uint32_t v = 0; v < M; ++v) min_fc_all = std::min(min_fc_all, Lc[v]), min_fm_all = std::min(min_fm_all, Lm[v]);

    auto epoch_end = [&] {
      std::vector<std::array<uint32_t, 3>> mem2;
      for (size_t q = 0; q < pid.size(); ++q)
        if (present[q]) mem2.push_back({pid[q], pfc[q], pfm[q]});
      for (auto& e : ov)
        if (e.valid) {
          if (!dead(e.fc, e.fm)) mem2.push_back({e.id, e.fc, e.fm});
          e.valid = false;
        }
      build(mem2);
      stats[ST_EPOCHS]++;
    };

    for (uint32_t base = 0; base < K; base += 64) {
      const uint32_t bn = std::min<uint32_t>(64, K - base);
      // batch pre-check against the CURRENT summaries (upper bounds): R = a relevant chunk / overlay lane may have room
      std::vector<uint8_t> R(bn, 0);
      for (uint32_t b = 0; b < bn; ++b) {
        const uint32_t k = base + b, jc = fx(j->cpus[k], kc), jm = fx(j->mem[k], km), gk = job_gk(k);
        const int L = level_of(jc);
        for (uint32_t ch = 0; ch < lv.size() && !R[b]; ++ch)
          if (cls[chunk_cls[ch]].gk == gk && lv[ch][L] > jm) R[b] = 1;
        for (auto& e : ov)
          if (e.valid && cls[e.cls].gk == gk && e.fc >= jc && e.fm >= jm) R[b] = 1;
      }
      for (uint32_t b = 0; b < bn; ++b) {
        const uint32_t k = base + b, jc = fx(j->cpus[k], kc), jm = fx(j->mem[k], km), gk = job_gk(k);
        const int L = level_of(jc);
        const bool slow = job_slow(k);
        uint32_t fail = 0;
        if (!(jc <= min_fc_all && jm <= min_fm_all)) fail |= 1u;
        int32_t win = -1;
        if (!R[b]) {
          stats[ST_PRESETTLED]++;
          // no relevant offer that could take it has room: every offer with room fails a constraint (class kind / occupied gpu host)
          bool room_any = false;
          for (uint32_t v = 0; v < M; ++v) {
            if (!(cur_fc[v] >= jc && cur_fm[v] >= jm)) continue;
            room_any = true;
            if (cls[ocls[v]].gk == gk && !(gk != 0 && occ_now(v))) std::abort();  // the pre-check must be conservative
          }
          if (room_any) fail |= 2u;
        } else {
          stats[ST_WALKED]++;
          if (slow) stats[ST_SLOW_JOBS]++;
          struct Cand {
            double fa;
            uint32_t v;
            int src;  // -1 overlay lane, else position
            uint32_t where;
          };
          std::vector<Cand> cands;
          bool room_cons_fail = false;
          // overlay wave
          for (int l = 0; l < OVN; ++l) {
            auto& e = ov[l];
            if (!e.valid || cls[e.cls].gk != gk || e.fc < jc || e.fm < jm) continue;
            if (!cons_ok(k, e.id, slow)) {
              room_cons_fail = true;
              continue;
            }
            cands.push_back({approx(cls[e.cls], e.fc, e.fm, jc, jm), e.id, -1, (uint32_t)l});
          }
          // class waves
          std::vector<uint32_t> scans_w(nwaves, 0);
          for (size_t ci = 0; ci < cls.size(); ++ci) {
            const Cls& c = cls[ci];
            if (c.gk != gk) continue;
            bool found = false;
            uint64_t E0 = 0;
            for (uint32_t x = 0; x < c.nchunks; ++x) {
              const uint32_t ch = c.chunk0 + x;
              const uint32_t p0 = c.off + x * CH, p1 = std::min(c.off + c.n, p0 + CH);
              if (!found && !(lv[ch][L] > jm)) continue;                   // pruned by the level summary
              if (found && E_of(c, pfc[p0], pfm[p0]) > E0 + c.dE) break;  // the band of the first feasible offer ended with the previous chunk
              scans_w[c.wave]++;
              stats[ST_SCANS]++;
              if (slow) stats[ST_SLOW_SCANS]++;
              bool any_lane_room = false, past_band = false;
              for (uint32_t q = p0; q < p1; ++q) {
                if (!present[q] || pfc[q] < jc || pfm[q] < jm) continue;
                any_lane_room = true;
                const uint32_t v = pid[q];
                const bool occ = c.gk != 0 && ((o->run_count ? o->run_count[v] : 0) + st.acount[v]) != 0;
                if (occ || !cons_ok(k, v, slow)) {
                  room_cons_fail = true;
                  continue;
                }
                const uint64_t E = E_of(c, pfc[q], pfm[q]);
                if (!found) {
                  found = true, E0 = E;
                } else if (E > E0 + c.dE) {
                  past_band = true;
                  break;
                }
                cands.push_back({approx(c, pfc[q], pfm[q], jc, jm), v, (int)q, ch});
              }
              if (!any_lane_room && !found) {
                stats[ST_EMPTY_SCANS]++;
                tighten(ch);
              }
              if (found && past_band) break;
            }
          }
          uint32_t crit = 0;
          for (auto s2 : scans_w) crit = std::max(crit, s2);
          if (!cands.empty()) {
            double fmax = -1.0;
            for (auto& cd : cands) fmax = std::max(fmax, cd.fa);
            std::vector<Cand> top;
            for (auto& cd : cands)
              if (cd.fa >= fmax - BAND) top.push_back(cd);
            Cand w = top[0];
            if (top.size() > 1) {
              stats[ST_BAND]++;
              double lb = -1.0;
              for (auto& cd : top) {
                stats[ST_LITERAL]++;
                const double lf = literal(cd.v, k);
                if (lf > lb || (lf == lb && cd.v < w.v)) lb = lf, w = cd;
              }
            }
            win = (int32_t)w.v;
            const Cls& c = cls[ocls[w.v]];
            // commit
            st.ac[w.v] += j->cpus[k], st.am[w.v] += j->mem[k], st.acount[w.v] += 1;
            cur_fc[w.v] -= jc, cur_fm[w.v] -= jm;
            if (std::ldexp((double)(Lc[w.v] - cur_fc[w.v]), -kc) != st.ac[w.v] || std::ldexp((double)(Lm[w.v] - cur_fm[w.v]), -km) != st.am[w.v]) std::abort();
            min_fc_all = std::min(min_fc_all, cur_fc[w.v]), min_fm_all = std::min(min_fm_all, cur_fm[w.v]);
            if (g && j->group && j->group[k] != COOK_NONE_U32) {
              const uint32_t gi = j->group[k];
              st.ghost[gi].push_back(o->host[w.v]);
              st.gattr[gi].push_back(g->type[gi] >= 2 ? offer_attr(o, w.v, g->attr_key[gi]) : 0);
            }
            if (w.src < 0) {
              stats[ST_OV_WIN]++;
              auto& e = ov[w.where];
              e.fc -= jc, e.fm -= jm;
              if (dead(e.fc, e.fm)) e.valid = false, stats[ST_DEAD_DROPS]++;
            } else if (c.gk != 0) {  // gpu classes: in place, the chunk's summaries exact again
              stats[ST_GPU_PLACE]++;
              pfc[w.src] -= jc, pfm[w.src] -= jm;
              tighten(w.where);
              stats[ST_TIGHTEN]--;
            } else {
              stats[ST_OPEN]++;
              present[w.src] = 0;
              {  // summaries stay EXACT: recompute the chunk's levels the removed member was the maximum of
                bool was_max = false;
                for (int i = 0; i < LV; ++i)
                  if (pfc[w.src] >= t[i] && lv[w.where][i] == pfm[w.src] + 1) was_max = true;
                if (was_max) tighten(w.where);
              }
              const uint32_t nfc = pfc[w.src] - jc, nfm = pfm[w.src] - jm;
              if (dead(nfc, nfm)) {
                stats[ST_OPEN_DEAD]++;
              } else {
                int l = -1;
                for (int x = 0; x < OVN && l < 0; ++x)
                  if (!ov[x].valid) l = x;
                if (l < 0) {
                  epoch_end();
                  l = 0;
                }
                ov[l] = OvEntry{true, w.v, ocls[w.v], nfc, nfm};
              }
            }
            stats[ST_CRIT_SCANS] += crit;
            stats[ST_CRIT_STEPS_MATCHED] += 3 + crit;  // job / ballot, scans, exchange, commit
            ++matched;
          } else {
            stats[ST_WALKED_UNMATCHED]++;
            stats[ST_CRIT_SCANS_UNMATCHED] += crit;
            stats[ST_CRIT_STEPS_UNMATCHED] += 2 + crit;
            if (gk != 0) stats[ST_GPU_UNMATCHED]++;
            if (slow) stats[ST_SLOW_UNMATCHED]++;
            // unmatched: every offer with room fails a constraint (the kernel: exact per-level maxima over all offers + the overlay lanes)
            if (room_cons_fail || any_room(jc, jm, gk, false) || any_room(jc, jm, gk, true)) fail |= 2u;
          }
          stats[ST_MAX_CRIT] = std::max<uint64_t>(stats[ST_MAX_CRIT], crit);
        }
        j2o[k] = win;
        if (fail_code) fail_code[k] = win >= 0 ? 0u : (fail ? fail : 8u);
      }
    }
    stats[ST_MATCHED] = matched;
    if (head_matched) *head_matched = (matched == 0 || (K > 0 && j2o[0] >= 0)) ? 1 : 0;
    return 0;
  }
};

}  // namespace cf

extern "C" int classfit_model_match(const cook_params* p, const cook_jobs* j, const cook_offers* o, const cook_groups* g, const uint32_t* reserved_hosts,
                                    uint32_t n_reserved, int32_t* job_to_offer, uint32_t* fail_code, uint8_t* head_matched, uint64_t* stats, uint32_t n_stats) {
  cf::Model m;
  m.p = p, m.j = j, m.o = o, m.g = g;
  const int rc = m.run(reserved_hosts, n_reserved, job_to_offer, fail_code, head_matched);
  for (uint32_t i = 0; i < n_stats && i < (uint32_t)cf::ST_N; ++i) stats[i] = m.stats[i];
  return rc;
}

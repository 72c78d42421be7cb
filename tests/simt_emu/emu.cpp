// tests/simt_emu/emu.cpp — fiber scheduler of the SIMT emulator (TEST INFRASTRUCTURE; see hip/hip_runtime.h).
#include <chrono>

#include "hip/hip_runtime.h"

namespace emu {

static State g_state;
State& S() { return g_state; }

static const size_t kStack = 256 * 1024;

static void fiber_entry() {
  State& s = S();
  Fiber* f = s.cur;
  (*s.body)();
  f->done = true;
  ++s.events;
  // a finished thread no longer takes part in rendezvous (terminated waves leave the barrier count)
  Group& w = s.waves[f->tid >> 6];
  if (w.size) {
    --w.size;
    if (w.size && w.count == w.size) {
      w.count = 0;
      ++w.gen;
      ++s.events;
    }
  }
  if (s.block.size) {
    --s.block.size;
    if (s.block.size && s.block.count == s.block.size) {
      s.block.count = 0;
      ++s.block.gen;
      ++s.events;
    }
  }
  swapcontext(&f->ctx, &s.main_ctx);
}

void arrive(Group& g) {
  State& s = S();
  Fiber* f = s.cur;
  const uint64_t my = g.gen;
  if (++g.count >= g.size) {
    g.count = 0;
    ++g.gen;
    ++s.events;
    return;
  }
  while (g.gen == my) {
    swapcontext(&f->ctx, &s.main_ctx);
    s.cur = f;
  }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  State& s = S();
  const unsigned nthreads = block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > 1024) {
    std::fprintf(stderr, "emu: bad block size %u\n", nthreads);
    std::abort();
  }
  if (s.fibers.size() < nthreads) s.fibers.resize(nthreads);
  const unsigned nwaves = (nthreads + 63) / 64;
  s.waves.assign(nwaves, Group());
  s.xchg.assign((size_t)nwaves * 64, 0);
  s.blockDim_ = block;
  s.gridDim_ = grid;
  s.body = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.blockIdx_ = dim3(bx, by, bz);
        s.block = Group();
        s.block.size = nthreads;
        for (unsigned w = 0; w < nwaves; ++w) {
          s.waves[w] = Group();
          s.waves[w].size = std::min(64u, nthreads - w * 64);
        }
        for (unsigned t = 0; t < nthreads; ++t) {
          Fiber& f = s.fibers[t];
          if (!f.stack) f.stack = (char*)std::malloc(kStack);
          f.done = false;
          f.site = "";
          f.tid = t;
          f.tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = kStack;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, fiber_entry, 0);
        }
        unsigned remaining = nthreads;
        unsigned spins = 0;
        while (remaining) {
          const uint64_t ev0 = s.events;
          for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = s.fibers[t];
            if (f.done) continue;
            s.cur = &f;
            swapcontext(&s.main_ctx, &f.ctx);
            if (f.done) --remaining;
          }
          // deadlock guard: every live fiber is parked and no rendezvous can complete
          if (s.events == ev0) {
            if (++spins > 1000) {
              std::fprintf(stderr, "emu: deadlock (threads waiting at a rendezvous not all threads reach)\n");
              for (unsigned t = 0; t < nthreads; ++t)
                if (!s.fibers[t].done) std::fprintf(stderr, "  live thread %u last site: %s\n", t, s.fibers[t].site);
              std::abort();
            }
          } else {
            spins = 0;
          }
        }
      }
  s.body = nullptr;
  s.cur = nullptr;
}

}  // namespace emu

double emu_now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

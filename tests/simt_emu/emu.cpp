// tests/simt_emu/emu.cpp — fiber scheduler of the SIMT emulator (TEST INFRASTRUCTURE; see hip/hip_runtime.h).
#include <chrono>

#include "hip/hip_runtime.h"

namespace emu {

static State g_state;
State& S() { return g_state; }

static const size_t kStack = 256 * 1024;

#ifdef EMU_FAST_SWITCH
// save the callee-saved registers on the current stack, publish its pointer, adopt the other stack, restore, return into it
extern "C" void emu_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch, .-emu_switch
)");
static void fiber_entry();
extern "C" void emu_fiber_start() {  // first activation of a fiber: never returns (fiber_entry ends with a switch to main)
  fiber_entry();
  std::abort();
}
static inline void to_main(Fiber* f, State& s) { emu_switch(&f->sp, s.main_sp); }
static inline void to_fiber(State& s, Fiber& f) { emu_switch(&s.main_sp, f.sp); }
static inline void prepare(Fiber& f) {
  // stack image emu_switch pops: r15 r14 r13 r12 rbx rbp, then the return address; at the entry of emu_fiber_start the stack
  // pointer must be 8 below a 16-byte boundary, as after a call
  uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
  void** sp = (void**)(top - 64);
  for (int i = 0; i < 6; ++i) sp[i] = nullptr;
  sp[6] = (void*)&emu_fiber_start;
  sp[7] = nullptr;
  f.sp = sp;
}
#else
static void fiber_entry();
static inline void to_main(Fiber* f, State& s) { swapcontext(&f->ctx, &s.main_ctx); }
static inline void to_fiber(State& s, Fiber& f) { swapcontext(&s.main_ctx, &f.ctx); }
static inline void prepare(Fiber& f) {
  getcontext(&f.ctx);
  f.ctx.uc_stack.ss_sp = f.stack;
  f.ctx.uc_stack.ss_size = kStack;
  f.ctx.uc_link = nullptr;
  makecontext(&f.ctx, fiber_entry, 0);
}
#endif

static void fiber_entry() {
  State& s = S();
  Fiber* f = s.cur;
  (*s.body)();
  f->done = true;
  ++s.events;
  // a finished thread no longer takes part in rendezvous (terminated waves leave the barrier count)
  BlockCtx& b = *f->blk;
  Group& w = b.waves[f->tid >> 6];
  if (w.size) {
    --w.size;
    if (w.size && w.count == w.size) {
      w.count = 0;
      ++w.gen;
      ++s.events;
    }
  }
  if (b.block.size) {
    --b.block.size;
    if (b.block.size && b.block.count == b.block.size) {
      b.block.count = 0;
      ++b.block.gen;
      ++s.events;
    }
  }
  to_main(f, s);
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
    to_main(f, s);
    s.cur = f;
  }
}

void yield() {
  State& s = S();
  Fiber* f = s.cur;
  to_main(f, s);
  s.cur = f;
}

char* block_lds(size_t bytes) {
  BlockCtx& b = B();
  if (b.lds.size() < bytes) b.lds.resize(bytes);  // first caller of the block sizes it (every thread asks for the same amount)
  return b.lds.data();
}

// run the blocks [b0, b1) of s.blocks concurrently until all their threads have finished
static void run_blocks(State& s, unsigned b0, unsigned b1, unsigned nthreads, dim3 block) {
  const unsigned nwaves = (nthreads + 63) / 64;
  const size_t nf = (size_t)(b1 - b0) * nthreads;
  if (s.fibers.size() < nf) s.fibers.resize(nf);
  for (unsigned bi = b0; bi < b1; ++bi) {
    BlockCtx& b = s.blocks[bi];
    b.block = Group();
    b.block.size = nthreads;
    b.waves.assign(nwaves, Group());
    b.xchg.assign((size_t)nwaves * 64, 0);
    for (unsigned w = 0; w < nwaves; ++w) b.waves[w].size = std::min(64u, nthreads - w * 64);
    for (unsigned t = 0; t < nthreads; ++t) {
      Fiber& f = s.fibers[(size_t)(bi - b0) * nthreads + t];
      if (!f.stack) f.stack = (char*)std::malloc(kStack);
      f.done = false;
      f.site = "";
      f.tid = t;
      f.tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
      f.blk = &b;
      prepare(f);
    }
  }
  size_t remaining = nf;
  unsigned spins = 0;
  while (remaining) {
    const uint64_t ev0 = s.events;
    for (size_t x = 0; x < nf; ++x) {
      Fiber& f = s.fibers[x];
      if (f.done) continue;
      s.cur = &f;
      to_fiber(s, f);
      if (f.done) --remaining;
    }
    // deadlock guard: every live fiber is parked (rendezvous or spin loop) and nothing changed during a whole pass
    if (s.events == ev0) {
      if (++spins > 2000) {
        std::fprintf(stderr, "emu: deadlock (threads waiting at a rendezvous not all threads reach, or spinning on a word nobody writes)\n");
        unsigned shown = 0;  // one line per run of threads at the same site
        for (size_t x = 0; x < nf && shown < 40;) {
          if (s.fibers[x].done) {
            ++x;
            continue;
          }
          size_t y = x;
          while (y + 1 < nf && !s.fibers[y + 1].done && s.fibers[y + 1].site == s.fibers[x].site && s.fibers[y + 1].blk == s.fibers[x].blk) ++y;
          std::fprintf(stderr, "  live threads %u..%u of block %u last site: %s\n", s.fibers[x].tid, s.fibers[y].tid, s.fibers[x].blk->bidx.x, s.fibers[x].site);
          ++shown;
          x = y + 1;
        }
        std::abort();
      }
    } else {
      spins = 0;
    }
  }
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body, bool coop) {
  State& s = S();
  const unsigned nthreads = block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > 1024) {
    std::fprintf(stderr, "emu: bad block size %u\n", nthreads);
    std::abort();
  }
  s.blockDim_ = block;
  s.gridDim_ = grid;
  s.body = &body;
  const unsigned nblocks = grid.x * grid.y * grid.z;
  if (coop) {
    s.blocks.assign(nblocks, BlockCtx());
    unsigned bi = 0;
    for (unsigned bz = 0; bz < grid.z; ++bz)
      for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) s.blocks[bi++].bidx = dim3(bx, by, bz);
    run_blocks(s, 0, nblocks, nthreads, block);
  } else {  // one block after another, in one reusable context
    if (s.blocks.size() != 1) s.blocks.assign(1, BlockCtx());
    for (unsigned bz = 0; bz < grid.z; ++bz)
      for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
          s.blocks[0].bidx = dim3(bx, by, bz);
          run_blocks(s, 0, 1, nthreads, block);
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

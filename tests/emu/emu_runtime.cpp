// TEST INFRASTRUCTURE: fiber scheduler of the host emulation of HIP kernels (see include/hip/hip_runtime.h).
// Every lane of a workgroup is a fiber with its own 64 KB stack; one OS thread runs them round-robin.
#include "hip/hip_runtime.h"

// x86-64 System V context switch: save the callee-saved registers on the current stack, publish its stack pointer,
// adopt the other stack and restore.  A fresh fiber's stack is pre-loaded so that the final `ret` enters its trampoline.
asm(R"(
.text
.globl fsr_emu_switch
.type fsr_emu_switch,@function
fsr_emu_switch:
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
.size fsr_emu_switch,.-fsr_emu_switch
)");

namespace emu {

Sched& sched() {
  static thread_local Sched s;
  return s;
}

static constexpr size_t kStack = 64 * 1024;

void fiber_yield() {
  Sched& s = sched();
  const int n = (int)s.fibers.size();
  int nxt = s.cur;
  for (int k = 0; k < n; ++k) {
    nxt = (nxt + 1 == n) ? 0 : nxt + 1;
    if (!s.fibers[nxt].done) break;
  }
  if (nxt == s.cur) {
    if (s.fibers[s.cur].done) {   // the last fiber finished: back to the launcher
      void* dummy;
      fsr_emu_switch(&dummy, s.main_sp);
    }
    fprintf(stderr, "emu: deadlock -- a lane waits at a barrier the rest of its workgroup never reaches\n");
    abort();
  }
  Fiber& from = s.fibers[s.cur];
  s.cur = nxt;
  fsr_emu_switch(&from.sp, s.fibers[nxt].sp);
}

static void fiber_main() {
  Sched& s = sched();
  (*s.body)();
  Sched& s2 = sched();
  s2.fibers[s2.cur].done = true;
  --s2.live;
  if (s2.live == 0) {
    void* dummy;
    fsr_emu_switch(&dummy, s2.main_sp);   // workgroup finished
  }
  fiber_yield();                          // never returns to a finished fiber
  abort();
}

void run_workgroups(dim3 grid, dim3 block, const std::function<void()>& body) {
  Sched& s = sched();
  const int nthreads = (int)(block.x * block.y * block.z);
  if ((int)s.fibers.size() != nthreads) {
    for (auto& f : s.fibers) free(f.stack);
    s.fibers.assign(nthreads, Fiber());
    for (auto& f : s.fibers) f.stack = (char*)aligned_alloc(64, kStack);
  }
  BlockState blk;
  blk.nthreads = nthreads;
  blk.bar.total = nthreads;
  blk.dyn_smem = (char*)aligned_alloc(256, 160 * 1024);
  for (int w = 0; w < nthreads / 64; ++w) {
    WaveState* ws = new WaveState();
    ws->bar.total = 64;
    blk.waves.push_back(ws);
  }
  s.body = &body;
  s.grid = grid;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        for (int t = 0; t < nthreads; ++t) {
          Fiber& f = s.fibers[t];
          f.done = false;
          f.c.blk = &blk;
          f.c.lin = t;
          f.c.lane = t & 63;
          f.c.wave = t >> 6;
          f.c.w = blk.waves[t >> 6];
          f.c.bdim = block;
          f.c.gdim = grid;
          f.c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          f.c.bid = dim3(bx, by, bz);
          // initial frame: six zeroed callee-saved slots, then the trampoline as the return address; after the `ret`
          // the stack pointer is 8 modulo 16, as at any function entry
          uintptr_t top = ((uintptr_t)(f.stack + kStack) & ~(uintptr_t)15) - 8;
          void** sp = (void**)top;
          *--sp = (void*)&fiber_main;
          for (int r = 0; r < 6; ++r) *--sp = nullptr;
          f.sp = (void*)sp;
        }
        s.live = nthreads;
        s.cur = 0;
        fsr_emu_switch(&s.main_sp, s.fibers[0].sp);   // returns when the last lane of the workgroup is done
      }
  for (auto* ws : blk.waves) delete ws;
  free(blk.dyn_smem);
}

}  // namespace emu

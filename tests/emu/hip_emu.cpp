/* tests/emu/hip_emu.cpp -- fiber scheduler of the SIMT emulator (TEST TOOL ONLY, see hip_emu.h) */
#include "hip_emu.h"
#undef threadIdx
#undef blockIdx
#undef blockDim
#undef gridDim

namespace emu {

State &S() {
  static thread_local State s; /* one emulated device per host thread */
  return s;
}

static void fiber_entry() {
  State &s = S();
  (*s.body)();
  Fiber &f = s.fibers[s.cur];
  f.done = true;
  s.alive--;
  s.waves[f.lin / WAVE].alive--;
  swapcontext(&f.ctx, &s.sched);
}

void yield() {
  State &s = S();
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

/* hardware s_barrier counts only waves that have not exited; same here with threads */
void block_barrier() {
  State &s = S();
  unsigned g = s.bar_gen;
  s.bar_count++;
  while (s.bar_gen == g) {
    if (s.bar_count >= s.alive) {
      s.bar_count = 0;
      s.bar_gen++;
      break;
    }
    yield();
  }
}

void wave_rendezvous() {
  State &s = S();
  State::WaveX &w = s.waves[wave_id()];
  unsigned g = w.gen;
  w.count++;
  while (w.gen == g) {
    if (w.count >= w.alive) {
      w.count = 0;
      w.gen++;
      break;
    }
    yield();
  }
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body) {
  State &s = S();
  unsigned nt = block.x * block.y * block.z;
  if (nt == 0 || nt > 1024) {
    fprintf(stderr, "emu: bad block size %u\n", nt);
    abort();
  }
  if (s.fibers.size() < nt) {
    size_t old = s.fibers.size();
    s.fibers.resize(nt);
    for (size_t i = old; i < nt; i++) s.fibers[i].stack = (char *)malloc(STACK);
  }
  s.bdim = block;
  s.gdim = grid;
  s.body = &body;
  s.nthreads = nt;
  s.waves.assign((nt + WAVE - 1) / WAVE, State::WaveX());
  std::vector<char> lds(shmem + 16);
  s.dyn_lds = lds.data();
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        s.bidx = {bx, by, bz};
        s.bar_count = 0;
        s.alive = nt;
        for (auto &w : s.waves) {
          w.count = 0;
          w.alive = 0;
          memset(w.valid, 0, sizeof w.valid);
        }
        for (unsigned t = 0; t < nt; t++) {
          Fiber &f = s.fibers[t];
          f.done = false;
          f.lin = t;
          f.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
          s.waves[t / WAVE].alive++;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = STACK;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, fiber_entry, 0);
        }
        unsigned long spins = 0;
        while (s.alive > 0) {
          for (unsigned t = 0; t < nt; t++) {
            Fiber &f = s.fibers[t];
            if (f.done) continue;
            s.cur = (int)t;
            s.tidx = f.tid;
            swapcontext(&s.sched, &f.ctx);
          }
          if (++spins > 50000000ul) {
            fprintf(stderr, "emu: deadlock (divergent barrier / wave op?)\n");
            abort();
          }
        }
      }
  s.cur = -1;
  s.body = nullptr;
}

}  // namespace emu

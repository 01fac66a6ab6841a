/* tests/emu/hip_emu.cpp -- fiber scheduler of the SIMT emulator (TEST TOOL ONLY, see hip_emu.h) */
/* fibers hop between malloc'ed stacks with _setjmp / _longjmp: glibc's fortified longjmp would take that for a
 * corrupted frame */
#undef _FORTIFY_SOURCE
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

/* swapcontext saves and restores the signal mask with a system call each way; a kernel test switches fibers
 * millions of times.  So a fiber is ENTERED once through its ucontext and from then on leaves and resumes through
 * _setjmp / _longjmp (registers only). */
/* (Under AddressSanitizer every longjmp makes the runtime re-scan the stack it leaves -- the sanitizer build of
 * tests/test_gsbatch.py went from one minute to twenty-four -- so that build keeps swapcontext.) */
#if defined(__SANITIZE_ADDRESS__)
#define EMU_UCONTEXT_ONLY 1
#else
#define EMU_UCONTEXT_ONLY 0
#endif
static void resume(Fiber &to) { /* never returns */
  if (to.started) _longjmp(to.jb, 1);
  to.started = true;
  setcontext(&to.ctx);
  abort();
}
/* leave the running context `from_*` for fiber `to` (or, to == nullptr, for the scheduler) */
static void hop(ucontext_t *from_ctx, jmp_buf &from_jb, Fiber *to) {
  State &s = S();
#if EMU_UCONTEXT_ONLY
  (void)from_jb;
  swapcontext(from_ctx, to ? &to->ctx : &s.sched);
#else
  (void)from_ctx;
  if (_setjmp(from_jb) == 0) {
    if (to) resume(*to);
    _longjmp(s.sched_jb, 1);
  }
#endif
}

static void fiber_entry() {
  State &s = S();
  (*s.body)();
  Fiber &f = s.fibers[s.cur];
  f.done = true;
  s.alive--;
  s.waves[f.lin / WAVE].alive--;
  hop(&f.ctx, f.jb, nullptr);
}

void yield() {
  State &s = S();
  Fiber &f = s.fibers[s.cur];
  hop(&f.ctx, f.jb, nullptr);
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

static void switch_to(unsigned t);

void wave_rendezvous() {
  State &s = S();
  const unsigned me = s.fibers[s.cur].lin, wv = me / WAVE, l = me % WAVE;
  State::WaveX &w = s.waves[wv];
  unsigned g = w.gen;
  w.count++;
  w.arrived[l] = g + 1;
  while (w.gen == g) {
    if (w.count >= w.alive) {
      w.count = 0;
      w.gen++;
      break;
    }
    /* hand the CPU to a lane of this wave that has not arrived yet (a full round of the block's fibers per wait
     * made wave-level code quadratic); a lane blocked elsewhere gives it back through the scheduler */
    bool handed = false;
    for (unsigned k = 1; k < (unsigned)WAVE && !handed; k++) {
      const unsigned t = wv * WAVE + (l + k) % WAVE;
      if (t < s.nthreads && !s.fibers[t].done && w.arrived[t % WAVE] != g + 1) {
        switch_to(t);
        handed = true;
      }
    }
    if (!handed) yield();
  }
}

/* run fiber t now; the caller stays runnable and is resumed by the scheduler's round-robin (or by a sibling) */
static void switch_to(unsigned t) {
  State &s = S();
  Fiber &me = s.fibers[s.cur], &to = s.fibers[t];
  s.cur = (int)t;
  s.tidx = to.tid;
  hop(&me.ctx, me.jb, &to);
  /* resumed: the scheduler (or a sibling) has set s.cur / s.tidx for us again */
}

uint64_t quad_exchange(uint64_t v, unsigned sel) {
  State &s = S();
  const unsigned me = s.fibers[s.cur].lin, q = me / 4, l = me & 3u;
  State::QuadX &x = s.quads[q];
  const unsigned g = x.gen, buf = g & 1u;
  unsigned live = 0;
  for (unsigned i = 0; i < 4; i++)
    if (q * 4 + i < s.nthreads && !s.fibers[q * 4 + i].done) live++;
  x.slot[buf][l] = v;
  x.count++;
  unsigned next = l;
  while (x.gen == g) {
    if (x.count >= live) {
      x.count = 0;
      x.gen++;
      break;
    }
    /* hand over to the next live sibling; if it is blocked elsewhere it comes straight back via the scheduler */
    bool handed = false;
    for (unsigned k = 1; k <= 3 && !handed; k++) {
      const unsigned t = q * 4 + ((next + k) & 3u);
      if (t < s.nthreads && !s.fibers[t].done && t != me) {
        next = (next + k) & 3u;
        switch_to(t);
        handed = true;
      }
    }
    if (!handed) yield();
  }
  return x.slot[buf][sel & 3u];
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
  s.quads.assign((nt + 3) / 4, State::QuadX());
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
          memset(w.arrived, 0, sizeof w.arrived);
          w.gen = 0;
        }
        for (auto &x : s.quads) x.count = 0;
        for (unsigned t = 0; t < nt; t++) {
          Fiber &f = s.fibers[t];
          f.done = false;
          f.started = false;
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
            hop(&s.sched, s.sched_jb, &f);
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

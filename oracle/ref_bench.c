/*
 * oracle/ref_bench.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The CPU baseline of bench.py: the UNMODIFIED reference header, #included from where
 * it lies, run over frames by a plain pthread loop -- "all host cores via a pthread loop over frames in the harness only,
 * reference code untouched".  Every thread owns preallocated, pre-touched buffers, waits behind a barrier and then
 * runs the configs[1] chain gs_blur(r) -> gs_sobel (zeroed dst) -> gs_otsu_threshold -> gs_threshold
 * (grayskull.h:268, :306, :205, :225) on `frames_per_thread` frames.  No allocation, no Python, no page faults inside
 * the timed region (round 3 timed 256 Python threads through ctypes / numpy: 3 % parallel efficiency).
 *
 * Build recipe: oracle/Makefile -> oracle/_ref/libgs_ref_bench.so (git-ignored, travels to the GPU box).
 */
#define _POSIX_C_SOURCE 200809L
#define _XOPEN_SOURCE 600
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "grayskull.h" /* GS_API keeps its default: static inline, the reference's native build */

struct job {
  unsigned w, h, radius, nframes, tid, nsrc;
  const uint8_t *frames; /* nsrc input frames, shared, read-only */
  uint8_t *a, *b;
  pthread_barrier_t *gate;
  unsigned long long check;
};

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *p) {
  struct job *j = (struct job *)p;
  const size_t fb = (size_t)j->w * j->h;
  struct gs_image a = {j->w, j->h, j->a}, b = {j->w, j->h, j->b};
  pthread_barrier_wait(j->gate);
  for (unsigned f = 0; f < j->nframes; f++) {
    struct gs_image s = {j->w, j->h, (uint8_t *)j->frames + fb * ((j->tid + f) % j->nsrc)};
    gs_blur(a, s, j->radius);
    memset(j->b, 0, fb);
    gs_sobel(b, a);
    const uint8_t t = gs_otsu_threshold(b);
    gs_threshold(b, t);
    j->check += t + j->b[fb / 2 + j->w / 2];
  }
  pthread_barrier_wait(j->gate);
  return NULL;
}

/* seconds between the release of the start barrier and the last thread's finish; < 0 on failure.
 * checksum (optional): something that depends on every frame's result, so nothing can be optimised away */
double ref_chain_frames(unsigned n_threads, unsigned frames_per_thread, unsigned w, unsigned h, unsigned radius,
                        const uint8_t *frames, unsigned nsrc, unsigned long long *checksum) {
  if (!n_threads || !frames_per_thread || !frames || !nsrc) return -1.0;
  const size_t fb = (size_t)w * h;
  struct job *jobs = (struct job *)calloc(n_threads, sizeof *jobs);
  pthread_t *th = (pthread_t *)calloc(n_threads, sizeof *th);
  pthread_barrier_t gate;
  if (!jobs || !th || pthread_barrier_init(&gate, NULL, n_threads + 1)) return -1.0;
  unsigned made = 0;
  for (unsigned i = 0; i < n_threads; i++) {
    jobs[i] = (struct job){w, h, radius, frames_per_thread, i, nsrc, frames, NULL, NULL, &gate, 0};
    jobs[i].a = (uint8_t *)malloc(fb), jobs[i].b = (uint8_t *)malloc(fb);
    if (!jobs[i].a || !jobs[i].b) break;
    memset(jobs[i].a, 1, fb), memset(jobs[i].b, 1, fb); /* pages exist before the clock starts */
    if (pthread_create(&th[i], NULL, worker, &jobs[i])) break;
    made++;
  }
  double dt = -1.0;
  if (made == n_threads) {
    pthread_barrier_wait(&gate); /* every thread exists and waits: the clock starts with the work */
    const double t0 = now();
    pthread_barrier_wait(&gate);
    dt = now() - t0;
  } else {
    /* could not start them all: let the ones that exist run through (they need the barrier count) -- give up instead */
    exit(3);
  }
  unsigned long long c = 0;
  for (unsigned i = 0; i < n_threads; i++) {
    pthread_join(th[i], NULL);
    c += jobs[i].check;
    free(jobs[i].a), free(jobs[i].b);
  }
  if (checksum) *checksum = c;
  pthread_barrier_destroy(&gate);
  free(jobs), free(th);
  return dt;
}

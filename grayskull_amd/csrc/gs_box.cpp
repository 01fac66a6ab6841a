/*
 * gs_box.cpp -- the sliding-box kernels (k_box.h) in their own translation unit: k_box16r is one kernel per radius
 * (1 .. 16) and mode, each unrolled 2 r + 1 rows deep -- 32 instantiations that take longer to compile than the rest
 * of the library together, so the Makefile builds this file beside gs_api.cpp.
 */
#include "k_box.h"

#ifndef GS_BOXR_MAX
#define GS_BOXR_MAX 16 /* largest radius built as a register-ring kernel (the sanitizer build of the tests uses 3) */
#endif

namespace gs {

unsigned box_ring_max() { return GS_BOXR_MAX; }

/* mode 0: gs_blur, 1: gs_adaptive_threshold (constant c).  ring_radius in 1 .. 16: the register-ring kernel for exactly
 * that radius (callers check its preconditions: w >= 32, h >= 2 r + 1, MODE 1: |c| < 2^30); 0: the any-radius kernel. */
void launch_box(int mode, unsigned ring_radius, dim3 grid, unsigned threads, hipStream_t st, uint8_t *dst, const uint8_t *src, unsigned w,
                unsigned h, unsigned T, size_t frame_bytes, unsigned r, int c) {
  switch (ring_radius) {
#define GS_BOXR(RR)                                                                                              \
  case RR:                                                                                                       \
    if (mode == 0) GS_LAUNCH((k_box16r<0, RR>), grid, dim3(threads), 0, st, dst, src, w, h, T, frame_bytes, c);      \
    else GS_LAUNCH((k_box16r<1, RR>), grid, dim3(threads), 0, st, dst, src, w, h, T, frame_bytes, c);                \
    break;
    GS_BOXR(1) GS_BOXR(2) GS_BOXR(3)
#if GS_BOXR_MAX >= 16
    GS_BOXR(4) GS_BOXR(5) GS_BOXR(6) GS_BOXR(7) GS_BOXR(8)
    GS_BOXR(9) GS_BOXR(10) GS_BOXR(11) GS_BOXR(12) GS_BOXR(13) GS_BOXR(14) GS_BOXR(15) GS_BOXR(16)
#endif
#undef GS_BOXR
    default:
      if (mode == 0) GS_LAUNCH(k_box16<0>, grid, dim3(threads), 0, st, dst, src, w, h, T, frame_bytes, r, c);
      else GS_LAUNCH(k_box16<1>, grid, dim3(threads), 0, st, dst, src, w, h, T, frame_bytes, r, c);
  }
}

/* ragged rows under a ring kernel: columns (w & ~15) - 16 .. w - 1 (k_box_edge), one wave per band of T rows */
void launch_box_edge(int mode, dim3 grid, hipStream_t st, uint8_t *dst, const uint8_t *src, unsigned w, unsigned h, unsigned T,
                     size_t frame_bytes, unsigned r, int c) {
  if (mode == 0) GS_LAUNCH(k_box_edge<0>, grid, dim3(64), 0, st, dst, src, w, h, T, frame_bytes, r, c);
  else GS_LAUNCH(k_box_edge<1>, grid, dim3(64), 0, st, dst, src, w, h, T, frame_bytes, r, c);
}

/* blocks of `threads` threads of that kernel a CU holds at once (256 threads: register-limited, 4 up to r = 9 / 7, then
 * 3, then 2; narrower blocks: also the 8 x 19.7 KB of LDS): the launcher sizes its bands for whole rounds of 256 x
 * this many blocks */
unsigned box_blocks_per_cu(int mode, unsigned ring_radius, unsigned threads) {
#ifdef GS_EMU
  (void)mode, (void)ring_radius, (void)threads;
  return 4;
#else
  static unsigned cache[3][2][17]; /* 0: not asked yet; racing threads write the same value */
  if (ring_radius > GS_BOXR_MAX) ring_radius = 0;
  unsigned &slot = cache[threads <= 64 ? 0 : threads <= 128 ? 1 : 2][mode ? 1 : 0][ring_radius];
  if (slot) return slot;
  const void *fn = nullptr;
  switch (ring_radius) {
#define GS_BOXR(RR) case RR: fn = mode == 0 ? (const void *)k_box16r<0, RR> : (const void *)k_box16r<1, RR>; break;
    GS_BOXR(1) GS_BOXR(2) GS_BOXR(3)
#if GS_BOXR_MAX >= 16
    GS_BOXR(4) GS_BOXR(5) GS_BOXR(6) GS_BOXR(7) GS_BOXR(8)
    GS_BOXR(9) GS_BOXR(10) GS_BOXR(11) GS_BOXR(12) GS_BOXR(13) GS_BOXR(14) GS_BOXR(15) GS_BOXR(16)
#endif
#undef GS_BOXR
    default: fn = mode == 0 ? (const void *)k_box16<0> : (const void *)k_box16<1>;
  }
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, (int)threads, 0) != hipSuccess || n < 1) n = 4;
  return slot = (unsigned)(n > 16 ? 16 : n);
#endif
}

}  // namespace gs

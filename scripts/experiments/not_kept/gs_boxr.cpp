/*
 * gs_boxr.cpp -- the RAGGED forms (w % 16 != 0) of the register-ring sliding-box kernels k_box16r (k_box.h), one per radius
 * and mode like the whole-strip forms of gs_box.cpp, in their own translation unit so that the two sets compile side by
 * side, and k_box_edge, which rewrites the r rightmost columns behind them.
 */
#include "k_box.h"

#ifndef GS_BOXR_MAX
#define GS_BOXR_MAX 16
#endif

namespace gs {

void launch_box_ragged(int mode, unsigned ring_radius, dim3 grid, unsigned threads, hipStream_t st, uint8_t *dst, const uint8_t *src,
                       unsigned w, unsigned h, unsigned T, size_t frame_bytes, int c) {
  switch (ring_radius) {
#define GS_BOXR(RR)                                                                                                         \
  case RR:                                                                                                                  \
    if (mode == 0) GS_LAUNCH((k_box16r<0, RR, true>), grid, dim3(threads), 0, st, dst, src, w, h, T, frame_bytes, c);         \
    else GS_LAUNCH((k_box16r<1, RR, true>), grid, dim3(threads), 0, st, dst, src, w, h, T, frame_bytes, c);                   \
    break;
    GS_BOXR(1) GS_BOXR(2) GS_BOXR(3)
#if GS_BOXR_MAX >= 16
    GS_BOXR(4) GS_BOXR(5) GS_BOXR(6) GS_BOXR(7) GS_BOXR(8)
    GS_BOXR(9) GS_BOXR(10) GS_BOXR(11) GS_BOXR(12) GS_BOXR(13) GS_BOXR(14) GS_BOXR(15) GS_BOXR(16)
#endif
#undef GS_BOXR
    default:
      fprintf(stderr, "grayskull_hip: no ragged ring kernel for radius %u\n", ring_radius);
      abort();
  }
  /* the r rightmost columns, which the kernel above divided as if they were interior pixels: one wave per 16 rows */
  const unsigned nbe = (h + kBoxEdgeRows - 1) / kBoxEdgeRows;
  if (mode == 0) GS_LAUNCH(k_box_edge<0>, dim3(1, nbe, grid.z), dim3(64), 0, st, dst, src, w, h, frame_bytes, ring_radius, c);
  else GS_LAUNCH(k_box_edge<1>, dim3(1, nbe, grid.z), dim3(64), 0, st, dst, src, w, h, frame_bytes, ring_radius, c);
}

}  // namespace gs

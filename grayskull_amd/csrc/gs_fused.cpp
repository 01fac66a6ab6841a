/*
 * gs_fused.cpp -- the fused blur+sobel+histogram kernel of gsh_edge_pipeline_batch in its own
 * translation unit, because it alone is built with -mllvm -amdgpu-sched-strategy=iterative-ilp:
 * the kernel is VALU-bound and made of long dependent packed-op chains (8 independent ones per
 * row), which that strategy interleaves where the default pads them with s_nop (+2 % measured).
 * Everything else uses hipcc's default scheduler (iterative-ilp crashes its register allocator
 * on k_integral_wave and buys nothing on the HBM-bound kernels).
 */
#include "k_fused.h"

#ifndef GS_FUSED_MINW
#define GS_FUSED_MINW 1
#endif

namespace gs {

void launch_blur_sobel_hist(unsigned radius, dim3 grid, dim3 block, hipStream_t st, uint8_t *dst,
                            const uint8_t *src, unsigned w, unsigned h, unsigned T, size_t frame_bytes,
                            unsigned *partial) {
  if (radius == 1) GS_LAUNCH(k_blur_sobel_hist16<1>, grid, block, 0, st, dst, src, w, h, T, frame_bytes, partial);
  else if (radius == 2) GS_LAUNCH(k_blur_sobel_hist16<2>, grid, block, 0, st, dst, src, w, h, T, frame_bytes, partial);
  else GS_LAUNCH(k_blur_sobel_hist16<3>, grid, block, 0, st, dst, src, w, h, T, frame_bytes, partial);
}

/* the same kernel without the histogram half (gsh_blur_sobel_batch) */
void launch_blur_sobel(unsigned radius, dim3 grid, dim3 block, hipStream_t st, uint8_t *dst, const uint8_t *src,
                       unsigned w, unsigned h, unsigned T, size_t frame_bytes, int rg) {
  unsigned *none = nullptr;
  if (rg == 1) { /* ragged rows: the tail strip anchored at w - 16 (k_strip.h) */
    if (radius == 1) GS_LAUNCH((k_blur_sobel_hist16<1, false, 1>), grid, block, 0, st, dst, src, w, h, T, frame_bytes, none);
    else if (radius == 2) GS_LAUNCH((k_blur_sobel_hist16<2, false, 1>), grid, block, 0, st, dst, src, w, h, T, frame_bytes, none);
    else GS_LAUNCH((k_blur_sobel_hist16<3, false, 1>), grid, block, 0, st, dst, src, w, h, T, frame_bytes, none);
    return;
  }
  if (rg == 2) { /* any byte phase: dword-aligned loads, realigned in registers */
    if (radius == 1) GS_LAUNCH((k_blur_sobel_hist16<1, false, 2>), grid, block, 0, st, dst, src, w, h, T, frame_bytes, none);
    else if (radius == 2) GS_LAUNCH((k_blur_sobel_hist16<2, false, 2>), grid, block, 0, st, dst, src, w, h, T, frame_bytes, none);
    else GS_LAUNCH((k_blur_sobel_hist16<3, false, 2>), grid, block, 0, st, dst, src, w, h, T, frame_bytes, none);
    return;
  }
  if (radius == 1) GS_LAUNCH((k_blur_sobel_hist16<1, false>), grid, block, 0, st, dst, src, w, h, T, frame_bytes, none);
  else if (radius == 2) GS_LAUNCH((k_blur_sobel_hist16<2, false, 0, GS_FUSED_MINW>), grid, block, 0, st, dst, src, w, h, T, frame_bytes, none);
  else GS_LAUNCH((k_blur_sobel_hist16<3, false>), grid, block, 0, st, dst, src, w, h, T, frame_bytes, none);
}

}  // namespace gs

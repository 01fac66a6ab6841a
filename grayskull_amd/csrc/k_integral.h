/*
 * k_integral.h -- gs_integral (grayskull.h:744-752): inclusive 2-D prefix sum, u8 -> u32,
 * same w x h layout as the reference (no padding row/column), arithmetic mod 2^32.
 *
 * Two passes: (1) k_integral_rows: per-row inclusive scan (wave scan on DPP-free shuffles +
 * LDS carry across the 4 waves), (2) k_integral_cols: running column sums in place,
 * coalesced across columns.  Algorithmic traffic 5 B/px (1 R + 4 W); this two-pass form
 * moves 13 B/px -- acceptable because gs_lbp_detect dominates every caller by >100x.
 *
 * k_integral_pad: copies an unpadded w x h table into the (w+1) x (h+1) zero-bordered layout
 * the cascade kernel reads (turns the x>0 / y>0 guards of gs_integral_sum, ref :754-763, into
 * plain loads).
 */
#ifndef GS_K_INTEGRAL_H
#define GS_K_INTEGRAL_H
#include "k_stencil.h"

namespace gs {

/* grid (h, n frames), block 256; each thread owns 4 consecutive px per 1024-px tile */
__global__ __launch_bounds__(256) void k_integral_rows(const uint8_t *src, unsigned w, unsigned h,
                                                       unsigned *ii) {
  __shared__ unsigned wsum[4];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const size_t row = ((size_t)blockIdx.y * h + blockIdx.x) * w;
  const uint8_t *s = src + row;
  unsigned *o = ii + row;
  const bool vec = (w % 4 == 0) && (((uintptr_t)src & 3) == 0) && (((uintptr_t)ii & 15) == 0);
  unsigned carry = 0;
  for (unsigned base = 0; base < w; base += 1024) {
    const unsigned x = base + tid * 4;
    unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    if (vec) {
      if (x < w) {
        const uint32_t d = *(const uint32_t *)(s + x);
        p0 = d & 0xff, p1 = (d >> 8) & 0xff, p2 = (d >> 16) & 0xff, p3 = d >> 24;
      }
    } else {
      if (x < w) p0 = s[x];
      if (x + 1 < w) p1 = s[x + 1];
      if (x + 2 < w) p2 = s[x + 2];
      if (x + 3 < w) p3 = s[x + 3];
    }
    p1 += p0, p2 += p1, p3 += p2;
    const unsigned inc = wave_incl_scan(p3);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned off = carry + inc - p3;
    for (unsigned k = 0; k < wv; k++) off += wsum[k];
    const unsigned tile_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (vec) {
      if (x < w) *(U4 *)(o + x) = U4{off + p0, off + p1, off + p2, off + p3};
    } else {
      if (x < w) o[x] = off + p0;
      if (x + 1 < w) o[x + 1] = off + p1;
      if (x + 2 < w) o[x + 2] = off + p2;
      if (x + 3 < w) o[x + 3] = off + p3;
    }
    carry += tile_total;
    __syncthreads();
  }
}

/* grid (ceil(w/256), n frames), block 256: thread = one column, rows top to bottom */
__global__ __launch_bounds__(256) void k_integral_cols(unsigned *ii, unsigned w, unsigned h) {
  const unsigned x = blockIdx.x * 256u + threadIdx.x;
  if (x >= w) return;
  unsigned *p = ii + (size_t)blockIdx.y * w * h + x;
  unsigned acc = 0;
  unsigned y = 0;
  for (; y + 8 <= h; y += 8) {
    unsigned v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = p[(size_t)(y + k) * w];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      acc += v[k];
      p[(size_t)(y + k) * w] = acc;
    }
  }
  for (; y < h; y++) {
    acc += p[(size_t)y * w];
    p[(size_t)y * w] = acc;
  }
}

/* grid (ceil((w+1)/64), ceil((h+1)/4), n), block (64,4) */
__global__ __launch_bounds__(256) void k_integral_pad(const unsigned *ii, unsigned w, unsigned h,
                                                      unsigned *padded) {
  const unsigned x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
  if (x > w || y > h) return;
  const unsigned v = (x && y) ? ii[(size_t)blockIdx.z * w * h + (size_t)(y - 1) * w + (x - 1)] : 0u;
  padded[(size_t)blockIdx.z * (w + 1) * (h + 1) + (size_t)y * (w + 1) + x] = v;
}

}  // namespace gs
#endif

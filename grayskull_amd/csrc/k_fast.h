/*
 * k_fast.h -- gs_fast (grayskull.h:482-534): FAST-9 score map, strict 3x3 NMS, raster emit.
 *
 * The reference walks the 16-px Bresenham ring for 16+9 steps with a signed run counter.
 * That is equivalent to "a circular run of >= 9 ring pixels of one class" (SURVEY.md 2.3,
 * verified over all 3^16 class assignments), which on a GPU is a 16-bit mask test:
 *   mm = m | m << 16;  r = mm & mm>>1;  r &= r>>2;  r &= r>>4;  r &= mm>>8;  (r & 0xffff) != 0
 * applied to the "brighter" mask and the "darker" mask.  The class tests keep the reference's
 * UNSIGNED arithmetic (`v > p + t`, else `v < p - t` with unsigned t): when p < t the darker
 * bound wraps and every non-brighter pixel counts as darker (ref :496-498).
 *
 * Algorithmic traffic is 3 B/px (score pass 1 R + 1 W, NMS pass 1 R), but the score pass is
 * VALU-bound (~140 lane-ops per pixel: 16 ring pixels x two class masks + the minimum |v - p|).
 */
#ifndef GS_K_FAST_H
#define GS_K_FAST_H
#include "k_compact.h"

namespace gs {

GS_DEV bool ring_has_run9(unsigned m) {
  const unsigned mm = m | (m << 16);
  unsigned r = mm & (mm >> 1); /* runs >= 2 */
  r &= r >> 2;                 /* >= 4 */
  r &= r >> 4;                 /* >= 8 */
  r &= mm >> 8;                /* >= 9 */
  return (r & 0xffffu) != 0;
}

/* score of the pixel whose centre value is p and ring values are v[0..15] (ref :491-513).
 * The kernel is VALU-bound, so the class masks are gathered from SIGN bits: hi - v < 0 <=> brighter,
 * v - lo < 0 <=> darker, and one v_alignbit_b32 shifts a sign bit into the mask
 * (acc = {acc, s} >> 31): 2 lane-ops per ring pixel and class instead of compare + select + or.
 * The masks come out bit-reversed (ring pixel 0 in bit 15), which a circular-run test cannot see.
 * When p < threshold the reference's unsigned `p - t` wraps and every non-brighter pixel is
 * "darker" (ref :496-498): that case is one select on the finished masks.  |v - p| is one
 * v_sad_u16 and the 16-way minimum folds into v_min3_u32. */
/* literal form of ref :491-513 in u32 arithmetic, for thresholds so large that p + t itself wraps */
GS_DEV unsigned fast_score_u32(unsigned p, const unsigned (&v)[16], unsigned threshold) {
  const unsigned hi = p + threshold, lo = p - threshold; /* u32 wrap-around on purpose */
  unsigned bright = 0, dark = 0, mind = 255;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const bool b = v[j] > hi;
    const bool d = !b && v[j] < lo;
    bright |= (unsigned)b << j;
    dark |= (unsigned)d << j;
    mind = umin(mind, absdiff_u16(v[j], p));
  }
  return (ring_has_run9(bright) || ring_has_run9(dark)) ? mind : 0u;
}

GS_DEV unsigned fast_score(unsigned p, const unsigned (&v)[16], unsigned threshold) {
  if (threshold > 0xffffff00u) return fast_score_u32(p, v, threshold); /* kernel argument: uniform */
  /* v, p <= 255; t in [256, 2^32-256] behaves like 256 (never brighter, p - t wraps): clamp so the
   * signed differences below cannot overflow */
  const int t = (int)(threshold < 256u ? threshold : 256u), hi = (int)p + t, lo = (int)p - t;
  uint32_t bright = 0, dark = 0;
  unsigned mind = 255;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    bright = alignbit(bright, (uint32_t)(hi - (int)v[j]), 31); /* sign set <=> v > p + t */
    dark = alignbit(dark, (uint32_t)((int)v[j] - lo), 31);     /* sign set <=> v < p - t (no wrap) */
    mind = umin(mind, absdiff_u16(v[j], p));
  }
  bright &= 0xffffu, dark &= 0xffffu;
  if (lo < 0) dark = bright ^ 0xffffu; /* unsigned wrap of p - t in the reference */
  return (ring_has_run9(bright) || ring_has_run9(dark)) ? mind : 0u;
}

/* pass 1, generic: grid (ceil((w-6)/64), ceil((h-6)/4), n), block (64,4) */
__global__ __launch_bounds__(256) void k_fast_score_px(const uint8_t *img, uint8_t *score,
                                                       unsigned w, unsigned h,
                                                       size_t frame_bytes, unsigned threshold) {
  const unsigned x = 3 + blockIdx.x * 64u + threadIdx.x, y = 3 + blockIdx.y * 4u + threadIdx.y;
  if (x + 3 >= w || y + 3 >= h) return;
  const uint8_t *c = img + (size_t)blockIdx.z * frame_bytes + (size_t)y * w + x;
  const long W = (long)w;
  const unsigned v[16] = {c[-3 * W],     c[-3 * W + 1], c[-2 * W + 2], c[-W + 3],
                          c[3],          c[W + 3],      c[2 * W + 2],  c[3 * W + 1],
                          c[3 * W],      c[3 * W - 1],  c[2 * W - 2],  c[W - 3],
                          c[-3],         c[-W - 3],     c[-2 * W - 2], c[-3 * W - 1]};
  score[(size_t)blockIdx.z * frame_bytes + (size_t)y * w + x] =
      (uint8_t)fast_score(c[0], v, threshold);
}

/* idx / d for idx < 2^26 with the host's magic = ceil(2^40 / d) (needs d > 256 to fit 32 bits; the
 * error term idx * (magic*d - 2^40) < 2^26 * 2^13 stays below 2^40); magic 0: plain division */
GS_DEV unsigned div_by(unsigned idx, unsigned d, unsigned magic) {
  return magic ? (unsigned)(((unsigned long long)idx * magic) >> 40) : idx / d;
}

/* pass 2: NMS flags over the interior in raster order, item = (y-3)*(w-6) + (x-3).
 * grid (nchunks, n frames), block 256, 8 items per thread (one chunk per block). */
__global__ __launch_bounds__(256) void k_fast_nms(const uint8_t *score, unsigned w, unsigned h,
                                                  size_t frame_bytes, unsigned long long *mask,
                                                  unsigned *chunk_count, unsigned nchunks,
                                                  unsigned div_magic) {
  const unsigned iw = w - 6, nitems = iw * (h - 6);
  const uint8_t *sf = score + (size_t)blockIdx.y * frame_bytes;
  const unsigned tid = threadIdx.x, wv = tid >> 6;
  const size_t chunk = (size_t)blockIdx.y * nchunks + blockIdx.x;
  for (unsigned k = 0; k < kChunkItems / 256u; k++) {
    const unsigned idx = blockIdx.x * kChunkItems + k * 256u + tid;
    bool kp = false;
    if (idx < nitems) {
      const unsigned yy = div_by(idx, iw, div_magic), x = 3 + (idx - yy * iw), y = 3 + yy;
      const uint8_t *c = sf + (size_t)y * w + x;
      const unsigned s = c[0];
      if (s) {
        const long W = (long)w;
        unsigned m = c[-W - 1];
        m = c[-W] > m ? c[-W] : m, m = c[-W + 1] > m ? c[-W + 1] : m;
        m = c[-1] > m ? c[-1] : m, m = c[1] > m ? c[1] : m;
        m = c[W - 1] > m ? c[W - 1] : m, m = c[W] > m ? c[W] : m, m = c[W + 1] > m ? c[W + 1] : m;
        kp = !(m > s); /* strict: ties survive (ref :524) */
      }
    }
    publish_flags(kp, mask, chunk_count, chunk * kChunkWords + k * 4u + wv);
  }
}

/* compaction functor: item -> gs_keypoint {{x,y}, score, 0, {0}} (ref :530), 48 B = 12 dwords */
struct FastEmit {
  const uint8_t *score;
  unsigned w;
  size_t frame_bytes;
  unsigned *kps; /* n frames x nkps x 12 u32 */
  unsigned nkps;
  GS_DEV void operator()(unsigned frame, size_t item, unsigned r) const {
    const unsigned iw = w - 6;
    const unsigned yy = (unsigned)(item / iw), x = 3 + (unsigned)(item - (size_t)yy * iw), y = 3 + yy;
    unsigned *o = kps + ((size_t)frame * nkps + r) * 12u;
    o[0] = x, o[1] = y, o[2] = score[(size_t)frame * frame_bytes + (size_t)y * w + x];
#pragma unroll
    for (int i = 3; i < 12; i++) o[i] = 0;
  }
};

}  // namespace gs
#endif

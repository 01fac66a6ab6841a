/*
 * k_geom.h -- SURVEY.md 8(f) rank 4: gs_crop / gs_copy (grayskull.h:154-162), gs_resize_nn (:164-169),
 * gs_resize (:171-187), gs_match_template (:705-724), gs_find_best_match (:726-739).
 * One thread per output pixel; byte accesses coalesce along x.  Not tuned: none of BASELINE.json's
 * configurations is bounded by them.
 */
#ifndef GS_K_GEOM_H
#define GS_K_GEOM_H
#include "prims.h"

namespace gs {

/* gs_get (ref :143-145): 0 outside the image */
GS_DEV unsigned geom_px(const uint8_t *img, unsigned w, unsigned h, unsigned x, unsigned y) {
  return (x < w && y < h) ? img[(size_t)y * w + x] : 0u;
}

/* grid (ceil(rw/64), ceil(rh/4)), block (64,4): dst(x,y) = src(rx+x, ry+y), dropped outside dst */
__global__ __launch_bounds__(256) void k_crop(uint8_t *dst, unsigned dw, unsigned dh, const uint8_t *src,
                                              unsigned sw, unsigned sh, unsigned rx, unsigned ry,
                                              unsigned rw, unsigned rh) {
  const unsigned x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
  if (x >= rw || y >= rh || x >= dw || y >= dh) return;
  dst[(size_t)y * dw + x] = (uint8_t)geom_px(src, sw, sh, rx + x, ry + y);
}

/* NN: the reference's unsigned index arithmetic (x * sw wraps like its u32 does) */
__global__ __launch_bounds__(256) void k_resize_nn(uint8_t *dst, unsigned dw, unsigned dh, const uint8_t *src,
                                                   unsigned sw, unsigned sh) {
  const unsigned x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
  if (x >= dw || y >= dh) return;
  const unsigned sx = x * sw / dw, sy = y * sh / dh;
  dst[(size_t)y * dw + x] = (uint8_t)geom_px(src, sw, sh, sx, sy);
}

/* bilinear: float32, operation for operation like ref :173-185 (no FMA contraction: the library is
 * built -ffp-contract=off); unsigned -> float conversions where the reference's C has them */
__global__ __launch_bounds__(256) void k_resize(uint8_t *dst, unsigned dw, unsigned dh, const uint8_t *src,
                                                unsigned sw, unsigned sh) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  const unsigned x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
  if (x >= dw || y >= dh) return;
  float sx = ((float)x + 0.5f) * (float)sw / (float)dw - 0.5f;
  float sy = ((float)y + 0.5f) * (float)sh / (float)dh - 0.5f;
  const float mx = (float)sw - 1.0f, my = (float)sh - 1.0f;
  sx = sx < mx ? sx : mx, sx = 0.0f > sx ? 0.0f : sx;
  sy = sy < my ? sy : my, sy = 0.0f > sy ? 0.0f : sy;
  const unsigned xi = (unsigned)sx, yi = (unsigned)sy;
  const unsigned x1 = xi + 1 < sw - 1 ? xi + 1 : sw - 1, y1 = yi + 1 < sh - 1 ? yi + 1 : sh - 1;
  const float dx = sx - (float)xi, dy = sy - (float)yi;
  const int c00 = (int)geom_px(src, sw, sh, xi, yi), c01 = (int)geom_px(src, sw, sh, x1, yi);
  const int c10 = (int)geom_px(src, sw, sh, xi, y1), c11 = (int)geom_px(src, sw, sh, x1, y1);
  const float p = ((float)c00 * (1 - dx) * (1 - dy)) + ((float)c01 * dx * (1 - dy)) +
                  ((float)c10 * (1 - dx) * dy) + ((float)c11 * dx * dy);
  dst[(size_t)y * dw + x] = (uint8_t)(int)p; /* float -> uint8_t truncation (value < 256) */
}

/* sum of squares of n bytes into *out (one block of 256; the template's constant term) */
__global__ __launch_bounds__(256) void k_sum_squares(const uint8_t *v, unsigned long long n,
                                                     unsigned long long *out) {
  __shared__ unsigned long long part[4];
  unsigned long long acc = 0;
  for (unsigned long long i = threadIdx.x; i < n; i += 256u) acc += (unsigned long long)v[i] * v[i];
  unsigned lo = (unsigned)acc, hi = (unsigned)(acc >> 32); /* per-thread sums fit far below 2^63 */
  unsigned long long w = acc;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    lo = shfl((uint32_t)w, (int)(lane_id() ^ (unsigned)d)), hi = shfl((uint32_t)(w >> 32), (int)(lane_id() ^ (unsigned)d));
    w += ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = part[0] + part[1] + part[2] + part[3];
}

/* gs_match_template (ref :705-724).  SSD = sum I^2 + sum T^2 - 2 sum I*T over the window: per group of
 * four taps one unaligned dword of the image row, one LDS dword of the template and two
 * v_dot4_u32_u8 (I.T and I.I); sum T^2 comes from k_sum_squares.  One thread per result pixel,
 * grid (ceil(rw/64), ceil(rh/4)), block (64,4).  The template is staged in LDS as whole rows padded
 * to a multiple of 4 bytes, up to kTmplTile bytes at a time (dynamic LDS); the tw % 4 tail taps
 * of a row go byte by byte.  Row sums (<= tw * 65025 < 2^32 for tw <= 16384) are added to 64-bit
 * totals after every template row.  Requires tw <= kTmplTile - 3 (wider: k_match_template_px). */
constexpr unsigned kTmplTile = 16384;
__global__ __launch_bounds__(256) void k_match_template(const uint8_t *img, unsigned iw, unsigned ih,
                                                        const uint8_t *tmpl, unsigned tw, unsigned th,
                                                        const unsigned long long *tmpl_sq,
                                                        uint8_t *result, unsigned rw, unsigned rh) {
  GS_DYN_LDS(smem);
  uint8_t *lt = (uint8_t *)smem;
  const unsigned tid = threadIdx.y * 64u + threadIdx.x;
  const unsigned rx = blockIdx.x * 64u + threadIdx.x, ry = blockIdx.y * 4u + threadIdx.y;
  const bool live = rx < rw && ry < rh;
  const unsigned twp = (tw + 3u) & ~3u, g4 = tw >> 2;
  const unsigned rows_per_tile = kTmplTile / twp;
  unsigned long long s_it = 0, s_ii = 0;
  for (unsigned t0 = 0; t0 < th; t0 += rows_per_tile) { /* block-uniform */
    const unsigned nr = th - t0 < rows_per_tile ? th - t0 : rows_per_tile;
    __syncthreads();
    for (unsigned i = tid; i < nr * twp; i += 256u) {
      const unsigned r = i / twp, c = i - r * twp;
      lt[i] = c < tw ? tmpl[(size_t)(t0 + r) * tw + c] : (uint8_t)0;
    }
    __syncthreads();
    if (live) {
      for (unsigned r = 0; r < nr; r++) {
        const uint8_t *ip = img + (size_t)(ry + t0 + r) * iw + rx; /* the window lies inside the image */
        const uint32_t *tp = (const uint32_t *)(lt + r * twp);
        uint32_t it = 0, ii = 0;
        for (unsigned q = 0; q < g4; q++) {
          const uint32_t I4 = load_u32_unaligned(ip + 4u * q);
          it = udot4(I4, tp[q], it), ii = udot4(I4, I4, ii);
        }
        for (unsigned c = g4 * 4u; c < tw; c++) {
          const uint32_t a = ip[c], b = lt[r * twp + c];
          it += a * b, ii += a * a;
        }
        s_it += it, s_ii += ii;
      }
    }
  }
  if (!live) return;
  const unsigned long long ntaps = (unsigned long long)tw * th;
  const unsigned long long sum = s_ii + tmpl_sq[0] - 2ull * s_it; /* = sum (I - T)^2 >= 0 */
  const unsigned long long max_diff = ntaps * 255ULL * 255ULL;
  const unsigned long long score = sum * 255ULL / max_diff;
  result[(size_t)ry * rw + rx] = (uint8_t)(255u - (unsigned)(score < 255ULL ? score : 255ULL));
}

/* Register-blocked form of the same: a thread owns FOUR adjacent result pixels (rx0 = 4*k), so per
 * group of four taps it needs image bytes rx0+4q .. rx0+4q+6: two consecutive image dwords (the second
 * one is the next step's first), both at 4-byte-aligned offsets from the row start (when iw % 4 == 0
 * and the frame is 4-byte aligned), coalesced across lanes.  The four shifted windows come from
 * v_alignbyte_b32; per step: 1 load, 3 alignbyte, 8 dot4 for 16 tap-results (the one-result kernel
 * needs 4 overlapping unaligned loads for the same work).  grid (ceil(rw/4/64), ceil(rh/4)). */

__global__ __launch_bounds__(256) void k_match_template4(const uint8_t *img, unsigned iw, unsigned ih,
                                                         const uint8_t *tmpl, unsigned tw, unsigned th,
                                                         const unsigned long long *tmpl_sq,
                                                         uint8_t *result, unsigned rw, unsigned rh) {
  GS_DYN_LDS(smem);
  uint8_t *lt = (uint8_t *)smem;
  const unsigned tid = threadIdx.y * 64u + threadIdx.x;
  const unsigned rx0 = (blockIdx.x * 64u + threadIdx.x) * 4u, ry = blockIdx.y * 4u + threadIdx.y;
  const bool live = rx0 < rw && ry < rh;
  const unsigned twp = (tw + 3u) & ~3u, g4 = tw >> 2, rem = tw & 3u;
  const unsigned rows_per_tile = kTmplTile / twp;
  const uint32_t tailmask = rem ? (0xffffffffu >> (8u * (4u - rem))) : 0u; /* the tw % 4 last taps of a row */
  /* image dwords are read up to byte rx0 + twp + 3 of a row: clamp through a buffer resource
   * (zero beyond the frame) because the last row's tail may stick out of the allocation */
  const BufRsrc I = make_buf(img, (size_t)iw * ih);
  unsigned long long s_it[4] = {0, 0, 0, 0}, s_ii[4] = {0, 0, 0, 0};
  for (unsigned t0 = 0; t0 < th; t0 += rows_per_tile) { /* block-uniform */
    const unsigned nr = th - t0 < rows_per_tile ? th - t0 : rows_per_tile;
    __syncthreads();
    for (unsigned i = tid; i < nr * twp; i += 256u) {
      const unsigned r = i / twp, c = i - r * twp;
      lt[i] = c < tw ? tmpl[(size_t)(t0 + r) * tw + c] : (uint8_t)0;
    }
    __syncthreads();
    if (live) {
      for (unsigned r = 0; r < nr; r++) {
        const uint32_t base = (ry + t0 + r) * iw + rx0; /* multiple of 4 */
        const uint32_t *tp = (const uint32_t *)(lt + r * twp);
        uint32_t it[4] = {0, 0, 0, 0}, ii[4] = {0, 0, 0, 0};
        uint32_t lo = buf_load4(I, base);
        const unsigned steps = g4 + (rem ? 1u : 0u);
        for (unsigned q = 0; q < steps; q++) {
          const uint32_t hi = buf_load4(I, base + 4u * q + 4u);
          const uint32_t m = q < g4 ? 0xffffffffu : tailmask; /* wave-uniform */
          const uint32_t T4 = tp[q];                           /* padded taps are 0 in LDS */
          const uint32_t w0 = lo & m, w1 = alignbyte(hi, lo, 1) & m, w2 = alignbyte(hi, lo, 2) & m,
                         w3 = alignbyte(hi, lo, 3) & m;
          it[0] = udot4(w0, T4, it[0]), ii[0] = udot4(w0, w0, ii[0]);
          it[1] = udot4(w1, T4, it[1]), ii[1] = udot4(w1, w1, ii[1]);
          it[2] = udot4(w2, T4, it[2]), ii[2] = udot4(w2, w2, ii[2]);
          it[3] = udot4(w3, T4, it[3]), ii[3] = udot4(w3, w3, ii[3]);
          lo = hi;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) s_it[j] += it[j], s_ii[j] += ii[j];
      }
    }
  }
  if (!live) return;
  const unsigned long long ntaps = (unsigned long long)tw * th, max_diff = ntaps * 255ULL * 255ULL;
  uint32_t out = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned long long sum = s_ii[j] + tmpl_sq[0] - 2ull * s_it[j];
    const unsigned long long score = sum * 255ULL / max_diff;
    out |= (255u - (unsigned)(score < 255ULL ? score : 255ULL)) << (8 * j);
  }
  uint8_t *o = result + (size_t)ry * rw + rx0;
  if (rx0 + 3 < rw && (((uintptr_t)o) & 3u) == 0) *(uint32_t *)o = out;
  else
    for (unsigned j = 0; j < 4 && rx0 + j < rw; j++) o[j] = (uint8_t)(out >> (8 * j));
}

/* any template width: one subtract-multiply-add per tap, template read from global memory */
__global__ __launch_bounds__(256) void k_match_template_px(const uint8_t *img, unsigned iw, unsigned ih,
                                                           const uint8_t *tmpl, unsigned tw, unsigned th,
                                                           uint8_t *result, unsigned rw, unsigned rh) {
  const unsigned rx = blockIdx.x * 64u + threadIdx.x, ry = blockIdx.y * 4u + threadIdx.y;
  if (rx >= rw || ry >= rh) return;
  unsigned long long sum = 0;
  for (unsigned ty = 0; ty < th; ty++)
    for (unsigned tx = 0; tx < tw; tx++) {
      const int d = (int)geom_px(img, iw, ih, rx + tx, ry + ty) - (int)tmpl[(size_t)ty * tw + tx];
      sum += (unsigned long long)(d * d);
    }
  const unsigned long long max_diff = (unsigned long long)tw * th * 255ULL * 255ULL;
  const unsigned long long score = sum * 255ULL / max_diff;
  result[(size_t)ry * rw + rx] = (uint8_t)(255u - (unsigned)(score < 255ULL ? score : 255ULL));
}

/* gs_find_best_match: key = value << 32 | ~index: the maximum key is the largest value at the
 * LOWEST raster index (the reference's strict '>' keeps the first maximum).  grid ceil(n/2048)
 * blocks of 256, 8 items per thread; out[blockIdx.x] = block maximum; the host (or a second
 * launch over `out`) finishes.  A zero maximum means (0,0) like the reference's initial state. */
__global__ __launch_bounds__(256) void k_argmax_first(const uint8_t *v, unsigned long long n,
                                                      unsigned long long *out) {
  __shared__ unsigned long long part[4];
  unsigned long long best = 0;
  for (unsigned k = 0; k < 8; k++) {
    const unsigned long long i = ((unsigned long long)blockIdx.x * 8u + k) * 256u + threadIdx.x;
    if (i < n) {
      const unsigned long long key = ((unsigned long long)v[i] << 32) | (0xffffffffu - (unsigned)i);
      best = key > best ? key : best;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t lo = shfl((uint32_t)best, (int)(lane_id() ^ (unsigned)d));
    const uint32_t hi = shfl((uint32_t)(best >> 32), (int)(lane_id() ^ (unsigned)d));
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    best = o > best ? o : best;
  }
  if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < 4; q++) best = part[q] > best ? part[q] : best;
    out[blockIdx.x] = best;
  }
}

}  // namespace gs
#endif

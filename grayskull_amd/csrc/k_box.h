/*
 * k_box.h -- clipped box sums of any radius <= 127 without an integral image, for gs_blur with
 * radius > 3 (grayskull.h:268-283) and gs_adaptive_threshold (:230-247).
 *
 * The integral-image route (k_integral_* + k_box_px) moves ~16 B/px.  This one moves 3-4: a
 * 256-thread block owns a band of rows of one frame and spans the whole row (thread = 16 px,
 * w <= 4096).  It keeps the vertical window sum V[16] (u32) of HORIZONTAL window sums in
 * registers and slides it down one row per step:  V += H(y+r+1) - H(y-r).  The horizontal sums of
 * the entering and the leaving source row are formed on the fly: both rows are staged as bytes in
 * LDS (zero halo of 128 px either side = the clipped sum), each thread adds up its first window
 * and slides it 15 times (two LDS byte reads per step).  Every step reads two source rows
 * (2 B/px; + the centre row for the adaptive compare) and writes one output row.
 *
 * Division by the number of in-image taps (cx * cy): estimate with two float multiplies by
 * precomputed reciprocals, then make it exact with the integer remainder (the estimate is within
 * +-1).  Sums stay below 2^24 (255 * 255^2).
 */
#ifndef GS_K_BOX_H
#define GS_K_BOX_H
#include "prims.h"

namespace gs {

/* LDS row layout: pixel x lives at byte kBoxPad + x + 4*floor(x/16): every thread's 16-px segment
 * starts 20 bytes after its neighbour's, so the byte reads of a wave (one per lane, same k) fall
 * into 32 different banks (5 dwords of stride; 16-byte stride would be an 8-way conflict). */
constexpr unsigned kBoxPad = 160, kBoxRowBytes = kBoxPad + 4096 + 4096 / 4 + kBoxPad + 16;
GS_DEV int box_off(int x) { return x + 4 * (x >> 4); } /* arithmetic shift = floor for x < 0 */

/* floor(sum / (cx * cy)) for sum < 2^24 given rx ~ 1/cx, ry ~ 1/cy */
GS_DEV unsigned box_div(unsigned sum, unsigned cx, unsigned cy, float rx, float ry) {
#ifndef GS_EMU
#pragma clang fp contract(off)
#endif
  const unsigned cnt = cx * cy;
  unsigned q = (unsigned)((float)sum * rx * ry);
  const int rem = (int)sum - (int)(q * cnt);
  if (rem < 0) q--;
  else if (rem >= (int)cnt) q++;
  return q;
}

/* MODE 0: dst = mean (gs_blur); MODE 1: dst = src > (int)(mean - (unsigned)c) ? 255 : 0.
 * grid (1, nbands, n frames), block 256; T rows per band; 1 <= r <= 127, w % 16 == 0, w <= 4096. */
template <int MODE>
__global__ __launch_bounds__(256) void k_box16(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h,
                                               unsigned T, size_t frame_bytes, unsigned r, int c) {
  __shared__ __attribute__((aligned(16))) uint8_t rows[2][2][kBoxRowBytes]; /* [phase][enter/leave] */
  const unsigned tid = threadIdx.x, x0 = tid * 16u;
  const bool act = x0 < w;
  const BufRsrc S = make_buf(src + (size_t)blockIdx.z * frame_bytes, frame_bytes);
  const BufRsrc D = make_buf(dst + (size_t)blockIdx.z * frame_bytes, frame_bytes);
  const int y0 = (int)(blockIdx.y * T);
  if (y0 >= (int)h) return; /* whole block */
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  /* zero all four row buffers once: the halos and the 4 pad bytes after every 16 px are never
   * written again (rows are only ever stored inside the image span) */
  for (unsigned i = tid; i < 4u * kBoxRowBytes / 4u; i += 256u) ((uint32_t *)&rows[0][0][0])[i] = 0;
  /* per-pixel column counts and their reciprocals */
  unsigned cx[16];
  float rcx[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int x = (int)x0 + j;
    const int xa = x - (int)r < 0 ? 0 : x - (int)r, xb = x + (int)r > (int)w - 1 ? (int)w - 1 : x + (int)r;
    cx[j] = act ? (unsigned)(xb - xa + 1) : 1u;
    rcx[j] = 1.0f / (float)cx[j];
  }
  auto row_load = [&](int yy) { /* this thread's 16 B of row yy, zeros outside the image */
    return buf_load16(S, (act && yy >= 0 && yy < (int)h) ? (uint32_t)yy * w + x0 : kOOB);
  };
  const unsigned seg = kBoxPad + 20u * tid; /* = kBoxPad + box_off(x0) */
  auto row_stage = [&](unsigned phase, unsigned which, const U4 &v) {
    if (act) {
      uint32_t *q = (uint32_t *)&rows[phase][which][seg]; /* 4-byte aligned (20 * tid) */
      q[0] = v.x, q[1] = v.y, q[2] = v.z, q[3] = v.w;
    }
  };
  /* horizontal window sums of this thread's 16 px from a staged row */
  auto hsum = [&](unsigned phase, unsigned which, unsigned (&H)[16]) {
    const uint8_t *p = &rows[phase][which][seg]; /* p[box_off(k)] = pixel x0 + k, zero outside the image */
    /* first window x0-r .. x0+r: single bytes up to the next multiple of 4, then whole dwords
     * (v_dot4 with ones; the 4 pad bytes after every 16 px are zero), then the last bytes.
     * All bounds depend on r only: wave-uniform control flow. */
    unsigned s = 0;
    int k = -(int)r;
    for (; (k & 3) != 0 && k <= (int)r; k++) s += p[box_off(k)];
    for (; k + 3 <= (int)r; k += 4) s = udot4(*(const uint32_t *)(p + box_off(k)), 0x01010101u, s);
    for (; k <= (int)r; k++) s += p[box_off(k)];
    /* two running byte offsets for the slide: `hi` = entering pixel, `lo` = leaving pixel */
    int lo = box_off(-(int)r), hi = box_off((int)r + 1);
    H[0] = s; /* hi = box_off(r + 1) */
#pragma unroll
    for (int j = 1; j < 16; j++) {
      s += p[hi];
      s -= p[lo];
      hi += ((j + (int)r + 1) & 15) == 0 ? 5 : 1; /* from pixel j+r to j+r+1 */
      lo += ((j - (int)r) & 15) == 0 ? 5 : 1;     /* from pixel j-1-r to j-r */
      H[j] = s;
    }
  };
  unsigned V[16];
#pragma unroll
  for (int j = 0; j < 16; j++) V[j] = 0;
  __syncthreads();
  /* prologue: V = sum of H(yy), yy = y0-r .. y0+r, two rows per barrier pair */
  for (int yy = y0 - (int)r; yy <= y0 + (int)r; yy += 2) { /* block-uniform */
    const bool two = yy + 1 <= y0 + (int)r;
    row_stage(0, 0, row_load(yy));
    if (two) row_stage(0, 1, row_load(yy + 1));
    __syncthreads();
    unsigned H[16];
    hsum(0, 0, H);
#pragma unroll
    for (int j = 0; j < 16; j++) V[j] += H[j];
    if (two) {
      hsum(0, 1, H);
#pragma unroll
      for (int j = 0; j < 16; j++) V[j] += H[j];
    }
    __syncthreads();
  }
  /* main loop: finish row y, then V += H(y+r+1) - H(y-r) */
  U4 nin = row_load(y0 + (int)r + 1), nout = row_load(y0 - (int)r), ncen = MODE ? row_load(y0) : U4{0, 0, 0, 0};
  for (int i = 0; i < nrows; i++) { /* block-uniform */
    const int y = y0 + i;
    const unsigned ph = (unsigned)i & 1u;
    row_stage(ph, 0, nin), row_stage(ph, 1, nout);
    const U4 cen = ncen;
    __syncthreads(); /* also orders this phase's writes after the reads of two iterations ago */
    nin = row_load(y + (int)r + 2), nout = row_load(y - (int)r + 1);
    if (MODE) ncen = row_load(y + 1);
    /* finish row y */
    const int ya = y - (int)r < 0 ? 0 : y - (int)r, yb = y + (int)r > (int)h - 1 ? (int)h - 1 : y + (int)r;
    const unsigned cy = (unsigned)(yb - ya + 1);
    const float rcy = 1.0f / (float)cy;
    uint32_t od[4] = {0, 0, 0, 0};
    const uint32_t cd[4] = {cen.x, cen.y, cen.z, cen.w};
    /* MODE 1 without the division: px > floor(V / cnt) - c  <=>  px + c >= floor(V / cnt) + 1  <=>  (px + c) * cnt > V
     * (cnt > 0).  k = px + c <= 0 can never exceed V >= 0, k >= 256 always does (V <= 255 cnt), in between the
     * product fits 32 bits (cnt <= 255^2).  Only for |c| < 2^30: beyond that the reference's unsigned `mean - c`
     * wraps and the literal form below reproduces it. */
    const bool by_product = MODE == 1 && c > -(1 << 30) && c < (1 << 30);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      unsigned o;
      if (MODE == 1 && by_product) {
        const int k = (int)((cd[j >> 2] >> (8 * (j & 3))) & 0xffu) + c;
        const unsigned kc = (unsigned)(k < 0 ? 0 : k > 256 ? 256 : k);
        o = kc * (cx[j] * cy) > V[j] ? 255u : 0u;
      } else {
        const unsigned q = box_div(V[j], cx[j], cy, rcx[j], rcy);
        if (MODE == 0) o = q & 0xffu;
        else {
          const int thr = (int)(q - (unsigned)c);
          const int px = (int)((cd[j >> 2] >> (8 * (j & 3))) & 0xffu);
          o = px > thr ? 255u : 0u;
        }
      }
      od[j >> 2] |= o << (8 * (j & 3));
    }
    buf_store16(D, act ? (uint32_t)y * w + x0 : kOOB, U4{od[0], od[1], od[2], od[3]});
    /* slide */
    unsigned Hin[16], Hout[16];
    hsum(ph, 0, Hin), hsum(ph, 1, Hout);
#pragma unroll
    for (int j = 0; j < 16; j++) V[j] += Hin[j] - Hout[j];
  }
}

}  // namespace gs
#endif

/*
 * k_integral.h -- gs_integral (grayskull.h:744-752): inclusive 2-D prefix sum, u8 -> u32,
 * same w x h layout as the reference (no padding row/column), arithmetic mod 2^32.
 *
 * Generic form, any w/h/alignment: (1) k_integral_rows: per-row inclusive scan (wave scan +
 * LDS carry across the 4 waves), (2) k_integral_cols: running column sums in place, coalesced
 * across columns -- 13 B/px of traffic for 5 B/px algorithmic (1 R + 4 W).  Fast path
 * (k_integral_colsum / _colbase / _band below): 6 B/px.
 *
 * k_integral_wave: barrier-free replacement of k_integral_band (one wave per band).
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

/* ---- banded form (w % 16 == 0, w <= 4096, aligned): 6 B/px instead of 13 -----------------------
 * The image is cut into bands of BH rows.  (1) k_integral_colsum: per band, the sum of every
 * column over the band's rows (reads 1 B/px, writes w u32 per band).  (2) k_integral_colbase:
 * exclusive prefix of those sums over the bands, per column (tiny).  (3) k_integral_band: one
 * 256-thread block per band spans the whole row (lane = 16 columns); it keeps the running column
 * sums V[16] in registers (starting from the band's base) and, per row, turns them into the
 * row's inclusive prefix: in-lane scan, wave scan, one LDS hand-off between the 4 waves
 * (double-buffered: one barrier per row), then writes 4 B/px once. */

/* grid (ceil(w/4096), nbands, n frames), block 256.  Any w >= 16 and any alignment: the strip that would cross the row
 * end is anchored at w - 16 instead (it overlaps its neighbour; both store the same sums), so every load lies inside
 * the row. */
/* SQ (k_integral_colsum, k_integral_wave): the table of (p - 128)^2 instead of p -- gs_match_template's window sums of
 * squares are four corners of it (k_tmatch.h); u32 arithmetic modulo 2^32 like the plain table */
GS_DEV unsigned integral_term(unsigned b, bool sq) {
  const int v = (int)b - 128;
  return sq ? (unsigned)(v * v) : b;
}
/* NT: the batch's source planes are more than the Infinity Cache keeps -- nothing of this pass's reads will be found there by
 * the third pass, so they stream (64 x 4K -4 ... -12 %, 16 x 4096^2 -7 %; on batches that fit, the same policy costs 20-36 %:
 * profiles/r06i_integral_nt_loads.log; the launcher decides) */
template <bool SQ = false, bool NT = false>
__global__ __launch_bounds__(256) void k_integral_colsum(const uint8_t *src, unsigned w, unsigned h,
                                                         unsigned BH, unsigned nbands,
                                                         unsigned *colsum) {
  const unsigned xg = (blockIdx.x * 256u + threadIdx.x) * 16u;
  const bool act = xg < w;
  const unsigned x0 = xg + 16u > w ? w - 16u : xg;
  const size_t fb = (size_t)w * h;
  const BufRsrc S = make_buf(src + (size_t)blockIdx.z * fb, fb);
  const unsigned y0 = blockIdx.y * BH, y1 = y0 + BH < h ? y0 + BH : h;
  unsigned V[16];
#pragma unroll
  for (int k = 0; k < 16; k++) V[k] = 0;
  U4 nxt = buf_load16_pol<NT>(S, act ? y0 * w + x0 : kOOB);
  for (unsigned y = y0; y < y1; y++) {
    const U4 cur = nxt;
    nxt = buf_load16_pol<NT>(S, (act && y + 1 < y1) ? (y + 1) * w + x0 : kOOB);
    const uint32_t d[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
    for (int k = 0; k < 16; k++) V[k] += integral_term((d[k >> 2] >> (8 * (k & 3))) & 0xffu, SQ);
  }
  if (act) {
    unsigned *o = colsum + ((size_t)blockIdx.z * nbands + blockIdx.y) * w + x0;
#pragma unroll
    for (int q = 0; q < 4; q++) store_u32x4_any(o + 4 * q, U4{V[4 * q], V[4 * q + 1], V[4 * q + 2], V[4 * q + 3]});
  }
}

/* in place: colsum[b][x] <- sum of colsum[b'][x] for b' < b.  grid (ceil(w/64), n), block (64,16):
 * a wave owns 64 columns of one sixteenth of the bands -- sums it, the 16 partial sums meet in
 * LDS, then it rewrites its bands as running sums (a few hundred bands walked by one thread per
 * column is 30 us of dependent latency; this is 5). */
__global__ __launch_bounds__(1024) void k_integral_colbase(unsigned *colsum, unsigned w, unsigned nbands) {
  __shared__ unsigned part[16][64];
  const unsigned x = blockIdx.x * 64u + threadIdx.x, g = threadIdx.y;
  const unsigned per = (nbands + 15u) / 16u;
  const unsigned b0 = g * per < nbands ? g * per : nbands, b1 = b0 + per < nbands ? b0 + per : nbands;
  unsigned *p = colsum + (size_t)blockIdx.y * nbands * w + x;
  unsigned sum = 0;
  if (x < w) {
    unsigned b = b0;
    for (; b + 8 <= b1; b += 8) {
      unsigned v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = p[(size_t)(b + k) * w];
#pragma unroll
      for (int k = 0; k < 8; k++) sum += v[k];
    }
    for (; b < b1; b++) sum += p[(size_t)b * w];
  }
  part[g][threadIdx.x] = sum;
  __syncthreads();
  if (x >= w) return;
  unsigned acc = 0;
  for (unsigned k = 0; k < g; k++) acc += part[k][threadIdx.x];
  unsigned b = b0;
  for (; b + 8 <= b1; b += 8) {
    unsigned v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = p[(size_t)(b + k) * w];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      p[(size_t)(b + k) * w] = acc;
      acc += v[k];
    }
  }
  for (; b < b1; b++) {
    const unsigned v = p[(size_t)b * w];
    p[(size_t)b * w] = acc;
    acc += v;
  }
}

/* grid (1, nbands, n frames), block 256 */
__global__ __launch_bounds__(256) void k_integral_band(const uint8_t *src, unsigned w, unsigned h,
                                                       unsigned BH, unsigned nbands,
                                                       const unsigned *colbase, unsigned *ii) {
  __shared__ unsigned wtot[2][4];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const unsigned x0 = tid * 16u;
  const size_t fb = (size_t)w * h;
  const BufRsrc S = make_buf(src + (size_t)blockIdx.z * fb, fb);
  const BufRsrc D = make_buf(ii + (size_t)blockIdx.z * fb, fb * 4);
  const unsigned y0 = blockIdx.y * BH, y1 = y0 + BH < h ? y0 + BH : h;
  unsigned V[16];
  {
    const unsigned *cb = colbase + ((size_t)blockIdx.z * nbands + blockIdx.y) * w + x0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      U4 t{0, 0, 0, 0};
      if (x0 < w) t = *(const U4 *)(cb + 4 * q);
      V[4 * q] = t.x, V[4 * q + 1] = t.y, V[4 * q + 2] = t.z, V[4 * q + 3] = t.w;
    }
  }
  U4 nxt = buf_load16(S, x0 < w ? y0 * w + x0 : kOOB);
  for (unsigned y = y0; y < y1; y++) { /* block-uniform trip count */
    const U4 cur = nxt;
    nxt = buf_load16(S, (x0 < w && y + 1 < y1) ? (y + 1) * w + x0 : kOOB);
    const uint32_t d[4] = {cur.x, cur.y, cur.z, cur.w};
    unsigned P[16], run = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      V[k] += (d[k >> 2] >> (8 * (k & 3))) & 0xffu; /* lanes beyond w add 0 to 0 */
      run += V[k];
      P[k] = run;
    }
    const unsigned inc = wave_incl_scan(run);
    const unsigned par = y & 1u;
    if (lane == 63) wtot[par][wv] = inc;
    __syncthreads();
    unsigned off = inc - run;
    for (unsigned q = 0; q < wv; q++) off += wtot[par][q];
    const uint32_t ob = x0 < w ? (y * w + x0) * 4u : kOOB; /* branch-free: dropped beyond w */
#pragma unroll
    for (int q = 0; q < 4; q++)
      buf_store16_wb(D, ob + 16u * q, U4{off + P[4 * q], off + P[4 * q + 1], off + P[4 * q + 2], off + P[4 * q + 3]});
  }
}

/* Barrier-free form of step (3): one WAVE per band spans the whole row.  The row is cut into TILES
 * tiles of 256 px; in tile t the lane owns the 4 px at t*256 + 4*lane, so every load is one
 * coalesced 256-B dword row per wave and every store one coalesced 1-KiB dwordx4 row.  Running
 * column sums V[TILES][4] live in registers; per row: in-lane prefix of 4, one DPP wave scan per
 * tile, and the tile totals chain through a scalar carry (v_readlane).  No LDS, no barrier, the
 * next row's TILES dwords are in flight during the arithmetic.
 * grid (1, ceil(nbands/4), n frames), block 256 = 4 waves = 4 bands.
 * RAGGED (w % 4 != 0): the one lane whose 4 px would cross the row end is anchored at w - 4; it overlaps its left
 * neighbour by ov = 1..3 px, so it enters the scan with the sum of its LAST 4 - ov px only -- the prefix at its
 * pixels is still (inclusive scan - own four) + in-lane prefix -- and both lanes store the same values where they
 * overlap: whole dwordx4 stores, nothing past the row end, no read-modify-write of the next row's head.
 * WIDE (w > TILES * 256): the wave walks the row's column chunks of TILES * 256 px one after the other, rows top to
 * bottom inside each; what a row hands from chunk to chunk -- its prefix at the chunk's left edge -- waits in a
 * register of lane (row - y0): v_readlane / one masked move per row, no LDS (BH <= kIntegralWideRows = 64). */
constexpr unsigned kIntegralWideRows = 64;
template <int TILES, bool RAGGED = false, bool WIDE = false, bool SQ = false, bool NT = false>
__global__ __launch_bounds__(256) void k_integral_wave(const uint8_t *src, unsigned w, unsigned h,
                                                       unsigned BH, unsigned nbands,
                                                       const unsigned *colbase, unsigned *ii) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned band = uniform(blockIdx.y * 4u + (threadIdx.x >> 6));
  unsigned rowcarry = 0; /* WIDE: lane i keeps row y0 + i's prefix at the left edge of the next chunk */
  if (band >= nbands) return; /* whole wave */
  const size_t fb = (size_t)w * h;
  const BufRsrc S = make_buf(src + (size_t)blockIdx.z * fb, fb);
  const BufRsrc D = make_buf(ii + (size_t)blockIdx.z * fb, fb * 4);
  const BufRsrc C = make_buf(colbase + ((size_t)blockIdx.z * nbands + band) * w, (size_t)w * 4);
  const unsigned y0 = band * BH, y1 = y0 + BH < h ? y0 + BH : h;
  const unsigned nchunk = WIDE ? (w + TILES * 256u - 1u) / (TILES * 256u) : 1u;
  for (unsigned ch = 0; ch < nchunk; ch++) { /* wave-uniform */
    const unsigned cx = ch * (unsigned)TILES * 256u;
    unsigned V[TILES][4];
    uint32_t xo[TILES], nxt[TILES]; /* per-tile byte offset of the lane's 4 px inside a row, or OOB */
    /* RAGGED: tile and lane-constant masks of the anchored lane (ov px shared with its left neighbour) */
    unsigned k0 = 0, k1 = 0, k2 = 0;
    int tail_tile = -1;
#pragma unroll
    for (int t = 0; t < TILES; t++) {
      unsigned x = cx + (unsigned)t * 256u + lane * 4u;
      const bool in = x < w;
      if constexpr (RAGGED) {
        if (cx + (unsigned)t * 256u < w && w < cx + (unsigned)(t + 1) * 256u) tail_tile = t; /* wave-uniform */
        if (in && x + 4u > w) {
          const unsigned ov = x + 4u - w;
          x = w - 4u, k0 = 0xffffffffu, k1 = ov >= 2u ? 0xffffffffu : 0u, k2 = ov >= 3u ? 0xffffffffu : 0u;
        }
      }
      xo[t] = in ? x : kOOB;
      const U4 b = buf_load16(C, in ? x * 4u : kOOB); /* zero beyond w */
      V[t][0] = b.x, V[t][1] = b.y, V[t][2] = b.z, V[t][3] = b.w;
      nxt[t] = buf_load4_pol<NT>(S, in ? y0 * w + x : kOOB);
    }
    for (unsigned y = y0; y < y1; y++) { /* wave-uniform trip count */
      uint32_t cur[TILES];
      const bool more = y + 1 < y1;
#pragma unroll
      for (int t = 0; t < TILES; t++) {
        cur[t] = nxt[t];
        nxt[t] = buf_load4_pol<NT>(S, (more && xo[t] != kOOB) ? (y + 1) * w + xo[t] : kOOB);
      }
      unsigned carry = 0; /* sum of the tiles to the left, wave-uniform */
      if constexpr (WIDE) {
        carry = readlane_at(rowcarry, y - y0); /* 0 in the first chunk */
      }
#pragma unroll
      for (int t = 0; t < TILES; t++) {
        const uint32_t d = cur[t];
        V[t][0] += integral_term(d & 0xffu, SQ), V[t][1] += integral_term((d >> 8) & 0xffu, SQ);
        V[t][2] += integral_term((d >> 16) & 0xffu, SQ), V[t][3] += integral_term(d >> 24, SQ);
        const unsigned p0 = V[t][0], p1 = p0 + V[t][1], p2 = p1 + V[t][2], p3 = p2 + V[t][3];
        unsigned mine = p3; /* what this lane adds to the row's running sum */
        if constexpr (RAGGED) {
          if (t == tail_tile) mine = p3 - ((V[t][0] & k0) + (V[t][1] & k1) + (V[t][2] & k2));
        }
        const unsigned inc = wave_incl_scan(mine);
        const unsigned off = carry + inc - p3;
        buf_store16_wb(D, xo[t] != kOOB ? (y * w + xo[t]) * 4u : kOOB, U4{off + p0, off + p1, off + p2, off + p3});
        carry += readlane_last(inc);
      }
      if constexpr (WIDE) {
        rowcarry = lane == y - y0 ? carry : rowcarry;
      }
    }
  }
}

/* grid (ceil((w+1)/64), ceil((h+1)/4), n), block (64,4) */
__global__ __launch_bounds__(256) void k_integral_pad(const unsigned *ii, unsigned w, unsigned h, unsigned *padded) {
  const unsigned x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
  if (x > w || y > h) return;
  const unsigned *f = ii + (size_t)blockIdx.z * w * h;
  const unsigned v = (x && y) ? f[(size_t)(y - 1) * w + (x - 1)] : 0u;
  padded[(size_t)blockIdx.z * (w + 1) * (h + 1) + (size_t)y * (w + 1) + x] = v;
}

}  // namespace gs
#endif

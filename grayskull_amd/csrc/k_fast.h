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
 * Algorithmic traffic is 3 B/px (score pass 1 R + 1 W, NMS pass 1 R), but the score pass is VALU-bound (0.76 VALU
 * wave-instructions per pixel on the block-noise frames of configs[3], profiles/fast_valu_pmc.json).  Two score kernels:
 * k_fast_score_q4 (default: LDS tile, 4 px per thread through the compass filter, candidates queued and scored 64 to a wave)
 * and k_fast_score_px (one global byte load per ring pixel; any threshold, incl. those where p + t wraps in 32 bits).  The
 * kernels that lost to them over rounds 2-4 -- the one-pixel-per-lane LDS tile form, its candidate-queue variant, the strip
 * form with image rows in registers, the strip NMS over every pixel -- were deleted in round 5 (scripts/experiments/not_kept/
 * keeps what was measured; profiles/r02i_fast_tile.log, r03i_fast_candidate_queue.log, r04z_fast.log).
 */
#ifndef GS_K_FAST_H
#define GS_K_FAST_H
#include "k_compact.h"
#include "k_strip.h"

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
   * signed differences below cannot overflow.  (Round 5 tried compare + add-with-carry pairs -- push_gt_u32, two
   * full-rate instructions per ring pixel and class instead of a subtraction and a half-rate v_alignbit_b32: the score
   * pass of 32 x 720p block noise went from 46-47 to 49-51 us, the inline-asm pairs pin VCC and keep the scheduler from
   * interleaving the 32 chains; profiles/r05e_fast.log.  Not kept.) */
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

/* Necessary condition for a circular run of >= 9 of one class: any 9 consecutive ring positions
 * contain at least two of the compass positions 0, 4, 8, 12, so at least two compass pixels must be
 * brighter, or at least two darker -- with exactly the class definitions of fast_score (unsigned
 * wrap of p - t included), so the filter never changes a score. */
GS_DEV bool fast_compass_candidate(unsigned p, unsigned v0, unsigned v4, unsigned v8, unsigned v12,
                                   unsigned threshold) {
  if (threshold > 0xffffff00u) return true; /* u32-wrap regime of fast_score_u32: no filter */
  const int t = (int)(threshold < 256u ? threshold : 256u), hi = (int)p + t, lo = (int)p - t;
  const int b = ((int)v0 > hi) + ((int)v4 > hi) + ((int)v8 > hi) + ((int)v12 > hi);
  const int d = lo < 0 ? 4 - b : ((int)v0 < lo) + ((int)v4 < lo) + ((int)v8 < lo) + ((int)v12 < lo);
  return b >= 2 || d >= 2;
}

/* pass 1, generic: grid (ceil((w-6)/64), ceil((h-6)/4), n), block (64,4).
 * A wave is 64 consecutive pixels of one row; when none of them passes the compass filter the
 * other 12 ring pixels are never loaded (wave-uniform branch: most waves of a frame with flat or
 * straight-edged regions). */
__global__ __launch_bounds__(256) void k_fast_score_px(const uint8_t *img, uint8_t *score,
                                                       unsigned w, unsigned h,
                                                       size_t frame_bytes, unsigned threshold) {
  const unsigned x = 3 + blockIdx.x * 64u + threadIdx.x, y = 3 + blockIdx.y * 4u + threadIdx.y;
  const bool in = x + 3 < w && y + 3 < h;
  const uint8_t *c = img + (size_t)blockIdx.z * frame_bytes + (size_t)(in ? y : 3u) * w + (in ? x : 3u);
  const long W = (long)w;
  const unsigned p = c[0], v0 = c[-3 * W], v4 = c[3], v8 = c[3 * W], v12 = c[-3];
  const bool cand = in && fast_compass_candidate(p, v0, v4, v8, v12, threshold);
  unsigned s = 0;
  if (ballot(cand) != 0) { /* wave-uniform */
    const unsigned v[16] = {v0,  c[-3 * W + 1], c[-2 * W + 2], c[-W + 3],
                            v4,  c[W + 3],      c[2 * W + 2],  c[3 * W + 1],
                            v8,  c[3 * W - 1],  c[2 * W - 2],  c[W - 3],
                            v12, c[-W - 3],     c[-2 * W - 2], c[-3 * W - 1]};
    s = fast_score(p, v, threshold);
  }
  if (in) score[(size_t)blockIdx.z * frame_bytes + (size_t)y * w + x] = (uint8_t)s;
}

/* idx / d for any idx with M = floor(2^32 / d) (0xffffffff for d = 1): the estimate is at most 2 short */
GS_DEV unsigned udiv_by_magic(unsigned idx, unsigned d, unsigned M) {
  unsigned q = __umulhi(idx, M), r = idx - q * d;
  if (r >= d) q++, r -= d;
  if (r >= d) q++;
  return q;
}

constexpr unsigned kFastTileDw = 18; /* 72 bytes per tile row of k_fast_score_q4: 64 + 6, rounded up to dwords */

/* pass 1, round 3 default (threshold <= 0xffffff00): LDS tile, FOUR pixels per thread for the compass filter, candidates
 * queued and scored 64 to a wave.  Where k_fast_score_tile's time goes on the configs[3] frames (72 us per 32 x 720p):
 * half of it is paid on flat frames too (38 us) -- one pixel per lane means 5 ds_read_u8, ~35 scalar-per-pixel VALU
 * operations and a byte store for EVERY pixel -- and the 16-pixel ring is walked for whole wave rows although only
 * 11 % of the pixels pass the filter (45 % of the wave rows hold at least one).  Here a thread takes 4 consecutive
 * pixels of a tile row: 7 aligned LDS dwords (centre row 3, rows y -+ 3 two each; v_alignbit shifts them into
 * place) feed the compass filter on packed u16 pairs exactly as in k_fast_score4 (second largest / second smallest
 * of the four compass pixels against p +- t; p < t always passes: the reference's unsigned wrap, ref :496-498), the
 * four scores are stored as ONE zero dword, and only the pixels that pass are queued (LDS, order irrelevant) and
 * scored from the tile's bytes with fast_score -- the same function as every other score kernel, and a pixel the
 * filter rejects has no run of 9, so the stored zero IS its score.  One pass over the 64 x 16 tile per block
 * (256 threads x 4 px) instead of four.  grid: one block per tile, 1-D (see the tile mapping below); block (64, 4). */
/* ROWS (round 4): tile height, a multiple of 16.  A block's life is one round trip to memory for its tile plus two barriers;
 * on flat frames that latency, not arithmetic, is the whole cost (35 of the 60 us per 32 x 720p at ROWS = 16: ~2.5 us per
 * block with 8 blocks per CU resident).  Taller tiles halve the blocks and the halo share (6 extra rows per 32 instead of
 * per 16) for the same round trip; each thread then filters 4 pixels of ROWS / 16 tile rows. */
typedef uint32_t gs_u32_unaligned __attribute__((aligned(1)));
/* NT: threads per block (64 x NT / 64): fewer threads per tile = more tiles in flight per CU for the same registers */
/* 256 threads, tiles up to 48 rows: held to 64 registers = 8 waves per SIMD (75 / 6 waves without the bound, no scratch with it) */
template <unsigned ROWS, unsigned NT = 256>
__global__ __launch_bounds__(NT, (NT == 256 && ROWS <= 48) ? 8 : 1) void k_fast_score_q4(const uint8_t *img, uint8_t *score, unsigned w, unsigned h,
                                                       size_t frame_bytes, unsigned threshold, unsigned tiles_x,
                                                       unsigned tiles_y, unsigned ntiles, unsigned xcd_share,
                                                       unsigned long long *nz, size_t nz_frame_words,
                                                       unsigned magic_x, unsigned magic_xy) { /* floor(2^32 / tiles_x), floor(2^32 / (tiles_x tiles_y)) */
  static_assert(ROWS % 16 == 0 && ROWS >= 16 && ROWS <= 64 && (NT == 128 || NT == 256), "a thread takes one row of every group of NT / 16; queue entries are 16-bit");
  constexpr unsigned RG = NT / 16; /* tile rows per row group */
  __shared__ uint32_t tile32[(ROWS + 6) * kFastTileDw + 2]; /* + 2: the last thread's third centre dword */
  __shared__ uint16_t queue[64 * ROWS];
  __shared__ unsigned qn;
  /* nz != nullptr (round 4): the pixels with a non-zero score as a bitmap for the sparse NMS pass (k_fast_nms.h) -- word
   * (y - 3) * tiles_x + tcol of the frame's nz_frame_words, bit qx <-> pixel x = 3 + 64 tcol + qx: a tile row is exactly
   * one word.  Every word of the interior rows is written (zeros included), so the bitmap needs no clearing. */
  __shared__ uint32_t nzw[2 * ROWS];
  /* tile of this block.  Workgroups go to the 8 XCDs round robin and each XCD has its own L2: with neighbouring tiles on
   * different XCDs every L2 fetches the shared halo rows and the 128-byte lines a 70-byte tile row straddles for itself
   * (FETCH_SIZE 4.1 x the frame bytes, WRITE_SIZE 1.46 x from the split lines of the score map).  xcd_share != 0: a 1-D
   * grid of 8 * xcd_share blocks, XCD k walks tiles [k * xcd_share, (k + 1) * xcd_share) in order, so the tiles in flight
   * on one XCD are a few consecutive tile rows of one frame. */
  unsigned tile = blockIdx.x;
  if (xcd_share) {
    tile = (blockIdx.x & 7u) * xcd_share + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
  }
  /* tile -> (frame, tile row, tile column): wave-uniform, but the scalar unit has no division and hipcc emitted three float
   * reciprocal sequences on the VALU (~70 of the ~440 VALU instructions a wave executes on a flat tile); with the host's
   * floor(2^32 / d) they are scalar multiply-highs with a fix-up (round 5). */
  const unsigned txy = tiles_x * tiles_y, tframe = udiv_by_magic(tile, txy, magic_xy), trem = tile - tframe * txy;
  const unsigned trow = udiv_by_magic(trem, tiles_x, magic_x), tcol = trem - trow * tiles_x;
  const uint8_t *frame = img + (size_t)tframe * frame_bytes;
  uint8_t *out = score + (size_t)tframe * frame_bytes;
  const unsigned tid = threadIdx.y * 64u + threadIdx.x;
  const unsigned x_t = tcol * 64u, y_t = trow * ROWS;
  if (tid == 0) qn = 0, tile32[(ROWS + 6) * kFastTileDw] = 0, tile32[(ROWS + 6) * kFastTileDw + 1] = 0;
  if (tid < 2 * ROWS) nzw[tid] = 0;
  /* block-uniform: every dword of the tile region lies inside the frame (all tiles but those of the last tile row) -- then
   * the copy is a plain loop: (row, column) advance by NT = 14 x 18 + 4 dwords without a division, no bounds test per load */
  const bool tile_inside = ((size_t)(y_t + ROWS + 5u) * w + x_t + kFastTileDw * 4u <= frame_bytes) && frame_bytes < 0x7fffffffull;
  if (tile_inside) {
    /* Round 5: the thread's (up to) four dwords are requested TOGETHER, as unconditional buffer loads (past the tile's last
     * dword: the out-of-range offset), and stored to the LDS afterwards.  With `if (i < ...) tile32[i] = load(...)` hipcc put
     * every load behind a branch and waited for it before issuing the next: four memory latencies in a row at the top of
     * every block.  Worth ~1 us of 70 (profiles/r05o_fast_tile_loads_together.log: flat 27.6 -> 25.9-27.8, block noise 48.4-50.8
     * -> 47.6-49.0 for the score pass): eight blocks per CU already covered the waits -- flat frames are VALU time too,
     * ~25 lane-operations per pixel for the tile copy, the compass filter and the zero stores. */
    constexpr unsigned dr = NT / kFastTileDw, dc = NT - dr * kFastTileDw, LIM = (ROWS + 6) * kFastTileDw, NIT = (LIM + NT - 1) / NT;
    const BufRsrc FB = make_buf(frame, frame_bytes); /* frame_bytes < 2 GiB on this path */
    unsigned r = tid / kFastTileDw, c = tid - r * kFastTileDw;
    const unsigned base = y_t * w + x_t;
    uint32_t v[NIT];
#pragma unroll
    for (unsigned it = 0; it < NIT; it++) {
      v[it] = buf_load4(FB, tid + it * NT < LIM ? base + r * w + c * 4u : kOOB);
      r += dr, c += dc;
      if (c >= kFastTileDw) c -= kFastTileDw, r++;
    }
#pragma unroll
    for (unsigned it = 0; it < NIT; it++)
      if (tid + it * NT < LIM) tile32[tid + it * NT] = v[it];
  } else {
    for (unsigned i = tid; i < (ROWS + 6) * kFastTileDw; i += NT) {
      const unsigned r = i / kFastTileDw, c = i - r * kFastTileDw;
      const size_t off = (size_t)(y_t + r) * w + x_t + c * 4u;
      uint32_t v = 0;
      if (off + 4 <= frame_bytes) {
        v = load_u32_unaligned(frame + off);
      } else {
        for (unsigned b = 0; b < 4; b++)
          if (off + b < frame_bytes) v |= (uint32_t)frame[off + b] << (8 * b);
      }
      tile32[i] = v;
    }
  }
  __syncthreads();
  const bool tile_interior = x_t + 69u < w && y_t + ROWS + 5u < h; /* block-uniform: every pixel of the tile is an interior pixel */
  /* thread -> tile row ry, pixels 4 xg .. 4 xg + 3 of it (tile byte columns 4 xg + 3 .. 4 xg + 6) */
  const unsigned xg = tid & 15u, x = 3 + x_t + 4u * xg;
  const uint32_t t16 = threshold < 256u ? threshold : 256u, tt = t16 | (t16 << 16);
  unsigned cands[ROWS / RG]; /* per row group, bit k: pixel k passes the compass filter */
#pragma unroll
  for (unsigned rg = 0; rg < ROWS / RG; rg++) {
  const unsigned ry = rg * RG + (tid >> 4), y = 3 + y_t + ry;
  const uint32_t *rc = tile32 + (ry + 3) * kFastTileDw + xg, *ru = tile32 + ry * kFastTileDw + xg,
                 *rd = tile32 + (ry + 6) * kFastTileDw + xg;
  const uint32_t d0 = rc[0], d1 = rc[1], d2 = rc[2];
  const uint32_t C = alignbit(d1, d0, 24), L = d0, Rr = alignbit(d2, d1, 16); /* p; (x - 3, y); (x + 3, y) */
  const uint32_t U = alignbit(ru[1], ru[0], 24), D = alignbit(rd[1], rd[0], 24); /* (x, y - 3); (x, y + 3) */
  unsigned cand = 0;
#pragma unroll
  for (int hp = 0; hp < 2; hp++) {
    const uint32_t P = hp ? unpack_hi(C) : unpack_lo(C), a = hp ? unpack_hi(U) : unpack_lo(U), c = hp ? unpack_hi(D) : unpack_lo(D);
    const uint32_t b = hp ? unpack_hi(Rr) : unpack_lo(Rr), d = hp ? unpack_hi(L) : unpack_lo(L);
    const uint32_t mx = pk_max_u16(a, b), mn = pk_min_u16(a, b), z = pk_max_u16(c, d), u = pk_min_u16(c, d);
    const uint32_t mid_hi = pk_min_u16(mx, z), mid_lo = pk_max_u16(mn, u);
    const uint32_t S2 = pk_max_u16(mid_hi, mid_lo), s2 = pk_min_u16(mid_hi, mid_lo); /* second largest / smallest */
    const uint32_t pass = pk_subsat_u16(S2, pk_add_u16(P, tt)) | pk_subsat_u16(pk_subsat_u16(P, tt), s2) | pk_subsat_u16(tt, P);
    cand |= ((pass & 0xffffu) ? 1u : 0u) << (2 * hp) | ((pass >> 16) ? 1u : 0u) << (2 * hp + 1);
  }
  /* pixels of the interior only (3 <= x < w - 3, 3 <= y < h - 3); the tile may stick out of it */
  unsigned inmask = 15u;
  if (!tile_interior) {
    inmask = 0;
#pragma unroll
    for (unsigned k = 0; k < 4; k++) inmask |= (x + k + 3u < w && y + 3u < h ? 1u : 0u) << k;
  }
  cand &= inmask;
  if (inmask == 15u) {
    *(gs_u32_unaligned *)(out + (size_t)y * w + x) = 0u; /* candidates are overwritten behind the barrier */
  } else {
#pragma unroll
    for (unsigned k = 0; k < 4; k++)
      if ((inmask >> k) & 1u) out[(size_t)y * w + x + k] = 0;
  }
  const uint64_t many = ballot(cand != 0u);
  if (many != 0ull) { /* wave-uniform: most wave rows of a frame hold no candidate at all */
    /* queue the candidates (order irrelevant): ONE LDS atomic per wave reserves the slots of all four pixel positions, the
     * four ballots only rank the lanes (round 4; one atomic and one round trip per position before) */
    uint64_t mk[4];
    unsigned nk = 0;
#pragma unroll
    for (unsigned k = 0; k < 4; k++) mk[k] = ballot((cand >> k) & 1u), nk += (unsigned)__popcll(mk[k]);
    const unsigned lane = lane_id();
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&qn, nk);
    base = readlane0(base);
#pragma unroll
    for (unsigned k = 0; k < 4; k++) {
      if ((cand >> k) & 1u) queue[base + mbcnt(mk[k])] = (uint16_t)(ry * 64u + 4u * xg + k);
      base += (unsigned)__popcll(mk[k]);
    }
  }
  cands[rg] = cand;
  }
  __syncthreads(); /* orders the zero stores above before the candidates' stores below (workgroup-scope release / acquire) */
  const unsigned ncand = qn;
  const uint8_t *tb = (const uint8_t *)tile32;
  constexpr int S = (int)kFastTileDw * 4;
  if (ncand * 2u >= 64u * ROWS) { /* most of the tile passes (noise, p < t regions): the queue would only add a round trip
                                              and scatter the LDS reads -- every thread scores its own pixels (32 x 720p random
                                              bytes: 138 us through the queue, 110 in place) */
#pragma unroll
    for (unsigned rg = 0; rg < ROWS / RG; rg++) {
      const unsigned cand = cands[rg], ry = rg * RG + (tid >> 4), y = 3 + y_t + ry;
#pragma unroll
      for (unsigned k = 0; k < 4; k++) {
        if (ballot((cand >> k) & 1u) == 0) continue; /* wave-uniform */
        const uint8_t *c = tb + (ry + 3) * S + 4u * xg + k + 3;
        const unsigned v[16] = {c[-3 * S],     c[-3 * S + 1], c[-2 * S + 2], c[-S + 3], c[3],  c[S + 3],  c[2 * S + 2],  c[3 * S + 1],
                                c[3 * S],      c[3 * S - 1],  c[2 * S - 2],  c[S - 3],  c[-3], c[-S - 3], c[-2 * S - 2], c[-3 * S - 1]};
        const unsigned sc = fast_score(c[0], v, threshold);
        if (((cand >> k) & 1u) && sc) {
          out[(size_t)y * w + x + k] = (uint8_t)sc;
          atomicOr(&nzw[2u * ry + (xg >> 3)], 1u << ((4u * xg + k) & 31u));
        }
      }
    }
  } else {
    for (unsigned i0 = 0; i0 < ncand; i0 += NT) { /* block-uniform trip count */
      const unsigned i = i0 + tid;
      if (i0 + (tid & ~63u) >= ncand) continue; /* whole wave past the queue's end */
      const unsigned e = queue[i < ncand ? i : ncand - 1u], qy = e >> 6, qx = e & 63u;
      const uint8_t *c = tb + (qy + 3) * S + qx + 3;
      const unsigned v[16] = {c[-3 * S],     c[-3 * S + 1], c[-2 * S + 2], c[-S + 3], c[3],  c[S + 3],  c[2 * S + 2],  c[3 * S + 1],
                              c[3 * S],      c[3 * S - 1],  c[2 * S - 2],  c[S - 3],  c[-3], c[-S - 3], c[-2 * S - 2], c[-3 * S - 1]};
      const unsigned sc = fast_score(c[0], v, threshold);
      if (i < ncand && sc) {
        out[(size_t)(3 + y_t + qy) * w + 3 + x_t + qx] = (uint8_t)sc;
        atomicOr(&nzw[2u * qy + (qx >> 5)], 1u << (qx & 31u));
      }
    }
  }
  if (nz) {
    if (ncand) __syncthreads(); /* block-uniform */
    if (tid < ROWS && y_t + tid + 6u < h)
      nz[(size_t)tframe * nz_frame_words + (size_t)(y_t + tid) * tiles_x + tcol] =
          ncand ? (unsigned long long)nzw[2u * tid] | ((unsigned long long)nzw[2u * tid + 1u] << 32) : 0ull;
  }
}

/* idx / d for idx < 2^26 with the host's magic = ceil(2^40 / d) (needs d > 256 to fit 32 bits; the
 * error term idx * (magic*d - 2^40) < 2^26 * 2^13 stays below 2^40); magic 0: plain division */
GS_DEV unsigned div_by(unsigned idx, unsigned d, unsigned magic) {
  return magic ? (unsigned)(((unsigned long long)idx * magic) >> 40) : idx / d;
}

/* strict 3x3 maximum test of one interior pixel, byte by byte (ref :519-529) */
GS_DEV bool fast_is_peak(const uint8_t *c, long W) {
  const unsigned s = c[0];
  if (!s) return false;
  unsigned m = c[-W - 1];
  m = c[-W] > m ? c[-W] : m, m = c[-W + 1] > m ? c[-W + 1] : m;
  m = c[-1] > m ? c[-1] : m, m = c[1] > m ? c[1] : m;
  m = c[W - 1] > m ? c[W - 1] : m, m = c[W] > m ? c[W] : m, m = c[W + 1] > m ? c[W + 1] : m;
  return !(m > s); /* strict: ties survive (ref :524) */
}

/* pass 2: NMS flags over the interior in raster order, item = (y-3)*(w-6) + (x-3).
 * grid (nchunks, n frames), block 256, one 2048-item chunk per block.  A lane owns 4 consecutive
 * items: one (unaligned) dword of scores; almost always it is 0 and the lane is done.  Otherwise
 * the 3x3 neighbourhoods of the 4 pixels are 6 more dwords.  The 4 ballots (one per slot) are
 * stored as the group's 4 mask words, slot-major; k_emit<F, QUAD> restores scan order. */
__global__ __launch_bounds__(256) void k_fast_nms(const uint8_t *score, unsigned w, unsigned h,
                                                  size_t frame_bytes, unsigned long long *mask,
                                                  unsigned *chunk_count, unsigned nchunks,
                                                  unsigned div_magic) {
  const unsigned iw = w - 6, nitems = iw * (h - 6);
  const uint8_t *sf = score + (size_t)blockIdx.y * frame_bytes;
  const unsigned tid = threadIdx.x, wv = tid >> 6;
  const size_t chunk = (size_t)blockIdx.y * nchunks + blockIdx.x;
  const long W = (long)w;
  __shared__ unsigned wave_hits[4];
  unsigned hits = 0; /* this wave's share of the chunk's count */
  /* The pass is a chain of dependent loads (score dword -> its neighbourhood) on few waves per CU, so a lane's two
   * groups of 4 items go through it TOGETHER: both score dwords are loaded first, then both neighbourhoods
   * (two memory round trips per block instead of four; 29 -> 27.5 us per 32 x 720p: the pass is not only that). */
  constexpr unsigned K = kChunkItems / 1024u;
  unsigned idx[K], xo[K];
  const uint8_t *c[K];
  uint32_t ctr[K];
  bool quad[K]; /* the 4 items are 4 consecutive pixels of one row (and all < nitems) */
#pragma unroll
  for (unsigned k = 0; k < K; k++) {
    idx[k] = blockIdx.x * kChunkItems + k * 1024u + tid * 4u;
    const unsigned yy = div_by(idx[k] < nitems ? idx[k] : 0u, iw, div_magic);
    xo[k] = (idx[k] < nitems ? idx[k] : 0u) - yy * iw;
    c[k] = sf + (size_t)(3 + yy) * w + 3 + xo[k];
    quad[k] = idx[k] < nitems && xo[k] + 3 < iw;
    ctr[k] = 0;
  }
#pragma unroll
  for (unsigned k = 0; k < K; k++)
    if (quad[k]) ctr[k] = load_u32_unaligned(c[k]);
  uint64_t up[K], mid[K], dn[K];
#pragma unroll
  for (unsigned k = 0; k < K; k++) {
    up[k] = mid[k] = dn[k] = 0;
    if (ctr[k]) { /* bytes x-1 .. x+6 of the three rows as two dwords each; pixel j's neighbours are window bytes j, j+1, j+2 */
      up[k] = load_u32_unaligned(c[k] - W - 1) | ((uint64_t)load_u32_unaligned(c[k] - W + 3) << 32);
      mid[k] = load_u32_unaligned(c[k] - 1) | ((uint64_t)load_u32_unaligned(c[k] + 3) << 32);
      dn[k] = load_u32_unaligned(c[k] + W - 1) | ((uint64_t)load_u32_unaligned(c[k] + W + 3) << 32);
    }
  }
#pragma unroll
  for (unsigned k = 0; k < K; k++) {
    bool kp[4] = {false, false, false, false};
    if (ctr[k]) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned s = (ctr[k] >> (8 * j)) & 0xffu;
        auto byte = [](uint64_t v, int i) { return (unsigned)(v >> (8 * i)) & 0xffu; };
        unsigned m = byte(up[k], j);
        m = byte(up[k], j + 1) > m ? byte(up[k], j + 1) : m, m = byte(up[k], j + 2) > m ? byte(up[k], j + 2) : m;
        m = byte(mid[k], j) > m ? byte(mid[k], j) : m, m = byte(mid[k], j + 2) > m ? byte(mid[k], j + 2) : m;
        m = byte(dn[k], j) > m ? byte(dn[k], j) : m, m = byte(dn[k], j + 1) > m ? byte(dn[k], j + 1) : m;
        m = byte(dn[k], j + 2) > m ? byte(dn[k], j + 2) : m;
        kp[j] = s != 0 && !(m > s);
      }
    } else if (idx[k] < nitems && !quad[k]) { /* the group crosses a row end or the end of the frame: item by item */
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned ij = idx[k] + (unsigned)j;
        if (ij < nitems) {
          const unsigned yj = div_by(ij, iw, div_magic), xj = ij - yj * iw; /* iw < 4: several row ends */
          kp[j] = fast_is_peak(sf + (size_t)(3 + yj) * w + 3 + xj, W);
        }
      }
    }
    const uint64_t b0 = ballot(kp[0]), b1 = ballot(kp[1]), b2 = ballot(kp[2]), b3 = ballot(kp[3]);
    /* the four ballots are the group's four mask words in slot-major form (k_emit<F, QUAD>) */
    hits += (unsigned)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
    if (lane_id() == 0) {
      const size_t w0 = chunk * kChunkWords + k * 16u + wv * 4u;
      mask[w0] = b0, mask[w0 + 1] = b1, mask[w0 + 2] = b2, mask[w0 + 3] = b3;
    }
  }
  /* one block = one chunk: its count is a plain store (no atomics on a zeroed array, no zeroing pass) */
  if (lane_id() == 0) wave_hits[wv] = hits;
  __syncthreads();
  if (tid == 0) chunk_count[chunk] = wave_hits[0] + wave_hits[1] + wave_hits[2] + wave_hits[3];
}

/* compaction functor: item -> gs_keypoint {{x,y}, score, 0, {0}} (ref :530), 48 B = 12 dwords */
/* gs_fast with a score map smaller than the image (ref :512, :518-524: the map is written through gs_set
 * and read through gs_get, i.e. positions outside it read 0): zero them between the two passes.
 * grid (ceil(w/64), ceil(h/4)), block (64, 4) */
__global__ __launch_bounds__(256) void k_fast_clip(uint8_t *score, unsigned w, unsigned h, unsigned clip_w,
                                                  unsigned clip_h) {
  const unsigned x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
  if (x < w && y < h && (x >= clip_w || y >= clip_h)) score[(size_t)y * w + x] = 0;
}

struct FastEmit {
  const uint8_t *score;
  unsigned w;
  size_t frame_bytes;
  unsigned *kps; /* n frames x nkps x 12 u32 */
  unsigned nkps;
  bool aligned16; /* kps is 16-byte aligned: a 48-byte record is three dwordx4 stores instead of twelve 4-byte ones */
  GS_DEV void operator()(unsigned frame, size_t item, unsigned r) const {
    const unsigned iw = w - 6;
    const unsigned it = (unsigned)item; /* (w-6)*(h-6) fits 32 bits (checked by the launcher's item count) */
    const unsigned yy = it / iw, x = 3 + (it - yy * iw), y = 3 + yy;
    unsigned *o = kps + ((size_t)frame * nkps + r) * 12u;
    const unsigned sc = score[(size_t)frame * frame_bytes + (size_t)y * w + x];
    if (aligned16) {
      store_u32x4(o, U4{x, y, sc, 0}), store_u32x4(o + 4, U4{0, 0, 0, 0}), store_u32x4(o + 8, U4{0, 0, 0, 0});
    } else {
      o[0] = x, o[1] = y, o[2] = sc;
#pragma unroll
      for (int i = 3; i < 12; i++) o[i] = 0;
    }
  }
};

}  // namespace gs
#endif

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
 * VALU-bound (16 ring pixels x two class masks + the minimum |v - p|: ~1.3 VALU wave-instructions per
 * pixel on textured frames); three score kernels: k_fast_score_tile (default: ring bytes from an LDS
 * tile), k_fast_score4 (strip form), k_fast_score_px (one global byte load per ring pixel).
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
  /* v, p <= 255; t in [256, 2^32-256] behaves like 256 (never brighter, p - t wraps).  Round 5: the class bits are pushed
   * into the masks with compare + add-with-carry pairs (push_gt_u32: two full-rate instructions per ring pixel and class;
   * the sign-bit form before it paid a half-rate v_alignbit_b32 each), in the reference's own UNSIGNED arithmetic:
   * lo = p - t wraps to a huge value when p < t and every ring pixel then compares below it. */
  const unsigned t = threshold < 256u ? threshold : 256u, hi = p + t, lo = p - t;
  uint32_t bright = 0, dark = 0;
  unsigned mind = 255;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    bright = push_gt_u32(bright, v[j], hi); /* v > p + t */
    dark = push_gt_u32(dark, lo, v[j]);     /* v < p - t (unsigned) */
    mind = umin(mind, absdiff_u16(v[j], p));
  }
  if (p < t) dark = bright ^ 0xffffu; /* the wrap: d = !b && v < huge = !b (ref :496-498) */
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

/* pass 1, default: the same per-pixel evaluation, but the 17 bytes of a pixel come from an LDS tile instead of
 * 17 global byte loads.  k_fast_score_px spends its time in the texture addresser: 4.5 M wave-level byte loads per
 * 32 x 720p launch at ~8 cycles each are ~90 % of its 98 us, while its VALU is 54 % busy
 * (profiles/r02i_pmc_features.txt).  Here a block of 256 threads copies the (16 + 6) x (64 + 6) pixel region of
 * its 64 x 16 output tile with ~1.5 (unaligned) dword loads per thread and every ring pixel is a ds_read_u8
 * (consecutive lanes read consecutive bytes: conflict-free).  grid (ceil((w-6)/64), ceil((h-6)/16), n), block (64,4);
 * thread (tx, ty) scores rows ty, ty+4, ty+8, ty+12 of the tile. */
constexpr unsigned kFastTileRows = 16, kFastTileDw = 18; /* 72 bytes per tile row: 64 + 6, rounded up to dwords */
__global__ __launch_bounds__(256) void k_fast_score_tile(const uint8_t *img, uint8_t *score, unsigned w, unsigned h,
                                                         size_t frame_bytes, unsigned threshold) {
  __shared__ uint32_t tile32[(kFastTileRows + 6) * kFastTileDw];
  const uint8_t *frame = img + (size_t)blockIdx.z * frame_bytes;
  const unsigned tid = threadIdx.y * 64u + threadIdx.x;
  const unsigned x_t = blockIdx.x * 64u, y_t = blockIdx.y * kFastTileRows; /* image position of tile byte (0, 0) */
  for (unsigned i = tid; i < (kFastTileRows + 6) * kFastTileDw; i += 256u) {
    const unsigned r = i / kFastTileDw, c = i - r * kFastTileDw;
    /* rows / columns past the image are only ever ring pixels of positions that are not scored: whatever the
     * frame holds there will do, but the last bytes of the frame are real ring pixels -- byte by byte */
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
  __syncthreads();
  const uint8_t *tb = (const uint8_t *)tile32;
  constexpr int S = (int)kFastTileDw * 4; /* tile row stride in bytes */
  const unsigned x = 3 + x_t + threadIdx.x;
#pragma unroll
  for (unsigned k = 0; k < kFastTileRows / 4; k++) {
    const unsigned ry = threadIdx.y + 4u * k, y = 3 + y_t + ry;
    const bool in = x + 3 < w && y + 3 < h;
    const uint8_t *c = tb + (ry + 3) * S + threadIdx.x + 3;
    const unsigned p = c[0], v0 = c[-3 * S], v4 = c[3], v8 = c[3 * S], v12 = c[-3];
    const bool cand = in && fast_compass_candidate(p, v0, v4, v8, v12, threshold);
    unsigned sc = 0;
    if (ballot(cand) != 0) { /* wave-uniform */
      const unsigned v[16] = {v0,  c[-3 * S + 1], c[-2 * S + 2], c[-S + 3],
                              v4,  c[S + 3],      c[2 * S + 2],  c[3 * S + 1],
                              v8,  c[3 * S - 1],  c[2 * S - 2],  c[S - 3],
                              v12, c[-S - 3],     c[-2 * S - 2], c[-3 * S - 1]};
      sc = fast_score(p, v, threshold);
    }
    if (in) score[(size_t)blockIdx.z * frame_bytes + (size_t)y * w + x] = (uint8_t)sc;
  }
}

/* pass 1 with block-local candidate compaction.  k_fast_score_tile is VALU-bound (0.70 of the issue rate,
 * profiles/fast_valu_pmc.json) and a wave scores all 64 of its pixels as soon as ONE passes the compass filter;
 * on the block-noise frames of configs[3] nearly every wave holds one, but only a quarter of the pixels do.  Here
 * the 1024 pixels of the tile go through the compass filter first (5 LDS bytes each); those that pass are queued
 * in LDS (one ds_add per wave row, order irrelevant) and the queue is scored 64 candidates to a wave -- every lane
 * of every fast_score is a real candidate; the others' score is 0 and is stored right away.  Tiles in which most
 * pixels pass (regions with p < threshold: the reference's unsigned wrap makes every pixel a candidate) gain
 * nothing and lose the queue round trip, so from half the tile on the rows are scored in place like
 * k_fast_score_tile does.  Same scores, same stores.  grid / block as k_fast_score_tile. */
__global__ __launch_bounds__(256) void k_fast_score_cq(const uint8_t *img, uint8_t *score, unsigned w, unsigned h,
                                                       size_t frame_bytes, unsigned threshold) {
  __shared__ uint32_t tile32[(kFastTileRows + 6) * kFastTileDw];
  __shared__ uint16_t queue[64 * kFastTileRows];
  __shared__ unsigned qn;
  const uint8_t *frame = img + (size_t)blockIdx.z * frame_bytes;
  uint8_t *out = score + (size_t)blockIdx.z * frame_bytes;
  const unsigned tid = threadIdx.y * 64u + threadIdx.x;
  const unsigned x_t = blockIdx.x * 64u, y_t = blockIdx.y * kFastTileRows;
  if (tid == 0) qn = 0;
  for (unsigned i = tid; i < (kFastTileRows + 6) * kFastTileDw; i += 256u) {
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
  __syncthreads();
  const uint8_t *tb = (const uint8_t *)tile32;
  constexpr int S = (int)kFastTileDw * 4;
  auto score_at = [&](const uint8_t *c) {
    const unsigned v[16] = {c[-3 * S],     c[-3 * S + 1], c[-2 * S + 2], c[-S + 3], c[3],  c[S + 3],  c[2 * S + 2],  c[3 * S + 1],
                            c[3 * S],      c[3 * S - 1],  c[2 * S - 2],  c[S - 3],  c[-3], c[-S - 3], c[-2 * S - 2], c[-3 * S - 1]};
    return fast_score(c[0], v, threshold);
  };
  const unsigned x = 3 + x_t + threadIdx.x;
  unsigned mine = 0; /* bit k: this thread's pixel of row ty + 4k passed the compass filter */
#pragma unroll
  for (unsigned k = 0; k < kFastTileRows / 4; k++) {
    const unsigned ry = threadIdx.y + 4u * k, y = 3 + y_t + ry;
    const bool in = x + 3 < w && y + 3 < h;
    const uint8_t *c = tb + (ry + 3) * S + threadIdx.x + 3;
    const bool cand = in && fast_compass_candidate(c[0], c[-3 * S], c[3], c[3 * S], c[-3], threshold);
    mine |= (cand ? 1u : 0u) << k;
    if (in && !cand) out[(size_t)y * w + x] = 0;
  }
  const unsigned total = wave_sum((unsigned)__popc(mine)); /* this wave's candidates */
  __shared__ unsigned wtot[4];
  if ((tid & 63u) == 0) wtot[tid >> 6] = total;
  __syncthreads();
  const unsigned ncand = wtot[0] + wtot[1] + wtot[2] + wtot[3]; /* block-uniform */
  if (ncand == 0) return;
  if (ncand * 2u >= 64u * kFastTileRows) { /* dense tile: in place, a whole wave row at a time */
#pragma unroll
    for (unsigned k = 0; k < kFastTileRows / 4; k++) {
      const unsigned ry = threadIdx.y + 4u * k, y = 3 + y_t + ry;
      if (ballot((mine >> k) & 1u) != 0) { /* wave-uniform */
        const unsigned sc = score_at(tb + (ry + 3) * S + threadIdx.x + 3);
        if ((mine >> k) & 1u) out[(size_t)y * w + x] = (uint8_t)sc;
      }
    }
    return;
  }
#pragma unroll
  for (unsigned k = 0; k < kFastTileRows / 4; k++) {
    const bool cand = (mine >> k) & 1u;
    const uint64_t m = ballot(cand);
    if (m) {
      const unsigned lane = lane_id();
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(&qn, (unsigned)__popcll(m));
      base = readlane0(base);
      if (cand) queue[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((threadIdx.y + 4u * k) * 64u + threadIdx.x);
    }
  }
  __syncthreads();
  for (unsigned i0 = 0; i0 < ncand; i0 += 256u) { /* block-uniform trip count */
    const unsigned i = i0 + tid;
    const unsigned e = queue[i < ncand ? i : ncand - 1u], ry = e >> 6, tx = e & 63u;
    const unsigned sc = score_at(tb + (ry + 3) * S + tx + 3);
    if (i < ncand) out[(size_t)(3 + y_t + ry) * w + 3 + x_t + tx] = (uint8_t)sc;
  }
}

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
                                                       unsigned *zero_words, unsigned zero_n,
                                                       unsigned long long *nz = nullptr, size_t nz_frame_words = 0) {
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
  /* the chunk counters of pass 2 are zeroed here (zero_n words, spread over the grid) instead of by a 5-us fill launch */
  for (unsigned i = blockIdx.x * NT + threadIdx.y * 64u + threadIdx.x; i < zero_n; i += gridDim.x * NT) zero_words[i] = 0;
  unsigned tile = blockIdx.x;
  if (xcd_share) {
    tile = (blockIdx.x & 7u) * xcd_share + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
  }
  const unsigned tcol = tile % tiles_x, trow = (tile / tiles_x) % tiles_y, tframe = tile / (tiles_x * tiles_y);
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
    constexpr unsigned dr = NT / kFastTileDw, dc = NT - dr * kFastTileDw;
    unsigned r = tid / kFastTileDw, c = tid - r * kFastTileDw;
    const uint8_t *p0 = frame + (size_t)y_t * w + x_t;
#pragma unroll
    for (unsigned i = tid, it = 0; it < ((ROWS + 6) * kFastTileDw + NT - 1) / NT; it++, i += NT) {
      if (i < (ROWS + 6) * kFastTileDw) tile32[i] = load_u32_unaligned(p0 + r * w + c * 4u);
      r += dr, c += dc;
      if (c >= kFastTileDw) c -= kFastTileDw, r++;
    }
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

/* pass 1, strips (w % 4 == 0, 4-byte aligned frames, threshold <= 0xffffff00): a lane owns 4
 * consecutive pixels (one dword per row), a wave 256 px of a row, and walks DOWN a band of T rows
 * with the 7 image rows y-3..y+3 in registers as 12-byte windows (L, C, R: the neighbour lanes'
 * dwords come by v_mov_b32_dpp, only lanes 0 / 63 fetch a halo dword), so every image byte is
 * loaded once per band and every ring pixel is a static byte of a register.
 *   Per row the wave first runs the compass filter on packed u16 pairs: with S2 / s2 the second
 * largest / second smallest of the four compass pixels (8 packed min/max per pair of pixels),
 * ">= 2 brighter" is S2 > p + t and ">= 2 darker" is s2 < p - t; p < t (the reference's unsigned
 * wrap, ref :496-498) always passes.  Only pixel slots in which some lane of the wave passes are
 * scored with fast_score -- the same function as the per-pixel kernel, so scores are identical.
 * grid (ceil(w/256), ceil(bands/4), n), block (64, 4): the 4 waves of a block are 4 bands. */
struct FastRow { uint32_t L, C, R; };
template <int B> GS_DEV unsigned fast_wbyte(const FastRow &r) { /* byte B (0..11) of the window */
  static_assert(B >= 0 && B < 12, "window byte");
  const uint32_t d = B < 4 ? r.L : B < 8 ? r.C : r.R;
  return (d >> (8 * (B & 3))) & 0xffu;
}
template <int K> GS_DEV unsigned fast_score_slot(const FastRow (&rw)[7], unsigned threshold) {
  /* rw[d + 3] = image row y + d; pixel K is window byte 4 + K; ring order of ref :485-486 */
  const unsigned p = fast_wbyte<4 + K>(rw[3]);
  const unsigned v[16] = {fast_wbyte<4 + K>(rw[0]),     fast_wbyte<4 + K + 1>(rw[0]), fast_wbyte<4 + K + 2>(rw[1]),
                          fast_wbyte<4 + K + 3>(rw[2]), fast_wbyte<4 + K + 3>(rw[3]), fast_wbyte<4 + K + 3>(rw[4]),
                          fast_wbyte<4 + K + 2>(rw[5]), fast_wbyte<4 + K + 1>(rw[6]), fast_wbyte<4 + K>(rw[6]),
                          fast_wbyte<4 + K - 1>(rw[6]), fast_wbyte<4 + K - 2>(rw[5]), fast_wbyte<4 + K - 3>(rw[4]),
                          fast_wbyte<4 + K - 3>(rw[3]), fast_wbyte<4 + K - 3>(rw[2]), fast_wbyte<4 + K - 2>(rw[1]),
                          fast_wbyte<4 + K - 1>(rw[0])};
  return fast_score(p, v, threshold);
}

__global__ __launch_bounds__(256) void k_fast_score4(const uint8_t *img, uint8_t *score, unsigned w,
                                                     unsigned h, unsigned T, size_t frame_bytes,
                                                     unsigned threshold) {
  const BufRsrc src = make_buf(img + (size_t)blockIdx.z * frame_bytes, frame_bytes);
  const unsigned lane = threadIdx.x & 63u, x0 = (blockIdx.x * 64u + threadIdx.x) * 4u;
  const unsigned band = uniform(blockIdx.y * blockDim.y + threadIdx.y);
  const int y0 = 3 + (int)(band * T);
  if (y0 >= (int)h - 3) return; /* whole wave */
  const int nrows = ((int)h - 3 - y0) < (int)T ? ((int)h - 3 - y0) : (int)T;
  uint8_t *out = score + (size_t)blockIdx.z * frame_bytes;
  struct Raw { uint32_t c, hh; };
  auto load = [&](int y) { /* rows outside the image are never used by an interior pixel */
    const bool ok = (unsigned)y < h && x0 < w;
    const uint32_t base = (uint32_t)y * w + x0;
    uint32_t ho = kOOB;
    if (lane == 0 && x0 > 0) ho = base - 4;
    if (lane == 63 && x0 + 4 < w) ho = base + 4;
    Raw r;
    r.c = buf_load4(src, ok ? base : kOOB);
    r.hh = buf_load4(src, ok ? ho : kOOB);
    return r;
  };
  auto widen = [&](const Raw &r) { return FastRow{wave_shr1(r.c, r.hh), r.c, wave_shl1(r.c, r.hh)}; };
  /* which of this lane's 4 pixels are interior columns (3 <= x < w - 3) */
  bool col_in[4];
#pragma unroll
  for (int k = 0; k < 4; k++) col_in[k] = x0 + k >= 3u && x0 + k + 3u < w;
  const uint32_t in01 = (col_in[0] ? 0xffffu : 0u) | (col_in[1] ? 0xffff0000u : 0u);
  const uint32_t in23 = (col_in[2] ? 0xffffu : 0u) | (col_in[3] ? 0xffff0000u : 0u);
  const bool whole = col_in[0] && col_in[3];
  const uint32_t t16 = threshold < 256u ? threshold : 256u, tt = t16 | (t16 << 16);

  FastRow ring[7]; /* iteration I (mod 7): image row y + d sits in slot (I + d + 3) % 7 */
  static_for<6>([&](auto K) { ring[decltype(K)::value] = widen(load(y0 - 3 + decltype(K)::value)); });
  Raw raw = load(y0 + 3);
  for (int base = 0; base < nrows; base += 7) {
    static_for<7>([&](auto I) {
      constexpr int ii = decltype(I)::value;
      const int i = base + ii;
      if (i >= nrows) return; /* wave-uniform */
      const int y = y0 + i;
      ring[(ii + 6) % 7] = widen(raw);
      raw = load(y + 4);
      const FastRow(&r0)[7] = ring;
      const FastRow rw[7] = {r0[(ii + 0) % 7], r0[(ii + 1) % 7], r0[(ii + 2) % 7], r0[(ii + 3) % 7],
                             r0[(ii + 4) % 7], r0[(ii + 5) % 7], r0[(ii + 6) % 7]};
      /* compass filter, pixels (0,1) and (2,3) as u16 pairs */
      uint32_t cand[2];
#pragma unroll
      for (int hp = 0; hp < 2; hp++) {
        const uint32_t P = hp ? unpack_hi(rw[3].C) : unpack_lo(rw[3].C);
        const uint32_t a = hp ? unpack_hi(rw[0].C) : unpack_lo(rw[0].C);  /* ( 0, -3) */
        const uint32_t c = hp ? unpack_hi(rw[6].C) : unpack_lo(rw[6].C);  /* ( 0, +3) */
        /* (+3, 0): window bytes 7+k;  (-3, 0): window bytes 1+k */
        const uint32_t b = hp ? perm_b32(rw[3].R, rw[3].C, 0x0c060c05u) : perm_b32(rw[3].R, rw[3].C, 0x0c040c03u);
        const uint32_t d = hp ? perm_b32(rw[3].C, rw[3].L, 0x0c040c03u) : perm_b32(rw[3].C, rw[3].L, 0x0c020c01u);
        const uint32_t x = pk_max_u16(a, b), yv = pk_min_u16(a, b), z = pk_max_u16(c, d), u = pk_min_u16(c, d);
        const uint32_t mid_hi = pk_min_u16(x, z), mid_lo = pk_max_u16(yv, u);
        const uint32_t S2 = pk_max_u16(mid_hi, mid_lo); /* second largest  */
        const uint32_t s2 = pk_min_u16(mid_hi, mid_lo); /* second smallest */
        const uint32_t bright2 = pk_subsat_u16(S2, pk_add_u16(P, tt));   /* != 0 <=> S2 > p + t */
        const uint32_t dark2 = pk_subsat_u16(pk_subsat_u16(P, tt), s2);  /* != 0 <=> s2 < p - t (p >= t) */
        const uint32_t wrap = pk_subsat_u16(tt, P);                      /* != 0 <=> p < t */
        cand[hp] = bright2 | dark2 | wrap;
      }
      cand[0] &= in01, cand[1] &= in23;
      unsigned s[4] = {0, 0, 0, 0};
      if (ballot((cand[0] | cand[1]) != 0) != 0) { /* wave-uniform; slot by slot */
        if (ballot((cand[0] & 0xffffu) != 0) != 0) s[0] = fast_score_slot<0>(rw, threshold);
        if (ballot((cand[0] >> 16) != 0) != 0) s[1] = fast_score_slot<1>(rw, threshold);
        if (ballot((cand[1] & 0xffffu) != 0) != 0) s[2] = fast_score_slot<2>(rw, threshold);
        if (ballot((cand[1] >> 16) != 0) != 0) s[3] = fast_score_slot<3>(rw, threshold);
      }
      if (x0 < w) {
        uint8_t *o = out + (size_t)y * w + x0;
        if (whole) {
          *(uint32_t *)o = s[0] | (s[1] << 8) | (s[2] << 16) | (s[3] << 24);
        } else { /* lanes holding border columns: the 3-px frame of the scoremap is never written (ref :489) */
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (col_in[k]) o[k] = (uint8_t)s[k];
        }
      }
    });
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

/*
 * k_fast_nms.h -- pass 2 of gs_fast (grayskull.h:518-529) behind k_fast_score_q4: strict 3 x 3 maximum flags of the SCORED
 * pixels only, published as mask words of the ordered compaction (k_compact.h).
 *
 * k_fast_nms (k_fast.h) walks the score map item by item with dependent loads -- score dword, then its neighbourhood -- on
 * few waves: 29 us per 32 x 720p with the waves waiting 90 % of the time (profiles/r02l_pmc_features.txt).  Round 3 ran the
 * 3 x 3 maximum over every pixel on the strip machinery instead (23 us); round 4's sparse pass below visits only the 3 %
 * of the pixels that have a score (9.5 us) and the strip form was deleted in round 5.
 *
 * Round 5 measured the ordered emit INSIDE this kernel (one launch less: a status word per chunk, zeroed by the score
 * kernel; every wave publishes its count and adds up its predecessors' -- as a decoupled look-back to the nearest published
 * prefix, then as one scan over all earlier chunks): 84 resp. 79 us per 32 x 720p against 70 with the separate k_emit launch
 * (flat frames 50 against 39).  Device-scope loads of words other workgroups just wrote cost more than the kernel boundary
 * they replace; not kept (profiles/r05h_fast_lookback_not_kept.log, r05i_fast_nms_emit_one_launch_not_kept.log).
 *
 * Items are numbered over the interior padded to whole words: item = (y - 3) * 64 tiles_x + (x - 3), so a 64-px tile row of
 * the score kernel is one mask word.  Raster order of the items is the reference's emit order (ref :518-530);
 * FastEmitPadded maps an item back to (x, y).
 */
#ifndef GS_K_FAST_NMS_H
#define GS_K_FAST_NMS_H
#include "k_compact.h"
#include "k_strip.h"

namespace gs {

/* pass 2, sparse (round 4; the default behind k_fast_score_q4, which leaves a bitmap of the pixels with a non-zero score).
 * On the configs[3] frames 3 % of the pixels have a score.  Only those are visited: one wave per chunk of 32 bitmap words (the
 * item numbering of k_emit: word (y - 3) * tiles_x + tcol, bit qx <-> pixel (3 + 64 tcol + qx, y)); the set bits are
 * expanded into an LDS queue -- a lane takes the 32 items of half a word -- so that every lane then tests ONE scored pixel
 * per trip: its 3 x 3 neighbourhood is three unaligned dword loads from the score map (x - 1 .. x + 2 of the rows y - 1, y,
 * y + 1: inside the row because x <= w - 4), a pixel is kept when its score is non-zero (the clip pass may have
 * zeroed it since) and no neighbour's is larger (ref :518-528; ties keep both).  The kept bits are
 * collected per word in LDS and the wave stores the chunk's 32 mask words and its counter: no atomics in memory, nothing to
 * zero beforehand.  Raster order of the items is the reference's emit order.  grid (ceil(nchunks / 4), n), block 256. */
__global__ __launch_bounds__(256) void k_fast_nms_sparse(const uint8_t *score, unsigned w, size_t frame_bytes,
                                                         const unsigned long long *nz, unsigned long long *mask,
                                                         unsigned *chunk_count, unsigned tiles_x, unsigned nwords,
                                                         unsigned nchunks) {
  __shared__ uint16_t queue[4][kChunkItems];
  __shared__ uint32_t kept[4][2 * kChunkWords], wy[4][kChunkWords], wc[4][kChunkWords];
  const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const unsigned c = uniform(blockIdx.x * 4u + wv);
  const bool live = c < nchunks; /* whole wave; the block's barriers are reached by all four */
  const size_t fc = (size_t)blockIdx.y * nchunks + (live ? c : 0u);
  const unsigned W = c * kChunkWords + lane; /* lanes < 32: word of the frame */
  const unsigned long long mine = (live && lane < kChunkWords && W < nwords) ? nz[fc * kChunkWords + lane] : 0ull;
  if (lane < kChunkWords) {
    const unsigned y = W / tiles_x;
    wy[wv][lane] = y, wc[wv][lane] = W - y * tiles_x;
  }
  kept[wv][lane] = 0;
  const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
  const uint32_t hlo = shfl(mlo, (int)(lane >> 1)), hhi = shfl(mhi, (int)(lane >> 1));
  uint32_t half = (lane & 1u) ? hhi : hlo; /* items 32 lane .. 32 lane + 31 of the chunk */
  const unsigned pc = (unsigned)__popc(half), incl = wave_incl_scan(pc);
  const unsigned total = readlane_at(incl, 63);
  unsigned at = incl - pc;
  while (half != 0u) {
    const unsigned b = (unsigned)__ffs((int)half) - 1u;
    half &= half - 1u;
    queue[wv][at++] = (uint16_t)(lane * 32u + b);
  }
  __syncthreads();
  const uint8_t *sf = score + (size_t)blockIdx.y * frame_bytes;
  for (unsigned j = lane; j < total; j += 64u) {
    const unsigned e = queue[wv][j], k = e >> 6;
    const unsigned x = 3u + 64u * wc[wv][k] + (e & 63u), y = 3u + wy[wv][k];
    const uint8_t *p = sf + (size_t)y * w + x - 1u;
    const uint32_t a = load_u32_unaligned(p - w), m = load_u32_unaligned(p), b = load_u32_unaligned(p + w);
    const unsigned sc = (m >> 8) & 0xffu;
    unsigned mx = umax(umax(a & 0xffu, (a >> 8) & 0xffu), (a >> 16) & 0xffu);
    mx = umax(mx, umax(m & 0xffu, (m >> 16) & 0xffu));
    mx = umax(mx, umax(umax(b & 0xffu, (b >> 8) & 0xffu), (b >> 16) & 0xffu));
    if (sc != 0u && mx <= sc) atomicOr(&kept[wv][e >> 5], 1u << (e & 31u));
  }
  __syncthreads();
  if (!live) return;
  unsigned long long word = 0;
  if (lane < kChunkWords) {
    word = (unsigned long long)kept[wv][2u * lane] | ((unsigned long long)kept[wv][2u * lane + 1u] << 32);
    mask[fc * kChunkWords + lane] = word;
  }
  const unsigned n = wave_sum((unsigned)__popcll(word));
  if (lane == 0) chunk_count[fc] = n;
}

/* compaction functor for the padded item numbering: item -> gs_keypoint {{x,y}, score, 0, {0}} (ref :530) */
struct FastEmitPadded {
  const uint8_t *score;
  unsigned w, wp; /* wp = 64 * words per row */
  size_t frame_bytes;
  unsigned *kps; /* n frames x nkps x 12 u32 */
  unsigned nkps;
  bool aligned16;
  unsigned off = 0; /* item (0, 0) is pixel (off, off): 3 behind k_fast_nms_sparse, whose words start at the interior */
  GS_DEV void operator()(unsigned frame, size_t item, unsigned r) const {
    const unsigned it = (unsigned)item, y0 = it / wp, x = it - y0 * wp + off, y = y0 + off;
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

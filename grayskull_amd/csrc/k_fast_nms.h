/*
 * k_fast_nms.h -- pass 2 of gs_fast (grayskull.h:518-529) in strip form: strict 3 x 3 maximum flags for every
 * interior pixel, published as mask words of the ordered compaction (k_compact.h).
 *
 * k_fast_nms (k_fast.h) walks the score map item by item with dependent loads -- score dword, then its
 * neighbourhood -- on few waves: 29 us per 32 x 720p with the waves waiting 90 % of the time
 * (profiles/r02l_pmc_features.txt).  But "no neighbour is larger" is "the 3 x 3 maximum equals the centre", i.e. the
 * dilation the strip machinery already does at HBM rate: lane = 16 consecutive pixels (one 16-byte load per row),
 * wave = 1024 px of a row walking down a band, three rows of horizontal 3-maxima in registers.  A pixel is a
 * keypoint iff its score is non-zero and equals the 3 x 3 maximum (ties survive, ref :524); the frame of the
 * caller's score map is read as it is (never written by pass 1, ref :489), exactly like the reference reads it.
 *
 * Items are numbered over the frame padded to whole words: item = y * wp + x with wp = 64 * ceil(w / 64), so the
 * 16 flags of a lane are 16 consecutive bits and four neighbouring lanes (a DPP quad) make one mask word: no
 * ballots, one 8-byte store per quad and row.  Raster order of the items is the reference's emit order (ref
 * :518-530); FastEmitPadded maps an item back to (x, y).  Chunk counters take one atomic per non-empty word
 * (keypoints are sparse); the launcher zeroes them.
 */
#ifndef GS_K_FAST_NMS_H
#define GS_K_FAST_NMS_H
#include "k_compact.h"
#include "k_strip.h"

namespace gs {

/* grid / block from strip_cfg(w, h - 6, n); score: n frames of w x h; mask: n x nchunks * 32 words, a frame's h * wpr
 * words first (rows
 * 0..2 and h-3.. are cleared by the first / last band); chunk_count: n x nchunks, pre-zeroed */
__global__ __launch_bounds__(256) void k_fast_nms16(const uint8_t *score, unsigned w, unsigned h, unsigned T,
                                                    size_t frame_bytes, unsigned long long *mask, unsigned *chunk_count,
                                                    unsigned wpr, unsigned nchunks) {
  const Strip<> S(score, const_cast<uint8_t *>(score), w, h, frame_bytes);
  if (S.wave_outside()) return; /* block wider than the frame */
  const int y0 = 3 + (int)(S.band * T);
  if (y0 >= (int)h - 3) return; /* whole wave */
  const int nrows = ((int)h - 3 - y0) < (int)T ? ((int)h - 3 - y0) : (int)T;
  /* interior columns 3 .. w-4 of this lane's 16.  Any width: the strips stay on the 16-px grid (no stores of pixels here),
   * the last lane of a ragged row reads on into the next row -- still inside the frame, the rows below y + 2 <= h - 2
   * included -- and those columns never reach a flag. */
  unsigned colmask = 0u;
  if (S.x0 < w) {
    const int lo = S.x0 >= 3u ? 0 : 3 - (int)S.x0, hi = (int)w - 4 - (int)S.x0 > 15 ? 15 : (int)w - 4 - (int)S.x0;
    if (hi >= lo) colmask = ((2u << hi) - 1u) & ~((1u << lo) - 1u);
  }
  auto hpass = [](const uint32_t(&U)[12], uint32_t(&H)[8]) { /* max of px x-1, x, x+1 for the own pairs */
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int j = k + 2;
      H[k] = pk_max_u16(pk_max_u16(alignbit(U[j], U[j - 1], 16), U[j]), alignbit(U[j + 1], U[j], 16));
    }
  };
  uint32_t ring[3][8], C[2][8]; /* horizontal maxima of rows y-1, y, y+1; own pairs of the last two rows loaded */
  {
    uint32_t U[12];
    S.unpack(S.load(y0 - 1), U);
    hpass(U, ring[0]);
    S.unpack(S.load(y0), U);
    hpass(U, ring[1]);
#pragma unroll
    for (int k = 0; k < 8; k++) C[1][k] = U[k + 2];
  }
  unsigned long long *mf = mask + (size_t)blockIdx.z * nchunks * kChunkWords; /* a frame's words fill whole chunks */
  unsigned *cf = chunk_count + (size_t)blockIdx.z * nchunks;
  const unsigned lane = S.lane, wx = S.x0 >> 6; /* word column of this lane's quad */
  /* the three frame rows above / below the interior hold no items: the first / last band clears their words, so
   * the mask needs no zeroing pass */
  if ((lane & 3u) == 0u && S.x0 < w) {
    if (y0 == 3)
      for (unsigned r = 0; r < 3; r++) mf[(size_t)r * wpr + wx] = 0ull;
    if (y0 + nrows == (int)h - 3) {
      for (unsigned r = h - 3; r < h; r++) mf[(size_t)r * wpr + wx] = 0ull;
      if (S.x0 == 0) /* and the words between the last row and the end of the last chunk (k_emit reads them) */
        for (size_t i = (size_t)h * wpr; i < (size_t)nchunks * kChunkWords; i++) mf[i] = 0ull;
    }
  }
  RawRow raw = S.load(y0 + 1);
  for (int base = 0; base < nrows; base += 6) {
    static_for<6>([&](auto I) {
      constexpr int ii = decltype(I)::value, ia = ii % 3, ib = (ii + 1) % 3, ic = (ii + 2) % 3, cc = (ii + 1) % 2, cn = ii % 2;
      const int i = base + ii;
      if (i >= nrows) return; /* wave-uniform */
      const int y = y0 + i;
      /* a 1024-px span of a score row is empty more often than not (55 % of them on the configs[3] frames): then its
       * horizontal maxima are 0 without unpacking anything (round 4; wave-uniform) */
      const RawRow cur = raw;
      raw = S.load(y + 2); /* in flight during the arithmetic */
      const uint32_t rowany = cur.v.x | cur.v.y | cur.v.z | cur.v.w | cur.hh;
      if (ballot(rowany != 0) != 0) {
        uint32_t U[12];
        S.unpack(cur, U);
        hpass(U, ring[ic]);
#pragma unroll
        for (int k = 0; k < 8; k++) C[cn][k] = U[k + 2]; /* row y+1: the centre of the next iteration */
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) ring[ic][k] = 0, C[cn][k] = 0;
      }
      /* centre row y sits in C[cc]: any score at all in this 1024-px span? */
      uint32_t any = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) any |= C[cc][k];
      unsigned m16 = 0;
      if (ballot(any != 0) != 0) { /* wave-uniform */
        uint32_t nib[4];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const uint32_t M = pk_max_u16(pk_max_u16(ring[ia][k], ring[ib][k]), ring[ic][k]);
          const uint32_t ne = pk_min_u16(M ^ C[cc][k], 0x00010001u); /* 1 where the maximum is another pixel's */
          const uint32_t nz = pk_min_u16(C[cc][k], 0x00010001u);     /* 1 where the score is non-zero */
          const uint32_t pk = nz & (ne ^ 0x00010001u);               /* bit 0 / bit 16: pixel 2k / 2k+1 is a peak */
          const uint32_t two = (pk | (pk >> 15)) & 3u;
          if (k % 2 == 0) nib[k / 2] = two;
          else nib[k / 2] |= two << 2;
        }
        m16 = (nib[0] | (nib[1] << 4) | (nib[2] << 8) | (nib[3] << 12)) & colmask;
      }
      /* four lanes = 64 pixels = one mask word: lane 4j collects it */
      const uint32_t pair = m16 | (quad_perm<1, 0, 3, 2>(m16) << 16); /* even lanes: (own, next) */
      const uint32_t hi = quad_perm<2, 2, 2, 2>(pair);
      if ((lane & 3u) == 0u && S.x0 < w) {
        const unsigned long long word = ((unsigned long long)hi << 32) | pair;
        const size_t widx = (size_t)y * wpr + wx;
        mf[widx] = word;
        if (word) atomicAdd(&cf[widx / kChunkWords], (unsigned)__popcll(word));
      }
    });
  }
}

/* compaction functor for the padded item numbering: item -> gs_keypoint {{x,y}, score, 0, {0}} (ref :530) */
struct FastEmitPadded {
  const uint8_t *score;
  unsigned w, wp; /* wp = 64 * words per row */
  size_t frame_bytes;
  unsigned *kps; /* n frames x nkps x 12 u32 */
  unsigned nkps;
  bool aligned16;
  GS_DEV void operator()(unsigned frame, size_t item, unsigned r) const {
    const unsigned it = (unsigned)item, y = it / wp, x = it - y * wp;
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

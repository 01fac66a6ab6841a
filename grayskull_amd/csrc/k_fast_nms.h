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
  /* The arithmetic per row is a few dozen instructions and the launch is a few thousand waves (32 x 720p: 5,760), all
   * resident at once: a wave's time is its chain of row loads.  So the loads run six rows ahead (round 4; one row ahead:
   * 32 x 720p NMS 18 us) -- a band of 8 rows has all but two of its 10 loads in flight before the first one is used. */
  constexpr int PF = 6;
  RawRow raws[PF];
  {
    const RawRow ra = S.load(y0 - 1), rb = S.load(y0);
#pragma unroll
    for (int j = 0; j < PF; j++)
      if (j < nrows) raws[j] = S.load(y0 + 1 + j); /* wave-uniform */
    uint32_t U[12];
    S.unpack(ra, U);
    hpass(U, ring[0]);
    S.unpack(rb, U);
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
  for (int base = 0; base < nrows; base += 6) {
    static_for<6>([&](auto I) {
      constexpr int ii = decltype(I)::value, ia = ii % 3, ib = (ii + 1) % 3, ic = (ii + 2) % 3, cc = (ii + 1) % 2, cn = ii % 2;
      const int i = base + ii;
      if (i >= nrows) return; /* wave-uniform */
      const int y = y0 + i;
      /* a 1024-px span of a score row is empty more often than not (55 % of them on the configs[3] frames): then its
       * horizontal maxima are 0 without unpacking anything (round 4; wave-uniform) */
      const RawRow cur = raws[ii];
      if (i + PF < nrows) raws[ii] = S.load(y + 1 + PF); /* wave-uniform; PF == the unroll, so the slots are static */
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

/* pass 2, sparse (round 4; the default behind k_fast_score_q4, which leaves a bitmap of the pixels with a non-zero score).
 * On the configs[3] frames 3 % of the pixels have a score and k_fast_nms16 above spends its 23 us per 32 x 720p running the
 * 3 x 3 maximum over the other 97 %.  Here only the scored pixels are visited: one wave per chunk of 32 bitmap words (the
 * item numbering of k_emit: word (y - 3) * tiles_x + tcol, bit qx <-> pixel (3 + 64 tcol + qx, y)); the set bits are
 * expanded into an LDS queue -- a lane takes the 32 items of half a word -- so that every lane then tests ONE scored pixel
 * per trip: its 3 x 3 neighbourhood is three unaligned dword loads from the score map (x - 1 .. x + 2 of the rows y - 1, y,
 * y + 1: inside the row because x <= w - 4), a pixel is kept when its score is non-zero (the clip pass may have
 * zeroed it since) and no neighbour's is larger (ref :518-528; ties keep both, like k_fast_nms16).  The kept bits are
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

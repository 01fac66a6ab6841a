/*
 * k_fast_fused.h -- gs_fast (grayskull.h:482-534) passes 1 and 2 in ONE walk over the frame: FAST-9 scores, the
 * strict 3 x 3 maximum test and the mask words of the ordered compaction (k_compact.h) from the same LDS tiles.
 *
 * The two-pass form (k_fast_score_q4 + k_fast_nms16) is instruction-bound, not traffic-bound: per 64 x 16 tile a
 * block spends ~400 instructions per thread, most of them before the first pixel is looked at (tile coordinates,
 * the tile load with its per-dword address arithmetic and halo re-load, two barriers), and the NMS pass spends
 * another ~300 per 1024 px unpacking and comparing every pixel of a score map that is 96 % zeros
 * (profiles/r03q_fast_xcd_tiles.log: the score pass costs the same with 4.1x or 1.0x its bytes fetched).
 *   Here a block of 256 threads owns a column strip of 62 interior pixels and walks DOWN a band of rows, 16 score
 * rows per step:
 *   - the image tile (22 rows x 72 bytes: 16 + 6 ring rows, 64 + 6 ring columns + 2 halo columns) lives in one of two
 *     LDS buffers; a step requests the 16 rows of the NEXT tile from memory before it starts its own arithmetic and
 *     copies the 6 rows both tiles share LDS to LDS, so no row is fetched twice and no load is waited for;
 *   - thread (r, q) runs the packed compass filter on score pixels 4q .. 4q+3 of row r exactly as k_fast_score_q4,
 *     stores the four zero scores as one dword, queues the pixels that pass, and the queue is scored from the tile
 *     with fast_score (the same function as every other score kernel);
 *   - scores also go to an LDS score tile (18 rows x 64 columns: two rows carried over from the previous step, one
 *     halo column either side computed redundantly, 2 / 62), where the caller's never-written 3-px frame of the
 *     score map (ref :489, :524) is loaded as it is; the 3 x 3 maximum test then reads LDS only, one row behind the
 *     scores (a row's test needs the row below), on packed u16 pairs as k_fast_nms16 does;
 *   - keypoint flags are OR-ed into per-row LDS words (rare) and one thread per row publishes them: atomicOr into
 *     the zeroed mask of the padded item numbering (item = y * 64 * wpr + x) and one atomicAdd per non-empty word
 *     into the chunk counters -- the input k_emit<FastEmitPadded> already takes.
 * A band owns 16 m - 2 rows so that m steps cover it (the test of its last row needs one more score row; 2 / 16 m
 * redundant).  Work items (strip, band, frame) are numbered with the strip fastest and dealt to the XCDs in eighths,
 * as in k_fast_score_q4.  Results are bit-identical to the two-pass form (tests run both).
 */
#ifndef GS_K_FAST_FUSED_H
#define GS_K_FAST_FUSED_H
#include "k_fast.h"

namespace gs {

constexpr int kFfCols = 62;                       /* owned interior columns per strip */
constexpr int kFfImgRows = 22, kFfImgDw = 18;     /* image tile: rows R-3 .. R+18, byte columns X0-4 .. X0+67 */
constexpr int kFfScRows = 18, kFfScDw = 18;       /* score tile: rows R-2 .. R+15; dwords 1..16 = score columns 0..63 */

struct FastFusedArgs {
  const uint8_t *img;
  uint8_t *score;
  unsigned w, h;
  size_t frame_bytes;
  unsigned threshold; /* <= 0xffffff00 */
  unsigned strips, bands, m, nitems, xcd_share;
  unsigned long long *mask; /* n x nchunks x kChunkWords, zeroed */
  unsigned *chunk_count;    /* n x nchunks, zeroed */
  unsigned wpr, nchunks;
};

/* dword at byte offset off of the frame, bytes outside [0, frame_bytes) read 0 (tile rows / columns beyond the image:
 * never part of a pixel that is scored) */
GS_DEV uint32_t ff_img_dword(const uint8_t *frame, long off, size_t frame_bytes) {
  if (off >= 0 && (size_t)off + 4 <= frame_bytes) return load_u32_unaligned(frame + off);
  uint32_t v = 0;
  for (int b = 0; b < 4; b++) {
    const long o = off + b;
    if (o >= 0 && (size_t)o < frame_bytes) v |= (uint32_t)frame[o] << (8 * b);
  }
  return v;
}

__global__ __launch_bounds__(256) void k_fast_fused(FastFusedArgs a) {
  __shared__ uint32_t ib[2][kFfImgRows * kFfImgDw + 2];
  __shared__ uint32_t sb[2][kFfScRows * kFfScDw];
  __shared__ uint16_t queue[64 * 16];
  __shared__ unsigned qn;
  __shared__ uint32_t rowbits[32]; /* [row][2]: flags of the 64 score columns of the 16 rows tested this step */
  unsigned item = blockIdx.x;
  if (a.xcd_share) {
    item = (blockIdx.x & 7u) * a.xcd_share + (blockIdx.x >> 3);
    if (item >= a.nitems) return;
  }
  const unsigned sx = item % a.strips, t1 = item / a.strips, bnd = t1 % a.bands, f = t1 / a.bands;
  const size_t fb = a.frame_bytes;
  const uint8_t *frame = a.img + (size_t)f * fb;
  uint8_t *out = a.score + (size_t)f * fb;
  unsigned long long *mf = a.mask + (size_t)f * a.nchunks * kChunkWords;
  unsigned *cf = a.chunk_count + (size_t)f * a.nchunks;
  const int w = (int)a.w, h = (int)a.h;
  const int X0 = 3 + kFfCols * (int)sx, Xe = X0 + kFfCols < w - 3 ? X0 + kFfCols : w - 3; /* owned columns [X0, Xe) */
  const int TB = 16 * (int)a.m - 2;
  const int Yb = 3 + TB * (int)bnd, Ye = Yb + TB < h - 3 ? Yb + TB : h - 3; /* owned rows [Yb, Ye) */
  const int K = (Ye - Yb + 1) / 16 + 1; /* steps: step k scores rows Yb-1+16k .. +15 and tests rows Yb-2+16k .. +15 */
  const unsigned tid = threadIdx.y * 64u + threadIdx.x, r = tid >> 4, q = tid & 15u;
  const unsigned threshold = a.threshold;

  /* publish the flags of the 16 rows tested in the step whose first score row was Rp (threads 0..15, one row each) */
  auto emit_rows = [&](int Rp) {
    if (tid < 16) {
      const unsigned long long bits = (unsigned long long)rowbits[2 * tid] | ((unsigned long long)rowbits[2 * tid + 1] << 32);
      if (bits) { /* bit j <-> column X0 - 1 + j of row Rp - 1 + tid (owned pixels only) */
        const unsigned y = (unsigned)(Rp - 1 + (int)tid), xb = (unsigned)(X0 - 1), sh = xb & 63u;
        const size_t widx = (size_t)y * a.wpr + (xb >> 6);
        const unsigned long long lo = bits << sh, hi = sh ? bits >> (64u - sh) : 0ull;
        if (lo) atomicOr(&mf[widx], lo), atomicAdd(&cf[widx / kChunkWords], (unsigned)__popcll(lo));
        if (hi) atomicOr(&mf[widx + 1], hi), atomicAdd(&cf[(widx + 1) / kChunkWords], (unsigned)__popcll(hi));
      }
    }
  };

  if (tid == 0) qn = 0, ib[0][kFfImgRows * kFfImgDw] = 0, ib[0][kFfImgRows * kFfImgDw + 1] = 0, ib[1][kFfImgRows * kFfImgDw] = 0,
                ib[1][kFfImgRows * kFfImgDw + 1] = 0;
  if (tid < 32) rowbits[tid] = 0;
  for (unsigned i = tid; i < 2u * kFfScRows * kFfScDw; i += 256u) (&sb[0][0])[i] = 0;
  for (unsigned i = tid; i < (unsigned)(kFfImgRows * kFfImgDw); i += 256u) {
    const int rr = (int)(i / kFfImgDw), c = (int)i - rr * kFfImgDw;
    ib[0][i] = ff_img_dword(frame, (long)(Yb - 4 + rr) * w + (X0 - 4) + 4 * c, fb);
  }
  __syncthreads();

  const uint32_t t16 = threshold < 256u ? threshold : 256u, tt = t16 | (t16 << 16);
  constexpr int S = kFfImgDw * 4, SS = kFfScDw * 4; /* row strides in bytes */
  uint32_t p0 = 0, p1 = 0;
  for (int k = 0; k < K; k++) {
    const unsigned cur = (unsigned)k & 1u;
    const int R = Yb - 1 + 16 * k; /* first score row of this step */
    uint32_t *IB = ib[cur], *SB = sb[cur];
    if (k > 0) emit_rows(R - 16);
    const bool more = k + 1 < K;
    if (more) { /* the 16 image rows only the next tile has: R+19 .. R+34, requested now, parked in LDS at the end of the step */
      const int r0 = (int)(tid / kFfImgDw), c0 = (int)tid - r0 * kFfImgDw;
      p0 = ff_img_dword(frame, (long)(R + 19 + r0) * w + (X0 - 4) + 4 * c0, fb);
      if (tid < 32) {
        const int i1 = 256 + (int)tid, r1 = i1 / kFfImgDw, c1 = i1 - r1 * kFfImgDw;
        p1 = ff_img_dword(frame, (long)(R + 19 + r1) * w + (X0 - 4) + 4 * c1, fb);
      }
    }
    /* ---- compass filter on score pixels (x + 0..3, y) (k_fast_score_q4's, tile byte columns 4q+3 .. 4q+6 of row r+3) */
    const int x = X0 - 1 + 4 * (int)q, y = R + (int)r;
    const uint32_t *rc = IB + (r + 3) * kFfImgDw + q, *ru = IB + r * kFfImgDw + q, *rd = IB + (r + 6) * kFfImgDw + q;
    const uint32_t d0 = rc[0], d1 = rc[1], d2 = rc[2];
    const uint32_t C = alignbit(d1, d0, 24), L = d0, Rr = alignbit(d2, d1, 16);
    const uint32_t U = alignbit(ru[1], ru[0], 24), D = alignbit(rd[1], rd[0], 24);
    unsigned cand = 0;
#pragma unroll
    for (int hp = 0; hp < 2; hp++) {
      const uint32_t P = hp ? unpack_hi(C) : unpack_lo(C), va = hp ? unpack_hi(U) : unpack_lo(U), vc = hp ? unpack_hi(D) : unpack_lo(D);
      const uint32_t vb = hp ? unpack_hi(Rr) : unpack_lo(Rr), vd = hp ? unpack_hi(L) : unpack_lo(L);
      const uint32_t mx = pk_max_u16(va, vb), mn = pk_min_u16(va, vb), z = pk_max_u16(vc, vd), u = pk_min_u16(vc, vd);
      const uint32_t mid_hi = pk_min_u16(mx, z), mid_lo = pk_max_u16(mn, u);
      const uint32_t S2 = pk_max_u16(mid_hi, mid_lo), s2 = pk_min_u16(mid_hi, mid_lo);
      const uint32_t pass = pk_subsat_u16(S2, pk_add_u16(P, tt)) | pk_subsat_u16(pk_subsat_u16(P, tt), s2) | pk_subsat_u16(tt, P);
      cand |= ((pass & 0xffffu) ? 1u : 0u) << (2 * hp) | ((pass >> 16) ? 1u : 0u) << (2 * hp + 1);
    }
    /* interior: 3 <= x < w-3, 3 <= y < h-3; owned: inside this block's strip and band as well */
    const bool yin = y >= 3 && y < h - 3, yown = y >= Yb && y < Ye;
    unsigned inmask = 0, own = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int xi = x + j;
      const bool in = yin && xi >= 3 && xi < w - 3;
      inmask |= (in ? 1u : 0u) << j;
      own |= (in && yown && xi >= X0 && xi < Xe ? 1u : 0u) << j;
    }
    cand &= inmask;
    if (own == 15u) {
      *(gs_u32_unaligned *)(out + (size_t)y * w + x) = 0u; /* candidates are overwritten behind the barrier */
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if ((own >> j) & 1u) out[(size_t)y * w + x + j] = 0;
    }
    /* score tile: zeros where scores will come, the caller's bytes where the frame of the score map shows through */
    uint32_t init = 0;
    if (inmask != 15u && y >= 0 && y < h) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int xi = x + j;
        if (!((inmask >> j) & 1u) && xi >= 0 && xi < w) init |= (uint32_t)out[(size_t)y * w + xi] << (8 * j);
      }
    }
    SB[(2 + r) * kFfScDw + 1 + q] = init;
#pragma unroll
    for (unsigned j = 0; j < 4; j++) { /* queue the candidates: one LDS atomic per wave and slot */
      const bool ck = (cand >> j) & 1u;
      const uint64_t m = ballot(ck);
      if (m) {
        const unsigned lane = lane_id();
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&qn, (unsigned)__popcll(m));
        base = readlane0(base);
        if (ck) queue[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(r * 64u + 4u * q + j);
      }
    }
    __syncthreads(); /* B1: tile initialised, queue complete, zero stores ordered before the score stores below */
    const unsigned ncand = qn;
    const uint8_t *tb = (const uint8_t *)IB;
    uint8_t *sbytes = (uint8_t *)SB;
    auto put = [&](unsigned qy, unsigned qx, unsigned sc) { /* score of score pixel (column qx, row qy) of this step */
      if (sc) {
        sbytes[(2 + qy) * SS + 4 + qx] = (uint8_t)sc;
        const int xi = X0 - 1 + (int)qx, yi = R + (int)qy;
        if (xi >= X0 && xi < Xe && yi >= Yb && yi < Ye) out[(size_t)yi * w + xi] = (uint8_t)sc;
      }
    };
    if (ncand * 2u >= 64u * 16u) { /* most of the tile passes: every thread scores its own pixels in place */
#pragma unroll
      for (unsigned j = 0; j < 4; j++) {
        if (ballot((cand >> j) & 1u) == 0) continue; /* wave-uniform */
        const uint8_t *c = tb + (r + 3) * S + 4u * q + j + 3;
        const unsigned v[16] = {c[-3 * S],     c[-3 * S + 1], c[-2 * S + 2], c[-S + 3], c[3],  c[S + 3],  c[2 * S + 2],  c[3 * S + 1],
                                c[3 * S],      c[3 * S - 1],  c[2 * S - 2],  c[S - 3],  c[-3], c[-S - 3], c[-2 * S - 2], c[-3 * S - 1]};
        const unsigned sc = fast_score(c[0], v, threshold);
        if ((cand >> j) & 1u) put(r, 4u * q + j, sc);
      }
    } else {
      for (unsigned i0 = 0; i0 < ncand; i0 += 256u) { /* block-uniform trip count */
        const unsigned i = i0 + tid;
        if (i0 + (tid & ~63u) >= ncand) continue; /* whole wave past the queue's end */
        const unsigned e = queue[i < ncand ? i : ncand - 1u], qy = e >> 6, qx = e & 63u;
        const uint8_t *c = tb + (qy + 3) * S + qx + 3;
        const unsigned v[16] = {c[-3 * S],     c[-3 * S + 1], c[-2 * S + 2], c[-S + 3], c[3],  c[S + 3],  c[2 * S + 2],  c[3 * S + 1],
                                c[3 * S],      c[3 * S - 1],  c[2 * S - 2],  c[S - 3],  c[-3], c[-S - 3], c[-2 * S - 2], c[-3 * S - 1]};
        const unsigned sc = fast_score(c[0], v, threshold);
        if (i < ncand) put(qy, qx, sc);
      }
    }
    if (tid < 32) rowbits[tid] = 0; /* last step's flags were published before B1 */
    __syncthreads(); /* B2: the score tile is complete */
    /* ---- 3 x 3 maximum test of row R - 1 + r (score tile row 1 + r), columns 4q .. 4q+3: a keypoint is a non-zero
     * score that equals the maximum of its neighbourhood (ties survive, ref :524) */
    {
      const int yn = R - 1 + (int)r;
      unsigned ownn = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int xi = x + j;
        ownn |= (yn >= Yb && yn < Ye && xi >= X0 && xi < Xe ? 1u : 0u) << j;
      }
      uint32_t M01 = 0, M23 = 0, C01 = 0, C23 = 0;
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const uint32_t *row = SB + (r + t) * kFfScDw + q;
        const uint32_t Dm = row[0], Dc = row[1], Dp = row[2]; /* columns 4q-4 .. 4q+7; needed: 4q-1 .. 4q+4 */
        const uint32_t Pa = (Dm >> 24) | ((Dc & 0xffu) << 16), Pb = ((Dc >> 8) & 0xffu) | ((Dc >> 16) & 0xffu) << 16,
                       Pc = (Dc >> 24) | ((Dp & 0xffu) << 16);
        const uint32_t Qa = alignbit(Pb, Pa, 16), Qb = alignbit(Pc, Pb, 16); /* (b0, b1), (b2, b3) */
        const uint32_t H01 = pk_max_u16(pk_max_u16(Pa, Qa), Pb), H23 = pk_max_u16(pk_max_u16(Pb, Qb), Pc);
        M01 = pk_max_u16(M01, H01), M23 = pk_max_u16(M23, H23);
        if (t == 1) C01 = Qa, C23 = Qb;
      }
      unsigned fl = 0;
#pragma unroll
      for (int hp = 0; hp < 2; hp++) {
        const uint32_t M = hp ? M23 : M01, Cc = hp ? C23 : C01;
        const uint32_t ne = pk_min_u16(M ^ Cc, 0x00010001u), nz = pk_min_u16(Cc, 0x00010001u);
        const uint32_t pk = nz & (ne ^ 0x00010001u);
        fl |= ((pk | (pk >> 15)) & 3u) << (2 * hp);
      }
      fl &= ownn;
      if (fl) atomicOr(&rowbits[2u * r + (q >> 3)], fl << (4u * (q & 7u)));
    }
    if (tid == 0) qn = 0;
    if (more) { /* next tile: its 16 own rows from the registers, the 6 shared rows and the 2 shared score rows by copy */
      uint32_t *IN = ib[cur ^ 1u], *SN = sb[cur ^ 1u];
      IN[6 * kFfImgDw + tid] = p0;
      if (tid < 32) IN[6 * kFfImgDw + 256 + tid] = p1;
      if (tid >= 64 && tid < 64 + 6 * kFfImgDw) IN[tid - 64] = IB[16 * kFfImgDw + tid - 64];
      if (tid >= 192 && tid < 192 + 2 * kFfScDw) SN[tid - 192] = SB[16 * kFfScDw + tid - 192];
    }
    __syncthreads(); /* B3 */
  }
  emit_rows(Yb - 1 + 16 * (K - 1));
}

}  // namespace gs
#endif

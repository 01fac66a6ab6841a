/*
 * k_tmatch.h -- gs_match_template (grayskull.h:705-724) on the matrix cores.
 *
 * The sum of squared differences over a tw x th window is the one dense contraction of this library:
 *   SSD(x, y) = sum I'^2 - 2 sum I' T' + sum T'^2      with I' = I - 128, T' = T - 128 (signed bytes; the shift cancels
 * in I - T).  The cross term is a correlation, and a correlation along x is a matrix product with a Toeplitz matrix:
 *   C[m][n] = sum_k A[m][k] B[k][n],   A[m][k] = I'[y0 + m + j][x0 + k],   B[k][n] = T'[j][k - n]  (0 outside the row)
 * gives C[m][n] = sum_i I'[y0 + m + j][x0 + n + i] T'[j][i], template row j's share of the 32 x 32 results at
 * (x0 + n, y0 + m); summed over j in the MFMA accumulator.  K runs over tw + 31 (the Toeplitz band), in steps of 32:
 * v_mfma_i32_32x32x32_i8, efficiency tw / (32 ceil((tw + 31) / 32)).  A operand of lane (m, g): 16 consecutive image
 * bytes of row y0 + m + j; B operand of lane (n, g): 16 consecutive bytes of the zero-padded template row, starting
 * 32 kc + 16 g - n bytes in -- both straight out of LDS (the image region of the block and the template, both
 * shifted to signed bytes once when they are staged).  A wave owns 32 rows x 64 columns of results (two accumulator
 * tiles: the right tile's A operand at step kc is the left tile's at kc + 1, so each LDS read feeds two MFMAs), a
 * block 2 x 2 waves = 64 x 128 results.
 *   sum I'^2 over the window comes from a row-prefix pass and a sliding column pass (k_tm_rowprefix, k_tm_colsq: u32, exact
 * modulo 2^32 and the window sums themselves are below it while
 * tw th <= 32768, which also keeps |sum I' T'| < 2^31), sum T'^2 comes with the padded template (k_tm_prep).
 * The quotient SSD 255 / (tw th 255^2) is a float estimate made exact by its 64-bit remainder.
 */
#ifndef GS_K_TMATCH_H
#define GS_K_TMATCH_H
#include "prims.h"

namespace gs {

struct TmArgs {
  const uint8_t *img;
  unsigned iw, ih;
  const uint8_t *tpad; /* th rows of tstride bytes: 32 zeros, the row as signed bytes, zeros (k_tm_prep) */
  const unsigned *tsq; /* sum (T - 128)^2 */
  unsigned tw, th;
  const unsigned *s2; /* rh x rw window sums of (I - 128)^2 */
  uint8_t *result;
  unsigned rw, rh;
  unsigned nkc;     /* K steps of 32: ceil((tw + 31) / 32) */
  unsigned istride; /* LDS image row: 96 + 32 nkc bytes + 16 (an odd number of 16-byte slots: conflict-free b128 reads) */
  unsigned tstride; /* template row: 32 zeros, the row, zeros up to 32 nkc + 48 (a multiple of 16) */
};

/* the template as the MFMA kernel wants it: th rows of tstride bytes (32 zeros, T ^ 0x80, zeros), and sum (T - 128)^2.
 * One block of 1024 threads (a template has at most 32768 taps). */
__global__ __launch_bounds__(1024) void k_tm_prep(const uint8_t *tmpl, unsigned tw, unsigned th, unsigned tstride, uint8_t *tpad,
                                                   unsigned *tsq) {
  __shared__ unsigned part[16];
  const unsigned tid = threadIdx.x, tdw = tstride / 4u, ntd = th * tdw;
  unsigned t2 = 0;
  for (unsigned i = tid; i < ntd; i += 1024u) {
    const unsigned j = i / tdw, q = 4u * (i - j * tdw);
    uint32_t sv = 0;
    for (unsigned b = 0; b < 4; b++)
      if (q + b >= 32u && q + b - 32u < tw) {
        const unsigned t = tmpl[(size_t)j * tw + q + b - 32u];
        const int d = (int)t - 128;
        t2 += (unsigned)(d * d);
        sv |= (t ^ 0x80u) << (8 * b);
      }
    ((uint32_t *)tpad)[i] = sv;
  }
  t2 = wave_sum(t2);
  if ((tid & 63u) == 0) part[tid >> 6] = t2;
  __syncthreads();
  if (tid == 0) {
    unsigned t = 0;
    for (unsigned k = 0; k < 16; k++) t += part[k];
    *tsq = t;
  }
}

/* P[r][x] = sum_{i < x} (I[r][i] - 128)^2 for x = 0 .. iw (iw + 1 entries per row): one wave per row, 16 pixels per lane
 * and pass, a wave scan per pass.  grid (ceil(ih / 4)), block (64, 4) */
__global__ __launch_bounds__(256) void k_tm_rowprefix(const uint8_t *img, unsigned iw, unsigned ih, unsigned *P) {
  const unsigned r = blockIdx.x * 4u + threadIdx.y, lane = threadIdx.x;
  if (r >= ih) return; /* whole wave */
  const uint8_t *row = img + (size_t)r * iw;
  unsigned *o = P + (size_t)r * (iw + 1u);
  unsigned carry = 0;
  if (lane == 0) o[0] = 0;
  auto fetch = [&](unsigned x, uint32_t (&d)[4]) { /* 16 pixels from x (zeros past the row's end) */
    if (x + 16u <= iw) {
#pragma unroll
      for (unsigned q = 0; q < 4; q++) d[q] = load_u32_unaligned(row + x + 4u * q);
    } else {
#pragma unroll
      for (unsigned q = 0; q < 4; q++) d[q] = 0x80808080u; /* 128 - 128 = 0 */
      for (unsigned k = 0; k < 16 && x + k < iw; k++) d[k >> 2] = (d[k >> 2] & ~(0xffu << (8 * (k & 3)))) | ((uint32_t)row[x + k] << (8 * (k & 3)));
    }
  };
  uint32_t nxt[4];
  fetch(lane * 16u, nxt);
  for (unsigned x0 = 0; x0 < iw; x0 += 1024u) { /* wave-uniform; the next pass's pixels are requested before this pass's scan */
    const unsigned x = x0 + lane * 16u;
    const uint32_t cur[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
    if (x0 + 1024u < iw) fetch(x + 1024u, nxt);
    unsigned loc[16], run = 0;
#pragma unroll
    for (unsigned k = 0; k < 16; k++) {
      const int v = (int)((cur[k >> 2] >> (8 * (k & 3))) & 0xffu) - 128;
      run += (unsigned)(v * v);
      loc[k] = run;
    }
    const unsigned before = carry + wave_incl_scan(run) - run;
    if (x + 16u <= iw) { /* 64 consecutive bytes per lane */
      struct __attribute__((aligned(4))) Q4 { uint32_t a, b, c, d; }; /* one 16-byte store at a 4-byte aligned address */
#pragma unroll
      for (unsigned k = 0; k < 16; k += 4)
        *(Q4 *)(o + x + k + 1u) = Q4{before + loc[k], before + loc[k + 1], before + loc[k + 2], before + loc[k + 3]};
    } else {
#pragma unroll
      for (unsigned k = 0; k < 16; k++)
        if (x + k < iw) o[x + k + 1u] = before + loc[k];
    }
    carry = shfl(before + run, 63);
  }
}
/* s2[y][x] = sum_{j < th} (P[y + j][x + tw] - P[y + j][x]); a thread slides down kTmRun result rows of one column, the rows
 * that enter and leave requested 16 steps at a time.  grid (ceil(rw / 64), ceil(rh / kTmRun)), block 64 */
constexpr unsigned kTmRun = 64;
__global__ __launch_bounds__(64) void k_tm_colsq(const unsigned *P, unsigned iw, unsigned tw, unsigned th, unsigned rw, unsigned rh,
                                                 unsigned *s2) {
  const unsigned x = blockIdx.x * 64u + threadIdx.x, y0 = blockIdx.y * kTmRun;
  if (x >= rw) return;
  const size_t ps = (size_t)iw + 1u;
  const unsigned last = rh - 1u + th - 1u; /* last image row a window touches */
  auto h = [&](unsigned r) { /* rows past the image only feed results that are not stored */
    const unsigned rr = r <= last ? r : last;
    return P[(size_t)rr * ps + x + tw] - P[(size_t)rr * ps + x];
  };
  unsigned s = 0;
#pragma unroll 16
  for (unsigned j = 0; j < th; j++) s += h(y0 + j);
  s2[(size_t)y0 * rw + x] = s;
  for (unsigned k0 = 1; k0 < kTmRun && y0 + k0 < rh; k0 += 16u) { /* uniform per block */
    unsigned in[16], out[16];
#pragma unroll
    for (unsigned k = 0; k < 16; k++) in[k] = h(y0 + k0 + k + th - 1u), out[k] = h(y0 + k0 + k - 1u);
#pragma unroll
    for (unsigned k = 0; k < 16; k++) {
      s += in[k] - out[k];
      if (k0 + k < kTmRun && y0 + k0 + k < rh) s2[(size_t)(y0 + k0 + k) * rw + x] = s;
    }
  }
}

/* s2 from the integral table Q of (I - 128)^2 (k_integral.h, SQ; round 4): four corners per result, modulo 2^32 like the
 * table -- three short launches for the table and this one instead of the row-prefix + sliding-column passes above, which
 * move 140 MB through a u32 prefix table for a 4K frame (75 us; this route: see DESIGN.md 3).  grid (ceil(rw / 256), rh), block 256 */
__global__ __launch_bounds__(256) void k_tm_s2_corners(const unsigned *Q, unsigned iw, unsigned tw, unsigned th, unsigned rw, unsigned rh,
                                                      unsigned *s2) {
  const unsigned x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y;
  if (x >= rw) return;
  const unsigned *lo = Q + (size_t)(y + th - 1u) * iw, *hi = y ? Q + (size_t)(y - 1u) * iw : nullptr;
  unsigned s = lo[x + tw - 1u];
  if (x) s -= lo[x - 1u];
  if (hi) {
    s -= hi[x + tw - 1u];
    if (x) s += hi[x - 1u];
  }
  s2[(size_t)y * rw + x] = s;
}

/* SPLIT = 1: a block's four waves own 2 x 2 tiles of 32 rows x 64 columns (64 x 128 results per block; grid (ceil(rw / 128),
 * ceil(rh / 64))) -- least staging per result, for launches with enough blocks to fill the chip.  SPLIT = 4: the four
 * waves share ONE 32 x 64 tile and take every fourth template row each, their accumulators meet in LDS (grid (ceil(rw / 64),
 * ceil(rh / 32))): 8 x the blocks, for video-sized images where 64 x 128 tiles leave most CUs idle.
 * block 256, dynamic LDS max(irows * istride + th * tstride, SPLIT == 4 ? 32 KB : 0) with irows = (SPLIT == 1 ? 63 : 31) + th */
#ifndef GS_TM_VARIANT
#define GS_TM_VARIANT 0 /* timing experiments (wrong results): 1 no MFMA, 2 no operand loads in the loop, 3 no image staging, 4 no template staging, 5 no epilogue */
#endif
/* NK = a.nkc, the K steps of this template width (2 .. 9: one instantiation each, round 4): the operand registers are sized
 * for it (with arrays for 9 steps whatever the template, a 128-px template ran with 178 registers) and the row loop has no
 * run-time guards -- one basic block per template row, so the next row's LDS reads are scheduled between this row's MFMAs
 * (guards on a run-time nkc cut the loop into a block per MFMA pair) */
/* BAND (SPLIT = 1; round 4): the template is taken kTmBand rows at a time -- per band the block stages the TR - 1 + kTmBand image
 * rows those template rows meet and the band's template rows, between two barriers -- so a block holds 33 KB of LDS at
 * tw = 128 instead of 79 and FOUR blocks (16 waves, four per SIMD) fit a CU instead of two: the kernel is paced by LDS
 * latency under the occupancy its footprint allows.  The rows shared by consecutive bands are staged again
 * (they come from the L2). */
constexpr unsigned kTmBand = 32;
template <int SPLIT, unsigned NK = 9, bool BAND = false>
__global__ __launch_bounds__(256) void k_match_template_mfma(TmArgs a) {
  static_assert(SPLIT == 1 || !BAND, "the split form keeps the whole template resident");
  GS_DYN_LDS(smem);
  const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  constexpr unsigned TR = SPLIT == 1 ? 64u : 32u, TC = SPLIT == 1 ? 128u : 64u;
  const unsigned bx0 = blockIdx.x * TC, by0 = blockIdx.y * TR;
  const unsigned jband = BAND ? kTmBand : a.th; /* template rows resident at a time */
  const unsigned irows = TR - 1u + jband, idw = (a.istride - 16u) / 4u;
  uint8_t *limg = (uint8_t *)smem;
  uint8_t *ltm = limg + (size_t)irows * a.istride;
  const unsigned i16 = idw / 4u; /* idw is a multiple of 8 */
  /* image rows by0 + jc .. + TR - 2 + nj of the block's columns, as signed bytes (bytes outside the image: 0; they only ever
   * meet template zeros or results that are not stored), and template rows jc .. jc + nj - 1 (k_tm_prep's padded form): 16
   * bytes per item, eight items per thread requested before the first is stored (a trip is one memory round trip) */
  auto stage = [&](unsigned jc, unsigned nj) {
    const unsigned nitems = (TR - 1u + nj) * i16;
    for (unsigned base = tid; base < (GS_TM_VARIANT == 3 ? 0u : nitems); base += 2048u) {
      /* four scalar arrays, every index a constant once unrolled: an array of structs written under a condition, or bytes
       * inserted at a run-time index, live in scratch memory with hipcc (144 B per lane and a round trip through it for every
       * staged dword until round 4) */
      uint32_t v0[8], v1[8], v2[8], v3[8];
#pragma unroll
      for (unsigned u = 0; u < 8; u++) {
        const unsigned i = base + 256u * u;
        v0[u] = v1[u] = v2[u] = v3[u] = 0;
        if (i < nitems) {
          const unsigned r = i / i16, c = i - r * i16;
          const unsigned y = by0 + jc + r, x = bx0 + 16u * c;
          if (y < a.ih && x < a.iw) {
            const uint8_t *p = a.img + (size_t)y * a.iw + x;
            if (x + 16u <= a.iw) {
              v0[u] = load_u32_unaligned(p) ^ 0x80808080u, v1[u] = load_u32_unaligned(p + 4) ^ 0x80808080u;
              v2[u] = load_u32_unaligned(p + 8) ^ 0x80808080u, v3[u] = load_u32_unaligned(p + 12) ^ 0x80808080u;
            } else {
              uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
              for (unsigned b = 0; b < 4; b++) {
                if (x + b < a.iw) d0 |= (uint32_t)(p[b] ^ 0x80u) << (8 * b);
                if (x + 4 + b < a.iw) d1 |= (uint32_t)(p[4 + b] ^ 0x80u) << (8 * b);
                if (x + 8 + b < a.iw) d2 |= (uint32_t)(p[8 + b] ^ 0x80u) << (8 * b);
                if (x + 12 + b < a.iw) d3 |= (uint32_t)(p[12 + b] ^ 0x80u) << (8 * b);
              }
              v0[u] = d0, v1[u] = d1, v2[u] = d2, v3[u] = d3;
            }
          }
        }
      }
#pragma unroll
      for (unsigned u = 0; u < 8; u++) {
        const unsigned i = base + 256u * u;
        if (i < nitems) {
          const unsigned r = i / i16, c = i - r * i16;
          *(U4 *)(limg + (size_t)r * a.istride + 16u * c) = U4{v0[u], v1[u], v2[u], v3[u]};
        }
      }
    }
    const unsigned nt16 = nj * a.tstride / 16u; /* tstride is a multiple of 16 */
    const U4 *tsrc = (const U4 *)(a.tpad + (size_t)jc * a.tstride);
    for (unsigned base = tid; base < (GS_TM_VARIANT == 4 ? 0u : nt16); base += 2048u) {
      uint32_t v0[8], v1[8], v2[8], v3[8];
#pragma unroll
      for (unsigned u = 0; u < 8; u++) {
        v0[u] = v1[u] = v2[u] = v3[u] = 0;
        if (base + 256u * u < nt16) {
          const U4 t = tsrc[base + 256u * u];
          v0[u] = t.x, v1[u] = t.y, v2[u] = t.z, v3[u] = t.w;
        }
      }
#pragma unroll
      for (unsigned u = 0; u < 8; u++)
        if (base + 256u * u < nt16) ((U4 *)ltm)[base + 256u * u] = U4{v0[u], v1[u], v2[u], v3[u]};
    }
  };
  const unsigned wy = SPLIT == 1 ? (wave >> 1) * 32u : 0u, wx = SPLIT == 1 ? (wave & 1u) * 64u : 0u, m = lane & 31u, g = lane >> 5;
  const unsigned j0 = SPLIT == 1 ? 0u : wave, jstep = (unsigned)SPLIT;
  int32_t acc0[16], acc1[16];
#pragma unroll
  for (int r = 0; r < 16; r++) acc0[r] = 0, acc1[r] = 0;
  /* Row by row: the NK + 1 image operands and NK template operands of template row j sit in registers (static indices) and
   * the operands of row j + jstep are requested before row j's 2 NK MFMAs are issued.  j counts inside the resident band. */
  constexpr unsigned kTmMaxK = NK; /* == a.nkc (the launcher picks the instantiation) */
  U4 Ac[kTmMaxK + 1], Bc[kTmMaxK], An[kTmMaxK + 1], Bn[kTmMaxK];
  auto load_row = [&](unsigned j, U4 (&A)[kTmMaxK + 1], U4 (&B)[kTmMaxK]) {
    const uint8_t *ar = limg + (wy + m + j) * a.istride + wx + 16u * g;
    const uint8_t *tr = ltm + j * a.tstride;
    const unsigned ob = 16u * g + 32u - m, sh = 8u * (ob & 3u); /* 32 kc keeps the byte phase */
    const uint32_t *tp = (const uint32_t *)(tr + (ob & ~3u));
#pragma unroll
    for (unsigned k = 0; k <= kTmMaxK; k++) A[k] = *(const U4 *)(ar + 32u * k);
#pragma unroll
    for (unsigned k = 0; k < kTmMaxK; k++) {
      const uint32_t d0 = tp[8u * k], d1 = tp[8u * k + 1u], d2 = tp[8u * k + 2u], d3 = tp[8u * k + 3u], d4 = tp[8u * k + 4u];
      B[k] = U4{alignbit(d1, d0, sh), alignbit(d2, d1, sh), alignbit(d3, d2, sh), alignbit(d4, d3, sh)};
    }
  };
  for (unsigned jc = 0; jc < a.th; jc += jband) { /* block-uniform: one trip unless BAND */
    const unsigned nj = a.th - jc < jband ? a.th - jc : jband;
    if (jc) __syncthreads(); /* the previous band's operand reads are done */
    stage(jc, nj);
    __syncthreads();
    if (j0 < nj) { /* wave-uniform */
      load_row(j0, Ac, Bc);
      for (unsigned j = j0; j < nj; j += jstep) {
        /* no branch inside a trip: the last one requests its own row again (and drops it) */
        const unsigned jn = j + jstep < nj ? j + jstep : j;
        if (GS_TM_VARIANT != 2) load_row(jn, An, Bn);
#pragma unroll
        for (unsigned k = 0; k < kTmMaxK; k++) {
#if GS_TM_VARIANT == 1
          acc0[0] += (int)(Ac[k].x ^ Bc[k].x), acc1[0] += (int)(Ac[k + 1].x ^ Bc[k].y);
#else
          mfma_i32_32x32x32_i8(Ac[k], Bc[k], acc0);
          mfma_i32_32x32x32_i8(Ac[k + 1], Bc[k], acc1);
#endif
        }
        if (GS_TM_VARIANT != 2) {
#pragma unroll
          for (unsigned k = 0; k <= kTmMaxK; k++) Ac[k] = An[k];
#pragma unroll
          for (unsigned k = 0; k < kTmMaxK; k++) Bc[k] = Bn[k];
        }
      }
    }
  }
  /* results: register r of lane (n, g) is row (r & 3) + 8 (r >> 2) + 4 g of the tile, column n */
  /* score = floor(SSD 255 / (taps 255^2)) = floor(SSD / (255 taps)); SSD <= 65025 taps < 2^32 for taps <= 32768: a float
   * estimate of the quotient (within 1), exact through its 32-bit remainder */
  const unsigned dv = 255u * a.tw * a.th;
  const float inv = 1.0f / (float)dv;
  const unsigned tsq = *a.tsq;
  auto s2_of = [&](int c, int r) -> unsigned { /* requested for all of a lane's results before the first is used */
    const unsigned x = bx0 + wx + 32u * (unsigned)c + m;
    const unsigned y = by0 + wy + (unsigned)((r & 3) + 8 * (r >> 2)) + 4u * g;
    return (x < a.rw && y < a.rh) ? a.s2[(size_t)y * a.rw + x] : 0u;
  };
  auto finish = [&](int c, int r, int d, unsigned s2v) {
    const unsigned x = bx0 + wx + 32u * (unsigned)c + m;
    const unsigned y = by0 + wy + (unsigned)((r & 3) + 8 * (r >> 2)) + 4u * g;
    if (x < a.rw && y < a.rh && (GS_TM_VARIANT != 5 || d == 12345)) {
      const unsigned ssd = s2v + tsq - 2u * (unsigned)d; /* >= 0 and < 2^32: modular arithmetic is exact */
      unsigned q = (unsigned)((float)ssd * inv);
      int rem = (int)(ssd - q * dv); /* |true quotient - q| <= 1, so the remainder fits */
      if (rem < 0) q--;
      else if ((unsigned)rem >= dv) q++;
      const unsigned score = q < 255u ? q : 255u;
      a.result[(size_t)y * a.rw + x] = (uint8_t)(255u - score);
    }
  };
  if constexpr (SPLIT == 1) {
    unsigned sv[32];
#pragma unroll
    for (int r = 0; r < 16; r++) sv[r] = s2_of(0, r), sv[16 + r] = s2_of(1, r);
#pragma unroll
    for (int r = 0; r < 16; r++) finish(0, r, acc0[r], sv[r]), finish(1, r, acc1[r], sv[16 + r]);
  } else {
    /* the four partial tiles meet in LDS ([wave][32 values][lane], over the image region, which is done with);
     * wave w then finishes values w, w + 4, ... */
    __syncthreads();
    int32_t *red = (int32_t *)smem;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      red[(wave * 32u + (unsigned)r) * 64u + lane] = acc0[r];
      red[(wave * 32u + 16u + (unsigned)r) * 64u + lane] = acc1[r];
    }
    __syncthreads();
    unsigned sv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) sv[k] = s2_of((int)((wave + 4u * (unsigned)k) >> 4), (int)((wave + 4u * (unsigned)k) & 15u));
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const unsigned v = wave + 4u * (unsigned)k; /* 0 .. 31: tile v >> 4, register v & 15 */
      int d = 0;
#pragma unroll
      for (unsigned w = 0; w < 4; w++) d += red[(w * 32u + v) * 64u + lane];
      finish((int)(v >> 4), (int)(v & 15u), d, sv[k]);
    }
  }
}

}  // namespace gs
#endif

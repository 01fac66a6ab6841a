/*
 * k_stencil.h -- gs_sobel / gs_blur / gs_erode / gs_dilate (+ adaptive threshold, filter,
 * downsample) for gfx950.  Reference semantics: grayskull.h:268-320, :230-266, :189-197.
 *
 * Two families:
 *
 *  *16 "strip" kernels (the fast path; w % 16 == 0, 16-B aligned frames)
 *      lane  = 16 consecutive pixels of a row (one global_load_dwordx4, row-coalesced:
 *              a wave reads 1 KiB contiguous per row);
 *      wave  = 1024-px wide column block, walks DOWN a band of T rows keeping the vertical
 *              window in registers (ring indexed at compile time), so each input byte is
 *              loaded once per band (+ halo rows);
 *      halo  = horizontal neighbours come from the adjacent lanes with two
 *              v_mov_b32_dpp wave_shr:1 / wave_shl:1 per row (no LDS round trip); only
 *              lane 0 / lane 63 fetch one extra dword from the neighbouring wave's columns;
 *      math  = bytes unpacked with v_perm_b32 to u16 pairs and processed with packed
 *              v_pk_*_u16 ops (2 px per lane-op): ~9 (sobel), ~7.5 (blur r=2), ~3.5 (morph)
 *              lane-ops per pixel against a budget of ~30 at 60 % of HBM peak.
 *      All of these are HBM-bound: 2 B/px algorithmic traffic (1 read + 1 write).
 *
 *  *_px kernels (any w, h, alignment): one thread per pixel, byte accesses.
 */
#ifndef GS_K_STENCIL_H
#define GS_K_STENCIL_H
#include "k_strip.h"

namespace gs {

/* ------------------------------------------------------------------ sobel, strips (helpers: k_strip.h) */
template <bool KEEP_COLS, int RG = 0>
__global__ __launch_bounds__(256) void k_sobel16(uint8_t *dst, const uint8_t *src, unsigned w,
                                                 unsigned h, unsigned T, size_t frame_bytes) {
  const Strip<false, RG> S(src, dst, w, h, frame_bytes);
  if (S.wave_outside()) return; /* block wider than the frame */
  const int y0 = 1 + (int)(S.band * T);
  if (y0 >= (int)h - 1) return; /* whole wave */
  const int nrows = ((int)h - 1 - y0) < (int)T ? ((int)h - 1 - y0) : (int)T;
  SobelState st;
  {
    uint32_t U0[12], U1[12];
    S.unpack(S.load(y0 - 1), U0);
    S.unpack(S.load(y0), U1);
    st.init(U0, U1);
  }
  auto body = [&](auto I, int, const uint32_t(&U)[12]) { return st.template step<decltype(I)::value>(U); };
  constexpr bool EXITS = !GS_SOBEL_NOEXIT;
  if constexpr (KEEP_COLS) strip_rows<2, false, EXITS>(S, y0, nrows, 1, S.load(y0 + 1), body, SobelKeepCols(S));
  else strip_rows<2, false, EXITS>(S, y0, nrows, 1, S.load(y0 + 1), body);
}

/* host-staged gs_sobel: columns 0 and w-1 of rows 1..h-2 of the caller's dst, gathered on the
 * host into cols[2*h] (cols[2y], cols[2y+1]), are planted into the device copy so the kernel can
 * preserve them and whole rows can be copied back.  grid ceil(2h/256), block 256 */
__global__ __launch_bounds__(256) void k_put_cols(uint8_t *img, const uint8_t *cols, unsigned w,
                                                  unsigned h) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= 2 * h) return;
  const unsigned y = i >> 1;
  if (y < 1 || y + 1 >= h) return;
  img[(size_t)y * w + ((i & 1) ? w - 1 : 0)] = cols[i];
}

/* ------------------------------------------------------------------ box blur, strips (helpers: k_strip.h) */
/* ceil(2^24 / (N * cols)) for a run-time column count R+1 .. N (RAGGED: the lane left of the tail lane may own
 * pixels less than R from the right edge too; evaluated once per lane, before the row loop) */
template <int R> GS_DEV uint32_t blur_mul_cols(unsigned cols) {
  constexpr unsigned N = 2 * R + 1;
  uint32_t m = BlurMagic<R>::mul;
#pragma unroll
  for (unsigned c = R + 1; c < N; c++) m = cols == c ? (0x1000000u + N * c - 1u) / (N * c) : m;
  return m;
}
template <int R, int RG = 0>
__global__ __launch_bounds__(256) void k_blur16(uint8_t *dst, const uint8_t *src, unsigned w,
                                                unsigned h, unsigned T, size_t frame_bytes) {
  constexpr int N = 2 * R + 1;
  const Strip<false, RG> S(src, dst, w, h, frame_bytes);
  if (S.wave_outside()) return; /* block wider than the frame */
  const bool first = S.x0 == 0, last = S.x0 + 16 == w;
  const int y0 = (int)(S.band * T);
  if (y0 >= (int)h) return;
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  uint32_t ring[N][8], V[8];
#pragma unroll
  for (int k = 0; k < 8; k++) V[k] = 0, ring[N - 1][k] = 0;
#pragma unroll
  for (int r = 0; r < N - 1; r++) { /* image rows y0-R .. y0+R-1 */
    uint32_t U[12];
    S.unpack(S.load(y0 - R + r), U);
    blur_hsum<R>(U, ring[r]);
#pragma unroll
    for (int k = 0; k < 8; k++) V[k] = add2(V[k], ring[r][k]);
  }
  /* RAGGED: columns in the image for the lane's R rightmost pixels, from the distance to the right edge */
  uint32_t mRr[R];
  if constexpr (RG != 0) {
#pragma unroll
    for (int q = 0; q < R; q++) {
      const int d = (int)w - 1 - (int)(S.x0 + 16 - R + q); /* >= 0 for every lane inside the image */
      mRr[q] = blur_mul_cols<R>((unsigned)(R + 1 + (d < R ? d : R)));
    }
  }
  strip_rows<N, false, !GS_BLUR_NOEXIT>(S, y0, nrows, R, S.load(y0 + R), [&](auto I, int, const uint32_t(&U)[12]) {
    constexpr int slot = (decltype(I)::value + N - 1) % N; /* row i-1 leaves, row i+2R enters */
    uint32_t Hn[8];
    blur_hsum<R>(U, Hn);
    /* per-pixel multipliers: interior constant, except the R columns next to an image edge
     * (compile-time constants behind one select each; rows clipped vertically are rewritten by
     * k_blur_edge_rows afterwards) */
    uint32_t mL[R], mR[R];
    constexpr uint32_t mC = BlurMagic<R>::mul;
#pragma unroll
    for (int q = 0; q < R; q++) {
      mL[q] = first ? (0x1000000u + N * (R + 1 + q) - 1u) / (N * (R + 1 + q)) : mC;
      mR[q] = last ? (0x1000000u + N * (2 * R - q) - 1u) / (N * (2 * R - q)) : mC;
      if constexpr (RG != 0) mR[q] = mRr[q];
    }
    uint32_t od[4];
#pragma unroll
    for (int g = 0; g < 4; g++) { /* 4 pixels = 2 pairs -> one output dword */
      uint32_t p[4];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int k = 2 * g + t;
        V[k] = sub2(add2(V[k], Hn[k]), ring[slot][k]); /* unsigned fields, no carry / borrow: plain 32-bit ops */
        ring[slot][k] = Hn[k];
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) { /* quotient = byte 3 of the product */
          const int q = 2 * k + hlf;
          const uint32_t sv = hlf ? (V[k] >> 16) : (V[k] & 0xffffu);
          const uint32_t m = q < R ? mL[q < R ? q : 0] : q >= 16 - R ? mR[q >= 16 - R ? q - (16 - R) : 0] : mC;
          p[2 * t + hlf] = sv * m;
        }
      }
      od[g] = perm_b32(p[3], perm_b32(p[2], perm_b32(p[1], p[0], 0x0c0c0703u), 0x0c070100u), 0x07020100u);
    }
    U4 o{od[0], od[1], od[2], od[3]};
    return o;
  });
}

/* rows [0,R) and [h-R,h): the window is clipped vertically, divisor = in-image taps
 * (ref :275-281).  One thread per pixel, coalesced along x.  grid (ceil(w/256), 2R, n frames). */
__global__ __launch_bounds__(256) void k_blur_edge_rows(uint8_t *dst, const uint8_t *src, unsigned w,
                                                        unsigned h, int R, size_t frame_bytes) {
  const int x = (int)(blockIdx.x * 256u + threadIdx.x);
  if (x >= (int)w) return;
  const int ry = (int)blockIdx.y, y = ry < R ? ry : (int)h - 2 * R + ry;
  const uint8_t *f = src + (size_t)blockIdx.z * frame_bytes;
  const int xa = x - R < 0 ? 0 : x - R, xb = x + R > (int)w - 1 ? (int)w - 1 : x + R;
  const int ya = y - R < 0 ? 0 : y - R, yb = y + R > (int)h - 1 ? (int)h - 1 : y + R;
  unsigned sum = 0;
  for (int yy = ya; yy <= yb; yy++)
    for (int xx = xa; xx <= xb; xx++) sum += f[(size_t)yy * w + xx];
  const unsigned cnt = (unsigned)((xb - xa + 1) * (yb - ya + 1));
  dst[(size_t)blockIdx.z * frame_bytes + (size_t)y * w + x] = (uint8_t)(sum / cnt);
}

/* ------------------------------------------------------------------ 3x3 erode / dilate, strips */
/* ref grayskull.h:285-304: max over in-image taps == max with 0 fill; erode runs as
 * ~dilate(~x) (Strip<INVERT>), i.e. min with 255 fill. */
template <bool DILATE, int RG = 0>
__global__ __launch_bounds__(256) void k_morph16(uint8_t *dst, const uint8_t *src, unsigned w,
                                                 unsigned h, unsigned T, size_t frame_bytes) {
  const Strip<!DILATE, RG> S(src, dst, w, h, frame_bytes);
  if (S.wave_outside()) return; /* block wider than the frame */
  const int y0 = (int)(S.band * T);
  if (y0 >= (int)h) return;
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  auto hpass = [](const uint32_t(&U)[12], uint32_t(&H)[8]) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int j = k + 2;
      H[k] = pk_max_u16(pk_max_u16(alignbit(U[j], U[j - 1], 16), U[j]), alignbit(U[j + 1], U[j], 16));
    }
  };
  uint32_t ring[3][8];
  {
    uint32_t U[12];
    S.unpack(S.load(y0 - 1), U);
    hpass(U, ring[0]);
    S.unpack(S.load(y0), U);
    hpass(U, ring[1]);
  }
  strip_rows<3, !DILATE>(S, y0, nrows, 1, S.load(y0 + 1), [&](auto I, int, const uint32_t(&U)[12]) {
    constexpr int ia = decltype(I)::value, ib = (ia + 1) % 3, ic = (ia + 2) % 3;
    hpass(U, ring[ic]);
    uint32_t M[8];
#pragma unroll
    for (int k = 0; k < 8; k++) M[k] = pk_max_u16(pk_max_u16(ring[ia][k], ring[ib][k]), ring[ic][k]);
    return U4{pack_lohi(M[0], M[1]), pack_lohi(M[2], M[3]), pack_lohi(M[4], M[5]),
              pack_lohi(M[6], M[7])};
  });
}

/* diagnostic: same traffic pattern as the strip kernels, no arithmetic (access-pattern ceiling) */
template <int RG = 0, bool HALO = false> /* HALO: keep the halo dword load of the stencils alive (probe) */
__global__ __launch_bounds__(256) void k_strip_copy(uint8_t *dst, const uint8_t *src, unsigned w,
                                                    unsigned h, unsigned T, size_t frame_bytes) {
  const Strip<false, RG> S(src, dst, w, h, frame_bytes);
  if (S.wave_outside()) return; /* block wider than the frame */
  const int y0 = (int)(S.band * T);
  if (y0 >= (int)h) return;
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  RawRow raw = S.load(y0);
  for (int i = 0; i < nrows; i++) {
    const U4 cur = raw.v;
    if constexpr (HALO) (void)opaque(raw.hh);
    raw = S.load(y0 + i + 1);
    S.store(y0 + i, true, cur);
  }
}

/* ------------------------------------------------------------------ 3x3 int8 filter, strips */
/* ref :255-266 (gs_filter with a 3x3 kernel): zero-padded correlation, then `sum / norm` evaluated
 * in unsigned (a negative sum with norm > 1 becomes a huge quotient -> 255; with norm == 1 it stays
 * negative -> 0) and a clamp to 0..255.  Fast path for sum |k| <= 128 (every partial sum fits
 * int16: <= 128*255) and norm <= 256.  Per input row the three horizontal dot products H0/H1/H2
 * (one per kernel row) are formed with v_pk_mad_u16 (two's complement: the low 16 bits are the
 * signed result); out(y) = H0(y-1) + H1(y) + H2(y+1) is carried in two partial-sum sets.  The
 * quotient of the clamped non-negative sum is the top byte of sum * ceil(2^24 / norm)
 * (exact for sum <= 255*norm, norm <= 256: 255*norm*(norm-1) < 2^24), packed like k_blur16.  Both factors are below 2^24 for
 * norm >= 2, so the product is v_mul_u32_u24 (full rate; the 16 quarter-rate v_mul_lo_u32 of a row until round 4: 1-3 % of
 * the launch in a same-box A/B, profiles/r04x_filter_mul_u24.log); norm == 1 (NORM1: multiplier 2^24) is a shift. */
struct FilterK { uint32_t k[3][3]; uint32_t mul, cap, neg_is_255; }; /* k: coefficient in both halves */

GS_DEV void filter_hrow(const uint32_t (&U)[12], const uint32_t (&kr)[3], uint32_t (&H)[8]) {
#pragma unroll
  for (int p = 0; p < 8; p++) {
    const int j = p + 2; /* U[j] = own pair p */
    const uint32_t L = alignbit(U[j], U[j - 1], 16), R = alignbit(U[j + 1], U[j], 16);
    H[p] = pk_mad_u16(R, kr[2], pk_mad_u16(U[j], kr[1], pk_mul_u16(L, kr[0])));
  }
}

template <int RG = 0, bool NORM1 = false>
__global__ __launch_bounds__(256) void k_filter16(uint8_t *dst, const uint8_t *src, unsigned w, unsigned h,
                                                  unsigned T, size_t frame_bytes, FilterK fk) {
  const Strip<false, RG> S(src, dst, w, h, frame_bytes);
  if (S.wave_outside()) return; /* block wider than the frame */
  const int y0 = (int)(S.band * T);
  if (y0 >= (int)h) return;
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  uint32_t P0[8], P1[8]; /* P1 = H0(y-1) + H1(y) waits for H2(y+1); P0 = H0(y) waits for H1(y+1) */
  {
    uint32_t U[12], Ha[8], Hb[8];
    S.unpack(S.load(y0 - 1), U);
    filter_hrow(U, fk.k[0], Ha); /* H0(y0-1) */
    S.unpack(S.load(y0), U);
    filter_hrow(U, fk.k[1], Hb); /* H1(y0) */
#pragma unroll
    for (int p = 0; p < 8; p++) P1[p] = pk_add_u16(Ha[p], Hb[p]);
    filter_hrow(U, fk.k[0], P0); /* H0(y0) */
  }
  strip_rows<1, false>(S, y0, nrows, 1, S.load(y0 + 1), [&](auto, int, const uint32_t(&U)[12]) {
    uint32_t H0[8], H1[8], H2[8], od[4];
    filter_hrow(U, fk.k[0], H0), filter_hrow(U, fk.k[1], H1), filter_hrow(U, fk.k[2], H2);
#pragma unroll
    for (int g = 0; g < 4; g++) {
      uint32_t prod[4], negm[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int p = 2 * g + t;
        const uint32_t sum = pk_add_u16(P1[p], H2[p]); /* int16 pair */
        P1[p] = pk_add_u16(P0[p], H1[p]), P0[p] = H0[p];
        negm[t] = pk_sar_i16(sum, 15);                                    /* 0xffff where negative */
        const uint32_t c = pk_min_u16(pk_max_i16(sum, 0u), fk.cap);       /* 0 .. min(255*norm, 32767) */
        if constexpr (NORM1) prod[2 * t] = c << 24, prod[2 * t + 1] = (c >> 16) << 24; /* c <= 255 */
        else prod[2 * t] = mul_u24(c & 0xffffu, fk.mul), prod[2 * t + 1] = mul_u24(c >> 16, fk.mul); /* quotient = byte 3 */
      }
      const uint32_t q = perm_b32(prod[3], perm_b32(prod[2], perm_b32(prod[1], prod[0], 0x0c0c0703u), 0x0c070100u), 0x07020100u);
      od[g] = q | (pack_lohi(negm[0], negm[1]) & fk.neg_is_255);
    }
    return U4{od[0], od[1], od[2], od[3]};
  });
}

/* ------------------------------------------------------------------ generic per-pixel kernels */
/* grid: (ceil(w/64), ceil(h/4), n), block (64,4) */
__global__ __launch_bounds__(256) void k_sobel_px(uint8_t *dst, const uint8_t *src, unsigned w,
                                                  unsigned h, size_t frame_bytes) {
  const unsigned x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
  if (x < 1 || y < 1 || x + 1 >= w || y + 1 >= h) return;
  const uint8_t *a = src + (size_t)blockIdx.z * frame_bytes + (size_t)(y - 1) * w + x;
  const uint8_t *b = a + w, *c = b + w;
  int gx = ((int)a[1] - a[-1]) + 2 * ((int)b[1] - b[-1]) + ((int)c[1] - c[-1]);
  int gy = ((int)c[-1] + 2 * c[0] + c[1]) - ((int)a[-1] + 2 * a[0] + a[1]);
  int m = ((gx < 0 ? -gx : gx) + (gy < 0 ? -gy : gy)) / 2;
  dst[(size_t)blockIdx.z * frame_bytes + (size_t)y * w + x] = (uint8_t)(m > 255 ? 255 : m);
}

template <bool DILATE>
__global__ __launch_bounds__(256) void k_morph_px(uint8_t *dst, const uint8_t *src, unsigned w,
                                                  unsigned h, size_t frame_bytes) {
  const int x = (int)(blockIdx.x * 64u + threadIdx.x), y = (int)(blockIdx.y * 4u + threadIdx.y);
  if (x >= (int)w || y >= (int)h) return;
  const uint8_t *f = src + (size_t)blockIdx.z * frame_bytes;
  unsigned v = DILATE ? 0u : 255u;
  for (int yy = y - 1; yy <= y + 1; yy++)
    for (int xx = x - 1; xx <= x + 1; xx++) {
      if (yy < 0 || yy >= (int)h || xx < 0 || xx >= (int)w) continue;
      unsigned p = f[(size_t)yy * w + xx];
      v = DILATE ? (p > v ? p : v) : (p < v ? p : v);
    }
  dst[(size_t)blockIdx.z * frame_bytes + (size_t)y * w + x] = (uint8_t)v;
}

/* Clipped box sum from an inclusive integral image (same w x h layout as gs_integral):
 * MODE 0: dst = sum / count                     (gs_blur, any radius; ref :268-283)
 * MODE 1: dst = src > (int)(sum/count - c)      (gs_adaptive_threshold; ref :230-247)
 * u32 modular arithmetic matches the reference's `unsigned sum`. */
template <int MODE>
__global__ __launch_bounds__(256) void k_box_px(uint8_t *dst, const uint8_t *src,
                                                const unsigned *ii, unsigned w, unsigned h,
                                                unsigned radius, int c, size_t frame_px) {
  const int x = (int)(blockIdx.x * 64u + threadIdx.x), y = (int)(blockIdx.y * 4u + threadIdx.y);
  if (x >= (int)w || y >= (int)h) return;
  const unsigned *I = ii + (size_t)blockIdx.z * frame_px;
  const long r = (long)radius;
  const long xa = x - r < 0 ? 0 : x - r, xb = x + r > (long)w - 1 ? (long)w - 1 : x + r;
  const long ya = y - r < 0 ? 0 : y - r, yb = y + r > (long)h - 1 ? (long)h - 1 : y + r;
  unsigned sum = I[(size_t)yb * w + xb];
  if (xa > 0) sum -= I[(size_t)yb * w + (xa - 1)];
  if (ya > 0) sum -= I[(size_t)(ya - 1) * w + xb];
  if (xa > 0 && ya > 0) sum += I[(size_t)(ya - 1) * w + (xa - 1)];
  const unsigned cnt = (unsigned)((xb - xa + 1) * (yb - ya + 1));
  const size_t o = (size_t)blockIdx.z * frame_px + (size_t)y * w + x;
  if (MODE == 0) dst[o] = (uint8_t)(sum / cnt);
  else {
    const int thr = (int)(sum / cnt - (unsigned)c);
    dst[o] = ((int)src[o] > thr) ? 255 : 0;
  }
}

/* ref :255-266 -- zero-padded correlation with an int8 kernel, unsigned division quirk */
__global__ __launch_bounds__(256) void k_filter_px(uint8_t *dst, const uint8_t *src, unsigned w,
                                                   unsigned h, size_t frame_bytes,
                                                   const int8_t *kern, unsigned kw, unsigned kh,
                                                   unsigned norm) {
  const unsigned x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
  if (x >= w || y >= h) return;
  const uint8_t *f = src + (size_t)blockIdx.z * frame_bytes;
  int sum = 0;
  for (unsigned j = 0; j < kh; j++)
    for (unsigned i = 0; i < kw; i++) {
      const unsigned sx = x + i - kw / 2, sy = y + j - kh / 2; /* wraps => out of range => 0 */
      const int p = (sx < w && sy < h) ? f[(size_t)sy * w + sx] : 0;
      sum += p * (int)kern[j * kw + i];
    }
  sum = (int)((unsigned)sum / norm);
  dst[(size_t)blockIdx.z * frame_bytes + (size_t)y * w + x] =
      (uint8_t)(sum < 0 ? 0 : sum > 255 ? 255 : sum);
}

/* ref :189-197 -- 2x2 mean; dst is (sw/2) x (sh/2) */
__global__ __launch_bounds__(256) void k_downsample_px(uint8_t *dst, const uint8_t *src,
                                                       unsigned sw, unsigned sh) {
  const unsigned dw = sw / 2, dh = sh / 2;
  const unsigned x = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
  if (x >= dw || y >= dh) return;
  const uint8_t *f = src + (size_t)blockIdx.z * ((size_t)sw * sh) + (size_t)(2 * y) * sw + 2 * x;
  unsigned s = (unsigned)f[0] + f[1] + f[sw] + f[sw + 1];
  dst[(size_t)blockIdx.z * ((size_t)dw * dh) + (size_t)y * dw + x] = (uint8_t)(s / 4);
}

/* same, 8 output px per thread: two 16-byte loads (rows 2y and 2y+1), one 8-byte store.  The 2x2
 * sums are formed two at a time on u16 pairs.  Any sw >= 16 at any alignment (round 4): the group that would cross the
 * end of the output row is anchored at dw - 8 (it overlaps its neighbour, same bytes), so loads and stores stay whole
 * and inside their rows.  grid (ceil(ceil(dw/8)/64), ceil(dh/4), n), block (64,4) */
__global__ __launch_bounds__(256) void k_downsample8(uint8_t *dst, const uint8_t *src, unsigned sw,
                                                     unsigned sh) {
  const unsigned dw = sw / 2, dh = sh / 2;
  const unsigned gx = blockIdx.x * 64u + threadIdx.x, y = blockIdx.y * 4u + threadIdx.y;
  if (gx * 8u >= dw || y >= dh) return;
  const unsigned ox = gx * 8u + 8u > dw ? dw - 8u : gx * 8u;
  const uint8_t *f = src + (size_t)blockIdx.z * ((size_t)sw * sh) + (size_t)(2 * y) * sw + 2u * ox;
  const U4 a = load_u32x4_any(f), b = load_u32x4_any(f + sw);
  const uint32_t ra[4] = {a.x, a.y, a.z, a.w}, rb[4] = {b.x, b.y, b.z, b.w};
  uint32_t o[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    uint32_t out = 0;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const uint32_t da = ra[2 * q + t], db = rb[2 * q + t]; /* 4 src px of each row -> 2 outputs */
      /* vertical sums as u16 pairs: (p0+q0, p1+q1) and (p2+q2, p3+q3) */
      const uint32_t v01 = add2(unpack_lo(da), unpack_lo(db)), v23 = add2(unpack_hi(da), unpack_hi(db));
      const uint32_t s0 = (v01 & 0xffffu) + (v01 >> 16), s1 = (v23 & 0xffffu) + (v23 >> 16);
      out |= ((s0 >> 2) | ((s1 >> 2) << 8)) << (16 * t);
    }
    o[q] = out;
  }
  store_u32x2_any(dst + (size_t)blockIdx.z * ((size_t)dw * dh) + (size_t)y * dw + ox, o[0], o[1]);
}

}  // namespace gs
#endif

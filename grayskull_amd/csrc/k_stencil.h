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
#include <type_traits>

#include "prims.h"

namespace gs {

struct alignas(16) U4 { uint32_t x, y, z, w; };

template <int N, class F> GS_DEV void static_for(F &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

/* ------------------------------------------------------------------ strip helpers */
struct RawRow { U4 v; uint32_t hl, hr; };

/* Row y of a frame, this lane's 16 B plus (lanes 0 / 63 only) the 4 B left / right of the
 * wave's 1 KiB; anything outside the image reads as FILL bytes. */
template <uint32_t FILL4>
GS_DEV RawRow strip_load(const uint8_t *frame, unsigned w, unsigned h, int y, unsigned x0,
                         unsigned lane) {
  RawRow r;
  r.v = U4{FILL4, FILL4, FILL4, FILL4};
  r.hl = FILL4;
  r.hr = FILL4;
  if (y >= 0 && y < (int)h) { /* wave-uniform */
    const uint8_t *rp = frame + (size_t)y * w;
    if (x0 < w) {
      r.v = *(const U4 *)(rp + x0);
      if (lane == 0 && x0 > 0) r.hl = *(const uint32_t *)(rp + x0 - 4);
      if (lane == 63 && x0 + 16 < w) r.hr = *(const uint32_t *)(rp + x0 + 16);
    }
  }
  return r;
}

/* 24 bytes = cols x0-4 .. x0+19 as 12 dwords of u16 pairs: U[j] = (px 2j-4, px 2j-3). */
GS_DEV void strip_unpack(const RawRow &r, uint32_t (&U)[12]) {
  uint32_t L = wave_shr1(r.v.w, r.hl); /* left neighbour's last dword  */
  uint32_t R = wave_shl1(r.v.x, r.hr); /* right neighbour's first dword */
  U[0] = unpack_lo(L), U[1] = unpack_hi(L);
  U[2] = unpack_lo(r.v.x), U[3] = unpack_hi(r.v.x);
  U[4] = unpack_lo(r.v.y), U[5] = unpack_hi(r.v.y);
  U[6] = unpack_lo(r.v.z), U[7] = unpack_hi(r.v.z);
  U[8] = unpack_lo(r.v.w), U[9] = unpack_hi(r.v.w);
  U[10] = unpack_lo(R), U[11] = unpack_hi(R);
}

GS_DEV void store_bytes(uint8_t *p, const U4 &v, int lo, int hi) { /* bytes [lo,hi) of v */
  const uint32_t d[4] = {v.x, v.y, v.z, v.w};
  for (int i = lo; i < hi; i++) p[i] = (uint8_t)(d[i >> 2] >> (8 * (i & 3)));
}

/* ------------------------------------------------------------------ sobel, strips */
/* ref grayskull.h:306-320: interior only, (|gx|+|gy|)/2 clamped to 255. */
__global__ __launch_bounds__(256) void k_sobel16(uint8_t *dst, const uint8_t *src, unsigned w,
                                                 unsigned h, unsigned T, size_t frame_bytes) {
  const unsigned lane = threadIdx.x;
  const unsigned x0 = (blockIdx.x * 64u + lane) * 16u;
  const unsigned band = blockIdx.y * blockDim.y + threadIdx.y;
  const int y0 = 1 + (int)(band * T);
  if (y0 >= (int)h - 1) return; /* whole wave */
  const int nrows = ((int)h - 1 - y0) < (int)T ? ((int)h - 1 - y0) : (int)T;
  const uint8_t *sf = src + (size_t)blockIdx.z * frame_bytes;
  uint8_t *df = dst + (size_t)blockIdx.z * frame_bytes;

  uint32_t ring[3][12];
  strip_unpack(strip_load<0u>(sf, w, h, y0 - 1, x0, lane), ring[0]);
  strip_unpack(strip_load<0u>(sf, w, h, y0, x0, lane), ring[1]);
  RawRow nxt = strip_load<0u>(sf, w, h, y0 + 1, x0, lane);

  for (int base = 0; base < nrows; base += 3) {
    static_for<3>([&](auto I) {
      constexpr int ia = I, ib = (I + 1) % 3, ic = (I + 2) % 3;
      const int i = base + ia;
      if (i >= nrows) return; /* wave-uniform */
      const int y = y0 + i;
      strip_unpack(nxt, ring[ic]);
      if (i + 1 < nrows) nxt = strip_load<0u>(sf, w, h, y + 2, x0, lane);
      const uint32_t(&a)[12] = ring[ia];
      const uint32_t(&b)[12] = ring[ib];
      const uint32_t(&c)[12] = ring[ic];
      /* vertical pass on px -2..17 (U[1..10]): S = a+2b+c, D = c-a */
      uint32_t S[10], D[10];
#pragma unroll
      for (int j = 0; j < 10; j++) {
        S[j] = pk_add_u16(pk_add_u16(a[j + 1], c[j + 1]), pk_shl_u16(b[j + 1], 1));
        D[j] = pk_sub_u16(c[j + 1], a[j + 1]);
      }
      /* odd-aligned pairs: SA[j] = (S px 2j-3, 2j-2) relative to own px 0 */
      uint32_t SA[9], DA[9];
#pragma unroll
      for (int j = 0; j < 9; j++) {
        SA[j] = alignbit(S[j + 1], S[j], 16);
        DA[j] = alignbit(D[j + 1], D[j], 16);
      }
      uint32_t M[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { /* own pair k = px (2k, 2k+1) = S[k+1] */
        uint32_t gx = pk_sub_u16(SA[k + 1], SA[k]);
        uint32_t gy = pk_add_u16(pk_add_u16(DA[k], DA[k + 1]), pk_shl_u16(D[k + 1], 1));
        uint32_t m = pk_shr_u16(pk_add_u16(pk_abs_i16(gx), pk_abs_i16(gy)), 1);
        M[k] = pk_min_u16(m, 0x00ff00ffu);
      }
      U4 o{pack_lohi(M[0], M[1]), pack_lohi(M[2], M[3]), pack_lohi(M[4], M[5]),
           pack_lohi(M[6], M[7])};
      if (x0 < w) {
        uint8_t *op = df + (size_t)y * w + x0;
        const bool first = x0 == 0, last = x0 + 16 >= w;
        if (!first && !last) *(U4 *)op = o;
        else store_bytes(op, o, first ? 1 : 0, last ? 15 : 16); /* never touch x=0 / x=w-1 */
      }
    });
  }
}

/* ------------------------------------------------------------------ box blur, strips */
/* ref grayskull.h:268-283.  Zero fill outside the image makes the clipped window sum equal
 * the padded one; the divisor is (#cols in image) x (#rows in image).  Interior pixels use
 * floor(s/d) == (s*MAGIC)>>SHIFT (exact for s <= 255*d, checked at compile time in tests);
 * frame pixels take the generic u32 division. */
template <int R> struct BlurMagic;
template <> struct BlurMagic<1> { static constexpr uint32_t mul = 7282, shift = 16; };  /* /9  */
template <> struct BlurMagic<2> { static constexpr uint32_t mul = 5243, shift = 17; };  /* /25 */
template <> struct BlurMagic<3> { static constexpr uint32_t mul = 2675, shift = 17; };  /* /49 */

template <int R>
GS_DEV void blur_hsum(const uint32_t (&U)[12], uint32_t (&H)[8]) {
  uint32_t A[11]; /* A[j] = pair starting one px after U[j] */
#pragma unroll
  for (int j = 0; j < 11; j++) A[j] = alignbit(U[j + 1], U[j], 16);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int j = k + 2;
    uint32_t s = pk_add_u16(pk_add_u16(A[j - 1], U[j]), A[j]);
    if constexpr (R >= 2) s = pk_add_u16(pk_add_u16(s, U[j - 1]), U[j + 1]);
    if constexpr (R >= 3) s = pk_add_u16(pk_add_u16(s, A[j - 2]), A[j + 1]);
    H[k] = s;
  }
}

template <int R>
__global__ __launch_bounds__(256) void k_blur16(uint8_t *dst, const uint8_t *src, unsigned w,
                                                unsigned h, unsigned T, size_t frame_bytes) {
  constexpr int N = 2 * R + 1;
  const unsigned lane = threadIdx.x;
  const unsigned x0 = (blockIdx.x * 64u + lane) * 16u;
  const unsigned band = blockIdx.y * blockDim.y + threadIdx.y;
  const int y0 = (int)(band * T);
  if (y0 >= (int)h) return;
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  const uint8_t *sf = src + (size_t)blockIdx.z * frame_bytes;
  uint8_t *df = dst + (size_t)blockIdx.z * frame_bytes;
  const bool edge_lane = x0 == 0 || x0 + 16 >= w;

  uint32_t ring[N][8], V[8];
#pragma unroll
  for (int k = 0; k < 8; k++) V[k] = 0, ring[N - 1][k] = 0;
  {
    uint32_t U[12];
#pragma unroll
    for (int r = 0; r < N - 1; r++) { /* image rows y0-R .. y0+R-1 */
      strip_unpack(strip_load<0u>(sf, w, h, y0 - R + r, x0, lane), U);
      blur_hsum<R>(U, ring[r]);
#pragma unroll
      for (int k = 0; k < 8; k++) V[k] = pk_add_u16(V[k], ring[r][k]);
    }
  }
  RawRow nxt = strip_load<0u>(sf, w, h, y0 + R, x0, lane);

  for (int base = 0; base < nrows; base += N) {
    static_for<N>([&](auto I) {
      constexpr int slot = (I + N - 1) % N; /* holds row i-1 (outgoing), receives row i+2R */
      const int i = base + I;
      if (i >= nrows) return;
      const int y = y0 + i;
      uint32_t U[12], Hn[8];
      strip_unpack(nxt, U);
      if (i + 1 < nrows) nxt = strip_load<0u>(sf, w, h, y + R + 1, x0, lane);
      blur_hsum<R>(U, Hn);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        V[k] = pk_sub_u16(pk_add_u16(V[k], Hn[k]), ring[slot][k]);
        ring[slot][k] = Hn[k];
      }
      uint32_t q[16];
      const bool edge_row = y < R || y + R >= (int)h;
      if (!edge_row && !edge_lane) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          q[2 * k] = ((V[k] & 0xffffu) * BlurMagic<R>::mul) >> BlurMagic<R>::shift;
          q[2 * k + 1] = ((V[k] >> 16) * BlurMagic<R>::mul) >> BlurMagic<R>::shift;
        }
      } else {
        const int ya = y - R < 0 ? 0 : y - R, yb = y + R > (int)h - 1 ? (int)h - 1 : y + R;
        const unsigned cy = (unsigned)(yb - ya + 1);
#pragma unroll
        for (int p = 0; p < 16; p++) {
          const int x = (int)x0 + p;
          const int xa = x - R < 0 ? 0 : x - R, xb = x + R > (int)w - 1 ? (int)w - 1 : x + R;
          const unsigned cnt = cy * (unsigned)(xb - xa + 1);
          const unsigned s = (p & 1) ? (V[p >> 1] >> 16) : (V[p >> 1] & 0xffffu);
          q[p] = x < (int)w ? s / cnt : 0u;
        }
      }
      U4 o;
      o.x = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
      o.y = q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24);
      o.z = q[8] | (q[9] << 8) | (q[10] << 16) | (q[11] << 24);
      o.w = q[12] | (q[13] << 8) | (q[14] << 16) | (q[15] << 24);
      if (x0 < w) *(U4 *)(df + (size_t)y * w + x0) = o;
    });
  }
}

/* ------------------------------------------------------------------ 3x3 erode / dilate, strips */
/* ref grayskull.h:285-304: min/max over in-image taps == min/max with 255/0 fill. */
template <bool DILATE>
__global__ __launch_bounds__(256) void k_morph16(uint8_t *dst, const uint8_t *src, unsigned w,
                                                 unsigned h, unsigned T, size_t frame_bytes) {
  constexpr uint32_t FILL = DILATE ? 0u : 0xffffffffu;
  const unsigned lane = threadIdx.x;
  const unsigned x0 = (blockIdx.x * 64u + lane) * 16u;
  const unsigned band = blockIdx.y * blockDim.y + threadIdx.y;
  const int y0 = (int)(band * T);
  if (y0 >= (int)h) return;
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  const uint8_t *sf = src + (size_t)blockIdx.z * frame_bytes;
  uint8_t *df = dst + (size_t)blockIdx.z * frame_bytes;

  auto op = [](uint32_t a, uint32_t b) { return DILATE ? pk_max_u16(a, b) : pk_min_u16(a, b); };
  auto hpass = [&](const RawRow &rr, uint32_t(&H)[8]) {
    uint32_t U[12];
    strip_unpack(rr, U);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int j = k + 2;
      H[k] = op(op(alignbit(U[j], U[j - 1], 16), U[j]), alignbit(U[j + 1], U[j], 16));
    }
  };
  uint32_t ring[3][8];
  hpass(strip_load<FILL>(sf, w, h, y0 - 1, x0, lane), ring[0]);
  hpass(strip_load<FILL>(sf, w, h, y0, x0, lane), ring[1]);
  RawRow nxt = strip_load<FILL>(sf, w, h, y0 + 1, x0, lane);

  for (int base = 0; base < nrows; base += 3) {
    static_for<3>([&](auto I) {
      constexpr int ia = I, ib = (I + 1) % 3, ic = (I + 2) % 3;
      const int i = base + ia;
      if (i >= nrows) return;
      const int y = y0 + i;
      hpass(nxt, ring[ic]);
      if (i + 1 < nrows) nxt = strip_load<FILL>(sf, w, h, y + 2, x0, lane);
      uint32_t M[8];
#pragma unroll
      for (int k = 0; k < 8; k++) M[k] = op(op(ring[ia][k], ring[ib][k]), ring[ic][k]);
      U4 o{pack_lohi(M[0], M[1]), pack_lohi(M[2], M[3]), pack_lohi(M[4], M[5]),
           pack_lohi(M[6], M[7])};
      if (x0 < w) *(U4 *)(df + (size_t)y * w + x0) = o;
    });
  }
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

}  // namespace gs
#endif

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

template <int N, class F> GS_DEV void static_for(F &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

/* ------------------------------------------------------------------ strip helpers */
struct RawRow { U4 v; uint32_t hh; };

/* One lane's view of a frame pair.  The block shape is free: blockDim.x is a multiple of 64 (a
 * wave's lanes are 64 consecutive strips of one band), blockDim.y stacks further bands.
 * INVERT complements in-image bytes on the way in and all bytes on the way out
 * (erode == ~dilate(~x)): the hardware's zero fill then acts as the 255 fill erosion needs. */
template <bool INVERT = false> struct Strip {
  BufRsrc src, dst;
  unsigned w, h, x0, lane, band;
  GS_DEV Strip(const uint8_t *s, uint8_t *d, unsigned w_, unsigned h_, size_t frame_bytes)
      : src(make_buf(s + (size_t)blockIdx.z * frame_bytes, frame_bytes)),
        dst(make_buf(d + (size_t)blockIdx.z * frame_bytes, frame_bytes)), w(w_), h(h_) {
    lane = threadIdx.x & 63u;
    x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    band = uniform(blockIdx.y * blockDim.y + threadIdx.y); /* same for the wave's 64 lanes: SGPR */
  }
  /* row y: this lane's 16 B; lane 0 also fetches the 4 B left of the wave's 1 KiB, lane 63 the
   * 4 B right of it (one shared instruction).  Everything outside the image reads 0. */
  GS_DEV RawRow load(int y) const {
    const bool ok = (unsigned)y < h && x0 < w;
    const uint32_t base = (uint32_t)y * w + x0;
    RawRow r;
    r.v = buf_load16(src, ok ? base : kOOB);
    uint32_t ho = kOOB;
    if (lane == 0 && x0 > 0) ho = base - 4;
    if (lane == 63 && x0 + 16 < w) ho = base + 16;
    r.hh = buf_load4(src, ok ? ho : kOOB);
    if (INVERT) { /* complement in-image bytes only: out-of-range stays 0 in the inverted domain */
      const uint32_t m = ok ? 0xffffffffu : 0u, hm = (ok && ho != kOOB) ? 0xffffffffu : 0u;
      r.v = U4{r.v.x ^ m, r.v.y ^ m, r.v.z ^ m, r.v.w ^ m};
      r.hh ^= hm;
    }
    return r;
  }
  /* whole 16 B of row y (dropped when !ok or the lane is outside the image) */
  GS_DEV void store(int y, bool ok, U4 o) const {
    if (INVERT) o = U4{~o.x, ~o.y, ~o.z, ~o.w};
    buf_store16(dst, (ok && x0 < w) ? (uint32_t)y * w + x0 : kOOB, o);
  }
};

/* 24 bytes = cols x0-4 .. x0+19 as 12 dwords of u16 pairs: U[j] = (px 2j-4, px 2j-3). */
GS_DEV void strip_unpack(const RawRow &r, uint32_t (&U)[12]) {
  const uint32_t L = wave_shr1(r.v.w, r.hh); /* left neighbour's last dword (lane 0: halo)   */
  const uint32_t R = wave_shl1(r.v.x, r.hh); /* right neighbour's first dword (lane 63: halo) */
  U[0] = unpack_lo(L), U[1] = unpack_hi(L);
  U[2] = unpack_lo(r.v.x), U[3] = unpack_hi(r.v.x);
  U[4] = unpack_lo(r.v.y), U[5] = unpack_hi(r.v.y);
  U[6] = unpack_lo(r.v.z), U[7] = unpack_hi(r.v.z);
  U[8] = unpack_lo(r.v.w), U[9] = unpack_hi(r.v.w);
  U[10] = unpack_lo(R), U[11] = unpack_hi(R);
}

/* Row loop shared by the strip kernels.  Per output row i of the band:
 *     wait for row i's raw data -> unpack (raw registers die) -> store row i-1's result ->
 *     issue the load of the next input row -> arithmetic for row i.
 * So the store and the next load are in flight during the arithmetic and the single
 * s_waitcnt at the top of the next row finds them (nearly) complete.  RING rows are unrolled
 * so the vertical window is indexed at compile time. */
struct NoFin {
  GS_DEV U4 operator()(const U4 &o, int) const { return o; } /* called right before row y's store */
  GS_DEV void prefetch(int) {}                              /* called when row y's loads issue   */
};
/* EXITS: leave the unrolled group at the first row >= nrows (true), or always run whole groups of
 * RING rows (false): then the group is one basic block -- register rotation resolves at compile
 * time with nothing to copy at block boundaries -- and rows i >= nrows of the last group are
 * computed and dropped (stores predicated off; their loads are in-frame rows of the next band or
 * out-of-range zero fill).  false costs registers; it pays for the VALU-heavy fused kernel only.
 * DEPTH: how many rows ahead the input is requested (1 or 2). */
#ifndef GS_FENCE
#define GS_FENCE 2
#endif
template <int RING, bool INVERT, bool EXITS = true, int DEPTH = 1, class Body, class Fin = NoFin>
GS_DEV void strip_rows(const Strip<INVERT> &S, int y0, int nrows, int lead, RawRow first, Body &&body,
                       Fin fin = Fin()) {
  RawRow raw = first; /* = load(y0 + lead): the newest input row output row 0 needs */
  RawRow raw2 = first;
  if constexpr (DEPTH == 2) raw2 = S.load(y0 + lead + 1);
  U4 o_prev{0, 0, 0, 0};
  fin.prefetch(y0);
  int base = 0;
  for (; base < nrows; base += RING) {
    static_for<RING>([&](auto I) {
      const int i = base + decltype(I)::value;
      if constexpr (EXITS) {
        if (i >= nrows) return; /* wave-uniform */
      }
      if constexpr (!EXITS && GS_FENCE >= 2) sched_fence(); /* the next row's unpack (= its vmcnt wait) stays down here */
      uint32_t U[12];
      strip_unpack(raw, U);
      S.store(y0 + i - 1, i > 0 && i <= nrows, fin(o_prev, y0 + i - 1));
      if constexpr (DEPTH == 2) {
        raw = raw2;
        raw2 = S.load(y0 + i + lead + 2);
      } else {
        raw = S.load(y0 + i + lead + 1);
      }
      fin.prefetch(y0 + i);
      if constexpr (!EXITS && GS_FENCE >= 1) sched_fence(); /* one big block: keep the loads ahead of the arithmetic */
      o_prev = body(I, i, U);
    });
  }
  if constexpr (EXITS) S.store(y0 + nrows - 1, nrows > 0, fin(o_prev, y0 + nrows - 1));
  else S.store(y0 + base - 1, base == nrows && nrows > 0, fin(o_prev, y0 + base - 1));
}

/* ------------------------------------------------------------------ sobel, strips */
/* ref grayskull.h:306-320: (|gx|+|gy|)/2 clamped to 255 on rows 1..h-2.
 * Horizontal pass once per input row:
 *   H1[x] = r[x-1] + 2 r[x] + r[x+1]      H2[x] = r[x+1] - r[x-1]
 * vertical pass per output row:  gx = H2a + 2 H2b + H2c,  gy = H1c - H1a  (SobelState).
 * The reference never writes columns 0 and w-1 (ref :309).  The kernel stores whole 16-byte
 * groups, so with KEEP_COLS the lane holding column 0 (w-1) fetches dst's own first (last) dword
 * of the row one iteration ahead and writes that byte back unchanged.  KEEP_COLS=false is for
 * callers that do not care (interior-only copy back, or frame zeroed afterwards). */
GS_DEV void sobel_hpass(const uint32_t (&U)[12], uint32_t (&H1)[8], uint32_t (&H2)[8]) {
  uint32_t A[11];
#pragma unroll
  for (int j = 1; j <= 9; j++) A[j] = alignbit(U[j + 1], U[j], 16); /* (px 2j-3, 2j-2) */
#pragma unroll
  for (int k = 0; k < 8; k++) {
    H2[k] = pk_sub_u16(A[k + 2], A[k + 1]);
    H1[k] = pk_mad2_u16(U[k + 2], pk_add_u16(A[k + 1], A[k + 2]));
  }
}

struct SobelKeepCols { /* Fin functor of strip_rows */
  BufRsrc dst;
  unsigned w, x0;
  bool first, last;
  uint32_t e_next = 0, e_cur = 0; /* dst dword holding the protected byte: rows y+1 and y */
  GS_DEV SobelKeepCols(const Strip<> &S) : dst(S.dst), w(S.w), x0(S.x0) {
    first = x0 == 0, last = x0 + 16 == w; /* launcher guarantees w >= 32: never both */
  }
  GS_DEV void prefetch(int y) {
    e_cur = e_next;
    const uint32_t row = (uint32_t)y * w + x0;
    e_next = buf_load4(dst, first ? row : last ? row + 12 : kOOB);
  }
  GS_DEV U4 operator()(U4 o, int) const {
    /* operator() for row y runs after prefetch(y+1): row y's dword is e_cur */
    o.x = first ? perm_b32(o.x, e_cur, 0x07060500u) : o.x; /* byte 0 <- dst */
    o.w = last ? perm_b32(o.w, e_cur, 0x03060504u) : o.w;  /* byte 3 <- dst */
    return o;
  }
};

/* vertical state of the sobel recurrence, one new input row b per step (output row y = b-1):
 *   gx(y) = H2(b-2) + 2 H2(b-1) + H2(b) = Pa + H2(b),   Pa' = H2(b-1) + 2 H2(b),
 *   gy(y) = H1(b) - H1(b-2).
 * 32 registers instead of a 3-row ring of (H1,H2) = 48, and only a 2-step static rotation. */
struct SobelState {
  uint32_t Pa[8], Hp[8], H1[2][8];
  /* prime with input rows y0-1 (r0) and y0 (r1) */
  GS_DEV void init(const uint32_t (&U0)[12], const uint32_t (&U1)[12]) {
    uint32_t h1[8], h2[8];
    sobel_hpass(U0, H1[0], h2);
    sobel_hpass(U1, h1, Hp);
#pragma unroll
    for (int k = 0; k < 8; k++) H1[1][k] = h1[k], Pa[k] = pk_mad2_u16(Hp[k], h2[k]);
  }
  /* PAR = parity of the step: H1[PAR] holds row b-2 and receives row b */
  template <int PAR> GS_DEV U4 step(const uint32_t (&U)[12]) {
    uint32_t M[8];
    return step<PAR>(U, M);
  }
  /* M: the 16 results as u16 pairs (before packing to bytes) */
  template <int PAR> GS_DEV U4 step(const uint32_t (&U)[12], uint32_t (&M)[8]) {
    uint32_t H1n[8], H2n[8];
    sobel_hpass(U, H1n, H2n);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t gx = pk_add_u16(Pa[k], H2n[k]);
      const uint32_t gy = pk_sub_u16(H1n[k], H1[PAR][k]);
      Pa[k] = pk_mad2_u16(H2n[k], Hp[k]);
      Hp[k] = H2n[k];
      H1[PAR][k] = H1n[k];
      const uint32_t m = pk_shr_u16(pk_add_u16(pk_abs_i16(gx), pk_abs_i16(gy)), 1);
      M[k] = pk_min_u16(m, 0x00ff00ffu);
    }
    return U4{pack_lohi(M[0], M[1]), pack_lohi(M[2], M[3]), pack_lohi(M[4], M[5]),
              pack_lohi(M[6], M[7])};
  }
  /* same, H1 history shifted instead of alternated (for callers whose unroll period is odd) */
  GS_DEV U4 step_shift(const uint32_t (&U)[12], uint32_t (&M)[8]) {
    const U4 o = step<0>(U, M); /* H1[0] (row b-2) consumed and overwritten with row b */
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t t = H1[0][k];
      H1[0][k] = H1[1][k], H1[1][k] = t;
    }
    return o;
  }
};

#ifndef GS_SOBEL_MINWAVES
#define GS_SOBEL_MINWAVES 1
#endif
#ifndef GS_BLUR_MINWAVES
#define GS_BLUR_MINWAVES 1
#endif
template <bool KEEP_COLS>
__global__ __launch_bounds__(256, GS_SOBEL_MINWAVES) void k_sobel16(uint8_t *dst, const uint8_t *src, unsigned w,
                                                 unsigned h, unsigned T, size_t frame_bytes) {
  const Strip<> S(src, dst, w, h, frame_bytes);
  const int y0 = 1 + (int)(S.band * T);
  if (y0 >= (int)h - 1) return; /* whole wave */
  const int nrows = ((int)h - 1 - y0) < (int)T ? ((int)h - 1 - y0) : (int)T;
  SobelState st;
  {
    uint32_t U0[12], U1[12];
    strip_unpack(S.load(y0 - 1), U0);
    strip_unpack(S.load(y0), U1);
    st.init(U0, U1);
  }
  auto body = [&](auto I, int, const uint32_t(&U)[12]) { return st.template step<decltype(I)::value>(U); };
  if constexpr (KEEP_COLS) strip_rows<2, false>(S, y0, nrows, 1, S.load(y0 + 1), body, SobelKeepCols(S));
  else strip_rows<2, false>(S, y0, nrows, 1, S.load(y0 + 1), body);
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

/* ------------------------------------------------------------------ box blur, strips */
/* ref grayskull.h:268-283.  Zero fill outside the image makes the clipped window sum equal the
 * padded one.  The strip kernel divides every pixel by the interior divisor d = (2R+1)^2 with
 * floor(s/d) == (s*MUL) >> 24 (exact for s <= 255*d; the quotient is the top byte of the 32-bit
 * product, so four of them pack with v_perm_b32).  Where the window is clipped the divisor is
 * the number of in-image taps (ref :275-281): the R leftmost / rightmost columns get their own
 * per-lane multipliers ceil(2^24 / (N * cols_in_image)) (selects, no branch; exact: for d <= N*N,
 * e = MUL*d - 2^24 < d and s*e <= 255*d*d < 2^24), and the R top / bottom rows -- 2R rows per
 * frame -- are rewritten by k_blur_edge_rows with a true division. */
template <int R> struct BlurMagic;
template <> struct BlurMagic<1> { static constexpr uint32_t mul = 1864136; };  /* ceil(2^24/9)  */
template <> struct BlurMagic<2> { static constexpr uint32_t mul = 671089; };   /* ceil(2^24/25) */
template <> struct BlurMagic<3> { static constexpr uint32_t mul = 342393; };   /* ceil(2^24/49) */

template <int R>
GS_DEV void blur_hsum(const uint32_t (&U)[12], uint32_t (&H)[8]) {
  uint32_t A[11], P[11];
  constexpr int jlo = R >= 3 ? 0 : 1, jhi = R >= 3 ? 10 : 9;
#pragma unroll
  for (int j = jlo; j <= jhi; j++) {
    A[j] = alignbit(U[j + 1], U[j], 16); /* pair starting one px after U[j] */
    P[j] = pk_add_u16(U[j], A[j]);       /* 2-px sums (x, x+1) for both halves */
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int j = k + 2; /* U[j] = own pair k */
    if constexpr (R == 1) H[k] = pk_add_u16(A[j - 1], P[j]);
    else if constexpr (R == 2) H[k] = pk_add_u16(pk_add_u16(P[j - 1], P[j]), U[j + 1]);
    else H[k] = pk_add_u16(pk_add_u16(pk_add_u16(A[j - 2], P[j - 1]), P[j]), P[j + 1]);
  }
}

template <int R>
__global__ __launch_bounds__(256, GS_BLUR_MINWAVES) void k_blur16(uint8_t *dst, const uint8_t *src, unsigned w,
                                                unsigned h, unsigned T, size_t frame_bytes) {
  constexpr int N = 2 * R + 1;
  const Strip<> S(src, dst, w, h, frame_bytes);
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
    strip_unpack(S.load(y0 - R + r), U);
    blur_hsum<R>(U, ring[r]);
#pragma unroll
    for (int k = 0; k < 8; k++) V[k] = pk_add_u16(V[k], ring[r][k]);
  }
  strip_rows<N, false>(S, y0, nrows, R, S.load(y0 + R), [&](auto I, int, const uint32_t(&U)[12]) {
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
    }
    uint32_t od[4];
#pragma unroll
    for (int g = 0; g < 4; g++) { /* 4 pixels = 2 pairs -> one output dword */
      uint32_t p[4];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int k = 2 * g + t;
        V[k] = pk_sub_u16(pk_add_u16(V[k], Hn[k]), ring[slot][k]);
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

/* ------------------------------------------------------------------ fused blur -> sobel -> histogram */
/* The config-2 chain (gs_blur(R); gs_sobel; histogram for gs_otsu_threshold) in ONE pass over the
 * frame: 1 B/px read + 1 B/px written instead of 2+2+1.  Per source row the lane forms the
 * (2R+1)-tap horizontal sums for pixels -2..17, keeps their running vertical sum, divides (exact
 * 2^24 multipliers, chosen per pixel / per row where the window is clipped: every divisor is
 * rows_in_image * cols_in_image, ref :275-281), and feeds the blurred row -- never written to
 * memory -- straight into the sobel recurrence.  The sobel bytes go to dst and into an
 * LDS-privatised histogram (32 bank-spread copies, see k_hist_partial); each block leaves 256
 * partial counts for k_hist_reduce.  Bit-identical to the separate calls (tests).  Columns 0 and
 * w-1 of dst receive junk here; the launcher zeroes the 1-px frame afterwards (config 2 runs
 * gs_sobel into a zeroed image), and the histogram counts those frame pixels as 0 analytically. */
constexpr uint32_t blur_k24(unsigned cx, unsigned cy) { return (0x1000000u + cx * cy - 1u) / (cx * cy); }

template <int R, unsigned CX>
GS_DEV uint32_t blur_mul_for_rows(unsigned cy) { /* cy in [R+1, 2R+1], wave-uniform */
  uint32_t m = blur_k24(CX, 2 * R + 1);
#pragma unroll
  for (unsigned c = R + 1; c < 2 * R + 1; c++) m = cy == c ? blur_k24(CX, c) : m;
  return m;
}

template <int R>
GS_DEV void blur_hsum10(const uint32_t (&U)[12], uint32_t (&H)[10]) { /* pairs = px -2..17 */
  uint32_t A[13]; /* A[j+1] = pair starting one px after U[j], j = -1..11 (ends zero-extended) */
  if constexpr (R >= 3) A[0] = alignbit(U[0], 0u, 16), A[12] = alignbit(0u, U[11], 16);
#pragma unroll
  for (int j = 0; j <= 10; j++) A[j + 1] = alignbit(U[j + 1], U[j], 16);
  if constexpr (R == 1) {
#pragma unroll
    for (int k = 0; k < 10; k++) H[k] = pk_add_u16(pk_add_u16(A[k + 1], U[k + 1]), A[k + 2]);
  } else {
    uint32_t Q[11]; /* Q[j] = U[j] + A[j+1]: the 2-px sums starting at both pixels of pair j */
#pragma unroll
    for (int j = 0; j <= 10; j++) Q[j] = pk_add_u16(U[j], A[j + 1]);
#pragma unroll
    for (int k = 0; k < 10; k++) {
      const int j = k + 1; /* U[j] = pair k */
      uint32_t t = pk_add_u16(pk_add_u16(Q[j - 1], Q[j]), U[j + 1]);        /* -2 .. +2 */
      if constexpr (R >= 3) t = pk_add_u16(pk_add_u16(t, A[j - 1]), A[j + 2]); /* -3, +3 */
      H[k] = t;
    }
  }
}

/* grid like the strip kernels; partial: [frame][blockIdx.y * gridDim.x + blockIdx.x][256] */
template <int R>
__global__ __launch_bounds__(256) void k_blur_sobel_hist16(uint8_t *dst, const uint8_t *src,
                                                           unsigned w, unsigned h, unsigned T,
                                                           size_t frame_bytes, unsigned *partial) {
  constexpr int N = 2 * R + 1;
  __shared__ unsigned lh[256 * 32];
  const unsigned tid = threadIdx.y * blockDim.x + threadIdx.x, copy = tid & 31u;
  for (unsigned i = tid; i < 256 * 32; i += 256) lh[i] = 0;
  __syncthreads();
  const Strip<> S(src, dst, w, h, frame_bytes);
  const int y0 = 1 + (int)(S.band * T);
  if (y0 < (int)h - 1) { /* wave-uniform; no early return: every wave reaches the barrier */
    const int nrows = ((int)h - 1 - y0) < (int)T ? ((int)h - 1 - y0) : (int)T;
    const bool first = S.x0 == 0, last = S.x0 + 16 == w, inimg = S.x0 < w;
    /* N+1 ring slots: the new row lands in the free slot and the unroll period N+1 is even, so
     * the sobel history alternates (step<parity>) without register moves */
    constexpr bool SPARE = R <= 2; /* R = 3: the 8th slot would cost the third wave per SIMD */
    constexpr int NS = SPARE ? N + 1 : N, P0 = SPARE ? 1 : 0; /* P0: slot of the first prologue row */
    uint32_t ring[NS][10], V[10];
    SobelState st;
    /* blurred row b as u16 pairs for pixels -2..17, placed where sobel_hpass expects U[1..10] */
    auto blurred = [&](int b, uint32_t(&UB)[12]) {
      const int ya = b - R < 0 ? 0 : b - R, yb = b + R > (int)h - 1 ? (int)h - 1 : b + R;
      const unsigned cy = (unsigned)(yb - ya + 1);
      const uint32_t mC = blur_mul_for_rows<R, N>(cy);
      uint32_t mL[R], mR[R];
      static_for<R>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        mL[q] = first ? blur_mul_for_rows<R, R + 1 + q>(cy) : mC;
        mR[q] = last ? blur_mul_for_rows<R, 2 * R - q>(cy) : mC;
      });
      UB[0] = 0, UB[11] = 0;
#pragma unroll
      for (int k = 0; k < 10; k++) { /* pair k = own pixels (2k-2, 2k-1) */
        uint32_t pr[2];
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
          const int q = 2 * k - 2 + hlf; /* own pixel index -2..17 */
          const uint32_t sv = hlf ? (V[k] >> 16) : (V[k] & 0xffffu);
          const uint32_t m = (q >= 0 && q < R) ? mL[(q >= 0 && q < R) ? q : 0]
                             : (q >= 16 - R && q < 16) ? mR[(q >= 16 - R && q < 16) ? q - (16 - R) : 0]
                                                       : mC;
          pr[hlf] = sv * m; /* quotient = byte 3 */
        }
        UB[k + 1] = perm_b32(pr[1], pr[0], 0x0c070c03u);
      }
    };
    /* prologue: source rows y0-1-R .. y0+R give blurred rows y0-1 and y0 */
#pragma unroll
    for (int k = 0; k < 10; k++) V[k] = 0;
    uint32_t UB0[12], UB1[12];
    static_for<N>([&](auto K) {
      constexpr int kk = decltype(K)::value;
      uint32_t U[12];
      strip_unpack(S.load(y0 - 1 - R + kk), U);
      blur_hsum10<R>(U, ring[kk + P0]);
#pragma unroll
      for (int k = 0; k < 10; k++) V[k] = pk_add_u16(V[k], ring[kk + P0][k]);
    });
    blurred(y0 - 1, UB0);
    {
      uint32_t U[12], Hn[10];
      strip_unpack(S.load(y0 + R), U);
      blur_hsum10<R>(U, Hn); /* enters slot 0 (SPARE: the free slot); the oldest row (slot P0) leaves */
#pragma unroll
      for (int k = 0; k < 10; k++) V[k] = pk_sub_u16(pk_add_u16(V[k], Hn[k]), ring[P0][k]), ring[0][k] = Hn[k];
    }
    blurred(y0, UB1);
    st.init(UB0, UB1);
    const uint32_t cb2 = (copy << 2) * 0x10001u; /* this lane's histogram copy, as a pair of byte offsets */

#ifndef GS_FUSED_DEPTH
#define GS_FUSED_DEPTH 1
#endif
#ifndef GS_FUSED_EXITS
#define GS_FUSED_EXITS false
#endif
    strip_rows<NS, false, GS_FUSED_EXITS, GS_FUSED_DEPTH>(S, y0, nrows, R + 1, S.load(y0 + R + 1), [&](auto I, int i, const uint32_t(&U)[12]) {
      /* iteration I, SPARE: slot I+1 is free (its row left last iteration), slot I+2 holds the
       * oldest row; otherwise the new row replaces the oldest (slot I+1) */
      constexpr int fr = (decltype(I)::value + 1) % NS, old = (decltype(I)::value + 1 + P0) % NS;
      uint32_t UB[12], M[8], Hn[10];
      blur_hsum10<R>(U, Hn);
#pragma unroll
      for (int k = 0; k < 10; k++) V[k] = pk_sub_u16(pk_add_u16(V[k], Hn[k]), ring[old][k]), ring[fr][k] = Hn[k];
      blurred(y0 + i + 1, UB);
      U4 o;
      if constexpr (NS % 2 == 0) o = st.template step<decltype(I)::value & 1>(UB, M);
      else o = st.step_shift(UB, M);
      /* histogram, branch-free: lanes outside the image, the two frame columns and the dropped
       * rows of the last group add 0.  LDS byte offsets bin*128 + copy*4 for both pixels of a
       * pair come from one v_pk_mad_u16. */
      const unsigned inc = (inimg && i < nrows) ? 1u : 0u;
      const unsigned inc0 = first ? 0u : inc, inc15 = last ? 0u : inc;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t a2 = pk_mad_u16_s(M[k], 0x00800080u, cb2);
        atomicAdd((unsigned *)((char *)lh + (a2 & 0xffffu)), k == 0 ? inc0 : inc);
        atomicAdd((unsigned *)((char *)lh + (a2 >> 16)), k == 7 ? inc15 : inc);
      }
      return o;
    });
  }
  __syncthreads();
  unsigned acc = 0;
#pragma unroll 8
  for (unsigned k = 0; k < 32; k++) acc += lh[tid * 32u + ((k + tid) & 31u)];
  const size_t blk = (size_t)blockIdx.z * gridDim.x * gridDim.y + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  partial[blk * 256u + tid] = acc;
}

/* ------------------------------------------------------------------ 3x3 erode / dilate, strips */
/* ref grayskull.h:285-304: max over in-image taps == max with 0 fill; erode runs as
 * ~dilate(~x) (Strip<INVERT>), i.e. min with 255 fill. */
template <bool DILATE>
__global__ __launch_bounds__(256) void k_morph16(uint8_t *dst, const uint8_t *src, unsigned w,
                                                 unsigned h, unsigned T, size_t frame_bytes) {
  const Strip<!DILATE> S(src, dst, w, h, frame_bytes);
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
    strip_unpack(S.load(y0 - 1), U);
    hpass(U, ring[0]);
    strip_unpack(S.load(y0), U);
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
__global__ __launch_bounds__(256) void k_strip_copy(uint8_t *dst, const uint8_t *src, unsigned w,
                                                    unsigned h, unsigned T, size_t frame_bytes) {
  const Strip<> S(src, dst, w, h, frame_bytes);
  const int y0 = (int)(S.band * T);
  if (y0 >= (int)h) return;
  const int nrows = ((int)h - y0) < (int)T ? ((int)h - y0) : (int)T;
  RawRow raw = S.load(y0);
  for (int i = 0; i < nrows; i++) {
    const U4 cur = raw.v;
    raw = S.load(y0 + i + 1);
    S.store(y0 + i, true, cur);
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

/*
 * k_strip.h -- the row-strip machinery shared by the stencil kernels (k_stencil.h) and the fused
 * pipeline kernel (k_fused.h): lane = 16 consecutive pixels of a row, wave = 1024-px column
 * block walking down a band of rows, vertical window in registers.  Device helpers only, no
 * kernels (so both translation units can include it).  See k_stencil.h for the design notes.
 */
#ifndef GS_K_STRIP_H
#define GS_K_STRIP_H
#include <type_traits>

#include "prims.h"

#ifndef GS_SOBEL_SAT
#define GS_SOBEL_SAT 1 /* 1: saturating-mad clamp in the per-call sobel kernels (A/B switch) */
#endif
#ifndef GS_BLUR_NOEXIT
#define GS_BLUR_NOEXIT 0
#endif
#ifndef GS_STRIP_PF
#define GS_STRIP_PF 1 /* input rows requested ahead of the arithmetic (experiment hook: 2) */
#endif
#ifndef GS_SOBEL_NOEXIT
#define GS_SOBEL_NOEXIT 1 /* 1: k_sobel16 runs whole unroll groups (one basic block, no phi copies) */
#endif

namespace gs {

template <int N, class F> GS_DEV void static_for(F &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

/* ------------------------------------------------------------------ strip helpers */

/* One lane's view of a frame pair.  The block shape is free: blockDim.x is a multiple of 64 (a
 * wave's lanes are 64 consecutive strips of one band), blockDim.y stacks further bands.
 * INVERT complements in-image bytes on the way in and all bytes on the way out
 * (erode == ~dilate(~x)): the hardware's zero fill then acts as the 255 fill erosion needs. */
/* Addressing: a row's byte offset is wave-uniform (SALU), the lane's column offset is fixed for the
 * whole kernel, so every buffer offset is ONE full-rate v_add_u32 of the two.  Out-of-image rows
 * and lanes are encoded in the summands themselves: kRowOOB + any valid column and any valid row +
 * kOOB both land beyond the frame (< 2^31 - 1 bytes, see strip_ok), and kRowOOB + kOOB = 2^32 - 1. */
constexpr uint32_t kRowOOB = 0x7fffffffu;
/* XCD-aware band mapping, requested by the launcher through the top bit of frame_bytes (gridDim.x == 1,
 * blockDim.y == 1, gridDim.y a multiple of 8).  The dispatcher places block b on XCD b % 8; with bands in
 * dispatch order the two halo rows a band shares with its neighbour are fetched by ANOTHER XCD, i.e. past the
 * L2 (k_sobel16 read 1.22x its algorithmic bytes, k_blur16<2> 1.31x: profiles/pmc_traffic.json round 2).
 * Mapped, XCD k walks the k-th eighth of every frame top to bottom, so a band's halo rows are the rows its
 * own L2 fetched for the previous band. */
constexpr size_t kStripXcdFlag = (size_t)1 << 63;
/* RAGGED (w % 16 != 0, w >= 32; the API has no stride field, ref grayskull.h:14-17, so rows of such a frame start at
 * every 16-byte phase): strips 0 .. t-1 (t = (w-1)/16) lie on the 16-px grid as usual and the LAST strip is anchored at
 * x0 = w - 16 -- it overlaps its left neighbour by 16 - w % 16 pixels, which both lanes compute (same inputs, same
 * bytes) and both store.  So every load and store is a whole 16 bytes inside the row: no partial store, no
 * read-modify-write of the next row's head, nothing read past the frame.  The two lanes at the seam cannot take their
 * horizontal neighbours from the adjacent lane (the tail lane's registers are shifted): the tail lane's left dword
 * (px w-20 .. w-17) and its neighbour's right dword (px 16t .. 16t+3) ride in the per-lane halo load that lane 0 /
 * lane 63 of every wave use anyway, selected behind the DPP move with one v_cndmask each (unpack).  A right-halo dword
 * that would cross the row end is fetched early and shifted down: pixels >= w read 0 like everything outside the image.
 * One lane has one halo slot: were the seam's left lane also lane 0 of a wave that needs its own left halo (t % 64 == 1),
 * the whole row is moved up by one lane (`shift`; lane 0 of the first wave idles).  Rows are w bytes apart whatever
 * w is, and so are these accesses: gfx950 serves buffer_load/store_dwordx4 at any byte address. */
__host__ __device__ inline unsigned strip_ragged_shift(unsigned w) { /* also the launcher's: one more strip to place */
  const unsigned t = (w - 1u) >> 4;
  return ((w & 15u) != 0u && (t & 63u) == 1u && t > 1u) ? 1u : 0u;
}
/* REALIGN (RG = 2; rows whose byte phase is not a multiple of 4: w % 4 != 0 or a frame at such an address).  Measured
 * on MI355X (profiles/r04b_byte_phase_cost.log): 16-byte loads at a dword-aligned address cost the stencils 3 % whatever
 * the 16-byte phase, at any other address 30-45 % (the plain copy loses nothing, a second row in flight changes nothing,
 * stores lose 5 %).  So this mode loads every lane's 16 bytes from the dword-aligned address BELOW its pixels and moves
 * the bytes into place in registers: the lane's 24-byte neighbourhood starts p = address & 3 bytes into the seven
 * dwords [left, own x 4, right, right'], one v_alignbyte_b32 per output dword.  The halo load is two dwords (lane 63
 * needs both right-hand ones).  The tail strip stays anchored at w - 16 as in RG = 1 -- its own phase, its left dword
 * from its halo slot -- and the lane behind it loads the 16 bytes that follow, so that the tail lane finds its
 * right-hand dwords (the one with the row's last p pixels) where every lane does; the last lane of a row of whole strips
 * fetches them itself, its halo slot is free.  Pixels left of column 0 / right of column w - 1 that come along are
 * masked.  Stores go out at the byte address as they are. */
__host__ __device__ inline unsigned strip_realign_shift(unsigned w) {
  const unsigned t = (w - 1u) >> 4, l = t & 63u;
  return ((w & 15u) != 0u && (l == 1u || l == 63u) && t > 1u) ? 1u : 0u; /* tail not in lane 63 (helper lane), its left neighbour not in lane 0 */
}
__host__ __device__ inline unsigned strip_realign_helper(unsigned w) { /* one more lane loads the 16 bytes behind the row's last strip */
  return ((w & 15u) != 0u || (((w - 1u) >> 4) & 63u) == 0u) ? 1u : 0u;
}
struct RawRow { U4 v; uint32_t hh; uint32_t hh2 = 0, p = 0; }; /* hh2, p: RG = 2 only */

template <bool INVERT = false, int RG = 0> struct Strip {
  static constexpr bool RAGGED = RG != 0;
  BufRsrc src, dst;
  unsigned w, h, x0, lane, band, gid;
  uint32_t col_off, halo_off; /* this lane's 16 B / its halo dword(s) inside a row (kOOB: none) */
  bool sel_l = false, sel_r = false; /* RAGGED: left / right neighbour dword comes from the halo load, not from the next lane */
  uint32_t halo_shift = 0;           /* RG 1: bits the right-halo dword was fetched early by */
  uint32_t keep_r = 0xffffffffu;     /* RG 1: 0 in the tail lane -- nothing lies right of it (as lane 63 its DPP fill is its LEFT halo) */
  uint32_t src_off = kOOB, bp = 0, mask_l = 0xffffffffu, mask_r = 0xffffffffu; /* RG 2: load column, base phase, in-image bytes of the outer dwords */
  GS_DEV Strip(const uint8_t *s, uint8_t *d, unsigned w_, unsigned h_, size_t frame_bytes)
      : src(make_buf(s + (size_t)blockIdx.z * (frame_bytes & ~kStripXcdFlag), frame_bytes & ~kStripXcdFlag)),
        dst(make_buf(d + (size_t)blockIdx.z * (frame_bytes & ~kStripXcdFlag), frame_bytes & ~kStripXcdFlag)), w(w_), h(h_) {
    lane = threadIdx.x & 63u;
    gid = blockIdx.x * blockDim.x + threadIdx.x;
    band = uniform(blockIdx.y * blockDim.y + threadIdx.y); /* same for the wave's 64 lanes: SGPR */
    if (frame_bytes & kStripXcdFlag) band = (blockIdx.y & 7u) * (gridDim.y >> 3) + (blockIdx.y >> 3);
    if constexpr (RG == 0) {
      x0 = gid * 16u;
      col_off = x0 < w ? x0 : kOOB;
      halo_off = kOOB;
      if (lane == 0 && x0 > 0 && x0 < w) halo_off = x0 - 4;
      if (lane == 63 && x0 + 16 < w) halo_off = x0 + 16;
    } else {
      const unsigned t = (w - 1u) >> 4, m = w & 15u;
      gid -= RG == 2 ? strip_realign_shift(w) : strip_ragged_shift(w); /* the idle lane wraps to 2^32 - 1: outside like everything right of the image */
      const bool tail = m != 0u && gid == t, in = gid <= t;
      x0 = tail ? w - 16u : gid * 16u;
      keep_r = tail ? 0u : 0xffffffffu;
      col_off = in ? x0 : kOOB;
      halo_off = kOOB;
      uint32_t right = kOOB; /* start of the dword right of this lane's 16 px, where the lane has to fetch it itself */
      if (in && x0 > 0 && (lane == 0 || tail)) halo_off = x0 - 4, sel_l = lane != 0;
      if (gid < t && (lane == 63 || (m != 0u && gid + 1u == t))) right = x0 + 16, sel_r = lane != 63;
      /* RG 2, whole strips: the row's last p pixels lie behind the last lane's 16 bytes; its halo slot is free unless it is lane 0 */
      if (RG == 2 && m == 0u && gid == t && lane != 0u) right = x0 + 16, sel_r = lane != 63;
      if constexpr (RG == 1) {
        if (right != kOOB) { /* in the row from `right` on (right <= 16 t < w); the dword may cross the row end */
          const uint32_t at = right + 4u > w ? w - 4u : right;
          halo_off = at, halo_shift = 8u * (right - at);
        }
      } else {
        if (right != kOOB) halo_off = right;
        const size_t fb = frame_bytes & ~kStripXcdFlag;
        const uint8_t *base = s + (size_t)blockIdx.z * fb;
        bp = (uint32_t)((uintptr_t)base & 3u);
        /* the dwords that hold the frame's first and last byte lie in the pages those bytes lie in; what is beyond reads 0 */
        src = make_buf(base - bp, (fb + bp + 3u) & ~(size_t)3u);
        src_off = in ? x0 : (strip_realign_helper(w) && gid == t + 1u) ? w : kOOB; /* the lane behind the tail lane: the 16 bytes that follow */
        mask_l = (in && x0 == 0u) ? 0u : 0xffffffffu;
        const int nv = (int)w - (int)(x0 + 16u); /* pixels x0 + 16 .. x0 + 19 inside the row */
        mask_r = !in || nv <= 0 ? 0u : nv >= 4 ? 0xffffffffu : (1u << (8 * nv)) - 1u;
      }
    }
  }
  /* the whole wave lies right of the image (its block is wider than the frame): nothing to load, compute or store */
  GS_DEV bool wave_outside() const {
    if constexpr (RAGGED) return (int)uniform(gid - lane) > (int)((w - 1u) >> 4) + (RG == 2 ? 1 : 0);
    else return uniform(x0 - lane * 16u) >= w;
  }
  GS_DEV bool in_image() const { return col_off != kOOB; }
  GS_DEV uint32_t row_off(int y, bool ok = true) const { /* y, ok wave-uniform */
    /* RG == 2 adds the base's byte phase bp (1..3) to the sum: three less keeps kRowOOB + bp + kOOB from wrapping to 0..2
     * (the frame's first bytes instead of the zero fill; those lanes' values were always masked, but the invariant above
     * should hold for every flavour) while row + bp + any valid column still lands beyond a frame of < 0x7fff0000 bytes */
    return (ok && (unsigned)y < h) ? (uint32_t)y * w : (RG == 2 ? kRowOOB - 3u : kRowOOB);
  }
  /* row y: this lane's 16 B; lane 0 also fetches the 4 B left of the wave's 1 KiB, lane 63 the
   * 4 B right of it (one shared instruction).  Everything outside the image reads 0. */
  GS_DEV RawRow load(int y) const {
    const uint32_t row = row_off(y);
    RawRow r;
    if constexpr (RG == 2) {
      const uint32_t a = row + bp + src_off; /* byte offset of the lane's first pixel from the dword-aligned base */
      r.v = buf_load16(src, a & ~3u);
      const U2 hx = buf_load8(src, (row + bp + halo_off) & ~3u); /* same phase: halo_off = x0 - 4 or x0 + 16 */
      r.hh = hx.x, r.hh2 = hx.y, r.p = a & 3u;
      if (INVERT) { /* every byte of an in-image row; what lies outside the row is masked in unpack() (stays 0 in the inverted domain) */
        const uint32_t m = (unsigned)y < h ? 0xffffffffu : 0u;
        r.v = U4{r.v.x ^ m, r.v.y ^ m, r.v.z ^ m, r.v.w ^ m};
        r.hh ^= m, r.hh2 ^= m;
      }
      return r;
    }
    r.v = buf_load16(src, row + col_off);
    r.hh = buf_load4(src, row + halo_off);
    if (INVERT) { /* complement in-image bytes only: out-of-range stays 0 in the inverted domain */
      const bool ok = (unsigned)y < h && in_image();
      const uint32_t m = ok ? 0xffffffffu : 0u, hm = (ok && halo_off != kOOB) ? 0xffffffffu : 0u;
      r.v = U4{r.v.x ^ m, r.v.y ^ m, r.v.z ^ m, r.v.w ^ m};
      r.hh ^= hm;
    }
    if constexpr (RG == 1) r.hh >>= halo_shift; /* the pixels past the row end: 0 (inverted domain included) */
    return r;
  }
  /* whole 16 B of row y (dropped when !ok or the lane is outside the image) */
  GS_DEV void store(int y, bool ok, U4 o) const {
    if (INVERT) o = U4{~o.x, ~o.y, ~o.z, ~o.w};
    buf_store16(dst, row_off(y, ok) + col_off, o);
  }
  /* 24 bytes = cols x0-4 .. x0+19 as 12 dwords of u16 pairs: U[j] = (px 2j-4, px 2j-3). */
  GS_DEV void unpack(const RawRow &r, uint32_t (&U)[12]) const {
    uint32_t L = wave_shr1(r.v.w, r.hh); /* left neighbour's last dword (lane 0: halo)   */
    uint32_t R = wave_shl1(r.v.x, r.hh); /* right neighbour's first dword (lane 63: halo) */
    if constexpr (RG == 2) {
      uint32_t R2 = wave_shl1(r.v.y, r.hh2);
      L = sel_l ? r.hh : L, R = sel_r ? r.hh : R, R2 = sel_r ? r.hh2 : R2;
      const uint32_t d0 = alignbyte(r.v.x, L, r.p) & mask_l, d1 = alignbyte(r.v.y, r.v.x, r.p), d2 = alignbyte(r.v.z, r.v.y, r.p),
                     d3 = alignbyte(r.v.w, r.v.z, r.p), d4 = alignbyte(R, r.v.w, r.p), d5 = alignbyte(R2, R, r.p) & mask_r;
      U[0] = unpack_lo(d0), U[1] = unpack_hi(d0);
      U[2] = unpack_lo(d1), U[3] = unpack_hi(d1);
      U[4] = unpack_lo(d2), U[5] = unpack_hi(d2);
      U[6] = unpack_lo(d3), U[7] = unpack_hi(d3);
      U[8] = unpack_lo(d4), U[9] = unpack_hi(d4);
      U[10] = unpack_lo(d5), U[11] = unpack_hi(d5);
      return;
    }
    if constexpr (RG == 1) L = sel_l ? r.hh : L, R = (sel_r ? r.hh : R) & keep_r;
    U[0] = unpack_lo(L), U[1] = unpack_hi(L);
    U[2] = unpack_lo(r.v.x), U[3] = unpack_hi(r.v.x);
    U[4] = unpack_lo(r.v.y), U[5] = unpack_hi(r.v.y);
    U[6] = unpack_lo(r.v.z), U[7] = unpack_hi(r.v.z);
    U[8] = unpack_lo(r.v.w), U[9] = unpack_hi(r.v.w);
    U[10] = unpack_lo(R), U[11] = unpack_hi(R);
  }
};

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
 * The input is requested one row ahead (two measured no better and costs 5 registers). */
template <int RING, bool INVERT, bool EXITS = true, int RG = 0, class Body, class Fin = NoFin>
GS_DEV void strip_rows(const Strip<INVERT, RG> &S, int y0, int nrows, int lead, RawRow first, Body &&body,
                       Fin fin = Fin()) {
  RawRow raw = first; /* = load(y0 + lead): the newest input row output row 0 needs */
#if GS_STRIP_PF == 2
  RawRow raw2 = S.load(y0 + lead + 1);
#endif
  U4 o_prev{0, 0, 0, 0};
  fin.prefetch(y0);
  int base = 0;
  for (; base < nrows; base += RING) {
    static_for<RING>([&](auto I) {
      const int i = base + decltype(I)::value;
      if constexpr (EXITS) {
        if (i >= nrows) return; /* wave-uniform */
      }
      if constexpr (!EXITS) sched_fence(); /* the next row's unpack (= its vmcnt wait) stays down here */
      uint32_t U[12];
      S.unpack(raw, U);
      S.store(y0 + i - 1, i > 0 && i <= nrows, fin(o_prev, y0 + i - 1));
#if GS_STRIP_PF == 2
      raw = raw2, raw2 = S.load(y0 + i + lead + 2);
#else
      raw = S.load(y0 + i + lead + 1);
#endif
      fin.prefetch(y0 + i);
      if constexpr (!EXITS) sched_fence(); /* one big block: keep the loads ahead of the arithmetic */
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
/* odd-aligned pairs of a row: A[j] = (px 2j-3, px 2j-2), j = 1..9 */
GS_DEV void sobel_apairs(const uint32_t (&U)[12], uint32_t (&A)[10]) {
#pragma unroll
  for (int j = 1; j <= 9; j++) A[j] = alignbit(U[j + 1], U[j], 16);
}
/* H1[k] = r[x-1] + 2 r[x] + r[x+1] for the own pairs k = 0..7 (<= 1020 per field) */
GS_DEV void sobel_h1(const uint32_t (&U)[12], const uint32_t (&A)[10], uint32_t (&H1)[8]) {
#pragma unroll
  for (int k = 0; k < 8; k++) H1[k] = pk_mad2_u16(U[k + 2], add2(A[k + 1], A[k + 2]));
}

struct SobelKeepCols { /* Fin functor of strip_rows */
  BufRsrc dst;
  unsigned w, x0;
  bool first, last;
  uint32_t keep_off;
  uint32_t e = 0; /* dst dword holding the protected byte of the row that is stored next */
  bool bytes = false; /* RG 2: one byte at its own address (a dword there would be a misaligned load) */
  template <int RG> GS_DEV SobelKeepCols(const Strip<false, RG> &S) : dst(S.dst), w(S.w), x0(S.x0) {
    first = S.in_image() && x0 == 0, last = S.in_image() && x0 + 16 == w; /* launcher guarantees w >= 32: never both; RAGGED: the tail lane ends at w */
    bytes = RG == 2;
    keep_off = first ? x0 : last ? x0 + (RG == 2 ? 15u : 12u) : kOOB;
  }
  /* strip_rows calls prefetch(y) in the iteration that COMPUTES row y and operator() in the next
   * one, right before row y is stored (and before the next prefetch): one iteration of latency */
  GS_DEV void prefetch(int y) {
    if (bytes) e = buf_load1(dst, (uint32_t)y * w + keep_off); /* compile-time choice: bytes = (RG == 2) */
    else e = buf_load4(dst, (uint32_t)y * w + keep_off); /* rows handed in are inside the frame */
  }
  GS_DEV U4 operator()(U4 o, int) const {
    o.x = first ? perm_b32(o.x, e, 0x07060500u) : o.x; /* byte 0 <- dst */
    o.w = last ? perm_b32(o.w, e, bytes ? 0x00060504u : 0x03060504u) : o.w;  /* byte 3 <- dst */
    return o;
  }
};

/* Vertical state of the sobel recurrence, one new input row b per step (output row y = b-1).
 * Everything is kept UNSIGNED so that sums are plain 32-bit adds on field pairs (full issue rate on
 * gfx950, prims.h add2) and only the two absolute differences need packed max / min:
 *   C(y)[x]  = r(b-2)[x] + 2 r(b-1)[x] + r(b)[x]   on the odd-aligned pairs A -- as T(b-1) + T(b)
 *              with T(b) = A(b-1) + A(b);            |gx| = |C[x+1] - C[x-1]| = absdiff(C_A[k+2], C_A[k+1])
 *   |gy|     = |H1(b) - H1(b-2)|,                     H1 = r[x-1] + 2 r[x] + r[x+1].
 * 34 registers of state (Ap, Tp: 9 + 9, H1 history 2 x 8), 2-step static rotation for H1 only.
 * (The signed form gx = Pa + H2(b), Pa' = H2(b-1) + 2 H2(b) needed 5 packed-16 ops per pair for
 * |gx|; this one needs 2 plus ~2 plain adds.) */
struct SobelState {
  uint32_t Ap[10], Tp[10], H1[2][8];
  /* prime with input rows y0-1 (r0) and y0 (r1) */
  GS_DEV void init(const uint32_t (&U0)[12], const uint32_t (&U1)[12]) {
    uint32_t A0[10];
    sobel_apairs(U0, A0);
    sobel_apairs(U1, Ap);
    sobel_h1(U0, A0, H1[0]);
    sobel_h1(U1, Ap, H1[1]);
#pragma unroll
    for (int j = 1; j <= 9; j++) Tp[j] = add2(A0[j], Ap[j]);
  }
  /* |gx| + |gy| (<= 2040 per field) for the 8 own pairs; PAR = parity of the step: H1[PAR] holds
   * row b-2 and receives row b */
  struct NoHook { GS_DEV void operator()(int, uint32_t &) const {} };
  /* hook(j, C[j]) is called as soon as C[j] exists (the fused kernel threads an LDS atomic of the
   * previous row through it, see k_fused.h) */
  template <int PAR, class Hook = NoHook> GS_DEV void magnitude(const uint32_t (&U)[12], uint32_t (&m)[8], Hook hook = Hook()) {
    uint32_t A[10], C[10], H1n[8];
    sobel_apairs(U, A);
#pragma unroll
    for (int j = 1; j <= 9; j++) {
      const uint32_t Tn = add2(Ap[j], A[j]);
      C[j] = add2(Tp[j], Tn);
      Tp[j] = Tn, Ap[j] = A[j];
      hook(j, C[j]);
    }
    sobel_h1(U, A, H1n);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      m[k] = add2(absdiff2(C[k + 2], C[k + 1]), absdiff2(H1n[k], H1[PAR][k]));
      H1[PAR][k] = H1n[k];
    }
  }
  /* Bytes only: (|gx|+|gy|)/2 clamped to 255 is the high byte of min((|gx|+|gy|) * 128, 65535),
   * one saturating v_pk_mad_u16 instead of shift + min. */
  template <int PAR> GS_DEV U4 step(const uint32_t (&U)[12]) {
#if GS_SOBEL_SAT
    uint32_t M[8];
    magnitude<PAR>(U, M);
#pragma unroll
    for (int k = 0; k < 8; k++) M[k] = pk_shl7_sat_u16(M[k]);
    return U4{pack_lohi_b1(M[0], M[1]), pack_lohi_b1(M[2], M[3]), pack_lohi_b1(M[4], M[5]),
              pack_lohi_b1(M[6], M[7])};
#else
    uint32_t M[8];
    return step<PAR>(U, M);
#endif
  }
  /* M: the 16 results as u16 pairs (before packing to bytes) */
  template <int PAR, class Hook = NoHook> GS_DEV U4 step(const uint32_t (&U)[12], uint32_t (&M)[8], Hook hook = Hook()) {
    magnitude<PAR>(U, M, hook);
#pragma unroll
    for (int k = 0; k < 8; k++) M[k] = pk_min_u16(pk_shr_u16(M[k], 1), 0x00ff00ffu);
    return U4{pack_lohi(M[0], M[1]), pack_lohi(M[2], M[3]), pack_lohi(M[4], M[5]),
              pack_lohi(M[6], M[7])};
  }
  /* same, H1 history shifted instead of alternated (for callers whose unroll period is odd) */
  GS_DEV void shift_history() {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t t = H1[0][k];
      H1[0][k] = H1[1][k], H1[1][k] = t;
    }
  }
  template <class Hook = NoHook> GS_DEV U4 step_shift(const uint32_t (&U)[12], uint32_t (&M)[8], Hook hook = Hook()) {
    const U4 o = step<0>(U, M, hook); /* H1[0] (row b-2) consumed and overwritten with row b */
    shift_history();
    return o;
  }
  GS_DEV U4 step_shift(const uint32_t (&U)[12]) { /* bytes only */
    const U4 o = step<0>(U);
    shift_history();
    return o;
  }
};

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
    P[j] = add2(U[j], A[j]);             /* 2-px sums (x, x+1) for both halves */
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int j = k + 2; /* U[j] = own pair k */
    if constexpr (R == 1) H[k] = add2(A[j - 1], P[j]);
    else if constexpr (R == 2) H[k] = add2(P[j - 1], P[j], U[j + 1]);
    else H[k] = add2(add2(A[j - 2], P[j - 1], P[j]), P[j + 1]);
  }
}

}  // namespace gs
#endif
